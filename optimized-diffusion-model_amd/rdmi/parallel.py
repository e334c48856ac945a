"""Multi-GPU sampling: batch sharding + ONE all-gather.

The reference's multi-GPU sampling is implicit: under mp.spawn every rank samples its own batch//ngpus shard,
saves sample_{rank}.npy and hits dist.barrier() (Reflected-Diffusion/run_train.py:124-129,181,189-191); shards are
never merged.  Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r owns
trajectories [r*B, (r+1)*B), no per-update collective (trajectories are independent given the weights), and a
single all-gather of the [B,1,H,W] fp32 shards (41 KB per GPU at B=128: latency-bound) at the end of the call.

Langevin caveat: the corrector's step size is a batch mean, so with corrector='langevin' each rank
uses the mean over ITS shard == the reference run at batch B per rank; with corrector='none' (the shipped
default) the sharded result is identical to the unsharded one.
"""
import torch
import torch.distributed as dist

from . import sampling


def sharded_sampling_fn(config, sde, shape, eps, device, seed=None, rank=None, world=None, group=None):
    """`shape` is the per-rank shape [B, C, H, W]; returns sampling_fn(model, ...) -> (x[world*B, C, H, W], nfe)."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if config.sampling.method.lower() != 'pc':
        raise ValueError('sharded sampling is built for the pc sampler')
    fn = sampling.get_pc_sampler(sde=sde, shape=shape,
                                 predictor=sampling.get_predictor(config.sampling.predictor.lower()),
                                 corrector=sampling.get_corrector(config.sampling.corrector.lower()),
                                 denoiser=sampling.get_denoiser(config.sampling.denoiser.lower()),
                                 snr=config.sampling.snr, n_steps=config.sampling.n_steps_each, eps=eps, device=device,
                                 seed=seed, shard=(rank, world))

    def sampling_fn(model, z=None, noise_removal_model=None, weight=0, class_labels=None):
        x, nfe = fn(model, z=z, noise_removal_model=noise_removal_model, weight=weight, class_labels=class_labels)
        if world == 1:
            return x, nfe
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out, nfe

    return sampling_fn
