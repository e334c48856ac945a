"""Counterpart of the reference's callers of the hot path.

The reference's own drivers cannot run here or on the GPU box (hydra / omegaconf / torchvision are absent):
  * run_vis.visualize                    (Reflected-Diffusion/run_vis.py:25-87)
  * run_train._run_single snapshot block (Reflected-Diffusion/run_train.py:243-245,272-282)
  * GTOHaloBenchmarker.generate_samples  (Benchmark/gto_halo_benchmarking.py:212-257)
They all do the same thing around the path: build RVESDE(sigma_min, sigma_max, N=num_scales), eps=1e-5,
shape=(B,1,S,S), get_sampling_fn, swap the EMA weights in (store / copy_to ... restore), call
sampling_fn(model, weight=w, class_labels=labels), and flatten (N,1,9,9) -> (N,81)[:, :67].  This module restates
that glue over the rdmi API so the path can be driven (and timed the way the Benchmark harness times it).
"""
import time

import torch

from . import sampling, sde_lib


def generate_samples(model, ema, config, num_samples, batch_size, device, guidance_weight=0.0, labels='uniform',
                     n_valid=67, unnormalize=False):
    """Returns (samples [num_samples, n_valid] on the CPU, per-batch wall times).  labels: 'uniform' (Benchmark:
    U[0,1](B,1)), 'zeros' (run_train snapshot, run_train.py:275) or a tensor [num_samples, num_classes].
    unnormalize=True additionally applies the Benchmark's un-normalisation to physical 67-vectors
    (gto_halo_benchmarking.py:255-333) on the device before the copy and returns (samples, times, n_spherical_clips)."""
    sde = sde_lib.RVESDE(sigma_min=config.sde.sigma_min, sigma_max=config.sde.sigma_max, N=config.sde.num_scales)
    S, Wd = config.model.image_size, getattr(config.model, 'image_width', config.model.image_size)
    out, times = [], []
    done = 0
    clips = 0
    while done < num_samples:
        B = min(batch_size, num_samples - done)
        shape = (B, config.model.channels, S, Wd)
        sampling_fn = sampling.get_sampling_fn(config, sde, shape, 1e-5, device)
        if isinstance(labels, str):
            lab = torch.rand(B, 1, device=device) if labels == 'uniform' else torch.zeros(B, 1, device=device)
        else:
            lab = labels[done:done + B].to(device)
        t0 = time.time()
        if ema is not None:
            ema.store(model.parameters())
            ema.copy_to(model.parameters())
        sample, nfe = sampling_fn(model, weight=guidance_weight, class_labels=lab)
        if ema is not None:
            ema.restore(model.parameters())
        if unnormalize:
            phys, c = unnormalize_gto(sample)
            clips += int(c.item())
            times.append(time.time() - t0)
            out.append(phys.cpu())
        else:
            sample = sample.cpu()                     # the reference's only sync point
            times.append(time.time() - t0)
            out.append(sample.reshape(B, -1)[:, :n_valid])
        done += B
    if unnormalize:
        return torch.cat(out, 0), times, clips
    return torch.cat(out, 0), times


def unnormalize_gto(samples):
    """[N,1,9,9] (or [N,>=67]) sampler output on the device -> ([N,67] physical vectors on the device, clip-count tensor).
    HIP counterpart of the numpy block in GTOHaloBenchmarker.generate_samples (:255-333) + _convert_to_spherical (:335-361)."""
    from . import _native
    return _native.gto_unnormalize(samples)
