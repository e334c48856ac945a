/* rdmi.h -- C ABI of librdmi.so: the MI355X-native (gfx950) NCSN++ score network and
 * reflected predictor-corrector sampler.
 *
 * The reference (sriramelango/optimized-diffusion-model) is 100 % Python/PyTorch and has no
 * FFI of its own; its boundary for this path is a set of Python closures.
 * Each entry point below names the reference closure/method it sits under
 * ("RD/" = Reflected-Diffusion/ in the reference tree).  The Python mirror of those
 * closures lives in optimized-diffusion-model_amd/rdmi/ and binds these symbols with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32 unless it
 *    says "host"; tensors are dense, row-major, in the reference's own layouts
 *    (x / score: [B, C=1, H, W]; labels: [B, num_classes]; sigma / t / weight: [B]).
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream); the caller keeps the buffers alive until the stream is synchronised.
 *  - return value: 0 = ok, non-zero = error; rdmi_last_error() gives the message
 *    (thread-local).  Nothing falls back to a CPU path.
 */
#ifndef RDMI_H
#define RDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDMI_MAX_LEVELS 8

/* Architecture keys read by NCSNpp.__init__ (RD/models/ncsnpp.py:51-93). */
typedef struct rdmi_arch {
    int nf;                         /* config.model.nf                                  */
    int n_levels;                   /* len(config.model.ch_mult)                        */
    int ch_mult[RDMI_MAX_LEVELS];   /* config.model.ch_mult                             */
    int num_res_blocks;             /* config.model.num_res_blocks                      */
    int attn_levels;                /* bit i set <=> image_size // 2**i in attn_resolutions */
    int channels;                   /* config.model.channels (1)                        */
    int num_classes;                /* config.model.num_classes (label_emb fan-in)      */
    int conditional;                /* config.model.conditional                         */
    int scale_by_sigma;             /* config.model.scale_by_sigma                      */
    float fourier_2pi_prescaled;    /* reserved, must be 0                              */
    int compute_dtype;              /* 0: fp32 (the reference's arithmetic); 1: bf16 MFMA operands with fp32 accumulate, fp32
                                       tensors and fp32 GroupNorm / softmax (tiled plan only: BASELINE config #5) */
} rdmi_arch;

typedef struct rdmi_ctx rdmi_ctx;

/* Create a context for model batches up to max_batch samples of H x W pixels on the current
 * HIP device; allocates the activation workspace, packed-weight arena and launch plan.
 * Replaces: NCSNpp.__init__ + .to(device) (RD/models/ncsnpp.py:42-224) for the compute side;
 * parameters stay owned by the Python module (see rdmi_set_param). */
int rdmi_create(const rdmi_arch* arch, int max_batch, int H, int W, rdmi_ctx** out);
int rdmi_destroy(rdmi_ctx* ctx);

/* Number of parameter tensors the context expects and the i-th one's reference state-dict
 * name / element count (e.g. "down_blocks.2.NIN_0.W", 8192).  Order = the reference's
 * registration order (what EMA and Adam iterate, RD/models/ema.py:28). */
int rdmi_num_params(const rdmi_ctx* ctx);
int rdmi_param_info(const rdmi_ctx* ctx, int index, const char** name, size_t* numel);

/* Bind a parameter by its reference state-dict name to a BORROWED device pointer in the
 * reference's layout (conv OIHW, Linear [out,in], NIN.W [in,out]).  No copy is kept in that
 * layout: EMA copy_to/restore and optimizer steps written through the same storage are seen by
 * the next call.  Replaces nn.Module parameter registration. */
int rdmi_set_param(rdmi_ctx* ctx, const char* name, const float* dev_ptr, size_t numel);

/* Re-read every bound parameter into the kernels' packed MFMA layout (one launch).  Called by
 * the entry points below unless RDMI_PARAMS_CACHED is passed. */
int rdmi_repack(rdmi_ctx* ctx, void* stream);

#define RDMI_PARAMS_CACHED 1u   /* caller guarantees parameters are unchanged since the last call */

/* out[B,1,H,W] = NCSNpp.forward(x, time_cond=sigma, class_labels=labels) in eval mode.
 * Replaces: model_fn of get_model_fn (RD/models/utils.py:66-82) -> NCSNpp.forward
 * (RD/models/ncsnpp.py:226-354).  labels may be NULL only if arch.conditional == 0. */
int rdmi_forward(rdmi_ctx* ctx, const float* x, const float* sigma, const float* labels, float* out,
                 int B, unsigned flags, void* stream);

/* score = model(x, sigma(t), labels) with sigma(t) = sigma_min*(sigma_max/sigma_min)^t.
 * Replaces: score_fn of get_score_fn (RD/models/utils.py:100-103) + RVESDE.marginal_prob
 * (RD/sde_lib.py:142-145). */
int rdmi_score(rdmi_ctx* ctx, const float* x, const float* t, const float* labels, float* out, int B,
               double sigma_min, double sigma_max, unsigned flags, void* stream);

/* Classifier-free-guidance score: one forward at 2B ([x;x], [t;t], [labels;0]) then
 * (1+w)*s_cond - w*s_uncond.  weight: device [B] or NULL (=0).
 * Replaces: weighted_score_fn of get_cf_score_fn (RD/models/utils.py:120-138). */
int rdmi_cf_score(rdmi_ctx* ctx, const float* x, const float* t, const float* labels, const float* weight,
                  float* out, int B, double sigma_min, double sigma_max, unsigned flags, void* stream);

/* cube.reflect (RD/cube.py:34-49), elementwise, in == out allowed. */
int rdmi_reflect(const float* in, float* out, size_t n, void* stream);

/* cube.score_hk (RD/cube.py:149-193): score of the reflected heat kernel started at x_orig with
 * std sigma[B]; per-sample switch between the eigenfunction series and the image sum. */
int rdmi_score_hk(const float* x, const float* x_orig, const float* sigma, float* out, int B, int elems_per_sample,
                  int efs, int refls, float min_cutoff, void* stream);

/* Score-matching loss pieces (RD/losses.py:79-93, get_sde_loss_fn/loss_fn):
 *   rdmi_perturb : perturbed = reflect(batch + sigma(t) * z)                                   (:81-82)
 *   rdmi_sm_loss : per_sample[b] = reduce( w_b * (score - score_hk(perturbed, batch, sigma_b))^2 ) with w = sigma^2
 *                  (g(t)^2 if likelihood_weighting) and reduce = 0.5*sum (mean if reduce_mean)  (:84-92);
 *                  the batch mean (:93) is left to the caller. */
int rdmi_perturb(const float* batch, const float* z, const float* t, float* out, int B, int elems_per_sample,
                 double sigma_min, double sigma_max, void* stream);
int rdmi_sm_loss(const float* score, const float* perturbed, const float* batch, const float* t, float* per_sample,
                 float* dscore /* NULL or [B,E]: d per_sample / d score for the backward */, int B, int elems_per_sample,
                 double sigma_min, double sigma_max, int likelihood_weighting, int reduce_mean, void* stream);

/* Post-sampling un-normalisation of the GTO-Halo samples (Benchmark/gto_halo_benchmarking.py:255-333, :335-361):
 * samples [N, row_elems] (the flattened [N,1,9,9] sampler output, row_elems >= 67) -> out [N, 67] physical vectors
 * (halo energy, shooting/coast times, 20 x (alpha, theta, |u| clipped to 1), fuel mass, halo period, manifold length);
 * *clip_count (device, may be NULL; caller zeroes it) += number of control magnitudes that exceeded 1. */
int rdmi_gto_unnormalize(const float* samples, float* out, unsigned long long* clip_count, int N, int row_elems, void* stream);

/* Training-data side (RD/datasets.py:82-98 GTOHaloImageDataset.__getitem__): gather rows idx[0..B) (NULL: rows
 * 0..B-1) of a device-resident table data[rows][row_len] into images [B, elems] = (zero-pad(vec, elems) - mean) / std and
 * labels [B] = vec[0] (un-normalised), replacing the per-item np.pad loader.  Indices are the caller's to keep in range. */
int rdmi_gto_pack(const float* data, const long long* idx, float* images, float* labels, int B, int row_len, int elems,
                  double mean, double std, void* stream);

/* Training step (RD/losses.py:141-149): train-mode forward of NCSNpp (Dropout_0 with probability dropout_p on the input of
 * every Conv_1, RD/models/layerspp.py:204; label drop is the caller's, RD/models/ncsnpp.py:242-246) keeping every
 * activation, and the backward pass.  rdmi_enable_training switches the context to the layer plan with per-tensor storage
 * and allocates gradient workspace (use a dedicated context for training).  rdmi_backward writes d loss / d parameter for
 * ALL parameters into grads_flat in the reference's parameter order (offset of parameter i = sum of numel of 0..i-1;
 * the frozen time_embed.W slice stays zero); `x` is the network input of the forward call. */
int rdmi_enable_training(rdmi_ctx* ctx);
int rdmi_train_forward(rdmi_ctx* ctx, const float* x, const float* sigma, const float* labels, float* out, int B,
                       float dropout_p, uint64_t seed, void* stream);
int rdmi_backward(rdmi_ctx* ctx, const float* grad_out, float* grads_flat, size_t grads_numel, const float* x, void* stream);
/* Diagnostic: the train-mode forward and the backward are recorded as launch graphs on their second call and replayed from then on
 * (RDMI_TRAIN_GRAPH=0: plain launches): how many recordings and replays this context has made. */
int rdmi_train_graph_stats(rdmi_ctx* ctx, long* records, long* replays);

/* One reflected Euler-Maruyama update given the score (RD/sampling.py:198-207 with
 * RSDE.sde, RD/sde_lib.py:93-101): x_mean = x + g(t)^2*score/N, x' = x_mean + g(t)*sqrt(1/N)*z,
 * both reflected.  t: device [B]; x_mean_out may be NULL. */
int rdmi_em_update(const float* x, const float* score, const float* z, const float* t, float* x_out,
                   float* x_mean_out, int B, int elems_per_sample, int N, double sigma_min, double sigma_max,
                   void* stream);

/* One reflected Langevin corrector step given the score (RD/sampling.py:222-231); the step
 * size uses the means over THIS batch of ||score_b|| and ||z_b||.  scratch: device [2*B+2]. */
int rdmi_langevin_update(const float* x, const float* score, const float* z, float* x_out, float* x_mean_out,
                         float* scratch, int B, int elems_per_sample, float snr, void* stream);

/* Sampler options (config.sampling.* and config.sde.*, RD/sampling.py:102-124, RD/run_vis.py:32-36). */
typedef struct rdmi_pc_opts {
    int N;                 /* sde.N (num_scales): N-1 updates are applied          */
    float eps;             /* last time of linspace(T=1, eps, N)                                */
    double sigma_min, sigma_max; /* python floats in the reference (sde_lib.py:122-123): kept in double */
    float snr;             /* config.sampling.snr                                               */
    int n_steps_each;      /* corrector inner steps                                             */
    int corrector;         /* 0 = none, 1 = langevin                                            */
    int use_cfg;           /* 1: class_labels given -> get_cf_score_fn (2B forward); 0: get_score_fn */
    uint64_t seed;         /* Philox seed for in-kernel N(0,1) noise when noise == NULL         */
    uint64_t seq_offset;   /* Philox stream offset (e.g. rank * B) so shards draw disjoint noise */
} rdmi_pc_opts;

/* The whole PC sampling loop on the device: x (in: the prior draw, out: the sample) is updated
 * N-1 times by [corrector x n_steps_each, predictor].  noise: NULL (Philox in-kernel) or device
 * [(N-1)*(n_steps_each*corrector+1), B, H*W] tensors in consumption order (parity testing).
 * trace: NULL or device [N-1, B, H*W] receiving x after every predictor update.
 * teacher: NULL or device [N-1, B, H*W]: update i+1 restarts from teacher[i] (parity testing).
 * Replaces: pc_sampler's hot loop (RD/sampling.py:322-337). */
int rdmi_pc_sample(rdmi_ctx* ctx, float* x, const float* labels, const float* weight, const float* noise,
                   float* trace, const float* teacher, int B, const rdmi_pc_opts* opts, unsigned flags,
                   void* stream);

/* ---- probability-flow ODE sampler ------------------------------------------------------------
 * Replaces get_ode_sampler's ode_sampler (RD/sampling.py:342-392): scipy.integrate.solve_ivp(method='RK45') over
 * drift_fn(score_fn, x, t) * bump(x), integrated from T to eps with the adaptive Dormand-Prince 5(4) scheme.  State, stages,
 * stage combinations, the error norm and the score network all stay on the device (float64 state like scipy, float32
 * right-hand side like the reference); the host carries only the scalar step controller.  SYNCHRONOUS (the controller
 * reads one scalar per attempted step), like the scipy loop it replaces. */
typedef struct {
    double T, eps;             /* integrate t from T down to eps (sde.T, sampler eps)                     */
    double rtol, atol;         /* solve_ivp tolerances (reference: 1e-5, 1e-5)                            */
    double sigma_min, sigma_max;
    float moll;                /* bump mollifier strength (config.sampling.moll); <= 0: bump(x) = x       */
    int use_cfg;               /* 1: class_labels given -> classifier-free-guidance score (2B forward)    */
    double first_step;         /* > 0: initial |h| (solve_ivp first_step); 0: scipy's select_initial_step */
    int max_steps;             /* > 0: stop after this many accepted steps (parity tests); 0: run to eps  */
    double* h_next_out;        /* NULL or host double receiving the controller's next |h|                 */
} rdmi_ode_opts;
/* x: in = the initial state (the reference's (1-2 side_eps) U[0,1] + side_eps draw, or z), out = y(eps) as float32.
 * nfev_out: number of right-hand-side evaluations (solution.nfev).  t_final: NULL or host double, the time reached. */
int rdmi_ode_sample(rdmi_ctx* ctx, float* x, const float* labels, const float* weight, int B, const rdmi_ode_opts* opts,
                    int* nfev_out, double* t_final, unsigned flags, void* stream);

/* ---- multi-tensor optimizer step -------------------------------------------------------------
 * Replaces, for all parameters in three launches: torch.nn.utils.clip_grad_norm_ + optimizer.step() of
 * optimize_fn (RD/losses.py:29-47: Adam / AdamW, get_optimizer :12-23) and ExponentialMovingAverage.update
 * (RD/models/ema.py:32-52).  Tensors stay in torch's storage: a slot is the five device pointers of one
 * parameter (ema may be NULL: no EMA for that tensor).  rdmi_opt_create copies the table; rebuild it when
 * a pointer changes. */
typedef struct rdmi_opt rdmi_opt;
typedef struct {
    float* param; float* grad; float* exp_avg; float* exp_avg_sq; float* ema;
    unsigned long long numel;
} rdmi_opt_slot;
typedef struct {
    float lr, beta1, beta2, eps, weight_decay;  /* this step's learning rate (after warm-up) etc., fp32 copies   */
    double lr_d, beta1_d, beta2_d;              /* the same as python floats: bias corrections are formed in double */
    int decoupled_wd;                           /* 0 Adam (L2 into the gradient), 1 AdamW                          */
    int step;                                   /* 1-based update count t (bias correction 1 - beta^t)             */
    float max_norm;                             /* clip_grad_norm_ max_norm; < 0 disables clipping                 */
    double ema_decay_d;                         /* this update's min(decay, (1+n)/(10+n)); ignored where ema NULL  */
    int write_back_grad;                        /* 1: leave the clipped gradient in grad, as clip_grad_norm_ does  */
} rdmi_opt_hyper;
int rdmi_opt_create(const rdmi_opt_slot* slots_host, int n_slots, rdmi_opt** out);
/* total_norm_out: NULL or a device float receiving the pre-clip global gradient norm (clip_grad_norm_'s return). */
/* The same tensors at new addresses (e.g. this step's gradients landed in a fresh buffer): one asynchronous upload of the table,
 * no allocation, no synchronisation.  Slot count and element counts must be those of rdmi_opt_create. */
int rdmi_opt_update_slots(rdmi_opt* opt, const rdmi_opt_slot* slots_host, int n_slots, void* stream);
int rdmi_opt_step(rdmi_opt* opt, const rdmi_opt_hyper* hyper, float* total_norm_out, void* stream);
int rdmi_opt_destroy(rdmi_opt* opt);

/* Copy an internal activation (by reference module name, e.g. "down_blocks.0", "temb") of the LAST
 * forward to dst as [B, C, H, W]; returns its C,H,W.  Debug / parity-test hook. */
int rdmi_get_tap(rdmi_ctx* ctx, const char* name, float* dst, size_t dst_numel, int* C, int* H, int* W,
                 void* stream);

/* Per-kernel timing of the last rdmi_forward/rdmi_pc_sample when enabled (HIP events on `stream`). */
int rdmi_set_profiling(rdmi_ctx* ctx, int enabled);
int rdmi_get_profile(rdmi_ctx* ctx, int index, const char** kernel_name, double* total_ms, long* launches,
                     double* flops_per_launch);

/* Which execution plan the context uses ("fused: ..." or "layers: ... (reason)"). */
const char* rdmi_path_info(rdmi_ctx* ctx);

/* Diagnostic (RDMI_STAMPS=1 at create): shader-clock cycles workgroup 0 spent in each op of the fused program
 * during the last forward, with a one-line description per op.  Returns the op count. */
int rdmi_debug_op_cycles(rdmi_ctx* ctx, long long* host_cycles, int cap, const char** desc, int desc_cap);

/* Diagnostic: has any co-operative launch of this context (groups of four workgroups sharing the low-resolution section of the
 * fused U-Net: DESIGN.md 4.2d) given up one of its bounded inter-workgroup waits?  *gave_up = 0: never; 1: yes -- the output
 * samples of the affected workgroups were overwritten with NaN by the kernel itself; from then on the context stops selecting the
 * co-operative program (its workgroups were evidently not resident together on this device).  Synchronises the device. */
int rdmi_coop_status(rdmi_ctx* ctx, int* gave_up);

/* Diagnostic: the N(0,1) draws the fused sampler makes when rdmi_pc_sample is called with noise == NULL
 * (replaces torch.randn_like at RD/sampling.py:200,224): z[i], i < n, = Philox4x32-10 + Box-Muller value of global element
 * elem_offset + i of noise tensor number `draw` under `seed` -- rdmi_pc_sample uses elem_offset = seq_offset * H*W*C + b * H*W*C + e
 * and numbers the noise tensors 0, 1, ... in the order the reference would draw them.  Lets tests check the stream's
 * distribution and feed the very same values to the oracle. */
int rdmi_philox_normal(float* z, size_t n, unsigned long long seed, unsigned long long elem_offset, unsigned draw, void* stream);

const char* rdmi_last_error(void);
const char* rdmi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RDMI_H */
