// Small kernels around the conv/attention core: weight repacking, the time/label embedding GEMMs,
// the sampler's elementwise updates (CFG combine, reflected Euler-Maruyama, reflected Langevin),
// Philox noise and layout copies.
#pragma once
#include "common.h"

// --------------------------------------------------------------------------------------------
// Weight repack: reference layouts -> [tap][K/16][Npad][16] (K = input channel, N = output channel).
// One launch for every parameter: blockIdx.y = job.
// --------------------------------------------------------------------------------------------
struct PackJob {
    const float* src;
    float* dst;
    int Cin, Cout;            // real dims of the source
    int Kpad;                 // padded input channels (multiple of 16)
    int Npad;                 // padded output channels OF THE DESTINATION (may hold several jobs side by side)
    int n_off;                // first destination column of this job
    int ntap;
    long s_co, s_ci, s_t;     // source strides (elements) of output channel, input channel, tap
    int kind;                 // 0: pack weights, 1: plain copy of Cout floats to dst + n_off, 2: pack weights as bf16 [tap][K/32][Npad][32],
                              // 3: pack the PRODUCT src (W [Cin][Cout], NIN layout) x src2 (W [Cout][Cout]), 4: the vector src (Cout floats) x src2 -> dst + n_off
    const float* src2;        // kinds 3 / 4: the right-hand factor (a NIN weight [in][out])
    int col_il;               // kind 2, tiled plan: 0 / 1 none; F = 2 / 4: destination columns are interleaved within groups of 16 F -- column g sits at
                              // position (g / 16F) * 16F + (g % F) * 16 + (g % 16F) / F, so that lane l of the F adjacent 16-column MFMA tiles of a
                              // group holds the F ADJACENT output columns 16F * group + F l .. + F - 1 (tconv_epilogue's vector stores)
};

__global__ __launch_bounds__(RDMI_THREADS) void pack_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int stride = gridDim.x * RDMI_THREADS;
    if (j.kind == 1) {
        for (int i = blockIdx.x * RDMI_THREADS + threadIdx.x; i < j.Cout; i += stride) j.dst[j.n_off + i] = j.src[i];
        return;
    }
    if (j.kind == 4) {        // row vector x matrix: the bias of a projection folded into the next one (rdmi.hip: attention value path)
        for (int i = blockIdx.x * RDMI_THREADS + threadIdx.x; i < j.Cout; i += stride) {
            float acc = 0.f;
            for (int k = 0; k < j.Cin; ++k) acc = fmaf(j.src[k], j.src2[(long)k * j.Cout + i], acc);
            j.dst[j.n_off + i] = acc;
        }
        return;
    }
    if (j.kind == 2) {        // bf16 copy of the weights for v_mfma_f32_16x16x32_bf16: 32 consecutive k per (k-slab, column)
        const int ncol = (j.Cout + 15) & ~15;
        bf16_t* d16 = reinterpret_cast<bf16_t*>(j.dst);
        const long total = (long)j.ntap * (j.Kpad >> 5) * ncol * 32;
        for (long i = blockIdx.x * RDMI_THREADS + threadIdx.x; i < total; i += stride) {
            const int kk = (int)(i & 31);
            long r = i >> 5;
            const int co = (int)(r % ncol); r /= ncol;
            const int ch = (int)(r % (j.Kpad >> 5));
            const int t = (int)(r / (j.Kpad >> 5));
            const int ci = ch * 32 + kk;
            float v = 0.f;
            if (co < j.Cout && ci < j.Cin) v = j.src[co * j.s_co + ci * j.s_ci + t * j.s_t];
            int pos = j.n_off + co;
            if (j.col_il > 1) { const int F = j.col_il, w = pos % (16 * F); pos = pos - w + (w % F) * 16 + w / F; }
            d16[(((long)t * (j.Kpad >> 5) + ch) * j.Npad + pos) * 32 + kk] = f2bf(v);
        }
        return;
    }
    // destination element (t, ch, co_local, kk): each job owns the columns [n_off, n_off + ceil16(Cout))
    const int ncol = (j.Cout + 15) & ~15;
    const long total = (long)j.ntap * (j.Kpad >> 4) * ncol * 16;
    for (long i = blockIdx.x * RDMI_THREADS + threadIdx.x; i < total; i += stride) {
        const int kk = (int)(i & 15);
        long r = i >> 4;
        const int co = (int)(r % ncol); r /= ncol;
        const int ch = (int)(r % (j.Kpad >> 4));
        const int t = (int)(r / (j.Kpad >> 4));
        const int ci = ch * 16 + kk;
        float v = 0.f;
        if (co < j.Cout && ci < j.Cin) {
            if (j.kind == 3) {                            // (W W2)[ci][co]; the middle dimension is W's output = W2's input = Cout channels
                for (int k = 0; k < j.Cout; ++k) v = fmaf(j.src[(long)ci * j.s_ci + k * j.s_co], j.src2[(long)k * j.Cout + co], v);
            } else v = j.src[co * j.s_co + ci * j.s_ci + t * j.s_t];
        }
        j.dst[(((long)t * (j.Kpad >> 4) + ch) * j.Npad + j.n_off + co) * 16 + kk] = v;
    }
}

// --------------------------------------------------------------------------------------------
// Y[M][N] = pre(X)[M][K] . W + bias (+ label embedding), W packed [K/16][Npad][16]; MFMA 16x16x4.
//   pre = 0: X as is; 1: SiLU(X); 2: Gaussian Fourier features of log(sigma)  (RD/models/layerspp.py:26-28)
// Used for time_mlp.0, time_mlp.2 (+label_emb) and the 17 concatenated Dense_0 projections
// (RD/models/ncsnpp.py:252-262, RD/models/layerspp.py:202).
// grid = (ceil(M/16), Npad/64): a workgroup is one 16-row tile x 64 columns, one column tile per wave.
// --------------------------------------------------------------------------------------------
struct LinArgs {
    const float* X; int ldx;
    const float* W; int Npad;
    const float* bias;
    float* Y; int ldy;
    int M, N, K;
    int pre;
    // pre == 1: X -> SiLU(X);  pre == 3: X row (ldx may be 0: one row shared by all samples) + label embedding -> SiLU
    //           (the sampler's precomputed time-MLP row, see rdmi_pc_sample);
    // pre == 2: X = sigma (or sde-time t when t_is_time) of sample (row % x_mod)
    const float* fourW; int nfour; int x_mod; int t_is_time; float smin, ratio;
    int use_scalar; float t_scalar;                 // X == null: every row uses this one time (the PC loop, RD/sampling.py:329)
    // optional label embedding added in the epilogue: labels [label_rows][ncls] (rows beyond are zero labels)
    const float* labels; const float* Wl; const float* bl; int ncls; int label_rows;
    // pre == 3 with row_div > 0: row r is (update r / row_div, sample r % row_div): X row = r / row_div, labels of sample r % row_div
    // (the sampler's Dense_0 outputs for a whole chunk of updates in one launch, see rdmi_pc_sample)
    int row_div;
};

__device__ __forceinline__ float sigma_of(float v, int t_is_time, float smin, float ratio) {
    // RVESDE.marginal_prob std (RD/sde_lib.py:143): sigma_min * (sigma_max/sigma_min) ** t
    // (ratio = fp32(sigma_max/sigma_min) is formed on the host in double, like the python float it is)
    return t_is_time ? smin * powf(ratio, v) : v;
}

// NT column tiles per wave (the workgroup covers 64*NT columns): the A fragment -- which may carry a Fourier / SiLU / label
// prologue -- is formed once per k-chunk and reused NT times.
template <int NT, int UNR = 4>
__global__ __launch_bounds__(RDMI_THREADS) void linear_mfma_kernel(LinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int row = blockIdx.x * 16 + lrow;
    int col[NT], lcol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        col[t] = blockIdx.y * (64 * NT) + (wave * NT + t) * 16 + lrow;
        lcol[t] = min(col[t], a.Npad - 16 + lrow);           // tiles past the padded width re-read the last one (never stored)
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ls = 0.f;
    if (a.pre == 2 && row < a.M) ls = logf(sigma_of(a.use_scalar ? a.t_scalar : a.X[row % a.x_mod], a.t_is_time, a.smin, a.ratio));
#pragma unroll UNR
    for (int ch = 0; ch < (a.K >> 4); ++ch) {
        f32x4 bf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const f32x4*>(a.W + ((size_t)ch * a.Npad + lcol[t]) * 16 + kq * 4);
        f32x4 af = {0.f, 0.f, 0.f, 0.f};
        if (row < a.M) {
            const int k0 = ch * 16 + kq * 4;
            if (a.pre == 2) {
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    // x[:, None] * W[None, :] * 2 * np.pi, evaluated left to right in fp32
                    const float arg = ((ls * a.fourW[k % a.nfour]) * 2.0f) * 3.14159265358979323846f;
                    af[j] = k < a.nfour ? sinf(arg) : cosf(arg);
                }
            } else {
                const int xrow = a.row_div > 0 ? row / a.row_div : row, srow = a.row_div > 0 ? row - xrow * a.row_div : row;
                af = *reinterpret_cast<const f32x4*>(a.X + (size_t)xrow * a.ldx + k0);
                if (a.pre == 3 && a.labels && srow < a.label_rows)     // same order of additions as the time_mlp.2 epilogue below
                    for (int j = 0; j < 4; ++j)
                        for (int c = 0; c < a.ncls; ++c) af[j] += a.labels[(size_t)srow * a.ncls + c] * a.Wl[(size_t)(k0 + j) * a.ncls + c];
                if (a.pre == 1 || a.pre == 3)
                    for (int j = 0; j < 4; ++j) af[j] = silu_f(af[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(af[j], bf[t][j], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (col[t] >= a.N) continue;
        float add = a.bias ? a.bias[col[t]] : 0.f;
        if (a.bl && a.pre != 3) add += a.bl[col[t]];
        for (int r = 0; r < 4; ++r) {
            const int orow = blockIdx.x * 16 + kq * 4 + r;
            if (orow >= a.M) continue;
            float v = acc[t][r] + add;
            if (a.labels && a.pre != 3 && orow < a.label_rows)
                for (int c = 0; c < a.ncls; ++c) v += a.labels[(size_t)orow * a.ncls + c] * a.Wl[(size_t)col[t] * a.ncls + c];
            a.Y[(size_t)orow * a.ldy + col[t]] = v;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator -> N(0,1) by Box-Muller.  counter = (element/4, draw index, stream),
// key = seed: any (draw, element) can be generated independently -> no RNG state, graph-replay safe.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t quad, uint32_t draw) {
    uint32_t c[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), draw, 0x52444d49u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float s = 2.3283064365386963e-10f;   // 2^-32
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
    const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    const float t0 = 6.283185307179586f * u1, t1 = 6.283185307179586f * u3;
    return f32x4{r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1)};
}

// z[i] for element i of draw `draw`; elements are numbered seq_offset*E + b*E + e so shards are disjoint
__global__ __launch_bounds__(RDMI_THREADS) void philox_normal_kernel(float* __restrict__ z, long n, uint64_t seed,
                                                                      uint64_t elem_offset, const int* draw_ctr,
                                                                      int draw_add) {
    const uint32_t draw = (uint32_t)((draw_ctr ? *draw_ctr : 0) + draw_add);
    const long q = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (q * 4 >= n) return;
    // quads are aligned to the GLOBAL element index so a shard draws exactly what the full batch would
    const uint64_t g0 = elem_offset + (uint64_t)q * 4;
    const f32x4 a = philox_normal4(seed, g0 >> 2, draw);
    if ((g0 & 3) == 0) {
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) z[q * 4 + j] = a[j];
    } else {
        const f32x4 b = philox_normal4(seed, (g0 >> 2) + 1, draw);
        const int sh = (int)(g0 & 3);
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) z[q * 4 + j] = (sh + j < 4) ? a[sh + j] : b[sh + j - 4];
    }
}

// --------------------------------------------------------------------------------------------
// Sampler elementwise kernels
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RDMI_THREADS) void reflect_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < n) out[i] = reflect_f(in[i]);
}

// score = (1 + w) * s[0:B] - w * s[B:2B]      (RD/models/utils.py:124-138)
__global__ __launch_bounds__(RDMI_THREADS) void cfg_combine_kernel(const float* __restrict__ s2, const float* __restrict__ w,
                                                                    float* __restrict__ out, int B, int E) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    const float wt = w ? w[i / E] : 0.f;
    out[i] = (1.0f + wt) * s2[i] - wt * s2[(long)B * E + i];
}

struct StepState {            // device-resident loop state so one captured step graph can be replayed
    int step;                 // index i of the current update (timesteps[i])
    int draw;                 // number of noise tensors consumed so far
};

// Reflected Euler-Maruyama update (RD/sampling.py:198-207, RD/sde_lib.py:93-101,135-140).
// t_vec: per-sample times, or (t_vec == null) the scalar ts[state->step] for every sample.
// z: noise tensor for this update: zbase + draw * B * E when zstride_by_draw, else zbase.
__global__ __launch_bounds__(RDMI_THREADS) void em_update_kernel(
    const float* __restrict__ x, const float* __restrict__ score, const float* __restrict__ zbase,
    const float* __restrict__ t_vec, const float* __restrict__ ts, const StepState* __restrict__ st,
    float* __restrict__ x_out, float* __restrict__ x_mean_out, float* __restrict__ trace,
    int B, int E, int N, float smin, float ratio, float gconst, int z_by_draw) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    const float t = t_vec ? t_vec[i / E] : ts[st->step];
    const float sigma = smin * powf(ratio, t);
    const float g = sigma * gconst;                       // diffusion, RD/sde_lib.py:138-139
    const float dt = -1.0f / (float)N;
    const float* z = z_by_draw ? zbase + (long)st->draw * B * E : zbase;
    const float drift = 0.0f - (g * g) * score[i];        // RSDE.sde, RD/sde_lib.py:97-98
    const float xm = x[i] + drift * dt;
    const float xn = xm + (g * sqrtf(-dt)) * z[i];
    const float xr = reflect_f(xn);
    x_out[i] = xr;
    if (x_mean_out) x_mean_out[i] = reflect_f(xm);
    if (trace) trace[(long)st->step * B * E + i] = xr;
}

// One launch per predictor update: CFG combine + N(0,1) draw (in-kernel Philox or injected) + reflected
// Euler-Maruyama + trace / teacher forcing.  s: [2B][E] (use_cfg) or [B][E] scores straight from the network.
__global__ __launch_bounds__(RDMI_THREADS) void em_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ w, const float* __restrict__ z,
    float* __restrict__ x_out, float* __restrict__ trace, const float* __restrict__ teacher,
    int B, int E, int N, float t, float smin, float ratio, float gconst, int use_cfg, uint64_t seed, uint64_t elem_offset, uint32_t draw) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    float score = s[i];
    if (use_cfg) { const float wt = w ? w[i / E] : 0.f; score = (1.0f + wt) * score - wt * s[(long)B * E + i]; }
    float zz;
    if (z) zz = z[i];
    else { const uint64_t g = elem_offset + (uint64_t)i; zz = philox_normal4(seed, g >> 2, draw)[(int)(g & 3)]; }
    const float sigma = smin * powf(ratio, t);
    const float g_ = sigma * gconst;
    const float dt = -1.0f / (float)N;
    const float drift = 0.0f - (g_ * g_) * score;
    const float xm = x[i] + drift * dt;
    const float xr = reflect_f(xm + (g_ * sqrtf(-dt)) * zz);
    if (trace) trace[i] = xr;
    x_out[i] = teacher ? teacher[i] : xr;
}

// Langevin, launch 1 of 2: per-sample CFG-combined score (stored), noise (stored) and their L2 norms.
__global__ __launch_bounds__(64) void langevin_prep_kernel(
    const float* __restrict__ s, const float* __restrict__ w, const float* __restrict__ z_in, float* __restrict__ score_out,
    float* __restrict__ z_out, float* __restrict__ norms, int B, int E, int use_cfg, uint64_t seed, uint64_t elem_offset, uint32_t draw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float wt = (use_cfg && w) ? w[b] : 0.f;
    float s2 = 0.f, z2 = 0.f;
    for (int e = lane; e < E; e += 64) {
        const long i = (long)b * E + e;
        float sc = s[i];
        if (use_cfg) sc = (1.0f + wt) * sc - wt * s[(long)B * E + i];
        float zz;
        if (z_in) zz = z_in[i];
        else { const uint64_t g = elem_offset + (uint64_t)i; zz = philox_normal4(seed, g >> 2, draw)[(int)(g & 3)]; }
        score_out[i] = sc; z_out[i] = zz;
        s2 += sc * sc; z2 += zz * zz;
    }
    for (int m = 32; m >= 1; m >>= 1) { s2 += __shfl_xor(s2, m); z2 += __shfl_xor(z2, m); }
    if (lane == 0) { norms[b] = sqrtf(s2); norms[B + b] = sqrtf(z2); }
}

// Langevin corrector, part 1: per-sample L2 norms of score and noise (RD/sampling.py:225-226).
__global__ __launch_bounds__(64) void row_norms_kernel(const float* __restrict__ score, const float* __restrict__ zbase,
                                                      const StepState* __restrict__ st, float* __restrict__ norms,
                                                      int B, int E, int z_by_draw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* z = z_by_draw ? zbase + (long)st->draw * B * E : zbase;
    float s2 = 0.f, z2 = 0.f;
    for (int e = lane; e < E; e += 64) {
        const float sv = score[(long)b * E + e], zv = z[(long)b * E + e];
        s2 += sv * sv; z2 += zv * zv;
    }
    for (int m = 32; m >= 1; m >>= 1) { s2 += __shfl_xor(s2, m); z2 += __shfl_xor(z2, m); }
    if (lane == 0) { norms[b] = sqrtf(s2); norms[B + b] = sqrtf(z2); }
}

// part 2: step = 2*(snr*mean||z||/mean||s||)^2 (one scalar for the batch); x_mean = x + step*s;
// x' = x_mean + sqrt(2*step)*z; reflect both (RD/sampling.py:227-231).  Every workgroup re-reduces the
// 2B norms in a fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(RDMI_THREADS) void langevin_update_kernel(
    const float* __restrict__ x, const float* __restrict__ score, const float* __restrict__ zbase,
    const float* __restrict__ norms, const StepState* __restrict__ st, float* __restrict__ x_out,
    float* __restrict__ x_mean_out, int B, int E, float snr, int z_by_draw) {
    float* red = reinterpret_cast<float*>(rdmi_lds);     // [2][4]
    const int tid = threadIdx.x;
    float gs = 0.f, zs = 0.f;
    for (int b = tid; b < B; b += RDMI_THREADS) { gs += norms[b]; zs += norms[B + b]; }
    for (int m = 32; m >= 1; m >>= 1) { gs += __shfl_xor(gs, m); zs += __shfl_xor(zs, m); }
    if ((tid & 63) == 0) { red[tid >> 6] = gs; red[4 + (tid >> 6)] = zs; }
    __syncthreads();
    const float gmean = (red[0] + red[1] + red[2] + red[3]) / (float)B;
    const float zmean = (red[4] + red[5] + red[6] + red[7]) / (float)B;
    const float r = snr * zmean / gmean;
    const float step = r * r * 2.0f;
    const float nscale = sqrtf(step * 2.0f);
    const float* z = z_by_draw ? zbase + (long)st->draw * B * E : zbase;
    const long i = (long)blockIdx.x * RDMI_THREADS + tid;
    if (i < (long)B * E) {
        const float xm = x[i] + step * score[i];
        x_out[i] = reflect_f(xm + nscale * z[i]);
        if (x_mean_out) x_mean_out[i] = reflect_f(xm);
    }
}

// parity testing: restart the next update from the recorded state of the reference trajectory
__global__ __launch_bounds__(RDMI_THREADS) void teacher_copy_kernel(float* __restrict__ x, const float* __restrict__ teacher,
                                                                     const StepState* __restrict__ st, long BE) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < BE) x[i] = teacher[(long)st->step * BE + i];
}

// advance the device-resident loop state (single work-item; stream order makes it race-free)
__global__ void step_advance_kernel(StepState* st, int dstep, int ddraw) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->step += dstep; st->draw += ddraw; }
}

// timesteps = torch.linspace(start, end, N)[0:n] on the device: fp32 step, value i = start + step*i in the first half and
// end - step*(N-1-i) in the second (ATen's linspace kernel), evaluated in double WITHOUT contraction into an fma so that it
// reproduces the host-side copy of the same expression bit for bit (rdmi_pc_sample passes that copy to the update kernels).
__global__ __launch_bounds__(RDMI_THREADS) void linspace_kernel(float* __restrict__ ts, int n, int N, float start, float end) {
    const int i = blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= n) return;
    const double step = (double)(float)((end - start) / (float)(N - 1));
#if defined(__HIP_DEVICE_COMPILE__)
    ts[i] = (float)(i < N / 2 ? __dadd_rn((double)start, __dmul_rn(step, (double)i)) : __dsub_rn((double)end, __dmul_rn(step, (double)(N - 1 - i))));
#else
    ts[i] = (float)(i < N / 2 ? (double)start + step * i : (double)end - step * (N - 1 - i));
#endif
}

// fill a [M] vector with ts[step] (all samples share one time per update, RD/sampling.py:329)
__global__ void fill_time_kernel(float* dst, const float* ts, const StepState* st, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) dst[i] = ts[st->step];
}

// cube.score_hk (RD/cube.py:73-193): score of the reflected heat kernel with t = sigma^2/2 per sample.
//   t > cutoff : eigenfunction series  -2pi sum_k k e^{-t k^2 pi^2} sin(k pi x) cos(k pi x0)
//                                      / (1 + 2 sum_k e^{-t k^2 pi^2} cos(k pi x) cos(k pi x0) + 1e-12)
//   otherwise  : images x_r in {2j + x, 2j - x}, j = -refls..refls:
//                sum sign (-2 (x_r - x0) / 4t) e^{-(x_r - x0)^2 / 4t} / (sum e^{...} + 1e-12)
__global__ __launch_bounds__(RDMI_THREADS) void score_hk_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                                 const float* __restrict__ sigma, float* __restrict__ out,
                                                                 int B, int E, int efs, int refls, float cutoff) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    const float sg = sigma[i / E];
    const float t = sg * sg / 2.0f;
    const float xv = x[i], ov = x0[i];
    const float pi = 3.14159265358979323846f;
    float num = 0.f, den = 0.f;
    if (t > cutoff) {
        const float pi2 = (float)(3.14159265358979323846 * 3.14159265358979323846);
        for (int k = 1; k <= efs; ++k) {
            const float kf = (float)k;
            const float xr = (pi * xv) * kf, orr = (pi * ov) * kf;
            const float ed = expf((-t * (kf * kf)) * pi2);
            const float co = cosf(orr);
            num += (ed * kf) * (sinf(xr) * co);
            den += ed * (cosf(xr) * co);
        }
        out[i] = ((float)(-2.0 * 3.14159265358979323846) * num) / ((1.0f + 2.0f * den) + 1e-12f);
    } else {
        const float fourt = 4.0f * t;
        for (int half = 0; half < 2; ++half)
            for (int j = -refls; j <= refls; ++j) {
                const float r = (float)(2 * j);
                const float xm = (half == 0 ? r + xv : r - xv) - ov;
                const float e = expf(-(xm * xm) / fourt);
                const float cf = -2.0f * xm / fourt;
                num += (cf * e) * (half == 0 ? 1.0f : -1.0f);
                den += e;
            }
        out[i] = num / (den + 1e-12f);
    }
}

// Score-matching loss pieces (RD/losses.py:79-93).
// perturbed = reflect(batch + sigma(t) * z)                                                     (:81-82)
__global__ __launch_bounds__(RDMI_THREADS) void perturb_kernel(const float* __restrict__ batch, const float* __restrict__ z,
                                                                const float* __restrict__ t, float* __restrict__ out, int B, int E,
                                                                float smin, float ratio) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    const float sigma = smin * powf(ratio, t[i / E]);
    out[i] = reflect_f(batch[i] + sigma * z[i]);
}

// per_sample[b] = reduce_e( weight_b * (score - score_hk(perturbed, batch, sigma_b))^2 ), one wave per sample:
//   weight = sigma^2 (or g(t)^2 with likelihood weighting), reduce = 0.5 * sum (or mean)        (:84-92)
__global__ __launch_bounds__(64) void sm_loss_kernel(const float* __restrict__ score, const float* __restrict__ perturbed,
                                                     const float* __restrict__ batch, const float* __restrict__ t,
                                                     float* __restrict__ per_sample, float* __restrict__ dscore, int B, int E, float smin, float ratio,
                                                     float gconst, int likelihood_weighting, int reduce_mean, int efs, int refls,
                                                     float cutoff) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float sg = smin * powf(ratio, t[b]);
    const float th = sg * sg / 2.0f;
    const float wgt = likelihood_weighting ? (sg * gconst) * (sg * gconst) : sg * sg;
    const float pi = 3.14159265358979323846f;
    float acc = 0.f;
    for (int e = lane; e < E; e += 64) {
        const long i = (long)b * E + e;
        const float xv = perturbed[i], ov = batch[i];
        float num = 0.f, den = 0.f, hk;
        if (th > cutoff) {
            const float pi2 = (float)(3.14159265358979323846 * 3.14159265358979323846);
            for (int k = 1; k <= efs; ++k) {
                const float kf = (float)k;
                const float ed = expf((-th * (kf * kf)) * pi2);
                const float co = cosf((pi * ov) * kf);
                num += (ed * kf) * (sinf((pi * xv) * kf) * co);
                den += ed * (cosf((pi * xv) * kf) * co);
            }
            hk = ((float)(-2.0 * 3.14159265358979323846) * num) / ((1.0f + 2.0f * den) + 1e-12f);
        } else {
            const float fourt = 4.0f * th;
            for (int half = 0; half < 2; ++half)
                for (int j = -refls; j <= refls; ++j) {
                    const float r = (float)(2 * j);
                    const float xm = (half == 0 ? r + xv : r - xv) - ov;
                    const float ee = expf(-(xm * xm) / fourt);
                    num += ((-2.0f * xm / fourt) * ee) * (half == 0 ? 1.0f : -1.0f);
                    den += ee;
                }
            hk = num / (den + 1e-12f);
        }
        const float d = score[i] - hk;
        acc += wgt * (d * d);
        if (dscore) dscore[i] = reduce_mean ? 2.0f * wgt * d / (float)E : wgt * d;     // d per_sample[b] / d score[i]
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) per_sample[b] = reduce_mean ? acc / (float)E : 0.5f * acc;
}

// NHWC -> NCHW copy (debug taps and the C=1 boundary is a no-op)
__global__ __launch_bounds__(RDMI_THREADS) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                     int NB, int HW, int C, int src_bf16) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)NB * HW * C) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const int p = (int)(r % HW);
    const long n = r / HW;
    dst[(n * C + c) * HW + p] = ldact1(src, (size_t)i, src_bf16);
}

// Post-sampling un-normalisation of the GTO-Halo 67-vectors (Benchmark/gto_halo_benchmarking.py:255-333 and
// _convert_to_spherical :335-361): first 67 of the 81 image values; column 0 = halo energy, 1..66 = model outputs
// m*0.1811+0.4652, then per-variable ranges; the 20 control triplets (ux,uy,uz)*2-1 -> (alpha, theta, min(|u|,1)).
// One work-item per (sample, unit): units 0..19 = control triplets, 20 = head columns 0..3, 21 = tail columns 64..66.
__global__ __launch_bounds__(RDMI_THREADS) void gto_unnormalize_kernel(const float* __restrict__ samples, float* __restrict__ out,
                                                                        unsigned long long* __restrict__ clips, int N, int E) {
    const int i = blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= N * 22) return;
    const int n = i / 22, unit = i - n * 22;
    const float* s = samples + (size_t)n * E;
    float* o = out + (size_t)n * 67;
    const float mean = 0.4652f, sd = 0.1811f, two_pi = 6.283185307179586f;
    auto m = [&](int j) { return s[1 + j] * sd + mean; };        // model_outputs[:, j]
    if (unit == 20) {
        o[0] = s[0] * (float)(0.095 - 0.008) + 0.008f;                 // halo energy from the normalised class label
        o[1] = m(0) * 40.0f + 0.0f;
        o[2] = m(1) * 15.0f + 0.0f;
        o[3] = m(2) * 15.0f + 0.0f;
    } else if (unit == 21) {
        o[64] = m(63) * 62.0f + 408.0f;                           // final fuel mass
        o[65] = m(64);                                            // halo period stays normalised (:322)
        o[66] = m(65) * 6.0f + 5.0f;                              // manifold length
    } else {
        const int j = 3 + 3 * unit;
        const float ux = m(j) * 2.0f * 1.0f - 1.0f, uy = m(j + 1) * 2.0f * 1.0f - 1.0f, uz = m(j + 2) * 2.0f * 1.0f - 1.0f;
        float u = sqrtf(ux * ux + uy * uy + uz * uz);
        float theta = u != 0.f ? asinf(uz / u) : 0.f;
        float alpha = atan2f(uy, ux);
        alpha = alpha >= 0.f ? alpha : two_pi + alpha;
        theta = theta >= 0.f ? theta : two_pi + theta;
        if (u > 1.f) { u = 1.f; if (clips) atomicAdd(clips, 1ull); }
        o[1 + j] = alpha; o[2 + j] = theta; o[3 + j] = u;
    }
}

// Training-data side (RD/datasets.py:82-98, GTOHaloImageDataset.__getitem__): row idx[b] of the device-resident table
// [rows][L] -> image[b] = (pad(vec, E) - mean) / std  (the zero padding is normalised too) and label[b] = vec[0].
__global__ __launch_bounds__(RDMI_THREADS) void gto_pack_kernel(const float* __restrict__ data, const long long* __restrict__ idx,
                                                                 float* __restrict__ images, float* __restrict__ labels, int B, int L, int E,
                                                                 float mean, float sd) {
    const int i = blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= B * E) return;
    const int b = i / E, e = i - b * E;
    const float* row = data + (size_t)(idx ? idx[b] : b) * L;
    const float v = e < L ? row[e] : 0.f;
    images[i] = (v - mean) / sd;
    if (e == 0) labels[b] = v;
}

