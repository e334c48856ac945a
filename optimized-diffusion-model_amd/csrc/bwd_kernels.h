// Backward pass of NCSN++ (training step, RD/losses.py:141-149 -> loss.backward()) on the LAYER plan.
//
// Correctness-first structure (round 1): every forward layer op is differentiated by a short sequence of generic
// kernels over NHWC global tensors; all data-gradient contractions reuse the forward implicit-GEMM kernel
// (conv_mfma_kernel) with transposed weight packs and adjoint tap tables, so the only new MFMA kernel is the
// weight-gradient GEMM.  Parameter gradients are accumulated with fp32 atomics into one flat buffer laid out in
// the reference's parameter order (the Python side hands its slices to autograd).
//
//   forward op:  V = concat(gather(A), B);  Act = drop(silu(GN(V)))  [or V];  Y = s * (conv(Act) + b [+ dense] [+ NIN(Vs) | + R])
//   backward  :  G = s * gY            (bwd_scale_kernel; also gR += G, gdense = colsum per sample, db = colsum)
//                GA = dgrad(G)         (conv_mfma_kernel, adjoint table, W^T)        GS = G . Wn^T (1x1, same kernel)
//                GV, Act, dgamma, dbeta = gn_bwd(V, GA)                               (gn_bwd_kernel)
//                dW += Act^T (*) G     (wgrad_mfma_kernel)                            dWn += Vs^T G
//                gA, gB += scatter(GV [+ GS])                                         (scatter_grad_kernel)
#pragma once
#include "common.h"
#include "misc_kernels.h"
#include "conv_kernel.h"

// G = scale * gY; optional identity-residual accumulation gR += G; one work-item per element.
__global__ __launch_bounds__(RDMI_THREADS) void bwd_scale_kernel(const float* __restrict__ gY, float* __restrict__ G,
                                                                  float* __restrict__ gR, float scale, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= n) return;
    const float g = gY[i] * scale;
    G[i] = g;
    if (gR) gR[i] += g;
}

// per-sample column sums of G [NB][HW][C]: gdense[n][off + c] = sum_p G[n][p][c] (optional);
// bias gradients db[c] (+ db2[c]) += sum_{n,p} G (atomics).  grid = NB, block = 256 (c strided).
__global__ __launch_bounds__(RDMI_THREADS) void bwd_colsum_kernel(const float* __restrict__ G, float* __restrict__ gdense,
                                                                   int dense_stride, int dense_off, float* __restrict__ db,
                                                                   float* __restrict__ db2, int HW, int C) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += RDMI_THREADS) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += G[((size_t)n * HW + p) * C + c];
        if (gdense) gdense[(size_t)n * dense_stride + dense_off + c] = s;
        if (db) atomicAdd(db + c, s);
        if (db2) atomicAdd(db2 + c, s);
    }
}

// GroupNorm(+SiLU+dropout) backward for one sample per workgroup, everything in LDS.
//   V  : raw virtual input, gathered like the forward (concat of mapped A and B)
//   GA : gradient w.r.t. the activated tensor  [n][HWv][Cv]   (overwritten by GV = gradient w.r.t. V)
//   ACT: the activated tensor itself is written out for the weight-gradient GEMM
// has_gn == 0: GV = GA, ACT = V (plain convs: up/down-sampling, input conv).
struct GnBwdArgs {
    const float* srcA; const float* srcB; const int* mapA;
    int CA, CB, Cv, HWa, HWv, srcA_mod, NB;
    float* GA; float* ACT;
    const float* gamma; const float* beta; float* dgamma; float* dbeta;
    int G, has_gn; float eps;
    float drop_p; uint64_t seed; uint32_t op_id;
};

__global__ __launch_bounds__(RDMI_THREADS) void gn_bwd_kernel(GnBwdArgs a) {
    const int tid = threadIdx.x, n = blockIdx.x;
    const int rs = a.Cv + 4;
    float* V = reinterpret_cast<float*>(rdmi_lds);               // [HWv][rs]
    float* Gt = V + (size_t)(a.HWv + 1) * rs;                     // [HWv][rs]  gy / gxhat
    float* stat = Gt + (size_t)a.HWv * rs;                        // [G][4]: mean, rstd, m1, m2
    // gather V (reuses the forward staging code: S = 1)
    {
        const int c4n = a.Cv >> 2, total = a.HWv * c4n;
        for (int i = tid; i < total; i += RDMI_THREADS) {
            const int v = i / c4n, c = (i - v * c4n) << 2;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (c < a.CA) {
                const int nA = a.srcA_mod > 0 ? n % a.srcA_mod : n;
                const float* p = a.srcA + ((size_t)nA * a.HWa + (a.mapA ? a.mapA[v] : v)) * a.CA + c;
                if ((a.CA & 3) == 0) val = *reinterpret_cast<const f32x4*>(p);
                else for (int j = 0; j < 4; ++j) if (c + j < a.CA) val[j] = p[j];
            } else if (c < a.CA + a.CB) {
                val = *reinterpret_cast<const f32x4*>(a.srcB + ((size_t)n * a.HWv + v) * a.CB + (c - a.CA));
            }
            *reinterpret_cast<f32x4*>(V + (size_t)v * rs + c) = val;
            *reinterpret_cast<f32x4*>(Gt + (size_t)v * rs + c) = *reinterpret_cast<const f32x4*>(a.GA + ((size_t)n * a.HWv + v) * a.Cv + c);
        }
    }
    __syncthreads();
    float* act_out = a.ACT + (size_t)n * a.HWv * a.Cv;
    float* gv_out = a.GA + (size_t)n * a.HWv * a.Cv;
    if (!a.has_gn) {
        for (int i = tid; i < a.HWv * a.Cv; i += RDMI_THREADS) { const int v = i / a.Cv, c = i - v * a.Cv; act_out[i] = V[(size_t)v * rs + c]; }
        return;                                                   // GV == GA already in place
    }
    const int G = a.G, Cg = a.Cv / G, cnt = Cg * a.HWv;
    // group statistics (two-pass), one group per work-item subset: T lanes per group
    {
        const int T = RDMI_THREADS / G;                           // G in {16, 32} -> T in {16, 8}
        const int g = tid / T, sub = tid - g * T;
        float s = 0.f;
        for (int e = sub; e < cnt; e += T) { const int v = e / Cg, cc = e - v * Cg; s += V[(size_t)v * rs + g * Cg + cc]; }
        for (int m = T >> 1; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        const float mean = s / (float)cnt;
        float q = 0.f;
        for (int e = sub; e < cnt; e += T) { const int v = e / Cg, cc = e - v * Cg; const float d = V[(size_t)v * rs + g * Cg + cc] - mean; q += d * d; }
        for (int m = T >> 1; m >= 1; m >>= 1) q += __shfl_xor(q, m);
        if (sub == 0) { stat[4 * g] = mean; stat[4 * g + 1] = 1.0f / sqrtf(q / (float)cnt + a.eps); }
    }
    __syncthreads();
    // per element: xhat, y, activation (+dropout), gy; V <- xhat, Gt <- gxhat = gy * gamma; channel sums for dgamma/dbeta
    for (int c = tid; c < a.Cv; c += RDMI_THREADS) {
        const int g = c / Cg;
        const float mean = stat[4 * g], rstd = stat[4 * g + 1], gm = a.gamma[c], bt = a.beta[c];
        float dg = 0.f, dbt = 0.f;
        for (int v = 0; v < a.HWv; ++v) {
            const float xh = (V[(size_t)v * rs + c] - mean) * rstd;
            const float y = xh * gm + bt;
            const float sg = 1.0f / (1.0f + __expf(-y));
            const float ds = dropout_scale(a.seed, a.op_id, ((uint64_t)n * a.HWv + v) * a.Cv + c, a.drop_p);
            act_out[(size_t)v * a.Cv + c] = y * sg * ds;
            const float gy = Gt[(size_t)v * rs + c] * ds * (sg * (1.0f + y * (1.0f - sg)));
            dg += gy * xh; dbt += gy;
            V[(size_t)v * rs + c] = xh;
            Gt[(size_t)v * rs + c] = gy * gm;
        }
        atomicAdd(a.dgamma + c, dg);
        atomicAdd(a.dbeta + c, dbt);
    }
    __syncthreads();
    {
        const int T = RDMI_THREADS / G;
        const int g = tid / T, sub = tid - g * T;
        float m1 = 0.f, m2 = 0.f;
        for (int e = sub; e < cnt; e += T) {
            const int v = e / Cg, cc = e - v * Cg;
            const float gx = Gt[(size_t)v * rs + g * Cg + cc];
            m1 += gx; m2 += gx * V[(size_t)v * rs + g * Cg + cc];
        }
        for (int m = T >> 1; m >= 1; m >>= 1) { m1 += __shfl_xor(m1, m); m2 += __shfl_xor(m2, m); }
        if (sub == 0) { stat[4 * g + 2] = m1 / (float)cnt; stat[4 * g + 3] = m2 / (float)cnt; }
    }
    __syncthreads();
    for (int i = tid; i < a.HWv * a.Cv; i += RDMI_THREADS) {
        const int v = i / a.Cv, c = i - v * a.Cv, g = c / Cg;
        gv_out[i] = stat[4 * g + 1] * (Gt[(size_t)v * rs + c] - stat[4 * g + 2] - V[(size_t)v * rs + c] * stat[4 * g + 3]);
    }
}

// gA[n][s][c] += sum_{v in inv(s)} GV[n][v][c]  (c < CA)   and   gB[n][v][c-CA] += GV[n][v][c]  (c >= CA)
// inv_start/inv_list: inverse of the nearest map (null = identity).  One work-item per destination element.
__global__ __launch_bounds__(RDMI_THREADS) void scatter_grad_kernel(const float* __restrict__ GV, float* __restrict__ gA,
                                                                     float* __restrict__ gB, const int* __restrict__ inv_start,
                                                                     const int* __restrict__ inv_list, int NB, int HWa, int HWv,
                                                                     int CA, int CB, int Cv) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    const long nA = (long)NB * HWa * CA, nB = (long)NB * HWv * CB;
    if (i < nA) {
        if (!gA) return;
        const int c = (int)(i % CA);
        const long r = i / CA;
        const int s = (int)(r % HWa);
        const long n = r / HWa;
        float acc = 0.f;
        if (inv_start) { for (int k = inv_start[s]; k < inv_start[s + 1]; ++k) acc += GV[((size_t)n * HWv + inv_list[k]) * Cv + c]; }
        else acc = GV[((size_t)n * HWv + s) * Cv + c];
        gA[i] += acc;
    } else if (i < nA + nB) {
        if (!gB) return;
        const long j = i - nA;
        const int c = (int)(j % CB);
        const long r = j / CB;                                   // n * HWv + v
        gB[j] += GV[(size_t)r * Cv + CA + c];
    }
}

// Weight gradient: dW[co][ci][tap] (reference OIHW layout, or NIN [ci][co] / Linear [co][ci] through the strides)
//   += sum_{n, o} ACT[n][in(o, tap)][ci] * G[n][o][co]
// grid = (ntap * ksplit, ceil(Cin/32), ceil(Cout/32)): a workgroup owns a 32x32 (ci x co) tile of one tap for a slice of
// the samples (split-K over the batch, partial tiles merged with fp32 atomics); wave w owns the 16x16 sub-tile
// (w&1, w>>1) and contracts over (sample, output pixel) with MFMA 16x16x4, eight k-steps of loads in flight per iteration.
struct WgradArgs {
    const float* ACT; const float* G; float* dW;
    const int* tab;        // [HWo][ntap] input pixel of (output pixel, tap) or -1   (null: identity, 1 tap)
    int NB, HWv, HWo, Cin, Cout, ntap;
    int lda;               // channels per pixel of ACT (>= Cin: padded input channels)
    int ksplit;            // number of sample slices
    long s_co, s_ci, s_t;  // strides of dW
};

__global__ __launch_bounds__(RDMI_THREADS) void wgrad_mfma_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int tap = blockIdx.x % a.ntap, slice = blockIdx.x / a.ntap;
    int* vt = reinterpret_cast<int*>(rdmi_lds);                  // [HWo4] input pixel of each output pixel for this tap (-1: none)
    const int HWo4 = (a.HWo + 3) & ~3;
    for (int o = threadIdx.x; o < HWo4; o += RDMI_THREADS) vt[o] = o < a.HWo ? (a.tab ? a.tab[o * a.ntap + tap] : o) : -1;
    __syncthreads();
    const int ci = blockIdx.y * 32 + (wave & 1) * 16 + lrow;      // A row  (this lane's input channel)
    const int co = blockIdx.z * 32 + (wave >> 1) * 16 + lrow;     // B col  (this lane's output channel)
    const bool ci_ok = ci < a.Cin, co_ok = co < a.Cout;
    const int per = (a.NB + a.ksplit - 1) / a.ksplit;
    const int n_lo = slice * per, n_hi = min(a.NB, n_lo + per);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;                                          // k-steps (of 4 pixels) per unrolled iteration
    for (int n = n_lo; n < n_hi; ++n) {
        const float* An = a.ACT + (size_t)n * a.HWv * a.lda + ci;
        const float* Gn = a.G + (size_t)n * a.HWo * a.Cout + co;
        for (int o0 = 0; o0 < HWo4; o0 += 4 * U) {
            float av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = o0 + 4 * u + kq;
                const int v = o < HWo4 ? vt[o] : -1;
                av[u] = (v >= 0 && ci_ok) ? ldg1(An + (size_t)v * a.lda) : 0.f;
                bv[u] = (o < a.HWo && co_ok) ? ldg1(Gn + (size_t)o * a.Cout) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = mfma16(av[u], bv[u], acc);
        }
    }
    // D[row = ci_local][col = co_local]: lane holds col = lrow, rows kq*4 + r
    for (int r = 0; r < 4; ++r) {
        const int ci_out = blockIdx.y * 32 + (wave & 1) * 16 + kq * 4 + r;
        if (ci_out < a.Cin && co_ok) atomicAdd(a.dW + co * a.s_co + ci_out * a.s_ci + tap * a.s_t, acc[r]);
    }
}

// C[M][N] (+)= A[M][K] . B[K][N] with arbitrary strides, optional elementwise pre-activation of A / B (0 none, 1 SiLU);
// one work-item per output element (tiny embedding GEMMs: K <= 2048).
struct SgemmArgs {
    const float* A; long a_m, a_k; int a_act;
    const float* B; long b_k, b_n; int b_act;
    float* C; long c_m, c_n; int accumulate;
    int M, N, K;
};
__global__ __launch_bounds__(RDMI_THREADS) void small_gemm_kernel(SgemmArgs a) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)a.M * a.N) return;
    const int m = (int)(i / a.N), n = (int)(i - (long)m * a.N);
    float s = 0.f;
    for (int k = 0; k < a.K; ++k) {
        float x = a.A[m * a.a_m + k * a.a_k], y = a.B[k * a.b_k + n * a.b_n];
        if (a.a_act) x = silu_f(x);
        if (a.b_act) y = silu_f(y);
        s += x * y;
    }
    float* c = a.C + m * a.c_m + n * a.c_n;
    *c = a.accumulate ? *c + s : s;
}

// g <- g * silu'(x)
__global__ __launch_bounds__(RDMI_THREADS) void silu_bwd_kernel(float* __restrict__ g, const float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= n) return;
    const float y = x[i], sg = 1.0f / (1.0f + __expf(-y));
    g[i] *= sg * (1.0f + y * (1.0f - sg));
}

// out[c] (+)= sum_m X[m][c]    (bias gradients of the embedding layers)
__global__ __launch_bounds__(RDMI_THREADS) void colsum2d_kernel(const float* __restrict__ X, float* __restrict__ out, int M, int C,
                                                                 int ldx) {
    const int c = blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += X[(size_t)m * ldx + c];
    out[c] += s;
}

// Fourier features of log(sigma) materialised for the time_mlp.0 weight gradient: F[m][0:nf] = sin, [nf:2nf] = cos
__global__ __launch_bounds__(RDMI_THREADS) void fourier_kernel(const float* __restrict__ sigma, const float* __restrict__ W, float* __restrict__ F,
                                                                int M, int nf) {
    const int i = blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= M * 2 * nf) return;
    const int m = i / (2 * nf), k = i - m * 2 * nf;
    const float arg = ((logf(sigma[m]) * W[k % nf]) * 2.0f) * 3.14159265358979323846f;
    F[i] = k < nf ? sinf(arg) : cosf(arg);
}

// AttnBlockpp backward, one sample per workgroup, plain fp32 loops over LDS-resident tensors (C = 64, L <= 96).
// Recomputes xn, q, k, v, P, O from x, then:  gH = s*gOut;  dW3 += O^T gH; gO = gH W3^T;  gP = gO V^T; gV = P^T gO;
// gS = P * (gP - rowsum(gP * P)) / sqrt(C);  gQ = gS K; gK = gS^T Q;  dWq/k/v += xn^T g{Q,K,V};  gxn = sum g. W^T;
// GroupNorm backward (no activation) -> gx;  gX += gx + gH (residual branch).
struct AttnBwdArgs {
    const float* x; const float* gOut; float* gX;
    const float* gamma; const float* beta; float* dgamma; float* dbeta;
    const float* W[4]; const float* b[4];      // NIN_0..3: W [in][out]
    float* dW[4]; float* db[4];
    int NB, L, G; float eps, scale, out_scale;
};

template <int C>
__host__ __device__ inline size_t attn_bwd_lds_bytes(int L, int G) { return ((size_t)5 * L * C + (size_t)2 * L * L + 4 * G) * 4; }

template <int C>
__global__ __launch_bounds__(RDMI_THREADS) void attn_bwd_kernel(AttnBwdArgs a) {
    const int tid = threadIdx.x, n = blockIdx.x, L = a.L;
    constexpr int NPT = (96 * C + RDMI_THREADS - 1) / RDMI_THREADS;     // elements of an [L][C] tensor per work-item
    float* XH = reinterpret_cast<float*>(rdmi_lds);  // [L][C] raw x, then xhat
    float* Q = XH + L * C; float* K = Q + L * C; float* V = K + L * C;
    float* O = V + L * C;                            // attention output -> gO -> gV
    float* P = O + L * C;                            // [L][L]
    float* GP = P + L * L;                           // [L][L] gP -> gS
    float* stat = GP + L * L;                        // [G][4]
    const float* xg = a.x + (size_t)n * L * C;
    const float* gog = a.gOut + (size_t)n * L * C;   // gH(p, c) = gog[p*C + c] * out_scale (re-read from L2 when needed)
    for (int i = tid; i < L * C; i += RDMI_THREADS) XH[i] = xg[i];
    __syncthreads();
    const int G = a.G, Cg = C / G, cnt = Cg * L;
    if (tid < G) {
        float s = 0.f;
        for (int e = 0; e < cnt; ++e) s += XH[(e / Cg) * C + tid * Cg + e % Cg];
        const float mean = s / (float)cnt;
        float q = 0.f;
        for (int e = 0; e < cnt; ++e) { const float d = XH[(e / Cg) * C + tid * Cg + e % Cg] - mean; q += d * d; }
        stat[4 * tid] = mean; stat[4 * tid + 1] = 1.0f / sqrtf(q / (float)cnt + a.eps);
    }
    __syncthreads();
    for (int i = tid; i < L * C; i += RDMI_THREADS) { const int g = (i % C) / Cg; XH[i] = (XH[i] - stat[4 * g]) * stat[4 * g + 1]; }
    __syncthreads();
#define XN_(p, j) (XH[(p) * C + (j)] * a.gamma[j] + a.beta[j])
    // q, k, v
    for (int i = tid; i < L * C; i += RDMI_THREADS) {
        const int p = i / C, c = i - p * C;
        float q = a.b[0][c], k = a.b[1][c], v = a.b[2][c];
        for (int j = 0; j < C; ++j) { const float xv = XN_(p, j); q += xv * a.W[0][j * C + c]; k += xv * a.W[1][j * C + c]; v += xv * a.W[2][j * C + c]; }
        Q[i] = q; K[i] = k; V[i] = v;
    }
    __syncthreads();
    // P = softmax(Q K^T * scale): one row per work-item
    for (int r = tid; r < L; r += RDMI_THREADS) {
        float mx = -3.0e38f;
        for (int j = 0; j < L; ++j) { float s = 0.f; for (int c = 0; c < C; ++c) s += Q[r * C + c] * K[j * C + c]; s *= a.scale; P[r * L + j] = s; mx = fmaxf(mx, s); }
        float sum = 0.f;
        for (int j = 0; j < L; ++j) { const float e = __expf(P[r * L + j] - mx); P[r * L + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < L; ++j) P[r * L + j] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < L * C; i += RDMI_THREADS) {
        const int p = i / C, c = i - p * C;
        float o = 0.f;
        for (int j = 0; j < L; ++j) o += P[p * L + j] * V[j * C + c];
        O[i] = o;
    }
    __syncthreads();
    // NIN_3: dW3[j][c] += sum_p O[p][j] gH[p][c]; db3[c] += sum_p gH[p][c]
    for (int i = tid; i < C * C; i += RDMI_THREADS) {
        const int j = i / C, c = i - j * C;
        float s = 0.f;
        for (int p = 0; p < L; ++p) s += O[p * C + j] * (gog[p * C + c] * a.out_scale);
        atomicAdd(a.dW[3] + i, s);
    }
    if (tid < C) { float s = 0.f; for (int p = 0; p < L; ++p) s += gog[p * C + tid] * a.out_scale; atomicAdd(a.db[3] + tid, s); }
    __syncthreads();
    // O <- gO[p][j] = sum_c gH[p][c] W3[j][c]
    for (int i = tid; i < L * C; i += RDMI_THREADS) {
        const int p = i / C, j = i - p * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += (gog[p * C + c] * a.out_scale) * a.W[3][j * C + c];
        O[i] = s;
    }
    __syncthreads();
    // gP[r][j] = sum_c gO[r][c] V[j][c]
    for (int i = tid; i < L * L; i += RDMI_THREADS) {
        const int r = i / L, j = i - r * L;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += O[r * C + c] * V[j * C + c];
        GP[i] = s;
    }
    // gV[j][c] = sum_r P[r][j] gO[r][c]   (held in registers until gO is dead)
    float gv[NPT];
    {
        int m = 0;
        for (int i = tid; i < L * C; i += RDMI_THREADS) {
            const int j = i / C, c = i - j * C;
            float s = 0.f;
            for (int r = 0; r < L; ++r) s += P[r * L + j] * O[r * C + c];
            gv[m++] = s;
        }
    }
    __syncthreads();
    // gS = P * (gP - rowsum(gP * P)) * scale, in place over GP;  O <- gV
    for (int r = tid; r < L; r += RDMI_THREADS) {
        float d = 0.f;
        for (int j = 0; j < L; ++j) d += GP[r * L + j] * P[r * L + j];
        for (int j = 0; j < L; ++j) GP[r * L + j] = P[r * L + j] * (GP[r * L + j] - d) * a.scale;
    }
    {
        int m = 0;
        for (int i = tid; i < L * C; i += RDMI_THREADS) O[i] = gv[m++];
    }
    __syncthreads();
    // gQ[r][c] = sum_j gS[r][j] K[j][c];  gK[r][c] = sum_j gS[j][r] Q[j][c]   (registers, then overwrite Q, K)
    {
        float gq[NPT], gk[NPT];
        int m = 0;
        for (int i = tid; i < L * C; i += RDMI_THREADS) {
            const int r = i / C, c = i - r * C;
            float s1 = 0.f, s2 = 0.f;
            for (int j = 0; j < L; ++j) { s1 += GP[r * L + j] * K[j * C + c]; s2 += GP[j * L + r] * Q[j * C + c]; }
            gq[m] = s1; gk[m] = s2; ++m;
        }
        __syncthreads();
        m = 0;
        for (int i = tid; i < L * C; i += RDMI_THREADS) { Q[i] = gq[m]; K[i] = gk[m]; ++m; }
    }
    __syncthreads();
    // now Q = gQ, K = gK, O = gV.  Weight/bias gradients of NIN_0..2
    for (int i = tid; i < C * C; i += RDMI_THREADS) {
        const int j = i / C, c = i - j * C;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int p = 0; p < L; ++p) { const float xv = XN_(p, j); s0 += xv * Q[p * C + c]; s1 += xv * K[p * C + c]; s2 += xv * O[p * C + c]; }
        atomicAdd(a.dW[0] + i, s0); atomicAdd(a.dW[1] + i, s1); atomicAdd(a.dW[2] + i, s2);
    }
    if (tid < C) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int p = 0; p < L; ++p) { s0 += Q[p * C + tid]; s1 += K[p * C + tid]; s2 += O[p * C + tid]; }
        atomicAdd(a.db[0] + tid, s0); atomicAdd(a.db[1] + tid, s1); atomicAdd(a.db[2] + tid, s2);
    }
    // V <- gy[p][j] = gxn = sum_c gQ W0[j][c] + gK W1[j][c] + gV W2[j][c]   (GroupNorm here has no activation)
    for (int i = tid; i < L * C; i += RDMI_THREADS) {
        const int p = i / C, j = i - p * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += Q[p * C + c] * a.W[0][j * C + c] + K[p * C + c] * a.W[1][j * C + c] + O[p * C + c] * a.W[2][j * C + c];
        V[i] = s;
    }
    __syncthreads();
    if (tid < C) {
        float dg = 0.f, dbt = 0.f;
        for (int p = 0; p < L; ++p) { dg += V[p * C + tid] * XH[p * C + tid]; dbt += V[p * C + tid]; }
        atomicAdd(a.dgamma + tid, dg); atomicAdd(a.dbeta + tid, dbt);
    }
    if (tid < G) {
        float m1 = 0.f, m2 = 0.f;
        for (int e = 0; e < cnt; ++e) { const int p = e / Cg, c = tid * Cg + e % Cg; const float gxh = V[p * C + c] * a.gamma[c]; m1 += gxh; m2 += gxh * XH[p * C + c]; }
        stat[4 * tid + 2] = m1 / (float)cnt; stat[4 * tid + 3] = m2 / (float)cnt;
    }
    __syncthreads();
    float* gxo = a.gX + (size_t)n * L * C;
    for (int i = tid; i < L * C; i += RDMI_THREADS) {
        const int c = i % C, g = c / Cg;
        const float gxh = V[i] * a.gamma[c];
        gxo[i] += stat[4 * g + 1] * (gxh - stat[4 * g + 2] - XH[i] * stat[4 * g + 3]) + gog[i] * a.out_scale;
    }
#undef XN_
}

// loss backward: gscore[i] = gper[b] * dper_dscore[i]
__global__ __launch_bounds__(RDMI_THREADS) void rowscale_kernel(const float* __restrict__ gper, const float* __restrict__ d, float* __restrict__ out,
                                                                 int B, int E) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < (long)B * E) out[i] = gper[i / E] * d[i];
}
