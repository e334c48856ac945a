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
//                gA, gB += scatter(GV [+ GS])                                         (tail of gn_bwd_kernel)
#pragma once
#include "common.h"
#include "misc_kernels.h"
#include "conv_kernel.h"

// First backward kernel of a conv op, one workgroup per sample: G = scale * gY (the gradient the data- and weight-gradient
// kernels consume), the identity-residual branch gR += G, and the per-sample column sums of G [HW][C]:
// gdense[n][off + c] = sum_p G[n][p][c] (Dense_0 path, optional) and the bias gradients db[c] (+ db2[c]) += sum (atomics).
// 256 work-items = (256 / cw) row lanes x cw columns per pass, partial sums merged in LDS.
struct ColsumArgs {
    const float* gY; float* G; float* gR; float scale; float* gdense; int dense_stride, dense_off; float* db; float* db2; int HW, C, g_bf16;
};
template <int NT>
__device__ __forceinline__ void bwd_scale_colsum_body(const ColsumArgs& a, int n, int tid, float* red) {      // red: NT floats of LDS
    for (int c0 = 0; c0 < a.C; c0 += NT) {
        const int cw = min(NT, a.C - c0), R = NT / cw;
        const int r = tid / cw, c = c0 + tid - r * cw;
        float s = 0.f;
        if (r < R)
            for (int p0 = r; p0 < a.HW; p0 += 4 * R) {            // four pixels per pass: their loads are issued before any store
                float gy[4], gr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * R;
                    const size_t i = ((size_t)n * a.HW + min(p, a.HW - 1)) * a.C + c;
                    gy[u] = a.gY[i]; gr[u] = a.gR ? a.gR[i] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * R;
                    if (p < a.HW) {
                        const size_t i = ((size_t)n * a.HW + p) * a.C + c;
                        const float g = gy[u] * a.scale;
                        stact1(a.G, i, g, a.g_bf16);
                        if (a.gR) a.gR[i] = gr[u] + g;
                        s += g;
                    }
                }
            }
        red[tid] = r < R ? s : 0.f;
        __syncthreads();
        if (tid < cw) {
            float tot = 0.f;
            for (int j = 0; j < R; ++j) tot += red[tid + j * cw];
            if (a.gdense) a.gdense[(size_t)n * a.dense_stride + a.dense_off + c] = tot;
            if (a.db) atomicAdd(a.db + c, tot);
            if (a.db2) atomicAdd(a.db2 + c, tot);
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(RDMI_THREADS) void bwd_scale_colsum_kernel(ColsumArgs a) {
    __shared__ float red[RDMI_THREADS];
    bwd_scale_colsum_body<RDMI_THREADS>(a, blockIdx.x, threadIdx.x, red);
}

// GroupNorm(+SiLU+dropout) backward for one sample per workgroup (GN_THREADS work-items), everything in LDS, followed by the
// scatter of the input gradient to the gradient accumulators of the op's sources.
//   V  : raw virtual input, gathered like the forward (concat of mapped A and B)
//   GA : gradient w.r.t. the activated tensor  [n][HWv][Cv]   (read only; GV = gradient w.r.t. V stays in LDS)
//   ACT: the activated tensor itself is written out for the weight-gradient GEMM
//   gA[n][s][c] += sum_{v in inv(s)} GV[v][c] (c < CA),  gB[n][v][c-CA] += GV[v][c] (c >= CA): the workgroup owns its sample's
//   rows of gA / gB, so these are plain read-modify-writes (inv_start / inv_list: inverse of the nearest map, null = identity).
// has_gn == 0: GV = GA, ACT = V (plain convs: up/down-sampling, input conv, NIN shortcut).
#define GN_THREADS 1024
struct GnBwdArgs {
    const float* srcA; const float* srcB; const int* mapA;
    int CA, CB, Cv, HWa, HWv, srcA_mod, NB;
    const float* GA; float* ACT;
    const float* gamma; const float* beta; float* dgamma; float* dbeta;
    int G, has_gn; float eps;
    float drop_p; uint64_t seed; uint32_t op_id;
    const unsigned long long* seed_dev;   // non-null: dropout seed in device memory (see ConvArgs::seed_dev)
    int a_bf16, b_bf16, s_bf16;   // element type of srcA, of srcB, and of the scratch tensors GA / ACT (0 fp32, 1 bf16)
    float* gA; float* gB; const int* inv_start; const int* inv_list;   // fp32 gradient accumulators (null: that source takes no gradient)
    // tail.G != null: this workgroup then runs the first backward kernel of the NEXT conv op (bwd_scale_colsum) for its sample -- the
    // gradient that kernel reads is complete for sample n once this workgroup's scatter is (every earlier contribution is an earlier
    // launch on the same stream).  One launch, and one ~6 us launch gap of the serial backward chain, less per conv op.
    ColsumArgs tail;
};
__host__ __device__ inline size_t gn_bwd_lds_bytes(int HWv, int Cv) { return ((size_t)2 * (HWv + 1) * (Cv + 4) + 4 * 32 + 2 * GN_THREADS) * 4; }

__global__ __launch_bounds__(GN_THREADS) void gn_bwd_kernel(GnBwdArgs a) {
    const int tid = threadIdx.x, n = blockIdx.x;
    const int rs = a.Cv + 4;
    float* V = reinterpret_cast<float*>(rdmi_lds);               // [HWv + 1][rs]
    float* Gt = V + (size_t)(a.HWv + 1) * rs;                     // [HWv][rs]  gy / gxhat / GV
    float* stat = Gt + (size_t)a.HWv * rs;                        // [G][4]: mean, rstd, m1, m2
    float* red = stat + 4 * 32;                                   // [2][GN_THREADS] partial dgamma / dbeta
    // gather V (same addressing as the forward staging with S = 1) and the incoming gradient
    {
        const int c4n = a.Cv >> 2, total = a.HWv * c4n;
        for (int i = tid; i < total; i += GN_THREADS) {
            const int v = i / c4n, c = (i - v * c4n) << 2;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (c < a.CA) {
                const int nA = a.srcA_mod > 0 ? n % a.srcA_mod : n;
                const size_t ia = ((size_t)nA * a.HWa + (a.mapA ? a.mapA[v] : v)) * a.CA + c;
                if ((a.CA & 3) == 0) val = ldact4(a.srcA, ia, a.a_bf16);
                else for (int j = 0; j < 4; ++j) if (c + j < a.CA) val[j] = ldact1(a.srcA, ia + j, a.a_bf16);
            } else if (c < a.CA + a.CB) {
                val = ldact4(a.srcB, ((size_t)n * a.HWv + v) * a.CB + (c - a.CA), a.b_bf16);
            }
            *reinterpret_cast<f32x4*>(V + (size_t)v * rs + c) = val;
            *reinterpret_cast<f32x4*>(Gt + (size_t)v * rs + c) = ldact4(a.GA, ((size_t)n * a.HWv + v) * a.Cv + c, a.s_bf16);
        }
    }
    __syncthreads();
    const size_t base = (size_t)n * a.HWv * a.Cv;               // element offset of this sample in ACT / GA
    if (!a.has_gn) {
        for (int i = tid; i < a.HWv * a.Cv; i += GN_THREADS) { const int v = i / a.Cv, c = i - v * a.Cv; stact1(a.ACT, base + i, V[(size_t)v * rs + c], a.s_bf16); }
    } else {
        const int G = a.G, Cg = a.Cv / G, cnt = Cg * a.HWv;
        const int T = min(64, GN_THREADS / G);                    // G in {16, 32} -> T in {64, 32} lanes per group (inside one wave)
        const int g = tid / T, sub = tid - g * T;                 // waves with g >= G sit the group phases out
        // group statistics (two-pass)
        if (g < G) {
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
        // per element: xhat, y, activation (+dropout), gy; V <- xhat, Gt <- gxhat = gy * gamma; channel sums for dgamma / dbeta.
        // Work-item (r, c) walks the rows r, r + R, ... of channel c.
        {
            const int R = GN_THREADS / a.Cv, r = tid / a.Cv, c = tid - r * a.Cv;
            float dg = 0.f, dbt = 0.f;
            if (r < R) {
                const int gg = c / Cg;
                const float mean = stat[4 * gg], rstd = stat[4 * gg + 1], gm = a.gamma[c], bt = a.beta[c];
                for (int v = r; v < a.HWv; v += R) {
                    const float xh = (V[(size_t)v * rs + c] - mean) * rstd;
                    const float y = xh * gm + bt;
                    const float sg = 1.0f / (1.0f + __expf(-y));
                    const float ds = dropout_scale(a.seed_dev ? (uint64_t)*a.seed_dev : a.seed, a.op_id, ((uint64_t)n * a.HWv + v) * a.Cv + c, a.drop_p);
                    stact1(a.ACT, base + (size_t)v * a.Cv + c, y * sg * ds, a.s_bf16);
                    const float gy = Gt[(size_t)v * rs + c] * ds * (sg * (1.0f + y * (1.0f - sg)));
                    dg += gy * xh; dbt += gy;
                    V[(size_t)v * rs + c] = xh;
                    Gt[(size_t)v * rs + c] = gy * gm;
                }
            }
            red[tid] = dg; red[GN_THREADS + tid] = dbt;
            __syncthreads();
            if (tid < a.Cv) {
                float sg2 = 0.f, sb2 = 0.f;
                for (int j = 0; j < R; ++j) { sg2 += red[j * a.Cv + tid]; sb2 += red[GN_THREADS + j * a.Cv + tid]; }
                atomicAdd(a.dgamma + tid, sg2);
                atomicAdd(a.dbeta + tid, sb2);
            }
        }
        if (g < G) {
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
        for (int i = tid; i < a.HWv * a.Cv; i += GN_THREADS) {
            const int v = i / a.Cv, c = i - v * a.Cv, gg = c / Cg;
            Gt[(size_t)v * rs + c] = stat[4 * gg + 1] * (Gt[(size_t)v * rs + c] - stat[4 * gg + 2] - V[(size_t)v * rs + c] * stat[4 * gg + 3]);
        }
    }
    __syncthreads();
    // scatter GV (in Gt) to the sources' gradient accumulators.  Four read-modify-writes per work-item are in flight at a time: the
    // loads of a group are issued before its stores (one load -> add -> store chain per iteration exposes a memory latency each).
    constexpr int SU = 4;
    if (a.gA) {
        float* ga = a.gA + (size_t)n * a.HWa * a.CA;
        const int tot = a.HWa * a.CA;
        for (int i0 = tid; i0 < tot; i0 += SU * GN_THREADS) {
            float old[SU], acc[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = i0 + u * GN_THREADS; old[u] = i < tot ? ga[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = i0 + u * GN_THREADS;
                acc[u] = 0.f;
                if (i < tot) {
                    const int sp = i / a.CA, c = i - sp * a.CA;
                    if (a.inv_start) { for (int k = a.inv_start[sp]; k < a.inv_start[sp + 1]; ++k) acc[u] += Gt[(size_t)a.inv_list[k] * rs + c]; }
                    else acc[u] = Gt[(size_t)sp * rs + c];
                }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = i0 + u * GN_THREADS; if (i < tot) ga[i] = old[u] + acc[u]; }
        }
    }
    if (a.gB) {
        float* gb = a.gB + (size_t)n * a.HWv * a.CB;
        const int tot = a.HWv * a.CB;
        for (int i0 = tid; i0 < tot; i0 += SU * GN_THREADS) {
            float old[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = i0 + u * GN_THREADS; old[u] = i < tot ? gb[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = i0 + u * GN_THREADS;
                if (i < tot) { const int v = i / a.CB, c = i - v * a.CB; gb[i] = old[u] + Gt[(size_t)v * rs + a.CA + c]; }
            }
        }
    }
    if (a.tail.G) {
        __syncthreads();                                         // this workgroup's scatter stores are complete and visible to all of its work-items
        bwd_scale_colsum_body<GN_THREADS>(a.tail, n, tid, red);
    }
}

// Weight gradient: dW[co][ci][tap] (reference OIHW layout, or NIN [ci][co] / Linear [co][ci] through the strides)
//   += sum_{n, o} ACT[n][in(o, tap)][ci] * G[n][o][co]
// grid = (ksplit, ceil(Cin/32), ceil(Cout/64)).  A workgroup owns the 32(ci) x 64(co) tile of ALL taps for a slice of the
// samples (split-K over the batch, merged with fp32 atomics).  Per chunk of S samples it stages the ACT rows [S*HWv][32]
// and the G rows [S*HWo][64] in LDS (register-prefetched one chunk ahead); the taps are row-offset views of the staged
// ACT through a byte table (out-of-image -> a zero row), so one G fragment feeds NTAP MFMAs.  Wave w: ci half w&1,
// co half w>>1, accumulators [NTAP][2].
struct WgradArgs {
    const float* ACT; const float* G; float* dW;
    const int* tab;        // [HWo][ntap] input pixel of (output pixel, tap) or -1   (null: identity, 1 tap)
    int NB, HWv, HWo, Cin, Cout, ntap;
    int lda;               // channels per pixel of ACT (>= Cin: padded input channels; multiple of 4)
    int ksplit;            // number of sample slices
    int S;                 // samples per staged chunk: S*HWv <= 255, S*HWo <= 128
    long s_co, s_ci, s_t;  // strides of dW
    int s_bf16;            // ACT and G are bf16 in HBM (widened to fp32 as they are staged)
    int bf16;              // 1: v_mfma_f32_16x16x32_bf16 over 32 staged rows per step (operands rounded to bf16 as they leave LDS, fp32 accumulate)
};
#define WG_AS 40
#define WG_GS 72
__host__ __device__ inline int wgrad_chunk(int HWv, int HWo) {
    int s = 128 / HWo; if (255 / HWv < s) s = 255 / HWv;
    return s < 1 ? 1 : s;
}
__host__ __device__ inline size_t wgrad_lds_bytes(int HWv, int HWo, int S, int bf16 = 0) {
    const int RI = S * HWv, RO4 = bf16 ? ((S * HWo + 31) & ~31) : ((S * HWo + 3) & ~3);
    const size_t stage = (size_t)(RI + 1) * WG_AS + (size_t)RO4 * WG_GS + (size_t)RO4 * 4, flush = (size_t)64 * (32 * 9 + 1);
    return (stage > flush ? stage : flush) * 4;
}

template <int NTAP, bool BF16 = false>
__global__ __launch_bounds__(RDMI_THREADS) void wgrad_mfma_kernel(WgradArgs a) {
    constexpr int NA = 8, NG = 8;   // float4 slots per work-item: 255 rows x 8 / 256, 128 rows x 16 / 256
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, kq = lane >> 4;
    const int S = a.S, RI = S * a.HWv, RO = S * a.HWo, RO4 = BF16 ? ((RO + 31) & ~31) : ((RO + 3) & ~3);   // staged G rows (bf16: whole 32-row MFMA steps)
    float* Al = reinterpret_cast<float*>(rdmi_lds);            // [RI + 1][WG_AS]; row RI = zeros
    float* Gl = Al + (RI + 1) * WG_AS;                         // [RO4][WG_GS]
    uint32_t* rt = reinterpret_cast<uint32_t*>(Gl + RO4 * WG_GS);   // [RO4][4]: byte t = staged ACT row of (row, tap t)
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 64;
    for (int e = tid; e < RO4 * 4; e += RDMI_THREADS) {
        const int row = e >> 2, w = e & 3;
        uint32_t v = 0;
        for (int bb = 0; bb < 4; ++bb) {
            const int t = 4 * w + bb;
            int idx = RI;
            if (t < NTAP && row < RO) {
                const int sidx = row / a.HWo, o = row - sidx * a.HWo;
                const int pix = a.tab ? a.tab[o * NTAP + t] : o;
                if (pix >= 0) idx = sidx * a.HWv + pix;
            }
            v |= (uint32_t)idx << (8 * bb);
        }
        rt[e] = v;
    }
    if (tid < WG_AS) Al[RI * WG_AS + tid] = 0.f;
    const int chunks = (a.NB + S - 1) / S, cper = (chunks + a.ksplit - 1) / a.ksplit;
    const int n_lo = blockIdx.x * cper * S, n_hi = min(a.NB, n_lo + cper * S);
    const bool gvec = (a.Cout & 3) == 0;
    f32x4 ra[NA], rg[NG];
    auto fetch = [&](int n0) {
        const int nv = min(S, n_hi - n0);
        const int rows_a = nv * a.HWv, rows_g = nv * a.HWo;
        const size_t Ab = (size_t)n0 * a.HWv * a.lda, Gb = (size_t)n0 * a.HWo * a.Cout;   // element offsets
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = i * RDMI_THREADS + tid, row = e >> 3, c = ci0 + 4 * (e & 7);
            ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows_a && c + 3 < a.lda) ra[i] = a.s_bf16 ? ldact4(a.ACT, Ab + (size_t)row * a.lda + c, 1) : ldg4(a.ACT + Ab + (size_t)row * a.lda + c);
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int e = i * RDMI_THREADS + tid, row = e >> 4, c = co0 + 4 * (e & 15);
            rg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows_g) {
                const size_t gi = Gb + (size_t)row * a.Cout + c;
                if (gvec) { if (c + 3 < a.Cout) rg[i] = a.s_bf16 ? ldact4(a.G, gi, 1) : ldg4(a.G + gi); }
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (c + q < a.Cout) rg[i][q] = a.s_bf16 ? ldact1(a.G, gi + q, 1) : ldg1(a.G + gi + q);
                }
            }
        }
    };
    f32x4 acc[NTAP][2];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
    if (n_lo < n_hi) fetch(n_lo);
    const int acol = (wave & 1) * 16 + lrow, gcol = (wave >> 1) * 32 + lrow;
    for (int n0 = n_lo; n0 < n_hi; n0 += S) {
        __syncthreads();                                        // previous chunk fully consumed (and table / zero row written)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = i * RDMI_THREADS + tid, row = e >> 3;
            if (row < RI) *reinterpret_cast<f32x4*>(Al + row * WG_AS + 4 * (e & 7)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int e = i * RDMI_THREADS + tid, row = e >> 4;
            if (row < RO4) *reinterpret_cast<f32x4*>(Gl + row * WG_GS + 4 * (e & 15)) = rg[i];
        }
        __syncthreads();
        if (n0 + S < n_hi) fetch(n0 + S);
        if (BF16) {
            for (int ks = 0; ks < RO4; ks += 32) {
                const int r0 = ks + kq * 8;                         // this lane's 8 consecutive k (staged G rows)
                float g0[8], g1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { g0[j] = Gl[(r0 + j) * WG_GS + gcol]; g1[j] = Gl[(r0 + j) * WG_GS + gcol + 16]; }
                const u32x4 b0 = {pack_bf16x2(g0[0], g0[1]), pack_bf16x2(g0[2], g0[3]), pack_bf16x2(g0[4], g0[5]), pack_bf16x2(g0[6], g0[7])};
                const u32x4 b1 = {pack_bf16x2(g1[0], g1[1]), pack_bf16x2(g1[2], g1[3]), pack_bf16x2(g1[4], g1[5]), pack_bf16x2(g1[6], g1[7])};
                if (NTAP == 1) {
                    float av[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = Al[min(r0 + j, RI) * WG_AS + acol];
                    const u32x4 af = {pack_bf16x2(av[0], av[1]), pack_bf16x2(av[2], av[3]), pack_bf16x2(av[4], av[5]), pack_bf16x2(av[6], av[7])};
                    acc[0][0] = mfma16_bf16(af, b0, acc[0][0]); acc[0][1] = mfma16_bf16(af, b1, acc[0][1]);
                } else {
                    uint32_t w[8][3];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { w[j][0] = rt[(r0 + j) * 4]; w[j][1] = rt[(r0 + j) * 4 + 1]; w[j][2] = rt[(r0 + j) * 4 + 2]; }
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) {
                        float av[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t ww = t < 4 ? w[j][0] : t < 8 ? w[j][1] : w[j][2];
                            av[j] = Al[(int)((ww >> (8 * (t & 3))) & 255u) * WG_AS + acol];
                        }
                        const u32x4 af = {pack_bf16x2(av[0], av[1]), pack_bf16x2(av[2], av[3]), pack_bf16x2(av[4], av[5]), pack_bf16x2(av[6], av[7])};
                        acc[t][0] = mfma16_bf16(af, b0, acc[t][0]); acc[t][1] = mfma16_bf16(af, b1, acc[t][1]);
                    }
                }
            }
        } else
        for (int ks = 0; ks < RO4; ks += 4) {
            const int row = ks + kq;
            const float b0 = Gl[row * WG_GS + gcol], b1 = Gl[row * WG_GS + gcol + 16];
            if (NTAP == 1) {
                const float av = Al[min(row, RI) * WG_AS + acol];
                acc[0][0] = mfma16(av, b0, acc[0][0]); acc[0][1] = mfma16(av, b1, acc[0][1]);
            } else {
                const uint32_t w0 = rt[row * 4], w1 = rt[row * 4 + 1], w2 = rt[row * 4 + 2];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const uint32_t w = t < 4 ? w0 : t < 8 ? w1 : w2;
                    const int idx = (int)((w >> (8 * (t & 3))) & 255u);
                    const float av = Al[idx * WG_AS + acol];
                    acc[t][0] = mfma16(av, b0, acc[t][0]); acc[t][1] = mfma16(av, b1, acc[t][1]);
                }
            }
        }
    }
    // D[row = ci_local][col = co_local]: lane holds col = lrow, rows kq*4 + r.
    if (NTAP == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + (wave & 1) * 16 + kq * 4 + r, co = co0 + (wave >> 1) * 32 + j * 16 + lrow;
                if (ci < a.Cin && co < a.Cout) atomicAdd(a.dW + co * a.s_co + ci * a.s_ci, acc[0][j][r]);
            }
    } else {
        // transpose through LDS to the OIHW order ([co][ci][tap] is contiguous per co) so that one atomic instruction
        // covers 64 consecutive floats instead of 64 different cache lines
        constexpr int RS = 32 * NTAP + 1;
        float* stg = reinterpret_cast<float*>(rdmi_lds);       // [64][RS]
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stg[((wave >> 1) * 32 + j * 16 + lrow) * RS + ((wave & 1) * 16 + kq * 4 + r) * NTAP + t] = acc[t][j][r];
        __syncthreads();
        for (int e = tid; e < 64 * 32 * NTAP; e += RDMI_THREADS) {
            const int col = e / (32 * NTAP), rem = e - col * (32 * NTAP), cil = rem / NTAP, t = rem - cil * NTAP;
            const int ci = ci0 + cil, co = co0 + col;
            if (ci < a.Cin && co < a.Cout) atomicAdd(a.dW + co * a.s_co + ci * a.s_ci + t * a.s_t, stg[col * RS + rem]);
        }
    }
}

// C[M][N] (+)= A[M][K] . B[K][N] with arbitrary strides, optional elementwise pre-activation of A / B (0 none, 1 SiLU);
// one work-item per output element (tiny embedding GEMMs: K <= 2048).
struct SgemmArgs {
    const float* A; long a_m, a_k; int a_act;
    const float* B; long b_k, b_n; int b_act;
    float* C; long c_m, c_n; int accumulate;
    int M, N, K;
    int no_split;   // C is not pre-zeroed: keep K in one workgroup
};
// C[M][N] (+)= act(A)[M][K] * act(B)[K][N], any strides.  64x64 output tile per workgroup (4 waves x 2x2 MFMA tiles),
// K split over blockIdx.z (atomics when split or accumulating; the caller zeroes C first in that case).
__device__ __forceinline__ void small_gemm_body(const SgemmArgs& a, int kz, int nkz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    if ((int)blockIdx.x * 64 >= a.M || (int)blockIdx.y * 64 >= a.N) return;
    const int m0 = blockIdx.x * 64 + (wave >> 1) * 32, n0 = blockIdx.y * 64 + (wave & 1) * 32;
    const int kc = (a.K + nkz - 1) / nkz;
    const int kb = kz * kc, ke = min(a.K, kb + kc);
    if (kb >= ke) return;
    f32x4 acc[2][2] = {};
    long ao[2], bo[2];
    bool av[2], bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + col, n = n0 + i * 16 + col;
        av[i] = m < a.M; bv[i] = n < a.N;
        ao[i] = (long)min(m, a.M - 1) * a.a_m;
        bo[i] = (long)min(n, a.N - 1) * a.b_n;
    }
    for (int k = kb; k < ke; k += 4) {
        const int kk = k + kq;
        const bool kv = kk < ke;
        const long ka = (long)min(kk, ke - 1);
        float x[2], y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x[i] = ldg1(a.A + ao[i] + ka * a.a_k);
            y[i] = ldg1(a.B + bo[i] + ka * a.b_k);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (a.a_act) x[i] = silu_f(x[i]);
            if (a.b_act) y[i] = silu_f(y[i]);
            x[i] = (kv && av[i]) ? x[i] : 0.f;
            y[i] = (kv && bv[i]) ? y[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(x[i], y[j], acc[i][j]);
    }
    const bool atomic = a.accumulate || nkz > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + kq * 4 + r, n = n0 + j * 16 + col;
                if (m < a.M && n < a.N) {
                    float* c = a.C + (long)m * a.c_m + (long)n * a.c_n;
                    if (atomic) atomicAdd(c, acc[i][j][r]);
                    else *c = acc[i][j][r];
                }
            }
}

// Several independent small GEMMs in one launch: blockIdx.z = job * ks + K slice (a no_split job uses slice 0 only);
// grid x / y cover the largest job, surplus workgroups of the smaller ones exit at once.
__global__ __launch_bounds__(RDMI_THREADS) void small_gemm_jobs_kernel(const SgemmArgs* __restrict__ jobs, int ks) {
    const int j = blockIdx.z / ks, kz = blockIdx.z - j * ks;
    const SgemmArgs a = jobs[j];
    if (a.no_split) { if (kz == 0) small_gemm_body(a, 0, 1); }
    else small_gemm_body(a, kz, ks);
}

// g <- g * silu'(x)
__global__ __launch_bounds__(RDMI_THREADS) void silu_bwd_kernel(float* __restrict__ g, const float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= n) return;
    const float y = x[i], sg = 1.0f / (1.0f + __expf(-y));
    g[i] *= sg * (1.0f + y * (1.0f - sg));
}

// Batched column sums: job j adds the column sums of X_j [M][C_j] (row stride ldx) to out_j.  grid (max C / 64, row slabs, jobs).
struct ColsumJob { const float* X; float* out; int C; int pad; };
__global__ __launch_bounds__(RDMI_THREADS) void colsum_jobs_kernel(const ColsumJob* __restrict__ jobs, int M, int ldx) {
    __shared__ float red[RDMI_THREADS];
    const ColsumJob jb = jobs[blockIdx.z];
    if ((int)blockIdx.x * 64 >= jb.C) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    const int per = (M + (int)gridDim.y - 1) / (int)gridDim.y;
    const int mb = blockIdx.y * per, me = min(M, mb + per);
    float s = 0.f;
    if (c < jb.C)
        for (int m = mb + r; m < me; m += 4) s += jb.X[(size_t)m * ldx + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (r == 0 && c < jb.C) atomicAdd(jb.out + c, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
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

// AttnBlockpp backward on the fp32 MFMA (C = 64, L <= 96).  A workgroup (8 waves) walks samples blockIdx.x, +gridDim.x, ...
// with every tensor of one sample in LDS ([L][C+4] / [L][L+4] rows), and keeps the four NIN weight gradients as MFMA
// accumulators across its samples (one atomic pass at the end instead of one per sample).
// Recomputes xn, q, k, v, P, O from x, then:  gH = s*gOut;  dW3 += O^T gH; gO = gH W3^T;  gP = gO V^T; gV = P^T gO;
// gS = P * (gP - rowsum(gP * P)) / sqrt(C);  gQ = gS K; gK = gS^T Q;  dWq/k/v += xn^T g{Q,K,V};  gxn = sum g. W^T;
// GroupNorm backward (no activation) -> gx;  gX += gx + gH (residual branch).
#define AB_THREADS 512
struct AttnBwdArgs {
    const float* x; const float* gOut; float* gX;
    const float* gamma; const float* beta; float* dgamma; float* dbeta;
    const float* W[4]; const float* b[4];      // NIN_0..3: W [in][out]
    float* dW[4]; float* db[4];
    int NB, L, G; float eps, scale, out_scale;
    int x_bf16;                                // x is bf16 in HBM (train_dtype = bf16)
};

template <int C>
__host__ __device__ inline size_t attn_bwd_lds_bytes(int L, int G) {
    const int LD = C + 4, LP = L + 4;
    const size_t big = (size_t)L * (LP > LD ? LP : LD);
    return ((size_t)4 * L * LD + 2 * big + 4 * G + 2 * C + 2 * (AB_THREADS / 64) * G + 8 * C) * 4;
}

// MFMA tile helpers: a lane's operand index is tile*16 + (lane & 15), its k index 4*step + (lane >> 4).  NT output tiles
// advance together per k-step (independent accumulators, one shared fragment) so LDS / global latency overlaps the MFMAs.
template <int NT, class FA, class FB>
__device__ __forceinline__ void ab_shareB(int steps, int kq, FA fa, FB fb, f32x4 (&acc)[NT]) {   // acc[i] += A_i B
#pragma unroll 2
    for (int s = 0; s < steps; ++s) {
        const int k = 4 * s + kq;
        const float b = fb(k);
        float av[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) av[i] = fa(i, k);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = mfma16(av[i], b, acc[i]);
    }
}
template <int NT, class FA, class FB>
__device__ __forceinline__ void ab_shareA(int steps, int kq, FA fa, FB fb, f32x4 (&acc)[NT]) {   // acc[i] += A B_i
#pragma unroll 2
    for (int s = 0; s < steps; ++s) {
        const int k = 4 * s + kq;
        const float av = fa(k);
        float bv[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) bv[i] = fb(i, k);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = mfma16(av, bv[i], acc[i]);
    }
}
// same with the shared B fragment of a K = 64 contraction already in registers (a global weight column, loaded up front)
template <int NT, class FA>
__device__ __forceinline__ void ab_regB(int kq, FA fa, const float (&bw)[16], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float av[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) av[i] = fa(i, 4 * s + kq);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = mfma16(av[i], bw[s], acc[i]);
    }
}

template <int C>
__global__ __launch_bounds__(AB_THREADS) void attn_bwd_kernel(AttnBwdArgs a) {
    static_assert(C == 64, "lane == channel below");
    constexpr int NW = AB_THREADS / 64, LD = C + 4, CT = C / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    const int L = a.L, LP = L + 4, MT = (L + 15) >> 4, KL = (L + 3) >> 2, G = a.G, Cg = C / G;
    const float cnt = (float)(Cg * L);
    const int big = L * (LP > LD ? LP : LD);
    float* Q = reinterpret_cast<float*>(rdmi_lds);
    float* K = Q + L * LD; float* V = K + L * LD; float* O = V + L * LD;
    float* P = O + L * LD;            // [L][LP] probabilities; later xhat [L][LD]
    float* GP = P + big;              // gH [L][LD]; later gP -> gS [L][LP]
    float* stat = GP + big;           // [G][4] mean, rstd, m1, m2
    float* gam = stat + 4 * G; float* bet = gam + C;
    float* red = bet + C;             // [NW][G][2]
    float* cred = red + 2 * NW * G;   // [8][C] column reductions at the very end
    if (tid < C) { gam[tid] = a.gamma[tid]; bet[tid] = a.beta[tid]; }
    f32x4 dw3[2] = {}, dw012[6] = {};
    float dbq = 0.f, dbk = 0.f, dbv = 0.f, dbh = 0.f, dgm = 0.f, dbt = 0.f;   // this work-item's channel = lane

    // group reduction of two per-work-item partials (channel = lane, so a group is Cg neighbouring lanes)
    auto group_reduce = [&](float u, float v, int slot) {
        for (int off = 1; off < Cg; off <<= 1) { u += __shfl_xor(u, off); v += __shfl_xor(v, off); }
        if ((lane & (Cg - 1)) == 0) { red[(wave * G + lane / Cg) * 2] = u; red[(wave * G + lane / Cg) * 2 + 1] = v; }
        __syncthreads();
        if (tid < G) {
            float su = 0.f, sv = 0.f;
            for (int w = 0; w < NW; ++w) { su += red[(w * G + tid) * 2]; sv += red[(w * G + tid) * 2 + 1]; }
            stat[4 * tid + slot] = su; stat[4 * tid + slot + 1] = sv;
        }
        __syncthreads();
    };

    for (int n = blockIdx.x; n < a.NB; n += gridDim.x) {
        const size_t xo = (size_t)n * L * C;
        const float* gog = a.gOut + (size_t)n * L * C;
        // ---- stage x (-> O) and gH (-> GP); GroupNorm statistics; O <- xn
        for (int p = wave; p < L; p += NW) { O[p * LD + lane] = a.x_bf16 ? ldact1(a.x, xo + p * C + lane, 1) : ldg1(a.x + xo + p * C + lane); GP[p * LD + lane] = ldg1(gog + p * C + lane) * a.out_scale; }
        {
            float s1 = 0.f;
            for (int p = wave; p < L; p += NW) s1 += O[p * LD + lane];
            group_reduce(s1, 0.f, 0);
            const float mean = stat[4 * (lane / Cg)] / cnt;
            float s2 = 0.f;
            for (int p = wave; p < L; p += NW) { const float d = O[p * LD + lane] - mean; s2 += d * d; }
            __syncthreads();                                   // everyone has read the sum before slot 0/1 are rewritten
            group_reduce(s2, 0.f, 2);
            if (tid < G) { const float m = stat[4 * tid] / cnt; stat[4 * tid + 1] = 1.0f / sqrtf(stat[4 * tid + 2] / cnt + a.eps); stat[4 * tid] = m; }
            __syncthreads();
            const float m = stat[4 * (lane / Cg)], rs = stat[4 * (lane / Cg) + 1];
            for (int p = wave; p < L; p += NW) O[p * LD + lane] = (O[p * LD + lane] - m) * rs * gam[lane] + bet[lane];
        }
        __syncthreads();
        // wave roles for [L][C] outputs: channel tile wn, row tiles wg, wg+2, wg+4 (three accumulators share a B fragment)
        const int wn = wave & 3, wg = wave >> 2;
        int mrow[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) mrow[i] = min((wg + 2 * i) * 16 + col, L - 1);
        const int ncol = wn * 16 + col;
        auto store3 = [&](float* dst, const f32x4 (&acc)[3], float addv) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) { const int mm = (wg + 2 * i) * 16 + kq * 4 + r4; if (mm < L) dst[mm * LD + ncol] = acc[i][r4] + addv; }
        };
        // ---- A: q, k, v = xn W + b   (the weight column of this wave's channel tile is loaded to registers up front)
#pragma unroll 1
        for (int which = 0; which < 3; ++which) {
            const float* W = a.W[which];
            float bw[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) bw[st] = ldg1(W + (4 * st + kq) * C + ncol);
            f32x4 acc[3] = {};
            ab_regB<3>(kq, [&](int i, int k) { return O[mrow[i] * LD + k]; }, bw, acc);
            store3(which == 0 ? Q : which == 1 ? K : V, acc, ldg1(a.b[which] + ncol));
        }
        __syncthreads();
        // ---- B: P = softmax(Q K^T * scale): wave w < MT owns key tile w for all row tiles
        if (wave < MT) {
            const int nn = wave * 16 + col, nc = min(nn, L - 1);
            f32x4 acc[6] = {};
            int mr[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) mr[i] = min(i * 16 + col, L - 1);
            ab_shareB<6>(C / 4, kq, [&](int i, int k) { return Q[mr[i] * LD + k]; }, [&](int k) { return K[nc * LD + k]; }, acc);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) { const int mm = i * 16 + kq * 4 + r4; if (mm < L && nn < L) P[mm * LP + nn] = acc[i][r4] * a.scale; }
        }
        __syncthreads();
        for (int r = wave; r < L; r += NW) {
            const float s0 = lane < L ? P[r * LP + lane] : -3.0e38f, s1 = lane + 64 < L ? P[r * LP + lane + 64] : -3.0e38f;
            float mx = fmaxf(s0, s1);
            for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float e0 = lane < L ? __expf(s0 - mx) : 0.f, e1 = lane + 64 < L ? __expf(s1 - mx) : 0.f;
            float sum = e0 + e1;
            for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
            const float inv = 1.0f / sum;
            if (lane < L) P[r * LP + lane] = e0 * inv;
            if (lane + 64 < L) P[r * LP + lane + 64] = e1 * inv;
        }
        __syncthreads();
        // ---- C: O = P V
        {
            f32x4 acc[3] = {};
            ab_shareB<3>(KL, kq, [&](int i, int k) { return k < L ? P[mrow[i] * LP + k] : 0.f; }, [&](int k) { return k < L ? V[k * LD + ncol] : 0.f; }, acc);
            store3(O, acc, 0.f);
        }
        __syncthreads();
        // ---- D: dW3 += O^T gH (two channel tiles share the O^T fragment);  gO = gH W3^T (registers, then over O)
        {
            const int j = (wave >> 1) * 16 + col, c0 = (wave & 1) * 32 + col;
            ab_shareA<2>(KL, kq, [&](int k) { return k < L ? O[k * LD + j] : 0.f; }, [&](int i, int k) { return k < L ? GP[k * LD + c0 + 16 * i] : 0.f; }, dw3);
        }
        for (int p = wave; p < L; p += NW) dbh += GP[p * LD + lane];
        f32x4 hold[6] = {};
        {
            const float* W3 = a.W[3];
            float bw[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) bw[st] = ldg1(W3 + ncol * C + 4 * st + kq);
            f32x4 (&h3)[3] = reinterpret_cast<f32x4(&)[3]>(hold[0]);
            ab_regB<3>(kq, [&](int i, int k) { return GP[mrow[i] * LD + k]; }, bw, h3);
        }
        __syncthreads();
        store3(O, reinterpret_cast<f32x4(&)[3]>(hold[0]), 0.f);
        __syncthreads();
        // ---- E: gP = gO V^T -> GP (gH is dead);  gV = P^T gO -> registers
        if (wave < MT) {
            const int nn = wave * 16 + col, nc = min(nn, L - 1);
            f32x4 acc[6] = {};
            int mr[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) mr[i] = min(i * 16 + col, L - 1);
            ab_shareB<6>(C / 4, kq, [&](int i, int k) { return O[mr[i] * LD + k]; }, [&](int k) { return V[nc * LD + k]; }, acc);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) { const int mm = i * 16 + kq * 4 + r4; if (mm < L && nn < L) GP[mm * LP + nn] = acc[i][r4]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) hold[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab_shareB<3>(KL, kq, [&](int i, int k) { return k < L ? P[k * LP + mrow[i]] : 0.f; }, [&](int k) { return k < L ? O[k * LD + ncol] : 0.f; },
                     reinterpret_cast<f32x4(&)[3]>(hold[0]));
        __syncthreads();
        // ---- F: gS = P * (gP - rowsum(gP * P)) * scale over GP;  O <- gV
        for (int r = wave; r < L; r += NW) {
            const float p0 = lane < L ? P[r * LP + lane] : 0.f, p1 = lane + 64 < L ? P[r * LP + lane + 64] : 0.f;
            const float g0 = lane < L ? GP[r * LP + lane] : 0.f, g1 = lane + 64 < L ? GP[r * LP + lane + 64] : 0.f;
            float d = p0 * g0 + p1 * g1;
            for (int off = 32; off; off >>= 1) d += __shfl_xor(d, off);
            if (lane < L) GP[r * LP + lane] = p0 * (g0 - d) * a.scale;
            if (lane + 64 < L) GP[r * LP + lane + 64] = p1 * (g1 - d) * a.scale;
        }
        store3(O, reinterpret_cast<f32x4(&)[3]>(hold[0]), 0.f);
        __syncthreads();
        // ---- G: gQ = gS K, gK = gS^T Q -> registers;  P region <- xhat
#pragma unroll
        for (int i = 0; i < 6; ++i) hold[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab_shareB<3>(KL, kq, [&](int i, int k) { return k < L ? GP[mrow[i] * LP + k] : 0.f; }, [&](int k) { return k < L ? K[k * LD + ncol] : 0.f; },
                     reinterpret_cast<f32x4(&)[3]>(hold[0]));
        ab_shareB<3>(KL, kq, [&](int i, int k) { return k < L ? GP[k * LP + mrow[i]] : 0.f; }, [&](int k) { return k < L ? Q[k * LD + ncol] : 0.f; },
                     reinterpret_cast<f32x4(&)[3]>(hold[3]));
        {
            const float m = stat[4 * (lane / Cg)], rs = stat[4 * (lane / Cg) + 1];
            for (int p = wave; p < L; p += NW) P[p * LD + lane] = ((a.x_bf16 ? ldact1(a.x, xo + p * C + lane, 1) : ldg1(a.x + xo + p * C + lane)) - m) * rs;
        }
        __syncthreads();
        store3(Q, reinterpret_cast<f32x4(&)[3]>(hold[0]), 0.f);
        store3(K, reinterpret_cast<f32x4(&)[3]>(hold[3]), 0.f);
        __syncthreads();
        // ---- H: Q = gQ, K = gK, O = gV.  dW0..2 += xn^T g. (six tiles share the xn^T fragment of input-channel tile wn:
        //      tile i = which*2 + c, channel tile 2*wg + c);  bias partials;  V <- gxn = gQ W0^T + gK W1^T + gV W2^T
        {
            const int j = wn * 16 + col, c0 = wg * 32 + col;
            const float gj = gam[j], bj = bet[j];
            ab_shareA<6>(KL, kq, [&](int k) { return k < L ? P[k * LD + j] * gj + bj : 0.f; },
                         [&](int i, int k) { const float* src = (i >> 1) == 0 ? Q : (i >> 1) == 1 ? K : O; return k < L ? src[k * LD + c0 + 16 * (i & 1)] : 0.f; }, dw012);
        }
        for (int p = wave; p < L; p += NW) { dbq += Q[p * LD + lane]; dbk += K[p * LD + lane]; dbv += O[p * LD + lane]; }
        {
            f32x4 acc[3] = {};
#pragma unroll 1
            for (int which = 0; which < 3; ++which) {
                const float* W = a.W[which];
                const float* src = which == 0 ? Q : which == 1 ? K : O;
                float bw[16];
#pragma unroll
                for (int st = 0; st < 16; ++st) bw[st] = ldg1(W + ncol * C + 4 * st + kq);
                ab_regB<3>(kq, [&](int i, int k) { return src[mrow[i] * LD + k]; }, bw, acc);
            }
            store3(V, acc, 0.f);
        }
        __syncthreads();
        // ---- J: GroupNorm backward (no activation), residual branch
        {
            float u = 0.f, v = 0.f;
            for (int p = wave; p < L; p += NW) {
                const float gy = V[p * LD + lane], xh = P[p * LD + lane];
                dgm += gy * xh; dbt += gy;
                const float gxh = gy * gam[lane];
                u += gxh; v += gxh * xh;
            }
            group_reduce(u, v, 2);
            const int g = lane / Cg;
            const float rs = stat[4 * g + 1], m1 = stat[4 * g + 2] / cnt, m2 = stat[4 * g + 3] / cnt;
            float* gxo = a.gX + (size_t)n * L * C;
            for (int p0 = wave; p0 < L; p0 += 4 * NW) {           // four rows per pass: the accumulator / gradient loads before the stores
                float old[4], go[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = min(p0 + u * NW, L - 1);
                    old[u] = gxo[p * C + lane]; go[u] = ldg1(gog + p * C + lane);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * NW;
                    if (p < L) {
                        const float gxh = V[p * LD + lane] * gam[lane];
                        gxo[p * C + lane] = old[u] + (rs * (gxh - m1 - P[p * LD + lane] * m2) + go[u] * a.out_scale);
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- flush the per-workgroup parameter gradients (tile roles as in phases D and H)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) atomicAdd(a.dW[3] + ((wave >> 1) * 16 + kq * 4 + r4) * C + (wave & 1) * 32 + 16 * i + col, dw3[i][r4]);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            atomicAdd(a.dW[i >> 1] + ((wave & 3) * 16 + kq * 4 + r4) * C + (wave >> 2) * 32 + 16 * (i & 1) + col, dw012[i][r4]);
    float* outs[6] = {a.db[0], a.db[1], a.db[2], a.db[3], a.dgamma, a.dbeta};
    const float vals[6] = {dbq, dbk, dbv, dbh, dgm, dbt};
    for (int q = 0; q < 6; ++q) {
        __syncthreads();
        cred[wave * C + lane] = vals[q];
        __syncthreads();
        if (tid < C) {
            float sum = 0.f;
            for (int w = 0; w < NW; ++w) sum += cred[w * C + tid];
            atomicAdd(outs[q] + tid, sum);
        }
    }
}

// loss backward: gscore[i] = gper[b] * dper_dscore[i]
__global__ __launch_bounds__(RDMI_THREADS) void rowscale_kernel(const float* __restrict__ gper, const float* __restrict__ d, float* __restrict__ out,
                                                                 int B, int E) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < (long)B * E) out[i] = gper[i / E] * d[i];
}
