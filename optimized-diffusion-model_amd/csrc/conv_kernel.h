// Fused [gather/concat/resize -> GroupNorm -> SiLU] -> 3x3 conv (+1x1 NIN shortcut) -> [bias, temb, residual,
// 1/sqrt2] as ONE implicit-GEMM kernel on the fp32 MFMA (v_mfma_f32_16x16x4_f32), for gfx950.
//
// Mirrors, per launch, one of (reference, RD/ = Reflected-Diffusion/):
//   ResnetBlockDDPMpp first half   GN0 -> SiLU -> Conv_0 -> + Dense_0(SiLU(temb))        RD/models/layerspp.py:200-202
//   ResnetBlockDDPMpp second half  GN1 -> SiLU -> Conv_1 ; x or NIN_0(x) ; (x+h)/sqrt2   RD/models/layerspp.py:203-214
//   Downsample  F.pad(0,1,0,1) -> conv3x3 stride 2                                        RD/models/layerspp.py:157-159
//   Upsample    nearest x2 -> conv3x3                                                     RD/models/layerspp.py:122-124
//   skip concat + nearest shape fix feeding an up block                                   RD/models/ncsnpp.py:319-325
//   input_conv / out_norm+out_act+out_conv                                                RD/models/ncsnpp.py:266,343-347
//
// Work decomposition (MI355X-first, not a cuDNN-style port):
//   * one workgroup = S whole samples x BN output channels.  Because a workgroup owns whole samples, the
//     GroupNorm statistics of its inputs are computed in LDS by the consumer itself: no separate
//     normalisation pass or launch, activations are read from HBM/L2 exactly once per N-slab.
//   * the activated input tile lives in LDS pixel-major ([pixel][C+4] fp32); the 9 taps are 9 row-offset
//     views of it (host-built table, out-of-image taps point at a zero row) -> no im2col, no halo copies.
//   * GEMM: rows = (sample, output pixel), cols = output channels, K = (tap, input channel).  Each lane
//     feeds 4 MFMAs from one ds_read_b128 (A) and one global_load_dwordx4 (B, weights pre-packed as
//     [tap][C/16][Cout][16] so a wave reads 1 KiB contiguous from L2).
//   * the NIN shortcut is folded in as one more K phase over the raw block input (second LDS region).
#pragma once
#include "common.h"
#include "misc_kernels.h"

// dropout keep-mask for element (op, n, pixel, channel): Philox keyed by the step seed; the backward recomputes it
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t op, uint64_t elem, float p) {
    if (p <= 0.f) return 1.f;
    uint32_t c[4] = {(uint32_t)(elem >> 2), (uint32_t)(elem >> 34), op, 0x44524f50u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u = ((float)c[elem & 3] + 0.5f) * 2.3283064365386963e-10f;
    return u < p ? 0.f : 1.0f / (1.0f - p);
}

struct ConvArgs {
    // virtual conv input = concat_c( gather(srcA, mapA) , srcB ), optionally GroupNorm+SiLU'd
    const float* srcA; const float* srcB;
    const int* mapA;              // [HWv] source pixel of A for each virtual pixel
    const float* gamma; const float* beta;
    // raw block input for the NIN shortcut phase (same gather form)
    const float* scA; const float* scB; const int* mapSc;
    const float* wpk;             // packed conv weights   [ntap][Cv/16][Cout_pad][16]
    const float* wsc;             // packed NIN weights          [Csc/16][Cout_pad][16]
    const float* bias; const float* bias_sc;
    const float* dense;           // per-sample, per-channel add (Dense_0 output incl. its bias) or null
    const float* resid;           // identity residual [n][HWo][Cout] or null
    float* out;                   // [n][HWo][Cout]
    const int* tab;               // [Mpad][ntap+1]: LDS row of (row m, tap); last column: shortcut row
    int NB;                       // samples in this call
    int srcA_mod, scA_mod;        // >0: sample index of A is n % mod (CFG batch reads x twice)
    int CA, CB, Cv;               // real channels of A and B; Cv = padded total (multiple of 16)
    int HWa, HWv, HWo;            // pixels per sample: source A, virtual input, output
    int ntap;
    int G;                        // GroupNorm groups (0: no GN/SiLU)
    int CscA, CscB, Csc, HWsa;    // shortcut channels / source-A pixels
    int Cout, Cout_pad;
    int dense_stride, dense_off;
    float out_scale, eps;
    int S, BN, Mpad;
    float drop_p; uint64_t drop_seed; uint32_t op_id;   // train mode: dropout after GN+SiLU (Dropout_0, RD/models/layerspp.py:204)
    const unsigned long long* seed_dev;   // non-null: the step's dropout seed is read from device memory (a recorded launch graph is replayed with a new seed)
    int bf16;                     // 1: wpk / wsc are the bf16 copies [tap][C/32][Cout_pad][32] and the MFMAs take bf16 operands (fp32 accumulate);
                                  //    needs Cv % 32 == 0 and Csc % 32 == 0 (training with compute_dtype = bf16, BASELINE config #4)
    int a_bf16, b_bf16, o_bf16;   // element type of srcA / scA, of srcB / scB / resid, and of out: 0 fp32, 1 bf16 (train_dtype = bf16 keeps the
                                  // activation workspace and the backward scratch tensors in bf16; the caller's x / out stay fp32)
    int dbg;                      // unused (kept so the argument block layout of recorded launches stays stable)
};

__host__ __device__ inline size_t conv_lds_bytes(const ConvArgs& a) {
    size_t f = (size_t)(a.S * a.HWv + 1) * (a.Cv + 4);
    if (a.Csc) f += (size_t)(a.S * a.HWo + 1) * (a.Csc + 4);
    f += (size_t)((2 * a.S * a.G + 3) & ~3);
    f += (size_t)a.Mpad * (a.ntap + 1);
    return f * 4;
}

// gather S samples of a [n][HWsrc][CA] (+ [n][HWdst][CB]) pair into LDS rows [s*HWdst + v][C + 4].
// Work-item i handles float4 #i of the tile; (pixel, channel) are advanced incrementally (no divisions in the loop).
__device__ __forceinline__ void conv_stage(float* __restrict__ L, int rs, const float* __restrict__ A,
                                           const float* __restrict__ B, const int* __restrict__ map, int CA,
                                           int CB, int Cpad, int HWsrc, int HWdst, int S, int n0, int NB,
                                           int a_mod, int tid, int a_bf = 0, int b_bf = 0, int lds_bf16 = 0) {
    // lds_bf16: the row keeps its byte stride (rs floats) but holds bf16 values in its first Cpad * 2 bytes -- the layout the bf16
    // GEMM reads (a lane's A fragment of v_mfma_f32_16x16x32_bf16 is then ONE ds_read_b128 and no conversion)
    const int c4n = Cpad >> 2;
    const int total = S * HWdst * c4n;
    const int dpv = RDMI_THREADS / c4n, dc4 = RDMI_THREADS - dpv * c4n;
    int pv = tid / c4n, c4 = tid - pv * c4n;
    for (int i = tid; i < total; i += RDMI_THREADS) {
        const int c = c4 << 2;
        int s = 0, v = pv;
        while (v >= HWdst) { v -= HWdst; ++s; }
        const int n = n0 + s;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (n < NB) {
            if (c < CA) {
                const int nA = a_mod > 0 ? n % a_mod : n;
                const size_t ia = ((size_t)nA * HWsrc + (map ? map[v] : v)) * CA + c;
                if ((CA & 3) == 0) {
                    val = ldact4(A, ia, a_bf);
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (c + j < CA) val[j] = ldact1(A, ia + j, a_bf);
                }
            } else if (c < CA + CB) {
                val = ldact4(B, ((size_t)n * HWdst + v) * CB + (c - CA), b_bf);
            }
        }
        if (lds_bf16) {
            typedef unsigned int u32x2 __attribute__((vector_size(8)));
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(L + (size_t)pv * rs) + c) = u32x2{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
        } else {
            *reinterpret_cast<f32x4*>(L + (size_t)pv * rs + c) = val;
        }
        pv += dpv; c4 += dc4;
        if (c4 >= c4n) { c4 -= c4n; ++pv; }
    }
    for (int i = tid; i < rs; i += RDMI_THREADS) L[(size_t)S * HWdst * rs + i] = 0.f;   // the zero row (all-zero bits in either layout)
}

// The GEMM + epilogue for a wave that owns NMT row tiles (compile-time) x NT column tiles.
// B fragments (weights) stream from L2 through a PF-deep register ring; A fragments come from LDS.
template <int WM, int WN, int WK, int NMT, int NT, int PF, bool BF16 = false>
__device__ __forceinline__ void conv_gemm(const ConvArgs& a, const float* __restrict__ X, const float* __restrict__ XS,
                                          const int* __restrict__ tabL, int rs, int rss, int wm, int wn, int wk,
                                          int lane, int wave, int n0, int co0) {
    const int lrow = lane & 15, kq = lane >> 4;
    const int tw = a.ntap + 1;
    const int colbase = co0 + wn * NT * 16;
    f32x4 acc[NMT > 0 ? NMT : 1][NT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (NMT > 0) {
        int trow[NMT > 0 ? NMT : 1];           // table row of each of this lane's m-tiles
#pragma unroll
        for (int i = 0; i < NMT; ++i) trow[i] = ((wm + i * WM) * 16 + lrow) * tw;

        if (BF16) {
            // bf16 operands (v_mfma_f32_16x16x32_bf16): a step is 32 channels of one tap; the LDS rows already hold bf16 (row byte
            // stride unchanged: 2 * rs bf16 elements), so the A fragment is one ds_read_b128; the B fragment 16 bytes of the bf16 weights
            const bf16_t* X16 = reinterpret_cast<const bf16_t*>(X);
            const bf16_t* XS16 = reinterpret_cast<const bf16_t*>(XS);
            const int rs16 = 2 * rs, rss16 = 2 * rss;
            {
                const int nch = a.Cv >> 5;
                const int nsteps = a.ntap * nch;
                const size_t bstride = (size_t)a.Cout_pad * 32;
                const bf16_t* Wl = reinterpret_cast<const bf16_t*>(a.wpk) + (size_t)(colbase + lrow) * 32 + kq * 8;
                u32x4 bring[PF][NT];
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    if (wk + u * WK < nsteps) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) bring[u][t] = *reinterpret_cast<const u32x4*>(Wl + (size_t)(wk + u * WK) * bstride + t * 512);
                    }
                int ph = wk / nch, ch = wk - ph * nch;
                int abase[NMT > 0 ? NMT : 1];
#pragma unroll
                for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + ph] * rs16 + kq * 8;
                for (int q = wk; q < nsteps; q += PF * WK) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const int qq = q + u * WK;
                        if (qq < nsteps) {
#pragma unroll
                            for (int i = 0; i < NMT; ++i) {
                                const u32x4 af = *reinterpret_cast<const u32x4*>(X16 + abase[i] + ch * 32);
#pragma unroll
                                for (int t = 0; t < NT; ++t) acc[i][t] = mfma16_bf16(af, bring[u][t], acc[i][t]);
                            }
                            if (qq + PF * WK < nsteps) {
#pragma unroll
                                for (int t = 0; t < NT; ++t)
                                    bring[u][t] = *reinterpret_cast<const u32x4*>(Wl + (size_t)(qq + PF * WK) * bstride + t * 512);
                            }
                            ch += WK;
                            if (ch >= nch) {
                                ch -= nch; ++ph;
                                if (ph < a.ntap) {
#pragma unroll
                                    for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + ph] * rs16 + kq * 8;
                                }
                            }
                        }
                    }
                }
            }
            if (a.Csc) {
                const int nch = a.Csc >> 5;
                const size_t bstride = (size_t)a.Cout_pad * 32;
                const bf16_t* Wl = reinterpret_cast<const bf16_t*>(a.wsc) + (size_t)(colbase + lrow) * 32 + kq * 8;
                int abase[NMT > 0 ? NMT : 1];
#pragma unroll
                for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + a.ntap] * rss16 + kq * 8;
                for (int ch = wk; ch < nch; ch += WK) {
                    u32x4 bf[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const u32x4*>(Wl + (size_t)ch * bstride + t * 512);
#pragma unroll
                    for (int i = 0; i < NMT; ++i) {
                        const u32x4 af = *reinterpret_cast<const u32x4*>(XS16 + abase[i] + ch * 32);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[i][t] = mfma16_bf16(af, bf[t], acc[i][t]);
                    }
                }
            }
        } else {
        // ---- conv phases: flattened step q = tap * nch + chunk; B for step q sits at wpk + q * bstride
        {
            const int nch = a.Cv >> 4;
            const int nsteps = a.ntap * nch;
            const size_t bstride = (size_t)a.Cout_pad * 16;
            const float* Wl = a.wpk + (size_t)(colbase + lrow) * 16 + kq * 4;
            f32x4 bring[PF][NT];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (wk + u * WK < nsteps) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) bring[u][t] = *reinterpret_cast<const f32x4*>(Wl + (size_t)(wk + u * WK) * bstride + t * 256);
                }
            int ph = wk / nch, ch = wk - ph * nch;
            int abase[NMT > 0 ? NMT : 1];
#pragma unroll
            for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + ph] * rs + kq * 4;
            for (int q = wk; q < nsteps; q += PF * WK) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int qq = q + u * WK;
                    if (qq < nsteps) {
                        f32x4 af[NMT > 0 ? NMT : 1];
#pragma unroll
                        for (int i = 0; i < NMT; ++i) af[i] = *reinterpret_cast<const f32x4*>(X + abase[i] + ch * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int i = 0; i < NMT; ++i)
#pragma unroll
                                for (int t = 0; t < NT; ++t) acc[i][t] = mfma16(af[i][j], bring[u][t][j], acc[i][t]);
                        if (qq + PF * WK < nsteps) {
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                bring[u][t] = *reinterpret_cast<const f32x4*>(Wl + (size_t)(qq + PF * WK) * bstride + t * 256);
                        }
                        ch += WK;
                        if (ch >= nch) {          // next tap: new row offsets (wave-uniform branch)
                            ch -= nch; ++ph;
                            if (ph < a.ntap) {
#pragma unroll
                                for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + ph] * rs + kq * 4;
                            }
                        }
                    }
                }
            }
        }
        // ---- NIN shortcut phase over the raw block input
        if (a.Csc) {
            const int nch = a.Csc >> 4;
            const size_t bstride = (size_t)a.Cout_pad * 16;
            const float* Wl = a.wsc + (size_t)(colbase + lrow) * 16 + kq * 4;
            int abase[NMT > 0 ? NMT : 1];
#pragma unroll
            for (int i = 0; i < NMT; ++i) abase[i] = tabL[trow[i] + a.ntap] * rss + kq * 4;
            f32x4 bcur[NT], bnext[NT];
            if (wk < nch) {
#pragma unroll
                for (int t = 0; t < NT; ++t) bcur[t] = *reinterpret_cast<const f32x4*>(Wl + (size_t)wk * bstride + t * 256);
            }
            for (int ch = wk; ch < nch; ch += WK) {
                if (ch + WK < nch) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) bnext[t] = *reinterpret_cast<const f32x4*>(Wl + (size_t)(ch + WK) * bstride + t * 256);
                }
                f32x4 af[NMT > 0 ? NMT : 1];
#pragma unroll
                for (int i = 0; i < NMT; ++i) af[i] = *reinterpret_cast<const f32x4*>(XS + abase[i] + ch * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NMT; ++i)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[i][t] = mfma16(af[i][j], bcur[t][j], acc[i][t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) bcur[t] = bnext[t];
            }
        }
        }   // fp32 / bf16 operand paths
    }

    // ---- split-K across waves: reduce through LDS (the input tile is dead by now)
    if (WK > 1) {
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(rdmi_lds);
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < NMT; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) red[(((wave * (NMT > 0 ? NMT : 1)) + i) * NT + t) * 64 + lane] = acc[i][t];
        }
        __syncthreads();
        if (wk == 0) {
            for (int k = 1; k < WK; ++k) {
                const int ow = wave + k;    // waves with the same (wm, wn) are consecutive in wk
#pragma unroll
                for (int i = 0; i < NMT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[i][t] += red[(((ow * (NMT > 0 ? NMT : 1)) + i) * NT + t) * 64 + lane];
            }
        }
    }

    // ---- epilogue: bias (+ NIN bias) (+ Dense_0(temb)) (+ identity residual), scale, store NHWC
    if (wk == 0) {
        const int rows = a.S * a.HWo;
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
            const int mt = wm + i * WM;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = colbase + t * 16 + lrow;
                if (col >= a.Cout) continue;
                float add = a.bias[col];
                if (a.bias_sc) add += a.bias_sc[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kq * 4 + r;
                    if (row >= rows) continue;
                    int s = 0, rr = row;
                    while (rr >= a.HWo) { rr -= a.HWo; ++s; }
                    const int n = n0 + s;
                    if (n >= a.NB) continue;
                    const size_t o = ((size_t)n0 * a.HWo + row) * a.Cout + col;
                    float v = acc[i][t][r] + add;
                    if (a.dense) v += a.dense[(size_t)n * a.dense_stride + a.dense_off + col];
                    if (a.resid) v += ldact1(a.resid, o, a.b_bf16);
                    stact1(a.out, o, v * a.out_scale, a.o_bf16);
                }
            }
        }
    }
}

template <int WM, int WN, int WK, int MT, int NT, int PF, bool BF16 = false>
__global__ __launch_bounds__(RDMI_THREADS) void conv_mfma_kernel(ConvArgs a) {
    static_assert(WM * WN * WK == 4, "four waves per workgroup");
    static_assert(WK == 1 || MT == 1, "split-K waves all own the same single row tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> scalar registers, scalar branches
    const int n0 = blockIdx.x * a.S;
    const int co0 = blockIdx.y * a.BN;

    float* X = reinterpret_cast<float*>(rdmi_lds);
    const int rs = a.Cv + 4;
    const int rowsX = a.S * a.HWv + 1;
    float* XS = X + (size_t)rowsX * rs;
    const int rss = a.Csc + 4;
    const int rowsS = a.Csc ? a.S * a.HWo + 1 : 0;
    float* stat = XS + (size_t)rowsS * rss;
    int* tabL = reinterpret_cast<int*>(stat + ((2 * a.S * a.G + 3) & ~3));
    const int tw = a.ntap + 1;

    // ---- stage 1: gather inputs, copy the tap table
    // bf16 GEMM: inputs that need no GroupNorm are staged as bf16 straight away; a GroupNorm input is staged fp32 (statistics) and
    // narrowed in place by the normalisation pass below
    conv_stage(X, rs, a.srcA, a.srcB, a.mapA, a.CA, a.CB, a.Cv, a.HWa, a.HWv, a.S, n0, a.NB, a.srcA_mod, tid, a.a_bf16, a.b_bf16, BF16 && a.G == 0);
    if (a.Csc)
        conv_stage(XS, rss, a.scA, a.scB, a.mapSc, a.CscA, a.CscB, a.Csc, a.HWsa, a.HWo, a.S, n0, a.NB, a.scA_mod, tid, a.a_bf16, a.b_bf16, BF16);
    for (int i = tid; i < a.Mpad * tw; i += RDMI_THREADS) tabL[i] = a.tab[i];

    // ---- stage 2: GroupNorm statistics (two-pass, in LDS) + affine + SiLU, in place
    if (a.G > 0) {
        __syncthreads();
        const int G = a.G, Cg = a.Cv / G;
        const int pairs = a.S * G;                 // power of two, <= 256 (host-checked)
        const int T = RDMI_THREADS / pairs;        // lanes cooperating on one (sample, group)
        const int pair = tid / T, sub = tid - pair * T;
        const int s = pair / G, g = pair - s * G;
        const float inv_cnt = 1.0f / (float)(Cg * a.HWv);
        const float* base = X + (size_t)s * a.HWv * rs + g * Cg;
        float sum = 0.f;
        for (int v = sub; v < a.HWv; v += T)
            for (int cc = 0; cc < Cg; ++cc) sum += base[(size_t)v * rs + cc];
        for (int m = T >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
        const float mean = sum * inv_cnt;
        float sq = 0.f;
        for (int v = sub; v < a.HWv; v += T)
            for (int cc = 0; cc < Cg; ++cc) {
                const float d = base[(size_t)v * rs + cc] - mean;
                sq += d * d;
            }
        for (int m = T >> 1; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
        if (sub == 0) {
            stat[2 * pair] = mean;
            stat[2 * pair + 1] = 1.0f / sqrtf(sq * inv_cnt + a.eps);
        }
        __syncthreads();
        if (BF16) {
            // normalise + narrow IN PLACE: row p keeps its byte stride but ends up holding bf16 in its first Cv * 2 bytes.  A wave owns
            // whole rows (64 / c4n of them per pass when c4n divides 64, else one), so no other wave touches a row it rewrites.
            typedef unsigned int u32x2 __attribute__((vector_size(8)));
            const int c4n = a.Cv >> 2;
            const int rpw = (64 % c4n) == 0 ? 64 / c4n : 1;                 // rows per wave pass
            const int rows = a.S * a.HWv;
            const int wv = tid >> 6, ln = tid & 63;
            for (int r0 = wv * rpw; r0 < rows; r0 += 4 * rpw) {
                for (int cb = 0; cb < (rpw > 1 ? 1 : (c4n + 63) / 64); ++cb) {
                    const int rl = rpw > 1 ? ln / c4n : 0, c4 = rpw > 1 ? ln - rl * c4n : cb * 64 + ln;
                    const int pv = r0 + rl;
                    const bool on = pv < rows && c4 < c4n;
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    const int c = c4 << 2;
                    if (on) {
                        const int ss = pv / a.HWv, v = pv - ss * a.HWv;
                        val = *reinterpret_cast<const f32x4*>(X + (size_t)pv * rs + c);
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c), bt = *reinterpret_cast<const f32x4*>(a.beta + c);
                        int gg = c / Cg, left = Cg - (c - gg * Cg);
                        for (int j = 0; j < 4; ++j) {
                            if (left == 0) { ++gg; left = Cg; }
                            const float mu = stat[2 * (ss * G + gg)], rstd = stat[2 * (ss * G + gg) + 1];
                            val[j] = silu_f((val[j] - mu) * rstd * gm[j] + bt[j]);
                            if (a.drop_p > 0.f) val[j] *= dropout_scale(a.seed_dev ? (uint64_t)*a.seed_dev : a.drop_seed, a.op_id, ((uint64_t)(n0 + ss) * a.HWv + v) * a.Cv + c + j, a.drop_p);
                            --left;
                        }
                    }
                    // every lane's fp32 read above precedes every lane's bf16 write below (a wave's LDS instructions execute in issue
                    // order): the bf16 of channel quad c4 lands on bytes [8 c4, 8 c4 + 8) = fp32 quad c4 / 2 of the same row
                    wave_order_point();
                    if (on) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(X + (size_t)pv * rs) + c) = u32x2{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};
                }
            }
        } else {
        // normalise: a work-item keeps its float4 channel column (fixed gamma/beta/group) and walks pixels
        const int c4n = a.Cv >> 2;
        const int total = a.S * a.HWv * c4n;
        const int dpv = RDMI_THREADS / c4n, dc4 = RDMI_THREADS - dpv * c4n;
        int pv = tid / c4n, c4 = tid - pv * c4n;
        for (int i = tid; i < total; i += RDMI_THREADS) {
            const int c = c4 << 2;
            int ss = 0, v = pv;
            while (v >= a.HWv) { v -= a.HWv; ++ss; }
            float* p = X + (size_t)pv * rs + c;
            f32x4 val = *reinterpret_cast<f32x4*>(p);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c), bt = *reinterpret_cast<const f32x4*>(a.beta + c);
            int gg = c / Cg, left = Cg - (c - gg * Cg);       // channels left in the current group
            for (int j = 0; j < 4; ++j) {
                if (left == 0) { ++gg; left = Cg; }
                const float mu = stat[2 * (ss * G + gg)], rstd = stat[2 * (ss * G + gg) + 1];
                val[j] = silu_f((val[j] - mu) * rstd * gm[j] + bt[j]);
                if (a.drop_p > 0.f) val[j] *= dropout_scale(a.seed_dev ? (uint64_t)*a.seed_dev : a.drop_seed, a.op_id, ((uint64_t)(n0 + ss) * a.HWv + v) * a.Cv + c + j, a.drop_p);
                --left;
            }
            *reinterpret_cast<f32x4*>(p) = val;
            pv += dpv; c4 += dc4;
            if (c4 >= c4n) { c4 -= c4n; ++pv; }
        }
    }
        }
    __syncthreads();

    // ---- stage 3/4: GEMM + epilogue, specialised on this wave's (uniform) number of row tiles
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int mtiles = a.Mpad >> 4;
    const int nmt = mtiles > wm ? (mtiles - wm + WM - 1) / WM : 0;
#define RDMI_CASE(K)                                                                                         \
    case K:                                                                                                  \
        if (K <= MT) conv_gemm<WM, WN, WK, (K <= MT ? K : 0), NT, PF, BF16>(a, X, XS, tabL, rs, rss, wm, wn, wk, lane, wave, n0, co0); \
        break;
    switch (nmt) {
        RDMI_CASE(0) RDMI_CASE(1) RDMI_CASE(2) RDMI_CASE(3) RDMI_CASE(4) RDMI_CASE(5) RDMI_CASE(6)
        default: break;
    }
#undef RDMI_CASE
}
