// Spatially TILED plan for NCSN++ shapes whose samples do not fit one workgroup (BASELINE config #5: the CIFAR-shape model,
// 32x32 / 16x16 / 8x8 / 4x4 with 128-256 channels, attention over L = 256 positions; RD/configs/model/ddpmpp.yaml,
// RD/models/ncsnpp.py:226-354, RD/models/layerspp.py:67-96,171-214).  Activations live in HBM as NHWC fp32
// [n][pixel][C]; every layer is a launch (the 9x9 GTO-Halo model keeps its workgroup-resident kernel).
//
//   tconv_kernel   3x3 / 1x1 convolution as an implicit GEMM on the exact-fp32 MFMA (or bf16 operands).  A workgroup owns a tile of
//                  whole output image rows (64 output pixels) x 64..256 output channels.  Per 32-channel slab of the input it stages the tile's
//                  VIRTUAL input window -- after concat, nearest x2 upsampling, zero padding, and GroupNorm + SiLU applied on
//                  the fly from precomputed per-(sample, group) statistics -- in LDS ([pixel][32+4] fp32: conflict-free
//                  ds_read_b128 A fragments); the 9 taps are row offsets into that window.  Weights use the same packed layout
//                  as everywhere else ([tap][Cin/16][Cout_pad][16]: a wave's B fragment is 1 KiB contiguous).
//                  Epilogue: bias, Dense_0(temb) column add, residual, 1/sqrt2, optional 1/sigma (scale_by_sigma).
//   gn_stats_kernel  two-pass mean / rstd per (sample, group) over a (possibly concatenated) tensor.
//   bgemm_nt_kernel  batched C = alpha * A B^T on the fp32 MFMA (attention scores Q K^T and P V with V transposed).
//   softmax_rows_kernel, transpose_lc_kernel, nchw/nhwc copies: the rest of AttnBlockpp and the API boundary.
#pragma once
#include "common.h"

struct TConvArgs {
    const float* srcA; const float* srcB;   // virtual input channels = concat(A [n][Ha*Wa][CA], B [n][Ha*Wa][CB]); B may be null
    int CA, CB, Cv;                         // Cv = CA + CB padded to a multiple of 32 (channels beyond CA + CB read as zero)
    int Ha, Wa;                             // source grid
    int up;                                 // 1: nearest x2 upsampling of the source (F.interpolate(scale_factor=2), layerspp.py:122)
    int Hv, Wv;                             // virtual input grid (after upsampling)
    int stride, pad_lo, ntap;               // 3x3: (1, 1, 9) or Downsample's (2, 0, 9) with the implicit bottom/right zero; 1x1: (1, 0, 1)
    int Ho, Wo, TR;                         // output grid; output image rows per tile (TR * Wo = 64, or the whole 4x4 image)
    const float* stats; int G, Cg;          // GroupNorm: [n][G][2] (mean, rstd) or null; Cg channels per group
    const float* gamma; const float* beta; int act;
    const float* wpk;                       // [ntap][Cv/16][Cout_pad][16]
    const float* bias;
    const float* dense; int dense_stride, dense_off;   // null or [n][dense_stride]: + dense[n][dense_off + col]
    const float* resid;                     // null or [n][Ho*Wo][Cout]
    float out_scale;
    const float* sig; int sig_is_time, sig_mod; float smin, ratio;   // scale_by_sigma: out /= sigma[n % sig_mod] (null: off)
    float* out; int Cout, Cout_pad;
    float* chsum;                           // null or [n][tiles per image][Cout][2]: per-channel (sum, squared deviations about the tile mean) of this tile's outputs,
                                            // from which the consumer's GroupNorm statistics are formed (gn_finalize_kernel): no extra pass over the tensor
    int NB;
    const void* zeros;                      // 256 zero bytes (iconv_kernel: source of window pixels outside the image)
    int col_il;                             // bf16 packs: column interleave factor F of the weights (PackJob::col_il; 0 / 1: none)
    int out_bf16;                           // 1: `out` is a bf16 [n][Ho*Wo][Cout] tensor (the q | k | v projection feeding flash_attn_bf16_kernel, which rounds to bf16 anyway)
};

__host__ __device__ inline int tconv_trv(const TConvArgs& a) { return a.ntap == 1 ? a.TR : (a.TR - 1) * a.stride + 3; }
__host__ __device__ inline int tconv_wl(const TConvArgs& a) { return a.ntap == 1 ? a.Wo : (a.Wo - 1) * a.stride + 3; }
// LDS row strides of the staged window.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS);
// with lane = 16 kq + row the group's 16 fragments fall on distinct 16-byte bank columns when the row stride is 10 (or 6) units of 16 bytes --
// the strides used before (9 units: 36 floats / 72 bf16, 5 units: 40 bf16) put two fragments on one column (SQ_LDS_BANK_CONFLICT = 46 % of the
// LDS cycles of tconv_pre_kernel, profiles/r03_pmc_cifar_b64_bf16.txt).
#ifndef RDMI_TC_ROW_F32
#define RDMI_TC_ROW_F32 36       // the fp32 conv measured no faster at 40 floats (10 units): left at 36
#endif
#ifndef RDMI_TC_ROW_BF16
#define RDMI_TC_ROW_BF16 48      // 6 units of 16 bytes
#endif
__host__ __device__ inline size_t tconv_lds_bytes(const TConvArgs& a) { return (size_t)tconv_trv(a) * tconv_wl(a) * RDMI_TC_ROW_F32 * 4; }

// NMT row tiles of 16 output pixels per wave (4: a 64-pixel tile; 1: the whole 4x4 image) x NCT adjacent 16-column tiles per wave:
// a workgroup covers 64 * NCT output channels, so the staged window (and its GroupNorm + SiLU arithmetic) is shared by up to 256
// output channels and every A fragment read from LDS feeds NCT MFMAs.
// BF16 (BASELINE config #5 asks bf16 + MFMA): same tiling and staging, but the activated window is kept in LDS as bf16
// ([pixel][32 + 16] bf16: one ds_read_b128 is a lane's whole A fragment of v_mfma_f32_16x16x32_bf16) and the weights come from the
// bf16 copy packed [tap][Cin/32][Cout_pad][32]; accumulation, GroupNorm arithmetic, epilogue and the tensors in HBM stay fp32.
// One MFMA per (tap, row tile, column tile, 32-channel slab) replaces eight fp32 ones at 1/16 of their cycles.
__host__ __device__ inline size_t tconv_bf16_lds_bytes(const TConvArgs& a) { return (size_t)tconv_trv(a) * tconv_wl(a) * RDMI_TC_ROW_BF16 * 2; }

// Staging is split in two so that the global loads of the NEXT 32-channel slab are in flight while the MFMAs of the current one
// run: tconv_fetch issues this work-item's (up to TC_MAXS) 16-byte loads of a slab into registers -- the per-element window
// geometry (source pixel or "outside") is slab-independent and computed once -- and tconv_commit applies GroupNorm + SiLU from
// the per-(sample, group) statistics, converts (bf16 plan) and writes the LDS window.
#define TC_MAXS 10                      // ceil(297 window pixels * 8 float4 / 256 work-items)
struct TcGeom { long sp[TC_MAXS]; };    // source pixel index of staged element k (-1: outside the image or beyond the window)

__device__ __forceinline__ void tconv_geom(const TConvArgs& a, int n, int vy0, int vx0, int Wl, int npix, int tid, TcGeom& g) {
#pragma unroll
    for (int k = 0; k < TC_MAXS; ++k) {
        const int i = tid + k * RDMI_THREADS;
        const int p = i >> 3;
        const int ry = p / Wl, rx = p - ry * Wl;
        const int vy = vy0 + ry, vx = vx0 + rx;
        long sp = -1;
        if (i < npix * 8 && vy >= 0 && vy < a.Hv && vx >= 0 && vx < a.Wv) {
            const int sy = a.up ? (vy >> 1) : vy, sx = a.up ? (vx >> 1) : vx;
            sp = (long)n * a.Ha * a.Wa + (long)sy * a.Wa + sx;
        }
        g.sp[k] = sp;
    }
}

__device__ __forceinline__ void tconv_fetch(const TConvArgs& a, const TcGeom& g, int c0, int tid, f32x4 (&raw)[TC_MAXS]) {
    const int Cin = a.CA + a.CB;
    const int c = c0 + (tid & 7) * 4;                       // work-item's channel quad (i & 7 == tid & 7 for every k)
#pragma unroll
    for (int k = 0; k < TC_MAXS; ++k) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (g.sp[k] >= 0 && c < Cin) {
            if (c < a.CA) {
                const float* ptr = a.srcA + (size_t)g.sp[k] * a.CA + c;
                if ((a.CA & 3) == 0) v = *reinterpret_cast<const f32x4*>(ptr);
                else
                    for (int j = 0; j < 4; ++j)
                        if (c + j < a.CA) v[j] = ptr[j];
            } else {
                v = *reinterpret_cast<const f32x4*>(a.srcB + (size_t)g.sp[k] * a.CB + (c - a.CA));
            }
        }
        raw[k] = v;
    }
}

template <bool BF16>
__device__ __forceinline__ void tconv_commit(const TConvArgs& a, const TcGeom& g, int n, int c0, int npix, int tid, const f32x4 (&raw)[TC_MAXS]) {
    const int Cin = a.CA + a.CB;
    const int q = tid & 7, c = c0 + q * 4;
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, rstd = {1.f, 1.f, 1.f, 1.f};      // per channel of the quad: a group of 6 channels (C = 192) straddles quads
    f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
    const bool gn = a.stats != nullptr && c < Cin;
    if (gn) {
        if ((a.Cg & 3) == 0) {                                   // the usual case: one group per quad
            const int grp = c / a.Cg;
            const float m0 = a.stats[((size_t)n * a.G + grp) * 2], r0 = a.stats[((size_t)n * a.G + grp) * 2 + 1];
            mean = f32x4{m0, m0, m0, m0}; rstd = f32x4{r0, r0, r0, r0};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int grp = min(c + j, Cin - 1) / a.Cg;
                mean[j] = a.stats[((size_t)n * a.G + grp) * 2]; rstd[j] = a.stats[((size_t)n * a.G + grp) * 2 + 1];
            }
        }
        gm = *reinterpret_cast<const f32x4*>(a.gamma + c); bt = *reinterpret_cast<const f32x4*>(a.beta + c);
    }
#pragma unroll
    for (int k = 0; k < TC_MAXS; ++k) {
        const int i = tid + k * RDMI_THREADS;
        if (i >= npix * 8) break;
        const int p = i >> 3;
        f32x4 v = raw[k];
        if (gn && g.sp[k] >= 0) {                           // padding stays zero: the conv pads the ACTIVATED tensor
            for (int j = 0; j < 4; ++j) {
                const float y = (v[j] - mean[j]) * (rstd[j] * gm[j]) + bt[j];
                v[j] = a.act ? silu_f(y) : y;
            }
        }
        if (BF16) {
            typedef unsigned int u32x2 __attribute__((vector_size(8)));
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(rdmi_lds) + (size_t)p * RDMI_TC_ROW_BF16 + q * 4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(rdmi_lds) + (size_t)p * RDMI_TC_ROW_F32 + q * 4) = v;
        }
    }
}

// Shared epilogue of the tiled convs: bias, Dense_0(temb) column add, residual, 1/sqrt2 (and 1/sigma), store, per-tile channel sums.
// Per column tile the NMT * 4 residual loads are issued together BEFORE any store (with one load -> wait -> store chain per element
// a workgroup spent 32 exposed memory latencies here: the ISA showed s_waitcnt vmcnt(0) after every load); invalid elements
// (columns beyond Cout, rows beyond the tile / image) read offset 0 and are masked at the store.
template <int N> struct alignas(4 * N) FVec { float v[N]; };      // N adjacent floats moved as one 8- / 16-byte access
// Column interleave (bf16 packs, TConvArgs::col_il = F): the weights' columns are permuted within groups of 16 F so that the F adjacent
// MFMA column tiles of a group give lane l the F ADJACENT output columns 16F g + F l .. + F - 1.  With NCT == F a lane's NCT accumulators
// of one row are therefore one 8- or 16-byte vector: residual loads and output stores are dwordx2 / dwordx4, a wave instruction covers
// 4 rows x 128 / 256 contiguous bytes (full lines) instead of 4 x 64 B, and there are NCT times fewer of them.  The arithmetic per
// element and the per-column reductions are those of the scalar form (same results bit for bit).
template <int NMT, int NCT>
__device__ __forceinline__ void tconv_epilogue(const TConvArgs& a, f32x4 (&acc)[NMT][NCT], int n, int tile, int tiles_per_img, int oy0, int col0, int kq) {
    float sdiv = 1.f;
    if (a.sig) {
        const float sv = a.sig[a.sig_mod > 0 ? n % a.sig_mod : n];
        sdiv = a.sig_is_time ? a.smin * powf(a.ratio, sv) : sv;
    }
    const float scale = a.out_scale / sdiv;
    const int HWo = a.Ho * a.Wo, tile_px = a.TR * a.Wo;
    const size_t nbase = (size_t)n * HWo * a.Cout;             // wave-uniform sample base; element offsets below fit 32 bits
    const float* rbase = a.resid ? a.resid + nbase : nullptr;
    float* obase = a.out + nbase;
    const int F = a.col_il > 1 ? a.col_il : 1;
    if constexpr (NCT > 1) if (F == NCT) {
        typedef FVec<NCT> vecF;
        const int lrow = col0 & 15;
        const int colb = ((col0 >> 4) / NCT) * (16 * NCT) + lrow * NCT;          // the lane's NCT adjacent columns (Cout % 16 NCT == 0: all or none exist)
        const bool colok = colb < a.Cout;
        float add[NCT];
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) {
            add[cc] = (colok && a.bias) ? a.bias[colb + cc] : 0.f;
            if (colok && a.dense) add[cc] += a.dense[(size_t)n * a.dense_stride + a.dense_off + colb + cc];
        }
        unsigned off[NMT][4]; bool ok[NMT][4];
#pragma unroll
        for (int i = 0; i < NMT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = i * 16 + kq * 4 + r, opix = oy0 * a.Wo + m;
                ok[i][r] = colok && m < tile_px && opix < HWo;
                off[i][r] = ok[i][r] ? (unsigned)(opix * a.Cout + colb) : 0u;
            }
        // every residual vector of the wave's tile is requested before the first store (one memory round trip, as in the scalar form;
        // two halves of 32 registers each let the NCT = 4 kernel hold a third wave per SIMD and measured no faster)
        constexpr int IH = NMT;
#pragma unroll
        for (int i0 = 0; i0 < NMT; i0 += IH) {
            vecF rv[IH][4];
#pragma unroll
            for (int i = 0; i < IH; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (rbase) rv[i][r] = *reinterpret_cast<const vecF*>(rbase + off[i0 + i][r]);
                    else
#pragma unroll
                        for (int cc = 0; cc < NCT; ++cc) rv[i][r].v[cc] = 0.f;
                }
#pragma unroll
            for (int ii = 0; ii < IH; ++ii) {
                const int i = i0 + ii;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vecF v;
#pragma unroll
                    for (int cc = 0; cc < NCT; ++cc) { v.v[cc] = (acc[i][cc][r] + add[cc] + rv[ii][r].v[cc]) * scale; acc[i][cc][r] = v.v[cc]; }
                    if (ok[i][r]) {
                        if (a.out_bf16) {
                            bf16_t* o16 = reinterpret_cast<bf16_t*>(a.out) + nbase + off[i][r];
                            if constexpr (NCT == 4) { typedef unsigned int u32x2 __attribute__((vector_size(8))); *reinterpret_cast<u32x2*>(o16) = u32x2{pack_bf16x2(v.v[0], v.v[1]), pack_bf16x2(v.v[2], v.v[3])}; }
                            else *reinterpret_cast<unsigned*>(o16) = pack_bf16x2(v.v[0], v.v[1]);
                        } else *reinterpret_cast<vecF*>(obase + off[i][r]) = v;
                    }
                }
            }
        }
        if (a.chsum) {
            const int cnt = min(tile_px, HWo - oy0 * a.Wo);                  // valid pixels of this tile (wave-uniform)
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < NMT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (ok[i][r]) s1 += acc[i][cc][r];
                s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                const float mt = s1 / (float)cnt;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NMT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = acc[i][cc][r] - mt; if (ok[i][r]) m2 += d * d; }
                m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
                if (kq == 0 && colok) {
                    float* cs = a.chsum + (((size_t)n * tiles_per_img + tile) * a.Cout + colb + cc) * 2;
                    cs[0] = s1; cs[1] = m2;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int cc = 0; cc < NCT; ++cc) {
        const int pos = col0 + cc * 16;                                        // position in the packed weights -> output column
        const int col = ((pos >> 4) / F) * (16 * F) + (pos & 15) * F + (pos >> 4) % F;
        const bool colok = col < a.Cout;
        float add = (colok && a.bias) ? a.bias[col] : 0.f;
        if (colok && a.dense) add += a.dense[(size_t)n * a.dense_stride + a.dense_off + col];
        unsigned off[NMT][4]; bool ok[NMT][4]; float rv[NMT][4];
#pragma unroll
        for (int i = 0; i < NMT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = i * 16 + kq * 4 + r, opix = oy0 * a.Wo + m;
                ok[i][r] = colok && m < tile_px && opix < HWo;
                off[i][r] = ok[i][r] ? (unsigned)(opix * a.Cout + col) : 0u;
                rv[i][r] = 0.f;
            }
        if (rbase) {
#pragma unroll
            for (int i = 0; i < NMT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[i][r] = rbase[off[i][r]];
        }
        float s1 = 0.f;
        float vals[NMT][4];
#pragma unroll
        for (int i = 0; i < NMT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (acc[i][cc][r] + add + rv[i][r]) * scale;
                vals[i][r] = v;
                if (ok[i][r]) { if (a.out_bf16) reinterpret_cast<bf16_t*>(a.out)[nbase + off[i][r]] = f2bf(v); else obase[off[i][r]] = v; s1 += v; }
            }
        if (a.chsum) {
            // per (sample, tile, column): the sum of the tile's outputs and their squared deviations about the TILE's own mean (no
            // E[x^2] - mean^2 cancellation: gn_finalize_kernel merges the tiles with Chan's formula).  This lane's column over its rows,
            // then over the four k-groups that hold the other rows of the tile.
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            const int cnt = min(tile_px, HWo - oy0 * a.Wo);                  // valid pixels of this tile (wave-uniform)
            const float mt = s1 / (float)cnt;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < NMT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = vals[i][r] - mt; if (ok[i][r]) m2 += d * d; }
            m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
            if (kq == 0 && colok) {   // every (sample, tile, column) has exactly one writer: no atomics, run-to-run identical
                float* cs = a.chsum + (((size_t)n * tiles_per_img + tile) * a.Cout + col) * 2;
                cs[0] = s1; cs[1] = m2;
            }
        }
    }
}

template <int NMT, int NCT, bool BF16>
__global__ __launch_bounds__(RDMI_THREADS) void tconv_kernel(TConvArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int tiles_per_img = (a.Ho + a.TR - 1) / a.TR;
    const int n = blockIdx.x / tiles_per_img, tile = blockIdx.x - n * tiles_per_img;
    const int oy0 = tile * a.TR;
    const int col0 = blockIdx.y * (64 * NCT) + wave * (16 * NCT) + lrow;       // column of this lane in its first column tile
    const int TRv = tconv_trv(a), Wl = tconv_wl(a);
    const int npix = TRv * Wl;
    const int vy0 = oy0 * a.stride - a.pad_lo, vx0 = -a.pad_lo;
    f32x4 acc[NMT][NCT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) acc[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS pixel of (row tile i, lane's output pixel) at tap (0, 0)
    int pbase[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
        const int m = min(i * 16 + lrow, a.TR * a.Wo - 1);
        const int oyl = m / a.Wo, ox = m - oyl * a.Wo;
        pbase[i] = (oyl * a.stride) * Wl + ox * a.stride;
    }
    // weight fragment pointers of the NCT column tiles (columns beyond the padded width re-read the last one; never stored)
    const float* Wf[NCT]; const bf16_t* Wh[NCT];
#pragma unroll
    for (int cc = 0; cc < NCT; ++cc) {
        const int cl = min(col0 + cc * 16, a.Cout_pad - 16 + lrow);
        Wf[cc] = a.wpk + (size_t)cl * 16 + kq * 4;
        Wh[cc] = reinterpret_cast<const bf16_t*>(a.wpk) + (size_t)cl * 32 + kq * 8;
    }
    const size_t bstride = (size_t)a.Cout_pad * (BF16 ? 32 : 16);
    const int nk = BF16 ? (a.Cv >> 5) : (a.Cv >> 4);                  // weight k-slabs per tap
    TcGeom geom;
    tconv_geom(a, n, vy0, vx0, Wl, npix, tid, geom);
    f32x4 raw[TC_MAXS];
    tconv_fetch(a, geom, 0, tid, raw);
    for (int c0 = 0; c0 < a.Cv; c0 += 32) {
        tconv_commit<BF16>(a, geom, n, c0, npix, tid, raw);
        __syncthreads();
        if (c0 + 32 < a.Cv) tconv_fetch(a, geom, c0 + 32, tid, raw);          // next slab's loads fly under this slab's MFMAs
        for (int t = 0; t < a.ntap; ++t) {
            const int toff = a.ntap == 1 ? 0 : (t / 3) * Wl + (t % 3);
            if (BF16) {
                u32x4 bf[NCT];
#pragma unroll
                for (int cc = 0; cc < NCT; ++cc) bf[cc] = *reinterpret_cast<const u32x4*>(Wh[cc] + ((size_t)t * nk + (c0 >> 5)) * bstride);
#pragma unroll
                for (int i = 0; i < NMT; ++i) {
                    const u32x4 af = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(rdmi_lds) + (size_t)(pbase[i] + toff) * RDMI_TC_ROW_BF16 + kq * 8);
#pragma unroll
                    for (int cc = 0; cc < NCT; ++cc) acc[i][cc] = mfma16_bf16(af, bf[cc], acc[i][cc]);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 bf[NCT];
#pragma unroll
                    for (int cc = 0; cc < NCT; ++cc) bf[cc] = ldg4(Wf[cc] + ((size_t)t * nk + (c0 >> 4) + h) * bstride);
#pragma unroll
                    for (int i = 0; i < NMT; ++i) {
                        const f32x4 af = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rdmi_lds) + (size_t)(pbase[i] + toff) * RDMI_TC_ROW_F32 + h * 16 + kq * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int cc = 0; cc < NCT; ++cc) acc[i][cc] = mfma16(af[j], bf[cc][j], acc[i][cc]);
                    }
                }
            }
        }
        __syncthreads();
    }
    tconv_epilogue<NMT, NCT>(a, acc, n, tile, tiles_per_img, oy0, col0, kq);
}

// ---- bf16 plan, pre-activated inputs -------------------------------------------------------------------------------------------
// PMC passes over the bf16 plan (profiles/r02_pmc_cifar_b64_bf16.txt) showed tconv_kernel<.., true> issuing ~12 VALU instructions
// per MFMA: GroupNorm + SiLU + bf16 conversion are redone for every staged window element -- 2.1x the tensor at 32x32 (halo rows)
// and once more per column-tile workgroup.  For the 3x3 stride-1 convs behind a GroupNorm the plan therefore writes the activated
// tensor ONCE as bf16 (gn_act_kernel, a streaming pass) and tconv_pre_kernel stages plain 16-byte copies of it.
struct GnActArgs {
    const float* A; const float* B;         // concat(A [n][HW][CA], B [n][HW][CB]) fp32; B may be null; CA, CB multiples of 4
    int CA, CB, Cv, HW, Cg, G, act;         // Cv = CA + CB padded to 32 (padding channels are written as zero)
    const float* stats;                     // [n][G][2] mean, rstd
    const float* gamma; const float* beta;
    bf16_t* out;                            // [n][HW][Cv]
    bf16_t* out2;                           // null, or [n][HW][Cv]: the RAW concat rounded to bf16 (the operand of the residual block's NIN_0 shortcut)
    int NB;
    // gn_act_fin_kernel: statistics formed in the kernel from the producers' per-tile channel records (what gn_finalize_kernel reads)
    const float* csA; const float* csB; int tilesA, tilesB, pxA, pxB; float eps;
};
__global__ __launch_bounds__(RDMI_THREADS) void gn_act_kernel(GnActArgs a) {
    const int U = a.Cv >> 3;                                   // 8-channel units per pixel
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)a.NB * a.HW * U) return;
    const int u = (int)(i % U);
    const long np = i / U;                                     // n * HW + p
    const int n = (int)(np / a.HW);
    const int Cin = a.CA + a.CB;
    unsigned o[4], o2[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = u * 8 + h * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        o2[2 * h] = o2[2 * h + 1] = 0u;
        if (c < Cin) {
            v = c < a.CA ? *reinterpret_cast<const f32x4*>(a.A + (size_t)np * a.CA + c) : *reinterpret_cast<const f32x4*>(a.B + (size_t)np * a.CB + (c - a.CA));
            o2[2 * h] = pack_bf16x2(v[0], v[1]); o2[2 * h + 1] = pack_bf16x2(v[2], v[3]);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c), bt = *reinterpret_cast<const f32x4*>(a.beta + c);
            const int g0 = c / a.Cg;
            const float m0 = a.stats[((size_t)n * a.G + g0) * 2], r0 = a.stats[((size_t)n * a.G + g0) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float mean = m0, rstd = r0;
                if (a.Cg & 3) {                                    // a group of 6 channels (C = 192) straddles quads: per-channel lookup
                    const int grp = (c + j) / a.Cg;
                    mean = a.stats[((size_t)n * a.G + grp) * 2]; rstd = a.stats[((size_t)n * a.G + grp) * 2 + 1];
                }
                const float y = (v[j] - mean) * (rstd * gm[j]) + bt[j];
                v[j] = a.act ? silu_f(y) : y;
            }
        }
        o[2 * h] = pack_bf16x2(v[0], v[1]); o[2 * h + 1] = pack_bf16x2(v[2], v[3]);
    }
    *reinterpret_cast<u32x4*>(a.out + (size_t)np * a.Cv + u * 8) = u32x4{o[0], o[1], o[2], o[3]};
    if (a.out2) *reinterpret_cast<u32x4*>(a.out2 + (size_t)np * a.Cv + u * 8) = u32x4{o2[0], o2[1], o2[2], o2[3]};
}

// gn_act_kernel with gn_finalize_kernel folded in (bf16 plan: one launch per GroupNorm instead of two, no statistics tensor):
// a workgroup owns a 64-channel slice x GA_PIX pixels of one sample and first forms (mean, rstd) of the <= 16 groups its slice
// touches from the producing convs' per-tile channel records -- 16 work-items per group add that group's (channel, tile) records
// in a fixed order, Chan's merge as in gn_finalize_kernel, xor-shuffles over the 16 lanes -- a few hundred bytes to a few KB of
// L2-resident records per workgroup, brought into LDS by one round of independent loads.  Then 16 work-items per pixel normalise / activate 4 channels each: 256 B read and 128 B
// written per pixel (full lines per wave instruction).  grid = (ceil(HW / GA_PIX), Cv / 64, NB).
#define GA_PIX 128
#define GA_MAXCH 96                                             // channels of the groups a 64-channel slice touches (Cg <= 16: 64 + 2 * 15 rounded up)
__host__ __device__ inline size_t gn_act_fin_lds_bytes(int tiles) { return 128 + (size_t)tiles * GA_MAXCH * 2 * sizeof(float); }
__global__ __launch_bounds__(RDMI_THREADS) void gn_act_fin_kernel(GnActArgs a) {
    float* tab = reinterpret_cast<float*>(rdmi_lds);          // [16][2] mean, rstd of the slice's groups
    float* rec = tab + 32;                                     // [tile][channel - lo][2]: the records of those groups' channels
    const int tid = threadIdx.x, n = blockIdx.z;
    const int Cin = a.CA + a.CB, c_lo = blockIdx.y * 64, c_hi = min(c_lo + 64, Cin);      // slice channels [c_lo, c_hi) (empty: padding only)
    const int g_first = c_lo / a.Cg;
    // 16 work-items per pixel, 4 channels each: a wave instruction reads whole 256-byte pixel rows of the slice (full lines) and writes
    // whole 128-byte rows.  The work-item's GA_PIX / 16 pixels are requested FIRST -- they do not depend on the statistics, so the
    // tensor streams in while the records are fetched and merged below
    const int c = c_lo + (tid & 15) * 4;
    const bool c_ok = c < Cin;
    constexpr int PP = RDMI_THREADS / 16, NP = GA_PIX / PP;
    const int p0 = blockIdx.x * GA_PIX + (tid >> 4);
    f32x4 v[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const size_t np = (size_t)n * a.HW + min(p0 + k * PP, a.HW - 1);
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c_ok) v[k] = c < a.CA ? *reinterpret_cast<const f32x4*>(a.A + np * a.CA + c) : *reinterpret_cast<const f32x4*>(a.B + np * a.CB + (c - a.CA));
    }
    if (c_lo < Cin) {
        // one round of independent loads brings the records in (a single L2 latency), the two passes then run out of LDS
        const int lo = g_first * a.Cg, hi = min(((c_hi - 1) / a.Cg + 1) * a.Cg, Cin), nch = hi - lo;
        const int tmax = max(a.tilesA, a.tilesB);
        typedef float f32x2 __attribute__((vector_size(8)));
        for (int e = tid; e < tmax * nch; e += RDMI_THREADS) {
            const int t = e / nch, c = lo + (e - t * nch);
            const bool inA = c < a.CA;
            const int tiles = inA ? a.tilesA : a.tilesB;
            f32x2 v = {0.f, 0.f};
            if (t < tiles) v = *reinterpret_cast<const f32x2*>((inA ? a.csA : a.csB) + (((size_t)n * tiles + t) * (inA ? a.CA : a.CB) + (inA ? c : c - a.CA)) * 2);
            *reinterpret_cast<f32x2*>(rec + (size_t)e * 2) = v;
        }
        __syncthreads();
        const int slot = tid >> 4, sub = tid & 15, g = g_first + slot;
        const bool live = g * a.Cg < c_hi && g < a.G;
        float s1 = 0.f;
        if (live)
            for (int e = sub; e < a.Cg * tmax; e += 16) {
                const int cc = e / tmax, t = e - cc * tmax;
                s1 += rec[((size_t)t * nch + (g * a.Cg + cc - lo)) * 2];
            }
        for (int m = 1; m < 16; m <<= 1) s1 += __shfl_xor(s1, m);
        const float cnt = (float)(a.Cg * a.HW);
        const float mean = s1 / cnt;
        float m2 = 0.f;
        if (live)
            for (int e = sub; e < a.Cg * tmax; e += 16) {
                const int cc = e / tmax, t = e - cc * tmax, c = g * a.Cg + cc;
                const bool inA = c < a.CA;
                const int tiles = inA ? a.tilesA : a.tilesB, px = inA ? a.pxA : a.pxB;
                if (t < tiles) {
                    const float* p = rec + ((size_t)t * nch + (c - lo)) * 2;
                    const float nt = (float)min(px, a.HW - t * px), d = p[0] / nt - mean;
                    m2 += p[1] + nt * d * d;
                }
            }
        for (int m = 1; m < 16; m <<= 1) m2 += __shfl_xor(m2, m);
        if (sub == 0) { tab[slot * 2] = mean; tab[slot * 2 + 1] = 1.0f / sqrtf(m2 / cnt + a.eps); }
    }
    __syncthreads();
    float mean[4], rg[4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = 0.f; rg[j] = 0.f; bt[j] = 0.f;
        if (c + j < Cin) {
            const int slot = (c + j) / a.Cg - g_first;
            mean[j] = tab[slot * 2]; rg[j] = tab[slot * 2 + 1] * a.gamma[c + j]; bt[j] = a.beta[c + j];
        }
    }
    typedef unsigned int u32x2 __attribute__((vector_size(8)));
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = p0 + k * PP;
        if (p >= a.HW) break;
        f32x4 w = v[k];
        if (a.out2) *reinterpret_cast<u32x2*>(a.out2 + ((size_t)n * a.HW + p) * a.Cv + c) = u32x2{pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3])};     // padding channels were loaded as zero
        if (c_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = (w[j] - mean[j]) * rg[j] + bt[j];
                w[j] = a.act ? silu_f(y) : y;
            }
        }
        *reinterpret_cast<u32x2*>(a.out + ((size_t)n * a.HW + p) * a.Cv + c) = u32x2{pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3])};
    }
}

// 3x3 (stride 1, pad 1) or 1x1 conv over a bf16 tensor that is already normalised / activated (TConvArgs: srcA = that tensor,
// CA = Cv, CB = 0, stats = null; Cv a multiple of 64 -- planner).  Same tiling as tconv_kernel.  A slab is KS channels of the
// window ([pixel][KS + 16] bf16 in LDS: that row stride keeps the ds_read_b128 A fragments conflict-free on gfx950's lane groups; the 16-byte
// staging stores are 8 contiguous lanes per pixel),
// staged as 16-byte copies that are register-prefetched one slab ahead; KS = 64 for 3x3 (18 MFMA groups per slab), 256 for 1x1
// (8 groups: a whole NIN contraction in one or two barrier pairs).
#ifndef RDMI_TPRE_PAD
#define RDMI_TPRE_PAD 16         // row stride KS + 16 bf16 = 10 (or 34) units of 16 bytes: conflict-free A fragments (see RDMI_TC_ROW_*); + 8 was 2-way
#endif
template <int NTAP> struct TpCfg { static constexpr int KS = NTAP == 1 ? 256 : 64, ROW = KS + RDMI_TPRE_PAD, UPP = KS / 8, GROUPS = NTAP * (KS / 32); };
__host__ __device__ inline size_t tconv_pre_lds_bytes(const TConvArgs& a) {
    const int row = a.ntap == 1 ? TpCfg<1>::ROW : TpCfg<9>::ROW;
    return (size_t)tconv_trv(a) * tconv_wl(a) * row * 2;
}

template <int NMT, int NCT, int NTAP>
__global__ __launch_bounds__(RDMI_THREADS) void tconv_pre_kernel(TConvArgs a) {
    constexpr int KS = TpCfg<NTAP>::KS, ROW = TpCfg<NTAP>::ROW, UPP = TpCfg<NTAP>::UPP, GROUPS = TpCfg<NTAP>::GROUPS, KSTEPS = KS / 32;
    constexpr int HALO = NTAP == 9 ? 1 : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int tiles_per_img = (a.Ho + a.TR - 1) / a.TR;
    const int n = blockIdx.x / tiles_per_img, tile = blockIdx.x - n * tiles_per_img;
    const int oy0 = tile * a.TR;
    const int col0 = blockIdx.y * (64 * NCT) + wave * (16 * NCT) + lrow;
    const int TRv = a.TR + 2 * HALO, Wl = a.Wo + 2 * HALO, npix = TRv * Wl;
    bf16_t* win = reinterpret_cast<bf16_t*>(rdmi_lds);
    f32x4 acc[NMT][NCT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) acc[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragments: wave-uniform (tap, k-step) block + 32-bit lane offset; A fragments: lane byte offset + wave-uniform tap offset
    unsigned whoff[NCT];
#pragma unroll
    for (int cc = 0; cc < NCT; ++cc) {
        const int cl = min(col0 + cc * 16, a.Cout_pad - 16 + lrow);
        whoff[cc] = (unsigned)(cl * 32 + kq * 8) * 2u;
    }
    const size_t bstride = (size_t)a.Cout_pad * 32 * 2;             // bytes per (tap, 32-channel k-step) block
    const int nk = a.Cv >> 5;
    int lbase[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
        const int m = min(i * 16 + lrow, a.TR * a.Wo - 1);
        const int oyl = m / a.Wo, ox = m - oyl * a.Wo;
        lbase[i] = ((oyl * Wl + ox) * ROW + kq * 8) * 2;
    }
    // staging geometry: unit i = tid + k * 256 is 8 channels (16 bytes) of window pixel i / UPP; source byte offset or -1
    const int u8 = (tid % UPP) * 8;
    int soff[TC_MAXS];
    const char* base = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(a.srcA) + (size_t)n * a.Ha * a.Wa * a.Cv);
#pragma unroll
    for (int k = 0; k < TC_MAXS; ++k) {
        const int i = tid + k * RDMI_THREADS, p = i / UPP;
        const int ry = p / Wl, rx = p - ry * Wl;
        const int vy = oy0 - HALO + ry, vx = rx - HALO;
        soff[k] = (i < npix * UPP && vy >= 0 && vy < a.Hv && vx >= 0 && vx < a.Wv) ? ((vy * a.Wa + vx) * a.Cv + u8) * 2 : -1;
    }
    u32x4 raw[TC_MAXS];
    auto fetch = [&](int c0) {
        const bool cok = c0 + u8 < a.Cv;
#pragma unroll
        for (int k = 0; k < TC_MAXS; ++k) {
            const bool ok = cok && soff[k] >= 0;
            const u32x4 v = *reinterpret_cast<const u32x4*>(base + (ok ? soff[k] + c0 * 2 : 0));
            raw[k] = u32x4{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u};
        }
    };
    // The (tap, k-step) MFMA groups of a slab are unrolled with the B fragments in a ring of register sets loaded several
    // groups ahead (the first ones before the slab's window is committed), so the weight loads' L2 latency is covered by MFMA work;
    // sched_fence / opaque_sgpr keep the compiler from hoisting every group's loads and addresses (which spills).
    // Group j = (tap j / KSTEPS, k-step j % KSTEPS); k-steps beyond the tensor's channels (1x1 over fewer than 256) are skipped.
    auto loadB = [&](int j, int c0, u32x4 (&b)[NCT]) {
        const int t = j / KSTEPS, ks = j % KSTEPS;
        const int kstep = min((c0 >> 5) + ks, nk - 1);          // clamped: a skipped group still loads a valid block
        const char* blk = reinterpret_cast<const char*>(a.wpk) + ((size_t)t * nk + kstep) * bstride;      // wave-uniform
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) b[cc] = *reinterpret_cast<const u32x4*>(blk + whoff[cc]);
    };
    auto mma = [&](int j, const u32x4 (&b)[NCT]) {
        const int t = j / KSTEPS, ks = j % KSTEPS;
        int toffb = (((t / 3) * Wl + (t % 3)) * ROW + ks * 32) * 2;
        opaque_sgpr(toffb);
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
            const u32x4 af = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(win) + lbase[i] + toffb);
#pragma unroll
            for (int cc = 0; cc < NCT; ++cc) acc[i][cc] = mfma16_bf16(af, b[cc], acc[i][cc]);
        }
    };
    // ring depth: the loads of group j + RING - 1 are issued before the MFMAs of group j (an L2 hit takes 500-800 cycles, a group
    // of NMT * NCT MFMAs 128-256): deeper for the narrow column tiles, whose groups are short and whose fragments are few registers
    constexpr int RING = NCT >= 4 ? 4 : 6, AHEAD = RING - 1;
    // the next slab's window is register-prefetched under this slab's MFMAs only where that does not cost a wave per SIMD (PF).  Registers are
    // VGPRs + the accumulators' AGPRs: NCT = 2 needs 150 + 32 with the prefetch (2 waves per SIMD) and 127 + 32 without (3 waves) and measured
    // faster without; NCT = 1 (110 + 16) holds 4 waves either way; NCT = 4 (184 + 64 = 248) is bound to 2 waves by its accumulators, weight ring
    // and vector epilogue, which is also why its ring cannot be deeper than 4
#ifndef RDMI_TPRE_PF
#define RDMI_TPRE_PF(NCT) ((NCT) != 2)
#endif
    constexpr bool PF = RDMI_TPRE_PF(NCT);
    if (PF) fetch(0);
    for (int c0 = 0; c0 < a.Cv; c0 += KS) {
        u32x4 br[RING][NCT];
        if (!PF) fetch(c0);
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
            if (j < GROUPS) loadB(j, c0, br[j]);
#pragma unroll
        for (int k = 0; k < TC_MAXS; ++k) {
            const int i = tid + k * RDMI_THREADS;
            if (i < npix * UPP) *reinterpret_cast<u32x4*>(win + (size_t)(i / UPP) * ROW + u8) = raw[k];
        }
        __syncthreads();
        if (PF && c0 + KS < a.Cv) fetch(c0 + KS);              // next slab's loads fly under this slab's MFMAs
        const int ksteps = min(KSTEPS, nk - (c0 >> 5));
#pragma unroll
        for (int j = 0; j < GROUPS; ++j) {
            if (j + AHEAD < GROUPS) loadB(j + AHEAD, c0, br[(j + AHEAD) % RING]);
            if (KSTEPS == 2 || (j % KSTEPS) < ksteps) mma(j, br[j % RING]);      // Cv % 64 == 0: a 64-channel slab is always whole
            sched_fence();
        }
        __syncthreads();
    }
    tconv_epilogue<NMT, NCT>(a, acc, n, tile, tiles_per_img, oy0, col0, kq);
}

// ---- bf16 plan, implicit GEMM (32x32 / 16x16 / 8x8 levels at sampling batches) -------------------------------------------------
// tconv_pre_kernel keeps a 64-pixel window in LDS and streams every weight fragment from L2 into registers: at NCT = 4 that is
// 1 KiB of L2 traffic per 4 MFMAs and wave (64 B/clk/CU at the matrix pipe's rate -- the CU's whole L1 fill rate), and a
// workgroup's 64 output pixels re-read all 1.2 MB of a 256 x 256 conv's weights.  iconv_kernel is the conv as a plain implicit GEMM,
//   out[m][co] = sum_{tap, ci} act[pixel(m) + tap][ci] * W[tap][ci][co],   m = (sample, oy, ox) over the WHOLE batch,
// on 128 x 128 output tiles with 64-deep K steps ((tap, 64-channel slab) pairs): both operand tiles go global -> LDS by the LDS-DMA
// (glds16: no staging registers, no ds_write pass), 16 KiB each, as [row][64 bf16] images whose 16-byte chunks are XOR-swizzled
// with (row & 7) on the SOURCE side (lane -> chunk) so that every ds_read_b128 fragment read is bank-conflict-free; each of the
// 2 x 2 waves owns 64 x 64 outputs (4 x 4 MFMA tiles: a fragment feeds four MFMAs, half the LDS bytes per MFMA of the window
// form, a quarter of the weight bytes per output).  The im2col gather is the per-lane source address: the tap shift and the zero
// padding (a lane whose window pixel is outside the image reads the 16 zero bytes of a.zeros); stride 1, no upsampling.  K order: slab-major, tap-minor, so the nine taps of a slab re-read the same
// (128 + halo) x 128 B of activations from the CU's L1.  One LDS buffer, two barriers per K step; latency is covered by the
// 3-4 workgroups a CU holds (<= 128 VGPRs, 32 KiB of LDS each).  Epilogue = tconv_epilogue (a wave's 64 rows are one 64-pixel
// tile of one sample: the planner's per-tile GroupNorm records keep their meaning).  Needs Ho * Wo % 64 == 0, Cout % 128 == 0,
// Cv % 64 == 0 (host-checked).
__host__ __device__ inline size_t iconv_lds_bytes() { return 2 * 128 * 128; }
#ifndef RDMI_ICONV_WAVES
#define RDMI_ICONV_WAVES 3
#endif
template <int NTAP>
__global__ __launch_bounds__(RDMI_THREADS, RDMI_ICONV_WAVES) void iconv_kernel(TConvArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int HWo = a.Ho * a.Wo;
    const long M = (long)a.NB * HWo;
    const long m0 = (long)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    // staging: this work-item copies chunk (tid & 7) of image rows (tid >> 3) + 32 j, j = 0..3, of both operand tiles; the chunk
    // holds the tile's k = 8 * ((tid & 7) ^ (row & 7)) .. + 7.  Per row: byte offset of the window's (0, 0) tap pixel (may point
    // before the tensor: only dereferenced where the tap's bit in `tapok` says the pixel exists) -- stride 1, no upsampling: a tap
    // is a wave-uniform byte delta.
    const int srow = tid >> 3;
    const int c8 = ((tid & 7) ^ (srow & 7)) * 8;
    long roff[4]; unsigned tapok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long gm = m0 + j * 32 + srow;
        roff[j] = 0; tapok[j] = 0;
        if (gm < M) {
            const int n = (int)(gm / HWo), p = (int)(gm - (long)n * HWo);
            const int oy = p / a.Wo, ox = p - oy * a.Wo;
            roff[j] = (((long)n * a.Ha + (oy - a.pad_lo)) * a.Wa + (ox - a.pad_lo)) * a.Cv * 2 + c8 * 2;
            for (int t = 0; t < NTAP; ++t) {
                const int vy = oy - a.pad_lo + (NTAP == 9 ? t / 3 : 0), vx = ox - a.pad_lo + (NTAP == 9 ? t % 3 : 0);
                if ((unsigned)vy < (unsigned)a.Hv && (unsigned)vx < (unsigned)a.Wv) tapok[j] |= 1u << t;
            }
        }
    }
    const char* abase = reinterpret_cast<const char*>(a.srcA);
    const char* zsrc = reinterpret_cast<const char*>(a.zeros) + (tid & 7) * 16;
    const int nk = a.Cv >> 5;
    // weight source of row (column) n0 + srow + 32 j at (tap t, slab c0): block (t * nk + (c0 + c8) / 32), 64 bytes per column
    const char* wbase = reinterpret_cast<const char*>(a.wpk) + ((size_t)(c8 >> 5) * a.Cout_pad + n0 + srow) * 64 + ((c8 >> 3) & 3) * 16;
    const long wblk = (long)a.Cout_pad * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned sw = (unsigned)(lrow & 7);
    const unsigned aoff = (unsigned)(wr * 64 + lrow) * 128u, boff = 16384u + (unsigned)(wc * 64 + lrow) * 128u;
    const unsigned dst = (unsigned)wave * 1024u;                   // + j * 4096 (+ 16384 for the weight tile)
    const long rowb = (long)a.Wa * a.Cv * 2;
    for (int c0 = 0; c0 < a.Cv; c0 += 64) {
        for (int t = 0; t < NTAP; ++t) {
            const int dy = NTAP == 9 ? t / 3 : 0, dx = NTAP == 9 ? t - dy * 3 : 0;
            const long tdelta = dy * rowb + (long)(dx * a.Cv + c0) * 2;          // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16((tapok[j] >> t & 1u) ? abase + (roff[j] + tdelta) : zsrc, dst + j * 4096u);
            const char* wsrc = wbase + ((long)t * nk + (c0 >> 5)) * wblk;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(wsrc + j * 2048, 16384u + dst + j * 4096u);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned ch = ((unsigned)(ks * 4 + kq) ^ sw) * 16u;
                u32x4 af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4*>(rdmi_lds + aoff + i * 2048u + ch);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) bf[cc] = *reinterpret_cast<const u32x4*>(rdmi_lds + boff + cc * 2048u + ch);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[i][cc] = mfma16_bf16(af[i], bf[cc], acc[i][cc]);
            }
            __syncthreads();
        }
    }
    const long gm0 = m0 + wr * 64;                                  // the wave's 64 rows: one 64-pixel tile of one sample
    if (gm0 < M) {
        const int n = (int)(gm0 / HWo), p0 = (int)(gm0 - (long)n * HWo);
        tconv_epilogue<4, 4>(a, acc, n, p0 >> 6, HWo >> 6, p0 / a.Wo, n0 + wc * 64 + lrow, kq);
    }
}

// GroupNorm statistics of concat(A, B) from the producers' per-tile channel records (sum, squared deviations about the tile's own
// mean): stats[n][g] = (mean, rstd).  The group's mean comes from the sums; its M2 is Chan's parallel merge
// M2 = sum_t [ M2_t + n_t (mean_t - mean)^2 ] -- no E[x^2] - mean^2 cancellation, so a group whose mean is large against its
// spread (trained checkpoints) keeps its variance (round-2 advisor finding).  pxA / pxB: pixels per full tile of each producer (the
// last tile of an image may hold fewer).  grid = NB, 256 work-items.
__global__ __launch_bounds__(RDMI_THREADS) void gn_finalize_kernel(const float* __restrict__ csA, const float* __restrict__ csB, int CA, int CB, int tilesA, int tilesB,
                                                                    int pxA, int pxB, int HW, int G, float eps, float* __restrict__ stats) {
    // eight lanes per group (G <= 32): lane `sub` adds the (channel, tile) pairs sub, sub + 8, ... in a fixed order, then the eight
    // partial sums are merged by xor-shuffles -- the same order on every run
    const int n = blockIdx.x, g = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int C = CA + CB, Cg = C / G;
    const int tmax = max(tilesA, tilesB);
    float s1 = 0.f;
    if (g < G) {
        for (int e = sub; e < Cg * tmax; e += 8) {
            const int cc = e / tmax, t = e - cc * tmax, c = g * Cg + cc;
            const bool inA = c < CA;
            const int tiles = inA ? tilesA : tilesB;
            if (t < tiles) s1 += ((inA ? csA : csB) + (((size_t)n * tiles + t) * (inA ? CA : CB) + (inA ? c : c - CA)) * 2)[0];
        }
    }
    for (int m = 1; m < 8; m <<= 1) s1 += __shfl_xor(s1, m);
    const float cnt = (float)(Cg * HW);
    const float mean = s1 / cnt;
    float m2 = 0.f;
    if (g < G) {
        for (int e = sub; e < Cg * tmax; e += 8) {
            const int cc = e / tmax, t = e - cc * tmax, c = g * Cg + cc;
            const bool inA = c < CA;
            const int tiles = inA ? tilesA : tilesB, px = inA ? pxA : pxB;
            if (t < tiles) {
                const float* p = (inA ? csA : csB) + (((size_t)n * tiles + t) * (inA ? CA : CB) + (inA ? c : c - CA)) * 2;
                const float nt = (float)min(px, HW - t * px), d = p[0] / nt - mean;
                m2 += p[1] + nt * d * d;
            }
        }
    }
    for (int m = 1; m < 8; m <<= 1) m2 += __shfl_xor(m2, m);
    if (g < G && sub == 0) {
        stats[((size_t)n * G + g) * 2] = mean;
        stats[((size_t)n * G + g) * 2 + 1] = 1.0f / sqrtf(m2 / cnt + eps);
    }
}

// mean / rstd of GroupNorm group g of sample n over concat(A, B): grid = (G, NB); two passes (exact like F.group_norm)
__global__ __launch_bounds__(RDMI_THREADS) void gn_stats_kernel(const float* __restrict__ A, const float* __restrict__ B, int CA, int CB, int HW,
                                                                 int G, float eps, float* __restrict__ stats) {
    float* red = reinterpret_cast<float*>(rdmi_lds);      // [4]
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int C = CA + CB, Cg = C / G, c0 = g * Cg;
    const int total = HW * Cg;
    auto at = [&](int i) {
        const int p = i / Cg, c = c0 + (i - p * Cg);
        return c < CA ? A[((size_t)n * HW + p) * CA + c] : B[((size_t)n * HW + p) * CB + (c - CA)];
    };
    auto block_sum = [&](float v) {
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    float s = 0.f;
    for (int i = tid; i < total; i += RDMI_THREADS) s += at(i);
    const float mean = block_sum(s) / (float)total;
    float q = 0.f;
    for (int i = tid; i < total; i += RDMI_THREADS) { const float d = at(i) - mean; q += d * d; }
    const float var = block_sum(q) / (float)total;
    if (tid == 0) { stats[((size_t)n * G + g) * 2] = mean; stats[((size_t)n * G + g) * 2 + 1] = 1.0f / sqrtf(var + eps); }
}

// C[n][m][j] = alpha * sum_k A[n][m][k] * B[n][j][k]  (both operands K-contiguous with row strides lda / ldb and batch strides
// sa / sb).  Workgroup = 64 x 64 tile: wave w owns rows 16w..16w+15 and all four 16-column tiles.  M, N multiples of 16, K of 16.
struct BgemmArgs { const float* A; const float* B; float* C; long sa, sb, sc; int lda, ldb, ldc; int M, N, K; float alpha; };
__global__ __launch_bounds__(RDMI_THREADS) void bgemm_nt_kernel(BgemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int n = blockIdx.z;
    const int row = blockIdx.x * 64 + wave * 16 + lrow;
    const float* Ap = a.A + (size_t)n * a.sa + (size_t)min(row, a.M - 1) * a.lda + kq * 4;
    const float* Bp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Bp[t] = a.B + (size_t)n * a.sb + (size_t)min((int)blockIdx.y * 64 + t * 16 + lrow, a.N - 1) * a.ldb + kq * 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; k += 16) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(Ap + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(Bp[t] + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t] = mfma16(af[j], bf[j], acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = blockIdx.y * 64 + t * 16 + lrow;
        if (c >= a.N) continue;
        for (int r = 0; r < 4; ++r) {
            const int m = blockIdx.x * 64 + wave * 16 + kq * 4 + r;
            if (m < a.M) a.C[(size_t)n * a.sc + (size_t)m * a.ldc + c] = acc[t][r] * a.alpha;
        }
    }
}

// bf16 variant: the fp32 operands are rounded to bf16 as they are loaded (8 consecutive k per lane), fp32 accumulate.  K % 32 == 0.
__global__ __launch_bounds__(RDMI_THREADS) void bgemm_nt_bf16_kernel(BgemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int n = blockIdx.z;
    const int row = blockIdx.x * 64 + wave * 16 + lrow;
    const float* Ap = a.A + (size_t)n * a.sa + (size_t)min(row, a.M - 1) * a.lda + kq * 8;
    const float* Bp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Bp[t] = a.B + (size_t)n * a.sb + (size_t)min((int)blockIdx.y * 64 + t * 16 + lrow, a.N - 1) * a.ldb + kq * 8;
    auto frag = [](const float* p) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p), y = *reinterpret_cast<const f32x4*>(p + 4);
        return u32x4{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    };
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < a.K; k += 32) {
        const u32x4 af = frag(Ap + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16_bf16(af, frag(Bp[t] + k), acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = blockIdx.y * 64 + t * 16 + lrow;
        if (c >= a.N) continue;
        for (int r = 0; r < 4; ++r) {
            const int m = blockIdx.x * 64 + wave * 16 + kq * 4 + r;
            if (m < a.M) a.C[(size_t)n * a.sc + (size_t)m * a.ldc + c] = acc[t][r] * a.alpha;
        }
    }
}

// in-place softmax over the last dimension of [rows][L]: one wave per row (F.softmax(w, dim=-1), layerspp.py:89)
__global__ __launch_bounds__(RDMI_THREADS) void softmax_rows_kernel(float* __restrict__ x, long rows, int L) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * L;
    float mx = -3.0e38f;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, p[i]);
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float s = 0.f;
    for (int i = lane; i < L; i += 64) { const float e = __expf(p[i] - mx); p[i] = e; s += e; }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float inv = 1.0f / s;
    for (int i = lane; i < L; i += 64) p[i] *= inv;
}

// ---- bf16 plan: softmax(Q K^T / sqrt(C)) V in one kernel (AttnBlockpp, RD/models/layerspp.py:84-92) -----------------------------
// Replaces bgemm (scores) + softmax_rows + bgemm (P V) and the [L][L] score buffer.  grid (L / 64, NB), 4 waves; wave w owns the 16
// queries 64 * blockIdx.x + 16 w.  Everything is computed TRANSPOSED so that no accumulator ever has to change layout:
//   S^T tile (16 keys x 16 queries) = K_tile (A operand: row = key) . Q^T (B operand: column = query, kept in registers, pre-scaled)
//   -> a lane holds, for ITS query (lane & 15), the keys 16 T + 4 (lane >> 4) + r: online softmax needs only two cross-lane steps;
//   O^T tile (16 channels x 16 queries) += V^T_tile (A: row = channel) . P^T (B: column = query): the contraction runs over keys
//   in the order the lane already holds them (slots 0..3 <- tile 2s, 4..7 <- tile 2s + 1), and the V^T fragment is read from LDS in
//   that same order (two ds_read_b64), so P goes from accumulator to operand with a bf16 conversion only.
// K blocks of 64 keys are staged as [key][C + 8] bf16, V^T blocks (from the [C][L] transpose the plan already makes) as
// [channel][64 + 8] bf16; fp32 running max / sum / output accumulators.
struct FlashArgs { const float* qkv; const float* vt; float* out; int L, NB; float alpha; int out_bf16; int in_bf16; };     // in_bf16: qkv and vt hold bf16 (same element layout)   // qkv [n][L][3C] (q | k | v), vt [n][C][L], out [n][L][C] (fp32, or bf16 for a tconv_pre consumer)
template <int C>
#ifndef RDMI_FLASH_KPAD
#define RDMI_FLASH_KPAD 8        // K rows of C + 8 bf16; C + 16 (conflict-free ds_read_b128 K fragments on gfx950's lane groups, as in the convs) measured SLOWER here: 12.60 vs 12.45 ms per update
#endif
__host__ __device__ inline size_t flash_lds_bytes() { return ((size_t)64 * (C + RDMI_FLASH_KPAD) + (size_t)C * 72) * 2; }

template <int C>
__global__ __launch_bounds__(RDMI_THREADS) void flash_attn_bf16_kernel(FlashArgs a) {
    constexpr int KS = C / 32, UT = C / 16, KR = C + RDMI_FLASH_KPAD, VR = 72;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int n = blockIdx.y, q = blockIdx.x * 64 + wave * 16 + l15;
    bf16_t* Kl = reinterpret_cast<bf16_t*>(rdmi_lds);        // [64][KR]
    bf16_t* Vl = Kl + 64 * KR;                                // [C][VR]
    const float* qkv = a.qkv + (a.in_bf16 ? (size_t)n * a.L * 3 * C / 2 : (size_t)n * a.L * 3 * C);
    const float* vt = a.vt + (a.in_bf16 ? (size_t)n * C * a.L / 2 : (size_t)n * C * a.L);
    const bf16_t* qkv16 = reinterpret_cast<const bf16_t*>(qkv);
    const bf16_t* vt16 = reinterpret_cast<const bf16_t*>(vt);
    // this lane's query fragments (B operand): channels 32 s + 8 g .. + 7, pre-scaled by 1 / sqrt(C)
    u32x4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const f32x4 v0 = ldact4(qkv, (size_t)q * 3 * C + 32 * s + 8 * g, a.in_bf16);
        const f32x4 v1 = ldact4(qkv, (size_t)q * 3 * C + 32 * s + 8 * g + 4, a.in_bf16);
        qf[s] = u32x4{pack_bf16x2(v0[0] * a.alpha, v0[1] * a.alpha), pack_bf16x2(v0[2] * a.alpha, v0[3] * a.alpha),
                      pack_bf16x2(v1[0] * a.alpha, v1[1] * a.alpha), pack_bf16x2(v1[2] * a.alpha, v1[3] * a.alpha)};
    }
    f32x4 o[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -3.0e38f, lsum = 0.f;
    typedef unsigned int u32x2 __attribute__((vector_size(8)));
    for (int kb = 0; kb < a.L; kb += 64) {
        __syncthreads();                                       // previous block fully consumed
        if (a.in_bf16) {
            // bf16 q | k | v: both blocks are plain 16-byte copies (half the instructions of the fp32 form, no conversion)
#pragma unroll 4
            for (int it = 0; it < C / 32; ++it) {              // K block: 64 keys x C channels
                const int idx = tid + it * RDMI_THREADS, key = idx / (C / 8), c8 = idx - key * (C / 8);
                *reinterpret_cast<u32x4*>(Kl + key * KR + 8 * c8) = *reinterpret_cast<const u32x4*>(qkv16 + (size_t)(kb + key) * 3 * C + C + 8 * c8);
            }
#pragma unroll 4
            for (int it = 0; it < C / 32; ++it) {              // V^T block: C channels x 64 keys
                const int idx = tid + it * RDMI_THREADS, c = idx >> 3, k8 = idx & 7;
                *reinterpret_cast<u32x4*>(Vl + c * VR + 8 * k8) = *reinterpret_cast<const u32x4*>(vt16 + (size_t)c * a.L + kb + 8 * k8);
            }
        } else {
#pragma unroll 4
        for (int it = 0; it < C / 16; ++it) {                  // K block: 64 keys x C channels (4 loads in flight: the accumulators need the registers)
            const int idx = tid + it * RDMI_THREADS, key = idx / (C / 4), c4 = idx - key * (C / 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(qkv + (size_t)(kb + key) * 3 * C + C + 4 * c4);
            *reinterpret_cast<u32x2*>(Kl + key * KR + 4 * c4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
#pragma unroll 4
        for (int it = 0; it < C / 16; ++it) {                  // V^T block: C channels x 64 keys
            const int idx = tid + it * RDMI_THREADS, c = idx >> 4, k4 = idx & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(vt + (size_t)c * a.L + kb + 4 * k4);
            *reinterpret_cast<u32x2*>(Vl + c * VR + 4 * k4) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
        }
        __syncthreads();
        // S^T: four key tiles x KS k-steps
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Kl + (16 * t + l15) * KR + 32 * s + 8 * g);
                st[t] = mfma16_bf16(kf, qf[s], st[t]);
            }
        // online softmax over this lane's 16 keys, then over the four k-groups that hold the same query
        float mx = m;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float corr = __expf(m - mx);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[t][r] = __expf(st[t][r] - mx); ps += st[t][r]; }
        ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
        lsum = lsum * corr + ps; m = mx;
#pragma unroll
        for (int u = 0; u < UT; ++u) o[u] *= corr;
        // O^T += V^T . P^T: two k-steps of 32 keys (tiles 2 s2, 2 s2 + 1)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4 pf = u32x4{pack_bf16x2(st[2 * s2][0], st[2 * s2][1]), pack_bf16x2(st[2 * s2][2], st[2 * s2][3]),
                                   pack_bf16x2(st[2 * s2 + 1][0], st[2 * s2 + 1][1]), pack_bf16x2(st[2 * s2 + 1][2], st[2 * s2 + 1][3])};
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                const bf16_t* vr = Vl + (16 * u + l15) * VR + 32 * s2 + 4 * g;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vr), hi = *reinterpret_cast<const u32x2*>(vr + 16);
                o[u] = mfma16_bf16(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf, o[u]);
            }
        }
    }
    const float inv = 1.0f / lsum;
    if (a.out_bf16) {
        bf16_t* ob = reinterpret_cast<bf16_t*>(a.out) + ((size_t)n * a.L + q) * C;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const f32x4 v = o[u] * inv;
            *reinterpret_cast<u32x2*>(ob + 16 * u + 4 * g) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
        return;
    }
    float* og = a.out + ((size_t)n * a.L + q) * C;
#pragma unroll
    for (int u = 0; u < UT; ++u) *reinterpret_cast<f32x4*>(og + 16 * u + 4 * g) = o[u] * inv;
}

// dst[n][c][l] = src[n][l][c0 + c] (row stride lds_): V -> V^T for the P V product
__global__ __launch_bounds__(RDMI_THREADS) void transpose_lc_kernel(const float* __restrict__ src, float* __restrict__ dst, int NB, int L, int C, int ld, int c0) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)NB * L * C) return;
    const int l = (int)(i % L);
    const long r = i / L;
    const int c = (int)(r % C);
    const long n = r / C;
    dst[i] = src[((size_t)n * L + l) * ld + c0 + c];
}

// same through a 64 x 64 LDS tile (L, C multiples of 64): both the strided read and the write are 256-byte runs.  grid (L/64, C/64, NB)
__global__ __launch_bounds__(RDMI_THREADS) void transpose_lc_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int L, int C, int ld, int c0) {
    __shared__ float tile[64][65];
    const int l0 = blockIdx.x * 64, cb = blockIdx.y * 64, n = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) tile[r][tx] = src[((size_t)n * L + l0 + r) * ld + c0 + cb + tx];
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) dst[((size_t)n * C + cb + r) * L + l0 + tx] = tile[tx][r];
}

// the same 64 x 64 tile transpose over bf16 tensors (q | k | v stored as bf16 for the fused attention core)
__global__ __launch_bounds__(RDMI_THREADS) void transpose_lc_tile16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int L, int C, int ld, int c0) {
    __shared__ bf16_t tile16[64][66];
    const int l0 = blockIdx.x * 64, cb = blockIdx.y * 64, n = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) tile16[r][tx] = src[((size_t)n * L + l0 + r) * ld + c0 + cb + tx];
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) dst[((size_t)n * C + cb + r) * L + l0 + tx] = tile16[tx][r];
}

// API boundary for channels > 1: NCHW (the reference's layout) <-> NHWC
__global__ __launch_bounds__(RDMI_THREADS) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int NB, int HW, int C, int x_mod) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)NB * HW * C) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const int p = (int)(r % HW);
    const long n = r / HW;
    const long ns = x_mod > 0 ? n % x_mod : n;
    dst[i] = src[(ns * C + c) * HW + p];
}
