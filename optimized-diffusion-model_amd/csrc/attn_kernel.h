// AttnBlockpp (RD/models/layerspp.py:67-96) as ONE kernel, one workgroup per sample, everything in LDS:
//   GroupNorm -> q,k,v = NIN_0..2 -> softmax_k( q.k / sqrt(C) ) -> .v -> NIN_3 -> (x + h)/sqrt2
// All five contractions run on the fp32 MFMA (16x16x4); the 81x81 score tile never leaves the CU
// ("one 81x81 fp32 score tile = 26 KB -> whole attention lives in one workgroup's LDS").
// Softmax is done in the MFMA accumulator layout: a wave owns whole query rows (16 per m-tile), so a row's
// max / sum is a 4-step xor-shuffle over the 16 lanes that hold its columns -- no LDS round trip.
#pragma once
#include "common.h"

struct AttnArgs {
    const float* x;        // [n][L][C]
    float* out;            // [n][L][C]
    const float* gamma; const float* beta;
    const float* wqkv;     // packed [3][C/16][C][16]   (NIN_0, NIN_1, NIN_2)
    const float* bqkv;     // [3][C]
    const float* w3;       // packed [C/16][C][16]      (NIN_3)
    const float* b3;       // [C]
    int NB, L, Lpad, G;
    float eps, scale, out_scale;
    int io_bf16;           // x and out are bf16 (training step with train_dtype = bf16)
};

template <int C>
__host__ __device__ inline size_t attn_lds_bytes(int Lpad, int G) {
    size_t f = (size_t)3 * Lpad * (C + 4) + (size_t)C * (Lpad + 4) + (size_t)Lpad * (Lpad + 4) + ((2 * G + 3) & ~3);
    return f * 4;
}

// C = channels (64 here); L <= 96 keys/queries.
template <int C>
__global__ __launch_bounds__(RDMI_THREADS) void attn_mfma_kernel(AttnArgs a) {
    constexpr int RS = C + 4;          // row stride of the [pixel][channel] images
    constexpr int NT = C / 16;         // channel tiles
    static_assert(NT == 4, "one channel tile per wave");
    constexpr int MTMAX = 6;           // Lpad <= 96
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int n = blockIdx.x;
    const int L = a.L, Lpad = a.Lpad, mtiles = Lpad >> 4;
    const int PS = Lpad + 4;           // row stride of P and Vt

    float* Xn = reinterpret_cast<float*>(rdmi_lds);       // [Lpad][RS]  normalised x, later O
    float* Q = Xn + (size_t)Lpad * RS;                     // [Lpad][RS]
    float* K = Q + (size_t)Lpad * RS;                      // [Lpad][RS]
    float* Vt = K + (size_t)Lpad * RS;                     // [C][PS]     V transposed
    float* P = Vt + (size_t)C * PS;                        // [Lpad][PS]
    float* stat = P + (size_t)Lpad * PS;                   // [G][2]

    // ---- load x (rows >= L zero)
    const float* xg = a.io_bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(a.x) + (size_t)n * L * C) : a.x + (size_t)n * L * C;
    for (int i = tid; i < Lpad * (C / 4); i += RDMI_THREADS) {
        const int p = i / (C / 4), c = (i - p * (C / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p < L) v = ldact4(xg, (size_t)p * C + c, a.io_bf16);
        *reinterpret_cast<f32x4*>(Xn + (size_t)p * RS + c) = v;
    }
    __syncthreads();
    // ---- GroupNorm (no activation): two-pass statistics, T lanes per group
    {
        const int G = a.G, Cg = C / G, T = RDMI_THREADS / G;
        const int g = tid / T, sub = tid - g * T;
        const int cnt = Cg * L;
        float sum = 0.f;
        for (int e = sub; e < cnt; e += T) { const int v = e / Cg, cc = e - v * Cg; sum += Xn[(size_t)v * RS + g * Cg + cc]; }
        for (int m = T >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
        const float mean = sum / (float)cnt;
        float sq = 0.f;
        for (int e = sub; e < cnt; e += T) { const int v = e / Cg, cc = e - v * Cg; const float d = Xn[(size_t)v * RS + g * Cg + cc] - mean; sq += d * d; }
        for (int m = T >> 1; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
        if (sub == 0) { stat[2 * g] = mean; stat[2 * g + 1] = 1.0f / sqrtf(sq / (float)cnt + a.eps); }
        __syncthreads();
        for (int i = tid; i < L * (C / 4); i += RDMI_THREADS) {
            const int p = i / (C / 4), c = (i - p * (C / 4)) * 4;
            float* q = Xn + (size_t)p * RS + c;
            f32x4 v = *reinterpret_cast<f32x4*>(q);
            for (int j = 0; j < 4; ++j) {
                const int gg = (c + j) / Cg;
                v[j] = (v[j] - stat[2 * gg]) * stat[2 * gg + 1] * a.gamma[c + j] + a.beta[c + j];
            }
            *reinterpret_cast<f32x4*>(q) = v;
        }
    }
    __syncthreads();

    // ---- q, k, v projections: wave w owns channel tile w for all row tiles
    const int col = wave * 16 + lrow;   // output channel of this lane
    for (int which = 0; which < 3; ++which) {
        f32x4 acc[MTMAX];
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* W = a.wqkv + (size_t)which * (C / 16) * C * 16;
        for (int ch = 0; ch < C / 16; ++ch) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(W + ((size_t)ch * C + col) * 16 + kq * 4);
#pragma unroll
            for (int i = 0; i < MTMAX; ++i) {
                if (i < mtiles) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(Xn + (size_t)(i * 16 + lrow) * RS + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = mfma16(af[j], bf[j], acc[i]);
                }
            }
        }
        const float b = a.bqkv[which * C + col];
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) {
            if (i < mtiles) {
                const int row0 = i * 16 + kq * 4;
                if (which == 2) {
                    f32x4 v = acc[i];
                    for (int r = 0; r < 4; ++r) v[r] += b;
                    *reinterpret_cast<f32x4*>(Vt + (size_t)col * PS + row0) = v;        // transposed: [channel][key]
                } else {
                    float* dst = which == 0 ? Q : K;
                    for (int r = 0; r < 4; ++r) dst[(size_t)(row0 + r) * RS + col] = acc[i][r] + b;
                }
            }
        }
    }
    __syncthreads();

    // ---- scores + softmax: wave w owns query tiles w, w+4
    for (int mt = wave; mt < mtiles; mt += 4) {
        f32x4 sacc[MTMAX];
#pragma unroll
        for (int t = 0; t < MTMAX; ++t) sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < C / 16; ++ch) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(Q + (size_t)(mt * 16 + lrow) * RS + ch * 16 + kq * 4);
#pragma unroll
            for (int t = 0; t < MTMAX; ++t) {
                if (t < mtiles) {
                    const f32x4 bf = *reinterpret_cast<const f32x4*>(K + (size_t)(t * 16 + lrow) * RS + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc[t] = mfma16(af[j], bf[j], sacc[t]);
                }
            }
        }
        // lane holds, for each key tile t, key = t*16 + lrow and query rows mt*16 + kq*4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < MTMAX; ++t)
                if (t < mtiles && t * 16 + lrow < L) { sacc[t][r] *= a.scale; mx = fmaxf(mx, sacc[t][r]); }
            for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < MTMAX; ++t)
                if (t < mtiles) {
                    const float e = (t * 16 + lrow < L) ? __expf(sacc[t][r] - mx) : 0.f;
                    sacc[t][r] = e;
                    sum += e;
                }
            for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
            const float inv = 1.0f / sum;
            const int row = mt * 16 + kq * 4 + r;
#pragma unroll
            for (int t = 0; t < MTMAX; ++t)
                if (t < mtiles) P[(size_t)row * PS + t * 16 + lrow] = sacc[t][r] * inv;
        }
    }
    __syncthreads();

    // ---- O = P . V : wave w owns channel tile w; keys are the K dimension (Lpad, padded keys have P = 0)
    {
        f32x4 acc[MTMAX];
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < mtiles; ++ch) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(Vt + (size_t)col * PS + ch * 16 + kq * 4);
#pragma unroll
            for (int i = 0; i < MTMAX; ++i) {
                if (i < mtiles) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(P + (size_t)(i * 16 + lrow) * PS + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = mfma16(af[j], bf[j], acc[i]);
                }
            }
        }
        // O overwrites Xn (its last reader was the projection phase, two barriers ago)
#pragma unroll
        for (int i = 0; i < MTMAX; ++i)
            if (i < mtiles)
                for (int r = 0; r < 4; ++r) Xn[(size_t)(i * 16 + kq * 4 + r) * RS + col] = acc[i][r];
    }
    __syncthreads();

    // ---- h = NIN_3(O); out = (x + h) * out_scale
    {
        f32x4 acc[MTMAX];
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < C / 16; ++ch) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(a.w3 + ((size_t)ch * C + col) * 16 + kq * 4);
#pragma unroll
            for (int i = 0; i < MTMAX; ++i) {
                if (i < mtiles) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(Xn + (size_t)(i * 16 + lrow) * RS + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = mfma16(af[j], bf[j], acc[i]);
                }
            }
        }
        const float b = a.b3[col];
        float* og = a.io_bf16 ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)n * L * C) : a.out + (size_t)n * L * C;
        // the residual loads of all row tiles are issued before the first store (a load -> wait -> store chain per element otherwise)
        float xr[MTMAX][4];
#pragma unroll
        for (int i = 0; i < MTMAX; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(i * 16 + kq * 4 + r, L - 1);
                xr[i][r] = ldact1(xg, (size_t)row * C + col, a.io_bf16);
            }
#pragma unroll
        for (int i = 0; i < MTMAX; ++i)
            if (i < mtiles)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + kq * 4 + r;
                    if (row < L) stact1(og, (size_t)row * C + col, (xr[i][r] + acc[i][r] + b) * a.out_scale, a.io_bf16);
                }
    }
}
