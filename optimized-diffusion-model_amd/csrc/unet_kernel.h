// Workgroup-resident U-Net: ONE launch runs NCSNpp.forward (RD/models/ncsnpp.py:226-354) for every sample,
// one 512-thread workgroup (8 waves, 2 per SIMD) per sample, with every activation of that sample held in
// the CU's 160 KiB LDS.  Only weights stream in (from L2 / Infinity Cache: 25 MB shared by all workgroups,
// each workgroup reads them once per forward) and only the skip tensors that do not fit round-trip through
// the workgroup's own slice of a global scratch buffer.  This is the MI355X-first shape of the path:
//   * 256 samples (B=128 with classifier-free guidance) == 256 CUs: no tail, no inter-workgroup dependency,
//     so the ~57 kernel boundaries (~5 us each) and per-layer prologues of a layer-by-layer plan disappear;
//   * GroupNorm statistics, SiLU, concat/resize gathers, residuals and the whole 81x81 attention stay in LDS;
//   * all contractions (3x3 convs as 9 row-offset views, NIN, attention, output head) run on the exact-fp32
//     MFMA v_mfma_f32_16x16x4_f32, B operands prefetched through a register ring.
// The kernel is an interpreter of a small host-built op list (csrc/rdmi.hip: build_fused_program), so any
// (ch_mult, num_res_blocks, H, W) the planner can fit in LDS runs without new device code.
#pragma once
#include <cstddef>

#include "common.h"

#define UW_THREADS 512
#define UW_WAVES 8
// weight-ring depths (16-byte loads in flight per wave) of the single-row-tile loops
#ifndef UW_PF_M4
#define UW_PF_M4 4
#endif
#ifndef UW_PF_N1
#define UW_PF_N1 4
#endif
#ifndef UW_PF_KS1
#define UW_PF_KS1 4       // co-operative single-row-tile convs: a wave's K slice is 9-36 steps.  Measured: 4 and 9 equal (15.5 k cycles per 9x128 conv), 18 slower
                          // (18.2 k: the whole 147 KB slice of a CU is requested before the first MFMA and arrives at the L2->CU rate, nothing overlaps)
#endif
#ifndef UW_PRIO
#define UW_PRIO 0
#endif
#ifndef UW_LIGHT_BARRIER
#define UW_LIGHT_BARRIER 0
#endif

// Workgroup barrier between two phases that exchange data through LDS only: wait for this wave's LDS operations, then
// s_barrier.  __syncthreads() additionally drains every outstanding global load (s_waitcnt vmcnt(0)) -- the op-descriptor,
// GroupNorm-parameter and weight prefetches that are deliberately left in flight across op boundaries.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__) && UW_LIGHT_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

enum { FOP_GATHER = 0, FOP_STORE = 1, FOP_GN = 2, FOP_CONV = 3, FOP_ATTN = 4, FOP_LOADTAB = 5, FOP_XCHG = 6 };

struct FPhase {            // one K-phase of a contraction: A rows come from an LDS tensor through a row table
    int lds_off;           // byte offset of the A tensor in LDS
    int rs;                // its row stride in floats
    int nch;               // 16-channel chunks in this phase
    int tab_off;           // byte offset in LDS of this phase's int16 row table [Mpad] (-1 entries = zero row)
    const float* w;        // packed weights [nch][Cout_pad][16] for this phase
};

struct FOp {
    int kind;
    // ---- GATHER: dst[row][0:CA+CB] = concat(A[map(row)], B[row]); sources in LDS or global ([n][rows][C])
    // ---- STORE : global[n][rows][C] = LDS tensor
    // ---- GN    : in-place GroupNorm (+SiLU when act) of an LDS tensor
    // ---- CONV  : dst = scale * (sum_phases A_phase . W_phase + bias [+bias2] [+dense[n]] [+resid])
    // ---- ATTN  : O = softmax(Q K^T * scale) V for LDS-resident Q, K, Vt
    int dst_off, dst_rs, rows, C;                 // generic destination / tensor description
    int CA, CB;                                   // GATHER
    int a_off, a_rs, a_hw, a_mod, a_map_off;      // GATHER source A: a_off < 0 -> global a_g ([n % a_mod][a_hw][CA])
    int b_off, b_rs;                              // GATHER source B: b_off < 0 -> global b_g ([n][rows][CB])
    const float* a_g; const float* b_g;
    float* g_out;                                 // STORE / CONV(dst_kind 2) global destination
    int G, act; float eps;                        // GN (src_off >= 0: read from that LDS tensor, write dst: fused copy)
    int src_off, src_rs;
    int logT, Cg, magic_c4n, magic_Cg; float inv_cnt;   // GN: host-precomputed (no integer/float divisions on the device)
    int logG;                                     // GN: log2(G); multi-sample ops spread the ns * G (sample, group) pairs over the workgroup
    const float* gamma; const float* beta;
    int ntap; int tab_off[9];                     // CONV main phases (taps) share lds/rs/nch, weights contiguous
    FPhase main_ph;                               //   (tab_off of main_ph unused; per-tap tables in tab_off[])
    int nsc; FPhase sc[2];                        // CONV extra phases (NIN shortcut over raw sources)
    int mtiles, Cout, Cout_pad;
    const float* bias; const float* bias2;
    int dense_off;                                // >= 0: add dense[n][dense_off + col]
    int resid_off, resid_rs;                      // >= 0: add LDS tensor [row][col]
    float scale;
    int dst_kind;                                 // 0: LDS [row][col]; 1: LDS transposed [col][row]; 2: global [n][row][col];
                                                  // 3: QKV split: cols [0,C) -> dst, [C,2C) -> dst2 (both [row][col]), [2C,3C) -> dst3 transposed
    int dst2_off, dst3_off, dst3_rs, split_C;
    int q_off, k_off, vt_off, p_off, qk_rs, ps;   // ATTN
    int L, Lpad; float att_scale;
    // CONV with a fused output GroupNorm (dst_kind 0 only): the op that would follow -- GroupNorm(+SiLU) of this conv's
    // output -- runs in the conv's epilogue on the accumulator registers: statistics from per-(wave, k-group) partial sums
    // parked in LDS, one extra barrier, then the normalised values are written to gn_off.  gn_raw: the raw output is ALSO
    // written to dst (the residual / shortcut of the next block still needs it).  gamma / beta / eps / inv_cnt as for GN.
    int gn_off, gn_rs, gn_act, gn_raw, gn_slot_off, gn_nslots;
    // Samples per workgroup (UnetArgs::S > 1, large batches): an op either belongs to ONE sample slot (samp >= 0: the
    // full-resolution sections run per sample, one after the other, in the same LDS) or covers ALL S samples at once
    // (samp < 0: the low-resolution section -- tensors are [S * hw rows][C], row r belongs to sample r >> hw_shift, so every
    // streamed weight fragment feeds S samples).  hw_shift = log2(rows per sample) for multi-sample ops (hw is 4 or 16).
    int samp, hw_shift;
    int qkv1;                                     // CONV dst_kind 3: run as the single-pass q/k/v projection (fconv_qkv)
    // ---- co-operative program (UnetArgs::coop: four workgroups = four CUs share the low-resolution section of their four samples)
    int coop;                                     // CONV: this workgroup computes only ITS quarter of the output columns (member m: column tiles
                                                  // [m * ntiles/4, (m+1) * ntiles/4)) for ALL four samples, K split over the wave groups (fop_conv_coop)
    int a_mstride;                                // GATHER: byte offset added to a_off per member (a member's own rows of a multi-sample tensor)
    // ---- XCHG: all-gather of an LDS tensor between the four members of a group (fop_xchg).  dst_off / dst_rs / rows: the tensor;
    //      C: slice width.  a_hw = 0: every member owns columns [m*C, m*C+C) of all `rows` rows (src_off >= 0: a second tensor of the same
    //      shape travels in the same exchange).  a_hw = 1: every member owns `rows` rows ([m*rows, (m+1)*rows)) of all C columns, its
    //      own block is the single-sample tensor a_off / a_rs.  xidx: index of this exchange in the program (epoch tag and slot parity).
    int xidx;
    // ---- training forward (UnetArgs::train, csrc/train_plan.h): the backward needs every layer's output where the layer plan keeps it
    float* stash;                                 // CONV: global [n][rows][Cout] copy of the finished output (bias, temb, residual, scale applied; before a fused GroupNorm)
    int stash_bf16;                               //   stored as bf16 (train_dtype = bf16)
    int drop_op;                                  // CONV with a fused GroupNorm: >= 0: Dropout_0 on the activated output, mask keyed by (step seed, this layer-plan op index, element)
};

struct UnetArgs {
    const FOp* prog; int nops;
    const short* tabs; int tab_bytes;             // all row tables, copied to LDS offset tab_base at start
    int tab_base;                                 // LDS byte offset of the table region
    int zero_off;                                 // LDS byte offset of a zero row (>= max row bytes)
    int zero_bytes;
    const float* dense; int dense_stride;         // [n][dense_stride] Dense_0 outputs of the embedding kernels
    const float* x_in; int x_mod;                 // network input [x_mod or NB][HW][channels] (GATHER with a_off == -2)
    float* out;                                   // network output [NB][HW][channels]   (CONV dst_kind 2, g_out null)
    int NB;
    int S;                                        // samples per workgroup: workgroup b owns samples b*S .. b*S+S-1 (clamped to NB-1)
    int dbg;                                      // diagnostic ablations (0 in production; results are wrong when set): bit 2 skips GN
                                                  // bodies, bit 3 CONV, bit 4 ATTN, bit 5 GATHER/STORE, bits 6/7 GN statistics / apply,
                                                  // bit 8 conv epilogues, bit 9 conv main loops (scripts/gpu_ablate.py)
    long long* stamps;                            // diagnostic: per-op shader-clock stamps of workgroup 0 (null in production)
    // co-operative program: groups of four workgroups (ids b, b + coop_stride, b + 2*coop_stride, b + 3*coop_stride inside each block of
    // 4*coop_stride ids: with stride 8 the four sit on one XCD under round-robin placement -- speed only, never correctness)
    int coop, coop_stride;
    unsigned long long* xbuf;                     // exchange slots [group][parity][member][xslot] of 8-byte {value, tag} granules (zeroed once)
    int xslot;                                    // granules per member slot
    unsigned epoch_base;                          // tag of exchange x of this launch = epoch_base + x + 1 (host: advanced by the program's exchange count per launch)
    int out_elems;                                // floats per sample of the network output
    float drop_p; const unsigned long long* seed_dev;   // training forward: dropout probability and the step's seed (device word)
    int* coop_err;                                // set to 1 by a workgroup whose bounded wait gave up (its output sample is then NaN)
    int coop_break;                               // test hook (RDMI_COOP_TEST_BREAK=1): member 3 of group 0 withholds its first publication, so the others' bounded wait must fire
};

// ---------------------------------------------------------------------------------------------------------
// Op descriptors live in REGISTERS: lane i of every wave holds 32-bit word i of the current FOp (two VGPRs cover
// the struct), loaded straight from global one op ahead.  A field is one v_readlane into an SGPR: no LDS or global
// latency on the op-transition path.
struct OpW { int w0, w1; };
static_assert(sizeof(FOp) <= 512, "FOp must fit two VGPRs per wave");
__device__ __forceinline__ int opw_at(const OpW& r, int word) {
    return word < 64 ? __builtin_amdgcn_readlane(r.w0, word) : __builtin_amdgcn_readlane(r.w1, word - 64);
}
#define OPI(r, field) opw_at(r, (int)(offsetof(FOp, field) / 4))
#define OPF(r, field) __builtin_bit_cast(float, OPI(r, field))
#define OPP(r, T, field) \
    reinterpret_cast<T*>((unsigned long long)(unsigned)opw_at(r, (int)(offsetof(FOp, field) / 4)) | ((unsigned long long)(unsigned)opw_at(r, (int)(offsetof(FOp, field) / 4) + 1) << 32))
__device__ __forceinline__ OpW opw_load(const FOp* op, int lane) {
    OpW r;
    constexpr int NW = (int)(sizeof(FOp) / 4);
    const int* g = reinterpret_cast<const int*>(op);
    r.w0 = lane < NW ? __builtin_bit_cast(int, ldg1(reinterpret_cast<const float*>(g + lane))) : 0;
    r.w1 = lane + 64 < NW ? __builtin_bit_cast(int, ldg1(reinterpret_cast<const float*>(g + lane + 64))) : 0;
    return r;
}

__device__ __forceinline__ float* lds_f(int off) { return reinterpret_cast<float*>(rdmi_lds + off); }

// sample index of row `row` of an op: slot `samp` for single-sample ops, row >> hw_shift for multi-sample ops; the global
// sample number is clamped to the batch (a tail workgroup recomputes the last sample in its unused slots)
__device__ __forceinline__ int samp_of(int samp, int hw_shift, int row) { return samp >= 0 ? samp : (row >> hw_shift); }

template <bool MS>
__device__ __forceinline__ void fop_gather(const OpW& w, const UnetArgs& u, int n0, int ncap, int member, int tid) {
    struct { int C, rows, dst_off, dst_rs, CA, CB, a_off, a_rs, a_hw, a_mod, a_map_off, b_off, b_rs, samp, hw_shift; const float* a_g; const float* b_g; } o;
    o.C = OPI(w, C); o.rows = OPI(w, rows); o.dst_off = OPI(w, dst_off); o.dst_rs = OPI(w, dst_rs); o.CA = OPI(w, CA); o.CB = OPI(w, CB);
    o.a_off = OPI(w, a_off); o.a_rs = OPI(w, a_rs); o.a_hw = OPI(w, a_hw); o.a_mod = OPI(w, a_mod); o.a_map_off = OPI(w, a_map_off);
    if (MS && o.a_off >= 0) o.a_off += member * OPI(w, a_mstride);        // co-operative program: this member's rows of a multi-sample tensor
    o.b_off = OPI(w, b_off); o.b_rs = OPI(w, b_rs); o.a_g = OPP(w, const float, a_g); o.b_g = OPP(w, const float, b_g);
    o.samp = MS ? OPI(w, samp) : 0; o.hw_shift = MS ? OPI(w, hw_shift) : 0;
    const int Cd = o.C;                               // padded total channels (multiple of 4)
    const int c4n = Cd >> 2;
    const int total = o.rows * c4n;
    float* dst = lds_f(o.dst_off);
    const short* map = o.a_map_off >= 0 ? reinterpret_cast<const short*>(rdmi_lds + o.a_map_off) : nullptr;
    const int dpv = UW_THREADS / c4n, dc4 = UW_THREADS - dpv * c4n;
    int row = tid / c4n, c4 = tid - row * c4n;
    const bool from_x = o.a_off == -2;
    const float* ag = from_x ? u.x_in : o.a_g;
    const int amod = from_x ? u.x_mod : o.a_mod;
    const int hw = o.samp >= 0 ? o.rows : (1 << o.hw_shift);          // destination rows per sample
    for (int i = tid; i < total; i += UW_THREADS) {
        const int c = c4 << 2;
        const int sl = samp_of(o.samp, o.hw_shift, row);             // sample slot and pixel of this destination row
        const int px = o.samp >= 0 ? row : row - (sl << o.hw_shift);
        const int n = min(n0 + sl, ncap - 1);
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (c < o.CA) {
            const int srow = map ? map[px] : px;
            if (o.a_off >= 0) {                                      // LDS source: multi-sample tensors hold a_hw rows per sample
                val = *reinterpret_cast<const f32x4*>(lds_f(o.a_off) + (size_t)((o.samp >= 0 ? 0 : sl * o.a_hw) + srow) * o.a_rs + c);
            } else {
                const int nA = amod > 0 ? n % amod : n;
                const float* p = ag + ((size_t)nA * o.a_hw + srow) * o.CA + c;
                if ((o.CA & 3) == 0) val = ldg4(p);
                else
                    for (int j = 0; j < 4; ++j)
                        if (c + j < o.CA) val[j] = ldg1(p + j);
            }
        } else if (c < o.CA + o.CB) {
            if (o.b_off >= 0) val = *reinterpret_cast<const f32x4*>(lds_f(o.b_off) + (size_t)row * o.b_rs + (c - o.CA));
            else val = ldg4(o.b_g + ((size_t)n * hw + px) * o.CB + (c - o.CA));
        }
        *reinterpret_cast<f32x4*>(dst + (size_t)row * o.dst_rs + c) = val;
        row += dpv; c4 += dc4;
        if (c4 >= c4n) { c4 -= c4n; ++row; }
    }
}

template <bool MS>
__device__ __forceinline__ void fop_store(const OpW& w, const UnetArgs& u, int n0, int ncap, int tid) {
    struct { int C, rows, dst_off, dst_rs, samp, hw_shift; float* g_out; } o;
    o.C = OPI(w, C); o.rows = OPI(w, rows); o.dst_off = OPI(w, dst_off); o.dst_rs = OPI(w, dst_rs); o.g_out = OPP(w, float, g_out);
    o.samp = MS ? OPI(w, samp) : 0; o.hw_shift = MS ? OPI(w, hw_shift) : 0;
    const int c4n = o.C >> 2;
    const int total = o.rows * c4n;
    const float* src = lds_f(o.dst_off);
    const int hw = o.samp >= 0 ? o.rows : (1 << o.hw_shift);
    const int dpv = UW_THREADS / c4n, dc4 = UW_THREADS - dpv * c4n;
    int row = tid / c4n, c4 = tid - row * c4n;
    for (int i = tid; i < total; i += UW_THREADS) {
        const int sl = samp_of(o.samp, o.hw_shift, row);
        const int px = o.samp >= 0 ? row : row - (sl << o.hw_shift);
        const int n = min(n0 + sl, ncap - 1);        // duplicate slots of a tail workgroup store the same values to the same place
        stg4(o.g_out + ((size_t)n * hw + px) * o.C + (c4 << 2), *reinterpret_cast<const f32x4*>(src + (size_t)row * o.dst_rs + (c4 << 2)));
        row += dpv; c4 += dc4;
        if (c4 >= c4n) { c4 -= c4n; ++row; }
    }
}

// sum over the 4 lanes of a quad (columns 4q..4q+3 of a 16-column tile): two VALU-DPP steps
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); return v; }
#else
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }
#endif

// sum over aligned groups of 2^logT lanes (4, 8, 16, 32 or 64): DPP row reductions (common.h), no LDS-crossbar shuffles
__device__ __forceinline__ float group_sum_rt(float v, int logT, int lane) {
    switch (logT) {
        case 6: return group64_sum(v);
        case 5: return group32_sum(v, lane);
        case 4: return row16_sum(v);
        case 3: return row8_sum(v);
        default: return quad_sum(v);       // 4 lanes per group (multi-sample ops)
    }
}

// GroupNorm (+SiLU) of an LDS tensor, in place or from a source tensor (fused copy).  Two-pass statistics with the
// group's values held in REGISTERS between the passes (one LDS read per element), T lanes per group reduced by
// xor-shuffles; the affine parameters of this work-item's fixed channel quad are prefetched by the caller one op
// ahead (pgm/pbt) so their global latency never sits on the op-transition path.
template <bool MS>
__device__ __forceinline__ void fop_gn(const OpW& w, float* stat, int tid, f32x4 pgm, f32x4 pbt, int dbg = 0) {
    float* X0 = lds_f(OPI(w, dst_off));
    const int src_off = OPI(w, src_off);
    const float* S0 = src_off >= 0 ? lds_f(src_off) : X0;
    const int srs = src_off >= 0 ? OPI(w, src_rs) : OPI(w, dst_rs);
    const int o_C = OPI(w, C), Cg = OPI(w, Cg), rs = OPI(w, dst_rs), o_act = OPI(w, act);
    // multi-sample ops (samp < 0): the tensor is ns samples of hw rows; statistics are per sample ([ns][2 * G] in `stat`)
    const int o_samp = MS ? OPI(w, samp) : 0, hw_shift = MS ? OPI(w, hw_shift) : 0;
    const int o_rows = o_samp >= 0 ? OPI(w, rows) : (1 << hw_shift);
    const int ns = o_samp >= 0 ? 1 : (OPI(w, rows) >> hw_shift);
    const int o_G = OPI(w, G);
    const float o_eps = OPF(w, eps);
    const int logT = OPI(w, logT), T = 1 << logT;     // 16 or 32 lanes per group
    const int g = tid >> logT, sub = tid & (T - 1);
    const float inv_cnt = OPF(w, inv_cnt);
    const int mg_c4n = OPI(w, magic_c4n), mg_Cg = OPI(w, magic_Cg);
    // this lane's share of the group: rows sub, sub+T, ... (<= 6 rows for 96 pixels at T = 16), Cg <= 8 channels
    constexpr int MAXR = 6;
    if (!(dbg & 64)) {
    {
    // multi-sample ops: the ns * G (sample, group) pairs are spread over the workgroup (vg = sm * G + gq, T = 512 / (ns * G) lanes
    // each): all samples' statistics form in ONE pass instead of one pass per sample
    const int sm = MS ? (g >> OPI(w, logG)) : 0, gq = MS ? (g & (o_G - 1)) : g;
    const float* base = S0 + (size_t)sm * o_rows * srs + gq * Cg;
    f32x4 v0[MAXR], v1[MAXR];
    float sum = 0.f;
    // Branch-free over rows: every lane loads its (clamped) rows back to back -- all reads in flight before the first
    // wait -- and rows beyond the tensor / channels beyond the group are masked out of the sums.  Only the group width
    // (Cg = 4, 8: one or two 16-byte reads; Cg = 6: three 8-byte reads, groups start at 8-byte boundaries) branches,
    // and that branch is wave-uniform.
    float rw[MAXR];                                    // 1 for a real row of this lane, else 0
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int v = sub + k * T;
        rw[k] = v < o_rows ? 1.f : 0.f;
        const float* p = base + (size_t)min(v, o_rows - 1) * srs;
        if ((Cg & 3) == 0) {
            v0[k] = *reinterpret_cast<const f32x4*>(p);
            v1[k] = Cg > 4 ? *reinterpret_cast<const f32x4*>(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            typedef float f32x2 __attribute__((vector_size(8)));
            const f32x2 a0 = *reinterpret_cast<const f32x2*>(p), a1 = *reinterpret_cast<const f32x2*>(p + 2), a2 = *reinterpret_cast<const f32x2*>(p + 4);
            v0[k] = f32x4{a0[0], a0[1], a1[0], a1[1]};
            v1[k] = f32x4{a2[0], a2[1], 0.f, 0.f};
        }
    }
#pragma unroll
    for (int k = 0; k < MAXR; ++k)
        sum += rw[k] * ((v0[k][0] + v0[k][1]) + (v0[k][2] + v0[k][3]) + (v1[k][0] + v1[k][1]) + (v1[k][2] + v1[k][3]));
    sum = group_sum_rt(sum, logT, tid & 63);
    const float mean = sum * inv_cnt;
    float sq = 0.f;
    const float m4 = Cg > 4 ? 1.f : 0.f, m6 = Cg > 6 ? 1.f : 0.f;      // which of the second quad's channels exist
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        float q = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { const float d = v0[k][cc] - mean; q += d * d; }
        const float d4 = v1[k][0] - mean, d5 = v1[k][1] - mean, d6 = v1[k][2] - mean, d7 = v1[k][3] - mean;
        q += m4 * (d4 * d4 + d5 * d5) + m6 * (d6 * d6 + d7 * d7);
        sq += rw[k] * q;
    }
    sq = group_sum_rt(sq, logT, tid & 63);
    if (sub == 0) { stat[g * 2] = mean; stat[g * 2 + 1] = 1.0f / sqrtf(sq * inv_cnt + o_eps); }
    }
    }
    // fixed channel quad per work-item: rows advance by rstep; work-items beyond rstep*c4n idle (C = 192)
    const int c4n = o_C >> 2;
    const int rstep = (UW_THREADS * mg_c4n) >> 16;      // floor(512 / c4n)
    const int r0 = (tid * mg_c4n) >> 16, c = (tid - r0 * c4n) << 2;
    const bool active = r0 < rstep;
    lds_barrier();
    if (active && !(dbg & 128)) {
        if (!MS || ns == 1) {
            f32x4 mu, rstd;
            for (int j = 0; j < 4; ++j) { const int gg = ((c + j) * mg_Cg) >> 16; mu[j] = stat[2 * gg]; rstd[j] = stat[2 * gg + 1] * pgm[j]; }
            for (int row = r0; row < o_rows; row += rstep) {
                float* p = X0 + (size_t)row * rs + c;
                f32x4 val = *reinterpret_cast<const f32x4*>(S0 + (size_t)row * srs + c);
                for (int j = 0; j < 4; ++j) {
                    const float y = (val[j] - mu[j]) * rstd[j] + pbt[j];
                    val[j] = o_act ? silu_f(y) : y;
                }
                *reinterpret_cast<f32x4*>(p) = val;
            }
        } else {                       // all samples' rows in one sweep; a row's statistics are its sample's
            const int total = ns * o_rows;
            int gq[4];
            for (int j = 0; j < 4; ++j) gq[j] = ((c + j) * mg_Cg) >> 16;
            for (int row = r0; row < total; row += rstep) {
                const int sb = (row >> hw_shift) * o_G;
                float* p = X0 + (size_t)row * rs + c;
                f32x4 val = *reinterpret_cast<const f32x4*>(S0 + (size_t)row * srs + c);
                for (int j = 0; j < 4; ++j) {
                    const float y = (val[j] - stat[2 * (sb + gq[j])]) * (stat[2 * (sb + gq[j]) + 1] * pgm[j]) + pbt[j];
                    val[j] = o_act ? silu_f(y) : y;
                }
                *reinterpret_cast<f32x4*>(p) = val;
            }
        }
    }
}

// Affine parameters of the NEXT op when it is a GroupNorm: this work-item's fixed channel quad (see fop_gn).
__device__ __forceinline__ void gn_prefetch(const OpW& nx, int tid, f32x4& pgm, f32x4& pbt) {
    if (OPI(nx, kind) != FOP_GN) return;
    const int c4n = OPI(nx, C) >> 2, mg = OPI(nx, magic_c4n);
    const int r0 = (tid * mg) >> 16, c = (tid - r0 * c4n) << 2;
    const float* gp = OPP(nx, const float, gamma);      // field reads are wave-uniform: keep them outside divergent code
    const float* bp = OPP(nx, const float, beta);
    if (r0 < ((UW_THREADS * mg) >> 16)) { pgm = ldg4(gp + c); pbt = ldg4(bp + c); }
}

// A-row byte offset for one table entry (-1 -> the shared zero row)
__device__ __forceinline__ int arow(const short* tab, int m, int lds_off, int rs, int zero_off) {
    const int r = tab[m];
    return r < 0 ? zero_off : lds_off + r * rs * 4;
}

// One wave's share of a CONV op, main loop: NMT row tiles (starting at tile mt0, stride WM) x one column tile -> acc[0..NMT-1].
//  * weights stream through a PF-deep register ring of straight-line global loads (the compiler waits with
//    vmcnt(PF-1), never draining the ring); the step count is padded up to a multiple of PF with steps whose A
//    rows are the zero row, so there is no tail code (instruction-cache footprint matters: the whole interpreter
//    must stay resident in the 64 KiB I-cache or every op transition refetches cold code);
//  * A fragments are read from LDS one step ahead of the MFMAs that consume them.
// The epilogue (fconv_epi) is shared by every instantiation: one copy of that code in the kernel.
// KS (co-operative program): the K steps are dealt round-robin to four wave groups -- this wave takes steps kg, kg + 4, ... of the
// (tap, 16-channel chunk) sequence (the chunk count is a multiple of 4, so it is chunk kg, kg + 4, ... of every tap); acc[] are
// PARTIAL sums that fop_conv_coop adds up.
template <bool DIAG, int NMT, int PF, bool M4 = false, bool LM4 = false, bool KS = false>
__device__ __forceinline__ void fconv_main(const OpW& w, const UnetArgs& u, int mt0, int WM, int nt, int lane, long long* fine, f32x4 (&acc)[4], int kg = 0) {
    const int lrow = lane & 15, kq = lane >> 4;
    if (DIAG && fine) fine[0] = clock64();
    const int o_Cout_pad = OPI(w, Cout_pad), o_ntap = OPI(w, ntap);
    const int m_lds = OPI(w, main_ph.lds_off), m_rs = OPI(w, main_ph.rs), nch = OPI(w, main_ph.nch);
    const float* m_w = OPP(w, const float, main_ph.w);
    const int tab_word = (int)(offsetof(FOp, tab_off) / 4);
    const int zero_off = u.zero_off;
    const int col = nt * 16 + lrow;
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NMT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int mrow[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)      // M4: images of <= 4 pixels; LM4: the LAST row tile has <= 4 real rows (81 = 5 x 16 + 1)
        mrow[i] = (mt0 + i * WM) * 16 + ((M4 || (LM4 && i == NMT - 1)) ? (lane & 3) : lrow);
    {
        const int nsteps = KS ? (o_ntap * nch) >> 2 : o_ntap * nch;
        const int npad = (DIAG && (u.dbg & 512)) ? 0 : ((nsteps + PF - 1) / PF) * PF;     // ablation: no main loop
        const size_t bstride = (size_t)o_Cout_pad * (KS ? 64 : 16);
        const float* Wl = m_w + (size_t)col * 16 + kq * 4 + (KS ? (size_t)kg * o_Cout_pad * 16 : 0);
        f32x4 ring[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) ring[p] = ldg4(Wl + (size_t)min(p, nsteps - 1) * bstride);
        int ph = 0, ch = KS ? kg : 0;
        // Row offsets of the current tap (abase), of the next one (anext) and -- still as the raw int16 table entry -- of the one after
        // (araw).  A tap transition is then register moves plus address arithmetic on a table entry that was READ A WHOLE TAP AGO: the
        // LDS read issued at the transition is not waited for until the next transition.  (With the lookup and its s_waitcnt inside the
        // transition, every wave stalled for two dependent LDS round trips 8 times per 3x3 conv -- and the two waves of a SIMD reach
        // their transitions together, so the matrix pipe idled: ~10 % of a 9x9 conv's main phase.)
        static_assert(offsetof(FOp, tab_off) / 4 + 9 <= 64, "tap tables must sit in the first descriptor register");
        auto tab_at = [&](int t) { return reinterpret_cast<const short*>(rdmi_lds + __builtin_amdgcn_readlane(w.w0, tab_word + t)); };
        auto row_addr = [&](int r) { return (r < 0 ? zero_off : m_lds + r * m_rs * 4) + kq * 16; };
        int abase[NMT], anext[NMT], araw[NMT];
#pragma unroll
        for (int i = 0; i < NMT; ++i) {
            abase[i] = row_addr(tab_at(0)[mrow[i]]);
            anext[i] = o_ntap > 1 ? row_addr(tab_at(1)[mrow[i]]) : zero_off + kq * 16;
            araw[i] = o_ntap > 2 ? (int)tab_at(2)[mrow[i]] : -1;
        }
        f32x4 afn[NMT];
#pragma unroll
        for (int i = 0; i < NMT; ++i) afn[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + abase[i] + (KS ? kg * 64 : 0));
        if (DIAG && fine) fine[1] = clock64();
        for (int q = 0; q < npad; q += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                f32x4 af[NMT];
#pragma unroll
                for (int i = 0; i < NMT; ++i) af[i] = afn[i];
                // advance to the next step and issue its A reads before this step's MFMAs
                bool wrap;
                if (KS) { ch += 4; wrap = ch >= nch; if (wrap) ch -= nch; }
                else { wrap = ++ch == nch; if (wrap) ch = 0; }
                if (wrap) {
                    ++ph;
#pragma unroll
                    for (int i = 0; i < NMT; ++i) {
                        abase[i] = anext[i];
                        anext[i] = ph + 1 < o_ntap ? row_addr(araw[i]) : zero_off + kq * 16;     // past the last tap: padding steps contribute 0
                    }
                    if (ph + 2 < o_ntap) {
#pragma unroll
                        for (int i = 0; i < NMT; ++i) araw[i] = tab_at(ph + 2)[mrow[i]];
                    }
                }
                if (!(DIAG && (u.dbg & 2048))) {     // (ablation bit: no A-fragment reads in the loop)
#pragma unroll
                    for (int i = 0; i < NMT; ++i) afn[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + abase[i] + ch * 64);
                }
                RDMI_SCHED_FENCE();   // keep those LDS reads ABOVE this step's MFMAs (the scheduler otherwise sinks them below
                                      // and the next step starts with a full lgkmcnt(0) wait on a just-issued read)
                if (M4) {           // <= 4 rows: the 4x4x1 multi-block form wastes no rows (13 vs 32 cycles per MFMA, see mfma4)
                    acc[0] = mfma4(af[0][0], ring[p][0], acc[0]); acc2 = mfma4(af[0][1], ring[p][1], acc2);
                    acc[0] = mfma4(af[0][2], ring[p][2], acc[0]); acc2 = mfma4(af[0][3], ring[p][3], acc2);
                } else if (NMT == 1) {     // single row tile: alternate two accumulators so the MFMAs are not a dependent chain
                    acc[0] = mfma16(af[0][0], ring[p][0], acc[0]); acc2 = mfma16(af[0][1], ring[p][1], acc2);
                    acc[0] = mfma16(af[0][2], ring[p][2], acc[0]); acc2 = mfma16(af[0][3], ring[p][3], acc2);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < NMT; ++i)
                            acc[i] = (LM4 && i == NMT - 1) ? mfma4(af[i][j], ring[p][j], acc[i]) : mfma16(af[i][j], ring[p][j], acc[i]);
                }
                if (!(DIAG && (u.dbg & 1024))) ring[p] = ldg4(Wl + (size_t)min(q + p + PF, nsteps - 1) * bstride);     // (ablation bit: no weight reloads)
            }
        }
    }
    if (NMT == 1) acc[0] += acc2;
    if (M4 || LM4) {        // sum the four k groups: afterwards every lane holds D[row r][col lrow], the kq == 0 lanes' share
        constexpr int ti = M4 ? 0 : NMT - 1;   // of the 16x16x4 result layout, so the epilogue is unchanged (other lanes' rows are >= 4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[ti][r];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[ti][r] = v;
        }
    }
    if (DIAG && fine) fine[4] = clock64();
}

// Main loop, TAP-MAJOR form, for ops whose 16-channel chunk count per tap is a multiple of 4 (every conv of the model but the
// first, which contracts 16 channels).  Same arithmetic, same order of accumulation and the same operands as fconv_main; what differs
// is the control code around the MFMAs.  fconv_main decides per step whether the tap changes (two branches, a 64-bit multiply for the
// weight address, a clamp): ~20 scalar instructions and two taken branches between the last MFMA of a step and the first of the
// next.  scripts/micro/mfma_lds.hip shows what that costs: two waves per SIMD running 12 MFMA + 3 ds_read_b128 + 1 global_load per
// step keep the matrix pipe 89 % busy when the steps follow each other directly and 77-82 % with that inter-step section -- the
// section of one wave does not hide under the other wave's MFMAs.  Here four steps (one turn of the weight ring) are straight-line
// code; a tap can only end at a group boundary, where the next A addresses are a select between "next chunk" and "next tap"; the
// weight address is a wave-uniform byte offset advanced by a scalar add + min.
// PF: steps per straight-line group = depth of the weight ring (4, or 8 where a wave has a single row tile: its steps are only four
// MFMAs long, so four steps of lookahead are ~1 k cycles -- no more than an L2 round trip under load); the chunk count per tap must be a multiple of PF.
template <bool DIAG, int NMT, bool LM4 = false, int PF = 4, bool M4 = false>
__device__ __forceinline__ void fconv_main_t(const OpW& w, const UnetArgs& u, int mt0, int WM, int nt, int lane, long long* fine, f32x4 (&acc)[4]) {
    const int lrow = lane & 15, kq = lane >> 4;
    if (DIAG && fine) fine[0] = clock64();
    const int o_Cout_pad = OPI(w, Cout_pad), o_ntap = OPI(w, ntap);
    const int m_lds = OPI(w, main_ph.lds_off), m_rs = OPI(w, main_ph.rs), nch = OPI(w, main_ph.nch);
    const WBuf wb = wbuf_make(OPP(w, const float, main_ph.w));
    const int tab_word = (int)(offsetof(FOp, tab_off) / 4);
    const int zero_off = u.zero_off;
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NMT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int mrow[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i) mrow[i] = (mt0 + i * WM) * 16 + ((LM4 && i == NMT - 1) ? 0 : M4 ? (lane & 3) : lrow);     // M4: an image of <= 4 pixels (4x4x1 MFMA form, see mfma4)
    // LM4: the last row tile holds ONE real row (81 = 5 x 16 + 1).  It is not worth a matrix instruction at all: every lane reads that
    // row's A fragment (an LDS broadcast) and multiplies it with the weight fragment it already holds for the full tiles -- four VALU
    // fma per step into element 0 of the tile's accumulator (the k-ordered fma chain an MFMA would run, per k quarter), summed over
    // the four k quarters after the loop.  The 4x4x1 MFMA form used before cost 13 of the matrix pipe's cycles per instruction, 17 %
    // of the pipe time of the wave that owns the tile, for one row in eighty-one.
    const int nsteps = o_ntap * nch;
    const unsigned bstride = (unsigned)o_Cout_pad * 64u;                       // bytes per step
    const unsigned wend = (unsigned)(nsteps - 1) * bstride;
    const unsigned lane_w = (unsigned)((nt * 16 + lrow) * 16 + kq * 4) * 4u;   // this lane's bytes inside a step's block
    f32x4 ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = wbuf_load4(wb, lane_w, min((unsigned)p * bstride, wend));
    unsigned woff = min((unsigned)PF * bstride, wend);                                   // byte offset of the next fragment to request
    static_assert(offsetof(FOp, tab_off) / 4 + 9 <= 64, "tap tables must sit in the first descriptor register");
    auto tab_at = [&](int t) { return reinterpret_cast<const short*>(rdmi_lds + __builtin_amdgcn_readlane(w.w0, tab_word + t)); };
    const int rs4 = m_rs * 4, lds_kq = m_lds + kq * 16, zero_kq = zero_off + kq * 16;
    auto row_addr = [&](int r) { return r < 0 ? zero_kq : mad_u24(r, rs4, lds_kq); };      // (row index and row bytes are far below 2^24)
    int acur[NMT], anext[NMT], araw[NMT];      // A byte addresses of the current chunk group, of the next tap, raw table entry of the tap after
#pragma unroll
    for (int i = 0; i < NMT; ++i) {
        acur[i] = row_addr(tab_at(0)[mrow[i]]);
        anext[i] = o_ntap > 1 ? row_addr(tab_at(1)[mrow[i]]) : zero_kq;
        araw[i] = o_ntap > 2 ? (int)tab_at(2)[mrow[i]] : -1;
    }
    f32x4 afn[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i) afn[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + acur[i]);
    if (DIAG && fine) fine[1] = clock64();
    const int gpt = nch / PF;                  // groups of PF steps per tap
    int gl = gpt, t = 0;
    const int ngroups = (DIAG && (u.dbg & 512)) ? 0 : o_ntap * gpt;
    for (int g = 0; g < ngroups; ++g) {
        const bool last = --gl == 0;           // this group ends its tap
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            f32x4 af[NMT];
#pragma unroll
            for (int i = 0; i < NMT; ++i) af[i] = afn[i];
            // the next step's A fragment of tile i: next chunk of this tap, or (last step of a group) the first chunk of whatever follows
            auto read_next = [&](int i) {
                if (p < PF - 1) afn[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + acur[i] + (p + 1) * 64);
                else { acur[i] = last ? anext[i] : acur[i] + PF * 64; afn[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + acur[i]); }
            };
            // The LDS reads are dealt out BETWEEN the MFMA groups and the weight request comes last: memory instructions issued in a
            // cluster ahead of the MFMAs cost matrix-pipe time, interleaved ones hide (scripts/micro/mfma_lds.hip: 814 -> 790 cycles
            // per step pair).  The fences pin that order against the machine scheduler.
            if (M4) {
                acc[0] = mfma4(af[0][0], ring[p][0], acc[0]); acc2 = mfma4(af[0][1], ring[p][1], acc2);
                RDMI_SCHED_FENCE(); read_next(0); RDMI_SCHED_FENCE();
                acc[0] = mfma4(af[0][2], ring[p][2], acc[0]); acc2 = mfma4(af[0][3], ring[p][3], acc2);
            } else if (NMT == 1) {
                acc[0] = mfma16(af[0][0], ring[p][0], acc[0]); acc2 = mfma16(af[0][1], ring[p][1], acc2);
                RDMI_SCHED_FENCE(); read_next(0); RDMI_SCHED_FENCE();
                acc[0] = mfma16(af[0][2], ring[p][2], acc[0]); acc2 = mfma16(af[0][3], ring[p][3], acc2);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < NMT; ++i) {
                        if (LM4 && i == NMT - 1) acc[i][0] = __builtin_fmaf(af[i][j], ring[p][j], acc[i][0]);
                        else acc[i] = mfma16(af[i][j], ring[p][j], acc[i]);
                    }
                    RDMI_SCHED_FENCE();
                    if (j < NMT) read_next(j);
                    RDMI_SCHED_FENCE();
                }
            }
            ring[p] = wbuf_load4(wb, lane_w, woff);
            woff = min(woff + bstride, wend);
        }
        if (last) {                            // tap transition, branch-free: register work on a table entry read a whole tap ago + one LDS read issued
            gl = gpt; ++t;                     // (araw is -1 = "zero row" once the taps run out: the clamped re-read of the last table is discarded)
            const bool more = t + 2 < o_ntap;
            const short* const tb = tab_at(min(t + 2, o_ntap - 1));
#pragma unroll
            for (int i = 0; i < NMT; ++i) {
                anext[i] = row_addr(araw[i]);
                const int r = tb[mrow[i]];
                araw[i] = more ? r : -1;
            }
        }
    }
    if (NMT == 1) acc[0] += acc2;
    if (M4) {                                  // sum the four k quarters: every lane then holds D[row r][col lrow] (see fconv_main)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[0][r];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[0][r] = v;
        }
    }
    if (LM4) {                                 // every lane ends with D[row 80][col lrow] in element 0 (rows 81..83 do not exist: 0)
        float v = acc[NMT - 1][0];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[NMT - 1][0] = v;
    }
    if (DIAG && fine) fine[4] = clock64();
}

// Tap-major main loop of the CO-OPERATIVE convs (one row tile, K dealt to four wave groups: wave group kg takes 16-channel chunks
// kg, kg + 4, ... of every tap -- SPT = nch / 4 steps per tap, 2 at 128 input channels, 4 at 256).  A straight-line group is four
// steps = 4 / SPT taps; taps beyond the last one (9 taps in groups of two) read the zero row and a clamped weight block: exact zeros.
// acc[0] is this wave's PARTIAL sum (fop_conv_coop adds the four).
template <bool DIAG, int SPT>
__device__ __forceinline__ void fconv_main_ks(const OpW& w, const UnetArgs& u, int nt, int lane, long long* fine, f32x4 (&acc)[4], int kg) {
    constexpr int TPG = 4 / SPT;               // taps per group
    const int lrow = lane & 15, kq = lane >> 4;
    if (DIAG && fine) fine[0] = clock64();
    const int o_Cout_pad = OPI(w, Cout_pad), o_ntap = OPI(w, ntap);
    const int m_lds = OPI(w, main_ph.lds_off), m_rs = OPI(w, main_ph.rs);
    const WBuf wb = wbuf_make(OPP(w, const float, main_ph.w));
    const int tab_word = (int)(offsetof(FOp, tab_off) / 4);
    const int zero_off = u.zero_off;
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = o_ntap * SPT;
    const unsigned bstride = (unsigned)o_Cout_pad * 256u;                      // bytes between this wave's steps (4 chunks)
    const unsigned w0 = (unsigned)kg * o_Cout_pad * 64u, wend = w0 + (unsigned)(nsteps - 1) * bstride;
    const unsigned lane_w = (unsigned)((nt * 16 + lrow) * 16 + kq * 4) * 4u;
    f32x4 ring[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ring[p] = wbuf_load4(wb, lane_w, min(w0 + (unsigned)p * bstride, wend));
    unsigned woff = min(w0 + 4u * bstride, wend);
    auto tab_at = [&](int t) { return reinterpret_cast<const short*>(rdmi_lds + __builtin_amdgcn_readlane(w.w0, tab_word + t)); };
    const int rs4 = m_rs * 4, lds_kq = m_lds + kg * 64 + kq * 16, zero_kq = zero_off + kg * 64 + kq * 16;      // (the zero row is as long as the widest tensor row)
    auto row_addr = [&](int r) { return r < 0 ? zero_kq : mad_u24(r, rs4, lds_kq); };
    int a[TPG], an[TPG], araw[TPG];            // this group's taps, the next group's, raw table entries of the group after
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        a[k] = k < o_ntap ? row_addr(tab_at(k)[lrow]) : row_addr(-1);
        an[k] = TPG + k < o_ntap ? row_addr(tab_at(TPG + k)[lrow]) : row_addr(-1);
        araw[k] = 2 * TPG + k < o_ntap ? (int)tab_at(2 * TPG + k)[lrow] : -1;
    }
    f32x4 afn = *reinterpret_cast<const f32x4*>(rdmi_lds + a[0]);
    if (DIAG && fine) fine[1] = clock64();
    const int ngroups = (DIAG && (u.dbg & 512)) ? 0 : (o_ntap + TPG - 1) / TPG;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 af = afn;
            acc[0] = mfma16(af[0], ring[p][0], acc[0]); acc2 = mfma16(af[1], ring[p][1], acc2);
            RDMI_SCHED_FENCE();
            if (p < 3) afn = *reinterpret_cast<const f32x4*>(rdmi_lds + a[(p + 1) / SPT] + ((p + 1) % SPT) * 256);
            else {
#pragma unroll
                for (int k = 0; k < TPG; ++k) a[k] = an[k];
                afn = *reinterpret_cast<const f32x4*>(rdmi_lds + a[0]);
            }
            RDMI_SCHED_FENCE();
            acc[0] = mfma16(af[2], ring[p][2], acc[0]); acc2 = mfma16(af[3], ring[p][3], acc2);
            ring[p] = wbuf_load4(wb, lane_w, woff);
            woff = min(woff + bstride, wend);
        }
        // group transition, branch-free: addresses from table entries read a group ago, one LDS read per tap issued
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            const int t = (g + 3) * TPG + k;
            an[k] = row_addr(araw[k]);
            const int r = tab_at(min(t, o_ntap - 1))[lrow];
            araw[k] = t < o_ntap ? r : -1;
        }
    }
    acc[0] += acc2;
    if (DIAG && fine) fine[4] = clock64();
}

// Shared CONV epilogue for one wave's nmt row tiles x one column tile.  Destinations each get their own (wave-uniform) branch
// so LDS stores stay ds_write and global stores stay global_store (a merged pointer would degrade both to flat_store).
// Returns true when the op carries a fused GroupNorm: acc[] then holds the finished raw outputs (bias, temb, residual, scale
// applied), this wave's partial sums are parked in LDS, and the caller runs fconv_gn_apply after a workgroup barrier.
template <bool MS, bool TR = false>
__device__ __forceinline__ bool fconv_epi(const OpW& w, const UnetArgs& u, int n0, int mt0, int WM, int nt, int nmt, int lane, float add, float dadd0, const float (&daddm)[4],
                                          f32x4 (&acc)[4], float (&ps1)[4], float (&ps2)[4]) {
    const int lrow = lane & 15, kq = lane >> 4;
    const int o_rows = OPI(w, rows), o_Cout = OPI(w, Cout);
    const int o_resid = OPI(w, resid_off), o_resid_rs = OPI(w, resid_rs);
    const int o_kind = OPI(w, dst_kind), o_dst = OPI(w, dst_off), o_dst_rs = OPI(w, dst_rs);
    const float o_scale = OPF(w, scale);
    const int col = nt * 16 + lrow;
    if (o_kind == 3) {                        // fused q/k/v projection: the column tile decides the destination
        const int sc_ = OPI(w, split_C);
        const int which = col / sc_, lc = col - which * sc_;          // wave-uniform (16-col tiles never straddle)
        if (which == 2) {
            float* dstp = lds_f(OPI(w, dst3_off));
            const int rs3 = OPI(w, dst3_rs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt) {
                    const int row0 = (mt0 + i * WM) * 16 + kq * 4;
                    f32x4 v = acc[i];
                    for (int r = 0; r < 4; ++r) v[r] = (v[r] + add) * o_scale;
                    *reinterpret_cast<f32x4*>(dstp + lc * rs3 + row0) = v;
                }
        } else {
            float* dstp = lds_f(which == 0 ? o_dst : OPI(w, dst2_off));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt) {
                    const int row0 = (mt0 + i * WM) * 16 + kq * 4;
                    if ((mt0 + i * WM) * 16 + 16 <= o_rows) {          // whole tile inside the image (wave-uniform): no per-row guards
#pragma unroll
                        for (int r = 0; r < 4; ++r) dstp[(row0 + r) * o_dst_rs + lc] = (acc[i][r] + add) * o_scale;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < o_rows) dstp[(row0 + r) * o_dst_rs + lc] = (acc[i][r] + add) * o_scale;
                    }
                }
        }
        return false;
    }
    if (o_kind == 1) {                        // LDS, transposed [col][row]; all padded rows written (finite)
        float* dstp = lds_f(o_dst);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nmt) {
                const int row0 = (mt0 + i * WM) * 16 + kq * 4;
                f32x4 v = acc[i];
                for (int r = 0; r < 4; ++r) v[r] = (v[r] + add) * o_scale;
                *reinterpret_cast<f32x4*>(dstp + col * o_dst_rs + row0) = v;
            }
        return false;
    }
    // (field reads are wave-uniform: they stay outside lane-divergent code)
    const int gn_off = o_kind == 0 ? OPI(w, gn_off) : -1;
    const bool write_raw = gn_off < 0 || OPI(w, gn_raw) != 0;
    float* const g_dst = OPP(w, float, g_out) ? OPP(w, float, g_out) : u.out;
    const int OPI_samp = MS ? OPI(w, samp) : 0, hw_shift = MS ? OPI(w, hw_shift) : 0;
    float dadd[4];              // Dense_0 row: one value per op for a single-sample op, one per row tile for a multi-sample op
#pragma unroll
    for (int i = 0; i < 4; ++i) dadd[i] = (MS && OPI_samp < 0) ? daddm[i] : dadd0;
    if (col < o_Cout) {
        if (o_resid >= 0) {
            const float* resp = lds_f(o_resid);
            float rv[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt) {
                    const int row0 = (mt0 + i * WM) * 16 + kq * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) rv[i][r] = resp[min(row0 + r, o_rows - 1) * o_resid_rs + col];   // unguarded: all reads in flight at once
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][r] = (acc[i][r] + (add + dadd[i]) + rv[i][r]) * o_scale;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][r] = (acc[i][r] + (add + dadd[i]) + 0.f) * o_scale;
        }
        if (TR && o_kind == 0) {                       // training forward: the layer's output also goes where the backward reads it
            float* const st = OPP(w, float, stash);
            if (st) {
                const int sbf = OPI(w, stash_bf16);
                const size_t sbase = (size_t)min(n0 + max(OPI_samp, 0), u.NB - 1) * o_rows * o_Cout + col;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nmt) {
                        const int row0 = (mt0 + i * WM) * 16 + kq * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < o_rows) stact1(st, sbase + (size_t)(row0 + r) * o_Cout, acc[i][r], sbf);
                    }
            }
        }
        if (o_kind == 0) {
            if (write_raw) {
                float* dstp = lds_f(o_dst);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nmt) {
                        const int row0 = (mt0 + i * WM) * 16 + kq * 4;
                        if ((mt0 + i * WM) * 16 + 16 <= o_rows) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dstp[(row0 + r) * o_dst_rs + col] = acc[i][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (row0 + r < o_rows) dstp[(row0 + r) * o_dst_rs + col] = acc[i][r];
                        }
                    }
            }
        } else {
            float* gp = g_dst + (size_t)min(n0 + max(OPI_samp, 0), u.NB - 1) * o_rows * o_Cout + col;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt) {
                    const int row0 = (mt0 + i * WM) * 16 + kq * 4;
                    if ((mt0 + i * WM) * 16 + 16 <= o_rows) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) stg1(gp + (row0 + r) * o_Cout, acc[i][r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < o_rows) stg1(gp + (row0 + r) * o_Cout, acc[i][r]);
                    }
                }
        }
    }
    if (gn_off < 0) return false;
    // ---- fused GroupNorm, part 1: partial (sum, sum of squares) of every 4-channel group of this wave's column tile.
    // Lane (lrow, kq) holds rows kq*4..+3 of each tile for column lrow: sum over its rows, then over the quad's 4 columns.
    //  * single-sample op: the tiles' partials are added and the quad leader parks the pair in slot (wm*4 + kq) of the group;
    //  * multi-sample op, 16 rows per sample: a tile IS a sample -- one slot per (sample, group, kq);
    //  * multi-sample op, 4 rows per sample: the lane's four rows ARE a sample -- the quad sum is already the whole group
    //    statistic and stays in registers (ps1 / ps2).
    // Slots are summed in a fixed order by the readers (no atomics: results are run-to-run identical).
    {
        float* const slots = lds_f(OPI(w, gn_slot_off));
        const int nslots = OPI(w, gn_nslots);
        const int G4 = o_Cout >> 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s1 = 0.f, s2 = 0.f;
            if (i < nmt) {
                const int row0 = (mt0 + i * WM) * 16 + kq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (row0 + r < o_rows && col < o_Cout) ? acc[i][r] : 0.f;
                    s1 += v; s2 += v * v;
                }
            }
            ps1[i] = s1; ps2[i] = s2;
        }
        if (OPI_samp >= 0) {                               // one sample: add the tiles, then one quad reduction
            const float t1 = quad_sum((ps1[0] + ps1[1]) + (ps1[2] + ps1[3])), t2 = quad_sum((ps2[0] + ps2[1]) + (ps2[2] + ps2[3]));
            if ((lrow & 3) == 0) {
                float* slot = slots + ((size_t)(col >> 2) * nslots + ((mt0 % WM) * 4 + kq)) * 2;
                slot[0] = t1; slot[1] = t2;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ps1[i] = quad_sum(ps1[i]); ps2[i] = quad_sum(ps2[i]); }
        }
        if (OPI_samp < 0 && hw_shift == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nmt && (lrow & 3) == 0) {
                    float* slot = slots + ((size_t)((mt0 + i * WM) * G4 + (col >> 2)) * 4 + kq) * 2;
                    slot[0] = ps1[i]; slot[1] = ps2[i];
                }
        }
    }
    return true;
}

// fused GroupNorm, part 2 (after the workgroup barrier): every lane re-reduces its column's group from the parked partials,
// normalises the values it still holds in registers, applies the affine map (+SiLU) and writes the result to gn_off.
template <bool MS, bool TR = false>
__device__ __forceinline__ void fconv_gn_apply(const OpW& w, const UnetArgs& u, int n0, int mt0, int WM, int nt, int nmt, int lane, float gmul, float gadd, const f32x4 (&acc)[4],
                                               const float (&ps1)[4], const float (&ps2)[4]) {
    const int lrow = lane & 15, kq = lane >> 4;
    const int o_rows = OPI(w, rows), o_Cout = OPI(w, Cout), nslots = OPI(w, gn_nslots);
    const int o_samp = MS ? OPI(w, samp) : 0, hw_shift = MS ? OPI(w, hw_shift) : 0;
    const int col = nt * 16 + lrow;
    const int gcol = min(col, o_Cout - 1) >> 2, G4 = o_Cout >> 2;
    const float* slots = lds_f(OPI(w, gn_slot_off));
    const float inv_cnt = OPF(w, inv_cnt), eps = OPF(w, eps);
    const int act = OPI(w, gn_act), rs = OPI(w, gn_rs);
    float* dstp = lds_f(OPI(w, gn_off));
    float mean = 0.f, rstd = 0.f;
    if (o_samp >= 0) {                       // one sample: the same statistics for every tile
        const float* slot = slots + (size_t)gcol * nslots * 2;
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < nslots; k += 2) {            // nslots is a multiple of 4
            const f32x4 p = *reinterpret_cast<const f32x4*>(slot + 2 * k);
            s1 += p[0]; s2 += p[1]; s1 += p[2]; s2 += p[3];
        }
        mean = s1 * inv_cnt;
        rstd = (1.0f / sqrtf(fmaxf(s2 * inv_cnt - mean * mean, 0.f) + eps)) * gmul;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nmt) {
            if (o_samp < 0) {
                float s1, s2;
                if (hw_shift == 4) {
                    const float* slot = slots + (size_t)((mt0 + i * WM) * G4 + gcol) * 8;
                    const f32x4 p = *reinterpret_cast<const f32x4*>(slot), q = *reinterpret_cast<const f32x4*>(slot + 4);
                    s1 = (p[0] + p[2]) + (q[0] + q[2]); s2 = (p[1] + p[3]) + (q[1] + q[3]);
                } else { s1 = ps1[i]; s2 = ps2[i]; }
                mean = s1 * inv_cnt;
                rstd = (1.0f / sqrtf(fmaxf(s2 * inv_cnt - mean * mean, 0.f) + eps)) * gmul;
            }
            const int row0 = (mt0 + i * WM) * 16 + kq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = (acc[i][r] - mean) * rstd + gadd;
                float ya = act ? silu_f(y) : y;
                if (TR) {                               // Dropout_0 of the next conv's input (RD/models/layerspp.py:204): the mask the layer plan and the backward use
                    const int dop = OPI(w, drop_op);
                    if (dop >= 0 && u.drop_p > 0.f)
                        ya *= dropout_scale((uint64_t)*u.seed_dev, (uint32_t)dop, ((uint64_t)min(n0, u.NB - 1) * o_rows + (row0 + r)) * o_Cout + col, u.drop_p);
                }
                if (row0 + r < o_rows && col < o_Cout) dstp[(row0 + r) * rs + col] = ya;
            }
        }
}

// The fused q/k/v projection of an attention block (dst_kind 3: one 1x1 conv with Cout = 3C, K = C = 64) in ONE pass per wave:
// the generic path gives each wave its three column tiles (the q, k and v tile of its column offset) as three separate passes of
// four k-steps each -- three prologues and epilogues around 48 MFMAs.  Here the wave loads its 12 weight fragments up front, reads
// each A fragment once for all three column tiles and runs the 144 MFMAs back to back into 9 accumulator tiles.
// Preconditions (host: FOp::qkv1 set by the planner): ntap == 1, nch == 4, Cout_pad == 12 tiles (WN = 4, WM = 2), mtiles <= 6.
__device__ __forceinline__ void fconv_qkv(const OpW& w, const UnetArgs& u, int wave, int lane) {
    const int lrow = lane & 15, kq = lane >> 4;
    const int wn = wave & 3, wm = wave >> 2;
    const int o_rows = OPI(w, rows), mtiles = OPI(w, mtiles), o_Cout_pad = OPI(w, Cout_pad);
    const int m_lds = OPI(w, main_ph.lds_off), m_rs = OPI(w, main_ph.rs);
    const float* m_w = OPP(w, const float, main_ph.w);
    const float* o_bias = OPP(w, const float, bias);
    const float o_scale = OPF(w, scale);
    const int sc_ = OPI(w, split_C), o_dst = OPI(w, dst_off), o_dst2 = OPI(w, dst2_off), o_dst3 = OPI(w, dst3_off), o_dst_rs = OPI(w, dst_rs), rs3 = OPI(w, dst3_rs);
    const short* tab = reinterpret_cast<const short*>(rdmi_lds + opw_at(w, (int)(offsetof(FOp, tab_off) / 4)));
    const size_t bstride = (size_t)o_Cout_pad * 16;
    f32x4 bf[3][4];
    float add[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = (wn + 4 * c) * 16 + lrow;
        add[c] = ldg1(o_bias + col);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) bf[c][ch] = ldg4(m_w + (size_t)col * 16 + kq * 4 + (size_t)ch * bstride);
    }
    int abase[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) abase[i] = arow(tab, min((wm + 2 * i) * 16 + lrow, mtiles * 16 - 1), m_lds, m_rs, u.zero_off) + kq * 16;
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        f32x4 af[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const f32x4*>(rdmi_lds + abase[i] + ch * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[i][c] = mfma16(af[i][j], bf[c][ch][j], acc[i][c]);
    }
    // epilogue: q -> dst, k -> dst2 ([row][col]), v -> dst3 transposed ([col][row], all padded rows written)
    const int lc = wn * 16 + lrow;                       // column inside q / k / v (split_C = 64 = 4 tiles: tile wn of each)
    (void)sc_;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int mt = wm + 2 * i;
        if (mt >= mtiles) continue;
        const int row0 = mt * 16 + kq * 4;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* dstp = lds_f(c == 0 ? o_dst : o_dst2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < o_rows) dstp[(row0 + r) * o_dst_rs + lc] = (acc[i][c][r] + add[c]) * o_scale;
        }
        f32x4 v = acc[i][2];
        for (int r = 0; r < 4; ++r) v[r] = (v[r] + add[2]) * o_scale;
        *reinterpret_cast<f32x4*>(lds_f(o_dst3) + lc * rs3 + row0) = v;
    }
}

template <bool DIAG, bool MS, bool TR = false>
__device__ __forceinline__ void fop_conv(const OpW& w, const UnetArgs& u, int n0, int wave, int lane, long long* fine) {
    const int ntiles = OPI(w, Cout_pad) >> 4, mtiles = OPI(w, mtiles);
    const int lWN = (ntiles >= 8 && (ntiles & 7) == 0) ? 3 : (ntiles >= 4 ? 2 : (ntiles >= 2 ? 1 : 0));      // log2 of waves along N
    const int WN = 1 << lWN, WM = UW_WAVES >> lWN, lWM = 3 - lWN;
    const int wn = wave & (WN - 1), wm = wave >> lWN;
    const int o_Cout = OPI(w, Cout), o_dense = OPI(w, dense_off);
    const float* o_bias = OPP(w, const float, bias); const float* o_bias2 = OPP(w, const float, bias2);
    if (OPI(w, dst_kind) == 3 && OPI(w, qkv1)) { fconv_qkv(w, u, wave, lane); return; }
    const bool fused_gn = OPI(w, dst_kind) == 0 && OPI(w, gn_off) >= 0;      // host guarantees: then every wave has at most one pass below
    const int o_samp = MS ? OPI(w, samp) : 0, hw_shift = MS ? OPI(w, hw_shift) : 0;
    const float* dense_base = u.dense + (size_t)max(o_dense, 0);
    f32x4 acc[4];
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
    // Epilogue operands (bias, bias of the folded shortcut, Dense_0 row, GroupNorm affine) are requested now so that their latency
    // hides under the GEMM -- as UNCONDITIONAL loads from clamped addresses whose results are first touched after the main loop:
    // a conditional load or an early `a += b` makes the compiler wait for the data right here, one exposed L2 round trip per conv.
    const int wcol = min(wn * 16 + (lane & 15), o_Cout - 1);
    float gmul = ldg1((fused_gn ? OPP(w, const float, gamma) : o_bias) + wcol), gadd = ldg1((fused_gn ? OPP(w, const float, beta) : o_bias) + wcol);
    int k_mt0 = 0, k_nt = 0, k_nmt = 0;
    for (int nt = wn; nt < ntiles; nt += WN) {
        const int col = nt * 16 + (lane & 15), colc = min(col, o_Cout - 1);
        float add1 = ldg1(o_bias + colc), add2 = ldg1((o_bias2 ? o_bias2 : o_bias) + colc);
        // this wave's row tiles wm, wm+WM, ... in groups of at most 4 (only NMT 1..4 are instantiated)
        for (int mt0 = wm; mt0 < mtiles; mt0 += 4 * WM) {
            const int left = (mtiles - mt0 + WM - 1) >> lWM;
            const int nmt = left >= 4 ? 4 : left;
            // Dense_0(SiLU(temb)) of the sample each row tile belongs to (this lane's rows kq*4..+3 of a tile are one sample)
            float daddm[4] = {0.f, 0.f, 0.f, 0.f};
            float dv = ldg1(dense_base + (o_dense >= 0 ? (size_t)min(n0 + max(o_samp, 0), u.NB - 1) * u.dense_stride + colc : 0));
            if (MS && o_dense >= 0 && o_samp < 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nmt) {
                        const int sl = ((mt0 + i * WM) * 16 + (lane >> 4) * 4) >> hw_shift;
                        daddm[i] = ldg1(dense_base + (size_t)min(n0 + sl, u.NB - 1) * u.dense_stride + colc);
                    }
            }
            const bool tapm = (OPI(w, main_ph.nch) & 3) == 0;        // tap-major main loop (fconv_main_t)
            const bool tapm8 = (OPI(w, main_ph.nch) & 7) == 0;
            switch (nmt) {
                case 1:
                    if (OPI(w, rows) <= 4 && OPI(w, dst_kind) != 1 && OPI(w, dst_kind) != 3) {
                        if (tapm8) fconv_main_t<DIAG, 1, false, 8, true>(w, u, mt0, WM, nt, lane, fine, acc);
                        else fconv_main<DIAG, 1, UW_PF_M4, true>(w, u, mt0, WM, nt, lane, fine, acc);
                    } else if (tapm8) fconv_main_t<DIAG, 1, false, 8>(w, u, mt0, WM, nt, lane, fine, acc);
                    else if (tapm) fconv_main_t<DIAG, 1>(w, u, mt0, WM, nt, lane, fine, acc);
                    else fconv_main<DIAG, 1, UW_PF_N1>(w, u, mt0, WM, nt, lane, fine, acc);
                    break;
                case 2: fconv_main<DIAG, 2, 8>(w, u, mt0, WM, nt, lane, fine, acc); break;
                case 3: {
                    // 81 rows = 5 full tiles + 1 row: the wave that owns the nearly empty last tile runs it on the 4x4x1 form
                    // (13 instead of 32 MFMA cycles per step); it shares its SIMD with a wave of full tiles, so the pipe time saved is real
                    const int last_rows = OPI(w, rows) - (mt0 + 2 * WM) * 16;
                    const bool lm4 = last_rows >= 1 && last_rows <= 4 && OPI(w, dst_kind) != 1 && OPI(w, dst_kind) != 3;
                    if (tapm) { if (lm4) fconv_main_t<DIAG, 3, true>(w, u, mt0, WM, nt, lane, fine, acc); else fconv_main_t<DIAG, 3>(w, u, mt0, WM, nt, lane, fine, acc); }
                    else if (lm4) fconv_main<DIAG, 3, 4, false, true>(w, u, mt0, WM, nt, lane, fine, acc);
                    else fconv_main<DIAG, 3, 4>(w, u, mt0, WM, nt, lane, fine, acc);
                    break;
                }
                case 4:
                    if (tapm) fconv_main_t<DIAG, 4>(w, u, mt0, WM, nt, lane, fine, acc);
                    else fconv_main<DIAG, 4, 4>(w, u, mt0, WM, nt, lane, fine, acc);
                    break;
                default: break;
            }
            if (DIAG && (u.dbg & 256)) continue;                 // ablation: no epilogue
            use_from_here(add1); use_from_here(add2); use_from_here(dv);
            const float add = col < o_Cout ? (o_bias2 ? add1 + add2 : add1) : 0.f, dadd0 = o_dense >= 0 ? dv : 0.f;
            fconv_epi<MS, TR>(w, u, n0, mt0, WM, nt, nmt, lane, add, dadd0, daddm, acc, ps1, ps2);
            k_mt0 = mt0; k_nt = nt; k_nmt = nmt;
            if (DIAG && fine) fine[5] = clock64();
        }
    }
    if (fused_gn) {
        use_from_here(gmul); use_from_here(gadd);
        lds_barrier();
        if (DIAG && fine) fine[6] = clock64();
        if (k_nmt > 0) fconv_gn_apply<MS, TR>(w, u, n0, k_mt0, WM, k_nt, k_nmt, lane, gmul, gadd, acc, ps1, ps2);
        if (DIAG && fine) fine[7] = clock64();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Co-operative program (UnetArgs::coop): a group of four workgroups -- four CUs -- owns four samples.  Every member runs the
// full-resolution sections for ITS sample alone (the S = 1 ops) and the low-resolution section for ALL four samples, but only
// its quarter of every conv's output columns: a CU then streams a quarter of the 17.7 MB of low-resolution weights per forward
// (that section is weight-stream bound with one sample per CU) and every weight fragment it does stream feeds four samples
// (16 rows at 2x2: a full 16x16x4 MFMA tile).  After each 3x3 conv the members all-gather the tensor the next conv contracts
// over (fop_xchg).  Nothing else is shared: skip tensors are spilled per workgroup, GroupNorm / gather ops of concat tensors
// run redundantly on the complete tensor.
//
// CONV of the co-operative section: member m computes column tiles [2m, 2m+2) (Cout_pad = 128) of the ONE row tile that holds the
// four samples' 4 pixels each.  The 8 waves are (column tile ct = wave & 1) x (K group kg = wave >> 1): wave (ct, kg) accumulates a
// PARTIAL sum over K steps kg, kg+4, ...; the kg == 0 wave finishes the tile: the other three park their partials in the
// destination tensor's FOREIGN columns -- dead space until the exchange fills it -- and after one barrier the finisher adds the
// four partials in the fixed order kg = 0..3 (run-to-run identical) and runs the common epilogue (bias, Dense_0 row, residual,
// scale, fused GroupNorm) on its tile.  (The interpreter must stay inside the 64 KiB instruction cache: a four-row-tile variant for
// the 4x4 level existed and was dropped with that level's sharing -- it did not pay)
template <bool DIAG>
__device__ __forceinline__ void fop_conv_coop(const OpW& w, const UnetArgs& u, int n_grp, int m, int wave, int lane, long long* fine) {
    const int lrow = lane & 15, kq = lane >> 4;
    const int ct = wave & 1, kg = wave >> 1;
    const int nt = 2 * m + ct, col = nt * 16 + lrow;
    const int o_Cout = OPI(w, Cout), o_dense = OPI(w, dense_off), hw_shift = OPI(w, hw_shift);
    const float* o_bias = OPP(w, const float, bias); const float* o_bias2 = OPP(w, const float, bias2);
    const bool fused_gn = OPI(w, gn_off) >= 0;
    const bool finisher = kg == 0;
    // epilogue operands: unconditional loads, first touched after the main loop (see fop_conv); every column is real (Cout = 128)
    float add1 = ldg1(o_bias + col), add2 = ldg1((o_bias2 ? o_bias2 : o_bias) + col);
    float gmul = ldg1((fused_gn ? OPP(w, const float, gamma) : o_bias) + col), gadd = ldg1((fused_gn ? OPP(w, const float, beta) : o_bias) + col);
    float dv = ldg1(u.dense + (size_t)max(o_dense, 0) + (o_dense >= 0 ? (size_t)min(n_grp + ((kq * 4) >> hw_shift), u.NB - 1) * u.dense_stride + col : 0));
    f32x4 acc[4];
    {
        const int spt = OPI(w, main_ph.nch) >> 2;           // steps per tap of a wave group
        if (spt == 2) fconv_main_ks<DIAG, 2>(w, u, nt, lane, fine, acc, kg);
        else if (spt == 4 && (OPI(w, main_ph.nch) & 3) == 0) fconv_main_ks<DIAG, 4>(w, u, nt, lane, fine, acc, kg);
        else fconv_main<DIAG, 1, UW_PF_KS1, false, false, true>(w, u, 0, 1, nt, lane, fine, acc, kg);
    }
    // park the partial this wave does not finish: K group kg goes to foreign column block (m + kg) & 3
    float* const dstp = lds_f(OPI(w, dst_off));
    const int drs = OPI(w, dst_rs);
    if (!finisher) {
        float* p = dstp + (size_t)(kq * 4) * drs + (((m + kg) & 3) * 32 + ct * 16 + lrow);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r * drs] = acc[0][r];
    }
    use_from_here(add1); use_from_here(add2); use_from_here(dv); use_from_here(gmul); use_from_here(gadd);
    lds_barrier();
    if (finisher) {
        f32x4 part[3];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float* p = dstp + (size_t)(kq * 4) * drs + (((m + k) & 3) * 32 + ct * 16 + lrow);
            for (int r = 0; r < 4; ++r) part[k - 1][r] = p[r * drs];
        }
        f32x4 v = (acc[0] + part[0]) + (part[1] + part[2]);
        if (DIAG && (u.dbg & 256)) return;
        // epilogue of the tile, written out here (the generic fconv_epi / fconv_gn_apply pair would cost the co-operative kernel ~5 KB of
        // instruction cache): every row is real (16 rows = 4 samples x 4 pixels), every column is real (Cout = 128); this lane's four
        // rows kq*4 .. +3 ARE sample kq, so a GroupNorm group's statistic (4 channels x 4 pixels) is one quad reduction in registers
        const int o_resid = OPI(w, resid_off), o_resid_rs = OPI(w, resid_rs);
        const float o_scale = OPF(w, scale), add_all = (o_bias2 ? add1 + add2 : add1) + (o_dense >= 0 ? dv : 0.f);
        const int row0 = kq * 4;
        if (o_resid >= 0) {
            const float* rp = lds_f(o_resid) + (size_t)row0 * o_resid_rs + col;
            float rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = rp[r * o_resid_rs];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (v[r] + add_all + rv[r]) * o_scale;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (v[r] + add_all + 0.f) * o_scale;
        }
        if (!fused_gn || OPI(w, gn_raw) != 0) {
            float* p = dstp + (size_t)row0 * drs + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r * drs] = v[r];
        }
        if (fused_gn) {
            const float s1 = quad_sum((v[0] + v[1]) + (v[2] + v[3])), s2 = quad_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            const float mean = s1 * OPF(w, inv_cnt);
            const float rstd = (1.0f / sqrtf(fmaxf(s2 * OPF(w, inv_cnt) - mean * mean, 0.f) + OPF(w, eps))) * gmul;
            const int act = OPI(w, gn_act), grs = OPI(w, gn_rs);
            float* p = lds_f(OPI(w, gn_off)) + (size_t)row0 * grs + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float y = (v[r] - mean) * rstd + gadd; p[r * grs] = act ? silu_f(y) : y; }
        }
    }
}

// XCHG: all-gather of one (or two) LDS tensors between the four members of a group.  Every member publishes the block it owns
// as 16-byte pairs of {value, tag} granules (device-scope write-through stores) into ITS slot of the exchange buffer and reads
// the other three members' slots with device-scope loads until every tag equals this exchange's epoch -- the data is the
// flag: no fence, no separate flag word, correct under any workgroup placement.  Slots alternate with the parity of the
// exchange index: a member can be at most one exchange ahead of another (it cannot pass exchange x+1 before every member has
// published x+1, i.e. has finished reading x), so two slots suffice.  All of a thread's loads are in flight together
// (3 x 16 B); the wait is bounded: on give-up the workgroup flags the launch (UnetArgs::coop_err, *failw).
__device__ __forceinline__ void fop_xchg(const OpW& w, const UnetArgs& u, int g, int m, int tid, int* failw) {
    const int mode = OPI(w, a_hw), rows = OPI(w, rows), Cs = OPI(w, C), xi = OPI(w, xidx);
    const int d_off = OPI(w, dst_off), d_rs = OPI(w, dst_rs), s_off = OPI(w, src_off), s_rs = OPI(w, src_rs);
    const int a_off = OPI(w, a_off), a_rs = OPI(w, a_rs);
    const unsigned epoch = u.epoch_base + (unsigned)xi + 1u;
    const int csh = 31 - __builtin_clz((unsigned)Cs);      // slice width is a power of two (host-checked)
    const int half = (rows << csh) >> 1;                   // granule pairs per tensor block
    const int np = s_off >= 0 ? 2 * half : half;
    const unsigned slotb = (unsigned)u.xslot * 8u;         // bytes per member slot
    const unsigned gbase = (unsigned)((g * 2 + (xi & 1)) * 4) * slotb;
    const XBuf xb = xbuf_make(u.xbuf, 0x7fffffffu);
    // float index in LDS of pair p's first element, for member j's block
    auto at = [&](int p, int j) -> int {
        const int sel = p >= half ? 1 : 0, i = (p - sel * half) << 1, row = i >> csh, c = i & (Cs - 1);
        if (mode == 0) return ((sel ? s_off : d_off) >> 2) + row * (sel ? s_rs : d_rs) + j * Cs + c;
        return (d_off >> 2) + (j * rows + row) * d_rs + c;
    };
    typedef float f32x2 __attribute__((vector_size(8)));
    constexpr int KMAX = 1;                                  // blocks hold at most 512 granule pairs (host-checked): one per thread
    // NOTE (ROCm 7.2 clang): __builtin_bit_cast applied DIRECTLY to a vector element (bit_cast<T>(v[i])) reads element 0 whatever i is
    // -- copy the element into a scalar first.
    // ---- publish my block (mode 1: it comes from the single-sample tensor a_off and is also copied into my rows of the destination)
    const bool withhold = u.coop_break != 0 && g == 0 && m == 3 && xi == 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + k * UW_THREADS;
        if (p < np && !withhold) {
            f32x2 v;
            if (mode == 0) v = *reinterpret_cast<const f32x2*>(lds_f(0) + at(p, m));
            else {
                const int i = p << 1, row = i >> csh, c = i & (Cs - 1);
                v = *reinterpret_cast<const f32x2*>(lds_f(a_off) + row * a_rs + c);
                *reinterpret_cast<f32x2*>(lds_f(0) + at(p, m)) = v;
            }
            const float v0 = v[0], v1 = v[1];
            xbuf_store16(xb, gbase + (unsigned)m * slotb + (unsigned)p * 16u, __builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), epoch);
        }
    }
    // ---- gather the other three blocks: every pass issues ALL of this thread's loads back to back (clamped to a valid pair, so
    //      there is no branch between them), then looks at the tags
    const int nk = (np + UW_THREADS - 1) / UW_THREADS;          // passes over the pairs (wave-uniform)
    u32x4 y[3 * KMAX];
    unsigned long long spins = 0;
    // a give-up anywhere on the device (this launch or an earlier one: the word is never cleared) ends all waiting: the launches
    // still queued finish at once instead of timing out exchange by exchange; their samples are marked like the first one's
    bool broken = __hip_atomic_load(u.coop_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (broken) *failw = 1;
    for (; !broken;) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < nk) {
                const unsigned po = (unsigned)min(tid + k * UW_THREADS, np - 1) * 16u;
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) y[k * 3 + jj] = xbuf_load16(xb, gbase + (unsigned)((m + 1 + jj) & 3) * slotb + po);
            }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < nk) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) { const unsigned t1 = y[k * 3 + jj][1], t3 = y[k * 3 + jj][3]; ok &= t1 == epoch && t3 == epoch; }
            }
        if (ok) break;
        if (++spins > RDMI_SPIN_LIMIT) { *failw = 1; __hip_atomic_store(u.coop_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        spin_relax();
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + k * UW_THREADS;
        if (!broken && k < nk && p < np) {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const unsigned a = y[k * 3 + jj][0], b = y[k * 3 + jj][2];
                *reinterpret_cast<f32x2*>(lds_f(0) + at(p, (m + 1 + jj) & 3)) = f32x2{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
            }
        }
    }
}

// Attention core on LDS tensors: P = softmax(Q K^T * scale) (rows = queries), O = P V.
// Q, K: [L][qk_rs]; Vt: [C][ps] (keys along the row, all Lpad columns finite); P: [L][ps]; O -> dst [L][dst_rs].
// C = 64 channels (one 16-wide channel tile per wave pair).
__device__ __forceinline__ void fop_attn(const OpW& w, int wave, int lane) {
    struct { int L, Lpad, q_off, k_off, vt_off, p_off, qk_rs, ps, C, dst_off, dst_rs; float att_scale; } o;
    o.L = OPI(w, L); o.Lpad = OPI(w, Lpad); o.q_off = OPI(w, q_off); o.k_off = OPI(w, k_off); o.vt_off = OPI(w, vt_off); o.p_off = OPI(w, p_off);
    o.qk_rs = OPI(w, qk_rs); o.ps = OPI(w, ps); o.C = OPI(w, C); o.dst_off = OPI(w, dst_off); o.dst_rs = OPI(w, dst_rs); o.att_scale = OPF(w, att_scale);
    const int lrow = lane & 15, kq = lane >> 4;
    const int L = o.L, mtiles = o.Lpad >> 4;
    const float* Q = lds_f(o.q_off);
    const float* K = lds_f(o.k_off);
    const float* Vt = lds_f(o.vt_off);
    float* P = lds_f(o.p_off);
    const int rs = o.qk_rs, ps = o.ps;
    const int nchq = o.C >> 4;
    // ---- scores + softmax: wave w owns query tile w (tiles >= 8 loop)
    for (int mt = wave; mt < mtiles; mt += UW_WAVES) {
        f32x4 s[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int qrow = min(mt * 16 + lrow, L - 1);
        for (int ch = 0; ch < nchq; ++ch) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(Q + (size_t)qrow * rs + ch * 16 + kq * 4);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (t < mtiles) {
                    const int krow = min(t * 16 + lrow, L - 1);
                    const f32x4 bf = *reinterpret_cast<const f32x4*>(K + (size_t)krow * rs + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[t] = mfma16(af[j], bf[j], s[t]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if (t < mtiles && t * 16 + lrow < L) { s[t][r] *= o.att_scale; mx = fmaxf(mx, s[t][r]); }
            mx = row16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if (t < mtiles) {
                    const float e = (t * 16 + lrow < L) ? __expf(s[t][r] - mx) : 0.f;
                    s[t][r] = e;
                    sum += e;
                }
            sum = row16_sum(sum);
            const float inv = 1.0f / sum;
            const int row = mt * 16 + kq * 4 + r;
            if (row < L) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    if (t < mtiles) P[(size_t)row * ps + t * 16 + lrow] = s[t][r] * inv;
            }
        }
    }
    lds_barrier();
    // ---- O = P V: 4 channel tiles x mtiles row tiles over 8 waves (wave = (row half, channel tile))
    {
        const int wn = wave & 3, wm = wave >> 2;
        const int col = wn * 16 + lrow;
        f32x4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < mtiles; ++ch) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(Vt + (size_t)col * ps + ch * 16 + kq * 4);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int mt = wm + 2 * i;
                if (mt < mtiles) {
                    const int prow = min(mt * 16 + lrow, L - 1);
                    const f32x4 af = *reinterpret_cast<const f32x4*>(P + (size_t)prow * ps + ch * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = mfma16(af[j], bf[j], acc[i]);
                }
            }
        }
        float* O = lds_f(o.dst_off);
        // with NIN_3 folded into the value projection (rdmi.hip: fused_attn) this IS the block's output: (P V' + b3 + x) / sqrt2
        const float* b3 = OPP(w, const float, bias);
        const int o_resid = OPI(w, resid_off), o_resid_rs = OPI(w, resid_rs);
        const float o_scale = OPF(w, scale), badd = b3 ? ldg1(b3 + col) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int mt = wm + 2 * i;
            if (mt < mtiles)
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + kq * 4 + r;
                    if (row < L) {
                        float v = acc[i][r];
                        if (b3) v = (v + badd + lds_f(o_resid)[(size_t)row * o_resid_rs + col]) * o_scale;
                        O[(size_t)row * o.dst_rs + col] = v;
                    }
                }
        }
    }
}

// DIAG = true is the diagnostic build of the same kernel (per-op cycle stamps, per-op-kind ablation); the production
// instantiation compiles all of that away -- the interpreter has to stay inside the 64 KiB instruction cache.
// MS = true: the program may hold multi-sample ops (S > 1 samples per workgroup); MS = false compiles every trace of that away
// (the S = 1 program keeps its leaner code).  COOP = true (with MS): the co-operative program -- see fop_conv_coop.
// TRAIN = true: the training forward -- every layer's output is also stashed for the backward, Dropout_0 is applied (fconv_epi / fconv_gn_apply).
template <bool DIAG, bool MS = false, bool COOP = false, bool TRAIN = false>
__global__ __launch_bounds__(UW_THREADS) void unet_wg_kernel(UnetArgs u) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // sample bookkeeping.  n: first sample of this workgroup's single-sample ops (ops add their slot); n_multi: first sample of
    // its multi-sample ops; n_spill / spill_cap: sample index (and bound) under which multi-sample ops address the spill buffer.
    int n = MS ? blockIdx.x * u.S : blockIdx.x, n_multi = n, n_spill = n, spill_cap = u.NB;
    int cg = 0, cm = 0;                                                        // co-operative group and member index
    if (COOP) {
        const int st = u.coop_stride, bid = blockIdx.x, blk = bid / (4 * st), r = bid - blk * 4 * st;
        cm = r / st; cg = blk * st + (r - cm * st);
        if (cg * 4 >= u.NB) return;                                            // a group without a real sample: all four members leave
        n = min(cg * 4 + cm, u.NB - 1);                                        // a member beyond the batch recomputes the last sample
        n_multi = cg * 4;
        n_spill = bid * 4; spill_cap = 0x7fffffff;                             // every member spills all four samples' skip tensors privately
    }
    // tables + zero row
    {
        const int nw = u.tab_bytes >> 2;
        const int* src = reinterpret_cast<const int*>(u.tabs);
        int* dst = reinterpret_cast<int*>(rdmi_lds + u.tab_base);
        for (int i = tid; i < nw; i += UW_THREADS) dst[i] = src[i];
        float* z = lds_f(u.zero_off);
        for (int i = tid; i < (u.zero_bytes >> 2); i += UW_THREADS) z[i] = 0.f;
    }
#if defined(__HIP_DEVICE_COMPILE__) && UW_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);       // the second-dispatched half loses every arbitration otherwise (MI355X_MICROARCH.md)
#elif defined(__HIP_DEVICE_COMPILE__) && UW_PRIO == 2
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
    float* stat = lds_f(u.zero_off + u.zero_bytes);      // [2 * 32] GroupNorm scratch right after the zero row
    int* const failw = reinterpret_cast<int*>(stat) + 270;      // co-operative program: "an exchange gave up" (last word of the scratch block)
    if (COOP && tid == 0) *failw = 0;
    if (DIAG && u.stamps && n == 0 && tid == 0) u.stamps[0] = clock64();
    // descriptors are kept TWO ops ahead (cur, nxt resident; the load for pc+2 is in flight) so that small operands
    // of the next op can be prefetched while the current one runs
    OpW cur = opw_load(u.prog, lane);
    OpW nxt = opw_load(u.prog + (u.nops > 1 ? 1 : 0), lane);
    f32x4 pgm = {1.f, 1.f, 1.f, 1.f}, pbt = {0.f, 0.f, 0.f, 0.f};
    gn_prefetch(cur, tid, pgm, pbt);
    for (int pc = 0; pc < u.nops; ++pc) {
        const OpW nn = opw_load(u.prog + (pc + 2 < u.nops ? pc + 2 : u.nops - 1), lane);
        long long* fine = (DIAG && u.stamps && n == 0 && tid == ((u.dbg >> 16) & 7) * 64) ? u.stamps + 1024 + pc * 8 : nullptr;   // RDMI_UDBG bits 16-18: the wave whose fine stamps are kept
        const int kind = OPI(cur, kind);
        if (DIAG && fine) fine[2] = clock64();
        f32x4 ngm = pgm, nbt = pbt;
        if (pc + 1 < u.nops) gn_prefetch(nxt, tid, ngm, nbt);
        if (DIAG && fine) fine[3] = clock64();
        const int skip = (DIAG && u.dbg) ? ((kind == FOP_GN ? 4 : kind == FOP_CONV ? 8 : kind == FOP_ATTN ? 16 : 32) & u.dbg) : 0;
        const bool multi = MS && OPI(cur, samp) < 0;
        switch (skip ? -1 : kind) {
            case FOP_GATHER: fop_gather<MS>(cur, u, multi ? n_spill : n, multi ? spill_cap : u.NB, COOP ? cm : 0, tid); break;
            case FOP_STORE: fop_store<MS>(cur, u, multi ? n_spill : n, multi ? spill_cap : u.NB, tid); break;
            case FOP_GN: fop_gn<MS>(cur, stat, tid, pgm, pbt, DIAG ? u.dbg : 0); break;
            case FOP_CONV:
                if (COOP && OPI(cur, coop)) fop_conv_coop<DIAG>(cur, u, n_multi, cm, wave, lane, fine);
                else fop_conv<DIAG, MS && !COOP, TRAIN>(cur, u, multi ? n_multi : n, wave, lane, fine);      // co-operative: every multi-sample conv is a coop conv
                break;
            case FOP_ATTN: fop_attn(cur, wave, lane); break;
            case FOP_LOADTAB: {      // row tables of the multi-sample section: global -> their LDS block
                const int* src = OPP(cur, const int, a_g);
                int* dst = reinterpret_cast<int*>(rdmi_lds + OPI(cur, dst_off));
                const int nw = OPI(cur, rows) >> 2;
                for (int i = tid; i < nw; i += UW_THREADS) dst[i] = src[i];
                break;
            }
            case FOP_XCHG: if (COOP) fop_xchg(cur, u, cg, cm, tid, failw); break;
            default: break;
        }
        // ops that stored to global memory (skip spills, the network output) end with the full barrier: their stores must have
        // completed before a later op of ANOTHER wave reads them back; everything else hands over through LDS only
        if (kind == FOP_STORE || (kind == FOP_CONV && OPI(cur, dst_kind) == 2)) __syncthreads();
        else lds_barrier();
        if (DIAG && u.stamps && n == 0 && tid == 0) u.stamps[pc + 1] = clock64();
        cur = nxt; nxt = nn; pgm = ngm; pbt = nbt;
    }
    if (COOP && *failw) {                                   // an exchange of this workgroup gave up: its result is not the network's -- make that loud
        const int E = (int)u.out_elems;
        for (int i = tid; i < E; i += UW_THREADS) u.out[(size_t)n * E + i] = __builtin_nanf("");
    }
}
