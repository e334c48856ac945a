// Shared device-side definitions for the gfx950 kernels of librdmi.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((vector_size(16)));

// The single dynamic-LDS symbol (all kernels carve it themselves; base is 16-B aligned).
extern __shared__ __attribute__((aligned(16))) unsigned char rdmi_lds[];

#define RDMI_THREADS 256

// instruction-scheduling fence: nothing is moved across it by the compiler's scheduler (no code is emitted)
#if defined(__HIP_DEVICE_COMPILE__)
#define RDMI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define RDMI_SCHED_FENCE() ((void)0)
#endif

__device__ __forceinline__ float silu_f(float y) { return __fdividef(y, 1.0f + __expf(-y)); }

// v_mfma_f32_16x16x4_f32: exact fp32 (k-ordered fma chain), 1024 MAC per wave-instruction.
// lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15];
// result: col = l&15, rows (l>>4)*4 + r in element r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4x1 outer products (block b = l>>2), exact fp32, 13 cycles per issue
// (measured, scripts/micro/mfma4.hip; the 16x16x4 form takes 32).  Lane 4b+i supplies A_b[i], lane 4b+j supplies B_b[j];
// element i of lane 4b+j receives D_b[i][j].  Used where an image has <= 4 pixels: with lane l reading the SAME
// operands as for mfma16 except A row = l&3, block (l>>2)&3 covers columns 4*((l>>2)&3)..+3 and l>>4 selects the k
// group, so the four k groups are summed across lanes l, l^16, l^32, l^48 at the end.
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// bf16 operands (BASELINE configs #4 / #5): storage is uint16 everywhere; v_mfma_f32_16x16x32_bf16 takes 8 bf16 per lane:
// lane l supplies A[row l&15][k = 8(l>>4) + j] and B[k = 8(l>>4) + j][col l&15], j = 0..7; C/D layout as the fp32 form.
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((vector_size(16)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 rdmi_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rdmi_bf16x8, a), __builtin_bit_cast(rdmi_bf16x8, b), c, 0, 0, 0);
}
#else
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#if defined(RDMI_EMU)
// CPU emulator (tests/emu): the wave's 64 fragments are exchanged and every lane forms its four outputs in fp32
__device__ __forceinline__ float bf2f_host(bf16_t h) { const unsigned u = (unsigned)h << 16; return __builtin_bit_cast(float, u); }
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    bf16_t A[64][8], B[64][8];
    std::memcpy(A, emu::wave_exchange(&a, 16), sizeof A);
    std::memcpy(B, emu::wave_exchange(&b, 16), sizeof B);
    const int lane = emu::lane_id(), col = lane & 15;
    for (int e = 0; e < 4; ++e) {
        const int r = (lane >> 4) * 4 + e;
        float s = c[e];
        for (int k = 0; k < 32; ++k) s += bf2f_host(A[r + 16 * (k >> 3)][k & 7]) * bf2f_host(B[col + 16 * (k >> 3)][k & 7]);
        c[e] = s;
    }
    return c;
}
#else
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4, u32x4, f32x4 c) { abort(); return c; }      // host pass of hipcc: never executed
#endif
#endif
// wave_order_point: program order between LDS accesses of DIFFERENT lanes of one wave.  On the GPU a wave's LDS instructions execute in
// issue order for all 64 lanes, so this is only a compiler fence; the CPU emulator runs lanes as independent fibers and needs a real
// rendezvous.  Every lane of the wave must reach it.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wave_order_point() { __builtin_amdgcn_wave_barrier(); }
#elif defined(RDMI_EMU)
__device__ __forceinline__ void wave_order_point() { int z = 0; (void)emu::wave_exchange(&z, sizeof z); }
#else
__device__ __forceinline__ void wave_order_point() {}
#endif
// compiler fences (no instructions): sched_fence keeps the machine scheduler from moving code across it; opaque_sgpr makes a
// wave-uniform value opaque to the optimiser so that addresses derived from it are recomputed where used instead of hoisted
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void opaque_sgpr(int& v) { asm volatile("" : "+s"(v)); }
// "this value is first needed HERE": pins the use of a loaded value below the point of the call (the optimiser otherwise hoists
// loop-invariant arithmetic on it -- and with it the s_waitcnt for the load -- above a long loop the load was meant to fly under)
__device__ __forceinline__ void use_from_here(float& v) { asm volatile("" : "+v"(v)); }
#else
__device__ __forceinline__ void sched_fence() {}
__device__ __forceinline__ void opaque_sgpr(int&) {}
__device__ __forceinline__ void use_from_here(float&) {}
#endif
// glds16: asynchronous 16-byte-per-lane copy global -> LDS without a register round trip (global_load_lds_dwordx4, the LDS-DMA of
// gfx950).  The SOURCE address is per lane; the DESTINATION is wave-uniform: lane l's 16 bytes land at rdmi_lds + wave_off + 16 l
// (M0 base + lane * size), so a swizzled LDS image is built by permuting the source addresses.  Completion is counted on vmcnt:
// the __syncthreads() that precedes the first read of the image drains it (hipcc emits s_waitcnt vmcnt(0) before the barrier).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void glds16(const void* g, unsigned wave_off) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(rdmi_lds + wave_off), 16, 0, 0);
}
#elif defined(RDMI_EMU)
__device__ __forceinline__ void glds16(const void* g, unsigned wave_off) { std::memcpy(rdmi_lds + wave_off + 16 * emu::lane_id(), g, 16); }
#else
__device__ __forceinline__ void glds16(const void*, unsigned) {}
#endif
// a * b + c on the full-rate 24-bit integer multiplier (operands < 2^24): v_mad_u32_u24 instead of the quarter-rate v_mul_lo_u32
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int mad_u24(int a, int b, int c) { return (int)(__umul24((unsigned)a, (unsigned)b) + (unsigned)c); }
#else
__device__ __forceinline__ int mad_u24(int a, int b, int c) { return a * b + c; }
#endif
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// Activation tensors of the training step are fp32 or (train_dtype = bf16) bf16 in HBM: same element indexing, `bf` selects the
// element width.  Loads widen exactly; stores round to nearest even.  idx must be a multiple of 4 for the 4-wide forms.
__device__ __forceinline__ f32x4 ldact4(const float* base, size_t idx, int bf) {
    if (bf) {
        typedef unsigned int u32x2 __attribute__((vector_size(8)));
        const u32x2 u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
        return f32x4{__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u), __builtin_bit_cast(float, u[1] << 16),
                     __builtin_bit_cast(float, u[1] & 0xffff0000u)};
    }
    return *reinterpret_cast<const f32x4*>(base + idx);
}
__device__ __forceinline__ float ldact1(const float* base, size_t idx, int bf) { return bf ? bf2f(reinterpret_cast<const bf16_t*>(base)[idx]) : base[idx]; }
__device__ __forceinline__ void stact1(float* base, size_t idx, float v, int bf) {
    if (bf) reinterpret_cast<bf16_t*>(base)[idx] = f2bf(v); else base[idx] = v;
}

// All-reduce over the 16 lanes of a DPP row (lanes 16r..16r+15) in four VALU-DPP steps (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror).  __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar round trip of ~100+ cycles per
// step, four to six dependent steps per reduction); these stay in the VALU.  Each step adds the same two partial sums
// the xor-4 / xor-8 butterfly would, so the result is bitwise that of the xor butterfly in the order 1, 2, 4, 8
// (which is what the host/emulator build executes).
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
    return v;
}
// value of lane `l` (compile-time) as a wave-uniform float
__device__ __forceinline__ float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
#else
__device__ __forceinline__ float row16_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; }
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float row8_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
__device__ __forceinline__ float lane_bcast(float v, int l) { return __shfl(v, l); }
#endif
// sum over aligned groups of 32 / 64 lanes: row sums, then the (row-uniform) sums of the other rows by v_readlane
__device__ __forceinline__ float group32_sum(float v, int lane) {
    const float r = row16_sum(v);
    const float s0 = lane_bcast(r, 0), s1 = lane_bcast(r, 16), s2 = lane_bcast(r, 32), s3 = lane_bcast(r, 48);
    return lane < 32 ? s0 + s1 : s2 + s3;
}
__device__ __forceinline__ float group64_sum(float v) {
    const float r = row16_sum(v);
    return (lane_bcast(r, 0) + lane_bcast(r, 16)) + (lane_bcast(r, 32) + lane_bcast(r, 48));
}

// cube.reflect (RD/cube.py:34-49): floor-mod 2 then fold (1,2] onto [0,1).
// x - 2*floor(x/2) is exact for the even multiple and rounds once, like torch.remainder.
__device__ __forceinline__ float reflect_f(float x) {
    float m = x - 2.0f * floorf(x * 0.5f);
    // guard the fp32 edge where a tiny negative x rounds m to exactly 2.0
    m = (m >= 2.0f) ? 0.0f : m;
    return (m > 1.0f) ? 2.0f - m : m;
}

// 16-byte load through the GLOBAL address space.  Pointers that the kernel reads out of memory (op lists) are
// "generic" to the compiler, which then emits flat_load: flat loads also count on lgkmcnt, so every LDS wait
// would drain the whole weight-prefetch ring.  The device pass casts to address space 1 (global_load_dwordx4);
// host-side passes (which never execute this) see a plain load.
__device__ __forceinline__ f32x4 ldg4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const f32x4 __attribute__((address_space(1))) * gptr_t;
    return *(gptr_t)(p);
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ float ldg1(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const float __attribute__((address_space(1))) * gptr_t;
    return *(gptr_t)(p);
#else
    return *p;
#endif
}

__device__ __forceinline__ void stg1(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float __attribute__((address_space(1))) * gptr_t;
    *(gptr_t)(p) = v;
#else
    *p = v;
#endif
}
__device__ __forceinline__ void stg4(float* p, f32x4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef f32x4 __attribute__((address_space(1))) * gptr_t;
    *(gptr_t)(p) = v;
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// ---- inter-workgroup exchange slots (co-operative U-Net program): 16-byte pairs of 8-byte {value, tag} granules, written and read
// with device-scope (sc1) buffer accesses: the store is write-through, the load bypasses this CU's L1 -- the data IS the flag
// (MI355X_MICROARCH.md "visibility", form R2: no fence, no separate flag, placement-independent).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int xq_t __attribute__((ext_vector_type(4)));
struct XBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ XBuf xbuf_make(unsigned long long* p, unsigned bytes) { return XBuf{__builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000)}; }
__device__ __forceinline__ void xbuf_store16(const XBuf& b, unsigned off, unsigned v0, unsigned v1, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(xq_t{v0, tag, v1, tag}, b.r, (int)off, 0, 16);      // aux 16 = sc1
}
__device__ __forceinline__ u32x4 xbuf_load16(const XBuf& b, unsigned off) {
    const xq_t q = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)off, 0, 16);
    return u32x4{q[0], q[1], q[2], q[3]};
}
// Weight stream of a conv main loop as a BUFFER load: wave-uniform base (resource in SGPRs) + a lane byte offset that never changes
// + a wave-uniform byte offset advanced by scalar adds -- no vector ALU instruction per load (a global_load needs a 64-bit VALU add
// for its address, and VALU instructions do not co-execute with fp32 MFMAs: scripts/micro/mfma_lds.hip, ~4 pipe cycles each).
struct WBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ WBuf wbuf_make(const float* p) { return WBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, -1, 0x00020000)}; }
__device__ __forceinline__ f32x4 wbuf_load4(const WBuf& b, unsigned lane_off, unsigned wave_off) {
    const xq_t q = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)lane_off, (int)wave_off, 0);
    return __builtin_bit_cast(f32x4, q);
}
__device__ __forceinline__ void spin_relax() { __builtin_amdgcn_s_sleep(1); }
#define RDMI_SPIN_LIMIT (1ull << 19)       // ~0.5 s of polling (a poll is a ~1 us round trip), then the workgroup gives up
#elif defined(RDMI_EMU)
struct XBuf { unsigned long long* p; };
__device__ __forceinline__ XBuf xbuf_make(unsigned long long* p, unsigned) { return XBuf{p}; }
__device__ __forceinline__ void xbuf_store16(const XBuf& b, unsigned off, unsigned v0, unsigned v1, unsigned tag) {
    __atomic_store_n(b.p + (off >> 3), ((unsigned long long)tag << 32) | v0, __ATOMIC_RELEASE);
    __atomic_store_n(b.p + (off >> 3) + 1, ((unsigned long long)tag << 32) | v1, __ATOMIC_RELEASE);
}
__device__ __forceinline__ u32x4 xbuf_load16(const XBuf& b, unsigned off) {
    const unsigned long long a = __atomic_load_n(b.p + (off >> 3), __ATOMIC_ACQUIRE), c = __atomic_load_n(b.p + (off >> 3) + 1, __ATOMIC_ACQUIRE);
    return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)c, (unsigned)(c >> 32)};
}
struct WBuf { const char* p; };
__device__ __forceinline__ WBuf wbuf_make(const float* p) { return WBuf{reinterpret_cast<const char*>(p)}; }
__device__ __forceinline__ f32x4 wbuf_load4(const WBuf& b, unsigned lane_off, unsigned wave_off) { return *reinterpret_cast<const f32x4*>(b.p + lane_off + wave_off); }
__device__ __forceinline__ void spin_relax() { emu::relax(); }
#define RDMI_SPIN_LIMIT (1ull << 40)
#else
struct XBuf { unsigned long long* p; };
__device__ __forceinline__ XBuf xbuf_make(unsigned long long* p, unsigned) { return XBuf{p}; }
__device__ __forceinline__ void xbuf_store16(const XBuf&, unsigned, unsigned, unsigned, unsigned) { abort(); }      // host pass of hipcc: never executed
__device__ __forceinline__ u32x4 xbuf_load16(const XBuf&, unsigned) { abort(); return u32x4{0, 0, 0, 0}; }
struct WBuf { const char* p; };
__device__ __forceinline__ WBuf wbuf_make(const float* p) { return WBuf{reinterpret_cast<const char*>(p)}; }
__device__ __forceinline__ f32x4 wbuf_load4(const WBuf& b, unsigned lane_off, unsigned wave_off) { return *reinterpret_cast<const f32x4*>(b.p + lane_off + wave_off); }
__device__ __forceinline__ void spin_relax() {}
#define RDMI_SPIN_LIMIT (1ull << 21)
#endif

// Keep a value alive without using it (for L2-warming touches): the wait for the load lands where this is placed.
#if defined(__HIP_DEVICE_COMPILE__)
#define RDMI_KEEP(x) asm volatile("" ::"v"(x))
#else
#define RDMI_KEEP(x) ((void)(x))
#endif

// Wave-uniform values read out of LDS/global land in VGPRs; these pin them into scalar registers once, so later
// uses are SGPR operands and (for values read from LDS) are not re-read after every LDS store.
__device__ __forceinline__ int sgpr_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sgpr_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <class T>
__device__ __forceinline__ T* sgpr_p(T* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
