// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the UNMODIFIED kernel/host sources of
// optimized-diffusion-model_amd/csrc into librdmi_emu.so, which executes every workgroup on the CPU as
// 256 cooperative fibers (one per work-item) with wave64 collectives (MFMA 16x16x4 f32, shuffles) and
// __syncthreads emulated.  Purpose: run the real launch plan + kernels against the oracle on the
// GPU-less build container (indexing, tables, packing, synchronisation) before spending
// GPU minutes.  It is never loaded by the product (rdmi/_native.py only opens librdmi.so).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>

#define RDMI_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef float f32x4_emu __attribute__((vector_size(16)));

namespace emu {
struct Lane;
extern thread_local Lane* cur;
extern thread_local dim3 t_bid, t_bdim, t_gdim;
const dim3& cur_tid();
void sync_block();
// all 64 lanes of the calling wave deposit `n` bytes; returns pointer to the wave's [64][n] table
const unsigned char* wave_exchange(const void* mine, size_t n);
int lane_id();
// a work-item that polls memory written by ANOTHER workgroup (an OS thread of its own): let the other fibers and threads run
void relax();
void launch(void (*tramp)(void*), void* packed_args, dim3 grid, dim3 block, size_t lds);
}  // namespace emu

#define threadIdx (emu::cur_tid())
#define blockIdx (emu::t_bid)
#define blockDim (emu::t_bdim)
#define gridDim (emu::t_gdim)

inline void __syncthreads() { emu::sync_block(); }

template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    const unsigned char* all = emu::wave_exchange(&v, sizeof(T));
    T r;
    std::memcpy(&r, all + (size_t)((emu::lane_id() ^ mask) & 63) * sizeof(T), sizeof(T));
    return r;
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    const unsigned char* all = emu::wave_exchange(&v, sizeof(T));
    T r;
    std::memcpy(&r, all + (size_t)(src & 63) * sizeof(T), sizeof(T));
    return r;
}

// v_mfma_f32_16x16x4_f32: lane l holds A[l&15][l>>4], B[l>>4][l&15]; D: col = l&15, rows (l>>4)*4 + r.
// Numerics: k-ordered fmaf chain (cdna_hip_programming.md, FP32-input MFMA).
inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int) {
    float ab[2] = {a, b};
    const float* all = reinterpret_cast<const float*>(emu::wave_exchange(ab, sizeof(ab)));
    int l = emu::lane_id();
    int col = l & 15, rg = l >> 4;
    f32x4_emu d = c;
    for (int r = 0; r < 4; ++r) {
        int row = rg * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(all[2 * (row + 16 * k)], all[2 * (col + 16 * k) + 1], acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 blocks (block = l>>2); lane 4b+i supplies A_b[i], lane 4b+j supplies B_b[j];
// D_b[i][j] lands in element i of lane 4b+j (layout probed on gfx950: scripts/micro/mfma4.hip).
inline f32x4_emu __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4_emu c, int, int, int) {
    float ab[2] = {a, b};
    const float* all = reinterpret_cast<const float*>(emu::wave_exchange(ab, sizeof(ab)));
    int l = emu::lane_id();
    int blk = l >> 2;
    f32x4_emu d = c;
    for (int i = 0; i < 4; ++i) d[i] = fmaf(all[2 * (4 * blk + i)], all[2 * l + 1], c[i]);
    return d;
}

inline long long clock64() { return 0; }
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
template <class T, class V> inline void __hip_atomic_store(T* p, V v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_RELEASE); }
inline float atomicAdd(float* p, float v) {   // workgroups run on parallel OS threads: a real atomic
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do { std::memcpy(&f, &old, 4); f += v; std::memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    std::memcpy(&f, &old, 4);
    return f;
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }
template <class T, class U> inline T __builtin_bit_cast_emu(U u) { T t; std::memcpy(&t, &u, sizeof(T)); return t; }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // callers pass wave-uniform values
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int __umulhi(unsigned int a, unsigned int b) { return (unsigned int)(((uint64_t)a * b) >> 32); }

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
typedef struct hipGraph_st* hipGraph_t;
typedef struct hipGraphExec_st* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };

inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "ok" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches are synchronous here
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// graph capture is not emulated: the host code falls back to plain launches when Begin fails
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

template <class... KArgs, class... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t, Args... args) {
    struct Pack {
        void (*k)(KArgs...);
        std::tuple<KArgs...> a;
    } pack{kernel, std::tuple<KArgs...>(static_cast<KArgs>(args)...)};
    emu::launch([](void* p) {
        Pack* q = static_cast<Pack*>(p);
        std::apply(q->k, q->a);
    }, &pack, grid, block, lds);
}
