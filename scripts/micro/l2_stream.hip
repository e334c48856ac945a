// Micro-benchmark (diagnostic, not part of the product): how fast can ONE workgroup per CU stream a buffer that every
// workgroup reads in the same order (the weight stream of the fused U-Net's low-resolution convs)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char lds[];

template <int U, int MODE>   // MODE bit0: 4 MFMA 4x4x1 per chunk, bit1: an LDS b128 read per chunk, bit2: 16x16x4 MFMAs instead
__global__ __launch_bounds__(512) void stream_kernel(const f32x4* __restrict__ buf, size_t n_vec, float* out, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const size_t chunks = n_vec / 64;            // 1 KiB chunks
    for (int r = 0; r < reps; ++r) {
        f32x4 ring[U];
        size_t c = wave;
#pragma unroll
        for (int u = 0; u < U; ++u) { ring[u] = __builtin_nontemporal_load(buf + 0) * 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t cc = c + (size_t)u * 8 < chunks ? c + (size_t)u * 8 : chunks - 1; ring[u] = buf[cc * 64 + lane]; }
        for (; c < chunks; c += 8 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE & 2) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(lds + ((lane & 3) * 528 + (lane >> 4) * 16 + ((c + u) & 7) * 64));
                    if (MODE & 1) {
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(af[0], ring[u][0], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[1], ring[u][1], acc2, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(af[2], ring[u][2], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(af[3], ring[u][3], acc2, 0, 0, 0);
                    } else if (MODE & 4) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], ring[u][0], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], ring[u][1], acc2, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], ring[u][2], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], ring[u][3], acc2, 0, 0, 0);
                    } else acc += ring[u] * af;
                } else if (MODE & 1) {
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, ring[u][0], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, ring[u][1], acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, ring[u][2], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, ring[u][3], acc2, 0, 0, 0);
                } else
                acc += ring[u];
                size_t nx = c + (size_t)(u + U) * 8;
                nx = nx < chunks ? nx : chunks - 1;
                ring[u] = buf[nx * 64 + lane];
            }
        }
    }
    acc += acc2;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[blockIdx.x] = acc[0];
    if (threadIdx.x == 0) lds[0] = 1;
}

template <int U, int MODE>
void run(const f32x4* buf, size_t bytes, float* out, int grid, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<U, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<U, MODE>), dim3(grid), dim3(512), 150 * 1024, 0, buf, bytes / 16, out, reps);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode=%d U=%2d grid=%3d buf=%5.1f MB: %.3f ms -> %.1f GB/s per workgroup, %.2f TB/s total\n", MODE, U, grid, bytes / 1e6, ms,
           (double)bytes * reps / ms / 1e6, (double)bytes * reps * grid / ms / 1e9);
    fflush(stdout);
}

int main() {
    const size_t maxb = 64u << 20;
    f32x4* buf; float* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 4096);
    hipMemset(buf, 0, maxb);
    for (size_t mb : {18}) {
        const size_t bytes = mb << 20;
        const int reps = (int)(200 / mb) + 1;
        for (int grid : {256}) {
            run<8, 0>(buf, bytes, out, grid, reps);
            run<8, 1>(buf, bytes, out, grid, reps);
            run<8, 2>(buf, bytes, out, grid, reps);
            run<8, 3>(buf, bytes, out, grid, reps);
            run<8, 6>(buf, bytes, out, grid, reps);
            run<4, 3>(buf, bytes, out, grid, reps);
            run<16, 3>(buf, bytes, out, grid, reps);
        }
    }
    // one pass over 590 KB (one low-resolution conv's weights): the ramp matters
    for (int reps : {1, 4}) { run<8, 0>(buf, 589824, out, 256, reps); run<8, 3>(buf, 589824, out, 256, reps); }
    return 0;
}
