// Diagnostic: operand/result layout and issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void layout(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + l;          // A value supplied by lane l
    const float b = 100.0f + l;         // B value supplied by lane l
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = threadIdx.x, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) out[1024] = (float)(t1 - t0) / (4.0f * iters);
}
__global__ __launch_bounds__(256) void rate_dep(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0};
    const float a = threadIdx.x, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3];
    if (threadIdx.x == 0) out[1024] = (float)(t1 - t0) / (4.0f * iters);
}
__global__ __launch_bounds__(256) void rate16(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = threadIdx.x, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) out[1024] = (float)(t1 - t0) / (4.0f * iters);
}
int main() {
    float* d; hipMalloc(&d, 8192); float h[2048];
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 63}) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
    printf("independent: %.2f shader-clock ticks per MFMA (1 wave)\n", h[1024]);
    hipLaunchKernelGGL(rate_dep, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
    printf("dependent:   %.2f ticks per MFMA (1 wave)\n", h[1024]);
    hipLaunchKernelGGL(rate, dim3(1), dim3(256), 0, 0, d, 10000);
    hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
    printf("independent, 4 waves (1 per SIMD): %.2f ticks per MFMA per wave\n", h[1024]);
    hipLaunchKernelGGL(rate16, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 8192, hipMemcpyDeviceToHost);
    printf("16x16x4 independent: %.2f ticks per MFMA (1 wave)\n", h[1024]);
    return 0;
}
