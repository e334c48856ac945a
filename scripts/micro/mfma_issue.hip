// Diagnostic: how well two waves per SIMD keep the fp32 matrix pipe busy when each step is 12 x v_mfma_f32_16x16x4_f32 (three
// accumulator chains, the NMT = 3 main loop of unet_kernel.h) plus scalar / LDS / global-load instructions in various arrangements.
// Prints shader-clock cycles per step for 4 waves (1 per SIMD) and 8 waves (2 per SIMD); ideal = 384 per wave of a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define SALU(x) asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 7" : "+s"(x))
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void step_kernel(float* out, const float* gsrc, int steps) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    f32x4 a0 = {1, 2, 3, 4}, a1 = a0, a2 = a0, b = {1, 1, 1, 1}, n0 = a0, n1 = a0, n2 = a0;
    f32x4 ring[4];
    const float* gp = gsrc + tid * 4;
    for (int p = 0; p < 4; ++p) ring[p] = *reinterpret_cast<const f32x4*>(gp + p * 2048);
    int s = steps, x = 1;
    const int la = (tid & 63) * 68 * 4;
    long long t0 = clock64();
    for (int q = 0; q < steps; q += 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (MODE >= 3) {                       // the step's A fragments were read one step ago; the next step's reads are issued now
                a0 = n0; a1 = n1; a2 = n2;
                if (MODE != 6) {
                n0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ((la + (q + p) * 16) & 0x7ff0));
                n1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ((la + 4096 + (q + p) * 16) & 0x7ff0));
                n2 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ((la + 8192 + (q + p) * 16) & 0x7ff0));
                }
                FENCE();
            }
            if (MODE >= 3) b = ring[p];
            MF(c0, a0[0], b[0]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c1, a1[0], b[0]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c2, a2[0], b[0]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c0, a0[1], b[1]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c1, a1[1], b[1]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c2, a2[1], b[1]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c0, a0[2], b[2]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c1, a1[2], b[2]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c2, a2[2], b[2]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c0, a0[3], b[3]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c1, a1[3], b[3]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            MF(c2, a2[3], b[3]); if (MODE == 2 || MODE >= 4) { SALU(x); } FENCE();
            if (MODE == 1 || MODE == 3) {          // 24 scalar instructions clustered at the end of the step
#pragma unroll
                for (int k = 0; k < 12; ++k) SALU(x);
                FENCE();
            }
            if (MODE >= 3 && MODE != 5) ring[p] = *reinterpret_cast<const f32x4*>(gp + ((q + p + 4) & 63) * 2048);
        }
    }
    long long t1 = clock64();
    f32x4 r = c0 + c1 + c2;
    out[blockIdx.x * 512 + tid] = r[0] + r[1] + r[2] + r[3] + x + s;
    if ((tid & 63) == 0 && blockIdx.x == 0) { long long* st = reinterpret_cast<long long*>(out + (1 << 20)); st[(tid >> 6) * 2] = t0; st[(tid >> 6) * 2 + 1] = t1; }
}

template <int MODE>
static void run(float* d, const float* g, const char* what) {
    double span[2], wmin[2], wmax[2];
    for (int w : {4, 8}) {
        hipLaunchKernelGGL(step_kernel<MODE>, dim3(256), dim3(64 * w), 64 * 1024, 0, d, g, 4000);
        hipDeviceSynchronize();
        long long st[16];
        hipMemcpy(st, d + (1 << 20), sizeof(st), hipMemcpyDeviceToHost);
        long long lo = st[0], hi = st[1]; double mn = 1e30, mx = 0;
        for (int k = 0; k < w; ++k) { lo = st[2 * k] < lo ? st[2 * k] : lo; hi = st[2 * k + 1] > hi ? st[2 * k + 1] : hi; double e = (double)(st[2 * k + 1] - st[2 * k]) / 4000; mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        span[w == 8] = (double)(hi - lo) / 4000; wmin[w == 8] = mn; wmax[w == 8] = mx;
    }
    printf("%-64s 1 wave/SIMD: %6.1f cycles/step (ideal 384) | 2 waves/SIMD: whole block %6.1f (ideal 768: %3.0f %% busy), per wave %6.1f .. %6.1f\n", what, span[0], span[1], 76800.0 / span[1], wmin[1], wmax[1]);
}
int main() {
    float *d, *g;
    hipMalloc(&d, ((1 << 20) + 64) * 4); hipMalloc(&g, 64 * 2048 * 4 + 8192 * 4); hipMemset(g, 0, 64 * 2048 * 4 + 8192 * 4);
    hipFuncSetAttribute((const void*)step_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    run<0>(d, g, "12 MFMA only");
    run<1>(d, g, "12 MFMA + 24 SALU clustered at the step end");
    run<2>(d, g, "12 MFMA + 24 SALU spread (2 after every MFMA)");
    run<3>(d, g, "12 MFMA + 24 SALU clustered + 3 ds_read_b128 (one step ahead) + 1 global_load (4 ahead)");
    run<4>(d, g, "12 MFMA + 24 SALU spread + 3 ds_read_b128 (one step ahead) + 1 global_load (4 ahead)");
    run<5>(d, g, "12 MFMA + 24 SALU spread + 3 ds_read_b128 (no global load)");
    run<6>(d, g, "12 MFMA + 24 SALU spread + 1 global_load (no LDS reads)");
    return 0;
}
