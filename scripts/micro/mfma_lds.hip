// Diagnostic: does LDS / global traffic of the issuing waves cost fp32-MFMA issue slots?  The whole timed loop is hand-written
// assembly (fixed registers), two steps per iteration with ping-pong A buffers -- the NMT = 3 step of unet_kernel.h:
// 12 x v_mfma_f32_16x16x4_f32 on three accumulator chains, A fragments read one step ahead, 2 scalar instructions per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#define S2 "s_mul_i32 s40, s40, 3\n s_add_i32 s40, s40, 7\n"
// MFMA k of a step: tile t = k % 3, element j = k / 3.  A buffers: X = v[64:75], Y = v[76:87]; B = v[88:91]; acc = v[92:103]
#define MF(t, j, base) "v_mfma_f32_16x16x4_f32 v[" #t "], v" #base ", v" #j ", v[" #t "]\n"
#define STEP_MF(a00,a01,a02,a03,a10,a11,a12,a13,a20,a21,a22,a23, MID1, MID2, MID3) \
    "v_mfma_f32_16x16x4_f32 v[92:95], v" #a00 ", v88, v[92:95]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[96:99], v" #a10 ", v88, v[96:99]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[100:103], v" #a20 ", v88, v[100:103]\n" S2 MID1 \
    "v_mfma_f32_16x16x4_f32 v[92:95], v" #a01 ", v89, v[92:95]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[96:99], v" #a11 ", v89, v[96:99]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[100:103], v" #a21 ", v89, v[100:103]\n" S2 MID2 \
    "v_mfma_f32_16x16x4_f32 v[92:95], v" #a02 ", v90, v[92:95]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[96:99], v" #a12 ", v90, v[96:99]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[100:103], v" #a22 ", v90, v[100:103]\n" S2 MID3 \
    "v_mfma_f32_16x16x4_f32 v[92:95], v" #a03 ", v91, v[92:95]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[96:99], v" #a13 ", v91, v[96:99]\n" S2 \
    "v_mfma_f32_16x16x4_f32 v[100:103], v" #a23 ", v91, v[100:103]\n" S2
#define MF_X(M1, M2, M3) STEP_MF(64,65,66,67,68,69,70,71,72,73,74,75, M1, M2, M3)
#define MF_Y(M1, M2, M3) STEP_MF(76,77,78,79,80,81,82,83,84,85,86,87, M1, M2, M3)
#define RDX0 "ds_read_b128 v[64:67], v110\n"
#define RDX1 "ds_read_b128 v[68:71], v110 offset:4352\n"
#define RDX2 "ds_read_b128 v[72:75], v110 offset:8704\n"
#define RDY0 "ds_read_b128 v[76:79], v110 offset:64\n"
#define RDY1 "ds_read_b128 v[80:83], v110 offset:4416\n"
#define RDY2 "ds_read_b128 v[84:87], v110 offset:8768\n"
#define RSX0 "ds_read_b32 v64, v110\n"
#define RSX1 "ds_read_b32 v68, v110 offset:4352\n"
#define RSX2 "ds_read_b32 v72, v110 offset:8704\n"
#define RSY0 "ds_read_b32 v76, v110 offset:64\n"
#define RSY1 "ds_read_b32 v80, v110 offset:4416\n"
#define RSY2 "ds_read_b32 v84, v110 offset:8768\n"
#define VA3 "v_add_u32 v104, v104, v110\n v_add_u32 v105, v105, v110\n v_add_u32 v106, v106, v110\n"
#define VF3 "v_fma_f32 v104, v88, v89, v104\n v_fma_f32 v105, v88, v89, v105\n v_fma_f32 v106, v88, v89, v106\n"
#define GLD "global_load_dwordx4 v[104:107], v[108:109], off\n"
#define PRO "v_mov_b32 v110, %0\n v_mov_b32 v108, %1\n v_mov_b32 v109, %2\n s_mov_b32 s41, %3\n s_mov_b32 s40, 1\n" \
            "v_mov_b32 v88, 1.0\n v_mov_b32 v89, 1.0\n v_mov_b32 v90, 1.0\n v_mov_b32 v91, 1.0\n" RDX0 RDX1 RDX2 RDY0 RDY1 RDY2 "s_waitcnt lgkmcnt(0)\n"
#define EPI "s_sub_i32 s41, s41, 2\n s_cmp_gt_i32 s41, 0\n s_cbranch_scc1 1b\n s_waitcnt vmcnt(0) lgkmcnt(0)\n"
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87", \
    "v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","s40","s41","scc","memory"

template <int MODE>
__global__ __launch_bounds__(512) void k(long long* st, const float* g, int steps) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-6f * i;
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 68 * 4;           // row stride 68 floats like the kernel's activation tensors
    const float* gp = g + threadIdx.x * 4;
    const unsigned glo = (unsigned)(unsigned long long)gp, ghi = (unsigned)((unsigned long long)gp >> 32);
    long long t0 = clock64();
    if (MODE == 0)      // no memory instructions in the loop
        asm volatile(PRO "1:\n" MF_X("","","") MF_Y("","","") EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 1) // next step's three ds_read_b128 at the top of the step (what the compiler emits for unet_kernel.h)
        asm volatile(PRO "1:\n" RDY0 RDY1 RDY2 "s_waitcnt lgkmcnt(3)\n" MF_X("","","") RDX0 RDX1 RDX2 "s_waitcnt lgkmcnt(3)\n" MF_Y("","","") EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 2) // the three reads spread over the step (after MFMA 3, 6, 9); the buffer they fill was last read a step ago
        asm volatile(PRO "1:\n" "s_waitcnt lgkmcnt(0)\n" MF_X(RDY0, RDY1, RDY2) "s_waitcnt lgkmcnt(0)\n" MF_Y(RDX0, RDX1, RDX2) EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 3) // top-of-step reads, 4 bytes per lane instead of 16
        asm volatile(PRO "1:\n" RSY0 RSY1 RSY2 "s_waitcnt lgkmcnt(3)\n" MF_X("","","") RSX0 RSX1 RSX2 "s_waitcnt lgkmcnt(3)\n" MF_Y("","","") EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 4) // mode 1 + one 16-byte global load per step
        asm volatile(PRO "1:\n" RDY0 RDY1 RDY2 "s_waitcnt lgkmcnt(3)\n" MF_X("","","") GLD RDX0 RDX1 RDX2 "s_waitcnt lgkmcnt(3)\n" MF_Y("","","") GLD EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 5) // global load only
        asm volatile(PRO "1:\n" MF_X("","","") GLD MF_Y("","","") GLD EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 6) // reads at the END of the step (after the last MFMA), waited for at the top of the step after next
        asm volatile(PRO "1:\n" MF_X("","","") RDX0 RDX1 RDX2 "s_waitcnt lgkmcnt(3)\n" MF_Y("","","") RDY0 RDY1 RDY2 "s_waitcnt lgkmcnt(3)\n" EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 7) // mode 4 + the compiler's inter-step section: two taken branches and ~16 scalar instructions before the next step's reads
        asm volatile(PRO "1:\n" RDY0 RDY1 RDY2 "s_waitcnt vmcnt(3) lgkmcnt(3)\n" MF_X("","","") GLD "s_cmp_gt_i32 s41, -5\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n" S2 S2 S2 S2 "s_cmp_gt_i32 s41, -7\n s_cbranch_scc1 3f\n s_nop 0\n 3:\n" S2 S2 S2 S2
                     RDX0 RDX1 RDX2 "s_waitcnt vmcnt(3) lgkmcnt(3)\n" MF_Y("","","") GLD "s_cmp_gt_i32 s41, -5\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n" S2 S2 S2 S2 "s_cmp_gt_i32 s41, -7\n s_cbranch_scc1 5f\n s_nop 0\n 5:\n" S2 S2 S2 S2 EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 8) // mode 7 with the waits where the compiler puts them: vmcnt(3) lgkmcnt(5) before MFMA 1, lgkmcnt(4) before 2, lgkmcnt(3) before 3 -- and only 1 scalar pair per MFMA
        asm volatile(PRO "1:\n" RDY0 RDY1 RDY2 "s_waitcnt vmcnt(3) lgkmcnt(3)\n" MF_X("","","") GLD "s_cmp_gt_i32 s41, -5\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n" S2 S2 S2 S2 S2 S2 S2 S2 "s_cmp_gt_i32 s41, -7\n s_cbranch_scc1 3f\n s_nop 0\n 3:\n" S2 S2 S2 S2 S2 S2 S2 S2
                     RDX0 RDX1 RDX2 "s_waitcnt vmcnt(3) lgkmcnt(3)\n" MF_Y("","","") GLD "s_cmp_gt_i32 s41, -5\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n" S2 S2 S2 S2 S2 S2 S2 S2 "s_cmp_gt_i32 s41, -7\n s_cbranch_scc1 5f\n s_nop 0\n 5:\n" S2 S2 S2 S2 S2 S2 S2 S2 EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 9) // reads spread (after MFMA 3, 6, 9) + global load at the step end
        asm volatile(PRO "1:\n" "s_waitcnt vmcnt(3) lgkmcnt(0)\n" MF_X(RDY0, RDY1, RDY2) GLD "s_waitcnt vmcnt(3) lgkmcnt(0)\n" MF_Y(RDX0, RDX1, RDX2) GLD EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 10) // reads after MFMA 3 and 9 (two + one), global load after MFMA 6
        asm volatile(PRO "1:\n" "s_waitcnt vmcnt(3) lgkmcnt(0)\n" MF_X(RDY0 RDY1, GLD, RDY2) "s_waitcnt vmcnt(3) lgkmcnt(0)\n" MF_Y(RDX0 RDX1, GLD, RDX2) EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 11) // global load first (before MFMA 1), reads spread
        asm volatile(PRO "1:\n" "s_waitcnt vmcnt(3) lgkmcnt(0)\n" GLD MF_X(RDY0, RDY1, RDY2) "s_waitcnt vmcnt(3) lgkmcnt(0)\n" GLD MF_Y(RDX0, RDX1, RDX2) EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 12) // no memory instructions, 9 independent v_add_u32 per step (3 after MFMA 3, 6, 9)
        asm volatile(PRO "1:\n" MF_X(VA3, VA3, VA3) MF_Y(VA3, VA3, VA3) EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    else if (MODE == 13) // ... 9 v_fma_f32 per step
        asm volatile(PRO "1:\n" MF_X(VF3, VF3, VF3) MF_Y(VF3, VF3, VF3) EPI :: "v"(la), "v"(glo), "v"(ghi), "s"(steps) : CLOB);
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { st[(threadIdx.x >> 6) * 2] = t0; st[(threadIdx.x >> 6) * 2 + 1] = t1; }
}
template <int MODE> static void run(long long* d, const float* g, const char* what) {
    double span[2], wmin[2], wmax[2];
    const int steps = 4000;
    for (int w : {4, 8}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * w), 64 * 1024, 0, d, g, steps);
        hipDeviceSynchronize();
        long long st[16];
        hipMemcpy(st, d, sizeof(st), hipMemcpyDeviceToHost);
        long long lo = st[0], hi = st[1]; double mn = 1e30, mx = 0;
        for (int i = 0; i < w; ++i) { lo = st[2 * i] < lo ? st[2 * i] : lo; hi = st[2 * i + 1] > hi ? st[2 * i + 1] : hi; double e = (double)(st[2 * i + 1] - st[2 * i]) / steps; mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        span[w == 8] = (double)(hi - lo) / steps; wmin[w == 8] = mn; wmax[w == 8] = mx;
    }
    printf("%-58s 1 wave/SIMD %6.1f cyc/step | 2 waves/SIMD: block %6.1f (%3.0f %% of the pipe), waves %6.1f .. %6.1f\n", what, span[0], span[1], 76800.0 / span[1], wmin[1], wmax[1]);
}
int main() {
    long long* d; float* g;
    hipMalloc(&d, 4096); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    run<0>(d, g, "12 MFMA + 24 SALU, no memory instructions");
    run<1>(d, g, "+ 3 ds_read_b128 at the top of the step");
    run<2>(d, g, "+ 3 ds_read_b128 spread (after MFMA 3, 6, 9)");
    run<3>(d, g, "+ 3 ds_read_b32 at the top of the step");
    run<6>(d, g, "+ 3 ds_read_b128 at the end of the step");
    run<5>(d, g, "+ 1 global_load_dwordx4 per step");
    run<4>(d, g, "+ 3 ds_read_b128 (top) + 1 global_load_dwordx4");
    run<9>(d, g, "+ 3 ds_read_b128 spread + global_load at the step end");
    run<10>(d, g, "+ reads after MFMA 3 (two) and 9, global_load after MFMA 6");
    run<11>(d, g, "+ global_load before MFMA 1, reads spread");
    run<12>(d, g, "no memory, + 9 v_add_u32 per step (spread)");
    run<13>(d, g, "no memory, + 9 v_fma_f32 per step (spread)");
    run<7>(d, g, "  + 2 taken branches + 16 SALU between steps");
    run<8>(d, g, "  + 2 taken branches + 32 SALU between steps");
    return 0;
}
