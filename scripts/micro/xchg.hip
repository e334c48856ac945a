// Micro-benchmark (diagnostic, not part of the product): what does ONE all-gather between the four workgroups of a
// co-operative group cost inside a launch?  Geometry of the fused U-Net's low-resolution section in its co-operative
// form: 256 workgroups of 512 threads, one per CU (160 KB of LDS each), groups of four; per exchange every member
// publishes a [R rows][32 cols] fp32 block as 8-byte {value, tag} granules (agent-scope relaxed atomic stores = sc1,
// MI355X_MICROARCH.md "visibility", form R2) and reads the other three members' blocks with agent-scope relaxed
// atomic loads, re-polling until every tag equals the exchange's epoch.  Two slots per member (parity of the exchange).
//   ./xchg            -> table of (group stride, rows) -> cycles and us per exchange, data checked every word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Args {
    u64* xbuf;            // [group][parity][member][SLOT] granules
    int* info;            // [wg][4]: xcc id, hw id, fails, bad words
    long long* cyc;       // [wg] cycles of the timed loop
    int iters, stride, rows, work;   // work: dependent fma chain length between exchanges (skew generator)
    unsigned salt;
};

constexpr int SLOT = 64 * 32;     // granules per member slot (largest block)

template <int VAR>
__global__ __launch_bounds__(512) void xchg_kernel(Args a) {
    const int tid = threadIdx.x, bid = blockIdx.x;
    const int s = a.stride, blk = bid / (4 * s), r = bid - blk * 4 * s, m = r / s, g = blk * s + (r - m * s);
    float* T = reinterpret_cast<float*>(lds);          // [64][132] tensor, own columns 32m..32m+31
    const int rs = 132;
    for (int i = tid; i < 64 * rs; i += 512) T[i] = 0.f;
    __syncthreads();
    int fails = 0, bad = 0;
    const int ngr = a.rows * 32;                        // granules per block
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    float chain = (float)tid;
    const long long t0 = clock64();
    for (int it = 0; it < a.iters; ++it) {
        const unsigned epoch = a.salt + (unsigned)it + 1u;
        // "compute": own columns of the tensor for this exchange (value encodes (it, m, row, col))
        for (int i = tid; i < ngr; i += 512) {
            const int row = i >> 5, c = i & 31;
            T[row * rs + 32 * m + c] = (float)((it & 1023) * 8 + m) + (float)i * (1.0f / 4096.0f);
        }
        for (int k = 0; k < a.work * (1 + ((bid >> 5) & 1)); ++k) chain = chain * 1.0000001f + 0.5f;     // uneven load
        __syncthreads();
        u64* base = a.xbuf + ((size_t)(g * 2 + (it & 1)) * 4) * SLOT;
        if (VAR == 0) {
        // publish
        for (int i = tid; i < ngr; i += 512) {
            const int row = i >> 5, c = i & 31;
            const unsigned v = __builtin_bit_cast(unsigned, T[row * rs + 32 * m + c]);
            __hip_atomic_store(base + (size_t)m * SLOT + i, ((u64)epoch << 32) | v, RLX_AGENT);
        }
        // receive the other three blocks: four granules of ONE member in flight, re-poll the late ones
        for (int jj = 1; jj < 4; ++jj) {
            const int j = (m + jj) & 3;
            const u64* src = base + (size_t)j * SLOT;
            for (int i0 = tid; i0 < ngr; i0 += 512 * 4) {
                u64 x[4];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + k * 512;
                        x[k] = i < ngr ? __hip_atomic_load(src + i, RLX_AGENT) : ((u64)epoch << 32);
                        ok &= (unsigned)(x[k] >> 32) == epoch;
                    }
                    if (ok) break;
                    if (++spins > (1u << 22)) { ++fails; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 512;
                    if (i < ngr) T[(i >> 5) * rs + 32 * j + (i & 31)] = __builtin_bit_cast(float, (unsigned)x[k]);
                }
            }
        }
        } else if (VAR == 1) {
        // 8-byte granules, ALL of this thread's granules (3 members x ngr/512) in flight at once
        for (int i = tid; i < ngr; i += 512) {
            const unsigned v = __builtin_bit_cast(unsigned, T[(i >> 5) * rs + 32 * m + (i & 31)]);
            __hip_atomic_store(base + (size_t)m * SLOT + i, ((u64)epoch << 32) | v, RLX_AGENT);
        }
        const int per = (ngr + 511) / 512;                  // 1 (rows 16) or 4 (rows 64)
        u64 x[12];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int jj = k / 4 + 1, kk = k & 3, i = tid + kk * 512;
                const bool live = kk < per && i < ngr;
                x[k] = live ? __hip_atomic_load(base + (size_t)((m + jj) & 3) * SLOT + i, RLX_AGENT) : ((u64)epoch << 32);
                ok &= (unsigned)(x[k] >> 32) == epoch;
            }
            if (ok) break;
            if (++spins > (1u << 22)) { ++fails; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int jj = k / 4 + 1, kk = k & 3, i = tid + kk * 512;
            if (kk < per && i < ngr) T[(i >> 5) * rs + 32 * ((m + jj) & 3) + (i & 31)] = __builtin_bit_cast(float, (unsigned)x[k]);
        }
        } else {
        // 16-byte stores / loads of granule PAIRS {v0, tag, v1, tag}, all in flight at once
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 __attribute__((address_space(1))) * gq_t;
        const int npair = ngr / 2, perp = (npair + 511) / 512;      // 1 or 2
        for (int p = tid; p < npair; p += 512) {
            const int i = 2 * p;
            const float* sp = &T[(i >> 5) * rs + 32 * m + (i & 31)];
            u32x4 q = {__builtin_bit_cast(unsigned, sp[0]), epoch, __builtin_bit_cast(unsigned, sp[1]), epoch};
            gq_t dst = (gq_t)(base + (size_t)m * SLOT + i);
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(q) : "memory");
        }
        u32x4 y[6];
        unsigned spins = 0;
        for (;;) {
            // six loads back to back, one wait
            gq_t p0 = (gq_t)(base + (size_t)((m + 1) & 3) * SLOT + 2 * tid), p1 = (gq_t)(base + (size_t)((m + 2) & 3) * SLOT + 2 * tid), p2 = (gq_t)(base + (size_t)((m + 3) & 3) * SLOT + 2 * tid);
            if (perp == 2) {
                gq_t q0 = p0 + 512, q1 = p1 + 512, q2 = p2 + 512;
                asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\tglobal_load_dwordx4 %2, %8, off sc1\n\t"
                             "global_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %11, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]) : "v"(p0), "v"(p1), "v"(p2), "v"(q0), "v"(q1), "v"(q2) : "memory");
            } else {
                const bool live = tid < npair;
                if (live) asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]) : "v"(p0), "v"(p1), "v"(p2) : "memory");
                else { y[0] = y[1] = y[2] = u32x4{0, epoch, 0, epoch}; }
                y[3] = y[4] = y[5] = u32x4{0, epoch, 0, epoch};
            }
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 6; ++k) ok &= y[k][1] == epoch && y[k][3] == epoch;
            if (ok) break;
            if (++spins > (1u << 22)) { ++fails; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int jj = k % 3 + 1, hh = k / 3, p = tid + hh * 512, i = 2 * p;
            if (hh < perp && p < npair) { float* dp = &T[(i >> 5) * rs + 32 * ((m + jj) & 3) + (i & 31)]; dp[0] = __builtin_bit_cast(float, y[k][0]); dp[1] = __builtin_bit_cast(float, y[k][2]); }
        }
        }
        __syncthreads();
        // check every word of the assembled tensor
        for (int i = tid; i < ngr * 4; i += 512) {
            const int j = i / ngr, q = i - j * ngr;
            const float want = (float)((it & 1023) * 8 + j) + (float)q * (1.0f / 4096.0f);
            if (T[(q >> 5) * rs + 32 * j + (q & 31)] != want) ++bad;
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (chain == 1.2345f) T[0] = chain;
    // reduce fails / bad over the workgroup through LDS
    __syncthreads();
    int* red = reinterpret_cast<int*>(lds);
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    if (fails) atomicAdd(&red[0], fails);
    if (bad) atomicAdd(&red[1], bad);
    __syncthreads();
    if (tid == 0) {
        a.info[bid * 4 + 0] = (int)(xcc & 15); a.info[bid * 4 + 1] = (int)hwid; a.info[bid * 4 + 2] = red[0]; a.info[bid * 4 + 3] = red[1];
        a.cyc[bid] = t1 - t0;
    }
}

// the same loop without the exchange: what the compute + barriers + check cost alone
__global__ __launch_bounds__(512) void base_kernel(Args a) {
    const int tid = threadIdx.x, bid = blockIdx.x, m = bid & 3;
    float* T = reinterpret_cast<float*>(lds);
    const int rs = 132, ngr = a.rows * 32;
    int bad = 0;
    float chain = (float)tid;
    const long long t0 = clock64();
    for (int it = 0; it < a.iters; ++it) {
        for (int i = tid; i < ngr; i += 512) T[(i >> 5) * rs + 32 * m + (i & 31)] = (float)((it & 1023) * 8 + m) + (float)i * (1.0f / 4096.0f);
        for (int k = 0; k < a.work * (1 + ((bid >> 5) & 1)); ++k) chain = chain * 1.0000001f + 0.5f;
        __syncthreads();
        __syncthreads();
        for (int i = tid; i < ngr * 4; i += 512) {
            const int j = i / ngr, q = i - j * ngr;
            if (j == m && T[(q >> 5) * rs + 32 * j + (q & 31)] != (float)((it & 1023) * 8 + j) + (float)q * (1.0f / 4096.0f)) ++bad;
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (chain == 1.2345f || bad) T[0] = chain;
    if (tid == 0) a.cyc[bid] = t1 - t0;
}

int main() {
    const int grid = 256;
    Args a{};
    const size_t xb = (size_t)(grid / 4) * 2 * 4 * SLOT * sizeof(u64);
    hipMalloc(&a.xbuf, xb); hipMemset(a.xbuf, 0, xb);
    hipMalloc(&a.info, grid * 4 * sizeof(int)); hipMalloc(&a.cyc, grid * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(xchg_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(xchg_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(xchg_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(base_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<int> info(grid * 4); std::vector<long long> cyc(grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned salt = 0;
    a.iters = 200;
    for (int var : {0, 1, 2})
    for (int work : {0, 40}) {
        for (int stride : {8, 1}) {
            for (int rows : {16, 64}) {
                a.stride = stride; a.rows = rows; a.work = work;
                float ms_b = 0, ms_x = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0); hipLaunchKernelGGL(base_kernel, dim3(grid), dim3(512), 150 * 1024, 0, a); hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms_b, e0, e1);
                }
                long long cb = 0;
                hipMemcpy(cyc.data(), a.cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
                for (auto c : cyc) cb = std::max(cb, c);
                for (int rep = 0; rep < 2; ++rep) {
                    a.salt = salt; salt += (unsigned)a.iters + 7u;
                    hipEventRecord(e0);
                    if (var == 0) hipLaunchKernelGGL(xchg_kernel<0>, dim3(grid), dim3(512), 150 * 1024, 0, a);
                    else if (var == 1) hipLaunchKernelGGL(xchg_kernel<1>, dim3(grid), dim3(512), 150 * 1024, 0, a);
                    else hipLaunchKernelGGL(xchg_kernel<2>, dim3(grid), dim3(512), 150 * 1024, 0, a);
                    hipEventRecord(e1);
                    if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
                    hipEventElapsedTime(&ms_x, e0, e1);
                }
                hipMemcpy(info.data(), a.info, grid * 4 * sizeof(int), hipMemcpyDeviceToHost);
                hipMemcpy(cyc.data(), a.cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
                long long cx = 0; int fails = 0, bad = 0, mixed = 0;
                for (auto c : cyc) cx = std::max(cx, c);
                for (int b = 0; b < grid; ++b) { fails += info[b * 4 + 2]; bad += info[b * 4 + 3]; }
                for (int b = 0; b < grid; ++b) {          // groups whose members sit on different XCDs
                    const int s = stride, blk = b / (4 * s), r = b - blk * 4 * s, m = r / s;
                    if (m == 0) for (int j = 1; j < 4; ++j) if (info[(b + j * s) * 4] != info[b * 4]) { ++mixed; break; }
                }
                printf("var=%d work=%3d stride=%2d rows=%2d: exchange %.2f us (%lld cycles) over a %.2f us base iteration; spin give-ups %d, bad words %d, groups spanning XCDs %d/64\n",
                       var, work, stride, rows, (ms_x - ms_b) * 1e3 / a.iters, (cx - cb) / a.iters, ms_b * 1e3 / a.iters, fails, bad, mixed);
                fflush(stdout);
            }
        }
    }
    printf("xcc ids of workgroups 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %d", info[b * 4]);
    printf("\n");
    return 0;
}
