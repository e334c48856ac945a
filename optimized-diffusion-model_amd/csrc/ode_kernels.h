// Probability-flow ODE sampler (RD/sampling.py:342-392) with the Dormand-Prince RK45 integrator of scipy.integrate.solve_ivp
// evaluated ON THE DEVICE: state y and the seven stage derivatives K live in device memory as float64 (scipy's working
// precision), every stage combination / error estimate / norm is a kernel here, the right-hand side is the HIP score
// network; the host only carries the scalar step-size controller (one 8-byte read-back per attempted step).
// The reference round-trips the whole state through numpy for each of the 6 right-hand sides of a step.
#pragma once
#include "common.h"

// x32 = float(y + h * sum_j a[j] K[j])  (rk_step: dy = np.dot(K[:s].T, a[:s]) * h; fun(t + c h, y + dy)); s == 0: x32 = float(y)
__global__ __launch_bounds__(RDMI_THREADS) void ode_stage_kernel(const double* __restrict__ y, const double* __restrict__ K, long n, int s,
                                                                  double a0, double a1, double a2, double a3, double a4, double a5, double h,
                                                                  double* __restrict__ y_out, float* __restrict__ x32) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= n) return;
    const double a[6] = {a0, a1, a2, a3, a4, a5};
    double dy = 0.0;
    for (int j = 0; j < s; ++j) dy += K[(long)j * n + i] * a[j];
    const double v = s > 0 ? y[i] + dy * h : y[i];
    if (y_out) y_out[i] = v;
    x32[i] = (float)v;
}

// K_out = double( drift(x32, t) * bump(x32) ), drift = 0 - g(t)^2 * score * 0.5 with the CFG combination of the raw network
// output folded in (RD/sde_lib.py:93-101 with probability_flow=True, RD/models/utils.py:124-138, bump: RD/sampling.py:371-375)
__global__ __launch_bounds__(RDMI_THREADS) void ode_rhs_kernel(const float* __restrict__ s2, const float* __restrict__ w, const float* __restrict__ x32,
                                                                double* __restrict__ Kout, int B, int E, float t, float smin, float ratio, float gconst,
                                                                int use_cfg, float moll) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i >= (long)B * E) return;
    float score = s2[i];
    if (use_cfg) { const float wt = w ? w[i / E] : 0.f; score = (1.0f + wt) * score - wt * s2[(long)B * E + i]; }
    const float sigma = smin * powf(ratio, t);
    const float g = sigma * gconst;
    const float drift = 0.0f - ((g * g) * score) * 0.5f;
    const float x = x32[i];
    const float d = 0.5f - x;
    const float bump = moll > 0.f ? expf((-1.0f / (0.25f - d * d) + 4.0f) / moll) : x;
    Kout[i] = (double)(drift * bump);
}

// One workgroup, fixed-order reduction (run-to-run identical): out[slot] = sum_i v_i^2 with
//   mode 0: v = a / (atol + |y0| rtol)                      (select_initial_step d0: a = y0; d1: a = f0)
//   mode 1: v = (a - b) / (atol + |y0| rtol)                (d2: a = f1, b = f0)
//   mode 2: v = h * sum_j E[j] K[j] / (atol + max(|y0|, |b|) rtol)   (error norm of a step: y0 = y, b = y_new)
struct OdeNormArgs { const double* a; const double* b; const double* y0; const double* K; long n; int mode; double atol, rtol, h; double E[7]; double* out; int slot; };
__global__ __launch_bounds__(RDMI_THREADS) void ode_norm_kernel(OdeNormArgs q) {
    double* red = reinterpret_cast<double*>(rdmi_lds);      // [RDMI_THREADS]
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (long i = tid; i < q.n; i += RDMI_THREADS) {
        double v;
        if (q.mode == 2) {
            double e = 0.0;
            for (int j = 0; j < 7; ++j) e += q.K[(long)j * q.n + i] * q.E[j];
            v = e * q.h / (q.atol + fmax(fabs(q.y0[i]), fabs(q.b[i])) * q.rtol);
        } else {
            const double num = q.mode == 1 ? q.a[i] - q.b[i] : q.a[i];
            v = num / (q.atol + fabs(q.y0[i]) * q.rtol);
        }
        acc += v * v;
    }
    red[tid] = acc;
    __syncthreads();
    for (int m = RDMI_THREADS / 2; m >= 1; m >>= 1) {
        if (tid < m) red[tid] += red[tid + m];
        __syncthreads();
    }
    if (tid == 0) q.out[q.slot] = red[0];
}

// y1 = y0 + hd * f0 as float64 and float32 (the Euler probe of select_initial_step), or a plain float32 -> float64 / back copy
__global__ __launch_bounds__(RDMI_THREADS) void ode_axpy_kernel(const double* __restrict__ y, const double* __restrict__ f, double hd, long n,
                                                                 float* __restrict__ x32) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < n) x32[i] = (float)(y[i] + hd * f[i]);
}
__global__ __launch_bounds__(RDMI_THREADS) void ode_f2d_kernel(const float* __restrict__ x, double* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < n) y[i] = (double)x[i];
}
__global__ __launch_bounds__(RDMI_THREADS) void ode_d2f_kernel(const double* __restrict__ y, float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * RDMI_THREADS + threadIdx.x;
    if (i < n) x[i] = (float)y[i];
}
