// Multi-tensor optimizer step: global-norm gradient clipping + Adam / AdamW + EMA of the parameters in THREE launches for all
// 260 tensors (the reference runs them as torch calls: clip_grad_norm_, optimizer.step(), ema.update() --
// RD/losses.py:29-47, RD/models/ema.py:32-52).  Tensors stay where torch keeps them: the kernels walk a device table of
// per-tensor pointers; a chunk list maps workgroups onto (tensor, offset) so every launch covers all tensors.
// Every reduction has a fixed order: results are run-to-run identical.
#pragma once
#include "common.h"

#define OPT_CHUNK 4096              // elements per workgroup (256 threads x 16)

struct OptSlot { float* p; float* g; float* m; float* v; float* ema; unsigned long long n; };
struct OptChunk { int slot; unsigned off; };

struct OptHyper {
    float lr, beta1, beta2, eps, weight_decay;
    int decoupled_wd;               // 0: Adam (L2 term added to the gradient), 1: AdamW (p *= 1 - lr * wd)
    float step_size;                // lr / (1 - beta1^t)
    float bc2_sqrt;                 // sqrt(1 - beta2^t)
    float max_norm;                 // < 0: no clipping
    float one_minus_beta1, one_minus_beta2;   // formed in double on the host like the python floats torch passes
    float one_minus_ema_decay;      // 1 - min(decay, (1 + n) / (10 + n)) of this update; used where slot.ema != null
    int write_back_grad;            // store the clipped gradient back (what clip_grad_norm_ leaves in p.grad)
};

// launch 1: partial[chunk] = sum of squares of the chunk's gradient elements (fixed tree)
__global__ __launch_bounds__(RDMI_THREADS) void opt_sumsq_kernel(const OptSlot* __restrict__ slots, const OptChunk* __restrict__ chunks,
                                                                  float* __restrict__ partial) {
    float* red = reinterpret_cast<float*>(rdmi_lds);      // [4]
    const OptChunk c = chunks[blockIdx.x];
    const OptSlot s = slots[c.slot];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < OPT_CHUNK / RDMI_THREADS; ++k) {
        const unsigned long long i = (unsigned long long)c.off + (unsigned)(k * RDMI_THREADS + tid);
        if (i < s.n) { const float g = s.g[i]; acc += g * g; }
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// launch 2 (one workgroup): per-tensor norms -> total norm -> clip coefficient  (torch.nn.utils.clip_grad_norm_:
// total = || stack(||g_t||) ||, coef = min(1, max_norm / (total + 1e-6)))
__global__ __launch_bounds__(RDMI_THREADS) void opt_norm_kernel(const float* __restrict__ partial, const int* __restrict__ first_chunk,
                                                                 int nslots, float max_norm, float* __restrict__ out) {
    float* red = reinterpret_cast<float*>(rdmi_lds);      // [RDMI_THREADS]
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int t = tid; t < nslots; t += RDMI_THREADS) {
        float s = 0.f;
        for (int c = first_chunk[t]; c < first_chunk[t + 1]; ++c) s += partial[c];
        const float nt = sqrtf(s);
        acc += nt * nt;
    }
    red[tid] = acc;
    __syncthreads();
    for (int m = RDMI_THREADS / 2; m >= 1; m >>= 1) {
        if (tid < m) red[tid] += red[tid + m];
        __syncthreads();
    }
    if (tid == 0) {
        const float total = sqrtf(red[0]);
        out[0] = total;
        // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total + 1e-6), clamped to 1.  A non-finite total norm must poison the
        // whole step like torch's clamp does (fminf(NaN, 1) would return 1 and only the NaN elements would be hit)
        const float coef = max_norm / (total + 1e-6f);
        out[1] = max_norm >= 0.f ? (coef != coef ? coef : fminf(coef, 1.0f)) : 1.0f;
    }
}

// launch 3: clip, Adam moments, bias-corrected update, EMA -- one pass over every tensor
__global__ __launch_bounds__(RDMI_THREADS) void opt_adam_ema_kernel(const OptSlot* __restrict__ slots, const OptChunk* __restrict__ chunks,
                                                                     OptHyper h, const float* __restrict__ coef_p) {
    const OptChunk c = chunks[blockIdx.x];
    const OptSlot s = slots[c.slot];
    const float coef = coef_p[1];
    const int tid = threadIdx.x;
    for (int k = 0; k < OPT_CHUNK / RDMI_THREADS; ++k) {
        const unsigned long long i = (unsigned long long)c.off + (unsigned)(k * RDMI_THREADS + tid);
        if (i >= s.n) break;
        float g = s.g[i] * coef;                                   // torch._foreach_mul_(grads, clip_coef_clamped)
        if (h.write_back_grad) s.g[i] = g;
        float p = s.p[i];
        if (h.weight_decay != 0.f) {
            if (h.decoupled_wd) p *= 1.0f - h.lr * h.weight_decay;  // AdamW: param.mul_(1 - lr * weight_decay)
            else g += h.weight_decay * p;                           // Adam: grad = grad.add(param, alpha=weight_decay)
        }
        float m = s.m[i], v = s.v[i];
        m = m + (g - m) * h.one_minus_beta1;                        // exp_avg.lerp_(grad, 1 - beta1)
        v = v * h.beta2 + h.one_minus_beta2 * (g * g);              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;        // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        p = p - h.step_size * (m / denom);                         // param.addcdiv_(exp_avg, denom, value=-step_size)
        s.m[i] = m; s.v[i] = v; s.p[i] = p;
        if (s.ema) { const float e = s.ema[i]; s.ema[i] = e - h.one_minus_ema_decay * (e - p); }   // s -= (1 - d) (s - p)
    }
}
