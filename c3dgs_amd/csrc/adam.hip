// adam.hip -- the optimizer step that closes the QAT inner loop (finetune.py:65-66: optimizer.step() of
// torch.optim.Adam(l, lr=0.0, eps=1e-15), scene/gaussian_model.py:296-308), fused for gfx950.
//
// torch's Adam walks its seven parameter tensors with a dozen multi-tensor ("foreach") launches, each a full pass over
// parameters, gradients and both moments: ~3 GB of traffic for the 35 M parameters of a 3M-Gaussian indexed scene. Here
// ONE launch updates every tensor of a step: per element 4 loads + 3 stores (28 B), float4 accesses, the arithmetic of
// torch's _single_tensor_adam in its order:
//     m <- m + (1 - b1) (g - m)                      exp_avg.lerp_(grad, 1 - beta1)
//     v <- v b2 + (1 - b2) g g                       exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound streaming.
#include "common.hpp"

namespace c3dgs {

struct AdamJobs {
    c3dgs_adam_tensor t[C3DGS_ADAM_MAX_TENSORS];
    int first_block[C3DGS_ADAM_MAX_TENSORS];
    int nblocks[C3DGS_ADAM_MAX_TENSORS];
    int n;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float eps,
                                         float neg_step, float bc2_sqrt)
{
    m = w1 < 0.5f ? m + w1 * (g - m) : g - (g - m) * (1.f - w1);          // at::lerp
    v = v * beta2;
    v = v + w2 * g * g;                                                    // addcmul: a + alpha * b * c
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p + neg_step * (m / denom);                                        // addcdiv: a + alpha * (b / c)
}

__global__ void __launch_bounds__(256)
adam_kernel(const AdamJobs jobs, float w1, float beta2, float w2, float eps)
{
    int j = 0;
#pragma unroll
    for (int k = 1; k < C3DGS_ADAM_MAX_TENSORS; k++)
        if (k < jobs.n && (int)blockIdx.x >= jobs.first_block[k]) j = k;
    const c3dgs_adam_tensor t = jobs.t[j];
    const float neg_step = -t.step_size;
    const long long stride = (long long)jobs.nblocks[j] * 256, i0 = (long long)(blockIdx.x - jobs.first_block[j]) * 256 + threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) | reinterpret_cast<uintptr_t>(t.exp_avg) |
                       reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15) == 0;
    const long long n4 = vec ? t.n >> 2 : 0;
    float4* p4 = reinterpret_cast<float4*>(t.param);
    const float4* g4 = reinterpret_cast<const float4*>(t.grad);
    float4* m4 = reinterpret_cast<float4*>(t.exp_avg);
    float4* v4 = reinterpret_cast<float4*>(t.exp_avg_sq);
    for (long long i = i0; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_one(p.x, g.x, m.x, v.x, w1, beta2, w2, eps, neg_step, t.bias_correction2_sqrt);
        adam_one(p.y, g.y, m.y, v.y, w1, beta2, w2, eps, neg_step, t.bias_correction2_sqrt);
        adam_one(p.z, g.z, m.z, v.z, w1, beta2, w2, eps, neg_step, t.bias_correction2_sqrt);
        adam_one(p.w, g.w, m.w, v.w, w1, beta2, w2, eps, neg_step, t.bias_correction2_sqrt);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    for (long long i = (n4 << 2) + i0; i < t.n; i += stride) {
        float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
        adam_one(p, t.grad[i], m, v, w1, beta2, w2, eps, neg_step, t.bias_correction2_sqrt);
        t.param[i] = p; t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
    }
}

void launch_adam(int n_tensors, const c3dgs_adam_tensor* tensors, double beta1, double beta2, double eps, hipStream_t s)
{
    AdamJobs J; J.n = 0;
    int nb = 0;
    for (int k = 0; k < n_tensors; k++) {
        if (tensors[k].n <= 0) continue;
        J.t[J.n] = tensors[k];
        J.first_block[J.n] = nb;
        long long want = (tensors[k].n / 4 + 256 * 4 - 1) / (256 * 4);
        J.nblocks[J.n] = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
        nb += J.nblocks[J.n];
        J.n++;
    }
    if (J.n == 0) return;
    // 1 - beta is formed in double and rounded once, as torch's Python side does (1.f - 0.999f is off by 1.3e-5 relative)
    adam_kernel<<<nb, 256, 0, s>>>(J, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
}

// acc += |g| (the accumulation step of the sensitivity pass, compress.py:110-113): one read of g, one read-modify-write of
// acc instead of torch's abs() temporary + add_ (4 passes over P x 48 floats per camera)
__global__ void __launch_bounds__(256) abs_accumulate_kernel(int64_t n4, int64_t n, const float* __restrict__ g, float* __restrict__ acc)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(g)[i];
        float4 b = reinterpret_cast<float4*>(acc)[i];
        b.x += fabsf(a.x); b.y += fabsf(a.y); b.z += fabsf(a.z); b.w += fabsf(a.w);
        reinterpret_cast<float4*>(acc)[i] = b;
    }
    if (i == 0) for (int64_t k = n4 * 4; k < n; k++) acc[k] += fabsf(g[k]);
}

void launch_abs_accumulate(int64_t n, const float* g, float* acc, hipStream_t s)
{
    if (n <= 0) return;
    const bool al = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(acc)) & 15u) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    const int64_t threads = n4 > 0 ? n4 : 1;
    abs_accumulate_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(n4, n, g, acc);
}

} // namespace c3dgs
