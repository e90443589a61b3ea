// qat.hip -- the QAT getters in front of every raster call, fused for gfx950.
//
//   reference: scene/gaussian_model.py:54-77 (activations), :109-118 (FakeQuantize(dtype=qint8) modules),
//              :213-267 (getters), :851-862 ([visible] gathers in render), :1405-1414 (FakeQuantizationHalf);
//   semantics of the modules themselves: torch.ao.quantization.FakeQuantize.forward (observer, then
//   fake_quantize_per_tensor_affine), MovingAverageMinMaxObserver.forward, _calculate_qparams (per-tensor affine).
//
// What the reference does per view: 7 module calls x (aminmax + ~10 scalar kernels + 2 host reads for
// float(scale)/int(zero_point)) + activations + 6 boolean-mask gathers (each a nonzero with a host sync) -- about
// a hundred launches and twenty syncs around ~100 MB of real traffic. Here the observer state lives in device
// memory (16 B per module) and is never read back:
//   observe     ONE streaming pass over all raw tensors -> per-block min/max of the activated values,
//   finalize    one tiny workgroup: moving-average update + scale / zero_point for all six modules,
//   codebooks   ONE launch: fq(normalize(relu(scaling))), normalize(fq(rotation)), cat(fq(dc), fq(rest)),
//   visible     frustum flags on the half-rounded positions + rocPRIM exclusive scan (row of every visible point),
//   points      ONE launch: compacted means3D / opacities / scale_factors / index rows,
// and two backward launches (points, codebooks) that recompute the straight-through masks from the raw tensors.
// All of it is HBM streaming: 4-16 B per element, float4 accesses where the layout allows.
//
// Compiled with -ffp-contract=off: every expression is evaluated as written (torch's kernels do not fuse either).
#include "common.hpp"
#include "gsmath.hpp"
#include <hip/hip_fp16.h>
#include <cfloat>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace c3dgs {

constexpr int OBS_MAXB = 512;      // partial min/max slots per tensor
constexpr float NORM_EPS = 1e-12f; // torch.nn.functional.normalize default eps

enum { ACT_IDENT = 0, ACT_SIGMOID = 1, ACT_NORMRELU3 = 2 };

struct ObsJob { const float* x; long long units; int act; int first_block; int nblocks; int slot; };
struct ObsJobs { ObsJob j[C3DGS_FQ_COUNT]; int n; };

// torch.sigmoid: 1 / (1 + exp(-x)) in fp32
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// normalize(relu(x)) of one 3-vector (gaussian_model.py:68: normalize(relu(x)), F.normalize eps 1e-12)
__device__ __forceinline__ void normrelu3(const float x[3], float u[3], float v[3], float& nrm)
{
    u[0] = fmaxf(x[0], 0.f); u[1] = fmaxf(x[1], 0.f); u[2] = fmaxf(x[2], 0.f);
    nrm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float d = fmaxf(nrm, NORM_EPS);
    v[0] = u[0] / d; v[1] = u[1] / d; v[2] = u[2] / d;
}

struct Fq { float scale, inv_scale, zp; bool on; };
__device__ __forceinline__ Fq load_fq(const c3dgs_fq_state* st, int slot, int enabled)
{
    Fq f;
    f.scale = st[slot].scale;
    f.inv_scale = 1.0f / f.scale;
    f.zp = (float)st[slot].zero_point;
    f.on = enabled != 0;
    return f;
}
// fake_quantize_per_tensor_affine (ATen fake_quantize_core: nearbyint(x * inv_scale) + zp, clamp, (q - zp) * scale);
// mask = quantised value inside [-128, 127] (the straight-through gradient mask)
__device__ __forceinline__ float fq_apply(const Fq& f, float v, bool& mask)
{
    if (!f.on) { mask = true; return v; }
    const float q = nearbyintf(v * f.inv_scale) + f.zp;
    mask = q >= -128.f && q <= 127.f;
    return (fminf(127.f, fmaxf(-128.f, q)) - f.zp) * f.scale;
}
__device__ __forceinline__ float fq_val(const Fq& f, float v) { bool m; return fq_apply(f, v, m); }

__device__ __forceinline__ float half_round(float v) { return __half2float(__float2half_rn(v)); }

// ------------------------------------------------------------------------------------------- observer
__global__ void __launch_bounds__(256)
qat_observe_kernel(const ObsJobs jobs, float* __restrict__ ws)
{
    int t = 0;
#pragma unroll
    for (int k = 1; k < C3DGS_FQ_COUNT; k++)
        if (k < jobs.n && (int)blockIdx.x >= jobs.j[k].first_block) t = k;
    const ObsJob job = jobs.j[t];
    const int b = blockIdx.x - job.first_block;
    const long long stride = (long long)job.nblocks * 256;
    float lo = INFINITY, hi = -INFINITY;
    if (job.act == ACT_NORMRELU3) {
        for (long long r = (long long)b * 256 + threadIdx.x; r < job.units; r += stride) {
            const float x[3] = { job.x[3 * r], job.x[3 * r + 1], job.x[3 * r + 2] };
            float u[3], v[3], n;
            normrelu3(x, u, v, n);
            lo = fminf(lo, fminf(v[0], fminf(v[1], v[2])));
            hi = fmaxf(hi, fmaxf(v[0], fmaxf(v[1], v[2])));
        }
    } else {
        // 16-byte loads when the base is aligned (whole torch allocations are), scalar loads for the remainder
        const long long n4 = (reinterpret_cast<uintptr_t>(job.x) & 15) == 0 ? job.units >> 2 : 0;
        const float4* x4 = reinterpret_cast<const float4*>(job.x);
        for (long long i = (long long)b * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = x4[i];
            if (job.act == ACT_SIGMOID) { v.x = sigmoid_f(v.x); v.y = sigmoid_f(v.y); v.z = sigmoid_f(v.z); v.w = sigmoid_f(v.w); }
            lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
            hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        for (long long i = (n4 << 2) + (long long)b * 256 + threadIdx.x; i < job.units; i += stride) {
            float v = job.x[i];
            if (job.act == ACT_SIGMOID) v = sigmoid_f(v);
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    __shared__ float s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[(job.slot * OBS_MAXB + b) * 2] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
        ws[(job.slot * OBS_MAXB + b) * 2 + 1] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    }
}

// MovingAverageMinMaxObserver.forward + _calculate_qparams (per_tensor_affine, qint8) for every observed module
__global__ void __launch_bounds__(64 * C3DGS_FQ_COUNT)
qat_finalize_kernel(const ObsJobs jobs, const float* __restrict__ ws, c3dgs_fq_state* __restrict__ state, float c)
{
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;          // one wave per observed module
    if (t >= jobs.n) return;
    const ObsJob job = jobs.j[t];
    float lo = INFINITY, hi = -INFINITY;
    for (int b = lane; b < job.nblocks; b += 64) {
        lo = fminf(lo, ws[(job.slot * OBS_MAXB + b) * 2]);
        hi = fmaxf(hi, ws[(job.slot * OBS_MAXB + b) * 2 + 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (lane == 0) {
        c3dgs_fq_state s = state[job.slot];
        if (s.min_val == INFINITY && s.max_val == -INFINITY) { s.min_val = lo; s.max_val = hi; }
        else {
            s.min_val = s.min_val + c * (lo - s.min_val);
            s.max_val = s.max_val + c * (hi - s.max_val);
        }
        if (s.min_val == INFINITY && s.max_val == -INFINITY) { s.scale = 1.0f; s.zero_point = 0; }  // check_min_max_valid
        else {
            const float min_neg = fminf(s.min_val, 0.f), max_pos = fmaxf(s.max_val, 0.f);
            s.scale = fmaxf((max_pos - min_neg) / 255.0f, FLT_EPSILON);
            const int zp = -128 - (int)rintf(min_neg / s.scale);
            s.zero_point = min(127, max(-128, zp));
        }
        state[job.slot] = s;
    }
}

static void add_job(ObsJobs& J, int& nb, const float* x, long long units, int act, int slot, int per_thread)
{
    if (!x || units <= 0) return;
    ObsJob& j = J.j[J.n++];
    j.x = x; j.units = units; j.act = act; j.slot = slot; j.first_block = nb;
    const long long work = act == ACT_NORMRELU3 ? units : (units + 3) / 4;
    long long want = (work + 256LL * per_thread - 1) / (256LL * per_thread);
    j.nblocks = (int)(want < 1 ? 1 : (want > OBS_MAXB ? OBS_MAXB : want));
    nb += j.nblocks;
}

size_t qat_workspace_bytes() { return (size_t)C3DGS_FQ_COUNT * OBS_MAXB * 2 * sizeof(float); }

void launch_qat_observe(const c3dgs_qat_params& q, void* workspace, hipStream_t s)
{
    ObsJobs J; J.n = 0;
    int nb = 0;
    const long long P = q.P, GS = q.GS, SHS = q.SHS;
    if (q.observer_enabled[C3DGS_FQ_OPACITY]) add_job(J, nb, q.opacity, P, ACT_SIGMOID, C3DGS_FQ_OPACITY, 4);
    if (q.observer_enabled[C3DGS_FQ_SCALING]) add_job(J, nb, q.scaling, GS, ACT_NORMRELU3, C3DGS_FQ_SCALING, 4);
    if (q.observer_enabled[C3DGS_FQ_SCALING_FACTOR]) add_job(J, nb, q.scaling_factor, P, ACT_IDENT, C3DGS_FQ_SCALING_FACTOR, 4);
    if (q.observer_enabled[C3DGS_FQ_ROTATION]) add_job(J, nb, q.rotation, GS * 4, ACT_IDENT, C3DGS_FQ_ROTATION, 4);
    if (q.observer_enabled[C3DGS_FQ_FEATURES_DC]) add_job(J, nb, q.features_dc, SHS * 3, ACT_IDENT, C3DGS_FQ_FEATURES_DC, 4);
    if (q.observer_enabled[C3DGS_FQ_FEATURES_REST])
        add_job(J, nb, q.features_rest, SHS * 3 * (q.M - 1), ACT_IDENT, C3DGS_FQ_FEATURES_REST, 4);
    if (J.n == 0) return;
    qat_observe_kernel<<<nb, 256, 0, s>>>(J, (float*)workspace);
    qat_finalize_kernel<<<1, 64 * C3DGS_FQ_COUNT, 0, s>>>(J, (const float*)workspace, q.state, q.averaging_constant);
}

// ------------------------------------------------------------------------------------------- codebooks
struct CbJobs { int first_block[3]; int nblocks[3]; };

__global__ void __launch_bounds__(256)
qat_codebooks_kernel(const c3dgs_qat_params q, const CbJobs jobs, float* __restrict__ scales_n, float* __restrict__ rotations,
                     float* __restrict__ shs)
{
    const int t = (int)blockIdx.x >= jobs.first_block[2] ? 2 : ((int)blockIdx.x >= jobs.first_block[1] ? 1 : 0);
    const int b = blockIdx.x - jobs.first_block[t];
    const long long stride = (long long)jobs.nblocks[t] * 256, i0 = (long long)b * 256 + threadIdx.x;
    if (t == 0) {          // get_scaling_normalized: fq(normalize(relu(_scaling)))
        const Fq f = load_fq(q.state, C3DGS_FQ_SCALING, q.fake_quant_enabled[C3DGS_FQ_SCALING]);
        for (long long r = i0; r < q.GS; r += stride) {
            const float x[3] = { q.scaling[3 * r], q.scaling[3 * r + 1], q.scaling[3 * r + 2] };
            float u[3], v[3], n;
            normrelu3(x, u, v, n);
            scales_n[3 * r] = fq_val(f, v[0]); scales_n[3 * r + 1] = fq_val(f, v[1]); scales_n[3 * r + 2] = fq_val(f, v[2]);
        }
    } else if (t == 1) {   // _rotation_post_activation: normalize(fq(_rotation))
        const Fq f = load_fq(q.state, C3DGS_FQ_ROTATION, q.fake_quant_enabled[C3DGS_FQ_ROTATION]);
        const float4* x4 = reinterpret_cast<const float4*>(q.rotation);
        float4* o4 = reinterpret_cast<float4*>(rotations);
        for (long long r = i0; r < q.GS; r += stride) {
            float4 w = x4[r];
            w.x = fq_val(f, w.x); w.y = fq_val(f, w.y); w.z = fq_val(f, w.z); w.w = fq_val(f, w.w);
            const float d = fmaxf(sqrtf(w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w), NORM_EPS);
            o4[r] = make_float4(w.x / d, w.y / d, w.z / d, w.w / d);
        }
    } else {               // _get_features_raw: cat(fq_dc(dc), fq_rest(rest), dim=1), one output element per step
        const Fq fd = load_fq(q.state, C3DGS_FQ_FEATURES_DC, q.fake_quant_enabled[C3DGS_FQ_FEATURES_DC]);
        const Fq fr = load_fq(q.state, C3DGS_FQ_FEATURES_REST, q.fake_quant_enabled[C3DGS_FQ_FEATURES_REST]);
        const int row = q.M * 3, rrow = row - 3;
        const long long total = (long long)q.SHS * row;
        if ((row & 3) == 0) {                                     // 4 consecutive outputs per thread, one 16-B store
            float4* o4 = reinterpret_cast<float4*>(shs);
            for (long long c = i0; c < (total >> 2); c += stride) {
                const long long e = c << 2, sidx = e / row;
                const int r0 = (int)(e - sidx * row);
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int rr = r0 + k;
                    o[k] = rr < 3 ? fq_val(fd, q.features_dc[sidx * 3 + rr]) : fq_val(fr, q.features_rest[sidx * rrow + rr - 3]);
                }
                o4[c] = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (long long e = i0; e < total; e += stride) {
                const long long sidx = e / row;
                const int rr = (int)(e - sidx * row);
                shs[e] = rr < 3 ? fq_val(fd, q.features_dc[sidx * 3 + rr]) : fq_val(fr, q.features_rest[sidx * rrow + rr - 3]);
            }
        }
    }
}

static CbJobs cb_jobs(const c3dgs_qat_params& q, bool j0, bool j1, bool j2, int& nb)
{
    CbJobs J;
    nb = 0;
    auto blocks = [](long long units, int per_thread) {
        long long want = (units + 256LL * per_thread - 1) / (256LL * per_thread);
        return (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    };
    const long long feat = (long long)q.SHS * q.M * 3;
    const int n[3] = { j0 && q.GS > 0 ? blocks(q.GS, 2) : 0, j1 && q.GS > 0 ? blocks(q.GS, 2) : 0,
                       j2 && q.SHS > 0 ? blocks(((q.M * 3) & 3) == 0 ? feat / 4 : feat, 4) : 0 };
    for (int t = 0; t < 3; t++) { J.first_block[t] = nb; J.nblocks[t] = n[t]; nb += n[t]; }
    // empty jobs must never be selected: push their first_block past the grid (selection tests t=2 first, then t=1)
    if (n[2] == 0) J.first_block[2] = 0x7fffffff;
    if (n[1] == 0) J.first_block[1] = n[2] ? J.first_block[2] : 0x7fffffff;
    return J;
}

void launch_qat_codebooks(const c3dgs_qat_params& q, float* scales_n, float* rotations, float* shs, hipStream_t s)
{
    int nb;
    const CbJobs J = cb_jobs(q, q.scaling && scales_n, q.rotation && rotations, q.features_dc && shs, nb);
    if (nb == 0) return;
    qat_codebooks_kernel<<<nb, 256, 0, s>>>(q, J, scales_n, rotations, shs);
}

__global__ void __launch_bounds__(256)
qat_codebooks_backward_kernel(const c3dgs_qat_params q, const CbJobs jobs, const float* __restrict__ g_scales,
                              const float* __restrict__ g_rot, const float* __restrict__ g_shs, float* __restrict__ d_scaling,
                              float* __restrict__ d_rotation, float* __restrict__ d_dc, float* __restrict__ d_rest)
{
    const int t = (int)blockIdx.x >= jobs.first_block[2] ? 2 : ((int)blockIdx.x >= jobs.first_block[1] ? 1 : 0);
    const int b = blockIdx.x - jobs.first_block[t];
    const long long stride = (long long)jobs.nblocks[t] * 256, i0 = (long long)b * 256 + threadIdx.x;
    if (t == 0) {
        // y = fq(v), v = u / max(|u|, eps), u = relu(x):  dv = g * mask;  du = dv/d - [|u| >= eps] u (u.dv) / (|u| d^2)
        const Fq f = load_fq(q.state, C3DGS_FQ_SCALING, q.fake_quant_enabled[C3DGS_FQ_SCALING]);
        for (long long r = i0; r < q.GS; r += stride) {
            const float x[3] = { q.scaling[3 * r], q.scaling[3 * r + 1], q.scaling[3 * r + 2] };
            float u[3], v[3], n, dv[3];
            normrelu3(x, u, v, n);
#pragma unroll
            for (int k = 0; k < 3; k++) { bool m; (void)fq_apply(f, v[k], m); dv[k] = m ? g_scales[3 * r + k] : 0.f; }
            const float d = fmaxf(n, NORM_EPS);
            const float dot = u[0] * dv[0] + u[1] * dv[1] + u[2] * dv[2];
            const float k2 = (n >= NORM_EPS && n > 0.f) ? dot / (n * d * d) : 0.f;   // clamp_min passes the gradient at norm >= eps
#pragma unroll
            for (int k = 0; k < 3; k++) d_scaling[3 * r + k] = x[k] > 0.f ? dv[k] / d - u[k] * k2 : 0.f;
        }
    } else if (t == 1) {
        // y = w / max(|w|, eps), w = fq(x):  dw = g/d - [|w| >= eps] w (w.g) / (|w| d^2);  dx = dw * mask
        const Fq f = load_fq(q.state, C3DGS_FQ_ROTATION, q.fake_quant_enabled[C3DGS_FQ_ROTATION]);
        const float4* x4 = reinterpret_cast<const float4*>(q.rotation);
        const float4* g4 = reinterpret_cast<const float4*>(g_rot);
        float4* o4 = reinterpret_cast<float4*>(d_rotation);
        for (long long r = i0; r < q.GS; r += stride) {
            const float4 xv = x4[r], g = g4[r];
            bool m0, m1, m2, m3;
            const float w0 = fq_apply(f, xv.x, m0), w1 = fq_apply(f, xv.y, m1), w2 = fq_apply(f, xv.z, m2), w3 = fq_apply(f, xv.w, m3);
            const float n = sqrtf(w0 * w0 + w1 * w1 + w2 * w2 + w3 * w3), d = fmaxf(n, NORM_EPS);
            const float dot = w0 * g.x + w1 * g.y + w2 * g.z + w3 * g.w;
            const float k2 = (n >= NORM_EPS && n > 0.f) ? dot / (n * d * d) : 0.f;
            o4[r] = make_float4(m0 ? g.x / d - w0 * k2 : 0.f, m1 ? g.y / d - w1 * k2 : 0.f, m2 ? g.z / d - w2 * k2 : 0.f,
                                m3 ? g.w / d - w3 * k2 : 0.f);
        }
    } else {
        const Fq fd = load_fq(q.state, C3DGS_FQ_FEATURES_DC, q.fake_quant_enabled[C3DGS_FQ_FEATURES_DC]);
        const Fq fr = load_fq(q.state, C3DGS_FQ_FEATURES_REST, q.fake_quant_enabled[C3DGS_FQ_FEATURES_REST]);
        const int row = q.M * 3, rrow = row - 3;
        const long long total = (long long)q.SHS * row;
        const bool vec = (row & 3) == 0;
        const long long steps = vec ? total >> 2 : total;
        for (long long c = i0; c < steps; c += stride) {
            const long long e = vec ? c << 2 : c, sidx = e / row;
            const int r0 = (int)(e - sidx * row);
            float g[4];
            if (vec) { const float4 gv = reinterpret_cast<const float4*>(g_shs)[c]; g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w; }
            else g[0] = g_shs[e];
            for (int k = 0; k < (vec ? 4 : 1); k++) {
                const int rr = r0 + k;
                bool m;
                if (rr < 3) { (void)fq_apply(fd, q.features_dc[sidx * 3 + rr], m); d_dc[sidx * 3 + rr] = m ? g[k] : 0.f; }
                else { (void)fq_apply(fr, q.features_rest[sidx * rrow + rr - 3], m); d_rest[sidx * rrow + rr - 3] = m ? g[k] : 0.f; }
            }
        }
    }
}

void launch_qat_codebooks_backward(const c3dgs_qat_params& q, const float* g_scales, const float* g_rot, const float* g_shs,
                                   float* d_scaling, float* d_rotation, float* d_dc, float* d_rest, hipStream_t s)
{
    int nb;
    const CbJobs J = cb_jobs(q, q.scaling && g_scales && d_scaling, q.rotation && g_rot && d_rotation,
                             q.features_dc && g_shs && d_dc, nb);
    if (nb == 0) return;
    qat_codebooks_backward_kernel<<<nb, 256, 0, s>>>(q, J, g_scales, g_rot, g_shs, d_scaling, d_rotation, d_dc, d_rest);
}

// ------------------------------------------------------------------------------------------- visibility + points
__global__ void __launch_bounds__(256)
qat_visible_kernel(int P, const float* __restrict__ xyz, int half_xyz, const float* __restrict__ view, uint8_t* __restrict__ visible)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    f3 p = { xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2] };
    if (half_xyz) { p.x = half_round(p.x); p.y = half_round(p.y); p.z = half_round(p.z); }
    const f3 pv = xform4x3(p, view);
    visible[i] = !(pv.z <= 0.01f) ? 1 : 0;     // same test as mark_visible_kernel (preprocess.hip)
}

struct FlagToInt {
    __host__ __device__ int32_t operator()(uint8_t f) const { return f ? 1 : 0; }
};
using FlagIt = rocprim::transform_iterator<const uint8_t*, FlagToInt, int32_t>;

size_t qat_scan_bytes(int P)
{
    size_t bytes = 0;
    FlagIt it((const uint8_t*)nullptr, FlagToInt{});
    (void)rocprim::exclusive_scan(nullptr, bytes, it, (int32_t*)nullptr, (int32_t)0, (size_t)(P > 0 ? P : 1), rocprim::plus<int32_t>());
    return bytes < 256 ? 256 : bytes;
}

__global__ void qat_count_kernel(int P, const uint8_t* __restrict__ visible, const int32_t* __restrict__ rank, int32_t* __restrict__ count)
{
    count[0] = rank[P - 1] + (visible[P - 1] ? 1 : 0);
}

hipError_t run_qat_visible(const c3dgs_qat_params& q, const float* view, uint8_t* visible, int32_t* rank, int32_t* count,
                           void* scan_ws, hipStream_t s)
{
    qat_visible_kernel<<<(q.P + 255) / 256, 256, 0, s>>>(q.P, q.xyz, q.half_xyz, view, visible);
    size_t bytes = qat_scan_bytes(q.P);
    FlagIt it(visible, FlagToInt{});
    hipError_t e = rocprim::exclusive_scan(scan_ws, bytes, it, rank, (int32_t)0, (size_t)q.P, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return e;
    qat_count_kernel<<<1, 1, 0, s>>>(q.P, visible, rank, count);
    return hipSuccess;
}

__global__ void __launch_bounds__(256)
qat_points_kernel(const c3dgs_qat_params q, const uint8_t* __restrict__ visible, const int32_t* __restrict__ rank,
                  const int64_t* __restrict__ sh_idx, const int64_t* __restrict__ g_idx, float* __restrict__ means3D,
                  float* __restrict__ opac, float* __restrict__ sfac, int64_t* __restrict__ sh_out, int64_t* __restrict__ g_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= q.P) return;
    if (visible && !visible[i]) return;
    const size_t j = visible ? (size_t)rank[i] : (size_t)i;
    if (means3D && q.xyz) {
        float x = q.xyz[3 * (size_t)i], y = q.xyz[3 * (size_t)i + 1], z = q.xyz[3 * (size_t)i + 2];
        if (q.half_xyz) { x = half_round(x); y = half_round(y); z = half_round(z); }
        means3D[3 * j] = x; means3D[3 * j + 1] = y; means3D[3 * j + 2] = z;
    }
    if (opac && q.opacity) {            // get_opacity: fq(sigmoid(_opacity))
        const Fq f = load_fq(q.state, C3DGS_FQ_OPACITY, q.fake_quant_enabled[C3DGS_FQ_OPACITY]);
        opac[j] = fq_val(f, sigmoid_f(q.opacity[i]));
    }
    if (sfac && q.scaling_factor) {     // get_scaling_factor: exp(fq(_scaling_factor))
        const Fq f = load_fq(q.state, C3DGS_FQ_SCALING_FACTOR, q.fake_quant_enabled[C3DGS_FQ_SCALING_FACTOR]);
        sfac[j] = expf(fq_val(f, q.scaling_factor[i]));
    }
    if (sh_out && sh_idx) sh_out[j] = sh_idx[i];
    if (g_out && g_idx) g_out[j] = g_idx[i];
}

void launch_qat_points(const c3dgs_qat_params& q, const uint8_t* visible, const int32_t* rank, const int64_t* sh_idx,
                       const int64_t* g_idx, float* means3D, float* opac, float* sfac, int64_t* sh_out, int64_t* g_out,
                       hipStream_t s)
{
    if (q.P <= 0) return;
    qat_points_kernel<<<(q.P + 255) / 256, 256, 0, s>>>(q, visible, rank, sh_idx, g_idx, means3D, opac, sfac, sh_out, g_out);
}

__global__ void __launch_bounds__(256)
qat_points_backward_kernel(const c3dgs_qat_params q, const uint8_t* __restrict__ visible, const int32_t* __restrict__ rank,
                           const float* __restrict__ g_m3, const float* __restrict__ g_m2, const float* __restrict__ g_op,
                           const float* __restrict__ g_sf, float* __restrict__ d_xyz, float* __restrict__ d_screen,
                           float* __restrict__ d_op, float* __restrict__ d_sf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= q.P) return;
    const bool vis = !visible || visible[i];
    const size_t j = vis ? (visible ? (size_t)rank[i] : (size_t)i) : 0;
    if (d_xyz) {                        // FakeQuantizationHalf.backward: identity
#pragma unroll
        for (int k = 0; k < 3; k++) d_xyz[3 * (size_t)i + k] = (vis && g_m3) ? g_m3[3 * j + k] : 0.f;
    }
    if (d_screen) {
#pragma unroll
        for (int k = 0; k < 3; k++) d_screen[3 * (size_t)i + k] = (vis && g_m2) ? g_m2[3 * j + k] : 0.f;
    }
    if (d_op) {
        float r = 0.f;
        if (vis && g_op) {
            const Fq f = load_fq(q.state, C3DGS_FQ_OPACITY, q.fake_quant_enabled[C3DGS_FQ_OPACITY]);
            const float sg = sigmoid_f(q.opacity[i]);
            bool m; (void)fq_apply(f, sg, m);
            r = m ? (g_op[j] * (1.f - sg)) * sg : 0.f;               // sigmoid_backward: grad * (1 - y) * y
        }
        d_op[i] = r;
    }
    if (d_sf) {
        float r = 0.f;
        if (vis && g_sf) {
            const Fq f = load_fq(q.state, C3DGS_FQ_SCALING_FACTOR, q.fake_quant_enabled[C3DGS_FQ_SCALING_FACTOR]);
            bool m;
            const float y = expf(fq_apply(f, q.scaling_factor[i], m));
            r = m ? g_sf[j] * y : 0.f;                                // exp backward: grad * result
        }
        d_sf[i] = r;
    }
}

void launch_qat_points_backward(const c3dgs_qat_params& q, const uint8_t* visible, const int32_t* rank, const float* g_m3,
                                const float* g_m2, const float* g_op, const float* g_sf, float* d_xyz, float* d_screen,
                                float* d_op, float* d_sf, hipStream_t s)
{
    if (q.P <= 0) return;
    qat_points_backward_kernel<<<(q.P + 255) / 256, 256, 0, s>>>(q, visible, rank, g_m3, g_m2, g_op, g_sf, d_xyz, d_screen, d_op, d_sf);
}

// ------------------------------------------------------------------------------------------- int8 payload (save_npz)
// torch.quantize_per_tensor(...).int_repr() of the activated tensors (scene/gaussian_model.py:525-617). The device
// kernel of torch evaluates nearbyint(double(x) / double(scale)) + zero_point (checked against torch on the MI355X box:
// bit-identical codes; the fp32 multiply-by-reciprocal of torch's CPU path differs at rounding ties), so does this one.
enum { QACT_IDENT = 0, QACT_SIGMOID = 1, QACT_NORMRELU3 = 2, QACT_NORM4 = 3, QACT_EXP = 4 };
struct QuantJob { const float* x; int8_t* out; long long units; int act; int slot; int first_block; int nblocks; };
struct QuantJobs { QuantJob j[C3DGS_FQ_COUNT]; int n; };

__device__ __forceinline__ int8_t quant_code(float v, double scale, int zp)
{
    const double q = nearbyint((double)v / scale) + (double)zp;
    return (int8_t)fmin(127.0, fmax(-128.0, q));
}

__global__ void __launch_bounds__(256)
qat_quantize_kernel(const QuantJobs jobs, const c3dgs_fq_state* __restrict__ state)
{
    int t = 0;
#pragma unroll
    for (int k = 1; k < C3DGS_FQ_COUNT; k++)
        if (k < jobs.n && (int)blockIdx.x >= jobs.j[k].first_block) t = k;
    const QuantJob job = jobs.j[t];
    const double scale = (double)state[job.slot].scale;
    const int zp = state[job.slot].zero_point;
    const long long stride = (long long)job.nblocks * 256, i0 = (long long)(blockIdx.x - job.first_block) * 256 + threadIdx.x;
    if (job.act == QACT_NORMRELU3) {
        for (long long r = i0; r < job.units; r += stride) {
            const float x[3] = { job.x[3 * r], job.x[3 * r + 1], job.x[3 * r + 2] };
            float u[3], v[3], n;
            normrelu3(x, u, v, n);
#pragma unroll
            for (int k = 0; k < 3; k++) job.out[3 * r + k] = quant_code(v[k], scale, zp);
        }
    } else if (job.act == QACT_NORM4) {      // rotation_activation(_rotation) = normalize(q), gaussian_model.py:606
        for (long long r = i0; r < job.units; r += stride) {
            const float4 w = reinterpret_cast<const float4*>(job.x)[r];
            const float d = fmaxf(sqrtf(w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w), NORM_EPS);
            char4 o;
            o.x = quant_code(w.x / d, scale, zp); o.y = quant_code(w.y / d, scale, zp);
            o.z = quant_code(w.z / d, scale, zp); o.w = quant_code(w.w / d, scale, zp);
            reinterpret_cast<char4*>(job.out)[r] = o;
        }
    } else {
        for (long long i = i0; i < job.units; i += stride) {
            float v = job.x[i];
            if (job.act == QACT_SIGMOID) v = sigmoid_f(v);
            else if (job.act == QACT_EXP) v = expf(v);
            job.out[i] = quant_code(v, scale, zp);
        }
    }
}

void launch_qat_quantize(const c3dgs_qat_params& q, int scaling_exp, int8_t* opacity, int8_t* scaling, int8_t* scaling_factor,
                         int8_t* rotation, int8_t* features_dc, int8_t* features_rest, hipStream_t s)
{
    QuantJobs J; J.n = 0;
    int nb = 0;
    auto add = [&](const float* x, int8_t* out, long long units, int act, int slot) {
        if (!x || !out || units <= 0) return;
        QuantJob& j = J.j[J.n++];
        j.x = x; j.out = out; j.units = units; j.act = act; j.slot = slot; j.first_block = nb;
        const long long want = (units + 1023) / 1024;
        j.nblocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
        nb += j.nblocks;
    };
    const long long P = q.P, GS = q.GS, SHS = q.SHS;
    add(q.opacity, opacity, P, QACT_SIGMOID, C3DGS_FQ_OPACITY);
    if (scaling_exp) add(q.scaling, scaling, GS * 3, QACT_EXP, C3DGS_FQ_SCALING);
    else add(q.scaling, scaling, GS, QACT_NORMRELU3, C3DGS_FQ_SCALING);
    add(q.scaling_factor, scaling_factor, P, QACT_IDENT, C3DGS_FQ_SCALING_FACTOR);
    add(q.rotation, rotation, GS, QACT_NORM4, C3DGS_FQ_ROTATION);
    add(q.features_dc, features_dc, SHS * 3, QACT_IDENT, C3DGS_FQ_FEATURES_DC);
    add(q.features_rest, features_rest, SHS * 3 * (q.M - 1), QACT_IDENT, C3DGS_FQ_FEATURES_REST);
    if (J.n == 0) return;
    qat_quantize_kernel<<<nb, 256, 0, s>>>(J, q.state);
}

// ------------------------------------------------------------------------------------------- stand-alone module
__global__ void __launch_bounds__(256)
fq_elementwise_kernel(long long n, const float* __restrict__ x, const c3dgs_fq_state* __restrict__ state, int enabled,
                      const float* __restrict__ g, float* __restrict__ out)
{
    const Fq f = load_fq(state, 0, enabled);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        bool m;
        const float y = fq_apply(f, x[i], m);
        out[i] = g ? (m ? g[i] : 0.f) : y;
    }
}

void launch_fake_quantize(long long n, const float* x, c3dgs_fq_state* state, int observe, int enabled, float c, float* out,
                          void* workspace, hipStream_t s)
{
    if (n <= 0) return;
    if (observe) {
        ObsJobs J; J.n = 0;
        int nb = 0;
        add_job(J, nb, x, n, ACT_IDENT, 0, 4);
        qat_observe_kernel<<<nb, 256, 0, s>>>(J, (float*)workspace);
        qat_finalize_kernel<<<1, 64 * C3DGS_FQ_COUNT, 0, s>>>(J, (const float*)workspace, state, c);
    }
    const long long want = (n + 1023) / 1024;
    fq_elementwise_kernel<<<(int)(want > 8192 ? 8192 : want), 256, 0, s>>>(n, x, state, enabled, nullptr, out);
}

void launch_fake_quantize_backward(long long n, const float* x, const c3dgs_fq_state* state, int enabled, const float* g,
                                   float* dx, hipStream_t s)
{
    if (n <= 0) return;
    const long long want = (n + 1023) / 1024;
    fq_elementwise_kernel<<<(int)(want > 8192 ? 8192 : want), 256, 0, s>>>(n, x, state, enabled, g, dx);
}

} // namespace c3dgs
