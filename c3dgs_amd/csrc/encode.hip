// encode.hip -- Morton ordering of the Gaussians for the compressed on-disk format.
//
//   reference: GaussianModel._sort_morton, scene/gaussian_model.py:997-1003, mortonEncode/splitBy3 :1417-1432
//     xyz_q = ((2**21 - 1) * (xyz - min) / (max - min)).long();  order = mortonEncode(xyz_q, diap.argsort()).sort().indices
//
// Three small HBM-bound passes: bounding box (wave + workgroup reduction, 6 ordered-int atomics per workgroup),
// quantise + 21-bit interleave (integer work, bit-exact with the reference's fp32 expression order), and a rocPRIM
// radix sort of (63-bit code, id) pairs. The sort is stable, so equal codes keep ascending id (torch.sort gives no
// such guarantee; every stable order is a valid reference output).
#include "common.hpp"
#include <rocprim/device/device_radix_sort.hpp>

namespace c3dgs {

// order-preserving float <-> uint mapping so that atomicMin/atomicMax on uint order floats
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__global__ void __launch_bounds__(256)
bbox_kernel(int P, const float* __restrict__ xyz, uint32_t* __restrict__ box /*[6]: min xyz, max xyz (ordered ints)*/)
{
    float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = xyz[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            atomicMin(&box[a], f2ord(mn[a]));
            atomicMax(&box[3 + a], f2ord(mx[a]));
        }
}

__device__ __forceinline__ uint64_t split_by_3(uint64_t a)      // gaussian_model.py:1417-1424
{
    uint64_t x = a & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256)
morton_codes_kernel(int P, const float* __restrict__ xyz, const uint32_t* __restrict__ box, uint64_t* __restrict__ codes,
                    uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float mn[3], diap[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(box[a]); diap[a] = ord2f(box[3 + a]) - mn[a]; }
    int ord[3] = { 0, 1, 2 };                                    // pp_diap.argsort(), ties keep the lower axis first
    if (diap[ord[1]] < diap[ord[0]]) { int t = ord[0]; ord[0] = ord[1]; ord[1] = t; }
    if (diap[ord[2]] < diap[ord[0]]) { int t = ord[0]; ord[0] = ord[2]; ord[2] = t; }
    if (diap[ord[2]] < diap[ord[1]]) { int t = ord[1]; ord[1] = ord[2]; ord[2] = t; }
    uint64_t q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = __fdiv_rn(__fmul_rn(2097151.0f, __fsub_rn(xyz[3 * (size_t)i + a], mn[a])), diap[a]);
        q[a] = (uint64_t)(long long)v;                           // .long(): truncation
    }
    codes[i] = split_by_3(q[ord[0]]) | split_by_3(q[ord[1]]) << 1 | split_by_3(q[ord[2]]) << 2;
    ids[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
widen_kernel(int P, const uint32_t* __restrict__ in, int64_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) out[i] = (int64_t)in[i];
}

size_t morton_workspace_bytes(int P)
{
    const size_t p = (size_t)(P > 0 ? P : 1);
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, p, 0u, 63u);
    return align_up(32) + align_up(p * 8) + align_up(p * 4) + align_up(p * 4) + align_up(temp < 256 ? 256 : temp);
}

int run_morton_order(int P, const float* xyz, int64_t* codes_out, int64_t* order_out, void* workspace, hipStream_t s)
{
    const size_t p = (size_t)P;
    char* w = (char*)workspace;
    uint32_t* box = (uint32_t*)w;                 w += align_up(32);
    uint64_t* codes_sorted = (uint64_t*)w;        w += align_up(p * 8);
    uint32_t* ids = (uint32_t*)w;                 w += align_up(p * 4);
    uint32_t* ids_sorted = (uint32_t*)w;          w += align_up(p * 4);
    void* temp = w;
    size_t temp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, p, 0u, 63u);
    const uint32_t init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    if (hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    const int grid = (P + 255) / 256;
    bbox_kernel<<<grid < 1024 ? grid : 1024, 256, 0, s>>>(P, xyz, box);
    morton_codes_kernel<<<grid, 256, 0, s>>>(P, xyz, box, (uint64_t*)codes_out, ids);
    if (rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t*)codes_out, codes_sorted, ids, ids_sorted, p, 0u, 63u, s) != hipSuccess)
        return 1;
    widen_kernel<<<grid, 256, 0, s>>>(P, ids_sorted, order_out);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// extract_rot_scale(to_full_cov(cov6))  (utils/splats.py:7-35, used by compress_covariance, compression/vq.py:186):
// eigendecomposition of the symmetric 3x3 covariance of every codebook entry -> (unit quaternion, sqrt eigenvalues).
// The reference calls torch.linalg.eigh (a batched LAPACK / solver call, ascending eigenvalues) and then
// matrix_to_quaternion(R * det(R)); here ONE thread per matrix runs a cyclic Jacobi iteration in fp64 (converges to
// fp64 round-off in <= 6 sweeps for 3x3, so the fp32 results carry no iteration error), sorts ascending, fixes the
// handedness and converts with the same best-conditioned-candidate rule. Eigenvector SIGNS are not determined by the
// problem (LAPACK's choice is arbitrary as well): parity is on the eigenvalues and on R diag(s^2) R^T.
__device__ __forceinline__ void jacobi_rotate(double a[3][3], double v[3][3], int p, int q)
{
    if (a[p][q] == 0.0) return;
    const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
    const int r = 3 - p - q;
    const double apr = a[p][r], aqr = a[q][r];
    a[p][p] -= t * a[p][q];
    a[q][q] += t * a[p][q];
    a[p][q] = a[q][p] = 0.0;
    a[p][r] = a[r][p] = c * apr - sn * aqr;
    a[q][r] = a[r][q] = sn * apr + c * aqr;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double vp = v[k][p], vq = v[k][q];
        v[k][p] = c * vp - sn * vq;
        v[k][q] = sn * vp + c * vq;
    }
}

__global__ void __launch_bounds__(256)
extract_rot_scale_kernel(int n, const float* __restrict__ cov6, float* __restrict__ rot, float* __restrict__ scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* cv = cov6 + 6 * (size_t)i;
    const float e = 1e-8f;                                       // cov + eye(3) * 1e-8, in fp32 as torch evaluates it
    double a[3][3] = { { (double)(cv[0] + e), (double)cv[1], (double)cv[2] },
                       { (double)cv[1], (double)(cv[3] + e), (double)cv[4] },
                       { (double)cv[2], (double)cv[4], (double)(cv[5] + e) } };
    double v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 10; sweep++) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        jacobi_rotate(a, v, 0, 1);
        jacobi_rotate(a, v, 0, 2);
        jacobi_rotate(a, v, 1, 2);
    }
    // ascending eigenvalues (torch.linalg.eigh order); columns of v follow
    int o0 = 0, o1 = 1, o2 = 2;
    double w0 = a[0][0], w1 = a[1][1], w2 = a[2][2];
    if (w0 > w1) { double t = w0; w0 = w1; w1 = t; int k = o0; o0 = o1; o1 = k; }
    if (w1 > w2) { double t = w1; w1 = w2; w2 = t; int k = o1; o1 = o2; o2 = k; }
    if (w0 > w1) { double t = w0; w0 = w1; w1 = t; int k = o0; o0 = o1; o1 = k; }
    const float S[3] = { (float)w0, (float)w1, (float)w2 };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float sq = sqrtf(S[k]);
        scale[3 * (size_t)i + k] = (sq != sq) ? 1e-6f : sq;      // S.sqrt().nan_to_num(nan=1e-6)
    }
    float m[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) { m[r][0] = (float)v[r][o0]; m[r][1] = (float)v[r][o1]; m[r][2] = (float)v[r][o2]; }
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                      m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) m[r][c] *= det;              // R * R.det(): determinant +1
    // matrix_to_quaternion (utils/splats.py:43-105): best-conditioned of the four candidates
    const float qa[4] = { 1.f + m[0][0] + m[1][1] + m[2][2], 1.f + m[0][0] - m[1][1] - m[2][2],
                          1.f - m[0][0] + m[1][1] - m[2][2], 1.f - m[0][0] - m[1][1] + m[2][2] };
    float q_abs[4];
    int best = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { q_abs[k] = qa[k] > 0.f ? sqrtf(qa[k]) : 0.f; if (q_abs[k] > q_abs[best]) best = k; }
    float c4[4];
    if (best == 0) { c4[0] = q_abs[0] * q_abs[0]; c4[1] = m[2][1] - m[1][2]; c4[2] = m[0][2] - m[2][0]; c4[3] = m[1][0] - m[0][1]; }
    else if (best == 1) { c4[0] = m[2][1] - m[1][2]; c4[1] = q_abs[1] * q_abs[1]; c4[2] = m[1][0] + m[0][1]; c4[3] = m[0][2] + m[2][0]; }
    else if (best == 2) { c4[0] = m[0][2] - m[2][0]; c4[1] = m[1][0] + m[0][1]; c4[2] = q_abs[2] * q_abs[2]; c4[3] = m[1][2] + m[2][1]; }
    else { c4[0] = m[1][0] - m[0][1]; c4[1] = m[2][0] + m[0][2]; c4[2] = m[2][1] + m[1][2]; c4[3] = q_abs[3] * q_abs[3]; }
    const float den = 2.0f * fmaxf(q_abs[best], 0.1f);
    float qv[4] = { c4[0] / den, c4[1] / den, c4[2] / den, c4[3] / den };
    const float nrm = fmaxf(sqrtf(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3]), 1e-12f);   // F.normalize
    reinterpret_cast<float4*>(rot)[i] = make_float4(qv[0] / nrm, qv[1] / nrm, qv[2] / nrm, qv[3] / nrm);
}

void launch_extract_rot_scale(int n, const float* cov6, float* rot, float* scale, hipStream_t s)
{
    if (n <= 0) return;
    extract_rot_scale_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, cov6, rot, scale);
}

} // namespace c3dgs
