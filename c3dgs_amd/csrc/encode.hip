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

} // namespace c3dgs
