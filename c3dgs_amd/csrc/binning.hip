// binning.hip -- the two sorts of the binning stage: dispatch to the hand-written onesweep (radix_sort.hip, default) or to
// rocPRIM (C3DGS_SORT_ROCPRIM=1, and beyond 2^30 items), and the scratch sizes both need (gfx950).
// The two prefix sums (cub::DeviceScan::InclusiveSum in the reference) are two-level scans folded into preprocess.hip.
//
// Reference: cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:162,275) and ONE
// cub::DeviceRadixSort::SortPairs<uint64,uint32> over all R tile instances on bits [0, 32+bit) (:184-187, 301-306):
// at 1080p that is 5-6 radix passes over 12 B x R and half of the forward's bytes.
//
// Here the same ordering is produced in two cheaper stages:
//   1. sort the P Gaussians by depth bits (32-bit keys, payload = id; stable, ids ascending on input);
//   2. emit the tile instances in that order and stable-sort them by the TILE id only (16-bit keys, `bit` <= 16
//      significant bits -> 2 passes over 6 B x R).
// A stable sort by tile of a (depth, id)-ordered list is exactly the (tile, depth)-sorted list with ties in
// emission (= id) order that the reference's single stable sort yields, so the sorted point list is bit-identical
// (asserted against the oracle). (The umbrella <rocprim/rocprim.hpp> does not compile on this ROCm install.)
#include "common.hpp"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace c3dgs {

size_t scan_temp_bytes(int P)
{
    size_t a = 0, b = 0;                                          // a: reserved for other users of the region
    (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)P, 0u, 32u);
    size_t m = a > b ? a : b;
    const size_t os = onesweep_depth_temp_bytes(P);               // the hand-written sort shares the same scratch region
    if (os > m) m = os;
    return m < 256 ? 256 : m;
}

size_t sort_temp_bytes(int R, int end_bit, int key_bytes)
{
    size_t bytes = 0;
    if (key_bytes == 4)
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                        (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)R, 0u, (unsigned)end_bit);
    else
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint16_t*)nullptr, (uint16_t*)nullptr,
                                        (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)R, 0u, (unsigned)end_bit);
    const size_t os = onesweep_tile_temp_bytes(R, end_bit, key_bytes);
    if (os > bytes) bytes = os;
    // after the sort the region is reused for the R quadrant-mask bytes
    const size_t reuse = align_up((size_t)(R > 0 ? R : 1));
    if (reuse > bytes) bytes = reuse;
    return bytes < 256 ? 256 : bytes;
}

__global__ void __launch_bounds__(256)
gather_u64_kernel(int n, const uint32_t* __restrict__ idx, const uint2* __restrict__ src, uint2* __restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// sorts (depth bits, id) and also delivers `rects_sorted[k]` = tile rectangle of the k-th nearest Gaussian (the one
// per-Gaussian record the pair emission needs): the hand-written sort gathers it while scattering its last digit pass,
// the rocPRIM path with one extra kernel
size_t depth_sort_clear_bytes(int P)
{
    return (onesweep_enabled() && (size_t)P < ((size_t)1 << 30)) ? onesweep_depth_clear_bytes(P) : 0;
}
size_t tile_sort_clear_bytes(int R, int end_bit, int key_bytes)
{
    return (onesweep_enabled() && (size_t)R < ((size_t)1 << 30)) ? onesweep_tile_clear_bytes(R, end_bit, key_bytes) : 0;
}

hipError_t run_depth_sort(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                          uint32_t* vout, int P, const uint2* rects, uint2* rects_sorted, hipStream_t s, bool ctrl_cleared,
                          bool rects_fit_bytes)
{
    if (onesweep_enabled() && (size_t)P < ((size_t)1 << 30))
        return onesweep_depth_sort(temp, temp_bytes, kin, kout, vin, vout, P, rects, rects_sorted, s, ctrl_cleared, rects_fit_bytes);
    // vin == nullptr: the payload is the Gaussian id itself (0 .. P-1)
    hipError_t e = vin ? rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)P, 0u, 32u, s)
                       : rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, rocprim::counting_iterator<uint32_t>(0u), vout,
                                                   (size_t)P, 0u, 32u, s);
    if (e != hipSuccess) return e;
    gather_u64_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, vout, rects, rects_sorted);
    return hipGetLastError();
}

hipError_t run_tile_sort(void* temp, size_t temp_bytes, const void* kin, void* kout, int key_bytes, const uint32_t* vin,
                         uint32_t* vout, int R, int end_bit, hipStream_t s, bool ctrl_cleared)
{
    if (key_bytes == 4) {       // more than 65,536 tiles: 32-bit tile keys, three digit passes for 17-24 tile bits
        if (onesweep_enabled() && (size_t)R < ((size_t)1 << 30))
            return onesweep_tile_sort32(temp, temp_bytes, (const uint32_t*)kin, (uint32_t*)kout, vin, vout, R, end_bit, s, ctrl_cleared);
        return rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)kin, (uint32_t*)kout, vin, vout, (size_t)R, 0u, (unsigned)end_bit, s);
    }
    if (onesweep_enabled() && (size_t)R < ((size_t)1 << 30))
        return onesweep_tile_sort(temp, temp_bytes, (const uint16_t*)kin, (uint16_t*)kout, vin, vout, R, end_bit, s, ctrl_cleared);
    return rocprim::radix_sort_pairs(temp, temp_bytes, (const uint16_t*)kin, (uint16_t*)kout, vin, vout, (size_t)R, 0u, (unsigned)end_bit, s);
}

} // namespace c3dgs
