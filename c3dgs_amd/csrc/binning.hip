// binning.hip -- prefix sum and the two radix sorts of the binning stage on rocPRIM (gfx950).
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
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace c3dgs {

struct TilesOf {
    const uint32_t* tiles;
    __host__ __device__ uint32_t operator()(uint32_t id) const { return tiles[id]; }
};
using OrderedTilesIt = rocprim::transform_iterator<const uint32_t*, TilesOf, uint32_t>;

size_t scan_temp_bytes(int P)
{
    size_t a = 0, b = 0;
    OrderedTilesIt it((const uint32_t*)nullptr, TilesOf{ nullptr });
    (void)rocprim::inclusive_scan(nullptr, a, it, (uint32_t*)nullptr, (size_t)P, rocprim::plus<uint32_t>());
    (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)P, 0u, 32u);
    size_t m = a > b ? a : b;
    const size_t os = onesweep_depth_temp_bytes(P);               // the hand-written sort shares the same scratch region
    if (os > m) m = os;
    return m < 256 ? 256 : m;
}

size_t sort_temp_bytes(int R, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint16_t*)nullptr, (uint16_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)R, 0u, (unsigned)end_bit);
    const size_t os = onesweep_tile_temp_bytes(R, end_bit);
    if (os > bytes) bytes = os;
    return bytes < 256 ? 256 : bytes;
}

hipError_t run_depth_sort(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                          uint32_t* vout, int P, hipStream_t s)
{
    if (onesweep_enabled() && (size_t)P < ((size_t)1 << 30)) return onesweep_depth_sort(temp, temp_bytes, kin, kout, vin, vout, P, s);
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)P, 0u, 32u, s);
}

hipError_t run_scan_in_order(void* temp, size_t temp_bytes, const uint32_t* order, const uint32_t* tiles_touched,
                             uint32_t* out, int P, hipStream_t s)
{
    OrderedTilesIt it(order, TilesOf{ tiles_touched });
    return rocprim::inclusive_scan(temp, temp_bytes, it, out, (size_t)P, rocprim::plus<uint32_t>(), s);
}

hipError_t run_tile_sort(void* temp, size_t temp_bytes, const uint16_t* kin, uint16_t* kout, const uint32_t* vin,
                         uint32_t* vout, int R, int end_bit, hipStream_t s)
{
    if (onesweep_enabled() && (size_t)R < ((size_t)1 << 30)) return onesweep_tile_sort(temp, temp_bytes, kin, kout, vin, vout, R, end_bit, s);
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)R, 0u, (unsigned)end_bit, s);
}

} // namespace c3dgs
