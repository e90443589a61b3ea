// binning.hip -- K3 prefix sum and K6 key sort on rocPRIM (gfx950).
//
// Reference: cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:162,275) and
// cub::DeviceRadixSort::SortPairs<uint64,uint32>(…, begin_bit=0, end_bit=32+bit) (:184-187, 301-306).
// Both are exact integer operations; the radix sort is stable, and duplicate_with_keys emits the
// instances in ascending Gaussian index, so the sorted point list is uniquely determined.
// (The umbrella <rocprim/rocprim.hpp> does not compile on this ROCm install; include the two
// device headers directly.)
#include "common.hpp"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace c3dgs {

size_t scan_temp_bytes(int P)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)P,
                                  rocprim::plus<uint32_t>());
    return bytes < 256 ? 256 : bytes;
}

size_t sort_temp_bytes(int R, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)R, 0u, (unsigned)end_bit);
    return bytes < 256 ? 256 : bytes;
}

hipError_t run_inclusive_scan(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, int P, hipStream_t s)
{
    return rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)P, rocprim::plus<uint32_t>(), s);
}

hipError_t run_sort_pairs(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                          uint32_t* vout, int R, int end_bit, hipStream_t s)
{
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)R, 0u, (unsigned)end_bit, s);
}

} // namespace c3dgs
