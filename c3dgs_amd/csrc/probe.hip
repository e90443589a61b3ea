// probe.hip -- measurement-only kernels (c3dgs_debug_gather_probe): known access patterns with a known byte count, run under
// the same two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) as the product kernels, to calibrate what those counters report
// for GATHERS on gfx950 (MI355X_MICROARCH.md only calibrates wide coalesced streams: FETCH_SIZE = 1/2 of the bytes).
// tools/pmc_calibrate.py drives them; nothing in the product path calls them.
#include "common.hpp"

namespace c3dgs {

// kind 0: coalesced stream, 16 bytes per lane (the calibrated case: the control)
__global__ void __launch_bounds__(256) probe_stream_kernel(size_t n16, const uint4* __restrict__ src, uint32_t* __restrict__ out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) out[0] = acc;           // keeps the loads alive, practically never stores
}

// kind 1: one RECORD of REC_VEC x 16 bytes per lane at a random record index (48-byte splat records: REC_VEC = 3, as the blend
// kernels gather them; 192-byte SH rows: REC_VEC = 12, as preprocess reads them: twelve 16-byte loads walking one row)
template <int REC_VEC>
__global__ void __launch_bounds__(256) probe_gather_kernel(size_t n, const uint4* __restrict__ table, const uint32_t* __restrict__ index,
                                                            uint32_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4* rec = table + (size_t)index[i] * REC_VEC;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < REC_VEC; k++) {
        const uint4 v = rec[k];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) out[0] = acc;
}

// kind 3: scattered 36-byte slot stores (nine floats at a 36-byte pitch, random slot per lane): render_backward's partial sums
__global__ void __launch_bounds__(256) probe_scatter36_kernel(size_t n, float* __restrict__ table, const uint32_t* __restrict__ index)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* dst = table + (size_t)index[i] * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) dst[k] = (float)k;
}

int launch_gather_probe(int kind, size_t n, void* table, const uint32_t* index, uint32_t* out, hipStream_t s)
{
    if (n == 0) return 0;
    const unsigned g = (unsigned)((n + 255) / 256);
    switch (kind) {
    case 0: probe_stream_kernel<<<2048, 256, 0, s>>>(n, (const uint4*)table, out); return 0;
    case 1: probe_gather_kernel<3><<<g, 256, 0, s>>>(n, (const uint4*)table, index, out); return 0;
    case 2: probe_gather_kernel<12><<<g, 256, 0, s>>>(n, (const uint4*)table, index, out); return 0;
    case 3: probe_scatter36_kernel<<<g, 256, 0, s>>>(n, (float*)table, index); return 0;
    default: return 1;
    }
}

} // namespace c3dgs
