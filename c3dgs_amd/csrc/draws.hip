// draws.hip -- the batch indices of one Lloyd step, `torch.randint(0, N, [B])` on the CPU generator
// (reference compression/vq.py:69), produced without the ~2 ns/draw of torch's scalar path.
//
// torch's CPU randint for a range below 2^28 is   idx[i] = mt19937() % N   with one 32-bit MT19937 output per
// element, drawn serially. The stream is continued here by a block-wise MT19937 (the 624-word reload written as
// three dependence-free loops the compiler vectorises, tempering as a separate pass) straight into pinned host
// memory as raw 32-bit words; the `% N` and the widening to int64 run on the GPU after one 4 B/draw copy (instead of
// 8 B/draw). The caller (c3dgs_amd/vq.py) reads the generator state from torch.get_rng_state() and writes the
// advanced state back, so everything drawn from the torch generator afterwards is unchanged too.
#include "common.hpp"

namespace c3dgs {

constexpr int MT_N = 624, MT_M = 397;

static inline uint32_t mt_twist(uint32_t u, uint32_t v)
{
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

// one reload of the 624-word state (Matsumoto & Nishimura's recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1])).
// Host code: built twice (AVX2 and baseline x86-64) with run-time dispatch, the loops vectorise 8 / 4 words wide.
#if defined(__HIP_DEVICE_COMPILE__)
#define C3DGS_HOST_SIMD                     // the device pass only parses these host functions
#else
#define C3DGS_HOST_SIMD __attribute__((target_clones("avx2", "default")))
#endif
C3DGS_HOST_SIMD static void mt_reload(uint32_t* x)
{
    for (int k = 0; k < MT_N - MT_M; k++) x[k] = x[k + MT_M] ^ mt_twist(x[k], x[k + 1]);             // reads old words only
    for (int k = MT_N - MT_M; k < MT_N - 1; k++) x[k] = x[k - (MT_N - MT_M)] ^ mt_twist(x[k], x[k + 1]); // new words 227 behind
    x[MT_N - 1] = x[MT_M - 1] ^ mt_twist(x[MT_N - 1], x[0]);
}

static inline uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

C3DGS_HOST_SIMD static void mt_temper_block(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int64_t n)
{
    for (int64_t k = 0; k < n; k++) out[k] = mt_temper(src[k]);
}

__global__ void __launch_bounds__(256)
draws_to_indices_kernel(int64_t n, uint32_t range, const uint32_t* __restrict__ raw, int64_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int64_t)(raw[i] % range);
}

} // namespace c3dgs

using namespace c3dgs;

extern "C" {

// `left` / `next` follow at::mt19937 (left = words remaining in the block + 1; a fresh or exhausted block has left == 1):
// per draw  `if (--left == 0) reload, left = 624, next = 0;  y = temper(state[next++])`.
int c3dgs_mt19937_fill(uint32_t* state, int64_t* left, int64_t* next, uint32_t* out, int64_t n)
{
    if (!state || !left || !next || (n > 0 && !out) || n < 0) return fail(C3DGS_E_INVALID, "mt19937_fill: bad arguments");
    int64_t l = *left, nx = *next;
    if (l < 1 || l > MT_N + 1 || nx < 0 || nx > MT_N || (l > 1 && nx + (l - 1) != MT_N))
        return fail(C3DGS_E_INVALID, "mt19937_fill: inconsistent generator state");
    while (n > 0) {
        if (l == 1) { mt_reload(state); l = MT_N + 1; nx = 0; }
        const int64_t take = (l - 1) < n ? (l - 1) : n;
        mt_temper_block(state + nx, out, take);
        out += take; n -= take; nx += take; l -= take;
    }
    *left = l; *next = nx;
    return C3DGS_OK;
}

int c3dgs_draws_to_indices(int64_t n, int64_t range, const uint32_t* raw, int64_t* out, void* stream)
{
    if (n < 0 || range <= 0 || range >= ((int64_t)1 << 32)) return fail(C3DGS_E_INVALID, "draws_to_indices: range must be in [1, 2^32)");
    if (n == 0) return C3DGS_OK;
    if (!raw || !out) return fail(C3DGS_E_INVALID, "draws_to_indices: bad arguments");
    draws_to_indices_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, (uint32_t)range, raw, out);
    C3DGS_STAGE("draws_to_indices", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

/* the device-side address of a page-locked host buffer (hipHostMalloc / torch pin_memory), or NULL if the buffer is not mapped
 * into the device's address space. With it, c3dgs_draws_to_indices reads the raw words straight from host memory: no copy
 * operation in the stream at all (a rank's slice of a batch is 128 KB; the copy's launch cost more than its transfer). */
void* c3dgs_host_buffer_device_address(const void* host_ptr)
{
    hipPointerAttribute_t at;
    if (!host_ptr || hipPointerGetAttributes(&at, host_ptr) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return at.devicePointer;
}

/* upload + convert in one call: `n` raw words from (pinned) host memory to `raw_dev`, then out[i] = raw_dev[i] % range */
int c3dgs_draws_upload(int64_t n, int64_t range, const uint32_t* raw_host, uint32_t* raw_dev, int64_t* out, void* stream)
{
    if (n < 0 || range <= 0 || range >= ((int64_t)1 << 32)) return fail(C3DGS_E_INVALID, "draws_upload: range must be in [1, 2^32)");
    if (n == 0) return C3DGS_OK;
    if (!raw_host || !raw_dev || !out) return fail(C3DGS_E_INVALID, "draws_upload: bad arguments");
    C3DGS_HIP_TRY(hipMemcpyAsync(raw_dev, raw_host, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
    draws_to_indices_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, (uint32_t)range, raw_dev, out);
    C3DGS_STAGE("draws_to_indices", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

} // extern "C"
