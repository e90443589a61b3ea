// Prototype of a hand-written onesweep LSD radix sort for the binning stage (gfx950), timed against rocPRIM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune/onesweep_proto.hip -o /tmp/onesweep && /tmp/onesweep
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>

#ifndef OS_BLOCK
#define OS_BLOCK 512
#endif
#ifndef OS_IPT
#define OS_IPT 8
#endif
constexpr int BLOCK = OS_BLOCK, IPT = OS_IPT, WAVES = BLOCK / 64, TILE = BLOCK * IPT, RADIX = 256;
constexpr uint32_t FLAG_AGG = 1u << 30, FLAG_PRE = 2u << 30, CNT_MASK = (1u << 30) - 1;

struct SortWs {
    uint32_t* hist;     // [passes][256] global digit counts
    uint32_t* status;   // [passes][blocks][256] look-back words
    uint32_t* ticket;   // [passes]
};

template <class K, int PASSES, int BITS0, int BITS_REST>
__global__ void __launch_bounds__(256) os_hist_kernel(const K* __restrict__ keys, size_t n, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[PASSES][RADIX];
    for (int q = threadIdx.x; q < PASSES * RADIX; q += 256) (&s_h[0][0])[q] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t k = keys[i];
        int shift = 0;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const int bits = p == 0 ? BITS0 : BITS_REST;
            atomicAdd(&s_h[p][(k >> shift) & ((1u << bits) - 1)], 1u);
            shift += bits;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < PASSES * RADIX; q += 256) {
        const uint32_t v = (&s_h[0][0])[q];
        if (v) atomicAdd(&hist[q], v);
    }
}

__device__ __forceinline__ uint32_t ld_status(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class K, int BITS>
__global__ void __launch_bounds__(BLOCK)
os_pass_kernel(const K* __restrict__ kin, K* __restrict__ kout, const uint32_t* __restrict__ vin, uint32_t* __restrict__ vout,
               uint32_t n, int shift, const uint32_t* __restrict__ hist /*[256] of this pass*/, uint32_t* __restrict__ status,
               uint32_t* __restrict__ ticket)
{
    constexpr uint32_t MASK = (1u << BITS) - 1;
    __shared__ uint32_t s_cnt[WAVES][RADIX];      // per-wave digit counters, later exclusive prefixes across waves
    __shared__ uint32_t s_start[RADIX];           // local start of every digit in the block-sorted order
    __shared__ int32_t s_gbase[RADIX];            // global position = s_gbase[d] + local position
    __shared__ uint32_t s_wtot[WAVES];
    __shared__ uint32_t s_bid;
    __shared__ K s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bid = atomicAdd(ticket, 1u);
    for (int q = tid; q < WAVES * RADIX; q += BLOCK) (&s_cnt[0][0])[q] = 0;
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint32_t block_start = bid * (uint32_t)TILE;
    const uint32_t valid = min((uint32_t)TILE, n - block_start);

    K key[IPT];
    uint32_t val[IPT], rank[IPT];
    const uint32_t wbase = block_start + (uint32_t)wave * 64u * IPT;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        const bool ok = idx < n;
        key[k] = ok ? kin[idx] : (K)~(K)0;
        val[k] = ok ? vin[idx] : 0u;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    volatile uint32_t* wc = s_cnt[wave];          // other lanes of the wave update these between iterations
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        const bool ok = idx < n;
        const uint32_t d = ((uint32_t)key[k] >> shift) & MASK;
        unsigned long long peers = __ballot(ok);                  // padding lanes of the last block take no part
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        rank[k] = 0;
        if (ok) {
            const uint32_t before = wc[d];
            rank[k] = before + (uint32_t)__popcll(peers & lt);
            if ((peers & lt) == 0) wc[d] = before + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // digit totals of the block, exclusive prefixes across waves, look-back across blocks
    uint32_t tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = tot; tot += c; }
    }
    // exclusive scan of `tot` over the 256 digit threads (waves 0..3)
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (tid < RADIX && lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    if (tid < RADIX) {
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        const uint32_t start = off + incl - tot;
        s_start[tid] = start;
        // global base of digit tid: exclusive scan of the global histogram (recomputed per block: 256 loads)
        // + counts of this digit in all earlier blocks (decoupled look-back)
        uint32_t* my = status + (size_t)bid * RADIX + tid;
        uint32_t pre = 0;
        if (bid == 0) st_status(my, FLAG_PRE | tot);
        else {
            st_status(my, FLAG_AGG | tot);
            for (int64_t b = (int64_t)bid - 1;; b--) {
                uint32_t s;
                do { s = ld_status(status + (size_t)b * RADIX + tid); } while ((s >> 30) == 0);
                pre += s & CNT_MASK;
                if (s & FLAG_PRE) break;
            }
            st_status(my, FLAG_PRE | (pre + tot));
        }
        s_gbase[tid] = (int32_t)pre - (int32_t)start;
    }
    // exclusive scan of the global histogram by the same 256 threads
    uint32_t h = tid < RADIX ? hist[tid] : 0u, hincl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(hincl, o); if (lane >= o) hincl += t; }
    __syncthreads();
    if (tid < RADIX && lane == 63) s_wtot[wave] = hincl;
    __syncthreads();
    if (tid < RADIX) {
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        s_gbase[tid] += (int32_t)(off + hincl - h);
    }
    // block-local stable reorder through LDS
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        if (idx < n) {
            const uint32_t d = ((uint32_t)key[k] >> shift) & MASK;
            const uint32_t p = s_start[d] + s_cnt[wave][d] + rank[k];
            s_keys[p] = key[k];
            s_vals[p] = val[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < IPT; m++) {
        const uint32_t p = (uint32_t)tid + (uint32_t)m * BLOCK;
        if (p < valid) {
            const K kk = s_keys[p];
            const uint32_t d = ((uint32_t)kk >> shift) & MASK;
            const uint32_t g = (uint32_t)(s_gbase[d] + (int32_t)p);
            kout[g] = kk;
            vout[g] = s_vals[p];
        }
    }
}

template <class K, int PASSES, int BITS0, int BITS_REST>
struct OneSweep {
    static size_t ws_bytes(size_t n)
    {
        const size_t blocks = (n + TILE - 1) / TILE;
        return (size_t)PASSES * RADIX * 4 + (size_t)PASSES * blocks * RADIX * 4 + 256;
    }
    // result lands in (kout, vout) when PASSES is odd, else in (kin, vin): the caller passes buffers accordingly
    static void run(void* ws, K* ka, K* kb, uint32_t* va, uint32_t* vb, size_t n, hipStream_t s)
    {
        const size_t blocks = (n + TILE - 1) / TILE;
        uint32_t* hist = (uint32_t*)ws;
        uint32_t* status = hist + PASSES * RADIX;
        uint32_t* ticket = status + (size_t)PASSES * blocks * RADIX;
        hipMemsetAsync(ws, 0, ws_bytes(n), s);
        const int hb = (int)std::min<size_t>((n + 256 * 32 - 1) / (256 * 32), 2048);
        os_hist_kernel<K, PASSES, BITS0, BITS_REST><<<hb, 256, 0, s>>>(ka, n, hist);
        int shift = 0;
        for (int p = 0; p < PASSES; p++) {
            K* ki = (p & 1) ? kb : ka; K* ko = (p & 1) ? ka : kb;
            uint32_t* vi = (p & 1) ? vb : va; uint32_t* vo = (p & 1) ? va : vb;
            if (p == 0)
                os_pass_kernel<K, BITS0><<<(unsigned)blocks, BLOCK, 0, s>>>(ki, ko, vi, vo, (uint32_t)n, shift, hist + p * RADIX,
                                                                           status + (size_t)p * blocks * RADIX, ticket + p);
            else
                os_pass_kernel<K, BITS_REST><<<(unsigned)blocks, BLOCK, 0, s>>>(ki, ko, vi, vo, (uint32_t)n, shift, hist + p * RADIX,
                                                                               status + (size_t)p * blocks * RADIX, ticket + p);
            shift += p == 0 ? BITS0 : BITS_REST;
        }
    }
};

template <class K, int PASSES, int BITS0, int BITS_REST>
static void bench(const char* name, size_t n, unsigned end_bit, unsigned maxkey)
{
    std::vector<K> hk(n);
    std::vector<uint32_t> hv(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) { hk[i] = (K)(maxkey ? rng() % maxkey : rng()); hv[i] = (uint32_t)i; }
    K *ka, *kb, *k0; uint32_t *va, *vb, *v0;
    hipMalloc(&ka, n * sizeof(K)); hipMalloc(&kb, n * sizeof(K)); hipMalloc(&k0, n * sizeof(K));
    hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4); hipMalloc(&v0, n * 4);
    hipMemcpy(k0, hk.data(), n * sizeof(K), hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    using OS = OneSweep<K, PASSES, BITS0, BITS_REST>;
    void* ws; hipMalloc(&ws, OS::ws_bytes(n));
    size_t rb = 0;
    rocprim::radix_sort_pairs(nullptr, rb, k0, kb, v0, vb, n, 0u, end_bit);
    void* rtmp; hipMalloc(&rtmp, rb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms_os = 0, ms_rp = 0, ms;
    const int R = 20;
    for (int it = 0; it < R + 3; it++) {
        hipMemcpyAsync(ka, k0, n * sizeof(K), hipMemcpyDeviceToDevice);
        hipMemcpyAsync(va, v0, n * 4, hipMemcpyDeviceToDevice);
        hipEventRecord(a);
        OS::run(ws, ka, kb, va, vb, n, 0);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (it >= 3) ms_os += ms;
    }
    K* kres = (PASSES & 1) ? kb : ka; uint32_t* vres = (PASSES & 1) ? vb : va;
    std::vector<K> ok(n); std::vector<uint32_t> ov(n);
    hipMemcpy(ok.data(), kres, n * sizeof(K), hipMemcpyDeviceToHost); hipMemcpy(ov.data(), vres, n * 4, hipMemcpyDeviceToHost);
    // reference: stable sort of (key & mask) with ids
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    const uint64_t kmask = end_bit >= 32 ? 0xffffffffull : ((1ull << end_bit) - 1);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return ((uint64_t)hk[x] & kmask) < ((uint64_t)hk[y] & kmask); });
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) if (ov[i] != order[i] || ok[i] != hk[order[i]]) { if (!bad) printf("  first mismatch at %zu\n", i); bad++; }
    for (int it = 0; it < R + 3; it++) {
        hipEventRecord(a);
        rocprim::radix_sort_pairs(rtmp, rb, k0, kb, v0, vb, n, 0u, end_bit);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (it >= 3) ms_rp += ms;
    }
    printf("%s n=%zu: hand-written onesweep %.4f ms (%s), rocPRIM %.4f ms  [block %d x %d items]\n", name, n, ms_os / R,
           bad ? "WRONG" : "stable order verified", ms_rp / R, BLOCK, IPT);
}

int main()
{
    bench<uint16_t, 2, 8, 5>("tile sort (u16 keys, 13 bits: 8+5)", 16403154, 13, 8160);
    bench<uint16_t, 2, 7, 6>("tile sort (u16 keys, 13 bits: 7+6)", 16403154, 13, 8160);
    bench<uint32_t, 4, 8, 8>("depth sort (u32 keys, 32 bits)", 3000000, 32, 0);
    bench<uint16_t, 2, 8, 5>("small", 5000, 13, 8160);
    return 0;
}
