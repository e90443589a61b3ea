// radix_sort.hip -- stable LSD radix sort of (key, u32 value) pairs for the two sorts of the binning stage (gfx950).
//
// Replaces rocPRIM's radix_sort_pairs in run_depth_sort / run_tile_sort (binning.hip keeps the rocPRIM path, selectable
// with C3DGS_SORT_ROCPRIM=1). Same algorithm family (onesweep: one read and one write of the data per digit, chained
// look-back instead of a global scan), written for this workload:
//   * ONE memset per sort (digit histograms + look-back words + tickets are one contiguous block) instead of the two
//     5-microsecond fill launches rocPRIM issues in front of every digit pass (14 per forward);
//   * 8192-item tiles of 1024 threads for the u16 tile keys, 12288-item tiles of 512 threads for the u32 depth keys: measured 0.228 ms against 0.267 ms for the 16.4 M (u16, u32) tile-key pairs;
//   * digit widths chosen per sort (13 tile bits = 7 + 6), keys of the native width.
// Structure of one digit pass (os_pass_kernel), per 8192-item tile:
//   ticket      blocks take their tile index from an atomic counter, so every predecessor of a block is already running
//               (forward progress of the look-back does not depend on the dispatch order);
//   rank        wave w owns a contiguous chunk of the tile and walks it 64 items at a time; lanes with the same digit find
//               each other with BITS ballots, rank = wave counter + popcount(peers below me), the first peer bumps the
//               counter -> ranks are in input order (stable);
//   look-back   thread d publishes the tile's count of digit d (aggregate), adds up the predecessors' words until it meets
//               an inclusive prefix, publishes its own inclusive prefix (one 32-bit word: 2 flag bits + 30 count bits,
//               relaxed agent-scope atomics);
//   scatter     items are reordered through LDS into tile-sorted order, then written with consecutive threads on
//               consecutive positions of each digit run.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>

namespace c3dgs {

constexpr int OS_RADIX = 256;
// tile shape per key width (measured): u16 keys 8192-item tiles, 1024 threads x 8 items; u32 keys C3DGS_OS_TILE32 items with
// C3DGS_OS_BLOCK32 threads (82 KB of LDS would allow only one 1024-thread workgroup per CU); only the u32 shape is a build knob
#ifndef C3DGS_OS_TILE32
#define C3DGS_OS_TILE32 12288   // round 2 (tools/ablate_sort2.sh, 3M keys): 8192 x 512 0.205 ms, 12288 x 512 0.191, 16384 x 512 0.209, 16384 x 1024 0.194
#endif
#ifndef C3DGS_OS_BLOCK32
#define C3DGS_OS_BLOCK32 512
#endif
// u16 keys: fixed shape. os_tile_hist_kernel (the look-back-free first pass) counts the SAME tiles with one 16-byte load of
// 8 keys per thread, so tile = 8 x 1024 is not a build knob (measured alternatives before that pre-pass existed, 16.4 M
// (u16, u32) pairs: 8192 x 1024 0.215 ms, 12288 x 1024 0.243, 16384 x 1024 0.225)
constexpr int OS_TILE16 = 8192, OS_BLOCK16 = 1024;
template <class K> struct OsShape {
    static constexpr int TILE = sizeof(K) == 2 ? OS_TILE16 : C3DGS_OS_TILE32;
    static constexpr int BLOCK = sizeof(K) == 2 ? OS_BLOCK16 : C3DGS_OS_BLOCK32;
    static constexpr int IPT = TILE / BLOCK;
};
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_PRE = 2u << 30, OS_CNT_MASK = (1u << 30) - 1;
constexpr int OS_MAX_PASSES = 4;
// Every look-back spin is bounded (a predecessor's word normally arrives within microseconds; tickets guarantee it is
// running). On time-out the pass finishes with wrong offsets and raises the device's STICKY error word g_os_error, which
//   * render_forward reads: a forward whose sorts timed out returns a NaN image instead of a plausible wrong one;
//   * every forward copies to the host together with num_rendered (the one natural device->host read): the call fails
//     with C3DGS_E_HIP and clears the word (c_abi.hip);
//   * debug mode reads back synchronously after each sort.
// C3DGS_OS_SPIN_LIMIT is a build parameter only so that a test build can force the time-out (tests/test_sort_gpu.py).
#ifndef C3DGS_OS_SPIN_LIMIT
#define C3DGS_OS_SPIN_LIMIT (1u << 22)
#endif
constexpr uint32_t OS_SPIN_LIMIT = C3DGS_OS_SPIN_LIMIT;
__device__ uint32_t g_os_error;           // zero-initialised at module load; bit 0 = tile-key sort, bit 1 = depth-key sort
#ifdef C3DGS_OS_TIMING
__device__ unsigned long long g_os_times[64 * 8];
#define OS_T(slot) if (tid == 0 && (bid & 3) == 0 && (bid >> 2) < 64) g_os_times[(bid >> 2) * 8 + (slot)] = __builtin_readcyclecounter();
#else
#define OS_T(slot)
#endif

struct OsPlan { int passes; int bits[OS_MAX_PASSES]; };

static OsPlan os_plan(int total_bits)
{
    OsPlan p;
    p.passes = (total_bits + 7) / 8;
    if (p.passes < 1) p.passes = 1;
    int left = total_bits < 1 ? 1 : total_bits;
    for (int i = 0; i < p.passes; i++) {                    // spread the bits evenly: 13 -> 7 + 6
        const int b = (left + (p.passes - i) - 1) / (p.passes - i);
        p.bits[i] = b;
        left -= b;
    }
    return p;
}

template <class K> static size_t os_blocks(size_t n) { return (n + OsShape<K>::TILE - 1) / OsShape<K>::TILE; }
// The tile-key sort (u16 keys, two passes) runs its FIRST pass without the chained look-back: the histogram kernel works tile by
// tile and leaves every tile's digit counts in a table, one small kernel turns the table's columns into exclusive prefixes
// over the tiles (and their totals into the global histograms), and the pass reads its offsets from there. Measured on the
// 16.4 M pairs of the bench view: the look-back costs ~30 us per pass at 2002 tiles (tools/time_sort.py knock-out).
template <class K> constexpr bool os_pre_pass0() { return sizeof(K) == 2; }
constexpr int OS_TABLE_ROW = 2 * OS_RADIX;          // table columns: [pass 0 digits | pass 1 digits], each over all tiles
template <class K> static size_t os_table_words(size_t n) { return os_pre_pass0<K>() ? os_blocks<K>(n) * OS_TABLE_ROW : 0; }
template <class K> static size_t os_ctrl_bytes(size_t n, int passes)
{
    return align_up(((size_t)passes * OS_RADIX + (size_t)passes * os_blocks<K>(n) * OS_RADIX + 64 + os_table_words<K>(n)) * sizeof(uint32_t));
}

// all digit histograms in one read of the keys: 1024-thread workgroups (at most 512 of them, so the final flush stays a
// few hundred atomics per global bin), four independent 16-byte loads in flight per thread
constexpr int OS_HIST_BLOCK = 1024;
template <class K>
__global__ void __launch_bounds__(OS_HIST_BLOCK) os_hist_kernel(const K* __restrict__ keys, size_t n, OsPlan plan, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[OS_MAX_PASSES][OS_RADIX];
    for (int q = threadIdx.x; q < OS_MAX_PASSES * OS_RADIX; q += OS_HIST_BLOCK) (&s_h[0][0])[q] = 0;
    __syncthreads();
    constexpr int VEC = 16 / sizeof(K);                     // keys per 16-byte load
    const size_t nvec = n / VEC;
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);
    auto add = [&](uint32_t k) {
        int shift = 0;
        for (int p = 0; p < plan.passes; p++) {
            atomicAdd(&s_h[p][(k >> shift) & ((1u << plan.bits[p]) - 1u)], 1u);
            shift += plan.bits[p];
        }
    };
    auto add4 = [&](const uint4 v) {
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (sizeof(K) == 4) add(w[e]);
            else { add(w[e] & 0xffffu); add(w[e] >> 16); }
        }
    };
    const size_t stride = (size_t)gridDim.x * OS_HIST_BLOCK;
    size_t i = (size_t)blockIdx.x * OS_HIST_BLOCK + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const uint4 v0 = k4[i], v1 = k4[i + stride], v2 = k4[i + 2 * stride], v3 = k4[i + 3 * stride];
        add4(v0); add4(v1); add4(v2); add4(v3);
    }
    for (; i < nvec; i += stride) add4(k4[i]);
    for (size_t j = nvec * VEC + (size_t)blockIdx.x * OS_HIST_BLOCK + threadIdx.x; j < n; j += stride) add((uint32_t)keys[j]);
    __syncthreads();
    for (int q = threadIdx.x; q < plan.passes * OS_RADIX; q += OS_HIST_BLOCK) {
        const uint32_t v = (&s_h[0][0])[q];
        if (v) atomicAdd(&hist[q], v);
    }
}

// tile-by-tile histograms of the u16 sort: workgroup b counts the digits of tile b (the pass kernel's tile b) for the first
// two passes and stores them as row b of the table -- plain stores, no global atomics (the closing atomics of os_hist_kernel
// are what makes it slower with more workgroups)
__global__ void __launch_bounds__(1024) os_tile_hist_kernel(const uint16_t* __restrict__ keys, uint32_t n, OsPlan plan, uint32_t* __restrict__ table)
{
    constexpr int TILE = OsShape<uint16_t>::TILE;
    static_assert(TILE == 8 * 1024, "one 16-byte load of 8 keys per thread");
    __shared__ uint32_t s_h[OS_TABLE_ROW];
    if (threadIdx.x < OS_TABLE_ROW) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)TILE + threadIdx.x * 8u;
    const uint32_t m0 = (1u << plan.bits[0]) - 1u, m1 = plan.passes > 1 ? (1u << plan.bits[1]) - 1u : 0u;
    auto add = [&](uint32_t k) {
        atomicAdd(&s_h[k & m0], 1u);
        if (plan.passes > 1) atomicAdd(&s_h[OS_RADIX + ((k >> plan.bits[0]) & m1)], 1u);
    };
    if (base + 8 <= n && (reinterpret_cast<uintptr_t>(keys) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(keys + base);
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int e = 0; e < 4; e++) { add(w[e] & 0xffffu); add(w[e] >> 16); }
    } else {
        for (uint32_t i = base; i < min(base + 8u, n); i++) add((uint32_t)keys[i]);
    }
    __syncthreads();
    // column-major table (a column = one (pass, digit) over all tiles, contiguous for the scan); unused digit columns stay unwritten
    const int t = threadIdx.x;
    if (t < OS_TABLE_ROW && (t >> 8) < plan.passes && (t & (OS_RADIX - 1)) < (1 << plan.bits[(t >> 8) & 1]))
        table[(size_t)t * gridDim.x + blockIdx.x] = s_h[t];
}

// one workgroup per USED table column (pass, digit): the column's total goes to the global histogram of that pass; the pass-0
// columns are replaced by their exclusive prefixes over the tiles (what the look-back would have produced)
__global__ void __launch_bounds__(256) os_table_scan_kernel(uint32_t* __restrict__ table, uint32_t blocks, OsPlan plan, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_w[4];
    const int n0 = 1 << plan.bits[0];
    const int pass = (int)blockIdx.x < n0 ? 0 : 1, digit = (int)blockIdx.x - (pass ? n0 : 0), col = pass * OS_RADIX + digit;
    uint32_t* column = table + (size_t)col * blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < blocks; b0 += 256) {                 // 256 tiles per sweep, coalesced
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t c = b < blocks ? column[b] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        __syncthreads();                                            // previous sweep's s_w has been read
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; w++) off += s_w[w];
        if (pass == 0 && b < blocks) column[b] = off + incl - c;
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    if (threadIdx.x == 0) hist[col] = carry;                        // hist is [pass][256]
}

__device__ __forceinline__ uint32_t os_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void os_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// PRE: the tile's exclusive digit prefixes come from the table os_table_scan_kernel prepared (`status` = the table, no ticket,
// no look-back); otherwise they are found by the chained look-back over the earlier tiles' status words.
// PAY2 (depth sort): a SECOND 32-bit payload travels with every item -- the Gaussian's tile rectangle packed to four bytes
// (os_pack_rect: grids up to 255 x 255 tiles). The first pass reads the rectangles coalesced, in id order, from `gather_src`
// (p2in == null); the passes hand the packed word on through p2in / p2out; the last pass (p2out == null) unpacks it into
// gather_dst in sorted order. Without it the last pass GATHERS gather_src[id]: 3M random 8-byte reads = 3M x 128-byte lines,
// 42 us at the fabric's line rate for P = 3M (tools/step_timeline.py) against ~4 us more per pass for the extra payload.
__device__ __forceinline__ uint32_t os_pack_rect(uint2 r)
{
    return (r.x & 0xffu) | ((r.x >> 16) << 8) | ((r.y & 0xffu) << 16) | ((r.y >> 16) << 24);
}
__device__ __forceinline__ uint2 os_unpack_rect(uint32_t w)
{
    return make_uint2((w & 0xffu) | (((w >> 8) & 0xffu) << 16), ((w >> 16) & 0xffu) | ((w >> 24) << 16));
}

template <class K, int BITS, bool PRE, bool PAY2 = false>
__global__ void __launch_bounds__(OsShape<K>::BLOCK)
os_pass_kernel(const K* __restrict__ kin, K* __restrict__ kout, const uint32_t* __restrict__ vin, uint32_t* __restrict__ vout,
               uint32_t n, int shift, const uint32_t* __restrict__ hist, uint32_t* __restrict__ status, uint32_t* __restrict__ ticket,
               const uint2* __restrict__ gather_src, uint2* __restrict__ gather_dst, const uint32_t* __restrict__ p2in = nullptr,
               uint32_t* __restrict__ p2out = nullptr)
{
    constexpr uint32_t ERR_BIT = sizeof(K) == 2 ? 1u : 2u;
    constexpr uint32_t MASK = (1u << BITS) - 1;
    constexpr int OS_BLOCK = OsShape<K>::BLOCK, OS_IPT = OsShape<K>::IPT, OS_WAVES = OS_BLOCK / 64;
    __shared__ uint32_t s_cnt[OS_WAVES][OS_RADIX];   // per-wave digit counters, later exclusive prefixes across the waves
    __shared__ uint32_t s_start[OS_RADIX];           // start of every digit in the tile-sorted order
    __shared__ int32_t s_gbase[OS_RADIX];            // global position = s_gbase[digit] + tile-sorted position
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_bid;
    constexpr int OS_TILE = OsShape<K>::TILE;
    __shared__ K s_keys[OS_TILE];
    __shared__ uint32_t s_vals[OS_TILE];
    __shared__ uint32_t s_vals2[PAY2 ? OS_TILE : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bid = PRE ? blockIdx.x : atomicAdd(ticket, 1u);
    for (int q = tid; q < OS_WAVES * OS_RADIX; q += OS_BLOCK) (&s_cnt[0][0])[q] = 0;
    __syncthreads();
    const uint32_t bid = s_bid;
    OS_T(0)
    const uint32_t block_start = bid * (uint32_t)OS_TILE;
    const uint32_t valid = min((uint32_t)OS_TILE, n - block_start);

    K key[OS_IPT];
    uint32_t val[OS_IPT], rank[OS_IPT];
    uint32_t val2[PAY2 ? OS_IPT : 1];
    const uint32_t wbase = block_start + (uint32_t)wave * 64u * OS_IPT;
#pragma unroll
    for (int k = 0; k < OS_IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        const bool ok = idx < n;
        key[k] = ok ? kin[idx] : (K)0;
        val[k] = ok ? (vin ? vin[idx] : idx) : 0u;             // no payload array: the payload is the item's index
        if (PAY2) val2[k] = ok ? (p2in ? p2in[idx] : os_pack_rect(gather_src[idx])) : 0u;
    }
    // ranking: lanes with the same digit find each other with BITS ballots. Written on 32-bit halves with the digit bit as a
    // 0 / -1 mask so that a bit costs six vector instructions (v_bfe_i32, v_cmp, 2 x v_xnor, 2 x v_and); the obvious
    // `peers &= bit ? bal : ~bal` compiled to 16 (two compares per bit, a 64-bit select built from v_cndmask + v_lshl_add_u64).
    // The per-wave digit counters are read and bumped through plain LDS instructions (wavefront-scope relaxed atomics): a
    // `volatile` pointer made them FLAT loads / stores with system-scope cache bits and a full vmcnt(0) wait each -- two memory
    // round trips per item, which was most of this loop's time (tools/sort_phases.py: rank loop 47 % of a depth-key pass).
    const uint32_t lt_lo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu, lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
#ifdef C3DGS_OS_TIMING
    if (key[0] == (K)0x12345678 && val[OS_IPT - 1] == 0x87654321u) kout[0] = key[OS_IPT - 1];   // force the loads to complete here
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    OS_T(1)
    uint32_t* wc = s_cnt[wave];                       // other lanes of the wave update these between iterations
#pragma unroll
    for (int k = 0; k < OS_IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        const bool ok = idx < n;
        const uint32_t d = ((uint32_t)key[k] >> shift) & MASK;
        const unsigned long long okb = __ballot(ok);  // padding lanes of the last tile take no part
        uint32_t plo = (uint32_t)okb, phi = (uint32_t)(okb >> 32);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const int sgn = __builtin_amdgcn_sbfe((int)d, b, 1);                      // 0 or -1
            const unsigned long long bal = __ballot(sgn != 0);
            plo &= ~((uint32_t)bal ^ (uint32_t)sgn);                                   // bit set: keep bal; clear: keep ~bal
            phi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)sgn);
        }
        rank[k] = 0;
        if (ok) {
            const uint32_t before = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint32_t below = (uint32_t)__popc(plo & lt_lo) + (uint32_t)__popc(phi & lt_hi);
            rank[k] = before + below;
            if (below == 0) __hip_atomic_store(&wc[d], before + (uint32_t)__popc(plo) + (uint32_t)__popc(phi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    OS_T(2)
    __syncthreads();
    OS_T(3)
    // thread d (< 256): the tile's count of digit d, exclusive prefixes across the waves, then across the digits
    uint32_t tot = 0;
    if (tid < OS_RADIX) {
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = tot; tot += c; }
    }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (tid < OS_RADIX && lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t start = 0, pre = 0;
    if (tid < OS_RADIX) {
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        start = off + incl - tot;
        s_start[tid] = start;
        // decoupled look-back over the earlier tiles for digit `tid`
        uint32_t* my = status + (size_t)bid * OS_RADIX + tid;
        if (PRE) pre = tid < (1 << BITS) ? status[(size_t)tid * gridDim.x + bid] : 0u;   // column-major table, pass-0 columns
        else if (bid == 0) os_store(my, OS_FLAG_PRE | tot);
        else {
            os_store(my, OS_FLAG_AGG | tot);
            for (int64_t b = (int64_t)bid - 1;; b--) {
                uint32_t s, spins = 0;
                do { s = os_load(status + (size_t)b * OS_RADIX + tid); } while ((s >> 30) == 0 && ++spins < OS_SPIN_LIMIT);
                if ((s >> 30) == 0) { atomicOr(&g_os_error, ERR_BIT); break; }   // never hang the device: give up, flag it
                pre += s & OS_CNT_MASK;
                if (s & OS_FLAG_PRE) break;
            }
            os_store(my, OS_FLAG_PRE | (pre + tot));
        }
    }
    OS_T(4)
    // exclusive scan of the global digit histogram by the same 256 threads
    const uint32_t h = tid < OS_RADIX ? hist[tid] : 0u;
    uint32_t hincl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(hincl, o); if (lane >= o) hincl += t; }
    __syncthreads();
    if (tid < OS_RADIX && lane == 63) s_wtot[wave] = hincl;
    __syncthreads();
    if (tid < OS_RADIX) {
        uint32_t off = 0;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        s_gbase[tid] = (int32_t)(off + hincl - h + pre) - (int32_t)start;
    }
    OS_T(5)
    // stable reorder of the tile through LDS
#pragma unroll
    for (int k = 0; k < OS_IPT; k++) {
        const uint32_t idx = wbase + (uint32_t)k * 64u + lane;
        if (idx < n) {
            const uint32_t d = ((uint32_t)key[k] >> shift) & MASK;
            const uint32_t p = s_start[d] + s_cnt[wave][d] + rank[k];
            s_keys[p] = key[k];
            s_vals[p] = val[k];
            if (PAY2) s_vals2[p] = val2[k];
        }
    }
    __syncthreads();
    OS_T(6)
#pragma unroll
    for (int m = 0; m < OS_IPT; m++) {
        const uint32_t p = (uint32_t)tid + (uint32_t)m * OS_BLOCK;
        if (p < valid) {
            const K kk = s_keys[p];
            const uint32_t d = ((uint32_t)kk >> shift) & MASK;
            const uint32_t g = (uint32_t)(s_gbase[d] + (int32_t)p);
            kout[g] = kk;
            const uint32_t vv = s_vals[p];
            vout[g] = vv;
            if (PAY2) {
                if (p2out) p2out[g] = s_vals2[p];
                else gather_dst[g] = os_unpack_rect(s_vals2[p]);   // last pass: the rectangles arrive with their items
            } else if (gather_src) gather_dst[g] = gather_src[vv]; // last pass of the depth sort: per-Gaussian data in sorted order
        }
    }
#ifdef C3DGS_OS_TIMING
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    OS_T(7)
}

template <class K, bool PRE>
static void os_launch_pass(int bits, unsigned blocks, hipStream_t s, const K* ki, K* ko, const uint32_t* vi, uint32_t* vo, uint32_t n,
                           int shift, const uint32_t* hist, uint32_t* status, uint32_t* ticket, const uint2* gsrc, uint2* gdst)
{
#define C3DGS_OS_CASE(B) case B: os_pass_kernel<K, B, PRE><<<blocks, OsShape<K>::BLOCK, 0, s>>>(ki, ko, vi, vo, n, shift, hist, status, ticket, gsrc, gdst); break
    switch (bits) {
        C3DGS_OS_CASE(1); C3DGS_OS_CASE(2); C3DGS_OS_CASE(3); C3DGS_OS_CASE(4);
        C3DGS_OS_CASE(5); C3DGS_OS_CASE(6); C3DGS_OS_CASE(7);
        default: os_pass_kernel<K, 8, PRE><<<blocks, OsShape<K>::BLOCK, 0, s>>>(ki, ko, vi, vo, n, shift, hist, status, ticket, gsrc, gdst); break;
    }
#undef C3DGS_OS_CASE
}

// temp = [control block | ping buffer (keys, values) | pong buffer]; the input arrays are never written
template <class K>
static size_t os_temp_bytes(size_t n, int total_bits, bool pay2 = false)
{
    const OsPlan plan = os_plan(total_bits);
    size_t b = os_ctrl_bytes<K>(n, plan.passes);
    const int bufs = plan.passes >= 3 ? 2 : (plan.passes == 2 ? 1 : 0);
    b += (size_t)bufs * (align_up(n * sizeof(K)) + align_up(n * sizeof(uint32_t)));
    if (pay2) b += 2 * align_up(n * sizeof(uint32_t));       // ping-pong of the second payload, behind everything else
    return b;
}

// the control words at the front of `temp` that must be ZERO when the sort starts (digit histograms the histogram kernel adds
// into, look-back status words, tickets). os_sort clears them itself unless the caller says a kernel it ran just before on the
// same stream already did (`ctrl_cleared`: the rasterizer folds the two clears of a forward into preprocess / duplicate_with_keys,
// whose thousands of workgroups do it for free -- a separate fill launch costs ~5 us of an otherwise idle GPU each)
template <class K>
static size_t os_clear_bytes(size_t n, int total_bits)
{
    const OsPlan plan = os_plan(total_bits);
    if (!os_pre_pass0<K>()) return os_ctrl_bytes<K>(n, plan.passes);
    return ((size_t)plan.passes * OS_RADIX + (size_t)plan.passes * os_blocks<K>(n) * OS_RADIX + 64) * sizeof(uint32_t);   // the table behind is written in full
}

template <class K>
static hipError_t os_sort(void* temp, size_t temp_bytes, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, size_t n,
                          int total_bits, hipStream_t s, const uint2* gather_src = nullptr, uint2* gather_dst = nullptr,
                          bool ctrl_cleared = false, bool pay2 = false)
{
    if (n == 0) return hipSuccess;
    // second payload (see os_pass_kernel): u32 keys in four 8-bit passes with a rectangle table only, and only if the caller's
    // scratch has room for its two buffers
    if (pay2 && (sizeof(K) != 4 || total_bits != 32 || !gather_src || !gather_dst || vin || os_temp_bytes<K>(n, total_bits, true) > temp_bytes))
        pay2 = false;
    if (n >= ((size_t)1 << 30) || os_temp_bytes<K>(n, total_bits) > temp_bytes) return hipErrorInvalidValue;
    const OsPlan plan = os_plan(total_bits);
    const size_t blocks = os_blocks<K>(n);
    char* base = (char*)temp;
    const size_t ctrl = os_ctrl_bytes<K>(n, plan.passes);
    uint32_t* hist = (uint32_t*)base;
    uint32_t* status = hist + (size_t)plan.passes * OS_RADIX;
    uint32_t* ticket = status + (size_t)plan.passes * blocks * OS_RADIX;
    uint32_t* table = ticket + 64;                                  // [OS_TABLE_ROW columns][blocks], u16 sort only
    constexpr bool pre0 = os_pre_pass0<K>();
    K* tk[2]; uint32_t* tv[2];
    char* q = base + ctrl;
    for (int i = 0; i < 2; i++) { tk[i] = (K*)q; q += align_up(n * sizeof(K)); tv[i] = (uint32_t*)q; q += align_up(n * sizeof(uint32_t)); }
    uint32_t* t2[2] = { nullptr, nullptr };
    if (pay2) {
        q = base + os_temp_bytes<K>(n, total_bits);
        for (int i = 0; i < 2; i++) { t2[i] = (uint32_t*)q; q += align_up(n * sizeof(uint32_t)); }
    }
    // the table is written in full by os_tile_hist_kernel: only the words in front of it need clearing
    if (!ctrl_cleared) {
        const hipError_t e = hipMemsetAsync(base, 0, os_clear_bytes<K>(n, total_bits), s);
        if (e != hipSuccess) return e;
    }
    if constexpr (pre0) {
        os_tile_hist_kernel<<<(unsigned)blocks, 1024, 0, s>>>((const uint16_t*)kin, (uint32_t)n, plan, table);
        os_table_scan_kernel<<<(unsigned)((1 << plan.bits[0]) + (plan.passes > 1 ? (1 << plan.bits[1]) : 0)), 256, 0, s>>>(table, (uint32_t)blocks, plan, hist);
    } else {
        const size_t nvec16 = n * sizeof(K) / 16 + 1;
        // 4 x 16 bytes per thread, at most 512 workgroups: measured optimum on both sorts (fewer workgroups: LDS-atomic bound;
        // more: the closing global atomics, one per workgroup and non-empty bin on the same few hundred words, take over)
        const unsigned hb = (unsigned)std::min<size_t>((nvec16 + OS_HIST_BLOCK * 4 - 1) / (OS_HIST_BLOCK * 4), 512);
        os_hist_kernel<K><<<hb, OS_HIST_BLOCK, 0, s>>>(kin, n, plan, hist);
    }
    int shift = 0;
    for (int p = 0; p < plan.passes; p++) {
        const K* ki = p == 0 ? kin : tk[(p - 1) & 1];
        const uint32_t* vi = p == 0 ? vin : tv[(p - 1) & 1];
        K* ko = p == plan.passes - 1 ? kout : tk[p & 1];
        uint32_t* vo = p == plan.passes - 1 ? vout : tv[p & 1];
        const bool last = p == plan.passes - 1;
        // ticket + p is this pass's counter
        if constexpr (sizeof(K) == 4) {
            if (pay2) {
                os_pass_kernel<K, 8, false, true><<<(unsigned)blocks, OsShape<K>::BLOCK, 0, s>>>(
                    ki, ko, vi, vo, (uint32_t)n, shift, hist + (size_t)p * OS_RADIX, status + (size_t)p * blocks * OS_RADIX, ticket + p,
                    gather_src, gather_dst, p == 0 ? nullptr : t2[(p - 1) & 1], last ? nullptr : t2[p & 1]);
                shift += plan.bits[p];
                continue;
            }
        }
        if (pre0 && p == 0)
            os_launch_pass<K, true>(plan.bits[p], (unsigned)blocks, s, ki, ko, vi, vo, (uint32_t)n, shift, hist, table, ticket,
                                    last ? gather_src : nullptr, last ? gather_dst : nullptr);
        else
            os_launch_pass<K, false>(plan.bits[p], (unsigned)blocks, s, ki, ko, vi, vo, (uint32_t)n, shift, hist + (size_t)p * OS_RADIX,
                                     status + (size_t)p * blocks * OS_RADIX, ticket + p, last ? gather_src : nullptr, last ? gather_dst : nullptr);
        shift += plan.bits[p];
    }
    return hipGetLastError();
}

#ifdef C3DGS_OS_TIMING
int os_read_times(unsigned long long* out512)
{
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_os_times), sizeof(unsigned long long) * 512) != hipSuccess;
}
#else
int os_read_times(unsigned long long*) { return 1; }
#endif

// address of the current device's sticky error word (cached per device; one process per GPU is the normal case)
uint32_t* onesweep_error_word()
{
    static uint32_t* cache[64] = { nullptr };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!cache[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_os_error)) != hipSuccess) return nullptr;
        cache[dev] = (uint32_t*)p;
    }
    return cache[dev];
}

// debug helper: did any look-back so far time out on this device? (synchronises the stream; clears the word)
int onesweep_timed_out(hipStream_t s)
{
    if (!onesweep_enabled()) return 0;
    uint32_t* e = onesweep_error_word();
    uint32_t w = 1;
    if (!e || hipMemcpyAsync(&w, e, sizeof(w), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (w) (void)hipMemsetAsync(e, 0, sizeof(w), s);
    return w != 0;
}

bool onesweep_enabled()
{
    static const bool on = []() { const char* e = std::getenv("C3DGS_SORT_ROCPRIM"); return !(e && e[0] == '1'); }();
    return on;
}
size_t onesweep_depth_temp_bytes(int P) { return os_temp_bytes<uint32_t>((size_t)(P > 0 ? P : 1), 32, true); }   // room for the second payload
size_t onesweep_tile_temp_bytes(int R, int end_bit, int key_bytes)
{
    return key_bytes == 4 ? os_temp_bytes<uint32_t>((size_t)(R > 0 ? R : 1), end_bit) : os_temp_bytes<uint16_t>((size_t)(R > 0 ? R : 1), end_bit);
}
size_t onesweep_depth_clear_bytes(int P) { return os_clear_bytes<uint32_t>((size_t)(P > 0 ? P : 1), 32); }
size_t onesweep_tile_clear_bytes(int R, int end_bit, int key_bytes)
{
    return key_bytes == 4 ? os_clear_bytes<uint32_t>((size_t)(R > 0 ? R : 1), end_bit) : os_clear_bytes<uint16_t>((size_t)(R > 0 ? R : 1), end_bit);
}
hipError_t onesweep_tile_sort32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                                int R, int end_bit, hipStream_t s, bool ctrl_cleared)
{
    return os_sort<uint32_t>(temp, temp_bytes, kin, kout, vin, vout, (size_t)R, end_bit, s, nullptr, nullptr, ctrl_cleared);
}
hipError_t onesweep_depth_sort(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                               int P, const uint2* gather_src, uint2* gather_dst, hipStream_t s, bool ctrl_cleared, bool rects_fit_bytes)
{
    // rects_fit_bytes: every coordinate of the tile rectangles is below 256 (grid of at most 255 x 255 tiles), so they can
    // travel with the items as a packed 32-bit second payload instead of being gathered behind the last pass
    static const bool no_pay2 = []() { const char* e = std::getenv("C3DGS_DEPTH_SORT_GATHER"); return e && e[0] == '1'; }();
    return os_sort<uint32_t>(temp, temp_bytes, kin, kout, vin, vout, (size_t)P, 32, s, gather_src, gather_dst, ctrl_cleared,
                             rects_fit_bytes && !no_pay2);
}
hipError_t onesweep_tile_sort(void* temp, size_t temp_bytes, const uint16_t* kin, uint16_t* kout, const uint32_t* vin, uint32_t* vout,
                              int R, int end_bit, hipStream_t s, bool ctrl_cleared)
{
    return os_sort<uint16_t>(temp, temp_bytes, kin, kout, vin, vout, (size_t)R, end_bit, s, nullptr, nullptr, ctrl_cleared);
}

} // namespace c3dgs
