// render.hip -- per-tile alpha blending, forward (K9) and backward (K10), for gfx950.
//
//   K9   reference FORWARD::renderCUDA   cuda_rasterizer/forward.cu:270-383
//   K10  reference BACKWARD::renderCUDA  cuda_rasterizer/backward.cu:399-557
//
// MI355X design (not a translation of the CUDA kernels):
//  * one 16x16 tile per 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel quadrant;
//  * per staged Gaussian an exact-conservative test decides which quadrants it can reach at all (minimum of the
//    quadratic form over the quadrant rectangle vs ln(255*opacity)); the four 256-bit masks are iterated as scalar
//    bitmasks, so a wave never spends an instruction on a Gaussian that cannot touch its pixels;
//  * the tile's slice of the sorted point list is read coalesced, the 48-byte splat records are gathered one cache
//    line each into a double-buffered LDS batch (one barrier per batch), the next batch's gather is issued before
//    the current batch is blended; the forward blend is branch-free (selects), early-out is a per-wave ballot plus
//    four LDS flags read after the batch barrier;
//  * workgroup ids are remapped so that each XCD (own L2) gets a contiguous band of tiles;
//  * backward: NO global atomics. Per pixel only 9 raw sums (3 colour terms + 6 moments of w = G*dL/dalpha about the
//    mean) are formed; 7 Gaussians x 9 sums are reduced over the wave together by a transposing DPP / permlane-swap
//    network, the 4 waves are combined through LDS, and one plain 36-byte store (+ a flag byte) per (Gaussian, tile)
//    instance goes to a slot array laid out in Gaussian-id order; the per-Gaussian kernel (backward_preprocess.hip)
//    then sums a contiguous run of slots. Plain stores run ~4-5x the chip-wide float-atomic rate on MI355X and the
//    result is bitwise reproducible (the reference's 9 atomics per pixel-Gaussian pair, backward.cu:523-554, are not).
#include "common.hpp"
#include <algorithm>

namespace c3dgs {

constexpr int BATCH = 256;

// Lane-efficiency counters of the blend kernels (test-only build variant "lanes": -DC3DGS_COUNT_LANES, c3dgs_amd/build.py).
// [fwd = 0 | bwd = 8] + { 0: (wave, Gaussian) pairs the blend loop ran (real list entries), 1: slots incl. the sentinel padding,
// 2: lanes whose pixel used the pair (forward: blended it; backward: hit), 3: pairs with at least one such lane,
// 4: sum over (wave, list) of max(pairs touching pixel rows 0-3, pairs touching rows 4-7) = iterations of a half-wave
//    (8x4-pixel) scheduling unit, 5: the same for four 4x4-pixel blocks, 6: lists walked, 7: lanes hit (forward, incl. finished pixels) }
// The product build never touches them; c3dgs_debug_lane_counters() reads and clears them.
__device__ unsigned long long g_lane_counters[16];

#ifndef C3DGS_BWD_ABLATE
#define C3DGS_BWD_ABLATE 0      // timing-only experiment builds (WRONG gradients): bit 0 = no partial-sum stores, bit 1 = cache-resident
                                // record gathers, bit 2 = without the two in-bank reduction levels, bit 3 = without any reduction, bit 4 = records read
                                // coalesced by list position (what parking the forward's staged records would give)
#endif
#ifdef C3DGS_BWD_TIMING
// phase clocks of render_backward (experiment build variant "bwdtime"): shader-clock ticks summed over all waves:
// g_lane_counters[8 + {0: staging incl. its two barriers, 1: list compaction, 2: group loop, 3: flush incl. barrier, 4: prologue, 5: waves}]
#define BT_STAMP(var) const unsigned long long var = __builtin_readcyclecounter();
#else
#define BT_STAMP(var)
#endif
#ifdef C3DGS_COUNT_LANES
struct LaneCount {
    unsigned long long pairs = 0, slots = 0, lanes = 0, live = 0, half = 0, blk = 0, lists = 0, aux = 0;
    int nA = 0, nB = 0, nb[4] = { 0, 0, 0, 0 };
    __device__ void pair(bool real, unsigned long long used, unsigned long long aux_mask)
    {
        slots++;
        if (!real) return;
        pairs++;
        lanes += __popcll(used);
        aux += __popcll(aux_mask);
        live += used != 0;
        nA += (used & 0x00000000ffffffffull) != 0;
        nB += (used & 0xffffffff00000000ull) != 0;
        nb[0] += (used & 0x000000000f0f0f0full) != 0;
        nb[1] += (used & 0x00000000f0f0f0f0ull) != 0;
        nb[2] += (used & 0x0f0f0f0f00000000ull) != 0;
        nb[3] += (used & 0xf0f0f0f000000000ull) != 0;
    }
    __device__ void end_list()
    {
        lists++;
        half += max(nA, nB);
        blk += max(max(nb[0], nb[1]), max(nb[2], nb[3]));
        nA = nB = nb[0] = nb[1] = nb[2] = nb[3] = 0;
    }
    __device__ void flush(int base, int lane)
    {
        if (lane != 0) return;
        const unsigned long long v[8] = { pairs, slots, lanes, live, half, blk, lists, aux };
        for (int q = 0; q < 8; q++) atomicAdd(&g_lane_counters[base + q], v[q]);
    }
};
#endif

// The staged LDS records carry the conic PRE-SCALED for the blend loops: {-0.5 a, -b, -0.5 c} x log2(e), so that the
// exponent of  G = exp(-0.5 (a dx^2 + c dy^2) - b dx dy)  is three multiplies and three multiply-adds feeding v_exp_f32
// (= 2^x) directly: two vector instructions per (pixel, Gaussian) pair fewer than the reference's form + exp (which is
// exp2 of a product with log2(e) on this hardware anyway). Done once per staged entry, after the culling mask, which
// works on the true conic.
__device__ __forceinline__ void prescale_conic(float4& a, float4& b)
{
    constexpr float LOG2E = 1.4426950408889634f;
    a.z *= -0.5f * LOG2E;
    a.w *= -LOG2E;
    b.x *= -0.5f * LOG2E;
}

// alpha of one Gaussian at one pixel; the SAME instruction sequence in forward and backward so both
// take identical skip decisions (explicit fma placement, independent of -ffp-contract).
// (ka, kb, kc) = the pre-scaled conic of prescale_conic().
// returns false when the reference `continue`s (forward.cu:344-354 / backward.cu:494-501).
__device__ __forceinline__ bool gaussian_alpha(float mx, float my, float ka, float kb, float kc, float op,
                                               float pxf, float pyf, float& dx, float& dy, float& G, float& alpha)
{
    dx = mx - pxf;
    dy = my - pyf;
    const float power2 = fmaf(kb, dx * dy, fmaf(kc, dy * dy, ka * (dx * dx)));     // = power * log2(e)
    // no early return: G and alpha are always written (callers select on the result anyway), which saves the compiler
    // a select per call; for power > 0 they hold values nobody uses
    G = __builtin_amdgcn_exp2f(power2);
    alpha = fminf(0.99f, op * G);
    return !(power2 > 0.0f) && !(alpha < 1.0f / 255.0f);
}

// Which of the tile's four 8x8 quadrants (= waves) can this Gaussian touch at all?  A pixel only blends the
// Gaussian if alpha = min(0.99, o*exp(power)) >= 1/255, i.e. f(d) = 0.5*d^T C d <= tau with tau = ln(255*o).
// f is convex, so its minimum over a quadrant's pixel rectangle is 0 if the mean lies inside it and otherwise sits
// on one of the four edges, where it is a clamped 1-D parabola minimum. A quadrant whose minimum exceeds tau
// (inflated by 1e-4 relative + 1e-4 absolute and 0.01 px of rectangle slack, far above the fp32 error of the
// per-pixel test) contains no contributing pixel, so skipping it changes no result. Bit q = qy*2+qx.
// Anything non-finite or non-positive-definite -> no culling.
__device__ __forceinline__ float min_power_on_rect(float ca, float cb, float cc, float nb_c, float nb_a,
                                                   float dx0, float dx1, float dy0, float dy1)
{
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return 0.f;
    auto f = [&](float dx, float dy) { return 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy; };
    const float ya = fminf(fmaxf(nb_c * dx0, dy0), dy1), yb = fminf(fmaxf(nb_c * dx1, dy0), dy1);
    const float xa = fminf(fmaxf(nb_a * dy0, dx0), dx1), xb = fminf(fmaxf(nb_a * dy1, dx0), dx1);
    return fminf(fminf(f(dx0, ya), f(dx1, yb)), fminf(f(xa, dy0), f(xb, dy1)));
}

__device__ __forceinline__ uint32_t quadrant_mask(const float4 a, const float4 b, float tile_x0, float tile_y0)
{
    const float mx = a.x, my = a.y, ca = a.z, cb = a.w, cc = b.x, op = b.y;
    if (op < (1.0f / 255.0f) * 0.999f) return 0u;          // alpha <= o < 1/255 everywhere
    const float det = ca * cc - cb * cb;
    const float tau = __logf(255.0f * op) * 1.0001f + 1e-4f;
    if (!(det > 0.0f) || !(ca > 0.0f) || !(cc > 0.0f) || !(tau < 1e30f)) return 0xfu;
    const float nb_c = -cb / cc, nb_a = -cb / ca;           // argmin of f along a vertical / horizontal line
    uint32_t mask = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float x0 = tile_x0 + (float)((q & 1) * 8) - 0.01f - mx, y0 = tile_y0 + (float)((q >> 1) * 8) - 0.01f - my;
        const float fmin = min_power_on_rect(ca, cb, cc, nb_c, nb_a, x0, x0 + 7.02f, y0, y0 + 7.02f);
        mask |= (uint32_t)(!(fmin > tau)) << q;              // NaN -> keep
    }
    return mask;
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2) under the observed
// round-robin dispatch, so give every XCD one contiguous band of row-major tiles. Speed only.
__device__ __forceinline__ int tile_of_block(int b, int T)
{
    const int chunk = (T + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// Tile schedule of the backward: tiles in descending order of the work they carry (tile_used = entries the tile really
// visits, known from the forward), so that the last workgroups to start are the short ones: -3 % on the bench scene.
// (The forward only knows its list lengths in advance; ordering it by those measured slower than the XCD-banded order.)
// One workgroup: counting sort on tile_used / 8 (1024 buckets). The same launch clears what the backward needs cleared (the
// written flags of the partial sums, the scatter-added codebook gradients): workgroup 0 orders the tiles, the others
// fill -- one launch instead of a 9 us single-workgroup kernel behind two or three fill launches.
__global__ void __launch_bounds__(1024) backward_prep_kernel(int T, const uint32_t* __restrict__ tile_used, uint32_t* __restrict__ order,
                                                              uint4* __restrict__ zero_a, size_t n16_a, uint4* __restrict__ zero_b, size_t n16_b)
{
    if (blockIdx.x > 0) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        const size_t stride = (size_t)(gridDim.x - 1) * 1024;
        for (size_t i = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < n16_a; i += stride) zero_a[i] = z;
        for (size_t i = (size_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < n16_b; i += stride) zero_b[i] = z;
        return;
    }
    if (T <= 0) return;
    auto work = [&](int i) { return tile_used[i]; };
    __shared__ uint32_t s_cnt[1024], s_w[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    s_cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < T; i += 1024) atomicAdd(&s_cnt[1023 - min(work(i) >> 3, 1023u)], 1u);   // bucket 0 = longest
    __syncthreads();
    const uint32_t c = s_cnt[t];
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; w++) off += s_w[w];
    s_cnt[t] = off + incl - c;                         // start of the bucket
    __syncthreads();
    for (int i = t; i < T; i += 1024) order[atomicAdd(&s_cnt[1023 - min(work(i) >> 3, 1023u)], 1u)] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
render_forward_kernel(int W, int H, int gx, int T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float4* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ out_color,
                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_used,
                      uint8_t* __restrict__ qmask, const uint32_t* __restrict__ sort_err)
{
    const int tile = tile_of_block(blockIdx.x, T);
    if (tile >= T || (blockIdx.x >> 3) >= ((T + 7) >> 3)) return;
    // A staged entry is a 32-byte record {x, y, conic a, b | conic c, opacity, r, g} plus its blue in a second array. The
    // candidate lists hold the records' BYTE OFFSETS inside s_ab, so a list word read back from LDS is the address operand
    // of the two ds_read_b128 as it is (the blue's address is that offset >> 3): no scalar unpacking, no readfirstlane, one
    // shift per Gaussian instead of three v_mov (the blend loop is VALU-issue bound). Entry BATCH of each buffer: sentinel
    // with opacity 0 (blends nothing).
    __shared__ float4 s_ab[2][BATCH + 1][2];
    __shared__ float s_c[2][BATCH + 1];
    __shared__ uint32_t s_list[4][BATCH + 8];        // per wave: its candidates of the batch, padded to a multiple of 8
    __shared__ int s_wdone[2][4];
    __shared__ unsigned long long s_mask[2][4][4];   // [buf][quadrant][staging wave]: which staged Gaussians reach it
    __shared__ uint32_t s_used;
    constexpr uint32_t REC_BYTES = 32, BUF_BYTES = (BATCH + 1) * REC_BYTES;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    bool done = !inside;
    if (tid == 0) s_used = 0;
    if (tid < 2) { s_ab[tid][BATCH][0] = make_float4(0, 0, 0, 0); s_ab[tid][BATCH][1] = make_float4(0, 0, 0, 0); s_c[tid][BATCH] = 0.f; }
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    // A sort whose look-back timed out (radix_sort.hip) leaves positions of the point list nobody wrote: their stale contents
    // must never be used as Gaussian ids. duplicate_with_keys / identify_ranges bail out on the same word (the ranges then stay
    // empty); a time-out of the LAST digit pass is caught here: the tile walks nothing and the image is NaN, not a plausible wrong
    // one. (wave-uniform scalar load)
    const bool poisoned = *sort_err != 0u;
    const uint2 range = poisoned ? make_uint2(0u, 0u) : ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + BATCH - 1) / BATCH;

    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    constexpr uint32_t NO_ENTRY = 0xffffffffu;
#ifdef C3DGS_COUNT_LANES
    LaneCount lc;
#endif
    const char* rec_base = reinterpret_cast<const char*>(&s_ab[0][0][0]);
    const char* blue_base = reinterpret_cast<const char*>(&s_c[0][0]);

    float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
    if (tid < n) {
        const uint32_t id = point_list[range.x + tid];
        ra = splat[3 * (size_t)id]; rb = splat[3 * (size_t)id + 1]; rc = splat[3 * (size_t)id + 2];
    }
    for (int r = 0; r < rounds; r++) {
        const int buf = r & 1;
        const uint32_t qm = (r * BATCH + tid < n) ? quadrant_mask(ra, rb, tile_x0, tile_y0) : 0u;
        prescale_conic(ra, rb);
        s_ab[buf][tid][0] = ra; s_ab[buf][tid][1] = rb; s_c[buf][tid] = rc.x;
        if (r * BATCH + tid < n) qmask[range.x + r * BATCH + tid] = (uint8_t)qm;   // the backward reuses it (same test, ~100 VALU ops)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long bm = __ballot((qm >> q) & 1u);
            if (lane == 0) s_mask[buf][q][wave] = bm;
        }
        const bool wave_done = __all(done);
        if (lane == 0) s_wdone[buf][wave] = wave_done;
        __syncthreads();
        if (s_wdone[buf][0] & s_wdone[buf][1] & s_wdone[buf][2] & s_wdone[buf][3]) break; // forward.cu:318-320
        const int nxt = (r + 1) * BATCH + tid;
        if (nxt < n) {                           // prefetch the next batch behind this batch's blending
            const uint32_t id = point_list[range.x + nxt];
            ra = splat[3 * (size_t)id]; rb = splat[3 * (size_t)id + 1]; rc = splat[3 * (size_t)id + 2];
        }
        if (wave_done) continue;
        // This wave's candidates of the batch, compacted into LDS once (record offsets, padded with the sentinel's to a
        // multiple of 8): the blend loop then needs no bit scanning and no per-Gaussian scalar control flow.
        int nw = 0;
        {
            const unsigned long long lt = (1ull << lane) - 1ull;
            const uint32_t buf_off = (uint32_t)buf * BUF_BYTES;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const unsigned long long m = uniform_u64(s_mask[buf][wave][c]);   // Gaussians that can reach this quadrant
                if ((m >> lane) & 1ull) s_list[wave][nw + (int)__popcll(m & lt)] = buf_off + (uint32_t)(c * 64 + lane) * REC_BYTES;
                nw += (int)__popcll(m);
            }
            if (lane < 8) s_list[wave][nw + lane] = buf_off + (uint32_t)BATCH * REC_BYTES;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        uint32_t last_off = NO_ENTRY;                    // record offset of the last Gaussian this pixel blended in this batch
        // (Requesting record g + 1 before Gaussian g is blended, as render_backward does, measured the same here, 0.265-0.274 ms, and
        // requesting the next row's list words and first record ahead measured slower, 0.288 ms; the compiler's own placement stays.)
        for (int k = 0; k < nw; k += 8) {
            // the wave's early-out is tested once per row of 8, not per Gaussian: finished pixels blend nothing either
            // way, and a per-Gaussian test (ballot + scalar branch) costs more than the work it saves
            if (__all(done)) break;
            const uint4 row0 = *reinterpret_cast<const uint4*>(&s_list[wave][k]);
            const uint4 row1 = *reinterpret_cast<const uint4*>(&s_list[wave][k + 4]);
            const uint32_t e[8] = { row0.x, row0.y, row0.z, row0.w, row1.x, row1.y, row1.z, row1.w };
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const float4 a = *reinterpret_cast<const float4*>(rec_base + e[g]);
                const float4 b = *reinterpret_cast<const float4*>(rec_base + e[g] + 16);
                const float cblue = *reinterpret_cast<const float*>(blue_base + (e[g] >> 3));   // 4-byte pitch = 32-byte pitch / 8
                float dx, dy, G, alpha;
                const bool hit = gaussian_alpha(a.x, a.y, a.z, a.w, b.x, b.y, pxf, pyf, dx, dy, G, alpha);
                // branch-free blend (selects instead of nested exec-mask regions)
                const bool live_px = hit && !done;
                const float test_T = Tr * (1.f - alpha);
                const bool stop = live_px && test_T < 0.0001f;   // forward.cu:355-360: stop BEFORE blending this one
                const bool blend = live_px != stop;              // stop implies live_px: one mask xor, no second compare
                const float w = blend ? alpha * Tr : 0.f;
                C0 = fmaf(b.z, w, C0); C1 = fmaf(b.w, w, C1); C2 = fmaf(cblue, w, C2);
                Tr = blend ? test_T : Tr;
                last_off = blend ? e[g] : last_off;
                done = done || stop;
#ifdef C3DGS_COUNT_LANES
                lc.pair(k + g < nw, __ballot(blend), __ballot(hit));
#endif
            }
        }
#ifdef C3DGS_COUNT_LANES
        lc.end_list();
#endif
        // offset -> 1-based position in the tile's list (once per batch, not per Gaussian)
        if (last_off != NO_ENTRY) last_contributor = (uint32_t)(r * BATCH) + ((last_off - (uint32_t)buf * BUF_BYTES) >> 5) + 1u;
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px, HW = (size_t)H * W;
        final_T[pix] = Tr;
        n_contrib[pix] = last_contributor;
        const float poison = poisoned ? __uint_as_float(0x7fc00000u) : 0.f;
        out_color[pix] = fmaf(Tr, bg[0], C0) + poison;
        out_color[HW + pix] = fmaf(Tr, bg[1], C1) + poison;
        out_color[2 * HW + pix] = fmaf(Tr, bg[2], C2) + poison;
    }
#ifdef C3DGS_COUNT_LANES
    lc.flush(0, lane);
#endif
    // tile_used = max over the tile's pixels of n_contrib: the backward never looks past it
    atomicMax(&s_used, last_contributor);
    __syncthreads();
    if (tid == 0) tile_used[tile] = s_used;
}

void launch_render_forward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                           const float* bg, float* out_color, uint8_t* qmask, const uint32_t* sort_err, hipStream_t s)
{
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    const int grid = ((T + 7) / 8) * 8;
    render_forward_kernel<<<grid, 256, 0, s>>>(W, H, gx, T, img.ranges, point_list, splat, bg, out_color, img.final_T,
                                               img.n_contrib, img.tile_used, qmask, sort_err);
}

// ---------------------------------------------------------------- backward
//
// Per (wave, Gaussian) nine gradient terms must be summed over the wave's 64 pixels: three colour terms and six moments
// of w = G * dL/dalpha. A plain butterfly costs 6 cross-lane adds per value (54 per Gaussian). Two ideas cut that to ~15:
//
// (1) SEPARABLE MOMENTS. Within a pixel row dy = mean.y - pixel.y is the same for the row's 8 lanes, so a lane only forms
//     w, w dx, w dx^2 (dx = mean.x - pixel.x, which the alpha evaluation already has) -- not w dy, w dx dy, w dy^2 --, the
//     sums over the row's 8 columns are taken on SIX values per Gaussian, and each row sum is expanded to the nine terms
//     with three multiplies (S_y = dy R0, S_xy = dy R1, S_yy = dy^2 R0) before the sums over the 8 rows. Same quantities as
//     the per-pixel products (dy is merely factored out of the row sums), three multiplies fewer per pair, and a third of
//     the values leave the first three reduction levels.
// (2) TRANSPOSING NETWORK. GROUP_G = 8 Gaussians are reduced TOGETHER: at every level a lane pair exchanges halves, each lane
//     keeps half of the values it held and adds its partner's copy of that half, so the live values per lane halve while the
//     lanes summed per value double:
//        columns (lane bits 2, 0, 1):  48 -> 24  row_half_mirror (bank-masked v_add_f32_dpp, no selects)
//                                      24 -> 12  quad_perm [1,0,3,2]
//                                      12 ->  6  quad_perm [2,3,0,1]     -> lane holds the 6 row sums of Gaussian beta(lane)
//        expand 6 -> 9 (three multiplies by dy, dy, dy^2 of Gaussian beta(lane) on this pixel row)
//        rows (lane bits 3, 4, 5):      8 ->  4  row_ror:8 (bank-masked)   + the ninth value by a plain add at each level
//                                       4 ->  2  v_permlane16_swap
//                                       2 ->  1  v_permlane32_swap        -> lane holds term m(lane) of Gaussian beta(lane)
//     ~124 vector instructions per 8 Gaussians (the 64-value network of round 1: ~140 per 7), and the 64 + 8 results sit in
//     different lanes, so two ds_add_f32 instructions deliver them.
constexpr int NPART = PARTIAL_FLOATS; // 9
constexpr int GROUP_G = 8;
constexpr int NCOL = 6;               // values per Gaussian that go through the column levels

template <int CTRL>
__device__ __forceinline__ float dpp_take(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int N, int CTRL>
__device__ __forceinline__ void transpose_reduce_step(float* v, bool hi)
{
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
        const float keep = hi ? v[k + N / 2] : v[k];
        const float send = hi ? v[k] : v[k + N / 2];
        v[k] = keep + dpp_take<CTRL>(send);
    }
}

// Steps 1 and 2 pair lanes that sit in different DPP banks (4-lane groups), so "which half do I keep" is
// expressed by the instruction's bank_mask instead of v_cndmask selects: per output two v_add_f32_dpp that each
// write half of the row's banks. Hand-placed because hipcc cannot emit bank-masked DPP adds from builtins (it
// selects with v_cndmask and pads every DPP read with s_nop: 237 instructions for the network vs ~150 here).
// The leading s_nop 1 covers the VALU-write -> DPP-read hazard for operands produced just before the block.
#define C3DGS_TR4(CTRL, LO, HI, o, a, b, base, half)                                                                      \
    asm volatile("s_nop 1\n\t"                                                                                            \
                 "v_add_f32_dpp %0, %4, %4 " CTRL " row_mask:0xf bank_mask:" LO "\n\t"                                    \
                 "v_add_f32_dpp %1, %5, %5 " CTRL " row_mask:0xf bank_mask:" LO "\n\t"                                    \
                 "v_add_f32_dpp %2, %6, %6 " CTRL " row_mask:0xf bank_mask:" LO "\n\t"                                    \
                 "v_add_f32_dpp %3, %7, %7 " CTRL " row_mask:0xf bank_mask:" LO "\n\t"                                    \
                 "v_add_f32_dpp %0, %8, %8 " CTRL " row_mask:0xf bank_mask:" HI "\n\t"                                    \
                 "v_add_f32_dpp %1, %9, %9 " CTRL " row_mask:0xf bank_mask:" HI "\n\t"                                    \
                 "v_add_f32_dpp %2, %10, %10 " CTRL " row_mask:0xf bank_mask:" HI "\n\t"                                  \
                 "v_add_f32_dpp %3, %11, %11 " CTRL " row_mask:0xf bank_mask:" HI                                         \
                 : "=&v"(o[base]), "=&v"(o[base + 1]), "=&v"(o[base + 2]), "=&v"(o[base + 3])                             \
                 : "v"(a[base]), "v"(a[base + 1]), "v"(a[base + 2]), "v"(a[base + 3]), "v"(b[base + half]),               \
                   "v"(b[base + half + 1]), "v"(b[base + half + 2]), "v"(b[base + half + 3]))

// column levels: v[48] = 8 Gaussians x 6 values -> u[0..5] = the sums over the lane's 8-pixel row of the 6 values of
// Gaussian slot beta(lane) = 4 * bit2 + 2 * bit0 + bit1.
// 48 -> 24 (in the group loop, interleaved with the Gaussians' bodies): row_half_mirror (i <-> 7 - i); lanes with bit 2 clear =
// banks 0,2 keep v[k], banks 1,3 keep v[k+24]. 24 -> 12 -> 6 here.
__device__ __forceinline__ void reduce_columns_24(float* u, int lane)
{
    transpose_reduce_step<24, 0xB1>(u, (lane & 1) != 0);   // quad_perm [1,0,3,2]: partners share a bank -> selects
    transpose_reduce_step<12, 0x4E>(u, (lane & 2) != 0);   // quad_perm [2,3,0,1]
}

// row levels: n[0..8] -> total = sum over the lane's column (8 rows) of n[m(lane)], m = 4 * bit3 + 2 * bit4 + bit5, and
// ninth = the same sum of n[8] (in every lane of the column)
__device__ __forceinline__ void reduce_rows_9(const float* n, float& total, float& ninth)
{
    float o[4];
    // 8 -> 4: row_ror:8 (lane i <-> i^8); lanes 0-7 = banks 0,1 keep n[k], lanes 8-15 = banks 2,3 keep n[k+4]
    C3DGS_TR4("row_ror:8", "0x3", "0xc", o, n, n, 0, 4);
    float t8;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(t8) : "v"(n[8]));
    // v_permlane16_swap: the odd 16-lane rows of the first operand trade places with the even rows of the second
    auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0]), __float_as_uint(o[2]), false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[1]), __float_as_uint(o[3]), false, false);
    auto s8 = __builtin_amdgcn_permlane16_swap(__float_as_uint(t8), __float_as_uint(t8), false, false);
    const float w0 = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
    const float w1 = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
    const float w8 = __uint_as_float(s8[0]) + __uint_as_float(s8[1]);
    auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w0), __float_as_uint(w1), false, false);
    auto s9 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w8), __float_as_uint(w8), false, false);
    total = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
    ninth = __uint_as_float(s9[0]) + __uint_as_float(s9[1]);
}

#ifndef C3DGS_BWD_WPE
#define C3DGS_BWD_WPE 5   // waves per SIMD the register allocator must reach (102 VGPRs); measured 3: 0.85 ms, 4: 0.745, 5: 0.707
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C3DGS_BWD_WPE, C3DGS_BWD_WPE)))
render_backward_kernel(int W, int H, int gx, int T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_used,
                       const uint32_t* __restrict__ point_list, const float4* __restrict__ splat,
                       const uint32_t* __restrict__ block_base, const float* __restrict__ bg, const float* __restrict__ final_Ts,
                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                       float* __restrict__ partials, uint8_t* __restrict__ touched, const uint8_t* __restrict__ qmask,
                       const uint32_t* __restrict__ tile_order, uint4* __restrict__ zero_span, size_t zero_n16)
{
    // every workgroup of the grid first clears its slice of `zero_span` (the scatter-added codebook gradients of the indexed
    // variant, which the NEXT kernel needs cleared): plain stores that cost this vector-issue-bound kernel nothing
    if (zero_n16) {
        const size_t per = (zero_n16 + gridDim.x - 1) / gridDim.x, z0 = (size_t)blockIdx.x * per, z1 = min(z0 + per, zero_n16);
        for (size_t i = z0 + threadIdx.x; i < z1; i += 256) zero_span[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // longest tiles first (tile_order: descending tile_used), so that the last workgroups to start are the short ones
    if ((int)blockIdx.x >= T) return;
    BT_STAMP(bt_begin)
#ifdef C3DGS_BWD_TIMING
    unsigned long long bt_stage = 0, bt_list = 0, bt_loop = 0, bt_flush = 0, bt_s0 = 0, bt_s1 = 0, bt_s2 = 0, bt_s3 = 0, bt_f0 = 0;
#endif
    const int tile = (int)tile_order[blockIdx.x];
    // staged entries as in the forward: 32-byte records {x, y, conic a, b | conic c, opacity, r, g}, blue and the instance's
    // backward slot in arrays of their own; the candidate lists hold record BYTE OFFSETS that feed the LDS reads directly
    // (see render_forward_kernel). Entry BATCH: sentinel, opacity 0.
    __shared__ float4 s_ab[BATCH + 1][2];
    __shared__ float s_c[BATCH + 1];
    __shared__ uint32_t s_slot[BATCH];
    // per wave: its candidates of HALF a batch (128 entries) at a time, 7 per 32-byte row (one row = one reduction group);
    // half batches keep the kernel's LDS at 31 KB = five workgroups per CU
    __shared__ uint32_t s_list[4][BATCH / 2 + GROUP_G];
    __shared__ unsigned long long s_mask[4][4];      // [quadrant][staging wave]
    __shared__ float s_part[2][BATCH][NPART];         // one plane per wave PAIR (see the flush below)
    constexpr uint32_t REC_BYTES = 32;
    const char* rec_base = reinterpret_cast<const char*>(&s_ab[0][0]);
    const char* blue_base = reinterpret_cast<const char*>(&s_c[0]);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix = (size_t)W * py + px, HW = (size_t)H * W;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int used = min(n, (int)tile_used[tile]);
    if (used <= 0) return;
    const int rounds = (used + BATCH - 1) / BATCH;
    if (tid == 0) { s_ab[BATCH][0] = make_float4(0, 0, 0, 0); s_ab[BATCH][1] = make_float4(0, 0, 0, 0); s_c[BATCH] = 0.f; }
    // every list word starts as the sentinel's offset: the group loop requests the row BEHIND its last group ahead of time
    for (int q = tid; q < 4 * (BATCH / 2 + GROUP_G); q += 256) (&s_list[0][0])[q] = (uint32_t)BATCH * REC_BYTES;

    const float T_final = inside ? final_Ts[pix] : 0.f;         // backward.cu:441-447
    float Tr = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) { dpx0 = dL_dpixels[pix]; dpx1 = dL_dpixels[HW + pix]; dpx2 = dL_dpixels[2 * HW + pix]; }
    // Running "what lies behind" term. The reference keeps the blended colour behind the current Gaussian as a
    // 3-channel recurrence (accum_rec / last_alpha / last_color, backward.cu:505-521) and adds the background term
    // separately (:531-534). Only its dot product with this pixel's dL/dC is ever used, so ONE scalar carries it:
    //   Sd_i = sum_{k behind i} alpha_k T_k (c_k . dL/dC)  +  T_final (bg . dL/dC),
    //   dL/dalpha_i = T_i (c_i . dL/dC) - Sd_i / (1 - alpha_i)          (same quantity, 19 instead of 36 VALU ops)
    float Sd = T_final * (bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2);
    // nothing behind the deepest last_contributor of this wave's 64 pixels can matter to this wave
    int wave_last = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, o));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);      // tell the compiler it is wave-uniform (scalar loop control)
    // where this lane's reduced values belong: Gaussian slot beta of the group, term my_m of the nine (see the network above)
    const int beta = ((lane >> 2) & 1) * 4 + (lane & 1) * 2 + ((lane >> 1) & 1);
    const int my_m = ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
#ifdef C3DGS_COUNT_LANES
    LaneCount lc;
#endif

    uint32_t next_id = 0u, next_qm = 0u;
    {
        const int p0 = used - 1 - tid;
        if (p0 >= 0) { next_id = point_list[range.x + p0]; next_qm = qmask[range.x + p0]; }
    }
    BT_STAMP(bt_pro)
    for (int r = 0; r < rounds; r++) {
        BT_STAMP(bt0)
        __syncthreads();                                         // previous flush has read s_part / s_slot
        BT_STAMP(btA)
        const int mypos = used - 1 - (r * BATCH + tid);          // back to front (backward.cu:466-479)
        uint32_t qm = 0u;
        if (mypos >= 0) {
            // the entry's Gaussian id and quadrant mask were requested one round ago (below): the record gather is the only
            // memory round trip left in front of this round's blending (measured with the "bwdtime" variant: the dependent chain
            // point list -> record was 31 % of a wave's lifetime, more than its group loop)
#if C3DGS_BWD_ABLATE & 16
            const uint32_t id = (range.x + (uint32_t)mypos) & 0x1fffffu;   // timing-only build: coalesced records, no dependence on the point list
#elif C3DGS_BWD_ABLATE & 2
            const uint32_t id = next_id & 4095u;                  // timing-only build: records from a cache-resident corner of the array
#else
            const uint32_t id = next_id;
#endif
            qm = next_qm;
            float4 a = splat[3 * (size_t)id], b = splat[3 * (size_t)id + 1];
            const float4 c = splat[3 * (size_t)id + 2];
            prescale_conic(a, b);
            const uint32_t off = __float_as_uint(c.y), lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
            const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff;
#if C3DGS_BWD_ABLATE & 18
            s_slot[tid] = range.x + (uint32_t)mypos;              // timing-only build: a slot that exists (the record is not this entry's)
            (void)off; (void)x0; (void)y0; (void)x1;
#else
            s_slot[tid] = block_base[id >> 8] + off + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
#endif
            s_ab[tid][0] = a; s_ab[tid][1] = b; s_c[tid] = c.x;
        }
        {   // next round's ids / masks: in flight behind this round's blending (two registers)
            const int npos = mypos - BATCH;
            if (npos >= 0) { next_id = point_list[range.x + npos]; next_qm = qmask[range.x + npos]; }   // qmask: written by the forward for every entry it staged
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long bm = __ballot((qm >> q) & 1u);
            if (lane == 0) s_mask[q][wave] = bm;
        }
        BT_STAMP(btB)
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
            for (int q = 0; q < NPART; q++) s_part[w][tid][q] = 0.f;
        BT_STAMP(btC)
        __syncthreads();
        BT_STAMP(bt1)
#ifdef C3DGS_BWD_TIMING
        bt_stage += bt1 - bt0;
        bt_s0 += btA - bt0; bt_s1 += btB - btA; bt_s2 += btC - btB; bt_s3 += bt1 - btC;
#endif

        const int cnt = min(BATCH, used - r * BATCH);
        const int pos0 = used - 1 - r * BATCH;                   // position of batch entry j is pos0 - j
        // a pixel blends entry j only if pos0 - j < last_contributor (backward.cu:486-488), i.e. record offset > thr
        const int thr = (pos0 - last_contributor) * (int)REC_BYTES;
        // This wave's candidates, compacted into LDS: 7 record offsets per 32-byte row (one row = one reduction group),
        // padded with the sentinel's. The group loop below then has NO data-dependent control flow: one broadcast row read,
        // seven fixed slots, one reduction. (A slot whose Gaussian turns out to touch no pixel of the wave -- 3.6 % on the
        // bench scene -- adds zeros; testing for it per Gaussian cost more than it saved.)
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
        BT_STAMP(bt2)
        int nw = 0;
        {
            const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const int c = half * 2 + cc;
                unsigned long long m = uniform_u64(s_mask[wave][c]);
                // entries with pos >= wave_last (j <= pos0 - wave_last) are behind every pixel of this wave
                const int jmin = pos0 - wave_last + 1 - c * 64;
                if (jmin >= 64) m = 0; else if (jmin > 0) m &= ~0ull << jmin;
                if ((m >> lane) & 1ull) s_list[wave][nw + (int)__popcll(m & lt)] = (uint32_t)(c * 64 + lane) * REC_BYTES;
                nw += (int)__popcll(m);
            }
            if (lane < GROUP_G) s_list[wave][nw + lane] = (uint32_t)BATCH * REC_BYTES;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        BT_STAMP(bt3)
        // Software pipeline, two levels: inside a group, record g + 1 is requested from LDS before Gaussian g is blended; across
        // groups, the NEXT group's list row and first record are requested behind this group's reduction (the list is padded, so
        // the row behind the last group exists; it is not used). Left to the compiler, a group started with two dependent LDS
        // round trips (row, then record) and every further record read sat directly in front of its first use.
        uint4 row0 = *reinterpret_cast<const uint4*>(&s_list[wave][0]);
        uint4 row1 = *reinterpret_cast<const uint4*>(&s_list[wave][4]);
        float4 a_n = *reinterpret_cast<const float4*>(rec_base + row0.x);
        float4 b_n = *reinterpret_cast<const float4*>(rec_base + row0.x + 16);
        float c_n = *reinterpret_cast<const float*>(blue_base + (row0.x >> 3));
        for (int k = 0; k < nw; k += GROUP_G) {
            const uint32_t* lrow = &s_list[wave][k];
            const uint32_t e[GROUP_G] = { row0.x, row0.y, row0.z, row0.w, row1.x, row1.y, row1.z, row1.w };
            float v[GROUP_G * NCOL];
            float u[GROUP_G * NCOL / 2];
            // batch entry of this lane's Gaussian slot (the list's padding entries are the sentinel, BATCH): requested here, used by
            // the row levels behind the loop (there it was an exposed chain of two LDS round trips)
            const uint32_t my_off = lrow[beta];
#pragma unroll
            for (int g = 0; g < GROUP_G; g++) {
                const float4 a = a_n, b = b_n;
                const float cblue = c_n;
                if (g + 1 < GROUP_G) {
                    a_n = *reinterpret_cast<const float4*>(rec_base + e[g + 1]);
                    b_n = *reinterpret_cast<const float4*>(rec_base + e[g + 1] + 16);
                    c_n = *reinterpret_cast<const float*>(blue_base + (e[g + 1] >> 3));
                }
                __builtin_amdgcn_sched_barrier(0);               // keep the reads above in front of this Gaussian's arithmetic
                float dx, dy, G, alpha;
                bool hit = gaussian_alpha(a.x, a.y, a.z, a.w, b.x, b.y, pxf, pyf, dx, dy, G, alpha);
                hit = hit && ((int)e[g] > thr);                  // backward.cu:486-488
#ifdef C3DGS_COUNT_LANES
                lc.pair(k + g < nw, __ballot(hit), 0ull);
#endif
                // branch-free: a pixel that does not blend this Gaussian runs the same instructions with
                // alpha = G = 0, which leaves T and Sd untouched and makes all six terms exactly 0
                const float a_eff = hit ? alpha : 0.f, G_eff = hit ? G : 0.f;
                // 1/(1-alpha) once, as a hardware reciprocal (1 ulp; 1-alpha is in [0.01, 1]; rcp(1) == 1)
                const float rinv = __builtin_amdgcn_rcpf(1.f - a_eff);
                Tr = Tr * rinv;                                  // transmittance in front of this Gaussian
                const float dchannel_dcolor = a_eff * Tr;
                const float cd = fmaf(cblue, dpx2, fmaf(b.w, dpx1, b.z * dpx0));
                v[g * NCOL + 0] = dchannel_dcolor * dpx0;
                v[g * NCOL + 1] = dchannel_dcolor * dpx1;
                v[g * NCOL + 2] = dchannel_dcolor * dpx2;
                const float dL_dalpha = fmaf(Tr, cd, -(rinv * Sd));
                Sd = fmaf(dchannel_dcolor, cd, Sd);
                // w = G * dL/dalpha and its first two x-moments about the mean; the y-moments follow from the row sums
                const float w = G_eff * dL_dalpha, wx = w * dx;
                v[g * NCOL + 3] = w;
                v[g * NCOL + 4] = wx;
                v[g * NCOL + 5] = wx * dx;
                // First column level (48 -> 24: value q pairs with value q + 24, i.e. Gaussian g with Gaussian g + 4) as soon as
                // both operands exist, four outputs at a time: the group then holds at most 24 + 6 live values instead of 48,
                // which leaves the register allocator room (96 registers = five waves per SIMD) to request the NEXT Gaussian's
                // record from LDS while this one is blended; with all 48 live every record read sat directly in front of its
                // first use (three LDS latencies per Gaussian and wave, exposed).
#if C3DGS_BWD_ABLATE & 8      // timing-only build: no reduction at all (the six values are merely kept alive)
#pragma unroll
                for (int q = 0; q < NCOL; q++) asm volatile("" : : "v"(v[g * NCOL + q]));
#else
                if (g == 4) { C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 0, 24); }
                if (g == 5) { C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 4, 24); C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 8, 24); }
                if (g == 6) { C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 12, 24); }
                if (g == 7) { C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 16, 24); C3DGS_TR4("row_half_mirror", "0x5", "0xa", u, v, v, 20, 24); }
#endif
            }
            row0 = *reinterpret_cast<const uint4*>(lrow + GROUP_G);          // next group's row (in bounds: the list is padded)
            row1 = *reinterpret_cast<const uint4*>(lrow + GROUP_G + 4);
#if !(C3DGS_BWD_ABLATE & 12)  // timing-only builds: bit 2 = without the two in-bank column levels, bit 3 = without any reduction
            reduce_columns_24(u, lane);
#endif
            a_n = *reinterpret_cast<const float4*>(rec_base + row0.x);          // ... and its first record, behind the row levels
            b_n = *reinterpret_cast<const float4*>(rec_base + row0.x + 16);
            c_n = *reinterpret_cast<const float*>(blue_base + (row0.x >> 3));
            // batch entry of this lane's Gaussian slot (the list's padding entries are the sentinel, BATCH) and this pixel
            // row's dy to that Gaussian: the same subtraction gaussian_alpha made for it
            const int myj = (int)(my_off >> 5);                  // offset / 32
            const float dyb = *reinterpret_cast<const float*>(rec_base + my_off + 4) - pyf;
            // row sums {c0, c1, c2, S0, Sx, Sxx} of Gaussian slot beta -> the nine terms {.., Sy, Sxy | Syy}
            float total, ninth;
#if !(C3DGS_BWD_ABLATE & 8)
            const float nine[NPART] = { u[0], u[1], u[2], u[3], u[4], u[5], dyb * u[3], dyb * u[4], (dyb * dyb) * u[3] };
#endif
#if C3DGS_BWD_ABLATE & 8
            total = v[0]; ninth = dyb;
#else
            reduce_rows_9(nine, total, ninth);
#endif
            // LDS float add into the plane this wave shares with ONE other wave: every (entry, term) receives at most
            // one add per wave, and a + b == b + a, so the result does not depend on which wave arrives first
            if (myj < BATCH) {
                __hip_atomic_fetch_add(&s_part[wave >> 1][myj][my_m], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane < 8) __hip_atomic_fetch_add(&s_part[wave >> 1][myj][8], ninth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#ifdef C3DGS_COUNT_LANES
        lc.end_list();
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the list is rebuilt for the second half
        __builtin_amdgcn_wave_barrier();
        BT_STAMP(bt4)
#ifdef C3DGS_BWD_TIMING
        bt_list += bt3 - bt2; bt_loop += bt4 - bt3;
#endif
        }
        BT_STAMP(bt5)
        __syncthreads();
        BT_STAMP(bt5b)
#ifdef C3DGS_BWD_TIMING
        bt_f0 += bt5b - bt5;
#endif
        if (tid < cnt) {
            const uint32_t slot = s_slot[tid];
            float* dst = partials + (size_t)slot * NPART;
            float t[NPART];
#pragma unroll
            for (int q = 0; q < NPART; q++) t[q] = s_part[0][tid][q] + s_part[1][tid][q];   // (w0 + w1) + (w2 + w3): fixed order
            // network order {c0, c1, c2, S0, Sx, Sxx, Sy, Sxy, Syy} -> what backward_preprocess.hip expects
            // (writing a slot cooperatively -- consecutive lanes on consecutive floats, 7 slots per store instruction -- measured
            // SLOWER: the flush went from 13 % to 19 % of a wave's lifetime, "bwdtime" variant)
#if !(C3DGS_BWD_ABLATE & 1)
            dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2];
            dst[3] = t[3]; dst[4] = t[4]; dst[5] = t[6];
            dst[6] = t[5]; dst[7] = t[7]; dst[8] = t[8];
#else
            if (t[0] == 123.456f) dst[0] = t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7] + t[8];     // timing-only build: no slot stores
#endif
            touched[slot] = 1;
        }
        BT_STAMP(bt6)
#ifdef C3DGS_BWD_TIMING
        bt_flush += bt6 - bt5;
#endif
    }
#ifdef C3DGS_COUNT_LANES
    lc.flush(8, lane);
#endif
#ifdef C3DGS_BWD_TIMING
    if (lane == 0) {
        atomicAdd(&g_lane_counters[8], bt_stage); atomicAdd(&g_lane_counters[9], bt_list); atomicAdd(&g_lane_counters[10], bt_loop);
        atomicAdd(&g_lane_counters[11], bt_flush); atomicAdd(&g_lane_counters[12], bt_pro - bt_begin); atomicAdd(&g_lane_counters[13], 1ull);
        atomicAdd(&g_lane_counters[14], (unsigned long long)__builtin_readcyclecounter() - bt_begin);
        atomicAdd(&g_lane_counters[0], bt_s0); atomicAdd(&g_lane_counters[1], bt_s1); atomicAdd(&g_lane_counters[2], bt_s2);
        atomicAdd(&g_lane_counters[3], bt_s3); atomicAdd(&g_lane_counters[4], bt_f0);
    }
#endif
}

int read_lane_counters(unsigned long long* out16, hipStream_t s)
{
    static const unsigned long long zeros[16] = { 0 };
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lane_counters), sizeof(zeros)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_counters), zeros, sizeof(zeros)) != hipSuccess;
}

// tile schedule + the backward's fills in one launch; zero_a / zero_b: 16-byte aligned spans of n16 x 16 bytes (or null / 0)
void launch_backward_prep(int W, int H, const ImgPtrs& img, uint32_t* tile_order, void* zero_a, size_t n16_a, void* zero_b,
                          size_t n16_b, hipStream_t s)
{
    const int T = (W > 0 && H > 0 && tile_order) ? tiles_x(W) * tiles_y(H) : 0;
    const size_t n16 = n16_a + n16_b;
    if (T <= 0 && n16 == 0) return;
    // fill workgroups: 16 KB each per sweep, at most four per CU
    const unsigned fill = n16 ? (unsigned)std::min<size_t>((n16 + 1023) / 1024, 1024) : 0u;
    backward_prep_kernel<<<1 + fill, 1024, 0, s>>>(T, img.tile_used, tile_order, (uint4*)zero_a, n16_a, (uint4*)zero_b, n16_b);
}

void launch_render_backward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                            const uint32_t* block_base, const float* bg, const float* dL_dpix, float* partials,
                            uint8_t* touched, const uint8_t* qmask, const uint32_t* tile_order, void* zero_span, size_t zero_n16,
                            hipStream_t s)
{
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    const int grid = ((T + 7) / 8) * 8;
    render_backward_kernel<<<grid, 256, 0, s>>>(W, H, gx, T, img.ranges, img.tile_used, point_list, splat, block_base, bg, img.final_T,
                                                img.n_contrib, dL_dpix, partials, touched, qmask, tile_order, (uint4*)zero_span,
                                                zero_span ? zero_n16 : 0);
}

} // namespace c3dgs
