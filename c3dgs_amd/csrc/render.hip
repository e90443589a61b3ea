// render.hip -- per-tile alpha blending, forward (K9) and backward (K10), for gfx950.
//
//   K9   reference FORWARD::renderCUDA   cuda_rasterizer/forward.cu:270-383
//   K10  reference BACKWARD::renderCUDA  cuda_rasterizer/backward.cu:399-557
//
// MI355X design (not a translation of the CUDA kernels):
//  * one 16x16 tile per 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel quadrant, so a
//    small Gaussian usually leaves whole waves idle and they skip it with one ballot;
//  * the tile's slice of the sorted point list is read coalesced, the 48-byte splat records are
//    gathered one cache line each into a double-buffered LDS batch (one barrier per batch), the
//    next batch's gather is issued before the current batch is blended;
//  * early-out is a per-wave ballot plus four LDS flags read after the batch barrier;
//  * workgroup ids are remapped so that each XCD (own L2) gets a contiguous band of tiles;
//  * backward: NO global atomics. The 9 per-Gaussian sums are reduced over the wave with DPP row
//    shifts + row broadcasts, over the 4 waves through LDS, and stored once per (Gaussian, tile)
//    instance into a slot array indexed like duplicate_with_keys' unsorted emission order; the
//    per-Gaussian kernel (backward_preprocess.hip) then sums a contiguous run of slots. Plain
//    stores run ~4-5x the chip-wide float-atomic rate on MI355X and the result is bitwise
//    reproducible (the reference's 9 atomics per pixel-Gaussian pair, backward.cu:523-554, are not).
#include "common.hpp"

namespace c3dgs {

constexpr int BATCH = 256;

// alpha of one Gaussian at one pixel; the SAME instruction sequence in forward and backward so both
// take identical skip decisions (explicit fma placement, independent of -ffp-contract).
// returns false when the reference `continue`s (forward.cu:344-354 / backward.cu:494-501).
__device__ __forceinline__ bool gaussian_alpha(float mx, float my, float ca, float cb, float cc, float op,
                                               float pxf, float pyf, float& dx, float& dy, float& G, float& alpha)
{
    dx = mx - pxf;
    dy = my - pyf;
    const float power = fmaf(-0.5f, fmaf(ca, dx * dx, cc * (dy * dy)), -(cb * (dx * dy)));
    if (power > 0.0f) return false;
    G = __expf(power);
    alpha = fminf(0.99f, op * G);
    return !(alpha < 1.0f / 255.0f);
}

// XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2) under the observed
// round-robin dispatch, so give every XCD one contiguous band of row-major tiles. Speed only.
__device__ __forceinline__ int tile_of_block(int b, int T)
{
    const int chunk = (T + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

__global__ void __launch_bounds__(256)
render_forward_kernel(int W, int H, int gx, int T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const float4* __restrict__ splat, const float* __restrict__ bg, float* __restrict__ out_color,
                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_used)
{
    const int tile = tile_of_block(blockIdx.x, T);
    if (tile >= T || (blockIdx.x >> 3) >= ((T + 7) >> 3)) return;
    __shared__ float4 s_a[2][BATCH];
    __shared__ float4 s_b[2][BATCH];
    __shared__ float s_c[2][BATCH];
    __shared__ int s_wdone[2][4];
    __shared__ uint32_t s_used;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    bool done = !inside;
    if (tid == 0) s_used = 0;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + BATCH - 1) / BATCH;

    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;

    float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
    if (tid < n) {
        const uint32_t id = point_list[range.x + tid];
        ra = splat[3 * (size_t)id]; rb = splat[3 * (size_t)id + 1]; rc = splat[3 * (size_t)id + 2];
    }
    for (int r = 0; r < rounds; r++) {
        const int buf = r & 1;
        s_a[buf][tid] = ra; s_b[buf][tid] = rb; s_c[buf][tid] = rc.x;
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
        const int cnt = min(BATCH, n - r * BATCH);
        const uint32_t base = (uint32_t)(r * BATCH);
        for (int j = 0; j < cnt; j++) {
            if (__all(done)) break;
            const float4 a = s_a[buf][j], b = s_b[buf][j];
            float dx, dy, G, alpha;
            const bool hit = gaussian_alpha(a.x, a.y, a.z, a.w, b.x, b.y, pxf, pyf, dx, dy, G, alpha);
            if (!__any(hit && !done)) continue;
            if (hit && !done) {
                const float test_T = Tr * (1.f - alpha);
                if (test_T < 0.0001f) {
                    done = true;                 // forward.cu:355-360: stop BEFORE blending this one
                } else {
                    const float w = alpha * Tr;
                    C0 = fmaf(b.z, w, C0); C1 = fmaf(b.w, w, C1); C2 = fmaf(s_c[buf][j], w, C2);
                    Tr = test_T;
                    last_contributor = base + (uint32_t)j + 1u;
                }
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px, HW = (size_t)H * W;
        final_T[pix] = Tr;
        n_contrib[pix] = last_contributor;
        out_color[pix] = fmaf(Tr, bg[0], C0);
        out_color[HW + pix] = fmaf(Tr, bg[1], C1);
        out_color[2 * HW + pix] = fmaf(Tr, bg[2], C2);
    }
    // tile_used = max over the tile's pixels of n_contrib: the backward never looks past it
    atomicMax(&s_used, last_contributor);
    __syncthreads();
    if (tid == 0) tile_used[tile] = s_used;
}

void launch_render_forward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                           const float* /*colors_precomp*/, const float* bg, float* out_color, hipStream_t s)
{
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    const int grid = ((T + 7) / 8) * 8;
    render_forward_kernel<<<grid, 256, 0, s>>>(W, H, gx, T, img.ranges, point_list, splat, bg, out_color, img.final_T,
                                               img.n_contrib, img.tile_used);
}

// ---------------------------------------------------------------- backward

// wave64 sum on gfx950 via DPP: 4 row shifts (sum of each 16-lane row lands in its lane 15), then
// row_bcast15 / row_bcast31 fold the four rows; the total is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = dpp_add<0x111, 0xf>(v); // row_shr:1
    v = dpp_add<0x112, 0xf>(v); // row_shr:2
    v = dpp_add<0x114, 0xf>(v); // row_shr:4
    v = dpp_add<0x118, 0xf>(v); // row_shr:8
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 -> rows 2,3
    return v;
}

constexpr int NPART = PARTIAL_FLOATS; // 9

__global__ void __launch_bounds__(256)
render_backward_kernel(int W, int H, int gx, int T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_used,
                       const uint32_t* __restrict__ point_list, const float4* __restrict__ splat,
                       const float* __restrict__ bg, const float* __restrict__ final_Ts,
                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                       float* __restrict__ partials)
{
    const int tile = tile_of_block(blockIdx.x, T);
    if (tile >= T || (blockIdx.x >> 3) >= ((T + 7) >> 3)) return;
    __shared__ float4 s_a[BATCH];
    __shared__ float4 s_b[BATCH];
    __shared__ float s_c[BATCH];
    __shared__ uint32_t s_slot[BATCH];
    __shared__ float s_part[4][BATCH][NPART];

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

    const float T_final = inside ? final_Ts[pix] : 0.f;         // backward.cu:441-447
    float Tr = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) { dpx0 = dL_dpixels[pix]; dpx1 = dL_dpixels[HW + pix]; dpx2 = dL_dpixels[2 * HW + pix]; }
    const float bg_dot = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;       // backward.cu:460-461

    for (int r = 0; r < rounds; r++) {
        __syncthreads();                                         // previous flush has read s_part / s_slot
        const int mypos = used - 1 - (r * BATCH + tid);          // back to front (backward.cu:466-479)
        if (mypos >= 0) {
            const uint32_t id = point_list[range.x + mypos];
            const float4 a = splat[3 * (size_t)id], b = splat[3 * (size_t)id + 1], c = splat[3 * (size_t)id + 2];
            s_a[tid] = a; s_b[tid] = b; s_c[tid] = c.x;
            const uint32_t off = __float_as_uint(c.y), lo = __float_as_uint(c.z), hi = __float_as_uint(c.w);
            const int x0 = lo & 0xffff, y0 = lo >> 16, x1 = hi & 0xffff;
            s_slot[tid] = off + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
        }
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int q = 0; q < NPART; q++) s_part[w][tid][q] = 0.f;
        __syncthreads();

        const int cnt = min(BATCH, used - r * BATCH);
        for (int j = 0; j < cnt; j++) {
            const int pos = used - 1 - (r * BATCH + j);          // 0-based position in the tile's list
            const float4 a = s_a[j], b = s_b[j];
            float dx, dy, G, alpha;
            bool hit = gaussian_alpha(a.x, a.y, a.z, a.w, b.x, b.y, pxf, pyf, dx, dy, G, alpha);
            hit = hit && (pos < last_contributor);               // backward.cu:486-488
            if (!__any(hit)) continue;                           // wave-uniform skip
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
            if (hit) {
                Tr = Tr / (1.f - alpha);
                const float dchannel_dcolor = alpha * Tr;
                const float c0 = b.z, c1 = b.w, c2 = s_c[j];
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2;
                float dL_dalpha = (c0 - acc0) * dpx0 + (c1 - acc1) * dpx1 + (c2 - acc2) * dpx2;
                v0 = dchannel_dcolor * dpx0; v1 = dchannel_dcolor * dpx1; v2 = dchannel_dcolor * dpx2;
                dL_dalpha *= Tr;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot; // backward.cu:531-534
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                v3 = dL_dG * dG_ddelx * ddelx_dx;
                v4 = dL_dG * dG_ddely * ddely_dy;
                v5 = -0.5f * gdx * dx * dL_dG;
                v6 = -0.5f * gdx * dy * dL_dG;
                v7 = -0.5f * gdy * dy * dL_dG;
                v8 = G * dL_dalpha;
            }
            v0 = wave_sum_lane63(v0); v1 = wave_sum_lane63(v1); v2 = wave_sum_lane63(v2);
            v3 = wave_sum_lane63(v3); v4 = wave_sum_lane63(v4); v5 = wave_sum_lane63(v5);
            v6 = wave_sum_lane63(v6); v7 = wave_sum_lane63(v7); v8 = wave_sum_lane63(v8);
            if (lane == 63) {
                float* d = s_part[wave][j];
                d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3; d[4] = v4; d[5] = v5; d[6] = v6; d[7] = v7; d[8] = v8;
            }
        }
        __syncthreads();
        if (tid < cnt) {
            float* dst = partials + (size_t)s_slot[tid] * NPART;
#pragma unroll
            for (int q = 0; q < NPART; q++)
                dst[q] = (s_part[0][tid][q] + s_part[1][tid][q]) + (s_part[2][tid][q] + s_part[3][tid][q]);
        }
    }
}

void launch_render_backward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                            const float* /*colors_precomp*/, const float* bg, const float* dL_dpix, float* partials,
                            hipStream_t s)
{
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    const int grid = ((T + 7) / 8) * 8;
    render_backward_kernel<<<grid, 256, 0, s>>>(W, H, gx, T, img.ranges, img.tile_used, point_list, splat, bg, img.final_T,
                                                img.n_contrib, dL_dpix, partials);
}

} // namespace c3dgs
