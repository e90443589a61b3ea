// preprocess.hip -- per-Gaussian forward stages for gfx950 (compiled with -ffp-contract=off).
//
//   K1  mark_visible        reference rasterizer_impl.cu:54-66 + auxiliary.h:139-166
//   K2  preprocess          reference forward.cu:163-265   (K2i: forward_indexed.cu:162-268)
//   K5  duplicate_with_keys reference rasterizer_impl.cu:70-111
//   K8  identify_ranges     reference rasterizer_impl.cu:116-138
//
// Design notes (MI355X): one thread per Gaussian, 256-thread blocks (4 waves). The stage is
// HBM-bound on the SH read (192 B of ~236 B per Gaussian at degree 3); culled Gaussians return
// before touching SH. Everything the blend kernels need per Gaussian is packed into ONE 48-byte
// record (3 x float4) so the per-tile gathers in render.hip touch a single cache line:
//   rec[0] = {mean2D.x, mean2D.y, conic.a, conic.b}
//   rec[1] = {conic.c, opacity, r, g}
//   rec[2] = {b, bits(first backward partial-sum slot, id order), bits(x0|y0<<16), bits(x1|y1<<16)}
// The last three words let the backward find the instance slot of (Gaussian, tile) without any
// extra gather (see render.hip).
#include "common.hpp"
#include "gsmath.hpp"

namespace c3dgs {

// four Gaussians per thread: 48 bytes of positions as three 16-byte loads, four flags as one 4-byte store (a thread per
// Gaussian with three 4-byte loads and a 1-byte store: 12-14 us for P = 3M; the pointers are 16- / 4-byte aligned when
// they come from torch allocations, anything else takes the scalar tail)
// POSE: `view` is the camera's 7-element pose (qx, qy, qz, qw, tx, ty, tz) instead of the 4x4 matrix; every thread forms the
// matrix entries the test reads (column 2 of the transposed view = row 2 of world->camera) with the fp32 operations of
// camera_from_pose_kernel below, in the same order (this file is built with -ffp-contract=off): same bits as the matrix path
// without that single-thread launch in front (5 us of an idle GPU per call).
template <bool POSE>
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view_or_pose, uint8_t* __restrict__ present, int vec)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = 4 * j;
    if (i0 >= P) return;
    float vm[16];
    const float* view = view_or_pose;
    if (POSE) {
        const float x = view_or_pose[0], y = view_or_pose[1], z = view_or_pose[2], w = view_or_pose[3], tz = view_or_pose[6];
        const float d2 = y * y + z * z + x * x;
#pragma unroll
        for (int q = 0; q < 16; q++) vm[q] = 0.f;
        vm[2] = 2.0f * (x * z - w * y); vm[6] = 2.0f * (y * z + w * x); vm[10] = 1.0f + 2.0f * (z * z - d2); vm[14] = tz;
        view = vm;
    }
    if (vec && i0 + 4 <= P) {
        const float4* src = reinterpret_cast<const float4*>(means3D) + 3 * (size_t)j;
        const float4 a = src[0], b = src[1], c = src[2];
        const f3 p0 = { a.x, a.y, a.z }, p1 = { a.w, b.x, b.y }, p2 = { b.z, b.w, c.x }, p3 = { c.y, c.z, c.w };
        const uint32_t v0 = !(xform4x3(p0, view).z <= 0.01f), v1 = !(xform4x3(p1, view).z <= 0.01f);
        const uint32_t v2 = !(xform4x3(p2, view).z <= 0.01f), v3 = !(xform4x3(p3, view).z <= 0.01f);
        reinterpret_cast<uint32_t*>(present)[j] = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
        return;
    }
    for (int i = i0; i < min(i0 + 4, P); i++) {
        f3 p = { means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2] };
        f3 pv = xform4x3(p, view);
        present[i] = !(pv.z <= 0.01f) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256)
pack_codebook_kernel(int GS, const float* __restrict__ scales, const float* __restrict__ rotations, float4* __restrict__ gtab)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= GS) return;
    gtab[2 * (size_t)g] = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)g);
    gtab[2 * (size_t)g + 1] = make_float4(scales[3 * (size_t)g], scales[3 * (size_t)g + 1], scales[3 * (size_t)g + 2], 0.f);
}

void launch_pack_codebook(const c3dgs_raster_params& p, float4* gtab, hipStream_t s)
{
    if (p.GS <= 0) return;
    pack_codebook_kernel<<<(p.GS + 255) / 256, 256, 0, s>>>(p.GS, p.scales, p.rotations, gtab);
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s)
{
    if (P <= 0) return;
    const int vec = ((reinterpret_cast<uintptr_t>(means3D) & 15u) == 0 && (reinterpret_cast<uintptr_t>(present) & 3u) == 0) ? 1 : 0;
    mark_visible_kernel<false><<<((P + 3) / 4 + 255) / 256, 256, 0, s>>>(P, means3D, view, present, vec);
}

void launch_mark_visible_pose(int P, const float* means3D, const float* pose, uint8_t* present, hipStream_t s)
{
    if (P <= 0) return;
    const int vec = ((reinterpret_cast<uintptr_t>(means3D) & 15u) == 0 && (reinterpret_cast<uintptr_t>(present) & 3u) == 0) ? 1 : 0;
    mark_visible_kernel<true><<<((P + 3) / 4 + 255) / 256, 256, 0, s>>>(P, means3D, pose, present, vec);
}

// ---- camera set-up ON THE DEVICE, from the pose's live values: the host part of the reference's autograd wrappers
// (DGR-NC __init__.py:32-40 quat_to_mat, :152-172 `extrinsic @ getProjectionMatrix(...)`, `extrinsic.inverse()[3, :3]`).
// The reference evaluates quat_to_mat with fp32 tensor arithmetic on the pose's elements; the same fp32 operations in the
// same order run here (this file is built with -ffp-contract=off), stream-ordered behind whatever last wrote the pose --
// so an optimiser that updates the pose through raw pointers or `.data` can never be rendered with a stale matrix, and
// no device->host read is needed. view is stored transposed like the reference's (m[0], m[4], m[8], m[12] = row 0).
__global__ void camera_from_pose_kernel(const float* __restrict__ pose, float inv_tan_x, float inv_tan_y, float pa, float pb,
                                        float* __restrict__ view, float* __restrict__ proj, float* __restrict__ campos)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float x = pose[0], y = pose[1], z = pose[2], w = pose[3], tx = pose[4], ty = pose[5], tz = pose[6];
    const float d2 = y * y + z * z + x * x;
    float M[4][4];                                             // world -> camera, row major
    M[0][0] = 1.0f + 2.0f * (x * x - d2); M[0][1] = 2.0f * (x * y - w * z);        M[0][2] = 2.0f * (x * z + w * y);        M[0][3] = tx;
    M[1][0] = 2.0f * (x * y + w * z);        M[1][1] = 1.0f + 2.0f * (y * y - d2); M[1][2] = 2.0f * (y * z - w * x);        M[1][3] = ty;
    M[2][0] = 2.0f * (x * z - w * y);        M[2][1] = 2.0f * (y * z + w * x);        M[2][2] = 1.0f + 2.0f * (z * z - d2); M[2][3] = tz;
    M[3][0] = 0.f; M[3][1] = 0.f; M[3][2] = 0.f; M[3][3] = 1.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) view[i * 4 + j] = M[j][i];
        // row i of view @ P^T with P^T = [[ix,0,0,0],[0,iy,0,0],[0,0,a,1],[0,0,b,0]] (getProjectionMatrix, znear .01 zfar 100)
        proj[i * 4 + 0] = M[0][i] * inv_tan_x;
        proj[i * 4 + 1] = M[1][i] * inv_tan_y;
        proj[i * 4 + 2] = M[2][i] * pa + M[3][i] * pb;
        proj[i * 4 + 3] = M[2][i];
    }
    // inverse(view)[3, :3] = -R^-1 t (camera centre); adjugate in double, R is not assumed orthonormal (the pose's quaternion
    // is not normalised by the reference either)
    const double a = M[0][0], b = M[0][1], c = M[0][2], d = M[1][0], e = M[1][1], f = M[1][2], g = M[2][0], h = M[2][1], k = M[2][2];
    const double A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double inv[3][3] = { { A / det, -(b * k - c * h) / det, (b * f - c * e) / det },
                               { B / det, (a * k - c * g) / det, -(a * f - c * d) / det },
                               { C / det, -(a * h - b * g) / det, (a * e - b * d) / det } };
#pragma unroll
    for (int r = 0; r < 3; r++) campos[r] = (float)-(inv[r][0] * (double)tx + inv[r][1] * (double)ty + inv[r][2] * (double)tz);
}

void launch_camera_from_pose(const float* pose, float inv_tan_x, float inv_tan_y, float* view, float* proj, float* campos, hipStream_t s)
{
    // torch.Tensor([...]) rounds getProjectionMatrix's double entries to fp32 (__init__.py:25-30)
    const float pa = (float)(1.0 * 100.0 / (100.0 - 0.01)), pb = (float)(-(100.0 * 0.01) / (100.0 - 0.01));
    camera_from_pose_kernel<<<1, 64, 0, s>>>(pose, inv_tan_x, inv_tan_y, pa, pb, view, proj, campos);
}

// ---- SH -> RGB, reference forward.cu:20-79. `c` holds the (DEG+1)^2 * 3 coefficients of this Gaussian.
template <int DEG>
__device__ __forceinline__ void sh_to_rgb(const float* c, float x, float y, float z, float out[3])
{
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float res = SH_C0 * c[0 * 3 + ch];
        if (DEG > 0) {
            res = res - SH_C1 * y * c[1 * 3 + ch] + SH_C1 * z * c[2 * 3 + ch] - SH_C1 * x * c[3 * 3 + ch];
            if (DEG > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2_0 * xy * c[4 * 3 + ch] + SH_C2_1 * yz * c[5 * 3 + ch] +
                      SH_C2_2 * (2.0f * zz - xx - yy) * c[6 * 3 + ch] + SH_C2_3 * xz * c[7 * 3 + ch] +
                      SH_C2_4 * (xx - yy) * c[8 * 3 + ch];
                if (DEG > 2) {
                    res = res + SH_C3_0 * y * (3.0f * xx - yy) * c[9 * 3 + ch] + SH_C3_1 * xy * z * c[10 * 3 + ch] +
                          SH_C3_2 * y * (4.0f * zz - xx - yy) * c[11 * 3 + ch] +
                          SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * c[12 * 3 + ch] +
                          SH_C3_4 * x * (4.0f * zz - xx - yy) * c[13 * 3 + ch] + SH_C3_5 * z * (xx - yy) * c[14 * 3 + ch] +
                          SH_C3_6 * x * (xx - 3.0f * yy) * c[15 * 3 + ch];
                }
            }
        }
        out[ch] = res + 0.5f;
    }
}

struct PreArgs {
    int P, M, W, H, gx, gy;
    const float* means3D; const float* sh; const float* colors_precomp; const float* opacities;
    const float* scales; const float* scale_factors; const float* rotations; const float* cov3D_precomp;
    const int64_t* sh_indices; const int64_t* g_indices;
    const float* view; const float* proj; const float* campos;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int prefiltered, clamp_color;
    int32_t* radii; float4* splat; uint16_t* rects; uint8_t* clamped;
    uint32_t* depth_keys;
    uint32_t* inst_offset; uint32_t* block_total;   // two-level id-order scan
    uint2* ranges; int T;                            // tile ranges, cleared here for identify_ranges (K7)
    const float4* gtab;                              // packed scale / rotation codebook (indexed variant)
    uint4* zero_span; size_t zero_n16;               // the depth sort's control words, cleared here (a slice per workgroup)
};

// The id-order scan of tiles_touched (where a Gaussian's backward partial-sum slots live; its total is num_rendered) is
// split in two levels so that no separate P-sized scan pass and no scattered "stamp" pass are needed: this kernel scans
// inside each 256-Gaussian workgroup (inst_offset = inclusive offset WITHIN the workgroup, record word 9 = exclusive one)
// and emits the workgroup totals; scan_blocks_kernel turns the ~P/256 totals into block_base[]. Consumers add
// block_base[id >> 8].
template <int DEG>
__global__ void __launch_bounds__(256) preprocess_kernel(const PreArgs a)
{
    __shared__ uint32_t s_wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = blockIdx.x * 256 + t;
    const bool in_range = i < a.P;
    for (int q = i; q < a.T; q += gridDim.x * 256) a.ranges[q] = make_uint2(0u, 0u);   // replaces a memset launch (rasterizer_impl.cu:308)
    if (a.zero_n16) {                                                                   // ... and the depth sort's own fill launch
        const size_t per = (a.zero_n16 + gridDim.x - 1) / gridDim.x, z0 = (size_t)blockIdx.x * per, z1 = min(z0 + per, a.zero_n16);
        for (size_t q = z0 + t; q < z1; q += 256) a.zero_span[q] = make_uint4(0u, 0u, 0u, 0u);
    }

    int32_t out_radius = 0;
    uint32_t out_tiles = 0;
    uint32_t out_key = 0xffffffffu;            // culled Gaussians sort behind every visible one
    bool alive = false;
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0;
    float rgb2 = 0.f;
    uint32_t rect_lo = 0, rect_hi = 0;
    if (in_range) do {
        const f3 p = { a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2] };
        const f3 p_view = xform4x3(p, a.view);
        if (!a.prefiltered && p_view.z <= 0.01f) break;           // auxiliary.h:156

        const float4 p_hom = xform4x4(p, a.proj);
        const float p_w = 1.0f / (p_hom.w + 0.0000001f);
        const float projx = p_hom.x * p_w, projy = p_hom.y * p_w;

        float cov3D[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int q = 0; q < 6; q++) cov3D[q] = a.cov3D_precomp[6 * (size_t)i + q];
        } else if (a.g_indices) {                                  // forward_indexed.cu:223
            const size_t g = (size_t)a.g_indices[i];
            const float4 rot = a.gtab[2 * g], sc = a.gtab[2 * g + 1];     // one 32-byte row: {rotation | scale, 0}
            cov3d_from_scale_rot(sc.x, sc.y, sc.z, a.scale_factors[i] * a.scale_modifier, rot, cov3D);
        } else {                                                   // forward.cu:220
            const float4 rot = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)i);
            cov3d_from_scale_rot(a.scales[3 * (size_t)i], a.scales[3 * (size_t)i + 1], a.scales[3 * (size_t)i + 2],
                                 a.scale_modifier, rot, cov3D);
        }
        const Cov2D cv = cov2d(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.view);

        const float det = (cv.a * cv.c - cv.b * cv.b);             // forward.cu:228-232
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float conic_a = cv.c * det_inv, conic_b = -cv.b * det_inv, conic_c = cv.a * det_inv;

        const float mid = 0.5f * (cv.a + cv.c);                    // forward.cu:238-246
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pix = ndc2pix(projx, a.W), piy = ndc2pix(projy, a.H);
        int x0, y0, x1, y1;
        get_rect(pix, piy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;

        float rgb[3];
        uint8_t clamp_bits = 0;
        if (a.colors_precomp) {
            rgb[0] = a.colors_precomp[3 * (size_t)i];
            rgb[1] = a.colors_precomp[3 * (size_t)i + 1];
            rgb[2] = a.colors_precomp[3 * (size_t)i + 2];
        } else {                                                   // forward.cu:20-79
            const size_t row = a.sh_indices ? (size_t)a.sh_indices[i] : (size_t)i;
            const float* shp = a.sh + row * (size_t)a.M * 3;
            constexpr int NC = (DEG + 1) * (DEG + 1) * 3;
            float c[NC];
            if ((a.M * 3) % 4 == 0 && NC % 4 == 0) {               // 16-byte aligned rows: M = 4, 16
#pragma unroll
                for (int q = 0; q < NC / 4; q++) {
                    const float4 v = reinterpret_cast<const float4*>(shp)[q];
                    c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NC; q++) c[q] = shp[q];
            }
            f3 dir = { p.x - a.campos[0], p.y - a.campos[1], p.z - a.campos[2] };
            const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
            dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
            sh_to_rgb<DEG>(c, dir.x, dir.y, dir.z, rgb);
            if (a.clamp_color) {                                   // forward.cu:65-72
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    if (rgb[ch] < 0) clamp_bits |= (uint8_t)(1u << ch);
                    rgb[ch] = fmaxf(rgb[ch], 0.0f);
                }
            }
        }

        alive = true;
        out_radius = (int32_t)my_radius;
        out_tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        out_key = __float_as_uint(p_view.z);
        a.clamped[i] = clamp_bits;
        uint16_t* rc = a.rects + 4 * (size_t)i;
        rc[0] = (uint16_t)x0; rc[1] = (uint16_t)y0; rc[2] = (uint16_t)x1; rc[3] = (uint16_t)y1;
        rec0 = make_float4(pix, piy, conic_a, conic_b);
        rec1 = make_float4(conic_c, a.opacities[i], rgb[0], rgb[1]);
        rgb2 = rgb[2];
        rect_lo = (uint32_t)x0 | ((uint32_t)y0 << 16);
        rect_hi = (uint32_t)x1 | ((uint32_t)y1 << 16);
    } while (false);

    // ---- id-order scan of tiles_touched within the workgroup
    uint32_t incl = out_tiles;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += s_wsum[w];
    const uint32_t my_incl = woff + incl;
    if (t == 255) a.block_total[blockIdx.x] = my_incl;
    if (!in_range) return;
    if (!alive) *reinterpret_cast<uint2*>(a.rects + 4 * (size_t)i) = make_uint2(0u, 0u);   // zero tiles for the pair emission
    if (alive) {
        float4* rec = a.splat + 3 * (size_t)i;
        rec[0] = rec0;
        rec[1] = rec1;
        // word 9: first backward partial-sum slot of this Gaussian, relative to block_base[i >> 8]
        rec[2] = make_float4(rgb2, __uint_as_float(my_incl - out_tiles), __uint_as_float(rect_lo), __uint_as_float(rect_hi));
    }
    a.inst_offset[i] = my_incl;
    a.radii[i] = out_radius;
    a.depth_keys[i] = out_key;
}

// exclusive scan of the workgroup totals, in place: base[b] = instances of all Gaussians before workgroup b;
// base[nb] = num_rendered. One workgroup; nb = P/256 is a few thousand to a few ten-thousand.
// `sort_err` (optional): the device's sticky sort time-out word, copied behind the total so that the forward's single
// device->host read of num_rendered brings it along (radix_sort.hip).
// `host_out` (optional): three words of MAPPED, coherent host memory {total, sort error word, host_seq}: the forward's one
// device->host read without a copy command -- the host polls the third word for `host_seq` (c_abi.hip). A copy command behind
// this kernel cost a 4 us launch of its own plus a ~6 us bubble on the stream.
__global__ void __launch_bounds__(1024) scan_blocks_kernel(int nb, uint32_t* __restrict__ base, const uint32_t* __restrict__ sort_err,
                                                           uint32_t* __restrict__ host_out, uint32_t host_seq)
{
    // One workgroup; a thread owns 16 CONSECUTIVE totals (four independent 16-byte loads), so 16384 totals cost one
    // memory round trip and one barrier. (One total per thread and a round trip + barrier per 1024 totals took 12 us for the
    // 11.7k totals of P = 3M: pure latency.) `base` is 256-byte aligned with room up to the next multiple of 16 entries + 2.
    __shared__ uint32_t s_w[2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t carry = 0;                                   // every thread tracks the running total itself
    int buf = 0;
    for (int c0 = 0; c0 < nb; c0 += 16384, buf ^= 1) {
        const int i0 = c0 + t * 16;
        uint32_t v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (i0 + 4 * q + 3 < nb) x = reinterpret_cast<const uint4*>(base + i0)[q];
            else {
                if (i0 + 4 * q < nb) x.x = base[i0 + 4 * q];
                if (i0 + 4 * q + 1 < nb) x.y = base[i0 + 4 * q + 1];
                if (i0 + 4 * q + 2 < nb) x.z = base[i0 + 4 * q + 2];
            }
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) mine += v[q];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) s_w[buf][wave] = incl;
        __syncthreads();                                  // s_w is double-buffered: one barrier per sweep
        uint32_t off = carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const uint32_t x = s_w[buf][w]; if (w < wave) off += x; all += x; }
        uint32_t run = off + incl - mine;                 // exclusive base of this thread's first total
#pragma unroll
        for (int q = 0; q < 16; q++) { const uint32_t x = v[q]; v[q] = run; run += x; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (i0 + 4 * q + 3 < nb) reinterpret_cast<uint4*>(base + i0)[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            else {
                if (i0 + 4 * q < nb) base[i0 + 4 * q] = v[4 * q];
                if (i0 + 4 * q + 1 < nb) base[i0 + 4 * q + 1] = v[4 * q + 1];
                if (i0 + 4 * q + 2 < nb) base[i0 + 4 * q + 2] = v[4 * q + 2];
            }
        }
        carry += all;
    }
    __syncthreads();                                      // the last sweep's stores precede the two words behind them
    if (t == 0) {
        base[nb] = carry;
        const uint32_t err = sort_err ? *sort_err : 0u;
        if (sort_err) base[nb + 1] = err;
        if (host_out) {
            __hip_atomic_store(host_out, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_out + 1, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_out + 2, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // publishes the two words above
        }
    }
}

void launch_preprocess(const c3dgs_raster_params& p, const GeomPtrs& g, int32_t* radii, uint2* ranges, const uint32_t* sort_err,
                       void* zero_span, size_t zero_n16, uint32_t* host_out, uint32_t host_seq, hipStream_t s)
{
    if (p.P <= 0) return;
    PreArgs a;
    a.P = p.P; a.M = p.M; a.W = p.W; a.H = p.H; a.gx = tiles_x(p.W); a.gy = tiles_y(p.H);
    a.means3D = p.means3D; a.sh = p.sh; a.colors_precomp = p.colors_precomp; a.opacities = p.opacities;
    a.scales = p.scales; a.scale_factors = p.scale_factors; a.rotations = p.rotations; a.cov3D_precomp = p.cov3D_precomp;
    a.sh_indices = p.sh_indices; a.g_indices = p.g_indices;
    a.view = p.viewmatrix; a.proj = p.projmatrix; a.campos = p.campos;
    a.tan_fovx = p.tan_fovx; a.tan_fovy = p.tan_fovy;
    a.focal_y = p.H / (2.0f * p.tan_fovy);                       // rasterizer_impl.cu:219-220
    a.focal_x = p.W / (2.0f * p.tan_fovx);
    a.scale_modifier = p.scale_modifier; a.prefiltered = p.prefiltered; a.clamp_color = p.clamp_color;
    a.radii = radii; a.splat = g.splat; a.rects = g.rects;
    a.clamped = g.clamped; a.depth_keys = g.depth_keys;
    a.inst_offset = g.inst_offset;
    a.block_total = g.block_base;     // totals in, exclusive bases out (scan_blocks_kernel)
    a.ranges = ranges; a.T = a.gx * a.gy;
    a.gtab = g.gtab;
    a.zero_span = (uint4*)zero_span; a.zero_n16 = zero_span ? zero_n16 : 0;
    const dim3 grid((p.P + 255) / 256), block(256);
    const int deg = p.colors_precomp ? 0 : p.D;
    switch (deg) {
        case 0: preprocess_kernel<0><<<grid, block, 0, s>>>(a); break;
        case 1: preprocess_kernel<1><<<grid, block, 0, s>>>(a); break;
        case 2: preprocess_kernel<2><<<grid, block, 0, s>>>(a); break;
        default: preprocess_kernel<3><<<grid, block, 0, s>>>(a); break;
    }
    scan_blocks_kernel<<<1, 1024, 0, s>>>((int)grid.x, g.block_base, sort_err, host_out, host_seq);
}

// ---- K5: one (tile, Gaussian) pair per Gaussian x tile, reference rasterizer_impl.cu:70-111, walked in
// (depth, id) order (binning.hip explains why). Thread k handles the k-th nearest Gaussian.
// Load-balanced expansion: a workgroup owns 256 consecutive Gaussians of the depth order, whose pairs form ONE
// contiguous output range; its threads walk that range (coalesced 2-byte / 4-byte stores) and find the owning
// Gaussian of each output by binary search over the 256 scan values in LDS. A thread-per-Gaussian loop (the
// reference's shape) strands 63 lanes behind the one big splat and scatters its stores.
// tiles covered by a packed tile rectangle {x0 | y0 << 16, x1 | y1 << 16} (zero for culled Gaussians)
__device__ __forceinline__ uint32_t rect_tiles(const uint2 rc)
{
    return (uint32_t)(((int)(rc.y & 0xffff) - (int)(rc.x & 0xffff)) * ((int)(rc.y >> 16) - (int)(rc.x >> 16)));
}

__global__ void __launch_bounds__(256)
block_totals_kernel(int n, const uint2* __restrict__ rects_sorted, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = i < n ? rect_tiles(rects_sorted[i]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

template <class K>   // tile key type: uint16_t, or uint32_t above 65,536 tiles
__global__ void __launch_bounds__(256)
duplicate_with_keys_kernel(int P, const uint32_t* __restrict__ order, const uint2* __restrict__ rects_sorted,
                           const uint32_t* __restrict__ depth_base, K* __restrict__ keys, uint32_t* __restrict__ values,
                           int grid_x, const uint32_t* __restrict__ sort_err, uint4* __restrict__ zero_span, size_t zero_n16)
{
    // the tile sort's control words (look-back status, tickets) are cleared here, a slice per workgroup, instead of by a fill
    // launch in front of the sort -- BEFORE the early-out below: that sort runs either way and must not walk stale status words
    if (zero_n16) {
        const size_t per = (zero_n16 + gridDim.x - 1) / gridDim.x, z0 = (size_t)blockIdx.x * per, z1 = min(z0 + per, zero_n16);
        for (size_t q = z0 + threadIdx.x; q < z1; q += 256) zero_span[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    // a depth sort whose look-back gave up (radix_sort.hip) left positions of `order` / `rects_sorted` unwritten: their stale
    // contents would be emission offsets -> never dereference them (the image is poisoned by render_forward, the host reports)
    if (*sort_err) return;
    __shared__ uint32_t s_end[256];      // inclusive emission offset of each of the block's Gaussians
    __shared__ uint32_t s_id[256];
    __shared__ uint2 s_rect[256];
    __shared__ uint32_t s_inv[256];      // ceil(2^32 / rectangle width): one division per Gaussian instead of one per instance
    __shared__ uint32_t s_wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int k0 = blockIdx.x * 256, k = k0 + t;
    const uint32_t out_begin = depth_base[blockIdx.x];
    uint32_t incl = 0;
    if (k < P) {                              // everything the emission needs arrives coalesced, in depth order
        const uint2 rc = rects_sorted[k];
        incl = rect_tiles(rc);
        s_id[t] = order[k];
        s_rect[t] = rc;
        const uint32_t w = (rc.y & 0xffff) - (rc.x & 0xffff);
        s_inv[t] = w > 1 ? 0xffffffffu / w + 1u : 0u;         // width 1: quotient = local (handled below); width 0: no instances
    }
    // depth-order scan inside the workgroup (the workgroup bases come from scan_blocks_kernel)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = out_begin;
    for (int w = 0; w < wave; w++) woff += s_wsum[w];
    s_end[t] = woff + incl;
    __syncthreads();
    const int nk = min(256, P - k0);
    const uint32_t out_end = s_end[nk - 1];
    // Owner of every output position WITHOUT a search: each Gaussian with instances drops its number at the position where its
    // run starts (`heads`), and a prefix maximum along the positions spreads it over the run (the numbers grow with the
    // position). A chunk is 1024 positions, four consecutive ones per lane, aligned to 4 so that the four 16-bit keys and the
    // four ids leave as one 8-byte and one 16-byte store. (A binary search per instance in the 256 run ends cost ~48 of this
    // kernel's ~70 instructions per instance: 0.050 -> 0.032 ms on the bench view.)
    constexpr int CH = 1024;
    __shared__ uint16_t s_head[CH];
    __shared__ uint32_t s_wmax[4];
    const uint32_t my_tiles = k < P ? rect_tiles(s_rect[t]) : 0u;
    const uint32_t my_begin = s_end[t] - my_tiles;
    uint32_t carry = 0;                                              // owner + 1 of the position in front of the chunk
    for (uint32_t c0 = out_begin & ~3u; c0 < out_end; c0 += CH) {
        reinterpret_cast<uint2*>(s_head)[t] = make_uint2(0u, 0u);
        __syncthreads();
        if (my_tiles && my_begin >= c0 && my_begin < c0 + CH) s_head[my_begin - c0] = (uint16_t)(t + 1);
        __syncthreads();
        const uint2 hv = reinterpret_cast<const uint2*>(s_head)[t];
        uint32_t m[4] = { hv.x & 0xffffu, hv.x >> 16, hv.y & 0xffffu, hv.y >> 16 };
        m[1] = max(m[0], m[1]); m[2] = max(m[1], m[2]); m[3] = max(m[2], m[3]);
        uint32_t scan = m[3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(scan, o); if (lane >= o) scan = max(scan, v); }
        uint32_t before = __shfl_up(scan, 1);
        if (lane == 0) before = 0;
        if (lane == 63) s_wmax[wave] = scan;
        __syncthreads();
        before = max(before, carry);
        for (int w = 0; w < wave; w++) before = max(before, s_wmax[w]);
        carry = max(max(carry, max(s_wmax[0], s_wmax[1])), max(s_wmax[2], s_wmax[3]));
        const uint32_t o0 = c0 + 4u * t;
        K kq[4];
        uint32_t vq[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t o = o0 + e;
            const uint32_t own = max(before, m[e]);                 // 0 only in front of out_begin
            kq[e] = 0; vq[e] = 0;
            if (o >= out_begin && o < out_end) {
                const int g = (int)own - 1;
                const uint32_t first = (g == 0) ? out_begin : s_end[g - 1];
                const uint2 rc = s_rect[g];
                const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff;
                const uint32_t local = o - first, w = (uint32_t)(x1 - x0), inv = s_inv[g];
                // local / w as a multiply-high by ceil(2^32 / w): exact while local * w < 2^32, and local < rectangle area <=
                // number of tiles (<= 2^16 with 16-bit tile keys; the C ABI admits at most 2^22 tiles), w <= 2^10 of them per row
                const uint32_t ry = inv ? __umulhi(local, inv) : local, rx = local - ry * w;   // emission order: y outer, x inner (:98-108)
                kq[e] = (K)((y0 + ry) * grid_x + (x0 + rx));
                vq[e] = s_id[g];
            }
        }
        if (o0 >= out_begin && o0 + 4 <= out_end) {
            if constexpr (sizeof(K) == 2)
                *reinterpret_cast<uint2*>(keys + o0) = make_uint2((uint32_t)kq[0] | ((uint32_t)kq[1] << 16), (uint32_t)kq[2] | ((uint32_t)kq[3] << 16));
            else
                *reinterpret_cast<uint4*>(keys + o0) = make_uint4((uint32_t)kq[0], (uint32_t)kq[1], (uint32_t)kq[2], (uint32_t)kq[3]);
            *reinterpret_cast<uint4*>(values + o0) = make_uint4(vq[0], vq[1], vq[2], vq[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (o0 + e >= out_begin && o0 + e < out_end) { keys[o0 + e] = kq[e]; values[o0 + e] = vq[e]; }
        }
        __syncthreads();                                             // s_head / s_wmax are rewritten by the next chunk
    }
}

// depth-order scan of tiles_touched, second level: totals of the 256-Gaussian groups of the depth order -> depth_base[]
void launch_depth_order_scan(int P, const GeomPtrs& g, hipStream_t s)
{
    if (P <= 0) return;
    const int nb = (P + 255) / 256;
    block_totals_kernel<<<nb, 256, 0, s>>>(P, g.sorted_offsets, g.depth_base);
    scan_blocks_kernel<<<1, 1024, 0, s>>>(nb, g.depth_base, nullptr, nullptr, 0u);
}

void launch_duplicate_with_keys(int P, const GeomPtrs& g, const BinPtrs& b, int grid_x, const uint32_t* sort_err,
                                void* zero_span, size_t zero_n16, hipStream_t s)
{
    if (P <= 0) return;
    if (!zero_span) zero_n16 = 0;
    if (b.key_bytes == 4)
        duplicate_with_keys_kernel<uint32_t><<<(P + 255) / 256, 256, 0, s>>>(P, g.depth_order, g.sorted_offsets, g.depth_base,
                                                                             (uint32_t*)b.keys_unsorted, b.values_unsorted, grid_x, sort_err, (uint4*)zero_span, zero_n16);
    else
        duplicate_with_keys_kernel<uint16_t><<<(P + 255) / 256, 256, 0, s>>>(P, g.depth_order, g.sorted_offsets, g.depth_base,
                                                                             (uint16_t*)b.keys_unsorted, b.values_unsorted, grid_x, sort_err, (uint4*)zero_span, zero_n16);
}

// ---- K8: reference rasterizer_impl.cu:116-138 (ranges pre-zeroed by the caller, :308)
template <class K>
__global__ void __launch_bounds__(256)
identify_ranges_kernel(int L, const K* __restrict__ keys, uint2* __restrict__ ranges, const uint32_t* __restrict__ sort_err)
{
    // after a sort time-out the "sorted" keys are stale memory: used as tile numbers they would index past `ranges`. The ranges
    // stay all-zero instead, so render_forward walks no list at all (and returns its NaN image)
    if (*sort_err) return;
    // eight sorted keys per thread (16-byte loads + the key in front of them): the buffer is 256-byte aligned and
    // padded, so the last thread's loads stay inside it
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = j * 8;
    if (i0 >= L) return;
    uint32_t cur8[8];
    if constexpr (sizeof(K) == 2) {
        const uint4 v = reinterpret_cast<const uint4*>(keys)[j];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int e = 0; e < 8; e++) cur8[e] = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
    } else {
        const uint4 v0 = reinterpret_cast<const uint4*>(keys)[2 * j], v1 = reinterpret_cast<const uint4*>(keys)[2 * j + 1];
        cur8[0] = v0.x; cur8[1] = v0.y; cur8[2] = v0.z; cur8[3] = v0.w; cur8[4] = v1.x; cur8[5] = v1.y; cur8[6] = v1.z; cur8[7] = v1.w;
    }
    uint32_t prev = i0 > 0 ? (uint32_t)keys[i0 - 1] : 0xffffffffu;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int idx = i0 + e;
        if (idx < L) {
            const uint32_t cur = cur8[e];
            if (cur != prev) {
                if (idx > 0) ranges[prev].y = (uint32_t)idx;
                ranges[cur].x = (uint32_t)idx;
            }
            if (idx == L - 1) ranges[cur].y = (uint32_t)L;
            prev = cur;
        }
    }
}

void launch_identify_ranges(int R, const void* keys_sorted, int key_bytes, uint2* ranges, const uint32_t* sort_err, hipStream_t s)
{
    if (R <= 0) return;
    const int threads = (R + 7) / 8;
    if (key_bytes == 4) identify_ranges_kernel<uint32_t><<<(threads + 255) / 256, 256, 0, s>>>(R, (const uint32_t*)keys_sorted, ranges, sort_err);
    else identify_ranges_kernel<uint16_t><<<(threads + 255) / 256, 256, 0, s>>>(R, (const uint16_t*)keys_sorted, ranges, sort_err);
}

} // namespace c3dgs
