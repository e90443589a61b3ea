/*
 * c3dgs_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT THE PRODUCT). See c3dgs_oracle.h.
 *
 * Reference citations are relative to /root/reference; DGR = submodules/diff-gaussian-rasterization,
 * WD = submodules/weighted_distance.  Build: see oracle/Makefile (-O2 -ffp-contract=off -fopenmp).
 *
 * glm convention used by the reference: mat3 m[c][r] is column c, row r, and
 * glm::mat3(a,b,c,d,e,f,g,h,i) fills column 0 with (a,b,c).  `m3` below keeps that layout and
 * m3_mul() keeps glm's operator* evaluation order, so fp32 results do not depend on the reader
 * re-deriving transposes.
 */
#include "c3dgs_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE_X 16   /* DGR/cuda_rasterizer/config.h:16 */
#define TILE_Y 16   /* DGR/cuda_rasterizer/config.h:17 */

/* DGR/cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

typedef struct { float x, y, z; } f3;
typedef struct { float c[3][3]; } m3; /* c[col][row], glm layout */

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* glm operator*(mat3, mat3): R[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2] */
static m3 m3_mul(const m3 a, const m3 b)
{
    m3 r;
    for (int c = 0; c < 3; c++)
        for (int q = 0; q < 3; q++)
            r.c[c][q] = a.c[0][q] * b.c[c][0] + a.c[1][q] * b.c[c][1] + a.c[2][q] * b.c[c][2];
    return r;
}
static m3 m3_t(const m3 a)
{
    m3 r;
    for (int c = 0; c < 3; c++)
        for (int q = 0; q < 3; q++)
            r.c[c][q] = a.c[q][c];
    return r;
}
static m3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    m3 r = { { { a, b, c }, { d, e, f }, { g, h, i } } };
    return r;
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t orc_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* auxiliary.h:58-66 */
static f3 xform4x3(f3 p, const float* m)
{
    f3 r = { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
             m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
             m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
    return r;
}
/* auxiliary.h:68-77 */
static void xform4x4(f3 p, const float* m, float out[4])
{
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}

/* auxiliary.h:41-44: the literals are doubles, so the arithmetic is fp64, rounded to fp32 once. */
static float ndc2pix(float v, int S)
{
    return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:46-56 */
static void get_rect(float px, float py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / TILE_X)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / TILE_Y)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + TILE_X - 1) / TILE_X)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + TILE_Y - 1) / TILE_Y)));
}

/* auxiliary.h:139-166 (only the near-plane test is live) */
static int in_frustum(f3 p, const float* view, int prefiltered, f3* p_view)
{
    if (prefiltered) return 1;
    *p_view = xform4x3(p, view);
    return !(p_view->z <= 0.01f);
}

/* rasterizer_impl.cu:54-66,141-149 */
void orc_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    (void)proj;
    for (int i = 0; i < P; i++) {
        f3 p = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] }, pv;
        present[i] = (uint8_t)in_frustum(p, view, 0, &pv);
    }
}

/* forward.cu:126-160 / forward_indexed.cu:125-159 */
static void cov3d_from_scale_rot(const float* scale, float mod, const float* rot, float* cov3D)
{
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 Mx = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_t(Mx), Mx);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}

/* forward.cu:82-121. Also returns the glm T (=W*J) and clamped t for the backward. */
static void cov2d(f3 mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* cov3D,
                  const float* view, float out[3], m3* T_out, f3* t_out, float* txtz_out, float* tytz_out)
{
    f3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;

    m3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0, 0, 0);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 T = m3_mul(Wm, J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
    cov.c[0][0] += 0.3f;
    cov.c[1][1] += 0.3f;
    out[0] = cov.c[0][0];
    out[1] = cov.c[0][1];
    out[2] = cov.c[1][1];
    if (T_out) *T_out = T;
    if (t_out) *t_out = t;
    if (txtz_out) *txtz_out = txtz;
    if (tytz_out) *tytz_out = tytz;
}

/* forward.cu:20-79 / forward_indexed.cu:20-78 (sh points at this Gaussian's [M,3] block) */
static void color_from_sh(int deg, f3 pos, f3 campos, const float* sh, int clamp_color,
                          uint8_t* clamped, float rgb[3])
{
    f3 dir = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
    float x = dir.x, y = dir.y, z = dir.z;
    for (int ch = 0; ch < 3; ch++) {
#define SHc(k) sh[(k) * 3 + ch]
        float res = SH_C0 * SHc(0);
        if (deg > 0) {
            res = res - SH_C1 * y * SHc(1) + SH_C1 * z * SHc(2) - SH_C1 * x * SHc(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SHc(4) + SH_C2[1] * yz * SHc(5) +
                      SH_C2[2] * (2.0f * zz - xx - yy) * SHc(6) + SH_C2[3] * xz * SHc(7) +
                      SH_C2[4] * (xx - yy) * SHc(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * SHc(9) + SH_C3[1] * xy * z * SHc(10) +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * SHc(11) +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHc(12) +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * SHc(13) + SH_C3[5] * z * (xx - yy) * SHc(14) +
                          SH_C3[6] * x * (xx - 3.0f * yy) * SHc(15);
                }
            }
        }
#undef SHc
        res += 0.5f;
        if (clamp_color) {
            clamped[ch] = (uint8_t)(res < 0);
            rgb[ch] = fmaxf(res, 0.0f);
        } else {
            clamped[ch] = 0;
            rgb[ch] = res;
        }
    }
}

/* forward.cu:163-265 (K2), forward_indexed.cu:162-268 (K2i), then the inclusive scan K3
 * (rasterizer_impl.cu:275). */
int orc_forward_stage1(const orc_params* p, orc_geom* g)
{
    const int P = p->P, W = p->W, H = p->H;
    const float focal_y = H / (2.0f * p->tan_fovy); /* rasterizer_impl.cu:219-220 */
    const float focal_x = W / (2.0f * p->tan_fovx);
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const int indexed = p->sh_indices != NULL || p->g_indices != NULL;
    const f3 campos = { p->campos[0], p->campos[1], p->campos[2] };

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        g->radii[i] = 0;
        g->tiles_touched[i] = 0;
        f3 po = { p->means3D[3 * i], p->means3D[3 * i + 1], p->means3D[3 * i + 2] };
        f3 p_view = { 0, 0, 0 };
        if (!in_frustum(po, p->viewmatrix, p->prefiltered, &p_view)) continue;
        if (p->prefiltered) p_view = xform4x3(po, p->viewmatrix); /* reference leaves p_view unset; depth would be garbage */

        float ph[4];
        xform4x4(po, p->projmatrix, ph);
        float p_w = 1.0f / (ph[3] + 0.0000001f);
        float projx = ph[0] * p_w, projy = ph[1] * p_w;

        const float* cov3D;
        if (p->cov3D_precomp) {
            cov3D = p->cov3D_precomp + 6 * (size_t)i;
        } else {
            if (indexed) { /* forward_indexed.cu:223 */
                int64_t gi = p->g_indices[i];
                cov3d_from_scale_rot(p->scales + 3 * gi, p->scale_factors[i] * p->scale_modifier,
                                     p->rotations + 4 * gi, g->cov3D + 6 * (size_t)i);
            } else {       /* forward.cu:220 */
                cov3d_from_scale_rot(p->scales + 3 * (size_t)i, p->scale_modifier,
                                     p->rotations + 4 * (size_t)i, g->cov3D + 6 * (size_t)i);
            }
            cov3D = g->cov3D + 6 * (size_t)i;
        }
        float cov[3];
        cov2d(po, focal_x, focal_y, p->tan_fovx, p->tan_fovy, cov3D, p->viewmatrix, cov, NULL, NULL, NULL, NULL);

        float det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };

        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix = ndc2pix(projx, W), piy = ndc2pix(projy, H);
        int rmin[2], rmax[2];
        get_rect(pix, piy, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (p->colors_precomp == NULL) {
            const float* sh = p->shs + (size_t)(indexed ? p->sh_indices[i] : i) * p->M * 3;
            color_from_sh(p->D, po, campos, sh, p->clamp_color, g->clamped + 3 * (size_t)i, g->rgb + 3 * (size_t)i);
        }
        g->depths[i] = p_view.z;
        g->radii[i] = (int32_t)my_radius;
        g->means2D[2 * (size_t)i] = pix;
        g->means2D[2 * (size_t)i + 1] = piy;
        g->conic_opacity[4 * (size_t)i + 0] = conic[0];
        g->conic_opacity[4 * (size_t)i + 1] = conic[1];
        g->conic_opacity[4 * (size_t)i + 2] = conic[2];
        g->conic_opacity[4 * (size_t)i + 3] = p->opacities[i];
        g->tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += g->tiles_touched[i]; g->point_offsets[i] = acc; }
    return P > 0 ? (int)g->point_offsets[P - 1] : 0;
}

/* stable LSD radix sort on bits [0,end_bit) of 64-bit keys with 32-bit payload:
 * the semantics of cub::DeviceRadixSort::SortPairs(..., 0, 32+bit) (rasterizer_impl.cu:301-306). */
static void radix_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                             size_t n, int end_bit)
{
    uint64_t* ka = (uint64_t*)malloc(n * sizeof(uint64_t) + 8);
    uint32_t* va = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
    uint64_t* kb = (uint64_t*)malloc(n * sizeof(uint64_t) + 8);
    uint32_t* vb = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
    memcpy(ka, kin, n * sizeof(uint64_t));
    memcpy(va, vin, n * sizeof(uint32_t));
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        size_t hist[257] = { 0 };
        for (size_t i = 0; i < n; i++) hist[((ka[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; i++) {
            size_t d = hist[(ka[i] >> shift) & mask]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, n * sizeof(uint64_t));
    memcpy(vout, va, n * sizeof(uint32_t));
    free(ka); free(va); free(kb); free(vb);
}

/* forward.cu:270-383 for one pixel; list = this tile's slice of point_list. */
static void blend_pixel_fwd(const orc_geom* g, const float* feat, const uint32_t* list, int n,
                            float pxf, float pyf, const float* bg, float outc[3], float* T_out,
                            uint32_t* ncontrib_out)
{
    float T = 1.0f, C[3] = { 0, 0, 0 };
    uint32_t contributor = 0, last_contributor = 0;
    for (int j = 0; j < n; j++) {
        contributor++;
        uint32_t id = list[j];
        float dx = g->means2D[2 * (size_t)id] - pxf, dy = g->means2D[2 * (size_t)id + 1] - pyf;
        const float* co = g->conic_opacity + 4 * (size_t)id;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf(0.99f, co[3] * expf(power));
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) break; /* done = true; nothing later changes this pixel */
        for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * (size_t)id + ch] * alpha * T;
        T = test_T;
        last_contributor = contributor;
    }
    *T_out = T;
    *ncontrib_out = last_contributor;
    for (int ch = 0; ch < 3; ch++) outc[ch] = C[ch] + T * bg[ch];
}

void orc_forward_stage2(const orc_params* p, const orc_geom* g, int R,
                        uint64_t* keys_unsorted, uint32_t* values_unsorted,
                        uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges,
                        float* out_color, float* final_T, uint32_t* n_contrib)
{
    const int P = p->P, W = p->W, H = p->H;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const int T = gx * gy;

    /* K5 duplicateWithKeys, rasterizer_impl.cu:70-111 */
    for (int i = 0; i < P; i++) {
        if (g->radii[i] > 0) {
            uint32_t off = (i == 0) ? 0 : g->point_offsets[i - 1];
            int rmin[2], rmax[2];
            get_rect(g->means2D[2 * (size_t)i], g->means2D[2 * (size_t)i + 1], g->radii[i], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &g->depths[i], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys_unsorted[off] = key;
                    values_unsorted[off] = (uint32_t)i;
                    off++;
                }
        }
    }
    /* K6, rasterizer_impl.cu:298-306 */
    int bit = (int)orc_get_higher_msb((uint32_t)T);
    radix_sort_pairs(keys_unsorted, values_unsorted, keys_sorted, point_list, (size_t)R, 32 + bit);

    /* K7 + K8, rasterizer_impl.cu:308-316, 116-138 */
    memset(ranges, 0, (size_t)T * 2 * sizeof(uint32_t));
    for (int idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }

    /* K9, forward.cu:270-383 */
    const float* feat = p->colors_precomp ? p->colors_precomp : g->rgb; /* rasterizer_impl.cu:319 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < T; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE_Y; ly++)
            for (int lx = 0; lx < TILE_X; lx++) {
                int px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
                if (px >= W || py >= H) continue;
                size_t pix_id = (size_t)W * py + px;
                float c[3], Tf;
                uint32_t nc;
                blend_pixel_fwd(g, feat, point_list + r0, (int)(r1 - r0), (float)px, (float)py, p->bg, c, &Tf, &nc);
                final_T[pix_id] = Tf;
                n_contrib[pix_id] = nc;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = c[ch];
            }
    }
}

/* ---------------------------------------------------------------- backward */

static inline void atomic_add_d(double* dst, double v)
{
#pragma omp atomic
    *dst += v;
}

/* backward.cu:20-139 / backward_indexed.cu:20-201.  dsh accumulates into doubles (the indexed
 * variant scatter-adds; the plain variant writes once, which accumulation into zeros equals). */
static void sh_backward(int deg, int M, f3 pos, f3 campos, const float* sh, const uint8_t* clamped,
                        const float dL_dcolor[3], double* dL_dsh /*[M,3]*/, float dL_dmean_add[3])
{
    (void)M;
    f3 d0 = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
    float len = sqrtf(d0.x * d0.x + d0.y * d0.y + d0.z * d0.z);
    float x = d0.x / len, y = d0.y / len, z = d0.z / len;
    float g[3];
    for (int c = 0; c < 3; c++) g[c] = dL_dcolor[c] * (clamped[c] ? 0.f : 1.f);

    float dx[3] = { 0, 0, 0 }, dy[3] = { 0, 0, 0 }, dz[3] = { 0, 0, 0 };
    float basis[16];
    int nb = 1;
    basis[0] = SH_C0;
    if (deg > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
        nb = 4;
        for (int c = 0; c < 3; c++) {
            dx[c] = -SH_C1 * sh[3 * 3 + c];
            dy[c] = -SH_C1 * sh[1 * 3 + c];
            dz[c] = SH_C1 * sh[2 * 3 + c];
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.f * zz - xx - yy);
            basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
            nb = 9;
            for (int c = 0; c < 3; c++) {
#define S(k) sh[(k) * 3 + c]
                dx[c] += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                dy[c] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                dz[c] += SH_C2[1] * y * S(5) + SH_C2[2] * 2.f * 2.f * z * S(6) + SH_C2[3] * x * S(7);
#undef S
            }
            if (deg > 2) {
                basis[9] = SH_C3[0] * y * (3.f * xx - yy);
                basis[10] = SH_C3[1] * xy * z;
                basis[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                basis[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                basis[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                basis[14] = SH_C3[5] * z * (xx - yy);
                basis[15] = SH_C3[6] * x * (xx - 3.f * yy);
                nb = 16;
                for (int c = 0; c < 3; c++) {
#define S(k) sh[(k) * 3 + c]
                    dx[c] += (SH_C3[0] * S(9) * 3.f * 2.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                              SH_C3[3] * S(12) * -3.f * 2.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                              SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy));
                    dy[c] += (SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                              SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -3.f * 2.f * yz +
                              SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz + SH_C3[6] * S(15) * -3.f * 2.f * xy);
                    dz[c] += (SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 4.f * 2.f * yz +
                              SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 4.f * 2.f * xz +
                              SH_C3[5] * S(14) * (xx - yy));
#undef S
                }
            }
        }
    }
    for (int k = 0; k < nb; k++)
        for (int c = 0; c < 3; c++) atomic_add_d(&dL_dsh[k * 3 + c], (double)(basis[k] * g[c]));

    float ddir[3] = { dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2],
                      dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2],
                      dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2] };
    /* auxiliary.h:107-117 dnormvdv(float3) */
    float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmean_add[0] = ((+sum2 - d0.x * d0.x) * ddir[0] - d0.y * d0.x * ddir[1] - d0.z * d0.x * ddir[2]) * invsum32;
    dL_dmean_add[1] = (-d0.x * d0.y * ddir[0] + (sum2 - d0.y * d0.y) * ddir[1] - d0.z * d0.y * ddir[2]) * invsum32;
    dL_dmean_add[2] = (-d0.x * d0.z * ddir[0] - d0.y * d0.z * ddir[1] + (sum2 - d0.z * d0.z) * ddir[2]) * invsum32;
}

/* backward.cu:278-341 / backward_indexed.cu:206-282, written against the standard rotation
 * matrix Rm (rows as listed in the reference constructor): L = Rm*diag(s), Sigma = L*L^T.
 * Outputs d_s (gradient w.r.t. the effective scale s, reference `dL_dscale` before any
 * scale_factor chain rule) and dq (w.r.t. the un-normalised quaternion (r,x,y,z)). */
static void cov3d_backward(const float s[3], const float* rot, const float* dL_dcov3D, float d_s[3], float dq[4])
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float Rm[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                       { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                       { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
    float G[3][3] = { { dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2] },
                      { 0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4] },
                      { 0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5] } };
    float dLdL[3][3]; /* 2 * G * L, with L[i][j] = Rm[i][j]*s[j] */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float acc = 0.f;
            for (int k = 0; k < 3; k++) acc += G[i][k] * (Rm[k][j] * s[j]);
            dLdL[i][j] = 2.0f * acc;
        }
    float Q[3][3];
    for (int j = 0; j < 3; j++) {
        d_s[j] = Rm[0][j] * dLdL[0][j] + Rm[1][j] * dLdL[1][j] + Rm[2][j] * dLdL[2][j];
        for (int i = 0; i < 3; i++) Q[i][j] = dLdL[i][j] * s[j];
    }
    dq[0] = 2 * z * (Q[1][0] - Q[0][1]) + 2 * y * (Q[0][2] - Q[2][0]) + 2 * x * (Q[2][1] - Q[1][2]);
    dq[1] = 2 * y * (Q[0][1] + Q[1][0]) + 2 * z * (Q[0][2] + Q[2][0]) + 2 * r * (Q[2][1] - Q[1][2]) - 4 * x * (Q[2][2] + Q[1][1]);
    dq[2] = 2 * x * (Q[0][1] + Q[1][0]) + 2 * r * (Q[0][2] - Q[2][0]) + 2 * z * (Q[2][1] + Q[1][2]) - 4 * y * (Q[2][2] + Q[0][0]);
    dq[3] = 2 * r * (Q[1][0] - Q[0][1]) + 2 * x * (Q[0][2] + Q[2][0]) + 2 * y * (Q[2][1] + Q[1][2]) - 4 * z * (Q[1][1] + Q[0][0]);
}

void orc_backward(const orc_params* p, const orc_geom* g, int R,
                  const uint32_t* point_list, const uint32_t* ranges,
                  const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                  float* dL_dscale_factor, float* dL_drot)
{
    (void)R;
    const int P = p->P, W = p->W, H = p->H, M = p->M;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const int T = gx * gy;
    const float focal_y = H / (2.0f * p->tan_fovy);
    const float focal_x = W / (2.0f * p->tan_fovx);
    const int indexed = p->sh_indices != NULL || p->g_indices != NULL;
    const float* colors = p->colors_precomp ? p->colors_precomp : g->rgb; /* rasterizer_impl.cu:388 */
    const float* bg = p->bg;

    /* K10 accumulators: [P][9] = dcolor(3) dmean2D(2) dconic(.x .y .w) dopacity, float64
     * (the reference accumulates with order-nondeterministic fp32 atomics, backward.cu:523-554). */
    double* acc = (double*)calloc((size_t)P * 9 + 1, sizeof(double));

    /* backward.cu:399-557 */
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < T; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        int n = (int)(r1 - r0);
        for (int ly = 0; ly < TILE_Y; ly++)
            for (int lx = 0; lx < TILE_X; lx++) {
                int px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
                if (px >= W || py >= H) continue;
                size_t pix_id = (size_t)W * py + px;
                const float pxf = (float)px, pyf = (float)py;
                const float T_final = final_T[pix_id];
                float Tt = T_final;
                const int last_contributor = (int)n_contrib[pix_id];
                float accum_rec[3] = { 0, 0, 0 }, last_color[3] = { 0, 0, 0 }, last_alpha = 0;
                float dpx[3];
                for (int c = 0; c < 3; c++) dpx[c] = dL_dpix[(size_t)c * H * W + pix_id];
                /* Gaussians behind the last contributor are skipped (contributor >= last_contributor) */
                for (int k = imin(n, last_contributor) - 1; k >= 0; k--) {
                    uint32_t id = point_list[r0 + k];
                    float dx = g->means2D[2 * (size_t)id] - pxf, dy = g->means2D[2 * (size_t)id + 1] - pyf;
                    const float* co = g->conic_opacity + 4 * (size_t)id;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float G = expf(power);
                    float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    Tt = Tt / (1.f - alpha);
                    float dchannel_dcolor = alpha * Tt;
                    float dL_dalpha = 0.0f;
                    double* a = acc + (size_t)id * 9;
                    for (int ch = 0; ch < 3; ch++) {
                        float c = colors[3 * (size_t)id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
                        atomic_add_d(&a[ch], (double)(dchannel_dcolor * dpx[ch]));
                    }
                    dL_dalpha *= Tt;
                    last_alpha = alpha;
                    float bg_dot = 0;
                    for (int c = 0; c < 3; c++) bg_dot += bg[c] * dpx[c];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;

                    float dL_dG = co[3] * dL_dalpha;
                    float gdx = G * dx, gdy = G * dy;
                    float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    float dG_ddely = -gdy * co[2] - gdx * co[1];
                    atomic_add_d(&a[3], (double)(dL_dG * dG_ddelx * ddelx_dx));
                    atomic_add_d(&a[4], (double)(dL_dG * dG_ddely * ddely_dy));
                    atomic_add_d(&a[5], (double)(-0.5f * gdx * dx * dL_dG));
                    atomic_add_d(&a[6], (double)(-0.5f * gdx * dy * dL_dG));
                    atomic_add_d(&a[7], (double)(-0.5f * gdy * dy * dL_dG));
                    atomic_add_d(&a[8], (double)(G * dL_dalpha));
                }
            }
    }
    for (int i = 0; i < P; i++) {
        const double* a = acc + (size_t)i * 9;
        dL_dcolors[3 * (size_t)i + 0] = (float)a[0];
        dL_dcolors[3 * (size_t)i + 1] = (float)a[1];
        dL_dcolors[3 * (size_t)i + 2] = (float)a[2];
        dL_dmean2D[3 * (size_t)i + 0] = (float)a[3];
        dL_dmean2D[3 * (size_t)i + 1] = (float)a[4];
        dL_dconic[4 * (size_t)i + 0] = (float)a[5];
        dL_dconic[4 * (size_t)i + 1] = (float)a[6];
        dL_dconic[4 * (size_t)i + 3] = (float)a[7];
        dL_dopacity[i] = (float)a[8];
    }
    free(acc);

    /* codebook-sized (indexed) or P-sized double accumulators for K12(i) */
    const size_t n_sh = p->shs ? (size_t)(indexed ? p->SHS : P) * M * 3 : 0;
    const size_t n_g = p->scales ? (size_t)(indexed ? p->GS : P) : 0;
    double* acc_sh = (double*)calloc(n_sh + 1, sizeof(double));
    double* acc_scale = (double*)calloc(n_g * 3 + 1, sizeof(double));
    double* acc_rot = (double*)calloc(n_g * 4 + 1, sizeof(double));
    const f3 campos = { p->campos[0], p->campos[1], p->campos[2] };
    const float* proj = p->projmatrix;
    const float* view = p->viewmatrix;

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(g->radii[i] > 0)) continue;
        f3 m = { p->means3D[3 * (size_t)i], p->means3D[3 * (size_t)i + 1], p->means3D[3 * (size_t)i + 2] };

        /* ---- K11 computeCov2DCUDA, backward.cu:144-274, in matrix form:
         * A = upper 2x3 of J*R_w2c (reference T[i][j] = A[i][j]); cov2D = A*Sigma*A^T + 0.3*I */
        const float* cov3D = (p->cov3D_precomp ? p->cov3D_precomp : g->cov3D) + 6 * (size_t)i;
        float cov[3], txtz, tytz;
        m3 Tg;
        f3 t;
        cov2d(m, focal_x, focal_y, p->tan_fovx, p->tan_fovy, cov3D, view, cov, &Tg, &t, &txtz, &tytz);
        const float limx = 1.3f * p->tan_fovx, limy = 1.3f * p->tan_fovy;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        float A[2][3] = { { Tg.c[0][0], Tg.c[0][1], Tg.c[0][2] }, { Tg.c[1][0], Tg.c[1][1], Tg.c[1][2] } };
        float Sg[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
        float a = cov[0], b = cov[1], c = cov[2];
        float dcon[3] = { dL_dconic[4 * (size_t)i], dL_dconic[4 * (size_t)i + 1], dL_dconic[4 * (size_t)i + 3] };
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov3D + 6 * (size_t)i;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcon[0] + 2 * b * c * dcon[1] + (denom - a * c) * dcon[2]);
            dL_dc = denom2inv * (-a * a * dcon[2] + 2 * a * b * dcon[1] + (denom - a * c) * dcon[0]);
            dL_db = denom2inv * 2 * (b * c * dcon[0] - (denom + 2 * b * b) * dcon[1] + a * b * dcon[2]);
            /* dL/dSigma_full = A^T * Gm * A, Gm = [[da, db/2],[db/2, dc]]; off-diagonals appear twice */
            float Gm[2][2] = { { dL_da, 0.5f * dL_db }, { 0.5f * dL_db, dL_dc } };
            float S3[3][3];
            for (int u = 0; u < 3; u++)
                for (int v = 0; v < 3; v++) {
                    float s = 0.f;
                    for (int q = 0; q < 2; q++)
                        for (int w = 0; w < 2; w++) s += A[q][u] * Gm[q][w] * A[w][v];
                    S3[u][v] = s;
                }
            dcov[0] = S3[0][0]; dcov[3] = S3[1][1]; dcov[5] = S3[2][2];
            dcov[1] = 2.f * S3[0][1]; dcov[2] = 2.f * S3[0][2]; dcov[4] = 2.f * S3[1][2];
        } else {
            for (int q = 0; q < 6; q++) dcov[q] = 0;
        }
        /* dL/dA = 2*Gm*A*Sigma */
        float AS[2][3], dA[2][3];
        for (int q = 0; q < 2; q++)
            for (int v = 0; v < 3; v++) AS[q][v] = A[q][0] * Sg[0][v] + A[q][1] * Sg[1][v] + A[q][2] * Sg[2][v];
        for (int v = 0; v < 3; v++) {
            dA[0][v] = 2 * AS[0][v] * dL_da + AS[1][v] * dL_db;
            dA[1][v] = 2 * AS[1][v] * dL_dc + AS[0][v] * dL_db;
        }
        /* R_w2c[k][j] = view[4*j+k]; dL/dJ[i][k] = sum_j dA[i][j]*R[k][j] */
        float dJ00 = view[0] * dA[0][0] + view[4] * dA[0][1] + view[8] * dA[0][2];
        float dJ02 = view[2] * dA[0][0] + view[6] * dA[0][1] + view[10] * dA[0][2];
        float dJ11 = view[1] * dA[1][0] + view[5] * dA[1][1] + view[9] * dA[1][2];
        float dJ12 = view[2] * dA[1][0] + view[6] * dA[1][1] + view[10] * dA[1][2];
        float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = x_grad_mul * -focal_x * tz2 * dJ02;
        float dty = y_grad_mul * -focal_y * tz2 * dJ12;
        float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2 * focal_x * t.x) * tz3 * dJ02 + (2 * focal_y * t.y) * tz3 * dJ12;
        /* auxiliary.h:89-97 transformVec4x3Transpose */
        float dmean[3] = { view[0] * dtx + view[1] * dty + view[2] * dtz,
                           view[4] * dtx + view[5] * dty + view[6] * dtz,
                           view[8] * dtx + view[9] * dty + view[10] * dtz };

        /* ---- K12 preprocessCUDA, backward.cu:346-396 / backward_indexed.cu:287-342 */
        float mh[4];
        xform4x4(m, proj, mh);
        float m_w = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        float d2x = dL_dmean2D[3 * (size_t)i], d2y = dL_dmean2D[3 * (size_t)i + 1];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;

        if (p->shs) {
            size_t si = (size_t)(indexed ? p->sh_indices[i] : i);
            float add[3];
            sh_backward(p->D, M, m, campos, p->shs + si * M * 3, g->clamped + 3 * (size_t)i,
                        dL_dcolors + 3 * (size_t)i, acc_sh + si * M * 3, add);
            dmean[0] += add[0]; dmean[1] += add[1]; dmean[2] += add[2];
        }
        dL_dmean3D[3 * (size_t)i + 0] = dmean[0];
        dL_dmean3D[3 * (size_t)i + 1] = dmean[1];
        dL_dmean3D[3 * (size_t)i + 2] = dmean[2];

        if (p->scales) {
            size_t gi = (size_t)(indexed ? p->g_indices[i] : i);
            const float* sc = p->scales + 3 * gi;
            float sf = indexed ? p->scale_factors[i] : 1.0f;
            float s[3];
            if (indexed) { /* backward_indexed.cu:224 : s = scale_factor * mod * scale */
                s[0] = sf * p->scale_modifier * sc[0]; s[1] = sf * p->scale_modifier * sc[1]; s[2] = sf * p->scale_modifier * sc[2];
            } else {       /* backward.cu:295 */
                s[0] = p->scale_modifier * sc[0]; s[1] = p->scale_modifier * sc[1]; s[2] = p->scale_modifier * sc[2];
            }
            float d_s[3], dq[4];
            cov3d_backward(s, p->rotations + 4 * gi, dcov, d_s, dq);
            if (indexed) { /* backward_indexed.cu:255-262 */
                for (int q = 0; q < 3; q++) atomic_add_d(&acc_scale[3 * gi + q], (double)(d_s[q] * sf));
                dL_dscale_factor[i] = d_s[0] * sc[0] + d_s[1] * sc[1] + d_s[2] * sc[2];
            } else {
                for (int q = 0; q < 3; q++) atomic_add_d(&acc_scale[3 * gi + q], (double)d_s[q]);
            }
            for (int q = 0; q < 4; q++) atomic_add_d(&acc_rot[4 * gi + q], (double)dq[q]);
        }
    }
    for (size_t k = 0; k < n_sh; k++) dL_dsh[k] = (float)acc_sh[k];
    for (size_t k = 0; k < n_g * 3; k++) dL_dscale[k] = (float)acc_scale[k];
    for (size_t k = 0; k < n_g * 4; k++) dL_drot[k] = (float)acc_rot[k];
    free(acc_sh); free(acc_scale); free(acc_rot);
}

/* ---------------------------------------------------------------- VQ */

/* WD/weighted_distance.cu:9-44.  `result += diff*diff` is written as fmaf(diff, diff, result):
 * the single contraction nvcc performs by default (-fmad=true) and, bit for bit, what a
 * k-ordered fp32 MFMA/FMA chain produces on gfx950. Strict '<' keeps the lowest index on ties. */
void orc_weighted_distance(int64_t N, int C, int K, const float* coefs, const float* codebook,
                           float* dist, int64_t* idx)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        const float* x = coefs + n * K;
        float best = FLT_MAX;
        int64_t besti = 0; /* reference leaves min_index uninitialised when nothing is < FLT_MAX */
        for (int c = 0; c < C; c++) {
            const float* cb = codebook + (size_t)c * K;
            float r = 0.f;
            for (int k = 0; k < K; k++) {
                float d = x[k] - cb[k];
                r = fmaf(d, d, r);
            }
            if (r < best) { best = r; besti = c; }
        }
        dist[n] = best;
        idx[n] = besti;
    }
}

/* compression/vq.py:28-35 + 45-46 */
double orc_vq_update(int64_t B, int K, int D, const float* x, const float* w, float* codebook,
                     float* entry_importance, double decay_d, double eps_d, float* min_dists, int64_t* idx_out)
{
    /* torch casts the Python-double scalars to fp32 once: mul_(decay), add_(new, alpha=1-decay), + eps */
    const float decay = (float)decay_d, alpha = (float)(1.0 - decay_d), eps = (float)eps_d;
    float* md = min_dists ? min_dists : (float*)malloc((size_t)B * sizeof(float));
    int64_t* ix = idx_out ? idx_out : (int64_t*)malloc((size_t)B * sizeof(int64_t));
    orc_weighted_distance(B, K, D, x, codebook, md, ix);
    double* acc_w = (double*)calloc((size_t)K, sizeof(double));
    double* acc_xw = (double*)calloc((size_t)K * D, sizeof(double));
    double msum = 0.0;
    for (int64_t n = 0; n < B; n++) {
        int64_t k = ix[n];
        acc_w[k] += (double)w[n];
        for (int d = 0; d < D; d++) acc_xw[k * D + d] += (double)(x[n * D + d] * w[n]); /* vq.py:33 product in fp32 */
        msum += (double)md[n];
    }
    for (int k = 0; k < K; k++) {
        float aw = (float)acc_w[k];
        /* ema_inplace: moving_avg.mul_(decay).add_(new, alpha=1-decay), vq.py:45-46 */
        entry_importance[k] = entry_importance[k] * decay + alpha * aw;
        for (int d = 0; d < D; d++) {
            float nw = (float)acc_xw[(size_t)k * D + d] / (aw + eps);
            codebook[(size_t)k * D + d] = codebook[(size_t)k * D + d] * decay + alpha * nw;
        }
    }
    free(acc_w); free(acc_xw);
    if (!min_dists) free(md);
    if (!idx_out) free(ix);
    return B > 0 ? msum / (double)B : 0.0;
}

/* compression/vq.py:73-77 */
void orc_vq_trace_normalize(int K, int D, float* codebook)
{
    if (D < 6) return;
    for (int k = 0; k < K; k++) {
        float* c = codebook + (size_t)k * D;
        float tr = c[0] + c[3] + c[5];
        for (int d = 0; d < D; d++) c[d] = c[d] / tr;
    }
}

/* ---------------------------------------------------------------- test hooks (golden-vector pinning) */
void orc_test_color_from_sh(int n, int deg, int M, const float* pos, const float* campos, const float* sh,
                            int clamp_color, uint8_t* clamped, float* rgb)
{
    f3 cp = { campos[0], campos[1], campos[2] };
    for (int i = 0; i < n; i++) {
        f3 p = { pos[3 * i], pos[3 * i + 1], pos[3 * i + 2] };
        color_from_sh(deg, p, cp, sh + (size_t)i * M * 3, clamp_color, clamped + 3 * i, rgb + 3 * i);
    }
}

void orc_test_cov3d(int n, const float* scales, float mod, const float* rots, float* cov)
{
    for (int i = 0; i < n; i++) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov + 6 * i);
}

/* backward.cu:20-139 per Gaussian: dL_dsh [n,M,3] (written) and the SH part of dL_dmean [n,3] for upstream dL_dcolor [n,3]. */
void orc_test_sh_backward(int n, int deg, int M, const float* pos, const float* campos, const float* sh,
                          const uint8_t* clamped, const float* dL_dcolor, float* dL_dsh, float* dL_dmean)
{
    f3 cp = { campos[0], campos[1], campos[2] };
    double* acc = (double*)malloc((size_t)M * 3 * sizeof(double));
    for (int i = 0; i < n; i++) {
        f3 p = { pos[3 * i], pos[3 * i + 1], pos[3 * i + 2] };
        for (int k = 0; k < M * 3; k++) acc[k] = 0.0;
        sh_backward(deg, M, p, cp, sh + (size_t)i * M * 3, clamped + 3 * i, dL_dcolor + 3 * i, acc, dL_dmean + 3 * i);
        for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = (float)acc[k];
    }
    free(acc);
}

/* backward.cu:278-341 per Gaussian: d_s [n,3] = the reference's dL_dscale (gradient w.r.t. mod * scale, :322-325) and
 * dq [n,4] w.r.t. the UN-normalised quaternion (:340), for upstream dL_dcov3D [n,6]. */
void orc_test_cov3d_backward(int n, const float* scales, float mod, const float* rots, const float* dL_dcov3D,
                             float* d_s, float* dq)
{
    for (int i = 0; i < n; i++) {
        const float s[3] = { mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2] };   /* backward.cu:295 */
        cov3d_backward(s, rots + 4 * i, dL_dcov3D + 6 * i, d_s + 3 * i, dq + 4 * i);
    }
}

/* ---------------------------------------------------------------- N3: L1 + SSIM loss (utils/loss_utils.py:17-63)
 * loss = (1-lambda)*mean|x-y| + lambda*(1 - mean(ssim_map)), the QAT loss of finetune.py:48.
 * ssim: 11x11 Gaussian window (sigma 1.5, normalised in fp32 like `gaussian`/`create_window`, :23-31), zero padding 5,
 * per channel; direct 121-tap sums in float64. dL_dimg (may be NULL) is the analytic gradient w.r.t. img. */
void orc_l1_ssim(int C, int H, int W, const float* img, const float* gt, double lambda, double* out /*[3]: loss, l1, ssim*/,
                 float* dL_dimg)
{
    float g[11], gs = 0.f;
    for (int i = 0; i < 11; i++) { g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); gs += g[i]; }
    for (int i = 0; i < 11; i++) g[i] = g[i] / gs;
    float w2[11][11];
    for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) w2[i][j] = g[i] * g[j];
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    const size_t HW = (size_t)H * W, N = (size_t)C * HW;
    double* Dmu = (double*)calloc(N + 1, sizeof(double));
    double* Ds1 = (double*)calloc(N + 1, sizeof(double));
    double* Ds12 = (double*)calloc(N + 1, sizeof(double));
    double ssim_sum = 0.0, l1_sum = 0.0;
#pragma omp parallel for reduction(+ : ssim_sum, l1_sum) schedule(static)
    for (int cy = 0; cy < C * H; cy++) {
        const int c = cy / H, y = cy % H;
        const float* X = img + (size_t)c * HW;
        const float* Y = gt + (size_t)c * HW;
        for (int x = 0; x < W; x++) {
            double mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
            for (int i = 0; i < 11; i++) {
                const int yy = y + i - 5;
                if (yy < 0 || yy >= H) continue;
                for (int j = 0; j < 11; j++) {
                    const int xx = x + j - 5;
                    if (xx < 0 || xx >= W) continue;
                    const double wv = w2[i][j], a = X[(size_t)yy * W + xx], b = Y[(size_t)yy * W + xx];
                    mu1 += wv * a; mu2 += wv * b; e11 += wv * a * a; e22 += wv * b * b; e12 += wv * a * b;
                }
            }
            const double s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const double A = 2 * mu1 * mu2 + C1, B = 2 * s12 + C2, Cc = mu1 * mu1 + mu2 * mu2 + C1, Dd = s1 + s2 + C2;
            const double m = (A * B) / (Cc * Dd);
            ssim_sum += m;
            const size_t p = (size_t)c * HW + (size_t)y * W + x;
            l1_sum += fabs((double)X[(size_t)y * W + x] - (double)Y[(size_t)y * W + x]);
            /* partials of m w.r.t. (mu1, E[x^2], E[xy]) with mu2, E[y^2] fixed */
            const double dm_ds1 = -A * B / (Cc * Dd * Dd);
            const double dm_ds12 = 2 * A / (Cc * Dd);
            const double dm_dmu1_s = (2 * mu2 * B) / (Cc * Dd) - (A * B * 2 * mu1) / (Cc * Cc * Dd);
            Dmu[p] = dm_dmu1_s - 2 * mu1 * dm_ds1 - mu2 * dm_ds12;
            Ds1[p] = dm_ds1;
            Ds12[p] = dm_ds12;
        }
    }
    const double l1 = l1_sum / (double)N, ss = ssim_sum / (double)N;
    out[0] = (1.0 - lambda) * l1 + lambda * (1.0 - ss);
    out[1] = l1;
    out[2] = ss;
    if (dL_dimg) {
#pragma omp parallel for schedule(static)
        for (int cy = 0; cy < C * H; cy++) {
            const int c = cy / H, y = cy % H;
            for (int x = 0; x < W; x++) {
                const size_t p = (size_t)c * HW + (size_t)y * W + x;
                double a = 0, b = 0, d = 0;
                for (int i = 0; i < 11; i++) {
                    const int yy = y + i - 5;
                    if (yy < 0 || yy >= H) continue;
                    for (int j = 0; j < 11; j++) {
                        const int xx = x + j - 5;
                        if (xx < 0 || xx >= W) continue;
                        const size_t q = (size_t)c * HW + (size_t)yy * W + xx;
                        const double wv = w2[i][j];
                        a += wv * Dmu[q]; b += wv * Ds1[q]; d += wv * Ds12[q];
                    }
                }
                const double xv = img[p], yv = gt[p];
                const double sgn = xv > yv ? 1.0 : (xv < yv ? -1.0 : 0.0);
                dL_dimg[p] = (float)((1.0 - lambda) * sgn / (double)N - lambda * (a + 2 * xv * b + yv * d) / (double)N);
            }
        }
    }
    free(Dmu); free(Ds1); free(Ds12);
}

/* ---------------------------------------------------------------- N4: Morton order (scene/gaussian_model.py:997-1003, 1417-1432)
 * xyz_q = ((2^21-1) * (xyz - min) / (max - min)).long()  in fp32, axes ordered by ascending extent
 * (pp_diap.argsort()), 21-bit interleave x | y<<1 | z<<2, then a STABLE ascending sort of the codes. */
static uint64_t split_by_3(uint64_t a)
{
    uint64_t x = a & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

void orc_morton_codes(int P, const float* xyz, int64_t* codes, int32_t axis_order[3])
{
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX }, diap[3];
    for (int i = 0; i < P; i++)
        for (int a = 0; a < 3; a++) {
            const float v = xyz[3 * (size_t)i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    for (int a = 0; a < 3; a++) diap[a] = mx[a] - mn[a];
    /* argsort ascending (ties: lower axis first, as torch's stable CPU sort of 3 elements would give) */
    int ord[3] = { 0, 1, 2 };
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (diap[ord[j]] < diap[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int a = 0; a < 3; a++) axis_order[a] = ord[a];
    for (int i = 0; i < P; i++) {
        uint64_t q[3];
        for (int a = 0; a < 3; a++) {
            const float v = 2097151.0f * (xyz[3 * (size_t)i + a] - mn[a]) / diap[a];
            q[a] = (uint64_t)(int64_t)v;
        }
        codes[i] = (int64_t)(split_by_3(q[ord[0]]) | split_by_3(q[ord[1]]) << 1 | split_by_3(q[ord[2]]) << 2);
    }
}
