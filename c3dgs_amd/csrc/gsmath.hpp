// gsmath.hpp -- per-Gaussian device math (projection, EWA covariance, SH) for gfx950.
//
// Translation units that include this file are compiled with -ffp-contract=off: radii, tile
// rectangles and depth bits feed integer tile keys, which are specified bit-exactly (every
// operation a single IEEE fp32 op in the order written; division and sqrt correctly rounded,
// which is hipcc's default). Matrices follow the reference's glm semantics: m.c[col][row],
// products evaluated as glm's operator* does.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace c3dgs {

// SH constants, reference cuda_rasterizer/auxiliary.h:22-39
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                           SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                           SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                           SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                           SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;

struct f3 { float x, y, z; };
struct m3 { float c[3][3]; }; // c[col][row]

__device__ __forceinline__ m3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    m3 r;
    r.c[0][0] = a; r.c[0][1] = b; r.c[0][2] = c;
    r.c[1][0] = d; r.c[1][1] = e; r.c[1][2] = f;
    r.c[2][0] = g; r.c[2][1] = h; r.c[2][2] = i;
    return r;
}
// glm operator*: R[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]
__device__ __forceinline__ m3 m3_mul(const m3& a, const m3& b)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.c[c][q] = a.c[0][q] * b.c[c][0] + a.c[1][q] * b.c[c][1] + a.c[2][q] * b.c[c][2];
    return r;
}
__device__ __forceinline__ m3 m3_t(const m3& a)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.c[c][q] = a.c[q][c];
    return r;
}

// auxiliary.h:58-66
__device__ __forceinline__ f3 xform4x3(const f3 p, const float* __restrict__ m)
{
    f3 r;
    r.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    r.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    r.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return r;
}
// auxiliary.h:68-77
__device__ __forceinline__ float4 xform4x4(const f3 p, const float* __restrict__ m)
{
    float4 r;
    r.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    r.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    r.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    r.w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    return r;
}
// auxiliary.h:41-44 (double arithmetic: the reference's literals are doubles)
__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// forward.cu:126-160 (scale already multiplied by the modifier / scale factor by the caller's `mod`)
__device__ __forceinline__ void cov3d_from_scale_rot(const float sx, const float sy, const float sz, const float mod,
                                                     const float4 rot, float cov3D[6])
{
    m3 S = m3_cols(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f);
    S.c[0][0] = mod * sx;
    S.c[1][1] = mod * sy;
    S.c[2][2] = mod * sz;
    const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
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

struct Cov2D {
    float a, b, c;   // cov2D (0,0) (0,1) (1,1), low-pass included
    m3 T;            // glm T = W*J: T.c[i][j] = (J*R_w2c)[i][j]
    f3 t;            // view-space mean with clamped x,y
    float txtz, tytz;
};

// forward.cu:82-121
__device__ __forceinline__ Cov2D cov2d(const f3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                       const float cov3D[6], const float* __restrict__ view)
{
    Cov2D o;
    f3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    m3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0.f, 0.f, 0.f);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    o.T = m3_mul(Wm, J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = m3_mul(m3_mul(m3_t(o.T), m3_t(Vrk)), o.T);
    o.a = cov.c[0][0] + 0.3f;
    o.b = cov.c[0][1];
    o.c = cov.c[1][1] + 0.3f;
    o.t = t;
    return o;
}

// auxiliary.h:46-56; result in tile units, clamped to the grid
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - max_radius) / 16)));
    y0 = min(gy, max(0, (int)((py - max_radius) / 16)));
    x1 = min(gx, max(0, (int)((px + max_radius + 16 - 1) / 16)));
    y1 = min(gy, max(0, (int)((py + max_radius + 16 - 1) / 16)));
}

} // namespace c3dgs
