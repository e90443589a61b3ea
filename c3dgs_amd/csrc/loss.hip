// loss.hip -- fused L1 + SSIM loss of the QAT / sensitivity loops for gfx950.
//
//   reference: utils/loss_utils.py:17-63 (l1_loss, gaussian, create_window, ssim, _ssim) and finetune.py:48
//       loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(ssim_map(x, y)))
//
// The reference runs five grouped 11x11 F.conv2d calls plus ~15 elementwise kernels forward and their autograd
// mirror backward. Here: ONE forward kernel and ONE backward kernel. A forward workgroup owns a 32x22 output tile of one
// channel, stages the 42x32 halo of both images in LDS, applies the separable Gaussian (11 + 11 taps instead of 121)
// to the five moments {x, y, x^2, y^2, xy} and evaluates the SSIM map in registers (both passes slide the window over
// registers: a thread produces 4 horizontal / 3 vertical outputs from 14 / 13 LDS reads); it emits only the three partial
// derivative maps the backward needs (d m / d mu1 | E[x^2],E[xy];  d m / d sigma1^2;  d m / d sigma12). The backward
// convolves those three maps with the same (symmetric) window and adds the L1 sign term:
//   dL/dx_p = g * [ (1-l)/N * sign(x_p - y_p) - l/N * (w*Dmu + 2 x_p (w*Ds1) + y_p (w*Ds12))_p ].
// Both kernels are HBM-bound streaming passes (reads 2 / 5 planes, writes 3 / 1).
#include "common.hpp"
#include <cmath>

namespace c3dgs {

constexpr int LR = 5;             // window radius (window_size 11)
// forward tile: 32 x 22 outputs per 256-thread workgroup. The halo is then 42 x 32: exactly 32 rows x 8 four-column
// groups = 256 horizontal tasks, one per thread; vertically a thread owns 3 consecutive rows of one column, so that both
// passes slide an 11-tap window over registers (14 / 13 LDS reads for 4 / 3 outputs instead of 11 per output).
constexpr int FW = 32, FH = 22, FHALO_W = FW + 2 * LR, FHALO_H = FH + 2 * LR;   // 42 x 32
static_assert(FHALO_H * (FW / 4) == 256, "one horizontal task per thread");

struct GaussWindow { float g[11]; };

// gaussian(11, 1.5) normalised in fp32, as utils/loss_utils.py:23-25 does
static GaussWindow make_window()
{
    GaussWindow w;
    float s = 0.f;
    for (int i = 0; i < 11; i++) { w.g[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.g[i]; }
    for (int i = 0; i < 11; i++) w.g[i] = w.g[i] / s;
    return w;
}

__global__ void __launch_bounds__(256)
l1_ssim_forward_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, const GaussWindow win,
                       float* __restrict__ Dmu, float* __restrict__ Ds1, float* __restrict__ Ds12,
                       double* __restrict__ sums /*[128]: 64 partial sums of |x-y|, then 64 of ssim*/)
{
    __shared__ float s_x[FHALO_H][FHALO_W + 1];
    __shared__ float s_y[FHALO_H][FHALO_W + 1];
    __shared__ float s_h[5][FHALO_H][FW + 1];     // horizontally filtered moments
    __shared__ double s_red[2][4];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FW, y0 = blockIdx.y * FH, c = blockIdx.z;
    const size_t plane = (size_t)c * H * W;

    for (int q = tid; q < FHALO_H * FHALO_W; q += 256) {
        const int r = q / FHALO_W, col = q - r * FHALO_W;
        const int yy = y0 + r - LR, xx = x0 + col - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;          // zero padding (padding=window_size//2)
        s_x[r][col] = in ? img[plane + (size_t)yy * W + xx] : 0.f;
        s_y[r][col] = in ? gt[plane + (size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    {   // horizontal pass: thread = (halo row, group of 4 output columns); 14 inputs feed 4 outputs
        const int r = tid >> 3, cg = (tid & 7) * 4;
        float xv[14], yv[14];
#pragma unroll
        for (int j = 0; j < 14; j++) { xv[j] = s_x[r][cg + j]; yv[j] = s_y[r][cg + j]; }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float g = win.g[k], x = xv[o + k], y = yv[o + k];
                a0 = fmaf(g, x, a0); a1 = fmaf(g, y, a1);
                a2 = fmaf(g, x * x, a2); a3 = fmaf(g, y * y, a3); a4 = fmaf(g, x * y, a4);
            }
            s_h[0][r][cg + o] = a0; s_h[1][r][cg + o] = a1; s_h[2][r][cg + o] = a2; s_h[3][r][cg + o] = a3; s_h[4][r][cg + o] = a4;
        }
    }
    __syncthreads();
    // vertical pass: thread = (column, group of 3 output rows); 13 rows feed 3 outputs
    const int tx = tid & 31, rg = (tid >> 5) * 3;
    double l1v = 0.0, ssv = 0.0;
    float hv[5][13];
#pragma unroll
    for (int m = 0; m < 5; m++)
#pragma unroll
        for (int j = 0; j < 13; j++) hv[m][j] = (rg + j < FHALO_H) ? s_h[m][rg + j][tx] : 0.f;
#pragma unroll
    for (int o = 0; o < 3; o++) {
        const int ty = rg + o;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = win.g[k];
            mu1 = fmaf(g, hv[0][o + k], mu1); mu2 = fmaf(g, hv[1][o + k], mu2);
            e11 = fmaf(g, hv[2][o + k], e11); e22 = fmaf(g, hv[3][o + k], e22);
            e12 = fmaf(g, hv[4][o + k], e12);
        }
        const int px = x0 + tx, py = y0 + ty;
        if (ty < FH && px < W && py < H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;              // loss_utils.py:54-55
            const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2, Cc = mu1 * mu1 + mu2 * mu2 + C1, Dd = s1 + s2 + C2;
            const float inv_cd = 1.0f / (Cc * Dd);
            const float mval = A * B * inv_cd;                                 // loss_utils.py:57
            const float d_s1 = -mval / Dd;                                     // dm/dsigma1^2
            const float d_s12 = 2.f * A * inv_cd;                              // dm/dsigma12
            const float d_mu1 = 2.f * mu2 * B * inv_cd - mval * 2.f * mu1 / Cc;   // dm/dmu1 at fixed sigmas
            const size_t p = plane + (size_t)py * W + px;
            if (Dmu) {
                Dmu[p] = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12;              // ... at fixed E[x^2], E[xy]
                Ds1[p] = d_s1;
                Ds12[p] = d_s12;
            }
            ssv += (double)mval;
            l1v += (double)fabsf(s_x[ty + LR][tx + LR] - s_y[ty + LR][tx + LR]);
        }
    }
    // block reduction -> two double atomics per workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l1v += __shfl_xor(l1v, o); ssv += __shfl_xor(ssv, o); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1v; s_red[1][tid >> 6] = ssv; }
    __syncthreads();
    if (tid == 0) {
        // 64 accumulators per quantity: thousands of workgroups adding into ONE address serialise at the memory side
        // (~12 ns per same-address atomic); spread over 64 addresses the tail is a few microseconds
        const int slot = (blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & 63;
        atomicAdd(&sums[slot], (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
        atomicAdd(&sums[64 + slot], (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    }
}

// backward: same 32 x 22 tiling and register-sliding windows, on the three derivative maps
__global__ void __launch_bounds__(256)
l1_ssim_backward_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, const GaussWindow win,
                        const float* __restrict__ Dmu, const float* __restrict__ Ds1, const float* __restrict__ Ds12,
                        const float* __restrict__ grad_loss /*device scalar*/, float l1_scale, float ssim_scale,
                        float* __restrict__ dL_dimg)
{
    __shared__ float s_m[3][FHALO_H][FHALO_W + 1];
    __shared__ float s_h[3][FHALO_H][FW + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FW, y0 = blockIdx.y * FH, c = blockIdx.z;
    const size_t plane = (size_t)c * H * W;
    for (int q = tid; q < FHALO_H * FHALO_W; q += 256) {
        const int r = q / FHALO_W, col = q - r * FHALO_W;
        const int yy = y0 + r - LR, xx = x0 + col - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t p = plane + (size_t)yy * W + xx;
        s_m[0][r][col] = in ? Dmu[p] : 0.f;
        s_m[1][r][col] = in ? Ds1[p] : 0.f;
        s_m[2][r][col] = in ? Ds12[p] : 0.f;
    }
    __syncthreads();
    {
        const int r = tid >> 3, cg = (tid & 7) * 4;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float v[14];
#pragma unroll
            for (int j = 0; j < 14; j++) v[j] = s_m[m][r][cg + j];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) a = fmaf(win.g[k], v[o + k], a);
                s_h[m][r][cg + o] = a;
            }
        }
    }
    __syncthreads();
    const int tx = tid & 31, rg = (tid >> 5) * 3;
    float hv[3][13];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int j = 0; j < 13; j++) hv[m][j] = (rg + j < FHALO_H) ? s_h[m][rg + j][tx] : 0.f;
    const float gl = grad_loss[0];
#pragma unroll
    for (int o = 0; o < 3; o++) {
        const int ty = rg + o, px = x0 + tx, py = y0 + ty;
        if (ty >= FH || px >= W || py >= H) continue;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = win.g[k];
            a = fmaf(g, hv[0][o + k], a); b = fmaf(g, hv[1][o + k], b); d = fmaf(g, hv[2][o + k], d);
        }
        const size_t p = plane + (size_t)py * W + px;
        const float xv = img[p], yv = gt[p];
        const float sgn = xv > yv ? 1.f : (xv < yv ? -1.f : 0.f);             // d|x-y|/dx as torch.abs' backward (0 at 0)
        dL_dimg[p] = gl * (l1_scale * sgn - ssim_scale * (a + 2.f * xv * b + yv * d));
    }
}

// value = l1_scale * sum(sums[0..63]) + ssim_scale * sum(sums[64..127]) + constant, in float64, stored as float32: the six torch
// kernels (a reduction, two scalings, two additions, a cast) that used to follow the forward, in one 64-thread launch
__global__ void __launch_bounds__(64) l1_ssim_value_kernel(const double* __restrict__ sums, double l1_scale, double ssim_scale,
                                                           double constant, float* __restrict__ out)
{
    double a = sums[threadIdx.x], b = sums[64 + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (threadIdx.x == 0) out[0] = (float)(l1_scale * a + ssim_scale * b + constant);
}

void launch_l1_ssim_value(const double* sums, double l1_scale, double ssim_scale, double constant, float* out, hipStream_t s)
{
    l1_ssim_value_kernel<<<1, 64, 0, s>>>(sums, l1_scale, ssim_scale, constant, out);
}

void launch_l1_ssim_forward(int C, int H, int W, const float* img, const float* gt, float* Dmu, float* Ds1, float* Ds12,
                            double* sums, hipStream_t s)
{
    static const GaussWindow win = make_window();
    const dim3 grid((W + FW - 1) / FW, (H + FH - 1) / FH, C);
    l1_ssim_forward_kernel<<<grid, 256, 0, s>>>(H, W, img, gt, win, Dmu, Ds1, Ds12, sums);
}

void launch_l1_ssim_backward(int C, int H, int W, const float* img, const float* gt, const float* Dmu, const float* Ds1,
                             const float* Ds12, const float* grad_loss, float l1_coeff, float ssim_coeff, float* dL_dimg,
                             hipStream_t s)
{
    static const GaussWindow win = make_window();
    const double n = (double)C * H * W;
    const dim3 grid((W + FW - 1) / FW, (H + FH - 1) / FH, C);
    // kernel computes g * (l1_scale * sgn - ssim_scale * conv): ssim_scale = -ssim_coeff / N
    l1_ssim_backward_kernel<<<grid, 256, 0, s>>>(H, W, img, gt, win, Dmu, Ds1, Ds12, grad_loss, (float)((double)l1_coeff / n),
                                                 (float)(-(double)ssim_coeff / n), dL_dimg);
}

} // namespace c3dgs
