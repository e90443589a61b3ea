// loss.hip -- fused L1 + SSIM loss of the QAT / sensitivity loops for gfx950.
//
//   reference: utils/loss_utils.py:17-63 (l1_loss, gaussian, create_window, ssim, _ssim) and finetune.py:48
//       loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(ssim_map(x, y)))
//
// The reference runs five grouped 11x11 F.conv2d calls plus ~15 elementwise kernels forward and their autograd
// mirror backward. Here: ONE forward kernel and ONE backward kernel. A workgroup owns a 16x16 output tile of one
// channel, stages the 26x26 halo of both images in LDS, applies the separable Gaussian (11 + 11 taps instead of 121)
// to the five moments {x, y, x^2, y^2, xy} and evaluates the SSIM map in registers; it emits only the three partial
// derivative maps the backward needs (d m / d mu1 | E[x^2],E[xy];  d m / d sigma1^2;  d m / d sigma12). The backward
// convolves those three maps with the same (symmetric) window and adds the L1 sign term:
//   dL/dx_p = g * [ (1-l)/N * sign(x_p - y_p) - l/N * (w*Dmu + 2 x_p (w*Ds1) + y_p (w*Ds12))_p ].
// Both kernels are HBM-bound streaming passes (reads 2 / 5 planes, writes 3 / 1).
#include "common.hpp"
#include <cmath>

namespace c3dgs {

constexpr int LT = 16;            // output tile
constexpr int LR = 5;             // window radius (window_size 11)
constexpr int LH = LT + 2 * LR;   // 26: staged halo tile

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
    __shared__ float s_x[LH][LH + 1];
    __shared__ float s_y[LH][LH + 1];
    __shared__ float s_h[5][LH][LT + 1];     // horizontally filtered moments
    __shared__ double s_red[2][4];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, c = blockIdx.z;
    const size_t plane = (size_t)c * H * W;

    for (int q = tid; q < LH * LH; q += 256) {
        const int r = q / LH, col = q - r * LH;
        const int yy = y0 + r - LR, xx = x0 + col - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;          // zero padding (padding=window_size//2)
        s_x[r][col] = in ? img[plane + (size_t)yy * W + xx] : 0.f;
        s_y[r][col] = in ? gt[plane + (size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int q = tid; q < LH * LT; q += 256) {                            // horizontal pass: 26 rows x 16 columns
        const int r = q / LT, col = q - r * LT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = win.g[k], xv = s_x[r][col + k], yv = s_y[r][col + k];
            a0 = fmaf(g, xv, a0); a1 = fmaf(g, yv, a1);
            a2 = fmaf(g, xv * xv, a2); a3 = fmaf(g, yv * yv, a3); a4 = fmaf(g, xv * yv, a4);
        }
        s_h[0][r][col] = a0; s_h[1][r][col] = a1; s_h[2][r][col] = a2; s_h[3][r][col] = a3; s_h[4][r][col] = a4;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {                                        // vertical pass
        const float g = win.g[k];
        mu1 = fmaf(g, s_h[0][ty + k][tx], mu1); mu2 = fmaf(g, s_h[1][ty + k][tx], mu2);
        e11 = fmaf(g, s_h[2][ty + k][tx], e11); e22 = fmaf(g, s_h[3][ty + k][tx], e22);
        e12 = fmaf(g, s_h[4][ty + k][tx], e12);
    }
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    double l1v = 0.0, ssv = 0.0;
    if (inside) {
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;              // loss_utils.py:54-55
        const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2, Cc = mu1 * mu1 + mu2 * mu2 + C1, Dd = s1 + s2 + C2;
        const float inv_cd = 1.0f / (Cc * Dd);
        const float m = A * B * inv_cd;                                    // loss_utils.py:57
        const float d_s1 = -m / Dd;                                        // dm/dsigma1^2
        const float d_s12 = 2.f * A * inv_cd;                              // dm/dsigma12
        const float d_mu1 = 2.f * mu2 * B * inv_cd - m * 2.f * mu1 / Cc;   // dm/dmu1 at fixed sigmas
        const size_t p = plane + (size_t)py * W + px;
        if (Dmu) {
            Dmu[p] = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12;              // ... at fixed E[x^2], E[xy]
            Ds1[p] = d_s1;
            Ds12[p] = d_s12;
        }
        ssv = (double)m;
        l1v = (double)fabsf(s_x[ty + LR][tx + LR] - s_y[ty + LR][tx + LR]);
    }
    // block reduction -> two double atomics per workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l1v += __shfl_xor(l1v, o); ssv += __shfl_xor(ssv, o); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1v; s_red[1][tid >> 6] = ssv; }
    __syncthreads();
    if (tid == 0) {
        // 64 accumulators per quantity: ~25 k workgroups adding into ONE address serialise at the memory side
        // (~12 ns per same-address atomic = 0.6 ms); spread over 64 addresses the tail is a few microseconds
        const int slot = (blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & 63;
        atomicAdd(&sums[slot], (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]));
        atomicAdd(&sums[64 + slot], (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
    }
}

__global__ void __launch_bounds__(256)
l1_ssim_backward_kernel(int H, int W, const float* __restrict__ img, const float* __restrict__ gt, const GaussWindow win,
                        const float* __restrict__ Dmu, const float* __restrict__ Ds1, const float* __restrict__ Ds12,
                        const float* __restrict__ grad_loss /*device scalar*/, float l1_scale, float ssim_scale,
                        float* __restrict__ dL_dimg)
{
    __shared__ float s_m[3][LH][LH + 1];
    __shared__ float s_h[3][LH][LT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, c = blockIdx.z;
    const size_t plane = (size_t)c * H * W;
    for (int q = tid; q < LH * LH; q += 256) {
        const int r = q / LH, col = q - r * LH;
        const int yy = y0 + r - LR, xx = x0 + col - LR;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t p = plane + (size_t)yy * W + xx;
        s_m[0][r][col] = in ? Dmu[p] : 0.f;
        s_m[1][r][col] = in ? Ds1[p] : 0.f;
        s_m[2][r][col] = in ? Ds12[p] : 0.f;
    }
    __syncthreads();
    for (int q = tid; q < LH * LT; q += 256) {
        const int r = q / LT, col = q - r * LT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = win.g[k];
            a0 = fmaf(g, s_m[0][r][col + k], a0); a1 = fmaf(g, s_m[1][r][col + k], a1); a2 = fmaf(g, s_m[2][r][col + k], a2);
        }
        s_h[0][r][col] = a0; s_h[1][r][col] = a1; s_h[2][r][col] = a2;
    }
    __syncthreads();
    const int px = x0 + tx, py = y0 + ty;
    if (px >= W || py >= H) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = win.g[k];
        a = fmaf(g, s_h[0][ty + k][tx], a); b = fmaf(g, s_h[1][ty + k][tx], b); d = fmaf(g, s_h[2][ty + k][tx], d);
    }
    const size_t p = plane + (size_t)py * W + px;
    const float xv = img[p], yv = gt[p];
    const float sgn = xv > yv ? 1.f : (xv < yv ? -1.f : 0.f);             // d|x-y|/dx as torch.abs' backward (0 at 0)
    dL_dimg[p] = grad_loss[0] * (l1_scale * sgn - ssim_scale * (a + 2.f * xv * b + yv * d));
}

void launch_l1_ssim_forward(int C, int H, int W, const float* img, const float* gt, float* Dmu, float* Ds1, float* Ds12,
                            double* sums, hipStream_t s)
{
    static const GaussWindow win = make_window();
    const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
    l1_ssim_forward_kernel<<<grid, 256, 0, s>>>(H, W, img, gt, win, Dmu, Ds1, Ds12, sums);
}

void launch_l1_ssim_backward(int C, int H, int W, const float* img, const float* gt, const float* Dmu, const float* Ds1,
                             const float* Ds12, const float* grad_loss, float l1_coeff, float ssim_coeff, float* dL_dimg,
                             hipStream_t s)
{
    static const GaussWindow win = make_window();
    const double n = (double)C * H * W;
    const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
    // kernel computes g * (l1_scale * sgn - ssim_scale * conv): ssim_scale = -ssim_coeff / N
    l1_ssim_backward_kernel<<<grid, 256, 0, s>>>(H, W, img, gt, win, Dmu, Ds1, Ds12, grad_loss, (float)((double)l1_coeff / n),
                                                 (float)(-(double)ssim_coeff / n), dL_dimg);
}

} // namespace c3dgs
