/*
 * c3dgs_hip.h -- C ABI of libc3dgs_hip.so: the MI355X (gfx950) rasterizer + VQ hot path.
 *
 * This is the drop-in boundary for the reference's two native extensions
 *   diff_gaussian_rasterization*._C   (submodules/diff-gaussian-rasterization/ext.cpp:15-21,
 *                                      signatures rasterize_points.h:18-122)
 *   weighted_distance._C              (submodules/weighted_distance/ext.cpp:4-6)
 * restated with plain pointers and sizes: no torch types, no C++ types, no CUDA/HIP types.
 * Every pointer is a DEVICE pointer unless marked "host". `stream` is a hipStream_t passed as
 * void* (NULL = the default stream). Every function returns 0 on success and a non-zero
 * C3DGS_E_* code on failure; c3dgs_last_error() then returns a message (thread-local).
 *
 * "Empty tensor" semantics of the reference (torch.Tensor([]) -> data_ptr()==nullptr,
 * forward.cu:214,250; backward.cu:390,394) are kept: an absent optional input is a NULL pointer.
 */
#ifndef C3DGS_HIP_H
#define C3DGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3DGS_ABI_VERSION 4

enum {
    C3DGS_OK = 0,
    C3DGS_E_INVALID = 1, /* bad shape / missing or contradictory arguments (reference: AT_ERROR)   */
    C3DGS_E_HIP = 2,     /* a HIP runtime or kernel error (reference: CHECK_CUDA under debug=True) */
    C3DGS_E_ALLOC = 3    /* a resize callback returned NULL                                         */
};

/* Scratch buffers are owned by the caller and grown through a callback, exactly like the
 * reference's std::function<char*(size_t)> resize lambdas (rasterize_points.cu:27-33,74-79;
 * rasterizer.h:31-34). The callback must return a device pointer to at least `bytes` bytes,
 * 256-byte aligned, that stays valid until the matching backward has run. */
typedef void* (*c3dgs_resize_fn)(void* user, size_t bytes);

/* Inputs shared by forward and backward (reference: the positional arguments of
 * RasterizeGaussiansCUDA / RasterizeGaussiansIndexedCUDA, rasterize_points.cu:35-56,
 * rasterize_points_indexed.cu:35-59). */
typedef struct c3dgs_raster_params {
    int32_t P;      /* Gaussians passed in                                                   */
    int32_t D;      /* active SH degree (0..3)                                               */
    int32_t M;      /* SH coefficients per row = sh.size(1); 0 when sh is absent             */
    int32_t W, H;   /* image size                                                            */
    int32_t SHS;    /* rows of the SH codebook   (indexed variant; else == P or 0)           */
    int32_t GS;     /* rows of the scale/rotation codebooks (indexed variant; else == P or 0) */
    const float* background;     /* [3]                                                      */
    const float* means3D;        /* [P,3]                                                    */
    const float* sh;             /* [P,M,3] | indexed: [SHS,M,3] | NULL                      */
    const float* colors_precomp; /* [P,3] | NULL        (exactly one of sh / colors_precomp) */
    const float* opacities;      /* [P]                                                      */
    const float* scales;         /* [P,3] | indexed: [GS,3] | NULL                           */
    const float* scale_factors;  /* indexed only: [P]                                        */
    const float* rotations;      /* [P,4] | indexed: [GS,4] | NULL   (r,x,y,z), not normalised */
    const float* cov3D_precomp;  /* [P,6] | NULL   (exactly one of scales+rotations / cov3D)  */
    const int64_t* sh_indices;   /* indexed only: [P]                                        */
    const int64_t* g_indices;    /* indexed only: [P]                                        */
    const float* viewmatrix;     /* [16] world->camera, transposed (m[0],m[4],m[8],m[12] = row 0) */
    const float* projmatrix;     /* [16] viewmatrix @ projection, same layout                */
    const float* campos;         /* [3]                                                      */
    float tan_fovx, tan_fovy;
    float scale_modifier;
    int32_t prefiltered;
    int32_t clamp_color;
    int32_t debug;               /* non-zero: synchronise and check after every stage        */
} c3dgs_raster_params;

/* Gradient outputs (reference: the tensors allocated in RasterizeGaussiansBackward*CUDA,
 * rasterize_points.cu:153-162, rasterize_points_indexed.cu:166-176). The callee writes EVERY
 * element of every non-NULL output (zeros for culled Gaussians), so the caller may pass
 * uninitialised memory. A NULL pointer skips that output. */
typedef struct c3dgs_raster_grads {
    float* dL_dmeans2D;       /* [P,3]  (z component is always 0)                             */
    float* dL_dcolors;        /* [P,3]                                                        */
    float* dL_dopacity;       /* [P]                                                          */
    float* dL_dmeans3D;       /* [P,3]                                                        */
    float* dL_dcov3D;         /* [P,6]                                                        */
    float* dL_dsh;            /* [P,M,3] | indexed: [SHS,M,3] (scatter-added)                 */
    float* dL_dscales;        /* [P,3]   | indexed: [GS,3]    (scatter-added)                 */
    float* dL_dscale_factors; /* indexed only: [P]                                            */
    float* dL_drotations;     /* [P,4]   | indexed: [GS,4]    (scatter-added)                 */
} c3dgs_raster_grads;

/* ---- camera set-up of the reference's autograd wrappers, on the device (DGR-NC .../__init__.py:32-40 quat_to_mat,
 * :19-30 getProjectionMatrix, :152-172 `extrinsic @ getProjectionMatrix(intrinsic)` and `extrinsic.inverse()[3, :3]`).
 * extrinsic_vector = device float[7] (qx, qy, qz, qw, tx, ty, tz), read in stream order -- the matrices always belong
 * to the pose's CURRENT values, whatever wrote them. inv_tan_half_fov* = 1 / tan(FoV / 2) as fp32 (the two
 * intrinsic-dependent entries of getProjectionMatrix). Outputs: viewmatrix[16], projmatrix[16] (both transposed like
 * the reference's), campos[3]. */
int c3dgs_camera_from_pose(const float* extrinsic_vector, float inv_tan_half_fovx, float inv_tan_half_fovy,
                           float* viewmatrix, float* projmatrix, float* campos, void* stream);

/* ---- _C.mark_visible (rasterize_points.cu:202-221 -> rasterizer_impl.cu:54-66,141-149) ---- */
int c3dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       uint8_t* present /*[P] bool*/, void* stream);
/* The same test from the camera's POSE (device pointer to qx, qy, qz, qw, tx, ty, tz: what GaussianRasterizer*.markVisible is
 * handed, DGR-NC __init__.py:937-949): the matrix entries the test reads are formed per thread with the fp32 operations of
 * c3dgs_camera_from_pose, so the flags equal c3dgs_camera_from_pose + c3dgs_mark_visible bit for bit, in one launch. */
int c3dgs_mark_visible_pose(int32_t P, const float* means3D, const float* extrinsic_vector, uint8_t* present /*[P] bool*/,
                            void* stream);

/* ---- _C.rasterize_gaussians (rasterize_points.cu:35-117 -> rasterizer_impl.cu:194-334) ----
 * p->sh_indices, p->g_indices, p->scale_factors must be NULL.
 * out_color [3,H,W] and radii [P] are fully written. *num_rendered (host) receives R. */
int c3dgs_rasterize_gaussians(const c3dgs_raster_params* p,
                              c3dgs_resize_fn geom_resize, void* geom_user,
                              c3dgs_resize_fn binning_resize, void* binning_user,
                              c3dgs_resize_fn image_resize, void* image_user,
                              float* out_color, int32_t* radii, int32_t* num_rendered /*host*/, void* stream);

/* ---- _C.rasterize_gaussians_indexed (rasterize_points_indexed.cu:35-123 -> rasterizer_impl.cu:440-586) */
int c3dgs_rasterize_gaussians_indexed(const c3dgs_raster_params* p,
                                      c3dgs_resize_fn geom_resize, void* geom_user,
                                      c3dgs_resize_fn binning_resize, void* binning_user,
                                      c3dgs_resize_fn image_resize, void* image_user,
                                      float* out_color, int32_t* radii, int32_t* num_rendered /*host*/, void* stream);

/* ---- _C.rasterize_gaussians_backward (rasterize_points.cu:119-200 -> rasterizer_impl.cu:338-435) ----
 * geom/binning/image buffers are the ones the forward filled; R is the forward's num_rendered.
 * `workspace_resize` provides the backward's own scratch (per-instance partial sums). */
int c3dgs_rasterize_gaussians_backward(const c3dgs_raster_params* p, const int32_t* radii,
                                       const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                                       int32_t R, const float* dL_dout_color /*[3,H,W]*/,
                                       c3dgs_resize_fn workspace_resize, void* workspace_user,
                                       const c3dgs_raster_grads* grads, void* stream);

/* ---- _C.rasterize_gaussians_backward_indexed (rasterize_points_indexed.cu:125-218 -> rasterizer_impl.cu:590-697) */
int c3dgs_rasterize_gaussians_backward_indexed(const c3dgs_raster_params* p, const int32_t* radii,
                                               const void* geom_buffer, const void* binning_buffer,
                                               const void* image_buffer, int32_t R, const float* dL_dout_color,
                                               c3dgs_resize_fn workspace_resize, void* workspace_user,
                                               const c3dgs_raster_grads* grads, void* stream);

/* ---- weighted_distance._C.weightedDistance (weighted_distance.cu:46-93) ----
 * coefs [N,K], codebook [C,K] row-major fp32 -> min squared distance [N] and argmin [N] (int64).
 * Exact reference semantics: fp32 k-ordered FMA chain, strict '<' (lowest index wins ties).
 * gather (optional, may be NULL): int64 [N] row indices into coefs, i.e. row n is coefs[gather[n]]
 * (fuses `features[batch]` of compression/vq.py:70). */
int c3dgs_weighted_distance(int64_t N, int32_t C, int32_t K, const float* coefs, const int64_t* gather,
                            const float* codebook, float* out_dist, int64_t* out_idx, void* stream);

/* the same with device scratch `ws` (16-byte aligned, c3dgs_weighted_distance_ws_bytes(N, C, K) bytes; less is legal):
 *   [the codebook scaled by one power of two per call and split into TWO fp16 pieces per value (v = h + l, h = fp16(v),
 *    l = fp16(v - h): 22 significant bits), laid out as matrix-core operand fragments, + the scaled ||c||^2 and the call's
 *    abs-max word][int32 list of the points whose two best candidates the fast search cannot tell apart].
 * With room for the split codebook (K = 48, 12 or 6; C >= 32) the candidate search runs on the fp16 matrix cores
 * (v_mfma_f32_32x32x16_f16) with THREE piece products per multiply (xh*ch + xh*cl + xl*ch; ~3 * 2^-22 relative error per
 * term); without, on the fp32 matrix cores. The scale puts the codebook's largest magnitude into [2^10, 2^11); whatever
 * leaves fp16's range after scaling (points far beyond the codebook's magnitude, inf, NaN) yields inf / NaN scores, fails
 * the margin test and is re-scanned exactly. Either way the winner's distance is recomputed with the reference's k-ordered
 * chain and every point inside the error margin is re-scanned exactly (listed points several per codebook pass): identical
 * results, distances and indices. */
size_t c3dgs_weighted_distance_ws_bytes(int64_t N, int32_t C, int32_t K);
int c3dgs_weighted_distance_ws(int64_t N, int32_t C, int32_t K, const float* coefs, const int64_t* gather,
                               const float* codebook, float* out_dist, int64_t* out_idx, void* ws, size_t ws_bytes,
                               void* stream);

/* diagnostics for tests: scores[n * C + c] = ||c||^2 - 2 x_n.c as the split-fp16 search forms them (fp32 accumulation of the three fp16 piece products,
 * divided back by the call's scale) (K = 48, N <= 256,
 * C >= 32; ws as above), so the error the ambiguity margin must cover can be measured against float64. */
int c3dgs_debug_wd_scores(int64_t N, int32_t C, int32_t K, const float* coefs, const float* codebook, float* scores, void* ws,
                          size_t ws_bytes, float* out_dist, int64_t* out_idx, void* stream);

/* ---- VectorQuantize.update, split at the point where a sharded run all-reduces (compression/vq.py:28-35) ----
 * accumulate: S[k, 0..D) += w_n * x_n ; S[k, D] += w_n for k = idx[n];  *dist_sum += sum_n dist[n] (may be NULL).
 * S [K, D+1] fp32 must be zeroed by the caller (it is the all-reduce payload). */
int c3dgs_vq_accumulate(int64_t B, int32_t K, int32_t D, const float* x, const float* w,
                        const int64_t* gather /*NULL, or row n of x and w is [gather[n]]*/,
                        const int64_t* idx, const float* dist, float* S, double* dist_sum, void* stream);

/* sums: the first half of a Lloyd step in ONE call -- clears S and *dist_sum, assigns the B (gathered) rows
 * (== c3dgs_weighted_distance into dist / idx) and accumulates them (== c3dgs_vq_accumulate).
 * ws / ws_bytes (optional): the scratch of c3dgs_weighted_distance_ws (c3dgs_weighted_distance_ws_bytes(B, K, D)); same
 * results without it. */
int c3dgs_vq_sums(int64_t B, int32_t K, int32_t D, const float* x, const float* w, const int64_t* gather,
                  const float* codebook, float* dist, int64_t* idx, float* S, double* dist_sum, void* ws, size_t ws_bytes,
                  void* stream);

/* ---- one Lloyd step of the loop of vq_features (compression/vq.py:68-77) in FIVE launches instead of eleven ----
 * The same two halves as c3dgs_vq_sums / c3dgs_vq_apply, for a loop that is the only writer of codebook, S and ws between its
 * own steps (step = 0, 1, 2, ... without gaps; ws = c3dgs_weighted_distance_ws_bytes(B, K, D) bytes, 16-byte aligned, kept
 * across the steps):
 *   step_sums(0)   = c3dgs_vq_sums (clears, abs-max + fp16 split of the codebook, search, exact re-scan, accumulation) and arms ws;
 *   step_apply(s)  = c3dgs_vq_apply's update (the same single-rounded operations, bit-identical codebooks) that ALSO leaves what
 *                    step s+1's search needs: the new codebook's fp16 split fragments and scaled ||c||^2 (scaled with the exponent
 *                    of the codebook it read; the new abs-max is gathered for the step after), S cleared, the list counter cleared;
 *   step_sums(s>0) = search + exact re-scan + accumulation only; *dist_sum must be zero on entry (it is added to). The search
 *                    checks the scale it was given against the codebook's true abs-max and sends every point through the exact
 *                    re-scan if the codebook grew 16x or shrank 64x within one update.
 * The min-distance sum (vq.py:71) is folded into the accumulation kernel. Served shapes: D = 48, 12 or 6, K >= 32
 * (c3dgs_vq_step_supported); a sharded run all-reduces S between the two halves exactly as with the unfused pair. */
int c3dgs_vq_step_supported(int32_t K, int32_t D, const float* x, const float* codebook, const void* ws, size_t ws_bytes);
int c3dgs_vq_step_sums(int32_t step, int64_t B, int32_t K, int32_t D, const float* x, const float* w, const int64_t* gather,
                       const float* codebook, float* dist, int64_t* idx, float* S, double* dist_sum, void* ws, size_t ws_bytes,
                       void* stream);
int c3dgs_vq_step_apply(int32_t step, int32_t K, int32_t D, float* S, float* codebook, float* entry_importance, float decay,
                        float alpha, float eps, int32_t scale_normalize, void* ws, size_t ws_bytes, void* stream);

/* apply: entry_importance = decay*entry_importance + alpha*S[:,D];
 *        codebook = decay*codebook + alpha * S[:, :D] / (S[:,D] + eps)        (ema_inplace, vq.py:45-46)
 * then, if scale_normalize (D>=6): codebook /= (cb[:,0]+cb[:,3]+cb[:,5])[:,None]   (vq.py:73-77). */
int c3dgs_vq_apply(int32_t K, int32_t D, const float* S, float* codebook, float* entry_importance,
                   float decay, float alpha, float eps, int32_t scale_normalize, void* stream);

/* ---- batch draws of vq_features: `batch = torch.randint(low=0, high=N, size=[vq_chunk])` on the CPU generator
 * (compression/vq.py:69) ----
 * mt19937_fill (host only, no GPU): continues at::mt19937's stream by n tempered 32-bit outputs. state = the 624 key
 * words, left / next = the generator's counters as torch.get_rng_state() serialises them (left == 1: block exhausted);
 * all three are updated so the caller can write the advanced state back. `out` may be pinned host memory.
 * draws_to_indices: out[i] = (int64) raw[i] % range on the device -- what torch's randint makes of each 32-bit output
 * for range < 2^28 (ATen uniform_int_from_to; larger ranges consume 64 bits per element and are left to torch). */
int c3dgs_mt19937_fill(uint32_t* state, int64_t* left, int64_t* next, uint32_t* out, int64_t n);
int c3dgs_draws_to_indices(int64_t n, int64_t range, const uint32_t* raw, int64_t* out, void* stream);
/* device-side address of a page-locked host buffer, or NULL when it is not mapped for the device: c3dgs_draws_to_indices may then be
 * given that address as `raw` and reads the words over the bus itself (no copy operation in the stream). */
void* c3dgs_host_buffer_device_address(const void* host_ptr);
/* the same fed from (pinned) host memory in one call: copies n raw words to raw_dev on the stream, then converts them. A rank of a
 * sharded run uploads only ITS slice of the batch's draws (raw_host + lo, n = hi - lo). */
int c3dgs_draws_upload(int64_t n, int64_t range, const uint32_t* raw_host, uint32_t* raw_dev, int64_t* out, void* stream);

/* ---- L1 + SSIM loss (reference utils/loss_utils.py:17-63, used at finetune.py:48) ----
 * forward: sums[0..63] add up to sum |img - gt|, sums[64..127] to sum ssim_map (float64, device, zeroed by the callee;
 * 64 partial accumulators each so that the per-workgroup atomics do not serialise on one address); 11x11 Gaussian
 * window sigma 1.5, zero padding, per channel. dmaps (3*C*H*W floats, may be NULL when no backward is needed)
 * receives the three partial-derivative maps the backward consumes.
 * backward: for a scalar  L = l1_coeff * mean|img-gt| + ssim_coeff * mean(ssim_map) + const  (QAT: 1-lambda, -lambda)
 * writes dL_dimg = grad_loss[0] * dL/dimg  ([C,H,W], fully written). */
int c3dgs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float* dmaps,
                          double* sums /*[128]*/, void* stream);
/* out[0] (float32, device) = l1_scale * sum(sums[0..63]) + ssim_scale * sum(sums[64..127]) + constant, evaluated in float64:
 * the scalar the reference assembles with torch arithmetic (finetune.py:48; l1_scale = (1 - lambda) / N, ssim_scale = -lambda / N,
 * constant = lambda), in one launch */
int c3dgs_l1_ssim_value(const double* sums /*[128], from c3dgs_l1_ssim_forward*/, double l1_scale, double ssim_scale,
                        double constant, float* out /*device [1]*/, void* stream);
int c3dgs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, const float* dmaps,
                           const float* grad_loss /*device [1]*/, float l1_coeff, float ssim_coeff, float* dL_dimg,
                           void* stream);

/* ---- Morton ordering (reference GaussianModel._sort_morton, scene/gaussian_model.py:997-1003,
 * mortonEncode :1417-1432).  codes[i] = 63-bit Morton code of xyz[i] (21 bits per axis, axes in ascending-extent order),
 * order = ids sorted stably by code (int64, usable as a torch index). workspace: c3dgs_morton_workspace_bytes(P). */
size_t c3dgs_morton_workspace_bytes(int32_t P);
int c3dgs_morton_order(int32_t P, const float* xyz /*[P,3]*/, int64_t* codes /*[P]*/, int64_t* order /*[P]*/,
                       void* workspace, void* stream);

/* ---- fused Adam step (the optimizer.step() that closes the QAT inner loop, finetune.py:65-66; optimizer set-up
 * scene/gaussian_model.py:296-308: torch.optim.Adam(param groups, lr=0.0, eps=1e-15), no weight decay, no amsgrad).
 * ONE launch updates up to C3DGS_ADAM_MAX_TENSORS tensors with torch's _single_tensor_adam arithmetic; the caller passes
 * per tensor step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) (computed in double, as torch does). */
#define C3DGS_ADAM_MAX_TENSORS 16
typedef struct c3dgs_adam_tensor {
    float* param;            /* [n] updated in place */
    const float* grad;       /* [n]                  */
    float* exp_avg;          /* [n] first moment     */
    float* exp_avg_sq;       /* [n] second moment    */
    int64_t n;
    float step_size;
    float bias_correction2_sqrt;
} c3dgs_adam_tensor;
int c3dgs_adam_step(int32_t n_tensors, const c3dgs_adam_tensor* tensors /*host*/, double beta1, double beta2, double eps, void* stream);

/* ---- sensitivity pass (compress.py:110-113): acc[i] += |g[i]| for n floats, one launch */
int c3dgs_abs_accumulate(int64_t n, const float* g, float* acc, void* stream);

/* ---- extract_rot_scale(to_full_cov(cov)) (utils/splats.py:7-35; compress_covariance, compression/vq.py:186):
 * cov6[n,6] = upper triangle (xx,xy,xz,yy,yz,zz) -> rot[n,4] unit quaternion (r,x,y,z) of the eigenvector frame with
 * determinant +1, scale[n,3] = sqrt of the ascending eigenvalues of cov + 1e-8 I (NaN -> 1e-6). */
int c3dgs_extract_rot_scale(int32_t n, const float* cov6, float* rot, float* scale, void* stream);

/* ---- QAT getters: activation + fake-quant + visibility gathers in front of every raster call -------------------
 * SURVEY.md 8(f) row N1; reference scene/gaussian_model.py:54-77 (activations), :109-118 (seven
 * torch.ao.quantization.FakeQuantize(dtype=qint8) modules = MovingAverageMinMaxObserver, per-tensor affine,
 * [-128,127], averaging constant 0.01), :213-267 (getters), :851-862 (the [visible] gathers of GaussianModel.render),
 * :1405-1414 (FakeQuantizationHalf). The reference spends ~100 small launches and ~20 host syncs per view here
 * (aminmax + `float(scale)` / `int(zero_point)` per module, nonzero per boolean-mask gather); this path keeps the
 * observer state on the device and never reads it back.
 *
 * Observer/quantiser state of ONE FakeQuantize module, resident in device memory (16 bytes): */
typedef struct c3dgs_fq_state {
    float min_val, max_val;   /* MovingAverageMinMaxObserver running range; +inf / -inf before the first batch */
    float scale;              /* max((max(max_val,0) - min(min_val,0)) / 255, eps_f32)                          */
    int32_t zero_point;       /* clamp(-128 - round(min(min_val,0)/scale), -128, 127)                           */
} c3dgs_fq_state;

enum { C3DGS_FQ_OPACITY = 0, C3DGS_FQ_SCALING = 1, C3DGS_FQ_SCALING_FACTOR = 2, C3DGS_FQ_ROTATION = 3,
       C3DGS_FQ_FEATURES_DC = 4, C3DGS_FQ_FEATURES_REST = 5, C3DGS_FQ_COUNT = 6 };

typedef struct c3dgs_qat_params {
    int32_t P, GS, SHS, M;          /* points, geometry codebook rows, colour codebook rows, SH coefficients (M >= 1) */
    const float* xyz;               /* [P,3]   raw _xyz                                            */
    const float* opacity;           /* [P,1]   raw _opacity (pre-sigmoid)                          */
    const float* scaling_factor;    /* [P,1]   raw _scaling_factor (log)                           */
    const float* scaling;           /* [GS,3]  raw _scaling                                        */
    const float* rotation;          /* [GS,4]  raw _rotation                                       */
    const float* features_dc;       /* [SHS,1,3]                                                   */
    const float* features_rest;     /* [SHS,M-1,3] (may be NULL when M == 1)                       */
    c3dgs_fq_state* state;          /* device, [C3DGS_FQ_COUNT]                                    */
    int32_t observer_enabled[6];    /* torch FakeQuantize.observer_enabled per module              */
    int32_t fake_quant_enabled[6];  /* torch FakeQuantize.fake_quant_enabled per module            */
    int32_t half_xyz;               /* 1: xyz_qa = x.half().float() (:1405-1414); 0: identity      */
    float averaging_constant;       /* 0.01                                                        */
} c3dgs_qat_params;

/* Any input pointer may be NULL: that tensor is skipped in every call below.
 * observe: one pass over the raw tensors -> min/max of the ACTIVATED values (sigmoid(opacity), normalize(relu(scaling)),
 *   the rest raw), moving-average update and qparams of every module whose observer is enabled (FakeQuantize.forward,
 *   first half). `workspace` >= c3dgs_qat_workspace_bytes() bytes of device scratch.
 * codebooks: scales_n[GS,3] = fq(normalize(relu(scaling))); rotations[GS,4] = normalize(fq(rotation));
 *   shs[SHS,M,3] = cat(fq_dc(features_dc), fq_rest(features_rest), dim=1)   (get_scaling_normalized,
 *   _rotation_post_activation, _get_features_raw).
 * visible: visible[p] = in_frustum(xyz_qa(xyz[p])) (rasterizer markVisible on get_xyz), rank = exclusive scan of it
 *   (the row of p in every `[visible]`-gathered tensor), *count (device int32) = number visible. visible == NULL input to
 *   points/points_backward means "all rows, rank = identity" (plain getters).
 * points: rows j = rank[p] of means3D = xyz_qa(xyz), opacities = fq(sigmoid(opacity)), scale_factors =
 *   exp(fq(scaling_factor)), sh_indices / g_indices copies (the five boolean-mask gathers of render()).
 * *_backward: straight-through fake-quant masks x activation derivatives, recomputed from the raw tensors and the
 *   state snapshot `state` (the caller passes a copy taken at forward time); P-sized outputs are fully written
 *   (zeros for invisible rows). */
size_t c3dgs_qat_workspace_bytes(void);
int c3dgs_qat_observe(const c3dgs_qat_params* q, void* workspace, void* stream);
int c3dgs_qat_codebooks(const c3dgs_qat_params* q, float* scales_n, float* rotations, float* shs, void* stream);
int c3dgs_qat_codebooks_backward(const c3dgs_qat_params* q, const float* dL_dscales_n, const float* dL_drotations,
                                 const float* dL_dshs, float* dL_dscaling, float* dL_drotation, float* dL_dfeatures_dc,
                                 float* dL_dfeatures_rest, void* stream);
size_t c3dgs_qat_scan_bytes(int32_t P);
int c3dgs_qat_visible(const c3dgs_qat_params* q, const float* viewmatrix, uint8_t* visible /*[P]*/, int32_t* rank /*[P]*/,
                      int32_t* count /*device [1]*/, void* scan_workspace, void* stream);
int c3dgs_qat_points(const c3dgs_qat_params* q, const uint8_t* visible, const int32_t* rank, const int64_t* sh_indices,
                     const int64_t* g_indices, float* means3D /*[V,3]*/, float* opacities /*[V,1]*/,
                     float* scale_factors /*[V,1]*/, int64_t* sh_indices_out /*[V]*/, int64_t* g_indices_out /*[V]*/,
                     void* stream);
int c3dgs_qat_points_backward(const c3dgs_qat_params* q, const uint8_t* visible, const int32_t* rank,
                              const float* dL_dmeans3D /*[V,3]*/, const float* dL_dmeans2D /*[V,3]*/,
                              const float* dL_dopacities /*[V,1]*/, const float* dL_dscale_factors /*[V,1]*/,
                              float* dL_dxyz /*[P,3]*/, float* dL_dscreenspace /*[P,3]*/, float* dL_dopacity /*[P,1]*/,
                              float* dL_dscaling_factor /*[P,1]*/, void* stream);
/* int8 payload of GaussianModel.save_npz (scene/gaussian_model.py:525-617): per tensor
 * torch.quantize_per_tensor(activation(raw), scale, zero_point, qint8).int_repr() with the module's CURRENT state:
 * opacity <- sigmoid, scaling <- normalize(relu) (or exp when scaling_is_exp), rotation <- normalize, the rest raw.
 * NULL outputs (or NULL inputs in `q`) are skipped. */
int c3dgs_qat_quantize(const c3dgs_qat_params* q, int32_t scaling_is_exp, int8_t* opacity /*[P,1]*/, int8_t* scaling /*[GS,3]*/,
                       int8_t* scaling_factor /*[P,1]*/, int8_t* rotation /*[GS,4]*/, int8_t* features_dc /*[SHS,1,3]*/,
                       int8_t* features_rest /*[SHS,M-1,3]*/, void* stream);
/* One stand-alone FakeQuantize module on an arbitrary tensor (the mirror of calling the torch module): observe (if
 * `observe`), then out = fake_quantize_per_tensor_affine(x) (copy when !enabled); backward: dx = g * mask. */
int c3dgs_fake_quantize(int64_t n, const float* x, c3dgs_fq_state* state, int32_t observe, int32_t enabled,
                        float averaging_constant, float* out, void* workspace, void* stream);
int c3dgs_fake_quantize_backward(int64_t n, const float* x, const c3dgs_fq_state* state, int32_t enabled, const float* g,
                                 float* dx, void* stream);

/* ---- introspection (tests and profiling only) ---- */
typedef struct c3dgs_geom_layout {   /* byte offsets into the geometry buffer for P Gaussians */
    size_t total_bytes;
    size_t splat;          /* float4[3*P]: {x, y, conic_a, conic_b} {conic_c, opacity, r, g} {b, bits(instance offset), bits(rect_lo), bits(rect_hi)} */
    size_t depth_keys;     /* uint32[P] bits of the view-space depth (the float itself for a visible Gaussian), 0xFFFFFFFF for
                            * culled ones; tiles_touched is the area of `rects`; the sort payload is the Gaussian id itself  */
    size_t depth_keys_sorted; /* uint32[P]                                                 */
    size_t depth_order;    /* uint32[P] Gaussian ids in (depth, id) order                  */
    size_t sorted_offsets; /* uint16[4*P] tile rectangles in DEPTH order (k-th nearest Gaussian); their areas scanned with depth_base[k>>8] give the emission offsets */
    size_t inst_offset;    /* uint32[P] inclusive scan of tiles_touched in ID order WITHIN each 256-Gaussian group; + block_base[i>>8] = backward slot end of i */
    size_t rects;          /* uint16[4*P]: xmin, ymin, xmax, ymax (tile units)             */
    size_t clamped;        /* uint8[P] bit c set = channel c was clamped                   */
    size_t scan_temp;      /* depth-sort temporary storage (control words + ping/pong buffers) */
    size_t scan_temp_bytes;
    size_t block_base;     /* uint32[ceil(P/256)+1] instances before each 256-Gaussian group; last = num_rendered */
    size_t depth_base;     /* uint32[ceil(P/256)+1] the same for groups of the DEPTH order (where the pairs are emitted) */
} c3dgs_geom_layout;

typedef struct c3dgs_binning_layout { /* byte offsets into the binning buffer for R instances */
    size_t total_bytes;
    size_t keys_unsorted;   /* uint16[R] tile id, emitted in (depth, id) order of the Gaussians */
    size_t values_unsorted; /* uint32[R] Gaussian id                                       */
    size_t keys_sorted;     /* uint16[R] tile id after the stable tile sort                */
    size_t point_list;      /* uint32[R] Gaussian ids sorted by (tile, depth, id)          */
    size_t sort_temp;
    size_t sort_temp_bytes;
} c3dgs_binning_layout;

typedef struct c3dgs_image_layout {   /* byte offsets into the image buffer                  */
    size_t total_bytes;
    size_t final_T;    /* float[W*H]                                                       */
    size_t n_contrib;  /* uint32[W*H]                                                      */
    size_t ranges;     /* uint32[2*T]                                                      */
    size_t tile_used;  /* uint32[T] max n_contrib over the tile's pixels                   */
    size_t tile_order; /* uint32[T] scratch of the backward: its tile schedule (ABI version 4)  */
} c3dgs_image_layout;

/* tests only: the binning stage's stable LSD radix sort (radix_sort.hip) on caller-provided pairs. key_bytes = 2 (tile
 * keys) or 4 (depth keys); bits [0, end_bit) are sorted; ties keep input order. temp >= c3dgs_debug_sort_temp_bytes(). */
size_t c3dgs_debug_sort_temp_bytes(int32_t key_bytes, int64_t n, int32_t end_bit);
int c3dgs_debug_sort_pairs(int32_t key_bytes, int64_t n, int32_t end_bit, const void* keys_in, void* keys_out,
                           const uint32_t* values_in, uint32_t* values_out, void* temp, size_t temp_bytes, void* stream);

/* experiment builds only (radix_sort.hip compiled with -DC3DGS_OS_TIMING): phase time stamps of the last digit pass, 64 tiles x 8
 * stamps of the shader clock; fails in the product build. */
int c3dgs_debug_sort_times(uint64_t* out /*[512], host*/);

/* profiling only: access patterns with a KNOWN byte count, for calibrating the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on this
 * GPU (tools/pmc_calibrate.py -> profiles/r03_pmc_calibration.txt). kind 0: coalesced 16-byte-per-lane read of n x 16 bytes of
 * `table`; 1: n lanes each read the 48-byte record index[i] (three 16-byte loads); 2: the 192-byte row index[i] (twelve); 3: n
 * lanes each store nine floats to the 36-byte slot index[i]. `out`: one word, practically never written. */
int c3dgs_debug_gather_probe(int32_t kind, int64_t n, void* table, const uint32_t* index, uint32_t* out, void* stream);

/* tests / profiling only: lane-efficiency counters of the two blend kernels, accumulated since the last call and cleared by it.
 * out[0..7] forward, out[8..15] backward: { (wave, Gaussian) pairs run, slots incl. list padding, pixel lanes that used the pair,
 * pairs with >= 1 such lane, iterations an 8x4-pixel half-wave unit would run, iterations a 4x4-pixel unit would run,
 * candidate lists walked, forward: lanes hit incl. finished pixels }. All zero unless the library is the "lanes" build
 * variant (render.hip compiled with -DC3DGS_COUNT_LANES; c3dgs_amd/build.py VARIANTS). Synchronises the stream. */
int c3dgs_debug_lane_counters(uint64_t* out /*[16], host*/, void* stream);

int c3dgs_get_geom_layout(int32_t P, c3dgs_geom_layout* out);
int c3dgs_get_binning_layout(int32_t R, int32_t W, int32_t H, c3dgs_binning_layout* out);
int c3dgs_get_image_layout(int32_t W, int32_t H, c3dgs_image_layout* out);
size_t c3dgs_backward_workspace_bytes(int32_t P, int32_t R);

/* ---- optional per-stage timing (HIP events on the caller's stream; used by bench.py's roofline leg) ----
 * c3dgs_profile_enable(1) makes every stage launch record a start/stop event pair; c3dgs_profile_read()
 * synchronises those events, returns one record per stage seen since the last read (count returned, at
 * most `capacity`) and resets the accumulators. Disabled (the default) it costs nothing. */
typedef struct c3dgs_stage_time {
    char name[32];
    double total_ms;
    int64_t count;
} c3dgs_stage_time;
int c3dgs_profile_enable(int on);
/* Restrict the event pairs to ONE stage (by the name c3dgs_profile_read reports, e.g. "render_backward"); NULL or "" =
 * every stage. Every event pair costs a few microseconds of queue time, so a timed region brackets only the kernel
 * it needs. */
int c3dgs_profile_only(const char* stage_name);
int c3dgs_profile_read(c3dgs_stage_time* out, int capacity);

const char* c3dgs_last_error(void);
int c3dgs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* C3DGS_HIP_H */
