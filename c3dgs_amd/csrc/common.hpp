// common.hpp -- shared host/device helpers for libc3dgs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/c3dgs_hip.h"

namespace c3dgs {

constexpr int TILE = 16;          // reference BLOCK_X = BLOCK_Y = 16 (cuda_rasterizer/config.h:16-17)
constexpr int TILE_PIX = 256;
constexpr int SPLAT_F4 = 3;       // float4 per splat record (48 B)
constexpr int PARTIAL_FLOATS = 9; // dcolor(3) dmean2D(2) dconic(3) dopacity(1) per (Gaussian,tile) instance

// ---------------------------------------------------------------- errors
void set_error(const std::string& msg);
inline int fail(int code, const std::string& msg) { set_error(msg); return code; }

#define C3DGS_HIP_TRY(expr)                                                                         \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return ::c3dgs::fail(C3DGS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));  \
    } while (0)

// after a kernel launch: always catch launch errors; with debug also synchronise (reference CHECK_CUDA,
// cuda_rasterizer/auxiliary.h:168-175)
#define C3DGS_STAGE(name, debug, stream)                                                            \
    do {                                                                                            \
        hipError_t e__ = hipGetLastError();                                                         \
        if (e__ == hipSuccess && (debug)) e__ = hipStreamSynchronize(stream);                       \
        if (e__ != hipSuccess)                                                                      \
            return ::c3dgs::fail(C3DGS_E_HIP, std::string("stage ") + name + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------- scratch layouts
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

size_t scan_temp_bytes(int P);      // max(scan, depth sort) temporary storage for P Gaussians
size_t sort_temp_bytes(int R, int end_bit, int key_bytes = 2);

inline int tiles_x(int W) { return (W + TILE - 1) / TILE; }
inline int tiles_y(int H) { return (H + TILE - 1) / TILE; }

// reference getHigherMsb, rasterizer_impl.cu:35-50
inline uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

inline void geom_layout(int P, c3dgs_geom_layout* L)
{
    size_t o = 0, p = (size_t)(P > 0 ? P : 1);
    L->splat = o;             o = align_up(o + p * SPLAT_F4 * 16);
    L->depth_keys = o;        o = align_up(o + p * 4);
    L->depth_keys_sorted = o; o = align_up(o + p * 4);
    L->depth_order = o;       o = align_up(o + p * 4);
    L->sorted_offsets = o;    o = align_up(o + p * 8);
    L->inst_offset = o;       o = align_up(o + p * 4);
    L->rects = o;             o = align_up(o + p * 8);
    L->clamped = o;           o = align_up(o + p);
    L->block_base = o;        o = align_up(o + ((p + 255) / 256 + 2) * 4);   // + num_rendered + sort error word
    L->depth_base = o;        o = align_up(o + ((p + 255) / 256 + 1) * 4);
    L->scan_temp = o;         L->scan_temp_bytes = scan_temp_bytes((int)p);
    o = align_up(o + L->scan_temp_bytes);
    L->total_bytes = o;
}

// The indexed variant gathers a Gaussian's rotation (16 B) and scale (12 B) from two codebooks by the same random index:
// four gather instructions over up to 256 different cache lines per wave. Packed once per forward into ONE 32-byte row per
// codebook entry {rotation | scale, 0} (stored behind the P-sized part of the geometry buffer, so the backward finds it too),
// the same data arrives with two 16-byte loads from one line per Gaussian (preprocess 0.268 -> see DESIGN.md).
inline size_t geom_gtab_bytes(const c3dgs_raster_params& p) { return (p.g_indices && p.scales) ? align_up((size_t)(p.GS > 0 ? p.GS : 1) * 32) : 0; }

// Tile keys are 16-bit up to 65,536 tiles (4096 x 4096 pixels) and 32-bit above (e.g. 7680 x 4320 = 129,600 tiles): the
// reference's 64-bit keys (rasterizer_impl.cu:98-108) have no tile limit, and neither has this path; the common case keeps
// its 6-byte instances and two digit passes.
inline int tile_key_bytes(int W, int H) { return (long long)tiles_x(W) * tiles_y(H) > 65536 ? 4 : 2; }

inline void binning_layout(int R, int W, int H, c3dgs_binning_layout* L)
{
    size_t o = 0, r = (size_t)(R > 0 ? R : 1);
    int end_bit = (int)higher_msb((uint32_t)(tiles_x(W) * tiles_y(H)));
    const size_t kb = (size_t)tile_key_bytes(W, H);
    L->keys_unsorted = o;   o = align_up(o + r * kb);
    L->values_unsorted = o; o = align_up(o + r * 4);
    L->keys_sorted = o;     o = align_up(o + r * kb);
    L->point_list = o;      o = align_up(o + r * 4);
    L->sort_temp = o;       L->sort_temp_bytes = sort_temp_bytes((int)r, end_bit, (int)kb);
    o = align_up(o + L->sort_temp_bytes);
    L->total_bytes = o;
}

inline void image_layout(int W, int H, c3dgs_image_layout* L)
{
    size_t o = 0, n = (size_t)W * H, t = (size_t)tiles_x(W) * tiles_y(H);
    L->final_T = o;   o = align_up(o + n * 4);
    L->n_contrib = o; o = align_up(o + n * 4);
    L->ranges = o;    o = align_up(o + t * 8);
    L->tile_used = o; o = align_up(o + t * 4);
    L->tile_order = o; o = align_up(o + t * 4);
    L->total_bytes = o;
}

// ---------------------------------------------------------------- kernel launchers (one per .hip file)
struct GeomPtrs {
    float4* splat; uint32_t* depth_keys; uint32_t* depth_keys_sorted;
    uint32_t* depth_order; uint2* sorted_offsets; uint32_t* inst_offset; uint16_t* rects;
    uint8_t* clamped; uint32_t* block_base; uint32_t* depth_base; void* scan_temp; size_t scan_temp_bytes;
    float4* gtab;     // indexed variant: the scale / rotation codebooks packed as one 32-byte row per entry (behind the P-sized part)
};
struct BinPtrs {
    void* keys_unsorted; uint32_t* values_unsorted; void* keys_sorted; uint32_t* point_list;   // keys: u16, or u32 above 65,536 tiles
    void* sort_temp; size_t sort_temp_bytes; int key_bytes;
};
struct ImgPtrs { float* final_T; uint32_t* n_contrib; uint2* ranges; uint32_t* tile_used; uint32_t* tile_order; };

inline GeomPtrs geom_ptrs(void* base, int P)
{
    c3dgs_geom_layout L; geom_layout(P, &L);
    char* b = (char*)base;
    return { (float4*)(b + L.splat), (uint32_t*)(b + L.depth_keys), (uint32_t*)(b + L.depth_keys_sorted), (uint32_t*)(b + L.depth_order),
             (uint2*)(b + L.sorted_offsets), (uint32_t*)(b + L.inst_offset), (uint16_t*)(b + L.rects),
             (uint8_t*)(b + L.clamped), (uint32_t*)(b + L.block_base), (uint32_t*)(b + L.depth_base), (void*)(b + L.scan_temp),
             L.scan_temp_bytes, (float4*)(b + L.total_bytes) };
}
inline BinPtrs bin_ptrs(void* base, int R, int W, int H)
{
    c3dgs_binning_layout L; binning_layout(R, W, H, &L);
    char* b = (char*)base;
    return { (void*)(b + L.keys_unsorted), (uint32_t*)(b + L.values_unsorted), (void*)(b + L.keys_sorted),
             (uint32_t*)(b + L.point_list), (void*)(b + L.sort_temp), L.sort_temp_bytes, tile_key_bytes(W, H) };
}
inline ImgPtrs img_ptrs(void* base, int W, int H)
{
    c3dgs_image_layout L; image_layout(W, H, &L);
    char* b = (char*)base;
    return { (float*)(b + L.final_T), (uint32_t*)(b + L.n_contrib), (uint2*)(b + L.ranges), (uint32_t*)(b + L.tile_used),
             (uint32_t*)(b + L.tile_order) };
}

// preprocess.hip
void launch_camera_from_pose(const float* pose, float inv_tan_x, float inv_tan_y, float* view, float* proj, float* campos, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
void launch_mark_visible_pose(int P, const float* means3D, const float* pose7, uint8_t* present, hipStream_t s);
void launch_pack_codebook(const c3dgs_raster_params& p, float4* gtab, hipStream_t s);
// zero_span / zero_n16: 16-byte words every workgroup clears a slice of before anything else (the depth sort's control words);
// host_out / host_seq: mapped host words {num_rendered, sort error flag, host_seq} the closing scan writes (or null)
void launch_preprocess(const c3dgs_raster_params& p, const GeomPtrs& g, int32_t* radii, uint2* ranges, const uint32_t* sort_err,
                       void* zero_span, size_t zero_n16, uint32_t* host_out, uint32_t host_seq, hipStream_t s);
void launch_depth_order_scan(int P, const GeomPtrs& g, hipStream_t s);   // block totals of tiles_sorted -> depth_base[]
void launch_duplicate_with_keys(int P, const GeomPtrs& g, const BinPtrs& b, int grid_x, const uint32_t* sort_err,
                                void* zero_span, size_t zero_n16, hipStream_t s);   // zero span: the tile sort's control words
void launch_identify_ranges(int R, const void* keys_sorted, int key_bytes, uint2* ranges, const uint32_t* sort_err, hipStream_t s);
// binning.hip
// ctrl_cleared: the first *_sort_clear_bytes(...) bytes of `temp` were zeroed by an earlier kernel on the same stream
// (0 bytes = this configuration's sort clears its own control words: pass false)
// rects_fit_bytes: all rectangle coordinates < 256 (at most 255 x 255 tiles): the rectangles may ride through the sort packed
hipError_t run_depth_sort(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                          uint32_t* vout, int P, const uint2* rects, uint2* rects_sorted, hipStream_t s, bool ctrl_cleared = false,
                          bool rects_fit_bytes = false);
hipError_t run_tile_sort(void* temp, size_t temp_bytes, const void* kin, void* kout, int key_bytes, const uint32_t* vin,
                         uint32_t* vout, int R, int end_bit, hipStream_t s, bool ctrl_cleared = false);
size_t depth_sort_clear_bytes(int P);
size_t tile_sort_clear_bytes(int R, int end_bit, int key_bytes);
// radix_sort.hip (hand-written onesweep; C3DGS_SORT_ROCPRIM=1 selects the rocPRIM path of binning.hip instead)
bool onesweep_enabled();
int onesweep_timed_out(hipStream_t s);   // debug mode only (synchronises): reads + clears the sticky error word
uint32_t* onesweep_error_word();          // device address of the sticky look-back time-out word (0 = fine)
size_t onesweep_depth_temp_bytes(int P);
size_t onesweep_tile_temp_bytes(int R, int end_bit, int key_bytes = 2);
size_t onesweep_depth_clear_bytes(int P);
size_t onesweep_tile_clear_bytes(int R, int end_bit, int key_bytes);
hipError_t onesweep_depth_sort(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                               int P, const uint2* gather_src, uint2* gather_dst, hipStream_t s, bool ctrl_cleared = false,
                               bool rects_fit_bytes = false);
hipError_t onesweep_tile_sort32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                                int R, int end_bit, hipStream_t s, bool ctrl_cleared = false);   // 32-bit tile keys (more than 65,536 tiles)
hipError_t onesweep_tile_sort(void* temp, size_t temp_bytes, const uint16_t* kin, uint16_t* kout, const uint32_t* vin, uint32_t* vout,
                              int R, int end_bit, hipStream_t s, bool ctrl_cleared = false);
// render.hip
int os_read_times(unsigned long long* out512);   // radix_sort.hip, experiment builds with -DC3DGS_OS_TIMING only
int read_lane_counters(unsigned long long* out16, hipStream_t s);   // render.hip; all zero unless built with -DC3DGS_COUNT_LANES
void launch_render_forward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                           const float* bg, float* out_color, uint8_t* qmask, const uint32_t* sort_err, hipStream_t s);
void launch_backward_prep(int W, int H, const ImgPtrs& img, uint32_t* tile_order, void* zero_a, size_t n16_a, void* zero_b,
                          size_t n16_b, hipStream_t s);
void launch_render_backward(int W, int H, const ImgPtrs& img, const uint32_t* point_list, const float4* splat,
                            const uint32_t* block_base, const float* bg, const float* dL_dpix, float* partials,
                            uint8_t* touched, const uint8_t* qmask, const uint32_t* tile_order, void* zero_span, size_t zero_n16,
                            hipStream_t s);
// backward_preprocess.hip
void launch_backward_preprocess(const c3dgs_raster_params& p, const int32_t* radii, const GeomPtrs& g,
                                float* partials, const uint8_t* touched, uint32_t* live_count, uint32_t* live_ids,
                                uint32_t* live_slots, const c3dgs_raster_grads& gr, hipStream_t s);
// vq.hip
int launch_weighted_distance(int64_t N, int C, int K, const float* coefs, const int64_t* gather, const float* codebook,
                             float* out_dist, int64_t* out_idx, hipStream_t s, void* ws = nullptr, size_t ws_bytes = 0, int presplit = -1);
bool wd_presplit_supported(int C, int K, const float* coefs, const float* codebook, const void* ws, size_t ws_bytes);
int launch_vq_apply_split(int K, int D, float* S, float* codebook, float* entry_importance, float decay, float alpha, float eps,
                          int scale_normalize, void* ws, size_t ws_bytes, int parity, hipStream_t s);
void launch_vq_seed_words(int K, int D, void* ws, hipStream_t s);
uint32_t* vq_next_absmax_word(int K, int D, void* ws, int parity);
size_t wd_ws_bytes(int64_t N, int C, int K);
int launch_wd_debug_scores(int64_t N, int C, int K, const float* coefs, const float* codebook, float* scores, void* ws, size_t ws_bytes,
                           float* out_dist, int64_t* out_idx, hipStream_t s);
void launch_vq_accumulate(int64_t B, int K, int D, const float* x, const float* w, const int64_t* gather,
                          const int64_t* idx, const float* dist, float* S, double* dist_sum, hipStream_t s, uint32_t* clear_word = nullptr);
void launch_vq_apply(int K, int D, const float* S, float* codebook, float* entry_importance, float decay, float alpha,
                     float eps, int scale_normalize, hipStream_t s);

// probe.hip (measurement only)
int launch_gather_probe(int kind, size_t n, void* table, const uint32_t* index, uint32_t* out, hipStream_t s);
// encode.hip
size_t morton_workspace_bytes(int P);
int run_morton_order(int P, const float* xyz, int64_t* codes_out, int64_t* order_out, void* workspace, hipStream_t s);
void launch_extract_rot_scale(int n, const float* cov6, float* rot, float* scale, hipStream_t s);
// adam.hip
void launch_adam(int n_tensors, const c3dgs_adam_tensor* tensors, double beta1, double beta2, double eps, hipStream_t s);
void launch_abs_accumulate(int64_t n, const float* g, float* acc, hipStream_t s);
// loss.hip
void launch_l1_ssim_value(const double* sums, double l1_scale, double ssim_scale, double constant, float* out, hipStream_t s);
void launch_l1_ssim_forward(int C, int H, int W, const float* img, const float* gt, float* Dmu, float* Ds1, float* Ds12,
                            double* sums, hipStream_t s);
void launch_l1_ssim_backward(int C, int H, int W, const float* img, const float* gt, const float* Dmu, const float* Ds1,
                             const float* Ds12, const float* grad_loss, float l1_coeff, float ssim_coeff, float* dL_dimg,
                             hipStream_t s);

// qat.hip
size_t qat_workspace_bytes();
size_t qat_scan_bytes(int P);
void launch_qat_observe(const c3dgs_qat_params& q, void* workspace, hipStream_t s);
void launch_qat_codebooks(const c3dgs_qat_params& q, float* scales_n, float* rotations, float* shs, hipStream_t s);
void launch_qat_codebooks_backward(const c3dgs_qat_params& q, const float* g_scales, const float* g_rot, const float* g_shs,
                                   float* d_scaling, float* d_rotation, float* d_dc, float* d_rest, hipStream_t s);
hipError_t run_qat_visible(const c3dgs_qat_params& q, const float* view, uint8_t* visible, int32_t* rank, int32_t* count,
                           void* scan_ws, hipStream_t s);
void launch_qat_points(const c3dgs_qat_params& q, const uint8_t* visible, const int32_t* rank, const int64_t* sh_idx,
                       const int64_t* g_idx, float* means3D, float* opac, float* sfac, int64_t* sh_out, int64_t* g_out,
                       hipStream_t s);
void launch_qat_points_backward(const c3dgs_qat_params& q, const uint8_t* visible, const int32_t* rank, const float* g_m3,
                                const float* g_m2, const float* g_op, const float* g_sf, float* d_xyz, float* d_screen,
                                float* d_op, float* d_sf, hipStream_t s);
void launch_qat_quantize(const c3dgs_qat_params& q, int scaling_exp, int8_t* opacity, int8_t* scaling, int8_t* scaling_factor,
                         int8_t* rotation, int8_t* features_dc, int8_t* features_rest, hipStream_t s);
void launch_fake_quantize(long long n, const float* x, c3dgs_fq_state* state, int observe, int enabled, float c, float* out,
                          void* workspace, hipStream_t s);
void launch_fake_quantize_backward(long long n, const float* x, const c3dgs_fq_state* state, int enabled, const float* g,
                                   float* dx, hipStream_t s);

} // namespace c3dgs
