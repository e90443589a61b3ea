// c_abi.hip -- extern "C" entry points of libc3dgs_hip.so (declared in include/c3dgs_hip.h).
// Stage order follows the reference's Rasterizer::forward / backward (cuda_rasterizer/rasterizer_impl.cu:194-334,
// 338-435, 440-586, 590-697); the stages themselves are the gfx950 kernels in this directory.
#include "common.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace c3dgs {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

// ---- optional per-stage timing with HIP events recorded on the caller's stream (bench.py's roofline leg).
// Disabled by default: zero cost. When enabled, every stage launch is bracketed by two events; nothing
// synchronises until c3dgs_profile_read().
enum Stage { ST_MARK_VISIBLE, ST_PREPROCESS, ST_DEPTH_SORT, ST_SCAN, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_RENDER_FWD, ST_ZERO_PARTIALS,
             ST_RENDER_BWD, ST_BWD_PREPROCESS, ST_WDIST, ST_VQ_ACC, ST_VQ_APPLY, ST_LOSS_FWD, ST_LOSS_BWD,
             ST_QAT_OBSERVE, ST_QAT_CODEBOOKS, ST_QAT_VISIBLE, ST_QAT_POINTS, ST_QAT_POINTS_BWD, ST_QAT_CODEBOOKS_BWD, ST_ADAM, ST_COUNT };
static const char* kStageNames[ST_COUNT] = { "mark_visible", "preprocess", "depth_sort", "scan", "duplicate_with_keys", "sort",
                                             "identify_ranges", "render_forward", "zero_partials", "render_backward",
                                             "backward_preprocess", "weighted_distance", "vq_accumulate", "vq_apply", "l1_ssim_forward",
                                             "l1_ssim_backward", "qat_observe", "qat_codebooks", "qat_visible", "qat_points",
                                             "qat_points_backward", "qat_codebooks_backward", "adam_step" };
struct ProfRec { int stage; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static int g_prof_only = -1;        // >= 0: bracket only this stage (two events per launch of ONE kernel)
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

struct StageTimer {
    bool on; int stage; hipStream_t s; hipEvent_t a{}, b{};
    StageTimer(int stage_, hipStream_t s_) : on(g_prof_on && (g_prof_only < 0 || g_prof_only == stage_)), stage(stage_), s(s_)
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto get = [&](hipEvent_t& e) {
            if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) on = false;
        };
        get(a); get(b);
        if (on) (void)hipEventRecord(a, s);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_recs.push_back({ stage, a, b });
    }
};

// Landing pad of the forward's single device->host read (one per calling thread): 64 bytes of page-locked host memory. When
// the allocation can be MAPPED into the device's address space (coherent, fine-grained: the normal case), the kernel that
// produces num_rendered stores {num_rendered, sort error word, sequence number} straight into it and the host polls the
// sequence number -- no copy command on the stream (a ~4 us launch + a ~6 us bubble per forward). Otherwise (`dev` null):
// a hipMemcpyAsync into it behind an event, as before.
struct HostRead {
    uint32_t* pinned = nullptr;
    uint32_t* dev = nullptr;      // device-side address of `pinned`, or null
    uint32_t seq = 0;
    hipEvent_t ev{};
    HostRead()
    {
        const char* e = std::getenv("C3DGS_HOST_READ_COPY");            // "1": force the copy path (test / diagnosis)
        const bool want_map = !(e && e[0] == '1');
        if (want_map && hipHostMalloc((void**)&pinned, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, pinned, 0) == hipSuccess) dev = (uint32_t*)d;
            std::memset(pinned, 0, 64);
        } else {
            (void)hipGetLastError();
            if (hipHostMalloc((void**)&pinned, 64, hipHostMallocDefault) != hipSuccess) { pinned = nullptr; return; }
        }
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(pinned); pinned = nullptr; dev = nullptr; }
    }
};
static HostRead& host_read()
{
    static thread_local HostRead h;
    return h;
}

static int validate(const c3dgs_raster_params* p, bool indexed, bool is_backward)
{
    if (!p) return fail(C3DGS_E_INVALID, "params is NULL");
    if (p->P < 0 || p->W <= 0 || p->H <= 0) return fail(C3DGS_E_INVALID, "P, W, H must be non-negative / positive");
    if (p->P == 0) return C3DGS_OK;
    if (!p->means3D) return fail(C3DGS_E_INVALID, "means3D must have dimensions (num_points, 3)"); // rasterize_points.cu:58-60
    if (!p->background || !p->viewmatrix || !p->projmatrix || !p->campos)
        return fail(C3DGS_E_INVALID, "background, viewmatrix, projmatrix and campos are required");
    if (!is_backward && !p->opacities) // the backward reads opacities from the geometry buffer, as the reference does
        return fail(C3DGS_E_INVALID, "opacities is required");
    if ((p->sh == nullptr) == (p->colors_precomp == nullptr))
        return fail(C3DGS_E_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    const bool has_sr = p->scales != nullptr && p->rotations != nullptr;
    if ((p->scales != nullptr) != (p->rotations != nullptr) || has_sr == (p->cov3D_precomp != nullptr))
        return fail(C3DGS_E_INVALID, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (p->sh) {
        if (p->D < 0 || p->D > 3) return fail(C3DGS_E_INVALID, "SH degree must be in [0,3]");
        if (p->M < (p->D + 1) * (p->D + 1)) return fail(C3DGS_E_INVALID, "sh has fewer coefficients than the active degree needs");
    }
    if (indexed) {
        if (p->sh && !p->sh_indices) return fail(C3DGS_E_INVALID, "indexed rasterizer: sh_indices is required with sh");
        if (has_sr && (!p->g_indices || !p->scale_factors))
            return fail(C3DGS_E_INVALID, "indexed rasterizer: g_indices and scale_factors are required with scales/rotations");
    } else if (p->sh_indices || p->g_indices || p->scale_factors) {
        return fail(C3DGS_E_INVALID, "non-indexed rasterizer: sh_indices / g_indices / scale_factors must be NULL");
    }
    if (tiles_x(p->W) > 65535 || tiles_y(p->H) > 65535) return fail(C3DGS_E_INVALID, "image too large for 16-bit tile coordinates");
    // Up to 65,536 tiles the tile keys are 16 bits, above 32 bits (common.hpp: tile_key_bytes). The bound that remains: the pair
    // emission divides an instance's number inside its rectangle by the rectangle's width with a multiply-high that is exact
    // while (tiles per row)^2 x (tile rows) < 2^32 -- e.g. 16384 x 16384 pixels = 2^30, 7680 x 4320 = 6.2e7.
    if ((unsigned long long)tiles_x(p->W) * tiles_x(p->W) * tiles_y(p->H) >= (1ull << 32))
        return fail(C3DGS_E_INVALID, "image too large: (tiles per row)^2 x (tile rows) must stay below 2^32");
    return C3DGS_OK;
}

static int forward_impl(const c3dgs_raster_params* pp, bool indexed, c3dgs_resize_fn geom_resize, void* geom_user,
                        c3dgs_resize_fn binning_resize, void* binning_user, c3dgs_resize_fn image_resize, void* image_user,
                        float* out_color, int32_t* radii, int32_t* num_rendered, void* stream_)
{
    if (int rc = validate(pp, indexed, false)) return rc;
    if (!out_color || !num_rendered) return fail(C3DGS_E_INVALID, "out_color and num_rendered are required");
    if (!geom_resize || !binning_resize || !image_resize) return fail(C3DGS_E_INVALID, "resize callbacks are required");
    c3dgs_raster_params p = *pp;
    if (!indexed) { p.sh_indices = nullptr; p.g_indices = nullptr; p.scale_factors = nullptr; }
    hipStream_t s = (hipStream_t)stream_;
    const int P = p.P, W = p.W, H = p.H;
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    *num_rendered = 0;

    c3dgs_image_layout IL; image_layout(W, H, &IL);
    void* img_base = image_resize(image_user, IL.total_bytes);
    if (!img_base) return fail(C3DGS_E_ALLOC, "image buffer allocation failed");
    const ImgPtrs img = img_ptrs(img_base, W, H);

    if (P == 0) { // reference returns zero-filled outputs (rasterize_points.cu:69-70,82)
        C3DGS_HIP_TRY(hipMemsetAsync(out_color, 0, (size_t)3 * W * H * sizeof(float), s));
        C3DGS_HIP_TRY(hipMemsetAsync(img_base, 0, IL.total_bytes, s));
        return C3DGS_OK;
    }
    if (!radii) return fail(C3DGS_E_INVALID, "radii is required");

    c3dgs_geom_layout GL; geom_layout(P, &GL);
    void* geom_base = geom_resize(geom_user, GL.total_bytes + geom_gtab_bytes(p));
    if (!geom_base) return fail(C3DGS_E_ALLOC, "geometry buffer allocation failed");
    const GeomPtrs g = geom_ptrs(geom_base, P);

    // K2 / K2i, with the id-order scan of tiles_touched folded in (per-workgroup offsets + block_base[])
    uint32_t* sort_err = onesweep_error_word();
    if (!sort_err) return fail(C3DGS_E_HIP, "cannot resolve the sort error word");
    HostRead& hr = host_read();
    if (!hr.pinned) return fail(C3DGS_E_HIP, "pinned host buffer allocation failed");
    const uint32_t seq = hr.dev ? ++hr.seq : 0u;
    if (hr.dev && seq == 0u) hr.seq = 1u;                                // (wrap-around: 0 is the pad's initial value)
    const uint32_t want_seq = hr.dev ? hr.seq : 0u;
    // the depth sort's control words are cleared by preprocess's workgroups (0 bytes: that sort clears its own)
    const size_t dclear = depth_sort_clear_bytes(P) <= g.scan_temp_bytes ? depth_sort_clear_bytes(P) : 0;
    { StageTimer t_(ST_PREPROCESS, s);
      if (geom_gtab_bytes(p)) launch_pack_codebook(p, g.gtab, s);
      launch_preprocess(p, g, radii, img.ranges, sort_err, g.scan_temp, dclear / 16, hr.dev, want_seq, s); }
    C3DGS_STAGE("preprocess", p.debug, s);
    // The one device->host read of the forward (K4, num_rendered) is issued as EARLY as its value exists: R is the last
    // entry of block_base[]. The copy lands in pinned memory behind an event while the depth sort and the depth-order
    // scan are already queued, so the GPU keeps working while the host waits, sizes the binning buffer and queues the
    // rest (the reference blocks the stream at this point, rasterizer_impl.cu:279).
    // second word: the device's sticky sort time-out flag as of the start of this call (radix_sort.hip)
    if (!hr.dev) {
        C3DGS_HIP_TRY(hipMemcpyAsync(hr.pinned, g.block_base + (P + 255) / 256, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        C3DGS_HIP_TRY(hipEventRecord(hr.ev, s));
    }
    { StageTimer t_(ST_DEPTH_SORT, s);                                               // binning stage 1: P Gaussians by depth
      C3DGS_HIP_TRY(run_depth_sort(g.scan_temp, g.scan_temp_bytes, g.depth_keys, g.depth_keys_sorted, nullptr, g.depth_order, P,
                                   reinterpret_cast<const uint2*>(g.rects), g.sorted_offsets, s, dclear != 0,
                                   /*rects_fit_bytes=*/gx <= 255 && gy <= 255)); }
    C3DGS_STAGE("depth_sort", p.debug, s);
    if (p.debug && onesweep_timed_out(s)) return fail(C3DGS_E_HIP, "depth sort: look-back timed out");
    { StageTimer t_(ST_SCAN, s); launch_depth_order_scan(P, g, s); }                 // K3, in depth order (two-level)
    C3DGS_STAGE("scan", p.debug, s);
    // Poll instead of sleeping in the driver: on a busy host the wake-up from a blocking event wait can take
    // milliseconds (seen as a 2x slower step with unchanged kernel times); the copy is normally done within ~100 us.
    if (hr.dev) {
        // mapped pad: wait for this call's sequence number. Every ~64k polls the stream is queried as well: if it has drained
        // (or failed) and the number still is not there, the store never became visible -> fetch the two words with a copy
        volatile uint32_t* pad = hr.pinned;
        bool seen = false;
        for (long spins = 0; !seen; spins++) {
            seen = __atomic_load_n(&pad[2], __ATOMIC_ACQUIRE) == want_seq;
            if (!seen && (spins & 0xffff) == 0xffff) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipErrorNotReady) continue;
                if (q != hipSuccess) return fail(C3DGS_E_HIP, std::string("num_rendered read: ") + hipGetErrorString(q));
                seen = __atomic_load_n(&pad[2], __ATOMIC_ACQUIRE) == want_seq;
                if (!seen) {
                    C3DGS_HIP_TRY(hipMemcpyAsync(hr.pinned, g.block_base + (P + 255) / 256, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    C3DGS_HIP_TRY(hipStreamSynchronize(s));
                    seen = true;
                }
            }
        }
    } else {
        hipError_t q = hipErrorNotReady;
        for (long spins = 0; spins < 20000000L && (q = hipEventQuery(hr.ev)) == hipErrorNotReady; spins++) { }
        if (q == hipErrorNotReady) q = hipEventSynchronize(hr.ev);
        if (q != hipSuccess) return fail(C3DGS_E_HIP, std::string("num_rendered read: ") + hipGetErrorString(q));
    }
    const uint32_t R_u = hr.pinned[0];
    if (hr.pinned[1] != 0) {
        // A radix-sort look-back timed out in an EARLIER rasterizer call on this device (that call's image was poisoned with
        // NaN by render_forward). Not silent outside debug mode: fail here, at the forward's one natural host read.
        C3DGS_HIP_TRY(hipMemsetAsync(sort_err, 0, sizeof(uint32_t), s));
        return fail(C3DGS_E_HIP, "radix sort look-back timed out in an earlier rasterizer call on this device: that call's "
                                 "image is NaN and its gradients are invalid (flag cleared, this call was not run)");
    }
    if (R_u > 0x7fffffffu) return fail(C3DGS_E_INVALID, "num_rendered overflows int32");
    const int R = (int)R_u;
    *num_rendered = R;

    c3dgs_binning_layout BL; binning_layout(R, W, H, &BL);
    void* bin_base = binning_resize(binning_user, BL.total_bytes);
    if (!bin_base) return fail(C3DGS_E_ALLOC, "binning buffer allocation failed");
    const BinPtrs b = bin_ptrs(bin_base, R, W, H);

    if (R > 0) {
        const int end_bit = (int)higher_msb((uint32_t)T);                           // tile bits only (rasterizer_impl.cu:298)
        // the tile sort's control words are cleared by the pair emission's workgroups (0 bytes: that sort clears its own)
        size_t tclear = tile_sort_clear_bytes(R, end_bit, b.key_bytes);
        if (tclear > b.sort_temp_bytes) tclear = 0;
        { StageTimer t_(ST_DUPLICATE, s); launch_duplicate_with_keys(P, g, b, gx, sort_err, b.sort_temp, tclear / 16, s); } // K5
        C3DGS_STAGE("duplicate_with_keys", p.debug, s);
        { StageTimer t_(ST_SORT, s);
          C3DGS_HIP_TRY(run_tile_sort(b.sort_temp, b.sort_temp_bytes, b.keys_unsorted, b.keys_sorted, b.key_bytes, b.values_unsorted,
                                      b.point_list, R, end_bit, s, tclear != 0)); }  // K6, binning stage 2
        C3DGS_STAGE("sort", p.debug, s);
        if (p.debug && onesweep_timed_out(s)) return fail(C3DGS_E_HIP, "tile sort: look-back timed out");
        { StageTimer t_(ST_RANGES, s); launch_identify_ranges(R, b.keys_sorted, b.key_bytes, img.ranges, sort_err, s); } // K8
        C3DGS_STAGE("identify_ranges", p.debug, s);
    }
    { StageTimer t_(ST_RENDER_FWD, s);
      // per-instance quadrant masks go to the (now idle) sort scratch: R bytes the backward reads instead of recomputing
      launch_render_forward(W, H, img, b.point_list, g.splat, p.background, out_color, (uint8_t*)b.sort_temp, sort_err, s); } // K9
    C3DGS_STAGE("render_forward", p.debug, s);
    return C3DGS_OK;
}

static int backward_impl(const c3dgs_raster_params* pp, bool indexed, const int32_t* radii, const void* geom_buffer,
                         const void* binning_buffer, const void* image_buffer, int32_t R, const float* dL_dout_color,
                         c3dgs_resize_fn ws_resize, void* ws_user, const c3dgs_raster_grads* grads, void* stream_)
{
    if (int rc = validate(pp, indexed, true)) return rc;
    if (!grads) return fail(C3DGS_E_INVALID, "grads is NULL");
    c3dgs_raster_params p = *pp;
    if (!indexed) { p.sh_indices = nullptr; p.g_indices = nullptr; p.scale_factors = nullptr; }
    hipStream_t s = (hipStream_t)stream_;
    const int P = p.P, W = p.W, H = p.H;

    // codebook-sized outputs of the indexed variant are scatter-added: zero them here
    void* cb_zero = nullptr;           // one 16-byte aligned span of whole 16-byte words: cleared by the prep kernel below
    size_t cb_zero16 = 0;
    if (indexed) {
        const size_t n_sh = (grads->dL_dsh && p.sh) ? (size_t)p.SHS * p.M * 3 * sizeof(float) : 0;
        const size_t n_rot = (grads->dL_drotations && p.scales) ? (size_t)p.GS * 4 * sizeof(float) : 0;
        const size_t n_sc = (grads->dL_dscales && p.scales) ? (size_t)p.GS * 3 * sizeof(float) : 0;
        char* a_sh = (char*)grads->dL_dsh; char* a_rot = (char*)grads->dL_drotations; char* a_sc = (char*)grads->dL_dscales;
        if (n_sh && n_rot && n_sc && a_rot == a_sh + n_sh && a_sc == a_rot + n_rot) {
            // the three tensors are carved from one allocation (c3dgs_amd/rasterizer.py does that): one fill
            const size_t n = n_sh + n_rot + n_sc;
            if (p.P > 0 && R >= 0 && ((uintptr_t)a_sh & 15u) == 0) {
                cb_zero = a_sh; cb_zero16 = n / 16;
                if (n % 16) C3DGS_HIP_TRY(hipMemsetAsync(a_sh + cb_zero16 * 16, 0, n % 16, s));
            } else {
                C3DGS_HIP_TRY(hipMemsetAsync(a_sh, 0, n, s));
            }
        } else {
            if (n_sh) C3DGS_HIP_TRY(hipMemsetAsync(a_sh, 0, n_sh, s));
            if (n_sc) C3DGS_HIP_TRY(hipMemsetAsync(a_sc, 0, n_sc, s));
            if (n_rot) C3DGS_HIP_TRY(hipMemsetAsync(a_rot, 0, n_rot, s));
        }
    }
    if (P == 0) return C3DGS_OK;
    if (!radii || !geom_buffer || !image_buffer || !dL_dout_color || (R > 0 && !binning_buffer))
        return fail(C3DGS_E_INVALID, "radii, forward buffers and dL_dout_color are required");
    if (!ws_resize) return fail(C3DGS_E_INVALID, "workspace callback is required");
    if (R < 0) return fail(C3DGS_E_INVALID, "R must be >= 0");

    const GeomPtrs g = geom_ptrs(const_cast<void*>(geom_buffer), P);
    const ImgPtrs img = img_ptrs(const_cast<void*>(image_buffer), W, H);
    const size_t ws_bytes = c3dgs_backward_workspace_bytes(P, R);
    float* partials = (float*)ws_resize(ws_user, ws_bytes);
    if (!partials) return fail(C3DGS_E_ALLOC, "backward workspace allocation failed");

    // only the 1-byte "written" flags are cleared (R bytes, not 36 R): the blend kernel never visits the instances
    // behind each tile's saturation point (73 % of them on the bench scene) and the per-Gaussian kernel skips them.
    // Workspace: partial sums | flags | tile schedule | per-workgroup lists of the blended Gaussians (id, slot) and their lengths
    const size_t r1 = (size_t)(R > 0 ? R : 1), p256 = ((size_t)P + 1023) / 1024 * 1024;   // whole lists (<= 1024 entries each)
    uint8_t* touched = (uint8_t*)partials + align_up(r1 * PARTIAL_FLOATS * sizeof(float));
    uint32_t* tile_order = img.tile_order;                     // T words of the image buffer (scratch of this call)
    uint32_t* live_ids = (uint32_t*)(touched + align_up(r1)) + 65536;   // (the 256 KB in front are the pre-version-4 home of the tile schedule)
    uint32_t* live_slots = live_ids + p256;
    uint32_t* live_count = live_slots + p256;
    // one launch: tile schedule of the blend kernel + the flag clear. The codebook-gradient clear (80 MB on the bench view, needed
    // only by the per-Gaussian kernel's scatter-adds) rides in the blend kernel itself, a slice per tile workgroup: that kernel is
    // bound by vector issue and leaves the memory system idle (without a blend launch the clear stays here)
    { StageTimer t_(ST_ZERO_PARTIALS, s);
      launch_backward_prep(R > 0 ? W : 0, H, img, tile_order, touched, align_up(r1) / 16, R > 0 ? nullptr : cb_zero,
                           R > 0 ? 0 : cb_zero16, s); }
    if (R > 0) {
        const BinPtrs b = bin_ptrs(const_cast<void*>(binning_buffer), R, W, H);
        { StageTimer t_(ST_RENDER_BWD, s);
          launch_render_backward(W, H, img, b.point_list, g.splat, g.block_base, p.background, dL_dout_color, partials, touched,
                                 (const uint8_t*)b.sort_temp, tile_order, cb_zero, cb_zero16, s); } // K10
        C3DGS_STAGE("render_backward", p.debug, s);
    }
    { StageTimer t_(ST_BWD_PREPROCESS, s);
      launch_backward_preprocess(p, radii, g, partials, touched, live_count, live_ids, live_slots, *grads, s); } // K11 + K12(i)
    C3DGS_STAGE("backward_preprocess", p.debug, s);
    return C3DGS_OK;
}

} // namespace c3dgs

using namespace c3dgs;

extern "C" {

const char* c3dgs_last_error(void) { return g_last_error.c_str(); }

int c3dgs_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return C3DGS_OK;
}

int c3dgs_profile_only(const char* stage_name)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_only = -1;
    if (!stage_name || !*stage_name) return C3DGS_OK;
    for (int i = 0; i < ST_COUNT; i++)
        if (std::strcmp(stage_name, kStageNames[i]) == 0) { g_prof_only = i; return C3DGS_OK; }
    return fail(C3DGS_E_INVALID, std::string("profile_only: unknown stage ") + stage_name);
}

int c3dgs_profile_read(c3dgs_stage_time* out, int capacity)
{
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_prof_recs);
    }
    double total[ST_COUNT] = { 0 };
    long long count[ST_COUNT] = { 0 };
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            total[r.stage] += (double)ms;
            count[r.stage]++;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (auto& r : recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    }
    int n = 0;
    for (int i = 0; i < ST_COUNT && n < capacity; i++) {
        if (!count[i]) continue;
        std::memset(&out[n], 0, sizeof(out[n]));
        std::strncpy(out[n].name, kStageNames[i], sizeof(out[n].name) - 1);
        out[n].total_ms = total[i];
        out[n].count = count[i];
        n++;
    }
    return n;
}
int c3dgs_abi_version(void) { return C3DGS_ABI_VERSION; }

int c3dgs_debug_sort_times(uint64_t* out /*[512]*/)
{
    unsigned long long v[512];
    if (!out || os_read_times(v)) return fail(C3DGS_E_INVALID, "debug_sort_times: not a C3DGS_OS_TIMING build");
    for (int i = 0; i < 512; i++) out[i] = (uint64_t)v[i];
    return C3DGS_OK;
}

int c3dgs_debug_gather_probe(int32_t kind, int64_t n, void* table, const uint32_t* index, uint32_t* out, void* stream)
{
    if (n < 0 || !table || !out || (kind != 0 && !index)) return fail(C3DGS_E_INVALID, "debug_gather_probe: bad arguments");
    if (launch_gather_probe(kind, (size_t)n, table, index, out, (hipStream_t)stream)) return fail(C3DGS_E_INVALID, "debug_gather_probe: kind must be 0..3");
    return C3DGS_OK;
}

int c3dgs_debug_lane_counters(uint64_t* out, void* stream)
{
    if (!out) return fail(C3DGS_E_INVALID, "debug_lane_counters: NULL buffer");
    unsigned long long v[16];
    if (read_lane_counters(v, (hipStream_t)stream)) return fail(C3DGS_E_HIP, "debug_lane_counters: copy failed");
    for (int i = 0; i < 16; i++) out[i] = (uint64_t)v[i];
    return C3DGS_OK;
}

size_t c3dgs_debug_sort_temp_bytes(int32_t key_bytes, int64_t n, int32_t end_bit)
{
    if (n <= 0 || n > 0x3fffffff) return 256;
    return key_bytes == 2 ? onesweep_tile_temp_bytes((int)n, end_bit) : onesweep_depth_temp_bytes((int)n);
}

int c3dgs_debug_sort_pairs(int32_t key_bytes, int64_t n, int32_t end_bit, const void* keys_in, void* keys_out,
                           const uint32_t* values_in, uint32_t* values_out, void* temp, size_t temp_bytes, void* stream)
{
    if ((key_bytes != 2 && key_bytes != 4) || n < 0 || n > 0x3fffffff || end_bit < 1 || end_bit > 8 * key_bytes)
        return fail(C3DGS_E_INVALID, "debug_sort_pairs: bad arguments");
    if (key_bytes == 4 && end_bit != 32) return fail(C3DGS_E_INVALID, "debug_sort_pairs: 4-byte keys are sorted on all 32 bits");
    if (n == 0) return C3DGS_OK;
    if (!keys_in || !keys_out || !values_in || !values_out || !temp) return fail(C3DGS_E_INVALID, "debug_sort_pairs: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    if (key_bytes == 2)
        C3DGS_HIP_TRY(onesweep_tile_sort(temp, temp_bytes, (const uint16_t*)keys_in, (uint16_t*)keys_out, values_in, values_out, (int)n,
                                         end_bit, s));
    else
        C3DGS_HIP_TRY(onesweep_depth_sort(temp, temp_bytes, (const uint32_t*)keys_in, (uint32_t*)keys_out, values_in, values_out, (int)n,
                                          nullptr, nullptr, s));
    C3DGS_STAGE("debug_sort_pairs", 1, s);
    if (onesweep_timed_out(s)) return fail(C3DGS_E_HIP, "debug_sort_pairs: look-back timed out");
    return C3DGS_OK;
}

int c3dgs_get_geom_layout(int32_t P, c3dgs_geom_layout* out)
{
    if (!out || P < 0) return fail(C3DGS_E_INVALID, "bad arguments");
    geom_layout(P, out);
    return C3DGS_OK;
}
int c3dgs_get_binning_layout(int32_t R, int32_t W, int32_t H, c3dgs_binning_layout* out)
{
    if (!out || R < 0 || W <= 0 || H <= 0) return fail(C3DGS_E_INVALID, "bad arguments");
    binning_layout(R, W, H, out);
    return C3DGS_OK;
}
int c3dgs_get_image_layout(int32_t W, int32_t H, c3dgs_image_layout* out)
{
    if (!out || W <= 0 || H <= 0) return fail(C3DGS_E_INVALID, "bad arguments");
    image_layout(W, H, out);
    return C3DGS_OK;
}
size_t c3dgs_backward_workspace_bytes(int32_t P, int32_t R)
{
    const size_t r = (size_t)(R > 0 ? R : 1), p256 = ((size_t)(P > 0 ? P : 1) + 1023) / 1024 * 1024;
    // partial sums + 1-byte written flags + the backward's tile schedule (at most 65536 tiles) + the per-workgroup lists of
    // blended Gaussians (id, slot of its sums: 256 entries per workgroup) + their lengths
    return align_up(r * PARTIAL_FLOATS * sizeof(float)) + align_up(r) + 65536 * sizeof(uint32_t) +
           (2 * p256 + p256 / 256) * sizeof(uint32_t) + 256;
}

int c3dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                       void* stream)
{
    (void)projmatrix; // the reference's in_frustum only tests view-space z (auxiliary.h:156)
    if (P < 0) return fail(C3DGS_E_INVALID, "P must be >= 0");
    if (P == 0) return C3DGS_OK;
    if (!means3D || !viewmatrix || !present) return fail(C3DGS_E_INVALID, "means3D, viewmatrix and present are required");
    { StageTimer t_(ST_MARK_VISIBLE, (hipStream_t)stream); launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream); }
    C3DGS_STAGE("mark_visible", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_mark_visible_pose(int32_t P, const float* means3D, const float* extrinsic_vector, uint8_t* present, void* stream)
{
    if (P < 0) return fail(C3DGS_E_INVALID, "P must be >= 0");
    if (P == 0) return C3DGS_OK;
    if (!means3D || !extrinsic_vector || !present) return fail(C3DGS_E_INVALID, "means3D, extrinsic_vector and present are required");
    { StageTimer t_(ST_MARK_VISIBLE, (hipStream_t)stream); launch_mark_visible_pose(P, means3D, extrinsic_vector, present, (hipStream_t)stream); }
    C3DGS_STAGE("mark_visible", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_camera_from_pose(const float* extrinsic_vector, float inv_tan_half_fovx, float inv_tan_half_fovy, float* viewmatrix,
                           float* projmatrix, float* campos, void* stream)
{
    if (!extrinsic_vector || !viewmatrix || !projmatrix || !campos) return fail(C3DGS_E_INVALID, "camera_from_pose: NULL pointer");
    launch_camera_from_pose(extrinsic_vector, inv_tan_half_fovx, inv_tan_half_fovy, viewmatrix, projmatrix, campos, (hipStream_t)stream);
    C3DGS_STAGE("camera_from_pose", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_rasterize_gaussians(const c3dgs_raster_params* p, c3dgs_resize_fn geom_resize, void* geom_user,
                              c3dgs_resize_fn binning_resize, void* binning_user, c3dgs_resize_fn image_resize,
                              void* image_user, float* out_color, int32_t* radii, int32_t* num_rendered, void* stream)
{
    return forward_impl(p, false, geom_resize, geom_user, binning_resize, binning_user, image_resize, image_user, out_color,
                        radii, num_rendered, stream);
}

int c3dgs_rasterize_gaussians_indexed(const c3dgs_raster_params* p, c3dgs_resize_fn geom_resize, void* geom_user,
                                      c3dgs_resize_fn binning_resize, void* binning_user, c3dgs_resize_fn image_resize,
                                      void* image_user, float* out_color, int32_t* radii, int32_t* num_rendered, void* stream)
{
    return forward_impl(p, true, geom_resize, geom_user, binning_resize, binning_user, image_resize, image_user, out_color,
                        radii, num_rendered, stream);
}

int c3dgs_rasterize_gaussians_backward(const c3dgs_raster_params* p, const int32_t* radii, const void* geom_buffer,
                                       const void* binning_buffer, const void* image_buffer, int32_t R,
                                       const float* dL_dout_color, c3dgs_resize_fn workspace_resize, void* workspace_user,
                                       const c3dgs_raster_grads* grads, void* stream)
{
    return backward_impl(p, false, radii, geom_buffer, binning_buffer, image_buffer, R, dL_dout_color, workspace_resize,
                         workspace_user, grads, stream);
}

int c3dgs_rasterize_gaussians_backward_indexed(const c3dgs_raster_params* p, const int32_t* radii, const void* geom_buffer,
                                               const void* binning_buffer, const void* image_buffer, int32_t R,
                                               const float* dL_dout_color, c3dgs_resize_fn workspace_resize,
                                               void* workspace_user, const c3dgs_raster_grads* grads, void* stream)
{
    return backward_impl(p, true, radii, geom_buffer, binning_buffer, image_buffer, R, dL_dout_color, workspace_resize,
                         workspace_user, grads, stream);
}

int c3dgs_weighted_distance(int64_t N, int32_t C, int32_t K, const float* coefs, const int64_t* gather,
                            const float* codebook, float* out_dist, int64_t* out_idx, void* stream)
{
    if (N < 0 || C < 0 || K <= 0) return fail(C3DGS_E_INVALID, "coefs and codebook must have same number of channels");
    if (N == 0) return C3DGS_OK;
    if (!coefs || !codebook || !out_dist || !out_idx) return fail(C3DGS_E_INVALID, "ceofs and codebook must have dimension 2");
    {
        StageTimer t_(ST_WDIST, (hipStream_t)stream);
        if (launch_weighted_distance(N, C, K, coefs, gather, codebook, out_dist, out_idx, (hipStream_t)stream))
            return fail(C3DGS_E_INVALID, "unsupported channel count");
    }
    C3DGS_STAGE("weighted_distance", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

size_t c3dgs_weighted_distance_ws_bytes(int64_t N, int32_t C, int32_t K) { return wd_ws_bytes(N, C, K); }

int c3dgs_weighted_distance_ws(int64_t N, int32_t C, int32_t K, const float* coefs, const int64_t* gather, const float* codebook,
                               float* out_dist, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream)
{
    if (N < 0 || C < 0 || K <= 0) return fail(C3DGS_E_INVALID, "coefs and codebook must have same number of channels");
    if (N == 0) return C3DGS_OK;
    if (!coefs || !codebook || !out_dist || !out_idx) return fail(C3DGS_E_INVALID, "ceofs and codebook must have dimension 2");
    {
        StageTimer t_(ST_WDIST, (hipStream_t)stream);
        if (launch_weighted_distance(N, C, K, coefs, gather, codebook, out_dist, out_idx, (hipStream_t)stream, ws, ws ? ws_bytes : 0))
            return fail(C3DGS_E_INVALID, "unsupported channel count");
    }
    C3DGS_STAGE("weighted_distance", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_vq_accumulate(int64_t B, int32_t K, int32_t D, const float* x, const float* w, const int64_t* gather,
                        const int64_t* idx, const float* dist, float* S, double* dist_sum, void* stream)
{
    if (B < 0 || K <= 0 || D <= 0) return fail(C3DGS_E_INVALID, "bad sizes");
    if (B == 0) return C3DGS_OK;
    if (!x || !w || !idx || !S) return fail(C3DGS_E_INVALID, "x, w, idx and S are required");
    { StageTimer t_(ST_VQ_ACC, (hipStream_t)stream); launch_vq_accumulate(B, K, D, x, w, gather, idx, dist, S, dist_sum, (hipStream_t)stream); }
    C3DGS_STAGE("vq_accumulate", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_vq_sums(int64_t B, int32_t K, int32_t D, const float* x, const float* w, const int64_t* gather,
                  const float* codebook, float* dist, int64_t* idx, float* S, double* dist_sum, void* ws, size_t ws_bytes,
                  void* stream)
{
    if (B < 0 || K <= 0 || D <= 0) return fail(C3DGS_E_INVALID, "bad sizes");
    if (!S || !dist_sum) return fail(C3DGS_E_INVALID, "S and dist_sum are required");
    C3DGS_HIP_TRY(hipMemsetAsync(S, 0, (size_t)K * (D + 1) * sizeof(float), (hipStream_t)stream));
    C3DGS_HIP_TRY(hipMemsetAsync(dist_sum, 0, sizeof(double), (hipStream_t)stream));
    if (B == 0) return C3DGS_OK;
    if (!x || !w || !codebook || !dist || !idx) return fail(C3DGS_E_INVALID, "x, w, codebook, dist and idx are required");
    {
        StageTimer t_(ST_WDIST, (hipStream_t)stream);
        if (launch_weighted_distance(B, K, D, x, gather, codebook, dist, idx, (hipStream_t)stream, ws, ws ? ws_bytes : 0))
            return fail(C3DGS_E_INVALID, "unsupported channel count");
    }
    C3DGS_STAGE("weighted_distance", 0, (hipStream_t)stream);
    return c3dgs_vq_accumulate(B, K, D, x, w, gather, idx, dist, S, dist_sum, stream);
}

int c3dgs_vq_step_supported(int32_t K, int32_t D, const float* x, const float* codebook, const void* ws, size_t ws_bytes)
{
    return (K > 0 && D > 0 && wd_presplit_supported(K, D, x, codebook, ws, ws_bytes)) ? 1 : 0;
}

int c3dgs_vq_step_sums(int32_t step, int64_t B, int32_t K, int32_t D, const float* x, const float* w, const int64_t* gather,
                       const float* codebook, float* dist, int64_t* idx, float* S, double* dist_sum, void* ws, size_t ws_bytes,
                       void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (step < 0 || B <= 0 || K <= 0 || D <= 0) return fail(C3DGS_E_INVALID, "vq_step_sums: bad sizes");
    if (!x || !w || !codebook || !dist || !idx || !S || !dist_sum) return fail(C3DGS_E_INVALID, "vq_step_sums: NULL argument");
    if (!wd_presplit_supported(K, D, x, codebook, ws, ws_bytes))
        return fail(C3DGS_E_INVALID, "vq_step_sums: shape / scratch not served by the fused step (see c3dgs_vq_step_supported)");
    if (step == 0) {            // nothing is prepared yet: the classic first half, then arm the scratch for c3dgs_vq_step_apply
        C3DGS_HIP_TRY(hipMemsetAsync(S, 0, (size_t)K * (D + 1) * sizeof(float), s));
        C3DGS_HIP_TRY(hipMemsetAsync(dist_sum, 0, sizeof(double), s));
    }
    {
        StageTimer t_(ST_WDIST, s);
        if (launch_weighted_distance(B, K, D, x, gather, codebook, dist, idx, s, ws, ws_bytes, step == 0 ? -1 : (step & 1)))
            return fail(C3DGS_E_INVALID, "vq_step_sums: unsupported shape");
        if (step == 0) launch_vq_seed_words(K, D, ws, s);
    }
    C3DGS_STAGE("weighted_distance", 0, s);
    { StageTimer t_(ST_VQ_ACC, s);
      launch_vq_accumulate(B, K, D, x, w, gather, idx, dist, S, dist_sum, s, vq_next_absmax_word(K, D, ws, step & 1)); }
    C3DGS_STAGE("vq_accumulate", 0, s);
    return C3DGS_OK;
}

int c3dgs_vq_step_apply(int32_t step, int32_t K, int32_t D, float* S, float* codebook, float* entry_importance, float decay,
                        float alpha, float eps, int32_t scale_normalize, void* ws, size_t ws_bytes, void* stream)
{
    if (step < 0 || K <= 0 || D <= 0 || !S || !codebook || !entry_importance) return fail(C3DGS_E_INVALID, "vq_step_apply: bad arguments");
    { StageTimer t_(ST_VQ_APPLY, (hipStream_t)stream);
      if (launch_vq_apply_split(K, D, S, codebook, entry_importance, decay, alpha, eps, scale_normalize, ws, ws_bytes, step & 1, (hipStream_t)stream))
          return fail(C3DGS_E_INVALID, "vq_step_apply: shape / scratch not served by the fused step (see c3dgs_vq_step_supported)"); }
    C3DGS_STAGE("vq_apply", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_debug_wd_scores(int64_t N, int32_t C, int32_t K, const float* coefs, const float* codebook, float* scores, void* ws,
                          size_t ws_bytes, float* out_dist, int64_t* out_idx, void* stream)
{
    if (!coefs || !codebook || !scores || !out_dist || !out_idx) return fail(C3DGS_E_INVALID, "debug_wd_scores: bad arguments");
    if (launch_wd_debug_scores(N, C, K, coefs, codebook, scores, ws, ws_bytes, out_dist, out_idx, (hipStream_t)stream))
        return fail(C3DGS_E_INVALID, "debug_wd_scores: K = 48, 1 <= N <= 256, C >= 32 and scratch of c3dgs_weighted_distance_ws_bytes");
    return C3DGS_OK;
}

int c3dgs_vq_apply(int32_t K, int32_t D, const float* S, float* codebook, float* entry_importance, float decay,
                   float alpha, float eps, int32_t scale_normalize, void* stream)
{
    if (K <= 0 || D <= 0 || !S || !codebook || !entry_importance) return fail(C3DGS_E_INVALID, "bad arguments");
    { StageTimer t_(ST_VQ_APPLY, (hipStream_t)stream); launch_vq_apply(K, D, S, codebook, entry_importance, decay, alpha, eps, scale_normalize, (hipStream_t)stream); }
    C3DGS_STAGE("vq_apply", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

size_t c3dgs_morton_workspace_bytes(int32_t P) { return morton_workspace_bytes(P); }

int c3dgs_morton_order(int32_t P, const float* xyz, int64_t* codes, int64_t* order, void* workspace, void* stream)
{
    if (P < 0) return fail(C3DGS_E_INVALID, "P must be >= 0");
    if (P == 0) return C3DGS_OK;
    if (!xyz || !codes || !order || !workspace) return fail(C3DGS_E_INVALID, "morton_order: bad arguments");
    if (run_morton_order(P, xyz, codes, order, workspace, (hipStream_t)stream)) return fail(C3DGS_E_HIP, "morton_order failed");
    C3DGS_STAGE("morton_order", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_abs_accumulate(int64_t n, const float* g, float* acc, void* stream)
{
    if (n < 0 || (n > 0 && (!g || !acc))) return fail(C3DGS_E_INVALID, "abs_accumulate: bad arguments");
    launch_abs_accumulate(n, g, acc, (hipStream_t)stream);
    C3DGS_STAGE("abs_accumulate", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_adam_step(int32_t n_tensors, const c3dgs_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream)
{
    if (n_tensors < 0 || n_tensors > C3DGS_ADAM_MAX_TENSORS) return fail(C3DGS_E_INVALID, "adam_step: between 0 and 16 tensors per call");
    if (n_tensors == 0) return C3DGS_OK;
    if (!tensors) return fail(C3DGS_E_INVALID, "adam_step: tensors is NULL");
    for (int k = 0; k < n_tensors; k++)
        if (tensors[k].n > 0 && (!tensors[k].param || !tensors[k].grad || !tensors[k].exp_avg || !tensors[k].exp_avg_sq))
            return fail(C3DGS_E_INVALID, "adam_step: NULL tensor pointer");
    { StageTimer t_(ST_ADAM, (hipStream_t)stream); launch_adam(n_tensors, tensors, beta1, beta2, eps, (hipStream_t)stream); }
    C3DGS_STAGE("adam_step", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_extract_rot_scale(int32_t n, const float* cov6, float* rot, float* scale, void* stream)
{
    if (n < 0) return fail(C3DGS_E_INVALID, "extract_rot_scale: n must be >= 0");
    if (n == 0) return C3DGS_OK;
    if (!cov6 || !rot || !scale || (reinterpret_cast<uintptr_t>(rot) & 15)) return fail(C3DGS_E_INVALID, "extract_rot_scale: bad arguments");
    launch_extract_rot_scale(n, cov6, rot, scale, (hipStream_t)stream);
    C3DGS_STAGE("extract_rot_scale", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float* dmaps, double* sums,
                          void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !sums) return fail(C3DGS_E_INVALID, "l1_ssim_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    C3DGS_HIP_TRY(hipMemsetAsync(sums, 0, 128 * sizeof(double), s));
    const size_t n = (size_t)C * H * W;
    {
        StageTimer t_(ST_LOSS_FWD, s);
        launch_l1_ssim_forward(C, H, W, img, gt, dmaps, dmaps ? dmaps + n : nullptr, dmaps ? dmaps + 2 * n : nullptr, sums, s);
    }
    C3DGS_STAGE("l1_ssim_forward", 0, s);
    return C3DGS_OK;
}

int c3dgs_l1_ssim_value(const double* sums, double l1_scale, double ssim_scale, double constant, float* out, void* stream)
{
    if (!sums || !out) return fail(C3DGS_E_INVALID, "l1_ssim_value: bad arguments");
    launch_l1_ssim_value(sums, l1_scale, ssim_scale, constant, out, (hipStream_t)stream);
    C3DGS_STAGE("l1_ssim_value", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, const float* dmaps,
                           const float* grad_loss, float l1_coeff, float ssim_coeff, float* dL_dimg, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !dmaps || !grad_loss || !dL_dimg)
        return fail(C3DGS_E_INVALID, "l1_ssim_backward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)C * H * W;
    {
        StageTimer t_(ST_LOSS_BWD, s);
        launch_l1_ssim_backward(C, H, W, img, gt, dmaps, dmaps + n, dmaps + 2 * n, grad_loss, l1_coeff, ssim_coeff, dL_dimg, s);
    }
    C3DGS_STAGE("l1_ssim_backward", 0, s);
    return C3DGS_OK;
}

// ---- QAT getters ----
static int qat_validate(const c3dgs_qat_params* q, const char* who)
{
    if (!q) return fail(C3DGS_E_INVALID, std::string(who) + ": params is NULL");
    if (q->P < 0 || q->GS < 0 || q->SHS < 0 || q->M < 1) return fail(C3DGS_E_INVALID, std::string(who) + ": bad sizes");
    if (!q->state) return fail(C3DGS_E_INVALID, std::string(who) + ": state is required");
    if (q->features_dc && q->M > 1 && !q->features_rest)
        return fail(C3DGS_E_INVALID, std::string(who) + ": features_rest is required with features_dc when M > 1");
    if (reinterpret_cast<uintptr_t>(q->rotation) & 15) return fail(C3DGS_E_INVALID, std::string(who) + ": rotation must be 16-byte aligned");
    return C3DGS_OK;
}
static bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

size_t c3dgs_qat_workspace_bytes(void) { return qat_workspace_bytes(); }
size_t c3dgs_qat_scan_bytes(int32_t P) { return qat_scan_bytes(P); }

int c3dgs_qat_observe(const c3dgs_qat_params* q, void* workspace, void* stream)
{
    if (int rc = qat_validate(q, "qat_observe")) return rc;
    if (!workspace) return fail(C3DGS_E_INVALID, "qat_observe: workspace is required");
    { StageTimer t_(ST_QAT_OBSERVE, (hipStream_t)stream); launch_qat_observe(*q, workspace, (hipStream_t)stream); }
    C3DGS_STAGE("qat_observe", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_qat_codebooks(const c3dgs_qat_params* q, float* scales_n, float* rotations, float* shs, void* stream)
{
    if (int rc = qat_validate(q, "qat_codebooks")) return rc;
    if (misaligned16(rotations) || misaligned16(shs)) return fail(C3DGS_E_INVALID, "qat_codebooks: outputs must be 16-byte aligned");
    { StageTimer t_(ST_QAT_CODEBOOKS, (hipStream_t)stream); launch_qat_codebooks(*q, scales_n, rotations, shs, (hipStream_t)stream); }
    C3DGS_STAGE("qat_codebooks", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_qat_codebooks_backward(const c3dgs_qat_params* q, const float* dL_dscales_n, const float* dL_drotations,
                                 const float* dL_dshs, float* dL_dscaling, float* dL_drotation, float* dL_dfeatures_dc,
                                 float* dL_dfeatures_rest, void* stream)
{
    if (int rc = qat_validate(q, "qat_codebooks_backward")) return rc;
    if (misaligned16(dL_drotations) || misaligned16(dL_drotation) || misaligned16(dL_dshs))
        return fail(C3DGS_E_INVALID, "qat_codebooks_backward: rotation / sh gradients must be 16-byte aligned");
    if (dL_dshs && q->features_dc && (!dL_dfeatures_dc || (q->M > 1 && !dL_dfeatures_rest)))
        return fail(C3DGS_E_INVALID, "qat_codebooks_backward: dL_dfeatures_dc / dL_dfeatures_rest are required with dL_dshs");
    {
        StageTimer t_(ST_QAT_CODEBOOKS_BWD, (hipStream_t)stream);
        launch_qat_codebooks_backward(*q, dL_dscales_n, dL_drotations, dL_dshs, dL_dscaling, dL_drotation, dL_dfeatures_dc,
                                      dL_dfeatures_rest, (hipStream_t)stream);
    }
    C3DGS_STAGE("qat_codebooks_backward", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_qat_visible(const c3dgs_qat_params* q, const float* viewmatrix, uint8_t* visible, int32_t* rank, int32_t* count,
                      void* scan_workspace, void* stream)
{
    if (int rc = qat_validate(q, "qat_visible")) return rc;
    if (!count) return fail(C3DGS_E_INVALID, "qat_visible: count is required");
    hipStream_t s = (hipStream_t)stream;
    if (q->P == 0) { C3DGS_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), s)); return C3DGS_OK; }
    if (!q->xyz || !viewmatrix || !visible || !rank || !scan_workspace) return fail(C3DGS_E_INVALID, "qat_visible: bad arguments");
    { StageTimer t_(ST_QAT_VISIBLE, s); C3DGS_HIP_TRY(run_qat_visible(*q, viewmatrix, visible, rank, count, scan_workspace, s)); }
    C3DGS_STAGE("qat_visible", 0, s);
    return C3DGS_OK;
}

int c3dgs_qat_points(const c3dgs_qat_params* q, const uint8_t* visible, const int32_t* rank, const int64_t* sh_indices,
                     const int64_t* g_indices, float* means3D, float* opacities, float* scale_factors, int64_t* sh_indices_out,
                     int64_t* g_indices_out, void* stream)
{
    if (int rc = qat_validate(q, "qat_points")) return rc;
    if ((visible == nullptr) != (rank == nullptr)) return fail(C3DGS_E_INVALID, "qat_points: visible and rank go together");
    {
        StageTimer t_(ST_QAT_POINTS, (hipStream_t)stream);
        launch_qat_points(*q, visible, rank, sh_indices, g_indices, means3D, opacities, scale_factors, sh_indices_out,
                          g_indices_out, (hipStream_t)stream);
    }
    C3DGS_STAGE("qat_points", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_qat_points_backward(const c3dgs_qat_params* q, const uint8_t* visible, const int32_t* rank, const float* dL_dmeans3D,
                              const float* dL_dmeans2D, const float* dL_dopacities, const float* dL_dscale_factors,
                              float* dL_dxyz, float* dL_dscreenspace, float* dL_dopacity, float* dL_dscaling_factor, void* stream)
{
    if (int rc = qat_validate(q, "qat_points_backward")) return rc;
    if ((visible == nullptr) != (rank == nullptr)) return fail(C3DGS_E_INVALID, "qat_points_backward: visible and rank go together");
    if ((dL_dopacity && dL_dopacities && !q->opacity) || (dL_dscaling_factor && dL_dscale_factors && !q->scaling_factor))
        return fail(C3DGS_E_INVALID, "qat_points_backward: the raw opacity / scaling_factor are needed to recompute the masks");
    {
        StageTimer t_(ST_QAT_POINTS_BWD, (hipStream_t)stream);
        launch_qat_points_backward(*q, visible, rank, dL_dmeans3D, dL_dmeans2D, dL_dopacities, dL_dscale_factors, dL_dxyz,
                                   dL_dscreenspace, dL_dopacity, dL_dscaling_factor, (hipStream_t)stream);
    }
    C3DGS_STAGE("qat_points_backward", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_qat_quantize(const c3dgs_qat_params* q, int32_t scaling_is_exp, int8_t* opacity, int8_t* scaling, int8_t* scaling_factor,
                       int8_t* rotation, int8_t* features_dc, int8_t* features_rest, void* stream)
{
    if (int rc = qat_validate(q, "qat_quantize")) return rc;
    if (rotation && (reinterpret_cast<uintptr_t>(rotation) & 3)) return fail(C3DGS_E_INVALID, "qat_quantize: rotation output must be 4-byte aligned");
    launch_qat_quantize(*q, scaling_is_exp, opacity, scaling, scaling_factor, rotation, features_dc, features_rest, (hipStream_t)stream);
    C3DGS_STAGE("qat_quantize", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_fake_quantize(int64_t n, const float* x, c3dgs_fq_state* state, int32_t observe, int32_t enabled,
                        float averaging_constant, float* out, void* workspace, void* stream)
{
    if (n < 0) return fail(C3DGS_E_INVALID, "fake_quantize: n must be >= 0");
    if (n == 0) return C3DGS_OK;
    if (!x || !state || !out || (observe && !workspace)) return fail(C3DGS_E_INVALID, "fake_quantize: bad arguments");
    launch_fake_quantize(n, x, state, observe, enabled, averaging_constant, out, workspace, (hipStream_t)stream);
    C3DGS_STAGE("fake_quantize", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

int c3dgs_fake_quantize_backward(int64_t n, const float* x, const c3dgs_fq_state* state, int32_t enabled, const float* g,
                                 float* dx, void* stream)
{
    if (n < 0) return fail(C3DGS_E_INVALID, "fake_quantize_backward: n must be >= 0");
    if (n == 0) return C3DGS_OK;
    if (!x || !state || !g || !dx) return fail(C3DGS_E_INVALID, "fake_quantize_backward: bad arguments");
    launch_fake_quantize_backward(n, x, state, enabled, g, dx, (hipStream_t)stream);
    C3DGS_STAGE("fake_quantize_backward", 0, (hipStream_t)stream);
    return C3DGS_OK;
}

} // extern "C"
