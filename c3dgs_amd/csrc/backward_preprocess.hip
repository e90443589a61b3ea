// backward_preprocess.hip -- per-Gaussian backward for gfx950, two kernels:
//   sum_partials_kernel (one lane per Gaussian, all P): sums the Gaussian's contiguous run of per-tile partial sums
//       written by render_backward_kernel, zero-fills the gradient rows of every Gaussian no pixel blended and appends
//       the others to a compact list (id + where its nine sums were parked);
//   backward_preprocess_kernel (one lane per LISTED Gaussian, full waves):
//       K11  computeCov2DCUDA        reference cuda_rasterizer/backward.cu:144-274,
//       K12  BACKWARD::preprocessCUDA reference backward.cu:346-396 (+ computeColorFromSH :20-139,
//            computeCov3D :278-341), K12i reference backward_indexed.cu:20-342.
// Only about a third of the Gaussians of a dense view are ever blended: as ONE kernel the ~600-instruction per-Gaussian
// part ran with a third of its lanes and at the four waves per SIMD its registers allow, which also throttled the
// streaming summation in front of it. Split, the summation runs at full occupancy and the heavy part on full waves.
// Together they write EVERY element of the P-sized outputs (zeros for culled Gaussians), so the caller needs
// no zero-fill of those (the reference memsets ~0.9 GB per call at P=3M, rasterize_points.cu:153-162).
// Only the codebook-sized outputs of the indexed variant are accumulated with fp32 atomics
// (pre-zeroed by the C-ABI entry point); atomicAdd(float*) is a single global_atomic_add_f32 on gfx950.
// 3D covariances are recomputed from scale/rotation instead of being stored by the forward.
#include "common.hpp"
#include "gsmath.hpp"

#ifndef C3DGS_BWD_CH
#define C3DGS_BWD_CH 128
#endif
#ifndef C3DGS_BWD_LONG_RUN
#define C3DGS_BWD_LONG_RUN 32   // sum_partials: a lane's run of more than this many staged entries of a chunk is summed by the whole wave
                                // (tools/ablate_bwdpre.sh, stage ms synth-v1 / heavy tail: 8 -> 0.316 / 0.280, 16 -> 0.261 / 0.279, 32 -> 0.256 / 0.276)
#endif
#ifndef C3DGS_BWD_FG
#define C3DGS_BWD_FG 8
#endif
#ifndef C3DGS_SUM_WAVES
#define C3DGS_SUM_WAVES 16  // waves per workgroup of sum_partials_kernel = 64 x this many Gaussians share one list of blended ones
#endif
#ifndef C3DGS_BWD_WAVES
#define C3DGS_BWD_WAVES 3   // waves per SIMD of backward_preprocess_kernel (168 VGPRs, nothing spilled); measured on one box,
#endif                      // interleaved: 3 -> 0.153 ms, 4 (128 VGPRs, 32 spilled) -> 0.166, 5 -> 0.20

namespace c3dgs {

struct BwdArgs {
    int P, D, M, W, H;
    const float* means3D; const float* sh; const float* scales; const float* scale_factors; const float* rotations;
    const float* cov3D_precomp; const int64_t* sh_indices; const int64_t* g_indices;
    const float* view; const float* proj; const float* campos;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    const int32_t* radii; const uint32_t* inst_offset; const uint32_t* block_base; const uint8_t* clamped; const float4* splat;
    float* partials; const uint8_t* touched;
    uint32_t* live_count; uint32_t* live_ids; uint32_t* live_slots;   // the compact list (sum_partials_kernel -> backward_preprocess_kernel)
    c3dgs_raster_grads g;
};

// SH backward. c = this Gaussian's coefficients; the gradient row is basis_out[k] * g[ch], which the caller writes (or
// scatter-adds) cooperatively, a wave per row.
template <int DEG>
__device__ __forceinline__ void sh_backward(const float* c, const f3 d0, const float g[3], float dmean_add[3], float* basis_out)
{
    const float len = sqrtf(d0.x * d0.x + d0.y * d0.y + d0.z * d0.z);
    const float x = d0.x / len, y = d0.y / len, z = d0.z / len;
    float dx[3] = { 0, 0, 0 }, dy[3] = { 0, 0, 0 }, dz[3] = { 0, 0, 0 };
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float basis[NB];
    basis[0] = SH_C0;
    if (DEG > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            dx[ch] = -SH_C1 * c[3 * 3 + ch];
            dy[ch] = -SH_C1 * c[1 * 3 + ch];
            dz[ch] = SH_C1 * c[2 * 3 + ch];
        }
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        basis[4] = SH_C2_0 * xy; basis[5] = SH_C2_1 * yz; basis[6] = SH_C2_2 * (2.f * zz - xx - yy);
        basis[7] = SH_C2_3 * xz; basis[8] = SH_C2_4 * (xx - yy);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float s4 = c[4 * 3 + ch], s5 = c[5 * 3 + ch], s6 = c[6 * 3 + ch], s7 = c[7 * 3 + ch], s8 = c[8 * 3 + ch];
            dx[ch] += SH_C2_0 * y * s4 + SH_C2_2 * 2.f * -x * s6 + SH_C2_3 * z * s7 + SH_C2_4 * 2.f * x * s8;
            dy[ch] += SH_C2_0 * x * s4 + SH_C2_1 * z * s5 + SH_C2_2 * 2.f * -y * s6 + SH_C2_4 * 2.f * -y * s8;
            dz[ch] += SH_C2_1 * y * s5 + SH_C2_2 * 2.f * 2.f * z * s6 + SH_C2_3 * x * s7;
        }
        if (DEG > 2) {
            basis[9] = SH_C3_0 * y * (3.f * xx - yy);
            basis[10] = SH_C3_1 * xy * z;
            basis[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
            basis[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
            basis[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
            basis[14] = SH_C3_5 * z * (xx - yy);
            basis[15] = SH_C3_6 * x * (xx - 3.f * yy);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float s9 = c[9 * 3 + ch], s10 = c[10 * 3 + ch], s11 = c[11 * 3 + ch], s12 = c[12 * 3 + ch],
                            s13 = c[13 * 3 + ch], s14 = c[14 * 3 + ch], s15 = c[15 * 3 + ch];
                dx[ch] += SH_C3_0 * s9 * 3.f * 2.f * xy + SH_C3_1 * s10 * yz + SH_C3_2 * s11 * -2.f * xy +
                          SH_C3_3 * s12 * -3.f * 2.f * xz + SH_C3_4 * s13 * (-3.f * xx + 4.f * zz - yy) +
                          SH_C3_5 * s14 * 2.f * xz + SH_C3_6 * s15 * 3.f * (xx - yy);
                dy[ch] += SH_C3_0 * s9 * 3.f * (xx - yy) + SH_C3_1 * s10 * xz + SH_C3_2 * s11 * (-3.f * yy + 4.f * zz - xx) +
                          SH_C3_3 * s12 * -3.f * 2.f * yz + SH_C3_4 * s13 * -2.f * xy + SH_C3_5 * s14 * -2.f * yz +
                          SH_C3_6 * s15 * -3.f * 2.f * xy;
                dz[ch] += SH_C3_1 * s10 * xy + SH_C3_2 * s11 * 4.f * 2.f * yz + SH_C3_3 * s12 * 3.f * (2.f * zz - xx - yy) +
                          SH_C3_4 * s13 * 4.f * 2.f * xz + SH_C3_5 * s14 * (xx - yy);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) basis_out[k] = basis[k];
    const float ddx = dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2];
    const float ddy = dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2];
    const float ddz = dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2];
    // dnormvdv, auxiliary.h:107-117
    const float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean_add[0] = ((+sum2 - d0.x * d0.x) * ddx - d0.y * d0.x * ddy - d0.z * d0.x * ddz) * invsum32;
    dmean_add[1] = (-d0.x * d0.y * ddx + (sum2 - d0.y * d0.y) * ddy - d0.z * d0.y * ddz) * invsum32;
    dmean_add[2] = (-d0.x * d0.z * ddx - d0.y * d0.z * ddy + (sum2 - d0.z * d0.z) * ddz) * invsum32;
}

// position of the r-th (0-based) set bit of m; r < popcount(m)
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long m, uint32_t r)
{
    uint32_t w = (uint32_t)m, at = 0, c = (uint32_t)__popc(w);
    if (r >= c) { r -= c; w = (uint32_t)(m >> 32); at = 32; }
    c = (uint32_t)__popc(w & 0xFFFFu); if (r >= c) { r -= c; w >>= 16; at += 16; }
    c = (uint32_t)__popc(w & 0xFFu);   if (r >= c) { r -= c; w >>= 8; at += 8; }
    c = (uint32_t)__popc(w & 0xFu);    if (r >= c) { r -= c; w >>= 4; at += 4; }
    c = (uint32_t)__popc(w & 0x3u);    if (r >= c) { r -= c; w >>= 2; at += 2; }
    if (r >= (w & 1u)) at += 1;
    return at;
}

// ---- kernel 1: per-Gaussian sums of the per-tile partials, zero rows, compact list of the blended Gaussians
// 64 registers at most: two 1024-thread workgroups per CU (75 KB of LDS each) are eight waves per SIMD; at 72 registers only one
// workgroup fits and the kernel runs 8 % slower on synth-v1
__global__ void __launch_bounds__(64 * C3DGS_SUM_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) sum_partials_kernel(const BwdArgs a)
{
    constexpr int SW = C3DGS_SUM_WAVES, LIST = 64 * SW;
    const int i = blockIdx.x * LIST + threadIdx.x;
    const size_t si = (size_t)i;
    const c3dgs_raster_grads& o = a.g;

    // ---- (a) sum each Gaussian's per-tile partials: slots [start, end) of inst_offset (id order, so the 64
    // Gaussians of a wave own ONE contiguous slot range). A per-lane loop over global memory would run as long as
    // the wave's largest Gaussian and pay a dependent HBM latency per slot; instead the wave streams its range
    // through LDS in chunks with coalesced, independent loads (skipping never-written slots by their flag byte)
    // and every lane then adds up its own run from LDS. Fixed order -> still bitwise reproducible.
    constexpr int CH = C3DGS_BWD_CH;
    // one LDS staging area per wave for the partial sums [CH][9]; a wave only ever touches its own
    __shared__ float s_buf[C3DGS_SUM_WAVES][CH * PARTIAL_FLOATS];
#define s_stage(w, sl, q) s_buf[w][(sl) * PARTIAL_FLOATS + (q)]
    // Only ~1/4 of the slots were ever written (the blend kernel stops at each tile's saturation point), so the wave
    // first reads the FLAG bytes of a whole sweep of FG*64 slots (independent byte loads, one wait), turns them into
    // per-group ballots + running counts, and stages only the WRITTEN slots, compacted: entry e of the sweep = the e-th
    // written slot. A lane's own run [start, end) maps to the contiguous entry range [below(start), below(end)), so the
    // summation loop touches no flags and one pass usually covers the wave's whole range.
    constexpr int FG = C3DGS_BWD_FG;
    __shared__ unsigned long long s_fl[C3DGS_SUM_WAVES][FG];
    __shared__ uint32_t s_pre[C3DGS_SUM_WAVES][FG + 1];
    bool any_written = false;            // did ANY pixel of ANY tile blend this Gaussian?
    float acc[PARTIAL_FLOATS];
#pragma unroll
    for (int q = 0; q < PARTIAL_FLOATS; q++) acc[q] = 0.f;
    const int lane_ = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane_) - 1ull;
    uint32_t start_;
    {
        const int ic = min(i, a.P - 1);                       // lanes past P clamp to the last Gaussian (empty run)
        // global slot offsets = per-workgroup inclusive offsets of the forward + block_base[] (preprocess.hip)
        const uint32_t end_ = a.inst_offset[ic] + a.block_base[ic >> 8];
        start_ = (ic == 0) ? 0u : a.inst_offset[ic - 1] + a.block_base[(ic - 1) >> 8];
        if (i >= a.P) start_ = end_;
        const uint32_t w_begin = __builtin_amdgcn_readfirstlane(start_);
        const uint32_t w_end = __builtin_amdgcn_readlane(end_, 63);
        for (uint32_t base = w_begin; base < w_end; base += FG * 64) {
            const uint32_t nsl = min((uint32_t)(FG * 64), w_end - base);
            uint32_t fbits = 0;                               // bit g: slot base + 64 g + lane was written
            uint8_t fb[FG];                                   // unconditional loads (clamped), so all FG are in flight together
#pragma unroll
            for (int g = 0; g < FG; g++) fb[g] = a.touched[base + min((uint32_t)(g * 64 + lane_), nsl - 1u)];
#pragma unroll
            for (int g = 0; g < FG; g++)
                if ((uint32_t)(g * 64 + lane_) < nsl && fb[g] != 0) fbits |= 1u << g;
            uint32_t W = 0;                                   // wave-uniform running count of written slots
            uint32_t pre[FG + 1];                             // wave-uniform: written slots in front of group g
#pragma unroll
            for (int g = 0; g < FG; g++) {
                const unsigned long long bm = __ballot((fbits >> g) & 1u);
                if (lane_ == 0) { s_fl[wv][g] = bm; s_pre[wv][g] = W; }
                pre[g] = W;
                W += (uint32_t)__popcll(bm);
            }
            pre[FG] = W;
            if (W == 0) continue;                             // nothing in this sweep was ever blended
            if (lane_ == 0) s_pre[wv][FG] = W;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            auto below = [&](uint32_t x) -> uint32_t {        // written slots of this sweep before slot offset x
                const uint32_t g = x >> 6;
                if (g >= (uint32_t)FG) return s_pre[wv][FG];
                return s_pre[wv][g] + (uint32_t)__popcll(s_fl[wv][g] & ((1ull << (x & 63u)) - 1ull));
            };
            const uint32_t lo = min(max(start_, base), base + nsl) - base, hi = min(max(end_, base), base + nsl) - base;
            const uint32_t clo = below(lo), chi = below(hi);
            any_written |= chi > clo;
            for (uint32_t p0 = 0; p0 < W; p0 += CH) {
                // Staging with the lanes on the COMPACTED entries: lane l fetches entries p0 + l, p0 + 64 + l, ... It finds
                // its entry's slot from the ballots (group by the running counts, then the r-th set bit of that group's
                // mask), so every load of the chunk is independent of every other and issued before the first wait --
                // with the lanes on the raw slots, each group's loads sat behind a branch and the wave paid one memory
                // round trip per group.
                float st[CH / 64][PARTIAL_FLOATS];
#pragma unroll
                for (int h = 0; h < CH / 64; h++) {
                    const uint32_t e = min(p0 + (uint32_t)(h * 64 + lane_), W - 1u);   // clamped: always a real entry
                    uint32_t g = 0, r = e;
#pragma unroll
                    for (int k = 1; k < FG; k++)
                        if (e >= pre[k]) { g = (uint32_t)k; r = e - pre[k]; }
                    const uint32_t slot = g * 64u + nth_set_bit(s_fl[wv][g], r);
                    const float* src = a.partials + (size_t)(base + slot) * PARTIAL_FLOATS;
#pragma unroll
                    for (int q = 0; q < PARTIAL_FLOATS; q++) st[h][q] = src[q];
                }
#pragma unroll
                for (int h = 0; h < CH / 64; h++)
#pragma unroll
                    for (int q = 0; q < PARTIAL_FLOATS; q++) s_stage(wv, h * 64 + lane_, q) = st[h][q];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t c_lo = max(clo, p0), c_hi = min(chi, p0 + CH);      // this lane's entries of the chunk
                const bool long_run = c_hi > c_lo + (uint32_t)C3DGS_BWD_LONG_RUN;
                // A Gaussian that covers hundreds of tiles owns most of a chunk: alone, its lane would add the entries up one
                // after the other while 63 lanes wait (on a heavy-tailed scene -- a per mille of the splats hundreds of pixels
                // wide -- nearly every workgroup holds one, and this kernel took 0.73 ms against 0.12 on synth-v1). Long runs
                // are summed by the whole wave instead: lane l takes entries l, l + 64, ... of the run, then a fixed butterfly;
                // which path a run takes depends only on its length, so the sums stay reproducible run to run.
                unsigned long long longm = __ballot(long_run);
                while (longm) {
                    const int owner = __ffsll((long long)longm) - 1;
                    longm &= longm - 1ull;
                    const uint32_t o_lo = (uint32_t)__shfl((int)c_lo, owner) - p0, o_hi = (uint32_t)__shfl((int)c_hi, owner) - p0;
                    float part[PARTIAL_FLOATS];
#pragma unroll
                    for (int q = 0; q < PARTIAL_FLOATS; q++) part[q] = 0.f;
                    for (uint32_t e = o_lo + (uint32_t)lane_; e < o_hi; e += 64u) {
#pragma unroll
                        for (int q = 0; q < PARTIAL_FLOATS; q++) part[q] += s_stage(wv, e, q);
                    }
#pragma unroll
                    for (int q = 0; q < PARTIAL_FLOATS; q++) {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) part[q] += __shfl_xor(part[q], o);
                        if (lane_ == owner) acc[q] += part[q];
                    }
                }
                if (!long_run) {
                    for (uint32_t c = c_lo; c < c_hi; c++) {
#pragma unroll
                        for (int q = 0; q < PARTIAL_FLOATS; q++) acc[q] += s_stage(wv, c - p0, q);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // A Gaussian none of whose tile instances was ever blended (culled, or behind every pixel's saturation point: the
    // majority on dense scenes) has exactly zero gradients: zero rows here, no SH / codebook gathers, no scatter-adds.
    // Same values as the reference, which adds nothing for it. (any_written implies i < P and radii[i] > 0.)
    const bool live = any_written;
    const unsigned long long lm = __ballot(live);
    const bool per_gaussian_rows = !a.sh_indices && !a.g_indices;          // non-indexed: [P, M, 3] SH gradient
    // Zero SH rows of the non-indexed variant (192 B per Gaussian at M = 16: the bulk of this kernel's stores on a dense view):
    // the wave's 64 rows are one contiguous span, cleared with coalesced 16-byte stores that skip the blended Gaussians'
    // rows (a lane zeroing its own row wrote 48 dwords at a 192-byte stride: 0.41 ms for this kernel at P = 3M, 0.12 indexed).
    bool sh_rows_done = false;
    if (per_gaussian_rows && o.dL_dsh && a.sh && ((a.M * 3) & 3) == 0 && (reinterpret_cast<uintptr_t>(o.dL_dsh) & 15u) == 0) {
        sh_rows_done = true;
        const int r4 = (a.M * 3) >> 2;                                     // 16-byte words per row
        const int first = blockIdx.x * LIST + wv * 64;                     // the wave's first Gaussian
        const int rows = min(64, a.P - first);
        float4* span = reinterpret_cast<float4*>(o.dL_dsh + (size_t)first * a.M * 3);
        int row = lane_ / r4, col = lane_ - row * r4;                      // 16-byte word q = row * r4 + col, walked with stride 64
        const int drow = 64 / r4, dcol = 64 - drow * r4;
        for (int q = lane_; q < rows * r4; q += 64) {
            if (!((lm >> row) & 1ull)) span[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            row += drow; col += dcol;
            if (col >= r4) { col -= r4; row++; }
        }
    }
    if (i < a.P && !live) {
        if (o.dL_dmeans2D) { o.dL_dmeans2D[3 * si] = 0.f; o.dL_dmeans2D[3 * si + 1] = 0.f; o.dL_dmeans2D[3 * si + 2] = 0.f; }
        if (o.dL_dcolors) { o.dL_dcolors[3 * si] = 0.f; o.dL_dcolors[3 * si + 1] = 0.f; o.dL_dcolors[3 * si + 2] = 0.f; }
        if (o.dL_dopacity) o.dL_dopacity[si] = 0.f;
        if (o.dL_dmeans3D) { o.dL_dmeans3D[3 * si] = 0.f; o.dL_dmeans3D[3 * si + 1] = 0.f; o.dL_dmeans3D[3 * si + 2] = 0.f; }
        if (o.dL_dcov3D) for (int q = 0; q < 6; q++) o.dL_dcov3D[6 * si + q] = 0.f;
        if (o.dL_dscale_factors) o.dL_dscale_factors[si] = 0.f;
        if (per_gaussian_rows) {
            if (o.dL_dsh && a.sh && !sh_rows_done) for (int q = 0; q < a.M * 3; q++) o.dL_dsh[si * a.M * 3 + q] = 0.f;
            if (o.dL_dscales && a.scales) for (int q = 0; q < 3; q++) o.dL_dscales[3 * si + q] = 0.f;
            if (o.dL_drotations && a.scales) for (int q = 0; q < 4; q++) o.dL_drotations[4 * si + q] = 0.f;
        }
    }
    // The blended ones go on the workgroup's list (256 entries per workgroup, filled from the front, in id order; a single
    // global list would cost one same-address atomic per wave). The nine sums are parked in the Gaussian's own FIRST slot
    // of the partial-sum array (only this wave reads this wave's slots, and it is done with them).
    __shared__ uint32_t s_cnt[SW];
    if (lane_ == 0) s_cnt[wv] = (uint32_t)__popcll(lm);
    __syncthreads();
    uint32_t run0 = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SW; w++) { const uint32_t c = s_cnt[w]; run0 += (w < wv) ? c : 0u; all += c; }
    if (live) {
        const size_t pos = (size_t)blockIdx.x * LIST + run0 + (uint32_t)__popcll(lm & lt_mask);
        a.live_ids[pos] = (uint32_t)i;
        a.live_slots[pos] = start_;
        float* dst = a.partials + (size_t)start_ * PARTIAL_FLOATS;
#pragma unroll
        for (int q = 0; q < PARTIAL_FLOATS; q++) dst[q] = acc[q];
    }
    if (threadIdx.x == 0) a.live_count[blockIdx.x] = all;
}
#undef s_stage

// ---- kernel 2: the blended Gaussians, one lane each, in list order
template <int DEG, bool INDEXED>
__global__ void __launch_bounds__(256)
#if C3DGS_BWD_WAVES
__attribute__((amdgpu_waves_per_eu(C3DGS_BWD_WAVES, C3DGS_BWD_WAVES)))
#endif
backward_preprocess_kernel(const BwdArgs a, const float* __restrict__ cam_view, const float* __restrict__ cam_proj,
                           const float* __restrict__ cam_pos)   // the camera as direct restrict parameters: scalar loads
{
    // the workgroup's list fills whole waves from the front: a wave past its end has nothing to do (waves are independent:
    // no workgroup barrier below)
    // Workgroup -> (list, quarter of the list): all FIRST quarters come first in the grid, then all second ones, ...
    // Lists fill from the front, so the populated quarters are spread evenly over the XCDs (consecutive workgroups go to
    // consecutive XCDs; quarter = blockIdx % 4 would send every full quarter to the same two of the eight) and the
    // mostly empty ones come last.
    const uint32_t n_lists = gridDim.x / (C3DGS_SUM_WAVES / 4);
    const uint32_t list = blockIdx.x % n_lists, quarter = blockIdx.x / n_lists;
    const uint32_t n_live = a.live_count[list];
    const uint32_t off = quarter * 256u + threadIdx.x;
    if ((off & ~63u) >= n_live) return;
    const size_t pos = (size_t)list * (64 * C3DGS_SUM_WAVES) + off;
    const bool live = off < n_live;
    const int i = live ? (int)a.live_ids[pos] : 0;
    const size_t si = (size_t)i;
    const c3dgs_raster_grads& o = a.g;
    constexpr int NB = (DEG + 1) * (DEG + 1);

    // Indexed variant: codebook-sized gradients are scatter-ADDED. One lane per Gaussian would issue each atomic
    // with 64 lanes in 64 different rows, the slowest shape for the chip's memory-side float atomics
    // (MI355X_MICROARCH.md, Global float atomics: ~17x below the contiguous rate). Instead every lane parks its
    // factors in LDS and the wave then walks its 64 Gaussians together: one atomic instruction per Gaussian whose
    // lanes cover that Gaussian's contiguous gradient row (up to 48 floats = 192 B for SH).
    // (The non-indexed variant writes its [P, M, 3] rows the same cooperative way, with plain stores.)
    __shared__ float s_g[256][3];
    __shared__ int32_t s_row[256];                    // codebook rows fit int32 (SHS, GS are int32 in the ABI), so do Gaussian ids
    __shared__ float s_ds[INDEXED ? 256 : 1][3];
    __shared__ float s_dq[INDEXED ? 256 : 1][4];
    __shared__ int32_t s_gi[INDEXED ? 256 : 1];
    __shared__ float s_bas[256][NB + 1];              // SH basis values of the workgroup's Gaussians
#define s_basis(t, k) s_bas[t][k]
    s_row[threadIdx.x] = -1;
    if (INDEXED) s_gi[threadIdx.x] = -1;

    if (live) {
    // Memory round trips are what this kernel's time is made of (random rows, four waves per SIMD), so the loads are
    // issued by dependency LEVEL, not where the arithmetic wants them: (1) everything addressed by the Gaussian's id or
    // slot, together; (2) the codebook rows addressed by what (1) returned, together; then the arithmetic, SH first so
    // that its 48 coefficients leave the registers early. As the reference orders them, the same loads form a chain of
    // seven dependent round trips.
    float acc[PARTIAL_FLOATS];
    {
        const float* src = a.partials + (size_t)a.live_slots[pos] * PARTIAL_FLOATS;
#pragma unroll
        for (int q = 0; q < PARTIAL_FLOATS; q++) acc[q] = src[q];
    }
    const float4 rec0 = a.splat[3 * si], rec1 = a.splat[3 * si + 1];
    const f3 m = { a.means3D[3 * si], a.means3D[3 * si + 1], a.means3D[3 * si + 2] };
    const uint8_t cl = a.clamped[i];
    size_t gi = si, row = si;
    float sf = 1.f;
    if (INDEXED) {
        if (a.g_indices) gi = (size_t)a.g_indices[i];
        if (a.scale_factors) sf = a.scale_factors[i];
        if (a.sh_indices) row = (size_t)a.sh_indices[i];
    }
    // level 2
    float sc[3] = { 0, 0, 0 };
    float4 rot = make_float4(1, 0, 0, 0);
    float cov3D[6];
    if (a.cov3D_precomp) {
#pragma unroll
        for (int q = 0; q < 6; q++) cov3D[q] = a.cov3D_precomp[6 * si + q];
    } else {
        sc[0] = a.scales[3 * gi]; sc[1] = a.scales[3 * gi + 1]; sc[2] = a.scales[3 * gi + 2];
        rot = *reinterpret_cast<const float4*>(a.rotations + 4 * gi);
    }
    constexpr int NC = (DEG + 1) * (DEG + 1) * 3;
    float c[NC];
    if (a.sh) {
        const float* shp = a.sh + row * (size_t)a.M * 3;
        if ((a.M * 3) % 4 == 0 && NC % 4 == 0) {
#pragma unroll
            for (int q = 0; q < NC / 4; q++) {
                const float4 v = reinterpret_cast<const float4*>(shp)[q];
                c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NC; q++) c[q] = shp[q];
        }
    }

    // ---- SH (backward.cu:20-139 / backward_indexed.cu:20-201); its share of dL/dmean is added further down, where the
    // reference adds it
    float sh_add[3] = { 0.f, 0.f, 0.f };
    if (a.sh) {
        const float g[3] = { (cl & 1) ? 0.f : acc[0], (cl & 2) ? 0.f : acc[1], (cl & 4) ? 0.f : acc[2] };
        const f3 d0 = { m.x - cam_pos[0], m.y - cam_pos[1], m.z - cam_pos[2] };
        float basis[NB];
        sh_backward<DEG>(c, d0, g, sh_add, basis);                       // the row itself is written cooperatively below
        if (o.dL_dsh) {
#pragma unroll
            for (int k = 0; k < NB; k++) s_basis(threadIdx.x, k) = basis[k];
            s_g[threadIdx.x][0] = g[0]; s_g[threadIdx.x][1] = g[1]; s_g[threadIdx.x][2] = g[2];
            s_row[threadIdx.x] = (int32_t)row;
        }
    }

    // acc = {sum alpha*T*dL_dpix (3), S0, Sx, Sy, Sxx, Sxy, Syy}: moments of w = G*dL_dalpha about the 2D mean
    // (render.hip). The reference's per-pair products (backward.cu:538-554) are linear in them:
    const float k_a = rec0.z, k_b = rec0.w, k_c = rec1.x, opac = rec1.y;
    const float dcol[3] = { acc[0], acc[1], acc[2] };
    const float d2x = -0.5f * (float)a.W * opac * (k_a * acc[4] + k_b * acc[5]);
    const float d2y = -0.5f * (float)a.H * opac * (k_c * acc[5] + k_b * acc[4]);
    const float dcon_x = -0.5f * opac * acc[6], dcon_y = -0.5f * opac * acc[7], dcon_w = -0.5f * opac * acc[8];
    acc[8] = acc[3];                       // dL_dopacity = sum G*dL_dalpha
    if (o.dL_dcolors) { o.dL_dcolors[3 * si] = dcol[0]; o.dL_dcolors[3 * si + 1] = dcol[1]; o.dL_dcolors[3 * si + 2] = dcol[2]; }
    if (o.dL_dmeans2D) { o.dL_dmeans2D[3 * si] = d2x; o.dL_dmeans2D[3 * si + 1] = d2y; o.dL_dmeans2D[3 * si + 2] = 0.f; }
    if (o.dL_dopacity) o.dL_dopacity[si] = acc[8];

    // ---- (b) conic -> cov2D -> cov3D / mean (backward.cu:144-274) in matrix form:
    // A = upper 2x3 of J*R_w2c (reference T[i][j] == A[i][j]); cov2D = A*Sigma*A^T + 0.3*I.
    if (!a.cov3D_precomp)
        cov3d_from_scale_rot(sc[0], sc[1], sc[2], INDEXED ? sf * a.scale_modifier : a.scale_modifier, rot, cov3D);
    const Cov2D cv = cov2d(m, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, cam_view);
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float x_grad_mul = (cv.txtz < -limx || cv.txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (cv.tytz < -limy || cv.tytz > limy) ? 0.f : 1.f;
    const float A[2][3] = { { cv.T.c[0][0], cv.T.c[0][1], cv.T.c[0][2] }, { cv.T.c[1][0], cv.T.c[1][1], cv.T.c[1][2] } };
    const float Sg[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
    const float ca = cv.a, cb = cv.b, cc = cv.c;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6] = { 0, 0, 0, 0, 0, 0 };
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcon_x + 2 * cb * cc * dcon_y + (denom - ca * cc) * dcon_w);
        dL_dc = denom2inv * (-ca * ca * dcon_w + 2 * ca * cb * dcon_y + (denom - ca * cc) * dcon_x);
        dL_db = denom2inv * 2 * (cb * cc * dcon_x - (denom + 2 * cb * cb) * dcon_y + ca * cb * dcon_w);
        const float Gm[2][2] = { { dL_da, 0.5f * dL_db }, { 0.5f * dL_db, dL_dc } };
        float S3[3][3];
#pragma unroll
        for (int u = 0; u < 3; u++)
#pragma unroll
            for (int v = u; v < 3; v++) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int w = 0; w < 2; w++) s += A[q][u] * Gm[q][w] * A[w][v];
                S3[u][v] = s;
            }
        dcov[0] = S3[0][0]; dcov[3] = S3[1][1]; dcov[5] = S3[2][2];
        dcov[1] = 2.f * S3[0][1]; dcov[2] = 2.f * S3[0][2]; dcov[4] = 2.f * S3[1][2];
    }
    if (o.dL_dcov3D)
#pragma unroll
        for (int q = 0; q < 6; q++) o.dL_dcov3D[6 * si + q] = dcov[q];

    float AS[2][3], dA[2][3];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int v = 0; v < 3; v++) AS[q][v] = A[q][0] * Sg[0][v] + A[q][1] * Sg[1][v] + A[q][2] * Sg[2][v];
#pragma unroll
    for (int v = 0; v < 3; v++) {
        dA[0][v] = 2 * AS[0][v] * dL_da + AS[1][v] * dL_db;
        dA[1][v] = 2 * AS[1][v] * dL_dc + AS[0][v] * dL_db;
    }
    const float* view = cam_view;
    const float dJ00 = view[0] * dA[0][0] + view[4] * dA[0][1] + view[8] * dA[0][2];
    const float dJ02 = view[2] * dA[0][0] + view[6] * dA[0][1] + view[10] * dA[0][2];
    const float dJ11 = view[1] * dA[1][0] + view[5] * dA[1][1] + view[9] * dA[1][2];
    const float dJ12 = view[2] * dA[1][0] + view[6] * dA[1][1] + view[10] * dA[1][2];
    const float tz = 1.f / cv.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = x_grad_mul * -a.focal_x * tz2 * dJ02;
    const float dty = y_grad_mul * -a.focal_y * tz2 * dJ12;
    const float dtz = -a.focal_x * tz2 * dJ00 - a.focal_y * tz2 * dJ11 + (2 * a.focal_x * cv.t.x) * tz3 * dJ02 +
                      (2 * a.focal_y * cv.t.y) * tz3 * dJ12;
    float dmean[3] = { view[0] * dtx + view[1] * dty + view[2] * dtz,
                       view[4] * dtx + view[5] * dty + view[6] * dtz,
                       view[8] * dtx + view[9] * dty + view[10] * dtz };

    // ---- (c) projection of the 2D mean (backward.cu:370-387)
    const float* proj = cam_proj;
    const float4 mh = xform4x4(m, proj);
    const float m_w = 1.0f / (mh.w + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    dmean[0] += (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
    dmean[1] += (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
    dmean[2] += (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;

    dmean[0] += sh_add[0]; dmean[1] += sh_add[1]; dmean[2] += sh_add[2];
    if (o.dL_dmeans3D) { o.dL_dmeans3D[3 * si] = dmean[0]; o.dL_dmeans3D[3 * si + 1] = dmean[1]; o.dL_dmeans3D[3 * si + 2] = dmean[2]; }

    // ---- scale / rotation (backward.cu:278-341 / backward_indexed.cu:206-282), against the standard
    // rotation matrix Rm: L = Rm*diag(s), Sigma = L*L^T, dL/dL = 2*G*L.
    if (a.scales) {
        const float s[3] = { (INDEXED ? sf * a.scale_modifier : a.scale_modifier) * sc[0],
                             (INDEXED ? sf * a.scale_modifier : a.scale_modifier) * sc[1],
                             (INDEXED ? sf * a.scale_modifier : a.scale_modifier) * sc[2] };
        const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
        const float Rm[3][3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                                 { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                                 { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } };
        const float G3[3][3] = { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] },
                                 { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] },
                                 { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
        float Q[3][3], d_s[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float col[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float accv = 0.f;
#pragma unroll
                for (int q = 0; q < 3; q++) accv += G3[k][q] * (Rm[q][j] * s[j]);
                col[k] = 2.0f * accv;
            }
            d_s[j] = Rm[0][j] * col[0] + Rm[1][j] * col[1] + Rm[2][j] * col[2];
#pragma unroll
            for (int k = 0; k < 3; k++) Q[k][j] = col[k] * s[j];
        }
        float dq[4];
        dq[0] = 2 * z * (Q[1][0] - Q[0][1]) + 2 * y * (Q[0][2] - Q[2][0]) + 2 * x * (Q[2][1] - Q[1][2]);
        dq[1] = 2 * y * (Q[0][1] + Q[1][0]) + 2 * z * (Q[0][2] + Q[2][0]) + 2 * r * (Q[2][1] - Q[1][2]) - 4 * x * (Q[2][2] + Q[1][1]);
        dq[2] = 2 * x * (Q[0][1] + Q[1][0]) + 2 * r * (Q[0][2] - Q[2][0]) + 2 * z * (Q[2][1] + Q[1][2]) - 4 * y * (Q[2][2] + Q[0][0]);
        dq[3] = 2 * r * (Q[1][0] - Q[0][1]) + 2 * x * (Q[0][2] + Q[2][0]) + 2 * y * (Q[2][1] + Q[1][2]) - 4 * z * (Q[1][1] + Q[0][0]);
        if (INDEXED) {                       // backward_indexed.cu:255-262, 276-281 (scatter-add below)
            s_ds[threadIdx.x][0] = d_s[0] * sf; s_ds[threadIdx.x][1] = d_s[1] * sf; s_ds[threadIdx.x][2] = d_s[2] * sf;
            s_dq[threadIdx.x][0] = dq[0]; s_dq[threadIdx.x][1] = dq[1]; s_dq[threadIdx.x][2] = dq[2]; s_dq[threadIdx.x][3] = dq[3];
            s_gi[threadIdx.x] = (int32_t)gi;
            if (o.dL_dscale_factors) o.dL_dscale_factors[si] = d_s[0] * sc[0] + d_s[1] * sc[1] + d_s[2] * sc[2];
        } else {
            if (o.dL_dscales) { o.dL_dscales[3 * si] = d_s[0]; o.dL_dscales[3 * si + 1] = d_s[1]; o.dL_dscales[3 * si + 2] = d_s[2]; }
            if (o.dL_drotations) *reinterpret_cast<float4*>(o.dL_drotations + 4 * si) = make_float4(dq[0], dq[1], dq[2], dq[3]);
        }
    } else if (INDEXED && o.dL_dscale_factors) {
        o.dL_dscale_factors[si] = 0.f;
    }
    } // live

    {
        // each wave scatter-adds (indexed) or stores (non-indexed) its own 64 rows: wave-level ordering of the LDS traffic is
        // all that is needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
        if (o.dL_dsh && a.sh) {
            const int k = min(lane / 3, NB - 1), ch = lane - 3 * (lane / 3);
            for (int j = 0; j < 64; j++) {
                const int32_t row = s_row[wbase + j];            // wave-uniform
                if (row < 0) continue;
                float* dst = o.dL_dsh + (size_t)row * a.M * 3 + lane;
                const float v = s_basis(wbase + j, k) * s_g[wbase + j][ch];
                if (INDEXED) { if (lane < NB * 3) atomicAdd(dst, v); }
                else if (lane < a.M * 3) *dst = lane < NB * 3 ? v : 0.f;   // coefficients above the active degree: zero
            }
        }
    }
    if (INDEXED) {
        const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
        if (a.scales) {
            if (o.dL_dscales)
#pragma unroll
                for (int it = 0; it < 3; it++) {
                    const int q = it * 64 + lane, j = q / 3, c = q - 3 * j;
                    const int32_t gi = s_gi[wbase + j];
                    if (gi >= 0) atomicAdd(o.dL_dscales + 3 * (size_t)gi + c, s_ds[wbase + j][c]);
                }
            if (o.dL_drotations)
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int q = it * 64 + lane, j = q >> 2, c = q & 3;
                    const int32_t gi = s_gi[wbase + j];
                    if (gi >= 0) atomicAdd(o.dL_drotations + 4 * (size_t)gi + c, s_dq[wbase + j][c]);
                }
        }
    }
}

void launch_backward_preprocess(const c3dgs_raster_params& p, const int32_t* radii, const GeomPtrs& g,
                                float* partials, const uint8_t* touched, uint32_t* live_count, uint32_t* live_ids,
                                uint32_t* live_slots, const c3dgs_raster_grads& gr, hipStream_t s)
{
    if (p.P <= 0) return;
    BwdArgs a;
    a.P = p.P; a.D = p.D; a.M = p.M; a.W = p.W; a.H = p.H;
    a.means3D = p.means3D; a.sh = p.sh; a.scales = p.scales; a.scale_factors = p.scale_factors; a.rotations = p.rotations;
    a.cov3D_precomp = p.cov3D_precomp; a.sh_indices = p.sh_indices; a.g_indices = p.g_indices;
    a.view = p.viewmatrix; a.proj = p.projmatrix; a.campos = p.campos;
    a.tan_fovx = p.tan_fovx; a.tan_fovy = p.tan_fovy;
    a.focal_y = p.H / (2.0f * p.tan_fovy);
    a.focal_x = p.W / (2.0f * p.tan_fovx);
    a.scale_modifier = p.scale_modifier;
    a.radii = radii; a.inst_offset = g.inst_offset; a.block_base = g.block_base; a.clamped = g.clamped; a.splat = g.splat;
    a.partials = partials; a.touched = touched; a.g = gr;
    a.live_count = live_count; a.live_ids = live_ids; a.live_slots = live_slots;
    constexpr int LIST = 64 * C3DGS_SUM_WAVES;
    const int n_lists = (p.P + LIST - 1) / LIST;
    sum_partials_kernel<<<n_lists, LIST, 0, s>>>(a);
    const dim3 grid(n_lists * (LIST / 256)), block(256);
    const bool indexed = p.sh_indices != nullptr || p.g_indices != nullptr;
    const int deg = p.sh ? p.D : 0;
#define C3DGS_LAUNCH(DEG)                                                                     \
    do {                                                                                      \
        if (indexed) backward_preprocess_kernel<DEG, true><<<grid, block, 0, s>>>(a, a.view, a.proj, a.campos);         \
        else backward_preprocess_kernel<DEG, false><<<grid, block, 0, s>>>(a, a.view, a.proj, a.campos);                \
    } while (0)
    switch (deg) {
        case 0: C3DGS_LAUNCH(0); break;
        case 1: C3DGS_LAUNCH(1); break;
        case 2: C3DGS_LAUNCH(2); break;
        default: C3DGS_LAUNCH(3); break;
    }
#undef C3DGS_LAUNCH
}

#undef s_basis

} // namespace c3dgs
