// vq.hip -- nearest-codeword search and the weighted EMA k-means update for gfx950.
//
//   K14  weighted_distance   reference submodules/weighted_distance/weighted_distance.cu:9-58
//   V3   vq_accumulate/apply reference compression/vq.py:28-35,45-46,73-77
//
// weighted_distance keeps the reference's exact semantics: per (point, codeword) a k-ordered fp32
// chain r = fma(d, d, r) with d = x_k - c_k, strict '<' so the lowest index wins ties. The reference
// launches 32-thread blocks with one thread per point and re-reads the whole codebook from global
// memory per thread; here a 256-thread workgroup keeps its points in registers (2 per lane) and
// streams the codebook through LDS in tiles that every lane reads as a broadcast.
#include "common.hpp"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace c3dgs {

constexpr int WD_BLOCK = 256;
constexpr int WD_PPT = 2;        // points per thread

template <int K> struct RowVec { };
template <> struct RowVec<48> { using type = float4; static constexpr int N = 4; };
template <> struct RowVec<12> { using type = float4; static constexpr int N = 4; };
template <> struct RowVec<6>  { using type = float2; static constexpr int N = 2; };

template <int K, int CT>
__global__ void __launch_bounds__(WD_BLOCK)
weighted_distance_kernel(int64_t N, int C, const float* __restrict__ coefs, const int64_t* __restrict__ gather,
                         const float* __restrict__ codebook, float* __restrict__ out_dist, int64_t* __restrict__ out_idx)
{
    using V = typename RowVec<K>::type;
    constexpr int VN = RowVec<K>::N;
    constexpr int KV = K / VN;
    __shared__ V s_cb[CT * KV];

    const int tid = threadIdx.x;
    float x[WD_PPT][K];
    int64_t n[WD_PPT];
    float best[WD_PPT];
    int besti[WD_PPT];
#pragma unroll
    for (int p = 0; p < WD_PPT; p++) {
        n[p] = ((int64_t)blockIdx.x * WD_PPT + p) * WD_BLOCK + tid;
        best[p] = FLT_MAX;
        besti[p] = 0;
        const int64_t row = n[p] < N ? (gather ? gather[n[p]] : n[p]) : 0;
        const V* src = reinterpret_cast<const V*>(coefs + row * K);
#pragma unroll
        for (int q = 0; q < KV; q++) {
            const V v = src[q];
            const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int e = 0; e < VN; e++) x[p][q * VN + e] = f[e];
        }
    }
    for (int c0 = 0; c0 < C; c0 += CT) {
        const int ct = min(CT, C - c0);
        __syncthreads();
        const V* gsrc = reinterpret_cast<const V*>(codebook + (size_t)c0 * K);
        for (int q = tid; q < ct * KV; q += WD_BLOCK) s_cb[q] = gsrc[q];
        __syncthreads();
        for (int c = 0; c < ct; c++) {
            float r[WD_PPT];
#pragma unroll
            for (int p = 0; p < WD_PPT; p++) r[p] = 0.f;
#pragma unroll
            for (int q = 0; q < KV; q++) {
                const V v = s_cb[c * KV + q];
                const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
                for (int e = 0; e < VN; e++)
#pragma unroll
                    for (int p = 0; p < WD_PPT; p++) {
                        const float d = x[p][q * VN + e] - f[e];
                        r[p] = fmaf(d, d, r[p]);
                    }
            }
#pragma unroll
            for (int p = 0; p < WD_PPT; p++)
                if (r[p] < best[p]) { best[p] = r[p]; besti[p] = c0 + c; }
        }
    }
#pragma unroll
    for (int p = 0; p < WD_PPT; p++)
        if (n[p] < N) { out_dist[n[p]] = best[p]; out_idx[n[p]] = (int64_t)besti[p]; }
}

// any K: one thread per point, codebook tile in LDS, point row re-read from global/L1.
template <int CT_FLOATS>
__global__ void __launch_bounds__(WD_BLOCK)
weighted_distance_generic_kernel(int64_t N, int C, int K, const float* __restrict__ coefs, const int64_t* __restrict__ gather,
                                 const float* __restrict__ codebook, float* __restrict__ out_dist, int64_t* __restrict__ out_idx)
{
    __shared__ float s_cb[CT_FLOATS];
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)blockIdx.x * WD_BLOCK + tid;
    const int CT = max(1, CT_FLOATS / K);
    const int64_t row = n < N ? (gather ? gather[n] : n) : 0;
    const float* xr = coefs + row * K;
    float best = FLT_MAX;
    int besti = 0;
    for (int c0 = 0; c0 < C; c0 += CT) {
        const int ct = min(CT, C - c0);
        __syncthreads();
        for (int q = tid; q < ct * K; q += WD_BLOCK) s_cb[q] = codebook[(size_t)c0 * K + q];
        __syncthreads();
        if (n < N)
            for (int c = 0; c < ct; c++) {
                float r = 0.f;
                for (int k = 0; k < K; k++) {
                    const float d = xr[k] - s_cb[c * K + k];
                    r = fmaf(d, d, r);
                }
                if (r < best) { best = r; besti = c0 + c; }
            }
    }
    if (n < N) { out_dist[n] = best; out_idx[n] = (int64_t)besti; }
}


// ---------------------------------------------------------------- MFMA nearest-codeword search (K = 48, 12, 6)
//
// s[c][n] = ||c||^2 - 2 x_n.c  (argmin_c s == argmin_c ||x_n - c||^2) on the fp32 matrix cores:
// v_mfma_f32_32x32x2_f32 with A = 32 codewords x 2 dims (from an LDS tile stored k-major, conflict-free),
// B = 2 dims x 32 points (pre-scaled by -2, resident in registers for the whole kernel), C initialised with
// ||c||^2. In the 32x32 accumulator layout a lane owns ONE point (column = lane&31) and 16 codeword rows, so the
// running (best, second best, index) update is pure per-lane VALU work that overlaps the next MFMAs; the two
// half-waves are merged once at the end.
//
// Exact reference semantics are kept: the winner's distance is recomputed with the reference's k-ordered FMA
// chain, and a point whose two best candidates are closer than the rounding-error bound of BOTH formulations
// (margin = 4e-5*(d_best + d_second + 2||x||^2)) is flagged (index -1) and resolved by
// wd_fixup_kernel with the exact chain over all codewords (lowest index wins ties).
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_CT = 128;          // codewords per LDS tile
constexpr int MF_PTS = 64;          // points per wave (two 32-point B operands)

// 3 VALU operations per distance. (1) The accumulator row (0..15) is written into the 4 low mantissa bits of the value
// (one v_and_or_b32), so the running minimum carries its own row and needs no compare + select; which 32-codeword group
// it came from is noted once per sub-tile. The values move by < 16 ulp = 1.9e-6 relative, which the ambiguity margin
// (4e-5, against a rigorous need of ~2.6e-5) covers -- a wrong pick can only happen inside the window,
// and everything inside the window is re-decided exactly. (2) With best <= second the new second smallest of
// {v, best, second} is their median: one v_med3_f32. The update is what the search kernel spends its VALU time on
// (a top-3 variant with 10 operations ran 14-50 % slower).
__device__ __forceinline__ void top2_update(float v, uint32_t code, uint32_t keep_mask, float& best, float& second)
{
    const float vp = __uint_as_float((__float_as_uint(v) & keep_mask) | code);
    second = __builtin_amdgcn_fmed3f(vp, best, second);
    // a bare v_min_f32 (NaN operand -> the other one, as fminf): fminf() itself costs a canonicalising v_max per operand
    // in IEEE mode, which made the update 5 instructions instead of 3 together with the accumulator read-back that
    // -amdgpu-mfma-vgpr-form removes (c3dgs_amd/build.py)
    float nb;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(nb) : "v"(vp), "v"(best));
    best = nb;
}

// SPLIT = false: the 4 waves of a workgroup own 64 points each and every wave scans the whole codebook tile.
// SPLIT = true (small N, e.g. a rank's slice of a sharded Lloyd batch: 2^18 / 8 points are only 128 such workgroups on
// 256 CUs): the 4 waves share ONE set of 64 points and split every 128-codeword tile into its four 32-codeword sub-tiles
// (wave w takes sub-tile w), i.e. 4x as many workgroups for the same N; their (best, second, index) triples are merged
// through LDS at the end. Same candidate values, same tie rule (lowest index) -> bit-identical results.
template <int MF_K, bool SPLIT>
__global__ void __launch_bounds__(256)
wd_mfma_kernel(int64_t N, int C, const float* __restrict__ coefs, const int64_t* __restrict__ gather,
               const float* __restrict__ codebook, float* __restrict__ out_dist, int64_t* __restrict__ out_idx,
               int* __restrict__ flag_list, int flag_cap)
{
    static_assert(MF_K % 2 == 0, "two dims per v_mfma_f32_32x32x2_f32 step");
    static_assert(MF_CT / 32 == 4, "SPLIT hands one sub-tile of a tile to each of the 4 waves");
    __shared__ float s_cb[2][MF_K][MF_CT];   // k-major tile: lane i reads s_cb[k][i] (consecutive banks)
    __shared__ float s_norm[2][MF_CT];
    __shared__ float s_mb[SPLIT ? 4 : 1][2][32], s_ms[SPLIT ? 4 : 1][2][32];
    __shared__ int s_mi[SPLIT ? 4 : 1][2][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t n_base = SPLIT ? (int64_t)blockIdx.x * MF_PTS : ((int64_t)blockIdx.x * 4 + wave) * MF_PTS;

    // B operands: b[g][t] = -2 * x[n_base + 32g + i][2t + h]
    float b[2][MF_K / 2];
    float xnorm[2];
    int64_t rows[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int64_t n = n_base + 32 * g + i;
        const int64_t row = n < N ? (gather ? gather[n] : n) : 0;
        rows[g] = row;
        const float2* src = reinterpret_cast<const float2*>(coefs + row * MF_K);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < MF_K / 2; t++) {
            const float2 v = src[t];
            b[g][t] = -2.0f * (h ? v.y : v.x);
            part = fmaf(v.x, v.x, part);
            part = fmaf(v.y, v.y, part);
        }
        xnorm[g] = part;
    }
    // start value: FLT_MAX with the row bits clear, so a point that never finds a smaller value (all-NaN input) decodes to
    // codeword 0 like the reference's `min_idx = 0` start
    const float big = __uint_as_float(0x7f7ffff0u);
    float best[2] = { big, big }, second[2] = { FLT_MAX, FLT_MAX };
    int grp[2] = { 0, 0 };                           // 32-codeword group of the current best (its row: low 4 bits of best)
    const uint32_t keep_mask = 0xfffffff0u;

    const int ntiles = (C + MF_CT - 1) / MF_CT;
    auto stage = [&](int tile, int buf) {
        if constexpr (MF_K % 8 == 0) {
            // one codeword per thread pair: thread t stages codeword (t & 127), dims [K/2*(t>>7), K/2*(t>>7)+K/2)
            const int c = tid & (MF_CT - 1), half = tid >> 7;
            const int cg = tile * MF_CT + c;
            const float4* src = reinterpret_cast<const float4*>(codebook + (size_t)(cg < C ? cg : 0) * MF_K + (MF_K / 2) * half);
#pragma unroll
            for (int q = 0; q < MF_K / 8; q++) {
                float4 v = src[q];
                if (cg >= C) v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int k = (MF_K / 2) * half + 4 * q;
                s_cb[buf][k][c] = v.x; s_cb[buf][k + 1][c] = v.y; s_cb[buf][k + 2][c] = v.z; s_cb[buf][k + 3][c] = v.w;
            }
        } else if (tid < MF_CT) {                  // small K: one codeword per thread, 8-byte loads
            const int cg = tile * MF_CT + tid;
            const float2* src = reinterpret_cast<const float2*>(codebook + (size_t)(cg < C ? cg : 0) * MF_K);
#pragma unroll
            for (int q = 0; q < MF_K / 2; q++) {
                float2 v = src[q];
                if (cg >= C) v = make_float2(0.f, 0.f);
                s_cb[buf][2 * q][tid] = v.x; s_cb[buf][2 * q + 1][tid] = v.y;
            }
        }
    };
    stage(0, 0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        if (tid < MF_CT) {          // ||c||^2 of this tile (rows past C get +inf-like so they never win)
            float nr = 0.f;
#pragma unroll
            for (int k = 0; k < MF_K; k++) nr = fmaf(s_cb[buf][k][tid], s_cb[buf][k][tid], nr);
            s_norm[buf][tid] = (tile * MF_CT + tid < C) ? nr : 3.0e38f;
        }
        if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
        __syncthreads();
#pragma unroll 1
        for (int sub = SPLIT ? wave : 0; sub < (SPLIT ? wave + 1 : MF_CT / 32); sub++) {
            if (tile * MF_CT + sub * 32 >= C) break;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float nv = s_norm[buf][sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                acc0[r] = nv; acc1[r] = nv;
            }
#pragma unroll
            for (int t = 0; t < MF_K / 2; t++) {
                const float a = s_cb[buf][2 * t + h][sub * 32 + i];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0][t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[1][t], acc1, 0, 0, 0);
            }
            const float was0 = best[0], was1 = best[1];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                top2_update(acc0[r], (uint32_t)r, keep_mask, best[0], second[0]);
                top2_update(acc1[r], (uint32_t)r, keep_mask, best[1], second[1]);
            }
            const int gid = tile * (MF_CT / 32) + sub;
            grp[0] = best[0] < was0 ? gid : grp[0];
            grp[1] = best[1] < was1 ? gid : grp[1];
        }
        __syncthreads();
    }
    // merge the two half-waves (same point in lanes l and l^32), decide, recompute the exact distance
#pragma unroll
    for (int g = 0; g < 2; g++) {
        // accumulator row r of lane half h in group G is codeword 32 G + (r & 3) + 8 (r >> 2) + 4 h
        const int code = (int)(__float_as_uint(best[g]) & 15u);
        const int my_idx = grp[g] * 32 + (code & 3) + 8 * (code >> 2) + 4 * h;
        const float ob = __shfl_xor(best[g], 32), os = __shfl_xor(second[g], 32);
        const int oi = __shfl_xor(my_idx, 32);
        float nb = fminf(best[g], ob);
        float ns = fminf(fminf(second[g], os), fmaxf(best[g], ob));
        int ni = (ob < best[g] || (ob == best[g] && oi < my_idx)) ? oi : my_idx;
        if (SPLIT) {                                  // merge the four waves' triples (same points, disjoint codewords)
            if (h == 0) { s_mb[wave][g][i] = nb; s_ms[wave][g][i] = ns; s_mi[wave][g][i] = ni; }
            __syncthreads();
            if (wave != 0) continue;
            nb = s_mb[0][g][i]; ns = s_ms[0][g][i]; ni = s_mi[0][g][i];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const float wb = s_mb[w][g][i], ws = s_ms[w][g][i];
                const int wi = s_mi[w][g][i];
                ns = fminf(fminf(ns, ws), fmaxf(nb, wb));
                const bool take = wb < nb || (wb == nb && wi < ni);
                nb = fminf(nb, wb);
                ni = take ? wi : ni;
            }
        }
        const int64_t n = n_base + 32 * g + i;
        if (h == 0 && n < N) {
            const float db = nb + xnorm[g], ds = ns + xnorm[g];
            const float margin = 4e-5f * (fabsf(db) + fabsf(ds) + 2.0f * xnorm[g]) + 1e-37f;
            const bool ambiguous = !(ns - nb > margin);
            const float* x = coefs + rows[g] * MF_K;
            const float* cb = codebook + (size_t)ni * MF_K;
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < MF_K; k++) {
                const float d = x[k] - cb[k];
                r = fmaf(d, d, r);
            }
            out_dist[n] = r;
            out_idx[n] = ambiguous ? (int64_t)-1 : (int64_t)ni;
            if (ambiguous && flag_list) {             // optional list of the flagged points: [0] = count, then the points
                const int pos = atomicAdd(&flag_list[0], 1);
                if (pos < flag_cap) flag_list[1 + pos] = (int)n;
            }
        }
    }
}

// ---------------------------------------------------------------- the same search on the fp16 matrix cores (split operands)
//
// v_mfma_f32_32x32x16_f16 runs 16x the multiply-adds per cycle of the fp32 form. With v = h + l + r, h = fp16(v),
// l = fp16(v - h) (2 x 11 significant bits, |r| <= 2^-22 |v|), a product x c is reproduced to 3 * 2^-22 relative by the
// three piece products
//     xh ch + xh cl + xl ch                     (dropped: xl cl and the two split residuals)
// each of which is EXACT in the matrix core's fp32 product stage (11 x 11 bits): three fp16 MFMAs replace eight fp32 ones
// per 16 dimensions, 9 x 32 cycles instead of 24 x 64 per 32 x 32 distances at K = 48. (Three bf16 pieces and six products
// give the same accuracy for twice the matrix time: 0.50 ms per 2^18 x 4096 x 48 batch, measured.)
// fp16 has a narrow exponent range, so both operands are scaled by ONE power of two 2^e chosen per call from the
// codebook's largest magnitude (it lands in [2^10, 2^11)); scores are compared per point, so a common exact scale changes no
// decision. Elements far below the largest lose their low piece to fp16's subnormal spacing: an absolute error of 2^-35
// of the largest element, nothing against the margin. Points more than ~16x beyond the codebook's largest magnitude
// overflow to inf / NaN scores, are flagged ambiguous by the margin test and take the exact re-scan like any other flagged
// point: the search is never the exact part of this path -- the winner's distance is recomputed with the reference's
// k-ordered chain and every point whose two best candidates lie within the error margin is re-scanned exactly -- so
// only the margin has to cover the approximation: the dropped terms (<= 7.2e-7 sum|x_k c_k|), the accumulation inside and
// between the 9 MFMAs (measured through c3dgs_debug_wd_scores, tests/test_vq_gpu.py) and top2_update's 16-ulp row packing.
// The codebook is split ONCE per call by wd_split_codebook_kernel into MFMA A-operand fragments
//     frag[tile][sub-tile (32 codewords)][k-step (16 dims)][piece][lane][8 x fp16]    (lane l: codeword l & 31, dims 8 (l >> 5) ..+7)
// so staging a tile is a plain 16-byte copy and a wave's operand read is one conflict-free ds_read_b128.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int HF_PIECES = 2;

// global -> LDS copies without a register stop (global_load_lds_dwordx4 / _dword): lane l's bytes land at lds_wave_base + l * SIZE.
// Device pass only: with the builtin inside a kernel TEMPLATE, hipcc (ROCm 7.2) silently drops the kernel's host-side launch
// stub from the object (undefined __device_stub__ at load time), so the host pass sees empty helpers.
__device__ __forceinline__ void lds_dma16(const void* g, void* lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ void lds_dma4(const void* g, void* lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
#endif
}
template <int MF_K> struct HfShape {
    static constexpr int KS = (MF_K + 15) / 16;                                  // k-steps of 16 dims (zero-padded)
    static constexpr int SUB_BYTES = KS * HF_PIECES * 64 * 16;                   // one 32-codeword sub-tile
    static constexpr int TILE_BYTES = (MF_CT / 32) * SUB_BYTES;                  // 24,576 B at K = 48
};

__device__ __forceinline__ void f16_split2(float v, _Float16& h, _Float16& l)
{
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);           // the difference is exact in fp32
}

// scale exponent from the largest |c| (its fp32 bits, gathered with atomicMax): max|c| * 2^e in [2^10, 2^11)
__device__ __forceinline__ int wd_scale_exp(uint32_t absmax_bits)
{
    const int ex = (int)(absmax_bits >> 23);                  // biased exponent; 0 = zero / subnormal, 255 = inf / NaN
    if (ex == 0 || ex == 255) return 0;
    return max(-100, min(100, 10 - (ex - 127)));
}

size_t wd_split_bytes(int C, int K)
{
    const int ks = (K + 15) / 16;
    const size_t ntiles = ((size_t)C + MF_CT - 1) / MF_CT;
    return ntiles * (size_t)(MF_CT / 32) * ks * HF_PIECES * 64 * 16 + ntiles * MF_CT * sizeof(float) + 16;   // + the scale word
}

__global__ void __launch_bounds__(256) wd_absmax_kernel(size_t n, const float* __restrict__ v, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_m[4];
    uint32_t m = 0;
    auto take = [&](float f) { const uint32_t b = __float_as_uint(f) & 0x7fffffffu; m = max(m, b <= 0x7f800000u ? b : 0u); };   // NaNs do not set the scale
    const size_t n4 = (((uintptr_t)v) & 15) == 0 ? n / 4 : 0;
    const float4* v4 = reinterpret_cast<const float4*>(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 f = v4[i];
        take(f.x); take(f.y); take(f.z); take(f.w);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) take(v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(out, m);
    }
}

// one thread per (tile, sub-tile, k-step, lane): 8 dims of one codeword -> its two 16-byte fragments;
// the first threads also compute ||c||^2 (k-ordered FMA chain, as the fp32 kernel does per tile), scaled by 2^2e
template <int MF_K>
__global__ void __launch_bounds__(256)
wd_split_codebook_kernel(int C, const float* __restrict__ codebook, uint4* __restrict__ frag, float* __restrict__ norms,
                         const uint32_t* __restrict__ absmax)
{
    constexpr int KS = HfShape<MF_K>::KS;
    const int ntiles = (C + MF_CT - 1) / MF_CT;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float sc = __builtin_ldexpf(1.0f, wd_scale_exp(*absmax));
    if (t < ntiles * MF_CT) {
        float nr = 3.0e38f;                                  // rows past C never win
        if (t < C) {
            nr = 0.f;
#pragma unroll
            for (int k = 0; k < MF_K; k++) nr = fmaf(codebook[(size_t)t * MF_K + k], codebook[(size_t)t * MF_K + k], nr);
            nr = (nr * sc) * sc;                             // exact: power of two, (max|c| 2^e)^2 K << FLT_MAX
        }
        norms[t] = nr;
    }
    if (t >= ntiles * (MF_CT / 32) * KS * 64) return;
    const int lane = t & 63, q = (t >> 6) % KS, ts = (t >> 6) / KS;      // ts = tile * 4 + sub
    const int cw = ts * 32 + (lane & 31), k0 = 16 * q + 8 * (lane >> 5);
    union { f16x8 v; uint4 u; } p[HF_PIECES];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float v = (cw < C && k0 + j < MF_K) ? codebook[(size_t)cw * MF_K + k0 + j] * sc : 0.f;
        _Float16 h, l;
        f16_split2(v, h, l);
        p[0].v[j] = h; p[1].v[j] = l;
    }
#pragma unroll
    for (int e = 0; e < HF_PIECES; e++) frag[((size_t)(ts * KS + q) * HF_PIECES + e) * 64 + lane] = p[e].u;
}

// SCORES (diagnostics, one workgroup's worth of points): also dumps s[n][c] = ||c||^2 - 2 x_n.c as the matrix cores produced it
// (unscaled), so a test can measure the approximation error the margin has to cover.
template <int MF_K, bool SPLIT, bool SCORES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))   // 173 VGPRs (two accumulator pairs); three waves per SIMD with fragments loaded per k-step measured the same
wd_f16_kernel(int64_t N, int C, const float* __restrict__ coefs, const int64_t* __restrict__ gather,
              const float* __restrict__ codebook, const uint4* __restrict__ frag, const float* __restrict__ norms,
              const uint32_t* __restrict__ absmax,
               float* __restrict__ out_dist, int64_t* __restrict__ out_idx, int* __restrict__ flag_list, int flag_cap,
               float margin_rel, float* __restrict__ scores, const uint32_t* __restrict__ absmax_true = nullptr)
{
    constexpr int KS = HfShape<MF_K>::KS;
    constexpr int TILE_V = HfShape<MF_K>::TILE_BYTES / 16, SUB_V = HfShape<MF_K>::SUB_BYTES / 16;
    constexpr int STAGE_PER_THREAD = (TILE_V + 255) / 256;
    __shared__ uint4 s_frag[2][TILE_V];
    __shared__ float s_norm[2][MF_CT];
    __shared__ float s_mb[SPLIT ? 4 : 1][2][32], s_ms[SPLIT ? 4 : 1][2][32];
    __shared__ int s_mi[SPLIT ? 4 : 1][2][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t n_base = SPLIT ? (int64_t)blockIdx.x * MF_PTS : ((int64_t)blockIdx.x * 4 + wave) * MF_PTS;

    // B operands: the two pieces of -2 * 2^e * x[n_base + 32 g + i][16 q + 8 h + j], j = 0..7
    const int sexp = wd_scale_exp(*absmax);
    const float sc2 = -2.0f * __builtin_ldexpf(1.0f, sexp), unscale = __builtin_ldexpf(1.0f, -sexp);
    // Fused Lloyd step (vq_apply_split_kernel): the fragments were scaled with the exponent of the PREVIOUS codebook's largest
    // magnitude; absmax_true is the current codebook's. The two normally differ by at most one binade. Should the codebook have
    // grown 16x or shrunk 64x within one EMA step, the split's error model no longer holds: every point is then declared
    // ambiguous and decided by the exact re-scan (slow, never wrong).
    bool scale_ok = true;
    if (absmax_true) { const int drift = sexp - wd_scale_exp(*absmax_true); scale_ok = drift < 4 && drift > -6; }
    f16x8 bh[2][KS], bl[2][KS];
    float xnorm[2];
    int64_t rows[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int64_t n = n_base + 32 * g + i;
        const int64_t row = n < N ? (gather ? gather[n] : n) : 0;
        rows[g] = row;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < KS; q++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 16 * q + 8 * h + j;
                const float v = k < MF_K ? coefs[row * MF_K + k] : 0.f;
                part = fmaf(v, v, part);
                _Float16 ph, pl;
                f16_split2(sc2 * v, ph, pl);
                bh[g][q][j] = ph; bl[g][q][j] = pl;
            }
        }
        xnorm[g] = part + __shfl_xor(part, 32);
    }
    const float big = __uint_as_float(0x7f7ffff0u);
    float best[2] = { big, big }, second[2] = { FLT_MAX, FLT_MAX };
    int grp[2] = { 0, 0 };
    const uint32_t keep_mask = 0xfffffff0u;

    const int ntiles = (C + MF_CT - 1) / MF_CT;
    static_assert(TILE_V % 256 == 0, "whole 16-byte vectors per thread");
    // staging: global -> LDS directly (global_load_lds_dwordx4: no staging registers; the LDS image is lane-linear, i.e. a wave's
    // 64 x 16 bytes land at a wave-uniform base + lane * 16, exactly the fragment order wd_split_codebook_kernel wrote)
#define C3DGS_HF_STAGE(tile_, buf_)                                                                                       \
    {                                                                                                                     \
        const uint4* src_ = frag + (size_t)(tile_) * TILE_V + tid;                                                        \
        _Pragma("unroll") for (int e = 0; e < STAGE_PER_THREAD; e++)                                                      \
            lds_dma16(src_ + e * 256, &s_frag[buf_][e * 256 + wave * 64]);                                                \
        if (wave < MF_CT / 64)                                                                                            \
            lds_dma4(norms + (size_t)(tile_) * MF_CT + tid, &s_norm[buf_][wave * 64]);                                    \
    }
    C3DGS_HF_STAGE(0, 0);
    __syncthreads();
#define C3DGS_HF_INIT(A0, A1, buf_, sub_)                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                                      \
        const float nv_ = s_norm[buf_][(sub_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];                                     \
        A0[r] = nv_; A1[r] = nv_;                                                                                         \
    }
#define C3DGS_HF_UPD(P0, P1, r_)                                                                                          \
    top2_update(P0[r_], (uint32_t)(r_), keep_mask, best[0], second[0]);                                                   \
    top2_update(P1[r_], (uint32_t)(r_), keep_mask, best[1], second[1]);
    // rows past C hold zeros with a norm of 3e38 (wd_split_codebook_kernel): they lose against every real codeword, so the
    // last tile needs no special case. The pipeline starts on a sub-tile's worth of FLT_MAX scores, which cannot displace
    // the start value of `best` (so a point whose real scores are all NaN keeps group 0, row 0 = codeword 0).
    f32x16 pa0, pa1, pb0, pb1;
#pragma unroll
    for (int r = 0; r < 16; r++) { pb0[r] = FLT_MAX; pb1[r] = FLT_MAX; }
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) C3DGS_HF_STAGE(tile + 1, buf ^ 1);   // in flight behind this tile's MFMAs; the buffer was last read before the previous barrier
        if constexpr (SPLIT || SCORES) {
#pragma unroll 1
            for (int sub = SPLIT ? wave : 0; sub < (SPLIT ? wave + 1 : MF_CT / 32); sub++) {
                C3DGS_HF_INIT(pa0, pa1, buf, sub);
                const uint4* fsub = &s_frag[buf][sub * SUB_V + lane];
#pragma unroll
                for (int q = 0; q < KS; q++) {
                    union { uint4 u; f16x8 v; } ah, al;
                    ah.u = fsub[(q * HF_PIECES + 0) * 64];
                    al.u = fsub[(q * HF_PIECES + 1) * 64];
                    pa0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh[0][q], pa0, 0, 0, 0);
                    pa1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh[1][q], pa1, 0, 0, 0);
                    pa0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl[0][q], pa0, 0, 0, 0);
                    pa1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl[1][q], pa1, 0, 0, 0);
                    pa0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh[0][q], pa0, 0, 0, 0);
                    pa1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh[1][q], pa1, 0, 0, 0);
                }
                if (SCORES) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int c = tile * MF_CT + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int64_t n0 = n_base + i, n1 = n_base + 32 + i;
                        if (c < C && n0 < N) scores[n0 * C + c] = (pa0[r] * unscale) * unscale;
                        if (c < C && n1 < N) scores[n1 * C + c] = (pa1[r] * unscale) * unscale;
                    }
                }
                const float was0 = best[0], was1 = best[1];
#pragma unroll
                for (int r = 0; r < 16; r++) { C3DGS_HF_UPD(pa0, pa1, r); }
                const int gid = tile * (MF_CT / 32) + sub;
                grp[0] = best[0] < was0 ? gid : grp[0];
                grp[1] = best[1] < was1 ? gid : grp[1];
            }
        } else {
            // One sub-tile = 32 codewords x 64 points: 6 KS MFMAs (per k-step {cl xh, ch xl, ch xh} for both 32-point groups,
            // smallest pieces first) and 32 x 3 vector instructions of top-2 update. An MFMA leaves 24 of its 32 cycles of
            // vector issue free, so the update of the PREVIOUS sub-tile's scores (PRV) is interleaved into this sub-tile's
            // MFMAs (ACC): one accumulator row of both groups = six vector instructions behind each of the first 16 MFMAs. The
            // placement is held by an empty asm that ties the MFMA's accumulator to the running top-2 (otherwise instruction
            // selection hoists the 32 row packings and 32 minima in front of the medians and spills 60 registers) plus a
            // sched_barrier per group; two accumulator pairs alternate, nothing is copied. Measured per 2^18 x 4096 x 48 batch
            // (MI355X, ~1.6 GHz under this load): 0.325 ms un-pipelined (MFMAs, then the update, other waves filling in),
            // 0.281 pipelined, 0.272 with LDS-DMA staging; the MFMAs alone 0.247, the updates alone 0.140 (temporary ablation
            // builds) -- i.e. what is left is the matrix pipe at the clock the chip holds.
#define C3DGS_HF_STEP(ACC0, ACC1, PRV0, PRV1, sub_, gid_prev_)                                                            \
            {                                                                                                             \
                C3DGS_HF_INIT(ACC0, ACC1, buf, sub_);                                                                     \
                const f16x8* fsub_ = reinterpret_cast<const f16x8*>(&s_frag[buf][(sub_) * SUB_V + lane]);                 \
                f16x8 ah_[KS], al_[KS];                                                                                   \
                _Pragma("unroll") for (int q = 0; q < KS; q++) {                                                          \
                    ah_[q] = fsub_[(q * HF_PIECES + 0) * 64];                                                             \
                    al_[q] = fsub_[(q * HF_PIECES + 1) * 64];                                                             \
                }                                                                                                         \
                const float was0_ = best[0], was1_ = best[1];                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                        \
                _Pragma("unroll") for (int q = 0; q < KS; q++) {                                                          \
                    _Pragma("unroll") for (int j = 0; j < 6; j++) {                                                       \
                        const f16x8 a_ = j < 2 ? al_[q] : ah_[q];                                                         \
                        const f16x8 b_ = (j >> 1) == 1 ? bl[j & 1][q] : bh[j & 1][q];                                     \
                        if (j & 1) ACC1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, ACC1, 0, 0, 0);                  \
                        else ACC0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, ACC0, 0, 0, 0);                        \
                        const int r_ = q * 6 + j;                                                                         \
                        if (r_ < 16) { C3DGS_HF_UPD(PRV0, PRV1, r_); }                                                    \
                        /* ordering fence (no instruction): ties this MFMA's accumulator and the running top-2 together, */ \
                        /* so instruction selection cannot move the update away from its MFMA                            */ \
                        if (j & 1) asm volatile("" : "+v"(ACC1), "+v"(best[0]), "+v"(second[0]), "+v"(best[1]), "+v"(second[1])); \
                        else asm volatile("" : "+v"(ACC0), "+v"(best[0]), "+v"(second[0]), "+v"(best[1]), "+v"(second[1])); \
                        __builtin_amdgcn_sched_barrier(0);                                                                \
                    }                                                                                                     \
                }                                                                                                         \
                if (6 * KS < 16) { _Pragma("unroll") for (int r_ = 6 * KS; r_ < 16; r_++) { C3DGS_HF_UPD(PRV0, PRV1, r_); } } \
                grp[0] = best[0] < was0_ ? (gid_prev_) : grp[0];                                                          \
                grp[1] = best[1] < was1_ ? (gid_prev_) : grp[1];                                                          \
            }
            const int g0 = tile * (MF_CT / 32);
            C3DGS_HF_STEP(pa0, pa1, pb0, pb1, 0, g0 - 1);   // g0 - 1 = -1 on the first tile: FLT_MAX scores, never taken
            C3DGS_HF_STEP(pb0, pb1, pa0, pa1, 1, g0 + 0);
            C3DGS_HF_STEP(pa0, pa1, pb0, pb1, 2, g0 + 1);
            C3DGS_HF_STEP(pb0, pb1, pa0, pa1, 3, g0 + 2);
        }
        __syncthreads();                                       // drains the LDS-DMA (vmcnt) first
    }
    if constexpr (!(SPLIT || SCORES)) {
        const float was0 = best[0], was1 = best[1];
#pragma unroll
        for (int r = 0; r < 16; r++) { C3DGS_HF_UPD(pb0, pb1, r); }
        const int gid = ntiles * (MF_CT / 32) - 1;
        grp[0] = best[0] < was0 ? gid : grp[0];
        grp[1] = best[1] < was1 ? gid : grp[1];
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int code = (int)(__float_as_uint(best[g]) & 15u);
        const int my_idx = grp[g] * 32 + (code & 3) + 8 * (code >> 2) + 4 * h;
        const float ob = __shfl_xor(best[g], 32), os = __shfl_xor(second[g], 32);
        const int oi = __shfl_xor(my_idx, 32);
        float nb = fminf(best[g], ob);
        float ns = fminf(fminf(second[g], os), fmaxf(best[g], ob));
        int ni = (ob < best[g] || (ob == best[g] && oi < my_idx)) ? oi : my_idx;
        if (SPLIT) {
            if (h == 0) { s_mb[wave][g][i] = nb; s_ms[wave][g][i] = ns; s_mi[wave][g][i] = ni; }
            __syncthreads();
            if (wave != 0) continue;
            nb = s_mb[0][g][i]; ns = s_ms[0][g][i]; ni = s_mi[0][g][i];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const float wb = s_mb[w][g][i], ws = s_ms[w][g][i];
                const int wi = s_mi[w][g][i];
                ns = fminf(fminf(ns, ws), fmaxf(nb, wb));
                const bool take = wb < nb || (wb == nb && wi < ni);
                nb = fminf(nb, wb);
                ni = take ? wi : ni;
            }
        }
        const int64_t n = n_base + 32 * g + i;
        if (h == 0 && n < N) {
            nb = (nb * unscale) * unscale; ns = (ns * unscale) * unscale;   // exact powers of two
            const float db = nb + xnorm[g], ds = ns + xnorm[g];
            const float margin = margin_rel * (fabsf(db) + fabsf(ds) + 2.0f * xnorm[g]) + 1e-37f;
            const bool ambiguous = !(ns - nb > margin) || !scale_ok;
            ni = min(max(ni, 0), C - 1);                   // a padded row can only come out of NaN / inf scores, which are re-scanned anyway
            const float* x = coefs + rows[g] * MF_K;
            const float* cb = codebook + (size_t)ni * MF_K;
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < MF_K; k++) {
                const float d = x[k] - cb[k];
                r = fmaf(d, d, r);
            }
            out_dist[n] = r;
            out_idx[n] = ambiguous ? (int64_t)-1 : (int64_t)ni;
            if (ambiguous && flag_list) {
                const int pos = atomicAdd(&flag_list[0], 1);
                if (pos < flag_cap) flag_list[1 + pos] = (int)n;
            }
        }
    }
}

// exact re-scan of the flagged points. A workgroup owns 64 consecutive points; for every flagged one its 256 threads
// split the codewords (thread t takes t, t+256, ...: k-ordered FMA chain per codeword, strict '<' per thread keeps its
// lowest index), then a wave + LDS argmin that prefers the lower index on ties (== the sequential strict '<').
// (One wave per flagged point took 60-140 us per Lloyd step for ~1 % flagged points; the flagged points are few but each
// costs a full codebook pass, so the latency of one pass is what the launch lasts.)
template <int K>
__global__ void __launch_bounds__(256)
wd_fixup_kernel(int64_t N, int C, const float* __restrict__ coefs, const int64_t* __restrict__ gather,
                const float* __restrict__ codebook, float* __restrict__ out_dist, int64_t* __restrict__ out_idx)
{
    __shared__ float s_best[4];
    __shared__ int s_besti[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    if (n0 >= N) return;
    const int64_t n = n0 + lane;
    unsigned long long m = __ballot(n < N && out_idx[n] < 0);      // every wave computes the same mask
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int64_t np = n0 + l;
        const int64_t row = gather ? gather[np] : np;
        const float* x = coefs + row * K;
        float xr[K];
#pragma unroll
        for (int k = 0; k < K; k++) xr[k] = x[k];
        float best = FLT_MAX;
        int besti = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {
            const float* cb = codebook + (size_t)c * K;
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const float d = xr[k] - cb[k];
                r = fmaf(d, d, r);
            }
            if (r < best) { best = r; besti = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(besti, o);
            if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        __syncthreads();                                             // previous round's s_best has been read
        if (lane == 0) { s_best[wave] = best; s_besti[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < 4; w++)
                if (s_best[w] < best || (s_best[w] == best && s_besti[w] < besti)) { best = s_best[w]; besti = s_besti[w]; }
            out_dist[np] = best;
            out_idx[np] = (int64_t)(besti == 0x7fffffff ? 0 : besti);
        }
    }
}

// exact re-scan of LISTED flagged points, WD_FB of them per codebook pass: the cost of a re-scan is reading the codebook
// (786 KB from L2 for K = 4096 x 48), so a workgroup loads every codeword row once into registers and runs the k-ordered
// FMA chain against WD_FB points whose rows sit in LDS (broadcast reads). Same result as wd_fixup_kernel (which still runs
// afterwards and picks up whatever did not fit the list).
// WD_FT threads per workgroup: a thread's share of the codebook is a serial chain of row loads (786 KB per group from L2, latency
// bound -- 16 rows per thread with 256 threads took 23-28 us whatever the list's length, a quarter of a rank's 2^15-point Lloyd
// step); 512 threads walk 8 rows each (1024 threads leave 128 registers each: the row prefetch spills).
constexpr int WD_FB = 4, WD_FT = 512;
template <int K>
__global__ void __launch_bounds__(WD_FT)
wd_fixup_list_kernel(int C, const float* __restrict__ coefs, const int64_t* __restrict__ gather, const float* __restrict__ codebook,
                     float* __restrict__ out_dist, int64_t* __restrict__ out_idx, const int* __restrict__ flag_list, int flag_cap)
{
    constexpr int NW = WD_FT / 64;
    __shared__ float s_x[WD_FB][K];
    __shared__ float s_best[WD_FB][NW];
    __shared__ int s_besti[WD_FB][NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int count = min(flag_list[0], flag_cap);
    for (int base = (int)blockIdx.x * WD_FB; base < count; base += (int)gridDim.x * WD_FB) {
        const int np = min(WD_FB, count - base);
        __syncthreads();                                             // previous group's LDS has been read
        for (int q = tid; q < WD_FB * K; q += WD_FT) {
            const int p = q / K, k = q - p * K;
            float v = 0.f;
            if (p < np) {
                const int64_t n = flag_list[1 + base + p];
                v = coefs[(gather ? gather[n] : n) * K + k];
            }
            s_x[p][k] = v;
        }
        __syncthreads();
        float best[WD_FB];
        int besti[WD_FB];
#pragma unroll
        for (int p = 0; p < WD_FB; p++) { best[p] = FLT_MAX; besti[p] = 0x7fffffff; }
        for (int c = tid; c < C; c += WD_FT) {
            float cb[K];
            const float* src = codebook + (size_t)c * K;
#pragma unroll
            for (int k = 0; k < K; k++) cb[k] = src[k];
#pragma unroll
            for (int p = 0; p < WD_FB; p++) {
                float r = 0.f;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const float d = s_x[p][k] - cb[k];
                    r = fmaf(d, d, r);
                }
                if (r < best[p]) { best[p] = r; besti[p] = c; }
            }
        }
#pragma unroll
        for (int p = 0; p < WD_FB; p++) {
            float b = best[p];
            int bi = besti[p];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(b, o);
                const int oi = __shfl_xor(bi, o);
                if (ob < b || (ob == b && oi < bi)) { b = ob; bi = oi; }
            }
            if (lane == 0) { s_best[p][wave] = b; s_besti[p][wave] = bi; }
        }
        __syncthreads();
        if (tid < np) {
            float b = s_best[tid][0];
            int bi = s_besti[tid][0];
#pragma unroll
            for (int w = 1; w < NW; w++)
                if (s_best[tid][w] < b || (s_best[tid][w] == b && s_besti[tid][w] < bi)) { b = s_best[tid][w]; bi = s_besti[tid][w]; }
            const int64_t n = flag_list[1 + base + tid];
            out_dist[n] = b;
            out_idx[n] = (int64_t)(bi == 0x7fffffff ? 0 : bi);
        }
    }
}

// margin of the split-fp16 search, relative to (d_best + d_second + 2 ||x||^2): the measured worst error of a score is
// ~1e-6 of that scale (tests/test_vq_gpu.py::test_split_scores_within_margin), the 16-ulp row packing adds 1.9e-6, the
// exact chain's own rounding the rest (the fp32 search's 4e-5 covers a 48-step chain on both sides with the same headroom)
constexpr float WD_SPLIT_MARGIN = 4e-5f;

// scratch layout of the split search: [fragments][scaled norms][scale word (largest |c| bits), 16 bytes]
template <int K>
static void wd_split_pointers(int C, void* ws, uint4*& frag, float*& norms, uint32_t*& absmax)
{
    const int ntiles = (C + MF_CT - 1) / MF_CT;
    const size_t frag_bytes = (size_t)ntiles * HfShape<K>::TILE_BYTES;
    frag = (uint4*)ws;
    norms = (float*)((char*)ws + frag_bytes);
    absmax = (uint32_t*)(norms + (size_t)ntiles * MF_CT);      // 4 words: [0] the scale the fragments were made with; [1], [2]: see vq_apply_split_kernel
}

template <int K>
static void launch_wd_split_codebook(int C, const float* codebook, void* ws, uint4*& frag, float*& norms, uint32_t*& absmax, hipStream_t s)
{   // the caller has cleared the scale word (it is the last 16 bytes of the split region, directly in front of the list)
    const int ntiles = (C + MF_CT - 1) / MF_CT;
    wd_split_pointers<K>(C, ws, frag, norms, absmax);
    const size_t n = (size_t)C * K;
    wd_absmax_kernel<<<(unsigned)std::min<size_t>((n + 4095) / 4096, 64), 256, 0, s>>>(n, codebook, absmax);
    const unsigned gs = (unsigned)((ntiles * (MF_CT / 32) * HfShape<K>::KS * 64 + 255) / 256);
    wd_split_codebook_kernel<K><<<std::max(gs, (unsigned)((ntiles * MF_CT + 255) / 256)), 256, 0, s>>>(C, codebook, frag, norms, absmax);
}

// ws (optional device scratch, 16-byte aligned): [split codebook: wd_split_bytes(C, K)][int32 list of ambiguous points: 1 + cap].
// With room for the split codebook the search runs on the fp16 matrix cores, otherwise on the fp32 ones; whatever is
// left holds the list (without a list the flagged points are re-scanned one by one).
template <int K>
static void launch_wd_mfma(int64_t N, int C, const float* coefs, const int64_t* gather, const float* codebook, float* out_dist,
                           int64_t* out_idx, void* ws, size_t ws_bytes, hipStream_t s, int presplit = -1)
{   // presplit >= 0 (fused Lloyd step): ws already holds the fragments / norms / scale word of THIS codebook and a cleared list
    // counter (vq_apply_split_kernel wrote them); presplit = the parity that selects the word with the codebook's true abs-max
    const unsigned g1 = (unsigned)((N + 4 * MF_PTS - 1) / (4 * MF_PTS)), g2 = (unsigned)((N + 63) / 64);
    // fewer than one 4-wave workgroup per CU (256 CUs): let the 4 waves share 64 points and split the codebook instead
    // (measured, K = 4096 x 48: N = 32,768: 267 -> 158 us; N = 65,536: 271 vs 283 us, so the plain kernel from there on;
    // profiles/r02a_vq_slices.txt)
    static const int split_env = []() { const char* e = getenv("C3DGS_VQ_SPLIT"); return e ? atoi(e) : -1; }();   // A/B switch for tests
    static const bool force_f32 = getenv("C3DGS_VQ_F32_MFMA") != nullptr;                                          // A/B switch
    static const float margin_env = []() { const char* e = getenv("C3DGS_VQ_SPLIT_MARGIN"); return e ? (float)atof(e) : WD_SPLIT_MARGIN; }();
    const bool split = split_env >= 0 ? split_env != 0 : g1 < 256;
    const size_t sb = wd_split_bytes(C, K);
    const bool f16 = ws && ws_bytes >= sb && (((uintptr_t)ws) & 15) == 0 && !force_f32;
    char* rest = (char*)ws + (f16 ? sb : 0);
    const size_t rest_bytes = ws ? ws_bytes - (f16 ? sb : 0) : 0;
    int* flag_list = rest_bytes >= 2 * sizeof(int) && N < ((int64_t)1 << 31) ? (int*)rest : nullptr;
    const int flag_cap = flag_list ? (int)std::min<size_t>(rest_bytes / sizeof(int) - 1, (size_t)0x7fffffff) : 0;
    const bool listed = flag_list != nullptr;
    // one fill clears the scale word (last 16 bytes of the split region) and the list's counter behind it
    if (presplit >= 0 && f16) { }
    else if (f16) (void)hipMemsetAsync((char*)ws + sb - 16, 0, 16 + (listed ? std::min<size_t>(rest_bytes, 16) : 0), s);   // up to 32 bytes: one aligned fill (the first list entries are rewritten by the search)
    else if (listed) (void)hipMemsetAsync(flag_list, 0, sizeof(int), s);
    if (f16) {
        uint4* frag; float* norms; uint32_t* absmax;
        const uint32_t* absmax_true = nullptr;
        if (presplit >= 0) { wd_split_pointers<K>(C, ws, frag, norms, absmax); absmax_true = absmax + 1 + (presplit & 1); }
        else launch_wd_split_codebook<K>(C, codebook, ws, frag, norms, absmax, s);
        if (split) wd_f16_kernel<K, true, false><<<g2, 256, 0, s>>>(N, C, coefs, gather, codebook, frag, norms, absmax, out_dist, out_idx, flag_list, flag_cap, margin_env, nullptr, absmax_true);
        else wd_f16_kernel<K, false, false><<<g1, 256, 0, s>>>(N, C, coefs, gather, codebook, frag, norms, absmax, out_dist, out_idx, flag_list, flag_cap, margin_env, nullptr, absmax_true);
    } else if (split) wd_mfma_kernel<K, true><<<g2, 256, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx, flag_list, flag_cap);
    else wd_mfma_kernel<K, false><<<g1, 256, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx, flag_list, flag_cap);
    if (listed) {
        const unsigned gl = (unsigned)std::min<int64_t>(1024, ((int64_t)flag_cap + WD_FB - 1) / WD_FB);
        wd_fixup_list_kernel<K><<<gl, WD_FT, 0, s>>>(C, coefs, gather, codebook, out_dist, out_idx, flag_list, flag_cap);
    }
    // whatever is still flagged: only possible when the list could not hold every point
    if (!listed || (int64_t)flag_cap < N) wd_fixup_kernel<K><<<g2, 256, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx);
}

size_t wd_ws_bytes(int64_t N, int C, int K)
{
    const size_t list = ((size_t)(N > 0 ? N : 0) + 4) * sizeof(int);               // ~1-2 % of the points are ambiguous, but a list that
                                                                                    // can hold them all saves the closing sweep's launch
    return ((K == 48 || K == 12 || K == 6) ? wd_split_bytes(C, K) : 0) + list;
}

// can the fused Lloyd step (vq_apply_split_kernel + pre-split search) serve this shape with this scratch?
bool wd_presplit_supported(int C, int K, const float* coefs, const float* codebook, const void* ws, size_t ws_bytes)
{
    static const bool off = getenv("C3DGS_VQ_EXACT_VALU") != nullptr || getenv("C3DGS_VQ_F32_MFMA") != nullptr || getenv("C3DGS_VQ_NO_FUSED_STEP") != nullptr;
    if (off || !(K == 48 || K == 12 || K == 6) || C < 32 || !ws || (((uintptr_t)ws) & 15)) return false;
    if ((((uintptr_t)coefs | (uintptr_t)codebook) & (K == 48 ? 15 : 7)) != 0) return false;
    return ws_bytes >= wd_split_bytes(C, K) + 2 * sizeof(int);
}

int launch_weighted_distance(int64_t N, int C, int K, const float* coefs, const int64_t* gather, const float* codebook,
                             float* out_dist, int64_t* out_idx, hipStream_t s, void* ws, size_t ws_bytes, int presplit)
{
    if (N <= 0) return 0;
    if (presplit >= 0) {
        if (!wd_presplit_supported(C, K, coefs, codebook, ws, ws_bytes)) return 1;
        if (K == 48) launch_wd_mfma<48>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s, presplit);
        else if (K == 12) launch_wd_mfma<12>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s, presplit);
        else launch_wd_mfma<6>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s, presplit);
        return 0;
    }
    const int64_t per_block = (int64_t)WD_BLOCK * WD_PPT;
    const unsigned grid = (unsigned)((N + per_block - 1) / per_block);
    const bool al16 = (((uintptr_t)coefs | (uintptr_t)codebook) & 15) == 0;
    const bool al8 = (((uintptr_t)coefs | (uintptr_t)codebook) & 7) == 0;
    static const bool force_exact = getenv("C3DGS_VQ_EXACT_VALU") != nullptr;   // A/B switch for tests and profiling
    if (K == 48 && al16 && C >= 32 && !force_exact) launch_wd_mfma<48>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s);
    else if (K == 12 && al8 && C >= 32 && !force_exact) launch_wd_mfma<12>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s);
    else if (K == 6 && al8 && C >= 32 && !force_exact) launch_wd_mfma<6>(N, C, coefs, gather, codebook, out_dist, out_idx, ws, ws_bytes, s);
    else if (K == 48 && al16) weighted_distance_kernel<48, 128><<<grid, WD_BLOCK, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx);
    else if (K == 12 && al16) weighted_distance_kernel<12, 512><<<grid, WD_BLOCK, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx);
    else if (K == 6 && al8) weighted_distance_kernel<6, 1024><<<grid, WD_BLOCK, 0, s>>>(N, C, coefs, gather, codebook, out_dist, out_idx);
    else {
        if (K > 6144) return 1;
        const unsigned g2 = (unsigned)((N + WD_BLOCK - 1) / WD_BLOCK);
        weighted_distance_generic_kernel<6144><<<g2, WD_BLOCK, 0, s>>>(N, C, K, coefs, gather, codebook, out_dist, out_idx);
    }
    return 0;
}

// diagnostics: the scores s[n][c] = ||c||^2 - 2 x_n.c exactly as the split-fp16 search forms them (K = 48, N <= 256 points)
int launch_wd_debug_scores(int64_t N, int C, int K, const float* coefs, const float* codebook, float* scores, void* ws, size_t ws_bytes,
                           float* out_dist, int64_t* out_idx, hipStream_t s)
{
    if (K != 48 || N <= 0 || N > 256 || C < 32 || !ws || ws_bytes < wd_split_bytes(C, K) || (((uintptr_t)ws) & 15)) return 1;
    uint4* frag; float* norms; uint32_t* absmax;
    (void)hipMemsetAsync((char*)ws + wd_split_bytes(C, K) - 16, 0, 16, s);
    launch_wd_split_codebook<48>(C, codebook, ws, frag, norms, absmax, s);
    wd_f16_kernel<48, false, true><<<1, 256, 0, s>>>(N, C, coefs, nullptr, codebook, frag, norms, absmax, out_dist, out_idx, nullptr, 0, WD_SPLIT_MARGIN, scores);
    return 0;
}

// ---- VectorQuantize.update, part 1: weighted scatter-sums (compression/vq.py:31,33) into S[K, D+1].
// A wave takes 64 points at a time (lane = point: index, row, weight), finds the lanes that share a codeword with BITS
// ballots (the match-any idiom of the radix sort's ranking), and then walks the GROUPS: the lanes turn into the D+1
// channels of one codeword row, sum the group's members (coalesced row reads, weight broadcast by readlane) and issue ONE
// contiguous (D+1)-float atomic run per group -- the access shape the chip's memory-side float atomics run fastest on.
// Contention-aware by construction: right after uniform_init a handful of codewords win every point (a 2^18-point batch
// used to pile 12.8 M atomics onto a few hundred addresses: 1.2 ms on the first Lloyd step against 0.08 ms later);
// merged per wave that is 64x fewer atomics in the worst case and the same number in the contention-free one.
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// one double atomic per workgroup (256 threads): wave shuffles, then LDS
__device__ __forceinline__ void block_add_double(double v, double* out)
{
    __shared__ double s_d[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_d[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (s_d[0] + s_d[1]) + (s_d[2] + s_d[3]);
        if (t != 0.0) atomicAdd(out, t);
    }
}

__global__ void __launch_bounds__(256)
vq_accumulate_kernel(int64_t B, int D, int idx_bits, const float* __restrict__ x, const float* __restrict__ w,
                     const int64_t* __restrict__ gather, const int64_t* __restrict__ idx, float* __restrict__ S,
                     const float* __restrict__ dist, double* __restrict__ dist_sum, uint32_t* __restrict__ clear_word, int pts)
{
    // pts = points per wave and round (64, or 16 for small batches: a rank's 2^15-point slice is only 512 waves of 64 points --
    // half the chip's SIMDs, each walking its chunk's 49 sweeps of gathers one memory round trip after the other; 16 points
    // per wave give four times the waves a quarter of the sweeps each)
    // housekeeping for the fused Lloyd step (vq_apply_split_kernel): the absmax word the NEXT apply accumulates into
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0u;
    double dacc = 0.0;                                           // sum of this lane's min distances (vq.py:71), folded in: no launch of its own
    const int D1 = D + 1;
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // second merge level, across the workgroup's waves: group sums of SHARED codewords go to a small direct-mapped table in
    // LDS first (slot = codeword & 31, claimed by compare-and-swap; a codeword that finds its slot taken by another one adds
    // to global memory directly) and reach global memory once per workgroup
    constexpr int SLOTS = 32;
    __shared__ int s_tag[SLOTS];
    __shared__ float s_sum[SLOTS][64];
    const bool use_slots = D1 <= 64;
    if (use_slots) {
        for (int q = threadIdx.x; q < SLOTS * 64; q += 256) (&s_sum[0][0])[q] = 0.f;
        if (threadIdx.x < SLOTS) s_tag[threadIdx.x] = -1;
    }
    __syncthreads();
    for (int64_t base = wave_id * pts; base < B; base += n_waves * pts) {
        const int64_t n = base + lane;
        const bool valid = lane < pts && n < B;
        const int64_t row = valid ? (gather ? gather[n] : n) : 0;
        const float wn = valid ? w[row] : 0.f;
        const uint32_t id = valid ? (uint32_t)idx[n] : 0u;
        if (dist && valid) dacc += (double)dist[n];
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < idx_bits; b++) {
            const bool bit = (id >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        unsigned long long leaders = __ballot(valid && (peers & lt) == 0ull);
        if (leaders == __ballot(valid)) {
            // no two points of the chunk share a codeword (the steady state): nothing to merge, so the chunk's cnt * (D+1)
            // elements are added densely, 64 consecutive elements per instruction (every lane busy; a point's D+1 atomics
            // stay contiguous)
            const int cnt = (int)__popcll(leaders), total = cnt * D1;   // valid lanes are the first cnt lanes of the wave
            const uint32_t row_lo = (uint32_t)row, row_hi = (uint32_t)((uint64_t)row >> 32);
            // The trip count is WAVE-UNIFORM and the shuffles run with every lane active: ds_bpermute returns 0 from a source
            // lane that is masked off, so a lane-dependent loop bound (lanes with e >= total leaving early in the last sweep)
            // lost the contributions of points whose owner lane had left -- ragged tails with (cnt * D1) % 64 in 1..cnt-1.
            // Seven sweeps of 64 elements per round (D + 1 = 49 = 7 x 7 for the colour codebook): the element loads of a round
            // are all issued before its first atomic. With one load per iteration the compiler cannot move a load across the
            // atomic in front of it, and a wave paid one dependent memory round trip per sweep -- invisible on a 2^18-point
            // batch (4096 waves hide each other), 41 us of a 2^15-point slice's 128 us of kernels.
            constexpr int U = 7;
            for (int e0 = 0; e0 < total; e0 += 64 * U) {             // the point's row / weight / codeword come from the lane
                float v[U];                                          // that already holds them (no dependent global loads)
                size_t dst[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int e = e0 + 64 * u + lane;
                    const int j = min(e / D1, cnt - 1), c = e - j * D1;
                    const int64_t rj = (int64_t)(((uint64_t)(uint32_t)__shfl((int)row_hi, j) << 32) | (uint32_t)__shfl((int)row_lo, j));
                    const float wj = __shfl(wn, j);
                    const uint32_t idj = (uint32_t)__shfl((int)id, j);
                    dst[u] = (size_t)idj * D1 + c;
                    v[u] = (e < total && c < D) ? x[rj * D + c] * wj : wj;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (e0 + 64 * u + lane < total) atomicAdd(S + dst[u], v[u]);
            }
            continue;
        }
        while (leaders) {
            const int l = __builtin_ctzll(leaders);
            leaders &= leaders - 1;
            const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)peers, l);
            const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(peers >> 32), l);
            unsigned long long members = ((unsigned long long)phi << 32) | plo;
            const int gid = __builtin_amdgcn_readlane((int)id, l);
            float* dst = S + (size_t)(uint32_t)gid * D1;
            if (use_slots) {
                int owner = 0;
                if (lane == 0) owner = atomicCAS(&s_tag[gid & (SLOTS - 1)], -1, gid);
                owner = __builtin_amdgcn_readfirstlane(owner);
                if (owner == -1 || owner == gid) dst = nullptr;      // this workgroup's slot of the codeword
            }
            for (int c0 = 0; c0 < D1; c0 += 64) {                    // D + 1 <= 64 for every codebook of the pipeline: one pass
                const int c = c0 + lane;
                float acc = 0.f;
                unsigned long long m = members;
                while (m) {                                          // four member rows in flight, added in member order
                    float xv[4], wv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const bool have = m != 0;
                        const int j = have ? __builtin_ctzll(m) : 0;
                        m &= m - 1;                                  // stays 0 once exhausted
                        const int64_t rj = readlane_i64(row, j);
                        wv[u] = have ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(wn), j)) : 0.f;
                        xv[u] = (have && c < D) ? x[rj * D + c] : 1.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) acc += xv[u] * wv[u];
                }
                if (c < D1) {
                    if (dst) atomicAdd(dst + c, acc);
                    else __hip_atomic_fetch_add(&s_sum[gid & (SLOTS - 1)][c], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    if (use_slots) {
        __syncthreads();
        for (int sl = threadIdx.x >> 6; sl < SLOTS; sl += 4) {
            const int tag = s_tag[sl];
            if (tag >= 0 && lane < D1) atomicAdd(S + (size_t)(uint32_t)tag * D1 + lane, s_sum[sl][lane]);
        }
    }
    if (dist && dist_sum) block_add_double(dacc, dist_sum);
}

// Small tables (K*(D+1) <= 16 K floats, e.g. the covariance codebook: 2048 x 7 = 56 KB) are accumulated in a
// per-workgroup LDS copy first: a million points hammering 2048 rows with global float atomics serialise at the
// memory side (MI355X_MICROARCH.md: "every workgroup into ONE row: 14x slower"), LDS atomics do not leave the CU.
constexpr int VQ_LDS_FLOATS = 16384;

__global__ void __launch_bounds__(256)
vq_accumulate_lds_kernel(int64_t B, int K, int D, const float* __restrict__ x, const float* __restrict__ w,
                         const int64_t* __restrict__ gather, const int64_t* __restrict__ idx, float* __restrict__ S,
                         const float* __restrict__ dist, double* __restrict__ dist_sum, uint32_t* __restrict__ clear_word)
{
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0u;
    __shared__ float s_S[VQ_LDS_FLOATS];
    const int D1 = D + 1, table = K * D1;
    for (int q = threadIdx.x; q < table; q += 256) s_S[q] = 0.f;
    __syncthreads();
    const int64_t total = B * D1;
    // contiguous slab of elements per workgroup (keeps a point's D+1 adds together)
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t e0 = (int64_t)blockIdx.x * per, e1 = e0 + per < total ? e0 + per : total;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int64_t n = e / D1;
        const int c = (int)(e - n * D1);
        const int64_t row = gather ? gather[n] : n;
        const float wn = w[row];
        const float v = (c < D) ? x[row * D + c] * wn : wn;
        atomicAdd(&s_S[(int)idx[n] * D1 + c], v);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < table; q += 256) {
        const float v = s_S[q];
        if (v != 0.f) atomicAdd(S + q, v);
    }
    if (dist && dist_sum) {                                        // the batch's distance sum, folded in (a point slab per workgroup)
        const int64_t perp = (B + gridDim.x - 1) / gridDim.x, p0 = (int64_t)blockIdx.x * perp, p1 = p0 + perp < B ? p0 + perp : B;
        double dacc = 0.0;
        for (int64_t n = p0 + threadIdx.x; n < p1; n += 256) dacc += (double)dist[n];
        block_add_double(dacc, dist_sum);
    }
}

void launch_vq_accumulate(int64_t B, int K, int D, const float* x, const float* w, const int64_t* gather,
                          const int64_t* idx, const float* dist, float* S, double* dist_sum, hipStream_t s, uint32_t* clear_word)
{
    if (B <= 0) return;
    const int64_t total = B * (D + 1);
    if ((int64_t)K * (D + 1) <= VQ_LDS_FLOATS && total >= (int64_t)1 << 16) {
        const unsigned grid = (unsigned)std::min<int64_t>((total + 4095) / 4096, 512);
        vq_accumulate_lds_kernel<<<grid, 256, 0, s>>>(B, K, D, x, w, gather, idx, S, dist, dist_sum, clear_word);
    } else {
        int idx_bits = 1;
        while (idx_bits < 32 && ((int64_t)1 << idx_bits) < (int64_t)K) idx_bits++;
        const int pts = B <= ((int64_t)1 << 16) ? 16 : 64;                                  // points per wave (see the kernel)
        const unsigned grid = (unsigned)std::min<int64_t>((B + 4 * pts - 1) / (4 * pts), 256 * 16);
        vq_accumulate_kernel<<<grid, 256, 0, s>>>(B, D, idx_bits, x, w, gather, idx, S, dist, dist_sum, clear_word, pts);
    }
}

// Single-rounded fp32 operations for the update kernels: plain operators, expanded INSIDE a function body that starts with
// `#pragma clang fp contract(off)`. (The __fmul_rn / __fadd_rn inline functions of this toolchain are plain operators that
// carry the translation unit's contraction flag with them: the compiler does fuse them into FMAs, differently from kernel
// to kernel -- found when the fused update kernel disagreed with vq_apply_kernel by one ulp.)
#define RN_MUL(a, b) ((a) * (b))
#define RN_ADD(a, b) ((a) + (b))
#define RN_DIV(a, b) ((a) / (b))
// ---- part 2: EMA + optional trace normalisation; one thread per codeword row. Every operation is
// a single rounded fp32 op (RN_* under contract(off)) in the order torch evaluates
// moving_avg.mul_(decay).add_(new, alpha) so all ranks of a sharded run stay bit-identical.
__global__ void __launch_bounds__(256)
vq_apply_kernel(int K, int D, const float* __restrict__ S, float* __restrict__ codebook, float* __restrict__ entry_importance,
                float decay, float alpha, float eps, int scale_normalize)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const float* srow = S + (size_t)k * (D + 1);
    float* cb = codebook + (size_t)k * D;
    const float aw = srow[D];
    entry_importance[k] = RN_ADD(RN_MUL(entry_importance[k], decay), RN_MUL(alpha, aw));
    const float den = RN_ADD(aw, eps);
    for (int d = 0; d < D; d++) {
        const float nw = RN_DIV(srow[d], den);
        cb[d] = RN_ADD(RN_MUL(cb[d], decay), RN_MUL(alpha, nw));
    }
    if (scale_normalize && D >= 6) {
        const float tr = RN_ADD(RN_ADD(cb[0], cb[3]), cb[5]);
        for (int d = 0; d < D; d++) cb[d] = RN_DIV(cb[d], tr);
    }
}

// the same update with one thread per codebook ELEMENT (no trace normalisation): a row per thread runs D correctly rounded
// divisions in sequence on 16 workgroups (16 us for K = 4096, D = 48: latency, not work)
__global__ void __launch_bounds__(256)
vq_apply_elem_kernel(int K, int D, const float* __restrict__ S, float* __restrict__ codebook, float* __restrict__ entry_importance,
                     float decay, float alpha, float eps)
{
#pragma clang fp contract(off)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= K * D) return;
    const int k = e / D, d = e - k * D;
    const float aw = S[(size_t)k * (D + 1) + D];
    if (d == 0) entry_importance[k] = RN_ADD(RN_MUL(entry_importance[k], decay), RN_MUL(alpha, aw));
    const float nw = RN_DIV(S[(size_t)k * (D + 1) + d], RN_ADD(aw, eps));
    codebook[e] = RN_ADD(RN_MUL(codebook[e], decay), RN_MUL(alpha, nw));
}

void launch_vq_apply(int K, int D, const float* S, float* codebook, float* entry_importance, float decay, float alpha,
                     float eps, int scale_normalize, hipStream_t s)
{
    if (K <= 0) return;
    if (!(scale_normalize && D >= 6) && (int64_t)K * D < ((int64_t)1 << 31))
        vq_apply_elem_kernel<<<(unsigned)(((int64_t)K * D + 255) / 256), 256, 0, s>>>(K, D, S, codebook, entry_importance, decay, alpha, eps);
    else
        vq_apply_kernel<<<(K + 255) / 256, 256, 0, s>>>(K, D, S, codebook, entry_importance, decay, alpha, eps, scale_normalize);
}


// ---- the Lloyd step's second half for the loop of vq_features, ONE launch: EMA update (the same single-rounded operations as
// vq_apply_kernel, so ranks and the unfused path stay bit-identical) + everything the NEXT step's search needs, which used to
// be four more launches per step (clear of S, clear of the list counter + scale word, wd_absmax_kernel,
// wd_split_codebook_kernel): the new codebook's fp16 split fragments and scaled ||c||^2, S cleared behind its last reader,
// the list counter cleared. A workgroup owns one 32-codeword sub-tile of the fragment layout.
// Scale: the fragments need ONE power of two for the whole codebook, i.e. its largest magnitude -- which this launch is only
// producing. It scales with the exponent of the codebook it READ (words[1 + parity], complete since the previous launch),
// accumulates the new codebook's abs-max into words[1 + (1 - parity)] (cleared by the accumulate kernel in between) and
// records the scale it used in words[0]; the search compares the two and falls back to the exact scan if they are far apart.
template <int MF_K>
__global__ void __launch_bounds__(256)
vq_apply_split_kernel(int K, float* __restrict__ S, float* __restrict__ codebook, float* __restrict__ entry_importance, float decay,
                      float alpha, float eps, int scale_normalize, uint4* __restrict__ frag, float* __restrict__ norms,
                      uint32_t* __restrict__ words, int parity, int* __restrict__ list_counter)
{
#pragma clang fp contract(off)   // the update must be the SAME single-rounded operations as vq_apply_kernel (explicit fmaf stays fused)
    constexpr int D = MF_K, D1 = MF_K + 1, KS = HfShape<MF_K>::KS;
    __shared__ float s_cb[32][D + 1];
    __shared__ float s_tr[32];
    __shared__ uint32_t s_m[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ts = blockIdx.x, row0 = ts * 32;
    const uint32_t a_cur = words[1 + parity];
    const float sc = __builtin_ldexpf(1.0f, wd_scale_exp(a_cur));
    if (tid < 32) {
        const int k = row0 + tid;
        if (k < K) {
            const float aw = S[(size_t)k * D1 + D];
            entry_importance[k] = RN_ADD(RN_MUL(entry_importance[k], decay), RN_MUL(alpha, aw));
        }
    }
    for (int e = tid; e < 32 * D; e += 256) {
        const int r = e / D, d = e - r * D, k = row0 + r;
        float v = 0.f;
        if (k < K) {
            const float aw = S[(size_t)k * D1 + D];
            const float nw = RN_DIV(S[(size_t)k * D1 + d], RN_ADD(aw, eps));
            v = RN_ADD(RN_MUL(codebook[(size_t)k * D + d], decay), RN_MUL(alpha, nw));
        }
        s_cb[r][d] = v;
    }
    __syncthreads();
    if (scale_normalize && D >= 6) {                             // vq.py:73-77: codebook /= (cb[:,0] + cb[:,3] + cb[:,5])[:, None]
        if (tid < 32) s_tr[tid] = RN_ADD(RN_ADD(s_cb[tid][0], s_cb[tid][3]), s_cb[tid][5]);
        __syncthreads();
        for (int e = tid; e < 32 * D; e += 256) {
            const int r = e / D, d = e - r * D;
            if (row0 + r < K) s_cb[r][d] = RN_DIV(s_cb[r][d], s_tr[r]);
        }
        __syncthreads();
    }
    uint32_t m = 0;
    for (int e = tid; e < 32 * D; e += 256) {
        const int r = e / D, d = e - r * D, k = row0 + r;
        if (k < K) {
            const float v = s_cb[r][d];
            codebook[(size_t)k * D + d] = v;
            const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
            m = max(m, b <= 0x7f800000u ? b : 0u);               // NaNs do not set the scale (as wd_absmax_kernel)
        }
    }
    for (int e = tid; e < 32 * D1; e += 256) {                   // S is consumed: cleared for the next step's accumulation
        const int k = row0 + e / D1;
        if (k < K) S[(size_t)row0 * D1 + e] = 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if (lane == 0) s_m[wave] = m;
    // scaled ||c||^2 (k-ordered FMA chain, as wd_split_codebook_kernel); rows past K never win
    if (tid < 32) {
        float nr = 3.0e38f;
        if (row0 + tid < K) {
            nr = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) nr = fmaf(s_cb[tid][k], s_cb[tid][k], nr);
            nr = (nr * sc) * sc;
        }
        norms[row0 + tid] = nr;
    }
    // fragments of this sub-tile: [k-step][piece][lane][8 x fp16]
    for (int t = tid; t < KS * 64; t += 256) {
        const int fl = t & 63, q = t >> 6;
        const int r = fl & 31, k0 = 16 * q + 8 * (fl >> 5);
        union { f16x8 v; uint4 u; } p[HF_PIECES];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = (row0 + r < K && k0 + j < D) ? s_cb[r][k0 + j] * sc : 0.f;
            _Float16 h, l;
            f16_split2(v, h, l);
            p[0].v[j] = h; p[1].v[j] = l;
        }
#pragma unroll
        for (int e = 0; e < HF_PIECES; e++) frag[((size_t)(ts * KS + q) * HF_PIECES + e) * 64 + fl] = p[e].u;
    }
    __syncthreads();
    if (tid == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(&words[1 + (1 - parity)], m);
        if (blockIdx.x == 0) { words[0] = a_cur; if (list_counter) *list_counter = 0; }
    }
}

// first half of the fused protocol: make the scratch of a codebook that was split the classic way (absmax kernel: words[0])
// ready for vq_apply_split_kernel with parity 0: words[1] = words[0] (true abs-max of the current codebook), words[2] = 0
__global__ void vq_seed_words_kernel(uint32_t* words) { words[1] = words[0]; words[2] = 0u; }

int launch_vq_apply_split(int K, int D, float* S, float* codebook, float* entry_importance, float decay, float alpha, float eps,
                          int scale_normalize, void* ws, size_t ws_bytes, int parity, hipStream_t s)
{
    if (!wd_presplit_supported(K, D, codebook, codebook, ws, ws_bytes)) return 1;
    const int ntiles = (K + MF_CT - 1) / MF_CT;
    uint4* frag; float* norms; uint32_t* words;
    int* list = (int*)((char*)ws + wd_split_bytes(K, D));
#define C3DGS_VQ_AS(KK) { wd_split_pointers<KK>(K, ws, frag, norms, words);                                              \
        vq_apply_split_kernel<KK><<<ntiles * (MF_CT / 32), 256, 0, s>>>(K, S, codebook, entry_importance, decay, alpha, eps,    \
                                                                        scale_normalize, frag, norms, words, parity & 1, list); }
    if (D == 48) C3DGS_VQ_AS(48) else if (D == 12) C3DGS_VQ_AS(12) else C3DGS_VQ_AS(6)
#undef C3DGS_VQ_AS
    return 0;
}

void launch_vq_seed_words(int K, int D, void* ws, hipStream_t s)
{
    uint4* frag; float* norms; uint32_t* words;
    if (D == 48) wd_split_pointers<48>(K, ws, frag, norms, words);
    else if (D == 12) wd_split_pointers<12>(K, ws, frag, norms, words);
    else wd_split_pointers<6>(K, ws, frag, norms, words);
    vq_seed_words_kernel<<<1, 1, 0, s>>>(words);
}

uint32_t* vq_next_absmax_word(int K, int D, void* ws, int parity)
{
    uint4* frag; float* norms; uint32_t* words;
    if (D == 48) wd_split_pointers<48>(K, ws, frag, norms, words);
    else if (D == 12) wd_split_pointers<12>(K, ws, frag, norms, words);
    else wd_split_pointers<6>(K, ws, frag, norms, words);
    return words + 1 + (1 - (parity & 1));
}

} // namespace c3dgs
