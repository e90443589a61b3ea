/*
 * c3dgs_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT THE PRODUCT).
 *
 * Plain-C restatement of the reference's rasterizer + VQ hot path, used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under c3dgs_amd/ may import, link or call this.
 *
 * Parity pinning: the reference ships NO tests, golden vectors or fixtures for
 * this path and its CUDA kernels cannot be compiled here
 * (no nvcc), so the raster part is "parity unpinned" by the reference's own
 * tests; it is pinned instead by (a) the importable reference Python
 * (utils/sh_utils.eval_sh, utils/general_utils.build_covariance_from_scaling_rotation,
 * compression/vq.py with two shims) through tests/golden/ fixtures, and
 * (b) a float64 torch-autograd dense re-derivation of the blend (tests/dense_ref.py).
 *
 * Float discipline: compiled with -ffp-contract=off; every operation is a single
 * IEEE fp32 op in the order written, except where an explicit fmaf() is used
 * (documented at each site).  The HIP kernels for K1/K2/K5 follow the same order,
 * so radii / tiles_touched / tile keys / sorted point_list compare bit-exactly.
 */
#ifndef C3DGS_ORACLE_H
#define C3DGS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int P;            /* number of Gaussians passed in                           */
    int D;            /* active SH degree                                        */
    int M;            /* SH coefficients per Gaussian (max_degree+1)^2           */
    int W, H;
    const float* bg;              /* [3]                                          */
    const float* means3D;         /* [P,3]                                        */
    const float* shs;             /* [P|SHS, M, 3] or NULL                        */
    const float* colors_precomp;  /* [P,3] or NULL                                */
    const float* opacities;       /* [P]                                          */
    const float* scales;          /* [P|GS,3] or NULL                             */
    const float* scale_factors;   /* [P] (indexed only) or NULL                   */
    const float* rotations;       /* [P|GS,4] or NULL                             */
    const float* cov3D_precomp;   /* [P,6] or NULL                                */
    const int64_t* sh_indices;    /* [P] or NULL  (non-NULL => indexed variant)   */
    const int64_t* g_indices;     /* [P] or NULL                                  */
    const float* viewmatrix;      /* [16] = W2C transposed (column-major W2C)     */
    const float* projmatrix;      /* [16]                                         */
    const float* campos;          /* [3]                                          */
    float tan_fovx, tan_fovy;
    float scale_modifier;
    int prefiltered;
    int clamp_color;
    int SHS, GS;      /* codebook sizes for the indexed variant (grad shapes)    */
} orc_params;

/* per-Gaussian state written by preprocess (reference: GeometryState) */
typedef struct {
    float*    depths;        /* [P]   */
    uint8_t*  clamped;       /* [P,3] */
    int32_t*  radii;         /* [P]   */
    float*    means2D;       /* [P,2] */
    float*    cov3D;         /* [P,6] */
    float*    conic_opacity; /* [P,4] */
    float*    rgb;           /* [P,3] */
    uint32_t* tiles_touched; /* [P]   */
    uint32_t* point_offsets; /* [P]   inclusive scan of tiles_touched */
} orc_geom;

uint32_t orc_get_higher_msb(uint32_t n);

void orc_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present);

/* K2/K2i + K3. Returns num_rendered. */
int orc_forward_stage1(const orc_params* p, orc_geom* g);

/* K5..K9. Buffers sized by the caller from num_rendered R and T tiles, N pixels. */
void orc_forward_stage2(const orc_params* p, const orc_geom* g, int R,
                        uint64_t* keys_unsorted, uint32_t* values_unsorted,
                        uint64_t* keys_sorted, uint32_t* point_list,
                        uint32_t* ranges /*[T,2]*/,
                        float* out_color /*[3,H,W]*/, float* final_T /*[N]*/,
                        uint32_t* n_contrib /*[N]*/);

/* K10..K12(i). All outputs must be zero-initialised by the caller. */
void orc_backward(const orc_params* p, const orc_geom* g, int R,
                  const uint32_t* point_list, const uint32_t* ranges,
                  const float* final_T, const uint32_t* n_contrib,
                  const float* dL_dpix /*[3,H,W]*/,
                  float* dL_dmean2D /*[P,3]*/, float* dL_dconic /*[P,4]*/,
                  float* dL_dopacity /*[P]*/, float* dL_dcolors /*[P,3]*/,
                  float* dL_dmean3D /*[P,3]*/, float* dL_dcov3D /*[P,6]*/,
                  float* dL_dsh /*[P|SHS,M,3]*/, float* dL_dscale /*[P|GS,3]*/,
                  float* dL_dscale_factor /*[P] or NULL*/, float* dL_drot /*[P|GS,4]*/);

/* K14: exact nearest codeword, fp32 fmaf chain, strict '<' (lowest index wins). */
void orc_weighted_distance(int64_t N, int C, int K, const float* coefs,
                           const float* codebook, float* dist, int64_t* idx);

/* V3: one weighted EMA Lloyd step (compression/vq.py:28-35), in place.
 * Returns the mean of min_dists. Sums are accumulated in float64. */
double orc_vq_update(int64_t B, int K, int D, const float* x, const float* w,
                     float* codebook, float* entry_importance, double decay, double eps,
                     float* min_dists /*[B] or NULL*/, int64_t* idx_out /*[B] or NULL*/);

/* vq.py:73-77 : codebook /= (cb[:,0]+cb[:,3]+cb[:,5])[:,None]  (D == 6) */
void orc_vq_trace_normalize(int K, int D, float* codebook);

int orc_num_threads(void);

/* N4: Morton codes of scene/gaussian_model.py:997-1003,1417-1432 */
void orc_morton_codes(int P, const float* xyz, int64_t* codes, int32_t axis_order[3]);

/* N3: L1 + SSIM loss of finetune.py:48 / utils/loss_utils.py:17-63 and its gradient w.r.t. img */
void orc_l1_ssim(int C, int H, int W, const float* img, const float* gt, double lambda, double* out /*[3]*/,
                 float* dL_dimg /*[C,H,W] or NULL*/);

/* test hooks: expose the per-Gaussian SH->RGB and cov3D helpers for golden-vector pinning */
void orc_test_color_from_sh(int n, int deg, int M, const float* pos, const float* campos, const float* sh,
                            int clamp_color, uint8_t* clamped, float* rgb);
void orc_test_cov3d(int n, const float* scales, float mod, const float* rots, float* cov);
void orc_test_sh_backward(int n, int deg, int M, const float* pos, const float* campos, const float* sh,
                          const uint8_t* clamped, const float* dL_dcolor, float* dL_dsh, float* dL_dmean);
void orc_test_cov3d_backward(int n, const float* scales, float mod, const float* rots, const float* dL_dcov3D,
                             float* d_s, float* dq);

#ifdef __cplusplus
}
#endif
#endif
