"""HIP-vs-oracle comparison of one view at BASELINE.json's FULL sizes (1M / 3M / 6M Gaussians at 1920x1080).

Shared by tests/test_fullsize_gpu.py (asserts the bars) and bench.py (`parity` object of the JSON line, reusing the
oracle run its `cpu_baseline` leg pays for anyway).  Test infrastructure: nothing under c3dgs_amd/ imports this.

What is compared (BASELINE.json north_star; reference path rasterizer_impl.cu:440-697):
  radii, tiles_touched, sorted 64-bit keys, sorted point_list, tile ranges      -> equal (counts of mismatches reported)
  image                                                                         -> PSNR(hip, oracle), |dPSNR| vs a fixed random target
  every gradient tensor                                                         -> ||g - g_ref||_inf / ||g_ref||_inf (oracle sums in float64)
plus the depth of the deepest tile list, because the gradient bar loosens with it (see test_fuzz_gpu.py).
"""
import time

import numpy as np
import torch

from tests import gpu_util


def config_inputs(name, P=None, W=1920, H=1080, focal=1200.0):
    """The BASELINE.json configurations as (inputs dict of CPU tensors, intrinsic, extrinsic_vector, indexed).
      "config2_1M_fwd"      configs[1]: 1M Gaussians, SH deg 3, non-indexed, forward only (render.py path)
      "config3_3M_indexed"  configs[2]: 3M, SH deg 3, indexed, clamp_color=True (finetune.py QAT loop) -- the headline
      "config4_6M_sens"     the sensitivity pass of configs[3]/[4]: 6M, non-indexed, cov3D_precomp, clamp_color=False
                            (compress.py:81-119)"""
    from tests import synth
    P = {"config2_1M_fwd": 1_000_000, "config3_3M_indexed": 3_000_000, "config4_6M_sens": 6_000_000}[name] if P is None else P
    intr, ev = synth.camera(W, H, focal)
    sc = synth.scene(P, W, H, focal, seed=1234, sh_degree=3)
    inp = dict(bg=torch.zeros(3), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], colors_precomp=None,
               scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None, scale_factors=None, sh_indices=None,
               g_indices=None, degree=3, scale_modifier=1.0, prefiltered=False, clamp_color=True)
    indexed = False
    if name == "config3_3M_indexed":
        ix = synth.index_scene(sc)
        inp.update(shs=ix["shs"], scales=ix["scales"], rotations=ix["rotations"], scale_factors=ix["scale_factors"],
                   sh_indices=ix["sh_indices"], g_indices=ix["g_indices"])
        indexed = True
    elif name == "config4_6M_sens":
        # cov3D_precomp = unit-scale covariance x scaling_factor^2 (compress.py:82-85,101): Sigma = R S^2 R^T, upper triangle
        q, s = sc["rotations"], sc["scales"]
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        Lm = Rm * s[:, None, :]
        Sg = Lm @ Lm.transpose(1, 2)
        inp["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]],
                                           1).float().contiguous()
        inp["scales"] = inp["rotations"] = None
        inp["clamp_color"] = False
    return inp, intr, ev, indexed


def oracle_view(inp, cam, dL=None):
    """One oracle view. -> (state, grads or None, seconds forward, seconds backward)."""
    from tests import cases
    from oracle import oracle as orc
    t0 = time.perf_counter()
    st = cases.oracle_forward(inp, cam)
    t1 = time.perf_counter()
    ref = orc.rasterize_backward(st, dL) if dL is not None else None
    return st, ref, t1 - t0, time.perf_counter() - t1


def flipped_pixels(u, st):
    """Pixels where an ulp of exp() flips a threshold decision (alpha >= 1/255, T < 1e-4) between the two implementations:
    n_contrib differs, or the colour moves by one borderline contribution. Any two exp() implementations (CUDA's expf,
    glibc's, v_exp_f32) disagree on a few of the ~10^9 decisions of a 1080p view; everything else agrees to ~1e-7.
    -> bool [H, W]"""
    a, b = u["out_color"], st.out_color
    return (u["n_contrib"] != st.n_contrib).reshape(st.H, st.W) | (np.abs(a - b) > 2e-5 + 1e-4 * np.abs(b)).any(0)


# ---- float64 proof that a flipped pixel is a true fp32 borderline ---------------------------------------------------
# Both implementations take, per (pixel, Gaussian) pair, the reference's three decisions (forward.cu:344-360, the same in
# backward.cu:486-501):   power > 0 -> skip;   alpha = min(0.99, o * exp(power)) < 1/255 -> skip;   T * (1 - alpha) < 1e-4 -> stop.
# They evaluate `power` from the SAME fp32 inputs (mean2D, conic, opacity and dx = mean.x - px, dy = mean.y - py are
# bit-identical on both sides, asserted by compare()) but with different roundings: the oracle one rounding per operation in
# the reference's order, the kernel three products + three FMAs on a conic pre-scaled by log2(e) feeding v_exp_f32.
# With u = 2^-24, Q = 0.5 (a dx^2 + c dy^2), X = |b dx dy| (|power| <= Q + X):
#   * either evaluation makes at most 3 roundings per product term and 1 per addition:
#         |power_fp32 - power_exact| <= u (3 Q + 3 X + |power|)         -> E_P = u (4 (Q + X) + |power|)   (one u of slack per term)
#   * exp (<= 1 ulp for expf and for v_exp_f32 of an exactly scaled argument... the scaling product is one more rounding,
#     counted in E_P's slack), the product with the opacity (1 rounding) and min():   rel. error of alpha <= E_A = E_P + 3 u
#   * T is a running product of (1 - alpha_k): every blended entry adds alpha_k E_A,k / (1 - alpha_k) + 2 u to its relative error.
# A decision whose float64 value lies farther from its threshold than these bounds is taken identically by ANY correct fp32
# evaluation. prove_flips() walks a flipped pixel's list in float64 from the bit-identical inputs, branches ONLY at decisions
# inside their band and requires the kernel's (n_contrib, final_T, colour) and the oracle's to each equal one leaf of that
# walk. A pixel whose kernel result matches no leaf is a flip OUTSIDE the band -- a real difference, not rounding.
_U = 2.0 ** -24


def _pixel_leaves(st, feat, x, y, max_leaves=256):
    """All outcomes of pixel (x, y) reachable by flipping in-band decisions. -> list of (n_contrib, T, colour[3], decisions)
    where decisions = [(kind, list position, margin / band)]; or None if there are more than max_leaves."""
    gx = (st.W + 15) // 16
    t = (y // 16) * gx + x // 16
    lst = st.point_list[st.ranges[t, 0]:st.ranges[t, 1]].astype(np.int64)
    A_THR, T_THR, A_MAX = float(np.float32(1.0) / np.float32(255.0)), float(np.float32(0.0001)), float(np.float32(0.99))
    bg = st.inputs["bg"].astype(np.float64)
    if lst.size == 0:
        return [(0, 1.0, bg.copy(), [])]
    m = st.means2D[lst]
    dx = (m[:, 0] - np.float32(x)).astype(np.float64)              # the fp32 subtraction both sides perform (forward.cu:342)
    dy = (m[:, 1] - np.float32(y)).astype(np.float64)
    co = st.conic_opacity[lst].astype(np.float64)
    qa, qc, xb = 0.5 * co[:, 0] * dx * dx, 0.5 * co[:, 2] * dy * dy, co[:, 1] * dx * dy
    power = -(qa + qc) - xb
    eP = _U * (4.0 * (np.abs(qa) + np.abs(qc) + np.abs(xb)) + np.abs(power))
    with np.errstate(over="ignore", invalid="ignore"):
        alpha = np.minimum(A_MAX, co[:, 3] * np.exp(np.minimum(power, 50.0)))
    eA = eP + 3.0 * _U                                              # relative
    col = feat[lst].astype(np.float64)
    # entries every fp32 evaluation skips: power clearly positive, or alpha clearly below 1/255
    surely_skipped = (power > eP) | ((alpha < A_THR) & (np.abs(alpha - A_THR) > alpha * eA))
    cand = np.nonzero(~surely_skipped)[0]
    leaves = []
    # depth-first over in-band decisions; state = (index into cand, T, relative error bound of T, colour, last, decisions)
    stack = [(0, 1.0, 0.0, np.zeros(3), 0, [])]
    while stack:
        ci, T, eT, C, last, dec = stack.pop()
        stopped = False
        while ci < cand.size:
            j = int(cand[ci]); ci += 1
            a, e = float(alpha[j]), float(eA[j])
            if abs(power[j]) <= eP[j]:                               # power > 0 ?  (in band: both ways)
                stack.append((ci, T, eT, C.copy(), last, dec + [("power>0", j, float(power[j] / eP[j]))]))   # the skip branch
            elif power[j] > 0:
                continue
            if abs(a - A_THR) <= a * e:                              # alpha < 1/255 ?  (in band: both ways)
                stack.append((ci, T, eT, C.copy(), last, dec + [("alpha<1/255", j, float((a - A_THR) / (a * e)))]))
                dec = dec + [("alpha>=1/255", j, float((a - A_THR) / (a * e)))]
            elif a < A_THR:
                continue
            test_T = T * (1.0 - a)
            eS = eT + a * e / (1.0 - a) + 2.0 * _U                   # relative error bound of test_T
            if abs(test_T - T_THR) <= T_THR * eS:                    # T (1 - alpha) < 1e-4 ?  (in band: both ways)
                leaves.append((last, T, C + T * bg, dec + [("stop", j, float((test_T - T_THR) / (T_THR * eS)))]))
                dec = dec + [("no stop", j, float((test_T - T_THR) / (T_THR * eS)))]
            elif test_T < T_THR:
                stopped = True
                break
            C = C + col[j] * (a * T)
            T, eT, last = test_T, eS, j + 1
        leaves.append((last, T, C + T * bg, dec))
        if len(leaves) + len(stack) > max_leaves:
            return None
    return leaves


def prove_flips(u, st, flipped):
    """For every flipped pixel: does the kernel's result equal a leaf of the float64 walk, and the oracle's another?
    -> dict(flips, outside_band, oracle_outside_band, decisions = [(kind, |margin| / band)] of the matched kernel leaves)."""
    feat = st.inputs["colors_precomp"] if st.inputs["colors_precomp"] is not None else st.rgb
    feat = feat.reshape(st.P, 3)
    out = dict(flips=int(flipped.sum()), outside_band=0, oracle_outside_band=0, decisions=[])

    def matches(leaf, nc, T, colour):
        n_l, T_l, C_l, _ = leaf
        return (n_l == int(nc) and abs(T_l - float(T)) <= 2e-4 * max(T_l, 1e-4)
                and bool(np.all(np.abs(C_l - colour.astype(np.float64)) <= 2e-5 + 2e-4 * np.abs(C_l))))

    ys, xs = np.nonzero(flipped)
    for y, x in zip(ys.tolist(), xs.tolist()):
        pix = y * st.W + x
        leaves = _pixel_leaves(st, feat, x, y)
        if leaves is None:
            out["outside_band"] += 1
            continue
        hip_leaf = [l for l in leaves if matches(l, u["n_contrib"][pix], u["final_T"][pix], u["out_color"][:, y, x])]
        orc_leaf = [l for l in leaves if matches(l, st.n_contrib[pix], st.final_T[pix], st.out_color[:, y, x])]
        if not hip_leaf:
            out["outside_band"] += 1
        else:
            best = min(hip_leaf, key=lambda l: len(l[3]))
            out["decisions"] += [(k, abs(r)) for k, _, r in best[3]]
        if not orc_leaf:
            out["oracle_outside_band"] += 1
    return out


def grad_errors(st, flipped, got, ref):
    """rel-inf error of every gradient, over everything and over the Gaussians (or codebook rows) that share NO 16x16 tile
    with a flipped pixel (the others legitimately differ by that pixel's term). -> (errs, errs_clean, affected count)"""
    gx = (st.W + 15) // 16
    ys, xs = np.nonzero(flipped)
    tiles = np.unique((ys // 16) * gx + xs // 16)
    affected = np.zeros(st.P, bool)
    for t in tiles:
        affected[st.point_list[st.ranges[t, 0]:st.ranges[t, 1]]] = True
    rows_of = {"dL_dsh": st.inputs["sh_indices"], "dL_dscales": st.inputs["g_indices"], "dL_drotations": st.inputs["g_indices"]}
    errs, errs_clean = {}, {}
    for k, v in got.items():
        r = ref[k]
        if r.size == 0 or v.shape != r.shape:
            continue
        if not np.isfinite(v).all():
            errs[k] = errs_clean[k] = float("inf")
            continue
        errs[k] = gpu_util.rel_inf(v, r)
        bad = affected
        if rows_of.get(k) is not None and r.shape[0] != st.P:       # codebook-sized: rows any affected Gaussian points at
            bad = np.zeros(r.shape[0], bool)
            bad[rows_of[k][affected]] = True
        d = np.abs(v.astype(np.float64) - r.astype(np.float64)).reshape(r.shape[0], -1).max(1)
        errs_clean[k] = float(d[~bad].max() / max(np.abs(r).max(), 1e-30)) if (~bad).any() else 0.0
    return errs, errs_clean, int(affected.sum())


def check_grads(st, u, got, ref, tol, what=""):
    """The gradient bar of the parity tests, aware of flipped pixels: every gradient within `tol` (rel-inf) -- or, when the
    forward has flipped pixels, within `tol` over everything that shares no tile with one and within 5 x tol overall
    (a flipped pixel adds or drops one whole contribution for the Gaussians of its tile). At most max(2, 2e-5 x pixels)
    pixels may flip, and EVERY flipped pixel must be proven a true fp32 borderline by prove_flips (the kernel's result and
    the oracle's each equal a leaf of a float64 walk that branches only at decisions inside their fp32 error band).
    -> number of flipped pixels"""
    flipped = flipped_pixels(u, st) if st.num_rendered > 0 else np.zeros((st.H, st.W), bool)
    n_flip = int(flipped.sum())
    assert n_flip <= max(2, int(2e-5 * st.W * st.H)), f"{what}: {n_flip} flipped pixels"
    if n_flip:          # every flip must be a PROVEN fp32 borderline (float64 walk, see prove_flips), not merely rare
        proof = prove_flips(u, st, flipped)
        assert proof["outside_band"] == 0 and proof["oracle_outside_band"] == 0, f"{what}: flip outside the fp32 band: {proof}"
    clean = grad_errors(st, flipped, got, ref)[1] if n_flip else None
    for k, v in got.items():
        r = ref[k]
        if r.size == 0 and v.size == 0:
            continue
        if v.shape != r.shape:          # absent inputs: reference shape [P,..] zeros vs oracle's empty
            assert r.size == 0 and not np.any(v), f"{what}: {k}"
            continue
        assert np.isfinite(v).all(), f"{what}: {k}"
        err = gpu_util.rel_inf(v, r)
        if clean is not None and k in clean:
            assert clean[k] <= tol, f"{what}: {k} rel-inf error {clean[k]:.3e} away from the {n_flip} flipped pixels"
            assert err <= 5 * tol, f"{what}: {k} rel-inf error {err:.3e} with {n_flip} flipped pixels"
        else:
            assert err <= tol, f"{what}: {k} rel-inf error {err:.3e}"
    return n_flip


def compare(inp, cam, indexed, st, ref=None, dL=None, fw=None):
    """HIP (through the C-ABI front-end, tests/gpu_util.py) against an oracle state `st` (+ gradients `ref` for `dL`).
    -> dict of plain numbers (JSON-serialisable)."""
    if fw is None:
        fw = gpu_util.hip_forward(inp, cam, indexed)
    u = gpu_util.unpack(fw)
    out = {"gaussians": int(st.P), "tile_instances": int(st.num_rendered), "num_rendered_equal": bool(u["num_rendered"] == st.num_rendered)}
    out["radii_mismatches"] = int((u["radii"] != st.radii).sum())
    out["tiles_touched_mismatches"] = int((u["tiles_touched"] != st.tiles_touched).sum())
    same_R = u["num_rendered"] == st.num_rendered
    out["sorted_keys_mismatches"] = int((u["keys_sorted"] != st.keys_sorted).sum()) if same_R else -1
    out["point_list_mismatches"] = int((u["point_list"] != st.point_list).sum()) if same_R else -1
    out["ranges_mismatches"] = int((u["ranges"] != st.ranges).sum())
    vis = st.radii > 0
    out["splat_float_bit_mismatches"] = int(
        (u["means2D"][vis].view(np.uint32) != st.means2D[vis].view(np.uint32)).sum()
        + (u["conic_opacity"][vis].view(np.uint32) != st.conic_opacity[vis].view(np.uint32)).sum()
        + (u["depths"][vis].view(np.uint32) != st.depths[vis].view(np.uint32)).sum()
        + ((u["rgb"][vis].view(np.uint32) != st.rgb[vis].view(np.uint32)).sum() if st.inputs["colors_precomp"] is None else 0))
    a, b = u["out_color"], st.out_color
    out["image_finite"] = bool(np.isfinite(a).all())
    out["psnr_db"] = float(gpu_util.psnr(a, b))
    target = np.random.default_rng(5).random(a.shape, dtype=np.float32)
    out["delta_psnr_db"] = float(abs(gpu_util.psnr(a, target) - gpu_util.psnr(b, target)))
    out["image_max_abs_err"] = float(np.abs(a - b).max())
    out["n_contrib_agreement"] = float((u["n_contrib"] == st.n_contrib).mean())
    rg = st.ranges.astype(np.int64)
    out["deepest_tile_list"] = int((rg[:, 1] - rg[:, 0]).max()) if rg.size else 0
    out["deepest_blend"] = int(st.n_contrib.max()) if st.n_contrib.size else 0
    flipped = flipped_pixels(u, st)
    out["flipped_pixels"] = int(flipped.sum())
    proof = prove_flips(u, st, flipped)
    out["flips_outside_band"] = proof["outside_band"] + proof["oracle_outside_band"]
    out["flip_decisions"] = sorted({k for k, _ in proof["decisions"]})
    out["flip_margin_over_band_max"] = max([r for _, r in proof["decisions"]], default=0.0)
    if ref is not None:
        got = gpu_util.hip_backward(fw, dL)
        errs, errs_clean, n_aff = grad_errors(st, flipped, got, ref)
        out["gaussians_sharing_a_tile_with_a_flip"] = n_aff
        out["grad_rel_inf"] = errs
        out["grad_rel_inf_max"] = max(errs.values()) if errs else 0.0
        out["grad_rel_inf_excluding_flips"] = errs_clean
        out["grad_rel_inf_excluding_flips_max"] = max(errs_clean.values()) if errs_clean else 0.0
    return out
