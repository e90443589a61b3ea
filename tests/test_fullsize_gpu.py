"""-m gpu parity at BASELINE.json's FULL sizes: the HIP path against the CPU oracle on the very workloads bench.py times
(1920x1080, 1M / 3M / 6M Gaussians) and the VQ shapes of configs[3] (covariance batch 2^20 x 2048 x 6, 5.4M-point final
assignment).  Bars are the north-star ones (tests/test_raster_gpu.py): integer results equal, PSNR >= 80 dB, dPSNR <= 0.05 dB,
gradients rel-inf <= 1e-4.  The oracle (OpenMP) needs a few seconds per view on the GPU box's host cores."""
import numpy as np
import pytest
import torch

from tests import fullsize, synth

pytestmark = pytest.mark.gpu

# Gradient bars at full size (measured on MI355X, gpurun_out/r02_diag_exp*.json -> DESIGN.md section 2):
#  * Of the ~10^9 blend decisions of a 1080p view (alpha >= 1/255, T < 1e-4) about TEN pixels (5e-6 of the image) fall on the
#    other side between the HIP path and the oracle, because `power` is rounded differently (explicit FMAs here, one rounding
#    per operation in the oracle; nvcc contracts the reference's expression in its own way again) -- with a compensated 1-ulp
#    exp() instead of __expf the same pixels flip and the gradient errors are unchanged, so exp() is not the cause.
#    A flipped pixel adds / removes one pixel's term for the Gaussians covering it: up to 4.2e-4 of a tensor's largest entry.
#  * Every Gaussian that does NOT share a 16x16 tile with such a pixel (99.3 % of them) meets the north-star bar of 1e-4
#    (measured <= 5.7e-5 on all tensors of all configurations).
#  * Round 3: the allowance is no longer count-based only. Every flipped pixel is re-walked in float64 from the bit-identical
#    splat records; the walk branches only at decisions whose float64 value lies inside the fp32 error band of its threshold
#    (band derived in tests/fullsize.py) and both results must be leaves of it. A flip outside the band fails the test.
GRAD_TOL = 1e-4            # Gaussians (codebook rows) not sharing a tile with a flipped pixel
GRAD_TOL_FLIPPED = 1e-3    # everything, flipped pixels included
MAX_FLIPPED_FRACTION = 1e-5


def _run(name, backward=True):
    from oracle import oracle as orc
    inp, intr, ev, indexed = fullsize.config_inputs(name)
    cam = orc.camera(intr.numpy(), ev.numpy())
    dL = synth.grad_image(cam["W"], cam["H"]).numpy() if backward else None
    st, ref, _, _ = fullsize.oracle_view(inp, cam, dL)
    res = fullsize.compare(inp, cam, indexed, st, ref, dL)
    print(name, {k: v for k, v in res.items() if k != "grad_rel_inf"}, res.get("grad_rel_inf"))
    return res


def _assert_bars(res, backward=True):
    assert res["num_rendered_equal"]
    for k in ("radii_mismatches", "tiles_touched_mismatches", "sorted_keys_mismatches", "point_list_mismatches",
              "ranges_mismatches", "splat_float_bit_mismatches"):
        assert res[k] == 0, (k, res[k])
    assert res["image_finite"] and res["psnr_db"] >= 80.0, res["psnr_db"]
    assert res["delta_psnr_db"] <= 0.05
    assert res["n_contrib_agreement"] >= 0.999
    assert res["flipped_pixels"] <= MAX_FLIPPED_FRACTION * 1920 * 1080, res["flipped_pixels"]
    # every flipped pixel is a PROVEN fp32 borderline: the kernel's and the oracle's results are both leaves of a float64
    # walk that branches only inside the fp32 error band of a threshold (tests/fullsize.py:prove_flips)
    assert res["flips_outside_band"] == 0, res
    if backward:
        for k, e in res["grad_rel_inf_excluding_flips"].items():
            assert e <= GRAD_TOL, f"{k}: rel-inf {e:.3e} away from flipped pixels"
        for k, e in res["grad_rel_inf"].items():
            assert e <= GRAD_TOL_FLIPPED, f"{k}: rel-inf {e:.3e} ({res['flipped_pixels']} flipped pixels)"


def test_config3_headline_3M_indexed_1080p_fwd_bwd(hip, orc):
    """BASELINE.json configs[2], exactly the workload bench.py times (rasterizer_impl.cu:440-697)."""
    res = _run("config3_3M_indexed")
    assert res["gaussians"] == 3_000_000 and res["tile_instances"] > 10_000_000
    _assert_bars(res)


def test_config2_1M_forward_1080p(hip, orc):
    """BASELINE.json configs[1]: forward-only render.py path, non-indexed (rasterizer_impl.cu:194-334)."""
    res = _run("config2_1M_fwd", backward=False)
    _assert_bars(res, backward=False)


def test_config4_sensitivity_shape_6M_cov_precomp_no_clamp(hip, orc):
    """The sensitivity pass of configs[3]/[4] (compress.py:81-119): 6M Gaussians, non-indexed, cov3D_precomp, clamp_color=False;
    the gradients it consumes are dL_dsh and dL_dcov3D."""
    res = _run("config4_6M_sens")
    assert res["gaussians"] == 6_000_000
    _assert_bars(res)
    assert "dL_dcov3D" in res["grad_rel_inf"] and "dL_dsh" in res["grad_rel_inf"]


def _exactness_properties(hip, x, cb, d, i, g, probes=64, rows=4096):
    exact = ((x - cb[i]) ** 2).sum(-1)
    torch.testing.assert_close(d, exact, rtol=1e-5, atol=1e-7)
    probe = torch.randint(0, cb.shape[0], (probes,), generator=g).cuda()
    other = ((x[:rows, None, :] - cb[probe][None]) ** 2).sum(-1)
    assert (d[:rows, None] <= other * (1 + 1e-5) + 1e-7).all()


def test_vq_covariance_batch_full_size(hip, orc):
    """configs[3] covariance codebook: batch 2^20 x K=2048 x D=6 (the LDS-accumulate shape). 2^15 points spread over the whole
    batch (stride 32) are checked bit-exactly against the oracle, all of them through the exactness properties, and one Lloyd
    step's sums against float64 index_add of the same assignment."""
    g = torch.Generator().manual_seed(8)
    B, K, D = 2 ** 20, 2048, 6
    x = (torch.randn(B, D, generator=g) * 0.1).float()
    x[:, [0, 3, 5]] = x[:, [0, 3, 5]].abs() + 0.2
    cb = x[torch.randperm(B, generator=g)[:K]].clone().contiguous()
    w = torch.rand(B, generator=g).pow(4).float()
    xd, cbd, wd = x.cuda(), cb.cuda(), w.cuda()
    d, i = hip.weightedDistance(xd, cbd)
    sel = torch.arange(0, B, B // 2 ** 15)[:2 ** 15]              # a strided sample over the whole batch, not its first points
    d_ref, i_ref = orc.weighted_distance(x[sel].numpy(), cb.numpy())
    np.testing.assert_array_equal(i[sel.cuda()].cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d[sel.cuda()].cpu().numpy().view(np.uint32), d_ref.view(np.uint32))
    _exactness_properties(hip, xd, cbd, d, i, g)
    vqm = hip.VectorQuantize(D, K, decay=0.8).cuda()
    vqm.codebook.data = cbd.clone()
    _, S, dsum = vqm.partial_sums(xd, wd)
    S_ref = torch.zeros(K, D + 1, dtype=torch.float64, device="cuda")
    S_ref.index_add_(0, i, torch.cat([xd.double() * wd.double()[:, None], wd.double()[:, None]], 1))
    torch.testing.assert_close(S.double(), S_ref, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(dsum, d.double().sum().reshape(1), rtol=1e-9, atol=0)


def test_vq_final_assignment_5p4M(hip, orc):
    """configs[3] colour codebook, the final assignment of all N = 5.4M points (vq.py:79-82): K=4096, D=48. Oracle-exact
    on a 2^14-point sample spread over the range, properties on everything, codewords map to themselves."""
    g = torch.Generator().manual_seed(9)
    N, K, D = 5_400_000, 4096, 48
    x = (torch.randn(N, D, generator=g) * 0.1).float()
    cb = (torch.randn(K, D, generator=g) * 0.1).float()
    xd, cbd = x.cuda(), cb.cuda()
    d, i = hip.weightedDistance(xd, cbd)
    assert d.shape == (N,) and int(i.min()) >= 0 and int(i.max()) < K
    sel = torch.arange(0, N, N // 2 ** 14)[:2 ** 14]
    d_ref, i_ref = orc.weighted_distance(x[sel].numpy(), cb.numpy())
    np.testing.assert_array_equal(i[sel.cuda()].cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d[sel.cuda()].cpu().numpy().view(np.uint32), d_ref.view(np.uint32))
    for lo in range(0, N, 1_350_000):                      # the exact-distance property in slices (memory)
        hi = min(N, lo + 1_350_000)
        exact = ((xd[lo:hi] - cbd[i[lo:hi]]) ** 2).sum(-1)
        torch.testing.assert_close(d[lo:hi], exact, rtol=1e-5, atol=1e-7)
    _exactness_properties(hip, xd, cbd, d, i, g)


@pytest.mark.parametrize("scene", ["opaque", "translucent"])
def test_config5_composed_pipeline_6M(hip, tmp_path, scene):
    """BASELINE.json configs[4] at full scene size through the public API (tools/run_config5.py: 6M Gaussians, 32 cameras at
    1080p, sensitivity pass -> prune + VQ with the reference's settings (colour 100 steps of 2^18, covariance 800 steps of
    2^20, 2^12 codebooks) -> QAT fine-tuning -> Morton-sorted npz). Only the number of fine-tuning iterations is cut (60 of
    5000: a fixed-cost loop of identical steps). Two scenes: synth-v1 as benched (`opaque`: the reference's zero-importance
    prune, compression/vq.py:205-211, keeps 0.34M of the 6M because 32 yaw cameras never blend the rest) and `translucent`
    (opacity sigmoid(N(-3, 1)): rays go deep, >= 3M survive, so the QAT stage runs at configs[2]'s scale; its camera 0 is also
    rendered by the CPU oracle). Checked: it runs, the payload is a real compression, and the compressed model still renders
    the uncompressed model's images (PSNR)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "config5.json"
    cmd = [sys.executable, "tools/run_config5.py", "--finetune", "60", "--scene", scene, "--out", str(out)]
    if scene == "translucent":
        cmd.append("--oracle-psnr")
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(out))
    assert res["gaussians"] == 6_000_000 and res["cameras"] == 32 and res["resolution"] == [1920, 1080]
    t = res["timings_s"]
    assert t["finetune_iterations"] == 60 and t["clustering"] < 60 and t["sensitivity_calculation"] < 60
    if scene == "opaque":
        assert res["compression_ratio"] > 50 and 0 < res["payload_MiB"] < res["uncompressed_fp32_MiB"]
        assert res["psnr_vs_uncompressed_after_vq_dB"] >= 40.0 and res["psnr_vs_uncompressed_after_finetune_dB"] >= 40.0
        assert 0 < t["finetune_ms_per_iteration"] < 50
    else:
        assert res["survivors_of_the_importance_prune"] >= 3_000_000, res["survivors_of_the_importance_prune"]
        assert res["compression_ratio"] > 8 and 0 < res["payload_MiB"] < res["uncompressed_fp32_MiB"]
        assert res["psnr_vs_uncompressed_after_vq_dB"] >= 40.0 and res["psnr_vs_uncompressed_after_finetune_dB"] >= 35.0
        assert 0 < t["finetune_ms_per_iteration"] < 100
        po = res["psnr_vs_oracle_camera0"]            # the HIP renders against the CPU oracle's render of the uncompressed scene
        assert po["uncompressed_hip_vs_oracle_dB"] >= 45.0 and po["compressed_hip_vs_oracle_uncompressed_dB"] >= 35.0, po
    print("config5", scene, {k: round(v, 3) if isinstance(v, float) else v for k, v in res.items() if k != "timings_s"}, t)
