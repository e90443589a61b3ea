"""-m gpu parity tests for the VQ path: weightedDistance is BIT-EXACT (fp32 distances and int64 indices) against
the oracle's k-ordered FMA chain; the Lloyd update matches to 1e-5 relative (fp32 atomics vs float64 sums)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(N, C, K, seed, scale=0.1):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, K, generator=g) * scale).float()
    cb = (torch.randn(C, K, generator=g) * scale).float()
    return x, cb


@pytest.mark.parametrize("N,C,K", [(1000, 256, 12), (5000, 4096, 48), (3001, 2048, 6), (777, 100, 48), (513, 33, 5),
                                   (64, 1, 48), (2000, 300, 27), (1, 7, 6)])
def test_weighted_distance_bit_exact(hip, orc, N, C, K):
    x, cb = _data(N, C, K, seed=N + C + K)
    d_ref, i_ref = orc.weighted_distance(x.numpy(), cb.numpy())
    d, i = hip.weightedDistance(x.cuda(), cb.cuda())
    assert d.dtype == torch.float32 and i.dtype == torch.int64 and d.is_cuda
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d_ref.view(np.uint32))


def test_weighted_distance_ties_lowest_index_wins(hip, orc):
    x, cb = _data(300, 64, 12, seed=1)
    cb = torch.cat([cb, cb, cb[:10]], 0).contiguous()          # every codeword duplicated: strict '<' keeps the first
    d, i = hip.weightedDistance(x.cuda(), cb.cuda())
    d_ref, i_ref = orc.weighted_distance(x.numpy(), cb.numpy())
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    assert i.max().item() < 64


def test_weighted_distance_gather(hip, orc):
    x, cb = _data(4000, 512, 48, seed=2)
    g = torch.Generator().manual_seed(3)
    batch = torch.randint(0, 4000, (2500,), generator=g)
    d, i = hip.weightedDistance(x.cuda(), cb.cuda(), gather=batch.cuda())
    d_ref, i_ref = orc.weighted_distance(x[batch].numpy(), cb.numpy())
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d_ref.view(np.uint32))


def test_weighted_distance_errors(hip):
    x, cb = _data(10, 4, 6, seed=0)
    with pytest.raises(RuntimeError, match="dimension 2"):
        hip.weightedDistance(x.cuda()[0], cb.cuda())
    with pytest.raises(RuntimeError, match="same number of channels"):
        hip.weightedDistance(x.cuda(), cb.cuda()[:, :5])
    with pytest.raises(RuntimeError, match="GPU"):
        hip.weightedDistance(x, cb)
    d, i = hip.weightedDistance(x.cuda()[:0], cb.cuda())
    assert d.shape == (0,) and i.shape == (0,)


@pytest.mark.parametrize("D,K,scale_normalize", [(12, 256, False), (48, 512, False), (6, 128, True)])
def test_vq_update_matches_oracle(hip, orc, D, K, scale_normalize):
    g = torch.Generator().manual_seed(11)
    B = 20000
    x = (torch.randn(B, D, generator=g) * 0.1).float()
    if D == 6:
        x[:, [0, 3, 5]] = x[:, [0, 3, 5]].abs() + 0.2
    w = torch.rand(B, generator=g).pow(4).float()
    cb0 = x[torch.randperm(B, generator=g)[:K]].clone().contiguous()
    vqm = hip.VectorQuantize(D, K, decay=0.8).cuda()
    vqm.codebook.data = cb0.cuda().clone()
    cb_ref = cb0.numpy().copy()
    ent_ref = np.zeros(K, np.float32)
    for step in range(3):
        md = vqm.update(x.cuda(), w.cuda())
        if scale_normalize:
            tr = vqm.codebook[:, [0, 3, 5]].sum(-1)
            vqm.codebook /= tr[:, None]
        md_ref, ix_ref, mean_ref = orc.vq_update(x.numpy(), w.numpy(), cb_ref, ent_ref, 0.8, 1e-5)
        if scale_normalize:
            from oracle.oracle import lib, _p, _f32p
            lib().orc_vq_trace_normalize(K, D, _p(cb_ref, _f32p))
        np.testing.assert_allclose(vqm.codebook.data.cpu().numpy(), cb_ref, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(vqm.entry_importance.data.cpu().numpy(), ent_ref, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(md.cpu().numpy(), md_ref, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("D,K,tails", [(48, 4096, (17, 21, 25, 63, 1)), (6, 4096, (10, 19, 33, 5)), (12, 2048, (7, 29, 50))])
def test_vq_accumulate_ragged_tail_chunks(hip, D, K, tails):
    """c3dgs_vq_accumulate on batches whose LAST 64-point chunk is ragged: the dense (no shared codeword) path broadcasts a
    point's row / weight / codeword by lane shuffles, which must run with every lane active (round-2 advisory: tails with
    (cnt * (D+1)) % 64 in 1..cnt-1 lost contributions). K * (D+1) > 16384 selects that kernel; all indices distinct so the
    dense path is the one taken. Checked against float64 sums of compression/vq.py:31-34."""
    import ctypes as C
    from c3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(77 + D)
    for tail in tails:
        for B in (tail, 64 * 5 + tail):
            x = torch.randn(B, D, generator=g).float()
            w = (torch.rand(B, generator=g) + 0.5).float()
            idx = torch.randperm(K, generator=g)[:B].contiguous()          # distinct codewords -> no merging
            S = torch.zeros(K, D + 1, device="cuda")
            xd, wd, id_ = x.cuda(), w.cuda(), idx.cuda()
            rc = L.c3dgs_vq_accumulate(B, K, D, xd.data_ptr(), wd.data_ptr(), None, id_.data_ptr(), None, S.data_ptr(), None,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
            ref = np.zeros((K, D + 1))
            np.add.at(ref[:, :D], idx.numpy(), x.numpy().astype(np.float64) * w.numpy()[:, None].astype(np.float64))
            np.add.at(ref[:, D], idx.numpy(), w.numpy().astype(np.float64))
            np.testing.assert_allclose(S.cpu().numpy(), ref, rtol=1e-6, atol=1e-7, err_msg=f"D={D} B={B}")


def test_vq_features_matches_oracle(hip, orc):
    """Config 1 of BASELINE.json at its FULL size (10k Gaussians, SH degree 1 -> D=12, K=256, 100 steps of 2^14), RNG draws
    passed as data to both sides."""
    g = torch.Generator().manual_seed(0)
    N, D, K, steps, chunk = 10_000, 12, 256, 100, 2 ** 14
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    imp = torch.rand(N, generator=g).pow(4).float()
    init = torch.rand(K, D, generator=g)
    batches = [torch.randint(0, N, (chunk,), generator=g) for _ in range(steps)]
    cb, idx, errs = hip.vq_features(f.cuda(), imp.cuda(), K, chunk, steps, batches=batches, init_rand=init, silent=True,
                                    return_errors=True)
    cb_ref, idx_ref, err_ref, _ = orc.vq_features(f.numpy(), imp.numpy(), K, init.numpy(), [b.numpy() for b in batches])
    np.testing.assert_allclose(cb.cpu().numpy(), cb_ref, rtol=1e-4, atol=1e-6)
    agree = (idx.cpu().numpy() == idx_ref).mean()
    assert agree >= 0.999, agree
    np.testing.assert_allclose(np.array(errs), err_ref, rtol=1e-4)


@pytest.mark.parametrize("name", ["vq_config0.npz", "vq_d48.npz"])
def test_vq_features_hip_vs_reference_fixture_from_seed_alone(hip, name):
    """The HIP path DIRECTLY against the reference's own vq_features (compression/vq.py:49-87, run by
    tests/golden/make_golden.py from torch.manual_seed alone): vq_config0.npz is BASELINE.json configs[0] exactly (N = 10k,
    D = 12, K = 256, 100 x 2^14), vq_d48.npz the colour codebook's width (N = 20k, D = 48, K = 512, 8 x 4096). The batches
    are NOT data here: the package continues the CPU generator's MT19937 stream itself (_BatchDraws / csrc/draws.hip)."""
    from tests import vq_fixture
    fx = vq_fixture.load(name)
    vq_fixture.seed_like_reference(fx)
    cb, idx = hip.vq_features(fx["features"].cuda(), fx["importance"].cuda(), fx["K"], fx["chunk"], fx["steps"], silent=True,
                              init_rand=fx["init_rand"])
    np.testing.assert_allclose(cb.cpu().numpy(), fx["codebook"], rtol=1e-4, atol=2e-7)
    assert (idx.cpu().numpy() == fx["indices"]).mean() >= 0.999
    # the generator is left where the reference's loop leaves it: the next draw equals the one after `steps` batches
    nxt = torch.randint(0, 1 << 30, (4,))
    vq_fixture.seed_like_reference(fx, model_built_inside=False)
    for _ in range(fx["steps"]):
        torch.randint(low=0, high=fx["N"], size=[fx["chunk"]])
    assert torch.equal(nxt, torch.randint(0, 1 << 30, (4,)))


def test_weighted_distance_full_size_properties(hip):
    """BASELINE.json config 4 colour shape (batch 2^18 x 4096 x 48): checked through properties the domain
    offers: the returned distance equals the exact distance to the returned codeword, no sampled codeword is
    closer, and codewords themselves map to distance 0 at their own (lowest duplicate) index."""
    g = torch.Generator().manual_seed(5)
    N, C, K = 2 ** 18, 4096, 48
    x = (torch.randn(N, K, generator=g) * 0.1).float().cuda()
    cb = (torch.randn(C, K, generator=g) * 0.1).float().cuda()
    d, i = hip.weightedDistance(x, cb)
    exact = ((x - cb[i]) ** 2).sum(-1)
    torch.testing.assert_close(d, exact, rtol=1e-5, atol=1e-7)
    probe = torch.randint(0, C, (64,), generator=g).cuda()
    other = ((x[:, None, :] [:4096] - cb[probe][None]) ** 2).sum(-1)
    assert (d[:4096, None] <= other * (1 + 1e-5) + 1e-7).all()
    d2, i2 = hip.weightedDistance(cb, cb)
    assert (d2 == 0).all() and (i2 == torch.arange(C, device="cuda")).all()


def test_batch_draws_on_device_equal_the_cpu_generator_stream(hip):
    """The Lloyd-step draws come from the library's MT19937 continuation + `% N` on the GPU (csrc/draws.hip): same numbers
    as `torch.randint(0, N, [chunk])` on the CPU default generator (reference vq.py:69), more steps than the pinned ring
    holds, and the torch generator ends where the reference's loop would leave it."""
    from c3dgs_amd.vq import _BatchDraws
    N, chunk, steps = 4_500_123, 70_001, 11
    torch.manual_seed(21)
    torch.rand(3)
    ref = [torch.randint(low=0, high=N, size=[chunk]) for _ in range(steps)]
    after = torch.rand(4)
    torch.manual_seed(21)
    torch.rand(3)
    d = _BatchDraws(N, chunk, steps, torch.device("cuda", 0))
    assert d.key is not None                             # the fast path is the one under test
    got = [d.next_batch().clone() for _ in range(steps)]       # the returned tensor is the loop's staging buffer: reused per call
    d.finish()
    for a, b in zip(ref, got):
        assert b.dtype == torch.int64 and torch.equal(a, b.cpu())
    assert torch.equal(torch.rand(4), after)
    # a loop that stops early: the batch drawn ahead is taken back, torch continues as after 5 reference draws
    torch.manual_seed(22)
    ref5 = [torch.randint(low=0, high=N, size=[chunk]) for _ in range(5)]
    after5 = torch.rand(4)
    torch.manual_seed(22)
    d = _BatchDraws(N, chunk, steps, torch.device("cuda", 0))
    got5 = [d.next_batch().clone() for _ in range(5)]
    d.finish()
    assert all(torch.equal(a, b.cpu()) for a, b in zip(ref5, got5)) and torch.equal(torch.rand(4), after5)
    # a rank's slice: the whole batch is drawn, only [lo, hi) is uploaded and converted
    torch.manual_seed(22)
    d = _BatchDraws(N, chunk, steps, torch.device("cuda", 0))
    lo, hi = chunk // 8 * 3, chunk // 8 * 4 + 5
    sl = [d.next_batch(lo, hi).clone() for _ in range(5)]
    d.finish()
    assert all(torch.equal(a[lo:hi], b.cpu()) for a, b in zip(ref5, sl)) and torch.equal(torch.rand(4), after5)


def test_vq_features_default_draws_equal_explicit_batches(hip):
    """vq_features with the library's MT19937 continuation == vq_features calling torch.randint per step, as the reference
    does (up to the order of the float atomics in the centroid sums)."""
    from c3dgs_amd import vq as vqm
    g = torch.Generator().manual_seed(1)
    N, D, K, steps, chunk = 20_000, 6, 64, 9, 2 ** 11
    f = (torch.randn(N, D, generator=g) * 0.1).float().cuda()
    f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, generator=g).pow(4).float().cuda()
    init = torch.rand(K, D, generator=g)
    torch.manual_seed(77)
    vqm._FAST_DRAWS = False
    try:
        cb_a, idx_a = hip.vq_features(f, imp, K, chunk, steps, init_rand=init, silent=True, scale_normalize=True)
    finally:
        vqm._FAST_DRAWS = True
    end_a = torch.rand(3)
    torch.manual_seed(77)
    cb_b, idx_b = hip.vq_features(f, imp, K, chunk, steps, init_rand=init, silent=True, scale_normalize=True)
    assert torch.equal(torch.rand(3), end_a)             # the torch generator is left in the same state
    assert torch.allclose(cb_a, cb_b, rtol=1e-4, atol=1e-6)
    assert (idx_a == idx_b).float().mean() >= 0.999
    # and a different seed really gives different draws (the comparison above is not vacuous)
    torch.manual_seed(78)
    cb_c, _ = hip.vq_features(f, imp, K, chunk, steps, init_rand=init, silent=True, scale_normalize=True)
    assert not torch.allclose(cb_a, cb_c, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("seed", list(range(40)))
def test_weighted_distance_random_shapes_and_near_ties(hip, orc, seed):
    """Random N, C, K (MFMA widths 6 / 12 / 48 and others, C below and above the MFMA path's minimum, ragged tiles), with
    the things that stress the exactness logic mixed in: duplicated and nearly duplicated codewords, points sitting on
    codewords and on midpoints between two of them, large offsets (||x|| >> distances), a gather. Bit-exact."""
    r = np.random.default_rng(seed)
    K = int(r.choice([6, 12, 48, 48, 6, int(r.integers(1, 65))]))
    C = int(r.choice([1, 31, 32, 33, 127, 128, 129, 4096, int(r.integers(2, 3000))]))
    N = int(r.choice([1, 63, 64, 65, 255, 257, int(r.integers(2, 20000))]))
    g = torch.Generator().manual_seed(seed)
    scale = float(np.exp(r.uniform(np.log(1e-3), np.log(10.0))))
    offset = float(r.choice([0.0, 0.0, 5.0, 100.0])) * scale
    cb = torch.randn(C, K, generator=g) * scale + offset
    if C > 4:
        nd = max(1, C // 8)
        src = torch.randint(0, C, (nd,), generator=g)
        dst = torch.randint(0, C, (nd,), generator=g)
        cb[dst] = cb[src]                                               # exact duplicates: lowest index must win
        dst2 = torch.randint(0, C, (nd,), generator=g)
        cb[dst2] = cb[src] * (1 + 1e-7) + 1e-9 * scale                  # near duplicates: inside the ambiguity window
    x = torch.randn(N, K, generator=g) * scale + offset
    if N > 8:
        x[::5] = cb[torch.randint(0, C, (len(x[::5]),), generator=g)]   # points on codewords (distance 0)
        i, j = torch.randint(0, C, (2, len(x[1::7])), generator=g)
        x[1::7] = 0.5 * (cb[i] + cb[j])                                 # midpoints: two exactly (or almost) equal distances
    x, cb = x.float().contiguous(), cb.float().contiguous()
    gather = None
    if r.random() < 0.4 and N > 1:
        gather = torch.randint(0, N, (N,), generator=g)
    d_ref, i_ref = orc.weighted_distance(x.numpy() if gather is None else x[gather].numpy(), cb.numpy())
    if gather is None:
        d, i = hip.weightedDistance(x.cuda(), cb.cuda())
    else:
        from c3dgs_amd.vq import weightedDistance
        d, i = weightedDistance(x.cuda(), cb.cuda(), gather.cuda())
    what = f"seed {seed}: N={N} C={C} K={K} scale={scale:.3g} offset={offset:.3g} gather={gather is not None}"
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref, err_msg=what)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d_ref.view(np.uint32), err_msg=what)


def test_weighted_distance_nan_rows_fall_back_to_codeword_zero(hip):
    """A query row of NaNs compares false against everything: like the reference's `if (d < min_dist)` it keeps index 0."""
    x, cb = _data(300, 512, 48, seed=9)
    x[7] = float("nan")
    x[200, 3] = float("nan")
    d, i = hip.weightedDistance(x.cuda(), cb.cuda())
    assert int(i[7]) == 0 and int(i[200]) == 0
    ok = torch.ones(300, dtype=torch.bool); ok[7] = ok[200] = False
    d2, i2 = hip.weightedDistance(x[ok].cuda(), cb.cuda())
    assert torch.equal(i[ok.cuda()], i2) and torch.equal(d[ok.cuda()], d2)
    # the same through the large-N form of the search (one 64-point set per wave, pipelined), ragged last codebook tile
    g = torch.Generator().manual_seed(10)
    xb = (torch.randn(70000, 48, generator=g) * 0.1).float()
    cbb = (torch.randn(300, 48, generator=g) * 0.1).float()
    xb[12345] = float("nan"); xb[69999, 47] = float("nan"); xb[64] = float("inf")
    db, ib = hip.weightedDistance(xb.cuda(), cbb.cuda())
    assert int(ib[12345]) == 0 and int(ib[69999]) == 0 and int(ib[64]) == 0 and 0 <= int(ib.min()) and int(ib.max()) < 300
    okb = torch.ones(70000, dtype=torch.bool); okb[12345] = okb[69999] = okb[64] = False
    d3, i3 = hip.weightedDistance(xb[okb].cuda(), cbb.cuda())
    assert torch.equal(ib[okb.cuda()], i3) and torch.equal(db[okb.cuda()], d3)


def test_split_scores_within_margin(hip):
    """The candidate search of K = 48 / 12 runs on the fp16 matrix cores with every operand (scaled by a common power of two)
    split into two fp16 pieces and three piece products per multiply (csrc/vq.hip). Its scores s[n][c] = ||c||^2 - 2 x_n.c only have to be accurate enough
    for the ambiguity margin (4e-5 of d + 2||x||^2 per candidate pair) to be a valid bound: measure them against float64."""
    import ctypes as C
    from c3dgs_amd import _lib
    L = _lib.lib()
    worst = 0.0
    for seed, scale, offset in [(0, 0.1, 0.0), (1, 1.0, 0.0), (2, 0.1, 0.5), (3, 10.0, 100.0), (4, 1e-3, 0.0), (5, 1.0, -3.0)]:
        g = torch.Generator().manual_seed(seed)
        N, Cn, K = 256, 1024 - 7 * seed, 48
        x = (torch.randn(N, K, generator=g) * scale + offset).float().cuda()
        cb = (torch.randn(Cn, K, generator=g) * scale * (1 + seed) + offset).float().cuda()
        nb = int(L.c3dgs_weighted_distance_ws_bytes(N, Cn, K))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        scores = torch.full((N, Cn), float("nan"), dtype=torch.float32, device="cuda")
        dist = torch.empty(N, dtype=torch.float32, device="cuda")
        idx = torch.empty(N, dtype=torch.int64, device="cuda")
        rc = L.c3dgs_debug_wd_scores(N, Cn, K, x.data_ptr(), cb.data_ptr(), scores.data_ptr(), ws.data_ptr(), nb,
                                     dist.data_ptr(), idx.data_ptr(), None)
        _lib.check(rc)
        torch.cuda.synchronize()
        xd, cd = x.double().cpu(), cb.double().cpu()
        exact = (cd ** 2).sum(-1)[None] - 2.0 * xd @ cd.T
        d = ((xd[:, None] - cd[None]) ** 2).sum(-1)
        rel = (scores.double().cpu() - exact).abs() / (d + 2.0 * (xd ** 2).sum(-1)[:, None])
        assert torch.isfinite(rel).all()
        worst = max(worst, float(rel.max()))
    print(f"split-fp16 score error: {worst:.3e} of (d + 2||x||^2)")
    assert worst < 5e-6, worst                                    # the margin is 4e-5 of (d_best + d_second + 2||x||^2)


@pytest.mark.parametrize("K", [48, 12, 6])
def test_split_search_extreme_magnitudes(hip, orc, K):
    """The fp16 pieces share ONE power-of-two scale taken from the codebook's largest magnitude. Whatever falls outside fp16's
    range after scaling -- points far larger than every codeword (inf / NaN scores), data near the fp32 limits, a codebook of
    zeros, an inf in the codebook -- must come out of the exact re-scan with the reference's answer, never a wrong index."""
    g = torch.Generator().manual_seed(100 + K)
    N, C = 3000, 300
    for scale_x, scale_c in [(1e-18, 1e-18), (1e15, 1e15), (1e3, 1.0), (1.0, 1e-4), (1e-30, 1e-30), (1.0, 0.0)]:
        x = (torch.randn(N, K, generator=g) * scale_x).float()
        cb = (torch.randn(C, K, generator=g) * scale_c).float()
        x[::7] = cb[torch.randint(0, C, (len(x[::7]),), generator=g)]
        x[5, 0] = 3.0e38                                                   # one point at the edge of fp32
        d_ref, i_ref = orc.weighted_distance(x.numpy(), cb.numpy())
        d, i = hip.weightedDistance(x.cuda(), cb.cuda())
        what = f"K={K} scale_x={scale_x} scale_c={scale_c}"
        np.testing.assert_array_equal(i.cpu().numpy(), i_ref, err_msg=what)
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d_ref.view(np.uint32), err_msg=what)
    x = (torch.randn(N, K, generator=g) * 0.1).float()
    cb = (torch.randn(C, K, generator=g) * 0.1).float()
    cb[17, 1] = float("inf")                                               # an inf codeword: distance inf (or NaN) to everything
    d_ref, i_ref = orc.weighted_distance(x.numpy(), cb.numpy())
    d, i = hip.weightedDistance(x.cuda(), cb.cuda())
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d_ref.view(np.uint32))


def test_fp32_and_split_search_agree_bit_for_bit(hip):
    """c3dgs_weighted_distance (no scratch: fp32 matrix cores) and c3dgs_weighted_distance_ws (split fp16) return the same
    distances and indices on data full of near ties; so does a scratch too small for the split codebook."""
    from c3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    N, Cn, K = 20000, 4096, 48
    cb = torch.randn(Cn, K, generator=g) * 0.1
    cb[100:600] = cb[1000:1500] * (1 + 1e-7)
    x = torch.randn(N, K, generator=g) * 0.1
    i, j = torch.randint(0, Cn, (2, len(x[1::3])), generator=g)
    x[1::3] = 0.5 * (cb[i] + cb[j])
    x, cb = x.float().cuda(), cb.float().cuda()
    d_b, i_b = hip.weightedDistance(x, cb)
    d_f = torch.empty(N, dtype=torch.float32, device="cuda"); i_f = torch.empty(N, dtype=torch.int64, device="cuda")
    _lib.check(L.c3dgs_weighted_distance(N, Cn, K, x.data_ptr(), None, cb.data_ptr(), d_f.data_ptr(), i_f.data_ptr(), None))
    d_s = torch.empty_like(d_f); i_s = torch.empty_like(i_f)
    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    _lib.check(L.c3dgs_weighted_distance_ws(N, Cn, K, x.data_ptr(), None, cb.data_ptr(), d_s.data_ptr(), i_s.data_ptr(),
                                            small.data_ptr(), int(small.numel()), None))
    torch.cuda.synchronize()
    assert torch.equal(i_b, i_f) and torch.equal(d_b.view(torch.int32), d_f.view(torch.int32))
    assert torch.equal(i_b, i_s) and torch.equal(d_b.view(torch.int32), d_s.view(torch.int32))


_NCCL_CHILD = r"""
import os, torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"          # MASTER_PORT: a free port chosen by the parent test
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import c3dgs_amd
g = torch.Generator().manual_seed(3)
N, D, K, steps, chunk = 60_000, 48, 256, 6, 2 ** 13
f = (torch.randn(N, D, generator=g) * 0.1).float().cuda()
imp = torch.rand(N, generator=g).pow(4).float().cuda()
init = torch.rand(K, D, generator=g)
torch.manual_seed(5)
cb_g, idx_g, err_g = c3dgs_amd.vq_features(f, imp, K, chunk, steps, init_rand=init, silent=True, group=True, return_errors=True)
torch.manual_seed(5)
cb_s, idx_s, err_s = c3dgs_amd.vq_features(f, imp, K, chunk, steps, init_rand=init, silent=True, return_errors=True)
assert torch.allclose(cb_g, cb_s, rtol=1e-4, atol=1e-6) and (idx_g == idx_s).float().mean() >= 0.999
assert all(abs(a - b) <= 1e-6 * abs(b) + 1e-12 for a, b in zip(err_g, err_s)), (err_g, err_s)
dist.destroy_process_group()
print("NCCL_OK")
"""


def test_sharded_path_runs_on_the_nccl_backend():
    """The sharded Lloyd loop's collectives (generator-state broadcast, in-place all-reduce of S, all-reduce of the error sums,
    all_gather of the final assignment) on the `nccl` (= RCCL) backend with device tensors -- a one-rank group is what a
    one-GPU box can host; the multi-rank logic is covered over gloo (tests/test_dist_cpu.py)."""
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    r = subprocess.run([sys.executable, "-c", _NCCL_CHILD], cwd=root, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(port)))
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("D,K,B,scale_normalize", [(48, 4096, 2 ** 15, False), (48, 512, 3000, False), (6, 2048, 2 ** 17, True),
                                                   (12, 256, 1000, False)])
def test_fused_lloyd_step_is_bit_identical_to_the_plain_pair(hip, D, K, B, scale_normalize):
    """vq_features' loop runs a Lloyd step as c3dgs_vq_step_sums + c3dgs_vq_step_apply (5 launches: the update also splits
    the new codebook for the next search, clears S and the list counter; the distance sum rides in the accumulation).
    Same single-rounded update, exact search: codebooks, indices and entry importances must equal the plain
    c3dgs_vq_sums / c3dgs_vq_apply pair bit for bit (sharded ranks rely on it), the per-step errors to summation order."""
    from c3dgs_amd import vq as vqm
    g = torch.Generator().manual_seed(100 + D)
    N, steps = 50_000, 7
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    if scale_normalize:
        f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, generator=g).pow(4).float()
    init = torch.rand(K, D, generator=g)
    batches = [torch.randint(0, N, (B,), generator=g) for _ in range(steps)]
    out = {}
    for fused in (True, False):
        vqm._FUSED_STEP = fused
        try:
            cb, idx, errs = hip.vq_features(f.cuda(), imp.cuda(), K, B, steps, batches=batches, init_rand=init, silent=True,
                                            scale_normalize=scale_normalize, return_errors=True)
        finally:
            vqm._FUSED_STEP = True
        out[fused] = (cb.cpu().numpy(), idx.cpu().numpy(), np.array(errs))
    # (the sums are float atomics: two runs of the SAME path differ by summation order too, so bitwise equality of whole loops is
    # not defined; the update's bit-identity on identical sums is asserted in test_fused_step_protocol_step_by_step)
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-5, atol=1e-7)
    assert (out[True][1] == out[False][1]).mean() >= 0.999
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-5)


@pytest.mark.parametrize("D,K,B,scale_normalize", [(48, 512, 4096, False), (6, 2048, 70_000, True), (12, 100, 999, False)])
def test_fused_step_protocol_step_by_step(hip, orc, D, K, B, scale_normalize):
    """c3dgs_vq_step_sums / c3dgs_vq_step_apply called directly, four steps: every step's distances and indices equal the
    oracle's on the codebook of that step bit for bit (the search ran on fragments the PREVIOUS update produced), the
    distance sum equals the float64 sum of the distances, and the update applied to the step's S equals c3dgs_vq_apply on a
    copy of the same S / codebook / entry importance bit for bit (the same single-rounded operations; compression/vq.py:31-35,
    73-77) while leaving S cleared."""
    import ctypes as C
    from c3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(19 + D)
    x = (torch.randn(B, D, generator=g) * 0.1).float()
    if scale_normalize:
        x[:, [0, 3, 5]] = x[:, [0, 3, 5]].abs() + 0.2
    x = x.cuda()
    w = (torch.rand(B, generator=g) + 0.1).float().cuda()
    cb = x[torch.randperm(B, generator=g)[:K].cuda()].clone().contiguous()
    ent = torch.zeros(K, device="cuda")
    dist, idx = torch.empty(B, device="cuda"), torch.empty(B, dtype=torch.int64, device="cuda")
    S = torch.empty(K, D + 1, device="cuda")
    dsum = torch.zeros(4, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(L.c3dgs_weighted_distance_ws_bytes(B, K, D)), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.c3dgs_vq_step_supported(K, D, x.data_ptr(), cb.data_ptr(), ws.data_ptr(), int(ws.numel())) == 1
    for step in range(4):
        cb_before = cb.clone()
        _lib.check(L.c3dgs_vq_step_sums(step, B, K, D, x.data_ptr(), w.data_ptr(), None, cb.data_ptr(), dist.data_ptr(), idx.data_ptr(),
                                        S.data_ptr(), dsum[step:step + 1].data_ptr(), ws.data_ptr(), int(ws.numel()), st))
        d_ref, i_ref = orc.weighted_distance(x.cpu().numpy(), cb_before.cpu().numpy())
        np.testing.assert_array_equal(idx.cpu().numpy(), i_ref)
        np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), d_ref.view(np.uint32))
        want = float(d_ref.astype(np.float64).sum())
        assert abs(float(dsum[step]) - want) <= 1e-9 * want
        S_ref = torch.zeros(K, D + 1, dtype=torch.float64, device="cuda")
        S_ref.index_add_(0, idx, torch.cat([x.double() * w.double()[:, None], w.double()[:, None]], 1))
        torch.testing.assert_close(S.double(), S_ref, rtol=2e-5, atol=1e-6)
        S2, cb2, ent2 = S.clone(), cb.clone(), ent.clone()
        _lib.check(L.c3dgs_vq_apply(K, D, S2.data_ptr(), cb2.data_ptr(), ent2.data_ptr(), 0.8, 0.2, 1e-5, int(scale_normalize), st))
        _lib.check(L.c3dgs_vq_step_apply(step, K, D, S.data_ptr(), cb.data_ptr(), ent.data_ptr(), 0.8, 0.2, 1e-5, int(scale_normalize),
                                         ws.data_ptr(), int(ws.numel()), st))
        np.testing.assert_array_equal(cb.cpu().numpy().view(np.uint32), cb2.cpu().numpy().view(np.uint32))
        np.testing.assert_array_equal(ent.cpu().numpy().view(np.uint32), ent2.cpu().numpy().view(np.uint32))
        assert not S.any()                                            # cleared behind its last reader


def test_fused_step_falls_back_to_the_exact_scan_when_the_codebook_collapses(hip, orc):
    """The fused step scales the next search's fp16 fragments with the exponent of the codebook it READ. A codebook of
    magnitude ~1e3 updated with decay 0 onto data of magnitude ~0.1 shrinks by 2^13 in one step -- far outside the window the
    split's error model covers -- so the second search must send every point through the exact re-scan: distances and indices
    still equal the oracle's bit for bit."""
    import ctypes as C
    from c3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    B, K, D = 4096, 512, 48
    x = (torch.randn(B, D, generator=g) * 0.1).float().cuda()
    w = (torch.rand(B, generator=g) + 0.1).float().cuda()
    cb = (torch.randn(K, D, generator=g) * 1000.0).float().cuda()
    ent = torch.zeros(K, device="cuda")
    dist, idx = torch.empty(B, device="cuda"), torch.empty(B, dtype=torch.int64, device="cuda")
    S = torch.empty(K, D + 1, device="cuda")
    dsum = torch.zeros(2, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(L.c3dgs_weighted_distance_ws_bytes(B, K, D)), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.c3dgs_vq_step_supported(K, D, x.data_ptr(), cb.data_ptr(), ws.data_ptr(), int(ws.numel())) == 1
    for step in range(2):
        cb_before = cb.cpu().numpy().copy()
        _lib.check(L.c3dgs_vq_step_sums(step, B, K, D, x.data_ptr(), w.data_ptr(), None, cb.data_ptr(), dist.data_ptr(), idx.data_ptr(),
                                        S.data_ptr(), dsum[step:step + 1].data_ptr(), ws.data_ptr(), int(ws.numel()), st))
        d_ref, i_ref = orc.weighted_distance(x.cpu().numpy(), cb_before)
        np.testing.assert_array_equal(idx.cpu().numpy(), i_ref)
        np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), d_ref.view(np.uint32))
        assert abs(float(dsum[step]) - float(d_ref.astype(np.float64).sum())) <= 1e-9 * float(d_ref.astype(np.float64).sum())
        _lib.check(L.c3dgs_vq_step_apply(step, K, D, S.data_ptr(), cb.data_ptr(), ent.data_ptr(), 0.0, 1.0, 1e-5, 0, ws.data_ptr(),
                                         int(ws.numel()), st))
        assert not S.any()                                            # cleared behind its last reader
    assert float(cb.abs().max()) < 10.0                            # the codebook did collapse onto the data's scale
    # unsupported shapes are refused, not mis-served
    assert L.c3dgs_vq_step_supported(K, 27, x.data_ptr(), cb.data_ptr(), ws.data_ptr(), int(ws.numel())) == 0
    assert L.c3dgs_vq_step_sums(0, B, K, 27, x.data_ptr(), w.data_ptr(), None, cb.data_ptr(), dist.data_ptr(), idx.data_ptr(), S.data_ptr(),
                                dsum.data_ptr(), ws.data_ptr(), int(ws.numel()), st) != 0
