"""CPU, world_size 2, gloo: the sharded VQ control flow of c3dgs_amd.vq.vq_features (same batch on every rank,
contiguous slices, one all-reduce of S[K,D+1] + distance sum per Lloyd step, identical EMA update, sharded final
assignment + all_gather).  Compute is injected from the oracle (tests/oracle_ops.py) because the product has no
CPU path; what is under test is the distribution logic, which is device-independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(scale_normalize):
    g = torch.Generator().manual_seed(0)
    N, D, K, steps, chunk = 3001, (6 if scale_normalize else 12), 32, 6, 501     # 3001 points, 501 per batch: ragged splits across 2 ranks
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    if scale_normalize:
        f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, generator=g).pow(4).float()
    init = torch.rand(K, D, generator=g)
    batches = [torch.randint(0, N, (chunk,), generator=g) for _ in range(steps)]
    return f, imp, K, chunk, steps, init, batches


def _worker(rank, world, port, scale_normalize, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from c3dgs_amd import vq
    from tests.oracle_ops import OracleOps
    f, imp, K, chunk, steps, init, batches = _problem(scale_normalize)
    cb, idx, errs = vq.vq_features(f, imp, K, chunk, steps, scale_normalize=scale_normalize, silent=True, group=True,
                                   batches=batches, init_rand=init, return_errors=True, ops=OracleOps)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cb=cb.numpy(), idx=idx.numpy(), errs=np.array(errs))
    # RNG-driven variant: every rank continues rank 0's generator stream (its state is broadcast ONCE), so ranks must still
    # agree bit for bit although they were seeded differently; and the collectives are counted: per Lloyd step exactly one
    # all-reduce (of S) and no broadcast
    torch.manual_seed(100 + rank)
    counts = {"all_reduce": 0, "broadcast": 0, "all_gather_into_tensor": 0}
    orig = {k: getattr(dist, k) for k in counts}

    def counting(name):
        def f_(*a, **k):
            counts[name] += 1
            return orig[name](*a, **k)
        return f_
    for k in counts:
        setattr(dist, k, counting(k))
    try:
        cb2, idx2 = vq.vq_features(f, imp, K, chunk, 3, scale_normalize=scale_normalize, silent=True, group=True, ops=OracleOps)
    finally:
        for k in counts:
            setattr(dist, k, orig[k])
    after = torch.randint(0, 1 << 30, (4,))                    # all ranks were left at the same generator state
    np.savez(os.path.join(out_dir, f"rng_rank{rank}.npz"), cb=cb2.numpy(), idx=idx2.numpy(), after=after.numpy(),
             counts=np.array([counts["all_reduce"], counts["broadcast"], counts["all_gather_into_tensor"]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("scale_normalize", [False, True])
def test_sharded_vq_matches_single_rank(tmp_path, scale_normalize):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), scale_normalize, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["cb"].view(np.uint32), r1["cb"].view(np.uint32))      # ranks bit-identical
    np.testing.assert_array_equal(r0["idx"], r1["idx"])
    q0, q1 = np.load(tmp_path / "rng_rank0.npz"), np.load(tmp_path / "rng_rank1.npz")
    np.testing.assert_array_equal(q0["cb"].view(np.uint32), q1["cb"].view(np.uint32))
    np.testing.assert_array_equal(q0["idx"], q1["idx"])
    np.testing.assert_array_equal(q0["after"], q1["after"])
    # 3 Lloyd steps: 3 all-reduces (one per step), 2 broadcasts in total (uniform_init's draw + the 5 KB generator state), one
    # all_gather for the sharded final assignment
    assert q0["counts"].tolist() == [3, 2, 1] and q1["counts"].tolist() == [3, 2, 1]
    # single rank, same draws
    from c3dgs_amd import vq
    from tests.oracle_ops import OracleOps
    f, imp, K, chunk, steps, init, batches = _problem(scale_normalize)
    cb, idx, errs = vq.vq_features(f, imp, K, chunk, steps, scale_normalize=scale_normalize, silent=True, batches=batches,
                                   init_rand=init, return_errors=True, ops=OracleOps)
    np.testing.assert_allclose(r0["cb"], cb.numpy(), rtol=1e-5, atol=1e-7)
    assert (r0["idx"] == idx.numpy()).mean() >= 0.999
    np.testing.assert_allclose(r0["errs"], np.array(errs), rtol=1e-6)
    assert r0["idx"].shape == (3001,)


def test_oracle_ops_equal_oracle_update(orc):
    """The injected CPU ops are the same arithmetic as the oracle's vq_update (so the test above tests distribution)."""
    from c3dgs_amd import vq
    from tests.oracle_ops import OracleOps
    f, imp, K, chunk, steps, init, batches = _problem(False)
    cb, idx, errs = vq.vq_features(f, imp, K, chunk, steps, silent=True, batches=batches, init_rand=init, return_errors=True,
                                   ops=OracleOps)
    cb_ref, idx_ref, err_ref, _ = orc.vq_features(f.numpy(), imp.numpy(), K, init.numpy(), [b.numpy() for b in batches])
    np.testing.assert_allclose(cb.numpy(), cb_ref, rtol=1e-5, atol=1e-7)
    assert (idx.numpy() == idx_ref).mean() >= 0.999
    np.testing.assert_allclose(np.array(errs), err_ref, rtol=1e-5)
