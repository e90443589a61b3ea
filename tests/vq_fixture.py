"""Replay of the seed-only VQ fixtures (tests/golden/vq_d48.npz, vq_config0.npz; test infrastructure).

The reference's vq_features (compression/vq.py:49-87) consumes the CPU default generator in this order after
torch.manual_seed(seed):  VectorQuantize.__init__'s kaiming_uniform_ of the [K, D] codebook (:19-21)  ->  rand_like of
uniform_init (:26, drawn on the FEATURES' device: the CPU for the fixture, the GPU generator in a real run -- which is why the
fixture stores that draw)  ->  one `torch.randint(0, N, [chunk])` per step (:69).  Features and importance are regenerated from
their own seeded generator (checksums in the fixture guard against a generator change)."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(G, name), allow_pickle=False)
    N, D, K, steps, chunk = (int(d[k]) for k in ("N", "D", "K", "steps", "chunk"))
    g = torch.Generator().manual_seed(int(d["data_seed"]))
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    imp = torch.rand(N, generator=g).pow(4).float()
    assert abs(float(f.double().sum()) - float(d["features_checksum"])) < 1e-9, "torch's seeded generator changed: regenerate the fixture"
    assert abs(float(imp.double().sum()) - float(d["importance_checksum"])) < 1e-9
    return dict(N=N, D=D, K=K, steps=steps, chunk=chunk, seed=int(d["seed"]), features=f, importance=imp,
                init_rand=torch.from_numpy(d["init_rand"]), codebook=d["codebook"], indices=d["indices"].astype(np.int64))


def seed_like_reference(fx, model_built_inside=True):
    """Positions the CPU generator where a vq_features call must find it so that its randint batches are the reference's.
    A callee that constructs VectorQuantize itself (c3dgs_amd.vq.vq_features: its __init__ runs the same kaiming_uniform_,
    K*D draws) and receives init_rand explicitly needs ONE [K, D] uniform fill burnt in front (the reference's rand_like);
    with model_built_inside=False both fills are burnt here and the next draw is the first batch."""
    torch.manual_seed(fx["seed"])
    torch.rand(fx["K"], fx["D"])                             # same consumption as a [K, D] uniform_ fill
    if not model_built_inside:
        r = torch.rand(fx["K"], fx["D"])
        assert torch.equal(r, fx["init_rand"]), "generator stream does not line up with the fixture's uniform_init draw"


def batches(fx):
    """The reference's batches as data (for the oracle, which takes draws as arrays)."""
    seed_like_reference(fx, model_built_inside=False)
    return [torch.randint(low=0, high=fx["N"], size=[fx["chunk"]]) for _ in range(fx["steps"])]
