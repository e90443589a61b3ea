"""CPU, gloo world_size 2: the camera-sharded accumulation of c3dgs_amd.sensitivity.calc_importance equals the
single-process sum over all cameras. The renderer is injected (a small differentiable torch function): what is under
test is the sharding / all-reduce logic, which is device independent; the real renderer is covered by the -m gpu test."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _problem():
    g = torch.Generator().manual_seed(0)
    P = 50
    dc = torch.randn(P, 1, 3, generator=g).requires_grad_()
    rest = torch.randn(P, 3, 3, generator=g).requires_grad_()
    cov = torch.rand(P, 6, generator=g).requires_grad_()
    cams = [SimpleNamespace(k=float(k + 1), original_image=torch.rand(3, 8, 8, generator=g)) for k in range(5)]

    def render(cam):                      # any differentiable function of the three leaves
        a = (dc.sum(1) * cam.k).sum(0)[:, None, None]
        b = (rest.sum(1).tanh() * cov[:, :3]).sum(0)[:, None, None]
        return (a + b).expand(3, 8, 8) * torch.linspace(0.5, 1.5, 64).reshape(1, 8, 8)
    return dc, rest, cov, cams, render


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from c3dgs_amd import sensitivity
    dc, rest, cov, cams, render = _problem()
    imp, cg = sensitivity.calc_importance(render, dc, rest, cov, cams, use_gt=True, group=True,
                                          loss_fn=lambda im, gt: (im - gt).abs().mean())
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), imp=imp.detach().numpy(), cg=cg.detach().numpy())
    dist.destroy_process_group()


def test_camera_sharding_matches_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from c3dgs_amd import sensitivity
    dc, rest, cov, cams, render = _problem()
    imp, cg = sensitivity.calc_importance(render, dc, rest, cov, cams, use_gt=True,
                                          loss_fn=lambda im, gt: (im - gt).abs().mean())
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(r0["imp"], r1["imp"])
    np.testing.assert_allclose(r0["imp"], imp.detach().numpy(), rtol=1e-6)
    np.testing.assert_allclose(r0["cg"], cg.detach().numpy(), rtol=1e-6)
    assert imp.shape == (50, 12) and cg.shape == (50, 6)
    # image.sum() variant (compress.py:103)
    imp2, _ = sensitivity.calc_importance(render, dc, rest, cov, cams, use_gt=False)
    assert torch.isfinite(imp2).all() and (imp2 >= 0).all()
