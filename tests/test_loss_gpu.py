"""-m gpu: fused L1 + SSIM loss (row N3) against the oracle and the committed reference-generated golden vectors.
Tolerances: scalar values 2e-6 absolute (fp32 separable window vs float64 direct sums), gradient rel-inf 1e-4."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_loss_matches_reference_golden(hip, tag):
    from c3dgs_amd import loss as L
    d = np.load(os.path.join(G, "loss.npz"), allow_pickle=False)
    img = torch.from_numpy(d[f"img_{tag}"]).cuda().requires_grad_()
    gt = torch.from_numpy(d[f"gt_{tag}"]).cuda()
    val = L.l1_ssim_loss(img, gt, 0.2)
    val.backward()
    assert abs(val.item() - float(d[f"loss_{tag}"])) < 2e-6
    ref = d[f"grad_{tag}"]
    assert np.abs(img.grad.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    assert abs(L.ssim(img.detach(), gt).item() - float(d[f"ssim_{tag}"])) < 2e-6
    assert abs(L.l1_loss(img.detach(), gt).item() - float(d[f"l1_{tag}"])) < 1e-6


@pytest.mark.parametrize("shape", [(3, 136, 200), (3, 131, 203), (3, 1080, 1920)])
def test_loss_matches_oracle(hip, orc, shape):
    from c3dgs_amd import loss as L
    g = torch.Generator().manual_seed(shape[1])
    gt = torch.rand(*shape, generator=g)
    img = (gt + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    x = img.cuda().requires_grad_()
    val = L.l1_ssim_loss(x, gt.cuda(), 0.2)
    (val * 3.0).backward()                      # non-unit upstream gradient
    lo, l1, ss, gr = orc.l1_ssim(img.numpy(), gt.numpy(), 0.2)
    assert abs(val.item() - lo) < 2e-6
    got = x.grad.cpu().numpy() / 3.0
    assert np.abs(got - gr).max() / np.abs(gr).max() < 1e-4


def test_ssim_is_differentiable_and_errors(hip):
    from c3dgs_amd import loss as L
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(3, 40, 40, generator=g).cuda()
    x = torch.rand(3, 40, 40, generator=g).cuda().requires_grad_()
    s = L.ssim(x, gt)
    s.backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    assert abs(L.ssim(gt, gt).item() - 1.0) < 1e-5
    with pytest.raises(RuntimeError, match="GPU"):
        L.l1_ssim_loss(x.cpu(), gt.cpu())
    with pytest.raises(RuntimeError, match="same shape"):
        L.l1_ssim_loss(x, gt[:, :20])
