"""-m gpu: fused Adam (csrc/adam.hip, c3dgs_amd/optim.py) against torch.optim.Adam on the same device, the optimizer
the reference builds for the QAT loop (scene/gaussian_model.py:296-308: param groups with their own lr, eps=1e-15)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(tensors, lrs):
    return [{"params": [t], "lr": lr} for t, lr in zip(tensors, lrs)]


def test_fused_adam_follows_torch_adam_over_many_steps():
    from c3dgs_amd.optim import Adam
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(100_003, 3), (5000, 1, 3), (5000, 15, 3), (100_003, 1), (7001, 3), (7001, 4), (100_003, 1), (17,)]
    lrs = [0.00016, 0.0025, 0.0025 / 20, 0.05, 0.005, 0.001, 0.005, 0.01]
    a = [torch.randn(s, device="cuda", generator=g).requires_grad_() for s in shapes]
    b = [t.detach().clone().requires_grad_() for t in a]
    ours = Adam(_groups(a, lrs), lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(_groups(b, lrs), lr=0.0, eps=1e-15)
    for step in range(25):
        for x, y in zip(a, b):
            grad = torch.randn(x.shape, device="cuda", generator=g) * (10.0 ** ((step % 5) - 3))
            if step == 3:
                grad[::2] = 0                                      # v stays tiny: exercises eps = 1e-15
            x.grad, y.grad = grad.clone(), grad.clone()
        ours.step()
        ref.step()
    for x, y, sh in zip(a, b, shapes):
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-7), (sh, float((x - y).abs().max()))
        so, sr = ours.state[x], ref.state[y]
        assert float(so["step"]) == float(sr["step"]) == 25.0
        # the moments cancel through zero: compare against their scale (1-ulp differences of the lerp / fma contraction)
        assert float((so["exp_avg"] - sr["exp_avg"]).abs().max()) <= 2e-6 * float(sr["exp_avg"].abs().max())
        assert float((so["exp_avg_sq"] - sr["exp_avg_sq"]).abs().max()) <= 2e-6 * float(sr["exp_avg_sq"].abs().max())
    # state dicts are interchangeable
    ref2 = torch.optim.Adam(_groups([t.detach().clone().requires_grad_() for t in a], lrs), lr=0.0, eps=1e-15)
    ref2.load_state_dict(ours.state_dict())


def test_fused_adam_skips_parameters_without_grad_and_rejects_cpu():
    from c3dgs_amd.optim import Adam
    p, q = torch.ones(10, device="cuda", requires_grad=True), torch.ones(10, device="cuda", requires_grad=True)
    opt = Adam([p, q], lr=0.1)
    p.grad = torch.ones(10, device="cuda")
    opt.step()
    assert torch.equal(q, torch.ones(10, device="cuda")) and float(p[0]) == pytest.approx(0.9, rel=1e-5)
    c = torch.ones(4, requires_grad=True)
    c.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        Adam([c], lr=0.1).step()
    with pytest.raises(RuntimeError, match="plain Adam"):
        Adam([p], weight_decay=0.1)
