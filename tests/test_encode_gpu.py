"""-m gpu: Morton ordering (row N4): integer work, bit-exact against the oracle and the reference-generated vector."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_morton_matches_reference_golden(hip):
    from c3dgs_amd import encode
    d = np.load(os.path.join(G, "morton.npz"), allow_pickle=False)
    xyz = torch.from_numpy(d["xyz"]).cuda()
    np.testing.assert_array_equal(encode.morton_codes(xyz).cpu().numpy(), d["codes"])


@pytest.mark.parametrize("P", [1, 777, 200_000, 3_000_000])
def test_morton_order_matches_oracle(hip, orc, P):
    from c3dgs_amd import encode
    g = torch.Generator().manual_seed(P)
    xyz = (torch.randn(P, 3, generator=g) * torch.tensor([0.5, 4.0, 1.5])).float()
    if P > 1000:
        xyz[100:200] = xyz[0]                      # ties: stable order keeps ascending ids
    codes_ref, _ = orc.morton_codes(xyz.numpy())
    order_ref = np.argsort(codes_ref, kind="stable")
    xc = xyz.cuda()
    np.testing.assert_array_equal(encode.morton_codes(xc).cpu().numpy(), codes_ref)
    np.testing.assert_array_equal(encode.morton_order(xc).cpu().numpy(), order_ref)


def test_morton_errors(hip):
    from c3dgs_amd import encode
    with pytest.raises(RuntimeError, match="GPU"):
        encode.morton_order(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        encode.morton_order(torch.zeros(4, 2).cuda())
    assert encode.morton_order(torch.zeros(0, 3).cuda()).shape == (0,)
