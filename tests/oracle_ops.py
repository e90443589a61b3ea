"""TEST INFRASTRUCTURE: a CPU stand-in for c3dgs_amd.vq.HipOps backed by the oracle, so that the sharding /
all-reduce control flow of vq_features can be exercised with the gloo backend on machines without a GPU.
Never used by the product package."""
import numpy as np
import torch

from oracle import oracle as orc


class OracleOps:
    @staticmethod
    def assign(x, codebook, gather=None):
        xs = x if gather is None else x[gather]
        d, i = orc.weighted_distance(xs.detach().numpy(), codebook.detach().numpy())
        return torch.from_numpy(d), torch.from_numpy(i)

    @staticmethod
    def accumulate(x, importance, gather, idx, min_dists, K):
        xs = x if gather is None else x[gather]
        w = importance if gather is None else importance[gather]
        D = xs.shape[1]
        S = torch.zeros(K, D + 1, dtype=torch.float64)
        S[:, :D].index_add_(0, idx, (xs * w[:, None]).double())
        S[:, D].index_add_(0, idx, w.double())
        return S.float(), min_dists.double().sum().reshape(1)

    @staticmethod
    def apply(S, codebook, entry_importance, decay, eps, scale_normalize):
        K, D = codebook.shape
        aw = S[:, D]
        d, a, e = np.float32(decay), np.float32(1.0 - decay), np.float32(eps)
        entry_importance.copy_(entry_importance * d + a * aw)
        codebook.copy_(codebook * d + a * (S[:, :D] / (aw[:, None] + e)))
        if scale_normalize:
            codebook.div_((codebook[:, 0] + codebook[:, 3] + codebook[:, 5])[:, None])
