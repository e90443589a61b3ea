"""PyTorch-CPU restatement of the reference's VQ loop -- TEST INFRASTRUCTURE / CPU BASELINE, NOT THE PRODUCT.

Only tests/ and bench.py's `cpu_baseline` legs import this module (the product package never does).

BASELINE.json's north star asks for "the reference's PyTorch-CPU VQ timed on the same box's host cores". The reference
file itself (compression/vq.py) cannot travel to the GPU box, and on CPU tensors it needs two native extensions that
have no CPU build (torch_scatter, weighted_distance._C), so this module restates

    VectorQuantize.update      compression/vq.py:28-35   (scatter == index_add_; ema_inplace :45-46)
    uniform_init               compression/vq.py:24-26
    vq_features                compression/vq.py:49-87   (torch.randint batches on the CPU generator, trace normalisation)

as torch CPU-tensor code and offers the nearest-codeword search in two forms:

    "direct"  chunked (x[:, None] - cb[None])**2 .sum(-1).min(1): the arithmetic of weighted_distance.cu:9-58 per pair --
              exactly the shim tests/golden/make_golden.py used when it ran the reference's own vq_features;
    "gemm"    ||x||^2 - 2 x.c^T + ||c||^2 argmin with the winner's distance recomputed directly: the form a CPU port would
              really use (one SGEMM per chunk), so the reported GPU/CPU ratio is not inflated by a slow stand-in.

Pinned in the build container: tests/test_oracle_golden.py shows both forms reproduce tests/golden/vq_color.npz and
vq_cov.npz (outputs of the reference's vq_features) from the captured RNG draws.
"""
import time

import torch


def weighted_distance_direct(x, cb, chunk=4096):
    """(min squared distance, argmin) by direct differences, chunked over points."""
    outd, outi = [], []
    for s in range(0, x.shape[0], chunk):
        d = ((x[s:s + chunk, None] - cb[None]) ** 2).sum(-1)
        m, i = d.min(1)
        outd.append(m)
        outi.append(i)
    return torch.cat(outd), torch.cat(outi)


def weighted_distance_gemm(x, cb, chunk=65536):
    """argmin of ||c||^2 - 2 x.c (one SGEMM per chunk); the winner's distance recomputed by direct differences."""
    cn = (cb * cb).sum(1)
    outd, outi = [], []
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        i = torch.addmm(cn[None], xs, cb.t(), beta=1.0, alpha=-2.0).argmin(1)
        outd.append(((xs - cb[i]) ** 2).sum(-1))
        outi.append(i)
    return torch.cat(outd), torch.cat(outi)


FORMS = {"direct": weighted_distance_direct, "gemm": weighted_distance_gemm}


class VectorQuantize:
    """compression/vq.py:15-42 on plain CPU tensors."""

    def __init__(self, channels, codebook_size, decay=0.8, form="direct"):
        self.decay, self.eps = decay, 1e-5
        self.codebook = torch.empty(codebook_size, channels)
        self.entry_importance = torch.zeros(codebook_size)
        self.search = FORMS[form]

    def uniform_init(self, x, rand=None):
        amin, amax = x.aminmax()
        r = torch.rand_like(self.codebook) if rand is None else rand
        self.codebook = r * (amax - amin) + amin

    def update(self, x, importance):
        K = self.codebook.shape[0]
        min_dists, idx = self.search(x, self.codebook)
        acc_importance = torch.zeros(K).index_add_(0, idx, importance)                    # scatter(importance, idx, sum)
        self.entry_importance.mul_(self.decay).add_(acc_importance, alpha=(1 - self.decay))
        codebook = torch.zeros(K, x.shape[1]).index_add_(0, idx, x * importance[:, None])
        self.codebook.mul_(self.decay).add_(codebook / (acc_importance[:, None] + self.eps), alpha=(1 - self.decay))
        return min_dists


def vq_features(features, importance, codebook_size, vq_chunk=2 ** 16, steps=1000, decay=0.8, scale_normalize=False,
                form="direct", init_rand=None, batches=None, final_assignment=True, stats=None):
    """compression/vq.py:49-87. `init_rand` / `batches` replace the two RNG draws with captured data (parity tests);
    `stats` (dict) receives the wall time of the Lloyd loop and of the final assignment."""
    importance_n = importance / importance.max()
    model = VectorQuantize(features.shape[-1], codebook_size, decay, form)
    model.uniform_init(features, init_rand)
    errors = []
    t0 = time.perf_counter()
    for s in range(steps if batches is None else len(batches)):
        batch = torch.randint(low=0, high=features.shape[0], size=[vq_chunk]) if batches is None else batches[s]
        error = model.update(features[batch], importance_n[batch]).mean().item()
        errors.append(error)
        if scale_normalize:
            tr = model.codebook[:, [0, 3, 5]].sum(-1)
            model.codebook /= tr[:, None]
    t1 = time.perf_counter()
    idx = model.search(features, model.codebook)[1] if final_assignment else None
    if stats is not None:
        stats.update(lloyd_seconds=t1 - t0, lloyd_steps=len(errors), final_assignment_seconds=time.perf_counter() - t1)
    return model.codebook, idx, errors


def time_lloyd_steps(N, D, K, batch, form, sample_points, steps, seed=7, threads=None):
    """Bounded timing of Lloyd steps of the given shape on this host: `steps` updates on batches of `sample_points` points
    (<= batch); the per-step time is scaled by batch / sample_points (the search is linear in the number of points; the
    K x D update is not scaled). -> dict(seconds_per_step, measured_seconds, ...)."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(max(N, sample_points), D, generator=g) * 0.1
    imp = torch.rand(feats.shape[0], generator=g).pow(4)
    imp_n = imp / imp.max()
    model = VectorQuantize(D, K, 0.8, form)
    model.uniform_init(feats, torch.rand(K, D, generator=g))
    draws = [torch.randint(0, feats.shape[0], (sample_points,), generator=g) for _ in range(steps + 1)]
    model.update(feats[draws[0]], imp_n[draws[0]])                                       # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for b in draws[1:]:
        model.update(feats[b], imp_n[b]).mean().item()
    el = time.perf_counter() - t0
    return dict(seconds_per_step=el / steps * (batch / sample_points), measured_seconds=el, steps=steps,
                sample_points=sample_points, scale=batch / sample_points, threads=torch.get_num_threads())
