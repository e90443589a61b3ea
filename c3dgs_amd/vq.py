"""Host-side mirror of the reference's VQ path on top of the MI355X C-ABI library:

    weightedDistance          <- weighted_distance._C.weightedDistance (submodules/weighted_distance/ext.cpp:5)
    VectorQuantize, ema_inplace, vq_features, join_features, CompressionSettings,
    compress_color, compress_covariance, compress_gaussians   <- compression/vq.py:15-223

Same names, argument meaning and error behaviour.  Additions (new functionality, BASELINE.json north_star):
  * `vq_features(..., group=...)`: every rank of a torch.distributed group draws the SAME batch, assigns its
    contiguous slice, and the per-cluster partial sums S[K, D+1] (sum w*x | sum w) plus the distance sum are
    all-reduced in place (RCCL over xGMI when the backend is "nccl") once per Lloyd step -- the only collective of a
    step: the batch draws follow rank 0's CPU generator state, broadcast once (5 KB), and the per-step distance sums
    are reduced once after the loop; every rank applies the identical EMA update, so codebooks stay bit-identical
    across ranks.
  * `batches=` / `init_rand=`: the two RNG draws can be passed in as data (parity tests, sharded runs).
There is no CPU path: tensors must live on the GPU and libc3dgs_hip.so must be present.
"""
import ctypes as C
import concurrent.futures as _futures
import time
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from . import _lib


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def weightedDistance(coefs: torch.Tensor, codebook: torch.Tensor, gather: Optional[torch.Tensor] = None):
    """(min squared distance f32[N], argmin i64[N]); reference weighted_distance.cu:60-93.
    `gather` (int64[N], optional) makes row n of the query `coefs[gather[n]]` without materialising it."""
    if coefs.dim() != 2 or codebook.dim() != 2:
        raise RuntimeError("ceofs and codebook must have dimension 2")          # weighted_distance.cu:64-67
    if codebook.size(1) != coefs.size(1):
        raise RuntimeError("coefs and codebook must have same number of channels")  # :69-72
    if not coefs.is_cuda or not codebook.is_cuda:
        raise RuntimeError("c3dgs_amd: weightedDistance needs GPU tensors (there is no CPU path)")
    L = _lib.lib()
    x = coefs.detach().contiguous().float()
    cb = codebook.detach().contiguous().float()
    if gather is not None:
        gather = gather.to(device=x.device, dtype=torch.int64).contiguous()
    N = int(gather.numel()) if gather is not None else int(x.size(0))
    dist = torch.zeros(N, dtype=torch.float32, device=x.device)
    idx = torch.zeros(N, dtype=torch.int64, device=x.device)
    if N > 0 and cb.size(0) > 0:
        with torch.cuda.device(x.device):
            # scratch: the codebook split into two fp16 pieces for the matrix cores + the list of ambiguous points (~1 %)
            ws = torch.empty(int(L.c3dgs_weighted_distance_ws_bytes(N, int(cb.size(0)), int(x.size(1)))), dtype=torch.uint8,
                             device=x.device)
            rc = L.c3dgs_weighted_distance_ws(N, int(cb.size(0)), int(x.size(1)), x.data_ptr(),
                                              gather.data_ptr() if gather is not None else None, cb.data_ptr(),
                                              dist.data_ptr(), idx.data_ptr(), ws.data_ptr(), int(ws.numel()),
                                              _stream(x.device))
        _lib.check(rc)
    return dist, idx


def ema_inplace(moving_avg: torch.Tensor, new: torch.Tensor, decay: float):
    """compression/vq.py:45-46."""
    moving_avg.data.mul_(decay).add_(new, alpha=(1 - decay))


class HipOps:
    """The three device operations of the VQ path, bound to libc3dgs_hip.so. (Tests of the sharding control
    flow may inject another object with the same three methods; the package itself only ever uses this one.)"""

    @staticmethod
    def assign(x, codebook, gather=None):
        return weightedDistance(x, codebook, gather)

    @staticmethod
    def accumulate(x, importance, gather, idx, min_dists, K):
        """S f32[K, D+1] = [sum w*x | sum w] over the rows selected by `gather`, and sum(min_dists) as f64[1]."""
        L = _lib.lib()
        D = int(x.size(1))
        S = torch.zeros(K, D + 1, dtype=torch.float32, device=x.device)
        dsum = torch.zeros(1, dtype=torch.float64, device=x.device)
        B = int(idx.numel())
        if B > 0:
            xw = x.detach().contiguous().float()
            w = importance.detach().contiguous().float()
            with torch.cuda.device(x.device):
                rc = L.c3dgs_vq_accumulate(B, K, D, xw.data_ptr(), w.data_ptr(),
                                           gather.data_ptr() if gather is not None else None, idx.data_ptr(),
                                           min_dists.data_ptr(), S.data_ptr(), dsum.data_ptr(), _stream(x.device))
            _lib.check(rc)
        return S, dsum

    @staticmethod
    def sums(x, importance, gather, codebook, scratch=None, dsum_out=None):
        """assign + accumulate as one library call (no intermediate tensors, no zero-fill launches). `scratch`: a dict
        that keeps the per-batch buffers (distances, indices, S) alive across steps of one loop; None allocates fresh.
        `dsum_out`: an f64[1] device view the distance sum is written to (e.g. one slot of a per-step array).
        -> (min_dists f32[B], S f32[K, D+1], dist_sum f64[1])."""
        L = _lib.lib()
        K, D = int(codebook.shape[0]), int(x.size(1))
        B = int(gather.numel()) if gather is not None else int(x.size(0))
        dev = x.device
        key = (B, K, D, dev)
        bufs = scratch.get(key) if scratch is not None else None
        if bufs is None:
            bufs = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.int64, device=dev),
                    torch.empty(K, D + 1, dtype=torch.float32, device=dev),
                    # the assignment's scratch: split codebook + list of ambiguous points (typically ~1 %)
                    torch.empty(int(L.c3dgs_weighted_distance_ws_bytes(B, K, D)), dtype=torch.uint8, device=dev))
            if scratch is not None:
                scratch.clear()
                scratch[key] = bufs
        dist, idx, S, ws = bufs
        dsum = torch.empty(1, dtype=torch.float64, device=dev) if dsum_out is None else dsum_out
        xw = x.detach().contiguous().float()
        w = importance.detach().contiguous().float()
        cb = codebook.detach().contiguous().float()
        if gather is not None:
            gather = gather.to(device=dev, dtype=torch.int64).contiguous()
        with torch.cuda.device(dev):
            rc = L.c3dgs_vq_sums(B, K, D, xw.data_ptr(), w.data_ptr(), gather.data_ptr() if gather is not None else None,
                                 cb.data_ptr(), dist.data_ptr(), idx.data_ptr(), S.data_ptr(), dsum.data_ptr(),
                                 ws.data_ptr(), int(ws.numel()), _stream(dev))
        _lib.check(rc)
        return dist, S, dsum

    @staticmethod
    def step_sums(state, x, importance, gather, codebook, dsum_out):
        """First half of a Lloyd step of vq_features' own loop (c3dgs_vq_step_sums): like `sums`, but from the second step on
        the search finds the codebook already split by the previous `step_apply`, S and the list counter already cleared and
        the distance sum folded into the accumulation -- 5 launches per step instead of 11. `state`: dict owned by the loop
        (buffers + the step counter of the protocol). Returns None when the shape is not served (caller uses `sums`)."""
        L = _lib.lib()
        K, D = int(codebook.shape[0]), int(x.size(1))
        B = int(gather.numel()) if gather is not None else int(x.size(0))
        dev = x.device
        key = (B, K, D, dev, codebook.data_ptr())
        if state.get("key") != key:
            state.clear()
            state.update(key=key, step=0,
                         bufs=(torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.int64, device=dev),
                               torch.empty(K, D + 1, dtype=torch.float32, device=dev),
                               torch.empty(int(L.c3dgs_weighted_distance_ws_bytes(B, K, D)), dtype=torch.uint8, device=dev)))
            dist, idx, S, ws = state["bufs"]
            state["ok"] = bool(B > 0 and x.dtype == torch.float32 and x.is_contiguous() and codebook.is_contiguous()
                               and L.c3dgs_vq_step_supported(K, D, x.data_ptr(), codebook.data_ptr(), ws.data_ptr(), int(ws.numel())))
        if not state["ok"]:
            return None
        dist, idx, S, ws = state["bufs"]
        w = importance
        with torch.cuda.device(dev):
            rc = L.c3dgs_vq_step_sums(state["step"], B, K, D, x.data_ptr(), w.data_ptr(), gather.data_ptr() if gather is not None else None,
                                      codebook.data_ptr(), dist.data_ptr(), idx.data_ptr(), S.data_ptr(), dsum_out.data_ptr(),
                                      ws.data_ptr(), int(ws.numel()), _stream(dev))
        _lib.check(rc)
        return dist, S, dsum_out

    @staticmethod
    def step_apply(state, codebook, entry_importance, decay, eps, scale_normalize):
        """Second half (c3dgs_vq_step_apply): the EMA update of `apply` + the next step's split codebook / cleared sums."""
        L = _lib.lib()
        K, D = codebook.shape
        _, _, S, ws = state["bufs"]
        with torch.cuda.device(codebook.device):
            rc = L.c3dgs_vq_step_apply(state["step"], K, D, S.data_ptr(), codebook.data_ptr(), entry_importance.data_ptr(), float(decay),
                                       float(1 - decay), float(eps), int(bool(scale_normalize)), ws.data_ptr(), int(ws.numel()),
                                       _stream(codebook.device))
        _lib.check(rc)
        state["step"] += 1

    @staticmethod
    def apply(S, codebook, entry_importance, decay, eps, scale_normalize):
        L = _lib.lib()
        K, D = codebook.shape
        assert codebook.is_contiguous() and entry_importance.is_contiguous() and codebook.dtype == torch.float32
        with torch.cuda.device(codebook.device):
            rc = L.c3dgs_vq_apply(K, D, S.data_ptr(), codebook.data_ptr(), entry_importance.data_ptr(), float(decay),
                                  float(1 - decay), float(eps), int(bool(scale_normalize)), _stream(codebook.device))
        _lib.check(rc)


class VectorQuantize(nn.Module):
    """compression/vq.py:15-42."""

    def __init__(self, channels: int, codebook_size: int = 2 ** 12, decay: float = 0.5, ops=None) -> None:
        super().__init__()
        self.decay = decay
        self.codebook = nn.Parameter(torch.empty(codebook_size, channels), requires_grad=False)
        nn.init.kaiming_uniform_(self.codebook)
        self.entry_importance = nn.Parameter(torch.zeros(codebook_size), requires_grad=False)
        self.eps = 1e-5
        self.ops = HipOps if ops is None else ops

    def uniform_init(self, x: torch.Tensor, rand: Optional[torch.Tensor] = None):
        amin, amax = x.aminmax()
        r = torch.rand_like(self.codebook) if rand is None else rand.to(self.codebook)
        self.codebook.data = r * (amax - amin) + amin

    def partial_sums(self, x: torch.Tensor, importance: torch.Tensor, gather: Optional[torch.Tensor] = None, scratch=None,
                     dsum_out=None):
        """Assignment + weighted scatter-sums of one (slice of a) batch.
        Returns (min_dists f32[B], S f32[K, D+1] = [sum w*x | sum w], dist_sum f64[1]). With `scratch` (a dict owned by the
        calling loop) min_dists and S are reused from step to step; `dsum_out` receives the distance sum in place."""
        if hasattr(self.ops, "sums") and x.is_cuda:
            return self.ops.sums(x, importance, gather, self.codebook.data, scratch, dsum_out)
        min_dists, idx = self.ops.assign(x, self.codebook.data, gather)
        S, dsum = self.ops.accumulate(x, importance, gather, idx, min_dists, int(self.codebook.shape[0]))
        if dsum_out is not None:
            dsum_out.copy_(dsum)
            dsum = dsum_out
        return min_dists, S, dsum

    def apply_sums(self, S: torch.Tensor, scale_normalize: bool = False):
        """EMA update from (all-reduced) sums; vq.py:32,34 (+ :73-77 when scale_normalize)."""
        self.ops.apply(S, self.codebook.data, self.entry_importance.data, self.decay, self.eps, scale_normalize)

    def update(self, x: torch.Tensor, importance: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            min_dists, S, _ = self.partial_sums(x, importance)
            self.apply_sums(S)
            return min_dists

    def forward(self, x: torch.Tensor, return_dists: bool = False):
        min_dists, idx = self.ops.assign(x.detach(), self.codebook.detach())
        if return_dists:
            return self.codebook[idx], idx, min_dists
        return self.codebook[idx], idx


# torch.get_rng_state() of the CPU generator: at::CPUGeneratorImplStateLegacy = { u64 seed; i32 left; i32 seeded; u64 next;
# u64 state[624]; normal-distribution cache } + the float-normal cache -- 5056 bytes. _MT_LEFT indexes the int32 view,
# _MT_NEXT / _MT_STATE the int64 view.
_MT_WORDS, _MT_LEFT, _MT_NEXT, _MT_STATE, _MT_BYTES = 624, 2, 2, 3, 5056
_FAST_DRAWS = True      # False: every batch through torch.randint itself (what the tests compare the fast path with)
_ZERO_COPY_DRAWS = True # False: the raw draws are copied to the device (c3dgs_draws_upload) instead of read in place by the kernel
_FUSED_STEP = True      # False: every Lloyd step through the plain sums / apply pair (what the tests compare the fused step with)


_RING_CACHE = {}            # (device, batch size) -> list of idle pinned rings
_FILL_POOL = None


def _fill_pool():
    """One worker thread per process for the draws of the next batch (created on first use)."""
    global _FILL_POOL
    if _FILL_POOL is None:
        _FILL_POOL = _futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="c3dgs-draws")
    return _FILL_POOL


class _BatchDraws:
    """The batch indices of one Lloyd step: exactly the reference's draws, `torch.randint(0, N, [chunk])` on the CPU
    default generator (vq.py:69), i.e. `mt19937() % N` per element. torch's scalar path costs ~2 ns per draw (2 ms for a
    2^20 batch -- more than the step's kernels), so the generator's stream is continued by the library's block-wise
    MT19937 (csrc/draws.hip) into a ring of pinned buffers as raw 32-bit words and reduced `% N` on the GPU by a kernel that
    reads the words straight from the page-locked host buffer (mapped for the device: no copy operation in the stream; a
    rank of a sharded run converts only its slice of the batch). The advanced state is written back to the torch generator by finish(), so later torch draws are
    unchanged as well. Falls back to torch.randint itself when the state layout is not the expected one or N >= 2^28 (from there on this
    torch consumes two outputs per element).
    device_rng=True draws on the GPU instead: no host work, different numbers."""

    RING = 4

    def __init__(self, N, chunk, steps, device, device_rng=False):
        device = torch.device(device)
        self.N, self.chunk, self.device, self.device_rng = N, chunk, device, device_rng
        self.key = None
        self._dev = None
        st = torch.get_rng_state()
        if not _FAST_DRAWS or device_rng or device.type != "cuda" or not (0 < N < 2 ** 28) or st.numel() != _MT_BYTES or chunk <= 0:
            return
        w = st.view(torch.int64)
        left, nxt = int(st.view(torch.int32)[_MT_LEFT]), int(w[_MT_NEXT])
        if not (1 <= left <= _MT_WORDS + 1 and 0 <= nxt <= _MT_WORDS and (left == 1 or nxt + left - 1 == _MT_WORDS)):
            return
        self._st = st
        self.key = w[_MT_STATE:_MT_STATE + _MT_WORDS].to(torch.int32).contiguous()      # the words are < 2^32: keep the low halves
        self.left, self.next = C.c_int64(left), C.c_int64(nxt)
        # the pinned ring is kept per (device, batch size) across calls: allocating (and later freeing) page-locked memory per
        # call showed up as sporadic 50-80 ms stalls inside the Lloyd loop of a LATER call; a ring is checked out here and
        # handed back by finish(), so two overlapping users never share one
        self._ring_key = (str(device), int(chunk))
        pool = _RING_CACHE.setdefault(self._ring_key, [])
        self.ring = pool.pop() if pool else [(torch.empty(chunk, dtype=torch.int32).pin_memory(), torch.cuda.Event())
                                             for _ in range(self.RING)]
        self.k = 0
        # The fill of batch k + 1 (0.26 ms for 2^20 draws, sequential by nature) runs on ONE worker thread while the caller
        # queues the kernels of batch k: ctypes releases the GIL for the call, so the two really overlap (a covariance Lloyd
        # step is host-bound otherwise: 0.79 -> 0.63 ms). The draws are the same stream in the same order; nothing is drawn
        # beyond `steps` batches and a batch drawn ahead of a loop that stops early is taken back by finish(), so the state
        # written back is exactly the reference's.
        self.steps = int(steps)
        self._pool = _fill_pool() if self.steps > 1 else None
        self._pending = None
        self._dev = None
        self._mapped = {}

    def _fill(self, k, ahead=False):
        host, ev = self.ring[k % self.RING]
        if k >= self.RING:
            ev.synchronize()                             # the copy out of this buffer four steps ago
        if ahead:                                        # a batch nobody may ask for (the loop can stop early): finish() takes it back
            self._snap = (self.key.clone(), self.left.value, self.next.value)
        _lib.check(_lib.lib().c3dgs_mt19937_fill(self.key.data_ptr(), C.byref(self.left), C.byref(self.next), host.data_ptr(), self.chunk))
        return host, ev

    def next_batch(self, lo=0, hi=None):
        """The next batch's indices [lo, hi) (default: all of them). A rank of a sharded run asks for ITS slice only: the whole
        batch is drawn (every rank follows the same stream) but only the slice's raw words cross to the device -- 128 KB
        instead of 1 MB for one of eight ranks of a 2^18-point batch, whose copy the step's first kernel used to wait for.
        The returned tensor is reused by the next call (consume it in stream order)."""
        hi = self.chunk if hi is None else hi
        if self.device_rng:
            return torch.randint(low=0, high=self.N, size=[self.chunk], device=self.device)[lo:hi]
        if self.key is None:
            return torch.randint(low=0, high=self.N, size=[self.chunk])[lo:hi].to(self.device)
        k = self.k
        self.k += 1
        host, ev = self._pending.result() if self._pending is not None else self._fill(k)
        self._pending = None
        if self._dev is None:            # device-side staging, once per loop: raw words + indices (reused every step, stream-ordered)
            self._dev = (torch.empty(self.chunk, dtype=torch.int32, device=self.device),
                         torch.empty(self.chunk, dtype=torch.int64, device=self.device),
                         torch.cuda.current_stream(self.device).cuda_stream)
        raw, out, stream = self._dev
        L = _lib.lib()
        if self._pool is not None and k + 1 < self.steps:
            # batch k + 1 is filled (ring buffer (k + 1) % RING: another one) while this step's kernels run; submitted BEFORE this
            # step's launches so that the hand-over to the worker thread does not sit between the conversion kernel and the search
            self._pending = self._pool.submit(self._fill, k + 1, True)
        with torch.cuda.device(self.device):
            mapped = self._mapped.get(host.data_ptr())
            if mapped is None:           # is the pinned buffer mapped for the device? (once per ring buffer)
                mapped = self._mapped[host.data_ptr()] = (L.c3dgs_host_buffer_device_address(host.data_ptr()) or 0) if _ZERO_COPY_DRAWS else 0
            if mapped:                   # the conversion kernel reads the words from host memory itself: no copy in the stream
                _lib.check(L.c3dgs_draws_to_indices(hi - lo, self.N, mapped + 4 * lo, out.data_ptr(), stream))
            else:
                _lib.check(L.c3dgs_draws_upload(hi - lo, self.N, host.data_ptr() + 4 * lo, raw.data_ptr(), out.data_ptr(), stream))
            ev.record()
        return out[:hi - lo]

    def finish(self):
        """Write the advanced generator state back to torch (call once, also on error paths)."""
        if self.key is None:
            return
        if self._pending is not None:                    # a batch drawn ahead that nobody took: back to the state in front of it
            self._pending.result()
            self._pending = None
            self.key, left, nxt = self._snap
            self.left, self.next = C.c_int64(left), C.c_int64(nxt)
        self._pool = None
        for _, ev in self.ring:                          # the ring goes back once the last copies out of it are done
            ev.synchronize()
        _RING_CACHE.setdefault(self._ring_key, []).append(self.ring)
        self.ring = None
        w = self._st.view(torch.int64)
        self._st.view(torch.int32)[_MT_LEFT] = self.left.value
        w[_MT_NEXT] = self.next.value
        w[_MT_STATE:_MT_STATE + _MT_WORDS] = self.key.to(torch.int64) & 0xffffffff
        torch.set_rng_state(self._st)
        self.key = None


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def _dist_info(group):
    import torch.distributed as dist
    if group is None or not dist.is_available() or not dist.is_initialized():
        return None, 0, 1
    g = None if group is True else group
    return dist, dist.get_rank(g), dist.get_world_size(g)


def vq_features(features: torch.Tensor, importance: torch.Tensor, codebook_size: int, vq_chunk: int = 2 ** 16,
                steps: int = 1000, decay: float = 0.8, scale_normalize: bool = False, silent: bool = False,
                group=None, batches=None, init_rand: Optional[torch.Tensor] = None, return_errors: bool = False,
                shard_final: bool = True, ops=None, device_rng: bool = False, stats: Optional[dict] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """compression/vq.py:49-87.  group=None: single GPU.  group=True (default process group) or a
    ProcessGroup: sharded Lloyd steps with one all-reduce of S[K, D+1] per step.
    The per-step `.item()` host sync of the reference (vq.py:71) is deferred to the end.
    Batch indices: by default the reference's draws exactly -- `torch.randint` on the CPU default generator (vq.py:69);
    device_rng=True draws on the GPU instead (removes the host-RNG floor of ~2 ms/step, different numbers).
    stats: optional dict that receives `lloyd_seconds`, `lloyd_steps`, `final_assignment_seconds` (wall clock, with a
    device synchronisation at the two phase boundaries -- only when given)."""
    dist, rank, world = _dist_info(group)
    pg = None if group is True else group
    dev = features.device
    N = features.shape[0]
    importance_n = importance / importance.max()
    vq_model = VectorQuantize(channels=features.shape[-1], codebook_size=codebook_size, decay=decay, ops=ops).to(device=dev)
    if init_rand is None and world > 1:
        init_rand = torch.rand(codebook_size, features.shape[-1], device=dev)
        dist.broadcast(init_rand, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg)
    vq_model.uniform_init(features, init_rand)

    feats = features.detach().contiguous().float()
    imp = importance_n.detach().contiguous().float()
    n_steps = steps if batches is None else len(batches)
    err_local = torch.zeros(max(n_steps, 1), dtype=torch.float64, device=dev)       # per-step sums of min distances
    batch_sizes = []
    scratch = {}
    fused = {}
    use_fused = _FUSED_STEP and hasattr(vq_model.ops, "step_sums") and feats.is_cuda
    it = range(steps) if batches is None else range(len(batches))
    src_rank = (dist.get_global_rank(pg, 0) if pg is not None else 0) if world > 1 else 0
    if world > 1 and batches is None and not device_rng:
        # Every rank must draw the SAME batches. Not a broadcast of the indices per step (2 MB for a 2^18 batch, 8 MB for
        # 2^20: more than the all-reduce it would accompany) but ONE broadcast of rank 0's CPU generator state (5 KB): all
        # ranks then continue the same MT19937 stream themselves (csrc/draws.hip), and all are left at the same state.
        st = torch.get_rng_state().to(dev)
        dist.broadcast(st, src=src_rank, group=pg)
        torch.set_rng_state(st.cpu())
    draws = None if batches is not None else _BatchDraws(N, vq_chunk, steps, dev, device_rng)
    if stats is not None:
        _sync(dev)
        t_loop = time.perf_counter()
    try:
        for s in it:
            if batches is not None:
                batch = batches[s].to(device=dev, dtype=torch.int64)
                B = int(batch.numel())
                lo, hi = (rank * B) // world, ((rank + 1) * B) // world
                batch = batch[lo:hi]
            elif world > 1 and device_rng:                                          # GPU generators differ per rank
                batch = draws.next_batch()
                dist.broadcast(batch, src=src_rank, group=pg)
                B = int(batch.numel())
                lo, hi = (rank * B) // world, ((rank + 1) * B) // world
                batch = batch[lo:hi]
            else:
                B = vq_chunk
                lo, hi = (rank * B) // world, ((rank + 1) * B) // world
                batch = draws.next_batch(lo, hi)                                     # this rank's slice of the common batch
            with torch.no_grad():
                # the slice's distance sum lands in this step's slot of err_local (reduced ONCE after the loop: the
                # errors are reporting only); the exchange of the step is ONE in-place all-reduce of S[K, D+1]
                gsl = batch.contiguous()
                # the loop owns codebook, sums and search scratch between its steps, so the step's second half can prepare the
                # next step's search (HipOps.step_sums / step_apply: 5 launches per step); other shapes / injected ops: the
                # plain pair
                res = (vq_model.ops.step_sums(fused, feats, imp, gsl, vq_model.codebook.data, err_local[s:s + 1])
                       if use_fused and hi > lo else None)
                if res is not None:
                    S = res[1]
                    if world > 1:
                        dist.all_reduce(S, group=pg)
                    vq_model.ops.step_apply(fused, vq_model.codebook.data, vq_model.entry_importance.data, vq_model.decay,
                                            vq_model.eps, scale_normalize)
                else:
                    fused.clear()                                                       # the protocol restarts at step 0
                    _, S, _ = vq_model.partial_sums(feats, imp, gather=gsl, scratch=scratch, dsum_out=err_local[s:s + 1])
                    if world > 1:
                        dist.all_reduce(S, group=pg)
                    vq_model.apply_sums(S, scale_normalize=scale_normalize)
            batch_sizes.append(B)
    finally:
        if draws is not None:
            draws.finish()
    if stats is not None:
        _sync(dev)
        stats["lloyd_seconds"] = time.perf_counter() - t_loop
        stats["lloyd_steps"] = len(batch_sizes)
    # (the reference runs gc.collect() + torch.cuda.empty_cache() here, vq.py:78-79: 32 ms of a 100 ms call, and nothing of
    # this loop waits for the collector -- the per-step scratch is reused, not reallocated)

    start = time.time()
    if world > 1 and shard_final:
        lo, hi = (rank * N) // world, ((rank + 1) * N) // world
        _, local = vq_model.ops.assign(feats[lo:hi], vq_model.codebook.data)
        # ranges differ by at most one point; RCCL's all-gather wants equal counts, so every rank sends ceil(N / world) slots
        sizes = [((r + 1) * N) // world - (r * N) // world for r in range(world)]
        width = max(sizes)
        send = torch.zeros(width, dtype=torch.int64, device=dev)
        send[:hi - lo] = local
        gathered = torch.empty(world * width, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, send, group=pg)
        vq_indices = torch.cat([gathered[r * width:r * width + sz] for r, sz in enumerate(sizes)], 0)
    else:
        _, vq_indices = vq_model(feats)
    if vq_indices.is_cuda:
        torch.cuda.synchronize(device=vq_indices.device)
    end = time.time()
    if stats is not None:
        stats["final_assignment_seconds"] = end - start
    if not silent:
        print(f"calculating indices took {end - start} seconds ")
    if return_errors:
        if world > 1:
            dist.all_reduce(err_local, group=pg)
        errors = [float(d) / max(b, 1) for d, b in zip(err_local.tolist(), batch_sizes)]
        return vq_model.codebook.data.detach(), vq_indices.detach(), errors
    return vq_model.codebook.data.detach(), vq_indices.detach()


def _say(silent: bool, msg: str):
    if not silent:
        print(msg)


def join_features(all_features: torch.Tensor, keep_mask: torch.Tensor, codebook: torch.Tensor,
                  codebook_indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same contract as compression/vq.py:90-103: the row table is [codebook ; kept rows in Gaussian order]; a kept
    Gaussian points at its own row behind the codebook, every other one at its codeword.
    The kept rows' numbers are the running count of the mask (no index tensors are materialised and scattered)."""
    n_code = codebook.shape[0]
    own_row = torch.cumsum(keep_mask, 0, dtype=torch.long) + (n_code - 1)
    rows = own_row.masked_scatter(~keep_mask, codebook_indices.to(torch.long))
    return torch.cat((codebook, all_features[keep_mask])), rows


@dataclass
class CompressionSettings:
    """Field-compatible with compression/vq.py:106-114 (the drivers construct it by keyword)."""
    codebook_size: int
    importance_prune: float
    importance_include: float
    importance_include_relative: float
    steps: int
    decay: float
    batch_size: int


def _table_for(features: torch.Tensor, importance: torch.Tensor, cfg: CompressionSettings, what: str, silent: bool, group,
               scale_normalize: bool = False):
    """One quantisation job: rows whose importance exceeds `cfg.importance_include` stay verbatim, the rest go through
    vq_features. -> (row table, per-Gaussian row numbers). Shared by the colour and the covariance path."""
    keep = importance > cfg.importance_include
    _say(silent, f"{what}: {100.0 * float(keep.float().mean()):.2f}% of the rows kept uncompressed")
    todo = ~keep
    if bool(todo.any()):
        _say(silent, f"{what}: vector-quantising {int(todo.sum())} rows into {cfg.codebook_size} codewords")
        codebook, assigned = vq_features(features[todo], importance[todo], cfg.codebook_size, cfg.batch_size, cfg.steps,
                                         scale_normalize=scale_normalize, silent=silent, group=group)
    else:                                   # everything kept: an empty codebook in front of the kept rows
        codebook = features.new_empty((0, features.shape[-1]))
        assigned = torch.empty(0, dtype=torch.long, device=features.device)
    return join_features(features, keep, codebook, assigned)


def compress_color(gaussians, color_importance: torch.Tensor, color_comp: CompressionSettings,
                   color_compress_non_dir: bool, silent: bool, group=None):
    """compression/vq.py:117-147 (`gaussians` is duck-typed: get_features, set_color_indexed). With
    color_compress_non_dir the DC coefficient is quantised with the rest, otherwise only the directional ones are."""
    sh = gaussians.get_features.detach()
    first = 0 if color_compress_non_dir else 1
    table, rows = _table_for(sh[:, first:].flatten(-2), color_importance, color_comp, "colour", silent, group)
    gaussians.set_color_indexed(table.reshape(-1, sh.shape[1] - first, 3), rows)


def compress_covariance(gaussians, gaussian_importance: torch.Tensor, gaussian_comp: CompressionSettings, silent: bool,
                        group=None, extract_rot_scale=None, to_full_cov=None):
    """compression/vq.py:149-191: quantise the normalised covariances (upper triangles, trace-normalised codewords), then
    split every table row back into rotation + scale. The eigendecomposition helpers default to c3dgs_amd.encode's HIP
    versions of utils/splats.py:7-35; the reference's own functions can still be passed in."""
    cov6 = gaussians.get_normalized_covariance(strip_sym=True).detach()
    table, rows = _table_for(cov6, gaussian_importance, gaussian_comp, "covariance", silent, group, scale_normalize=True)
    if extract_rot_scale is None or to_full_cov is None:
        from . import encode
        extract_rot_scale, to_full_cov = encode.extract_rot_scale, encode.to_full_cov
    rot, scale = extract_rot_scale(to_full_cov(table))
    gaussians.set_gaussian_indexed(rot.to(table.device), scale.to(table.device), rows)


def _resolve_threshold(cfg: Optional[CompressionSettings], importance: torch.Tensor, what: str, silent: bool):
    """importance_include=None means "take the importance_include_relative quantile" (compression/vq.py:209-218)."""
    if cfg is not None and cfg.importance_include is None:
        cfg.importance_include = float(torch.quantile(importance, cfg.importance_include_relative))
        _say(silent, f"{what}: keep threshold set to {cfg.importance_include}")


def compress_gaussians(gaussians, color_importance: torch.Tensor, gaussian_importance: torch.Tensor,
                       color_comp: Optional[CompressionSettings], gaussian_comp: Optional[CompressionSettings],
                       color_compress_non_dir: bool, prune_threshold: float = 0., silent: bool = False, group=None,
                       extract_rot_scale=None, to_full_cov=None):
    """compression/vq.py:194-223: prune by colour importance, then the two quantisation jobs. (Unlike the reference,
    a job whose settings are None is skipped before its threshold is looked at.)"""
    with torch.no_grad():
        if prune_threshold >= 0:
            survivors = color_importance > prune_threshold
            _say(silent, f"pruning {100.0 * (1.0 - float(survivors.float().mean())):.2f}% of the Gaussians")
            gaussians.mask_splats(survivors)
            color_importance, gaussian_importance = color_importance[survivors], gaussian_importance[survivors]
        _resolve_threshold(color_comp, color_importance, "colour", silent)
        _resolve_threshold(gaussian_comp, gaussian_importance, "covariance", silent)
        if color_comp is not None:
            compress_color(gaussians, color_importance, color_comp, color_compress_non_dir, silent=silent, group=group)
        if gaussian_comp is not None:
            compress_covariance(gaussians, gaussian_importance, gaussian_comp, silent=silent, group=group,
                                extract_rot_scale=extract_rot_scale, to_full_cov=to_full_cov)
