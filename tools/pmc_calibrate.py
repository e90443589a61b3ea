"""The profiled program of tools/pmc_calibrate.sh: known access patterns (csrc/probe.hip) on a table far larger than the
256 MiB Infinity Cache, with the byte counts every counting model would predict written next to the counters.

Patterns: a coalesced 16-byte-per-lane stream (the case MI355X_MICROARCH.md calibrates: FETCH_SIZE reports 1/2), random
48-byte records (the splat records the blend kernels gather), random 192-byte rows (the SH rows preprocess gathers), random
36-byte slot stores (render_backward's partial sums)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from c3dgs_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.lib()
TABLE = 6 * 1024 ** 3
table = torch.empty(TABLE, dtype=torch.uint8, device=dev)
table.zero_()
out = torch.zeros(4, dtype=torch.int32, device=dev)
g = torch.Generator(device=dev).manual_seed(5)
expected = []


def lines(byte_lo, byte_hi, gran):
    """unique `gran`-byte lines touched by the ranges [lo, hi)"""
    a, b = byte_lo // gran, (byte_hi - 1) // gran
    ids = [a, b] if gran >= 64 else [a + k for k in range(0, 8)]
    if gran >= 64:
        allv = torch.cat([a, b])
        if int(((b - a) > 1).sum()):
            allv = torch.cat([allv, a + 1])
    else:
        allv = torch.cat([torch.minimum(a + k, b) for k in range(int((b - a).max()) + 1)])
    return int(torch.unique(allv).numel())


def run(kind, name, rec_bytes, n, reps=3):
    for r in range(reps):
        if kind == 0:
            nbytes = n * 16
            _lib.check(L.c3dgs_debug_gather_probe(0, n, table.data_ptr() + r * nbytes, None, out.data_ptr(), None))
            e = dict(requested=nbytes, u32=nbytes, u64=nbytes, u128=nbytes)
        else:
            nrec = TABLE // rec_bytes
            idx = torch.randint(0, nrec, (n,), generator=g, device=dev, dtype=torch.int64)
            lo = idx * rec_bytes
            e = dict(requested=n * rec_bytes, u32=lines(lo, lo + rec_bytes, 32) * 32, u64=lines(lo, lo + rec_bytes, 64) * 64,
                     u128=lines(lo, lo + rec_bytes, 128) * 128, unique_records=int(torch.unique(idx).numel()))
            idx32 = idx.to(torch.int32)            # < 2^31 records
            torch.cuda.synchronize()
            _lib.check(L.c3dgs_debug_gather_probe(kind, n, table.data_ptr(), idx32.data_ptr(), out.data_ptr(), None))
        torch.cuda.synchronize()
        expected.append(dict(kernel=name, kind=kind, launch=r, n=n, record_bytes=rec_bytes, **e))


run(0, "probe_stream_kernel", 16, 128 * 1024 ** 2 // 16 * 8)          # 1 GiB per launch
run(1, "probe_gather_kernel<3>", 48, 8_000_000)
run(2, "probe_gather_kernel<12>", 192, 3_000_000)
run(3, "probe_scatter36_kernel", 36, 8_000_000)
json.dump(expected, open(os.environ.get("PMC_CAL_OUT", "gpurun_out/pmc_cal_expected.json"), "w"), indent=1)
print("done")
