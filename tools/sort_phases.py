"""Where a digit pass of the onesweep sorts spends its time: phase stamps (shader clock) of 64 tiles of the LAST pass of a sort,
from the "ostime" build variant.  C3DGS_LIB_PATH=c3dgs_amd/libc3dgs_hip_ostime.so python tools/sort_phases.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = ["ticket->", "loads", "rank loop", "barrier", "scan+look-back", "hist scan", "LDS reorder+barrier", "stores"]


def run(keys, vals, end_bit, label):
    kb, n = keys.element_size(), keys.numel()
    tb = int(L.c3dgs_debug_sort_temp_bytes(kb, n, end_bit))
    temp = torch.empty(max(tb, 256), dtype=torch.uint8, device=dev)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    for _ in range(3):
        _lib.check(L.c3dgs_debug_sort_pairs(kb, n, end_bit, keys.data_ptr(), ko.data_ptr(), vals.data_ptr(), vo.data_ptr(), temp.data_ptr(), tb, st))
    torch.cuda.synchronize()
    out = (C.c_uint64 * 512)()
    _lib.check(L.c3dgs_debug_sort_times(out))
    t = torch.tensor(list(out), dtype=torch.float64).view(64, 8)
    t = t[t[:, 0] > 0]
    d = (t[:, 1:] - t[:, :-1])
    print(f"== {label}: {t.shape[0]} sampled tiles of the last pass; shader-clock cycles (100 MHz constant clock if s_memrealtime; here readcyclecounter)")
    for k in range(7):
        print(f"  {NAMES[k + 1]:22s} mean {d[:, k].mean():10.0f}  min {d[:, k].min():10.0f}  max {d[:, k].max():10.0f}")
    tot = t[:, 7] - t[:, 0]
    print(f"  total per tile        mean {tot.mean():10.0f}  max {tot.max():10.0f}; first start .. last end = {t[:, 7].max() - t[:, 0].min():.0f}")


g = torch.Generator(device=dev).manual_seed(1)
z = torch.rand(3_000_000, device=dev, generator=g) * 10.0 + 2.0
dk = z.view(torch.int32)
run(dk, torch.arange(dk.numel(), device=dev, dtype=torch.int32), 32, "depth sort 3M x u32, pass 4 of 4")
tk = torch.randint(0, 8160, (16_400_000,), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
run(tk, torch.arange(tk.numel(), device=dev, dtype=torch.int32), 13, "tile sort 16.4M x u16, pass 2 of 2")
