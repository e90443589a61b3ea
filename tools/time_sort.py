"""Time the two onesweep sorts alone (C-ABI c3dgs_debug_sort_pairs): depth-like 32-bit keys of 3M items, tile-like 16-bit keys of
16.4M items. python tools/time_sort.py [--knockout]  (knock-out builds give wrong orders: timing only, results not checked)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(keys, vals, end_bit, n_iter=20):
    kb, n = keys.element_size(), keys.numel()
    tb = int(L.c3dgs_debug_sort_temp_bytes(kb, n, end_bit))
    temp = torch.empty(max(tb, 256), dtype=torch.uint8, device=dev)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    f = lambda: _lib.check(L.c3dgs_debug_sort_pairs(kb, n, end_bit, keys.data_ptr(), ko.data_ptr(), vals.data_ptr(), vo.data_ptr(), temp.data_ptr(), tb, st))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_iter):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_iter, ko, vo


g = torch.Generator(device=dev).manual_seed(1)
z = torch.rand(3_000_000, device=dev, generator=g) * 28.0 + 2.0
dk = z.view(torch.int32)
dv = torch.arange(dk.numel(), device=dev, dtype=torch.int32)
ms, ko, vo = run(dk, dv, 32)
ok = bool((ko[1:] >= ko[:-1]).all())
print(f"depth sort 3M x u32: {ms*1e3:.1f} us  sorted={ok}")
tk = torch.randint(0, 8160, (16_400_000,), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
tv = torch.arange(tk.numel(), device=dev, dtype=torch.int32)
ms, ko, vo = run(tk, tv, 13)
ok = bool((ko[1:] >= ko[:-1]).all())
print(f"tile sort 16.4M x u16 (13 bits): {ms*1e3:.1f} us  sorted={ok}")
