"""Diagnose one fuzz case of tests/test_fuzz_gpu.py at full detail (flipped pixels, gradient errors with and without them).
python tools/diag_fuzz.py SEED [SEED ...]   (needs a GPU)"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import cases, gpu_util, synth, fullsize
from tests.test_fuzz_gpu import _random_case

for seed in [int(x) for x in sys.argv[1:]]:
    inp, cam, indexed, what = _random_case(seed, large=seed >= 1000)
    st = cases.oracle_forward(inp, cam)
    from oracle import oracle as orc
    dL = synth.grad_image(cam["W"], cam["H"], seed=seed).numpy()
    ref = orc.rasterize_backward(st, dL)
    fw = gpu_util.hip_forward(inp, cam, indexed)
    try:
        out = fullsize.compare(inp, cam, indexed, st, ref, dL, fw)
    except KeyError as e:           # degenerate case (nothing rendered): nothing to compare
        print(what, "-- skipped:", repr(e))
        continue
    keep = {k: out[k] for k in ("flipped_pixels", "deepest_tile_list", "deepest_blend", "psnr_db", "grad_rel_inf",
                                "grad_rel_inf_excluding_flips", "gaussians_sharing_a_tile_with_a_flip") if k in out}
    print(what)
    print(json.dumps(keep, indent=1))
