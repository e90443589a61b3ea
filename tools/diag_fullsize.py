#!/usr/bin/env python3
"""Prints the full-size HIP-vs-oracle comparison (tests/fullsize.py) of one BASELINE configuration as JSON.
    python tools/diag_fullsize.py config3_3M_indexed [config4_6M_sens ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from tests import fullsize, synth  # noqa: E402

for name in sys.argv[1:] or ["config3_3M_indexed"]:
    inp, intr, ev, indexed = fullsize.config_inputs(name)
    cam = orc.camera(intr.numpy(), ev.numpy())
    backward = name != "config2_1M_fwd"
    dL = synth.grad_image(cam["W"], cam["H"]).numpy() if backward else None
    st, ref, tf, tb = fullsize.oracle_view(inp, cam, dL)
    res = fullsize.compare(inp, cam, indexed, st, ref, dL)
    res["oracle_seconds"] = [tf, tb]
    print(json.dumps({name: res}), flush=True)
