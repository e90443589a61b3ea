#!/usr/bin/env python3
"""Prints the headline numbers of bench.py JSON lines (files given as arguments) on one line each."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    st = d.get("stages_ms", {})
    vq = d.get("vq", {})
    print(f"{f}: {d['value']:.1f} views/s {d['ms_per_step']:.3f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in st.items())
          + (f" | vq {vq.get('value', 0):.0f}/s assign {vq.get('assign_kernel_ms', 0):.3f} acc {vq.get('accumulate_kernel_ms', 0):.3f} "
             f"first {vq.get('accumulate_first_step_ms', 0):.3f}" if vq and "value" in vq else "")
          + (f" | parity ok={d['parity'].get('ok')}" if "parity" in d else ""))
