"""Summarise FETCH_SIZE / WRITE_SIZE PMC passes into HBM bytes per launch for each c3dgs kernel.

Units and corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
1/2 of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so it is DOUBLED here; WRITE_SIZE
reads exactly for streaming stores and float atomics.
Calibrated for THIS code's access patterns in round 3 (tools/pmc_calibrate.sh -> profiles/r03_pmc_calibration.txt): random
48-byte record gathers and 192-byte row gathers issue 128-byte read requests ONLY (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ), each
tallied at 64 B, so the x2 is exact for the gather kernels as well (`fetch_calibration` in the output); WRITE_SIZE counts
32-byte sectors for scattered partial-line stores (reported as is)."""
import collections
import csv
import json
import re
import sys

STAGE = [("mark_visible", "mark_visible"), ("sum_partials", "sum_partials"), ("backward_prep_kernel", "backward_prep"),
         ("backward_preprocess", "backward_preprocess"), ("preprocess_kernel", "preprocess"),
         ("duplicate_with_keys", "duplicate_with_keys"), ("stamp_slots", "duplicate_with_keys"),
         ("identify_ranges", "identify_ranges"), ("render_forward", "render_forward"), ("render_backward", "render_backward"),
         ("wd_mfma", "weighted_distance"), ("weighted_distance_kernel", "weighted_distance")]


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for key, stage in STAGE:
            if key in name:
                per[stage].append(float(r["Counter_Value"]))
                break
    return per


def summarize(fetch_csv, write_csv):
    fetch = load(fetch_csv, "FETCH_SIZE")
    write = load(write_csv, "WRITE_SIZE")
    out = {}
    for st in sorted(set(fetch) | set(write)):
        f = fetch.get(st, [0.0])
        w = write.get(st, [0.0])
        # warm launches only: drop the first one
        f = f[1:] if len(f) > 1 else f
        w = w[1:] if len(w) > 1 else w
        fb = 2.0 * 1024.0 * sum(f) / len(f)
        wb = 1024.0 * sum(w) / len(w)
        out[st] = {"hbm_bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "fetch_x2_applied": True,
                   "fetch_calibration": "x2 exact: every read request is a 128-B line tallied at 64 B, streams and gathers alike (profiles/r03_pmc_calibration.txt)",
                   "launches_averaged": len(f)}
    return out


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1], sys.argv[2]), indent=1))
