#!/bin/bash
# SQ counters of the VQ search kernel (rocprofv3 --pmc passes, kernel trace only). bash tools/pmc_sq_vq.sh
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_sq_vq
rm -rf "$OUT"; mkdir -p "$OUT"
ITERS=3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/a" -o a -- python3 tools/prof_vq.py > "$OUT/a.log" 2>&1
ITERS=3 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/b" -o b -- python3 tools/prof_vq.py > "$OUT/b.log" 2>&1
python3 - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc_sq_vq")
for tag in ("a", "b"):
    f = glob.glob(os.path.join(out, tag, "*counter_collection.csv"))
    if not f:
        print(tag, "no counter file"); print(open(os.path.join(out, tag + ".log")).read()[-1500:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        for name in ("wd_f16_kernel", "wd_mfma_kernel", "wd_fixup_list"):
            if name in k:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    for name, d in acc.items():
        print(tag, name, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
