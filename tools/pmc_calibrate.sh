#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE for gathers and scattered stores on this GPU: two rocprofv3 --pmc passes (kernel trace
# only, as MI355X_MICROARCH.md prescribes) over tools/pmc_calibrate.py, raw counters next to the byte counts each counting model
# predicts.   bash tools/pmc_calibrate.sh [out.txt]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=${1:-gpurun_out/pmc_calibration.txt}
D=gpurun_out/pmc_cal; rm -rf $D; mkdir -p $D
PMC_CAL_OUT=$D/expected_f.json rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o f -- python3 tools/pmc_calibrate.py > $D/fetch.log 2>&1
PMC_CAL_OUT=$D/expected_r.json rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $D/rdreq -o r -- python3 tools/pmc_calibrate.py > $D/rdreq.log 2>&1
PMC_CAL_OUT=$D/expected_w.json rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o w -- python3 tools/pmc_calibrate.py > $D/write.log 2>&1
python3 - > "$OUT" <<'PY'
import csv, glob, json, collections
def load(pat, counter):
    f = glob.glob(pat, recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "probe_" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0].replace("void c3dgs::", "").replace("c3dgs::", "")].append(float(r["Counter_Value"]) * 1024.0)
    return per
def load_req():
    f = glob.glob("gpurun_out/pmc_cal/rdreq/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            if "probe_" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void c3dgs::", "").replace("c3dgs::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per
req = load_req()
fetch, write = load("gpurun_out/pmc_cal/fetch/**/*counter_collection.csv", "FETCH_SIZE"), load("gpurun_out/pmc_cal/write/**/*counter_collection.csv", "WRITE_SIZE")
exp = json.load(open("gpurun_out/pmc_cal/expected_f.json"))
print("PMC calibration on this GPU (raw counter x 1024 = bytes; no correction applied). Table 6 GiB >> 256 MiB Infinity Cache.")
print("models: requested = lanes x record bytes; u32 / u64 / u128 = unique 32-byte sectors / 64-byte / 128-byte lines touched x their size")
for name in ("probe_stream_kernel", "probe_gather_kernel<3>", "probe_gather_kernel<12>", "probe_scatter36_kernel"):
    es = [e for e in exp if e["kernel"] == name]
    for which, per in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        vals = per.get(name, [])
        for e, v in zip(es, vals):
            if v == 0 and which == "WRITE_SIZE" and e["kind"] != 3: continue
            if which == "FETCH_SIZE" and e["kind"] == 3 and v == 0: continue
            if which == "FETCH_SIZE" and name in req and len(req[name].get("TCC_EA0_RDREQ_sum", [])) > e["launch"]:
                q = {k: v[e["launch"]] for k, v in req[name].items()}
                by = 128 * q.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * q.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * q.get("TCC_EA0_RDREQ_32B_sum", 0)
                print(f"{name:26s} launch {e['launch']} read requests: total {q.get('TCC_EA0_RDREQ_sum', 0)/1e6:.3f} M = 128B {q.get('TCC_EA0_RDREQ_128B_sum', 0)/1e6:.3f} M + 64B {q.get('TCC_EA0_RDREQ_64B_sum', 0)/1e6:.3f} M + 32B {q.get('TCC_EA0_RDREQ_32B_sum', 0)/1e6:.3f} M"
                      f" -> bytes by request size {by/1e6:10.2f} MB (x{by / v if v else 0:.3f} of raw FETCH_SIZE; {by/e['u64']:.3f} of u64, {by/e['u128']:.3f} of u128)")
            print(f"{name:26s} launch {e['launch']} {which}: raw {v/1e6:10.2f} MB | requested {e['requested']/1e6:9.2f} (raw/req {v/e['requested']:.3f}) | "
                  f"u32 {e['u32']/1e6:9.2f} ({v/e['u32']:.3f}) | u64 {e['u64']/1e6:9.2f} ({v/e['u64']:.3f}) | u128 {e['u128']/1e6:9.2f} ({v/e['u128']:.3f})")
PY
cat "$OUT"
