"""Kernel timeline of ONE headline step from a rocprofv3 kernel trace of bench.py (start offset, duration, gap to the previous
kernel's end, name) -- shows every launch a step makes, including torch's fills and copy kernels that no stage timer covers.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-vq --no-extras --no-live-traffic --steps 6
    python tools/step_timeline.py gpurun_out/tl [step]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "render_backward" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else min(6, len(idx) - 1)
a, b = idx[k - 1], idx[k]
t0, prev = int(rows[a + 1]["Start_Timestamp"]), int(rows[a]["End_Timestamp"])
tot = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
    prev, tot = e, tot + e - s
print(f"kernels {tot / 1e3:.1f} us in a span of {(prev - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us, {b - a} launches")
