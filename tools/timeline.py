"""Print the last kernels of a rocprofv3 kernel trace as a timeline: python tools/timeline.py gpurun_out/kstats_<tag>/k_kernel_trace.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
