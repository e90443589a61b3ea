"""Median durations of the VQ assignment kernels from tools/kstats_vq.sh's trace."""
import csv, statistics as st
rows = list(csv.DictReader(open("gpurun_out/kstats_vq/k_kernel_trace.csv")))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in ["wd_mfma_kernel<48>", "wd_fixup_kernel<48>", "wd_mfma_kernel<6>", "wd_fixup_kernel<6>", "vq_accumulate"]:
    d = sorted(dur(r) for r in rows if name in r["Kernel_Name"])
    if d:
        print(name, len(d), "median %.1f us" % st.median(d), "p10 %.1f p90 %.1f max %.1f" % (d[len(d) // 10], d[len(d) * 9 // 10], d[-1]))
