"""Print the kernels of the LAST registration in a rocprofv3 kernel_trace.csv in launch order: start offset, duration, gap."""
import csv, sys, glob
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last registration starts at the last pair of grid builds: find the second-to-last pack kernel
idx = [i for i, r in enumerate(rows) if "k_pack_points" in r["Kernel_Name"] or "k_pack_bbox_dims" in r["Kernel_Name"]]
start = idx[-2]
t0 = int(rows[start]["Start_Timestamp"]); prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void qn::", "").replace("qn::", "")
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:60]))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
