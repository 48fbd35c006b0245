"""Kernels of the LAST batched call in a rocprofv3 kernel_trace.csv, in launch order with their hardware queue: start offset, duration, name.
The last call starts at the last PackBBoxK launch that follows a gap of more than `gap_us` (default 300) without kernels."""
import csv, sys
f = sys.argv[1]; gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = 0; prev_end = 0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if i and s - prev_end > gap: start = i
    prev_end = max(prev_end, e)
t0 = int(rows[start]["Start_Timestamp"])
queues = {}
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r.get("Queue_Id", "?"), "q%d" % len(queues))
    name = r["Kernel_Name"].split("(")[0].replace("void qn::", "").replace("qn::", "").replace("k_lanes<", "")
    print("%9.1f us  dur %7.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:70]))
print("total %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows[start:]) - t0) / 1e3))
