"""From a rocprofv3 kernel trace of a multi-stream run: in the busiest 50 ms window (the timed multi-stream region of bench.py), the fraction
of wall time with >= 1 kernel running and the mean number of kernels running concurrently (how much of the stream-level parallelism the
GPU actually overlaps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0 = iv[0][0]; W = 50_000_000
def stats(lo, hi):
    ev = []
    for s, e in iv:
        if e <= lo or s >= hi: continue
        ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
    ev.sort()
    cur = 0; last = lo; busy = 0; area = 0; hist = {}
    for t, d in ev:
        dt = t - last
        if dt > 0:
            if cur > 0: busy += dt
            area += cur * dt; hist[cur] = hist.get(cur, 0) + dt
        cur += d; last = t
    return busy / (hi - lo), area / (hi - lo), hist, len(ev) // 2
best = None
lo = t0
while lo + W < iv[-1][1]:
    b, a, h, n = stats(lo, lo + W)
    if best is None or n > best[3]: best = (b, a, h, n, lo)
    lo += W // 2
b, a, h, n, lo = best
print("busiest 50 ms window (at +%.0f ms, %d kernels): busy %.1f %%  mean concurrent kernels %.2f" % ((lo - t0) / 1e6, n, 100 * b, a))
print("time share by number of concurrent kernels:", {k: round(100 * v / W, 1) for k, v in sorted(h.items())})
