"""Where the GPU idles inside the timed steps: reads a rocprofv3 --kernel-trace CSV (one row per kernel with start / end timestamps), takes the last
`frac` of the trace (the timed region of bench.py: setup and warm-up come first) and prints the busy fraction, the idle time by gap size and the
kernels that most often stand in front of a gap > 5 us (host synchronisation points: status polls of the Krylov loops, downloads).
usage: python tools/gpu_gaps.py kernel_trace.csv [frac=0.5]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
ks = ks[int(len(ks) * (1.0 - frac)):]
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy, end, gaps = 0, ks[0][0], []
for s, e, n in ks:
    if s > end:
        gaps.append((s - end, prev, n))
        busy += e - s
        end = e
    else:
        busy += max(0, e - max(s, end))
        end = max(end, e)
    prev = n
wall = t1 - t0
print("kernels %d  wall %.2f ms  busy %.2f ms = %.1f %%" % (len(ks), wall / 1e6, busy / 1e6, 100.0 * busy / wall))
for lo, hi in ((0, 2000), (2000, 5000), (5000, 20000), (20000, 100000), (100000, 10 ** 12)):
    g = [d for d, _, _ in gaps if lo <= d < hi]
    print("  gaps %6.0f-%-8.0f us: %6d  total %.2f ms" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
c = collections.Counter()
tsum = collections.Counter()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
for d, n, nxt in gaps:
    if d > 5000:
        key = short(n) + " -> " + short(nxt)
        c[key] += 1
        tsum[key] += d
for k, v in tsum.most_common(12):
    print("  gap > 5 us between: %-84s %5d x  %.2f ms" % (k, c[k], v / 1e6))
