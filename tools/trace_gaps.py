"""Reads a rocprofv3 kernel-trace CSV: per kernel name count / total / mean duration, and the idle time between
consecutive kernels (gaps), for the second half of the trace (the second pass of tools/gpu_warm_trace.py)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
ev = ev[len(ev) // 2:]
tot = collections.defaultdict(lambda: [0, 0])
gap = collections.defaultdict(lambda: [0, 0])
for i, (s, e, n) in enumerate(ev):
    tot[n][0] += 1; tot[n][1] += e - s
    if i:
        g = s - ev[i - 1][1]
        if g < 5_000_000:
            gap[n][0] += 1; gap[n][1] += max(g, 0)
span = ev[-1][1] - ev[0][0]
print(f"span {span/1e6:.1f} ms, kernels {sum(v[1] for v in tot.values())/1e6:.1f} ms, gaps {sum(v[1] for v in gap.values())/1e6:.1f} ms")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:28s} n={c:6d} total {t/1e6:8.2f} ms mean {t/c/1e3:8.2f} us   gap before: mean {gap[n][1]/max(gap[n][0],1)/1e3:6.2f} us total {gap[n][1]/1e6:7.2f} ms")
