"""Diagnostic (timing build): which role of which worker ends an encode launch last.  Runs the bench workload's first
blocks through tools/libfqsx_timing.so and reads the per-launch role time stamps (fqsx_dna_trace).
usage: python tools/gpu_roles.py [n_blocks=100] [len=150] [T=64]"""
import ctypes as C, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
G, gs = (7500000, 8) if L == 150 else (5000000, 5)
reads = synth_reads(1000000, L, G, 2)
rec = hp.Records([read_id(i) for i in range(len(reads))], reads, reads)
header = hp.make_header(T, "se_sorted", gs)
blocks = hp.form_blocks(rec, "se_sorted")[:nblk]
lib = os.path.join(ROOT, "tools", "libfqsx_timing.so")
c = DnaCodec(header, lib_path=lib)
segs = []   # (block, n_segments)
for g, idx in enumerate(blocks):
    bases, off = hp.block_arrays(rec, idx)
    before = c.kernel_times()["encode_launches"] if False else None
    c.encode_block(bases, off, g)
c._lib.fqsx_dna_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
c._lib.fqsx_dna_trace.restype = C.c_int
buf = np.zeros((4096, T, 32), dtype=np.uint64)
n = c._lib.fqsx_dna_trace(c._h, buf.ctypes.data, 4096)
names = ["head", "resolve", "models", "scout", "inserter", "range coder"]
slots = [1, 2, 3, 4, 5, 7]   # (slot 6: resolving wave, reads done)


def report(tr, tag):
    n = len(tr)
    start = tr[:, :, 0].min(axis=1, keepdims=True)          # launch start = earliest resolve start
    ends = (tr[:, :, slots] - start[:, :, None]) * 0.01      # us
    ends[tr[:, :, slots] == 0] = 0
    launch_len = ends.max(axis=(1, 2))
    wend = ends.max(axis=2)                                  # per worker: its last role's end
    last_worker = wend.argmax(axis=1)
    last_role = np.array([ends[i, last_worker[i]].argmax() for i in range(n)])
    res = {"launches": tag, "n": int(n), "mean_launch_us": float(launch_len.mean()), "mean_worker_end_us": float(wend.mean()),
           "mean_worker_over_launch": float((wend.mean(axis=1) / launch_len).mean()),
           "last_role_histogram": {names[k]: int((last_role == k).sum()) for k in range(len(names))},
           "mean_end_us_by_role": {names[k]: round(float(ends[:, :, k].mean()), 1) for k in range(len(names))},
           "slowest_worker_end_us_by_role": {names[k]: round(float(np.mean([ends[i, last_worker[i], k] for i in range(n)])), 1) for k in range(len(names))},
           "resolve_reads_done_us_mean": round(float(((tr[:, :, 6] - start) * 0.01).mean()), 1),
           "p50_p90_p99_launch_us": [round(float(np.percentile(launch_len, q)), 1) for q in (50, 90, 99)]}
    # what distinguishes the slowest worker of a launch from the average one (resolving wave's per-launch counters)
    cn = ["slow positions", "sweeps merged", "dirty chunks", "fast positions", "sweep-frontier wait us", "scout wait us", "generic positions", "local-insert wait us"]
    vals = tr[:, :, 8:16].astype(np.float64)
    vals[:, :, [4, 5, 7]] *= 0.01
    res["resolver_counters_mean_worker"] = {cn[k]: round(float(vals[:, :, k].mean()), 2) for k in range(8)}
    res["resolver_counters_slowest_worker"] = {cn[k]: round(float(np.mean([vals[i, last_worker[i], k] for i in range(n)])), 2) for k in range(8)}
    xn = ["slow path", "slow: resolve counts", "slow: sweep merge (+frontier wait)", "find_counts", "flush_pushes (all)", "code_keys", "read head wait", "coding-queue wait",
          "quiet_miss_mask", "quiet stretches", "prologue", "chunk into queue", "end-of-chunk flush", "total"]
    xv = tr[:, :, 16:30].astype(np.float64) * 0.01
    res["resolver_sections_us_mean_worker"] = {xn[k]: round(float(xv[:, :, k].mean()), 1) for k in range(14)}
    res["resolver_sections_us_slowest_worker"] = {xn[k]: round(float(np.mean([xv[i, last_worker[i], k] for i in range(n)])), 1) for k in range(14)}
    res["generic_early_mean_slowest"] = [round(float(tr[:, :, 30].mean()), 2), round(float(np.mean([tr[i, last_worker[i], 30] for i in range(n)])), 2)]
    res["generic_flag3_nocascade_mean_slowest"] = [round(float(tr[:, :, 31].mean()), 2), round(float(np.mean([tr[i, last_worker[i], 31] for i in range(n)])), 2)]
    rend = ends[:, :, 1]   # resolving wave's end per worker
    res["corr_resolve_end_with"] = {cn[k]: round(float(np.corrcoef(rend.reshape(-1), vals[:, :, k].reshape(-1))[0, 1]), 3) for k in range(8)}
    print(json.dumps(res))


tr = buf[:n].astype(np.int64)
report(tr[:min(n, 2100)], "blocks 0..69 (two reads per worker and launch)")
for a, b in ((0, 10), (10, 30), (30, 70)):
    if n >= 30 * b:
        report(tr[30 * a:30 * b], f"blocks {a}..{b - 1}")
if nblk > 110 and n > 2100:
    k = nblk - 100
    report(tr[n - k:], "blocks >= 100 (one launch per block)")
