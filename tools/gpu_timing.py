"""Diagnostic: run the -DFQSX_TIMING build over the first N reads of the bench workload and print
the in-kernel section times summed over workers (10 ns ticks -> seconds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else os.path.join(ROOT, "tools", "libfqsx_timing.so")
max_blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30   # only the first blocks of the file
L = int(sys.argv[5]) if len(sys.argv) > 5 else 150           # read length: 150 = the metric's workload, 100 = BASELINE configs[1]
G, gs = (7500000, 8) if L == 150 else (5000000, 5)
reads = synth_reads(1000000, L, G, 2)[:n]
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", gs)
blocks = hp.form_blocks(rec, "se_sorted")
c = DnaCodec(header, lib_path=lib)
c.set_profiling(True)
t0 = time.time()
n_done = 0
if len(sys.argv) > 6 and sys.argv[6] == "warm":
    max_blocks = min(max_blocks, 70)
st_mid = None
for g, idx in enumerate(blocks[:max_blocks]):
    if g == 100:
        st_mid, kt_mid, n_mid, t_mid = c.stats(), c.kernel_times(), n_done, time.time()
    bases, off = hp.block_arrays(rec, idx)
    c.encode_block(bases, off, g)
    n_done += len(idx)
n = n_done
dt = time.time() - t0
st = c.stats(); kt = c.kernel_times()
if len(sys.argv) > 6 and sys.argv[6] == "warm":
    max_blocks = min(max_blocks, 70)
if st_mid is not None and len(sys.argv) > 6 and sys.argv[6] == "steady":   # report blocks >= 100 only
    st = {k: ([a - b for a, b in zip(v, st_mid[k])] if isinstance(v, list) else v - st_mid[k]) for k, v in st.items()}
    kt = {k: v - kt_mid[k] for k, v in kt.items()}
    n -= n_mid
    dt = time.time() - t_mid
    print("blocks >= 100 only:")
names = ["total", "spec", "fast", "slow", "post_q", "read_head", "lq_flush", "rough", "repair_missing", "find_counts"]
cn = dict(zip(["n_fast", "n_slow", "n_chunk", "n_dirty", "n_rough", "n_repm", "n_ext", "n_generic", "n_lqflush", "_rrwait", "_cqwait", "n_early", "n_p2"], st["timers"][10:23]))
cn["resolver: waiting for coding-queue space s"] = cn.pop("_cqwait") * 1e-8
cn["resolver: waiting for a chunk's sweep frontier s"] = cn.pop("_rrwait") * 1e-8
cn["coder wave: launch start to last symbol s"] = st["timers"][23] * 1e-8  # vs "total" = the resolving wave's
cn["slow: resolve counts s"] = st["timers"][26] * 1e-8
cn["coder wave: idle (queue empty) s"] = st["timers"][24] * 1e-8
cn["scout wave: waiting (ring full / no head) s"] = st["timers"][25] * 1e-8
for k, nm in enumerate(["code_run: S probes s", "code_run: same-slot+validate s", "code_run: avg loop s", "code_run: commit+rc s", "code_keys s"]):
    cn[nm] = st["timers"][27 + k] * 1e-8
for k, nm in enumerate(["scouts: stage P s", "scouts: early look-ups s", "scouts: sweeps s", "scouts: idle at the end s"]):
    cn[nm] = st["timers"][32 + k] * 1e-8
cn["scouts: chunks made"], cn["scouts: chunks given up"] = st["timers"][36], st["timers"][37]
tm = [x * 1e-8 for x in st["timers"][:10]]
print(f"{n} reads T={T}: wall {dt:.2f}s  {n*L/dt/1e6:.2f} Mbases/s  kernels: {kt}")
print("section seconds summed over workers:", {k: round(v, 3) for k, v in zip(names, tm)})
tot = tm[0] or 1
print("shares of worker time:", {k: round(v / tot, 3) for k, v in zip(names, tm)})
print("events:", cn)
print({k: v for k, v in st.items() if k != "timers"})
ib = st["timers"][40:46]
if ib[5]:
    print("global insert batches:", {"batches": ib[5], "rounds": ib[4], "probe walk s": round(ib[0] * 1e-8, 3), "clash test s": round(ib[1] * 1e-8, 3),
                                     "draws + stores s": round(ib[2] * 1e-8, 3), "store wait s": round(ib[3] * 1e-8, 3),
                                     "us per batch": round(sum(ib[:4]) * 1e-2 / ib[5], 2)})
sp = [st["timers"][i] * 1e-8 for i in (38, 39, 46, 47)]
if os.environ.get("FQSX_TIMING_MODELS"):
    print("code_run tail (models wave; build_timing.py models):", {"commit stores + wait s": round(sp[0], 2), "range-coder queue wait s": round(sp[1], 2), "r_sym loop s": round(sp[2], 2), "runs": st["timers"][47], "symbols per run": round(st["coded"] / max(st["timers"][47], 1), 1)})
print("stage P sections (wave clock at each lane's branch; summed over scouts) s:", {"roll": round(sp[0], 2), "global b probe": round(sp[1], 2), "hit: keys/rank/repair": round(sp[2], 2), "miss cascade": round(sp[3], 2)})
