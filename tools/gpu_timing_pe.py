"""Diagnostic: the -DFQSX_TIMING build over a paired-end file (150 k pairs x 150 bp, sorted), section times as tools/gpu_timing.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
mode = sys.argv[2] if len(sys.argv) > 2 else "pe_sorted"
T, L = 64, 150
G = 2 * n_pairs * L // 20
gs = max(1, G // 1_000_000)
r1, r2 = synth_pairs(n_pairs, L, G, 3)
q = synth_quals(2 * n_pairs, L, 2)
rec1 = hp.Records([read_id(i, 1) for i in range(n_pairs)], r1, q[:n_pairs])
rec2 = hp.Records([read_id(i, 2) for i in range(n_pairs)], r2, q[n_pairs:])
header = hp.make_header(T, mode, gs)
c = DnaCodec(header, lib_path=os.path.join(ROOT, "tools", "libfqsx_timing.so"))
c.set_profiling(True)
t0 = time.time()
nb = 0
for g, idx in enumerate(hp.form_blocks_pe(rec1, rec2, mode)):
    b, o = hp.block_arrays_pe(rec1, rec2, idx)
    c.encode_block(b, o, g)
    nb += int(o[-1])
dt = time.time() - t0
st = c.stats(); kt = c.kernel_times()
tm = st["timers"]
names = ["total", "spec", "fast", "slow", "post_q", "read_head", "lq_flush", "rough", "repair_missing", "find_counts"]
print(f"{n_pairs} pairs {mode} T={T}: wall {dt:.2f}s {nb/dt/1e6:.2f} Mbases/s kernels {kt}")
print("section s summed over workers:", {k: round(v * 1e-8, 3) for k, v in zip(names, tm[:10])})
print("counts:", dict(zip(["n_fast", "n_slow", "n_chunk", "n_dirty", "n_rough", "n_repm", "n_ext", "n_generic", "n_lqflush"], tm[10:19])))
print("more s:", {"rrwait": tm[19] * 1e-8, "cqwait": tm[20] * 1e-8, "coder total": tm[23] * 1e-8, "coder idle": tm[24] * 1e-8, "scout wait": tm[25] * 1e-8, "resolve counts": tm[26] * 1e-8,
                  "code_keys": tm[31] * 1e-8, "scouts stage P": tm[32] * 1e-8, "scouts early": tm[33] * 1e-8, "scouts sweeps": tm[34] * 1e-8, "scouts idle": tm[35] * 1e-8, "chunks made": tm[36], "given up": tm[37]})
print({k: v for k, v in st.items() if k != "timers"})
print("compress_pair steps (build_timing.py pe) s:", {"first mate's head": tm[5] * 1e-8, "minimizers + staging": tm[6] * 1e-8, "pair-table look-ups": tm[38] * 1e-8, "merge + search in mate 2": tm[39] * 1e-8,
                                                     "anchored: ids, copy, seed": tm[46] * 1e-8, "direct: second mate's head": tm[47] * 1e-8})
