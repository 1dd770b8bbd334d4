"""Diagnostic: reads of novel sequence only (coverage << 1), few workers: which wavefront role bounds such reads?
-DFQSX_TIMING build; prints the resolving wave's and the coder wave's end-to-end times next to the kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reads = synth_reads(n, 100, 80000000, 5)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", 5)
order = np.concatenate(hp.sorted_order(rec))
c = DnaCodec(header, lib_path=(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tools", "libfqsx_timing.so")))
c.set_profiling(True)
B = 2000
for g, lo in enumerate(range(0, n, B)):
    bases, off = hp.block_arrays(rec, order[lo:lo + B])
    c.encode_block(bases, off, g + 100)   # generation >= 100: one segment per block
st = c.stats(); kt = c.kernel_times()
tm = st["timers"]
print({"reads": n, "T": T, "encode_kernel_s_x_T": round(kt["encode_ms"] * 1e-3 * T, 4), "resolve_total_s": round(tm[0] * 1e-8, 4),
       "coder_end_s": round(tm[23] * 1e-8, 4), "coder_busy_s": round(tm[2] * 1e-8, 4), "spec_wait_s": round(tm[1] * 1e-8, 4),
       "slow_s": round(tm[3] * 1e-8, 4), "rough_incl_frontier_wait_s": round(tm[7] * 1e-8, 4), "find_counts_s": round(tm[9] * 1e-8, 4),
       "lq_flush_s": round(tm[6] * 1e-8, 4), "resolve_counts_s": round(tm[26] * 1e-8, 4), "pushes_repairs_s": round(tm[25] * 1e-8, 4), "n_slow": tm[11], "n_rough": tm[14], "n_generic": tm[17], "n_conflict": tm[19], "n_chunk": tm[12], "n_dirty": tm[13], "code_one_calls": tm[22], "coded": st["coded"], "us_per_read_kernel": round(kt["encode_ms"] * 1e3 * T / n, 1)})
