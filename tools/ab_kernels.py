#!/usr/bin/env python3
"""Per-kernel times (HIP events around every launch) and probe counters of several builds on the bench workload, one pass each.
usage: python tools/ab_kernels.py libA.so libB.so ...   (paths relative to the repo root)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import read_id, synth_reads
reads = synth_reads(1_000_000, 150, 7_500_000, 2)
rec = hp.Records([read_id(i) for i in range(len(reads))], reads, reads)
header = hp.make_header(64, "se_sorted", 8)
dev = []
for idx in hp.form_blocks(rec, "se_sorted"):
    bases, off = hp.block_arrays(rec, idx)
    dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))
for l in sys.argv[1:]:
    for rep in range(2):
        c = DnaCodec(header, device=0, lib_path=os.path.join(ROOT, l))
        c.set_profiling(rep == 1)
        marks = {}
        for g, (d_b, d_o, off) in enumerate(dev):
            if g in (70, 100):
                marks[g] = (c.kernel_times(), c.stats())
            c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        kt, st, cap = c.kernel_times(), c.stats(), c.capacity()
        c.close()
    def d(a, b): return {k: round(a[k] - b[k], 2) for k in a if isinstance(a[k], (int, float))}
    print(json.dumps({"lib": l, "kernels_file": kt, "kernels_blocks_0_69": marks[70][0], "kernels_blocks_ge_100": d(kt, marks[100][0]),
                      "slots_per_gprobe": round(st["gslot"] / st["gprobe"], 3), "slots_per_lprobe": round(st["lslot"] / max(1, st["lprobe"]), 3),
                      "slots_per_ginsert": round(st["gins_slot"] / st["gins"], 3), "growths": cap["growths"], "bytes_per_bmer": cap["bytes_per_bmer"]}))
