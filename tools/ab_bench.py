#!/usr/bin/env python3
"""A/B of product-library builds on ONE GPU box in ONE call (boxes differ by a few per cent): the bench workload
(1 M x 150 bp SE sorted, T = 64, inputs resident in HBM) through every library given, alternating, `reps` passes each.
usage: python tools/ab_bench.py [--reps 2] [--gs 8] [--reads N] libA.so libB.so ...   (paths relative to the repo root)"""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import read_id, synth_reads

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--len", type=int, default=150)
ap.add_argument("--genome", type=int, default=7_500_000)
ap.add_argument("--gs", type=int, default=8)
ap.add_argument("--threads", type=int, default=64)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
reads = synth_reads(a.reads, a.len, a.genome, 2)
rec = hp.Records([read_id(i) for i in range(a.reads)], reads, reads)
header = hp.make_header(a.threads, "se_sorted", a.gs)
dev = []
for idx in hp.form_blocks(rec, "se_sorted"):
    bases, off = hp.block_arrays(rec, idx)
    dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))
torch.cuda.synchronize()
res = {l: [] for l in a.libs}
for rep in range(a.reps + 1):          # (pass 0 of every library is its warm-up)
    for l in a.libs:
        c = DnaCodec(header, device=0, lib_path=os.path.join(ROOT, l))
        t = [time.perf_counter()]
        nb = 0
        for g, (d_b, d_o, off) in enumerate(dev):
            nb += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            t.append(time.perf_counter())
        c.close()
        if rep:
            tot = a.reads * a.len / (t[-1] - t[0]) / 1e6
            warm = sum(int(o[-1]) for (_, _, o) in dev[:70]) / (t[70] - t[0]) / 1e6 if len(dev) > 110 else 0
            steady = sum(int(o[-1]) for (_, _, o) in dev[100:]) / (t[-1] - t[100]) / 1e6 if len(dev) > 110 else 0
            res[l].append((round(tot, 2), round(warm, 2), round(steady, 2), nb))
for l in a.libs:
    best = max(res[l])
    print(json.dumps({"lib": l, "file_mbases_s": best[0], "blocks_0_69": max(x[1] for x in res[l]), "blocks_ge_100": max(x[2] for x in res[l]), "dna_bytes": best[3], "passes": res[l]}))
