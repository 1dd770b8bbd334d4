"""Diagnostic: per-block wall time of the bench workload (each block call ends with a D2H of its streams, so the host
clock around the call is the block's GPU time).  Blocks 0-69 run 30 synchronisation segments each (2 reads per worker
and segment), blocks 70-99 fewer and fewer, blocks >= 100 a single one (application.h:85-92)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
G = int(sys.argv[3]) if len(sys.argv) > 3 else 5000000
gs = int(sys.argv[4]) if len(sys.argv) > 4 else 5
T = int(sys.argv[5]) if len(sys.argv) > 5 else 64
reads = synth_reads(n, L, G, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", gs)
blocks = hp.form_blocks(rec, "se_sorted")
dev = []
for idx in blocks:
    bases, off = hp.block_arrays(rec, idx)
    dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off, len(idx), int(off[-1])))
for rep in range(2):
    c = DnaCodec(header, device=0)
    times = []
    torch.cuda.synchronize()
    for g, (d_b, d_o, off, nr, nb) in enumerate(dev):
        t0 = time.perf_counter()
        c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        times.append(time.perf_counter() - t0)
    c.close()
tb = np.array(times); bb = np.array([x[4] for x in dev], dtype=np.float64)
def rate(a, b):
    return float(bb[a:b].sum() / tb[a:b].sum() / 1e6) if b > a else None
nb = len(dev)
print({"reads": n, "len": L, "T": T, "blocks": nb, "total_mbases_s": round(rate(0, nb), 2),
       "blocks_0_69": round(rate(0, min(70, nb)), 2), "blocks_70_99": round(rate(70, min(100, nb)), 2) if nb > 70 else None,
       "blocks_100_up": round(rate(100, nb), 2) if nb > 100 else None,
       "share_of_time_0_69": round(float(tb[:70].sum() / tb.sum()), 3), "share_of_time_100_up": round(float(tb[100:].sum() / tb.sum()), 3) if nb > 100 else None})
