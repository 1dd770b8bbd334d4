"""Throughput of the quality-stream kernel (row N1) on the benchmark's read set: 1M x 100 qualities, T workers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import QualCodec
from fqsqueezer_amd.synth import synth_reads, synth_quals, read_id
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1000000
reads = synth_reads(n, 100, 5000000, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, synth_quals(n, 100, 2))
blocks = hp.form_blocks(rec, "se_sorted")
for qm in ("illumina_8", "lossless"):
    header = hp.make_header(T, "se_sorted", 5, qm, "none")
    arrs = [hp.qual_arrays(rec, idx) for idx in blocks]
    c = QualCodec(header)
    t0 = time.time(); nb = 0
    for q, off in arrs:
        nb += sum(len(s) for s in c.encode_block(q, off))
    dt = time.time() - t0
    print(f"quality {qm} T={T}: {n*100/dt/1e6:.2f} Msymbols/s, {8*nb/(n*100):.3f} bits/symbol, wall {dt:.2f}s")
