"""T=255 (the bitstream's maximum worker count): parity against the oracle on a prefix + throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id
from oracle.pyoracle import OracleCodec
T = int(sys.argv[1]) if len(sys.argv) > 1 else 255
n = 1000000
reads = synth_reads(n, 100, 5000000, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", 5)
blocks = hp.form_blocks(rec, "se_sorted")
g_, o_ = DnaCodec(header), OracleCodec(header)
for g, idx in enumerate(blocks[:6]):
    bases, off = hp.block_arrays(rec, idx)
    assert g_.encode_block(bases, off, g) == o_.encode_block(bases, off, g), g
print("parity vs oracle on 6 blocks: OK")
g_.close()
c = DnaCodec(header); c.set_profiling(True)
t0 = time.time(); nb = 0
for g, idx in enumerate(blocks):
    bases, off = hp.block_arrays(rec, idx)
    nb += sum(len(s) for s in c.encode_block(bases, off, g))
dt = time.time() - t0
print(f"T={T}: {n*100/dt/1e6:.2f} Mbases/s wall {dt:.2f}s bits/base {8*nb/(n*100):.4f} kernels {c.kernel_times()}")
