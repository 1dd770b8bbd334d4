"""debug: c4 ragged reads, T=64, sorted via the GPU pre-pass: GPU DNA streams vs the oracle, block by block"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import c4_records
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec, sort_order
from oracle.pyoracle import OracleCodec
rec = c4_records()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
header = hp.make_header(T, "se_sorted", 1)
bases, off = hp.block_arrays(rec, np.arange(len(rec), dtype=np.int64))
groups = sort_order(bases, off, device=0)
blocks = hp.form_blocks(rec, "se_sorted", groups=groups)
print("blocks", len(blocks), [len(b) for b in blocks][:10])
lib = sys.argv[2] if len(sys.argv) > 2 else None
a, b = DnaCodec(header, device=0, lib_path=lib), OracleCodec(header)
for g, idx in enumerate(blocks):
    bs, of = hp.block_arrays(rec, idx)
    t0 = time.time()
    x = a.encode_block(bs, of, g)
    dt = time.time() - t0
    y = b.encode_block(bs, of, g)
    bad = [w for w in range(T) if x[w] != y[w]]
    print("block", g, "reads", len(idx), "time %.3f" % dt, "differing workers", bad[:10])
    if bad:
        break
