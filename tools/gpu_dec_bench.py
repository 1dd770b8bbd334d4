"""Decoder rate with the product library: encodes the first N reads of the bench workload, decodes them (round trip checked),
prints Mbases/s.  usage: python tools/gpu_dec_bench.py [n_reads=300000] [lib.so ...]  (several libraries: alternating passes)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
libs = sys.argv[2:] or [None]
T = 64
reads = synth_reads(n, 150, n * 150 // 20, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", max(1, n * 150 // 20 // 1000000))
blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
enc = DnaCodec(header)
streams = [enc.encode_block(b, o, g) for g, (b, o) in enumerate(blocks)]
enc.close()
res = {str(l): [] for l in libs}
for rep in range(2 if len(libs) > 1 else 1):
    for l in libs:
        dec = DnaCodec(header, lib_path=os.path.join(ROOT, l) if l else None)
        t0 = time.time()
        ok = True
        for g, (b, o) in enumerate(blocks):
            ok = ok and bool(np.array_equal(dec.decode_block(streams[g], o, g), np.asarray(b)))
        dt = time.time() - t0
        dec.close()
        res[str(l)].append((round(n * 150 / dt / 1e6, 3), ok))
print(json.dumps({"reads": n, "T": T, "decode_mbases_s": res}))
