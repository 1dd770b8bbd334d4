"""Diagnostic: the -DFQSX_TIMING build's section times of the DECODER (one wave per worker): encodes the first N reads of the
bench workload with the product library, decodes them with tools/libfqsx_timing.so, prints seconds summed over workers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
T = 64
reads = synth_reads(n, 150, n * 150 // 20, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
header = hp.make_header(T, "se_sorted", max(1, n * 150 // 20 // 1000000))
blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
enc = DnaCodec(header)
streams = [enc.encode_block(b, o, g) for g, (b, o) in enumerate(blocks)]
enc.close()
dec = DnaCodec(header, lib_path=os.path.join(ROOT, "tools", "libfqsx_timing.so"))
dec.set_profiling(True)
t0 = time.time()
for g, (b, o) in enumerate(blocks):
    assert np.array_equal(dec.decode_block(streams[g], o, g), np.asarray(b))
dt = time.time() - t0
st, kt = dec.stats(), dec.kernel_times()
tm = st["timers"]
names = {0: "total", 9: "find_counts (+ next cluster request)", 7: "rough / level none", 31: "ctx keys", 27: "level search (parallel levels)", 30: "decode + un_rank", 4: "state update, pushes, repairs"}
print(f"{n} reads T={T}: wall {dt:.2f}s {n*150/dt/1e6:.2f} Mbases/s kernels {kt}")
print({v: round(tm[k] * 1e-8, 3) for k, v in names.items()})
print({k: v for k, v in st.items() if k != "timers"})
