"""Diagnostic: encode the first N blocks of the bench workload (product library); run under
`rocprofv3 --kernel-trace` to get the kernel timeline of the warm-up phases (tools/trace_gaps.py reads the CSV)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 70
reads = synth_reads(1000000, 150, 7500000, 2)
rec = hp.Records([read_id(i) for i in range(len(reads))], reads, reads)
header = hp.make_header(64, "se_sorted", 8)
blocks = hp.form_blocks(rec, "se_sorted")[:nblk]
db = []
for idx in blocks:
    b, o = hp.block_arrays(rec, idx)
    db.append((torch.from_numpy(np.ascontiguousarray(b)).cuda(), torch.from_numpy(o.view(np.int64)).cuda(), o))
for rep in range(2):
    c = DnaCodec(header)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for g, (d_b, d_o, off) in enumerate(db):
        c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        nb += int(off[-1])
    dt = time.perf_counter() - t0
    c.close()
    print(f"pass {rep}: {nblk} blocks, {nb/dt/1e6:.2f} Mbases/s, {dt*1e3:.1f} ms")
