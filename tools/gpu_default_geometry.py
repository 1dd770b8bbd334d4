"""Default geometry (-gs 3100, BASELINE configs[3]) on the metric's 1 M x 150 bp file: throughput, probe mix.
usage: python tools/gpu_default_geometry.py [T=64]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, read_id

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, L, G = 1000000, 150, 7500000
reads = synth_reads(n, L, G, 2)
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
out = {"reads": n, "len": L, "workers_T": T}
for gs in (3100, 8):
    header = hp.make_header(T, "se_sorted", gs)
    dev = []
    for idx in hp.form_blocks(rec, "se_sorted"):
        bases, off = hp.block_arrays(rec, idx)
        dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))
    t_c = time.perf_counter()
    c = DnaCodec(header, device=0)
    torch.cuda.synchronize()
    create_s = time.perf_counter() - t_c
    c.set_profiling(False)
    t0 = time.perf_counter()
    nbytes = 0
    for g, (d_b, d_o, off) in enumerate(dev):
        nbytes += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
    dt = time.perf_counter() - t0
    st = c.stats()
    c.close()
    probes = st["gprobe"] + st["lprobe"]
    slots = st["gslot"] + st["lslot"]
    out[f"gs{gs}"] = {"k": list(header[10:14]), "mbases_s": round(n * L / dt / 1e6, 3), "create_s": round(create_s, 2), "bits_per_base": round(8 * nbytes / (n * L), 5),
                      "probes_per_base": round(probes / st["bases"], 3), "slots_per_probe": round(slots / probes, 3),
                      "bytes_per_probe": round((24 * probes + 4 * slots) / probes, 2), "siv_words_per_base": round(st["siv_words"] / st["bases"], 2),
                      "global_inserts_per_base": round(st["gins"] / st["bases"], 3)}
print(json.dumps(out))
