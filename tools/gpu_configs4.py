#!/usr/bin/env python3
"""BASELINE configs[4] as a property run: N pairs x 150 bp, `-p -om s -qm o -im o` (lossless qualities and ids), T = 64, all four
streams on the GPU (DNA, quality and id kernels side by side on their own streams, meta on a host thread), host buffers in,
container blocks out.  Reports rates, stream sizes, the device memory in use at the end of the file (DNA tables / everything:
the rest is the quality and id context tables and the block buffers), and decodes the DNA of the first blocks again.
usage: python tools/gpu_configs4.py [pairs=10000000] [decode_blocks=4]
(configs[4] names 100 M pairs: what bounds a run here is the host memory of the box -- bases, qualities and ids of both mates
are held as numpy arrays / a list of bytes objects -- and the 20 minutes of a gpurun call; both are printed.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, psutil
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.fqsfile import compress_records_pe
from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_dec = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L, T = 150, 64
G = n * 2 * L // 20                      # 20x coverage
gs = max(1, G // 1_000_000)
t0 = time.time()
vm = psutil.virtual_memory()
print(f"host memory {vm.total / 2**30:.0f} GiB total, {vm.available / 2**30:.0f} GiB available", flush=True)
r1, r2 = synth_pairs(n, L, G, 5)
rec1 = hp.Records([read_id(i, 1) for i in range(n)], r1, synth_quals(n, L, 5))
rec2 = hp.Records([read_id(i, 2) for i in range(n)], r2, synth_quals(n, L, 6))
print(f"input generated {time.time() - t0:.0f}s", flush=True)
stats = {}
t1 = time.time()
header, blocks = compress_records_pe(rec1, rec2, T, "s", gs, quality_mode="lossless", id_mode="lossless", as_blocks=True, stats=stats)
sizes = {hp.STREAM_META: 0, hp.STREAM_ID: 0, hp.STREAM_DNA: 0, hp.STREAM_QUALITY: 0}
kept, n_blocks = [], 0
for g, b in enumerate(blocks):
    if g == 0:
        t_first = time.time()
    for st in b.streams:
        for k, v in st.items():
            sizes[k] += len(v)
    if g < n_dec:
        kept.append((b.n_reads, [st[hp.STREAM_DNA] for st in b.streams]))
    n_blocks += 1
    if g % 64 == 63:
        print(f"block {g} {time.time() - t1:.0f}s", flush=True)
t_enc = time.time() - t1
cap = stats.get("dna_capacity", {})
# ---- decode the first blocks' DNA again (the decoder needs the read lengths only: all 150 here)
dec = DnaCodec(header, device=0)
ok = True
from fqsqueezer_amd.fqsfile import _gpu_groups
groups = _gpu_groups(rec1, 0, None)
blks = hp.form_blocks_pe(rec1, rec2, "pe_sorted", groups=groups)
for g, (n_reads, streams) in enumerate(kept):
    bases, off = hp.block_arrays_pe(rec1, rec2, blks[g])
    assert len(off) - 1 == n_reads
    out = dec.decode_block(streams, off, g)
    ok = ok and bool(np.array_equal(out, np.asarray(bases)))
dec.close()
nb = 2 * n * L
names = {hp.STREAM_META: "meta", hp.STREAM_ID: "id", hp.STREAM_DNA: "dna", hp.STREAM_QUALITY: "quality"}
print(json.dumps({"config": f"{n} pairs x {L} bp, G={G} (20x), -p -om s -qm o -im o -gs {gs}, T={T}", "blocks": n_blocks, "round_trip_ok_first_blocks": ok, "decoded_blocks": len(kept),
                  "mbases_s_whole_file_host_buffers": round(nb / t_enc / 1e6, 2), "encode_s": round(t_enc, 1), "sort_and_first_block_s": round(t_first - t1, 1),
                  "stream_bytes": {names[k]: v for k, v in sizes.items()}, "bits_per_base_dna": round(8.0 * sizes[hp.STREAM_DNA] / nb, 4),
                  "bits_per_quality": round(8.0 * sizes[hp.STREAM_QUALITY] / nb, 4), "bits_per_id": round(8.0 * sizes[hp.STREAM_ID] / (2 * n), 3),
                  "device_bytes_in_use_at_end": stats.get("device_bytes_in_use"), "dna_device_bytes": cap.get("device_bytes"), "dna_device_bytes_peak": cap.get("device_bytes_peak"),
                  "quality_id_tables_and_buffers_bytes": (stats.get("device_bytes_in_use", 0) - cap.get("device_bytes", 0)) if cap else None,
                  "dna_capacity": cap, "host_memory_gib": round(vm.total / 2**30)}))
