#!/usr/bin/env python3
"""Size-independent property at capacity scale: encode a large synthetic file on the GPU, decode every block again on the GPU
(the decoder rebuilds all tables from the streams alone) and compare with the input.
usage: python tools/gpu_roundtrip_big.py [reads=30000000] [genome=900000000] [gs=900] [decode_blocks=all]
decode_blocks = n: only the file's first n blocks are decoded again (the decoder runs at a tenth of the encoder's rate)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec, sort_order
from fqsqueezer_amd.synth import synth_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 900_000_000
gs = int(sys.argv[3]) if len(sys.argv) > 3 else 900
n_dec = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
L, T = 150, 64
t0 = time.time()
reads = synth_reads(n, L, G, 2)
rec = hp.Records([b""] * 0, reads, reads)
print(f"reads generated {time.time() - t0:.0f}s", flush=True)
groups = sort_order(reads.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
order = np.concatenate(groups)
per = -(-n // 256)
blocks = [order[i:i + per] for i in range(0, n, per)]          # 256 blocks of the sorted order (block formation is the host's choice)
print(f"sorted {time.time() - t0:.0f}s, {len(blocks)} blocks", flush=True)
header = hp.make_header(T, "se_sorted", gs)
off = np.arange(per + 1, dtype=np.uint64) * np.uint64(L)
enc = DnaCodec(header, device=0)
streams, n_bytes = [], 0
t1 = time.time()
for g, idx in enumerate(blocks):
    bases = np.ascontiguousarray(reads[idx]).reshape(-1)
    s = enc.encode_block(bases, off[:len(idx) + 1], g)
    streams.append(s)
    n_bytes += sum(len(x) for x in s)
    if g % 32 == 31:
        print(f"encoded block {g} {time.time() - t1:.0f}s", flush=True)
t_enc = time.time() - t1
cap = enc.capacity()
enc.close()
dec = DnaCodec(header, device=0)
t2 = time.time()
n_dec_bases = 0
for g, idx in enumerate(blocks[:n_dec]):
    bases = np.ascontiguousarray(reads[idx]).reshape(-1)
    n_dec_bases += len(bases)
    out = dec.decode_block(streams[g], off[:len(idx) + 1], g)
    assert np.array_equal(out, bases), f"block {g} did not round-trip"
    if g % 8 == 7:
        print(f"decoded block {g} {time.time() - t2:.0f}s", flush=True)
t_dec = time.time() - t2
print(json.dumps({"reads": n, "len": L, "genome": G, "gs": gs, "workers_T": T, "round_trip_ok": True, "dna_bytes": n_bytes, "bits_per_base": round(8.0 * n_bytes / (n * L), 5),
                  "encode_mbases_s_host_buffers": round(n * L / t_enc / 1e6, 2), "decoded_blocks": min(n_dec, len(blocks)), "decode_mbases_s": round(n_dec_bases / t_dec / 1e6, 2),
                  "smer_slots": cap["smer_slots"], "bmer_slots": cap["bmer_slots"], "bytes_per_bmer": cap["bytes_per_bmer"], "growths": cap["growths"],
                  "bmers": cap["bmers"], "smers": cap["smers"], "table_bytes": cap["table_bytes_held"], "device_bytes_peak": cap["device_bytes_peak"]}))
