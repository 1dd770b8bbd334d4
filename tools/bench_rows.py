#!/usr/bin/env python3
"""Throughput of the rows beside the headline path (SURVEY.md 8f and the other dna_modes) on one MI355X, inputs resident
in HBM where the entry point allows it: DNA decoder, quality coder (device-resident and host-buffer), paired-end encoder,
original-order encoder.  Called by bench.py (extras of the N=1 line: `other_rows`) and by tools/gpu_rows.sh under
rocprofv3 for the kernel statistics in profiles/.  Prints one JSON object."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def measure(n_reads: int = 300_000, read_len: int = 150, T: int = 64, device: int = 0) -> dict:
    import torch
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec, QualCodec
    from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals, synth_reads

    G, gs = n_reads * read_len // 20, max(1, n_reads * read_len // 20 // 1_000_000)
    reads = synth_reads(n_reads, read_len, G, 2)
    quals = synth_quals(n_reads, read_len, 2)
    rec = hp.Records([read_id(i) for i in range(n_reads)], reads, quals)
    n_bases = n_reads * read_len
    out = {"reads": n_reads, "len": read_len, "workers_T": T, "genome": G}

    def dev(bases, off):
        return torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda()

    # ---- DNA decoder (k_decode_*): encode the file once, then time decoding it (streams host -> device per block)
    header = hp.make_header(T, "se_sorted", gs)
    blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
    enc = DnaCodec(header, device=device)
    streams = [enc.encode_block(b, o, g) for g, (b, o) in enumerate(blocks)]
    enc.close()
    dec = DnaCodec(header, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = True
    for g, (b, o) in enumerate(blocks):
        got = dec.decode_block(streams[g], o, g)
        ok = ok and bool(np.array_equal(got, np.asarray(b)))
    dt = time.perf_counter() - t0
    dec.close()
    out["decode_mbases_s"] = round(n_bases / dt / 1e6, 3)
    out["decode_round_trip_ok"] = ok

    # ---- encoders of the other modes, blocks resident in HBM
    def encode_rate(header, dblocks, nb):
        c = DnaCodec(header, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nbytes = 0
        for g, (d_b, d_o, off) in enumerate(dblocks):
            nbytes += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        dt = time.perf_counter() - t0
        c.close()
        return round(nb / dt / 1e6, 3), round(8.0 * nbytes / nb, 5)

    db = [dev(b, o) + (o,) for (b, o) in blocks]
    out["sorted_mbases_s"], out["sorted_bits_per_base"] = encode_rate(header, db, n_bases)
    hdr_o = hp.make_header(T, "se_original", gs)
    db = [dev(*hp.block_arrays(rec, idx)) + (hp.block_arrays(rec, idx)[1],) for idx in hp.form_blocks(rec, "se_original")]
    out["original_order_mbases_s"], out["original_order_bits_per_base"] = encode_rate(hdr_o, db, n_bases)

    n_pairs = n_reads // 2
    r1, r2 = synth_pairs(n_pairs, read_len, G, 3)
    rec1 = hp.Records([read_id(i, 1) for i in range(n_pairs)], r1, quals[:n_pairs])
    rec2 = hp.Records([read_id(i, 2) for i in range(n_pairs)], r2, quals[n_pairs:2 * n_pairs])
    hdr_p = hp.make_header(T, "pe_sorted", gs)
    db = []
    for idx in hp.form_blocks_pe(rec1, rec2, "pe_sorted"):
        b, o = hp.block_arrays_pe(rec1, rec2, idx)
        db.append(dev(b, o) + (o,))
    out["pe_sorted_mbases_s"], out["pe_sorted_bits_per_base"] = encode_rate(hdr_p, db, 2 * n_pairs * read_len)

    # ---- quality coder (k_qual_encode), iid 7-level qualities: lossless (-qm o) and Illumina-8 (-qm 8)
    for qm, tag in (("lossless", "quality_o"), ("illumina_8", "quality_8")):
        hq = hp.make_header(T, "se_sorted", gs, qm, "none")
        qb = [hp.qual_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
        dq = [dev(q, o) + (o,) for (q, o) in qb]
        c = QualCodec(hq, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nbytes = 0
        for d_q, d_o, off in dq:
            nbytes += c.encode_block_dev(d_q.data_ptr(), d_o.data_ptr(), off)
        dt = time.perf_counter() - t0
        c.close()
        out[tag + "_msym_s"] = round(n_bases / dt / 1e6, 3)
        out[tag + "_bits_per_symbol"] = round(8.0 * nbytes / n_bases, 4)
        c = QualCodec(hq, device=device)
        t0 = time.perf_counter()
        for q, o in qb:
            c.encode_block(q, o)
        out[tag + "_pcie_inclusive_msym_s"] = round(n_bases / (time.perf_counter() - t0) / 1e6, 3)
        c.close()
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    print(json.dumps(measure(n)))
