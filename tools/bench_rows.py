#!/usr/bin/env python3
"""The rows beside the headline path (SURVEY.md 8f and the other dna_modes) on one MI355X, inputs resident in HBM where
the entry point allows it: DNA decoder, original-order encoder, paired-end encoder, quality coder, and one end-to-end
full-mode row (DNA + quality kernels side by side on two HIP streams, meta / id coders on host threads).

Every row carries
  * its rate (one timed pass, no profiling),
  * `roofline`: algorithmic bytes of the row's kernel (SURVEY.md 8d formula from the device counters; quality: context
    slot read + write per symbol) / the kernel's time, measured with HIP events around every launch on the codec's
    stream in a second pass, against 8 TB/s,
  * `cpu`: the unmodified reference (oracle/_ref/fqs-1.1) timed on this box on a bounded sample of the same data in the
    row's own modes (whole process, -t 8: its fastest setting on a many-core host); absent when the binary is.
Called by bench.py (extras of the N=1 line: `other_rows`) and by tools/gpu_rows.sh under rocprofv3 for the kernel
statistics in profiles/.  Prints one JSON object."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "fqs-1.1")
PEAK = 8000.0


def dna_bytes(st: dict, pairs: int = 0) -> float:
    """B = sum_probes(24 + 4 slots) + sum_global_inserts(24 + 4 slots + 4) + 8 siv_words + 24 ctx_slots + 80 coded;
    paired-end adds the pair table: 8 finds + 14 inserts of a 16-byte item behind a 24-byte descriptor per pair."""
    # (slots per cluster scan: the reference's measured 3.96 -- this library's two-choice buckets read eight slots per probe whatever they hold; bench.py)
    probes = st["gprobe"] + st["lprobe"]
    return ((24.0 + 4.0 * 3.96) * probes + (28.0 + 4.0 * 3.96) * st["gins"] + 8.0 * (st["siv_words"] + st.get("siv_saved", 0))
            + 24.0 * st["ctx_slots"] + 80.0 * st["coded"] + 22.0 * 40.0 * pairs)


def roofline(alg_bytes: float, kernel_ms: float, launches: int, kernel: str) -> dict:
    s = max(kernel_ms, 1e-9) / 1e3
    ach = alg_bytes / s / 1e9
    return {"bound": "hbm", "achieved": round(ach, 3), "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 6), "traffic": None,
            "kernel": kernel, "launches": launches, "avg_launch_ms": round(kernel_ms / max(1, launches), 4),
            "algorithmic_bytes_per_launch": round(alg_bytes / max(1, launches), 1)}


def ref_run(args, units: float, unit: str, what: str):
    """whole-process wall of one reference command"""
    if not os.path.exists(REF):
        return None
    t0 = time.perf_counter()
    subprocess.run([REF] + args, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    return {"value": round(units / dt / 1e6, 4), "unit": unit, "cores": 8, "kind": "reference", "wall_s": round(dt, 2), "sample": what}


def measure(n_reads: int = 300_000, read_len: int = 150, T: int = 64, device: int = 0, cpu_reads: int = 100_000, only=None) -> dict:
    """only: a set of row names (decode, sorted, original_order, pe_sorted, quality_o, quality_8, id_lossless, full_mode_pe_q8) -- the
    rocprofv3 --pmc passes run one kernel's rows at a time, so that a kernel's counters are one row's (tools/gpu_round.sh rows)"""
    def want(row):
        return only is None or row in only
    import torch
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec, QualCodec
    from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals, synth_reads, write_fastq

    G, gs = n_reads * read_len // 20, max(1, n_reads * read_len // 20 // 1_000_000)
    reads = synth_reads(n_reads, read_len, G, 2)
    quals = synth_quals(n_reads, read_len, 2)
    rec = hp.Records([read_id(i) for i in range(n_reads)], reads, quals)
    n_bases = n_reads * read_len
    out = {"reads": n_reads, "len": read_len, "workers_T": T, "genome": G, "gs": gs,
           "cpu_note": f"reference legs: first {cpu_reads} reads / pairs of the row's data, `fqs-1.1 ... -t 8 -gs {gs}`, whole-process wall on this box"}

    def dev(bases, off):
        return torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda()

    td = tempfile.mkdtemp(prefix="fqsx_rows_")
    tmp = os.path.join(td, "tmp_")
    nc = min(cpu_reads, n_reads)
    fq = os.path.join(td, "s.fq")
    have_ref = os.path.exists(REF)
    if have_ref:
        write_fastq(fq, reads[:nc], quals[:nc])

    # ---- DNA decoder (k_decode_*): encode the file once, then time decoding it (streams host -> device per block)
    header = hp.make_header(T, "se_sorted", gs)
    blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
    streams = []
    if want("decode"):
        enc = DnaCodec(header, device=device)
        streams = [enc.encode_block(b, o, g) for g, (b, o) in enumerate(blocks)]
        enc.close()

    def decode_pass(profile):
        dec = DnaCodec(header, device=device)
        dec.set_profiling(profile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = True
        for g, (b, o) in enumerate(blocks):
            got = dec.decode_block(streams[g], o, g)
            ok = ok and bool(np.array_equal(got, np.asarray(b)))
        dt = time.perf_counter() - t0
        st, kt = dec.stats(), dec.kernel_times()
        dec.close()
        return dt, ok, st, kt

    if want("decode"):
      dt, ok, _, _ = decode_pass(False)
      _, _, st, kt = decode_pass(True)
      out["decode"] = {"value": round(n_bases / dt / 1e6, 3), "unit": "Mbases/s", "round_trip_ok": ok,
                     "roofline": roofline(dna_bytes(st), kt["encode_ms"], kt["encode_launches"], "k_decode_se_sorted")}
      out["decode_mbases_s"], out["decode_round_trip_ok"] = out["decode"]["value"], ok
    if have_ref and want("decode"):
        f = os.path.join(td, "d.fqs")
        subprocess.run([REF, "e", "-s", "-om", "s", "-t", "8", "-gs", str(gs), "-qm", "n", "-im", "n", "-v", "0", "-tmp", tmp, "-out", f, fq],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out["decode"]["cpu"] = ref_run(["d", "-out", os.path.join(td, "d.fq"), f], nc * read_len, "Mbases/s", f"`fqs-1.1 d` of its own `-om s -t 8` file of {nc} reads")

    # ---- encoders of the other modes, blocks resident in HBM
    def encode_row(header, dblocks, nb, kernel, pairs=0):
        res = {}
        for profile in (False, True):
            c = DnaCodec(header, device=device)
            c.set_profiling(profile)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nbytes = 0
            for g, (d_b, d_o, off) in enumerate(dblocks):
                nbytes += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            dt = time.perf_counter() - t0
            if not profile:
                res = {"value": round(nb / dt / 1e6, 3), "unit": "Mbases/s", "bits_per_base": round(8.0 * nbytes / nb, 5)}
            else:
                st, kt = c.stats(), c.kernel_times()
                res["roofline"] = roofline(dna_bytes(st, pairs) - (28.0 + 4.0 * 3.96) * st["gins"], kt["encode_ms"], kt["encode_launches"], kernel)
            c.close()
        return res

    if want("sorted"):
        db = [dev(b, o) + (o,) for (b, o) in blocks]
        out["sorted"] = encode_row(header, db, n_bases, "k_encode_se_sorted")
        out["sorted_mbases_s"], out["sorted_bits_per_base"] = out["sorted"]["value"], out["sorted"]["bits_per_base"]
    hdr_o = hp.make_header(T, "se_original", gs)
    if want("original_order"):
        db = [dev(*hp.block_arrays(rec, idx)) + (hp.block_arrays(rec, idx)[1],) for idx in hp.form_blocks(rec, "se_original")]
        out["original_order"] = encode_row(hdr_o, db, n_bases, "k_encode_se_orig")
        out["original_order_mbases_s"], out["original_order_bits_per_base"] = out["original_order"]["value"], out["original_order"]["bits_per_base"]
    if have_ref and want("original_order"):
        out["original_order"]["cpu"] = ref_run(["e", "-s", "-om", "o", "-t", "8", "-gs", str(gs), "-qm", "n", "-im", "n", "-v", "0", "-tmp", tmp, "-out", os.path.join(td, "o.fqs"), fq],
                                               nc * read_len, "Mbases/s", f"`fqs-1.1 e -s -om o -qm n -im n -t 8`, {nc} reads")

    n_pairs = n_reads // 2
    r1, r2 = synth_pairs(n_pairs, read_len, G, 3)
    rec1 = hp.Records([read_id(i, 1) for i in range(n_pairs)], r1, quals[:n_pairs])
    rec2 = hp.Records([read_id(i, 2) for i in range(n_pairs)], r2, quals[n_pairs:2 * n_pairs])
    hdr_p = hp.make_header(T, "pe_sorted", gs)
    db = []
    if want("pe_sorted"):
        for idx in hp.form_blocks_pe(rec1, rec2, "pe_sorted"):
            b, o = hp.block_arrays_pe(rec1, rec2, idx)
            db.append(dev(b, o) + (o,))
        out["pe_sorted"] = encode_row(hdr_p, db, 2 * n_pairs * read_len, "k_encode_pe_sorted", pairs=n_pairs)
        out["pe_sorted_mbases_s"], out["pe_sorted_bits_per_base"] = out["pe_sorted"]["value"], out["pe_sorted"]["bits_per_base"]
    npc = min(cpu_reads // 2, n_pairs)
    f1, f2 = os.path.join(td, "p1.fq"), os.path.join(td, "p2.fq")
    if have_ref and (want("pe_sorted") or want("full_mode_pe_q8")):
        write_fastq(f1, r1[:npc], quals[:npc], mate=1)
        write_fastq(f2, r2[:npc], quals[n_pairs:n_pairs + npc], mate=2)
    if have_ref and want("pe_sorted"):
        out["pe_sorted"]["cpu"] = ref_run(["e", "-p", "-om", "s", "-t", "8", "-gs", str(gs), "-qm", "n", "-im", "n", "-v", "0", "-tmp", tmp, "-out", os.path.join(td, "p.fqs"), f1, f2],
                                          2 * npc * read_len, "Mbases/s", f"`fqs-1.1 e -p -om s -qm n -im n -t 8`, {npc} pairs")

    # ---- quality coder (k_qual_encode), iid 7-level qualities: lossless (-qm o) and Illumina-8 (-qm 8)
    for qm, tag, flag in (("lossless", "quality_o", "o"), ("illumina_8", "quality_8", "8")):
        if not want(tag):
            continue
        hq = hp.make_header(T, "se_sorted", gs, qm, "none")
        qb = [hp.qual_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")]
        dq = [dev(q, o) + (o,) for (q, o) in qb]
        row = {}
        for profile in (False, True):
            c = QualCodec(hq, device=device)
            c.set_profiling(profile)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nbytes = 0
            for d_q, d_o, off in dq:
                nbytes += c.encode_block_dev(d_q.data_ptr(), d_o.data_ptr(), off)
            dt = time.perf_counter() - t0
            if not profile:
                row = {"value": round(n_bases / dt / 1e6, 3), "unit": "Msymbols/s", "bits_per_symbol": round(8.0 * nbytes / n_bases, 4)}
            else:
                kt = c.kernel_times()
                n_sym = {"lossless": 96, "illumina_8": 8}[qm]
                slot = 8 * (1 + (n_sym + 1 + 3) // 4)      # key word + packed u16 statistics and total (csrc/fqsx_qual.h)
                row["roofline"] = roofline(2.0 * slot * n_bases, kt["encode_ms"], kt["encode_launches"], "k_qual_encode")
                row["roofline"]["algorithmic_bytes_per_symbol"] = 2 * slot
            c.close()
        c = QualCodec(hq, device=device)
        t0 = time.perf_counter()
        for q, o in qb:
            c.encode_block(q, o)
        row["pcie_inclusive_msym_s"] = round(n_bases / (time.perf_counter() - t0) / 1e6, 3)
        c.close()
        if have_ref:
            row["cpu"] = ref_run(["e", "-s", "-om", "s", "-t", "8", "-gs", str(gs), "-qm", flag, "-im", "n", "-v", "0", "-tmp", tmp, "-out", os.path.join(td, "q.fqs"), fq],
                                 nc * read_len, "Msymbols/s", f"`fqs-1.1 e -s -om s -qm {flag} -im n -t 8`, {nc} reads: WHOLE process (its DNA and quality coders run in one loop; the DNA-only run is the `sorted` row's)")
        out[tag] = row
        out[tag + "_msym_s"], out[tag + "_bits_per_symbol"], out[tag + "_pcie_inclusive_msym_s"] = row["value"], row["bits_per_symbol"], row["pcie_inclusive_msym_s"]

    # ---- read-id coder (k_id_encode, row N4): lossless ids of the generator ("@SRR000001.<i> <i>/1"), GPU kernel vs the host coder
    # (one host thread per worker) on the same blocks
    from fqsqueezer_amd.codec import IdCodec
    hi = hp.make_header(T, "se_sorted", gs, "none", "lossless")
    ib = [hp.id_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")] if want("id_lossless") else []
    id_bytes = sum(int(o[-1]) for (_, o) in ib)
    row = {}
    for tag2, devarg in ((("gpu", device), ("host_threads", None)) if want("id_lossless") else ()):
        c = IdCodec(hi, device=devarg)
        t0 = time.perf_counter()
        nbytes = sum(sum(len(x) for x in c.encode_block(a, o, False)) for (a, o) in ib)
        dt = time.perf_counter() - t0
        c.close()
        row[tag2] = {"value": round(n_reads / dt / 1e6, 3), "unit": "Mids/s", "id_mbytes_s": round(id_bytes / dt / 1e6, 2), "stream_bytes": nbytes}
    if want("id_lossless"):
        row["identical_streams"] = row["gpu"]["stream_bytes"] == row["host_threads"]["stream_bytes"]
        row["note"] = "host buffers in, streams out (the upload is inside); the DNA path codes 0.67 Mreads/s of 150 bp at 100 Mbases/s, so either id coder runs beside it"
        out["id_lossless"] = row

    # ---- full mode end to end (BASELINE configs[2]'s modes: -p -om s -qm 8, default -im i): the DNA and the quality kernels on
    # two HIP streams, the id kernel on a third, the meta coder on a host thread (fqsfile.encode_blocks), host buffers in, container blocks out
    from fqsqueezer_amd.fqsfile import compress_records_pe
    if want("full_mode_pe_q8"):
        t0 = time.perf_counter()
        hdr_f, blks = compress_records_pe(rec1, rec2, T, "s", gs, device=device, quality_mode="illumina_8", id_mode="instrument", as_blocks=True)
        n_out = sum(len(ch) for ch in hp.fqs_chunks(hdr_f, blks))
        dt = time.perf_counter() - t0
        out["full_mode_pe_q8"] = {"value": round(2 * n_pairs * read_len / dt / 1e6, 3), "unit": "Mbases/s", "file_bytes": n_out,
                                  "note": "whole file from host buffers, GPU sort pre-pass included: 4 streams per block, DNA + quality + id kernels concurrently, meta on a host thread"}
    if have_ref and want("full_mode_pe_q8"):
        out["full_mode_pe_q8"]["cpu"] = ref_run(["e", "-p", "-om", "s", "-t", "8", "-gs", str(gs), "-qm", "8", "-v", "0", "-tmp", tmp, "-out", os.path.join(td, "f.fqs"), f1, f2],
                                                2 * npc * read_len, "Mbases/s", f"`fqs-1.1 e -p -om s -qm 8 -t 8` (default -im i), {npc} pairs")
    import shutil
    shutil.rmtree(td, ignore_errors=True)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    print(json.dumps(measure(n, only=set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None)))
