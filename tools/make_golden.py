#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference binary
(oracle/_ref/fqs-1.1, built by `make -C oracle ref` from /root/reference).

Fixtures (data only -- inputs are re-generated deterministically by fqsqueezer_amd.synth):
  c1_10k_{o,s}_t{1,4}.fqs   complete reference outputs, 10k x 100bp, G=200kbp, seed 1, -gs 1
  c2_1M_s_t{1,8,64}.json    per-block SHA-256 + sizes of the reference DNA streams,
                            1M x 100bp, G=5Mbp, seed 2, -om s -gs 5   (BASELINE configs[1])
  c4_ragged_{o,s}_t3.fqs   3000 ragged reads (30-160 bp, N runs, duplicates), G=60kbp, seed 4, -gs 1
  c5_pe4k_{o,s}_t{1,4}.fqs  4000 pairs x 100bp (fragments 300-600), G=60kbp, seed 5, `-p`, -gs 1 (paired-end path)
  c6_20k_gs300_s_t2.json    20k x 100bp, G=1Mbp, seed 6, -gs 300 (k = 12/17/21/26: 4 GiB p-mer vector, 256-way partial look-ups)
  c9_20k150_gs3100_s_t2.json  20k x 150bp, G=200kbp, seed 9, DEFAULT -gs 3100 (k = 13/18/21/27: 16 GiB p-mer vector, up to 1024-way
                            partial look-ups; the reference needs ~45 GiB and ~10 min for it)
  c20_pelong_{o,s}_t2.fqs   240 pairs x 5000bp (fragments 6000-9000), G=40kbp, seed 21, `-p`, -gs 1 (mates beyond the LDS staging size)
  c7_mixedlen_{o,s}_t3.fqs  1500 reads of 20-31 / 60-199 / 4200-5999 bp with N runs, G=40kbp, seed 7, -gs 1
  c8_qual_{o,8,4,2}_t4.json, c8_qual_pe8_t3.json   per-block SHA-256 of the QUALITY streams (all four quality modes; PE)
  c10_full_*.fqs/.json     3000 x 100bp (G=80kbp, seed 10) with varied Illumina-style ids: complete default-mode files
                            (-qm o -im o) and digests of all four streams for -om s with -im i -qm 8 / -im o -qm o
  c11_pe_full_*.json        2000 pairs (G=60kbp, seed 11), varied ids incl. typical and atypical mate ids, -p, full modes
  c3_50k150_s_t8.json       50k x 150bp, G=250kbp, seed 3, -om s -gs 8 (150 bp metric shape)
  c12_1M150_s_t{8,64}.json  1M x 150bp, G=7.5Mbp, seed 2, -om s -gs 8: the workload BASELINE.json's metric is quoted on
                            (bench.py's default), per-block SHA-256 of the reference DNA streams
  c13_sat_{o,s}_t4.json     counter saturation: 300k (-om o, G=2.5kbp) / 500k (-om s, G=2kbp) x 100bp on a two-haplotype genome at
                            ~12000x coverage (synth_two_haplotypes, seed 13): counts_level_t::mixed, bmer_unc, cinc_b / cinc_s above
                            their thresholds (asserted through the oracle's fqo_levels)
  c14_pe150_s_q8_t8.json    100k pairs x 150bp, G=1.5Mbp, seed 14, -p -om s -qm 8 -im n -gs 2 (BASELINE configs[2]'s shape): all streams
  c15_pe150_s_oo_t4.json    20k pairs x 150bp, G=300kbp, seed 15, varied ids, -p -om s -qm o -im o -gs 1 (configs[4]'s modes): all streams
  c16_1M150_s_ids_t8.json   the c12 input with -om s -im o -qm n -t 8: meta / id / DNA digests -- pins the sorted read ORDER
                            (incl. the order of equal reads, which only the id stream sees) at 1 M reads
  c17_1M150_gs3100_s_t8.json  the c12 input at the DEFAULT geometry -gs 3100 (k = 13/18/21/27; BASELINE configs[3]'s geometry on a 1 M-read
                            prefix-sized file): DNA digests; the reference needs ~45 GiB and ~10 min (only with --only c17)
  c18_pe5M_s_q8_t8.json     BASELINE configs[2] at FULL size: 5 M pairs x 150bp (fragments 300-600), G=75Mbp, seed 3, -p -om s -qm 8 -gs 75
                            (default -im i), T=8: every stream of every block (only with --only c18; the reference needs ~30-60 min)
  c19_10M150_gs300_s_t{8,64}.json  a configs[3]-shaped SE file: 10 M x 150bp, G=300Mbp, seed 19, -om s -gs 300 (k = 12/17/21/26; tables far
                            beyond the Infinity Cache): DNA digests per block (only with --only c19 / c19t64; ~15 / ~45 GiB, 30-60 min)
  c21_5M150_G3100_gs3100_s_t8.json  SURVEY 8d-4: a prefix of configs[3]'s own file -- 5 M x 150bp from a G = 3.1 Gbp genome, seed 21,
                            -om s at the DEFAULT -gs 3100, T = 8: DNA digests per block (only with --only c21; ~50 GiB, about an hour)
  c22_pe1M_s_oo_t8.json     BASELINE configs[4]'s modes at 1 M pairs: 150bp mates, G=15Mbp, seed 22, varied ids, -p -om s -qm o -im o -gs 15, T=8:
                            all four streams of every block + the file's SHA-256 (only with --only c22; ~10 min)
Usage: python tools/make_golden.py [--work /tmp/w] [--only c1|c2|c3]
"""
import argparse, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.synth import synth_reads, write_fastq

REF = os.path.join(ROOT, "oracle", "_ref", "fqs-1.1")
GOLD = os.path.join(ROOT, "tests", "golden")


def run_ref(fq, out, om, t, gs, work):
    if not os.path.exists(out):
        subprocess.check_call([REF, "e", "-s", "-om", om, "-t", str(t), "-gs", str(gs), "-qm", "n", "-im", "n", "-v", "0",
                               "-tmp", os.path.join(work, "tmp_%d_" % os.getpid()), "-out", out, fq], stdout=subprocess.DEVNULL)


def digest(fqs_path, meta):
    header, blocks = hp.parse_fqs(open(fqs_path, "rb").read())
    d = dict(meta, header=header.hex(), n_blocks=len(blocks), blocks=[])
    total = 0
    for b in blocks:
        h = hashlib.sha256()
        sizes = []
        for st in b.streams:
            h.update(st[hp.STREAM_DNA])
            sizes.append(len(st[hp.STREAM_DNA]))
        total += sum(sizes)
        d["blocks"].append({"n_reads": b.n_reads, "sha256": h.hexdigest(), "bytes": sum(sizes)})
    d["dna_bytes"] = total
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/w")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.work, exist_ok=True)
    if a.only in ("", "c1"):
        fq = os.path.join(a.work, "c1.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_reads(10000, 100, 200000, 1), seed=1)
        for om in "os":
            for t in (1, 4):
                run_ref(fq, os.path.join(GOLD, f"c1_10k_{om}_t{t}.fqs"), om, t, 1, a.work)
    if a.only in ("", "c2"):
        fq = os.path.join(a.work, "c2.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_reads(1000000, 100, 5000000, 2), seed=2)
        for t in (1, 8, 64):
            out = os.path.join(a.work, f"c2_s_t{t}.fqs")
            run_ref(fq, out, "s", t, 5, a.work)
            meta = {"reads": 1000000, "len": 100, "genome": 5000000, "seed": 2, "gs": 5, "om": "s", "threads": t}
            json.dump(digest(out, meta), open(os.path.join(GOLD, f"c2_1M_s_t{t}.json"), "w"))
    if a.only in ("", "c4"):
        ids, seqs, quals = ragged()
        fq = os.path.join(a.work, "c4.fq")
        with open(fq, "wb") as f:
            for i, sq, q in zip(ids, seqs, quals):
                f.write(i + b"\n" + sq + b"\n+\n" + q + b"\n")
        for om in "os":
            run_ref(fq, os.path.join(GOLD, f"c4_ragged_{om}_t3.fqs"), om, 3, 1, a.work)
    if a.only in ("", "c5"):
        from fqsqueezer_amd.synth import synth_pairs
        r1, r2 = synth_pairs(4000, 100, 60000, 5)
        f1, f2 = os.path.join(a.work, "c5_1.fq"), os.path.join(a.work, "c5_2.fq")
        write_fastq(f1, r1, seed=5, mate=1)
        write_fastq(f2, r2, seed=6, mate=2)
        for om in "os":
            for t in (1, 4):
                out = os.path.join(GOLD, f"c5_pe4k_{om}_t{t}.fqs")
                if not os.path.exists(out):
                    subprocess.check_call([REF, "e", "-p", "-om", om, "-t", str(t), "-gs", "1", "-qm", "n", "-im", "n", "-v", "0",
                                           "-tmp", os.path.join(a.work, "tmpp_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
    if a.only in ("", "c20"):   # paired-end mates longer than the 4096 bases a worker stages in LDS (the reference takes up to 2^24, fqs/meta.cpp:69)
        from fqsqueezer_amd.synth import synth_pairs
        r1, r2 = synth_pairs(240, 5000, 40000, 21, frag_min=6000, frag_max=9000)
        f1, f2 = os.path.join(a.work, "c20_1.fq"), os.path.join(a.work, "c20_2.fq")
        write_fastq(f1, r1, seed=21, mate=1)
        write_fastq(f2, r2, seed=22, mate=2)
        for om in "os":
            out = os.path.join(GOLD, f"c20_pelong_{om}_t2.fqs")
            if not os.path.exists(out):
                subprocess.check_call([REF, "e", "-p", "-om", om, "-t", "2", "-gs", "1", "-qm", "n", "-im", "n", "-v", "0",
                                       "-tmp", os.path.join(a.work, "tmpq_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
    if a.only in ("", "c6"):
        fq = os.path.join(a.work, "c6.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_reads(20000, 100, 1000000, 6), seed=6)
        out = os.path.join(a.work, "c6_s_t2.fqs")
        run_ref(fq, out, "s", 2, 300, a.work)
        meta = {"reads": 20000, "len": 100, "genome": 1000000, "seed": 6, "gs": 300, "om": "s", "threads": 2}
        json.dump(digest(out, meta), open(os.path.join(GOLD, "c6_20k_gs300_s_t2.json"), "w"))
    if a.only == "c9":   # default geometry: only on request (45 GiB, 10 min)
        fq = os.path.join(a.work, "c9.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_reads(20000, 150, 200000, 9), seed=9)
        out = os.path.join(a.work, "c9_s_t2.fqs")
        run_ref(fq, out, "s", 2, 3100, a.work)
        meta = {"reads": 20000, "len": 150, "genome": 200000, "seed": 9, "gs": 3100, "om": "s", "threads": 2}
        json.dump(digest(out, meta), open(os.path.join(GOLD, "c9_20k150_gs3100_s_t2.json"), "w"))
    if a.only in ("", "c7"):
        from fqsqueezer_amd.synth import synth_mixed_lengths
        ids, seqs, quals = synth_mixed_lengths()
        fq = os.path.join(a.work, "c7.fq")
        with open(fq, "wb") as f:
            for i, sq, q in zip(ids, seqs, quals):
                f.write(i + b"\n" + sq + b"\n+\n" + q + b"\n")
        for om in "os":
            run_ref(fq, os.path.join(GOLD, f"c7_mixedlen_{om}_t3.fqs"), om, 3, 1, a.work)
    if a.only in ("", "c8"):   # quality streams (-qm o/8/4/2 on c1; -qm 8 on the c5 pairs): per-block SHA-256 of the quality streams
        def qdigest(path, meta):
            header, blocks = hp.parse_fqs(open(path, "rb").read())
            d = dict(meta, header=header.hex(), n_blocks=len(blocks), blocks=[])
            for b in blocks:
                h = hashlib.sha256()
                n = 0
                for st in b.streams:
                    h.update(st[hp.STREAM_QUALITY])
                    n += len(st[hp.STREAM_QUALITY])
                d["blocks"].append({"n_reads": b.n_reads, "sha256": h.hexdigest(), "bytes": n})
            return d
        fq = os.path.join(a.work, "c1.fq")
        for qm in ("o", "8", "4", "2"):
            out = os.path.join(a.work, f"q1_{qm}_t4.fqs")
            if not os.path.exists(out):
                subprocess.check_call([REF, "e", "-s", "-om", "o", "-t", "4", "-gs", "1", "-qm", qm] + (["-qt", "25"] if qm == "2" else []) +
                                      ["-im", "n", "-v", "0", "-tmp", os.path.join(a.work, "tmpq_"), "-out", out, fq], stdout=subprocess.DEVNULL)
            json.dump(qdigest(out, {"input": "c1 (10k x 100bp, seed 1)", "qm": qm, "om": "o", "threads": 4}),
                      open(os.path.join(GOLD, f"c8_qual_{qm}_t4.json"), "w"))
        out = os.path.join(a.work, "q5_8_t3.fqs")
        if not os.path.exists(out):
            subprocess.check_call([REF, "e", "-p", "-om", "s", "-t", "3", "-gs", "1", "-qm", "8", "-im", "n", "-v", "0", "-tmp", os.path.join(a.work, "tmpq_"),
                                   "-out", out, os.path.join(a.work, "c5_1.fq"), os.path.join(a.work, "c5_2.fq")], stdout=subprocess.DEVNULL)
        json.dump(qdigest(out, {"input": "c5 (4000 pairs, seed 5)", "qm": "8", "om": "s", "threads": 3, "paired": True}),
                  open(os.path.join(GOLD, "c8_qual_pe8_t3.json"), "w"))
    if a.only in ("", "c10"):   # complete files in full modes: ids (host coder), qualities, meta, DNA
        from fqsqueezer_amd.synth import synth_ids_varied, synth_pairs, synth_quals

        def write_fq(path, ids, reads, quals):
            with open(path, "wb") as f:
                for i in range(len(ids)):
                    f.write(ids[i] + b"\n" + reads[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")

        def fdigest(path, meta):
            data = open(path, "rb").read()
            header, blocks = hp.parse_fqs(data)
            d = dict(meta, header=header.hex(), n_blocks=len(blocks), file_sha256=hashlib.sha256(data).hexdigest(), file_bytes=len(data), blocks=[])
            for b in blocks:
                e = {"n_reads": b.n_reads}
                for sid in hp.stored_streams(header):
                    h = hashlib.sha256()
                    for st in b.streams:
                        h.update(st[sid])
                    e[str(sid)] = h.hexdigest()
                d["blocks"].append(e)
            return d

        n = 3000
        fq = os.path.join(a.work, "c10.fq")
        write_fq(fq, synth_ids_varied(n, 10), synth_reads(n, 100, 80000, 10), synth_quals(n, 100, 10))
        out = os.path.join(GOLD, "c10_full_o_t3.fqs")
        subprocess.check_call([REF, "e", "-s", "-om", "o", "-t", "3", "-gs", "1", "-qm", "o", "-im", "o", "-v", "0",
                               "-tmp", os.path.join(a.work, "tmpf_"), "-out", out, fq], stdout=subprocess.DEVNULL)
        for tag, om, qm, im, t in (("s_i8_t4", "s", "8", "i", 4), ("s_oo_t2", "s", "o", "o", 2), ("o_i2_t5", "o", "2", "i", 5)):
            out = os.path.join(a.work, f"c10_{tag}.fqs")
            subprocess.check_call([REF, "e", "-s", "-om", om, "-t", str(t), "-gs", "1", "-qm", qm, "-im", im, "-v", "0",
                                   "-tmp", os.path.join(a.work, "tmpf_"), "-out", out, fq], stdout=subprocess.DEVNULL)
            json.dump(fdigest(out, {"input": "c10 (3000 x 100bp, G=80kbp, seed 10, synth_ids_varied)", "om": om, "qm": qm, "im": im, "threads": t}),
                      open(os.path.join(GOLD, f"c10_full_{tag}.json"), "w"))
        np_ = 2000
        r1, r2 = synth_pairs(np_, 100, 60000, 11)
        f1, f2 = os.path.join(a.work, "c11_1.fq"), os.path.join(a.work, "c11_2.fq")
        write_fq(f1, synth_ids_varied(np_, 11, 1), r1, synth_quals(np_, 100, 11))
        write_fq(f2, synth_ids_varied(np_, 11, 2), r2, synth_quals(np_, 100, 12))
        for tag, om, qm, im, t in (("s_o4_t3", "s", "4", "o", 3), ("o_io_t2", "o", "o", "i", 2)):
            out = os.path.join(a.work, f"c11_{tag}.fqs")
            subprocess.check_call([REF, "e", "-p", "-om", om, "-t", str(t), "-gs", "1", "-qm", qm, "-im", im, "-v", "0",
                                   "-tmp", os.path.join(a.work, "tmpf_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
            json.dump(fdigest(out, {"input": "c11 (2000 pairs x 100bp, G=60kbp, seed 11, synth_ids_varied mates 1/2)", "om": om, "qm": qm, "im": im,
                                    "threads": t, "paired": True}), open(os.path.join(GOLD, f"c11_pe_full_{tag}.json"), "w"))
    if a.only in ("", "c12"):
        c12(a)
    if a.only in ("", "c13"):
        c13(a)
    if a.only in ("", "c14"):
        c14_c15(a)
    if a.only in ("", "c16"):
        c16(a)
    if a.only == "c17":
        c17(a)
    if a.only == "c18":
        c18(a)
    if a.only in ("c19", "c19t64"):
        c19(a, 64 if a.only == "c19t64" else 8)
    if a.only == "c21":
        c21(a, 8)
    if a.only == "c22":
        c22(a)
    if a.only in ("", "c3"):
        fq = os.path.join(a.work, "c3.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_reads(50000, 150, 250000, 3), seed=3)
        out = os.path.join(a.work, "c3_s_t8.fqs")
        run_ref(fq, out, "s", 8, 8, a.work)
        meta = {"reads": 50000, "len": 150, "genome": 250000, "seed": 3, "gs": 8, "om": "s", "threads": 8}
        json.dump(digest(out, meta), open(os.path.join(GOLD, "c3_50k150_s_t8.json"), "w"))


def c12(a):
    fq = os.path.join(a.work, "c12.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(1000000, 150, 7500000, 2), seed=2)
    for t in (64, 8):
        out = os.path.join(a.work, f"c12_s_t{t}.fqs")
        run_ref(fq, out, "s", t, 8, a.work)
        meta = {"reads": 1000000, "len": 150, "genome": 7500000, "seed": 2, "gs": 8, "om": "s", "threads": t}
        json.dump(digest(out, meta), open(os.path.join(GOLD, f"c12_1M150_s_t{t}.json"), "w"))


def fdigest(path, meta):
    """per-block SHA-256 of every stored stream + digest of the whole file"""
    data = open(path, "rb").read()
    header, blocks = hp.parse_fqs(data)
    d = dict(meta, header=header.hex(), n_blocks=len(blocks), file_sha256=hashlib.sha256(data).hexdigest(), file_bytes=len(data), blocks=[])
    for b in blocks:
        e = {"n_reads": b.n_reads}
        for sid in hp.stored_streams(header):
            h = hashlib.sha256()
            for st in b.streams:
                h.update(st[sid])
            e[str(sid)] = h.hexdigest()
        d["blocks"].append(e)
    return d


def write_fq_ids(path, ids, reads, quals):
    with open(path, "wb") as f:
        for i in range(len(ids)):
            f.write(ids[i] + b"\n" + reads[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")


def c13(a):
    from fqsqueezer_amd.synth import synth_two_haplotypes
    for om, n, G in (("o", 300000, 2500), ("s", 500000, 2000)):
        fq = os.path.join(a.work, f"c13_{om}.fq")
        if not os.path.exists(fq):
            write_fastq(fq, synth_two_haplotypes(n, 100, G, 13), seed=13)
        out = os.path.join(a.work, f"c13_{om}_t4.fqs")
        run_ref(fq, out, om, 4, 1, a.work)
        meta = {"reads": n, "len": 100, "genome": G, "seed": 13, "gs": 1, "om": om, "threads": 4, "synth": "two_haplotypes"}
        json.dump(digest(out, meta), open(os.path.join(GOLD, f"c13_sat_{om}_t4.json"), "w"))


def c14_c15(a):
    from fqsqueezer_amd.synth import synth_ids_varied, synth_pairs, synth_quals, read_id
    for tag, npairs, G, seed, gs, qm, im, t, varied in (("c14_pe150_s_q8_t8", 100000, 1500000, 14, 2, "8", "n", 8, False),
                                                        ("c15_pe150_s_oo_t4", 20000, 300000, 15, 1, "o", "o", 4, True)):
        r1, r2 = synth_pairs(npairs, 150, G, seed)
        f1, f2 = os.path.join(a.work, tag + "_1.fq"), os.path.join(a.work, tag + "_2.fq")
        ids1 = synth_ids_varied(npairs, seed, 1) if varied else [read_id(i, 1) for i in range(npairs)]
        ids2 = synth_ids_varied(npairs, seed, 2) if varied else [read_id(i, 2) for i in range(npairs)]
        write_fq_ids(f1, ids1, r1, synth_quals(npairs, 150, seed))
        write_fq_ids(f2, ids2, r2, synth_quals(npairs, 150, seed + 1))
        out = os.path.join(a.work, tag + ".fqs")
        subprocess.check_call([REF, "e", "-p", "-om", "s", "-t", str(t), "-gs", str(gs), "-qm", qm, "-im", im, "-v", "0",
                               "-tmp", os.path.join(a.work, "tmpc_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
        meta = {"pairs": npairs, "len": 150, "genome": G, "seed": seed, "gs": gs, "om": "s", "qm": qm, "im": im, "threads": t,
                "paired": True, "varied_ids": varied}
        json.dump(fdigest(out, meta), open(os.path.join(GOLD, tag + ".json"), "w"))


def c16(a):
    fq = os.path.join(a.work, "c12.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(1000000, 150, 7500000, 2), seed=2)
    out = os.path.join(a.work, "c16_s_ids_t8.fqs")
    if not os.path.exists(out):
        subprocess.check_call([REF, "e", "-s", "-om", "s", "-t", "8", "-gs", "8", "-qm", "n", "-im", "o", "-v", "0",
                               "-tmp", os.path.join(a.work, "tmpi_"), "-out", out, fq], stdout=subprocess.DEVNULL)
    meta = {"reads": 1000000, "len": 150, "genome": 7500000, "seed": 2, "gs": 8, "om": "s", "qm": "n", "im": "o", "threads": 8}
    json.dump(fdigest(out, meta), open(os.path.join(GOLD, "c16_1M150_s_ids_t8.json"), "w"))


def c17(a):
    fq = os.path.join(a.work, "c12.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(1000000, 150, 7500000, 2), seed=2)
    out = os.path.join(a.work, "c17_s_t8.fqs")
    run_ref(fq, out, "s", 8, 3100, a.work)
    meta = {"reads": 1000000, "len": 150, "genome": 7500000, "seed": 2, "gs": 3100, "om": "s", "threads": 8}
    json.dump(digest(out, meta), open(os.path.join(GOLD, "c17_1M150_gs3100_s_t8.json"), "w"))


def c18(a):
    from fqsqueezer_amd.synth import synth_pairs, synth_quals, read_id
    npairs, G, seed = 5000000, 75000000, 3
    f1, f2 = os.path.join(a.work, "c18_1.fq"), os.path.join(a.work, "c18_2.fq")
    if not os.path.exists(f2):
        r1, r2 = synth_pairs(npairs, 150, G, seed)
        write_fastq(f1, r1, seed=seed, mate=1)
        write_fastq(f2, r2, seed=seed + 1, mate=2)
    out = os.path.join(a.work, "c18.fqs")
    if not os.path.exists(out):
        subprocess.check_call([REF, "e", "-p", "-om", "s", "-t", "8", "-gs", "75", "-qm", "8", "-v", "0",
                               "-tmp", os.path.join(a.work, "tmp18_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
    meta = {"pairs": npairs, "len": 150, "genome": G, "seed": seed, "gs": 75, "om": "s", "qm": "8", "im": "i", "threads": 8,
            "paired": True, "varied_ids": False}
    json.dump(fdigest(out, meta), open(os.path.join(GOLD, "c18_pe5M_s_q8_t8.json"), "w"))


def c19(a, t):
    fq = os.path.join(a.work, "c19.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(10000000, 150, 300000000, 19), seed=19)
    out = os.path.join(a.work, f"c19_s_t{t}.fqs")
    run_ref(fq, out, "s", t, 300, a.work)
    meta = {"reads": 10000000, "len": 150, "genome": 300000000, "seed": 19, "gs": 300, "om": "s", "threads": t}
    json.dump(digest(out, meta), open(os.path.join(GOLD, f"c19_10M150_gs300_s_t{t}.json"), "w"))


def c21(a, t):
    """SURVEY 8d-4's check: a 5 M-read prefix of BASELINE configs[3]'s file -- reads sampled from a G = 3.1 Gbp genome, coded at the
    reference's DEFAULT geometry -gs 3100 (k = 13/18/21/27, 16 GiB p-mer vector, 4^14 b-mer sub-tables of 16 slots each before
    the first read).  -t 8 and 5 M reads are what this 62 GiB container allows: the reference holds ~45 GiB before the first read
    (c17), ~0.45 GiB more per thread, and its s-mer sub-tables grow by ~0.9 GB per million reads of uncovered sequence (at -t 64
    with 10 M reads the kernel killed it)."""
    fq = os.path.join(a.work, "c21.fq")
    if not os.path.exists(fq):
        write_fastq(fq, synth_reads(5000000, 150, 3100000000, 21), seed=21)
    out = os.path.join(a.work, f"c21_s_t{t}.fqs")
    run_ref(fq, out, "s", t, 3100, a.work)
    meta = {"reads": 5000000, "len": 150, "genome": 3100000000, "seed": 21, "gs": 3100, "om": "s", "threads": t}
    json.dump(digest(out, meta), open(os.path.join(GOLD, f"c21_5M150_G3100_gs3100_s_t{t}.json"), "w"))


def c22(a):
    """BASELINE configs[4]'s modes at 1 M pairs: 150 bp mates (fragments 300-600) of a 15 Mbp genome (20x), varied Illumina-style ids,
    -p -om s -qm o -im o (lossless qualities and ids) -gs 15, T = 8: every stream of every block + the file's SHA-256."""
    from fqsqueezer_amd.synth import synth_ids_varied, synth_pairs, synth_quals
    npairs, G, seed = 1000000, 15000000, 22
    r1, r2 = synth_pairs(npairs, 150, G, seed)
    f1, f2 = os.path.join(a.work, "c22_1.fq"), os.path.join(a.work, "c22_2.fq")
    write_fq_ids(f1, synth_ids_varied(npairs, seed, 1), r1, synth_quals(npairs, 150, seed))
    write_fq_ids(f2, synth_ids_varied(npairs, seed, 2), r2, synth_quals(npairs, 150, seed + 1))
    out = os.path.join(a.work, "c22.fqs")
    subprocess.check_call([REF, "e", "-p", "-om", "s", "-t", "8", "-gs", "15", "-qm", "o", "-im", "o", "-v", "0",
                           "-tmp", os.path.join(a.work, "tmp22_"), "-out", out, f1, f2], stdout=subprocess.DEVNULL)
    meta = {"pairs": npairs, "len": 150, "genome": G, "seed": seed, "gs": 15, "om": "s", "qm": "o", "im": "o", "threads": 8,
            "paired": True, "varied_ids": True}
    json.dump(fdigest(out, meta), open(os.path.join(GOLD, "c22_pe1M_s_oo_t8.json"), "w"))


def ragged():
    """c4: ragged reads with N runs and duplicates, both orders, T=3 (complete reference outputs)."""
    from fqsqueezer_amd.synth import synth_ragged
    ids, seqs, quals = synth_ragged(3000, 60000, 4)
    return ids, seqs, quals


if __name__ == "__main__":
    main()
