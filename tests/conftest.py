import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU time; enabled with FQSX_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("FQSX_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set FQSX_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build_emu()
    g.build_oracle_restate() if hasattr(g, "build_oracle_restate") else None
    return g


# ---- shared helpers ---------------------------------------------------------------------
def c1_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import read_id, synth_quals, synth_reads
    reads = synth_reads(10000, 100, 200000, 1)
    return hp.Records([read_id(i) for i in range(10000)], reads, synth_quals(10000, 100, 1))


def c4_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import synth_ragged
    ids, seqs, quals = synth_ragged(3000, 60000, 4)
    return hp.Records(ids, seqs, quals)


def digest_records(meta):
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import read_id, synth_quals, synth_reads
    n, L = meta["reads"], meta["len"]
    if meta.get("synth") == "two_haplotypes":
        from fqsqueezer_amd.synth import synth_two_haplotypes
        reads = synth_two_haplotypes(n, L, meta["genome"], meta["seed"])
    else:
        reads = synth_reads(n, L, meta["genome"], meta["seed"])
    return hp.Records([read_id(i) for i in range(n)], reads, synth_quals(n, L, meta["seed"]))


def check_against_fqs(make_codec, rec, fqs_name):
    """Encode `rec` block by block and compare every worker's DNA stream with the reference .fqs."""
    from fqsqueezer_amd import hostpipe as hp
    header, blocks = hp.parse_fqs(open(os.path.join(GOLD, fqs_name), "rb").read())
    mode = {0: "se_original", 1: "se_sorted"}[header[5]]
    blks = hp.form_blocks(rec, mode)
    assert len(blks) == len(blocks)
    codec = make_codec(header)
    for g, (idx, ref) in enumerate(zip(blks, blocks)):
        assert len(idx) == ref.n_reads
        bases, off = hp.block_arrays(rec, idx)
        streams = codec.encode_block(bases, off, g)
        for w, s in enumerate(streams):
            assert s == ref.streams[w][hp.STREAM_DNA], f"{fqs_name}: block {g} worker {w} differs from the reference"
    return codec


def check_against_digest(make_codec, json_name, max_blocks=None):
    from fqsqueezer_amd import hostpipe as hp
    d = json.load(open(os.path.join(GOLD, json_name)))
    rec = digest_records(d)
    header = bytes.fromhex(d["header"])
    blks = hp.form_blocks(rec, "se_sorted" if header[5] == 1 else "se_original")
    assert len(blks) == d["n_blocks"]
    codec = make_codec(header)
    total = 0
    for g, (idx, ref) in enumerate(zip(blks, d["blocks"])):
        if max_blocks is not None and g >= max_blocks:
            break
        assert len(idx) == ref["n_reads"]
        bases, off = hp.block_arrays(rec, idx)
        streams = codec.encode_block(bases, off, g)
        h = hashlib.sha256()
        for s in streams:
            h.update(s)
        total += sum(len(s) for s in streams)
        assert sum(len(s) for s in streams) == ref["bytes"], f"{json_name}: block {g} size differs"
        assert h.hexdigest() == ref["sha256"], f"{json_name}: block {g} differs from the reference"
    if max_blocks is None:
        assert total == d["dna_bytes"]
    return codec


def c10_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import synth_ids_varied, synth_quals, synth_reads
    return hp.Records(synth_ids_varied(3000, 10), synth_reads(3000, 100, 80000, 10), synth_quals(3000, 100, 10))


def c11_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import synth_ids_varied, synth_pairs, synth_quals
    r1, r2 = synth_pairs(2000, 100, 60000, 11)
    return (hp.Records(synth_ids_varied(2000, 11, 1), r1, synth_quals(2000, 100, 11)),
            hp.Records(synth_ids_varied(2000, 11, 2), r2, synth_quals(2000, 100, 12)))


QM = {"o": "lossless", "8": "illumina_8", "4": "illumina_4", "2": "binary", "n": "none"}
IM = {"o": "lossless", "i": "instrument", "n": "none"}


def check_full_file_digest(data: bytes, json_name: str):
    """Complete .fqs file vs the digest of the reference's file: every stream of every block, then the whole file."""
    from fqsqueezer_amd import hostpipe as hp
    d = json.load(open(os.path.join(GOLD, json_name)))
    header, blocks = hp.parse_fqs(data)
    assert header.hex() == d["header"]
    assert len(blocks) == d["n_blocks"]
    names = {hp.STREAM_META: "meta", hp.STREAM_ID: "id", hp.STREAM_DNA: "dna", hp.STREAM_QUALITY: "quality"}
    for g, (b, ref) in enumerate(zip(blocks, d["blocks"])):
        assert b.n_reads == ref["n_reads"]
        for sid in hp.stored_streams(header):
            h = hashlib.sha256()
            for st in b.streams:
                h.update(st[sid])
            assert h.hexdigest() == ref[str(sid)], f"{json_name}: block {g}: {names[sid]} stream differs from the reference"
    assert len(data) == d["file_bytes"]
    assert hashlib.sha256(data).hexdigest() == d["file_sha256"]


def pe_digest_records(meta):
    """the two mate files of a paired-end digest fixture (tools/make_golden.py c14 / c15)"""
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import read_id, synth_ids_varied, synth_pairs, synth_quals
    n, seed = meta["pairs"], meta["seed"]
    r1, r2 = synth_pairs(n, meta["len"], meta["genome"], seed)
    ids1 = synth_ids_varied(n, seed, 1) if meta["varied_ids"] else [read_id(i, 1) for i in range(n)]
    ids2 = synth_ids_varied(n, seed, 2) if meta["varied_ids"] else [read_id(i, 2) for i in range(n)]
    return (hp.Records(ids1, r1, synth_quals(n, meta["len"], seed)), hp.Records(ids2, r2, synth_quals(n, meta["len"], seed + 1)))


def c5_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals
    r1, r2 = synth_pairs(4000, 100, 60000, 5)
    return (hp.Records([read_id(i, 1) for i in range(4000)], r1, synth_quals(4000, 100, 5)),
            hp.Records([read_id(i, 2) for i in range(4000)], r2, synth_quals(4000, 100, 6)))


def c20_records():
    """240 pairs x 5000 bp: mates longer than the 4096 bases a worker stages in LDS (the reference takes up to 2^24)"""
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals
    r1, r2 = synth_pairs(240, 5000, 40000, 21, frag_min=6000, frag_max=9000)
    return (hp.Records([read_id(i, 1) for i in range(240)], r1, synth_quals(240, 5000, 21)),
            hp.Records([read_id(i, 2) for i in range(240)], r2, synth_quals(240, 5000, 22)))


def check_against_fqs_pe(make_codec, recs, fqs_name):
    from fqsqueezer_amd import hostpipe as hp
    rec1, rec2 = recs
    header, blocks = hp.parse_fqs(open(os.path.join(GOLD, fqs_name), "rb").read())
    mode = {2: "pe_original", 3: "pe_sorted"}[header[5]]
    blks = hp.form_blocks_pe(rec1, rec2, mode)
    assert len(blks) == len(blocks)
    codec = make_codec(header)
    for g, (idx, ref) in enumerate(zip(blks, blocks)):
        assert 2 * len(idx) == ref.n_reads
        bases, off = hp.block_arrays_pe(rec1, rec2, idx)
        streams = codec.encode_block(bases, off, g)
        for w, s in enumerate(streams):
            assert s == ref.streams[w][hp.STREAM_DNA], f"{fqs_name}: block {g} worker {w} differs from the reference"
    return codec


def c7_records():
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.synth import synth_mixed_lengths
    ids, seqs, quals = synth_mixed_lengths()
    return hp.Records(ids, seqs, quals)


def check_decode_fqs(make_codec, rec, fqs_name):
    """Decode every block of a reference .fqs (its DNA streams + the known read lengths) and compare with the input."""
    import numpy as np
    from fqsqueezer_amd import hostpipe as hp
    header, blocks = hp.parse_fqs(open(os.path.join(GOLD, fqs_name), "rb").read())
    paired = header[5] >= 2
    if paired:
        mode = "pe_sorted" if header[5] == 3 else "pe_original"
        blks = hp.form_blocks_pe(rec[0], rec[1], mode)
    else:
        blks = hp.form_blocks(rec, "se_sorted" if header[5] == 1 else "se_original")
    codec = make_codec(header)
    for g, (idx, ref) in enumerate(zip(blks, blocks)):
        bases, off = hp.block_arrays_pe(rec[0], rec[1], idx) if paired else hp.block_arrays(rec, idx)
        out = codec.decode_block([ref.streams[w][hp.STREAM_DNA] for w in range(header[4])], off, g)
        assert np.array_equal(np.frombuffer(bytes(out), dtype=np.uint8), np.asarray(bases)), f"{fqs_name}: block {g} decoded wrongly"
    return codec


def check_quality_digest(make_codec, json_name):
    from fqsqueezer_amd import hostpipe as hp
    d = json.load(open(os.path.join(GOLD, json_name)))
    header = bytes.fromhex(d["header"])
    codec = make_codec(header)
    if d.get("paired"):
        r1, r2 = c5_records()
        blks = hp.form_blocks_pe(r1, r2, "pe_sorted")
        arrays = lambda idx: hp.qual_arrays_pe(r1, r2, idx)
    else:
        rec = c1_records()
        blks = hp.form_blocks(rec, "se_original")
        arrays = lambda idx: hp.qual_arrays(rec, idx)
    assert len(blks) == d["n_blocks"]
    for g, (idx, ref) in enumerate(zip(blks, d["blocks"])):
        q, off = arrays(idx)
        streams = codec.encode_block(q, off)
        h = hashlib.sha256()
        for s in streams:
            h.update(s)
        assert sum(len(s) for s in streams) == ref["bytes"] and h.hexdigest() == ref["sha256"], f"{json_name}: block {g} quality stream differs"
