"""Parity tests proper: the HIP path (through the C ABI) on a real MI355X against the reference
goldens, the oracle, and size-independent properties at the benchmark's full size."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import c20_records, GOLD, c1_records, c4_records, c5_records, c7_records, check_against_digest, check_against_fqs, check_against_fqs_pe, check_decode_fqs
from fqsqueezer_amd import hostpipe as hp

pytestmark = pytest.mark.gpu


def gpu(header):
    from fqsqueezer_amd.codec import DnaCodec
    return DnaCodec(header, device=0)


@pytest.mark.parametrize("name", ["c1_10k_o_t1.fqs", "c1_10k_o_t4.fqs", "c1_10k_s_t1.fqs", "c1_10k_s_t4.fqs"])
def test_hip_matches_reference_10k(name):
    check_against_fqs(gpu, c1_records(), name)


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c1_10k_o_t1.fqs"])
def test_hip_growth_posted_by_the_device_is_recovered(name, monkeypatch):
    """Nothing is read back inside a block: the device checks the growth rule, stops the block's queue of kernels, and
    the host grows the tables and resumes at that phase's inserts (fqsx_api.hip: phase_skip / block_recover).  From
    256 slots per owner that happens in most blocks -- with the queue of real kernels in flight, which the emulation
    build's test of the same name cannot show."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    check_against_fqs(gpu, c1_records(), name)


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c1_10k_o_t1.fqs"])
def test_hip_chunked_tables_grow_sub_table_by_sub_table(name, monkeypatch):
    """fqsx_dna_use_chunked_tables: the k-mer tables in hipMemCreate chunks (one per sub-table) mapped into one reserved range;
    growths re-insert sub-table by sub-table and give each old chunk back at once.  Same streams, same table contents."""
    from fqsqueezer_amd.codec import DnaCodec
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_CHUNK_STEP_KB", "1")      # (one sub-table per growth step although the tables are small)
    codec = check_against_fqs(lambda h: DnaCodec(h, device=0, chunked_tables=True), c1_records(), name)
    cap, plain = codec.capacity(), check_against_fqs(gpu, c1_records(), name).capacity()
    assert cap["growths"] >= 4 and cap["table_bytes_held"] >= 8 * (cap["smer_slots"] + cap["bmer_slots"])   # (a chunk is at least 2 MiB)
    assert (plain["smers"], plain["bmers"], plain["growths"]) == (cap["smers"], cap["bmers"], cap["growths"])


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c5_pe4k_s_t4.fqs"])
def test_hip_overflow_chains_of_the_two_choice_tables(name, monkeypatch):
    """Two-choice buckets driven to 85 % load from 64-slot sub-tables: hundreds of keys whose two buckets are both full go down the
    overflow chains -- lane-parallel look-ups of the scouts, in-order batch inserts, parallel re-inserts at ~50 growths, and the
    decoder walking the same chains.  Streams stay the reference's (tests/test_emu_parity.py has the 1-lane twin)."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "64")
    monkeypatch.setenv("FQSX_TAB_LOAD_PCT", "85")
    monkeypatch.setenv("FQSX_TAB_AFTER_PCT", "80")
    if name.startswith("c5"):
        check_against_fqs_pe(gpu, c5_records(), name)
        check_decode_fqs(gpu, c5_records(), name)
    else:
        cap = check_against_fqs(gpu, c1_records(), name).capacity()
        assert cap["bytes_per_bmer"] < 10.2 and cap["growths"] >= 20, cap
        check_decode_fqs(gpu, c1_records(), name)


def test_hip_tables_turn_into_chunked_tables_at_a_size_by_themselves(monkeypatch):
    """a table of one GPU that reaches FQSX_CHUNK_AUTO_KB (default 2 GiB: the c19 test passes it) continues as a chunked table"""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_CHUNK_STEP_KB", "1")
    monkeypatch.setenv("FQSX_CHUNK_AUTO_KB", "64")
    cap = check_against_fqs(gpu, c1_records(), "c1_10k_s_t4.fqs").capacity()
    assert cap["growths"] >= 6 and cap["table_bytes_held"] >= 8 * (cap["smer_slots"] + cap["bmer_slots"])


@pytest.mark.parametrize("name", ["c4_ragged_o_t3.fqs", "c4_ragged_s_t3.fqs"])
def test_hip_matches_reference_ragged(name):
    check_against_fqs(gpu, c4_records(), name)


@pytest.mark.parametrize("name", ["c5_pe4k_o_t1.fqs", "c5_pe4k_o_t4.fqs", "c5_pe4k_s_t1.fqs", "c5_pe4k_s_t4.fqs"])
def test_hip_matches_reference_paired_end(name):
    check_against_fqs_pe(gpu, c5_records(), name)


@pytest.mark.parametrize("name", ["c5_pe4k_o_t4.fqs", "c5_pe4k_s_t4.fqs"])
def test_hip_paired_end_growth_posted_by_the_device_is_recovered(name, monkeypatch):
    """Paired-end encoding reads nothing back inside a block either: the partition kernel checks the k-mer tables' growth rule,
    the demand kernel the pair table's; a posted phase stops the block's queue and the host grows and resumes at that phase's
    inserts (pair table and k-mer tables).  From 256-slot k-mer sub-tables and 64-slot pair sub-tables both happen in most blocks."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_PTAB_INIT", "64")
    codec = check_against_fqs_pe(gpu, c5_records(), name)
    cap = codec.capacity()
    assert cap["growths"] >= 8 and cap["pair_slots"] > 64 * 4, cap


@pytest.mark.parametrize("name", ["c20_pelong_o_t2.fqs", "c20_pelong_s_t2.fqs"])
def test_hip_paired_end_mates_longer_than_the_lds_staging(name):
    """5000 bp mates (the reference takes up to 2^24 bases, fqs/meta.cpp:69): code lines and the reverse-complement line of an
    anchored second mate in the worker's HBM scratch instead of LDS; encode against the reference's file, then decode it"""
    check_against_fqs_pe(gpu, c20_records(), name)
    check_decode_fqs(gpu, c20_records(), name)


def test_hip_matches_oracle_paired_end_many_workers():
    from oracle.pyoracle import OracleCodec
    rec1, rec2 = c5_records()
    for T, mode in [(16, "pe_sorted"), (64, "pe_original")]:
        header = hp.make_header(T, mode, 1)
        a, b = gpu(header), OracleCodec(header)
        for g, idx in enumerate(hp.form_blocks_pe(rec1, rec2, mode)[:40]):
            bases, off = hp.block_arrays_pe(rec1, rec2, idx)
            assert a.encode_block(bases, off, g) == b.encode_block(bases, off, g)


@pytest.mark.parametrize("name", ["c7_mixedlen_o_t3.fqs", "c7_mixedlen_s_t3.fqs"])
def test_hip_matches_reference_short_and_long_reads(name):
    check_against_fqs(gpu, c7_records(), name)


def test_hip_matches_reference_large_k_geometry():
    check_against_digest(gpu, "c6_20k_gs300_s_t2.json")


def test_hip_matches_reference_default_geometry():
    """-gs 3100, the reference's default and BASELINE configs[3]'s geometry: 16 GiB p-mer vector in HBM."""
    check_against_digest(gpu, "c9_20k150_gs3100_s_t2.json")


@pytest.mark.parametrize("name,recs", [("c1_10k_o_t4.fqs", c1_records), ("c1_10k_s_t4.fqs", c1_records), ("c4_ragged_s_t3.fqs", c4_records),
                                       ("c7_mixedlen_o_t3.fqs", c7_records), ("c7_mixedlen_s_t3.fqs", c7_records),
                                       ("c5_pe4k_o_t4.fqs", c5_records), ("c5_pe4k_s_t4.fqs", c5_records)])
def test_hip_decodes_reference_streams(name, recs):
    check_decode_fqs(gpu, recs(), name)


def test_hip_matches_reference_150bp():
    check_against_digest(gpu, "c3_50k150_s_t8.json")


def test_hip_matches_reference_1M_t64_bit_exact():
    """BASELINE configs[1] at full size: every block's DNA streams hash-identical to `fqs-1.1 e -t 64`."""
    codec = check_against_digest(gpu, "c2_1M_s_t64.json")
    st = codec.stats()
    assert st["bases"] <= 100_000_000 and st["coded"] > 80_000_000


@pytest.mark.parametrize("t", [1, 8])
def test_hip_matches_reference_1M_other_worker_counts(t):
    """SURVEY 8(d) config 2 asks T in {1, 8, 64}: the same file as `fqs-1.1 e -t 1` / `-t 8` would write it (T = 1 is one
    workgroup for the whole file: the first 40 blocks)."""
    check_against_digest(gpu, f"c2_1M_s_t{t}.json", max_blocks=40 if t == 1 else None)


@pytest.mark.parametrize("t", [64, 8])
def test_hip_matches_reference_1M_150bp_bit_exact(t):
    """The workload BASELINE.json's metric is quoted on (bench.py's default): 1 M x 150 bp SE sorted, -gs 8, every
    block's DNA streams hash-identical to `fqs-1.1 e -t 64` (and `-t 8`)."""
    codec = check_against_digest(gpu, f"c12_1M150_s_t{t}.json")
    st = codec.stats()
    assert st["bases"] <= 150_000_000 and st["coded"] > 120_000_000


def test_hip_matches_reference_default_geometry_1M():
    """BASELINE configs[3]'s geometry (-gs 3100: k = 13/18/21/27, 16 GiB p-mer vector, up to 1024-way partial look-ups) on
    the metric's 1 M x 150 bp file: every block hash-identical to `fqs-1.1 e -t 8 -gs 3100` (which needs 45 GiB and 10 min)."""
    check_against_digest(gpu, "c17_1M150_gs3100_s_t8.json")


@pytest.mark.parametrize("om", ["o", "s"])
def test_hip_matches_reference_saturated_counters(om):
    """c13: counters at their maxima, counts_level_t::mixed / bmer_unc, probabilistic increments on s-mers (the oracle
    test of the same fixture asserts that these branches are reached)."""
    check_against_digest(gpu, f"c13_sat_{om}_t4.json")


def test_hip_matches_oracle_many_workers_and_tiny_blocks():
    from oracle.pyoracle import OracleCodec
    rec = c4_records()
    for T, mode in [(7, "se_sorted"), (16, "se_original"), (64, "se_original")]:
        header = hp.make_header(T, mode, 1)
        a, b = gpu(header), OracleCodec(header)
        blks = hp.form_blocks(rec, mode)[:12] if mode == "se_sorted" else [np.arange(0, 700), np.arange(700, 731), np.arange(731, 733)]
        for g, idx in enumerate(blks):
            bases, off = hp.block_arrays(rec, np.asarray(idx))
            assert a.encode_block(bases, off, g) == b.encode_block(bases, off, g)


def test_hip_worker_pipeline_matches_oracle_sorted_mode():
    """The five wavefronts of a worker (resolve / coder / inserter / read head / scout) against the CPU oracle in the
    mode that uses all of them: the bitstream's maximum of 255 workers on reads with substitutions and N (k-mer
    corrections make the resolving wave drop the scout's chunks), and 64 workers on ragged reads with N runs,
    duplicates and lengths from below the k-mer sizes to several chunks (`c4`, `c7`)."""
    from oracle.pyoracle import OracleCodec
    from fqsqueezer_amd.synth import synth_reads
    reads = synth_reads(30000, 100, 200000, 23)
    cases = [(255, hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)), (64, c4_records()), (64, c7_records())]
    for T, rec in cases:
        header = hp.make_header(T, "se_sorted", 1)
        a, b = gpu(header), OracleCodec(header)
        order = np.concatenate(hp.sorted_order(rec))          # the file in sorted order, cut into blocks of >= 2 T reads
        B = 2048 if T == 255 else 1500
        for g, lo in enumerate(range(0, len(order) - 2 * T, B)):
            bases, off = hp.block_arrays(rec, order[lo:lo + B])
            assert a.encode_block(bases, off, g) == b.encode_block(bases, off, g), f"T={T} block {g}"


def test_hip_long_reads_with_corrections_followed_by_duplicates():
    """350 bp reads (six stage-P chunks) with substitutions -- k-mer corrections restart the scouts in mid-read -- each
    followed by exact duplicates, which the resolving wave finishes without taking a single chunk: the case in which
    round 1's scout ring could stay full of stale chunks.  Sorted mode, T = 16 and 64, against the oracle."""
    from oracle.pyoracle import OracleCodec
    from fqsqueezer_amd.synth import synth_reads
    base = synth_reads(2500, 350, 60000, 41, sub_rate=0.01)
    reads = np.repeat(base, 3, axis=0)[: 3 * 2500 : 1]          # every read three times
    reads = reads[np.random.Generator(np.random.PCG64(5)).permutation(len(reads))]
    rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
    for T in (16, 64):
        header = hp.make_header(T, "se_sorted", 1)
        a, b = gpu(header), OracleCodec(header)
        order = np.concatenate(hp.sorted_order(rec))
        for g, lo in enumerate(range(0, len(order) - 2 * T, 1200)):
            bases, off = hp.block_arrays(rec, order[lo:lo + 1200])
            assert a.encode_block(bases, off, g) == b.encode_block(bases, off, g), f"T={T} block {g}"


def test_gpu_encode_decode_round_trip_many_workers():
    """encode -> decode entirely on the GPU at T=64 (size-independent property: the block comes back)."""
    from fqsqueezer_amd.synth import synth_reads
    reads = synth_reads(60000, 100, 600000, 11)
    rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
    header = hp.make_header(64, "se_sorted", 1)
    enc, dec = gpu(header), gpu(header)
    for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")):
        bases, off = hp.block_arrays(rec, idx)
        streams = enc.encode_block(bases, off, g)
        assert np.array_equal(dec.decode_block(streams, off, g), np.asarray(bases)), f"block {g} did not round-trip"


def test_sharded_path_on_the_gpu_matches_plain_run():
    """The sharded mode's kernels (count matrix, rank-major pack, merge, item collection, replica update) on the real GPU:
    a world of one rank over RCCL -- every collective degenerates, every kernel runs (the replica update on the rank's
    own items, which must change nothing).  The multi-rank exchange itself is covered on CPU (tests/test_sharded_cpu.py)."""
    import torch
    import torch.distributed as dist
    from fqsqueezer_amd.sharded import ShardedDnaCodec
    from fqsqueezer_amd.synth import synth_reads
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    reads = synth_reads(20000, 100, 150000, 37)
    rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
    header = hp.make_header(16, "se_sorted", 1)
    sh, one = ShardedDnaCodec(header, 0, 1, device=0, apply_own=True), gpu(header)
    for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:60]):
        bases, off = hp.block_arrays(rec, idx)
        mine, ref = sh.encode_block(bases, off, g), one.encode_block(bases, off, g)
        assert [mine[w] for w in range(16)] == ref, f"block {g}"
    assert sh.traffic["phases"] >= 60
    dist.destroy_process_group()


def test_native_sharded_driver_on_the_gpu_over_rccl():
    """fqsx_shard_encode_block with the RCCL transport inside the library (a world of one rank: every collective runs through
    librccl on the codec's stream, every kernel of the native phase loop runs), single- and paired-end, with the replica
    update on the rank's own items switched on.  The multi-rank exchange is covered on CPU (tests/test_sharded_cpu.py)."""
    from fqsqueezer_amd.sharded import NativeShardedDnaCodec
    from fqsqueezer_amd.synth import synth_reads
    os.environ["FQSX_SHARD_APPLY_OWN"] = "1"
    try:
        reads = synth_reads(20000, 100, 150000, 37)
        rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
        header = hp.make_header(16, "se_sorted", 1)
        sh = NativeShardedDnaCodec(header, 0, 1, device=0, transport="rccl", id_bytes=NativeShardedDnaCodec.rccl_unique_id())
        one = gpu(header)
        for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:60]):
            bases, off = hp.block_arrays(rec, idx)
            mine, ref = sh.encode_block(bases, off, g), one.encode_block(bases, off, g)
            assert [mine[w] for w in range(16)] == ref, f"block {g}"
        tr = sh.traffic
        votes = tr["collectives"] - 3 * tr["phases"]   # (one status vote per phase in which buffers or tables grow)
        assert tr["phases"] >= 60 and 0 <= votes <= tr["phases"] // 2, tr
        sh.close()
        rec1, rec2 = c5_records()
        header = hp.make_header(8, "pe_sorted", 1)
        sh = NativeShardedDnaCodec(header, 0, 1, device=0, transport="rccl", id_bytes=NativeShardedDnaCodec.rccl_unique_id())
        one = gpu(header)
        for g, idx in enumerate(hp.form_blocks_pe(rec1, rec2, "pe_sorted")[:30]):
            bases, off = hp.block_arrays_pe(rec1, rec2, idx)
            mine, ref = sh.encode_block(bases, off, g), one.encode_block(bases, off, g)
            assert [mine[w] for w in range(8)] == ref, f"paired-end block {g}"
        sh.close()
    finally:
        os.environ.pop("FQSX_SHARD_APPLY_OWN", None)


def test_device_and_host_entry_points_agree():
    import torch
    rec = c1_records()
    header = hp.make_header(4, "se_sorted", 1)
    a, b = gpu(header), gpu(header)
    for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:20]):
        bases, off = hp.block_arrays(rec, idx)
        d_b = torch.from_numpy(np.ascontiguousarray(bases)).cuda()
        d_o = torch.from_numpy(off.view(np.int64)).cuda()
        torch.cuda.synchronize()
        assert a.encode_block(bases, off, g) == b.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g)


def test_determinism_and_stream_framing():
    rec = c1_records()
    header = hp.make_header(8, "se_sorted", 1)
    runs = []
    for _ in range(2):
        c = gpu(header)
        out = []
        for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:30]):
            bases, off = hp.block_arrays(rec, idx)
            out.append(c.encode_block(bases, off, g))
        runs.append(out)
    assert runs[0] == runs[1]
    assert all(len(s) >= 8 for blk in runs[0] for s in blk)      # every stream ends with 8 flush bytes
