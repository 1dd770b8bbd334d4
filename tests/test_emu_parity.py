"""The HIP kernels compiled for a 1-lane host wave (tests/emu) against the reference goldens and
the oracle.  This exercises the kernel *logic* (state layout, schedule, mailboxes, table growth)
in the GPU-less container; the 64-lane paths are covered by tests/test_gpu_parity.py on the GPU."""
import numpy as np
import pytest

from conftest import c20_records, EMU_LIB, c1_records, c4_records, c5_records, c7_records, check_against_digest, check_against_fqs, check_against_fqs_pe, check_decode_fqs
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from oracle.pyoracle import OracleCodec


def emu(header):
    return DnaCodec(header, lib_path=EMU_LIB)


@pytest.fixture(scope="module", autouse=True)
def _build(built):
    yield


@pytest.mark.parametrize("name", ["c1_10k_o_t1.fqs", "c1_10k_o_t4.fqs", "c1_10k_s_t1.fqs", "c1_10k_s_t4.fqs"])
def test_emu_matches_reference_10k(name):
    codec = check_against_fqs(emu, c1_records(), name)
    st = codec.stats()
    assert st["bases"] > 0 and st["coded"] > 0 and st["gprobe"] > 0


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c1_10k_o_t4.fqs"])
def test_growth_posted_by_the_device_is_recovered(name, monkeypatch):
    """Single-end encoding reads nothing back inside a block: when a phase's inserts would overfill a global sub-table,
    the device stops the block's queue, and the host grows the tables and takes the block up at that phase's inserts
    (fqsx_api.hip: phase_skip / block_recover).  Tables that start at 256 slots per owner make that happen in most
    blocks; the streams stay the reference's."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    check_against_fqs(emu, c1_records(), name)


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c1_10k_o_t4.fqs"])
def test_chunked_tables_grow_sub_table_by_sub_table(name, monkeypatch):
    """FQSX_CHUNKED_TABLES / fqsx_dna_use_chunked_tables: one chunk of memory per sub-table inside a reserved address range
    (memfd + mmap in this build, hipMemCreate + hipMemMap on the GPU); a growth re-inserts sub-table by sub-table and returns
    each old chunk before the next new one exists.  Streams stay the reference's; the peak stays below old + new."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_CHUNK_STEP_KB", "1")      # (one sub-table per growth step although the tables are small)
    monkeypatch.setenv("FQSX_CHUNKED_TABLES", "1")
    codec = check_against_fqs(emu, c1_records(), name)
    cap = codec.capacity()
    # (a sub-table is one chunk: its capacity -- any number of four-slot buckets -- rounded up to the allocation granule, a page here)
    assert cap["growths"] >= 4 and 0 <= cap["table_bytes_held"] - 8 * (cap["smer_slots"] + cap["bmer_slots"]) <= 2 * 4 * 4096
    monkeypatch.setenv("FQSX_CHUNKED_TABLES", "0")
    plain = check_against_fqs(emu, c1_records(), name).capacity()
    assert (plain["smers"], plain["bmers"], plain["growths"]) == (cap["smers"], cap["bmers"], cap["growths"])
    # the last growth doubled one table: side by side that is + 1/2 of the new table on top of the final footprint; in chunks
    # only one old sub-table (1/8 of that here: T = 4)
    assert cap["device_bytes_peak"] - cap["device_bytes"] < (plain["device_bytes_peak"] - plain["device_bytes"]) // 2, (cap, plain)


def test_tables_turn_into_chunked_tables_at_a_size_by_themselves(monkeypatch):
    """a table of one GPU that reaches FQSX_CHUNK_AUTO_KB (default 2 GiB) continues as a chunked table from that growth on"""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_CHUNK_STEP_KB", "1")
    monkeypatch.setenv("FQSX_CHUNK_AUTO_KB", "64")
    codec = check_against_fqs(emu, c1_records(), "c1_10k_s_t4.fqs")
    cap = codec.capacity()
    assert cap["growths"] >= 6 and 0 <= cap["table_bytes_held"] - 8 * (cap["smer_slots"] + cap["bmer_slots"]) <= 2 * 4 * 4096
    assert cap["device_bytes_peak"] - cap["device_bytes"] < 8 * cap["bmer_slots"] // 4, cap   # (the last growths went sub-table by sub-table)


@pytest.mark.parametrize("name", ["c4_ragged_o_t3.fqs", "c4_ragged_s_t3.fqs"])
def test_emu_matches_reference_ragged(name):
    check_against_fqs(emu, c4_records(), name)


@pytest.mark.parametrize("name", ["c5_pe4k_o_t1.fqs", "c5_pe4k_o_t4.fqs", "c5_pe4k_s_t1.fqs", "c5_pe4k_s_t4.fqs"])
def test_emu_matches_reference_paired_end(name):
    check_against_fqs_pe(emu, c5_records(), name)


@pytest.mark.parametrize("name", ["c5_pe4k_o_t4.fqs", "c5_pe4k_s_t4.fqs"])
def test_emu_paired_end_growth_posted_by_the_device_is_recovered(name, monkeypatch):
    """Paired-end encoding reads nothing back inside a block either: the partition kernel checks the k-mer tables' growth rule,
    the demand kernel the pair table's; a posted phase stops the block's queue and the host grows and resumes at that phase's
    inserts (pair table and k-mer tables).  From 256-slot k-mer sub-tables and 64-slot pair sub-tables both happen in most blocks."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_PTAB_INIT", "64")
    codec = check_against_fqs_pe(emu, c5_records(), name)
    cap = codec.capacity()
    assert cap["growths"] >= 8 and cap["pair_slots"] > 64 * 4, cap


@pytest.mark.parametrize("name", ["c5_pe4k_o_t4.fqs", "c5_pe4k_s_t4.fqs"])
def test_emu_paired_end_chunked_pair_table(name, monkeypatch):
    """chunked tables of a paired-end codec: the pair table's key and value arrays are chunked with the k-mer tables (one chunk per
    sub-table and array), and grow sub-table by sub-table"""
    monkeypatch.setenv("FQSX_GTAB_INIT", "256")
    monkeypatch.setenv("FQSX_PTAB_INIT", "64")
    monkeypatch.setenv("FQSX_CHUNK_STEP_KB", "1")
    monkeypatch.setenv("FQSX_CHUNKED_TABLES", "1")
    codec = check_against_fqs_pe(emu, c5_records(), name)
    cap = codec.capacity()
    assert cap["growths"] >= 8 and cap["pair_slots"] > 64 * 4, cap
    assert 0 <= cap["pair_bytes_held"] - 16 * cap["pair_slots"] <= 2 * 4 * 4096, cap


@pytest.mark.parametrize("name", ["c20_pelong_o_t2.fqs", "c20_pelong_s_t2.fqs"])
def test_emu_paired_end_mates_longer_than_the_lds_staging(name):
    """5000 bp mates (the reference takes up to 2^24 bases, fqs/meta.cpp:69): code lines and the reverse-complement line of an
    anchored second mate in the worker's HBM scratch instead of LDS; encode against the reference's file, then decode it"""
    check_against_fqs_pe(emu, c20_records(), name)
    check_decode_fqs(emu, c20_records(), name)


@pytest.mark.parametrize("name", ["c7_mixedlen_o_t3.fqs", "c7_mixedlen_s_t3.fqs"])
def test_emu_matches_reference_short_and_long_reads(name):
    check_against_fqs(emu, c7_records(), name)


def test_emu_matches_reference_large_k_geometry():
    check_against_digest(emu, "c6_20k_gs300_s_t2.json")


def test_emu_matches_reference_default_geometry():
    """-gs 3100 (the reference's default, BASELINE configs[3]): k = 13/18/21/27, 16 GiB p-mer vector (untouched pages stay
    unmapped), partial look-ups of up to 1024 trials."""
    check_against_digest(emu, "c9_20k150_gs3100_s_t2.json")


@pytest.mark.parametrize("name,recs", [("c1_10k_o_t4.fqs", c1_records), ("c1_10k_s_t4.fqs", c1_records), ("c4_ragged_s_t3.fqs", c4_records),
                                       ("c7_mixedlen_o_t3.fqs", c7_records), ("c7_mixedlen_s_t3.fqs", c7_records),
                                       ("c5_pe4k_o_t4.fqs", c5_records), ("c5_pe4k_s_t4.fqs", c5_records)])
def test_emu_decodes_reference_streams(name, recs):
    check_decode_fqs(emu, recs(), name)


def test_emu_matches_reference_150bp():
    check_against_digest(emu, "c3_50k150_s_t8.json")


def test_emu_matches_oracle_many_workers_and_tiny_blocks():
    rec = c4_records()
    for T, mode in [(7, "se_sorted"), (16, "se_original")]:
        header = hp.make_header(T, mode, 1)
        a, b = emu(header), OracleCodec(header)
        blks = hp.form_blocks(rec, mode)[:12] if mode == "se_sorted" else [np.arange(0, 700), np.arange(700, 731), np.arange(731, 733)]
        for g, idx in enumerate(blks):
            bases, off = hp.block_arrays(rec, np.asarray(idx))
            assert a.encode_block(bases, off, g) == b.encode_block(bases, off, g)


def test_abi_rejects_bad_headers():
    from fqsqueezer_amd.codec import FqsxError
    good = hp.make_header(2, "se_sorted", 1)
    for bad in [b"XCSD" + good[4:], good[:4] + bytes([0]) + good[5:], good[:5] + bytes([4]) + good[6:]]:
        with pytest.raises(FqsxError):
            emu(bad)


@pytest.mark.parametrize("name", ["c1_10k_s_t4.fqs", "c5_pe4k_s_t4.fqs"])
def test_overflow_chains_of_the_two_choice_tables(name, monkeypatch):
    """The global k-mer tables are two-choice buckets (four slots, two hashed buckets per k-mer, new keys into the emptier one); a key
    whose two buckets are both full goes down an overflow chain, which at the default 80 % load is rare.  Tables that start at 64
    slots per owner and are only grown at 85 % load (to 80 %) run hundreds of keys through the chains -- look-ups, in-order batch
    inserts, re-inserts at a growth -- and the streams must stay the reference's (results never depend on the layout)."""
    monkeypatch.setenv("FQSX_GTAB_INIT", "64")
    monkeypatch.setenv("FQSX_TAB_LOAD_PCT", "85")
    monkeypatch.setenv("FQSX_TAB_AFTER_PCT", "80")
    if name.startswith("c5"):
        check_against_fqs_pe(emu, c5_records(), name)
        check_decode_fqs(emu, c5_records(), name)        # (the decoder walks the same chains)
    else:
        codec = check_against_fqs(emu, c1_records(), name)
        cap = codec.capacity()
        assert cap["bytes_per_bmer"] < 10.2 and cap["growths"] >= 20, cap   # (8 bytes per slot at >= 80 % load)
        check_decode_fqs(emu, c1_records(), name)
