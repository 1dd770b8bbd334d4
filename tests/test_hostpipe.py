"""Host logic: header, varint, partition/sync schedule, block formation, container round trip."""
import os

import numpy as np
import pytest

from conftest import GOLD, c1_records, c4_records
from fqsqueezer_amd import hostpipe as hp


def test_varint_roundtrip():
    for x in [0, 1, 0x7FFF, 0x8000, 0x3FFFFF, 0x400000, 0x3FFFFFFF, 12345, 5000000]:
        b = hp.put_varint(x)
        assert hp.get_varint(b, 0) == (x, len(b))
    assert hp.put_varint(10000) == bytes([0x27, 0x10])          # SURVEY.md App. A example
    with pytest.raises(ValueError):
        hp.put_varint(0x40000000)


def test_header_matches_reference():
    hdr, _ = hp.parse_fqs(open(os.path.join(GOLD, "c1_10k_s_t4.fqs"), "rb").read())
    assert hdr == hp.make_header(4, "se_sorted", 1)
    hdr, _ = hp.parse_fqs(open(os.path.join(GOLD, "c1_10k_o_t1.fqs"), "rb").read())
    assert hdr == hp.make_header(1, "se_original", 1)
    assert hp.kmer_lengths(3100)[:4] == (13, 18, 21, 27)
    assert hp.kmer_lengths(8)[:4] == (10, 15, 18, 21)


def test_partition_and_sync_schedule():
    assert hp.partition_for_workers(10, 4) == [(0, 2), (2, 4), (4, 6), (6, 10)]
    assert hp.partition_for_workers(7, 2) == [(0, 2), (2, 7)]
    assert hp.no_synchronizations(0, 10000, 1) == 99
    assert hp.no_synchronizations(0, 27, 1) == 12
    assert hp.no_synchronizations(99, 10000, 4) == 0
    assert hp.no_synchronizations(100, 10000, 4) == 0
    assert hp.no_synchronizations(5, 3, 4) == 0


@pytest.mark.parametrize("name", ["c1_10k_o_t4.fqs", "c1_10k_s_t4.fqs", "c4_ragged_s_t3.fqs", "c4_ragged_o_t3.fqs"])
def test_block_formation_matches_reference(name):
    rec = c4_records() if name.startswith("c4") else c1_records()
    data = open(os.path.join(GOLD, name), "rb").read()
    header, blocks = hp.parse_fqs(data)
    mode = "se_sorted" if header[5] == 1 else "se_original"
    blks = hp.form_blocks(rec, mode)
    assert [len(b) for b in blks] == [b.n_reads for b in blocks]
    # worker offsets recorded in the container = byte offset of the worker's first read in the block buffer
    # (original order only: in sorted mode reads with identical DNA are ordered by libstdc++'s unstable
    #  std::sort, which moves id lengths -- and hence these offsets -- but not the DNA stream)
    sizes = rec.record_sizes()
    for idx, blk in zip(blks, blocks if mode == "se_original" else []):
        cs = np.concatenate([[0], np.cumsum(sizes[idx])])
        for (first, _), off in zip(hp.partition_for_workers(len(idx), header[4]), blk.offsets):
            assert cs[first] == off
    assert hp.write_fqs(header, blocks) == data                   # container writer is the parser's inverse


def test_sorted_order_is_sorted():
    rec = c4_records()
    order = np.concatenate(hp.sorted_order(rec))
    assert sorted(order.tolist()) == list(range(len(rec)))
    nt = bytes(i if i in b"ACG" else ord("T") for i in range(256))
    keys = [(rec.seq[i].translate(nt), len(rec.seq[i]), rec.seq[i]) for i in order]
    assert keys == sorted(keys)


def test_record_sizes_count_the_separator_line(tmp_path):
    """read_size() (defs.h:79-81) spans all four lines, so a `+id` separator line moves the block boundaries
    (reads_block.h:119-139) although its text is not stored: read_fastq keeps the per-record lengths."""
    p = tmp_path / "a.fq"
    recs = [b"@r1\nACGT\n+r1\nIIII\n", b"@r2 x\nACGTA\n+\nIIIII\n", b"@r3\nAC\n+r3 again\nII\n"]
    p.write_bytes(b"".join(recs))
    rec = hp.read_fastq(str(p))
    assert rec.record_sizes().tolist() == [len(r) for r in recs]
    q = tmp_path / "b.fq"
    q.write_bytes(b"@r1\nACGT\n+\nIIII\n")
    assert hp.read_fastq(str(q)).plus_len is None
