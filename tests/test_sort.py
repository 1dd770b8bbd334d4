"""Row N3: read order of sorted mode from the GPU pre-pass (radix sort + ranks on the device, std::sort replay on the
host) against the reference-equivalent host sort (fqsx_sort_bin = libstdc++ std::sort with the reference's comparator)."""
import numpy as np
import pytest

from conftest import EMU_LIB, c1_records, c4_records, c7_records, c10_records
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import sort_order


def _check(rec, lib, max_batch_bases=0, min_batches=1):
    bases, off = hp.block_arrays(rec, np.arange(len(rec), dtype=np.int64))
    st = {}
    got = sort_order(bases, off, lib_path=lib, max_batch_bases=max_batch_bases, stats=st)
    want = hp.sorted_order_exact(rec)
    assert len(got) == len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), f"bin #{b}: order differs from the reference's std::sort"
    assert st["batches"] >= min_batches, st


def _odd_records():
    """Reads with every tie-break the comparator has: N vs T, other IUPAC / lower-case bytes (all 'code 3'), prefixes of
    one another, exact duplicates, lengths 4..300."""
    rng = np.random.Generator(np.random.PCG64(77))
    alpha = np.frombuffer(b"ACGTACGTACGTACGTNNRYKMacgtn", dtype=np.uint8)
    seqs = []
    for i in range(6000):
        L = int(rng.integers(4, 60)) if i % 3 else int(rng.integers(60, 300))
        seqs.append(alpha[rng.integers(0, len(alpha), L)].tobytes())
    for i in range(0, 3000, 3):
        seqs.append(seqs[i])                               # duplicates
        seqs.append(seqs[i + 1][:max(4, len(seqs[i + 1]) // 2)])    # proper prefixes
        seqs.append(seqs[i + 2].replace(b"N", b"T"))       # equal under N->T
    ids = [b"@r%d" % i for i in range(len(seqs))]
    return hp.Records(ids, seqs, [b"I" * len(s) for s in seqs])


@pytest.mark.parametrize("rec_fn", [c4_records, c7_records, c10_records, _odd_records])
def test_emu_sort_order_equals_reference_sort(built, rec_fn):
    _check(rec_fn(), EMU_LIB)


@pytest.mark.parametrize("rec_fn,limit", [(c7_records, 40000), (_odd_records, 1), (c4_records, 10 ** 9)])
def test_emu_batched_sort_order_equals_reference_sort(built, rec_fn, limit):
    """bounded device memory (fqsx_sort_order_batched): bins packed into batches of at most `limit` bases -- every bin on its
    own for limit 1 -- give the order of the one-allocation sort"""
    _check(rec_fn(), EMU_LIB, max_batch_bases=limit, min_batches=1 if limit > 10 ** 8 else 8)


@pytest.mark.gpu
def test_gpu_batched_sort_order_1M_reads():
    """1 M x 100 bp in batches of at most 8 Mbases (>= 13 uploads instead of one 100 MB allocation)"""
    from fqsqueezer_amd.synth import synth_reads
    reads = synth_reads(1000000, 100, 5000000, 2)
    rec = hp.Records([b""] * len(reads), reads, reads)
    _check(rec, None, max_batch_bases=8_000_000, min_batches=13)


@pytest.mark.gpu
@pytest.mark.parametrize("rec_fn", [c1_records, c4_records, c7_records, c10_records, _odd_records])
def test_gpu_sort_order_equals_reference_sort(rec_fn):
    _check(rec_fn(), None)


@pytest.mark.gpu
def test_gpu_sort_order_1M_reads():
    from fqsqueezer_amd.synth import synth_reads
    reads = synth_reads(1000000, 100, 5000000, 2)
    rec = hp.Records([b""] * len(reads), reads, reads)
    _check(rec, None)
