"""Complete .fqs files written around the DNA path: byte identity with the reference's own files and
round trip through the unmodified reference decompressor (oracle/_ref/fqs-1.1 d)."""
import os
import subprocess

import numpy as np
import pytest

import json

from conftest import EMU_LIB, GOLD, IM, QM, ROOT, c1_records, c4_records, c10_records, c11_records, check_full_file_digest, digest_records, pe_digest_records
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.fqsfile import compress_records, compress_records_pe

REF = os.path.join(ROOT, "oracle", "_ref", "fqs-1.1")


def _decode_with_reference(data: bytes, tmp_path):
    f = tmp_path / "x.fqs"
    f.write_bytes(data)
    out = tmp_path / "x.fq"
    subprocess.check_call([REF, "d", "-out", str(out), str(f)], stdout=subprocess.DEVNULL)
    return out.read_bytes().split(b"\n")[1::4]


def _lib(request):
    return None if request.node.get_closest_marker("gpu") else EMU_LIB


@pytest.mark.parametrize("name,rec_fn,T,gs", [("c4_ragged_o_t3.fqs", c4_records, 3, 1), ("c1_10k_o_t4.fqs", c1_records, 4, 1)])
def test_file_identical_to_reference_original_order(built, name, rec_fn, T, gs):
    data = compress_records(rec_fn(), T, "o", gs, lib_path=EMU_LIB)
    assert data == open(os.path.join(GOLD, name), "rb").read()


def test_default_mode_file_identical_to_reference(built):
    """-qm o -im o: ids, qualities, read lengths and DNA -- the whole file byte for byte."""
    data = compress_records(c10_records(), 3, "o", 1, lib_path=EMU_LIB, quality_mode="lossless", id_mode="lossless")
    assert data == open(os.path.join(GOLD, "c10_full_o_t3.fqs"), "rb").read()


FULL_SE = [("c10_full_s_i8_t4.json", "s", "8", "i", 4), ("c10_full_s_oo_t2.json", "s", "o", "o", 2), ("c10_full_o_i2_t5.json", "o", "2", "i", 5)]
FULL_PE = [("c11_pe_full_s_o4_t3.json", "s", "4", "o", 3), ("c11_pe_full_o_io_t2.json", "o", "o", "i", 2)]


@pytest.mark.parametrize("name,om,qm,im,T", FULL_SE)
def test_full_mode_files_match_reference_digests(built, name, om, qm, im, T):
    check_full_file_digest(compress_records(c10_records(), T, om, 1, lib_path=EMU_LIB, quality_mode=QM[qm], id_mode=IM[im]), name)


@pytest.mark.parametrize("name,om,qm,im,T", FULL_PE)
def test_full_mode_paired_files_match_reference_digests(built, name, om, qm, im, T):
    r1, r2 = c11_records()
    check_full_file_digest(compress_records_pe(r1, r2, T, om, 1, lib_path=EMU_LIB, quality_mode=QM[qm], id_mode=IM[im]), name)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built")
@pytest.mark.parametrize("order", ["o", "s"])
def test_reference_decoder_reads_our_file(built, tmp_path, order):
    rec = c4_records()
    data = compress_records(rec, 5, order, 1, lib_path=EMU_LIB)
    got = _decode_with_reference(data, tmp_path)
    if order == "o":
        want = [rec.seq[i] for i in range(len(rec))]
    else:
        want = [rec.seq[int(i)] for i in np.concatenate(hp.sorted_order(rec))]
    assert got[:len(want)] == want


@pytest.mark.gpu
def test_gpu_full_mode_files_identical_to_reference():
    data = compress_records(c10_records(), 3, "o", 1, quality_mode="lossless", id_mode="lossless")
    assert data == open(os.path.join(GOLD, "c10_full_o_t3.fqs"), "rb").read()
    name, om, qm, im, T = FULL_SE[0]
    check_full_file_digest(compress_records(c10_records(), T, om, 1, quality_mode=QM[qm], id_mode=IM[im]), name)
    name, om, qm, im, T = FULL_PE[0]
    r1, r2 = c11_records()
    check_full_file_digest(compress_records_pe(r1, r2, T, om, 1, quality_mode=QM[qm], id_mode=IM[im]), name)


@pytest.mark.gpu
def test_gpu_file_identical_to_reference_and_decodes(tmp_path):
    data = compress_records(c1_records(), 4, "o", 1)
    assert data == open(os.path.join(GOLD, "c1_10k_o_t4.fqs"), "rb").read()
    if os.path.exists(REF):
        rec = c4_records()
        got = _decode_with_reference(compress_records(rec, 64, "s", 1), tmp_path)
        assert got[:len(rec)] == [rec.seq[int(i)] for i in np.concatenate(hp.sorted_order(rec))]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c14_pe150_s_q8_t8.json", "c15_pe150_s_oo_t4.json"])
def test_gpu_paired_end_150bp_files_match_reference(name):
    """BASELINE configs[2]'s shape (150 bp pairs, -om s -qm 8; 100 k pairs, T = 8) and configs[4]'s modes (-qm o -im o):
    every stream of every block, then the whole file, against the reference's own file."""
    d = json.load(open(os.path.join(GOLD, name)))
    r1, r2 = pe_digest_records(d)
    check_full_file_digest(compress_records_pe(r1, r2, d["threads"], d["om"], d["gs"], quality_mode=QM[d["qm"]], id_mode=IM[d["im"]]), name)


@pytest.mark.gpu
def test_gpu_sorted_order_of_1M_reads_is_the_reference_order():
    """N3 against the reference itself: the GPU sort pre-pass orders the metric's 1 M x 150 bp file; the id stream (which
    sees the order of equal reads too) and the DNA stream of the resulting file match the reference's `-om s -im o` file."""
    name = "c16_1M150_s_ids_t8.json"
    d = json.load(open(os.path.join(GOLD, name)))
    check_full_file_digest(compress_records(digest_records(d), d["threads"], "s", d["gs"], quality_mode="none", id_mode="lossless"), name)
