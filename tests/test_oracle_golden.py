"""Pins the oracle (CPU restatement, oracle/fqs_oracle.cpp) to outputs of the unmodified reference."""
import ctypes as C

import numpy as np
import pytest

from conftest import c20_records, c1_records, c4_records, c5_records, c7_records, check_against_digest, check_against_fqs, check_against_fqs_pe, check_decode_fqs
from oracle.pyoracle import OracleCodec, lib


@pytest.mark.parametrize("name", ["c1_10k_o_t1.fqs", "c1_10k_o_t4.fqs", "c1_10k_s_t1.fqs", "c1_10k_s_t4.fqs"])
def test_oracle_matches_reference_10k(name):
    check_against_fqs(OracleCodec, c1_records(), name)


@pytest.mark.parametrize("name", ["c4_ragged_o_t3.fqs", "c4_ragged_s_t3.fqs"])
def test_oracle_matches_reference_ragged(name):
    check_against_fqs(OracleCodec, c4_records(), name)


@pytest.mark.parametrize("name", ["c5_pe4k_o_t1.fqs", "c5_pe4k_o_t4.fqs", "c5_pe4k_s_t1.fqs", "c5_pe4k_s_t4.fqs"])
def test_oracle_matches_reference_paired_end(name):
    check_against_fqs_pe(OracleCodec, c5_records(), name)


@pytest.mark.parametrize("name", ["c20_pelong_o_t2.fqs", "c20_pelong_s_t2.fqs"])
def test_oracle_matches_reference_long_paired_end_mates(name):
    check_against_fqs_pe(OracleCodec, c20_records(), name)


@pytest.mark.parametrize("name", ["c7_mixedlen_o_t3.fqs", "c7_mixedlen_s_t3.fqs"])
def test_oracle_matches_reference_short_and_long_reads(name):
    check_against_fqs(OracleCodec, c7_records(), name)


def test_oracle_matches_reference_large_k_geometry():
    check_against_digest(OracleCodec, "c6_20k_gs300_s_t2.json")


@pytest.mark.slow
def test_oracle_matches_reference_default_geometry():
    check_against_digest(OracleCodec, "c9_20k150_gs3100_s_t2.json")   # ~100 s: the p-mer sweep covers 4^18 fields per worker


def test_oracle_matches_reference_150bp():
    check_against_digest(OracleCodec, "c3_50k150_s_t8.json")


@pytest.mark.slow
@pytest.mark.parametrize("t", [1, 8, 64])
def test_oracle_matches_reference_1M(t):
    check_against_digest(OracleCodec, f"c2_1M_s_t{t}.json")


@pytest.mark.parametrize("om", ["o", "s"])
def test_oracle_matches_reference_saturated_counters(om):
    """~12000x coverage of a two-haplotype genome: the fixture provably reaches the rarely taken branches -- both alleles
    of a site at the counter maximum (counts_level_t::mixed, dna.cpp:470-480), the uncorrected b-mer level (:489) and
    the probabilistic counters above their thresholds (utils.h:272-325) -- and the oracle still matches the reference."""
    codec = check_against_digest(OracleCodec, f"c13_sat_{om}_t4.json")
    lv = codec.levels()
    assert lv["mixed"] > 1000 and lv["bmer_unc"] > 0 and lv["draws_b"] > 1_000_000 and lv["draws_s"] > 1_000_000 and lv["draws_lb"] > 100_000
    if om == "o":
        assert lv["pmer"] > 100_000


@pytest.mark.slow
@pytest.mark.parametrize("t", [8, 64])
def test_oracle_matches_reference_1M_150bp(t):
    check_against_digest(OracleCodec, f"c12_1M150_s_t{t}.json")   # the metric's workload (bench.py default)


@pytest.mark.parametrize("name,recs", [("c1_10k_o_t4.fqs", c1_records), ("c1_10k_s_t4.fqs", c1_records), ("c4_ragged_s_t3.fqs", c4_records),
                                       ("c7_mixedlen_o_t3.fqs", c7_records), ("c5_pe4k_o_t4.fqs", c5_records), ("c5_pe4k_s_t4.fqs", c5_records)])
def test_oracle_decodes_reference_streams(name, recs):
    check_decode_fqs(OracleCodec, recs(), name)


def test_mt19937_known_answer():
    out = np.zeros(10000, dtype=np.uint32)
    lib().fqo_kat_mt19937(5489, 10000, out.ctypes.data)
    assert int(out[9999]) == 4123659995            # ISO C++ [rand.predef]: 10000th value of default mt19937
    rs = np.random.RandomState(5481)               # init_genrand seeding, same as std::mt19937(5481)
    ref = rs.randint(0, 2**32, size=2000, dtype=np.uint64).astype(np.uint32)
    lib().fqo_kat_mt19937(5481, 2000, out.ctypes.data)
    assert (out[:2000] == ref).all()


def _py_decode(data, script):
    """Independent Python decoder of the carry-less range coder (sub_rc.h:93-158)."""
    TOP, M64, MASK = 0x00ffffffffffff, 0xFF << 56, (1 << 64) - 1   # TopValue is 2^48-1 (14 hex digits, sub_rc.h:38)
    pos = 8
    buf = int.from_bytes(data[:8], "big")
    low, rng = 0, M64
    out = []
    for tot, cums in script:
        rng //= tot
        c = buf // rng
        sym = max(i for i, x in enumerate(cums[:-1]) if x <= c)
        out.append(sym)
        r = cums[sym] * rng
        buf -= r
        low = (low + r) & MASK
        rng *= cums[sym + 1] - cums[sym]
        while rng <= TOP:
            if (low ^ (low + rng)) & M64:
                rng = ((low | TOP) - low) & MASK
            buf = ((buf << 8) + data[pos]) & MASK
            pos += 1
            low = (low << 8) & MASK
            rng = (rng << 8) & MASK
    return out


def test_range_coder_roundtrip():
    rs = np.random.RandomState(3)
    n = 5000
    script, syms = [], []
    for _ in range(n):
        k = int(rs.randint(2, 6))
        f = rs.randint(1, 2000, size=k)
        cums = [0] + np.cumsum(f).tolist()
        script.append((cums[-1], cums))
        syms.append(int(rs.choice(k, p=f / f.sum())))
    freq = np.array([s[1][x + 1] - s[1][x] for s, x in zip(script, syms)], dtype=np.uint32)
    cum = np.array([s[1][x] for s, x in zip(script, syms)], dtype=np.uint32)
    tot = np.array([s[0] for s in script], dtype=np.uint32)
    out = np.zeros(4 * n + 64, dtype=np.uint8)
    ln = lib().fqo_kat_rc(n, freq.ctypes.data, cum.ctypes.data, tot.ctypes.data, out.ctypes.data, len(out))
    data = out[:ln].tobytes() + b"\0" * 16
    assert _py_decode(data, script) == syms


def test_counter_incrementer_properties():
    # Increment(c) below the threshold is exact; merges below the threshold are exact sums
    a = np.arange(0, 8, dtype=np.uint32)
    b = np.full(8, 0xFFFFFFFF, dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    lib().fqo_kat_cinc(7, 2, 63, 8, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert out.tolist() == [1, 2, 3, 4, 5, 6, 7, 8]
    a = np.array([0, 3, 7, 63, 63], dtype=np.uint32)
    b = np.array([5, 4, 0, 63, 0], dtype=np.uint32)
    out = np.zeros(5, dtype=np.uint32)
    lib().fqo_kat_cinc(7, 2, 63, 5, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert out.tolist()[:3] == [5, 7, 7] and out[3] == 63 and out[4] == 63
