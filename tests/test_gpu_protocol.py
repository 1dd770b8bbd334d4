"""The fall-back branches of the eight-wave worker protocol, taken deterministically.

In the product run, which wave settles what depends on wave timing: whether a scout made a chunk or the resolving
wave went on without it (sc_abandoned), where a correction window (the resolving wave's own stage P for the bmer - 1
positions after a k-mer correction) ends inside a scout's chunk, whether a local look-up finds its entries applied
already.  The results must not depend on any of that.  DevCfg.dbg (env FQSX_PROTO_DEBUG, read by
fqsx_dna_create) forces each branch for a whole run; every run must reproduce the reference goldens bit for bit.
Only the real kernels have these waves, so this is a GPU test (the emulation build runs every role inline)."""
import pytest

from conftest import (c1_records, c4_records, c5_records, c7_records, check_against_digest, check_against_fqs,
                      check_against_fqs_pe)

pytestmark = pytest.mark.gpu

SCOUTS_OFF, ABANDON, RESTART, INSERTER_STALL = 1, 2, 4, 8
MODES = [SCOUTS_OFF, ABANDON, RESTART, INSERTER_STALL, ABANDON | RESTART | INSERTER_STALL]
IDS = ["scouts_off", "abandon_every_3rd_read", "window_after_every_chunk", "inserter_stalled", "abandon+window+stall"]


def gpu(header):
    from fqsqueezer_amd.codec import DnaCodec
    return DnaCodec(header, device=0)


@pytest.fixture(params=MODES, ids=IDS)
def proto(request, monkeypatch):
    monkeypatch.setenv("FQSX_PROTO_DEBUG", str(request.param))
    return request.param


def test_forced_branches_10k_sorted(proto):
    check_against_fqs(gpu, c1_records(), "c1_10k_s_t4.fqs")


def test_forced_branches_ragged_reads_with_n_runs_and_duplicates(proto):
    check_against_fqs(gpu, c4_records(), "c4_ragged_s_t3.fqs")


def test_forced_branches_short_and_long_reads(proto):
    check_against_fqs(gpu, c7_records(), "c7_mixedlen_s_t3.fqs")


def test_forced_branches_saturated_counters(proto):
    """c13: the merges of the partial look-ups and the Hamming-1 sweeps draw from the RNG streams here, so the order in
    which the resolving wave consumes probes made by itself / by a scout is visible in the bitstream."""
    check_against_digest(gpu, "c13_sat_s_t4.json", max_blocks=120)


def test_forced_branches_metric_workload_first_blocks(proto):
    """the first 40 blocks of the benchmark's own file at T = 64 (30 synchronisation phases each)"""
    check_against_digest(gpu, "c12_1M150_s_t64.json", max_blocks=40)


def test_forced_branches_request_mode_scouts_original_order(proto):
    check_against_fqs(gpu, c1_records(), "c1_10k_o_t4.fqs")


def test_forced_branches_request_mode_scouts_paired_end(proto):
    check_against_fqs_pe(gpu, c5_records(), "c5_pe4k_s_t4.fqs")
