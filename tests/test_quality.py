"""Quality stream: oracle restatement, emulated kernel and HIP kernel against the
reference's quality streams (all four quality modes, single- and paired-end)."""
import pytest

from conftest import EMU_LIB, check_quality_digest

CASES = ["c8_qual_o_t4.json", "c8_qual_8_t4.json", "c8_qual_4_t4.json", "c8_qual_2_t4.json", "c8_qual_pe8_t3.json"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_quality_matches_reference(name):
    from oracle.pyoracle import OracleQual
    check_quality_digest(OracleQual, name)


@pytest.mark.parametrize("name", CASES)
def test_emu_quality_matches_reference(built, name):
    from fqsqueezer_amd.codec import QualCodec
    check_quality_digest(lambda h: QualCodec(h, lib_path=EMU_LIB), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_quality_matches_reference(name):
    from fqsqueezer_amd.codec import QualCodec
    check_quality_digest(lambda h: QualCodec(h, device=0), name)
