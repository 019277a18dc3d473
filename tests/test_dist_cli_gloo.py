"""The multi-rank driver on CPU: 1 and 2 processes over gloo, the kernels from the SIMT-emulator build; files compared
byte for byte with the single-process command line (wgatools_emu) and with the oracle's expectation."""
import pytest

import dist_cli_cases as dc
from wgatools_amd import build


@pytest.fixture(scope="module")
def libs():
    return build.build_emu(), build.build_cli_emu()


def test_paf2maf_ordered_output(libs, tmp_path):
    dc.check_paf2maf(tmp_path, libs[0], libs[1], (1, 2), 29600)


def test_paf2maf_first_error_in_input_order(libs, tmp_path):
    dc.check_paf2maf_error(tmp_path, libs[0], (1, 2), 29610)


def test_pafcov_sharded_and_spread(libs, tmp_path):
    dc.check_pafcov(tmp_path, libs[0], libs[1], (1, 2), 29620)


def test_totals_allreduce(libs, tmp_path):
    dc.check_totals(tmp_path, libs[0], (1, 2), 29650)


def test_bad_cigar_ends_every_rank(libs, tmp_path):
    dc.check_bad_cigar_ends_every_rank(tmp_path, libs[0], (1, 2), 29660)


def test_compressed_output_names_are_refused(libs, tmp_path):
    """the ranks write records at their offsets in the plain output: a `.gz` name is refused before anything starts (the
    single-process command line is the one that deflates, on the device)"""
    gz = str(tmp_path / "out.maf.gz")
    r = dc.launch(1, libs[0], 29670, "paf2maf", "nothing.paf", "-g", "t.fa", "-q", "q.fa", "-o", gz, expect_rc=1)
    assert "plain offsets" in r.stderr and not (tmp_path / "out.maf.gz").exists()
