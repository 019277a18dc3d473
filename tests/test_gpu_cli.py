"""End-to-end drop-in check on a real GPU: the `wgatools` binary (C++ host layer + libwgahip.so)."""
import os

import pytest

from wgatools_amd import build
from cli_cases import *  # noqa: F401,F403  (the test functions)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(build.CLI_BIN):
        build.build_cli()
    return build.CLI_BIN


def test_dist_cli_one_rank_writes_the_command_lines_bytes(cli, tmp_path):
    """the multi-rank driver (wgatools_amd/dist_cli.py) with one rank on libwgahip.so: same files as the `wgatools`
    binary; 2 ranks run over gloo on the emulator build in test_dist_cli_gloo.py, N GPUs over RCCL are the driver's"""
    import dist_cli_cases as dc
    dc.check_paf2maf(tmp_path, None, cli, (1,), 29700)
    dc.check_paf2maf_error(tmp_path, None, (1,), 29710)
    dc.check_pafcov(tmp_path, None, cli, (1,), 29720)
    dc.check_totals(tmp_path, None, (1,), 29740)
