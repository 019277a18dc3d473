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


def test_dist_cli_two_ranks_over_rccl(cli, tmp_path):
    """the multi-rank driver with TWO ranks, one per GPU, over RCCL (backend "nccl"): the size all-reduce that orders the
    output, the totals all-reduce and the reduce-scatter of a spread pafcov target cross xGMI.  Skips on a one-GPU box; the
    same cases run with two ranks over gloo on the emulator build (test_dist_cli_gloo.py)"""
    import ctypes
    import torch  # noqa: F401
    import dist_cli_cases as dc
    if ctypes.CDLL(build.HIP_LIB).wga_device_count() < 2:
        pytest.skip("one device on this box: two ranks over RCCL need two")
    dc.check_paf2maf(tmp_path, None, cli, (2,), 29800)
    dc.check_pafcov(tmp_path, None, cli, (2,), 29820)
    dc.check_totals(tmp_path, None, (2,), 29840)


def test_gpus_flag_with_the_devices_of_this_box(cli, tmp_path):
    """`wgatools --gpus N` (C++ worker threads, one context per device) with every device this box has — one on the
    driver's boxes, where the sharded path then runs with a single worker: the same bytes as the plain command line and
    the oracle; one device more than visible is refused.  2 and 3 devices run on the emulator build
    (test_emu_cli_multi.py)."""
    import ctypes
    import torch  # noqa: F401  first, as in conftest._gpu_engine(): libwgahip.so then shares torch's HIP runtime in this process
    import multi_gpu_cli_cases as mc
    have = ctypes.CDLL(build.HIP_LIB).wga_device_count()
    assert have >= 1
    env = dict(os.environ)
    gpus = tuple(sorted({1, have, min(have, 2)}))
    mc.check_paf2maf(cli, tmp_path, gpus, env)
    mc.check_paf2maf_errors(cli, tmp_path, gpus, env)
    mc.check_stat(cli, tmp_path, gpus, env)
    mc.check_pafcov(cli, tmp_path, gpus, env)
    mc.check_call_paf(cli, tmp_path, gpus, env)
    mc.check_too_many(cli, env, have)


def test_pafpseudo_walk_at_a_million_records(cli, tmp_path):
    """BASELINE configs[4]'s all-to-all part at a size where the grouping / sorted insertion / contained-skip / overlap-trim
    walk (pseudomaf.rs:25-42,86-95,147-210) does real work: 1.2 M records over 16 x 64 (target, query) pairs; one target's
    file (64 query rows, ~70 000 records) is compared with the oracle's rows"""
    from cli_cases import pafpseudo_walk_case
    n, checked = pafpseudo_walk_case(cli, tmp_path, 1_200_000, 16, 64, 1)
    assert n > 1_000_000 and checked > 50_000
