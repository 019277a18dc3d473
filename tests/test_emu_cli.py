"""Host-logic check on CPU: the `wgatools` host code linked against the emulator build of the
kernels (tests/emu/wgatools_emu).  Same cases as the GPU run."""
import pytest

from wgatools_amd import build
from cli_cases import *  # noqa: F401,F403  (the test functions)


@pytest.fixture(scope="module")
def cli():
    return build.build_cli_emu()
