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
