"""The oracle is the yardstick of every parity test and cannot be pinned to a run of the reference, so
its own memory safety is checked: the golden suite and a slice of the oracle-heavy parity cases run once against an
AddressSanitizer + UndefinedBehaviorSanitizer build of oracle.c (`make -C oracle asan`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update(WGA_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"), LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_emu_parity.py"), "-k",
                        "(golden or oracle or readme or html or stat_random or paf2maf_edge or pafcov or pafpseudo or "
                        "maf_pair or call or tokeniser or chain or dotplot) and not (look_back or many_windows or long or "
                        "piece or random_bytes or random_shapes or nasty or hundreds or steps or without or 2-200 or stream_kernel)"],       # the emulator-bound cases add nothing for the oracle
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "passed" in r.stdout and "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]
