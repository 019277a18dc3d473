"""Host-logic check on CPU: the `wgatools` host code linked against the emulator build of the
kernels (tests/emu/wgatools_emu).  Same cases as the GPU run."""
import pytest

from wgatools_amd import build
from cli_cases import *  # noqa: F401,F403  (the test functions)


@pytest.fixture(scope="module")
def cli():
    return build.build_cli_emu()


def _reader_outputs(cli, hook, path, env_name):
    import os, subprocess
    a = subprocess.run([cli, hook, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([cli, hook, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **{env_name: "host"}))
    strip = lambda r: (r.returncode, r.stdout.split(b"\n", 1)[1] if b"\n" in r.stdout else r.stdout,
                       r.stderr.split(b" ERROR ", 1)[-1])
    return a, strip(a), strip(b)


def test_readers_agree_on_random_files(cli, tmp_path):
    """whatever the bytes, the device splitters + fallback give what the host readers give: same records or same error"""
    from hypothesis import given, settings, strategies as st, HealthCheck
    atoms_paf = ["\t", "\n", "\r\n", "q", "chr1", "12", "0", "+", "-", "+7", "x", '"', "#", "cg:Z:5=2X", "cs:Z::5", "tp:A:P",
                 "18446744073709551616", " ", "é", "\r"]
    good_paf = "q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tcg:Z:10=\n"
    atoms_maf = ["\n", "\r\n", "s", " ", "\t", "a score=1", "ref.chr", "12", "+", "-", "ACGT-", "x", "é", "#", "i", "\x0b"]
    good_maf = "a score=0\ns t 1 5 + 100 ACGT-\ns q 2 4 - 90 AC-TG\n\n"
    path = str(tmp_path / "r.txt")

    @settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(st.one_of(st.sampled_from(atoms_paf), st.just(good_paf)), max_size=40))
    def run_paf(parts):
        open(path, "w", encoding="utf-8", newline="").write("".join(parts))
        _, dev, host = _reader_outputs(cli, "__paf_reader", path, "WGA_PAF_READER")
        assert dev == host, ("".join(parts), dev, host)

    @settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(st.one_of(st.sampled_from(atoms_maf), st.just(good_maf)), max_size=40))
    def run_maf(parts):
        open(path, "w", encoding="utf-8", newline="").write("##maf version=1\n" + "".join(parts))
        _, dev, host = _reader_outputs(cli, "__maf_reader", path, "WGA_MAF_READER")
        assert dev == host, ("".join(parts), dev, host)

    run_paf()
    run_maf()
