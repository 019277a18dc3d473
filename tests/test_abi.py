"""The C-ABI library loads and exports every symbol include/wga_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from wgatools_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "wga_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wga_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_binding():
    assert header_functions() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_symbol():
    lib = build.build_hip()
    h = ctypes.CDLL(lib)
    for name in header_functions():
        assert hasattr(h, name), name
    h.wga_abi_version.restype = ctypes.c_int
    assert h.wga_abi_version() == 3


def test_no_gpu_fails_loudly():
    """without a device the product refuses to create a context — there is no CPU fallback"""
    lib = _lib.load(build.build_hip())
    if lib.wga_device_count() > 0:
        pytest.skip("a GPU is visible")
    ctx = ctypes.c_void_p()
    assert lib.wga_ctx_create(0, ctypes.byref(ctx)) == -2
    assert b"no HIP device" in lib.wga_last_error()


def test_host_packer():
    lib = _lib.load(build.build_hip())
    from wgatools_amd.engine import Engine
    e = Engine.__new__(Engine)
    e.lib = lib
    ops, err, tok = Engine.pack_cigar(e, "10M2I3D4=5X")
    assert err == 0 and [(int(w) >> 4, int(w) & 15) for w in ops] == [(10, 0), (2, 1), (3, 2), (4, 7), (5, 8)]
    # lengths >= 2^28 are split; I/D pieces after the first carry continuation codes
    ops, err, _ = Engine.pack_cigar(e, "600000000I5N")
    assert err == 0 and [int(w) & 15 for w in ops] == [1, 9, 9, 3]
    assert sum(int(w) >> 4 for w in ops[:3]) == 600000000
    assert Engine.pack_cigar(e, "10M5")[1] == 2          # CigarOpInvalid("")
    assert Engine.pack_cigar(e, "10MM")[1:] == (2, (2, 2))
    assert Engine.pack_cigar(e, "M")[1] == 3             # ParseIntError("")
    assert Engine.pack_cigar(e, "")[1] == 6              # the reference panics on an empty CIGAR
    assert Engine.pack_cigar(e, "3é")[1] == 0       # one multi-byte char is one (OTHER) op
    assert int(Engine.pack_cigar(e, "3é")[0][0]) & 15 == 11


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing in the package imports, links or loads it (build.py only
    knows how to compile it for the tests), and neither product binary depends on it"""
    import subprocess
    pkg = os.path.join(ROOT, "wgatools_amd")
    seen = 0
    for where in (pkg, os.path.join(ROOT, "include")):
        for dirpath, dirs, files in os.walk(where):
            dirs[:] = [d for d in dirs if d != "__pycache__"]
            for f in files:
                path = os.path.join(dirpath, f)
                if f == "build.py" or f.endswith((".so", ".o", ".a", ".pyc", ".hsaco", ".co")) or os.path.dirname(path).endswith("bin"):
                    continue                      # built artefacts are checked through their dynamic sections below
                raw = open(path, "rb").read()
                if b"\0" in raw[:4096]:
                    continue                      # some other binary
                seen += 1
                assert not re.search(r"oracle_py|liboracle|\borc_|oracle/", raw.decode(errors="replace")), path
    assert seen > 40                              # every source of the package, whatever its extension (.inc, .hip, ...)
    for binary in (build.build_hip(), build.build_cli()):
        needed = subprocess.run(["readelf", "-d", binary], stdout=subprocess.PIPE).stdout.decode()
        assert "oracle" not in needed, binary


def test_struct_layouts_are_frozen(tmp_path):
    """tests/abi_layout.c: _Static_assert on the size of and every offset in the structs of include/wga_hip.h — the numbers a
    binding in another language has to reproduce (INTEGRATION.md section 2 quotes them)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c11", "-I" + os.path.join(root, "include"), "-c", os.path.join(root, "tests", "abi_layout.c"),
                        "-o", str(tmp_path / "abi_layout.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
