"""Build helpers for the native pieces.

build_hip()    libwgahip.so  — the product: HIP kernels + C-ABI for gfx950 (hipcc cross-compiles
                               without a GPU).
build_oracle() oracle/liboracle.so — the CPU checker (tests / smoke / cpu_baseline only).
build_emu()    tests/emu/libwgaemu.so — the same kernel source under a SIMT emulator, used by the
                               CPU test-suite to check kernel logic; never loaded by the product.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wgatools_amd", "csrc")
HIP_LIB = os.path.join(ROOT, "wgatools_amd", "libwgahip.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libwgaemu.so")
CLI_EMU_BIN = os.path.join(ROOT, "tests", "emu", "wgatools_emu")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _locked(fn):
    """one build at a time across processes (pytest-xdist workers would otherwise link the same output at once, and run a
    binary while another worker rewrites it)"""
    import fcntl
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        if _lock_depth[0]:     # a build that builds what it links against
            return fn(*a, **kw)
        with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            _lock_depth[0] += 1
            try:
                return fn(*a, **kw)
            finally:
                _lock_depth[0] -= 1
                fcntl.flock(lk, fcntl.LOCK_UN)
    return wrapper


_lock_depth = [0]


def _run(cmd, cwd=None):
    """runs a compiler; an `-o <target>` is produced next to the target and renamed over it (never a half-written file)"""
    cmd = list(cmd)
    final = None
    if "-o" in cmd:
        k = cmd.index("-o") + 1
        final = cmd[k]
        cmd[k] = final + ".tmp%d" % os.getpid()
    r = _run_raw(cmd, cwd)
    if final:
        os.replace(cmd[cmd.index("-o") + 1], final)
    return r


def _run_raw(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [
        os.path.join(ROOT, "include", "wga_hip.h")
    ]


@_locked
def build_hip(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not force and not _newer(HIP_LIB, _sources()):
        return HIP_LIB
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libwgahip.so cannot be built here")
    objs = []
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall"]
    flags += os.environ.get("WGA_EXTRA_FLAGS", "").split()  # A/B builds (see build_hip_variant)
    for src, xflag in (("wga_capi.cpp", ["-x", "hip"]), ("wga_pack.cpp", [])):
        obj = os.path.join(CSRC, src.replace(".cpp", ".o"))
        out = _run([hipcc] + flags + xflag + ["-c", os.path.join(CSRC, src), "-o", obj])
        if verbose and out:
            print(out)
        objs.append(obj)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-o", HIP_LIB] + objs
         + ["-Wl,-rpath,/opt/rocm/lib"])
    return HIP_LIB


@_locked
def build_hip_variant(name, extra_flags):
    """build_variants/libwgahip_<name>.so: the library with extra compiler flags, next to the product build (A/B
    measurements in ONE process on the same buffers: scripts/gpu_k2_same_buffers.py).  Never loaded by the product."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    vdir = os.path.join(ROOT, "build_variants")
    os.makedirs(vdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall"] + list(extra_flags)
    objs = []
    for src, xflag in (("wga_capi.cpp", ["-x", "hip"]), ("wga_pack.cpp", [])):
        obj = os.path.join(vdir, "%s_%s" % (name, src.replace(".cpp", ".o")))
        _run([hipcc] + flags + xflag + ["-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    lib = os.path.join(vdir, "libwgahip_%s.so" % name)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
    for o in objs:
        os.remove(o)
    return lib


@_locked
def build_emu(force=False):
    srcs = _sources() + [os.path.join(ROOT, "tests", "emu", f) for f in ("simt_emu.h", "wga_intrin_emu.h")]
    if not force and not _newer(EMU_LIB, srcs):
        return EMU_LIB
    # -DWGA_MAF_FOLD_STEPS: fold the MAF walks' 16-bit lane counters every 3 steps, so that small test rows reach that path
    _run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-DWGA_EMU", "-DWGA_MAF_FOLD_STEPS=3u", "-Wall",
          "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "tests", "emu")]
         + [os.path.join(CSRC, "wga_capi.cpp"), os.path.join(CSRC, "wga_pack.cpp"), "-o", EMU_LIB])
    return EMU_LIB


CLI_BIN = os.path.join(ROOT, "wgatools_amd", "bin", "wgatools")


@_locked
def build_cli(force=False):
    """the `wgatools` drop-in command line (C++ host layer over the C-ABI)"""
    hdir = os.path.join(ROOT, "wgatools_amd", "host")
    srcs = [os.path.join(hdir, f) for f in ("wgatools_main.cpp", "wga_host.cpp", "wga_host.hpp")]
    parts = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith(".inc")]   # the commands, included by wgatools_main.cpp
    lib = build_hip()
    if not force and not _newer(CLI_BIN, srcs + parts + [lib]):
        return CLI_BIN
    os.makedirs(os.path.dirname(CLI_BIN), exist_ok=True)
    _run(["g++", "-O2", "-g", "-std=c++17", "-Wall", srcs[0], srcs[1], "-o", CLI_BIN,
          "-L" + os.path.dirname(lib), "-lwgahip", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/..",
          "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return CLI_BIN


@_locked
def build_cli_emu(force=False):
    """tests/emu/wgatools_emu — the CLI host code linked against the emulator build of the kernels.
    Test infrastructure: lets the CPU suite exercise the host logic end to end without a GPU."""
    lib = build_emu(force)
    hdir = os.path.join(ROOT, "wgatools_amd", "host")
    srcs = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith((".cpp", ".hpp", ".inc"))] + [lib]
    if not force and not _newer(CLI_EMU_BIN, srcs):
        return CLI_EMU_BIN
    _run(["g++", "-O1", "-g", "-std=c++17", "-Wall", os.path.join(ROOT, "wgatools_amd", "host", "wgatools_main.cpp"),
          os.path.join(ROOT, "wgatools_amd", "host", "wga_host.cpp"), "-o", CLI_EMU_BIN, "-L" + os.path.dirname(lib),
          "-lwgaemu", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    return CLI_EMU_BIN


@_locked
def build_oracle(force=False):
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("oracle.c", "cpu_bench.c", "oracle.h")]
    if not force and not _newer(ORACLE_LIB, srcs):
        return ORACLE_LIB
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return ORACLE_LIB


if __name__ == "__main__":
    print(build_oracle(force=True))
    print(build_emu(force=True))
    print(build_hip(force=True, verbose=True))
