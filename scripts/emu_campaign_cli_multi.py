"""`wgatools --gpus N` of the emulator build (N worker threads, N contexts) on random inputs: its files against `--gpus 1`:
python scripts/emu_campaign_cli_multi.py <first seed> <seconds>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import os, pathlib, shutil, subprocess, sys, tempfile, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import cli_cases as cc
import parity_cases as pc
from wgatools_amd import build, synth
cli = build.CLI_EMU_BIN
ENV = dict(os.environ, WGA_EMU_DEVICES="5")
def run(*args):
    r = subprocess.run([cli] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV)
    return r.returncode, r.stdout, r.stderr.decode()
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="wga_camp_"))
    try:
        n = int(rng.integers(1, 50)); mean = int(rng.choice([2, 15, 80, 300]))
        b = synth.make_paf_batch(s, n, mean, 60000, use_m=bool(rng.integers(0, 2)))
        mapq = rng.integers(0, 61, n)
        nt = int(rng.integers(1, 7))
        # several targets, so that the records spread over the devices: the same pool under several names
        t_fa, q_fa, paf = cc._write_paf2maf_case(tmp, b, mapq)
        seq = open(t_fa, "rb").read().split(b"\n", 1)[1]
        with open(t_fa, "wb") as f:
            for j in range(nt):
                f.write(b">tchr%d\n" % j + seq)
        lines = open(paf).read().splitlines()
        with open(paf, "w") as f:
            for i, ln in enumerate(lines):
                f.write((ln.replace("\ttchr\t", "\ttchr%d\t" % int(rng.integers(0, nt))) if not ln.startswith("#") else ln) + "\n")
        ref = {}
        for g in (1, int(rng.choice([2, 3, 5]))):
            outs = {}
            for name, argv in (("paf2maf", ["paf2maf", paf, "-g", t_fa, "-q", q_fa]), ("pafcov", ["pafcov", paf]), ("spread", ["--spread", "pafcov", paf])):
                outp = str(tmp / ("%s.%d" % (name, g)))
                rc, _, err = run("--gpus", str(g), *argv, "-o", outp)
                assert rc == 0, (name, g, err[-300:])
                outs[name] = open(outp, "rb").read()
            rc, out, err = run("--gpus", str(g), "stat", "-f", "paf", paf)
            assert rc == 0, ("stat", g, err[-300:])
            outs["stat"] = out
            if g == 1:
                ref = outs
                assert ref["pafcov"] == ref["spread"] and ref["paf2maf"].count(b"a score=") == n
            else:
                for name in ref:
                    assert outs[name] == ref[name], (name, g)
    except Exception:
        fails += 1
        print("FAIL seed", s); traceback.print_exc(); sys.stdout.flush()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if k % 10 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
