"""The multi-rank driver (tests/dist_cli.py: torch.distributed over gloo, the emulator build of the kernels, two and three ranks)
on random inputs: its files against the oracle's expectation and the single-process command line:
python scripts/emu_campaign_dist_cli.py <first seed> <seconds> <first port>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import os, pathlib, shutil, subprocess, sys, tempfile, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import dist_cli_cases as dc
from wgatools_amd import build, synth
lib, cli = build.EMU_LIB, build.CLI_EMU_BIN
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); port0 = int(sys.argv[3]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="wga_camp_"))
    try:
        n = int(rng.integers(1, 45)); mean = int(rng.choice([3, 40, 250]))
        b = synth.make_paf_batch(s, n, mean, 40000, use_m=bool(rng.integers(0, 2)))
        mapq = rng.integers(0, 61, n)
        t_fa, q_fa, paf = dc.write_case(tmp, b, mapq)
        want = dc.expected_maf(b, mapq, t_fa, q_fa, n)
        bed = dc.expected_bed(b)
        w = int(rng.choice([2, 3]))
        chunk = str(int(rng.choice([3000, 20000, 1 << 20])))
        port = port0 + (k % 200) * 4
        outp = str(tmp / "out.maf")
        dc.launch(w, lib, port, "paf2maf", paf, "-g", t_fa, "-q", q_fa, "-o", outp, "--chunk-bytes", chunk)
        assert open(outp, "rb").read() == want, "paf2maf"
        for j, spread in enumerate((False, True)):
            outp = str(tmp / ("cov%d.bed" % spread))
            dc.launch(w, lib, port + 1 + j, "pafcov", paf, "-o", outp, "--chunk-bytes", chunk, *(["--spread"] if spread else []))
            assert open(outp, "rb").read() == bed, ("pafcov", spread)
    except Exception:
        fails += 1
        print("FAIL seed", s); traceback.print_exc(); sys.stdout.flush()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
