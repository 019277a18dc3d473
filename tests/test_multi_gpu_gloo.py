"""The N > 1 path on CPU: 2 processes over gloo, each owning the records hash(target) assigns to
it, running the kernel source on the SIMT emulator, all-reducing the stat totals.  Checks that the
shards partition the input, that the reduced totals equal the oracle's totals over the whole input,
and that every shard's paf2maf rows match the oracle (no data-path collective is involved)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.distributed as dist
from wgatools_amd import build, engine, _lib, synth, shard
import parity_cases as pc, oracle_py as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
b = synth.make_paf_batch(123, 40, 120, 40000)
names = ["chr%d" % (i % 7) for i in range(40)]            # target names of the records
mine = shard.shard_records(names, world, rank)
sb = shard.select_batch(b, mine)
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
pc.check_paf2maf(eng, sb)                                   # this shard's rows == oracle rows
batch = eng.make_batch(sb["ops"], sb["op_off"], sb["strand_neg"])
counts, _, _ = eng.cigar_stat(batch)
c = counts.numpy()
tot = torch.tensor([int(c[k].sum()) for k in engine.COUNTS_DTYPE.names], dtype=torch.int64)
n_mine = torch.tensor([len(mine)], dtype=torch.int64)
shard.allreduce_totals(tot, dist)
dist.all_reduce(n_mine)
if rank == 0:
    exp = np.zeros(11, dtype=np.int64)
    for i in range(40):
        exp += np.array(orc.parse_paf_to_cigar(pc.rec_text(b, i), b["strand_neg"][i]), dtype=np.int64)
    assert int(n_mine) == 40, int(n_mine)
    assert (tot.numpy() == exp).all(), (tot.numpy(), exp)
    print("GLOO_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharding_and_totals(tmp_path):
    from wgatools_amd import build
    build.build_emu()
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\n" % ROOT + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29561", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "GLOO_OK" in r.stdout, r.stdout[-3000:]


def test_shard_function_is_a_partition():
    from wgatools_amd import shard
    names = ["g%02d#1#chr%d" % (i % 13, i % 5) for i in range(500)]
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard.shard_records(names, world, r))
        assert seen == list(range(500))
        # all records of one target land on one rank
        for t in set(names):
            assert len({shard.shard_of(t, world)}) == 1
    assert shard.fnv1a64("abc") == 0xE71FA2190541574B
