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
from wgatools_amd import build, engine, _lib, synth, shard, multigpu
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
multigpu.allreduce_totals(tot, dist)
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


WORKER_ORDERED = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.distributed as dist
from wgatools_amd import build, engine, _lib, synth, shard, multigpu
import parity_cases as pc, oracle_py as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
N = 36
b = synth.make_paf_batch(321, N, 90, 30000)
rng = np.random.default_rng(4)
names = ["g%02d#1#chr%d" % (int(z) % 5, int(z) % 3) for z in rng.zipf(1.6, N)]      # skewed: a few hot targets
pre = (rng.integers(1, 40, N), rng.integers(1, 40, N), rng.integers(0, 5, N))      # MAF line text around the rows
owner = multigpu.owners(names, world)
mine = multigpu.my_records(owner, rank)
sb = shard.select_batch(b, mine)
r = pc.run_paf2maf(eng, sb, pre=tuple(p[mine] for p in pre), fill=0x23)           # this rank's K1 + layout + K2
sizes = np.diff(r["rec_off"].astype(np.int64))
goff, total = multigpu.ordered_offsets(N, mine, sizes, dist)
path = OUT
if rank == 0:
    multigpu.write_ordered(path, r["out"], r["rec_off"][:-1], sizes, goff, total=total, create=True)
dist.barrier()
if rank != 0:
    multigpu.write_ordered(path, r["out"], r["rec_off"][:-1], sizes, goff)
dist.barrier()
w, imb = multigpu.imbalance(int(sb["op_off"][-1]), dist)
if rank == 0:
    want = bytearray()
    for i in range(N):                                     # input order, rows from the ORACLE
        et, eq = pc.oracle_rows(b, i)
        want += b"#" * int(pre[0][i]) + et + b"#" * int(pre[1][i]) + eq + b"#" * int(pre[2][i])
    got = open(path, "rb").read()
    assert len(got) == total == len(want), (len(got), total, len(want))
    assert got == bytes(want)
    assert abs(sum(w) - int(b["op_off"][-1])) < 0.5 and imb >= 1.0
    print("ORDERED_OK imbalance %.2f" % imb)
# ---- a hot target whose records are spread over the ranks: coverage reduce ----
n2 = 30
b2 = pc.sprinkle_ops(np.random.default_rng(9), synth.make_paf_batch(77, n2, 50, 1000))
tlen = 3000
tstart = (np.random.default_rng(10).random(n2) * tlen * 1.02).astype(np.uint64)
part = np.arange(n2)[rank::world]                            # record-sharded, not target-sharded
sb2 = shard.select_batch(b2, part)
batch = eng.make_batch(sb2["ops"], sb2["op_off"], sb2["strand_neg"])
cov = eng.empty(tlen + 8, np.int32).fill(0)
d_off, d_len = eng.upload(np.zeros(1, np.uint64)), eng.upload(np.array([tlen], dtype=np.uint64))
eng.pafcov_accumulate(batch, eng.upload(np.zeros(len(part), np.uint32)), eng.upload(tstart[part]), d_off, d_len, cov, tlen)
eng.pafcov_finalize(1, d_off, d_len, cov)
lo, hi, sl = multigpu.hot_target_coverage(torch.from_numpy(cov.numpy()[:tlen].copy()), dist)
pieces = [None] * world
dist.all_gather_object(pieces, (lo, hi, sl.numpy()))
if rank == 0:
    exp = np.zeros(tlen, dtype=np.uint64)
    for i in range(n2):
        orc.update_cov_vec(exp, pc.text_any(pc.rec_ops(b2, i)), int(tstart[i]))
    got = np.zeros(tlen, dtype=np.int64)
    covered = 0
    for lo_, hi_, v in pieces:
        got[lo_:hi_] = v
        covered += hi_ - lo_
    assert covered == tlen and (got == exp.astype(np.int64)).all()
    print("COVREDUCE_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_ordered_output_and_coverage_reduce(tmp_path):
    """paf2maf over two ranks (target-hash sharding, skewed targets) lands in ONE file in input order, byte-identical to
    the oracle's rows; the coverage of a record-sharded hot target is reduced over the ranks and equals the oracle's"""
    from wgatools_amd import build
    build.build_emu()
    script = tmp_path / "worker2.py"
    script.write_text("ROOT = %r\nOUT = %r\n" % (ROOT, str(tmp_path / "ordered.bin")) + WORKER_ORDERED)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29563", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "ORDERED_OK" in r.stdout and "COVREDUCE_OK" in r.stdout, r.stdout[-3000:]


def test_ordered_offsets_single_process():
    from wgatools_amd import multigpu
    import numpy as np
    owner = multigpu.owners(["a", "b", "a", "c", "b", "a"], 3)
    assert sorted(np.concatenate([multigpu.my_records(owner, r) for r in range(3)]).tolist()) == list(range(6))
    off, total = multigpu.ordered_offsets(4, np.array([0, 1, 2, 3]), np.array([5, 0, 7, 1]))
    assert off.tolist() == [0, 5, 5, 12] and total == 13


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
