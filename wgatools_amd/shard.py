"""Multi-GPU partitioning of the hot path: records are independent, so a job is
sharded by hash(target_name) % G — every target's coverage array and pseudo-MAF rows then live on
exactly one GPU and no data-path collective is needed.  The only exchange is the all-reduce of the
global stat totals (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
import numpy as np

FNV_OFFSET, FNV_PRIME = 0xCBF29CE484222325, 0x100000001B3


def fnv1a64(name):
    """deterministic across processes and runs (Python's hash() is salted)"""
    h = FNV_OFFSET
    for b in name.encode() if isinstance(name, str) else bytes(name):
        h = ((h ^ b) * FNV_PRIME) & 0xFFFFFFFFFFFFFFFF
    return h


def shard_of(target_name, world):
    return fnv1a64(target_name) % world


def shard_records(target_names, world, rank):
    """indices (input order preserved) of the records rank `rank` owns"""
    return [i for i, t in enumerate(target_names) if shard_of(t, world) == rank]


def select_batch(b, idx):
    """sub-batch of a numpy paf batch (wgatools_amd.synth.make_paf_batch layout)"""
    idx = np.asarray(idx, dtype=np.int64)
    lens = (b["op_off"][1:] - b["op_off"][:-1]).astype(np.int64)[idx]
    off = np.zeros(len(idx) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    ops = np.concatenate([b["ops"][int(b["op_off"][i]):int(b["op_off"][i + 1])] for i in idx]) \
        if len(idx) else np.zeros(0, np.uint32)
    out = dict(b)
    out.update(ops=ops.astype(np.uint32), op_off=off)
    for k in ("strand_neg", "t_src_off", "t_src_len", "q_src_off", "q_src_len"):
        out[k] = b[k][idx]
    return out
