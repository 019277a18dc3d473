"""The multi-GPU product path (north_star: "records shard naturally by hash(target_name) across the
8 GPUs of one node with an RCCL reduce over xGMI only for the global stat / pafcov totals").

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  Every rank sees the
same global record list (names, sizes) and works on the records it owns; the kernels are the single-GPU ones behind the
C-ABI.  Row bytes never cross GPUs.  The exchanges are:

  * paf2maf / call — output must come out IN INPUT ORDER (the reference's loop is serial, converter.rs:196-263).  Each
    record's output size is known after K1 + the layout scan, before a single row byte is written; one all-reduce of the
    per-record byte counts (8 B per record) gives every rank the global offsets, and each rank writes its records at
    their final place (`ordered_offsets`, `write_ordered`): an ordered gather without a gather.
  * stat — all-reduce of the 11 global counters (88 B), `allreduce_totals`.
  * pafcov — none with target-hash sharding (a target's coverage lives on one GPU).  A HOT target whose records are
    spread over the ranks instead (`hot_target_coverage`) costs one reduce-scatter of its int32 array (the element-wise
    merge of per-thread arrays in pafcov.rs:29-53, moved to xGMI): every rank ends up owning one slice of the summed
    coverage, ready to be formatted where it lies.
"""
import os

import numpy as np

from . import shard

__all__ = ["owners", "owners_lpt", "human_like_targets", "my_records", "ordered_offsets", "write_ordered", "allreduce_totals", "hot_target_coverage",
           "imbalance"]


def owners(target_names, world):
    """owner rank of every record: fnv1a64(target_name) % world (hashes computed once per distinct name)"""
    uniq = {}
    out = np.empty(len(target_names), dtype=np.int32)
    for i, t in enumerate(target_names):
        o = uniq.get(t)
        if o is None:
            o = uniq[t] = shard.shard_of(t, world)
        out[i] = o
    return out


def owners_lpt(target_names, weights, world):
    """Size-aware assignment of TARGETS to ranks (every record of a target stays on one rank, as the hash rule keeps it):
    targets sorted by their total weight (ops), heaviest first, each to the rank that is lightest so far (LPT).  The
    hash rule needs no table and no pass over the input; this one needs the per-target totals first (one pass over the PAF's
    first columns) and gives an imbalance near 1 when a few targets are large.  Returns the owner rank of every record."""
    import heapq
    names = list(target_names)
    w = {}
    for t, x in zip(names, weights):
        w[t] = w.get(t, 0) + int(x)
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    owner_of = {}
    for t in sorted(w, key=lambda t: (-w[t], t)):
        load, r = heapq.heappop(heap)
        owner_of[t] = r
        heapq.heappush(heap, (load + w[t], r))
    return np.fromiter((owner_of[t] for t in names), dtype=np.int32, count=len(names))


HUMAN_CONTIG_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]
HUMAN_CONTIG_NAMES = ["chr%d" % k for k in range(1, 23)] + ["chrX", "chrY"]


def human_like_targets(genomes=64):
    """target names and sizes (Mb) of a pangenome of `genomes` human-like assemblies in PanSN naming (sample#hap#contig):
    chr1 .. chrY at their GRCh38 sizes — what a 64-way all-to-all PAF holds, instead of equal-sized gNN#1#chr1 names whose
    hashes happen to split evenly"""
    names, sizes = [], []
    for g in range(genomes):
        sample = "HG%05d" % (2000 + 7 * g)
        for c, mb in zip(HUMAN_CONTIG_NAMES, HUMAN_CONTIG_MB):
            names.append("%s#%d#%s" % (sample, 1 + (g & 1), c))
            sizes.append(mb)
    return names, np.asarray(sizes, dtype=np.float64)


def my_records(owner, rank):
    """input-order indices of the records `rank` owns"""
    return np.flatnonzero(np.asarray(owner) == rank)


def _dist_ok(dist):
    return dist is not None and dist.is_initialized() and dist.get_world_size() > 1


def ordered_offsets(n_global, mine, my_sizes, dist=None, device=None):
    """Global placement of every record's output in input order.

    mine: input-order indices of this rank's records; my_sizes: their output byte counts (same order; a torch tensor on
    `device` or a numpy array).  One all-reduce (sum) of an n_global-long int64 vector in which every rank fills only its
    own positions.  Returns (global offsets of MY records as numpy int64, total bytes)."""
    import torch
    sizes = torch.zeros(n_global, dtype=torch.int64, device=device)
    idx = torch.as_tensor(np.asarray(mine, dtype=np.int64), device=device)
    ms = my_sizes if hasattr(my_sizes, "to") else torch.as_tensor(np.asarray(my_sizes, dtype=np.int64))
    sizes[idx] = ms.to(device=device, dtype=torch.int64)
    if _dist_ok(dist):
        dist.all_reduce(sizes)
    off = torch.cumsum(sizes, 0) - sizes
    total = int(sizes.sum())
    return off[idx].cpu().numpy(), total


def write_ordered(path, local_bytes, local_off, local_len, global_off, total=None, create=False):
    """Every rank writes the records it produced into ONE output file at their final (input-order) offsets.

    local_bytes: this rank's output buffer on the host (numpy uint8); record k occupies
    local_bytes[local_off[k] : local_off[k] + local_len[k]] and belongs at global_off[k].  Neighbouring records that are
    also neighbours in the file go out in one pwrite.  Rank 0 calls with create=True (and `total`) first; the others
    open the existing file (put a barrier between)."""
    flags = os.O_WRONLY | (os.O_CREAT | os.O_TRUNC if create else 0)
    fd = os.open(path, flags, 0o644)
    try:
        if create and total is not None:
            os.ftruncate(fd, total)
        k, n = 0, len(global_off)
        mv = memoryview(np.ascontiguousarray(local_bytes))
        while k < n:
            j = k
            while (j + 1 < n and int(global_off[j]) + int(local_len[j]) == int(global_off[j + 1])
                   and int(local_off[j]) + int(local_len[j]) == int(local_off[j + 1])):
                j += 1
            a = int(local_off[k])
            z = int(local_off[j]) + int(local_len[j])
            os.pwrite(fd, mv[a:z], int(global_off[k]))
            k = j + 1
    finally:
        os.close(fd)


def allreduce_totals(totals, dist=None):
    """sum the 11 stat counters over ranks (in place on a torch tensor): RCCL over xGMI, 88 bytes"""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(totals)
    return totals


def hot_target_coverage(cov, dist=None):
    """Sum the per-rank coverage arrays of ONE target whose records were spread over the ranks, and leave every rank
    with its slice of the result.

    cov: this rank's int32 coverage of the whole target (torch tensor, after its own accumulate + finalize over its
    share of the records; coverage is additive over disjoint record sets).  Returns (lo, hi, slice): the rank owns
    positions [lo, hi) of the summed coverage.  NCCL/RCCL: one reduce_scatter_tensor over a length padded to a multiple
    of the world size; gloo (CPU tests) has no reduce-scatter, so an all-reduce followed by the same slicing."""
    import torch
    n = int(cov.numel())
    if not _dist_ok(dist):
        return 0, n, cov
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    if dist.get_backend() == "nccl":
        padded = cov if per * world == n else torch.cat([cov, torch.zeros(per * world - n, dtype=cov.dtype, device=cov.device)])
        out = torch.empty(per, dtype=cov.dtype, device=cov.device)
        dist.reduce_scatter_tensor(out, padded)
        return lo, hi, out[: hi - lo]
    full = cov.clone()
    dist.all_reduce(full)
    return lo, hi, full[lo:hi]


def imbalance(work_mine, dist=None, device=None):
    """(per-rank work list, max / mean) for a scalar amount of work (ops, bytes) per rank"""
    import torch
    if not _dist_ok(dist):
        return [float(work_mine)], 1.0
    world, rank = dist.get_world_size(), dist.get_rank()
    v = torch.zeros(world, dtype=torch.float64, device=device)
    v[rank] = float(work_mine)
    dist.all_reduce(v)
    w = v.cpu().tolist()
    mean = sum(w) / len(w)
    return w, (max(w) / mean if mean > 0 else 1.0)
