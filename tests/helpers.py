"""Small host-side helpers shared by the tests (fixture readers; no reference code)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_maf_blocks(path):
    """MAFRecords::next semantics (maf.rs:371-421): first line is the header; a block is a
    maximal run of lines starting with 's'."""
    blocks, cur = [], []
    with open(path, "rb") as f:
        f.readline()
        for line in f:
            line = line.rstrip(b"\n")
            if line.startswith(b"s"):
                p = line.split()
                assert len(p) == 7
                cur.append(dict(name=p[1].decode(), start=int(p[2]), align=int(p[3]),
                                strand=p[4].decode(), size=int(p[5]), seq=p[6]))
            elif cur:
                blocks.append(cur)
                cur = []
    if cur:
        blocks.append(cur)
    return blocks


def read_paf(path):
    recs = []
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith("#"):
                continue
            p = line.rstrip("\n").split("\t")
            cg = next((t for t in p[12:] if t.startswith("cg:Z:")), None)
            recs.append(dict(qname=p[0], qlen=int(p[1]), qstart=int(p[2]), qend=int(p[3]),
                             strand=p[4], tname=p[5], tlen=int(p[6]), tstart=int(p[7]),
                             tend=int(p[8]), matches=int(p[9]), block=int(p[10]), mapq=int(p[11]),
                             cg=cg))
    return recs


def pack_records(eng, cigars):
    """list of CIGAR texts (no tag) -> (ops, op_off, errs) through the product's host packer"""
    ops, off, errs = [], [0], []
    for c in cigars:
        o, e, _ = eng.pack_cigar(c)
        ops.append(o)
        off.append(off[-1] + len(o))
        errs.append(e)
    ops = np.concatenate(ops) if ops else np.zeros(0, np.uint32)
    return ops.astype(np.uint32), np.array(off, dtype=np.uint64), errs
