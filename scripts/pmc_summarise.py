"""Turn the rocprofv3 counter CSVs of scripts/gpu_pmc.sh into the tracked summaries under profiles/.
usage: python scripts/pmc_summarise.py gpurun_out/pmc06 profiles/r01 <launches> <records> <mean_ops> <ops> [kernel]"""
import collections
import csv
import glob
import json
import os
import sys

src, dst, launches, records, mean_ops, ops = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
rows = []
agg_all = {}
for d in ("sq1", "sq2", "sq3", "fetch", "write", "tcc"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not (k.startswith("k_") or k.startswith("void k_")):
                continue
            agg[k.split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in sorted(agg.items()):
            for c, x in sorted(v.items()):
                rows.append((d, k, c, x))
                agg_all[(k, c)] = x
with open(dst + "_pmc.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <counters>, one pass per line group (scripts/gpu_pmc.sh), summed over the\n"
            "# %d launches of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --check 0` (our kernels only)\n" % launches)
    f.write("%-6s %-28s %-24s %s\n" % ("pass", "kernel", "counter", "sum over launches"))
    for d, k, c, x in rows:
        f.write("%-6s %-28s %-24s %.6g\n" % (d, k, c, x))
k = sys.argv[7] if len(sys.argv) > 7 else "k_paf2maf_expand_s"
fetch_kb, write_kb = agg_all[(k, "FETCH_SIZE")], agg_all[(k, "WRITE_SIZE")]
rd = 2.0 * fetch_kb * 1024 / launches   # gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B
wr = write_kb * 1024 / launches
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha: the traffic figure is only valid for the kernel source it was measured on)
json.dump({
    "kernel": k, "launches": launches, "kernel_source_sha": bench.kernel_source_sha(),
    "workload": {"records": records, "mean_ops": mean_ops, "ops": ops},
    "FETCH_SIZE_KB_sum": fetch_kb, "WRITE_SIZE_KB_sum": write_kb,
    "TCC_EA0_RDREQ_sum": agg_all.get((k, "TCC_EA0_RDREQ_sum")), "TCC_EA0_WRREQ_sum": agg_all.get((k, "TCC_EA0_WRREQ_sum")),
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half of a wide streaming "
                  "read; 2 x FETCH is compared with the 4n + target + query bytes the kernel must read in DESIGN.md); "
                  "WRITE_SIZE taken as reported (= TCC_EA0_WRREQ x 64 B)",
}, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(open(dst + "_pmc_traffic.json").read())
