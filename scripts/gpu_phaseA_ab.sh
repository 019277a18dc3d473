R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for V in "0|-DWGA_V1_ABLATE=1" "1|-DWGA_P_ABLATE=1"; do
  var=${V%%|*}; F=${V#*|}
  WGA_EXTRA_FLAGS="$F" python -c "from wgatools_amd import build; build.build_hip(force=True)" > /dev/null 2>&1
  (cd /tmp; WGA_EXPAND_VARIANT=$var rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pa$var -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --check 0 > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
for f in glob.glob("/tmp/pa$var/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("k_paf2maf_expand"): agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("variant $var:", {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items()) if sum(v)})
PY
  WGA_EXPAND_VARIANT=$var python bench.py --no-cpu-baseline --no-extras --check 0 --steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $var phase A only: K2 %.3f ms' % d['kernel_ms']['k_paf2maf_expand'])"
done
