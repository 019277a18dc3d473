#!/bin/bash
# file-to-file wall times of the command line + the phase timer of paf2maf / call (profiles/r02_cli_e2e.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
[ "$1" = "phases" ] || python scripts/gpu_cli_e2e.py 20000 /tmp/wga_e2e 100000
[ -f /tmp/wga_e2e/in.paf ] || python scripts/gpu_cli_e2e.py 20000 /tmp/wga_e2e 100000 > /dev/null
TIMEFORMAT="   wall %R s"
for rep in 1 2; do
  echo "paf2maf:";  time WGA_TIMING=1 wgatools_amd/bin/wgatools paf2maf /tmp/wga_e2e/in.paf -g /tmp/wga_e2e/t.fa -q /tmp/wga_e2e/q.fa -o /tmp/wga_e2e/o.maf -r
  echo "call -s:";  time WGA_TIMING=1 wgatools_amd/bin/wgatools call -s -l 50 /tmp/wga_e2e/in.maf -o /tmp/wga_e2e/o.vcf -r
  echo "stat maf:"; time WGA_TIMING=1 wgatools_amd/bin/wgatools stat /tmp/wga_e2e/in.maf -o /tmp/wga_e2e/o.tsv -r
done
echo "paf2chain:"; time WGA_TIMING=1 wgatools_amd/bin/wgatools paf2chain /tmp/wga_e2e/in.paf -o /tmp/wga_e2e/o.chain -r
echo "pafcov:"; time WGA_TIMING=1 wgatools_amd/bin/wgatools pafcov /tmp/wga_e2e/in.paf -o /tmp/wga_e2e/o.bed -r
echo "stat paf:"; time WGA_TIMING=1 wgatools_amd/bin/wgatools stat -f paf /tmp/wga_e2e/in.paf -o /tmp/wga_e2e/o.tsv -r
