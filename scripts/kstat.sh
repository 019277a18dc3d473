#!/bin/bash
# register / LDS / scratch / occupancy of the kernels matching $1 (default: paf2maf) in a gfx950 build of wga_capi.cpp
# usage: scripts/kstat.sh [pattern] [extra compiler flags...]
PAT=${1:-paf2maf}; shift
cd $(dirname $0)/../wgatools_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWGA_STAGE2 "$@" -x hip -c wga_capi.cpp -o /tmp/wga_capi.o -save-temps=obj 2>/dev/null || { echo "compile failed"; exit 1; }
S=/tmp/wga_capi-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v pat="$PAT" '
/\.amdhsa_kernel / {name=$2}
/\.amdhsa_next_free_vgpr/ {v[name]=$2}
/\.amdhsa_next_free_sgpr/ {s[name]=$2}
/\.amdhsa_group_segment_fixed_size/ {l[name]=$2}
/\.amdhsa_private_segment_fixed_size/ {p[name]=$2}
END {for (k in v) if (k ~ pat) printf "%-56s vgpr %3d sgpr %3d lds %6d scratch %d\n", k, v[k], s[k], l[k], p[k]}' $S
awk -v pat="$PAT" '/^_Z.*:/ {n=$1} /; Occupancy:/ {if (n ~ pat) print n, $0}' $S
