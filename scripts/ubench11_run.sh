#!/bin/bash
# Run on the GPU box: scripts/ubench11 timings, then HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) of chosen variants.
#   scripts/ubench11_run.sh [variant ...]      -> gpurun_out/ubench11.txt, gpurun_out/ubench11_traffic.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
if [ -z "${UB11_SKIP_TIMING:-}" ]; then timeout 300 $R/scripts/ubench11 > $OUT/ubench11.txt 2>&1; echo "timing exit $?" >> $OUT/ubench11.txt; fi
VARS=${@:-"ref16_0 ref12_0 A64_t4_ip_0 A64_t2_ip_0 A32_t4_ip_0 A64_t4_slab_0 A64_t4_ip_w B_def_grp_8 B_nt_grp_8 B_def_ord_8 B_def_grp_16 Bteam_t4_0"}
: > $OUT/ubench11_traffic.txt
for v in $VARS; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/ub11_${v}_$ctr; rm -rf $D
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d $D -o t --output-format csv -- $R/scripts/ubench11 $v 2 > $D.log 2>&1
    python3 - $D $v $ctr >> $OUT/ubench11_traffic.txt <<'PY'
import csv, glob, sys
d, v, ctr = sys.argv[1:4]
hits = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not hits:
    print(f"{v} {ctr} MISSING"); sys.exit(0)
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(hits[0])) if r["Counter_Name"] == ctr and any(k in r["Kernel_Name"] for k in ("k_team", "k_rows", "k_tile", "k_rewrite", "k_handoff"))]
if not vals:
    print(f"{v} {ctr} no kernel rows"); sys.exit(0)
mult = 2 * 1024 if ctr == "FETCH_SIZE" else 1024         # gfx950: FETCH_SIZE tallies 64 B per request of a coalesced stream
per_col = mult * sum(vals) / len(vals) / 8
print(f"{v:16s} {ctr:10s} {per_col / 1e6:9.1f} MB per column and launch  ({per_col / (8 << 24):.2f} x the column, {len(vals)} launches)")
PY
  done
done
cat $OUT/ubench11.txt | tail -60
cat $OUT/ubench11_traffic.txt
