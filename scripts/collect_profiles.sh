#!/bin/bash
# Run on the GPU box (gpurun): the measurements DESIGN.md sections 4 and 6 and bench.py's `roofline.traffic` cite.
#   scripts/collect_profiles.sh rNN
#   1. bench.py (default command) under rocprofv3 --kernel-trace --stats      -> per-kernel average durations
#   2. the same workload under --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing domains)
#   3. SQ counters of the NTT passes and of the SHA-256 commit phase
#   4. plain bench.py JSON line, scripts/bench_configs.py, scripts/bench_commit.py, the C2 sweep
# Raw outputs land in gpurun_out/profiles_raw/ and are REDUCED HERE, on the box that produced them, into
# gpurun_out/profiles_raw/summary/rNN_* (scripts/summarise_profiles.py); those files are what goes to profiles/.
# Round 3 lost its SQ counter file to the size filter below and a summary was then made from the previous round's copy that
# gpurun's merge had left in place: hence the stamp (every input must be newer than it), the reduction before any delete,
# and a non-zero exit when a counter file did not come back.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_raw
rm -rf $OUT; mkdir -p $OUT
date +%s.%N > $OUT/RUN_STAMP
cd $R
FAIL=0
python bench.py > $OUT/bench.json 2> $OUT/bench.err         # the one line (< 4 KiB) ...
cp $R/bench_detail.json $OUT/bench_detail.json                # ... and the full record it summarises (round 6)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o ntt --output-format csv -- python bench.py --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
PM="python bench.py --steps 2 --warmup 1 --cols 2 --no-cpu-baseline --no-extras --settle 0.3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ntt --output-format csv -- $PM > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o ntt --output-format csv -- $PM > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o ntt --output-format csv -- $PM > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sha -o sha --output-format csv -- python scripts/bench_commit.py 23 32 1 > $OUT/pmc_sha.log 2>&1
for d in stats pmc_fetch pmc_write pmc_sq pmc_sha; do
    if ! find $OUT/$d -name "*.csv" 2>/dev/null | grep -q .; then echo "collect_profiles: MISSING rocprofv3 output under $d (see $OUT/$d.log)"; FAIL=1; fi
done
python scripts/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
python scripts/c2_sweep.py --all > $OUT/c2_sweep.json 2>/dev/null; python scripts/c2_sweep.py --all --inverse >> $OUT/c2_sweep.json 2>/dev/null
python scripts/c2_sweep.py --small > $OUT/c2_sweep_small.json 2>/dev/null
NCOLS=32 ./scripts/sq_probe.sh lde python scripts/lde_probe.py > /dev/null 2>&1; cp $R/gpurun_out/sq_lde.txt $OUT/lde_sq_counters.txt || FAIL=1
python bench.py --mode lde-commit --steps 3 --warmup 1 > $OUT/bench_lde_commit_n1.json 2>/dev/null
{ FQ3=1 LOGN=20 LOGB=3 NCOLS=4 python scripts/lde_probe.py; MS_LDE2_FQ3=0 FQ3=1 LOGN=20 LOGB=3 NCOLS=4 python scripts/lde_probe.py; LOGN=22 LOGB=2 NCOLS=8 python scripts/lde_probe.py; } > $OUT/summary_lde_probe.txt 2>/dev/null
python scripts/bench_commit.py 23 32 3 > $OUT/bench_commit.json 2>&1
# reduce on this box, from this run's files only
python scripts/summarise_profiles.py $TAG --raw $OUT --out $OUT/summary || FAIL=1
find $OUT -name "*.csv" -size +4M -not -path "*/summary/*" -delete      # per-dispatch traces can be large; the summaries above are what travels
ls -la $OUT/summary 2>/dev/null | head -40
if [ $FAIL -ne 0 ]; then echo "collect_profiles: INCOMPLETE (see messages above)"; exit 1; fi
echo "collect_profiles: complete, stamp $(cat $OUT/RUN_STAMP)"
