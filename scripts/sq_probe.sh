#!/bin/bash
# SQ counters of a command's kernels: two PMC passes (8 SQ slots each), reduced per kernel name by scripts/sq_reduce.py.
#   scripts/sq_probe.sh <tag> <command...>          -> gpurun_out/sq_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/sq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/b -o p --output-format csv -- "$@" > $OUT/b.log 2>&1
python scripts/sq_reduce.py $OUT > $R/gpurun_out/sq_$TAG.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $R/gpurun_out/sq_$TAG.txt
