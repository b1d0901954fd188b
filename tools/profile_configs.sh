#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of tools/bench_configs.py for the given configs, summary under gpurun_out/.
# usage: tools/profile_configs.sh <tag> <configs> [bench_configs args...]
set -u
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/tools/bench_configs.py --configs $CFG "$@" > $OUT/bench_configs_under_rocprof.jsonl 2> $OUT/trace.err
cd $REPO
T=$(find $OUT/trace -name '*.db' | head -1)
python tools/rocpd_summary.py $OUT/summary.md "kernel-trace --stats, tools/bench_configs.py --configs $CFG $*=$T"
rm -rf $OUT/trace
head -30 $OUT/summary.md
