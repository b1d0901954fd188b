#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE — never combined with trace domains other than --kernel-trace), and writes
# the summaries under gpurun_out/prof_<tag>/.   usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu --no-hbm-point "$@" > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-hbm-point "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-hbm-point "$@" > /dev/null 2> $OUT/write.err
cd $REPO
T=$(find $OUT/trace -name '*.db' | head -1); Fh=$(find $OUT/fetch -name '*.db' | head -1); W=$(find $OUT/write -name '*.db' | head -1)
python tools/rocpd_summary.py $OUT/summary.md "kernel-trace --stats, bench.py $*=$T" "pmc FETCH_SIZE (KiB), bench.py $*=$Fh" "pmc WRITE_SIZE (KiB), bench.py $*=$W"
# keep only the summaries (the databases are large)
rm -rf $OUT/trace $OUT/fetch $OUT/write
tail -3 $OUT/bench_under_rocprof.json
