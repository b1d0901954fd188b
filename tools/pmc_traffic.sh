#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic per launch of every kernel bench.py reports a roofline for.  Two separate PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X guide, counter table) over tools/traffic_workloads.py, with --kernel-trace only (no other trace domain).
# usage: tools/pmc_traffic.sh <tag> [traffic_workloads args...]   ->  gpurun_out/pmc_<tag>/traffic.json + .md
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o $c -- python $REPO/tools/traffic_workloads.py "$@" > $OUT/$c.out 2> $OUT/$c.err
done
cd $REPO
python tools/pmc_traffic_summary.py $OUT "$*" > $OUT/traffic.md
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
cat $OUT/traffic.md
