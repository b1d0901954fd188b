#!/bin/bash
# PMC passes (no trace domains besides --kernel-trace) over tools/bench_configs.py: LDS and instruction counters of the scan kernels of the given configs.
# usage: tools/pmc_configs.sh <tag> <configs> [bench_configs args...]     -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o g$i -- python $REPO/tools/bench_configs.py --configs $CFG "$@" > /dev/null 2> $OUT/g$i.err
done
cd $REPO
python - > $OUT/summary.txt <<PY
import sqlite3, glob
print("rocprofv3 --pmc (two passes) over tools/bench_configs.py --configs $CFG $*: per-launch averages, kernels above 0.1 ms")
for db in sorted(glob.glob("$OUT/g*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%scan%' or kernel_name like '%rows_kernel%' or kernel_name like '%prefilter%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for n, cn, k, avg in rows:
        if k >= 3: print(f"{n[:90]:90s} {cn:24s} launches={k} avg={avg:.4g}")
PY
rm -rf $OUT/g*/
cat $OUT/summary.txt
