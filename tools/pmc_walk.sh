#!/bin/bash
# PMC pass (no trace domains besides --kernel-trace) over the HNSW walk kernels: where a wave's cycles go.
#   usage: tools/pmc_walk.sh <tag> [bench_hnsw args...]      -> gpurun_out/pmc_walk_<tag>/summary.txt
# SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked on s_waitcnt / barriers) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY (issuing), quad-cycles, summed over waves.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_walk_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o g$i -- python $REPO/tools/bench_hnsw.py "$@" > $OUT/g$i.out 2> $OUT/g$i.err
done
cd $REPO
python - > $OUT/summary.txt <<PY
import sqlite3, glob
print("rocprofv3 --pmc (two passes) over tools/bench_hnsw.py $*: per-launch averages of the walk kernels")
for db in sorted(glob.glob("$OUT/g*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%hnsw_search_kernel%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for n, cn, k, avg in rows:
        print(f"{n[:100]:100s} {cn:22s} launches={k} avg={avg:.5g}")
PY
rm -rf $OUT/g*/
cat $OUT/summary.txt
