#!/bin/bash
# PMC passes over the C3 walk at full size (tools/walk_variants.py, the graph built once and cached on the box's /tmp): where the wave cycles go and what the
# memory system moves per launch.   usage: tools/pmc_walk10m.sh <tag> [walk_variants args...]   -> gpurun_out/pmc_walk10m_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_walk10m_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
python $REPO/tools/walk_variants.py --graph-cache /tmp/c3_graph.npz "$@" > $OUT/plain.jsonl 2> $OUT/plain.err
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o g$i -- python $REPO/tools/walk_variants.py --graph-cache /tmp/c3_graph.npz --reps 2 "$@" > $OUT/g$i.out 2> $OUT/g$i.err
done
cd $REPO
python - > $OUT/summary.txt <<PY
import sqlite3, glob
print("rocprofv3 --pmc over tools/walk_variants.py $*: per-launch averages of the walk kernel (launch order = the variants' order, 1 warm-up + reps launches each)")
for db in sorted(glob.glob("$OUT/g*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%hnsw_search_kernel%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for n, cn, k, avg, mn, mx in rows:
        print(f"{n[:100]:100s} {cn:22s} launches={k} avg={avg:.5g} min={mn:.5g} max={mx:.5g}")
PY
rm -rf $OUT/g*/
cat $OUT/plain.jsonl | cut -c1-330
cat $OUT/summary.txt
