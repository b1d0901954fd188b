#!/bin/bash
# PMC passes (no trace domains besides --kernel-trace) over the 128-query TurboQuant pass (scan_tq4w.hip): where the waves' cycles go, the LDS, HBM traffic.
#   usage: tools/pmc_tqw.sh <tag> [tq_wide_bench args...]      -> gpurun_out/pmc_tqw_<tag>/summary.txt
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_tqw_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o g$i -- python $REPO/tools/tq_wide_bench.py --reps 2 "$@" > $OUT/g$i.out 2> $OUT/g$i.err
done
cd $REPO
python - > $OUT/summary.txt <<PY
import sqlite3, glob
print("rocprofv3 --pmc (one pass per group) over tools/tq_wide_bench.py --reps 2 $*: per-launch averages")
for db in sorted(glob.glob("$OUT/g*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%scan_tq4w_kernel%' or kernel_name like '%scan_sqw_kernel%' or kernel_name like '%scan_sq_mfma_kernel%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for n, cn, k, avg in rows:
        print(f"{n[:70]:70s} {cn:26s} launches={k} avg={avg:.6g}")
PY
rm -rf $OUT/g*/
cat $OUT/summary.txt
