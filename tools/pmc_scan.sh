#!/bin/bash
# PMC passes over bench.py's scan kernel (gpurun): usage tools/pmc_scan.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o g$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-hbm-point "$@" > /dev/null 2> $OUT/g$i.err
done
cd $REPO
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/g*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%scan_%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for n, cn, k, avg in rows:
        print(f"{n[:60]:60s} {cn:28s} n={k} avg={avg:.4g}")
PY
rm -rf $OUT/g*/
