export TMPDIR=/tmp
R=$PWD
cd /tmp
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
rocprofv3 --pmc $grp --kernel-trace -d /tmp/pp -o g -- python $R/tools/tq_wide_bench.py --storage sq --reps 2 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/pp/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for n, cn, k, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%scan_sqw_kernel%' group by kernel_name, counter_name"):
        print(n[:40], cn, k, avg)
PY
rm -rf /tmp/pp
done
