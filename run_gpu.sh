cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout 900 python -m pytest tests -m gpu -q --tb=line -x 2> gpurun_out/stderr.log | grep -v "^$" | tail -30 > gpurun_out/pytest_gpu.log
grep -v "^$" gpurun_out/stderr.log | sort | uniq -c | sort -rn | head -30 >> gpurun_out/pytest_gpu.log
