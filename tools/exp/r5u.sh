set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5u/tests.log 2>&1 || { tail -40 gpurun_out/r5u/tests.log; exit 1; }
tail -2 gpurun_out/r5u/tests.log
bash tools/wide_sweep.sh r5_sweep 21 22 | tee gpurun_out/r5u/sweep.txt
