set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
VSOM_SL_SWEEP_N=400 timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py -x -q -k random_shortlist > gpurun_out/r5q/sl_sweep.log 2>&1 || { tail -30 gpurun_out/r5q/sl_sweep.log; exit 1; }
tail -2 gpurun_out/r5q/sl_sweep.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5q/tests.log 2>&1 || { tail -40 gpurun_out/r5q/tests.log; exit 1; }
tail -2 gpurun_out/r5q/tests.log
