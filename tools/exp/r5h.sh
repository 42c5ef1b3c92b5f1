set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout -k 10 600 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "not clr and not c5 and not c4" > gpurun_out/r5h/tests.log 2>&1 || { tail -40 gpurun_out/r5h/tests.log; exit 1; }
tail -2 gpurun_out/r5h/tests.log
python tools/exp/ab_env.py VSOM_SL_RING old two 2>/dev/null | grep '^{'
python tools/exp/ab_env.py VSOM_SL_RING old two --map 64 2>/dev/null | grep '^{'
