set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_fullsize_properties.py -x -q -m gpu > gpurun_out/r5j/tests.log 2>&1 || { tail -40 gpurun_out/r5j/tests.log; exit 1; }
tail -2 gpurun_out/r5j/tests.log
python tools/exp/ab_env.py VSOM_SL_EPI f64 f32 2>/dev/null | grep '^{'
python tools/exp/ab_env.py VSOM_SL_RING old two 2>/dev/null | grep '^{'
python tools/exp/ab_env.py VSOM_SL_RING old two --data float 2>/dev/null | grep '^{'
python tools/exp/ab_env.py VSOM_SL_EPI f64 f32 --map 64 2>/dev/null | grep '^{'
bash tools/exp/sl_dbg2.sh 2>&1 | grep "dbg=0\|dbg=2\|dbg=1 "
