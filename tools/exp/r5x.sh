set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_ingest.py tests/test_gpu_batch_parity.py tests/test_gpu_short_vectors.py tests/test_gpu_goldens.py tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py -x -q > gpurun_out/r5x/tests.log 2>&1 || { tail -40 gpurun_out/r5x/tests.log; exit 1; }
tail -2 gpurun_out/r5x/tests.log
bash tools/exp/kstats.sh r5x_c3 --steps 20 | grep "stage_rows\|quant\|transpose\|cc_scan"
bash tools/exp/kstats.sh r5x_c4 --config c4 --steps 20 | grep "stage_rows\|quant"
