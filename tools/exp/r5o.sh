set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout -k 10 900 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_fma_mode.py tests/test_gpu_batch_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_fullsize_properties.py tests/test_gpu_late_epochs.py -x -q > gpurun_out/r5o/tests.log 2>&1 || { tail -40 gpurun_out/r5o/tests.log; exit 1; }
tail -2 gpurun_out/r5o/tests.log
VSOM_LIB=tools/exp/bin/libvsom_dev.so python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_deadskip.hsaco | tee gpurun_out/r5o/ab_dead.jsonl
AB_MAP=64 VSOM_LIB=tools/exp/bin/libvsom_dev.so python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_deadskip.hsaco | tee -a gpurun_out/r5o/ab_dead.jsonl
