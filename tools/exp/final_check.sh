# Development: the validation run before a commit of kernel changes (full GPU suite, a 400-case search sweep, C4 profiles, the default bench line).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5fin
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5fin/tests.log 2>&1 || { tail -40 gpurun_out/r5fin/tests.log; exit 1; }
tail -2 gpurun_out/r5fin/tests.log
VSOM_SL_SWEEP_N=400 VSOM_SWEEP_SEED=77 timeout -k 10 600 python -m pytest tests/test_gpu_shortlist.py -x -q -k random_shortlist 2>&1 | tail -1
bash tools/collect_profiles.sh r5_c4 --config c4 --steps 10 > /dev/null 2>&1; echo c4 profiles done
python bench.py > gpurun_out/r5fin/bench_default.json 2> gpurun_out/r5fin/bench_default.err
python tools/exp/show_bench.py gpurun_out/r5fin/bench_default.json
