set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5y/tests.log 2>&1 || { tail -40 gpurun_out/r5y/tests.log; exit 1; }
tail -2 gpurun_out/r5y/tests.log
for c in c3 c2 c4 c5; do python bench.py --config $c --no-cpu --no-data-variants > gpurun_out/r5y/bench_$c.json 2> gpurun_out/r5y/bench_$c.err; done
python tools/exp/show_bench.py gpurun_out/r5y/bench_c3.json gpurun_out/r5y/bench_c2.json gpurun_out/r5y/bench_c4.json gpurun_out/r5y/bench_c5.json
