set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_random_shapes.py tests/test_gpu_batch_parity.py -x -q > gpurun_out/r3_t3.log 2>&1; tail -2 gpurun_out/r3_t3.log
bash tools/collect_profiles.sh r3_c3 --steps 10 > /dev/null 2>&1; echo c3 done
bash tools/collect_profiles.sh r3_c3_sigma --steps 10 --arith sigma --no-other-arith > /dev/null 2>&1; echo c3 sigma done
bash tools/collect_profiles.sh r3_c3_contracted --steps 10 --arith contracted --no-other-arith > /dev/null 2>&1; echo c3 contracted done
bash tools/collect_profiles.sh r3_c2 --config c2 --steps 10 > /dev/null 2>&1; echo c2 done
bash tools/collect_profiles.sh r3_c4 --config c4 --steps 10 > /dev/null 2>&1; echo c4 done
bash tools/collect_profiles.sh r3_c5 --config c5 --steps 10 > /dev/null 2>&1; echo c5 done
bash tools/collect_profiles.sh r3_online --config online --steps 5 > /dev/null 2>&1; echo online done
ls gpurun_out/r3_c3/summary
