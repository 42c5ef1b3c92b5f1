cd $GRAFT_REPO_ROOT
for i in 1 2; do
  VSOM_LIB=$GRAFT_REPO_ROOT/tools/exp/bin/libvsom_headA.so python bench.py --config c5 --no-cpu --no-data-variants > gpurun_out/c5_A$i.json 2>/dev/null
  python tools/exp/show_bench.py gpurun_out/c5_A$i.json | cut -c1-190
  python bench.py --config c5 --no-cpu --no-data-variants > gpurun_out/c5_B$i.json 2>/dev/null
  python tools/exp/show_bench.py gpurun_out/c5_B$i.json | cut -c1-190
done
