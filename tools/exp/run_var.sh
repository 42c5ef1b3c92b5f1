# times bench c3 with variant code objects through the development build of the library (tools/exp/bin/libvsom_dev.so)
cp variational-self-organizing-maps_amd/libvsom_hip.so /tmp/keep.so
cp tools/exp/bin/libvsom_dev.so variational-self-organizing-maps_amd/libvsom_hip.so
for v in $*; do
  VSOM_ASM_HSACO=$PWD/tools/exp/bin/$v.hsaco timeout -k 10 300 python bench.py --config c3 --no-cpu --no-other-arith --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms_per_step']['update'])"
done
cp /tmp/keep.so variational-self-organizing-maps_amd/libvsom_hip.so
