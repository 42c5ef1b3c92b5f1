# times bench c3 with variant code objects through the development build of the library (tools/exp/bin/libvsom_dev.so).
# The shipped library is never overwritten: the binding loads the file VSOM_LIB names (variational-self-organizing-maps_amd/capi.py).
for v in $*; do
  VSOM_LIB=$PWD/tools/exp/bin/libvsom_dev.so VSOM_ASM_HSACO=$PWD/tools/exp/bin/$v.hsaco timeout -k 10 300 python bench.py --config c3 --no-cpu --no-other-arith --no-data-variants --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms_per_step']['update'])"
done
