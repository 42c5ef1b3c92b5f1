# the same around the FIRST and SECOND round boundary (1024 / 2048 workgroups of 21 column blocks per node group)
for m in 52 55 56 58 64 72 78 79 82; do
  timeout -k 10 200 python bench.py --config c3 --map $m --no-cpu --no-other-arith --no-data-variants --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); n=$m*$m; g=(n+63)//64; print('map', $m, 'nodes', n, 'groups', g, 'wgs', g*21, 'rounds %.2f'%(g*21/1024), 'update', d['kernel_ms_per_step']['update'], 'us/group %.3f'%(d['kernel_ms_per_step']['update']*1e3/g))"
done
