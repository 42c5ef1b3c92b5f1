# does the chain kernel's time follow ceil(workgroups / 1024 slots)?  C3 data, square maps around round boundaries
for m in 104 110 111 113 116 120 124 128; do
  timeout -k 10 200 python bench.py --config c3 --map $m --no-cpu --no-other-arith --no-data-variants --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); n=$m*$m; g=(n+63)//64; print('map', $m, 'nodes', n, 'groups', g, 'wgs', g*21, 'rounds %.2f'%(g*21/1024), 'update', d['kernel_ms_per_step']['update'], 'us/group %.3f'%(d['kernel_ms_per_step']['update']*1e3/g))"
done
