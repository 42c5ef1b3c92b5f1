# is the ~0.2 ms above the marginal rate a fixed cost per launch or proportional to the chain length?
for ch in 2048 4096 8192 16384; do for m in 55 110; do
  timeout -k 10 200 python bench.py --config c3 --map $m --chunk $ch --no-cpu --no-other-arith --no-data-variants --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); n=$m*$m; g=(n+63)//64; u=d['kernel_ms_per_step']['update']; print('map', $m, 'chunk', $ch, 'groups', g, 'update', u, 'us/group/4096 samples %.3f'%(u*1e3/g*4096/$ch))"
done; done
