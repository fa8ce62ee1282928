# A/B of the host's wait policy on the default bench step (run on the GPU box): gap = ms_per_step - sum of kernel times
set -u
for rep in 1 2 3; do
  for w in "" spin yield block; do
    SEEKR_HOST_WAIT=$w python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-target-200k --no-f16f8-arm 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernels_ms_per_step']
print('wait=%-6s step %.3f ms  kernels %.3f  gap %.3f' % ('$w' or 'auto', d['ms_per_step'], sum(k.values()), d['ms_per_step']-sum(k.values())))"
  done
done
