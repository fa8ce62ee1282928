set -u
for occ in 16 17 18 15 16 17; do
SEEKR_COUNT_OCC=$occ python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16f8-arm 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('occ $occ', d['kernels_ms_per_step']['count_kmers_f32'], d['roofline_count']['frac'], d['ms_per_step'])"
done
