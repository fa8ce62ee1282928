set -u
python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_count']['frac'], d['verified'], d['cpu_baseline']['value'], d['f16f8_arm']['value'], d['f16f8_arm']['worst_error_over_bar'])"
