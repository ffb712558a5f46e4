set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_full3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_full3.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('bench', round(l['ms_per_step'],1), {k:round(v,1) for k,v in l['breakdown_ms_per_step'].items()}, l['roofline']['frac'])"
