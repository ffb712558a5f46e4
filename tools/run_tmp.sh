set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python tools/bench_configs.py c1 c2 c3 c5 > gpurun_out/r04/other_configs.log 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/r04/other_configs.log | tail -14
timeout -k 10 300 python bench.py --kernel matern32 --steps 2 --warmup 1 --no-cpu-baseline --no-modes 2>/dev/null | python -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('matern32 C4 step', round(l['ms_per_step'],1), {k:round(v,1) for k,v in l['breakdown_ms_per_step'].items()})"
