set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_matvec_kernels.py tests/test_gpu_parity.py tests/test_gpu_graphs.py tests/test_gpu_next_tier.py -x -q -m gpu > gpurun_out/t_a.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/t_a.log
timeout -k 10 300 python tools/bench_configs.py c2 2>&1 | tail -3
MFX_LIBRARY_PATH=tools/ab/libmfx_r03.so timeout -k 10 300 python tools/bench_configs.py c2 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/bench_r04a.json'))
print(l['ms_per_step'], l['ms_per_step_with_kernel_timers'], l['breakdown_ms_per_step'], l['param_grad_gemm']['achieved_TFLOPs'], l['roofline']['frac'], l['modes'])
PY
