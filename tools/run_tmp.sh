set -o pipefail
mkdir -p gpurun_out/r04
bash tools/prof_bench.sh > gpurun_out/r04/prof_bench2.log 2>&1; echo "prof_bench rc=$?"; tail -6 gpurun_out/r04/prof_bench2.log | cut -c1-200
cp gpurun_out/prof_bench/kernel_stats.csv gpurun_out/r04/bench2_kernel_stats.csv; cp gpurun_out/prof_bench/bench_under_rocprof.json gpurun_out/r04/bench2_under_rocprof.json
timeout -k 10 600 python bench.py > gpurun_out/r04/bench2_default.json 2> gpurun_out/r04/bench2_default.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r04/bench2_default.json'))
print(l['ms_per_step'], l['ms_per_step_with_kernel_timers'], l['value'], l['breakdown_ms_per_step'], l['param_grad_gemm']['achieved_TFLOPs'], l['roofline']['frac'], l['roofline']['avg_launch_ms'], l['modes'])
PY
bash tools/prof_grad.sh > gpurun_out/r04/prof_grad2.log 2>&1; echo "prof_grad rc=$?"; grep -v amdgpu gpurun_out/r04/prof_grad2.log | tail -30
