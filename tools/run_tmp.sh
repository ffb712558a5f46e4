set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "rccl" > gpurun_out/t_e.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_e.log
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --share-gpus --steps 1 --warmup 0 --no-modes --no-cpu-baseline > gpurun_out/r04/bench_2rank_gloo.json 2> gpurun_out/r04/bench_2rank_gloo.err; echo "bench2 rc=$?"; tail -3 gpurun_out/r04/bench_2rank_gloo.err
python - <<'PY'
import json
l=json.load(open('gpurun_out/r04/bench_2rank_gloo.json'))
print(l['n_gpus'], l['ms_per_step'], l['config']['parallelism'], l['result'])
PY
