set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rbf or slq" > gpurun_out/t_parity_rbf.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/t_parity_rbf.log
for i in 1 2; do
  MFX_LIBRARY_PATH=tools/ab/libmfx_r03.so timeout -k 10 120 python tools/bench_grad.py 131072 2560 3 2>&1 | tail -1
  timeout -k 10 120 python tools/bench_grad.py 131072 2560 3 2>&1 | tail -1
done
