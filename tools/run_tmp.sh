set -o pipefail
timeout -k 10 240 python -m pytest tests/test_gpu_sharded.py -x -v -m gpu -k "eight_logical" > gpurun_out/t_b.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -8 gpurun_out/t_b.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/rehearse_eight_ranks.py > gpurun_out/r04/rehearse8.json 2> gpurun_out/r04/rehearse8.err; echo "rehearse rc=$?"; tail -3 gpurun_out/r04/rehearse8.err; cat gpurun_out/r04/rehearse8.json
