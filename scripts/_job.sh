set -u
mkdir -p gpurun_out/j4
timeout 1500 python -m pytest tests -x -q -m gpu < /dev/null > gpurun_out/j4/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/j4/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline --extra-configs 0 --steps 10 < /dev/null > gpurun_out/j4/bench.json 2> gpurun_out/j4/bench.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/j4/bench.json
