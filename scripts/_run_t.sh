set -u
mkdir -p gpurun_out/t
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_multirank_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "not bf16_policy and not egobody" < /dev/null > gpurun_out/t/test.log 2>&1
echo "test rc=$?"; tail -8 gpurun_out/t/test.log
timeout 600 python bench.py --no-cpu-baseline --extra-configs 0 --steps 10 < /dev/null > gpurun_out/t/bench.json 2> gpurun_out/t/bench.err
echo "bench rc=$?"; cut -c1-250 gpurun_out/t/bench.json
