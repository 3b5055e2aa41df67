set -u
mkdir -p gpurun_out/t
timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_env_gpu.py -x -q -m gpu -k "not egobody and not full_size" < /dev/null > gpurun_out/t/test.log 2>&1
echo "test rc=$?"; tail -3 gpurun_out/t/test.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --extra-configs 0 --steps 10 < /dev/null > gpurun_out/t/bench$i.json 2> gpurun_out/t/bench.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/t/bench$i.json
done
