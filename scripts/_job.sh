set -u
timeout 600 python -m pytest tests/test_nets_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 | tail -8
echo "== packed"; timeout 300 python scripts/bench_policy.py 2>&1 | grep "A="
echo "== fp32 path"; EGX_POLICY_PACKED=0 timeout 300 python scripts/bench_policy.py 2>&1 | grep "A="
