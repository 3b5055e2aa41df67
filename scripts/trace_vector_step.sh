# Development aid (GPU box): kernel sequence of one collect vector step and of one minibatch of the bench loop
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_vs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_vs -o t -- python $R/bench.py --no-cpu-baseline --extra-configs 0 --steps 3 --warmup 2 --repeats 1 > /dev/null 2> /tmp/trace_vs.err < /dev/null
f=$(find /tmp/trace_vs -name "*kernel_trace.csv" | head -1)
echo "trace: $f"
python $R/scripts/trace_sequence.py "$f" env_step_post | cut -c1-150
echo ==== minibatch
python $R/scripts/trace_sequence.py "$f" gather_rows | cut -c1-150
