#!/bin/bash
# End-of-round checks on the GPU box: the whole `-m gpu` suite, __graft_entry__.smoke(), the default bench line and the
# rocprofv3 kernel-trace summary of the bench command; everything lands under gpurun_out/final/.
#   gpurun --timeout 4500 -- 'bash scripts/run_final_checks.sh'
set -u
mkdir -p gpurun_out/final
export EGX_P3_TABLE=$PWD/gpurun_out/final/p3_table.txt EGX_C5_HIST=$PWD/gpurun_out/final/c5_hist.txt EGX_TOL_REPORT=$PWD/gpurun_out/final/env_tolerance_usage.txt
rm -f $EGX_P3_TABLE $EGX_TOL_REPORT
EGX_DRIFT_TABLE=gpurun_out/final/drift_table.txt timeout 3000 python -m pytest tests -x -q -m gpu < /dev/null > gpurun_out/final/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/final/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > gpurun_out/final/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --extra-configs 2 < /dev/null > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/final/bench.json; cp bench_detail.json gpurun_out/final/bench_detail.json
bash scripts/run_profile.sh final > gpurun_out/final/profile.log 2>&1
tail -3 gpurun_out/final/profile.log
PMC_SETS="FETCH_SIZE WRITE_SIZE TCC_HIT_sum,TCC_MISS_sum GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU_MFMA_MOPS_F16 TCP_TCC_READ_REQ_sum,TCP_TOTAL_CACHE_ACCESSES_sum SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU" bash scripts/pmc_lbs.sh final_pmc > gpurun_out/final/pmc.log 2>&1
tail -12 gpurun_out/final/pmc.log
