set -u
mkdir -p gpurun_out/j1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/xcc_probe.hip -o /tmp/xcc_probe 2>/dev/null && /tmp/xcc_probe > gpurun_out/j1/xcc.txt 2>&1
cat gpurun_out/j1/xcc.txt
bash scripts/ab_lbs.sh build/lib_r02_head.so build/lib_scalar_fma.so build/lib_scalar_setprio.so > gpurun_out/j1/ab.txt 2>&1
cat gpurun_out/j1/ab.txt
timeout 300 python scripts/overlap_probe2.py 256 > gpurun_out/j1/ov256.txt 2>&1; cat gpurun_out/j1/ov256.txt
timeout 300 python scripts/overlap_probe2.py 512 > gpurun_out/j1/ov512.txt 2>&1; cat gpurun_out/j1/ov512.txt
