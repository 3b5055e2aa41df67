// Development aid: practical ceiling of v_mfma_f32_32x32x2_f32 on this chip with the launch shape of
// egx_lbs_fused_kernel (256 threads, 6 accumulator tuples per wave, 1416 MFMAs per wave), no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, int iters, float seed) {
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-7f;
  }
  float s = 0.f;
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 59;  // x24 MFMAs = 1416 per wave
  for (int occ = 1; occ <= 2; ++occ)
    for (int grid : {512, 2048, 13120}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (occ == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)grid * 4 * iters * 24 * 4096.0;
        if (rep == 2) printf("launch_bounds occ=%d grid=%5d  %.3f ms  %.1f TFLOP/s\n", occ, grid, ms, fl / ms / 1e9);
      }
    }
  // long run for the sustained clock
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<2>, dim3(13120), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("50 back-to-back launches: %.3f ms each, %.1f TFLOP/s\n", ms / 50, 13120.0 * 4 * iters * 24 * 4096 / (ms / 50) / 1e9);
  }
  return 0;
}
