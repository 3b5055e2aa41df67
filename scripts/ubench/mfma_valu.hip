// Development aid: do fp32 VALU FMAs hide under v_mfma_f32_32x32x2_f32 on gfx950, or do they add?
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/mfma_valu.hip -o /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// every wave: ITERS x (24 MFMAs + NV scalar FMAs [PK: NV/2 packed FMAs]) ; ROLE 1: odd waves of a SIMD pair do only VALU
template <int NV, bool PK, int ROLE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  float v[16];
  f32x2 v2[8];
  for (int i = 0; i < 16; ++i) v[i] = seed * i;
  for (int i = 0; i < 8; ++i) v2[i] = f32x2{seed * i, seed + i};
  const bool do_mfma = (ROLE == 0) || (wave < 4);
  const bool do_valu = (ROLE == 0) || (wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
      if (PK) {
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) v2[j % 8] = __builtin_elementwise_fma(v2[j % 8], f32x2{a, a}, f32x2{b, b});
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j % 16] = fmaf(v[j % 16], a, b);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += v2[i][0] + v2[i][1];
  if (s == 123.456f) out[0] = s;
}

template <int NV, bool PK, int ROLE>
void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 600, grid = 256 * 4;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, PK, ROLE>), dim3(grid), dim3(512), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double mf = (double)grid * (ROLE ? 4 : 8) * iters * 24 * 4096.0;
  const double cyc_per_iter = best * 1e-3 * 2.4e9 / (grid / 256) / iters;  // per SIMD (2 waves), nominal 2.4 GHz
  printf("%-44s %.3f ms  MFMA %.1f TFLOP/s  %.0f nominal cyc per iteration-pair\n", name, best, mf / best / 1e9, cyc_per_iter);
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<0, false, 0>(out, "MFMA only (both waves)");
  run<48, false, 0>(out, "+48 v_fma per 24 MFMA, same wave");
  run<96, false, 0>(out, "+96 v_fma per 24 MFMA, same wave");
  run<192, false, 0>(out, "+192 v_fma per 24 MFMA, same wave");
  run<96, true, 0>(out, "+48 v_pk_fma per 24 MFMA, same wave");
  run<192, true, 0>(out, "+96 v_pk_fma per 24 MFMA, same wave");
  run<0, false, 1>(out, "role split: 4 waves MFMA, 4 idle");
  run<96, false, 1>(out, "role split: MFMA | 96 v_fma");
  run<192, false, 1>(out, "role split: MFMA | 192 v_fma");
  run<384, false, 1>(out, "role split: MFMA | 384 v_fma");
  run<384, true, 1>(out, "role split: MFMA | 192 v_pk_fma");
  return 0;
}
