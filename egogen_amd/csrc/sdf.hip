// calc_sdf (crowd_ppo/utils.py:54-84) as a standalone gather kernel: one point per lane, grid-stride.
#include "egx_common.h"

__global__ __launch_bounds__(256) void egx_sdf_sample_kernel(SdfDev s, const float* __restrict__ pts, int64_t n,
                                                            float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    out[i] = egx_sdf_neg_trilinear(s, x, y, z);
  }
}

extern "C" int egx_sdf_sample(const egx_sdf_grid* sdf, const float* pts, int64_t n, float* out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad sdf grid");
  if (n == 0) return EGX_OK;  // empty input is legal
  EGX_REQUIRE(pts && out && n > 0, "null points / output");
  SdfDev s{sdf->grid, sdf->d0, sdf->d1, sdf->d2, sdf->center[0], sdf->center[1], sdf->center[2], sdf->scale};
  const int64_t blocks = (n + 255) / 256;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  hipLaunchKernelGGL(egx_sdf_sample_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), s, pts, n, out);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
