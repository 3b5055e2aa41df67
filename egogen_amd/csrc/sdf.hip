// calc_sdf (crowd_ppo/utils.py:54-84) as a standalone gather kernel: one point per lane, grid-stride.
#include "egx_common.h"

__global__ __launch_bounds__(256) void egx_sdf_sample_kernel(SdfDev s, const float* __restrict__ pts, int64_t n,
                                                            float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    out[i] = egx_sdf_neg_trilinear(s, x, y, z);
  }
}

// Bracket table of the penetration count (egx_lbs_forward): entry (jx,jy,jz) of a [(c0+2)][(c1+2)][(c2+2)] grid holds
// {min,max} of the fine samples the interpolation can touch for a point whose UNCLAMPED voxel coordinate falls into
// that cell.  Per axis: j = 0 -> the point is clamped onto the first sample plane {0}; j = c+1 -> onto the last plane
// {d-1} (grid_sample padding_mode="border": the neighbouring layer has weight exactly 0 or is dropped); 1 <= j <= c ->
// block b = j-1, footprint [4b, min(4b+4, d-1)].  The footprint of an entry is the product of the three ranges, so faces,
// edges and corners of the cube get their own (tight) brackets and a lookup is one uniform index computation.
__global__ void egx_sdf_build_coarse_kernel(const float* __restrict__ grid, int d0, int d1, int d2, int c0, int c1, int c2,
                                            float2* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int e1 = c1 + 2, e2 = c2 + 2;
  if (idx >= (c0 + 2) * e1 * e2) return;
  const int jz = idx % e2, jy = (idx / e2) % e1, jx = idx / (e1 * e2);
  auto range = [](int j, int c, int d, int& lo, int& hi) {
    if (j == 0) { lo = hi = 0; }
    else if (j == c + 1) { lo = hi = d - 1; }
    else { lo = 4 * (j - 1); hi = min(lo + 4, d - 1); }
  };
  int x0, x1, y0, y1, z0, z1;
  range(jx, c0, d0, x0, x1); range(jy, c1, d1, y0, y1); range(jz, c2, d2, z0, z1);
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int x = x0; x <= x1; ++x)
    for (int y = y0; y <= y1; ++y)
      for (int z = z0; z <= z1; ++z) {
        const float v = grid[((size_t)x * d1 + y) * d2 + z];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
  out[idx] = make_float2(mn, mx);
}

extern "C" size_t egx_sdf_coarse_bytes(int d0, int d1, int d2) {
  if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
  const size_t c0 = egx_ceil_div(d0, 4), c1 = egx_ceil_div(d1, 4), c2 = egx_ceil_div(d2, 4);
  return (c0 + 2) * (c1 + 2) * (c2 + 2) * sizeof(float2);
}

extern "C" int egx_sdf_build_coarse(const egx_sdf_grid* sdf, void* coarse_out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && coarse_out && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad arguments");
  const int c0 = egx_ceil_div(sdf->d0, 4), c1 = egx_ceil_div(sdf->d1, 4), c2 = egx_ceil_div(sdf->d2, 4);
  const int n = (c0 + 2) * (c1 + 2) * (c2 + 2);
  hipLaunchKernelGGL(egx_sdf_build_coarse_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     sdf->grid, sdf->d0, sdf->d1, sdf->d2, c0, c1, c2, static_cast<float2*>(coarse_out));
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_sdf_sample(const egx_sdf_grid* sdf, const float* pts, int64_t n, float* out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad sdf grid");
  EGX_REQUIRE(egx_sdf_dims_ok(sdf->d0, sdf->d1, sdf->d2), "sdf grid needs d2 >= 2 and fewer than 2^32 samples");
  if (n == 0) return EGX_OK;  // empty input is legal
  EGX_REQUIRE(pts && out && n > 0, "null points / output");
  SdfDev s{sdf->grid, sdf->d0, sdf->d1, sdf->d2, sdf->center[0], sdf->center[1], sdf->center[2], sdf->scale, nullptr, 0, 0, 0};
  const int64_t blocks = (n + 255) / 256;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  hipLaunchKernelGGL(egx_sdf_sample_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), s, pts, n, out);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
