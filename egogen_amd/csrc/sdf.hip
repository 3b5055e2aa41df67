// calc_sdf (crowd_ppo/utils.py:54-84) as a standalone gather kernel: one point per lane, grid-stride.
#include "egx_common.h"

__global__ __launch_bounds__(256) void egx_sdf_sample_kernel(SdfDev s, const float* __restrict__ pts, int64_t n,
                                                            float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    out[i] = egx_sdf_neg_trilinear(s, x, y, z);
  }
}

// {min,max} over the fine samples a point inside coarse block (bx,by,bz) can touch: indices [4b, 4b+4] per axis
__global__ void egx_sdf_build_coarse_kernel(const float* __restrict__ grid, int d0, int d1, int d2, int c0, int c1, int c2,
                                            float2* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= c0 * c1 * c2) return;
  const int bz = idx % c2, by = (idx / c2) % c1, bx = idx / (c1 * c2);
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int x = 4 * bx; x <= min(4 * bx + 4, d0 - 1); ++x)
    for (int y = 4 * by; y <= min(4 * by + 4, d1 - 1); ++y)
      for (int z = 4 * bz; z <= min(4 * bz + 4, d2 - 1); ++z) {
        const float v = grid[((size_t)x * d1 + y) * d2 + z];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
  out[idx] = make_float2(mn, mx);
}

// Face brackets: a point clamped onto a border layer of the grid (outside the cube, or exactly on its first / last
// sample plane) only ever touches samples OF THAT LAYER (the neighbouring layer has weight exactly 0 or is dropped), so
// its bracket is taken over the layer alone.  Six 2-D tables of 4x4 blocks follow the block table:
// x-lo, x-hi over (y,z); y-lo, y-hi over (x,z); z-lo, z-hi over (x,y).
__global__ void egx_sdf_build_faces_kernel(const float* __restrict__ grid, int d0, int d1, int d2, int c0, int c1, int c2,
                                           float2* __restrict__ out /* already offset past the block table */) {
  const int n_x = c1 * c2, n_y = c0 * c2, n_z = c0 * c1;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * (n_x + n_y + n_z)) return;
  const int o = idx;
  int axis, hi;
  if (idx < 2 * n_x) { axis = 0; hi = idx / n_x; idx -= hi * n_x; }
  else if (idx < 2 * n_x + 2 * n_y) { idx -= 2 * n_x; axis = 1; hi = idx / n_y; idx -= hi * n_y; }
  else { idx -= 2 * n_x + 2 * n_y; axis = 2; hi = idx / n_z; idx -= hi * n_z; }
  const int du = (axis == 0) ? d1 : d0, dv = (axis == 2) ? d1 : d2, cv = (axis == 2) ? c1 : c2;
  const int bu = idx / cv, bv = idx % cv;
  const int layer = hi ? ((axis == 0) ? d0 : (axis == 1) ? d1 : d2) - 1 : 0;
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int u = 4 * bu; u <= min(4 * bu + 4, du - 1); ++u)
    for (int v = 4 * bv; v <= min(4 * bv + 4, dv - 1); ++v) {
      const int x = (axis == 0) ? layer : u, y = (axis == 0) ? u : (axis == 1) ? layer : v, z = (axis == 2) ? layer : v;
      const float val = grid[((size_t)x * d1 + y) * d2 + z];
      mn = fminf(mn, val);
      mx = fmaxf(mx, val);
    }
  out[o] = make_float2(mn, mx);
}

extern "C" size_t egx_sdf_coarse_bytes(int d0, int d1, int d2) {
  if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
  const size_t c0 = egx_ceil_div(d0, 4), c1 = egx_ceil_div(d1, 4), c2 = egx_ceil_div(d2, 4);
  return (c0 * c1 * c2 + 2 * (c1 * c2 + c0 * c2 + c0 * c1)) * sizeof(float2);
}

extern "C" int egx_sdf_build_coarse(const egx_sdf_grid* sdf, void* coarse_out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && coarse_out && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad arguments");
  const int c0 = egx_ceil_div(sdf->d0, 4), c1 = egx_ceil_div(sdf->d1, 4), c2 = egx_ceil_div(sdf->d2, 4);
  const int n = c0 * c1 * c2;
  hipLaunchKernelGGL(egx_sdf_build_coarse_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     sdf->grid, sdf->d0, sdf->d1, sdf->d2, c0, c1, c2, static_cast<float2*>(coarse_out));
  const int nf = 2 * (c1 * c2 + c0 * c2 + c0 * c1);
  hipLaunchKernelGGL(egx_sdf_build_faces_kernel, dim3(egx_ceil_div(nf, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     sdf->grid, sdf->d0, sdf->d1, sdf->d2, c0, c1, c2, static_cast<float2*>(coarse_out) + n);
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
