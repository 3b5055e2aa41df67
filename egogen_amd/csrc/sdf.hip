// calc_sdf (crowd_ppo/utils.py:54-84) as a standalone gather kernel: four points per lane and trip, grid-stride.  What bounds it is
// the L2 -> L1 line traffic of the corner gathers (a 128-byte line per 8-byte corner pair), not the 16 bytes per point of
// coordinates and output: profiles/r06_sdf_sample.md (incl. the LDS-staged per-body variant that was built, measured slower and
// removed).
#include "egx_common.h"

#ifndef EGX_SDF_PPT
#define EGX_SDF_PPT 4   // points per lane and trip: their 4 x PPT corner-pair gathers are in flight together
#endif
__global__ __launch_bounds__(256) void egx_sdf_sample_kernel(SdfDev s, const float* __restrict__ pts, int64_t n,
                                                            float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += EGX_SDF_PPT * stride) {
    float r[EGX_SDF_PPT];
#pragma unroll
    for (int u = 0; u < EGX_SDF_PPT; ++u) {
      const int64_t i = min(i0 + u * stride, n - 1);
      r[u] = egx_sdf_neg_trilinear(s, pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]);
    }
#pragma unroll
    for (int u = 0; u < EGX_SDF_PPT; ++u)
      if (i0 + u * stride < n) out[i0 + u * stride] = r[u];
  }
}

// The same values from the bricked copy of the grid (egx_common.h: egx_sdf_bricks_offset; built by egx_sdf_build_coarse): what
// bounds the kernel above is the number of 128-byte lines its corner gathers pull from L2, and a point's eight corners span four
// lines of the row-major grid but 2.3 on average of the bricked one.  Same samples, same arithmetic: bit-identical values.
__device__ __forceinline__ float egx_brick_at(const float* __restrict__ bricks, unsigned nb1, unsigned nb2, unsigned x, unsigned y, unsigned z) {
  const unsigned b = ((x >> 2) * nb1 + (y >> 2)) * nb2 + (z >> 2);
  return bricks[(size_t)b * 64 + ((x & 3) << 4) + ((y & 3) << 2) + (z & 3)];
}
__device__ __forceinline__ float egx_sdf_neg_trilinear_bricks(const SdfDev& s, const float* __restrict__ bricks, float x, float y, float z) {
  float px, py, pz;
  egx_sdf_voxel_coords(s, x, y, z, px, py, pz);
  const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
  const float wx1 = egx_sub(px, x0), wy1 = egx_sub(py, y0), wz1 = egx_sub(pz, z0);
  const float wx0 = egx_sub(egx_add(x0, 1.f), px), wy0 = egx_sub(egx_add(y0, 1.f), py), wz0 = egx_sub(egx_add(z0, 1.f), pz);
  const unsigned ix0 = (unsigned)x0, iy0 = (unsigned)y0, iz0 = (unsigned)z0;
  const bool x1_in = ix0 + 1 < (unsigned)s.d0, y1_in = iy0 + 1 < (unsigned)s.d1, z1_in = iz0 + 1 < (unsigned)s.d2;
  const unsigned zb = z1_in ? iz0 : iz0 - 1;                      // pair (zb, zb+1) always inside the row
  const unsigned ix1 = x1_in ? ix0 + 1 : ix0, iy1 = y1_in ? iy0 + 1 : iy0;   // off-grid corners: weight exactly 0, any valid sample
  const unsigned nb1 = (unsigned)s.c1, nb2 = (unsigned)s.c2;
  const float c00x = egx_brick_at(bricks, nb1, nb2, ix0, iy0, zb), c00y = egx_brick_at(bricks, nb1, nb2, ix0, iy0, zb + 1);
  const float c01x = egx_brick_at(bricks, nb1, nb2, ix0, iy1, zb), c01y = egx_brick_at(bricks, nb1, nb2, ix0, iy1, zb + 1);
  const float c10x = egx_brick_at(bricks, nb1, nb2, ix1, iy0, zb), c10y = egx_brick_at(bricks, nb1, nb2, ix1, iy0, zb + 1);
  const float c11x = egx_brick_at(bricks, nb1, nb2, ix1, iy1, zb), c11y = egx_brick_at(bricks, nb1, nb2, ix1, iy1, zb + 1);
  const float wx1e = x1_in ? wx1 : 0.f, wy1e = y1_in ? wy1 : 0.f, wz1e = z1_in ? wz1 : 0.f;
  const float v000 = z1_in ? c00x : c00y, v010 = z1_in ? c01x : c01y, v100 = z1_in ? c10x : c10y, v110 = z1_in ? c11x : c11y;
  const float a00 = egx_mul(wx0, wy0), a01 = egx_mul(wx0, wy1e), a10 = egx_mul(wx1e, wy0), a11 = egx_mul(wx1e, wy1e);
  float acc;
  acc = egx_mul(v000, egx_mul(a00, wz0));
  acc = egx_add(acc, egx_mul(c00y, egx_mul(a00, wz1e)));
  acc = egx_add(acc, egx_mul(v010, egx_mul(a01, wz0)));
  acc = egx_add(acc, egx_mul(c01y, egx_mul(a01, wz1e)));
  acc = egx_add(acc, egx_mul(v100, egx_mul(a10, wz0)));
  acc = egx_add(acc, egx_mul(c10y, egx_mul(a10, wz1e)));
  acc = egx_add(acc, egx_mul(v110, egx_mul(a11, wz0)));
  acc = egx_add(acc, egx_mul(c11y, egx_mul(a11, wz1e)));
  return -acc;
}
__global__ __launch_bounds__(256) void egx_sdf_sample_bricks_kernel(SdfDev s, const float* __restrict__ bricks, const float* __restrict__ pts,
                                                                   int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += EGX_SDF_PPT * stride) {
    float r[EGX_SDF_PPT];
#pragma unroll
    for (int u = 0; u < EGX_SDF_PPT; ++u) {
      const int64_t i = min(i0 + u * stride, n - 1);
      r[u] = egx_sdf_neg_trilinear_bricks(s, bricks, pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]);
    }
#pragma unroll
    for (int u = 0; u < EGX_SDF_PPT; ++u)
      if (i0 + u * stride < n) out[i0 + u * stride] = r[u];
  }
}
__global__ __launch_bounds__(256) void egx_sdf_build_bricks_kernel(const float* __restrict__ grid, int d0, int d1, int d2, int c0, int c1, int c2,
                                                                  float* __restrict__ bricks) {
  const size_t n = (size_t)c0 * c1 * c2 * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned e = (unsigned)(i & 63);
    const size_t b = i >> 6;
    const int bz = (int)(b % c2), by = (int)((b / c2) % c1), bx = (int)(b / ((size_t)c1 * c2));
    const int x = min(bx * 4 + (int)(e >> 4), d0 - 1), y = min(by * 4 + (int)((e >> 2) & 3), d1 - 1), z = min(bz * 4 + (int)(e & 3), d2 - 1);
    bricks[i] = grid[((size_t)x * d1 + y) * d2 + z];
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

// free-space pyramid (egx_common.h): level l entry = max over its 2^l-cubed block of padded bracket cells of their `max`
__global__ void egx_sdf_build_mip_kernel(const float2* __restrict__ table, int c0, int c1, int c2, int l, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int e0 = egx_sdf_mip_dim(c0, l), e1 = egx_sdf_mip_dim(c1, l), e2 = egx_sdf_mip_dim(c2, l);
  if (idx >= e0 * e1 * e2) return;
  const int iz = idx % e2, iy = (idx / e2) % e1, ix = idx / (e1 * e2);
  const int n = 1 << l;
  float mx = -3.4e38f;
  for (int x = ix * n; x < min((ix + 1) * n, c0 + 2); ++x)
    for (int y = iy * n; y < min((iy + 1) * n, c1 + 2); ++y)
      for (int z = iz * n; z < min((iz + 1) * n, c2 + 2); ++z) mx = fmaxf(mx, table[((size_t)x * (c1 + 2) + y) * (c2 + 2) + z].y);
  out[idx] = mx;
}

// aux[0..2] (egx_common.h): largest |difference| of neighbouring samples along each axis; aux[3]: the steepest slope of the
// interpolated field in value per METRE.  Inside a cell every component of the trilinear gradient is a convex combination of the
// cell's four edge differences along that axis, so |gradient| <= sqrt(sum_a (k_a max edge_a)^2) per cell, k_a = samples per
// metre; outside the cube the border clamp only removes components.  Non-negative floats order like their bit patterns, so the
// reductions are integer atomicMax.
__global__ __launch_bounds__(256) void egx_sdf_axis_steps_kernel(const float* __restrict__ grid, int d0, int d1, int d2, float k0, float k1,
                                                                float k2, unsigned* __restrict__ aux) {
  const size_t n = (size_t)d0 * d1 * d2;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int z = (int)(i % d2), y = (int)((i / d2) % d1), x = (int)(i / ((size_t)d1 * d2));
    const int x1 = min(x + 1, d0 - 1), y1 = min(y + 1, d1 - 1), z1 = min(z + 1, d2 - 1);
    auto at = [&](int xx, int yy, int zz) { return grid[((size_t)xx * d1 + yy) * d2 + zz]; };
    // the cell with origin (x, y, z) (degenerate along an axis at the last plane: those edges are zero)
    const float v000 = at(x, y, z), v001 = at(x, y, z1), v010 = at(x, y1, z), v011 = at(x, y1, z1);
    const float v100 = at(x1, y, z), v101 = at(x1, y, z1), v110 = at(x1, y1, z), v111 = at(x1, y1, z1);
    const float ex = fmaxf(fmaxf(fabsf(v100 - v000), fabsf(v101 - v001)), fmaxf(fabsf(v110 - v010), fabsf(v111 - v011)));
    const float ey = fmaxf(fmaxf(fabsf(v010 - v000), fabsf(v011 - v001)), fmaxf(fabsf(v110 - v100), fabsf(v111 - v101)));
    const float ez = fmaxf(fmaxf(fabsf(v001 - v000), fabsf(v011 - v010)), fmaxf(fabsf(v101 - v100), fabsf(v111 - v110)));
    g[0] = fmaxf(g[0], ex); g[1] = fmaxf(g[1], ey); g[2] = fmaxf(g[2], ez);
    g[3] = fmaxf(g[3], sqrtf(k0 * ex * k0 * ex + k1 * ey * k1 * ey + k2 * ez * k2 * ez));
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float m = g[a];
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && !(m != m)) atomicMax(&aux[a], __float_as_uint(m * (a == 3 ? 1.0001f : 1.f)));
  }
}

extern "C" size_t egx_sdf_coarse_bytes(int d0, int d1, int d2) {
  if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
  const int c0 = egx_ceil_div(d0, 4), c1 = egx_ceil_div(d1, 4), c2 = egx_ceil_div(d2, 4);
  return egx_sdf_bricks_offset(c0, c1, c2) + (size_t)c0 * c1 * c2 * 64 * sizeof(float);
}

extern "C" int egx_sdf_build_coarse(const egx_sdf_grid* sdf, void* coarse_out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && coarse_out && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad arguments");
  const int c0 = egx_ceil_div(sdf->d0, 4), c1 = egx_ceil_div(sdf->d1, 4), c2 = egx_ceil_div(sdf->d2, 4);
  const int n = (c0 + 2) * (c1 + 2) * (c2 + 2);
  hipLaunchKernelGGL(egx_sdf_build_coarse_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     sdf->grid, sdf->d0, sdf->d1, sdf->d2, c0, c1, c2, static_cast<float2*>(coarse_out));
  float* mips = reinterpret_cast<float*>(static_cast<char*>(coarse_out) + egx_sdf_table_bytes(c0, c1, c2));
  for (int l = 1; l <= EGX_SDF_MIP_LEVELS; ++l) {
    const int ne = egx_sdf_mip_dim(c0, l) * egx_sdf_mip_dim(c1, l) * egx_sdf_mip_dim(c2, l);
    hipLaunchKernelGGL(egx_sdf_build_mip_kernel, dim3(egx_ceil_div(ne, 128)), dim3(128), 0, static_cast<hipStream_t>(stream_),
                       static_cast<const float2*>(coarse_out), c0, c1, c2, l, mips + egx_sdf_mip_offset(c0, c1, c2, l));
  }
  hipLaunchKernelGGL(egx_sdf_build_bricks_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream_), sdf->grid, sdf->d0, sdf->d1, sdf->d2,
                     c0, c1, c2, reinterpret_cast<float*>(static_cast<char*>(coarse_out) + egx_sdf_bricks_offset(c0, c1, c2)));
  unsigned* aux = reinterpret_cast<unsigned*>(static_cast<char*>(coarse_out) + egx_sdf_aux_offset(c0, c1, c2));
  EGX_HIP_CHECK(hipMemsetAsync(aux, 0, EGX_SDF_AUX_FLOATS * sizeof(float), static_cast<hipStream_t>(stream_)));
  hipLaunchKernelGGL(egx_sdf_axis_steps_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream_), sdf->grid, sdf->d0, sdf->d1,
                     sdf->d2, sdf->scale * (float)sdf->d0 * 0.5f, sdf->scale * (float)sdf->d1 * 0.5f, sdf->scale * (float)sdf->d2 * 0.5f, aux);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_sdf_sample(const egx_sdf_grid* sdf, const float* pts, int64_t n, float* out, void* stream_) {
  EGX_REQUIRE(sdf && sdf->grid && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0, "bad sdf grid");
  EGX_REQUIRE(egx_sdf_dims_ok(sdf->d0, sdf->d1, sdf->d2), "sdf grid needs d2 >= 2 and fewer than 2^32 samples");
  if (n == 0) return EGX_OK;  // empty input is legal
  EGX_REQUIRE(pts && out && n > 0, "null points / output");
  SdfDev s{sdf->grid, sdf->d0, sdf->d1, sdf->d2, sdf->center[0], sdf->center[1], sdf->center[2], sdf->scale, nullptr, 0, 0, 0};
  const int64_t blocks = (n + 256 * EGX_SDF_PPT - 1) / (256 * EGX_SDF_PPT);
  const int grid = (int)(blocks < 8192 ? blocks : 8192);
  if (sdf->coarse_minmax) {   // the tables of egx_sdf_build_coarse are there: gather from the bricked copy
    s.c0 = egx_ceil_div(sdf->d0, 4); s.c1 = egx_ceil_div(sdf->d1, 4); s.c2 = egx_ceil_div(sdf->d2, 4);
    const float* bricks = reinterpret_cast<const float*>(static_cast<const char*>(sdf->coarse_minmax) + egx_sdf_bricks_offset(s.c0, s.c1, s.c2));
    hipLaunchKernelGGL(egx_sdf_sample_bricks_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), s, bricks, pts, n, out);
  } else {
    hipLaunchKernelGGL(egx_sdf_sample_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream_), s, pts, n, out);
  }
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Scene preparation: signed-distance grid of a closed triangle mesh in the storage convention of
// `sdf_dict` (crowd_ppo/utils.py:54-84): grid[d0][d1][d2] indexed by (x, y, z), sample (i, j, k) at the cell centre
// center + ((2 i + 1) / d - 1) / scale (grid_sample, align_corners=False).  The reference ships room0_sdf.pkl ready-made
// and has no generator in its tree (README.md:97); this is the tool for new scenes.
// One sample per lane; the triangles stream through LDS in chunks every lane reads in lock step (broadcast reads).
// Distance: exact closest point on each triangle (Voronoi-region walk).  Sign: crossing parity of the +z ray, each
// edge's crossing evaluated from its endpoints in canonical order, so the two triangles sharing an edge always agree on
// which of them a ray through that edge hits (watertight rule) - no epsilon, no jitter.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int MESH_CHUNK = 512;

__device__ __forceinline__ float tri_dist2(const float* t, float px, float py, float pz) {
  const float ax = t[0], ay = t[1], az = t[2];
  const float abx = t[3] - ax, aby = t[4] - ay, abz = t[5] - az;
  const float acx = t[6] - ax, acy = t[7] - ay, acz = t[8] - az;
  const float apx = px - ax, apy = py - ay, apz = pz - az;
  const float d1 = abx * apx + aby * apy + abz * apz, d2 = acx * apx + acy * apy + acz * apz;
  float cx, cy, cz;  // closest point - a
  if (d1 <= 0.f && d2 <= 0.f) {
    cx = cy = cz = 0.f;
  } else {
    const float bpx = apx - abx, bpy = apy - aby, bpz = apz - abz;
    const float d3 = abx * bpx + aby * bpy + abz * bpz, d4 = acx * bpx + acy * bpy + acz * bpz;
    const float cpx = apx - acx, cpy = apy - acy, cpz = apz - acz;
    const float d5 = abx * cpx + aby * cpy + abz * cpz, d6 = acx * cpx + acy * cpy + acz * cpz;
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d3 >= 0.f && d4 <= d3) {
      cx = abx; cy = aby; cz = abz;
    } else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
      const float v = d1 / (d1 - d3);
      cx = v * abx; cy = v * aby; cz = v * abz;
    } else if (d6 >= 0.f && d5 <= d6) {
      cx = acx; cy = acy; cz = acz;
    } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
      const float w = d2 / (d2 - d6);
      cx = w * acx; cy = w * acy; cz = w * acz;
    } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
      const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
      cx = abx + w * (acx - abx); cy = aby + w * (acy - aby); cz = abz + w * (acz - abz);
    } else {
      const float den = 1.f / (va + vb + vc);
      const float v = vb * den, w = vc * den;
      cx = abx * v + acx * w; cy = aby * v + acy * w; cz = abz * v + acz * w;
    }
  }
  const float dx = apx - cx, dy = apy - cy, dz = apz - cz;
  return dx * dx + dy * dy + dz * dz;
}

// does the projection of edge (p, q) onto the xy plane cross the +x half line from (x, y)?  canonical endpoint order
__device__ __forceinline__ bool edge_cross(float px, float py, float qx, float qy, float x, float y) {
  if (py > qy || (py == qy && px > qx)) { float t = px; px = qx; qx = t; t = py; py = qy; qy = t; }
  if ((py > y) == (qy > y)) return false;
  return x < px + (y - py) * (qx - px) / (qy - py);
}

__global__ __launch_bounds__(256) void egx_mesh_sdf_kernel(const float* __restrict__ tris, int F, float cx, float cy, float cz, float inv_scale,
                                                          int d0, int d1, int d2, int inside_positive, float* __restrict__ out) {
  __shared__ float s_t[MESH_CHUNK * 9];
  const size_t n = (size_t)d0 * d1 * d2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t id = idx < n ? idx : n - 1;
  const int k = (int)(id % d2), j = (int)((id / d2) % d1), i = (int)(id / ((size_t)d1 * d2));
  const float px = cx + ((2 * i + 1) / (float)d0 - 1.f) * inv_scale;
  const float py = cy + ((2 * j + 1) / (float)d1 - 1.f) * inv_scale;
  const float pz = cz + ((2 * k + 1) / (float)d2 - 1.f) * inv_scale;
  float best = 3.4e38f;
  int parity = 0;
  for (int f0 = 0; f0 < F; f0 += MESH_CHUNK) {
    const int nf = min(MESH_CHUNK, F - f0);
    __syncthreads();
    for (int e = threadIdx.x; e < nf * 9; e += 256) s_t[e] = tris[(size_t)f0 * 9 + e];
    __syncthreads();
    for (int f = 0; f < nf; ++f) {
      const float* t = s_t + f * 9;
      best = fminf(best, tri_dist2(t, px, py, pz));
      // the vertical line through (px, py) meets the triangle iff (px, py) is inside its projection (odd crossings)
      const bool in = edge_cross(t[0], t[1], t[3], t[4], px, py) ^ edge_cross(t[3], t[4], t[6], t[7], px, py) ^
                      edge_cross(t[6], t[7], t[0], t[1], px, py);
      if (in) {
        const float ux = t[3] - t[0], uy = t[4] - t[1], uz = t[5] - t[2];
        const float vx = t[6] - t[0], vy = t[7] - t[1], vz = t[8] - t[2];
        const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        if (nz != 0.f) {
          const float zt = t[2] - (nx * (px - t[0]) + ny * (py - t[1])) / nz;
          parity ^= (zt > pz) ? 1 : 0;
        }
      }
    }
  }
  if (idx < n) {
    const float d = sqrtf(best);
    out[idx] = ((parity != 0) == (inside_positive != 0)) ? d : -d;
  }
}
}  // namespace

extern "C" int egx_mesh_sdf(const float* triangles, int num_triangles, const float* center_host, float scale, int d0, int d1, int d2,
                            int inside_positive, float* out_grid, void* stream_) {
  EGX_REQUIRE(triangles && center_host && out_grid && num_triangles > 0, "null mesh / centre / grid");
  EGX_REQUIRE(scale > 0.f && egx_sdf_dims_ok(d0, d1, d2), "scale must be positive, the grid needs d2 >= 2 and fewer than 2^32 samples");
  const size_t n = (size_t)d0 * d1 * d2;
  hipLaunchKernelGGL(egx_mesh_sdf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     triangles, num_triangles, center_host[0], center_host[1], center_host[2], 1.f / scale, d0, d1, d2, inside_positive,
                     out_grid);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
