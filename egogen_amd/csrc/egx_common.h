// Shared helpers for libegogen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/egogen_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ inline unsigned short egx_bf16_rne(float x) {
  union { float f; unsigned u; } c;
  c.f = x;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((c.u >> 16) | 0x40u);
  return (unsigned short)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float egx_bf16_to_f32(unsigned short h) {
  union { float f; unsigned u; } c;
  c.u = (unsigned)h << 16;
  return c.f;
}


void egx_set_error(const std::string& msg);

#define EGX_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      egx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
      return EGX_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define EGX_REQUIRE(cond, msg)                                                                \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      egx_set_error(std::string("argument check failed: ") + #cond + " - " + (msg));          \
      return EGX_ERR_ARG;                                                                     \
    }                                                                                         \
  } while (0)

static inline int egx_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t egx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Trilinear SDF lookup with torch.grid_sample semantics (align_corners=False, padding 'border'),
// NEGATED as crowd_ppo/utils.py:83-84 does.  Corner order / weight products follow
// aten GridSamplerKernel so that fp32 results match the CPU path to the last bits.
struct SdfDev {
  const float* grid;
  int d0, d1, d2;
  float cx, cy, cz, scale;
  const float2* coarse;  // optional padded bracket table [(c0+2)][(c1+2)][(c2+2)] {min,max} (egx_sdf_build_coarse)
  int c0, c1, c2;
};

// Free-space pyramid behind the bracket table (same buffer, after the float2 entries, 256-byte aligned): level l = 1..4 holds,
// per block of 2^l x 2^l x 2^l PADDED bracket cells, the maximum of their `max` entries.  A box of padded cells whose pyramid
// entries are all < 0 contains only points whose interpolated value is < 0 (convexity of the interpolation), i.e. free space
// under the count's sign convention (a vertex counts when -trilinear < 0).  Used by the LBS work-item culling (body_model.hip).
constexpr int EGX_SDF_MIP_LEVELS = 4;
__host__ __device__ inline int egx_sdf_mip_dim(int c, int l) { return ((c + 2) + (1 << l) - 1) >> l; }
__host__ __device__ inline size_t egx_sdf_mip_offset(int c0, int c1, int c2, int l) {   // in floats, from the start of the pyramid
  size_t off = 0;
  for (int k = 1; k < l; ++k) off += (size_t)egx_sdf_mip_dim(c0, k) * egx_sdf_mip_dim(c1, k) * egx_sdf_mip_dim(c2, k);
  return off;
}
__host__ __device__ inline size_t egx_sdf_table_bytes(int c0, int c1, int c2) {       // bracket table, padded to 256 bytes
  return ((size_t)(c0 + 2) * (c1 + 2) * (c2 + 2) * 8 + 255) / 256 * 256;
}

// After the pyramid (256-byte aligned): EGX_SDF_AUX_FLOATS floats written by egx_sdf_build_coarse -
//   [0..2] the largest |difference| of two neighbouring samples along x, y, z
//   [3]    the steepest slope of the interpolated field, value per metre of world space (sdf.hip: egx_sdf_axis_steps_kernel):
//          a position error bound times this is a bound on the change of the sampled value (fix-up of the mixed blend).
constexpr int EGX_SDF_AUX_FLOATS = 64;
__host__ __device__ inline size_t egx_sdf_aux_offset(int c0, int c1, int c2) {       // in bytes, from the start of the table
  return (egx_sdf_table_bytes(c0, c1, c2) + egx_sdf_mip_offset(c0, c1, c2, EGX_SDF_MIP_LEVELS + 1) * sizeof(float) + 255) / 256 * 256;
}

// After the aux floats (256-byte aligned): the grid again as 4 x 4 x 4 BRICKS of 256 contiguous bytes ([bx][by][bz][x & 3][y & 3][z & 3],
// samples past the grid's end repeat the last plane) for the standalone calc_sdf kernel: the eight corners of a point then sit
// in 2.3 cache lines on average instead of four (sdf.hip: egx_sdf_sample_bricks_kernel).
__host__ __device__ inline size_t egx_sdf_bricks_offset(int c0, int c1, int c2) {
  return (egx_sdf_aux_offset(c0, c1, c2) + EGX_SDF_AUX_FLOATS * sizeof(float) + 255) / 256 * 256;
}

inline bool egx_sdf_dims_ok(int d0, int d1, int d2) {
  return d2 >= 2 && (unsigned long long)d0 * (unsigned long long)d1 * (unsigned long long)d2 < (1ull << 32);
}

// One rounding per operation, whatever the surrounding code: hip-clang's __fmul_rn / __fadd_rn are plain `*` / `+` compiled with
// the default -ffp-contract=fast, i.e. the compiler fuses them into FMAs differently from kernel to kernel (a last-bit
// difference between two kernels that inline the same function).  These are compiled with contraction off.
__device__ __forceinline__ float egx_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float egx_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float egx_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// Continuous voxel coordinates of a world point, clamped to the grid (align_corners=False, padding "border").
// every operation rounds on its own (contraction off below), so the coordinates round exactly like the CPU path
__device__ __forceinline__ void egx_sdf_voxel_coords(const SdfDev& s, float x, float y, float z, float& px, float& py, float& pz) {
#pragma clang fp contract(off)   // __fmul_rn / __fadd_rn are plain * and + under hip-clang: without this the compiler fuses them
  const float nx = egx_mul(egx_sub(x, s.cx), s.scale), ny = egx_mul(egx_sub(y, s.cy), s.scale),
              nz = egx_mul(egx_sub(z, s.cz), s.scale);
  px = egx_mul(egx_sub(egx_mul(egx_add(nx, 1.f), (float)s.d0), 1.f), 0.5f);
  py = egx_mul(egx_sub(egx_mul(egx_add(ny, 1.f), (float)s.d1), 1.f), 0.5f);
  pz = egx_mul(egx_sub(egx_mul(egx_add(nz, 1.f), (float)s.d2), 1.f), 0.5f);
  px = fminf(fmaxf(px, 0.f), (float)(s.d0 - 1));
  py = fminf(fmaxf(py, 0.f), (float)(s.d1 - 1));
  pz = fminf(fmaxf(pz, 0.f), (float)(s.d2 - 1));
}

// {min,max} bracket of the samples the interpolation can touch for a point with UNCLAMPED voxel coordinates (rx,ry,rz):
// entry (jx,jy,jz) of the padded table built by egx_sdf_build_coarse (sdf.hip), j = clamp(floor(r / 4) + 1, 0, c + 1).
// A coordinate within round-off of a cell border may land in the neighbouring cell; both cells' footprints contain the
// samples such a point touches (block footprints include their border planes), so the bracket stays valid.
__device__ __forceinline__ float2 egx_sdf_coarse_at_raw(const SdfDev& s, float rx, float ry, float rz) {
  const float jx = __builtin_amdgcn_fmed3f(floorf(fmaf(rx, 0.25f, 1.f)), 0.f, (float)(s.c0 + 1));
  const float jy = __builtin_amdgcn_fmed3f(floorf(fmaf(ry, 0.25f, 1.f)), 0.f, (float)(s.c1 + 1));
  const float jz = __builtin_amdgcn_fmed3f(floorf(fmaf(rz, 0.25f, 1.f)), 0.f, (float)(s.c2 + 1));
  const unsigned idx = ((unsigned)jx * (unsigned)(s.c1 + 2) + (unsigned)jy) * (unsigned)(s.c2 + 2) + (unsigned)jz;
  return s.coarse[idx];
}

// The same lookup from CELL coordinates c = r / 4 + 1 (the caller folds the 1/4 and the +1 into its affine map).  The table
// index is formed in fp32 - exact while the padded table has fewer than 2^24 entries (a 256^3 grid: 66^3) - and converted
// once: two FMAs and one conversion instead of three conversions, two integer multiplies (quarter rate) and two adds.
__device__ __forceinline__ float2 egx_sdf_coarse_at_cell(const SdfDev& s, float cx, float cy, float cz) {
  const float jx = __builtin_amdgcn_fmed3f(floorf(cx), 0.f, (float)(s.c0 + 1));
  const float jy = __builtin_amdgcn_fmed3f(floorf(cy), 0.f, (float)(s.c1 + 1));
  const float jz = __builtin_amdgcn_fmed3f(floorf(cz), 0.f, (float)(s.c2 + 1));
  unsigned idx;
  if ((unsigned)(s.c0 + 2) * (unsigned)(s.c1 + 2) * (unsigned)(s.c2 + 2) < (1u << 24))   // uniform
    idx = (unsigned)fmaf(fmaf(jx, (float)(s.c1 + 2), jy), (float)(s.c2 + 2), jz);
  else
    idx = ((unsigned)jx * (unsigned)(s.c1 + 2) + (unsigned)jy) * (unsigned)(s.c2 + 2) + (unsigned)jz;
  return s.coarse[idx];
}

struct __attribute__((packed, aligned(4))) EgxF2 { float x, y; };  // two z-neighbours, 4-byte aligned

// -trilinear(grid) at clamped voxel coordinates: aten grid_sampler_3d corner order and rounding.  The two z-neighbours
// of each (x,y) corner column are one 8-byte load; corners that fall off the grid have weight exactly 0 (aten drops
// them) and a clamped index.  Requires d2 >= 2 and d0*d1*d2 < 2^32.
__device__ __forceinline__ float egx_sdf_neg_trilinear_at(const SdfDev& s, float px, float py, float pz) {
#pragma clang fp contract(off)   // products and sums round one by one as in aten's kernel, in every kernel this is inlined into
  const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
  const float wx1 = egx_sub(px, x0), wy1 = egx_sub(py, y0), wz1 = egx_sub(pz, z0);
  const float wx0 = egx_sub(egx_add(x0, 1.f), px), wy0 = egx_sub(egx_add(y0, 1.f), py), wz0 = egx_sub(egx_add(z0, 1.f), pz);
  const unsigned ix0 = (unsigned)x0, iy0 = (unsigned)y0, iz0 = (unsigned)z0;
  const bool x1_in = ix0 + 1 < (unsigned)s.d0, y1_in = iy0 + 1 < (unsigned)s.d1, z1_in = iz0 + 1 < (unsigned)s.d2;
  const unsigned zb = z1_in ? iz0 : iz0 - 1;                      // pair (zb, zb+1) always inside the row
  const unsigned i00 = (ix0 * (unsigned)s.d1 + iy0) * (unsigned)s.d2 + zb;
  const unsigned sy = y1_in ? (unsigned)s.d2 : 0u, sx = x1_in ? (unsigned)s.d1 * (unsigned)s.d2 : 0u;
  const EgxF2* g = reinterpret_cast<const EgxF2*>(s.grid);
  const EgxF2 c00 = *reinterpret_cast<const EgxF2*>(s.grid + i00), c01 = *reinterpret_cast<const EgxF2*>(s.grid + (i00 + sy)),
              c10 = *reinterpret_cast<const EgxF2*>(s.grid + (i00 + sx)), c11 = *reinterpret_cast<const EgxF2*>(s.grid + (i00 + sx + sy));
  (void)g;
  const float wx1e = x1_in ? wx1 : 0.f, wy1e = y1_in ? wy1 : 0.f, wz1e = z1_in ? wz1 : 0.f;
  const float v000 = z1_in ? c00.x : c00.y, v010 = z1_in ? c01.x : c01.y, v100 = z1_in ? c10.x : c10.y, v110 = z1_in ? c11.x : c11.y;
  const float a00 = egx_mul(wx0, wy0), a01 = egx_mul(wx0, wy1e), a10 = egx_mul(wx1e, wy0), a11 = egx_mul(wx1e, wy1e);
  float acc;
  acc = egx_mul(v000, egx_mul(a00, wz0));
  acc = egx_add(acc, egx_mul(c00.y, egx_mul(a00, wz1e)));
  acc = egx_add(acc, egx_mul(v010, egx_mul(a01, wz0)));
  acc = egx_add(acc, egx_mul(c01.y, egx_mul(a01, wz1e)));
  acc = egx_add(acc, egx_mul(v100, egx_mul(a10, wz0)));
  acc = egx_add(acc, egx_mul(c10.y, egx_mul(a10, wz1e)));
  acc = egx_add(acc, egx_mul(v110, egx_mul(a11, wz0)));
  acc = egx_add(acc, egx_mul(c11.y, egx_mul(a11, wz1e)));
  return -acc;
}

__device__ __forceinline__ float egx_sdf_neg_trilinear(const SdfDev& s, float x, float y, float z) {
  float px, py, pz;
  egx_sdf_voxel_coords(s, x, y, z, px, py, pz);
  return egx_sdf_neg_trilinear_at(s, px, py, pz);
}

// ---- rotation helpers shared with the env kernels (torchgeometry 0.1.2 semantics) ----
__device__ __forceinline__ void egx_tgm_rotmat_to_aa(const float* R /*row-major 3x3*/, float* aa) {
  // rotation_matrix_to_quaternion works on R^T: m[a][b] = R[b][a]
  const float m00 = R[0], m01 = R[3], m02 = R[6], m10 = R[1], m11 = R[4], m12 = R[7], m20 = R[2], m21 = R[5], m22 = R[8];
  float q0, q1, q2, q3, t;
  if (m22 < 1e-6f) {
    if (m00 > m11) {
      t = 1.f + m00 - m11 - m22;
      q0 = m12 - m21; q1 = t; q2 = m01 + m10; q3 = m20 + m02;
    } else {
      t = 1.f - m00 + m11 - m22;
      q0 = m20 - m02; q1 = m01 + m10; q2 = t; q3 = m12 + m21;
    }
  } else {
    if (m00 < -m11) {
      t = 1.f - m00 - m11 + m22;
      q0 = m01 - m10; q1 = m20 + m02; q2 = m12 + m21; q3 = t;
    } else {
      t = 1.f + m00 + m11 + m22;
      q0 = t; q1 = m12 - m21; q2 = m20 - m02; q3 = m01 - m10;
    }
  }
  const float sc = 0.5f / sqrtf(t);
  q0 *= sc; q1 *= sc; q2 *= sc; q3 *= sc;
  const float sin_sq = q1 * q1 + q2 * q2 + q3 * q3;
  const float sin_t = sqrtf(sin_sq);
  const float two_theta = 2.f * ((q0 < 0.f) ? atan2f(-sin_t, -q0) : atan2f(sin_t, q0));
  const float k = (sin_sq > 0.f) ? two_theta / sin_t : 2.f;
  aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
}

__device__ __forceinline__ void egx_tgm_aa_to_rotmat(const float* aa, float* R) {
  const float theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > 1e-6f) {
    const float theta = sqrtf(theta2);
    const float wx = aa[0] / (theta + 1e-6f), wy = aa[1] / (theta + 1e-6f), wz = aa[2] / (theta + 1e-6f);
    const float c = cosf(theta), s = sinf(theta), oc = 1.f - c;
    R[0] = c + wx * wx * oc;      R[1] = wx * wy * oc - wz * s; R[2] = wy * s + wx * wz * oc;
    R[3] = wz * s + wx * wy * oc; R[4] = c + wy * wy * oc;      R[5] = -wx * s + wy * wz * oc;
    R[6] = -wy * s + wx * wz * oc; R[7] = wx * s + wy * wz * oc; R[8] = c + wz * wz * oc;
  } else {
    R[0] = 1.f; R[1] = -aa[2]; R[2] = aa[1];
    R[3] = aa[2]; R[4] = 1.f; R[5] = -aa[0];
    R[6] = -aa[1]; R[7] = aa[0]; R[8] = 1.f;
  }
}

