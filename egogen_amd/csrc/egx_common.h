// Shared helpers for libegogen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/egogen_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  const float2* coarse;  // optional [c0][c1][c2] {min,max} of the fine values each 4^3 block's samples can touch
  int c0, c1, c2;
};

__device__ __forceinline__ float2 egx_sdf_coarse_fetch(const SdfDev& s, float x, float y, float z) {
  const float nx = __fmul_rn(__fsub_rn(x, s.cx), s.scale), ny = __fmul_rn(__fsub_rn(y, s.cy), s.scale),
              nz = __fmul_rn(__fsub_rn(z, s.cz), s.scale);
  float px = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.f), (float)s.d0), 1.f), 0.5f);
  float py = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.f), (float)s.d1), 1.f), 0.5f);
  float pz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nz, 1.f), (float)s.d2), 1.f), 0.5f);
  px = fminf(fmaxf(px, 0.f), (float)(s.d0 - 1));
  py = fminf(fmaxf(py, 0.f), (float)(s.d1 - 1));
  pz = fminf(fmaxf(pz, 0.f), (float)(s.d2 - 1));
  // clamped to [0, d-1]: truncation is floor; 32-bit index (the table has at most 2^31 / 8 entries) keeps the address
  // arithmetic off the quarter-rate 64-bit VALU paths
  const unsigned ix = (unsigned)px >> 2, iy = (unsigned)py >> 2, iz = (unsigned)pz >> 2;
  const unsigned idx = (ix * (unsigned)s.c1 + iy) * (unsigned)s.c2 + iz;
  return s.coarse[idx];
}

// Sign of calc_sdf at a point without touching the fine grid when the coarse {min,max} brackets decide it:
// trilinear interpolation is a convex combination of the 8 corners, so if every value the footprint can touch is
// negative (free space) the result is negative and -result > 0: not penetrating; if all are positive: penetrating.
// Returns -1 (not penetrating), +1 (penetrating) or 0 (mixed: evaluate the fine grid).  Exact, not an approximation.
__device__ __forceinline__ int egx_sdf_coarse_sign(const SdfDev& s, float x, float y, float z) {
  const float nx = __fmul_rn(__fsub_rn(x, s.cx), s.scale), ny = __fmul_rn(__fsub_rn(y, s.cy), s.scale),
              nz = __fmul_rn(__fsub_rn(z, s.cz), s.scale);
  float px = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.f), (float)s.d0), 1.f), 0.5f);
  float py = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.f), (float)s.d1), 1.f), 0.5f);
  float pz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nz, 1.f), (float)s.d2), 1.f), 0.5f);
  px = fminf(fmaxf(px, 0.f), (float)(s.d0 - 1));
  py = fminf(fmaxf(py, 0.f), (float)(s.d1 - 1));
  pz = fminf(fmaxf(pz, 0.f), (float)(s.d2 - 1));
  const int ix = (int)floorf(px) >> 2, iy = (int)floorf(py) >> 2, iz = (int)floorf(pz) >> 2;
  const float2 mm = s.coarse[((size_t)ix * s.c1 + iy) * s.c2 + iz];
  return (mm.y < 0.f) ? -1 : ((mm.x > 0.f) ? 1 : 0);
}

__device__ __forceinline__ float egx_sdf_neg_trilinear(const SdfDev& s, float x, float y, float z) {
  // explicit _rn intrinsics: immune to fma contraction, so the coordinates round exactly like the CPU path
  const float nx = __fmul_rn(__fsub_rn(x, s.cx), s.scale), ny = __fmul_rn(__fsub_rn(y, s.cy), s.scale),
              nz = __fmul_rn(__fsub_rn(z, s.cz), s.scale);
  float px = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.f), (float)s.d0), 1.f), 0.5f);
  float py = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.f), (float)s.d1), 1.f), 0.5f);
  float pz = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(nz, 1.f), (float)s.d2), 1.f), 0.5f);
  px = fminf(fmaxf(px, 0.f), (float)(s.d0 - 1));
  py = fminf(fmaxf(py, 0.f), (float)(s.d1 - 1));
  pz = fminf(fmaxf(pz, 0.f), (float)(s.d2 - 1));
  const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
  const float wx1 = __fsub_rn(px, x0), wy1 = __fsub_rn(py, y0), wz1 = __fsub_rn(pz, z0);
  const float wx0 = __fsub_rn(__fadd_rn(x0, 1.f), px), wy0 = __fsub_rn(__fadd_rn(y0, 1.f), py), wz0 = __fsub_rn(__fadd_rn(z0, 1.f), pz);
  const int ix0 = (int)x0, iy0 = (int)y0, iz0 = (int)z0;
  const int ix1 = min(ix0 + 1, s.d0 - 1), iy1 = min(iy0 + 1, s.d1 - 1);
  const bool z1_in = (iz0 + 1) < s.d2;
  const float* r00 = s.grid + ((size_t)ix0 * s.d1 + iy0) * s.d2 + iz0;
  const float* r01 = s.grid + ((size_t)ix0 * s.d1 + iy1) * s.d2 + iz0;
  const float* r10 = s.grid + ((size_t)ix1 * s.d1 + iy0) * s.d2 + iz0;
  const float* r11 = s.grid + ((size_t)ix1 * s.d1 + iy1) * s.d2 + iz0;
  const int dz = z1_in ? 1 : 0;  // weight is exactly 0 when the +1 corner falls off the grid
  // out-of-range corners contribute 0 in aten; here their weight is exactly 0 and the index is clamped
  const float wx1e = (ix0 + 1 < s.d0) ? wx1 : 0.f, wy1e = (iy0 + 1 < s.d1) ? wy1 : 0.f, wz1e = z1_in ? wz1 : 0.f;
  float acc;
  acc = __fmul_rn(r00[0], __fmul_rn(__fmul_rn(wx0, wy0), wz0));
  acc = __fadd_rn(acc, __fmul_rn(r00[dz], __fmul_rn(__fmul_rn(wx0, wy0), wz1e)));
  acc = __fadd_rn(acc, __fmul_rn(r01[0], __fmul_rn(__fmul_rn(wx0, wy1e), wz0)));
  acc = __fadd_rn(acc, __fmul_rn(r01[dz], __fmul_rn(__fmul_rn(wx0, wy1e), wz1e)));
  acc = __fadd_rn(acc, __fmul_rn(r10[0], __fmul_rn(__fmul_rn(wx1e, wy0), wz0)));
  acc = __fadd_rn(acc, __fmul_rn(r10[dz], __fmul_rn(__fmul_rn(wx1e, wy0), wz1e)));
  acc = __fadd_rn(acc, __fmul_rn(r11[0], __fmul_rn(__fmul_rn(wx1e, wy1e), wz0)));
  acc = __fadd_rn(acc, __fmul_rn(r11[dz], __fmul_rn(__fmul_rn(wx1e, wy1e), wz1e)));
  return -acc;
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

