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
};

__device__ __forceinline__ float egx_sdf_neg_trilinear(const SdfDev& s, float x, float y, float z) {
  const float nx = (x - s.cx) * s.scale, ny = (y - s.cy) * s.scale, nz = (z - s.cz) * s.scale;
  float px = ((nx + 1.f) * (float)s.d0 - 1.f) * 0.5f;
  float py = ((ny + 1.f) * (float)s.d1 - 1.f) * 0.5f;
  float pz = ((nz + 1.f) * (float)s.d2 - 1.f) * 0.5f;
  px = fminf(fmaxf(px, 0.f), (float)(s.d0 - 1));
  py = fminf(fmaxf(py, 0.f), (float)(s.d1 - 1));
  pz = fminf(fmaxf(pz, 0.f), (float)(s.d2 - 1));
  const float x0 = floorf(px), y0 = floorf(py), z0 = floorf(pz);
  const float wx1 = px - x0, wy1 = py - y0, wz1 = pz - z0;
  const float wx0 = (x0 + 1.f) - px, wy0 = (y0 + 1.f) - py, wz0 = (z0 + 1.f) - pz;
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
