// Development aid: how much of the v_mfma_f32_32x32x2_f32 ceiling survives when the operand stream of
// egx_lbs_fused_kernel (5 x global_load_dwordx4 per 24 MFMAs, L2-hot data) is added.
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/mfma_loads.hip -o /tmp/mfma_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: no loads; 1: loads issued, results only consumed at the end; 2: loads feed the MFMAs (ring of 3 slots)
template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const f32x4* __restrict__ dirs, const f32x4* __restrict__ feat, float* out, int groups) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[3][2];
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
  const f32x4* dp = dirs + lane;
  const f32x4* fp0 = feat + lane;
  const f32x4* fp1 = feat + 59 * 64 + lane;
  f32x4 a_st[3][3], b_st[3][2];
  for (int sl = 0; sl < 3; ++sl) {
    for (int c = 0; c < 3; ++c) a_st[sl][c] = dp[(sl * 3 + c) * 64];
    b_st[sl][0] = fp0[sl * 64];
    b_st[sl][1] = fp1[sl * 64];
  }
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#define KG(G, U)                                                                                               \
  {                                                                                                            \
    const int gn_ = ((G) + 2 < groups) ? (G) + 2 : groups - 1;                                                 \
    if (MODE == 2) {                                                                                           \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) a_st[((U) + 2) % 3][c] = dp[(gn_ * 3 + c) * 64];          \
      b_st[((U) + 2) % 3][0] = fp0[gn_ * 64];                                                                  \
      b_st[((U) + 2) % 3][1] = fp1[gn_ * 64];                                                                  \
    } else if (MODE == 3) { /* as 2, but everything outstanding is waited for before the MFMAs start */      \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) a_st[((U) + 2) % 3][c] = dp[(gn_ * 3 + c) * 64];          \
      b_st[((U) + 2) % 3][0] = fp0[gn_ * 64];                                                                  \
      b_st[((U) + 2) % 3][1] = fp1[gn_ * 64];                                                                  \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    } else if (MODE == 4) { /* as 2, but the loads are issued AFTER the group's MFMAs */                      \
    } else if (MODE == 1) {                                                                                    \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) sink += dp[(gn_ * 3 + c) * 64];                           \
      sink += fp0[gn_ * 64];                                                                                   \
      sink += fp1[gn_ * 64];                                                                                   \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                              \
      _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                            \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                          \
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_st[(U)][c][e], b_st[(U)][q][e], acc[c][q], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (MODE == 4) {                                                                                           \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) a_st[(U)][c] = dp[(((G) + 3 < groups ? (G) + 3 : groups - 1) * 3 + c) * 64]; \
      b_st[(U)][0] = fp0[((G) + 3 < groups ? (G) + 3 : groups - 1) * 64];                                      \
      b_st[(U)][1] = fp1[((G) + 3 < groups ? (G) + 3 : groups - 1) * 64];                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  }
  for (int g0 = 0; g0 + 2 < groups; g0 += 3) {
    KG(g0, 0)
    KG(g0 + 1, 1)
    KG(g0 + 2, 2)
  }
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) s += acc[c][q][r];
  if (s == 123.456f) out[0] = s;
}

template <int MODE, int OCC>
void run(const f32x4* dirs, const f32x4* feat, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int groups = 57, grid = 13120;
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, dirs, feat, out, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s occ=%d  %.3f ms  %.1f TFLOP/s\n", name, OCC, best, (double)grid * 4 * groups * 24 * 4096.0 / best / 1e9);
}

int main() {
  const size_t nd = 59 * 3 * 64, nf = 2 * 59 * 64;
  std::vector<float> h((nd + nf) * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
  f32x4 *dirs, *feat; float* out;
  hipMalloc(&dirs, nd * 16); hipMalloc(&feat, nf * 16); hipMalloc(&out, 4);
  hipMemcpy(dirs, h.data(), nd * 16, hipMemcpyHostToDevice);
  hipMemcpy(feat, h.data() + nd * 4, nf * 16, hipMemcpyHostToDevice);
  run<0, 2>(dirs, feat, out, "no loads, random operands in regs");
  run<1, 2>(dirs, feat, out, "loads issued, not feeding MFMA");
  run<2, 2>(dirs, feat, out, "loads feed MFMA (ring of 3)");
  run<3, 2>(dirs, feat, out, "ring of 3, vmcnt(0) before MFMAs");
  run<4, 2>(dirs, feat, out, "ring of 3, loads after the MFMAs");
  run<0, 1>(dirs, feat, out, "no loads, random operands in regs");
  run<2, 1>(dirs, feat, out, "loads feed MFMA (ring of 3)");
  hipMemset(dirs, 0, nd * 16); hipMemset(feat, 0, nf * 16);
  run<0, 2>(dirs, feat, out, "ZERO data: no loads");
  run<2, 2>(dirs, feat, out, "ZERO data: loads feed MFMA");
  const float one = 1.0f;
  std::vector<float> ones((nd + nf) * 4, one);
  hipMemcpy(dirs, ones.data(), nd * 16, hipMemcpyHostToDevice);
  hipMemcpy(feat, ones.data(), nf * 16, hipMemcpyHostToDevice);
  run<0, 2>(dirs, feat, out, "ONES data: no loads");
  run<2, 2>(dirs, feat, out, "ONES data: loads feed MFMA");
  return 0;
}
