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

// operands read from LDS with ds_read_b128 (LDS pre-filled once); WAIT: 0 = reads of the next group in flight during
// the MFMAs (ring of 3), 1 = lgkmcnt(0) before the MFMAs
template <int WAIT, int OCC>
__global__ __launch_bounds__(256, OCC) void klds(const f32x4* __restrict__ dirs, const f32x4* __restrict__ feat, float* out, int groups) {
  __shared__ f32x4 sd[8 * 3 * 64];      // 8 k-groups of bases
  __shared__ f32x4 sf[4][2][8 * 64];    // per wave: 2 body tiles x 8 k-groups of features
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 3 * 64; i += 256) sd[i] = dirs[i];
  for (int i = lane; i < 8 * 64; i += 64) { sf[wave][0][i] = feat[i]; sf[wave][1][i] = feat[59 * 64 + i]; }
  __syncthreads();
  f32x16 acc[3][2];
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
  f32x4 a_st[3][3], b_st[3][2];
  for (int sl = 0; sl < 3; ++sl) {
    for (int c = 0; c < 3; ++c) a_st[sl][c] = sd[(sl * 3 + c) * 64 + lane];
    b_st[sl][0] = sf[wave][0][sl * 64 + lane];
    b_st[sl][1] = sf[wave][1][sl * 64 + lane];
  }
#define KL(G, U)                                                                                               \
  {                                                                                                            \
    const int gn_ = ((G) + 2) & 7;                                                                             \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) a_st[((U) + 2) % 3][c] = sd[(gn_ * 3 + c) * 64 + lane];      \
    b_st[((U) + 2) % 3][0] = sf[wave][0][gn_ * 64 + lane];                                                     \
    b_st[((U) + 2) % 3][1] = sf[wave][1][gn_ * 64 + lane];                                                     \
    if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                              \
      _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                            \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                          \
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_st[(U)][c][e], b_st[(U)][q][e], acc[c][q], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
  for (int g0 = 0; g0 + 2 < groups; g0 += 3) {
    KL(g0, 0)
    KL(g0 + 1, 1)
    KL(g0 + 2, 2)
  }
  float s = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) s += acc[c][q][r];
  if (s == 123.456f) out[0] = s;
}

// operands DMA'd global -> LDS (global_load_lds_dwordx4, wave-private 2-slot ring), read back with ds_read_b128:
// the DMA of group g+1 is in flight while the MFMAs of group g execute
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int OCC>
__global__ __launch_bounds__(256, OCC) void kdma(const f32x4* __restrict__ dirs, const f32x4* __restrict__ feat, float* out, int groups) {
  __shared__ f32x4 ring[4][2][5 * 64];  // [wave][slot][3 bases + 2 feature tiles][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[3][2];
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
  const f32x4* dp = dirs + lane;
  const f32x4* fp0 = feat + lane;
  const f32x4* fp1 = feat + 59 * 64 + lane;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(&ring[wave][0][0]));
  auto issue = [&](int g, int slot) {
    const unsigned d = base + slot * (5 * 64 * 16);
    glds16(dp + (g * 3 + 0) * 64, d);
    glds16(dp + (g * 3 + 1) * 64, d + 1024);
    glds16(dp + (g * 3 + 2) * 64, d + 2048);
    glds16(fp0 + g * 64, d + 3072);
    glds16(fp1 + g * 64, d + 4096);
  };
  issue(0, 0);
  for (int g = 0; g < groups; ++g) {
    const int slot = g & 1;
    issue((g + 1 < groups) ? g + 1 : g, slot ^ 1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    f32x4 a[3], b[2];
    const f32x4* sl = &ring[wave][slot][0];
    for (int c = 0; c < 3; ++c) a[c] = sl[c * 64 + lane];
    b[0] = sl[3 * 64 + lane];
    b[1] = sl[4 * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[q][e], acc[c][q], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 2; ++q)
      for (int r = 0; r < 16; ++r) s += acc[c][q][r];
  if (s == 123.456f) out[0] = s;
}
template <int OCC>
void rundma(const f32x4* dirs, const f32x4* feat, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int groups = 57, grid = 13120;
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kdma<OCC>), dim3(grid), dim3(256), 0, 0, dirs, feat, out, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s occ=%d  %.3f ms  %.1f TFLOP/s\n", name, OCC, best, (double)grid * 4 * groups * 24 * 4096.0 / best / 1e9);
}

template <int WAIT, int OCC>
void runlds(const f32x4* dirs, const f32x4* feat, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int groups = 57, grid = 13120;
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((klds<WAIT, OCC>), dim3(grid), dim3(256), 0, 0, dirs, feat, out, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s occ=%d  %.3f ms  %.1f TFLOP/s\n", name, OCC, best, (double)grid * 4 * groups * 24 * 4096.0 / best / 1e9);
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
  runlds<0, 2>(dirs, feat, out, "LDS operands, reads in flight");
  runlds<1, 2>(dirs, feat, out, "LDS operands, lgkmcnt(0) first");
  rundma<2>(dirs, feat, out, "LDS-DMA ring, DMA in flight");
  rundma<1>(dirs, feat, out, "LDS-DMA ring, DMA in flight");
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
