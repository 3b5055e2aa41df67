// Development aid: bf16 MFMA waves fed through an LDS ring by OTHER waves of the workgroup (LDS-DMA + flags), while
// those loader waves also run VALU work - the structure of a loader/consumer LBS kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/mfma_feed.hip -o /tmp/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int R = 3, PIECES = 33, STEPS = 30;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NV>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ dirs, const bf16x8* __restrict__ feat, float* out, int* err) {
  __shared__ __attribute__((aligned(16))) bf16x8 ring[R][PIECES][64];
  __shared__ int landed[STEPS], consumed[STEPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = wave >> 2, w4 = wave & 3;
  if (threadIdx.x < STEPS) { landed[threadIdx.x] = 0; consumed[threadIdx.x] = 0; }
  __syncthreads();
  float acc_out = 0.f;
  if (role == 1) {  // loader + VALU
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i * 0.25f + lane;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(&ring[0][0][0]));
    for (int s = 0; s < STEPS; ++s) {
      int spin = 0;
      while (s >= R && __atomic_load_n(&consumed[s - R], __ATOMIC_RELAXED) < 4) {
        __builtin_amdgcn_s_sleep(1);
        if (++spin > (1 << 22)) { if (lane == 0) atomicAdd(err, 1); break; }
      }
      for (int p = w4; p < PIECES; p += 4) {
        const bf16x8* src = (p < 9) ? dirs + ((size_t)(s & 7) * 9 + p) * 64 + lane : feat + ((size_t)(s & 7) * 24 + (p - 9)) * 64 + lane;
        glds16(src, __builtin_amdgcn_readfirstlane(base + ((s % R) * PIECES + p) * 1024));
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j % 8] = fmaf(v[j % 8], 1.0001f, 0.5f);
      // the DMA of stage s stays in flight; everything older has landed once at most one stage's pieces are outstanding
      if (w4 == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (s > 0 && lane == 0) __atomic_fetch_add(&landed[s - 1], 1, __ATOMIC_RELEASE);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __atomic_fetch_add(&landed[STEPS - 1], 1, __ATOMIC_RELEASE);
    for (int i = 0; i < 8; ++i) acc_out += v[i];
  } else {  // MFMA consumer
    f32x16 acc[3][2];
    for (int c = 0; c < 3; ++c)
      for (int q = 0; q < 2; ++q)
        for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
    for (int s = 0; s < STEPS; ++s) {
      int spin = 0;
      while (__atomic_load_n(&landed[s], __ATOMIC_ACQUIRE) < 4) {
        __builtin_amdgcn_s_sleep(1);
        if (++spin > (1 << 22)) { if (lane == 0) atomicAdd(err, 1); break; }
      }
      const bf16x8(*st)[64] = ring[s % R];
      bf16x8 a[3][3], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[pl][c] = st[pl * 3 + c][lane];
#pragma unroll
        for (int q = 0; q < 2; ++q) b[pl][q] = st[9 + w4 * 6 + pl * 2 + q][lane];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __atomic_fetch_add(&consumed[s], 1, __ATOMIC_RELAXED);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][c], b[1][q], acc[c][q], 0, 0, 0);
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][c], b[2][q], acc[c][q], 0, 0, 0);
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][c], b[0][q], acc[c][q], 0, 0, 0);
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][c], b[1][q], acc[c][q], 0, 0, 0);
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][c], b[0][q], acc[c][q], 0, 0, 0);
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][c], b[0][q], acc[c][q], 0, 0, 0);
        }
    }
    for (int c = 0; c < 3; ++c)
      for (int q = 0; q < 2; ++q)
        for (int r = 0; r < 16; ++r) acc_out += acc[c][q][r];
  }
  if (acc_out == 123.456f) out[0] = acc_out;
}

template <int NV>
void run(const bf16x8* dirs, const bf16x8* feat, float* out, int* err, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 26;  // 6656 workgroups of 4 MFMA waves ~ the 13120 x 4 wave-items of the real launch / 2
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(grid), dim3(512), 0, 0, dirs, feat, out, err);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  const double fl = (double)grid * 4 * STEPS * 36 * 32768.0;
  printf("%-40s %.3f ms  %.0f TFLOP/s bf16 executed (%.0f f32-eq)  per MFMA-wave item %.0f nominal cycles  timeouts %d\n", name, best,
         fl / best / 1e9, fl / 6 / best / 1e9, best * 1e-3 * 2.3e9 / (grid / 256.0), herr);
}

int main() {
  const size_t nd = 8 * 9 * 64, nf = 8 * 24 * 64;
  std::vector<unsigned short> h((nd + nf) * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) % 512);
  bf16x8 *dirs, *feat; float* out; int* err;
  hipMalloc(&dirs, nd * 16); hipMalloc(&feat, nf * 16); hipMalloc(&out, 4); hipMalloc(&err, 4);
  hipMemset(err, 0, 4);
  hipMemcpy(dirs, h.data(), nd * 16, hipMemcpyHostToDevice);
  hipMemcpy(feat, h.data() + nd * 8, nf * 16, hipMemcpyHostToDevice);
  run<0>(dirs, feat, out, err, "loader waves: DMA only");
  run<64>(dirs, feat, out, err, "loader waves: DMA + 64 v_fma per step");
  run<160>(dirs, feat, out, err, "loader waves: DMA + 160 v_fma per step");
  run<320>(dirs, feat, out, err, "loader waves: DMA + 320 v_fma per step");
  return 0;
}
