// Development aid: what does a dependent dense layer cost INSIDE one launch when only the workgroups that
// share a 32-row activation tile synchronise - against the same layer as its own kernel launch (the 4.9 us boundary)?
//
// Shape of the motion prior's decoder at 512 agents: 16 row tiles of 32 agents; a layer = [32 x K] x [K x N] per tile with
// K, N in {256, 512}; every fp32 operand as three bf16 planes in MFMA fragment order (6 bytes per element), six partial products
// per k (the arithmetic of csrc/dense3.hip).  One workgroup per CU (256 threads): G = 16 workgroups own one row tile - N / G
// output columns each - and sit on ONE XCD (workgroups are dealt to the XCDs round-robin: block b -> XCD b % 8), so the tile's
// activations are exchanged through that XCD's L2:
//    consume  the whole [32 x K] tile of the previous layer (48 KB at K = 256, 96 KB at K = 512), straight into MFMA registers
//    compute  four waves split K, LDS reduction, tanh, split into planes
//    publish  [32 x N/G] columns as planes (write-through stores), one release + one atomic per workgroup on the tile's counter
//    wait     until the tile's counter says all G slices of this layer are there (one lane polls, s_sleep)
// `chain` mode runs the same body as one launch per layer (no flags).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/tile_handoff.hip -o /tmp/tile_handoff && /tmp/tile_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 32, TILES = 16, G = 16;
struct Layer { int K, N; size_t w_off; };   // weights: [N / 16 col tiles][K / 32 k-steps][3 planes][64 lanes] bf16x8
struct Args {
  const bf16x8* W;
  bf16x8* act[2];            // [tile][2 row tiles of 16][k-step][3 planes][64 lanes], sized for 512 columns
  unsigned* counters;        // [TILES][n_layers]
  Layer layers[64];
  int n_layers, first, last; // layers [first, last) in this launch
  float* sink;
};

template <bool PERSISTENT>
__global__ __launch_bounds__(256) void chain_kernel(Args a) {
  __shared__ float red[4][ROWS][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;      // 32 workgroups per XCD
  const int tile = xcd * 2 + (idx >> 4), slot = idx & 15;     // 2 row tiles per XCD, G = 16 column slots each
  for (int l = a.first; l < a.last; ++l) {
    const Layer L = a.layers[l];
    const int S = L.K / 32, ncol = L.N / G;                     // k-steps; this workgroup's columns (16 or 32)
    const int NI = ncol / 16;
    const bf16x8* in = a.act[l & 1] + (size_t)tile * 2 * 16 * 3 * 64;
    bf16x8* out = a.act[(l + 1) & 1] + (size_t)tile * 2 * 16 * 3 * 64;
    if (PERSISTENT && l > a.first) {
      // wait for the G slices of layer l - 1 of THIS tile
      // (cdna_hip_programming.md publish / consume recipe: ONE lane polls ONE word relaxed, ONE acquire fence drops the CU's
      // stale L1 lines, then plain loads)
      if (threadIdx.x == 0) {
        const unsigned* c = a.counters + tile * a.n_layers + (l - 1);
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    f32x4 acc[2][2] = {};
    for (int s = wave; s < S; s += 4) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[mi][p] = in[((size_t)(mi * 16 + s) * 3 + p) * 64 + lane];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        if (ni < NI)
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[ni][p] = a.W[L.w_off + (((size_t)(slot * NI + ni) * S + s) * 3 + p) * 64 + lane];
      const int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            if (ni < NI) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi][pa[pr]], fb[ni][pb[pr]], acc[mi][ni], 0, 0, 0);
    }
    // reduce the four K quarters through LDS; lane = (col 0..15, row group 0..3), 4 rows each
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        if (ni < NI)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[wave][mi * 16 + (lane >> 4) * 4 + e][ni * 16 + (lane & 15)] = acc[mi][ni][e];
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS * ncol / 8; i += 256) {    // one thread = 8 consecutive columns of one row
      const int row = i / (ncol / 8), c8 = i % (ncol / 8);
      bf16x8 pl[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        float v = tanhf(red[0][row][c] + red[1][row][c] + red[2][row][c] + red[3][row][c]);
        for (int p = 0; p < 3; ++p) {
          const unsigned u = __float_as_uint(v) & 0xffff0000u;
          pl[p][e] = (short)(u >> 16);
          v -= __uint_as_float(u);
        }
      }
      const int col = slot * ncol + c8 * 8, ks = col / 32, half = (col % 32) / 8;   // fragment lane = row % 16 + 16 * (k / 8 % 4)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        bf16x8* dst = &out[((size_t)((row / 16) * 16 + ks) * 3 + p) * 64 + (row % 16) + 16 * half];
        if (PERSISTENT)   // write-through (sc1) payload: no release fence, hence no L2 write-back
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" :: "v"(dst), "v"(__builtin_bit_cast(f32x4, pl[p])) : "memory");
        else
          *dst = pl[p];
      }
    }
    if (PERSISTENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (PERSISTENT) {
      if (threadIdx.x == 0)
        __hip_atomic_fetch_add(a.counters + tile * a.n_layers + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.sink && threadIdx.x == 0 && blockIdx.x == 0) a.sink[0] = red[0][0][0];
}

int main() {
  // the decoder's per-frame layers: GRU cell (two 256-wide products -> modelled as one K = 512 layer to 256), MLP 256 -> 512, 512 -> 256
  std::vector<Layer> frame = {{512, 256, 0}, {256, 512, 0}, {512, 256, 0}};
  Args a{};
  size_t woff = 0;
  const int frames = 18;
  for (int f = 0; f < frames; ++f)
    for (int j = 0; j < 3; ++j) {
      Layer L = frame[j];
      L.w_off = (j == 0 ? 0 : (j == 1 ? (size_t)16 * 16 * 3 * 64 : (size_t)16 * 16 * 3 * 64 + (size_t)32 * 8 * 3 * 64));   // weights shared by all frames
      a.layers[a.n_layers++] = L;
    }
  woff = (size_t)16 * 16 * 3 * 64 + (size_t)32 * 8 * 3 * 64 + (size_t)16 * 16 * 3 * 64;
  bf16x8 *W, *act0, *act1;
  unsigned* cnt;
  float* sink;
  CHECK(hipMalloc(&W, woff * 16));
  CHECK(hipMemset(W, 0x11, woff * 16));
  const size_t act_frags = (size_t)TILES * 2 * 16 * 3 * 64;
  CHECK(hipMalloc(&act0, act_frags * 16)); CHECK(hipMalloc(&act1, act_frags * 16));
  CHECK(hipMemset(act0, 0x11, act_frags * 16)); CHECK(hipMemset(act1, 0x11, act_frags * 16));
  CHECK(hipMalloc(&cnt, TILES * 64 * 4)); CHECK(hipMalloc(&sink, 4));
  a.W = W; a.act[0] = act0; a.act[1] = act1; a.counters = cnt; a.sink = sink;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    // (1) one launch per layer
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it)
      for (int l = 0; l < a.n_layers; ++l) {
        a.first = l; a.last = l + 1;
        hipLaunchKernelGGL(chain_kernel<false>, dim3(256), dim3(256), 0, 0, a);
      }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
    // (2) one persistent launch with tile-local hand-offs
    float ms2 = 0;
    for (int it = 0; it < 10; ++it) {
      CHECK(hipMemsetAsync(cnt, 0, TILES * 64 * 4));
      a.first = 0; a.last = a.n_layers;
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain_kernel<true>, dim3(256), dim3(256), 0, 0, a);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float t; CHECK(hipEventElapsedTime(&t, e0, e1));
      ms2 += t;
    }
    printf("%d layers (18 frames x 3): one launch per layer %.2f us per layer | persistent, tile-local hand-off %.2f us per layer (%.1f us per chain)\n",
           a.n_layers, ms1 / 10 / a.n_layers * 1e3, ms2 / 10 / a.n_layers * 1e3, ms2 / 10 * 1e3);
  }
  return 0;
}
