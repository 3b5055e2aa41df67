// Development aid: cost of a grid-wide barrier (all co-resident workgroups, device-scope atomics) on the 8-XCD MI355X,
// and of a barrier among the workgroups of ONE XCD (block id % 8).  Decides whether a persistent cooperative decoder
// (4 barriers per step) could beat 4 kernel launches per step.
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/grid_barrier.hip -o /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned* counters, int rounds, int per_xcd, unsigned* err) {
  // sense-free monotonically increasing barrier: round r waits for count >= (r+1) * participants
  const int group = per_xcd ? (blockIdx.x & 7) : 0;
  const unsigned participants = per_xcd ? gridDim.x / 8 : gridDim.x;
  unsigned* ctr = counters + group * 32;  // separate cache lines
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();  // release: publish this workgroup's writes device-wide
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * participants;
      int spin = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spin > (1 << 24)) { atomicAdd(err, 1u); break; }
      }
      __threadfence();  // acquire
    }
    __syncthreads();
  }
}

int main() {
  unsigned *ctr, *err;
  hipMalloc(&ctr, 8 * 32 * 4); hipMalloc(&err, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int per_xcd = 0; per_xcd <= 1; ++per_xcd)
    for (int grid : {128, 256}) {
      const int rounds = 200;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipMemset(ctr, 0, 8 * 32 * 4); hipMemset(err, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ctr, rounds, per_xcd, err);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      unsigned herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("%s barrier, %3d workgroups: %.2f us per barrier (timeouts %u)\n", per_xcd ? "per-XCD (32 or 16 WGs each)" : "grid-wide", grid,
             best * 1e3 / rounds, herr);
    }
  // reference: an empty kernel launch back to back
  float ms;
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, ctr, 0, 0, err);
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("empty 256-workgroup kernel, back-to-back launches: %.2f us each\n", ms * 1e3 / 200);
  return 0;
}
