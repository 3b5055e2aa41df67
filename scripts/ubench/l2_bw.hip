// Development aid: achievable L2 -> L1 (vector cache fill) bandwidth on MI355X.  Every workgroup sweeps a buffer that fits
// the XCD's 4 MiB L2 but not the CU's 32 KiB L1, 16 B per lane per load, `NLD` independent loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/l2_bw.hip -o /tmp/l2_bw && /tmp/l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NLD>
__global__ __launch_bounds__(256) void sweep(const f32x4* __restrict__ buf, size_t n_vec, int iters, float* out) {
  // workgroups start at different offsets so that co-resident groups do not read the same lines at the same time
  size_t base = ((size_t)blockIdx.x * 7919u * 256u) % n_vec;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i >= n_vec) i -= n_vec;
      v[u] = buf[i];
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) acc += v[u];
    base += (size_t)NLD * 256;
    if (base >= n_vec) base -= n_vec;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
int main() {
  for (size_t mb : {1, 2, 3, 16, 64}) {
    const size_t bytes = mb << 20, n_vec = bytes / 16;
    f32x4* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 0, bytes);
    for (int wgs : {256, 512, 1024, 2048}) {
      const int iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(sweep<8>, dim3(wgs), dim3(256), 0, 0, d, n_vec, 200, o);
      hipEventRecord(e0);
      hipLaunchKernelGGL(sweep<8>, dim3(wgs), dim3(256), 0, 0, d, n_vec, iters, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double tb = (double)wgs * 256 * 16 * 8 * iters / (ms * 1e-3) / 1e12;
      printf("buffer %3zu MiB, %4d workgroups x 256 threads, 8 x 16 B in flight per lane: %6.2f TB/s\n", mb, wgs, tb);
    }
    hipFree(d); hipFree(o);
  }
  return 0;
}
