// Development probe: which XCDs / compute units does a CU-masked stream run on?  (hipExtStreamCreateWithCUMask bit -> CU map)
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);          // HW_REG_HW_ID[15:0]
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}

static void run(const char* name, const std::vector<uint32_t>& mask, int blocks) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  unsigned* d;
  hipMalloc(&d, blocks * 4);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(blocks);
  hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> cus;   // xcc -> distinct (se, sh, cu)
  std::map<unsigned, int> cnt;
  for (unsigned v : h) { cus[v >> 16].insert((v >> 8) & 0xff); cnt[v >> 16]++; }
  int bits = 0;
  for (uint32_t m : mask) bits += __builtin_popcount(m);
  printf("%-28s bits=%3d:", name, bits);
  int total = 0;
  for (auto& kv : cus) { printf("  xcc%u: %zu CUs (%d blk)", kv.first, kv.second.size(), cnt[kv.first]); total += (int)kv.second.size(); }
  printf("  | distinct CUs %d\n", total);
  uint32_t back[16] = {0};
  hipExtStreamGetCUMask(s, 16, back);
  int b2 = 0; for (uint32_t m : back) b2 += __builtin_popcount(m);
  printf("%-28s GetCUMask bits=%d\n", "", b2);
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  const int W = 8;
  auto all = [&]() { return std::vector<uint32_t>(W, 0xffffffffu); };
  { auto m = all(); run("all 256", m, 4096); }
  { std::vector<uint32_t> m(W, 0); m[0] = 0xffffffffu; run("bits 0..31", m, 4096); }
  { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 7; ++i) m[i] = 0xffffffffu; run("bits 0..223", m, 4096); }
  { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; ++i) if (i % 8 != 7) m[i / 32] |= 1u << (i % 32); run("i%8 != 7", m, 4096); }
  { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; ++i) if (i % 8 == 7) m[i / 32] |= 1u << (i % 32); run("i%8 == 7", m, 4096); }
  { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; ++i) if (i % 8 < 6) m[i / 32] |= 1u << (i % 32); run("i%8 < 6", m, 4096); }
  { std::vector<uint32_t> m(W, 0); for (int i = 0; i < 256; ++i) if ((i / 8) % 4 != 3) m[i / 32] |= 1u << (i % 32); run("(i/8)%4 != 3", m, 4096); }
  hipStream_t s0; hipStreamCreate(&s0);
  uint32_t back[16] = {0};
  hipError_t e = hipExtStreamGetCUMask(s0, 16, back);
  int b2 = 0; for (uint32_t m : back) b2 += __builtin_popcount(m);
  printf("plain stream: GetCUMask rc=%d bits=%d\n", (int)e, b2);
  return 0;
}
