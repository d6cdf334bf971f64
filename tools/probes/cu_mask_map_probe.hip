// Which CUs does a bit of hipExtStreamCreateWithCUMask's mask enable on this card? Each workgroup records the XCC, shader engine
// and CU it ran on (s_getreg HW_ID / XCC_ID); the host prints, per mask, how many distinct CUs were seen and where.
//   hipcc --offload-arch=gfx950 -O2 -o build/cu_mask_map_probe tools/probes/cu_mask_map_probe.hip && build/cu_mask_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#include <string>

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) { }
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int n = 8192;
  unsigned* out;
  hipMalloc(&out, n * 4);
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 0, st, out, 20000);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(n);
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;  // xcc -> {se:sh:cu}
  for (unsigned v : h) per[v >> 16].insert((v >> 8) & 0xff);
  size_t total = 0;
  for (auto& kv : per) total += kv.second.size();
  printf("%-34s %3zu CUs:", name, total);
  for (auto& kv : per) printf(" x%u=%zu", kv.first, kv.second.size());
  printf("\n");
  hipFree(out);
  hipStreamDestroy(st);
}

int main() {
  auto bits = [](int words, auto pred) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < words * 32; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  run("all of 8 words", bits(8, [](int) { return true; }));
  run("all of 10 words", bits(10, [](int) { return true; }));
  run("low 128 of 8 words", bits(8, [](int i) { return i < 128; }));
  run("low 128 of 10 words", bits(10, [](int i) { return i < 128; }));
  run("low 64 of 10 words", bits(10, [](int i) { return i < 64; }));
  run("low 32 of 10 words", bits(10, [](int i) { return i < 32; }));
  run("low 8 of 10 words", bits(10, [](int i) { return i < 8; }));
  run("bits 0-7 and 64-71 of 10 words", bits(10, [](int i) { return i < 8 || (i >= 64 && i < 72); }));
  run("every other bit of 10 words", bits(10, [](int i) { return i % 2 == 0; }));
  run("every 8th bit of 10 words", bits(10, [](int i) { return i % 8 == 0; }));
  run("bits with (i/8)%2==0, 10 words", bits(10, [](int i) { return (i / 8) % 2 == 0; }));
  run("bits i%8<4, 10 words", bits(10, [](int i) { return i % 8 < 4; }));
  run("bits i%16<8, 10 words", bits(10, [](int i) { return i % 16 < 8; }));
  run("bits i%32<16, 10 words", bits(10, [](int i) { return i % 32 < 16; }));
  run("bits i%64<32, 10 words", bits(10, [](int i) { return i % 64 < 32; }));
  run("one word, all", bits(1, [](int) { return true; }));
  run("one word, low 16", bits(1, [](int i) { return i < 16; }));
  run("two words, all", bits(2, [](int) { return true; }));
  return 0;
}
