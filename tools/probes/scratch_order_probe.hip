// Does a scratch_load return IN ORDER with an older global_load in this wave's vmcnt queue on gfx950?
// (round 5: csrc/Makefile builds lfi_encoder.hip without the SLP vectoriser because the spilling build of enc_gru_bwd_fused_kernel<true>
// is not reproducible run to run; its spill reloads sit between weight-fragment global_loads under counted s_waitcnt vmcnt(N).)
// Every lane: sentinel -> G; global_load G from a cold 128-byte line; scratch_load S from a hot private slot; s_waitcnt vmcnt(1)
// - by in-order return, G must have landed - copy G; then vmcnt(0). Counts the lanes whose copy still holds the sentinel.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/scratch_order_probe.hip -o build/scratch_order_probe && build/scratch_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>   // 0: global then scratch, wait vmcnt(1), read G | 1: control - global then GLOBAL (hot line), wait vmcnt(1), read G
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ big, unsigned long nlines, int iters, const unsigned* __restrict__ hot,
                                              unsigned* __restrict__ bad, unsigned* __restrict__ sink) {
  volatile unsigned slot[8];                       // dynamic index below: lives in scratch
  const unsigned lane = blockIdx.x * 256u + threadIdx.x, total = gridDim.x * 256u;
  for (int i = 0; i < 8; ++i) slot[(i + lane) & 7] = lane * 8u + i;
  const unsigned soff = (unsigned)(unsigned long)(&slot[lane & 7]);   // private aperture: the low 32 bits are the scratch offset
  unsigned nbad = 0, acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long line = ((unsigned long)it * total + lane) % nlines;
    const unsigned* p = big + line * 32;
    const unsigned* h = hot + (lane & 31);
    unsigned g, s, gc;
    if (MODE == 0) {
      asm volatile(
          "v_mov_b32 %0, 0xdeadbeef\n"
          "global_load_dword %0, %3, off\n"
          "scratch_load_dword %1, %4, off\n"
          "s_waitcnt vmcnt(1)\n"
          "v_mov_b32 %2, %0\n"
          "s_waitcnt vmcnt(0)\n"
          : "=&v"(g), "=&v"(s), "=&v"(gc) : "v"(p), "v"(soff) : "memory");
    } else {
      asm volatile(
          "v_mov_b32 %0, 0xdeadbeef\n"
          "global_load_dword %0, %3, off\n"
          "global_load_dword %1, %4, off\n"
          "s_waitcnt vmcnt(1)\n"
          "v_mov_b32 %2, %0\n"
          "s_waitcnt vmcnt(0)\n"
          : "=&v"(g), "=&v"(s), "=&v"(gc) : "v"(p), "v"(h) : "memory");
    }
    nbad += (gc != (unsigned)line) ? 1u : 0u;
    acc += g + s;
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[lane] = acc;
}

int main() {
  const unsigned long nlines = 8ul << 20;           // 1 GiB of 128-byte lines, each read once
  const int grid = 1024, iters = 32;                 // 262 144 lanes x 32 lines = 8 Mi lines
  unsigned *big, *hot, *bad, *sink;
  CHECK(hipMalloc(&big, nlines * 128));
  CHECK(hipMalloc(&hot, 128));
  CHECK(hipMalloc(&bad, 8));
  CHECK(hipMalloc(&sink, (size_t)grid * 256 * 4));
  {
    std::vector<unsigned> hbuf(nlines * 32);
    for (unsigned long l = 0; l < nlines; ++l) for (int w = 0; w < 32; ++w) hbuf[l * 32 + w] = (unsigned)l;
    CHECK(hipMemcpy(big, hbuf.data(), nlines * 128, hipMemcpyHostToDevice));
  }
  CHECK(hipMemset(hot, 0, 128));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(bad, 0, 8));
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, big, nlines, iters, hot, bad, sink);
      else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, big, nlines, iters, hot, bad, sink);
      CHECK(hipDeviceSynchronize());
      unsigned nb = 0;
      CHECK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      printf("%s: %u of %lu lane-loads read G before it had landed after s_waitcnt vmcnt(1)\n",
             mode == 0 ? "global_load G (cold) ; scratch_load S (hot)" : "global_load G (cold) ; global_load S (hot line)  [control]", nb,
             (unsigned long)grid * 256 * iters);
    }
  }
  return 0;
}
