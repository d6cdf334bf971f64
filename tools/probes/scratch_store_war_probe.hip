// Does a VALU write of v[a:a+1] in the instruction right after `scratch_store_dwordx2 off, v[a:a+1], off` corrupt the stored data on gfx950?
// (round 5, the SLP chase: the spilling build of enc_gru_bwd_fused_kernel<true> has exactly that pair five times in its prologue -
//  `scratch_store_dwordx2 off, v[6:7], off offset:8 ; v_lshl_add_u64 v[6:7], s[16:17], 0, v[4:5]` - spilling the weight-fragment pointers
//  of the three gates; hipcc inserts no wait state there: the documented store-data hazard covers MORE than 64 bits of data.)
// Every lane and iteration: v[a:a+1] = OLD; store; overwrite with NEW by the named instruction (0 .. 3 s_nop between); drain; reload; compare.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/scratch_store_war_probe.hip -o build/scratch_store_war_probe && build/scratch_store_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// KIND 0: v_lshl_add_u64 (the compiler's pair) 1: v_mov_b64 3: v_pk_mov_b32;  NOPS: s_nop wait states in between
template <int KIND, int NOPS, int FORM>   // FORM 0: VGPR offset; 1: `off ... off offset:64` (the compiler's spill form: no address VGPR)
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long* __restrict__ bad, unsigned long long* __restrict__ sink, int busy) {
  volatile unsigned long long slot[32];   // 256 bytes of private memory: the FORM-1 accesses at byte 64 of the frame stay inside it
  const unsigned lane = blockIdx.x * 256u + threadIdx.x;
  for (int i = 0; i < 32; ++i) slot[(i + lane) & 31] = 0;
  const unsigned soff = (unsigned)(unsigned long)(&slot[lane & 3]);
  unsigned long long nbad = 0, acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long oldv = 0x1111000000000000ull + ((unsigned long long)lane << 16) + it;
    const unsigned long long newv = 0x2222000000000000ull + ((unsigned long long)lane << 16) + it;
    unsigned long long reg = oldv, back;
    if (busy) {   // other memory traffic of this wave in front of the store (as in the kernel's prologue)
      slot[(lane + 1) & 3] = it; slot[(lane + 2) & 3] = it;
    }
#define BODY(OVERWRITE)                                                                       \
    if (FORM == 0)                                                                            \
      asm volatile("scratch_store_dwordx2 %3, %0, off\n" NOPSTR OVERWRITE                     \
                   "s_waitcnt vmcnt(0)\n"                                                     \
                   "scratch_load_dwordx2 %1, %3, off\n"                                       \
                   "s_waitcnt vmcnt(0)\n"                                                     \
                   : "+v"(reg), "=&v"(back) : "v"(newv), "v"(soff) : "memory");               \
    else                                                                                      \
      asm volatile("scratch_store_dwordx2 off, %0, off offset:64\n" NOPSTR OVERWRITE          \
                   "s_waitcnt vmcnt(0)\n"                                                     \
                   "scratch_load_dwordx2 %1, off, off offset:64\n"                            \
                   "s_waitcnt vmcnt(0)\n"                                                     \
                   : "+v"(reg), "=&v"(back) : "v"(newv), "v"(soff) : "memory")
#define NOPSTR ""
    if (NOPS == 0) {
      if (KIND == 0) BODY("v_lshl_add_u64 %0, %2, 0, 0\n");
      else if (KIND == 1) BODY("v_mov_b64 %0, %2\n");
      else BODY("v_pk_mov_b32 %0, %2, %2\n");
    }
#undef NOPSTR
#define NOPSTR "s_nop 0\n"
    if (NOPS == 1) { if (KIND == 0) BODY("v_lshl_add_u64 %0, %2, 0, 0\n"); else BODY("v_mov_b64 %0, %2\n"); }
#undef NOPSTR
#define NOPSTR "s_nop 1\n"
    if (NOPS == 2) { if (KIND == 0) BODY("v_lshl_add_u64 %0, %2, 0, 0\n"); else BODY("v_mov_b64 %0, %2\n"); }
#undef NOPSTR
#define NOPSTR "s_nop 3\n"
    if (NOPS == 4) { if (KIND == 0) BODY("v_lshl_add_u64 %0, %2, 0, 0\n"); else BODY("v_mov_b64 %0, %2\n"); }
#undef NOPSTR
    nbad += back != oldv ? 1 : 0;
    acc += reg + back;
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[lane] = acc;
}

int main() {
  const int grid = 2048, iters = 256;
  unsigned long long *bad, *sink;
  CHECK(hipMalloc(&bad, 8));
  CHECK(hipMalloc(&sink, (size_t)grid * 256 * 8));
  const char* names[] = {"v_lshl_add_u64", "v_mov_b64", "-", "v_pk_mov_b32"};
#define RUN(K, N, BUSY) RUNF(K, N, BUSY, 0) RUNF(K, N, BUSY, 1)
#define RUNF(K, N, BUSY, FM)                                                                                          \
  {                                                                                                                \
    CHECK(hipMemset(bad, 0, 8));                                                                                   \
    hipLaunchKernelGGL((probe<K, N, FM>), dim3(grid), dim3(256), 0, 0, iters, bad, sink, BUSY);                        \
    CHECK(hipDeviceSynchronize());                                                                                 \
    unsigned long long nb = 0;                                                                                     \
    CHECK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost));                                                          \
    printf("scratch_store_dwordx2 %s ; %d wait state(s) ; %-14s%s: %llu of %llu stores held the NEW register value\n", \
           FM ? "off, v[a:a+1], off offset:64" : "v_off, v[a:a+1], off        ", N, names[K], BUSY ? " (two more scratch stores in front)" : "", nb, (unsigned long long)grid * 256 * iters); \
  }
  for (int busy = 0; busy < 2; ++busy) {
    RUN(0, 0, busy) RUN(1, 0, busy) RUN(3, 0, busy) RUN(0, 1, busy) RUN(0, 2, busy) RUN(0, 4, busy) RUN(1, 1, busy)
  }
  return 0;
}
