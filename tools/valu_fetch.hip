// valu_fetch.hip -- does the LENGTH of a straight-line VALU loop body or the spread of its register operands
// change what v_bitop3_b32 / v_and_b32 sustain?  (The bit-sliced median's plane loop is ~600 eight-byte
// instructions over ~160 registers.)   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/valu_fetch tools/valu_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

// BODY instructions per loop iteration, operands cycling through NREG registers; OP 0: v_bitop3 (8 bytes),
// 1: v_and_b32 e32 (4 bytes), 2: v_bitop3 whose three sources sit in the same register bank (index mod 4)
template <int BODY, int NREG, int OP>
__global__ __launch_bounds__(256) void k_fetch(uint32_t *out, uint32_t seed, int iters) {
  uint32_t r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = seed + i * 77u + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < BODY; ++i) {
      constexpr int dummy = 0;
      (void)dummy;
      const int d = i % NREG, a = (i * 7 + 3) % NREG, b = (i * 13 + 5) % NREG;
      if (OP == 0) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[d]) : "v"(r[a]), "v"(r[b]));
      if (OP == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[d]) : "v"(r[a]));
      if (OP == 2) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[d]) : "v"(r[(d + 4) % NREG]), "v"(r[(d + 8) % NREG]));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s ^= r[i];
  if (s == 0x12345u) out[0] = s;
}

template <int BODY, int NREG, int OP>
void run(const char *name) {
  uint32_t *out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("%-46s", name);
  for (int waves : {1, 2, 3, 4}) {
    const int iters = 200000 / BODY * 8, blocks = 256 * waves;
    hipLaunchKernelGGL((k_fetch<BODY, NREG, OP>), dim3(blocks), dim3(256), 0, 0, out, 1u, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_fetch<BODY, NREG, OP>), dim3(blocks), dim3(256), 0, 0, out, 1u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %d w/SIMD %5.2f", waves, ms * 1e-3 * 2.4e9 / (double(iters) * BODY * waves));
  }
  printf("\n");
  hipFree(out);
}

int main() {
  printf("cycles per wave-instruction per SIMD (clock taken as 2.4 GHz)\n");
  run<64, 8, 0>("bitop3, body 64 (0.5 KB), 8 registers");
  run<64, 64, 0>("bitop3, body 64, 64 registers");
  run<512, 64, 0>("bitop3, body 512 (4 KB), 64 registers");
  run<2048, 64, 0>("bitop3, body 2048 (16 KB), 64 registers");
  run<8192, 64, 0>("bitop3, body 8192 (64 KB), 64 registers");
  run<64, 64, 2>("bitop3, sources in ONE bank, body 64");
  run<64, 64, 1>("v_and e32, body 64, 64 registers");
  run<2048, 64, 1>("v_and e32, body 2048 (8 KB), 64 registers");
  run<8192, 64, 1>("v_and e32, body 8192 (32 KB), 64 registers");
  return 0;
}
