// Which XCD does block b of a 1-D grid run on?  (hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip && ./xcc_map)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t *o) {
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) o[blockIdx.x] = xcc;
}
int main() {
  for (int threads : {320, 256}) {
    for (int grid : {768, 1024, 64}) {
      uint32_t *d;
      hipMalloc(&d, grid * 4);
      hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d);
      std::vector<uint32_t> h(grid);
      hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
      int mism = 0;
      for (int b = 0; b < grid; ++b) mism += (h[b] & 15u) != uint32_t(b & 7);
      printf("threads %d grid %d: raw reg of blocks 0..15:", threads, grid);
      for (int b = 0; b < 16; ++b) printf(" %x", h[b]);
      printf("  | blocks with (reg & 15) != block %% 8: %d\n", mism);
      hipFree(d);
    }
  }
  return 0;
}
