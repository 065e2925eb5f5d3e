// membench8.hip -- does the block -> address mapping across the 8 XCDs matter for a streaming
// write (fill) or a 4 B -> 16 B expand?  Blocks are dispatched round-robin to XCDs (xcd = block % 8).
//   map 0: tile t = block, block + grid, ...                (neighbouring tiles on different XCDs)
//   map 1: every XCD owns one contiguous eighth of the buffer and streams through it
//   map 2: XCDs interleaved at a coarser grain: runs of RUN consecutive tiles belong to one XCD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int PXT = 8;
constexpr size_t kTile = 256 * PXT;  // elements per tile (32 KB of output)

__device__ __forceinline__ size_t tile_of(int map, int run, size_t k, size_t ntiles) {
  // k-th tile of this block's sequence
  const size_t b = blockIdx.x, g = gridDim.x;
  if (map == 0) return b + k * g;
  const size_t xcd = b % 8, lb = b / 8, lg = g / 8;  // position among the blocks of my XCD
  const size_t j = lb + k * lg;                     // j-th tile of my XCD
  if (map == 1) return xcd * (ntiles / 8) + j;
  return ((j / run) * 8 + xcd) * run + (j % run);    // map 2
}
template <bool READ> __global__ void k_stream(const float *__restrict__ in, v4f *__restrict__ out, size_t ntiles, int map, int run) {
  const size_t per_block = map == 0 ? (ntiles + gridDim.x - 1) / gridDim.x : (ntiles / 8 + gridDim.x / 8 - 1) / (gridDim.x / 8);
  for (size_t k = 0; k < per_block; ++k) {
    const size_t t = tile_of(map, run, k, ntiles);
    if (t >= ntiles) continue;
    if (map != 0 && (blockIdx.x / 8) + k * (gridDim.x / 8) >= ntiles / 8) continue;
    const size_t base = t * kTile;
    float d[PXT];
#pragma unroll
    for (int q = 0; q < PXT; ++q) d[q] = READ ? in[base + q * 256 + threadIdx.x] : 1.f;
#pragma unroll
    for (int q = 0; q < PXT; ++q) { v4f p = {d[q], d[q] * 2.f, d[q] + 1.f, 1.f}; __builtin_nontemporal_store(p, out + base + q * 256 + threadIdx.x); }
  }
}
int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  const size_t ntiles = n / kTile / 8 * 8;
  float *in; v4f *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16)); CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct V { int map, run; };
  const V vs[] = {{0, 0}, {1, 0}, {2, 1}, {2, 2}, {2, 4}, {2, 8}, {2, 32}, {2, 128}};
  for (int grid : {256 * 8, 256 * 32}) {
    for (int rd = 0; rd < 2; ++rd)
      for (int rep = 0; rep < 2; ++rep)
        for (const V &v : vs) {
          auto launch = [&] { if (rd) hipLaunchKernelGGL((k_stream<true>), dim3(grid), dim3(256), 0, 0, in, out, ntiles, v.map, v.run); else hipLaunchKernelGGL((k_stream<false>), dim3(grid), dim3(256), 0, 0, in, out, ntiles, v.map, v.run); };
          launch(); CK(hipDeviceSynchronize());
          std::vector<float> t;
          for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < 4; ++i) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 4 * 1e3); }
          std::sort(t.begin(), t.end());
          const double bytes = double(ntiles) * kTile * (rd ? 20 : 16);
          printf("grid %5d  %-6s map %d run %3d : %8.1f us  %7.0f GB/s\n", grid, rd ? "expand" : "fill", v.map, v.run, t[2], bytes / t[2] / 1e3);
          fflush(stdout);
        }
  }
  return 0;
}
