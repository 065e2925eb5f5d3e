// membench10.hip -- WHY do long-lived blocks write slower than one-shot blocks on this chip (membench9: 4.9-5.9 against
// 6.9 TB/s for a plain 2-GB fill, and the gap differs between devices)?  Two candidate causes are separated here:
//   * block lifetime itself (a wave that issues store after store), or
//   * the ADDRESS ORDER of the whole launch: one-shot blocks are dispatched in index order, so at any moment the chip
//     writes one compact window of the buffer that moves forward; a grid-stride loop writes gridDim scattered pieces.
// Forms (256 threads, CHUNK = S x 4 KiB per block and step, plain 16-byte stores):
//   once      one block per chunk (the reference point)
//   stride    persistent blocks, chunk c = b + i * gridDim            (scattered: the shape of membench / rounds 1-2)
//   ticket    persistent blocks, chunk c = atomicAdd(counter)          (compact window, long-lived blocks)
//   ranges    persistent blocks, block b owns one contiguous range    (gridDim sequential streams)
//   frames    persistent blocks, F sub-buffers with a ticket each, block b serves sub-buffer b % F   (the single pass's order)
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench10 tools/membench10.hip && ./tools/membench10
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int S> __device__ __forceinline__ void chunk_fill(v4f *out, size_t n, size_t chunk) {
  const v4f v = {1.f, 2.f, 3.f, 1.f};
  const size_t base = chunk * 256u * S + threadIdx.x;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const size_t i = base + size_t(j) * 256u;
    if (i < n) out[i] = v;
  }
}
template <int S> __global__ __launch_bounds__(256) void k_once(v4f *out, size_t n) { chunk_fill<S>(out, n, blockIdx.x); }
template <int S> __global__ __launch_bounds__(256) void k_stride(v4f *out, size_t n, size_t chunks) {
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) chunk_fill<S>(out, n, c);
}
template <int S> __global__ __launch_bounds__(256) void k_ticket(v4f *out, size_t n, size_t chunks, unsigned *counter) {
  __shared__ unsigned s_c;
  for (;;) {
    if (threadIdx.x == 0) s_c = atomicAdd(counter, 1u);
    __syncthreads();
    const unsigned c = s_c;
    __syncthreads();
    if (c >= chunks) break;
    chunk_fill<S>(out, n, c);
  }
}
template <int S> __global__ __launch_bounds__(256) void k_ranges(v4f *out, size_t n, size_t chunks) {
  const size_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const size_t c0 = size_t(blockIdx.x) * per, c1 = std::min(chunks, c0 + per);
  for (size_t c = c0; c < c1; ++c) chunk_fill<S>(out, n, c);
}
template <int S> __global__ __launch_bounds__(256) void k_frames(v4f *out, size_t n, size_t chunks, unsigned *counters, unsigned frames) {
  __shared__ unsigned s_c;
  const unsigned f = blockIdx.x % frames;
  const size_t per = chunks / frames;  // chunks per sub-buffer
  for (;;) {
    if (threadIdx.x == 0) s_c = atomicAdd(counters + f * 64u, 1u);
    __syncthreads();
    const unsigned c = s_c;
    __syncthreads();
    if (c >= per) break;
    chunk_fill<S>(out, n, size_t(f) * per + c);
  }
}

template <class F> static double time_ms(F launch, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    ms.push_back(t / iters);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

template <int S> static void run(v4f *out, size_t n, unsigned *counters, int cus) {
  const size_t chunks = (n + 256u * S - 1) / (256u * S);
  const double gb = double(n) * 16 / 1e9;
  printf("-- chunk = %d KiB (%d stores per thread and step), %zu chunks\n", 4 * S, S, chunks);
  printf("   once                         %7.1f GB/s\n", gb / time_ms([&] { hipLaunchKernelGGL(k_once<S>, dim3(unsigned(chunks)), dim3(256), 0, 0, out, n); }, 20) * 1e3);
  for (int bpc : {3, 8}) {
    const unsigned g = unsigned(cus * bpc);
    printf("   stride  %2d blocks per CU     %7.1f GB/s\n", bpc, gb / time_ms([&] { hipLaunchKernelGGL(k_stride<S>, dim3(g), dim3(256), 0, 0, out, n, chunks); }, 20) * 1e3);
    printf("   ticket  %2d blocks per CU     %7.1f GB/s\n", bpc, gb / time_ms([&] {
             CK(hipMemsetAsync(counters, 0, 4, 0));
             hipLaunchKernelGGL(k_ticket<S>, dim3(g), dim3(256), 0, 0, out, n, chunks, counters);
           }, 20) * 1e3);
    printf("   ranges  %2d blocks per CU     %7.1f GB/s\n", bpc, gb / time_ms([&] { hipLaunchKernelGGL(k_ranges<S>, dim3(g), dim3(256), 0, 0, out, n, chunks); }, 20) * 1e3);
    for (unsigned frames : {16u, 4u}) {
      printf("   frames  %2d blocks per CU, %2u sub-buffers  %7.1f GB/s\n", bpc, frames, gb / time_ms([&] {
               CK(hipMemsetAsync(counters, 0, 64 * 64 * 4, 0));
               hipLaunchKernelGGL(k_frames<S>, dim3(g), dim3(256), 0, 0, out, n, chunks / frames * frames, counters, frames);
             }, 20) * 1e3);
    }
  }
}

int main() {
  const size_t n = size_t(16) * 7820800;  // the headline launch's output: 2.0 GB of 16-byte points
  v4f *out;
  unsigned *counters;
  CK(hipMalloc(&out, n * 16));
  CK(hipMalloc(&counters, 64 * 64 * 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s, %d CUs; plain fill of %.2f GB, median of 7 rounds of 20 launches\n", p.name, p.multiProcessorCount, double(n) * 16 / 1e9);
  run<1>(out, n, counters, p.multiProcessorCount);
  run<4>(out, n, counters, p.multiProcessorCount);
  run<8>(out, n, counters, p.multiProcessorCount);
  return 0;
}
