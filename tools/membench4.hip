// membench4.hip -- does streaming-write speed depend on WHICH allocation is written?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
template <int PXT> __global__ void k_expand(const float *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t tile = size_t(blockDim.x) * PXT;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
    float d[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? in[i] : 0.f; }
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) out[i] = p; }
  }
}
__global__ void k_fill(v4f *out, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += st) { v4f p = {float(i), 1.f, 2.f, 1.f}; out[i] = p; }
}
template <class F> double time_us(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 5 * 1e3); }
  std::sort(t.begin(), t.end()); return t[2];
}
int main(int argc, char **argv) {
  const size_t n = size_t(16) * 3840 * 2160;
  const int K = 8;
  float *in; CK(hipMalloc(&in, n * 4)); CK(hipMemset(in, 1, n * 4));
  std::vector<v4f *> outs(K);
  const size_t alloc_bytes = argc > 1 ? (size_t(atoll(argv[1])) << 20) : n * 16;  // MiB
  printf("per-buffer allocation: %zu bytes\n", alloc_bytes);
  for (int k = 0; k < K; ++k) { CK(hipMalloc(&outs[k], alloc_bytes)); CK(hipMemset(outs[k], 0, n * 16)); }
  v4f *arena; CK(hipMalloc(&arena, size_t(4) * n * 16));
  for (int k = 0; k < 4; ++k) outs.push_back(arena + size_t(k) * n);
  const int g = 256 * 16;
  for (size_t k = 0; k < outs.size(); ++k) {
    double te = time_us([&] { hipLaunchKernelGGL((k_expand<16>), dim3(g), dim3(256), 0, 0, in, outs[k], n); });
    double tf = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 2), dim3(256), 0, 0, outs[k], n); });
    printf("buf %2zu %s @%p: expand %7.1f us (%6.0f GB/s)   fill %7.1f us (%6.0f GB/s)\n", k, k < (size_t)K ? "own  " : "arena", (void *)outs[k], te, n * 20 / te / 1e3, tf, n * 16 / tf / 1e3);
    fflush(stdout);
  }
  return 0;
}
