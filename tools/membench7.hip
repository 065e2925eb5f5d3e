// membench7.hip -- is the 4 B -> 16 B stream bound by per-CU request concurrency (TCP pending
// requests x loaded latency) or by the memory system?  Runs the expand kernel on streams restricted
// to a subset of the CUs (hipExtStreamCreateWithCUMask): a memory-bound stream keeps its rate when
// CUs are taken away, a concurrency-bound one loses rate in proportion.
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
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) __builtin_nontemporal_store(p, out + i); }
  }
}
__global__ void k_fill(v4f *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { v4f p = {1.f, 2.f, 3.f, 1.f}; __builtin_nontemporal_store(p, out + i); }
}
int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  float *in; v4f *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16)); CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 16));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%d CUs\n", cus);
  // CU mask bit i = CU i; which physical CU / XCD that is depends on the driver's numbering, so take
  // evenly spread subsets (every 2nd, 4th ...) as well as the first N
  struct Pat { const char *name; int keep_of; int first; };
  const Pat pats[] = {{"all", 1, 0}, {"every 2nd", 2, 0}, {"every 4th", 4, 0}, {"first half", 0, cus / 2}, {"first quarter", 0, cus / 4}};
  for (const Pat &p : pats) {
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    int active = 0;
    for (int i = 0; i < cus; ++i) { bool on = p.keep_of ? (i % p.keep_of == 0) : (i < p.first); if (on) { mask[i / 32] |= 1u << (i % 32); ++active; } }
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, uint32_t(mask.size()), mask.data()));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int which = 0; which < 2; ++which) {
      auto launch = [&] { if (which == 0) hipLaunchKernelGGL((k_expand<8>), dim3(cus * 32), dim3(256), 0, s, in, out, n); else hipLaunchKernelGGL(k_fill, dim3(cus * 32), dim3(256), 0, s, out, n); };
      launch(); CK(hipStreamSynchronize(s));
      std::vector<float> t;
      for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a, s)); for (int i = 0; i < 4; ++i) launch(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 4 * 1e3); }
      std::sort(t.begin(), t.end());
      const double bytes = which == 0 ? double(n) * 20 : double(n) * 16;
      printf("%-14s %3d CUs  %-7s %8.1f us  %7.0f GB/s  (%5.1f GB/s per CU)\n", p.name, active, which == 0 ? "expand" : "fill", t[2], bytes / t[2] / 1e3, bytes / t[2] / 1e3 / active);
      fflush(stdout);
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
