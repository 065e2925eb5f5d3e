// membench5.hip -- does the allocation TYPE of the output buffer change the streaming-write rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
template <int PXT, bool NT> __global__ void k_expand(const float *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t tile = size_t(blockDim.x) * PXT;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
    float d[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? in[i] : 0.f; }
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) { if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; } }
  }
}
template <class F> double time_us(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 5 * 1e3); }
  std::sort(t.begin(), t.end()); return t[2];
}
int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  float *in; CK(hipMalloc(&in, n * 4)); CK(hipMemset(in, 1, n * 4));
  struct B { const char *name; v4f *p; };
  std::vector<B> bufs;
  for (int r = 0; r < 2; ++r) {
    v4f *p;
    CK(hipMalloc(&p, n * 16)); bufs.push_back({"hipMalloc", p});
    if (hipExtMallocWithFlags((void **)&p, n * 16, hipDeviceMallocFinegrained) == hipSuccess) bufs.push_back({"finegrained", p}); else printf("finegrained alloc failed\n");
    if (hipExtMallocWithFlags((void **)&p, n * 16, hipDeviceMallocUncached) == hipSuccess) bufs.push_back({"uncached", p}); else printf("uncached alloc failed\n");
  }
  const int g = 256 * 128;
  for (auto &b : bufs) {
    CK(hipMemset(b.p, 0, n * 16));
    double t1 = time_us([&] { hipLaunchKernelGGL((k_expand<8, false>), dim3(g), dim3(256), 0, 0, in, b.p, n); });
    double t2 = time_us([&] { hipLaunchKernelGGL((k_expand<8, true>), dim3(g), dim3(256), 0, 0, in, b.p, n); });
    printf("%-12s @%p: expand plain %7.1f us (%6.0f GB/s)   nt %7.1f us (%6.0f GB/s)\n", b.name, (void *)b.p, t1, n * 20 / t1 / 1e3, t2, n * 20 / t2 / 1e3);
    fflush(stdout);
  }
  return 0;
}
