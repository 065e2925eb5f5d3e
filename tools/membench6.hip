// membench6.hip -- time-division of reads and writes: all waves load only in
// the first R ticks of every P-tick period of the chip-wide 100 MHz clock
// (s_memrealtime) and store only in the rest.  Does separating the two
// directions at the DRAM help a 4 B -> 16 B stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void wait_window(unsigned P, unsigned lo, unsigned hi) {
  if (P == 0) return;
  for (;;) {
    const unsigned t = unsigned(__builtin_amdgcn_s_memrealtime() % P);
    if (t >= lo && t < hi) return;
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int PXT> __global__ void k_expand(const float *__restrict__ in, v4f *__restrict__ out, size_t n, unsigned P, unsigned R) {
  const size_t tile = size_t(blockDim.x) * PXT;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
    float d[PXT];
    wait_window(P, 0, R);
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? in[i] : 0.f; }
    // make sure the data has landed before the write window is awaited
    float s = 0; 
#pragma unroll
    for (int k = 0; k < PXT; ++k) s += d[k];
    if (s == 123.456f) out[0] = v4f{0, 0, 0, 0};
    wait_window(P, R, P);
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) __builtin_nontemporal_store(p, out + i); }
  }
}
template <class F> double time_us(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < 4; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 4 * 1e3); }
  std::sort(t.begin(), t.end()); return t[2];
}
int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  float *in; v4f *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16)); CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 16));
  for (int bpc : {4, 8}) {
    const int g = 256 * bpc;
    double t0 = time_us([&] { hipLaunchKernelGGL((k_expand<16>), dim3(g), dim3(256), 0, 0, in, out, n, 0u, 0u); });
    printf("bpc %d  ungated            : %7.1f us (%6.0f GB/s)\n", bpc, t0, n * 20 / t0 / 1e3);
    for (unsigned P : {500u, 1000u, 2000u, 4000u})
      for (unsigned Rpct : {15u, 25u, 40u}) {
        const unsigned R = P * Rpct / 100;
        double t = time_us([&] { hipLaunchKernelGGL((k_expand<16>), dim3(g), dim3(256), 0, 0, in, out, n, P, R); });
        printf("bpc %d  P=%5.1f us R=%2u%%   : %7.1f us (%6.0f GB/s)\n", bpc, P / 100.0, Rpct, t, n * 20 / t / 1e3);
        fflush(stdout);
      }
  }
  return 0;
}
