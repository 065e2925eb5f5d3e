// membench9.hip -- does the store stream of the headline kernel prefer MANY one-shot blocks?  torch's fill (one 16-byte
// store per thread, 488,800 blocks of 256) writes 2 GB at 6.8 TB/s on a device where a grid-stride fill of 256-2048 blocks
// reaches 5.1-5.9.  Shapes of a pure fill and of the 4 B -> 16 B expansion (the PARITY kernel's traffic without its
// arithmetic), each with S stores per thread, blocks of 256 threads, one block per 256 * S elements (no grid-stride loop),
// against grid-stride forms.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench9 tools/membench9.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int S, bool NT> __global__ __launch_bounds__(256) void k_fill_once(v4f *out, size_t n) {
  const size_t base = size_t(blockIdx.x) * 256u * S + threadIdx.x;
  const v4f v = {1.f, 2.f, 3.f, 1.f};
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const size_t i = base + size_t(j) * 256u;
    if (i < n) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
  }
}
template <int S, bool NT> __global__ __launch_bounds__(256) void k_fill_stride(v4f *out, size_t n) {
  const v4f v = {1.f, 2.f, 3.f, 1.f};
  for (size_t base = size_t(blockIdx.x) * 256u * S + threadIdx.x; base < n; base += size_t(gridDim.x) * 256u * S)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const size_t i = base + size_t(j) * 256u;
      if (i < n) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
}
// 4 B in, 16 B out, S elements per thread, loads first
template <int S, bool NT> __global__ __launch_bounds__(256) void k_expand_once(const float *in, v4f *out, size_t n) {
  const size_t base = size_t(blockIdx.x) * 256u * S + threadIdx.x;
  float d[S];
#pragma unroll
  for (int j = 0; j < S; ++j) { const size_t i = base + size_t(j) * 256u; d[j] = i < n ? in[i] : 0.f; }
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const size_t i = base + size_t(j) * 256u;
    const v4f p = {d[j], d[j] * 2.f, d[j] + 1.f, 1.f};
    if (i < n) { if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; }
  }
}
template <int S, bool NT> __global__ __launch_bounds__(256) void k_expand_stride(const float *in, v4f *out, size_t n) {
  for (size_t base = size_t(blockIdx.x) * 256u * S + threadIdx.x; base < n; base += size_t(gridDim.x) * 256u * S) {
    float d[S];
#pragma unroll
    for (int j = 0; j < S; ++j) { const size_t i = base + size_t(j) * 256u; d[j] = i < n ? in[i] : 0.f; }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const size_t i = base + size_t(j) * 256u;
      const v4f p = {d[j], d[j] * 2.f, d[j] + 1.f, 1.f};
      if (i < n) { if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; }
    }
  }
}

// block-size variants of the one-shot expansion (S = 2)
template <int BS, int S, bool NT> __global__ __launch_bounds__(BS) void k_expand_once_bs(const float *in, v4f *out, size_t n) {
  const size_t base = size_t(blockIdx.x) * BS * S + threadIdx.x;
  float d[S];
#pragma unroll
  for (int j = 0; j < S; ++j) { const size_t i = base + size_t(j) * BS; d[j] = i < n ? in[i] : 0.f; }
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const size_t i = base + size_t(j) * BS;
    const v4f p = {d[j], d[j] * 2.f, d[j] + 1.f, 1.f};
    if (i < n) { if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; }
  }
}

template <class F> double time_us(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 60; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 9; ++r) {
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 100.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const size_t n = size_t(16) * 7820800;  // points of the headline launch
  float *in; v4f *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(in, 0x3f, n * 4));
  auto once = [&](size_t per) { return unsigned((n + 256 * per - 1) / (256 * per)); };
#define FILL_ONCE(S, NT) printf("fill   once   S=%2d %s: %7.1f us %7.1f GB/s\n", S, NT ? "nt   " : "plain", us = time_us([&] { hipLaunchKernelGGL((k_fill_once<S, NT>), dim3(once(S)), dim3(256), 0, 0, out, n); }), n * 16 / us / 1e3)
#define FILL_STR(S, NT, G) printf("fill   stride S=%2d %s grid %6d: %7.1f us %7.1f GB/s\n", S, NT ? "nt   " : "plain", G, us = time_us([&] { hipLaunchKernelGGL((k_fill_stride<S, NT>), dim3(G), dim3(256), 0, 0, out, n); }), n * 16 / us / 1e3)
#define EXP_ONCE(S, NT) printf("expand once   S=%2d %s: %7.1f us %7.1f GB/s\n", S, NT ? "nt   " : "plain", us = time_us([&] { hipLaunchKernelGGL((k_expand_once<S, NT>), dim3(once(S)), dim3(256), 0, 0, in, out, n); }), n * 20 / us / 1e3)
#define EXP_STR(S, NT, G) printf("expand stride S=%2d %s grid %6d: %7.1f us %7.1f GB/s\n", S, NT ? "nt   " : "plain", G, us = time_us([&] { hipLaunchKernelGGL((k_expand_stride<S, NT>), dim3(G), dim3(256), 0, 0, in, out, n); }), n * 20 / us / 1e3)
  double us;
  FILL_ONCE(1, false); FILL_ONCE(1, true); FILL_ONCE(2, false); FILL_ONCE(4, false); FILL_ONCE(4, true); FILL_ONCE(8, false); FILL_ONCE(8, true);
  FILL_STR(4, false, 2048); FILL_STR(4, false, 32768); FILL_STR(8, true, 32768);
  EXP_ONCE(1, false); EXP_ONCE(1, true); EXP_ONCE(2, false); EXP_ONCE(2, true); EXP_ONCE(4, false); EXP_ONCE(4, true); EXP_ONCE(8, false); EXP_ONCE(8, true);
  EXP_ONCE(16, true);
  EXP_STR(4, true, 32768); EXP_STR(8, true, 32768); EXP_STR(8, true, 15275);
#define EXP_BS(BS, S) printf("expand once   S=%2d nt    block %4d: %7.1f us %7.1f GB/s\n", S, BS, us = time_us([&] { hipLaunchKernelGGL((k_expand_once_bs<BS, S, true>), dim3(unsigned((n + size_t(BS) * S - 1) / (size_t(BS) * S))), dim3(BS), 0, 0, in, out, n); }), n * 20 / us / 1e3)
  EXP_BS(64, 2); EXP_BS(128, 2); EXP_BS(256, 2); EXP_BS(512, 2); EXP_BS(1024, 2); EXP_BS(64, 4); EXP_BS(128, 4); EXP_BS(64, 8); EXP_BS(128, 3); EXP_BS(256, 3);
  return 0;
}
