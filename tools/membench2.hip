// membench2.hip -- explores WRITE-stream shapes on MI355X (which store pattern
// gets closest to HBM peak?).  hipcc --offload-arch=gfx950 -O3 -o tools/membench2 tools/membench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

enum { ST_PLAIN = 0, ST_NT = 1, ST_SC1 = 2, ST_SC0SC1 = 3, ST_SC0 = 4, ST_NTSC1 = 5 };

template <int ST> __device__ __forceinline__ void st16(v4f *p, v4f v) {
  if (ST == ST_PLAIN) *p = v;
  else if (ST == ST_NT) __builtin_nontemporal_store(v, p);
  else if (ST == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_SC0) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// grid-stride: block b writes 4 KiB (256 thr x 16 B) pieces b, b+G, b+2G ...
template <int ST> __global__ void k_fill_stride(v4f *out, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += st) { v4f p = {float(i), 1.f, 2.f, 1.f}; st16<ST>(out + i, p); }
}
// chunked: block b owns one contiguous span of n/G elements
template <int ST> __global__ void k_fill_chunk(v4f *out, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b0 = blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) { v4f p = {float(i), 1.f, 2.f, 1.f}; st16<ST>(out + i, p); }
}
// tile-strided with UNROLL stores in flight per thread; tile = blockDim*U elements
template <int ST, int U> __global__ void k_fill_tile(v4f *out, size_t n) {
  const size_t tile = size_t(blockDim.x) * U;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
#pragma unroll
    for (int k = 0; k < U; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {float(i), 1.f, 2.f, 1.f}; if (i < n) st16<ST>(out + i, p); }
  }
}
// XCD-aware: blocks b%8 == x write region x (each XCD streams its own eighth)
template <int ST> __global__ void k_fill_xcd(v4f *out, size_t n) {
  const size_t x = blockIdx.x % 8, j = blockIdx.x / 8, per_x = n / 8, gx = gridDim.x / 8;
  v4f *o = out + x * per_x;
  for (size_t i = j * size_t(blockDim.x) + threadIdx.x; i < per_x; i += gx * blockDim.x) { v4f p = {float(i), 1.f, 2.f, 1.f}; st16<ST>(o + i, p); }
}
template <int ST, int PXT> __global__ void k_expand(const float *in, v4f *out, size_t n) {
  const size_t tile = size_t(blockDim.x) * PXT;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
    float d[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? __builtin_nontemporal_load(in + i) : 0.f; }
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) st16<ST>(out + i, p); }
  }
}

template <class F> double time_ms(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
#define RUN(label, bytes, ...) do { double ms = time_ms([&] { __VA_ARGS__; }, 4); printf("%-44s %8.1f GB/s\n", label, double(bytes) / ms / 1e6); fflush(stdout); } while (0)
#define L(k, g, b, ...) hipLaunchKernelGGL(k, dim3(g), dim3(b), 0, 0, __VA_ARGS__)

int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  float *in; v4f *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 16));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  char lab[128];
  RUN("hipMemsetAsync(D8)", n * 16, CK(hipMemsetAsync(out, 7, n * 16, 0)));
  RUN("hipMemsetD32Async", n * 16, CK(hipMemsetD32Async((hipDeviceptr_t)out, 7, n * 4, 0)));
  for (int bpc : {1, 2, 4, 8}) {
    const int g = cus * bpc;
    snprintf(lab, sizeof lab, "stride plain bpc%d", bpc);   RUN(lab, n * 16, L(k_fill_stride<ST_PLAIN>, g, 256, out, n));
    snprintf(lab, sizeof lab, "stride nt bpc%d", bpc);      RUN(lab, n * 16, L(k_fill_stride<ST_NT>, g, 256, out, n));
    snprintf(lab, sizeof lab, "stride sc1 bpc%d", bpc);     RUN(lab, n * 16, L(k_fill_stride<ST_SC1>, g, 256, out, n));
    snprintf(lab, sizeof lab, "stride sc0sc1 bpc%d", bpc);  RUN(lab, n * 16, L(k_fill_stride<ST_SC0SC1>, g, 256, out, n));
    snprintf(lab, sizeof lab, "stride sc0 bpc%d", bpc);     RUN(lab, n * 16, L(k_fill_stride<ST_SC0>, g, 256, out, n));
    snprintf(lab, sizeof lab, "stride nt+sc1 bpc%d", bpc);  RUN(lab, n * 16, L(k_fill_stride<ST_NTSC1>, g, 256, out, n));
    snprintf(lab, sizeof lab, "chunk plain bpc%d", bpc);    RUN(lab, n * 16, L(k_fill_chunk<ST_PLAIN>, g, 256, out, n));
    snprintf(lab, sizeof lab, "chunk nt bpc%d", bpc);       RUN(lab, n * 16, L(k_fill_chunk<ST_NT>, g, 256, out, n));
    snprintf(lab, sizeof lab, "xcd plain bpc%d", bpc);      RUN(lab, n * 16, L(k_fill_xcd<ST_PLAIN>, g, 256, out, n));
    snprintf(lab, sizeof lab, "xcd nt bpc%d", bpc);         RUN(lab, n * 16, L(k_fill_xcd<ST_NT>, g, 256, out, n));
    snprintf(lab, sizeof lab, "tile4 nt bpc%d", bpc);       RUN(lab, n * 16, L((k_fill_tile<ST_NT, 4>), g, 256, out, n));
    snprintf(lab, sizeof lab, "tile16 nt bpc%d", bpc);      RUN(lab, n * 16, L((k_fill_tile<ST_NT, 16>), g, 256, out, n));
    snprintf(lab, sizeof lab, "stride plain 1024thr bpc%d", bpc); RUN(lab, n * 16, L(k_fill_stride<ST_PLAIN>, g / 4 > 0 ? g / 4 : 1, 1024, out, n));
    snprintf(lab, sizeof lab, "stride nt 64thr bpc%d", bpc);      RUN(lab, n * 16, L(k_fill_stride<ST_NT>, g * 4, 64, out, n));
    snprintf(lab, sizeof lab, "expand8 nt bpc%d", bpc);     RUN(lab, n * 20, L((k_expand<ST_NT, 8>), g, 256, in, out, n));
    snprintf(lab, sizeof lab, "expand8 sc1 bpc%d", bpc);    RUN(lab, n * 20, L((k_expand<ST_SC1, 8>), g, 256, in, out, n));
    snprintf(lab, sizeof lab, "expand8 sc0sc1 bpc%d", bpc); RUN(lab, n * 20, L((k_expand<ST_SC0SC1, 8>), g, 256, in, out, n));
    snprintf(lab, sizeof lab, "expand8 plain bpc%d", bpc);  RUN(lab, n * 20, L((k_expand<ST_PLAIN, 8>), g, 256, in, out, n));
  }
  return 0;
}
