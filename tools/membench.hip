// membench.hip -- calibrates what a 1:4 read:write stream can reach on this
// MI355X, to put the d2pc kernels' roofline fraction in context.
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT> __global__ void k_fill(v4f *out, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += st) { v4f p = {float(i), 1.f, 2.f, 1.f}; if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; }
}
template <bool NT> __global__ void k_copy(const v4f *in, v4f *out, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += st) { v4f p = NT ? __builtin_nontemporal_load(in + i) : in[i]; if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; }
}
template <bool NT> __global__ void k_read(const v4f *in, float *sink, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  float acc = 0;
  for (; i < n; i += st) { v4f p = NT ? __builtin_nontemporal_load(in + i) : in[i]; acc += p.x + p.y + p.z + p.w; }
  if (acc == 123.456f) *sink = acc;
}
// 4 B in, 16 B out per element: the d2pc traffic shape with no arithmetic
template <bool NT, int PXT> __global__ void k_expand(const float *in, v4f *out, size_t n) {
  const size_t tile = size_t(blockDim.x) * PXT;
  for (size_t b = blockIdx.x * tile; b < n; b += size_t(gridDim.x) * tile) {
    float d[PXT];
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? (NT ? __builtin_nontemporal_load(in + i) : in[i]) : 0.f; }
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) { if (NT) __builtin_nontemporal_store(p, out + i); else out[i] = p; } }
  }
}

// 4 B per lane reads (the d2pc kernels' load shape) of a known byte count:
// calibrates rocprofv3's FETCH_SIZE for this access width on gfx950.
__global__ void k_read4(const float *in, float *sink, size_t n) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x, st = size_t(gridDim.x) * blockDim.x;
  float acc = 0;
  for (; i < n; i += st) acc += __builtin_nontemporal_load(in + i);
  if (acc == 123.456f) *sink = acc;
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

int main(int argc, char **argv) {
  const size_t n = size_t(16) * 3840 * 2160;  // 16 4K frames of pixels
  float *in; v4f *out; float *sink;
  CK(hipMalloc(&in, n * 4 * 4)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 16));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  const int cus = prop.multiProcessorCount;
  if (argc > 1 && !strcmp(argv[1], "calib")) {
    // known traffic for PMC calibration: read n*4 bytes with dword loads, 3 launches;
    // then read n*16 bytes with dwordx4 loads, 3 launches; then write n*16 bytes, 3 launches
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_read4, dim3(cus * 8), dim3(256), 0, 0, in, sink, n * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_read<false>, dim3(cus * 8), dim3(256), 0, 0, (const v4f *)in, sink, n);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_fill<false>, dim3(cus * 8), dim3(256), 0, 0, out, n);
    CK(hipDeviceSynchronize());
    printf("calib: k_read4 reads %zu bytes; k_read reads %zu bytes; k_fill writes %zu bytes per launch\n", n * 16, n * 16, n * 16);
    return 0;
  }
  for (int bpc : {4, 8, 16, 32}) {
    const int grid = cus * bpc;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill<false>, dim3(grid), dim3(256), 0, 0, out, n); }, 5);
    printf("bpc %2d fill plain   : %8.1f GB/s\n", bpc, n * 16 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill<true>, dim3(grid), dim3(256), 0, 0, out, n); }, 5);
    printf("bpc %2d fill nt      : %8.1f GB/s\n", bpc, n * 16 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, (const v4f *)out, sink, n); }, 5);
    printf("bpc %2d read plain   : %8.1f GB/s\n", bpc, n * 16 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, (const v4f *)out, sink, n); }, 5);
    printf("bpc %2d read nt      : %8.1f GB/s\n", bpc, n * 16 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, (const v4f *)in, out, n); }, 5);
    printf("bpc %2d copy plain   : %8.1f GB/s\n", bpc, n * 32 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, (const v4f *)in, out, n); }, 5);
    printf("bpc %2d copy nt      : %8.1f GB/s\n", bpc, n * 32 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_expand<false, 4>), dim3(grid), dim3(256), 0, 0, in, out, n); }, 5);
    printf("bpc %2d expand4 plain: %8.1f GB/s\n", bpc, n * 20 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_expand<true, 4>), dim3(grid), dim3(256), 0, 0, in, out, n); }, 5);
    printf("bpc %2d expand4 nt   : %8.1f GB/s\n", bpc, n * 20 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_expand<true, 8>), dim3(grid), dim3(256), 0, 0, in, out, n); }, 5);
    printf("bpc %2d expand8 nt   : %8.1f GB/s\n", bpc, n * 20 / ms / 1e6);
    fflush(stdout);
  }
  return 0;
}
