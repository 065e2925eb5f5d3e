// membench3.hip -- 4B-in/16B-out "expand" stream: which launch shape gets
// closest to the ~6.5 TB/s a pure fill can reach on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ void st16(v4f *p, v4f v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// XCD = true: blocks with equal blockIdx%8 (one XCD under round-robin placement)
// stream one contiguous eighth of the buffer; inside it, tiles are dealt
// round-robin to that XCD's blocks.  PIPE = true: next tile's loads are issued
// before this tile's stores.
template <bool NT, int PXT, bool XCD, bool PIPE>
__global__ void k_expand(const float *__restrict__ in, v4f *__restrict__ out, size_t n) {
  const size_t tile = size_t(blockDim.x) * PXT;
  const size_t ntiles = (n + tile - 1) / tile;
  size_t t, tstep, tend;
  if (XCD) {
    const size_t x = blockIdx.x % 8, j = blockIdx.x / 8, gx = gridDim.x / 8, per = (ntiles + 7) / 8;
    t = x * per + j; tstep = gx; tend = (x + 1) * per < ntiles ? (x + 1) * per : ntiles;
  } else { t = blockIdx.x; tstep = gridDim.x; tend = ntiles; }
  float d[PXT], dn[PXT];
  if (PIPE && t < tend) {
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = t * tile + k * blockDim.x + threadIdx.x; d[k] = i < n ? __builtin_nontemporal_load(in + i) : 0.f; }
  }
  for (; t < tend; t += tstep) {
    const size_t b = t * tile;
    if (PIPE) {
      const size_t tn = t + tstep;
      if (tn < tend) {
#pragma unroll
        for (int k = 0; k < PXT; ++k) { size_t i = tn * tile + k * blockDim.x + threadIdx.x; dn[k] = i < n ? __builtin_nontemporal_load(in + i) : 0.f; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; d[k] = i < n ? __builtin_nontemporal_load(in + i) : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < PXT; ++k) { size_t i = b + k * blockDim.x + threadIdx.x; v4f p = {d[k], d[k] * 2.f, d[k] + 1.f, 1.f}; if (i < n) st16<NT>(out + i, p); }
    if (PIPE) {
#pragma unroll
      for (int k = 0; k < PXT; ++k) d[k] = dn[k];
    }
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

template <bool NT, int PXT, bool XCD, bool PIPE>
void run(const char *name, int cus, const float *in, v4f *out, size_t n) {
  for (int thr : {256, 512, 1024}) {
    printf("%-24s thr=%4d :", name, thr);
    for (int wpc : {4, 8, 16, 32}) {  // waves per CU
      const int bpc_x4 = wpc * 64 * 4 / thr;  // blocks per CU x4
      if (bpc_x4 < 4 && (cus * bpc_x4) % 4) { printf("     -   "); continue; }
      int g = cus * bpc_x4 / 4; if (g < 8) g = 8; g = g / 8 * 8;
      double ms = time_ms([&] { hipLaunchKernelGGL((k_expand<NT, PXT, XCD, PIPE>), dim3(g), dim3(thr), 0, 0, in, out, n); }, 4);
      printf(" w%-2d %6.0f", wpc, double(n) * 20 / ms / 1e6);
    }
    printf("  GB/s\n"); fflush(stdout);
  }
}

int main() {
  const size_t n = size_t(16) * 3840 * 2160;
  float *in; v4f *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 16));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  run<false, 8, false, false>("plain p8", cus, in, out, n);
  run<false, 16, false, false>("plain p16", cus, in, out, n);
  run<false, 32, false, false>("plain p32", cus, in, out, n);
  run<true, 16, false, false>("nt p16", cus, in, out, n);
  run<false, 8, true, false>("plain p8 xcd", cus, in, out, n);
  run<false, 16, true, false>("plain p16 xcd", cus, in, out, n);
  run<false, 32, true, false>("plain p32 xcd", cus, in, out, n);
  run<true, 16, true, false>("nt p16 xcd", cus, in, out, n);
  run<false, 8, true, true>("plain p8 xcd pipe", cus, in, out, n);
  run<false, 16, true, true>("plain p16 xcd pipe", cus, in, out, n);
  run<false, 32, true, true>("plain p32 xcd pipe", cus, in, out, n);
  run<false, 16, false, true>("plain p16 pipe", cus, in, out, n);
  run<true, 16, true, true>("nt p16 xcd pipe", cus, in, out, n);
  return 0;
}
