// membench11.hip -- what do COMPACTED stores cost?  The COMPACT kernels store the survivors of a 64-pixel slot as one
// contiguous piece of (count x 16) bytes at a 16-byte-aligned position, from the lanes that happen to hold a survivor.
// With 30 % iid holes the register-resident kernel runs SLOWER than with all points valid (33.5 vs 30.9 us per 4K frame)
// and 64 x 64 blocky holes are fast (27.1): the bytes are not what costs.  Shapes, one-shot blocks of 256 threads, two
// slots per thread, K of 64 lanes storing per slot (K = 45: 30 % holes), output positions as a real compaction has them:
//   ragged     the K lanes of a rotating mask store; addresses contiguous by rank           (what the kernels do)
//   packed     lanes 0..K-1 store; addresses contiguous, piece starts where the last ended  (survivors moved to low lanes)
//   rows       every instruction stores a whole 64-byte-aligned row of 64 points            (survivors re-blocked into rows)
//   full       K = 64: the PARITY shape
// and the same for 4-byte index stores.   hipcc --offload-arch=gfx950 -O3 -o tools/membench11 tools/membench11.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }

// MODE 0 ragged, 1 packed, 2 rows
template <int MODE, bool NT, bool IDX> __global__ __launch_bounds__(256) void k_store(v4f *out, uint32_t *idx, uint64_t mask0, uint32_t K, size_t nslots) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const size_t slot = (size_t(blockIdx.x) * 4u + wave) * 2u + s;
    if (slot >= nslots) return;
    const uint32_t rot = uint32_t(slot * 7u) & 63u;
    const uint64_t m = MODE == 0 ? ((mask0 << rot) | (rot ? mask0 >> (64u - rot) : 0)) : ~0ull;
    bool on;
    size_t pos;
    if (MODE == 0) { on = (m >> lane) & 1; pos = slot * K + mbcnt64(m); }
    else if (MODE == 1) { on = lane < K; pos = slot * K + lane; }
    else { on = true; pos = slot * 64u + lane; if (slot * 64u >= nslots * K) return; }
    const v4f p = {float(lane), 2.f, 3.f, 1.f};
    if (on) {
      if (NT) __builtin_nontemporal_store(p, out + pos); else out[pos] = p;
      if (IDX) idx[pos] = uint32_t(pos);
    }
  }
}

template <class F> double time_us(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 40; ++i) f();
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
  const size_t nslots = size_t(16) * 7820800 / 64;   // 16 x 4K frames' worth of 64-pixel slots
  v4f *out; uint32_t *idx;
  CK(hipMalloc(&out, nslots * 64 * 16)); CK(hipMalloc(&idx, nslots * 64 * 4));
  const unsigned grid = unsigned((nslots + 7) / 8);
  for (uint32_t K : {64u, 58u, 45u, 26u, 6u}) {
    uint64_t mask = 0; uint32_t have = 0;   // K lanes spread over the wave
    for (uint32_t l = 0; l < 64 && have < K; ++l) if ((l * 37u + 11u) % 64u < K || K == 64) { mask |= 1ull << l; ++have; }
    for (uint32_t l = 0; l < 64 && have < K; ++l) if (!((mask >> l) & 1)) { mask |= 1ull << l; ++have; }
    const double bytes = double(nslots) * K * 16, ibytes = double(nslots) * K * 20;
    printf("K = %2u of 64 lanes (%.0f %% holes), %.2f GB of points per launch\n", K, 100.0 * (64 - K) / 64, bytes / 1e9);
#define RUN(name, MODE, NT, IDX) { double us = time_us([&] { hipLaunchKernelGGL((k_store<MODE, NT, IDX>), dim3(grid), dim3(256), 0, 0, out, idx, mask, K, nslots); }); \
    printf("  %-28s %8.1f us  %7.1f GB/s\n", name, us, (IDX ? ibytes : bytes) / us / 1e3); }
    RUN("ragged, nt", 0, true, false);
    RUN("ragged, plain", 0, false, false);
    RUN("packed, nt", 1, true, false);
    RUN("packed, plain", 1, false, false);
    RUN("rows, nt", 2, true, false);
    RUN("ragged, nt + index", 0, true, true);
    RUN("packed, nt + index", 1, true, true);
    RUN("rows, nt + index", 2, true, true);
  }
  return 0;
}
