// d2pc_membench.hip -- two streaming kernels that calibrate THIS device in the same run as the benchmark:
// a plain fill and a plain copy, 16 bytes per lane, every wave instruction one contiguous 1-KiB piece (the
// store shape of k_reproject_pack).  bench.py reports their rates as roofline.device_fill_GBs /
// device_copy_GBs, so that a kernel's fraction of the 8 TB/s specification can also be read against what
// the device it ran on gives a kernel that does nothing else.  No counterpart in the reference.
#include "d2pc_launch.hpp"

namespace d2pc {

typedef float v4f __attribute__((ext_vector_type(4)));

// A block owns runs of UNROLL KiB-per-wave pieces: thread t of the block touches element base + j*256 + t for
// j < UNROLL, so each wave instruction covers 1 KiB and the block's step UNROLL x 4 KiB.  Two launch shapes:
//   persistent  blocks = CUs x membench_blocks_per_cu, each walking the buffer with a grid stride (UNROLL 4, plain
//               stores): the shape of the single-pass kernels -- and the one that tells device classes apart
//   one-shot    one block per UNROLL x 4 KiB (tuning membench_blocks_per_cu = 0): the shape of the headline kernel and
//               of the runtime's own fill; with UNROLL 1 it is the fastest writer found on this chip
//               (profiles/r03_membench9.txt), i.e. the ceiling a store stream can be read against
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_membench_fill(v4f *__restrict__ dst, uint64_t n16) {
  const uint64_t step = uint64_t(gridDim.x) * 256u * UNROLL;
  const v4f v = {1.0f, 2.0f, 3.0f, 1.0f};
  for (uint64_t base = uint64_t(blockIdx.x) * 256u * UNROLL; base < n16; base += step) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      if (i < n16) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
      }
    }
  }
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_membench_copy(const v4f *__restrict__ src, v4f *__restrict__ dst, uint64_t n16) {
  const uint64_t step = uint64_t(gridDim.x) * 256u * UNROLL;
  for (uint64_t base = uint64_t(blockIdx.x) * 256u * UNROLL; base < n16; base += step) {
    v4f v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      v[j] = i < n16 ? src[i] : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      if (i < n16) {
        if (NT) __builtin_nontemporal_store(v[j], dst + i);
        else dst[i] = v[j];
      }
    }
  }
}

// The shader clock WHILE other kernels run (d2pc_clock_probe_device): 8 one-wave blocks -- dealt round-robin to the 8 XCDs --
// sleep until `min_ticks` ticks of the constant 100 MHz counter have passed and report how many shader cycles (s_memtime)
// that took.  A wave that sleeps takes no issue slots from the kernel it is launched beside (another stream); the loop ends
// after min_ticks <= 2 s whatever happens.  bench.py: the callback body holds ~1.7 GHz under its select, 2.3 elsewhere.
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *__restrict__ out, uint64_t min_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  uint64_t t = t0;
  while (t - t0 < min_ticks) {
    __builtin_amdgcn_s_sleep(64);
    t = __builtin_amdgcn_s_memrealtime();
  }
  const uint64_t c = __builtin_amdgcn_s_memtime();
  out[2u * blockIdx.x] = c - c0;
  out[2u * blockIdx.x + 1u] = t - t0;
}

hipError_t launch_clock_probe(void *out16, uint32_t min_us, hipStream_t stream) {
  const uint64_t ticks = uint64_t(min_us > 2000000u ? 2000000u : min_us) * 100u;
  hipLaunchKernelGGL(k_clock_probe, dim3(8), dim3(64), 0, stream, static_cast<unsigned long long *>(out16), ticks);
  return hipGetLastError();
}

// blocks = 0: one-shot (one block per unroll x 4 KiB); unroll 1, 2 or 4
static uint32_t membench_grid(uint64_t n16, uint32_t blocks, int unroll) {
  if (blocks) return blocks;
  const uint64_t per = 256ull * uint64_t(unroll);
  const uint64_t g = (n16 + per - 1) / per;
  return uint32_t(g > 0x7fffffffull ? 0x7fffffffull : g);
}

hipError_t launch_membench_fill(void *dst, size_t bytes, uint32_t blocks, int unroll, bool nt, hipStream_t stream) {
  const uint64_t n16 = uint64_t(bytes / 16);
  const uint32_t grid = membench_grid(n16, blocks, unroll);
#define D2PC_MB_FILL(U, N) hipLaunchKernelGGL((k_membench_fill<U, N>), dim3(grid), dim3(256), 0, stream, static_cast<v4f *>(dst), n16)
  switch (unroll * 2 + (nt ? 1 : 0)) {
    case 2: D2PC_MB_FILL(1, false); break;
    case 3: D2PC_MB_FILL(1, true); break;
    case 4: D2PC_MB_FILL(2, false); break;
    case 5: D2PC_MB_FILL(2, true); break;
    case 8: D2PC_MB_FILL(4, false); break;
    case 9: D2PC_MB_FILL(4, true); break;
    default: return hipErrorInvalidValue;
  }
#undef D2PC_MB_FILL
  return hipGetLastError();
}

hipError_t launch_membench_copy(const void *src, void *dst, size_t bytes, uint32_t blocks, int unroll, bool nt, hipStream_t stream) {
  const uint64_t n16 = uint64_t(bytes / 16);
  const uint32_t grid = membench_grid(n16, blocks, unroll);
#define D2PC_MB_COPY(U, N)                                                                                              \
  hipLaunchKernelGGL((k_membench_copy<U, N>), dim3(grid), dim3(256), 0, stream, static_cast<const v4f *>(src), static_cast<v4f *>(dst), n16)
  switch (unroll * 2 + (nt ? 1 : 0)) {
    case 2: D2PC_MB_COPY(1, false); break;
    case 3: D2PC_MB_COPY(1, true); break;
    case 4: D2PC_MB_COPY(2, false); break;
    case 5: D2PC_MB_COPY(2, true); break;
    case 8: D2PC_MB_COPY(4, false); break;
    case 9: D2PC_MB_COPY(4, true); break;
    default: return hipErrorInvalidValue;
  }
#undef D2PC_MB_COPY
  return hipGetLastError();
}

}  // namespace d2pc
