// d2pc_membench.hip -- two streaming kernels that calibrate THIS device in the same run as the benchmark:
// a plain fill and a plain copy, 16 bytes per lane, every wave instruction one contiguous 1-KiB piece (the
// store shape of k_reproject_pack).  bench.py reports their rates as roofline.device_fill_GBs /
// device_copy_GBs, so that a kernel's fraction of the 8 TB/s specification can also be read against what
// the device it ran on gives a kernel that does nothing else.  No counterpart in the reference.
#include "d2pc_launch.hpp"

namespace d2pc {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kMbUnroll = 4;  // independent 16-byte accesses per thread and step

// A block owns runs of kMbUnroll KiB-per-wave pieces: thread t of the block touches element
// base + j*256 + t for j < kMbUnroll, so each wave instruction covers 1 KiB and the block's step 16 KiB.
__global__ __launch_bounds__(256) void k_membench_fill(v4f *__restrict__ dst, uint64_t n16) {
  const uint64_t step = uint64_t(gridDim.x) * 256u * kMbUnroll;
  const v4f v = {1.0f, 2.0f, 3.0f, 1.0f};
  for (uint64_t base = uint64_t(blockIdx.x) * 256u * kMbUnroll; base < n16; base += step) {
#pragma unroll
    for (int j = 0; j < kMbUnroll; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      if (i < n16) dst[i] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_membench_copy(const v4f *__restrict__ src, v4f *__restrict__ dst, uint64_t n16) {
  const uint64_t step = uint64_t(gridDim.x) * 256u * kMbUnroll;
  for (uint64_t base = uint64_t(blockIdx.x) * 256u * kMbUnroll; base < n16; base += step) {
    v4f v[kMbUnroll];
#pragma unroll
    for (int j = 0; j < kMbUnroll; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      v[j] = i < n16 ? src[i] : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < kMbUnroll; ++j) {
      const uint64_t i = base + uint64_t(j) * 256u + threadIdx.x;
      if (i < n16) dst[i] = v[j];
    }
  }
}

hipError_t launch_membench_fill(void *dst, size_t bytes, uint32_t blocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_membench_fill, dim3(blocks), dim3(256), 0, stream, static_cast<v4f *>(dst), uint64_t(bytes / 16));
  return hipGetLastError();
}

hipError_t launch_membench_copy(const void *src, void *dst, size_t bytes, uint32_t blocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_membench_copy, dim3(blocks), dim3(256), 0, stream, static_cast<const v4f *>(src),
                     static_cast<v4f *>(dst), uint64_t(bytes / 16));
  return hipGetLastError();
}

}  // namespace d2pc
