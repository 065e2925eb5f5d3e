// graph_memset_repro.hip -- does a hipMemsetAsync captured into a hipGraph zero its buffer on EVERY replay?
//
// Round 1 of this repo zeroed the compaction state with hipMemsetAsync; captured into a graph, the second and later
// replays found stale tickets and a stale timeout flag.  Round 2 replaced the memset with a kernel (k_state_clear) and
// attributed the failure to the runtime's memset node without isolating it.  This program isolates it: the captured
// sequence is exactly [hipMemsetAsync(buf, 0, bytes)] -> [kernel that counts the non-zero words it finds, then
// dirties every word] -> [kernel that advances a replay counter], replayed several times back to back.
// A correct memset node means every replay finds 0 non-zero words.  It also prints the parameters the runtime
// recorded for the memset node.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_memset_repro tools/graph_memset_repro.hip
//   tools/graph_memset_repro [bytes ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                      \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      printf("HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #x);       \
      exit(1);                                                                                     \
    }                                                                                              \
  } while (0)

constexpr int kMaxReplays = 8;

__global__ void k_check_and_dirty(uint32_t *buf, uint32_t n_words, unsigned long long *nonzero, const uint32_t *replay) {
  const uint32_t r = *replay < kMaxReplays ? *replay : kMaxReplays - 1;
  uint32_t bad = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
    // agent-scope load: what the memory system holds, not a line this CU may have cached from an earlier replay
    bad += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  }
  if (bad) atomicAdd(nonzero + r, (unsigned long long)bad);
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x)
    buf[i] = 0xA5A50000u | (r + 1u);
}

// the same check with PLAIN loads (what a kernel that trusts the memset would do)
__global__ void k_check_plain_and_dirty(uint32_t *buf, uint32_t n_words, unsigned long long *nonzero, const uint32_t *replay) {
  const uint32_t r = *replay < kMaxReplays ? *replay : kMaxReplays - 1;
  uint32_t bad = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) bad += buf[i] != 0u;
  if (bad) atomicAdd(nonzero + r, (unsigned long long)bad);
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x)
    buf[i] = 0xA5A50000u | (r + 1u);
}

__global__ void k_next(uint32_t *replay) { *replay += 1u; }

static void print_memset_nodes(hipGraph_t g) {
  size_t n = 0;
  CK(hipGraphGetNodes(g, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  CK(hipGraphGetNodes(g, nodes.data(), &n));
  for (hipGraphNode_t nd : nodes) {
    hipGraphNodeType t;
    CK(hipGraphNodeGetType(nd, &t));
    if (t == hipGraphNodeTypeMemset) {
      hipMemsetParams p;
      CK(hipGraphMemsetNodeGetParams(nd, &p));
      printf("    memset node: dst %p elementSize %u width %zu height %zu pitch %zu value %u\n", p.dst, p.elementSize,
             p.width, p.height, p.pitch, p.value);
    }
  }
  printf("    graph has %zu nodes\n", n);
}

static int run(size_t bytes, bool plain_loads, hipStreamCaptureMode mode, const char *mode_name, int replays) {
  uint32_t *buf = nullptr, *replay = nullptr;
  unsigned long long *nonzero = nullptr;
  const size_t alloc = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);  // the library allocates whole MiB
  CK(hipMalloc(&buf, alloc));
  CK(hipMalloc(&replay, 4));
  CK(hipMalloc(&nonzero, sizeof(unsigned long long) * kMaxReplays));
  CK(hipMemset(buf, 0xFF, alloc));
  CK(hipMemset(replay, 0, 4));
  CK(hipMemset(nonzero, 0, sizeof(unsigned long long) * kMaxReplays));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const uint32_t n_words = uint32_t(bytes / 4);
  const uint32_t grid = (n_words + 255) / 256 < 1024 ? (n_words + 255) / 256 : 1024;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, mode));
  CK(hipMemsetAsync(buf, 0, bytes, s));
  if (plain_loads) hipLaunchKernelGGL(k_check_plain_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
  else hipLaunchKernelGGL(k_check_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
  hipLaunchKernelGGL(k_next, dim3(1), dim3(1), 0, s, replay);
  CK(hipStreamEndCapture(s, &g));
  print_memset_nodes(g);
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  unsigned long long h[kMaxReplays];
  CK(hipMemcpy(h, nonzero, sizeof h, hipMemcpyDeviceToHost));
  int bad = 0;
  printf("  %zu bytes, %s loads, capture mode %s:", bytes, plain_loads ? "plain" : "agent-scope", mode_name);
  for (int r = 0; r < replays; ++r) {
    printf(" replay %d: %llu non-zero", r, h[r]);
    bad += h[r] != 0;
  }
  printf("  => %s\n", bad ? "STALE" : "ok");
  // control: the same sequence eagerly
  CK(hipMemset(replay, 0, 4));
  CK(hipMemset(nonzero, 0, sizeof(unsigned long long) * kMaxReplays));
  for (int r = 0; r < replays; ++r) {
    CK(hipMemsetAsync(buf, 0, bytes, s));
    if (plain_loads) hipLaunchKernelGGL(k_check_plain_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
    else hipLaunchKernelGGL(k_check_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
    hipLaunchKernelGGL(k_next, dim3(1), dim3(1), 0, s, replay);
  }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h, nonzero, sizeof h, hipMemcpyDeviceToHost));
  int bad_eager = 0;
  for (int r = 0; r < replays; ++r) bad_eager += h[r] != 0;
  printf("    eager control: %s\n", bad_eager ? "STALE" : "ok");
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipStreamDestroy(s));
  CK(hipFree(buf));
  CK(hipFree(replay));
  CK(hipFree(nonzero));
  return bad + bad_eager;
}

int main(int argc, char **argv) {
  // 9280 = the round-1 failure's state size (64-byte header + 4 frames x 2304 bytes: 640x480, border 40, 2048-pixel
  // tiles); 9284 / 9288: 4 and 8 over a multiple of 16; two large states (16 x 4K, 32 x 1080p)
  std::vector<size_t> sizes = {9280, 9284, 9288, 64 + 16 * 62208, 64 + 32 * 15104, size_t(8) << 20};
  if (argc > 1) {
    sizes.clear();
    for (int i = 1; i < argc; ++i) sizes.push_back(size_t(atoll(argv[i])));
  }
  int bad = 0;
  for (size_t b : sizes) {
    bad += run(b, false, hipStreamCaptureModeGlobal, "global", 4);
    bad += run(b, true, hipStreamCaptureModeGlobal, "global", 4);
    bad += run(b, true, hipStreamCaptureModeRelaxed, "relaxed", 4);
  }
  printf("%s\n", bad ? "RESULT: a captured memset left non-zero words behind" : "RESULT: every replay saw zeroed memory");
  return 0;
}
