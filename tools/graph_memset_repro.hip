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

// the single pass's own access kinds: agent-scope (sc1) loads, returning atomicAdds on some words, agent-scope
// atomic stores and fetch-adds on others
__global__ void k_check_and_dirty_atomics(uint32_t *buf, uint32_t n_words, unsigned long long *nonzero, const uint32_t *replay) {
  const uint32_t r = *replay < kMaxReplays ? *replay : kMaxReplays - 1;
  uint32_t bad = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
    uint32_t v;
    if ((i & 63u) == 0) v = atomicAdd(buf + i, 1u);  // a ticket word
    else v = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bad += v != 0u;
  }
  if (bad) atomicAdd(nonzero + r, (unsigned long long)bad);
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
    if ((i & 3u) == 1) __hip_atomic_fetch_add(buf + i, 0x10001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(buf + i, 0xA5A50000u | (r + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A longer-running check: phase 1 counts non-zero words and writes a marker, then the thread idles for ~50 us and
// re-reads its own markers.  A marker that reads zero again was WIPED while the kernel ran, i.e. the memset node was
// not ordered before this kernel node (reported in nonzero[kMaxReplays + r]).
__global__ void k_check_dirty_recheck(uint32_t *buf, uint32_t n_words, unsigned long long *nonzero, const uint32_t *replay) {
  const uint32_t r = *replay < kMaxReplays ? *replay : kMaxReplays - 1;
  uint32_t bad = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) {
    bad += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    __hip_atomic_store(buf + i, 0xA5A50000u | (r + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (bad) atomicAdd(nonzero + r, (unsigned long long)bad);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 5000ull) __builtin_amdgcn_s_sleep(32);  // 50 us at 100 MHz
  uint32_t wiped = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x)
    wiped += __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (0xA5A50000u | (r + 1u));
  if (wiped) atomicAdd(nonzero + kMaxReplays + r, (unsigned long long)wiped);
}

__global__ void k_fill32(uint32_t *p, uint32_t n, uint32_t v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
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

enum Flavour { AGENT_LOADS = 0, PLAIN_LOADS = 1, ATOMICS = 2, RECHECK = 3 };
static void launch_check(int flavour, uint32_t grid, hipStream_t s, uint32_t *buf, uint32_t n_words, unsigned long long *nonzero,
                         uint32_t *replay) {
  if (flavour == PLAIN_LOADS) hipLaunchKernelGGL(k_check_plain_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
  else if (flavour == RECHECK) hipLaunchKernelGGL(k_check_dirty_recheck, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
  else if (flavour == ATOMICS) hipLaunchKernelGGL(k_check_and_dirty_atomics, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
  else hipLaunchKernelGGL(k_check_and_dirty, dim3(grid), dim3(256), 0, s, buf, n_words, nonzero, replay);
}

// host_reads: a synchronous 64-byte D2H copy of the buffer's head between replays (what d2pc_check_async_error does);
// auto_free: instantiate with hipGraphInstantiateFlagAutoFreeOnLaunch (what torch.cuda.CUDAGraph does)
static int run(size_t bytes, int plain_loads, hipStreamCaptureMode mode, const char *mode_name, int replays,
               bool host_reads = false, bool auto_free = false, bool replay_on_null_stream = false, bool torch_like = false,
               bool eager_before = false) {
  uint32_t *buf = nullptr, *replay = nullptr;
  unsigned long long *nonzero = nullptr;
  const size_t alloc = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);  // the library allocates whole MiB
  CK(hipMalloc(&buf, alloc));
  CK(hipMalloc(&replay, 4));
  CK(hipMalloc(&nonzero, sizeof(unsigned long long) * 2 * kMaxReplays));
  uint32_t *other = nullptr;  // another small device buffer (the probe's `counts`)
  CK(hipMalloc(&other, 256));
  CK(hipMemset(buf, 0xFF, alloc));
  CK(hipMemset(replay, 0, 4));
  CK(hipMemset(nonzero, 0, sizeof(unsigned long long) * 2 * kMaxReplays));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const uint32_t n_words = uint32_t(bytes / 4);
  const uint32_t grid = (n_words + 255) / 256 < 1024 ? (n_words + 255) / 256 : 1024;
  hipGraph_t g;
  hipGraphExec_t ge;
  if (eager_before) {  // the library's warm-up launch: the same memset + kernel, eagerly, on the NULL stream, before the capture
    CK(hipMemsetAsync(buf, 0, bytes, nullptr));
    launch_check(plain_loads, grid, nullptr, buf, n_words, nonzero, replay);
    CK(hipDeviceSynchronize());
    CK(hipMemset(nonzero, 0, sizeof(unsigned long long) * 2 * kMaxReplays));
  }
  CK(hipStreamBeginCapture(s, mode));
  CK(hipMemsetAsync(buf, 0, bytes, s));
  launch_check(plain_loads, grid, s, buf, n_words, nonzero, replay);
  hipLaunchKernelGGL(k_next, dim3(1), dim3(1), 0, s, replay);
  CK(hipStreamEndCapture(s, &g));
  print_memset_nodes(g);
  if (auto_free) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
  else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipStream_t ls = replay_on_null_stream ? nullptr : s;  // captured on `s`, launched on the legacy default stream
  for (int r = 0; r < replays; ++r) {
    if (torch_like) hipLaunchKernelGGL(k_fill32, dim3(1), dim3(64), 0, ls, other, 6u, 0u);  // b.counts.fill_(0) on the launch stream
    CK(hipGraphLaunch(ge, ls));
    if (host_reads) {
      char head[64];
      CK(hipDeviceSynchronize());
      if (torch_like) {  // counts.cpu(): a D2H copy of another buffer into pageable memory, on the launch stream
        uint32_t c[6];
        CK(hipMemcpyAsync(c, other, sizeof c, hipMemcpyDeviceToHost, ls));
        CK(hipStreamSynchronize(ls));
      }
      CK(hipMemcpy(head, buf, sizeof head, hipMemcpyDeviceToHost));
    }
  }
  CK(hipDeviceSynchronize());
  unsigned long long h[2 * kMaxReplays];
  CK(hipMemcpy(h, nonzero, sizeof h, hipMemcpyDeviceToHost));
  int bad = 0;
  printf("  %zu bytes, %s, capture mode %s%s%s:", bytes,
         plain_loads == PLAIN_LOADS ? "plain loads" : plain_loads == ATOMICS ? "atomics + agent-scope stores" : "agent-scope loads",
         mode_name, host_reads ? ", D2H read between replays" : "", auto_free ? ", AutoFreeOnLaunch" : "");
  if (replay_on_null_stream) printf(" [replayed on the NULL stream]");
  if (torch_like) printf(" [fill kernel before, D2H of another buffer after each replay]");
  if (eager_before) printf(" [eager memset + kernel on the NULL stream before the capture]");
  for (int r = 0; r < replays; ++r) {
    printf(" replay %d: %llu non-zero", r, h[r]);
    if (plain_loads == RECHECK) printf(" %llu wiped", h[kMaxReplays + r]);
    bad += h[r] != 0 || (plain_loads == RECHECK && h[kMaxReplays + r] != 0);
  }
  printf("  => %s\n", bad ? "STALE" : "ok");
  // control: the same sequence eagerly
  CK(hipMemset(replay, 0, 4));
  CK(hipMemset(nonzero, 0, sizeof(unsigned long long) * 2 * kMaxReplays));
  for (int r = 0; r < replays; ++r) {
    CK(hipMemsetAsync(buf, 0, bytes, s));
    launch_check(plain_loads, grid, s, buf, n_words, nonzero, replay);
    hipLaunchKernelGGL(k_next, dim3(1), dim3(1), 0, s, replay);
  }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h, nonzero, sizeof h, hipMemcpyDeviceToHost));
  CK(hipFree(other));
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
    bad += run(b, AGENT_LOADS, hipStreamCaptureModeGlobal, "global", 4);
    bad += run(b, PLAIN_LOADS, hipStreamCaptureModeGlobal, "global", 4);
    bad += run(b, PLAIN_LOADS, hipStreamCaptureModeRelaxed, "relaxed", 4);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, true, false);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, true, true);
    bad += run(b, AGENT_LOADS, hipStreamCaptureModeGlobal, "global", 4, false, true);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, true, true, true);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, false, true, true);
    bad += run(b, PLAIN_LOADS, hipStreamCaptureModeGlobal, "global", 4, true, false, true);
    bad += run(b, RECHECK, hipStreamCaptureModeGlobal, "global", 4, true, false, true, true);
    bad += run(b, RECHECK, hipStreamCaptureModeGlobal, "global", 4, true, true, true, true);
    bad += run(b, RECHECK, hipStreamCaptureModeGlobal, "global", 4, true, false, false, true);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, true, true, true, true);
    bad += run(b, RECHECK, hipStreamCaptureModeGlobal, "global", 4, true, false, true, true, true);
    bad += run(b, ATOMICS, hipStreamCaptureModeGlobal, "global", 4, true, true, true, true, true);
    bad += run(b, AGENT_LOADS, hipStreamCaptureModeGlobal, "global", 4, true, false, true, false, true);
  }
  printf("%s\n", bad ? "RESULT: a captured memset left non-zero words behind" : "RESULT: every replay saw zeroed memory");
  return 0;
}
