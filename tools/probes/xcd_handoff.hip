// Which store / load pairs hand a word from one block to another INSIDE one XCD (through its L2), and how long does it take?
//   hipcc --offload-arch=gfx950 -O2 -o xcd_handoff xcd_handoff.hip && ./xcd_handoff
// Two blocks that find themselves on XCD 0 (HW_REG_XCC_ID) pair up; the producer waits ~30 us, stamps the 100 MHz
// clock into a word with store kind S; the consumer polls the word with load kind L (bounded) and reports the clock
// difference when it sees it.  A second pair on XCDs 0 and 1 shows what the same pair does ACROSS XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
struct Shared {
  unsigned long long flag;  // the word handed over (0 = not yet)
  unsigned long long pad0[15];
  unsigned int arrivals[2];  // blocks seen on XCD 0 / on the consumer's XCD
  unsigned int pad1[30];
  unsigned long long seen_at, written_at, polls, timed_out;
};
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }
template <int S>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
  using g64 = __attribute__((address_space(1))) unsigned long long;
  if (S == 0) *(volatile unsigned long long *)p = v;
  if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (S == 2) __hip_atomic_store((g64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (S == 3) __hip_atomic_fetch_add((g64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (S == 4) __hip_atomic_store((g64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (S == 5) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int L>
__device__ __forceinline__ unsigned long long get(unsigned long long *p) {
  using g64 = __attribute__((address_space(1))) unsigned long long;
  unsigned long long v = 0;
  if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 4) v = __hip_atomic_load((g64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (L == 5) v = __hip_atomic_fetch_add((g64 *)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (L == 6) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 7) {  // the SCALAR path (scalar data cache -> L2), bypassing the scalar cache
    unsigned long long sv;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(p) : "memory");
    v = sv;
  }
  if (L == 8) {  // scalar path, scalar cache invalidated first, no glc
    unsigned long long sv;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(p) : "memory");
    v = sv;
  }
  return v;
}
template <int S, int L>
__global__ void k(Shared *sh, int consumer_xcc) {
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  if (threadIdx.x != 0) return;
  // arrival order on XCD 0 (and, for a consumer elsewhere, on its XCD): ONE atomic per block
  unsigned int a0 = ~0u, a1 = ~0u;
  if (xcc == 0u) a0 = atomicAdd(&sh->arrivals[0], 1u);
  else if (xcc == (unsigned)consumer_xcc) a1 = atomicAdd(&sh->arrivals[1], 1u);
  if (a0 == 0u) {  // the producer: first block on XCD 0
    const unsigned long long t0 = now();
    while (now() - t0 < 3000ull) __builtin_amdgcn_s_sleep(8);  // 30 us: the consumer has read the line (empty) by then
    const unsigned long long t = now();
    put<S>(&sh->flag, t);
    sh->written_at = t;
    return;
  }
  // the consumer: on XCD 0 the second arrival, elsewhere the first
  if ((consumer_xcc == 0 && a0 == 1u) || (consumer_xcc != 0 && a1 == 0u)) {
    const unsigned long long t0 = now();
    unsigned long long polls = 0, v = 0;
    while ((v = get<L>(&sh->flag)) == 0ull) {
      ++polls;
      if (now() - t0 > 200000ull) {  // 2 ms
        sh->timed_out = 1;
        break;
      }
    }
    sh->seen_at = now();
    sh->polls = polls;
  }
}
template <int S, int L>
void run(Shared *d, int consumer_xcc, const char *sname, const char *lname) {
  hipMemset(d, 0, sizeof(Shared));
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<S, L>), dim3(64), dim3(64), 0, 0, d, consumer_xcc);
  hipDeviceSynchronize();
  Shared h;
  hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  if (h.timed_out || h.written_at == 0 || h.seen_at == 0)
    printf("  %-28s -> %-34s : NEVER SEEN in 2 ms (%llu polls)\n", sname, lname, h.polls);
  else
    printf("  %-28s -> %-34s : seen %6.2f us after the store (%llu polls before)\n", sname, lname,
           (double)((long long)h.seen_at - (long long)h.written_at) * 0.01, h.polls);
}
#define ROW(S, SN)                                                \
  run<S, 0>(d, cx, SN, "plain load");                             \
  run<S, 1>(d, cx, SN, "load sc0");                               \
  run<S, 2>(d, cx, SN, "buffer_inv sc0 + load sc0");              \
  run<S, 3>(d, cx, SN, "buffer_inv sc1 + plain load");            \
  run<S, 4>(d, cx, SN, "load sc1 (agent scope)");                 \
  run<S, 5>(d, cx, SN, "atomic add 0, returning (no sc1)");       \
  run<S, 6>(d, cx, SN, "load nt");                                \
  run<S, 7>(d, cx, SN, "s_load glc (scalar path)");               \
  run<S, 8>(d, cx, SN, "s_dcache_inv + s_load");
int main() {
  Shared *d;
  hipMalloc(&d, sizeof(Shared));
  for (int cx : {0, 1}) {
    printf("consumer on XCD %d, producer on XCD 0:\n", cx);
    ROW(0, "plain store")
    ROW(1, "store sc0")
    ROW(3, "atomic add (no sc1)")
    ROW(4, "store sc1 (agent scope)")
    ROW(5, "store nt")
  }
  hipFree(d);
  return 0;
}
