// CPU test of host/pinned_allocator.hpp: the two ABI functions it wraps are replaced by counting stand-ins
// (malloc/free), so the allocator's own logic -- size threshold, cache reuse, eviction, default-initialising
// construct -- runs without a GPU and under ASan/UBSan.  Built and run by tests/test_host_cpp.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_allocs = 0, g_frees = 0;
extern "C" void *d2pc_host_alloc(size_t bytes) { ++g_allocs; return std::malloc(bytes); }
extern "C" void d2pc_host_free(void *p) { ++g_frees; std::free(p); }

#include "../../host/pinned_allocator.hpp"

#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

int main() {
  typedef std::vector<uint8_t, d2pc::PinnedAllocator<uint8_t>> Bytes;
  const size_t big = 4300800;  // a 752x480 cloud
  {
    Bytes a;
    a.resize(big);            // first frame: one pinned allocation
    CHECK(g_allocs == 1);
    std::memset(a.data(), 0xAB, big);
  }                            // freed into the cache, not to the runtime
  CHECK(g_frees == 0);
  {
    Bytes b;
    b.resize(big);            // second frame: the cached block, and NOT value-initialised (the kernels overwrite it)
    CHECK(g_allocs == 1);
    CHECK(b[0] == 0xAB && b[big - 1] == 0xAB);
    Bytes c;
    c.resize(big);            // two messages alive at once: a second block
    CHECK(g_allocs == 2);
  }
  {
    Bytes small(1000, 7);     // strings, PointFields: malloc, never pinned
    CHECK(g_allocs == 2 && small[999] == 7);
    std::vector<std::string, d2pc::PinnedAllocator<std::string>> names(3, std::string("xyz"));  // non-trivial types are constructed
    CHECK(names[2] == "xyz");
  }
  for (size_t i = 0; i < 12; ++i) {  // a camera that keeps changing resolution: the cache stays bounded
    Bytes d;
    d.resize(big + 4096 * (i + 1));
  }
  CHECK(g_allocs == 14);
  CHECK(g_frees >= 14 - 8 - 2 + 0 && g_frees <= 14);  // at most 8 blocks cached
  std::printf("pinned allocator ok: %d pinned allocations, %d returned to the runtime so far\n", g_allocs, g_frees);
  return 0;
}
