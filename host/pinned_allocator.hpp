// pinned_allocator.hpp -- a std::allocator over d2pc_host_alloc / d2pc_host_free (include/d2pc.h), so that
// the byte vector a PointCloud2 message publishes (output.data, cpp:84-85) can itself be the page-locked
// buffer the kernels write into: SURVEY.md section 8(f) #2, "zero-copy publish".  ROS 1 message types are
// templated on their container allocator (sensor_msgs::PointCloud2_<Alloc>), so the node publishes
// sensor_msgs::PointCloud2_<d2pc::PinnedAllocator<void>> and nothing else changes.
//
// The reference allocates a fresh message per callback (cpp:48,63,84); pinning memory per call would cost far
// more than the callback itself, so freed blocks are kept in a small per-process cache keyed by size and
// handed out again (camera frames have ONE cloud size).  Small requests -- strings, the three PointFields --
// go to malloc: only the payload is worth pinning.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../include/d2pc.h"

namespace d2pc {

class PinnedCache {
 public:
  static constexpr size_t kPinThreshold = 64 * 1024;  // bytes; smaller blocks come from malloc
  static PinnedCache &instance() {
    static PinnedCache c;
    return c;
  }
  void *get(size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second == bytes) {
          void *p = free_[i].first;
          free_.erase(free_.begin() + long(i));
          return p;
        }
    }
    return d2pc_host_alloc(bytes);
  }
  void put(void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    if (free_.size() >= kMaxCached) {  // evict the oldest: a camera that changed its resolution
      d2pc_host_free(free_.front().first);
      free_.erase(free_.begin());
    }
    free_.emplace_back(p, bytes);
  }
  ~PinnedCache() {
    for (auto &b : free_) d2pc_host_free(b.first);
  }

 private:
  static constexpr size_t kMaxCached = 8;
  std::mutex mu_;
  std::vector<std::pair<void *, size_t>> free_;
};

template <class T>
struct PinnedAllocator {
  typedef T value_type;
  PinnedAllocator() = default;
  template <class U>
  PinnedAllocator(const PinnedAllocator<U> &) {}
  template <class U>
  struct rebind {
    typedef PinnedAllocator<U> other;
  };
  T *allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    void *p = bytes >= PinnedCache::kPinThreshold ? PinnedCache::instance().get(bytes) : std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= PinnedCache::kPinThreshold) PinnedCache::instance().put(p, bytes);
    else std::free(p);
  }
  // vector::resize(n) value-initialises: a 4.3 MB memset per frame for bytes the kernels overwrite anyway
  // (pcl::toROSMsg pays it at cpp:85).  Default-initialise instead -- trivial types are left as they are.
  template <class U>
  void construct(U *p) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... Args>
  void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
  template <class U>
  bool operator==(const PinnedAllocator<U> &) const { return true; }
  template <class U>
  bool operator!=(const PinnedAllocator<U> &) const { return false; }
};
template <>
struct PinnedAllocator<void> {  // what a ROS message template is instantiated with
  typedef void value_type;
  template <class U>
  struct rebind {
    typedef PinnedAllocator<U> other;
  };
};

}  // namespace d2pc
