// SYNTAX-CHECK STUB (see tests/stubs/README.md): declarations only, never linked or run.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; };
void init(int &argc, char **argv, const std::string &name);
void spin();
struct Publisher {
  template <class M> void publish(const M &m) const;
};
struct Subscriber {};
struct NodeHandle {
  explicit NodeHandle(const std::string &ns = std::string());
  bool getParam(const std::string &name, double &v) const;
  template <class T> bool param(const std::string &name, T &v, const T &def) const;
  template <class M> Publisher advertise(const std::string &topic, uint32_t queue, bool latch = false);
  template <class M> Subscriber subscribe(const std::string &topic, uint32_t queue,
                                          const std::function<void(const std::shared_ptr<const M> &)> &cb);
};
}  // namespace ros
