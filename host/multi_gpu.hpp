// multi_gpu.hpp -- the native multi-GPU form of the node (SURVEY.md section 2.3 C1/C2, section 7 step 6,
// section 8(e); BASELINE.json north_star: "frames shard trivially one-per-GPU ... via a RCCL-over-xGMI broadcast
// of Q and a per-rank frame queue").  The reference deploys as C++ processes (src/disparity_to_point_cloud_node.cpp:
// 46-52: ros::init, construct, ros::spin); a maintainer with N cameras on one node has no Python.  This is that
// deployment without ROS, in ONE process:
//
//   d2pc_replay --gpus N [--devices a,b,..] [--device D] [--share-device] [--frames F] [--width W --height H]
//               [--encoding mono8|mono16] [--median K] [--compact] [--depth P] [--in frame.raw] [--out prefix] [name=value ...]
//               (name=value: the private parameters fx_ fy_ cx_ cy_ base_line_, and reproject_form=0|24|4)
//
//   * ncclCommInitAll over the N devices (RCCL; xGMI between the GPUs of a node)
//   * rank 0 owns the calibration (hpp:84-104: the private parameters -> Q): it packs the 136-byte blob
//     (d2pc_calib_pack) and ncclBroadcast carries it, on each device's stream, into every device's memory;
//     every rank -- rank 0 included -- configures its context from the bytes it RECEIVED (d2pc_import_calibration)
//   * one host thread + one d2pc_ctx (cfg.device_id) + one d2pc_pipeline_* queue per GPU: the per-rank frame
//     queue (subscribe("/disparity", 1, ...) of hpp:77-78 per camera); nothing crosses GPUs per frame
//   * the ranks' {frames, pixels, points, busy ns} are summed with ncclAllReduce for the one report line
//
// RCCL is resolved at run time (dlopen("librccl.so.1")): only this mode needs it; libd2pc.so and the
// single-GPU commands of the harness stay free of it.  Never re-executes the process.
//
// REHEARSAL (--share-device): all N ranks on ONE device (the first of --devices, default 0) -- what a one-GPU box
// allows.  RCCL refuses two ranks on a device, so the three collectives go through an in-process loopback that fills
// the same function-pointer table dlsym fills (Loopback below: device-to-device copies for the broadcast, a host-side
// sum for the all-reduce, every rank's part on that rank's stream).  Everything else is the real thing: N rank
// threads, N contexts, N frame queues running concurrently, each configured from the blob it RECEIVED.  The report
// line says "rehearsal_shared_device": it is a concurrency rehearsal of the deployment, never a scaling measurement.
#pragma once
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../include/d2pc.h"

namespace d2pc_multi {

struct Options {
  int gpus = 0;                 // 0 = not the multi-GPU mode
  std::vector<int> devices;     // HIP ordinals, one per rank
  int frames = 32;              // frames per rank
  int width = 752, height = 480;  // hpp:102-103: the reference's native size
  std::string encoding = "mono8";
  int median = 11;              // cpp:57
  bool compact = false;
  int depth = 3;                // frames in flight per rank
  std::string in_path, out_prefix;
  double fx = 714.24, fy = 713.5, cx = 376, cy = 240, base_line = 0.09;  // hpp:66-71
  int reproject_form = 0;       // d2pc_set_reproject_form on every rank (0, 24, 4)
  bool share_device = false;    // rehearsal: every rank on devices[0], collectives through the in-process loopback
  std::string error;            // non-empty: bad command line
};

inline bool parse_int(const char *s, int lo, int hi, int *out) {
  char *end = nullptr;
  const long v = strtol(s, &end, 10);
  if (!s[0] || *end || v < lo || v > hi) return false;
  *out = int(v);
  return true;
}

// Parses argv[1..]; `gpus` stays 0 when neither --gpus nor --device is given.
inline Options parse(int argc, char **argv) {
  Options o;
  bool multi = false;
  int single_device = -1;
  auto need = [&](int i) -> const char * { return i + 1 < argc ? argv[i + 1] : nullptr; };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto bad = [&](const std::string &why) { if (o.error.empty()) o.error = a + ": " + why; };
    if (a == "--gpus") {
      multi = true;
      if (!need(i) || !parse_int(argv[++i], 1, 64, &o.gpus)) bad("expects a number of GPUs in 1..64");
    } else if (a == "--device") {
      multi = true;
      if (!need(i) || !parse_int(argv[++i], 0, 1023, &single_device)) bad("expects a HIP device ordinal");
    } else if (a == "--devices") {
      multi = true;
      const char *v = need(i);
      if (!v) { bad("expects a comma-separated list of HIP device ordinals"); continue; }
      ++i;
      std::string tok;
      for (const char *p = v;; ++p) {
        if (*p == ',' || !*p) {
          int d = -1;
          if (!parse_int(tok.c_str(), 0, 1023, &d)) bad("bad device list");
          else o.devices.push_back(d);
          tok.clear();
          if (!*p) break;
        } else {
          tok.push_back(*p);
        }
      }
    } else if (a == "--frames") {
      if (!need(i) || !parse_int(argv[++i], 1, 1000000, &o.frames)) bad("expects a frame count >= 1");
    } else if (a == "--width") {
      if (!need(i) || !parse_int(argv[++i], 1, 65535, &o.width)) bad("expects a width");
    } else if (a == "--height") {
      if (!need(i) || !parse_int(argv[++i], 1, 65535, &o.height)) bad("expects a height");
    } else if (a == "--median") {
      if (!need(i) || !parse_int(argv[++i], 0, 11, &o.median) || (o.median > 1 && o.median % 2 == 0)) bad("expects 0, 1 or an odd size in 3..11");
    } else if (a == "--depth") {
      if (!need(i) || !parse_int(argv[++i], 1, 8, &o.depth)) bad("expects a pipeline depth in 1..8");
    } else if (a == "--encoding") {
      if (!need(i)) { bad("expects mono8 or mono16"); continue; }
      o.encoding = argv[++i];
      if (o.encoding != "mono8" && o.encoding != "mono16") bad("expects mono8 or mono16");
    } else if (a == "--in") {
      if (!need(i)) bad("expects a file"); else o.in_path = argv[++i];
    } else if (a == "--out") {
      if (!need(i)) bad("expects a path prefix"); else o.out_prefix = argv[++i];
    } else if (a == "--compact") {
      o.compact = true;
    } else if (a == "--share-device") {
      multi = true;
      o.share_device = true;
    } else if (a.find('=') != std::string::npos) {  // private parameters, as a launch file would set them (hpp:84-88)
      const size_t eq = a.find('=');
      const std::string k = a.substr(0, eq);
      const double v = atof(a.c_str() + eq + 1);
      if (k == "fx_") o.fx = v;
      else if (k == "fy_") o.fy = v;
      else if (k == "cx_") o.cx = v;
      else if (k == "cy_") o.cy = v;
      else if (k == "base_line_") o.base_line = v;
      else if (k == "reproject_form") o.reproject_form = int(v);
      else bad("unknown parameter");
    } else if (multi) {
      bad("unknown argument");
    }
  }
  if (!multi) {  // the positional single-frame commands of replay_main.cpp: nothing here applies
    o.error.clear();
    return o;
  }
  if (single_device >= 0) {
    if (o.gpus > 1 || !o.devices.empty()) { o.error = "--device selects ONE GPU: do not combine it with --gpus N > 1 or --devices"; }
    o.gpus = 1;
    o.devices.assign(1, single_device);
  }
  if (!o.devices.empty() && o.gpus == 0 && !o.share_device) o.gpus = int(o.devices.size());
  if (o.gpus == 0 && o.error.empty()) o.error = "--gpus N missing";
  if (!o.share_device && !o.devices.empty() && int(o.devices.size()) != o.gpus && o.error.empty()) o.error = "--devices lists another number of GPUs than --gpus";
  if (o.share_device) {  // rehearsal: N ranks on the first listed device (default 0)
    if (single_device >= 0 && o.error.empty()) o.error = "--share-device goes with --gpus N (and optionally --devices D): --device D means one rank";
    const int d = o.devices.empty() ? 0 : o.devices[0];
    if (o.devices.size() > 1 && o.error.empty()) o.error = "--share-device takes ONE device (--devices D)";
    if (o.gpus == 0 && o.error.empty()) o.error = "--gpus N missing";
    if (o.gpus > 16 && o.error.empty()) o.error = "--share-device rehearses at most 16 ranks on a device";
    o.devices.assign(size_t(o.gpus > 0 ? o.gpus : 1), d);
    return o;
  }
  if (o.devices.empty())
    for (int i = 0; i < o.gpus; ++i) o.devices.push_back(i);
  for (size_t i = 0; i < o.devices.size(); ++i)
    for (size_t j = i + 1; j < o.devices.size(); ++j)
      if (o.devices[i] == o.devices[j] && o.error.empty()) o.error = "a device is listed twice (RCCL wants one rank per device)";
  return o;
}

// The RCCL entry points this mode uses, resolved from librccl at run time.
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
  bool load() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      error = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(lib, n);
      if (!p && error.empty()) error = std::string("librccl lacks ") + n;
      return p;
    };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    return error.empty();
  }
  bool load_loopback();  // --share-device: the same table, filled with the in-process loopback (below)
};

// ---- in-process loopback of the three collectives (--share-device) --------------------------------------------
// Same signatures as RCCL's, same call pattern (ncclCommInitAll, then grouped per-rank calls between ncclGroupStart
// and ncclGroupEnd, each rank's call on its own stream).  A "communicator" is a small record; the grouped calls are
// collected and carried out at ncclGroupEnd, when every rank's buffers are known:
//   broadcast   the root's buffer is copied device-to-device into every other rank's, on that rank's stream
//   all-reduce  (ncclUint64, ncclSum) every rank's input is read back, summed on the host, and the sum is copied into
//               every rank's output on its stream
// One world per process (the harness makes one); calls come from the thread that runs d2pc_multi::run.
namespace loopback {
struct Comm {
  int rank = 0, n = 0;
};
struct Call {
  int kind = 0;  // 1 broadcast, 2 all-reduce
  const void *send = nullptr;
  void *recv = nullptr;
  size_t count = 0;
  ncclDataType_t type = ncclUint8;
  int root = 0;
  Comm *comm = nullptr;
  hipStream_t stream = nullptr;
};
struct World {
  std::mutex mu;
  std::vector<Comm *> comms;
  std::vector<Call> pending;
  int depth = 0;
  std::string last_error;
};
inline World &world() {
  static World w;
  return w;
}
inline ncclResult_t fail(const std::string &why) {
  world().last_error = why;
  return ncclInvalidUsage;
}
inline ncclResult_t CommInitAll(ncclComm_t *comms, int n, const int *devs) {
  (void)devs;
  if (!comms || n < 1) return fail("loopback: bad ncclCommInitAll arguments");
  std::lock_guard<std::mutex> lock(world().mu);
  for (int r = 0; r < n; ++r) {
    Comm *c = new Comm{r, n};
    world().comms.push_back(c);
    comms[r] = reinterpret_cast<ncclComm_t>(c);
  }
  return ncclSuccess;
}
inline ncclResult_t CommDestroy(ncclComm_t comm) {
  std::lock_guard<std::mutex> lock(world().mu);
  Comm *c = reinterpret_cast<Comm *>(comm);
  for (Comm *&x : world().comms)
    if (x == c) {
      delete c;
      x = nullptr;
      return ncclSuccess;
    }
  return fail("loopback: unknown communicator");
}
inline ncclResult_t GroupStart() {
  std::lock_guard<std::mutex> lock(world().mu);
  ++world().depth;
  return ncclSuccess;
}
inline size_t type_bytes(ncclDataType_t t) { return t == ncclUint64 || t == ncclInt64 || t == ncclFloat64 ? 8 : t == ncclUint8 || t == ncclInt8 ? 1 : 4; }
inline ncclResult_t flush(World &w) {
  std::vector<Call> calls;
  calls.swap(w.pending);
  if (calls.empty()) return ncclSuccess;
  const int n = calls[0].comm->n;
  if (int(calls.size()) != n) return fail("loopback: a collective needs one call per rank inside the group");
  std::vector<const Call *> by_rank(size_t(n), nullptr);
  for (const Call &c : calls) {
    if (c.kind != calls[0].kind || c.count != calls[0].count || c.type != calls[0].type || c.root != calls[0].root || c.comm->n != n)
      return fail("loopback: the ranks disagree about the collective");
    if (by_rank[size_t(c.comm->rank)]) return fail("loopback: a rank called twice");
    by_rank[size_t(c.comm->rank)] = &c;
  }
  const size_t bytes = calls[0].count * type_bytes(calls[0].type);
  if (calls[0].kind == 1) {
    const Call &root = *by_rank[size_t(calls[0].root)];
    if (hipStreamSynchronize(root.stream) != hipSuccess) return ncclUnhandledCudaError;  // the root's bytes are final
    for (int r = 0; r < n; ++r) {
      const Call &c = *by_rank[size_t(r)];
      if (c.recv != root.send && hipMemcpyAsync(c.recv, root.send, bytes, hipMemcpyDeviceToDevice, c.stream) != hipSuccess)
        return ncclUnhandledCudaError;
    }
    return ncclSuccess;
  }
  if (calls[0].type != ncclUint64) return fail("loopback: the all-reduce rehearsal handles ncclUint64 sums only");
  std::vector<unsigned long long> sum(calls[0].count, 0ull), part(calls[0].count);
  for (int r = 0; r < n; ++r) {
    const Call &c = *by_rank[size_t(r)];
    if (hipStreamSynchronize(c.stream) != hipSuccess || hipMemcpy(part.data(), c.send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    for (size_t i = 0; i < sum.size(); ++i) sum[i] += part[i];
  }
  for (int r = 0; r < n; ++r) {
    const Call &c = *by_rank[size_t(r)];
    // (pageable source: the copy has left the host buffer when the call returns)
    if (hipMemcpyAsync(c.recv, sum.data(), bytes, hipMemcpyHostToDevice, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}
inline ncclResult_t GroupEnd() {
  std::lock_guard<std::mutex> lock(world().mu);
  World &w = world();
  if (w.depth <= 0) return fail("loopback: ncclGroupEnd without ncclGroupStart");
  if (--w.depth > 0) return ncclSuccess;
  return flush(w);
}
inline ncclResult_t enqueue(const Call &c) {
  std::lock_guard<std::mutex> lock(world().mu);
  World &w = world();
  w.pending.push_back(c);
  if (w.depth == 0) return c.comm->n == 1 ? flush(w) : fail("loopback: several ranks in one process call inside a group");
  return ncclSuccess;
}
inline ncclResult_t Broadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || !recv || root < 0 || root >= c->n) return fail("loopback: bad ncclBroadcast arguments");
  Call k;
  k.kind = 1; k.send = send; k.recv = recv; k.count = count; k.type = type; k.root = root; k.comm = c; k.stream = stream;
  return enqueue(k);
}
inline ncclResult_t AllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || !send || !recv || op != ncclSum) return fail("loopback: bad ncclAllReduce arguments (sums only)");
  Call k;
  k.kind = 2; k.send = send; k.recv = recv; k.count = count; k.type = type; k.root = 0; k.comm = c; k.stream = stream;
  return enqueue(k);
}
inline const char *GetErrorString(ncclResult_t r) {
  static thread_local std::string s;
  s = "loopback collective failed (" + std::to_string(int(r)) + "): " + world().last_error;
  return s.c_str();
}
}  // namespace loopback

inline bool Rccl::load_loopback() {
  CommInitAll = &loopback::CommInitAll;
  CommDestroy = &loopback::CommDestroy;
  Broadcast = &loopback::Broadcast;
  AllReduce = &loopback::AllReduce;
  GroupStart = &loopback::GroupStart;
  GroupEnd = &loopback::GroupEnd;
  GetErrorString = &loopback::GetErrorString;
  return true;
}

struct RankResult {
  unsigned long long counters[4] = {0, 0, 0, 0};  // frames, pixels, points, busy nanoseconds
  std::string error;
};

// The seeded frame generator of SURVEY.md section 8(d): mt19937_64 is fully specified by the C++ standard.
inline std::vector<uint8_t> synth_frame(const Options &o, int rank, int frame) {
  const size_t bpp = o.encoding == "mono16" ? 2 : 1;
  std::vector<uint8_t> img(size_t(o.width) * o.height * bpp);
  std::mt19937_64 gen(0xD2C00000ull + 5000ull + uint64_t(rank) * 100000ull + uint64_t(frame));
  for (size_t i = 0; i < img.size(); i += 8) {
    const uint64_t v = gen();
    memcpy(&img[i], &v, img.size() - i < 8 ? img.size() - i : 8);
  }
  return img;
}

// One rank: its own context on its own GPU, its own frame queue.  `blob` is what THIS rank received.
inline void run_rank(const Options &o, int rank, int device, const unsigned char *blob, const std::vector<uint8_t> *fixed_frame,
                     RankResult *res) {
  auto fail = [&](const std::string &what, d2pc_ctx *ctx, int st) {
    res->error = "rank " + std::to_string(rank) + " (device " + std::to_string(device) + "): " + what + ": " +
                 d2pc_status_string(st) + (ctx ? std::string(": ") + d2pc_last_error(ctx) : std::string());
  };
  d2pc_config cfg;
  d2pc_config_init(&cfg);  // border 40 (cpp:70,72); the mode comes with the blob
  cfg.device_id = device;
  d2pc_ctx *ctx = nullptr;
  int st = d2pc_create(&cfg, &ctx);
  if (st != D2PC_OK) return fail("d2pc_create", nullptr, st);
  struct Guard {
    d2pc_ctx *c;
    ~Guard() { d2pc_destroy(c); }
  } guard{ctx};
  if ((st = d2pc_import_calibration(ctx, blob, D2PC_CALIB_BLOB_BYTES)) != D2PC_OK) return fail("d2pc_import_calibration", ctx, st);
  if ((st = d2pc_set_reproject_form(ctx, o.reproject_form)) != D2PC_OK) return fail("reproject_form", ctx, st);
  if ((st = d2pc_pipeline_configure(ctx, o.depth, 1)) != D2PC_OK) return fail("d2pc_pipeline_configure", ctx, st);
  const bool m16 = o.encoding == "mono16";
  d2pc_frame_desc desc;
  memset(&desc, 0, sizeof desc);
  desc.dtype = m16 ? D2PC_DTYPE_MONO16 : D2PC_DTYPE_U8;
  desc.scale = 1.0f / 8.0f;  // cpp:61
  desc.width = o.width;
  desc.height = o.height;
  desc.row_stride_bytes = size_t(o.width) * (m16 ? 2 : 1);
  desc.median_ksize = o.median;
  // the camera's frames: a ring of distinct seeded frames generated BEFORE the clock starts (the generator is not
  // part of the path; a real producer is cv_bridge decoding into the pinned slot)
  std::vector<std::vector<uint8_t>> ring;
  if (!fixed_frame)
    for (int f = 0; f < (o.frames < 8 ? o.frames : 8); ++f) ring.push_back(synth_frame(o, rank, f));
  int in_flight = 0, submitted = 0, collected = 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto collect = [&]() -> bool {
    int slot = -1;
    const void *pts = nullptr;
    size_t n = 0;
    uint64_t tag = 0;
    if ((st = d2pc_pipeline_collect(ctx, &slot, &pts, nullptr, &n, &tag)) != D2PC_OK) {
      fail("d2pc_pipeline_collect", ctx, st);
      return false;
    }
    res->counters[0] += 1;
    res->counters[1] += (unsigned long long)o.width * o.height;
    res->counters[2] += n;
    if (!o.out_prefix.empty() && int(tag) == o.frames - 1) {  // the rank's last cloud, for the test to check against the oracle
      std::ofstream f(o.out_prefix + ".rank" + std::to_string(rank) + ".cloud", std::ios::binary);
      f.write(static_cast<const char *>(pts), std::streamsize(n * 16));
    }
    ++collected;
    --in_flight;
    if ((st = d2pc_pipeline_release(ctx, slot)) != D2PC_OK) {
      fail("d2pc_pipeline_release", ctx, st);
      return false;
    }
    return true;
  };
  while (collected < o.frames) {
    if (submitted < o.frames && in_flight < o.depth) {
      desc.tag = uint64_t(submitted);
      void *hin = nullptr;
      int slot = -1;
      if ((st = d2pc_pipeline_acquire(ctx, &desc, &hin, &slot)) != D2PC_OK) return fail("d2pc_pipeline_acquire", ctx, st);
      // the producer (cv_bridge::toCvCopy, cpp:50) decodes straight into the pinned slot
      const std::vector<uint8_t> &img = fixed_frame ? *fixed_frame : ring[size_t(submitted) % ring.size()];
      memcpy(hin, img.data(), img.size());
      if ((st = d2pc_pipeline_submit(ctx, slot)) != D2PC_OK) return fail("d2pc_pipeline_submit", ctx, st);
      ++submitted;
      ++in_flight;
    } else if (!collect()) {
      return;
    }
  }
  res->counters[3] = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

#define D2PC_MULTI_HIP(call)                                                                  \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "d2pc_replay --gpus: %s failed: %s\n", #call, hipGetErrorString(e_));   \
      return 6;                                                                               \
    }                                                                                         \
  } while (0)
#define D2PC_MULTI_NCCL(call)                                                                 \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      fprintf(stderr, "d2pc_replay --gpus: %s failed: %s\n", #call, rccl.GetErrorString(r_)); \
      return 7;                                                                               \
    }                                                                                         \
  } while (0)

// Exit codes: 0 ok, 2 bad command line, 5 not enough devices, 6 HIP error, 7 RCCL error, 8 a rank failed.
inline int run(const Options &o) {
  if (!o.error.empty()) {
    fprintf(stderr, "d2pc_replay --gpus: %s\n(usage: see host/multi_gpu.hpp)\n", o.error.c_str());
    return 2;
  }
  const int n = o.gpus;
  const int have = d2pc_device_count();  // (counting devices does not initialise one)
  for (int d : o.devices)
    if (d >= have) {
      fprintf(stderr, "d2pc_replay --gpus: device %d requested but this node exposes %d HIP device(s); there is no CPU path\n", d, have);
      return 5;
    }
  std::vector<uint8_t> fixed;
  if (!o.in_path.empty()) {
    std::ifstream f(o.in_path, std::ios::binary);
    if (!f) {
      fprintf(stderr, "d2pc_replay --gpus: cannot read %s\n", o.in_path.c_str());
      return 2;
    }
    fixed.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    const size_t want = size_t(o.width) * o.height * (o.encoding == "mono16" ? 2 : 1);
    if (fixed.size() != want) {
      fprintf(stderr, "d2pc_replay --gpus: %s holds %zu bytes, a %dx%d %s frame has %zu\n", o.in_path.c_str(), fixed.size(), o.width,
              o.height, o.encoding.c_str(), want);
      return 2;
    }
  }
  Rccl rccl;
  if (o.share_device ? !rccl.load_loopback() : !rccl.load()) {
    fprintf(stderr, "d2pc_replay --gpus: %s\n", rccl.error.c_str());
    return 7;
  }
  // rank 0's calibration (hpp:84-104 through the closed form of stereoRectify the host mirror uses)
  double q[16];
  if (d2pc_make_q_flavour(o.fx, o.fy, o.cx, o.cy, o.base_line, 752, 480, D2PC_STEREORECTIFY_CV24, q) != D2PC_OK) {
    fprintf(stderr, "d2pc_replay --gpus: bad calibration parameters\n");
    return 2;
  }
  unsigned char blob0[D2PC_CALIB_BLOB_BYTES];
  if (d2pc_calib_pack(q, 40, o.compact ? D2PC_MODE_COMPACT : D2PC_MODE_PARITY, blob0) != D2PC_OK) return 2;

  // everything the collectives own, released on every way out of this function
  struct Resources {
    const Options &o;
    Rccl &rccl;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<unsigned char *> d_blob;
    std::vector<unsigned long long *> d_cnt;
    Resources(const Options &opt, Rccl &r, size_t n) : o(opt), rccl(r), comms(n, nullptr), streams(n, nullptr), d_blob(n, nullptr), d_cnt(n, nullptr) {}
    ~Resources() {
      for (size_t r = 0; r < comms.size(); ++r) {
        (void)hipSetDevice(o.devices[r]);
        if (d_blob[r]) (void)hipFree(d_blob[r]);
        if (d_cnt[r]) (void)hipFree(d_cnt[r]);
        if (streams[r]) (void)hipStreamDestroy(streams[r]);
        if (comms[r]) (void)rccl.CommDestroy(comms[r]);
      }
    }
  } res(o, rccl, size_t(n));
  std::vector<ncclComm_t> &comms = res.comms;
  std::vector<hipStream_t> &streams = res.streams;
  std::vector<unsigned char *> &d_blob = res.d_blob;
  std::vector<unsigned long long *> &d_cnt = res.d_cnt;
  D2PC_MULTI_NCCL(rccl.CommInitAll(comms.data(), n, o.devices.data()));
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_HIP(hipStreamCreateWithFlags(&streams[size_t(r)], hipStreamNonBlocking));
    D2PC_MULTI_HIP(hipMalloc(reinterpret_cast<void **>(&d_blob[size_t(r)]), D2PC_CALIB_BLOB_BYTES));
    D2PC_MULTI_HIP(hipMalloc(reinterpret_cast<void **>(&d_cnt[size_t(r)]), 4 * sizeof(unsigned long long)));
    // only the root's buffer holds the calibration; every other device starts from a poison pattern
    if (r == 0) D2PC_MULTI_HIP(hipMemcpy(d_blob[0], blob0, sizeof blob0, hipMemcpyHostToDevice));
    else D2PC_MULTI_HIP(hipMemset(d_blob[size_t(r)], 0xEE, D2PC_CALIB_BLOB_BYTES));
    // hipMemset of device memory and hipMemcpy from pageable memory may return before the device has the bytes, and
    // the collectives' streams are non-blocking (they do not wait for the NULL stream): drain the device first
    D2PC_MULTI_HIP(hipDeviceSynchronize());
  }
  // C1: the 136-byte blob from rank 0 into every device's memory, over xGMI
  D2PC_MULTI_NCCL(rccl.GroupStart());
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_NCCL(rccl.Broadcast(d_blob[size_t(r)], d_blob[size_t(r)], D2PC_CALIB_BLOB_BYTES, ncclUint8, 0, comms[size_t(r)],
                                   streams[size_t(r)]));
  }
  D2PC_MULTI_NCCL(rccl.GroupEnd());
  std::vector<std::vector<unsigned char>> blobs(size_t(n), std::vector<unsigned char>(D2PC_CALIB_BLOB_BYTES));
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_HIP(hipStreamSynchronize(streams[size_t(r)]));
    D2PC_MULTI_HIP(hipMemcpy(blobs[size_t(r)].data(), d_blob[size_t(r)], D2PC_CALIB_BLOB_BYTES, hipMemcpyDeviceToHost));
    if (!o.out_prefix.empty()) {
      std::ofstream f(o.out_prefix + ".rank" + std::to_string(r) + ".blob", std::ios::binary);
      f.write(reinterpret_cast<const char *>(blobs[size_t(r)].data()), D2PC_CALIB_BLOB_BYTES);
    }
  }
  // one thread, one context, one frame queue per GPU
  std::vector<RankResult> results(static_cast<size_t>(n));
  std::vector<std::thread> threads;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < n; ++r)
    threads.emplace_back(run_rank, std::cref(o), r, o.devices[size_t(r)], blobs[size_t(r)].data(), fixed.empty() ? nullptr : &fixed,
                         &results[size_t(r)]);
  for (std::thread &t : threads) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int failed = 0;
  for (const RankResult &rr : results)
    if (!rr.error.empty()) {
      fprintf(stderr, "d2pc_replay --gpus: %s\n", rr.error.c_str());
      ++failed;
    }
  // C2: the scaling report's counters, summed over the ranks on the devices
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_HIP(hipMemcpy(d_cnt[size_t(r)], results[size_t(r)].counters, sizeof results[size_t(r)].counters, hipMemcpyHostToDevice));
    D2PC_MULTI_HIP(hipDeviceSynchronize());  // (as above: the all-reduce runs on a non-blocking stream)
  }
  D2PC_MULTI_NCCL(rccl.GroupStart());
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_NCCL(rccl.AllReduce(d_cnt[size_t(r)], d_cnt[size_t(r)], 4, ncclUint64, ncclSum, comms[size_t(r)], streams[size_t(r)]));
  }
  D2PC_MULTI_NCCL(rccl.GroupEnd());
  unsigned long long total[4] = {0, 0, 0, 0};
  for (int r = 0; r < n; ++r) {
    D2PC_MULTI_HIP(hipSetDevice(o.devices[size_t(r)]));
    D2PC_MULTI_HIP(hipStreamSynchronize(streams[size_t(r)]));
    unsigned long long got[4];
    D2PC_MULTI_HIP(hipMemcpy(got, d_cnt[size_t(r)], sizeof got, hipMemcpyDeviceToHost));
    if (r == 0) memcpy(total, got, sizeof got);
    else if (memcmp(total, got, sizeof got) != 0) {
      fprintf(stderr, "d2pc_replay --gpus: rank %d's all-reduced counters differ from rank 0's\n", r);
      ++failed;
    }
  }
  printf("{\"n_gpus\": %d, %s\"devices\": [", n, o.share_device ? "\"rehearsal_shared_device\": true, " : "");
  for (int r = 0; r < n; ++r) printf("%s%d", r ? ", " : "", o.devices[size_t(r)]);
  printf("], \"frames\": %llu, \"pixels\": %llu, \"points\": %llu, \"busy_ns_sum\": %llu, \"wall_s\": %.6f, "
         "\"Mpixels_per_s\": %.1f, \"what\": \"%dx%d %s, median %d, %s, pipeline depth %d, PCIe-inclusive host path; "
         "calibration by ncclBroadcast, counters by ncclAllReduce%s\", \"per_rank_frames\": [",
         total[0], total[1], total[2], total[3], wall, wall > 0 ? double(total[1]) / wall / 1e6 : 0.0, o.width, o.height,
         o.encoding.c_str(), o.median, o.compact ? "compact" : "parity", o.depth,
         o.share_device ? " (REHEARSAL: all ranks on one device, collectives through the in-process loopback)" : "");
  for (int r = 0; r < n; ++r) printf("%s%llu", r ? ", " : "", results[size_t(r)].counters[0]);
  printf("]}\n");
  return failed ? 8 : 0;
}

}  // namespace d2pc_multi
