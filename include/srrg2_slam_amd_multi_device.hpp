// srrg2_slam_amd_multi_device.hpp -- ONE process driving G aligner handles (one per device) from host threads.
//
// The reference is a single process: MultiLoopDetectorBruteForce_::compute binds the fixed cloud once and loops over the
// hints (S/registration/loop_detector/multi_loop_detector_brute_force_impl.cpp:63-79), MultiRelocalizer_::compute does the
// same over its candidates (S/registration/relocalization/multi_relocalizer_impl.cpp:74-137).  The alignments are
// independent, so a C++ caller spreads them over the GPUs of a node WITHOUT leaving its process: alignment k goes to handle
// k mod G (the rule of srrg2_multi_gpu_shard_indices and of distributed.py), every handle is driven by its own host
// thread (handles are independent objects: include/srrg2_slam_amd.h "Conventions"), the result records are merged in host
// memory, and the K x SRRG2_RECORD_FLOATS table -- every handle fills its own rows of a zero table, the tables are summed --
// is the one the multi-process path all-reduces over RCCL (north_star's "all-reduce of the final Hessian": H travels in
// the rows).  No communicator is needed for this mode.
//
//   ShardedAligners<A>        K alignments over G handles; setFixed on all of them; record table
//   HostPointShardReducer     the reduction hook of ONE alignment sharded by moving points (srrg2_aligner_set_point_shard)
//                             for G handles of one process: sums / maxima through host memory, one barrier pair per call
//   RcclPointShardReducer     the same hook on RCCL (librccl.so opened at run time, ncclCommInitAll over the devices of the
//                             handles, ncclAllReduce on each aligner's stream: nothing but launches on the hook's path)
//   RcclRecordExchange        ONE PROCESS PER GPU (the layout of bench.py --gpus N and of distributed.py): the K x
//                             SRRG2_RECORD_FLOATS record table of a sharded batch by ONE ncclAllReduce(sum) over the int64 bit
//                             patterns of its rows -- north_star's "RCCL all-reduce of the final Hessian", native: no
//                             torch.distributed in a C++ host
#pragma once
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "srrg2_slam_amd.hpp"

namespace srrg2_slam_amd {

template <typename AlignerType>
class ShardedAligners {
public:
  static constexpr int Dim = AlignerType::Dim;
  using EstimateType       = typename AlignerType::EstimateType;

  // handles: one per device (or several on one device), same slices and PARAMs on each; not owned
  explicit ShardedAligners(std::vector<AlignerType*> handles) : _a(std::move(handles)) {
    if (_a.empty()) throw std::runtime_error("ShardedAligners| no aligner");
    for (AlignerType* a : _a)
      if (!a) throw std::runtime_error("ShardedAligners| null aligner");
  }
  int size() const { return (int) _a.size(); }
  AlignerType& handle(int g) { return *_a[(size_t) g]; }

  // aligner->setFixed on every handle: the fixed cloud is replicated, its search structure built once per device
  // (multi_loop_detector_brute_force_impl.cpp:63)
  void setFixed(int slice, const float* coords, int stride_bytes, const float* normals, int normal_stride_bytes, int n) {
    forEach([&](int g) { _a[(size_t) g]->setFixed(slice, coords, stride_bytes, normals, normal_stride_bytes, n); });
  }

  // K independent alignments: k -> handle k mod G, one compute_batch per handle, all handles at once; results in
  // alignment order.  Equal, record for record, to one handle running the whole batch (tests/cpp/test_loop_closure.cpp).
  std::vector<srrg2_batch_result> computeBatch(const std::vector<const float*>& clouds, const std::vector<int>& sizes,
                                               const std::vector<const float*>& normals,
                                               const std::vector<EstimateType>& guesses) {
    const int K = (int) clouds.size(), G = size();
    if ((int) sizes.size() != K || (int) guesses.size() != K || (!normals.empty() && (int) normals.size() != K))
      throw std::runtime_error("ShardedAligners::computeBatch|inconsistent argument sizes");
    std::vector<srrg2_batch_result> results((size_t) K);
    forEach([&](int g) {
      std::vector<const float*> c, nr;
      std::vector<int> sz;
      std::vector<EstimateType> gs;
      for (int k = g; k < K; k += G) {
        c.push_back(clouds[(size_t) k]);
        if (!normals.empty()) nr.push_back(normals[(size_t) k]);
        sz.push_back(sizes[(size_t) k]);
        gs.push_back(guesses[(size_t) k]);
      }
      if (c.empty()) return;
      const std::vector<srrg2_batch_result> r = _a[(size_t) g]->computeBatch(c, sz, nr, gs);
      for (size_t j = 0; j < r.size(); ++j) results[(size_t) g + j * (size_t) G] = r[j];
    });
    return results;
  }

  // The K x SRRG2_RECORD_FLOATS table of SURVEY.md 8e: handle g packs ITS rows (srrg2_multi_gpu_pack_record) into a
  // zero table, the G tables are added.  x + 0 = x: the sum holds every record unchanged, H included -- what the
  // multi-process path obtains with ONE all-reduce(sum) over RCCL.
  std::vector<double> recordTable(const std::vector<srrg2_batch_result>& results, int variable_kind) const {
    const int K = (int) results.size(), G = size();
    std::vector<double> table((size_t) K * SRRG2_RECORD_FLOATS, 0.0), mine;
    for (int g = 0; g < G; ++g) {
      mine.assign(table.size(), 0.0);
      for (int k = g; k < K; k += G) check(srrg2_multi_gpu_pack_record(k, variable_kind, &results[(size_t) k], mine.data() + (size_t) k * SRRG2_RECORD_FLOATS));
      for (size_t i = 0; i < table.size(); ++i) table[i] += mine[i];
    }
    return table;
  }

private:
  // fn(g) on one host thread per handle; the first exception (the reference throws std::runtime_error on misuse) is
  // re-thrown on the caller's thread after all threads have joined
  template <typename Fn>
  void forEach(Fn fn) {
    const int G = size();
    if (G == 1) {
      fn(0);
      return;
    }
    std::vector<std::thread> threads;
    std::vector<std::exception_ptr> errors((size_t) G);
    for (int g = 0; g < G; ++g)
      threads.emplace_back([&, g]() {
        try {
          fn(g);
        } catch (...) {
          errors[(size_t) g] = std::current_exception();
        }
      });
    for (std::thread& t : threads) t.join();
    for (const std::exception_ptr& e : errors)
      if (e) std::rethrow_exception(e);
  }
  std::vector<AlignerType*> _a;
};

// ---- ONE alignment over G handles of one process, sharded by moving points -------------------------------------------
// Usage: HostPointShardReducer red(G); handle g: setPointShard(HostPointShardReducer::hook, red.participant(g), total);
// the G compute() calls run on G threads.  Every hook call is a collective of all G participants.
class HostPointShardReducer {
public:
  struct Participant {
    HostPointShardReducer* owner;
    int rank;
  };
  explicit HostPointShardReducer(int G) : _G(G), _bufs((size_t) G), _parts((size_t) G) {
    for (int g = 0; g < G; ++g) _parts[(size_t) g] = Participant{this, g};
  }
  void* participant(int g) { return &_parts[(size_t) g]; }

  static int hook(void* user, int op, void* device_buffer, size_t count, void* stream) {
    Participant* p = static_cast<Participant*>(user);
    return p->owner->reduce(p->rank, op, device_buffer, count, stream);
  }

private:
  // Every call takes BOTH barriers whatever happens on this rank: a rank that left early (a failed wait or copy) would
  // leave its peers blocked in the first barrier, and -- run_compute keeps calling the hook after a failure so that the
  // collectives stay matched -- its next call would pair with the wrong barrier generation.  A local failure is recorded
  // in the shared flag before the first barrier; every rank returns 1 for that call, after the second barrier, and the
  // last one to leave clears the flag so that the reducer can be used again.
  int reduce(int rank, int op, void* dev, size_t count, void* stream) {
    const size_t bytes = count * (op == SRRG2_REDUCE_SUM_I64 ? 8 : 4);
    std::vector<char>& mine = _bufs[(size_t) rank];
    mine.assign(bytes, 0);
    // the producers of the buffer are ordered on `stream`: wait for them, then read
    if (srrg2_amd_stream_synchronize(stream) || srrg2_amd_memcpy(mine.data(), dev, bytes, 0, nullptr)) {
      std::lock_guard<std::mutex> lock(_m);
      _failed = true;
    }
    barrier();
    if (rank == 0) {
      _result.assign(bytes, 0);
      for (int g = 0; g < _G; ++g) {
        if (_bufs[(size_t) g].size() != bytes) {
          std::lock_guard<std::mutex> lock(_m);
          _failed = true;
          break;
        }
        if (op == SRRG2_REDUCE_SUM_I64) {
          long long* acc       = reinterpret_cast<long long*>(_result.data());
          const long long* src = reinterpret_cast<const long long*>(_bufs[(size_t) g].data());
          for (size_t i = 0; i < count; ++i) acc[i] = (long long) ((unsigned long long) acc[i] + (unsigned long long) src[i]);
        } else {
          unsigned* acc       = reinterpret_cast<unsigned*>(_result.data());
          const unsigned* src = reinterpret_cast<const unsigned*>(_bufs[(size_t) g].data());
          for (size_t i = 0; i < count; ++i) acc[i] = src[i] > acc[i] ? src[i] : acc[i];
        }
      }
    }
    barrier();
    bool failed;
    {
      std::lock_guard<std::mutex> lock(_m);
      failed = _failed;
    }
    // (synchronous copy: visible to the stream's next launch; a failed call leaves the buffer as it was)
    const int rc = failed ? 1 : (srrg2_amd_memcpy(dev, _result.data(), bytes, 1, nullptr) ? 1 : 0);
    {  // the last rank to have read the flag clears it: the next collective starts clean
      std::lock_guard<std::mutex> lock(_m);
      if (++_observed == _G) {
        _observed = 0;
        _failed   = false;
      }
    }
    // (no rank can set _failed for the NEXT call before every rank has left this one's second barrier -- all of them have,
    // or this code would not run -- but one may set it before the slowest rank has read it here; the third barrier closes
    // that window)
    barrier();
    return rc;
  }
  void barrier() {
    std::unique_lock<std::mutex> lock(_m);
    const int gen = _generation;
    if (++_arrived == _G) {
      _arrived = 0;
      ++_generation;
      _cv.notify_all();
    } else {
      _cv.wait(lock, [&] { return _generation != gen; });
    }
  }
  int _G;
  std::vector<std::vector<char>> _bufs;
  std::vector<Participant> _parts;
  std::vector<char> _result;
  std::mutex _m;
  std::condition_variable _cv;
  int _arrived = 0, _generation = 0, _observed = 0;
  bool _failed = false;
};

// ---- the same hook on RCCL: one communicator per device of this process (ncclCommInitAll), ncclAllReduce on the aligner's
// stream.  librccl.so is opened at run time: a caller of this header needs neither RCCL's nor HIP's headers.
class RcclPointShardReducer {
public:
  struct Participant {
    RcclPointShardReducer* owner;
    int rank;
  };
  // devices[g] = HIP device ordinal of handle g (RCCL wants distinct devices for distinct ranks)
  explicit RcclPointShardReducer(const std::vector<int>& devices) : _comms(devices.size(), nullptr), _parts(devices.size()) {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      _lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (_lib) break;
    }
    if (!_lib) throw std::runtime_error(std::string("RcclPointShardReducer| cannot open librccl.so: ") + dlerror());
    _init_all   = reinterpret_cast<InitAll>(dlsym(_lib, "ncclCommInitAll"));
    _all_reduce = reinterpret_cast<AllReduce>(dlsym(_lib, "ncclAllReduce"));
    _destroy    = reinterpret_cast<Destroy>(dlsym(_lib, "ncclCommDestroy"));
    if (!_init_all || !_all_reduce || !_destroy) throw std::runtime_error("RcclPointShardReducer| librccl.so lacks ncclCommInitAll / ncclAllReduce");
    const int rc = _init_all(_comms.data(), (int) devices.size(), devices.data());
    if (rc != 0) throw std::runtime_error("RcclPointShardReducer| ncclCommInitAll failed, code " + std::to_string(rc));
    for (size_t g = 0; g < devices.size(); ++g) _parts[g] = Participant{this, (int) g};
  }
  ~RcclPointShardReducer() {
    for (void* c : _comms)
      if (c && _destroy) _destroy(c);
    // (the library stays loaded: RCCL keeps threads that outlive its communicators)
  }
  RcclPointShardReducer(const RcclPointShardReducer&)            = delete;
  RcclPointShardReducer& operator=(const RcclPointShardReducer&) = delete;
  void* participant(int g) { return &_parts[(size_t) g]; }

  static int hook(void* user, int op, void* device_buffer, size_t count, void* stream) {
    Participant* p = static_cast<Participant*>(user);
    // ncclDataType_t: ncclUint32 = 3, ncclInt64 = 4; ncclRedOp_t: ncclSum = 0, ncclMax = 2 (nccl.h / rccl.h)
    const int dtype = op == SRRG2_REDUCE_SUM_I64 ? 4 : 3, red = op == SRRG2_REDUCE_SUM_I64 ? 0 : 2;
    return p->owner->_all_reduce(device_buffer, device_buffer, count, dtype, red, p->owner->_comms[(size_t) p->rank], stream) != 0;
  }

private:
  using InitAll   = int (*)(void**, int, const int*);
  using AllReduce = int (*)(const void*, void*, size_t, int, int, void*, void*);
  using Destroy   = int (*)(void*);
  void* _lib            = nullptr;
  InitAll _init_all     = nullptr;
  AllReduce _all_reduce = nullptr;
  Destroy _destroy      = nullptr;
  std::vector<void*> _comms;
  std::vector<Participant> _parts;
};

// ---- the record table between PROCESSES (one per GPU) -------------------------------------------------------------------
// Every rank runs its share of the K alignments (k mod world: srrg2_multi_gpu_shard_indices), packs ITS records into a zero
// table (srrg2_multi_gpu_pack_record) and calls allReduce: afterwards every rank holds every record -- X, statistics, H -- and
// applies the accept gates of multi_loop_detector_brute_force_impl.cpp:94-112 to all K of them.  The sum runs over the rows'
// int64 bit patterns (x + 0 = x for every pattern, -0.0 included): the table of distributed.py's all_reduce_records, bit for bit.
// The communicator: rank 0 creates an id (createId) and hands its 128 bytes to the other ranks by whatever the application
// launches its processes with (a file, an environment variable, MPI, a socket); every rank constructs with it -- a collective.
class RcclRecordExchange {
public:
  struct UniqueId {
    char bytes[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
  };
  static UniqueId createId() {
    Api api = load();
    UniqueId id;
    std::memset(&id, 0, sizeof(id));
    const int rc = api.get_id(&id);
    if (rc != 0) throw std::runtime_error("RcclRecordExchange| ncclGetUniqueId failed, code " + std::to_string(rc));
    return id;
  }
  // local_rank: this process's GPU (srrg2_multi_gpu_init: device = local_rank mod device count, bound to the calling thread)
  RcclRecordExchange(const UniqueId& id, int world, int rank, int local_rank) : _api(load()), _world(world), _rank(rank) {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("RcclRecordExchange| bad world / rank");
    check(srrg2_multi_gpu_init(local_rank, &_device));
    const int rc = _api.init_rank(&_comm, world, id, rank);
    if (rc != 0) throw std::runtime_error("RcclRecordExchange| ncclCommInitRank failed, code " + std::to_string(rc));
  }
  ~RcclRecordExchange() {
    if (_buf) (void) srrg2_amd_device_free(_buf);
    if (_comm) (void) _api.destroy(_comm);
  }
  RcclRecordExchange(const RcclRecordExchange&)            = delete;
  RcclRecordExchange& operator=(const RcclRecordExchange&) = delete;
  int world() const { return _world; }
  int rank() const { return _rank; }
  int device() const { return _device; }

  // this rank's rows of the K x SRRG2_RECORD_FLOATS table from its results (alignment k of this rank = results[j], k = rank + j world)
  std::vector<double> localTable(const std::vector<srrg2_batch_result>& mine, int K, int variable_kind) const {
    std::vector<double> table((size_t) K * SRRG2_RECORD_FLOATS, 0.0);
    for (size_t j = 0; j < mine.size(); ++j) {
      const int k = _rank + (int) j * _world;
      if (k >= K) throw std::runtime_error("RcclRecordExchange::localTable| more results than this rank's share");
      check(srrg2_multi_gpu_pack_record(k, variable_kind, &mine[j], table.data() + (size_t) k * SRRG2_RECORD_FLOATS));
    }
    return table;
  }
  // in place: every rank's rows summed over the ranks (a collective: every rank calls it with a table of the same size)
  void allReduce(std::vector<double>& table) {
    const size_t bytes = table.size() * sizeof(double);
    if (bytes == 0) return;
    if (bytes > _cap) {
      if (_buf) check(srrg2_amd_device_free(_buf));
      _buf = nullptr;
      _cap = 0;
      check(srrg2_amd_device_malloc(bytes, &_buf));
      _cap = bytes;
    }
    check(srrg2_amd_memcpy(_buf, table.data(), bytes, 1, nullptr));
    // ncclInt64 = 4, ncclSum = 0; the null stream: this exchange happens once per batch, after the results are on the host
    const int rc = _api.all_reduce(_buf, _buf, table.size(), 4, 0, _comm, nullptr);
    if (rc != 0) throw std::runtime_error("RcclRecordExchange| ncclAllReduce failed, code " + std::to_string(rc));
    check(srrg2_amd_stream_synchronize(nullptr));
    check(srrg2_amd_memcpy(table.data(), _buf, bytes, 0, nullptr));
  }

private:
  struct Api {
    int (*get_id)(UniqueId*);
    int (*init_rank)(void**, int, UniqueId, int);
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, void*);
    int (*destroy)(void*);
  };
  static Api load() {
    void* lib = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) throw std::runtime_error(std::string("RcclRecordExchange| cannot open librccl.so: ") + dlerror());
    Api api;
    api.get_id     = reinterpret_cast<int (*)(UniqueId*)>(dlsym(lib, "ncclGetUniqueId"));
    api.init_rank  = reinterpret_cast<int (*)(void**, int, UniqueId, int)>(dlsym(lib, "ncclCommInitRank"));
    api.all_reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, void*)>(dlsym(lib, "ncclAllReduce"));
    api.destroy    = reinterpret_cast<int (*)(void*)>(dlsym(lib, "ncclCommDestroy"));
    if (!api.get_id || !api.init_rank || !api.all_reduce || !api.destroy)
      throw std::runtime_error("RcclRecordExchange| librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce");
    return api;  // (the library stays loaded: RCCL keeps threads that outlive its communicators)
  }
  Api _api;
  int _world, _rank, _device = 0;
  void* _comm = nullptr;
  void* _buf  = nullptr;
  size_t _cap = 0;
};

}  // namespace srrg2_slam_amd
