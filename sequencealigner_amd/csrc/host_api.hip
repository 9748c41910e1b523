// Host-buffer entry points: the device batch queue that takes the place of the reference's thread pool
// dispatch (src/mapper.cpp:662-679).  Units (pairs / sequences) are independent, so they are split over
// the GPUs of the node by cost with no inter-GPU communication; one host thread per GPU stages its shard
// through pinned memory, runs the device-resident batch call on its own stream and scatters the results
// back to the caller's arrays in input order.
//
// Streams and buffers are kept between calls (HostContext): a call of one pair - what the C++ drop-in makes
// of every Align() - costs copies, launches and one synchronisation, not a dozen allocations.  Large shards
// are staged and their CIGAR text is formatted by several host threads.
#include "sa_internal.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <memory>
#include <new>
#include <stdexcept>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <numeric>
#include <thread>
#include <vector>

namespace sa {
namespace {

// A buffer that only grows; device or pinned host memory.
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void drop() {
    if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
    p = nullptr;
    cap = 0;
  }
  int ensure(size_t bytes) {
    if (bytes <= cap) return SA_OK;
    drop();
    const size_t want = round_up(std::max(bytes, (size_t)4096), 4096);
    const hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(SA_ERR_HIP, "%s(%zu) failed: %s", pinned ? "hipHostMalloc" : "hipMalloc", want,
                  hipGetErrorString(e));
    }
    cap = want;
    return SA_OK;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

// One stream and one set of buffers on one GPU, used by one call at a time and kept for the next call.
// Buffers above kKeepBytes are given back when the call ends, so that a single large batch does not pin
// memory for the rest of the process.
constexpr size_t kKeepBytes = (size_t)256 << 20;
constexpr int kBufs = 12;
struct HostContext {
  int device = 0;
  hipStream_t stream = nullptr;
  Buf pin[kBufs], dev[kBufs];
  HostContext() {
    for (auto& b : pin) b.pinned = true;
  }
};

class ContextPool {
 public:
  HostContext* acquire(int device) {
    {
      std::lock_guard<std::mutex> g(m_);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i]->device == device) {
          HostContext* c = free_[i];
          free_.erase(free_.begin() + i);
          return c;
        }
    }
    HostContext* c = new HostContext();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return nullptr;
    }
    return c;
  }
  void release(HostContext* c) {
    for (auto& b : c->pin)
      if (b.cap > kKeepBytes) b.drop();
    for (auto& b : c->dev)
      if (b.cap > kKeepBytes) b.drop();
    std::lock_guard<std::mutex> g(m_);
    free_.push_back(c);
  }

 private:
  std::mutex m_;
  std::vector<HostContext*> free_;  // never destroyed: the HIP runtime may be gone before static destructors run
};

ContextPool& pool() {
  static ContextPool* p = new ContextPool();
  return *p;
}

struct Lease {
  HostContext* c;
  explicit Lease(int device) : c(pool().acquire(device)) {}
  ~Lease() {
    if (!c) return;
    // a call that fails half way leaves work queued that still reads or writes the context's buffers
    (void)hipStreamSynchronize(c->stream);
    pool().release(c);
  }
};

// Host threads a call may use for staging and text: SA_HOST_THREADS, else OMP_NUM_THREADS, else every core (at most 64).
unsigned host_threads() {
  static const unsigned n = [] {
    const char* e = getenv("SA_HOST_THREADS");
    if (!e) e = getenv("OMP_NUM_THREADS");
    unsigned v = e ? (unsigned)atoi(e) : 0;
    if (!v) v = std::max(1u, std::thread::hardware_concurrency());
    return std::min(v, 64u);
  }();
  return n;
}

// SA_HOST_TRACE=1 prints the wall time of each stage of a shard to stderr.
struct StageTimer {
  int device;
  bool on = getenv("SA_HOST_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit StageTimer(int dev) : device(dev) {}
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sa host dev %d] %-24s %8.3f ms\n", device, what,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// f(begin, end) over [0, n) on up to `threads` host threads, at least `grain` items each.
template <class F>
void parallel_for(size_t n, size_t grain, unsigned threads, F f) {
  const size_t parts = std::min<size_t>(std::max<size_t>(threads, 1), std::max<size_t>(n / grain, 1));
  if (parts <= 1) {
    f((size_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  for (size_t t = 1; t < parts; ++t) th.emplace_back([=]() { f(n * t / parts, n * (t + 1) / parts); });
  f((size_t)0, n / parts);
  for (auto& t : th) t.join();
}

// Number of shards a host batch call deals its units to: one per GPU.  With the "fake_devices" option the
// shards are that many, all of them run on GPU 0 side by side (own host thread, stream and buffers each),
// which exercises the multi-GPU path on a one-GPU machine.
int pick_devices(int requested) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
  const long fake = options().fake_devices;
  if (fake > 0) count = (int)std::min(fake, 64L);
  if (requested <= 0 || requested > count) return count;
  return requested;
}
int device_of_shard(int shard) { return options().fake_devices > 0 ? 0 : shard; }

__global__ void gather_runs(const uint32_t* runs, const uint32_t* nruns, const uint64_t* out_off,
                            uint64_t stride, uint32_t n, uint32_t* out) {
  const uint32_t p = blockIdx.x;
  if (p >= n) return;
  const uint32_t c = nruns[p];
  const uint32_t* src = runs + (uint64_t)(p + 1) * stride - c;
  uint32_t* dst = out + out_off[p];
  for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) dst[i] = src[i];
}

const char kOps[4] = {'M', 'X', 'I', 'D'};

// "<count><op>..." exactly as CompressCigar prints it (src/alignment.cpp:17-33)
char* runs_to_text(const uint32_t* runs, uint32_t n) {
  char* out = (char*)malloc((size_t)n * 11 + 1);  // a run is at most 10 digits and a letter
  if (!out) return nullptr;
  char* w = out;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t v = runs[i] >> 2;
    char digits[10];
    int d = 0;
    do {
      digits[d++] = (char)('0' + v % 10);
      v /= 10;
    } while (v);
    while (d) *w++ = digits[--d];
    *w++ = kOps[runs[i] & 3];
  }
  *w = 0;
  return out;
}

// One part of a GPU's share of an alignment batch, in two phases.  `begin` stages the part's pairs through pinned
// memory and queues everything the GPU has to do for it - copies in, the batch call, the scan of the pairs' run
// counts, the gathering of their runs into one block, copies out - without waiting for any of it; `finish` waits,
// scatters the results into the caller's arrays and formats the CIGAR text.  A GPU's host thread begins all its
// parts and then finishes them in order: while it stages part k + 1 the GPU copies and aligns part k, and while
// it formats part k's text the GPU works on the parts after it.  `threads` host threads help with staging and
// text; `mem_share` parts hold workspaces on the GPU at once.
struct AlignJob {
  std::unique_ptr<Lease> lease;
  const std::vector<uint32_t>* idx = nullptr;
  uint64_t stride = 0;
  bool want_cigar = false, tb_traced = false, has_tb = false;
  StageTimer trace{0};
};
enum { ABUF_Q, ABUF_T, ABUF_QO, ABUF_TO, ABUF_SCORE, ABUF_TB, ABUF_NRUNS, ABUF_RUNS, ABUF_OFF, ABUF_WS, ABUF_COMPACT, ABUF_TMP };
static_assert(ABUF_TMP < kBufs, "buffers of a context");

int align_shard_begin(AlignJob& job, int device, const std::vector<uint32_t>& idx, const char* const* queries,
                      const uint32_t* qlens, const char* const* targets, const uint32_t* tlens, int type,
                      int match, int mismatch, int gap, bool want_target_begin, bool want_cigar, unsigned threads,
                      unsigned mem_share) {
  const size_t n = idx.size();  // below 2^32 (checked by the caller)
  job.idx = &idx;
  if (!n) return SA_OK;
  SA_HIP_TRY(hipSetDevice(device));
  job.lease.reset(new Lease(device));
  if (!job.lease->c) return fail(SA_ERR_HIP, "cannot create a stream on GPU %d", device);
  HostContext& c = *job.lease->c;
  hipStream_t st = c.stream;
  job.trace = StageTimer(device);
  StageTimer& trace = job.trace;
  // Pinned staging, ONE block - both offset arrays, then the queries, then the targets (16 bytes of slack each:
  // the kernels read aligned 8-byte words) - which goes to the device in one copy: four copies on a stream start
  // a good hundred microseconds apart each, and a part's kernels wait for the last of them.
  int rc;
  uint64_t qtot = 0, ttot = 0;
  uint32_t maxq = 0, maxt = 0, minq = ~0u, mint = ~0u;
  for (size_t j = 0; j < n; ++j) {
    const uint32_t i = idx[j];
    qtot += qlens[i];
    ttot += tlens[i];
    maxq = std::max(maxq, qlens[i]);
    maxt = std::max(maxt, tlens[i]);
    minq = std::min(minq, qlens[i]);
    mint = std::min(mint, tlens[i]);
  }
  const size_t at_qo = 0, at_to = (n + 1) * 8, at_q = 2 * (n + 1) * 8, at_t = round_up(at_q + qtot + 16, 256);
  const size_t in_bytes = at_t + ttot + 16;
  if ((rc = c.pin[ABUF_Q].ensure(in_bytes))) return rc;
  char* hin = c.pin[ABUF_Q].as<char>();
  uint64_t* hqo = reinterpret_cast<uint64_t*>(hin + at_qo);
  uint64_t* hto = reinterpret_cast<uint64_t*>(hin + at_to);
  {
    uint64_t qa = 0, ta = 0;
    for (size_t j = 0; j < n; ++j) {
      const uint32_t i = idx[j];
      hqo[j] = qa;
      hto[j] = ta;
      qa += qlens[i];
      ta += tlens[i];
    }
    hqo[n] = qa;
    hto[n] = ta;
  }
  trace.mark("offsets");
  char* hq = hin + at_q;
  char* ht = hin + at_t;
  parallel_for(n, 4096, threads, [&](size_t b, size_t e) {
    for (size_t j = b; j < e; ++j) {
      const uint32_t i = idx[j];
      memcpy(hq + hqo[j], queries[i], qlens[i]);
      memcpy(ht + hto[j], targets[i], tlens[i]);
    }
  });
  memset(hq + qtot, 0, 16);
  memset(ht + ttot, 0, 16);
  trace.mark("staging");

  const uint64_t stride = (uint64_t)maxq + maxt;
  job.stride = stride;
  job.want_cigar = want_cigar;
  if ((rc = c.dev[ABUF_Q].ensure(in_bytes)) || (rc = c.dev[ABUF_SCORE].ensure(n * 4)) || (rc = c.dev[ABUF_TB].ensure(n * 4)))
    return rc;
  if (want_cigar && ((rc = c.dev[ABUF_RUNS].ensure((size_t)n * stride * 4)) || (rc = c.dev[ABUF_NRUNS].ensure((n + 1) * 4))))
    return rc;
  SA_HIP_TRY(hipMemcpyAsync(c.dev[ABUF_Q].p, hin, in_bytes, hipMemcpyHostToDevice, st));
  char* din = c.dev[ABUF_Q].as<char>();

  sa_align_batch b{};
  b.d_q = reinterpret_cast<uint8_t*>(din + at_q);
  b.d_q_off = reinterpret_cast<uint64_t*>(din + at_qo);
  b.d_t = reinterpret_cast<uint8_t*>(din + at_t);
  b.d_t_off = reinterpret_cast<uint64_t*>(din + at_to);
  b.n = (uint32_t)n;
  b.max_qlen = maxq;
  b.max_tlen = maxt;
  b.min_qlen = minq;
  b.min_tlen = mint;
  b.type = type;
  b.match = match;
  b.mismatch = mismatch;
  b.gap = gap;
  b.d_score = c.dev[ABUF_SCORE].as<int32_t>();
  // Global / Semiglobal derive *target_begin from the CIGAR text; without a CIGAR the reference inspects an
  // empty string and stores 1 (src/alignment.cpp:129-136, 342-348), and no traceback is needed
  job.tb_traced = want_target_begin && (want_cigar || type == SA_LOCAL);
  b.d_target_begin = (job.tb_traced || want_cigar) ? c.dev[ABUF_TB].as<uint32_t>() : nullptr;
  job.has_tb = b.d_target_begin != nullptr;
  b.d_cigar_runs = want_cigar ? c.dev[ABUF_RUNS].as<uint32_t>() : nullptr;
  b.d_cigar_nruns = want_cigar ? c.dev[ABUF_NRUNS].as<uint32_t>() : nullptr;
  b.cigar_stride = stride;

  size_t min_ws = 0;
  size_t ws = sa_align_workspace_bytes(&b, &min_ws);
  if (ws > c.dev[ABUF_WS].cap) {  // a workspace kept from an earlier call serves as long as it holds one pass
    size_t free_b = 0, total_b = 0;
    SA_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = (size_t)((free_b + c.dev[ABUF_WS].cap) * 0.8 / mem_share);  // parts side by side
    if (ws > budget) ws = std::max(min_ws, budget / min_ws * min_ws);
    if ((rc = c.dev[ABUF_WS].ensure(ws))) return rc;
  }
  trace.mark("buffers + copies queued");
  if ((rc = sa_align_batch_device(device, st, &b, c.dev[ABUF_WS].p, c.dev[ABUF_WS].cap))) return rc;

  if ((rc = c.pin[ABUF_SCORE].ensure(n * 4)) || (rc = c.pin[ABUF_TB].ensure(n * 4))) return rc;
  SA_HIP_TRY(hipMemcpyAsync(c.pin[ABUF_SCORE].p, c.dev[ABUF_SCORE].p, n * 4, hipMemcpyDeviceToHost, st));
  if (job.has_tb) SA_HIP_TRY(hipMemcpyAsync(c.pin[ABUF_TB].p, c.dev[ABUF_TB].p, n * 4, hipMemcpyDeviceToHost, st));
  if (want_cigar) {
    // where every pair's runs go in one block: the exclusive scan of the run counts, on the device, so that the
    // gathering follows the traceback without a round trip to the host
    size_t tmp_bytes = 0;
    SA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, n + 1,
                                       rocprim::plus<uint64_t>(), st));
    if ((rc = c.dev[ABUF_TMP].ensure(tmp_bytes)) || (rc = c.dev[ABUF_OFF].ensure((n + 1) * 8)) ||
        (rc = c.dev[ABUF_COMPACT].ensure((size_t)n * stride * 4)) || (rc = c.pin[ABUF_OFF].ensure((n + 1) * 8)) ||
        (rc = c.pin[ABUF_NRUNS].ensure(n * 4)))
      return rc;
    SA_HIP_TRY(rocprim::exclusive_scan(c.dev[ABUF_TMP].p, tmp_bytes, c.dev[ABUF_NRUNS].as<uint32_t>(),
                                       c.dev[ABUF_OFF].as<uint64_t>(), (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(gather_runs, dim3((uint32_t)n), dim3(64), 0, st, c.dev[ABUF_RUNS].as<uint32_t>(),
                       c.dev[ABUF_NRUNS].as<uint32_t>(), c.dev[ABUF_OFF].as<uint64_t>(), stride, (uint32_t)n,
                       c.dev[ABUF_COMPACT].as<uint32_t>());
    SA_HIP_TRY(hipGetLastError());
    SA_HIP_TRY(hipMemcpyAsync(c.pin[ABUF_OFF].p, c.dev[ABUF_OFF].p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    SA_HIP_TRY(hipMemcpyAsync(c.pin[ABUF_NRUNS].p, c.dev[ABUF_NRUNS].p, n * 4, hipMemcpyDeviceToHost, st));
  }
  trace.mark("kernels queued");
  return SA_OK;
}

int align_shard_finish(AlignJob& job, int32_t* scores, uint32_t* target_begin, char** cigars, unsigned threads) {
  if (!job.lease || !job.lease->c) return SA_OK;  // an empty part
  const std::vector<uint32_t>& idx = *job.idx;
  const size_t n = idx.size();
  HostContext& c = *job.lease->c;
  hipStream_t st = c.stream;
  StageTimer& trace = job.trace;
  int rc;
  SA_HIP_TRY(hipSetDevice(c.device));
  SA_HIP_TRY(hipStreamSynchronize(st));
  trace.mark("device done");
  const uint64_t* off = c.pin[ABUF_OFF].as<uint64_t>();
  if (job.want_cigar) {  // the runs, now that their number is known (the copy runs under the scattering)
    if ((rc = c.pin[ABUF_RUNS].ensure(off[n] * 4 + 4))) return rc;
    SA_HIP_TRY(hipMemcpyAsync(c.pin[ABUF_RUNS].p, c.dev[ABUF_COMPACT].p, off[n] * 4, hipMemcpyDeviceToHost, st));
  }
  const int32_t* hscore = c.pin[ABUF_SCORE].as<int32_t>();
  const uint32_t* htb = c.pin[ABUF_TB].as<uint32_t>();
  for (size_t j = 0; j < n; ++j) {
    scores[idx[j]] = hscore[j];
    if (target_begin) target_begin[idx[j]] = job.tb_traced ? htb[j] : 1u;
  }
  trace.mark("scores scattered");
  if (job.want_cigar) {
    SA_HIP_TRY(hipStreamSynchronize(st));
    trace.mark("runs copied");
    const uint32_t* hnr = c.pin[ABUF_NRUNS].as<uint32_t>();
    const uint32_t* runs = c.pin[ABUF_RUNS].as<uint32_t>();
    std::atomic<bool> oom{false};
    parallel_for(n, 2048, threads, [&](size_t b0, size_t e0) {
      for (size_t j = b0; j < e0; ++j) {
        cigars[idx[j]] = runs_to_text(runs + off[j], hnr[j]);
        if (!cigars[idx[j]]) oom = true;
      }
    });
    if (oom) return fail(SA_ERR_MEMORY, "out of host memory");
    trace.mark("CIGAR text");
  }
  return SA_OK;
}

int minimize_shard(int device, const std::vector<uint32_t>& idx, const char* const* seqs,
                   const uint32_t* lens, uint32_t k, uint32_t w, int variant, uint32_t** out_triples,
                   uint64_t* out_counts, unsigned threads) {
  const size_t n = idx.size();  // below 2^32 (checked by the caller)
  if (!n) return SA_OK;
  SA_HIP_TRY(hipSetDevice(device));
  Lease lease(device);
  if (!lease.c) return fail(SA_ERR_HIP, "cannot create a stream on GPU %d", device);
  HostContext& c = *lease.c;
  hipStream_t st = c.stream;
  enum { SEQ, OFF, OUT, OOFF, STATUS, WS };
  StageTimer trace(device);
  int rc;
  if ((rc = c.pin[OFF].ensure((n + 1) * 8))) return rc;
  uint64_t* ho = c.pin[OFF].as<uint64_t>();
  uint64_t tot = 0;
  uint32_t maxl = 0, minl = ~0u;
  for (size_t j = 0; j < n; ++j) {
    ho[j] = tot;
    tot += lens[idx[j]];
    maxl = std::max(maxl, lens[idx[j]]);
    minl = std::min(minl, lens[idx[j]]);
  }
  ho[n] = tot;
  if ((rc = c.pin[SEQ].ensure(tot + 16))) return rc;
  char* hs = c.pin[SEQ].as<char>();
  parallel_for(n, 1024, threads, [&](size_t b, size_t e) {
    for (size_t j = b; j < e; ++j) memcpy(hs + ho[j], seqs[idx[j]], lens[idx[j]]);
  });
  memset(hs + tot, 0, 16);
  trace.mark("staging");
  const uint64_t cap = sa_minimize_out_capacity((uint32_t)n, tot, k, w);
  const size_t ws = sa_minimize_workspace_bytes((uint32_t)n, tot, maxl, k, w, variant);
  if ((rc = c.dev[SEQ].ensure(tot + 16)) || (rc = c.dev[OFF].ensure((n + 1) * 8)) || (rc = c.dev[OUT].ensure(cap * 12)) ||
      (rc = c.dev[OOFF].ensure((n + 1) * 8)) || (rc = c.dev[STATUS].ensure(4)) || (rc = c.dev[WS].ensure(ws)))
    return rc;
  SA_HIP_TRY(hipMemcpyAsync(c.dev[SEQ].p, hs, tot + 16, hipMemcpyHostToDevice, st));
  SA_HIP_TRY(hipMemcpyAsync(c.dev[OFF].p, ho, (n + 1) * 8, hipMemcpyHostToDevice, st));
  sa_minimize_batch b{};
  b.d_seq = c.dev[SEQ].as<uint8_t>();
  b.d_off = c.dev[OFF].as<uint64_t>();
  b.n = (uint32_t)n;
  b.total_len = tot;
  b.max_len = maxl;
  b.min_len = minl;
  b.k = k;
  b.w = w;
  b.variant = variant;
  b.d_out = c.dev[OUT].as<uint32_t>();
  b.out_capacity = cap;
  b.d_out_off = c.dev[OOFF].as<uint64_t>();
  b.d_status = c.dev[STATUS].as<int>();
  trace.mark("buffers + copies queued");
  if ((rc = sa_minimize_batch_device(device, st, &b, c.dev[WS].p, ws))) return rc;
  trace.mark("kernels queued");
  if ((rc = c.pin[OOFF].ensure((n + 1) * 8)) || (rc = c.pin[STATUS].ensure(4))) return rc;
  const uint64_t* ooff = c.pin[OOFF].as<uint64_t>();
  SA_HIP_TRY(hipMemcpyAsync(c.pin[OOFF].p, c.dev[OOFF].p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
  SA_HIP_TRY(hipMemcpyAsync(c.pin[STATUS].p, c.dev[STATUS].p, 4, hipMemcpyDeviceToHost, st));
  SA_HIP_TRY(hipStreamSynchronize(st));
  trace.mark("device done");
  const int status = *c.pin[STATUS].as<int>();
  if (status == SA_ERR_RANGE)
    return fail(SA_ERR_RANGE, "decimal Minimize: a k-mer value exceeds INT_MAX (stoi out of range)");
  if (status == SA_ERR_ARGUMENT && variant == SA_MIN_DECIMAL)
    return fail(SA_ERR_ARGUMENT, "decimal Minimize: a sequence holds a byte other than ACGT / acgt (not supported)");
  if (status) return fail(status, "device reported status %d", status);
  if ((rc = c.pin[OUT].ensure(ooff[n] * 12 + 12))) return rc;
  const uint32_t* all = c.pin[OUT].as<uint32_t>();
  SA_HIP_TRY(hipMemcpyAsync(c.pin[OUT].p, c.dev[OUT].p, ooff[n] * 12, hipMemcpyDeviceToHost, st));
  SA_HIP_TRY(hipStreamSynchronize(st));
  trace.mark("triples copied");
  std::atomic<bool> oom{false};
  parallel_for(n, 1024, threads, [&](size_t b0, size_t e0) {
    for (size_t j = b0; j < e0; ++j) {
      const uint64_t cnt = ooff[j + 1] - ooff[j];
      uint32_t* dst = (uint32_t*)malloc(cnt ? cnt * 12 : 4);
      if (!dst) {
        oom = true;
        continue;
      }
      memcpy(dst, all + ooff[j] * 3, cnt * 12);
      out_triples[idx[j]] = dst;
      out_counts[idx[j]] = cnt;
    }
  });
  if (oom) return fail(SA_ERR_MEMORY, "out of host memory");
  trace.mark("lists handed out");
  return SA_OK;
}

// Sorted split: units ordered by decreasing `key` (so that neighbouring lanes / waves see similar shapes; left
// in input order when all keys are equal), cut into blocks of `block` units, and every block given to the shard
// with the smallest total `cost` so far (the first such shard): longest-processing-time-first over blocks.
std::vector<std::vector<uint32_t>> deal(const std::vector<uint64_t>& key, const std::vector<uint64_t>& cost,
                                        int ndev, uint32_t block) {
  std::vector<uint32_t> order(key.size());
  std::iota(order.begin(), order.end(), 0u);
  const bool uniform = std::all_of(key.begin(), key.end(), [&](uint64_t c) { return c == key[0]; });
  if (!uniform)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b]; });
  std::vector<std::vector<uint32_t>> shards(ndev);
  std::vector<long double> load(ndev, 0);
  for (size_t b0 = 0; b0 < order.size(); b0 += block) {
    const size_t b1 = std::min(order.size(), b0 + block);
    long double c = 0;
    for (size_t i = b0; i < b1; ++i) c += (long double)cost[order[i]];
    const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    load[d] += c;
    shards[d].insert(shards[d].end(), order.begin() + b0, order.begin() + b1);
  }
  return shards;
}

template <class F>
int run_on_devices(int ndev, F f) {  // ndev tasks, one host thread each
  if (ndev == 1) return f(0);  // on the caller's thread
  std::vector<int> rcs(ndev, SA_OK);
  std::vector<std::string> msgs(ndev);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; ++d)
    th.emplace_back([&, d]() {
      try {
        rcs[d] = f(d);
        if (rcs[d]) msgs[d] = last_error();
      } catch (...) {  // nothing may leave a thread (or the C boundary) as an exception
        rcs[d] = SA_ERR_MEMORY;
      }
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < ndev; ++d)
    if (rcs[d]) return fail(rcs[d], "shard %d: %s", d, msgs[d].c_str());
  return SA_OK;
}

struct Triple {
  uint32_t value, pos, strand;
};
struct TripleLess {
  __host__ __device__ bool operator()(const Triple& a, const Triple& b) const {
    if (a.value != b.value) return a.value < b.value;
    if (a.pos != b.pos) return a.pos < b.pos;
    return a.strand < b.strand;
  }
};

__global__ void triple_flags(const Triple* t, uint32_t n, uint32_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flags[i] = 0;
    return;
  }
  flags[i] = (i == 0 || t[i].value != t[i - 1].value || t[i].pos != t[i - 1].pos ||
              t[i].strand != t[i - 1].strand)
                 ? 1u
                 : 0u;
}

__global__ void triple_compact(const Triple* t, const uint32_t* flags, const uint32_t* pos, uint32_t n,
                               Triple* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) out[pos[i]] = t[i];
}

// The two batch calls hand out malloc'ed results.  A call that fails gives back what it had handed out so
// far and leaves NULL in every slot, and no C++ exception crosses the boundary.
template <class F>
int guarded(void** handed_out, size_t n, F body) {
  if (handed_out) std::fill(handed_out, handed_out + n, nullptr);
  int rc;
  try {
    rc = body();
  } catch (const std::bad_alloc&) {
    rc = fail(SA_ERR_MEMORY, "out of host memory");
  } catch (const std::exception& e) {
    rc = fail(SA_ERR_HIP, "unexpected failure: %s", e.what());
  }
  if (rc && handed_out)
    for (size_t i = 0; i < n; ++i) {
      free(handed_out[i]);
      handed_out[i] = nullptr;
    }
  return rc;
}

int align_batch_host(const char* const* queries, const uint32_t* query_lens,
                            const char* const* targets, const uint32_t* target_lens, size_t n, int type,
                            int match, int mismatch, int gap, int n_devices, int32_t* scores,
                            uint32_t* target_begin, char** cigars) {
  const int ndev = pick_devices(n_devices);
  if (ndev == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
  // A device batch is planned for its longest query and target: pairs are grouped by size class (powers of
  // two of both lengths, at least 64) and every class is run as a batch of its own, so that a few long
  // outliers do not make every pair pay for their matrix.  Classes of fewer than 64 pairs run together.
  auto log2_ceil = [](uint32_t v) {
    uint32_t b = 6;
    while (b < 32 && (1ull << b) < v) ++b;
    return b;
  };
  auto class_of = [&](size_t i) { return (log2_ceil(target_lens[i]) << 8) | log2_ceil(query_lens[i]); };
  std::vector<std::vector<uint32_t>> classes;
  {
    // the common case first: every pair in the class of pair 0 (two range tests per pair)
    const uint32_t bq = log2_ceil(query_lens[0]), bt = log2_ceil(target_lens[0]);
    const uint64_t q_hi = 1ull << bq, q_lo = bq == 6 ? 0 : 1ull << (bq - 1);
    const uint64_t t_hi = 1ull << bt, t_lo = bt == 6 ? 0 : 1ull << (bt - 1);
    size_t i = 0;
    for (; i < n; ++i) {
      const uint64_t q = query_lens[i], t = target_lens[i];
      if (q > q_hi || (q <= q_lo && q_lo) || t > t_hi || (t <= t_lo && t_lo)) break;
    }
    if (i == n) {
      classes.emplace_back(n);
      std::iota(classes[0].begin(), classes[0].end(), 0u);
    }
  }
  const bool in_order = !classes.empty();  // one class that lists the pairs in their order
  if (classes.empty()) {
    std::vector<std::pair<uint32_t, uint32_t>> keyed(n);  // (class, pair)
    for (size_t i = 0; i < n; ++i) keyed[i] = {class_of(i), (uint32_t)i};
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
    std::vector<uint32_t> rest;
    for (size_t i = 0; i < n;) {
      size_t e = i;
      while (e < n && keyed[e].first == keyed[i].first) ++e;
      std::vector<uint32_t>& into = e - i >= 64 ? classes.emplace_back() : rest;
      for (size_t x = i; x < e; ++x) into.push_back(keyed[x].second);
      i = e;
    }
    if (!rest.empty()) classes.push_back(std::move(rest));
  }

  for (const std::vector<uint32_t>& members : classes) {
    std::vector<uint64_t> key(members.size()), cost(members.size());
    for (size_t i = 0; i < members.size(); ++i) {  // order by target, then query length; balance by cells
      key[i] = ((uint64_t)target_lens[members[i]] << 32) | query_lens[members[i]];
      cost[i] = (uint64_t)target_lens[members[i]] * query_lens[members[i]];
    }
    auto shards = deal(key, cost, ndev, 1024);
    if (!in_order)  // class members -> pair numbers
      for (auto& sh : shards)
        for (uint32_t& x : sh) x = members[x];
    // Large shards are cut into a few parts, each with a stream and buffers of its own: the GPU's host thread
    // begins them one after the other and then finishes them in that order (AlignJob).
    size_t largest = 0;
    for (const auto& sh : shards) largest = std::max(largest, sh.size());
    // (measured, 150 x 300 Local with CIGAR text: 200 000 pairs 10.4 / 8.7 / 10.0 / 11.0 ms in 1 / 2 / 3 / 4 parts - a
    // fill launch of a few ten thousand pairs takes as long as one of 400 000, its wavefronts run alone on their
    // SIMDs - and 10^6 pairs 35.8 / 37.2 ms in 2 / 4 parts, 30.5 / 26.5 ms without text); SA_HOST_PARTS pins it
    unsigned parts = largest >= 393216 ? 4 : largest >= 65536 ? 2 : 1;
    if (const char* e = getenv("SA_HOST_PARTS")) parts = (unsigned)std::min(16, std::max(1, atoi(e)));
    std::vector<std::vector<uint32_t>> cut;
    for (const auto& sh : shards)
      for (unsigned p = 0; p < parts; ++p)
        cut.emplace_back(sh.begin() + sh.size() * p / parts, sh.begin() + sh.size() * (p + 1) / parts);
    const unsigned threads = std::max(1u, host_threads() / (unsigned)ndev);
    const int rc = run_on_devices(ndev, [&](int d) {
      std::vector<AlignJob> jobs(parts);
      for (unsigned p = 0; p < parts; ++p)
        if (int r = align_shard_begin(jobs[p], device_of_shard(d), cut[(size_t)d * parts + p], queries, query_lens, targets,
                                      target_lens, type, match, mismatch, gap, target_begin != nullptr,
                                      cigars != nullptr, threads, parts))
          return r;
      for (unsigned p = 0; p < parts; ++p)
        if (int r = align_shard_finish(jobs[p], scores, target_begin, cigars, threads)) return r;
      return (int)SA_OK;
    });
    if (rc) return rc;
  }
  return SA_OK;
}

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" {

void sa_free(void* p) { free(p); }

int sa_align_batch_host(const char* const* queries, const uint32_t* query_lens,
                        const char* const* targets, const uint32_t* target_lens, size_t n, int type,
                        int match, int mismatch, int gap, int n_devices, int32_t* scores,
                        uint32_t* target_begin, char** cigars) {
  if (n == 0) return (type < 0 || type > 2) ? fail(SA_ERR_ARGUMENT, "unknown alignment type %d", type) : SA_OK;
  return guarded((void**)cigars, n, [&] {
    if (type < 0 || type > 2) return fail(SA_ERR_ARGUMENT, "unknown alignment type %d", type);
    if (!queries || !query_lens || !targets || !target_lens || !scores)
      return fail(SA_ERR_ARGUMENT, "NULL argument");
    if (n > 0xFFFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many pairs");
    return align_batch_host(queries, query_lens, targets, target_lens, n, type, match, mismatch, gap,
                            n_devices, scores, target_begin, cigars);
  });
}

int sa_minimize_batch_host(const char* const* seqs, const uint32_t* lens, size_t n, uint32_t k,
                           uint32_t w, int variant, int n_devices, uint32_t** out_triples,
                           uint64_t* out_counts) {
  if (n == 0) return SA_OK;
  return guarded((void**)out_triples, n, [&] {
    if (!seqs || !lens || !out_triples || !out_counts) return fail(SA_ERR_ARGUMENT, "NULL argument");
    if (n > 0xFFFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many sequences");
    const int ndev = pick_devices(n_devices);
    if (ndev == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
    std::vector<uint64_t> cost(n);
    for (size_t i = 0; i < n; ++i) cost[i] = lens[i];
    const auto shards = deal(cost, cost, ndev, 16);
    const unsigned threads = std::max(1u, host_threads() / (unsigned)ndev);
    return run_on_devices(ndev, [&](int d) {
      return minimize_shard(device_of_shard(d), shards[d], seqs, lens, k, w, variant, out_triples, out_counts, threads);
    });
  });
}

int sa_shard_units(const uint64_t* cost, size_t n, int n_shards, uint32_t block, uint32_t* shard_of,
                   uint32_t* order) {
  if (n_shards <= 0 || block == 0 || (n && (!cost || !shard_of)) || n > 0xFFFFFFF0ull)
    return fail(SA_ERR_ARGUMENT, "sa_shard_units: bad argument");
  try {
    const std::vector<uint64_t> c(cost, cost + n);
    const auto shards = deal(c, c, n_shards, block);
    size_t at = 0;
    for (int d = 0; d < n_shards; ++d)
      for (uint32_t u : shards[d]) {
        shard_of[u] = (uint32_t)d;
        if (order) order[at] = u;
        ++at;
      }
  } catch (const std::bad_alloc&) {
    return fail(SA_ERR_MEMORY, "out of host memory");
  }
  return SA_OK;
}

int sa_remove_duplicate_host(uint32_t* triples, size_t n, uint32_t* out, size_t* out_n) {
  if (out_n) *out_n = 0;
  if (n == 0) return SA_OK;
  if (!triples || !out || !out_n) return fail(SA_ERR_ARGUMENT, "NULL argument");
  if (n > 0x7FFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many tuples");
  if (pick_devices(1) == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
  SA_HIP_TRY(hipSetDevice(0));
  Lease lease(0);
  if (!lease.c) return fail(SA_ERR_HIP, "cannot create a stream on GPU 0");
  HostContext& c = *lease.c;
  hipStream_t st = c.stream;
  enum { IN, SORTED, FLAGS, POS, OUT, TMP };
  int rc;
  const uint32_t m = (uint32_t)n;
  if ((rc = c.dev[IN].ensure(n * 12)) || (rc = c.dev[SORTED].ensure(n * 12)) || (rc = c.dev[FLAGS].ensure((n + 1) * 4)) ||
      (rc = c.dev[POS].ensure((n + 1) * 4)) || (rc = c.dev[OUT].ensure(n * 12)))
    return rc;
  Triple* din = c.dev[IN].as<Triple>();
  Triple* dsorted = c.dev[SORTED].as<Triple>();
  uint32_t* dflags = c.dev[FLAGS].as<uint32_t>();
  uint32_t* dpos = c.dev[POS].as<uint32_t>();
  SA_HIP_TRY(hipMemcpyAsync(din, triples, n * 12, hipMemcpyHostToDevice, st));
  size_t t1 = 0, t2 = 0;
  SA_HIP_TRY(rocprim::merge_sort(nullptr, t1, din, dsorted, n, TripleLess(), st));
  SA_HIP_TRY(rocprim::exclusive_scan(nullptr, t2, dflags, dpos, 0u, n + 1, rocprim::plus<uint32_t>(), st));
  if ((rc = c.dev[TMP].ensure(std::max(t1, t2)))) return rc;
  SA_HIP_TRY(rocprim::merge_sort(c.dev[TMP].p, t1, din, dsorted, n, TripleLess(), st));
  hipLaunchKernelGGL(triple_flags, dim3((m + 256) / 256), dim3(256), 0, st, dsorted, m, dflags);
  SA_HIP_TRY(rocprim::exclusive_scan(c.dev[TMP].p, t2, dflags, dpos, 0u, n + 1, rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(triple_compact, dim3((m + 255) / 256), dim3(256), 0, st, dsorted, dflags, dpos, m,
                     c.dev[OUT].as<Triple>());
  SA_HIP_TRY(hipGetLastError());
  uint32_t uniq = 0;
  SA_HIP_TRY(hipMemcpyAsync(&uniq, dpos + n, 4, hipMemcpyDeviceToHost, st));
  SA_HIP_TRY(hipMemcpyAsync(triples, dsorted, n * 12, hipMemcpyDeviceToHost, st));  // sorted in place
  SA_HIP_TRY(hipStreamSynchronize(st));
  SA_HIP_TRY(hipMemcpyAsync(out, c.dev[OUT].p, (size_t)uniq * 12, hipMemcpyDeviceToHost, st));
  SA_HIP_TRY(hipStreamSynchronize(st));
  *out_n = uniq;
  return SA_OK;
}

}  // extern "C"
