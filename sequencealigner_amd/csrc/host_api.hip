// Host-buffer entry points: the device batch queue that takes the place of the reference's thread pool
// dispatch (src/mapper.cpp:662-679).  Units (pairs / sequences) are independent, so they are split over
// the GPUs of the node by cost with no inter-GPU communication; one host thread per GPU stages its shard
// through pinned memory, runs the device-resident batch call on its own stream and scatters the results
// back to the caller's arrays in input order.
#include "sa_internal.h"

#include <cstring>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <numeric>
#include <thread>
#include <vector>

namespace sa {
namespace {

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (p) (void)hipFree(p);
    p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
    if (e != hipSuccess) return fail(SA_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SA_OK;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

struct PinBuf {
  void* p = nullptr;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
  int alloc(size_t bytes) {
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 256, hipHostMallocDefault);
    if (e != hipSuccess)
      return fail(SA_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SA_OK;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

struct Stream {
  hipStream_t s = nullptr;
  ~Stream() {
    if (s) (void)hipStreamDestroy(s);
  }
};

int pick_devices(int requested) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
  if (requested <= 0 || requested > count) return count;
  return requested;
}

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
  std::string s;
  s.reserve((size_t)n * 4 + 1);
  for (uint32_t i = 0; i < n; ++i) {
    s += std::to_string(runs[i] >> 2);
    s += kOps[runs[i] & 3];
  }
  char* out = (char*)malloc(s.size() + 1);
  if (out) memcpy(out, s.c_str(), s.size() + 1);
  return out;
}

// One GPU's share of an alignment batch.
int align_shard(int device, const std::vector<uint32_t>& idx, const char* const* queries,
                const uint32_t* qlens, const char* const* targets, const uint32_t* tlens, int type,
                int match, int mismatch, int gap, int32_t* scores, uint32_t* target_begin,
                char** cigars) {
  const uint32_t n = (uint32_t)idx.size();
  if (!n) return SA_OK;
  SA_HIP_TRY(hipSetDevice(device));
  Stream st;
  SA_HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));

  uint64_t qtot = 0, ttot = 0;
  uint32_t maxq = 0, maxt = 0, minq = ~0u, mint = ~0u;
  for (uint32_t i : idx) {
    qtot += qlens[i];
    ttot += tlens[i];
    maxq = std::max(maxq, qlens[i]);
    maxt = std::max(maxt, tlens[i]);
    minq = std::min(minq, qlens[i]);
    mint = std::min(mint, tlens[i]);
  }
  // pinned staging: sequences + offsets (8 bytes of slack: the kernels read aligned 8-byte words)
  PinBuf hq, ht, hqo, hto;
  int rc;
  if ((rc = hq.alloc(qtot + 16)) || (rc = ht.alloc(ttot + 16)) || (rc = hqo.alloc((n + 1) * 8)) ||
      (rc = hto.alloc((n + 1) * 8)))
    return rc;
  uint64_t qo = 0, to = 0;
  for (uint32_t j = 0; j < n; ++j) {
    const uint32_t i = idx[j];
    hqo.as<uint64_t>()[j] = qo;
    hto.as<uint64_t>()[j] = to;
    memcpy(hq.as<char>() + qo, queries[i], qlens[i]);
    memcpy(ht.as<char>() + to, targets[i], tlens[i]);
    qo += qlens[i];
    to += tlens[i];
  }
  hqo.as<uint64_t>()[n] = qo;
  hto.as<uint64_t>()[n] = to;
  memset(hq.as<char>() + qo, 0, 16);
  memset(ht.as<char>() + to, 0, 16);

  DevBuf dq, dt, dqo, dto, dscore, dtb, druns, dnruns, dws;
  if ((rc = dq.alloc(qtot + 16)) || (rc = dt.alloc(ttot + 16)) || (rc = dqo.alloc((n + 1) * 8)) ||
      (rc = dto.alloc((n + 1) * 8)) || (rc = dscore.alloc(n * 4)) || (rc = dtb.alloc(n * 4)))
    return rc;
  const uint64_t stride = (uint64_t)maxq + maxt;
  const bool want_cigar = cigars != nullptr;
  if (want_cigar && ((rc = druns.alloc((size_t)n * stride * 4)) || (rc = dnruns.alloc(n * 4)))) return rc;
  SA_HIP_TRY(hipMemcpyAsync(dq.p, hq.p, qtot + 16, hipMemcpyHostToDevice, st.s));
  SA_HIP_TRY(hipMemcpyAsync(dt.p, ht.p, ttot + 16, hipMemcpyHostToDevice, st.s));
  SA_HIP_TRY(hipMemcpyAsync(dqo.p, hqo.p, (n + 1) * 8, hipMemcpyHostToDevice, st.s));
  SA_HIP_TRY(hipMemcpyAsync(dto.p, hto.p, (n + 1) * 8, hipMemcpyHostToDevice, st.s));

  sa_align_batch b{};
  b.d_q = dq.as<uint8_t>();
  b.d_q_off = dqo.as<uint64_t>();
  b.d_t = dt.as<uint8_t>();
  b.d_t_off = dto.as<uint64_t>();
  b.n = n;
  b.max_qlen = maxq;
  b.max_tlen = maxt;
  b.min_qlen = minq;
  b.min_tlen = mint;
  b.type = type;
  b.match = match;
  b.mismatch = mismatch;
  b.gap = gap;
  b.d_score = dscore.as<int32_t>();
  b.d_target_begin = (target_begin || want_cigar) ? dtb.as<uint32_t>() : nullptr;
  b.d_cigar_runs = want_cigar ? druns.as<uint32_t>() : nullptr;
  b.d_cigar_nruns = want_cigar ? dnruns.as<uint32_t>() : nullptr;
  b.cigar_stride = stride;

  size_t min_ws = 0;
  size_t ws = sa_align_workspace_bytes(&b, &min_ws);
  size_t free_b = 0, total_b = 0;
  SA_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t budget = (size_t)(free_b * 0.8);
  if (ws > budget) ws = std::max(min_ws, budget / min_ws * min_ws);
  if ((rc = dws.alloc(ws))) return rc;
  if ((rc = sa_align_batch_device(device, st.s, &b, dws.p, ws))) return rc;

  std::vector<int32_t> hscore(n);
  std::vector<uint32_t> htb(n), hnr;
  SA_HIP_TRY(hipMemcpyAsync(hscore.data(), dscore.p, n * 4, hipMemcpyDeviceToHost, st.s));
  if (b.d_target_begin)
    SA_HIP_TRY(hipMemcpyAsync(htb.data(), dtb.p, n * 4, hipMemcpyDeviceToHost, st.s));
  if (want_cigar) {
    hnr.resize(n);
    SA_HIP_TRY(hipMemcpyAsync(hnr.data(), dnruns.p, n * 4, hipMemcpyDeviceToHost, st.s));
  }
  SA_HIP_TRY(hipStreamSynchronize(st.s));
  for (uint32_t j = 0; j < n; ++j) {
    scores[idx[j]] = hscore[j];
    if (target_begin) target_begin[idx[j]] = htb[j];
  }
  if (want_cigar) {
    // compact the runs on the device, then one copy back
    std::vector<uint64_t> off(n + 1, 0);
    for (uint32_t j = 0; j < n; ++j) off[j + 1] = off[j] + hnr[j];
    DevBuf doff, dcompact;
    if ((rc = doff.alloc((n + 1) * 8)) || (rc = dcompact.alloc(off[n] * 4))) return rc;
    SA_HIP_TRY(hipMemcpyAsync(doff.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st.s));
    hipLaunchKernelGGL(gather_runs, dim3(n), dim3(64), 0, st.s, druns.as<uint32_t>(),
                       dnruns.as<uint32_t>(), doff.as<uint64_t>(), stride, n, dcompact.as<uint32_t>());
    SA_HIP_TRY(hipGetLastError());
    std::vector<uint32_t> runs(off[n] ? off[n] : 1);
    SA_HIP_TRY(hipMemcpyAsync(runs.data(), dcompact.p, off[n] * 4, hipMemcpyDeviceToHost, st.s));
    SA_HIP_TRY(hipStreamSynchronize(st.s));
    for (uint32_t j = 0; j < n; ++j) {
      cigars[idx[j]] = runs_to_text(runs.data() + off[j], hnr[j]);
      if (!cigars[idx[j]]) return fail(SA_ERR_ARGUMENT, "out of host memory");
    }
  }
  return SA_OK;
}

int minimize_shard(int device, const std::vector<uint32_t>& idx, const char* const* seqs,
                   const uint32_t* lens, uint32_t k, uint32_t w, int variant, uint32_t** out_triples,
                   uint64_t* out_counts) {
  const uint32_t n = (uint32_t)idx.size();
  if (!n) return SA_OK;
  SA_HIP_TRY(hipSetDevice(device));
  Stream st;
  SA_HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
  uint64_t tot = 0;
  uint32_t maxl = 0, minl = ~0u;
  for (uint32_t i : idx) {
    tot += lens[i];
    maxl = std::max(maxl, lens[i]);
    minl = std::min(minl, lens[i]);
  }
  PinBuf hs, ho;
  int rc;
  if ((rc = hs.alloc(tot + 16)) || (rc = ho.alloc((n + 1) * 8))) return rc;
  uint64_t o = 0;
  for (uint32_t j = 0; j < n; ++j) {
    ho.as<uint64_t>()[j] = o;
    memcpy(hs.as<char>() + o, seqs[idx[j]], lens[idx[j]]);
    o += lens[idx[j]];
  }
  ho.as<uint64_t>()[n] = o;
  memset(hs.as<char>() + o, 0, 16);
  const uint64_t cap = sa_minimize_out_capacity(n, tot, k, w);
  const size_t ws = sa_minimize_workspace_bytes(n, tot, maxl, k, w, variant);
  DevBuf ds, doff, dout, dooff, dstatus, dws;
  if ((rc = ds.alloc(tot + 16)) || (rc = doff.alloc((n + 1) * 8)) || (rc = dout.alloc(cap * 12)) ||
      (rc = dooff.alloc((n + 1) * 8)) || (rc = dstatus.alloc(4)) || (rc = dws.alloc(ws)))
    return rc;
  SA_HIP_TRY(hipMemcpyAsync(ds.p, hs.p, tot + 16, hipMemcpyHostToDevice, st.s));
  SA_HIP_TRY(hipMemcpyAsync(doff.p, ho.p, (n + 1) * 8, hipMemcpyHostToDevice, st.s));
  sa_minimize_batch b{};
  b.d_seq = ds.as<uint8_t>();
  b.d_off = doff.as<uint64_t>();
  b.n = n;
  b.total_len = tot;
  b.max_len = maxl;
  b.min_len = minl;
  b.k = k;
  b.w = w;
  b.variant = variant;
  b.d_out = dout.as<uint32_t>();
  b.out_capacity = cap;
  b.d_out_off = dooff.as<uint64_t>();
  b.d_status = dstatus.as<int>();
  if ((rc = sa_minimize_batch_device(device, st.s, &b, dws.p, ws))) return rc;
  std::vector<uint64_t> ooff(n + 1);
  int status = 0;
  SA_HIP_TRY(hipMemcpyAsync(ooff.data(), dooff.p, (n + 1) * 8, hipMemcpyDeviceToHost, st.s));
  SA_HIP_TRY(hipMemcpyAsync(&status, dstatus.p, 4, hipMemcpyDeviceToHost, st.s));
  SA_HIP_TRY(hipStreamSynchronize(st.s));
  if (status == SA_ERR_RANGE)
    return fail(SA_ERR_RANGE, "decimal Minimize: a k-mer value exceeds INT_MAX (stoi out of range)");
  if (status) return fail(status, "device reported status %d", status);
  std::vector<uint32_t> all(ooff[n] * 3 + 3);
  SA_HIP_TRY(hipMemcpyAsync(all.data(), dout.p, ooff[n] * 12, hipMemcpyDeviceToHost, st.s));
  SA_HIP_TRY(hipStreamSynchronize(st.s));
  for (uint32_t j = 0; j < n; ++j) {
    const uint64_t c = ooff[j + 1] - ooff[j];
    uint32_t* dst = (uint32_t*)malloc(c ? c * 12 : 4);
    if (!dst) return fail(SA_ERR_ARGUMENT, "out of host memory");
    memcpy(dst, all.data() + ooff[j] * 3, c * 12);
    out_triples[idx[j]] = dst;
    out_counts[idx[j]] = c;
  }
  return SA_OK;
}

// Sorted split: units ordered by size (so that neighbouring lanes/waves see similar shapes), dealt to
// the GPUs in blocks of `block` units round-robin, which also balances the total cost.
std::vector<std::vector<uint32_t>> deal(const std::vector<uint64_t>& cost, int ndev, uint32_t block) {
  std::vector<uint32_t> order(cost.size());
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  std::vector<std::vector<uint32_t>> shards(ndev);
  for (size_t i = 0; i < order.size(); ++i) shards[(i / block) % ndev].push_back(order[i]);
  return shards;
}

template <class F>
int run_on_devices(int ndev, F f) {
  std::vector<int> rcs(ndev, SA_OK);
  std::vector<std::string> msgs(ndev);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; ++d)
    th.emplace_back([&, d]() {
      rcs[d] = f(d);
      if (rcs[d]) msgs[d] = last_error();
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < ndev; ++d)
    if (rcs[d]) return fail(rcs[d], "GPU %d: %s", d, msgs[d].c_str());
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

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" {

void sa_free(void* p) { free(p); }

int sa_align_batch_host(const char* const* queries, const uint32_t* query_lens,
                        const char* const* targets, const uint32_t* target_lens, size_t n, int type,
                        int match, int mismatch, int gap, int n_devices, int32_t* scores,
                        uint32_t* target_begin, char** cigars) {
  if (type < 0 || type > 2) return fail(SA_ERR_ARGUMENT, "unknown alignment type %d", type);
  if (n == 0) return SA_OK;
  if (!queries || !query_lens || !targets || !target_lens || !scores)
    return fail(SA_ERR_ARGUMENT, "NULL argument");
  if (n > 0xFFFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many pairs");
  const int ndev = pick_devices(n_devices);
  if (ndev == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
  std::vector<uint64_t> cost(n);
  for (size_t i = 0; i < n; ++i)
    cost[i] = ((uint64_t)target_lens[i] << 32) | query_lens[i];  // order by target, then query length
  const auto shards = deal(cost, ndev, 1024);
  return run_on_devices(ndev, [&](int d) {
    return align_shard(d, shards[d], queries, query_lens, targets, target_lens, type, match, mismatch,
                       gap, scores, target_begin, cigars);
  });
}

int sa_minimize_batch_host(const char* const* seqs, const uint32_t* lens, size_t n, uint32_t k,
                           uint32_t w, int variant, int n_devices, uint32_t** out_triples,
                           uint64_t* out_counts) {
  if (n == 0) return SA_OK;
  if (!seqs || !lens || !out_triples || !out_counts) return fail(SA_ERR_ARGUMENT, "NULL argument");
  if (n > 0xFFFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many sequences");
  const int ndev = pick_devices(n_devices);
  if (ndev == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
  std::vector<uint64_t> cost(n);
  for (size_t i = 0; i < n; ++i) cost[i] = lens[i];
  const auto shards = deal(cost, ndev, 16);
  return run_on_devices(ndev, [&](int d) {
    return minimize_shard(d, shards[d], seqs, lens, k, w, variant, out_triples, out_counts);
  });
}

int sa_remove_duplicate_host(uint32_t* triples, size_t n, uint32_t* out, size_t* out_n) {
  if (out_n) *out_n = 0;
  if (n == 0) return SA_OK;
  if (!triples || !out || !out_n) return fail(SA_ERR_ARGUMENT, "NULL argument");
  if (n > 0x7FFFFFF0ull) return fail(SA_ERR_ARGUMENT, "too many tuples");
  if (pick_devices(1) == 0) return fail(SA_ERR_NO_DEVICE, "no GPU visible (there is no CPU fallback)");
  SA_HIP_TRY(hipSetDevice(0));
  Stream st;
  SA_HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
  DevBuf din, dsorted, dflags, dpos, dout, dtmp;
  int rc;
  const uint32_t m = (uint32_t)n;
  if ((rc = din.alloc(n * 12)) || (rc = dsorted.alloc(n * 12)) || (rc = dflags.alloc((n + 1) * 4)) ||
      (rc = dpos.alloc((n + 1) * 4)) || (rc = dout.alloc(n * 12)))
    return rc;
  SA_HIP_TRY(hipMemcpyAsync(din.p, triples, n * 12, hipMemcpyHostToDevice, st.s));
  size_t t1 = 0, t2 = 0;
  SA_HIP_TRY(rocprim::merge_sort(nullptr, t1, din.as<Triple>(), dsorted.as<Triple>(), n, TripleLess(), st.s));
  SA_HIP_TRY(rocprim::exclusive_scan(nullptr, t2, dflags.as<uint32_t>(), dpos.as<uint32_t>(), 0u, n + 1,
                                     rocprim::plus<uint32_t>(), st.s));
  if ((rc = dtmp.alloc(std::max(t1, t2)))) return rc;
  SA_HIP_TRY(rocprim::merge_sort(dtmp.p, t1, din.as<Triple>(), dsorted.as<Triple>(), n, TripleLess(), st.s));
  hipLaunchKernelGGL(triple_flags, dim3((m + 256) / 256), dim3(256), 0, st.s, dsorted.as<Triple>(), m,
                     dflags.as<uint32_t>());
  SA_HIP_TRY(rocprim::exclusive_scan(dtmp.p, t2, dflags.as<uint32_t>(), dpos.as<uint32_t>(), 0u, n + 1,
                                     rocprim::plus<uint32_t>(), st.s));
  hipLaunchKernelGGL(triple_compact, dim3((m + 255) / 256), dim3(256), 0, st.s, dsorted.as<Triple>(),
                     dflags.as<uint32_t>(), dpos.as<uint32_t>(), m, dout.as<Triple>());
  SA_HIP_TRY(hipGetLastError());
  uint32_t uniq = 0;
  SA_HIP_TRY(hipMemcpyAsync(&uniq, dpos.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st.s));
  SA_HIP_TRY(hipMemcpyAsync(triples, dsorted.p, n * 12, hipMemcpyDeviceToHost, st.s));  // sorted in place
  SA_HIP_TRY(hipStreamSynchronize(st.s));
  SA_HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)uniq * 12, hipMemcpyDeviceToHost, st.s));
  SA_HIP_TRY(hipStreamSynchronize(st.s));
  *out_n = uniq;
  return SA_OK;
}

}  // extern "C"
