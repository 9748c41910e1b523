// Seed chaining of a batch of reads against one reference, on the device: the steps the reference's
// mapper performs per read between MinimizeBinary and Align (reference src/mapper.cpp):
//   getMinimizer :172-196                 index  value -> [(position, strand)]
//   ignoreTooFrequentMinimizer :224-232   drop the (int)((int)distinct * f) values with the largest
//                                         (occurrences, value)
//   matchMinimizer :234-290               all (read position, reference position) pairs of equal value and
//                                         strand, sorted by (read position, reference position)
//   LongestIncreasingSubsequence :292-352 strictly increasing reference positions
// Both minimizer lists arrive sorted by (value, position, strand) (that is how removeDuplicate returns
// them), so the index is the run structure of the reference list and a lookup is a binary search.
#include "sa_internal.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>

namespace sa {
namespace {

constexpr uint32_t kThreads = 256;

// ---- reference index ---------------------------------------------------------------------------------
// head[x] = 1 where reference triple x starts a run of equal values; strand1[x] = its strand flag
__global__ void ref_heads(const uint32_t* ref, uint32_t n, uint32_t* head, uint32_t* strand1) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x > n) return;
  if (x == n) {
    head[x] = 0;
    strand1[x] = 0;
    return;
  }
  head[x] = (x == 0 || ref[3 * x] != ref[3 * (x - 1)]) ? 1u : 0u;
  strand1[x] = ref[3 * x + 2] ? 1u : 0u;
}

// run id -> first triple, value; run_start[m] = n
__global__ void ref_runs(const uint32_t* ref, uint32_t n, const uint32_t* head, const uint32_t* id,
                         uint32_t* run_start, uint32_t* run_value) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x > n) return;
  if (x == n) {
    run_start[id[n]] = n;  // id[n] = number of runs
    return;
  }
  if (head[x]) {
    run_start[id[x]] = x;
    run_value[id[x]] = ref[3 * x];
  }
}

// sort key of a run: (occurrences, value); slots beyond the number of runs sort to the end
__global__ void ref_occurrences(uint32_t n, const uint32_t* id, const uint32_t* run_start,
                                const uint32_t* run_value, uint64_t* keys, uint8_t* alive) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t m = id[n];
  alive[r] = 1;
  keys[r] = r < m ? ((uint64_t)(run_start[r + 1] - run_start[r]) << 32) | run_value[r] : ~0ull;
}

// Lookup of a value among the m distinct reference values: a table over the values' top bits narrows the
// binary search to the few runs that share them.  bucket[b] = first run whose value >> shift is >= b;
// shift is chosen so that the largest value falls into the last of the kBuckets buckets.
constexpr uint32_t kBucketBits = 18, kBuckets = 1u << kBucketBits;

__device__ __forceinline__ uint32_t bucket_shift(uint32_t largest) {
  const uint32_t bits = 32 - __builtin_clz(largest | 1u);
  return bits > kBucketBits ? bits - kBucketBits : 0;
}

__global__ void ref_buckets(uint32_t n, const uint32_t* id, const uint32_t* run_value, uint32_t* bucket) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > kBuckets) return;
  const uint32_t m = id[n];
  const uint32_t shift = bucket_shift(run_value[m - 1]);
  uint32_t lo = 0, hi = m;  // first run whose value is >= b << shift
  if (b == kBuckets) {
    lo = m;
  } else {
    const uint32_t least = b << shift;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (run_value[mid] < least)
        lo = mid + 1;
      else
        hi = mid;
    }
  }
  bucket[b] = lo;
}

__device__ __forceinline__ uint32_t find_run(const uint32_t* run_value, const uint32_t* bucket, uint32_t m,
                                             uint32_t value) {
  const uint32_t b = value >> bucket_shift(run_value[m - 1]);
  if (b >= kBuckets) return ~0u;  // above the largest reference value
  uint32_t lo = bucket[b], hi = bucket[b + 1];  // first run with value >= `value` lies in [lo, hi]
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (run_value[mid] < value)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < m && run_value[lo] == value) ? lo : ~0u;
}

// the last (int)((int)m * f) keys of the sorted list are the values to drop (src/mapper.cpp:226-231)
__global__ void ref_drop(uint32_t n, const uint32_t* id, const uint64_t* sorted, const uint32_t* run_value,
                         const uint32_t* bucket, float f, uint8_t* alive) {
  const uint32_t m = id[n];
  const int index = (int)((float)(int)m * f);
  const int from = (int)m - index;
  const int x = from + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (x < 0 || x >= (int)m) return;
  const uint32_t r = find_run(run_value, bucket, m, (uint32_t)sorted[x]);
  if (r != ~0u) alive[r] = 0;
}

// What a lookup needs of a run once the frequency filter is final, in one 16-byte entry: first triple,
// number of triples, number of those on the forward strand, alive.
__global__ void ref_info(uint32_t n, const uint32_t* id, const uint32_t* run_start, const uint32_t* strand_pre,
                         const uint8_t* alive, uint4* info) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= id[n]) return;
  const uint32_t b = run_start[r], e = run_start[r + 1];
  info[r] = make_uint4(b, e - b, strand_pre[e] - strand_pre[b], alive[r]);
}

// ---- matches -------------------------------------------------------------------------------------------
struct ChainArgs {
  const uint32_t* ref;
  uint32_t n_ref;
  const uint32_t* id;          // id[n_ref] = number of runs
  const uint32_t* run_start;
  const uint32_t* run_value;
  const uint32_t* bucket;      // lookup table over the values' top bits (find_run)
  const uint32_t* strand_pre;  // exclusive prefix of the reference strand flags
  const uint8_t* alive;
  const uint4* info;           // per run: first triple, triples, forward-strand triples, alive
  uint32_t* rid;               // per read triple: its run (or ~0), kept from the counting pass for the writing pass
  const uint32_t* reads;
  const uint64_t* read_off;
  uint32_t n_reads;
  uint64_t n_read_triples;
};

// matches a read triple takes part in: reference entries of its value's run with the same strand
__global__ void match_count(ChainArgs a, uint32_t* cnt) {
  const uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (x > a.n_read_triples) return;
  uint32_t c = 0;
  if (x < a.n_read_triples) {
    const uint32_t m = a.id[a.n_ref];
    uint32_t r = find_run(a.run_value, a.bucket, m, a.reads[3 * x]);
    if (r != ~0u) {
      const uint4 run = a.info[r];
      if (run.w)
        c = a.reads[3 * x + 2] ? run.z : run.y - run.z;
      else
        r = ~0u;
    }
    a.rid[x] = r;
  }
  cnt[x] = c;
}

// writes them as keys (read position << 32 | reference position); a run lists its entries by position
__global__ void match_write(ChainArgs a, const uint64_t* moff, uint64_t* keys) {
  const uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (x >= a.n_read_triples || moff[x + 1] == moff[x]) return;
  const uint4 run = a.info[a.rid[x]];
  const uint32_t strand = a.reads[3 * x + 2] ? 1u : 0u;
  const uint64_t qpos = (uint64_t)a.reads[3 * x + 1] << 32;
  uint64_t o = moff[x];
  for (uint32_t y = run.x; y < run.x + run.y; ++y)
    if ((a.ref[3 * y + 2] ? 1u : 0u) == strand) keys[o++] = qpos | a.ref[3 * y + 1];
}

__global__ void read_segments(const uint64_t* read_off, uint32_t n_reads, const uint64_t* moff,
                              uint32_t* seg_begin, uint32_t* seg_end, uint64_t* n_matches) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t b = moff[read_off[r]], e = moff[read_off[r + 1]];
  seg_begin[r] = (uint32_t)b;
  seg_end[r] = (uint32_t)e;
  if (n_matches) n_matches[r] = e - b;
}

// ---- longest increasing subsequence, one wavefront per read (src/mapper.cpp:292-352) --------------------
// The reference keeps, per length, the match that ends the best subsequence of that length (tails) and a
// predecessor per match, and walks the predecessors back from the longest one to find where it starts.
// Here tails[len] / tailv[len] hold that match and its reference position (a search step is one load) and
// root[i] the FIRST match of the best subsequence that ends in match i, which is all the walk back is for:
// root[i] = i when i starts a subsequence, else root of its predecessor.  A match beyond the current best
// end is appended without a search - the search would end there too - and then its predecessor and root are
// already in registers.  The wave fetches 64 matches at a time; every lane runs the same sequential code.
// The read's slices start at seg_begin[r] (+ r for tails / tailv: n + 1 entries).
__global__ __launch_bounds__(kThreads) void chain_lis(const uint64_t* sorted, const uint32_t* seg_begin,
                                                      const uint32_t* seg_end, uint32_t n_reads,
                                                      int32_t* root_all, int32_t* tails_all, uint32_t* tailv_all,
                                                      int32_t* out) {
  const uint32_t r = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (r >= n_reads) return;
  const uint64_t* mt = sorted + seg_begin[r];
  const int n = (int)(seg_end[r] - seg_begin[r]);
  int32_t* o = out + 5 * (uint64_t)r;
  if (n == 0) {
    if (lane < 5) o[lane] = 0;
    return;
  }
  int32_t* root = root_all + seg_begin[r];
  int32_t* tails = tails_all + seg_begin[r] + r;
  uint32_t* tailv = tailv_all + seg_begin[r] + r;
  int length = 0;
  uint32_t best_end = 0;        // tailv[length]
  int top = 0, root_top = 0;    // tails[length] and its root
  for (int base = 0; base < n; base += 64) {
    const uint64_t mine = base + (int)lane < n ? mt[base + lane] : 0;
    const int cnt = n - base < 64 ? n - base : 64;
    for (int j = 0; j < cnt; ++j) {
      const int i = base + j;
      const uint32_t ref_i = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, j);
      int low, root_i;
      if (length == 0 || best_end < ref_i) {
        low = length + 1;
        root_i = length == 0 ? i : root_top;
      } else {
        low = 1;
        int high = length;
        while (low <= high) {
          const int mid = (low + high) / 2;
          if (tailv[mid] < ref_i)
            low = mid + 1;
          else
            high = mid - 1;
        }
        root_i = low == 1 ? i : root[tails[low - 1]];
      }
      // every lane stores the same values to the same places
      root[i] = root_i;
      tails[low] = i;
      tailv[low] = ref_i;
      if (low > length) length = low;
      if (low == length) {
        best_end = ref_i;
        top = i;
        root_top = root_i;
      }
    }
  }
  if (lane != 0) return;
  const uint64_t last = mt[top], first = mt[root_top];
  o[0] = length;
  o[1] = (int32_t)(uint32_t)first;
  o[2] = (int32_t)(uint32_t)last;
  o[3] = (int32_t)(first >> 32);
  o[4] = (int32_t)(last >> 32);
}

struct ChainPlan {
  size_t o_head, o_strand, o_id, o_strand_pre, o_run_start, o_run_value, o_bucket, o_info, o_rid, o_okeys, o_osorted, o_alive, o_cnt,
      o_moff, o_seg_begin, o_seg_end, o_tmp_fixed, fixed_bytes;
  size_t scan_ref_tmp, scan_read_tmp, sort_ref_tmp;
};

int make_chain_plan(uint32_t n_ref, uint64_t n_read_triples, uint32_t n_reads, ChainPlan* pl) {
  uint32_t* d32 = nullptr;
  uint64_t* d64 = nullptr;
  pl->scan_ref_tmp = pl->scan_read_tmp = pl->sort_ref_tmp = 0;
  if (rocprim::exclusive_scan(nullptr, pl->scan_ref_tmp, d32, d32, 0u, (size_t)n_ref + 1, rocprim::plus<uint32_t>()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::exclusive_scan(nullptr, pl->scan_read_tmp, d32, d64, (uint64_t)0, (size_t)n_read_triples + 1, rocprim::plus<uint64_t>()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::radix_sort_keys(nullptr, pl->sort_ref_tmp, d64, d64, (size_t)std::max(n_ref, 1u)) != hipSuccess) return SA_ERR_HIP;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += round_up(bytes ? bytes : 1, 256);
    return at;
  };
  const size_t r1 = (size_t)n_ref + 1;
  pl->o_head = take(r1 * 4);
  pl->o_strand = take(r1 * 4);
  pl->o_id = take(r1 * 4);
  pl->o_strand_pre = take(r1 * 4);
  pl->o_run_start = take(r1 * 4);
  pl->o_run_value = take(r1 * 4);
  pl->o_bucket = take(((size_t)kBuckets + 2) * 4);
  pl->o_info = take(r1 * 16);
  pl->o_rid = take((n_read_triples + 1) * 4);
  pl->o_okeys = take(r1 * 8);
  pl->o_osorted = take(r1 * 8);
  pl->o_alive = take(r1);
  pl->o_cnt = take((n_read_triples + 1) * 4);
  pl->o_moff = take((n_read_triples + 1) * 8);
  pl->o_seg_begin = take(((size_t)n_reads + 1) * 4);
  pl->o_seg_end = take(((size_t)n_reads + 1) * 4);
  pl->o_tmp_fixed = take(std::max(std::max(pl->scan_ref_tmp, pl->scan_read_tmp), pl->sort_ref_tmp));
  pl->fixed_bytes = o;
  return SA_OK;
}

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" int sa_chain_batch_device(int device, void* stream_, const sa_chain_batch* b, void* d_workspace,
                                     size_t workspace_bytes, size_t* needed_bytes) {
  if (!b) return fail(SA_ERR_ARGUMENT, "batch is NULL");
  if (needed_bytes) *needed_bytes = 0;
  if (b->n_reads == 0) return SA_OK;
  if (!b->d_ref || !b->d_reads || !b->d_read_off || !b->d_out)
    return fail(SA_ERR_ARGUMENT, "NULL reference / read / output pointer");
  if (b->n_ref == 0 || b->n_ref >= 0x7FFFFFF0u) return fail(SA_ERR_ARGUMENT, "reference list empty or too long");
  ChainPlan pl;
  if (make_chain_plan(b->n_ref, b->n_read_triples, b->n_reads, &pl) != SA_OK)
    return fail(SA_ERR_HIP, "cannot size the chaining workspace");
  if (needed_bytes) *needed_bytes = pl.fixed_bytes;
  if (!d_workspace || workspace_bytes < pl.fixed_bytes)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the %zu needed before matching",
                workspace_bytes, pl.fixed_bytes);
  SA_HIP_TRY(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)d_workspace;
  uint32_t* head = (uint32_t*)(ws + pl.o_head);
  uint32_t* strand = (uint32_t*)(ws + pl.o_strand);
  uint32_t* id = (uint32_t*)(ws + pl.o_id);
  uint32_t* strand_pre = (uint32_t*)(ws + pl.o_strand_pre);
  uint32_t* run_start = (uint32_t*)(ws + pl.o_run_start);
  uint32_t* run_value = (uint32_t*)(ws + pl.o_run_value);
  uint32_t* bucket = (uint32_t*)(ws + pl.o_bucket);
  uint4* info = (uint4*)(ws + pl.o_info);
  uint32_t* rid = (uint32_t*)(ws + pl.o_rid);
  uint64_t* okeys = (uint64_t*)(ws + pl.o_okeys);
  uint64_t* osorted = (uint64_t*)(ws + pl.o_osorted);
  uint8_t* alive = (uint8_t*)(ws + pl.o_alive);
  uint32_t* cnt = (uint32_t*)(ws + pl.o_cnt);
  uint64_t* moff = (uint64_t*)(ws + pl.o_moff);
  uint32_t* seg_begin = (uint32_t*)(ws + pl.o_seg_begin);
  uint32_t* seg_end = (uint32_t*)(ws + pl.o_seg_end);
  void* tmp = ws + pl.o_tmp_fixed;
  auto blocks = [](uint64_t items) { return dim3((unsigned)((items + kThreads - 1) / kThreads)); };
  const uint32_t n = b->n_ref;

  // reference index
  hipLaunchKernelGGL(ref_heads, blocks((uint64_t)n + 1), dim3(kThreads), 0, stream, b->d_ref, n, head, strand);
  size_t t = pl.scan_ref_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, head, id, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
  t = pl.scan_ref_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, strand, strand_pre, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(),
                                     stream));
  hipLaunchKernelGGL(ref_runs, blocks((uint64_t)n + 1), dim3(kThreads), 0, stream, b->d_ref, n, head, id,
                     run_start, run_value);
  hipLaunchKernelGGL(ref_buckets, blocks((uint64_t)kBuckets + 1), dim3(kThreads), 0, stream, n, id, run_value, bucket);
  hipLaunchKernelGGL(ref_occurrences, blocks(n), dim3(kThreads), 0, stream, n, id, run_start, run_value, okeys,
                     alive);
  SA_HIP_TRY(hipGetLastError());
  t = pl.sort_ref_tmp;
  SA_HIP_TRY(rocprim::radix_sort_keys(tmp, t, okeys, osorted, (size_t)n, 0u, 64u, stream));
  hipLaunchKernelGGL(ref_drop, blocks(n), dim3(kThreads), 0, stream, n, id, osorted, run_value, bucket,
                     b->frequent_fraction, alive);
  hipLaunchKernelGGL(ref_info, blocks(n), dim3(kThreads), 0, stream, n, id, run_start, strand_pre, alive, info);
  SA_HIP_TRY(hipGetLastError());

  // matches: count, offsets, total
  ChainArgs a{b->d_ref, n, id, run_start, run_value, bucket, strand_pre, alive, info, rid, b->d_reads, b->d_read_off, b->n_reads,
              b->n_read_triples};
  hipLaunchKernelGGL(match_count, blocks(b->n_read_triples + 1), dim3(kThreads), 0, stream, a, cnt);
  SA_HIP_TRY(hipGetLastError());
  t = pl.scan_read_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, cnt, moff, (uint64_t)0, (size_t)b->n_read_triples + 1,
                                     rocprim::plus<uint64_t>(), stream));
  uint64_t total = 0;  // the number of matches sizes the rest of the workspace
  SA_HIP_TRY(hipMemcpyAsync(&total, moff + b->n_read_triples, 8, hipMemcpyDeviceToHost, stream));
  SA_HIP_TRY(hipStreamSynchronize(stream));
  if (total >= 0xFFFFFFF0ull) return fail(SA_ERR_ARGUMENT, "%llu seed matches: split the batch", (unsigned long long)total);
  size_t sort_tmp = 0;
  {
    uint64_t* d64 = nullptr;
    uint32_t* d32 = nullptr;
    if (rocprim::segmented_radix_sort_keys(nullptr, sort_tmp, d64, d64, (unsigned)std::max<uint64_t>(total, 1),
                                           b->n_reads, d32, d32) != hipSuccess)
      return fail(SA_ERR_HIP, "cannot size the match sort");
  }
  size_t o = pl.fixed_bytes;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += round_up(bytes ? bytes : 1, 256);
    return at;
  };
  const size_t o_keys = take((total + 1) * 8), o_sorted = take((total + 1) * 8), o_parent = take((total + 1) * 4),
               o_tails = take((total + b->n_reads + 1) * 4), o_tailv = take((total + b->n_reads + 1) * 4),
               o_sort_tmp = take(sort_tmp);
  if (needed_bytes) *needed_bytes = o;
  if (workspace_bytes < o)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the %zu needed for %llu matches",
                workspace_bytes, o, (unsigned long long)total);
  uint64_t* keys = (uint64_t*)(ws + o_keys);
  uint64_t* sorted = (uint64_t*)(ws + o_sorted);
  hipLaunchKernelGGL(match_write, blocks(b->n_read_triples), dim3(kThreads), 0, stream, a, moff, keys);
  hipLaunchKernelGGL(read_segments, blocks(b->n_reads), dim3(kThreads), 0, stream, b->d_read_off, b->n_reads,
                     moff, seg_begin, seg_end, b->d_n_matches);
  SA_HIP_TRY(hipGetLastError());
  if (total) {
    t = sort_tmp;
    SA_HIP_TRY(rocprim::segmented_radix_sort_keys(ws + o_sort_tmp, t, keys, sorted, (unsigned)total, b->n_reads,
                                                  seg_begin, seg_end, 0u, 64u, stream));
  }
  hipLaunchKernelGGL(chain_lis, blocks((uint64_t)b->n_reads * 64), dim3(kThreads), 0, stream, sorted, seg_begin, seg_end,
                     b->n_reads, (int32_t*)(ws + o_parent), (int32_t*)(ws + o_tails), (uint32_t*)(ws + o_tailv),
                     b->d_out);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}
