// Device-resident batched MinimizeBinary()/Minimize(): planning, kernels and launches.
#include "minimizer_kernels.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>

namespace sa {
namespace {

constexpr uint32_t kThreads = 256;

// ---- chunk bookkeeping ---------------------------------------------------------------------------
__global__ void min_seq_chunks(const uint64_t* off, uint32_t n, uint32_t k, uint32_t w, int variant,
                               uint32_t* nchunks) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  if (s == n) {
    nchunks[s] = 0;
    return;
  }
  const uint32_t len = (uint32_t)(off[s + 1] - off[s]);
  const uint32_t windows = len - w - k + 2;
  nchunks[s] = (variant == SA_MIN_BINARY) ? (windows + kChunk - 1) / kChunk : 1;
}

// First pass: proves each chunk's entry state (exactly from the sequence start, or by collapsing the
// candidate set during the warm-up) and counts the chunk's candidates.
__global__ __launch_bounds__(kThreads) void min_count(MinArgs a) {
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
  ChunkRef c;
  if (chunk >= a.max_chunks) return;
  a.counts[chunk] = 0;
  a.resolved[chunk] = 1;
  if (!locate_chunk(a, chunk, &c)) return;
  const uint32_t k = a.k, w = a.w;
  uint32_t pf = 0, pc = 0;
  if (c.first_window > kWarm) {
    const uint32_t i_s = c.first_window - kWarm;
    uint32_t j = i_s + w - 1;
    BaseStream bs;
    bs.init(a.seq + c.base, (j >= k + 1) ? j - k - 1 : 0, k);
    uint32_t Mf = (k + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (k + 1)) - 1u);
    if (j < k + 1) Mf = (1u << j) - 1u;  // only positions >= 0 exist
    uint32_t Mc = Mf;
    for (uint32_t i = i_s; i < c.first_window; ++i, ++j) {
      bs.advance_to((uint64_t)j + k - 1);
      const uint32_t vj = bs.kmer(j);
      Mf = step_set<true>(bs, j, k, w, vj, Mf);
      Mc = step_set<false>(bs, j, k, w, vj, Mc);
    }
    if (__builtin_popcount(Mf) != 1 || __builtin_popcount(Mc) != 1) {
      a.resolved[chunk] = 0;  // left to min_fix
      return;
    }
    pf = j - 1 - __builtin_ctz(Mf);
    pc = j - 1 - __builtin_ctz(Mc);
  } else if (c.first_window > 0) {
    // close to the start of the sequence: run the few windows before the chunk exactly
    ChunkRef head = c;
    head.first_window = 0;
    head.end_window = c.first_window;
    run_chunk<false>(a, head, pf, pc, nullptr);
  }
  a.counts[chunk] = run_chunk<false>(a, c, pf, pc, nullptr);
  a.exit_f[chunk] = pf;
  a.exit_c[chunk] = pc;
}

// Second pass, one thread per sequence: chunks whose candidate set did not collapse take their entry
// state from the (now final) exit state of the chunk before them, in order.
__global__ __launch_bounds__(kThreads) void min_fix(MinArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n) return;
  for (uint32_t chunk = a.chunk_base[s]; chunk < a.chunk_base[s + 1]; ++chunk) {
    if (a.resolved[chunk]) continue;
    ChunkRef c;
    locate_chunk(a, chunk, &c);
    uint32_t pf = a.exit_f[chunk - 1], pc = a.exit_c[chunk - 1];
    a.counts[chunk] = run_chunk<false>(a, c, pf, pc, nullptr);
    a.exit_f[chunk] = pf;
    a.exit_c[chunk] = pc;
    a.resolved[chunk] = 1;
  }
}

// Third pass: every chunk's entry state is the exit state of its predecessor; write the keys.
__global__ __launch_bounds__(kThreads) void min_emit(MinArgs a) {
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
  ChunkRef c;
  if (chunk >= a.max_chunks || !locate_chunk(a, chunk, &c)) return;
  uint32_t pf = 0, pc = 0;
  if (c.first_window > 0) {
    pf = a.exit_f[chunk - 1];
    pc = a.exit_c[chunk - 1];
  }
  run_chunk<true>(a, c, pf, pc, a.keys + a.cand_off[chunk]);
}

// ---- decimal variant (reference src/minimizer.cpp:124-214), one thread per sequence ----------------
// k-mer value = the k base codes read as a decimal number (stoi of the digit string, :74,77,91,114).
// Window i: when the previous minimum's index is not i-1 only the newest k-mer is compared (strict <,
// :86-105); otherwise the window is re-scanned starting from the sentinel 444 with a strict > test
// (:107-122), which keeps the stale index when no value is below 444.  The first window is a plain
// leftmost strict minimum (:71-84); its index variable is uninitialised in the reference when the first
// k-mer is the minimum and is taken as 0 here.
struct DecStrand {
  int pre_min, pre_idx, cur_min, cur_idx;
};

template <bool WRITE>
__device__ uint32_t run_decimal(const MinArgs& a, uint32_t s, uint64_t* out, bool* range_error) {
  const uint8_t* seq = a.seq + a.off[s];
  const uint32_t len = (uint32_t)(a.off[s + 1] - a.off[s]);
  const uint32_t k = a.k, w = a.w;
  const uint32_t windows = len - (w + k - 1) + 1;
  auto value = [&](uint32_t x, bool fwd, int* v) -> bool {
    unsigned long long acc = 0;
    for (uint32_t t = 0; t < k; ++t) {
      const uint32_t c = base_code(seq[x + t]);
      acc = acc * 10 + (fwd ? c : 3 - c);
    }
    *v = (int)acc;
    return acc <= 2147483647ull;
  };
  DecStrand st[2];
  bool ok = true;
  for (int h = 0; h < 2; ++h) {
    DecStrand& x = st[h];
    x.pre_idx = 0;
    ok &= value(0, h == 0, &x.pre_min);
    for (uint32_t i = 1; i < w; ++i) {
      int v;
      ok &= value(i, h == 0, &v);
      if (x.pre_min > v) {
        x.pre_min = v;
        x.pre_idx = (int)i;
      }
    }
    x.cur_min = x.pre_min;
    x.cur_idx = x.pre_idx;
  }
  uint32_t n_out = 0;
  uint64_t last = ~0ull;
  auto emit = [&](int fmin, int fidx, int cmin, int cidx) {
    const uint64_t key = (fmin < cmin) ? make_key((uint32_t)fmin, (uint32_t)fidx, 1, a.key_shift)
                                       : make_key((uint32_t)cmin, (uint32_t)cidx, 0, a.key_shift);
    if (key != last) {
      if (WRITE) out[n_out] = key;
      ++n_out;
      last = key;
    }
  };
  emit(st[0].pre_min, st[0].pre_idx, st[1].pre_min, st[1].pre_idx);
  for (uint32_t i = 1; i < windows; ++i) {
    for (int h = 0; h < 2; ++h) {
      DecStrand& x = st[h];
      if (x.pre_idx != (int)i - 1) {
        int v;
        ok &= value(i + w - 1, h == 0, &v);
        if (v < x.pre_min) {
          x.cur_min = v;
          x.cur_idx = (int)(i + w - 1);
        } else {
          x.cur_min = x.pre_min;
          x.cur_idx = x.pre_idx;
        }
      } else {
        x.cur_min = 444;
        for (uint32_t r = 0; r < w; ++r) {
          int v;
          ok &= value(i + r, h == 0, &v);
          if (x.cur_min > v) {
            x.cur_min = v;
            x.cur_idx = (int)(i + r);
          }
        }
      }
    }
    emit(st[0].cur_min, st[0].cur_idx, st[1].cur_min, st[1].cur_idx);
    for (int h = 0; h < 2; ++h) {
      st[h].pre_min = st[h].cur_min;
      st[h].pre_idx = st[h].cur_idx;
    }
  }
  *range_error = !ok;
  return n_out;
}

__global__ __launch_bounds__(kThreads) void dec_count(MinArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.max_chunks) return;
  a.counts[s] = 0;
  if (s >= a.n) return;
  bool bad = false;
  a.counts[s] = run_decimal<false>(a, s, nullptr, &bad);
  if (bad && a.status) atomicMax(a.status, SA_ERR_RANGE);
}

__global__ __launch_bounds__(kThreads) void dec_emit(MinArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n) return;
  bool bad = false;
  run_decimal<true>(a, s, a.keys + a.cand_off[s], &bad);
}

// ---- sort / unique / output ---------------------------------------------------------------------
__global__ void min_segments(const uint32_t* chunk_base, const uint64_t* cand_off, uint32_t n,
                             uint32_t* seg_begin, uint32_t* seg_end) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  seg_begin[s] = (uint32_t)cand_off[chunk_base[s]];
  seg_end[s] = (uint32_t)cand_off[chunk_base[s + 1]];
}

__global__ void min_heads(const uint32_t* seg_begin, const uint32_t* seg_end, uint32_t n, uint8_t* flags) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (seg_end[s] > seg_begin[s]) flags[seg_begin[s]] = 2;  // marks a segment start
}

// flags[i] = 1 where sorted key i starts a run of equal keys inside its sequence's segment
__global__ void min_flags(const uint64_t* sorted, const uint64_t* cand_off, uint32_t last_chunk,
                          uint8_t* flags) {
  const uint64_t total = cand_off[last_chunk];
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x)
    flags[i] = (flags[i] == 2 || sorted[i] != sorted[i - (i ? 1 : 0)] || i == 0) ? 1 : 0;
}

__global__ void min_zero_tail(const uint64_t* cand_off, uint32_t last_chunk, uint64_t capacity,
                              uint8_t* flags) {
  const uint64_t total = cand_off[last_chunk];
  for (uint64_t i = total + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < capacity;
       i += (uint64_t)gridDim.x * blockDim.x)
    flags[i] = 0;
}

__global__ void min_out_offsets(const uint32_t* seg_begin, const uint64_t* upos, uint32_t n,
                                const uint64_t* cand_off, uint32_t last_chunk, uint64_t out_capacity,
                                uint64_t* out_off, int* status) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  const uint64_t total = cand_off[last_chunk];
  const uint64_t v = (s == n) ? upos[total] : upos[seg_begin[s]];
  out_off[s] = v;
  if (s == n && v > out_capacity && status) atomicMax(status, SA_ERR_CAPACITY);
}

__global__ void min_write(const uint64_t* sorted, const uint8_t* flags, const uint64_t* upos,
                          const uint64_t* cand_off, uint32_t last_chunk, uint32_t key_shift,
                          uint64_t out_capacity, uint32_t* out) {
  const uint64_t total = cand_off[last_chunk];
  const uint64_t low = (1ull << key_shift) - 1;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const uint64_t o = upos[i];
    if (o >= out_capacity) continue;
    const uint64_t key = sorted[i];
    out[3 * o] = (uint32_t)(key >> key_shift);
    out[3 * o + 1] = (uint32_t)((key & low) >> 1);
    out[3 * o + 2] = (uint32_t)(key & 1);
  }
}

struct MinPlan {
  uint32_t max_chunks;
  uint64_t capacity;  // candidate keys (upper bound: one per window)
  uint32_t key_shift, end_bit;
  size_t scan32_tmp, scan64_tmp, flagscan_tmp, sort_tmp;
  size_t total_bytes;
  // workspace offsets
  size_t o_nchunks, o_chunk_base, o_counts, o_exit_f, o_exit_c, o_resolved, o_cand_off, o_keys, o_sorted,
      o_flags, o_upos, o_seg_begin, o_seg_end, o_tmp;
};

uint32_t bits_for(uint64_t v) {
  uint32_t b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b ? b : 1;
}

int make_min_plan(uint32_t n, uint64_t total_len, uint32_t max_len, uint32_t k, uint32_t w, int variant,
                  MinPlan* pl) {
  if (k < 1 || k > 16 || w < 1) return SA_ERR_ARGUMENT;
  if (variant == SA_MIN_BINARY && w > k + 1) return SA_ERR_ARGUMENT;
  if (variant == SA_MIN_DECIMAL && k > 10) return SA_ERR_RANGE;
  const uint64_t span = (uint64_t)k + w - 2;
  pl->capacity = total_len > span * n ? total_len - span * n : 1;  // total windows
  if (pl->capacity >= 0xFFFFFFF0ull) return SA_ERR_ARGUMENT;
  pl->max_chunks = (variant == SA_MIN_BINARY) ? (uint32_t)(total_len / kChunk + n) : n;
  const uint32_t value_bits = (variant == SA_MIN_BINARY) ? 2 * k : 31;
  pl->key_shift = bits_for(max_len) + 1;
  pl->end_bit = pl->key_shift + value_bits;
  if (pl->end_bit > 64) return SA_ERR_ARGUMENT;
  const size_t mc = pl->max_chunks + 1;
  // rocprim temporary storage sizes
  uint32_t* d32 = nullptr;
  uint64_t* d64 = nullptr;
  uint8_t* d8 = nullptr;
  pl->scan32_tmp = pl->scan64_tmp = pl->flagscan_tmp = pl->sort_tmp = 0;
  if (rocprim::exclusive_scan(nullptr, pl->scan32_tmp, d32, d32, 0u, (size_t)n + 1, rocprim::plus<uint32_t>()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::exclusive_scan(nullptr, pl->scan64_tmp, d32, d64, (uint64_t)0, mc, rocprim::plus<uint64_t>()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::exclusive_scan(nullptr, pl->flagscan_tmp, d8, d64, (uint64_t)0, (size_t)pl->capacity + 1, rocprim::plus<uint64_t>()) != hipSuccess) return SA_ERR_HIP;
  if (n == 1) {
    if (rocprim::radix_sort_keys(nullptr, pl->sort_tmp, d64, d64, (size_t)pl->capacity, 0u, pl->end_bit) != hipSuccess) return SA_ERR_HIP;
  } else {
    if (rocprim::segmented_radix_sort_keys(nullptr, pl->sort_tmp, d64, d64, (unsigned)pl->capacity, n, d32, d32, 0u, pl->end_bit) != hipSuccess) return SA_ERR_HIP;
  }
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += round_up(bytes ? bytes : 1, 256);
    return at;
  };
  pl->o_nchunks = take(((size_t)n + 1) * 4);
  pl->o_chunk_base = take(((size_t)n + 1) * 4);
  pl->o_counts = take(mc * 4);
  pl->o_exit_f = take(mc * 4);
  pl->o_exit_c = take(mc * 4);
  pl->o_resolved = take(mc);
  pl->o_cand_off = take(mc * 8);
  pl->o_keys = take((pl->capacity + 1) * 8);
  pl->o_sorted = take((pl->capacity + 1) * 8);
  pl->o_flags = take(pl->capacity + 2);
  pl->o_upos = take((pl->capacity + 2) * 8);
  pl->o_seg_begin = take(((size_t)n + 1) * 4);
  pl->o_seg_end = take(((size_t)n + 1) * 4);
  pl->o_tmp = take(std::max(std::max(pl->scan32_tmp, pl->scan64_tmp), std::max(pl->flagscan_tmp, pl->sort_tmp)));
  pl->total_bytes = o;
  return SA_OK;
}

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" {

size_t sa_minimize_workspace_bytes(uint32_t n, uint64_t total_len, uint32_t max_len, uint32_t k,
                                   uint32_t w, int variant) {
  MinPlan pl;
  if (make_min_plan(n, total_len, max_len, k, w, variant, &pl) != SA_OK) return 0;
  return pl.total_bytes;
}

uint64_t sa_minimize_out_capacity(uint32_t n, uint64_t total_len, uint32_t k, uint32_t w) {
  const uint64_t span = (uint64_t)k + w - 2;
  return total_len > span * n ? total_len - span * n : 1;  // one minimizer per window at most
}

int sa_minimize_batch_device(int device, void* stream_, const sa_minimize_batch* b, void* d_workspace,
                             size_t workspace_bytes) {
  if (!b) return fail(SA_ERR_ARGUMENT, "batch is NULL");
  if (b->n == 0) return SA_OK;
  if (b->variant != SA_MIN_BINARY && b->variant != SA_MIN_DECIMAL)
    return fail(SA_ERR_ARGUMENT, "unknown minimizer variant %d", b->variant);
  if (!b->d_seq || !b->d_off || !b->d_out || !b->d_out_off)
    return fail(SA_ERR_ARGUMENT, "NULL sequence/offset/output pointer");
  if (b->k < 1 || b->k > 16 || b->w < 1 || (b->variant == SA_MIN_BINARY && b->w > b->k + 1))
    return fail(SA_ERR_ARGUMENT, "need 1 <= k <= 16, w >= 1 and (MinimizeBinary) w <= k + 1 (k=%u, w=%u)",
                b->k, b->w);
  if (b->min_len < b->k + b->w - 1)
    return fail(SA_ERR_ARGUMENT, "every sequence must have at least k + w - 1 = %u bases (min_len=%u)",
                b->k + b->w - 1, b->min_len);
  MinPlan pl;
  int rc = make_min_plan(b->n, b->total_len, b->max_len, b->k, b->w, b->variant, &pl);
  if (rc == SA_ERR_RANGE)
    return fail(SA_ERR_RANGE, "decimal Minimize with k=%u: k-mer values exceed INT_MAX", b->k);
  if (rc) return fail(rc, "cannot plan this minimizer batch (n=%u total_len=%llu)", b->n,
                      (unsigned long long)b->total_len);
  if (!d_workspace || workspace_bytes < pl.total_bytes)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the required %zu", workspace_bytes,
                pl.total_bytes);
  SA_HIP_TRY(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)d_workspace;
  uint32_t* nchunks = (uint32_t*)(ws + pl.o_nchunks);
  uint32_t* chunk_base = (uint32_t*)(ws + pl.o_chunk_base);
  uint64_t* cand_off = (uint64_t*)(ws + pl.o_cand_off);
  uint64_t* keys = (uint64_t*)(ws + pl.o_keys);
  uint64_t* sorted = (uint64_t*)(ws + pl.o_sorted);
  uint8_t* flags = (uint8_t*)(ws + pl.o_flags);
  uint64_t* upos = (uint64_t*)(ws + pl.o_upos);
  uint32_t* seg_begin = (uint32_t*)(ws + pl.o_seg_begin);
  uint32_t* seg_end = (uint32_t*)(ws + pl.o_seg_end);
  void* tmp = ws + pl.o_tmp;

  MinArgs a{};
  a.seq = b->d_seq;
  a.off = b->d_off;
  a.n = b->n;
  a.k = b->k;
  a.w = b->w;
  a.key_shift = pl.key_shift;
  a.max_chunks = pl.max_chunks + 1;  // one trailing empty chunk holds the total after the scan
  a.chunk_base = chunk_base;
  a.counts = (uint32_t*)(ws + pl.o_counts);
  a.exit_f = (uint32_t*)(ws + pl.o_exit_f);
  a.exit_c = (uint32_t*)(ws + pl.o_exit_c);
  a.resolved = (uint8_t*)(ws + pl.o_resolved);
  a.cand_off = cand_off;
  a.keys = keys;
  a.status = b->d_status;
  const uint32_t last_chunk = pl.max_chunks;
  if (b->d_status) SA_HIP_TRY(hipMemsetAsync(b->d_status, 0, sizeof(int), stream));

  auto blocks = [](uint64_t items) { return dim3((unsigned)((items + kThreads - 1) / kThreads)); };
  hipLaunchKernelGGL(min_seq_chunks, blocks(b->n + 1), dim3(kThreads), 0, stream, b->d_off, b->n, b->k,
                     b->w, b->variant, nchunks);
  size_t t = pl.scan32_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, nchunks, chunk_base, 0u, (size_t)b->n + 1,
                                     rocprim::plus<uint32_t>(), stream));
  if (b->variant == SA_MIN_BINARY) {
    hipLaunchKernelGGL(min_count, blocks(a.max_chunks), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(min_fix, blocks(b->n), dim3(kThreads), 0, stream, a);
  } else {
    hipLaunchKernelGGL(dec_count, blocks(a.max_chunks), dim3(kThreads), 0, stream, a);
  }
  SA_HIP_TRY(hipGetLastError());
  t = pl.scan64_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, a.counts, cand_off, (uint64_t)0, (size_t)a.max_chunks,
                                     rocprim::plus<uint64_t>(), stream));
  if (b->n == 1)  // keys past the total must sort to the end of the single segment
    SA_HIP_TRY(hipMemsetAsync(keys, 0xFF, (pl.capacity + 1) * 8, stream));
  if (b->variant == SA_MIN_BINARY)
    hipLaunchKernelGGL(min_emit, blocks(a.max_chunks), dim3(kThreads), 0, stream, a);
  else
    hipLaunchKernelGGL(dec_emit, blocks(b->n), dim3(kThreads), 0, stream, a);
  hipLaunchKernelGGL(min_segments, blocks(b->n), dim3(kThreads), 0, stream, chunk_base, cand_off, b->n,
                     seg_begin, seg_end);
  SA_HIP_TRY(hipGetLastError());
  t = pl.sort_tmp;
  if (b->n == 1) {
    SA_HIP_TRY(rocprim::radix_sort_keys(tmp, t, keys, sorted, (size_t)pl.capacity, 0u, pl.end_bit, stream));
  } else {
    SA_HIP_TRY(rocprim::segmented_radix_sort_keys(tmp, t, keys, sorted, (unsigned)pl.capacity, b->n,
                                                  seg_begin, seg_end, 0u, pl.end_bit, stream));
  }
  SA_HIP_TRY(hipMemsetAsync(flags, 0, pl.capacity + 2, stream));
  hipLaunchKernelGGL(min_heads, blocks(b->n), dim3(kThreads), 0, stream, seg_begin, seg_end, b->n, flags);
  const dim3 wide(std::min<uint64_t>((pl.capacity + kThreads - 1) / kThreads, 16384));
  hipLaunchKernelGGL(min_flags, wide, dim3(kThreads), 0, stream, sorted, cand_off, last_chunk, flags);
  SA_HIP_TRY(hipGetLastError());
  t = pl.flagscan_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, flags, upos, (uint64_t)0, (size_t)pl.capacity + 1,
                                     rocprim::plus<uint64_t>(), stream));
  hipLaunchKernelGGL(min_out_offsets, blocks(b->n + 1), dim3(kThreads), 0, stream, seg_begin, upos, b->n,
                     cand_off, last_chunk, b->out_capacity, b->d_out_off, b->d_status);
  hipLaunchKernelGGL(min_write, wide, dim3(kThreads), 0, stream, sorted, flags, upos, cand_off,
                     last_chunk, pl.key_shift, b->out_capacity, b->d_out);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

}  // extern "C"
