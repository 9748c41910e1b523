// Device-resident batched MinimizeBinary()/Minimize(): planning, kernels and launches.
#include "minimizer_fast.h"

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>

namespace sa {
namespace {

constexpr uint32_t kThreads = 256;

// ---- chunk bookkeeping ---------------------------------------------------------------------------
__global__ void min_seq_plan(const uint64_t* off, uint32_t n, uint32_t k, uint32_t w, uint32_t chunk,
                             uint32_t* nchunks, uint32_t* windows) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  if (s == n) {
    nchunks[s] = 0;
    windows[s] = 0;
    return;
  }
  const uint32_t len = (uint32_t)(off[s + 1] - off[s]);
  const uint32_t win = len - w - k + 2;
  windows[s] = win;
  nchunks[s] = (win + chunk - 1) / chunk;
}

__device__ __forceinline__ uint64_t load8(const uint8_t* p) {  // any alignment
  return *reinterpret_cast<const uint64_t*>(p);
}

// The bookkeeping of a call on a single sequence in one launch: chunk and window counts with their (trivial)
// prefix sums, counters and status cleared.
__global__ void min_plan_single(const uint64_t* off, uint32_t k, uint32_t w, uint32_t chunk, uint32_t* nchunks,
                                uint32_t* chunk_base, uint32_t* windows, uint32_t* wbase, uint32_t* flags,
                                int* status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t len = (uint32_t)(off[1] - off[0]);
  const uint32_t win = len >= k + w - 1 ? len - (k + w - 1) + 1 : 0;
  const uint32_t nc = (win + chunk - 1) / chunk;
  nchunks[0] = nc;
  nchunks[1] = 0;
  chunk_base[0] = 0;
  chunk_base[1] = nc;
  windows[0] = win;
  windows[1] = 0;
  wbase[0] = 0;
  wbase[1] = win;
  flags[0] = flags[1] = flags[2] = 0;  // unresolved chunks, long sequences, other bytes
  if (status) *status = 0;
}

// ---- key output ---------------------------------------------------------------------------------
// Every chunk owns the key slots of its own windows: chunk c of sequence s writes its keys (at most one per
// window, adjacent repeats dropped) from keys[wbase[s] + first_window] on and leaves their number in
// chunk_count[c].  No lane ever writes into another lane's range, so the output needs no atomics and no
// scan; the consumers (sort_emit_small for reads, bucket_scatter for one long sequence, compact_keys for the
// rest) walk a sequence's chunks.  What is stored is the RAW key: value in the low word, position << 1 |
// strand in the high word (compose_key shifts them together); lanes collect raw keys in LDS (kBuf slots each)
// and the wave writes them out together, kCopyLanes lanes per source lane and 16 bytes per lane, so that a
// store instruction covers eight lanes' runs of up to 128 contiguous bytes instead of 64 scattered keys.
constexpr int kBuf = 16;                 // (36 KB of LDS per block with the flush headers: four blocks per CU)
constexpr uint32_t kFlushAt = kBuf - 5;  // a check every 4 steps: flush when some lane holds more keys than this
constexpr uint32_t kCopyLanes = kBuf / 2;  // lanes that copy one source lane's keys, two keys each
static_assert(kBuf % 2 == 0 && 64 % kCopyLanes == 0, "flush_keys: whole teams of kCopyLanes lanes per wave");

__device__ __forceinline__ uint64_t compose_key(uint64_t raw, uint32_t key_shift) {
  return ((uint64_t)(uint32_t)raw << key_shift) | (raw >> 32);
}

struct KeyOut {
  uint64_t* dst;     // the chunk's first key slot
  uint32_t written;  // keys written so far
  uint32_t cnt;      // keys waiting in LDS
  bool ok;           // false: the lane's keys are dropped (no chunk, or a chunk left to min_fix)
};

struct FlushHeader {  // what a lane tells the lanes that copy its keys
  uint64_t* to;
  uint32_t cnt, pad;
};

// Writes the lanes' buffered entries (x: value << SHIFT | tag bits below, y: position << 1 | strand).  Called by
// all lanes of a wave.
template <uint32_t SHIFT>
__device__ __forceinline__ void flush_keys(const MinArgs& a, uint2 (*buf)[kThreads], FlushHeader* hdr, uint32_t tid,
                                           KeyOut& o) {
  constexpr uint32_t kTeams = 64 / kCopyLanes;
  const uint32_t lane = tid & 63, team = lane / kCopyLanes, idx = lane % kCopyLanes;
  FlushHeader mine;
  mine.to = o.dst + o.written;
  mine.cnt = o.ok ? o.cnt : 0;
  mine.pad = 0;
  hdr[tid] = mine;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS accesses of one wave execute in program order)
  __builtin_amdgcn_wave_barrier();
#pragma unroll 4
  for (uint32_t g = 0; g < 64 / kTeams; ++g) {
    const uint32_t src = (tid - lane) + g * kTeams + team;
    const FlushHeader h = hdr[src];
    uint2 e0 = buf[2 * idx][src], e1 = buf[2 * idx + 1][src];
    e0.x >>= SHIFT;
    e1.x >>= SHIFT;
    if (2 * idx + 1 < h.cnt) {
      *reinterpret_cast<uint4*>(h.to + 2 * idx) = make_uint4(e0.x, e0.y, e1.x, e1.y);
    } else if (2 * idx < h.cnt) {
      *reinterpret_cast<uint2*>(h.to + 2 * idx) = e0;
    }
    if (a.hist) {  // (calls on one sequence) the bucketed sort's histogram
      if (2 * idx < h.cnt)
        atomicAdd(&a.hist[fine_slot((uint32_t)(compose_key(((uint64_t)e0.y << 32) | e0.x, a.key_shift) >> a.fine_shift))], 1u);
      if (2 * idx + 1 < h.cnt)
        atomicAdd(&a.hist[fine_slot((uint32_t)(compose_key(((uint64_t)e1.y << 32) | e1.x, a.key_shift) >> a.fine_shift))], 1u);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (o.ok) o.written += o.cnt;
  o.cnt = 0;
}

// 16 bytes from p, any alignment; bytes at or beyond `end` read as 0 (the slow branch is taken by the last
// lanes of the last sequence only)
__device__ __forceinline__ uint4 load16_guard(const uint8_t* p, const uint8_t* end) {
  if (p + 16 <= end) return *reinterpret_cast<const uint4*>(p);
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // byte by byte, last byte first: a 128-bit shift register
#pragma nounroll
  for (int x = 15; x >= 0; --x) {
    const uint32_t byte = (p + x < end) ? (uint32_t)p[x] : 0u;
    w0 = (w0 << 8) | (w1 >> 24);
    w1 = (w1 << 8) | (w2 >> 24);
    w2 = (w2 << 8) | (w3 >> 24);
    w3 = (w3 << 8) | byte;
  }
  return make_uint4(w3, w2, w1, w0);  // byte 0 ended in the low byte of w3
}

// A lane's bases in order, 16 bytes per load with the next 16 already on their way: the steps in front of a
// chunk (history, candidate sets) would otherwise wait for a byte load each.
struct ByteStream {
  const uint8_t* p;    // address of the next byte
  const uint8_t* end;
  uint4 cur, nxt;      // cur: the bytes from p on, shifted down as they are taken; nxt: the 16 bytes after cur's
  uint32_t left;       // bytes left in cur
  __device__ __forceinline__ void open(const uint8_t* from, const uint8_t* end_) {
    p = from;
    end = end_;
    cur = load16_guard(p, end);
    nxt = load16_guard(p + 16, end);
    left = 16;
  }
  __device__ __forceinline__ uint32_t next() {
    if (left == 0) {
      cur = nxt;
      nxt = load16_guard(p + 16, end);
      left = 16;
    }
    const uint32_t byte = cur.x & 0xFFu;
    cur.x = __builtin_amdgcn_alignbit(cur.y, cur.x, 8);
    cur.y = __builtin_amdgcn_alignbit(cur.z, cur.y, 8);
    cur.z = __builtin_amdgcn_alignbit(cur.w, cur.z, 8);
    cur.w >>= 8;
    ++p;
    --left;
    return byte;
  }
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
    v = o < v ? o : v;
  }
  return v;
}

// The steps r0 .. r0 + 15 of Fast5 on the 16 bytes `bytes`.
//   MODE 0: windows before every lane's chunk - nothing is emitted
//   MODE 1: every lane emits
//   MODE 2: lanes emit while lo <= r < hi; a lane's state after its last window is kept in ex_pf / ex_pc
template <int MODE, class FM, class M>
__device__ __forceinline__ void fast_block16(FM& fm, uint32_t r0, const uint4& bytes, uint32_t& other, uint2 (*buf)[kThreads],
                                             FlushHeader* hdr, uint32_t tid, KeyOut& o, uint32_t lo, uint32_t hi,
                                             uint32_t& ex_pf, uint32_t& ex_pc, const MinArgs& a, uint32_t j0) {
  constexpr uint32_t S = FM::S;
  const uint32_t base2 = 2u * (j0 + r0);  // stored positions are positions in the sequence
  constexpr uint32_t kSlotBytes = kThreads * 8u;
  uint32_t slot = o.cnt * kSlotBytes + tid * 8u;  // byte offset of buf[o.cnt][tid]
  // Bytes other than bases count as evidence only where they belong to the lane's own windows: the steps run
  // on for lanes whose chunk has ended (and for lanes without a chunk), over bytes of the next sequence or of
  // the slack behind the blob.
  uint32_t cw[4], oth = 0;
  cw[0] = M::feed4(bytes.x, oth);
  cw[1] = M::feed4(bytes.y, oth);
  cw[2] = M::feed4(bytes.z, oth);
  cw[3] = M::feed4(bytes.w, oth);
  if (r0 + 16 <= hi) {
    other |= oth;
  } else if (r0 < hi) {
    const uint32_t ws[4] = {bytes.x, bytes.y, bytes.z, bytes.w};
    for (uint32_t x = 0; x < hi - r0; ++x) other |= other1((ws[x >> 2] >> (8 * (x & 3))) & 0xFFu);
  }
  if (S <= 6) {
#pragma unroll
    for (int x = 0; x < 4; ++x) cw[x] <<= S;
  }
  auto one = [&](auto u_tag) __attribute__((always_inline)) {
    constexpr uint32_t U = decltype(u_tag)::value;
    const uint32_t r = r0 + U;
    uint32_t cs;
    if (S <= 6)
      cs = (cw[U >> 2] >> (8 * (U & 3))) & ((4u << S) - 1u);  // (the S bits below the code are zero)
    else
      cs = ((cw[U >> 2] >> (8 * (U & 3))) & 3u) << S;
    uint32_t word, p2;
    const bool fresh = fm.template step<U, MODE != 0>(cs, &word, &p2);
    // every window's tuple goes to the lane's next free slot; the slot only moves on when the tuple is new
    if (MODE != 0) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(&buf[0][0]) + slot) = make_uint2(word, p2 + base2);
    if (MODE == 1) {
      slot += fresh ? kSlotBytes : 0u;
    } else if (MODE == 2) {
      slot += (fresh && (r - lo) < (hi - lo)) ? kSlotBytes : 0u;
      const bool last = r + 1 == hi;
      ex_pf = last ? fm.pf + r0 : ex_pf;
      ex_pc = last ? fm.pc + r0 : ex_pc;
    }
    if (MODE != 0 && (U & 3) == 3) {
      o.cnt = (slot - tid * 8u) / kSlotBytes;
      if (__any(o.cnt > kFlushAt)) {
        flush_keys<S>(a, buf, hdr, tid, o);
        slot = tid * 8u;
      }
    }
  };
#define SA_FAST_STEP(X) one(std::integral_constant<uint32_t, X>{});
  SA_FAST_STEP(0) SA_FAST_STEP(1) SA_FAST_STEP(2) SA_FAST_STEP(3) SA_FAST_STEP(4) SA_FAST_STEP(5) SA_FAST_STEP(6)
  SA_FAST_STEP(7) SA_FAST_STEP(8) SA_FAST_STEP(9) SA_FAST_STEP(10) SA_FAST_STEP(11) SA_FAST_STEP(12) SA_FAST_STEP(13)
  SA_FAST_STEP(14) SA_FAST_STEP(15)
#undef SA_FAST_STEP
  if (MODE != 0) o.cnt = (slot - tid * 8u) / kSlotBytes;
  fm.rebase();
}

// One thread per chunk: proves the chunk's entry state (exactly from the sequence start, or by collapsing
// the candidate sets during the warm-up), runs the chunk's windows and writes their keys, adjacent
// duplicates dropped.
// M = Machine<W>: the fast machine, exact as long as every byte is a base - it reports other bytes in
// a.other.  M = MachineX: the exact machine for such batches; returns at once unless a.other is set.
// FASTK != 0 (with M = Machine<5>, k = FASTK): once the entry state is known the windows run on Fast5<FASTK>,
// all lanes of a wave in step.
template <class M, uint32_t FASTK>
__global__ __launch_bounds__(kThreads) void min_run(MinArgs a, uint32_t* n_unresolved) {
  __shared__ uint2 buf[kBuf][kThreads];  // buffered tuples
  __shared__ FlushHeader hdr[kThreads];
  if (M::kExact && *a.other == 0) return;
  uint32_t other = 0;
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = blockIdx.x * kThreads + tid;
  ChunkRef c;
  const bool valid = locate_chunk(a, chunk, &c);
  const uint8_t* s = a.seq + (valid ? c.base : 0);
  const uint8_t* blob_end = a.seq + a.off[a.n] + 8;  // the blob is readable 8 bytes past its last sequence
  const uint32_t F = valid ? c.first_window : 0, E = valid ? c.end_window : 0;
  const uint32_t k = a.k, w = a.w;
  KeyOut o;
  o.dst = a.keys + (valid ? (uint64_t)a.wbase[c.seq] + F : 0);
  o.written = o.cnt = 0;
  o.ok = valid;
  M m;
  m.reset(k, w);
  uint32_t i = E;  // next window
  uint32_t lastv = 0, lastp = ~0u;  // the tuple written last (no real tuple has position bits ~0)
  auto emit = [&]() __attribute__((always_inline)) {
    uint32_t v, p2;
    m.tuple(&v, &p2);
    if (v != lastv || p2 != lastp) {
      buf[o.cnt][tid] = make_uint2(v, p2);
      ++o.cnt;
      lastv = v;
      lastp = p2;
    }
  };
  bool live = valid, sets = false;
  uint32_t Mf = 1, Mc = 1;
  // The bases in front of the chunk: close to the sequence start (the warm-up would begin before k-mer k + 1,
  // whose candidates reach back to position 0) the windows before the chunk run exactly, from window 0;
  // elsewhere the chunk starts kWarm windows early with every candidate state of step j, the positions
  // j-k-1 .. j-1.
  const bool from_start = valid && F <= kWarm + k + 2;
  uint32_t first_base = 0, n_push = 0;  // the history: bases first_base .. first_base + n_push - 1
  if (valid) {
    if (from_start) {
      n_push = w + k - 1;
    } else {
      i = F - kWarm;
      first_base = (i + w - 1) - k - 1;
      n_push = 2 * k;
    }
  }
  ByteStream in;
  in.open(s + first_base, blob_end);
  {
    const uint32_t max_push = wave_max_u32(n_push);
    for (uint32_t b = 0; b < max_push; ++b)
      if (b < n_push) m.push(M::feed1(in.next(), other));
  }
  if (valid) {
    if (from_start) {
      m.first_window();
      if (F == 0) emit();
      i = 1;
    } else {
      m.j = i + w - 1;
      m.init_window();
      Mf = Mc = (2u << k) - 1u;
      sets = true;
    }
  }
  for (uint32_t guard = 0; guard < kWarm; ++guard) {
    if (!__any(sets && (__builtin_popcount(Mf) != 1 || __builtin_popcount(Mc) != 1))) break;
    if (sets) {
      m.step_sets(M::feed1(in.next(), other), Mf, Mc);
      ++i;
    }
  }
  if (sets) {
    if (__builtin_popcount(Mf) == 1 && __builtin_popcount(Mc) == 1)
      m.enter(m.j, m.j - 1 - __builtin_ctz(Mf), m.j - 1 - __builtin_ctz(Mc));
    else
      live = false;  // left to min_fix
  }
  o.ok = live;
  if (i > E) i = E;  // (a one-window chunk at the sequence start)
  uint32_t exit_pf = m.pf, exit_pc = m.pc;
  if constexpr (FASTK != 0) {
    // Every lane's next window is i (its step r handles window i + r); lanes emit for lo <= r < hi.
    using FM = Fast5<(FASTK ? FASTK : 15)>;
    const uint32_t j0 = m.j;
    const uint32_t lo = live ? (F > i ? F - i : 0u) : 0u, hi = live ? E - i : 0u;
    if (__any(o.cnt > 0)) flush_keys<0>(a, buf, hdr, tid, o);
    FM fm;
    fm.enter(m, j0, lastp);
    const uint32_t lo_min = __builtin_amdgcn_readfirstlane(wave_min_u32(live ? lo : ~0u));
    const uint32_t lo_max = __builtin_amdgcn_readfirstlane(wave_max_u32(live ? lo : 0u));
    const uint32_t hi_min = __builtin_amdgcn_readfirstlane(wave_min_u32(live ? hi : ~0u));
    const uint32_t hi_max = __builtin_amdgcn_readfirstlane(wave_max_u32(live ? hi : 0u));
    const uint32_t end_all = (hi_max + 15u) & ~15u;                       // 0 when no lane is live
    const uint32_t end_a = lo_min == ~0u ? 0u : (lo_min & ~15u);          // [0, end_a): before every chunk
    uint32_t end_b = (lo_max + 15u) & ~15u;                               // [end_a, end_b): some lanes emit
    uint32_t end_c = hi_min == ~0u ? 0u : (hi_min & ~15u);                // [end_b, end_c): all lanes emit
    if (end_b > end_all) end_b = end_all;
    if (end_c < end_b) end_c = end_b;
    if (end_c > end_all) end_c = end_all;
    const uint8_t* bp = s + (j0 + FM::kK - 1);  // the base step r consumes is bp[r]
    uint32_t ex_pf = fm.pf, ex_pc = fm.pc;
    uint4 cur = load16_guard(bp, blob_end);
    uint32_t r0 = 0;
    for (; r0 < end_a; r0 += 16) {
      const uint4 nxt = load16_guard(bp + r0 + 16, blob_end);
      fast_block16<0, FM, M>(fm, r0, cur, other, buf, hdr, tid, o, lo, hi, ex_pf, ex_pc, a, j0);
      cur = nxt;
    }
    if (end_a) fm.lastp = FM::kNoTuple;  // (the steps above do not keep it)
    for (; r0 < end_b; r0 += 16) {
      const uint4 nxt = load16_guard(bp + r0 + 16, blob_end);
      fast_block16<2, FM, M>(fm, r0, cur, other, buf, hdr, tid, o, lo, hi, ex_pf, ex_pc, a, j0);
      cur = nxt;
    }
    for (; r0 < end_c; r0 += 16) {
      const uint4 nxt = load16_guard(bp + r0 + 16, blob_end);
      fast_block16<1, FM, M>(fm, r0, cur, other, buf, hdr, tid, o, lo, hi, ex_pf, ex_pc, a, j0);
      cur = nxt;
    }
    if (hi == r0) {  // the lane's last window was the last step of the stretch in which every lane emits
      ex_pf = fm.pf + r0;
      ex_pc = fm.pc + r0;
    }
    for (; r0 < end_all; r0 += 16) {
      const uint4 nxt = load16_guard(bp + r0 + 16, blob_end);
      fast_block16<2, FM, M>(fm, r0, cur, other, buf, hdr, tid, o, lo, hi, ex_pf, ex_pc, a, j0);
      cur = nxt;
    }
    if (__any(o.cnt > 0)) flush_keys<FM::S>(a, buf, hdr, tid, o);
    exit_pf = ex_pf + j0;
    exit_pc = ex_pc + j0;
  } else {
    // single steps until the windows left before the chunk are a multiple of 8: a group of 8 steps below
    // then lies entirely before the chunk (nothing to emit) or entirely inside it
    {
      const uint32_t align = (live && i < F) ? ((F - i) & 7u) : 0;
      const uint32_t max_align = wave_max_u32(align);
      for (uint32_t u = 0; u < max_align; ++u) {
        if (u < align) {
          m.step(M::feed1(s[m.j + k - 1], other));
          ++i;
        }
      }
    }
    const uint32_t rem = live ? E - i : 0;
    const uint32_t groups = rem >> 3, tail = rem & 7;
    const uint32_t max_groups = wave_max_u32(groups), max_tail = wave_max_u32(tail);
    const uint8_t* next_base = s + (m.j + k - 1);
    uint64_t word = rem ? load8(next_base) : 0;
    for (uint32_t g = 0; g < max_groups; ++g) {
      if (g < groups) {
        const uint32_t c0 = M::feed4((uint32_t)word, other), c1 = M::feed4((uint32_t)(word >> 32), other);
        next_base += 8;
        if (g + 1 < groups || tail) word = load8(next_base);
        if (i >= F) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            m.step(((u < 4 ? c0 : c1) >> (8 * (u & 3))) & M::kFeedMask);
            emit();
          }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) m.step(((u < 4 ? c0 : c1) >> (8 * (u & 3))) & M::kFeedMask);
        }
        i += 8;
      }
      if (__any(o.cnt > (uint32_t)(kBuf - 8))) flush_keys<0>(a, buf, hdr, tid, o);
    }
    for (uint32_t u = 0; u < max_tail; ++u) {
      if (u < tail) {
        m.step(M::feed1((uint32_t)(word >> (8 * u)) & 0xFFu, other));
        emit();
      }
    }
    if (__any(o.cnt > 0)) flush_keys<0>(a, buf, hdr, tid, o);
    exit_pf = m.pf;
    exit_pc = m.pc;
  }
  if (valid) {
    a.chunk_count[chunk] = o.written;
    a.exit_f[chunk] = exit_pf;
    a.exit_c[chunk] = exit_pc;
    a.resolved[chunk] = live ? 1 : 0;
    if (!live) atomicAdd(n_unresolved, 1u);
  }
  if (!M::kExact && other) atomicOr(a.other, 1u);
}

// One wave per sequence, only when some chunk's candidate set did not collapse: those chunks take their
// entry state from the (final) exit state of the chunk before them, in order, and write their keys anew.
// Lane 0 runs the machine.
// A batch in which the first run met bytes other than bases runs again on the exact machine: the first fix
// kernel then only forgets the unresolved chunks and the histogram of the first run (every chunk writes its
// keys and their number again).
template <class M>
__global__ __launch_bounds__(kThreads) void min_fix(MinArgs a, uint32_t* n_unresolved, uint32_t hist_words) {
  if (!M::kExact && *a.other) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i == 0) *n_unresolved = 0;
    if (a.hist)  // (single-sequence calls) the keys counted so far
      for (uint32_t x = i; x < hist_words; x += gridDim.x * kThreads) a.hist[x] = 0;
    return;
  }
  if (*n_unresolved == 0) return;
  if (M::kExact && *a.other == 0) return;
  const uint32_t s = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (s >= a.n) return;
  const uint32_t c_begin = a.chunk_base[s], c_end = a.chunk_base[s + 1];
  const uint8_t* seq = a.seq + a.off[s];
  const uint32_t windows = (uint32_t)(a.off[s + 1] - a.off[s]) - a.w - a.k + 2;
  const uint32_t k = a.k, w = a.w;
  for (uint32_t c0 = c_begin; c0 < c_end; c0 += 64) {
    const uint32_t mine = c0 + lane;
    uint64_t todo = __ballot(mine < c_end && !a.resolved[mine]);
    while (todo) {
      const uint32_t chunk = c0 + (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1;
      if (lane == 0) {
        const uint32_t F = (chunk - c_begin) * a.chunk;
        const uint32_t E = F + a.chunk < windows ? F + a.chunk : windows;
        uint64_t* out = a.keys + (uint64_t)a.wbase[s] + F;
        M m;
        uint32_t other = 0;  // the run kernel has looked at every byte already
        m.reset(k, w);
        const uint32_t j = F + w - 1;
        for (uint32_t b = j - k - 1; b + 1 < j + k; ++b) m.push(M::feed1(seq[b], other));
        m.enter(j, a.exit_f[chunk - 1], a.exit_c[chunk - 1]);
        uint32_t n_out = 0;
        uint64_t last = ~0ull;
        for (uint32_t i = F; i < E; ++i) {
          m.step(M::feed1(seq[m.j + k - 1], other));
          uint32_t v, p2;
          m.tuple(&v, &p2);
          const uint64_t raw = ((uint64_t)p2 << 32) | v;
          if (raw != last) {
            out[n_out++] = raw;
            if (a.hist) atomicAdd(&a.hist[fine_slot((uint32_t)(compose_key(raw, a.key_shift) >> a.fine_shift))], 1u);
            last = raw;
          }
        }
        a.chunk_count[chunk] = n_out;
        a.exit_f[chunk] = m.pf;
        a.exit_c[chunk] = m.pc;
        a.resolved[chunk] = 1;
      }
    }
  }
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

// A byte other than ACGT / acgt: the reference drops such bytes from its digit strings but keeps counting windows
// by the original length (src/minimizer.cpp:13-40, 174), so later windows see truncated k-mers and the call may
// throw from substr / stoi; that is not reproduced - the byte is reported and the call fails.
template <bool WRITE>
__device__ uint32_t run_decimal(const MinArgs& a, uint32_t s, uint64_t* out, bool* range_error, bool* other_byte) {
  const uint8_t* seq = a.seq + a.off[s];
  const uint32_t len = (uint32_t)(a.off[s + 1] - a.off[s]);
  const uint32_t k = a.k, w = a.w;
  const uint32_t windows = len - (w + k - 1) + 1;
  uint32_t other = 0;
  auto value = [&](uint32_t x, bool fwd, int* v) -> bool {
    unsigned long long acc = 0;
    for (uint32_t t = 0; t < k; ++t) {
      const uint32_t c = base_code(seq[x + t]);
      other |= other1(seq[x + t]);
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
    const uint64_t key = (fmin < cmin) ? (((uint64_t)((uint32_t)fidx << 1 | 1u) << 32) | (uint32_t)fmin)
                                       : (((uint64_t)((uint32_t)cidx << 1) << 32) | (uint32_t)cmin);  // raw
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
  *other_byte = other != 0;
  return n_out;
}

__global__ __launch_bounds__(kThreads) void dec_run(MinArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n) return;
  bool bad = false, other = false;
  a.chunk_count[a.chunk_base[s]] = run_decimal<true>(a, s, a.keys + a.wbase[s], &bad, &other);  // one chunk per sequence
  if (bad && a.status) atomicMax(a.status, SA_ERR_RANGE);
  if (other && a.status) atomicMax(a.status, SA_ERR_ARGUMENT);
}

// ---- sort / unique / output ---------------------------------------------------------------------
// The sort and unique kernels want a sequence's keys in one piece: compact_keys moves every chunk's keys to
// cscan[chunk] (the exclusive scan of the chunks' key counts) of a second array, where sequence s then owns
// [cscan[chunk_base[s]], cscan[chunk_base[s + 1]]).  Batches of reads skip this (sort_emit_small collects a
// read's chunks itself): with skip_unless_big set the kernel returns at once unless some sequence has more
// than kTile keys.
__global__ __launch_bounds__(kThreads) void compact_keys(MinArgs a, const uint32_t* cscan, uint64_t* dst,
                                                         const uint32_t* skip_unless_big) {
  if (skip_unless_big && *skip_unless_big == 0) return;
  const uint32_t nchunks = a.chunk_base[a.n];
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    ChunkRef c;
    locate_chunk(a, chunk, &c);
    const uint64_t* src = a.keys + (uint64_t)a.wbase[c.seq] + c.first_window;
    const uint32_t count = a.chunk_count[chunk], to = cscan[chunk];
    for (uint32_t x = threadIdx.x; x < count; x += kThreads) dst[to + x] = compose_key(src[x], a.key_shift);
  }
}

// Reads: a sequence's number of keys from its (few) chunks, and the number of sequences with more than
// kTile keys, which decides between sort_emit_small and the several-kernel path.
__global__ void min_count_direct(MinArgs a, uint32_t* seq_count, uint32_t* n_big) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n) return;
  uint32_t c = 0;
  for (uint32_t x = a.chunk_base[s]; x < a.chunk_base[s + 1]; ++x) c += a.chunk_count[x];
  seq_count[s] = c;
  if (c > kTile) atomicAdd(n_big, 1u);
}

// big_end / n_big: sequences with more than kTile keys are "big" - they are left to the segmented radix sort,
// whose segment s is [seg_begin[s], big_end[s]) (empty for the others, which are sorted inside LDS).
// count_big = 0: n_big has been counted already (min_count_direct).
__global__ void min_segments(const uint32_t* cscan, const uint32_t* chunk_base, uint32_t n,
                             uint32_t* seg_begin, uint32_t* seg_end, uint32_t* big_end, uint32_t* tiles,
                             uint32_t* n_big, uint32_t count_big) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > n) return;
  if (s == n) {
    tiles[s] = 0;
    return;
  }
  const uint32_t first = cscan[chunk_base[s]], c = cscan[chunk_base[s + 1]] - first;
  seg_begin[s] = first;
  seg_end[s] = first + c;
  big_end[s] = c > kTile ? first + c : first;
  if (count_big && c > kTile) atomicAdd(n_big, 1u);
  tiles[s] = (c + kTile - 1) / kTile;
}

struct UniqArgs {
  const uint64_t* sorted;
  const uint32_t* seg_begin;
  const uint32_t* seg_end;
  const uint32_t* tile_base;  // n + 1: first tile of every sequence
  uint32_t n;
  uint32_t skip_small;        // uniq_count: tiles of sequences with <= kSmallMax keys are already counted
  const uint32_t* n_big;      // uniq_count: sequences left to the segmented sort (skip_small only)
  uint32_t* tile_count;       // per tile: keys that differ from their predecessor
  const uint64_t* tile_off;   // exclusive scan of tile_count
  uint32_t key_shift;
  uint64_t out_capacity;
  uint32_t* out;
  uint64_t* out_off;
  int* status;
};

constexpr uint32_t kTileItems = kTile / kThreads;

// Keys [*begin, *end) of the sorted array that form tile `tile`, and the tile's sequence.
__device__ __forceinline__ uint32_t locate_tile(const UniqArgs& u, uint32_t tile, uint32_t* begin,
                                                uint32_t* end, uint32_t* seg_first) {
  uint32_t lo = 0, hi = u.n;  // largest s with tile_base[s] <= tile (sequences without keys own no tile)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (u.tile_base[mid] <= tile)
      lo = mid;
    else
      hi = mid;
  }
  *seg_first = u.seg_begin[lo];
  *begin = u.seg_begin[lo] + (tile - u.tile_base[lo]) * kTile;
  const uint32_t e = *begin + kTile;
  *end = e < u.seg_end[lo] ? e : u.seg_end[lo];
  return lo;
}

// flags of the thread's kTileItems consecutive keys, packed into a bit mask; keys[] receives them
__device__ __forceinline__ uint32_t tile_flags(const UniqArgs& u, uint32_t begin, uint32_t end,
                                               uint32_t seg_first, uint64_t* keys) {
  const uint32_t first = begin + threadIdx.x * kTileItems;
  uint64_t prev = (first > seg_first && first < end) ? u.sorted[first - 1] : 0;
  uint32_t flags = 0;
#pragma unroll
  for (uint32_t e = 0; e < kTileItems; ++e) {
    const uint32_t i = first + e;
    if (i < end) {
      keys[e] = u.sorted[i];
      if (i == seg_first || keys[e] != prev) flags |= 1u << e;
      prev = keys[e];
    }
  }
  return flags;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* lds, uint32_t* excl) {
  // exclusive prefix of v over the block's threads and the block total
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if ((int)lane >= d) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t x = 0; x < kThreads / 64; ++x) {
    if (x < wave) before += lds[x];
    total += lds[x];
  }
  __syncthreads();
  *excl = before + inc - v;
  return total;
}

// Sort of one sequence's keys inside LDS, one block per sequence, for sequences with at most kSmallMax
// keys (one tile): a counting sort on a monotone function of the value groups the keys into bins of a
// few keys each, then every key finds its rank among the keys of its bin by comparing.  The same
// comparison tells whether an equal key precedes it, which gives the tile's unique count.
// Longer sequences are left to the segmented radix sort: big_end[s] = seg_begin[s] hides the sorted ones.
constexpr uint32_t kSmallMax = kTile;
constexpr uint32_t kBins = 2048;

// Minimizer values crowd the low end of their range: a window's value is the smallest of about 2 w k-mer values
// (both strands), so its distribution function is about 1 - (1 - x)^(2w) for x = value / 2^value_bits.  The bins
// follow that function (powers by repeated squaring), which spreads the keys of a random sequence evenly over
// them; any non-decreasing function sorts correctly, an uneven one only costs time.  In single precision: every
// step - conversion, 1 - x, squares and products of non-negative numbers, the final truncation - rounds
// monotonically, so the result never decreases with v, and its multiplications run at full rate where the 32-bit
// integer multiply-high of a fixed-point version runs at a quarter (the power for w = 5 is spelled out; on the
// batch of 10^5 reads the kernel takes the same time either way).
__device__ __forceinline__ uint32_t value_bin(uint32_t v, uint32_t value_bits, uint32_t power) {
  if (value_bits <= 11) return v;
  float t = 1.0f - (float)(v << (32 - value_bits)) * 2.3283064365386963e-10f;  // 1 - x, x = value / 2^value_bits
  float acc;
  if (power == 10) {
    const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4;
    acc = t2 * t8;
  } else {
    acc = 1.0f;
    for (uint32_t m = power; m; m >>= 1) {  // (the same trip count in every lane)
      if (m & 1u) acc *= t;
      t *= t;
    }
  }
  const uint32_t b = (uint32_t)((1.0f - acc) * (float)kBins);
  return b < kBins - 1 ? b : kBins - 1;
}

__global__ __launch_bounds__(kThreads) void sort_small(const uint64_t* keys, uint64_t* sorted,
                                                       const uint32_t* seg_begin, const uint32_t* seg_end,
                                                       const uint32_t* tile_base, uint32_t* tile_count,
                                                       uint32_t key_shift, uint32_t value_bits, uint32_t power,
                                                       const uint32_t* n_big) {
  __shared__ uint64_t grouped[kSmallMax];
  __shared__ uint32_t bins[kBins];
  __shared__ uint32_t lds[kThreads / 64];
  if (*n_big == 0) return;  // a batch without big sequences is served by sort_emit_small
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  const uint32_t begin = seg_begin[s], count = seg_end[s] - begin;
  if (count > kSmallMax || count == 0) return;
  constexpr uint32_t kPer = kSmallMax / kThreads, kBinsPer = kBins / kThreads;
#pragma unroll
  for (uint32_t e = 0; e < kBinsPer; ++e) bins[tid + e * kThreads] = 0;
  __syncthreads();
  uint64_t mine[kPer];
  uint32_t bin[kPer];
#pragma unroll
  for (uint32_t e = 0; e < kPer; ++e) {
    const uint32_t i = tid + e * kThreads;
    if (i < count) {
      mine[e] = keys[begin + i];
      bin[e] = value_bin((uint32_t)(mine[e] >> key_shift), value_bits, power);
      atomicAdd(&bins[bin[e]], 1u);
    }
  }
  __syncthreads();
  // exclusive scan of the bin counts: thread t owns bins t*kBinsPer ..
  uint32_t local[kBinsPer], sum = 0;
#pragma unroll
  for (uint32_t e = 0; e < kBinsPer; ++e) {
    local[e] = bins[tid * kBinsPer + e];
    sum += local[e];
  }
  uint32_t excl;
  block_sum(sum, lds, &excl);
#pragma unroll
  for (uint32_t e = 0; e < kBinsPer; ++e) {
    bins[tid * kBinsPer + e] = excl;
    excl += local[e];
  }
  __syncthreads();
#pragma unroll
  for (uint32_t e = 0; e < kPer; ++e) {
    const uint32_t i = tid + e * kThreads;
    if (i < count) grouped[atomicAdd(&bins[bin[e]], 1u)] = mine[e];
  }
  __syncthreads();
  // bins[b] is now the end of bin b
  uint32_t uniq = 0;
  for (uint32_t p = tid; p < count; p += kThreads) {
    const uint64_t key = grouped[p];
    const uint32_t b = value_bin((uint32_t)(key >> key_shift), value_bits, power);
    const uint32_t lo = b ? bins[b - 1] : 0, hi = bins[b];
    uint32_t rank = 0;
    bool dup = false;
    for (uint32_t m = lo; m < hi; ++m) {
      const uint64_t other = grouped[m];
      const bool same_before = other == key && m < p;
      rank += (other < key || same_before) ? 1u : 0u;
      dup |= same_before;
    }
    sorted[begin + lo + rank] = key;
    uniq += dup ? 0u : 1u;
  }
  uint32_t unused;
  const uint32_t total = block_sum(uniq, lds, &unused);
  if (tid == 0) tile_count[tile_base[s]] = total;
}


// ---- sort + unique + output in one kernel, for batches in which every sequence has at most kSmallMax keys
// (reads).  One block per sequence: the counting sort of sort_small, then the block knows its number of
// distinct keys and needs the number of all sequences before it: a single-pass chained scan with look-back
// (every block publishes its own count at once and its inclusive prefix as soon as it knows it; a block adds
// up the counts of its predecessors backwards until it meets one that already has its prefix).  A block
// takes the sequence of its own number (see the kernel).  The triples are staged in LDS and leave as one
// contiguous run: the sorted keys never travel to memory and back.
constexpr uint64_t kScanCount = 1ull << 62, kScanPrefix = 2ull << 62, kScanValue = (1ull << 62) - 1;

struct EmitArgs {
  const uint64_t* keys;
  const uint32_t* seg_begin;
  const uint32_t* seg_end;
  uint32_t n;
  uint32_t key_shift, value_bits;
  uint32_t power;        // exponent of value_bin's distribution function
  uint64_t* scan_state;  // n entries, zeroed: flag << 62 | value
  const uint32_t* n_big; // sequences with more than kSmallMax keys
  uint64_t out_capacity;
  uint32_t* out;
  uint64_t* out_off;
  int* status;
  // buckets != 0: the segments are the value-ordered buckets of ONE sequence (bucket_plan): keys are binned
  // linearly between the bucket's lowest and highest possible key, empty buckets only pass the count on, and
  // out_off gets the sequence's two entries
  uint32_t buckets;
  const uint32_t* nseg_used;  // buckets in use (device side); the launch covers a.n >= that many
  uint32_t fine_shift;
  const uint32_t* seg_lof;    // first fine bin of the bucket, one past its last
  const uint32_t* seg_hif;
  const uint64_t* seg_lokey;  // set by bucket_split for the slices of an oversized bucket (seg_hif == 0 there)
  const uint64_t* seg_span;
  // direct != 0 (reads): sequence s's keys lie where its chunks wrote them - chunk c's chunk_count[c] keys from
  // keys[wbase[s] + (c - chunk_base[s]) * chunk] on, at most kDirectChunks chunks per sequence - and seg_begin
  // holds the sequences' key counts
  uint32_t direct;
  const uint32_t* chunk_base;
  const uint32_t* chunk_count;
  const uint32_t* wbase;
  uint32_t chunk;
};
constexpr uint32_t kDirectChunks = 8;

enum { kEmitPacked = 0, kEmitDirect = 1, kEmitBuckets = 2 };

template <int MODE>
__global__ __launch_bounds__(kThreads) void sort_emit_small(EmitArgs a) {
  // one buffer, two uses: [0, 4096) grouped keys + [4096, 5120) bin counters (16 bits each, two per dword)
  // while sorting, then the staged (value, position << 1 | strand) pairs.  20.3 KB with the small arrays below:
  // seven blocks per CU.
  __shared__ __attribute__((aligned(16))) uint32_t buf[2 * kSmallMax + kBins / 2];
  __shared__ uint32_t lds[kThreads / 64];
  __shared__ uint32_t distinct[kSmallMax / 32];  // bit r: the key of sorted rank r is the first of its value
  __shared__ uint64_t s_excl;
  __shared__ uint64_t lb_sum[kThreads / 64];   // look-back: per wave, the counts it may add
  __shared__ uint32_t lb_flag[kThreads / 64];  //   bit 0: usable, bit 1: ends at a prefix
  __shared__ uint32_t ccount[kDirectChunks];   // direct: the key counts of the sequence's chunks
  uint64_t* grouped = reinterpret_cast<uint64_t*>(buf);
  uint32_t* bins2 = buf + 2 * kSmallMax;  // bin b: half b & 1 of bins2[b >> 1] (counts and positions <= 2048)
  uint32_t* stage = buf;
  static_assert(kBins == kSmallMax && kSmallMax <= 0x8000, "LDS layout, 16-bit bin counters");
  auto bin_get = [&](uint32_t b) -> uint32_t { return (bins2[b >> 1] >> (16 * (b & 1))) & 0xFFFFu; };
  auto bin_inc = [&](uint32_t b) -> uint32_t {  // value before the increment
    return (atomicAdd(&bins2[b >> 1], 1u << (16 * (b & 1))) >> (16 * (b & 1))) & 0xFFFFu;
  };
  const uint32_t tid = threadIdx.x;
  // a batch with big sequences takes the several-kernel path (looked at further down, when the block's other first
  // loads have left as well: one round trip instead of two)
  const uint32_t n_big = MODE != kEmitBuckets ? *a.n_big : 0u;
  const uint32_t nsegs = MODE == kEmitBuckets ? *a.nseg_used : a.n;
  // Which segment a block takes.  A block waits, in the look-back, for the segments before its own, so those must
  // belong to blocks that run or have finished.  It takes its own number - numbers from a ticket counter, a
  // returning atomic per block on ONE word, bound the kernel at about 90 blocks per microsecond (1.1 of 1.8 ms
  // for 10^5 reads, 50 of 66 us for the 4494 bucket numbers of a 4.6 Mbp sequence).  Every XCD starts its
  // share of the blocks in increasing order, so the lowest-numbered unfinished segment always finds a free slot
  // on its XCD (everything before it on that XCD has finished) and needs no other block: the launch cannot stall;
  // the look-back's waits are bounded all the same and fail the call rather than hang.
  for (uint32_t round = 0; round < 1; ++round) {
    const uint32_t s = blockIdx.x;
    if (s >= nsegs) return;
    {
      if (tid < kSmallMax / 32) distinct[tid] = 0;
      __syncthreads();
      const uint32_t begin = MODE == kEmitDirect ? a.wbase[s] : a.seg_begin[s];
      uint32_t count = MODE == kEmitDirect ? a.seg_begin[s] : a.seg_end[s] - begin;  // <= kSmallMax: n_big == 0, bucket_split
      const uint32_t c0 = MODE == kEmitDirect ? a.chunk_base[s] : 0u;
      const uint32_t nch = MODE == kEmitDirect ? a.chunk_base[s + 1] - c0 : 0u;  // <= kDirectChunks (the host checks the bound)
      if (n_big != 0) return;
      if (count > kSmallMax) {  // never expected: fail the call rather than overrun the buffers
        if (tid == 0 && a.status) atomicMax(a.status, SA_ERR_HIP);
        count = kSmallMax;
      }
      constexpr uint32_t kPer = kSmallMax / kThreads, kBinsPer = kBins / kThreads;
      // the bin of a key: reads - a function of the value; buckets - linear between the bucket's bounds
      uint64_t lokey = 0;
      uint32_t bshift = 0;
      if (MODE == kEmitBuckets && count) {
        uint64_t span;
        if (a.seg_hif[s]) {
          lokey = (uint64_t)a.seg_lof[s] << a.fine_shift;
          span = (uint64_t)(a.seg_hif[s] - a.seg_lof[s]) << a.fine_shift;
        } else {
          lokey = a.seg_lokey[s];
          span = a.seg_span[s];
        }
        const uint32_t bits = span > 1 ? 64u - (uint32_t)__builtin_clzll(span - 1) : 0u;
        bshift = bits > 11 ? bits - 11 : 0;
      }
      auto bin_of = [&](uint64_t key) -> uint32_t {
        if (MODE != kEmitBuckets) return value_bin((uint32_t)(key >> a.key_shift), a.value_bits, a.power);
        const uint64_t b = (key - lokey) >> bshift;
        return b < kBins ? (uint32_t)b : kBins - 1;
      };
    #pragma unroll
      for (uint32_t e = 0; e < kBinsPer / 2; ++e) bins2[tid + e * kThreads] = 0;
      uint64_t mine[kPer];
      if (MODE == kEmitDirect) {
        // The sequence's keys lie where its chunks wrote them (raw).  Chunk by chunk they are copied into LDS, in
        // order, and every thread takes its keys tid, tid + 256, ... from there: all loads of a chunk are in
        // flight together and no thread has to look up which chunk holds its key.
        if (tid < kDirectChunks) ccount[tid] = tid < nch ? a.chunk_count[c0 + tid] : 0u;
        __syncthreads();
        // (every chunk's first 2 x 256 keys are fetched before any of them is stored: sixteen loads in flight per
        // thread instead of eight round trips one after the other - a quarter of the block's time)
        uint64_t got[kDirectChunks][2];
    #pragma unroll
        for (uint32_t x = 0; x < kDirectChunks; ++x) {
          const uint32_t n = ccount[x];  // (0 beyond the sequence's chunks)
          const uint64_t* src = a.keys + (uint64_t)begin + (uint64_t)x * a.chunk;
          got[x][0] = tid < n ? src[tid] : 0;
          got[x][1] = tid + kThreads < n ? src[tid + kThreads] : 0;
        }
        uint32_t at = 0;
    #pragma unroll
        for (uint32_t x = 0; x < kDirectChunks; ++x) {
          const uint32_t n = ccount[x];
          if (tid < n && at + tid < kSmallMax) grouped[at + tid] = compose_key(got[x][0], a.key_shift);
          if (tid + kThreads < n && at + tid + kThreads < kSmallMax)
            grouped[at + tid + kThreads] = compose_key(got[x][1], a.key_shift);
          if (n > 2 * kThreads) {
            const uint64_t* src = a.keys + (uint64_t)begin + (uint64_t)x * a.chunk;
            for (uint32_t t = tid + 2 * kThreads; t < n && at + t < kSmallMax; t += kThreads)
              grouped[at + t] = compose_key(src[t], a.key_shift);
          }
          at += n;
        }
        __syncthreads();
    #pragma unroll
        for (uint32_t e = 0; e < kPer; ++e) {
          const uint32_t i = tid + e * kThreads;
          mine[e] = i < count ? grouped[i] : 0;
        }
      } else {
    #pragma unroll
        for (uint32_t e = 0; e < kPer; ++e) {  // (all loads first)
          const uint32_t i = tid + e * kThreads;
          mine[e] = i < count ? a.keys[begin + i] : 0;
        }
      }
      __syncthreads();  // the bin counters are cleared (and, direct, every thread has its keys out of LDS)
      uint32_t bin[kPer];
    #pragma unroll
      for (uint32_t e = 0; e < kPer; ++e) {
        const uint32_t i = tid + e * kThreads;
        if (i < count) {
          bin[e] = bin_of(mine[e]);
          bin_inc(bin[e]);
        }
      }
      __syncthreads();
      // exclusive scan of the bin counts: thread t owns bins t * kBinsPer .. (kBinsPer / 2 dwords)
      uint32_t local[kBinsPer], sum = 0;
    #pragma unroll
      for (uint32_t e = 0; e < kBinsPer / 2; ++e) {
        const uint32_t w = bins2[tid * (kBinsPer / 2) + e];
        local[2 * e] = w & 0xFFFFu;
        local[2 * e + 1] = w >> 16;
        sum += local[2 * e] + local[2 * e + 1];
      }
      uint32_t excl;
      block_sum(sum, lds, &excl);
    #pragma unroll
      for (uint32_t e = 0; e < kBinsPer / 2; ++e) {
        const uint32_t lo16 = excl, hi16 = excl + local[2 * e];
        bins2[tid * (kBinsPer / 2) + e] = lo16 | (hi16 << 16);
        excl = hi16 + local[2 * e + 1];
      }
      __syncthreads();
      uint32_t place[kPer];  // where the key went in the grouped array
    #pragma unroll
      for (uint32_t e = 0; e < kPer; ++e) {
        const uint32_t i = tid + e * kThreads;
        if (i < count) {
          place[e] = bin_inc(bin[e]);
          grouped[place[e]] = mine[e];
        }
      }
      __syncthreads();
      // bins[b] is now the end of bin b: rank every key among the keys of its bin; a key that has an equal key
      // before it in the grouped array is a duplicate and is dropped
      uint32_t rank[kPer];
    #pragma unroll
      for (uint32_t e = 0; e < kPer; ++e) {
        const uint32_t i = tid + e * kThreads;
        rank[e] = ~0u;
        if (i < count) {
          const uint64_t key = mine[e];
          const uint32_t b = bin[e], p = place[e];
          const uint32_t lo = b ? bin_get(b - 1) : 0, hi = bin_get(b);
          uint32_t r = lo;
          bool dup = false;
          for (uint32_t m = lo; m < hi; ++m) {
            const uint64_t other = grouped[m];
            const bool same_before = other == key && m < p;
            r += (other < key || same_before) ? 1u : 0u;
            dup |= same_before;
          }
          if (dup)
            r |= 0x80000000u;
          else
            atomicOr(&distinct[r >> 5], 1u << (r & 31));
          rank[e] = r;
        }
      }
      __syncthreads();  // grouped / bins are free from here on
      // Place of every distinct key among the distinct keys = distinct keys of smaller rank: every wave scans the
      // 64 words of the bit map for itself (lane l holds word l and the number of bits in the words before it).
      static_assert(kSmallMax / 32 == 64, "one word of the bit map per lane");
      const uint32_t dword = distinct[tid & 63];
      uint32_t dincl = __builtin_popcount(dword);
    #pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(dincl, d);
        if ((int)(tid & 63) >= d) dincl += t;
      }
      const uint32_t dexcl = dincl - __builtin_popcount(dword);
      const uint32_t uniq = __shfl(dincl, 63);
      uint32_t upos[kPer];
    #pragma unroll
      for (uint32_t e = 0; e < kPer; ++e) {
        const uint32_t r = rank[e] & 0x7FFu;
        const uint32_t before = __shfl(dexcl, r >> 5), word = __shfl(dword, r >> 5);
        upos[e] = before + __builtin_popcount(word & ((1u << (r & 31)) - 1u));
      }
      // Reads: the count of distinct keys is published at once, the look-back over the predecessors' counts comes
      // after the staging below - blocks that started together publish at about the same time, and a block that
      // looks back right away mostly waits for the slowest of its 64 predecessors.
      if (MODE != kEmitBuckets && tid == 0)
        __hip_atomic_store(&a.scan_state[s], (s == 0 ? kScanPrefix : kScanCount) | uniq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      // Buckets of one sequence: all blocks run at once and finish together, nobody has a prefix yet: 256
      // predecessors at a time (wave v takes the 64 predecessors from top - 64 v downwards); whole waves are used
      // up in order until one meets a predecessor that already carries its prefix, or one whose predecessors
      // have not all published yet
      if (MODE == kEmitBuckets) {
        const uint32_t lane = tid & 63, wv = tid >> 6;
        if (tid == 0)
          __hip_atomic_store(&a.scan_state[s], (s == 0 ? kScanPrefix : kScanCount) | uniq, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        uint64_t before = 0;
        int64_t top = (int64_t)s - 1;  // newest predecessor not yet added
        uint32_t polls = 0;
        bool failed = false;
        while (top >= 0) {
          const int64_t idx = top - (int64_t)(wv * 64 + lane);
          uint64_t st = kScanPrefix;  // beyond the first sequence: nothing before it
          if (idx >= 0) st = __hip_atomic_load(&a.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t ready = __ballot((st >> 62) != 0);
          const uint64_t final = __ballot((st >> 62) == 2);
          // lanes 0 .. stop are usable when all of them are ready and lane `stop` carries a prefix (or the wave's
          // window has no prefix and all 64 are ready)
          const uint32_t stop = final ? (uint32_t)__builtin_ctzll(final) : 63u;
          const uint64_t need = stop == 63 ? ~0ull : ((2ull << stop) - 1);
          uint64_t v = (lane <= stop && idx >= 0) ? (st & kScanValue) : 0;
    #pragma unroll
          for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
          if (lane == 0) {
            lb_sum[wv] = v;
            lb_flag[wv] = ((ready & need) == need ? 1u : 0u) | (final ? 2u : 0u);
          }
          __syncthreads();
          uint32_t used = 0;
          bool done = false;
          for (uint32_t x = 0; x < kThreads / 64 && !done; ++x) {
            if (!(lb_flag[x] & 1u)) break;
            before += lb_sum[x];
            ++used;
            done = (lb_flag[x] & 2u) != 0;
          }
          __syncthreads();
          if (done) break;
          top -= 64 * (int64_t)used;
          if (used < kThreads / 64) {  // a wave's predecessors have not all published their counts yet
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1u << 24)) {  // the predecessors run already and wait for nobody but theirs: a safety net
              failed = true;
              break;
            }
          }
        }
        if (tid == 0) {
          if (failed && a.status) atomicMax(a.status, SA_ERR_HIP);
          if (s != 0)
            __hip_atomic_store(&a.scan_state[s], kScanPrefix | (before + uniq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_excl = before;
          if (s == 0) a.out_off[s] = before;
          if (s + 1 == nsegs) {
            a.out_off[1] = before + uniq;
            if (before + uniq > a.out_capacity && a.status) atomicMax(a.status, SA_ERR_CAPACITY);
          }
        }
      }
      const uint64_t low = (1ull << a.key_shift) - 1;
    #pragma unroll
      for (uint32_t e = 0; e < kPer; ++e) {
        if (rank[e] & 0x80000000u) continue;  // beyond the keys, or a duplicate
        const uint32_t u = upos[e];
        stage[2 * u] = (uint32_t)(mine[e] >> a.key_shift);
        stage[2 * u + 1] = (uint32_t)(mine[e] & low);  // position << 1 | strand (sequences here are far below 2^31)
      }
      __syncthreads();
      // this sequence's place among all sequences: look back over the predecessors' counts.  Reads: one wave,
      // 64 predecessors at a time (most find a prefix within a window or two; all four waves on 256
      // predecessors, as for the buckets, took 13 % longer)
      if (MODE != kEmitBuckets && tid < 64) {
        const uint32_t lane = tid;
        uint64_t before = 0;
        int64_t top = (int64_t)s - 1;  // newest predecessor not yet added
        uint32_t polls = 0;
        constexpr int kWin = 1;  // windows of 64 predecessors per poll (four, all loads in flight together, took
                                 // 10 % longer: what costs is the polling itself)
        while (top >= 0) {
          uint64_t st[kWin];
#pragma unroll
          for (int q = 0; q < kWin; ++q) {
            const int64_t idx = top - (64 * q + (int64_t)lane);
            st[q] = kScanPrefix;  // beyond the first sequence: nothing before it
            if (idx >= 0) st[q] = __hip_atomic_load(&a.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool done = false, stalled = false;
          uint32_t used = 0;
#pragma unroll
          for (int q = 0; q < kWin; ++q) {
            if (done || stalled) continue;
            const int64_t idx = top - (64 * q + (int64_t)lane);
            const uint64_t ready = __ballot((st[q] >> 62) != 0);
            const uint64_t final = __ballot((st[q] >> 62) == 2);
            // lanes 0 .. stop are usable when all of them are ready and lane `stop` carries a prefix (or the
            // window has no prefix and all 64 are ready)
            const uint32_t stop = final ? (uint32_t)__builtin_ctzll(final) : 63u;
            const uint64_t need = stop == 63 ? ~0ull : ((2ull << stop) - 1);
            if ((ready & need) != need) {
              stalled = true;
              continue;
            }
            uint64_t v = (lane <= stop && idx >= 0) ? (st[q] & kScanValue) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            before += v;
            ++used;
            done = final != 0;
          }
          if (done) break;
          top -= 64 * (int64_t)used;
          if (stalled) {
            __builtin_amdgcn_s_sleep(16);
            if (++polls > (1u << 24)) {  // the predecessors run already and wait for nobody but theirs: a safety net
              if (lane == 0 && a.status) atomicMax(a.status, SA_ERR_HIP);
              break;
            }
          }
        }
        if (lane == 0) {
          if (s != 0)
            __hip_atomic_store(&a.scan_state[s], kScanPrefix | (before + uniq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_excl = before;
          a.out_off[s] = before;
          if (s + 1 == a.n) {
            a.out_off[a.n] = before + uniq;
            if (before + uniq > a.out_capacity && a.status) atomicMax(a.status, SA_ERR_CAPACITY);
          }
        }
      }
      if (MODE != kEmitBuckets) __syncthreads();  // s_excl
      // the (value, position, strand) triples leave in order, twelve bytes per thread and store
      const uint64_t first_out = s_excl;
      const uint64_t room = first_out < a.out_capacity ? a.out_capacity - first_out : 0;
      const uint32_t n_out = (uint32_t)(uniq < room ? uniq : room);
      uint32_t* dst = a.out + 3 * first_out;
      for (uint32_t u = tid; u < n_out; u += kThreads) {
        const uint32_t value = stage[2 * u], ps = stage[2 * u + 1];
        uint32_t* to = dst + 3 * (uint64_t)u;
        to[0] = value;
        to[1] = ps >> 1;
        to[2] = ps & 1u;
      }
    }
  }
}

// ---- one long sequence: its keys are cut into value-ordered buckets of about kBucketKeys keys, which
// sort_emit_small then sorts, de-duplicates and writes like the reads of a batch - five launches instead of a
// device-wide radix sort's twenty-odd, which is what a call on a single genome-sized sequence costs.
//   (min_run / min_fix count the keys per fine bin - the top kFineBits bits of a key's significant bits -
//   as they write them)
//   bucket_plan     exclusive prefix P of the counts (left in place as the scatter's cursors);
//                   bucket of a fine bin = P >> 10, so buckets are runs of whole fine bins and hold at most
//                   kBucketKeys keys plus one fine bin; writes every bucket's key range and fine-bin range
//   bucket_scatter  moves every key to its fine bin's range (order inside a bin is arbitrary)
//   bucket_split    only for a bucket beyond kSmallMax keys (one fine bin that holds thousands of keys: a value
//                   repeated at thousands of positions): sorts it in place in global memory (bitonic network,
//                   one block) and hands its slices to the bucket numbers its length left unused
// (2^16 fine bins: with 2^14 the counters' atomics in min_run and bucket_scatter collide four times as often and
// cost more than bucket_plan saves)
constexpr uint32_t kFineBits = 16, kFine = 1u << kFineBits;
constexpr uint32_t kPlanBlocks = 64;  // bucket_plan: 1024 fine bins a block
constexpr uint32_t kBucketShift = 10, kBucketKeys = 1u << kBucketShift;
constexpr uint32_t kNoBin = 0xFFFFFFFFu;

struct BucketArgs {
  const uint64_t* keys;     // the sequence's keys as min_run left them: chunk c's from slot c * chunk on
  uint64_t* grouped;        // the same keys, grouped by fine bin
  const uint32_t* nchunks;  // chunks of the sequence (device side)
  const uint32_t* chunk_count;
  uint32_t chunk;
  uint32_t fine_shift;
  uint32_t* hist;           // the fine bins' counters, zeroed; after bucket_plan: exclusive prefix / cursors
  uint32_t nseg;            // bucket numbers 0 .. nseg - 1 (more than the keys can fill)
  uint32_t key_shift;       // (the chunks hold raw keys)
  uint32_t* plan;           // bucket_plan's exchange between its blocks: 3 * kPlanBlocks + 1 words, zeroed
  int* status;
  uint32_t* nseg_used;      // bucket numbers the keys of this call reach: (count >> kBucketShift) + 1
  uint32_t* seg_begin;      // all zeroed before bucket_plan
  uint32_t* seg_end;
  uint32_t* seg_lof;
  uint32_t* seg_hif;
  uint64_t* seg_lokey;
  uint64_t* seg_span;
};

// kPlanBlocks blocks, all resident at once (far fewer than the GPU has CUs): every block sums its 1024 fine
// bins and publishes the sum with its last non-empty bin, waits until all blocks have done so (a bounded wait;
// the call fails rather than hangs), and then knows where its bins' keys start.
__global__ __launch_bounds__(kThreads) void bucket_plan(BucketArgs a) {
  constexpr uint32_t kPer = 4, kOwn = kPer * kThreads;
  static_assert(kOwn * kPlanBlocks == kFine && kThreads == 256 && kPlanBlocks == 64, "fine_slot's layout, one wave reads the blocks' sums");
  __shared__ uint32_t cnt[kOwn];  // the block's counters in bin order, later their exclusive prefixes
  __shared__ uint32_t lds[kThreads / 64];
  __shared__ uint32_t last_start[kThreads], last_bin[kThreads];
  __shared__ uint32_t last_owner;  // 1 + the last thread that owns a non-empty fine bin
  __shared__ uint32_t wave_owner[kThreads / 64];  // the same per wave
  __shared__ uint32_t s_base, s_total, s_prev, s_prev_start, s_is_last;
  const uint32_t tid = threadIdx.x, blk = blockIdx.x, lane = tid & 63;
  uint32_t* plan_total = a.plan;
  uint32_t* plan_lastbin = a.plan + kPlanBlocks;    // 1 + the block's last non-empty bin, 0: none
  uint32_t* plan_laststart = a.plan + 2 * kPlanBlocks;  // that bin's first key, counted from the block's first
  uint32_t* plan_arrived = a.plan + 3 * kPlanBlocks;
  if (tid == 0) last_owner = 0;
  {  // bin blk * 1024 + i * 256 + e lives in word e * 256 + 4 * blk + i
    const uint4 v = *reinterpret_cast<const uint4*>(a.hist + tid * 256 + 4 * blk);
    cnt[tid] = v.x;
    cnt[256 + tid] = v.y;
    cnt[512 + tid] = v.z;
    cnt[768 + tid] = v.w;
  }
  __syncthreads();
  // thread t owns the block's bins 4t .. 4t + 3
  const uint32_t first = blk * kOwn + tid * kPer;
  uint32_t cs[kPer], sum = 0, lb = kNoBin, ls = 0;
#pragma unroll
  for (uint32_t x = 0; x < kPer; ++x) {
    cs[x] = cnt[tid * kPer + x];
    if (cs[x]) {
      lb = first + x;
      ls = sum;  // (relative to the thread's first key)
    }
    sum += cs[x];
  }
  uint32_t excl;
  const uint32_t block_total = block_sum(sum, lds, &excl);
  last_start[tid] = excl + ls;  // start of the thread's last non-empty fine bin, for the threads after it
  last_bin[tid] = lb;
  const uint64_t owners = __ballot(lb != kNoBin);  // (most bins of the upper half of the value range are empty)
  if (lane == 0) wave_owner[tid >> 6] = owners ? (tid | 63u) + 1 - (uint32_t)__builtin_clzll(owners) : 0;
  if (lb != kNoBin) atomicMax(&last_owner, tid + 1);
  __syncthreads();
  if (tid == 0) {
    const uint32_t owner = last_owner;
    plan_total[blk] = block_total;
    plan_lastbin[blk] = owner ? last_bin[owner - 1] + 1 : 0;
    plan_laststart[blk] = owner ? last_start[owner - 1] : 0;
    __threadfence();
    atomicAdd(plan_arrived, 1u);
  }
  if (tid < 64) {  // the first wave waits for the other blocks and reads what they published
    uint32_t spins = 0;
    while (__hip_atomic_load(plan_arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < kPlanBlocks) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) {  // (seconds) never expected: the other blocks did not start
        if (tid == 0 && a.status) atomicMax(a.status, SA_ERR_HIP);
        break;
      }
    }
    const uint32_t tot = __hip_atomic_load(plan_total + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lbin = __hip_atomic_load(plan_lastbin + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lstart = __hip_atomic_load(plan_laststart + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d);
      if ((int)lane >= d) inc += t;
    }
    const uint32_t base_of_lane = inc - tot;
    const uint64_t nonempty = __ballot(lbin != 0);
    const uint64_t before = blk ? nonempty & ((1ull << blk) - 1ull) : 0ull;
    const uint64_t after = blk < 63 ? nonempty >> (blk + 1) : 0ull;
    const uint32_t src = before ? 63u - (uint32_t)__builtin_clzll(before) : 0u;
    const uint32_t p_bin = __shfl(lbin, src), p_start = __shfl(base_of_lane + lstart, src);
    const uint32_t my_base = __shfl(base_of_lane, blk), all = __shfl(inc, 63);
    if (tid == 0) {
      s_base = my_base;
      s_total = all;
      s_prev = before ? p_bin - 1 : kNoBin;
      s_prev_start = before ? p_start : 0;
      s_is_last = after == 0;
    }
  }
  __syncthreads();
  const uint32_t base = s_base, total = s_total;
  // the non-empty fine bin before this thread's first one: of the nearest owner among the lanes below, else of
  // the last owner of a wave before, else of a block before
  uint32_t prev = s_prev, prev_start = s_prev_start;
  {
    const uint64_t below = owners & ((1ull << lane) - 1ull);
    uint32_t owner = below ? (tid & ~63u) + 64 - (uint32_t)__builtin_clzll(below) : 0;  // 1 + its thread
    for (int wv = (int)(tid >> 6) - 1; wv >= 0 && owner == 0; --wv) owner = wave_owner[wv];
    if (owner) {
      prev = last_bin[owner - 1];
      prev_start = base + last_start[owner - 1];
    }
  }
  uint32_t P = base + excl;
#pragma unroll
  for (uint32_t x = 0; x < kPer; ++x) {
    const uint32_t f = first + x;
    const uint32_t c = cs[x];
    cnt[tid * kPer + x] = P;
    if (c) {
      const uint32_t id = P >> kBucketShift;
      if (prev == kNoBin || (prev_start >> kBucketShift) != id) {  // first fine bin of bucket id
        a.seg_begin[id] = P;
        a.seg_lof[id] = f;
        if (prev != kNoBin) {
          a.seg_end[prev_start >> kBucketShift] = P;
          a.seg_hif[prev_start >> kBucketShift] = prev + 1;
        }
      }
      prev = f;
      prev_start = P;
    }
    P += c;
  }
  // the last bucket ends with the keys
  if (s_is_last && last_owner == tid + 1) {
    a.seg_end[prev_start >> kBucketShift] = total;
    a.seg_hif[prev_start >> kBucketShift] = prev + 1;
  }
  if (blk == 0 && tid == 0) *a.nseg_used = (total >> kBucketShift) + 1;
  __syncthreads();
  *reinterpret_cast<uint4*>(a.hist + tid * 256 + 4 * blk) = make_uint4(cnt[tid], cnt[256 + tid], cnt[512 + tid], cnt[768 + tid]);
}

// (chunk c's keys start at key slot c * chunk; `share` lanes per chunk: 16 for chunks of up to 64 windows, which
// hold about a third as many keys, up to the whole wave)
__global__ __launch_bounds__(kThreads) void bucket_scatter(BucketArgs a, uint32_t share) {
  const uint32_t nchunks = *a.nchunks, lane = threadIdx.x & 63;
  const uint32_t per_wave = 64 / share, sub = lane / share, li = lane % share;
  const uint32_t waves = gridDim.x * (kThreads / 64);
  for (uint32_t c0 = (blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * per_wave; c0 < nchunks; c0 += waves * per_wave) {
    const uint32_t c = c0 + sub;
    if (c >= nchunks) continue;
    const uint32_t n = a.chunk_count[c];
    const uint64_t* src = a.keys + (uint64_t)c * a.chunk;
    for (uint32_t i = li; i < n; i += share) {
      const uint64_t key = compose_key(src[i], a.key_shift);
      a.grouped[atomicAdd(&a.hist[fine_slot((uint32_t)(key >> a.fine_shift))], 1u)] = key;
    }
  }
}

__global__ __launch_bounds__(kThreads) void bucket_split(BucketArgs a) {
  const uint32_t id0 = blockIdx.x, tid = threadIdx.x;
  if (id0 >= *a.nseg_used) return;
  const uint32_t begin = a.seg_begin[id0], end = a.seg_end[id0], n = end - begin;
  if (n <= kSmallMax) return;
  uint64_t* v = a.grouped + begin;
  // bitonic sorting network in which every comparison puts the smaller key first; positions beyond n count as
  // keys larger than any, so comparisons with them never swap
  uint32_t np = 1;
  while (np < n) np <<= 1;
  for (uint32_t k = 2; k <= np; k <<= 1) {
    for (uint32_t x = tid; x < np / 2; x += kThreads) {  // first step of a merge: mirror pairs
      const uint32_t blk = x / (k / 2), off = x % (k / 2);
      const uint32_t i = blk * k + off, j = blk * k + k - 1 - off;
      if (j < n) {
        const uint64_t vi = v[i], vj = v[j];
        if (vi > vj) {
          v[i] = vj;
          v[j] = vi;
        }
      }
    }
    __syncthreads();
    for (uint32_t d = k / 4; d > 0; d >>= 1) {
      for (uint32_t x = tid; x < np / 2; x += kThreads) {
        const uint32_t i = 2 * d * (x / d) + x % d, j = i + d;
        if (j < n) {
          const uint64_t vi = v[i], vj = v[j];
          if (vi > vj) {
            v[i] = vj;
            v[j] = vi;
          }
        }
      }
      __syncthreads();
    }
  }
  // The bucket's keys start in block id0 of kBucketKeys key positions and the next bucket's in block
  // end >> kBucketShift: the numbers in between belong to nobody.  Equal slices for them, every cut moved
  // forward past keys equal to the one before it (equal keys must meet in one slice to be de-duplicated).
  if (tid == 0) {
    const uint32_t avail = (end >> kBucketShift) - id0;  // >= 2
    uint32_t cut = 0;
    for (uint32_t m = 0; m < avail; ++m) {
      uint32_t next = m + 1 == avail ? n : (uint32_t)(((uint64_t)n * (m + 1)) / avail);
      if (next < cut) next = cut;
      while (next < n && next > 0 && v[next] == v[next - 1]) ++next;
      a.seg_begin[id0 + m] = begin + cut;
      a.seg_end[id0 + m] = begin + next;
      a.seg_hif[id0 + m] = 0;  // bounds given as keys
      a.seg_lokey[id0 + m] = next > cut ? v[cut] : 0;
      a.seg_span[id0 + m] = next > cut ? v[next - 1] - v[cut] + 1 : 0;
      cut = next;
    }
  }
}

__global__ __launch_bounds__(kThreads) void uniq_count(UniqArgs u) {
  __shared__ uint32_t lds[kThreads / 64];
  if (u.skip_small && *u.n_big == 0) return;  // every tile was counted by sort_small
  const uint32_t tiles = u.tile_base[u.n];
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    uint32_t begin, end, seg_first;
    const uint32_t s = locate_tile(u, tile, &begin, &end, &seg_first);
    if (u.skip_small && u.seg_end[s] - seg_first <= kTile) continue;
    uint64_t keys[kTileItems];
    const uint32_t flags = tile_flags(u, begin, end, seg_first, keys);
    uint32_t excl;
    const uint32_t total = block_sum(__builtin_popcount(flags), lds, &excl);
    if (threadIdx.x == 0) u.tile_count[tile] = total;
  }
}

__global__ __launch_bounds__(kThreads) void uniq_write(UniqArgs u) {
  __shared__ uint32_t lds[kThreads / 64];
  __shared__ uint32_t stage[3 * kTile];
  if (u.skip_small && *u.n_big == 0) return;  // sort_emit_small has written the output
  const uint32_t tiles = u.tile_base[u.n];
  const uint64_t low = (1ull << u.key_shift) - 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t total = u.tile_off[tiles];
    u.out_off[u.n] = total;
    if (total > u.out_capacity && u.status) atomicMax(u.status, SA_ERR_CAPACITY);
  }
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    uint32_t begin, end, seg_first;
    const uint32_t s = locate_tile(u, tile, &begin, &end, &seg_first);
    uint64_t keys[kTileItems];
    const uint32_t flags = tile_flags(u, begin, end, seg_first, keys);
    uint32_t excl;
    const uint32_t total = block_sum(__builtin_popcount(flags), lds, &excl);
    const uint64_t tile_first = u.tile_off[tile];
    if (threadIdx.x == 0 && begin == seg_first) u.out_off[s] = tile_first;
    // the tile's triples are gathered in LDS and leave as one contiguous run of dwords
    uint32_t r = excl;
#pragma unroll
    for (uint32_t e = 0; e < kTileItems; ++e) {
      if (!((flags >> e) & 1u)) continue;
      stage[3 * r] = (uint32_t)(keys[e] >> u.key_shift);
      stage[3 * r + 1] = (uint32_t)((keys[e] & low) >> 1);
      stage[3 * r + 2] = (uint32_t)(keys[e] & 1);
      ++r;
    }
    __syncthreads();
    const uint64_t room = tile_first < u.out_capacity ? u.out_capacity - tile_first : 0;
    const uint32_t dwords = 3 * (uint32_t)(total < room ? total : room);
    uint32_t* dst = u.out + 3 * tile_first;
    for (uint32_t x = threadIdx.x; x < dwords; x += kThreads) dst[x] = stage[x];
    __syncthreads();
  }
}

struct MinPlan {
  uint32_t chunk;       // windows per chunk
  uint32_t max_chunks;
  uint32_t max_tiles;
  uint64_t capacity;    // candidate keys (upper bound: one per window)
  uint32_t key_shift, end_bit;
  size_t scan32_tmp, scan64_tmp, sort_tmp;
  size_t total_bytes;
  // workspace offsets
  size_t o_nchunks, o_chunk_base, o_windows, o_wbase, o_flags, o_chunk_count, o_cscan, o_exit_f, o_exit_c, o_resolved,
      o_keys, o_sorted, o_seg_begin, o_seg_end, o_big_end, o_tiles, o_tile_base, o_tile_count, o_tile_off, o_scan_state,
      o_tmp;
  size_t scan_chunks_tmp;
  bool direct;          // reads: at most kDirectChunks chunks per sequence, sort_emit_small collects them itself
  // one sequence of moderate length: bucketed sort (bucket_plan)
  bool buckets;
  uint32_t nseg, fine_shift;
  size_t o_bucket_zero, bucket_zero_bytes;  // one zeroed block: histogram, bucket tables, scan state + ticket
  size_t o_hist, o_bseg_begin, o_bseg_end, o_bseg_lof, o_bseg_hif, o_bseg_lokey, o_bseg_span, o_bscan, o_bplan;
};

// longest single sequence (in windows) that takes the bucketed sort: beyond it a device-wide radix sort is
// bound by bandwidth, not by its number of launches, and the fine bins would hold too many keys each
constexpr uint64_t kBucketMaxWindows = 32u << 20;

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
  if (k < 1 || k > 31 || w < 1) return SA_ERR_ARGUMENT;
  if (variant == SA_MIN_BINARY && w > k + 1) return SA_ERR_ARGUMENT;
  if (variant == SA_MIN_DECIMAL && k > 10) return SA_ERR_RANGE;
  const uint64_t span = (uint64_t)k + w - 2;
  pl->capacity = total_len > span * n ? total_len - span * n : 1;  // total windows
  if (pl->capacity >= 0xFFFFFFF0ull) return SA_ERR_ARGUMENT;
  // enough chunks to occupy the GPU: about 2048 windows each for large batches (a chunk pays 160 warm-up
  // steps), 256 down to 64 for small ones (one 4.6 Mbp sequence, the whole call: 200 / 185 / 197 / 204 / 202 us
  // with 64 / 80 / 96 / 112 / 128)
  pl->chunk = (total_len / 2048 >= 65536)  ? 2048
              : (total_len / 1024 >= 65536) ? 1024
              : (total_len / 256 >= 65536)  ? 256
              : (total_len / 128 >= 65536)  ? 128
                                            : 64;
  if (pl->chunk == 64 && total_len / 64 > 65536)  // between the two: one block of chunks per CU
    pl->chunk = (uint32_t)((total_len / 65536 + 16) / 16 * 16);
  const uint64_t max_windows = max_len > span ? max_len - span : 1;
  // sequences of about one length (reads): equal chunks - the lanes of a wave run in step, a short last chunk
  // idles its lane
  if (n > 1 && pl->chunk >= 1024 && max_windows > pl->chunk && pl->capacity >= (uint64_t)n * max_windows / 16 * 15) {
    const uint64_t per_seq = (max_windows + pl->chunk - 1) / pl->chunk;
    pl->chunk = (uint32_t)(((max_windows + per_seq - 1) / per_seq + 15) / 16 * 16);
  }
  {  // test hook: any chunk length from 8 to 4096 windows
    const long v = options().min_chunk;
    if (v >= 8 && v <= 4096) pl->chunk = (uint32_t)v;
  }
  if (variant == SA_MIN_DECIMAL)  // one thread per sequence: one chunk per sequence
    pl->chunk = (uint32_t)std::min<uint64_t>((max_windows + 15) / 16 * 16, 0xFFFFFFF0u);
  pl->max_chunks = (uint32_t)(total_len / pl->chunk + n);
  pl->direct = n > 1 && (max_windows + pl->chunk - 1) / pl->chunk <= kDirectChunks;
  pl->max_tiles = (uint32_t)(pl->capacity / kTile + n);
  // w = 1 can emit k-mer 0, whose value the reference builds without the mask: with a non-base byte in it
  // the value has ones above its 2k bits
  // (k > 16: the reference's 32-bit values keep the last 16 bases of a k-mer)
  const uint32_t value_bits = (variant == SA_MIN_BINARY) ? ((w == 1 || k > 16) ? 32 : 2 * k) : 31;
  pl->key_shift = bits_for(max_len) + 1;
  pl->end_bit = pl->key_shift + value_bits;
  if (pl->end_bit > 64) return SA_ERR_ARGUMENT;
  const size_t mc = pl->max_chunks + 1, mt = (size_t)pl->max_tiles + 1;
  // rocprim temporary storage sizes
  uint32_t* d32 = nullptr;
  uint64_t* d64 = nullptr;
  pl->scan32_tmp = pl->scan64_tmp = pl->sort_tmp = 0;
  if (rocprim::exclusive_scan(nullptr, pl->scan32_tmp, d32, d32, 0u, (size_t)n + 1, rocprim::plus<uint32_t>()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::exclusive_scan(nullptr, pl->scan64_tmp, d32, d64, (uint64_t)0, mt, rocprim::plus<uint64_t>()) != hipSuccess) return SA_ERR_HIP;
  pl->scan_chunks_tmp = 0;
  if (rocprim::exclusive_scan(nullptr, pl->scan_chunks_tmp, d32, d32, 0u, mc, rocprim::plus<uint32_t>()) != hipSuccess) return SA_ERR_HIP;
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
  const size_t n1 = ((size_t)n + 1) * 4;
  pl->o_nchunks = take(n1);
  pl->o_chunk_base = take(n1);
  pl->o_windows = take(n1);
  pl->o_wbase = take(n1);
  pl->o_flags = take(16);        // unresolved chunks, long sequences, "bytes other than bases seen"
  pl->o_chunk_count = take(mc * 4);
  pl->o_cscan = take(mc * 4);
  pl->o_exit_f = take(mc * 4);
  pl->o_exit_c = take(mc * 4);
  pl->o_resolved = take(mc);
  pl->o_keys = take((pl->capacity + 1) * 8);
  pl->o_sorted = take((pl->capacity + 1) * 8);
  pl->o_seg_begin = take(n1);
  pl->o_seg_end = take(n1);
  pl->o_big_end = take(n1);
  pl->o_tiles = take(n1);
  pl->o_tile_base = take(n1);
  pl->o_tile_count = take(mt * 4);
  pl->o_tile_off = take(mt * 8);
  pl->o_scan_state = take(((size_t)n + 1) * 8);  // look-back scan of the fused sort/output kernel + its ticket
  pl->buckets = n == 1 && variant == SA_MIN_BINARY && pl->capacity <= kBucketMaxWindows && !options().no_buckets;
  pl->nseg = (uint32_t)(pl->capacity / kBucketKeys + 2);
  pl->fine_shift = pl->end_bit > kFineBits ? pl->end_bit - kFineBits : 0;
  if (pl->buckets) {
    const size_t ns = pl->nseg;
    pl->o_bucket_zero = o;
    pl->o_hist = take((size_t)kFine * 4);
    pl->o_bseg_begin = take(ns * 4);
    pl->o_bseg_end = take(ns * 4);
    pl->o_bseg_lof = take(ns * 4);
    pl->o_bseg_hif = take(ns * 4);
    pl->o_bseg_lokey = take(ns * 8);
    pl->o_bseg_span = take(ns * 8);
    pl->o_bscan = take((ns + 2) * 8);  // scan state, buckets in use
    pl->o_bplan = take((3 * kPlanBlocks + 1) * 4);
    pl->bucket_zero_bytes = o - pl->o_bucket_zero;
  }
  pl->o_tmp = take(std::max(std::max(std::max(pl->scan32_tmp, pl->scan64_tmp), pl->scan_chunks_tmp), pl->sort_tmp));
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
  if (b->k < 1 || b->k > 31 || b->w < 1 || (b->variant == SA_MIN_BINARY && b->w > b->k + 1))
    return fail(SA_ERR_ARGUMENT, "need 1 <= k <= 31, w >= 1 and (MinimizeBinary) w <= k + 1 (k=%u, w=%u)",
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
  uint32_t* windows = (uint32_t*)(ws + pl.o_windows);
  uint32_t* wbase = (uint32_t*)(ws + pl.o_wbase);
  uint32_t* flags = (uint32_t*)(ws + pl.o_flags);
  uint32_t* n_unresolved = flags;
  uint32_t* n_big = flags + 1;
  uint32_t* other = flags + 2;
  uint32_t* chunk_count = (uint32_t*)(ws + pl.o_chunk_count);
  uint32_t* cscan = (uint32_t*)(ws + pl.o_cscan);
  uint64_t* keys = (uint64_t*)(ws + pl.o_keys);
  uint64_t* sorted = (uint64_t*)(ws + pl.o_sorted);
  uint32_t* seg_begin = (uint32_t*)(ws + pl.o_seg_begin);
  uint32_t* seg_end = (uint32_t*)(ws + pl.o_seg_end);
  uint32_t* big_end = (uint32_t*)(ws + pl.o_big_end);
  uint32_t* tiles = (uint32_t*)(ws + pl.o_tiles);
  uint32_t* tile_base = (uint32_t*)(ws + pl.o_tile_base);
  uint32_t* tile_count = (uint32_t*)(ws + pl.o_tile_count);
  uint64_t* tile_off = (uint64_t*)(ws + pl.o_tile_off);
  void* tmp = ws + pl.o_tmp;

  MinArgs a{};
  a.seq = b->d_seq;
  a.off = b->d_off;
  a.n = b->n;
  a.k = b->k;
  a.w = b->w;
  a.key_shift = pl.key_shift;
  a.chunk = pl.chunk;
  a.chunk_base = chunk_base;
  a.wbase = wbase;
  a.chunk_count = chunk_count;
  a.exit_f = (uint32_t*)(ws + pl.o_exit_f);
  a.exit_c = (uint32_t*)(ws + pl.o_exit_c);
  a.resolved = (uint8_t*)(ws + pl.o_resolved);
  a.keys = keys;
  a.status = b->d_status;
  a.other = other;
  a.hist = pl.buckets ? (uint32_t*)(ws + pl.o_hist) : nullptr;
  a.fine_shift = pl.fine_shift;
  auto blocks = [](uint64_t items) { return dim3((unsigned)((items + kThreads - 1) / kThreads)); };
  const uint32_t bin_power = std::min<uint32_t>(2 * b->w, 64);  // value_bin: a window's value is the least of ~2w
  size_t t = 0;
  if (pl.buckets) {  // one sequence: the bookkeeping in one launch, the bucket tables cleared alongside
    SA_HIP_TRY(hipMemsetAsync(ws + pl.o_bucket_zero, 0, pl.bucket_zero_bytes, stream));
    hipLaunchKernelGGL(min_plan_single, dim3(1), dim3(64), 0, stream, b->d_off, b->k, b->w, pl.chunk, nchunks,
                       chunk_base, windows, wbase, flags, b->d_status);
  } else {
    if (b->d_status) SA_HIP_TRY(hipMemsetAsync(b->d_status, 0, sizeof(int), stream));
    SA_HIP_TRY(hipMemsetAsync(flags, 0, 16, stream));
    SA_HIP_TRY(hipMemsetAsync(tile_count, 0, ((size_t)pl.max_tiles + 1) * 4, stream));
    if (!pl.direct)  // (the scan over the chunks' counts runs over the whole array)
      SA_HIP_TRY(hipMemsetAsync(chunk_count, 0, ((size_t)pl.max_chunks + 1) * 4, stream));
    hipLaunchKernelGGL(min_seq_plan, blocks(b->n + 1), dim3(kThreads), 0, stream, b->d_off, b->n, b->k,
                       b->w, pl.chunk, nchunks, windows);
    t = pl.scan32_tmp;
    SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, nchunks, chunk_base, 0u, (size_t)b->n + 1,
                                       rocprim::plus<uint32_t>(), stream));
    t = pl.scan32_tmp;
    SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, windows, wbase, 0u, (size_t)b->n + 1,
                                       rocprim::plus<uint32_t>(), stream));
  }
  if (b->variant == SA_MIN_BINARY) {
    const bool long_k = b->k > 16;  // a third history word
    auto* run = long_k                    ? min_run<Machine<0, true>, 0>
                : (b->w == 5 && b->k == 15) ? min_run<Machine<5>, 15>
                : b->w == 5                 ? min_run<Machine<5>, 0>
                : b->w == 10                ? min_run<Machine<10>, 0>
                                            : min_run<Machine<0>, 0>;
    hipLaunchKernelGGL(run, blocks(pl.max_chunks), dim3(kThreads), 0, stream, a, n_unresolved);
    auto* fix = long_k ? min_fix<Machine<0, true>> : min_fix<Machine<0>>;
    hipLaunchKernelGGL(fix, blocks((uint64_t)b->n * 64), dim3(kThreads), 0, stream, a, n_unresolved, kFine);
    // a batch with bytes other than bases runs again on the exact machine (the two kernels return at
    // once otherwise)
    auto* run_x = long_k ? min_run<MachineX<true>, 0> : min_run<MachineX<false>, 0>;
    auto* fix_x = long_k ? min_fix<MachineX<true>> : min_fix<MachineX<false>>;
    hipLaunchKernelGGL(run_x, blocks(pl.max_chunks), dim3(kThreads), 0, stream, a, n_unresolved);
    hipLaunchKernelGGL(fix_x, blocks((uint64_t)b->n * 64), dim3(kThreads), 0, stream, a, n_unresolved, kFine);
  } else {
    hipLaunchKernelGGL(dec_run, blocks(b->n), dim3(kThreads), 0, stream, a);
  }
  if (pl.buckets) {
    // one sequence: value-ordered buckets of about a thousand keys, sorted and written by sort_emit_small
    BucketArgs ba{};
    ba.keys = keys;
    ba.grouped = sorted;
    ba.nchunks = nchunks;
    ba.chunk_count = chunk_count;
    ba.chunk = pl.chunk;
    ba.key_shift = pl.key_shift;
    ba.fine_shift = pl.fine_shift;
    ba.hist = (uint32_t*)(ws + pl.o_hist);
    ba.nseg = pl.nseg;
    ba.seg_begin = (uint32_t*)(ws + pl.o_bseg_begin);
    ba.seg_end = (uint32_t*)(ws + pl.o_bseg_end);
    ba.seg_lof = (uint32_t*)(ws + pl.o_bseg_lof);
    ba.seg_hif = (uint32_t*)(ws + pl.o_bseg_hif);
    ba.seg_lokey = (uint64_t*)(ws + pl.o_bseg_lokey);
    ba.seg_span = (uint64_t*)(ws + pl.o_bseg_span);
    ba.nseg_used = (uint32_t*)((uint64_t*)(ws + pl.o_bscan) + pl.nseg + 1);
    ba.plan = (uint32_t*)(ws + pl.o_bplan);
    ba.status = b->d_status;
    // (the scatter is a chain of atomics that return a value)
    const uint32_t share = pl.chunk <= 64 ? 16 : pl.chunk <= 128 ? 32 : 64;
    const dim3 sweep((unsigned)std::min<uint64_t>(pl.max_chunks / (kThreads / share) + 1, 65535));
    hipLaunchKernelGGL(bucket_plan, dim3(kPlanBlocks), dim3(kThreads), 0, stream, ba);
    hipLaunchKernelGGL(bucket_scatter, sweep, dim3(kThreads), 0, stream, ba, share);
    hipLaunchKernelGGL(bucket_split, dim3(pl.nseg), dim3(kThreads), 0, stream, ba);
    EmitArgs ea{};
    ea.keys = sorted;
    ea.seg_begin = ba.seg_begin;
    ea.seg_end = ba.seg_end;
    ea.n = pl.nseg;
    ea.key_shift = pl.key_shift;
    ea.value_bits = pl.end_bit - pl.key_shift;
    ea.power = bin_power;
    ea.scan_state = (uint64_t*)(ws + pl.o_bscan);
    ea.n_big = n_big;
    ea.out_capacity = b->out_capacity;
    ea.out = b->d_out;
    ea.out_off = b->d_out_off;
    ea.status = b->d_status;
    ea.buckets = 1;
    ea.nseg_used = ba.nseg_used;
    ea.fine_shift = pl.fine_shift;
    ea.seg_lof = ba.seg_lof;
    ea.seg_hif = ba.seg_hif;
    ea.seg_lokey = ba.seg_lokey;
    ea.seg_span = ba.seg_span;
    hipLaunchKernelGGL(sort_emit_small<kEmitBuckets>, dim3(pl.nseg), dim3(kThreads), 0, stream, ea);
    SA_HIP_TRY(hipGetLastError());
    return SA_OK;
  }
  if (pl.direct) {
    // reads: no sequence with more than kSmallMax keys, as a rule - one kernel collects a read's chunks, sorts,
    // removes duplicates and writes the triples; otherwise it returns at once and the kernels after it do the work
    hipLaunchKernelGGL(min_count_direct, blocks(b->n), dim3(kThreads), 0, stream, a, seg_begin, n_big);
    EmitArgs ea{};
    ea.keys = keys;
    ea.seg_begin = seg_begin;  // (direct: the sequences' key counts)
    ea.seg_end = seg_end;
    ea.n = b->n;
    ea.key_shift = pl.key_shift;
    ea.value_bits = pl.end_bit - pl.key_shift;
    ea.power = bin_power;
    ea.scan_state = (uint64_t*)(ws + pl.o_scan_state);
    ea.n_big = n_big;
    ea.out_capacity = b->out_capacity;
    ea.out = b->d_out;
    ea.out_off = b->d_out_off;
    ea.status = b->d_status;
    ea.direct = 1;
    ea.chunk_base = chunk_base;
    ea.chunk_count = chunk_count;
    ea.wbase = wbase;
    ea.chunk = pl.chunk;
    SA_HIP_TRY(hipMemsetAsync(ea.scan_state, 0, ((size_t)b->n + 1) * 8, stream));
    hipLaunchKernelGGL(sort_emit_small<kEmitDirect>, dim3(b->n), dim3(kThreads), 0, stream, ea);
    SA_HIP_TRY(hipGetLastError());
    // sequences of at most kSmallMax windows cannot have more keys: nothing is left to do
    if ((uint64_t)b->max_len - (b->k + b->w - 2) <= kSmallMax) return SA_OK;
  }
  // The several-kernel path: the chunks' keys are packed into `sorted`, sequence by sequence, and sorted from
  // there back into `keys`.  (After sort_emit_small has served a batch of reads every kernel below returns at
  // once or finds only empty segments.)
  t = pl.scan_chunks_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, chunk_count, cscan, 0u, (size_t)pl.max_chunks + 1,
                                     rocprim::plus<uint32_t>(), stream));
  hipLaunchKernelGGL(compact_keys, dim3(std::min<uint32_t>(pl.max_chunks, 16384)), dim3(kThreads), 0, stream, a,
                     cscan, sorted, pl.direct ? n_big : nullptr);
  hipLaunchKernelGGL(min_segments, blocks(b->n + 1), dim3(kThreads), 0, stream, cscan, chunk_base, b->n,
                     seg_begin, seg_end, big_end, tiles, n_big, pl.direct ? 0u : 1u);
  SA_HIP_TRY(hipGetLastError());
  std::swap(keys, sorted);  // keys: packed, unsorted; sorted: the sort's output
  if (!pl.direct && b->n > 1) {
    // sequences of many chunks each, none with more than kSmallMax keys: sort_emit_small on the packed keys
    // (it returns at once otherwise)
    EmitArgs ea{};
    ea.keys = keys;
    ea.seg_begin = seg_begin;
    ea.seg_end = seg_end;
    ea.n = b->n;
    ea.key_shift = pl.key_shift;
    ea.value_bits = pl.end_bit - pl.key_shift;
    ea.power = bin_power;
    ea.scan_state = (uint64_t*)(ws + pl.o_scan_state);
    ea.n_big = n_big;
    ea.out_capacity = b->out_capacity;
    ea.out = b->d_out;
    ea.out_off = b->d_out_off;
    ea.status = b->d_status;
    SA_HIP_TRY(hipMemsetAsync(ea.scan_state, 0, ((size_t)b->n + 1) * 8, stream));
    hipLaunchKernelGGL(sort_emit_small<kEmitPacked>, dim3(b->n), dim3(kThreads), 0, stream, ea);
  }
  t = pl.scan32_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, tiles, tile_base, 0u, (size_t)b->n + 1,
                                     rocprim::plus<uint32_t>(), stream));
  t = pl.sort_tmp;
  if (b->n == 1) {
    // one long sequence: a device-wide sort of exactly its keys (their number is needed on the host)
    uint32_t count = 0;
    SA_HIP_TRY(hipMemcpyAsync(&count, seg_end, 4, hipMemcpyDeviceToHost, stream));
    SA_HIP_TRY(hipStreamSynchronize(stream));
    if (count)
      SA_HIP_TRY(rocprim::radix_sort_keys(tmp, t, keys, sorted, (size_t)count, 0u, pl.end_bit, stream));
  } else {
    hipLaunchKernelGGL(sort_small, dim3(b->n), dim3(kThreads), 0, stream, keys, sorted, seg_begin, seg_end,
                       tile_base, tile_count, pl.key_shift, pl.end_bit - pl.key_shift, bin_power, n_big);
    SA_HIP_TRY(hipGetLastError());
    SA_HIP_TRY(rocprim::segmented_radix_sort_keys(tmp, t, keys, sorted, (unsigned)pl.capacity, b->n,
                                                  seg_begin, big_end, 0u, pl.end_bit, stream));
  }
  UniqArgs u{};
  u.sorted = sorted;
  u.seg_begin = seg_begin;
  u.seg_end = seg_end;
  u.tile_base = tile_base;
  u.n = b->n;
  u.skip_small = b->n > 1;
  u.n_big = n_big;
  u.tile_count = tile_count;
  u.tile_off = tile_off;
  u.key_shift = pl.key_shift;
  u.out_capacity = b->out_capacity;
  u.out = b->d_out;
  u.out_off = b->d_out_off;
  u.status = b->d_status;
  const dim3 grid(std::min<uint32_t>(pl.max_tiles, 16384));
  hipLaunchKernelGGL(uniq_count, grid, dim3(kThreads), 0, stream, u);
  SA_HIP_TRY(hipGetLastError());
  t = pl.scan64_tmp;
  SA_HIP_TRY(rocprim::exclusive_scan(tmp, t, tile_count, tile_off, (uint64_t)0,
                                     (size_t)pl.max_tiles + 1, rocprim::plus<uint64_t>(), stream));
  hipLaunchKernelGGL(uniq_write, grid, dim3(kThreads), 0, stream, u);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

}  // extern "C"
