// (k,w)-minimizer extraction kernels.
//
// MinimizeBinary semantics (reference src/minimizer_binary.cpp:13-178):
//   * base codes C0 A1 T2 G3 on the forward strand, 3 - code on the other strand (:13-50), case-insensitive;
//   * k-mer value V[j] = bases j..j+k-1 read as a base-4 number, first base most significant (:53-97);
//     the other strand's value is therefore mask - V[j] (every digit complemented);
//   * per strand the running minimum is a state machine on its position p (:112-133): on the step that
//     brings in k-mer j, if j - k - 1 == p the minimum is re-searched as the leftmost strict minimum of
//     V[p+1 .. p+w] (:99-110, 120-123); otherwise p = j only when V[j] < V[p] (:126-131);
//   * window i = 0 starts from the leftmost strict minimum of V[0 .. w-1] (:149-150), window i >= 1
//     performs the step j = i + w - 1 (:161-164); each window emits the forward minimum when its value
//     is strictly smaller than the other strand's, else the other strand's (:153-158, 167-172);
//   * the emitted (value, position, strand) tuples are sorted and de-duplicated (src/utility.cpp:8-26).
//
// The state machine is sequential, so a sequence is cut into chunks of kChunk windows, one thread per
// chunk.  Invariant (for w <= k + 1): before step j the state lies in [j-k-1, j-1], i.e. at most k+1
// candidates.  A chunk starts kWarm steps early with ALL candidates (a bit mask of offsets) and steps
// the whole set; when the set has collapsed to one element before the chunk begins, the chunk's entry
// state is proven.  Chunks whose sets did not collapse (constant stretches: equal values never merge
// under a strict compare) are finished by a second kernel that walks them in order from the proven exit
// state of their predecessor.
#ifndef SA_MINIMIZER_KERNELS_H
#define SA_MINIMIZER_KERNELS_H

#include "sa_internal.h"

namespace sa {

constexpr uint32_t kChunk = 1024;  // windows per chunk
constexpr uint32_t kWarm = 160;    // warm-up steps in front of a chunk

struct MinArgs {
  const uint8_t* seq;
  const uint64_t* off;        // n + 1 sequence offsets
  uint32_t n;
  uint32_t k, w;
  uint32_t key_shift;         // key = value << key_shift | position << 1 | strand
  uint32_t max_chunks;
  const uint32_t* chunk_base; // n + 1: first chunk of every sequence
  uint32_t* counts;           // per chunk: emitted candidates (adjacent duplicates dropped)
  uint32_t* exit_f;           // per chunk: forward-strand state after the chunk's last window
  uint32_t* exit_c;           // per chunk: other-strand state
  uint8_t* resolved;          // per chunk: 1 once counts/exit are final
  const uint64_t* cand_off;   // per chunk: first slot in keys (exclusive scan of counts)
  uint64_t* keys;             // candidate keys
  int* status;
};

// forward-strand code of an ASCII base: C0 A1 T2 G3 (either case).  (c >> 1) & 3 gives A0 C1 T2 G3.
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
  const uint32_t x = (c >> 1) & 3u;
  return x ^ ((~x >> 1) & 1u);
}

// Streams the 2-bit codes of one sequence, keeping the last 64 bases in a 128-bit history.  After
// advance_to(b) the newest base is b; kmer(x) = V[x] needs x + k - 1 <= b and b - x <= 63 - k.
struct BaseStream {
  const uint8_t* s;
  unsigned __int128 hist;
  uint64_t next;      // index of the next base to load
  uint64_t buf;       // the aligned 8 bytes of the blob that contain base `next - 1`
  uint64_t buf_addr;  // their address (8-byte aligned), ~0 when nothing is buffered
  uint32_t k;
  uint32_t mask;
  __device__ __forceinline__ void init(const uint8_t* seq, uint64_t first, uint32_t k_) {
    s = seq;
    next = first;
    hist = 0;
    buf = 0;
    buf_addr = ~0ull;
    k = k_;
    mask = (k_ >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k_)) - 1u);
  }
  // Bases are fetched 8 at a time with aligned loads (the blob must be 8-byte aligned and readable up
  // to the next multiple of 8 past its end).
  __device__ __forceinline__ void advance_to(uint64_t b) {
    while (next <= b) {
      const uint64_t addr = (uint64_t)s + next;
      if ((addr & ~7ull) != buf_addr) {
        buf_addr = addr & ~7ull;
        buf = *reinterpret_cast<const uint64_t*>(buf_addr);
      }
      hist = (hist << 2) | base_code((uint32_t)(buf >> (8 * (addr & 7))) & 0xFFu);
      ++next;
    }
  }
  // forward value of the k-mer starting at base x
  __device__ __forceinline__ uint32_t kmer(uint64_t x) const {
    const uint32_t sh = 2u * (uint32_t)((next - 1) - (x + k - 1));
    return (uint32_t)(hist >> sh) & mask;
  }
};

// Strand view: forward compares values as they are, the other strand compares mask - value, i.e. the
// order is reversed.  less(a, b) is "strand value of a < strand value of b" on forward values.
template <bool FWD>
__device__ __forceinline__ bool strand_less(uint32_t a, uint32_t b) {
  return FWD ? (a < b) : (a > b);
}

// leftmost strict minimum (in strand order) of V[start .. start + w - 1]
template <bool FWD>
__device__ __forceinline__ uint32_t leftmost_min(const BaseStream& bs, uint32_t start, uint32_t w,
                                                 uint32_t* val) {
  uint32_t p = start, v = bs.kmer(start);
  for (uint32_t x = start + 1; x < start + w; ++x) {
    const uint32_t vx = bs.kmer(x);
    if (strand_less<FWD>(vx, v)) {
      v = vx;
      p = x;
    }
  }
  *val = v;
  return p;
}

// One step of the state machine for a single state (position p, forward value pv).
template <bool FWD>
__device__ __forceinline__ void step_one(const BaseStream& bs, uint32_t j, uint32_t k, uint32_t w,
                                         uint32_t vj, uint32_t& p, uint32_t& pv) {
  if (j - k - 1 == p) {  // unsigned arithmetic, as in the reference
    p = leftmost_min<FWD>(bs, p + 1, w, &pv);
  } else if (strand_less<FWD>(vj, pv)) {
    p = j;
    pv = vj;
  }
}

// One step for a SET of candidate states, bit o of M = candidate at position j - 1 - o, o in [0, k].
template <bool FWD>
__device__ __forceinline__ uint32_t step_set(const BaseStream& bs, uint32_t j, uint32_t k, uint32_t w,
                                             uint32_t vj, uint32_t M) {
  uint32_t out = 0;
  if ((M >> k) & 1u) {  // the expiring candidate p = j - k - 1
    uint32_t dummy;
    const uint32_t p2 = leftmost_min<FWD>(bs, j - k, w, &dummy);
    out |= 1u << (j - p2);  // offset in the next frame: (j + 1) - 1 - p2
  }
  uint32_t rest = M & ((1u << k) - 1u);
  while (rest) {
    const uint32_t o = __builtin_ctz(rest);
    rest &= rest - 1;
    const uint32_t pv = bs.kmer((uint64_t)j - 1 - o);
    if (strand_less<FWD>(vj, pv))
      out |= 1u;  // moves to j
    else
      out |= 1u << (o + 1);
  }
  return out;
}

__device__ __forceinline__ uint64_t make_key(uint32_t value, uint32_t pos, uint32_t strand,
                                             uint32_t shift) {
  return ((uint64_t)value << shift) | ((uint64_t)pos << 1) | strand;
}

struct ChunkRef {
  uint32_t seq;
  uint32_t first_window, end_window;  // windows [first, end) of the sequence
  uint32_t windows;                   // windows of the whole sequence
  uint64_t base;                      // offset of the sequence in the blob
};

__device__ __forceinline__ bool locate_chunk(const MinArgs& a, uint32_t chunk, ChunkRef* r) {
  if (chunk >= a.chunk_base[a.n]) return false;
  uint32_t lo = 0, hi = a.n;  // largest s with chunk_base[s] <= chunk
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.chunk_base[mid] <= chunk)
      lo = mid;
    else
      hi = mid;
  }
  r->seq = lo;
  r->base = a.off[lo];
  const uint32_t len = (uint32_t)(a.off[lo + 1] - a.off[lo]);
  r->windows = len - a.w - a.k + 2;
  r->first_window = (chunk - a.chunk_base[lo]) * kChunk;
  const uint32_t e = r->first_window + kChunk;
  r->end_window = e < r->windows ? e : r->windows;
  return true;
}

// Runs windows [first, end) of a sequence from a known state and either counts the candidates or writes
// their keys.  On entry (pf, pc) is the state AFTER window first - 1 (ignored when first == 0).
template <bool WRITE>
__device__ __forceinline__ uint32_t run_chunk(const MinArgs& a, const ChunkRef& c, uint32_t& pf,
                                              uint32_t& pc, uint64_t* out) {
  const uint32_t k = a.k, w = a.w;
  BaseStream bs;
  uint32_t vf, vc;  // forward values at pf / pc
  uint32_t i = c.first_window;
  uint32_t n_out = 0;
  uint64_t last = ~0ull;
  auto emit = [&]() {
    // strictly smaller forward-strand value wins, else the other strand (tie -> other strand)
    const uint32_t cval = bs.mask - vc;
    const uint64_t key = (vf < cval) ? make_key(vf, pf, 1, a.key_shift) : make_key(cval, pc, 0, a.key_shift);
    if (key != last) {
      if (WRITE) out[n_out] = key;
      ++n_out;
      last = key;
    }
  };
  if (i == 0) {
    bs.init(a.seq + c.base, 0, k);
    bs.advance_to((uint64_t)w - 1 + k - 1);
    pf = leftmost_min<true>(bs, 0, w, &vf);
    pc = leftmost_min<false>(bs, 0, w, &vc);
    emit();
    i = 1;
  } else {
    // states lie in [j-k-1, j-1] for the first step j = i + w - 1: prime the history from base j-k-1
    const uint32_t j = i + w - 1;
    const uint32_t lowest = (j >= k + 1) ? j - k - 1 : 0;
    bs.init(a.seq + c.base, lowest, k);
    bs.advance_to((uint64_t)j - 1 + k - 1);
    vf = bs.kmer(pf);
    vc = bs.kmer(pc);
  }
  for (; i < c.end_window; ++i) {
    const uint32_t j = i + w - 1;
    bs.advance_to((uint64_t)j + k - 1);
    const uint32_t vj = bs.kmer(j);
    step_one<true>(bs, j, k, w, vj, pf, vf);
    step_one<false>(bs, j, k, w, vj, pc, vc);
    emit();
  }
  return n_out;
}

}  // namespace sa

#endif
