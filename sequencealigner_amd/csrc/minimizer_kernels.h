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
//   * the emitted (value, position, strand) tuples are sorted and de-duplicated (src/utility.cpp:8-26);
//   * any other byte has code -1 on both strands (:28-29, :46-47): OR-ed into the value it sets every bit,
//     so a k-mer reads as 3s from its first base up to its last such byte; k-mer 0 is built without the
//     mask (:53-60) and keeps the ones above its 2k bits.
//
// The state machine is sequential, so a sequence is cut into chunks of `chunk` windows, one thread per
// chunk.  Invariant (for w <= k + 1): before step j the state lies in [j-k-1, j-1], i.e. at most k+1
// candidates.  A chunk starts kWarm steps early with ALL candidates (a bit mask of offsets) and steps the
// whole set until it has collapsed to one element, which proves the chunk's entry state; it then runs as
// a single state.  Chunks whose sets did not collapse before their first window (constant stretches:
// equal values never merge under a strict compare) are finished by a second kernel that walks them in
// order from the exit state of their predecessor.
//
// All k-mer values a step can need lie in the last 2k bases, so the history is one 64-bit register pair
// and every value is a shift and a mask.  For k > 16 the reference's 32-bit values wrap (:53-78: the left
// shifts drop the high bits, getMask saturates at 0xFFFFFFFF): a k-mer's value is its LAST 16 bases, while
// positions, the expiry rule and the window count still use k.  The machines then keep a third history word
// (LONG): the k-mer that starts `back` positions before the newest one ends `back` bases before the newest
// base, back <= k <= 31, and its 32 value bits need 2 * back + 32 <= 94 bits of history.  The other strand is never materialised: its order is the
// reverse of the forward order, so its minimum is the forward maximum.  That identity needs every byte to
// be a base: Machine only notices other bytes (other4 / other1), and a batch that has any is run again
// with MachineX, which carries a second history of "not a base" marks and both strands' values.
#ifndef SA_MINIMIZER_KERNELS_H
#define SA_MINIMIZER_KERNELS_H

#include "sa_internal.h"

namespace sa {

constexpr uint32_t kWarm = 160;   // warm-up steps in front of a chunk
constexpr uint32_t kTile = 2048;  // sorted keys per block in the unique pass

struct MinArgs {
  const uint8_t* seq;
  const uint64_t* off;         // n + 1 sequence offsets
  uint32_t n;
  uint32_t k, w;
  uint32_t key_shift;          // key = value << key_shift | position << 1 | strand
  uint32_t chunk;              // windows per chunk (multiple of 8)
  const uint32_t* chunk_base;  // n + 1: first chunk of every sequence
  const uint32_t* wbase;       // n + 1: first key slot of every sequence (exclusive scan of its windows)
  uint32_t* chunk_count;       // per chunk: keys it wrote, from key slot wbase[sequence] + its first window on
  uint32_t* exit_f;            // per chunk: forward-strand state after the chunk's last window
  uint32_t* exit_c;            // per chunk: other-strand state
  uint8_t* resolved;           // per chunk: 1 once its keys are written and its exit state is final
  uint64_t* keys;              // candidate keys, one slot per window
  int* status;
  uint32_t* other;             // set to 1 by a thread that met a byte other than ACGT / acgt
  // single-sequence calls: every key written is also counted in the histogram of the bucketed sort
  // (kFine counters, slot fine_slot(key >> fine_shift)); NULL otherwise
  uint32_t* hist;
  uint32_t fine_shift;
};

// where fine bin f's counter lives: neighbouring bins (values are dense at the low end) 1 KB apart, which
// spreads their atomics; bucket_plan's block b owns the bins b * 1024 .. b * 1024 + 1023 and reads and writes
// them as 256 pieces of 16 bytes (words e * 256 + 4b .. 4b + 3 for e = 0 .. 255)
__device__ __forceinline__ uint32_t fine_slot(uint32_t f) { return ((f & 255u) << 8) | (f >> 8); }

// forward-strand code of an ASCII base: C0 A1 T2 G3 (either case).  (c >> 1) & 3 gives A0 C1 T2 G3.
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
  const uint32_t x = (c >> 1) & 3u;
  return x ^ ((~x >> 1) & 1u);
}

// The same for 4 bytes at once: byte b of the result is the code of byte b of `c`.
__device__ __forceinline__ uint32_t base_code4(uint32_t c) {
  const uint32_t x = (c >> 1) & 0x03030303u;
  return x ^ ((~x >> 1) & 0x01010101u);
}

// Nonzero when one of the 4 bytes of `c` is not one of ACGT / acgt: (c >> 1) & 3 names the base a byte
// would be (A0 C1 T2 G3), a byte permute fetches that letter, the upper-cased byte must equal it.
__device__ __forceinline__ uint32_t other4(uint32_t c) {
  const uint32_t x = (c >> 1) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x47544341u, x) ^ (c & 0xDFDFDFDFu);
}
__device__ __forceinline__ uint32_t other1(uint32_t c) {
  return ((0x47544341u >> (8 * ((c >> 1) & 3u))) & 0xFFu) ^ (c & 0xDFu);
}

__device__ __forceinline__ uint64_t make_key(uint32_t value, uint32_t pos, uint32_t strand,
                                             uint32_t shift) {
  return ((uint64_t)value << shift) | ((uint64_t)pos << 1) | strand;
}

// The two-strand state machine of one thread.  W > 0 fixes the window length at compile time (the scan
// over the window unrolls); W = 0 takes it from reset().
template <uint32_t W, bool LONG = false>
struct Machine {
  // how the kernels feed bases: feed1 / feed4 turn 1 / 4 bytes into the per-base argument(s) of push() and
  // step() (8 bits apart in feed4), kFeedMask extracts one; `other` collects evidence of non-base bytes
  static constexpr bool kExact = false;  // false: results are valid only while `other` stays 0
  static constexpr uint32_t kFeedMask = 3u;
  __device__ static __forceinline__ uint32_t feed1(uint32_t byte, uint32_t& other) {
    other |= other1(byte);
    return base_code(byte);
  }
  __device__ static __forceinline__ uint32_t feed4(uint32_t bytes, uint32_t& other) {
    other |= other4(bytes);
    return base_code4(bytes);
  }

  uint32_t lo, hi, hi2;     // the last 32 (LONG: 48) base codes, newest in the low bits of lo
  uint32_t pf, vf, pc, vc;  // per strand: position of the minimum and its FORWARD value
  uint32_t j;               // k-mer the next step brings in (the step consumes base j + k - 1)
  uint32_t k, w, mask;
  // W > 0: the w k-mers an expiring minimum is re-searched in, V[j-1-k .. j-1-k+w-1] after step j-1; each
  // step drops the oldest and extracts one new value instead of all w
  uint32_t win[W ? W : 1];

  __device__ __forceinline__ void reset(uint32_t k_, uint32_t w_) {
    k = k_;
    w = W ? W : w_;
    mask = (k_ >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k_)) - 1u);
    lo = hi = hi2 = 0;
    pf = pc = vf = vc = j = 0;
  }
  __device__ __forceinline__ void push(uint32_t code) {
    if (LONG) hi2 = __builtin_amdgcn_alignbit(hi2, hi, 30);
    hi = __builtin_amdgcn_alignbit(hi, lo, 30);
    lo = (lo << 2) | code;
  }
  // forward value of the k-mer that starts `back` positions before the newest complete k-mer
  __device__ __forceinline__ uint32_t value_back(uint32_t back) const {
    if (LONG) {  // bits [2 * back, 2 * back + 32) of hi2:hi:lo
      const uint32_t sh = 2 * back;
      const uint32_t w0 = sh < 32 ? lo : hi, w1 = sh < 32 ? hi : hi2;
      return __builtin_amdgcn_alignbit(w1, w0, sh & 31u);
    }
    const uint64_t h = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(h >> (2 * back)) & mask;
  }
  // Leftmost strict minimum and leftmost strict maximum (= the other strand's minimum) of the w k-mers
  // that start `back`, `back - 1`, ... positions before the newest one; t* = offsets from the first.
  __device__ __forceinline__ void scan(uint32_t back, uint32_t* tmin, uint32_t* vmin, uint32_t* tmax,
                                       uint32_t* vmax) const {
    uint32_t a = value_back(back), b = a, ta = 0, tb = 0;
#pragma unroll
    for (uint32_t t = 1; t < (W ? W : w); ++t) {
      const uint32_t v = value_back(back - t);
      if (v < a) {
        a = v;
        ta = t;
      }
      if (v > b) {
        b = v;
        tb = t;
      }
    }
    *tmin = ta;
    *vmin = a;
    *tmax = tb;
    *vmax = b;
  }
  // (Re)builds win[] from the history; the newest complete k-mer must be j - 1.
  __device__ __forceinline__ void init_window() {
#pragma unroll
    for (uint32_t t = 0; t < W; ++t) win[t] = value_back(k - t);
  }
  // Pushes base j + k - 1 and finds the re-search result of step j: leftmost strict minimum / maximum of
  // V[j-k .. j-k+w-1] as offsets from j-k.
  __device__ __forceinline__ void advance(uint32_t code, uint32_t* tmin, uint32_t* vmin, uint32_t* tmax,
                                          uint32_t* vmax) {
    push(code);
    if (W == 0) {
      scan(k, tmin, vmin, tmax, vmax);
      return;
    }
#pragma unroll
    for (uint32_t t = 0; t + 1 < W; ++t) win[t] = win[t + 1];
    win[W ? W - 1 : 0] = value_back(k - (W - 1));
    uint32_t a = win[0], b = a, ta = 0, tb = 0;
#pragma unroll
    for (uint32_t t = 1; t < W; ++t) {
      const uint32_t v = win[t];
      if (v < a) {
        a = v;
        ta = t;
      }
      if (v > b) {
        b = v;
        tb = t;
      }
    }
    *tmin = ta;
    *vmin = a;
    *tmax = tb;
    *vmax = b;
  }
  // Window 0 (src/minimizer_binary.cpp:149-150) once bases 0 .. w+k-2 have been pushed.
  __device__ __forceinline__ void first_window() {
    scan(w - 1, &pf, &vf, &pc, &vc);
    j = w;
    init_window();
  }
  // Enters at step j_ with known positions, once bases up to j_ + k - 2 have been pushed.
  __device__ __forceinline__ void enter(uint32_t j_, uint32_t pf_, uint32_t pc_) {
    j = j_;
    pf = pf_;
    pc = pc_;
    vf = value_back(j_ - 1 - pf_);
    vc = value_back(j_ - 1 - pc_);
    init_window();
  }
  // One step (src/minimizer_binary.cpp:112-133 for both strands): consumes base j + k - 1.
  __device__ __forceinline__ void step(uint32_t code) {
    uint32_t tf, rf, tc, rc;
    advance(code, &tf, &rf, &tc, &rc);  // V[j-k .. j-k+w-1], used by a strand whose minimum expires
    const uint32_t vj = lo & mask;
    const uint32_t expiring = j - k - 1;  // unsigned, as in the reference
    const bool ef = pf == expiring, ec = pc == expiring;
    const bool lf = vj < vf, lc = vj > vc;
    pf = ef ? j - k + tf : (lf ? j : pf);
    vf = ef ? rf : (lf ? vj : vf);
    pc = ec ? j - k + tc : (lc ? j : pc);
    vc = ec ? rc : (lc ? vj : vc);
    ++j;
  }
  // The tuple this window emits (src/minimizer_binary.cpp:153-158, 167-172): its value and
  // position << 1 | strand.
  __device__ __forceinline__ void tuple(uint32_t* value, uint32_t* pos2) const {
    const uint32_t cval = mask - vc;
    const bool f = vf < cval;
    *value = f ? vf : cval;
    *pos2 = f ? (pf << 1 | 1u) : (pc << 1);
  }
  __device__ __forceinline__ uint64_t key(uint32_t shift) const {
    uint32_t value, pos2;
    tuple(&value, &pos2);
    return ((uint64_t)value << shift) | pos2;
  }

  // One step for SETS of candidate states: bit o of M = candidate at position j - 1 - o, o in [0, k].
  __device__ __forceinline__ void step_sets(uint32_t code, uint32_t& Mf, uint32_t& Mc) {
    uint32_t tf, rf, tc, rc;
    advance(code, &tf, &rf, &tc, &rc);
    const uint32_t vj = lo & mask;
    const uint32_t low = (1u << k) - 1u;
    uint32_t of = ((Mf >> k) & 1u) ? 1u << (k - tf) : 0u;  // offset in the next frame: j - (j-k+t)
    uint32_t oc = ((Mc >> k) & 1u) ? 1u << (k - tc) : 0u;
    for (uint32_t rest = Mf & low; rest; rest &= rest - 1) {
      const uint32_t o = __builtin_ctz(rest);
      of |= (vj < value_back(o + 1)) ? 1u : (2u << o);
    }
    for (uint32_t rest = Mc & low; rest; rest &= rest - 1) {
      const uint32_t o = __builtin_ctz(rest);
      oc |= (vj > value_back(o + 1)) ? 1u : (2u << o);
    }
    Mf = of;
    Mc = oc;
    ++j;
  }
};

// The state machine for sequences with bytes other than bases: same interface as Machine, both strands'
// values explicit.  A second history holds a mark (binary 11) for every such byte; a k-mer's value is its
// codes OR-ed with ones from its newest mark upwards - within the 2k-bit mask, except for k-mer 0.
template <bool LONG = false>
struct MachineX {
  static constexpr bool kExact = true;
  static constexpr uint32_t kFeedMask = 7u;  // code | mark << 2
  __device__ static __forceinline__ uint32_t feed1(uint32_t byte, uint32_t& other) {
    (void)other;
    return base_code(byte) | (other1(byte) ? 4u : 0u);
  }
  __device__ static __forceinline__ uint32_t feed4(uint32_t bytes, uint32_t& other) {
    (void)other;
    const uint32_t o = other4(bytes);
    // 0x04 in every byte of o that is not zero
    const uint32_t nz = (((o & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | o) & 0x80808080u;
    return base_code4(bytes) | (nz >> 5);
  }

  uint32_t lo, hi, hi2, mlo, mhi, mhi2;  // codes and marks of the last 32 (LONG: 48) bytes, newest in the low bits
  uint32_t pf, vf, pc, vc;               // per strand: position of the minimum and the strand's own value
  uint32_t j, k, w, mask;

  __device__ __forceinline__ void reset(uint32_t k_, uint32_t w_) {
    k = k_;
    w = w_;
    mask = (k_ >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k_)) - 1u);
    lo = hi = hi2 = mlo = mhi = mhi2 = 0;
    pf = pc = vf = vc = j = 0;
  }
  __device__ __forceinline__ void push(uint32_t fed) {
    if (LONG) hi2 = __builtin_amdgcn_alignbit(hi2, hi, 30);
    hi = __builtin_amdgcn_alignbit(hi, lo, 30);
    lo = (lo << 2) | (fed & 3u);
    if (LONG) mhi2 = __builtin_amdgcn_alignbit(mhi2, mhi, 30);
    mhi = __builtin_amdgcn_alignbit(mhi, mlo, 30);
    mlo = (mlo << 2) | ((fed & 4u) ? 3u : 0u);
  }
  __device__ static __forceinline__ uint32_t window32(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t back) {
    if (LONG) {  // bits [2 * back, 2 * back + 32) of w2:w1:w0
      const uint32_t sh = 2 * back;
      return __builtin_amdgcn_alignbit(sh < 32 ? w1 : w2, sh < 32 ? w0 : w1, sh & 31u);
    }
    return (uint32_t)((((uint64_t)w1 << 32) | w0) >> (2 * back));
  }
  // Both strands' values of k-mer `index`, which starts `back` positions before the newest complete one.
  __device__ __forceinline__ void values(uint32_t back, uint32_t index, uint32_t* f, uint32_t* c) const {
    const uint32_t raw = window32(lo, hi, hi2, back) & mask;
    const uint32_t marks = window32(mlo, mhi, mhi2, back) & mask;
    uint32_t ones = marks ? 0u - (marks & (0u - marks)) : 0u;  // from the newest mark upwards
    if (index != 0) ones &= mask;
    *f = raw | ones;
    *c = (mask - raw) | ones;
  }
  // Leftmost strict minima of the w k-mers first .. first + w - 1; the newest complete k-mer is `newest`.
  __device__ __forceinline__ void scan(uint32_t first, uint32_t newest, uint32_t* pmin_f, uint32_t* vmin_f,
                                       uint32_t* pmin_c, uint32_t* vmin_c) const {
    uint32_t a, b;
    values(newest - first, first, &a, &b);
    uint32_t pa = first, pb = first;
    for (uint32_t t = 1; t < w; ++t) {
      uint32_t f, c;
      values(newest - first - t, first + t, &f, &c);
      if (f < a) {
        a = f;
        pa = first + t;
      }
      if (c < b) {
        b = c;
        pb = first + t;
      }
    }
    *pmin_f = pa;
    *vmin_f = a;
    *pmin_c = pb;
    *vmin_c = b;
  }
  __device__ __forceinline__ void init_window() {}
  __device__ __forceinline__ void first_window() {  // bases 0 .. w+k-2 pushed: newest k-mer w - 1
    scan(0, w - 1, &pf, &vf, &pc, &vc);
    j = w;
  }
  __device__ __forceinline__ void enter(uint32_t j_, uint32_t pf_, uint32_t pc_) {  // newest k-mer j_ - 1
    uint32_t unused;
    j = j_;
    pf = pf_;
    pc = pc_;
    values(j_ - 1 - pf_, pf_, &vf, &unused);
    values(j_ - 1 - pc_, pc_, &unused, &vc);
  }
  __device__ __forceinline__ void step(uint32_t fed) {
    push(fed);
    uint32_t vjf, vjc;
    values(0, j, &vjf, &vjc);
    const uint32_t expiring = j - k - 1;  // unsigned, as in the reference
    const bool ef = pf == expiring, ec = pc == expiring;
    if (ef || ec) {
      uint32_t qf, rf, qc, rc;
      scan(j - k, j, &qf, &rf, &qc, &rc);
      if (ef) {
        pf = qf;
        vf = rf;
      }
      if (ec) {
        pc = qc;
        vc = rc;
      }
    }
    if (!ef && vjf < vf) {
      pf = j;
      vf = vjf;
    }
    if (!ec && vjc < vc) {
      pc = j;
      vc = vjc;
    }
    ++j;
  }
  __device__ __forceinline__ void tuple(uint32_t* value, uint32_t* pos2) const {
    const bool f = vf < vc;
    *value = f ? vf : vc;
    *pos2 = f ? (pf << 1 | 1u) : (pc << 1);
  }
  __device__ __forceinline__ uint64_t key(uint32_t shift) const {
    uint32_t value, pos2;
    tuple(&value, &pos2);
    return ((uint64_t)value << shift) | pos2;
  }
  // One step for sets of candidate states (see Machine::step_sets).
  __device__ __forceinline__ void step_sets(uint32_t fed, uint32_t& Mf, uint32_t& Mc) {
    push(fed);
    uint32_t vjf, vjc, qf, rf, qc, rc;
    values(0, j, &vjf, &vjc);
    scan(j - k, j, &qf, &rf, &qc, &rc);
    const uint32_t low = (1u << k) - 1u;
    uint32_t of = ((Mf >> k) & 1u) ? 1u << (j - qf) : 0u;  // offset in the next frame
    uint32_t oc = ((Mc >> k) & 1u) ? 1u << (j - qc) : 0u;
    for (uint32_t rest = Mf & low; rest; rest &= rest - 1) {
      const uint32_t o = __builtin_ctz(rest);
      uint32_t f, c;
      values(o + 1, j - 1 - o, &f, &c);
      of |= (vjf < f) ? 1u : (2u << o);
    }
    for (uint32_t rest = Mc & low; rest; rest &= rest - 1) {
      const uint32_t o = __builtin_ctz(rest);
      uint32_t f, c;
      values(o + 1, j - 1 - o, &f, &c);
      oc |= (vjc < c) ? 1u : (2u << o);
    }
    Mf = of;
    Mc = oc;
    ++j;
  }
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
    v = o > v ? o : v;
  }
  return v;
}

struct ChunkRef {
  uint32_t seq;
  uint32_t first_window, end_window;  // windows [first, end) of the sequence
  uint64_t base;                      // offset of the sequence in the blob
  uint32_t len;
};

__device__ __forceinline__ bool locate_chunk(const MinArgs& a, uint32_t chunk, ChunkRef* r) {
  const uint32_t total = a.chunk_base[a.n];
  if (chunk >= total) return false;
  uint32_t lo = 0, hi = a.n;  // largest s with chunk_base[s] <= chunk
  {  // sequences of about one length: the proportional guess is right or nearly so (a bisection over 10^5
     // sequences is 17 dependent loads in front of everything else the thread does)
    uint32_t g = (uint32_t)(((uint64_t)chunk * a.n) / total);
    if (g >= a.n) g = a.n - 1;
    const uint32_t cb = a.chunk_base[g], cb1 = a.chunk_base[g + 1];
    if (cb > chunk) {
      hi = g;
    } else if (cb1 <= chunk) {
      lo = g + 1;
    } else {
      lo = g;
      hi = g + 1;
    }
  }
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.chunk_base[mid] <= chunk)
      lo = mid;
    else
      hi = mid;
  }
  r->seq = lo;
  r->base = a.off[lo];
  r->len = (uint32_t)(a.off[lo + 1] - a.off[lo]);
  const uint32_t windows = r->len - a.w - a.k + 2;
  r->first_window = (chunk - a.chunk_base[lo]) * a.chunk;
  const uint32_t e = r->first_window + a.chunk;
  r->end_window = e < windows ? e : windows;
  return true;
}

}  // namespace sa

#endif
