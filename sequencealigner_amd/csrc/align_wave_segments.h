// Traceback of a FEW LONG pairs by many wavefronts.  The path of one pair is a chain: where it crosses a row is
// known only once it has been followed down to that row, so one wavefront walks 200 000 cells of a
// 100 kbp x 100 kbp pair alone while the GPU idles.  But a path is a function of the cell it starts from (every
// cell has ONE direction), so two walks that meet stay together, and walks that start on the same row a few
// columns apart do meet after a few moves when the sequences are related.  Hence:
//   * the target rows are cut into segments of `rows` rows, one wavefront each.  A segment's walk starts on
//     its top row (the topmost segment: at the end cell) and stops at the first cell it reaches on the top row
//     of the segment below; its runs go to the segment's own part of the pair's run slot;
//   * every segment walks at once from a GUESSED start column (on or to the right of the likely path) and publishes
//     the column it lands on, marked "speculative".  A segment that sees the segment above publish a column other
//     than the one it started from walks again from that column.  The topmost segment's landing is final; a
//     segment whose upstream landing is final AND equal to the column it last started from marks its own
//     landing final.  Finality moves down one segment per memory round trip, the walks themselves run in
//     parallel, and whatever the guesses were, every segment ends up walking from the true path's column: the
//     result is the path the single wavefront finds, run for run after `stitch_segments` has joined the parts;
//   * speculation is bounded: a segment walks at most twice on speculative columns, and such a walk gives up
//     after `spec_iters` iterations (a path with a gap of tens of thousands of columns is far from every guess,
//     and walks that never meet it would queue up one behind the other).  A segment that has given up waits for
//     the final column, so the worst case is the one wavefront's walk cut into pieces plus the bounded guesses.
// A published landing is one 64-bit word {column, state}, written and read with single device-scope accesses.
// Segments take their numbers from a ticket counter, topmost first, so a waiting wavefront waits only for
// wavefronts that already run; every wait is bounded and raises the abort flag.  A segment whose runs outgrow
// its part of the slot marks the pair, which the one-wavefront kernel then walks again from the start.
// Global and Semiglobal (Local's stop rule needs the score carried along the path).
#ifndef SA_ALIGN_WAVE_SEGMENTS_H
#define SA_ALIGN_WAVE_SEGMENTS_H

#include "align_wave_i32.h"

namespace sa {

constexpr uint32_t kMaxSegments = 1024;  // per pair (arrays of stitch_segments)
constexpr uint32_t kSegEnded = 0xFFFFFFFFu;  // "the path ended above this row" in place of a column
enum : uint32_t { SEG_NONE = 0, SEG_SPECULATIVE = 1, SEG_FINAL = 2 };

struct SegArgs {
  uint32_t rows;        // target rows per segment
  uint32_t nseg;        // segments per pair: ceil(max_t / rows), at most kMaxSegments
  uint32_t cap;         // runs a segment may write: cigar_stride / nseg
  uint32_t spec_iters;  // iterations a speculative walk may take
  uint32_t spec_walks;  // walks a segment may make on speculative columns
  uint64_t* pub;        // [pair of the pass][nseg]: landing of segment k = start of segment k - 1; zero at launch
  uint32_t* seg_nruns;  // [pair of the pass][nseg]; zero at launch
  uint32_t* redo;       // [pair of the pass]: nonzero = walk this pair with one wavefront instead; zero at launch
  uint32_t* ticket;     // zero at launch
  int* abort_flag;
  uint32_t spin_limit;  // polls a wait may take before it gives up
};

// words to clear before a launch, starting at SegArgs::pub (the three arrays are laid out back to back)
__host__ __device__ inline size_t seg_state_words(size_t pairs, uint32_t nseg) {
  return pairs * nseg * 2 + pairs * nseg + pairs;
}

template <int MODE, class Fmt>
__global__ __launch_bounds__(64) void traceback_wave_segments(AlignArgs a, SegArgs sg) {
  static_assert(MODE != SA_LOCAL, "Global and Semiglobal only");
  const uint32_t lane = threadIdx.x;
  if (a.gave_up && *a.gave_up) return;  // the fill is incomplete: nothing to walk
  uint32_t ticket = 0;
  if (lane == 0) ticket = atomicAdd(sg.ticket, 1u);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  const uint32_t pl = ticket / sg.nseg, k = sg.nseg - 1 - ticket % sg.nseg;  // topmost segments first
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
  const uint32_t T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  const uint32_t i_end = __builtin_amdgcn_readfirstlane((MODE == SA_GLOBAL) ? T : min(a.end_i[pl], T));
  const uint32_t ktop = i_end ? (i_end - 1) / sg.rows : 0;
  if (k > ktop) return;  // above the end cell: no runs (seg_nruns stays 0)
  const uint32_t lo = k * sg.rows;
  const uint32_t hi = k == ktop ? i_end : lo + sg.rows;
  uint32_t* part_end = a.cigar_runs + (uint64_t)p * a.cigar_stride + (uint64_t)(k + 1) * sg.cap;
  uint64_t* pub_out = sg.pub + (uint64_t)pl * sg.nseg + k;
  const uint64_t* pub_in = pub_out + 1;  // read only when k < ktop

  uint32_t land = 0, nruns = 0;
  bool overflow = false, gave_up = false;
  auto walk = [&](uint32_t col, uint32_t max_iters) __attribute__((always_inline)) {
    gave_up = false;
    if (col == kSegEnded) {
      land = kSegEnded;
      nruns = 0;
      overflow = false;
      return;
    }
    const WalkEnd e = walk_wave<MODE, Fmt>(a, pl, hi, col, lo, part_end, sg.cap, max_iters);
    land = (lo && e.i > lo) ? kSegEnded : e.r;  // stopped above row lo: the path ended there
    nruns = e.nruns;
    overflow = e.overflow;
    gave_up = e.gave_up;
  };
  auto publish = [&](uint32_t state) __attribute__((always_inline)) {
    if (k && lane == 0)
      __hip_atomic_store(pub_out, ((uint64_t)state << 32) | land, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto finish = [&]() __attribute__((always_inline)) {
    if (lane == 0) {
      sg.seg_nruns[(uint64_t)pl * sg.nseg + k] = nruns;
      if (overflow) sg.redo[pl] = 1;
    }
  };

  auto upstream = [&]() -> uint64_t {
    const uint64_t u = __hip_atomic_load(pub_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  };
  // the column to walk from next and what is known about it
  uint32_t next_col = Q, next_state = SEG_FINAL;  // the topmost segment starts at the end cell
  bool pending = true;
  if (k < ktop) {
    pending = sg.spec_walks && (uint32_t)(upstream() >> 32) == SEG_NONE;
    if (pending) {
      // Nothing known yet.  A walk that starts to the RIGHT of the path reaches it along its own row (a run of
      // Left moves, 64 per iteration, no target row used up), one that starts to the left of it needs Up moves
      // and may leave the segment before it gets there: so the guess is the rightmost of three lines - the
      // straight line from the origin to the end cell and the diagonals through the origin and the end cell.
      const uint64_t straight = (uint64_t)hi * Q / i_end;
      const uint64_t from_origin = hi;
      const uint64_t from_end = (uint64_t)Q + hi > i_end ? (uint64_t)Q + hi - i_end : 0;
      const uint64_t right = straight > from_origin ? (straight > from_end ? straight : from_end)
                                                    : (from_origin > from_end ? from_origin : from_end);
      next_col = (uint32_t)(right < Q ? right : Q);
      next_state = SEG_SPECULATIVE;
    }
  }
  bool walked = false;        // `land` and the runs are those of a finished walk from column `used`
  uint32_t used = 0;
  uint32_t spec_left = sg.spec_walks;  // walks this segment may still make on speculative columns
  for (uint32_t polls = 0;;) {
    if (!pending) {
      const uint64_t u = upstream();
      const uint32_t state = (uint32_t)(u >> 32), col = (uint32_t)u;
      const bool fresh = state == SEG_FINAL || (state == SEG_SPECULATIVE && spec_left && !(walked && col == used));
      if (!fresh) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls >= sg.spin_limit ||
            ((polls & 1023u) == 1023u && __hip_atomic_load(sg.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(sg.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
        continue;
      }
      next_col = col;
      next_state = state;
    }
    pending = false;
    if (!walked || next_col != used) {
      const bool final_col = next_state == SEG_FINAL;
      if (!final_col) --spec_left;
      walk(next_col, final_col ? ~0u : sg.spec_iters);
      used = next_col;
      walked = !gave_up;
    }
    if (walked) publish(next_state);
    if (next_state == SEG_FINAL) break;
  }
  finish();
}

// Joins the segments' runs of a pair into one list at the end of the pair's slot, in place: neighbouring runs
// of the same kind across a segment border become one, and the runs move up to their final places (a run's new
// place is never below its old one) in chunks from the top, with a barrier between a chunk's loads and its
// stores.  Writes the run count and target_begin.  One block per pair.
template <int MODE>
__global__ __launch_bounds__(1024) void stitch_segments(AlignArgs a, SegArgs sg) {
  __shared__ uint32_t n_[kMaxSegments], first_[kMaxSegments], last_[kMaxSegments];
  __shared__ uint32_t dstart_[kMaxSegments], drop_[kMaxSegments];
  __shared__ uint32_t patch_at[kMaxSegments], patch_val[kMaxSegments];
  __shared__ uint32_t s_total, s_np;
  const uint32_t pl = blockIdx.x, tid = threadIdx.x;
  if (pl >= a.pairs_in_pass || sg.redo[pl] || (a.gave_up && *a.gave_up)) return;
  const uint32_t p = a.pair_base + pl;
  uint32_t* slot = a.cigar_runs + (uint64_t)p * a.cigar_stride;
  const uint32_t K = sg.nseg;
  for (uint32_t k = tid; k < K; k += blockDim.x) {
    const uint32_t n = sg.seg_nruns[(uint64_t)pl * K + k];
    const uint32_t* part_end = slot + (uint64_t)(k + 1) * sg.cap;
    n_[k] = n;
    first_[k] = n ? part_end[-(int64_t)n] : 0;
    last_[k] = n ? part_end[-1] : 0;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0, np = 0, last_at = 0, lastv = 0;
    bool have = false;
    for (uint32_t k = 0; k < K; ++k) {
      const uint32_t nk = n_[k];
      drop_[k] = 0;
      dstart_[k] = total;
      if (!nk) continue;
      const uint32_t f = first_[k];
      if (have && (f & 3u) == (lastv & 3u)) {  // continues the run the part below ended with
        lastv += f & ~3u;
        drop_[k] = 1;
      }
      const uint32_t kept = nk - drop_[k];
      if (kept) {
        if (have) {
          patch_at[np] = last_at;
          patch_val[np++] = lastv;
        }
        last_at = total + kept - 1;
        lastv = last_[k];
        have = true;
      }
      total += kept;
    }
    if (have) {
      patch_at[np] = last_at;
      patch_val[np++] = lastv;
    }
    s_total = total;
    s_np = np;
  }
  __syncthreads();
  const uint32_t total = s_total;
  uint32_t* list = slot + a.cigar_stride - total;  // the pair's runs end at the end of its slot
  // The kept runs in the order of the final list, from its end: a chunk of them is loaded (every run finds its
  // part by bisection on the parts' first indices), then stored.  A run never moves down and chunks go from
  // the top, so a chunk's stores cannot reach a run that a later chunk still has to load.
  constexpr uint32_t kPer = 4;
  for (uint32_t off = 0; off < total; off += kPer * blockDim.x) {
    uint32_t v[kPer];
#pragma unroll
    for (uint32_t x = 0; x < kPer; ++x) {
      const uint32_t back = off + x * blockDim.x + tid;
      v[x] = 0;
      if (back < total) {
        const uint32_t f = total - 1 - back;
        uint32_t lo_k = 0, hi_k = K - 1;  // the last part whose first index is <= f (it keeps a run: an empty
                                          // part shares its first index with the part above it)
        while (lo_k < hi_k) {
          const uint32_t mid = (lo_k + hi_k + 1) / 2;
          if (dstart_[mid] <= f)
            lo_k = mid;
          else
            hi_k = mid - 1;
        }
        const uint32_t kept = n_[lo_k] - drop_[lo_k];
        v[x] = slot[(uint64_t)(lo_k + 1) * sg.cap - kept + (f - dstart_[lo_k])];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t x = 0; x < kPer; ++x) {
      const uint32_t back = off + x * blockDim.x + tid;
      if (back < total) list[total - 1 - back] = v[x];
    }
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t x = tid; x < s_np; x += blockDim.x) list[patch_at[x]] = patch_val[x];
  if (tid == 0) {
    if (a.cigar_nruns) a.cigar_nruns[p] = total;
    if (a.target_begin) {
      // the first run of the list: the lowest part's first run, or the patched run at index 0
      uint32_t f = 0;
      for (uint32_t k = 0; k < K && !f; ++k) f = n_[k] ? first_[k] : 0;
      if (s_np && patch_at[0] == 0) f = patch_val[0];
      a.target_begin[p] = target_begin_of_first_run(f & 3u, f >> 2);
    }
  }
}

}  // namespace sa

#endif
