// Device-side pieces shared by the alignment kernels: argument blocks, border values, the layouts of
// the 2-bit direction arrays and the traceback kernel that reads them.
#ifndef SA_ALIGN_COMMON_H
#define SA_ALIGN_COMMON_H

#include "sa_internal.h"

namespace sa {

struct AlignArgs {
  const uint8_t* q;
  const uint64_t* q_off;
  const uint8_t* t;
  const uint64_t* t_off;
  uint32_t pair_base;      // first pair of this pass
  uint32_t pairs_in_pass;  // pairs handled by this launch
  uint32_t max_q, max_t;   // capacity the workspace was laid out for
  int match, mismatch, gap;
  // workspace
  uint32_t* dirs;            // direction words
  uint64_t dirs_unit_stride; // dwords per wave (lane kernels) or per pair (wavefront kernels)
  int32_t* brow;             // strip boundary values
  uint64_t brow_unit_stride; // ints per wave / per pair
  uint32_t nstrips;
  int32_t* end_score;        // per pair of the pass: score found by the fill
  uint32_t* end_i;           // per pair: row (target index, 1-based) where the traceback starts
  uint32_t* end_r;           // per pair: column (query index, 1-based) where the traceback starts
  // outputs
  int32_t* score;
  uint32_t* target_begin;
  uint32_t* cigar_runs;
  uint32_t* cigar_nruns;
  uint64_t cigar_stride;
  // packed lane kernel, tables form only (else rec == NULL): the pairs' sequences once more as 2-bit codes, one
  // contiguous record per pair - 16-byte chunks of 64 symbols, rec_q_chunks for the query, then the target's -
  // for the traceback, which compares symbols of one pair at a time
  const uint32_t* rec;
  uint64_t rec_unit_stride;  // dwords between the records of consecutive units of 128 pairs
  uint32_t rec_stride;       // dwords per pair
  uint32_t rec_q_chunks;
  const struct AlphabetInfo* rec_alpha;  // the records are valid when small && !violated
  // traceback_wave: when set, only the pairs (of the pass) with a nonzero word are walked
  const uint32_t* tb_only;
  // cooperative kernels: set when a wait ran out; the fill's results are then incomplete and the traceback
  // kernels return at once (the call fails)
  const int* gave_up;
  // packed cooperative kernel: target rows per segment of a strip (a multiple of 64; 0: a strip is one piece)
  // and segments per strip
  uint32_t seg_rows, nseg;
};

// Border values of the DP matrix, rows = target (index i), columns = query (index r).
//   Global:      H[i][0] = gap*i (Up),  H[0][r] = gap*r (Left)          (src/alignment.cpp:263-269)
//   Local:       all 0 (None) when gap < 0, else as Global              (src/alignment.cpp:152-168)
//   Semiglobal:  H[i][0] = 0 (None),    H[0][r] = gap*r (Left)          (src/alignment.cpp:46-52)
template <int MODE>
__device__ __forceinline__ int border_col0(int i, int gap) {
  if (MODE == SA_GLOBAL) return gap * i;
  if (MODE == SA_LOCAL) return gap < 0 ? 0 : gap * i;
  return 0;
}
template <int MODE>
__device__ __forceinline__ int border_row0(int r, int gap) {
  if (MODE == SA_LOCAL) return gap < 0 ? 0 : gap * r;
  return gap * r;
}
template <int MODE>
__device__ __forceinline__ uint32_t border_dir(uint32_t i, uint32_t r, int gap) {
  if (i == 0 && r == 0) return DIR_NONE;
  if (r == 0) {
    if (MODE == SA_GLOBAL) return DIR_UP;
    if (MODE == SA_LOCAL) return gap < 0 ? DIR_NONE : DIR_UP;
    return DIR_NONE;
  }
  if (MODE == SA_LOCAL) return gap < 0 ? DIR_NONE : DIR_LEFT;
  return DIR_LEFT;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off);
    v = o > v ? o : v;
  }
  return v;
}

// ---- direction-array layouts ---------------------------------------------------------------------
// A layout maps (pair within pass, i, r) with 1 <= i <= T, 1 <= r <= Q to a 2-bit field.

// Lane kernels, one pair per lane, 32-bit scores: a strip is K target rows; for each query column r the
// lane stores K/16 dwords, 16 rows per dword.  Words of the 64 lanes of a wave are interleaved so that a
// wave's store is one contiguous 256-byte line.
template <int K>
struct LaneI32Layout {
  static constexpr bool kImplicitNone = false;
  const uint32_t* dirs;
  uint64_t wave_stride;
  uint32_t max_q;
  __device__ __forceinline__ uint32_t get(uint32_t pair_local, uint32_t i, uint32_t r) const {
    const uint32_t wave = pair_local >> 6, lane = pair_local & 63;
    const uint32_t s = (i - 1) / K, k = (i - 1) % K;
    const uint64_t w = ((uint64_t)(s * max_q + (r - 1)) * (K / 16) + (k >> 4)) * 64 + lane;
    return (dirs[wave * wave_stride + w] >> (2 * (k & 15))) & 3u;
  }
};

// Traceback + run-length encoding, one pair per thread.  Restates the three traceback loops and
// CompressCigar of the reference (src/alignment.cpp:98-136, 212-246, 311-348, 17-33):
//   Diag -> 'M' when score - match == diagonal score, i.e. when the symbols are equal or
//           match == mismatch, else 'X';  Left -> 'I' (consumes query);  Up -> 'D' (consumes target).
// Runs are produced last-first, so they are stored downwards from the end of the pair's slot range.
template <int MODE, class Layout>
__global__ __launch_bounds__(256) void traceback_kernel(AlignArgs a, Layout lay) {
  const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint8_t* q = a.q + a.q_off[p];
  const uint8_t* t = a.t + a.t_off[p];
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
  const uint32_t T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  uint32_t i = (MODE == SA_GLOBAL) ? T : a.end_i[pl];
  uint32_t r = (MODE == SA_LOCAL) ? a.end_r[pl] : Q;
  const bool same = (a.match == a.mismatch);

  uint32_t* slot_end = a.cigar_runs ? a.cigar_runs + (uint64_t)(p + 1) * a.cigar_stride : nullptr;
  uint32_t nruns = 0, cur_op = 4, cur_len = 0;
  // Layouts that do not store Local's None state: a cell is None exactly when its score is 0
  // (src/alignment.cpp:190-192 stores {0, None} for max <= 0 and a positive score otherwise), so the
  // traceback carries the score of the current cell and stops when it reaches 0.
  int remaining = (MODE == SA_LOCAL && Layout::kImplicitNone) ? a.score[p] : 1;
  for (;;) {
    if (MODE == SA_GLOBAL && i == 0 && r == 0) break;
    if (MODE == SA_SEMIGLOBAL && r == 0) break;
    if (MODE == SA_LOCAL && Layout::kImplicitNone && i != 0 && r != 0 && remaining == 0) break;
    const uint32_t d = (i == 0 || r == 0) ? border_dir<MODE>(i, r, a.gap) : lay.get(pl, i, r);
    if (d == DIR_NONE) break;
    uint32_t op;
    if (d == DIR_DIAG) {
      const bool eq = t[i - 1] == q[r - 1];
      op = (same || eq) ? SA_OP_M : SA_OP_X;
      if (MODE == SA_LOCAL && Layout::kImplicitNone) remaining -= eq ? a.match : a.mismatch;
      --i;
      --r;
    } else if (d == DIR_LEFT) {
      op = SA_OP_I;
      if (MODE == SA_LOCAL && Layout::kImplicitNone) remaining -= a.gap;
      --r;
    } else {
      op = SA_OP_D;
      if (MODE == SA_LOCAL && Layout::kImplicitNone) remaining -= a.gap;
      --i;
    }
    if (op == cur_op) {
      ++cur_len;
    } else {
      if (cur_len) {
        ++nruns;
        if (slot_end) slot_end[-(int64_t)nruns] = (cur_len << 2) | cur_op;
      }
      cur_op = op;
      cur_len = 1;
    }
  }
  if (cur_len) {
    ++nruns;
    if (slot_end) slot_end[-(int64_t)nruns] = (cur_len << 2) | cur_op;
  }
  if (a.cigar_nruns) a.cigar_nruns[p] = nruns;
  if (a.target_begin) {
    if (MODE == SA_LOCAL)
      a.target_begin[p] = i;  // row where the traceback stopped (src/alignment.cpp:244-246)
    else                      // index 1 of the CIGAR text is 'D' only for a leading D-run of 1..9
      a.target_begin[p] = (cur_len && cur_op == SA_OP_D && cur_len <= 9) ? 3u : 1u;
  }
}

}  // namespace sa

#endif
