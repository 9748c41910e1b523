// Inter-pair batched DP fill, one pair per lane, 32-bit scores: the general kernel (any symbols, any
// scoring values that fit int32, ragged lengths).  The register strip runs along the TARGET (K rows per
// strip); the lane's loop runs along the QUERY.  Recurrence and tie-break as in the reference
// (src/alignment.cpp:63-85, 176-201, 281-303): m = max(diag+sub, up+gap, left+gap), direction
// Diag if m == diag+sub, else Left if m == left+gap, else Up; Local additionally clamps m <= 0 to
// {0, None} (src/alignment.cpp:190-192).
#ifndef SA_ALIGN_LANE_I32_H
#define SA_ALIGN_LANE_I32_H

#include "align_common.h"

namespace sa {

template <int MODE, int K>
__global__ __launch_bounds__(256) void fill_lane_i32(AlignArgs a) {
  static_assert(K % 16 == 0, "K must be a multiple of 16");
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = tid >> 6;
  const bool active = tid < a.pairs_in_pass;
  const uint32_t p = a.pair_base + (active ? tid : 0);
  uint32_t Q = 0, T = 0;
  const uint8_t* q = a.q;
  const uint8_t* t = a.t;
  if (active) {
    q += a.q_off[p];
    t += a.t_off[p];
    Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
    T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  }
  const uint32_t maxQ = wave_max_u32(Q), maxT = wave_max_u32(T);
  const int match = a.match, mismatch = a.mismatch, gap = a.gap;
  uint32_t* dirs = a.dirs + (uint64_t)wave * a.dirs_unit_stride;
  int32_t* brow = a.brow + (uint64_t)wave * a.brow_unit_stride;

  int best = 0;
  uint32_t bi = 0, br = 0;
  // Global: H[T][Q]; the value for an empty query or target comes straight from the border.
  int final_score = (T == 0) ? border_row0<MODE>((int)Q, gap) : border_col0<MODE>((int)T, gap);

  const uint32_t nstrips = (maxT + K - 1) / K;
  for (uint32_t s = 0; s < nstrips; ++s) {
    const uint32_t i0 = s * K;
    int tc[K], Hc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t idx = i0 + k;
      tc[k] = idx < T ? (int)t[idx] : 256;  // 256: equals no byte and not the query filler either
      Hc[k] = border_col0<MODE>((int)(i0 + 1 + k), gap);
    }
    int top_prev = (i0 == 0) ? 0 : border_col0<MODE>((int)i0, gap);  // H[i0][0]
    for (uint32_t r = 1; r <= maxQ; ++r) {
      const int qc = r <= Q ? (int)q[r - 1] : 512;
      // (read past this CU's L1: the row was read and then overwritten while the previous strip ran)
      const int top = (s == 0) ? border_row0<MODE>((int)r, gap)
                               : __hip_atomic_load(&brow[(uint64_t)r * 64 + lane], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
      int diag = top_prev, up = top;
      uint32_t word = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int left = Hc[k];
        const int d = diag + (tc[k] == qc ? match : mismatch);
        const int u = up + gap;
        const int l = left + gap;
        int m = max(max(d, u), l);
        uint32_t dir = (m == d) ? DIR_DIAG : (m == l) ? DIR_LEFT : DIR_UP;
        if (MODE == SA_LOCAL) {
          if (m <= 0) {
            m = 0;
            dir = DIR_NONE;
          }
          // first strict maximum in row-major order (src/alignment.cpp:203-207): highest score, then
          // smallest row; equal rows are visited in increasing column order here.
          const uint32_t i = i0 + 1 + k;
          if (i <= T && r <= Q && (m > best || (m == best && i < bi))) {
            best = m;
            bi = i;
            br = r;
          }
        }
        diag = left;
        Hc[k] = m;
        up = m;
        word |= dir << (2 * (k & 15));
        if ((k & 15) == 15) {
          // written once, read once by the traceback: keep it out of the caches
          __builtin_nontemporal_store(word, &dirs[((uint64_t)(s * a.max_q + (r - 1)) * (K / 16) + (k >> 4)) * 64 + lane]);
          word = 0;
        }
      }
      if (s + 1 < nstrips) brow[(uint64_t)r * 64 + lane] = up;  // H[i0+K][r] for the next strip
      top_prev = top;
      if (r == Q) {
        if (MODE == SA_GLOBAL) {
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (i0 + 1 + k == T) final_score = Hc[k];
        } else if (MODE == SA_SEMIGLOBAL) {
          // first strict maximum > 0 of the last column, rows in increasing order
          // (src/alignment.cpp:87-92)
#pragma unroll
          for (int k = 0; k < K; ++k)
            if (i0 + 1 + k <= T && Hc[k] > best) {
              best = Hc[k];
              bi = i0 + 1 + k;
            }
        }
      }
    }
  }
  if (!active) return;
  int score;
  if (MODE == SA_GLOBAL) {
    score = final_score;
    bi = T;
    br = Q;
  } else if (MODE == SA_LOCAL) {
    score = best;
  } else {
    score = bi ? best : border_row0<MODE>((int)Q, gap);  // no positive cell: row 0 (defined here)
    br = Q;
  }
  a.score[p] = score;
  a.end_i[tid] = bi;
  a.end_r[tid] = br;
}

}  // namespace sa

#endif
