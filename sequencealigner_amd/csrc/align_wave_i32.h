// Intra-pair anti-diagonal ("wavefront") DP fill for long pairs: ONE pair per wavefront, 32-bit scores.
// The 64 lanes split a strip of 64*16 = 1024 query columns, 16 columns per lane, held in registers; the
// strip is swept down the target rows with a skew of one row per lane: at step t lane L computes row
// i = t - L of its 16 columns, so that the cells computed in one step lie on an anti-diagonal of
// 16-column blocks.  The value a lane needs from its left neighbour (last column of the block to the
// left, same row) was produced by lane L-1 in the previous step and is passed with a one-lane DPP shift;
// the diagonal value is the one received the step before.  The right-most column of a strip is handed
// to the next strip through a per-pair boundary column in memory, read and written 64 rows at a time
// with coalesced accesses (lane 0 picks its row's value with v_readlane, lane 63's results are gathered
// into one register, one row per lane).
// Same recurrence, tie-break and clamping as the reference (src/alignment.cpp:63-85, 176-201, 281-303).
#ifndef SA_ALIGN_WAVE_I32_H
#define SA_ALIGN_WAVE_I32_H

#include "align_common.h"

namespace sa {

constexpr int kWaveCols = 16;                  // query columns per lane
constexpr int kWaveStrip = 64 * kWaveCols;     // query columns per strip

// Direction storage of the wavefront kernel: one dword per (strip, step, lane) = 16 cells x 2 bits,
//   dword[(s * nsteps + t) * 64 + L],  t = i + L,  field 2 * ((r - 1) % 16)
// so each step's store is one contiguous 256-byte line.  nsteps = max_t + 64.
struct WaveI32Layout {
  static constexpr bool kImplicitNone = false;
  const uint32_t* dirs;
  uint64_t pair_stride;
  uint32_t nsteps;
  __device__ __forceinline__ uint32_t get(uint32_t pair_local, uint32_t i, uint32_t r) const {
    const uint32_t c = r - 1;
    const uint32_t s = c / kWaveStrip, L = (c % kWaveStrip) / kWaveCols, k = c % kWaveCols;
    const uint64_t w = ((uint64_t)s * nsteps + (i + L)) * 64 + L;
    return (dirs[pair_local * pair_stride + w] >> (2 * k)) & 3u;
  }
};

__device__ __forceinline__ int dpp_from_lane_below(int v) {
  // wave_shr:1 - lane L receives lane L-1's value, lane 0 keeps `old` (0)
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
}

template <int MODE>
__global__ __launch_bounds__(256) void fill_wave_i32(AlignArgs a) {
  constexpr int C = kWaveCols;
  const uint32_t lane = threadIdx.x & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep them in scalar registers
  const uint32_t pl = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint8_t* q = a.q + a.q_off[p];
  const uint8_t* t = a.t + a.t_off[p];
  const uint32_t Q = __builtin_amdgcn_readfirstlane((uint32_t)(a.q_off[p + 1] - a.q_off[p]));
  const uint32_t T = __builtin_amdgcn_readfirstlane((uint32_t)(a.t_off[p + 1] - a.t_off[p]));
  const int match = a.match, mismatch = a.mismatch, gap = a.gap;
  uint32_t* dirs = a.dirs + (uint64_t)pl * a.dirs_unit_stride;
  int32_t* bcol = a.brow + (uint64_t)pl * a.brow_unit_stride;  // H[i][strip's last column], i = 0..T
  const uint32_t nsteps = a.max_t + 64;
  const uint32_t nstrips = (Q + kWaveStrip - 1) / kWaveStrip;

  // lane-local best cell (Local) / last-column maximum (Semiglobal) / final score (Global)
  int best = 0;
  uint32_t bi = 0, br = 0;
  int final_score = (T == 0) ? border_row0<MODE>((int)Q, gap) : border_col0<MODE>((int)T, gap);
  const uint32_t lastc = Q ? Q - 1 : 0;
  const uint32_t lane_q = (lastc % kWaveStrip) / C, reg_q = lastc % C;  // where column Q lives

  for (uint32_t s = 0; s < nstrips; ++s) {
    const uint32_t c0 = s * kWaveStrip + lane * C;  // this lane's columns are r = c0+1 .. c0+C
    const bool last_strip = (s + 1 == nstrips);
    int qc[C], Hrow[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      qc[k] = (c0 + k < Q) ? (int)q[c0 + k] : 512 + k;  // beyond the query: equals no byte
      Hrow[k] = border_row0<MODE>((int)(c0 + 1 + k), gap);
    }
    int ldiag = border_row0<MODE>((int)c0, gap);  // H[i-1][c0] for the lane's first row (H[0][0] = 0)
    int lastcol = 0;                              // H[i][c0 + C] of the row just computed
    int bnext = 0, bout = 0;                      // 64 boundary rows in flight (one per lane)
    const uint32_t steps = T + 63;
    for (uint32_t st = 1; st <= steps; ++st) {
      const uint32_t j = (st - 1) & 63;
      if (s > 0 && j == 0) {  // rows st .. st+63 of the previous strip's last column
        const uint32_t row = st + lane;
        // read past this CU's L1: the same rows were read and then overwritten during the previous strip
        bnext = row <= T ? __hip_atomic_load(&bcol[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      }
      int recv = dpp_from_lane_below(lastcol);
      if (lane == 0)
        recv = (s == 0) ? border_col0<MODE>((int)st, gap) : __builtin_amdgcn_readlane(bnext, j);
      const uint32_t i = st - lane;  // row of this lane in this step (valid when 1 <= i <= T)
      if (st > lane && i <= T) {
        const int tc = (int)t[i - 1];
        int diag = ldiag, left = recv;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int up = Hrow[k];
          const int d = diag + (tc == qc[k] ? match : mismatch);
          const int u = up + gap;
          const int l = left + gap;
          int m = max(max(d, u), l);
          uint32_t dir = (m == d) ? DIR_DIAG : (m == l) ? DIR_LEFT : DIR_UP;
          if (MODE == SA_LOCAL) {
            if (m <= 0) {
              m = 0;
              dir = DIR_NONE;
            }
            // first strict maximum in row-major order (src/alignment.cpp:203-207): highest score,
            // then smallest row, then smallest column; a lane meets equal rows in column order.
            const uint32_t r = c0 + 1 + k;
            if (r <= Q && (m > best || (m == best && i < bi))) {
              best = m;
              bi = i;
              br = r;
            }
          }
          diag = up;
          Hrow[k] = m;
          left = m;
          word |= dir << (2 * k);
        }
        dirs[((uint64_t)s * nsteps + st) * 64 + lane] = word;
        ldiag = recv;
        lastcol = left;
        if (last_strip && lane == lane_q) {
          int v = Hrow[0];
#pragma unroll
          for (int k = 1; k < C; ++k)
            if ((uint32_t)k == reg_q) v = Hrow[k];
          if (MODE == SA_SEMIGLOBAL && v > best) {  // src/alignment.cpp:87-92, rows in increasing order
            best = v;
            bi = i;
          }
          if (MODE == SA_GLOBAL && i == T) final_score = v;
        }
      }
      // lane 63 finished row st - 63 of the strip's last column: collect 64 of them, store coalesced
      if (!last_strip && st >= 64) {
        const int v63 = __builtin_amdgcn_readlane(lastcol, 63);
        const uint32_t row = st - 63;  // 1 ..
        const uint32_t slot = (row - 1) & 63;
        if (lane == slot) bout = v63;
        if (slot == 63 || row == T) {
          const uint32_t base = row - slot;  // first row held in bout
          if (base + lane <= row) bcol[base + lane] = bout;
        }
      }
    }
  }

  // combine the lanes' candidates
  if (MODE == SA_LOCAL) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int ob = __shfl_xor(best, off);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off), orr = (uint32_t)__shfl_xor((int)br, off);
      const bool take = ob > best || (ob == best && (oi < bi || (oi == bi && orr < br)));
      if (take) {
        best = ob;
        bi = oi;
        br = orr;
      }
    }
  } else if (Q > 0) {
    best = __builtin_amdgcn_readlane(best, lane_q);
    bi = (uint32_t)__builtin_amdgcn_readlane((int)bi, lane_q);
    final_score = __builtin_amdgcn_readlane(final_score, lane_q);
  }
  if (lane != 0) return;
  int score;
  if (MODE == SA_GLOBAL) {
    score = final_score;
    bi = T;
    br = Q;
  } else if (MODE == SA_LOCAL) {
    score = best;
    if (best == 0) bi = br = 0;
  } else {
    score = bi ? best : border_row0<MODE>((int)Q, gap);
    br = Q;
  }
  a.score[p] = score;
  a.end_i[pl] = bi;
  a.end_r[pl] = br;
}

}  // namespace sa

#endif
