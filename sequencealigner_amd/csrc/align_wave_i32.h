// Intra-pair anti-diagonal ("wavefront") DP fill for long pairs, 32-bit scores.
//
// A strip is 64*C query columns, C per lane, held in registers; it is swept down the target rows with a
// skew of one row per lane: at step st lane L computes row i = st - L of its C columns, so the cells of
// one step lie on an anti-diagonal of C-column blocks.  The value a lane needs from its left neighbour
// (last column of the block to the left, same row) was produced by lane L-1 in the previous step and is
// passed with a one-lane DPP shift; the diagonal value is the one received the step before.  The
// right-most column of a strip is handed to the next strip through a boundary column in memory, read
// and written 64 rows at a time with coalesced accesses.
//
// Two ways to run the strips of a pair:
//   * sequential: one wavefront per pair walks its strips one after the other (many pairs per launch);
//   * cooperative (COOP): one wavefront per (pair, strip), all resident at once, each strip following
//     its left neighbour at a distance of about two boundary batches.  A boundary entry is one 64-bit
//     word {value, tag of the writing strip}, written and read with single device-scope accesses: the
//     next strip polls the 64 entries of its batch until all carry the expected tag, so the data is its
//     own ready flag and no fence is needed.  Strip numbers are handed out by a ticket counter in start
//     order, so a waiting wave only ever waits for waves that already run; every wait is bounded and
//     raises an abort flag.
//
// Same recurrence, tie-break and clamping as the reference (src/alignment.cpp:63-85, 176-201, 281-303):
// m = max(diag + sub, up + gap, left + gap), Diag if m == diag + sub, else Left if m == left + gap, else
// Up; Local clamps m <= 0 to {0, None}.  Registers hold H + gap, so "up" and "left" candidates need no
// add and the diagonal candidate adds (sub - gap).  Per cell two flags are stored, "diag candidate != m"
// and "left candidate != m" (the sign bits of candidate - m, shifted into the direction word); None is
// not stored: a Local cell is None exactly when its score is 0, which the traceback knows.
#ifndef SA_ALIGN_WAVE_I32_H
#define SA_ALIGN_WAVE_I32_H

#include "align_common.h"

#include <type_traits>

namespace sa {

// Which byte values occur in a wavefront's sequences?  `seen` is a 64-dword LDS array of the wavefront.
// alphabet_clear, then alphabet_mark for every sequence (each lane marks the bytes of its share: aligned
// 8-byte words - the blobs are 8-byte aligned and readable up to 8 bytes past their end), then alphabet_number
// turns seen[] into the byte -> code table (values in increasing order from 0) and returns how many occur.
// With at most four, a cell's substitution score is one v_perm_b32: the column keeps the selector "code of
// its query symbol", the row a word with (match - mismatch) in the byte of its target symbol.
__device__ __forceinline__ void alphabet_clear(uint32_t* seen, uint32_t lane) {
  seen[lane] = 0;
  __syncthreads();
}
__device__ __forceinline__ void alphabet_mark(uint32_t* seen, uint32_t lane, const uint8_t* p, uint32_t len) {
  uint8_t* mark = reinterpret_cast<uint8_t*>(seen);
  const uint64_t b0 = (uint64_t)p, b1 = b0 + len;
  for (uint64_t w = (b0 & ~7ull) + 8ull * lane; w < b1; w += 512) {
    const uint64_t word = *reinterpret_cast<const uint64_t*>(w);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (w + j >= b0 && w + j < b1) mark[(uint32_t)(word >> (8 * j)) & 0xFFu] = 1;
  }
}
__device__ __forceinline__ uint32_t alphabet_number(uint32_t* seen, uint32_t lane) {
  __syncthreads();
  const uint32_t v = seen[lane];  // byte values 4 lane .. 4 lane + 3
  uint32_t total = 0, rank = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t m = __ballot((v >> (8 * j)) & 1u);
    total += (uint32_t)__builtin_popcountll(m);
    rank += (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));  // values in the lanes below
  }
  uint32_t cv = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    cv |= (rank & 0xFFu) << (8 * j);
    rank += (v >> (8 * j)) & 1u;
  }
  __syncthreads();
  seen[lane] = cv;
  __syncthreads();
  return total;
}

template <int C>
struct WaveWord;
template <>
struct WaveWord<16> {
  using type = uint32_t;
};
template <>
struct WaveWord<8> {
  using type = uint16_t;
};
template <>
struct WaveWord<4> {
  using type = uint8_t;
};

// Direction storage: one word of 2*C bits per (strip, step, lane),
//   word[(s * nsteps + st) * 64 + L],  st = i + L,  cell k = (r - 1) % C:
//   bit 2C-1-2k = "diag candidate != m", bit 2C-2-2k = "left candidate != m",
// so each step's store is one contiguous line.  nsteps = max_t + 64.
template <int C>
struct WaveI32Layout {
  using word_t = typename WaveWord<C>::type;
  static constexpr uint32_t kStrip = 64 * C;
  __device__ static __forceinline__ uint32_t decode(uint32_t word, uint32_t k) {
    const uint32_t two = (word >> (2 * C - 2 - 2 * k)) & 3u;  // (not diag) << 1 | (not left)
    return !(two & 2u) ? DIR_DIAG : !(two & 1u) ? DIR_LEFT : DIR_UP;
  }
};

struct WaveCoop {
  uint32_t* ticket;    // one counter per launch
  int* abort_flag;     // set when a bounded wait ran out
  int32_t* cand;       // [pair][strip][3]: Local candidates (score, row, column)
  uint32_t spin_limit; // polls a wait may take before it gives up (kSpinLimit; the option "spin_limit" lowers it)
};

constexpr uint32_t kSpinLimit = 1u << 24;

// f(integral_constant<0>) ... f(integral_constant<N-1>): indices that are constants from the start, so that a
// small array indexed with them stays in registers (a loop the optimiser unrolls later left the traceback's
// prefetched tile in scratch memory, and with it a wait for the prefetch at the place it was issued)
template <uint32_t N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<uint32_t, N - 1>{});
  }
}

__device__ __forceinline__ int dpp_from_lane_below(int v) {
  // wave_shr:1 - lane L receives lane L-1's value, lane 0 keeps `old` (0)
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ int max3i(int a, int b, int c) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// acc << 1 | sign bit of diff
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t acc, int diff) {
  return __builtin_amdgcn_alignbit(acc, (uint32_t)diff, 31);
}

// Local requires gap <= 0 and mismatch <= 0 here (checked on the host): cells of the padding columns
// beyond the query then never exceed an earlier real cell, so they need no masking.
template <int MODE, int C, bool COOP>
__global__ __launch_bounds__(64) void fill_wave_i32(AlignArgs a, WaveCoop co, uint32_t tables) {
  using word_t = typename WaveWord<C>::type;
  constexpr uint32_t STRIP = 64 * C;
  const uint32_t lane = threadIdx.x;
  uint32_t pl, s_begin, s_end;  // this wave handles strips [s_begin, s_end) of pair pl
  if (COOP) {
    uint32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(co.ticket, 1u);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    pl = ticket / a.nstrips;
    s_begin = ticket % a.nstrips;
    s_end = s_begin + 1;
  } else {
    pl = blockIdx.x;
    s_begin = 0;
    s_end = a.nstrips;
  }
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint8_t* q = a.q + a.q_off[p];
  const uint8_t* t = a.t + a.t_off[p];
  const uint32_t Q = __builtin_amdgcn_readfirstlane((uint32_t)(a.q_off[p + 1] - a.q_off[p]));
  const uint32_t T = __builtin_amdgcn_readfirstlane((uint32_t)(a.t_off[p + 1] - a.t_off[p]));
  const int gap = a.gap;
  if (Q == 0 || T == 0) {  // no cells: the result is a border value
    if (s_begin == 0 && lane == 0) {
      a.score[p] = (MODE == SA_GLOBAL)  ? (T == 0 ? border_row0<MODE>((int)Q, gap) : border_col0<MODE>((int)T, gap))
                   : (MODE == SA_LOCAL) ? 0
                                        : border_row0<MODE>((int)Q, gap);
      a.end_i[pl] = (MODE == SA_GLOBAL) ? T : 0;
      a.end_r[pl] = (MODE == SA_LOCAL) ? 0 : Q;
    }
    return;
  }
  const uint32_t nstrips_p = (Q + STRIP - 1) / STRIP;
  if (s_end > nstrips_p) s_end = nstrips_p;
  if (s_begin >= s_end) return;
  // (sub - gap) for equal / unequal symbols, opaque to the optimiser so that the select is not re-split
  const int em = __builtin_amdgcn_readfirstlane(a.match - gap);
  const int ex = __builtin_amdgcn_readfirstlane(a.mismatch - gap);
  __shared__ uint32_t tring[256];  // per target position: the symbol, or the row's table word
  __shared__ uint32_t seen[64];    // byte values of the wavefront's sequences; then their codes
  // Substitution scores by table (see alphabet_number) when the wavefront's query columns and its target hold
  // at most four distinct byte values; `tables`: match - mismatch fits a byte.
  bool use_tab = false;
  const uint8_t* codes = reinterpret_cast<const uint8_t*>(seen);
  if (tables) {
    alphabet_clear(seen, lane);
    const uint32_t q0 = s_begin * STRIP, q1 = s_end * STRIP < Q ? s_end * STRIP : Q;
    alphabet_mark(seen, lane, q + q0, q1 - q0);
    alphabet_mark(seen, lane, t, T);
    use_tab = alphabet_number(seen, lane) <= 4;
  }
  const uint32_t delta = (uint32_t)(a.match - a.mismatch);
  word_t* dirs = reinterpret_cast<word_t*>(a.dirs + (uint64_t)pl * a.dirs_unit_stride);
  // boundary columns: 64-bit entries {H + gap, tag}; COOP uses two of them alternately
  uint64_t* bbase = reinterpret_cast<uint64_t*>(a.brow + (uint64_t)pl * a.brow_unit_stride);
  const uint64_t bhalf = COOP ? a.brow_unit_stride / 4 : 0;
  const uint32_t nsteps = a.max_t + 64;

  int best = 0;  // Local: best score; Semiglobal: best last-column score
  uint32_t bi = 0, br = 0;
  int final_score = 0;
  const uint32_t lastc = Q - 1;
  const uint32_t lane_q = (lastc % STRIP) / C, reg_q = lastc % C;  // where column Q lives

  auto sweep = [&](auto tab_tag) __attribute__((always_inline)) -> bool {
  constexpr bool TAB = decltype(tab_tag)::value;
  auto tsym = [&](uint32_t pos) -> uint32_t {  // what the ring holds for a target position
    if (pos >= T) return 0u;
    return TAB ? delta << (8 * (uint32_t)codes[t[pos]]) : (uint32_t)t[pos];
  };
  for (uint32_t s = s_begin; s < s_end; ++s) {
    const uint32_t c0 = s * STRIP + lane * C;  // this lane's columns are r = c0+1 .. c0+C
    const bool last_strip = (s + 1 == nstrips_p);
    const uint64_t* bin = bbase + ((s + 1) & 1) * bhalf;  // written by strip s-1 with tag s
    uint64_t* bout_p = bbase + (s & 1) * bhalf;           // written here with tag s+1
    int qc[C], Hg[C];  // query symbols (tables: v_perm selectors); H[i-1][column] + gap
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (TAB)  // bytes 1-3 and a column beyond the query: the constant zero
        qc[k] = (int)(0x0C0C0C00u | ((c0 + k < Q) ? (uint32_t)codes[q[c0 + k]] : 0x0Cu));
      else
        qc[k] = (c0 + k < Q) ? (int)q[c0 + k] : 512 + k;  // beyond the query: equals no byte
      Hg[k] = border_row0<MODE>((int)(c0 + 1 + k), gap) + gap;
    }
    int ldiag = border_row0<MODE>((int)c0, gap) + gap;  // H[i-1][c0] + gap for the lane's first row
    int lastcol = 0;                                    // H[i][c0 + C] + gap of the row just computed
    int bnext = 0, bout = 0;                            // 64 boundary rows in flight (one per lane)
    // Target symbols go through a ring in LDS (index = target position & 255), 64 new ones every 64
    // steps; a lane reads the symbol of its next row one step ahead.  Inside the step loop nothing is
    // left pending on the memory counter between the 64-step batches: a wait on it would also wait for
    // the step's own stores.
    if (lane == 0) tring[0] = tsym(0);
    int tc_next = (int)tsym(0);  // lane 0's first row; the other lanes read theirs from the ring
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads above and the query symbols
    const uint32_t steps = T + 63;
    word_t* drow = dirs + (uint64_t)s * nsteps * 64 + lane;
    // stores the boundary rows gathered in bout: rows row - slot .. row, tagged with this strip
    auto publish = [&](uint32_t row, uint32_t slot) __attribute__((always_inline)) {
      const uint32_t base = row - slot;
      if (base + lane <= row)
        __hip_atomic_store(&bout_p[base + lane], ((uint64_t)(s + 1) << 32) | (uint32_t)bout,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // One step.  FULL: every lane has a valid row (64 <= st <= T), nothing is predicated.
    auto step = [&](uint32_t st, auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      const uint32_t j = (st - 1) & 63;
      const int tc = tc_next;
      tc_next = (int)tring[(st - lane) & 255u];  // position of row st + 1 - lane
      const int from_left = dpp_from_lane_below(lastcol);
      const int first = __builtin_amdgcn_readlane(bnext, j);  // column c0 of lane 0: border or previous strip
      const int recv = lane == 0 ? first : from_left;
      const uint32_t i = st - lane;  // row of this lane in this step (valid when 1 <= i <= T)
      if (FULL || (st > lane && i <= T)) {
        // pass 1 (no dependencies between cells): diagonal candidates from the previous row's values
        // still held in Hg, already maximised with the "up" candidate (and with 0 for Local)
        int dc[C], mx[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int diag = k ? Hg[k - 1] : ldiag;
          if (TAB)
            dc[k] = diag + ex + (int)__builtin_amdgcn_perm(0u, (uint32_t)tc, (uint32_t)qc[k]);
          else
            dc[k] = diag + (tc == qc[k] ? em : ex);
          mx[k] = (MODE == SA_LOCAL) ? max3i(dc[k], Hg[k], 0) : max(dc[k], Hg[k]);
        }
        // pass 2: the dependent chain along the row (max with the left candidate, then + gap), with the
        // flag work of the cell before interleaved so that the chain's latencies are covered
        int left = recv, lin_prev = recv;
        uint32_t acc = 0, rowkey = 0;
        auto flags = [&](int k, int lin) __attribute__((always_inline)) {
          acc = shift_in_sign(acc, dc[k] - mx[k]);
          acc = shift_in_sign(acc, lin - mx[k]);
          if (MODE == SA_LOCAL) {  // leftmost maximum of the row: larger score, then smaller k
            const uint32_t key = ((uint32_t)mx[k] << 4) + (uint32_t)(15 - k);
            rowkey = key > rowkey ? key : rowkey;
          }
        };
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int lin = left;
          mx[k] = max(mx[k], left);
          left = mx[k] + gap;
          Hg[k] = left;
          if (k > 0) flags(k - 1, lin_prev);
          lin_prev = lin;
        }
        flags(C - 1, lin_prev);
        // written once, read once by the traceback: keep it out of the caches
        __builtin_nontemporal_store((word_t)acc, drow + (uint64_t)st * 64);
        ldiag = recv;
        lastcol = left;
        if (MODE == SA_LOCAL) {
          // first strict maximum in row-major order (src/alignment.cpp:203-207): inside a strip a lane
          // meets its rows in increasing order, so a later row must be strictly better; a later strip
          // (larger columns) revisits the rows from the top, and there an equal score in a smaller row wins
          const int rowbest = (int)(rowkey >> 4);
          const bool better = rowbest > best || (!COOP && rowbest == best && best > 0 && i < bi);
          best = better ? rowbest : best;
          bi = better ? i : bi;
          br = better ? c0 + 16 - (rowkey & 15u) : br;
        }
        if (last_strip && lane == lane_q) {
          // column Q's register, picked with masks (a dynamic index would move Hg out of registers)
          uint32_t pick = 0;
#pragma unroll
          for (int k = 0; k < C; ++k) pick |= (uint32_t)Hg[k] & ((uint32_t)k == reg_q ? ~0u : 0u);
          const int v = (int)pick - gap;
          if (MODE == SA_SEMIGLOBAL && v > best) {  // src/alignment.cpp:87-92, rows in increasing order
            best = v;
            bi = i;
          }
          if (MODE == SA_GLOBAL && i == T) final_score = v;
        }
      }
      // lane 63 finished row st - 63 of the strip's last column: collect 64 of them, store coalesced
      // (in a full batch the store falls on the batch's last step but one: done by the caller)
      if (!last_strip && (FULL || st >= 64)) {
        const int v63 = __builtin_amdgcn_readlane(lastcol, 63);
        const uint32_t row = st - 63;  // 1 ..
        const uint32_t slot = (row - 1) & 63;
        bout = lane == slot ? v63 : bout;
        if (!FULL && (slot == 63 || row == T)) publish(row, slot);
      }
    };
    for (uint32_t b0 = 1; b0 <= steps; b0 += 64) {  // batches of 64 steps
      {  // target positions b0 .. b0+63
        const uint32_t pos = b0 + lane;
        tring[pos & 255u] = tsym(pos);
      }
      if (s > 0) {  // rows b0 .. b0+63 of the previous strip's last column
        const uint32_t row = b0 + lane;
        const bool want = row <= T;
        uint64_t e = 0;
        for (uint32_t polls = 0;; ++polls) {
          // device-scope load: past this CU's L1 (the location is rewritten every other strip)
          if (want) e = __hip_atomic_load(&bin[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!COOP || __all(!want || (uint32_t)(e >> 32) == s)) break;
          __builtin_amdgcn_s_sleep(1);
          if (polls >= co.spin_limit ||
              ((polls & 1023u) == 1023u &&
               __hip_atomic_load(co.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(co.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
        }
        bnext = (int)(uint32_t)e;
      } else {  // column 0 of the matrix
        bnext = border_col0<MODE>((int)(b0 + lane), gap) + gap;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) here, so that no step has to wait on it
      if (b0 >= 64 && b0 + 63 <= T) {
        for (uint32_t x = 0; x < 63; ++x) step(b0 + x, std::true_type{});
        if (!last_strip) publish(b0 - 1, 63);  // rows b0-64 .. b0-1 are complete after step b0+62
        step(b0 + 63, std::true_type{});
      } else {
        const uint32_t e = b0 + 63 < steps ? b0 + 63 : steps;
        for (uint32_t st = b0; st <= e; ++st) step(st, std::false_type{});
      }
    }
  }
  return true;
  };
  if (!(use_tab ? sweep(std::true_type{}) : sweep(std::false_type{}))) return;  // gave up waiting

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
    if (COOP) {  // one candidate per strip; reduced by wave_local_reduce
      if (lane == 0) {
        int32_t* c = co.cand + ((uint64_t)pl * a.nstrips + s_begin) * 3;
        c[0] = best;
        c[1] = (int32_t)bi;
        c[2] = (int32_t)br;
      }
      return;
    }
  } else {
    if (s_end != nstrips_p) return;  // only the strip with column Q knows the result
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

// Cooperative Local: picks each pair's best cell among its strips' candidates: highest score, then
// smallest row, then smallest column (the first strict maximum in row-major order).
__global__ void wave_local_reduce(AlignArgs a, WaveCoop co) {
  const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]), T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  if (Q == 0 || T == 0) return;  // written by the fill
  int best = 0;
  uint32_t bi = 0, br = 0;
  for (uint32_t s = 0; s < a.nstrips; ++s) {
    const int32_t* c = co.cand + ((uint64_t)pl * a.nstrips + s) * 3;
    const int ob = c[0];
    const uint32_t oi = (uint32_t)c[1], orr = (uint32_t)c[2];
    if (ob > best || (ob == best && ob > 0 && (oi < bi || (oi == bi && orr < br)))) {
      best = ob;
      bi = oi;
      br = orr;
    }
  }
  a.score[p] = best;
  a.end_i[pl] = best ? bi : 0;
  a.end_r[pl] = best ? br : 0;
}

// How a wavefront kernel stores its direction words, as seen by the traceback.
template <int C_>
struct WaveFmtI32 {
  static constexpr int C = C_;
  using word_t = typename WaveWord<C_>::type;
  static constexpr uint32_t kTileSteps = 64;     // steps of one LDS tile
  static constexpr uint32_t kStepsPerWord = 1;   // steps whose flags share a word
  static constexpr uint32_t kTileLanes = (16 / sizeof(word_t)) > 8 ? (16 / sizeof(word_t)) : 8;  // lanes of a tile
  static constexpr uint32_t kPairs = 1;          // pairs that share a wavefront's words
  // rows of words per strip (one row = the 64 lanes' words of kStepsPerWord steps)
  __host__ __device__ static uint32_t rows(uint32_t max_t) { return max_t + 64; }
  __device__ static __forceinline__ uint32_t decode(word_t w, uint32_t k, uint32_t, uint32_t) {
    return WaveI32Layout<C_>::decode(w, k);
  }
};

// What a walk leaves behind: the number of runs written, the cell it stopped at, the run that ended the walk
// (the first of the CIGAR) and whether the runs outgrew the room they were given.
struct WalkEnd {
  uint32_t nruns, i, r, last_op, last_len;
  bool overflow;
  bool gave_up;  // stopped after max_iters iterations, nowhere in particular
};

// Traceback + run-length encoding for the wavefront layouts by one wavefront (same rules as
// traceback_kernel).  The direction words of kTileSteps consecutive steps of a strip are one contiguous
// block: the wave copies that block, the strip's query symbols and the target symbols of the rows it
// spans into LDS and walks inside it; every lane follows the same path, the runs leave one per lane.
// The walk starts at cell (i, r) and stops where the path ends or, with lo > 0, at the first cell of row lo
// (no move is taken from that cell).  Runs are written backwards from slot_end, at most `cap` of them.  A walk
// gives up after max_iters iterations (one iteration: a diagonal stretch plus a gap move, or up to 64 moves of
// a gap run).
template <int MODE, class Fmt>
__device__ __forceinline__ WalkEnd walk_wave(const AlignArgs& a, uint32_t pl, uint32_t i, uint32_t r, uint32_t lo,
                                             uint32_t* slot_end, uint32_t cap, uint32_t max_iters) {
  using word_t = typename Fmt::word_t;
  constexpr int C = Fmt::C;
  constexpr uint32_t STRIP = 64 * C;
  constexpr uint32_t TS = Fmt::kTileSteps;
  constexpr uint32_t SPW = Fmt::kStepsPerWord;          // steps per word; such formats count steps from 0:
  constexpr uint32_t U0 = SPW > 1 ? 1 : 0;              //   word row of step st = (st - U0) / SPW
  constexpr uint32_t ROWS = TS / SPW;                   // word rows of one LDS tile
  constexpr uint32_t WPV = 16 / sizeof(word_t);         // words per 16-byte vector
  constexpr uint32_t TW = Fmt::kTileLanes;              // lanes (column blocks) of one LDS tile
  constexpr uint32_t VPR = TW / WPV;                    // vectors per tile row
  constexpr uint32_t kVecPerLane = ROWS * VPR / 64;     // vectors a lane moves per tile
  static_assert(TW % WPV == 0 && (ROWS * VPR) % 64 == 0 && TS % SPW == 0 && TS % 64 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) word_t tile[ROWS * TW];
  __shared__ uint8_t qs[STRIP];
  __shared__ uint8_t ts[TS + 64];
  const uint32_t lane = threadIdx.x;
  const uint32_t p = a.pair_base + pl;
  const uint32_t half = pl % Fmt::kPairs;
  const uint8_t* q = a.q + a.q_off[p];
  const uint8_t* t = a.t + a.t_off[p];
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
  const uint32_t T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  const word_t* dirs =
      reinterpret_cast<const word_t*>(a.dirs + (uint64_t)(pl / Fmt::kPairs) * a.dirs_unit_stride);
  const uint32_t nrows = Fmt::rows(a.max_t);
  const bool same = (a.match == a.mismatch);
  bool overflow = false;
  uint32_t nruns = 0, cur_op = 4, cur_len = 0;
  int remaining = (MODE == SA_LOCAL) ? a.score[p] : 1;
  // Finished runs wait in `runbuf`, one per lane, and leave with one store per iteration: the walk is a chain
  // of scalar instructions on one wavefront, a store per run (address, lane mask) was a good part of it.
  uint32_t runbuf = 0, nbuf = 0;
  auto flush_runs = [&]() __attribute__((always_inline)) {
    if (nruns + nbuf > cap) overflow = true;
    if (slot_end && !overflow && lane < nbuf) slot_end[-(int64_t)(nruns + 1 + lane)] = runbuf;
    nruns += nbuf;
    nbuf = 0;
  };
  auto push_run = [&](uint32_t run) __attribute__((always_inline)) {
    if (nbuf == 64) flush_runs();
    runbuf = lane == nbuf ? run : runbuf;
    ++nbuf;
  };
  // a run of `len` moves of kind op, in path order: joins the run in progress or closes it
  auto emit = [&](uint32_t op, uint32_t len) __attribute__((always_inline)) {
    if (op == cur_op) {
      cur_len += len;
    } else {
      if (cur_len) push_run((cur_len << 2) | cur_op);
      cur_op = op;
      cur_len = len;
    }
  };
  // The tile holds steps [tile_lo, tile_lo + TS) x lanes [tile_lb, tile_lb + TW) of strip tile_s (steps
  // counted from U0): the path moves towards smaller steps and lanes, about one lane per C diagonal moves.
  uint32_t tile_s = ~0u, tile_lo = 0, tile_lb = 0;
  // (a native vector type: copies of the uint4 struct are memcpy calls between address spaces, which kept this
  // array in scratch memory and made the prefetch wait for its loads at once)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pre[kVecPerLane];              // the predicted next tile, on its way in
  uint32_t pre_s = ~0u, pre_lo = 0, pre_lb = 0;
  uint32_t ts_base = 0;                // ts[x] = t[ts_base + x]
  // lane base of a tile whose top lanes include L (vector-aligned, inside the 64 lanes)
  auto lane_base = [&](uint32_t top) -> uint32_t {
    const uint32_t lb = top + WPV >= TW ? ((top + WPV - TW) / WPV) * WPV : 0u;
    return lb > 64 - TW ? 64 - TW : lb;
  };
  auto tile_vec = [&](uint32_t s, uint32_t lo, uint32_t lb, uint32_t x) -> const u32x4* {
    const uint32_t v = x * 64 + lane, row = v / VPR, colv = v % VPR;
    return reinterpret_cast<const u32x4*>(dirs + ((uint64_t)s * nrows + lo / SPW + row) * 64 + lb) + colv;
  };
  bool gave_up = false;
  for (uint32_t iters = 0;; ++iters) {
    if (lo && i <= lo) break;
    if (iters >= max_iters) {
      gave_up = true;
      break;
    }
    if (MODE == SA_GLOBAL && i == 0 && r == 0) break;
    if (MODE == SA_SEMIGLOBAL && r == 0) break;
    uint32_t d;
    uint32_t len = 1;  // moves of kind d taken in this iteration
    if (i == 0 || r == 0) {
      d = border_dir<MODE>(i, r, a.gap);   // None, Left or Up
      len = (d == DIR_LEFT) ? r : i - lo;  // a border is followed to its end
    } else {
      if (MODE == SA_LOCAL && remaining == 0) break;  // an inner cell with score 0 is None
      const uint32_t c = r - 1;
      const uint32_t s = c / STRIP, L = (c % STRIP) / C;
      const uint32_t st = i + L - U0;
      if (s != tile_s || (st & ~(TS - 1)) != tile_lo || L < tile_lb || L >= tile_lb + TW) {
        __syncthreads();
        const uint32_t lo = st & ~(TS - 1);
        u32x4* dst = reinterpret_cast<u32x4*>(tile);
        if (s == pre_s && lo == pre_lo && L >= pre_lb && L < pre_lb + TW) {
          tile_lb = pre_lb;
          static_for<kVecPerLane>([&](auto x) { dst[x * 64 + lane] = pre[x]; });
        } else {
          tile_lb = lane_base(L);
          static_for<kVecPerLane>([&](auto x) { dst[x * 64 + lane] = *tile_vec(s, lo, tile_lb, x); });
        }
        if (s != tile_s) {
#pragma unroll
          for (uint32_t x = 0; x < C; ++x) {
            const uint32_t col = s * STRIP + x * 64 + lane;
            qs[x * 64 + lane] = col < Q ? q[col] : 0;
          }
          tile_s = s;
        }
        tile_lo = lo;
        // rows of the tile: i = st' + U0 - L' for st' in [lo, lo+TS), L' in [0, 63]
        ts_base = lo + U0 >= 64 ? lo + U0 - 64 : 0;  // index of row (lo + U0 - 63) is lo + U0 - 64
#pragma unroll
        for (uint32_t x = 0; x < (TS + 64) / 64; ++x) {
          const uint32_t idx = ts_base + x * 64 + lane;
          ts[x * 64 + lane] = idx < T ? t[idx] : 0;
        }
        // Fetch the tile the path is likely to enter next while walking this one.  A diagonal move takes
        // 1 + 1/C steps: the path leaves through lane 0 into the strip to the left when that comes before
        // the tile's first step, otherwise downwards, about TS / (C + 1) lanes further left.
        const uint32_t to_lane0 = (L - tile_lb) * (C + 1);  // steps until a diagonal path passes the tile's first lane
        if (tile_lb == 0 && s > 0 && to_lane0 + C < st - lo) {
          pre_s = s - 1;
          pre_lo = (st - L * (C + 1)) & ~(TS - 1);
          pre_lb = lane_base(63);
        } else if (lo >= TS) {
          pre_s = s;
          pre_lo = lo - TS;
          // lanes the path moves left before it leaves this tile downwards; one lane of slack upwards, since
          // Up moves use steps without changing the lane
          const uint32_t drift = (st - lo) / (C + 1);
          pre_lb = lane_base(L + 1 > drift ? (L + 1 - drift < 63 ? L + 1 - drift : 63) : 0);
        } else {
          pre_s = ~0u;
        }
        if (pre_s != ~0u) static_for<kVecPerLane>([&](auto x) { pre[x] = *tile_vec(pre_s, pre_lo, pre_lb, x); });
        __syncthreads();
      }
      // Lane x looks at the cell x steps further along the diagonal, (i - x, r - x): the path follows
      // the diagonal for as many steps as the leading lanes all find Diag inside this tile.
      const uint32_t ix = i - lane, rx = r - lane;
      bool ok = lane < i - lo && lane < r && rx - 1 >= s * STRIP;
      const uint32_t cx = ok ? rx - 1 : c;
      const uint32_t Lx = (cx % STRIP) / C, kx = cx % C, stx = (ok ? ix : i) + Lx - U0;
      ok = ok && stx >= tile_lo && Lx >= tile_lb;
      const uint32_t dx = ok ? Fmt::decode(tile[((stx - tile_lo) / SPW) * TW + (Lx - tile_lb)], kx, half, stx % SPW)
                             : (uint32_t)DIR_NONE;
      const bool eqx = ok && ts[ix - 1 - ts_base] == qs[cx % STRIP];
      const uint64_t eqm = __ballot(eqx);
      bool diag_ok = ok && dx == DIR_DIAG;
      if (MODE == SA_LOCAL) {  // the score of cell x, were all cells before it diagonal moves
        const int pe = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(eqm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)eqm, 0u));
        const int val = remaining - (a.match - a.mismatch) * pe - a.mismatch * (int)lane;
        diag_ok = diag_ok && val != 0;
      }
      const uint64_t dm = __ballot(diag_ok);
      const uint32_t n = ~dm ? (uint32_t)__builtin_ctzll(~dm) : 64u;  // leading diagonal steps
      if (n > 0) {
        const uint64_t in_run = n == 64 ? ~0ull : ((1ull << n) - 1);
        const uint64_t eqs = same ? in_run : (eqm & in_run);
        if (MODE == SA_LOCAL) {
          const int ne = __builtin_popcountll(eqm & in_run);
          remaining -= a.match * ne + a.mismatch * ((int)n - ne);
        }
        i -= n;
        r -= n;
        {  // maximal runs of equal bits, in path order: the first may continue the run in progress, the others
           // alternate between M and X and start where neighbouring bits differ
          const uint32_t first = (eqs & 1ull) ? SA_OP_M : SA_OP_X;
          if (first != cur_op) {
            if (cur_len) push_run((cur_len << 2) | cur_op);
            cur_op = first;
            cur_len = 0;
          }
          uint64_t turns = (eqs ^ (eqs >> 1)) & (in_run >> 1);  // bit x: cell x + 1 differs from cell x
          uint32_t prev = 0;
          while (turns) {
            const uint32_t at = (uint32_t)__builtin_ctzll(turns) + 1;
            turns &= turns - 1;
            push_run(((cur_len + at - prev) << 2) | cur_op);
            cur_op ^= SA_OP_M ^ SA_OP_X;
            cur_len = 0;
            prev = at;
          }
          cur_len += n - prev;
        }
        // Lane n looked at the cell the run stopped at: when that is a gap move (and, for Local, not a cell
        // with score 0, which is None whatever its flags say), take it in the same iteration - single gaps
        // between diagonal stretches are the common case, and this halves the number of iterations.
        const uint32_t dn = n < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)dx, (int)n) : (uint32_t)DIR_NONE;
        if ((dn == DIR_UP || dn == DIR_LEFT) && (MODE != SA_LOCAL || remaining != 0)) {
          const uint32_t op = dn == DIR_LEFT ? SA_OP_I : SA_OP_D;
          if (dn == DIR_LEFT)
            r -= 1;
          else
            i -= 1;
          if (MODE == SA_LOCAL) remaining -= a.gap;
          emit(op, 1);
        }
        flush_runs();
        continue;
      }
      d = __builtin_amdgcn_readfirstlane(dx);  // lane 0 holds the current cell
      if (d == DIR_UP) {
        // Lane x looks at the cell x rows further up in the same column, (i - x, r): one step earlier in
        // the same lane's words.  A long gap in the query (a chain that spans far more reference than the
        // read is long) is walked 64 rows per iteration instead of one.  (Local: the score grows on the
        // way up, gap < 0, so no cell of the run is a zero cell.)
        const bool okx = lane < i - lo && st >= tile_lo + lane;
        const uint32_t ux = okx ? Fmt::decode(tile[((st - lane - tile_lo) / SPW) * TW + (L - tile_lb)], c % C, half,
                                              (st - lane) % SPW)
                                : (uint32_t)DIR_NONE;
        const uint64_t um = __ballot(okx && ux == DIR_UP);
        len = ~um ? (uint32_t)__builtin_ctzll(~um) : 64u;  // >= 1: lane 0 is the current cell
      } else if (d == DIR_LEFT) {
        // the same along the row: lane x looks at (i, r - x), as far as the strip and the tile reach
        const bool inx = lane < r && c - lane >= s * STRIP && c >= lane;
        const uint32_t cl = inx ? c - lane : c;
        const uint32_t Ll = (cl % STRIP) / C, stl = i + Ll - U0;
        const bool okx = inx && Ll >= tile_lb && stl >= tile_lo;
        const uint32_t lx = okx ? Fmt::decode(tile[((stl - tile_lo) / SPW) * TW + (Ll - tile_lb)], cl % C, half,
                                              stl % SPW)
                                : (uint32_t)DIR_NONE;
        const uint64_t lm = __ballot(okx && lx == DIR_LEFT);
        len = ~lm ? (uint32_t)__builtin_ctzll(~lm) : 64u;
      }
    }
    if (d == DIR_NONE || d == DIR_DIAG) break;  // (an inner Diag cell is handled above: None at score 0)
    uint32_t op;
    if (d == DIR_LEFT) {
      op = SA_OP_I;
      r -= len;
    } else {
      op = SA_OP_D;
      i -= len;
    }
    if (MODE == SA_LOCAL) remaining -= a.gap * (int)len;
    emit(op, len);
    flush_runs();
  }
  if (cur_len) push_run((cur_len << 2) | cur_op);
  flush_runs();
  return WalkEnd{nruns, i, r, cur_op, cur_len, overflow, gave_up};
}

// The first character pair of the CIGAR text decides what Global / Semiglobal report as target_begin: index 1
// of the text is 'D' only for a leading D-run of 1..9 (src/alignment.cpp:244-246 reads the text).
__device__ __forceinline__ uint32_t target_begin_of_first_run(uint32_t op, uint32_t len) {
  return (len && op == SA_OP_D && len <= 9) ? 3u : 1u;
}

// One wavefront per pair walks the whole path.  a.tb_only (optional): only the pairs it marks.
template <int MODE, class Fmt>
__global__ __launch_bounds__(64) void traceback_wave(AlignArgs a) {
  const uint32_t pl = blockIdx.x;
  if (pl >= a.pairs_in_pass) return;
  if (a.tb_only && !a.tb_only[pl]) return;
  if (a.gave_up && *a.gave_up) return;
  const uint32_t p = a.pair_base + pl;
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
  const uint32_t T = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  // (the end cell lies inside the matrix whatever the fill left behind)
  const uint32_t i = (MODE == SA_GLOBAL) ? T : min(a.end_i[pl], T);
  const uint32_t r = (MODE == SA_LOCAL) ? min(a.end_r[pl], Q) : Q;
  uint32_t* slot_end = a.cigar_runs ? a.cigar_runs + (uint64_t)(p + 1) * a.cigar_stride : nullptr;
  const WalkEnd e = walk_wave<MODE, Fmt>(a, pl, i, r, 0, slot_end, ~0u, ~0u);
  if (threadIdx.x != 0) return;
  if (a.cigar_nruns) a.cigar_nruns[p] = e.nruns;
  if (a.target_begin) {
    if (MODE == SA_LOCAL)
      a.target_begin[p] = e.i;  // row where the traceback stopped (src/alignment.cpp:244-246)
    else
      a.target_begin[p] = target_begin_of_first_run(e.last_op, e.last_len);
  }
}

}  // namespace sa

#endif
