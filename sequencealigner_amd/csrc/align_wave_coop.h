// Cooperative wavefront DP fill for a FEW LONG pairs (one 100 kbp x 100 kbp pair is the extreme case):
// one wavefront per (pair, strip of 64*C query columns), all strips of the batch resident at once, every
// strip sweeping down the target rows behind its left neighbour (see align_wave_i32.h for the sweep and
// for the boundary protocol: 64-bit {value, tag} entries, written and polled with single device-scope
// accesses, strip numbers from a ticket counter).  Global and Semiglobal; Local keeps the older
// cooperative form of fill_wave_i32.
//
// A single pair is bound by its dependency chain, not by throughput: strip s+1 can start `lag` steps after
// strip s, so the fill takes (T + strips * lag) steps of one wavefront, however many SIMDs are idle.  What
// counts is therefore the number of instructions in ONE step and the lag, and this kernel is built around
// those two:
//   * narrow strips (C = 2 columns per lane while the strips of the batch still get a SIMD each): a step
//     is 8 or 9 instructions per cell plus a fixed part, and the fixed part is four instructions: the lane's
//     target symbol of the row comes out of an LDS ring of the last 128 rows (every row stored twice, 128 entries
//     apart, so the 16 rows of a batch are contiguous from any start and each read is one instruction with a
//     constant offset), lane 0's left value out of LDS too (all lanes read it, a one-lane DPP shift then gives
//     every other lane the value of the lane below), and every lane drops its last column's value into LDS,
//     where lanes 47..62 find lane 63's sixteen values after the batch.  Inputs are read one step ahead.  No
//     address arithmetic and no branch inside a batch of 16 steps;
//   * boundary rows are handed over in batches of 16 and fetched 32 at a time, one fetch ahead, by lanes
//     0..31 of the next strip: on gfx950 a wait for a load also waits for every store issued before it, so the
//     kernel waits once per 32 steps.  A strip follows its neighbour at about 63 + 48 steps plus the memory
//     latency;
//   * direction flags of 16 / C consecutive steps share one dword per lane (one store per 16 / C steps):
//       dword[(s * ngroups + (st - 1) / SPW) * 64 + L],  SPW = 16 / C,  step st = i + L,
//       step (st - 1) % SPW = x sits at bits [(SPW - 1 - x) * 2C, (SPW - x) * 2C), inside it cell k (column
//       c0 + 1 + k of the lane) at bit 2C-1-2k = "diag candidate != m" and 2C-2-2k = "left candidate != m".
#ifndef SA_ALIGN_WAVE_COOP_H
#define SA_ALIGN_WAVE_COOP_H

#include "align_wave_i32.h"

namespace sa {

constexpr uint32_t kCoopBatch = 16;  // steps per batch = rows per hand-over to the next strip
constexpr uint32_t kCoopFetch = 32;  // rows a strip fetches at a time (two batches): the wait for a fetch also
                                     // waits for every store issued before it, so it should come rarely and late
constexpr uint32_t kCoopRing = 128;  // rows of target symbols a strip keeps at hand: lane 63 is 63 rows behind
                                     // lane 0, which has up to two fetches before it

// steps a strip's direction words are laid out for: every step up to max_t + 63, whole batches, plus the
// one extra batch in which the last boundary rows are handed over
__host__ __device__ inline uint32_t coop_steps_padded(uint32_t max_t) {
  return (max_t + 63 + kCoopBatch + kCoopFetch - 1) / kCoopFetch * kCoopFetch;
}

template <int C_>
struct WaveFmtCoop {
  static constexpr int C = C_;
  using word_t = uint32_t;
  static constexpr uint32_t kStepsPerWord = 16 / C_;
  static constexpr uint32_t kTileSteps = 128;
  static constexpr uint32_t kTileLanes = 64;  // a tile spans the strip
  static constexpr uint32_t kPairs = 1;
  __host__ __device__ static uint32_t rows(uint32_t max_t) { return coop_steps_padded(max_t) / kStepsPerWord; }
  __device__ static __forceinline__ uint32_t decode(uint32_t w, uint32_t k, uint32_t, uint32_t x) {
    const uint32_t two = (w >> ((kStepsPerWord - 1 - x) * 2 * C_ + (2 * C_ - 2 - 2 * k))) & 3u;
    return !(two & 2u) ? DIR_DIAG : !(two & 1u) ? DIR_LEFT : DIR_UP;
  }
};

__device__ __forceinline__ int dpp_shr1(int old, int v) {  // lane L <- lane L-1, lane 0 keeps `old`
  return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int dpp_shl1(int old, int v) {  // lane L <- lane L+1, lane 63 keeps `old`
  return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xF, 0xF, false);
}

template <int MODE, int C>
__global__ __launch_bounds__(64) void fill_wave_coop2(AlignArgs a, WaveCoop co, uint32_t tables) {
  static_assert(MODE != SA_LOCAL, "Global and Semiglobal only");
  __shared__ uint32_t seen[64];  // byte values of the strip's query columns and the target; then their codes
  // Inputs and outputs of the steps travel through LDS, one access with a constant offset per step each:
  __shared__ int trows[2 * kCoopRing];        // target symbol (tables: row word) of the last kCoopRing rows, row i at
                                              // i % kCoopRing and again kCoopRing further on: 16 rows from any start
                                              // are contiguous
  __shared__ int bleft[2 * kCoopFetch];       // left boundary values of the current and the next kCoopFetch rows
  __shared__ int lastcols[kCoopBatch * 64];   // every lane's last-column value of the 16 steps of a batch
  constexpr uint32_t STRIP = 64 * C;
  constexpr uint32_t SPW = 16 / C;
  constexpr uint32_t B = kCoopBatch;
  const uint32_t lane = threadIdx.x;
  uint32_t ticket = 0;
  if (lane == 0) ticket = atomicAdd(co.ticket, 1u);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  const uint32_t pl = ticket / a.nstrips, s = ticket % a.nstrips;
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint8_t* q = a.q + a.q_off[p];
  const uint8_t* t = a.t + a.t_off[p];
  const uint32_t Q = __builtin_amdgcn_readfirstlane((uint32_t)(a.q_off[p + 1] - a.q_off[p]));
  const uint32_t T = __builtin_amdgcn_readfirstlane((uint32_t)(a.t_off[p + 1] - a.t_off[p]));
  const int gap = a.gap;
  if (Q == 0 || T == 0) {  // no cells: the result is a border value
    if (s == 0 && lane == 0) {
      a.score[p] = (MODE == SA_GLOBAL) ? (T == 0 ? border_row0<MODE>((int)Q, gap) : border_col0<MODE>((int)T, gap))
                                       : border_row0<MODE>((int)Q, gap);
      a.end_i[pl] = (MODE == SA_GLOBAL) ? T : 0;
      a.end_r[pl] = Q;
    }
    return;
  }
  const uint32_t nstrips_p = (Q + STRIP - 1) / STRIP;
  if (s >= nstrips_p) return;
  const bool last_strip = (s + 1 == nstrips_p);
  const int em = __builtin_amdgcn_readfirstlane(a.match - gap);
  const int ex = __builtin_amdgcn_readfirstlane(a.mismatch - gap);
  const uint32_t ngroups = coop_steps_padded(a.max_t) / SPW;
  uint32_t* dw = a.dirs + (uint64_t)pl * a.dirs_unit_stride + (uint64_t)s * ngroups * 64 + lane;
  uint64_t* bbase = reinterpret_cast<uint64_t*>(a.brow + (uint64_t)pl * a.brow_unit_stride);
  const uint64_t bhalf = a.brow_unit_stride / 4;
  const uint64_t* bin = bbase + ((s + 1) & 1) * bhalf;  // written by strip s-1 with tag s
  uint64_t* bout_p = bbase + (s & 1) * bhalf;           // written here with tag s+1
  const uint32_t c0 = s * STRIP + lane * C;             // this lane's columns are r = c0+1 .. c0+C
  const uint32_t lastc = Q - 1;
  const uint32_t lane_q = (lastc % STRIP) / C, reg_q = lastc % C;  // where column Q lives (last strip)

  // Substitution scores by table when the target and this strip's query columns hold at most four distinct
  // byte values (`tables`: match - mismatch fits a byte): a column keeps the v_perm selector of its query
  // symbol's code, a row the word with (match - mismatch) in the byte of its target symbol, so a cell's
  // diagonal candidate is perm + three-operand add instead of compare + select + add.
  bool use_tab = false;
  const uint8_t* codes = reinterpret_cast<const uint8_t*>(seen);
  if (tables) {
    alphabet_clear(seen, lane);
    const uint32_t q0 = s * STRIP;
    alphabet_mark(seen, lane, q + q0, Q - q0 < STRIP ? Q - q0 : STRIP);
    alphabet_mark(seen, lane, t, T);
    use_tab = alphabet_number(seen, lane) <= 4;
  }
  const uint32_t delta = (uint32_t)(a.match - a.mismatch);

  int best = 0, final_score = 0;  // Semiglobal: best last-column score; Global: H[T][Q]
  uint32_t bi = 0;
  auto sweep = [&](auto tab_tag) __attribute__((always_inline)) -> bool {
  constexpr bool TAB = decltype(tab_tag)::value;
  int qc[C], Hg[C];  // query symbols (tables: selectors); H[i-1][column] + gap
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
  uint32_t acc = 0;                                   // direction flags of the last SPW steps
  const uint32_t steps = T + 63;

  // Inputs of the 32 steps of two batches, fetched by lanes 0..31: the target symbol of row b0 + L and the entry
  // of the left boundary column for that row.
  auto want_row = [&](uint32_t b0) { return lane < kCoopFetch && b0 + lane <= T; };
  auto load_syms = [&](uint32_t b0) -> int { return want_row(b0) ? (int)t[b0 + lane - 1] : 256; };
  // tables: symbols -> row words, once the loads have arrived
  auto row_words = [&](int sym) -> int {
    if (!TAB) return sym;
    return sym < 256 ? (int)(delta << (8 * (uint32_t)codes[sym])) : 0;
  };
  auto load_entries = [&](uint32_t b0) -> uint64_t {
    if (s == 0)  // column 0 of the matrix, tagged as if strip -1 had written it
      return ((uint64_t)s << 32) | (uint32_t)(border_col0<MODE>((int)(b0 + lane), gap) + gap);
    return want_row(b0) ? __hip_atomic_load(&bin[b0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  };
  auto entries_ready = [&](uint32_t b0, uint64_t e) {
    return s == 0 || __all(!want_row(b0) || (uint32_t)(e >> 32) == s);
  };
  // waits (bounded) until the entries of batch b0 carry this strip's tag; false: give up
  auto poll_entries = [&](uint32_t b0, uint64_t& e) -> bool {
    for (uint32_t polls = 0; !entries_ready(b0, e); ++polls) {
      __builtin_amdgcn_s_sleep(1);
      if (polls >= co.spin_limit ||
          ((polls & 1023u) == 1023u && __hip_atomic_load(co.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(co.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
      e = load_entries(b0);
    }
    return true;
  };
  // Lanes hand values to other lanes of the wavefront through LDS (stage_rows -> batch, step -> publish).  The
  // hardware keeps a wavefront's LDS accesses in program order; this keeps the compiler from moving a load
  // above the stores of other lanes it reads (it could prove the addresses of ONE lane disjoint).  No
  // instruction is emitted for either builtin.
  auto lds_handoff = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // hands the fetched rows b0 .. b0 + 31 (b0 = 1 + 32 m) to the steps
  auto stage_rows = [&](uint32_t b0, int words, int entry) __attribute__((always_inline)) {
    if (lane < kCoopFetch) {
      const uint32_t at = (b0 + lane) % kCoopRing;
      trows[at] = words;
      trows[at + kCoopRing] = words;
      bleft[(((b0 - 1) / kCoopFetch) & 1u) * kCoopFetch + lane] = entry;
    }
    lds_handoff();
  };

  // One step at position X of its batch.  FULL: every lane has a valid row, nothing is predicated.
  // TRACK: Semiglobal's last strip, which follows the last column's maximum row by row.
  // tcv: the lane's target rows of this batch, blv: lane 0's left values of the batch.
  auto step = [&](uint32_t b0, auto x_tag, auto full_tag, auto track_tag, const int (&tcv)[B], const int (&blv)[B])
                  __attribute__((always_inline)) {
    constexpr uint32_t X = decltype(x_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool TRACK = decltype(track_tag)::value;
    const uint32_t st = b0 + X;
    const int tc = tcv[X];
    // lane 0 takes the boundary value, every other lane the value of the lane below
    const int recv = dpp_shr1(blv[X], lastcol);
    uint32_t bits = FULL ? acc : 0u;
    const uint32_t i = st - lane;  // row of this lane in this step (valid when 1 <= i <= T)
    if (FULL || (st > lane && i <= T)) {
      int dc[C], mx[C];
#pragma unroll
      for (int k = 0; k < C; ++k) {  // independent of each other: diagonal and "up" candidates
        const int diag = k ? Hg[k - 1] : ldiag;
        if (TAB)
          dc[k] = diag + ex + (int)__builtin_amdgcn_perm(0u, (uint32_t)tc, (uint32_t)qc[k]);
        else
          dc[k] = diag + (tc == qc[k] ? em : ex);
        mx[k] = max(dc[k], Hg[k]);
      }
      int left = recv;
#pragma unroll
      for (int k = 0; k < C; ++k) {  // the dependent chain along the row
        const int lin = left;
        mx[k] = max(mx[k], lin);
        left = mx[k] + gap;
        Hg[k] = left;
        bits = shift_in_sign(bits, dc[k] - mx[k]);
        bits = shift_in_sign(bits, lin - mx[k]);
      }
      ldiag = recv;
      lastcol = left;
      if (TRACK) {
        uint32_t pick = 0;  // column Q's register, picked with masks (a dynamic index would leave the registers)
#pragma unroll
        for (int k = 0; k < C; ++k) pick |= (uint32_t)Hg[k] & ((uint32_t)k == reg_q ? ~0u : 0u);
        const int v = (int)pick - gap;
        if (v > best) {  // src/alignment.cpp:87-92, rows in increasing order
          best = v;
          bi = i;
        }
      }
    }
    acc = FULL ? bits : ((acc << (2 * C)) | bits);
    lastcols[X * 64 + lane] = lastcol;  // lane 63's are the rows handed to the next strip
    if (X % SPW == SPW - 1)  // written once, read once by the traceback: keep it out of the caches
      __builtin_nontemporal_store(acc, dw + (uint64_t)((st - 1) / SPW) * 64);
  };
  // One batch of 16 steps starting at step b0; half: which half of the fetched rows it uses.
  auto batch = [&](uint32_t b0, uint32_t half, auto full_tag, auto track_tag) __attribute__((always_inline)) {
    const int* tp = trows + ((b0 - lane) % kCoopRing);
    const int* bp = bleft + ((((b0 - 1) / kCoopFetch) & 1u) * kCoopFetch + half * B);
    int tcv[B], blv[B];  // the batch's inputs, read in one go: no step waits for LDS
#pragma unroll
    for (uint32_t x = 0; x < B; ++x) {
      tcv[x] = tp[x];
      blv[x] = bp[x];
    }
#define SA_COOP_STEP(X) step(b0, std::integral_constant<uint32_t, X>{}, full_tag, track_tag, tcv, blv);
    SA_COOP_STEP(0) SA_COOP_STEP(1) SA_COOP_STEP(2) SA_COOP_STEP(3) SA_COOP_STEP(4) SA_COOP_STEP(5)
    SA_COOP_STEP(6) SA_COOP_STEP(7) SA_COOP_STEP(8) SA_COOP_STEP(9) SA_COOP_STEP(10) SA_COOP_STEP(11)
    SA_COOP_STEP(12) SA_COOP_STEP(13) SA_COOP_STEP(14) SA_COOP_STEP(15)
#undef SA_COOP_STEP
  };

  auto publish = [&](uint32_t b0) __attribute__((always_inline)) {
    // lane 63 finished the rows b0 - 63 .. b0 - 48 in this batch; lanes 47..62 pick them up and store them
    lds_handoff();
    if (!last_strip) {
      const uint32_t row = b0 + lane - 110;
      if (lane >= 47 && lane <= 62 && b0 + lane >= 111 && row <= T)
        __hip_atomic_store(&bout_p[row], ((uint64_t)(s + 1) << 32) | (uint32_t)lastcols[(lane - 47) * 64 + 63],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds_handoff();  // the next batch's steps overwrite what these lanes have just read
  };
  // One batch of 16 steps.
  auto run_batch = [&](uint32_t b0, uint32_t half) __attribute__((always_inline)) {
    const bool full = b0 >= 64 && b0 + B - 1 <= T;
    if (MODE == SA_SEMIGLOBAL && last_strip) {
      if (full)
        batch(b0, half, std::true_type{}, std::true_type{});
      else
        batch(b0, half, std::false_type{}, std::true_type{});
    } else {
      if (full)
        batch(b0, half, std::true_type{}, std::false_type{});
      else
        batch(b0, half, std::false_type{}, std::false_type{});
    }
  };

  uint64_t e_cur = load_entries(1);
  int t_cur = load_syms(1);
  if (!poll_entries(1, e_cur)) return false;
  stage_rows(1, row_words(t_cur), (int)(uint32_t)e_cur);
  // up to one batch beyond the last step, in which the last rows of the boundary are handed over
#ifdef SA_COOP_TIMERS  // a build for tools/coopbench.py: where a strip's time goes (100 MHz ticks)
  const uint64_t tm_start = wall_clock64();
  uint64_t tm_batches = 0, tm_wait = 0;
#define SA_TM(var, code)                     \
  {                                          \
    const uint64_t tm0 = wall_clock64();     \
    code;                                    \
    var += wall_clock64() - tm0;             \
  }
#else
#define SA_TM(var, code) code;
#endif
  for (uint32_t b0 = 1; b0 <= steps + B; b0 += kCoopFetch) {
    // fetch the inputs of the next two batches now, look at them after this pair of batches
    uint64_t e_next = load_entries(b0 + kCoopFetch);
    const int t_next = load_syms(b0 + kCoopFetch);
    SA_TM(tm_batches, run_batch(b0, 0); publish(b0); run_batch(b0 + B, 1); publish(b0 + B))
    SA_TM(tm_wait, if (!poll_entries(b0 + kCoopFetch, e_next)) return false)
    stage_rows(b0 + kCoopFetch, row_words(t_next), (int)(uint32_t)e_next);
  }
#undef SA_TM
#ifdef SA_COOP_TIMERS
  if (lane == 0 && (s < 3 || s % 100 == 0 || last_strip))
    printf("strip %u: started %.1f us after the launch's first tick, ran %.1f us: batches %.1f us, waits %.1f us\n", s,
           (double)(tm_start % 100000000ull) / 100.0, (double)(wall_clock64() - tm_start) / 100.0,
           (double)tm_batches / 100.0, (double)tm_wait / 100.0);
#endif
  {  // Global: a lane's registers keep the values of its last row, H[T][column] + gap
    uint32_t pick = 0;
#pragma unroll
    for (int k = 0; k < C; ++k) pick |= (uint32_t)Hg[k] & ((uint32_t)k == reg_q ? ~0u : 0u);
    final_score = (int)pick - gap;
  }
  return true;
  };
  if (!(use_tab ? sweep(std::true_type{}) : sweep(std::false_type{}))) return;  // gave up waiting
  if (!last_strip) return;  // only the strip with column Q knows the result
  best = __builtin_amdgcn_readlane(best, lane_q);
  bi = (uint32_t)__builtin_amdgcn_readlane((int)bi, lane_q);
  final_score = __builtin_amdgcn_readlane(final_score, lane_q);
  if (lane != 0) return;
  int score;
  if (MODE == SA_GLOBAL) {
    score = final_score;
    bi = T;
  } else {
    score = bi ? best : border_row0<MODE>((int)Q, gap);
  }
  a.score[p] = score;
  a.end_i[pl] = bi;
  a.end_r[pl] = Q;
}

}  // namespace sa

#endif
