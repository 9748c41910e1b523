// Wavefront DP fill (see align_wave_i32.h for the sweep) with packed 16-bit scores: TWO pairs per
// wavefront, pair A in the low halves and pair B in the high halves of every register.
// Number format as in align_lane_pk16.h: S(x) = (x + BIAS) << SH with bit 15 of each half set,
// so plain 32-bit add/sub act as two independent 16-bit operations and unsigned packed max orders the
// halves.  Per cell (both pairs): xor + packed min (symbols differ?), add, packed multiply-add (diagonal
// candidate), two packed max, two packed subtractions whose sign bits are the direction flags (one
// v_perm spreads the four signs into four bytes, one v_and_or drops them into the step's word), one
// subtract for H + gap: 11 instructions for two cells.  A row's cells are computed in two passes, the
// independent part first, so that only max + subtract remain on the dependent chain.
//
// The two pairs may have different lengths: the sweep covers the longer query and target, positions
// beyond a pair's own sequence hold padding symbols that match nothing, and every pair's result is
// taken at its own last column / row.  One wavefront walks the strips of its two pairs one after the
// other (the sequential mode of the 32-bit kernel; thousands of pairs per launch).
//
// Local keeps the scores times 16 (SH = 4; 0 otherwise): the four free bits of a cell's value take
// 15 - (column within the lane), so one packed max per cell finds the leftmost maximum of the lane's row
// segment for both pairs, and one strict packed compare per step keeps the first row that reached the
// lane's best (src/alignment.cpp:203-207: the first strict maximum in row-major order).  Every strip's
// candidate is merged into the lane's at the strip's end (equal score: the smaller row wins - a later
// strip has larger columns), the lanes' candidates at the end of the sweep.  Cells of the padding never
// win: each is strictly below a cell of the pair itself (a gap or mismatch away).
//
// Substitution scores by table (the `tables` argument allows it: mismatch - gap >= 0 and (match - mismatch)
// << SH within a byte): a wavefront first marks the byte values of its pairs' sequences in LDS; with at most
// four of them it numbers them 0..3, keeps for every column a v_perm selector {code of pair A's query
// symbol, zero, 4 + code of pair B's, zero} (a padding column selects the constant zero), and for every row
// two words with (match - mismatch) in the byte of the pair's target symbol.  One v_perm then gives both
// pairs' "symbols equal ? match - mismatch : 0" and one three-operand add forms the diagonal candidates:
// 9 instructions for two cells instead of 11.  Other pairs compare the symbols as described above.
//
// Global and Semiglobal keep every lane's scores RELATIVE to a base of the lane's own (two 32-bit numbers, one
// per pair): a stored half is S(x - base).  The values a lane holds at one time - 16 neighbouring columns of
// one row, the value handed over by the lane below, the candidates formed from them - lie within a few dozen
// times D = max(-gap, match - gap, match - mismatch) of each other, whatever the length of the sequences, since
// neighbouring cells of the matrix differ by at most D.  Every 64 steps a lane moves its base to the value of
// its first column (one packed subtraction per register), and a value from the lane below is brought to the
// lane's own base with one packed addition per step (the difference of the two bases, renewed with them).  So
// sums, maxima and the direction flags are those of the 32-bit kernel for pairs of any length; only the values
// that leave the registers - the boundary column between strips, the scores at the pairs' last column - are
// made absolute again (base + relative value) by the lane that holds them.  Local keeps absolute scores (its
// clamp at zero needs them) and with them its bound on the score.
//
// COOP: one wavefront per (two pairs, strip), strip numbers from a ticket counter; the boundary column goes
// through memory 16 rows at a time as 64-bit entries {tag of the writing strip, both pairs' values as 24-bit
// numbers}, written and polled with single device-scope accesses (the protocol of align_wave_i32.h).  For long
// pairs that cannot fill the GPU evenly with a wavefront each.
// A launch with more strips than the GPU holds wavefronts cuts every strip into segments of target rows, one
// wavefront per (segment, two pairs, strip), tickets in this order: a wavefront stops after its segment's last
// step, with its lanes in 64 different rows as they are, leaves its registers (the columns' values, the lanes'
// bases, the values collected for the next strip, the best last-column scores so far, the table of symbol
// codes) in memory behind the boundary columns and a flag after them; the wavefront of the next segment waits
// for the flag, takes the registers, fills the ring of target symbols with the 64 rows in flight and goes on
// with the next step - the steps and their results are those of the strip in one piece.  All strips of the
// launch then advance at the same pace, segment by segment, and end together; in one piece the strips that start
// last run on a GPU that empties while each waits for its left neighbour on a fuller SIMD (1024 pairs of 10 kbp
// x 11 kbp: 25.6 ms in one piece, 17.9 in segments).  The segments of strip s lie s * 128 steps earlier than
// those of strip 0: the rows a segment needs from the strip to its left are then all written by that strip's
// segment of the same number or an earlier one, so no wavefront waits for a ticket later than its own.
//
// Preconditions (checked on the host): gap <= 0, mismatch <= match, gap <= match; Global / Semiglobal:
// D <= 64 and every score within 24 bits; Local: gap < 0, mismatch < 0, every value within an 11-bit window.
#ifndef SA_ALIGN_WAVE_PK16_H
#define SA_ALIGN_WAVE_PK16_H

#include "align_lane_pk16.h"
#include "align_wave_i32.h"

namespace sa {

constexpr int kWavePkCols = 16;  // query columns per lane (the cooperative form also runs narrower strips)
constexpr uint32_t kWavePkCenter = 0x4000;  // Global / Semiglobal: a lane's values stay around S(base) = this
constexpr uint32_t kWaveStateSlots = 32;   // cooperative form: dwords per lane kept between two segments of a strip
constexpr uint32_t kWaveSegShift = 128;    // ... and the shift of the segments from one strip to the next, in steps
// host side: wavefronts of the cooperative form the GPU holds at once (three per SIMD); a launch with more strips
// than that runs them in segments of kWavePkSegRows steps (1024 pairs of 10 kbp x 11 kbp: 25.6 ms in one piece,
// 21.1 / 19.0 / 17.9 / 18.2 / 20.2 ms with 4096 / 2048 / 1024 / 768 / 512; with fewer strips than that, segments that
// wait for their predecessors take the places of ones that could run: 256 pairs 10.3 against 19 ms); strips of
// one pass at most
constexpr uint64_t kWavePkResident = 3072;
constexpr uint32_t kWavePkSegRows = 1024;
constexpr uint64_t kWavePkPassStrips = 1u << 18;
constexpr long kWavePkMaxStep = 64;         // ... for scorings with D up to this

// Direction storage: per (strip, step, lane) one uint2 - .x cells 0-7, .y cells 8-15 of the lane's
// columns; byte h = "left candidate != m" of pair h, byte 2+h = "diag candidate != m", bit = cell % 8.
//   word[(s * nsteps + st) * 64 + L],  st = i + L,  nsteps = max_t + 64.
template <int C_>
struct WaveFmtPk16 {
  static constexpr int C = C_;
  using word_t = uint2;
  static constexpr uint32_t kTileSteps = 64;
  static constexpr uint32_t kStepsPerWord = 1;
  static constexpr uint32_t kTileLanes = 8;
  static constexpr uint32_t kPairs = 2;
  __host__ __device__ static uint32_t rows(uint32_t max_t) { return max_t + 64; }
  __device__ static __forceinline__ uint32_t decode(uint2 w, uint32_t k, uint32_t half, uint32_t) {
    const uint32_t word = k < 8 ? w.x : w.y, b = k & 7;
    const uint32_t not_diag = (word >> (16 + 8 * half + b)) & 1u;
    const uint32_t not_left = (word >> (8 * half + b)) & 1u;
    return !not_diag ? DIR_DIAG : (!not_left ? DIR_LEFT : DIR_UP);
  }
};

__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// boundary entries of the relative form: tag << 48 | pair B's value << 24 | pair A's value (24-bit numbers)
__device__ __forceinline__ uint64_t wave_pk_entry(uint32_t tag, int va, int vb) {
  return ((uint64_t)tag << 48) | ((uint64_t)((uint32_t)vb & 0xFFFFFFu) << 24) | ((uint32_t)va & 0xFFFFFFu);
}
__device__ __forceinline__ int wave_pk_entry_a(uint64_t e) { return (int)((uint32_t)e << 8) >> 8; }
__device__ __forceinline__ int wave_pk_entry_b(uint64_t e) { return (int)((uint32_t)(e >> 24) << 8) >> 8; }

// The work of one wavefront: two pairs (COOP: one segment of one strip of theirs, `ticket` says which).
template <int MODE, int C, bool COOP>
__device__ __forceinline__ void wave_pk16_task(const AlignArgs& a, const WaveCoop& co, uint32_t bias, uint32_t tables,
                                               uint32_t ticket) {
  static_assert(C <= 16 && C % 2 == 0, "one uint2 of flags per step");
  static_assert(C + 13 < (int)kWaveStateSlots, "room for a wavefront's registers between two segments");
  static_assert(!(COOP && MODE == SA_LOCAL), "the cooperative form is Global / Semiglobal only");
  constexpr bool REL = MODE != SA_LOCAL;  // scores relative to the lane's bases
  constexpr int SH = (MODE == SA_LOCAL) ? 4 : 0;
  constexpr uint32_t STRIP = 64 * C;
  constexpr uint32_t NB = COOP ? 16 : 64;  // steps per batch = rows fetched at a time from the strip before (32: no gain)
  // per target position (index = position & 255, and again 256 further on, so that 16 positions from any start
  // are contiguous): both pairs' symbols in .x, or the pairs' table words
  __shared__ uint2 tring[512];
  __shared__ uint32_t seen[64];  // byte values that occur in the wavefront's sequences; then their codes
  const uint32_t lane = threadIdx.x;
  uint32_t unit = ticket, s_mine = 0, seg = 0;
  bool results_here = true;  // COOP: the wavefront ran its strip's last segment
  if (COOP) {
    // segment by segment over all strips of the launch, a unit's strips next to each other
    const uint32_t per_seg = ((a.pairs_in_pass + 1) / 2) * a.nstrips;
    seg = ticket / per_seg;
    unit = (ticket % per_seg) / a.nstrips;
    s_mine = ticket % a.nstrips;
  }
  const uint32_t plA = 2 * unit, plB = plA + 1;
  if (plA >= a.pairs_in_pass) return;
  const bool hasB = plB < a.pairs_in_pass;
  const uint32_t pA = a.pair_base + plA, pB = a.pair_base + (hasB ? plB : plA);
  const uint8_t* qA = a.q + a.q_off[pA];
  const uint8_t* tA = a.t + a.t_off[pA];
  const uint8_t* qB = a.q + a.q_off[pB];
  const uint8_t* tB = a.t + a.t_off[pB];
  const uint32_t QA = __builtin_amdgcn_readfirstlane((uint32_t)(a.q_off[pA + 1] - a.q_off[pA]));
  const uint32_t TA = __builtin_amdgcn_readfirstlane((uint32_t)(a.t_off[pA + 1] - a.t_off[pA]));
  const uint32_t QB = hasB ? __builtin_amdgcn_readfirstlane((uint32_t)(a.q_off[pB + 1] - a.q_off[pB])) : 0;
  const uint32_t TB = hasB ? __builtin_amdgcn_readfirstlane((uint32_t)(a.t_off[pB + 1] - a.t_off[pB])) : 0;
  const int gap = a.gap;
  const uint32_t Q = QA > QB ? QA : QB, T = TA > TB ? TA : TB;  // the sweep covers both pairs

  // per pair: best last-column score (Semiglobal) / final score (Global), kept by the lane that owns
  // the pair's last column
  // (Local: every lane's best cell so far - score, row, column)
  int bestA = 0, bestB = 0, finalA = 0, finalB = 0;
  uint32_t biA = 0, biB = 0, brA = 0, brB = 0;

  // where each pair's last column lives (a pair without cells never matches a strip)
  const uint32_t lastA = QA ? QA - 1 : 0, lastB = QB ? QB - 1 : 0;
  const uint32_t stripA = (QA && TA) ? lastA / STRIP : ~0u, stripB = (QB && TB) ? lastB / STRIP : ~0u;
  if (COOP && s_mine != 0 && s_mine != stripA && s_mine != stripB && s_mine >= (Q + STRIP - 1) / STRIP) return;
  // COOP, strips in segments: segment `seg` of strip s is the steps seg_lo + 1 .. seg_lo + seg_rows that exist.  A
  // strip needs its left neighbour's row r after that one's step r + 63 + (a batch): with the segments of strip s
  // shifted by s * kWaveSegShift steps, a segment never waits for a later one, which keeps the order of the
  // tickets (segment, unit, strip) free of cycles.
  const long seg_lo = COOP ? (long)seg * a.seg_rows - (long)s_mine * (long)kWaveSegShift : 0;
  if (COOP && a.seg_rows) {
    if (seg != 0 && (Q == 0 || T == 0)) return;
    if (seg_lo + (long)a.seg_rows < 1 || seg_lo + 1 > (long)T + 63) return;  // nothing of the strip in this segment
  }

  if (Q != 0 && T != 0) {
    const uint32_t G = kGuard2;
    auto SG = [&](int v) -> uint32_t { return rep2((uint32_t)((int)bias + v) << SH) | G; };
    // the two pairs' values, each relative to its own base (REL)
    auto S2 = [&](int va, int vb) -> uint32_t {
      return (((uint32_t)((int)bias + va)) & 0x7FFFu) | ((((uint32_t)((int)bias + vb)) & 0x7FFFu) << 16) | G;
    };
    const uint32_t gsub2 = rep2((uint32_t)(-gap) << SH);                                  // |gap|
    const uint32_t mg2 = rep2((uint32_t)(a.match - gap) << SH);                           // match - gap >= 0
    const uint32_t negdelta2 = rep2(((uint32_t)(a.mismatch - a.match) << SH) & 0xFFFFu);  // mod 2^16
    const uint32_t floor2 = SG(0);  // Local: a cell's score is at least 0
    const uint32_t one2 = 0x00010001u;
    const uint32_t nstrips_p = (Q + STRIP - 1) / STRIP;
    uint2* dirs = reinterpret_cast<uint2*>(a.dirs + (uint64_t)unit * a.dirs_unit_stride);
    uint64_t* bbase = reinterpret_cast<uint64_t*>(a.brow + (uint64_t)unit * a.brow_unit_stride);
    const uint64_t bhalf = ((a.max_t + 1 + 63) / 64 + 1) * 64;  // COOP: two columns of entries, used in turn
    // COOP, strips in segments: behind the columns, per strip the registers of a wavefront between two segments
    // ([kWaveStateSlots][64] dwords; the last dword: number of the segment they are ready for)
    uint32_t* sbase = reinterpret_cast<uint32_t*>(bbase + 2 * bhalf);
    const uint32_t nsteps = a.max_t + 64;
    const uint32_t laneA = (lastA % STRIP) / C, regA = lastA % C;
    const uint32_t laneB = (lastB % STRIP) / C, regB = lastB % C;

    // Do the four sequences hold at most four distinct byte values?  (seen[] becomes the byte -> code table.)
    bool use_tab = false;
    const uint8_t* codes = reinterpret_cast<const uint8_t*>(seen);
    uint32_t* state = sbase + (uint64_t)s_mine * (kWaveStateSlots * 64);  // COOP
    const bool resumed = COOP && a.seg_rows && seg_lo > 0;  // a later segment: takes over the registers of the one before
    if (resumed) {
      // (bounded) until the wavefront of the segment before has left them
      uint32_t* flag = state + kWaveStateSlots * 64 - 1;
      for (uint32_t polls = 0; __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seg; ++polls) {
        __builtin_amdgcn_s_sleep(4);
        if (polls >= co.spin_limit ||
            ((polls & 1023u) == 1023u && __hip_atomic_load(co.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(co.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
      seen[lane] = __builtin_nontemporal_load(state + (C + 11) * 64 + lane);  // the table of codes, if one is in use
      use_tab = __builtin_nontemporal_load(state + (C + 12) * 64 + lane) != 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the table is read by every lane)
      __builtin_amdgcn_wave_barrier();
    } else if (tables) {
      alphabet_clear(seen, lane);
      alphabet_mark(seen, lane, qA, QA);
      alphabet_mark(seen, lane, tA, TA);
      if (hasB) {
        alphabet_mark(seen, lane, qB, QB);
        alphabet_mark(seen, lane, tB, TB);
      }
      use_tab = alphabet_number(seen, lane) <= 4;
    }
    const uint32_t delta = (uint32_t)(a.match - a.mismatch) << SH;       // tables: a byte
    const uint32_t exg2 = rep2((uint32_t)(a.mismatch - gap) << SH);      // tables: mismatch - gap >= 0

    auto sweep = [&](auto tab_tag) __attribute__((always_inline)) -> bool {
    constexpr bool TAB = decltype(tab_tag)::value;
    const uint32_t s_lo = COOP ? s_mine : 0u;
    const uint32_t s_hi = COOP ? (s_mine < nstrips_p ? s_mine + 1 : s_mine) : nstrips_p;
    for (uint32_t s = s_lo; s < s_hi; ++s) {
      const uint32_t c0 = s * STRIP + lane * C;  // this lane's columns are r = c0+1 .. c0+C
      const bool last_strip = (s + 1 == nstrips_p);
      const uint64_t* bin = COOP ? bbase + ((s + 1) & 1u) * bhalf : bbase;  // COOP: written by strip s - 1 with tag s
      uint64_t* bcol = COOP ? bbase + (s & 1u) * bhalf : bbase;             //       written here with tag s + 1
      // REL: the lane's bases, at first the border value left of its first column
      int baseA = REL ? border_row0<MODE>((int)c0, gap) : 0, baseB = baseA;
      uint32_t adj2 = 0;                         // REL: (base of the lane below) - (own base), both pairs
      int b63A = 0, b63B = 0;                    // REL: lane 63's bases
      // query symbols (A | B << 16) or the columns' table selectors; S(H[i-1][column] + gap)
      uint32_t qc[C], Hg[C];
#pragma unroll
      for (int k = 0; k < C; ++k) {
        if (TAB) {  // bytes 0 / 2: which byte of pair A's / pair B's row word (0x0C: the constant zero)
          const uint32_t ca = (c0 + k < QA) ? (uint32_t)codes[qA[c0 + k]] : 0x0Cu;
          const uint32_t cb = (c0 + k < QB) ? 4u + (uint32_t)codes[qB[c0 + k]] : 0x0Cu;
          qc[k] = ca | (cb << 16) | 0x0C000C00u;
        } else {
          const uint32_t sa = (c0 + k < QA) ? (uint32_t)qA[c0 + k] : kQueryPad;
          const uint32_t sb = (c0 + k < QB) ? (uint32_t)qB[c0 + k] : kQueryPad;
          qc[k] = sa | (sb << 16);
        }
        Hg[k] = SG(border_row0<MODE>((int)(c0 + 1 + k), gap) + gap - baseA);
      }
      uint32_t ldiag = SG(border_row0<MODE>((int)c0, gap) + gap - baseA);  // S(H[i-1][c0] + gap), lane's first row
      uint32_t lastcol = 0, bnext = 0, bout = 0;
      // Local: this strip's best cell of the lane, both pairs - S(score) | 15 - column in the low bits of
      // each half (the value of "no cell yet" is S(0)), and the rows they were found in
      uint32_t skey = floor2, siA = 0, siB = 0;
      // both pairs' target symbols at a position, as loaded (a position beyond a pair's target: the padding
      // symbol), and as the steps want them; tables: per pair (match - mismatch) in the symbol's byte
      auto traw = [&](uint32_t pos) -> uint32_t {
        const uint32_t sa = pos < TA ? (uint32_t)tA[pos] : kTargetPad;
        const uint32_t sb = pos < TB ? (uint32_t)tB[pos] : kTargetPad;
        return sa | (sb << 16);
      };
      auto tconv = [&](uint32_t raw) -> uint2 {
        if (TAB) {
          const uint32_t sa = raw & 0xFFFFu, sb = raw >> 16;
          const uint32_t wa = sa < 256u ? delta << (8 * (uint32_t)codes[sa]) : 0u;
          const uint32_t wb = sb < 256u ? delta << (8 * (uint32_t)codes[sb]) : 0u;
          return make_uint2(wa, wb);
        }
        return make_uint2(raw, 0u);
      };
      // the entries of the previous strip's last column for rows b0 .. b0 + NB - 1
      auto load_entries = [&](uint32_t b0) -> uint64_t {
        const uint32_t row = b0 + lane;
        return (s > 0 && lane < NB && row <= T) ? __hip_atomic_load(&bin[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                : 0ull;
      };
      auto ring_put = [&](uint32_t pos, uint2 v) {
        tring[pos & 255u] = v;
        tring[(pos & 255u) + 256u] = v;
      };
      const uint32_t steps = T + 63;
      // the batches of this wavefront: all of the strip's, or those of its segment `seg`
      uint32_t b_first = 1, b_last = steps;
      if (COOP && a.seg_rows) {
        b_first = seg_lo < 0 ? 1u : (uint32_t)seg_lo + 1u;
        b_last = seg_lo + (long)a.seg_rows < (long)steps ? (uint32_t)(seg_lo + (long)a.seg_rows) : steps;
        results_here = b_last == steps;
      }
      if (resumed) {
        const uint32_t* in = state + lane;
#pragma unroll
        for (int k = 0; k < C; ++k) Hg[k] = __builtin_nontemporal_load(in + k * 64);
        ldiag = __builtin_nontemporal_load(in + (C + 0) * 64);
        lastcol = __builtin_nontemporal_load(in + (C + 1) * 64);
        bout = __builtin_nontemporal_load(in + (C + 2) * 64);
        baseA = (int)__builtin_nontemporal_load(in + (C + 3) * 64);
        baseB = (int)__builtin_nontemporal_load(in + (C + 4) * 64);
        bestA = (int)__builtin_nontemporal_load(in + (C + 5) * 64);
        biA = __builtin_nontemporal_load(in + (C + 6) * 64);
        bestB = (int)__builtin_nontemporal_load(in + (C + 7) * 64);
        biB = __builtin_nontemporal_load(in + (C + 8) * 64);
        finalA = (int)__builtin_nontemporal_load(in + (C + 9) * 64);
        finalB = (int)__builtin_nontemporal_load(in + (C + 10) * 64);
        ring_put(b_first - 64 + lane, tconv(traw(b_first - 64 + lane)));  // the rows the lanes are in
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (other lanes read them)
        __builtin_amdgcn_wave_barrier();
      } else if (lane == 0) {
        ring_put(0, tconv(traw(0)));
      }
      // a batch's inputs are fetched one batch ahead, so that no batch starts with a wait for memory
      uint32_t t_pre = lane < NB ? traw(b_first + lane) : 0u;
      uint64_t e_pre = load_entries(b_first);
      uint2* drow = dirs + (uint64_t)s * nsteps * 64 + lane;
      uint32_t pick_mask[C];  // wave-uniform: which half of which column holds a pair's last column
#pragma unroll
      for (int k = 0; k < C; ++k)
        pick_mask[k] = __builtin_amdgcn_readfirstlane(((uint32_t)k == regA ? 0x0000FFFFu : 0u) |
                                                      ((uint32_t)k == regB ? 0xFFFF0000u : 0u));
      // The cells of one step: row i = st - lane of the lane's columns, from the row's target symbols tc and the
      // value recv handed over from the left; the step's direction words go to dst.
      // (pick_tag: whether the step may be the one that holds a Global pair's result)
      auto cells = [&](uint2 tc, uint32_t recv, uint32_t i, uint32_t st, uint2* dst, auto pick_tag)
                       __attribute__((always_inline)) {
        constexpr bool PICK = decltype(pick_tag)::value;
        // pass 1 (no dependencies between cells): diagonal candidates from the previous row's values
        // still held in Hg, already maximised with the "up" candidate
        uint32_t dc[C], mx[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const uint32_t diag = k ? Hg[k - 1] : ldiag;
          if (TAB) {  // S(H + mismatch) + (symbols equal ? match - mismatch : 0), both pairs
            dc[k] = diag + exg2 + __builtin_amdgcn_perm(tc.y, tc.x, qc[k]);
          } else {
            const uint32_t ne = pk_min_u16(tc.x ^ qc[k], one2);  // 1 where the symbols differ
            dc[k] = pk_mad_u16(ne, negdelta2, diag + mg2);
          }
          mx[k] = pk_max_u16(dc[k], Hg[k]);
          if (MODE == SA_LOCAL) mx[k] = pk_max_u16(mx[k], floor2);
        }
        // pass 2: the dependent chain along the row (max with the left candidate, then + gap), with the
        // flag work of the cell before interleaved so that the chain's latencies are covered.  Flags:
        // candidate - max is negative (bit 15 of the half set) exactly when the candidate is NOT the
        // maximum; one v_perm turns the four sign bits into four 0x00/0xFF bytes.
        uint32_t left = recv, acc0 = 0, acc1 = 0, rowkey = 0;
        auto flags = [&](int k, uint32_t lin) __attribute__((always_inline)) {
          if (MODE == SA_LOCAL) rowkey = pk_max_u16(rowkey, mx[k] | rep2((uint32_t)(15 - k)));
          const uint32_t signs =
              __builtin_amdgcn_perm(pk_sub_u16(dc[k], mx[k]), pk_sub_u16(lin, mx[k]), 0x0B0A0908u);
          if (k < 8)
            acc0 = and_or(signs, 0x01010101u << (k & 7), acc0);
          else
            acc1 = and_or(signs, 0x01010101u << (k & 7), acc1);
        };
        uint32_t lin_prev = recv;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const uint32_t lin = left;
          mx[k] = pk_max_u16(mx[k], left);
          left = mx[k] - gsub2;  // S(H[i][column] + gap)
          Hg[k] = left;
          if (k > 0) flags(k - 1, lin_prev);
          lin_prev = lin;
        }
        flags(C - 1, lin_prev);
        // written once, read once by the traceback: keep it out of the caches
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u32x2{acc0, acc1}, reinterpret_cast<u32x2*>(dst));
        ldiag = recv;
        lastcol = left;
        if (MODE == SA_LOCAL) {
          // a row replaces the strip's candidate only with a strictly larger score: with the low bits of
          // the old key filled, an equal score never exceeds it
          const uint32_t old = skey | 0x000F000Fu;
          const uint32_t now = pk_max_u16(old, rowkey);
          const uint32_t changed = now ^ old;
          const bool upA = (changed & 0xFFFFu) != 0, upB = (changed >> 16) != 0;
          skey = (upA ? now & 0x0000FFFFu : skey & 0x0000FFFFu) | (upB ? now & 0xFFFF0000u : skey & 0xFFFF0000u);
          siA = upA ? i : siA;
          siB = upB ? i : siB;
        }
        // a pair's last column is in this strip; Global wants it on one step only, the pair's last row
        if (MODE != SA_LOCAL && PICK && (s == stripA || s == stripB) &&
            (MODE != SA_GLOBAL || (s == stripA && st == TA + laneA) || (s == stripB && st == TB + laneB))) {
          uint32_t pick = 0;  // low half: pair A's column regA, high half: pair B's column regB
#pragma unroll
          for (int k = 0; k < C; ++k) pick = and_or(Hg[k], pick_mask[k], pick);
          const uint32_t pickA = pick, pickB = pick;
          if (s == stripA && lane == laneA && i <= TA) {
            const int v = (int)(pickA & 0x7FFFu) - (int)bias - gap + baseA;
            if (MODE == SA_SEMIGLOBAL && v > bestA) {  // src/alignment.cpp:87-92, rows in increasing order
              bestA = v;
              biA = i;
            }
            if (MODE == SA_GLOBAL && i == TA) finalA = v;
          }
          if (s == stripB && lane == laneB && i <= TB) {
            const int v = (int)((pickB >> 16) & 0x7FFFu) - (int)bias - gap + baseB;
            if (MODE == SA_SEMIGLOBAL && v > bestB) {
              bestB = v;
              biB = i;
            }
            if (MODE == SA_GLOBAL && i == TB) finalB = v;
          }
        }
      };
      // Lane 63 finishes row st - 63 of the strip's last column in step st.  Sixteen of them are collected in lanes
      // 0 .. 15 of `bout` (row - 1) % 16 and go to the next strip together: lanes 0 .. last hold rows base ..
      auto hand_over = [&](uint32_t last, uint32_t base) __attribute__((always_inline)) {
        if (lane <= last) {
          uint64_t e = (uint64_t)bout;
          if (REL)  // absolute again: lane 63's bases
            e = wave_pk_entry(s + 1, (int)(bout & 0x7FFFu) - (int)bias + b63A,
                              (int)((bout >> 16) & 0x7FFFu) - (int)bias + b63B);
          __hip_atomic_store(&bcol[base + lane], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      // One step anywhere in the sweep: lanes whose row lies outside the matrix sit it out.
      auto step = [&](uint32_t b0, uint32_t st) __attribute__((always_inline)) {
        const uint2 tc = tring[(st - lane - 1) & 255u];  // (lanes that have not started read some other row)
        uint32_t from_left = (uint32_t)dpp_from_lane_below((int)lastcol);
        if (REL) from_left = pk_add_u16(from_left, adj2);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)bnext, st - b0);
        const uint32_t recv = lane == 0 ? first : from_left;
        const uint32_t i = st - lane;  // row of this lane in this step (valid when 1 <= i <= T)
        if (st > lane && i <= T) cells(tc, recv, i, st, drow + (uint64_t)st * 64, std::true_type{});
        if (!last_strip && st >= 64) {
          const uint32_t v63 = (uint32_t)__builtin_amdgcn_readlane((int)lastcol, 63);
          const uint32_t row = st - 63;  // 1 ..
          const uint32_t slot = (row - 1) & 15u;
          bout = lane == slot ? v63 : bout;
          if (slot == 15u || row == T) hand_over(slot, row - slot);
        }
      };
      // Sixteen steps from step g0 (g0 % 16 == 1, g0 >= 64, g0 + 15 <= T: every lane has a row in each of them), of
      // batch b0: nothing is predicated, and everything that depends on the step's number within the group is a
      // constant - the ring offset of the row's symbols, the lane of `bnext`, the slot of lane 63's value.
      // fetch_tag: the group also fetches the inputs of the batch after b0 - the loads leave in its first step and
      // are waited for after its last, with the group's sixteen stores behind them in the queue: waiting for
      // them at the start of the next batch would wait for those stores as well (a microsecond per batch).
      auto steps16 = [&](uint32_t b0, uint32_t g0, auto fetch_tag) __attribute__((always_inline)) {
        constexpr bool FETCH = decltype(fetch_tag)::value;
        const uint2* tp = tring + ((g0 - lane - 1) & 255u);
        uint2* dst = drow + (uint64_t)g0 * 64;
        const uint32_t j0 = g0 - b0;
        // the rows' symbols leave LDS four steps before they are needed
        // (without tables only the first word is in use: a loaded second word would be a dead register, handed out
        // again at once and waited for then)
        uint32_t tcx[16], tcy[16];
        auto ring_get = [&](auto yt) __attribute__((always_inline)) {
          constexpr uint32_t Y = decltype(yt)::value;
          if (TAB) {
            const uint2 v = tp[Y];
            tcx[Y] = v.x;
            tcy[Y] = v.y;
          } else {
            tcx[Y] = tp[Y].x;
            tcy[Y] = 0;
          }
        };
        static_for<4>(ring_get);
        static_for<16>([&](auto xt) __attribute__((always_inline)) {
          constexpr uint32_t X = decltype(xt)::value;
          if (FETCH && X == 0) {
            t_pre = lane < NB ? traw(b0 + NB + lane) : 0u;
            e_pre = load_entries(b0 + NB);
          }
          if (X % 4 == 0 && X + 4 < 16)
            static_for<4>([&](auto yt) __attribute__((always_inline)) {
              ring_get(std::integral_constant<uint32_t, X + 4 + decltype(yt)::value>{});
            });
          const uint2 tc = make_uint2(tcx[X], tcy[X]);
          const int first = __builtin_amdgcn_readlane((int)bnext, j0 + X);
          // lane 0 takes the boundary value, every other lane the value of the lane below
          uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(first, (int)lastcol, 0x138, 0xF, 0xF, false);
          if (REL) recv = pk_add_u16(recv, adj2);  // (lane 0's adjustment is zero)
          cells(tc, recv, g0 + X - lane, g0 + X, dst + (uint64_t)X * 64, std::integral_constant<bool, MODE != SA_GLOBAL>{});
          if (!last_strip) {  // row g0 + X - 63 of the last column: slot (g0 + X - 64) % 16 = (X + 1) % 16
            const int v63 = __builtin_amdgcn_readlane((int)lastcol, 63);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(bout) : "s"(v63), "n"((X + 1) & 15u));
            if (X == 14) hand_over(15u, g0 - 64);
          }
          if (FETCH && X == 15) asm volatile("" ::"v"(t_pre), "v"(e_pre));  // (the wait for the loads comes here)
        });
      };
      for (uint32_t b0 = b_first; b0 <= b_last; b0 += NB) {  // batches of NB steps
        if (lane < NB) ring_put(b0 + lane, tconv(t_pre));  // target positions b0 .. b0 + NB - 1
        uint64_t entry = e_pre;
        const bool full = b0 >= 64 && b0 + NB - 1 <= T;  // every lane has a row in every step of the batch
        if (!full && b0 + NB <= steps) {
          t_pre = lane < NB ? traw(b0 + NB + lane) : 0u;
          e_pre = load_entries(b0 + NB);
        }
        if (s > 0) {  // rows b0 .. b0 + NB - 1 of the previous strip's last column
          if (COOP) {  // (bounded) until all of them carry this strip's tag
            const bool want = lane < NB && b0 + lane <= T;
            for (uint32_t polls = 0; !__all(!want || (uint32_t)(entry >> 48) == s); ++polls) {
              __builtin_amdgcn_s_sleep(1);
              if (polls >= co.spin_limit ||
                  ((polls & 1023u) == 1023u && __hip_atomic_load(co.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(co.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
              }
              entry = load_entries(b0);
            }
          }
          if (!REL) bnext = (uint32_t)entry;
        } else if (!REL) {
          bnext = SG(border_col0<MODE>((int)(b0 + lane), gap) + gap);
        }
        if (REL) {
          if ((b0 & 63u) == 1u) {
            // every 64 steps: the lane's bases move to the value of its first column; what the registers hold
            // follows, and so do the values of lane 63 collected for the next strip
            const int dA = (int)(Hg[0] & 0x7FFFu) - (int)bias, dB = (int)((Hg[0] >> 16) & 0x7FFFu) - (int)bias;
            const uint32_t d2 = ((uint32_t)dA & 0xFFFFu) | ((uint32_t)dB << 16);
#pragma unroll
            for (int k = 0; k < C; ++k) Hg[k] = pk_sub_u16(Hg[k], d2);
            ldiag = pk_sub_u16(ldiag, d2);
            lastcol = pk_sub_u16(lastcol, d2);
            baseA += dA;
            baseB += dB;
            bout = pk_sub_u16(bout, (uint32_t)__builtin_amdgcn_readlane((int)d2, 63));
            b63A = __builtin_amdgcn_readlane(baseA, 63);
            b63B = __builtin_amdgcn_readlane(baseB, 63);
            // (the lane shifts run with every lane enabled: a disabled lane's value cannot be read)
            const int belowA = dpp_from_lane_below(baseA), belowB = dpp_from_lane_below(baseB);
            const uint32_t adj = ((uint32_t)(belowA - baseA) & 0xFFFFu) | ((uint32_t)(belowB - baseB) << 16);
            adj2 = lane == 0 ? 0u : adj;
          }
          // lane 0's left values of the batch, relative to lane 0's bases
          const int b0A = __builtin_amdgcn_readfirstlane(baseA), b0B = __builtin_amdgcn_readfirstlane(baseB);
          const int left0 = border_col0<MODE>((int)(b0 + lane), gap) + gap;
          bnext = s == 0 ? S2(left0 - b0A, left0 - b0B)
                         : S2(wave_pk_entry_a(entry) - b0A, wave_pk_entry_b(entry) - b0B);
        }
        if (full) {
          // (Global: the group with the step in which a pair's last cell is computed takes the general steps)
          auto group = [&](uint32_t g0, auto fetch_tag) __attribute__((always_inline)) {
            const uint32_t pickA = TA + laneA - g0, pickB = TB + laneB - g0;
            if (MODE == SA_GLOBAL && ((s == stripA && pickA < 16u) || (s == stripB && pickB < 16u))) {
              if (decltype(fetch_tag)::value) {
                t_pre = lane < NB ? traw(b0 + NB + lane) : 0u;
                e_pre = load_entries(b0 + NB);
              }
              for (uint32_t st = g0; st < g0 + 16; ++st) step(b0, st);
            } else {
              steps16(b0, g0, fetch_tag);
            }
          };
          group(b0, std::true_type{});
#pragma nounroll
          for (uint32_t g0 = b0 + 16; g0 < b0 + NB; g0 += 16) group(g0, std::false_type{});
        } else {
          const uint32_t e = b0 + NB - 1 < steps ? b0 + NB - 1 : steps;
          for (uint32_t st = b0; st <= e; ++st) step(b0, st);
        }
      }
      if (COOP && !results_here) {  // the registers, for the wavefront of the next segment
        uint32_t* out = state + lane;
#pragma unroll
        for (int k = 0; k < C; ++k) __builtin_nontemporal_store(Hg[k], out + k * 64);
        __builtin_nontemporal_store(ldiag, out + (C + 0) * 64);
        __builtin_nontemporal_store(lastcol, out + (C + 1) * 64);
        __builtin_nontemporal_store(bout, out + (C + 2) * 64);
        __builtin_nontemporal_store((uint32_t)baseA, out + (C + 3) * 64);
        __builtin_nontemporal_store((uint32_t)baseB, out + (C + 4) * 64);
        __builtin_nontemporal_store((uint32_t)bestA, out + (C + 5) * 64);
        __builtin_nontemporal_store(biA, out + (C + 6) * 64);
        __builtin_nontemporal_store((uint32_t)bestB, out + (C + 7) * 64);
        __builtin_nontemporal_store(biB, out + (C + 8) * 64);
        __builtin_nontemporal_store((uint32_t)finalA, out + (C + 9) * 64);
        __builtin_nontemporal_store((uint32_t)finalB, out + (C + 10) * 64);
        __builtin_nontemporal_store(seen[lane], out + (C + 11) * 64);
        __builtin_nontemporal_store(TAB ? 1u : 0u, out + (C + 12) * 64);
        __threadfence();
        if (lane == 63)
          __hip_atomic_store(state + kWaveStateSlots * 64 - 1, seg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (MODE == SA_LOCAL) {  // the strip's candidates against the lane's: higher score, then smaller row
        const int scA = (int)((skey & 0x7FF0u) >> 4) - (int)bias, scB = (int)((skey >> 20) & 0x7FFu) - (int)bias;
        if (scA > bestA || (scA == bestA && scA > 0 && siA < biA)) {
          bestA = scA;
          biA = siA;
          brA = c0 + 16 - (skey & 15u);
        }
        if (scB > bestB || (scB == bestB && scB > 0 && siB < biB)) {
          bestB = scB;
          biB = siB;
          brB = c0 + 16 - ((skey >> 16) & 15u);
        }
      }
    }
    return true;
    };
    if (!(use_tab ? sweep(std::true_type{}) : sweep(std::false_type{}))) return;  // gave up waiting
    if (MODE == SA_LOCAL) {  // the lanes' candidates: highest score, smallest row, smallest column
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int obA = __shfl_xor(bestA, off), obB = __shfl_xor(bestB, off);
        const uint32_t oiA = (uint32_t)__shfl_xor((int)biA, off), orA = (uint32_t)__shfl_xor((int)brA, off);
        const uint32_t oiB = (uint32_t)__shfl_xor((int)biB, off), orB = (uint32_t)__shfl_xor((int)brB, off);
        if (obA > bestA || (obA == bestA && (oiA < biA || (oiA == biA && orA < brA)))) {
          bestA = obA;
          biA = oiA;
          brA = orA;
        }
        if (obB > bestB || (obB == bestB && (oiB < biB || (oiB == biB && orB < brB)))) {
          bestB = obB;
          biB = oiB;
          brB = orB;
        }
      }
    }
    if (MODE != SA_LOCAL) {
    bestA = __builtin_amdgcn_readlane(bestA, laneA);
    biA = (uint32_t)__builtin_amdgcn_readlane((int)biA, laneA);
    finalA = __builtin_amdgcn_readlane(finalA, laneA);
    bestB = __builtin_amdgcn_readlane(bestB, laneB);
    biB = (uint32_t)__builtin_amdgcn_readlane((int)biB, laneB);
    finalB = __builtin_amdgcn_readlane(finalB, laneB);
    }
  }
  if (lane != 0 || !results_here) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && !hasB) break;
    const uint32_t pl = h ? plB : plA, p = h ? pB : pA;
    const uint32_t Qh = h ? QB : QA, Th = h ? TB : TA;
    if (COOP) {  // the strip with the pair's last column knows its result (a pair without cells: strip 0)
      const uint32_t owner = h ? stripB : stripA;
      if (s_mine != (owner == ~0u ? 0u : owner)) continue;
    }
    const int best = h ? bestB : bestA, fin = h ? finalB : finalA;
    uint32_t bi = h ? biB : biA, br = Qh;
    int score;
    if (MODE == SA_LOCAL) {
      const bool any = Qh != 0 && Th != 0 && best > 0;
      score = any ? best : 0;
      bi = any ? bi : 0;
      br = any ? (h ? brB : brA) : 0;
    } else if (Qh == 0 || Th == 0) {  // no cells: the result is a border value
      score = (MODE == SA_GLOBAL) ? (Th == 0 ? border_row0<MODE>((int)Qh, gap) : border_col0<MODE>((int)Th, gap))
                                  : border_row0<MODE>((int)Qh, gap);
      bi = (MODE == SA_GLOBAL) ? Th : 0;
    } else if (MODE == SA_GLOBAL) {
      score = fin;
      bi = Th;
    } else {
      score = bi ? best : border_row0<MODE>((int)Qh, gap);
    }
    a.score[p] = score;
    a.end_i[pl] = bi;
    a.end_r[pl] = br;
  }
}

// One wavefront per two pairs; COOP: per ticket, (segment, unit, strip) in this order, in the order in which the
// wavefronts start.  A wavefront only ever waits for the work of an earlier ticket, which is done or with a
// wavefront that runs, so the launch ends however few of its wavefronts the GPU holds at a time.
// (Three wavefronts per SIMD: left to itself the compiler takes a few registers too many for the cooperative form
// with sixteen columns and the GPU then holds two; held to three it spills a few pointers before and after the
// strip's loop, none inside.)
template <int MODE, int C, bool COOP>
__global__ __launch_bounds__(64, 3) void fill_wave_pk16(AlignArgs a, WaveCoop co, uint32_t bias, uint32_t tables) {
  uint32_t ticket = blockIdx.x;
  if (COOP) {
    if (threadIdx.x == 0) ticket = atomicAdd(co.ticket, 1u);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
  }
  wave_pk16_task<MODE, C, COOP>(a, co, bias, tables, ticket);
}

}  // namespace sa

#endif
