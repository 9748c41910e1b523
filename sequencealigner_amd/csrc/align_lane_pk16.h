// Inter-pair batched DP fill for short pairs: TWO pairs per lane (128 pairs per wavefront), scores as
// packed unsigned 16-bit halves of one VGPR.  Same recurrence, tie-break and clamping as the reference
// (src/alignment.cpp:63-85, 176-201, 281-303); what differs is only the number representation:
//
//   * a score x is held as S(x) = x + BIAS with bit 15 of the half always set (a "guard" bit), so that
//     plain 32-bit v_add/v_sub on the register act as two independent 16-bit operations (no carry or
//     borrow ever crosses the halves) and unsigned packed max orders the halves correctly;
//   * the register strip runs along the TARGET (K rows), the lane's loop along the QUERY;
//     Hc[k] holds S(H[i][r-1] + gap) of its row, which serves both as the "left" candidate of the next
//     column and, shifted by (match - gap), as the "diagonal" base of the row below;
//   * per cell two flags are kept, "diag candidate != max" and "left candidate != max": the sign of the
//     packed difference candidate - max (candidates never exceed the max).  Diag > Left > Up follows
//     from reading the diag flag first, then the left flag.  Local's None state is not stored: a cell is
//     None exactly when its score is 0, which the traceback knows from the running score.
//   * Local's best cell (first strict maximum in row-major order, src/alignment.cpp:203-207) is tracked
//     with one packed max per cell on the key (S(H) << 6) | (63 - k): larger score first, then smaller
//     row inside the strip; strips are visited in increasing row order and columns in increasing order.
//
// Preconditions (checked on the host by lane_pk16_applicable): gap <= 0, mismatch <= match,
// match >= gap, all scores within the 15-bit window; Local additionally gap < 0, mismatch < 0 and a
// 9-bit score window.  Lengths may differ between the pairs of a batch: positions beyond a pair's own
// query or target hold padding symbols that match nothing, and every pair's result is taken at its own
// last column / row.
#ifndef SA_ALIGN_LANE_PK16_H
#define SA_ALIGN_LANE_PK16_H

#include "align_common.h"

namespace sa {

constexpr uint32_t kGuard2 = 0x80008000u;
constexpr uint32_t kQueryPad = 0x0100u;   // symbol codes outside the byte range: never equal to a real
constexpr uint32_t kTargetPad = 0x0200u;  // symbol nor to each other

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// per half: a * b + c (mod 2^16)
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// (mask & a) | (~mask & b)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}
// per half: a - b (mod 2^16)
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a & mask) | c, with the mask in a scalar register
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t rep2(uint32_t v16) { return (v16 & 0xFFFFu) * 0x00010001u; }

// 0x7FFF in every half whose value is non-zero (value < 0x8000), else 0.
__device__ __forceinline__ uint32_t nonzero_mask15(uint32_t v) {
  const uint32_t y = (v + 0x7FFF7FFFu) & kGuard2;
  return y - (y >> 15);
}

struct Pk16Args {
  // Packed sequences, written by pack_lane_pk16 in one of two forms (see AlphabetInfo::small):
  //  symbols: qpk[unit][max_q][64] dwords, tpk[unit][t_pad][64] dwords - the byte of pair A in the low
  //           half, of pair B in the high half (pads: kQueryPad / kTargetPad);
  //  tables:  one BYTE per position, four positions per dword: qpk[unit][max_q / 4][64], tpk[unit][t_pad / 4][64]
  //           (a quarter of the symbols form's bytes to write, and to read once per strip).
  //           Query byte: the symbol CODES (0..3) of pair A / pair B in the low / high nibble, 0xF beyond the
  //           query; per column the fill expands each code into a 4-byte table whose byte L is
  //           (sub(L, q) - gap) << SH (all 0 beyond the query).
  //           Target byte: the nibbles of the v_perm selector {code A, 0x0C, 4 + code B, 0x0C} that picks the
  //           pair's entry out of {table B, table A} (0xC = constant 0, also used beyond the target).
  const uint32_t* qpk;
  const uint32_t* tpk;
  uint32_t t_pad;       // nstrips * K
  uint32_t bias;        // BIAS
  uint32_t units;       // waves of 128 pairs in this pass
  const struct AlphabetInfo* alpha;  // NULL: symbols form only
  uint32_t entry_match, entry_mismatch;  // tables form: (match - gap) << SH, (mismatch - gap) << SH
};

// Result of scanning a batch's bytes: with at most four distinct symbols the substitution score of a
// cell is a table lookup (one v_perm_b32) instead of compare + select arithmetic.  A large batch is only
// sampled (alphabet_scan); the packing kernel, which reads every byte anyway, checks the sample's verdict
// and raises `violated` at the first byte the sample had not seen - the pass is then packed again and
// filled in the symbols form.
struct AlphabetInfo {
  uint32_t present[8];  // bit b set: byte value b occurs in the bytes that were scanned
  uint32_t small;       // 1: at most 4 distinct bytes, code[] maps them to 0..3 (kUnseen for the others)
  uint32_t violated;    // 1: a byte outside code[] turned up while packing; the tables form is off
  uint8_t code[256];
};
constexpr uint32_t kUnseen = 0x80;

// Number of direction dwords a lane writes per query column and strip: 2 planes per group of 16 rows.
template <int K>
struct LanePk16Geom {
  static constexpr int NG = (K + 15) / 16;
  static constexpr int WORDS = 2 * NG;
};

// Direction storage of the packed lane kernel.  Per strip s, per block of 8 query columns rb and per
// group g of 16 target rows, the 64 lanes of a wave own one 4 KiB tile:
//     dword[((s * nrb + rb) * NG + g) * 1024 + lane * 16 + (column % 8) * 2 + plane]
// i.e. every lane (two pairs) has its 8 columns x 2 planes contiguous in one 64-byte segment, which is
// what the traceback fetches: a diagonal path stays ~8 steps inside one segment.  The fill kernel
// stages the words of 8 columns in LDS and writes each tile with fully coalesced 1 KiB stores.
constexpr int kColBlock = 8;

template <int K>
struct LanePk16Tiles {
  static constexpr int NG = LanePk16Geom<K>::NG;
  __host__ __device__ static uint32_t col_blocks(uint32_t max_q) { return (max_q + kColBlock - 1) / kColBlock; }
  __host__ __device__ static uint64_t unit_dwords(uint32_t nstrips, uint32_t max_q) {
    return (uint64_t)nstrips * col_blocks(max_q) * NG * 1024;
  }
};

// Per-lane state of the fill that outlives a strip.
struct LanePk16State {
  uint32_t best;     // Local: packed keys S(best score) | (63 - row in strip), guard bit set
  uint32_t bi2, br2; // Local: packed row / column of the best cells
  uint32_t sgbest;   // Semiglobal: packed S(best last-column score + gap), guard bit set
  uint32_t sgrow2;   // Semiglobal: packed rows of those
  uint32_t final2;   // Global: packed S(H[T][Q] + gap)
};

// One strip of KS target rows (rows i0+1 .. i0+KS) over all query columns.  KMAX (>= KS) is the
// strip height the direction tiles and the LDS staging area are laid out for.  SH is the left shift
// applied to every score (6 for Local, where the low 6 bits of the best-cell key hold the row).
template <int MODE, int KS, int KMAX, int SH, bool PERM>
__device__ __forceinline__ void lane_pk16_strip(const AlignArgs& a, const Pk16Args& x, uint32_t s,
                                                uint32_t nstrips, uint32_t maxQ, uint32_t lane,
                                                const uint32_t* qpk, const uint32_t* tpk, uint32_t* brow,
                                                uint32_t* dirs, uint2* stage, uint32_t QA, uint32_t QB,
                                                uint32_t TA, uint32_t TB, LanePk16State& st) {
  constexpr int NG = LanePk16Geom<KMAX>::NG;
  constexpr int NGS = LanePk16Geom<KS>::NG;
  const int gap = a.gap;
  const uint32_t G = kGuard2;
  const uint32_t bias = x.bias;
  const uint32_t gsub2 = rep2((uint32_t)(-gap) << SH);                       // |gap|, gap <= 0
  const uint32_t mg2 = rep2((uint32_t)(a.match - gap) << SH);                // match - gap >= 0
  const uint32_t negdelta2 = rep2((uint32_t)((a.mismatch - a.match) * (1 << SH)));  // mod 2^16
  const uint32_t one2 = 0x00010001u;
  const uint32_t floorG = rep2(bias << SH) | G;  // S(0)
  auto SG = [&](int v) -> uint32_t { return rep2((uint32_t)((int)bias + v) << SH) | G; };
  const uint32_t nrb = LanePk16Tiles<KMAX>::col_blocks(a.max_q);
  const uint32_t i0 = s * KMAX;

  // Writes the staged words of column block rb as NGS tiles of 4 KiB.  Lane L stores 16 bytes at tile
  // offset c*1024 + L*16: segment (16c + L/4), columns 2*(L%4) and 2*(L%4)+1.
  auto flush_tiles = [&](uint32_t rb) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < NGS; ++g) {
      uint32_t* tile = dirs + ((uint64_t)(s * nrb + rb) * NG + g) * 1024;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t seg = 16 * c + (lane >> 2), col = 2 * (lane & 3);
        const uint2 w0 = stage[(g * kColBlock + col) * 64 + seg];
        const uint2 w1 = stage[(g * kColBlock + col + 1) * 64 + seg];
        // written once, read once by the traceback after 13 GB more have gone by: keep it out of the caches
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{w0.x, w0.y, w1.x, w1.y},
                                    reinterpret_cast<u32x4*>(tile + c * 256 + lane * 4));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  uint32_t Tk[KS], Hc[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    if (!PERM) Tk[k] = tpk[(uint64_t)(i0 + k) * 64];
    Hc[k] = SG(border_col0<MODE>((int)(i0 + 1 + k), gap) + gap);  // column 0: S(H[i][0] + gap)
  }
  if (PERM) {  // four rows per dword (i0 is a multiple of 16): selector nibbles -> selectors
#pragma unroll
    for (int k4 = 0; k4 < KS; k4 += 4) {
      const uint32_t w = tpk[(uint64_t)(i0 / 4 + k4 / 4) * 64];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (k4 + x >= KS) break;
        const uint32_t b = (w >> (8 * x)) & 0xFFu;
        Tk[k4 + x] = (b & 0xFu) | ((b >> 4) << 16) | 0x0C000C00u;
      }
    }
  }
  // S(H[i0][0] + gap): base of the diagonal of row i0+1 at column 1
  uint32_t top_prev = SG(((i0 == 0) ? 0 : border_col0<MODE>((int)i0, gap)) + gap);
  const uint32_t best_strip0 = st.best;
  uint32_t best = st.best;
  const uint32_t tab_base = x.entry_mismatch * 0x01010101u;
  const uint32_t tab_delta = x.entry_match - x.entry_mismatch;
  uint32_t qq = qpk[0];  // symbols: column 1's symbols; tables: the bytes of columns 1..4
  uint32_t qq_next = 0;
  uint32_t top_next = 0;
  if (s != 0)
    top_next = __hip_atomic_load(&brow[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t r = 1; r <= maxQ; ++r) {
    // symbols form: x = the two pairs' query symbols; tables form: (x, y) = the two pairs' score tables
    uint2 qcur = make_uint2(qq, 0);
    if (PERM) {
      const uint32_t x = (r - 1) & 3u;  // the column's byte of the group's dword
      if (x == 0 && r + 4 <= maxQ) qq_next = qpk[(uint64_t)((r - 1) / 4 + 1) * 64];  // the next four columns
      const uint32_t b = (qq >> (8 * x)) & 0xFFu;
      const uint32_t ca = b & 0xFu, cb = b >> 4;
      qcur.x = ca > 3 ? 0u : tab_base + (tab_delta << (8 * ca));
      qcur.y = cb > 3 ? 0u : tab_base + (tab_delta << (8 * cb));
      if (x == 3) qq = qq_next;
    } else if (r < maxQ) {
      qq = qpk[(uint64_t)r * 64];  // prefetch the next column
    }
    // The boundary row of the strip above is fetched one column ahead as well.  (Read past this CU's
    // L1: the same addresses were read and then overwritten while the previous strip ran.)
    const uint32_t top = (s == 0) ? SG(border_row0<MODE>((int)r, gap) + gap) : top_next;
    if (s != 0 && r < maxQ)
      top_next = __hip_atomic_load(&brow[(uint64_t)(r + 1) * 64], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    uint32_t dsrc = top_prev, up = top;
    const uint32_t best_before = best;
#pragma unroll
    for (int c0 = 0; c0 < KS; c0 += 16) {
      constexpr int CHMAX = 16;
      // pass 1 (no dependencies between cells): diagonal candidates S(H[i-1][r-1] + sub) of the chunk,
      // from the previous column's values still held in Hc
      uint32_t dG[CHMAX];
#pragma unroll
      for (int j = 0; j < CHMAX; ++j) {
        const int k = c0 + j;
        if (k >= KS) break;
        const uint32_t base = (j == 0) ? dsrc : Hc[k - 1];
        if (PERM) {
          // base = S(H[i-1][r-1] + gap); the table entry is (sub - gap) << SH >= 0
          dG[j] = base + __builtin_amdgcn_perm(qcur.y, qcur.x, Tk[k]);
        } else {
          const uint32_t ne = pk_min_u16(Tk[k] ^ qcur.x, one2);  // 1 where the symbols differ
          dG[j] = pk_mad_u16(ne, negdelta2, base + mg2);
        }
      }
      // previous column's value of the chunk's last row, for the next chunk
      dsrc = Hc[(c0 + CHMAX < KS ? c0 + CHMAX : KS) - 1];
      // pass 2: the dependent chain down the column.  Flags: candidate - max is negative (bit 15 of the
      // half set) exactly when the candidate is NOT the maximum; one v_perm turns the four sign bits
      // (left/diag x pair A/B) into four 0x00/0xFF bytes, of which bit (k % 8) is kept per cell.
      uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
      for (int j = 0; j < CHMAX; ++j) {
        const int k = c0 + j;
        if (k >= KS) break;
        const uint32_t left = Hc[k];  // S(H[i][r-1] + gap)
        uint32_t mG = pk_max_u16(dG[j], up);
        if (MODE == SA_LOCAL)
          mG = pk_max_u16(mG, pk_max_u16(left, floorG));
        else
          mG = pk_max_u16(mG, left);
        const uint32_t signs =
            __builtin_amdgcn_perm(pk_sub_u16(dG[j], mG), pk_sub_u16(left, mG), 0x0B0A0908u);
        if (j < 8)
          acc0 = and_or(signs, 0x01010101u << (j & 7), acc0);
        else
          acc1 = and_or(signs, 0x01010101u << (j & 7), acc1);
        if (MODE == SA_LOCAL) best = pk_max_u16(best, mG | rep2(63 - k));
        const uint32_t hg = mG - gsub2;  // S(H[i][r] + gap)
        Hc[k] = hg;
        up = hg;
      }
      stage[((c0 >> 4) * kColBlock + ((r - 1) & (kColBlock - 1))) * 64 + lane] = make_uint2(acc0, acc1);
    }
    if (((r - 1) & (kColBlock - 1)) == kColBlock - 1 || r == maxQ) flush_tiles((r - 1) / kColBlock);
    if (s + 1 < nstrips) brow[(uint64_t)r * 64] = up;
    top_prev = top;
    if (MODE == SA_LOCAL) {
      const uint32_t ch = nonzero_mask15(best ^ best_before);
      st.br2 = bfi(ch, rep2(r), st.br2);
    } else if (r == QA || r == QB) {
      // This is the last column of one (or both) of the lane's pairs: Hc holds S(H[i][Q] + gap).
      const uint32_t hm = (r == QA ? 0x0000FFFFu : 0u) | (r == QB ? 0xFFFF0000u : 0u);
      if (MODE == SA_SEMIGLOBAL) {
        // first strict maximum > 0 over the rows in increasing order (src/alignment.cpp:87-92)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const uint32_t i = i0 + 1 + k;
          const uint32_t nb = pk_max_u16(st.sgbest, Hc[k]);
          uint32_t ch = nonzero_mask15(nb ^ st.sgbest) & hm;
          if (i > TA) ch &= 0xFFFF0000u;  // rows beyond the pair's target are padding
          if (i > TB) ch &= 0x0000FFFFu;
          st.sgbest = bfi(ch, nb, st.sgbest);
          st.sgrow2 = bfi(ch, rep2(i), st.sgrow2);
        }
      } else {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const uint32_t i = i0 + 1 + k;
          if (i == TA && r == QA) st.final2 = (st.final2 & 0xFFFF0000u) | (Hc[k] & 0x0000FFFFu);
          if (i == TB && r == QB) st.final2 = (st.final2 & 0x0000FFFFu) | (Hc[k] & 0xFFFF0000u);
        }
      }
    }
  }
  if (MODE == SA_LOCAL) {
    const uint32_t ch = nonzero_mask15(best ^ best_strip0);
    const uint32_t rows = rep2(i0 + 64) - (best & 0x003F003Fu);  // i0 + 1 + k, k = 63 - low bits
    st.bi2 = bfi(ch, rows, st.bi2);
    st.best = best | 0x003F003Fu;  // a later strip (larger rows) wins only with a strictly larger score
  }
}

// All strips of one unit: full strips of K rows, then one strip of 16/32/48 rows for the remainder.
template <int MODE, int K, int SH, bool PERM>
__device__ __forceinline__ void lane_pk16_strips(const AlignArgs& a, const Pk16Args& x, uint32_t nfull,
                                                 uint32_t rem, uint32_t nstrips, uint32_t maxQ,
                                                 uint32_t lane, const uint32_t* qpk, const uint32_t* tpk,
                                                 uint32_t* brow, uint32_t* dirs, uint2* stage,
                                                 uint32_t QA, uint32_t QB, uint32_t TA, uint32_t TB,
                                                 LanePk16State& st) {
  for (uint32_t s = 0; s < nfull; ++s)
    lane_pk16_strip<MODE, K, K, SH, PERM>(a, x, s, nstrips, maxQ, lane, qpk, tpk, brow, dirs, stage, QA,
                                          QB, TA, TB, st);
  if (rem) {
    if (rem <= 16)
      lane_pk16_strip<MODE, 16, K, SH, PERM>(a, x, nfull, nstrips, maxQ, lane, qpk, tpk, brow, dirs,
                                             stage, QA, QB, TA, TB, st);
    else if (rem <= 32 || K <= 32)
      lane_pk16_strip<MODE, (K < 32 ? K : 32), K, SH, PERM>(a, x, nfull, nstrips, maxQ, lane, qpk, tpk,
                                                            brow, dirs, stage, QA, QB, TA, TB, st);
    else if (rem <= 48 || K <= 48)
      lane_pk16_strip<MODE, (K < 48 ? K : 48), K, SH, PERM>(a, x, nfull, nstrips, maxQ, lane, qpk, tpk,
                                                            brow, dirs, stage, QA, QB, TA, TB, st);
    else
      lane_pk16_strip<MODE, K, K, SH, PERM>(a, x, nfull, nstrips, maxQ, lane, qpk, tpk, brow, dirs,
                                            stage, QA, QB, TA, TB, st);
  }
}

// TABLES: the launch may use the score-table form of the packed sequences (decided by the device-side
// alphabet scan); without it only the symbols form is compiled in.
template <int MODE, int K, bool TABLES>
__global__ __launch_bounds__(256) void fill_lane_pk16(AlignArgs a, Pk16Args x) {
  constexpr int NG = LanePk16Geom<K>::NG;
  constexpr int SH = (MODE == SA_LOCAL) ? 6 : 0;
  // per wave: NG groups x 8 columns x 64 lanes of (plane A, plane B) words awaiting the tile store
  __shared__ uint2 stage_all[4][NG * kColBlock * 64];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (unit >= x.units) return;  // whole waves only: blockDim is a multiple of 64
  uint2* stage = stage_all[threadIdx.x >> 6];

  // the two pairs of this lane
  const uint32_t plA = unit * 128 + lane, plB = plA + 64;
  const bool actA = plA < a.pairs_in_pass, actB = plB < a.pairs_in_pass;
  uint32_t QA = 0, TA = 0, QB = 0, TB = 0;
  if (actA) {
    const uint32_t p = a.pair_base + plA;
    QA = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
    TA = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  }
  if (actB) {
    const uint32_t p = a.pair_base + plB;
    QB = (uint32_t)(a.q_off[p + 1] - a.q_off[p]);
    TB = (uint32_t)(a.t_off[p + 1] - a.t_off[p]);
  }
  const uint32_t maxQ = wave_max_u32(QA > QB ? QA : QB);
  const uint32_t maxT = wave_max_u32(TA > TB ? TA : TB);
  // full strips of K rows, then one strip of 16/32/48 rows covering the remainder
  const uint32_t nfull = maxT / K, rem = maxT - nfull * K;
  const uint32_t nstrips = nfull + (rem ? 1 : 0);

  uint32_t* brow = reinterpret_cast<uint32_t*>(a.brow) + (uint64_t)unit * a.brow_unit_stride + lane;
  uint32_t* dirs = a.dirs + (uint64_t)unit * a.dirs_unit_stride;
  const uint32_t* tpk = x.tpk + (uint64_t)unit * x.t_pad * 64 + lane;

  const int gap = a.gap;
  const uint32_t bias = x.bias;
  LanePk16State st;
  st.best = (rep2(bias << SH) | kGuard2) | 0x003F003Fu;  // key of "score 0": no 0 cell beats it
  st.bi2 = st.br2 = 0;
  st.sgbest = rep2((uint32_t)((int)bias + gap)) | kGuard2;  // S(0 + gap), compared with Hc values
  st.sgrow2 = 0;
  st.final2 = 0;

  // The batch scan decides (on the device, uniformly for the launch) which packed form was written.
  const bool tables = TABLES && x.alpha->small && !x.alpha->violated;
  if (tables) {
    const uint32_t* qpk = x.qpk + (uint64_t)unit * a.max_q * 64 + lane;
    lane_pk16_strips<MODE, K, SH, true>(a, x, nfull, rem, nstrips, maxQ, lane, qpk, tpk, brow, dirs,
                                        stage, QA, QB, TA, TB, st);
  } else {
    const uint32_t* qpk = x.qpk + (uint64_t)unit * a.max_q * 64 + lane;
    lane_pk16_strips<MODE, K, SH, false>(a, x, nfull, rem, nstrips, maxQ, lane, qpk, tpk, brow, dirs,
                                         stage, QA, QB, TA, TB, st);
  }

  // unpack the two pairs' results
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bool act = h ? actB : actA;
    if (!act) continue;
    const uint32_t pl = h ? plB : plA;
    const uint32_t Q = h ? QB : QA, T = h ? TB : TA;
    const uint32_t p = a.pair_base + pl;
    const uint32_t sh = 16 * h;
    int score;
    uint32_t ei, er;
    if (MODE == SA_LOCAL) {
      score = (int)(((st.best >> sh) & 0x7FFFu) >> 6) - (int)bias;
      ei = (st.bi2 >> sh) & 0xFFFFu;
      er = (st.br2 >> sh) & 0xFFFFu;
    } else if (MODE == SA_SEMIGLOBAL) {
      ei = (st.sgrow2 >> sh) & 0xFFFFu;
      er = Q;
      score = ei ? (int)((st.sgbest >> sh) & 0x7FFFu) - (int)bias - gap : border_row0<MODE>((int)Q, gap);
    } else {
      ei = T;
      er = Q;
      if (Q == 0 || T == 0)
        score = (T == 0) ? border_row0<MODE>((int)Q, gap) : border_col0<MODE>((int)T, gap);
      else
        score = (int)((st.final2 >> sh) & 0x7FFFu) - (int)bias - gap;
    }
    a.score[p] = score;
    a.end_i[pl] = ei;
    a.end_r[pl] = er;
  }
}

// Traceback + run-length encoding for the tiled layout above, one pair per thread; same rules as
// traceback_kernel (align_common.h).  The thread keeps the current 64-byte direction segment in LDS (one
// dword read per step) and 8 symbols of each sequence in registers, so a path of ~(Q+T) steps costs
// ~(Q+T)/8 memory round trips.  The step itself is written without branches (the 64 paths of a wave
// diverge at every step): the move is decoded into "consumes a row" / "consumes a column" flags.
template <int MODE, int K>
__global__ __launch_bounds__(256) void traceback_lane_pk16(AlignArgs a) {
  constexpr int NG = LanePk16Geom<K>::NG;
  constexpr uint32_t kSegStride = 17;  // dwords per thread in LDS: 16 + 1, conflict-free across lanes
  __shared__ uint32_t seg[256 * kSegStride];
  const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= a.pairs_in_pass) return;
  const uint32_t p = a.pair_base + pl;
  const uint64_t qb = a.q_off[p], tb0 = a.t_off[p];
  const uint32_t Q = (uint32_t)(a.q_off[p + 1] - qb);
  const uint32_t T = (uint32_t)(a.t_off[p + 1] - tb0);
  uint32_t i = (MODE == SA_GLOBAL) ? T : a.end_i[pl];
  uint32_t r = (MODE == SA_LOCAL) ? a.end_r[pl] : Q;
  const bool same = (a.match == a.mismatch);
  const uint32_t unit = pl >> 7, lane = pl & 63, half = (pl >> 6) & 1;
  const uint32_t nrb = LanePk16Tiles<K>::col_blocks(a.max_q);
  const uint32_t* dirs = a.dirs + (uint64_t)unit * a.dirs_unit_stride + lane * 16;
  uint32_t* myseg = seg + threadIdx.x * kSegStride;

  uint32_t* slot_end = a.cigar_runs ? a.cigar_runs + (uint64_t)(p + 1) * a.cigar_stride : nullptr;
  uint32_t nruns = 0, cur_op = 4, cur_len = 0;
  int remaining = (MODE == SA_LOCAL) ? a.score[p] : 1;

  uint32_t cached_tile = 0xFFFFFFFFu;
  // Symbols come through 16-byte windows held in registers.  Two sources: the pair's record of 2-bit codes
  // (a window is 64 symbols; tables form - a million 150 x 300 pairs then fetch 8 windows each instead of 28,
  // and every miss costs a 128-byte line), or the blobs themselves (a window is 16 bytes, 16-byte aligned in the
  // blob; fetched with one 16-byte load where that stays within 8 bytes of the pair's end, which is what the
  // blobs promise, else with an 8-byte load).
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const bool use_rec = a.rec && a.rec_alpha->small && !a.rec_alpha->violated;  // the same for every thread
  const uint32_t* myrec =
      use_rec ? a.rec + (uint64_t)unit * a.rec_unit_stride + (uint64_t)(pl & 127u) * a.rec_stride : nullptr;
  const uint8_t* qarr = use_rec ? reinterpret_cast<const uint8_t*>(myrec) : a.q;
  const uint8_t* tarr = use_rec ? reinterpret_cast<const uint8_t*>(myrec + 4 * a.rec_q_chunks) : a.t;
  const uint64_t qorg = use_rec ? 0 : qb, torg = use_rec ? 0 : tb0;  // index of the pair's first symbol
  const uint64_t qend = use_rec ? ~0ull >> 8 : qb + Q, tend = use_rec ? ~0ull >> 8 : tb0 + T;  // in bytes
  const uint32_t wshift = use_rec ? 6 : 4, sbits = use_rec ? 2 : 8, smask = use_rec ? 3u : 0xFFu;
  // The walk moves towards smaller indices one symbol at a time, so the window below the current one is
  // fetched as soon as the current one is entered: thousands of threads share a CU and each iteration of a
  // wavefront would otherwise wait for some lane's window.
  constexpr uint64_t kNoWindow = ~0ull;
  uint64_t qwin_id = kNoWindow, twin_id = kNoWindow;
  u32x4 qwin = {0, 0, 0, 0}, twin = {0, 0, 0, 0}, qnext = {0, 0, 0, 0}, tnext = {0, 0, 0, 0};
  constexpr uint64_t kNoFetch = ~0ull;
  uint64_t q_fetch = kNoFetch, t_fetch = kNoFetch;  // byte offset of the window to fetch at the end of this iteration
  auto load_window = [&](const uint8_t* arr, uint64_t base, uint64_t end) -> u32x4 {
    if (base + 16 <= end + 8) return *reinterpret_cast<const u32x4*>(arr + base);
    const uint64_t lo = *reinterpret_cast<const uint64_t*>(arr + base);
    return u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), 0u, 0u};
  };
  auto window_symbol = [&](u32x4 w, uint64_t index) -> uint32_t {
    const uint32_t bit = ((uint32_t)index & ((1u << wshift) - 1u)) * sbits;
    const uint32_t d = bit >> 5;
    const uint32_t word = d == 0 ? w.x : d == 1 ? w.y : d == 2 ? w.z : w.w;
    return (word >> (bit & 31u)) & smask;
  };

  for (;;) {
    if (MODE == SA_GLOBAL && i == 0 && r == 0) break;
    if (MODE == SA_SEMIGLOBAL && r == 0) break;
    // Local (gap < 0 here): the borders are None, and an inner cell with score 0 is None
    if (MODE == SA_LOCAL && (remaining == 0 || i == 0 || r == 0)) break;
    const bool interior = i != 0 && r != 0;
    const uint32_t ii = interior ? i - 1 : 0, rr = interior ? r - 1 : 0;
    const uint32_t s = ii / K, k = ii - s * K;
    const uint32_t g = k >> 4, j = k & 15;
    const uint32_t tile = (s * nrb + (rr >> 3)) * NG + g;
    if (interior && tile != cached_tile) {
      const uint4* src = reinterpret_cast<const uint4*>(dirs + (uint64_t)tile * 1024);
      const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
      myseg[0] = v0.x; myseg[1] = v0.y; myseg[2] = v0.z; myseg[3] = v0.w;
      myseg[4] = v1.x; myseg[5] = v1.y; myseg[6] = v1.z; myseg[7] = v1.w;
      myseg[8] = v2.x; myseg[9] = v2.y; myseg[10] = v2.z; myseg[11] = v2.w;
      myseg[12] = v3.x; myseg[13] = v3.y; myseg[14] = v3.z; myseg[15] = v3.w;
      cached_tile = tile;
    }
    // per column a lane stores two dwords: rows 0-7 and rows 8-15 of the group; byte 2+half holds the
    // "diagonal candidate is not the maximum" bits, byte half the "left candidate is not" bits
    const uint32_t word = myseg[2 * (rr & 7) + (j >> 3)] >> (8 * half + (j & 7));
    const bool not_left = word & 1u, not_diag = (word >> 16) & 1u;
    // border cells (Global / Semiglobal only get here): row 0 moves Left, column 0 moves Up
    const bool is_diag = interior && !not_diag;
    const bool is_left = interior ? (not_diag && !not_left) : (i == 0);
    bool eq = false;
    if (is_diag) {
      const uint64_t qa = qorg + rr, ta = torg + ii;
      if ((qa >> wshift) != qwin_id) {
        const uint64_t id = qa >> wshift;
        qwin = (id + 1 == qwin_id) ? qnext : load_window(qarr, id * 16, qend);
        qwin_id = id;
        q_fetch = id ? (id - 1) * 16 : kNoFetch;
      }
      if ((ta >> wshift) != twin_id) {
        const uint64_t id = ta >> wshift;
        twin = (id + 1 == twin_id) ? tnext : load_window(tarr, id * 16, tend);
        twin_id = id;
        t_fetch = id ? (id - 1) * 16 : kNoFetch;
      }
      eq = window_symbol(qwin, qa) == window_symbol(twin, ta);
    }
    const uint32_t op = is_diag ? ((same || eq) ? SA_OP_M : SA_OP_X) : (is_left ? SA_OP_I : SA_OP_D);
    if (MODE == SA_LOCAL) remaining -= is_diag ? (eq ? a.match : a.mismatch) : a.gap;
    i -= (is_diag || !is_left) ? 1u : 0u;
    r -= (is_diag || is_left) ? 1u : 0u;
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
    // The prefetches go out last, after every use of loaded data in this iteration: a wait placed before such a
    // use would otherwise wait for them as well.  (The window below the current one ends at or before the
    // current symbol: a plain 16-byte load stays inside the pair.)
    if (q_fetch != kNoFetch) {
      qnext = *reinterpret_cast<const u32x4*>(qarr + q_fetch);
      q_fetch = kNoFetch;
    }
    if (t_fetch != kNoFetch) {
      tnext = *reinterpret_cast<const u32x4*>(tarr + t_fetch);
      t_fetch = kNoFetch;
    }
  }
  if (cur_len) {
    ++nruns;
    if (slot_end) slot_end[-(int64_t)nruns] = (cur_len << 2) | cur_op;
  }
  if (a.cigar_nruns) a.cigar_nruns[p] = nruns;
  if (a.target_begin) {
    if (MODE == SA_LOCAL)
      a.target_begin[p] = i;
    else
      a.target_begin[p] = (cur_len && cur_op == SA_OP_D && cur_len <= 9) ? 3u : 1u;
  }
}

// Marks which byte values occur in the query and target blobs of a batch: every thread looks at up to
// kScanWords 8-byte words, the words spread evenly over the blob - all of them when the blob has no more
// words than that allows (2 MiB with the grid the caller launches), a sample otherwise.  A workgroup
// collects the bytes in a 256-entry LDS table indexed by the byte itself (plain stores of 1: every writer
// writes the same value), then ORs the entries that were hit into the global bit set - a few atomics per
// workgroup, all on the same words, which is why the grid is kept small.
constexpr uint32_t kScanWords = 4;
constexpr uint32_t kScanBlocks = 256;
__global__ __launch_bounds__(256) void alphabet_scan(const uint8_t* q, const uint64_t* q_off,
                                                     const uint8_t* t, const uint64_t* t_off, uint32_t n,
                                                     AlphabetInfo* info) {
  __shared__ uint8_t seen[256];
  seen[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const uint8_t* base = which ? t : q;
    const uint64_t b0 = which ? t_off[0] : q_off[0], b1 = which ? t_off[n] : q_off[n];
    // 8 bytes per thread and iteration through aligned loads (the blob is 8-byte aligned and readable
    // up to 8 bytes past its end)
    const uint64_t w0 = b0 & ~7ull, w1 = (b1 + 7) & ~7ull;
    const uint64_t slots = nthreads * kScanWords;
    const uint64_t every = ((w1 - w0) / 8 + slots - 1) / slots;  // 1: all words
#pragma unroll
    for (uint32_t it = 0; it < kScanWords; ++it) {
      const uint64_t w = w0 + 8 * (tid + it * nthreads) * every;
      if (w >= w1) break;
      const uint64_t word = *reinterpret_cast<const uint64_t*>(base + w);
      const bool inner = w >= b0 && w + 8 <= b1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!inner && (w + j < b0 || w + j >= b1)) continue;
        seen[(uint32_t)(word >> (8 * j)) & 0xFFu] = 1;
      }
    }
  }
  __syncthreads();
  if (seen[threadIdx.x]) atomicOr(&info->present[threadIdx.x >> 5], 1u << (threadIdx.x & 31));
}

// Numbers the symbols that occur (in increasing byte order) when there are at most four of them.
__global__ void alphabet_table(AlphabetInfo* info) {  // one workgroup of 256 threads
  const uint32_t b = threadIdx.x;
  uint32_t below = 0, total = 0;
#pragma unroll
  for (uint32_t wd = 0; wd < 8; ++wd) {
    const uint32_t m = info->present[wd];
    total += __builtin_popcount(m);
    if (wd < (b >> 5))
      below += __builtin_popcount(m);
    else if (wd == (b >> 5))
      below += __builtin_popcount(m & ((1u << (b & 31)) - 1u));
  }
  const bool on = (info->present[b >> 5] >> (b & 31)) & 1u;
  info->code[b] = (uint8_t)(on && below < 4 ? below : kUnseen);
  if (b == 0) info->small = total <= 4 ? 1u : 0u;
}

// Rewrites the byte sequences of a pass in the lane-interleaved layout the fill kernel reads (the two
// forms are described at Pk16Args).  The layout is [position][lane] and a lane owns its two pairs, so no
// transposition is needed: one wavefront per (unit, tile of 64 positions); every lane fetches 64 bytes of
// each of its pairs (8-byte loads; the blobs are readable up to 8 bytes past their end) and the wave
// writes one 256-byte line per position.  blockIdx.x walks the query tiles first, then the target tiles,
// four tiles per workgroup.
constexpr uint32_t kPackTile = 64;

struct PackArgs {
  const uint8_t* q;
  const uint64_t* q_off;
  const uint8_t* t;
  const uint64_t* t_off;
  uint32_t pair_base, pairs_in_pass;
  uint32_t q_cap, t_cap;     // positions per unit in qpk / tpk
  uint32_t q_tiles, t_tiles; // tiles of the query side / target side
  uint32_t* qpk;
  uint32_t* tpk;
  AlphabetInfo* alpha;
  uint32_t again;  // 1: the launch that repeats the packing in the symbols form if `violated` was raised
  // tables form: per-pair records of 2-bit codes for the traceback (AlignArgs::rec); NULL: none
  uint32_t* rec;
  uint64_t rec_unit_stride;
  uint32_t rec_stride;
};

__global__ __launch_bounds__(256) void pack_lane_pk16(PackArgs a) {
  __shared__ uint8_t code[256];
  if (a.again && !a.alpha->violated) return;
  const bool tables = a.alpha && a.alpha->small && !a.alpha->violated;
  if (tables) code[threadIdx.x] = a.alpha->code[threadIdx.x];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.q_tiles + a.t_tiles) return;
  const bool is_query = tile < a.q_tiles;
  const uint32_t units = (a.pairs_in_pass + 127) / 128;
  // gridDim.y is the number of units, or fewer for the repeat launch (which mostly has nothing to do)
  for (uint32_t unit = blockIdx.y; unit < units; unit += gridDim.y) {
  const uint32_t t0 = (is_query ? tile : tile - a.q_tiles) * kPackTile;
  const uint8_t* seq = is_query ? a.q : a.t;
  const uint64_t* off = is_query ? a.q_off : a.t_off;
  const uint32_t len_cap = is_query ? a.q_cap : a.t_cap;
  uint32_t* out = (is_query ? a.qpk : a.tpk) + ((uint64_t)unit * len_cap + t0) * 64 + lane;
  const uint32_t plA = unit * 128 + lane, plB = plA + 64;
  uint64_t offA = 0, offB = 0;
  uint32_t lenA = 0, lenB = 0;
  if (plA < a.pairs_in_pass) {
    offA = off[a.pair_base + plA];
    lenA = (uint32_t)(off[a.pair_base + plA + 1] - offA);
  }
  if (plB < a.pairs_in_pass) {
    offB = off[a.pair_base + plB];
    lenB = (uint32_t)(off[a.pair_base + plB + 1] - offB);
  }
  uint64_t wa[8], wb[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t pos = t0 + 8 * j;
    wa[j] = pos < lenA ? *reinterpret_cast<const uint64_t*>(seq + offA + pos) : 0;
    wb[j] = pos < lenB ? *reinterpret_cast<const uint64_t*>(seq + offB + pos) : 0;
  }
  const uint32_t pad = is_query ? kQueryPad : kTargetPad;
  uint32_t unseen = 0;
  if (!tables) {
    // the word index is unrolled (registers), the byte index is a loop: the compact form keeps more
    // wavefronts resident, which matters more here than the instruction count
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      uint64_t ra = wa[j], rb = wb[j];
#pragma unroll 1
      for (uint32_t c = 8 * j; c < 8 * j + 8; ++c, ra >>= 8, rb >>= 8) {
        const uint32_t pos = t0 + c;
        if (pos >= len_cap) break;
        const bool inA = pos < lenA, inB = pos < lenB;
        const uint32_t ba = (uint32_t)ra & 0xFFu, bb = (uint32_t)rb & 0xFFu;
        const uint32_t v = (inA ? ba : pad) | ((inB ? bb : pad) << 16);
        __builtin_nontemporal_store(v, &out[(uint64_t)c * 64]);  // read once, by the fill, much later
      }
    }
  } else {
    // one byte per position (nibbles: pair A, pair B), four positions per dword
    uint32_t* out4 = (is_query ? a.qpk : a.tpk) + ((uint64_t)unit * len_cap + t0 / 4) * 64 + lane;
    const uint32_t padn = is_query ? 0xFu : 0xCu, addb = is_query ? 0u : 4u;
    uint32_t reca[4] = {0, 0, 0, 0}, recb[4] = {0, 0, 0, 0};  // the tile's 64 codes of either pair, 2 bits each
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
      uint64_t ra = wa[j], rb = wb[j];
#pragma unroll 1
      for (uint32_t g = 0; g < 2; ++g) {
        const uint32_t c0 = 8 * j + 4 * g;
        if (t0 + c0 >= len_cap) break;
        uint32_t v = 0, ha = 0, hb = 0;
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x, ra >>= 8, rb >>= 8) {
          const uint32_t pos = t0 + c0 + x;
          const bool inA = pos < lenA, inB = pos < lenB;
          const uint32_t ca = inA ? (uint32_t)code[(uint32_t)ra & 0xFFu] : 0u;
          const uint32_t cb = inB ? (uint32_t)code[(uint32_t)rb & 0xFFu] : 0u;
          unseen |= ca | cb;
          const uint32_t na = inA ? (ca & 3u) : padn, nb = inB ? addb + (cb & 3u) : padn;
          v |= (na | (nb << 4)) << (8 * x);
          ha |= (ca & 3u) << (2 * x);
          hb |= (cb & 3u) << (2 * x);
        }
        __builtin_nontemporal_store(v, &out4[(uint64_t)(c0 / 4) * 64]);  // read once, by the fill, much later
        reca[j / 2] |= ha << (16 * (j & 1) + 8 * g);
        recb[j / 2] |= hb << (16 * (j & 1) + 8 * g);
      }
    }
    if (a.rec) {  // chunk of this tile: query chunks first, then the target's
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      uint32_t* r0 = a.rec + (uint64_t)unit * a.rec_unit_stride + (uint64_t)tile * 4;
      if (plA < a.pairs_in_pass)
        *reinterpret_cast<u32x4*>(r0 + (uint64_t)lane * a.rec_stride) = u32x4{reca[0], reca[1], reca[2], reca[3]};
      if (plB < a.pairs_in_pass)
        *reinterpret_cast<u32x4*>(r0 + (uint64_t)(lane + 64) * a.rec_stride) = u32x4{recb[0], recb[1], recb[2], recb[3]};
    }
  }
  if (unseen & kUnseen) a.alpha->violated = 1;
  }
}

}  // namespace sa

#endif
