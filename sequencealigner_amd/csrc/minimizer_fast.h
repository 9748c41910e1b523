// The MinimizeBinary state machine for window length 5 and a compile-time k-mer length K <= 15 (the
// mapper's defaults are k = 15, w = 5: reference src/mapper.cpp:467-471), written for the fewest vector
// instructions per window.  Same machine as Machine<5> in minimizer_kernels.h (reference
// src/minimizer_binary.cpp:99-133, 149-172); what differs is the number format:
//
//   * a k-mer value V lives left-justified in a 32-bit word, y = V << S with S = 32 - 2K >= 2, so the rolling
//     update y' = (y << 2) | (code << S) needs no mask (the shift drops the oldest base);
//   * the S >= 2 free low bits carry an index, so that the re-search of an expiring minimum (leftmost strict
//     minimum / maximum of 5 consecutive k-mers, :99-110) runs on plain unsigned min / max instructions:
//     pairs min(y[i] | 0, y[i+1] | 1), fours min(pair[i], pair[i+2] | 2) - on equal values the smaller tag, the
//     leftmost k-mer, wins - and the fifth k-mer joins with one compare (it wins only when strictly smaller:
//     tag 3 on its side).  The maximum side mirrors it with the tags complemented.  A pair or a four is
//     computed once and used by every window that contains it;
//   * positions are counted from the first step of the current block of 16 steps, so the position of a new
//     minimum, the position that expires and the base of a re-search are compile-time constants of the unrolled
//     block;
//   * the forward strand's state value keeps tag 1, the other strand's tag 0: the window's tuple (the smaller
//     of forward value and mask - other value, the other strand on a tie, :153-158, 167-172) is one unsigned
//     min whose low bit is the strand.
//
// The other strand's value of a k-mer is mask - V (every base complemented, same reading direction), so its
// minimum is the forward maximum; the identity needs every byte to be a base, which the caller checks on
// the bytes it feeds (Machine::feed4's `other`).
#ifndef SA_MINIMIZER_FAST_H
#define SA_MINIMIZER_FAST_H

#ifdef SA_FAST5_HOST  // host build for the CPU check (tests/cpp/test_fast5.cpp): plain C++ in place of the GPU forms
#include <cstdint>
#define __device__
#define __forceinline__ inline
#else
#include "minimizer_kernels.h"
#endif

namespace sa {

template <uint32_t K>
struct Fast5 {
  static_assert(K >= 4 && K <= 15, "two free tag bits below the value, the ring below one loop body");
  static constexpr uint32_t kK = K;
  static constexpr uint32_t S = 32 - 2 * K;      // value bits start here
  static constexpr uint32_t kTagMask = 3u;
  static constexpr uint32_t kMaskW = ((1u << (2 * K)) - 1u) << S;  // the k-mer mask, left-justified
  static constexpr uint32_t kD = 16 - K;         // slot of the oldest k-mer of a re-search, relative to the step's
  // "no tuple emitted yet" for lastp.  Positions count from the current block of 16 steps, so small negative
  // numbers are real positions (-1 = the k-mer of the step before the block: position << 1 | strand = ~0 for
  // the forward strand); this value would be position 2^30, which no step within a block can produce.
  static constexpr uint32_t kNoTuple = 0x80000000u;

  // k-mer values of the last 16 steps, slot = step & 15: clean (tag 0) and with tag 1
  uint32_t Y[16], B[16];
  // pairs: slot of the pair's first k-mer.  Pm = min(Y[i], B[i+1]) (tag: 1 = the right one),
  // Px2 = max(B[i], Y[i+1]) | 2 (tag bit 0: 1 = the left one; bit 1 set for use as the left pair of a four)
  uint32_t Pm[16], Px[16];
  uint32_t vf1, pf;  // forward strand: value (tag 1) and position of the running minimum
  uint32_t vc, pc;   // other strand: FORWARD value (tag 0) and position of its minimum (the forward maximum)
  uint32_t lastp;    // position << 1 | strand of the tuple emitted last

  // Takes over from a Machine whose next step brings in k-mer j0 (the newest complete k-mer is j0 - 1);
  // positions are relative to j0 from here on.  lastp_abs: position << 1 | strand of the tuple emitted last,
  // ~0 when there is none.
  template <class M>
  __device__ __forceinline__ void enter(const M& m, uint32_t j0, uint32_t lastp_abs) {
#pragma unroll
    for (uint32_t back = 0; back < 15; ++back) {
      const uint32_t y = m.value_back(back) << S;
      Y[15 - back] = y;
      B[15 - back] = y | 1u;
    }
    Y[0] = B[0] = 0;
#pragma unroll
    for (uint32_t s = 0; s < 16; ++s) Pm[s] = Px[s] = 0;
    // the pairs the first two steps find ready (step U computes the pair that starts at slot U + kD + 2)
#pragma unroll
    for (uint32_t x = 0; x < 2; ++x) {
      Pm[(kD + x) & 15] = umin(Y[(kD + x) & 15], B[(kD + x + 1) & 15]);
      Px[(kD + x) & 15] = umax(B[(kD + x) & 15], Y[(kD + x + 1) & 15]);
    }
    vf1 = (m.vf << S) | 1u;
    vc = m.vc << S;
    pf = m.pf - j0;
    pc = m.pc - j0;
    lastp = lastp_abs == ~0u ? kNoTuple : lastp_abs - 2 * j0;
  }

  __device__ static __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
  __device__ static __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

  // Positions are counted from the first step of the current block of 16 steps (so that every position a step
  // compares with or assigns is a small compile-time constant); rebase() moves them to the next block.
  __device__ __forceinline__ void rebase() {
    pf -= 16u;
    pc -= 16u;
    lastp -= 32u;
  }

  // One step.  U = number of the step within its block of 16 (compile time), cs = the new base's code << S.
  // Returns the window's tuple: *word = value << S | strand, *pos2 = position << 1 | strand (position relative
  // to the block's first step), and whether it differs from the tuple of the window before.
  // EMIT = false: the state advances, the tuple is not formed (*word, *pos2 and lastp are left alone).
  template <uint32_t U, bool EMIT>
  __device__ __forceinline__ bool step(uint32_t cs, uint32_t* word, uint32_t* pos2) {
    constexpr uint32_t s0 = (U + kD) & 15, s2 = (U + kD + 2) & 15, s3 = (U + kD + 3) & 15, s4 = (U + kD + 4) & 15;
    constexpr uint32_t prev = (U + 15) & 15;
    constexpr uint32_t base = U - K;                   // position of the first k-mer of the re-search range
    constexpr uint32_t expiring = U - K - 1u;
    // the new k-mer
    const uint32_t y = (Y[prev] << 2) | cs;
    const uint32_t b = y | 1u;
    Y[U] = y;
    B[U] = b;
    // the pair that starts two k-mers into the re-search range, then the range's first four
    const uint32_t pmn = umin(Y[s2], B[s3]);
    const uint32_t pxn = umax(B[s2], Y[s3]);
    Pm[s2] = pmn;
    Px[s2] = pxn;
    const uint32_t q = umin(Pm[s0], pmn | 2u);         // tag = offset 0..3 of the leftmost minimum
    const uint32_t qx = umax(Px[s0] | 2u, pxn);        // tag = 3 - offset of the leftmost maximum
    // The fifth k-mer joins: strictly smaller / strictly larger only (tag 3 on its side of the minimum, 0 of the
    // maximum).  Position of the result = base + offset, offset = tag (+ 1 when the fifth won: 3 + 1) for the
    // minimum, 3 - tag (+ 1 when the fifth won: tag 0) for the maximum: a compare whose result goes straight into
    // the carry of the add / subtract (the compiler would materialise it as 0 / 1 first).
    const uint32_t y4 = Y[s4];
    const uint32_t d4 = y4 | 3u;
    uint32_t rfw, rcw, tagf, tagc, pf_res, pc_res;
#ifdef SA_FAST5_HOST
    rfw = umin(q, d4);
    tagf = rfw & 3u;
    pf_res = base + tagf + (d4 < q ? 1u : 0u);
    rcw = umax(qx, y4);
    tagc = rcw & 3u;
    pc_res = (base + 4u) - tagc - (y4 <= qx ? 1u : 0u);
#else
    asm("v_cmp_lt_u32 vcc, %[d4], %[q]\n\t"          // V4 < the four's minimum
        "v_min_u32 %[rfw], %[q], %[d4]\n\t"
        "v_and_b32 %[tag], 3, %[rfw]\n\t"
        "v_addc_co_u32 %[res], vcc, %[base], %[tag], vcc"
        : [rfw] "=&v"(rfw), [tag] "=&v"(tagf), [res] "=v"(pf_res)
        : [d4] "v"(d4), [q] "v"(q), [base] "i"((int)base)
        : "vcc");
    asm("v_cmp_le_u32 vcc, %[y4], %[qx]\n\t"         // not (V4 > the four's maximum)
        "v_max_u32 %[rcw], %[qx], %[y4]\n\t"
        "v_and_b32 %[tag], 3, %[rcw]\n\t"
        "v_subb_co_u32 %[res], vcc, %[base4], %[tag], vcc"
        : [rcw] "=&v"(rcw), [tag] "=&v"(tagc), [res] "=v"(pc_res)
        : [y4] "v"(y4), [qx] "v"(qx), [base4] "i"((int)base + 4)
        : "vcc");
#endif
    uint32_t rf1 = (rfw & ~kTagMask) | 1u;
    uint32_t rc = rcw & ~kTagMask;
    // the state machine, both strands (src/minimizer_binary.cpp:112-133): an expiring minimum takes the
    // re-search result, otherwise the new k-mer replaces a strictly larger minimum
#ifdef SA_FAST5_HOST
    const bool ef = pf == expiring, ec = pc == expiring;
    const bool lf = b < vf1, lc = y > vc;              // V[j] < forward minimum; V[j] > forward maximum
    const uint32_t nvf = umin(vf1, b), nvc = umax(vc, y);
    pf = ef ? pf_res : (lf ? U : pf);
    vf1 = ef ? rf1 : nvf;
    pc = ec ? pc_res : (lc ? U : pc);
    vc = ec ? rc : nvc;
#else
    // (the two strands interleaved: an instruction that reads a compare's mask needs two others between itself
    // and the compare, and the compiler fills those slots with s_nop when it orders the strands one after the other)
    uint32_t nvf, nvc, tpf, tpc;
    uint64_t mask;
    asm("v_cmp_ge_u32 vcc, %[b], %[vf]\n\t"            // not (V[j] < forward minimum)
        "v_cmp_le_u32 %[m], %[y], %[vc]\n\t"           // not (V[j] > forward maximum)
        "v_min_u32 %[nvf], %[vf], %[b]\n\t"
        "v_max_u32 %[nvc], %[vc], %[y]\n\t"
        "v_cndmask_b32 %[tpf], %[u], %[pf], vcc\n\t"
        "v_cndmask_b32 %[tpc], %[u], %[pc], %[m]\n\t"
        "v_cmp_eq_u32 vcc, %[ex], %[pf]\n\t"           // the forward minimum expires
        "v_cmp_eq_u32 %[m], %[ex], %[pc]\n\t"          // the other strand's expires
        "s_nop 0\n\t"
        "v_cndmask_b32 %[vf], %[nvf], %[rf1], vcc\n\t"
        "v_cndmask_b32 %[pf], %[tpf], %[pfr], vcc\n\t"
        "v_cndmask_b32 %[vc], %[nvc], %[rc], %[m]\n\t"
        "v_cndmask_b32 %[pc], %[tpc], %[pcr], %[m]"
        : [vf] "+v"(vf1), [pf] "+v"(pf), [vc] "+v"(vc), [pc] "+v"(pc), [nvf] "=&v"(nvf), [nvc] "=&v"(nvc),
          [tpf] "=&v"(tpf), [tpc] "=&v"(tpc), [m] "=&s"(mask)
        : [b] "v"(b), [y] "v"(y), [u] "i"((int)U), [ex] "i"((int)expiring), [rf1] "v"(rf1), [pfr] "v"(pf_res),
          [rc] "v"(rc), [pcr] "v"(pc_res)
        : "vcc");
#endif
    if (!EMIT) return false;
    // the window's tuple (:153-158, 167-172): forward when its value is strictly smaller, position << 1 | strand
    const uint32_t cw = kMaskW - vc;                   // the other strand's own value, tag 0
    uint32_t wd, sel, p2;
#ifdef SA_FAST5_HOST
    wd = umin(vf1, cw);
    sel = vf1 < cw ? pf : pc;
    p2 = sel + sel + (vf1 < cw ? 1u : 0u);
#else
    asm("v_cmp_lt_u32 vcc, %[vf], %[cw]\n\t"
        "v_min_u32 %[wd], %[vf], %[cw]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[sel], %[pc], %[pf], vcc\n\t"
        "v_addc_co_u32 %[p2], vcc, %[sel], %[sel], vcc"
        : [wd] "=&v"(wd), [sel] "=&v"(sel), [p2] "=v"(p2)
        : [vf] "v"(vf1), [cw] "v"(cw), [pf] "v"(pf), [pc] "v"(pc)
        : "vcc");
#endif
    *word = wd;
    *pos2 = p2;
    const bool fresh = p2 != lastp;
    lastp = p2;
    return fresh;
  }
};

}  // namespace sa

#endif
