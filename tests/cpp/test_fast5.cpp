// CPU check of sa::Fast5 (sequencealigner_amd/csrc/minimizer_fast.h), the register-level form of the
// MinimizeBinary state machine for w = 5: stepped beside a plain restatement of the reference's machine
// (src/minimizer_binary.cpp:99-133, 149-172) on random and repetitive sequences, entered at every kind of
// position, compared window by window: both strands' positions and values, the emitted tuple, the "tuple
// changed" flag.  Built by g++ (the header's device code has a plain C++ form for the host).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define SA_FAST5_HOST 1
#include "minimizer_fast.h"

namespace {

constexpr uint32_t K = 15, W = 5;
constexpr uint32_t kMask = (1u << (2 * K)) - 1u;

struct Ref {  // the reference's machine on the array of k-mer values
  const std::vector<uint32_t>& V;
  uint32_t pf, pc, j;
  explicit Ref(const std::vector<uint32_t>& v) : V(v), pf(0), pc(0), j(0) {}
  uint32_t cv(uint32_t i) const { return kMask - V[i]; }
  void first_window() {
    pf = pc = 0;
    for (uint32_t i = 1; i < W; ++i) {
      if (V[i] < V[pf]) pf = i;
      if (cv(i) < cv(pc)) pc = i;
    }
    j = W;
  }
  void step() {
    const uint32_t expiring = j - K - 1;
    if (pf == expiring) {
      uint32_t q = pf + 1;
      for (uint32_t i = pf + 2; i <= pf + W; ++i)
        if (V[i] < V[q]) q = i;
      pf = q;
    } else if (V[j] < V[pf]) {
      pf = j;
    }
    if (pc == expiring) {
      uint32_t q = pc + 1;
      for (uint32_t i = pc + 2; i <= pc + W; ++i)
        if (cv(i) < cv(q)) q = i;
      pc = q;
    } else if (cv(j) < cv(pc)) {
      pc = j;
    }
    ++j;
  }
  void tuple(uint32_t* value, uint32_t* pos2) const {
    const bool f = V[pf] < cv(pc);
    *value = f ? V[pf] : cv(pc);
    *pos2 = f ? (pf << 1 | 1u) : (pc << 1);
  }
};

struct HostMachine {  // what Fast5::enter reads from the machine it takes over from
  const std::vector<uint32_t>& V;
  uint32_t newest;  // index of the newest complete k-mer
  uint32_t vf, vc, pf, pc;
  uint32_t value_back(uint32_t back) const { return back <= newest ? V[newest - back] : 0u; }
};

// quiet_blocks: that many blocks of 16 steps run without forming tuples (as the kernel runs the windows before
// every lane's chunk); lastp is then set to "no tuple yet" and the first tuple formed must count as new.
int check(const std::vector<uint8_t>& codes, uint32_t enter_at, uint32_t quiet_blocks, uint64_t* steps_done) {
  const uint32_t n = (uint32_t)codes.size();
  std::vector<uint32_t> V(n - K + 1);
  uint32_t x = 0;
  for (uint32_t i = 0; i < n; ++i) {
    x = ((x << 2) | codes[i]) & kMask;
    if (i + 1 >= K) V[i + 1 - K] = x;
  }
  Ref ref(V);
  ref.first_window();
  while (ref.j < enter_at) ref.step();
  // take over before step j0 = ref.j
  const uint32_t j0 = ref.j;
  HostMachine hm{V, j0 - 1, V[ref.pf], V[ref.pc], ref.pf, ref.pc};
  uint32_t lv, lp;
  ref.tuple(&lv, &lp);
  sa::Fast5<K> fm;
  fm.enter(hm, j0, lp);
  uint32_t last_p2 = lp;
  const uint32_t total = (uint32_t)V.size() - j0;
  bool first_tuple = false;
  for (uint32_t r0 = 0; r0 + 16 <= total; r0 += 16) {
    const bool quiet = r0 / 16 < quiet_blocks;
    if (!quiet && quiet_blocks && r0 / 16 == quiet_blocks) {
      fm.lastp = sa::Fast5<K>::kNoTuple;
      first_tuple = true;
    }
    auto one = [&](auto u_tag) -> bool {
      constexpr uint32_t U = decltype(u_tag)::value;
      const uint32_t j = j0 + r0 + U;
      const uint32_t cs = (uint32_t)codes[j + K - 1] << sa::Fast5<K>::S;
      uint32_t word = 0, p2 = 0;
      if (quiet) {
        fm.template step<U, false>(cs, &word, &p2);
        ref.step();
        uint32_t rv, rp;
        ref.tuple(&rv, &rp);
        last_p2 = rp;
        return fm.pf + j0 + r0 == ref.pf && fm.pc + j0 + r0 == ref.pc;
      }
      const bool fresh = fm.template step<U, true>(cs, &word, &p2);
      ref.step();
      uint32_t rv, rp;
      ref.tuple(&rv, &rp);
      if (first_tuple) {  // after the quiet blocks the tuple before is unknown: the first one must be reported
        first_tuple = false;
        if (!fresh) {
          std::printf("first tuple after %u quiet blocks not reported as new (k-mer %u)\n", quiet_blocks, j);
          return false;
        }
        last_p2 = ~rp;
      }
      const uint32_t got_p2 = p2 + 2 * (j0 + r0);
      const bool ok = (word >> sa::Fast5<K>::S) == rv && got_p2 == rp && (word & 1u) == (rp & 1u) &&
                      fm.pf + j0 + r0 == ref.pf && fm.pc + j0 + r0 == ref.pc &&
                      (fm.vf1 >> sa::Fast5<K>::S) == V[ref.pf] && (fm.vc >> sa::Fast5<K>::S) == V[ref.pc] &&
                      fresh == (rp != last_p2);
      if (!ok) {
        std::printf("mismatch at k-mer %u (entered at %u): value %u/%u pos2 %u/%u pf %u/%u pc %u/%u fresh %d/%d\n", j, j0,
                    word >> sa::Fast5<K>::S, rv, got_p2, rp, fm.pf + j0 + r0, ref.pf, fm.pc + j0 + r0, ref.pc, (int)fresh,
                    (int)(rp != last_p2));
        return false;
      }
      last_p2 = rp;
      ++*steps_done;
      return true;
    };
#define STEP(X) if (!one(std::integral_constant<uint32_t, X>{})) return 1;
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13)
    STEP(14) STEP(15)
#undef STEP
    fm.rebase();
  }
  return 0;
}

}  // namespace

int main() {
  std::mt19937_64 rng(12345);
  uint64_t steps = 0;
  for (int trial = 0; trial < 2000; ++trial) {
    const uint32_t n = 600 + (uint32_t)(rng() % 3000);
    std::vector<uint8_t> codes(n);
    const int kind = trial % 5;
    if (kind == 0) {  // short tandem repeats and constant stretches: equal values inside a window
      const uint32_t period = 1 + (uint32_t)(rng() % 6);
      std::vector<uint8_t> unit(period);
      for (auto& c : unit) c = rng() & 3;
      for (uint32_t i = 0; i < n; ++i) codes[i] = (rng() % 97 == 0) ? (uint8_t)(rng() & 3) : unit[i % period];
    } else if (kind == 1) {  // two-letter alphabet: many ties
      for (auto& c : codes) c = (rng() & 1) ? 0 : 3;
    } else {
      for (auto& c : codes) c = rng() & 3;
    }
    const uint32_t enter_at = (trial % 3 == 0) ? W : W + (uint32_t)(rng() % 300);
    if (check(codes, enter_at, (trial % 2) ? 1 + (uint32_t)(rng() % 12) : 0, &steps)) return 1;
  }
  std::printf("fast5 ok: %llu steps compared\n", (unsigned long long)steps);
  return 0;
}
