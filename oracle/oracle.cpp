/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement ("oracle") of the reference's alignment and minimizer
 * algorithms.  It is the checker for the GPU path and the optional timed CPU comparison; it is never
 * linked into, loaded by or called from the shipped library.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_golden.py checks every function here against the five
 * known-answer tests of the reference (test/testAlignmentMinimizer.cpp:7-82) and against fixtures
 * under tests/golden/ that were produced by the reference's own sources compiled unmodified
 * (oracle/_ref, recipe in oracle/Makefile, generator tests/golden/make_golden.py).
 *
 * Each function cites the reference lines it restates.  Behaviour the reference leaves undefined
 * (reads of uninitialised variables / out-of-range indices) is given a fixed meaning here, listed at
 * the function; those inputs are outside every parity claim.
 */
#include "sa_oracle.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

enum Edge : uint32_t { kUp = 0, kLeft = 1, kDiag = 2, kNone = 3 };  // alignment.cpp:9

// 8-byte cell, same footprint as the reference's struct matrix (alignment.cpp:11-14) so that timing
// this code is an honest stand-in for the reference's memory behaviour.
struct Cell {
  int32_t score;
  uint32_t edge;
};

// alignment.cpp:17-33.  Input ops are in traceback (reverse) order; output is forward-order RLE,
// count always printed.  Empty input yields "" here (the reference reads uCigar[-1]).
std::string compress_reverse_ops(const std::string& rev) {
  std::string out;
  if (rev.empty()) return out;
  size_t i = rev.size();
  while (i > 0) {
    char op = rev[i - 1];
    size_t run = 0;
    while (i > 0 && rev[i - 1] == op) {
      --i;
      ++run;
    }
    out += std::to_string(run);
    out += op;
  }
  return out;
}

// alignment.cpp:129-136 and 342-348: the value stored in *target_begin for Global and Semiglobal is
// 1, advanced by 2 while the character at that index of the compressed CIGAR is 'D'.  Indexing past
// the end of the string is treated as "not D" here.
uint32_t odd_target_begin(const std::string& cigar) {
  uint32_t c = 1;
  while (c < cigar.size() && cigar[c] == 'D') c += 2;
  return c;
}

struct AlignOut {
  int score;
  std::string cigar;
  uint32_t target_begin;
};

// Shared fill: rows = target, columns = query (alignment.cpp:60-95, 173-209, 278-308).
// mode 0 Global, 1 Local, 2 Semiglobal.
AlignOut align_full(const char* q, uint32_t Q, const char* t, uint32_t T, int mode, int match,
                    int mismatch, int gap, bool want_cigar) {
  const size_t W = (size_t)Q + 1;
  std::vector<std::vector<Cell>> H(T + 1, std::vector<Cell>(W));

  // borders: alignment.cpp:263-269 (Global), 152-168 (Local), 46-52 (Semiglobal)
  const bool local_zero = (mode == 1 && gap < 0);
  for (uint32_t i = 1; i <= T; ++i) {
    if (mode == 0 || (mode == 1 && !local_zero))
      H[i][0] = {gap * (int)i, kUp};
    else
      H[i][0] = {0, kNone};
  }
  for (uint32_t r = 1; r <= Q; ++r) {
    if (local_zero)
      H[0][r] = {0, kNone};
    else
      H[0][r] = {gap * (int)r, kLeft};
  }
  H[0][0] = {0, kNone};

  int best = 0;            // alignment.cpp:56, 171 (maxOverall = 0)
  uint32_t best_row = 0;   // uninitialised in the reference; 0 here (outside parity when unused)
  uint32_t best_col = 0;

  for (uint32_t i = 1; i <= T; ++i) {
    const std::vector<Cell>& up = H[i - 1];
    std::vector<Cell>& cur = H[i];
    for (uint32_t r = 1; r <= Q; ++r) {
      const int sub = (t[i - 1] == q[r - 1]) ? match : mismatch;
      const int d = up[r - 1].score + sub;
      const int u = up[r].score + gap;
      const int l = cur[r - 1].score + gap;
      const int m = std::max(std::max(d, u), l);
      // tie-break Diag > Left > Up (alignment.cpp:77-85, 193-201, 295-303)
      uint32_t e = (m == d) ? kDiag : (m == l) ? kLeft : kUp;
      if (mode == 1) {
        if (m <= 0)  // alignment.cpp:190-192
          cur[r] = {0, kNone};
        else
          cur[r] = {m, e};
        if (m > best) {  // alignment.cpp:203-207, first strict maximum in row-major order
          best = m;
          best_row = i;
          best_col = r;
        }
      } else {
        cur[r] = {m, e};
        if (mode == 2 && r == Q && m > best) {  // alignment.cpp:87-92
          best = m;
          best_row = i;
        }
      }
    }
  }

  uint32_t row, col;
  if (mode == 0) {
    row = T;
    col = Q;
  } else if (mode == 1) {
    row = best_row;
    col = best_col;
  } else {
    row = best_row;
    col = Q;
  }
  const uint32_t end_row = row;

  // traceback (alignment.cpp:98-121, 212-236, 311-334); M iff score - match == diagonal score
  std::string rev;
  auto keep_going = [&]() {
    if (mode == 0) return row != 0 || col != 0;
    if (mode == 1) return H[row][col].edge != kNone;
    return col != 0;
  };
  while (keep_going()) {
    const Cell& c = H[row][col];
    if (c.edge == kDiag) {
      rev += (c.score - match == H[row - 1][col - 1].score) ? 'M' : 'X';
      --row;
      --col;
    } else if (c.edge == kLeft) {
      rev += 'I';
      --col;
    } else if (c.edge == kUp) {
      rev += 'D';
      --row;
    } else {
      break;  // a None edge outside Local: only reachable for inputs outside the parity envelope
    }
  }

  AlignOut out;
  if (want_cigar) out.cigar = compress_reverse_ops(rev);
  if (mode == 1)
    out.target_begin = row;  // alignment.cpp:244-246
  else
    out.target_begin = odd_target_begin(out.cigar);
  if (mode == 0)
    out.score = H[T][Q].score;  // alignment.cpp:350
  else if (mode == 1)
    out.score = best;  // alignment.cpp:248
  else
    out.score = H[end_row][Q].score;  // alignment.cpp:138
  return out;
}

// The score alone in linear memory: the same borders, recurrence and end-cell rules as align_full
// (alignment.cpp:46-52, 60-95, 152-209, 263-308), but only two rows of the matrix are kept, so pairs far
// beyond what the 8-byte-per-cell matrix allows (100 kbp x 100 kbp = 80 GB) can be scored on the CPU.
// Checked against the compiled reference on pairs up to 5 kbp (tests/test_oracle_vs_reference.py).
int score_linear(const char* q, uint32_t Q, const char* t, uint32_t T, int mode, int match, int mismatch,
                 int gap) {
  const bool local_zero = (mode == 1 && gap < 0);
  std::vector<int32_t> prev((size_t)Q + 1), cur((size_t)Q + 1), tmp((size_t)Q + 1);
  prev[0] = 0;
  for (uint32_t r = 1; r <= Q; ++r) prev[r] = local_zero ? 0 : gap * (int)r;
  int best = 0;
  int best_last = 0;  // Semiglobal: first strict maximum > 0 of the last column; H[0][Q] without one
  bool have_last = false;
  for (uint32_t i = 1; i <= T; ++i) {
    cur[0] = (mode == 0 || (mode == 1 && !local_zero)) ? gap * (int)i : 0;
    const char ti = t[i - 1];
    // max(diag, up) has no dependency along the row (the compiler vectorises this loop) ...
    for (uint32_t r = 1; r <= Q; ++r)
      tmp[r] = std::max(prev[r - 1] + (ti == q[r - 1] ? match : mismatch), prev[r] + gap);
    // ... the left candidate is the sequential part
    int left = cur[0];
    if (mode == 1) {
      for (uint32_t r = 1; r <= Q; ++r) {
        int m = std::max(tmp[r], left + gap);
        if (m > best) best = m;  // alignment.cpp:203-207 (the score of the first strict maximum is the maximum)
        if (m <= 0) m = 0;       // alignment.cpp:190-192
        cur[r] = left = m;
      }
    } else {
      for (uint32_t r = 1; r <= Q; ++r) cur[r] = left = std::max(tmp[r], left + gap);
      if (mode == 2 && cur[Q] > best_last) {  // alignment.cpp:87-92
        best_last = cur[Q];
        have_last = true;
      }
    }
    prev.swap(cur);
  }
  if (mode == 0) return prev[Q];
  if (mode == 1) return best;
  return have_last ? best_last : (local_zero ? 0 : gap * (int)Q);  // row 0 when nothing is positive (as align_full)
}

// ---- minimizers ------------------------------------------------------------------------------

// minimizer_binary.cpp:13-50.  forward: C0 A1 T2 G3; other strand: 3 - forward; else -1.
int base_code(char c, bool forward) {
  int f;
  switch (c) {
    case 'c': case 'C': f = 0; break;
    case 'a': case 'A': f = 1; break;
    case 't': case 'T': f = 2; break;
    case 'g': case 'G': f = 3; break;
    default: return -1;
  }
  return forward ? f : 3 - f;
}

// minimizer_binary.cpp:53-97: V[0] unmasked, V[j>0] = ((V[j-1] << 2) | code) & mask.
std::vector<uint32_t> kmer_values(const char* s, uint32_t n, uint32_t k, bool forward) {
  std::vector<uint32_t> v;
  uint32_t mask = 3;
  for (uint32_t i = 1; i < k; ++i) mask = (mask << 2) | 3;  // :71-78
  uint32_t cur = 0;
  for (uint32_t i = 0; i < k; ++i) cur = (cur << 2) | (uint32_t)base_code(s[i], forward);
  v.push_back(cur);
  for (uint32_t j = 1; j + k <= n; ++j) {
    cur = ((cur << 2) | (uint32_t)base_code(s[j + k - 1], forward)) & mask;
    v.push_back(cur);
  }
  return v;
}

// minimizer_binary.cpp:99-110: leftmost strict minimum of V[start .. start+w-1] (unsigned compare).
uint32_t leftmost_min(const std::vector<uint32_t>& v, uint32_t start, uint32_t w) {
  uint32_t p = start;
  for (uint32_t i = start + 1; i < start + w; ++i)
    if (v[i] < v[p]) p = i;
  return p;
}

struct Triple {
  uint32_t value, pos, strand;
  bool operator<(const Triple& o) const {
    if (value != o.value) return value < o.value;
    if (pos != o.pos) return pos < o.pos;
    return strand < o.strand;
  }
  bool operator==(const Triple& o) const {
    return value == o.value && pos == o.pos && strand == o.strand;
  }
};

// utility.cpp:8-26
std::vector<Triple> sort_unique(std::vector<Triple>& v) {
  std::sort(v.begin(), v.end());
  std::vector<Triple> out;
  for (const Triple& x : v)
    if (out.empty() || !(out.back() == x)) out.push_back(x);
  return out;
}

// minimizer_binary.cpp:112-178.  Per strand the running minimum is a state machine on its position
// p: stepping to new k-mer index j, if j - k - 1 == p the minimum is re-searched over the w k-mers
// V[p+1 .. p+w] (:120-123), otherwise it is replaced only when V[j] < V[p] (:126-131).
std::vector<Triple> minimize_binary(const char* s, uint32_t n, uint32_t k, uint32_t w) {
  std::vector<uint32_t> vf = kmer_values(s, n, k, true);
  std::vector<uint32_t> vc = kmer_values(s, n, k, false);
  std::vector<Triple> picks;
  uint32_t pf = leftmost_min(vf, 0, w);  // :149
  uint32_t pc = leftmost_min(vc, 0, w);  // :150
  auto emit = [&]() {
    if (vf[pf] < vc[pc])  // :153-158, :167-172 (tie -> other strand)
      picks.push_back({vf[pf], pf, 1});
    else
      picks.push_back({vc[pc], pc, 0});
  };
  emit();
  const uint32_t windows = n - w - k + 2;  // :161
  for (uint32_t i = 1; i < windows; ++i) {
    const uint32_t j = i + w - 1;
    if (j - k - 1 == pf)  // evaluated in unsigned arithmetic, as in the reference
      pf = leftmost_min(vf, pf + 1, w);
    else if (vf[j] < vf[pf])
      pf = j;
    if (j - k - 1 == pc)
      pc = leftmost_min(vc, pc + 1, w);
    else if (vc[j] < vc[pc])
      pc = j;
    emit();
  }
  return sort_unique(picks);
}

// minimizer.cpp:124-214.  k-mer value = the k base codes read as a DECIMAL number (stoi, :74,77,91,
// 114).  Returns false where stoi would throw (value > INT_MAX).
bool decimal_value(const std::vector<int>& digits, size_t start, uint32_t k, int* out) {
  int64_t v = 0;
  for (uint32_t i = 0; i < k; ++i) {
    v = v * 10 + digits[start + i];
    if (v > 2147483647LL) return false;
  }
  *out = (int)v;
  return true;
}

bool minimize_decimal(const char* s, uint32_t n, uint32_t k, uint32_t w, std::vector<Triple>* res) {
  // minimizer.cpp:13-69: characters other than ACGTacgt are dropped.
  std::vector<int> df, dc;
  for (uint32_t i = 0; i < n; ++i) {
    int f = base_code(s[i], true);
    if (f < 0) continue;
    df.push_back(f);
    dc.push_back(3 - f);
  }
  const uint32_t L = w + k - 1;  // :137
  if (df.size() < L || k == 0 || w == 0) return false;
  std::vector<Triple> picks;

  struct Strand {
    const std::vector<int>* d;
    int pre_min = 0, pre_idx = 0;  // pre_idx is uninitialised in the reference (:132-133); 0 here
    int cur_min = 0, cur_idx = 0;
  } st[2];
  st[0].d = &df;
  st[1].d = &dc;

  for (Strand& x : st) {  // initWindow, :71-84
    if (!decimal_value(*x.d, 0, k, &x.pre_min)) return false;
    for (uint32_t i = 1; i < w; ++i) {
      int v;
      if (!decimal_value(*x.d, i, k, &v)) return false;
      if (x.pre_min > v) {
        x.pre_min = v;
        x.pre_idx = (int)i;
      }
    }
    x.cur_idx = x.pre_idx;
  }
  if (st[0].pre_min < st[1].pre_min)  // :153-160
    picks.push_back({(uint32_t)st[0].pre_min, (uint32_t)st[0].pre_idx, 1});
  else
    picks.push_back({(uint32_t)st[1].pre_min, (uint32_t)st[1].pre_idx, 0});

  const uint32_t windows = n - L + 1;  // :163, unsigned arithmetic
  for (uint32_t i = 1; i < windows; ++i) {
    if ((size_t)i + L > df.size()) return false;  // substr/stoi would throw: outside the envelope
    for (Strand& x : st) {
      if (x.pre_idx != (int)i - 1) {  // compareLastKmer, :86-105
        int v;
        if (!decimal_value(*x.d, i + w - 1, k, &v)) return false;
        if (v < x.pre_min) {
          x.cur_min = v;
          x.cur_idx = (int)(i + w - 1);
        } else {
          x.cur_min = x.pre_min;
          x.cur_idx = x.pre_idx;
        }
      } else {  // compareKmer, :107-122 - sentinel 444, index kept when nothing is below it
        x.cur_min = 444;
        for (uint32_t r = 0; r < w; ++r) {
          int v;
          if (!decimal_value(*x.d, i + r, k, &v)) return false;
          if (x.cur_min > v) {
            x.cur_min = v;
            x.cur_idx = (int)(i + r);
          }
        }
      }
    }
    if (st[0].cur_min < st[1].cur_min)  // :189-196
      picks.push_back({(uint32_t)st[0].cur_min, (uint32_t)st[0].cur_idx, 1});
    else
      picks.push_back({(uint32_t)st[1].cur_min, (uint32_t)st[1].cur_idx, 0});
    for (Strand& x : st) {  // :200-204
      x.pre_min = x.cur_min;
      x.pre_idx = x.cur_idx;
    }
  }
  *res = sort_unique(picks);
  return true;
}

size_t copy_out(const std::vector<Triple>& v, uint32_t* out, size_t cap) {
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    out[3 * i] = v[i].value;
    out[3 * i + 1] = v[i].pos;
    out[3 * i + 2] = v[i].strand;
  }
  return v.size();
}

template <class F>
void task_farm(size_t n, int threads, F f) {
  if (threads < 1) threads = 1;
  std::atomic<size_t> next(0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n) break;
        f(i);
      }
    });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int sa_cpu_align(const char* query, uint32_t query_len, const char* target, uint32_t target_len,
                 int type, int match, int mismatch, int gap, char* cigar_buf, size_t cigar_cap,
                 size_t* cigar_len, uint32_t* target_begin) {
  if (type < 0 || type > 2) return 9999;  // alignment.cpp:371-373
  AlignOut r = align_full(query, query_len, target, target_len, type, match, mismatch, gap,
                          cigar_buf != nullptr);
  if (cigar_buf) {
    if (cigar_cap) {
      size_t c = std::min(cigar_cap - 1, r.cigar.size());
      std::memcpy(cigar_buf, r.cigar.data(), c);
      cigar_buf[c] = 0;
    }
    if (cigar_len) *cigar_len = r.cigar.size();
  }
  if (target_begin) *target_begin = r.target_begin;
  return r.score;
}

int sa_cpu_score_linear(const char* query, uint32_t query_len, const char* target, uint32_t target_len,
                        int type, int match, int mismatch, int gap) {
  if (type < 0 || type > 2) return 9999;
  return score_linear(query, query_len, target, target_len, type, match, mismatch, gap);
}

int sa_cpu_score_linear_batch(const char* q_blob, const uint64_t* q_off, const char* t_blob,
                              const uint64_t* t_off, size_t n, int type, int match, int mismatch, int gap,
                              int threads, int32_t* scores) {
  task_farm(n, threads, [&](size_t i) {
    scores[i] = sa_cpu_score_linear(q_blob + q_off[i], (uint32_t)(q_off[i + 1] - q_off[i]), t_blob + t_off[i],
                                    (uint32_t)(t_off[i + 1] - t_off[i]), type, match, mismatch, gap);
  });
  return 0;
}

size_t sa_cpu_minimize_binary(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                              uint32_t window_len, uint32_t* out_triples, size_t cap) {
  return copy_out(minimize_binary(sequence, sequence_len, kmer_len, window_len), out_triples, cap);
}

size_t sa_cpu_minimize_decimal(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                               uint32_t window_len, uint32_t* out_triples, size_t cap) {
  std::vector<Triple> r;
  if (!minimize_decimal(sequence, sequence_len, kmer_len, window_len, &r)) return (size_t)-1;
  return copy_out(r, out_triples, cap);
}

size_t sa_cpu_remove_duplicate(uint32_t* triples, size_t n, uint32_t* out_triples) {
  std::vector<Triple> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = {triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]};
  std::vector<Triple> u = sort_unique(v);
  copy_out(v, triples, n);
  return copy_out(u, out_triples, u.size());
}

int sa_cpu_align_batch(const char* q_blob, const uint64_t* q_off, const char* t_blob,
                       const uint64_t* t_off, size_t n, int type, int match, int mismatch, int gap,
                       int want_cigar, int threads, int32_t* scores, uint32_t* target_begin) {
  task_farm(n, threads, [&](size_t i) {
    std::vector<char> buf;
    size_t len = 0;
    uint32_t Q = (uint32_t)(q_off[i + 1] - q_off[i]), T = (uint32_t)(t_off[i + 1] - t_off[i]);
    if (want_cigar) buf.resize(2 * ((size_t)Q + T) + 16);
    uint32_t tb = 0;
    scores[i] = sa_cpu_align(q_blob + q_off[i], Q, t_blob + t_off[i], T, type, match, mismatch, gap,
                             want_cigar ? buf.data() : nullptr, buf.size(), &len, &tb);
    if (target_begin) target_begin[i] = tb;
  });
  return 0;
}

int sa_cpu_minimize_batch(const char* blob, const uint64_t* off, const uint32_t* len, size_t n,
                          uint32_t kmer_len, uint32_t window_len, int threads, uint64_t* counts,
                          uint64_t* hashes) {
  task_farm(n, threads, [&](size_t i) {
    std::vector<Triple> list = minimize_binary(blob + off[i], len[i], kmer_len, window_len);
    counts[i] = list.size();
    if (hashes) {
      uint64_t h = 0;
      for (size_t x = 0; x < list.size(); ++x)
        h += sa_triple_hash(list[x].value, list[x].pos, list[x].strand ? 1u : 0u, x);
      hashes[i] = h;
    }
  });
  return 0;
}

// mapper.cpp:371-399: digits accumulate into `value`; the first non-digit after a number adds it to the
// block length; an 'm' adds the current value to the match length.
void sa_cpu_cigar_block_match(const char* cigar, int* abl, int* ml) {
  int value = 0, sum = 0, match = 0;
  bool in_number = false;
  for (const char* c = cigar; *c; ++c) {
    if (*c >= '0' && *c <= '9') {
      value = in_number ? value * 10 + (*c - '0') : (*c - '0');
      in_number = true;
    } else {
      if (in_number) {
        sum += value;
        in_number = false;
      }
      if (*c == 'm') match += value;
    }
  }
  *abl = sum;
  *ml = match;
}

namespace {
using SeedIndex = std::unordered_map<uint32_t, std::vector<std::pair<uint32_t, bool>>>;

// mapper.cpp:172-196: append (position, strand) to the value's list, in list order
SeedIndex build_index(const uint32_t* triples, size_t n) {
  SeedIndex idx;
  for (size_t x = 0; x < n; ++x) idx[triples[3 * x]].emplace_back(triples[3 * x + 1], triples[3 * x + 2] != 0);
  return idx;
}

// mapper.cpp:198-205, 224-232
void drop_frequent(SeedIndex& idx, float f) {
  std::vector<std::tuple<int, uint32_t>> occ;
  for (auto& kv : idx) occ.emplace_back((int)kv.second.size(), kv.first);
  std::sort(occ.begin(), occ.end());
  const int index = (int)occ.size() * f;  // (int)size, times a float, truncated
  const int from = (int)occ.size() - index;
  for (int i = from; i < (int)occ.size(); ++i) idx.erase(std::get<1>(occ[i]));
}

// mapper.cpp:234-290
std::vector<std::tuple<uint32_t, uint32_t>> match_seeds(const SeedIndex& ref, const SeedIndex& frag) {
  std::vector<std::tuple<uint32_t, uint32_t>> match;
  for (auto& kv : frag) {
    auto hit = ref.find(kv.first);
    if (hit == ref.end()) continue;
    std::vector<uint32_t> fo, fr, ro, rr;
    for (auto& e : kv.second) (e.second ? fo : fr).push_back(e.first);
    for (auto& e : hit->second) (e.second ? ro : rr).push_back(e.first);
    std::sort(ro.begin(), ro.end());
    std::sort(rr.begin(), rr.end());
    for (uint32_t a : fo)
      for (uint32_t b : ro) match.emplace_back(a, b);
    for (uint32_t a : fr)
      for (uint32_t b : rr) match.emplace_back(a, b);
  }
  std::sort(match.begin(), match.end());
  return match;
}

// mapper.cpp:292-352
int chain_lis(const std::vector<std::tuple<uint32_t, uint32_t>>& m, int& t_begin, int& t_end, int& q_begin,
              int& q_end) {
  const int n = (int)m.size();
  const int INF = 1000000000;
  if (n == 0) return 0;
  if (n == 1) {
    t_begin = t_end = (int)std::get<1>(m[0]);
    q_begin = q_end = (int)std::get<0>(m[0]);  // (q_end: see the header)
    return 1;
  }
  std::vector<int> parent(n + 1, INF), tails(n + 1, INF);
  int length = 0;
  for (int i = 0; i < n; ++i) {
    int low = 1, high = length;
    while (low <= high) {
      const int mid = (low + high) / 2;  // ceil() of an integer quotient
      if (std::get<1>(m[tails[mid]]) < std::get<1>(m[i]))
        low = mid + 1;
      else
        high = mid - 1;
    }
    const int pos = low;
    parent[i] = tails[pos - 1];
    tails[pos] = i;
    if (pos > length) length = pos;
  }
  std::vector<int> refs(length), frags(length);
  int k = tails[length];
  for (int j = length - 1; j >= 0; --j) {
    refs[j] = (int)std::get<1>(m[k]);
    frags[j] = (int)std::get<0>(m[k]);
    k = parent[k];
  }
  t_begin = refs[0];
  t_end = refs[length - 1];
  q_begin = frags[0];
  q_end = frags[length - 1];
  return length;
}
}  // namespace

void sa_cpu_chain_batch(const uint32_t* ref_triples, size_t n_ref, float f, const uint32_t* read_triples,
                        const uint64_t* read_off, size_t n_reads, int32_t* out, uint64_t* n_matches) {
  SeedIndex ref = build_index(ref_triples, n_ref);
  drop_frequent(ref, f);
  for (size_t r = 0; r < n_reads; ++r) {
    const SeedIndex frag = build_index(read_triples + 3 * read_off[r], (size_t)(read_off[r + 1] - read_off[r]));
    const auto m = match_seeds(ref, frag);
    int tb = 0, te = 0, qb = 0, qe = 0;
    const int len = chain_lis(m, tb, te, qb, qe);
    int32_t* o = out + 5 * r;
    o[0] = len;
    o[1] = tb;
    o[2] = te;
    o[3] = qb;
    o[4] = qe;
    if (n_matches) n_matches[r] = m.size();
  }
}

// ---------------------------------------------------------------------------------------------------
// Sequence files -> records, the way the mapper's parser calls deliver them (reference
// src/mapper.cpp:532-558 into Sequence{name, data, quality}, :21-37).  The parser itself is bioparser, a
// git submodule the reference does not vendor, so this is a restatement of its documented behaviour and
// PARITY UNPINNED: FASTA records start at a '>' outside a header line, the name is the header up to its
// first white-space byte, the data are the non-white-space bytes up to the next record; FASTQ records are
// four lines ('@' header, bases, '+' line, qualities of the same length).  Anything else is an error.
namespace {
bool blank(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

struct Record {
  uint64_t name_start;
  uint32_t name_len;
  uint64_t qual_start;
  std::string data;
};

uint32_t name_span(const uint8_t* file, uint64_t n, uint64_t from) {
  uint64_t q = from;
  while (q < n && !blank(file[q])) ++q;
  return (uint32_t)(q - from);
}

bool parse_fasta(const uint8_t* file, uint64_t n, std::vector<Record>& recs) {
  if (n == 0) return true;
  if (file[0] != '>') return false;
  uint64_t i = 0;
  while (i < n) {
    // at a '>' that opens a record
    Record r;
    r.name_start = i + 1;
    r.name_len = name_span(file, n, i + 1);
    r.qual_start = 0;
    if (r.name_len == 0) return false;
    while (i < n && file[i] != '\n') ++i;  // rest of the header line
    while (i < n && file[i] != '>') {
      if (!blank(file[i])) r.data.push_back((char)file[i]);
      ++i;
    }
    if (r.data.empty()) return false;
    recs.push_back(std::move(r));
  }
  return true;
}

bool parse_fastq(const uint8_t* file, uint64_t n, std::vector<Record>& recs) {
  std::vector<std::pair<uint64_t, uint64_t>> lines;  // [begin, end) without the '\n'
  uint64_t b = 0;
  for (uint64_t i = 0; i < n; ++i)
    if (file[i] == '\n') {
      lines.emplace_back(b, i);
      b = i + 1;
    }
  if (b < n) lines.emplace_back(b, n);
  if (lines.size() % 4) return false;
  auto stripped = [&](std::pair<uint64_t, uint64_t> l) {
    std::string s;
    for (uint64_t i = l.first; i < l.second; ++i)
      if (!blank(file[i])) s.push_back((char)file[i]);
    return s;
  };
  for (size_t l = 0; l < lines.size(); l += 4) {
    Record r;
    if (lines[l].first >= lines[l].second || file[lines[l].first] != '@') return false;
    r.name_start = lines[l].first + 1;
    r.name_len = name_span(file, n, r.name_start);
    if (r.name_len == 0) return false;
    r.data = stripped(lines[l + 1]);
    if (r.data.empty()) return false;
    if (lines[l + 2].first >= lines[l + 2].second || file[lines[l + 2].first] != '+') return false;
    r.qual_start = lines[l + 3].first;
    uint64_t qe = lines[l + 3].second;
    while (qe > r.qual_start && blank(file[qe - 1])) --qe;
    if (qe - r.qual_start != r.data.size()) return false;
    recs.push_back(std::move(r));
  }
  return true;
}
}  // namespace

int64_t sa_cpu_parse(const uint8_t* file, uint64_t n, int format, uint8_t* seq, uint64_t* seq_off,
                     uint64_t* name_start, uint32_t* name_len, uint64_t* qual_start, uint64_t cap_records) {
  std::vector<Record> recs;
  const bool ok = format == 0 ? parse_fasta(file, n, recs) : parse_fastq(file, n, recs);
  if (!ok) return -1;
  if (recs.size() > cap_records) return -2;
  uint64_t at = 0;
  for (size_t r = 0; r < recs.size(); ++r) {
    seq_off[r] = at;
    memcpy(seq + at, recs[r].data.data(), recs[r].data.size());
    at += recs[r].data.size();
    name_start[r] = recs[r].name_start;
    name_len[r] = recs[r].name_len;
    if (qual_start) qual_start[r] = recs[r].qual_start;
  }
  seq_off[recs.size()] = at;
  return (int64_t)recs.size();
}

const char* sa_cpu_kind(void) { return "restatement"; }

}  // extern "C"
