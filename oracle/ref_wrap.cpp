/*
 * TEST INFRASTRUCTURE ONLY - C-interface wrapper (oracle/sa_oracle.h) around the REFERENCE's own
 * functions.  oracle/Makefile compiles this file together with the reference's alignment.cpp,
 * minimizer.cpp, minimizer_binary.cpp and utility.cpp, taken from where the reference checkout is
 * mounted (never copied), into oracle/_ref/libsa_ref.so.  Used to validate the restatement, to
 * generate the fixtures under tests/golden/, and as the timed CPU comparison.
 */
#include <string>
#include <tuple>
#include <vector>

#include "alignment.h"          // reference header, found through -I<reference>/src
#include "minimizer.h"          // reference header
#include "minimizer_binary.h"   // reference header
#include "utility.h"            // reference header

#include <atomic>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "sa_oracle.h"

namespace {

using Tuple = std::tuple<unsigned int, unsigned int, bool>;

size_t copy_out(const std::vector<Tuple>& v, uint32_t* out, size_t cap) {
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    out[3 * i] = std::get<0>(v[i]);
    out[3 * i + 1] = std::get<1>(v[i]);
    out[3 * i + 2] = std::get<2>(v[i]) ? 1u : 0u;
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
  std::string cigar;
  unsigned int tb = 0;
  int score = Align(query, query_len, target, target_len, (AlignmentType)type, match, mismatch, gap,
                    cigar_buf ? &cigar : nullptr, target_begin ? &tb : nullptr);
  if (cigar_buf) {
    if (cigar_cap) {
      size_t c = cigar.size() < cigar_cap - 1 ? cigar.size() : cigar_cap - 1;
      std::memcpy(cigar_buf, cigar.data(), c);
      cigar_buf[c] = 0;
    }
    if (cigar_len) *cigar_len = cigar.size();
  }
  if (target_begin) *target_begin = tb;
  return score;
}

size_t sa_cpu_minimize_binary(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                              uint32_t window_len, uint32_t* out_triples, size_t cap) {
  return copy_out(MinimizeBinary(sequence, sequence_len, kmer_len, window_len), out_triples, cap);
}

size_t sa_cpu_minimize_decimal(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                               uint32_t window_len, uint32_t* out_triples, size_t cap) {
  try {
    return copy_out(Minimize(sequence, sequence_len, kmer_len, window_len), out_triples, cap);
  } catch (const std::out_of_range&) {
    return (size_t)-1;
  } catch (const std::invalid_argument&) {
    return (size_t)-1;
  }
}

size_t sa_cpu_remove_duplicate(uint32_t* triples, size_t n, uint32_t* out_triples) {
  std::vector<Tuple> v;
  for (size_t i = 0; i < n; ++i)
    v.emplace_back(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2] != 0);
  std::vector<Tuple> u = removeDuplicate(v);
  copy_out(v, triples, n);
  return copy_out(u, out_triples, u.size());
}

int sa_cpu_align_batch(const char* q_blob, const uint64_t* q_off, const char* t_blob,
                       const uint64_t* t_off, size_t n, int type, int match, int mismatch, int gap,
                       int want_cigar, int threads, int32_t* scores, uint32_t* target_begin) {
  task_farm(n, threads, [&](size_t i) {
    std::string cigar;
    unsigned int tb = 0;
    scores[i] = Align(q_blob + q_off[i], (unsigned)(q_off[i + 1] - q_off[i]), t_blob + t_off[i],
                      (unsigned)(t_off[i + 1] - t_off[i]), (AlignmentType)type, match, mismatch, gap,
                      want_cigar ? &cigar : nullptr, (want_cigar || type == 1) ? &tb : nullptr);
    if (target_begin) target_begin[i] = tb;
  });
  return 0;
}

int sa_cpu_minimize_batch(const char* blob, const uint64_t* off, const uint32_t* len, size_t n,
                          uint32_t kmer_len, uint32_t window_len, int threads, uint64_t* counts,
                          uint64_t* hashes) {
  task_farm(n, threads, [&](size_t i) {
    std::vector<Tuple> list = MinimizeBinary(blob + off[i], len[i], kmer_len, window_len);
    counts[i] = list.size();
    if (hashes) {
      uint64_t h = 0;
      for (size_t x = 0; x < list.size(); ++x)
        h += sa_triple_hash(std::get<0>(list[x]), std::get<1>(list[x]), std::get<2>(list[x]) ? 1u : 0u, x);
      hashes[i] = h;
    }
  });
  return 0;
}

const char* sa_cpu_kind(void) { return "reference"; }

}  // extern "C"
