// C++ drop-in for the reference's hot-path libraries: exports the reference's own symbols
// (include/alignment.h, minimizer.h, minimizer_binary.h, utility.h) and forwards every call to the
// C-ABI of libsa_hip.so as a batch of one.  std::string / std::vector<std::tuple> exist only on this
// side of the boundary.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "alignment.h"
#include "minimizer.h"
#include "minimizer_binary.h"
#include "sa_hip.h"
#include "utility.h"

namespace {

using Tuple = std::tuple<unsigned int, unsigned int, bool>;

[[noreturn]] void raise(const char* what) {
  throw std::runtime_error(std::string(what) + ": " + sa_last_error());
}

int align_one(const char* query, unsigned int query_len, const char* target, unsigned int target_len,
              int type, int match, int mismatch, int gap, std::string* cigar,
              unsigned int* target_begin) {
  int32_t score = 0;
  uint32_t tb = 0;
  char* text = nullptr;
  const uint32_t ql = query_len, tl = target_len;
  // Local's target_begin does not depend on the CIGAR; Global/Semiglobal derive theirs from the CIGAR
  // text (reference src/alignment.cpp:129-136, 342-348), so without a cigar argument they see the empty
  // string and report 1.
  const bool need_tb = target_begin && (cigar || type == SA_LOCAL);
  if (sa_align_batch_host(&query, &ql, &target, &tl, 1, type, match, mismatch, gap, 1, &score,
                          need_tb ? &tb : nullptr, cigar ? &text : nullptr) != SA_OK)
    raise("Align");
  if (cigar) {
    *cigar = text ? text : "";
    sa_free(text);
  }
  if (target_begin) *target_begin = need_tb ? tb : 1u;
  return score;
}

std::vector<Tuple> minimize_one(const char* sequence, unsigned int len, unsigned int k, unsigned int w,
                                int variant) {
  uint32_t* triples = nullptr;
  uint64_t count = 0;
  const uint32_t l = len;
  const int rc = sa_minimize_batch_host(&sequence, &l, 1, k, w, variant, 1, &triples, &count);
  if (rc == SA_ERR_RANGE) throw std::out_of_range("stoi");
  if (rc != SA_OK) raise(variant == SA_MIN_BINARY ? "MinimizeBinary" : "Minimize");
  std::vector<Tuple> out;
  out.reserve(count);
  for (uint64_t i = 0; i < count; ++i)
    out.emplace_back(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2] != 0);
  sa_free(triples);
  return out;
}

}  // namespace

std::string CompressCigar(std::string uCigar) {
  std::string cigar;
  size_t i = uCigar.size();
  while (i > 0) {
    const char op = uCigar[i - 1];
    size_t run = 0;
    while (i > 0 && uCigar[i - 1] == op) {
      --i;
      ++run;
    }
    cigar += std::to_string(run);
    cigar += op;
  }
  return cigar;
}

int NeedlemanWunschAlgorithm(const char* query, unsigned int query_len, const char* target,
                             unsigned int target_len, int match, int mismatch, int gap,
                             std::string* cigar, unsigned int* target_begin) {
  return align_one(query, query_len, target, target_len, SA_GLOBAL, match, mismatch, gap, cigar,
                   target_begin);
}

int SmithWatermanAlgorithm(const char* query, unsigned int query_len, const char* target,
                           unsigned int target_len, int match, int mismatch, int gap,
                           std::string* cigar, unsigned int* target_begin) {
  return align_one(query, query_len, target, target_len, SA_LOCAL, match, mismatch, gap, cigar,
                   target_begin);
}

int SemiGlobalAlgorithm(const char* query, unsigned int query_len, const char* target,
                        unsigned int target_len, int match, int mismatch, int gap, std::string* cigar,
                        unsigned int* target_begin) {
  return align_one(query, query_len, target, target_len, SA_SEMIGLOBAL, match, mismatch, gap, cigar,
                   target_begin);
}

int Align(const char* query, unsigned int query_len, const char* target, unsigned int target_len,
          AlignmentType type, int match, int mismatch, int gap, std::string* cigar,
          unsigned int* target_begin) {
  if (type != Global && type != Local && type != Semiglobal)
    return SA_UNKNOWN_TYPE_SCORE;  // reference src/alignment.cpp:371-373
  return align_one(query, query_len, target, target_len, (int)type, match, mismatch, gap, cigar,
                   target_begin);
}

std::vector<Tuple> MinimizeBinary(const char* sequence, unsigned int sequence_len,
                                  unsigned int kmer_len, unsigned int window_len) {
  return minimize_one(sequence, sequence_len, kmer_len, window_len, SA_MIN_BINARY);
}

std::vector<Tuple> Minimize(const char* sequence, unsigned int sequence_len, unsigned int kmer_len,
                            unsigned int window_len) {
  return minimize_one(sequence, sequence_len, kmer_len, window_len, SA_MIN_DECIMAL);
}

std::vector<Tuple> removeDuplicate(std::vector<Tuple>& kmerList) {
  std::vector<Tuple> out;
  const size_t n = kmerList.size();
  if (n == 0) return out;  // the reference reads kmerList[0] here (undefined)
  std::vector<uint32_t> in(3 * n), uniq(3 * n);
  for (size_t i = 0; i < n; ++i) {
    in[3 * i] = std::get<0>(kmerList[i]);
    in[3 * i + 1] = std::get<1>(kmerList[i]);
    in[3 * i + 2] = std::get<2>(kmerList[i]) ? 1u : 0u;
  }
  size_t m = 0;
  if (sa_remove_duplicate_host(in.data(), n, uniq.data(), &m) != SA_OK) raise("removeDuplicate");
  for (size_t i = 0; i < n; ++i) kmerList[i] = Tuple(in[3 * i], in[3 * i + 1], in[3 * i + 2] != 0);
  out.reserve(m);
  for (size_t i = 0; i < m; ++i) out.emplace_back(uniq[3 * i], uniq[3 * i + 1], uniq[3 * i + 2] != 0);
  return out;
}
