// Replays the reference's unit tests (reference test/testAlignmentMinimizer.cpp:7-82) against the
// drop-in library through the reference's own C++ prototypes, with plain checks instead of googletest,
// plus the CIGAR / target_begin values the compiled reference produces for the same calls.
#include <cstdio>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "alignment.h"
#include "minimizer.h"
#include "minimizer_binary.h"
#include "utility.h"

static int failures = 0;
#define CHECK(cond)                                                   \
  do {                                                                \
    if (!(cond)) {                                                    \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++failures;                                                     \
    }                                                                 \
  } while (0)

using Tuple = std::tuple<unsigned int, unsigned int, bool>;

int main() {
  {
    std::string cigar;
    unsigned int tb = 0;
    CHECK(NeedlemanWunschAlgorithm("CTCTGTTCG", 9, "CGTATCTTGA", 10, 0, -1, -1, &cigar, &tb) == -5);
    CHECK(cigar == "1M1D1M1X1M1X2M2X" && tb == 1);
  }
  {
    std::string q = "CTCTGAG", t = "TGTCAGT", cigar;
    unsigned int tb = 0;
    CHECK(SmithWatermanAlgorithm(q.c_str(), q.length(), t.c_str(), t.length(), 2, -2, -1, &cigar, &tb) == 6);
    CHECK(cigar == "2M2I2M" && tb == 2);
  }
  {
    std::string cigar;
    unsigned int tb = 0;
    CHECK(SemiGlobalAlgorithm("AGCATGCAAT", 10, "ATCCGAACATCCAATCGAAGC", 21, 2, -1, -1, &cigar, &tb) == 14);
    CHECK(cigar == "1M1X3M1X4M" && tb == 1);
  }
  {
    std::string cigar;
    unsigned int tb = 0;
    CHECK(Align("ACGT", 4, "TTTACGT", 7, Global, 1, -1, -1, &cigar, &tb) == 1 && cigar == "3D4M" && tb == 3);
    CHECK(Align("ACGT", 4, "TTTACGT", 7, Semiglobal, 1, -1, -1, &cigar, &tb) == 4 && cigar == "4M" && tb == 1);
    CHECK(Align("ACGT", 4, "TTTACGT", 7, Local, 1, -1, -1, &cigar, &tb) == 4 && cigar == "4M" && tb == 3);
    CHECK(Align("ACGT", 4, "TTACGT", 6, Local, 1, -1, -1, nullptr, &tb) == 4 && tb == 2);
    CHECK(Align("ACGT", 4, "TTTACGT", 7, Global, 1, -1, -1, nullptr, nullptr) == 1);
    CHECK(Align("ACGT", 4, "ACGT", 4, (AlignmentType)7, 1, -1, -1, &cigar, &tb) == 9999);
  }
  {
    std::vector<Tuple> m = MinimizeBinary("TGACGTACATGGACA", 15, 3, 4);
    std::vector<Tuple> want = {Tuple(2, 10, false), Tuple(6, 4, false), Tuple(6, 7, true), Tuple(11, 1, false)};
    CHECK(m == want);
  }
  {
    std::vector<Tuple> m = Minimize("TGACGTACATGGACA", 15, 3, 4);
    std::vector<Tuple> want = {Tuple(2, 10, false), Tuple(12, 4, false), Tuple(12, 7, true), Tuple(23, 1, false)};
    CHECK(m == want);
    bool threw = false;
    try {
      Minimize("TGACGTACATGGACATTGACCAGTACGATCGATCGGG", 37, 15, 5);
    } catch (const std::out_of_range&) {
      threw = true;
    }
    CHECK(threw);
  }
  {
    std::vector<Tuple> v = {Tuple(5, 1, true), Tuple(2, 9, false), Tuple(5, 1, true), Tuple(2, 9, true), Tuple(2, 3, false)};
    std::vector<Tuple> u = removeDuplicate(v);
    std::vector<Tuple> sorted = {Tuple(2, 3, false), Tuple(2, 9, false), Tuple(2, 9, true), Tuple(5, 1, true), Tuple(5, 1, true)};
    std::vector<Tuple> uniq = {Tuple(2, 3, false), Tuple(2, 9, false), Tuple(2, 9, true), Tuple(5, 1, true)};
    CHECK(v == sorted);
    CHECK(u == uniq);
  }
  CHECK(CompressCigar("MMXDDI") == "1I2D1X2M");
  {
    // The mapper calls Align and MinimizeBinary from the threads of its pool (reference
    // src/mapper.cpp:662-679 -> :410, :423): the same calls from 8 threads at once must give the answers
    // of the single-threaded calls.
    std::vector<std::string> reads;
    unsigned x = 12345;
    for (int i = 0; i < 8; ++i) {
      std::string r;
      for (int j = 0; j < 600 + 150 * i; ++j) {
        x = x * 1664525u + 1013904223u;
        r.push_back("ACGT"[(x >> 24) & 3]);
      }
      reads.push_back(r);
    }
    std::vector<int> score(8);
    std::vector<std::string> cig(8);
    std::vector<std::vector<Tuple>> mins(8);
    for (int i = 0; i < 8; ++i) {
      const std::string& q = reads[i];
      const std::string& t = reads[(i + 1) % 8];
      unsigned tb = 0;
      score[i] = Align(q.c_str(), q.size(), t.c_str(), t.size(), (AlignmentType)(i % 3), 1, -1, -1, &cig[i], &tb);
      mins[i] = MinimizeBinary(q.c_str(), q.size(), 15, 5);
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int i = 0; i < 8; ++i)
      pool.emplace_back([&, i]() {
        const std::string& q = reads[i];
        const std::string& t = reads[(i + 1) % 8];
        for (int rep = 0; rep < 5; ++rep) {
          std::string c;
          unsigned tb = 0;
          if (Align(q.c_str(), q.size(), t.c_str(), t.size(), (AlignmentType)(i % 3), 1, -1, -1, &c, &tb) != score[i] ||
              c != cig[i])
            ++bad;
          if (MinimizeBinary(q.c_str(), q.size(), 15, 5) != mins[i]) ++bad;
        }
      });
    for (auto& th : pool) th.join();
    CHECK(bad == 0);
  }
  std::printf(failures ? "%d check(s) failed\n" : "all drop-in checks passed\n", failures);
  return failures ? 1 : 0;
}
