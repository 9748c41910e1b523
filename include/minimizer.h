// Drop-in replacement for the reference's src/minimizer.h (reference src/minimizer.h:2-5).
#ifndef SEQUENCEALIGNER_AMD_MINIMIZER_H
#define SEQUENCEALIGNER_AMD_MINIMIZER_H

#include <tuple>
#include <vector>

// Decimal-string (k,w)-minimizers (reference src/minimizer.cpp:124-214).  Throws std::out_of_range
// where the reference's stoi does (a k-mer's decimal value above INT_MAX).
std::vector<std::tuple<unsigned int, unsigned int, bool>> Minimize(
    const char* sequence, unsigned int sequence_len,
    unsigned int kmer_len,
    unsigned int window_len);

#endif
