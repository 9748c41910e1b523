// Drop-in replacement for the reference's src/minimizer_binary.h (reference src/minimizer_binary.h:1-4).
#ifndef SEQUENCEALIGNER_AMD_MINIMIZER_BINARY_H
#define SEQUENCEALIGNER_AMD_MINIMIZER_BINARY_H

#include <tuple>
#include <vector>

// 2-bit (k,w)-minimizers as used by the mapper (reference src/minimizer_binary.cpp:136-178).
std::vector<std::tuple<unsigned int, unsigned int, bool>> MinimizeBinary(
    const char* sequence, unsigned int sequence_len,
    unsigned int kmer_len,
    unsigned int window_len);

#endif
