// Drop-in replacement for the reference's src/utility.h (reference src/utility.h:1-2).
#ifndef SEQUENCEALIGNER_AMD_UTILITY_H
#define SEQUENCEALIGNER_AMD_UTILITY_H

#include <tuple>
#include <vector>

// Sorts kmerList in place by (value, position, strand) and returns the list without duplicates
// (reference src/utility.cpp:8-26).
std::vector<std::tuple<unsigned int, unsigned int, bool>> removeDuplicate(
    std::vector<std::tuple<unsigned int, unsigned int, bool>>& kmerList);

#endif
