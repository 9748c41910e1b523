// Drop-in replacement for the reference's src/alignment.h (same prototypes, same plain enum, global
// namespace; reference src/alignment.h:1-36), self-contained.  The functions are implemented by
// libsequencealigner_amd.so on the MI355X through the C-ABI in sa_hip.h; there is no CPU fallback
// (a call without a usable GPU throws std::runtime_error).
#ifndef SEQUENCEALIGNER_AMD_ALIGNMENT_H
#define SEQUENCEALIGNER_AMD_ALIGNMENT_H

#include <string>

// Run-length encodes a traceback string given in reverse order (reference src/alignment.cpp:17-33).
std::string CompressCigar(std::string uCigar);

int SemiGlobalAlgorithm(
    const char* query, unsigned int query_len,
    const char* target, unsigned int target_len,
    int match,
    int mismatch,
    int gap, std::string* cigar, unsigned int* target_begin);

int SmithWatermanAlgorithm(
    const char* query, unsigned int query_len,
    const char* target, unsigned int target_len,
    int match,
    int mismatch,
    int gap, std::string* cigar, unsigned int* target_begin);

int NeedlemanWunschAlgorithm(
    const char* query, unsigned int query_len,
    const char* target, unsigned int target_len,
    int match,
    int mismatch,
    int gap, std::string* cigar, unsigned int* target_begin);

enum AlignmentType {Global, Local, Semiglobal};

int Align(
    const char* query, unsigned int query_len,
    const char* target, unsigned int target_len,
    AlignmentType type,
    int match,
    int mismatch,
    int gap,
    std::string* cigar,
    unsigned int* target_begin);

#endif
