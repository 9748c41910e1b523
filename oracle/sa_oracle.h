/*
 * TEST INFRASTRUCTURE ONLY.  CPU checker interface for the alignment / minimizer hot path.
 *
 * Two shared libraries implement this same C interface:
 *   oracle/libsa_oracle.so      - an independent CPU restatement of the reference algorithms
 *                                 (oracle/oracle.cpp), built with g++ from this repository alone;
 *   oracle/_ref/libsa_ref.so    - the reference's own alignment.cpp / minimizer*.cpp / utility.cpp
 *                                 compiled where they lie (oracle/ref_wrap.cpp + oracle/Makefile),
 *                                 available only where the reference checkout is mounted at build
 *                                 time.
 * Nothing in the shipped GPU path links, loads or calls either library: they are used by tests/,
 * by the smoke check and by the CPU-baseline leg of the benchmark, as the checker / the timed CPU
 * comparison, never as the thing measured as "GPU" or shipped.
 */
#ifndef SA_ORACLE_H
#define SA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

/* Hash of entry `index` of a minimizer list (the splitmix64 finalizer over the packed entry plus a multiple
 * of the index), summed over a list by sa_cpu_minimize_batch. */
static inline uint64_t sa_triple_hash(uint32_t value, uint32_t pos, uint32_t strand, uint64_t index) {
  uint64_t x = ((uint64_t)value | ((uint64_t)pos << 32) | ((uint64_t)strand << 63)) +
               0x9E3779B97F4A7C15ull * (index + 1);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

#ifdef __cplusplus
extern "C" {
#endif

/* type: 0 Global (Needleman-Wunsch), 1 Local (Smith-Waterman), 2 Semiglobal; anything else -> 9999
 * (reference src/alignment.h:25, src/alignment.cpp:353-374).
 * cigar_buf may be NULL (no CIGAR requested; mirrors cigar == nullptr).  If non-NULL the compressed
 * CIGAR is written NUL-terminated (at most cigar_cap bytes incl. NUL; *cigar_len gets the full length).
 * target_begin may be NULL.  Returns the alignment score. */
int sa_cpu_align(const char* query, uint32_t query_len, const char* target, uint32_t target_len,
                 int type, int match, int mismatch, int gap,
                 char* cigar_buf, size_t cigar_cap, size_t* cigar_len, uint32_t* target_begin);

/* The alignment score alone, computed in linear memory (two rows of the matrix): same value as
 * sa_cpu_align returns, for pairs whose full matrix would not fit in host memory.  Only the restatement
 * library has it (the reference has no such function); it is checked against the compiled reference's
 * Align() on pairs of up to 5 kbp. */
int sa_cpu_score_linear(const char* query, uint32_t query_len, const char* target, uint32_t target_len,
                        int type, int match, int mismatch, int gap);

/* sa_cpu_score_linear over n independent pairs (blobs + n+1 offsets) on `threads` std::threads. Returns 0. */
int sa_cpu_score_linear_batch(const char* q_blob, const uint64_t* q_off, const char* t_blob,
                              const uint64_t* t_off, size_t n, int type, int match, int mismatch, int gap,
                              int threads, int32_t* scores);

/* MinimizeBinary (reference src/minimizer_binary.cpp:136-178).  sequence must be NUL-terminated.
 * Writes up to cap triples (value, position, strand) to out; returns the number of minimizers. */
size_t sa_cpu_minimize_binary(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                              uint32_t window_len, uint32_t* out_triples, size_t cap);

/* Minimize, decimal-string variant (reference src/minimizer.cpp:124-214).  Returns the number of
 * minimizers, or (size_t)-1 when the reference would throw std::out_of_range from stoi. */
size_t sa_cpu_minimize_decimal(const char* sequence, uint32_t sequence_len, uint32_t kmer_len,
                               uint32_t window_len, uint32_t* out_triples, size_t cap);

/* removeDuplicate (reference src/utility.cpp:8-26): sorts triples in place by (value, position,
 * strand) and writes the de-duplicated list to out; returns its length. n must be >= 1. */
size_t sa_cpu_remove_duplicate(uint32_t* triples, size_t n, uint32_t* out_triples);

/* Task farm over independent pairs with `threads` std::threads - the stand-in for the reference's
 * thread_pool dispatch (reference src/mapper.cpp:662-679).  Sequences are given as one blob plus
 * n+1 offsets each.  scores/target_begin have n entries; cigar text is discarded after being built
 * (want_cigar != 0 builds it, as the mapper does with -c). Returns 0. */
int sa_cpu_align_batch(const char* q_blob, const uint64_t* q_off, const char* t_blob,
                       const uint64_t* t_off, size_t n, int type, int match, int mismatch, int gap,
                       int want_cigar, int threads, int32_t* scores, uint32_t* target_begin);

/* Task farm over independent NUL-terminated sequences (seq i = blob + off[i], length len[i]).
 * counts[i] receives the number of minimizers of sequence i and, when hashes is not NULL, hashes[i] the
 * sum (mod 2^64) of sa_triple_hash over the list's entries in list order: two lists with equal count and
 * hash are equal entry for entry, up to a 2^-64 chance.  Returns 0. */
int sa_cpu_minimize_batch(const char* blob, const uint64_t* off, const uint32_t* len, size_t n,
                          uint32_t kmer_len, uint32_t window_len, int threads, uint64_t* counts,
                          uint64_t* hashes);

/* "restatement" or "reference" */
/* Restates getAlignmentBlockLengthAndMatchLength (reference src/mapper.cpp:371-399): *abl = sum of the
 * CIGAR's counts, *ml = sum of the counts in front of a LOWER-case 'm' (which CompressCigar never
 * prints: the reference's own match length is always 0 on its own CIGARs).  That function is static
 * inside mapper.cpp, which cannot be built here (un-vendored dependencies): parity unpinned. */
void sa_cpu_cigar_block_match(const char* cigar, int* abl, int* ml);

/* Seed chaining of reads against one reference, restating the mapper's own steps (reference
 * src/mapper.cpp, static functions - that file cannot be built here, so this part is parity unpinned):
 *   getMinimizer :172-196          index  value -> [(position, strand)] of a sorted minimizer list
 *   ignoreTooFrequentMinimizer :224-232   drop the (int)((int)distinct * f) values with the largest
 *                                  (occurrences, value)
 *   matchMinimizer :234-290        all (read position, reference position) pairs of equal value and strand,
 *                                  sorted by (read position, reference position)
 *   LongestIncreasingSubsequence :292-352   strictly increasing reference positions; reports its length and
 *                                  the first / last reference and read positions on it
 * ref_triples / read_triples are (value, position, strand) lists as MinimizeBinary + removeDuplicate
 * return them; read r owns triples [read_off[r], read_off[r+1]).  out[5*r ..] = {length, t_begin, t_end,
 * q_begin, q_end}; with no match all five are 0 (the reference leaves its outputs untouched), with one
 * match q_end = q_begin (the reference leaves q_end uninitialised, :302-308).  n_matches (optional)
 * receives the number of matched pairs per read. */
void sa_cpu_chain_batch(const uint32_t* ref_triples, size_t n_ref, float f, const uint32_t* read_triples,
                        const uint64_t* read_off, size_t n_reads, int32_t* out, uint64_t* n_matches);

/* FASTA (format 0) / FASTQ (format 1) file bytes -> records, as the mapper's parser calls deliver them
 * (reference src/mapper.cpp:532-558; the parser is bioparser, which the reference does not vendor: a
 * restatement of its documented rules, parity unpinned).  seq (capacity n) receives the bases of all records
 * back to back, seq_off (cap_records + 1) their first bytes, name_start / name_len the names as spans of the
 * file, qual_start (optional) the first byte of each FASTQ quality line.  Returns the number of records,
 * -1 for a file that breaks the format, -2 if cap_records is too small. */
int64_t sa_cpu_parse(const uint8_t* file, uint64_t n, int format, uint8_t* seq, uint64_t* seq_off,
                     uint64_t* name_start, uint32_t* name_len, uint64_t* qual_start, uint64_t cap_records);

const char* sa_cpu_kind(void);

#ifdef __cplusplus
}
#endif
#endif
