/*
 * C-ABI of the MI355X (gfx950) implementation of the SequenceAligner hot path:
 * batched Align() (Needleman-Wunsch / Smith-Waterman / semi-global with CIGAR traceback) and batched
 * Minimize()/MinimizeBinary() (k,w)-minimizer extraction.  Plain pointers and sizes only.
 *
 * The reference has no C interface of its own: its hot path is four C++ static libraries with free
 * functions (reference src/alignment.h:1-36, src/minimizer.h:2-5, src/minimizer_binary.h:1-4,
 * src/utility.h:1-2).  The C++ drop-in for those prototypes lives in include/alignment.h,
 * include/minimizer.h, include/minimizer_binary.h and include/utility.h and is a thin layer over the
 * functions below.  The batch functions take the place of the reference's one-task-per-read thread
 * pool dispatch (reference src/mapper.cpp:662-679): a batch is a set of independent pairs / sequences.
 *
 * Every function returns 0 on success and a non-zero code on failure; sa_last_error() then returns a
 * message for the calling thread.  Nothing here falls back to a CPU implementation: without a GPU the
 * calls fail.
 */
#ifndef SA_HIP_H
#define SA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentType values, identical to the reference's plain enum (src/alignment.h:25). */
enum { SA_GLOBAL = 0, SA_LOCAL = 1, SA_SEMIGLOBAL = 2 };

/* Value returned as "score" for an unknown alignment type (reference src/alignment.cpp:371-373). */
#define SA_UNKNOWN_TYPE_SCORE 9999

/* Score of every pair of a batch whose cooperative kernels gave up waiting, with the option "coop_async" (no
 * counterpart in the reference; no alignment has this score). */
#define SA_SCORE_GAVE_UP (-2147483647 - 1)

/* CIGAR run encoding on the device: (length << 2) | op, op 0='M' 1='X' 2='I' 3='D'
 * (same four letters as reference src/alignment.cpp:104-119). */
#define SA_OP_M 0u
#define SA_OP_X 1u
#define SA_OP_I 2u
#define SA_OP_D 3u

/* Minimizer variants. SA_MIN_BINARY = MinimizeBinary (src/minimizer_binary.cpp:136-178),
 * SA_MIN_DECIMAL = Minimize (src/minimizer.cpp:124-214). */
enum { SA_MIN_BINARY = 0, SA_MIN_DECIMAL = 1 };

/* Status codes. */
enum {
  SA_OK = 0,
  SA_ERR_HIP = 1,          /* a HIP runtime call failed */
  SA_ERR_ARGUMENT = 2,     /* invalid argument */
  SA_ERR_WORKSPACE = 3,    /* workspace too small */
  SA_ERR_NO_DEVICE = 4,    /* no usable GPU */
  SA_ERR_RANGE = 5,        /* decimal Minimize: k-mer value exceeds INT_MAX (reference throws from stoi) */
  SA_ERR_CAPACITY = 6,     /* an output buffer is too small */
  SA_ERR_FORMAT = 7,       /* a sequence file breaks the rules of its format */
  SA_ERR_MEMORY = 8        /* host memory exhausted */
};

const char* sa_last_error(void);
int sa_device_count(void);
/* "gfx950..." name of the device's architecture, or NULL. */
const char* sa_device_arch(int device);

/* Tuning and test options of the process (no counterpart in the reference).  Every option starts from an
 * environment variable that is read once, when the library first needs an option; afterwards only this call
 * changes it, so that no call reads the environment while another thread may be changing it.
 *   "force_i32"     (SA_FORCE_I32)     1: never use the packed 16-bit kernels
 *   "force_family"  (SA_FORCE_FAMILY=lane|wave)  0 automatic, 1 lane kernels, 2 wavefront kernels
 *   "force_coop"    (SA_FORCE_COOP)    -1 automatic, 0 / 1: cooperative wavefront kernel off / on where possible
 *   "no_tables"     (SA_NO_TABLES)     1: no kernel looks substitution scores up in tables
 *   "coop_c"        (SA_COOP_C)        columns per lane of the cooperative kernel, 0 automatic
 *   "coop_pk"       (SA_COOP_PK)       -1 automatic, 0 / 1: the packed form of the cooperative Global / Semiglobal
 *                                      kernel (two pairs per wavefront) off / on where it applies
 *   "no_overlap"    (SA_NO_OVERLAP)    1: never run the traceback of a launch's first part on a second stream
 *                                      beside the fill of the rest
 *   "coop_waves"    (SA_COOP_WAVES)    wavefronts (strips of pairs) in one pass of the cooperative Global / Semiglobal
 *                                      kernels, 64..8192 (default 8192: the packed kernel then takes what the
 *                                      workspace holds, up to 2^18 strips)
 *   "coop_async"    (SA_COOP_ASYNC)    1: a call that runs the cooperative kernels only enqueues, like every other
 *                                      call (sa_align_may_synchronize then answers 0); should one of the kernels'
 *                                      bounded waits run out, every score of the batch is SA_SCORE_GAVE_UP and the
 *                                      other outputs are undefined, instead of the call failing with SA_ERR_HIP
 *   "coop_seg_rows" (SA_COOP_SEG_ROWS) packed cooperative kernel: target rows per segment of a strip (one wavefront
 *                                      per segment), -1 automatic (1024 when a pass has more than 3072 strips,
 *                                      else whole strips), 0 whole strips
 *   "pk16_k"        (SA_PK16_K)        rows per strip of the packed lane kernel: 32, 48 or 64; 0 automatic (48 or 64,
 *                                      by the rows and the strips the targets then take)
 *   "min_chunk"     (SA_MIN_CHUNK)     windows per minimizer chunk, 0 automatic
 *   "no_buckets"    (SA_NO_BUCKETS)    1: a single sequence is sorted by the device-wide radix sort, as sequences
 *                                      beyond 32 M windows always are
 *   "no_segments"   (SA_NO_SEGMENTS)   1: the path of a long Global / Semiglobal pair is walked by one wavefront,
 *                                      not by one wavefront per segment of target rows
 *   "seg_rows"      (SA_SEG_ROWS)      target rows per such segment, 0 automatic (1024, more for targets
 *                                      beyond 1 M rows)
 *   "spin_limit"    (SA_SPIN_LIMIT)    polls a wait inside a cooperative kernel may take before the kernels give up
 *                                      and the call fails with SA_ERR_HIP (default 2^24; a small value is a test hook)
 *   "seg_spec"      (SA_SEG_SPEC)      walks a segment may make from columns that are not final yet (default 2)
 *   "seg_cap"       (SA_SEG_CAP)       runs a segment may hold, 0: its share of the pair's run slot; a small
 *                                      value sends pairs to the one-wavefront walk (a test hook)
 *   "fake_devices"  (SA_FAKE_DEVICES)  the host batch calls deal their units to this many shards, all run on
 *                                      GPU 0 at the same time (the multi-GPU path on a one-GPU machine)
 * A call that is planning a batch while an option changes may see either value; set options between calls. */
int sa_set_option(const char* name, long value);

/* ---------------------------------------------------------------------------------------------
 * Alignment, device-resident batch.  Takes the place of one Align() call per pair
 * (reference src/alignment.cpp:353-374, called per read at src/mapper.cpp:423).
 *
 * Sequences: one byte per symbol (compared as opaque bytes, like target[i-1]==query[r-1] at
 * src/alignment.cpp:63,176,281); pair p has query d_q[q_off[p] .. q_off[p+1]) and target
 * d_t[t_off[p] .. t_off[p+1]).  All d_* pointers are device pointers on `device`; `stream` is a
 * hipStream_t (NULL = default stream).  d_q and d_t must be readable up to 8 bytes past their last
 * offset (symbols are fetched 4 or 8 at a time).  The call only enqueues work on the stream, except where
 * it runs the cooperative kernels, which synchronise the stream once at the end to read their abort flag:
 * batches of few pairs (hundreds) of more than a few hundred symbols, Global / Semiglobal batches of any size with
 * queries of 1800 symbols and more unless a wavefront per two pairs fills the GPU in whole rounds about as well
 * (kernel names "wave_i32_coop" and "wave_pk16_coop", see sa_align_kernel_name), batches of long pairs that need
 * several passes over the workspace and leave at most 2048 pairs for the last pass - those pairs (and a head of
 * few pairs) are planned again as batches of their own -, and batches of long pairs of which the workspace
 * holds fewer than 2048 wavefronts' worth at a time: these may take the cooperative kernels although the
 * batch's kernel name says "wave_pk16" / "wave_i32".  sa_align_may_synchronize tells which it will be for a
 * batch and a workspace size; sa_align_batch_device_static never synchronises.  Apart from that the call may
 * use a second stream of its own beside `stream` (forked from it and joined back into it with events before the
 * call returns its work to `stream`'s order).
 *
 * The batch is planned for max_qlen x max_tlen: every pair pays for direction words and steps of that shape.
 * Callers with mixed sizes should issue one call per size class (sa_align_batch_host does).
 *
 * Outputs (each may be NULL if not wanted, except d_score):
 *   d_score[p]         alignment score
 *   d_target_begin[p]  the value the reference stores in *target_begin (src/alignment.cpp:129-136,
 *                      244-246, 342-348): Local = 0-based first target row; Global/Semiglobal = 3 if the
 *                      CIGAR starts with a D-run of length <= 9, else 1
 *   d_cigar_runs       n * cigar_stride uint32 run slots; pair p owns slots [p*stride, (p+1)*stride) and
 *                      its runs, in forward (CIGAR text) order, end at the end of its slot range:
 *                      first run at p*stride + stride - d_cigar_nruns[p].  Needs
 *                      cigar_stride >= max_qlen + max_tlen (a run per op in the worst case).  The slots
 *                      before a pair's first run hold nothing the caller may rely on (the traceback of long
 *                      pairs assembles the runs there).
 *   d_cigar_nruns[p]   number of runs
 * ------------------------------------------------------------------------------------------- */
typedef struct sa_align_batch {
  const uint8_t* d_q;
  const uint64_t* d_q_off; /* n + 1 entries */
  const uint8_t* d_t;
  const uint64_t* d_t_off; /* n + 1 entries */
  uint32_t n;
  uint32_t max_qlen; /* >= every query length in the batch */
  uint32_t max_tlen; /* >= every target length in the batch */
  uint32_t min_qlen; /* <= every query length (0 = unknown); min == max lets equal-length kernels run */
  uint32_t min_tlen; /* <= every target length (0 = unknown) */
  int type;          /* SA_GLOBAL / SA_LOCAL / SA_SEMIGLOBAL */
  int match, mismatch, gap;
  int32_t* d_score;
  uint32_t* d_target_begin;
  uint32_t* d_cigar_runs;
  uint32_t* d_cigar_nruns;
  uint64_t cigar_stride;
} sa_align_batch;

/* Bytes of device scratch sa_align_batch_device needs for this batch (direction bits, strip boundary
 * rows, repacked sequences, per-pair end cells); only the shape and scoring fields of `batch` are read.  A smaller workspace is accepted as long as it holds at least one
 * block of pairs: the batch is then processed in several passes.  *min_bytes (optional) receives that
 * minimum. */
size_t sa_align_workspace_bytes(const sa_align_batch* batch, size_t* min_bytes);

int sa_align_batch_device(int device, void* stream, const sa_align_batch* batch, void* d_workspace,
                          size_t workspace_bytes);

/* sa_align_batch_device without a workspace argument and without any allocation.  This entry point has no
 * counterpart in the reference; it is for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The scratch memory (direction bits, strip
 * boundary rows, repacked sequences, per-pair end cells) is an array of sa_align_static_scratch_bytes() bytes
 * (64 MiB) that is part of the library's code object: every GPU the library runs on holds its own copy from
 * the first call on that GPU until the process ends.  Same inputs, outputs and results as
 * sa_align_batch_device; a batch larger than one pass of the array is processed in several passes.  The call
 * only enqueues kernels on `stream`: it never allocates, frees or synchronises, and never picks the cooperative
 * kernels (which synchronise).  A batch whose single pass does not fit (for example pairs beyond about
 * 10 kbp x 11 kbp) is rejected with SA_ERR_WORKSPACE.  All calls on one GPU share that GPU's array: calls
 * that overlap in time on one GPU (different streams or host threads) race and must be serialised by the
 * caller.  sa_align_profile() must be off (its timing events are allocations). */
size_t sa_align_static_scratch_bytes(void);
int sa_align_batch_device_static(int device, void* stream, const sa_align_batch* batch);
/* Looks up the array's address on `device` (hipGetSymbolAddress, which may load the code object on that GPU).
 * sa_align_batch_device_static does the same on its first call per GPU; callers that capture the call into a
 * HIP graph call this once per GPU beforehand, outside the capture.  (No counterpart in the reference.) */
int sa_align_static_init(int device);

/* 1 when sa_align_batch_device would synchronise `stream` for this batch over a workspace of
 * `workspace_bytes` (it runs cooperative kernels for the batch or for a part of it), else 0: such a call
 * must not be captured into a graph or relied on to return before its kernels have run.  Only the batch's
 * n, type, scoring and maximum lengths are read.  (No counterpart in the reference.) */
int sa_align_may_synchronize(int device, const sa_align_batch* batch, size_t workspace_bytes);

/* Name of the kernel family the dispatcher picks for this shape ("lane_pk16", "lane_i32", "wave_pk16",
 * "wave_pk16_coop", "wave_i32", "wave_i32_coop"), for logs and profiles. */
const char* sa_align_kernel_name(const sa_align_batch* batch);

/* Kernel timing for benchmarks (no counterpart in the reference).  While enabled, every
 * sa_align_batch_device call on this process brackets its kernels with HIP events on the call's own
 * stream; sa_align_profile_read() synchronises those events and returns the accumulated
 * milliseconds and launch counts per kernel class since the last read: index 0 = sequence packing,
 * 1 = DP fill, 2 = traceback.  Not thread-safe; meant for one benchmarking thread. */
int sa_align_profile(int enable);
int sa_align_profile_read(double ms[3], uint64_t launches[3]);

/* CIGAR statistics on the device, for the PAF columns the reference's mapper derives from every
 * alignment (getAlignmentBlockLengthAndMatchLength, reference src/mapper.cpp:371-399, called at :428
 * and printed by generatePAFString, :354-369): d_block_len[p] = sum of all run lengths (the mapper's
 * "alignment block length"); d_op_len[4*p + op] = total length of the runs of op (SA_OP_M, _X, _I, _D).
 * The mapper's own "match length" looks for a lower-case 'm' (:392) that CompressCigar never prints and
 * is therefore always 0; d_op_len[4*p + SA_OP_M] is the number a corrected caller would want.
 * d_cigar_runs / d_cigar_nruns / cigar_stride are the arrays sa_align_batch_device filled; either
 * output may be NULL. */
int sa_cigar_stats_device(int device, void* stream, uint32_t n, const uint32_t* d_cigar_runs,
                          const uint32_t* d_cigar_nruns, uint64_t cigar_stride, uint32_t* d_block_len,
                          uint32_t* d_op_len);

/* ---------------------------------------------------------------------------------------------
 * Alignment, host buffers.  The device batch queue: pairs are split over `n_devices` GPUs
 * (0 = all visible) by cost (query_len * target_len), staged through pinned buffers, aligned on each
 * GPU's own stream and gathered back in input order - no inter-GPU communication.  Pairs are grouped by
 * size class (powers of two of query and target length) and each class runs as a batch of its own.
 * cigars (optional) receives n malloc'ed NUL-terminated CIGAR strings, "<count><op>..." exactly as
 * CompressCigar prints them (src/alignment.cpp:17-33); free each with sa_free().  target_begin (optional)
 * receives what Align() stores in *target_begin for the same arguments: for Global / Semiglobal without
 * `cigars` that is 1 (the reference then inspects an empty CIGAR string).  When the call fails,
 * every entry of cigars is NULL and nothing is left to free.
 * ------------------------------------------------------------------------------------------- */
int sa_align_batch_host(const char* const* queries, const uint32_t* query_lens,
                        const char* const* targets, const uint32_t* target_lens, size_t n, int type,
                        int match, int mismatch, int gap, int n_devices, int32_t* scores,
                        uint32_t* target_begin, char** cigars);
void sa_free(void* p);

/* ---------------------------------------------------------------------------------------------
 * Minimizers, device-resident batch.  Takes the place of one MinimizeBinary()/Minimize() call per
 * sequence (reference src/minimizer_binary.cpp:136-178, called at src/mapper.cpp:179).
 *
 * Sequence s is d_seq[off[s] .. off[s+1]) (ASCII; ACGT in either case are bases, any other byte counts as
 * the reference's code -1, src/minimizer_binary.cpp:28-29 - SA_MIN_BINARY reproduces what that does to the
 * k-mer values, a batch containing such bytes takes about 2.5 times as long).  Output: for each sequence its
 * minimizers sorted by (value, position, strand) with duplicates removed (reference
 * src/utility.cpp:8-26), as uint32 triples {value, position, strand(1 = forward, 0 = other strand)};
 * sequence s owns triples [d_out_off[s], d_out_off[s+1]) of d_out.  out_capacity is in triples;
 * sa_minimize_out_capacity() gives a sufficient value.  d_status (one int, optional) receives SA_OK,
 * SA_ERR_CAPACITY, SA_ERR_RANGE or - SA_MIN_DECIMAL on a sequence with a byte other than ACGT / acgt, which the
 * reference drops while it keeps counting windows by the original length (src/minimizer.cpp:13-40, 174); not
 * reproduced - SA_ERR_ARGUMENT, written by the device when the work completes.
 * Requires 1 <= k <= 31, w >= 1, every length >= k + w - 1 and, for SA_MIN_BINARY, w <= k + 1; other arguments
 * are rejected with SA_ERR_ARGUMENT.  For k > 16 the values are those of the reference's 32-bit arithmetic
 * (the last 16 bases of the k-mer, src/minimizer_binary.cpp:53-78).  The limits k <= 31 and w <= k + 1 are this
 * implementation's (it tracks at most 32 candidate positions per strand); the reference itself runs w > k + 1
 * and reads past its k-mer list only near the end of a sequence.  SA_MIN_DECIMAL with
 * k > 10 is rejected with SA_ERR_RANGE (every 11-digit value overflows the reference's stoi unless it
 * starts with zeros); for k = 10 the range is checked per k-mer on the device (d_status).
 * d_seq must be readable up to 8 bytes past off[n] (bases are fetched eight at a time).
 * ------------------------------------------------------------------------------------------- */
typedef struct sa_minimize_batch {
  const uint8_t* d_seq;
  const uint64_t* d_off; /* n + 1 entries */
  uint32_t n;
  uint64_t total_len; /* == off[n] */
  uint32_t max_len;
  uint32_t min_len;
  uint32_t k, w;
  int variant; /* SA_MIN_BINARY / SA_MIN_DECIMAL */
  uint32_t* d_out;
  uint64_t out_capacity;
  uint64_t* d_out_off; /* n + 1 entries */
  int* d_status;
} sa_minimize_batch;

size_t sa_minimize_workspace_bytes(uint32_t n, uint64_t total_len, uint32_t max_len, uint32_t k,
                                   uint32_t w, int variant);
uint64_t sa_minimize_out_capacity(uint32_t n, uint64_t total_len, uint32_t k, uint32_t w);

/* Only enqueues work on `stream` when the batch holds reads (every sequence at most 2048 windows long) or
 * ONE sequence of up to 32 M windows; other batches (several long sequences) synchronise the stream once
 * inside the segmented sort they use. */
int sa_minimize_batch_device(int device, void* stream, const sa_minimize_batch* batch,
                             void* d_workspace, size_t workspace_bytes);

/* Minimizers, host buffers, sequences split over n_devices GPUs by length (0 = all visible).
 * out_triples[s] receives a malloc'ed array of out_counts[s] triples (free with sa_free); when the call
 * fails, every entry of out_triples is NULL. */
int sa_minimize_batch_host(const char* const* seqs, const uint32_t* lens, size_t n, uint32_t k,
                           uint32_t w, int variant, int n_devices, uint32_t** out_triples,
                           uint64_t* out_counts);

/* ---------------------------------------------------------------------------------------------
 * Seed chaining, device-resident batch: what the reference's mapper does per read between
 * MinimizeBinary and Align (reference src/mapper.cpp: getMinimizer :172-196 index,
 * ignoreTooFrequentMinimizer :224-232, matchMinimizer :234-290, LongestIncreasingSubsequence :292-352,
 * called from mapping() :410-419).
 *
 * d_ref: the reference's n_ref minimizer triples, d_reads / d_read_off: the reads' triples, both as
 * sa_minimize_batch_device writes them (sorted by value, position, strand).  frequent_fraction is the
 * mapper's -f: the (int)((int)distinct * f) reference values with the largest (occurrences, value) are
 * ignored.  d_out[5*r ..] = {chain length, t_begin, t_end, q_begin, q_end} of read r: first / last
 * reference and read positions on the longest chain of matches with strictly increasing reference
 * positions (all 0 without a match; with a single match q_end = q_begin, where the reference leaves
 * q_end uninitialised).  d_n_matches (optional): matched pairs per read.
 * The number of matches decides the workspace: the call synchronises the stream once to read it, sets
 * *needed_bytes (optional) and returns SA_ERR_WORKSPACE if the workspace is smaller - call again with a
 * workspace of that size. */
typedef struct sa_chain_batch {
  const uint32_t* d_ref;
  uint32_t n_ref;
  float frequent_fraction;
  const uint32_t* d_reads;
  const uint64_t* d_read_off; /* n_reads + 1 entries */
  uint32_t n_reads;
  uint64_t n_read_triples; /* == read_off[n_reads] */
  int32_t* d_out;
  uint64_t* d_n_matches;
} sa_chain_batch;

int sa_chain_batch_device(int device, void* stream, const sa_chain_batch* batch, void* d_workspace,
                          size_t workspace_bytes, size_t* needed_bytes);

/* ---------------------------------------------------------------------------------------------
 * Sequence file ingest on the device: the bytes of a FASTA / FASTQ file, copied to device memory as they
 * are, become the sequence blob + offsets the batch calls above read.  Takes the place of the parser calls
 * in the reference's mapper (reference src/mapper.cpp:532-558, bioparser FastaParser / FastqParser filling
 * Sequence{name, data, quality}, :21-37); host code no longer touches the bases.
 *
 * FASTA: a '>' outside a header line starts a record, its header runs to the next '\n', its name is the
 * header up to the first white-space byte, its data are the non-white-space bytes up to the next record.
 * FASTQ: four lines per record - '@' header, bases, '+' line, qualities (as many as bases); files that split
 * a record's bases over several lines are rejected.  A file that breaks these rules (no leading '>', empty
 * name, record without bases, ...) fails with SA_ERR_FORMAT where bioparser throws.
 *
 * d_file: n_bytes of file content, 16-byte aligned.  Outputs, all device pointers:
 *   d_seq              the bases of all records back to back (16-byte aligned, seq_capacity bytes; n_bytes
 *                      always suffices - add the 8 bytes of slack the batch calls want)
 *   d_seq_off[r]       first byte of record r in d_seq; max_records + 1 entries, entry n_records = n_bases
 *   d_name_start[r], d_name_len[r]   the record's name as a span of the file
 *   d_qual_start[r]    FASTQ: first byte of the quality line in the file (its length is the record's)
 * The call synchronises the stream (twice), then fills n_records, n_bases, min_len and max_len.  With
 * n_records > max_records or n_bases > seq_capacity it returns SA_ERR_CAPACITY after setting n_records and
 * n_bases, before writing any output: call again with larger buffers. */
enum { SA_FMT_FASTA = 0, SA_FMT_FASTQ = 1 };

typedef struct sa_ingest_batch {
  const uint8_t* d_file;
  uint64_t n_bytes;
  int format; /* SA_FMT_FASTA / SA_FMT_FASTQ */
  uint8_t* d_seq;
  uint64_t seq_capacity;
  uint64_t* d_seq_off;
  uint64_t* d_name_start;
  uint32_t* d_name_len;
  uint64_t* d_qual_start; /* FASTQ only */
  uint64_t max_records;
  /* results */
  uint64_t n_records;
  uint64_t n_bases;
  uint32_t min_len, max_len; /* shortest / longest record */
} sa_ingest_batch;

size_t sa_ingest_workspace_bytes(uint64_t n_bytes);
int sa_ingest_device(int device, void* stream, sa_ingest_batch* batch, void* d_workspace, size_t workspace_bytes);

/* Two bits per base, for keeping large read sets resident or shipping them over PCIe at a quarter of the
 * size: base i of d_seq goes to bits 2*(i%32).. of d_packed[i/32] with the codes of the reference's mapNct
 * (C 0, A 1, T 2, G 3 in either case, src/minimizer_binary.cpp:13-33); bit i%32 of d_other[i/32] is set where
 * byte i is none of those.  Both arrays have (n_bases + 31) / 32 entries.  sa_unpack2bit_device is the
 * inverse up to case: upper-case ACGT, 'N' where d_other is set.  d_seq must be 16-byte aligned. */
int sa_pack2bit_device(int device, void* stream, const uint8_t* d_seq, uint64_t n_bases, uint64_t* d_packed,
                       uint32_t* d_other);
int sa_unpack2bit_device(int device, void* stream, const uint64_t* d_packed, const uint32_t* d_other,
                         uint64_t n_bases, uint8_t* d_seq);

/* Copies n spans of a device blob back to back: span i is d_src[d_src_start[i] .. + d_dst_off[i+1] - d_dst_off[i])
 * and lands at d_dst + d_dst_off[i].  The device form of the substr() calls with which the mapper cuts the
 * chained parts out of read and reference before Align (reference src/mapper.cpp:423-424), so that ingested
 * records can be aligned without visiting the host. */
int sa_gather_spans_device(int device, void* stream, const uint8_t* d_src, const uint64_t* d_src_start,
                           const uint64_t* d_dst_off, uint32_t n, uint8_t* d_dst);

/* The dealing rule of the two host batch calls above, for schedulers and tests (pure host code, no GPU needed):
 * units are ordered by decreasing cost (stable; left in input order when all costs are equal) and dealt to
 * n_shards shards in blocks of `block` units, every block to the shard with the smallest total cost so far.  shard_of[u] = shard of unit u; order (optional, n
 * entries) = the units shard by shard, each shard in the order it processes them.  This is the device batch
 * queue's replacement for the reference's one-task-per-read submission loop (src/mapper.cpp:665-675). */
int sa_shard_units(const uint64_t* cost, size_t n, int n_shards, uint32_t block, uint32_t* shard_of,
                   uint32_t* order);

/* removeDuplicate on the device (reference src/utility.cpp:8-26): sorts n triples in place by
 * (value, position, strand) and writes the distinct ones to out (capacity n); *out_n = their number. */
int sa_remove_duplicate_host(uint32_t* triples, size_t n, uint32_t* out, size_t* out_n);

#ifdef __cplusplus
}
#endif
#endif
