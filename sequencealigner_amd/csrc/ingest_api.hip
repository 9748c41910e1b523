// FASTA / FASTQ ingest on the device: the bytes of a sequence file, already in device memory, are split into
// records and the bases are compacted into one blob + offsets - the layout sa_minimize_batch_device and
// sa_align_batch_device read.  Takes the place of the parser calls of the reference's mapper
// (reference src/mapper.cpp:532-558: bioparser FastaParser / FastqParser filling Sequence{name, data,
// quality}, :21-37).  bioparser is a git submodule the reference does not vendor (.gitmodules), so the
// format rules below restate its documented behaviour for well-formed files:
//   FASTA  a '>' outside a header line starts a record; the header runs to the next '\n'; the record's
//          name is the header up to its first white-space byte; its data are the non-white-space bytes up
//          to the next record.
//   FASTQ  four lines per record: '@' header (name as above), bases, a line starting with '+', qualities
//          (as many as bases).  Records whose bases or qualities are split over several lines are rejected.
//
// Structure: one wave per tile of the file (16 KiB, less for small files).  The wave walks its tile 64 bytes at a time, one byte
// per lane; __ballot turns "is '\n'", "is '>'", "is white space" into 64-bit lane masks that live in
// scalar registers, and the header / line-phase state machine runs on those masks: for FASTA the
// "inside a header" latch (set by '>', cleared by '\n') is the carry chain of one 64-bit addition.
// Pass 1 (count) leaves one summary per tile that does not depend on the state the tile is entered in;
// a scan with a non-commutative join gives every tile its entry state, first blob offset and first record
// number; pass 2 (write) repeats the walk with those and writes blob, offsets and name positions.  Kept
// bytes go through a per-wave LDS buffer that keeps the blob's 16-byte phase, so almost all of the blob is
// written with 1 KiB coalesced stores.
#include "sa_internal.h"

#include <rocprim/device/device_scan.hpp>

#include <algorithm>

namespace sa {
namespace {

constexpr uint32_t kWaves = 4;          // waves per block, one tile each
constexpr uint32_t kSlab = 1024;        // bytes of the tile a wave holds in LDS at a time (16 per lane)
constexpr uint32_t kMaxSlabsPerTile = 16;  // tiles are 1 .. 16 slabs: small files still fill the GPU

// error bits (device) -> messages (host)
enum : uint32_t {
  E_FASTA_START = 1u, E_NAME = 2u, E_DATA = 4u, E_AT = 8u, E_PLUS = 16u, E_QUAL = 32u, E_LONG = 64u
};

// What a tile contributes.  FASTA: c[0] / c[1] = bytes kept when the tile is entered outside / inside a
// header line, c[2] / c[3] = records started, a = bit s: inside a header after the tile when entered in
// state s.  FASTQ: c[j] = non-white-space bytes on lines whose index relative to the tile's first line is
// j mod 4, a = number of '\n'.
struct TileSum {
  uint64_t c[4];
  uint64_t a;
};

struct FastaJoin {
  __host__ __device__ TileSum operator()(const TileSum& x, const TileSum& y) const {
    TileSum r;
    r.a = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t mid = (uint32_t)(x.a >> s) & 1u;
      r.c[s] = x.c[s] + y.c[mid];
      r.c[2 + s] = x.c[2 + s] + y.c[2 + mid];
      r.a |= ((y.a >> mid) & 1u) << s;
    }
    return r;
  }
};

struct FastqJoin {
  __host__ __device__ TileSum operator()(const TileSum& x, const TileSum& y) const {
    TileSum r;
#pragma unroll
    for (uint32_t m = 0; m < 4; ++m) r.c[m] = x.c[m] + y.c[(m - (uint32_t)x.a) & 3u];
    r.a = x.a + y.a;
    return r;
  }
};

struct IngestArgs {
  const uint8_t* file;
  uint64_t n;
  uint64_t n_tiles;
  uint32_t slabs;      // slabs per tile
  const TileSum* pre;  // inclusive scan of the tile summaries
  uint8_t* seq;
  uint64_t* seq_off;
  uint64_t* name_start;
  uint32_t* name_len;
  uint64_t* qual_start;
  uint64_t n_records, n_bases;
  uint32_t* err;    // one word of E_* bits
  uint32_t* stats;  // min length, max length
};

__device__ __forceinline__ bool is_space(uint32_t b) { return (b - 9u) < 5u || b == 32u; }

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Lane L fetches bytes [base + 16 L, base + 16 L + 16) of the file; bytes past the end of the file read as
// blanks, which no rule below keeps or reacts to.  The fetch of the next slab is issued before the current
// one is walked and lands in the wave's LDS slab afterwards (put_slab).
__device__ __forceinline__ uint4 fetch_slab(const uint8_t* file, uint64_t n, uint64_t base, uint32_t lane) {
  const uint64_t at = base + 16ull * lane;
  uint4 v = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
  if (at + 16 <= n) {
    v = *reinterpret_cast<const uint4*>(file + at);
  } else if (at < n) {
    uint32_t w[4] = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (at + j < n) w[j >> 2] = (w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | ((uint32_t)file[at + j] << (8 * (j & 3)));
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return v;
}

__device__ __forceinline__ void put_slab(uint4 v, uint32_t lane, uint8_t* slab) {
  __builtin_amdgcn_wave_barrier();  // the previous slab has been read
  *reinterpret_cast<uint4*>(slab + 16 * lane) = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// FASTA, 64 bytes: `inside` (0 / 1) says whether the byte before these lies inside a header line.
struct FastaMasks {
  uint64_t nl, keep, starts;
};
__device__ __forceinline__ FastaMasks fasta_round(uint32_t b, uint32_t& inside) {
  FastaMasks m;
  const uint64_t gt = __ballot(b == '>'), ws = __ballot(is_space(b));
  m.nl = __ballot(b == '\n');
  if (!(gt | m.nl)) {  // no event: all 64 bytes share the state
    m.keep = inside ? 0 : ~ws;
    m.starts = 0;
    return m;
  }
  // latch: after_i = gt_i | (~nl_i & after_{i-1}).  In the sum ~nl + gt + inside bit i generates a carry
  // where gt, propagates one where neither gt nor nl and kills it where nl, so the carry into bit i is
  // "inside a header before byte i".
  const uint64_t x = ~m.nl, sum = x + gt + inside;
  const uint64_t before = sum ^ x ^ gt;
  const uint64_t after = gt | (~m.nl & before);
  inside = (uint32_t)(after >> 63);
  m.keep = ~after & ~ws;
  m.starts = gt & ~before;
  return m;
}

// FASTQ, 64 bytes with at least one line end or line start among them: phase[j] = the bytes on lines with
// index j mod 4, `line` = index of the line the first byte is on, `at_line_start` = 1 if that byte begins
// its line.  (Groups without either lie on one line and are handled by the callers directly.)
struct FastqMasks {
  uint64_t line_start, phase[4];
};
__device__ __forceinline__ FastqMasks fastq_round(uint64_t nl, uint64_t line, uint32_t& at_line_start) {
  FastqMasks m;
  m.line_start = (nl << 1) | at_line_start;
  at_line_start = (uint32_t)(nl >> 63);
  m.phase[0] = m.phase[1] = m.phase[2] = m.phase[3] = 0;
  uint64_t rem = nl, from = ~0ull;  // `from`: the bytes at and after the current line's first byte
  uint32_t ph = (uint32_t)line & 3u;
  while (true) {
    const uint64_t upto = rem ? (((rem & (0 - rem)) << 1) - 1) : ~0ull;  // up to and including its '\n'
    const uint64_t seg = from & upto;
    m.phase[0] |= ph == 0 ? seg : 0;
    m.phase[1] |= ph == 1 ? seg : 0;
    m.phase[2] |= ph == 2 ? seg : 0;
    m.phase[3] |= ph == 3 ? seg : 0;
    if (!rem) break;
    rem &= rem - 1;
    from = ~upto;
    ph = (ph + 1) & 3u;
  }
  return m;
}

template <int FORMAT>
__global__ __launch_bounds__(kWaves * 64) void ingest_count(const uint8_t* file, uint64_t n, uint64_t n_tiles,
                                                            uint32_t slabs_per_tile, TileSum* sums) {
  __shared__ __align__(16) uint8_t slabs[kWaves][kSlab];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t tile = (uint64_t)blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  uint8_t* slab = slabs[wave];
  TileSum t;
  t.c[0] = t.c[1] = t.c[2] = t.c[3] = 0;
  t.a = 0;
  uint32_t inside = 0;
  bool seen_nl = false;
  uint64_t pre_kept = 0, pre_starts = 0;  // FASTA: kept / started before the tile's first '\n'
  uint64_t on_line = 0;                   // FASTQ: non-white-space bytes on the current line so far
  uint4 next = fetch_slab(file, n, tile * slabs_per_tile * kSlab, lane);
  for (uint32_t s = 0; s < slabs_per_tile; ++s) {
    const uint64_t base = (tile * slabs_per_tile + s) * kSlab;
    if (base >= n) break;
    put_slab(next, lane, slab);
    if (s + 1 < slabs_per_tile) next = fetch_slab(file, n, base + kSlab, lane);
#pragma unroll
    for (uint32_t r = 0; r < kSlab / 64; ++r) {
      const uint32_t b = slab[64 * r + lane];
      if (FORMAT == SA_FMT_FASTA) {
        const FastaMasks m = fasta_round(b, inside);
        t.c[0] += __popcll(m.keep);
        t.c[2] += __popcll(m.starts);
        if (!seen_nl) {
          const uint64_t head = m.nl ? (m.nl & (0 - m.nl)) - 1 : ~0ull;
          pre_kept += __popcll(m.keep & head);
          pre_starts += __popcll(m.starts & head);
          seen_nl = m.nl != 0;
        }
      } else {
        const uint64_t nl = __ballot(b == '\n'), text = ~__ballot(is_space(b));
        if (!nl) {
          on_line += __popcll(text);
        } else {  // close every line that ends here
          uint64_t rem = nl, from = ~0ull;
          while (rem) {
            const uint64_t upto = ((rem & (0 - rem)) << 1) - 1;
            on_line += __popcll(text & from & upto);
            const uint32_t ph = (uint32_t)t.a & 3u;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) t.c[j] += ph == j ? on_line : 0;
            on_line = 0;
            t.a += 1;
            rem &= rem - 1;
            from = ~upto;
          }
          on_line = __popcll(text & from);
        }
      }
    }
  }
  if (FORMAT == SA_FMT_FASTA) {
    // entered inside a header, everything before the first '\n' is header text
    t.c[1] = t.c[0] - pre_kept;
    t.c[3] = t.c[2] - pre_starts;
    t.a = inside | ((seen_nl ? inside : 1u) << 1);
  } else {
    const uint32_t ph = (uint32_t)t.a & 3u;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) t.c[j] += ph == j ? on_line : 0;
    if (tile == n_tiles - 1 && file[n - 1] != '\n')
      t.a += 1;  // the last line has no '\n': count it, so that the total is the number of lines
  }
  if (lane == 0) sums[tile] = t;
}

// Name = header text after its first byte up to the first white-space byte.
__device__ __forceinline__ uint32_t name_length(const uint8_t* file, uint64_t n, uint64_t from) {
  uint64_t q = from;
  while (q < n && !is_space(file[q])) ++q;
  return (uint32_t)std::min<uint64_t>(q - from, 0xFFFFFFFFull);
}

template <int FORMAT>
__global__ __launch_bounds__(kWaves * 64) void ingest_write(IngestArgs a) {
  __shared__ __align__(16) uint8_t slabs[kWaves][kSlab];
  __shared__ __align__(16) uint8_t outs[kWaves][kSlab + 32];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t tile = (uint64_t)blockIdx.x * kWaves + wave;
  if (tile >= a.n_tiles) return;
  uint8_t* slab = slabs[wave];
  uint8_t* out = outs[wave];

  TileSum pre;
  pre.c[0] = pre.c[1] = pre.c[2] = pre.c[3] = 0;
  pre.a = 0;
  if (tile) pre = a.pre[tile - 1];
  // the file starts outside a header (FASTA) / on line 0 (FASTQ)
  uint32_t inside = (uint32_t)pre.a & 1u;
  uint64_t line = pre.a;
  uint32_t at_line_start = 1;
  if (FORMAT == SA_FMT_FASTQ && tile) at_line_start = a.file[tile * a.slabs * kSlab - 1] == '\n';
  uint64_t rec = FORMAT == SA_FMT_FASTA ? pre.c[2] : 0;
  const uint64_t first = FORMAT == SA_FMT_FASTA ? pre.c[0] : pre.c[1];  // blob offset of the tile's first kept byte
  if (tile == 0 && lane == 0) {
    a.seq_off[a.n_records] = a.n_bases;
    if (FORMAT == SA_FMT_FASTA && a.file[0] != '>') atomicOr(a.err, E_FASTA_START);
  }

  // out[i] holds blob byte unit + i; unit is a multiple of 16, so full 16-byte groups go out as aligned
  // stores.  The first `head` bytes of the first group belong to the tile before.
  uint32_t head = (uint32_t)first & 15u, fill = head;
  uint64_t unit = first - head;

  uint4 next = fetch_slab(a.file, a.n, tile * a.slabs * kSlab, lane);
  for (uint32_t s = 0; s < a.slabs; ++s) {
    const uint64_t base = (tile * a.slabs + s) * kSlab;
    if (base >= a.n) break;
    put_slab(next, lane, slab);
    if (s + 1 < a.slabs) next = fetch_slab(a.file, a.n, base + kSlab, lane);
#pragma unroll
    for (uint32_t r = 0; r < kSlab / 64; ++r) {
      const uint32_t b = slab[64 * r + lane];
      const uint64_t pos = base + 64 * r + lane;
      uint64_t keep;
      if (FORMAT == SA_FMT_FASTA) {
        const FastaMasks m = fasta_round(b, inside);
        keep = m.keep;
        if (m.starts) {
          if ((m.starts >> lane) & 1u) {
            const uint64_t id = rec + lanes_below(m.starts);
            a.seq_off[id] = unit + fill + lanes_below(keep);
            a.name_start[id] = pos + 1;
            const uint32_t len = name_length(a.file, a.n, pos + 1);
            a.name_len[id] = len;
            if (len == 0) atomicOr(a.err, E_NAME);
          }
          rec += __popcll(m.starts);
        }
      } else {
        const uint64_t nl = __ballot(b == '\n');
        if (!nl && !at_line_start) {  // one line, nothing starts: bases are kept, the rest is skipped
          keep = (line & 3u) == 1 ? ~__ballot(is_space(b)) : 0;
        } else {
          const FastqMasks m = fastq_round(nl, line, at_line_start);
          keep = m.phase[1] & ~__ballot(is_space(b));
          const uint64_t events = m.line_start & ~m.phase[1] & __ballot(pos < a.n);
          if ((events >> lane) & 1u) {
            const uint64_t my_line = line + lanes_below(nl);
            const uint64_t id = my_line >> 2;
            const uint32_t ph = (uint32_t)my_line & 3u;
            if (ph == 0) {
              if (b != '@') atomicOr(a.err, E_AT);
              if (id < a.n_records) {
                a.seq_off[id] = unit + fill + lanes_below(keep);
                a.name_start[id] = pos + 1;
                const uint32_t len = name_length(a.file, a.n, pos + 1);
                a.name_len[id] = len;
                if (len == 0) atomicOr(a.err, E_NAME);
              }
            } else if (ph == 2) {
              if (b != '+') atomicOr(a.err, E_PLUS);
            } else if (id < a.n_records) {
              a.qual_start[id] = pos;
            }
          }
          line += __popcll(nl);
        }
      }
      if (!keep) continue;
      if ((keep >> lane) & 1u) out[fill + lanes_below(keep)] = (uint8_t)b;
      fill += __popcll(keep);
    }
    // write the full 16-byte groups, keep the rest for the next slab
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t full = fill & ~15u;
    for (uint32_t u = 16 * lane; u < full; u += 16 * 64) {
      if (u == 0 && head) {
        for (uint32_t i = head; i < 16; ++i) a.seq[unit + i] = out[i];
      } else {
        *reinterpret_cast<uint4*>(a.seq + unit + u) = *reinterpret_cast<const uint4*>(out + u);
      }
    }
    if (full) {
      const uint32_t tail = fill - full;
      const uint8_t carry = lane < tail ? out[full + lane] : 0;
      __builtin_amdgcn_wave_barrier();
      if (lane < tail) out[lane] = carry;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unit += full;
      fill = tail;
      head = 0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane >= head && lane < fill) a.seq[unit + lane] = out[lane];  // fewer than 16 bytes are left
}

// Per record: length statistics and the format rules that need the finished tables.
template <int FORMAT>
__global__ void ingest_check(IngestArgs a) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_records) return;
  const uint64_t len = a.seq_off[r + 1] - a.seq_off[r];
  if (len == 0) atomicOr(a.err, E_DATA);
  if (len > 0xFFFFFFFFull) atomicOr(a.err, E_LONG);
  atomicMin(&a.stats[0], (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFFull));
  atomicMax(&a.stats[1], (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFFull));
  if (FORMAT == SA_FMT_FASTQ) {
    // the quality line ends before the next record's '@' (or at the end of the file)
    const uint64_t qs = a.qual_start[r];
    uint64_t qe = r + 1 < a.n_records ? a.name_start[r + 1] - 2 : a.n;
    while (qe > qs && is_space(a.file[qe - 1])) --qe;
    if (qe - qs != len) atomicOr(a.err, E_QUAL);
  }
}

// 2 bits per base, 32 bases per thread: C = 0, A = 1, T = 2, G = 3 in either case (the codes of the
// reference's mapNct, src/minimizer_binary.cpp:13-33); `other` flags every byte that is none of those.
__global__ void pack2bit(const uint8_t* seq, uint64_t n, uint64_t* packed, uint32_t* other) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t at = t * 32;
  if (at >= n) return;
  uint32_t w[8];
  if (at + 32 <= n) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(seq + at), v1 = *reinterpret_cast<const uint4*>(seq + at + 16);
    w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
    w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (at + j < n) w[j >> 2] |= (uint32_t)seq[at + j] << (8 * (j & 3));
  }
  uint64_t bits = 0;
  uint32_t flags = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t x = (w[j] >> 1) & 0x03030303u;
    const uint32_t code = x ^ ((~x >> 1) & 0x01010101u);
    bits |= (uint64_t)((code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xFFu) << (8 * j);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t up = (w[j] >> (8 * i)) & 0xDFu;
      const bool base = up == 'A' || up == 'C' || up == 'G' || up == 'T';
      flags |= (base ? 0u : 1u) << (4 * j + i);
    }
  }
  if (at + 32 > n) {  // nothing past the end
    const uint32_t valid = (uint32_t)(n - at);
    bits &= (1ull << (2 * valid)) - 1;
    flags &= (1u << valid) - 1;
  }
  packed[t] = bits;
  other[t] = flags;
}

__global__ void unpack2bit(const uint64_t* packed, const uint32_t* other, uint64_t n, uint8_t* seq) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t at = t * 32;
  if (at >= n) return;
  const uint64_t bits = packed[t];
  const uint32_t flags = other[t];
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w[j] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t code = (uint32_t)(bits >> (2 * (4 * j + i))) & 3u;
      const uint32_t ch = ((flags >> (4 * j + i)) & 1u) ? 'N' : (0x47544143u >> (8 * code)) & 0xFFu;  // "CATG"
      w[j] |= ch << (8 * i);
    }
  }
  if (at + 32 <= n) {
    *reinterpret_cast<uint4*>(seq + at) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(seq + at + 16) = make_uint4(w[4], w[5], w[6], w[7]);
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (at + j < n) seq[at + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
  }
}

// dst[dst_off[i] + j] = src[src_start[i] + j] for j < dst_off[i + 1] - dst_off[i]; one wave per span.
__global__ __launch_bounds__(256) void gather_spans(const uint8_t* src, const uint64_t* src_start,
                                                    const uint64_t* dst_off, uint32_t n, uint8_t* dst) {
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (i >= n) return;
  const uint64_t to = dst_off[i], len = dst_off[i + 1] - to;
  const uint8_t* from = src + src_start[i];
  for (uint64_t j = lane; j < len; j += 64) dst[to + j] = from[j];
}

struct IngestPlan {
  uint64_t n_tiles;
  uint32_t slabs;
  size_t o_sums, o_pre, o_flags, o_tmp, scan_tmp, total;
};

int make_ingest_plan(uint64_t n_bytes, IngestPlan* pl) {
  // at least ~16 K tiles (waves) where the file allows it, tiles of at most 16 KiB
  pl->slabs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_bytes / ((uint64_t)kSlab << 14), 1), kMaxSlabsPerTile);
  const uint64_t tile_bytes = (uint64_t)pl->slabs * kSlab;
  pl->n_tiles = (n_bytes + tile_bytes - 1) / tile_bytes;
  const size_t tiles = (size_t)std::max<uint64_t>(pl->n_tiles, 1);
  size_t t1 = 0, t2 = 0;
  TileSum* p = nullptr;
  if (rocprim::inclusive_scan(nullptr, t1, p, p, tiles, FastaJoin()) != hipSuccess) return SA_ERR_HIP;
  if (rocprim::inclusive_scan(nullptr, t2, p, p, tiles, FastqJoin()) != hipSuccess) return SA_ERR_HIP;
  pl->scan_tmp = std::max(t1, t2);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += round_up(bytes ? bytes : 1, 256);
    return at;
  };
  pl->o_sums = take(tiles * sizeof(TileSum));
  pl->o_pre = take(tiles * sizeof(TileSum));
  pl->o_flags = take(64);
  pl->o_tmp = take(pl->scan_tmp);
  pl->total = o;
  return SA_OK;
}

template <int FORMAT>
void launch_count(hipStream_t stream, const uint8_t* file, uint64_t n, const IngestPlan& pl, TileSum* sums) {
  hipLaunchKernelGGL(ingest_count<FORMAT>, dim3((unsigned)((pl.n_tiles + kWaves - 1) / kWaves)), dim3(kWaves * 64),
                     0, stream, file, n, pl.n_tiles, pl.slabs, sums);
}

template <int FORMAT>
void launch_write(hipStream_t stream, const IngestArgs& a) {
  hipLaunchKernelGGL(ingest_write<FORMAT>, dim3((unsigned)((a.n_tiles + kWaves - 1) / kWaves)), dim3(kWaves * 64),
                     0, stream, a);
  if (a.n_records)
    hipLaunchKernelGGL(ingest_check<FORMAT>, dim3((unsigned)((a.n_records + 255) / 256)), dim3(256), 0, stream, a);
}

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" size_t sa_ingest_workspace_bytes(uint64_t n_bytes) {
  IngestPlan pl;
  if (make_ingest_plan(n_bytes, &pl) != SA_OK) return 0;
  return pl.total;
}

extern "C" int sa_ingest_device(int device, void* stream_, sa_ingest_batch* b, void* d_workspace,
                                size_t workspace_bytes) {
  if (!b) return fail(SA_ERR_ARGUMENT, "batch is NULL");
  b->n_records = b->n_bases = 0;
  b->min_len = b->max_len = 0;
  if (b->format != SA_FMT_FASTA && b->format != SA_FMT_FASTQ) return fail(SA_ERR_ARGUMENT, "unknown format %d", b->format);
  if (b->n_bytes == 0) return SA_OK;  // an empty file has no records
  if (b->n_bytes >= (1ull << 46)) return fail(SA_ERR_ARGUMENT, "file of %llu bytes is too large", (unsigned long long)b->n_bytes);
  if (!b->d_file || ((uintptr_t)b->d_file & 15u)) return fail(SA_ERR_ARGUMENT, "d_file must be a 16-byte aligned device pointer");
  IngestPlan pl;
  if (make_ingest_plan(b->n_bytes, &pl) != SA_OK) return fail(SA_ERR_HIP, "cannot size the ingest workspace");
  if (!d_workspace || workspace_bytes < pl.total)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the %zu needed", workspace_bytes, pl.total);
  SA_HIP_TRY(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)d_workspace;
  TileSum* sums = (TileSum*)(ws + pl.o_sums);
  TileSum* pre = (TileSum*)(ws + pl.o_pre);
  uint32_t* flags = (uint32_t*)(ws + pl.o_flags);  // error bits, min length, max length
  const bool fasta = b->format == SA_FMT_FASTA;

  if (fasta) launch_count<SA_FMT_FASTA>(stream, b->d_file, b->n_bytes, pl, sums);
  else launch_count<SA_FMT_FASTQ>(stream, b->d_file, b->n_bytes, pl, sums);
  SA_HIP_TRY(hipGetLastError());
  size_t t = pl.scan_tmp;
  if (fasta) SA_HIP_TRY(rocprim::inclusive_scan(ws + pl.o_tmp, t, sums, pre, (size_t)pl.n_tiles, FastaJoin(), stream));
  else SA_HIP_TRY(rocprim::inclusive_scan(ws + pl.o_tmp, t, sums, pre, (size_t)pl.n_tiles, FastqJoin(), stream));
  TileSum total;
  const uint32_t init[3] = {0u, 0xFFFFFFFFu, 0u};
  SA_HIP_TRY(hipMemcpyAsync(flags, init, sizeof(init), hipMemcpyHostToDevice, stream));
  SA_HIP_TRY(hipMemcpyAsync(&total, pre + (pl.n_tiles - 1), sizeof(TileSum), hipMemcpyDeviceToHost, stream));
  SA_HIP_TRY(hipStreamSynchronize(stream));

  if (fasta) {
    b->n_bases = total.c[0];
    b->n_records = total.c[2];
  } else {
    if (total.a % 4)
      return fail(SA_ERR_FORMAT, "FASTQ: %llu lines is not a multiple of four (records split over lines are not supported)",
                  (unsigned long long)total.a);
    b->n_bases = total.c[1];
    b->n_records = total.a / 4;
  }
  if (b->n_records > b->max_records || b->n_bases > b->seq_capacity)
    return fail(SA_ERR_CAPACITY, "%llu records / %llu bases do not fit %llu / %llu",
                (unsigned long long)b->n_records, (unsigned long long)b->n_bases,
                (unsigned long long)b->max_records, (unsigned long long)b->seq_capacity);
  if (!b->d_seq || !b->d_seq_off || !b->d_name_start || !b->d_name_len || (!fasta && !b->d_qual_start))
    return fail(SA_ERR_ARGUMENT, "NULL output pointer");
  if ((uintptr_t)b->d_seq & 15u) return fail(SA_ERR_ARGUMENT, "d_seq must be 16-byte aligned");

  IngestArgs a{b->d_file, b->n_bytes, pl.n_tiles, pl.slabs, pre, b->d_seq, b->d_seq_off, b->d_name_start, b->d_name_len,
               b->d_qual_start, b->n_records, b->n_bases, flags, flags + 1};
  if (fasta) launch_write<SA_FMT_FASTA>(stream, a);
  else launch_write<SA_FMT_FASTQ>(stream, a);
  SA_HIP_TRY(hipGetLastError());
  uint32_t got[3];
  SA_HIP_TRY(hipMemcpyAsync(got, flags, sizeof(got), hipMemcpyDeviceToHost, stream));
  SA_HIP_TRY(hipStreamSynchronize(stream));
  if (got[0]) {
    const uint32_t e = got[0];
    return fail(SA_ERR_FORMAT, "invalid %s file:%s%s%s%s%s%s%s", fasta ? "FASTA" : "FASTQ",
                e & E_FASTA_START ? " does not start with '>';" : "", e & E_NAME ? " record without a name;" : "",
                e & E_DATA ? " record without bases;" : "", e & E_AT ? " header line not starting with '@';" : "",
                e & E_PLUS ? " third line not starting with '+';" : "",
                e & E_QUAL ? " quality and base counts differ;" : "", e & E_LONG ? " record above 4 Gbases;" : "");
  }
  if (b->n_records) {
    b->min_len = got[1];
    b->max_len = got[2];
  }
  return SA_OK;
}

extern "C" int sa_pack2bit_device(int device, void* stream_, const uint8_t* d_seq, uint64_t n_bases,
                                  uint64_t* d_packed, uint32_t* d_other) {
  if (n_bases == 0) return SA_OK;
  if (!d_seq || !d_packed || !d_other || ((uintptr_t)d_seq & 15u))
    return fail(SA_ERR_ARGUMENT, "NULL pointer or d_seq not 16-byte aligned");
  SA_HIP_TRY(hipSetDevice(device));
  const uint64_t words = (n_bases + 31) / 32;
  hipLaunchKernelGGL(pack2bit, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, d_seq,
                     n_bases, d_packed, d_other);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

extern "C" int sa_unpack2bit_device(int device, void* stream_, const uint64_t* d_packed, const uint32_t* d_other,
                                    uint64_t n_bases, uint8_t* d_seq) {
  if (n_bases == 0) return SA_OK;
  if (!d_seq || !d_packed || !d_other || ((uintptr_t)d_seq & 15u))
    return fail(SA_ERR_ARGUMENT, "NULL pointer or d_seq not 16-byte aligned");
  SA_HIP_TRY(hipSetDevice(device));
  const uint64_t words = (n_bases + 31) / 32;
  hipLaunchKernelGGL(unpack2bit, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     d_packed, d_other, n_bases, d_seq);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

extern "C" int sa_gather_spans_device(int device, void* stream_, const uint8_t* d_src, const uint64_t* d_src_start,
                                      const uint64_t* d_dst_off, uint32_t n, uint8_t* d_dst) {
  if (n == 0) return SA_OK;
  if (!d_src || !d_src_start || !d_dst_off || !d_dst) return fail(SA_ERR_ARGUMENT, "NULL pointer");
  SA_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(gather_spans, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream_, d_src, d_src_start,
                     d_dst_off, n, d_dst);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}
