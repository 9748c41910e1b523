// Device-resident batched Align(): workspace planning, kernel choice and launches.
#include "align_common.h"
#include "align_lane_i32.h"
#include "align_lane_pk16.h"
#include "align_wave_i32.h"
#include "align_wave_pk16.h"
#include "align_wave_coop.h"
#include "align_wave_segments.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

namespace sa {

std::string& last_error() {
  static thread_local std::string s;
  return s;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

Options& options() {
  static Options* o = [] {
    Options* x = new Options();
    auto num = [](const char* name, long dflt) {
      const char* e = getenv(name);
      return e ? atol(e) : dflt;
    };
    x->force_i32 = getenv("SA_FORCE_I32") != nullptr;
    if (const char* f = getenv("SA_FORCE_FAMILY")) x->force_family = !strcmp(f, "wave") ? 2 : 1;
    x->force_coop = num("SA_FORCE_COOP", -1);
    x->no_tables = getenv("SA_NO_TABLES") != nullptr;
    x->coop_c = num("SA_COOP_C", 0);
    x->coop_waves = num("SA_COOP_WAVES", 8192);
    x->coop_pk = num("SA_COOP_PK", -1);
    x->no_overlap = num("SA_NO_OVERLAP", 0);
    x->coop_seg_rows = num("SA_COOP_SEG_ROWS", -1);
    x->coop_async = num("SA_COOP_ASYNC", 0);
    x->pk16_k = num("SA_PK16_K", 0);
    x->min_chunk = num("SA_MIN_CHUNK", 0);
    x->no_buckets = num("SA_NO_BUCKETS", 0);
    x->no_segments = num("SA_NO_SEGMENTS", 0);
    x->seg_rows = num("SA_SEG_ROWS", 0);
    x->seg_cap = num("SA_SEG_CAP", 0);
    x->seg_spec = num("SA_SEG_SPEC", 2);
    x->spin_limit = num("SA_SPIN_LIMIT", 1 << 24);
    x->fake_devices = num("SA_FAKE_DEVICES", 0);
    return x;
  }();
  return *o;
}

namespace {

constexpr int kLaneI32K = 32;     // target rows per register strip of the 32-bit lane kernel
constexpr uint32_t kBlock = 256;  // threads per workgroup of the lane kernels
constexpr size_t kFixedWorkspace = 1024;  // head of the workspace: AlphabetInfo

enum class Family { LaneI32, LanePk16, WaveI32 };

struct Plan {
  Family family;
  int K;                      // rows per strip
  uint32_t nstrips;
  uint32_t t_pad;             // nstrips * K
  uint32_t bias;              // packed kernels: score offset
  uint32_t pairs_per_unit;    // pairs per wave
  uint64_t dirs_unit_stride;  // dwords per unit
  uint64_t brow_unit_stride;  // ints per unit
  uint64_t qpk_unit_words, tpk_unit_words;
  bool tables;                // packed kernels: the score-table form may be used (mismatch >= gap, bytes)
  uint32_t entry_match, entry_mismatch;
  int wave_c;                 // wavefront kernels: query columns per lane (8 or 16)
  bool coop;                  // wavefront kernels: one wave per (pair, strip), all resident
  bool wave_pk;               // wavefront kernels: packed 16-bit scores, two pairs per wave
  size_t bytes_per_block;     // workspace bytes per workgroup (4 units)
  uint32_t units_per_block;   // wavefronts per workgroup (the workspace is laid out per workgroup)
  uint32_t pairs_per_block;
  const char* name;
};

// Can the packed 16-bit lane kernel reproduce this batch exactly?  (Preconditions at the top of
// align_lane_pk16.h.)  Fills K / bias on success.
bool lane_pk16_applicable(const sa_align_batch& b, int K, uint32_t* bias_out) {
  if (K > 64) return false;
  const long match = b.match, mismatch = b.mismatch, gap = b.gap;
  if (gap > 0 || mismatch > match || match < gap) return false;
  const long Q = b.max_qlen, T = b.max_tlen;
  if (Q == 0 || T == 0) return false;
  const long t_pad = (T + K - 1) / K * K;
  long lo, hi, window;
  if (b.type == SA_LOCAL) {
    if (gap >= 0 || mismatch >= 0 || match <= 0) return false;
    lo = std::min(gap, mismatch);
    hi = match * std::min(Q, t_pad);
    window = 511;  // the best-cell key keeps 6 bits for the row inside the strip
  } else {
    lo = (b.type == SA_GLOBAL ? gap * (Q + t_pad) : gap * Q) + std::min(gap, std::min(mismatch, 0L));
    hi = std::max(match, 0L) * std::min(Q, t_pad);
    window = 0x7FFF;
  }
  const long bias = 1 - lo;  // S(x) = x + bias >= 1 for every value formed
  if (hi + bias + (match - gap) + 1 > window) return false;
  if (Q >= 0x7FFF || t_pad + 64 >= 0x7FFF) return false;  // packed row/column indices
  *bias_out = (uint32_t)bias;
  return true;
}

// Can the packed 16-bit wavefront kernel reproduce this batch exactly?  (Preconditions at the top of
// align_wave_pk16.h; the matrix it sweeps is padded to whole strips of columns.)
bool wave_pk16_applicable(const sa_align_batch& b, uint32_t* bias_out) {
  const long match = b.match, mismatch = b.mismatch, gap = b.gap;
  if (gap > 0 || mismatch > match || match < gap) return false;
  const long Q = b.max_qlen, T = b.max_tlen;
  if (Q == 0 || T == 0) return false;
  if (b.type == SA_LOCAL) {
    // scores are kept times 16: values from min(gap, mismatch) (a candidate next to a zero) to the best score
    // possible, plus the (match - gap) the diagonal candidate is formed with, in 11 bits
    if (gap >= 0 || mismatch >= 0) return false;
    const long bias = 1 - std::min(gap, mismatch);
    const long hi = std::max(match, 0L) * std::min(Q, T);
    if (hi + bias + (match - gap) + 1 > 0x7FF) return false;
    *bias_out = (uint32_t)bias;
    return true;
  }
  // Global / Semiglobal: scores relative to per-lane bases (align_wave_pk16.h) - any length, as long as
  // neighbouring cells differ by little and the scores themselves fit the 24 bits of a boundary entry
  const long step = std::max(-gap, match - gap);
  if (step > kWavePkMaxStep || match - mismatch > 4096) return false;
  const long worst = std::max(std::max(std::labs(match), std::labs(mismatch)), std::labs(gap));
  if (worst * (Q + T + 2 * 64 * kWavePkCols) >= (1L << 23)) return false;
  *bias_out = kWavePkCenter;
  return true;
}

// Rows per strip of the packed lane kernel, 48 or 64.  An estimate of the fill's time decides:
//   (rows the strips cover + 7 per strip) x (rounds of the launch's wavefronts),
// the last strip having 16, 32 or 48 rows either way, every strip paying about seven rows' worth for reading and
// writing the boundary row and for its own start, strips of 64 rows costing 3 % more per row; the GPU holds 3 / 2
// (Local) or 2 / 1 (Global, Semiglobal) wavefronts per SIMD with strips of 48 / 64 rows, and a round with 1 / 2 / 3
// wavefronts on its fullest SIMD takes 1.22 / 2.1 / 3 times a lone wavefront's time.  Measured, strips of 48
// against 64 (ms): 10^6 Local pairs of 150 x 300 9.48 / 9.35 (what the 0.96 below is fitted to), of 150 x 150
// 5.23 / 5.18, of 100 x 200 4.48 / 4.51; 4 10^5 of 150 x 100 2.38 / 2.11; 3 10^5 Global 250 x 400 7.96 / 8.48;
// 10^5 Semiglobal 500 x 600 8.38 / 7.82 - the estimate picks the faster in all but the third (0.6 % apart).
// Strips of 32 rows: 9.8 ms on the first.  The option "pk16_k" pins the height.
int pk16_strip_rows(const sa_align_batch& b) {
  const long v = options().pk16_k;
  if (v == 32 || v == 48 || v == 64) return (int)v;
  const uint32_t t = b.max_tlen ? b.max_tlen : 1;
  const uint64_t units = ((uint64_t)b.n + 127) / 128;
  auto estimate = [&](uint32_t k, double per_row, uint32_t per_simd) {
    const uint32_t nfull = t / k, rem = t - nfull * k;
    const double unit = (nfull * k + (rem + 15) / 16 * 16) * per_row + 7.0 * (nfull + (rem ? 1 : 0));
    const double round_time[4] = {0.0, 1.22, 2.1, 3.0};
    const uint64_t resident = 1024ull * per_simd, full = units / resident, rest = units - full * resident;
    return unit * (full * round_time[per_simd] + round_time[(rest + 1023) / 1024]);
  };
  const bool local = b.type == SA_LOCAL;
  uint32_t unused = 0;
  return (0.96 * estimate(64, 1.03, local ? 2 : 1) < estimate(48, 1.0, local ? 3 : 2) &&
          lane_pk16_applicable(b, 64, &unused))
             ? 64
             : 48;
}

// pk_coop_passes: the caller's workspace holds so few pairs of this batch at a time that the one-wavefront-per-
// two-pairs kernel would leave most of the GPU idle: take the packed cooperative kernel whatever the batch's size
Plan make_plan(const sa_align_batch& b, bool allow_coop = true, bool pk_coop_passes = false) {
  Plan pl{};
  const uint32_t max_q = b.max_qlen ? b.max_qlen : 1, max_t = b.max_tlen ? b.max_tlen : 1;
  uint32_t bias = 0;
  const int K16 = pk16_strip_rows(b);
  const Options& opt = options();
  const bool force_i32 = opt.force_i32 != 0;
  const long force_family = opt.force_family;  // 1 lane, 2 wave
  // The lane kernels put 64 or 128 pairs on a wavefront, the wavefront kernels one pair (or one strip of
  // a pair): which family fills the GPU better depends on the batch.  Rough throughput estimates (TCUPS,
  // measured on MI355X) decide; pairs longer than 2048 always take the wavefront kernels.
  // (Local there needs gap <= 0, mismatch <= 0 and scores below 2^27; other scorings take the general
  // lane kernel.)  SA_FORCE_FAMILY=lane|wave and SA_FORCE_COOP=0|1 are test hooks that pin the choice.
  const bool pk16_ok = !force_i32 && lane_pk16_applicable(b, K16, &bias);
  bool want_wave;
  if (force_family) {
    want_wave = force_family == 2;
  } else if (max_q > 2048 || max_t > 2048) {
    want_wave = true;
  } else {
    // a lane kernel's wavefront delivers its rate only with all of its 128 / 64 lanes' pairs present
    const double lane_rate = std::min(b.n / (1024.0 * (pk16_ok ? 128.0 : 64.0)), 1.0) * (pk16_ok ? 5.3 : 0.9);
    const double strip_eff = max_q / (std::ceil(max_q / 1024.0) * 1024.0) * max_t / (max_t + 63.0);
    uint32_t unused_bias = 0;
    const bool wpk = !force_i32 && wave_pk16_applicable(b, &unused_bias);
    const double wave_rate = std::min(b.n / (wpk ? 6144.0 : 2048.0), 1.0) *
                             (wpk ? (b.type == SA_LOCAL ? 4.0 : 5.6) : 3.0) * strip_eff;
    want_wave = wave_rate > lane_rate;
  }
  if (b.type == SA_LOCAL &&
      (b.gap > 0 || b.mismatch > 0 ||
       (long)std::max(b.match, 0) * std::min(max_q, max_t) >= (1L << 27)))
    want_wave = false;
  if (want_wave) {
    // Few long pairs: spread every pair's strips over as many wavefronts (narrow strips, all resident).
    // columns per lane of the cooperative kernel: 4 (more, shorter steps in flight) while all wavefronts of
    // the batch stay resident, else 8; SA_COOP_C=4|8 pins it
    const long coop_opt = opt.coop_c;
    const long force_coop = opt.force_coop;
    // Measured on MI355X (tools/cooppass.py): the one-wavefront-per-pair kernels need about a thousand wavefronts
    // before they use the GPU (256 pairs of 10 kbp x 11 kbp: 48 ms, as long as 2048 pairs), the cooperative
    // Global / Semiglobal kernel, run in passes of up to 8192 wavefronts, fills at 2.3 TCUPS whatever the number
    // of pairs (256 pairs: 12 ms, 1024: 45 ms, 2048: 91 ms; 64 pairs of 30 kbp x 33 kbp 27 against 316 ms).
    // (the cooperative Local kernel, an older design: 256 pairs of 10 kbp 16 against 47 ms, 512 pairs 34 against 47)
    const uint64_t coop_pairs = b.type == SA_LOCAL ? 512 : 768;
    // Queries of a few hundred columns gain as well while the pairs are few (tools/cooppass.py: 100 pairs of
    // 800 x 50 000 3.7 against 24 ms, 100 pairs of 1000 x 1100 0.18 against 0.58 ms, 400 pairs 0.26 against 0.58).
    // (Local, the older cooperative kernel: 100 pairs of 1000 x 1100 0.32 against 0.69 ms, of 800 x 50 000 9.3
    // against 28.8 ms)
    const uint32_t coop_min_q = 256;
    const bool coop_wanted =
        allow_coop && (force_coop >= 0 ? force_coop != 0 : (b.n <= coop_pairs && max_q > coop_min_q));
    int coop_c = 0;
    if (b.type == SA_LOCAL) {
      // columns per lane of the cooperative Local kernel: 4 (more, shorter steps in flight) while every strip of
      // the batch gets a SIMD of its own, else 8 (400 pairs of 1000 x 1100: 0.41 against 0.48 ms); the option
      // "coop_c" = 4 | 8 pins it
      coop_c = coop_opt == 4 || coop_opt == 8 ? (int)coop_opt : ((uint64_t)b.n * ((max_q + 255) / 256) <= 1024 ? 4 : 8);
    } else {
      // Global / Semiglobal: 2, 4 or 8 columns per lane, among the widths that still give every strip of the
      // batch a SIMD of its own.  A pair takes (target rows x time of a step) + (strips x time between the
      // starts of neighbouring strips), see align_wave_coop.h; both times grow with the width (measured on
      // MI355X: 68 / 95 / 155 ns and 8.8 / 11.6 / 17.1 us), the number of strips falls with it.
      double best_time = 0;
      for (int c : {2, 4, 8}) {
        const uint64_t strips = (max_q + 64 * c - 1) / (64 * c);
        if ((uint64_t)b.n * strips > 1024) continue;
        const double step_ns = c == 2 ? 68 : c == 4 ? 95 : 155, between_ns = c == 2 ? 8800 : c == 4 ? 11600 : 17100;
        const double time = max_t * step_ns + strips * between_ns;
        if (!coop_c || time < best_time) {
          coop_c = c;
          best_time = time;
        }
      }
      if (!coop_c) coop_c = 8;
      if (coop_opt == 2 || coop_opt == 4 || coop_opt == 8) coop_c = (int)coop_opt;
    }
    if (!coop_c) coop_c = 8;
    uint32_t wbias = 0;
    const bool wpk_ok = !force_i32 && wave_pk16_applicable(b, &wbias);
    // The packed cooperative kernel (two pairs per wavefront, Global / Semiglobal): when the 32-bit one cannot
    // give every strip of the batch a SIMD of its own the fill is bound by throughput, and the packed form
    // needs about half the instructions per cell.  Fewer pairs are bound by the chain of strips and keep the
    // 32-bit kernel's narrow strips.
    bool pk_coop = false;
    if (b.type != SA_LOCAL && wpk_ok && b.n >= 2 && max_q > coop_min_q && allow_coop) {
      const bool many = (uint64_t)b.n * ((max_q + 511) / 512) > 1024;
      // (tools/pkcoop.py, pairs of 10 kbp x 11 kbp: 256 pairs 8.0 ms against 13.4 on the 32-bit cooperative kernel
      // and 40 with a wavefront per two pairs; 1024 pairs 18 against 47 and 42; 2048 pairs 35 against 93 and 43)
      // Long queries: a wavefront per two pairs runs in rounds of kWavePkResident wavefronts, and a round that is
      // not full takes the time of its fullest SIMD (one / two / three wavefronts: 0.41 / 0.70 / 1 of a full
      // round's); the cooperative form, its strips in segments, shares the SIMDs evenly at about the rate of full
      // rounds (tools/pkcoop.py, call: 3000 pairs of 10 kbp x 11 kbp 48 against 69 ms, 6144 pairs 90 against 87;
      // 9000 pairs of 3000 x 3300 14.5 against 16.6, 6144 pairs 10.4 against 10.3; 8000 pairs of 2000 x 2200 6.4
      // against 6.9; 30 000 pairs of 1200 x 1300 13.8 against 13.0 - its traceback does not run under the fill)
      const double rounds = (((uint64_t)b.n + 1) / 2) / (double)kWavePkResident, part = rounds - std::floor(rounds);
      const double alone = std::floor(rounds) + (part == 0 ? 0.0 : part <= 1 / 3.0 ? 0.41 : part <= 2 / 3.0 ? 0.70 : 1.0);
      const bool few_long = force_coop < 0 && max_q >= 1800 && 1.03 * rounds < alone;
      pk_coop = opt.coop_pk >= 0 ? (opt.coop_pk != 0 && (coop_wanted || pk_coop_passes))
                                 : (((coop_wanted || few_long) && many) || (pk_coop_passes && force_coop != 0));
    }
    if (pk_coop) {
      // columns per lane: 8, 10, 12 or 16 - the width with which the launch's wavefronts spread most evenly over
      // the SIMDs (time ~ steps x wavefronts on the fullest SIMD x instructions of a step); "coop_c" pins it
      const uint64_t units = ((uint64_t)b.n + 1) / 2;
      double best_time = 0;
      coop_c = 0;
      // (more strips than the GPU holds wavefronts: they run in segments and share the SIMDs evenly)
      for (int c : {8, 10, 12, 16}) {
        const uint64_t strips = (max_q + 64 * c - 1) / (64 * c);
        const bool shared = units * strips > kWavePkResident;
        const double per_simd = shared ? units * strips / 1024.0 : std::ceil(units * strips / 1024.0);
        const double time = (max_t + 63.0 + (shared ? 0.0 : strips * 90.0)) * per_simd * (9.0 * c + 30.0);
        if (!coop_c || time < best_time) {
          coop_c = c;
          best_time = time;
        }
      }
      if (coop_opt == 8 || coop_opt == 10 || coop_opt == 12 || coop_opt == 16) coop_c = (int)coop_opt;
    }
    const uint64_t strips8 = (max_q + 64 * coop_c - 1) / (64 * coop_c);
    // passes of as many pairs as the limit on a pass's wavefronts allows, at least one
    const uint64_t coop_waves = (uint64_t)std::max(64L, std::min(8192L, (long)opt.coop_waves));
    pl.coop = (coop_wanted || pk_coop) && strips8 <= coop_waves && strips8 < 65535;
    pk_coop = pk_coop && pl.coop;
    // A small batch of short queries is bound by the time of one step, not by throughput: 4 or 8 columns
    // per lane instead of 16 put more lanes of the wavefront to work and shorten the step.
    const bool narrow = !pl.coop && max_q <= 512 && b.n <= 4096;
    pl.wave_c = pl.coop ? coop_c : narrow ? (max_q <= 256 ? 4 : 8) : 16;
    pl.wave_pk = pk_coop || (!pl.coop && !narrow && wpk_ok);
    const uint32_t strip = 64 * pl.wave_c;
    pl.family = Family::WaveI32;
    pl.name = pk_coop ? "wave_pk16_coop" : pl.coop ? "wave_i32_coop" : pl.wave_pk ? "wave_pk16" : "wave_i32";
    pl.K = pl.wave_c;
    pl.bias = wbias;
    pl.nstrips = (max_q + strip - 1) / strip;
    pl.pairs_per_unit = pl.wave_pk ? 2 : 1;
    pl.dirs_unit_stride = (uint64_t)pl.nstrips * (max_t + 64) * 64 *
                          (pl.wave_pk ? 8 : pl.wave_c == 16 ? 4 : pl.wave_c == 8 ? 2 : 1) / 4;
    if (pl.coop && !pl.wave_pk && b.type != SA_LOCAL)  // one dword per lane per 16 / C steps
      pl.dirs_unit_stride = (uint64_t)pl.nstrips * (coop_steps_padded(max_t) * pl.wave_c / 16) * 64;
    pl.brow_unit_stride = (round_up((size_t)max_t + 1, 64) + 64) * 2 * (pl.coop ? 2 : 1);  // 64-bit entries
    if (pk_coop) pl.brow_unit_stride += (uint64_t)pl.nstrips * kWaveStateSlots * 64;  // registers between segments
  } else if (pk16_ok) {
    pl.family = Family::LanePk16;
    pl.name = "lane_pk16";
    pl.K = K16;
    pl.bias = bias;
    pl.nstrips = (max_t + K16 - 1) / K16;
    pl.t_pad = pl.nstrips * K16;
    const uint32_t ng = (K16 + 15) / 16;
    pl.pairs_per_unit = 128;
    pl.dirs_unit_stride = (uint64_t)pl.nstrips * ((max_q + kColBlock - 1) / kColBlock) * ng * 1024;
    pl.brow_unit_stride = (uint64_t)(max_q + 1) * 64;
    pl.qpk_unit_words = (uint64_t)max_q * 64;
    pl.tpk_unit_words = (uint64_t)pl.t_pad * 64;
    const int sh = (b.type == SA_LOCAL) ? 6 : 0;
    const long em = ((long)b.match - b.gap) << sh, ex = ((long)b.mismatch - b.gap) << sh;
    pl.tables = ex >= 0 && em <= 255 && !opt.no_tables;
    pl.entry_match = (uint32_t)em;
    pl.entry_mismatch = (uint32_t)(ex >= 0 ? ex : 0);
  } else {
    pl.family = Family::LaneI32;
    pl.name = "lane_i32";
    pl.K = kLaneI32K;
    pl.nstrips = (max_t + kLaneI32K - 1) / kLaneI32K;
    pl.t_pad = pl.nstrips * kLaneI32K;
    pl.pairs_per_unit = 64;
    pl.dirs_unit_stride = (uint64_t)pl.nstrips * max_q * (kLaneI32K / 16) * 64;
    pl.brow_unit_stride = (uint64_t)(max_q + 1) * 64;
  }
  const size_t units = pl.family == Family::WaveI32 ? 1 : kBlock / 64;  // wavefront kernels: 64-thread groups
  pl.units_per_block = (uint32_t)units;
  pl.pairs_per_block = (uint32_t)units * pl.pairs_per_unit;
  pl.bytes_per_block = round_up(units * pl.dirs_unit_stride * 4, 256) +
                       round_up(units * pl.brow_unit_stride * 4, 256) +
                       round_up(units * pl.qpk_unit_words * 4, 256) +
                       round_up(units * pl.tpk_unit_words * 4, 256) +
                       round_up((size_t)pl.pairs_per_block * 4, 256) * 3;
  if (pl.coop)  // per (pair, strip): Local candidate
    pl.bytes_per_block += round_up((size_t)pl.pairs_per_block * pl.nstrips * 12, 256);
  return pl;
}

// ---- optional kernel timing (sa_align_profile) ---------------------------------------------------
struct Span {
  hipEvent_t a, b;
  int cls;
};
std::atomic<bool> g_profile{false};
std::mutex g_spans_mutex;  // batch calls may run on several host threads at once
std::vector<Span> g_spans;

struct Timed {
  hipStream_t stream;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  Timed(hipStream_t s, int c) : stream(s), cls(c) {
    if (!g_profile) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      a = b = nullptr;
      return;
    }
    (void)hipEventRecord(a, stream);
  }
  ~Timed() {
    if (!a) return;
    (void)hipEventRecord(b, stream);
    std::lock_guard<std::mutex> g(g_spans_mutex);
    g_spans.push_back({a, b, cls});
  }
};

// One wavefront per pair sums the pair's run lengths per op (reference src/mapper.cpp:371-399 walks the
// CIGAR text for the same sums).
__global__ __launch_bounds__(256) void cigar_stats_kernel(uint32_t n, const uint32_t* runs, const uint32_t* nruns,
                                                          uint64_t stride, uint32_t* block_len, uint32_t* op_len) {
  const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (p >= n) return;
  const uint32_t nr = nruns[p];
  const uint32_t* first = runs + (uint64_t)(p + 1) * stride - nr;  // runs end at the end of the slot range
  uint32_t sum[4] = {0, 0, 0, 0};
  for (uint32_t x = lane; x < nr; x += 64) {
    const uint32_t run = first[x];
#pragma unroll
    for (uint32_t op = 0; op < 4; ++op) sum[op] += (run & 3u) == op ? run >> 2 : 0u;
  }
#pragma unroll
  for (uint32_t op = 0; op < 4; ++op)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum[op] += (uint32_t)__shfl_xor((int)sum[op], off);
  if (lane == 0) {
    if (block_len) block_len[p] = sum[0] + sum[1] + sum[2] + sum[3];
    if (op_len) {
#pragma unroll
      for (uint32_t op = 0; op < 4; ++op) op_len[4 * (uint64_t)p + op] = sum[op];
    }
  }
}

// Clears a few words (flags, counters) from the stream: a kernel, not hipMemsetAsync, so that a caller may
// capture the batch call into a HIP graph.
__global__ void clear_words(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// Option "coop_async": a call that ran the cooperative kernels does not wait for them; if one of their bounded
// waits ran out, every score of the batch becomes SA_SCORE_GAVE_UP instead.
__global__ void mark_scores_if_gave_up(const int* gave_up, int32_t* score, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) score[i] = SA_SCORE_GAVE_UP;
}

// A second stream for a call: the traceback of the pairs whose fill has finished runs there while the last,
// partly filled round of the fill's wavefronts runs on the caller's stream.  The objects are kept per device
// and lent to one call at a time (batch calls may run on several host threads at once).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t forked = nullptr, joined = nullptr;
};
std::mutex g_side_mutex;
std::vector<SideStream> g_side_free[64];

struct SideLease {
  int device;
  SideStream s;
  bool ok = false;
  explicit SideLease(int dev) : device(dev) {
    if (dev < 0 || dev >= 64) return;
    {
      std::lock_guard<std::mutex> g(g_side_mutex);
      if (!g_side_free[dev].empty()) {
        s = g_side_free[dev].back();
        g_side_free[dev].pop_back();
        ok = true;
        return;
      }
    }
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return;
    if (hipEventCreateWithFlags(&s.forked, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.joined, hipEventDisableTiming) != hipSuccess)
      return;
    ok = true;
  }
  ~SideLease() {
    if (!ok) return;
    std::lock_guard<std::mutex> g(g_side_mutex);
    g_side_free[device].push_back(s);
  }
};

// Wavefronts of a fill launch come and go in rounds (they all take the same time): `resident` run at once, and a
// launch of 2.5 rounds leaves half the GPU idle for the length of a round.  Units (wavefronts) of the launch's
// whole rounds, when the rest is a partial round worth filling with other work; 0: do not split.
uint32_t whole_rounds(uint32_t units, uint32_t resident) {
  if (!resident || units < resident + resident / 8) return 0;
  const uint32_t head = units / resident * resident, tail = units - head;
  if (tail < resident / 10 || tail > resident - resident / 10) return 0;
  return head;
}

int device_cus(int device) {
  static int cus[64] = {};
  if (device < 0 || device >= 64) return 256;
  if (!cus[device]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
    cus[device] = v;
  }
  return cus[device];
}

// The arguments of the part of a pass that starts at its unit u0 and holds `pairs` pairs.
AlignArgs sub_pass(const Plan& pl, const AlignArgs& a, uint32_t u0, uint32_t pairs) {
  AlignArgs x = a;
  const uint32_t p0 = u0 * pl.pairs_per_unit;
  x.pair_base = a.pair_base + p0;
  x.pairs_in_pass = pairs;
  x.dirs = a.dirs + (uint64_t)u0 * pl.dirs_unit_stride;
  x.brow = a.brow + (uint64_t)u0 * pl.brow_unit_stride;
  x.end_score = a.end_score + p0;
  x.end_i = a.end_i + p0;
  x.end_r = a.end_r + p0;
  return x;
}

template <int MODE>
int launch_lane_i32(const AlignArgs& a, hipStream_t stream, bool need_traceback) {
  const uint32_t blocks = (a.pairs_in_pass + kBlock - 1) / kBlock;
  {
    Timed tm(stream, 1);
    hipLaunchKernelGGL((fill_lane_i32<MODE, kLaneI32K>), dim3(blocks), dim3(kBlock), 0, stream, a);
  }
  SA_HIP_TRY(hipGetLastError());
  if (need_traceback) {
    LaneI32Layout<kLaneI32K> lay{a.dirs, a.dirs_unit_stride, a.max_q ? a.max_q : 1};
    {
      Timed tm(stream, 2);
      hipLaunchKernelGGL((traceback_kernel<MODE, LaneI32Layout<kLaneI32K>>), dim3(blocks),
                         dim3(kBlock), 0, stream, a, lay);
    }
    SA_HIP_TRY(hipGetLastError());
  }
  return SA_OK;
}

template <int MODE, int K>
int launch_lane_pk16(const Plan& pl, const AlignArgs& a, uint32_t* qpk, uint32_t* tpk,
                     AlphabetInfo* alpha, hipStream_t stream, bool need_traceback, int overlap_device) {
  const uint32_t units = (a.pairs_in_pass + 127) / 128;
  // Tables form: the traceback reads the pairs' symbols from per-pair records of 2-bit codes.  They live in the
  // three quarters of every unit's target area that the one-byte-per-position form leaves unused, when they fit.
  const uint32_t q_tiles = (a.max_q + kPackTile - 1) / kPackTile, t_tiles = (pl.t_pad + kPackTile - 1) / kPackTile;
  const uint64_t rec_unit_dwords = 128ull * 4 * (q_tiles + t_tiles);
  const bool with_rec = alpha && need_traceback && pl.t_pad % 4 == 0 &&
                        (uint64_t)pl.t_pad / 4 * 64 + rec_unit_dwords <= pl.tpk_unit_words;
  uint32_t* rec = with_rec ? tpk + (uint64_t)pl.t_pad / 4 * 64 : nullptr;
  {
    Timed tm(stream, 0);
    PackArgs pk{a.q, a.q_off, a.t, a.t_off, a.pair_base, a.pairs_in_pass, a.max_q, pl.t_pad,
                q_tiles, t_tiles, qpk, tpk, alpha, 0, rec, pl.tpk_unit_words, 4 * (q_tiles + t_tiles)};
    const dim3 grid((pk.q_tiles + pk.t_tiles + 3) / 4, units);
    hipLaunchKernelGGL(pack_lane_pk16, grid, dim3(256), 0, stream, pk);
    if (alpha) {  // returns at once unless the first launch met a byte the alphabet sample had missed
      pk.again = 1;
      hipLaunchKernelGGL(pack_lane_pk16, dim3(grid.x, std::min(units, 256u)), dim3(256), 0, stream, pk);
    }
  }
  SA_HIP_TRY(hipGetLastError());
  // the fill and the traceback of the units [u0, u0 + nu) of the pass
  auto fill = [&](uint32_t u0, uint32_t nu, hipStream_t st) {
    const AlignArgs x = sub_pass(pl, a, u0, std::min<uint32_t>(nu * 128, a.pairs_in_pass - u0 * 128));
    Pk16Args y{qpk + (uint64_t)u0 * pl.qpk_unit_words, tpk + (uint64_t)u0 * pl.tpk_unit_words, pl.t_pad, pl.bias, nu,
               alpha, pl.entry_match, pl.entry_mismatch};
    Timed tm(st, 1);
    if (alpha)
      hipLaunchKernelGGL((fill_lane_pk16<MODE, K, true>), dim3((nu + 3) / 4), dim3(kBlock), 0, st, x, y);
    else
      hipLaunchKernelGGL((fill_lane_pk16<MODE, K, false>), dim3((nu + 3) / 4), dim3(kBlock), 0, st, x, y);
  };
  auto traceback = [&](uint32_t u0, uint32_t nu, hipStream_t st) {
    AlignArgs x = sub_pass(pl, a, u0, std::min<uint32_t>(nu * 128, a.pairs_in_pass - u0 * 128));
    x.rec = nullptr;
    if (with_rec) {
      x.rec = rec + (uint64_t)u0 * pl.tpk_unit_words;
      x.rec_unit_stride = pl.tpk_unit_words;
      x.rec_stride = 4 * (q_tiles + t_tiles);
      x.rec_q_chunks = q_tiles;
      x.rec_alpha = alpha;
    }
    Timed tm(st, 2);
    hipLaunchKernelGGL((traceback_lane_pk16<MODE, K>), dim3((x.pairs_in_pass + kBlock - 1) / kBlock), dim3(kBlock), 0, st, x);
  };
  // units of the fill's whole rounds (blocks of four wavefronts, as many per CU as their registers allow)
  uint32_t head = 0;
  if (need_traceback && overlap_device >= 0) {
    static int per_cu = 0;  // (the same for every instance of the template on this machine's GPUs)
    if (!per_cu) {
      int v = 0;
      const hipError_t e = alpha ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fill_lane_pk16<MODE, K, true>, kBlock, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fill_lane_pk16<MODE, K, false>, kBlock, 0);
      per_cu = (e == hipSuccess && v > 0) ? v : -1;
    }
    if (per_cu > 0) head = whole_rounds(units, (uint32_t)per_cu * device_cus(overlap_device) * 4);
  }
  if (head) {
    SideLease side(overlap_device);
    if (side.ok) {
      fill(0, head, stream);
      SA_HIP_TRY(hipEventRecord(side.s.forked, stream));
      SA_HIP_TRY(hipStreamWaitEvent(side.s.stream, side.s.forked, 0));
      traceback(0, head, side.s.stream);
      SA_HIP_TRY(hipEventRecord(side.s.joined, side.s.stream));
      fill(head, units - head, stream);
      traceback(head, units - head, stream);
      SA_HIP_TRY(hipStreamWaitEvent(stream, side.s.joined, 0));
      SA_HIP_TRY(hipGetLastError());
      return SA_OK;
    }
  }
  fill(0, units, stream);
  SA_HIP_TRY(hipGetLastError());
  if (need_traceback) {
    traceback(0, units, stream);
    SA_HIP_TRY(hipGetLastError());
  }
  return SA_OK;
}

template <int MODE, int C, bool COOP>
int launch_wave_i32(const AlignArgs& a, const WaveCoop& co, hipStream_t stream, bool need_traceback) {
  // score tables inside the kernel: (match - mismatch) must be a byte
  const uint32_t tables = !options().no_tables && a.match >= a.mismatch && a.match - a.mismatch <= 255;
  {
    Timed tm(stream, 1);
    if (COOP) {
      const size_t cells = (size_t)a.pairs_in_pass * a.nstrips;
      SA_HIP_TRY(hipMemsetAsync(co.ticket, 0, 4, stream));
      // boundary entries carry the tag of the strip that wrote them; 0 = not written in this launch
      SA_HIP_TRY(hipMemsetAsync(a.brow, 0, (size_t)a.pairs_in_pass * a.brow_unit_stride * 4, stream));
      SA_HIP_TRY(hipMemsetAsync(co.cand, 0xFF, cells * 12, stream));
      hipLaunchKernelGGL((fill_wave_i32<MODE, C, true>), dim3((unsigned)cells), dim3(64), 0, stream, a, co, tables);
      if (MODE == SA_LOCAL)
        hipLaunchKernelGGL(wave_local_reduce, dim3((a.pairs_in_pass + 63) / 64), dim3(64), 0, stream, a, co);
    } else {
      hipLaunchKernelGGL((fill_wave_i32<MODE, C, false>), dim3(a.pairs_in_pass), dim3(64), 0, stream, a, co, tables);
    }
  }
  SA_HIP_TRY(hipGetLastError());
  if (need_traceback) {
    {
      Timed tm(stream, 2);
      hipLaunchKernelGGL((traceback_wave<MODE, WaveFmtI32<C>>), dim3(a.pairs_in_pass), dim3(64), 0, stream, a);
    }
    SA_HIP_TRY(hipGetLastError());
  }
  return SA_OK;
}

// The traceback of the cooperative kernels' long pairs: by one wavefront per segment of target rows
// (align_wave_segments.h) when the pairs of the pass are few, the state of the segments in the boundary area,
// which the fill has finished with; else one wavefront per pair.
template <int MODE, class Fmt>
int launch_long_traceback(const AlignArgs& a, const WaveCoop& co, size_t brow_words, hipStream_t stream) {
  const Options& opt = options();
  // rows per segment: 1024, for targets below 8192 rows 512; from four segments up
  uint32_t rows = opt.seg_rows > 0 ? (uint32_t)round_up((size_t)opt.seg_rows, 64) : a.max_t >= 8192 ? 1024 : 512;
  rows = std::max<uint32_t>(rows, (uint32_t)round_up(((size_t)a.max_t + kMaxSegments - 1) / kMaxSegments, 64));
  const uint32_t nseg = (a.max_t + rows - 1) / rows;
  const size_t state_words = seg_state_words(a.pairs_in_pass, nseg) + 1;
  // Worth it when the path spans four segments or more (a Semiglobal path has about as many rows as the
  // query has columns, however long the target) and the pairs of the pass are too few to fill the GPU with
  // one wavefront each (400 pairs of 500 x 5000: 0.05 ms that way, 0.19 ms in segments).
  const uint32_t path_rows = MODE == SA_SEMIGLOBAL ? std::min(a.max_q, a.max_t) : a.max_t;
  const bool segments = !opt.no_segments && a.cigar_runs && nseg >= 4 && path_rows >= 4 * rows &&
                        (a.pairs_in_pass <= 256 || opt.seg_rows > 0) && a.cigar_stride / nseg >= 64 &&
                        a.cigar_stride / nseg <= 0xFFFFFFFFull && state_words <= brow_words &&
                        (size_t)a.pairs_in_pass * nseg < ((size_t)1 << 31);
  Timed tm(stream, 2);
  if (segments) {
    SegArgs sg{};
    sg.rows = rows;
    sg.nseg = nseg;
    sg.cap = (uint32_t)(a.cigar_stride / nseg);
    if (opt.seg_cap > 0 && (uint64_t)opt.seg_cap < sg.cap) sg.cap = (uint32_t)opt.seg_cap;
    // generous: a segment along the straight line to the end cell has rows + rows * Q / T cells, at worst one
    // or two per iteration; a guess that runs into the bound costs the segment that time once, not the pair
    sg.spec_iters = (uint32_t)std::min<uint64_t>(2 * ((uint64_t)rows + (uint64_t)rows * a.max_q / a.max_t) + 64, 1u << 30);
    sg.pub = reinterpret_cast<uint64_t*>(a.brow);
    sg.seg_nruns = reinterpret_cast<uint32_t*>(sg.pub + (size_t)a.pairs_in_pass * nseg);
    sg.redo = sg.seg_nruns + (size_t)a.pairs_in_pass * nseg;
    sg.ticket = sg.redo + a.pairs_in_pass;
    sg.abort_flag = co.abort_flag;
    sg.spin_limit = co.spin_limit;
    sg.spec_walks = (uint32_t)std::max(0L, std::min(8L, (long)opt.seg_spec));
    hipLaunchKernelGGL(clear_words, dim3((unsigned)((state_words + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t*>(a.brow), (uint32_t)state_words);
    hipLaunchKernelGGL((traceback_wave_segments<MODE, Fmt>), dim3(a.pairs_in_pass * nseg), dim3(64), 0, stream, a, sg);
    hipLaunchKernelGGL((stitch_segments<MODE>), dim3(a.pairs_in_pass), dim3(1024), 0, stream, a, sg);
    AlignArgs ar = a;  // pairs whose segments ran out of room: one wavefront each, from the start
    ar.tb_only = sg.redo;
    hipLaunchKernelGGL((traceback_wave<MODE, Fmt>), dim3(a.pairs_in_pass), dim3(64), 0, stream, ar);
  } else {
    hipLaunchKernelGGL((traceback_wave<MODE, Fmt>), dim3(a.pairs_in_pass), dim3(64), 0, stream, a);
  }
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

template <int MODE, int C>
int launch_wave_coop2(const AlignArgs& a, const WaveCoop& co, hipStream_t stream, bool need_traceback) {
  if constexpr (MODE == SA_LOCAL) {
    return fail(SA_ERR_ARGUMENT, "cooperative Global / Semiglobal kernel chosen for Local");
  } else {
    {
      Timed tm(stream, 1);
      const size_t waves = (size_t)a.pairs_in_pass * a.nstrips;
      SA_HIP_TRY(hipMemsetAsync(co.ticket, 0, 4, stream));
      // boundary entries carry the tag of the strip that wrote them; 0 = not written in this launch
      SA_HIP_TRY(hipMemsetAsync(a.brow, 0, (size_t)a.pairs_in_pass * a.brow_unit_stride * 4, stream));
      // score tables inside the kernel: (match - mismatch) must be a byte
      const uint32_t tables = !options().no_tables && a.match >= a.mismatch && a.match - a.mismatch <= 255;
      hipLaunchKernelGGL((fill_wave_coop2<MODE, C>), dim3((unsigned)waves), dim3(64), 0, stream, a, co, tables);
    }
    SA_HIP_TRY(hipGetLastError());
    if (need_traceback)
      return launch_long_traceback<MODE, WaveFmtCoop<C>>(a, co, (size_t)a.pairs_in_pass * a.brow_unit_stride, stream);
    return SA_OK;
  }
}

template <int MODE, int C, bool COOP>
int launch_wave_pk16(const Plan& pl, const AlignArgs& a, const WaveCoop& co, hipStream_t stream, bool need_traceback,
                     int overlap_device) {
  if constexpr (COOP && MODE == SA_LOCAL) {
    return fail(SA_ERR_ARGUMENT, "packed cooperative kernel chosen for Local");
  } else {
    const uint32_t units = (a.pairs_in_pass + 1) / 2;
    // score tables inside the kernel (a wavefront whose sequences hold at most four distinct bytes): the
    // entries (mismatch - gap) and (match - mismatch) << SH must be non-negative bytes
    const int sh = MODE == SA_LOCAL ? 4 : 0;
    const uint32_t tables = !options().no_tables && a.mismatch - a.gap >= 0 &&
                            ((long)(a.mismatch - a.gap) << sh) <= 255 && ((long)(a.match - a.mismatch) << sh) <= 255;
    auto fill = [&](const AlignArgs& x, uint32_t blocks) {
      Timed tm(stream, 1);
      hipLaunchKernelGGL((fill_wave_pk16<MODE, C, COOP>), dim3(blocks), dim3(64), 0, stream, x, co, pl.bias, tables);
    };
    // One wavefront per two pairs: the traceback of the pairs of the fill's whole rounds runs on a second
    // stream under the last, partly filled round.
    uint32_t head = 0;
    if (!COOP && need_traceback && overlap_device >= 0) {
      static int per_cu = 0;
      if (!per_cu) {
        int v = 0;
        per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fill_wave_pk16<MODE, C, COOP>, 64, 0) == hipSuccess && v > 0) ? v : -1;
      }
      if (per_cu > 0) head = whole_rounds(units, (uint32_t)per_cu * device_cus(overlap_device));
    }
    if (head) {
      SideLease side(overlap_device);
      if (side.ok) {
        const AlignArgs xa = sub_pass(pl, a, 0, head * 2), xb = sub_pass(pl, a, head, a.pairs_in_pass - head * 2);
        fill(xa, head);
        SA_HIP_TRY(hipEventRecord(side.s.forked, stream));
        SA_HIP_TRY(hipStreamWaitEvent(side.s.stream, side.s.forked, 0));
        {
          Timed tm(side.s.stream, 2);
          hipLaunchKernelGGL((traceback_wave<MODE, WaveFmtPk16<C>>), dim3(xa.pairs_in_pass), dim3(64), 0, side.s.stream, xa);
        }
        SA_HIP_TRY(hipEventRecord(side.s.joined, side.s.stream));
        fill(xb, units - head);
        {
          Timed tm(stream, 2);
          hipLaunchKernelGGL((traceback_wave<MODE, WaveFmtPk16<C>>), dim3(xb.pairs_in_pass), dim3(64), 0, stream, xb);
        }
        SA_HIP_TRY(hipStreamWaitEvent(stream, side.s.joined, 0));
        SA_HIP_TRY(hipGetLastError());
        return SA_OK;
      }
    }
    if (COOP) {
      SA_HIP_TRY(hipMemsetAsync(co.ticket, 0, 4, stream));
      // boundary entries carry the tag of the strip that wrote them; 0 = not written in this launch
      SA_HIP_TRY(hipMemsetAsync(a.brow, 0, (size_t)units * a.brow_unit_stride * 4, stream));
      // Strips in segments of target rows, a wavefront per segment: the strips of the launch then advance
      // together and end together, instead of the last ones running on a GPU that empties.
      AlignArgs x = a;
      const long opt_rows = options().coop_seg_rows;
      const uint32_t steps = a.max_t + 63;
      const uint64_t strips = (uint64_t)units * a.nstrips;
      uint32_t rows = opt_rows >= 0 ? (uint32_t)round_up((size_t)opt_rows, 64) : (strips > kWavePkResident ? kWavePkSegRows : 0u);
      if (rows >= steps) rows = 0;
      x.seg_rows = rows;
      x.nseg = rows ? (uint32_t)((steps + (uint64_t)(a.nstrips - 1) * kWaveSegShift + rows - 1) / rows) : 1;
      if (strips * x.nseg > 0x7FFFFFFFull) return fail(SA_ERR_ARGUMENT, "too many segments in a cooperative pass");
      fill(x, (uint32_t)(strips * x.nseg));
    } else {
      fill(a, units);
    }
    SA_HIP_TRY(hipGetLastError());
    if (need_traceback) {
      if constexpr (COOP) {
        return launch_long_traceback<MODE, WaveFmtPk16<C>>(a, co, (size_t)units * a.brow_unit_stride, stream);
      } else {
        Timed tm(stream, 2);
        hipLaunchKernelGGL((traceback_wave<MODE, WaveFmtPk16<C>>), dim3(a.pairs_in_pass), dim3(64), 0, stream, a);
        SA_HIP_TRY(hipGetLastError());
      }
    }
    return SA_OK;
  }
}

template <int MODE>
int launch_pass(const Plan& pl, const AlignArgs& a, const WaveCoop& co, uint32_t* qpk, uint32_t* tpk,
                AlphabetInfo* alpha, hipStream_t stream, bool need_tb, int overlap_device) {
  if (pl.family == Family::LaneI32) return launch_lane_i32<MODE>(a, stream, need_tb);
  if (pl.family == Family::WaveI32 && pl.wave_pk && pl.coop)
    return pl.wave_c == 8    ? launch_wave_pk16<MODE, 8, true>(pl, a, co, stream, need_tb, -1)
           : pl.wave_c == 10 ? launch_wave_pk16<MODE, 10, true>(pl, a, co, stream, need_tb, -1)
           : pl.wave_c == 12 ? launch_wave_pk16<MODE, 12, true>(pl, a, co, stream, need_tb, -1)
                             : launch_wave_pk16<MODE, 16, true>(pl, a, co, stream, need_tb, -1);
  if (pl.family == Family::WaveI32 && pl.wave_pk) return launch_wave_pk16<MODE, 16, false>(pl, a, co, stream, need_tb, overlap_device);
  if (pl.family == Family::WaveI32 && pl.coop && MODE != SA_LOCAL)
    return pl.wave_c == 2   ? launch_wave_coop2<MODE, 2>(a, co, stream, need_tb)
           : pl.wave_c == 4 ? launch_wave_coop2<MODE, 4>(a, co, stream, need_tb)
                            : launch_wave_coop2<MODE, 8>(a, co, stream, need_tb);
  if (pl.family == Family::WaveI32)
    return pl.coop ? (pl.wave_c == 4 ? launch_wave_i32<MODE, 4, true>(a, co, stream, need_tb)
                                     : launch_wave_i32<MODE, 8, true>(a, co, stream, need_tb))
           : pl.wave_c == 4 ? launch_wave_i32<MODE, 4, false>(a, co, stream, need_tb)
           : pl.wave_c == 8 ? launch_wave_i32<MODE, 8, false>(a, co, stream, need_tb)
                            : launch_wave_i32<MODE, 16, false>(a, co, stream, need_tb);
  switch (pl.K) {
    case 32: return launch_lane_pk16<MODE, 32>(pl, a, qpk, tpk, alpha, stream, need_tb, overlap_device);
    case 64: return launch_lane_pk16<MODE, 64>(pl, a, qpk, tpk, alpha, stream, need_tb, overlap_device);
    default: return launch_lane_pk16<MODE, 48>(pl, a, qpk, tpk, alpha, stream, need_tb, overlap_device);
  }
}

}  // namespace
}  // namespace sa

using namespace sa;

extern "C" {

const char* sa_last_error(void) { return sa::last_error().c_str(); }

int sa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sa_set_option(const char* name, long value) {
  if (!name) return fail(SA_ERR_ARGUMENT, "option name is NULL");
  Options& o = options();
  struct {
    const char* name;
    std::atomic<long>* slot;
  } table[] = {{"force_i32", &o.force_i32},   {"force_family", &o.force_family}, {"force_coop", &o.force_coop},
               {"no_tables", &o.no_tables},   {"coop_c", &o.coop_c},             {"coop_waves", &o.coop_waves},
               {"coop_pk", &o.coop_pk}, {"coop_seg_rows", &o.coop_seg_rows}, {"coop_async", &o.coop_async},       {"no_overlap", &o.no_overlap},
               {"pk16_k", &o.pk16_k},
               {"min_chunk", &o.min_chunk},   {"fake_devices", &o.fake_devices},
               {"no_buckets", &o.no_buckets}, {"no_segments", &o.no_segments}, {"seg_rows", &o.seg_rows},
               {"seg_cap", &o.seg_cap},     {"seg_spec", &o.seg_spec},
               {"spin_limit", &o.spin_limit}};
  for (auto& e : table)
    if (!strcmp(name, e.name)) {
      *e.slot = value;
      return SA_OK;
    }
  return fail(SA_ERR_ARGUMENT, "unknown option '%s'", name);
}

const char* sa_device_arch(int device) {
  static thread_local char name[256];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
  snprintf(name, sizeof name, "%s", prop.gcnArchName);
  return name;
}

int sa_align_profile(int enable) {
  g_profile = enable != 0;
  return SA_OK;
}

int sa_align_profile_read(double ms[3], uint64_t launches[3]) {
  for (int i = 0; i < 3; ++i) {
    ms[i] = 0;
    launches[i] = 0;
  }
  std::vector<Span> spans;
  {
    std::lock_guard<std::mutex> g(g_spans_mutex);
    spans.swap(g_spans);
  }
  for (Span& sp : spans) {
    float t = 0;
    SA_HIP_TRY(hipEventSynchronize(sp.b));
    SA_HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
    ms[sp.cls] += t;
    launches[sp.cls] += 1;
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  return SA_OK;
}

int sa_cigar_stats_device(int device, void* stream, uint32_t n, const uint32_t* d_cigar_runs,
                          const uint32_t* d_cigar_nruns, uint64_t cigar_stride, uint32_t* d_block_len,
                          uint32_t* d_op_len) {
  if (n == 0) return SA_OK;
  if (!d_cigar_runs || !d_cigar_nruns) return fail(SA_ERR_ARGUMENT, "NULL run arrays");
  SA_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(cigar_stats_kernel, dim3((unsigned)(((uint64_t)n * 64 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, n, d_cigar_runs, d_cigar_nruns, cigar_stride, d_block_len, d_op_len);
  SA_HIP_TRY(hipGetLastError());
  return SA_OK;
}

size_t sa_align_workspace_bytes(const sa_align_batch* b, size_t* min_bytes) {
  if (!b) return 0;
  Plan pl = make_plan(*b);
  const size_t blocks = ((size_t)b->n + pl.pairs_per_block - 1) / pl.pairs_per_block;
  if (min_bytes) *min_bytes = pl.bytes_per_block + kFixedWorkspace;
  return pl.bytes_per_block * (blocks ? blocks : 1) + kFixedWorkspace;
}

const char* sa_align_kernel_name(const sa_align_batch* b) {
  if (!b) return "";
  return make_plan(*b).name;
}

}  // extern "C"

namespace sa {
namespace {

// The scratch memory of sa_align_batch_device_static: one array per GPU, part of the library's code object,
// so it exists from the moment the library is loaded on a GPU until the process ends and no call ever
// allocates it.  In host code the array's name is only a handle for hipGetSymbolAddress; g_scratch_ptr[device]
// holds the array's address on that GPU.
constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on `device`, which must be the current device; looked up once per device.
int static_scratch(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return fail(SA_ERR_ARGUMENT, "device %d out of range", device);
  std::lock_guard<std::mutex> g(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    SA_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(SA_ERR_HIP, "hipGetSymbolAddress gave no address for the static scratch array");
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return SA_OK;
}

int align_batch_device(int device, void* stream_, const sa_align_batch* b, void* d_workspace, size_t workspace_bytes,
                       bool static_scratch_call);

// Blocks of pairs one pass over a workspace of `workspace_bytes` (>= the plan's minimum) holds.
size_t blocks_in_a_pass(const Plan& pl, const sa_align_batch& b, size_t workspace_bytes, int device) {
  const size_t total_blocks = ((size_t)b.n + pl.pairs_per_block - 1) / pl.pairs_per_block;
  size_t blocks_per_pass = (workspace_bytes - kFixedWorkspace) / pl.bytes_per_block;
  if (blocks_per_pass > total_blocks) blocks_per_pass = total_blocks;
  // a pass stays within the grid limits of its launches (65 535 in y for the packing kernel's units, 2^32
  // threads in all for the one-wavefront-per-pair kernels)
  const size_t max_blocks = std::max<size_t>(1, ((size_t)1 << 22) / pl.pairs_per_block);
  if (blocks_per_pass > max_blocks) blocks_per_pass = max_blocks;
  if (pl.family == Family::WaveI32 && !pl.coop && blocks_per_pass < total_blocks) {
    // One wavefront per unit, all of about the same duration: a pass takes as long as the SIMD with the most
    // wavefronts, so a pass that is cut by the workspace holds a whole number of wavefronts per SIMD
    // (3200 wavefronts on 1024 SIMDs take four rounds, 3072 take three).
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    const size_t simds = (size_t)cus * 4;
    if (blocks_per_pass >= simds) blocks_per_pass = blocks_per_pass / simds * simds;
  }
  return blocks_per_pass;
}

// The plan of a batch over a workspace of this size.  Long pairs need hundreds of megabytes of direction words
// each: when the workspace holds fewer than 2048 wavefronts' worth of them at a time, the kernel with one
// wavefront per two pairs would leave most SIMDs idle in every pass, and the batch takes the packed cooperative
// kernel (a wavefront per strip) whatever its size.
Plan plan_for_workspace(const sa_align_batch& b, bool allow_coop, size_t workspace_bytes, int device) {
  Plan pl = make_plan(b, allow_coop);
  if (allow_coop && pl.family == Family::WaveI32 && pl.wave_pk && !pl.coop && b.type != SA_LOCAL &&
      workspace_bytes >= pl.bytes_per_block + kFixedWorkspace) {
    const size_t total_blocks = ((size_t)b.n + pl.pairs_per_block - 1) / pl.pairs_per_block;
    const size_t per_pass = blocks_in_a_pass(pl, b, workspace_bytes, device);
    if (per_pass < total_blocks && per_pass * pl.units_per_block < 2048) {
      const Plan coop = make_plan(b, allow_coop, true);
      if (coop.coop && coop.wave_pk && workspace_bytes >= coop.bytes_per_block + kFixedWorkspace) pl = coop;
    }
  }
  return pl;
}

// Pairs of a several-pass batch of the one-wavefront-per-pair kernels that are planned again as a batch of their
// own (0: none): the last pass's pairs when they are too few to fill the GPU.
size_t leftover_pairs(const Plan& pl, const sa_align_batch& b, size_t pairs_per_pass, bool static_scratch_call) {
  if (pl.family != Family::WaveI32 || pl.coop || static_scratch_call || b.n <= pairs_per_pass) return 0;
  const size_t rest = b.n % pairs_per_pass;
  return rest <= 2048 ? rest : 0;
}

// Whether align_batch_device would run cooperative kernels (and therefore synchronise the stream) for this batch
// over a workspace of this size: directly, or for the head or the left-over pairs of a several-pass batch.
bool batch_synchronises(const sa_align_batch& b, size_t workspace_bytes, int device) {
  if (b.n == 0 || b.type < 0 || b.type > 2 || options().coop_async) return false;
  const Plan pl = plan_for_workspace(b, true, workspace_bytes, device);
  if (pl.coop) return true;
  if (workspace_bytes < pl.bytes_per_block + kFixedWorkspace) return false;  // the call fails before any launch
  const size_t pairs_per_pass = blocks_in_a_pass(pl, b, workspace_bytes, device) * pl.pairs_per_block;
  const size_t rest_n = leftover_pairs(pl, b, pairs_per_pass, false);
  if (!rest_n) return false;
  sa_align_batch head = b, rest = b;
  head.n = (uint32_t)(b.n - rest_n);
  rest.n = (uint32_t)rest_n;
  return batch_synchronises(head, workspace_bytes, device) || batch_synchronises(rest, workspace_bytes, device);
}

}  // namespace
}  // namespace sa

extern "C" {

int sa_align_batch_device(int device, void* stream_, const sa_align_batch* b, void* d_workspace,
                          size_t workspace_bytes) {
  return align_batch_device(device, stream_, b, d_workspace, workspace_bytes, false);
}

size_t sa_align_static_scratch_bytes(void) { return kStaticScratchBytes; }

int sa_align_may_synchronize(int device, const sa_align_batch* b, size_t workspace_bytes) {
  if (!b) return 0;
  return batch_synchronises(*b, workspace_bytes, device) ? 1 : 0;
}

int sa_align_static_init(int device) {
  SA_HIP_TRY(hipSetDevice(device));
  void* p = nullptr;
  return static_scratch(device, &p);
}

int sa_align_batch_device_static(int device, void* stream, const sa_align_batch* b) {
  return align_batch_device(device, stream, b, nullptr, 0, true);
}

}  // extern "C"

namespace sa {
namespace {

int align_batch_device(int device, void* stream_, const sa_align_batch* b, void* d_workspace, size_t workspace_bytes,
                       bool static_scratch_call) {
  if (!b) return fail(SA_ERR_ARGUMENT, "batch is NULL");
  if (b->type < 0 || b->type > 2) return fail(SA_ERR_ARGUMENT, "unknown alignment type %d", b->type);
  if (b->n == 0) return SA_OK;
  if (!b->d_q || !b->d_q_off || !b->d_t || !b->d_t_off || !b->d_score)
    return fail(SA_ERR_ARGUMENT, "NULL sequence/offset/score pointer");
  if (b->d_cigar_runs && b->cigar_stride < (uint64_t)b->max_qlen + b->max_tlen)
    return fail(SA_ERR_ARGUMENT, "cigar_stride %llu < max_qlen + max_tlen = %llu",
                (unsigned long long)b->cigar_stride,
                (unsigned long long)b->max_qlen + b->max_tlen);
  if (b->d_cigar_runs && !b->d_cigar_nruns)
    return fail(SA_ERR_ARGUMENT, "d_cigar_runs given without d_cigar_nruns");
  SA_HIP_TRY(hipSetDevice(device));
  hipStream_t stream = (hipStream_t)stream_;

  // the static-scratch call only enqueues work: never the cooperative kernels, which synchronise the stream
  const Plan pl = plan_for_workspace(*b, !static_scratch_call, workspace_bytes, device);
  if (static_scratch_call) {
    if (pl.bytes_per_block + kFixedWorkspace > kStaticScratchBytes)
      return fail(SA_ERR_WORKSPACE,
                  "sa_align_batch_device_static: one pass of this batch (%s, query <= %u, target <= %u) needs %zu "
                  "bytes of scratch, the static array holds %zu; use sa_align_batch_device with a workspace",
                  pl.name, b->max_qlen, b->max_tlen, pl.bytes_per_block + kFixedWorkspace, kStaticScratchBytes);
    if (int rc = static_scratch(device, &d_workspace)) return rc;
    workspace_bytes = kStaticScratchBytes;
  }
  if (!d_workspace || workspace_bytes < pl.bytes_per_block + kFixedWorkspace)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the minimum %zu", workspace_bytes,
                pl.bytes_per_block + kFixedWorkspace);
  const size_t blocks_per_pass = blocks_in_a_pass(pl, *b, workspace_bytes, device);
  const size_t units_per_pass = blocks_per_pass * pl.units_per_block;
  size_t pairs_per_pass = blocks_per_pass * pl.pairs_per_block;

  // One wavefront per pair in several passes: the pairs left over for the last pass are a batch of their own,
  // planned anew - a few hundred of them are too few wavefronts for these kernels (424 pairs of 10 kbp x 11 kbp
  // take as long as 2048) and run on the cooperative kernel instead.
  if (const size_t rest_n = leftover_pairs(pl, *b, pairs_per_pass, static_scratch_call)) {
    const size_t head_n = b->n - rest_n;
    sa_align_batch head = *b, rest = *b;
    head.n = (uint32_t)head_n;
    rest.n = (uint32_t)rest_n;
    rest.d_q_off += head_n;
    rest.d_t_off += head_n;
    rest.d_score += head_n;
    if (rest.d_target_begin) rest.d_target_begin += head_n;
    if (rest.d_cigar_runs) rest.d_cigar_runs += head_n * b->cigar_stride;
    if (rest.d_cigar_nruns) rest.d_cigar_nruns += head_n;
    if (int rc = align_batch_device(device, stream_, &head, d_workspace, workspace_bytes, false)) return rc;
    return align_batch_device(device, stream_, &rest, d_workspace, workspace_bytes, false);
  }

  // carve the workspace
  static_assert(sizeof(AlphabetInfo) <= kFixedWorkspace, "fixed workspace too small");
  AlphabetInfo* alpha = nullptr;
  if (pl.family == Family::LanePk16 && pl.tables) {
    // device-side decision, uniform for the call: at most four distinct symbols -> score tables
    alpha = (AlphabetInfo*)d_workspace;
    static_assert(sizeof(AlphabetInfo) % 4 == 0 && sizeof(AlphabetInfo) <= 1024, "AlphabetInfo: one block of words");
    Timed tm(stream, 0);
    hipLaunchKernelGGL(clear_words, dim3(1), dim3(256), 0, stream, (uint32_t*)alpha, (uint32_t)(sizeof(AlphabetInfo) / 4));
    hipLaunchKernelGGL(alphabet_scan, dim3(kScanBlocks), dim3(256), 0, stream, b->d_q, b->d_q_off, b->d_t,
                       b->d_t_off, b->n, alpha);
    hipLaunchKernelGGL(alphabet_table, dim3(1), dim3(256), 0, stream, alpha);
    SA_HIP_TRY(hipGetLastError());
  }
  char* w = (char*)d_workspace + kFixedWorkspace;
  AlignArgs a{};
  a.q = b->d_q;
  a.q_off = b->d_q_off;
  a.t = b->d_t;
  a.t_off = b->d_t_off;
  a.max_q = b->max_qlen ? b->max_qlen : 1;
  a.max_t = b->max_tlen ? b->max_tlen : 1;
  a.match = b->match;
  a.mismatch = b->mismatch;
  a.gap = b->gap;
  a.dirs = (uint32_t*)w;
  a.dirs_unit_stride = pl.dirs_unit_stride;
  w += round_up(units_per_pass * pl.dirs_unit_stride * 4, 256);
  a.brow = (int32_t*)w;
  a.brow_unit_stride = pl.brow_unit_stride;
  w += round_up(units_per_pass * pl.brow_unit_stride * 4, 256);
  uint32_t* qpk = (uint32_t*)w;
  w += round_up(units_per_pass * pl.qpk_unit_words * 4, 256);
  uint32_t* tpk = (uint32_t*)w;
  w += round_up(units_per_pass * pl.tpk_unit_words * 4, 256);
  a.end_score = (int32_t*)w;
  w += round_up(pairs_per_pass * 4, 256);
  a.end_i = (uint32_t*)w;
  w += round_up(pairs_per_pass * 4, 256);
  a.end_r = (uint32_t*)w;
  w += round_up(pairs_per_pass * 4, 256);
  WaveCoop co{};
  if (pl.coop) {
    co.ticket = (uint32_t*)d_workspace;  // head of the workspace (the packed kernels keep their table there)
    co.abort_flag = (int*)d_workspace + 1;
    co.cand = (int32_t*)w;
    co.spin_limit = (uint32_t)std::max(1L, std::min((long)kSpinLimit, (long)options().spin_limit));
    a.gave_up = co.abort_flag;
    SA_HIP_TRY(hipMemsetAsync(co.abort_flag, 0, 4, stream));
    // Wavefronts of one pass.  They need not all be resident: strips take their numbers from a ticket counter in
    // start order, so a strip only ever waits for strips that already run.
    // (The packed kernel runs the strips of a large pass in segments, all strips advancing together: its passes
    // are as large as the workspace allows unless the option asks for less than its default.)
    size_t coop_waves = (size_t)std::max(64L, std::min(8192L, (long)options().coop_waves));
    if (pl.wave_pk && coop_waves == 8192) coop_waves = kWavePkPassStrips;
    const size_t cap = std::max<size_t>(1, coop_waves / pl.nstrips) * pl.pairs_per_unit;
    if (pairs_per_pass > cap) pairs_per_pass = cap;
  }
  a.nstrips = pl.nstrips;
  a.score = b->d_score;
  a.target_begin = b->d_target_begin;
  a.cigar_runs = b->d_cigar_runs;
  a.cigar_nruns = b->d_cigar_nruns;
  a.cigar_stride = b->cigar_stride;

  const bool need_tb = b->d_target_begin || b->d_cigar_runs || b->d_cigar_nruns;
  // (the static-scratch call creates nothing, a second stream included; the option "no_overlap" is a test hook)
  const int overlap_device = (static_scratch_call || options().no_overlap) ? -1 : device;
  for (size_t base = 0; base < b->n; base += pairs_per_pass) {
    a.pair_base = (uint32_t)base;
    a.pairs_in_pass = (uint32_t)((b->n - base) < pairs_per_pass ? (b->n - base) : pairs_per_pass);
    int rc;
    if (b->type == SA_GLOBAL)
      rc = launch_pass<SA_GLOBAL>(pl, a, co, qpk, tpk, alpha, stream, need_tb, overlap_device);
    else if (b->type == SA_LOCAL)
      rc = launch_pass<SA_LOCAL>(pl, a, co, qpk, tpk, alpha, stream, need_tb, overlap_device);
    else
      rc = launch_pass<SA_SEMIGLOBAL>(pl, a, co, qpk, tpk, alpha, stream, need_tb, overlap_device);
    if (rc) return rc;
  }
  if (pl.coop && options().coop_async) {
    hipLaunchKernelGGL(mark_scores_if_gave_up, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, stream, co.abort_flag,
                       b->d_score, (uint32_t)b->n);
    SA_HIP_TRY(hipGetLastError());
  } else if (pl.coop) {  // the cooperative kernels report a wait that ran out through a flag
    int aborted = 0;
    SA_HIP_TRY(hipMemcpyAsync(&aborted, co.abort_flag, 4, hipMemcpyDeviceToHost, stream));
    SA_HIP_TRY(hipStreamSynchronize(stream));
    if (aborted) return fail(SA_ERR_HIP, "cooperative kernels: a strip or a traceback segment waited too long for its neighbour");
  }
  return SA_OK;
}

}  // namespace
}  // namespace sa
