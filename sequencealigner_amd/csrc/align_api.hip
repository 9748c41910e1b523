// Device-resident batched Align(): workspace planning, kernel choice and launches.
#include "align_common.h"
#include "align_lane_i32.h"

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

namespace {

constexpr int kLaneI32K = 32;      // target rows per register strip of the 32-bit lane kernel
constexpr uint32_t kBlock = 256;   // threads per workgroup of the lane kernels

enum class Family { LaneI32 };

struct Plan {
  Family family;
  uint32_t nstrips;
  uint64_t dirs_unit_stride;  // dwords per unit (wave)
  uint64_t brow_unit_stride;  // ints per unit
  uint32_t pairs_per_unit;    // pairs sharing one unit of workspace
  size_t bytes_per_block;     // workspace bytes per block of kBlock threads
  uint32_t pairs_per_block;
};

Plan make_plan(uint32_t /*n*/, uint32_t max_q, uint32_t max_t, int /*type*/, int /*match*/,
               int /*mismatch*/, int /*gap*/) {
  Plan pl{};
  pl.family = Family::LaneI32;
  pl.nstrips = (max_t + kLaneI32K - 1) / kLaneI32K;
  if (pl.nstrips == 0) pl.nstrips = 1;
  pl.dirs_unit_stride = (uint64_t)pl.nstrips * (max_q ? max_q : 1) * (kLaneI32K / 16) * 64;
  pl.brow_unit_stride = (uint64_t)(max_q + 1) * 64;
  pl.pairs_per_unit = 64;
  pl.pairs_per_block = kBlock;
  const size_t units = kBlock / 64;
  pl.bytes_per_block = round_up(units * pl.dirs_unit_stride * 4, 256) +
                       round_up(units * pl.brow_unit_stride * 4, 256) +
                       round_up((size_t)pl.pairs_per_block * 4, 256) * 3;
  return pl;
}

// ---- optional kernel timing (sa_align_profile) ---------------------------------------------------
struct Span {
  hipEvent_t a, b;
  int cls;
};
bool g_profile = false;
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
    g_spans.push_back({a, b, cls});
  }
};

template <int MODE>
int launch_pass(const Plan& pl, const AlignArgs& a, hipStream_t stream, bool need_traceback) {
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
  (void)pl;
  return SA_OK;
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
  for (Span& sp : g_spans) {
    float t = 0;
    SA_HIP_TRY(hipEventSynchronize(sp.b));
    SA_HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
    ms[sp.cls] += t;
    launches[sp.cls] += 1;
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  g_spans.clear();
  return SA_OK;
}

size_t sa_align_workspace_bytes(uint32_t n, uint32_t max_qlen, uint32_t max_tlen, int type,
                                size_t* min_bytes) {
  Plan pl = make_plan(n, max_qlen, max_tlen, type, 1, -1, -1);
  const size_t blocks = ((size_t)n + pl.pairs_per_block - 1) / pl.pairs_per_block;
  if (min_bytes) *min_bytes = pl.bytes_per_block;
  return pl.bytes_per_block * (blocks ? blocks : 1);
}

const char* sa_align_kernel_name(uint32_t, uint32_t, uint32_t, int, int, int, int) {
  return "lane_i32";
}

int sa_align_batch_device(int device, void* stream_, const sa_align_batch* b, void* d_workspace,
                          size_t workspace_bytes) {
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

  const Plan pl = make_plan(b->n, b->max_qlen, b->max_tlen, b->type, b->match, b->mismatch, b->gap);
  if (!d_workspace || workspace_bytes < pl.bytes_per_block)
    return fail(SA_ERR_WORKSPACE, "workspace of %zu bytes is below the minimum %zu", workspace_bytes,
                pl.bytes_per_block);
  const size_t total_blocks = ((size_t)b->n + pl.pairs_per_block - 1) / pl.pairs_per_block;
  size_t blocks_per_pass = workspace_bytes / pl.bytes_per_block;
  if (blocks_per_pass > total_blocks) blocks_per_pass = total_blocks;
  const size_t units_per_pass = blocks_per_pass * (kBlock / 64);
  const size_t pairs_per_pass = blocks_per_pass * pl.pairs_per_block;

  // carve the workspace
  char* w = (char*)d_workspace;
  AlignArgs a{};
  a.q = b->d_q;
  a.q_off = b->d_q_off;
  a.t = b->d_t;
  a.t_off = b->d_t_off;
  a.max_q = b->max_qlen;
  a.max_t = b->max_tlen;
  a.match = b->match;
  a.mismatch = b->mismatch;
  a.gap = b->gap;
  a.dirs = (uint32_t*)w;
  a.dirs_unit_stride = pl.dirs_unit_stride;
  w += round_up(units_per_pass * pl.dirs_unit_stride * 4, 256);
  a.brow = (int32_t*)w;
  a.brow_unit_stride = pl.brow_unit_stride;
  w += round_up(units_per_pass * pl.brow_unit_stride * 4, 256);
  a.end_score = (int32_t*)w;
  w += round_up(pairs_per_pass * 4, 256);
  a.end_i = (uint32_t*)w;
  w += round_up(pairs_per_pass * 4, 256);
  a.end_r = (uint32_t*)w;
  a.nstrips = pl.nstrips;
  a.score = b->d_score;
  a.target_begin = b->d_target_begin;
  a.cigar_runs = b->d_cigar_runs;
  a.cigar_nruns = b->d_cigar_nruns;
  a.cigar_stride = b->cigar_stride;

  const bool need_tb = b->d_target_begin || b->d_cigar_runs || b->d_cigar_nruns;
  for (size_t base = 0; base < b->n; base += pairs_per_pass) {
    a.pair_base = (uint32_t)base;
    a.pairs_in_pass = (uint32_t)((b->n - base) < pairs_per_pass ? (b->n - base) : pairs_per_pass);
    int rc;
    if (b->type == SA_GLOBAL)
      rc = launch_pass<SA_GLOBAL>(pl, a, stream, need_tb);
    else if (b->type == SA_LOCAL)
      rc = launch_pass<SA_LOCAL>(pl, a, stream, need_tb);
    else
      rc = launch_pass<SA_SEMIGLOBAL>(pl, a, stream, need_tb);
    if (rc) return rc;
  }
  return SA_OK;
}

}  // extern "C"
