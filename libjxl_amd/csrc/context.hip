// context.hip -- the C ABI of include/jxl_hip.h: context, frame set-up, input
// hand-off, the two decode phases, halo regions, profiling.  Host code only;
// kernels live in kernels_*.hip.  No CPU fallback: every entry point that needs
// a device fails with JXLHIP_ERR_NO_DEVICE / JXLHIP_ERR_HIP when there is none.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/jxl_hip_codestream.h"
#include "../../include/jxl_hip_entropy.h"
#include "kernels.h"
#include "dither_pattern.inc"

#include "env_switches.h"  // (the switches themselves live in entropy.cc: that file is also built alone, by the fuzz harnesses)
extern "C" __attribute__((visibility("default"))) void jxlhip_debug_reload_env(void) {
  std::lock_guard<std::mutex> lock(jxlhip_env::g.mu);
  jxlhip_env::LoadLocked();
}

using namespace jxlhip;

namespace {

constexpr int kPoolStreams = 8;
constexpr int kMaxBlockStreams = 8;
constexpr int kMaxBands = 64;
// Counter blocks (kCountStride u32 each): [0, kMaxBands) the bands of a direct decode (one-band frames alternate between
// blocks 0 and 1, each frame's k_prepare zeroing the next one's); a frame recorded into a hipGraph uses the same indices
// + kCaptureBase, blocks no direct call ever touches -- a replay dirties its blocks behind the host's back, and the
// host's "clean" flags describe blocks 0 / 1 only (round 5 put captured frames on block 0: a replay between two direct
// calls left k_prepare starting on non-zero counters).
constexpr int kCaptureBase = kMaxBands;
constexpr int kCountBlocks = 2 * kMaxBands;
// pinned staging buffers of jxlhip_ac_group_decode_submit (0.4 / 0.8 MB each): kStageSlotsFirst at first use, one more
// whenever a thread would otherwise have to wait for an upload to finish, up to kStageSlots.  (An upload is microseconds
// of PCIe, but the runtime now and then sits on a queued copy for 10-30 ms -- profiles/r04_e2e_waits.txt -- and with 32
// slots for 64 decoding threads that stall became every thread's.)
constexpr int kStageSlots = 128, kStageSlotsFirst = 32;
// (slots are pinned kStageChunk at a time: one hipHostMalloc of 12 MB takes a tenth of the time of 32 of 0.4 MB, and a
// context's first frame -- all a one-shot tool ever decodes -- waited for them)
constexpr int kStageChunk = 32;

// jxlhip_profile_enable: ONE event between consecutive launches (it ends the span of the launch before it and starts
// the span of the one after: rounds 1-5 recorded two, and the pass inflated every launch by ~9 %)
struct ProfMarkRec {
  hipEvent_t ev;
  int slot_after;  // kernel slot of the span that STARTS at this event; < 0: none (the end of a group of launches)
};

}  // namespace

struct jxlhip_ctx {
  // jxlhip_create_multi: the context is a PARENT over one child context per device (a device may be
  // listed more than once); frame-level calls fan out to the children, each of which decodes a stripe
  // of group rows.  A parent owns no device memory of its own except the halo staging below.
  std::vector<jxlhip_ctx*> children;
  std::vector<std::pair<uint32_t, uint32_t>> stripes;  // (group_y0, group_rows) per child of the current frame
  std::vector<float*> halo_send[2], halo_recv[2];      // per child: dense [3][halo][xsize] staging (0: up, 1: down)
  std::vector<size_t> halo_floats;
  std::vector<uint8_t*> stripe_out;                    // per child: its output stripe when the frame goes to another device / the host
  std::vector<size_t> stripe_out_bytes;
  std::vector<hipEvent_t> ev_halo[2], ev_pull[2], ev_done;
  JxlMemoryManagerHip mm{};                            // jxlhip_create_ex / _multi: who allocated this object
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // the one launches go to
  char err[512] = {0};
  bool have_frame = false;
  bool have_inputs = false;
  bool blocks_done = false;
  // One-band decodes of whole frames alternate between counter blocks 0 and 1: k_prepare of frame N zeroes the block
  // frame N + 1 will use (DevFrame::zero_counts) -- no memset launch per frame.  clean[b]: block b is all zero.
  int counts_slot = 0;
  bool counts_clean[2] = {false, false};
  double cs_phase_ms[8] = {};  // jxlhip_codestream_phase_ms
  int concurrency = 1;  // jxlhip_set_concurrency_hint: contexts the caller keeps busy on this device at a time
  bool handover_fresh = false;  // frame_begin started the hand-over and upload_side_info has not been called since
  bool blocks_fused = false;  // jxlhip_decode_blocks ran in fused-stripe mode: the planes lack the inner DCT8 blocks
  jxlhip_frame_params p{};
  DevFrame f{};
  FilterParams fp{};
  SharpLut lut{};
  // context-owned device memory
  float* planes = nullptr;  // 3 planes
  size_t planes_floats = 0;
  unsigned char* orient_dev = nullptr;  // undo_orientation: the frame in coded orientation (jxlhip_decode_frame)
  size_t orient_bytes = 0;
  float* planes2 = nullptr;  // epf_iters == 3: EPF0 output, the EPF1 + EPF2 march's input (kernels_epf0.hip)
  size_t planes2_floats = 0;
  float* inv_sigma = nullptr;
  size_t sigma_floats = 0;
  WorkItem* lists = nullptr;
  size_t lists_items = 0;
  uint32_t* counts = nullptr;    // kNumClasses
  int32_t* error_flag = nullptr; // [0] stream error, [1] table status
  float* tables = nullptr;       // wc[512] + resample[64]
  jxlhip_quant_encoding* quant_enc = nullptr;  // device: the 17 resolved encodings of the last table build
  jxlhip_quant_encoding quant_enc_host[JXLHIP_NUM_QUANT_TABLES];
  WorkLists wl{};
  uint32_t max_items[kNumClasses] = {0};
  // upload path
  void* up_coeffs[3] = {nullptr, nullptr, nullptr};
  size_t up_coeff_bytes = 0;
  uint32_t up_groups = 0;  // geometry the upload buffers were laid out for
  size_t up_esz = 0;
  uint8_t* up_side = nullptr;  // one slab: acs, quant, sharp, ytox, ytob, dc*3, dequant
  size_t up_side_bytes = 0;
  jxlhip_frame_inputs up_inputs{};
  // sparse coefficient hand-off (jxlhip_ac_group_decode_submit, single-pass 16-bit frames): a group's non-zero
  // coefficients go up as (position << 16 | value) words into sp_dev + group * kSparseStride; BeginDecode expands the
  // groups whose sp_mode byte is set into the dense upload buffer (k_expand_sparse)
  bool sparse_upload = true;  // JXLHIP_SPARSE_UPLOAD=0 turns it off
  uint8_t* sp_dev = nullptr;
  size_t sp_bytes = 0;
  uint32_t frame_serial = 0;
  std::atomic<bool> sp_any{false};
  std::atomic<size_t> sp_arena_used{0};   // sp_dev is a per-frame bump arena: a staging slot's worth of groups per copy
  uint32_t* sp_off_host[2] = {nullptr, nullptr};  // pinned, per frame parity: arena offset / 16 of every group's header,
  size_t sp_off_items = 0;                        // 0xFFFFFFFF = the group was handed over densely
  hipEvent_t sp_off_ev[2] = {nullptr, nullptr};   // "the copy of sp_off_host[parity] has executed"
  bool sp_off_pending[2] = {false, false};
  uint32_t* sp_off_dev = nullptr;
  size_t sp_off_dev_items = 0;
  hipStream_t pool[kPoolStreams] = {nullptr};
  hipEvent_t pool_ev[kPoolStreams] = {nullptr};
  hipEvent_t frame_ev = nullptr;  // jxlhip_frame_begin: "everything queued for the previous frame", see there
  bool pool_dirty[kPoolStreams] = {false};
  std::mutex pool_mu;
  uint32_t pool_next = 0;
  // entropy-decode staging: pinned host buffers (3 channels x 65536 coefficients
  // each), reused round-robin; stage_ev[i] fires when slot i's upload is done
  void* stage[kStageSlots] = {nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr};
  int stage_state[kStageSlots] = {0};  // 0 free, 1 owned by a decoding thread, 2 upload queued (stage_ev)
  int stage_count = 0;                 // slots allocated so far (<= stage_cap), kStageChunk at a time
  int stage_cap = kStageSlots;         // JXLHIP_STAGE_SLOTS (read at jxlhip_create): pinned host memory per context is at
                                       // most stage_cap x 0.8 MB -- several contexts per device share the host's lockable memory
  void* stage_chunk[kStageSlots / kStageChunk] = {nullptr};  // the allocations the slots are carved from
  size_t stage_bytes = 0;
  std::mutex stage_mu;
  std::condition_variable stage_cv;
  // dc scratch
  float* dc_tmp = nullptr;
  size_t dc_tmp_floats = 0;
  uint8_t* dc_prec = nullptr;  // per-DC-group extra_precision of jxlhip_dequant_dc_groups
  size_t dc_prec_bytes = 0;
  uint8_t* host_frame_dev = nullptr;  // jxlhip_decode_frame_host: the device frame in front of the D2H copy
  size_t host_frame_bytes = 0;
  void* pinned_frame = nullptr;  // jxlhip_decode_frame_pinned: context-owned pinned host frame (StageAlloc)
  size_t pinned_frame_bytes = 0;
  float* alpha_dev = nullptr;  // jxlhip_set_alpha: the frame's alpha plane (xsize floats per row)
  size_t alpha_items = 0;
  void* alpha_host = nullptr;  // jxlhip_alpha_staging: pinned plane the caller fills
  size_t alpha_host_items = 0;
  int32_t* qdc_dev = nullptr;  // jxlhip_decode_codestream: the quantized DC planes on their way to jxlhip_dequant_dc_groups
  size_t qdc_dev_items = 0;
  // transform-kernel fan-out (JXLHIP_BLOCK_STREAMS: 3 = one stream per family; default 1 = back to back on the
  // main stream, measured 15 % faster than letting the families compete for the CUs)
  int nblock_streams = 1;
  hipStream_t bstreams[kMaxBlockStreams] = {nullptr};
  hipEvent_t bev[kMaxBlockStreams] = {nullptr};
  hipEvent_t fork_ev = nullptr;
  int band_rows = 0;  // JXLHIP_BAND_ROWS: group rows per band of decode_frame (0 = whole stripe, the default:
                      // measured on MI355X, bands of 1-9 group rows under-fill the chip and lose 10-70 %)
  bool generic_filters = false;  // JXLHIP_FILTERS=generic: LDS kernel for every stage list
  int mfma = -1;                 // DCT32X32 / DCT16X16 on the matrix cores (kernels_mfma.hip; the 16x16 rule is in
                                 // LaunchBlocksBand).  -1 (default): when the caller's
                                 // used_acs says DCT32X32 is the only class of the row-per-lane 32-point family in
                                 // the frame (the class kernel then is a launch of its own anyway; measured on c5:
                                 // 219 -> 193 us); on mixed frames the butterflies inside the merged launch win
                                 // (c3: blocks 95 -> 105 us with a separate MFMA launch).  JXLHIP_MFMA=0 / 1 forces.
  int fuse = -1;                 // the fused kernel (kernels_fused.hip) in jxlhip_decode_frame.  -1 (default): for
                                 // frames of 12 Mpx and more -- a fused wave pays its halo rows and a fill per 8 rows,
                                 // which only amortises when the frame gives every resident wave enough rows (8K d1.0:
                                 // fused 89.9 vs 80 Gpx/s two-phase; 6144x3456: 88.5 vs 77.4; 5120x2880: 87.5 vs 82.2;
                                 // 4K: 71.2 vs 80.3; 1024^2: 16.9 vs 18.3; profiles/r02_fused_rows_sweep*.txt,
                                 // r02_fused_size_threshold.txt).  JXLHIP_FUSE=0 / 1 forces.
  uint2* cell_info = nullptr;    // fused mode: per-cell coefficient offset + quant / CfL word (k_prepare)
  size_t cell_info_items = 0;
  // profiling
  bool profiling = false;
  std::vector<ProfMarkRec> marks;
};

namespace {

int Fail(jxlhip_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIPCHK(c, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess)                                                            \
      return Fail(c, e_ == hipErrorOutOfMemory ? JXLHIP_ERR_OUT_OF_MEMORY           \
                                               : JXLHIP_ERR_HIP,                     \
                  "%s: %s", #call, hipGetErrorString(e_));                           \
  } while (0)

template <typename T>
int Grow(jxlhip_ctx* c, T** ptr, size_t* have, size_t need) {
  if (need <= *have && *ptr) return JXLHIP_OK;
  if (*ptr) HIPCHK(c, hipFree(*ptr));
  *ptr = nullptr;
  *have = 0;
  HIPCHK(c, hipMalloc((void**)ptr, need * sizeof(T)));
  *have = need;
  return JXLHIP_OK;
}

void ProfBegin(jxlhip_ctx* c) {
  if (!c->profiling) return;
  hipEvent_t e;
  (void)hipEventCreate(&e);
  (void)hipEventRecord(e, c->stream);
  c->marks.push_back({e, -1});
}
// closes the span [previous mark, now) for `slot` and opens the next one
void ProfMark(jxlhip_ctx* c, int slot) {
  if (!c->profiling || c->marks.empty()) return;
  c->marks.back().slot_after = slot;
  hipEvent_t e;
  (void)hipEventCreate(&e);
  (void)hipEventRecord(e, c->stream);
  c->marks.push_back({e, -1});
}
void ProfEnd(jxlhip_ctx* c) { (void)c; }  // (the last mark's slot_after stays -1: nothing starts there)

// Pinned staging slots: memory from the caller's JxlMemoryManager when there is one (pinned in place
// with hipHostRegister: "caller owns the host memory, the library pins it"), else hipHostMalloc.
int StageAlloc(jxlhip_ctx* c, void** p, size_t bytes) {
  if (c->mm.alloc) {
    *p = c->mm.alloc(c->mm.opaque, bytes);
    if (!*p) return JXLHIP_ERR_OUT_OF_MEMORY;
    if (hipHostRegister(*p, bytes, hipHostRegisterDefault) != hipSuccess) {
      c->mm.free(c->mm.opaque, *p);
      *p = nullptr;
      return JXLHIP_ERR_OUT_OF_MEMORY;
    }
    return JXLHIP_OK;
  }
  return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess ? JXLHIP_OK : JXLHIP_ERR_OUT_OF_MEMORY;
}
void StageFree(jxlhip_ctx* c, void* p) {
  if (c->mm.alloc) {
    (void)hipHostUnregister(p);
    c->mm.free(c->mm.opaque, p);
  } else {
    (void)hipHostFree(p);
  }
}

void MultiDestroy(jxlhip_ctx* c);
int MultiFrameBegin(jxlhip_ctx* c, const jxlhip_frame_params* p);
int MultiOwner(const jxlhip_ctx* c, uint32_t group_idx);
int MultiDecodeFrame(jxlhip_ctx* c, void* out_dev, void* host_out, size_t out_stride, size_t out_plane_stride);
int MultiSync(jxlhip_ctx* c);
int MultiCheck(jxlhip_ctx* c, jxlhip_ctx* child, int rc);
#define JXLHIP_NO_MULTI(c)                                                                                   \
  do {                                                                                                       \
    if ((c) && !(c)->children.empty())                                                                       \
      return Fail((c), JXLHIP_ERR_UNSUPPORTED, "%s is not available on a multi-device context", __func__); \
  } while (0)

}  // namespace

extern "C" {

// ---- static helpers -------------------------------------------------------
int jxlhip_covered_blocks_x(int s) {
  return (s >= 0 && s < JXLHIP_NUM_STRATEGIES) ? kCoveredX[s] : 0;
}
int jxlhip_covered_blocks_y(int s) {
  return (s >= 0 && s < JXLHIP_NUM_STRATEGIES) ? kCoveredY[s] : 0;
}
int jxlhip_log2_covered_blocks(int s) {
  if (s < 0 || s >= JXLHIP_NUM_STRATEGIES) return -1;
  int n = kCoveredX[s] * kCoveredY[s], l = 0;
  while ((1 << l) < n) l++;
  return l;
}
int jxlhip_quant_table_of_strategy(int s) {
  return (s >= 0 && s < JXLHIP_NUM_STRATEGIES) ? kQuantKind[s] : -1;
}
size_t jxlhip_dequant_table_offset(int s, int c) {
  if (s < 0 || s >= JXLHIP_NUM_STRATEGIES || c < 0 || c > 2) return (size_t)-1;
  const int kind = kQuantKind[s];
  return DequantOffset(s) + (size_t)c * 64u * kKindShort[kind] * kKindLong[kind];
}
const char* jxlhip_status_string(int status) {
  switch (status) {
    case JXLHIP_OK: return "ok";
    case JXLHIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case JXLHIP_ERR_NO_DEVICE: return "no HIP device";
    case JXLHIP_ERR_OUT_OF_MEMORY: return "out of device memory";
    case JXLHIP_ERR_HIP: return "HIP runtime error";
    case JXLHIP_ERR_BAD_STREAM: return "side info violates a format constraint";
    case JXLHIP_ERR_STATE: return "call sequence error";
    case JXLHIP_ERR_UNSUPPORTED: return "stream feature outside this back-end";
    case JXLHIP_ERR_RANGE: return "coefficient outside the 16-bit range: redo the frame with JXLHIP_COEFF_I32";
    default: return "unknown status";
  }
}

// ---- context ----------------------------------------------------------------
static jxlhip_ctx* NewCtx(const JxlMemoryManagerHip* mm) {
  JxlMemoryManagerHip m{};
  if (mm) m = *mm;
  void* mem = m.alloc ? m.alloc(m.opaque, sizeof(jxlhip_ctx)) : malloc(sizeof(jxlhip_ctx));
  if (!mem) return nullptr;
  jxlhip_ctx* c = new (mem) jxlhip_ctx();
  c->mm = m;
  return c;
}
static void DeleteCtx(jxlhip_ctx* c) {
  const JxlMemoryManagerHip m = c->mm;
  c->~jxlhip_ctx();
  if (m.free) m.free(m.opaque, c);
  else free(c);
}

int jxlhip_create(int device, jxlhip_ctx** out) { return jxlhip_create_ex(device, nullptr, out); }

int jxlhip_create_ex(int device, const JxlMemoryManagerHip* memory_manager, jxlhip_ctx** out) {
  if (!out) return JXLHIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  // both callbacks or none (lib/threads/thread_parallel_runner.cc:37-53, lib/jxl/memory_manager_internal.h)
  if (memory_manager && ((memory_manager->alloc == nullptr) != (memory_manager->free == nullptr)))
    return JXLHIP_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return JXLHIP_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return JXLHIP_ERR_INVALID_ARGUMENT;
  jxlhip_ctx* c = NewCtx(memory_manager);
  if (!c) return JXLHIP_ERR_OUT_OF_MEMORY;
  c->device = device;
  {  // the debug / test switches follow the environment as it is when a context is created (and at no other time)
    std::lock_guard<std::mutex> lock(jxlhip_env::g.mu);
    jxlhip_env::LoadLocked();
  }
  {
    const char* e = getenv("JXLHIP_FILTERS");
    c->generic_filters = e && !strcmp(e, "generic");
    const char* b = getenv("JXLHIP_BLOCK_STREAMS");
    if (b) c->nblock_streams = atoi(b);
    const char* su = getenv("JXLHIP_SPARSE_UPLOAD");
    if (su) c->sparse_upload = atoi(su) != 0;
    const char* fu = getenv("JXLHIP_FUSE");
    if (fu) c->fuse = atoi(fu) != 0 ? 1 : 0;
    const char* mf = getenv("JXLHIP_MFMA");
    if (mf) c->mfma = atoi(mf) != 0 ? 1 : 0;
    if (const char* ss = getenv("JXLHIP_STAGE_SLOTS")) {  // whole chunks, at least the first allocation
      const int v = (atoi(ss) + kStageChunk - 1) / kStageChunk * kStageChunk;
      c->stage_cap = v < kStageSlotsFirst ? kStageSlotsFirst : (v > kStageSlots ? kStageSlots : v);
    }
    const char* br = getenv("JXLHIP_BAND_ROWS");
    if (br) c->band_rows = atoi(br);
    if (c->band_rows < 0) c->band_rows = 0;
    if (c->nblock_streams < 1) c->nblock_streams = 1;
    if (c->nblock_streams > kMaxBlockStreams) c->nblock_streams = kMaxBlockStreams;
  }
  auto fail = [&](int code) {
    jxlhip_destroy(c);
    return code;
  };
  if (hipSetDevice(device) != hipSuccess) return fail(JXLHIP_ERR_HIP);
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess)
    return fail(JXLHIP_ERR_HIP);
  c->stream = c->own_stream;
  for (int i = 0; i < kPoolStreams; i++) {
    if (hipStreamCreateWithFlags(&c->pool[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->pool_ev[i], hipEventDisableTiming) != hipSuccess)
      return fail(JXLHIP_ERR_HIP);
  }
  for (int i = 0; i < c->nblock_streams; i++) {
    if (hipStreamCreateWithFlags(&c->bstreams[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->bev[i], hipEventDisableTiming) != hipSuccess)
      return fail(JXLHIP_ERR_HIP);
  }
  if (hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->frame_ev, hipEventDisableTiming) != hipSuccess)
    return fail(JXLHIP_ERR_HIP);
  if (hipMalloc((void**)&c->counts, sizeof(uint32_t) * kCountStride * kCountBlocks) != hipSuccess ||
      hipMalloc((void**)&c->error_flag, sizeof(int32_t) * 2) != hipSuccess ||
      hipMalloc((void**)&c->tables, sizeof(float) * (512 + 64 + 1024 + 2048 + 256)) != hipSuccess ||
      hipMalloc((void**)&c->quant_enc, sizeof(jxlhip_quant_encoding) * JXLHIP_NUM_QUANT_TABLES) != hipSuccess)
    return fail(JXLHIP_ERR_OUT_OF_MEMORY);
  if (hipMemset(c->error_flag, 0, sizeof(int32_t) * 2) != hipSuccess ||
      hipMemcpy(c->tables, kWcHost, sizeof(float) * 512, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->tables + 512, kResampleUpHost, sizeof(float) * 64, hipMemcpyHostToDevice) !=
          hipSuccess ||
      hipMemcpy(c->tables + 576, kDitherPattern, sizeof(float) * 1024, hipMemcpyHostToDevice) !=
          hipSuccess)
    return fail(JXLHIP_ERR_HIP);
  {
    float mfma_tab[2048 + 256];
    MfmaDct32Constants(mfma_tab);
    MfmaDct16Constants(mfma_tab + 2048);
    if (hipMemcpy(c->tables + 1600, mfma_tab, sizeof(mfma_tab), hipMemcpyHostToDevice) != hipSuccess)
      return fail(JXLHIP_ERR_HIP);
  }
  *out = c;
  return JXLHIP_OK;
}

void jxlhip_destroy(jxlhip_ctx* c) {
  if (!c) return;
  if (!c->children.empty()) return MultiDestroy(c);
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto& m : c->marks)
    if (m.ev) (void)hipEventDestroy(m.ev);
  for (int i = 0; i < kPoolStreams; i++) {
    if (c->pool[i]) {
      (void)hipStreamSynchronize(c->pool[i]);
      (void)hipStreamDestroy(c->pool[i]);
    }
    if (c->pool_ev[i]) (void)hipEventDestroy(c->pool_ev[i]);
  }
  for (int i = 0; i < kMaxBlockStreams; i++) {
    if (c->bstreams[i]) {
      (void)hipStreamSynchronize(c->bstreams[i]);
      (void)hipStreamDestroy(c->bstreams[i]);
    }
    if (c->bev[i]) (void)hipEventDestroy(c->bev[i]);
  }
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  if (c->frame_ev) (void)hipEventDestroy(c->frame_ev);
  for (int i = 0; i < kStageSlots; i++) {
    if (c->stage_ev[i]) {
      if (c->stage_state[i] == 2) (void)hipEventSynchronize(c->stage_ev[i]);
      (void)hipEventDestroy(c->stage_ev[i]);
    }
  }
  for (void*& chunk : c->stage_chunk) {
    if (chunk) StageFree(c, chunk);
    chunk = nullptr;
  }
  if (c->pinned_frame) StageFree(c, c->pinned_frame);
  if (c->sp_dev) (void)hipFree(c->sp_dev);
  if (c->alpha_host) StageFree(c, c->alpha_host);
  if (c->sp_off_dev) (void)hipFree(c->sp_off_dev);
  for (int i = 0; i < 2; i++) {
    if (c->sp_off_host[i]) StageFree(c, c->sp_off_host[i]);
    if (c->sp_off_ev[i]) (void)hipEventDestroy(c->sp_off_ev[i]);
  }
  void* bufs[] = {c->planes, c->inv_sigma, c->lists,        c->counts,
                  c->error_flag, c->tables, c->up_coeffs[0], c->up_side,
                  c->dc_tmp,     c->quant_enc,  c->dc_prec,      c->cell_info,
                  c->qdc_dev,    c->host_frame_dev, c->planes2, c->orient_dev,
                  c->alpha_dev};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  DeleteCtx(c);
}

const char* jxlhip_last_error(const jxlhip_ctx* c) { return c ? c->err : ""; }

int jxlhip_set_stream(jxlhip_ctx* c, void* hip_stream, int external) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) return jxlhip_set_stream(c->children[0], hip_stream, external);  // the frame's consumer is on devices[0]
  const hipStream_t st = external ? (hipStream_t)hip_stream : c->own_stream;
  // the zeroing of the next frame's counter block is ordered on the OLD stream only: a new stream starts with a memset
  if (st != c->stream) c->counts_clean[0] = c->counts_clean[1] = false;
  c->stream = st;
  return JXLHIP_OK;
}

// ---- frame set-up -------------------------------------------------------------
// A new hand-over of a frame's data begins (jxlhip_frame_begin, and jxlhip_upload_side_info: the same frame may be
// handed over again without a new frame_begin).  Frames may follow each other without a jxlhip_sync: the group
// uploads travel on the pool streams into buffers the previous decode's kernels (main stream) may still be reading,
// so the pool streams wait for everything queued on the main stream; the sparse arena and its offset table start empty.
static int BeginHandover(jxlhip_ctx* c) {
  const DevFrame& f = c->f;
  c->frame_serial++;
  c->sp_any.store(false);
  c->sp_arena_used.store(0);
  if (c->sparse_upload && f.coeff_type == JXLHIP_COEFF_I16) {
    const size_t ng = (size_t)f.xsg * f.ysg;
    const int par = (int)(c->frame_serial & 1u);
    if (c->sp_off_items < ng) {
      for (int i = 0; i < 2; i++) {
        if (c->sp_off_pending[i]) (void)hipEventSynchronize(c->sp_off_ev[i]);
        c->sp_off_pending[i] = false;
        if (c->sp_off_host[i]) StageFree(c, c->sp_off_host[i]);
        c->sp_off_host[i] = nullptr;
        if (StageAlloc(c, (void**)&c->sp_off_host[i], ng * 4)) return Fail(c, JXLHIP_ERR_OUT_OF_MEMORY, "sparse offset table");
        if (!c->sp_off_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->sp_off_ev[i], hipEventDisableTiming));
      }
      c->sp_off_items = ng;
    }
    // the table of two hand-overs ago has long been copied; make sure before it is overwritten
    if (c->sp_off_pending[par]) HIPCHK(c, hipEventSynchronize(c->sp_off_ev[par]));
    c->sp_off_pending[par] = false;
    memset(c->sp_off_host[par], 0xFF, ng * 4);
  }
  if (c->up_coeffs[0]) {
    HIPCHK(c, hipEventRecord(c->frame_ev, c->stream));
    for (int i = 0; i < kPoolStreams; i++) HIPCHK(c, hipStreamWaitEvent(c->pool[i], c->frame_ev, 0));
  }
  return JXLHIP_OK;
}

int jxlhip_frame_begin(jxlhip_ctx* c, const jxlhip_frame_params* p) {
  if (c && p && p->undo_orientation > 8) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "undo_orientation %u", p->undo_orientation);
  if (!c || !p) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) return MultiFrameBegin(c, p);
  if (p->xsize == 0 || p->ysize == 0 || p->xsize > (1u << 19) || p->ysize > (1u << 19))
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "frame size %ux%u out of range", p->xsize,
                p->ysize);
  if (p->coeff_type > JXLHIP_COEFF_I32 || p->output_kind > JXLHIP_OUT_PACKED)
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad coeff_type/output_kind");
  if (p->output_kind == JXLHIP_OUT_PACKED) {
    const jxlhip_output_format& o = p->out_format;
    const uint32_t max_bits = o.sample_type == JXLHIP_SAMPLE_U8 ? 8 : 16;
    if (o.transfer > JXLHIP_TF_HLG || o.sample_type > JXLHIP_SAMPLE_F16 ||
        ((o.transfer == JXLHIP_TF_PQ || o.transfer == JXLHIP_TF_GAMMA || o.transfer == JXLHIP_TF_HLG) &&
         !(o.tf_param > 0.0f)) ||
        (o.num_channels != 3 && o.num_channels != 4) || o.swap_endianness > 1 ||
        ((o.sample_type == JXLHIP_SAMPLE_U8 || o.sample_type == JXLHIP_SAMPLE_U16) &&
         (o.bits_per_sample == 0 || o.bits_per_sample > max_bits)))
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad output format");
  }
  if (p->global_scale <= 0 || p->quant_dc <= 0 || p->cfl_color_factor == 0)
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad quantizer parameters");
  if (p->lf.gab > 1 || p->lf.epf_iters > 3)
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad loop filter parameters");
  HIPCHK(c, hipSetDevice(c->device));
  DevFrame f{};
  f.xsize = p->xsize;
  f.ysize = p->ysize;
  f.xsb = (p->xsize + 7) / 8;
  f.ysb = (p->ysize + 7) / 8;
  f.xsg = (p->xsize + 255) / 256;
  f.ysg = (p->ysize + 255) / 256;
  f.xtiles = (f.xsb + 7) / 8;
  f.group_y0 = p->stripe_group_y0;
  f.group_rows = p->stripe_group_rows ? p->stripe_group_rows : f.ysg - f.group_y0;
  f.used_acs = p->used_acs & 0x7FFFFFFu;
  if (f.group_y0 >= f.ysg || f.group_y0 + f.group_rows > f.ysg)
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "stripe [%u,+%u) outside %u group rows",
                f.group_y0, f.group_rows, f.ysg);
  f.y0 = f.group_y0 * 256;
  f.y1 = (f.group_y0 + f.group_rows) * 256;
  if (f.y1 > f.ysize) f.y1 = f.ysize;
  f.fy0 = f.y0;
  f.fy1 = f.y1;
  f.band_g0 = f.group_y0;
  f.band_g1 = f.group_y0 + f.group_rows;
  static const uint32_t kEpfPad[4] = {0, 2, 3, 6};  // loop_filter.h:26-29
  f.halo = kEpfPad[p->lf.epf_iters] + p->lf.gab;
  f.coeff_type = p->coeff_type;
  f.inv_global_scale = (float)(1.0 * 65536 / p->global_scale);   // quantizer.h:82-85
  f.quant_scale = (float)(p->global_scale * (1.0 / 65536));
  f.x_dm = p->x_dm_multiplier;
  f.b_dm = p->b_dm_multiplier;
  memcpy(f.biases, p->quant_biases, sizeof(f.biases));
  f.cfl_base_x = p->cfl_base_x;
  f.cfl_base_b = p->cfl_base_b;
  f.color_scale = 1.0f / (float)p->cfl_color_factor;
  // XYB planes, block-major: the stripe's block rows plus one tile row above
  // and below for the halo rows (halo <= 7 < 8)
  const uint32_t rows_blocks = (f.group_y0 + f.group_rows) * 32 > f.ysb
                                   ? f.ysb * 8 - f.y0
                                   : f.group_rows * 256;
  f.tile_stride = (f.xsb + 1) & ~1u;
  f.plane_y0 = (int32_t)f.y0 - 8;
  f.plane_tile_rows = rows_blocks / 8 + 2;
  const size_t plane_floats = (size_t)f.plane_tile_rows * f.tile_stride * 64;
  int rc;
  if ((rc = Grow(c, &c->planes, &c->planes_floats, 3 * plane_floats))) return rc;
  for (int ch = 0; ch < 3; ch++) f.xyb[ch] = c->planes + ch * plane_floats;
  if (p->lf.epf_iters == 3 && (rc = Grow(c, &c->planes2, &c->planes2_floats, 3 * plane_floats))) return rc;
  if ((rc = Grow(c, &c->inv_sigma, &c->sigma_floats, (size_t)f.xsb * f.ysb))) return rc;
  f.inv_sigma = c->inv_sigma;
  f.error_flag = c->error_flag;
  // work lists, worst case per class
  const size_t cells = (size_t)f.xsg * f.group_rows * 1024;
  size_t total = 0;
  size_t offs[kNumClasses];
  for (int k = 0; k < kNumClasses; k++) {
    offs[k] = total;
    const size_t m = cells / ClassMinCovered(k);
    c->max_items[k] = (uint32_t)m;
    total += m;
  }
  // +64: transform kernels fetch their list entry before they know the count
  if ((rc = Grow(c, &c->lists, &c->lists_items, total + 64))) return rc;
  for (int k = 0; k < kNumClasses; k++) c->wl.list[k] = c->lists + offs[k];
  c->wl.count = c->counts;
  // stage parameters, computed as the reference stages do
  FilterParams fp{};
  for (int ch = 0; ch < 3; ch++) {
    float w0 = 1.0f, w1 = p->lf.gab_weights[2 * ch], w2 = p->lf.gab_weights[2 * ch + 1];
    const float div = w0 + 4 * (w1 + w2);  // stage_gaborish.cc:36-53
    const float mul = 1.0f / div;
    fp.gab_w[ch][0] = w0 * mul;
    fp.gab_w[ch][1] = w1 * mul;
    fp.gab_w[ch][2] = w2 * mul;
    fp.ch_scale[ch] = p->lf.epf_channel_scale[ch];
    fp.opsin_bias[ch] = p->opsin_biases[ch];
    fp.cbrt_bias[ch] = cbrtf(p->opsin_biases[ch]);  // dec_xyb.cc:158-161
  }
  // stage_epf.cc:98-115,237-255,428-446
  fp.sm[0] = (float)(p->lf.epf_pass0_sigma_scale * 1.65);
  fp.sm[1] = 1.65f;
  fp.sm[2] = (float)(p->lf.epf_pass2_sigma_scale * 1.65);
  for (int i = 0; i < 3; i++) fp.bsm[i] = fp.sm[i] * p->lf.epf_border_sad_mul;
  memcpy(fp.minv, p->inverse_opsin_matrix, sizeof(fp.minv));
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < 4; k++) fp.mcol[j][k] = fp.minv[3 * (k % 3) + j];
  for (int ch = 0; ch < 3; ch++) {
    fp.xyb_bias[ch] = -fp.cbrt_bias[ch];
    fp.xyb_bias[3 + ch] = fp.opsin_bias[ch];
  }
  if (p->output_kind == JXLHIP_OUT_PACKED) {
    fp.fmt = p->out_format;
    const bool is_int = fp.fmt.sample_type == JXLHIP_SAMPLE_U8 || fp.fmt.sample_type == JXLHIP_SAMPLE_U16;
    fp.sample_mul = is_int ? (float)((1u << fp.fmt.bits_per_sample) - 1u) : 1.0f;  // stage_write.cc:528
    fp.dither = c->tables + 576;
    // TF_PQ's display_scaling_factor_to_10000_nits_ (transfer_functions-inl.h:146-148)
    fp.tf_scale = fp.fmt.transfer == JXLHIP_TF_PQ ? fp.fmt.tf_param * (1.0f / 10000.0f) : fp.fmt.tf_param;
    fp.hlg_exponent = 0.0f;
    if (fp.fmt.transfer == JXLHIP_TF_HLG) {
      // HlgOOTF::ToSceneLight + HlgOOTF_Base (cms/tone_mapping-inl.h:113-119, tone_mapping.h:120-126)
      const float gamma = (1 / 1.2f) * powf(1.111f, -log2f(fp.fmt.tf_param / 1000.f));
      const float e = gamma - 1;
      if (e < -0.01f || 0.01f < e) fp.hlg_exponent = e;
    }
  }
  memcpy(c->lut.v, p->lf.epf_sharp_lut, sizeof(c->lut.v));
  {
    // undo_orientation: the kernels write coded orientation into a staging frame (DecodeFrame below), only the
    // 8-bit dither pattern has to follow the flipped coordinates already
    const uint32_t o = p->undo_orientation;
    const bool fx = o == 2 || o == 3 || o == 7 || o == 8, fy = o == 3 || o == 4 || o == 6 || o == 7;
    fp.dither_x0 = fx ? (int32_t)p->xsize - 1 : 0;
    fp.dither_xs = fx ? -1 : 1;
    fp.dither_y0 = fy ? (int32_t)p->ysize - 1 : 0;
    fp.dither_ys = fy ? -1 : 1;
  }
  c->fp = fp;
  c->f = f;
  c->p = *p;
  {
    const int rc = BeginHandover(c);
    if (rc) return rc;
    c->handover_fresh = true;
  }
  c->have_frame = true;
  c->have_inputs = false;
  c->blocks_done = false;
  return JXLHIP_OK;
}

static void ApplyInputs(jxlhip_ctx* c, const jxlhip_frame_inputs* in) {
  c->f.coef_stride64 = in == &c->up_inputs ? 3072u : 1024u;
  for (int ch = 0; ch < 3; ch++) {
    c->f.coeffs[ch] = in->coeffs[ch];
    c->f.dc[ch] = in->dc[ch];
  }
  c->f.acs = in->ac_strategy;
  c->f.raw_quant = in->raw_quant;
  c->f.sharp = in->epf_sharpness;
  c->f.ytox = in->ytox_map;
  c->f.ytob = in->ytob_map;
  c->f.dequant = in->dequant_table;
}

int jxlhip_frame_set_inputs(jxlhip_ctx* c, const jxlhip_frame_inputs* in) {
  if (!c || !in) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);  // device pointers belong to one device: use the host-upload path
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "frame_set_inputs before frame_begin");
  for (int ch = 0; ch < 3; ch++)
    if (!in->coeffs[ch] || !in->dc[ch])
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "null coeffs/dc pointer");
  if (!in->ac_strategy || !in->raw_quant || !in->ytox_map || !in->ytob_map ||
      !in->dequant_table || (c->p.lf.epf_iters > 0 && !in->epf_sharpness))
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "null side-info pointer");
  if (((uintptr_t)in->dequant_table & 15) || ((uintptr_t)in->coeffs[0] & 15) ||
      ((uintptr_t)in->coeffs[1] & 15) || ((uintptr_t)in->coeffs[2] & 15))
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "coeffs/dequant_table must be 16-byte aligned");
  ApplyInputs(c, in);
  c->have_inputs = true;
  c->blocks_done = false;
  return JXLHIP_OK;
}

// lays the side-info slab out; returns total bytes
static size_t SideLayout(const DevFrame& f, size_t off[9]) {
  const size_t nb = (size_t)f.xsb * f.ysb;
  const size_t nt = (size_t)f.xtiles * ((f.ysb + 7) / 8);
  size_t pos = 0;
  auto take = [&](size_t bytes) {
    const size_t o = pos;
    pos += (bytes + 255) & ~(size_t)255;
    return o;
  };
  off[0] = take(nb);                   // acs
  off[1] = take(nb * 4);               // raw_quant
  off[2] = take(nb);                   // sharpness
  off[3] = take(nt);                   // ytox
  off[4] = take(nt);                   // ytob
  off[5] = take(nb * 4);               // dc x
  off[6] = take(nb * 4);               // dc y
  off[7] = take(nb * 4);               // dc b
  off[8] = take(sizeof(float) * JXLHIP_DEQUANT_TABLE_FLOATS);
  return pos;
}

static int EnsureUploadBuffers(jxlhip_ctx* c) {
  const DevFrame& f = c->f;
  const size_t esz = f.coeff_type == JXLHIP_COEFF_I16 ? 2 : 4;
  // one buffer, [group][channel][65536]: a group's three channels are contiguous, so the
  // staging slot of jxlhip_ac_group_decode_submit goes up with ONE copy (three copies per group
  // = ~400 hipMemcpyAsync calls per 4K frame were a 4.5 ms serial floor: the runtime serialises
  // them whatever thread they come from)
  const size_t cbytes = (size_t)f.xsg * f.ysg * 3 * JXLHIP_GROUP_COEFFS * esz;
  if (cbytes > c->up_coeff_bytes || !c->up_coeffs[0]) {
    if (c->up_coeffs[0]) HIPCHK(c, hipFree(c->up_coeffs[0]));
    c->up_coeffs[0] = c->up_coeffs[1] = c->up_coeffs[2] = nullptr;
    c->up_coeff_bytes = 0;
    HIPCHK(c, hipMalloc(&c->up_coeffs[0], cbytes));
    c->up_coeff_bytes = cbytes;
  }
  if (c->sparse_upload && esz == 2) {  // the landing zone of the sparse hand-off (see SubmitSparse)
    const size_t need = (size_t)f.xsg * f.ysg * 3 * JXLHIP_GROUP_COEFFS * 2;
    if (c->sp_bytes < need) {
      if (c->sp_dev) HIPCHK(c, hipFree(c->sp_dev));
      c->sp_dev = nullptr;
      c->sp_bytes = 0;
      // (never cleared: k_expand_sparse reads only what the offset table points at, and those bytes were uploaded.
      // A hipMemset here runs on the NULL stream, unordered against the uploads on the non-blocking pool streams: it
      // once landed AFTER the first batch and turned a frame into its DC image.)
      HIPCHK(c, hipMalloc((void**)&c->sp_dev, need));
      c->sp_bytes = need;
    }
  }
  c->up_groups = f.xsg * f.ysg;
  c->up_esz = esz;
  c->up_coeffs[1] = (char*)c->up_coeffs[0] + (size_t)JXLHIP_GROUP_COEFFS * esz;
  c->up_coeffs[2] = (char*)c->up_coeffs[0] + 2 * (size_t)JXLHIP_GROUP_COEFFS * esz;
  size_t off[9];
  const size_t sbytes = SideLayout(f, off);
  int rc;
  if ((rc = Grow(c, &c->up_side, &c->up_side_bytes, sbytes))) return rc;
  jxlhip_frame_inputs in{};
  for (int ch = 0; ch < 3; ch++) {
    in.coeffs[ch] = c->up_coeffs[ch];
    in.dc[ch] = (const float*)(c->up_side + off[5 + ch]);
  }
  in.ac_strategy = c->up_side + off[0];
  in.raw_quant = (const int32_t*)(c->up_side + off[1]);
  in.epf_sharpness = c->up_side + off[2];
  in.ytox_map = (const int8_t*)(c->up_side + off[3]);
  in.ytob_map = (const int8_t*)(c->up_side + off[4]);
  in.dequant_table = (const float*)(c->up_side + off[8]);
  c->up_inputs = in;
  return JXLHIP_OK;
}

int jxlhip_alpha_staging(jxlhip_ctx* c, float** plane, size_t* stride_floats) {
  if (!c || !plane || !stride_floats) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "alpha_staging before frame_begin");
  const size_t need = (size_t)c->f.xsize * c->f.ysize;
  if (need > c->alpha_host_items) {
    if (c->alpha_host) {
      const int rc = jxlhip_sync(c);  // nothing may still be reading the old plane
      if (rc) return rc;
      StageFree(c, c->alpha_host);
      c->alpha_host = nullptr;
      c->alpha_host_items = 0;
    }
    if (StageAlloc(c, &c->alpha_host, need * sizeof(float))) return Fail(c, JXLHIP_ERR_OUT_OF_MEMORY, "pinned alpha plane");
    c->alpha_host_items = need;
  }
  *plane = (float*)c->alpha_host;
  *stride_floats = c->f.xsize;
  return JXLHIP_OK;
}

// The alpha channel of the current frame for 4-channel packed outputs (what reaches WriteToOutputStage as
// input channel alpha_c, stage_write.cc:350-366); frame_begin resets to "opaque".
int jxlhip_set_alpha(jxlhip_ctx* c, const float* host_plane, size_t stride_floats) {
  if (!c || !host_plane) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "set_alpha before frame_begin");
  if (!c->children.empty()) {  // every stripe takes its own rows of the plane
    for (jxlhip_ctx* k : c->children) {
      const int rc = jxlhip_set_alpha(k, host_plane, stride_floats);
      if (rc) return MultiCheck(c, k, rc);
    }
    return JXLHIP_OK;
  }
  const size_t w = c->f.xsize, y0 = c->f.y0, rows = c->f.y1 - c->f.y0;
  if (stride_floats < w) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "alpha stride %zu < xsize", stride_floats);
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = Grow(c, &c->alpha_dev, &c->alpha_items, w * rows);
  if (rc) return rc;
  // the rows of this context's stripe; the kernels index the plane by IMAGE row: the base pointer is that of row 0
  HIPCHK(c, hipMemcpy2DAsync(c->alpha_dev, w * sizeof(float), host_plane + y0 * stride_floats, stride_floats * sizeof(float),
                             w * sizeof(float), rows, hipMemcpyHostToDevice, c->stream));
  c->fp.alpha = c->alpha_dev - y0 * w;
  c->fp.alpha_stride = (uint32_t)w;
  return JXLHIP_OK;
}

int jxlhip_upload_side_info(jxlhip_ctx* c, const uint8_t* ac_strategy, const int32_t* raw_quant,
                            const uint8_t* epf_sharpness, const int8_t* ytox_map,
                            const int8_t* ytob_map, const float* const dc[3],
                            const float* dequant_table) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) {
    for (jxlhip_ctx* k : c->children) {
      const int rc = jxlhip_upload_side_info(k, ac_strategy, raw_quant, epf_sharpness, ytox_map, ytob_map, dc, dequant_table);
      if (rc) return MultiCheck(c, k, rc);
    }
    return JXLHIP_OK;
  }
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "upload_side_info before frame_begin");
  if (!ac_strategy || !raw_quant || !ytox_map || !ytob_map || !dc || !dc[0] || !dc[1] ||
      !dc[2] || !dequant_table || (c->p.lf.epf_iters > 0 && !epf_sharpness))
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "null side-info pointer");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = EnsureUploadBuffers(c))) return rc;
  // frame_begin has just started this frame's hand-over (serial, sparse table parity, arena): starting another one
  // here would advance the serial twice per frame -- the double-buffered offset table would then sit on ONE parity
  // and wait for the previous frame's copy every time -- and would drop groups submitted before the side info.  Only
  // a frame handed over AGAIN (a second upload_side_info without a frame_begin) starts over.
  if (c->handover_fresh) c->handover_fresh = false;
  else if ((rc = BeginHandover(c))) return rc;
  const DevFrame& f = c->f;
  const size_t nb = (size_t)f.xsb * f.ysb;
  const size_t nt = (size_t)f.xtiles * ((f.ysb + 7) / 8);
  const jxlhip_frame_inputs& in = c->up_inputs;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync((void*)in.ac_strategy, ac_strategy, nb, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync((void*)in.raw_quant, raw_quant, nb * 4, hipMemcpyHostToDevice, st));
  if (epf_sharpness)
    HIPCHK(c, hipMemcpyAsync((void*)in.epf_sharpness, epf_sharpness, nb, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync((void*)in.ytox_map, ytox_map, nt, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync((void*)in.ytob_map, ytob_map, nt, hipMemcpyHostToDevice, st));
  for (int ch = 0; ch < 3; ch++)
    HIPCHK(c, hipMemcpyAsync((void*)in.dc[ch], dc[ch], nb * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync((void*)in.dequant_table, dequant_table,
                           sizeof(float) * JXLHIP_DEQUANT_TABLE_FLOATS, hipMemcpyHostToDevice, st));
  ApplyInputs(c, &in);
  c->have_inputs = true;
  c->blocks_done = false;
  return JXLHIP_OK;
}

// JXLHIP_CODESTREAM_VERBOSE: the longest single wait of the upload path during one AC phase, microseconds
// [0] a pinned slot (AcquireSlot), [1] one hipMemcpyAsync call, [2] one hipEventRecord call
static std::atomic<int64_t> g_upload_wait_us[3];
static std::atomic<bool> g_upload_wait_on{false};
struct UploadWaitClock {
  int which;
  std::chrono::steady_clock::time_point t0;
  explicit UploadWaitClock(int w) : which(w) {
    if (g_upload_wait_on.load(std::memory_order_relaxed)) t0 = std::chrono::steady_clock::now();
  }
  ~UploadWaitClock() {
    if (!g_upload_wait_on.load(std::memory_order_relaxed)) return;
    const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    int64_t seen = g_upload_wait_us[which].load(std::memory_order_relaxed);
    while (seen < us && !g_upload_wait_us[which].compare_exchange_weak(seen, us, std::memory_order_relaxed)) {
    }
  }
};

static int jxlhip_submit_group_ev(jxlhip_ctx* c, uint32_t group_idx, const void* const coeffs[3],
                                  size_t ncoeffs, hipEvent_t done);

int jxlhip_submit_group(jxlhip_ctx* c, uint32_t group_idx, const void* const coeffs[3],
                        size_t ncoeffs) {
  return jxlhip_submit_group_ev(c, group_idx, coeffs, ncoeffs, nullptr);
}

// `done` (optional) is recorded behind the three copies on the slot's stream
static int jxlhip_submit_group_ev(jxlhip_ctx* c, uint32_t group_idx, const void* const coeffs[3],
                                  size_t ncoeffs, hipEvent_t done) {
  if (!c || !coeffs) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) {
    if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "submit_group before frame_begin");
    const int o = MultiOwner(c, group_idx);
    if (o < 0) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad group %u", group_idx);
    return MultiCheck(c, c->children[o], jxlhip_submit_group_ev(c->children[o], group_idx, coeffs, ncoeffs, done));
  }
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "submit_group before frame_begin");
  const DevFrame& f = c->f;
  if (group_idx >= f.xsg * f.ysg || ncoeffs > JXLHIP_GROUP_COEFFS || !coeffs[0] || !coeffs[1] ||
      !coeffs[2])
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad group %u / ncoeffs %zu", group_idx, ncoeffs);
  const size_t esz = f.coeff_type == JXLHIP_COEFF_I16 ? 2 : 4;
  int slot;
  {
    // only the bookkeeping is serialised: the copies themselves are issued concurrently by the
    // runner's threads (with the lock around them, ~400 hipMemcpyAsync calls per 4K frame were
    // a 4.5 ms serial floor of the whole upload path).  A thread issues its three copies and
    // then its event on ONE stream in program order, so the event still follows its copies
    // however other threads' calls interleave on that stream.
    std::lock_guard<std::mutex> lock(c->pool_mu);
    if (hipSetDevice(c->device) != hipSuccess) return JXLHIP_ERR_HIP;
    // (re)sized for THIS frame's group count and coefficient type: a context reused for a larger
    // frame or another coefficient type must not write past the previous frame's allocation
    if (!c->up_coeffs[0] || c->up_groups != f.xsg * f.ysg || c->up_esz != esz) {
      int rc = EnsureUploadBuffers(c);
      if (rc) return rc;
    }
    slot = (int)(c->pool_next++ % kPoolStreams);
    c->pool_dirty[slot] = true;
  }
  if (hipSetDevice(c->device) != hipSuccess) return JXLHIP_ERR_HIP;
  const size_t chan = (size_t)JXLHIP_GROUP_COEFFS * esz;
  char* dst0 = (char*)c->up_coeffs[0] + (size_t)group_idx * 3 * chan;
  if ((const char*)coeffs[1] == (const char*)coeffs[0] + chan && (const char*)coeffs[2] == (const char*)coeffs[0] + 2 * chan) {
    // the three channels sit in one staging slot: one copy up to the last used coefficient
    hipError_t e;
    {
      UploadWaitClock w(1);
      e = hipMemcpyAsync(dst0, coeffs[0], 2 * chan + ncoeffs * esz, hipMemcpyHostToDevice, c->pool[slot]);
    }
    if (e != hipSuccess) return Fail(c, JXLHIP_ERR_HIP, "submit_group: %s", hipGetErrorString(e));
  } else {
    for (int ch = 0; ch < 3; ch++) {
      hipError_t e = hipMemcpyAsync(dst0 + ch * chan, coeffs[ch], ncoeffs * esz, hipMemcpyHostToDevice, c->pool[slot]);
      if (e != hipSuccess) return Fail(c, JXLHIP_ERR_HIP, "submit_group: %s", hipGetErrorString(e));
    }
  }
  if (done && hipEventRecord(done, c->pool[slot]) != hipSuccess)
    return Fail(c, JXLHIP_ERR_HIP, "submit_group: event record failed");
  return JXLHIP_OK;
}

// ---- sparse hand-off ------------------------------------------------------------------------------------------
// A single-pass, 16-bit group crosses PCIe as its NON-ZERO coefficients: 16 header bytes (three counts) + one
// (position << 16 | value) word per non-zero, the three channels' lists back to back -- nine out of ten
// coefficients of a d1.0 frame are zero, and the dense stream is 384 KB per group whatever it holds (8K: 196 MB per
// frame).  The compact groups of one runner thread are collected in a staging slot and go up TOGETHER: one
// hipMemcpyAsync costs ~15 us inside the runtime whatever thread issues it, serialised -- 510 per-group copies
// were an 8 ms floor under an 8K frame however many threads decoded.  sp_dev is a per-frame bump arena;
// sp_off_host[parity][g] says where group g's header landed (0xFFFFFFFF: handed over densely); BeginDecode uploads
// that table and k_expand_sparse rebuilds the dense block stream (zero + scatter) behind the uploads.
// entries per channel (X, Y, B), one slot's worth in total: the luma list can take EVERY coefficient of the group (a
// noise patch at d1.0 has 45 000 non-zero luma coefficients in a group), the chroma lists a quarter each
static constexpr uint32_t kSparseCap[3] = {16382u, 65536u, 16382u};
static constexpr size_t kSparseStride = 3u * (size_t)JXLHIP_GROUP_COEFFS * 2u;     // arena bytes per group, worst case
static constexpr int kBatchGroups = 96;

struct SparseBatch {  // what one runner thread has collected (in its own heap buffer: a pinned staging slot is only
  std::vector<uint8_t> buf;  // held for the moment of the copy -- more threads than slots must not starve each other)
  size_t used = 0;
  int n = 0;
  uint32_t group[kBatchGroups];
  uint32_t at[kBatchGroups];  // byte offset of the group's header inside the slot
};

static int AcquireSlot(jxlhip_ctx* c, size_t slot_bytes, int* out);
static void ReleaseSlot(jxlhip_ctx* c, int slot, bool uploaded);


// the batch goes up as one copy through a pinned staging slot; its groups' headers are entered into the offset table
static int SparseFlush(jxlhip_ctx* c, SparseBatch* b) {
  if (b->n == 0) return JXLHIP_OK;
  int slot = -1;
  int rc;
  {
    UploadWaitClock w(0);
    rc = AcquireSlot(c, kSparseStride, &slot);
  }
  if (rc) return rc;
  memcpy(c->stage[slot], b->buf.data(), b->used);
  const size_t bytes = (b->used + 255) & ~(size_t)255;
  const size_t off = c->sp_arena_used.fetch_add(bytes);
  int stream;
  {
    std::lock_guard<std::mutex> lock(c->pool_mu);
    stream = (int)(c->pool_next++ % kPoolStreams);
    c->pool_dirty[stream] = true;
  }
  if (off + bytes > c->sp_bytes) rc = Fail(c, JXLHIP_ERR_STATE, "sparse arena overflow");
  if (!rc && hipSetDevice(c->device) != hipSuccess) rc = JXLHIP_ERR_HIP;
  if (!rc) {
    hipError_t e;
    {
      UploadWaitClock w(1);
      e = hipMemcpyAsync(c->sp_dev + off, c->stage[slot], b->used, hipMemcpyHostToDevice, c->pool[stream]);
    }
    if (e != hipSuccess) {
      rc = Fail(c, JXLHIP_ERR_HIP, "sparse submit: %s", hipGetErrorString(e));
    } else {
      UploadWaitClock w(2);
      if (hipEventRecord(c->stage_ev[slot], c->pool[stream]) != hipSuccess) rc = Fail(c, JXLHIP_ERR_HIP, "sparse submit: event record failed");
    }
  }
  if (!rc) {
    uint32_t* table = c->sp_off_host[c->frame_serial & 1u];
    for (int i = 0; i < b->n; i++) table[b->group[i]] = (uint32_t)((off + b->at[i]) >> 4);
    c->sp_any.store(true);
  }
  ReleaseSlot(c, slot, rc == JXLHIP_OK);
  b->used = 0;
  b->n = 0;
  return rc;
}

// One group, single pass, decoded into `scratch` (kSparseStride bytes) and appended to the batch.
// JXLHIP_ERR_RANGE: not representable (a chroma channel with more than kSparseCap non-zeros, a value outside 16 bits):
// the caller hands the group over densely.
static int SparseAppend(jxlhip_ctx* c, SparseBatch* b, uint8_t* scratch, const jxlhip_ac_pass* pass, uint32_t shift,
                        uint32_t group_idx, const uint8_t* ac_strategy, const int32_t* raw_quant, const uint8_t* quant_dc,
                        const uint8_t* data, size_t size, size_t* bit_pos) {
  const DevFrame& f = c->f;
  uint32_t* const ent[3] = {(uint32_t*)(scratch + 16), (uint32_t*)(scratch + 16) + kSparseCap[0],
                            (uint32_t*)(scratch + 16) + kSparseCap[0] + kSparseCap[1]};
  uint32_t cnt[3] = {0, 0, 0};
  size_t pos = *bit_pos, ncoeffs = 0;
  int rc = jxlhip_ac_group_decode_sparse(pass, f.xsb, f.ysb, group_idx % f.xsg, group_idx / f.xsg, ac_strategy, raw_quant, quant_dc,
                                         data, size, &pos, shift, ent, kSparseCap, cnt, &ncoeffs);
  if (rc) return rc;
  *bit_pos = pos;
  const size_t bytes = 16 + 4 * ((size_t)cnt[0] + cnt[1] + cnt[2]);
  if (b->buf.size() < kSparseStride) b->buf.resize(kSparseStride);
  if (b->n && (b->used + bytes > kSparseStride || b->n == kBatchGroups)) {
    if ((rc = SparseFlush(c, b))) return rc;
  }
  uint8_t* dst = b->buf.data() + b->used;
  uint32_t* hdr = (uint32_t*)dst;
  hdr[0] = cnt[0], hdr[1] = cnt[1], hdr[2] = cnt[2], hdr[3] = 0;
  memcpy(dst + 16, ent[0], (size_t)cnt[0] * 4);
  memcpy(dst + 16 + (size_t)cnt[0] * 4, ent[1], (size_t)cnt[1] * 4);
  memcpy(dst + 16 + ((size_t)cnt[0] + cnt[1]) * 4, ent[2], (size_t)cnt[2] * 4);
  b->group[b->n] = group_idx;
  b->at[b->n] = (uint32_t)b->used;
  b->n++;
  b->used += (bytes + 15) & ~(size_t)15;
  return JXLHIP_OK;
}

static bool SparseEligible(const jxlhip_ctx* c, uint32_t num_passes) {
  return c->sparse_upload && num_passes == 1 && c->f.coeff_type == JXLHIP_COEFF_I16 && c->sp_dev &&
         c->sp_bytes >= (size_t)c->f.xsg * c->f.ysg * kSparseStride && c->sp_off_items >= (size_t)c->f.xsg * c->f.ysg;
}

static int SubmitPassesImpl(jxlhip_ctx* c, uint32_t num_passes, const jxlhip_ac_pass* const* passes, const uint32_t* shifts,
                            uint32_t group_idx, const uint8_t* ac_strategy, const int32_t* raw_quant, const uint8_t* quant_dc,
                            const uint8_t* const* data, const size_t* sizes, size_t* bit_pos, bool allow_sparse,
                            std::vector<uint8_t>* dense_scratch = nullptr);

// f1: entropy-decode all passes of one AC group into a pinned staging slot and
// queue its upload.  The slot is reused only after its copies completed.
int jxlhip_ac_group_decode_submit_passes(jxlhip_ctx* c, uint32_t num_passes,
                                         const jxlhip_ac_pass* const* passes, const uint32_t* shifts,
                                         uint32_t group_idx, const uint8_t* ac_strategy,
                                         const int32_t* raw_quant, const uint8_t* quant_dc,
                                         const uint8_t* const* data, const size_t* sizes,
                                         size_t* bit_pos) {
  if (!c || !passes || !ac_strategy || !raw_quant || !data || !sizes || !bit_pos || num_passes == 0 ||
      num_passes > 11)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t p = 0; p < num_passes; p++)
    if (!passes[p] || !data[p] || (shifts && shifts[p] > 3)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "ac_group_decode_submit before frame_begin");
  if (!c->children.empty()) {
    const int o = MultiOwner(c, group_idx);
    if (o < 0) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad group %u", group_idx);
    return MultiCheck(c, c->children[o], jxlhip_ac_group_decode_submit_passes(c->children[o], num_passes, passes, shifts, group_idx, ac_strategy,
                                                                              raw_quant, quant_dc, data, sizes, bit_pos));
  }
  return SubmitPassesImpl(c, num_passes, passes, shifts, group_idx, ac_strategy, raw_quant, quant_dc, data, sizes, bit_pos, true);
}

// A free pinned staging slot (state 1 = owned by the caller); blocks while all are in flight / owned.
static int AcquireSlot(jxlhip_ctx* c, size_t slot_bytes, int* out) {
  int slot = -1;
  std::unique_lock<std::mutex> lock(c->stage_mu);
  if (hipSetDevice(c->device) != hipSuccess) return JXLHIP_ERR_HIP;
  // kStageChunk more slots (the chunk after the ones there are), free
  auto grow = [&]() -> int {
    const int k = c->stage_count / kStageChunk;
    if (c->stage_count + kStageChunk > c->stage_cap)
      return Fail(c, JXLHIP_ERR_OUT_OF_MEMORY, "pinned staging: the cap of %d slots (JXLHIP_STAGE_SLOTS) is reached", c->stage_cap);
    // the events first, then the chunk, and only then is anything published: a failure half way leaves nothing
    // behind that a later grow() would overwrite (an event that exists already is simply kept)
    for (int i = c->stage_count; i < c->stage_count + kStageChunk; i++)
      if (!c->stage_ev[i] && hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming) != hipSuccess)
        return Fail(c, JXLHIP_ERR_HIP, "event creation failed");
    void* chunk = nullptr;
    if (StageAlloc(c, &chunk, (size_t)kStageChunk * c->stage_bytes) != JXLHIP_OK)
      return Fail(c, JXLHIP_ERR_OUT_OF_MEMORY, "pinned staging allocation failed");
    c->stage_chunk[k] = chunk;
    for (int i = c->stage_count; i < c->stage_count + kStageChunk; i++) {
      c->stage[i] = (char*)chunk + (size_t)(i - c->stage_count) * c->stage_bytes;
      c->stage_state[i] = 0;
    }
    c->stage_count += kStageChunk;
    return JXLHIP_OK;
  };
  if (c->stage_bytes < slot_bytes) {
    // (re)allocation: only when no thread owns a slot
    c->stage_cv.wait(lock, [&] {
      for (int i = 0; i < c->stage_count; i++)
        if (c->stage_state[i] == 1) return false;
      return true;
    });
    if (c->stage_bytes < slot_bytes) {
      for (int i = 0; i < c->stage_count; i++) {
        if (c->stage_state[i] == 2) (void)hipEventSynchronize(c->stage_ev[i]);
        c->stage[i] = nullptr;
        c->stage_state[i] = 0;
      }
      for (void*& chunk : c->stage_chunk) {
        if (chunk) StageFree(c, chunk);
        chunk = nullptr;
      }
      c->stage_bytes = slot_bytes;
      c->stage_count = 0;
      static_assert(kStageSlotsFirst % kStageChunk == 0 && kStageSlots % kStageChunk == 0, "whole chunks");
      while (c->stage_count < kStageSlotsFirst) {
        const int rc = grow();
        if (rc) return rc;
      }
    }
  }
  while (slot < 0) {
    int pending = -1;
    for (int i = 0; i < c->stage_count && slot < 0; i++) {
      if (c->stage_state[i] == 0) slot = i;
      else if (c->stage_state[i] == 2) {
        if (hipEventQuery(c->stage_ev[i]) == hipSuccess) slot = i;
        else if (pending < 0) pending = i;
      }
    }
    if (slot >= 0) break;
    if (c->stage_count < c->stage_cap) {  // nothing free: more slots rather than a wait
      slot = c->stage_count;
      const int rc = grow();
      if (rc) return rc;
    } else if (pending >= 0) {  // every slot is in flight: wait for one upload, without keeping the others out
      hipEvent_t ev = c->stage_ev[pending];
      lock.unlock();
      const hipError_t e = hipEventSynchronize(ev);
      lock.lock();
      if (e != hipSuccess) return JXLHIP_ERR_HIP;
    } else {  // every slot is owned by another decoding thread
      c->stage_cv.wait(lock);
    }
  }
  c->stage_state[slot] = 1;
  *out = slot;
  return JXLHIP_OK;
}

static void ReleaseSlot(jxlhip_ctx* c, int slot, bool uploaded) {
  {
    std::lock_guard<std::mutex> lock(c->stage_mu);
    c->stage_state[slot] = uploaded ? 2 : 0;
  }
  c->stage_cv.notify_all();
}

static int SubmitPassesImpl(jxlhip_ctx* c, uint32_t num_passes, const jxlhip_ac_pass* const* passes, const uint32_t* shifts,
                            uint32_t group_idx, const uint8_t* ac_strategy, const int32_t* raw_quant, const uint8_t* quant_dc,
                            const uint8_t* const* data, const size_t* sizes, size_t* bit_pos, bool allow_sparse,
                            std::vector<uint8_t>* dense_scratch) {
  const DevFrame& f = c->f;
  if (group_idx >= f.xsg * f.ysg) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "bad group %u", group_idx);
  const size_t esz = f.coeff_type == JXLHIP_COEFF_I16 ? 2 : 4;
  const size_t slot_bytes = 3 * (size_t)JXLHIP_GROUP_COEFFS * esz;
  int rc = JXLHIP_ERR_RANGE;
  if (allow_sparse && SparseEligible(c, num_passes)) {  // one group = one batch (callers that submit groups one by one)
    SparseBatch b;
    std::vector<uint8_t> scratch(kSparseStride);
    rc = SparseAppend(c, &b, scratch.data(), passes[0], shifts ? shifts[0] : 0, group_idx, ac_strategy, raw_quant, quant_dc,
                      data[0], sizes[0], &bit_pos[0]);
    const int rf = SparseFlush(c, &b);
    if (rc == JXLHIP_OK) rc = rf;
  }
  if (rc == JXLHIP_ERR_RANGE) {
    // the dense form: decoded into the caller's (or a local) heap buffer; a pinned staging slot is held only for the
    // copy into it and the submit -- a textured group decodes for milliseconds, and with the slots held that long the
    // 33rd such group of a frame waited for the first to finish
    std::vector<uint8_t> local;
    std::vector<uint8_t>& buf = dense_scratch ? *dense_scratch : local;
    if (buf.size() < slot_bytes) buf.resize(slot_bytes);
    char* base = (char*)buf.data();
    void* const ch[3] = {base, base + (size_t)JXLHIP_GROUP_COEFFS * esz, base + 2 * (size_t)JXLHIP_GROUP_COEFFS * esz};
    size_t ncoeffs = 0;
    memset(base, 0, slot_bytes);  // coefficients are accumulated (dec_group.cc:527-531)
    rc = JXLHIP_OK;
    for (uint32_t p = 0; p < num_passes && rc == JXLHIP_OK; p++)
      rc = jxlhip_ac_group_decode(passes[p], f.xsb, f.ysb, group_idx % f.xsg, group_idx / f.xsg, ac_strategy,
                                  raw_quant, quant_dc, data[p], sizes[p], &bit_pos[p], shifts ? shifts[p] : 0,
                                  f.coeff_type, ch, &ncoeffs);
    if (rc == JXLHIP_OK) {
      int slot = -1;
      {
        UploadWaitClock w(0);
        rc = AcquireSlot(c, slot_bytes, &slot);
      }
      if (rc) return rc;
      char* pinned = (char*)c->stage[slot];
      const size_t chan = (size_t)JXLHIP_GROUP_COEFFS * esz;
      memcpy(pinned, base, 2 * chan + ncoeffs * esz);  // (what jxlhip_submit_group_ev sends up in one copy)
      const void* const src[3] = {pinned, pinned + chan, pinned + 2 * chan};
      rc = jxlhip_submit_group_ev(c, group_idx, src, ncoeffs, c->stage_ev[slot]);
      ReleaseSlot(c, slot, rc == JXLHIP_OK);
    }
  }
  if (rc == JXLHIP_ERR_BAD_STREAM) return Fail(c, rc, "AC group %u: invalid entropy-coded data", group_idx);
  return rc;
}

namespace {
struct GroupsJob {
  jxlhip_ctx* c;
  uint32_t num_passes, num_groups;
  const jxlhip_ac_pass* const* passes;
  const uint32_t* shifts;
  const uint8_t* acs;
  const int32_t* raw_quant;
  const uint8_t* quant_dc;
  const uint8_t* const* sections;
  const size_t* sizes;
  size_t* end_bits = nullptr;
  std::atomic<int> status{JXLHIP_OK};
  // the runner's task t is group order[t]: the sections with the most bytes first.  A group's decode time follows its
  // bytes (r = 0.98 on the 8K d1.0 stream of tests/data) and a textured patch takes five times the mean: handed out
  // last, one such group is the tail the whole frame waits for
  std::vector<uint32_t> order;
  // JXLHIP_CODESTREAM_VERBOSE=1: per task {start ms, end ms, thread}
  std::vector<float> timeline;
  std::chrono::steady_clock::time_point t0;
  // sparse hand-off: one open staging slot + one decode scratch per runner thread
  bool sparse = false;
  std::vector<SparseBatch> batch;
  std::vector<std::vector<uint8_t>> scratch;
  std::vector<std::vector<uint8_t>> dense;  // per runner thread: where a group that goes up densely is decoded
  // test hook (JXLHIP_TEST_RANGE_GROUP=g, read by GroupsInit): group g reports a coefficient beyond 16 bits on the
  // 16-bit attempt -- no stream libjxl's encoder writes at ordinary settings does, and the redo with int32 buffers
  // (through the single runner call and through the three barriers) has to be reachable by a test
  int64_t test_range_group = -1;
};
int GroupsInit(void* opaque, size_t num_threads) {
  GroupsJob* j = static_cast<GroupsJob*>(opaque);
  if (j->sparse) {
    j->batch.assign(num_threads ? num_threads : 1, SparseBatch());
    j->scratch.assign(num_threads ? num_threads : 1, std::vector<uint8_t>());
  }
  j->dense.assign(num_threads ? num_threads : 1, std::vector<uint8_t>());
  j->test_range_group = jxlhip_env::Get().test_range_group.load(std::memory_order_relaxed);
  return 0;
}
// A section this large carries more non-zeros than a chroma list of the sparse form takes (kSparseCap; the stream above:
// every group that overflowed had 32 000 bytes or more, none below 34 300 fitted with much to spare): decoded densely
// straight away instead of finding that out three quarters of the way through the sparse attempt.
static constexpr size_t kDenseFirstBytes = 30000;
void GroupsFuncBody(GroupsJob* j, uint32_t g, size_t thread);
// one group, with its entry in the timeline (keyed by group) when one is kept
void GroupsOne(GroupsJob* j, uint32_t g, size_t thread) {
  if (j->timeline.empty()) return GroupsFuncBody(j, g, thread);
  const double a = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - j->t0).count();
  GroupsFuncBody(j, g, thread);
  const double b = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - j->t0).count();
  j->timeline[3 * (size_t)g] = (float)a, j->timeline[3 * (size_t)g + 1] = (float)b, j->timeline[3 * (size_t)g + 2] = (float)thread;
}
void GroupsFunc(void* opaque, uint32_t task, size_t thread) {
  GroupsJob* j = static_cast<GroupsJob*>(opaque);
  GroupsOne(j, j->order.empty() ? task : j->order[task], thread);
}
void GroupsFuncBody(GroupsJob* j, uint32_t g, size_t thread) {
  if (j->status.load(std::memory_order_relaxed) != JXLHIP_OK) return;
  const DevFrame& f = j->c->f;
  const uint32_t gy = g / f.xsg;
  if (gy < f.group_y0 || gy >= f.group_y0 + f.group_rows) return;  // another rank's stripe
  const uint8_t* data[11];
  size_t sizes[11], pos[11];
  for (uint32_t p = 0; p < j->num_passes; p++) {
    data[p] = j->sections[(size_t)p * j->num_groups + g];
    sizes[p] = j->sizes[(size_t)p * j->num_groups + g];
    pos[p] = 0;
  }
  int rc = JXLHIP_ERR_RANGE;
  if ((int64_t)g == j->test_range_group && f.coeff_type == JXLHIP_COEFF_I16) {
    int expected = JXLHIP_OK;
    j->status.compare_exchange_strong(expected, JXLHIP_ERR_RANGE);
    return;
  }
  if (j->sparse && thread < j->batch.size() && sizes[0] < kDenseFirstBytes) {
    if (j->scratch[thread].empty()) j->scratch[thread].resize(kSparseStride);
    rc = SparseAppend(j->c, &j->batch[thread], j->scratch[thread].data(), j->passes[0], j->shifts ? j->shifts[0] : 0, g, j->acs,
                      j->raw_quant, j->quant_dc, data[0], sizes[0], &pos[0]);
    if (rc == JXLHIP_ERR_BAD_STREAM) Fail(j->c, rc, "AC group %u: invalid entropy-coded data", g);
  }
  if (rc == JXLHIP_ERR_RANGE)
    rc = SubmitPassesImpl(j->c, j->num_passes, j->passes, j->shifts, g, j->acs, j->raw_quant, j->quant_dc, data, sizes, pos, false,
                          thread < j->dense.size() ? &j->dense[thread] : nullptr);
  if (rc != JXLHIP_OK) {
    int expected = JXLHIP_OK;
    j->status.compare_exchange_strong(expected, rc);
  } else if (j->end_bits) {
    for (uint32_t p = 0; p < j->num_passes; p++) j->end_bits[(size_t)p * j->num_groups + g] = pos[p];
  }
}
}  // namespace

// (JXLHIP_CODESTREAM_VERBOSE) per thread: first start, last end, busy time; and the longest task
static void GroupsTimelineReport(const GroupsJob& job) {
  const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.t0).count();
  struct Th { float first = 1e9f, last = 0, busy = 0; int n = 0; };
  std::vector<Th> th(1024);
  int used = 0;
  float longest = 0, first_min = 1e9f, first_max = 0, last_min = 1e9f, busy_min = 1e9f, busy_max = 0;
  for (uint32_t t = 0; t < job.num_groups; t++) {
    const float a = job.timeline[3 * (size_t)t], b = job.timeline[3 * (size_t)t + 1];
    if (b <= 0) continue;
    Th& h = th[std::min<size_t>((size_t)job.timeline[3 * (size_t)t + 2], 1023)];
    h.first = std::min(h.first, a), h.last = std::max(h.last, b), h.busy += b - a, h.n++;
    longest = std::max(longest, b - a);
  }
  for (const Th& h : th) {
    if (!h.n) continue;
    used++;
    first_min = std::min(first_min, h.first), first_max = std::max(first_max, h.first), last_min = std::min(last_min, h.last);
    busy_min = std::min(busy_min, h.busy), busy_max = std::max(busy_max, h.busy);
  }
  fprintf(stderr, "[codestream] longest single wait in the upload path: pinned slot %.2f ms, hipMemcpyAsync %.2f ms, hipEventRecord %.2f ms\n",
          g_upload_wait_us[0].exchange(0) * 1e-3, g_upload_wait_us[1].exchange(0) * 1e-3, g_upload_wait_us[2].exchange(0) * 1e-3);
  fprintf(stderr, "[codestream] AC groups: %.2f ms after the runner call began, on %d threads; first group started at %.2f, last thread started at "
          "%.2f, first finished at %.2f; busy per thread %.2f .. %.2f ms; longest group %.2f ms\n", total, used, first_min, first_max, last_min,
          busy_min, busy_max, longest);
}

int jxlhip_ac_groups_decode_submit(jxlhip_ctx* c, jxlhip_parallel_runner runner, void* runner_opaque,
                                   uint32_t num_passes, const jxlhip_ac_pass* const* passes,
                                   const uint32_t* shifts, const uint8_t* ac_strategy,
                                   const int32_t* raw_quant, const uint8_t* quant_dc,
                                   const uint8_t* const* sections, const size_t* sizes) {
  return jxlhip_ac_groups_decode_submit_ex(c, runner, runner_opaque, num_passes, passes, shifts, ac_strategy, raw_quant, quant_dc,
                                           sections, sizes, nullptr);
}

int jxlhip_ac_groups_decode_submit_ex(jxlhip_ctx* c, jxlhip_parallel_runner runner, void* runner_opaque,
                                      uint32_t num_passes, const jxlhip_ac_pass* const* passes,
                                      const uint32_t* shifts, const uint8_t* ac_strategy,
                                      const int32_t* raw_quant, const uint8_t* quant_dc,
                                      const uint8_t* const* sections, const size_t* sizes, size_t* end_bits) {
  if (!c || !passes || !ac_strategy || !raw_quant || !sections || !sizes || num_passes == 0 || num_passes > 11)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "ac_groups_decode_submit before frame_begin");
  if (!c->children.empty()) {  // every child takes the groups of its stripe (GroupsFunc skips the others)
    for (jxlhip_ctx* k : c->children) {
      const int rc = jxlhip_ac_groups_decode_submit_ex(k, runner, runner_opaque, num_passes, passes, shifts, ac_strategy, raw_quant,
                                                       quant_dc, sections, sizes, end_bits);
      if (rc) return MultiCheck(c, k, rc);
    }
    return JXLHIP_OK;
  }
  GroupsJob job;
  job.c = c;
  job.num_passes = num_passes;
  job.num_groups = c->f.xsg * c->f.ysg;
  job.passes = passes;
  job.shifts = shifts;
  job.acs = ac_strategy;
  job.raw_quant = raw_quant;
  job.quant_dc = quant_dc;
  job.sections = sections;
  job.sizes = sizes;
  job.end_bits = end_bits;
  job.sparse = SparseEligible(c, num_passes);
  if (runner && job.num_groups > 1) {
    std::vector<uint64_t> key(job.num_groups);  // (bytes over all passes) << 32 | ~group: sorted descending = largest first, ties in group order
    for (uint32_t g = 0; g < job.num_groups; g++) {
      uint64_t bytes = 0;
      for (uint32_t p = 0; p < num_passes; p++) bytes += sizes[(size_t)p * job.num_groups + g];
      key[g] = (std::min<uint64_t>(bytes, 0xFFFFFFFFu) << 32) | (uint32_t)~g;
    }
    std::sort(key.begin(), key.end(), std::greater<uint64_t>());
    job.order.resize(job.num_groups);
    for (uint32_t t = 0; t < job.num_groups; t++) job.order[t] = ~(uint32_t)key[t];
  }
  const bool verbose = jxlhip_env::Get().codestream_verbose.load(std::memory_order_relaxed);
  if (verbose) {
    job.timeline.assign(3 * (size_t)job.num_groups, 0.0f);
    job.t0 = std::chrono::steady_clock::now();
  }
  if (runner) {
    if (runner(runner_opaque, &job, GroupsInit, GroupsFunc, 0, job.num_groups) != 0)
      return Fail(c, JXLHIP_ERR_STATE, "parallel runner failed");
    if (verbose) GroupsTimelineReport(job);
  } else {
    GroupsInit(&job, 1);
    for (uint32_t g = 0; g < job.num_groups; g++) GroupsFunc(&job, g, 0);
  }
  for (SparseBatch& b : job.batch) {  // what the threads still hold
    const int rc = SparseFlush(c, &b);
    if (rc != JXLHIP_OK) {
      int expected = JXLHIP_OK;
      job.status.compare_exchange_strong(expected, rc);
    }
  }
  return job.status.load();
}

int jxlhip_ac_group_decode_submit(jxlhip_ctx* c, const jxlhip_ac_pass* pass, uint32_t group_idx,
                                  const uint8_t* ac_strategy, const int32_t* raw_quant,
                                  const uint8_t* quant_dc, const uint8_t* data, size_t size,
                                  size_t* bit_pos) {
  return jxlhip_ac_group_decode_submit_passes(c, 1, &pass, nullptr, group_idx, ac_strategy, raw_quant,
                                              quant_dc, &data, &size, bit_pos);
}

// ---- decode -------------------------------------------------------------------
namespace {

// k_prepare + the transform kernels for group rows [g0, g1) of the stripe,
// using counter slot `band`.
// fused: 0 = two-phase, 1 = the whole frame through the fused kernel, 2 = a STRIPE through it (the DCT8 cells of
// the stripe's first / last block row are decoded into the planes as well: they are the halo rows its neighbours pull)
int LaunchBlocksBand(jxlhip_ctx* c, uint32_t g0, uint32_t g1, int band, int fused = 0,
                     const FilterParams* emit = nullptr, int zero_band = -1) {
  hipStream_t st = c->stream;
  DevFrame f = c->f;
  f.band_g0 = g0;
  f.band_g1 = g1;
  f.fused = (uint32_t)fused;
  f.cell_info = c->cell_info;
  {
    constexpr uint32_t kOthers32 = (1u << 8) | (1u << 9) | (1u << 10) | (1u << 11);  // 32x8 .. 16x32
    const bool lone32 = (f.used_acs & (1u << 5)) && !(f.used_acs & kOthers32);
    f.mfma32 = (c->mfma > 0 || (c->mfma < 0 && lone32)) ? c->tables + 1600 : nullptr;
    // DCT16X16: the same rule against the 16-point row-per-lane family (16x8, 8x16), only when no 32-point class
    // pulls the merged launch in anyway, and on frames of 16 Mpx and more (measured, all-DCT16X16 frames: 8K blocks
    // 132 -> 118 us, 16x16 + 32x32 176 -> 161 us; 4K 33.6 -> 37.6 us: the butterflies stay; on the mixed c3 frame a
    // launch of its own costs 97 -> 117 us, like DCT32X32)
    constexpr uint32_t kOthers16 = (1u << 6) | (1u << 7);
    const bool lone16 = (f.used_acs & (1u << 4)) && !(f.used_acs & (kOthers16 | kOthers32)) &&
                        (!(f.used_acs & (1u << 5)) || f.mfma32) && (uint64_t)f.xsize * f.ysize >= (16u << 20);
    f.mfma16 = (c->mfma > 0 || (c->mfma < 0 && lone16)) ? c->tables + 1600 + 2048 : nullptr;
  }
  f.zero_counts = zero_band >= 0 ? c->counts + (size_t)zero_band * kCountStride : nullptr;
  if (fused == 2)  // a stripe: every cell "from the planes" until k_prepare says otherwise (whole frames: k_prepare writes every cell)
    HIPCHK(c, hipMemsetAsync(c->cell_info, 0xFF, sizeof(uint2) * (size_t)f.xsb * f.ysb, st));
  WorkLists wl = c->wl;
  wl.count = c->counts + (size_t)band * kCountStride;
  const uint32_t cells = f.xsg * (g1 - g0) * 1024u;
  ProfBegin(c);
  LaunchPrepare(f, wl, c->p.lf.epf_iters > 0, c->p.lf.epf_quant_mul, c->lut, st);
  ProfMark(c, JXLHIP_KERNEL_PREPARE);
  if (c->nblock_streams > 1) {
    // fork: the class kernels wait for k_prepare, run side by side, and the
    // main stream joins them all before anything that reads the planes
    HIPCHK(c, hipEventRecord(c->fork_ev, st));
    for (int i = 0; i < c->nblock_streams; i++)
      HIPCHK(c, hipStreamWaitEvent(c->bstreams[i], c->fork_ev, 0));
    LaunchBlocks(f, wl, cells, c->tables, c->tables + 512, c->bstreams, c->nblock_streams, emit);
    for (int i = 0; i < c->nblock_streams; i++) {
      HIPCHK(c, hipEventRecord(c->bev[i], c->bstreams[i]));
      HIPCHK(c, hipStreamWaitEvent(st, c->bev[i], 0));
    }
  } else {
    LaunchBlocks(f, wl, cells, c->tables, c->tables + 512, &st, 1, emit);
  }
  ProfMark(c, JXLHIP_KERNEL_BLOCKS);
  ProfEnd(c);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

// phase 2 for pixel rows [fy0, fy1) of the stripe
int LaunchFiltersRows(jxlhip_ctx* c, const FilterParams& fp, uint32_t fy0, uint32_t fy1, bool fused = false) {
  DevFrame f = c->f;
  f.fy0 = fy0;
  f.fy1 = fy1;
  f.fused = fused ? 1u : 0u;
  f.cell_info = c->cell_info;
  ProfBegin(c);
  if (fused && c->p.lf.epf_iters == 3) {
    // EPF0 from the producer's slab into the second plane set (k_fused_pc0), EPF1 + EPF2 + output from there
    float* dst[3];
    const size_t plane_floats = (size_t)f.plane_tile_rows * f.tile_stride * 64;
    for (int ch = 0; ch < 3; ch++) dst[ch] = c->planes2 + ch * plane_floats;
    if (!LaunchFusedEpf0(f, fp, (int)c->p.lf.gab, dst, c->stream))
      return Fail(c, JXLHIP_ERR_STATE, "fused EPF0 kernel refused a frame FusedEpf0Supported accepted");
    ProfMark(c, JXLHIP_KERNEL_EPF0);
    DevFrame f2 = f;
    for (int ch = 0; ch < 3; ch++) f2.xyb[ch] = dst[ch];
    f2.linear_stride = f.tile_stride * 32u;
    if (!LaunchFiltersFast(f2, fp, 0, 2, (int)c->p.output_kind, c->stream))
      return Fail(c, JXLHIP_ERR_STATE, "EPF1 + EPF2 march refused a frame the fused EPF0 march accepted");
    ProfMark(c, JXLHIP_KERNEL_FILTERS);
    ProfEnd(c);
    HIPCHK(c, hipGetLastError());
    return JXLHIP_OK;
  }
  if (fused) {
    if (!LaunchFused(f, fp, (int)c->p.lf.gab, (int)c->p.lf.epf_iters, (int)c->p.output_kind, c->stream))
      return Fail(c, JXLHIP_ERR_STATE, "fused kernel refused a frame FusedSupported accepted");
    ProfMark(c, JXLHIP_KERNEL_FUSED);
    ProfEnd(c);
    HIPCHK(c, hipGetLastError());
    return JXLHIP_OK;
  }
  bool fast = false;
  if (!c->generic_filters && fy1 > fy0 && c->p.lf.epf_iters == 3 && c->planes2) {
    // EPF0 into the second plane set, EPF1 + EPF2 + output from there
    float* dst[3];
    const size_t plane_floats = (size_t)f.plane_tile_rows * f.tile_stride * 64;
    for (int ch = 0; ch < 3; ch++) dst[ch] = c->planes2 + ch * plane_floats;
    if (LaunchEpf0(f, fp, (int)c->p.lf.gab, dst, c->stream)) {
      ProfMark(c, JXLHIP_KERNEL_EPF0);
      DevFrame f2 = f;
      for (int ch = 0; ch < 3; ch++) f2.xyb[ch] = dst[ch];
      f2.linear_stride = f.tile_stride * 32u;
      fast = LaunchFiltersFast(f2, fp, 0, 2, (int)c->p.output_kind, c->stream);
      if (!fast) return Fail(c, JXLHIP_ERR_STATE, "EPF1 + EPF2 march refused a frame the EPF0 march accepted");
    }
  } else if (!c->generic_filters && fy1 > fy0) {
    fast = LaunchFiltersFast(f, fp, (int)c->p.lf.gab, (int)c->p.lf.epf_iters, (int)c->p.output_kind, c->stream);
  }
  if (!fast && LaunchFilters(f, fp, (int)c->p.lf.gab, (int)c->p.lf.epf_iters,
                             (int)c->p.output_kind, c->stream) != 0)
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "unsupported filter configuration");
  ProfMark(c, JXLHIP_KERNEL_FILTERS);
  ProfEnd(c);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

// Does this frame (or stripe of it) go through the fused kernel (kernels_fused.hip)?  auto: with a filter, frames
// of 12 Mpx and more (see jxlhip_ctx::fuse); without one the fused wave has no halo rows to pay for and wins at 4K as
// well (95.8 vs 83.4 Gpx/s); never when the caller's used_acs says the frame has no DCT8 block -- then the slab is
// only a detour (configs[4]: 76.1 vs 79.6 Gpx/s).  Packed outputs: the two-phase filter kernel has the formats djxl
// writes most (8-bit sRGB RGB / RGBA, 16-bit sRGB RGB) fixed at compile time, the fused kernel only the general
// per-sample format path -- measured at 8K d1.0: 0.42 ms two-phase against 0.74 - 0.82 ms fused
// (profiles/r03_packed_paths.txt).
bool WantFused(const jxlhip_ctx* c) {
  const DevFrame& f = c->f;
  // alone a context fuses from 12 Mpx; with several frames in flight on the device (jxlhip_set_concurrency_hint) the
  // step is bound by HBM traffic and the fused path, which moves less, pays from 6 Mpx (profiles/r04_path_choice.txt)
  const uint64_t min_px = c->concurrency > 1 ? (6ull << 20) : (12ull << 20);
  const bool big = (uint64_t)f.xsize * f.ysize >= min_px || (c->p.lf.gab == 0 && c->p.lf.epf_iters == 0);
  const bool has_dct8 = f.used_acs == 0 || (f.used_acs & 1u);
  bool packed_fixed = false;
  if (c->p.output_kind == JXLHIP_OUT_PACKED) {
    const jxlhip_output_format& o = c->p.out_format;
    packed_fixed = FastFixedFormat(o);
  }
  if (c->p.lf.epf_iters == 3) {
    // three EPF iterations: EPF0 marches from the fused producer's slab (k_fused_pc0), the rest as before; whole frames
    // only, and only with several frames in flight on the device (jxlhip_set_concurrency_hint): the producer / consumer
    // form has half the marching waves of k_epf0 per CU and is slower on its own (8K d1.0: the EPF0 launch 0.33 against
    // 0.23 ms, the step 0.651 against 0.623 ms of kernels), but it moves the DCT8 share's 24 bytes per pixel less, and a
    // device kept busy by other frames is bound by traffic: 64.8 -> 67.0 Gpx/s with three in flight
    // (profiles/r04_epf3_fused.txt)
    const bool many = c->concurrency > 1 && (uint64_t)f.xsize * f.ysize >= (6ull << 20);
    return (c->fuse > 0 || (c->fuse < 0 && many && has_dct8)) && !c->generic_filters && c->band_rows == 0 && c->planes2 &&
           f.group_y0 == 0 && f.group_rows == f.ysg && FusedEpf0Supported(f, (int)c->p.lf.gab);
  }
  return (c->fuse > 0 || (c->fuse < 0 && big && has_dct8 && !packed_fixed)) && !c->generic_filters && c->band_rows == 0 &&
         FusedSupported(f, (int)c->p.lf.gab, (int)c->p.lf.epf_iters, (int)c->p.output_kind);
}

int BeginDecode(jxlhip_ctx* c, uint32_t nbands, uint32_t first_block = 0) {
  if (!c->have_frame || !c->have_inputs)
    return Fail(c, JXLHIP_ERR_STATE, "decode needs frame_begin + inputs");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  {
    std::lock_guard<std::mutex> lock(c->pool_mu);
    for (int i = 0; i < kPoolStreams; i++) {
      if (!c->pool_dirty[i]) continue;
      HIPCHK(c, hipEventRecord(c->pool_ev[i], c->pool[i]));
      HIPCHK(c, hipStreamWaitEvent(st, c->pool_ev[i], 0));
      c->pool_dirty[i] = false;
    }
  }
  if (c->sp_any.load() && c->f.coeffs[0] == c->up_coeffs[0]) {
    // the groups that came up as non-zero lists: their offset table, then zero + scatter into the dense upload
    // buffer, behind the uploads.  (Idempotent: a second decode of the same frame repeats it.)
    const size_t ng = (size_t)c->f.xsg * c->f.ysg;
    const int par = (int)(c->frame_serial & 1u);
    int rc;
    if ((rc = Grow(c, &c->sp_off_dev, &c->sp_off_dev_items, ng))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->sp_off_dev, c->sp_off_host[par], ng * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->sp_off_ev[par], st));
    c->sp_off_pending[par] = true;
    LaunchExpandSparse(c->sp_dev, c->sp_off_dev, (int16_t*)c->up_coeffs[0], c->f.group_y0 * c->f.xsg, c->f.group_rows * c->f.xsg, st);
  }
  if (nbands > (uint32_t)kMaxBands) nbands = kMaxBands;
  if (nbands) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    uint32_t* blocks = c->counts + (size_t)first_block * kCountStride;
    if (cap == hipStreamCaptureStatusActive) {  // (see LaunchZeroU32: no memset node at the root of a frame graph)
      LaunchZeroU32(blocks, (uint32_t)(kCountStride * nbands), st);
      HIPCHK(c, hipGetLastError());
    } else {
      HIPCHK(c, hipMemsetAsync(blocks, 0, sizeof(uint32_t) * kCountStride * nbands, st));
    }
    if (first_block == 0) c->counts_clean[0] = c->counts_clean[1] = false;  // used by the bands that follow
  }
  return JXLHIP_OK;
}

int CheckOutArgs(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride) {
  const DevFrame& f = c->f;
  if (!out) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "null output");
  if (c->p.output_kind == JXLHIP_OUT_LINEAR_RGB_F32) {
    if (out_stride < (size_t)f.xsize * 12 || (out_stride & 3))
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "RGB row stride %zu too small", out_stride);
  } else if (c->p.output_kind == JXLHIP_OUT_PACKED) {
    const jxlhip_output_format& o = c->p.out_format;
    const size_t ssz = o.sample_type == JXLHIP_SAMPLE_U8 ? 1 : (o.sample_type == JXLHIP_SAMPLE_F32 ? 4 : 2);
    if (out_stride < (size_t)f.xsize * o.num_channels * ssz || (out_stride % ssz) ||
        ((uintptr_t)out % ssz))
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "packed row stride %zu / alignment invalid", out_stride);
  } else if (out_stride < f.xsize ||
             out_plane_stride < out_stride * (size_t)(f.y1 - f.y0 - 1) + f.xsize) {
    return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "XYB strides too small");
  }
  return JXLHIP_OK;
}

}  // namespace

int jxlhip_decode_blocks(jxlhip_ctx* c) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  int rc = BeginDecode(c, 1);
  if (rc) return rc;
  // A STRIPE of a frame (what jxlhip_create_multi / libjxl_amd.stripes run per device) takes the fused kernel
  // by the whole-frame rule: its DCT8 blocks are decoded inside the filter march, except that those of its
  // first / last block row ALSO reach the planes -- the halo rows the neighbours pull with jxlhip_halo_export.
  // A whole frame through the split calls stays two-phase (the taps read the planes).
  const bool stripe = c->f.group_y0 != 0 || c->f.group_rows != c->f.ysg;
  c->blocks_fused = stripe && WantFused(c);
  if (c->blocks_fused && (rc = Grow(c, &c->cell_info, &c->cell_info_items, (size_t)c->f.xsb * c->f.ysb))) return rc;
  rc = LaunchBlocksBand(c, c->f.group_y0, c->f.group_y0 + c->f.group_rows, 0, c->blocks_fused ? 2 : 0);
  if (rc) return rc;
  c->blocks_done = true;
  return JXLHIP_OK;
}

int jxlhip_halo_rows(const jxlhip_ctx* c) {
  if (!c || !c->have_frame) return JXLHIP_ERR_STATE;
  return (int)c->f.halo;
}

static int HaloCopy(jxlhip_ctx* c, int which, float* dev, bool to_dense) {
  if (!c || !dev || which < 0 || which > 1) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "halo copy before frame_begin");
  const DevFrame& f = c->f;
  const uint32_t stripe_rows = f.y1 - f.y0;
  if (f.halo > stripe_rows)
    return Fail(c, JXLHIP_ERR_STATE, "stripe of %u rows is shorter than the %u-row halo",
                stripe_rows, f.halo);
  if (f.halo == 0) return JXLHIP_OK;
  int y_first;
  if (to_dense) y_first = which == 0 ? (int)f.y0 : (int)(f.y1 - f.halo);
  else y_first = which == 0 ? (int)f.y0 - (int)f.halo : (int)f.y1;
  if (y_first < 0 || y_first + (int)f.halo > (int)f.ysize)
    return Fail(c, JXLHIP_ERR_STATE, "no neighbouring stripe on that side");
  HIPCHK(c, hipSetDevice(c->device));
  LaunchRowsCopy(f, dev, y_first, (int)f.halo, (int)f.xsize, f.xsize,
                 (size_t)f.halo * f.xsize, 3, to_dense, c->stream);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

int jxlhip_halo_export(jxlhip_ctx* c, int which, float* dev) {
  if (c && !c->blocks_done) return Fail(c, JXLHIP_ERR_STATE, "halo_export before decode_blocks");
  return HaloCopy(c, which, dev, true);
}

int jxlhip_halo_import(jxlhip_ctx* c, int which, const float* dev) {
  return HaloCopy(c, which, const_cast<float*>(dev), false);
}

int jxlhip_decode_filters(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  return jxlhip_decode_filters_rows(c, out, out_stride, out_plane_stride, c->f.y0, c->f.y1);
}

int jxlhip_decode_filters_rows(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride, uint32_t y_begin,
                               uint32_t y_end) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  if (!c->blocks_done) return Fail(c, JXLHIP_ERR_STATE, "decode_filters before decode_blocks");
  if (c->p.undo_orientation > 1) return Fail(c, JXLHIP_ERR_UNSUPPORTED, "undo_orientation with the split calls");
  int rc = CheckOutArgs(c, out, out_stride, out_plane_stride);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  FilterParams fp = c->fp;
  fp.out = out;
  fp.out_stride = out_stride;
  fp.out_plane_stride = out_plane_stride;
  const bool whole = y_begin == c->f.y0 && y_end == c->f.y1;
  if (!whole) {
    if (y_begin == y_end && y_begin >= c->f.y0 && y_end <= c->f.y1) return JXLHIP_OK;  // (an edge stripe has no boundary rows on its outer side)
    if (y_begin < c->f.y0 || y_end > c->f.y1 || y_begin > y_end ||
        ((y_begin & 7u) && y_begin != c->f.y0) || ((y_end & 7u) && y_end != c->f.y1))
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "rows [%u, %u) of a stripe [%u, %u): block-row multiples inside it only", y_begin,
                  y_end, c->f.y0, c->f.y1);
    if (c->p.lf.epf_iters == 3) return Fail(c, JXLHIP_ERR_UNSUPPORTED, "row ranges with epf_iters = 3 (two marches over a second plane set)");
  }
  return LaunchFiltersRows(c, fp, y_begin, y_end, c->blocks_fused);
}

// ---- one stripe step in three calls (round 5) ----------------------------------------------------
// What a rank of the striped decode (libjxl_amd/stripes.py: one process per GPU) enqueues around its halo exchange was
// seven to nine C calls per frame -- phase 1, two exports, up to three row ranges of phase 2, two imports -- each
// through the host language's FFI: on a 16K frame over 8 GPUs a rank's kernels take ~170 us, and the host must not take
// as long to enqueue them.  jxlhip_stripe_begin = phase 1 + both exports; [the caller posts its sends / receives, then
// jxlhip_decode_filters_rows for the interior]; jxlhip_stripe_finish = both imports + the boundary block rows.
// send_* / recv_* = dense [3][halo][xsize] device buffers, nullptr = no neighbour on that side.
int jxlhip_stripe_begin(jxlhip_ctx* c, float* send_up, float* send_down) {
  int rc = jxlhip_decode_blocks(c);
  if (rc) return rc;
  if (send_up && (rc = jxlhip_halo_export(c, 0, send_up))) return rc;
  if (send_down && (rc = jxlhip_halo_export(c, 1, send_down))) return rc;
  return JXLHIP_OK;
}

// y_interior_begin / _end: the rows jxlhip_decode_filters_rows has filtered already between the two calls (equal:
// none -- everything is filtered here, behind the imports)
int jxlhip_stripe_finish(jxlhip_ctx* c, const float* recv_up, const float* recv_down, void* out, size_t out_stride,
                         size_t out_plane_stride, uint32_t y_interior_begin, uint32_t y_interior_end) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  int rc;
  if (recv_up && (rc = jxlhip_halo_import(c, 0, recv_up))) return rc;
  if (recv_down && (rc = jxlhip_halo_import(c, 1, recv_down))) return rc;
  if (y_interior_begin >= y_interior_end) return jxlhip_decode_filters(c, out, out_stride, out_plane_stride);
  if ((rc = jxlhip_decode_filters_rows(c, out, out_stride, out_plane_stride, c->f.y0, y_interior_begin))) return rc;
  return jxlhip_decode_filters_rows(c, out, out_stride, out_plane_stride, y_interior_end, c->f.y1);
}

// Both phases.  With JXLHIP_BAND_ROWS = n > 0 the stripe is walked in bands of n
// group rows -- blocks(b) then filters(b-1) -- which was meant to keep a band's
// XYB planes in the 256 MB Infinity Cache; measured on MI355X it only loses
// time, so the default is one band = the whole stripe.
static int DecodeFrameCoded(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride);

// bytes of one interleaved output pixel (0: planar XYB)
static size_t OutPixelBytes(const jxlhip_ctx* c) {
  if (c->p.output_kind == JXLHIP_OUT_LINEAR_RGB_F32) return 12;
  if (c->p.output_kind != JXLHIP_OUT_PACKED) return 0;
  const jxlhip_output_format& o = c->p.out_format;
  return (size_t)o.num_channels * (o.sample_type == JXLHIP_SAMPLE_U8 ? 1 : (o.sample_type == JXLHIP_SAMPLE_F32 ? 4 : 2));
}

int jxlhip_decode_frame(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) return out ? MultiDecodeFrame(c, out, nullptr, out_stride, out_plane_stride) : JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "decode needs frame_begin + inputs");
  if (c->p.undo_orientation <= 1) return DecodeFrameCoded(c, out, out_stride, out_plane_stride);
  // undo_orientation: coded orientation into a staging frame, k_orient into the caller's buffer
  const DevFrame& f = c->f;
  const size_t bpp = OutPixelBytes(c);
  if (!out || bpp == 0) return Fail(c, JXLHIP_ERR_UNSUPPORTED, "undo_orientation needs an interleaved output");
  if (f.group_y0 != 0 || f.group_rows != f.ysg) return Fail(c, JXLHIP_ERR_UNSUPPORTED, "undo_orientation with stripes");
  const bool transposed = c->p.undo_orientation >= 5;
  const size_t need = (size_t)(transposed ? f.ysize : f.xsize) * bpp;
  if (out_stride < need) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "row stride %zu too small for the oriented frame", out_stride);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t row = ((size_t)f.xsize * bpp + 255) & ~(size_t)255;
  int rc;
  if ((rc = Grow(c, &c->orient_dev, &c->orient_bytes, (size_t)f.ysize * row))) return rc;
  if ((rc = DecodeFrameCoded(c, c->orient_dev, row, 0))) return rc;
  if (!LaunchOrient(c->orient_dev, row, f.xsize, f.ysize, (uint32_t)bpp, c->p.undo_orientation, out, out_stride, c->stream))
    return Fail(c, JXLHIP_ERR_UNSUPPORTED, "undo_orientation %u with %zu-byte pixels", c->p.undo_orientation, bpp);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

static int DecodeFrameCoded(jxlhip_ctx* c, void* out, size_t out_stride, size_t out_plane_stride) {
  const DevFrame& f = c->f;
  c->blocks_fused = false;
  uint32_t br = c->band_rows ? (uint32_t)c->band_rows : f.group_rows;
  while ((f.group_rows + br - 1) / br > (uint32_t)kMaxBands) br++;
  // one band (the default): the counter blocks 0 / 1 alternate and k_prepare zeroes the next frame's -- no memset launch
  const uint32_t nbands = (f.group_rows + br - 1) / br;
  // Under stream capture -- the caller records the frame's launches into a hipGraph and replays it (bench.py's
  // `graph_replay`: the command processor's ~5-8 us per dependent launch are paid once per graph instead) -- every
  // replay must find the SAME counter blocks zeroed by a node of the graph itself, and must not touch a block the direct
  // calls keep a "clean" flag for: captured frames use the blocks from kCaptureBase on (see there).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(c->stream, &cap);
  const bool capturing = cap == hipStreamCaptureStatusActive;
  const bool one_band = nbands == 1 && !capturing;
  const int block0 = capturing ? kCaptureBase : 0;
  const int slot = one_band ? c->counts_slot : block0;
  int rc = BeginDecode(c, one_band ? 0 : nbands, (uint32_t)block0);
  if (rc) return rc;
  if (one_band && !c->counts_clean[slot])
    HIPCHK(c, hipMemsetAsync(c->counts + (size_t)slot * kCountStride, 0, sizeof(uint32_t) * kCountStride, c->stream));
  // From here on block `slot` is in use: whatever happens below (a failed launch after k_prepare ran), it must not be
  // taken for clean by the next frame.  Block slot ^ 1 becomes clean only when the launches that zero it succeeded.
  if (one_band) c->counts_clean[slot] = false;
  auto rotate = [&]() {  // after a successful LaunchBlocksBand(.., slot, .., slot ^ 1)
    c->counts_clean[slot ^ 1] = true;
    c->counts_slot = slot ^ 1;
  };
  rc = CheckOutArgs(c, out, out_stride, out_plane_stride);
  if (rc) return rc;
  FilterParams fp = c->fp;
  fp.out = out;
  fp.out_stride = out_stride;
  fp.out_plane_stride = out_plane_stride;
  // A frame of DCT32X32 varblocks only, no loop filter, linear float RGB out (BASELINE configs[4]): the matrix-core
  // class kernel applies the opsin inverse and writes the pixels itself (kernels_mfma.hip, EMIT) -- no XYB planes,
  // no second kernel.  used_acs is the caller's promise; k_prepare reports any other strategy it meets.
  if (c->mfma != 0 && f.used_acs == (1u << 5) && c->p.lf.gab == 0 && c->p.lf.epf_iters == 0 &&
      c->p.output_kind == JXLHIP_OUT_LINEAR_RGB_F32 && !c->generic_filters && c->band_rows == 0) {
    rc = LaunchBlocksBand(c, f.group_y0, f.group_y0 + f.group_rows, slot, 0, &fp, one_band ? slot ^ 1 : -1);
    if (!rc && one_band) rotate();
    c->blocks_done = false;  // nothing in the planes
    return rc;
  }
  // Whole frame on this context, one band: the fused kernel decodes the DCT8 blocks inside the filter
  // march (kernels_fused.hip).  The split calls (jxlhip_decode_blocks / _filters) stay two-phase: a
  // stripe's halo rows must exist in the planes for its neighbours.
  // auto: with a filter, frames of 12 Mpx and more (see jxlhip_ctx::fuse); without one the fused wave has no
  // halo rows to pay for and wins at 4K as well (95.8 vs 83.4 Gpx/s); never when the caller's used_acs says the
  // frame has no DCT8 block -- then the slab is only a detour (configs[4]: 76.1 vs 79.6 Gpx/s)
  if (WantFused(c) && f.group_y0 == 0 && f.group_rows == f.ysg) {
    if ((rc = Grow(c, &c->cell_info, &c->cell_info_items, (size_t)f.xsb * f.ysb))) return rc;
    rc = LaunchBlocksBand(c, f.group_y0, f.group_y0 + f.group_rows, slot, 1, nullptr, one_band ? slot ^ 1 : -1);
    if (rc) return rc;
    if (one_band) rotate();
    c->blocks_done = false;  // the planes do not hold the whole frame
    return LaunchFiltersRows(c, fp, f.y0, f.y1, true);
  }
  const uint32_t g_end = f.group_y0 + f.group_rows;
  uint32_t prev_y0 = f.y0;
  int band = 0;
  for (uint32_t g0 = f.group_y0; g0 < g_end; g0 += br, band++) {
    const uint32_t g1 = g0 + br < g_end ? g0 + br : g_end;
    rc = one_band ? LaunchBlocksBand(c, g0, g1, slot, 0, nullptr, slot ^ 1) : LaunchBlocksBand(c, g0, g1, block0 + band);
    if (rc) return rc;
    if (one_band) rotate();
    if (g0 > f.group_y0) {  // rows of the previous band: its lower halo now exists
      rc = LaunchFiltersRows(c, fp, prev_y0, g0 * 256);
      if (rc) return rc;
      prev_y0 = g0 * 256;
    }
  }
  c->blocks_done = true;
  return LaunchFiltersRows(c, fp, prev_y0, f.y1);
}

// The boundary handing over a HOST buffer (what JxlDecoderSetImageOutBuffer gives libjxl): both phases
// into a context-owned device frame, one strided device-to-host copy, synchronised.
int jxlhip_decode_frame_host(jxlhip_ctx* c, void* host_out, size_t out_stride, size_t out_plane_stride) {
  if (!c || !host_out) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) {
    const int rc = MultiDecodeFrame(c, nullptr, host_out, out_stride, out_plane_stride);
    return rc ? rc : MultiSync(c);
  }
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "decode needs frame_begin + inputs");
  const DevFrame& f = c->f;
  const bool transposed = c->p.undo_orientation >= 5;  // the oriented frame is ysize wide, xsize high
  const size_t rows = transposed ? f.xsize : f.y1 - f.y0;
  const size_t cols = transposed ? f.ysize : f.xsize;
  size_t row_bytes;
  if (c->p.output_kind == JXLHIP_OUT_LINEAR_RGB_F32) row_bytes = cols * 12;
  else if (c->p.output_kind == JXLHIP_OUT_PACKED) row_bytes = cols * OutPixelBytes(c);
  else row_bytes = cols * 4;
  const bool planar = c->p.output_kind == JXLHIP_OUT_XYB_PLANAR;
  const size_t host_row = planar ? out_stride * 4 : out_stride;
  if (host_row < row_bytes) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "host row stride %zu too small", out_stride);
  const size_t dev_row = (row_bytes + 255) & ~(size_t)255;
  const size_t planes = planar ? 3 : 1;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = Grow(c, &c->host_frame_dev, &c->host_frame_bytes, planes * rows * dev_row))) return rc;
  rc = planar ? jxlhip_decode_frame(c, c->host_frame_dev, dev_row / 4, rows * dev_row / 4)
              : jxlhip_decode_frame(c, c->host_frame_dev, dev_row, 0);
  if (rc) return rc;
  for (size_t pl = 0; pl < planes; pl++)
    HIPCHK(c, hipMemcpy2DAsync((char*)host_out + pl * out_plane_stride * 4, host_row, c->host_frame_dev + pl * rows * dev_row,
                               dev_row, row_bytes, rows, hipMemcpyDeviceToHost, c->stream));
  return jxlhip_sync(c);
}

int jxlhip_decode_frame_pinned(jxlhip_ctx* c, const void** host_frame, size_t* stride) {
  if (!c || !host_frame || !stride) return JXLHIP_ERR_INVALID_ARGUMENT;
  const jxlhip_frame_params& p = c->p;
  if (p.output_kind == JXLHIP_OUT_XYB_PLANAR) return Fail(c, JXLHIP_ERR_UNSUPPORTED, "pinned frames are interleaved outputs");
  const bool transposed = p.undo_orientation >= 5;
  const size_t rows = transposed ? p.xsize : p.ysize;
  const size_t cols = transposed ? p.ysize : p.xsize;
  const size_t row_bytes = cols * (p.output_kind == JXLHIP_OUT_PACKED ? OutPixelBytes(c) : 12);
  const size_t pitch = (row_bytes + 63) & ~(size_t)63;
  if (rows * pitch > c->pinned_frame_bytes) {
    if (c->pinned_frame) {
      int rc0 = jxlhip_sync(c);  // nothing may still be writing the old frame
      if (rc0) return rc0;
      StageFree(c, c->pinned_frame);
      c->pinned_frame = nullptr;
      c->pinned_frame_bytes = 0;
    }
    if (StageAlloc(c, &c->pinned_frame, rows * pitch)) return Fail(c, JXLHIP_ERR_OUT_OF_MEMORY, "pinned frame of %zu bytes", rows * pitch);
    c->pinned_frame_bytes = rows * pitch;
  }
  const int rc = jxlhip_decode_frame_host(c, c->pinned_frame, pitch, 0);
  if (rc) return rc;
  *host_frame = c->pinned_frame;
  *stride = pitch;
  return JXLHIP_OK;
}

int jxlhip_sync(jxlhip_ctx* c) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!c->children.empty()) return MultiSync(c);
  HIPCHK(c, hipSetDevice(c->device));
  int32_t flag[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(flag, c->error_flag, sizeof(flag), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (flag[0] || flag[1]) {
    HIPCHK(c, hipMemsetAsync(c->error_flag, 0, sizeof(flag), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return Fail(c, JXLHIP_ERR_BAD_STREAM,
                flag[0] ? "AC strategy map violates the group/stream constraints "
                          "(dec_modular.cc:539-549, dec_group.cc:359)"
                        : "dequant table weight out of range (quant_weights.cc:329-339)");
  }
  return JXLHIP_OK;
}

// ---- taps ------------------------------------------------------------------------
int jxlhip_export_xyb(jxlhip_ctx* c, float* const dst[3], size_t dst_stride) {
  if (!c || !dst || !dst[0] || !dst[1] || !dst[2]) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "no frame");
  if (!c->blocks_done || c->blocks_fused)
    return Fail(c, JXLHIP_ERR_STATE, "export_xyb needs a two-phase jxlhip_decode_blocks (a fused decode -- jxlhip_decode_frame, or "
                                     "a stripe of a frame of 12 Mpx and more -- leaves DCT8 blocks out of the planes; JXLHIP_FUSE=0)");
  const DevFrame& f = c->f;
  const int rows = (int)(f.plane_tile_rows - 2) * 8;
  if (dst_stride < (size_t)f.xsb * 8) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "stride too small");
  HIPCHK(c, hipSetDevice(c->device));
  // the three destination planes may be separate allocations: one launch each
  for (int ch = 0; ch < 3; ch++) {
    DevFrame g = f;
    g.xyb[0] = f.xyb[ch];
    LaunchRowsCopy(g, dst[ch], (int)f.y0, rows, (int)f.xsb * 8, dst_stride, 0, 1, true, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

int jxlhip_get_sigma(jxlhip_ctx* c, float** inv_sigma, size_t* row_stride) {
  if (!c || !inv_sigma || !row_stride) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "no frame");
  *inv_sigma = c->f.inv_sigma;
  *row_stride = c->f.xsb;
  return JXLHIP_OK;
}

int jxlhip_set_concurrency_hint(jxlhip_ctx* c, int frames_in_flight) {
  if (!c || frames_in_flight < 1) return JXLHIP_ERR_INVALID_ARGUMENT;
  c->concurrency = frames_in_flight;
  for (jxlhip_ctx* k : c->children) k->concurrency = frames_in_flight;
  return JXLHIP_OK;
}

int jxlhip_profile_enable(jxlhip_ctx* c, int enable) {
  if (!c) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  c->profiling = enable != 0;
  return JXLHIP_OK;
}

int jxlhip_profile_read(jxlhip_ctx* c, float ms[JXLHIP_KERNEL_COUNT],
                        uint32_t launches[JXLHIP_KERNEL_COUNT]) {
  if (!c || !ms || !launches) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < JXLHIP_KERNEL_COUNT; i++) {
    ms[i] = 0;
    launches[i] = 0;
  }
  for (size_t i = 0; i + 1 < c->marks.size(); i++) {
    const int slot = c->marks[i].slot_after;
    if (slot >= 0 && slot < JXLHIP_KERNEL_COUNT) {
      float t = 0;
      if (hipEventElapsedTime(&t, c->marks[i].ev, c->marks[i + 1].ev) == hipSuccess) {
        ms[slot] += t;
        launches[slot]++;
      }
    }
  }
  for (auto& m : c->marks) (void)hipEventDestroy(m.ev);
  c->marks.clear();
  return JXLHIP_OK;
}

// ---- a5 / a8 ------------------------------------------------------------------------
// Library encodings -> the parameters they stand for (DequantMatrices::Library,
// quant_weights.cc:532-1188; data in format_constants.inc).  The AFV library entry
// takes its 4x8 and 4x4 band parameters from the DCT4X8 and DCT4X4 entries.
static void ResolveLibrary(int kind, jxlhip_quant_encoding* e) {
  static const uint32_t kModeOfLib[6] = {JXLHIP_QUANT_DCT,  JXLHIP_QUANT_ID,     JXLHIP_QUANT_DCT2,
                                         JXLHIP_QUANT_DCT4, JXLHIP_QUANT_DCT4X8, JXLHIP_QUANT_AFV};
  const QuantLibEntry& l = kQuantLib[kind];
  memset(e, 0, sizeof(*e));
  e->mode = kModeOfLib[l.mode];
  const QuantLibEntry& b = l.mode == 5 ? kQuantLib[9] : l;
  e->num_bands = (uint32_t)b.nb;
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < 8; i++) e->bands[c][i] = b.bands[c][i];
    for (int i = 0; i < 9; i++) e->weights[c][i] = l.w[c][i];
  }
  if (l.mode == 5) {
    e->num_bands_afv_4x4 = (uint32_t)kQuantLib[3].nb;
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < 8; i++) e->bands_afv_4x4[c][i] = kQuantLib[3].bands[c][i];
  }
}

int jxlhip_dequant_tables(jxlhip_ctx* c, const jxlhip_quant_encoding* enc, float* table_dev) {
  if (!c || !table_dev) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  static const uint32_t kSingleBlockKinds = 0x60F;  // kinds whose matrix is one 8x8 block
  for (int k = 0; k < JXLHIP_NUM_QUANT_TABLES; k++) {
    jxlhip_quant_encoding* e = &c->quant_enc_host[k];
    if (!enc || enc[k].mode == JXLHIP_QUANT_LIBRARY) {
      ResolveLibrary(k, e);
      continue;
    }
    *e = enc[k];
    if (e->mode == JXLHIP_QUANT_RAW)
      return Fail(c, JXLHIP_ERR_UNSUPPORTED, "kQuantModeRAW dequant tables are modular-coded");
    const bool has_bands = e->mode == JXLHIP_QUANT_DCT || e->mode == JXLHIP_QUANT_DCT4 ||
                           e->mode == JXLHIP_QUANT_DCT4X8 || e->mode == JXLHIP_QUANT_AFV;
    if (e->mode > JXLHIP_QUANT_RAW || (e->mode != JXLHIP_QUANT_DCT && !((kSingleBlockKinds >> k) & 1)) ||
        (has_bands && (e->num_bands < 1 || e->num_bands > JXLHIP_MAX_DISTANCE_BANDS)) ||
        (e->mode == JXLHIP_QUANT_AFV &&
         (e->num_bands_afv_4x4 < 1 || e->num_bands_afv_4x4 > JXLHIP_MAX_DISTANCE_BANDS)))
      return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "malformed quant encoding");
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->quant_enc, c->quant_enc_host, sizeof(c->quant_enc_host), hipMemcpyHostToDevice,
                           c->stream));
  LaunchDequantTables(table_dev, c->quant_enc, c->error_flag + 1, c->stream);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

int jxlhip_default_dequant_tables(jxlhip_ctx* c, float* table_dev) {
  return jxlhip_dequant_tables(c, nullptr, table_dev);
}

int jxlhip_dequant_dc(jxlhip_ctx* c, const int32_t* const quant_dc[3], float* const dc_out[3],
                      const float dc_quant[3], float cfl_x_dc, float cfl_b_dc, int smooth) {
  return jxlhip_dequant_dc_groups(c, quant_dc, dc_out, dc_quant, cfl_x_dc, cfl_b_dc, smooth, nullptr);
}

int jxlhip_dequant_dc_groups(jxlhip_ctx* c, const int32_t* const quant_dc[3], float* const dc_out[3],
                             const float dc_quant[3], float cfl_x_dc, float cfl_b_dc, int smooth,
                             const uint8_t* extra_precision) {
  if (!c || !quant_dc || !dc_out) return JXLHIP_ERR_INVALID_ARGUMENT;
  JXLHIP_NO_MULTI(c);
  if (!c->have_frame) return Fail(c, JXLHIP_ERR_STATE, "dequant_dc before frame_begin");
  for (int ch = 0; ch < 3; ch++)
    if (!quant_dc[ch] || !dc_out[ch]) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "null plane");
  HIPCHK(c, hipSetDevice(c->device));
  const DevFrame& f = c->f;
  static const float kDefaultDcQuant[3] = {1.0f / 4096.0f, 1.0f / 512.0f, 1.0f / 256.0f};
  const float* dq = dc_quant ? dc_quant : kDefaultDcQuant;
  float mul_dc[3];
  for (int ch = 0; ch < 3; ch++)
    mul_dc[ch] = (f.inv_global_scale / (float)c->p.quant_dc) * dq[ch];  // quantizer.h:133-139
  const size_t n = (size_t)f.xsb * f.ysb;
  int rc;
  if ((rc = Grow(c, &c->dc_tmp, &c->dc_tmp_floats, 3 * n))) return rc;
  float* tmp[3] = {c->dc_tmp, c->dc_tmp + n, c->dc_tmp + 2 * n};
  const uint8_t* prec_dev = nullptr;
  if (extra_precision) {
    // at most a few dozen bytes (one per 2048x2048-pixel DC group): staged behind the DC scratch
    const size_t ndc = (size_t)((f.xsb + 255) / 256) * ((f.ysb + 255) / 256);
    bool any = false;
    for (size_t i = 0; i < ndc; i++) {
      if (extra_precision[i] > 3) return Fail(c, JXLHIP_ERR_INVALID_ARGUMENT, "extra_precision > 3");
      any |= extra_precision[i] != 0;
    }
    if (any) {
      if ((rc = Grow(c, &c->dc_prec, &c->dc_prec_bytes, ndc))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->dc_prec, extra_precision, ndc, hipMemcpyHostToDevice, c->stream));
      // the source may be a short-lived host array: the copy is pageable, hence already staged by
      // the runtime when hipMemcpyAsync returns
      prec_dev = c->dc_prec;
    }
  }
  LaunchDequantDC(f.xsb, f.ysb, quant_dc, dc_out, tmp, mul_dc, cfl_x_dc, cfl_b_dc, smooth, prec_dev,
                  c->stream);
  HIPCHK(c, hipGetLastError());
  return JXLHIP_OK;
}

#include "multi.inc"
#include "codestream.inc"

}  // extern "C"
