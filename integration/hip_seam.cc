// The reference-side binding of INTEGRATION.md, compiled for real: what a libjxl maintainer adds to the tree to put the
// HIP back-end behind JxlDecoder (built into oracle/_ref/libjxl_dec_hip.so by integration/build_seam.py, on the reference
// sources where they lie; libjxl_hip.so itself never links or loads it).
//
// integration/build_seam.py makes a patched COPY of the reference's lib/jxl/dec_frame.cc (six insertions, see there)
// in which FrameDecoder::ProcessSections (1) hands a VarDCT frame's DC groups to JxlHipDcGroup() below -- the product's
// host front-end decodes them and writes the reference's own state -- and (2), once DC global / DC groups / AC global
// are in and every AC section of the frame is present, calls JxlHipTryAcGroups() below instead of
// running DecodeGroup + the CPU render pipeline per group (lib/jxl/dec_frame.cc:694-731).  This function is the
// ~150 lines a libjxl maintainer would write: it lifts the per-frame state out of PassesSharedState /
// PassesDecoderState (dec_cache.h:86-229, passes_state.h:48-95) into the C ABI of include/jxl_hip.h, hands the
// AC sections' BYTES to the product's entropy decoder on the decoder's own JxlParallelRunner
// (jxlhip_ac_groups_decode_submit), runs the HIP back-end and copies the pixels into the caller's
// JxlDecoderSetImageOutBuffer buffer or hands them row by row to the JxlDecoderSetImageOutCallback callback, in
// whatever sample format the caller chose.  Extra channels are decoded from the frame's Modular bytes by the
// product's host front-end (jxlhip_modular_*): an alpha channel going to an RGBA output is written by the back-end,
// extra-channel buffers (JxlDecoderSetExtraChannelBuffer: float, or integers of the channel's bit depth) are filled
// here.  Frames it does not take (Modular, blending, a CMS stage, tone mapping ...) fall through to the untouched
// CPU path.
//
// FrameDecoder's members are private; a maintainer would add this as a member function.  Here the class
// definition is taken as is and its access checks are lifted for this translation unit only.
// (every standard header the reference headers pull in comes first, with its access specifiers intact)
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#define private public
#define protected public
#include "lib/jxl/dec_frame.h"
#undef private
#undef protected

#include "lib/jxl/compressed_dc.h"
#include "lib/jxl/epf.h"
#include "lib/jxl/modular/modular_image.h"

#include "jxl_hip.h"
#include "jxl_hip_entropy.h"
#include "jxl_hip_frame.h"

// The seam's switches, read from the environment ONCE per process (first use), not per frame on the decoder's threads
// (getenv races with a host application's setenv): JXLHIP_SEAM_DISABLE (everything to libjxl's CPU path),
// JXLHIP_SEAM_LIBJXL_DC (the DC groups stay with libjxl), JXLHIP_SEAM_VERBOSE, JXLHIP_SEAM_DEVICE.  The tests, which
// switch them inside one process, call jxlhip_seam_reload_env().
namespace {
struct SeamSwitches {
  std::atomic<bool> loaded{false}, disable{false}, libjxl_dc{false}, verbose{false};
  std::atomic<int> device{0};
  std::mutex mu;
};
SeamSwitches g_sw;
void LoadSeamSwitchesLocked() {
  g_sw.disable.store(getenv("JXLHIP_SEAM_DISABLE") != nullptr);
  g_sw.libjxl_dc.store(getenv("JXLHIP_SEAM_LIBJXL_DC") != nullptr);
  g_sw.verbose.store(getenv("JXLHIP_SEAM_VERBOSE") != nullptr);
  const char* e = getenv("JXLHIP_SEAM_DEVICE");
  g_sw.device.store(e ? atoi(e) : 0);
  g_sw.loaded.store(true, std::memory_order_release);
}
const SeamSwitches& Seam() {
  if (!g_sw.loaded.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_sw.mu);
    if (!g_sw.loaded.load(std::memory_order_relaxed)) LoadSeamSwitchesLocked();
  }
  return g_sw;
}
}  // namespace

extern "C" {
static std::atomic<int> g_frames{0};
// how many frames went through the HIP back-end (the GPU test asserts the path was actually taken)
__attribute__((visibility("default"))) int jxlhip_seam_frames_decoded() { return g_frames.load(); }
// tests: read the JXLHIP_SEAM_* switches again (JXLHIP_SEAM_DEVICE only counts before the first frame)
__attribute__((visibility("default"))) void jxlhip_seam_reload_env() {
  std::lock_guard<std::mutex> lock(g_sw.mu);
  LoadSeamSwitchesLocked();
}
}

namespace jxl {

namespace {
// the seam's own clock (JXLHIP_SEAM_VERBOSE): when the current ProcessSections call began, when the last frame left
double NowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::atomic<double> g_sections_begin{0.0}, g_last_frame_end{0.0};
}  // namespace
void JxlHipNoteSectionsBegin() { g_sections_begin.store(NowMs(), std::memory_order_relaxed); }

// ---- the DC groups of a VarDCT frame through the product's host front-end (round 5) ------------------------------------
// FrameDecoder::ProcessDCGroup = DecodeVarDCTDC + the ModularDC stream + DecodeAcMetadata (dec_frame.cc:318-342,
// dec_modular.cc:427-562): three Modular streams per 2048 x 2048 pixels, decoded by libjxl's general Modular loop -- on
// the 8K stream of bench.py 15 ms per frame on the runner's threads, the longest single item of a djxl repetition once
// the AC groups run on the device.  The product's front-end has its own decoder for exactly these streams
// (jxlhip_dc_group_decode, include/jxl_hip_frame.h: specialised channel loops, 4.6 ms for the largest group); here its
// output is written INTO the reference's state -- dc_storage / quant_dc through libjxl's own DequantDC, the strategy
// image, quant field, sharpness and colour-correlation maps as DecodeAcMetadata writes them -- so that everything
// behind (FinalizeDC, ProcessACGlobal, the AC seam below or libjxl's CPU path) finds what ProcessDCGroup would have
// left.  The CPU suite decodes whole files this way WITHOUT a device and compares with the unpatched decoder bit for bit.
// Anything the front-end does not take (JXLHIP_ERR_UNSUPPORTED: transforms libjxl's encoder does not write here,
// squeezed extra channels, chroma subsampling, a DC frame) leaves the group to ProcessDCGroup, untouched.
namespace {
struct DcFrameState {
  jxlhip_frame_header fh = {};
  jxlhip_modular_tree* tree = nullptr;
  std::vector<int32_t> qdc, rq;  // qdc: X, Y, B planes of xsb x ysb
  std::vector<uint8_t> acs, sharp;
  std::vector<int8_t> ytox, ytob;
  std::atomic<uint32_t> groups_taken{0};
  ~DcFrameState() { jxlhip_modular_tree_destroy(tree); }
};
std::mutex g_dc_mu;
std::vector<std::pair<const FrameDecoder*, std::shared_ptr<DcFrameState>>> g_dc;  // (a handful of frames at most)

std::shared_ptr<DcFrameState> DcStateOf(const FrameDecoder* fd, bool take = false) {
  std::lock_guard<std::mutex> lock(g_dc_mu);
  for (size_t i = 0; i < g_dc.size(); i++)
    if (g_dc[i].first == fd) {
      auto s = g_dc[i].second;
      if (take) g_dc.erase(g_dc.begin() + i);
      return s;
    }
  return nullptr;
}

void FillFrameHeader(const FrameDecoder* fd, jxlhip_frame_header* mfh) {
  const FrameHeader& fh = fd->frame_header_;
  const FrameDimensions& dim = fd->frame_dim_;
  const ImageMetadata& md = fh.nonserialized_metadata->m;
  mfh->upsampling = fh.upsampling;
  mfh->flags = fh.flags;
  mfh->num_passes = static_cast<uint32_t>(fh.passes.num_passes);
  mfh->num_downsample = fh.passes.num_downsample;
  for (uint32_t i = 0; i < fh.passes.num_downsample; i++) {
    mfh->downsample[i] = fh.passes.downsample[i];
    mfh->last_pass[i] = fh.passes.last_pass[i];
  }
  mfh->xsize = static_cast<uint32_t>(dim.xsize);
  mfh->ysize = static_cast<uint32_t>(dim.ysize);
  mfh->xsize_blocks = static_cast<uint32_t>(dim.xsize_blocks);
  mfh->ysize_blocks = static_cast<uint32_t>(dim.ysize_blocks);
  mfh->group_dim = static_cast<uint32_t>(dim.group_dim);
  mfh->xsize_groups = static_cast<uint32_t>(dim.xsize_groups);
  mfh->ysize_groups = static_cast<uint32_t>(dim.ysize_groups);
  mfh->num_groups = dim.num_groups;
  mfh->num_dc_groups = dim.num_dc_groups;
  mfh->num_extra_channels = md.num_extra_channels;
  mfh->image_bits = md.bit_depth.bits_per_sample;
  for (size_t i = 0; i < md.num_extra_channels && i < 4; i++) mfh->ec_upsampling[i] = fh.extra_channel_upsampling[i];
}
}  // namespace

// Called when ProcessDCGlobal has succeeded, with the DC-global section's reader: the product's view of that section
// (quantizer and the global Modular tree), which its DC-group decoder needs.
void JxlHipAfterDcGlobal(FrameDecoder* fd, const BitReader* br) {
  (void)DcStateOf(fd, /*take=*/true);  // (a frame decoder that starts over)
  if (Seam().disable.load(std::memory_order_relaxed) || Seam().libjxl_dc.load(std::memory_order_relaxed)) return;
  const FrameHeader& fh = fd->frame_header_;
  const FrameDimensions& dim = fd->frame_dim_;
  const ImageMetadata& md = fh.nonserialized_metadata->m;
  if (fh.encoding != FrameEncoding::kVarDCT || (fh.flags & FrameHeader::kUseDcFrame) || !fh.chroma_subsampling.Is444() ||
      fh.upsampling != 1 || md.num_extra_channels > 4)
    return;
  if (dim.num_groups == 1 && fh.passes.num_passes == 1) return;  // a one-section frame: nothing to gain
  auto st = std::make_shared<DcFrameState>();
  FillFrameHeader(fd, &st->fh);
  jxlhip_dc_global dcg;
  size_t pos = 0;
  if (jxlhip_dc_global_decode(br->FirstByte(), br->TotalBytes(), &pos, fh.flags, &dcg) != JXLHIP_OK) return;
  if (jxlhip_modular_global_decode(br->FirstByte(), br->TotalBytes(), &pos, &st->fh, &st->tree) != JXLHIP_OK) return;
  if (st->tree && jxlhip_modular_uses_dc_groups(st->tree)) return;  // extra channels with levels in the DC groups: libjxl keeps them
  const size_t nb = dim.xsize_blocks * dim.ysize_blocks;
  const size_t nt = ((dim.xsize_blocks + 7) / 8) * ((dim.ysize_blocks + 7) / 8);
  st->qdc.assign(3 * nb, 0);
  st->rq.assign(nb, 0);
  st->acs.assign(nb, 0);
  st->sharp.assign(nb, 0);
  st->ytox.assign(nt, 0);
  st->ytob.assign(nt, 0);
  std::lock_guard<std::mutex> lock(g_dc_mu);
  // (entries leave when their frame reaches its AC groups or starts over, DcStateOf(.., take); what is dropped here are
  // frames that never got there -- with 64 slots a process must hold 64 half-decoded frames before a live one loses its
  // fast path, which then only costs speed: its DC groups fall back to libjxl's own decode)
  if (g_dc.size() >= 64) g_dc.erase(g_dc.begin());
  g_dc.emplace_back(fd, std::move(st));
}

// One DC group (on a runner thread).  *handled = false: ProcessDCGroup decodes it.
Status JxlHipDcGroup(FrameDecoder* fd, size_t dc_group, BitReader* br, bool* handled) {
  *handled = false;
  const std::shared_ptr<DcFrameState> st = DcStateOf(fd);
  if (!st) return true;
  PassesDecoderState* ds = fd->dec_state_;
  const FrameHeader& fh = fd->frame_header_;
  const FrameDimensions& dim = fd->frame_dim_;
  const size_t xsb = dim.xsize_blocks, nb = xsb * dim.ysize_blocks;
  int32_t* qdc3[3] = {st->qdc.data(), st->qdc.data() + nb, st->qdc.data() + 2 * nb};
  uint32_t precision = 0, used = 0;
  size_t pos = 0;
  const int rc = jxlhip_dc_group_decode(st->tree, br->FirstByte(), br->TotalBytes(), &pos, &st->fh, static_cast<uint32_t>(dc_group),
                                        qdc3, &precision, st->acs.data(), st->rq.data(), st->sharp.data(), st->ytox.data(),
                                        st->ytob.data(), &used);
  if (rc != JXLHIP_OK) return true;  // unsupported or damaged: the reference decodes the section and reports
  JxlMemoryManager* mm = ds->memory_manager();
  const Rect r = dim.DCGroupRect(dc_group);
  {  // DecodeVarDCTDC's tail (dec_modular.cc:459-463): channel 0 = Y, 1 = X, 2 = B of the stream's image
    JXL_ASSIGN_OR_RETURN(Image image, Image::Create(mm, r.xsize(), r.ysize(), 8, 3));
    static const int kPlaneOf[3] = {1, 0, 2};
    for (int c = 0; c < 3; c++)
      for (size_t y = 0; y < r.ysize(); y++)
        memcpy(image.channel[c].plane.Row(y), qdc3[kPlaneOf[c]] + (r.y0() + y) * xsb + r.x0(), r.xsize() * sizeof(int32_t));
    DequantDC(r, &ds->shared_storage.dc_storage, &ds->shared_storage.quant_dc, image, ds->shared->quantizer.MulDC(),
              1.0f / static_cast<float>(1u << precision), ds->shared->cmap.base().DCFactors(), fh.chroma_subsampling,
              ds->shared->block_ctx_map);
  }
  {  // DecodeAcMetadata's tail (dec_modular.cc:497-559)
    const size_t xt = (xsb + 7) / 8;
    const Rect cr(r.x0() >> 3, r.y0() >> 3, (r.xsize() + 7) >> 3, (r.ysize() + 7) >> 3);
    for (size_t y = 0; y < cr.ysize(); y++) {
      memcpy(cr.Row(&ds->shared_storage.cmap.ytox_map, y), &st->ytox[(cr.y0() + y) * xt + cr.x0()], cr.xsize());
      memcpy(cr.Row(&ds->shared_storage.cmap.ytob_map, y), &st->ytob[(cr.y0() + y) * xt + cr.x0()], cr.xsize());
    }
    auto& ac_strategy = ds->shared_storage.ac_strategy;
    for (size_t iy = 0; iy < r.ysize(); iy++) {
      const size_t y = r.y0() + iy;
      int32_t* row_qf = r.Row(&ds->shared_storage.raw_quant_field, iy);
      uint8_t* row_epf = r.Row(&ds->shared_storage.epf_sharpness, iy);
      const uint8_t* a = &st->acs[y * xsb + r.x0()];
      const int32_t* q = &st->rq[y * xsb + r.x0()];
      memcpy(row_epf, &st->sharp[y * xsb + r.x0()], r.xsize());
      for (size_t ix = 0; ix < r.xsize(); ix++) {
        if (!(a[ix] & 1)) continue;  // (a block a larger varblock covers)
        JXL_RETURN_IF_ERROR(ac_strategy.SetNoBoundsCheck(r.x0() + ix, y, static_cast<AcStrategyType>(a[ix] >> 1)));
        row_qf[ix] = q[ix];
      }
    }
    ds->used_acs |= used;
    if (fh.loop_filter.epf_iters > 0) JXL_RETURN_IF_ERROR(ComputeSigma(fh.loop_filter, r, ds));
  }
  br->SkipBits(pos);
  fd->decoded_dc_groups_[dc_group] = JXL_TRUE;
  st->groups_taken.fetch_add(1);
  *handled = true;
  return true;
}

namespace {
jxlhip_ctx* Context() {
  static jxlhip_ctx* ctx = [] {
    jxlhip_ctx* c = nullptr;
    if (jxlhip_create(Seam().device.load(std::memory_order_relaxed), &c) != JXLHIP_OK) return static_cast<jxlhip_ctx*>(nullptr);
    return c;
  }();
  return ctx;
}

struct HipPasses {
  jxlhip_ac_pass* p[11] = {nullptr};
  ~HipPasses() {
    for (auto* q : p)
      if (q) jxlhip_ac_pass_destroy(q);
  }
};
}  // namespace

Status JxlHipTryAcGroups(FrameDecoder* fd, const FrameDecoder::SectionInfo* sections, size_t num,
                         const std::vector<std::vector<size_t>>& ac_group_sec,
                         const std::vector<size_t>& desired_num_ac_passes, size_t ac_global_sec,
                         size_t ac_global_bit, FrameDecoder::SectionStatus* section_status, bool* done) {
  *done = false;
  const std::shared_ptr<DcFrameState> dc_state = DcStateOf(fd, /*take=*/true);  // (JxlHipDcGroup's state: its job is done)
  if (Seam().disable.load(std::memory_order_relaxed)) return true;
  const bool verbose = Seam().verbose.load(std::memory_order_relaxed);
  if (verbose && dc_state)
    fprintf(stderr, "jxlhip seam: %u of %zu DC groups decoded by the product's front-end into libjxl's state\n",
            dc_state->groups_taken.load(), static_cast<size_t>(fd->frame_dim_.num_dc_groups));
  auto decline = [&](const char* why) -> Status {  // the CPU path takes the frame
    if (verbose) fprintf(stderr, "jxlhip seam declines the frame: %s\n", why);
    return true;
  };

  const FrameHeader& fh = fd->frame_header_;
  PassesDecoderState* ds = fd->dec_state_;
  const PassesSharedState& sh = *ds->shared;
  const FrameDimensions& dim = fd->frame_dim_;
  const ImageMetadata& md = fh.nonserialized_metadata->m;
  const OutputEncodingInfo& oe = ds->output_encoding_info;
  const ImageOutput& mo = ds->main_output;
  const size_t np = fh.passes.num_passes;
  // ---- frames the back-end takes; everything else keeps the CPU path
  if (!fd->decoded_ac_global_ || ac_global_sec == num) return true;  // AC global must arrive with the groups
  if (fh.encoding != FrameEncoding::kVarDCT || fh.color_transform != ColorTransform::kXYB || !md.xyb_encoded) return decline("not an XYB VarDCT frame");
  if (fh.flags & (FrameHeader::kNoise | FrameHeader::kPatches | FrameHeader::kSplines | FrameHeader::kUseDcFrame)) return decline("noise / patches / splines / DC frame");
  if (!fh.chroma_subsampling.Is444() || fh.upsampling != 1 || fh.dc_level != 0) return decline("chroma subsampling / upsampling / DC level");
  if (fh.custom_size_or_origin || fh.blending_info.mode != BlendMode::kReplace) return decline("frame origin / blending");
  // extra channels: only the first alpha channel is ever written, and only into an RGBA main output
  // (WriteToOutputStage, stage_write.cc:288-366); separate extra-channel outputs and un-premultiplication: CPU path
  int alpha_ec = -1;
  if (md.num_extra_channels > 4) return decline("more than four extra channels");
  for (size_t i = 0; i < md.num_extra_channels; i++) {
    const ExtraChannelInfo& e = md.extra_channel_info[i];
    if (e.dim_shift != 0 || e.bit_depth.floating_point_sample || e.bit_depth.bits_per_sample > 24)
      return decline("extra channel format");
    if (fh.extra_channel_upsampling[i] != 1) return decline("extra channel upsampling");
    if (fh.extra_channel_blending_info[i].mode != BlendMode::kReplace) return decline("extra channel blending");
    if (e.type == ExtraChannel::kAlpha && alpha_ec < 0) alpha_ec = static_cast<int>(i);
  }
  // extra channels asked for in buffers of their own (JxlDecoderSetExtraChannelBuffer: what djxl does with the alpha
  // channel for .ppm / .npy, lib/extras/dec/jxl.cc:574-607): float samples are handed out by this function from the
  // planes the host front-end decodes; integer conversions stay with the CPU path
  bool extra_buffers = false;
  for (size_t i = 0; i < ds->extra_output.size(); i++) {
    const ImageOutput& eo = ds->extra_output[i];
    if (eo.callback.IsPresent()) return decline("extra-channel callback");
    if (!eo.buffer) continue;
    if (i >= md.num_extra_channels || eo.format.num_channels != 1) return decline("extra-channel buffer layout");
    // float samples, or integers of the channel's own bit depth (then the sample IS the coded integer: x (2^bits - 1),
    // the 8-bit dither of magnitude < 0.5 and the rounding of MakeUnsigned, stage_write.cc:263-284, land on it)
    const uint32_t ec_bits = md.extra_channel_info[i].bit_depth.bits_per_sample;
    const bool is_float = eo.format.data_type == JXL_TYPE_FLOAT && eo.format.endianness != JXL_BIG_ENDIAN;
    const bool is_int = (eo.format.data_type == JXL_TYPE_UINT8 && ec_bits <= 8 && eo.bits_per_sample == ec_bits) ||
                        (eo.format.data_type == JXL_TYPE_UINT16 && ec_bits <= 16 && eo.bits_per_sample == ec_bits);
    if (!is_float && !is_int) return decline("extra-channel buffer: neither float nor integers of the channel's bit depth");
    extra_buffers = true;
  }
  if (alpha_ec >= 0 && ds->unpremul_alpha) return decline("un-premultiplied alpha");
  const bool alpha_in_main = alpha_ec >= 0 && (mo.format.num_channels == 4 || mo.format.num_channels == 2);
  const bool want_alpha = alpha_in_main || extra_buffers;  // = the frame's Modular image is needed
  if (!fh.is_last || fh.CanBeReferenced() || fh.frame_type != FrameType::kRegularFrame) return decline("not a single regular frame");
  if (fd->decoded_->IsJPEG()) return decline("JPEG reconstruction");
  // ---- the output: every ImageOutput WriteToOutputStage serves for a colour image without alpha
  // (stage_write.cc:288-700) -- buffer or row callback, uint8 / uint16 / float16 / float, either endianness,
  // RGB or RGBA (opaque alpha) -- which is what djxl asks for: image-out callback always
  // (lib/extras/dec/jxl.h:64, jxl.cc:543-556), uint8 / uint16 for PNG / PPM, big-endian for PNM
  // (lib/extras/enc/pnm.cc:118-132), float for PFM / NPY
  const bool to_callback = mo.callback.IsPresent();
  if (!to_callback && !mo.buffer) return decline("no image output set");
  // grey outputs (1 or 2 channels, what djxl asks for from a grey image: .pgm / .npy): the back-end writes RGB(A) --
  // for a grey original the rows of its opsin inverse are equal (luminances x matrix, dec_xyb.cc:226-230), R = G = B
  // -- and the first (and alpha) sample of every pixel is handed out below
  const uint32_t out_nc = mo.format.num_channels;
  const bool grey_out = out_nc == 1 || out_nc == 2;
  if (out_nc < 1 || out_nc > 4) return decline("output channel count");
  if (grey_out && !oe.color_encoding.IsGray()) return decline("grey output of a colour image");
  const uint32_t dev_nc = out_nc == 1 ? 3 : (out_nc == 2 ? 4 : out_nc);
  uint32_t sample_type, bits = 32;
  switch (mo.format.data_type) {
    case JXL_TYPE_FLOAT: sample_type = JXLHIP_SAMPLE_F32; break;
    case JXL_TYPE_FLOAT16: sample_type = JXLHIP_SAMPLE_F16; bits = 16; break;
    case JXL_TYPE_UINT16: sample_type = JXLHIP_SAMPLE_U16; bits = static_cast<uint32_t>(mo.bits_per_sample); break;
    case JXL_TYPE_UINT8: sample_type = JXLHIP_SAMPLE_U8; bits = static_cast<uint32_t>(mo.bits_per_sample); break;
    default: return decline("output sample type");
  }
  if (sample_type == JXLHIP_SAMPLE_U8 && (bits < 1 || bits > 8)) return decline("output bit depth");
  if (sample_type == JXLHIP_SAMPLE_U16 && (bits < 1 || bits > 16)) return decline("output bit depth");
  // the stage list behind XYBStage must be FromLinearStage alone (dec_cache.cc:255-345): the output space is an
  // RGB one, no CMS stage (the encoding is the original one or no CMS was given), no tone mapping
  // (stage_tone_mapping.cc:33-60)
  if (oe.color_encoding.GetColorSpace() == ColorSpace::kXYB) return decline("XYB output");
  if (oe.color_encoding.IsGray() && oe.color_encoding.GetWhitePointType() != WhitePoint::kD65) return decline("grey, not D65");
  if (!oe.color_encoding_is_original && oe.cms_set) return decline("a CMS stage is needed");
  {
    const auto& otf = oe.orig_color_encoding.Tf();
    if (oe.desired_intensity_target != oe.orig_intensity_target &&
        ((otf.IsPQ() && oe.desired_intensity_target < oe.orig_intensity_target) ||
         (otf.IsHLG() && !oe.color_encoding.Tf().IsHLG())))
      return decline("tone mapping");
  }
  uint32_t transfer;
  float tf_param = 0.0f;
  {  // GetFromLinearStage's choice (stage_from_linear.cc:161-182)
    const auto& tf = oe.color_encoding.Tf();
    if (tf.IsLinear()) transfer = JXLHIP_TF_LINEAR;
    else if (tf.IsSRGB()) transfer = JXLHIP_TF_SRGB;
    else if (tf.IsPQ()) transfer = JXLHIP_TF_PQ, tf_param = oe.orig_intensity_target;
    else if (tf.IsHLG()) transfer = JXLHIP_TF_HLG, tf_param = oe.desired_intensity_target;
    else if (tf.Is709()) transfer = JXLHIP_TF_709;
    else if (tf.have_gamma || tf.IsDCI()) transfer = JXLHIP_TF_GAMMA, tf_param = oe.inverse_gamma;
    else return decline("transfer function of the output");
  }
  for (size_t g = 0; g < dim.num_groups; g++)  // the whole frame, nothing drawn yet
    if (desired_num_ac_passes[g] != np || fd->decoded_passes_per_ac_group_[g] != 0) return decline("not every pass of every AC group at once");
  jxlhip_ctx* ctx = Context();
  if (!ctx) return decline("no device");

  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_side = 0, t_entropy = 0, t_decode = 0;
  auto check = [&](int rc, const char* what) -> Status {
    if (rc == JXLHIP_OK) return true;
    if (verbose) fprintf(stderr, "jxlhip seam: %s failed: %s (%s)\n", what, jxlhip_status_string(rc), jxlhip_last_error(ctx));
    return JXL_FAILURE("jxlhip %s: %s (%s)", what, jxlhip_status_string(rc), jxlhip_last_error(ctx));
  };
  // ---- alpha: the global part of the frame's Modular image once more, from the DC-global section's bytes, into the
  // product's own state (the reference's copy lives in ModularFrameDecoder and feeds its render pipeline directly)
  struct TreeOwner {
    jxlhip_modular_tree* t = nullptr;
    ~TreeOwner() { jxlhip_modular_tree_destroy(t); }
  } mtree;
  jxlhip_frame_header mfh = {};
  if (want_alpha) {
    const BitReader* dbr = nullptr;
    for (size_t i = 0; i < num; i++)
      if (sections[i].id == 0) dbr = sections[i].br;  // DC global (or the frame's only section)
    if (!dbr) return decline("DC global arrived in an earlier call");
    FillFrameHeader(fd, &mfh);
    jxlhip_dc_global dcg;
    size_t mpos = 0;
    int rc = jxlhip_dc_global_decode(dbr->FirstByte(), dbr->TotalBytes(), &mpos, fh.flags, &dcg);
    if (rc == JXLHIP_OK) rc = jxlhip_modular_global_decode(dbr->FirstByte(), dbr->TotalBytes(), &mpos, &mfh, &mtree.t);
    if (rc == JXLHIP_ERR_UNSUPPORTED) return decline("extra channels coded with transforms outside the front-end");
    JXL_RETURN_IF_ERROR(check(rc, "modular global"));
    // squeezed (progressive) extra channels: their coarse levels came with the DC groups, which libjxl has already
    // taken in -- the product's front-end reads them in jxlhip_dc_group_decode (jxlhip_decode_codestream), not here
    if (jxlhip_modular_uses_dc_groups(mtree.t)) return decline("squeezed extra channels (levels in the DC groups)");
  }
  // ---- per-frame parameters
  jxlhip_frame_params p = {};
  p.xsize = dim.xsize;
  p.ysize = dim.ysize;
  p.output_kind = JXLHIP_OUT_PACKED;
  p.undo_orientation = static_cast<uint32_t>(ds->undo_orientation);  // the back-end writes display orientation
  p.global_scale = sh.quantizer.global_scale_;
  p.quant_dc = sh.quantizer.quant_dc_;
  p.x_dm_multiplier = ds->x_dm_multiplier;
  p.b_dm_multiplier = ds->b_dm_multiplier;
  memcpy(p.quant_biases, oe.opsin_params.quant_biases, sizeof(p.quant_biases));
  p.cfl_base_x = sh.cmap.base().GetBaseCorrelationX();
  p.cfl_base_b = sh.cmap.base().GetBaseCorrelationB();
  p.cfl_color_factor = static_cast<uint32_t>(sh.cmap.base().GetColorFactor());
  const LoopFilter& lf = fh.loop_filter;
  p.lf.gab = lf.gab ? 1 : 0;
  const float gw[6] = {lf.gab_x_weight1, lf.gab_x_weight2, lf.gab_y_weight1, lf.gab_y_weight2, lf.gab_b_weight1, lf.gab_b_weight2};
  memcpy(p.lf.gab_weights, gw, sizeof(gw));
  p.lf.epf_iters = lf.epf_iters;
  memcpy(p.lf.epf_sharp_lut, lf.epf_sharp_lut, sizeof(p.lf.epf_sharp_lut));
  memcpy(p.lf.epf_channel_scale, lf.epf_channel_scale, sizeof(p.lf.epf_channel_scale));
  p.lf.epf_quant_mul = lf.epf_quant_mul;
  p.lf.epf_pass0_sigma_scale = lf.epf_pass0_sigma_scale;
  p.lf.epf_pass2_sigma_scale = lf.epf_pass2_sigma_scale;
  p.lf.epf_border_sad_mul = lf.epf_border_sad_mul;
  for (int i = 0; i < 3; i++) p.opsin_biases[i] = oe.opsin_params.opsin_biases[i];
  for (int i = 0; i < 9; i++) p.inverse_opsin_matrix[i] = oe.opsin_params.inverse_opsin_matrix[i * 4];
  p.out_format.transfer = transfer;
  p.out_format.sample_type = sample_type;
  p.out_format.num_channels = dev_nc;
  p.out_format.bits_per_sample = bits;
  // SwapEndianness (stage_write.cc:238-252): this host is little-endian
  p.out_format.swap_endianness = (mo.format.endianness == JXL_BIG_ENDIAN && sample_type != JXLHIP_SAMPLE_U8) ? 1 : 0;
  p.out_format.tf_param = tf_param;
  for (int i = 0; i < 3; i++) p.out_format.luminances[i] = oe.luminances[i];
  p.used_acs = ds->used_acs;

  // ---- side info out of PassesSharedState, as dense arrays
  const size_t xsb = dim.xsize_blocks, ysb = dim.ysize_blocks, nb = xsb * ysb;
  const size_t xt = (xsb + 7) / 8, yt = (ysb + 7) / 8;
  std::vector<uint8_t> acs(nb), sharp(nb), qctx(nb);
  std::vector<int32_t> rq(nb);
  std::vector<int8_t> ytox(xt * yt), ytob(xt * yt);
  std::vector<float> dc(3 * nb);
  // the dense arrays are at hand already when every DC group came through the product's front-end (JxlHipDcGroup)
  const bool dense_at_hand = dc_state && dc_state->groups_taken.load() == dim.num_dc_groups && dc_state->acs.size() == nb;
  if (dense_at_hand) {
    acs.swap(dc_state->acs);
    rq.swap(dc_state->rq);
    sharp.swap(dc_state->sharp);
    ytox.swap(dc_state->ytox);
    ytob.swap(dc_state->ytob);
  }
  for (size_t y = 0; y < ysb; y++) {
    if (!dense_at_hand) {
      const AcStrategyRow row = sh.ac_strategy.ConstRow(y);
      for (size_t x = 0; x < xsb; x++)
        acs[y * xsb + x] = static_cast<uint8_t>((row[x].RawStrategy() << 1) | (row[x].IsFirstBlock() ? 1 : 0));
      memcpy(&rq[y * xsb], sh.raw_quant_field.ConstRow(y), xsb * sizeof(int32_t));
      memcpy(&sharp[y * xsb], sh.epf_sharpness.ConstRow(y), xsb);
    }
    memcpy(&qctx[y * xsb], sh.quant_dc.ConstRow(y), xsb);
    for (int c = 0; c < 3; c++) memcpy(&dc[c * nb + y * xsb], sh.dc->ConstPlaneRow(c, y), xsb * sizeof(float));
  }
  for (size_t y = 0; y < yt && !dense_at_hand; y++) {
    memcpy(&ytox[y * xt], sh.cmap.ytox_map.ConstRow(y), xt);
    memcpy(&ytob[y * xt], sh.cmap.ytob_map.ConstRow(y), xt);
  }
  const float* dc3[3] = {dc.data(), dc.data() + nb, dc.data() + 2 * nb};
  const float* table = sh.matrices.Matrix(AcStrategyType::DCT, 0);  // = table_ (quant_weights.h:364-367), EnsureComputed by ProcessACGlobal

  t_side = now();
  // ---- AC global once more, from its bytes, into the product's pass objects (histograms, coefficient orders)
  jxlhip_block_ctx_map bcm = {};
  for (int c = 0; c < 3; c++) {
    bcm.num_dc_thresholds[c] = static_cast<uint32_t>(sh.block_ctx_map.dc_thresholds[c].size());
    for (size_t i = 0; i < sh.block_ctx_map.dc_thresholds[c].size(); i++) bcm.dc_thresholds[c][i] = sh.block_ctx_map.dc_thresholds[c][i];
  }
  bcm.num_dc_ctxs = static_cast<uint32_t>(sh.block_ctx_map.num_dc_ctxs);
  bcm.num_qf_thresholds = static_cast<uint32_t>(sh.block_ctx_map.qf_thresholds.size());
  for (size_t i = 0; i < sh.block_ctx_map.qf_thresholds.size(); i++) bcm.qf_thresholds[i] = sh.block_ctx_map.qf_thresholds[i];
  bcm.num_ctxs = static_cast<uint32_t>(sh.block_ctx_map.num_ctxs);
  bcm.ctx_map_size = static_cast<uint32_t>(sh.block_ctx_map.ctx_map.size());
  if (bcm.ctx_map_size > JXLHIP_BLOCK_CTX_MAP_MAX) return true;
  memcpy(bcm.ctx_map, sh.block_ctx_map.ctx_map.data(), bcm.ctx_map_size);
  const BitReader* gbr = sections[ac_global_sec].br;
  jxlhip_quant_encoding enc[JXLHIP_NUM_QUANT_TABLES];
  uint32_t num_hist = 0;
  HipPasses passes;
  size_t gpos = ac_global_bit;
  JXL_RETURN_IF_ERROR(check(jxlhip_ac_global_decode_at(gbr->FirstByte(), gbr->TotalBytes(), &gpos, dim.num_groups, np,
                                                       ds->used_acs, &bcm, enc, &num_hist, passes.p),
                            "AC global"));

  // ---- the groups' section bytes; a one-section frame carries its AC group behind AC global in the same reader
  const bool single = dim.num_groups == 1 && np == 1;
  std::vector<const uint8_t*> sec(np * dim.num_groups);
  std::vector<size_t> sec_size(np * dim.num_groups);
  std::vector<uint32_t> shifts(np);
  for (size_t ps = 0; ps < np; ps++) shifts[ps] = fh.passes.shift[ps];
  for (size_t ps = 0; ps < np && !single; ps++)
    for (size_t g = 0; g < dim.num_groups; g++) {
      const BitReader* br = sections[ac_group_sec[g][ps]].br;
      sec[ps * dim.num_groups + g] = br->FirstByte();
      sec_size[ps * dim.num_groups + g] = br->TotalBytes();
    }

  std::vector<size_t> end_bits(want_alpha ? np * dim.num_groups : 0, 0);
  const jxlhip_ac_pass* pp[11];
  for (size_t i = 0; i < np; i++) pp[i] = passes.p[i];
  JxlParallelRunner runner = fd->pool_ ? fd->pool_->runner() : nullptr;
  void* runner_opaque = fd->pool_ ? fd->pool_->runner_opaque() : nullptr;
  for (uint32_t ct = JXLHIP_COEFF_I16; ct <= JXLHIP_COEFF_I32; ct++) {
    p.coeff_type = ct;
    JXL_RETURN_IF_ERROR(check(jxlhip_frame_begin(ctx, &p), "frame_begin"));
    JXL_RETURN_IF_ERROR(check(jxlhip_upload_side_info(ctx, acs.data(), rq.data(), sharp.data(), ytox.data(), ytob.data(), dc3, table),
                              "upload_side_info"));
    int rc;
    if (single) {
      const uint8_t* d1[1] = {gbr->FirstByte()};
      size_t s1[1] = {gbr->TotalBytes()}, b1[1] = {gpos};
      rc = jxlhip_ac_group_decode_submit_passes(ctx, 1, pp, shifts.data(), 0, acs.data(), rq.data(), qctx.data(), d1, s1, b1);
      if (want_alpha) {
        end_bits[0] = b1[0];
        sec[0] = d1[0];
        sec_size[0] = s1[0];
      }
    } else {
      rc = jxlhip_ac_groups_decode_submit_ex(ctx, reinterpret_cast<jxlhip_parallel_runner>(runner), runner_opaque,
                                             static_cast<uint32_t>(np), pp, shifts.data(), acs.data(), rq.data(), qctx.data(),
                                             sec.data(), sec_size.data(), want_alpha ? end_bits.data() : nullptr);
    }
    if (rc == JXLHIP_ERR_RANGE && ct == JXLHIP_COEFF_I16) continue;  // a coefficient needs 32 bits: redo
    JXL_RETURN_IF_ERROR(check(rc, "AC groups"));
    break;
  }
  std::vector<float> alpha;
  if (want_alpha) {
    // what follows the coefficients in every AC-group section (ProcessACGroup's second half, dec_frame.cc:497-530),
    // on the decoder's pool; then the samples as floats, as ModularImageToDecodedRect makes them
    std::atomic<int> status{JXLHIP_OK};
    // Alpha for the main output only, in a channel larger than a group: the groups' threads write the float samples
    // straight into the context's pinned plane.  Otherwise the samples are collected and converted below.
    const bool direct = alpha_in_main && !extra_buffers && (dim.xsize > dim.group_dim || dim.ysize > dim.group_dim) &&
                        jxlhip_modular_groups_are_final(mtree.t);
    float* staging = nullptr;
    size_t staging_stride = 0;
    uint32_t ec_bits[4] = {8, 8, 8, 8};
    float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    if (direct) {
      JXL_RETURN_IF_ERROR(check(jxlhip_alpha_staging(ctx, &staging, &staging_stride), "alpha_staging"));
      for (size_t i = 0; i < md.num_extra_channels; i++) ec_bits[i] = md.extra_channel_info[i].bit_depth.bits_per_sample;
      planes[alpha_ec] = staging;
    }
    const auto group = [&](uint32_t g, size_t /*thread*/) -> Status {
      for (size_t ps = 0; ps < np; ps++) {
        const size_t i = ps * dim.num_groups + g;
        size_t pos = end_bits[i];
        const int rc = direct ? jxlhip_modular_ac_group_decode_f32(mtree.t, &mfh, g, static_cast<uint32_t>(ps), sec[i], sec_size[i],
                                                                   &pos, ec_bits, md.bit_depth.bits_per_sample, planes, staging_stride)
                              : jxlhip_modular_ac_group_decode(mtree.t, &mfh, g, static_cast<uint32_t>(ps), sec[i], sec_size[i], &pos);
        if (rc != JXLHIP_OK) {
          int expected = JXLHIP_OK;
          status.compare_exchange_strong(expected, rc);
          break;
        }
      }
      return true;
    };
    JXL_RETURN_IF_ERROR(RunOnPool(fd->pool_, 0, static_cast<uint32_t>(dim.num_groups), ThreadPool::NoInit, group, "jxlhip modular"));
    JXL_RETURN_IF_ERROR(check(status.load(), "modular AC groups"));
    if (direct) {
      JXL_RETURN_IF_ERROR(check(jxlhip_set_alpha(ctx, staging, staging_stride), "set_alpha"));
    } else {
      alpha.resize(dim.xsize * dim.ysize);
      // extra-channel buffers: the plane in display orientation (WriteToOutputStage's flips / transpose,
      // stage_write.cc:441-457,664-680), rows of `stride` bytes
      const uint32_t o = static_cast<uint32_t>(ds->undo_orientation);
      const bool fx = o == 2 || o == 3 || o == 7 || o == 8, fy = o == 3 || o == 4 || o == 6 || o == 7, tr = o >= 5;
      for (size_t i = 0; i < ds->extra_output.size(); i++) {
        const ImageOutput& eo = ds->extra_output[i];
        if (!eo.buffer) continue;
        JXL_RETURN_IF_ERROR(check(jxlhip_modular_extra_channel_f32(mtree.t, static_cast<uint32_t>(i),
                                                                   md.extra_channel_info[i].bit_depth.bits_per_sample,
                                                                   md.bit_depth.bits_per_sample, alpha.data(), dim.xsize),
                                  "extra channel samples"));
        const size_t ow = tr ? dim.ysize : dim.xsize, oh = tr ? dim.xsize : dim.ysize;
        const size_t sb = eo.format.data_type == JXL_TYPE_FLOAT ? 4 : (eo.format.data_type == JXL_TYPE_UINT8 ? 1 : 2);
        const float maxval = static_cast<float>((1u << md.extra_channel_info[i].bit_depth.bits_per_sample) - 1);
        const bool swap = sb == 2 && eo.format.endianness == JXL_BIG_ENDIAN;
        if (eo.stride < ow * sb || eo.buffer_size < (oh - 1) * eo.stride + ow * sb)
          return JXL_FAILURE("extra channel buffer too small");
        const auto row = [&](uint32_t y, size_t /*thread*/) -> Status {
          const float* src = alpha.data() + static_cast<size_t>(y) * dim.xsize;
          const size_t yo = fy ? dim.ysize - 1 - y : y;
          for (size_t x = 0; x < dim.xsize; x++) {
            const size_t xo = fx ? dim.xsize - 1 - x : x;
            char* d = static_cast<char*>(eo.buffer) + (tr ? xo * eo.stride + yo * sb : yo * eo.stride + xo * sb);
            if (sb == 4) {
              memcpy(d, &src[x], 4);
            } else {
              const uint32_t k = static_cast<uint32_t>(lrintf(src[x] * maxval));
              if (sb == 1) {
                *reinterpret_cast<uint8_t*>(d) = static_cast<uint8_t>(k);
              } else {
                const uint16_t v = swap ? static_cast<uint16_t>((k >> 8) | (k << 8)) : static_cast<uint16_t>(k);
                memcpy(d, &v, 2);
              }
            }
          }
          return true;
        };
        JXL_RETURN_IF_ERROR(RunOnPool(fd->pool_, 0, static_cast<uint32_t>(dim.ysize), ThreadPool::NoInit, row, "jxlhip extra channel"));
      }
      if (alpha_in_main) {
        JXL_RETURN_IF_ERROR(check(jxlhip_modular_extra_channel_f32(mtree.t, static_cast<uint32_t>(alpha_ec),
                                                                   md.extra_channel_info[alpha_ec].bit_depth.bits_per_sample,
                                                                   md.bit_depth.bits_per_sample, alpha.data(), dim.xsize),
                                  "alpha samples"));
        JXL_RETURN_IF_ERROR(check(jxlhip_set_alpha(ctx, alpha.data(), dim.xsize), "set_alpha"));
      }
    }
  }
  t_entropy = now();
  const size_t sample_bytes = sample_type == JXLHIP_SAMPLE_F32 ? 4 : (sample_type == JXLHIP_SAMPLE_U8 ? 1 : 2);
  if (!to_callback && !grey_out) {
    JXL_RETURN_IF_ERROR(check(jxlhip_decode_frame_host(ctx, mo.buffer, mo.stride, 0), "decode_frame"));
    t_decode = now();
  } else {
    // Row callback (JxlDecoderSetImageOutCallback / SetMultithreadedImageOutCallback, decode.cc:2655-2700): the
    // frame arrives in the context's pinned host frame, already in display orientation, and is handed out as
    // WriteToOutputStage does it (stage_write.cc:324-338,399-409,662-700): Init(num_threads, chunk) once, row
    // runs of at most kChunkSize = 1024 pixels from the pool's threads with their thread id, destroy at the end.
    // Grey outputs: sample 0 (and the alpha sample) of every RGB(A) pixel, into the buffer or a chunk for the callback.
    const void* frame = nullptr;
    size_t pitch = 0;
    JXL_RETURN_IF_ERROR(check(jxlhip_decode_frame_pinned(ctx, &frame, &pitch), "decode_frame"));
    t_decode = now();
    const bool transposed = static_cast<uint32_t>(ds->undo_orientation) >= 5;
    const size_t ow = transposed ? dim.ysize : dim.xsize, oh = transposed ? dim.xsize : dim.ysize;
    const size_t dev_px = dev_nc * sample_bytes, out_px = out_nc * sample_bytes;
    constexpr size_t kChunk = 1024;
    auto to_grey = [&](const uint8_t* src, size_t n, uint8_t* dst) {
      for (size_t i = 0; i < n; i++) {
        memcpy(dst + i * out_px, src + i * dev_px, sample_bytes);
        if (out_nc == 2) memcpy(dst + i * out_px + sample_bytes, src + i * dev_px + 3 * sample_bytes, sample_bytes);
      }
    };
    if (!to_callback) {
      if (mo.stride < ow * out_px || mo.buffer_size < (oh - 1) * mo.stride + ow * out_px) return JXL_FAILURE("image out buffer too small");
      const auto row = [&](uint32_t y, size_t /*thread*/) -> Status {
        to_grey(static_cast<const uint8_t*>(frame) + static_cast<size_t>(y) * pitch, ow,
                static_cast<uint8_t*>(mo.buffer) + static_cast<size_t>(y) * mo.stride);
        return true;
      };
      JXL_RETURN_IF_ERROR(RunOnPool(fd->pool_, 0, static_cast<uint32_t>(oh), ThreadPool::NoInit, row, "jxlhip grey rows"));
    } else {
      void* run_opaque = nullptr;
      const PixelCallback& cb = mo.callback;
      std::vector<std::vector<uint8_t>> chunk;
      const auto init = [&](size_t num_threads) -> Status {
        run_opaque = cb.Init(num_threads, kChunk);
        if (grey_out) chunk.assign(num_threads ? num_threads : 1, std::vector<uint8_t>(kChunk * out_px));
        return run_opaque != nullptr;
      };
      const auto row = [&](uint32_t y, size_t thread) -> Status {
        const uint8_t* src = static_cast<const uint8_t*>(frame) + static_cast<size_t>(y) * pitch;
        for (size_t x = 0; x < ow; x += kChunk) {
          const size_t n = std::min(kChunk, ow - x);
          if (grey_out) {
            to_grey(src + x * dev_px, n, chunk[thread].data());
            cb.run(run_opaque, thread, x, y, n, chunk[thread].data());
          } else {
            cb.run(run_opaque, thread, x, y, n, src + x * dev_px);
          }
        }
        return true;
      };
      const Status ok = RunOnPool(fd->pool_, 0, static_cast<uint32_t>(oh), init, row, "jxlhip rows");
      if (run_opaque) cb.destroy(run_opaque);
      JXL_RETURN_IF_ERROR(ok);
    }
  }
  if (verbose)
    fprintf(stderr, "jxlhip seam: frame %zux%zu decoded on the HIP back-end (%s, %u-bit sample type %u, %u channels); ms: "
                    "side info %.2f, AC global + entropy decode + uploads %.2f, kernels + copy out %.2f, row callbacks %.2f; "
                    "before the seam: libjxl's DC global + DC groups in this ProcessSections call %.2f, the caller + headers "
                    "since the previous frame left %.2f\n",
            static_cast<size_t>(dim.xsize), static_cast<size_t>(dim.ysize), to_callback ? "callback" : "buffer", bits,
            sample_type, out_nc, t_side - t_begin, t_entropy - t_side, t_decode - t_entropy, now() - t_decode,
            t_begin - g_sections_begin.load(), g_last_frame_end.load() > 0 ? g_sections_begin.load() - g_last_frame_end.load() : 0.0);
  g_last_frame_end.store(now());
  for (size_t g = 0; g < dim.num_groups; g++) {
    fd->decoded_passes_per_ac_group_[g] = static_cast<uint8_t>(np);
    for (size_t ps = 0; ps < np && !single; ps++) section_status[ac_group_sec[g][ps]] = FrameDecoder::SectionStatus::kDone;
  }
  // The frame's pixels are out.  A Modular image the reference kept whole (extra channels of a frame that fits one
  // group are coded globally: ModularFrameDecoder::use_full_image) would be rendered by FinalizeFrame ->
  // FinalizeDecoding (dec_modular.cc:739-790) through the CPU pipeline -- whose colour buffers were never filled --
  // and handed to the output a second time.
  fd->modular_frame_decoder_.use_full_image = false;
  g_frames.fetch_add(1);
  *done = true;
  return true;
}

}  // namespace jxl
