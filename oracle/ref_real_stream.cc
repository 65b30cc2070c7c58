// ref_real_stream.cc -- TEST INFRASTRUCTURE (oracle/_ref).  A whole libjxl round trip
// on natural-looking content, with taps at the product's C-ABI boundary:
//   1. a procedural float image -> the reference ENCODER (EncodeFrame: adaptive
//      quantization, AC-strategy search, chroma-from-luma, EPF sharpness, coefficient
//      orders, ANS) -> a genuine VarDCT codestream;
//   2. the reference DECODER (FrameDecoder) decodes it to linear float RGB;
//   3. everything the back-end's boundary takes is dumped from the decoder's state
//      (PassesSharedState / PassesDecoderState / FrameHeader) together with the byte
//      ranges of the AC-global and AC-group sections (TOC).
// The product then entropy-decodes the REAL AC sections (include/jxl_hip_entropy.h) and
// renders them (C oracle on the CPU, HIP kernels on the GPU); the pixels must match
// step 2.  Nothing here is shipped or measured.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "lib/jxl/ac_strategy.h"
#include "lib/jxl/base/status.h"
#include "lib/jxl/chroma_from_luma.h"
#include "lib/jxl/color_encoding_internal.h"
#include "lib/jxl/dec_bit_reader.h"
#include "lib/jxl/dec_cache.h"
#include "lib/jxl/dec_frame.h"
#include "lib/jxl/enc_aux_out.h"
#include "lib/jxl/enc_bit_writer.h"
#include "lib/jxl/enc_context_map.h"
#include "lib/jxl/enc_fields.h"
#include "lib/jxl/enc_frame.h"
#include "lib/jxl/enc_params.h"
#include "lib/jxl/enc_ans.h"
#include "lib/jxl/frame_header.h"
#include "lib/jxl/icc_codec.h"
#include "lib/jxl/icc_codec_common.h"
#include "lib/jxl/image.h"
#include "lib/jxl/image_bundle.h"
#include "lib/jxl/image_metadata.h"
#include "lib/jxl/memory_manager_internal.h"
#include "lib/jxl/passes_state.h"
#include "lib/jxl/quantizer.h"
#include "lib/jxl/splines.h"

#include <string>

#include "jxl_oracle.h"

#include <brotli/encode.h>

#include <jxl/cms.h>

#include "lib/jxl/enc_icc_codec.h"
#include "lib/jxl/jpeg/enc_jpeg_data.h"
#include "lib/jxl/jpeg/jpeg_data.h"

#define JXR_EXPORT extern "C" __attribute__((visibility("default")))

// Link-time stand-ins for code paths the oracle never takes (see hwy_shim/brotli/encode.h;
// JPEG transcoding lives in lib/jxl/jpeg/, which is not part of this build).
extern "C" {
BrotliEncoderState* BrotliEncoderCreateInstance(brotli_alloc_func, brotli_free_func, void*) { return nullptr; }
void BrotliEncoderDestroyInstance(BrotliEncoderState*) {}
BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState*, BrotliEncoderParameter, uint32_t) { return BROTLI_FALSE; }
BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState*, BrotliEncoderOperation, size_t*, const uint8_t**,
                                        size_t*, uint8_t**, size_t*) { return BROTLI_FALSE; }
BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState*) { return BROTLI_TRUE; }
size_t BrotliEncoderMaxCompressedSize(size_t n) { return n + 1024; }
}
extern "C" const JxlCmsInterface* JxlGetDefaultCms() { return nullptr; }
namespace jxl {
namespace jpeg {
Status EncodeJPEGData(JxlMemoryManager*, JPEGData&, std::vector<uint8_t>*, const CompressParams&) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
Status SetColorEncodingFromJpegData(const jpeg::JPEGData&, ColorEncoding*) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
StatusOr<std::unique_ptr<JPEGData>> ParseJPG(JxlMemoryManager*, Bytes) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
Status SetBlobsFromJpegData(const jpeg::JPEGData&, Blobs*) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
Status SetChromaSubsamplingFromJpegData(const JPEGData&, YCbCrChromaSubsampling*) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
Status SetColorTransformFromJpegData(const JPEGData&, ColorTransform*) {
  return JXL_FAILURE("JPEG transcoding is not part of the oracle build");
}
}  // namespace jpeg
}  // namespace jxl

namespace {
using namespace jxl;  // NOLINT

// one frame's worth of results, owned by the handle
struct RealCase {
  std::vector<uint8_t> codestream;
  size_t frame_offset = 0;    // first byte of the frame (header + TOC + sections)
  size_t sections_offset = 0; // first byte of the first section
  size_t toc_bit_offset = 0;  // bits from the frame's first byte to the TOC (= size of the frame header)
  std::vector<uint64_t> section_offset, section_size;  // indexed by logical section id
  uint32_t xsize = 0, ysize = 0, num_groups = 0, num_dc_groups = 0, num_histograms = 0, used_acs = 0;
  uint32_t num_passes = 1, shift[kMaxNumPasses] = {};  // frame_header.passes
  jxlhip_frame_params params{};
  std::vector<uint8_t> acs, sharp, quant_dc, bctx_bytes;
  std::vector<int32_t> raw_quant;
  std::vector<int8_t> ytox, ytob;
  std::vector<float> dc[3];
  std::vector<float> rgb;     // reference decoder output, interleaved linear RGB
  std::vector<float> alpha;   // ... and its alpha channel (JXR_ALPHA streams)
  std::vector<float> extra;   // JXR_EXTRA streams: the other extra channels' planes, one behind the other
  std::vector<float> dequant; // the frame's DequantMatrices table (JXLHIP_DEQUANT_TABLE_FLOATS)
};

// dark gradient + textured disc + noise patch + sharp edges (after test_image.cc's GetSomeTestImage)
void FillImage(Image3F* img, uint32_t seed) {
  const size_t xs = img->xsize(), ys = img->ysize();
  uint32_t s = seed * 2654435761u + 12345u;
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (s & 0xffffff) / 16777216.0f;
  };
  float c0[3], c1[3];
  for (int c = 0; c < 3; c++) {
    c0[c] = 0.02f + 0.25f * rnd();
    c1[c] = 0.15f + 0.6f * rnd();
  }
  const float cx = xs * (0.3f + 0.4f * rnd()), cy = ys * (0.3f + 0.4f * rnd());
  const float rad = 0.28f * std::min(xs, ys);
  // broadband texture (amplitude ~ 1/f, 14 oriented components) whose strength varies slowly
  // over the image: leaves smooth areas, mid-frequency areas and busy areas, like a photograph
  constexpr int kWaves = 14;
  float wfx[kWaves], wfy[kWaves], wph[kWaves], wamp[kWaves], wcol[kWaves][3];
  for (int k = 0; k < kWaves; k++) {
    const float freq = 0.015f * std::pow(1.45f, static_cast<float>(k));  // 0.015 .. 1.9 rad/px
    const float ang = 6.2831853f * rnd();
    wfx[k] = freq * std::cos(ang);
    wfy[k] = freq * std::sin(ang);
    wph[k] = 6.2831853f * rnd();
    wamp[k] = 0.035f / std::sqrt(freq / 0.015f);
    for (int c = 0; c < 3; c++) wcol[k][c] = 0.6f + 0.8f * rnd();
  }
  const float mfx = 6.2831853f / (0.37f * xs + 80.0f), mfy = 6.2831853f / (0.29f * ys + 60.0f);
  for (size_t y = 0; y < ys; y++) {
    float* rows[3] = {img->PlaneRow(0, y), img->PlaneRow(1, y), img->PlaneRow(2, y)};
    for (size_t x = 0; x < xs; x++) {
      const float t = static_cast<float>(y) / ys, u = static_cast<float>(x) / xs;
      float v[3];
      for (int c = 0; c < 3; c++) v[c] = c0[c] + (c1[c] - c0[c]) * (0.7f * t + 0.3f * u);
      {
        float strength = 0.5f + 0.5f * std::sin(mfx * x + 0.7f) * std::cos(mfy * y);
        strength = strength * strength * 1.6f;
        for (int k = 0; k < kWaves; k++) {
          const float w = strength * wamp[k] * std::sin(wfx[k] * x + wfy[k] * y + wph[k]);
          for (int c = 0; c < 3; c++) v[c] += w * wcol[k][c];
        }
        // sensor-like noise, stronger in the dark (shot noise), so that no area is perfectly smooth
        const float grain = 0.012f + 0.02f * strength;
        for (int c = 0; c < 3; c++) v[c] = std::max(v[c] + grain * (rnd() - 0.5f), 0.002f);
      }
      const float dx = x - cx, dy = y - cy;
      if (dx * dx + dy * dy < rad * rad) {  // textured disc
        const float tex = 0.5f + 0.5f * std::sin(0.11f * x) * std::cos(0.07f * y + 0.013f * x);
        v[0] = 0.55f * tex + 0.1f;
        v[1] = 0.35f + 0.3f * tex * t;
        v[2] = 0.2f + 0.5f * (1.0f - tex);
      }
      if (x > xs * 0.62f && x < xs * 0.93f && y > ys * 0.12f && y < ys * 0.38f) {  // noise patch
        for (int c = 0; c < 3; c++) v[c] = 0.15f + 0.5f * rnd();
      }
      if ((x / 24 + y / 24) % 7 == 0 && y > ys * 0.7f) {  // checker edges
        for (int c = 0; c < 3; c++) v[c] = 0.9f - v[c] * 0.5f;
      }
      for (int c = 0; c < 3; c++) rows[c][x] = v[c];
    }
  }
}

Status Run(uint32_t xs, uint32_t ys, uint32_t seed, float distance, int speed_tier, int epf, int progressive,
           RealCase* out) {
  JxlMemoryManager mm;
  JXL_RETURN_IF_ERROR(MemoryManagerInit(&mm, nullptr));
  // ---- 1. encode
  CodecMetadata metadata;
  metadata.m.SetFloat32Samples();
  metadata.m.xyb_encoded = true;
  metadata.m.color_encoding = ColorEncoding::LinearSRGB(/*is_gray=*/false);
  // JXR_ORIGINAL=srgb8 | srgb16: the stream DESCRIBES an 8- / 16-bit sRGB original (what cjxl writes for a PNG), so
  // that JxlDecoder's default output is sRGB-encoded integer samples (tests/test_djxl.py); the pixels handed to the
  // encoder stay linear either way (no CMS in this build)
  if (const char* e = getenv("JXR_ORIGINAL")) {
    if (!strcmp(e, "srgb8") || !strcmp(e, "srgb16") || !strcmp(e, "srgb10") || !strcmp(e, "srgb12")) {  // (bit depth of the original)
      metadata.m.SetUintSamples(static_cast<uint32_t>(atoi(e + 4)));
      metadata.m.color_encoding = ColorEncoding::SRGB(/*is_gray=*/false);
    }
    // other enumerated originals: the decoder adapts the inverse opsin matrix to their primaries / white point
    // (OutputEncodingInfo::SetColorEncoding) and applies their transfer function
    ColorEncoding c;
    c.SetColorSpace(ColorSpace::kRGB);
    if (!strcmp(e, "p3")) {  // Display P3
      JXL_RETURN_IF_ERROR(c.SetWhitePointType(WhitePoint::kD65));
      JXL_RETURN_IF_ERROR(c.SetPrimariesType(Primaries::kP3));
      c.Tf().SetTransferFunction(TransferFunction::kSRGB);
      metadata.m.color_encoding = c;
    } else if (!strcmp(e, "rec2100pq")) {  // HDR10-like, 1000 nits
      JXL_RETURN_IF_ERROR(c.SetWhitePointType(WhitePoint::kD65));
      JXL_RETURN_IF_ERROR(c.SetPrimariesType(Primaries::k2100));
      c.Tf().SetTransferFunction(TransferFunction::kPQ);
      metadata.m.color_encoding = c;
      metadata.m.SetIntensityTarget(1000.0f);
    } else if (!strcmp(e, "customxy")) {  // Adobe-RGB-like primaries, a D50 white point, gamma 2.2
      JxlColorEncoding ext = {};
      ext.color_space = JXL_COLOR_SPACE_RGB;
      ext.white_point = JXL_WHITE_POINT_CUSTOM;
      ext.white_point_xy[0] = 0.3457, ext.white_point_xy[1] = 0.3585;
      ext.primaries = JXL_PRIMARIES_CUSTOM;
      ext.primaries_red_xy[0] = 0.64, ext.primaries_red_xy[1] = 0.33;
      ext.primaries_green_xy[0] = 0.21, ext.primaries_green_xy[1] = 0.71;
      ext.primaries_blue_xy[0] = 0.15, ext.primaries_blue_xy[1] = 0.06;
      ext.transfer_function = JXL_TRANSFER_FUNCTION_GAMMA;
      ext.gamma = 1.0 / 2.2;
      ext.rendering_intent = JXL_RENDERING_INTENT_RELATIVE;
      JXL_RETURN_IF_ERROR(c.FromExternal(ext));
      metadata.m.color_encoding = c;
    }
  }
  // JXR_ORIGINAL=gray8: a grey original (8-bit, sRGB transfer function): the pixels handed over are grey as well
  const bool gray = getenv("JXR_ORIGINAL") && !strcmp(getenv("JXR_ORIGINAL"), "gray8");
  if (gray) {
    metadata.m.SetUintSamples(8);
    metadata.m.color_encoding = ColorEncoding::SRGB(/*is_gray=*/true);
  }
  // JXR_ICC_FILE=path: the original carries that ICC profile (bytes taken as they are, no CMS to parse them): the
  // stream gets want_icc + the coded profile, and the decoder below -- like JxlDecoder without a CMS -- produces linear
  // sRGB (grey with JXR_ORIGINAL=gray8), SetFromMetadata dec_xyb.cc:160-164
  bool have_icc = false;
  if (const char* path = getenv("JXR_ICC_FILE")) {
    FILE* f = fopen(path, "rb");
    if (!f) return JXL_FAILURE("JXR_ICC_FILE");
    IccBytes icc;
    uint8_t buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) icc.insert(icc.end(), buf, buf + got);
    fclose(f);
    if (icc.empty()) return JXL_FAILURE("JXR_ICC_FILE is empty");
    ColorEncoding c;
    c.SetColorSpace(gray ? ColorSpace::kGray : ColorSpace::kRGB);
    c.SetICCRaw(std::move(icc));
    metadata.m.color_encoding = c;
    have_icc = true;
  }
  const ColorEncoding c_pixels = ColorEncoding::LinearSRGB(/*is_gray=*/gray);
  JXL_RETURN_IF_ERROR(metadata.size.Set(xs, ys));
  // JXR_ALPHA=8 | 16: an alpha channel of that many bits (the encoder codes it losslessly in the frame's Modular
  // sub-bitstream, like cjxl does for an RGBA PNG): soft-edged disc, a ramp, a hard-edged box, a noisy band
  uint32_t alpha_bits = 0;
  if (const char* e = getenv("JXR_ALPHA")) alpha_bits = static_cast<uint32_t>(atoi(e));
  if (alpha_bits) metadata.m.SetAlphaBits(alpha_bits);
  // JXR_EXTRA=n (1..3): n more extra channels behind the alpha channel (depth 16 bit, thermal 8 bit, optional 12 bit),
  // coded losslessly like it; the decoder below hands them out through extra-channel buffers
  uint32_t num_more = 0;
  if (const char* e = getenv("JXR_EXTRA")) num_more = std::min(3, std::max(0, atoi(e)));
  static const ExtraChannel kMoreType[3] = {ExtraChannel::kDepth, ExtraChannel::kThermal, ExtraChannel::kOptional};
  static const uint32_t kMoreBits[3] = {16, 8, 12};
  for (uint32_t i = 0; i < num_more; i++) {
    ExtraChannelInfo eci;
    eci.type = kMoreType[i];
    eci.bit_depth.bits_per_sample = kMoreBits[i];
    eci.bit_depth.exponent_bits_per_sample = 0;
    eci.bit_depth.floating_point_sample = false;
    eci.dim_shift = 0;
    metadata.m.extra_channel_info.push_back(eci);
  }
  metadata.m.num_extra_channels = static_cast<uint32_t>(metadata.m.extra_channel_info.size());
  // JXR_ORIENTATION=2..8: ImageMetadata::orientation of the written stream (tests of undo_orientation); the
  // FrameDecoder run below keeps the coded orientation, JxlDecoder (tests/test_seam.py) undoes it
  if (const char* e = getenv("JXR_ORIENTATION")) {
    const int o = atoi(e);
    if (o >= 1 && o <= 8) metadata.m.orientation = static_cast<uint32_t>(o);
  }
  // The ENCODER works from a copy of the metadata that says "linear sRGB": with the original's own primaries it would
  // call the CMS (absent from this build) to bring the pixels there.  The frame's bytes depend on the metadata only
  // through xyb_encoded / the intensity target / the extra channels, which the copy shares; the headers written to
  // the stream and the decoder below carry the original's colour encoding.
  CodecMetadata metadata_enc = metadata;
  if (have_icc || !(metadata.m.color_encoding.GetPrimariesType() == Primaries::kSRGB &&
                    metadata.m.color_encoding.GetWhitePointType() == WhitePoint::kD65))
    metadata_enc.m.color_encoding = ColorEncoding::LinearSRGB(/*is_gray=*/gray);
  ImageBundle ib(&mm, &metadata_enc.m);
  {
    JXL_ASSIGN_OR_RETURN(Image3F img, Image3F::Create(&mm, xs, ys));
    FillImage(&img, seed);
    if (gray) {
      for (size_t y = 0; y < ys; y++) {
        float* r = img.PlaneRow(0, y);
        float* g = img.PlaneRow(1, y);
        float* b = img.PlaneRow(2, y);
        for (size_t x = 0; x < xs; x++) r[x] = g[x] = b[x] = 0.2126f * r[x] + 0.7152f * g[x] + 0.0722f * b[x];
      }
    }
    JXL_RETURN_IF_ERROR(ib.SetFromImage(std::move(img), c_pixels));
  }
  if (alpha_bits) {
    JXL_ASSIGN_OR_RETURN(ImageF alpha, ImageF::Create(&mm, xs, ys));
    uint32_t s = seed * 747796405u + 2891336453u;
    const float levels = static_cast<float>((1u << alpha_bits) - 1);
    // JXR_ALPHA_LEVELS=n: only n distinct alpha values (a cut-out mask has 2): the encoder then codes the channel
    // through a palette transform
    const char* lv = getenv("JXR_ALPHA_LEVELS");
    const float coarse = lv && atoi(lv) > 1 ? static_cast<float>(atoi(lv) - 1) : 0.0f;
    for (size_t y = 0; y < ys; y++) {
      float* row = alpha.Row(y);
      for (size_t x = 0; x < xs; x++) {
        const float dx = (x - 0.55f * xs) / (0.35f * xs), dy = (y - 0.45f * ys) / (0.4f * ys);
        float a = 1.2f - std::sqrt(dx * dx + dy * dy);                   // soft-edged disc
        a = std::min(1.0f, std::max(0.0f, a * 2.0f));
        if (y < ys / 6) a = static_cast<float>(x) / xs;                    // ramp
        if (x > xs * 0.1f && x < xs * 0.25f && y > ys * 0.5f && y < ys * 0.8f) a = 0.0f;  // hard-edged hole
        if (y > ys * 0.9f) {                                               // noise
          s = s * 1664525u + 1013904223u;
          a = (s >> 8) / 16777216.0f;
        }
        if (coarse != 0.0f) a = std::floor(a * coarse + 0.5f) / coarse;
        row[x] = std::floor(a * levels + 0.5f) / levels;
      }
    }
    JXL_RETURN_IF_ERROR(ib.SetAlpha(std::move(alpha)));
  }
  if (num_more) {
    std::vector<ImageF> all;
    if (alpha_bits) {
      JXL_ASSIGN_OR_RETURN(ImageF a, ImageF::Create(&mm, xs, ys));
      JXL_RETURN_IF_ERROR(CopyImageTo(*ib.alpha(), &a));
      all.emplace_back(std::move(a));
    }
    for (uint32_t i = 0; i < num_more; i++) {
      JXL_ASSIGN_OR_RETURN(ImageF pl, ImageF::Create(&mm, xs, ys));
      const float levels = static_cast<float>((1u << kMoreBits[i]) - 1);
      uint32_t s2 = seed * 2654435761u + i * 977u + 1u;
      for (size_t y = 0; y < ys; y++) {
        float* row = pl.Row(y);
        for (size_t x = 0; x < xs; x++) {
          float v;
          if (i == 0) {  // depth: a tilted plane with a step
            v = 0.15f + 0.6f * (static_cast<float>(x) / xs) * (static_cast<float>(y) / ys) + (x > xs / 2 ? 0.2f : 0.0f);
          } else if (i == 1) {  // blobs
            v = 0.5f + 0.5f * std::sin(x * 0.031f) * std::cos(y * 0.047f);
          } else {  // stripes + a little noise
            s2 = s2 * 1664525u + 1013904223u;
            v = ((x / 16 + y / 24) & 1 ? 0.75f : 0.25f) + ((s2 >> 24) / 255.0f - 0.5f) * 0.02f;
          }
          row[x] = std::floor(std::min(1.0f, std::max(0.0f, v)) * levels + 0.5f) / levels;
        }
      }
      all.emplace_back(std::move(pl));
    }
    JXL_RETURN_IF_ERROR(ib.SetExtraChannels(std::move(all)));
  }
  CompressParams cparams;
  cparams.butteraugli_distance = distance;
  cparams.speed_tier = static_cast<SpeedTier>(speed_tier);
  cparams.patches = Override::kOff;
  cparams.dots = Override::kOff;
  cparams.noise = Override::kOff;
  cparams.epf = epf;  // -1 = the encoder's choice
  // progressive: 1 = AC passes by frequency band, 2 = by quantisation (shifted passes)
  if (progressive == 1) cparams.progressive_mode = Override::kOn;
  if (progressive == 2) cparams.qprogressive_mode = Override::kOn;
  cparams.color_transform = ColorTransform::kXYB;
  JXL_RETURN_IF_ERROR(ParamsPostInit(&cparams));
  BitWriter writer{&mm};
  JXL_RETURN_IF_ERROR(WriteCodestreamHeaders(&metadata, &writer, nullptr));
  if (have_icc)  // encode.cc:817-821
    JXL_RETURN_IF_ERROR(WriteICC(Bytes(metadata.m.color_encoding.ICC()), &writer, LayerType::Header, nullptr));
  JXL_RETURN_IF_ERROR(writer.WithMaxBits(8, LayerType::Header, nullptr, [&] {
    writer.ZeroPadToByte();
    return true;
  }));
  out->frame_offset = writer.BitsWritten() / 8;
  FrameInfo info;
  info.is_last = true;
  JxlCmsInterface no_cms{};  // never used: the input already is linear sRGB
  if (getenv("JXR_TRACE")) fprintf(stderr, "before EncodeFrame\n");
  JXL_RETURN_IF_ERROR(EncodeFrame(&mm, cparams, info, &metadata_enc, ib, no_cms, nullptr, &writer, nullptr));
  if (getenv("JXR_TRACE")) fprintf(stderr, "after EncodeFrame\n");
  {
    PaddedBytes bytes = std::move(writer).TakeBytes();
    out->codestream.assign(bytes.data(), bytes.data() + bytes.size());
  }

  // ---- 2. decode with the reference's FrameDecoder (the body of jxl::DecodeFrame, dec_frame.cc:82-133)
  auto dec_state = jxl::make_unique<PassesDecoderState>(&mm);
  JXL_RETURN_IF_ERROR(dec_state->output_encoding_info.SetFromMetadata(metadata));
  const uint32_t nch = alpha_bits ? 4 : 3;
  out->rgb.assign(static_cast<size_t>(xs) * ys * nch, 0.0f);
  ImageBundle decoded(&mm, &metadata.m);
  FrameDecoder fd(dec_state.get(), metadata, nullptr, /*use_slow_rendering_pipeline=*/false);
  const uint8_t* in = out->codestream.data() + out->frame_offset;
  const size_t avail = out->codestream.size() - out->frame_offset;
  BitReader reader(Bytes(in, avail));
  if (getenv("JXR_TRACE")) fprintf(stderr, "before InitFrame\n");
  JXL_RETURN_IF_ERROR(fd.InitFrame(&reader, &decoded, false));
  JXL_RETURN_IF_ERROR(fd.InitFrameOutput());
  // PassesDecoderState::Init (from InitFrameOutput) clears main_output: set it now, as decode.cc:1470 does
  JXL_RETURN_IF_ERROR(fd.SetImageOutput(PixelCallback(), out->rgb.data(), out->rgb.size() * sizeof(float), xs, ys,
                                        JxlPixelFormat{nch, JXL_TYPE_FLOAT, JXL_NATIVE_ENDIAN, 0}, 32,
                                        /*unpremul_alpha=*/false, /*undo_orientation=*/false));
  if (num_more) {  // one entry per extra channel, in order (decode.cc:1477-1490); the alpha channel rides in the main output
    out->extra.assign(static_cast<size_t>(xs) * ys * num_more, 0.0f);
    const JxlPixelFormat one{1, JXL_TYPE_FLOAT, JXL_NATIVE_ENDIAN, 0};
    if (alpha_bits) JXL_RETURN_IF_ERROR(fd.AddExtraChannelOutput(nullptr, 0, xs, one, 32));
    for (uint32_t i = 0; i < num_more; i++)
      JXL_RETURN_IF_ERROR(fd.AddExtraChannelOutput(out->extra.data() + static_cast<size_t>(xs) * ys * i,
                                                   static_cast<size_t>(xs) * ys * sizeof(float), xs, one, 32));
  }
  const size_t header_bytes = reader.TotalBitsConsumed() / kBitsPerByte;
  JXL_RETURN_IF_ERROR(reader.Close());
  out->sections_offset = out->frame_offset + header_bytes;
  {
    BitReader br2(Bytes(in, avail));
    FrameHeader fh2(&metadata);
    JXL_RETURN_IF_ERROR(ReadFrameHeader(&br2, &fh2));
    out->toc_bit_offset = br2.TotalBitsConsumed();
    JXL_RETURN_IF_ERROR(br2.Close());
  }
  const FrameHeader& fh = fd.GetFrameHeader();
  const FrameDimensions fdim = fh.ToFrameDimensions();
  out->num_groups = fdim.num_groups;
  out->num_dc_groups = fdim.num_dc_groups;
  const size_t num_sections = fd.Toc().size();
  out->section_offset.assign(num_sections, 0);
  out->section_size.assign(num_sections, 0);
  {
    Status close_ok = true;
    std::vector<std::unique_ptr<BitReader>> readers;
    std::vector<std::unique_ptr<BitReaderScopedCloser>> closers;
    std::vector<FrameDecoder::SectionInfo> infos;
    size_t pos = header_bytes, index = 0;
    for (auto e : fd.Toc()) {
      JXL_RETURN_IF_ERROR(pos + e.size <= avail);
      out->section_offset[e.id] = out->frame_offset + pos;
      out->section_size[e.id] = e.size;
      auto br = make_unique<BitReader>(Bytes(in + pos, e.size));
      infos.emplace_back(FrameDecoder::SectionInfo{br.get(), e.id, index++});
      closers.emplace_back(make_unique<BitReaderScopedCloser>(*br, close_ok));
      readers.emplace_back(std::move(br));
      pos += e.size;
    }
    std::vector<FrameDecoder::SectionStatus> status(infos.size());
    JXL_RETURN_IF_ERROR(fd.ProcessSections(infos.data(), infos.size(), status.data()));
    for (auto st : status) JXL_RETURN_IF_ERROR(st == FrameDecoder::kDone);
    closers.clear();
    JXL_RETURN_IF_ERROR(close_ok);
  }
  if (getenv("JXR_TRACE")) fprintf(stderr, "before FinalizeFrame\n");
  JXL_RETURN_IF_ERROR(fd.FinalizeFrame());
  if (alpha_bits) {  // split the interleaved RGBA
    out->alpha.resize(static_cast<size_t>(xs) * ys);
    std::vector<float> rgb(static_cast<size_t>(xs) * ys * 3);
    for (size_t i = 0; i < static_cast<size_t>(xs) * ys; i++) {
      for (int c = 0; c < 3; c++) rgb[i * 3 + c] = out->rgb[i * 4 + c];
      out->alpha[i] = out->rgb[i * 4 + 3];
    }
    out->rgb.swap(rgb);
  }

  // ---- 3. the inputs of the product's boundary, from the decoder's state
  if (fh.encoding != FrameEncoding::kVarDCT || fh.upsampling != 1 ||
      !fh.chroma_subsampling.Is444() || (fh.flags & (FrameHeader::kNoise | FrameHeader::kPatches |
                                                    FrameHeader::kSplines | FrameHeader::kUseDcFrame))) {
    return JXL_FAILURE("frame uses features outside the back-end's scope");
  }
  const PassesSharedState& sh = dec_state->shared_storage;
  const size_t xsb = fdim.xsize_blocks, ysb = fdim.ysize_blocks;
  out->xsize = xs;
  out->ysize = ys;
  out->num_histograms = sh.num_histograms;
  out->num_passes = fh.passes.num_passes;
  for (uint32_t i = 0; i < fh.passes.num_passes; i++) out->shift[i] = fh.passes.shift[i];
  out->used_acs = dec_state->used_acs;
  out->acs.resize(xsb * ysb);
  out->sharp.resize(xsb * ysb);
  out->quant_dc.resize(xsb * ysb);
  out->raw_quant.resize(xsb * ysb);
  for (int c = 0; c < 3; c++) out->dc[c].resize(xsb * ysb);
  for (size_t by = 0; by < ysb; by++) {
    AcStrategyRow arow = sh.ac_strategy.ConstRow(by);
    const int32_t* q = sh.raw_quant_field.ConstRow(by);
    const uint8_t* sp = sh.epf_sharpness.ConstRow(by);
    const uint8_t* qdc = sh.quant_dc.ConstRow(by);
    for (size_t bx = 0; bx < xsb; bx++) {
      const AcStrategy a = arow[bx];
      out->acs[by * xsb + bx] = static_cast<uint8_t>((a.RawStrategy() << 1) | (a.IsFirstBlock() ? 1 : 0));
      out->raw_quant[by * xsb + bx] = q[bx];
      out->sharp[by * xsb + bx] = sp[bx];
      out->quant_dc[by * xsb + bx] = qdc[bx];
    }
    for (int c = 0; c < 3; c++) {
      memcpy(out->dc[c].data() + by * xsb, sh.dc->ConstPlaneRow(c, by), xsb * sizeof(float));
    }
  }
  const size_t xst = DivCeil(xsb, kColorTileDimInBlocks), yst = DivCeil(ysb, kColorTileDimInBlocks);
  out->ytox.resize(xst * yst);
  out->ytob.resize(xst * yst);
  for (size_t ty = 0; ty < yst; ty++) {
    memcpy(out->ytox.data() + ty * xst, sh.cmap.ytox_map.ConstRow(ty), xst);
    memcpy(out->ytob.data() + ty * xst, sh.cmap.ytob_map.ConstRow(ty), xst);
  }
  {
    BitWriter bw{&mm};
    JXL_RETURN_IF_ERROR(EncodeBlockCtxMap(sh.block_ctx_map, &bw, nullptr));
    bw.ZeroPadToByte();
    Span<const uint8_t> sp = bw.GetSpan();
    out->bctx_bytes.assign(sp.data(), sp.data() + sp.size());
  }
  {
    // table_ is one contiguous block starting at the DCT8 X matrix (quant_weights.cc:1247-1262);
    // FrameDecoder has computed every matrix in used_acs, the rest of the block is never read
    JXL_RETURN_IF_ERROR(const_cast<DequantMatrices&>(sh.matrices).EnsureComputed(&mm, ~0u));
    out->dequant.resize(JXLHIP_DEQUANT_TABLE_FLOATS);
    memcpy(out->dequant.data(), sh.matrices.Matrix(AcStrategyType::DCT, 0), sizeof(float) * JXLHIP_DEQUANT_TABLE_FLOATS);
  }
  jxlhip_frame_params& p = out->params;
  memset(&p, 0, sizeof(p));
  p.xsize = xs;
  p.ysize = ys;
  p.coeff_type = JXLHIP_COEFF_I16;  // the test checks jxlhip_ac_pass_max_num_bits() < 16
  p.output_kind = JXLHIP_OUT_LINEAR_RGB_F32;
  p.used_acs = dec_state->used_acs;
  const QuantizerParams qp = sh.quantizer.GetParams();
  p.global_scale = qp.global_scale;
  p.quant_dc = qp.quant_dc;
  p.x_dm_multiplier = dec_state->x_dm_multiplier;
  p.b_dm_multiplier = dec_state->b_dm_multiplier;
  const OpsinParams& op = dec_state->output_encoding_info.opsin_params;
  for (int i = 0; i < 4; i++) p.quant_biases[i] = op.quant_biases[i];
  for (int i = 0; i < 3; i++) p.opsin_biases[i] = op.opsin_biases[i];
  for (int i = 0; i < 9; i++) p.inverse_opsin_matrix[i] = op.inverse_opsin_matrix[i * 4];
  p.cfl_base_x = sh.cmap.base().GetBaseCorrelationX();
  p.cfl_base_b = sh.cmap.base().GetBaseCorrelationB();
  p.cfl_color_factor = static_cast<uint32_t>(sh.cmap.base().GetColorFactor());
  const LoopFilter& lf = fh.loop_filter;
  p.lf.gab = lf.gab ? 1 : 0;
  p.lf.gab_weights[0] = lf.gab_x_weight1;
  p.lf.gab_weights[1] = lf.gab_x_weight2;
  p.lf.gab_weights[2] = lf.gab_y_weight1;
  p.lf.gab_weights[3] = lf.gab_y_weight2;
  p.lf.gab_weights[4] = lf.gab_b_weight1;
  p.lf.gab_weights[5] = lf.gab_b_weight2;
  p.lf.epf_iters = lf.epf_iters;
  for (int i = 0; i < 8; i++) p.lf.epf_sharp_lut[i] = lf.epf_sharp_lut[i];
  for (int i = 0; i < 3; i++) p.lf.epf_channel_scale[i] = lf.epf_channel_scale[i];
  p.lf.epf_quant_mul = lf.epf_quant_mul;
  p.lf.epf_pass0_sigma_scale = lf.epf_pass0_sigma_scale;
  p.lf.epf_pass2_sigma_scale = lf.epf_pass2_sigma_scale;
  p.lf.epf_border_sad_mul = lf.epf_border_sad_mul;
  return true;
}

}  // namespace

// ---- streams with the features the product's seam DECLINES (round 5, tests/test_djxl.py) ------------------------------
// PreparePipeline adds noise / patches / splines stages when the frame header asks (dec_cache.cc:124,193-200); Modular
// frames, multi-frame files and progressive passes take other paths again.  The reference ENCODER writes one small stream
// per feature here; the GPU suite runs djxl_ref and djxl_hip on them with a device present.
//   feature: "noise" (photon noise, ISO 6400), "splines" (two hand-made splines), "patches" (screenshot-like content: a
//   flat page with repeated glyphs -> a reference frame + patches), "modular" (a lossless Modular frame), "animation" (two
//   VarDCT frames, 10 ticks per second), "progressive" (AC passes by frequency band), "plain" (none: the control)
namespace {
void FillPage(Image3F* img, uint32_t seed) {  // a flat page with rows of repeated 6x9 dot-matrix glyphs
  const size_t xs = img->xsize(), ys = img->ysize();
  uint32_t s = seed * 747796405u + 2891336453u;
  uint64_t glyph[6];
  for (auto& g : glyph) {
    s = s * 1664525u + 1013904223u;
    const uint64_t a = s;
    s = s * 1664525u + 1013904223u;
    g = (a << 32) | s;
  }
  for (size_t y = 0; y < ys; y++) {
    float* rows[3] = {img->PlaneRow(0, y), img->PlaneRow(1, y), img->PlaneRow(2, y)};
    for (size_t x = 0; x < xs; x++) {
      float v = 0.85f;
      const size_t cx = x / 10, cy = y / 16, ix = x % 10, iy = y % 16;
      if (cx >= 2 && cx + 2 < xs / 10 && cy >= 1 && cy + 1 < ys / 16 && ix < 6 && iy >= 3 && iy < 12 && (cy % 3) != 2) {
        const uint64_t g = glyph[(cx * 7 + cy * 3) % 6];
        if ((g >> ((iy - 3) * 6 + ix)) & 1u) v = 0.05f;
      }
      for (int c = 0; c < 3; c++) rows[c][x] = v;
    }
  }
}

Status FeatureStream(uint32_t xs, uint32_t ys, uint32_t seed, float distance, const char* feature,
                     std::vector<uint8_t>* out) {
  const std::string f = feature;
  JxlMemoryManager mm;
  JXL_RETURN_IF_ERROR(MemoryManagerInit(&mm, nullptr));
  const bool modular = f == "modular";
  const bool animation = f == "animation";
  CodecMetadata metadata;
  metadata.m.SetUintSamples(8);  // an 8-bit sRGB original, what cjxl writes for a PNG: djxl's default output is then a PPM
  metadata.m.xyb_encoded = !modular;
  metadata.m.color_encoding = ColorEncoding::SRGB(/*is_gray=*/false);
  JXL_RETURN_IF_ERROR(metadata.size.Set(xs, ys));
  if (animation) {
    metadata.m.have_animation = true;
    metadata.m.animation.tps_numerator = 10;
    metadata.m.animation.tps_denominator = 1;
    metadata.m.animation.num_loops = 0;
    metadata.m.animation.have_timecodes = false;
  }
  // the lossy encoder is handed linear pixels under a "linear sRGB" copy of the metadata (no CMS in this build, as in
  // Run above); the lossless one takes the 8-bit samples as they are, in the original's own encoding
  CodecMetadata metadata_enc = metadata;
  if (!modular) metadata_enc.m.color_encoding = ColorEncoding::LinearSRGB(false);
  CompressParams cparams;
  cparams.butteraugli_distance = distance;
  cparams.speed_tier = SpeedTier::kSquirrel;
  cparams.patches = Override::kOff;
  cparams.dots = Override::kOff;
  cparams.noise = Override::kOff;
  cparams.color_transform = ColorTransform::kXYB;
  std::vector<QuantizedSpline> qsplines;
  std::vector<Spline::Point> starts;
  if (f == "noise") cparams.photon_noise_iso = 6400.0f;
  if (f == "patches") cparams.patches = Override::kOn;
  if (f == "progressive") cparams.progressive_mode = Override::kOn;
  if (f == "splines") {
    for (int k = 0; k < 2; k++) {
      Spline sp;
      for (int i = 0; i < 5; i++)
        sp.control_points.emplace_back(xs * (0.1f + 0.2f * i), ys * (k ? 0.25f + 0.1f * ((i * 3) % 4) : 0.8f - 0.12f * ((i * 2) % 5)));
      for (auto& d : sp.color_dct) d.fill(0.0f);
      sp.sigma_dct.fill(0.0f);
      sp.color_dct[1][0] = k ? 0.35f : 0.2f;   // Y
      sp.color_dct[0][0] = k ? 0.01f : -0.02f;  // X
      sp.color_dct[2][0] = k ? 0.1f : 0.25f;   // B
      sp.color_dct[1][1] = 0.05f;
      sp.sigma_dct[0] = k ? 4.5f : 3.0f;
      sp.sigma_dct[1] = 0.5f;
      JXL_ASSIGN_OR_RETURN(QuantizedSpline q, QuantizedSpline::Create(sp, /*quantization_adjustment=*/0, 0.0f, 1.0f));
      qsplines.push_back(std::move(q));
      starts.push_back(sp.control_points[0]);
    }
    cparams.custom_splines.splines = Span<const QuantizedSpline>(qsplines.data(), qsplines.size());
    cparams.custom_splines.starting_points = Span<const Spline::Point>(starts.data(), starts.size());
  }
  if (modular) {
    cparams.SetLossless();
  }
  JXL_RETURN_IF_ERROR(ParamsPostInit(&cparams));
  BitWriter writer{&mm};
  JXL_RETURN_IF_ERROR(WriteCodestreamHeaders(&metadata, &writer, nullptr));
  JXL_RETURN_IF_ERROR(writer.WithMaxBits(8, LayerType::Header, nullptr, [&] {
    writer.ZeroPadToByte();
    return true;
  }));
  JxlCmsInterface no_cms{};
  const int frames = animation ? 2 : 1;
  for (int fi = 0; fi < frames; fi++) {
    ImageBundle ib(&mm, &metadata_enc.m);
    JXL_ASSIGN_OR_RETURN(Image3F img, Image3F::Create(&mm, xs, ys));
    if (f == "patches") FillPage(&img, seed);
    else FillImage(&img, seed + 31u * fi);
    if (modular) {  // 8-bit samples of an sRGB original: quantise, so that "lossless" has integers to keep
      for (int c = 0; c < 3; c++)
        for (size_t y = 0; y < ys; y++) {
          float* r = img.PlaneRow(c, y);
          for (size_t x = 0; x < xs; x++) r[x] = std::floor(std::min(1.0f, std::max(0.0f, r[x])) * 255.0f + 0.5f) / 255.0f;
        }
    }
    JXL_RETURN_IF_ERROR(ib.SetFromImage(std::move(img), modular ? ColorEncoding::SRGB(false) : ColorEncoding::LinearSRGB(false)));
    FrameInfo info;
    info.is_last = fi + 1 == frames;
    if (animation) info.duration = 1 + fi;
    JXL_RETURN_IF_ERROR(EncodeFrame(&mm, cparams, info, &metadata_enc, ib, no_cms, nullptr, &writer, nullptr));
  }
  PaddedBytes bytes = std::move(writer).TakeBytes();
  out->assign(bytes.data(), bytes.data() + bytes.size());
  return true;
}
}  // namespace

// -> bytes written (0 = failed, or `cap` too small: call again with the returned size... the streams are < 1 MB)
JXR_EXPORT size_t jxr_feature_stream(uint32_t xs, uint32_t ys, uint32_t seed, float distance, const char* feature,
                                     uint8_t* out, size_t cap) {
  std::vector<uint8_t> bytes;
  if (!FeatureStream(xs, ys, seed, distance, feature, &bytes)) return 0;
  if (bytes.size() > cap) return 0;
  memcpy(out, bytes.data(), bytes.size());
  return bytes.size();
}

JXR_EXPORT void* jxr_real_case_create(uint32_t xs, uint32_t ys, uint32_t seed, float distance, int speed_tier,
                                      int epf, int progressive) {
  auto c = std::make_unique<RealCase>();
  if (!Run(xs, ys, seed, distance, speed_tier, epf, progressive, c.get())) return nullptr;
  return c.release();
}
JXR_EXPORT void jxr_real_case_destroy(void* h) { delete static_cast<RealCase*>(h); }

// what = 0 codestream, 1 acs, 2 raw_quant, 3 sharpness, 4 ytox, 5 ytob, 6..8 dc x/y/b, 9 quant_dc,
// 10 block-ctx-map bytes, 11 rgb, 12 section offsets (u64), 13 section sizes (u64), 14 frame params,
// 15 dequant table, 16 alpha (JXR_ALPHA), 17 the other extra channels (JXR_EXTRA)
JXR_EXPORT const void* jxr_real_case_data(void* h, int what, size_t* bytes) {
  RealCase* c = static_cast<RealCase*>(h);
  auto ret = [&](const void* p, size_t n) {
    *bytes = n;
    return p;
  };
  switch (what) {
    case 0: return ret(c->codestream.data(), c->codestream.size());
    case 1: return ret(c->acs.data(), c->acs.size());
    case 2: return ret(c->raw_quant.data(), c->raw_quant.size() * 4);
    case 3: return ret(c->sharp.data(), c->sharp.size());
    case 4: return ret(c->ytox.data(), c->ytox.size());
    case 5: return ret(c->ytob.data(), c->ytob.size());
    case 6: case 7: case 8: return ret(c->dc[what - 6].data(), c->dc[what - 6].size() * 4);
    case 9: return ret(c->quant_dc.data(), c->quant_dc.size());
    case 10: return ret(c->bctx_bytes.data(), c->bctx_bytes.size());
    case 11: return ret(c->rgb.data(), c->rgb.size() * 4);
    case 12: return ret(c->section_offset.data(), c->section_offset.size() * 8);
    case 13: return ret(c->section_size.data(), c->section_size.size() * 8);
    case 14: return ret(&c->params, sizeof(c->params));
    case 15: return ret(c->dequant.data(), c->dequant.size() * 4);
    case 16: return ret(c->alpha.data(), c->alpha.size() * 4);
    case 17: return ret(c->extra.data(), c->extra.size() * 4);
    default: *bytes = 0; return static_cast<const void*>(nullptr);
  }
}
// 0 num_groups, 1 num_dc_groups, 2 num_histograms, 3 used_acs, 4 frame offset, 5 sections offset,
// 6 num_passes, 16+i shift of pass i
JXR_EXPORT uint64_t jxr_real_case_info(void* h, int what) {
  RealCase* c = static_cast<RealCase*>(h);
  switch (what) {
    case 0: return c->num_groups;
    case 1: return c->num_dc_groups;
    case 2: return c->num_histograms;
    case 3: return c->used_acs;
    case 4: return c->frame_offset;
    case 5: return c->sections_offset;
    case 6: return c->num_passes;
    case 7: return c->toc_bit_offset;
    case 16: case 17: case 18: case 19: case 20: case 21: case 22: case 23: case 24: case 25: case 26:
      return c->shift[what - 16];
    default: return 0;
  }
}

// ---- ICC differential harness (tests/test_icc.py): an ARBITRARY "predicted profile" byte string entropy-coded the way
// WriteICC codes PredictICC's output (enc_icc_codec.cc:455-481; prefix codes like libjxl, or ANS; LZ77 or not), then
// the reference's ICCReader over the result.  Returns -1 = the harness itself failed, 0 = ICCReader rejects the stream,
// 1 = accepted (*icc_size bytes in icc).  *bits = where ICCReader stopped reading.
JXR_EXPORT int jxr_icc_stream(const uint8_t* enc, size_t n, int use_ans, int lz77, uint8_t* stream, size_t stream_cap,
                              size_t* stream_size, uint8_t* icc, size_t icc_cap, size_t* icc_size, size_t* bits) {
  using namespace jxl;
  JxlMemoryManager mm;
  if (!MemoryManagerInit(&mm, nullptr)) return -1;
  BitWriter writer{&mm};
  auto write = [&]() -> Status {
    std::vector<std::vector<Token>> tokens(1);
    JXL_RETURN_IF_ERROR(writer.WithMaxBits(128, LayerType::Header, nullptr, [&] { return U64Coder::Write(n, &writer); }));
    for (size_t i = 0; i < n; i++)
      tokens[0].emplace_back(static_cast<uint32_t>(ICCANSContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0)), enc[i]);
    HistogramParams params;
    params.lz77_method = lz77 == 0   ? HistogramParams::LZ77Method::kNone
                         : lz77 == 1 ? HistogramParams::LZ77Method::kOptc256
                                     : HistogramParams::LZ77Method::kLZ77b3w3f;
    params.force_huffman = !use_ans;
    EntropyEncodingData code;
    JXL_ASSIGN_OR_RETURN(size_t cost, BuildAndEncodeHistograms(&mm, params, kNumICCContexts, tokens, &code, &writer,
                                                               LayerType::Header, nullptr));
    (void)cost;
    JXL_RETURN_IF_ERROR(WriteTokens(tokens[0], code, 0, &writer, LayerType::Header, nullptr));
    return writer.WithMaxBits(8, LayerType::Header, nullptr, [&] {
      writer.ZeroPadToByte();
      return true;
    });
  };
  if (!write()) return -1;
  PaddedBytes bytes = std::move(writer).TakeBytes();
  *stream_size = bytes.size();
  if (bytes.size() > stream_cap) return -1;
  memcpy(stream, bytes.data(), bytes.size());
  BitReader reader(Bytes(bytes.data(), bytes.size()));
  ICCReader icc_reader(&mm);
  PaddedBytes profile{&mm};
  Status ok = icc_reader.Init(&reader);
  if (ok) ok = icc_reader.Process(&reader, &profile);
  *bits = reader.TotalBitsConsumed();
  (void)reader.Close();
  *icc_size = 0;
  if (!ok || profile.empty()) return 0;  // (an empty profile: decode.cc:1124)
  *icc_size = profile.size();
  if (profile.size() > icc_cap) return -1;
  memcpy(icc, profile.data(), profile.size());
  return 1;
}
