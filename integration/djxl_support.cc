// TEST INFRASTRUCTURE ONLY (linked into oracle/_ref/djxl_ref and djxl_hip by integration/build_djxl.py; never part of
// the product).  The two symbols tools/djxl_main.cc + lib/extras need from the FULL libjxl that a decoder-only
// library (lib/jxl/decode.cc over the decoder translation units) does not carry:
//
//  * JxlEncoderInitBasicInfo -- extras::PackedPixelFile's constructor presets its JxlBasicInfo with it
//    (lib/extras/packed_image.cc:181; the function lives in the encoder API, lib/jxl/encode.cc:1248-1275).  djxl
//    overwrites the whole struct with JxlDecoderGetBasicInfo before it is read (lib/extras/dec/jxl.cc:329-345),
//    so only the documented defaults matter: an 8-bit, 3-channel image in identity orientation, 10 ticks/s.
//  * JxlGetDefaultCms -- only when lcms2 is not installed (then `djxl --color_space=...` has no CMS; the plain
//    decode never asks for one).  With lcms2 the reference's own lib/jxl/cms/jxl_cms.cc is compiled instead.
#include <jxl/cms_interface.h>
#include <jxl/codestream_header.h>

#include <cstring>

extern "C" {

__attribute__((visibility("default"))) void JxlEncoderInitBasicInfo(JxlBasicInfo* info) {
  memset(info, 0, sizeof(*info));
  info->bits_per_sample = 8;
  info->orientation = JXL_ORIENT_IDENTITY;
  info->num_color_channels = 3;
  info->animation.tps_numerator = 10;
  info->animation.tps_denominator = 1;
}

#if !DJXL_SUPPORT_HAVE_LCMS
__attribute__((visibility("default"))) const JxlCmsInterface* JxlGetDefaultCms() { return nullptr; }
#endif

}  // extern "C"
