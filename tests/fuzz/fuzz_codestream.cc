// fuzz_codestream.cc -- sanitizer harness for the host front-end as a chain (libjxl_amd/csrc/
// entropy.cc + modular.inc compiled INTO this binary with -fsanitize=address,undefined): damaged
// copies of genuine codestreams go through jxlhip_image_header_decode (-> jxlhip_icc_decode) -> jxlhip_frame_header_decode
// -> jxlhip_toc_decode -> jxlhip_dc_global_decode -> jxlhip_modular_global_decode ->
// jxlhip_dc_group_decode, every buffer an exact-size heap block.  Prints "<ok> <rejected>".
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/jxl_hip_entropy.h"
#include "../../include/jxl_hip_frame.h"

namespace {
struct Rng {
  uint64_t s;
  uint32_t Next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (uint32_t)(s >> 11);
  }
  uint32_t Below(uint32_t n) { return n ? Next() % n : 0; }
};
typedef std::vector<uint8_t> Bytes;

struct Exact {  // an exact-size heap copy: one byte past the end is an ASAN report
  uint8_t* p;
  size_t n;
  Exact(const uint8_t* d, size_t len) : p((uint8_t*)malloc(len ? len : 1)), n(len) {
    if (len) memcpy(p, d, len);
  }
  ~Exact() { free(p); }
};

bool Chain(const Bytes& cs) {
  Exact all(cs.data(), cs.size());
  jxlhip_image_header ih;
  jxlhip_extra_channel ec[4];
  size_t pos = 0;
  if (jxlhip_image_header_decode(all.p, all.n, &pos, ec, 4, &ih) != JXLHIP_OK) return false;
  if (ih.color_encoding.want_icc) {  // the coded ICC profile: once for its size, once into an exact-size block
    size_t icc_size = 0, p2 = pos;
    if (jxlhip_icc_decode(all.p, all.n, &pos, nullptr, 0, &icc_size) != JXLHIP_OK) return false;
    uint8_t* profile = (uint8_t*)malloc(icc_size);
    const int rc = jxlhip_icc_decode(all.p, all.n, &p2, profile, icc_size, &icc_size);
    free(profile);
    if (rc != JXLHIP_OK || p2 != pos) abort();
  }
  uint8_t shifts[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < ih.num_extra_channels && i < 4; i++) shifts[i] = (uint8_t)ec[i].dim_shift;
  if (ih.num_extra_channels > 4) return false;
  jxlhip_image_info info = {ih.xsize, ih.ysize, ih.xyb_encoded, ih.num_extra_channels, shifts,
                            ih.have_animation, ih.have_timecodes, 0, ih.bit_depth.bits_per_sample};
  jxlhip_frame_header fh;
  if (jxlhip_frame_header_decode(all.p, all.n, &pos, &info, &fh) != JXLHIP_OK) return false;
  if (fh.num_toc_entries == 0 || fh.num_toc_entries > 4096) return false;
  if ((uint64_t)fh.xsize_blocks * fh.ysize_blocks > (1u << 18)) return false;  // bound the fuzzer's allocations
  const uint32_t nt = (uint32_t)fh.num_toc_entries;
  std::vector<uint64_t> off(nt);
  std::vector<uint32_t> sz(nt);
  uint64_t total = 0;
  if (jxlhip_toc_decode(all.p, all.n, &pos, nt, off.data(), sz.data(), &total) != JXLHIP_OK) return false;
  const size_t start = pos / 8;
  auto section = [&](uint32_t i, const uint8_t** p, size_t* n) {
    if (i >= nt || start + off[i] > all.n || sz[i] > all.n - start - off[i]) return false;
    *p = all.p + start + off[i];
    *n = sz[i];
    return true;
  };
  const uint8_t* d;
  size_t n;
  if (nt == 1 || !section(0, &d, &n)) return false;
  Exact s0(d, n);
  jxlhip_dc_global dcg;
  size_t dpos = 0;
  if (jxlhip_dc_global_decode(s0.p, s0.n, &dpos, fh.flags, &dcg) != JXLHIP_OK) return false;
  jxlhip_modular_tree* tree = nullptr;
  if (jxlhip_modular_global_decode(s0.p, s0.n, &dpos, &fh, &tree) != JXLHIP_OK) return false;
  const size_t nb = (size_t)fh.xsize_blocks * fh.ysize_blocks;
  const size_t nc = (size_t)((fh.xsize_blocks + 7) / 8) * ((fh.ysize_blocks + 7) / 8);
  std::vector<int32_t> q0(nb), q1(nb), q2(nb), rq(nb);
  std::vector<uint8_t> acs(nb), sharp(nb);
  std::vector<int8_t> ytox(nc), ytob(nc);
  int32_t* q[3] = {q0.data(), q1.data(), q2.data()};
  bool ok = true;
  uint32_t used = 0;
  for (uint32_t g = 0; ok && g < fh.num_dc_groups; g++) {
    if (!section(1 + g, &d, &n)) {
      ok = false;
      break;
    }
    Exact sg(d, n);
    size_t gp = 0;
    uint32_t prec = 0;
    ok = jxlhip_dc_group_decode(tree, sg.p, sg.n, &gp, &fh, g, q, &prec, acs.data(), rq.data(), sharp.data(),
                                ytox.data(), ytob.data(), &used) == JXLHIP_OK;
  }
  // extra channels: what follows the coefficients is out of reach here (the AC groups are not entropy-decoded), but the
  // collecting form on the groups' sections from bit 0 and the final undo (squeeze, palettes) must hold up on whatever
  // state the damaged stream left
  if (ok && fh.num_extra_channels && tree && jxlhip_modular_groups_are_final(tree)) {
    // the form jxlhip_decode_codestream takes by default: every group converts its samples on the spot (pending
    // single-channel palettes through their table) and writes them into exact-size planes -- ASAN sees an overrun
    const uint32_t first_ac = 2 + (uint32_t)fh.num_dc_groups;
    const uint32_t ne = fh.num_extra_channels < 4 ? fh.num_extra_channels : 4;
    std::vector<std::vector<float>> store(ne, std::vector<float>((size_t)fh.xsize * fh.ysize));
    float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t strides[4] = {fh.xsize, fh.xsize, fh.xsize, fh.xsize};
    uint32_t bits[4] = {8, 8, 8, 8};
    for (uint32_t e = 0; e < ne; e++) planes[e] = store[e].data();
    for (uint32_t g = 0; g < fh.num_groups && g < 4; g++) {
      if (!section(first_ac + g, &d, &n)) break;
      Exact sg(d, n);
      size_t gp = 0;
      (void)jxlhip_modular_ac_group_decode_f32_strided(tree, &fh, g, 0, sg.p, sg.n, &gp, bits, ih.bit_depth.bits_per_sample,
                                                       planes, strides);
    }
    (void)jxlhip_modular_finalize(tree, nullptr, nullptr);
    for (uint32_t e = 0; e < ne; e++)
      (void)jxlhip_modular_extra_channel_rows_f32(tree, e, 8, ih.bit_depth.bits_per_sample, 0, fh.ysize, store[e].data(), fh.xsize);
  } else if (ok && fh.num_extra_channels && tree) {
    const uint32_t first_ac = 2 + (uint32_t)fh.num_dc_groups;
    for (uint32_t g = 0; g < fh.num_groups && g < 4; g++) {
      if (!section(first_ac + g, &d, &n)) break;
      Exact sg(d, n);
      size_t gp = 0;
      (void)jxlhip_modular_ac_group_decode(tree, &fh, g, 0, sg.p, sg.n, &gp);
    }
    std::vector<float> plane((size_t)fh.xsize * fh.ysize);
    for (uint32_t e = 0; e < fh.num_extra_channels; e++)
      (void)jxlhip_modular_extra_channel_f32(tree, e, 8, ih.bit_depth.bits_per_sample, plane.data(), fh.xsize);
  }
  jxlhip_modular_tree_destroy(tree);
  return ok;
}

void Damage(Rng* r, Bytes* b) {
  switch (r->Below(5)) {
    case 0:
      for (uint32_t i = 0, k = 1 + r->Below(4); i < k; i++) (*b)[r->Below(b->size())] ^= 1u << r->Below(8);
      break;
    case 1:
      for (uint32_t i = 0, k = 1 + r->Below(8); i < k; i++) (*b)[r->Below(b->size())] = (uint8_t)r->Next();
      break;
    case 2:
      b->resize(1 + r->Below(b->size()));
      break;
    case 3: {  // damage concentrated in the first sections (headers, DC global, DC groups)
      const uint32_t lim = (uint32_t)(b->size() < 600 ? b->size() : 600);
      for (uint32_t i = 0, k = 1 + r->Below(3); i < k; i++) (*b)[r->Below(lim)] ^= 1u << r->Below(8);
      break;
    }
    default: {
      const uint32_t a = r->Below(b->size()), len = 1 + r->Below(16);
      for (uint32_t i = a; i < a + len && i < b->size(); i++) (*b)[i] = (uint8_t)r->Next();
    }
  }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const uint64_t iters = strtoull(argv[1], nullptr, 10);
  Rng r{strtoull(argv[2], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1};
  std::vector<Bytes> cases;
  for (int i = 3; i < argc; i++) {
    FILE* f = fopen(argv[i], "rb");
    if (!f) return 2;
    Bytes b;
    uint8_t buf[4096];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + k);
    fclose(f);
    if (!Chain(b)) {
      fprintf(stderr, "undamaged codestream %s does not parse\n", argv[i]);
      return 1;
    }
    cases.push_back(b);
  }
  uint64_t ok = 0, rejected = 0;
  for (uint64_t i = 0; i < iters; i++) {
    Bytes b = cases[i % cases.size()];
    Damage(&r, &b);
    (Chain(b) ? ok : rejected)++;
  }
  printf("%llu %llu\n", (unsigned long long)ok, (unsigned long long)rejected);
  return 0;
}
