// fuzz_entropy.cc -- sanitizer harness for the host entropy decoder
// (libjxl_amd/csrc/entropy.cc, compiled INTO this binary with
// -fsanitize=address,undefined by tests/test_entropy_fuzz.py).
//
// Input: case files written by the test (genuine libjxl streams + side info).
// Every iteration damages the bytes and/or the side info in a seeded way and
// runs the whole host path on them: block context map, AC global (dequant
// encodings, coefficient orders, histograms), every pass of every group.  The
// decoder may return any status; it must not read or write out of bounds, hit
// undefined behaviour, hang or leak.  Prints "<ok> <rejected>" counts.
//
// Case file (little endian): magic "JXHF", u32 xsize, ysize, num_groups,
// num_passes, used_acs, shift[11], then blobs {u64 n, bytes}: block-ctx bytes,
// ac_strategy, raw_quant (i32), quant_dc, AC global, then num_passes*num_groups
// group sections (pass-major).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
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

struct Case {
  uint32_t xsize, ysize, num_groups, num_passes, used_acs, shift[11];
  Bytes bctx, acs, raw_quant, quant_dc, global;
  std::vector<Bytes> groups;
};

bool ReadBlob(FILE* f, Bytes* b) {
  uint64_t n;
  if (fread(&n, 8, 1, f) != 1 || n > (1u << 28)) return false;
  b->resize(n);
  return n == 0 || fread(b->data(), 1, n, f) == n;
}

bool Load(const char* path, Case* c) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char magic[4];
  bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "JXHF", 4) && fread(&c->xsize, 4, 1, f) == 1 &&
            fread(&c->ysize, 4, 1, f) == 1 && fread(&c->num_groups, 4, 1, f) == 1 &&
            fread(&c->num_passes, 4, 1, f) == 1 && fread(&c->used_acs, 4, 1, f) == 1 &&
            fread(c->shift, 4, 11, f) == 11 && ReadBlob(f, &c->bctx) && ReadBlob(f, &c->acs) &&
            ReadBlob(f, &c->raw_quant) && ReadBlob(f, &c->quant_dc) && ReadBlob(f, &c->global);
  if (ok) {
    c->groups.resize((size_t)c->num_passes * c->num_groups);
    for (auto& g : c->groups) ok = ok && ReadBlob(f, &g);
  }
  fclose(f);
  return ok;
}

void Damage(Rng* r, Bytes* b) {
  if (b->empty()) return;
  switch (r->Below(6)) {
    case 0:  // bit flips
      for (uint32_t i = 0, n = 1 + r->Below(4); i < n; i++) (*b)[r->Below(b->size())] ^= 1u << r->Below(8);
      break;
    case 1:  // random bytes
      for (uint32_t i = 0, n = 1 + r->Below(8); i < n; i++) (*b)[r->Below(b->size())] = (uint8_t)r->Next();
      break;
    case 2:  // truncate
      b->resize(r->Below(b->size()));
      break;
    case 3: {  // overwrite a run
      const size_t at = r->Below(b->size()), n = 1 + r->Below(32);
      const uint8_t v = r->Below(2) ? 0xFF : 0;
      for (size_t i = at; i < b->size() && i < at + n; i++) (*b)[i] = v;
      break;
    }
    case 4: {  // splice from elsewhere in the buffer
      const size_t from = r->Below(b->size()), to = r->Below(b->size()), n = 1 + r->Below(64);
      for (size_t i = 0; i < n && from + i < b->size() && to + i < b->size(); i++) (*b)[to + i] = (*b)[from + i];
      break;
    }
    default:  // drop the head (desynchronise)
      b->erase(b->begin(), b->begin() + r->Below(1 + b->size() / 4));
  }
}

// exact-size heap copies so that ASAN sees every overrun
struct Exact {
  explicit Exact(const Bytes& b) : n(b.size()), p((uint8_t*)malloc(n ? n : 1)) {
    if (n) memcpy(p, b.data(), n);
  }
  ~Exact() { free(p); }
  size_t n;
  uint8_t* p;
};

int RunOnce(const Case& base, Rng* r, uint64_t* ok, uint64_t* rejected) {
  Case c = base;
  const uint32_t what = r->Below(16);
  if (what < 6) Damage(r, &c.global);
  if (what >= 4 && what < 11 && !c.groups.empty()) Damage(r, &c.groups[r->Below(c.groups.size())]);
  if (what == 11) Damage(r, &c.bctx);
  if (what == 12) {  // side info damage keeps the sizes (they are frame-sized by contract)
    for (uint32_t i = 0, n = 1 + r->Below(8); i < n; i++) c.acs[r->Below(c.acs.size())] = (uint8_t)r->Next();
  }
  if (what == 13) {
    for (uint32_t i = 0, n = 1 + r->Below(8); i < n; i++) c.raw_quant[r->Below(c.raw_quant.size())] = (uint8_t)r->Next();
  }
  if (what == 14) {
    for (uint32_t i = 0, n = 1 + r->Below(8); i < n; i++) c.quant_dc[r->Below(c.quant_dc.size())] = (uint8_t)r->Next();
  }
  uint32_t used_acs = c.used_acs;
  if (what == 15) used_acs = r->Next() & 0x7FFFFFF;

  {  // the stand-alone parsers on whatever bytes the damaged AC-global section now holds
    Exact g(c.global);
    const uint32_t n = 1 + r->Below(300);
    std::vector<uint64_t> off(n);
    std::vector<uint32_t> sz(n);
    size_t tp = r->Below(16);
    uint64_t total = 0;
    (void)jxlhip_toc_decode(g.p, g.n, &tp, n, off.data(), sz.data(), &total);
    jxlhip_quant_encoding enc[JXLHIP_NUM_QUANT_TABLES];
    size_t dp = r->Below(64);
    (void)jxlhip_dequant_encodings_decode(g.p, g.n, &dp, enc);
    uint8_t shifts[8];
    for (auto& sh : shifts) sh = (uint8_t)r->Below(4);
    jxlhip_image_info im = {1 + r->Below(4000), 1 + r->Below(4000), r->Below(2), r->Below(9), shifts,
                            r->Below(2), r->Below(2), r->Below(4) == 0};
    jxlhip_frame_header fh;
    size_t hp = r->Below(64);
    (void)jxlhip_frame_header_decode(g.p, g.n, &hp, &im, &fh);
    jxlhip_dc_global dg;
    size_t gp = r->Below(64);
    (void)jxlhip_dc_global_decode(g.p, g.n, &gp, 0, &dg);
    Bytes hb = c.global;  // the same bytes behind a codestream signature: the image header parser
    if (hb.size() >= 2) {
      hb[0] = 0xFF;
      hb[1] = 0x0A;
    }
    Exact hx(hb);
    std::vector<jxlhip_extra_channel> ec(r->Below(5));
    jxlhip_image_header ih;
    size_t ip = 0;
    (void)jxlhip_image_header_decode(hx.p, hx.n, &ip, ec.empty() ? nullptr : ec.data(), ec.size(), &ih);
  }
  jxlhip_block_ctx_map bcm;
  size_t pos = 0;
  Exact bc(c.bctx);
  const jxlhip_block_ctx_map* bcmp = nullptr;
  if (jxlhip_block_ctx_map_decode(bc.p, bc.n, &pos, &bcm) == JXLHIP_OK) bcmp = &bcm;

  jxlhip_quant_encoding enc[JXLHIP_NUM_QUANT_TABLES];
  uint32_t num_histograms = 0;
  jxlhip_ac_pass* passes[11] = {};
  size_t bits = 0;
  Exact gl(c.global);
  int rc = jxlhip_ac_global_decode(gl.p, gl.n, c.num_groups, c.num_passes, used_acs, bcmp, enc, &num_histograms,
                                   passes, &bits);
  if (rc != JXLHIP_OK) {
    (*rejected)++;
    return 0;
  }
  const uint32_t xsb = (c.xsize + 7) / 8, ysb = (c.ysize + 7) / 8, xsg = (c.xsize + 255) / 256;
  const bool i32 = r->Below(4) == 0;
  const size_t esz = i32 ? 4 : 2;
  Exact acs(c.acs), rq(c.raw_quant), qdc(c.quant_dc);
  bool all_ok = true;
  for (uint32_t g = 0; g < c.num_groups; g++) {
    uint8_t* bufs[3];
    for (int ch = 0; ch < 3; ch++) bufs[ch] = (uint8_t*)calloc(JXLHIP_GROUP_COEFFS, esz);
    void* const ptrs[3] = {bufs[0], bufs[1], bufs[2]};
    for (uint32_t p = 0; p < c.num_passes; p++) {
      Exact sec(c.groups[(size_t)p * c.num_groups + g]);
      size_t gp = 0, n = 0;
      rc = jxlhip_ac_group_decode(passes[p], xsb, ysb, g % xsg, g / xsg, acs.p, (const int32_t*)rq.p, qdc.p, sec.p,
                                  sec.n, &gp, c.shift[p], i32 ? JXLHIP_COEFF_I32 : JXLHIP_COEFF_I16, ptrs, &n);
      if (rc != JXLHIP_OK) all_ok = false;
      if (rc == JXLHIP_OK && n > JXLHIP_GROUP_COEFFS) {
        fprintf(stderr, "ncoeffs %zu out of range\n", n);
        abort();
      }
    }
    for (int ch = 0; ch < 3; ch++) free(bufs[ch]);
  }
  for (uint32_t p = 0; p < c.num_passes; p++) jxlhip_ac_pass_destroy(passes[p]);
  (all_ok ? *ok : *rejected)++;
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <iterations> <seed> <case>...\n", argv[0]);
    return 2;
  }
  const uint64_t iters = strtoull(argv[1], nullptr, 10);
  Rng r{strtoull(argv[2], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1};
  std::vector<Case> cases;
  for (int i = 3; i < argc; i++) {
    Case c;
    if (!Load(argv[i], &c)) {
      fprintf(stderr, "cannot load %s\n", argv[i]);
      return 2;
    }
    cases.push_back(std::move(c));
  }
  uint64_t ok = 0, rejected = 0;
  // every case must decode before it is damaged
  for (const Case& c : cases) {
    jxlhip_block_ctx_map bcm;
    size_t pos = 0;
    const bool have = jxlhip_block_ctx_map_decode(c.bctx.data(), c.bctx.size(), &pos, &bcm) == JXLHIP_OK;
    jxlhip_quant_encoding enc[JXLHIP_NUM_QUANT_TABLES];
    uint32_t nh = 0;
    jxlhip_ac_pass* passes[11] = {};
    size_t bits = 0;
    if (jxlhip_ac_global_decode(c.global.data(), c.global.size(), c.num_groups, c.num_passes, c.used_acs,
                                have ? &bcm : nullptr, enc, &nh, passes, &bits) != JXLHIP_OK) {
      fprintf(stderr, "undamaged case does not decode\n");
      return 1;
    }
    for (uint32_t p = 0; p < c.num_passes; p++) jxlhip_ac_pass_destroy(passes[p]);
  }
  for (uint64_t i = 0; i < iters; i++) RunOnce(cases[i % cases.size()], &r, &ok, &rejected);
  printf("%llu %llu\n", (unsigned long long)ok, (unsigned long long)rejected);
  return 0;
}
