// entropy.cc -- the host half of the back-end's input side (include/jxl_hip_entropy.h):
// JPEG XL's entropy decoder for the AC coefficient stream, written for this
// library -- bit reader, prefix codes, ANS with alias tables, hybrid-uint
// tokens, LZ77, context maps, coefficient orders and the per-varblock token
// walk that fills the ACImage-layout coefficient buffers the HIP kernels read.
// Host C++ only; one thread per AC group (the JxlParallelRunner's workers).
//
// Behavioural specification: ISO/IEC 18181-1 annexes C (entropy coding) and
// I.3.5-I.3.7, as implemented by the reference at the lines cited per function.
#include "env_switches.h"
#include <stdlib.h>
namespace jxlhip_env {
Switches g;
void LoadLocked() {
  g.wp_general.store(getenv("JXLHIP_WP_GENERAL") != nullptr);
  g.codestream_verbose.store(getenv("JXLHIP_CODESTREAM_VERBOSE") != nullptr);
  g.no_pipeline.store(getenv("JXLHIP_NO_PIPELINE") != nullptr);
  const char* e = getenv("JXLHIP_TEST_RANGE_GROUP");
  g.test_range_group.store(e ? atoll(e) : -1);
  const char* ife = getenv("JXLHIP_MULTI_INTERIOR_FIRST");
  g.multi_interior_first.store(!ife || atoi(ife) != 0 ? 1 : 0);
  auto num = [](const char* name, int unset) {
    const char* v = getenv(name);
    return v ? atoi(v) : unset;
  };
  g.fused_pc_rh.store(num("JXLHIP_FUSED_PC_RH", 0));
  g.filter_rh.store(num("JXLHIP_FILTER_RH", 0));
  g.big_wgs.store(num("JXLHIP_BIG_WGS", Switches::kUnset));
  g.multi_force_gather.store(getenv("JXLHIP_MULTI_FORCE_GATHER") != nullptr);
  {
    // unparsable / empty / 0: the default (2^30 pixels = four 16K frames' worth), not "refuse every frame"
    const char* mp = getenv("JXLHIP_MAX_PIXELS");
    char* end = nullptr;
    const unsigned long long v = mp ? strtoull(mp, &end, 10) : 0ull;
    g.max_pixels.store((mp && end != mp && v != 0) ? v : (1ull << 30));
  }
  g.loaded.store(true, std::memory_order_release);
}
}  // namespace jxlhip_env
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/jxl_hip_entropy.h"
#include "../../include/jxl_hip_frame.h"

namespace {

constexpr int kOk = JXLHIP_OK;
constexpr int kBad = JXLHIP_ERR_BAD_STREAM;

// ---------------------------------------------------------------- bit reader
// LSB-first (lib/jxl/dec_bit_reader.h).  Reads past the end return zeros and
// are detected by Healthy().
class BitReader {
 public:
  BitReader(const uint8_t* data, size_t size, size_t bit_pos)
      : data_(data), size_(size), pos_(std::min(size, bit_pos / 8)) {
    if (bit_pos / 8 > size) overrun_ = true;
    Refill();
    Consume(bit_pos % 8);
  }
  inline void Refill() {
    if (size_ - pos_ >= 8 && pos_ <= size_) {
      uint64_t v;
      memcpy(&v, data_ + pos_, 8);
      buf_ |= v << nbits_;
      pos_ += (63 - nbits_) >> 3;
      nbits_ |= 56;
    } else {
      while (nbits_ <= 56) {
        const uint64_t b = pos_ < size_ ? data_[pos_] : 0;
        pos_++;
        buf_ |= b << nbits_;
        nbits_ += 8;
      }
    }
  }
  inline uint64_t Peek(uint32_t n) const { return buf_ & ((1ull << n) - 1ull); }
  inline void Consume(uint32_t n) {
    buf_ >>= n;
    nbits_ -= n;
  }
  // n <= 32
  inline uint32_t Read(uint32_t n) {
    Refill();
    const uint32_t v = (uint32_t)Peek(n);
    Consume(n);
    return v;
  }
  size_t BitsConsumed() const { return pos_ * 8 - nbits_; }
  // nothing beyond the last byte has been CONSUMED (peeking is fine)
  bool Healthy() const { return !overrun_ && BitsConsumed() <= size_ * 8; }

 private:
  const uint8_t* data_;
  size_t size_;
  size_t pos_;  // bytes loaded into buf_ (may run past size_: zeros)
  uint64_t buf_ = 0;
  uint32_t nbits_ = 0;
  bool overrun_ = false;
};

inline uint32_t FloorLog2(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
inline uint32_t CeilLog2(uint32_t v) { return v <= 1 ? 0 : FloorLog2(v - 1) + 1; }

// U32 with four (value | bits+offset) choices selected by two bits (fields.h U32Enc)
struct U32Dist {
  uint32_t bits[4];
  uint32_t offset[4];
};
uint32_t ReadU32(BitReader* br, const U32Dist& d) {
  const uint32_t sel = br->Read(2);
  return d.offset[sel] + (d.bits[sel] ? br->Read(d.bits[sel]) : 0);
}

// ------------------------------------------------------------- prefix codes
// Canonical prefix codes read LSB-first (RFC 7932 section 3.2-3.5 as used by
// JPEG XL; reference: dec_huffman.cc, huffman_table.cc).  Two-level table:
// 8 root bits, then one sub-table per root slot that has longer codes.
constexpr int kRootBits = 8;
constexpr int kMaxCodeLength = 15;

struct PrefixEntry {
  uint8_t bits;    // root: code length, or kRootBits + sub-table width; sub: length - kRootBits
  uint16_t value;  // symbol, or offset of the sub-table relative to this root slot
};

struct PrefixCode {
  std::vector<PrefixEntry> table;

  inline uint32_t ReadSymbol(BitReader* br) const {
    const PrefixEntry* e = &table[br->Peek(kRootBits)];
    uint32_t n = e->bits;
    if (n > kRootBits) {
      br->Consume(kRootBits);
      e += e->value + br->Peek(n - kRootBits);
    }
    br->Consume(e->bits);
    return e->value;
  }

  void SetSingle(uint16_t symbol) {
    table.assign(1u << kRootBits, PrefixEntry{0, symbol});
  }

  // lengths[i] in 0..15.  root_bits <= 8.  Returns false for an over- or
  // under-subscribed code (except the one-symbol code, which reads 0 bits).
  bool Build(const uint8_t* lengths, size_t n, int root_bits) {
    uint32_t count[kMaxCodeLength + 1] = {0};
    for (size_t i = 0; i < n; i++) count[lengths[i]]++;
    uint32_t used = 0;
    for (int l = 1; l <= kMaxCodeLength; l++) used += count[l];
    const size_t root_size = (size_t)1 << root_bits;
    if (used == 0) return false;
    if (used == 1) {
      for (size_t i = 0; i < n; i++)
        if (lengths[i]) {
          table.assign(root_size, PrefixEntry{0, (uint16_t)i});
          return true;
        }
    }
    // Kraft sum must be exactly 1
    uint32_t space = 1u << kMaxCodeLength;
    for (int l = 1; l <= kMaxCodeLength; l++) {
      const uint32_t need = count[l] << (kMaxCodeLength - l);
      if (need > space) return false;
      space -= need;
    }
    if (space != 0) return false;
    // canonical codes: increasing length, then symbol order
    uint32_t next_code[kMaxCodeLength + 2];
    {
      uint32_t code = 0;
      for (int l = 1; l <= kMaxCodeLength; l++) {
        next_code[l] = code;
        code = (code + count[l]) << 1;
      }
    }
    struct Sym {
      uint32_t rev;  // the code as it appears in the LSB-first stream
      uint8_t len;
      uint16_t symbol;
    };
    std::vector<Sym> syms;
    syms.reserve(used);
    for (size_t i = 0; i < n; i++) {
      const int l = lengths[i];
      if (!l) continue;
      uint32_t c = next_code[l]++, r = 0;
      for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
      syms.push_back(Sym{r, (uint8_t)l, (uint16_t)i});
    }
    // widest code hanging off each root slot
    std::vector<uint8_t> sub_bits(root_size, 0);
    for (const Sym& s : syms)
      if (s.len > root_bits) {
        uint8_t& w = sub_bits[s.rev & (root_size - 1)];
        w = std::max<uint8_t>(w, (uint8_t)(s.len - root_bits));
      }
    size_t total = root_size;
    std::vector<uint32_t> sub_at(root_size, 0);
    for (size_t r = 0; r < root_size; r++)
      if (sub_bits[r]) {
        sub_at[r] = (uint32_t)total;
        total += (size_t)1 << sub_bits[r];
      }
    table.assign(total, PrefixEntry{0, 0});
    for (size_t r = 0; r < root_size; r++)
      if (sub_bits[r]) {
        table[r].bits = (uint8_t)(root_bits + sub_bits[r]);
        table[r].value = (uint16_t)(sub_at[r] - r);
      }
    for (const Sym& s : syms) {
      if (s.len <= root_bits) {
        for (size_t k = s.rev; k < root_size; k += (size_t)1 << s.len)
          table[k] = PrefixEntry{s.len, s.symbol};
      } else {
        const size_t r = s.rev & (root_size - 1);
        const uint32_t sl = s.len - root_bits, w = sub_bits[r];
        for (size_t k = s.rev >> root_bits; k < ((size_t)1 << w); k += (size_t)1 << sl)
          table[sub_at[r] + k] = PrefixEntry{(uint8_t)sl, s.symbol};
      }
    }
    return true;
  }
};

// fixed code of the code-length code lengths (dec_huffman.cc:191-194): symbol, length, LSB-first pattern
struct FixedCode {
  uint8_t symbol, len, pattern;
};
constexpr FixedCode kCodeLengthLengthCode[6] = {{0, 2, 0}, {4, 2, 1}, {3, 2, 2}, {2, 3, 3}, {1, 4, 7}, {5, 4, 15}};
constexpr uint8_t kCodeLengthOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};

// HuffmanDecodingData::ReadFromBitStream (dec_huffman.cc:170-243)
bool ReadPrefixCode(BitReader* br, size_t alphabet_size, PrefixCode* out) {
  if (alphabet_size > (1u << kMaxCodeLength)) return false;
  const uint32_t kind = br->Read(2);
  if (kind == 1) {  // simple code: 1..4 symbols listed explicitly
    const uint32_t max_bits = alphabet_size > 1 ? FloorLog2((uint32_t)alphabet_size - 1) + 1 : 0;
    uint32_t n = br->Read(2) + 1;
    uint16_t s[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n; i++) {
      s[i] = (uint16_t)br->Read(max_bits);
      if (s[i] >= alphabet_size) return false;
    }
    for (uint32_t i = 0; i + 1 < n; i++)
      for (uint32_t j = i + 1; j < n; j++)
        if (s[i] == s[j]) return false;
    if (n == 4) n += br->Read(1);  // two shapes for four symbols
    std::vector<uint8_t> len(alphabet_size, 0);
    switch (n) {
      case 1: out->SetSingle(s[0]); return true;
      case 2: len[s[0]] = 1; len[s[1]] = 1; break;
      case 3: len[s[0]] = 1; len[s[1]] = 2; len[s[2]] = 2; break;
      case 4: len[s[0]] = len[s[1]] = len[s[2]] = len[s[3]] = 2; break;
      default: len[s[0]] = 1; len[s[1]] = 2; len[s[2]] = 3; len[s[3]] = 3; break;
    }
    return out->Build(len.data(), alphabet_size, kRootBits);
  }
  // code lengths coded with a code-length code whose own lengths use a fixed code
  uint8_t cl_len[18] = {0};
  int space = 32, num_codes = 0;
  for (size_t i = kind; i < 18 && space > 0; i++) {
    br->Refill();
    const uint32_t idx = (uint32_t)br->Peek(4);
    uint8_t v = 0;
    for (const FixedCode& c : kCodeLengthLengthCode)
      if ((idx & ((1u << c.len) - 1)) == c.pattern) {
        br->Consume(c.len);
        v = c.symbol;
        break;
      }
    cl_len[kCodeLengthOrder[i]] = v;
    if (v) {
      space -= 32 >> v;
      num_codes++;
    }
  }
  if (!(num_codes == 1 || space == 0)) return false;
  PrefixCode cl;
  if (!cl.Build(cl_len, 18, 5)) return false;
  // ReadHuffmanCodeLengths (dec_huffman.cc:29-96): run-length coded lengths
  std::vector<uint8_t> len(alphabet_size, 0);
  size_t symbol = 0;
  uint8_t prev_len = 8, repeat_len = 0;
  int repeat = 0;
  int left = 32768;
  while (symbol < alphabet_size && left > 0) {
    br->Refill();
    const PrefixEntry e = cl.table[br->Peek(5)];
    br->Consume(e.bits);
    const uint8_t code = (uint8_t)e.value;
    if (code < 16) {
      repeat = 0;
      len[symbol++] = code;
      if (code) {
        prev_len = code;
        left -= 32768 >> code;
      }
    } else {
      const int extra = code - 14;  // 16: repeat previous length (2 bits), 17: repeat zero (3 bits)
      const uint8_t new_len = code == 16 ? prev_len : 0;
      if (repeat_len != new_len) {
        repeat = 0;
        repeat_len = new_len;
      }
      const int old = repeat;
      if (repeat > 0) repeat = (repeat - 2) << extra;
      repeat += (int)br->Read(extra) + 3;
      const int delta = repeat - old;
      if (symbol + delta > alphabet_size) return false;
      memset(&len[symbol], repeat_len, delta);
      symbol += delta;
      if (repeat_len) left -= delta << (15 - repeat_len);
    }
  }
  if (left != 0) return false;
  return out->Build(len.data(), alphabet_size, kRootBits);
}

// ---------------------------------------------------------------------- ANS
constexpr uint32_t kAnsLogTab = 12, kAnsTab = 1u << kAnsLogTab;
constexpr uint32_t kAnsSignature = 0x13;

// alias table entry: slot i of the 2^log_alpha buckets covers `entry_size`
// states; the first `cutoff` belong to symbol i, the rest to `right_value`
struct AliasEntry {
  uint8_t cutoff;
  uint8_t right_value;
  uint16_t freq0;
  uint16_t offsets1;
  uint16_t freq1;
};

// InitAliasTable (ans_common.cc:41-146): the order in which over-full buckets
// donate to under-full ones is part of the format (the encoder mirrors it)
bool BuildAliasTable(std::vector<int32_t> dist, uint32_t log_alpha, AliasEntry* a) {
  const uint32_t table_size = 1u << log_alpha;
  while (!dist.empty() && dist.back() == 0) dist.pop_back();
  if (dist.empty()) dist.push_back((int32_t)kAnsTab);
  if (dist.size() > table_size) return false;
  const uint32_t entry_size = kAnsTab >> log_alpha;
  int single = -1;
  uint32_t sum = 0;
  for (size_t s = 0; s < dist.size(); s++) {
    sum += (uint32_t)dist[s];
    if (dist[s] == (int32_t)kAnsTab) single = (int)s;
  }
  if (sum != kAnsTab) return false;
  if (single >= 0) {  // state-preserving table for one-symbol distributions
    for (uint32_t i = 0; i < table_size; i++) {
      a[i].right_value = (uint8_t)single;
      a[i].cutoff = 0;
      a[i].offsets1 = (uint16_t)(entry_size * i);
      a[i].freq0 = 0;
      a[i].freq1 = (uint16_t)kAnsTab;
    }
    return true;
  }
  std::vector<uint32_t> under, over, cutoffs(table_size, 0);
  for (size_t i = 0; i < dist.size(); i++) {
    cutoffs[i] = (uint32_t)dist[i];
    if (cutoffs[i] > entry_size) over.push_back((uint32_t)i);
    else if (cutoffs[i] < entry_size) under.push_back((uint32_t)i);
  }
  for (uint32_t i = (uint32_t)dist.size(); i < table_size; i++) under.push_back(i);
  while (!over.empty()) {
    const uint32_t o = over.back();
    over.pop_back();
    if (under.empty()) return false;
    const uint32_t u = under.back();
    under.pop_back();
    const uint32_t by = entry_size - cutoffs[u];
    cutoffs[o] -= by;
    a[u].right_value = (uint8_t)o;
    a[u].offsets1 = (uint16_t)cutoffs[o];
    if (cutoffs[o] < entry_size) under.push_back(o);
    else if (cutoffs[o] > entry_size) over.push_back(o);
  }
  for (uint32_t i = 0; i < table_size; i++) {
    if (cutoffs[i] == entry_size) {
      a[i].right_value = (uint8_t)i;
      a[i].offsets1 = 0;
      a[i].cutoff = 0;
    } else {
      a[i].offsets1 = (uint16_t)(a[i].offsets1 - cutoffs[i]);
      a[i].cutoff = (uint8_t)cutoffs[i];
    }
    const uint32_t f0 = i < dist.size() ? (uint32_t)dist[i] : 0;
    const uint32_t r = a[i].right_value;
    const uint32_t f1 = r < dist.size() ? (uint32_t)dist[r] : 0;
    a[i].freq0 = (uint16_t)f0;
    a[i].freq1 = (uint16_t)f1;
  }
  return true;
}

uint32_t ReadVarLenU8(BitReader* br) {  // dec_ans.cc:33-44
  if (!br->Read(1)) return 0;
  const uint32_t n = br->Read(3);
  return n == 0 ? 1 : br->Read(n) + (1u << n);
}
uint32_t ReadVarLenU16(BitReader* br) {  // dec_ans.cc:47-58
  if (!br->Read(1)) return 0;
  const uint32_t n = br->Read(4);
  return n == 0 ? 1 : br->Read(n) + (1u << n);
}

// fixed code of the log-counts (dec_ans.cc:104-121): symbol, length, LSB-first pattern
constexpr FixedCode kLogCountCode[14] = {{10, 3, 0},  {7, 3, 2},  {6, 3, 4},  {8, 3, 5},   {9, 3, 6},
                                         {3, 4, 3},   {5, 4, 7},  {4, 4, 9},  {1, 4, 11},  {2, 4, 15},
                                         {0, 5, 17},  {11, 6, 33}, {12, 7, 1}, {13, 7, 65}};

// ReadHistogram (dec_ans.cc:60-201): a 12-bit ANS distribution
bool ReadDistribution(BitReader* br, std::vector<int32_t>* counts) {
  const int range = 1 << kAnsLogTab;
  if (br->Read(1)) {  // one or two symbols
    const uint32_t n = br->Read(1) + 1;
    uint32_t s[2] = {0, 0};
    uint32_t mx = 0;
    for (uint32_t i = 0; i < n; i++) {
      s[i] = ReadVarLenU8(br);
      mx = std::max(mx, s[i]);
    }
    counts->assign(mx + 1, 0);
    if (n == 1) {
      (*counts)[s[0]] = range;
    } else {
      if (s[0] == s[1]) return false;
      (*counts)[s[0]] = (int32_t)br->Read(kAnsLogTab);
      (*counts)[s[1]] = range - (*counts)[s[0]];
    }
    return true;
  }
  if (br->Read(1)) {  // flat
    const int n = (int)ReadVarLenU8(br) + 1;
    if (n > range) return false;
    counts->assign(n, range / n);
    for (int i = 0; i < range % n; i++) (*counts)[i]++;
    return true;
  }
  uint32_t shift;
  {
    const int upper = (int)FloorLog2(kAnsLogTab + 1);
    int log = 0;
    for (; log < upper; log++)
      if (br->Read(1) == 0) break;
    shift = (br->Read(log) | (1u << log)) - 1;
    if (shift > kAnsLogTab + 1) return false;
  }
  const size_t length = ReadVarLenU8(br) + 3;
  counts->assign(length, 0);
  std::vector<int> logcounts(length, 0), same(length, 0);
  int omit_log = -1, omit_pos = -1;
  for (size_t i = 0; i < length; i++) {
    br->Refill();
    const uint32_t idx = (uint32_t)br->Peek(7);
    int sym = 0;
    for (const FixedCode& c : kLogCountCode)
      if ((idx & ((1u << c.len) - 1)) == c.pattern) {
        br->Consume(c.len);
        sym = c.symbol;
        break;
      }
    logcounts[i] = sym - 1;
    if (logcounts[i] == (int)kAnsLogTab) {  // run-length symbol
      const int rle = (int)ReadVarLenU8(br);
      same[i] = rle + 5;
      i += rle + 3;
      continue;
    }
    if (logcounts[i] > omit_log) {
      omit_log = logcounts[i];
      omit_pos = (int)i;
    }
  }
  if (omit_pos < 0) return false;
  if ((size_t)omit_pos + 1 < length && logcounts[omit_pos + 1] == (int)kAnsLogTab) return false;
  int prev = 0, numsame = 0, total = 0;
  for (size_t i = 0; i < length; i++) {
    if (same[i]) {
      numsame = same[i] - 1;
      prev = i > 0 ? (*counts)[i - 1] : 0;
    }
    if (numsame > 0) {
      (*counts)[i] = prev;
      numsame--;
    } else {
      const int code = logcounts[i];
      if ((int)i == omit_pos || code < 0) continue;
      if (shift == 0 || code == 0) {
        (*counts)[i] = 1 << code;
      } else {
        // GetPopulationCountPrecision (ans_common.h:27-34)
        int bitcount = std::min<int>(code, (int)shift - (int)((kAnsLogTab - code) >> 1));
        if (bitcount < 0) bitcount = 0;
        (*counts)[i] = (1 << code) + (int)(br->Read(bitcount) << (code - bitcount));
      }
    }
    total += (*counts)[i];
  }
  (*counts)[omit_pos] = range - total;
  return (*counts)[omit_pos] > 0;
}

// ---------------------------------------------------------- hybrid integers
struct HybridUint {  // dec_ans.h:66-112
  uint32_t split_exponent = 4, split_token = 16, msb_in_token = 2, lsb_in_token = 0;
};

bool ReadHybridUintConfig(BitReader* br, uint32_t log_alpha, HybridUint* c) {  // dec_ans.cc:291-315
  const uint32_t split_exponent = br->Read(CeilLog2(log_alpha + 1));
  uint32_t msb = 0, lsb = 0;
  if (split_exponent != log_alpha) {
    msb = br->Read(CeilLog2(split_exponent + 1));
    if (msb > split_exponent) return false;
    lsb = br->Read(CeilLog2(split_exponent - msb + 1));
  }
  if (lsb + msb > split_exponent) return false;
  c->split_exponent = split_exponent;
  c->split_token = 1u << split_exponent;
  c->msb_in_token = msb;
  c->lsb_in_token = lsb;
  return true;
}

inline uint32_t FinishHybridUint(const HybridUint& c, uint32_t token, BitReader* br) {  // dec_ans.h:221-252
  if (token < c.split_token) return token;
  uint32_t nbits = c.split_exponent - (c.msb_in_token + c.lsb_in_token) +
                   ((token - c.split_token) >> (c.msb_in_token + c.lsb_in_token));
  nbits &= 31u;
  const uint32_t low = token & ((1u << c.lsb_in_token) - 1);
  token >>= c.lsb_in_token;
  const uint32_t bits = (uint32_t)br->Peek(nbits);
  br->Consume(nbits);
  return (((((1u << c.msb_in_token) | (token & ((1u << c.msb_in_token) - 1))) << nbits) | bits)
          << c.lsb_in_token) |
         low;
}

// -------------------------------------------------------------- entropy code
struct Lz77Params {
  bool enabled = false;
  uint32_t min_symbol = 224, min_length = 3;
  HybridUint length_config;
  uint32_t distance_context = 0;
};

struct EntropyCode {  // ANSCode
  bool use_prefix = false;
  uint32_t log_alpha = 0;
  std::vector<uint8_t> context_map;
  std::vector<HybridUint> configs;
  std::vector<AliasEntry> alias;  // num_histograms << log_alpha
  std::vector<PrefixCode> prefix;
  Lz77Params lz77;
  uint32_t max_num_bits = 0;

  void UpdateMaxNumBits(size_t ctx, uint32_t symbol) {  // dec_ans.cc:339-362
    const HybridUint* c = &configs[ctx];
    if (lz77.enabled && lz77.distance_context != ctx && symbol >= lz77.min_symbol) {
      symbol -= lz77.min_symbol;
      c = &lz77.length_config;
    }
    if (symbol < c->split_token) {
      max_num_bits = std::max(max_num_bits, c->split_exponent);
      return;
    }
    const uint32_t extra = c->split_exponent - (c->msb_in_token + c->lsb_in_token) +
                           ((symbol - c->split_token) >> (c->msb_in_token + c->lsb_in_token));
    max_num_bits = std::max(max_num_bits, c->msb_in_token + c->lsb_in_token + extra + 1);
  }
};

constexpr uint32_t kLz77Window = 1u << 20;

class SymbolReader {  // ANSSymbolReader
 public:
  SymbolReader(const EntropyCode* code, BitReader* br) : code_(code) {
    if (!code->use_prefix) {
      state_ = br->Read(32);
      log_entry_ = kAnsLogTab - code->log_alpha;
      entry_mask_ = (1u << log_entry_) - 1;
    }
    if (code->lz77.enabled) window_.reset(new (std::nothrow) uint32_t[kLz77Window]);
  }
  bool Ok() const { return !code_->lz77.enabled || window_; }
  bool FinalStateOk() const { return state_ == (kAnsSignature << 16); }
  // The same with the histogram's alias row and hybrid-uint config handed in (plain ANS): a caller that knows the TWO
  // contexts the next symbol can have before this symbol's value is in looks both rows up ahead and only selects
  // between them afterwards -- the context map and the row address leave the chain from one symbol to the next.
  const AliasEntry* AliasRow(uint32_t ctx) const { return &code_->alias[(size_t)ctx << code_->log_alpha]; }
  const HybridUint* Config(uint32_t ctx) const { return &code_->configs[ctx]; }
  __attribute__((always_inline)) inline uint32_t ReadHybridUintAnsRow(const AliasEntry* t, const HybridUint& cfg, BitReader* br) {
    br->Refill();
    const uint32_t res = state_ & (kAnsTab - 1);
    const uint32_t i = res >> log_entry_, pos = res & entry_mask_;
    const AliasEntry& e = t[i];
    const bool right = pos >= e.cutoff;
    const uint32_t token = right ? e.right_value : i;
    const uint32_t offset = (right ? e.offsets1 : 0u) + pos;
    const uint32_t freq = right ? e.freq1 : e.freq0;
    state_ = freq * (state_ >> kAnsLogTab) + offset;
    if (state_ < (1u << 16)) {
      state_ = (state_ << 16) | (uint32_t)br->Peek(16);
      br->Consume(16);
    }
    return FinishHybridUint(cfg, token, br);
  }
  // (loops that keep the ANS state in a register of their own: modular.inc's self-correcting-predictor track)
  uint32_t State() const { return state_; }
  void SetState(uint32_t s) { state_ = s; }

  inline uint32_t ReadToken(uint32_t histo, BitReader* br) {
    if (code_->use_prefix) return code_->prefix[histo].ReadSymbol(br);
    const uint32_t res = state_ & (kAnsTab - 1);
    const AliasEntry* t = &code_->alias[(size_t)histo << code_->log_alpha];
    const uint32_t i = res >> log_entry_, pos = res & entry_mask_;
    const AliasEntry& e = t[i];
    const bool right = pos >= e.cutoff;
    const uint32_t value = right ? e.right_value : i;
    const uint32_t offset = (right ? e.offsets1 : 0u) + pos;
    const uint32_t freq = right ? e.freq1 : e.freq0;
    state_ = freq * (state_ >> kAnsLogTab) + offset;
    if (state_ < (1u << 16)) {
      state_ = (state_ << 16) | (uint32_t)br->Peek(16);
      br->Consume(16);
    }
    return value;
  }

  // ReadHybridUintClusteredInlined (dec_ans.h:289-356); ctx is a CLUSTERED context
  inline uint32_t ReadHybridUint(uint32_t ctx, BitReader* br) {
    if (window_) {
      if (num_to_copy_ > 0) return CopyOne();
      br->Refill();
      const uint32_t token = ReadToken(ctx, br);
      const Lz77Params& z = code_->lz77;
      if (token >= z.min_symbol) {
        num_to_copy_ = FinishHybridUint(z.length_config, token - z.min_symbol, br) + z.min_length;
        br->Refill();
        const uint32_t d_token = ReadToken(z.distance_context, br);
        uint32_t distance = FinishHybridUint(code_->configs[z.distance_context], d_token, br);
        distance += 1;  // no special distances for 1-D streams (distance_multiplier == 0)
        if (distance > num_decoded_) distance = num_decoded_;
        if (distance > kLz77Window) distance = kLz77Window;
        copy_pos_ = num_decoded_ - distance;
        if (distance == 0) memset(window_.get(), 0, std::min<size_t>(num_to_copy_, kLz77Window) * 4);
        if (num_to_copy_ < z.min_length) {  // length wrapped around
          num_to_copy_ = 0;
          corrupt_ = true;
          return 0;
        }
        return CopyOne();
      }
      const uint32_t v = FinishHybridUint(code_->configs[ctx], token, br);
      window_[(num_decoded_++) & (kLz77Window - 1)] = v;
      return v;
    }
    br->Refill();
    const uint32_t token = ReadToken(ctx, br);
    return FinishHybridUint(code_->configs[ctx], token, br);
  }
  bool Corrupt() const { return corrupt_; }

  // The same for codes known to be plain ANS (no prefix codes, no LZ77) -- what libjxl's encoder
  // writes for AC coefficients: no per-symbol mode tests, always inlined into the coefficient loop.
  bool PlainAns() const { return !code_->use_prefix && !window_; }
  __attribute__((always_inline)) inline uint32_t ReadHybridUintAns(uint32_t ctx, BitReader* br) {
    br->Refill();
    const uint32_t res = state_ & (kAnsTab - 1);
    const AliasEntry* t = &code_->alias[(size_t)ctx << code_->log_alpha];
    const uint32_t i = res >> log_entry_, pos = res & entry_mask_;
    const AliasEntry& e = t[i];
    const bool right = pos >= e.cutoff;
    const uint32_t token = right ? e.right_value : i;
    const uint32_t offset = (right ? e.offsets1 : 0u) + pos;
    const uint32_t freq = right ? e.freq1 : e.freq0;
    state_ = freq * (state_ >> kAnsLogTab) + offset;
    if (state_ < (1u << 16)) {
      state_ = (state_ << 16) | (uint32_t)br->Peek(16);
      br->Consume(16);
    }
    return FinishHybridUint(code_->configs[ctx], token, br);
  }

 private:
  inline uint32_t CopyOne() {
    const uint32_t v = window_[(copy_pos_++) & (kLz77Window - 1)];
    num_to_copy_--;
    window_[(num_decoded_++) & (kLz77Window - 1)] = v;
    return v;
  }
  const EntropyCode* code_;
  uint32_t state_ = kAnsSignature << 16;
  uint32_t log_entry_ = 0, entry_mask_ = 0;
  std::unique_ptr<uint32_t[]> window_;
  uint32_t num_decoded_ = 0, num_to_copy_ = 0, copy_pos_ = 0;
  bool corrupt_ = false;
};

int DecodeEntropyCode(BitReader* br, size_t num_contexts, EntropyCode* code, bool disallow_lz77, int depth);

// DecodeContextMap (dec_context_map.cc:47-95)
int DecodeContextMap(BitReader* br, std::vector<uint8_t>* map, size_t* num_histograms, int depth) {
  if (br->Read(1)) {  // simple: fixed width entries
    const uint32_t bits = br->Read(2);
    for (uint8_t& e : *map) e = bits ? (uint8_t)br->Read(bits) : 0;
  } else {
    const bool use_mtf = br->Read(1) != 0;
    if (depth > 1) return kBad;  // a context map's code has exactly one context: no further nesting
    EntropyCode code;
    int rc = DecodeEntropyCode(br, 1, &code, /*disallow_lz77=*/map->size() <= 2, depth + 1);
    if (rc) return rc;
    SymbolReader reader(&code, br);
    if (!reader.Ok()) return JXLHIP_ERR_OUT_OF_MEMORY;
    uint32_t maxsym = 0;
    for (uint8_t& e : *map) {
      const uint32_t sym = reader.ReadHybridUint(code.context_map[0], br);
      maxsym = std::max(maxsym, sym);
      e = (uint8_t)sym;
    }
    if (maxsym >= 256 || reader.Corrupt() || !reader.FinalStateOk()) return kBad;
    if (use_mtf) {  // inverse move-to-front (inverse_mtf-inl.h)
      uint8_t mtf[256];
      for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
      for (uint8_t& e : *map) {
        const uint8_t idx = e;
        const uint8_t v = mtf[idx];
        e = v;
        if (idx) {
          memmove(mtf + 1, mtf, idx);
          mtf[0] = v;
        }
      }
    }
  }
  *num_histograms = (size_t)*std::max_element(map->begin(), map->end()) + 1;
  std::vector<bool> seen(*num_histograms, false);  // every histogram must be referenced
  size_t found = 0;
  for (uint8_t e : *map)
    if (!seen[e]) {
      seen[e] = true;
      found++;
    }
  return found == *num_histograms ? kOk : kBad;
}

// DecodeHistograms (dec_ans.cc:364-395) + DecodeANSCodes (dec_ans.cc:205-290)
int DecodeEntropyCode(BitReader* br, size_t num_contexts, EntropyCode* code, bool disallow_lz77, int depth) {
  Lz77Params& z = code->lz77;
  z.enabled = br->Read(1) != 0;  // LZ77Params::VisitFields (dec_ans.cc:328-337)
  if (z.enabled) {
    static const U32Dist kMinSymbol = {{0, 0, 0, 15}, {224, 512, 4096, 8}};
    static const U32Dist kMinLength = {{0, 0, 2, 8}, {3, 4, 5, 9}};
    z.min_symbol = ReadU32(br, kMinSymbol);
    z.min_length = ReadU32(br, kMinLength);
    num_contexts++;
    if (!ReadHybridUintConfig(br, 8, &z.length_config)) return kBad;
    if (disallow_lz77) return kBad;
  }
  size_t num_histograms = 1;
  code->context_map.assign(num_contexts, 0);
  if (num_contexts > 1) {
    int rc = DecodeContextMap(br, &code->context_map, &num_histograms, depth);
    if (rc) return rc;
  }
  z.distance_context = code->context_map.back();
  code->use_prefix = br->Read(1) != 0;
  code->log_alpha = code->use_prefix ? (uint32_t)kMaxCodeLength : br->Read(2) + 5;
  code->configs.resize(num_histograms);
  for (HybridUint& c : code->configs)
    if (!ReadHybridUintConfig(br, code->log_alpha, &c)) return kBad;
  const size_t max_alphabet = (size_t)1 << code->log_alpha;
  if (code->use_prefix) {
    code->prefix.resize(num_histograms);
    std::vector<uint32_t> sizes(num_histograms);
    for (uint32_t& s : sizes) {
      s = ReadVarLenU16(br) + 1;
      if (s > max_alphabet) return kBad;
    }
    for (size_t h = 0; h < num_histograms; h++) {
      if (sizes[h] > 1) {
        if (!ReadPrefixCode(br, sizes[h], &code->prefix[h])) return kBad;
      } else {
        code->prefix[h].SetSingle(0);
      }
      // every entry that names a symbol, second-level ones included (dec_ans.cc:226-230); root
      // slots that point to a sub-table carry bits > kRootBits
      for (const PrefixEntry& e : code->prefix[h].table)
        if (e.bits <= kRootBits) code->UpdateMaxNumBits(h, e.value);
      if (!br->Healthy()) return kBad;
    }
  } else {
    code->alias.assign(num_histograms << code->log_alpha, AliasEntry{});
    for (size_t h = 0; h < num_histograms; h++) {
      std::vector<int32_t> counts;
      if (!ReadDistribution(br, &counts)) return kBad;
      if (counts.size() > max_alphabet) return kBad;
      while (!counts.empty() && counts.back() == 0) counts.pop_back();
      for (size_t s = 0; s < counts.size(); s++)
        if (counts[s]) code->UpdateMaxNumBits(h, (uint32_t)s);
          if (!BuildAliasTable(counts, code->log_alpha, &code->alias[h << code->log_alpha])) return kBad;
      if (!br->Healthy()) return kBad;
    }
  }
  return br->Healthy() ? kOk : kBad;
}

// ------------------------------------------------------- strategy geometry
constexpr uint8_t kCovX[27] = {1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32};
constexpr uint8_t kCovY[27] = {1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16};
// kStrategyOrder (coeff_order.h:44-47): strategies that share a natural order share a bucket
constexpr uint8_t kStrategyOrder[27] = {0, 1, 1, 1, 2, 3, 4, 4, 5,  5,  6,  6,  1,  1,
                                        1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
constexpr int kNumOrders = 13;
constexpr uint32_t kPermutationContexts = 8;

struct OrderLayout {
  uint32_t blocks[kNumOrders];       // covered blocks of the bucket's transforms
  size_t offset[kNumOrders][3 + 1];  // in coefficients (kCoeffOrderOffset * 64)
  size_t total;
  OrderLayout() {
    for (int o = 0; o < kNumOrders; o++) blocks[o] = 0;
    for (int s = 0; s < 27; s++) blocks[kStrategyOrder[s]] = (uint32_t)kCovX[s] * kCovY[s];
    size_t pos = 0;
    for (int o = 0; o < kNumOrders; o++)
      for (int c = 0; c < 3; c++) {
        offset[o][c] = pos;
        pos += (size_t)blocks[o] * 64;
        offset[o][c + 1] = pos;
      }
    total = pos;
  }
};
const OrderLayout& Layout() {
  static const OrderLayout l;
  return l;
}

// AcStrategy::ComputeNaturalCoeffOrder (ac_strategy.cc:28-79): zig-zag over the
// cx x cx square, keeping the lines of the cy x cx coefficient rectangle, with
// the cx*cy LLF coefficients first
void NaturalOrder(int strategy, uint32_t* out) {
  size_t cx = kCovX[strategy], cy = kCovY[strategy];
  if (cy > cx) std::swap(cx, cy);  // CoefficientLayout: rows = the smaller side
  const size_t ratio = cx / cy, mask = ratio - 1, shift = CeilLog2((uint32_t)ratio);
  const size_t side = cx * 8;
  size_t cur = cx * cy;
  auto emit = [&](size_t x, size_t y) {
    if (y & mask) return;
    y >>= shift;
    const size_t val = (x < cx && y < cy) ? y * cx + x : cur++;
    out[val] = (uint32_t)(y * side + x);
  };
  for (size_t i = 0; i < side; i++)
    for (size_t j = 0; j <= i; j++) {
      size_t x = j, y = i - j;
      if (i & 1) std::swap(x, y);
      emit(x, y);
    }
  for (size_t ip = side - 1; ip > 0; ip--) {
    const size_t i = ip - 1;
    for (size_t j = 0; j <= i; j++) {
      size_t x = side - 1 - (i - j), y = side - 1 - j;
      if (i & 1) std::swap(x, y);
      emit(x, y);
    }
  }
}

// DecodeLehmerCode (lehmer_code.h:60-100): i-th unused element via a Fenwick tree
bool LehmerToPermutation(const uint32_t* code, size_t n, uint32_t* perm) {
  const uint32_t log2n = CeilLog2((uint32_t)n);
  const size_t padded = (size_t)1 << log2n;
  std::vector<uint32_t> tree(padded);
  for (size_t i = 0; i < padded; i++) tree[i] = (uint32_t)((i + 1) & (~(i + 1) + 1));
  for (size_t i = 0; i < n; i++) {
    if (code[i] + i >= n) return false;
    uint32_t rank = code[i] + 1;
    size_t bit = padded, next = 0;
    for (uint32_t b = 0; b <= log2n; b++) {
      const size_t cand = next + bit;
      bit >>= 1;
      if (tree[cand - 1] < rank) {
        next = cand;
        rank -= tree[cand - 1];
      }
    }
    perm[i] = (uint32_t)next;
    next += 1;
    while (next <= padded) {
      tree[next - 1] -= 1;
      next += next & (~next + 1);
    }
  }
  return true;
}

uint32_t CoeffOrderContext(uint32_t v) {  // coeff_order.cc:30-34: token of HybridUint(0,0,0), capped
  if (v == 0) return 0;
  return std::min(FloorLog2(v) + 1, kPermutationContexts - 1);
}

// ReadPermutation (coeff_order.cc:37-64)
int ReadPermutation(size_t skip, size_t size, uint32_t* order, BitReader* br, SymbolReader* reader,
                    const EntropyCode& code) {
  std::vector<uint32_t> lehmer(size, 0);
  const uint32_t end = reader->ReadHybridUint(code.context_map[CoeffOrderContext((uint32_t)size)], br) +
                       (uint32_t)skip;
  if (end > size) return kBad;
  uint32_t last = 0;
  for (size_t i = skip; i < end; i++) {
    lehmer[i] = reader->ReadHybridUint(code.context_map[CoeffOrderContext(last)], br);
    last = lehmer[i];
    if (lehmer[i] >= size - i) return kBad;
  }
  if (!order) return kOk;
  return LehmerToPermutation(lehmer.data(), size, order) ? kOk : kBad;
}

}  // namespace

// ------------------------------------------------------------------ the pass
struct jxlhip_ac_pass {
  uint32_t used_orders = 0;
  uint32_t num_histograms = 1;
  std::vector<uint32_t> orders;  // Layout().total
  EntropyCode code;
  // BlockCtxMap (ac_context.h:85-150)
  uint32_t num_dc_ctxs = 1;
  std::vector<uint32_t> qf_thresholds;
  std::vector<uint8_t> block_ctx;
  uint32_t num_block_ctxs = 0;
  uint32_t NumAcContexts() const { return num_block_ctxs * (37 + 458); }
};

namespace {

// kDefaultCtxMap (ac_context.h:91-97)
constexpr uint8_t kDefaultBlockCtx[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12,
                                          13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};

// DecodeCoeffOrders (coeff_order.cc:97-156)
int DecodeOrders(BitReader* br, uint32_t used_orders, uint32_t used_acs, uint32_t* order) {
  EntropyCode code;
  std::unique_ptr<SymbolReader> reader;
  if (used_orders) {
    int rc = DecodeEntropyCode(br, kPermutationContexts, &code, false, 0);
    if (rc) return rc;
    reader.reset(new (std::nothrow) SymbolReader(&code, br));
    if (!reader || !reader->Ok()) return JXLHIP_ERR_OUT_OF_MEMORY;
  }
  uint32_t acs_mask = 0;
  for (int s = 0; s < 27; s++)
    if (used_acs & (1u << s)) acs_mask |= 1u << kStrategyOrder[s];
  const OrderLayout& L = Layout();
  std::vector<uint32_t> natural;
  uint32_t computed = 0;
  for (int s = 0; s < 27; s++) {
    const uint32_t ord = kStrategyOrder[s];
    if (computed & (1u << ord)) continue;
    computed |= 1u << ord;
    const bool used = (acs_mask >> ord) & 1;
    const size_t llf = L.blocks[ord], size = llf * 64;
    const bool transmitted = (used_orders >> ord) & 1;
    if (used || transmitted) {
      natural.resize(size);
      NaturalOrder(s, natural.data());
    }
    if (!transmitted) {
      if (used)
        for (int c = 0; c < 3; c++) memcpy(order + L.offset[ord][c], natural.data(), size * sizeof(uint32_t));
      continue;
    }
    for (int c = 0; c < 3; c++) {
      uint32_t* dst = used ? order + L.offset[ord][c] : nullptr;
      int rc = ReadPermutation(llf, size, dst, br, reader.get(), code);
      if (rc) return rc;
      if (dst)
        for (size_t k = 0; k < size; k++) dst[k] = natural[dst[k]];
    }
  }
  if (used_orders && (reader->Corrupt() || !reader->FinalStateOk())) return kBad;
  return br->Healthy() ? kOk : kBad;
}

// ZeroDensityContext tables (ac_context.h:32-48), as ranges
inline uint32_t CoeffFreqContext(uint32_t k) {  // k in 1..63
  return k < 16 ? k - 1 : (k < 32 ? 15 + (k - 16) / 2 : 23 + (k - 32) / 4);
}
inline uint32_t CoeffNumNonzeroContext(uint32_t n) {  // n in 1..63
  if (n < 2) return 0;
  if (n < 3) return 31;
  if (n < 5) return 62;
  if (n < 9) return 93;
  if (n < 13) return 123;
  if (n < 21) return 152;
  if (n < 33) return 180;
  return 206;
}

struct ZeroDensityLut {
  uint16_t v[64][64];  // [nonzeros_left][k]
  ZeroDensityLut() {
    for (int n = 0; n < 64; n++)
      for (int k = 0; k < 64; k++)
        v[n][k] = (n && k) ? (uint16_t)((CoeffNumNonzeroContext(n) + CoeffFreqContext(k)) * 2) : 0;
  }
};
const ZeroDensityLut& ZdLut() {
  static const ZeroDensityLut l;
  return l;
}

// Where the decoded coefficients of a group go: the reference's dense block stream (DenseSink: values are ADDED,
// dec_group.cc:527-531 -- progressive passes accumulate), or the non-zero ones as (position << 16 | value) words
// per channel (SparseSink, single-pass frames: nine out of ten coefficients of a d1.0 frame are zero, and the
// dense stream is what crosses PCIe otherwise).
template <typename T>
struct DenseSink {
  static constexpr bool kSparse = false;
  T* const* coeffs;
  bool out_of_range = false;  // 16-bit buffers only
  bool overflow = false;
  inline void Put(int c, size_t pos, int32_t coeff) {
    if constexpr (sizeof(T) == 2) {
      const int32_t sum = (int32_t)coeffs[c][pos] + coeff;
      out_of_range |= sum != (int32_t)(int16_t)sum;
      coeffs[c][pos] = (T)sum;
    } else {
      coeffs[c][pos] = (T)(coeffs[c][pos] + (T)coeff);
    }
  }
};
struct SparseSink {
  static constexpr bool kSparse = true;
  uint32_t* ent[3];
  uint32_t cnt[3] = {0, 0, 0};
  uint32_t cap[3];
  bool out_of_range = false;
  bool overflow = false;
  inline void Put(int c, size_t pos, int32_t coeff) {
    if (coeff == 0) return;
    out_of_range |= coeff != (int32_t)(int16_t)coeff;
    if (cnt[c] >= cap[c]) {
      overflow = true;
      return;
    }
    ent[c][cnt[c]++] = ((uint32_t)pos << 16) | (uint32_t)(uint16_t)(int16_t)coeff;
  }
};

template <typename Sink, bool kPlainAns>
int DecodeGroupImpl(const jxlhip_ac_pass* pass, uint32_t xsb, uint32_t ysb, uint32_t gx, uint32_t gy,
                    const uint8_t* acs_map, const int32_t* raw_quant, const uint8_t* quant_dc, BitReader* br_io,
                    uint32_t shift, Sink& sink, size_t* ncoeffs) {
  // the bit reader in a local for the length of the group (through the caller's pointer its bit count is reloaded after
  // every coefficient store -- same type -- which puts a store-to-load round trip into the chain between two symbols)
  BitReader local_br = *br_io;
  BitReader* const br = &local_br;
  struct WriteBack {
    BitReader* to;
    const BitReader* from;
    ~WriteBack() { *to = *from; }
  } write_back{br_io, &local_br};
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  if (bx0 >= xsb || by0 >= ysb) return JXLHIP_ERR_INVALID_ARGUMENT;
  const uint32_t gw = std::min(32u, xsb - bx0), gh = std::min(32u, ysb - by0);
  // histogram set of this group (dec_group.cc:606-613)
  uint32_t selector = 0;
  if (pass->num_histograms > 1) selector = br->Read(CeilLog2(pass->num_histograms));
  if (selector >= pass->num_histograms) return kBad;
  const uint32_t ctx_offset = selector * pass->NumAcContexts();
  SymbolReader reader(&pass->code, br);
  if (!reader.Ok()) return JXLHIP_ERR_OUT_OF_MEMORY;
  const uint8_t* cmap = pass->code.context_map.data();
  const OrderLayout& L = Layout();
  const ZeroDensityLut& zd = ZdLut();
  const uint32_t nb = pass->num_block_ctxs;
  const uint32_t nqf = (uint32_t)pass->qf_thresholds.size();
  // number of non-zeros per block, for the context of the next ones (GroupDecCache::num_nzeroes).
  // Zeroed: a damaged strategy map can leave cells that no varblock covers, and their
  // (never written) counts feed the neighbours' contexts.  The map itself is validated on the
  // device by k_prepare; here it only has to be harmless.
  int32_t nz[3][32][32];
  memset(nz, 0, sizeof(nz));
  size_t offset = 0;
  for (uint32_t by = 0; by < gh; by++) {
    for (uint32_t bx = 0; bx < gw; bx++) {
      const size_t cell = (size_t)(by0 + by) * xsb + bx0 + bx;
      const uint32_t raw = acs_map[cell];
      if (!(raw & 1)) continue;  // not the first block of its varblock
      const uint32_t s = raw >> 1;
      if (s >= 27) return kBad;
      const uint32_t cx = kCovX[s], cy = kCovY[s];
      if (bx + cx > gw || by + cy > gh) return kBad;
      const uint32_t covered = cx * cy, log2c = FloorLog2(covered), size = covered * 64;
      if (offset + size > 65536) return kBad;
      const uint32_t ord = kStrategyOrder[s];
      const uint32_t qf = (uint32_t)raw_quant[cell];
      uint32_t qf_idx = 0;
      for (uint32_t t = 0; t < nqf; t++) qf_idx += qf > pass->qf_thresholds[t];
      const uint32_t dc_idx = quant_dc ? quant_dc[cell] : 0;
      if (dc_idx >= pass->num_dc_ctxs) return kBad;
      for (int c : {1, 0, 2}) {
        // BlockCtxMap::Context (ac_context.h:102-112)
        uint32_t idx = c < 2 ? (uint32_t)(c ^ 1) : 2u;
        idx = idx * kNumOrders + ord;
        idx = idx * (nqf + 1) + qf_idx;
        idx = idx * pass->num_dc_ctxs + dc_idx;
        const uint32_t block_ctx = pass->block_ctx[idx];
        // PredictFromTopAndLeft (entropy_coder.h:25-35)
        int32_t predicted;
        if (bx == 0) predicted = by == 0 ? 32 : nz[c][by - 1][bx];
        else if (by == 0) predicted = nz[c][by][bx - 1];
        else predicted = (nz[c][by - 1][bx] + nz[c][by][bx - 1] + 1) / 2;
        // BlockCtxMap::NonZeroContext (ac_context.h:133-143)
        uint32_t nzp = predicted >= 64 ? 64u : (uint32_t)predicted;
        const uint32_t nzc = nzp < 8 ? nzp : 4 + nzp / 2;
        uint32_t nzeros = kPlainAns ? reader.ReadHybridUintAns(cmap[ctx_offset + nzc * nb + block_ctx], br)
                                    : reader.ReadHybridUint(cmap[ctx_offset + nzc * nb + block_ctx], br);
        if (nzeros > size - covered) return kBad;
        const int32_t per_block = (int32_t)((nzeros + covered - 1) >> log2c);
        for (uint32_t y = 0; y < cy; y++)
          for (uint32_t x = 0; x < cx; x++) nz[c][by + y][bx + x] = per_block;
        // DecodeACVarBlock's coefficient loop (dec_group.cc:510-538)
        const uint8_t* hmap = cmap + ctx_offset + nb * 37 + 458 * block_ctx;
        const uint32_t* order = pass->orders.data() + L.offset[ord][c];
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        // sparse sink: the write cursor lives in locals for the block (through the sink it would be reloaded after
        // every store: the entries and the counts are both uint32_t)
        uint32_t* w = nullptr;
        uint32_t* wend = nullptr;
        bool oor = false;
        if constexpr (Sink::kSparse) {
          w = sink.ent[c] + sink.cnt[c];
          wend = sink.ent[c] + sink.cap[c];
        }
        // one coefficient's value, placed (shared by the two forms of the loop below)
        auto place = [&](uint32_t k, uint32_t u) __attribute__((always_inline)) {
          const uint32_t magnitude = u >> 1, neg = (~u) & 1;  // UnpackSigned
          const int32_t coeff = (int32_t)((magnitude ^ (neg - 1)) << shift);
          if constexpr (Sink::kSparse) {
            if (u != 0) {
              oor |= coeff != (int32_t)(int16_t)coeff;
              if (w < wend) *w++ = ((uint32_t)(offset + order[k]) << 16) | (uint32_t)(uint16_t)(int16_t)coeff;
              else sink.overflow = true;
            }
          } else {
            sink.Put(c, offset + order[k], coeff);
          }
        };
        if constexpr (kPlainAns) {
          // The context of a coefficient depends on the one before it twice over: on whether that one was zero (prev)
          // and on the number of non-zeros still to come (entropy_coder.h:203-230, dec_group.cc:510-538) -- context map,
          // histogram row and hybrid-uint config of the next symbol wait for this symbol's value.  But there are only two
          // possibilities, known beforehand: both rows are looked up while this symbol is being decoded, and its value
          // only selects one of them.  (nzeros <= size - covered was checked: left < 64 throughout.)
          if (nzeros != 0 && covered < size) {
            uint32_t k = covered;
            const uint32_t first = hmap[zd.v[(nzeros + covered - 1) >> log2c][k >> log2c] + prev];
            const AliasEntry* row = reader.AliasRow(first);
            const HybridUint* cfg = reader.Config(first);
            for (;;) {
              const uint32_t kn = std::min((k + 1) >> log2c, 63u);  // (k + 1 == size: nothing is read with it)
              const uint32_t c0 = hmap[zd.v[(nzeros + covered - 1) >> log2c][kn]];          // this one zero
              const uint32_t c1 = hmap[zd.v[(nzeros + covered - 2) >> log2c][kn] + 1u];     // this one not
              const AliasEntry* const row0 = reader.AliasRow(c0);
              const AliasEntry* const row1 = reader.AliasRow(c1);
              const HybridUint* const cfg0 = reader.Config(c0);
              const HybridUint* const cfg1 = reader.Config(c1);
              const uint32_t u = reader.ReadHybridUintAnsRow(row, *cfg, br);
              place(k, u);
              const bool nonzero = u != 0;
              nzeros -= nonzero;
              k++;
              if (k >= size || nzeros == 0) break;
              row = nonzero ? row1 : row0;
              cfg = nonzero ? cfg1 : cfg0;
            }
          }
        } else {
          for (uint32_t k = covered; k < size && nzeros != 0; k++) {
            const uint32_t left = (nzeros + covered - 1) >> log2c;
            if (left >= 64) return kBad;  // more non-zeros than positions: invalid stream
            const uint32_t ctx = zd.v[left][k >> log2c] + prev;
            const uint32_t u = reader.ReadHybridUint(hmap[ctx], br);
            place(k, u);
            prev = u != 0;
            nzeros -= prev;
          }
        }
        if constexpr (Sink::kSparse) {
          sink.cnt[c] = (uint32_t)(w - sink.ent[c]);
          sink.out_of_range |= oor;
        }
        if (nzeros != 0) return kBad;
      }
      offset += size;
    }
  }
  if (reader.Corrupt() || !reader.FinalStateOk()) return kBad;
  if (!br->Healthy()) return kBad;
  if (sink.out_of_range || sink.overflow) return JXLHIP_ERR_RANGE;
  if (ncoeffs) *ncoeffs = offset;
  return kOk;
}

template <typename Sink>
int DecodeGroupS(const jxlhip_ac_pass* pass, uint32_t xsb, uint32_t ysb, uint32_t gx, uint32_t gy,
                 const uint8_t* acs_map, const int32_t* raw_quant, const uint8_t* quant_dc, BitReader* br,
                 uint32_t shift, Sink& sink, size_t* ncoeffs) {
  if (!pass->code.use_prefix && !pass->code.lz77.enabled)
    return DecodeGroupImpl<Sink, true>(pass, xsb, ysb, gx, gy, acs_map, raw_quant, quant_dc, br, shift, sink, ncoeffs);
  return DecodeGroupImpl<Sink, false>(pass, xsb, ysb, gx, gy, acs_map, raw_quant, quant_dc, br, shift, sink, ncoeffs);
}

template <typename T>
int DecodeGroupT(const jxlhip_ac_pass* pass, uint32_t xsb, uint32_t ysb, uint32_t gx, uint32_t gy,
                 const uint8_t* acs_map, const int32_t* raw_quant, const uint8_t* quant_dc, BitReader* br,
                 uint32_t shift, T* const coeffs[3], size_t* ncoeffs) {
  DenseSink<T> sink;
  sink.coeffs = coeffs;
  return DecodeGroupS(pass, xsb, ysb, gx, gy, acs_map, raw_quant, quant_dc, br, shift, sink, ncoeffs);
}

}  // namespace

extern "C" {

int jxlhip_ac_pass_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint32_t used_acs,
                          uint32_t num_histograms, const jxlhip_block_ctx_map* bcm, jxlhip_ac_pass** out) {
  if (!data || !bit_pos || !out || num_histograms == 0 || (used_acs >> 27)) return JXLHIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::unique_ptr<jxlhip_ac_pass> p(new (std::nothrow) jxlhip_ac_pass());
  if (!p) return JXLHIP_ERR_OUT_OF_MEMORY;
  if (bcm) {
    if (bcm->num_dc_ctxs == 0 || bcm->num_qf_thresholds > 15 || bcm->ctx_map_size > JXLHIP_BLOCK_CTX_MAP_MAX ||
        bcm->ctx_map_size != 3u * kNumOrders * (bcm->num_qf_thresholds + 1) * bcm->num_dc_ctxs)
      return JXLHIP_ERR_INVALID_ARGUMENT;
    p->num_dc_ctxs = bcm->num_dc_ctxs;
    p->qf_thresholds.assign(bcm->qf_thresholds, bcm->qf_thresholds + bcm->num_qf_thresholds);
    p->block_ctx.assign(bcm->ctx_map, bcm->ctx_map + bcm->ctx_map_size);
  } else {
    p->block_ctx.assign(kDefaultBlockCtx, kDefaultBlockCtx + 39);
  }
  p->num_block_ctxs = (uint32_t)*std::max_element(p->block_ctx.begin(), p->block_ctx.end()) + 1;
  p->num_histograms = num_histograms;
  BitReader br(data, size, *bit_pos);
  static const U32Dist kOrderEnc = {{0, 0, 0, 13}, {0x5F, 0x13, 0, 0}};  // frame_header.h:503-504
  p->used_orders = ReadU32(&br, kOrderEnc);
  p->orders.assign(Layout().total, 0);
  int rc = DecodeOrders(&br, p->used_orders, used_acs, p->orders.data());
  if (rc) return rc;
  const size_t num_contexts = (size_t)num_histograms * p->NumAcContexts();
  rc = DecodeEntropyCode(&br, num_contexts, &p->code, false, 0);
  if (rc) return rc;
  // the coefficient loop may look 16 contexts past the last one (dec_frame.cc:411-413)
  p->code.context_map.resize(num_contexts + (474 - 458), 0);
  *bit_pos = br.BitsConsumed();
  *out = p.release();
  return kOk;
}

void jxlhip_ac_pass_destroy(jxlhip_ac_pass* pass) { delete pass; }

// ---- dequant-matrix encodings (quant_weights.cc:373-511) -----------------------
namespace {
// F16Coder::Read (fields.cc:550-574)
bool ReadF16(BitReader* br, float* v) {
  const uint32_t bits16 = br->Read(16);
  const uint32_t sign = bits16 >> 15, biased_exp = (bits16 >> 10) & 0x1F, mantissa = bits16 & 0x3FF;
  if (biased_exp == 31) return false;  // infinity / NaN
  if (biased_exp == 0) {
    *v = (1.0f / 16384) * (mantissa * (1.0f / 1024));
    if (sign) *v = -*v;
    return true;
  }
  const uint32_t bits32 = (sign << 31) | ((biased_exp + (127 - 15)) << 23) | (mantissa << 13);
  memcpy(v, &bits32, 4);
  return true;
}

constexpr float kAlmostZero = 1e-8f;

bool ReadDctParams(BitReader* br, uint32_t* nb, float bands[3][JXLHIP_MAX_DISTANCE_BANDS]) {  // :373-386
  *nb = br->Read(4) + 1;
  for (int c = 0; c < 3; c++) {
    for (uint32_t i = 0; i < *nb; i++)
      if (!ReadF16(br, &bands[c][i])) return false;
    if (bands[c][0] < kAlmostZero) return false;
    bands[c][0] *= 64.0f;
  }
  return true;
}

// table kinds whose matrix is a single 8x8 block (required_size_x * required_size_y == 1)
constexpr uint32_t kSingleBlockKinds = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 9) | (1u << 10);

int ReadQuantEncoding(BitReader* br, uint32_t kind, jxlhip_quant_encoding* e) {  // Decode, :388-470
  memset(e, 0, sizeof(*e));
  const uint32_t mode = br->Read(3);
  const bool single = (kSingleBlockKinds >> kind) & 1;
  auto weights = [&](int n, int scaled, bool check) {
    for (int c = 0; c < 3; c++)
      for (int i = 0; i < n; i++) {
        if (!ReadF16(br, &e->weights[c][i])) return false;
        if (check && fabsf(e->weights[c][i]) < kAlmostZero) return false;
        if (i < scaled) e->weights[c][i] *= 64;
      }
    return true;
  };
  switch (mode) {
    case JXLHIP_QUANT_LIBRARY: break;  // kCeilLog2NumPredefinedTables == 0 bits
    case JXLHIP_QUANT_ID:
      if (!single || !weights(3, 3, true)) return kBad;
      break;
    case JXLHIP_QUANT_DCT2:
      if (!single || !weights(6, 6, true)) return kBad;
      break;
    case JXLHIP_QUANT_DCT4X8:
      if (!single || !weights(1, 0, true) || !ReadDctParams(br, &e->num_bands, e->bands)) return kBad;
      break;
    case JXLHIP_QUANT_DCT4:
      if (!single || !weights(2, 0, true) || !ReadDctParams(br, &e->num_bands, e->bands)) return kBad;
      break;
    case JXLHIP_QUANT_AFV:
      if (!single || !weights(9, 6, false) || !ReadDctParams(br, &e->num_bands, e->bands) ||
          !ReadDctParams(br, &e->num_bands_afv_4x4, e->bands_afv_4x4))
        return kBad;
      break;
    case JXLHIP_QUANT_DCT:
      if (!ReadDctParams(br, &e->num_bands, e->bands)) return kBad;
      break;
    default:  // kQuantModeRAW: ModularFrameDecoder::DecodeQuantTable
      e->mode = mode;
      return JXLHIP_ERR_UNSUPPORTED;
  }
  e->mode = mode;
  return br->Healthy() ? kOk : kBad;
}

int ReadQuantEncodings(BitReader* br, jxlhip_quant_encoding* enc) {  // DequantMatrices::Decode, :497-511
  const bool all_default = br->Read(1);
  memset(enc, 0, sizeof(*enc) * JXLHIP_NUM_QUANT_TABLES);
  if (all_default) return br->Healthy() ? kOk : kBad;
  for (uint32_t k = 0; k < JXLHIP_NUM_QUANT_TABLES; k++) {
    const int rc = ReadQuantEncoding(br, k, &enc[k]);
    if (rc) return rc;
  }
  return kOk;
}
}  // namespace

uint32_t jxlhip_num_toc_entries(uint32_t num_groups, uint32_t num_dc_groups, uint32_t num_passes) {
  if (num_groups == 1 && num_passes == 1) return 1;
  return 2 + num_dc_groups + num_groups * num_passes;
}

int jxlhip_toc_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint32_t num_entries, uint64_t* offsets,
                      uint32_t* sizes, uint64_t* total_size) {
  if (!data || !bit_pos || !offsets || !sizes || num_entries == 0) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (num_entries > 65536) return kBad;  // toc.cc:32-37
  if (*bit_pos >= size * 8) return kBad;
  BitReader br(data, size, *bit_pos);
  auto to_byte_boundary = [&]() {  // BitReader::JumpToByteBoundary: the padding must be zero
    const uint32_t rem = (uint32_t)(br.BitsConsumed() % 8);
    return rem == 0 || br.Read(8 - rem) == 0;
  };
  std::vector<uint32_t> perm;
  if (br.Read(1)) {
    // DecodePermutation (coeff_order.cc:66-80)
    if ((size_t)num_entries * 12 > size * 8 - std::min(size * 8, br.BitsConsumed())) return kBad;
    EntropyCode code;
    int rc = DecodeEntropyCode(&br, kPermutationContexts, &code, false, 0);
    if (rc) return rc;
    SymbolReader reader(&code, &br);
    if (!reader.Ok()) return JXLHIP_ERR_OUT_OF_MEMORY;
    perm.resize(num_entries);
    rc = ReadPermutation(0, num_entries, perm.data(), &br, &reader, code);
    if (rc) return rc;
    if (reader.Corrupt() || !reader.FinalStateOk()) return kBad;
  }
  if (!to_byte_boundary()) return kBad;
  static const U32Dist kTocDist = {{10, 14, 22, 30}, {0, 1024, 17408, 4211712}};  // toc.h:25-26
  std::vector<uint32_t> raw(num_entries);
  for (uint32_t i = 0; i < num_entries; i++) raw[i] = ReadU32(&br, kTocDist);
  if (!to_byte_boundary() || !br.Healthy()) return kBad;
  std::vector<uint64_t> off(num_entries);
  uint64_t total = 0;
  for (uint32_t i = 0; i < num_entries; i++) {
    off[i] = total;
    total += raw[i];
  }
  for (uint32_t i = 0; i < num_entries; i++) {
    const uint32_t src = perm.empty() ? i : perm[i];
    offsets[i] = off[src];
    sizes[i] = raw[src];
  }
  if (total_size) *total_size = total;
  *bit_pos = br.BitsConsumed();
  return kOk;
}

// ---- frame header (frame_header.cc, loop_filter.cc, fields.cc) -------------------------------
namespace {
struct FieldReader {
  explicit FieldReader(BitReader* b) : br(b) {}
  BitReader* br;
  bool ok = true;
  bool Bool() { return br->Read(1) != 0; }
  uint32_t Bits(uint32_t n) { return br->Read(n); }
  // U32 with a four-way distribution: sel 0..3 -> offset[sel] + Bits(bits[sel])
  uint32_t U32(const U32Dist& d) { return ReadU32(br, d); }
  uint64_t U64() {  // U64Coder::Read, fields.cc:494-520
    const uint32_t sel = br->Read(2);
    if (sel == 0) return 0;
    if (sel == 1) return 1 + br->Read(4);
    if (sel == 2) return 17 + br->Read(8);
    uint64_t result = br->Read(12);
    uint32_t shift = 12;
    while (br->Read(1)) {
      if (shift == 60) {
        result |= (uint64_t)br->Read(4) << shift;
        break;
      }
      result |= (uint64_t)br->Read(8) << shift;
      shift += 8;
      if (!br->Healthy()) break;
    }
    return result;
  }
  float F16() {
    float v = 0;
    if (!ReadF16(br, &v)) ok = false;
    return v;
  }
  // BeginExtensions + EndExtensions (fields.cc:201-255): the extension sizes, then skip their bits.
  // The reference keeps the position / total of the LAST bundle that had extensions in the visitor
  // and does not clear them for the next bundle, which is observable: a loop filter with extensions
  // inside a frame header without any fails ("Read more extension bits than budgeted"), and with
  // both, the frame header skips the loop filter's bit count again.  Reproduced as is.
  uint64_t pos_after_ext_size = 0, total_extension_bits = 0;
  bool Extensions(uint64_t* ext) {
    *ext = U64();
    if (*ext != 0) {
      for (uint64_t rem = *ext; rem != 0; rem &= rem - 1) {
        const uint64_t bits = U64();
        if (total_extension_bits + bits < total_extension_bits) return false;
        total_extension_bits += bits;
      }
      pos_after_ext_size = br->BitsConsumed();
    }
    if (pos_after_ext_size == 0) return true;
    if (!br->Healthy()) return false;
    const uint64_t end = pos_after_ext_size + total_extension_bits;
    if (end < pos_after_ext_size) return false;
    const uint64_t bits_read = br->BitsConsumed();
    if (bits_read > end) return false;
    uint64_t remaining = end - bits_read;
    while (remaining > 0) {
      const uint32_t n = remaining > 32 ? 32u : (uint32_t)remaining;
      br->Read(n);
      remaining -= n;
      if (!br->Healthy()) return false;
    }
    return true;
  }
};

int ReadLoopFilter(FieldReader* r, bool is_modular, jxlhip_frame_header* h) {  // loop_filter.cc:18-106
  jxlhip_loop_filter& lf = h->lf;
  // defaults
  lf.gab = 1;
  const float w1 = (float)(1.1 * 0.104699568f), w2 = (float)(1.1 * 0.055680538f);
  for (int c = 0; c < 3; c++) {
    lf.gab_weights[2 * c] = w1;
    lf.gab_weights[2 * c + 1] = w2;
  }
  lf.epf_iters = 2;
  for (int i = 0; i < 8; i++) lf.epf_sharp_lut[i] = (float)i / 7.0f;
  lf.epf_channel_scale[0] = 40.0f;
  lf.epf_channel_scale[1] = 5.0f;
  lf.epf_channel_scale[2] = 3.5f;
  h->epf_pass1_zeroflush = 0.45f;
  h->epf_pass2_zeroflush = 0.6f;
  lf.epf_quant_mul = 0.46f;
  lf.epf_pass0_sigma_scale = 0.9f;
  lf.epf_pass2_sigma_scale = 6.5f;
  lf.epf_border_sad_mul = 0.6666666666666666f;
  h->epf_sigma_for_modular = 1.0f;
  h->gab_custom = h->epf_sharp_custom = h->epf_weight_custom = h->epf_sigma_custom = 0;
  h->lf_extensions = 0;
  h->lf_all_default = r->Bool();
  if (h->lf_all_default) return kOk;
  lf.gab = r->Bool();
  if (lf.gab) {
    h->gab_custom = r->Bool();
    if (h->gab_custom) {
      for (int c = 0; c < 3; c++) {
        lf.gab_weights[2 * c] = r->F16();
        lf.gab_weights[2 * c + 1] = r->F16();
        if (fabsf(1.0f + (lf.gab_weights[2 * c] + lf.gab_weights[2 * c + 1]) * 4) < 1e-8) return kBad;
      }
    }
  }
  lf.epf_iters = r->Bits(2);
  if (lf.epf_iters > 0) {
    if (!is_modular) {
      h->epf_sharp_custom = r->Bool();
      if (h->epf_sharp_custom)
        for (int i = 0; i < 8; i++) lf.epf_sharp_lut[i] = r->F16();
    }
    h->epf_weight_custom = r->Bool();
    if (h->epf_weight_custom) {
      for (int c = 0; c < 3; c++) lf.epf_channel_scale[c] = r->F16();
      h->epf_pass1_zeroflush = r->F16();
      h->epf_pass2_zeroflush = r->F16();
    }
    h->epf_sigma_custom = r->Bool();
    if (h->epf_sigma_custom) {
      if (!is_modular) lf.epf_quant_mul = r->F16();
      lf.epf_pass0_sigma_scale = r->F16();
      lf.epf_pass2_sigma_scale = r->F16();
      lf.epf_border_sad_mul = r->F16();
    }
    if (is_modular) {
      h->epf_sigma_for_modular = r->F16();
      if (h->epf_sigma_for_modular < 1e-8) return kBad;
    }
  }
  if (!r->ok) return kBad;
  if (!r->Extensions(&h->lf_extensions)) return kBad;
  return kOk;
}

struct Blending {
  uint32_t mode = 0, alpha_channel = 0, clamp = 0, source = 0;
};
int ReadBlending(FieldReader* r, uint32_t num_ec, bool partial, Blending* b) {  // frame_header.cc:65-93
  static const U32Dist kMode = {{0, 0, 0, 2}, {0, 1, 2, 3}};
  b->mode = r->U32(kMode);
  if (b->mode > 4) return kBad;
  const bool blend = b->mode == 2 || b->mode == 3;  // kBlend, kAlphaWeightedAdd
  if (num_ec > 0 && blend) {
    static const U32Dist kAlpha = {{0, 0, 0, 3}, {0, 1, 2, 3}};
    b->alpha_channel = r->U32(kAlpha);
    if (b->alpha_channel >= num_ec) return kBad;
  }
  if ((num_ec > 0 && blend) || b->mode == 4) b->clamp = r->Bool();
  if (b->mode != 0 || partial) {
    static const U32Dist kSource = {{0, 0, 0, 0}, {0, 1, 2, 3}};
    b->source = r->U32(kSource);
  }
  return kOk;
}

uint32_t DivCeilU(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
}  // namespace

int jxlhip_frame_header_decode(const uint8_t* data, size_t size, size_t* bit_pos, const jxlhip_image_info* im,
                               jxlhip_frame_header* h) {
  if (!data || !bit_pos || !im || !h || im->num_extra_channels > 4096) return JXLHIP_ERR_INVALID_ARGUMENT;
  memset(h, 0, sizeof(*h));
  BitReader br(data, size, *bit_pos);
  FieldReader r(&br);
  // defaults of every field (Bundle::Init / SetDefault)
  h->upsampling = 1;
  h->group_size_shift = 1;
  h->x_qm_scale = 3;
  h->b_qm_scale = 2;
  h->num_passes = 1;
  for (uint32_t& u : h->ec_upsampling) u = 1;
  h->image_bits = im->bits_per_sample;
  h->is_last = 1;
  h->color_transform = JXLHIP_CT_XYB;
  const bool xyb = im->xyb_encoded != 0;
  const uint32_t num_ec = im->num_extra_channels;
  bool is_partial = false;
  int rc = kOk;
  h->all_default = r.Bool();
  if (h->all_default) {
    // SetDefault (fields.cc:88-118 visiting every field with its default): a regular, last VarDCT
    // frame, one pass, default loop filter; the colour transform follows the image
    h->color_transform = xyb ? JXLHIP_CT_XYB : JXLHIP_CT_NONE;
    static const uint8_t kOne = 1;  // "all_default = 1" for the nested loop filter
    BitReader lbr(&kOne, 1, 0);
    FieldReader lr(&lbr);
    ReadLoopFilter(&lr, false, h);
    h->save_before_color_transform = 0;
  } else {
    static const U32Dist kType = {{0, 0, 0, 0}, {0, 1, 2, 3}};
    h->frame_type = r.U32(kType);
    if (im->is_preview && h->frame_type != JXLHIP_FRAME_REGULAR) return kBad;
    h->is_modular = r.Bool();
    h->flags = r.U64();
    if (xyb) {
      h->color_transform = JXLHIP_CT_XYB;
    } else {
      h->color_transform = r.Bool() ? JXLHIP_CT_YCBCR : JXLHIP_CT_NONE;
    }
    const bool use_dc_frame = (h->flags & JXLHIP_FLAG_USE_DC_FRAME) != 0;
    if (h->color_transform == JXLHIP_CT_YCBCR && !use_dc_frame)
      for (int c = 0; c < 3; c++) h->chroma_mode[c] = r.Bits(2);
    static const U32Dist kPow2 = {{0, 0, 0, 0}, {1, 2, 4, 8}};
    if (!use_dc_frame) {
      h->upsampling = r.U32(kPow2);
      for (uint32_t i = 0; i < num_ec; i++) {
        const uint32_t ds = im->ec_dim_shift ? im->ec_dim_shift[i] : 0;
        uint32_t ec = r.U32(kPow2);
        ec <<= ds;
        if (ec < h->upsampling || ec > 8) return kBad;
        if (!br.Healthy()) return kBad;
        if (i < 4) h->ec_upsampling[i] = ec;
      }
    }
    if (h->is_modular) h->group_size_shift = r.Bits(2);
    if (!h->is_modular && h->color_transform == JXLHIP_CT_XYB) {
      h->x_qm_scale = r.Bits(3);
      h->b_qm_scale = r.Bits(3);
    } else {
      h->x_qm_scale = h->b_qm_scale = 2;
    }
    if (h->frame_type != JXLHIP_FRAME_REFERENCE_ONLY) {  // Passes::VisitFields, frame_header.cc:137-176
      static const U32Dist kNumPasses = {{0, 0, 0, 3}, {1, 2, 3, 4}};
      h->num_passes = r.U32(kNumPasses);
      if (h->num_passes != 1) {
        static const U32Dist kNumDs = {{0, 0, 0, 1}, {0, 1, 2, 3}};
        h->num_downsample = r.U32(kNumDs);
        if (h->num_downsample > h->num_passes) return kBad;
        for (uint32_t i = 0; i + 1 < h->num_passes; i++) h->shift[i] = r.Bits(2);
        h->shift[h->num_passes - 1] = 0;
        for (uint32_t i = 0; i < h->num_downsample; i++) {
          h->downsample[i] = r.U32(kPow2);
          if (i > 0 && h->downsample[i] >= h->downsample[i - 1]) return kBad;
        }
        static const U32Dist kLastPass = {{0, 0, 0, 3}, {0, 1, 2, 0}};
        for (uint32_t i = 0; i < h->num_downsample; i++) {
          h->last_pass[i] = r.U32(kLastPass);
          if (i > 0 && h->last_pass[i] <= h->last_pass[i - 1]) return kBad;
          if (h->last_pass[i] >= h->num_passes) return kBad;
        }
      }
    }
    if (h->frame_type == JXLHIP_FRAME_DC) {
      static const U32Dist kLevel = {{0, 0, 0, 0}, {1, 2, 3, 4}};
      h->dc_level = r.U32(kLevel);
    }
    if (h->frame_type != JXLHIP_FRAME_DC) {
      h->custom_size_or_origin = r.Bool();
      if (h->custom_size_or_origin) {
        static const U32Dist kSize = {{8, 11, 14, 30}, {0, 256, 2304, 18688}};
        const bool regular = h->frame_type == JXLHIP_FRAME_REGULAR || h->frame_type == JXLHIP_FRAME_SKIP_PROGRESSIVE;
        if (regular) {
          const uint32_t ux = r.U32(kSize), uy = r.U32(kSize);
          h->x0 = (int32_t)((ux >> 1) ^ (~(ux & 1) + 1));  // UnpackSigned
          h->y0 = (int32_t)((uy >> 1) ^ (~(uy & 1) + 1));
        }
        h->coded_xsize = r.U32(kSize);
        h->coded_ysize = r.U32(kSize);
        if (h->coded_xsize == 0 || h->coded_ysize == 0) return kBad;
        if (regular) {
          is_partial |= h->x0 > 0;
          is_partial |= h->y0 > 0;
          is_partial |= (int32_t)h->coded_xsize + h->x0 < (int32_t)im->xsize;
          is_partial |= (int32_t)h->coded_ysize + h->y0 < (int32_t)im->ysize;
        }
      }
    }
    bool replace_all = true;
    if (h->frame_type == JXLHIP_FRAME_REGULAR || h->frame_type == JXLHIP_FRAME_SKIP_PROGRESSIVE) {
      Blending b;
      if ((rc = ReadBlending(&r, num_ec, is_partial, &b))) return rc;
      h->blend_mode = b.mode;
      h->blend_alpha_channel = b.alpha_channel;
      h->blend_clamp = b.clamp;
      h->blend_source = b.source;
      replace_all = b.mode == 0;
      for (uint32_t i = 0; i < num_ec; i++) {
        Blending e;
        if ((rc = ReadBlending(&r, num_ec, is_partial, &e))) return rc;
        replace_all &= e.mode == 0;
        if (!br.Healthy()) return kBad;
      }
      if (im->is_preview && (!replace_all || h->custom_size_or_origin)) return kBad;
      if (im->have_animation) {  // AnimationFrame, frame_header.cc:120-133
        static const U32Dist kDuration = {{0, 0, 8, 32}, {0, 1, 0, 0}};
        h->duration = r.U32(kDuration);
        if (im->have_timecodes) h->timecode = r.Bits(32);
      }
      h->is_last = r.Bool();
    } else {
      h->is_last = 0;
    }
    if (h->frame_type != JXLHIP_FRAME_DC && !h->is_last) {
      static const U32Dist kRef = {{0, 0, 0, 0}, {0, 1, 2, 3}};
      h->save_as_reference = r.U32(kRef);
    }
    if (h->frame_type != JXLHIP_FRAME_DC) {
      const bool can_be_referenced = !h->is_last && (h->duration == 0 || h->save_as_reference != 0);
      const bool regular = h->frame_type == JXLHIP_FRAME_REGULAR || h->frame_type == JXLHIP_FRAME_SKIP_PROGRESSIVE;
      if (can_be_referenced && h->blend_mode == 0 && !is_partial && regular) {
        h->save_before_color_transform = r.Bool();
      } else if (h->frame_type == JXLHIP_FRAME_REFERENCE_ONLY) {
        h->save_before_color_transform = r.Bool();
        const uint32_t fx = h->custom_size_or_origin ? h->coded_xsize : im->xsize;
        const uint32_t fy = h->custom_size_or_origin ? h->coded_ysize : im->ysize;
        if (!h->save_before_color_transform && (fx < im->xsize || fy < im->ysize || h->x0 != 0 || h->y0 != 0))
          return kBad;
      }
    } else {
      h->save_before_color_transform = 1;
    }
    {  // VisitNameString, frame_header.h:35-50
      static const U32Dist kName = {{0, 4, 5, 10}, {0, 0, 16, 48}};
      h->name_length = r.U32(kName);
      for (uint32_t i = 0; i < h->name_length; i++) {
        r.Bits(8);
        if (!br.Healthy()) return kBad;
      }
    }
    if ((rc = ReadLoopFilter(&r, h->is_modular != 0, h))) return rc;
    if (!r.Extensions(&h->extensions)) return kBad;
  }
  if (!r.ok || !br.Healthy()) return kBad;
  // FrameHeader::ToFrameDimensions
  uint32_t xs = h->coded_xsize ? h->coded_xsize : im->xsize, ys = h->coded_ysize ? h->coded_ysize : im->ysize;
  if (h->dc_level != 0) {
    xs = DivCeilU(xs, 1u << (3 * h->dc_level));
    ys = DivCeilU(ys, 1u << (3 * h->dc_level));
  }
  static const uint8_t kHShift[4] = {0, 1, 1, 0}, kVShift[4] = {0, 1, 0, 1};
  uint32_t maxh = 0, maxv = 0;
  for (int c = 0; c < 3; c++) {
    maxh = std::max<uint32_t>(maxh, kHShift[h->chroma_mode[c]]);
    maxv = std::max<uint32_t>(maxv, kVShift[h->chroma_mode[c]]);
  }
  h->group_dim = (256u >> 1) << h->group_size_shift;
  h->xsize = DivCeilU(xs, h->upsampling);
  h->ysize = DivCeilU(ys, h->upsampling);
  h->xsize_blocks = DivCeilU(h->xsize, 8u << maxh) << maxh;
  h->ysize_blocks = DivCeilU(h->ysize, 8u << maxv) << maxv;
  h->xsize_groups = DivCeilU(h->xsize, h->group_dim);
  h->ysize_groups = DivCeilU(h->ysize, h->group_dim);
  h->num_groups = (uint64_t)h->xsize_groups * h->ysize_groups;
  h->num_dc_groups = (uint64_t)DivCeilU(h->xsize_blocks, h->group_dim) * DivCeilU(h->ysize_blocks, h->group_dim);
  h->num_toc_entries =
      (h->num_groups == 1 && h->num_passes == 1) ? 1 : 2 + h->num_dc_groups + h->num_groups * h->num_passes;
  h->x_dm_multiplier = powf(1 / (1.25f), (float)h->x_qm_scale - 2.0f);  // dec_cache.h:161-162
  h->b_dm_multiplier = powf(1 / (1.25f), (float)h->b_qm_scale - 2.0f);
  h->num_extra_channels = im->num_extra_channels;
  *bit_pos = br.BitsConsumed();
  return kOk;
}

// ---- image header: SizeHeader, ImageMetadata, CustomTransformData (include/jxl_hip_frame.h) ----
namespace {
// Visitor::Enum (fields.h:205-216) + EnumValid (field_encodings.h:119-131)
bool ReadEnum(FieldReader* r, uint64_t valid, uint32_t* v) {
  static const U32Dist kEnum = {{0, 0, 4, 6}, {0, 1, 2, 18}};
  *v = r->U32(kEnum);
  return *v < 64 && ((valid >> *v) & 1) != 0;
}

uint32_t AspectRatioX(uint32_t ratio, uint32_t ysize) {  // headers.cc:36-47,61-63
  static const uint32_t kNum[7] = {1, 12, 4, 3, 16, 5, 2}, kDen[7] = {1, 10, 3, 2, 9, 4, 1};
  return (uint32_t)((uint64_t)ysize * kNum[ratio - 1] / kDen[ratio - 1]);
}

void ReadSizeHeader(FieldReader* r, uint32_t* xsize, uint32_t* ysize) {  // headers.cc:127-152
  static const U32Dist kDim = {{9, 13, 18, 30}, {1, 1, 1, 1}};
  const bool small = r->Bool();
  *ysize = small ? (r->Bits(5) + 1) * 8 : r->U32(kDim);
  const uint32_t ratio = r->Bits(3);
  if (ratio != 0) {
    *xsize = AspectRatioX(ratio, *ysize);
  } else {
    *xsize = small ? (r->Bits(5) + 1) * 8 : r->U32(kDim);
  }
}

void ReadPreviewHeader(FieldReader* r, uint32_t* xsize, uint32_t* ysize) {  // headers.cc:154-180
  static const U32Dist kDiv8 = {{0, 0, 5, 9}, {16, 32, 1, 33}}, kFull = {{6, 8, 10, 12}, {1, 65, 321, 1345}};
  const bool div8 = r->Bool();
  *ysize = div8 ? r->U32(kDiv8) * 8 : r->U32(kFull);
  const uint32_t ratio = r->Bits(3);
  if (ratio != 0) {
    *xsize = AspectRatioX(ratio, *ysize);
  } else {
    *xsize = div8 ? r->U32(kDiv8) * 8 : r->U32(kFull);
  }
}

bool ReadBitDepth(FieldReader* r, jxlhip_bit_depth* b) {  // image_metadata.cc:25-61
  b->floating_point_sample = r->Bool();
  if (!b->floating_point_sample) {
    static const U32Dist kInt = {{0, 0, 0, 6}, {8, 10, 12, 1}};
    b->bits_per_sample = r->U32(kInt);
    b->exponent_bits_per_sample = 0;
    return b->bits_per_sample <= 31;
  }
  static const U32Dist kFloat = {{0, 0, 0, 6}, {32, 16, 24, 1}};
  b->bits_per_sample = r->U32(kFloat);
  b->exponent_bits_per_sample = r->Bits(4) + 1;
  if (b->exponent_bits_per_sample < 2 || b->exponent_bits_per_sample > 8) return false;
  const int mantissa = (int)b->bits_per_sample - (int)b->exponent_bits_per_sample - 1;
  return mantissa >= 2 && mantissa <= 23;
}

bool ReadExtraChannel(FieldReader* r, jxlhip_extra_channel* e) {  // image_metadata.cc:199-245
  memset(e, 0, sizeof(*e));
  e->bit_depth.bits_per_sample = 8;
  e->cfa_channel = 1;
  e->all_default = r->Bool();
  if (e->all_default) return true;  // an 8-bit alpha channel, not associated
  constexpr uint64_t kTypes = 0x7Full | (1ull << 15) | (1ull << 16);  // EnumBits(ExtraChannel), image_metadata.h:75-80
  if (!ReadEnum(r, kTypes, &e->type)) return false;
  if (!ReadBitDepth(r, &e->bit_depth)) return false;
  static const U32Dist kShift = {{0, 0, 0, 3}, {0, 3, 4, 1}};
  e->dim_shift = r->U32(kShift);
  if ((1u << e->dim_shift) > 8) return false;
  static const U32Dist kName = {{0, 4, 5, 10}, {0, 0, 16, 48}};  // VisitNameString, frame_header.h:35-50
  e->name_length = r->U32(kName);
  for (uint32_t i = 0; i < e->name_length; i++) {
    r->Bits(8);
    if (!r->br->Healthy()) return false;
  }
  if (e->type == JXLHIP_EC_ALPHA) e->alpha_associated = r->Bool();
  if (e->type == JXLHIP_EC_SPOT_COLOR) {
    for (float& c : e->spot_color) c = r->F16();
  }
  if (e->type == JXLHIP_EC_CFA) {
    static const U32Dist kCfa = {{0, 2, 4, 8}, {1, 0, 3, 19}};
    e->cfa_channel = r->U32(kCfa);
  }
  // kUnknown is a valid code that the reference nevertheless refuses (image_metadata.cc:236-243)
  return r->ok && e->type != JXLHIP_EC_UNKNOWN;
}

int32_t ReadCustomXy(FieldReader* r) {  // color_encoding_internal.cc:106-120, pack_signed.h
  static const U32Dist kXy = {{19, 19, 20, 21}, {0, 524288, 1048576, 2097152}};
  const uint32_t u = r->U32(kXy);
  return (int32_t)((u >> 1) ^ (0u - (u & 1)));
}

// ---- can the reference synthesise an ICC profile for these custom chromaticities? ----
// ColorEncoding::VisitFields ends in CreateICC() (color_encoding_internal.cc:207), so a header whose
// CUSTOM white point / primaries make MaybeCreateProfileImpl fail is rejected by the reference.  Only
// the conditions are restated here (no profile is written), in the reference's own mix of float and
// double arithmetic: cms/jxl_cms_internal.h:43-126 (PrimariesToXYZ, AdaptToXYZD50), :235-244
// (CIEXYZFromWhiteCIExy), :354-372, :403-409 (the s15Fixed16 range), base/matrix_ops.h.
typedef float Mat3[3][3];
bool S15Fixed16Ok(float v) { return -32767.995f <= v && v <= 32767.995f; }

void MulMat(const Mat3 a, const Mat3 b, Mat3 c) {
  for (int x = 0; x < 3; x++) {
    const double t[3] = {b[0][x], b[1][x], b[2][x]};
    for (int y = 0; y < 3; y++) c[y][x] = (float)(a[y][0] * t[0] + a[y][1] * t[1] + a[y][2] * t[2]);
  }
}

void MulVec(const Mat3 a, const float b[3], float c[3]) {
  for (int y = 0; y < 3; y++) {
    double e = 0;
    for (int x = 0; x < 3; x++) e += (double)a[y][x] * b[x];
    c[y] = (float)e;
  }
}

bool InvMat(Mat3 m) {
  double t[3][3];
  t[0][0] = (double)m[1][1] * m[2][2] - (double)m[1][2] * m[2][1];
  t[0][1] = (double)m[0][2] * m[2][1] - (double)m[0][1] * m[2][2];
  t[0][2] = (double)m[0][1] * m[1][2] - (double)m[0][2] * m[1][1];
  t[1][0] = (double)m[1][2] * m[2][0] - (double)m[1][0] * m[2][2];
  t[1][1] = (double)m[0][0] * m[2][2] - (double)m[0][2] * m[2][0];
  t[1][2] = (double)m[0][2] * m[1][0] - (double)m[0][0] * m[1][2];
  t[2][0] = (double)m[1][0] * m[2][1] - (double)m[1][1] * m[2][0];
  t[2][1] = (double)m[0][1] * m[2][0] - (double)m[0][0] * m[2][1];
  t[2][2] = (double)m[0][0] * m[1][1] - (double)m[0][1] * m[1][0];
  const double det = m[0][0] * t[0][0] + m[0][1] * t[1][0] + m[0][2] * t[2][0];
  if (std::abs(det) < 1e-10) return false;
  const double idet = 1.0 / det;
  for (int j = 0; j < 3; j++) {
    for (int i = 0; i < 3; i++) m[j][i] = (float)(t[j][i] * idet);
  }
  return true;
}

bool WhiteInRange(float wx, float wy) { return (wx >= 0) && (wx <= 1) && (wy > 0) && (wy <= 1); }

bool PrimariesToXyz(const float p[6], float wx, float wy, Mat3 out) {
  if (!WhiteInRange(wx, wy)) return false;
  const Mat3 prim = {{p[0], p[2], p[4]}, {p[1], p[3], p[5]}, {1.0f - p[0] - p[1], 1.0f - p[2] - p[3], 1.0f - p[4] - p[5]}};
  Mat3 inv;
  memcpy(inv, prim, sizeof(inv));
  if (!InvMat(inv)) return false;
  const float w[3] = {wx / wy, 1.0f, (1.0f - wx - wy) / wy};
  if (!std::isfinite(w[0]) || !std::isfinite(w[2])) return false;
  float xyz[3];
  MulVec(inv, w, xyz);
  const Mat3 a = {{xyz[0], 0, 0}, {0, xyz[1], 0}, {0, 0, xyz[2]}};
  MulMat(prim, a, out);
  return true;
}

bool AdaptToXyzD50(float wx, float wy, Mat3 out) {
  static const Mat3 kBradford = {{0.8951f, 0.2664f, -0.1614f}, {-0.7502f, 1.7135f, 0.0367f}, {0.0389f, -0.0685f, 1.0296f}};
  static const Mat3 kBradfordInv = {{0.9869929f, -0.1470543f, 0.1599627f},
                                    {0.4323053f, 0.5183603f, 0.0492912f},
                                    {-0.0085287f, 0.0400428f, 0.9684867f}};
  if (!WhiteInRange(wx, wy)) return false;
  const float w[3] = {wx / wy, 1.0f, (1.0f - wx - wy) / wy};
  if (!std::isfinite(w[0]) || !std::isfinite(w[2])) return false;
  const float w50[3] = {0.96422f, 1.0f, 0.82521f};
  float lms[3], lms50[3];
  MulVec(kBradford, w, lms);
  MulVec(kBradford, w50, lms50);
  if (lms[0] == 0 || lms[1] == 0 || lms[2] == 0) return false;
  const Mat3 a = {{lms50[0] / lms[0], 0, 0}, {0, lms50[1] / lms[1], 0}, {0, 0, lms50[2] / lms[2]}};
  if (!std::isfinite(a[0][0]) || !std::isfinite(a[1][1]) || !std::isfinite(a[2][2])) return false;
  Mat3 b;
  MulMat(a, kBradford, b);
  MulMat(kBradfordInv, b, out);
  return true;
}

bool IccExpressible(const jxlhip_color_encoding& c) {
  const bool custom_white = c.white_point == JXLHIP_WP_CUSTOM;
  const bool rgb = c.color_space == JXLHIP_CS_RGB;
  if (c.color_space == JXLHIP_CS_XYB || !(custom_white || (rgb && c.primaries == JXLHIP_PRIM_CUSTOM))) return true;
  double wx, wy;  // ColorEncoding::GetWhitePoint, cms/color_encoding_cms.h:435-465
  switch (c.white_point) {
    case JXLHIP_WP_CUSTOM: wx = c.white_xy[0] * (1.0 / 1000000), wy = c.white_xy[1] * (1.0 / 1000000); break;
    case JXLHIP_WP_DCI: wx = 0.314, wy = 0.351; break;
    case JXLHIP_WP_E: wx = wy = 1.0 / 3; break;
    default: wx = 0.3127, wy = 0.3290; break;
  }
  if (c.color_space == JXLHIP_CS_GRAY) {  // the white point tag
    if (std::abs(wy) < 1e-12) return false;
    const float factor = (float)(1 / wy);
    return S15Fixed16Ok((float)(wx * factor)) && S15Fixed16Ok((float)((1 - wx - wy) * factor));
  }
  if (wy == 0) return false;
  Mat3 chad;  // the chromatic adaptation tag
  if (!AdaptToXyzD50((float)wx, (float)wy, chad)) return false;
  for (int j = 0; j < 3; j++) {
    for (int i = 0; i < 3; i++) {
      if (!S15Fixed16Ok(chad[j][i])) return false;
    }
  }
  if (!rgb) return true;
  static const double kSrgb[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};
  static const double k2100[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046};
  static const double kP3[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060};
  float p[6];  // GetPrimaries, cms/color_encoding_cms.h:354-395
  for (int i = 0; i < 6; i++) {
    p[i] = (float)(c.primaries == JXLHIP_PRIM_CUSTOM ? c.primaries_xy[i] * (1.0 / 1000000)
                   : c.primaries == JXLHIP_PRIM_2100 ? k2100[i]
                   : c.primaries == JXLHIP_PRIM_P3   ? kP3[i]
                                                     : kSrgb[i]);
  }
  Mat3 to_xyz, d50, m;  // the rXYZ / gXYZ / bXYZ tags
  if (!PrimariesToXyz(p, (float)wx, (float)wy, to_xyz) || !AdaptToXyzD50((float)wx, (float)wy, d50)) return false;
  MulMat(d50, to_xyz, m);
  for (int j = 0; j < 3; j++) {
    for (int i = 0; i < 3; i++) {
      if (!S15Fixed16Ok(m[j][i])) return false;
    }
  }
  return true;
}

bool ReadColorEncoding(FieldReader* r, jxlhip_color_encoding* c) {  // color_encoding_internal.cc:148-216
  memset(c, 0, sizeof(*c));
  c->white_point = JXLHIP_WP_D65;
  c->primaries = JXLHIP_PRIM_SRGB;
  c->transfer_function = 13;
  c->rendering_intent = 1;
  c->gamma = 10000000;
  c->all_default = r->Bool();
  if (c->all_default) return true;
  c->want_icc = r->Bool();
  if (!ReadEnum(r, 0xF, &c->color_space)) return false;
  if (c->want_icc) return true;  // the fields are not coded; the profile follows the headers
  constexpr uint64_t kWhite = (1ull << 1) | (1ull << 2) | (1ull << 10) | (1ull << 11);
  constexpr uint64_t kPrim = (1ull << 1) | (1ull << 2) | (1ull << 9) | (1ull << 11);
  constexpr uint64_t kTf = (1ull << 1) | (1ull << 2) | (1ull << 8) | (1ull << 13) | (1ull << 16) | (1ull << 17) |
                           (1ull << 18);
  if (c->color_space != JXLHIP_CS_XYB) {  // XYB: implicitly D65
    if (!ReadEnum(r, kWhite, &c->white_point)) return false;
    if (c->white_point == JXLHIP_WP_CUSTOM) {
      c->white_xy[0] = ReadCustomXy(r);
      c->white_xy[1] = ReadCustomXy(r);
    }
  }
  if (c->color_space != JXLHIP_CS_GRAY && c->color_space != JXLHIP_CS_XYB) {
    if (!ReadEnum(r, kPrim, &c->primaries)) return false;
    if (c->primaries == JXLHIP_PRIM_CUSTOM) {
      for (int32_t& v : c->primaries_xy) v = ReadCustomXy(r);
    }
  }
  if (c->color_space == JXLHIP_CS_XYB) {  // CustomTransferFunction::SetImplicit: gamma 1/3
    c->have_gamma = 1;
    c->gamma = 3333333;
  } else {
    c->have_gamma = r->Bool();
    if (c->have_gamma) {
      c->gamma = r->Bits(24);
      if (c->gamma > 10000000u || (uint64_t)c->gamma * 8192 < 10000000u) return false;
    } else if (!ReadEnum(r, kTf, &c->transfer_function)) {
      return false;
    }
  }
  if (!ReadEnum(r, 0xF, &c->rendering_intent)) return false;
  if (c->color_space == JXLHIP_CS_UNKNOWN || (!c->have_gamma && c->transfer_function == 2)) return false;
  // MaybeCreateProfileImpl (cms/jxl_cms_internal.h:989-995): an XYB profile exists for one intent only
  if (c->color_space == JXLHIP_CS_XYB && c->rendering_intent != 0) return false;
  return IccExpressible(*c);
}
}  // namespace

// OutputEncodingInfo::SetFromMetadata + SetColorEncoding (dec_xyb.cc:144-165,180-249) for an XYB image decoded into
// its ORIGINAL colour space: the inverse opsin matrix lands in linear sRGB (D65); for other primaries / white points
// it is followed by sRGB -> XYZ(D50) -> the original space, so that the kernels' 3x3 product comes out in the
// original primaries; luminances = the Y row of the original space's RGB -> XYZ matrix (the HLG OOTF's weights).
int jxlhip_output_opsin_matrix(const jxlhip_image_header* ih, float inverse_matrix[9], float luminances[3]) {
  if (!ih || !inverse_matrix || !luminances) return JXLHIP_ERR_INVALID_ARGUMENT;
  const jxlhip_color_encoding& c = ih->color_encoding;
  memcpy(inverse_matrix, ih->inverse_opsin_matrix, 9 * sizeof(float));
  luminances[0] = 0.2126f, luminances[1] = 0.7152f, luminances[2] = 0.0722f;
  if (!ih->xyb_encoded) return JXLHIP_ERR_UNSUPPORTED;
  if (c.all_default) return kOk;
  // An ICC original: without a CMS the reference cannot reach the profile's space and falls back to linear sRGB, grey
  // for a grey profile (SetFromMetadata, dec_xyb.cc:160-164) -- the coded matrix, or its luminance rows
  if (c.want_icc && c.color_space != JXLHIP_CS_GRAY) return kOk;
  if (c.color_space == JXLHIP_CS_GRAY) {
    // a grey original (D65: CanOutputToColorEncoding, dec_xyb.cc:137-140): every output channel is the luminance of
    // the linear sRGB pixel -- the matrix' rows become luminances x matrix (:226-230)
    if (!c.want_icc && c.white_point != JXLHIP_WP_D65) return JXLHIP_ERR_UNSUPPORTED;
    const Mat3 luma = {{luminances[0], luminances[1], luminances[2]},
                       {luminances[0], luminances[1], luminances[2]},
                       {luminances[0], luminances[1], luminances[2]}};
    Mat3 orig, out;
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < 3; i++) orig[j][i] = ih->inverse_opsin_matrix[j * 3 + i];
    MulMat(luma, orig, out);
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < 3; i++) inverse_matrix[j * 3 + i] = out[j][i];
    return kOk;
  }
  if (c.color_space != JXLHIP_CS_RGB) return JXLHIP_ERR_UNSUPPORTED;
  // CanOutputToColorEncoding: every enumerated transfer function the kernels have (the caller picks it)
  if (!c.have_gamma && c.transfer_function != 1 && c.transfer_function != 8 && c.transfer_function != 13 &&
      c.transfer_function != 16 && c.transfer_function != 17 && c.transfer_function != 18)
    return JXLHIP_ERR_UNSUPPORTED;
  if (c.primaries == JXLHIP_PRIM_SRGB && c.white_point == JXLHIP_WP_D65) return kOk;
  double wx, wy;
  switch (c.white_point) {
    case JXLHIP_WP_CUSTOM: wx = c.white_xy[0] * (1.0 / 1000000), wy = c.white_xy[1] * (1.0 / 1000000); break;
    case JXLHIP_WP_DCI: wx = 0.314, wy = 0.351; break;
    case JXLHIP_WP_E: wx = wy = 1.0 / 3; break;
    default: wx = 0.3127, wy = 0.3290; break;
  }
  static const double kSrgb[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};
  static const double k2100[6] = {0.708, 0.292, 0.170, 0.797, 0.131, 0.046};
  static const double kP3[6] = {0.680, 0.320, 0.265, 0.690, 0.150, 0.060};
  float ps[6], po[6];
  for (int i = 0; i < 6; i++) {
    ps[i] = (float)kSrgb[i];
    po[i] = (float)(c.primaries == JXLHIP_PRIM_CUSTOM ? c.primaries_xy[i] * (1.0 / 1000000)
                    : c.primaries == JXLHIP_PRIM_2100 ? k2100[i]
                    : c.primaries == JXLHIP_PRIM_P3   ? kP3[i]
                                                      : kSrgb[i]);
  }
  Mat3 srgb_xyz, srgb_d50, srgb_to_xyzd50, original_to_xyz, adapt, xyzd50_to_original, srgb_to_original, orig, out;
  if (!PrimariesToXyz(ps, 0.3127f, 0.3290f, srgb_xyz) || !AdaptToXyzD50(0.3127f, 0.3290f, srgb_d50)) return kBad;
  MulMat(srgb_d50, srgb_xyz, srgb_to_xyzd50);
  if (!PrimariesToXyz(po, (float)wx, (float)wy, original_to_xyz)) return kBad;
  for (int i = 0; i < 3; i++) luminances[i] = original_to_xyz[1][i];
  if (!AdaptToXyzD50((float)wx, (float)wy, adapt)) return kBad;
  MulMat(adapt, original_to_xyz, xyzd50_to_original);
  if (!InvMat(xyzd50_to_original)) return kBad;
  MulMat(xyzd50_to_original, srgb_to_xyzd50, srgb_to_original);
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) orig[j][i] = ih->inverse_opsin_matrix[j * 3 + i];
  MulMat(srgb_to_original, orig, out);
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) inverse_matrix[j * 3 + i] = out[j][i];
  return kOk;
}

int jxlhip_image_header_decode(const uint8_t* data, size_t size, size_t* bit_pos, jxlhip_extra_channel* extra,
                               size_t extra_capacity, jxlhip_image_header* h) {
  if (!data || !bit_pos || !h || (extra_capacity && !extra)) return JXLHIP_ERR_INVALID_ARGUMENT;
  memset(h, 0, sizeof(*h));
  if (size < 2 || data[0] != 0xFF || data[1] != 0x0A) return kBad;  // decode.cc:1049
  {
    BitReader br(data, size, 16);
    FieldReader r(&br);  // one visitor per Bundle::Read
    ReadSizeHeader(&r, &h->xsize, &h->ysize);
    if (!br.Healthy()) return kBad;
    *bit_pos = br.BitsConsumed();
  }
  // ImageMetadata (image_metadata.cc:270-338); defaults first
  h->orientation = 1;
  h->bit_depth.bits_per_sample = 8;
  h->modular_16_bit_buffer_sufficient = 1;
  h->xyb_encoded = 1;
  h->tps_numerator = 100;
  h->tps_denominator = 1;
  h->tone_mapping_all_default = 1;
  h->intensity_target = 255.0f;
  {
    BitReader br(data, size, *bit_pos);
    FieldReader r(&br);
    h->all_default = r.Bool();
    if (h->all_default) {
      static const uint8_t kOne = 1;
      BitReader cbr(&kOne, 1, 0);
      FieldReader cr(&cbr);
      ReadColorEncoding(&cr, &h->color_encoding);
    } else {
      const bool extra_fields = r.Bool();
      if (extra_fields) {
        h->orientation = r.Bits(3) + 1;
        h->have_intrinsic_size = r.Bool();
        if (h->have_intrinsic_size) ReadSizeHeader(&r, &h->intrinsic_xsize, &h->intrinsic_ysize);
        h->have_preview = r.Bool();
        if (h->have_preview) ReadPreviewHeader(&r, &h->preview_xsize, &h->preview_ysize);
        h->have_animation = r.Bool();
        if (h->have_animation) {  // AnimationHeader, headers.cc:183-196
          static const U32Dist kNum = {{0, 0, 10, 30}, {100, 1000, 1, 1}}, kDen = {{0, 0, 8, 10}, {1, 1001, 1, 1}},
                               kLoops = {{0, 3, 16, 32}, {0, 0, 0, 0}};
          h->tps_numerator = r.U32(kNum);
          h->tps_denominator = r.U32(kDen);
          h->num_loops = r.U32(kLoops);
          h->have_timecodes = r.Bool();
        }
      }
      if (!ReadBitDepth(&r, &h->bit_depth)) return kBad;
      h->modular_16_bit_buffer_sufficient = r.Bool();
      static const U32Dist kNumEc = {{0, 0, 4, 12}, {0, 1, 2, 1}};
      h->num_extra_channels = r.U32(kNumEc);
      for (uint32_t i = 0; i < h->num_extra_channels; i++) {
        jxlhip_extra_channel e;
        if (!ReadExtraChannel(&r, &e) || !br.Healthy()) return kBad;
        if (i < extra_capacity) extra[i] = e;
      }
      h->xyb_encoded = r.Bool();
      if (!ReadColorEncoding(&r, &h->color_encoding)) return kBad;
      if (extra_fields) {  // ToneMapping, image_metadata.cc:359-392
        h->tone_mapping_all_default = r.Bool();
        if (!h->tone_mapping_all_default) {
          h->intensity_target = r.F16();
          if (!r.ok || !(h->intensity_target > 0.f)) return kBad;
          h->min_nits = r.F16();
          if (!r.ok || h->min_nits < 0.f || h->min_nits > h->intensity_target) return kBad;
          h->relative_to_max_display = r.Bool();
          h->linear_below = r.F16();
          if (!r.ok || h->linear_below < 0.f || (h->relative_to_max_display && h->linear_below > 1.0f)) return kBad;
        }
      }
      if (!r.Extensions(&h->extensions)) return kBad;
    }
    if (!r.ok || !br.Healthy()) return kBad;
    *bit_pos = br.BitsConsumed();
  }
  // CustomTransformData (image_metadata.cc:72-197) with its nested OpsinInverseMatrix (340-357)
  static const float kInverseOpsin[9] = {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f,
                                         -3.254147380392157f, 4.418770392156863f,  -0.16462299647058826f,
                                         -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
  memcpy(h->inverse_opsin_matrix, kInverseOpsin, sizeof(kInverseOpsin));
  for (float& b : h->opsin_biases) b = -0.0037930732552754493f;
  h->quant_biases[0] = 1.0f - 0.05465007330715401f;
  h->quant_biases[1] = 1.0f - 0.07005449891748593f;
  h->quant_biases[2] = 1.0f - 0.049935103337343655f;
  h->quant_biases[3] = 0.145f;
  h->opsin_all_default = 1;
  {
    BitReader br(data, size, *bit_pos);
    FieldReader r(&br);
    h->transform_all_default = r.Bool();
    if (!h->transform_all_default) {
      if (h->xyb_encoded) {
        h->opsin_all_default = r.Bool();
        if (!h->opsin_all_default) {
          for (float& v : h->inverse_opsin_matrix) v = r.F16();
          for (float& v : h->opsin_biases) v = r.F16();
          for (float& v : h->quant_biases) v = r.F16();
        }
      }
      h->custom_weights_mask = r.Bits(3);
      if (h->custom_weights_mask & 1) {
        for (float& v : h->upsampling2_weights) v = r.F16();
      }
      if (h->custom_weights_mask & 2) {
        for (float& v : h->upsampling4_weights) v = r.F16();
      }
      if (h->custom_weights_mask & 4) {
        for (float& v : h->upsampling8_weights) v = r.F16();
      }
    }
    if (!r.ok || !br.Healthy()) return kBad;
    if (!h->color_encoding.want_icc) {  // decode.cc:1133: the first frame starts on a byte
      const uint32_t rem = (uint32_t)(br.BitsConsumed() % 8);
      if (rem != 0 && br.Read(8 - rem) != 0) return kBad;
      if (!br.Healthy()) return kBad;
    }
    *bit_pos = br.BitsConsumed();
  }
  return kOk;
}


// ---- ICC profile of the original (icc_codec.cc, icc_codec_common.cc) ---------------------------
// The profile travels as a byte stream `enc` = varint(profile size) varint(size of the command part) commands data,
// entropy coded one byte per symbol with a context made of the two bytes before it.  The command part drives the
// reconstruction: header bytes as differences from a predicted header, the tag table from one-byte tag codes, the
// tag data as runs that are copied, de-interleaved or summed onto an order-0/1/2 linear prediction.
namespace {

inline uint32_t IccContext(size_t i, uint8_t b1, uint8_t b2) {  // ICCANSContext, icc_codec_common.cc:19-41,171-174
  if (i <= 128) return 0;
  auto wordlike = [](uint8_t b, uint32_t* kind) {
    if ((uint8_t)((b | 0x20) - 'a') < 26) return *kind = 0, true;
    if ((uint8_t)(b - '0') < 10 || b == '.' || b == ',') return *kind = 1, true;
    return false;
  };
  uint32_t k1, k2;
  if (!wordlike(b1, &k1)) k1 = b1 == 0 ? 2 : b1 == 1 ? 3 : b1 < 16 ? 4 : b1 == 255 ? 6 : b1 > 240 ? 5 : 7;
  if (!wordlike(b2, &k2)) k2 = b2 < 16 ? 2 : b2 > 240 ? 3 : 4;
  return 1 + k1 + 8 * k2;
}

struct IccCursor {  // one of the two read positions inside enc: [pos, end)
  const uint8_t* p;
  size_t pos, end;
  bool ok = true;
  size_t Left() const { return end - pos; }
  uint8_t Byte() {
    if (pos >= end) return ok = false, 0;
    return p[pos++];
  }
  uint64_t Varint() {  // base 128, little endian groups, at most ten bytes (DecodeVarInt, icc_codec.cc:62-88)
    uint64_t v = 0;
    for (int i = 0; i < 10; i++) {
      const uint8_t b = Byte();
      if (!ok) return 0;
      if (i == 9 && (b & 0xFE)) return ok = false, 0;
      v |= (uint64_t)(b & 0x7F) << (7 * i);
      if (!(b & 0x80)) return v;
    }
    return ok = false, 0;
  }
};

// The encoder wrote the `width` interleaved byte planes of a run one after the other; back to sample order
// (Shuffle, icc_codec.cc:36-56): output byte i comes from plane i % width, element i / width, where the first
// n % width planes (all of them when n divides) hold ceil(n / width) bytes
void IccInterleave(const uint8_t* planes, size_t n, size_t width, uint8_t* out) {
  const size_t rows = (n + width - 1) / width;
  size_t src = 0, plane = 0;
  for (size_t i = 0; i < n; i++) {
    out[i] = planes[src];
    src += rows;
    if (src >= n) src = ++plane;
  }
}

constexpr char kIccTagNames[] = "cprtwtptbkptrXYZgXYZbXYZkXYZrTRCgTRCbTRCkTRCchaddescchrmdmnddmddlumi";  // tag codes 4..20
constexpr char kIccTypeNames[] = "XYZ desctextmlucparacurvsf32gbd ";                                     // commands 16..23

int UnpredictIcc(const std::vector<uint8_t>& enc, std::vector<uint8_t>* out) {
  IccCursor cmd{enc.data(), 0, enc.size()};
  const uint64_t osize = cmd.Varint();
  const uint64_t csize = cmd.Varint();
  if (!cmd.ok || (osize >> 32) || (csize >> 32) || csize > cmd.Left()) return kBad;
  if (osize + 65536 < enc.size() || osize > (1u << 28)) return kBad;  // CheckPreamble (:93-113)
  IccCursor dat{enc.data(), cmd.pos + (size_t)csize, enc.size()};
  cmd.end = dat.pos;
  std::vector<uint8_t>& icc = *out;
  icc.clear();
  icc.reserve((size_t)std::min<uint64_t>(osize, 1u << 20));  // (the claimed size is not trusted with an allocation)
  auto put32 = [&](uint64_t v) {
    if (v >> 32) return false;
    for (int s = 24; s >= 0; s -= 8) icc.push_back((uint8_t)(v >> s));
    return true;
  };
  auto put_name = [&](const char* four) { icc.insert(icc.end(), four, four + 4); };
  auto finished = [&]() { return icc.size() == osize && cmd.Left() == 0 && dat.Left() == 0 ? kOk : kBad; };

  // 128 header bytes: differences from a predicted header that learns from the bytes already there
  // (kIccInitialHeaderPrediction + ICCPredictHeader, icc_codec_common.cc:87-138)
  uint8_t guess[128] = {};
  for (int s = 0; s < 4; s++) guess[s] = (uint8_t)(osize >> (24 - 8 * s));
  guess[8] = 4;
  memcpy(guess + 12, "mntrRGB XYZ ", 12);
  memcpy(guess + 36, "acsp", 4);
  guess[70] = 246, guess[71] = 214, guess[73] = 1, guess[78] = 211, guess[79] = 45;
  for (size_t i = 0; i < 128; i++) {
    if (icc.size() == osize) return finished();
    if (i == 8) memcpy(guess + 80, icc.data() + 4, 4);  // the creator field repeats the preferred CMM
    if (i == 41) {
      if (icc[40] == 'A') memcpy(guess + 41, "PPL", 3);
      if (icc[40] == 'M') memcpy(guess + 41, "SFT", 3);
    }
    if (i == 42) {
      if (icc[40] == 'S' && icc[41] == 'G') memcpy(guess + 42, "I ", 2);
      if (icc[40] == 'S' && icc[41] == 'U') memcpy(guess + 42, "NW", 2);
    }
    const uint8_t d = dat.Byte();
    if (!dat.ok) return kBad;
    icc.push_back((uint8_t)(d + guess[i]));
  }
  if (icc.size() == osize) return finished();
  if (cmd.Left() == 0) return kBad;

  // tag table: count + 1, then one command per entry (or per rTRC/gTRC/bTRC, rXYZ/gXYZ/bXYZ triple)
  uint64_t ntags = cmd.Varint();
  if (!cmd.ok) return kBad;
  if (ntags != 0) {
    ntags--;
    if (!put32(ntags)) return kBad;
    uint64_t prev_start = 128 + ntags * 12, prev_size = 0;
    while (cmd.Left() != 0) {
      if (icc.size() > osize) return kBad;
      const uint8_t command = cmd.Byte();
      const uint32_t code = command & 63;
      if (code == 0) break;
      char name[4];
      if (code == 1) {  // spelled out in the data part
        for (char& ch : name) ch = (char)dat.Byte();
        if (!dat.ok) return kBad;
      } else if (code == 2) {
        memcpy(name, "rTRC", 4);
      } else if (code == 3) {
        memcpy(name, "rXYZ", 4);
      } else if (code - 4 < 17) {
        memcpy(name, kIccTagNames + 4 * (code - 4), 4);
      } else {
        return kBad;
      }
      put_name(name);
      uint64_t size = prev_size;
      if ((!memcmp(name + 1, "XYZ", 3) && name[0] && strchr("rgbk", name[0])) || !memcmp(name, "wtpt", 4) ||
          !memcmp(name, "bkpt", 4) || !memcmp(name, "lumi", 4))
        size = 20;
      uint64_t start;
      if (command & 64) {
        start = cmd.Varint();
      } else {
        if (prev_start >> 32) return kBad;
        start = prev_start + prev_size;
      }
      if (!cmd.ok || !put32(start)) return kBad;
      if (command & 128) size = cmd.Varint();
      if (!cmd.ok || !put32(size)) return kBad;
      prev_start = start, prev_size = size;
      if (code == 2) {  // three curves sharing one tag body
        put_name("gTRC"), put32(start), put32(size);
        put_name("bTRC"), put32(start), put32(size);
      } else if (code == 3) {  // three colorants one behind the other
        if ((start + size * 2) >> 32) return kBad;
        put_name("gXYZ"), put32(start + size), put32(size);
        put_name("bXYZ"), put32(start + size * 2), put32(size);
      }
    }
  }

  // tag data
  std::vector<uint8_t> run;
  while (cmd.Left() != 0) {
    if (icc.size() > osize) return kBad;
    const uint8_t command = cmd.Byte();
    if (command == 1 || command == 2 || command == 3) {  // verbatim / 2 planes / 4 planes
      const uint64_t n = cmd.Varint();
      if (!cmd.ok || n > dat.Left()) return kBad;
      const size_t at = icc.size();
      icc.resize(at + (size_t)n);
      if (command == 1) {
        if (n) memcpy(icc.data() + at, dat.p + dat.pos, (size_t)n);
      } else {
        IccInterleave(dat.p + dat.pos, (size_t)n, command == 2 ? 2 : 4, icc.data() + at);
      }
      dat.pos += (size_t)n;
    } else if (command == 4) {  // residuals of a linear prediction over samples `width` bytes wide, `stride` apart
      if (cmd.Left() < 2) return kBad;
      const uint8_t flags = cmd.Byte();
      const size_t width = (flags & 3) + 1;
      const int order = (flags >> 2) & 3;
      if (width == 3 || order == 3) return kBad;
      uint64_t stride = width;
      if (flags & 16) {
        stride = cmd.Varint();
        if (!cmd.ok || stride < width) return kBad;
      }
      if (icc.empty() || ((icc.size() - 1) >> 2) < stride) return kBad;  // three samples back must exist
      const uint64_t n = cmd.Varint();
      if (!cmd.ok || n > dat.Left()) return kBad;
      run.resize((size_t)n);
      if (width > 1) IccInterleave(dat.p + dat.pos, (size_t)n, width, run.data());
      else if (n) memcpy(run.data(), dat.p + dat.pos, (size_t)n);
      dat.pos += (size_t)n;
      const size_t start = icc.size(), st = (size_t)stride;
      icc.resize(start + (size_t)n);
      uint8_t* d = icc.data();
      // big-endian sample ending before `limit` (a sample cut short by the bytes written so far reads as 0,
      // DecodeUint32's bound in LinearPredictICCValue, icc_codec_common.cc:145-169)
      for (size_t i = 0; i < (size_t)n; i++) {
        const size_t at = start + i, base = start + (i & ~(width - 1));
        auto sample = [&](size_t back) -> uint32_t {
          const size_t q = base - back * st;
          if (width == 1) return d[at - back * st];
          if (width == 2) return ((uint32_t)d[q] << 8) | d[q + 1];
          if (q + 4 > at) return 0;
          return ((uint32_t)d[q] << 24) | ((uint32_t)d[q + 1] << 16) | ((uint32_t)d[q + 2] << 8) | d[q + 3];
        };
        uint32_t pred;
        if (order == 0) pred = sample(1);
        else if (order == 1) pred = 2 * sample(1) - sample(2);
        else pred = 3 * sample(1) - 3 * sample(2) + sample(3);
        const uint32_t byte_in_sample = (uint32_t)(i & (width - 1));
        const uint8_t pb = (uint8_t)(pred >> (8 * (width - 1 - byte_in_sample)));
        d[at] = (uint8_t)(pb + run[i]);
      }
    } else if (command == 10) {  // an 'XYZ ' tag body: signature, reserved, 12 data bytes
      if (dat.Left() < 12) return kBad;
      put_name("XYZ ");
      icc.insert(icc.end(), 4, 0);
      icc.insert(icc.end(), dat.p + dat.pos, dat.p + dat.pos + 12);
      dat.pos += 12;
    } else if (command >= 16 && command < 24) {  // a type signature + reserved
      put_name(kIccTypeNames + 4 * (command - 16));
      icc.insert(icc.end(), 4, 0);
    } else {
      return kBad;
    }
  }
  return finished();
}

}  // namespace

int jxlhip_icc_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint8_t* icc, size_t icc_capacity,
                      size_t* icc_size) {
  if (!data || !bit_pos || (icc_capacity && !icc)) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (icc_size) *icc_size = 0;
  try {
    BitReader br(data, size, *bit_pos);
    const size_t first_bit = *bit_pos;
    FieldReader r(&br);
    const uint64_t enc_size = r.U64();
    if (!br.Healthy() || enc_size > (1u << 28)) return kBad;  // ICCReader::Init, icc_codec.cc:343-349
    EntropyCode code;
    int rc = DecodeEntropyCode(&br, 41, &code, /*disallow_lz77=*/false, 0);
    if (rc) return rc;
    SymbolReader reader(&code, &br);
    if (!reader.Ok()) return JXLHIP_ERR_OUT_OF_MEMORY;
    std::vector<uint8_t> enc;
    enc.reserve((size_t)std::min<uint64_t>(enc_size, 1u << 16));
    uint8_t b1 = 0, b2 = 0;
    for (size_t i = 0; i < enc_size; i++) {
      if ((i & 0xFFF) == 0 && i) {
        // a stream that ends early, or one that claims more than 256 profile bytes per coded byte, is damaged (:403-408)
        if (!br.Healthy() || reader.Corrupt()) return kBad;
        if ((i & 0xFFFF) == 0 && (double)i > (double)(br.BitsConsumed() - first_bit) / 8.0 * 256.0) return kBad;
      }
      const uint8_t b = (uint8_t)reader.ReadHybridUint(code.context_map[IccContext(i, b1, b2)], &br);
      enc.push_back(b);
      b2 = b1, b1 = b;
    }
    if (!br.Healthy() || reader.Corrupt() || !reader.FinalStateOk()) return kBad;
    std::vector<uint8_t> profile;
    if ((rc = UnpredictIcc(enc, &profile))) return rc;
    if (profile.empty()) return kBad;  // decode.cc:1124
    // the first frame starts on a byte (decode.cc:1133)
    const uint32_t rem = (uint32_t)(br.BitsConsumed() % 8);
    if (rem != 0 && br.Read(8 - rem) != 0) return kBad;
    if (!br.Healthy()) return kBad;
    *bit_pos = br.BitsConsumed();
    if (icc_size) *icc_size = profile.size();
    if (icc) {
      if (icc_capacity < profile.size()) return JXLHIP_ERR_INVALID_ARGUMENT;
      memcpy(icc, profile.data(), profile.size());
    }
    return kOk;
  } catch (const std::bad_alloc&) {
    return JXLHIP_ERR_OUT_OF_MEMORY;
  }
}

#include "modular.inc"

int jxlhip_dequant_encodings_decode(const uint8_t* data, size_t size, size_t* bit_pos, jxlhip_quant_encoding* enc) {
  if (!data || !bit_pos || !enc) return JXLHIP_ERR_INVALID_ARGUMENT;
  BitReader br(data, size, *bit_pos);
  const int rc = ReadQuantEncodings(&br, enc);
  if (rc) return rc;
  *bit_pos = br.BitsConsumed();
  return kOk;
}

int jxlhip_ac_global_decode(const uint8_t* data, size_t size, uint32_t num_groups, uint32_t num_passes,
                            uint32_t used_acs, const jxlhip_block_ctx_map* block_ctx_map, jxlhip_quant_encoding* enc,
                            uint32_t* num_histograms, jxlhip_ac_pass** passes, size_t* bits_consumed) {
  size_t pos = 0;
  const int rc = jxlhip_ac_global_decode_at(data, size, &pos, num_groups, num_passes, used_acs, block_ctx_map, enc,
                                            num_histograms, passes);
  if (rc == kOk && bits_consumed) *bits_consumed = pos;
  return rc;
}

int jxlhip_ac_global_decode_at(const uint8_t* data, size_t size, size_t* bit_pos, uint32_t num_groups,
                               uint32_t num_passes, uint32_t used_acs, const jxlhip_block_ctx_map* block_ctx_map,
                               jxlhip_quant_encoding* enc, uint32_t* num_histograms, jxlhip_ac_pass** passes) {
  if (!data || !bit_pos || !enc || !num_histograms || !passes || num_groups == 0 || num_passes == 0 || num_passes > 11)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < num_passes; i++) passes[i] = nullptr;
  size_t pos = *bit_pos;
  {
    BitReader br(data, size, pos);
    const int rc = ReadQuantEncodings(&br, enc);
    if (rc) return rc;
    uint32_t bits = 0;  // CeilLog2Nonzero(num_groups)
    while ((1ull << bits) < num_groups) bits++;
    *num_histograms = 1 + br.Read(bits);  // dec_frame.cc:383-386
    if (!br.Healthy()) return kBad;
    pos = br.BitsConsumed();
  }
  for (uint32_t i = 0; i < num_passes; i++) {
    const int rc = jxlhip_ac_pass_decode(data, size, &pos, used_acs, *num_histograms, block_ctx_map, &passes[i]);
    if (rc) {
      for (uint32_t j = 0; j < i; j++) {
        jxlhip_ac_pass_destroy(passes[j]);
        passes[j] = nullptr;
      }
      return rc;
    }
  }
  *bit_pos = pos;
  return kOk;
}

int jxlhip_block_ctx_map_decode(const uint8_t* data, size_t size, size_t* bit_pos, jxlhip_block_ctx_map* out) {
  if (!data || !bit_pos || !out) return JXLHIP_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof(*out));
  BitReader br(data, size, *bit_pos);
  out->num_dc_ctxs = 1;
  if (br.Read(1)) {  // default map
    out->ctx_map_size = 39;
    memcpy(out->ctx_map, kDefaultBlockCtx, 39);
    out->num_ctxs = 15;
  } else {
    // kDCThresholdDist / kQFThresholdDist (entropy_coder.h:37-42)
    static const U32Dist kDcDist = {{4, 8, 16, 32}, {0, 16, 272, 65808}};
    static const U32Dist kQfDist = {{2, 3, 5, 8}, {0, 4, 12, 44}};
    for (int c = 0; c < 3; c++) {
      out->num_dc_thresholds[c] = br.Read(4);
      out->num_dc_ctxs *= out->num_dc_thresholds[c] + 1;
      for (uint32_t i = 0; i < out->num_dc_thresholds[c]; i++) {
        const uint32_t u = ReadU32(&br, kDcDist);
        out->dc_thresholds[c][i] = (int32_t)((u >> 1) ^ (~(u & 1) + 1));  // UnpackSigned
      }
    }
    out->num_qf_thresholds = br.Read(4);
    for (uint32_t i = 0; i < out->num_qf_thresholds; i++) out->qf_thresholds[i] = ReadU32(&br, kQfDist) + 1;
    if (out->num_dc_ctxs * (out->num_qf_thresholds + 1) > 64) return kBad;
    out->ctx_map_size = 3u * kNumOrders * out->num_dc_ctxs * (out->num_qf_thresholds + 1);
    std::vector<uint8_t> map(out->ctx_map_size, 0);
    size_t num = 0;
    int rc = DecodeContextMap(&br, &map, &num, 0);
    if (rc) return rc;
    if (num > 16) return kBad;
    out->num_ctxs = (uint32_t)num;
    memcpy(out->ctx_map, map.data(), map.size());
  }
  if (!br.Healthy()) return kBad;
  *bit_pos = br.BitsConsumed();
  return kOk;
}

int jxlhip_dc_global_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint64_t frame_flags,
                            jxlhip_dc_global* out) {
  if (!data || !bit_pos || !out) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (frame_flags & (JXLHIP_FLAG_PATCHES | JXLHIP_FLAG_SPLINES | JXLHIP_FLAG_NOISE)) return JXLHIP_ERR_UNSUPPORTED;
  memset(out, 0, sizeof(*out));
  size_t pos;
  {
    BitReader br(data, size, *bit_pos);
    // DequantMatrices::DecodeDC
    out->dc_quant[0] = 1.0f / 4096.0f;
    out->dc_quant[1] = 1.0f / 512.0f;
    out->dc_quant[2] = 1.0f / 256.0f;
    if (!br.Read(1)) {
      for (int c = 0; c < 3; c++) {
        float v;
        if (!ReadF16(&br, &v)) return kBad;
        v *= 1.0f / 128.0f;
        if (v < 1e-8f) return kBad;
        out->dc_quant[c] = v;
      }
    }
    // QuantizerParams
    static const U32Dist kGlobalScale = {{11, 11, 12, 16}, {1, 2049, 4097, 8193}};
    static const U32Dist kQuantDc = {{0, 5, 8, 16}, {16, 1, 1, 1}};
    out->global_scale = (int32_t)ReadU32(&br, kGlobalScale);
    out->quant_dc = (int32_t)ReadU32(&br, kQuantDc);
    if (!br.Healthy()) return kBad;
    pos = br.BitsConsumed();
  }
  int rc = jxlhip_block_ctx_map_decode(data, size, &pos, &out->block_ctx_map);
  if (rc) return rc;
  BitReader br(data, size, pos);
  out->cfl_color_factor = 84;
  out->cfl_base_x = 0.0f;
  out->cfl_base_b = 1.0f;  // jxl::cms::kYToBRatio
  if (!br.Read(1)) {
    static const U32Dist kColorFactor = {{0, 0, 8, 16}, {84, 256, 2, 258}};
    out->cfl_color_factor = ReadU32(&br, kColorFactor);
    if (!ReadF16(&br, &out->cfl_base_x) || fabsf(out->cfl_base_x) > 4.0f) return kBad;
    if (!ReadF16(&br, &out->cfl_base_b) || fabsf(out->cfl_base_b) > 4.0f) return kBad;
    out->ytox_dc = (int32_t)br.Read(8) - 128;
    out->ytob_dc = (int32_t)br.Read(8) - 128;
  }
  if (!br.Healthy()) return kBad;
  *bit_pos = br.BitsConsumed();
  return kOk;
}

int jxlhip_quant_dc_contexts(const jxlhip_block_ctx_map* map, size_t n, const int32_t* const q[3], uint8_t* out) {
  if (!out || (map && map->num_dc_ctxs > 1 && (!q || !q[0] || !q[1] || !q[2]))) return JXLHIP_ERR_INVALID_ARGUMENT;
  if (!map || map->num_dc_ctxs <= 1) {
    memset(out, 0, n);
    return kOk;
  }
  for (size_t i = 0; i < n; i++) {
    uint32_t b[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++)
      for (uint32_t t = 0; t < map->num_dc_thresholds[c]; t++) b[c] += q[c][i] > map->dc_thresholds[c][t];
    uint32_t bucket = b[0];
    bucket = bucket * (map->num_dc_thresholds[2] + 1) + b[2];
    bucket = bucket * (map->num_dc_thresholds[1] + 1) + b[1];
    out[i] = (uint8_t)bucket;
  }
  return kOk;
}

uint32_t jxlhip_ac_pass_max_num_bits(const jxlhip_ac_pass* pass) { return pass ? pass->code.max_num_bits : 0; }
uint32_t jxlhip_ac_pass_used_orders(const jxlhip_ac_pass* pass) { return pass ? pass->used_orders : 0; }
const uint32_t* jxlhip_ac_pass_order(const jxlhip_ac_pass* pass, uint32_t ord, uint32_t c) {
  if (!pass || ord >= (uint32_t)kNumOrders || c > 2) return nullptr;
  return pass->orders.data() + Layout().offset[ord][c];
}

int jxlhip_ac_group_decode(const jxlhip_ac_pass* pass, uint32_t xsb, uint32_t ysb, uint32_t gx, uint32_t gy,
                           const uint8_t* acs, const int32_t* raw_quant, const uint8_t* quant_dc,
                           const uint8_t* data, size_t size, size_t* bit_pos, uint32_t shift, uint32_t coeff_type,
                           void* const coeffs[3], size_t* ncoeffs) {
  if (!pass || !acs || !raw_quant || !data || !bit_pos || !coeffs || !coeffs[0] || !coeffs[1] || !coeffs[2] ||
      coeff_type > JXLHIP_COEFF_I32 || shift > 24)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  BitReader br(data, size, *bit_pos);
  int rc;
  if (coeff_type == JXLHIP_COEFF_I16) {
    int16_t* const c16[3] = {(int16_t*)coeffs[0], (int16_t*)coeffs[1], (int16_t*)coeffs[2]};
    rc = DecodeGroupT<int16_t>(pass, xsb, ysb, gx, gy, acs, raw_quant, quant_dc, &br, shift, c16, ncoeffs);
  } else {
    int32_t* const c32[3] = {(int32_t*)coeffs[0], (int32_t*)coeffs[1], (int32_t*)coeffs[2]};
    rc = DecodeGroupT<int32_t>(pass, xsb, ysb, gx, gy, acs, raw_quant, quant_dc, &br, shift, c32, ncoeffs);
  }
  if (rc == kOk) *bit_pos = br.BitsConsumed();
  return rc;
}

int jxlhip_ac_group_decode_sparse(const jxlhip_ac_pass* pass, uint32_t xsb, uint32_t ysb, uint32_t gx, uint32_t gy,
                                  const uint8_t* acs, const int32_t* raw_quant, const uint8_t* quant_dc,
                                  const uint8_t* data, size_t size, size_t* bit_pos, uint32_t shift,
                                  uint32_t* const entries[3], const uint32_t capacity[3], uint32_t counts[3], size_t* ncoeffs) {
  if (!pass || !acs || !raw_quant || !data || !bit_pos || !entries || !entries[0] || !entries[1] || !entries[2] ||
      !counts || !capacity || shift > 24)
    return JXLHIP_ERR_INVALID_ARGUMENT;
  BitReader br(data, size, *bit_pos);
  SparseSink sink;
  for (int c = 0; c < 3; c++) sink.ent[c] = entries[c], sink.cap[c] = capacity[c];
  const int rc = DecodeGroupS(pass, xsb, ysb, gx, gy, acs, raw_quant, quant_dc, &br, shift, sink, ncoeffs);
  for (int c = 0; c < 3; c++) counts[c] = sink.cnt[c];
  if (rc == kOk) *bit_pos = br.BitsConsumed();
  return rc;
}

}  // extern "C"
