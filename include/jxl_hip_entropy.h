/* jxl_hip_entropy.h -- host side of the VarDCT back-end's input: the AC entropy
 * decoder that turns the bytes of a frame's AC-global and AC-group sections
 * into the quantized coefficient stream jxlhip_submit_group() uploads
 * (SURVEY.md section 8, row f1).  Plain C ABI, part of libjxl_hip.so; host
 * code only (ANS / prefix decoding is serial per group and parallel across
 * groups: it runs on the JxlParallelRunner's threads while the GPU decodes the
 * previous groups).
 *
 * Replaces, behaviour for behaviour (libjxl tree, lib/jxl/):
 *   DecodeHistograms, ANSSymbolReader      dec_ans.cc:31-421, dec_ans.h:148-470
 *   DecodeContextMap                        dec_context_map.cc:47-95
 *   HuffmanDecodingData                     dec_huffman.cc:20-245, huffman_table.cc:55-159
 *   InitAliasTable / AliasTable::Lookup     ans_common.cc:41-146, ans_common.h:107-149
 *   DecodeCoeffOrders, natural orders       coeff_order.cc:36-156, ac_strategy.cc:28-79,
 *                                           lehmer_code.h:60-100
 *   the per-pass part of ProcessACGlobal    dec_frame.cc:396-416
 *   GetBlockFromBitstream + DecodeACVarBlock dec_group.cc:466-640
 * Output layout = the reference's ACImage (dct_util.h:23-96): per group and
 * channel 65536 slots, varblocks in raster visit order, 64*covered_blocks
 * coefficients each in dequant-matrix order, LLF slots untouched.
 */
#ifndef JXL_HIP_ENTROPY_H_
#define JXL_HIP_ENTROPY_H_

#include <stddef.h>
#include <stdint.h>

#include "jxl_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors jxl::BlockCtxMap (lib/jxl/ac_context.h:85-150) as decoded from the
 * DC-global section (DecodeBlockCtxMap, entropy_coder.cc:25-61).  A NULL pointer
 * wherever this struct is expected means the default map (kDefaultCtxMap, no
 * thresholds).  Channels of the DC thresholds in X, Y, B order. */
#define JXLHIP_BLOCK_CTX_MAP_MAX (3 * 13 * 64)
typedef struct jxlhip_block_ctx_map {
  uint32_t num_dc_thresholds[3]; /* each <= 15 */
  int32_t dc_thresholds[3][15];
  uint32_t num_dc_ctxs;          /* product of (num_dc_thresholds[c] + 1) */
  uint32_t num_qf_thresholds;    /* <= 15 */
  uint32_t qf_thresholds[15];
  uint32_t num_ctxs;             /* distinct block contexts, <= 16 */
  uint32_t ctx_map_size;         /* 3 * 13 * (num_qf_thresholds + 1) * num_dc_ctxs */
  uint8_t ctx_map[JXLHIP_BLOCK_CTX_MAP_MAX];
} jxlhip_block_ctx_map;

/* DecodeBlockCtxMap: reads the map at bit *bit_pos of data (advanced). */
JXLHIP_EXPORT int jxlhip_block_ctx_map_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                              jxlhip_block_ctx_map* out);
/* The per-block DC context index the AC contexts depend on (what DequantDC
 * leaves in PassesSharedState::quant_dc, compressed_dc.cc:251-295): n blocks,
 * quantized DC planes in X, Y, B order; map may be NULL (all zero). */
JXLHIP_EXPORT int jxlhip_quant_dc_contexts(const jxlhip_block_ctx_map* map, size_t n,
                                           const int32_t* const quant_dc[3], uint8_t* out);

/* DequantMatrices::Decode (quant_weights.cc:373-511), the first field of the
 * AC-global section: one "all default" bit, else 17 x (3-bit mode + that mode's
 * F16 parameters).  Fills enc[JXLHIP_NUM_QUANT_TABLES] for
 * jxlhip_dequant_tables(); *bit_pos advanced.  JXLHIP_ERR_BAD_STREAM on the
 * reference's failures (too-small weights, F16 inf/NaN, non-8x8 mode on a larger
 * table, truncation), JXLHIP_ERR_UNSUPPORTED on kQuantModeRAW (its table is
 * modular-coded). */
JXLHIP_EXPORT int jxlhip_dequant_encodings_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                                  jxlhip_quant_encoding* enc);

/* One pass of the AC-global section: coefficient orders + entropy code. */
typedef struct jxlhip_ac_pass jxlhip_ac_pass;

/* Decodes `U32(kOrderEnc) used_orders; DecodeCoeffOrders; DecodeHistograms`
 * (dec_frame.cc:399-410) starting at bit *bit_pos of data[0..size); *bit_pos is
 * advanced past what was read.  used_acs: bit mask of the raw AC strategies
 * present in the frame (PassesDecoderState::used_acs); num_histograms: the
 * frame's number of histogram sets (dec_frame.cc:383-386).
 * Returns JXLHIP_OK, JXLHIP_ERR_BAD_STREAM (invalid or truncated data) or
 * JXLHIP_ERR_INVALID_ARGUMENT / JXLHIP_ERR_OUT_OF_MEMORY. */
JXLHIP_EXPORT int jxlhip_ac_pass_decode(const uint8_t* data, size_t size, size_t* bit_pos,
                                        uint32_t used_acs, uint32_t num_histograms,
                                        const jxlhip_block_ctx_map* block_ctx_map,
                                        jxlhip_ac_pass** out);
JXLHIP_EXPORT void jxlhip_ac_pass_destroy(jxlhip_ac_pass* pass);
/* ANSCode::max_num_bits: the frame can use JXLHIP_COEFF_I16 when, over all its
 * passes, max + ceil(log2(num_passes)) < 16 (dec_frame.cc:414-421). */
JXLHIP_EXPORT uint32_t jxlhip_ac_pass_max_num_bits(const jxlhip_ac_pass* pass);
/* the 13-bit mask of transmitted coefficient orders (debugging / tests) */
JXLHIP_EXPORT uint32_t jxlhip_ac_pass_used_orders(const jxlhip_ac_pass* pass);
/* coefficient order of order bucket `ord` (0..12) and channel c: 64 * covered
 * blocks entries (tests) */
JXLHIP_EXPORT const uint32_t* jxlhip_ac_pass_order(const jxlhip_ac_pass* pass, uint32_t ord, uint32_t c);

/* The whole AC-global section as FrameDecoder::ProcessACGlobal reads it
 * (dec_frame.cc:372-416): jxlhip_dequant_encodings_decode, then num_histograms
 * (1 + CeilLog2Nonzero(num_groups) bits), then num_passes x jxlhip_ac_pass_decode.
 * passes[0..num_passes) receive the pass handles (caller destroys them; all are
 * NULL on failure).  data/size: the section's bytes, read from bit 0. */
JXLHIP_EXPORT int jxlhip_ac_global_decode(const uint8_t* data, size_t size, uint32_t num_groups,
                                          uint32_t num_passes, uint32_t used_acs,
                                          const jxlhip_block_ctx_map* block_ctx_map,
                                          jxlhip_quant_encoding* enc, uint32_t* num_histograms,
                                          jxlhip_ac_pass** passes, size_t* bits_consumed);

/* The same from bit *bit_pos of data (advanced): frames that consist of ONE section carry DC global,
 * the DC group, AC global and the AC group back to back without byte alignment
 * (FrameDecoder::ProcessSections, dec_frame.cc:596-616). */
JXLHIP_EXPORT int jxlhip_ac_global_decode_at(const uint8_t* data, size_t size, size_t* bit_pos,
                                             uint32_t num_groups, uint32_t num_passes, uint32_t used_acs,
                                             const jxlhip_block_ctx_map* block_ctx_map,
                                             jxlhip_quant_encoding* enc, uint32_t* num_histograms,
                                             jxlhip_ac_pass** passes);

/* Decodes one pass of one AC group (DecodeGroup with GetBlockFromBitstream,
 * dec_group.cc:560-640,780-815): the histogram-set selector, then per varblock
 * in raster visit order and per channel (Y, X, B) the number of non-zeros and
 * the coefficients, ADDED (<< shift) into coeffs[c][...] -- the caller zeroes
 * the three 65536-slot buffers before the group's first pass.
 *   xsize_blocks/ysize_blocks: frame size in 8x8 blocks; group_x/y: AC group
 *   ac_strategy, raw_quant: whole-frame side info as in jxlhip_frame_inputs
 *   quant_dc: per-block DC context index (PassesSharedState::quant_dc), NULL = 0
 *   coeff_type: JXLHIP_COEFF_I16 / I32 element type of coeffs[].  The reference picks 16-bit
 *               buffers only when no token of the pass's code CAN carry 16 bits
 *               (jxlhip_ac_pass_max_num_bits; one flat histogram in the stream is enough to
 *               exceed that, and libjxl's encoder writes those routinely) although the values
 *               rarely need them.  I16 may therefore be tried whatever max_num_bits says:
 *               JXLHIP_ERR_RANGE reports a coefficient that left the 16-bit range (the frame
 *               is then redone with I32), otherwise the result is exactly the reference's.
 *   ncoeffs (optional): slots used by the group (what jxlhip_submit_group takes)
 * 4:4:4 only (the VarDCT back-end does not implement chroma subsampling). */
JXLHIP_EXPORT int jxlhip_ac_group_decode(const jxlhip_ac_pass* pass, uint32_t xsize_blocks,
                                         uint32_t ysize_blocks, uint32_t group_x, uint32_t group_y,
                                         const uint8_t* ac_strategy, const int32_t* raw_quant,
                                         const uint8_t* quant_dc, const uint8_t* data, size_t size,
                                         size_t* bit_pos, uint32_t shift, uint32_t coeff_type,
                                         void* const coeffs[3], size_t* ncoeffs);

/* The same group as NON-ZERO coefficients only: entries[c][i] = (position << 16) | (uint16_t)value, position =
 * the coefficient's index in the group's channel stream (what jxlhip_ac_group_decode would have written to
 * coeffs[c][position]), in decode order; counts[c] entries per channel, at most capacity[c].  Single pass
 * only (values are not accumulated).  JXLHIP_ERR_RANGE: a value does not fit 16 bits or a channel has more than
 * capacity[c] non-zeros -- decode the group densely instead.  Nine out of ten coefficients of a d1.0 frame are zero:
 * this is the form in which jxlhip_ac_group_decode_submit sends a group over PCIe (JXLHIP_SPARSE_UPLOAD=0 turns it
 * off); the context expands it into the dense block stream on the device. */
JXLHIP_EXPORT int jxlhip_ac_group_decode_sparse(const jxlhip_ac_pass* pass, uint32_t xsize_blocks,
                                                uint32_t ysize_blocks, uint32_t group_x, uint32_t group_y,
                                                const uint8_t* ac_strategy, const int32_t* raw_quant,
                                                const uint8_t* quant_dc, const uint8_t* data, size_t size,
                                                size_t* bit_pos, uint32_t shift, uint32_t* const entries[3],
                                                const uint32_t capacity[3], uint32_t counts[3], size_t* ncoeffs);

/* Entropy-decode a single-pass group straight into a pinned staging buffer of
 * the context and queue its upload (jxlhip_submit_group): the call a
 * JxlParallelRunner worker makes per AC group.  Thread-safe.  Side info
 * pointers are HOST memory (the same arrays given to jxlhip_upload_side_info). */
JXLHIP_EXPORT int jxlhip_ac_group_decode_submit(jxlhip_ctx* ctx, const jxlhip_ac_pass* pass,
                                                uint32_t group_idx, const uint8_t* ac_strategy,
                                                const int32_t* raw_quant, const uint8_t* quant_dc,
                                                const uint8_t* data, size_t size, size_t* bit_pos);

/* The multi-pass form (progressive AC: FrameDecoder::ProcessACGroup hands DecodeGroup one
 * reader per pass, dec_frame.cc:455-516, dec_group.cc:575-590): pass p of the group is read
 * from data[p][0..sizes[p]) starting at bit bit_pos[p] (advanced) with passes[p] and added
 * << shifts[p] (frame_header.passes.shift; NULL = all 0).  The caller passes the sections of
 * every pass it has; a later call for the same group with more passes re-decodes from pass 0
 * (the slot is zeroed), which is what a progressive re-render does. */
JXLHIP_EXPORT int jxlhip_ac_group_decode_submit_passes(jxlhip_ctx* ctx, uint32_t num_passes,
                                                       const jxlhip_ac_pass* const* passes,
                                                       const uint32_t* shifts, uint32_t group_idx,
                                                       const uint8_t* ac_strategy, const int32_t* raw_quant,
                                                       const uint8_t* quant_dc, const uint8_t* const* data,
                                                       const size_t* sizes, size_t* bit_pos);

/* ---- frame table of contents (row f4: the part of the front-end that locates the sections) ---- */
/* NumTocEntries (toc.h:33-43): 1 when the frame is a single section, else DC global + DC groups
 * + AC global + one per pass and AC group. */
JXLHIP_EXPORT uint32_t jxlhip_num_toc_entries(uint32_t num_groups, uint32_t num_dc_groups, uint32_t num_passes);
/* ReadGroupOffsets (toc.cc:28-115): the TOC that follows the frame header at bit *bit_pos of
 * data -- optional permutation (Lehmer code, entropy coded), byte alignment, num_entries
 * U32(kTocDist) sizes, byte alignment.  offsets[i] / sizes[i]: byte offset (relative to the first
 * byte after the TOC) and size of LOGICAL section i (the permutation applied).  *bit_pos is
 * advanced to that first byte; *total_size (optional) = sum of the sizes. */
JXLHIP_EXPORT int jxlhip_toc_decode(const uint8_t* data, size_t size, size_t* bit_pos, uint32_t num_entries,
                                    uint64_t* offsets, uint32_t* sizes, uint64_t* total_size);

/* All AC groups of a frame on a JxlParallelRunner (include/jxl/parallel_runner.h:127; e.g.
 * JxlThreadParallelRunner of libjxl_threads_hip.so): what FrameDecoder::ProcessSections does
 * with RunOnPool over ProcessACGroup (dec_frame.cc:700-760) once every AC section is present.
 * runner == NULL runs the groups on the calling thread.  sections[p * num_groups + g] /
 * sizes[...]: bytes of pass p of group g (each read from bit 0).  Groups outside the
 * context's stripe are skipped.  Returns the first error of any group (JXLHIP_ERR_RANGE:
 * redo the frame with JXLHIP_COEFF_I32), JXLHIP_ERR_STATE when the runner fails. */
typedef int (*jxlhip_parallel_runner)(void* runner_opaque, void* jpegxl_opaque,
                                      int (*init)(void* jpegxl_opaque, size_t num_threads),
                                      void (*func)(void* jpegxl_opaque, uint32_t value, size_t thread_id),
                                      uint32_t start_range, uint32_t end_range);
JXLHIP_EXPORT int jxlhip_ac_groups_decode_submit(jxlhip_ctx* ctx, jxlhip_parallel_runner runner,
                                                 void* runner_opaque, uint32_t num_passes,
                                                 const jxlhip_ac_pass* const* passes, const uint32_t* shifts,
                                                 const uint8_t* ac_strategy, const int32_t* raw_quant,
                                                 const uint8_t* quant_dc, const uint8_t* const* sections,
                                                 const size_t* sizes);
/* The same; end_bits (optional, [num_passes * num_groups], indexed like sections) receives the bit position in each
 * section right behind the VarDCT coefficients -- where the group's part of the frame's Modular image starts
 * (jxlhip_modular_ac_group_decode, dec_frame.cc:497-530).  Entries of groups outside the stripe are left alone. */
JXLHIP_EXPORT int jxlhip_ac_groups_decode_submit_ex(jxlhip_ctx* ctx, jxlhip_parallel_runner runner,
                                                    void* runner_opaque, uint32_t num_passes,
                                                    const jxlhip_ac_pass* const* passes, const uint32_t* shifts,
                                                    const uint8_t* ac_strategy, const int32_t* raw_quant,
                                                    const uint8_t* quant_dc, const uint8_t* const* sections,
                                                    const size_t* sizes, size_t* end_bits);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_ENTROPY_H_ */
