/*
 * jxl_oracle.h -- CPU restatement of libjxl's VarDCT decode back-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or executed from the product library (libjxl_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / reported baseline.
 *
 * Parity status: PINNED AGAINST THE REFERENCE ITSELF.  The reference's hot path
 * is compiled in place from /root/reference/lib/jxl (73 decoder translation
 * units, unmodified) by oracle/build_ref.py into oracle/_ref/libjxl_ref.so; the
 * only stand-in is oracle/hwy_shim, a from-scratch single-lane implementation of
 * the Highway API subset libjxl uses (third_party/highway is an un-vendored
 * submodule, deps.sh:20).  oracle/ref_driver.cc feeds the same in-memory inputs
 * to the reference's DecodeGroupForRoundtrip + ComputeSigma + real render
 * pipeline (both executors).  tests/test_reference_parity.py holds this
 * restatement to BIT-EXACT equality with that library (all 27 strategies, all 8
 * stage lists, ragged sizes, int32 coefficients, custom LoopFilter fields, DC
 * dequant + adaptive smoothing, dequant tables), and tests/golden/ (npz files) are the
 * reference's own outputs.  In addition the reference's fixture-free
 * known-answer tests are restated in tests/test_oracle_kat.py:
 *   lib/jxl/dct_test.cc:165-214,251-300,314-475   (DCT/IDCT vs f64 matrix)
 *   lib/jxl/ac_strategy_test.cc:28-245            (27 strategies: roundtrip,
 *        DC = mean, LLF<->DC, 8x8-mean(IDCT(LLF)) = DC, AFV orthonormal)
 *   lib/jxl/quant_weights_test.cc:185-271         (dequant tables)
 *   lib/jxl/opsin_inverse_test.cc:27-49           (XYB inverse of forward)
 * Caveat stated once: the shim is scalar with MulAdd = fmaf and exact
 * reciprocals, i.e. the reference as built for an FMA target with
 * JXL_HIGH_PRECISION=1; an AVX2 build differs from it in the last bits only where
 * libjxl uses ApproximateReciprocal (within the reference's own 2e-4 bar).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * libjxl tree).  Arithmetic is fp32 with the reference's operation order;
 * where the reference uses a fused multiply-add (hwy MulAdd/NegMulAdd) this
 * code calls fmaf() explicitly and is compiled with -ffp-contract=off.
 */
#ifndef JXL_ORACLE_H_
#define JXL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/jxl_hip.h" /* shared POD parameter structs only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- strategies (ac_strategy.h:35-173, quant_weights.h:302-348) --------- */
int jxo_covered_blocks_x(int s);
int jxo_covered_blocks_y(int s);
int jxo_log2_covered_blocks(int s);
int jxo_quant_table_of_strategy(int s);
size_t jxo_dequant_table_offset(int s, int c);

/* ---- 1-D / 2-D DCT (dct-inl.h) ------------------------------------------ */
/* fast fp32 N-point IDCT / DCT on one strided column (dct-inl.h:158-232) */
void jxo_idct1d(int n, const float* from, size_t from_stride, float* to,
                size_t to_stride);
void jxo_dct1d(int n, float* mem, size_t stride); /* in place, unscaled */
/* ComputeScaledIDCT<ROWS,COLS> / ComputeScaledDCT<ROWS,COLS>
 * (dct-inl.h:351-397).  coeffs: min(R,C) x max(R,C) matrix, transposed
 * storage when R >= C.  coeffs is clobbered by the inverse (as in the
 * reference). */
void jxo_scaled_idct(int rows, int cols, float* coeffs, float* pixels,
                     size_t pixels_stride);
void jxo_scaled_dct(int rows, int cols, const float* pixels,
                    size_t pixels_stride, float* coeffs);
/* double-precision matrix forms, dct_for_test.h:20-95 */
void jxo_idct1d_slow(int n, const double* in, double* out);
void jxo_dct1d_slow(int n, const double* in, double* out);

/* ---- per-varblock transforms (dec_transforms-inl.h, enc_transforms-inl.h) */
void jxo_transform_to_pixels(int strategy, float* coeffs, float* pixels,
                             size_t pixels_stride);
void jxo_transform_from_pixels(int strategy, const float* pixels,
                               size_t pixels_stride, float* coeffs);
void jxo_llf_from_dc(int strategy, const float* dc, size_t dc_stride,
                     float* llf);
void jxo_dc_from_llf(int strategy, const float* block, float* dc,
                     size_t dc_stride);
const float* jxo_afv_basis(void); /* 16x16, row j = basis function j */

/* ---- dequant tables (quant_weights.cc) ----------------------------------- */
float jxo_fast_powf(float base, float exponent);
/* Default library, all 17 kinds. table: JXLHIP_DEQUANT_TABLE_FLOATS floats.
 * inv_table may be NULL.  Returns 0 on success. */
int jxo_default_dequant_tables(float* table, float* inv_table);
/* ... of 17 jxlhip_quant_encoding (NULL / JXLHIP_QUANT_LIBRARY = default); -1 on the
 * reference's "Invalid distance bands" / "Invalid quantization table" failures */
int jxo_dequant_tables(const jxlhip_quant_encoding* encodings, float* table, float* inv_table);

/* ---- dequant (quantizer-inl.h:34-67, dec_group.cc:115-181) -------------- */
float jxo_adjust_quant_bias(int c, int32_t q, const float biases[4]);

/* ---- frame-level ---------------------------------------------------------- */
typedef struct jxo_frame {
  jxlhip_frame_params p;
  /* host pointers, layouts as jxlhip_frame_inputs */
  const void* coeffs[3];
  const uint8_t* ac_strategy;
  const int32_t* raw_quant;
  const uint8_t* epf_sharpness;
  const int8_t* ytox_map;
  const int8_t* ytob_map;
  const float* dc[3];
  const float* dequant_table;
} jxo_frame;

/* Phase 1: DecodeGroupImpl for every group (dec_group.cc:183-457).
 * xyb[c]: planes of ysize_padded rows x row_stride floats (row_stride >=
 * xsize_padded).  group_begin/group_end select a range of groups (raster
 * group index) so callers can thread over groups. Returns 0 / -1 on a
 * malformed strategy map. */
int jxo_decode_groups(const jxo_frame* f, float* const xyb[3],
                      size_t row_stride, uint32_t group_begin,
                      uint32_t group_end);
/* ComputeSigma (epf.cc:39-133) without the padding border: inv_sigma per 8x8
 * cell, dense xsize_blocks*ysize_blocks. */
void jxo_compute_sigma(const jxo_frame* f, float* inv_sigma);

/* Render-pipeline stages with SimpleRenderPipeline semantics
 * (simple_render_pipeline.cc:129-211): whole frame, true size, mirror border.
 * in/out: 3 planes xsize*ysize, row stride `stride`.  row_begin/row_end let the
 * caller thread over rows. */
void jxo_gaborish(const jxo_frame* f, const float* const in[3],
                  float* const out[3], size_t stride, uint32_t row_begin,
                  uint32_t row_end);
/* which: 0,1,2 = EPF0,EPF1,EPF2 (stage_epf.cc) */
void jxo_epf(const jxo_frame* f, int which, const float* inv_sigma,
             const float* const in[3], float* const out[3], size_t stride,
             uint32_t row_begin, uint32_t row_end);
/* XybToRgb (dec_xyb-inl.h:38-86); out interleaved RGB, row stride in floats */
void jxo_xyb_to_linear_rgb(const jxo_frame* f, const float* const in[3],
                           size_t stride, float* rgb, size_t rgb_stride,
                           uint32_t row_begin, uint32_t row_end);

/* Whole path, single thread or `threads` pthreads over groups / rows.
 * out per p.output_kind.  Returns 0 on success. */
/* FromLinearStage + WriteToOutputStage (output.c) for JXLHIP_OUT_PACKED */
float jxo_srgb_from_linear(float v);
float jxo_pq_from_linear(float v, float intensity_target);
float jxo_709_from_linear(float x);
float jxo_gamma_from_linear(float x, float inverse_gamma);
float jxo_hlg_from_linear(float x);
void jxo_pack_output(const jxo_frame* f, const float* rgb, size_t rgb_stride, void* out,
                     size_t out_stride_bytes, uint32_t row_begin, uint32_t row_end);
int jxo_decode_frame(const jxo_frame* f, float* out, size_t out_stride_floats,
                     size_t out_plane_stride, int threads);

/* DequantDC (444 case) + AdaptiveDCSmoothing (compressed_dc.cc:128-250) */
void jxo_dequant_dc(uint32_t xsize_blocks, uint32_t ysize_blocks,
                    const int32_t* const quant_dc[3], float* const dc[3],
                    const float mul_dc[3], float cfl_x_dc, float cfl_b_dc);
void jxo_adaptive_dc_smoothing(uint32_t xsize_blocks, uint32_t ysize_blocks,
                               const float mul_dc[3], float* const dc[3]);

/* Forward XYB for generators/tests: LinearRGBToXYB (enc_xyb.cc:83-105) */
void jxo_linear_rgb_to_xyb(float r, float g, float b, float xyb[3]);

#ifdef __cplusplus
}
#endif
#endif
