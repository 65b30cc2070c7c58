/* TEST INFRASTRUCTURE (oracle/_ref build only): declarations of the few Brotli
 * encoder entry points lib/jxl/encode.cc names for its "brob" box compression.
 * The oracle never compresses boxes; oracle/ref_real_stream.cc defines these
 * functions as failing stubs so that encode.cc (needed for the output-processor
 * plumbing of jxl::EncodeFrame) links without libbrotlienc. */
#ifndef ORACLE_SHIM_BROTLI_ENCODE_H_
#define ORACLE_SHIM_BROTLI_ENCODE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct BrotliEncoderStateStruct BrotliEncoderState;
typedef enum { BROTLI_OPERATION_PROCESS = 0, BROTLI_OPERATION_FLUSH = 1, BROTLI_OPERATION_FINISH = 2 } BrotliEncoderOperation;
typedef enum { BROTLI_PARAM_MODE = 0, BROTLI_PARAM_QUALITY = 1, BROTLI_PARAM_LGWIN = 2, BROTLI_PARAM_SIZE_HINT = 5 } BrotliEncoderParameter;
typedef void* (*brotli_alloc_func)(void* opaque, size_t size);
typedef void (*brotli_free_func)(void* opaque, void* address);
#define BROTLI_BOOL int
#define BROTLI_TRUE 1
#define BROTLI_FALSE 0
BrotliEncoderState* BrotliEncoderCreateInstance(brotli_alloc_func, brotli_free_func, void*);
void BrotliEncoderDestroyInstance(BrotliEncoderState*);
BROTLI_BOOL BrotliEncoderSetParameter(BrotliEncoderState*, BrotliEncoderParameter, uint32_t);
BROTLI_BOOL BrotliEncoderCompressStream(BrotliEncoderState*, BrotliEncoderOperation, size_t* available_in,
                                        const uint8_t** next_in, size_t* available_out, uint8_t** next_out,
                                        size_t* total_out);
BROTLI_BOOL BrotliEncoderIsFinished(BrotliEncoderState*);
size_t BrotliEncoderMaxCompressedSize(size_t input_size);
#ifdef __cplusplus
}
#endif
#endif
