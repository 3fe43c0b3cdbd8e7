/* dclib_boundary.h -- everything the DEGA/LZMH plugin rows and a DCCLI-style driver need from the host side, in one place.
 *
 * The NAMES and SIGNATURES here are the reference's public ones (they are the drop-in boundary: a codec source written
 * against the reference compiles against this file and vice versa -- src/dega_plugin.c is compiled both ways on every
 * build); the declarations are regrouped by what a plugin author needs, and the implementation behind them
 * (src/bit_io.c, src/enc_dec.c) is this project's own.  The reference spreads them over
 *   common/inc/io.h:12-14,62-63           integer types of the I/O layer
 *   common/inc/err_codes.h:8-39           status codes and their texts
 *   DCIOLib/inc/file_buffer.h:10-41       byte stream behind a bit buffer
 *   DCIOLib/inc/bit_file_buffer.h:12-33   bit-granular stream I/O
 *   DCLib/inc/enc_dec.h:10-71             codec table, option system, the enc_dec_function_t type
 * and the five headers of those names next to this file only include it.
 */
#ifndef DC_AMD_DCLIB_BOUNDARY_H
#define DC_AMD_DCLIB_BOUNDARY_H
#define DC_AMD_ENC_DEC_H /* tells src/dega_plugin.c that options_t has num_channels */

#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* ---- 1. integers: the reference's IO_SIZE_BITS = 64 configuration ------------------------------------------------- */
#define IO_SIZE_BITS 64
typedef int64_t io_int_t;   /* what every enc_dec_function_t returns; also bit counts of the stream calls */
typedef uint64_t io_uint_t; /* single values of 1..64 bits */

/* ---- 2. status codes (part of the boundary: a codec returns them, DCCLI prints ERROR_MESSAGE_STRING(code)) ----------- */
#define NO_ERROR 0
#define ERROR_INVALID_VALUE (-1)                 /* range violations: diff.c:17-18, normalize.c:21-22 */
#define ERROR_VALUE_LARGER_THAN_USABLE_SIZE (-2)
#define ERROR_INVALID_FORMAT (-3)                /* undecodable stream */
#define ERROR_INVALID_MODE (-4)
#define ERROR_FILE_IO (-5)
#define ERROR_MEMORY (-6)
#define ERROR_LIBRARY_INIT (-10)                 /* here: no usable GPU */
#define ERROR_LIBRARY_CALL (-11)                 /* short reads; here also: a HIP call failed */

static inline const char *dc_error_message(long code)
{
  switch (code)
  {
    case NO_ERROR: return "Successful";
    case ERROR_INVALID_VALUE: return "Invalid value";
    case ERROR_VALUE_LARGER_THAN_USABLE_SIZE: return "Value larger than usable size";
    case ERROR_INVALID_FORMAT: return "Invalid format";
    case ERROR_INVALID_MODE: return "Invalid mode";
    case ERROR_FILE_IO: return "File I/O error";
    case ERROR_MEMORY: return "Memory error";
    case ERROR_LIBRARY_INIT: return "Error initializing library";
    case ERROR_LIBRARY_CALL: return "Error calling library";
    default: return "Unknown error";
  }
}
#define ERROR_MESSAGE_STRING(code) dc_error_message((long)(code))

/* ---- 3. streams ------------------------------------------------------------------------------------------------------
 * A bit buffer sits on a byte buffer, which is a FILE or growable memory.  The format is what the GPU kernels reproduce:
 * bits fill a byte from its most significant bit (bit_file_buffer.c:220-248); an n-bit value goes most significant bit
 * first, so 32-bit values are big-endian (:297-308); a stream may end on a fractional byte -- a write->read mode switch
 * keeps the exact bit count (:127-144), a file gets the last byte zero padded, and an empty output still writes one
 * 0x00 byte (:310-333). */
typedef enum file_buffer_mode_t { FBM_INVALID = -1, FBM_READING = 0, FBM_WRITING = 1 } file_buffer_mode_t;
typedef struct file_buffer_t file_buffer_t;
typedef struct bit_file_buffer_t bit_file_buffer_t;

file_buffer_t *AllocateFileBuffer(void);
void FreeFileBuffer(file_buffer_t *fb);
int InitFileBuffer(file_buffer_t *fb, FILE *file, file_buffer_mode_t mode, size_t buffer_size);
int InitFileBufferInMemory(file_buffer_t *fb, file_buffer_mode_t mode, size_t buffer_size);
void UninitFileBuffer(file_buffer_t *fb);
file_buffer_mode_t GetFileBufferMode(const file_buffer_t *fb);
size_t GetFileBufferSize(const file_buffer_t *fb);

bit_file_buffer_t *AllocateBitFileBuffer(void);
void FreeBitFileBuffer(bit_file_buffer_t *bb);
void InitBitFileBuffer(bit_file_buffer_t *bb, file_buffer_t *fb);
void UninitBitFileBuffer(bit_file_buffer_t *bb);
int SetBitFileBufferMode(bit_file_buffer_t *bb, file_buffer_mode_t mode);
int ResetBitFileBuffer(bit_file_buffer_t *bb, file_buffer_mode_t mode);
int EndOfBitFileBuffer(const bit_file_buffer_t *bb);
void GetActualBitFileOffset(const bit_file_buffer_t *bb, io_int_t *byte_offset, uint8_t *bit_offset);
/* whole streams: bytes in memory order, `nbits` of them (a trailing fraction is left aligned); returns bits moved */
io_int_t ReadBitFileBuffer(bit_file_buffer_t *bb, uint8_t *dst, size_t nbits);
io_int_t WriteBitFileBuffer(bit_file_buffer_t *bb, const uint8_t *src, size_t nbits);
/* one value: its low `nbits` bits, most significant first */
io_int_t ReadSingleValueFromBitFileBuffer(bit_file_buffer_t *bb, io_uint_t *value, size_t nbits);
io_int_t WriteSingleValueToBitFileBuffer(bit_file_buffer_t *bb, const io_uint_t *value, size_t nbits);

/* ---- 4. codecs and options -------------------------------------------------------------------------------------------
 * options_t keeps the reference's field order (enc_dec.h:28-41) so that its layout is a prefix of this one; the one
 * addition sits at the end. */
typedef struct options_t options_t;
typedef io_int_t enc_dec_function_t(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options);
typedef struct enc_dec_t
{
  enc_dec_function_t *const encoder;
  enc_dec_function_t *const decoder; /* may be NULL */
} enc_dec_t;
typedef enum option_type_t { OT_INVALID = 0, OT_BOOL, OT_SIZE, OT_FLOAT, OT_CHAR } option_type_t;
struct options_t
{
  FILE *error_log_file;
  int encode;
  const enc_dec_t *encoder_decoder;
  size_t block_size_bits;
  size_t value_size_bits;
  int adaptive;
  size_t column;
  char separator_char;
  size_t num_decimal_places;
  float normalization_factor;
  size_t num_values;
  size_t num_channels; /* NOT in the reference: channels in one stream -- interleaved samples (dega / fdega), equal pieces (glzmh); default 1 */
};

/* the table of codecs: sorted by name, looked up by bsearch with a prefix comparison (enc_dec.c:89-98) */
size_t GetNumberOfEncoders(void);
void GetEncoderNames(const char **names);
const enc_dec_t *GetEncoder(const char *name);
const char *GetEncoderDescription(const char *name);
const char *GetEncoderNameFromFunction(enc_dec_function_t *function, int encoder);
int EncoderSupportsOption(const char *encoder_name, const char *option_name);
int EncoderFromFunctionSupportsOption(enc_dec_function_t *function, int encoder, const char *option_name);
/* the table of options */
size_t GetNumberOfOptions(void);
void GetOptionNames(const char **names);
int OptionNameExists(const char *name);
const char *GetOptionDescription(const char *name);
option_type_t GetOptionType(const char *name);
int GetAllowedOptionValueRange(const char *name, int *restricted, size_t *min, size_t *max);
/* typed access by name */
void SetDefaultOptions(options_t *options);
int GetOptionValueBool(const options_t *options, const char *name, int *value);
int GetOptionValueSize(const options_t *options, const char *name, size_t *value);
int GetOptionValueFloat(const options_t *options, const char *name, float *value);
int GetOptionValueChar(const options_t *options, const char *name, char *value);
int SetOptionValueBool(options_t *options, const char *name, int value);
int SetOptionValueSize(options_t *options, const char *name, size_t value);
int SetOptionValueFloat(options_t *options, const char *name, float value);
int SetOptionValueChar(options_t *options, const char *name, char value);

/* ---- 5. the codecs registered in src/enc_dec.c ----------------------------------------------------------------------- */
enc_dec_function_t CopyBits;                         /* "copy":  plumbing / tests, no GPU */
enc_dec_function_t EncodeDEGA, DecodeDEGA;           /* "dega":  valuesize-bit big-endian values <-> DEGA stream (GPU) */
enc_dec_function_t EncodeDEGAFloat, DecodeDEGAFloat; /* "fdega": raw float32 <-> DEGA stream, normalize fused (GPU) */
enc_dec_function_t EncodeLZMHGPU, DecodeLZMHGPU;     /* "glzmh": bytes <-> the reference's LZMH stream (GPU) */

#endif
