/* enc_dec.h -- the encoder/decoder plugin table and the option system: the drop-in boundary.
   Types, field order and function names follow the reference's DCLib/inc/enc_dec.h:10-71 so that a codec source file
   written against the reference compiles against this header and vice versa (src/dega_plugin.c does both).
   One addition, at the END of options_t so that the reference's layout is a prefix: num_channels (new surface). */
#ifndef DC_AMD_ENC_DEC_H
#define DC_AMD_ENC_DEC_H

#include "bit_file_buffer.h"

typedef struct options_t options_t;
typedef io_int_t enc_dec_function_t(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options);

typedef struct enc_dec_t
{
  enc_dec_function_t *const encoder;
  enc_dec_function_t *const decoder; /* may be NULL */
} enc_dec_t;

typedef enum option_type_t
{
  OT_INVALID = 0,
  OT_BOOL,
  OT_SIZE,
  OT_FLOAT,
  OT_CHAR
} option_type_t;

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
  size_t num_channels; /* NOT in the reference: channels interleaved in one stream (dega / fdega only), default 1 */
};

size_t GetNumberOfEncoders(void);
void GetEncoderNames(const char **encoder_names);
const enc_dec_t *GetEncoder(const char *name);
const char *GetEncoderDescription(const char *name);
const char *GetEncoderNameFromFunction(enc_dec_function_t *function, int encoder);

size_t GetNumberOfOptions(void);
void GetOptionNames(const char **option_names);
int OptionNameExists(const char *name);
const char *GetOptionDescription(const char *name);
option_type_t GetOptionType(const char *name);
int GetAllowedOptionValueRange(const char *name, int *restricted, size_t *min, size_t *max);
int EncoderSupportsOption(const char *encoder_name, const char *option_name);

void SetDefaultOptions(options_t *options);
int GetOptionValueBool(const options_t *options, const char *name, int *value);
int GetOptionValueSize(const options_t *options, const char *name, size_t *value);
int GetOptionValueFloat(const options_t *options, const char *name, float *value);
int GetOptionValueChar(const options_t *options, const char *name, char *value);
int SetOptionValueBool(options_t *options, const char *name, int value);
int SetOptionValueSize(options_t *options, const char *name, size_t value);
int SetOptionValueFloat(options_t *options, const char *name, float value);
int SetOptionValueChar(options_t *options, const char *name, char value);

/* the codecs registered in src/enc_dec.c */
enc_dec_function_t CopyBits;                       /* "copy": plumbing / tests, no GPU */
enc_dec_function_t EncodeDEGA, DecodeDEGA;         /* "dega":  big-endian int32 values <-> DEGA stream (GPU) */
enc_dec_function_t EncodeDEGAFloat, DecodeDEGAFloat; /* "fdega": raw float32 <-> DEGA stream, normalize fused (GPU) */
enc_dec_function_t EncodeLZMHGPU, DecodeLZMHGPU;   /* "glzmh": bytes <-> the reference's LZMH stream (GPU) */

#endif
