/* enc_dec.c -- plugin and option tables.  Same lookup semantics as the reference's DCLib/src/enc_dec.c:
 *   - both tables are sorted by name and searched with bsearch + a comparator that compares only strlen(table name)
 *     characters (enc_dec.c:89-98, 135-144), so a key that merely STARTS with a registered name finds it; new names
 *     therefore must not extend an existing one ("dega", "fdega" are safe; "bac_gpu" would not be);
 *   - each codec row carries the bitmask of options it accepts (enc_dec.c:51-60);
 *   - options are stored through offsetof() into options_t by typed setters (enc_dec.c:227-283);
 *   - defaults as enc_dec.c:187-197 (valuesize 32, normalization_factor 100, adaptive off, ...).
 * Rows: the GPU-backed DEGA codecs of this project plus "copy" (host-only plumbing check).  The reference's CPU codecs
 * are NOT re-implemented here: this library is the drop-in for the DEGA path only. */
#include "enc_dec.h"
#include "err_codes.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

enum
{
  OPT_BLOCKSIZE = 1 << 0,
  OPT_VALUESIZE = 1 << 1,
  OPT_ADAPTIVE = 1 << 2,
  OPT_COLUMN = 1 << 3,
  OPT_SEPARATOR = 1 << 4,
  OPT_NORMALIZATION = 1 << 5,
  OPT_DECIMALS = 1 << 6,
  OPT_NUM_VALUES = 1 << 7,
  OPT_NUM_CHANNELS = 1 << 8
};

typedef struct codec_row_t
{
  const char *name;
  const char *description;
  enc_dec_t functions;
  unsigned options;
} codec_row_t;

typedef struct option_row_t
{
  const char *name;
  unsigned bit;
  const char *description;
  option_type_t type;
  size_t min_value, max_value;
  size_t offset;
} option_row_t;

static const codec_row_t codecs[] = { /* sorted by name */
  { "copy", "Copies input to output", { &CopyBits, &CopyBits }, OPT_BLOCKSIZE },
  { "dega", "diff + seg + bac on the GPU (MI355X), big-endian integer values in", { &EncodeDEGA, &DecodeDEGA }, OPT_ADAPTIVE | OPT_VALUESIZE | OPT_NUM_CHANNELS },
  { "fdega", "normalize + diff + seg + bac on the GPU (MI355X), raw floats in", { &EncodeDEGAFloat, &DecodeDEGAFloat }, OPT_ADAPTIVE | OPT_VALUESIZE | OPT_NORMALIZATION | OPT_NUM_CHANNELS },
  { "glzmh", "LZMH on the GPU (MI355X): the stream of the reference's lzmh, bit for bit", { &EncodeLZMHGPU, &DecodeLZMHGPU }, OPT_NUM_CHANNELS },
};
static const size_t num_codecs = sizeof(codecs) / sizeof(codecs[0]);

static const option_row_t option_rows[] = { /* sorted by name */
  { "adaptive", OPT_ADAPTIVE, "Perform adaptive arithmetic coding", OT_BOOL, 0, 1, offsetof(options_t, adaptive) },
  { "blocksize", OPT_BLOCKSIZE, "Use blocks of <n> bits size for I/O", OT_SIZE, 1, SIZE_MAX, offsetof(options_t, block_size_bits) },
  { "column", OPT_COLUMN, "Use column <n>", OT_SIZE, 1, SIZE_MAX, offsetof(options_t, column) },
  { "normalization_factor", OPT_NORMALIZATION, "Use multiplier <n> for normalization and <1/n> for denormalization", OT_FLOAT, 0, SIZE_MAX, offsetof(options_t, normalization_factor) },
  { "num_channels", OPT_NUM_CHANNELS, "Treat the input as <n> channels coded as one batch on the GPU (dega / fdega: interleaved samples; glzmh: <n> equal pieces)", OT_SIZE, 1, SIZE_MAX, offsetof(options_t, num_channels) },
  { "num_decimal_places", OPT_DECIMALS, "Use <n> decimal places to print floats into CSV files", OT_SIZE, 0, 6, offsetof(options_t, num_decimal_places) },
  { "num_values", OPT_NUM_VALUES, "Use <n> values for aggregation", OT_SIZE, 0, SIZE_MAX, offsetof(options_t, num_values) },
  { "separator_char", OPT_SEPARATOR, "Use <n> as CSV entry separator", OT_CHAR, 0, CHAR_MAX, offsetof(options_t, separator_char) },
  { "valuesize", OPT_VALUESIZE, "Use values of <n> bits size", OT_SIZE, 1, IO_SIZE_BITS, offsetof(options_t, value_size_bits) },
};
static const size_t num_option_rows = sizeof(option_rows) / sizeof(option_rows[0]);

static int cmp_codec(const void *key, const void *row)
{
  const char *name = ((const codec_row_t *)row)->name;
  return strncmp((const char *)key, name, strlen(name)); /* prefix match, as the reference */
}

static int cmp_option(const void *key, const void *row)
{
  const char *name = ((const option_row_t *)row)->name;
  return strncmp((const char *)key, name, strlen(name));
}

static const codec_row_t *find_codec(const char *name)
{
  return (const codec_row_t *)bsearch(name, codecs, num_codecs, sizeof(codecs[0]), cmp_codec);
}

static const option_row_t *find_option(const char *name)
{
  return (const option_row_t *)bsearch(name, option_rows, num_option_rows, sizeof(option_rows[0]), cmp_option);
}

size_t GetNumberOfEncoders(void)
{
  return num_codecs;
}

void GetEncoderNames(const char **names)
{
  size_t i;
  for (i = 0; i < num_codecs; i++)
    names[i] = codecs[i].name;
}

const enc_dec_t *GetEncoder(const char *name)
{
  const codec_row_t *r = find_codec(name);
  return r ? &r->functions : NULL;
}

const char *GetEncoderDescription(const char *name)
{
  const codec_row_t *r = find_codec(name);
  return r ? r->description : NULL;
}

const char *GetEncoderNameFromFunction(enc_dec_function_t *function, int encoder)
{
  size_t i;
  for (i = 0; i < num_codecs; i++)
    if ((encoder ? codecs[i].functions.encoder : codecs[i].functions.decoder) == function)
      return codecs[i].name;
  return NULL;
}

size_t GetNumberOfOptions(void)
{
  return num_option_rows;
}

void GetOptionNames(const char **names)
{
  size_t i;
  for (i = 0; i < num_option_rows; i++)
    names[i] = option_rows[i].name;
}

int OptionNameExists(const char *name)
{
  return find_option(name) != NULL;
}

const char *GetOptionDescription(const char *name)
{
  const option_row_t *o = find_option(name);
  return o ? o->description : NULL;
}

option_type_t GetOptionType(const char *name)
{
  const option_row_t *o = find_option(name);
  return o ? o->type : OT_INVALID;
}

int GetAllowedOptionValueRange(const char *name, int *restricted, size_t *min, size_t *max)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *restricted = o->type == OT_SIZE;
  if (o->type == OT_SIZE)
  {
    *min = o->min_value;
    *max = o->max_value;
  }
  return NO_ERROR;
}

int EncoderSupportsOption(const char *encoder_name, const char *option_name)
{
  const codec_row_t *r = find_codec(encoder_name);
  const option_row_t *o = find_option(option_name);
  return r != NULL && o != NULL && (r->options & o->bit) == o->bit;
}

/* enc_dec.c:216-225 of the reference: the option mask of the row a codec FUNCTION belongs to (DCCLI checks options of
   an already-resolved stage this way, DCCLI/src/cli.c:305) */
int EncoderFromFunctionSupportsOption(enc_dec_function_t *function, int encoder, const char *option_name)
{
  const option_row_t *o = find_option(option_name);
  size_t i;
  for (i = 0; i < num_codecs; i++)
    if ((encoder && codecs[i].functions.encoder == function) || (!encoder && codecs[i].functions.decoder == function))
      return o != NULL && (codecs[i].options & o->bit) == o->bit;
  return 0;
}

void SetDefaultOptions(options_t *options)
{
  options->adaptive = 0;
  options->block_size_bits = 8;
  options->column = 1;
  options->normalization_factor = 100;
  options->num_decimal_places = 2;
  options->separator_char = ',';
  options->value_size_bits = 32;
  options->num_values = 2;
  options->num_channels = 1;
}

#define FIELD(T, base, o) ((T *)((uint8_t *)(base) + (o)->offset))
#define CFIELD(T, base, o) ((const T *)((const uint8_t *)(base) + (o)->offset))

int GetOptionValueBool(const options_t *options, const char *name, int *value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *value = *CFIELD(int, options, o);
  return NO_ERROR;
}

int GetOptionValueSize(const options_t *options, const char *name, size_t *value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *value = *CFIELD(size_t, options, o);
  return NO_ERROR;
}

int GetOptionValueFloat(const options_t *options, const char *name, float *value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *value = *CFIELD(float, options, o);
  return NO_ERROR;
}

int GetOptionValueChar(const options_t *options, const char *name, char *value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *value = *CFIELD(char, options, o);
  return NO_ERROR;
}

int SetOptionValueBool(options_t *options, const char *name, int value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *FIELD(int, options, o) = value;
  return NO_ERROR;
}

int SetOptionValueSize(options_t *options, const char *name, size_t value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *FIELD(size_t, options, o) = value;
  return NO_ERROR;
}

int SetOptionValueFloat(options_t *options, const char *name, float value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *FIELD(float, options, o) = value;
  return NO_ERROR;
}

int SetOptionValueChar(options_t *options, const char *name, char value)
{
  const option_row_t *o = find_option(name);
  if (o == NULL)
    return ERROR_INVALID_VALUE;
  *FIELD(char, options, o) = value;
  return NO_ERROR;
}

/* "copy": blocksize bits at a time; a final short block is passed on as it is (plumbing check for the bit layer) */
io_int_t CopyBits(bit_file_buffer_t *const in, bit_file_buffer_t *const out, const options_t *const options)
{
  const size_t block = options->block_size_bits;
  uint8_t *buf = (uint8_t *)malloc((block + 7) / 8 + 1);
  io_int_t ret = NO_ERROR;
  if (buf == NULL)
    return ERROR_MEMORY;
  while (!EndOfBitFileBuffer(in))
  {
    const io_int_t got = ReadBitFileBuffer(in, buf, block);
    if (got < 0)
    {
      ret = got;
      break;
    }
    if (WriteBitFileBuffer(out, buf, (size_t)got) != got)
    {
      ret = ERROR_LIBRARY_CALL;
      break;
    }
  }
  free(buf);
  return ret;
}
