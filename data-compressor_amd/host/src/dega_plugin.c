/* dega_plugin.c -- the reference-side binding of libdega_hip.so: enc_dec_function_t implementations that a DCLib
 * encoders_decoders[] row can point at (DCLib/src/enc_dec.c:51-60; function type DCLib/inc/enc_dec.h:11).
 *
 * Written against the PUBLIC interface only (enc_dec.h, bit_file_buffer.h, err_codes.h), so the same file compiles
 * against this project's mirror headers (host/inc) and against the reference's own headers -- see INTEGRATION.md; the
 * oracle Makefile also links it into a DCCLI built from the reference's sources (oracle/_ref/DCCLI_gpu).
 *
 *   "dega"   encode: big-endian valuesize-bit integers in (what `encode normalize` emits) -> the stream that
 *                    `encode diff # encode seg # encode bac [adaptive]` would produce, bit for bit
 *            decode: the inverse (`decode bac [adaptive] # decode seg # decode diff`)
 *   "fdega"  the same with Normalize / Denormalize running inside the coder kernels: raw float32 in or out (what
 *            `decode csv` emits / `encode csv` eats), valuesize 1..64
 *   "glzmh"  the reference's second codec on the GPU: bytes in -> the stream `encode lzmh` produces, bit for bit, and
 *            its inverse (DCLib/src/lzmh.c:130-574).  Named so that it does not extend "lzmh" (prefix lookup).
 *
 * Contract kept: consume the input until EndOfBitFileBuffer, write all output with Write*, never
 * flush/close/free either buffer, return NO_ERROR or a negative code and log to options->error_log_file.
 * The work itself happens on the GPU -- on EVERY visible GPU for a batch: the binding holds one dega_hip_group (the
 * devices in DEGA_DEVICES, else all), which splits the channels into one range per device and concatenates the packed
 * streams on the host.  Without a GPU the codecs fail with ERROR_LIBRARY_INIT -- there is no CPU path here.
 *
 * num_channels=n (this project's mirror, and a reference tree with the edit of INTEGRATION.md 2b; the reference as shipped
 * has no batch notion):
 *   dega / fdega: the input is n interleaved channels, sample-major -- i.e. the [T][C] layout the kernels want -- and the
 *                 output is a small container (all integers big-endian):
 *                   "DEGB" | u32 version = 1 | u64 C | u64 T | C x u64 bit lengths | the C streams, each padded to a whole byte
 *   glzmh:        the input is cut into n pieces of ceil(bytes / n) bytes, each coded as a stream of its own:
 *                   "LZMB" | u32 version = 1 | u64 C | C x (u64 piece bytes, u64 bit length) | the C streams, byte padded
 */
#include "err_codes.h"
#include "enc_dec.h"
#include "dega_hip.h"

#include <stdlib.h>
#include <string.h>

#define LOG_TO(f, ...) do { if ((f) != NULL) fprintf((f), __VA_ARGS__); } while (0)

#define DEGA_PLUGIN_MAX_T ((size_t)1 << 25) /* samples per channel and call of the library (include/dega_hip.h) */

static dega_hip_group *g_group = NULL; /* like the reference's codecs, this binding is single threaded (bac.c:33-37) */

static io_int_t get_group(FILE *log, dega_hip_group **group)
{
  if (g_group == NULL)
  {
    const int ret = dega_hip_group_create(NULL, 0, &g_group); /* DEGA_DEVICES / DEGA_DEVICE restrict the set */
    if (ret != DEGA_OK)
    {
      LOG_TO(log, "dega: no usable GPU (%d visible): %s\n", dega_hip_device_count(), ERROR_MESSAGE_STRING(ret));
      g_group = NULL;
      return ret;
    }
  }
  *group = g_group;
  return NO_ERROR;
}

static size_t channels_of(const options_t *options)
{
#if defined(DC_AMD_ENC_DEC_H) || defined(DC_OPTIONS_HAVE_NUM_CHANNELS)
  /* this project's mirror header, or the reference's enc_dec.h with the num_channels edit of INTEGRATION.md 2b */
  return options->num_channels > 0 ? options->num_channels : 1;
#else
  (void)options;
  return 1; /* the reference's options_t as shipped has no batch dimension */
#endif
}

typedef struct byte_vec
{
  uint8_t *p;
  size_t n, cap;
} byte_vec;

static int vec_reserve(byte_vec *v, size_t need)
{
  if (need > v->cap)
  {
    size_t ncap = v->cap ? v->cap : 65536;
    uint8_t *np;
    while (ncap < need)
      ncap *= 2;
    if ((np = (uint8_t *)realloc(v->p, ncap)) == NULL)
      return ERROR_MEMORY;
    v->p = np;
    v->cap = ncap;
  }
  return NO_ERROR;
}

/* the whole remaining input as bytes + exact bit count */
static io_int_t slurp(bit_file_buffer_t *in, byte_vec *v, uint64_t *nbits)
{
  const size_t chunk_bits = 8u << 20;
  *nbits = 0;
  v->n = 0;
  while (!EndOfBitFileBuffer(in))
  {
    io_int_t got;
    if (vec_reserve(v, v->n + chunk_bits / 8 + 8) != NO_ERROR)
      return ERROR_MEMORY;
    if ((*nbits & 7) != 0)
      return ERROR_LIBRARY_CALL; /* a fractional chunk can only be the last one */
    got = ReadBitFileBuffer(in, v->p + v->n, chunk_bits);
    if (got < 0)
      return got;
    *nbits += (uint64_t)got;
    v->n += ((size_t)got + 7) / 8;
    if (got == 0)
      break;
  }
  if (vec_reserve(v, v->n + 16) != NO_ERROR)
    return ERROR_MEMORY;
  memset(v->p + v->n, 0, 16);
  return NO_ERROR;
}

static void put_be64(uint8_t *p, uint64_t v)
{
  int i;
  for (i = 0; i < 8; i++)
    p[i] = (uint8_t)(v >> (56 - 8 * i));
}

static uint64_t get_be64(const uint8_t *p)
{
  uint64_t v = 0;
  int i;
  for (i = 0; i < 8; i++)
    v = (v << 8) | p[i];
  return v;
}

static io_int_t first_error(const int32_t *err, size_t n, FILE *log, const char *what)
{
  size_t c;
  for (c = 0; c < n; c++)
    if (err[c] != 0)
    {
      LOG_TO(log, "dega: %s while %s channel %lu\n", ERROR_MESSAGE_STRING(err[c]), what, (unsigned long)c);
      return err[c];
    }
  return NO_ERROR;
}

/* a * b without wrapping, 0 on overflow */
static size_t mul_or_zero(size_t a, size_t b)
{
  if (a != 0 && b > (size_t)-1 / a)
    return 0;
  return a * b;
}

static void fill_job(dega_hip_job *job, size_t C, size_t T, const options_t *options, int samples)
{
  job->C = C;
  job->T = T;
  job->ld = C;
  job->adaptive = options->adaptive;
  job->valuesize = (int)options->value_size_bits;
  job->samples = samples;
  job->factor = options->normalization_factor;
}

/* ---- encode ---------------------------------------------------------------------------------------------------------- */

static io_int_t encode_common(bit_file_buffer_t *const in, bit_file_buffer_t *const out, const options_t *const options, int is_float)
{
  FILE *const log = options->error_log_file;
  const size_t C = channels_of(options);
  dega_hip_group *group;
  dega_hip_job job;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0;
  uint8_t *packed = NULL;
  uint64_t *bits = NULL, *offsets = NULL;
  int32_t *err = NULL;
  void *unpacked = NULL; /* native containers for the value sizes that are not a whole 32-bit word */
  const void *samples;
  size_t T, c, t, packed_cap;
  const size_t vs = options->value_size_bits;
  const size_t in_vs = is_float ? 32 : vs; /* normalize always reads 32-bit floats (normalize.c:15) */
  int attempt, kind;
  io_int_t ret;

  if (vs < 1 || vs > 64)
  {
    LOG_TO(log, "dega: valuesize 1..64 is supported on the GPU path\n");
    return ERROR_INVALID_VALUE;
  }
  if ((ret = get_group(log, &group)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (nbits % in_vs != 0) /* the reference's READ_VALUE_BITS_CHECKED would stop on the short last value */
  {
    LOG_TO(log, "Only read %lu bits instead of %lu\n", (unsigned long)(nbits % in_vs), (unsigned long)in_vs);
    ret = ERROR_LIBRARY_CALL;
    goto done;
  }
  if ((nbits / in_vs) % C != 0)
  {
    LOG_TO(log, "dega: %lu values do not divide into %lu channels\n", (unsigned long)(nbits / in_vs), (unsigned long)C);
    ret = ERROR_INVALID_VALUE;
    goto done;
  }
  T = (size_t)(nbits / in_vs) / C;
  bits = (uint64_t *)calloc(C, sizeof(uint64_t));
  offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
  err = (int32_t *)calloc(C, sizeof(int32_t));
  if (bits == NULL || offsets == NULL || err == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  if (is_float) /* raw native-endian float32, as `decode csv` writes them (csv.c:13-44): the kernel normalizes */
  {
    kind = DEGA_SAMPLES_F32;
    samples = raw.p;
  }
  else if (vs == 32) /* big-endian words, as `encode normalize` writes them: the kernel swaps */
  {
    kind = DEGA_SAMPLES_BE32;
    samples = raw.p;
  }
  else if (vs > 32) /* valuesize-bit values, MSB first, zero extended -> int64 containers */
  {
    int64_t *x64 = (int64_t *)malloc((T * C + 1) * sizeof(int64_t));
    if ((unpacked = x64) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    for (t = 0; t < T * C; t++)
    {
      uint64_t v = 0;
      size_t k;
      for (k = 0; k < vs; k++)
      {
        const uint64_t at = (uint64_t)t * vs + k;
        v = (v << 1) | ((raw.p[at >> 3] >> (7 - (at & 7))) & 1u);
      }
      x64[t] = (int64_t)v;
    }
    kind = DEGA_SAMPLES_I64;
    samples = x64;
  }
  else /* 1..31 bits: int32 containers (diff.c:15 does not sign extend) */
  {
    int32_t *x = (int32_t *)malloc((T * C + 1) * sizeof(int32_t));
    if ((unpacked = x) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    for (t = 0; t < T * C; t++)
    {
      const uint64_t at = (uint64_t)t * vs;
      uint64_t w = 0;
      size_t k;
      for (k = 0; k < 5; k++)
        w = (w << 8) | raw.p[(at >> 3) + k]; /* slurp() pads the buffer with 16 zero bytes */
      x[t] = (int32_t)((w >> (40 - (at & 7) - vs)) & (((uint64_t)1 << vs) - 1));
    }
    kind = DEGA_SAMPLES_I32;
    samples = x;
  }
  fill_job(&job, C, T, options, kind);
  /* the streams come back packed, channel after channel -- which is how they are written anyway; a stream is rarely
     longer than its samples, and if this batch's are the call says how much room it needs */
  packed_cap = C * (T * (vs > 32 ? 8 : 4) + 64);
  for (attempt = 0;; attempt++)
  {
    free(packed);
    if ((packed = (uint8_t *)malloc(packed_cap + 1)) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    ret = dega_hip_group_encode(group, &job, samples, packed, packed_cap, offsets, bits, err);
    if (ret != DEGA_ERROR_MEMORY || attempt > 0 || offsets[C] <= packed_cap)
      break;
    packed_cap = (size_t)offsets[C];
  }
  if (ret != DEGA_OK)
  {
    LOG_TO(log, "dega: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_group_last_error(group));
    goto done;
  }
  if ((ret = first_error(err, C, log, "encoding")) != NO_ERROR)
    goto done;
  if (C == 1) /* a bare stream with its exact bit length: identical to the reference chain's output */
  {
    if (WriteBitFileBuffer(out, packed, (size_t)bits[0]) != (io_int_t)bits[0])
      ret = ERROR_LIBRARY_CALL;
  }
  else
  {
    uint8_t head[24];
    memcpy(head, "DEGB", 4);
    head[4] = head[5] = head[6] = 0;
    head[7] = 1;
    put_be64(head + 8, (uint64_t)C);
    put_be64(head + 16, (uint64_t)T);
    if (WriteBitFileBuffer(out, head, 8 * sizeof(head)) != (io_int_t)(8 * sizeof(head)))
      ret = ERROR_LIBRARY_CALL;
    for (c = 0; c < C && ret == NO_ERROR; c++)
    {
      put_be64(head, bits[c]);
      if (WriteBitFileBuffer(out, head, 64) != 64)
        ret = ERROR_LIBRARY_CALL;
    }
    if (ret == NO_ERROR && WriteBitFileBuffer(out, packed, 8 * (size_t)offsets[C]) != (io_int_t)(8 * offsets[C]))
      ret = ERROR_LIBRARY_CALL;
  }
done:
  free(raw.p);
  free(packed);
  free(bits);
  free(offsets);
  free(err);
  free(unpacked);
  return ret;
}

io_int_t EncodeDEGA(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  return encode_common(in_bit_buf, out_bit_buf, options, 0);
}

io_int_t EncodeDEGAFloat(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  return encode_common(in_bit_buf, out_bit_buf, options, 1);
}

/* ---- decode ---------------------------------------------------------------------------------------------------------- */

static io_int_t decode_common(bit_file_buffer_t *const in, bit_file_buffer_t *const out, const options_t *const options, int is_float)
{
  FILE *const log = options->error_log_file;
  dega_hip_group *group;
  dega_hip_job job;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0;
  uint64_t *bits = NULL, *counts = NULL, *offsets = NULL;
  int32_t *err = NULL;
  void *x = NULL;
  size_t C = 1, T = 0, c, t, packed_at = 0;
  const size_t vs = options->value_size_bits;
  const int kind = is_float ? DEGA_SAMPLES_F32 : (vs == 32 ? DEGA_SAMPLES_BE32 : (vs > 32 ? DEGA_SAMPLES_I64 : DEGA_SAMPLES_I32));
  const size_t elem = kind == DEGA_SAMPLES_I64 ? 8 : 4;
  int known_T = 0;
  io_int_t ret;

  if (vs < 1 || vs > 64)
  {
    LOG_TO(log, "dega: valuesize 1..64 is supported on the GPU path\n");
    return ERROR_INVALID_VALUE;
  }
  if ((ret = get_group(log, &group)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (channels_of(options) > 1) /* container */
  {
    if (nbits < 8 * 24 || memcmp(raw.p, "DEGB", 4) != 0 || raw.p[7] != 1)
    {
      LOG_TO(log, "dega: not a DEGB container\n");
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    C = (size_t)get_be64(raw.p + 8);
    T = (size_t)get_be64(raw.p + 16);
    known_T = 1;
    /* nothing of the header is trusted: the counts must match the option, fit the library and the bytes that are there */
    if (C != channels_of(options) || T > DEGA_PLUGIN_MAX_T || mul_or_zero(C, 8) == 0 || (raw.n - 24) / 8 < C ||
        (T != 0 && mul_or_zero(mul_or_zero(T, C), elem) == 0))
    {
      LOG_TO(log, "dega: damaged DEGB header\n");
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    packed_at = 24 + 8 * C;
  }
  bits = (uint64_t *)calloc(C, sizeof(uint64_t));
  offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
  err = (int32_t *)calloc(C, sizeof(int32_t));
  counts = (uint64_t *)calloc(C, sizeof(uint64_t));
  if (bits == NULL || offsets == NULL || err == NULL || counts == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  if (known_T)
  {
    const uint64_t room = (uint64_t)(raw.n - packed_at);
    for (c = 0; c < C; c++)
    {
      bits[c] = get_be64(raw.p + 24 + 8 * c);
      if (bits[c] / 8 > room - offsets[c]) /* compared without rounding up first: a length near 2^64 must not wrap */
      {
        LOG_TO(log, "dega: damaged DEGB header (channel %lu)\n", (unsigned long)c);
        ret = ERROR_INVALID_FORMAT;
        goto done;
      }
      offsets[c + 1] = offsets[c] + bits[c] / 8 + ((bits[c] & 7) != 0 ? 1 : 0);
    }
    if (offsets[C] > room)
    {
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
  }
  else /* a bare stream: the sample count is only implied by the EOF symbol (bac.c:256) */
  {
    bits[0] = nbits;
    offsets[1] = raw.n;
    /* first guess: no sample takes less than one coded bit's worth of a well-compressed stream; grown while the stream
       holds more, up to what one call of the library takes */
    T = nbits < 4096 ? 8192 : (size_t)(2 * nbits);
    if (T > DEGA_PLUGIN_MAX_T)
      T = DEGA_PLUGIN_MAX_T;
  }
  for (;;)
  {
    free(x);
    if ((x = malloc((T * C + 1) * elem)) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    fill_job(&job, C, T, options, kind);
    ret = dega_hip_group_decode(group, &job, raw.p + packed_at, offsets, bits, x, known_T ? NULL : counts, err);
    if (ret != DEGA_OK)
    {
      LOG_TO(log, "dega: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_group_last_error(group));
      goto done;
    }
    if (!known_T && err[0] == ERROR_MEMORY && T < DEGA_PLUGIN_MAX_T) /* more samples than room: try again */
    {
      T = T > DEGA_PLUGIN_MAX_T / 4 ? DEGA_PLUGIN_MAX_T : T * 4;
      continue;
    }
    break;
  }
  if (!known_T && err[0] == ERROR_MEMORY && T >= DEGA_PLUGIN_MAX_T)
    LOG_TO(log, "dega: the stream holds more than %lu samples, the most one channel may have per call of the library (the reference has no such limit: "
                "split the series, or code it as a batch with num_channels)\n", (unsigned long)DEGA_PLUGIN_MAX_T);
  if ((ret = first_error(err, C, log, "decoding")) != NO_ERROR)
    goto done;
  if (!known_T)
    T = (size_t)counts[0];
  if (kind == DEGA_SAMPLES_F32 || kind == DEGA_SAMPLES_BE32)
  {
    /* raw native-endian float32 (normalize.c:39) / big-endian 32-bit values (diff.c:34 through the value writer): the
       bytes are already what the stream holds */
    if (WriteBitFileBuffer(out, (const uint8_t *)x, T * C * 32) != (io_int_t)(T * C * 32))
      ret = ERROR_LIBRARY_CALL;
  }
  else
    for (t = 0; t < T * C && ret == NO_ERROR; t++)
    {
      const io_uint_t v = vs > 32 ? (io_uint_t)((const int64_t *)x)[t] : (io_uint_t)((const uint32_t *)x)[t];
      if (WriteSingleValueToBitFileBuffer(out, &v, vs) != (io_int_t)vs)
        ret = ERROR_LIBRARY_CALL;
    }
done:
  free(raw.p);
  free(bits);
  free(offsets);
  free(counts);
  free(err);
  free(x);
  return ret;
}

io_int_t DecodeDEGA(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  return decode_common(in_bit_buf, out_bit_buf, options, 0);
}

io_int_t DecodeDEGAFloat(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  return decode_common(in_bit_buf, out_bit_buf, options, 1);
}

/* ---- LZMH on the GPU ("glzmh") ----------------------------------------------------------------------------------------- */

io_int_t EncodeLZMHGPU(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  FILE *const log = options->error_log_file;
  const size_t C = channels_of(options);
  dega_hip_group *group;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0, total;
  uint64_t *in_len = NULL, *out_bits = NULL, *offsets = NULL;
  uint8_t *in = NULL, *out = NULL;
  int32_t *err = NULL;
  size_t stride, out_cap, piece, c;
  io_int_t ret;

  if ((ret = get_group(log, &group)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in_bit_buf, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (nbits % 8 != 0) /* the reference's READ_VALUE_BITS_CHECKED(8) would stop on the short last byte (lzmh.c:163) */
  {
    LOG_TO(log, "Only read %lu bits instead of 8\n", (unsigned long)(nbits % 8));
    ret = ERROR_LIBRARY_CALL;
    goto done;
  }
  total = nbits / 8;
  piece = C > 1 ? (size_t)((total + C - 1) / C) : (size_t)total; /* one channel = the whole input, as the reference codes it */
  stride = (piece + 16) / 16 * 16;
  /* the streams come back packed, one behind the other (a stream is at most 10 bits per byte of text): every visible GPU
     codes a range of the pieces, the library's host pipeline overlaps copies and kernels (dega_hip_group_lzmh_encode) */
  out_cap = (size_t)(total + total / 4) + 64 * C + 64;
  in_len = (uint64_t *)calloc(C, sizeof(uint64_t));
  out_bits = (uint64_t *)calloc(C, sizeof(uint64_t));
  offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
  err = (int32_t *)calloc(C, sizeof(int32_t));
  if (in_len == NULL || out_bits == NULL || offsets == NULL || err == NULL || mul_or_zero(C, stride) == 0 || posix_memalign((void **)&in, 16, C * stride) != 0 ||
      (out = (uint8_t *)malloc(out_cap)) == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  memset(in, 0, C * stride);
  for (c = 0; c < C; c++)
  {
    const uint64_t at = (uint64_t)c * piece;
    in_len[c] = at >= total ? 0 : (total - at < piece ? total - at : piece);
    if (in_len[c] > 0)
      memcpy(in + c * stride, raw.p + at, (size_t)in_len[c]);
  }
  if ((ret = dega_hip_group_lzmh_encode(group, in, stride, in_len, C, out, out_cap, offsets, out_bits, err)) != DEGA_OK)
  {
    LOG_TO(log, "glzmh: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_group_last_error(group));
    goto done;
  }
  if ((ret = first_error(err, C, log, "encoding")) != NO_ERROR)
    goto done;
  if (C == 1)
  {
    if (WriteBitFileBuffer(out_bit_buf, out, (size_t)out_bits[0]) != (io_int_t)out_bits[0])
      ret = ERROR_LIBRARY_CALL;
  }
  else
  {
    uint8_t head[16];
    memcpy(head, "LZMB", 4);
    head[4] = head[5] = head[6] = 0;
    head[7] = 1;
    put_be64(head + 8, (uint64_t)C);
    if (WriteBitFileBuffer(out_bit_buf, head, 8 * sizeof(head)) != (io_int_t)(8 * sizeof(head)))
      ret = ERROR_LIBRARY_CALL;
    for (c = 0; c < C && ret == NO_ERROR; c++)
    {
      put_be64(head, in_len[c]);
      put_be64(head + 8, out_bits[c]);
      if (WriteBitFileBuffer(out_bit_buf, head, 128) != 128)
        ret = ERROR_LIBRARY_CALL;
    }
    if (ret == NO_ERROR && offsets[C] > 0 && WriteBitFileBuffer(out_bit_buf, out, (size_t)offsets[C] * 8) != (io_int_t)(offsets[C] * 8))
      ret = ERROR_LIBRARY_CALL; /* the streams, each padded to a whole byte, exactly as the library packed them */
  }
done:
  free(raw.p);
  free(in);
  free(out);
  free(in_len);
  free(out_bits);
  free(offsets);
  free(err);
  return ret;
}

io_int_t DecodeLZMHGPU(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  FILE *const log = options->error_log_file;
  dega_hip_group *group;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0;
  uint64_t *in_bits = NULL, *want_len = NULL, *out_len = NULL, *offsets = NULL;
  uint8_t *out = NULL;
  int32_t *err = NULL;
  size_t C = 1, stride, c, at;
  int attempt, container = 0;
  io_int_t ret;

  if ((ret = get_group(log, &group)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in_bit_buf, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (channels_of(options) > 1)
  {
    if (nbits < 8 * 16 || memcmp(raw.p, "LZMB", 4) != 0 || raw.p[7] != 1)
    {
      LOG_TO(log, "glzmh: not an LZMB container\n");
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    C = (size_t)get_be64(raw.p + 8);
    if (C != channels_of(options) || mul_or_zero(C, 16) == 0 || (raw.n - 16) / 16 < C)
    {
      LOG_TO(log, "glzmh: damaged LZMB header\n");
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    container = 1;
  }
  in_bits = (uint64_t *)calloc(C, sizeof(uint64_t));
  want_len = (uint64_t *)calloc(C, sizeof(uint64_t));
  out_len = (uint64_t *)calloc(C, sizeof(uint64_t));
  err = (int32_t *)calloc(C, sizeof(int32_t));
  offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
  if (in_bits == NULL || want_len == NULL || out_len == NULL || err == NULL || offsets == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  if (container)
  {
    uint64_t longest_text = 0, sum = 0;
    const uint64_t room = (uint64_t)raw.n - 16 - 16 * (uint64_t)C;
    for (c = 0; c < C; c++)
    {
      want_len[c] = get_be64(raw.p + 16 + 16 * c);
      in_bits[c] = get_be64(raw.p + 24 + 16 * c);
      if (in_bits[c] / 8 > room - sum || want_len[c] > ((uint64_t)1 << 31))
      {
        LOG_TO(log, "glzmh: damaged LZMB header (piece %lu)\n", (unsigned long)c);
        ret = ERROR_INVALID_FORMAT;
        goto done;
      }
      offsets[c] = sum; /* the streams lie packed in the container, each padded to a whole byte: used where they are */
      sum += (in_bits[c] + 7) / 8;
      longest_text = want_len[c] > longest_text ? want_len[c] : longest_text;
    }
    offsets[C] = sum;
    if (sum > room)
    {
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    stride = ((size_t)longest_text + 8 + 7) / 8 * 8;
  }
  else
  {
    in_bits[0] = nbits;
    offsets[1] = (nbits + 7) / 8;
    /* a bare LZMH stream does not say how long its text is: start from 4x and grow while the row overflows */
    stride = (4 * raw.n + 4096) / 8 * 8;
  }
  at = container ? 16 + 16 * C : 0;
  for (attempt = 0;; attempt++)
  {
    free(out);
    if (mul_or_zero(C, stride) == 0 || (out = (uint8_t *)malloc(C * stride)) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    if ((ret = dega_hip_group_lzmh_decode(group, raw.p + at, offsets, in_bits, C, out, stride, out_len, err)) != DEGA_OK)
    {
      LOG_TO(log, "glzmh: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_group_last_error(group));
      goto done;
    }
    if (container || err[0] != ERROR_MEMORY || attempt == 5)
      break;
    stride *= 8;
  }
  if ((ret = first_error(err, C, log, "decoding")) != NO_ERROR)
    goto done;
  for (c = 0; c < C && ret == NO_ERROR; c++)
  {
    if (container && out_len[c] != want_len[c])
    {
      LOG_TO(log, "glzmh: piece %lu decodes to %lu bytes, the container says %lu\n", (unsigned long)c, (unsigned long)out_len[c], (unsigned long)want_len[c]);
      ret = ERROR_INVALID_FORMAT;
    }
    else if (WriteBitFileBuffer(out_bit_buf, out + c * stride, (size_t)out_len[c] * 8) != (io_int_t)(out_len[c] * 8))
      ret = ERROR_LIBRARY_CALL;
  }
done:
  free(raw.p);
  free(out);
  free(in_bits);
  free(want_len);
  free(out_len);
  free(offsets);
  free(err);
  return ret;
}
