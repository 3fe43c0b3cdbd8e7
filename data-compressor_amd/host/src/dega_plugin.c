/* dega_plugin.c -- the reference-side binding of libdega_hip.so: enc_dec_function_t implementations that a DCLib
 * encoders_decoders[] row can point at (DCLib/src/enc_dec.c:51-60; function type DCLib/inc/enc_dec.h:11).
 *
 * Written against the PUBLIC interface only (enc_dec.h, bit_file_buffer.h, err_codes.h), so the same file compiles
 * against this project's mirror headers (host/inc) and against the reference's own headers -- see INTEGRATION.md.
 *
 *   "dega"   encode: big-endian valuesize-bit integers in (what `encode normalize` emits) -> the stream that
 *                    `encode diff # encode seg # encode bac [adaptive]` would produce, bit for bit
 *            decode: the inverse (`decode bac [adaptive] # decode seg # decode diff`)
 *   "fdega"  the same with normalize / denormalize fused in: raw float32 in or out (what `decode csv` emits / `encode csv` eats)
 *
 *   "glzmh"  the reference's second codec on the GPU: bytes in -> the stream `encode lzmh` produces, bit for bit, and
 *            its inverse (DCLib/src/lzmh.c:130-574).  Named so that it does not extend "lzmh" (prefix lookup).
 *
 * Contract kept: consume the input until EndOfBitFileBuffer, write all output with Write*, never
 * flush/close/free either buffer, return NO_ERROR or a negative code and log to options->error_log_file.
 * The work itself happens on the GPU; without one the codec fails with ERROR_LIBRARY_INIT -- there is no CPU path here.
 *
 * num_channels=n (this project's mirror only; the reference has no batch notion): the input is n interleaved channels,
 * sample-major -- i.e. the [T][C] layout the kernels want -- and the output is a small container:
 *   "DEGB" | u32 version = 1 | u64 C | u64 T | C x u64 bit lengths | the C streams, each padded to a whole byte   (big-endian)
 */
#include "err_codes.h"
#include "enc_dec.h"
#include "dega_hip.h"

#include <stdlib.h>
#include <string.h>

#define LOG_TO(f, ...) do { if ((f) != NULL) fprintf((f), __VA_ARGS__); } while (0)

static dega_hip_ctx *g_ctx = NULL; /* like the reference's codecs, this binding is single threaded (bac.c:33-37) */

static io_int_t get_context(FILE *log, dega_hip_ctx **ctx)
{
  if (g_ctx == NULL)
  {
    const char *dev = getenv("DEGA_DEVICE");
    const int ret = dega_hip_create(dev != NULL ? atoi(dev) : 0, &g_ctx);
    if (ret != DEGA_OK)
    {
      LOG_TO(log, "dega: no usable GPU (%d visible): %s\n", dega_hip_device_count(), ERROR_MESSAGE_STRING(ret));
      g_ctx = NULL;
      return ret;
    }
  }
  *ctx = g_ctx;
  return NO_ERROR;
}

static size_t channels_of(const options_t *options)
{
#ifdef DC_AMD_ENC_DEC_H
  return options->num_channels > 0 ? options->num_channels : 1;
#else
  (void)options;
  return 1; /* the reference's options_t has no batch dimension */
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

/* ---- encode ---------------------------------------------------------------------------------------------------------- */

static io_int_t encode_common(bit_file_buffer_t *const in, bit_file_buffer_t *const out, const options_t *const options, int is_float)
{
  FILE *const log = options->error_log_file;
  const size_t C = channels_of(options);
  dega_hip_ctx *ctx;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0;
  uint8_t *streams = NULL;
  uint64_t *bits = NULL;
  int32_t *err = NULL;
  int32_t *x = NULL;
  size_t T, cap, c, t, packed_bytes = 0;
  int packed_streams = 0;
  const size_t vs = options->value_size_bits;
  const size_t in_vs = is_float ? 32 : vs; /* normalize always reads 32-bit floats (normalize.c:15) */
  io_int_t ret;

  if (vs < 1 || vs > 64 || (is_float && vs > 32))
  {
    LOG_TO(log, "dega: valuesize 1..64 is supported on the GPU path (1..32 with the float entry)\n");
    return ERROR_INVALID_VALUE;
  }
  if ((ret = get_context(log, &ctx)) != NO_ERROR)
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
  cap = vs > 32 ? dega_hip_worst_case_bytes64(T) : dega_hip_worst_case_bytes(T);
  streams = (uint8_t *)malloc(C * cap);
  bits = (uint64_t *)calloc(C, sizeof(uint64_t));
  err = (int32_t *)calloc(C, sizeof(int32_t));
  x = (int32_t *)malloc((T * C + 1) * (vs > 32 ? sizeof(int64_t) : sizeof(int32_t)));
  if (streams == NULL || bits == NULL || err == NULL || x == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  if (is_float)
  {
    memcpy(x, raw.p, T * C * 4); /* raw native-endian float32, as `decode csv` writes them (csv.c:13-44) */
    ret = dega_hip_encode_f32_host(ctx, (const float *)(const void *)x, C, T, C, options->normalization_factor, options->adaptive, (int)vs,
                                   streams, cap, bits, err);
  }
  else if (vs > 32) /* 64-bit containers */
  {
    int64_t *const x64 = (int64_t *)(void *)x;
    for (t = 0; t < T * C; t++) /* valuesize-bit values, MSB first, zero extended */
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
    ret = dega_hip_encode64_host(ctx, x64, C, T, C, options->adaptive, (int)vs, streams, cap, bits, err);
  }
  else
  {
    if (vs == 32)
      for (t = 0; t < T * C; t++) /* big-endian values -> native int32 */
        x[t] = (int32_t)(((uint32_t)raw.p[4 * t] << 24) | ((uint32_t)raw.p[4 * t + 1] << 16) | ((uint32_t)raw.p[4 * t + 2] << 8) | raw.p[4 * t + 3]);
    else
      for (t = 0; t < T * C; t++) /* valuesize-bit values, MSB first, zero extended (diff.c:15 does not sign extend) */
      {
        const uint64_t at = (uint64_t)t * vs;
        uint64_t w = 0;
        size_t k;
        for (k = 0; k < 5; k++)
          w = (w << 8) | raw.p[(at >> 3) + k]; /* slurp() pads the buffer with 16 zero bytes */
        x[t] = (int32_t)((w >> (40 - (at & 7) - vs)) & (((uint64_t)1 << vs) - 1));
      }
    if (C > 1) /* the container wants the streams back to back: have them packed on the device, only they cross PCIe */
    {
      uint64_t *const offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
      packed_streams = 1;
      if (offsets == NULL)
        ret = ERROR_MEMORY;
      else
      {
        ret = dega_hip_encode_packed_host(ctx, x, C, T, C, options->adaptive, (int)vs, streams, C * cap, offsets, bits, err);
        packed_bytes = (size_t)offsets[C];
        free(offsets);
      }
    }
    else
      ret = dega_hip_encode_host(ctx, x, C, T, C, options->adaptive, (int)vs, streams, cap, bits, err);
  }
  if (ret != DEGA_OK)
  {
    LOG_TO(log, "dega: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_last_error(ctx));
    goto done;
  }
  if ((ret = first_error(err, C, log, "encoding")) != NO_ERROR)
    goto done;
  if (C == 1) /* a bare stream with its exact bit length: identical to the reference chain's output */
  {
    if (WriteBitFileBuffer(out, streams, (size_t)bits[0]) != (io_int_t)bits[0])
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
    if (packed_streams)
    {
      if (ret == NO_ERROR && WriteBitFileBuffer(out, streams, 8 * packed_bytes) != (io_int_t)(8 * packed_bytes))
        ret = ERROR_LIBRARY_CALL;
    }
    else
      for (c = 0; c < C && ret == NO_ERROR; c++)
      {
        const size_t nb = (size_t)((bits[c] + 7) / 8) * 8;
        if (WriteBitFileBuffer(out, streams + c * cap, nb) != (io_int_t)nb)
          ret = ERROR_LIBRARY_CALL;
      }
  }
done:
  free(raw.p);
  free(streams);
  free(bits);
  free(err);
  free(x);
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
  dega_hip_ctx *ctx;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0;
  uint8_t *streams = NULL;
  uint64_t *bits = NULL, *counts = NULL, *offsets = NULL;
  int32_t *err = NULL;
  int32_t *x = NULL;
  size_t C = 1, T = 0, cap = 0, c, t, packed_at = 0;
  const size_t vs = options->value_size_bits;
  int known_T = 0;
  io_int_t ret;

  if (vs < 1 || vs > 64 || (is_float && vs > 32))
  {
    LOG_TO(log, "dega: valuesize 1..64 is supported on the GPU path (1..32 with the float entry)\n");
    return ERROR_INVALID_VALUE;
  }
  if ((ret = get_context(log, &ctx)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (channels_of(options) > 1) /* container */
  {
    size_t off;
    if (nbits < 8 * 24 || memcmp(raw.p, "DEGB", 4) != 0 || raw.p[7] != 1)
    {
      LOG_TO(log, "dega: not a DEGB container\n");
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    C = (size_t)get_be64(raw.p + 8);
    T = (size_t)get_be64(raw.p + 16);
    known_T = 1;
    if (C != channels_of(options) || nbits / 8 < 24 + 8 * (uint64_t)C)
    {
      ret = ERROR_INVALID_FORMAT;
      goto done;
    }
    bits = (uint64_t *)calloc(C, sizeof(uint64_t));
    if (bits == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    for (c = 0; c < C; c++)
    {
      bits[c] = get_be64(raw.p + 24 + 8 * c);
      if ((bits[c] + 7) / 8 > cap)
        cap = (size_t)((bits[c] + 7) / 8);
    }
    cap = (cap + 16 + 3) & ~(size_t)3;
    off = 24 + 8 * C;
    if (!is_float && vs <= 32) /* the container's stream area is the packed form: hand it over as it is */
    {
      offsets = (uint64_t *)calloc(C + 1, sizeof(uint64_t));
      if (offsets == NULL)
      {
        ret = ERROR_MEMORY;
        goto done;
      }
      for (c = 0; c < C; c++)
        offsets[c + 1] = offsets[c] + (bits[c] + 7) / 8;
      if (off + offsets[C] > raw.n)
      {
        ret = ERROR_INVALID_FORMAT;
        goto done;
      }
      packed_at = off;
    }
    else
    {
      streams = (uint8_t *)calloc(C, cap);
      if (streams == NULL)
      {
        ret = ERROR_MEMORY;
        goto done;
      }
      for (c = 0; c < C; c++)
      {
        const size_t nb = (size_t)((bits[c] + 7) / 8);
        if (off + nb > raw.n)
        {
          ret = ERROR_INVALID_FORMAT;
          goto done;
        }
        memcpy(streams + c * cap, raw.p + off, nb);
        off += nb;
      }
    }
  }
  else /* a bare stream: the sample count is only implied by the EOF symbol (bac.c:256) */
  {
    bits = (uint64_t *)calloc(1, sizeof(uint64_t));
    cap = (raw.n + 16 + 3) & ~(size_t)3;
    streams = (uint8_t *)calloc(1, cap);
    if (bits == NULL || streams == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    memcpy(streams, raw.p, raw.n);
    bits[0] = nbits;
    T = (size_t)(nbits < 4096 ? 8192 : 2 * nbits); /* first guess; doubled while the stream holds more */
  }
  err = (int32_t *)calloc(C, sizeof(int32_t));
  counts = (uint64_t *)calloc(C, sizeof(uint64_t));
  if (err == NULL || counts == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  for (;;)
  {
    free(x);
    if ((x = (int32_t *)malloc((T * C + 1) * (vs > 32 ? sizeof(int64_t) : sizeof(int32_t)))) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    if (offsets != NULL) /* container of int32-sized values: packed streams in, only they cross PCIe */
      ret = dega_hip_decode_packed_host(ctx, raw.p + packed_at, offsets, bits, C, T, C, options->adaptive, (int)vs, x, NULL, err);
    else if (vs > 32) /* 64-bit containers: the variable-length entry serves both cases */
    {
      ret = dega_hip_decode64_var_host(ctx, streams, cap, bits, C, T, C, options->adaptive, (int)vs, (int64_t *)(void *)x, counts, err);
      if (ret == DEGA_OK && known_T)
        for (c = 0; c < C; c++)
          if (err[c] == NO_ERROR && counts[c] != T)
            err[c] = ERROR_INVALID_FORMAT;
    }
    else if (is_float)
      ret = known_T ? dega_hip_decode_f32_host(ctx, streams, cap, bits, C, T, C, options->normalization_factor, options->adaptive, (int)vs, (float *)(void *)x, err)
                    : dega_hip_decode_f32_var_host(ctx, streams, cap, bits, C, T, C, options->normalization_factor, options->adaptive, (int)vs, (float *)(void *)x, counts, err);
    else
      ret = known_T ? dega_hip_decode_host(ctx, streams, cap, bits, C, T, C, options->adaptive, (int)vs, x, err)
                    : dega_hip_decode_var_host(ctx, streams, cap, bits, C, T, C, options->adaptive, (int)vs, x, counts, err);
    if (ret != DEGA_OK)
    {
      LOG_TO(log, "dega: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_last_error(ctx));
      goto done;
    }
    if (!known_T && err[0] == ERROR_MEMORY && T < ((size_t)1 << 31)) /* more samples than room: try again */
    {
      T *= 4;
      continue;
    }
    break;
  }
  if ((ret = first_error(err, C, log, "decoding")) != NO_ERROR)
    goto done;
  if (!known_T)
    T = (size_t)counts[0];
  for (t = 0; t < T * C && ret == NO_ERROR; t++)
  {
    if (is_float)
    {
      if (WriteBitFileBuffer(out, (const uint8_t *)&x[t], 32) != 32) /* raw native-endian float32 (normalize.c:39) */
        ret = ERROR_LIBRARY_CALL;
    }
    else
    {
      const io_uint_t v = vs > 32 ? (io_uint_t)((const int64_t *)(const void *)x)[t] : (io_uint_t)(uint32_t)x[t];
      if (WriteSingleValueToBitFileBuffer(out, &v, vs) != (io_int_t)vs)
        ret = ERROR_LIBRARY_CALL;
    }
  }
done:
  free(raw.p);
  free(streams);
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
  dega_hip_ctx *ctx;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0, in_len, out_bits = 0;
  uint8_t *in = NULL, *out = NULL;
  int32_t err = 0;
  size_t stride, cap;
  io_int_t ret;

  if ((ret = get_context(log, &ctx)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in_bit_buf, &raw, &nbits)) != NO_ERROR)
    goto done;
  if (nbits % 8 != 0) /* the reference's READ_VALUE_BITS_CHECKED(8) would stop on the short last byte (lzmh.c:163) */
  {
    LOG_TO(log, "Only read %lu bits instead of 8\n", (unsigned long)(nbits % 8));
    ret = ERROR_LIBRARY_CALL;
    goto done;
  }
  in_len = nbits / 8;
  stride = ((size_t)in_len + 16) / 16 * 16;
  cap = dega_hip_lzmh_worst_case_bytes(stride);
  if (posix_memalign((void **)&in, 16, stride) != 0 || posix_memalign((void **)&out, 16, cap) != 0)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  memset(in, 0, stride);
  memcpy(in, raw.p, (size_t)in_len);
  if ((ret = dega_hip_lzmh_encode_host(ctx, in, stride, &in_len, 1, out, cap, &out_bits, &err)) != DEGA_OK)
  {
    LOG_TO(log, "glzmh: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_last_error(ctx));
    goto done;
  }
  if ((ret = first_error(&err, 1, log, "encoding")) != NO_ERROR)
    goto done;
  if (WriteBitFileBuffer(out_bit_buf, out, (size_t)out_bits) != (io_int_t)out_bits)
    ret = ERROR_LIBRARY_CALL;
done:
  free(raw.p);
  free(in);
  free(out);
  return ret;
}

io_int_t DecodeLZMHGPU(bit_file_buffer_t *const in_bit_buf, bit_file_buffer_t *const out_bit_buf, const options_t *const options)
{
  FILE *const log = options->error_log_file;
  dega_hip_ctx *ctx;
  byte_vec raw = { NULL, 0, 0 };
  uint64_t nbits = 0, out_len = 0;
  uint8_t *in = NULL, *out = NULL;
  int32_t err = 0;
  size_t cap, stride;
  int attempt;
  io_int_t ret;

  if ((ret = get_context(log, &ctx)) != NO_ERROR)
    return ret;
  if ((ret = slurp(in_bit_buf, &raw, &nbits)) != NO_ERROR)
    goto done;
  cap = (raw.n + 8) / 4 * 4;
  if ((in = (uint8_t *)calloc(cap, 1)) == NULL)
  {
    ret = ERROR_MEMORY;
    goto done;
  }
  memcpy(in, raw.p, raw.n);
  /* a bare LZMH stream does not say how long its text is: start from 4x and grow while the row overflows */
  stride = (4 * raw.n + 4096) / 8 * 8;
  for (attempt = 0;; attempt++)
  {
    free(out);
    if ((out = (uint8_t *)malloc(stride)) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    if ((ret = dega_hip_lzmh_decode_host(ctx, in, cap, &nbits, 1, out, stride, &out_len, &err)) != DEGA_OK)
    {
      LOG_TO(log, "glzmh: %s (%s)\n", ERROR_MESSAGE_STRING(ret), dega_hip_last_error(ctx));
      goto done;
    }
    if (err != ERROR_MEMORY || attempt == 5)
      break;
    stride *= 8;
  }
  if ((ret = first_error(&err, 1, log, "decoding")) != NO_ERROR)
    goto done;
  if (WriteBitFileBuffer(out_bit_buf, out, (size_t)out_len * 8) != (io_int_t)(out_len * 8))
    ret = ERROR_LIBRARY_CALL;
done:
  free(raw.p);
  free(in);
  free(out);
  return ret;
}
