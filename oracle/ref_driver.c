/* ref_driver.c -- in-memory driver over the REAL reference library, for oracle/_ref/libdcref.so.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with the reference's own DCIOLib and
 * DCLib sources where they lie under /root/reference (oracle/Makefile), against the reference's headers, and only
 * uses the reference's public API (DCLib/inc/enc_dec.h:43-71, DCIOLib/inc/bit_file_buffer.h:17-36,
 * DCIOLib/inc/file_buffer.h:22-44).  It runs a chain of stages on memory buffers the way DCCLI does
 * (DCCLI/src/cli.c:430-466: in-memory temp buffers, write->read mode switch between stages so that the exact
 * bit length is handed on), so tests can compare the restatement in dega_oracle.c with the reference itself and
 * bench.py can time the reference as the CPU baseline ("kind": "reference").
 */
#include "err_codes.h"
#include "enc_dec.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define DCREF_MAX_STAGES 16
#define DCREF_TEMP_SIZE (2 * 1024) /* same starting size as DCCLI's temp buffers (cli.c:33); they auto-grow */

typedef struct mem_stream
{
  file_buffer_t *fb;
  bit_file_buffer_t *bb;
} mem_stream_t;

static int stream_open(mem_stream_t *s)
{
  s->fb = AllocateFileBuffer();
  s->bb = AllocateBitFileBuffer();
  if (s->fb == NULL || s->bb == NULL)
    return ERROR_MEMORY;
  if (InitFileBufferInMemory(s->fb, FBM_WRITING, DCREF_TEMP_SIZE) != NO_ERROR)
    return ERROR_LIBRARY_INIT;
  InitBitFileBuffer(s->bb, s->fb);
  return NO_ERROR;
}

static void stream_close(mem_stream_t *s)
{
  if (s->fb != NULL)
  {
    UninitFileBuffer(s->fb);
    FreeFileBuffer(s->fb);
  }
  if (s->bb != NULL)
    FreeBitFileBuffer(s->bb);
  s->fb = NULL;
  s->bb = NULL;
}

/* One stage description: "encode <name> [opt[=value]]..." / "decode <name> ..." -- the grammar of one DCCLI stage
   (DCCLI/src/params.c:15).  Options are applied through the reference's typed setters. */
static int parse_stage(const char *spec, options_t *opt, char *errbuf, size_t errlen)
{
  char buf[256];
  char *tok, *save = NULL;
  const char *name;
  strncpy(buf, spec, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = '\0';
  if ((tok = strtok_r(buf, " ", &save)) == NULL)
    return ERROR_INVALID_VALUE;
  if (strcmp(tok, "encode") == 0)
    opt->encode = 1;
  else if (strcmp(tok, "decode") == 0)
    opt->encode = 0;
  else
    return ERROR_INVALID_VALUE;
  if ((tok = strtok_r(NULL, " ", &save)) == NULL)
    return ERROR_INVALID_VALUE;
  name = tok;
  if ((opt->encoder_decoder = GetEncoder(name)) == NULL)
  {
    snprintf(errbuf, errlen, "unknown encoder %s", name);
    return ERROR_INVALID_VALUE;
  }
  if ((opt->encode ? opt->encoder_decoder->encoder : opt->encoder_decoder->decoder) == NULL)
    return ERROR_INVALID_MODE;
  SetDefaultOptions(opt);
  while ((tok = strtok_r(NULL, " ", &save)) != NULL)
  {
    char *eq = strchr(tok, '=');
    const char *val = NULL;
    if (eq != NULL)
    {
      *eq = '\0';
      val = eq + 1;
    }
    if (!OptionNameExists(tok))
    {
      snprintf(errbuf, errlen, "unknown option %s", tok);
      return ERROR_INVALID_VALUE;
    }
    switch (GetOptionType(tok))
    {
      case OT_BOOL:
        SetOptionValueBool(opt, tok, val == NULL ? 1 : atoi(val));
        break;
      case OT_SIZE:
        if (val == NULL)
          return ERROR_INVALID_VALUE;
        SetOptionValueSize(opt, tok, (size_t)strtoull(val, NULL, 10));
        break;
      case OT_FLOAT:
        if (val == NULL)
          return ERROR_INVALID_VALUE;
        SetOptionValueFloat(opt, tok, strtof(val, NULL));
        break;
      case OT_CHAR:
        if (val == NULL)
          return ERROR_INVALID_VALUE;
        SetOptionValueChar(opt, tok, val[0]);
        break;
      default:
        return ERROR_INVALID_VALUE;
    }
  }
  return NO_ERROR;
}

/* Runs `n_stages` stages over the input bits.  in_nbits may end on a fractional byte.
   *out is malloc'ed (free with dcref_free); *out_nbits is the exact bit length of the last stage's output.
   stage_seconds (may be NULL) receives the per-stage CPU time measured with clock() as cli.c:445-455 does. */
int64_t dcref_run_chain(const uint8_t *in, uint64_t in_nbits, const char *const *stage_specs, size_t n_stages,
                        uint8_t **out, uint64_t *out_nbits, double *stage_seconds)
{
  mem_stream_t a = { NULL, NULL }, b = { NULL, NULL };
  mem_stream_t *rd = &a, *wr = &b;
  options_t opts[DCREF_MAX_STAGES];
  char errbuf[128];
  int64_t ret = NO_ERROR;
  size_t i;
  *out = NULL;
  *out_nbits = 0;
  if (n_stages == 0 || n_stages > DCREF_MAX_STAGES)
    return ERROR_INVALID_VALUE;
  for (i = 0; i < n_stages; i++)
  {
    memset(&opts[i], 0, sizeof(opts[i]));
    opts[i].error_log_file = NULL; /* LOG() skips NULL files (common/inc/log.h:10-13) */
    if ((ret = parse_stage(stage_specs[i], &opts[i], errbuf, sizeof(errbuf))) != NO_ERROR)
      return ret;
  }
  if ((ret = stream_open(rd)) != NO_ERROR || (ret = stream_open(wr)) != NO_ERROR)
    goto done;
  /* Load the input through the writing side, then switch to reading (keeps a fractional last byte). */
  if (in_nbits > 0 && WriteBitFileBuffer(rd->bb, in, (size_t)in_nbits) != (io_int_t)in_nbits)
  {
    ret = ERROR_LIBRARY_CALL;
    goto done;
  }
  if (SetBitFileBufferMode(rd->bb, FBM_READING) != NO_ERROR)
  {
    ret = ERROR_LIBRARY_CALL;
    goto done;
  }
  for (i = 0; i < n_stages; i++)
  {
    enc_dec_function_t *const fn = opts[i].encode ? opts[i].encoder_decoder->encoder : opts[i].encoder_decoder->decoder;
    mem_stream_t *tmp;
    const clock_t t0 = clock();
    ret = (*fn)(rd->bb, wr->bb, &opts[i]);
    if (stage_seconds != NULL)
      stage_seconds[i] = (double)(clock() - t0) / CLOCKS_PER_SEC;
    if (ret != NO_ERROR)
      goto done;
    /* SwitchTempBuffers (cli.c:212-223) */
    if (SetBitFileBufferMode(wr->bb, FBM_READING) != NO_ERROR)
    {
      ret = ERROR_LIBRARY_CALL;
      goto done;
    }
    tmp = rd;
    rd = wr;
    wr = tmp;
    ResetBitFileBuffer(wr->bb, FBM_WRITING);
  }
  /* Drain the last output exactly: whole bytes, then bit by bit. */
  {
    io_int_t size_bytes;
    uint8_t size_bits;
    size_t cap, got = 0;
    GetActualBitFileSize(rd->bb, &size_bytes, &size_bits);
    if (size_bytes < 0)
    {
      ret = size_bytes;
      goto done;
    }
    cap = (size_t)size_bytes + 2;
    if ((*out = (uint8_t *)calloc(cap, 1)) == NULL)
    {
      ret = ERROR_MEMORY;
      goto done;
    }
    while (!EndOfBitFileBuffer(rd->bb) && got / 8 < cap)
    {
      uint8_t byte = 0;
      const io_int_t n = ReadBitFileBuffer(rd->bb, &byte, 8);
      if (n <= 0)
        break;
      (*out)[got / 8] = byte; /* a short read is left-aligned by ReadBitFileBuffer (bit_file_buffer.c:203) */
      got += (size_t)n;
      if (n < 8)
        break;
    }
    *out_nbits = got;
  }
done:
  stream_close(&a);
  stream_close(&b);
  if (ret != NO_ERROR && *out != NULL)
  {
    free(*out);
    *out = NULL;
  }
  return ret;
}

void dcref_free(void *p)
{
  free(p);
}

/* DEGA encode of one int32 channel through the reference: "encode diff # encode seg # encode bac [adaptive]".
   seconds[0..2] = per-stage CPU time.  Returns the reference's error code. */
int64_t dcref_dega_encode_i32(const int32_t *x, size_t T, int adaptive, uint8_t *out, size_t out_cap, uint64_t *out_nbits, double *seconds)
{
  const char *stages[3] = { "encode diff", "encode seg", adaptive ? "encode bac adaptive" : "encode bac" };
  uint8_t *be = (uint8_t *)malloc(T * 4 + 1);
  uint8_t *res = NULL;
  int64_t ret;
  size_t t;
  if (be == NULL)
    return ERROR_MEMORY;
  for (t = 0; t < T; t++)
  {
    const uint32_t u = (uint32_t)x[t];
    be[4 * t + 0] = (uint8_t)(u >> 24);
    be[4 * t + 1] = (uint8_t)(u >> 16);
    be[4 * t + 2] = (uint8_t)(u >> 8);
    be[4 * t + 3] = (uint8_t)u;
  }
  ret = dcref_run_chain(be, (uint64_t)T * 32, stages, 3, &res, out_nbits, seconds);
  free(be);
  if (ret == NO_ERROR)
  {
    const size_t nbytes = (size_t)((*out_nbits + 7) / 8);
    if (nbytes > out_cap)
      ret = ERROR_MEMORY;
    else
      memcpy(out, res, nbytes);
  }
  free(res);
  return ret;
}

/* Inverse: "decode bac [adaptive] # decode seg # decode diff" -> native int32 samples. */
int64_t dcref_dega_decode_i32(const uint8_t *in, uint64_t in_nbits, int adaptive, int32_t *x, size_t max_T, size_t *out_T, double *seconds)
{
  const char *stages[3] = { adaptive ? "decode bac adaptive" : "decode bac", "decode seg", "decode diff" };
  uint8_t *res = NULL;
  uint64_t nbits = 0;
  int64_t ret = dcref_run_chain(in, in_nbits, stages, 3, &res, &nbits, seconds);
  if (ret == NO_ERROR)
  {
    const size_t T = (size_t)(nbits / 32);
    size_t t;
    if (T > max_T)
      ret = ERROR_MEMORY;
    else
    {
      for (t = 0; t < T; t++)
        x[t] = (int32_t)(((uint32_t)res[4 * t] << 24) | ((uint32_t)res[4 * t + 1] << 16) | ((uint32_t)res[4 * t + 2] << 8) | (uint32_t)res[4 * t + 3]);
      *out_T = T;
    }
  }
  free(res);
  return ret;
}
