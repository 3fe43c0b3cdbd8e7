/* dccli.c -- command line pipe-chain driver with the operator surface of the reference's DCCLI
 * (DCCLI/doc/readme.md:6-10, DCCLI/src/params.c:15, DCCLI/src/cli.c:407-472):
 *
 *     dccli_amd <input file> <output file> (encode|decode) <name> [<option>[=<value>]]... [# (encode|decode) <name> ...]...
 *
 * Stages run one after the other; the first reads the input file, the last writes the output file, everything in
 * between lives in memory and is handed on with its exact bit length (cli.c:430-466: temp buffers, write->read mode
 * switch).  Per stage the CPU time and the bytes/bits written are printed like DCCLI's diagnostics (cli.c:445-458).
 * On a failing stage the message of its error code is printed and the process exits with ERROR_LIBRARY_CALL, as
 * cli.c:447-453 does.  Only the codecs of src/enc_dec.c are registered: this driver exists to run the DEGA path. */
#include "enc_dec.h"
#include "err_codes.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAX_STAGES 16 /* MAX_OPTIONS, DCCLI/inc/params.h:10 */

static void usage(FILE *f)
{
  size_t i, n = GetNumberOfEncoders();
  const char **names = (const char **)malloc(n * sizeof(*names));
  fprintf(f, "Usage: <input file> <output file> ('encode'|'decode') <encoder/decoder> [<options>] [# ('encode'|'decode') <encoder/decoder> [<options>] ...]\n");
  if (names != NULL)
  {
    GetEncoderNames(names);
    for (i = 0; i < n; i++)
      fprintf(f, "  %s: %s\n", names[i], GetEncoderDescription(names[i]));
    free((void *)names);
  }
}

static int set_option(options_t *opt, const char *codec, const char *arg, FILE *log)
{
  char name[64];
  const char *eq = strchr(arg, '=');
  const size_t len = eq ? (size_t)(eq - arg) : strlen(arg);
  if (len == 0 || len >= sizeof(name))
    return ERROR_INVALID_FORMAT;
  memcpy(name, arg, len);
  name[len] = '\0';
  if (!OptionNameExists(name))
  {
    fprintf(log, "Unknown option '%s'\n", name);
    return ERROR_INVALID_VALUE;
  }
  if (!EncoderSupportsOption(codec, name))
  {
    fprintf(log, "Option '%s' is not supported by '%s'\n", name, codec);
    return ERROR_INVALID_VALUE;
  }
  switch (GetOptionType(name))
  {
    case OT_BOOL:
      return SetOptionValueBool(opt, name, eq ? atoi(eq + 1) : 1); /* bare name switches it on (params.c:126-170) */
    case OT_SIZE:
    {
      int restricted = 0;
      size_t lo = 0, hi = 0, v;
      char *end = NULL;
      if (eq == NULL || eq[1] == '\0')
      {
        fprintf(log, "Expected '=' and a value after option '%s'\n", name);
        return ERROR_INVALID_FORMAT;
      }
      v = (size_t)strtoull(eq + 1, &end, 10);
      if (end == NULL || *end != '\0')
        return ERROR_INVALID_VALUE;
      GetAllowedOptionValueRange(name, &restricted, &lo, &hi);
      if (restricted && (v < lo || v > hi))
      {
        fprintf(log, "Value of option '%s' out of range\n", name);
        return ERROR_INVALID_VALUE;
      }
      return SetOptionValueSize(opt, name, v);
    }
    case OT_FLOAT:
      if (eq == NULL || eq[1] == '\0')
        return ERROR_INVALID_FORMAT;
      return SetOptionValueFloat(opt, name, strtof(eq + 1, NULL));
    case OT_CHAR:
      if (eq == NULL || eq[1] == '\0')
        return ERROR_INVALID_FORMAT;
      return SetOptionValueChar(opt, name, eq[1]);
    default:
      return ERROR_INVALID_VALUE;
  }
}

typedef struct stream_t
{
  file_buffer_t *fb;
  bit_file_buffer_t *bb;
} stream_t;

static int open_stream(stream_t *s, FILE *file, file_buffer_mode_t mode)
{
  int ret;
  s->fb = AllocateFileBuffer();
  s->bb = AllocateBitFileBuffer();
  if (s->fb == NULL || s->bb == NULL)
    return ERROR_MEMORY;
  ret = file ? InitFileBuffer(s->fb, file, mode, 1024) : InitFileBufferInMemory(s->fb, mode, 2048);
  if (ret != NO_ERROR)
    return ERROR_LIBRARY_INIT;
  InitBitFileBuffer(s->bb, s->fb);
  return NO_ERROR;
}

static void close_stream(stream_t *s)
{
  if (s->bb != NULL && s->fb != NULL)
    UninitBitFileBuffer(s->bb); /* a writing file stream is flushed here, last byte zero padded */
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

int main(int argc, char **argv)
{
  options_t stages[MAX_STAGES];
  const char *stage_names[MAX_STAGES];
  size_t n_stages = 0, i;
  FILE *in_file, *out_file;
  stream_t in = { NULL, NULL }, out = { NULL, NULL }, tmp_a = { NULL, NULL }, tmp_b = { NULL, NULL };
  stream_t *rd_tmp = &tmp_a, *wr_tmp = &tmp_b;
  double total = 0.0;
  int a, ret;

  if (argc < 5)
  {
    usage(stderr);
    return ERROR_INVALID_FORMAT;
  }
  for (a = 3; a < argc;)
  {
    options_t *opt;
    if (n_stages == MAX_STAGES)
    {
      fprintf(stderr, "Too many encoders/decoders (max. %d)\n", MAX_STAGES);
      return ERROR_INVALID_VALUE;
    }
    opt = &stages[n_stages];
    memset(opt, 0, sizeof(*opt));
    opt->error_log_file = stderr;
    if (strcmp(argv[a], "encode") == 0)
      opt->encode = 1;
    else if (strcmp(argv[a], "decode") == 0)
      opt->encode = 0;
    else
    {
      fprintf(stderr, "Expected 'encode' or 'decode' instead of '%s'\n", argv[a]);
      usage(stderr);
      return ERROR_INVALID_FORMAT;
    }
    if (++a >= argc)
    {
      usage(stderr);
      return ERROR_INVALID_FORMAT;
    }
    if ((opt->encoder_decoder = GetEncoder(argv[a])) == NULL)
    {
      fprintf(stderr, "Unknown encoder/decoder '%s'\n", argv[a]);
      usage(stderr);
      return ERROR_INVALID_VALUE;
    }
    if ((opt->encode ? opt->encoder_decoder->encoder : opt->encoder_decoder->decoder) == NULL)
    {
      fprintf(stderr, "'%s' cannot %s\n", argv[a], opt->encode ? "encode" : "decode");
      return ERROR_INVALID_MODE; /* params.c:241-246 */
    }
    stage_names[n_stages] = argv[a];
    SetDefaultOptions(opt);
    for (++a; a < argc && strcmp(argv[a], "#") != 0; a++)
      if ((ret = set_option(opt, stage_names[n_stages], argv[a], stderr)) != NO_ERROR)
        return ret;
    if (a < argc)
      a++; /* skip '#' */
    n_stages++;
  }
  if ((in_file = fopen(argv[1], "rb")) == NULL)
  {
    fprintf(stderr, "Error opening input file '%s'\n", argv[1]);
    return ERROR_FILE_IO;
  }
  if ((out_file = fopen(argv[2], "wb")) == NULL)
  {
    fprintf(stderr, "Error opening output file '%s'\n", argv[2]);
    fclose(in_file);
    return ERROR_FILE_IO;
  }
  if ((ret = open_stream(&in, in_file, FBM_READING)) != NO_ERROR || (ret = open_stream(&out, out_file, FBM_WRITING)) != NO_ERROR ||
      (n_stages > 1 && ((ret = open_stream(&tmp_a, NULL, FBM_READING)) != NO_ERROR || (ret = open_stream(&tmp_b, NULL, FBM_WRITING)) != NO_ERROR)))
  {
    fprintf(stderr, "%s while initializing buffers\n", ERROR_MESSAGE_STRING(ret));
    return ret;
  }
  ret = 0;
  for (i = 0; i < n_stages; i++)
  {
    options_t *const opt = &stages[i];
    bit_file_buffer_t *const rd = i == 0 ? in.bb : rd_tmp->bb;
    bit_file_buffer_t *const wr = i == n_stages - 1 ? out.bb : wr_tmp->bb;
    enc_dec_function_t *const fn = opt->encode ? opt->encoder_decoder->encoder : opt->encoder_decoder->decoder;
    io_int_t code, bytes;
    uint8_t bits;
    clock_t t0;
    double secs;
    printf("Executing %s %s (%lu of %lu total)...\n", opt->encode ? "encoder" : "decoder", stage_names[i], (unsigned long)(i + 1), (unsigned long)n_stages);
    t0 = clock();
    code = (*fn)(rd, wr, opt);
    secs = (double)(clock() - t0) / CLOCKS_PER_SEC;
    if (code != NO_ERROR)
    {
      fprintf(stderr, "\033[0;31m%s while executing encoder/decoder %lu of %lu\033[0m\n", ERROR_MESSAGE_STRING(code), (unsigned long)(i + 1), (unsigned long)n_stages);
      ret = ERROR_LIBRARY_CALL;
      break;
    }
    total += secs;
    printf("  Time elapsed: %.3f ms (%.3f s)\n", secs * 1e3, secs);
    GetActualBitFileOffset(wr, &bytes, &bits);
    printf("  Wrote %" PRId64 " bytes and %u bits\n", (int64_t)bytes, (unsigned)bits);
    if (i + 1 < n_stages) /* hand the memory stream on: exact bit length survives the mode switch */
    {
      stream_t *t;
      SetBitFileBufferMode(wr_tmp->bb, FBM_READING);
      t = rd_tmp;
      rd_tmp = wr_tmp;
      wr_tmp = t;
      ResetBitFileBuffer(wr_tmp->bb, FBM_WRITING);
    }
  }
  if (ret == 0)
    printf("Total time elapsed: %.3f ms (%.3f s)\n", total * 1e3, total);
  close_stream(&tmp_a);
  close_stream(&tmp_b);
  close_stream(&out);
  close_stream(&in);
  fclose(out_file);
  fclose(in_file);
  return ret;
}
