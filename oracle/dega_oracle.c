/* dega_oracle.c -- CPU restatement of the reference's DEGA hot path (normalize -> diff -> seg -> bac and inverse).
 *
 * TEST INFRASTRUCTURE ONLY (see dega_oracle.h).  Plain C, one in-memory bit stream per stage, no DCIOLib.
 * Each function cites the reference lines it restates; paths are relative to /root/reference/DataCompressor/.
 */
#include "dega_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------------
 * Bit streams.  Format definition: DCIOLib/src/bit_file_buffer.c:220-248 (bits fill a byte from its MSB),
 * :297-308 (an n-bit value is emitted most significant bit first), :310-320 (final partial byte zero padded).
 * ------------------------------------------------------------------------------------------------------------------ */

void orc_bits_init(orc_bits_t *b)
{
  b->data = NULL;
  b->nbits = 0;
  b->cap_bytes = 0;
}

void orc_bits_free(orc_bits_t *b)
{
  free(b->data);
  orc_bits_init(b);
}

void orc_bits_clear(orc_bits_t *b)
{
  if (b->data != NULL && b->cap_bytes > 0)
    memset(b->data, 0, b->cap_bytes);
  b->nbits = 0;
}

static int bits_reserve(orc_bits_t *b, size_t total_bits)
{
  const size_t need = (total_bits + 7) / 8 + 8;
  if (need > b->cap_bytes)
  {
    size_t ncap = b->cap_bytes == 0 ? 4096 : b->cap_bytes;
    uint8_t *nd;
    while (ncap < need)
      ncap *= 2;
    if ((nd = (uint8_t *)realloc(b->data, ncap)) == NULL)
      return ORC_ERROR_MEMORY;
    memset(nd + b->cap_bytes, 0, ncap - b->cap_bytes);
    b->data = nd;
    b->cap_bytes = ncap;
  }
  return ORC_NO_ERROR;
}

int orc_bits_assign(orc_bits_t *b, const uint8_t *bytes, size_t nbits)
{
  int ret;
  orc_bits_clear(b);
  if ((ret = bits_reserve(b, nbits)) != ORC_NO_ERROR)
    return ret;
  if (nbits > 0)
  {
    memcpy(b->data, bytes, (nbits + 7) / 8);
    if (nbits % 8 != 0) /* keep the padding bits zero */
      b->data[nbits / 8] &= (uint8_t)(0xFF00u >> (nbits % 8));
  }
  b->nbits = nbits;
  return ORC_NO_ERROR;
}

size_t orc_bits_nbytes(const orc_bits_t *b)
{
  return b->nbits == 0 ? 1 : (b->nbits + 7) / 8;
}

/* Append the low k bits of v, most significant first (WriteSingleValueToBitFileBuffer, bit_file_buffer.c:297-308). */
static int put_bits(orc_bits_t *b, uint64_t v, unsigned k)
{
  int ret;
  if (k == 0)
    return ORC_NO_ERROR;
  if (k > 64)
    return ORC_ERROR_INVALID_VALUE; /* bit_file_buffer.c:302-303 */
  if ((ret = bits_reserve(b, b->nbits + k)) != ORC_NO_ERROR)
    return ret;
  if (k < 64)
    v &= (((uint64_t)1 << k) - 1);
  while (k > 0)
  {
    const unsigned used = (unsigned)(b->nbits & 7);
    const unsigned room = 8 - used;
    const unsigned take = k < room ? k : room;
    const uint8_t chunk = (uint8_t)((v >> (k - take)) & ((1u << take) - 1));
    b->data[b->nbits >> 3] |= (uint8_t)(chunk << (room - take));
    b->nbits += take;
    k -= take;
  }
  return ORC_NO_ERROR;
}

typedef struct bit_reader
{
  const uint8_t *d;
  size_t n;   /* total bits */
  size_t pos; /* next bit */
} bit_reader_t;

static void rd_init(bit_reader_t *r, const orc_bits_t *b)
{
  r->d = b->data;
  r->n = b->nbits;
  r->pos = 0;
}

static int rd_eof(const bit_reader_t *r) /* EndOfBitFileBuffer, bit_file_buffer.c:53-62 */
{
  return r->pos >= r->n;
}

/* ReadSingleValueFromBitFileBuffer (bit_file_buffer.c:280-294) behind READ_VALUE_BITS_CHECKED (io_macros.h:61-66):
   a short read is reported as ERROR_LIBRARY_CALL by the macro (io_macros.h:13-27). */
static int get_bits(bit_reader_t *r, unsigned k, uint64_t *v)
{
  uint64_t acc = 0;
  if (k > 64)
    return ORC_ERROR_INVALID_VALUE;
  if (r->n - r->pos < k)
  {
    r->pos = r->n;
    return ORC_ERROR_LIBRARY_CALL;
  }
  while (k > 0)
  {
    const unsigned used = (unsigned)(r->pos & 7);
    const unsigned room = 8 - used;
    const unsigned take = k < room ? k : room;
    const uint8_t byte = r->d[r->pos >> 3];
    acc = (acc << take) | ((byte >> (room - take)) & ((1u << take) - 1));
    r->pos += take;
    k -= take;
  }
  *v = acc;
  return ORC_NO_ERROR;
}

static int64_t sign_extend(uint64_t v, unsigned bits) /* EXTEND_IO_INT_SIGN, io_macros.h:89-90 */
{
  if (bits == 64)
    return (int64_t)v;
  return (int64_t)(v << (64 - bits)) >> (64 - bits);
}

/* ------------------------------------------------------------------------------------------------------------------
 * normalize (DCLib/src/normalize.c)
 * ------------------------------------------------------------------------------------------------------------------ */

/* normalize.c:16-24 for one sample.  The multiply and the +-0.5 are two separately rounded float operations (the
   reference build has no FMA: SURVEY.md Appendix A.1); volatile keeps this compiler from contracting them. */
static int normalize_one(float value, float factor, unsigned valuesize, int64_t *out)
{
  volatile float prod;
  if (value > 0)
  {
    prod = value * factor;
    value = prod + 0.5f;
  }
  else if (value < 0)
  {
    prod = value * factor;
    value = prod - 0.5f;
  }
  if (value < (-(float)((uint64_t)1 << (valuesize - 1))) || value > (float)(((uint64_t)1 << (valuesize - 1)) - 1)) /* :21 */
    return ORC_ERROR_INVALID_VALUE;
  *out = (int64_t)value; /* :23 truncation toward zero */
  return ORC_NO_ERROR;
}

int orc_normalize_encode(const orc_bits_t *in, orc_bits_t *out, float factor, unsigned valuesize)
{
  bit_reader_t r;
  rd_init(&r, in);
  while (!rd_eof(&r)) /* normalize.c:11 */
  {
    uint8_t raw[4];
    float value;
    int64_t n;
    int ret, i;
    for (i = 0; i < 4; i++) /* READ_BITS_CHECKED of 32 bits: bytes in stream order = memory image (normalize.c:16) */
    {
      uint64_t byte;
      if ((ret = get_bits(&r, 8, &byte)) != ORC_NO_ERROR)
        return ret;
      raw[i] = (uint8_t)byte;
    }
    memcpy(&value, raw, 4);
    if ((ret = normalize_one(value, factor, valuesize, &n)) != ORC_NO_ERROR)
      return ret;
    if ((ret = put_bits(out, (uint64_t)n, valuesize)) != ORC_NO_ERROR) /* normalize.c:24 */
      return ret;
  }
  return ORC_NO_ERROR;
}

int orc_normalize_decode(const orc_bits_t *in, orc_bits_t *out, float factor, unsigned valuesize)
{
  bit_reader_t r;
  rd_init(&r, in);
  while (!rd_eof(&r)) /* normalize.c:31 */
  {
    uint64_t u;
    uint8_t raw[4];
    float value;
    int ret, i;
    if ((ret = get_bits(&r, valuesize, &u)) != ORC_NO_ERROR) /* :36 */
      return ret;
    value = (float)sign_extend(u, valuesize) / factor; /* :37-38, true division */
    memcpy(raw, &value, 4);
    for (i = 0; i < 4; i++) /* :39 raw native-endian float */
      if ((ret = put_bits(out, raw[i], 8)) != ORC_NO_ERROR)
        return ret;
  }
  return ORC_NO_ERROR;
}

/* ------------------------------------------------------------------------------------------------------------------
 * diff (DCLib/src/diff.c)
 * ------------------------------------------------------------------------------------------------------------------ */

int orc_diff_encode(const orc_bits_t *in, orc_bits_t *out, unsigned valuesize)
{
  bit_reader_t r;
  int64_t last = 0; /* diff.c:11 */
  rd_init(&r, in);
  while (!rd_eof(&r))
  {
    uint64_t u;
    int64_t value, d;
    int ret;
    if ((ret = get_bits(&r, valuesize, &u)) != ORC_NO_ERROR) /* diff.c:15 -- value is NOT sign extended */
      return ret;
    value = (int64_t)u;
    d = (int64_t)((uint64_t)value - (uint64_t)last); /* :16 */
    if (valuesize != 64 && (d < -(int64_t)((uint64_t)1 << (valuesize - 1)) || d > (((int64_t)1 << (valuesize - 1)) - 1))) /* :17 */
      return ORC_ERROR_INVALID_VALUE;
    if ((ret = put_bits(out, (uint64_t)d, valuesize)) != ORC_NO_ERROR) /* :19 */
      return ret;
    last = value; /* :20 */
  }
  return ORC_NO_ERROR;
}

int orc_diff_decode(const orc_bits_t *in, orc_bits_t *out, unsigned valuesize)
{
  bit_reader_t r;
  int64_t last = 0; /* diff.c:27 */
  rd_init(&r, in);
  while (!rd_eof(&r))
  {
    uint64_t u;
    int64_t value;
    int ret;
    if ((ret = get_bits(&r, valuesize, &u)) != ORC_NO_ERROR) /* :31 */
      return ret;
    value = (int64_t)((uint64_t)sign_extend(u, valuesize) + (uint64_t)last); /* :32-33 */
    if ((ret = put_bits(out, (uint64_t)value, valuesize)) != ORC_NO_ERROR) /* :34 */
      return ret;
    last = value; /* :35 */
  }
  return ORC_NO_ERROR;
}

/* ------------------------------------------------------------------------------------------------------------------
 * seg (DCLib/src/seg.c)
 * ------------------------------------------------------------------------------------------------------------------ */

static int seg_put_codeword(orc_bits_t *out, int64_t value)
{
  /* seg.c:23-29: positive -> odd code numbers, zero/negative -> even */
  const uint64_t mag = value < 0 ? (uint64_t)0 - (uint64_t)value : (uint64_t)value;
  const uint64_t code_number = value > 0 ? 2 * mag - 1 : 2 * mag;
  /* seg.c:11-21: order-0 exp-Golomb of code_number */
  const uint64_t w = code_number + 1;
  uint64_t t = w;
  unsigned prefix = 0;
  int ret;
  while ((t >>= 1) != 0) /* :16-17 */
    prefix++;
  if ((ret = put_bits(out, 0, prefix)) != ORC_NO_ERROR) /* :18 */
    return ret;
  return put_bits(out, w, 1 + prefix); /* :19 */
}

int orc_seg_encode(const orc_bits_t *in, orc_bits_t *out, unsigned valuesize)
{
  bit_reader_t r;
  rd_init(&r, in);
  while (!rd_eof(&r)) /* seg.c:34 */
  {
    uint64_t u;
    int ret;
    if ((ret = get_bits(&r, valuesize, &u)) != ORC_NO_ERROR) /* :37 */
      return ret;
    if ((ret = seg_put_codeword(out, sign_extend(u, valuesize))) != ORC_NO_ERROR) /* :38-39 */
      return ret;
  }
  return ORC_NO_ERROR;
}

/* seg.c:45-68.  *eos is set when the stream ends inside a non-empty zero prefix (how byte padding is swallowed). */
static int seg_get_code_number(bit_reader_t *r, unsigned max_prefix, uint64_t *code_number, int *eos)
{
  unsigned prefix = 0;
  uint64_t bit = 0, rest = 0;
  int ret;
  *eos = 0;
  while (bit == 0 && !rd_eof(r)) /* :50 */
  {
    if ((ret = get_bits(r, 1, &bit)) != ORC_NO_ERROR)
      return ret;
    if (bit == 0)
      prefix++;
    if (prefix >= max_prefix) /* :55-56 */
      return ORC_ERROR_INVALID_FORMAT;
  }
  if (rd_eof(r) && prefix != 0) /* :58-62 */
  {
    *eos = 1;
    return ORC_NO_ERROR;
  }
  if (prefix > 0 && (ret = get_bits(r, prefix, &rest)) != ORC_NO_ERROR) /* :64 */
    return ret;
  *code_number = (rest | ((uint64_t)1 << prefix)) - 1; /* :65-66 */
  return ORC_NO_ERROR;
}

int orc_seg_decode(const orc_bits_t *in, orc_bits_t *out, unsigned valuesize)
{
  bit_reader_t r;
  const unsigned max_prefix = valuesize + 1 > 64 ? 64 : valuesize + 1; /* seg.c:74 */
  rd_init(&r, in);
  while (!rd_eof(&r)) /* :84 */
  {
    uint64_t code_number;
    int64_t value;
    int eos, ret;
    if ((ret = seg_get_code_number(&r, max_prefix, &code_number, &eos)) != ORC_NO_ERROR)
      return ret;
    if (eos) /* :90-91 */
      break;
    value = (int64_t)((code_number + 1) / 2); /* :76 */
    if ((code_number & 1) == 0)              /* :77-78 */
      value = -value;
    if ((ret = put_bits(out, (uint64_t)value, valuesize)) != ORC_NO_ERROR) /* :92 */
      return ret;
  }
  return ORC_NO_ERROR;
}

/* ------------------------------------------------------------------------------------------------------------------
 * bac (DCLib/src/bac.c) -- Witten/Neal/Cleary coder, 16-bit range, symbols {0,1} + EOF
 * ------------------------------------------------------------------------------------------------------------------ */

#define BAC_MAX_RANGE 0xFFFFu          /* bac.c:22 */
#define BAC_QUARTER 0x4000u            /* :23 */
#define BAC_HALF 0x8000u               /* :24 */
#define BAC_THREE_QUARTERS 0xC000u     /* :25 */
#define BAC_MAX_FREQUENCY 0x3FFFu      /* :27 */
#define BAC_EOF_INDEX 3                /* :30 */

typedef struct bac_model
{
  unsigned sym2idx[2]; /* bac.c:33 */
  int idx2sym[4];      /* :34 */
  uint16_t freq[4];    /* :36 */
  uint16_t cum[4];     /* :37 */
} bac_model_t;

static void bac_model_init(bac_model_t *m) /* bac.c:39-52 */
{
  unsigned i;
  for (i = 0; i < 2; i++)
  {
    m->sym2idx[i] = i + 1;
    m->idx2sym[i + 1] = (int)i;
  }
  m->idx2sym[0] = 0;
  for (i = 0; i <= 3; i++)
  {
    m->freq[i] = i == 0 ? 0 : 1;
    m->cum[i] = (uint16_t)(3 - i);
  }
}

static void bac_model_update(bac_model_t *m, unsigned last) /* bac.c:54-81 */
{
  unsigned i;
  if (m->cum[0] == BAC_MAX_FREQUENCY) /* :57-67 */
  {
    uint16_t c = 0;
    i = 4;
    while (i-- != 0)
    {
      m->freq[i] = (uint16_t)((m->freq[i] + 1) / 2);
      m->cum[i] = c;
      c = (uint16_t)(c + m->freq[i]);
    }
  }
  for (i = last; m->freq[i] == m->freq[i - 1]; i--) /* :68 */
    ;
  if (i < last) /* :69-77 */
  {
    const int cur_sym = m->idx2sym[i];
    const int last_sym = m->idx2sym[last];
    m->idx2sym[i] = last_sym;
    m->idx2sym[last] = cur_sym;
    m->sym2idx[cur_sym] = last;
    m->sym2idx[last_sym] = i;
  }
  m->freq[i]++; /* :78 */
  while (i-- > 0) /* :79-80 */
    m->cum[i]++;
}

typedef struct bac_encoder
{
  uint16_t start, end; /* bac.c:83 */
  size_t pending;      /* :84 next_bits */
  orc_bits_t *out;
} bac_encoder_t;

static int bac_emit(bac_encoder_t *e, int bit) /* OutputNextBits, bac.c:93-105 */
{
  int ret;
  if ((ret = put_bits(e->out, bit ? 1 : 0, 1)) != ORC_NO_ERROR)
    return ret;
  while (e->pending != 0)
  {
    const unsigned k = e->pending > 64 ? 64 : (unsigned)e->pending;
    if ((ret = put_bits(e->out, bit ? 0 : ~(uint64_t)0, k)) != ORC_NO_ERROR)
      return ret;
    e->pending -= k;
  }
  return ORC_NO_ERROR;
}

static int bac_encode_symbol(bac_encoder_t *e, const bac_model_t *m, unsigned idx) /* bac.c:107-139 */
{
  const uint64_t range = (uint64_t)(e->end - e->start) + 1;                           /* :109 */
  e->end = (uint16_t)(e->start + (uint16_t)((range * m->cum[idx - 1]) / m->cum[0]) - 1); /* :110 */
  e->start = (uint16_t)(e->start + (uint16_t)((range * m->cum[idx]) / m->cum[0]));       /* :111 */
  for (;;)
  {
    int ret;
    if (e->end < BAC_HALF) /* :115-119 */
    {
      if ((ret = bac_emit(e, 0)) != ORC_NO_ERROR)
        return ret;
    }
    else if (e->start >= BAC_HALF) /* :120-126 */
    {
      if ((ret = bac_emit(e, 1)) != ORC_NO_ERROR)
        return ret;
      e->start = (uint16_t)(e->start - BAC_HALF);
      e->end = (uint16_t)(e->end - BAC_HALF);
    }
    else if (e->start >= BAC_QUARTER && e->end < BAC_THREE_QUARTERS) /* :127-132 */
    {
      e->pending++;
      e->start = (uint16_t)(e->start - BAC_QUARTER);
      e->end = (uint16_t)(e->end - BAC_QUARTER);
    }
    else
      break;
    e->start = (uint16_t)(e->start * 2);   /* :135 */
    e->end = (uint16_t)(2 * e->end + 1);   /* :136 */
  }
  return ORC_NO_ERROR;
}

int orc_bac_encode(const orc_bits_t *in, orc_bits_t *out, int adaptive)
{
  bac_model_t m;
  bac_encoder_t e;
  bit_reader_t r;
  int ret;
  bac_model_init(&m); /* bac.c:150 */
  e.start = 0;        /* :86-91 */
  e.end = BAC_MAX_RANGE;
  e.pending = 0;
  e.out = out;
  rd_init(&r, in);
  while (!rd_eof(&r)) /* :152 */
  {
    uint64_t bit;
    unsigned idx;
    if ((ret = get_bits(&r, 1, &bit)) != ORC_NO_ERROR) /* :156 */
      return ret;
    idx = m.sym2idx[bit];                               /* :157 */
    if ((ret = bac_encode_symbol(&e, &m, idx)) != ORC_NO_ERROR)
      return ret;
    if (adaptive) /* :160-161 */
      bac_model_update(&m, idx);
  }
  if ((ret = bac_encode_symbol(&e, &m, BAC_EOF_INDEX)) != ORC_NO_ERROR) /* :163 */
    return ret;
  e.pending++;                                                /* :143 */
  return bac_emit(&e, e.start < BAC_QUARTER ? 0 : 1);         /* :144 */
}

typedef struct bac_decoder
{
  uint16_t start, end, value; /* bac.c:83,168 */
  unsigned after_eof;         /* :169 */
  bit_reader_t r;
} bac_decoder_t;

static int bac_read_bit(bac_decoder_t *d, unsigned *bit) /* ReadBitSpecial, bac.c:171-186 */
{
  uint64_t b;
  int ret;
  if (rd_eof(&d->r))
  {
    if (d->after_eof > 0)
    {
      d->after_eof--;
      *bit = 0;
      return ORC_NO_ERROR;
    }
    return ORC_ERROR_INVALID_FORMAT;
  }
  if ((ret = get_bits(&d->r, 1, &b)) != ORC_NO_ERROR)
    return ret;
  *bit = (unsigned)b;
  return ORC_NO_ERROR;
}

static int bac_decode_symbol(bac_decoder_t *d, const bac_model_t *m, unsigned *idx_out) /* bac.c:206-242 */
{
  const uint64_t range = (uint64_t)(d->end - d->start) + 1;
  const uint16_t cf = (uint16_t)((((uint64_t)(d->value - d->start) + 1) * m->cum[0] - 1) / range); /* :209 */
  unsigned idx;
  for (idx = 1; m->cum[idx] > cf; idx++) /* :210 */
    ;
  d->end = (uint16_t)(d->start + (uint16_t)((range * m->cum[idx - 1]) / m->cum[0]) - 1); /* :211 */
  d->start = (uint16_t)(d->start + (uint16_t)((range * m->cum[idx]) / m->cum[0]));       /* :212 */
  for (;;)
  {
    unsigned bit;
    int ret;
    if (d->end < BAC_HALF)
    {
    }
    else if (d->start >= BAC_HALF)
    {
      d->value = (uint16_t)(d->value - BAC_HALF);
      d->start = (uint16_t)(d->start - BAC_HALF);
      d->end = (uint16_t)(d->end - BAC_HALF);
    }
    else if (d->start >= BAC_QUARTER && d->end < BAC_THREE_QUARTERS)
    {
      d->value = (uint16_t)(d->value - BAC_QUARTER);
      d->start = (uint16_t)(d->start - BAC_QUARTER);
      d->end = (uint16_t)(d->end - BAC_QUARTER);
    }
    else
      break;
    d->start = (uint16_t)(d->start * 2);
    d->end = (uint16_t)(2 * d->end + 1);
    if ((ret = bac_read_bit(d, &bit)) != ORC_NO_ERROR) /* :237-238 */
      return ret;
    d->value = (uint16_t)(2 * d->value + bit);
  }
  *idx_out = idx;
  return ORC_NO_ERROR;
}

int orc_bac_decode(const orc_bits_t *in, orc_bits_t *out, int adaptive)
{
  bac_model_t m;
  bac_decoder_t d;
  unsigned i;
  int ret;
  bac_model_init(&m); /* bac.c:247 */
  rd_init(&d.r, in);
  d.value = 0; /* StartDecoding, :188-204 */
  d.after_eof = 16 - 2;
  for (i = 0; i < 16; i++)
  {
    unsigned bit;
    if ((ret = bac_read_bit(&d, &bit)) != ORC_NO_ERROR)
      return ret;
    d.value = (uint16_t)(2 * d.value + bit);
  }
  d.start = 0;
  d.end = BAC_MAX_RANGE;
  for (;;) /* :250-262 */
  {
    unsigned idx;
    if ((ret = bac_decode_symbol(&d, &m, &idx)) != ORC_NO_ERROR)
      return ret;
    if (idx == BAC_EOF_INDEX)
      break;
    if ((ret = put_bits(out, (uint64_t)m.idx2sym[idx], 1)) != ORC_NO_ERROR)
      return ret;
    if (adaptive)
      bac_model_update(&m, idx);
  }
  return ORC_NO_ERROR;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Whole-chain helpers (what DCCLI's stage loop does with in-memory temp buffers: cli.c:430-466)
 * ------------------------------------------------------------------------------------------------------------------ */

size_t orc_dega_worst_case_bytes(size_t T)
{
  /* seg: at most 65 bits per 32-bit sample (seg.c:18-19); the non-adaptive coder can expand a bit by log2(3) bits,
     so allow 2 output bits per seg bit plus the EOF/flush tail. */
  return (T * 65 * 2 + 64) / 8 + 16;
}

static int chain_encode_tail(orc_bits_t *values, int adaptive, uint8_t *out, size_t out_cap, uint64_t *out_nbits)
{
  orc_bits_t a, b;
  int ret;
  orc_bits_init(&a);
  orc_bits_init(&b);
  if ((ret = orc_diff_encode(values, &a, 32)) == ORC_NO_ERROR && (ret = orc_seg_encode(&a, &b, 32)) == ORC_NO_ERROR)
  {
    orc_bits_clear(&a);
    if ((ret = orc_bac_encode(&b, &a, adaptive)) == ORC_NO_ERROR)
    {
      if ((a.nbits + 7) / 8 > out_cap)
        ret = ORC_ERROR_MEMORY;
      else
      {
        memcpy(out, a.data, (a.nbits + 7) / 8);
        *out_nbits = a.nbits;
      }
    }
  }
  orc_bits_free(&a);
  orc_bits_free(&b);
  return ret;
}

int orc_dega_encode_i32(const int32_t *x, size_t T, int adaptive, uint8_t *out, size_t out_cap, uint64_t *out_nbits)
{
  orc_bits_t v;
  size_t t;
  int ret = ORC_NO_ERROR;
  orc_bits_init(&v);
  for (t = 0; t < T && ret == ORC_NO_ERROR; t++)
    ret = put_bits(&v, (uint32_t)x[t], 32);
  if (ret == ORC_NO_ERROR)
    ret = chain_encode_tail(&v, adaptive, out, out_cap, out_nbits);
  orc_bits_free(&v);
  return ret;
}

static int chain_decode_head(const uint8_t *in, uint64_t in_nbits, int adaptive, orc_bits_t *values)
{
  orc_bits_t a, b;
  int ret;
  orc_bits_init(&a);
  orc_bits_init(&b);
  if ((ret = orc_bits_assign(&a, in, (size_t)in_nbits)) == ORC_NO_ERROR && (ret = orc_bac_decode(&a, &b, adaptive)) == ORC_NO_ERROR)
  {
    orc_bits_clear(&a);
    if ((ret = orc_seg_decode(&b, &a, 32)) == ORC_NO_ERROR)
      ret = orc_diff_decode(&a, values, 32);
  }
  orc_bits_free(&a);
  orc_bits_free(&b);
  return ret;
}

int orc_dega_decode_i32(const uint8_t *in, uint64_t in_nbits, int adaptive, int32_t *x, size_t max_T, size_t *out_T)
{
  orc_bits_t v;
  int ret;
  orc_bits_init(&v);
  if ((ret = chain_decode_head(in, in_nbits, adaptive, &v)) == ORC_NO_ERROR)
  {
    bit_reader_t r;
    size_t t = 0;
    rd_init(&r, &v);
    while (!rd_eof(&r) && t < max_T)
    {
      uint64_t u = 0;
      if ((ret = get_bits(&r, 32, &u)) != ORC_NO_ERROR)
        break;
      x[t++] = (int32_t)(uint32_t)u;
    }
    if (ret == ORC_NO_ERROR && !rd_eof(&r))
      ret = ORC_ERROR_MEMORY; /* more samples than the caller has room for */
    *out_T = t;
  }
  orc_bits_free(&v);
  return ret;
}

int orc_dega_encode_f32(const float *v, size_t T, float factor, int adaptive, uint8_t *out, size_t out_cap, uint64_t *out_nbits)
{
  orc_bits_t raw, norm;
  int ret;
  orc_bits_init(&raw);
  orc_bits_init(&norm);
  if ((ret = orc_bits_assign(&raw, (const uint8_t *)v, T * 32)) == ORC_NO_ERROR && (ret = orc_normalize_encode(&raw, &norm, factor, 32)) == ORC_NO_ERROR)
    ret = chain_encode_tail(&norm, adaptive, out, out_cap, out_nbits);
  orc_bits_free(&raw);
  orc_bits_free(&norm);
  return ret;
}

int orc_dega_decode_f32(const uint8_t *in, uint64_t in_nbits, float factor, int adaptive, float *v, size_t max_T, size_t *out_T)
{
  orc_bits_t ints, raw;
  int ret;
  orc_bits_init(&ints);
  orc_bits_init(&raw);
  if ((ret = chain_decode_head(in, in_nbits, adaptive, &ints)) == ORC_NO_ERROR && (ret = orc_normalize_decode(&ints, &raw, factor, 32)) == ORC_NO_ERROR)
  {
    const size_t T = raw.nbits / 32;
    if (T > max_T)
      ret = ORC_ERROR_MEMORY;
    else
    {
      memcpy(v, raw.data, T * 4);
      *out_T = T;
    }
  }
  orc_bits_free(&ints);
  orc_bits_free(&raw);
  return ret;
}

int orc_dega_encode_batch_tc(const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive,
                             uint8_t *out, size_t out_cap_per_ch, uint64_t *out_bits, int32_t *err)
{
  int32_t *col = (int32_t *)malloc((T > 0 ? T : 1) * sizeof(int32_t));
  size_t c, t;
  int worst = ORC_NO_ERROR;
  if (col == NULL)
    return ORC_ERROR_MEMORY;
  for (c = 0; c < C; c++)
  {
    int ret;
    for (t = 0; t < T; t++)
      col[t] = x_tc[t * ld + c];
    out_bits[c] = 0;
    ret = orc_dega_encode_i32(col, T, adaptive, out + c * out_cap_per_ch, out_cap_per_ch, &out_bits[c]);
    err[c] = ret;
    if (ret != ORC_NO_ERROR && worst == ORC_NO_ERROR)
      worst = ret;
  }
  free(col);
  return worst;
}

int orc_dega_decode_batch_tc(const uint8_t *in, size_t in_cap_per_ch, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                             int adaptive, int32_t *x_tc, int32_t *err)
{
  int32_t *col = (int32_t *)malloc((T > 0 ? T : 1) * sizeof(int32_t));
  size_t c, t;
  int worst = ORC_NO_ERROR;
  if (col == NULL)
    return ORC_ERROR_MEMORY;
  for (c = 0; c < C; c++)
  {
    size_t got = 0;
    int ret = orc_dega_decode_i32(in + c * in_cap_per_ch, in_bits[c], adaptive, col, T, &got);
    if (ret == ORC_NO_ERROR && got != T)
      ret = ORC_ERROR_INVALID_FORMAT;
    err[c] = ret;
    if (ret == ORC_NO_ERROR)
      for (t = 0; t < T; t++)
        x_tc[t * ld + c] = col[t];
    else if (worst == ORC_NO_ERROR)
      worst = ret;
  }
  free(col);
  return worst;
}
