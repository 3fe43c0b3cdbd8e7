/* lzmh_oracle.c -- CPU restatement of the reference's LZMH codec (DCLib/src/lzmh.c; third-party LZ77 + adaptive
 * frequency-list code by M. Ringwelski, adapted by A. Unterweger), for BASELINE config 4.
 *
 * TEST INFRASTRUCTURE ONLY (see dega_oracle.h).  Works on in-memory bit streams like the other oracle stages.
 * Stream format (lzmh.c:34-46):  00+byte literal | 010+offset(7)+len | 0110/01110/011110/011111+len = one of the four
 * most recent offsets | 1+list code = literal coded by its position in the frequency-sorted symbol list (19 static
 * codes, lzmh.c:86-106);  len: 0+3 bits (3..10) | 10+3 bits (11..18) | 11+8 bits (19..274).
 * The quirks of the reference are kept, they are part of its observable behaviour:
 *   - the look-back distance before the 403-byte ring has wrapped is (uint8_t)read_index (lzmh.c:176-180), so it drops to 0
 *     at read index 256;
 *   - an input of exactly 403 bytes produces an empty output (write index wraps onto the read index, lzmh.c:168-172);
 *   - the decoder stops as soon as its bit buffer holds only zero bits after the last input bit (lzmh.c:571), and
 *     returns success on an unknown list code (lzmh.c:447-449).
 */
#include "dega_oracle.h"

#include <stdlib.h>
#include <string.h>

#define LZ_MAX_OFFSET 128
#define LZ_MAX_LENGTH 274
#define RING (LZ_MAX_OFFSET + LZ_MAX_LENGTH + 1) /* 403, lzmh.c:64 */
#define LIST_LEN 48
#define TREE_LEN 19

static const struct { uint8_t code, length; } list_code[TREE_LEN] = { /* lzmh.c:86-106 */
  { 0x0F, 4 }, { 0x0E, 4 }, { 0x0D, 4 }, { 0x0C, 4 }, { 0x17, 5 }, { 0x16, 5 }, { 0x15, 5 }, { 0x14, 5 }, { 0x13, 5 },
  { 0x25, 6 }, { 0x24, 6 }, { 0x23, 6 }, { 0x22, 6 }, { 0x43, 7 }, { 0x42, 7 }, { 0x83, 8 }, { 0x82, 8 }, { 0x81, 8 }, { 0x80, 8 }
};

/* helpers shared with dega_oracle.c (duplicated here to keep the two files independent) */
static int lz_reserve(orc_bits_t *b, size_t total_bits)
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

static int lz_put(orc_bits_t *b, uint32_t v, unsigned k) /* low k bits of v, MSB first */
{
  int ret;
  if ((ret = lz_reserve(b, b->nbits + k)) != ORC_NO_ERROR)
    return ret;
  while (k > 0)
  {
    k--;
    if ((v >> k) & 1u)
      b->data[b->nbits >> 3] |= (uint8_t)(0x80u >> (b->nbits & 7));
    b->nbits++;
  }
  return ORC_NO_ERROR;
}

static unsigned wrap(int pos) /* getPosInBuffer, lzmh.c:111-119 */
{
  while (pos < 0)
    pos += RING;
  while (pos >= RING)
    pos -= RING;
  return (unsigned)pos;
}

int orc_lzmh_encode(const orc_bits_t *in, orc_bits_t *out)
{
  uint8_t buf[RING];
  struct { uint8_t symbol; int count; } list[LIST_LEN];
  int offsets[4] = { 0, 0, 0, 0 };
  size_t W = 0, R = 0, H = 0; /* write / read / history indices, lzmh.c:143-146 */
  size_t ip = 0;              /* next input byte */
  const size_t n = in->nbits / 8;
  int i, ret;
  if (in->nbits % 8 != 0)
    return ORC_ERROR_LIBRARY_CALL; /* the last READ_VALUE_BITS_CHECKED(8) would come up short */
  memset(buf, 0, sizeof(buf));
  for (i = 0; i < LIST_LEN; i++)
  {
    list[i].count = 0;
    list[i].symbol = 0;
  }
  while (ip < n && W < RING) /* lzmh.c:161-167 */
    buf[W++] = in->data[ip++];
  if (W >= RING)
    W -= RING;
  while (ip < n || W != R) /* :174 */
  {
    const int maxoffset = H != 0 ? LZ_MAX_OFFSET : (int)(uint8_t)R; /* :176-180 */
    int maxlength, bestlength = 2, bestoffset = 0, offset, length;
    if (W > R) /* :181-191 */
      maxlength = (int)(W - R) > LZ_MAX_LENGTH ? LZ_MAX_LENGTH : (int)(W - R);
    else if (W < R)
      maxlength = (int)(W + RING - R) > LZ_MAX_LENGTH ? LZ_MAX_LENGTH : (int)(W + RING - R);
    else
      maxlength = LZ_MAX_LENGTH;
    for (offset = 1; offset <= maxoffset && bestlength < maxlength; ++offset) /* :196-214 */
    {
      if (buf[wrap((int)R - offset)] == buf[R] && buf[wrap((int)R - offset + bestlength)] == buf[wrap((int)R + bestlength)])
      {
        for (length = 1; length < maxlength && buf[wrap((int)R - offset + length)] == buf[wrap((int)R + length)]; ++length)
          ;
        if (length > bestlength)
        {
          bestlength = length;
          bestoffset = offset;
        }
      }
    }
    if (bestlength >= 3) /* :216-283 */
    {
      if (offsets[0] == bestoffset)
        ret = lz_put(out, 0x06, 4);
      else if (offsets[1] == bestoffset)
      {
        offsets[1] = offsets[0];
        offsets[0] = bestoffset;
        ret = lz_put(out, 0x0E, 5);
      }
      else if (offsets[2] == bestoffset)
      {
        offsets[2] = offsets[1];
        offsets[1] = offsets[0];
        offsets[0] = bestoffset;
        ret = lz_put(out, 0x1E, 6);
      }
      else if (offsets[3] == bestoffset)
      {
        offsets[3] = offsets[2];
        offsets[2] = offsets[1];
        offsets[1] = offsets[0];
        offsets[0] = bestoffset;
        ret = lz_put(out, 0x1F, 6);
      }
      else
      {
        offsets[3] = offsets[2];
        offsets[2] = offsets[1];
        offsets[1] = offsets[0];
        offsets[0] = bestoffset;
        ret = lz_put(out, 0x100u | (unsigned)(bestoffset - 1), 10);
      }
      if (ret != ORC_NO_ERROR)
        return ret;
      if (bestlength < 11)
        ret = lz_put(out, (unsigned)(bestlength - 3), 4);
      else if (bestlength < 19)
        ret = lz_put(out, 0x10u | (unsigned)(bestlength - 11), 5);
      else
        ret = lz_put(out, 0x300u | (unsigned)(bestlength - 19), 10);
      if (ret != ORC_NO_ERROR)
        return ret;
      R += (size_t)bestlength;
      if (R >= RING)
        R -= RING;
    }
    else /* literal through the frequency list, :285-333 */
    {
      const uint8_t sym = buf[R++];
      int found = 0xFFFF;
      length = 0;
      if (R >= RING)
        R -= RING;
      while (length < LIST_LEN && list[length].count > 0)
      {
        if (list[length].symbol == sym)
        {
          found = length;
          if (list[length].count < 65535)
          {
            const int nc = list[length].count + 1;
            while (length > 0 && nc > list[length - 1].count) /* only the symbols move (:306-309) */
            {
              list[length].symbol = list[length - 1].symbol;
              length--;
            }
            list[length].count = nc;
            list[length].symbol = sym;
          }
          break;
        }
        length++;
      }
      if (found == 0xFFFF && length < LIST_LEN)
      {
        list[length].symbol = sym;
        list[length].count = 1;
      }
      if (found < TREE_LEN)
        ret = lz_put(out, list_code[found].code, list_code[found].length);
      else
        ret = lz_put(out, sym, 10); /* 00 + byte */
      if (ret != ORC_NO_ERROR)
        return ret;
    }
    if (R > H) /* :343-351 */
    {
      if (R - H > LZ_MAX_OFFSET)
        H = R - LZ_MAX_OFFSET;
    }
    else if (R + RING - H > LZ_MAX_OFFSET)
      H = (R + RING - LZ_MAX_OFFSET) % RING;
    while (ip < n && W != H) /* :354-363 */
    {
      buf[W++] = in->data[ip++];
      if (W >= RING)
        W -= RING;
    }
  }
  return ORC_NO_ERROR;
}

int orc_lzmh_decode(const orc_bits_t *in, orc_bits_t *out)
{
  uint8_t hist[LZ_MAX_OFFSET];
  struct { uint8_t symbol; int count; } list[LIST_LEN];
  int offsets[4] = { 0, 0, 0, 0 };
  uint32_t code_sym = 0;
  int8_t code_length = 0;
  size_t hp = 0, ip = 0; /* history position, next input bit */
  size_t i;
  int length, offset, ret;
  memset(hist, 0, sizeof(hist));
  for (i = 0; i < LIST_LEN; i++)
  {
    list[i].count = 0;
    list[i].symbol = 0;
  }
  do /* lzmh.c:408-571 */
  {
    while (ip < in->nbits && (32 - code_length) >= 8)
    {
      const uint64_t bit = (in->data[ip >> 3] >> (7 - (ip & 7))) & 1u;
      ip++;
      code_length = (int8_t)(code_length + 1);
      code_sym |= (uint32_t)(bit << ((32 - code_length) & 63));
    }
    if ((code_sym & 0x80000000u) != 0) /* list code */
    {
      for (i = 0; i < TREE_LEN; i++)
      {
        if (code_length >= list_code[i].length && (code_sym >> (32 - list_code[i].length)) == list_code[i].code)
        {
          const uint8_t sym = list[i].symbol;
          code_length = (int8_t)(code_length - list_code[i].length);
          code_sym <<= list_code[i].length;
          if ((ret = lz_put(out, sym, 8)) != ORC_NO_ERROR)
            return ret;
          hist[hp++] = sym;
          if (hp >= LZ_MAX_OFFSET)
            hp -= LZ_MAX_OFFSET;
          if (list[i].count < 65535)
          {
            length = list[i].count + 1;
            while (i > 0 && length > list[i - 1].count)
            {
              list[i].symbol = list[i - 1].symbol;
              i--;
            }
            list[i].count = (uint16_t)length;
            list[i].symbol = sym;
          }
          break;
        }
      }
      if (i == TREE_LEN)
        return ORC_NO_ERROR; /* :447-449 */
    }
    else if ((code_sym & 0x40000000u) == 0) /* 00 + byte */
    {
      const uint8_t sym = (uint8_t)((code_sym >> 22) & 0xFF);
      code_length = (int8_t)(code_length - 10);
      code_sym <<= 10;
      if ((ret = lz_put(out, sym, 8)) != ORC_NO_ERROR)
        return ret;
      hist[hp++] = sym;
      if (hp >= LZ_MAX_OFFSET)
        hp -= LZ_MAX_OFFSET;
      for (length = 0; length < LIST_LEN && list[length].count > 0 && list[length].symbol != sym; length++)
        ;
      if (length < LIST_LEN && list[length].count < 65535)
      {
        const int nc = list[length].count + 1;
        while (length > 0 && nc > list[length - 1].count)
        {
          list[length] = list[length - 1];
          length--;
        }
        list[length].count = nc;
        list[length].symbol = sym;
      }
    }
    else /* match */
    {
      code_length = (int8_t)(code_length - 2);
      code_sym <<= 2;
      if ((code_sym & 0x80000000u) == 0)
      {
        offset = (int)((code_sym >> 24) & 0x7F) + 1;
        code_length = (int8_t)(code_length - 8);
        code_sym <<= 8;
        offsets[3] = offsets[2];
        offsets[2] = offsets[1];
        offsets[1] = offsets[0];
        offsets[0] = offset;
      }
      else
      {
        code_length = (int8_t)(code_length - 1);
        code_sym <<= 1;
        if ((code_sym & 0x80000000u) == 0)
          offset = offsets[0];
        else
        {
          code_length = (int8_t)(code_length - 1);
          code_sym <<= 1;
          if ((code_sym & 0x80000000u) == 0)
          {
            offset = offsets[1];
            offsets[1] = offsets[0];
            offsets[0] = offset;
          }
          else
          {
            code_length = (int8_t)(code_length - 1);
            code_sym <<= 1;
            if ((code_sym & 0x80000000u) == 0)
            {
              offset = offsets[2];
              offsets[2] = offsets[1];
              offsets[1] = offsets[0];
              offsets[0] = offset;
            }
            else
            {
              offset = offsets[3];
              offsets[3] = offsets[2];
              offsets[2] = offsets[1];
              offsets[1] = offsets[0];
              offsets[0] = offset;
            }
          }
        }
        code_length = (int8_t)(code_length - 1);
        code_sym <<= 1;
      }
      if ((code_sym & 0x80000000u) == 0)
      {
        length = (int)((code_sym >> 28) & 0x07) + 3;
        code_length = (int8_t)(code_length - 4);
        code_sym <<= 4;
      }
      else
      {
        code_length = (int8_t)(code_length - 1);
        code_sym <<= 1;
        if ((code_sym & 0x80000000u) == 0)
        {
          length = (int)((code_sym >> 28) & 0x7) + 11;
          code_length = (int8_t)(code_length - 4);
          code_sym <<= 4;
        }
        else
        {
          length = (int)((code_sym >> 23) & 0xFF) + 19;
          code_length = (int8_t)(code_length - 9);
          code_sym <<= 9;
        }
      }
      for (i = 0; i < (size_t)length; ++i)
      {
        int p = (int)hp - offset;
        uint8_t sym;
        while (p < 0)
          p += LZ_MAX_OFFSET;
        while (p >= LZ_MAX_OFFSET)
          p -= LZ_MAX_OFFSET;
        sym = hist[p];
        if ((ret = lz_put(out, sym, 8)) != ORC_NO_ERROR)
          return ret;
        hist[hp++] = sym;
        if (hp >= LZ_MAX_OFFSET)
          hp -= LZ_MAX_OFFSET;
      }
    }
  } while (ip < in->nbits || code_sym > 0);
  return ORC_NO_ERROR;
}
