/* bit_io.c -- byte streams (FILE or memory) and the bit-granular layer on top of them.
 *
 * Our implementation of the API named in inc/file_buffer.h and inc/bit_file_buffer.h (the reference's is
 * DCIOLib/src/{buffer,file_buffer,bit_file_buffer}.c).  Design: a stream is a growable byte array plus a bit count;
 * FILE streams are read in whole when first opened for reading and written out when closed.  Observable behaviour
 * that the codecs and DCCLI rely on is kept: MSB-first bit packing, exact bit length across a write->read switch,
 * zero padding of the last file byte, and the one 0x00 byte an empty output file gets (bit_file_buffer.c:310-333).
 */
#include "bit_file_buffer.h"
#include "err_codes.h"

#include <stdlib.h>
#include <string.h>

struct file_buffer_t
{
  FILE *file; /* NULL for memory streams */
  file_buffer_mode_t mode;
  uint8_t *data;
  size_t nbits;    /* valid bits in data */
  size_t cap;      /* allocated bytes */
  size_t read_pos; /* next bit to read */
  size_t written;  /* bits written since the last reset (statistics) */
};

struct bit_file_buffer_t
{
  file_buffer_t *fb;
};

file_buffer_t *AllocateFileBuffer(void)
{
  return (file_buffer_t *)calloc(1, sizeof(file_buffer_t));
}

void FreeFileBuffer(file_buffer_t *fb)
{
  free(fb);
}

static int reserve_bits(file_buffer_t *fb, size_t total_bits)
{
  const size_t need = (total_bits + 7) / 8 + 8;
  if (need > fb->cap)
  {
    size_t ncap = fb->cap ? fb->cap : 1024;
    uint8_t *nd;
    while (ncap < need)
      ncap *= 2;
    if ((nd = (uint8_t *)realloc(fb->data, ncap)) == NULL)
      return ERROR_MEMORY;
    memset(nd + fb->cap, 0, ncap - fb->cap);
    fb->data = nd;
    fb->cap = ncap;
  }
  return NO_ERROR;
}

static int init_common(file_buffer_t *fb, FILE *file, file_buffer_mode_t mode, size_t buffer_size)
{
  if (buffer_size == 0)
    return ERROR_INVALID_VALUE;
  memset(fb, 0, sizeof(*fb));
  fb->file = file;
  fb->mode = mode;
  return reserve_bits(fb, 8 * buffer_size);
}

int InitFileBuffer(file_buffer_t *fb, FILE *file, file_buffer_mode_t mode, size_t buffer_size)
{
  int ret;
  if ((ret = init_common(fb, file, mode, buffer_size)) != NO_ERROR)
    return ret;
  if (mode == FBM_READING && file != NULL) /* slurp the file */
  {
    size_t total = 0;
    for (;;)
    {
      size_t got;
      if ((ret = reserve_bits(fb, 8 * (total + 65536))) != NO_ERROR)
        return ret;
      got = fread(fb->data + total, 1, 65536, file);
      total += got;
      if (got < 65536)
        break;
    }
    if (ferror(file))
      return ERROR_FILE_IO;
    fb->nbits = 8 * total;
  }
  return NO_ERROR;
}

int InitFileBufferInMemory(file_buffer_t *fb, file_buffer_mode_t mode, size_t buffer_size)
{
  return init_common(fb, NULL, mode, buffer_size);
}

void UninitFileBuffer(file_buffer_t *fb)
{
  free(fb->data);
  fb->data = NULL;
  fb->cap = 0;
}

file_buffer_mode_t GetFileBufferMode(const file_buffer_t *fb)
{
  return fb->mode;
}

size_t GetFileBufferSize(const file_buffer_t *fb)
{
  return fb->cap;
}

bit_file_buffer_t *AllocateBitFileBuffer(void)
{
  return (bit_file_buffer_t *)calloc(1, sizeof(bit_file_buffer_t));
}

void FreeBitFileBuffer(bit_file_buffer_t *bb)
{
  free(bb);
}

void InitBitFileBuffer(bit_file_buffer_t *bb, file_buffer_t *fb)
{
  bb->fb = fb;
}

/* Closing a writing stream that is backed by a file writes the bytes out, the last one zero padded; an empty stream
   still produces one zero byte, like the reference's unconditional fractional flush. */
void UninitBitFileBuffer(bit_file_buffer_t *bb)
{
  file_buffer_t *fb = bb->fb;
  if (fb != NULL && fb->mode == FBM_WRITING && fb->file != NULL)
  {
    const size_t nbytes = fb->nbits == 0 ? 1 : (fb->nbits + 7) / 8;
    fwrite(fb->data, 1, nbytes, fb->file);
    fflush(fb->file);
  }
}

int EndOfBitFileBuffer(const bit_file_buffer_t *bb)
{
  const file_buffer_t *fb = bb->fb;
  if (fb->mode != FBM_READING)
    return 0;
  return fb->read_pos >= fb->nbits;
}

void GetActualBitFileOffset(const bit_file_buffer_t *bb, io_int_t *byte_offset, uint8_t *bit_offset)
{
  const file_buffer_t *fb = bb->fb;
  const size_t pos = fb->mode == FBM_READING ? fb->read_pos : fb->nbits;
  *byte_offset = (io_int_t)(pos / 8);
  *bit_offset = (uint8_t)(pos % 8);
}

int SetBitFileBufferMode(bit_file_buffer_t *bb, file_buffer_mode_t mode)
{
  file_buffer_t *fb = bb->fb;
  if (fb->mode == mode)
    return NO_ERROR;
  if (fb->mode == FBM_WRITING && mode == FBM_READING) /* the only supported switch; keeps the exact bit length */
  {
    fb->mode = FBM_READING;
    fb->read_pos = 0;
    return NO_ERROR;
  }
  return ERROR_INVALID_MODE;
}

int ResetBitFileBuffer(bit_file_buffer_t *bb, file_buffer_mode_t mode)
{
  file_buffer_t *fb = bb->fb;
  if (fb->file != NULL)
    return ERROR_FILE_IO; /* like the reference: only memory streams can be cleared */
  if (fb->data != NULL)
    memset(fb->data, 0, fb->cap);
  fb->nbits = 0;
  fb->read_pos = 0;
  fb->written = 0;
  fb->mode = mode;
  return NO_ERROR;
}

static uint64_t take_bits(file_buffer_t *fb, unsigned k) /* k <= 64, enough bits available */
{
  uint64_t acc = 0;
  while (k > 0)
  {
    const unsigned used = (unsigned)(fb->read_pos & 7);
    const unsigned room = 8 - used;
    const unsigned take = k < room ? k : room;
    acc = (acc << take) | ((fb->data[fb->read_pos >> 3] >> (room - take)) & ((1u << take) - 1));
    fb->read_pos += take;
    k -= take;
  }
  return acc;
}

static void give_bits(file_buffer_t *fb, uint64_t v, unsigned k) /* k <= 64, space reserved */
{
  while (k > 0)
  {
    const unsigned used = (unsigned)(fb->nbits & 7);
    const unsigned room = 8 - used;
    const unsigned take = k < room ? k : room;
    const unsigned chunk = (unsigned)((v >> (k - take)) & ((1u << take) - 1));
    fb->data[fb->nbits >> 3] |= (uint8_t)(chunk << (room - take));
    fb->nbits += take;
    k -= take;
  }
}

io_int_t ReadBitFileBuffer(bit_file_buffer_t *bb, uint8_t *output, size_t nbits)
{
  file_buffer_t *fb = bb->fb;
  size_t left, done = 0;
  if (fb->mode != FBM_READING)
    return ERROR_INVALID_MODE;
  left = fb->nbits - fb->read_pos;
  if (nbits > left)
    nbits = left; /* short read: return what is there */
  while (done < nbits)
  {
    const unsigned k = nbits - done >= 8 ? 8 : (unsigned)(nbits - done);
    output[done / 8] = (uint8_t)(take_bits(fb, k) << (8 - k)); /* a trailing fraction is left aligned */
    done += k;
  }
  return (io_int_t)done;
}

io_int_t WriteBitFileBuffer(bit_file_buffer_t *bb, const uint8_t *input, size_t nbits)
{
  file_buffer_t *fb = bb->fb;
  size_t done = 0;
  if (fb->mode != FBM_WRITING)
    return ERROR_INVALID_MODE;
  if (reserve_bits(fb, fb->nbits + nbits) != NO_ERROR)
    return ERROR_MEMORY;
  while (done < nbits)
  {
    const unsigned k = nbits - done >= 8 ? 8 : (unsigned)(nbits - done);
    give_bits(fb, (uint64_t)(input[done / 8] >> (8 - k)), k);
    done += k;
  }
  fb->written += nbits;
  return (io_int_t)nbits;
}

io_int_t ReadSingleValueFromBitFileBuffer(bit_file_buffer_t *bb, io_uint_t *value, size_t nbits)
{
  file_buffer_t *fb = bb->fb;
  if (fb->mode != FBM_READING)
    return ERROR_INVALID_MODE;
  if (nbits > 64)
    return ERROR_INVALID_VALUE;
  if (fb->nbits - fb->read_pos < nbits) /* short read: consume the rest, report the count */
  {
    const size_t left = fb->nbits - fb->read_pos;
    fb->read_pos = fb->nbits;
    return (io_int_t)left;
  }
  *value = nbits ? take_bits(fb, (unsigned)nbits) : 0;
  return (io_int_t)nbits;
}

io_int_t WriteSingleValueToBitFileBuffer(bit_file_buffer_t *bb, const io_uint_t *value, size_t nbits)
{
  file_buffer_t *fb = bb->fb;
  if (fb->mode != FBM_WRITING)
    return ERROR_INVALID_MODE;
  if (nbits > 64)
    return ERROR_INVALID_VALUE;
  if (reserve_bits(fb, fb->nbits + nbits) != NO_ERROR)
    return ERROR_MEMORY;
  if (nbits > 0)
    give_bits(fb, nbits < 64 ? (*value & (((uint64_t)1 << nbits) - 1)) : *value, (unsigned)nbits);
  fb->written += nbits;
  return (io_int_t)nbits;
}
