/* io.h -- integer types of the I/O layer.  Mirrors the reference's common/inc/io.h:12-14,62-63 for IO_SIZE_BITS = 64
   (its default): io_int_t / io_uint_t are the 64-bit types every enc_dec_function_t returns and the value I/O uses. */
#ifndef DC_AMD_IO_H
#define DC_AMD_IO_H

#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#define IO_SIZE_BITS 64
typedef int64_t io_int_t;
typedef uint64_t io_uint_t;

#endif
