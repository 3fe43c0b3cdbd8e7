/* dega_hip.h -- C ABI of libdega_hip.so: the MI355X (gfx950) implementation of DCLib's DEGA hot path.
 *
 * This is the drop-in boundary for the path  normalize -> diff -> seg -> bac [adaptive]  (and its inverse) of
 * CenterForSecureEnergyInformatics/data-compressor.  Plain C: pointers, sizes and integer codes only; no torch types.
 * Paths below are relative to the reference's DataCompressor/ directory.
 *
 * What each entry point replaces in the reference:
 *   dega_hip_encode_*   the stage functions  EncodeDifferential (DCLib/src/diff.c:9-23)  ->  EncodeSEG (DCLib/src/seg.c:31-43)
 *                       ->  EncodeBAC (DCLib/src/bac.c:147-166)  run back to back by DCCLI's stage loop
 *                       (DCCLI/src/cli.c:430-466) -- one call codes C independent channels instead of one.
 *   dega_hip_decode_*   DecodeBAC (bac.c:244-263) -> DecodeSEG (seg.c:82-94) -> DecodeDifferential (diff.c:25-37).
 *   dega_hip_normalize_* / dega_hip_denormalize_*   Normalize / Denormalize (DCLib/src/normalize.c:9-27, :29-41).
 *   the bit format       DCIOLib/src/bit_file_buffer.c:220-248, 297-308 (MSB-first bits, big-endian values).
 * The reference-side binding (a row in encoders_decoders[], DCLib/src/enc_dec.c:51-60, whose enc_dec_function_t
 * (DCLib/inc/enc_dec.h:11) pulls the stream out of in_bit_buf, calls these, and pushes the result into out_bit_buf)
 * is shown in INTEGRATION.md and implemented in data-compressor_amd/host/.
 *
 * Per channel the produced bytes and the exact bit length equal what the reference's chain
 *     encode diff # encode seg # encode bac [adaptive]        (valuesize 1..64)
 * produces for that channel alone; errors are per channel and use the reference's codes (common/inc/err_codes.h:8-32).
 *
 * Layouts
 *   samples  x_tc : int32 [T][ld]   time-major, channel c in column c (ld >= C elements per row; lanes = channels read
 *                                   consecutive int32 -> coalesced 256-byte rows per wavefront)
 *   streams  out  : uint8 [C][cap]  channel c's stream starts at out + c*cap; cap is a multiple of 4
 *            bits : uint64 [C]      exact stream length in bits (the last byte is zero padded)
 *            err  : int32 [C]       DEGA_OK or a negative reference error code for that channel
 * "dev" entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t passed as void*, NULL = default
 * stream) without synchronising.  "host" entry points take host pointers and are synchronous; inside they are a
 * pipeline over chunks of channels, and dega_hip_group_* spreads one over every GPU of the node.
 * There is no CPU fallback: without a usable GPU every call fails with DEGA_ERROR_LIBRARY_INIT.
 */
#ifndef DEGA_HIP_H
#define DEGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same values as the reference's common/inc/err_codes.h:8-32 */
#define DEGA_OK 0
#define DEGA_ERROR_INVALID_VALUE (-1)   /* diff/normalize range violation (diff.c:17-18, normalize.c:21-22), bad argument */
#define DEGA_ERROR_INVALID_FORMAT (-3)  /* undecodable stream (seg.c:55-56, bac.c:171-186) */
#define DEGA_ERROR_MEMORY (-6)          /* out of device memory, or a channel's stream does not fit its `cap` bytes */
#define DEGA_ERROR_LIBRARY_INIT (-10)   /* no GPU / HIP runtime failure at init */
#define DEGA_ERROR_LIBRARY_CALL (-11)   /* HIP failure during a call */

typedef struct dega_hip_ctx dega_hip_ctx;     /* one context = one device; not thread safe (like the reference's codecs) */
typedef struct dega_hip_group dega_hip_group; /* several contexts: one per GPU of the node, channels split between them */

/* What the samples of a batch are (the `samples` field of dega_hip_job). */
#define DEGA_SAMPLES_I32 0  /* int32 [T][ld], native byte order, valuesize 1..32 */
#define DEGA_SAMPLES_BE32 1 /* the same as 32-bit big-endian words: what `encode normalize` writes and `decode diff` emits
                               for valuesize 32 (DCIOLib/src/bit_file_buffer.c:297-308) -- swapped on the device, not by the caller */
#define DEGA_SAMPLES_I64 2  /* int64 [T][ld], valuesize 33..64 */
#define DEGA_SAMPLES_F32 3  /* float32 [T][ld] readings, valuesize 1..64: Normalize / Denormalize (DCLib/src/normalize.c:9-41)
                               run inside the encode / decode kernel, one launch per direction */

/* One batch of C channels x T samples for the host-pointer entry points. */
typedef struct dega_hip_job
{
  size_t C, T, ld; /* channels, samples per channel, row pitch of `samples` in elements (>= C) */
  int adaptive;    /* 0 = `bac`, 1 = `bac adaptive` */
  int valuesize;   /* the `valuesize` option of the stages, 1..64 */
  int samples;     /* DEGA_SAMPLES_* */
  float factor;    /* normalization_factor (DEGA_SAMPLES_F32 only) */
} dega_hip_job;

/* ---- lifetime ---------------------------------------------------------------------------------------------------- */
int dega_hip_device_count(void);                            /* number of visible GPUs, 0 if none / no runtime */
int dega_hip_create(int device, dega_hip_ctx **ctx);        /* DEGA_OK or DEGA_ERROR_LIBRARY_INIT / _MEMORY */
void dega_hip_destroy(dega_hip_ctx *ctx);
const char *dega_hip_last_error(const dega_hip_ctx *ctx);   /* text of the last failure on this context ("" if none) */
const char *dega_hip_version(void);

/* Bytes per channel that always suffice for T samples (seg worst case 65 bits/sample, bac expansion, EOF + flush). */
size_t dega_hip_worst_case_bytes(size_t T);

/* ---- DEGA encode / decode, device pointers ----------------------------------------------------------------------- */
/* valuesize: 1..32, the `valuesize` option of the three stages (DCLib/src/enc_dec.c:72).  A sample is the low valuesize
   bits of its int32 container, read unsigned as diff.c:15 does; bits above are ignored on encode and zero on decode.
   The difference must fit valuesize bits signed, else that channel reports ERROR_INVALID_VALUE (diff.c:17-18); the
   decoder caps a codeword's zero prefix at valuesize + 1 (seg.c:55-56,74).  33..64: the *64 entry points below.  adaptive: 0 = `bac`, 1 = `bac adaptive`. */
int dega_hip_encode_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                        uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream);
/* The same over several calls, each taking the next T_seg rows of the channels (x_tc = the first of them): a channel's
   encoder state -- bit queue, interval and model of EncodeBAC (bac.c:33-37,83-84: the statics a reference process holds
   between symbols), finished bits, the held-back word -- is saved to `state` (dega_hip_encode_state_bytes(C) bytes of
   device memory, owned by the caller, opaque) when flags has DEGA_SEGMENT_MORE and picked up when it has
   DEGA_SEGMENT_CONTINUES; out / cap / out_bits / err as above, the same in every call, final after the last one (the
   one without DEGA_SEGMENT_MORE; T_seg = 0 is allowed there).  The streams are those of one call over all the rows.
   This is how the host pipeline codes a batch of few, long channels while its rows are still arriving (one call per
   band of rows, each behind its band's copy in plain stream order), and how a caller codes series of more than 2^25
   samples. */
#define DEGA_SEGMENT_CONTINUES 1
#define DEGA_SEGMENT_MORE 2
size_t dega_hip_encode_state_bytes(size_t C);
int dega_hip_encode_segment_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T_seg, size_t ld, int adaptive, int valuesize,
                                uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *state, unsigned flags, void *stream);
/* in_bits[c] is the exact bit length, or 8*bytes when the stream comes from a zero-padded file.  Decodes exactly T
   samples per channel; a stream holding fewer or more yields DEGA_ERROR_INVALID_FORMAT for that channel. */
int dega_hip_decode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                        int adaptive, int valuesize, int32_t *x_tc, int32_t *err, void *stream);

/* Like dega_hip_decode_dev, for streams whose sample count is not known (a DCLib stream has no header: the count is
   implied by the EOF symbol, bac.c:256): decodes up to max_T samples per channel and reports each channel's count in
   out_count[c]; a channel holding more than max_T samples gets DEGA_ERROR_MEMORY (call again with more room). */
int dega_hip_decode_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                            int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err, void *stream);

/* ---- valuesize 33..64: the same path with 64-bit containers -------------------------------------------------------- */
/* x_tc is int64 [T][ld]; a sample is the low valuesize bits, unsigned; for valuesize 64 the difference wraps and is not
   range checked (diff.c:17); a difference of magnitude 2^63 is coded like 0 (the reference's code number wraps, seg.c:25-28)
   -- the one lossy case, reproduced.  The decoder caps the zero prefix at min(valuesize + 1, 64) (seg.c:74).
   Slabs: dega_hip_worst_case_bytes64. */
size_t dega_hip_worst_case_bytes64(size_t T);
int dega_hip_encode64_dev(dega_hip_ctx *ctx, const int64_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                          uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream);
int dega_hip_decode64_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                          int adaptive, int valuesize, int64_t *x_tc, int32_t *err, void *stream);
int dega_hip_decode64_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                              int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err, void *stream);
int dega_hip_encode64_host(dega_hip_ctx *ctx, const int64_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                           uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err);
int dega_hip_decode64_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                               int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err);

/* ---- float entry / exit (normalize.c), device pointers ----------------------------------------------------------- */
/* v: float32 [T][ld] -> x: int32 [T][ld]; err[c] = DEGA_ERROR_INVALID_VALUE if any sample of channel c fails the range check. */
int dega_hip_normalize_dev(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int valuesize,
                           int32_t *x_tc, int32_t *err, void *stream);
int dega_hip_denormalize_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, float factor, int valuesize,
                             float *v_tc, void *stream);

/* ---- stream compaction: [C][cap] slabs -> one contiguous buffer --------------------------------------------------- */
/* offsets[c] (uint64 [C+1], device) = byte offset of channel c in `packed`; a channel occupies ceil(bits/8) bytes.
   Two steps so that the caller can size `packed`: _offsets fills offsets (exclusive prefix sum, offsets[C] = total),
   _gather copies. */
int dega_hip_compact_offsets_dev(dega_hip_ctx *ctx, const uint64_t *bits, size_t C, uint64_t *offsets, void *stream);
int dega_hip_compact_gather_dev(dega_hip_ctx *ctx, const uint8_t *slabs, size_t cap, const uint64_t *offsets, size_t C,
                                uint8_t *packed, void *stream);

/* ---- synthetic load profiles: deterministic integer random walk, generated on the device ------------ */
/* x[c][0] = 10000 + h(c) mod 50000;  x[c][t] = clamp(x[c][t-1] + (h(c,t) mod (2S+1)) - S, 0, 2^31-1);  channel ids start at
   c0 (so that rank r of a multi-GPU job generates its own channel range). */
int dega_hip_synth_dev(dega_hip_ctx *ctx, int32_t *x_tc, size_t C, size_t T, size_t ld, uint64_t seed, uint64_t c0, uint32_t S, void *stream);

/* ---- LZMH, the reference's second codec (BASELINE config 4) ------------------------------------------------------- */
/* Replaces EncodeLZMH (DCLib/src/lzmh.c:130-370) and DecodeLZMH (:383-574), table row "lzmh" (DCLib/src/enc_dec.c:51-60),
   per channel and bit for bit, including the codec's quirks (an input of exactly 403 bytes encodes to nothing, :161-174;
   the decoder stops once its code register is empty after the last input bit, :571).  One GPU lane per channel.
   encode: in = uint8 [C][stride] (device, 16-byte aligned, stride a multiple of 16), in_len[c] <= stride bytes of channel c;
           out = uint8 [C][cap] slabs of 32-bit big-endian words (cap a multiple of 16, >= dega_hip_lzmh_worst_case_bytes(n)
           never overflows), out_bits[c] = exact bit length, err[c] = 0 | ERROR_MEMORY (slab too small) | ERROR_INVALID_VALUE.
   decode: the inverse: in/cap/in_bits as produced by encode (cap a multiple of 4), out = uint8 [C][stride] (stride a
           multiple of 8), out_len[c] = decoded bytes; ERROR_MEMORY when a channel does not fit its row.
   render: the synthetic LZMH workload of SURVEY.md 8(d): int32 channels [T][ld] (centi-units) as ASCII "%d.%02d\n" lines. */
size_t dega_hip_lzmh_worst_case_bytes(size_t n);
int dega_hip_lzmh_encode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out, size_t cap,
                             uint64_t *out_bits, int32_t *err, void *stream);
int dega_hip_lzmh_decode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out, size_t stride,
                             uint64_t *out_len, int32_t *err, void *stream);
int dega_hip_lzmh_render_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, uint8_t *out, size_t stride,
                             uint64_t *out_len, int32_t *err, void *stream);
int dega_hip_lzmh_encode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out, size_t cap,
                              uint64_t *out_bits, int32_t *err);
int dega_hip_lzmh_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out, size_t stride,
                              uint64_t *out_len, int32_t *err);
/* The same for a group, through the host pipeline: chunks of channels on their own streams (upload, kernel, pack, download
   overlapped), contiguous channel ranges per device, streams packed back to back: channel c at packed[offsets[c] ..
   offsets[c+1]) (C + 1 offsets, ceil(bits / 8) bytes each).  Host memory; pinned memory is used in place.  encode returns
   ERROR_MEMORY with the size needed in offsets[C] when packed_cap is too small; stride as for the _dev forms. */
int dega_hip_group_lzmh_encode(dega_hip_group *group, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *packed,
                               size_t packed_cap, uint64_t *offsets, uint64_t *out_bits, int32_t *err);
int dega_hip_group_lzmh_decode(dega_hip_group *group, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, uint8_t *out,
                               size_t stride, uint64_t *out_len, int32_t *err);

/* ---- float entry / exit fused into the coder kernels, device pointers ----------------------------------- */
/* v_tc: float32 [T][ld].  One launch: Normalize on each value as it enters the fill phase (normalize.c:16-24; a value
   failing the range check of :21 gives that channel DEGA_ERROR_INVALID_VALUE), then diff -> seg -> bac as above.  No int32
   intermediate exists in HBM.  valuesize 1..64.  decode: Denormalize (:36-38) in the row write; out_count NULL = exactly
   T samples per channel, else up to T and the counts are reported. */
int dega_hip_encode_f32_dev(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int adaptive, int valuesize,
                            uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream);
int dega_hip_decode_f32_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                            float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err, void *stream);

/* ---- host pointers: the pipelined path DCCLI's stage loop (DCCLI/src/cli.c:430-466) ends up on ---------------------------- */
/* `samples` and the outputs are HOST memory (pageable or pinned).  The batch is cut into chunks of channels, each on a
   stream of its own: upload of its columns, kernels, packing, download of its stream bytes -- copies and kernels of
   different chunks overlap, device and pinned buffers belong to the context and only grow, and only stream bytes come
   back: channel c occupies packed[offsets[c] .. offsets[c+1]) (ceil(bits / 8) bytes, channel order; offsets has C + 1
   entries).  If packed_cap is too small the call returns DEGA_ERROR_MEMORY with the size needed in offsets[C] (bits and err
   are valid then).  decode: the inverse; out_count NULL = every channel holds exactly T samples, else up to T and the
   counts are reported (a DCLib stream has no header: bac.c:256). */
int dega_hip_encode_job_host(dega_hip_ctx *ctx, const dega_hip_job *job, const void *samples, uint8_t *packed, size_t packed_cap,
                             uint64_t *offsets, uint64_t *out_bits, int32_t *err);
int dega_hip_decode_job_host(dega_hip_ctx *ctx, const dega_hip_job *job, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits,
                             void *samples, uint64_t *out_count, int32_t *err);

/* ---- every GPU of the node: channel ranges per device, host-side concatenate, no collective ------------------------------- */
/* Channels are independent units (every stream starts from last_value = 0, DCLib/src/diff.c:11, and InitModel(),
   DCLib/src/bac.c:150), so a batch shards as contiguous channel ranges [g*C/G, (g+1)*C/G): one context, one host thread
   and one set of streams per device; every device packs its own streams and copies them to their final place in `packed`
   once the sizes of the ranges in front of it are known.  Same arguments and results as the single-context calls above --
   a group of one IS that call.  devices NULL / n <= 0: every visible device, or the comma separated list in the
   environment variable DEGA_DEVICES (DEGA_DEVICE for a single index). */
/* The partition itself (no GPU needed): cuts[g] .. cuts[g+1] is device g's channel range, cuts has G + 1 entries; ranges
   are whole 512-channel workgroup pairs where the batch is large enough. */
int dega_hip_split_channels(size_t C, int G, size_t *cuts);
int dega_hip_group_create(const int *devices, int n, dega_hip_group **group);
void dega_hip_group_destroy(dega_hip_group *group);
int dega_hip_group_size(const dega_hip_group *group);
dega_hip_ctx *dega_hip_group_context(dega_hip_group *group, int i); /* member i (for the LZMH calls, profiling, last_error) */
const char *dega_hip_group_last_error(const dega_hip_group *group);
int dega_hip_group_encode(dega_hip_group *group, const dega_hip_job *job, const void *samples, uint8_t *packed, size_t packed_cap,
                          uint64_t *offsets, uint64_t *out_bits, int32_t *err);
int dega_hip_group_decode(dega_hip_group *group, const dega_hip_job *job, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits,
                          void *samples, uint64_t *out_count, int32_t *err);

/* Pinned host memory for callers that can keep their samples there: copies then run at link speed without the
   runtime's staging of pageable memory.  NULL when there is no GPU runtime. */
void *dega_hip_pinned_alloc(size_t bytes);
void dega_hip_pinned_free(void *p);

/* ---- host pointers, the earlier forms (slabs in and out; the same pipeline underneath) --------------------------------- */
int dega_hip_encode_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                         uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err);
int dega_hip_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                         int adaptive, int valuesize, int32_t *x_tc, int32_t *err);
int dega_hip_decode_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                             int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err);
/* The same encode, streams returned packed: channel c occupies packed[offsets[c] .. offsets[c+1]) -- ceil(bits/8) bytes, in
   channel order, which is how DCCLI-style callers concatenate them anyway -- so only the stream bytes cross PCIe, not
   C x cap slab bytes.  offsets has C + 1 entries; if packed_cap is too small the call returns DEGA_ERROR_MEMORY with the
   size needed in offsets[C] (bits and err are valid then). */
int dega_hip_encode_packed_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                uint8_t *packed, size_t packed_cap, uint64_t *offsets, uint64_t *out_bits, int32_t *err);
/* The inverse: channel c's stream is packed[offsets[c] .. offsets[c+1]) (at least ceil(in_bits[c] / 8) bytes); out_count NULL
   = every channel holds exactly T samples, else up to T and the counts are reported (as dega_hip_decode_var_host). */
int dega_hip_decode_packed_host(dega_hip_ctx *ctx, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, size_t T,
                                size_t ld, int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err);
/* float32 channels in, DEGA streams out: Normalize runs inside the encode kernel, Denormalize inside the decode kernel
   (one launch per direction, valuesize 1..64). */
int dega_hip_encode_f32_host(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int adaptive, int valuesize,
                             uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err);
int dega_hip_decode_f32_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                             float factor, int adaptive, int valuesize, float *v_tc, int32_t *err);

int dega_hip_decode_f32_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                 float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err);

/* ---- measurement hook ---------------------------------------------------------------------------------------------- */
/* Average duration in milliseconds of the DEGA encode (which=0) / DEGA decode (1) / LZMH encode (2) / LZMH decode (3) kernel launches enqueued since the last
   reset, measured with hipEvents on the stream each launch used (enabled with dega_hip_profile(ctx, 1)); returns the
   number of launches measured.  Used by bench.py for the roofline figure. */
int dega_hip_profile(dega_hip_ctx *ctx, int enable);
int dega_hip_profile_read(dega_hip_ctx *ctx, int which, double *avg_ms, int reset);

#ifdef __cplusplus
}
#endif

#endif
