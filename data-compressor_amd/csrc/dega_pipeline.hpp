// dega_pipeline.hpp -- the host-pointer entry points of libdega_hip.so (included by dega_hip.hip).
//
// What DCCLI's stage loop hands a codec is host memory (DCCLI/src/cli.c:430-466); a batch of C channels therefore has to
// cross PCIe twice.  Everything here is about doing that at link speed:
//   * a batch is cut into chunks of channels; every chunk has a stream of its own that carries   H2D of its columns ->
//     encode kernel -> offsets -> gather -> D2H of its packed streams   (decode: the mirror image), so that the copy
//     engines and the kernels of different chunks overlap.  One channel of T samples takes the same time whatever the
//     batch size (the coder is serial per channel), so chunks also have to run side by side -- they do, on their streams;
//   * nothing is allocated per call: device buffers and the small pinned mirrors belong to the context and only grow;
//   * only stream bytes come back (packed, channel order); slabs are sized for the usual case (no longer than the
//     samples) and a chunk that does not fit is redone with worst-case slabs -- never a worst-case memset or D2H.
// A group runs one such pipeline per device on a host thread each, over contiguous channel ranges; the packed streams
// of the devices are concatenated by the host (each device copies to its final place once the sizes in front of it are
// known).  No collective, no peer traffic (channels are independent: diff.c:11, bac.c:150).
#pragma once

#include <sched.h>

struct GrowDev // device buffer that only grows
{
  void *p = nullptr;
  size_t cap = 0;
  hipError_t need(size_t n)
  {
    if (n <= cap)
      return hipSuccess;
    if (p != nullptr)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess)
    {
      (void)hipGetLastError();
      want = n;
      e = hipMalloc(&p, want);
    }
    if (e == hipSuccess)
      cap = want;
    else
      p = nullptr;
    return e;
  }
  void release()
  {
    if (p != nullptr)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct GrowPin // pinned host buffer that only grows
{
  void *p = nullptr;
  size_t cap = 0;
  hipError_t need(size_t n)
  {
    if (n <= cap)
      return hipSuccess;
    if (p != nullptr)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 8 + 4096;
    const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess)
      cap = want;
    else
      p = nullptr;
    return e;
  }
  void release()
  {
    if (p != nullptr)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// A few host threads that copy rows between the caller's (pageable, pitched) arrays and the pinned staging ring; eight of
// them with cache-bypassing stores keep up with the PCIe link (~57 GB/s) on pitched rows.
// memcpy whose stores bypass the cache: the destination (a staging buffer the DMA engine reads next, or the caller's
// array) is not read back by this thread, and ordinary stores would first fetch every destination line
static void copy_streaming(uint8_t *dst, const uint8_t *src, size_t n)
{
  typedef long long v2 __attribute__((vector_size(16), aligned(16)));
  typedef long long v2u __attribute__((vector_size(16), aligned(1)));
  if (n < 4096)
  {
    memcpy(dst, src, n);
    return;
  }
  const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  memcpy(dst, src, head);
  dst += head;
  src += head;
  n -= head;
  const size_t body = n & ~(size_t)63;
  for (size_t i = 0; i < body; i += 64)
  {
    const v2 a = *(const v2u *)(src + i), b = *(const v2u *)(src + i + 16), c = *(const v2u *)(src + i + 32), d = *(const v2u *)(src + i + 48);
    __builtin_nontemporal_store(a, (v2 *)(dst + i));
    __builtin_nontemporal_store(b, (v2 *)(dst + i + 16));
    __builtin_nontemporal_store(c, (v2 *)(dst + i + 32));
    __builtin_nontemporal_store(d, (v2 *)(dst + i + 48));
  }
  memcpy(dst + body, src + body, n - body);
}

struct CopyPool
{
  static constexpr int N = 8; // (measured again in round 3 with 12 and 16 threads on a 16-core share: no difference -- the ring is not the limit)
  std::thread th[N];
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::function<void(int)> job;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false, started = false;

  void start()
  {
    if (started)
      return;
    started = true;
    for (int i = 0; i < N; i++)
      th[i] = std::thread([this, i] {
        uint64_t seen = 0;
        for (;;)
        {
          std::function<void(int)> f;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            if (stop)
              return;
            seen = generation;
            f = job;
          }
          f(i);
          {
            std::lock_guard<std::mutex> lk(m);
            if (--pending == 0)
              cv_done.notify_all();
          }
        }
      });
  }
  void run(const std::function<void(int)> &f) // f(part) for part = 0 .. N-1, returns when all are done
  {
    start();
    std::unique_lock<std::mutex> lk(m);
    job = f;
    pending = N;
    generation++;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~CopyPool()
  {
    if (!started)
      return;
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (int i = 0; i < N; i++)
      th[i].join();
  }
  // rows x row_bytes from src (pitch sp) to dst (pitch dp), split over the threads
  void copy_rows(uint8_t *dst, size_t dp, const uint8_t *src, size_t sp, size_t row_bytes, size_t rows)
  {
    if (rows * row_bytes < ((size_t)1 << 20))
    {
      for (size_t r = 0; r < rows; r++)
        memcpy(dst + r * dp, src + r * sp, row_bytes);
      return;
    }
    if (rows >= (size_t)N)
      run([=](int part) {
        const size_t r0 = rows * (size_t)part / N, r1 = rows * ((size_t)part + 1) / N;
        if (dp == row_bytes && sp == row_bytes)
          copy_streaming(dst + r0 * dp, src + r0 * sp, (r1 - r0) * row_bytes);
        else
          for (size_t r = r0; r < r1; r++)
            copy_streaming(dst + r * dp, src + r * sp, row_bytes);
        __builtin_ia32_sfence();
      });
    else // few long rows: split each row
      run([=](int part) {
        const size_t b0 = row_bytes * (size_t)part / N, b1 = row_bytes * ((size_t)part + 1) / N;
        for (size_t r = 0; r < rows; r++)
          copy_streaming(dst + r * dp + b0, src + r * sp + b0, b1 - b0);
        __builtin_ia32_sfence();
      });
  }
};

// The pinned staging ring between pageable host memory and the device: NB buffers in flight, each with the event of the
// DMA that last used it.  Uploads: rows are packed into a buffer by the copy threads, then one contiguous DMA.  Downloads:
// one contiguous DMA into a buffer, and when the ring comes round (or at drain) its rows are copied out to their place.
struct Stager
{
  static constexpr int NB = 6;
  static constexpr size_t SB = (size_t)24 << 20;
  void *buf[NB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[NB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool busy[NB] = {false, false, false, false, false, false};
  struct Out
  {
    uint8_t *dst = nullptr; // where the buffer's rows go when its download has landed (nullptr: nothing to do)
    size_t pitch = 0, row_bytes = 0, rows = 0;
  } out[NB];
  int next = 0;
  CopyPool pool;

  hipError_t init()
  {
    for (int i = 0; i < NB; i++)
      if (buf[i] == nullptr)
      {
        hipError_t e = hipHostMalloc(&buf[i], SB, hipHostMallocDefault);
        if (e != hipSuccess)
          return e;
        if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess)
          return e;
      }
    return hipSuccess;
  }
  void release()
  {
    for (int i = 0; i < NB; i++)
    {
      if (buf[i] != nullptr)
        (void)hipHostFree(buf[i]);
      if (ev[i] != nullptr)
        (void)hipEventDestroy(ev[i]);
      buf[i] = nullptr;
      ev[i] = nullptr;
    }
  }
  hipError_t settle(int b) // wait for buffer b's DMA; finish its download
  {
    if (!busy[b])
      return hipSuccess;
    const hipError_t e = hipEventSynchronize(ev[b]);
    busy[b] = false;
    if (e != hipSuccess)
      return e;
    if (out[b].dst != nullptr)
      pool.copy_rows(out[b].dst, out[b].pitch, (const uint8_t *)buf[b], out[b].row_bytes, out[b].row_bytes, out[b].rows);
    out[b].dst = nullptr;
    return hipSuccess;
  }
  hipError_t drain()
  {
    hipError_t first = hipSuccess;
    for (int k = 0; k < NB; k++)
    {
      const hipError_t e = settle((next + k) % NB);
      if (e != hipSuccess && first == hipSuccess)
        first = e;
    }
    return first;
  }
  // host rows (pitch sp) -> contiguous rows at dev
  hipError_t upload(hipStream_t s, void *dev, const uint8_t *src, size_t sp, size_t row_bytes, size_t rows)
  {
    hipError_t e;
    if ((e = init()) != hipSuccess)
      return e;
    if (row_bytes == 0 || rows == 0)
      return hipSuccess;
    if (row_bytes > SB) // a single row longer than a buffer (one long channel): treat the block as a byte string
    {
      if (rows != 1 && sp != row_bytes)
        return hipErrorInvalidValue;
      const size_t total = row_bytes * rows;
      for (size_t o = 0; o < total; o += SB)
        if ((e = upload(s, (uint8_t *)dev + o, src + o, std::min(SB, total - o), std::min(SB, total - o), 1)) != hipSuccess)
          return e;
      return hipSuccess;
    }
    const size_t per = std::max<size_t>(1, SB / row_bytes);
    for (size_t r0 = 0; r0 < rows; r0 += per)
    {
      const size_t n = std::min(per, rows - r0);
      const int b = next;
      next = (next + 1) % NB;
      if ((e = settle(b)) != hipSuccess)
        return e;
      pool.copy_rows((uint8_t *)buf[b], row_bytes, src + r0 * sp, sp, row_bytes, n);
      if ((e = hipMemcpyAsync((uint8_t *)dev + r0 * row_bytes, buf[b], n * row_bytes, hipMemcpyHostToDevice, s)) != hipSuccess)
        return e;
      if ((e = hipEventRecord(ev[b], s)) != hipSuccess)
        return e;
      busy[b] = true;
    }
    return hipSuccess;
  }
  // contiguous rows at dev -> host rows (pitch dp); complete only after drain()
  hipError_t download(hipStream_t s, uint8_t *dst, size_t dp, const void *dev, size_t row_bytes, size_t rows)
  {
    hipError_t e;
    if ((e = init()) != hipSuccess)
      return e;
    if (row_bytes == 0 || rows == 0)
      return hipSuccess;
    if (row_bytes > SB)
    {
      if (rows != 1 && dp != row_bytes)
        return hipErrorInvalidValue;
      const size_t total = row_bytes * rows;
      for (size_t o = 0; o < total; o += SB)
        if ((e = download(s, dst + o, std::min(SB, total - o), (const uint8_t *)dev + o, std::min(SB, total - o), 1)) != hipSuccess)
          return e;
      return hipSuccess;
    }
    const size_t per = std::max<size_t>(1, SB / row_bytes);
    for (size_t r0 = 0; r0 < rows; r0 += per)
    {
      const size_t n = std::min(per, rows - r0);
      const int b = next;
      next = (next + 1) % NB;
      if ((e = settle(b)) != hipSuccess)
        return e;
      if ((e = hipMemcpyAsync(buf[b], (const uint8_t *)dev + r0 * row_bytes, n * row_bytes, hipMemcpyDeviceToHost, s)) != hipSuccess)
        return e;
      if ((e = hipEventRecord(ev[b], s)) != hipSuccess)
        return e;
      busy[b] = true;
      out[b].dst = dst + r0 * dp;
      out[b].pitch = dp;
      out[b].row_bytes = row_bytes;
      out[b].rows = n;
    }
    return hipSuccess;
  }
};

// true when p is host memory the runtime has pinned (hipHostMalloc / hipHostRegister): DMA can use it in place
static bool is_pinned(const void *p)
{
  if (p == nullptr)
    return false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

// Per-chunk small arrays, same layout on the device and in the pinned mirror
struct MetaView
{
  uint64_t *bits, *offsets, *counts;
  int32_t *err;
  static size_t bytes(size_t n)
  {
    return (3 * n + 1) * sizeof(uint64_t) + n * sizeof(int32_t);
  }
  MetaView(void *base, size_t n)
  {
    bits = (uint64_t *)base;
    offsets = bits + n;
    counts = offsets + n + 1;
    err = (int32_t *)(counts + n);
  }
};

struct Slot
{
  hipStream_t s = nullptr;
  GrowDev a, b, meta; // encode: a = samples, then the packed streams; b = slabs.  decode: a = packed streams, b = slabs, c = samples
  GrowDev c;
  GrowPin hmeta, stage;
  // few, long channels: encode uploads the rows in bands and codes every band as it lands (one launch per band, the lanes'
  // state saved in between: EncodeArgs::seg_state); decode downloads in bands beside the running kernel.  The copies'
  // stream, one event per band, the state of the encode launches / the waves' progress words of the decode kernel.
  hipStream_t s2 = nullptr;
  std::vector<hipEvent_t> band_ev;
  hipEvent_t reuse_ev = nullptr;
  GrowDev seg_state;
  GrowPin progress_values;
};

constexpr int MAX_SLOTS = 16;

struct Pipeline
{
  Slot slot[MAX_SLOTS];
  hipStream_t up = nullptr; // decode: the chunks' packed streams go up one after the other on this stream (each at the link's full rate)
  GrowDev redo_slabs, redo_packed, redo_meta;
  GrowPin redo_hmeta;
  Stager stager;
};

// Host rows <-> device, by the faster route: in place when the caller's memory is pinned and the rows are long enough for
// the DMA engine's 2-D copies (they cost about a microsecond per row), through the staging ring otherwise.
constexpr size_t DIRECT_2D_MIN_ROW = (size_t)16 << 10;
static hipError_t rows_to_device(Pipeline *pl, hipStream_t s, void *dev, const uint8_t *src, size_t sp, size_t row_bytes, size_t rows, bool pinned)
{
  if (rows == 0 || row_bytes == 0)
    return hipSuccess;
  if (pinned && sp == row_bytes)
    return hipMemcpyAsync(dev, src, rows * row_bytes, hipMemcpyHostToDevice, s);
  if (pinned && row_bytes >= DIRECT_2D_MIN_ROW)
    return hipMemcpy2DAsync(dev, row_bytes, src, sp, row_bytes, rows, hipMemcpyHostToDevice, s);
  return pl->stager.upload(s, dev, src, sp, row_bytes, rows);
}
static hipError_t rows_to_host(Pipeline *pl, hipStream_t s, uint8_t *dst, size_t dp, const void *dev, size_t row_bytes, size_t rows, bool pinned)
{
  if (rows == 0 || row_bytes == 0)
    return hipSuccess;
  if (pinned && dp == row_bytes)
    return hipMemcpyAsync(dst, dev, rows * row_bytes, hipMemcpyDeviceToHost, s);
  if (pinned && row_bytes >= DIRECT_2D_MIN_ROW)
    return hipMemcpy2DAsync(dst, dp, dev, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, s);
  return pl->stager.download(s, dst, dp, dev, row_bytes, rows);
}

// DEGA_PIPELINE_TRACE=1: host timestamps of the pipeline's stages on stderr (measurement aid)
static bool trace_on()
{
  static const int on = getenv("DEGA_PIPELINE_TRACE") != nullptr ? 1 : 0;
  return on != 0;
}
static double trace_now()
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define TRACE(...) do { if (trace_on()) { fprintf(stderr, "[dega %10.3f ms] ", trace_now()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static void pipeline_destroy(Pipeline *p)
{
  if (p == nullptr)
    return;
  for (Slot &sl : p->slot)
  {
    if (sl.s != nullptr)
      (void)hipStreamDestroy(sl.s);
    if (sl.s2 != nullptr)
      (void)hipStreamDestroy(sl.s2);
    for (hipEvent_t e : sl.band_ev)
      (void)hipEventDestroy(e);
    sl.band_ev.clear();
    if (sl.reuse_ev != nullptr)
      (void)hipEventDestroy(sl.reuse_ev);
    sl.seg_state.release();
    sl.progress_values.release();
    sl.a.release();
    sl.b.release();
    sl.c.release();
    sl.meta.release();
    sl.hmeta.release();
    sl.stage.release();
  }
  if (p->up != nullptr)
    (void)hipStreamDestroy(p->up);
  p->redo_slabs.release();
  p->redo_packed.release();
  p->redo_meta.release();
  p->redo_hmeta.release();
  p->stager.release();
  delete p;
}

static int pipeline_get(dega_hip_ctx *ctx, Pipeline **out)
{
  if (ctx->pipe == nullptr)
    ctx->pipe = new Pipeline();
  *out = ctx->pipe;
  return DEGA_OK;
}

static int slot_stream(dega_hip_ctx *ctx, Slot &sl)
{
  if (sl.s == nullptr)
    HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

static size_t sample_bytes(const Shape &j)
{
  return j.samples == DEGA_SAMPLES_I64 ? 8 : 4;
}

static size_t worst_cap(const Shape &j)
{
  return j.valuesize > 32 ? dega_hip_worst_case_bytes64(j.T) : dega_hip_worst_case_bytes(j.T);
}

// slab bytes per channel for the first attempt: a stream no longer than its samples (plus the coder's tail)
static size_t usual_cap(const Shape &j)
{
  const size_t vbytes = j.valuesize > 32 ? 8 : 4;
  const size_t c = (j.T * vbytes + 64 + 3) & ~(size_t)3;
  return std::min(c, worst_cap(j));
}

struct ChunkPlan
{
  size_t chunk_channels; // multiple of 512 unless the batch is one chunk
  size_t nchunks;
  int nslots;
};

// Chunks: enough of them to overlap copies and kernels (about 64 MiB of samples each, at most MAX_SLOTS at a time), whole
// workgroups of channels, and -- when the batch does not fit the device beside its slabs -- as many slots as fit.
static ChunkPlan plan_chunks(size_t C, size_t bytes_per_channel_in, size_t bytes_per_channel_dev, int want_all_resident, size_t most = 8)
{
  ChunkPlan p;
  const size_t total = C * bytes_per_channel_in;
  size_t n = total / ((size_t)64 << 20);
  n = std::max<size_t>(1, std::min<size_t>(n, most));
  if (getenv("DEGA_PIPELINE_CHUNKS") != nullptr) // measurement knob
    n = std::max(1, atoi(getenv("DEGA_PIPELINE_CHUNKS")));
  if (C <= 512)
    n = 1;
  size_t cc = (C + n - 1) / n;
  if (n > 1)
    cc = std::max<size_t>((cc + 511) / 512 * 512, std::min<size_t>(C, 8192)); // rows of a chunk: 32 KiB or more (copies of narrow rows are slow: two chunks of 4 096 channels took 81 ms where one of 8 192 takes 66)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
    free_b = (size_t)8 << 30;
  const size_t budget = free_b / 10 * 8;
  while (cc > 512 && cc * bytes_per_channel_dev > budget) // a single chunk must fit
    cc = (cc / 2 + 511) / 512 * 512;
  p.chunk_channels = std::max<size_t>(cc, 1);
  p.nchunks = C == 0 ? 0 : (C + p.chunk_channels - 1) / p.chunk_channels;
  size_t fit = std::max<size_t>(1, budget / std::max<size_t>(1, p.chunk_channels * bytes_per_channel_dev));
  p.nslots = (int)std::min<size_t>(std::min<size_t>(fit, MAX_SLOTS), std::max<size_t>(p.nchunks, 1));
  if (want_all_resident && (size_t)p.nslots < p.nchunks)
    p.nslots = -1; // the caller has to split the batch
  return p;
}

// Rows per band when a chunk's samples cross the link in bands (0: in one piece, the usual way): encode codes every band
// with a launch of its own as soon as it has landed -- plain stream order, an event per band; nothing on the device ever
// waits for the host -- and decode sends every band home as soon as all waves have stored it.  Worth it when the channels
// are long (a launch per band has its fixed costs); bands end on multiples of 32 rows (whole 128-byte lines of the device
// array) and are about 32 MiB each.
static size_t band_rows_of(const ChunkPlan &plan, const Shape &j, size_t row_bytes)
{
  size_t band_bytes = (size_t)32 << 20, min_T = 4096;
  if (const char *e = getenv("DEGA_PIPELINE_BAND_BYTES")) // measurement / test knob: 0 = no bands, else the size of a band
  {
    band_bytes = (size_t)strtoull(e, nullptr, 10);
    min_T = 64;
  }
  (void)plan; // (any number of chunks: a chunk's kernels start with its first band, so only the last band's kernel and the
              //  last chunk's download are left over when the last byte has gone up)
  if (band_bytes == 0 || j.T < min_T || row_bytes == 0)
    return 0;
  size_t rows = std::max<size_t>(band_bytes / row_bytes, 32);
  rows = (rows + 31) / 32 * 32;
  return rows >= j.T ? 0 : rows;
}

// ---- encode ------------------------------------------------------------------------------------------------------------

struct EncodeSink
{
  uint8_t *packed = nullptr; // packed mode: streams back to back ...
  size_t packed_cap = 0;
  uint64_t *offsets = nullptr; // ... channel c at packed[offsets[c] .. offsets[c+1]); C + 1 entries
  uint8_t *slabs = nullptr;    // slab mode: channel c at slabs + c * slab_cap
  size_t slab_cap = 0;
  uint64_t *bits = nullptr;
  int32_t *err = nullptr;
};

struct EncChunk
{
  size_t c0 = 0, n = 0;
  int slot = 0;
  uint64_t total = 0; // packed bytes of the chunk
  bool gathered = false, redone = false;
  bool bands = false; // its rows went up in bands, each coded by a launch of its own
  const uint8_t *rows = nullptr; // where the kernels read the chunk's samples: the slot's buffer, or the caller's pinned array in place
  size_t rows_ld = 0;            // ... and its row pitch in samples
  std::vector<uint8_t> redo_bytes; // a chunk that needed worst-case slabs: its packed streams, already on the host
};

struct EncodeRun // state of one device's share of a batch between the two phases of a group call
{
  std::vector<EncChunk> chunks;
  uint64_t total = 0;
};

// the worst-case pass over one chunk, a few channels at a time, synchronous: rare (streams longer than their samples)
static int encode_redo_chunk(dega_hip_ctx *ctx, Pipeline *pl, Slot &sl, const Shape &cj, size_t batch_C, EncChunk &ch, MetaView &hm)
{
  const size_t wc = worst_cap(cj);
  size_t step = std::max<size_t>(64, std::min<size_t>(1024, ((size_t)1 << 30) / std::max<size_t>(wc, 1) / 64 * 64));
  ch.redo_bytes.clear();
  uint64_t running = 0;
  for (size_t j0 = 0; j0 < ch.n; j0 += step)
  {
    const size_t n = std::min(step, ch.n - j0);
    HIP_TRY(ctx, pl->redo_slabs.need(n * wc), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, pl->redo_meta.need(MetaView::bytes(n)), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, pl->redo_hmeta.need(MetaView::bytes(n)), DEGA_ERROR_MEMORY);
    MetaView dm(pl->redo_meta.p, n), rm(pl->redo_hmeta.p, n);
    Shape sj = cj;
    sj.C = n; // columns j0 .. j0+n of the chunk's [T][chunk] image; ld stays the image's pitch
    sj.ld = ch.rows_ld;
    int ret;
    if ((ret = launch_encode(ctx, ch.rows + j0 * sample_bytes(cj), sj, batch_C, (uint8_t *)pl->redo_slabs.p, wc, dm.bits, dm.err, sl.s)) != DEGA_OK)
      return ret;
    hipLaunchKernelGGL(dega_offsets_kernel, dim3(1), dim3(1024), 0, sl.s, dm.bits, n, dm.offsets);
    HIP_TRY(ctx, hipMemcpyAsync(pl->redo_hmeta.p, pl->redo_meta.p, MetaView::bytes(n), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    const uint64_t tot = rm.offsets[n];
    HIP_TRY(ctx, pl->redo_packed.need((size_t)tot + 16), DEGA_ERROR_MEMORY);
    if (tot > 0)
    {
      GatherArgs g{(const uint8_t *)pl->redo_slabs.p, wc, dm.offsets, n, (uint8_t *)pl->redo_packed.p};
      hipLaunchKernelGGL(dega_gather_kernel, dim3((unsigned)((n + WAVES - 1) / WAVES)), dim3(BLOCK), 0, sl.s, g);
      ch.redo_bytes.resize((size_t)(running + tot));
      HIP_TRY(ctx, hipMemcpyAsync(ch.redo_bytes.data() + running, pl->redo_packed.p, (size_t)tot, hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
      HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    }
    for (size_t i = 0; i < n; i++)
    {
      hm.bits[j0 + i] = rm.bits[i];
      hm.err[j0 + i] = rm.err[i];
      hm.offsets[j0 + i] = running + rm.offsets[i];
    }
    running += tot;
  }
  hm.offsets[ch.n] = running;
  ch.redone = true;
  return DEGA_OK;
}

// DEGA_PIPELINE_AHEAD=n: chunks whose first stage is enqueued before the oldest one's second stage (default 2; measurement)
static size_t stages_ahead()
{
  const char *e = getenv("DEGA_PIPELINE_AHEAD");
  const long v = e != nullptr ? atol(e) : 2;
  return (size_t)(v < 1 ? 1 : v);
}

static bool decode_uploads_first() // DEGA_PIPELINE_UPLOADS_FIRST=0: upload and kernel enqueued chunk by chunk (measurement)
{
  const char *e = getenv("DEGA_PIPELINE_UPLOADS_FIRST");
  return e == nullptr || atoi(e) != 0;
}
static size_t decode_stages_ahead() // DEGA_PIPELINE_AHEAD_DECODE=n: the same for a decode call (default: as far as the slots allow)
{
  const char *e = getenv("DEGA_PIPELINE_AHEAD_DECODE");
  const long v = e != nullptr ? atol(e) : MAX_SLOTS;
  return (size_t)(v < 1 ? 1 : v);
}

// Pinned samples are read by the encode kernel where they lie (DEGA_PIPELINE_IN_PLACE=0: through the copy engine instead)
static bool read_in_place()
{
  const char *e = getenv("DEGA_PIPELINE_IN_PLACE");
  return e == nullptr || atoi(e) != 0;
}

// Phase A of one device's share: channels [0, j.C) of `samples` (host, row pitch j.ld).  Uploads, codes and sizes every
// chunk; with `deliver` the packed bytes of each chunk also go out at once (base = running total); without, they stay
// on the device (gathered) for encode_deliver().  bits / err / offsets (relative to this share) are final on return.
static int encode_share(dega_hip_ctx *ctx, const Shape &j, const void *samples, const EncodeSink &sink, bool deliver, EncodeRun &run)
{
  int ret;
  Pipeline *pl;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  const size_t esz = sample_bytes(j);
  const size_t cap = usual_cap(j);
  const size_t dev_per_channel = std::max(j.T * esz + 64, cap) + 16 + cap + 64;
  const ChunkPlan plan = plan_chunks(j.C, j.T * esz, dev_per_channel, deliver ? 0 : 1);
  if (plan.nslots < 0)
    return fail(ctx, DEGA_ERROR_MEMORY, "the device's share of the batch does not fit its memory", hipSuccess);
  run.chunks.assign(plan.nchunks, EncChunk());
  run.total = 0;
  bool out_full = false;
  const bool samples_pinned = is_pinned(samples), packed_pinned = is_pinned(sink.packed);
  const bool in_place = samples_pinned && read_in_place();

  TRACE("encode share: C %zu T %zu, %zu chunks of %zu channels, %d slots", j.C, j.T, plan.nchunks, plan.chunk_channels, plan.nslots);
  auto stage1 = [&](size_t k) -> int {
    EncChunk &ch = run.chunks[k];
    TRACE("chunk %zu stage1 begin", k);
    ch.c0 = k * plan.chunk_channels;
    ch.n = std::min(plan.chunk_channels, j.C - ch.c0);
    ch.slot = (int)(k % (size_t)plan.nslots);
    Slot &sl = pl->slot[ch.slot];
    int r;
    if ((r = slot_stream(ctx, sl)) != DEGA_OK)
      return r;
    HIP_TRY(ctx, sl.a.need(ch.n * std::max(j.T * esz + 64, cap) + 4096), DEGA_ERROR_MEMORY); // the samples, later the packed streams
    HIP_TRY(ctx, sl.b.need(ch.n * cap + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.meta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.hmeta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    MetaView dm(sl.meta.p, ch.n);
    Shape cj = j;
    cj.C = ch.n;
    cj.ld = ch.n;
    const uint8_t *const src = (const uint8_t *)samples + ch.c0 * esz;
    const size_t band_rows = band_rows_of(plan, j, ch.n * esz);
    ch.rows = (const uint8_t *)sl.a.p;
    ch.rows_ld = ch.n;
    void *dev_src = nullptr;
    if (in_place && ch.n == j.ld && hipHostGetDevicePointer(&dev_src, const_cast<uint8_t *>(src), 0) != hipSuccess)
    {
      (void)hipGetLastError(); // (pinned, but not mapped into the device's address space: the copy engine takes it)
      dev_src = nullptr;
    }
    if (dev_src != nullptr)
    {
      // The caller's array is pinned and the chunk is all of its columns: the kernel's filling waves fetch their rows
      // from it themselves (LDS-DMA over the link, 256-byte row segments: 8 192 x 86 400 in 65.2 ms against 66.5 ms
      // with bands through the copy engine) -- no image of the samples on the device, one launch.  Only for whole
      // rows: a chunk of the columns of a wider array is read at 32 - 36 GB/s this way (rows 256 KiB apart: a
      // translation per row), where the copy engine runs at the link's rate (56.8 GB/s, tools/pcie_bench).
      ch.rows = (const uint8_t *)dev_src;
      ch.rows_ld = j.ld;
      cj.ld = j.ld;
      if ((r = launch_encode(ctx, ch.rows, cj, j.C, (uint8_t *)sl.b.p, cap, dm.bits, dm.err, sl.s)) != DEGA_OK)
        return r;
      TRACE("chunk %zu: read in place", k);
    }
    else if (band_rows == 0)
    {
      HIP_TRY(ctx, rows_to_device(pl, sl.s, sl.a.p, src, j.ld * esz, ch.n * esz, j.T, samples_pinned), DEGA_ERROR_LIBRARY_CALL);
      if ((r = launch_encode(ctx, sl.a.p, cj, j.C, (uint8_t *)sl.b.p, cap, dm.bits, dm.err, sl.s)) != DEGA_OK)
        return r;
    }
    else
    {
      // Few, long channels: a channel's serial chain takes the same kernel time however few channels there are, so
      // upload and kernel of the (only) chunk must not take turns.  The rows go up in bands on a second stream, and every
      // band is coded by a launch of its own that waits for nothing but its band's event: each launch takes the lanes'
      // state where the one before left it (EncodeArgs::seg_state), the last one ends the streams.
      if (sl.s2 == nullptr)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.s2, hipStreamNonBlocking), DEGA_ERROR_LIBRARY_CALL);
      if (sl.reuse_ev == nullptr)
        HIP_TRY(ctx, hipEventCreateWithFlags(&sl.reuse_ev, hipEventDisableTiming), DEGA_ERROR_LIBRARY_CALL);
      const size_t nbands = (j.T + band_rows - 1) / band_rows;
      while (sl.band_ev.size() < nbands)
      {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming), DEGA_ERROR_LIBRARY_CALL);
        sl.band_ev.push_back(e);
      }
      HIP_TRY(ctx, sl.seg_state.need((size_t)ENC_STATE_WORDS * ch.n * sizeof(uint32_t)), DEGA_ERROR_MEMORY);
      // the slot's buffers may still be read by what its previous chunk left on the slot's stream (gather, download):
      // the copies start behind that
      HIP_TRY(ctx, hipEventRecord(sl.reuse_ev, sl.s), DEGA_ERROR_LIBRARY_CALL);
      HIP_TRY(ctx, hipStreamWaitEvent(sl.s2, sl.reuse_ev, 0), DEGA_ERROR_LIBRARY_CALL);
      for (size_t b = 0; b < nbands; b++)
      {
        const size_t t0 = b * band_rows, t1 = std::min(j.T, t0 + band_rows);
        uint8_t *const dev_rows = (uint8_t *)sl.a.p + t0 * ch.n * esz;
        HIP_TRY(ctx, rows_to_device(pl, sl.s2, dev_rows, src + t0 * j.ld * esz, j.ld * esz, ch.n * esz, t1 - t0, samples_pinned), DEGA_ERROR_LIBRARY_CALL);
        HIP_TRY(ctx, hipEventRecord(sl.band_ev[b], sl.s2), DEGA_ERROR_LIBRARY_CALL);
        HIP_TRY(ctx, hipStreamWaitEvent(sl.s, sl.band_ev[b], 0), DEGA_ERROR_LIBRARY_CALL);
        Shape bj = cj;
        bj.T = t1 - t0;
        const uint32_t flags = (b > 0 ? ENC_SEG_CONTINUES : 0u) | (b + 1 < nbands ? ENC_SEG_MORE : 0u);
        if ((r = launch_encode(ctx, dev_rows, bj, j.C, (uint8_t *)sl.b.p, cap, dm.bits, dm.err, sl.s, (uint32_t *)sl.seg_state.p, flags)) != DEGA_OK)
          return r;
      }
      TRACE("chunk %zu: %zu bands of %zu rows, one launch each", k, nbands, band_rows);
      ch.bands = true;
    }
    hipLaunchKernelGGL(dega_offsets_kernel, dim3(1), dim3(1024), 0, sl.s, dm.bits, ch.n, dm.offsets);
    HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, hipMemcpyAsync(sl.hmeta.p, sl.meta.p, MetaView::bytes(ch.n), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    TRACE("chunk %zu stage1 enqueued", k);
    return DEGA_OK;
  };

  // sizes of chunk k are on the host: hand out bits / err / offsets, gather its streams, and (deliver) send them home
  auto stage2 = [&](size_t k) -> int {
    EncChunk &ch = run.chunks[k];
    Slot &sl = pl->slot[ch.slot];
    TRACE("chunk %zu stage2 wait", k);
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    TRACE("chunk %zu sizes on the host", k);
    bool too_small = false;
    if (cap < worst_cap(j))
      for (size_t i = 0; i < ch.n && !too_small; i++)
        too_small = hm.err[i] == DEGA_ERROR_MEMORY;
    Shape cj = j;
    cj.C = ch.n;
    cj.ld = ch.n;
    int r;
    if (too_small && (r = encode_redo_chunk(ctx, pl, sl, cj, j.C, ch, hm)) != DEGA_OK)
      return r;
    ch.total = hm.offsets[ch.n];
    const uint64_t base = run.total;
    for (size_t i = 0; i < ch.n; i++)
    {
      sink.bits[ch.c0 + i] = hm.bits[i];
      sink.err[ch.c0 + i] = hm.err[i];
    }
    if (sink.offsets != nullptr)
      for (size_t i = 0; i < ch.n; i++)
        sink.offsets[ch.c0 + i] = base + hm.offsets[i];
    run.total += ch.total;
    if (!ch.redone && ch.total > 0)
    {
      // the samples are no longer needed: their buffer takes the packed streams (it holds max(samples, slabs) bytes)
      GatherArgs g{(const uint8_t *)sl.b.p, cap, dm.offsets, ch.n, (uint8_t *)sl.a.p};
      hipLaunchKernelGGL(dega_gather_kernel, dim3((unsigned)((ch.n + WAVES - 1) / WAVES)), dim3(BLOCK), 0, sl.s, g);
      HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    }
    ch.gathered = true;
    if (!deliver)
      return DEGA_OK;
    if (sink.slabs != nullptr)
    {
      // slab mode: through the pinned stage, then channel by channel to its slab
      const uint8_t *srcb = nullptr;
      if (ch.redone)
        srcb = ch.redo_bytes.data();
      else if (ch.total > 0)
      {
        HIP_TRY(ctx, sl.stage.need((size_t)ch.total), DEGA_ERROR_MEMORY);
        HIP_TRY(ctx, hipMemcpyAsync(sl.stage.p, sl.a.p, (size_t)ch.total, hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
        HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
        srcb = (const uint8_t *)sl.stage.p;
      }
      for (size_t i = 0; i < ch.n; i++)
      {
        const size_t nb = (size_t)(hm.offsets[i + 1] - hm.offsets[i]);
        if (nb > sink.slab_cap && sink.err[ch.c0 + i] == DEGA_OK)
          sink.err[ch.c0 + i] = DEGA_ERROR_MEMORY; // the stream does not fit the caller's slab: its head is there, its length is reported
        if (nb > 0)
          memcpy(sink.slabs + (ch.c0 + i) * sink.slab_cap, srcb + hm.offsets[i], std::min(nb, sink.slab_cap));
      }
      return DEGA_OK;
    }
    if (base + ch.total > sink.packed_cap)
    {
      out_full = true; // keep sizing the rest: the caller learns what it needs
      return DEGA_OK;
    }
    if (out_full || ch.total == 0)
      return DEGA_OK;
    if (ch.redone)
      memcpy(sink.packed + base, ch.redo_bytes.data(), (size_t)ch.total);
    else
      HIP_TRY(ctx, rows_to_host(pl, sl.s, sink.packed + base, (size_t)ch.total, sl.a.p, (size_t)ch.total, 1, packed_pinned), DEGA_ERROR_LIBRARY_CALL);
    return DEGA_OK;
  };

  // chunk k + nslots reuses the slot of chunk k: its second stage has to be enqueued first
  // How far the first stages run ahead of the second ones.  Not as far as the slots would allow: HIP puts the streams
  // on a few hardware queues, and a queue takes its packets in order -- with all eight chunks' first stages enqueued up
  // front, chunk 0's gather and download (enqueued when its sizes were on the host, 7 ms into the call) sat behind the
  // launches of a later chunk that were still waiting for their bands, and went out at 46 ms; every chunk's streams
  // came home after the last upload (trace of 65 536 x 10 800, gpurun_out/e2eprof; 66 ms a call).  Two chunks ahead
  // keep the link busy (enqueueing a chunk takes 0.15 ms) and a download waits for one chunk's launches at most.
  const size_t ahead = std::min<size_t>((size_t)plan.nslots, stages_ahead());
  for (size_t k = 0; k < plan.nchunks + ahead; k++)
  {
    if (k >= ahead && (ret = stage2(k - ahead)) != DEGA_OK)
      return ret;
    if (k < plan.nchunks && (ret = stage1(k)) != DEGA_OK)
      return ret;
  }
  TRACE("all chunks enqueued");
  HIP_TRY(ctx, pl->stager.drain(), DEGA_ERROR_LIBRARY_CALL);
  for (int s = 0; s < plan.nslots; s++)
    if (pl->slot[s].s != nullptr)
      HIP_TRY(ctx, hipStreamSynchronize(pl->slot[s].s), DEGA_ERROR_LIBRARY_CALL);
  TRACE("encode share done");
  if (sink.offsets != nullptr)
    sink.offsets[j.C] = run.total;
  if (deliver && sink.slabs == nullptr && out_full)
    return fail(ctx, DEGA_ERROR_MEMORY, "packed buffer too small: offsets[C] holds the size needed", hipSuccess);
  return DEGA_OK;
}

// Phase B: the packed streams of this device's chunks, still on the device, to packed + base (host)
static int encode_deliver(dega_hip_ctx *ctx, EncodeRun &run, uint8_t *packed, uint64_t base)
{
  Pipeline *pl = ctx->pipe;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  const bool pinned = is_pinned(packed);
  for (EncChunk &ch : run.chunks)
  {
    Slot &sl = pl->slot[ch.slot];
    if (ch.total > 0)
    {
      if (ch.redone)
        memcpy(packed + base, ch.redo_bytes.data(), (size_t)ch.total);
      else
        HIP_TRY(ctx, rows_to_host(pl, sl.s, packed + base, (size_t)ch.total, sl.a.p, (size_t)ch.total, 1, pinned), DEGA_ERROR_LIBRARY_CALL);
    }
    base += ch.total;
  }
  HIP_TRY(ctx, pl->stager.drain(), DEGA_ERROR_LIBRARY_CALL);
  for (EncChunk &ch : run.chunks)
    HIP_TRY(ctx, hipStreamSynchronize(pl->slot[ch.slot].s), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

// ---- decode ------------------------------------------------------------------------------------------------------------

static int decode_share(dega_hip_ctx *ctx, const Shape &j, const uint8_t *packed, const uint64_t *offsets, const uint64_t *bits, void *samples,
                        uint64_t *out_count, int32_t *err)
{
  int ret;
  Pipeline *pl;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  const size_t osz = sample_bytes(j);
  // the longest stream sets the slab size of a chunk; the average one the device bytes per channel
  const uint64_t span = j.C > 0 ? offsets[j.C] - offsets[0] : 0;
  uint64_t longest_all = 0;
  for (size_t c = 0; c < j.C; c++)
    longest_all = std::max<uint64_t>(longest_all, offsets[c + 1] - offsets[c]);
  const size_t dev_per_channel = j.T * osz + (size_t)(span / std::max<size_t>(j.C, 1)) + (size_t)longest_all + 64;
  // At most four chunks: a decode kernel takes its channels' serial time however few they are, the hardware queues run
  // few of them side by side, and the rows go home in bands beside the running kernel anyway (65 536 x 10 800 from
  // pinned memory: 8 chunks 83 ms, 4: 63 ms, 2: 66.5 ms, 1: 66 ms).
  const ChunkPlan plan = plan_chunks(j.C, j.T * osz, dev_per_channel, 0, 4);
  struct DecChunk
  {
    size_t c0, n;
    int slot;
    size_t band_rows; // 0: the samples come home after the kernel; else in bands of rows beside it
    size_t cap;       // slab bytes per channel
  };
  std::vector<DecChunk> chunks(plan.nchunks);
  const bool samples_pinned = is_pinned(samples), packed_pinned = is_pinned(packed);

  // (the slot's stream is idle when stage1a runs -- it synchronises it -- so nothing of the slot's is still in use)
  const bool one_upload_stream = packed_pinned && decode_uploads_first();
  // first stage, part a: the chunk's packed streams go up and are spread into slabs
  auto stage1a = [&](size_t k) -> int {
    DecChunk &ch = chunks[k];
    ch.c0 = k * plan.chunk_channels;
    ch.n = std::min(plan.chunk_channels, j.C - ch.c0);
    ch.slot = (int)(k % (size_t)plan.nslots);
    Slot &sl = pl->slot[ch.slot];
    int r;
    if ((r = slot_stream(ctx, sl)) != DEGA_OK)
      return r;
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL); // the pinned mirror is about to be rewritten
    const uint64_t o0 = offsets[ch.c0], nbytes = offsets[ch.c0 + ch.n] - o0;
    uint64_t longest = 0;
    for (size_t i = 0; i < ch.n; i++)
      longest = std::max<uint64_t>(longest, offsets[ch.c0 + i + 1] - offsets[ch.c0 + i]);
    const size_t cap = ((size_t)longest + 16 + 3) & ~(size_t)3; // room for the decoder's word look-ahead
    ch.cap = cap;
    HIP_TRY(ctx, sl.a.need((size_t)nbytes + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.b.need(ch.n * cap + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.c.need(ch.n * j.T * osz + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.meta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.hmeta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    for (size_t i = 0; i < ch.n; i++)
    {
      hm.bits[i] = bits[ch.c0 + i];
      hm.offsets[i] = offsets[ch.c0 + i] - o0;
    }
    hm.offsets[ch.n] = nbytes;
    // The uploads of all chunks share ONE stream: side by side on their own streams they share the link, every one of
    // them takes as long as all together, and the first kernel starts when the last upload is done (two chunks: 74 ms a
    // call where one chunk takes 66).  The chunk's own stream goes on behind the upload's event.
    hipStream_t us = sl.s;
    if (one_upload_stream)
    {
      if (pl->up == nullptr)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&pl->up, hipStreamNonBlocking), DEGA_ERROR_LIBRARY_CALL);
      if (sl.reuse_ev == nullptr)
        HIP_TRY(ctx, hipEventCreateWithFlags(&sl.reuse_ev, hipEventDisableTiming), DEGA_ERROR_LIBRARY_CALL);
      us = pl->up;
    }
    HIP_TRY(ctx, hipMemcpyAsync(sl.meta.p, sl.hmeta.p, (2 * ch.n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, us), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, rows_to_device(pl, us, sl.a.p, packed + o0, (size_t)nbytes, (size_t)nbytes, 1, packed_pinned), DEGA_ERROR_LIBRARY_CALL);
    {
      GatherArgs g{(const uint8_t *)sl.b.p, cap, dm.offsets, ch.n, (uint8_t *)sl.a.p};
      hipLaunchKernelGGL(dega_scatter_kernel, dim3((unsigned)((ch.n + WAVES - 1) / WAVES)), dim3(BLOCK), 0, us, g);
      HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    }
    if (us != sl.s)
    {
      HIP_TRY(ctx, hipEventRecord(sl.reuse_ev, us), DEGA_ERROR_LIBRARY_CALL);
      HIP_TRY(ctx, hipStreamWaitEvent(sl.s, sl.reuse_ev, 0), DEGA_ERROR_LIBRARY_CALL);
    }
    return DEGA_OK;
  };
  // first stage, part b: the decode kernel
  auto stage1b = [&](size_t k) -> int {
    DecChunk &ch = chunks[k];
    Slot &sl = pl->slot[ch.slot];
    const size_t cap = ch.cap;
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    int r;
    Shape cj = j;
    cj.C = ch.n;
    cj.ld = ch.n;
    // few, long channels (see encode_share): the rows go home in bands while the kernel is still decoding; every wave of 64
    // channels reports the rows it has stored in a word of pinned host memory
    ch.band_rows = out_count == nullptr ? band_rows_of(plan, j, ch.n * osz) : 0;
    uint32_t *reports = nullptr;
    if (ch.band_rows != 0)
    {
      const size_t waves = (ch.n + 63) / 64;
      HIP_TRY(ctx, sl.progress_values.need(sizeof(uint32_t) * waves), DEGA_ERROR_MEMORY);
      reports = (uint32_t *)sl.progress_values.p;
      memset(reports, 0, sizeof(uint32_t) * waves);
      if (sl.s2 == nullptr)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.s2, hipStreamNonBlocking), DEGA_ERROR_LIBRARY_CALL);
    }
    if ((r = launch_decode(ctx, (const uint8_t *)sl.b.p, cap, dm.bits, cj, j.C, sl.c.p, out_count != nullptr ? dm.counts : nullptr, dm.err, sl.s, reports,
                           (uint32_t)ch.band_rows)) != DEGA_OK)
      return r;
    // counts and err come back right behind the kernel
    HIP_TRY(ctx, hipMemcpyAsync(hm.counts, dm.counts, ch.n * sizeof(uint64_t) + ch.n * sizeof(int32_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    return DEGA_OK;
  };
  // the samples of chunk k come home (through the staging ring unless the caller's array is pinned)
  auto stage2 = [&](size_t k) -> int {
    DecChunk &ch = chunks[k];
    Slot &sl = pl->slot[ch.slot];
    if (ch.band_rows == 0)
      HIP_TRY(ctx, rows_to_host(pl, sl.s, (uint8_t *)samples + ch.c0 * osz, j.ld * osz, sl.c.p, ch.n * osz, j.T, samples_pinned), DEGA_ERROR_LIBRARY_CALL);
    else
    {
      const volatile uint32_t *const reports = (const volatile uint32_t *)sl.progress_values.p;
      const size_t waves = (ch.n + 63) / 64;
      bool kernel_over = false;
      for (size_t t0 = 0; t0 < j.T; t0 += ch.band_rows)
      {
        const size_t t1 = std::min(j.T, t0 + ch.band_rows);
        for (;;) // until every wave has stored the band's rows (or the kernel is over: then they are all there)
        {
          uint32_t least = 0xFFFFFFFFu;
          for (size_t w = 0; w < waves; w++)
          {
            const uint32_t v = reports[w];
            least = v < least ? v : least;
          }
          if (least >= t1 || kernel_over)
            break;
          if (hipStreamQuery(sl.s) != hipErrorNotReady)
            kernel_over = true;
          else
            sched_yield();
        }
        HIP_TRY(ctx, rows_to_host(pl, sl.s2, (uint8_t *)samples + (t0 * j.ld + ch.c0) * osz, j.ld * osz, (const uint8_t *)sl.c.p + t0 * ch.n * osz, ch.n * osz,
                                  t1 - t0, samples_pinned),
                DEGA_ERROR_LIBRARY_CALL);
      }
      HIP_TRY(ctx, hipStreamSynchronize(sl.s2), DEGA_ERROR_LIBRARY_CALL);
    }
    HIP_TRY(ctx, pl->stager.drain(), DEGA_ERROR_LIBRARY_CALL); // the slot's buffers are free for its next chunk only after this
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    MetaView hm(sl.hmeta.p, ch.n);
    for (size_t i = 0; i < ch.n; i++)
      err[ch.c0 + i] = hm.err[i];
    if (out_count != nullptr)
      for (size_t i = 0; i < ch.n; i++)
        out_count[ch.c0 + i] = hm.counts[i];
    return DEGA_OK;
  };
  if ((size_t)plan.nslots >= plan.nchunks && decode_uploads_first())
  {
    // Every chunk has a slot of its own: all the uploads are enqueued before the first decode kernel.  HIP puts the
    // streams on a few hardware queues that take their packets in order; with upload and kernel enqueued chunk by chunk,
    // the upload of chunk 4 sat behind the 8 ms kernel of chunk 0 on their queue, chunks 6 and 7 behind that of chunk 4,
    // and the link stood idle for 20 ms of an 80 ms call (trace of 65 536 x 10 800, gpurun_out/e2eprof_dec).
    for (size_t k = 0; k < plan.nchunks; k++)
      if ((ret = stage1a(k)) != DEGA_OK)
        return ret;
    for (size_t k = 0; k < plan.nchunks; k++)
      if ((ret = stage1b(k)) != DEGA_OK)
        return ret;
    for (size_t k = 0; k < plan.nchunks; k++)
      if ((ret = stage2(k)) != DEGA_OK)
        return ret;
    return DEGA_OK;
  }
  const size_t ahead = std::min<size_t>((size_t)plan.nslots, decode_stages_ahead());
  for (size_t k = 0; k < plan.nchunks + ahead; k++)
  {
    if (k >= ahead && (ret = stage2(k - ahead)) != DEGA_OK)
      return ret;
    if (k < plan.nchunks && ((ret = stage1a(k)) != DEGA_OK || (ret = stage1b(k)) != DEGA_OK))
      return ret;
  }
  return DEGA_OK;
}

static int check_packed_input(dega_hip_ctx *ctx, const Shape &j, const uint8_t *packed, const uint64_t *offsets, const uint64_t *bits)
{
  if (offsets == nullptr || bits == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (j.C > 0 && packed == nullptr && offsets[j.C] != offsets[0])
    return DEGA_ERROR_INVALID_VALUE;
  for (size_t c = 0; c < j.C; c++)
  {
    // (bits + 7) / 8 would wrap for lengths near 2^64: compare without the rounding add
    if (offsets[c + 1] < offsets[c] || bits[c] / 8 > offsets[c + 1] - offsets[c] || (bits[c] / 8 == offsets[c + 1] - offsets[c] && (bits[c] & 7) != 0))
      return fail(ctx, DEGA_ERROR_INVALID_VALUE, "offsets must grow and hold ceil(bits / 8) bytes per channel", hipSuccess);
    if (offsets[c + 1] - offsets[c] > ((uint64_t)1 << 29) - 64)
      return fail(ctx, DEGA_ERROR_INVALID_VALUE, "a stream of more than 512 MiB", hipSuccess);
  }
  return DEGA_OK;
}

// ---- the group: one pipeline per device, a host thread each ------------------------------------------------------------------

struct dega_hip_group
{
  std::vector<dega_hip_ctx *> ctx;
  char last_error[320];
};

extern "C" int dega_hip_group_create(const int *devices, int n, dega_hip_group **out)
{
  if (out == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  *out = nullptr;
  std::vector<int> devs;
  if (devices != nullptr && n > 0)
    devs.assign(devices, devices + n);
  else
  {
    // every visible device, or the list in DEGA_DEVICES ("0,2,3"); DEGA_DEVICE (one index) is honoured too
    const char *list = getenv("DEGA_DEVICES");
    const char *one = getenv("DEGA_DEVICE");
    if (list != nullptr && *list != '\0')
    {
      const char *p = list;
      while (*p != '\0')
      {
        char *end;
        const long v = strtol(p, &end, 10);
        if (end == p)
          break;
        devs.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
      }
    }
    else if (one != nullptr && *one != '\0')
      devs.push_back(atoi(one));
    else
      for (int d = 0; d < dega_hip_device_count(); d++)
        devs.push_back(d);
  }
  if (devs.empty())
    return DEGA_ERROR_LIBRARY_INIT; // no GPU: no CPU fallback
  dega_hip_group *g = new dega_hip_group();
  g->last_error[0] = '\0';
  for (int d : devs)
  {
    dega_hip_ctx *c = nullptr;
    const int ret = dega_hip_create(d, &c);
    if (ret != DEGA_OK)
    {
      for (dega_hip_ctx *x : g->ctx)
        dega_hip_destroy(x);
      delete g;
      return ret;
    }
    g->ctx.push_back(c);
  }
  *out = g;
  return DEGA_OK;
}

extern "C" void dega_hip_group_destroy(dega_hip_group *g)
{
  if (g == nullptr)
    return;
  for (dega_hip_ctx *c : g->ctx)
    dega_hip_destroy(c);
  delete g;
}

extern "C" int dega_hip_group_size(const dega_hip_group *g)
{
  return g == nullptr ? 0 : (int)g->ctx.size();
}

extern "C" dega_hip_ctx *dega_hip_group_context(dega_hip_group *g, int i)
{
  return g == nullptr || i < 0 || i >= (int)g->ctx.size() ? nullptr : g->ctx[(size_t)i];
}

extern "C" const char *dega_hip_group_last_error(const dega_hip_group *g)
{
  return g == nullptr ? "no group" : g->last_error;
}

static Shape shape_from_job(const dega_hip_job *job)
{
  return shape_of(job->C, job->T, job->ld, job->adaptive, job->valuesize, job->samples, job->factor);
}

// contiguous channel ranges [c_g, c_g+1), whole 512-channel workgroups where the batch allows
static std::vector<size_t> split_channels(size_t C, size_t G)
{
  std::vector<size_t> cut(G + 1, 0);
  for (size_t g = 1; g < G; g++)
  {
    size_t c = C / G * g + std::min(C % G, g);
    if (C >= G * 1024)
      c = c / 512 * 512;
    cut[g] = c;
  }
  cut[G] = C;
  return cut;
}

extern "C" int dega_hip_split_channels(size_t C, int G, size_t *cuts)
{
  if (G < 1 || cuts == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  const std::vector<size_t> c = split_channels(C, (size_t)G);
  for (size_t g = 0; g <= (size_t)G; g++)
    cuts[g] = c[g];
  return DEGA_OK;
}

static int group_fail(dega_hip_group *g, int code, dega_hip_ctx *c, size_t dev_index)
{
  snprintf(g->last_error, sizeof(g->last_error), "device %d (member %zu): %s", c->device, dev_index, c->last_error);
  return code;
}

static int encode_on_group(dega_hip_group *grp, const Shape &j, const void *samples, const EncodeSink &sink)
{
  if (grp == nullptr || grp->ctx.empty())
    return DEGA_ERROR_LIBRARY_INIT;
  int ret;
  if ((ret = check_job_shape(grp->ctx[0], j, 0)) != DEGA_OK)
    return group_fail(grp, ret, grp->ctx[0], 0);
  if (sink.bits == nullptr || sink.err == nullptr || (sink.slabs == nullptr && (sink.offsets == nullptr || (sink.packed == nullptr && sink.packed_cap != 0))) ||
      (samples == nullptr && j.C * j.T != 0))
    return DEGA_ERROR_INVALID_VALUE;
  const size_t esz = sample_bytes(j);
  // devices that get no channels are left out; a single device delivers as it goes (copies overlap the kernels)
  const size_t G = std::max<size_t>(1, std::min<size_t>(grp->ctx.size(), (j.C + 511) / 512));
  if (G == 1)
  {
    EncodeRun run;
    ret = encode_share(grp->ctx[0], j, samples, sink, true, run);
    return ret == DEGA_OK ? DEGA_OK : group_fail(grp, ret, grp->ctx[0], 0);
  }
  // Rounds: what the devices can hold at once (samples + slabs resident until the sizes in front are known) -- by the
  // member with the least free memory, and with the bytes per channel that encode_share reserves
  size_t free_b = ~(size_t)0;
  for (size_t g = 0; g < G; g++)
  {
    size_t f = 0, t = 0;
    if (hipSetDevice(grp->ctx[g]->device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess)
    {
      (void)hipGetLastError();
      f = (size_t)64 << 30;
    }
    free_b = std::min(free_b, f);
  }
  const size_t per_channel = std::max(j.T * esz + 64, usual_cap(j)) + 16 + usual_cap(j) + 64 + 256;
  size_t round_channels = std::max<size_t>(G * 512, std::min<size_t>(j.C, free_b / 10 * 6 / per_channel * G / 512 * 512));
  uint64_t base = 0;
  bool out_full = false;
  for (size_t r0 = 0; r0 < j.C;)
  {
    const size_t rC = std::min(round_channels, j.C - r0);
    const std::vector<size_t> cut = split_channels(rC, G);
    std::vector<EncodeRun> runs(G);
    std::vector<int> rets(G, DEGA_OK);
    std::vector<std::vector<uint64_t>> rel(G);
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++)
      th.emplace_back([&, g] {
        Shape sj = j;
        sj.C = cut[g + 1] - cut[g];
        rel[g].assign(sj.C + 1, 0);
        EncodeSink ss;
        ss.offsets = rel[g].data();
        ss.bits = sink.bits + r0 + cut[g];
        ss.err = sink.err + r0 + cut[g];
        rets[g] = encode_share(grp->ctx[g], sj, (const uint8_t *)samples + (r0 + cut[g]) * esz, ss, false, runs[g]);
      });
    for (std::thread &t : th)
      t.join();
    {
      // a member that cannot keep its share resident after all (another process took memory meanwhile): a smaller round
      bool too_big = false;
      for (size_t g = 0; g < G; g++)
        too_big = too_big || (rets[g] == DEGA_ERROR_MEMORY && round_channels > G * 512);
      if (too_big)
      {
        round_channels = std::max<size_t>(G * 512, round_channels / 2 / 512 * 512);
        continue;
      }
    }
    for (size_t g = 0; g < G; g++)
      if (rets[g] != DEGA_OK)
        return group_fail(grp, rets[g], grp->ctx[g], g);
    // host-side concatenate: every device's streams go behind those of the devices (and rounds) in front of it
    std::vector<uint64_t> dev_base(G);
    for (size_t g = 0; g < G; g++)
    {
      dev_base[g] = base;
      if (sink.offsets != nullptr)
        for (size_t i = 0; i < cut[g + 1] - cut[g]; i++)
          sink.offsets[r0 + cut[g] + i] = base + rel[g][i];
      base += runs[g].total;
    }
    if (sink.slabs == nullptr && base > sink.packed_cap)
      out_full = true;
    if (out_full)
    {
      r0 += rC;
      continue; // keep sizing
    }
    th.clear();
    std::vector<std::vector<uint8_t>> tmp(G);
    for (size_t g = 0; g < G; g++)
      th.emplace_back([&, g] {
        if (sink.slabs != nullptr)
        {
          tmp[g].resize((size_t)runs[g].total + 1);
          rets[g] = encode_deliver(grp->ctx[g], runs[g], tmp[g].data(), 0);
          for (size_t i = 0; rets[g] == DEGA_OK && i < cut[g + 1] - cut[g]; i++)
          {
            const size_t c = r0 + cut[g] + i, nb = (size_t)(rel[g][i + 1] - rel[g][i]);
            if (nb > sink.slab_cap)
              sink.err[c] = sink.err[c] == DEGA_OK ? DEGA_ERROR_MEMORY : sink.err[c];
            if (nb > 0)
              memcpy(sink.slabs + c * sink.slab_cap, tmp[g].data() + rel[g][i], std::min(nb, sink.slab_cap));
          }
        }
        else
          rets[g] = encode_deliver(grp->ctx[g], runs[g], sink.packed, dev_base[g]);
      });
    for (std::thread &t : th)
      t.join();
    for (size_t g = 0; g < G; g++)
      if (rets[g] != DEGA_OK)
        return group_fail(grp, rets[g], grp->ctx[g], g);
    r0 += rC;
  }
  if (sink.offsets != nullptr)
    sink.offsets[j.C] = base;
  if (out_full)
  {
    snprintf(grp->last_error, sizeof(grp->last_error), "packed buffer too small: offsets[C] holds the size needed");
    return DEGA_ERROR_MEMORY;
  }
  return DEGA_OK;
}

static int decode_on_group(dega_hip_group *grp, const Shape &j, const uint8_t *packed, const uint64_t *offsets, const uint64_t *bits, void *samples,
                           uint64_t *out_count, int32_t *err)
{
  if (grp == nullptr || grp->ctx.empty())
    return DEGA_ERROR_LIBRARY_INIT;
  int ret;
  if ((ret = check_job_shape(grp->ctx[0], j, 0)) != DEGA_OK || (ret = check_packed_input(grp->ctx[0], j, packed, offsets, bits)) != DEGA_OK)
    return group_fail(grp, ret, grp->ctx[0], 0);
  if (err == nullptr || (samples == nullptr && j.C * j.T != 0))
    return DEGA_ERROR_INVALID_VALUE;
  const size_t G = std::max<size_t>(1, std::min<size_t>(grp->ctx.size(), (j.C + 511) / 512));
  const size_t osz = sample_bytes(j);
  const std::vector<size_t> cut = split_channels(j.C, G);
  std::vector<int> rets(G, DEGA_OK);
  auto work = [&](size_t g) {
    Shape sj = j;
    sj.C = cut[g + 1] - cut[g];
    rets[g] = decode_share(grp->ctx[g], sj, packed, offsets + cut[g], bits + cut[g], (uint8_t *)samples + cut[g] * osz,
                           out_count != nullptr ? out_count + cut[g] : nullptr, err + cut[g]);
  };
  if (G == 1)
    work(0);
  else
  {
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++)
      th.emplace_back(work, g);
    for (std::thread &t : th)
      t.join();
  }
  for (size_t g = 0; g < G; g++)
    if (rets[g] != DEGA_OK)
      return group_fail(grp, rets[g], grp->ctx[g], g);
  return DEGA_OK;
}

extern "C" int dega_hip_group_encode(dega_hip_group *grp, const dega_hip_job *job, const void *samples, uint8_t *packed, size_t packed_cap,
                                     uint64_t *offsets, uint64_t *out_bits, int32_t *err)
{
  if (job == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  EncodeSink sink;
  sink.packed = packed;
  sink.packed_cap = packed_cap;
  sink.offsets = offsets;
  sink.bits = out_bits;
  sink.err = err;
  if (offsets != nullptr)
    offsets[0] = 0;
  return encode_on_group(grp, shape_from_job(job), samples, sink);
}

extern "C" int dega_hip_group_decode(dega_hip_group *grp, const dega_hip_job *job, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits,
                                     void *samples, uint64_t *out_count, int32_t *err)
{
  if (job == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return decode_on_group(grp, shape_from_job(job), packed, offsets, in_bits, samples, out_count, err);
}

// ---- single-context forms ----------------------------------------------------------------------------------------------------

static int encode_on_ctx(dega_hip_ctx *ctx, const Shape &j, const void *samples, const EncodeSink &sink)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  dega_hip_group one;
  one.ctx.push_back(ctx);
  one.last_error[0] = '\0';
  return encode_on_group(&one, j, samples, sink);
}

static int decode_on_ctx(dega_hip_ctx *ctx, const Shape &j, const uint8_t *packed, const uint64_t *offsets, const uint64_t *bits, void *samples,
                         uint64_t *out_count, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  dega_hip_group one;
  one.ctx.push_back(ctx);
  one.last_error[0] = '\0';
  return decode_on_group(&one, j, packed, offsets, bits, samples, out_count, err);
}

extern "C" int dega_hip_encode_job_host(dega_hip_ctx *ctx, const dega_hip_job *job, const void *samples, uint8_t *packed, size_t packed_cap,
                                        uint64_t *offsets, uint64_t *out_bits, int32_t *err)
{
  if (job == nullptr || offsets == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  EncodeSink sink;
  sink.packed = packed;
  sink.packed_cap = packed_cap;
  sink.offsets = offsets;
  sink.bits = out_bits;
  sink.err = err;
  offsets[0] = 0;
  return encode_on_ctx(ctx, shape_from_job(job), samples, sink);
}

extern "C" int dega_hip_decode_job_host(dega_hip_ctx *ctx, const dega_hip_job *job, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits,
                                        void *samples, uint64_t *out_count, int32_t *err)
{
  if (job == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return decode_on_ctx(ctx, shape_from_job(job), packed, offsets, in_bits, samples, out_count, err);
}

// slabs in and out: the older, simpler surface on top of the same pipeline
static int encode_slabs(dega_hip_ctx *ctx, const Shape &j, const void *samples, uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (cap % 4 != 0)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "cap must be a multiple of 4", hipSuccess);
  if (out == nullptr && j.C != 0)
    return DEGA_ERROR_INVALID_VALUE;
  EncodeSink sink;
  sink.slabs = out;
  sink.slab_cap = cap;
  sink.bits = out_bits;
  sink.err = err;
  return encode_on_ctx(ctx, j, samples, sink);
}

static int decode_slabs(dega_hip_ctx *ctx, const Shape &j, const uint8_t *in, size_t cap, const uint64_t *in_bits, void *samples, uint64_t *out_count,
                        int32_t *err)
{
  if (ctx == nullptr || in_bits == nullptr || (in == nullptr && j.C != 0))
    return DEGA_ERROR_INVALID_VALUE;
  // pack on the host: only ceil(bits / 8) bytes per channel cross PCIe, not C x cap
  std::vector<uint64_t> offsets(j.C + 1, 0);
  for (size_t c = 0; c < j.C; c++)
  {
    const uint64_t nb = in_bits[c] / 8 + ((in_bits[c] & 7) != 0 ? 1 : 0);
    if (nb > cap)
      return fail(ctx, DEGA_ERROR_INVALID_VALUE, "a stream's bit length exceeds its slab", hipSuccess);
    offsets[c + 1] = offsets[c] + nb;
  }
  std::vector<uint8_t> packed((size_t)offsets[j.C] + 1);
  for (size_t c = 0; c < j.C; c++)
    memcpy(packed.data() + offsets[c], in + c * cap, (size_t)(offsets[c + 1] - offsets[c]));
  return decode_on_ctx(ctx, j, packed.data(), offsets.data(), in_bits, samples, out_count, err);
}

extern "C" int dega_hip_encode_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                    uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return encode_slabs(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), x_tc, out, cap, out_bits, err);
}

extern "C" int dega_hip_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                    int adaptive, int valuesize, int32_t *x_tc, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return decode_slabs(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), in, cap, in_bits, x_tc, nullptr, err);
}

extern "C" int dega_hip_decode_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                        int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape(ctx, C, max_T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return decode_slabs(ctx, shape_of(C, max_T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), in, cap, in_bits, x_tc, out_count, err);
}

extern "C" int dega_hip_encode_f32_host(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int adaptive, int valuesize,
                                        uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  return encode_slabs(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_F32, factor), v_tc, out, cap, out_bits, err);
}

extern "C" int dega_hip_decode_f32_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                        float factor, int adaptive, int valuesize, float *v_tc, int32_t *err)
{
  return decode_slabs(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_F32, factor), in, cap, in_bits, v_tc, nullptr, err);
}

extern "C" int dega_hip_decode_f32_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                            float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err)
{
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return decode_slabs(ctx, shape_of(C, max_T, ld, adaptive, valuesize, DEGA_SAMPLES_F32, factor), in, cap, in_bits, v_tc, out_count, err);
}

extern "C" int dega_hip_encode64_host(dega_hip_ctx *ctx, const int64_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                      uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape64(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return encode_slabs(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I64), x_tc, out, cap, out_bits, err);
}

extern "C" int dega_hip_decode64_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                          int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape64(ctx, C, max_T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return decode_slabs(ctx, shape_of(C, max_T, ld, adaptive, valuesize, DEGA_SAMPLES_I64), in, cap, in_bits, x_tc, out_count, err);
}

extern "C" int dega_hip_encode_packed_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                           uint8_t *packed, size_t packed_cap, uint64_t *offsets, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  if (offsets == nullptr || out_bits == nullptr || err == nullptr || (packed == nullptr && packed_cap != 0))
    return DEGA_ERROR_INVALID_VALUE;
  EncodeSink sink;
  sink.packed = packed;
  sink.packed_cap = packed_cap;
  sink.offsets = offsets;
  sink.bits = out_bits;
  sink.err = err;
  offsets[0] = 0;
  return encode_on_ctx(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), x_tc, sink);
}

extern "C" int dega_hip_decode_packed_host(dega_hip_ctx *ctx, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, size_t T,
                                           size_t ld, int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if (ctx == nullptr || offsets == nullptr || in_bits == nullptr || x_tc == nullptr || err == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  return decode_on_ctx(ctx, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), packed, offsets, in_bits, x_tc, out_count, err);
}

// ---- pinned host memory for callers that can place their samples there (copies then run at link speed without staging) ---
extern "C" void *dega_hip_pinned_alloc(size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes > 0 ? bytes : 1, hipHostMallocDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

extern "C" void dega_hip_pinned_free(void *p)
{
  if (p != nullptr)
    (void)hipHostFree(p);
}

// ---- LZMH, host pointers: context-owned buffers, no per-call allocation --------------------------------------------------------

extern "C" int dega_hip_lzmh_encode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out,
                                         size_t cap, uint64_t *out_bits, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  Pipeline *pl;
  int ret;
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  Slot &sl = pl->slot[0];
  if ((ret = slot_stream(ctx, sl)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, sl.a.need(C * stride + 64), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, sl.b.need(C * cap + 64), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, sl.meta.need(MetaView::bytes(C)), DEGA_ERROR_MEMORY);
  MetaView dm(sl.meta.p, C);
  HIP_TRY(ctx, hipMemcpyAsync(sl.a.p, in, C * stride, hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpyAsync(dm.counts, in_len, C * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_lzmh_encode_dev(ctx, (const uint8_t *)sl.a.p, stride, dm.counts, C, (uint8_t *)sl.b.p, cap, dm.bits, dm.err, sl.s)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipMemcpyAsync(out_bits, dm.bits, C * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpyAsync(err, dm.err, C * sizeof(int32_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
  // only the bytes each stream holds come back (whole 16-byte groups, as the kernel stores them)
  uint64_t longest = 0;
  for (size_t c = 0; c < C; c++)
    longest = std::max(longest, std::min<uint64_t>(cap, ((out_bits[c] + 7) / 8 + 15) / 16 * 16));
  if (longest > 0)
    HIP_TRY(ctx, hipMemcpy2DAsync(out, cap, sl.b.p, cap, (size_t)longest, C, hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_lzmh_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out,
                                         size_t stride, uint64_t *out_len, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  Pipeline *pl;
  int ret;
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  Slot &sl = pl->slot[0];
  if ((ret = slot_stream(ctx, sl)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, sl.a.need(C * cap + 64), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, sl.b.need(C * stride + 64), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, sl.meta.need(MetaView::bytes(C)), DEGA_ERROR_MEMORY);
  MetaView dm(sl.meta.p, C);
  HIP_TRY(ctx, hipMemcpyAsync(sl.a.p, in, C * cap, hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpyAsync(dm.bits, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_lzmh_decode_dev(ctx, (const uint8_t *)sl.a.p, cap, dm.bits, C, (uint8_t *)sl.b.p, stride, dm.counts, dm.err, sl.s)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipMemcpyAsync(out, sl.b.p, C * stride, hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpyAsync(out_len, dm.counts, C * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpyAsync(err, dm.err, C * sizeof(int32_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

// ---- LZMH through the pipeline: chunks of channels on their own streams, packed streams out / in, and the group -----------
// A channel's text is a row of `stride` bytes ([C][stride], in_len[c] of them meaningful), so a range of channels is one
// contiguous block: upload -> encode kernel -> offsets -> gather -> download of the packed stream bytes per chunk, the
// chunks overlapping on their streams exactly as the DEGA path's do.  Channels are independent (every stream starts from
// an empty history: lzmh.c:139-143,396), so a group gives every device a contiguous range and the host concatenates.

static ChunkPlan lzmh_plan(size_t C, size_t stride, size_t cap)
{
  ChunkPlan p = plan_chunks(C, stride, 2 * stride + 2 * cap + 64, 0);
  // (whole 256-channel workgroups per chunk where the batch allows: plan_chunks rounds to 512)
  return p;
}

static int lzmh_encode_share(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *packed, size_t packed_cap,
                             uint64_t *offsets, uint64_t *out_bits, int32_t *err, uint64_t *total_out)
{
  int ret;
  Pipeline *pl;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  const size_t cap = dega_hip_lzmh_worst_case_bytes(stride);
  const ChunkPlan plan = lzmh_plan(C, stride, cap);
  const bool in_pinned = is_pinned(in), out_pinned = is_pinned(packed);
  struct Ch
  {
    size_t c0, n;
    int slot;
  };
  std::vector<Ch> chunks(plan.nchunks);
  uint64_t running = 0;
  bool out_full = false;
  auto stage1 = [&](size_t k) -> int {
    Ch &ch = chunks[k];
    ch.c0 = k * plan.chunk_channels;
    ch.n = std::min(plan.chunk_channels, C - ch.c0);
    ch.slot = (int)(k % (size_t)plan.nslots);
    Slot &sl = pl->slot[ch.slot];
    int r;
    if ((r = slot_stream(ctx, sl)) != DEGA_OK)
      return r;
    HIP_TRY(ctx, sl.a.need(ch.n * stride + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.b.need(ch.n * cap + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.meta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.hmeta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    memcpy(hm.counts, in_len + ch.c0, ch.n * sizeof(uint64_t));
    HIP_TRY(ctx, hipMemcpyAsync(dm.counts, hm.counts, ch.n * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, rows_to_device(pl, sl.s, sl.a.p, in + ch.c0 * stride, ch.n * stride, ch.n * stride, 1, in_pinned), DEGA_ERROR_LIBRARY_CALL);
    if ((r = dega_hip_lzmh_encode_dev(ctx, (const uint8_t *)sl.a.p, stride, dm.counts, ch.n, (uint8_t *)sl.b.p, cap, dm.bits, dm.err, sl.s)) != DEGA_OK)
      return r;
    hipLaunchKernelGGL(dega_offsets_kernel, dim3(1), dim3(1024), 0, sl.s, dm.bits, ch.n, dm.offsets);
    HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, hipMemcpyAsync(sl.hmeta.p, sl.meta.p, (2 * ch.n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, hipMemcpyAsync(hm.err, dm.err, ch.n * sizeof(int32_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    return DEGA_OK;
  };
  auto stage2 = [&](size_t k) -> int {
    Ch &ch = chunks[k];
    Slot &sl = pl->slot[ch.slot];
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    const uint64_t tot = hm.offsets[ch.n];
    for (size_t i = 0; i < ch.n; i++)
    {
      out_bits[ch.c0 + i] = hm.bits[i];
      err[ch.c0 + i] = hm.err[i];
      offsets[ch.c0 + i] = running + hm.offsets[i];
    }
    const uint64_t base = running;
    running += tot;
    if (running > packed_cap)
      out_full = true; // keep sizing: the caller learns what it needs
    if (out_full || tot == 0)
      return DEGA_OK;
    HIP_TRY(ctx, sl.c.need((size_t)tot + 64), DEGA_ERROR_MEMORY);
    GatherArgs g{(const uint8_t *)sl.b.p, cap, dm.offsets, ch.n, (uint8_t *)sl.c.p};
    hipLaunchKernelGGL(dega_gather_kernel, dim3((unsigned)((ch.n + WAVES - 1) / WAVES)), dim3(BLOCK), 0, sl.s, g);
    HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, rows_to_host(pl, sl.s, packed + base, (size_t)tot, sl.c.p, (size_t)tot, 1, out_pinned), DEGA_ERROR_LIBRARY_CALL);
    return DEGA_OK;
  };
  for (size_t k = 0; k < plan.nchunks + (size_t)plan.nslots; k++)
  {
    if (k >= (size_t)plan.nslots && (ret = stage2(k - (size_t)plan.nslots)) != DEGA_OK)
      return ret;
    if (k < plan.nchunks && (ret = stage1(k)) != DEGA_OK)
      return ret;
  }
  HIP_TRY(ctx, pl->stager.drain(), DEGA_ERROR_LIBRARY_CALL);
  for (int s = 0; s < plan.nslots; s++)
    if (pl->slot[s].s != nullptr)
      HIP_TRY(ctx, hipStreamSynchronize(pl->slot[s].s), DEGA_ERROR_LIBRARY_CALL);
  offsets[C] = running;
  *total_out = running;
  if (out_full)
    return fail(ctx, DEGA_ERROR_MEMORY, "packed buffer too small: offsets[C] holds the size needed", hipSuccess);
  return DEGA_OK;
}

static int lzmh_decode_share(dega_hip_ctx *ctx, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, uint8_t *out, size_t stride,
                             uint64_t *out_len, int32_t *err)
{
  int ret;
  Pipeline *pl;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = pipeline_get(ctx, &pl)) != DEGA_OK)
    return ret;
  uint64_t longest_all = 0;
  for (size_t c = 0; c < C; c++)
    longest_all = std::max<uint64_t>(longest_all, offsets[c + 1] - offsets[c]);
  const ChunkPlan plan = lzmh_plan(C, stride, (size_t)longest_all + 32);
  const bool in_pinned = is_pinned(packed), out_pinned = is_pinned(out);
  struct Ch
  {
    size_t c0, n;
    int slot;
  };
  std::vector<Ch> chunks(plan.nchunks);
  auto stage1 = [&](size_t k) -> int {
    Ch &ch = chunks[k];
    ch.c0 = k * plan.chunk_channels;
    ch.n = std::min(plan.chunk_channels, C - ch.c0);
    ch.slot = (int)(k % (size_t)plan.nslots);
    Slot &sl = pl->slot[ch.slot];
    int r;
    if ((r = slot_stream(ctx, sl)) != DEGA_OK)
      return r;
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL); // the pinned mirror is about to be rewritten
    const uint64_t o0 = offsets[ch.c0], nbytes = offsets[ch.c0 + ch.n] - o0;
    uint64_t longest = 0;
    for (size_t i = 0; i < ch.n; i++)
      longest = std::max<uint64_t>(longest, offsets[ch.c0 + i + 1] - offsets[ch.c0 + i]);
    const size_t cap = ((size_t)longest + 16 + 3) & ~(size_t)3;
    HIP_TRY(ctx, sl.a.need((size_t)nbytes + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.b.need(ch.n * cap + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.c.need(ch.n * stride + 64), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.meta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    HIP_TRY(ctx, sl.hmeta.need(MetaView::bytes(ch.n)), DEGA_ERROR_MEMORY);
    MetaView hm(sl.hmeta.p, ch.n), dm(sl.meta.p, ch.n);
    for (size_t i = 0; i < ch.n; i++)
    {
      hm.bits[i] = in_bits[ch.c0 + i];
      hm.offsets[i] = offsets[ch.c0 + i] - o0;
    }
    hm.offsets[ch.n] = nbytes;
    HIP_TRY(ctx, hipMemcpyAsync(sl.meta.p, sl.hmeta.p, (2 * ch.n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, rows_to_device(pl, sl.s, sl.a.p, packed + o0, (size_t)nbytes, (size_t)nbytes, 1, in_pinned), DEGA_ERROR_LIBRARY_CALL);
    GatherArgs g{(const uint8_t *)sl.b.p, cap, dm.offsets, ch.n, (uint8_t *)sl.a.p};
    hipLaunchKernelGGL(dega_scatter_kernel, dim3((unsigned)((ch.n + WAVES - 1) / WAVES)), dim3(BLOCK), 0, sl.s, g);
    HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
    if ((r = dega_hip_lzmh_decode_dev(ctx, (const uint8_t *)sl.b.p, cap, dm.bits, ch.n, (uint8_t *)sl.c.p, stride, dm.counts, dm.err, sl.s)) != DEGA_OK)
      return r;
    HIP_TRY(ctx, hipMemcpyAsync(hm.counts, dm.counts, ch.n * sizeof(uint64_t) + ch.n * sizeof(int32_t), hipMemcpyDeviceToHost, sl.s), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, rows_to_host(pl, sl.s, out + ch.c0 * stride, ch.n * stride, sl.c.p, ch.n * stride, 1, out_pinned), DEGA_ERROR_LIBRARY_CALL);
    return DEGA_OK;
  };
  auto stage2 = [&](size_t k) -> int {
    Ch &ch = chunks[k];
    Slot &sl = pl->slot[ch.slot];
    HIP_TRY(ctx, pl->stager.drain(), DEGA_ERROR_LIBRARY_CALL);
    HIP_TRY(ctx, hipStreamSynchronize(sl.s), DEGA_ERROR_LIBRARY_CALL);
    MetaView hm(sl.hmeta.p, ch.n);
    for (size_t i = 0; i < ch.n; i++)
    {
      out_len[ch.c0 + i] = hm.counts[i];
      err[ch.c0 + i] = hm.err[i];
    }
    return DEGA_OK;
  };
  for (size_t k = 0; k < plan.nchunks + (size_t)plan.nslots; k++)
  {
    if (k >= (size_t)plan.nslots && (ret = stage2(k - (size_t)plan.nslots)) != DEGA_OK)
      return ret;
    if (k < plan.nchunks && (ret = stage1(k)) != DEGA_OK)
      return ret;
  }
  return DEGA_OK;
}

static int lzmh_check(dega_hip_group *grp, size_t stride, bool encode)
{
  if (grp == nullptr || grp->ctx.empty())
    return DEGA_ERROR_LIBRARY_INIT;
  if (encode ? (stride == 0 || (stride & 15u) != 0 || stride > 0x7FFFFFF0u) : (stride < 8 || (stride & 7u) != 0))
  {
    snprintf(grp->last_error, sizeof(grp->last_error), encode ? "lzmh encode: stride must be a multiple of 16" : "lzmh decode: stride must be a multiple of 8");
    return DEGA_ERROR_INVALID_VALUE;
  }
  return DEGA_OK;
}

extern "C" int dega_hip_group_lzmh_encode(dega_hip_group *grp, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *packed,
                                          size_t packed_cap, uint64_t *offsets, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = lzmh_check(grp, stride, true)) != DEGA_OK)
    return ret;
  if (offsets == nullptr || out_bits == nullptr || err == nullptr || in_len == nullptr || (in == nullptr && C != 0) || (packed == nullptr && packed_cap != 0))
    return DEGA_ERROR_INVALID_VALUE;
  offsets[0] = 0;
  if (C == 0)
    return DEGA_OK;
  const size_t G = std::max<size_t>(1, std::min<size_t>(grp->ctx.size(), (C + 255) / 256));
  uint64_t total = 0;
  if (G == 1)
  {
    ret = lzmh_encode_share(grp->ctx[0], in, stride, in_len, C, packed, packed_cap, offsets, out_bits, err, &total);
    return ret == DEGA_OK ? DEGA_OK : group_fail(grp, ret, grp->ctx[0], 0);
  }
  // every device codes its range of channels into a buffer of its own; the host puts them behind one another
  std::vector<size_t> cut(G + 1, 0);
  for (size_t g = 1; g < G; g++)
    cut[g] = (C / G * g + std::min(C % G, g)) / 256 * 256;
  cut[G] = C;
  std::vector<std::vector<uint8_t>> tmp(G);
  std::vector<std::vector<uint64_t>> rel(G);
  std::vector<uint64_t> tot(G, 0);
  std::vector<int> rets(G, DEGA_OK);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; g++)
    th.emplace_back([&, g] {
      const size_t n = cut[g + 1] - cut[g];
      uint64_t text = 0;
      for (size_t i = 0; i < n; i++)
        text += in_len[cut[g] + i];
      tmp[g].resize((size_t)(text + text / 4 + 64 * n + 64)); // a stream is at most 10 bits per byte of text
      rel[g].assign(n + 1, 0);
      rets[g] = n == 0 ? DEGA_OK
                       : lzmh_encode_share(grp->ctx[g], in + cut[g] * stride, stride, in_len + cut[g], n, tmp[g].data(), tmp[g].size(), rel[g].data(),
                                           out_bits + cut[g], err + cut[g], &tot[g]);
    });
  for (std::thread &t : th)
    t.join();
  for (size_t g = 0; g < G; g++)
    if (rets[g] != DEGA_OK)
      return group_fail(grp, rets[g], grp->ctx[g], g);
  uint64_t base = 0;
  for (size_t g = 0; g < G; g++)
  {
    for (size_t i = 0; i < cut[g + 1] - cut[g]; i++)
      offsets[cut[g] + i] = base + rel[g][i];
    if (base + tot[g] <= packed_cap && tot[g] > 0)
      memcpy(packed + base, tmp[g].data(), (size_t)tot[g]);
    base += tot[g];
  }
  offsets[C] = base;
  if (base > packed_cap)
  {
    snprintf(grp->last_error, sizeof(grp->last_error), "packed buffer too small: offsets[C] holds the size needed");
    return DEGA_ERROR_MEMORY;
  }
  return DEGA_OK;
}

extern "C" int dega_hip_group_lzmh_decode(dega_hip_group *grp, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, uint8_t *out,
                                          size_t stride, uint64_t *out_len, int32_t *err)
{
  int ret;
  if ((ret = lzmh_check(grp, stride, false)) != DEGA_OK)
    return ret;
  if (offsets == nullptr || in_bits == nullptr || out_len == nullptr || err == nullptr || (out == nullptr && C != 0))
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  for (size_t c = 0; c < C; c++)
    if (offsets[c + 1] < offsets[c] || in_bits[c] / 8 > offsets[c + 1] - offsets[c] || (in_bits[c] / 8 == offsets[c + 1] - offsets[c] && (in_bits[c] & 7) != 0) ||
        offsets[c + 1] - offsets[c] > ((uint64_t)1 << 31))
    {
      snprintf(grp->last_error, sizeof(grp->last_error), "offsets must grow and hold ceil(bits / 8) bytes per channel");
      return DEGA_ERROR_INVALID_VALUE;
    }
  const size_t G = std::max<size_t>(1, std::min<size_t>(grp->ctx.size(), (C + 255) / 256));
  std::vector<size_t> cut(G + 1, 0);
  for (size_t g = 1; g < G; g++)
    cut[g] = (C / G * g + std::min(C % G, g)) / 256 * 256;
  cut[G] = C;
  std::vector<int> rets(G, DEGA_OK);
  auto work = [&](size_t g) {
    const size_t n = cut[g + 1] - cut[g];
    if (n != 0)
      rets[g] = lzmh_decode_share(grp->ctx[g], packed, offsets + cut[g], in_bits + cut[g], n, out + cut[g] * stride, stride, out_len + cut[g], err + cut[g]);
  };
  if (G == 1)
    work(0);
  else
  {
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++)
      th.emplace_back(work, g);
    for (std::thread &t : th)
      t.join();
  }
  for (size_t g = 0; g < G; g++)
    if (rets[g] != DEGA_OK)
      return group_fail(grp, rets[g], grp->ctx[g], g);
  return DEGA_OK;
}
