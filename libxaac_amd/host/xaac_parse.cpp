/*
 * xaac_parse.cpp -- C ABI of the host-side bitstream front end (include/xaac_parse.h): ADTS framing around the AAC-LC
 * syntax decoder of aac_core.cpp.
 */
#define XAAC_PARSE_NO_SIZED_MACROS /* this file defines the symbols of both generations */
#include "../../include/xaac_parse.h"
#include <stddef.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <limits.h>
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <linux/futex.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "aac_core.h"
#include "sbr_side.h"

static const int32_t k_sample_rate[12] = {96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000};

struct xaac_parser {
  int sr_index;
  XhCoreState core;
  XhElement el;
  int sbr_ready, sampling_rate, esbr;
  XsDecoder sbr;
  /* an ADTS frame with several raw data blocks (api.c:2909-2925): the blocks of the frame still to come, the bytes of the
     frame behind the last delivered block, and whether a 16-bit CRC follows every block (ISO/IEC 13818-7 adts_frame():
     raw_data_block() + crc_check per block when protection_absent == 0; headerdecode.c:356-362 reads the positions).
     DELIBERATELY NOT THE REFERENCE'S BEHAVIOUR for protected multi-block frames: api.c:3760-3767 means to skip that word,
     but its `adts` struct (api.c:2626, zero-initialised per call) is only filled by the call that reads a header, so in the
     follow-up calls no_raw_data_blocks reads 0, the CRC is never skipped and blocks 2..N of a protected multi-block frame are
     misparsed there.  This parser follows the syntax; parity with the reference is claimed (and tested against the real
     decoder) for the unprotected layout only. */
  int blocks_left, block_crc;
  size_t frame_left;
};

/* A small persistent team for xaac_parse_batch_run.  Workers wait for the next call on a generation counter: a short spin
   (batches of a running decoder follow each other within microseconds), then asleep in the kernel on a futex -- a host
   that also drives a GPU must not have its cores spun on by idle parser threads, and a wake-up must not pass a mutex
   from thread to thread (with a condition variable the team's wake-up alone cost more than the parsing from 128 threads
   on).  Items are taken in chunks from a shared counter and the caller works along.  One batch call at a time (calls
   from several threads are serialised). */
namespace {
inline void futex_wait(std::atomic<uint32_t> *a, uint32_t expected) {
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t> *a) {
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
}
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
/* CPUs this process can run on at once: the scheduler affinity mask, cut by the cgroup CPU quota (v2 cpu.max, v1
   cpu.cfs_quota_us / cpu.cfs_period_us) when there is one; 0 = unknown */
int usable_cpus() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  double quota = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    long long qv = 0, period = 0;
    if (fscanf(g, "%lld", &qv) == 1 && qv > 0) {
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(h, "%lld", &period) == 1 && period > 0) quota = (double)qv / (double)period;
        fclose(h);
      }
    }
    fclose(g);
  }
  if (quota > 0) {
    const int q = (int)(quota + 0.5) < 1 ? 1 : (int)(quota + 0.5);
    if (n == 0 || q < n) n = q;
  }
  return n;
}
class Team {
 public:
  /* One job at a time: acquire() ... release() bracket it (a binary semaphore on a futex word rather than a mutex, because
     the asynchronous pair xaac_parse_batch_start / _wait may take and give it back on different threads). */
  void acquire() {
    for (;;) {
      uint32_t free_word = 0;
      if (busy_.compare_exchange_strong(free_word, 1, std::memory_order_acquire)) return;
      futex_wait(&busy_, 1);
    }
  }
  void release() {
    busy_.store(0, std::memory_order_release);
    futex_wake_all(&busy_);
  }
  /* between acquire() and release(): hands `items` to `threads` threads -- the team's workers, and the caller itself as one
     of them if it is going to call join(true) */
  void launch(int items, int threads, std::function<void(int)> fn, bool caller_works) {
    if (threads < 1) threads = 1;
    const int helpers = caller_works ? threads - 1 : threads;
    grow(helpers);
    /* Every worker of the team -- not only the `helpers` that take items -- acknowledges every generation through
       pending_: a worker reads fn_ / items_ / active_ only between seeing the new generation and its decrement, and join()
       does not return (so the next launch() cannot rewrite those fields) before all decrements. */
    fn_ = std::move(fn), items_ = items, active_ = helpers;
    t_launch_ = std::chrono::steady_clock::now();
    next_.store(0, std::memory_order_relaxed);
    busy_ns_.store(0, std::memory_order_relaxed);
    pending_.store((uint32_t)workers_.size(), std::memory_order_relaxed);
    generation_.fetch_add(1, std::memory_order_release); /* publishes the job */
    if (!workers_.empty()) futex_wake_all(&generation_);
  }
  void join(bool caller_works) {
    if (caller_works) work();
    for (int spin = 0;;) { /* every worker has seen this generation and is done with the job's fields */
      const uint32_t p = pending_.load(std::memory_order_acquire);
      if (p == 0) break;
      if (++spin < kSpin) cpu_relax();
      else futex_wait(&pending_, p);
    }
    fn_ = nullptr;
  }
  /* after join(): from launch() until the last helper ran out of items */
  double helpers_busy_seconds() const { return 1e-9 * (double)busy_ns_.load(std::memory_order_relaxed); }
  ~Team() {
    quit_.store(true, std::memory_order_release);
    generation_.fetch_add(1, std::memory_order_release);
    futex_wake_all(&generation_);
    for (auto &t : workers_) t.join();
  }

 private:
  void work() {
    for (;;) {
      const int first = next_.fetch_add(kChunk, std::memory_order_relaxed);
      if (first >= items_) return;
      const int last = first + kChunk < items_ ? first + kChunk : items_;
      for (int i = first; i < last; i++) fn_(i);
    }
  }
  void grow(int n) {
    while ((int)workers_.size() < n) {
      const int id = (int)workers_.size();
      const uint32_t born = generation_.load(std::memory_order_acquire);
      workers_.emplace_back([this, id, born] {
        uint32_t seen = born;
        for (;;) {
          uint32_t g;
          for (int spin = 0;;) {
            g = generation_.load(std::memory_order_acquire);
            if (g != seen) break;
            if (++spin < kSpin) cpu_relax();
            else futex_wait(&generation_, seen);
          }
          if (quit_.load(std::memory_order_acquire)) return;
          seen = g;
          if (id < active_) { /* else: this call uses fewer threads than the team has; acknowledge only */
            work();
            const int64_t t = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_launch_).count();
            int64_t seen_t = busy_ns_.load(std::memory_order_relaxed);
            while (t > seen_t && !busy_ns_.compare_exchange_weak(seen_t, t, std::memory_order_relaxed)) {
            }
          }
          if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) futex_wake_all(&pending_);
        }
      });
    }
  }
  static constexpr int kChunk = 4;
  static constexpr int kSpin = 4000; /* a few tens of microseconds */
  std::vector<std::thread> workers_;
  std::function<void(int)> fn_;
  std::atomic<int> next_{0};
  std::atomic<uint32_t> generation_{0}, pending_{0}, busy_{0};
  std::atomic<bool> quit_{false};
  std::atomic<int64_t> busy_ns_{0};
  std::chrono::steady_clock::time_point t_launch_;
  int items_ = 0, active_ = 0;
};
Team &team() {
  static Team t;
  return t;
}
}  // namespace

extern "C" {

int32_t xaac_parser_create(xaac_parser **p) {
  if (!p) return XAAC_PARSE_ERR_SYNTAX;
  xaac_parser *q = new (std::nothrow) xaac_parser;
  if (!q) return XAAC_PARSE_ERR_UNSUPPORTED;
  memset(static_cast<void *>(q), 0, sizeof(*q));
  q->sr_index = -1;
  *p = q;
  return XAAC_PARSE_OK;
}

void xaac_parser_destroy(xaac_parser *p) { delete p; }

int32_t xaac_adts_parse_header(const uint8_t *data, size_t n, xaac_adts_header *h) { /* headerdecode.c:316-368, :838-849 */
  if (n < 7) return XAAC_PARSE_NEED_DATA;
  XhBits br(data, n);
  if (br.get(12) != 0xfff) return XAAC_PARSE_ERR_SYNC;
  h->id = br.get1();
  h->layer = (int32_t)br.get(2);
  h->protection_absent = br.get1();
  h->profile = (int32_t)br.get(2) + 1;
  h->sr_index = (int32_t)br.get(4);
  br.get(1); /* private_bit */
  h->channel_config = (int32_t)br.get(3);
  br.get(4); /* original_copy, home, copyright_identification_bit, copyright_identification_start */
  h->frame_bytes = (int32_t)br.get(13);
  br.get(11); /* adts_buffer_fullness */
  h->raw_blocks = (int32_t)br.get(2);
  h->header_bytes = h->protection_absent ? 7 : 9 + 2 * h->raw_blocks;
  if (h->profile != 2 || h->sr_index > 11 || h->layer != 0 || h->frame_bytes < 8) return XAAC_PARSE_ERR_HEADER;
  h->sampling_rate = k_sample_rate[h->sr_index];
  if (n < (size_t)h->header_bytes) return XAAC_PARSE_NEED_DATA;
  return XAAC_PARSE_OK;
}

static int32_t tools_of(const XhElement &el) {
  int32_t tools = 0;
  if (el.n_ch == 2 && el.common_window) /* (ms_used is cleared per element and written inside these bounds only) */
    for (int g = 0; g < el.ch[0].ics.num_groups; g++)
      for (int sfb = 0; sfb < el.ch[0].ics.max_sfb; sfb++)
        if (el.ms_used[g][sfb]) tools |= XAAC_TOOL_MS;
  for (int c = 0; c < el.n_ch; c++) {
    const XhChannel &ch = el.ch[c];
    if (ch.pns_active) tools |= XAAC_TOOL_PNS;
    if (ch.tns.present) tools |= XAAC_TOOL_TNS;
    if (ch.pulse.present) tools |= XAAC_TOOL_PULSE;
    if (ch.ics.window_sequence == XH_EIGHT_SHORT) tools |= XAAC_TOOL_SHORT;
    for (int g = 0; g < ch.ics.num_groups; g++)
      for (int sfb = 0; sfb < ch.ics.max_sfb; sfb++) {
        if (ch.cb[16 * g + sfb] >= XH_INTENSITY_HCB2) tools |= XAAC_TOOL_INTENSITY;
        if (ch.cb[16 * g + sfb] == XH_ESC_HCB) tools |= XAAC_TOOL_ESCAPE;
      }
  }
  return tools;
}

/* the frame at data[0 .. n) into p->el */
static int32_t parse_frame(xaac_parser *p, const uint8_t *data, size_t n, int32_t stage, size_t *consumed) {
  /* the frame's lines live in a buffer of the calling thread until the caller has copied them out (both entry points do,
     before they return): nothing of a stream outlives the frame there */
  static thread_local int32_t lines[2][XH_SPEC_WORDS];
  if (p->blocks_left > 0) {
    /* the next raw data block of the ADTS frame the call before started (data points behind what that call consumed): the
       reference reads a header only when its block count has run out (api.c:2914) */
    if (n < p->frame_left) return XAAC_PARSE_NEED_DATA;
    XhBits br(data, p->frame_left);
    p->el.ch[0].spec_mem = lines[0], p->el.ch[1].spec_mem = lines[1];
    const int32_t r = xh_parse_raw_data_block(&p->core, &br, &p->el, stage);
    if (r) {
      p->blocks_left = 0; /* (the rest of the frame goes with it: the caller looks for the next header) */
      if (consumed) *consumed = p->frame_left;
      return r;
    }
    size_t used = br.pos / 8 + (p->block_crc ? 2 : 0);
    p->blocks_left--;
    if (p->blocks_left == 0 || used > p->frame_left) used = p->frame_left; /* the last block ends the frame */
    p->frame_left -= used;
    if (consumed) *consumed = used;
    return XAAC_PARSE_OK;
  }
  xaac_adts_header h;
  const int32_t e = xaac_adts_parse_header(data, n, &h);
  if (e) return e;
  if (n < (size_t)h.frame_bytes) return XAAC_PARSE_NEED_DATA;
  if (h.frame_bytes < h.header_bytes) return XAAC_PARSE_ERR_UNSUPPORTED;
  if (consumed) *consumed = (size_t)h.frame_bytes;
  if (p->sr_index != h.sr_index) {
    const XhCoreState keep = p->core;
    if (xh_core_init(&p->core, h.sr_index)) return XAAC_PARSE_ERR_HEADER;
    if (p->sr_index >= 0) { /* a change of sampling rate in mid-stream: the noise generator runs on */
      p->core.pns_seed = keep.pns_seed;
      memcpy(p->core.pns_corr_seed, keep.pns_corr_seed, sizeof(keep.pns_corr_seed));
    }
    p->sr_index = h.sr_index;
    p->sampling_rate = h.sampling_rate;
  }
  XhBits br(data + h.header_bytes, (size_t)(h.frame_bytes - h.header_bytes));
  p->el.ch[0].spec_mem = lines[0], p->el.ch[1].spec_mem = lines[1];
  const int32_t r = xh_parse_raw_data_block(&p->core, &br, &p->el, stage);
  if (r || h.raw_blocks == 0) return r;
  /* number_of_raw_data_blocks_in_frame > 0 (headerdecode.c:353): this call delivers the first block and consumes the header
     and that block (+ its CRC in a protected frame); the calls that follow deliver the others, one each */
  size_t used = (size_t)h.header_bytes + br.pos / 8 + (h.protection_absent ? 0 : 2);
  if (used > (size_t)h.frame_bytes) used = (size_t)h.frame_bytes;
  p->blocks_left = h.raw_blocks;
  p->block_crc = h.protection_absent ? 0 : 1;
  p->frame_left = (size_t)h.frame_bytes - used;
  if (consumed) *consumed = used;
  return XAAC_PARSE_OK;
}

int32_t xaac_parse_adts_frame(xaac_parser *p, const uint8_t *data, size_t n, int32_t stage, xaac_core_frame *out,
                              size_t *consumed) {
  const int32_t r = parse_frame(p, data, n, stage, consumed);
  if (r) return r;
  const XhElement &el = p->el;
  out->n_ch = el.n_ch;
  out->element_id = el.id;
  out->common_window = el.common_window;
  out->sbr_ext_type = el.sbr_ext_type;
  out->sbr_bytes = el.sbr_bytes;
  memcpy(out->sbr, el.sbr, sizeof(out->sbr));
  out->tools = tools_of(el);
  for (int c = 0; c < el.n_ch; c++) {
    out->ics[c].window_sequence = (int16_t)el.ch[c].ics.window_sequence;
    out->ics[c].window_shape = (int16_t)el.ch[c].ics.window_shape;
    out->ics[c].max_sfb = (int16_t)el.ch[c].ics.max_sfb;
    out->ics[c].num_window_groups = (int16_t)el.ch[c].ics.num_groups;
    memcpy(out->spec[c], const_cast<XhElement &>(el).ch[c].spec(), sizeof(out->spec[c]));
  }
  return XAAC_PARSE_OK;
}

int32_t xaac_parse_sbr_side(xaac_parser *p, int32_t ps_enable, xaac_sbr_side *side) {
  if (!p || !side || p->sr_index < 0 || p->el.n_ch < 1) return XAAC_PARSE_ERR_SYNTAX;
  if (!p->sbr_ready) {
    xs_init(&p->sbr, p->sampling_rate, p->el.n_ch, ps_enable && p->el.n_ch == 1, p->esbr);
    p->sbr_ready = 1;
  }
  XsFrameResult r;
  if (xs_decode_frame(&p->sbr, p->el.sbr, p->el.sbr_bytes, p->el.sbr_ext_type, &side->header, side->frame, &side->ps_frame, &r))
    return XAAC_PARSE_ERR_SYNTAX;
  xs_frame_done(&p->sbr, &r); /* what the frame's ixheaacd_sbr_dec leaves for the next frame's delta decoding */
  side->apply = r.apply, side->reset = r.reset, side->reset_channels = r.reset_channels, side->upsampling = r.upsampling;
  side->stereo = r.stereo, side->ps = r.ps, side->ps_start = r.ps_start, side->frame_ok = r.frame_ok;
  return XAAC_PARSE_OK;
}

int32_t xaac_parser_set_esbr(xaac_parser *p, int32_t esbr) {
  if (!p || p->sbr_ready || (esbr != 0 && esbr != 1)) return XAAC_PARSE_ERR_SYNTAX;
  p->esbr = esbr;
  return XAAC_PARSE_OK;
}

int32_t xaac_parse_esbr_side(xaac_parser *p, int32_t channel, xaac_esbr_side *side) {
  if (!p || !side || !p->sbr_ready || !p->esbr || channel < 0 || channel > 1) return XAAC_PARSE_ERR_SYNTAX;
  xs_export_esbr_side(&p->sbr, channel, side);
  return XAAC_PARSE_OK;
}

int32_t xaac_parse_reset_pitch(xaac_parser *p, int32_t *pitch_in_bins) {
  if (!p || !pitch_in_bins || !p->sbr_ready) return XAAC_PARSE_ERR_SYNTAX;
  *pitch_in_bins = p->sbr.reset_pitch;
  return XAAC_PARSE_OK;
}

int32_t xaac_inverse_quant(int32_t magnitude, int32_t *out) {
  int err = 0;
  if (magnitude < 0 || !out) return XAAC_PARSE_ERR_SYNTAX;
  *out = xh_inverse_quant(magnitude, &err);
  return err ? XAAC_PARSE_ERR_ESCAPE : XAAC_PARSE_OK;
}

namespace {
/* the batch the team is working on (covered by the team's acquire() ... release()) */
struct BatchJob {
  xaac_parse_batch b;
  std::atomic<int> ok{0};
  std::atomic<bool> in_flight{false}; /* an xaac_parse_batch_start whose _wait has not come yet */
  std::atomic<uint64_t> owner{0};     /* ... and the thread that made it */
};
BatchJob g_job;

/* frame t (of the call's b->frames) of stream i; the arrays of step t lie one step's array behind those of step t - 1 */
int32_t parse_one(const xaac_parse_batch *b, int i, int t, std::atomic<int> *ok) {
  const int n_ch = b->n_ch;
  const size_t S = (size_t)b->n_streams * (size_t)t, SC = S * (size_t)n_ch; /* streams / channels in front of this step's rows */
  xaac_parser *p = b->parser[i];
  size_t used = 0;
  const uint64_t at = b->pos ? (b->pos[i] < b->bytes[i] ? b->pos[i] : b->bytes[i]) : 0;
  /* a step that fails behind a successful parse_frame (channel count, SBR side info) leaves the stream's read position where
     it was, so the multi-block bookkeeping parse_frame has already moved on has to go back with it: the next call would
     otherwise read the ADTS header it is pointed at as a raw data block */
  const int32_t keep_blocks = p->blocks_left, keep_crc = p->block_crc;
  const size_t keep_left = p->frame_left;
  int32_t r = parse_frame(p, b->data[i] + at, (size_t)(b->bytes[i] - at), b->stage, &used);
  const bool framed = r == 0;
  if (r == 0 && p->el.n_ch != n_ch) r = XAAC_PARSE_ERR_UNSUPPORTED;
  xaac_sbr_side *side = nullptr;
  if (r == 0 && b->with_sbr) {
    static thread_local xaac_sbr_side side_of_thread; /* copied into the batch's arrays below */
    side = &side_of_thread;
    r = xaac_parse_sbr_side(p, b->ps_enable, side);
  }
  b->status[S + i] = r;
  if (r) {
    if (framed) p->blocks_left = keep_blocks, p->block_crc = keep_crc, p->frame_left = keep_left;
    return r;
  }
  (*ok)++;
  b->consumed[i] += used;
  if (b->pos) b->pos[i] = at + used;
  if (b->tools) b->tools[S + i] = tools_of(p->el);
  for (int c = 0; c < n_ch; c++) {
    const size_t row = SC + (size_t)i * n_ch + c;
    memcpy(b->spec + row * 1024, p->el.ch[c].spec(), 1024 * sizeof(int32_t));
    b->ics[row * 2 + 0] = (uint8_t)p->el.ch[c].ics.window_sequence;
    b->ics[row * 2 + 1] = (uint8_t)p->el.ch[c].ics.window_shape;
    if (side) {
      b->header[row] = side->header;
      b->frame[row] = side->frame[c];
    }
  }
  if (b->lines) {
    int top = 0;
    for (int c = 0; c < n_ch; c++) {
      const int32_t *x = p->el.ch[c].spec();
      int blk = 1024 / 16;
      for (; blk > top; blk--) { /* the topmost block of 16 lines with a non-zero word */
        const int32_t *q = x + 16 * (blk - 1);
        int32_t any = 0;
        for (int k = 0; k < 16; k++) any |= q[k];
        if (any) break;
      }
      top = blk > top ? blk : top;
    }
    b->lines[S + i] = 16 * top;
  }
  if (side && b->reset_pitch && side->reset) b->reset_pitch[S + i] = p->sbr.reset_pitch;
  if (side && b->esbr_side && p->esbr)
    for (int c = 0; c < n_ch; c++) xs_export_esbr_side(&p->sbr, c, b->esbr_side + SC + (size_t)i * n_ch + c);
  if (side) {
    if (b->ps_frame) b->ps_frame[S + i] = side->ps_frame;
    int32_t *f = b->flags + (S + i) * 8;
    f[0] = side->apply, f[1] = side->reset, f[2] = side->reset_channels, f[3] = side->upsampling;
    f[4] = side->stereo, f[5] = side->ps, f[6] = side->ps_start, f[7] = side->frame_ok;
  }
  return 0;
}

/* one stream's frames of the call, one behind the other while its parser state and its bytes are in this core's caches */
void parse_item(const xaac_parse_batch *b, int i, std::atomic<int> *ok) {
  const int frames = b->frames > 1 ? b->frames : 1;
  b->consumed[i] = 0;
  for (int t = 0; t < frames; t++) {
    const int32_t r = parse_one(b, i, t, ok);
    if (r) { /* the stream stops here for this call: the later steps carry the same word */
      for (int u = t + 1; u < frames; u++) b->status[(size_t)b->n_streams * u + i] = r;
      return;
    }
  }
}

bool batch_ok(const xaac_parse_batch *b) {
  return b && b->n_streams >= 0 && (b->n_ch == 1 || b->n_ch == 2) && b->parser && b->data && b->bytes && b->spec && b->ics &&
         b->consumed && b->status && (!b->with_sbr || (b->header && b->frame && b->flags)) && b->frames >= 0 &&
         (b->frames <= 1 || b->pos); /* several frames per call: the positions are the library's */
}

int batch_threads(const xaac_parse_batch *b) {
  /* default: half the machine's hardware threads, at most 48 -- measured on a 2 x 64-core host (tools/bench_parser_scaling.py):
     3 - 4 x 10^6 HE-AACv2 frames/s at 32 .. 48 threads, less from 64 on (4096 streams x 20 KB of parser state are a
     latency-bound walk through memory that the second socket's threads only slow down; docs/NOTEBOOK.md 5l) */
  static const int usable = usable_cpus(); /* what the process may really use: a container lists 256 CPUs and grants 16 */
  int hw = (int)std::thread::hardware_concurrency();
  int threads = b->threads > 0 ? b->threads : (hw >= 4 ? (hw / 2 > 48 ? 48 : hw / 2) : (hw > 0 ? hw : 1));
  /* ... and at most twice what the process is granted: measured on a box that lists 256 CPUs and grants 16
     (tools/bench_parser_scaling.py, 4096 HE-AACv2 streams): 16 threads 1.9, 32 threads 2.9, 48 threads 1.6, 96 threads
     0.85 x 10^6 frames/s -- the workers wait on memory, so some oversubscription pays; spinning ones beyond it steal time */
  if (b->threads <= 0 && usable > 0 && threads > 2 * usable) threads = 2 * usable;
  if (threads > (b->n_streams + 3) / 4) threads = b->n_streams > 0 ? (b->n_streams + 3) / 4 : 1;
  return threads;
}

/* takes the team, hands it the batch (a copy of the descriptor: the caller's may go out of scope before the team is done) */
uint64_t this_thread() { return (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1u; }

void batch_launch(const xaac_parse_batch *b, bool caller_works) {
  team().acquire();
  /* _start: the owner is known from the moment this call has the team (the self-deadlock checks of _run / _start read it),
     but the batch only counts as in flight once the team has been handed it -- a _wait on another thread that found the flag
     set before the launch would join a team whose pending count still is the previous job's zero, return at once and release
     the team under the starter's feet.  Until the flag is set such a _wait gets ERR_SYNTAX ("nothing was started") */
  if (!caller_works) g_job.owner.store(this_thread(), std::memory_order_relaxed);
  g_job.b = *b;
  g_job.ok.store(0, std::memory_order_relaxed);
  team().launch(b->n_streams, batch_threads(b), [](int i) { parse_item(&g_job.b, i, &g_job.ok); }, caller_works);
  if (!caller_works) g_job.in_flight.store(true, std::memory_order_release);
}
}  // namespace

/* the caller's descriptor at the library's size: what the caller's struct does not hold reads as zero */
static bool batch_sized(const xaac_parse_batch *b, uint64_t struct_size, xaac_parse_batch *out) {
  if (!b || struct_size < offsetof(xaac_parse_batch, pos)) return false; /* (the original layout ends in front of pos) */
  memset(out, 0, sizeof(*out));
  memcpy(out, b, struct_size < sizeof(*out) ? (size_t)struct_size : sizeof(*out));
  return true;
}

int32_t xaac_parse_batch_run_sized(const xaac_parse_batch *b_in, uint64_t struct_size) {
  xaac_parse_batch full;
  if (!batch_sized(b_in, struct_size, &full)) return XAAC_PARSE_ERR_SYNTAX;
  const xaac_parse_batch *b = &full;
  if (!batch_ok(b)) return XAAC_PARSE_ERR_SYNTAX;
  /* between this thread's _start and its _wait the team belongs to that batch: waiting for it here would wait for the caller's
     own _wait (another thread's batch in flight is simply waited for, as two _run calls wait for each other) */
  if (g_job.in_flight.load(std::memory_order_acquire) && g_job.owner.load(std::memory_order_relaxed) == this_thread())
    return XAAC_PARSE_ERR_SYNTAX;
  batch_launch(b, true);
  team().join(true);
  const int ok = g_job.ok.load();
  team().release();
  return ok;
}

int32_t xaac_parse_batch_start_sized(const xaac_parse_batch *b_in, uint64_t struct_size) {
  xaac_parse_batch full;
  if (!batch_sized(b_in, struct_size, &full)) return XAAC_PARSE_ERR_SYNTAX;
  const xaac_parse_batch *b = &full;
  if (!batch_ok(b)) return XAAC_PARSE_ERR_SYNTAX;
  /* a second _start (or a _run) of the thread whose _start is unanswered is an error it gets back, not a wait for a release that
     only its own _wait would bring; behind another thread's batch the call waits like any other */
  if (g_job.in_flight.load(std::memory_order_acquire) && g_job.owner.load(std::memory_order_relaxed) == this_thread())
    return XAAC_PARSE_ERR_SYNTAX;
  batch_launch(b, false);
  return XAAC_PARSE_OK;
}

/* the symbols of the original layout (include/xaac_parse.h): the descriptor up to and including reset_pitch */
int32_t xaac_parse_batch_run(const xaac_parse_batch *b) { return xaac_parse_batch_run_sized(b, offsetof(xaac_parse_batch, pos)); }
int32_t xaac_parse_batch_start(const xaac_parse_batch *b) { return xaac_parse_batch_start_sized(b, offsetof(xaac_parse_batch, pos)); }

int32_t xaac_parse_batch_wait(double *busy_seconds) {
  if (!g_job.in_flight.exchange(false, std::memory_order_acq_rel)) return XAAC_PARSE_ERR_SYNTAX; /* nothing was started */
  team().join(false);
  const int ok = g_job.ok.load();
  if (busy_seconds) *busy_seconds = team().helpers_busy_seconds();
  team().release();
  return ok;
}

void xaac_sbr_state_init(xaac_sbr_state *s) {
  memset(s, 0, sizeof(*s));
  s->ov_lb_scale = s->hb_scale = s->ov_hb_scale = 31;
  s->st_syn_scale = -6;
  s->prev_end_position = 16;
  s->start_up = 1;
  s->tansient_env_prev = -1;
}

void xaac_ps_state_init(xaac_ps_state *s) {
  memset(s, 0, sizeof(*s));
  s->sample_ser[0] = 3, s->sample_ser[1] = 4, s->sample_ser[2] = 5; /* rev_link_delay_ser */
  memset(s->h11_h12_vec, 0xff, sizeof(s->h11_h12_vec));
  s->st_syn_scale_r = -6;
  s->ov_lb_scale_r = s->hb_scale_r = 31;
}

void xaac_esbr_state_init(xaac_esbr_state *s) {
  memset(s, 0, sizeof(*s));
  s->esbr_start_up = 1;
}

void xaac_esbr_ps_state_init(xaac_esbr_ps_state *s) {
  memset(s, 0, sizeof(*s));
  for (int b = 0; b < 20; b++) s->h_prev[0][b] = s->h_prev[1][b] = 1.0f;
}

void xaac_hbe_state_init(xaac_hbe_state *s) { memset(s, 0, sizeof(*s)); }

int32_t xaac_hbe_state_reinit(xaac_hbe_state *s, const xaac_sbr_header *h) {
  const int n_lo = h->num_sf_bands[0], n_hi = h->num_sf_bands[1];
  if (n_lo < 0 || n_lo > XAAC_SBR_MAX_FREQ_COEFFS / 2 || n_hi < 0 || n_hi > XAAC_SBR_MAX_FREQ_COEFFS) return -1;
  const int16_t *lo = h->freq_band_tbl_lo, *hi = h->freq_band_tbl_hi;
  if (lo[0] < 0 || lo[0] > 32) return -1;
  s->start_band = lo[0];
  s->end_band = lo[n_lo];
  s->synth_size = 4 * ((s->start_band + 4) / 8 + 1);
  s->k_start = xs_hbe_k_start(s->start_band);
  memset(s->synth_buf, 0, sizeof(s->synth_buf));
  memset(s->analy_buf, 0, sizeof(s->analy_buf));
  if (s->synth_size != 20) s->fft_ready = 1; /* the one bank size the reference sets no FFT pointers for keeps what a bank
                                                before it left (hbe_trans.c:164-169) */
  memset(s->x_over_qmf, 0, sizeof(s->x_over_qmf));
  int sfb = 0;
  for (int patch = 1; patch <= 4; patch++) { /* MAX_STRETCH */
    while (sfb <= n_lo && lo[sfb] <= patch * s->start_band) sfb++;
    if (sfb <= n_lo) {
      if (sfb > 0 && patch * s->start_band - lo[sfb - 1] <= 3) {
        s->x_over_qmf[patch - 1] = lo[sfb - 1];
      } else {
        int k = 0;
        while (k <= n_hi && hi[k] <= patch * s->start_band) k++;
        s->x_over_qmf[patch - 1] = k > 0 ? hi[k - 1] : 0;
      }
    } else {
      s->x_over_qmf[patch - 1] = s->end_band;
      s->max_stretch = patch < 4 ? patch : 4;
      break;
    }
  }
  return s->k_start < 0 ? -1 : 0;
}

namespace {
#include "tables_hbe_dft.inc"

/* ixheaacd_calc_anal_synth_window (hbe_dft_trans.c:140-270) for the transform sizes the reference has transforms for:
   sin(pi (j + 1/2) / fft_size) put together from the ROM's sines in float arithmetic, as the reference does */
bool xd_time_window(int fft_size, float *win) {
  const float *tab;
  int hop, stride, by2 = -1; /* by2: where the ROM holds sin / cos(pi / (2 N)) itself (the sizes that use every entry of their table) */
  switch (fft_size) {
    case 128: tab = xd_sine_pi_n_by_1024, hop = 8, stride = 512; break;
    case 256: tab = xd_sine_pi_n_by_1024, hop = 4, stride = 512; break;
    case 512: tab = xd_sine_pi_n_by_1024, hop = 2, stride = 512; break;
    case 192: tab = xd_sine_pi_n_by_768, hop = 4, stride = 384; break;
    case 384: tab = xd_sine_pi_n_by_768, hop = 2, stride = 384; break;
    case 448: tab = xd_sine_pi_n_by_896, hop = 2, stride = 448; break;
    case 768: tab = xd_sine_pi_n_by_768, hop = 1, stride = 384, by2 = 8; break;
    default: return false; /* (the other sizes of the reference's switch have no transform behind them: hbe_dft_trans.c:508-549) */
  }
  const float sin_pi_2_n = by2 >= 0 ? xd_sine_pi_by_2_n[by2] : tab[hop >> 1], cos_pi_2_n = by2 >= 0 ? xd_sine_pi_by_2_n[by2 + 1] : tab[stride + (hop >> 1)];
  int i = 0, j = 0;
  for (; j < (fft_size >> 1); i += hop, j++) {
    const float cos_val = tab[i + stride], sin_val = tab[i];
    win[j] = cos_val * sin_pi_2_n + sin_val * cos_pi_2_n;
  }
  for (; j < fft_size; j++, i += hop) {
    const float cos_val = tab[i - stride], sin_val = tab[i];
    win[j] = sin_val * cos_pi_2_n - cos_val * sin_pi_2_n;
  }
  return true;
}

/* ixheaacd_create_dft_hbe_window (:113-138): zero, a rising slope around x_over_bin1, one, a falling slope around x_over_bin2, zero --
   the first 772 entries of the reference's `size` (the transposer reads bins 0 .. fft_size / 2 <= 768); false where the
   rising slope would start in front of the array (the reference writes outside its own there) */
bool xd_fd_window(float *win, int x1, int x2, int ts, int size) {
  const float *slope = ts == 12 ? xd_window_ts_12 : xd_window_ts_18;
  if (x1 - ts / 2 < 0) return false;
  const auto put = [&](int n, float v) {
    if (n < 772) win[n] = v;
  };
  int n;
  for (n = 0; n < x1 - ts / 2; n++) put(n, 0);
  for (n = x1 - ts / 2; n <= x1 + ts / 2; n++) put(n, slope[n - (x1 - ts / 2)]);
  for (n = x1 + ts / 2 + 1; n < x2 - ts / 2; n++) put(n, 1.0f);
  for (n = x2 - ts / 2; n <= x2 + ts / 2; n++)
    if (n >= 0) put(n, 1.0f - slope[n - (x2 - ts / 2)]);
  for (n = x2 + ts / 2 + 1; n < size; n++)
    if (n >= 0) put(n, 0.0f);
  return true;
}
}  // namespace

int32_t xaac_hbe_dft_state_reinit(xaac_hbe_dft_state *s, xaac_hbe_dft_cfg *cfg, float *coef_re, float *coef_im, const xaac_sbr_header *h) {
  const int n_lo = h->num_sf_bands[0], n_hi = h->num_sf_bands[1];
  if (n_lo < 0 || n_lo > XAAC_SBR_MAX_FREQ_COEFFS / 2 || n_hi < 0 || n_hi > XAAC_SBR_MAX_FREQ_COEFFS) return -1;
  const int16_t *lo = h->freq_band_tbl_lo, *hi = h->freq_band_tbl_hi;
  if (lo[0] < 0 || lo[0] > 32) return -1;
  const int fft_size[2] = {1024, 1536}, trans_samp[2] = {12, 18};
  s->start_band = lo[0];
  s->end_band = lo[n_lo];
  s->synth_size = 4 * ((s->start_band + 4) / 8 + 1);
  s->k_start = xs_hbe_k_start(s->start_band);
  const int ana0 = (int)(s->synth_size / 32.0f * fft_size[0]);
  memset(cfg, 0, sizeof(*cfg));
  if (ana0 > XAAC_HBE_DFT_MAX_ANA || !xd_time_window(ana0, cfg->anal_window)) return -1;
  memset(s->synth_buf, 0, sizeof(s->synth_buf));
  const int temp_start = 2 * ((s->start_band - 1) / 2);
  const int top = (s->end_band + 1 < 64 ? s->end_band + 1 : 64);
  s->anal.analy_size = 4 * ((top - temp_start - 1) / 4 + 1);
  const int over = temp_start + s->anal.analy_size - 64;
  s->anal.a_start = temp_start - (over > 0 ? over : 0);
  const int L = s->anal.analy_size;
  const int syn0 = (int)(L / 64.0f * fft_size[0]);
  if (L < 4 || L > 64 || syn0 > XAAC_HBE_DFT_MAX_SYN || !xd_time_window(syn0, cfg->synth_window)) return -1;
  memset(s->x_over_qmf, 0, sizeof(s->x_over_qmf));
  int x_over_bin[4][2];
  memset(x_over_bin, 0, sizeof(x_over_bin));
  memset(coef_re, 0, sizeof(float) * 64 * 128);
  memset(coef_im, 0, sizeof(float) * 64 * 128);
  for (int k = 0; k < L; k++)
    for (int l = 0; l < 2 * L; l++) {
      const double a = 3.14159265358979323846 / (2 * L) * ((k + 0.5) * (2 * l - L / 64.0) - L / 64.0 * s->anal.a_start);
      coef_re[128 * k + l] = (float)cos(a);
      coef_im[128 * k + l] = (float)sin(a);
    }
  int sfb = 0;
  for (int patch = 1; patch <= 4; patch++) {
    while (sfb <= n_lo && lo[sfb] <= patch * s->start_band) sfb++;
    int band;
    if (sfb <= n_lo) {
      if (sfb > 0 && patch * s->start_band - lo[sfb - 1] <= 3) {
        band = lo[sfb - 1];
      } else {
        int k = 0;
        while (k <= n_hi && hi[k] <= patch * s->start_band) k++;
        band = k > 0 ? hi[k - 1] : 0;
      }
      s->x_over_qmf[patch - 1] = band;
      for (int o = 0; o < 2; o++) x_over_bin[patch - 1][o] = (int)(fft_size[o] * band / 128 + 0.5);
    } else {
      s->x_over_qmf[patch - 1] = s->end_band;
      for (int o = 0; o < 2; o++) x_over_bin[patch - 1][o] = (int)(fft_size[o] * s->end_band / 128 + 0.5);
      s->max_stretch = patch < 4 ? patch : 4;
      break;
    }
  }
  for (int patch = 0; patch < s->max_stretch - 1 && patch < 3; patch++)
    for (int o = 0; o < 2; o++)
      if (!xd_fd_window(cfg->fd_win[patch][o], x_over_bin[patch][o], x_over_bin[patch + 1][o], trans_samp[o], fft_size[o])) return -1;
  s->last_status = 0;
  return s->k_start < 0 ? -1 : 0;
}

int32_t xaac_hbe_state_reinit_tails(uint8_t *tails, const xaac_sbr_header *headers, int32_t n) {
  static thread_local xaac_hbe_state tmp; /* only its tail matters: the buffers the re-initialisation clears are the caller's */
  uint8_t *tail = reinterpret_cast<uint8_t *>(&tmp) + offsetof(xaac_hbe_state, synth_size);
  for (int32_t i = 0; i < n; i++) {
    memcpy(tail, tails + (size_t)i * XAAC_HBE_TAIL_BYTES, XAAC_HBE_TAIL_BYTES);
    if (xaac_hbe_state_reinit(&tmp, headers + i)) return i;
    memcpy(tails + (size_t)i * XAAC_HBE_TAIL_BYTES, tail, XAAC_HBE_TAIL_BYTES);
  }
  return -1;
}

void xaac_sbr_state_apply_side(xaac_sbr_state *s, const xaac_sbr_side *side, int32_t channel) {
  if (side->reset && channel < side->reset_channels) {
    s->ph_index = 0;
    s->filt_buf_noise_e = 0;
    s->start_up = 1;
    s->syn_lsb = s->codec_usb = side->header.sub_band_start;
    s->syn_usb = side->header.sub_band_end;
    memset(s->bw_array_prev, 0, sizeof(s->bw_array_prev));
  }
  if (side->upsampling) {
    s->syn_lsb = s->codec_usb = 32;
    s->syn_usb = 64;
  }
}

void xaac_ps_state_apply_side(xaac_ps_state *s, const xaac_sbr_side *side) {
  if (side->reset && side->reset_channels > 1) {
    s->syn_lsb_r = side->header.sub_band_start;
    s->syn_usb_r = side->header.sub_band_end;
  }
  if (side->upsampling) {
    s->syn_lsb_r = 32;
    s->syn_usb_r = 64;
  }
}

}  // extern "C"
