/*
 * tests/fuzz/tsan_team.cpp -- TEST INFRASTRUCTURE: xaac_parse_batch_run's worker team under ThreadSanitizer.  The batch call
 * is run many times back to back with the thread count alternating between calls (1, 7, 2, 8, 3, ...) and stream counts that
 * change the clamp `threads <= (n_streams + 3) / 4`, from two caller threads at once: a worker that is idle in one call
 * must never read the next call's job fields under the old generation, parse a stream twice, or touch the caller's
 * (stack-allocated) job after the call returned.  Every third call goes through xaac_parse_batch_start / _wait instead (the
 * descriptor freed in between, the wait on another thread every other time).  Results are compared with a single-threaded
 * pass of the same frames.
 *   tsan_team <stream.aac> <calls>
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/xaac_parse.h"

struct Streams {
  int n = 0;
  std::vector<xaac_parser *> parser;
  std::vector<const uint8_t *> data;
  std::vector<uint64_t> bytes, consumed;
  std::vector<int32_t> spec, status, flags;
  std::vector<uint8_t> ics;
  std::vector<xaac_sbr_header> header;
  std::vector<xaac_sbr_frame> frame;
  std::vector<size_t> pos;
};

static uint32_t crc_of(const std::vector<int32_t> &v) {
  uint32_t h = 2166136261u;
  for (int32_t x : v) h = (h ^ (uint32_t)x) * 16777619u;
  return h;
}

static int run_caller(const std::vector<uint8_t> &file, int n_streams, int calls, int first_threads, std::vector<uint32_t> *sums) {
  Streams s;
  s.n = n_streams;
  s.parser.resize(n_streams), s.data.resize(n_streams), s.bytes.resize(n_streams), s.consumed.resize(n_streams);
  s.spec.resize((size_t)n_streams * 1024), s.status.resize(n_streams), s.flags.resize((size_t)n_streams * 8);
  s.ics.resize((size_t)n_streams * 2), s.header.resize(n_streams), s.frame.resize(n_streams), s.pos.assign(n_streams, 0);
  for (int i = 0; i < n_streams; i++)
    if (xaac_parser_create(&s.parser[i])) return 2;
  for (int c = 0; c < calls; c++) {
    /* streams drop out of the call like -ilist streams that ended: the active count and with it the clamp changes */
    const int live = (c % 5 == 4) ? (n_streams > 6 ? 5 : n_streams) : n_streams;
    for (int i = 0; i < live; i++) s.data[i] = file.data() + s.pos[i], s.bytes[i] = file.size() - s.pos[i];
    xaac_parse_batch b;
    memset(&b, 0, sizeof(b));
    b.n_streams = live, b.n_ch = 1, b.with_sbr = 1, b.ps_enable = 1, b.stage = 1;
    b.threads = first_threads > 0 ? 1 + (c * 3 + first_threads) % 8 : 1;
    b.parser = s.parser.data(), b.data = s.data.data(), b.bytes = s.bytes.data(), b.spec = s.spec.data(), b.ics = s.ics.data();
    b.header = s.header.data(), b.frame = s.frame.data(), b.flags = s.flags.data(), b.consumed = s.consumed.data();
    b.status = s.status.data();
    int ok;
    if (first_threads > 0 && c % 3 == 1) { /* the two-halves form: the descriptor dies between the halves, the waiter may be another thread */
      xaac_parse_batch *tmp = new xaac_parse_batch(b);
      if (xaac_parse_batch_start(tmp)) return 1;
      memset(tmp, 0xee, sizeof(*tmp));
      delete tmp;
      double busy = -1.0;
      if (c % 2) {
        std::thread waiter([&] { ok = xaac_parse_batch_wait(&busy); });
        waiter.join();
      } else {
        ok = xaac_parse_batch_wait(&busy);
      }
      if (busy < 0.0) return 1;
    } else {
      ok = xaac_parse_batch_run(&b);
    }
    if (ok != live) {
      fprintf(stderr, "call %d: %d of %d streams parsed\n", c, ok, live);
      return 1;
    }
    for (int i = 0; i < live; i++) s.pos[i] += (size_t)s.consumed[i];
    sums->push_back(crc_of(s.spec) ^ (uint32_t)live);
  }
  for (int i = 0; i < n_streams; i++) xaac_parser_destroy(s.parser[i]);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> file(1 << 20);
  file.resize(fread(file.data(), 1, file.size(), f));
  fclose(f);
  const int calls = atoi(argv[2]);
  std::vector<uint32_t> want, got_a, got_b;
  if (run_caller(file, 29, calls, 0, &want)) return 1; /* one thread: the reference result */
  int ra = 0, rb = 0;
  std::thread ta([&] { ra = run_caller(file, 29, calls, 1, &got_a); });
  std::thread tb([&] { rb = run_caller(file, 29, calls, 4, &got_b); });
  ta.join(), tb.join();
  if (ra || rb) return 1;
  if (got_a != want || got_b != want) {
    fprintf(stderr, "threaded results differ from the single-threaded pass\n");
    return 1;
  }
  printf("%d calls x 2 callers with alternating thread counts: equal to the single-threaded pass\n", calls);
  return 0;
}
