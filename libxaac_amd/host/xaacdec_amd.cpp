/*
 * xaacdec_amd.cpp -- the native command line decoder of this repo: ADTS AAC-LC / HE-AAC / HE-AACv2 in, 16-bit WAV out, as
 * `xaacdec` (test/decoder/ixheaacd_main.c over decoder/ixheaacd_api.c:2624-3788) decodes it -- with the reference's default
 * -esbr:1 (SBR streams through the float eSBR tools, "Path A") or with -esbr:0 (the fixed-point SBR tools) -- with the repo's own
 * host front end (include/xaac_parse.h, CPU threads) in front of the GPU entry points of include/xaac_amd.h and every
 * stream's state resident in device memory.  No reference code, no Python, no torch: HIP runtime + the two libraries.
 *
 *   xaacdec_amd -ifile:<in.aac> -ofile:<out.wav> [-esbr:<0|1>] [-copies:<N>] [-verify] [-threads:<T>] [-quiet]
 *   xaacdec_amd -ilist:<file with one input path per line> -odir:<directory> [-esbr:<0|1>] [-threads:<T>] [-quiet]
 *   ... [-gpus:<G>] [-device:<k>] [-plan]
 *
 * -gpus:G shards the batch's streams over G devices of this node (k, k + 1, ... from -device:k, default 0): contiguous ranges
 * whose sizes differ by at most one -- libxaac_amd/dist.py: shard_range, the split bench.py --gpus N makes over ranks -- one host
 * thread, HIP context, stream and set of resident states per device, nothing shared between the shards but the read-only
 * inputs (streams are independent: no data-path exchange; the PCM of every shard comes down to its own host thread).
 * -plan prints the split and exits before anything touches a device (tests/test_cli_plan_cpu.py); -wrap_devices lets the
 * shards wrap around the devices the node has (two shards on one device: the -gpus host path on a one-GPU box, tests/test_cli_gpu.py).
 *
 * -copies:N decodes N instances of the stream in one lock-step batch (the first one's PCM is written; with -verify all N are
 * compared with it word for word) and prints the end-to-end rate: the shape a serving host has, with one input here for brevity.
 * -ilist decodes different streams of one kind (sampling rate, channels, SBR / PS or not) in one batch, each to <odir>/<name>.wav;
 * a stream that ends drops out of the steps, the others go on.
 * libxaac_amd/decoder.py is the same loop in Python (used by the tests for its ease of inspection).
 */
#include <hip/hip_runtime_api.h>
#include <ctype.h>
#include <sched.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/xaac_amd.h"
#include "../../include/xaac_esbr.h"
#include "../../include/xaac_parse.h"

namespace {

[[noreturn]] void die(const char *what, long code = 0) {
  fprintf(stderr, "xaacdec_amd: %s failed (%ld)\n", what, code);
  exit(2);
}
#define HIP(x)                         \
  do {                                 \
    hipError_t e_ = (x);               \
    if (e_ != hipSuccess) die(#x, e_); \
  } while (0)
#define XA(x)                 \
  do {                        \
    int32_t e_ = (x);         \
    if (e_ != 0) die(#x, e_); \
  } while (0)

template <class T>
T *dev(size_t n) {
  void *p = nullptr;
  HIP(hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : 16));
  HIP(hipMemset(p, 0, n * sizeof(T) ? n * sizeof(T) : 16));
  return static_cast<T *>(p);
}
/* CPUs of the NUMA node the GPU hangs off (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node -> the node's
   cpulist); an empty set where that cannot be read.  Pinned staging memory is allocated and first touched from there:
   with the staging on the other socket the bus carries one direction at full rate but both at once -- spectra going up
   beside the PCM of the step before coming down -- at 38 GiB/s in total instead of 63 (a two-socket MI355X host). */
cpu_set_t gpu_node_cpus(int device, bool *known) {
  cpu_set_t set;
  CPU_ZERO(&set);
  *known = false;
  char id[64] = {0}, path[160];
  if (hipDeviceGetPCIBusId(id, (int)sizeof(id), device) != hipSuccess) return set;
  for (char *c = id; *c; c++) *c = (char)tolower(*c);
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
  int node = -1;
  if (FILE *f = fopen(path, "r")) {
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return set;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return set;
  int a, b;
  for (;;) {
    if (fscanf(f, "%d", &a) != 1) break;
    b = a;
    int ch = fgetc(f);
    if (ch == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      ch = fgetc(f);
    }
    for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &set), *known = true;
    if (ch != ',') break;
  }
  fclose(f);
  return set;
}
/* the GPU's NUMA node is looked up once per device (the shards of a -gpus run sit on different nodes of a two-socket host) */
cpu_set_t near_cpus(int device, bool *known) { /* (by value: the table may grow under another shard's thread) */
  static std::mutex mu;
  static std::vector<std::pair<int, std::pair<bool, cpu_set_t>>> seen;
  std::lock_guard<std::mutex> lk(mu);
  for (auto &e : seen)
    if (e.first == device) return *known = e.second.first, e.second.second;
  bool k = false;
  const cpu_set_t set = gpu_node_cpus(device, &k);
  seen.push_back({device, {k, set}});
  *known = k;
  return seen.back().second.second;
}
template <class T>
T *pinned(size_t n, int device) {
  bool known = false;
  const cpu_set_t near = near_cpus(device, &known);
  cpu_set_t before, both;
  /* first touch on the GPU's NUMA node: only CPUs this thread may run on anyway (a cpuset that does not meet the node leaves the
     thread where it is) */
  bool moved = false;
  if (known && sched_getaffinity(0, sizeof(before), &before) == 0) {
    CPU_AND(&both, &before, &near);
    moved = CPU_COUNT(&both) > 0 && sched_setaffinity(0, sizeof(both), &both) == 0;
  }
  void *p = nullptr;
  HIP(hipHostMalloc(&p, n * sizeof(T) ? n * sizeof(T) : 16, hipHostMallocDefault));
  memset(p, 0, n * sizeof(T) ? n * sizeof(T) : 16);
  if (moved) sched_setaffinity(0, sizeof(before), &before);
  return static_cast<T *>(p);
}

struct Staging { /* what one step's parse leaves for the GPU */
  int32_t *spec;
  uint8_t *ics;
  xaac_sbr_header *header;
  xaac_sbr_frame *frame;
  xaac_ps_frame *ps;
  xaac_esbr_side *eside;
  std::vector<int32_t> flags, status, reset_pitch;
  int delivered;
  int lines; /* leading spectral lines that may be non-zero in a delivered row (xaac_parse_batch::lines, rounded up to 64) */
};

/* what one parser call fills: kFramesPerParse steps, their pinned arrays one behind the other (xaac_parse_batch::frames) */
constexpr int kFramesPerParse = 4;
struct StagingGroup {
  std::vector<int32_t> flags, status, reset_pitch, lines;
  std::vector<uint64_t> consumed;
};

void write_wav(const std::string &path, const std::vector<int16_t> &pcm, int channels, int rate) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) die("fopen(output)");
  const uint32_t data = (uint32_t)(pcm.size() * 2), riff = 36 + data, fmt = 16, byte_rate = (uint32_t)(rate * channels * 2);
  const uint16_t pcm_tag = 1, ch = (uint16_t)channels, align = (uint16_t)(channels * 2), bits = 16;
  const uint32_t sr = (uint32_t)rate;
  fwrite("RIFF", 1, 4, f), fwrite(&riff, 4, 1, f), fwrite("WAVEfmt ", 1, 8, f), fwrite(&fmt, 4, 1, f);
  fwrite(&pcm_tag, 2, 1, f), fwrite(&ch, 2, 1, f), fwrite(&sr, 4, 1, f), fwrite(&byte_rate, 4, 1, f);
  fwrite(&align, 2, 1, f), fwrite(&bits, 2, 1, f), fwrite("data", 1, 4, f), fwrite(&data, 4, 1, f);
  fwrite(pcm.data(), 2, pcm.size(), f);
  fclose(f);
}

/* what the command line fixes for every shard */
struct Job {
  std::vector<std::vector<uint8_t>> datas; /* -ilist: one per stream; else the one input */
  xaac_adts_header hdr;
  int n_ch, sbr, esbr, out_ch, rate, out_rate, per;
  int hq; /* -esbr_hq:1: the DFT harmonic transposer in the QMF one's place (Path A only) */
  int threads, verify, profile;
  bool list_mode;
};
/* one device's share of the batch: streams [lo, lo + n) on HIP device `device`, decoded by one host thread */
struct Shard {
  int device, lo, n;
  std::vector<std::vector<int16_t>> pcms; /* every stream's output (-ilist), or the shard's first stream's */
  long frames = 0, mismatched = 0, first_frames = 0;
  double parse_s = 0, wall = 0, steady = 0, phase_s[4] = {0, 0, 0, 0};
};

/* libxaac_amd/dist.py: shard_range -- contiguous [lo, hi) of n items owned by shard r of g; sizes differ by at most one */
void shard_range(int n, int r, int g, int *lo, int *hi) {
  const int base = n / g, rem = n % g;
  *lo = r * base + (r < rem ? r : rem);
  *hi = *lo + base + (r < rem ? 1 : 0);
}

void decode_shard(const Job &J, Shard &S) {
  const int device = S.device, N = S.n, n_ch = J.n_ch, sbr = J.sbr, esbr = J.esbr, out_ch = J.out_ch, rate = J.rate, per = J.per;
  const int threads = J.threads, verify = J.verify, profile = J.profile, hq = J.esbr && J.hq;
  const bool list_mode = J.list_mode;
  const int NC = N * n_ch, NCD = NC;
  S.first_frames = N;
  xaac_ctx *ctx = nullptr;
  hipStream_t stream;
  HIP(hipSetDevice(device));
  HIP(hipStreamCreate(&stream));
  XA(xaac_create(&ctx, device, stream));
  XA(xaac_warm_up(ctx)); /* the kernels' code objects are on the device before the first batch (and the run's clock) */
  std::vector<xaac_parser *> parser((size_t)N);
  for (auto &p : parser) {
    XA(xaac_parser_create(&p));
    if (esbr) XA(xaac_parser_set_esbr(p, 1));
  }
  std::vector<const uint8_t *> ptr((size_t)N);
  std::vector<uint64_t> left((size_t)N), pos((size_t)N, 0);
  std::vector<char> broken((size_t)N, 0); /* -ilist: a stream whose frame did not parse is treated as over from there on */

  /* device-resident state and per-step device buffers */
  int32_t *d_overlap = dev<int32_t>((size_t)NCD * 512), *d_spec = dev<int32_t>((size_t)NCD * 1024);
  xaac_ovl_state *d_ovl = dev<xaac_ovl_state>((size_t)NCD);
  xaac_ics_info *d_ics = dev<xaac_ics_info>((size_t)NCD);
  /* PCM and status of a step in two sets: the copy down of step k (a second stream) runs beside the copies up and the kernels
     of step k + 1 */
  int16_t *d_pcm2[2], *h_pcm2[2], *d_mono2[2] = {nullptr, nullptr};
  int32_t *d_status2[2], *h_status2[2];
  hipStream_t down;
  hipEvent_t ev_kernels[2], ev_down[2];
  HIP(hipStreamCreate(&down));
  for (int k = 0; k < 2; k++) {
    d_pcm2[k] = dev<int16_t>((size_t)N * per * out_ch), h_pcm2[k] = pinned<int16_t>((size_t)N * per * 2, device);
    d_status2[k] = dev<int32_t>((size_t)NC), h_status2[k] = pinned<int32_t>((size_t)NC, device);
    HIP(hipEventCreateWithFlags(&ev_kernels[k], hipEventDisableTiming));
    HIP(hipEventCreateWithFlags(&ev_down[k], hipEventDisableTiming));
  }
  /* AAC-LC */
  int32_t *d_out32 = nullptr;
  int8_t *d_qadj = nullptr;
  xaac_limiter_state *d_lim = nullptr;
  int delay = 0;
  /* SBR */
  int16_t *d_core = nullptr;
  xaac_sbr_header *d_header = nullptr;
  xaac_sbr_frame *d_frame = nullptr;
  xaac_sbr_state *d_state = nullptr;
  xaac_ps_frame *d_psf = nullptr;
  xaac_ps_state *d_ps_state = nullptr;
  int32_t *d_idx = nullptr;
  int32_t *d_flags = nullptr, *h_flags = nullptr; /* the parser's flag rows as xaac_sbr_state_apply_side_batch takes them */
  /* Path A (-esbr:1) */
  xaac_esbr_side *d_eside = nullptr;
  xaac_esbr_state *d_estate = nullptr;
  xaac_hbe_state *d_hbe = nullptr, *d_hbe_tmp = nullptr; /* d_hbe_tmp: the resetting channels of a step gathered (partial resets) */
  xaac_esbr_ps_state *d_eps = nullptr;
  float *d_fcore = nullptr, *d_out_l = nullptr, *d_out_r = nullptr, *d_q = nullptr, *d_pv = nullptr;
  std::vector<uint8_t> hbe_tail; /* every channel's transposer integers, kept between resets: some survive one (max_stretch, fft_ready) */
  const auto hbe_hint = [&]() { /* the largest bank of the batch, as the ABI's LDS hint takes it: 8, or 0 = any */
    int32_t smax = 0, v;
    constexpr size_t kTail = sizeof(xaac_hbe_state) - offsetof(xaac_hbe_state, synth_size);
    for (size_t i = 0; i * kTail < hbe_tail.size(); i++) memcpy(&v, &hbe_tail[i * kTail], 4), smax = v > smax ? v : smax;
    return smax <= 8 ? 8 : 0;
  };
  /* -esbr_hq:1: every channel's DFT transposer, the configurations their headers' band tables gave (shared by channels with the
     same tables), and what the host keeps between resets: max_stretch (the re-initialisation leaves it alone when four patches
     fit), the last processed frame's over_sampling_flag (the reset-time runs use the transposer's: sbr_dec.c:884 sets it) */
  xaac_hbe_dft_state *d_dft = nullptr, *d_dft_tmp = nullptr;
  xaac_hbe_dft_cfg *d_dcfg = nullptr;
  float *d_dcoef = nullptr;       /* [2][NC][64][128]: real matrices of the configurations, then the imaginary ones */
  int32_t *d_dslot = nullptr, *d_dslot_tmp = nullptr, *d_dovs = nullptr; /* [NC] configuration of a channel; of the gathered channels; their flags */
  std::vector<int32_t> dft_ms, dft_ovs, dft_slot;
  std::map<std::string, int> dft_cfgs;
  float *d_older = nullptr; /* [2][NC][24][64]: rows 8..31 of the QMF history as the frame before found them (see the reset) */
  void *d_ws = nullptr;
  uint64_t ws_bytes = 0;
  if (!sbr) {
    d_out32 = dev<int32_t>((size_t)N * 1024 * n_ch);
    d_qadj = dev<int8_t>((size_t)N * n_ch);
    d_lim = dev<xaac_limiter_state>((size_t)N);
    xaac_limiter_state l0;
    delay = xaac_peak_limiter_init(&l0, (uint32_t)n_ch, (uint32_t)rate);
    if (delay < 0) die("xaac_peak_limiter_init", delay);
    for (int i = 0; i < N; i++) HIP(hipMemcpy(d_lim + i, &l0, sizeof(l0), hipMemcpyHostToDevice));
    ws_bytes = xaac_peak_limiter_workspace_bytes(N);
  } else if (esbr) {
    d_core = dev<int16_t>((size_t)NC * 1024);
    d_header = dev<xaac_sbr_header>((size_t)NC);
    d_frame = dev<xaac_sbr_frame>((size_t)NC);
    d_eside = dev<xaac_esbr_side>((size_t)NC);
    d_estate = dev<xaac_esbr_state>((size_t)NC);
    if (hq) {
      d_dft = dev<xaac_hbe_dft_state>((size_t)NC); /* all zero for a new stream: refused (last_status -1) until a header sets it up */
      d_dft_tmp = dev<xaac_hbe_dft_state>((size_t)NC);
      d_dcfg = dev<xaac_hbe_dft_cfg>((size_t)NC);
      d_dcoef = dev<float>((size_t)2 * NC * 64 * 128);
      d_dslot = dev<int32_t>((size_t)NC), d_dslot_tmp = dev<int32_t>((size_t)NC), d_dovs = dev<int32_t>((size_t)NC);
      dft_ms.assign((size_t)NC, 0), dft_ovs.assign((size_t)NC, 0), dft_slot.assign((size_t)NC, 0);
    } else {
      d_hbe = dev<xaac_hbe_state>((size_t)NC); /* all zero for a new stream */
    }
    d_fcore = dev<float>((size_t)NC * 1024);
    d_out_l = dev<float>((size_t)NC * 2048);
    d_older = dev<float>((size_t)2 * NC * 24 * 64);
    static thread_local xaac_esbr_state e0; /* (thread_local: one decode_shard per device thread) */
    xaac_esbr_state_init(&e0);
    for (int i = 0; i < NC; i++) HIP(hipMemcpy(d_estate + i, &e0, sizeof(e0), hipMemcpyHostToDevice));
    if (n_ch == 1) {
      d_psf = dev<xaac_ps_frame>((size_t)N);
      d_eps = dev<xaac_esbr_ps_state>((size_t)N);
      d_out_r = dev<float>((size_t)N * 2048);
      static thread_local xaac_esbr_ps_state p0;
      xaac_esbr_ps_state_init(&p0);
      for (int i = 0; i < N; i++) HIP(hipMemcpy(d_eps + i, &p0, sizeof(p0), hipMemcpyHostToDevice));
    }
    ws_bytes = xaac_esbr_workspace_bytes(NC);
  } else {
    d_core = dev<int16_t>((size_t)NC * 1024);
    d_header = dev<xaac_sbr_header>((size_t)NC);
    d_frame = dev<xaac_sbr_frame>((size_t)NC);
    d_state = dev<xaac_sbr_state>((size_t)NC);
    xaac_sbr_state s0;
    xaac_sbr_state_init(&s0);
    {
      std::vector<xaac_sbr_state> all((size_t)NC, s0);
      HIP(hipMemcpy(d_state, all.data(), all.size() * sizeof(s0), hipMemcpyHostToDevice));
    }
    d_flags = dev<int32_t>((size_t)N * 8), h_flags = pinned<int32_t>((size_t)N * 8, device);
    if (n_ch == 1) {
      d_psf = dev<xaac_ps_frame>((size_t)N);
      d_ps_state = dev<xaac_ps_state>((size_t)N);
      d_mono2[0] = dev<int16_t>((size_t)N * 2048), d_mono2[1] = dev<int16_t>((size_t)N * 2048);
      d_idx = dev<int32_t>((size_t)N);
      xaac_ps_state p0;
      xaac_ps_state_init(&p0);
      {
        std::vector<xaac_ps_state> all((size_t)N, p0);
        HIP(hipMemcpy(d_ps_state, all.data(), all.size() * sizeof(p0), hipMemcpyHostToDevice));
      }
      ws_bytes = xaac_sbr_hq_workspace_bytes(N, 1);
    } else {
      ws_bytes = xaac_sbr_lp_workspace_bytes(NC);
    }
  }
  HIP(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 16));

  /* three groups of kFramesPerParse steps: the parse of group g + 1 | the copies up and kernels of group g's steps | the copy
     down of the step before.  A stream's parser state and bytes are fetched once per call for kFramesPerParse frames (on the
     2 x 64-core box 32 threads parse 3.9-4.6 x 10^6 frames/s one frame per call, 4.9-5.6 x 10^6 with 2..8). */
  constexpr int T = kFramesPerParse;
  Staging st[3 * T];
  StagingGroup grp[3];
  for (int g = 0; g < 3; g++) {
    int32_t *spec = pinned<int32_t>((size_t)T * NC * 1024, device);
    uint8_t *ics = pinned<uint8_t>((size_t)T * NC * 2, device);
    xaac_sbr_header *header = sbr ? pinned<xaac_sbr_header>((size_t)T * NC, device) : nullptr;
    xaac_sbr_frame *frame = sbr ? pinned<xaac_sbr_frame>((size_t)T * NC, device) : nullptr;
    xaac_ps_frame *psf = (sbr && n_ch == 1) ? pinned<xaac_ps_frame>((size_t)T * N, device) : nullptr;
    xaac_esbr_side *eside = esbr ? pinned<xaac_esbr_side>((size_t)T * NC, device) : nullptr;
    grp[g].flags.assign((size_t)T * N * 8, 0), grp[g].status.assign((size_t)T * N, 0), grp[g].reset_pitch.assign((size_t)T * N, 0);
    grp[g].lines.assign((size_t)T * N, 0), grp[g].consumed.assign((size_t)N, 0);
    for (int t = 0; t < T; t++) {
      Staging &s = st[g * T + t];
      s.spec = spec + (size_t)t * NC * 1024, s.ics = ics + (size_t)t * NC * 2;
      s.header = header ? header + (size_t)t * NC : nullptr, s.frame = frame ? frame + (size_t)t * NC : nullptr;
      s.ps = psf ? psf + (size_t)t * N : nullptr, s.eside = eside ? eside + (size_t)t * NC : nullptr;
      s.flags.assign((size_t)N * 8, 0), s.status.assign((size_t)N, 0), s.reset_pitch.assign((size_t)N, 0);
      s.delivered = 0, s.lines = 1024;
    }
  }
  for (int i = 0; i < N; i++) {
    const std::vector<uint8_t> &d = J.datas[list_mode ? (size_t)(S.lo + i) : 0];
    ptr[(size_t)i] = d.data(), left[(size_t)i] = d.size(); /* the whole streams: the library keeps the read positions (pos) */
  }
  double &parse_s = S.parse_s;
  auto parse = [&](int g) { /* the next kFramesPerParse frames of every stream into one group of staging sets */
    const auto t0 = std::chrono::steady_clock::now();
    StagingGroup &G = grp[g];
    for (int i = 0; i < N; i++)
      if (broken[(size_t)i]) left[(size_t)i] = 0;
    xaac_parse_batch b;
    memset(&b, 0, sizeof(b));
    b.n_streams = N, b.n_ch = n_ch, b.with_sbr = sbr, b.ps_enable = 1, b.stage = 2, b.threads = threads;
    b.parser = parser.data(), b.data = ptr.data(), b.bytes = left.data(), b.pos = pos.data(), b.frames = T;
    Staging &s0 = st[g * T];
    b.spec = s0.spec, b.ics = s0.ics, b.header = s0.header, b.frame = s0.frame, b.ps_frame = s0.ps;
    b.flags = G.flags.data(), b.consumed = G.consumed.data(), b.status = G.status.data(), b.esbr_side = s0.eside;
    b.reset_pitch = G.reset_pitch.data(), b.lines = G.lines.data();
    const int ok = xaac_parse_batch_run(&b);
    if (ok < 0) die("xaac_parse_batch_run", ok);
    for (int t = 0; t < T; t++) {
      Staging &s = st[g * T + t];
      int delivered = 0, lines = 0;
      for (int i = 0; i < N; i++) {
        int32_t r = G.status[(size_t)t * N + i];
        if (r < 0) {
          /* one file of a list with trailing bytes or damage must not take the other streams' output along: that stream ends
             here (what it delivered so far is written), the batch goes on */
          if (!list_mode) die("a frame does not parse", r);
          if (!broken[(size_t)i])
            fprintf(stderr, "xaacdec_amd: stream %d: a frame does not parse (%d) at byte %llu: the stream ends here\n", S.lo + i, r,
                    (unsigned long long)pos[(size_t)i]);
          broken[(size_t)i] = 1;
          r = XAAC_PARSE_NEED_DATA;
        }
        s.status[(size_t)i] = r;
        s.reset_pitch[(size_t)i] = G.reset_pitch[(size_t)t * N + i];
        for (int k = 0; k < 8; k++) s.flags[(size_t)i * 8 + k] = G.flags[((size_t)t * N + i) * 8 + k];
        if (r == 0) {
          delivered++;
          lines = G.lines[(size_t)t * N + i] > lines ? G.lines[(size_t)t * N + i] : lines;
        }
      }
      s.delivered = delivered;
      s.lines = (lines + 63) & ~63;
    }
    parse_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };

  double (&phase_s)[4] = S.phase_s; /* -profile: copies up, kernels, copies down, host work on the PCM */
  auto t_phase = std::chrono::steady_clock::now();
  auto lap = [&](int k) {
    if (!profile) return;
    HIP(hipStreamSynchronize(stream));
    const auto now = std::chrono::steady_clock::now();
    phase_s[k] += std::chrono::duration<double>(now - t_phase).count();
    t_phase = now;
  };
  std::vector<std::vector<int16_t>> &pcms = S.pcms;
  pcms.assign((size_t)(list_mode ? N : 1), std::vector<int16_t>()); /* every stream's output (-ilist), or stream 0's */
  std::vector<char> ended((size_t)N, 0);
  std::vector<xaac_limiter_state> lim_at_end; /* -ilist, AAC-LC: the limiter state a stream leaves behind its last frame */
  if (list_mode && !sbr) lim_at_end.resize((size_t)N);
  long &frames = S.frames, &mismatched = S.mismatched;
  int lines_held = 0; /* leading spectral lines that may be non-zero in d_spec */
  bool first = true;
  const auto t_all = std::chrono::steady_clock::now();
  auto t_first = t_all;
  /* one helper thread for the whole run: it parses the next staging set when told to, the main thread waits for `done` */
  std::mutex mu;
  std::condition_variable cv;
  int job = 0, done = -1;
  bool quit = false;
  std::thread worker([&] {
    for (int expect = 0;; expect++) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return quit || job > expect; });
      if (quit) return;
      lk.unlock();
      parse(expect % 3);
      lk.lock();
      done = expect;
      cv.notify_all();
    }
  });
  auto start_parse = [&] {
    std::lock_guard<std::mutex> lk(mu);
    job++;
    cv.notify_all();
  };
  auto wait_parse = [&](int step) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done >= step; });
  };
  /* what a step leaves for the host once its copy down has arrived */
  std::vector<uint8_t> refused((size_t)N, 0); /* -ilist: streams a kernel refused a frame of (their output has ended) */
  struct Pending {
    bool valid, mono_twice, first;
    int slot, which;
    bool some_mono; /* a PS batch with streams that have no PS in this frame: their left samples also go to the right */
  } pending = {false, false, false, 0, 0, false};
  auto consume = [&]() {
    if (!pending.valid) return;
    HIP(hipEventSynchronize(ev_down[pending.slot]));
    int16_t *h_pcm = h_pcm2[pending.slot];
    const int32_t *h_status = h_status2[pending.slot];
    const std::vector<int32_t> &alive = st[pending.which].status; /* 0: the stream delivered a frame in that step */
    if (sbr) { /* (rows of streams that are over re-run their last staging rows: what the kernels say about those is not looked at) */
      const int rows = (pending.mono_twice || (n_ch == 1 && !esbr)) ? N : NC;
      for (int i = 0; i < rows; i++) {
        const size_t si = (size_t)(rows == N ? i : i / n_ch);
        if (h_status[i] < 0 && alive[si] == 0 && !refused[si]) {
          /* side info the kernels do not take (the boundary's own checks: a parser's output passes them, a damaged payload that
             still parses may not).  One file of a list must not take the others' output along: that stream's output ends in
             front of this frame, the batch goes on (its rows keep running; nothing more of them is written) */
          if (!list_mode) die("the SBR kernels refused a frame", i);
          fprintf(stderr, "xaacdec_amd: stream %zu: the SBR kernels refused a frame: the stream ends here\n", (size_t)S.lo + si);
          refused[si] = 1;
        }
      }
    }
    if (pending.mono_twice) /* mono duplicated to stereo (api.c:3639-3660), from the back so that it can be done in place */
      for (long k = (long)N * 2048 - 1; k >= 0; k--) h_pcm[2 * k] = h_pcm[2 * k + 1] = h_pcm[k];
    if (pending.some_mono) { /* the same for the streams of a PS batch whose frame carried no PS: the bank pair wrote their left
                                samples into the interleaved rows and left the right ones alone */
      const std::vector<int32_t> &fl = st[pending.which].flags;
      for (int i = 0; i < N; i++)
        if (alive[(size_t)i] == 0 && fl[(size_t)i * 8 + 5] == 0)
          for (int k = 0; k < 2048; k++) h_pcm[(size_t)i * 4096 + 2 * k + 1] = h_pcm[(size_t)i * 4096 + 2 * k];
    }
    lap(2);
    const size_t skip = (!sbr && pending.first) ? (size_t)delay * out_ch : 0; /* the limiter's delay is cut from the first frame */
    /* (with -esbr:1 the reference's command line decoder does not write an SBR stream's first frame:
       test/decoder/ixheaacd_main.c:2181-2186) */
    if (!(esbr && pending.first))
      for (size_t i = 0; i < pcms.size(); i++)
        if (alive[i] == 0 && !refused[i]) pcms[i].insert(pcms[i].end(), h_pcm + i * per * out_ch + skip, h_pcm + (i + 1) * per * out_ch);
    for (int i = 1; verify && i < N; i++)
      mismatched += memcmp(h_pcm, h_pcm + (size_t)i * per * out_ch, (size_t)per * out_ch * 2) != 0;
    frames += st[pending.which].delivered;
    if (pending.first) t_first = std::chrono::steady_clock::now(); /* the first step also loads the kernels' code objects */
    pending.valid = false;
    lap(3);
  };
  start_parse();
  for (int step = 0;; step++) {
    const int which = step % (3 * T), slot = step & 1;
    if (step % T == 0) wait_parse(step / T);
    Staging &s = st[which];
    if (s.delivered == 0) break;
    int16_t *d_pcm = d_pcm2[slot], *d_mono = d_mono2[slot];
    int32_t *d_status = d_status2[slot];
    bool mono_twice = false, some_mono = false;
    if (s.delivered != N) {
      if (!list_mode) die("streams of different lengths in one batch");
      for (int i = 0; i < N; i++)
        if (s.status[(size_t)i] != 0 && !ended[(size_t)i]) { /* this stream is over: the other rows go on, its own run idle */
          ended[(size_t)i] = 1;
          if (!sbr) HIP(hipMemcpy(&lim_at_end[(size_t)i], d_lim + i, sizeof(xaac_limiter_state), hipMemcpyDeviceToHost)); /* (waits for the step before) */
        }
    }
    if (step % T == 0) start_parse(); /* the next group's frames are parsed while the GPU works on this one's */
    t_phase = std::chrono::steady_clock::now();
    { /* only the leading lines that are not zero in every delivered row go up, and what the device array still holds beyond
         them from the step before (the host rows are zero there) */
      const int width = s.lines > lines_held ? s.lines : lines_held;
      lines_held = s.lines;
      if (width >= 1024) HIP(hipMemcpyAsync(d_spec, s.spec, (size_t)NC * 4096, hipMemcpyHostToDevice, stream));
      else if (width > 0) HIP(hipMemcpy2DAsync(d_spec, 4096, s.spec, 4096, (size_t)width * 4, (size_t)NC, hipMemcpyHostToDevice, stream));
    }
    HIP(hipMemcpyAsync(d_ics, s.ics, (size_t)NC * 2, hipMemcpyHostToDevice, stream));
    lap(0);
    xaac_imdct_batch ib;
    memset(&ib, 0, sizeof(ib));
    ib.n_ch = NCD, ib.ch_fac = n_ch, ib.spec = d_spec, ib.ics = d_ics, ib.overlap = d_overlap, ib.state = d_ovl;
    if (!sbr) { /* AAC-LC: IMDCT -> limiter -> round16 (api.c:3662-3692) */
      ib.out32 = d_out32, ib.qshift_adj = d_qadj;
      XA(xaac_imdct_process_batch(ctx, &ib));
      xaac_limiter_batch lb;
      memset(&lb, 0, sizeof(lb));
      lb.n_streams = N, lb.frame_len = 1024, lb.samples = d_out32, lb.stride = 1024 * n_ch, lb.qshift_adj = d_qadj, lb.state = d_lim;
      lb.num_channels = n_ch, lb.pcm16 = d_pcm, lb.workspace = d_ws, lb.workspace_bytes = ws_bytes;
      XA(xaac_peak_limiter_process_batch(ctx, &lb));
    } else if (esbr) { /* Path A: IMDCT -> float planes -> eSBR chain (+ transposer, float PS) -> samples_sat */
      ib.pcm16 = d_core, ib.pcm_mode = XAAC_PCM_SBR;
      XA(xaac_imdct_process_batch(ctx, &ib));
      int resets = 0, with_ps = 0;
      for (int i = 0; i < N; i++)
        if (s.status[(size_t)i] == 0) resets += s.flags[(size_t)i * 8 + 1] != 0, with_ps += s.flags[(size_t)i * 8 + 5] != 0;
      /* streams with and without PS in one step: the float PS launch copies left to right for those without (esbr_ps_kernel.hip) */
      const size_t row = 64 * sizeof(float), st_pitch = sizeof(xaac_esbr_state), q_pitch = 2048 * sizeof(float);
      float *older_re = d_older, *older_im = d_older + (size_t)NC * 24 * 64;
      if (resets != 0 && hq) {
        /* ixheaacd_sbr_dec_reset with -esbr_hq:1 (sbrdecoder.c:175-236): the resetting channels gathered into a compact batch (all
           of them, or the few whose headers changed): ixheaacd_dft_hbe_data_reinit on the host (xaac_hbe_dft_state_reinit: sizes,
           windows, matrices; a configuration is shared by the channels whose band tables are the same), then the transposer's
           two runs over rows 8..39 and 40..71 of the QMF buffer as the frame before left it.  Its output rows are written whole
           (rows32): the second run's last eight become the state's ph rows. */
        std::vector<int> chs;
        for (int i = 0; i < NC; i++)
          if (s.status[(size_t)(i / n_ch)] == 0 && s.flags[(size_t)(i / n_ch) * 8 + 1] != 0) chs.push_back(i);
        const int nr = (int)chs.size();
        if (!d_q) d_q = dev<float>((size_t)NC * 2 * 2048), d_pv = dev<float>((size_t)NC * 2 * 2048);
        if (!d_idx) d_idx = dev<int32_t>((size_t)NC);
        float *q_re = d_q, *q_im = d_q + (size_t)NC * 2048, *pv_re = d_pv, *pv_im = d_pv + (size_t)NC * 2048;
        std::vector<int32_t> pitch((size_t)nr), slots((size_t)nr), ovs((size_t)nr);
        HIP(hipStreamSynchronize(stream));
        const auto d2d = [&](void *dst, const void *src, size_t bytes) { HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream)); };
        static thread_local xaac_hbe_dft_state h0;
        static thread_local xaac_hbe_dft_cfg c0;
        static thread_local float k_re[64 * 128], k_im[64 * 128];
        constexpr size_t kInts = sizeof(xaac_hbe_dft_state) - offsetof(xaac_hbe_dft_state, anal.analy_size); /* the integers behind the signals */
        for (int k = 0; k < nr; k++) {
          const int i = chs[(size_t)k];
          const xaac_sbr_header &hd = s.header[(size_t)i];
          memset(&h0, 0, sizeof(h0));
          h0.max_stretch = dft_ms[(size_t)i];
          if (xaac_hbe_dft_state_reinit(&h0, &c0, k_re, k_im, &hd)) die("the DFT transposer has no windows for the SBR band tables");
          dft_ms[(size_t)i] = h0.max_stretch;
          const std::string key(reinterpret_cast<const char *>(&hd.num_sf_bands[0]),
                                reinterpret_cast<const char *>(&hd.freq_band_tbl_noise[0]) - reinterpret_cast<const char *>(&hd.num_sf_bands[0]));
          const std::string key2 = key + std::string(reinterpret_cast<const char *>(&h0.max_stretch), 4);
          auto it = dft_cfgs.find(key2);
          if (it == dft_cfgs.end()) { /* a configuration no channel of the shard has had yet: its windows and matrices go up (synchronously: host temporaries) */
            const int slot = (int)dft_cfgs.size();
            if (slot >= NC) die("more DFT transposer configurations than channels");
            it = dft_cfgs.emplace(key2, slot).first;
            HIP(hipMemcpy(&d_dcfg[slot], &c0, sizeof(c0), hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_dcoef + (size_t)slot * 64 * 128, k_re, sizeof(k_re), hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_dcoef + ((size_t)NC + slot) * 64 * 128, k_im, sizeof(k_im), hipMemcpyHostToDevice));
          }
          dft_slot[(size_t)i] = slots[(size_t)k] = it->second;
          HIP(hipMemcpy(&d_dft[i].anal.analy_size, &h0.anal.analy_size, kInts, hipMemcpyHostToDevice));
          HIP(hipMemset(&d_dft[i].synth_buf[0], 0, sizeof(h0.synth_buf))); /* hbe_dft_trans.c:302 */
          d2d(&d_dft_tmp[k], &d_dft[i], sizeof(xaac_hbe_dft_state));
          pitch[(size_t)k] = s.reset_pitch[(size_t)(i / n_ch)];
          ovs[(size_t)k] = dft_ovs[(size_t)i];
          /* run 1: buffer rows 8..39 = the 24 older rows, then the state's first eight */
          d2d(q_re + (size_t)k * 2048, older_re + (size_t)i * 24 * 64, 24 * row);
          d2d(q_im + (size_t)k * 2048, older_im + (size_t)i * 24 * 64, 24 * row);
          d2d(q_re + (size_t)k * 2048 + 24 * 64, &d_estate[i].qmf_re[0][0], 8 * row);
          d2d(q_im + (size_t)k * 2048 + 24 * 64, &d_estate[i].qmf_im[0][0], 8 * row);
        }
        HIP(hipMemcpyAsync(d_idx, pitch.data(), (size_t)nr * 4, hipMemcpyHostToDevice, stream));
        HIP(hipMemcpyAsync(d_dslot_tmp, slots.data(), (size_t)nr * 4, hipMemcpyHostToDevice, stream));
        HIP(hipMemcpyAsync(d_dovs, ovs.data(), (size_t)nr * 4, hipMemcpyHostToDevice, stream));
        HIP(hipMemcpyAsync(d_dslot, dft_slot.data(), (size_t)NC * 4, hipMemcpyHostToDevice, stream));
        HIP(hipStreamSynchronize(stream)); /* (the vectors are host memory of this scope) */
        xaac_hbe_dft_apply_batch db;
        memset(&db, 0, sizeof(db));
        db.n_ch = nr, db.qmf_re = q_re, db.qmf_im = q_im, db.pitch_in_bins = d_idx, db.oversampling = d_dovs, db.cfg_tab = d_dcfg;
        db.coef_re = d_dcoef, db.coef_im = d_dcoef + (size_t)NC * 64 * 128, db.cfg = d_dslot_tmp, db.state = d_dft_tmp;
        db.pv_re = pv_re, db.pv_im = pv_im, db.status = d_status, db.rows32 = 1;
        XA(xaac_hbe_dft_apply_batch_run(ctx, &db));
        for (int k = 0; k < nr; k++) { /* run 2: buffer rows 40..71 */
          const int i = chs[(size_t)k];
          d2d(q_re + (size_t)k * 2048, &d_estate[i].qmf_re[8][0], 32 * row);
          d2d(q_im + (size_t)k * 2048, &d_estate[i].qmf_im[8][0], 32 * row);
        }
        XA(xaac_hbe_dft_apply_batch_run(ctx, &db));
        for (int k = 0; k < nr; k++) {
          const int i = chs[(size_t)k];
          d2d(&d_estate[i].ph_re[0][0], pv_re + (size_t)k * 2048 + 24 * 64, 8 * row);
          d2d(&d_estate[i].ph_im[0][0], pv_im + (size_t)k * 2048 + 24 * 64, 8 * row);
          d2d(&d_dft[i], &d_dft_tmp[k], sizeof(xaac_hbe_dft_state));
        }
      } else if (resets != 0 && resets != s.delivered) {
        /* Only some of the step's streams reset the SBR decoder (their headers changed: independent streams do that at
           different frames).  The same sequence as below for every stream at once, on those streams' channels gathered into
           a compact batch: their transposer states into d_hbe_tmp (new parameters from the band tables, delay lines
           cleared), their rows into the first slots of the scratch planes, the two transposer runs over that batch, states
           and ph rows back to their places.  The other streams' states are not touched. */
        static thread_local xaac_hbe_state h0;
        constexpr size_t kTail = sizeof(xaac_hbe_state) - offsetof(xaac_hbe_state, synth_size);
        if (hbe_tail.empty()) hbe_tail.assign((size_t)NC * kTail, 0);
        std::vector<int> chs;
        for (int i = 0; i < NC; i++)
          if (s.status[(size_t)(i / n_ch)] == 0 && s.flags[(size_t)(i / n_ch) * 8 + 1] != 0) chs.push_back(i);
        const int nr = (int)chs.size();
        if (!d_hbe_tmp) d_hbe_tmp = dev<xaac_hbe_state>((size_t)NC);
        if (!d_q) d_q = dev<float>((size_t)NC * 2 * 2048), d_pv = dev<float>((size_t)NC * 2 * 2048);
        if (!d_idx) d_idx = dev<int32_t>((size_t)NC);
        float *q_re = d_q, *q_im = d_q + (size_t)NC * 2048, *pv_re = d_pv, *pv_im = d_pv + (size_t)NC * 2048;
        std::vector<int32_t> pitch((size_t)nr);
        HIP(hipStreamSynchronize(stream));
        const auto d2d = [&](void *dst, const void *src, size_t bytes) { HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream)); };
        for (int k = 0; k < nr; k++) {
          const int i = chs[(size_t)k];
          xaac_hbe_state_init(&h0);
          memcpy(&h0.synth_size, &hbe_tail[(size_t)i * kTail], kTail);
          if (xaac_hbe_state_reinit(&h0, &s.header[(size_t)i])) die("the QMF transposer refused the SBR band tables");
          memcpy(&hbe_tail[(size_t)i * kTail], &h0.synth_size, kTail);
          d2d(&d_hbe_tmp[k], &d_hbe[i], sizeof(xaac_hbe_state));
          HIP(hipMemcpyAsync(&d_hbe_tmp[k].synth_size, &hbe_tail[(size_t)i * kTail], kTail, hipMemcpyHostToDevice, stream));
          HIP(hipMemsetAsync(&d_hbe_tmp[k].synth_buf[0], 0, sizeof(h0.synth_buf), stream));
          HIP(hipMemsetAsync(&d_hbe_tmp[k].analy_buf[0], 0, sizeof(h0.analy_buf), stream));
          pitch[(size_t)k] = s.reset_pitch[(size_t)(i / n_ch)];
          /* run 1: buffer rows 8..39 = the 24 older rows, then the state's first eight */
          d2d(q_re + (size_t)k * 2048, older_re + (size_t)i * 24 * 64, 24 * row);
          d2d(q_im + (size_t)k * 2048, older_im + (size_t)i * 24 * 64, 24 * row);
          d2d(q_re + (size_t)k * 2048 + 24 * 64, &d_estate[i].qmf_re[0][0], 8 * row);
          d2d(q_im + (size_t)k * 2048 + 24 * 64, &d_estate[i].qmf_im[0][0], 8 * row);
        }
        HIP(hipMemcpyAsync(d_idx, pitch.data(), (size_t)nr * 4, hipMemcpyHostToDevice, stream));
        HIP(hipStreamSynchronize(stream)); /* (pitch and the tails are host memory of this scope) */
        xaac_hbe_apply_batch_desc hb;
        memset(&hb, 0, sizeof(hb));
        hb.n_ch = nr, hb.qmf_re = q_re, hb.qmf_im = q_im, hb.state = d_hbe_tmp, hb.pv_re = pv_re, hb.pv_im = pv_im, hb.status = d_status;
        hb.pitch_in_bins = d_idx;
        hb.max_synth_size = hbe_hint();
        XA(xaac_hbe_apply_batch(ctx, &hb));
        for (int k = 0; k < nr; k++) { /* run 2: buffer rows 40..71; its output rows 24..31 start from the state's ph rows */
          const int i = chs[(size_t)k];
          d2d(q_re + (size_t)k * 2048, &d_estate[i].qmf_re[8][0], 32 * row);
          d2d(q_im + (size_t)k * 2048, &d_estate[i].qmf_im[8][0], 32 * row);
          d2d(pv_re + (size_t)k * 2048 + 24 * 64, &d_estate[i].ph_re[0][0], 8 * row);
          d2d(pv_im + (size_t)k * 2048 + 24 * 64, &d_estate[i].ph_im[0][0], 8 * row);
        }
        XA(xaac_hbe_apply_batch(ctx, &hb));
        for (int k = 0; k < nr; k++) {
          const int i = chs[(size_t)k];
          d2d(&d_estate[i].ph_re[0][0], pv_re + (size_t)k * 2048 + 24 * 64, 8 * row);
          d2d(&d_estate[i].ph_im[0][0], pv_im + (size_t)k * 2048 + 24 * 64, 8 * row);
          d2d(&d_hbe[i], &d_hbe_tmp[k], sizeof(xaac_hbe_state));
        }
      } else if (resets) {
        /* ixheaacd_sbr_dec_reset for Path A (sbrdecoder.c:175-236): the transposer's parameters from the new band tables (its
           two delay lines cleared, hbe_trans.c:102-222), then its two runs over rows 8..39 and 40..71 of the QMF buffer (the codec bank's num_time_slots is 32) as the
           frame before left it: rows 8..31 are what that frame found as its history rows 8..31 (d_older), rows 32..71 are
           the state's history.  The second run's last eight output rows are the state's ph rows (bands outside
           the transposer's range keep what they held). */
        static thread_local xaac_hbe_state h0;
        constexpr size_t kTail = sizeof(xaac_hbe_state) - offsetof(xaac_hbe_state, synth_size); /* the integers behind the buffers */
        if (hbe_tail.empty()) hbe_tail.assign((size_t)NC * kTail, 0);
        std::vector<uint8_t> &tail = hbe_tail;
        for (int i = 0; i < NC; i++) {
          xaac_hbe_state_init(&h0);
          memcpy(&h0.synth_size, &tail[(size_t)i * kTail], kTail);
          if (xaac_hbe_state_reinit(&h0, &s.header[(size_t)i])) die("the QMF transposer refused the SBR band tables");
          memcpy(&tail[(size_t)i * kTail], &h0.synth_size, kTail);
        }
        HIP(hipStreamSynchronize(stream));
        HIP(hipMemcpy2D(&d_hbe[0].synth_size, sizeof(xaac_hbe_state), tail.data(), kTail, kTail, (size_t)NC, hipMemcpyHostToDevice));
        HIP(hipMemset2DAsync(&d_hbe[0].synth_buf[0], sizeof(xaac_hbe_state), 0, sizeof(h0.synth_buf), (size_t)NC, stream));
        HIP(hipMemset2DAsync(&d_hbe[0].analy_buf[0], sizeof(xaac_hbe_state), 0, sizeof(h0.analy_buf), (size_t)NC, stream));
        if (!d_q) d_q = dev<float>((size_t)NC * 2 * 2048), d_pv = dev<float>((size_t)NC * 2 * 2048);
        float *q_re = d_q, *q_im = d_q + (size_t)NC * 2048, *pv_re = d_pv, *pv_im = d_pv + (size_t)NC * 2048;
        xaac_hbe_apply_batch_desc hb;
        memset(&hb, 0, sizeof(hb));
        hb.n_ch = NC, hb.qmf_re = q_re, hb.qmf_im = q_im, hb.state = d_hbe, hb.pv_re = pv_re, hb.pv_im = pv_im, hb.status = d_status;
        hb.max_synth_size = hbe_hint();
        {
          std::vector<int32_t> pitch((size_t)NC);
          for (int i = 0; i < NC; i++) pitch[(size_t)i] = s.reset_pitch[(size_t)(i / n_ch)];
          if (!d_idx) d_idx = dev<int32_t>((size_t)NC);
          HIP(hipMemcpy(d_idx, pitch.data(), (size_t)NC * 4, hipMemcpyHostToDevice));
          hb.pitch_in_bins = d_idx;
        }
        const auto copy_rows = [&](float *dst, size_t dpitch, const float *src, size_t spitch, int rows) {
          HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, rows * row, (size_t)NC, hipMemcpyDeviceToDevice, stream));
        };
        /* run 1: buffer rows 8..39 */
        copy_rows(q_re, q_pitch, older_re, 24 * row, 24);
        copy_rows(q_im, q_pitch, older_im, 24 * row, 24);
        copy_rows(q_re + 24 * 64, q_pitch, &d_estate[0].qmf_re[0][0], st_pitch, 8);
        copy_rows(q_im + 24 * 64, q_pitch, &d_estate[0].qmf_im[0][0], st_pitch, 8);
        XA(xaac_hbe_apply_batch(ctx, &hb));
        /* run 2: buffer rows 40..71; its output rows 24..31 start from the state's ph rows */
        copy_rows(q_re, q_pitch, &d_estate[0].qmf_re[8][0], st_pitch, 32);
        copy_rows(q_im, q_pitch, &d_estate[0].qmf_im[8][0], st_pitch, 32);
        copy_rows(pv_re + 24 * 64, q_pitch, &d_estate[0].ph_re[0][0], st_pitch, 8);
        copy_rows(pv_im + 24 * 64, q_pitch, &d_estate[0].ph_im[0][0], st_pitch, 8);
        XA(xaac_hbe_apply_batch(ctx, &hb));
        copy_rows(&d_estate[0].ph_re[0][0], st_pitch, pv_re + 24 * 64, q_pitch, 8);
        copy_rows(&d_estate[0].ph_im[0][0], st_pitch, pv_im + 24 * 64, q_pitch, 8);
      }
      /* what this frame finds as rows 8..31 of its history: the frame behind it may need them at a reset */
      HIP(hipMemcpy2DAsync(older_re, 24 * row, &d_estate[0].qmf_re[8][0], st_pitch, 24 * row, (size_t)NC, hipMemcpyDeviceToDevice, stream));
      HIP(hipMemcpy2DAsync(older_im, 24 * row, &d_estate[0].qmf_im[8][0], st_pitch, 24 * row, (size_t)NC, hipMemcpyDeviceToDevice, stream));
      HIP(hipMemcpyAsync(d_header, s.header, (size_t)NC * sizeof(xaac_sbr_header), hipMemcpyHostToDevice, stream));
      HIP(hipMemcpyAsync(d_frame, s.frame, (size_t)NC * sizeof(xaac_sbr_frame), hipMemcpyHostToDevice, stream));
      HIP(hipMemcpyAsync(d_eside, s.eside, (size_t)NC * sizeof(xaac_esbr_side), hipMemcpyHostToDevice, stream));
      xaac_esbr_core_in_batch cb = {NC, n_ch, d_core, d_fcore};
      XA(xaac_esbr_core_from_pcm16_batch(ctx, &cb));
      xaac_esbr_sbr_batch b;
      memset(&b, 0, sizeof(b));
      b.n_ch = NC, b.core = d_fcore, b.header = d_header, b.frame = d_frame, b.side = d_eside, b.state = d_estate, b.out = d_out_l;
      b.status = d_status, b.workspace = d_ws, b.workspace_bytes = ws_bytes;
      if (hq) {
        b.hbe_dft_state = d_dft, b.hbe_dft_cfg_tab = d_dcfg, b.hbe_dft_cfg = d_dslot;
        b.hbe_dft_coef_re = d_dcoef, b.hbe_dft_coef_im = d_dcoef + (size_t)NC * 64 * 128;
        for (int i = 0; i < NC; i++) /* the flag the transposer keeps for a reset that may follow */
          if (s.status[(size_t)(i / n_ch)] == 0 && s.frame[(size_t)i].apply_processing)
            dft_ovs[(size_t)i] = (s.eside[(size_t)i].harmonic_sbr & XAAC_ESBR_OVERSAMPLING) ? 1 : 0;
      } else {
        b.hbe_state = d_hbe;
        b.hbe_max_synth_size = hbe_hint();
      }
      xaac_esbr_pcm_out_batch ob = {N, 2048, d_out_l, d_out_l, d_pcm}; /* a mono channel twice (api.c:3639-3660) */
      if (with_ps) {
        HIP(hipMemcpyAsync(d_psf, s.ps, (size_t)N * sizeof(xaac_ps_frame), hipMemcpyHostToDevice, stream));
        b.ps_frame = d_psf, b.ps_state = d_eps, b.out_r = d_out_r;
        ob.right = d_out_r;
        some_mono = with_ps != s.delivered; /* streams without PS in this step: no right channel comes back for them (their right
                                               bank is left alone); their left samples are doubled on the host */
      } else if (n_ch == 2) {
        ob.stride = 4096, ob.right = d_out_l + 2048;
      }
      XA(xaac_esbr_sbr_process_batch(ctx, &b));
      XA(xaac_esbr_pcm16_from_float_batch(ctx, &ob));
    } else {
      ib.pcm16 = d_core, ib.pcm_mode = XAAC_PCM_SBR;
      XA(xaac_imdct_process_batch(ctx, &ib));
      HIP(hipMemcpyAsync(d_header, s.header, (size_t)NC * sizeof(xaac_sbr_header), hipMemcpyHostToDevice, stream));
      HIP(hipMemcpyAsync(d_frame, s.frame, (size_t)NC * sizeof(xaac_sbr_frame), hipMemcpyHostToDevice, stream));
      { /* frames that reset the SBR decoder or fall back to plain up-sampling rewrite a few words of the resident state: on
           the device, from the flag rows (a stream that is over keeps its last frame's flags: its row goes up as zeros) */
        bool any = false;
        for (int i = 0; i < N; i++) {
          const int32_t *f = &s.flags[(size_t)i * 8];
          any = any || (s.status[(size_t)i] == 0 && (f[1] || f[3]));
        }
        if (any) {
          /* h_flags is one pinned buffer: an earlier step's copy up may still be reading it, so the stream is drained BEFORE the
             rows are rewritten (rewriting first and draining behind, as this did, could hand that copy the new rows) */
          HIP(hipStreamSynchronize(stream));
          for (int i = 0; i < N; i++) {
            const int32_t *f = &s.flags[(size_t)i * 8];
            const bool live = s.status[(size_t)i] == 0;
            for (int k = 0; k < 8; k++) h_flags[(size_t)i * 8 + k] = live ? f[k] : 0;
          }
          HIP(hipMemcpyAsync(d_flags, h_flags, (size_t)N * 8 * 4, hipMemcpyHostToDevice, stream));
          xaac_sbr_apply_side_batch ab;
          memset(&ab, 0, sizeof(ab));
          ab.n_streams = N, ab.ch_fac = n_ch, ab.header = d_header, ab.flags = d_flags, ab.state = d_state;
          ab.ps_state = n_ch == 1 ? d_ps_state : nullptr;
          XA(xaac_sbr_state_apply_side_batch(ctx, &ab));
        }
      }
      if (n_ch == 2) {
        xaac_sbr_lp_batch b;
        memset(&b, 0, sizeof(b));
        b.n_ch = NC, b.in_ch_fac = 2, b.out_ch_fac = 2, b.pcm_in = d_core, b.header = d_header, b.frame = d_frame;
        b.state = d_state, b.pcm_out = d_pcm, b.status = d_status, b.workspace = d_ws, b.workspace_bytes = ws_bytes;
        XA(xaac_sbr_lp_process_batch(ctx, &b));
      } else {
        int with_ps = 0, starts = 0;
        std::vector<int32_t> idx;
        for (int i = 0; i < N; i++) {
          with_ps += s.status[(size_t)i] == 0 && s.flags[(size_t)i * 8 + 5] != 0;
          if (s.status[(size_t)i] == 0 && s.flags[(size_t)i * 8 + 6]) idx.push_back(i), starts++;
        }
        /* streams with and without parametric stereo in one step (independent HE-AAC / HE-AACv2 streams, or streams whose PS
           starts at different frames): the batch runs with the PS launch, which passes a stream without PS through as the mono
           frame it is (sbr_ps_kernel.hip: sbr_dec.c:1246) -- its right bank stays idle, its left samples are doubled on the host */
        some_mono = with_ps != 0 && with_ps != s.delivered;
        xaac_sbr_hq_batch b;
        memset(&b, 0, sizeof(b));
        b.n_ch = N, b.in_ch_fac = 1, b.out_ch_fac = 1, b.pcm_in = d_core, b.header = d_header, b.frame = d_frame;
        b.state = d_state, b.status = d_status, b.workspace = d_ws, b.workspace_bytes = ws_bytes;
        if (with_ps) {
          if (starts) { /* the right bank starts from the left one's filter states (sbrdecoder.c:762-775) */
            HIP(hipMemcpyAsync(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, stream));
            HIP(hipStreamSynchronize(stream));
            xaac_sbr_handover_batch hb;
            memset(&hb, 0, sizeof(hb));
            hb.n = starts, hb.mode = XAAC_HANDOVER_PS_START, hb.src = d_idx, hb.dst = d_idx, hb.state = d_state, hb.ps_state = d_ps_state;
            XA(xaac_sbr_state_handover(ctx, &hb));
          }
          HIP(hipMemcpyAsync(d_psf, s.ps, (size_t)N * sizeof(xaac_ps_frame), hipMemcpyHostToDevice, stream));
          b.ps_frame = d_psf, b.ps_state = d_ps_state, b.pcm_out = d_pcm;
        } else {
          b.pcm_out = d_mono;
        }
        XA(xaac_sbr_hq_process_batch(ctx, &b));
        mono_twice = !with_ps;
      }
    }
    lap(1);
    HIP(hipEventRecord(ev_kernels[slot], stream));
    HIP(hipStreamWaitEvent(down, ev_kernels[slot], 0));
    if (mono_twice) HIP(hipMemcpyAsync(h_pcm2[slot], d_mono, (size_t)N * 2048 * 2, hipMemcpyDeviceToHost, down));
    else HIP(hipMemcpyAsync(h_pcm2[slot], d_pcm, (size_t)N * per * out_ch * 2, hipMemcpyDeviceToHost, down));
    if (sbr) HIP(hipMemcpyAsync(h_status2[slot], d_status, (size_t)NC * 4, hipMemcpyDeviceToHost, down));
    HIP(hipEventRecord(ev_down[slot], down));
    consume(); /* the step before this one: its PCM has been on its way while this step's work was queued */
    pending = {true, mono_twice, first, slot, which, some_mono};
    if (profile) consume(); /* phase timing wants one step at a time */
    first = false;
  }
  consume();
  {
    std::lock_guard<std::mutex> lk(mu);
    quit = true;
    cv.notify_all();
  }
  worker.join();
  const auto t_end = std::chrono::steady_clock::now();
  S.wall = std::chrono::duration<double>(t_end - t_all).count();
  S.steady = std::chrono::duration<double>(t_end - t_first).count();
  if (!sbr) { /* the limiter's delay line holds the last attack_time_samples samples: api.c:2824-2866 */
    static thread_local xaac_limiter_state l;
    for (size_t i = 0; i < pcms.size(); i++) {
      if (list_mode && ended[i]) l = lim_at_end[i];
      else HIP(hipMemcpy(&l, d_lim + i, sizeof(l), hipMemcpyDeviceToHost));
      const uint32_t att = l.attack_time_samples, at = l.delayed_input_index;
      for (uint32_t k = 0; k < att; k++)
        for (int c = 0; c < n_ch; c++) {
          const float v = l.delayed_input[(size_t)((at + k) % att) * n_ch + c];
          const int64_t w = (v >= 2147483648.0f || v < -2147483648.0f || v != v) ? INT32_MIN : (int64_t)v; /* (WORD32)v as x86 has it */
          int64_t r = w + 0x8000;
          if (r > INT32_MAX) r = INT32_MAX;
          pcms[i].push_back((int16_t)(r >> 16));
        }
    }
  }
}

}  // namespace

int main(int argc, char **argv) {
  std::string in, out, ilist, odir;
  int copies = 1, threads = 0, quiet = 0, verify = 0, profile = 0, esbr = 1, gpus = 1, device0 = 0, plan = 0, wrap = 0, hq = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a.rfind("-ifile:", 0) == 0) in = a.substr(7);
    else if (a.rfind("-ofile:", 0) == 0) out = a.substr(7);
    else if (a.rfind("-ilist:", 0) == 0) ilist = a.substr(7);
    else if (a.rfind("-odir:", 0) == 0) odir = a.substr(6);
    else if (a.rfind("-copies:", 0) == 0) copies = atoi(a.c_str() + 8);
    else if (a.rfind("-threads:", 0) == 0) threads = atoi(a.c_str() + 9);
    else if (a.rfind("-gpus:", 0) == 0) gpus = atoi(a.c_str() + 6);
    else if (a.rfind("-device:", 0) == 0) device0 = atoi(a.c_str() + 8);
    else if (a == "-plan") plan = 1;
    else if (a == "-wrap_devices") wrap = 1; /* shard r on device (k + r) mod the node's count: the -gpus host path on a box with fewer devices */
    else if (a == "-quiet") quiet = 1;
    else if (a == "-verify") verify = 1;
    else if (a == "-profile") profile = 1; /* synchronise behind every phase of a step and report the seconds spent in each */
    else if (a == "-esbr:0") esbr = 0;
    else if (a == "-esbr:1") esbr = 1;
    else if (a == "-esbr_hq:1") hq = 1;
    else if (a == "-esbr_hq:0") hq = 0;
    else if (a.rfind("-esbr", 0) == 0) die("-esbr:0 or -esbr:1 (and -esbr_hq:0 or -esbr_hq:1)");
  }
  std::vector<std::string> inputs;
  if (!ilist.empty()) {
    FILE *f = fopen(ilist.c_str(), "r");
    if (!f) die("fopen(-ilist)");
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
      std::string sline(line);
      while (!sline.empty() && (sline.back() == '\n' || sline.back() == '\r' || sline.back() == ' ')) sline.pop_back();
      if (!sline.empty()) inputs.push_back(sline);
    }
    fclose(f);
    if (inputs.empty() || odir.empty()) die("-ilist needs paths and -odir");
    copies = 1, verify = 0;
  } else if (!in.empty()) {
    inputs.push_back(in);
  }
  if (inputs.empty() || (ilist.empty() && out.empty() && !plan) || copies < 1 || gpus < 1 || device0 < 0) {
    fprintf(stderr, "usage: xaacdec_amd -ifile:<in.aac> -ofile:<out.wav> [-esbr:0|1] [-esbr_hq:0|1] [-copies:N] [-threads:T] [-gpus:G] [-device:k] [-plan] [-quiet]\n");
    return 1;
  }
  std::vector<std::vector<uint8_t>> datas(inputs.size());
  for (size_t k = 0; k < inputs.size(); k++) {
    FILE *f = fopen(inputs[k].c_str(), "rb");
    if (!f) die("fopen(input)");
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    datas[k].resize((size_t)n + 16);
    if (fread(datas[k].data(), 1, (size_t)n, f) != (size_t)n) die("fread");
    fclose(f);
    datas[k].resize((size_t)n);
  }
  const std::vector<uint8_t> &data = datas[0];
  /* a look at frame 0: channels, SBR or not (api.c:3369-3373: the SBR tools run for frames with an SBR payload; a stream at
     24 kHz and below has an SBR decoder object by implicit signalling, api.c:2160, which is never called without payloads) */
  xaac_adts_header hdr;
  if (xaac_adts_parse_header(data.data(), data.size(), &hdr)) die("ADTS header");
  int n_ch, sbr;
  {
    xaac_parser *probe = nullptr;
    XA(xaac_parser_create(&probe));
    std::vector<xaac_core_frame> cf(1);
    size_t used = 0;
    const int32_t rc = xaac_parse_adts_frame(probe, data.data(), data.size(), 1, cf.data(), &used);
    if (rc) die("first frame", rc);
    n_ch = cf[0].n_ch;
    sbr = cf[0].sbr_bytes > 0;
    xaac_parser_destroy(probe);
    for (size_t k = 1; k < datas.size(); k++) { /* -ilist: one kind of stream per batch */
      xaac_adts_header h2;
      if (xaac_adts_parse_header(datas[k].data(), datas[k].size(), &h2)) die("ADTS header");
      XA(xaac_parser_create(&probe));
      if (xaac_parse_adts_frame(probe, datas[k].data(), datas[k].size(), 1, cf.data(), &used)) die("first frame");
      if (h2.sampling_rate != hdr.sampling_rate || cf[0].n_ch != n_ch || (cf[0].sbr_bytes > 0) != (sbr != 0))
        die("-ilist: streams of different kinds (sampling rate, channels, SBR) in one batch");
      xaac_parser_destroy(probe);
    }
  }
  if (!sbr) esbr = 0; /* AAC-LC streams decode the same either way */
  const int out_ch = sbr ? 2 : n_ch; /* SBR streams come out in stereo (PS, or the mono column twice); AAC-LC as coded */
  const int N = ilist.empty() ? copies : (int)datas.size(), rate = hdr.sampling_rate, out_rate = sbr ? 2 * rate : rate, per = sbr ? 2048 : 1024;


  /* the split: contiguous stream ranges over the devices, as bench.py --gpus N splits over ranks (a shard without streams is
     not started: -gpus larger than the batch uses as many devices as there are streams) */
  const bool list_mode = !ilist.empty();
  if (gpus > N) gpus = N;
  std::vector<Shard> shards((size_t)gpus);
  for (int r = 0; r < gpus; r++) {
    int lo, hi;
    shard_range(N, r, gpus, &lo, &hi);
    shards[(size_t)r].device = device0 + r, shards[(size_t)r].lo = lo, shards[(size_t)r].n = hi - lo;
  }
  if (plan) { /* nothing below this line runs: no HIP call has been made */
    printf("{\"streams\": %d, \"gpus\": %d, \"channels\": %d, \"sbr\": %d, \"esbr\": %d, \"shards\": [", N, gpus, n_ch, sbr, esbr);
    for (int r = 0; r < gpus; r++)
      printf("%s{\"device\": %d, \"lo\": %d, \"n\": %d}", r ? ", " : "", shards[(size_t)r].device, shards[(size_t)r].lo, shards[(size_t)r].n);
    printf("]}\n");
    return 0;
  }
  {
    int have = 0;
    HIP(hipGetDeviceCount(&have));
    if (wrap && have > 0)
      for (Shard &S : shards) S.device %= have;
    else if (device0 + gpus > have) {
      fprintf(stderr, "xaacdec_amd: -device:%d -gpus:%d asks for devices %d..%d, the node has %d\n", device0, gpus, device0, device0 + gpus - 1, have);
      return 2;
    }
  }
  Job J;
  J.datas = std::move(datas);
  J.hdr = hdr;
  J.hq = hq;
  J.n_ch = n_ch, J.sbr = sbr, J.esbr = esbr, J.out_ch = out_ch, J.rate = rate, J.out_rate = out_rate, J.per = per;
  J.threads = threads, J.verify = verify, J.profile = profile, J.list_mode = list_mode;
  const auto t_run = std::chrono::steady_clock::now();
  if (gpus == 1) {
    decode_shard(J, shards[0]);
  } else {
    std::vector<std::thread> team;
    for (int r = 0; r < gpus; r++) team.emplace_back([&, r] { decode_shard(J, shards[(size_t)r]); });
    for (auto &t : team) t.join();
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count();
  long frames = 0, mismatched = 0, first_frames = 0;
  double parse_s = 0, steady = 0, phase_s[4] = {0, 0, 0, 0};
  for (const Shard &S : shards) {
    frames += S.frames, mismatched += S.mismatched, first_frames += S.first_frames;
    parse_s = S.parse_s > parse_s ? S.parse_s : parse_s;
    steady = S.steady > steady ? S.steady : steady;
    for (int k = 0; k < 4; k++) phase_s[k] = S.phase_s[k] > phase_s[k] ? S.phase_s[k] : phase_s[k];
    /* -verify: every shard's copies were compared with its first one; the shards' first ones with the first shard's */
    if (verify && !list_mode && &S != &shards[0]) mismatched += S.pcms[0] != shards[0].pcms[0];
  }
  const std::vector<int16_t> &pcm = shards[0].pcms[0];
  if (list_mode) {
    for (const Shard &S : shards)
    for (size_t i = 0; i < S.pcms.size(); i++) {
      std::string base = inputs[(size_t)S.lo + i];
      const size_t slash = base.find_last_of('/');
      if (slash != std::string::npos) base = base.substr(slash + 1);
      const size_t dot = base.find_last_of('.');
      if (dot != std::string::npos) base = base.substr(0, dot);
      write_wav(odir + "/" + base + ".wav", S.pcms[i], out_ch, out_rate);
    }
  } else
  write_wav(out, pcm, out_ch, out_rate);
  if (!quiet && gpus > 1) { /* (the run's own line stays the last one) */
    printf("{\"per_gpu_frames_per_s\": [");
    for (size_t r = 0; r < shards.size(); r++) printf("%s%.1f", r ? ", " : "", shards[r].wall > 0 ? shards[r].frames / shards[r].wall : 0.0);
    printf("]}\n");
  }
  if (!quiet)
    printf("{\"frames\": %ld, \"streams\": %d, \"wall_s\": %.4f, \"parse_s\": %.4f, \"frames_per_s\": %.1f, "
           "\"frames_per_s_after_first_step\": %.1f, \"mismatched_copies\": %ld, \"samples\": %zu, \"rate\": %d, \"sbr\": %d, "
           "\"channels\": %d, \"esbr\": %d, \"gpus\": %d}\n",
           frames, N, wall, parse_s, frames / wall, frames > first_frames && steady > 0 ? (frames - first_frames) / steady : 0.0, mismatched,
           pcm.size() / out_ch, out_rate, sbr, n_ch, esbr, gpus);
  if (profile)
    printf("{\"h2d_s\": %.4f, \"kernels_s\": %.4f, \"d2h_s\": %.4f, \"host_pcm_s\": %.4f}\n", phase_s[0], phase_s[1], phase_s[2], phase_s[3]);
  return mismatched ? 3 : 0;
}
