// A model of HIP streams and events for the library's host scheduler (basebandboard_amd/csrc/bbb_api.hip compiled for the host
// against tests/sched_model/hip/hip_runtime.h).  No GPU, no kernels: every launch entry point of csrc/awgn_launch.hpp /
// bbb_common.hpp is replaced by a stub that records WHICH device buffers the real kernel reads and writes, on which stream.
// The model keeps a vector clock per stream, per event and for the host, and checks every access:
//   a write must come after the buffer's last write and after every read since then, a read after the last write
// where "after" is the order streams, hipStreamWaitEvent and the host's synchronisations establish -- exactly the guarantees
// the real runtime gives.  An access that is not ordered is what a data race on the GPU is made of, whether or not a given
// run of the hardware shows it (the round-3 race showed on seed 3 of 3).
// "Device memory" is address space only (mmap PROT_NONE): the host scheduler must never dereference it.
#include "model.hpp"

#include <sys/mman.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../basebandboard_amd/csrc/bbb_common.hpp"
#include "../../basebandboard_amd/csrc/awgn_launch.hpp"

struct MockStream { int id; model::VC vc; bool destroyed = false; };
struct MockEvent { bool recorded = false; model::VC vc; };

namespace model {

static std::mutex g_mu;                         // (the library's sweep threads may call in concurrently)
static std::vector<MockStream *> &g_streams = *new std::vector<MockStream *>;     // id -> stream; id 0 = the null stream (never destroyed: streams outlive statics)
static VC g_host;                               // what the host has synchronised with
static std::map<char *, Buffer> g_buffers;      // by base address
static std::vector<std::string> g_errors;
static std::vector<std::string> g_trace;        // the operations since the last reset (printed with an error)
static uint64_t g_seq = 0, g_nops = 0;
static int g_device = 0;

static void join(VC &a, const VC &b) {
    if (a.size() < b.size()) a.resize(b.size(), 0);
    for (size_t i = 0; i < b.size(); i++) a[i] = std::max(a[i], b[i]);
}
static bool ordered_before(const Access &a, const VC &t) {      // a happens-before an operation stamped t
    return (size_t)a.stream < t.size() && a.vc[(size_t)a.stream] <= t[(size_t)a.stream];
}

static MockStream *stream_of(hipStream_t s) {
    if (s) return s;
    if (g_streams.empty()) { g_streams.push_back(new MockStream{0, VC()}); }
    return g_streams[0];
}

static void error(const std::string &what);
// Stream LIFETIMES (round 5; the round-4 crash was a cache that remembered a hipStream_t and synchronised with it after its
// owner had destroyed it -- ordering alone cannot see that): hipStreamDestroy poisons the handle, and every later use of it
// -- a launch, a record, a wait, a synchronisation, a query -- is reported.  (The object itself is kept: the report must not crash.)
static MockStream *live_stream(hipStream_t s, const char *call) {
    MockStream *m = stream_of(s);
    if (m->destroyed) error(std::string(call) + " on stream " + std::to_string(m->id) + ", which was destroyed (use after hipStreamDestroy)");
    return m;
}

static Buffer *buffer_of(const void *p) {
    if (!p) return nullptr;
    auto it = g_buffers.upper_bound((char *)p);
    if (it == g_buffers.begin()) return nullptr;
    --it;
    Buffer &b = it->second;
    return ((char *)p < b.base + b.size) ? &b : nullptr;
}

static void error(const std::string &what) {
    if (g_errors.size() < 20) {
        std::string e = what + "\n    operations since the last reset:";
        const size_t n0 = g_trace.size() > 60 ? g_trace.size() - 60 : 0;
        for (size_t i = n0; i < g_trace.size(); i++) e += "\n      " + g_trace[i];
        g_errors.push_back(e);
    }
}

static std::string describe(const Buffer &b) { return "buffer #" + std::to_string(b.id) + " (" + std::to_string(b.size) + " B, " + b.tag + ")"; }

// one operation on a stream
void op(hipStream_t hs, const std::string &what, std::initializer_list<const void *> reads, std::initializer_list<const void *> writes) {
    std::lock_guard<std::mutex> g(g_mu);
    MockStream *s = live_stream(hs, what.c_str());
    join(s->vc, g_host);                                         // issued by the host now: after everything it has waited for
    if (s->vc.size() <= (size_t)s->id) s->vc.resize((size_t)s->id + 1, 0);
    s->vc[(size_t)s->id]++;
    const VC t = s->vc;
    g_nops++;
    g_trace.push_back("[" + std::to_string(++g_seq) + "] stream " + std::to_string(s->id) + ": " + what);
    auto access = [&](const void *p, bool write) {
        if (!p) return;
        Buffer *b = buffer_of(p);
        if (!b) { error(what + ": pointer " + std::to_string((uintptr_t)p) + " is not inside a live device allocation"); return; }
        if (b->freed) { error(what + ": " + describe(*b) + " was freed"); return; }
        const Access me{t, s->id, what, g_seq};
        if (b->has_write && !ordered_before(b->last_write, t))
            error(std::string(write ? "WRITE" : "READ") + " of " + describe(*b) + " by [" + std::to_string(g_seq) + "] " + what + " (stream " +
                  std::to_string(s->id) + ") is not ordered after its last WRITE by [" + std::to_string(b->last_write.seq) + "] " +
                  b->last_write.what + " (stream " + std::to_string(b->last_write.stream) + ")");
        if (write) {
            for (const Access &r : b->reads)
                if (!ordered_before(r, t))
                    error("WRITE of " + describe(*b) + " by [" + std::to_string(g_seq) + "] " + what + " (stream " + std::to_string(s->id) +
                          ") is not ordered after a READ by [" + std::to_string(r.seq) + "] " + r.what + " (stream " + std::to_string(r.stream) + ")");
            b->last_write = me; b->has_write = true; b->reads.clear();
        } else {
            // keep one read per stream (the latest covers the earlier ones of the same stream)
            bool replaced = false;
            for (Access &r : b->reads)
                if (r.stream == s->id) { r = me; replaced = true; break; }
            if (!replaced) b->reads.push_back(me);
        }
    };
    for (const void *p : reads) access(p, false);
    for (const void *p : writes) access(p, true);
}

void host_note(const std::string &what) {
    std::lock_guard<std::mutex> g(g_mu);
    g_trace.push_back("-- " + what);
}

void reset_trace() {
    std::lock_guard<std::mutex> g(g_mu);
    g_trace.clear();
}

const std::vector<std::string> &errors() { return g_errors; }
void clear_errors() { g_errors.clear(); }
uint64_t ops_checked() { return g_nops; }

void tag(const void *p, const std::string &t) {
    std::lock_guard<std::mutex> g(g_mu);
    if (Buffer *b = buffer_of(p)) b->tag = t;
}

}  // namespace model

using namespace model;

// ---- the runtime ---------------------------------------------------------------------------------------------------------
extern "C" {

const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 2; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { g_device = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::memset(p, 0, sizeof *p); std::strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }

hipError_t hipDeviceSynchronize(void) {
    std::lock_guard<std::mutex> g(g_mu);
    for (MockStream *s : g_streams) join(g_host, s->vc);
    g_trace.push_back("-- hipDeviceSynchronize");
    return hipSuccess;
}

static int g_next_buffer = 0;
hipError_t hipMalloc(void **p, size_t n) {
    if (n == 0) n = 1;
    const size_t len = (n + 4095) & ~(size_t)4095;
    void *m = mmap(nullptr, len, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(g_mu);
    Buffer b;
    b.base = (char *)m; b.size = n; b.maplen = len; b.id = ++g_next_buffer;
    g_buffers[b.base] = b;
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_buffers.find((char *)p);
    if (it == g_buffers.end()) { error("hipFree of a pointer that is not the base of a live allocation"); return hipErrorInvalidValue; }
    // (the real call waits for the device: everything issued so far is done before the memory goes away)
    for (MockStream *s : g_streams) join(g_host, s->vc);
    g_trace.push_back("-- hipFree of buffer #" + std::to_string(it->second.id) + " (" + it->second.tag + ")");
    munmap(it->second.base, it->second.maplen);
    g_buffers.erase(it);
    return hipSuccess;
}
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFreeAsync(void *p, hipStream_t s) {
    if (!p) return hipSuccess;
    op(s, "hipFreeAsync", {}, {p});                               // ordered like a write: every user must be in front of it on this stream
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_buffers.find((char *)p);
    if (it != g_buffers.end()) { munmap(it->second.base, it->second.maplen); g_buffers.erase(it); }
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s) {
    switch (k) {
    case hipMemcpyHostToDevice: op(s, "memcpy H2D", {}, {dst}); break;
    case hipMemcpyDeviceToHost: op(s, "memcpy D2H", {src}, {}); std::memset(dst, 0, n); break;     // (the model has no values: the host reads zeros)
    case hipMemcpyDeviceToDevice: op(s, "memcpy D2D", {src}, {dst}); break;
    default: std::memmove(dst, src, n); break;
    }
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) {
    hipMemcpyAsync(dst, src, n, k, nullptr);
    std::lock_guard<std::mutex> g(g_mu);
    join(g_host, stream_of(nullptr)->vc);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int, size_t, hipStream_t s) { op(s, "memset", {}, {p}); return hipSuccess; }
hipError_t hipMemset(void *p, int v, size_t n) {
    hipMemsetAsync(p, v, n, nullptr);
    std::lock_guard<std::mutex> g(g_mu);
    join(g_host, stream_of(nullptr)->vc);
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    std::lock_guard<std::mutex> g(g_mu);
    stream_of(nullptr);
    MockStream *m = new MockStream{(int)g_streams.size(), VC()};
    g_streams.push_back(m);
    *s = m;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {                          // (the object is kept: a destroyed stream's work still completes)
    std::lock_guard<std::mutex> g(g_mu);
    if (!s) { error("hipStreamDestroy of the null stream"); return hipErrorInvalidValue; }
    MockStream *m = live_stream(s, "hipStreamDestroy");
    m->destroyed = true;
    g_trace.push_back("-- hipStreamDestroy(stream " + std::to_string(m->id) + ")");
    return hipSuccess;
}
// hipStreamQuery: hipSuccess means the host has SEEN everything on the stream complete (it is then synchronised with it);
// work the host has not waited for may or may not be done -- the model answers either way, so that both branches of a caller
// that asks are explored
static uint64_t g_coin = 0x9E3779B97F4A7C15ull;
hipError_t hipStreamQuery(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    MockStream *m = live_stream(s, "hipStreamQuery");
    bool pending = false;
    for (size_t i = 0; i < m->vc.size(); i++)
        if (m->vc[i] > (i < g_host.size() ? g_host[i] : 0u)) pending = true;
    if (pending) {
        g_coin ^= g_coin << 13; g_coin ^= g_coin >> 7; g_coin ^= g_coin << 17;
        if (g_coin & 1) { g_trace.push_back("-- hipStreamQuery(stream " + std::to_string(m->id) + "): not ready"); return hipErrorNotReady; }
        join(g_host, m->vc);
    }
    g_trace.push_back("-- hipStreamQuery(stream " + std::to_string(m->id) + "): done");
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    MockStream *m = live_stream(s, "hipStreamSynchronize");
    join(g_host, m->vc);
    g_trace.push_back("-- hipStreamSynchronize(stream " + std::to_string(m->id) + ")");
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    std::lock_guard<std::mutex> g(g_mu);
    MockStream *m = live_stream(s, "hipStreamWaitEvent");
    if (e && e->recorded) join(m->vc, e->vc);
    g_trace.push_back("-- stream " + std::to_string(m->id) + " waits for event " + std::to_string((uintptr_t)e & 0xffff) + (e && e->recorded ? "" : " (never recorded: no wait)"));
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new MockEvent; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new MockEvent; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    MockStream *m = live_stream(s, "hipEventRecord");
    join(m->vc, g_host);
    e->recorded = true;
    e->vc = m->vc;
    g_trace.push_back("-- event " + std::to_string((uintptr_t)e & 0xffff) + " recorded on stream " + std::to_string(m->id));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    std::lock_guard<std::mutex> g(g_mu);
    if (e && e->recorded) join(g_host, e->vc);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }

}  // extern "C"

// ---- the launch entry points: what each kernel (chain) reads and writes ----------------------------------------------------
namespace bbb {

bool awgn256_matches(int k, const uint16_t *, const uint32_t *) { return k == 256; }       // (the driver uses the shipped n256 matrix)
bool awgn512p_matches(int, const uint16_t *, const uint32_t *) { return false; }
int awgn_small_matches(int, const uint16_t *, const uint32_t *) { return 0; }
int prbs_state_at_host(int k, uint64_t init_state, uint64_t nbits, uint64_t *state) {
    const uint64_t mask = (1ull << k) - 1ull;
    uint64_t s = (init_state * 0x9E3779B97F4A7C15ull + nbits) & mask;
    *state = s ? s : 1;                                                                      // (any non-zero state: values do not matter here)
    return BBB_OK;
}

int awgn_seed_launch(int, const uint32_t *d_tabs, const uint32_t *, uint64_t, uint32_t *d_states, uint64_t, unsigned, uint32_t *d_planes,
                     hipStream_t st, int, int) {
    op(st, "seeding (levels + bitslice)", {d_tabs, d_states}, {d_states});
    op(st, "bitslice", {d_states}, {d_planes});
    return BBB_OK;
}
int awgn_seed_head_launch(int, const uint32_t *d_tabs, const uint32_t *, uint64_t, uint32_t *d_states, hipStream_t st, const PrbsSeedRide *ride) {
    if (ride) op(st, "seed_head_kernel (+ PRBS lanes)", {d_tabs, ride->d_tabs}, {d_states, ride->d_planes});
    else op(st, "seed_head_kernel", {d_tabs}, {d_states});
    return BBB_OK;
}
int awgn_seed_tail_planes_launch(int, const uint32_t *d_top, uint64_t, const uint32_t *d_states, unsigned, uint32_t *d_planes, hipStream_t st) {
    op(st, "seed_tail_planes_kernel", {d_top, d_states}, {d_planes});
    return BBB_OK;
}
int prbs_seed_planes_launch(int, const uint32_t *d_tabs, const uint32_t *, uint64_t, uint32_t *d_states, unsigned, uint32_t *d_planes, hipStream_t st) {
    op(st, "PRBS seeding", {d_tabs}, {d_states});
    op(st, "PRBS bitslice", {d_states}, {d_planes});
    return BBB_OK;
}
int prbs_seed_lanes_launch(int, const uint32_t *d_tabs, const uint32_t *, const uint32_t *, uint64_t, unsigned, uint32_t *d_planes, hipStream_t st) {
    op(st, "PRBS seeding (one state per lane + Q steps)", {d_tabs}, {d_planes});
    return BBB_OK;
}
int awgn512p_fill_launch(const uint32_t *d_planes, int16_t *dst, uint64_t, unsigned, uint64_t, unsigned, hipStream_t st) {
    op(st, "awgn512p kernel", {d_planes}, {dst});
    return BBB_OK;
}
int bitslice512p_launch(const uint32_t *d_states, uint64_t, uint64_t, unsigned, uint32_t *d_planes, hipStream_t st) {
    op(st, "bitslice512p", {d_states}, {d_planes});
    return BBB_OK;
}
int awgn256_fill_launch(const uint32_t *d_planes, int8_t *dst, uint64_t, unsigned, uint64_t, unsigned, hipStream_t st) {
    op(st, "awgn256_kernel (one-kernel form)", {d_planes}, {dst});
    return BBB_OK;
}
int awgn256_planes_launch(const uint32_t *d_planes, void *stage, unsigned, unsigned, hipStream_t st, bool small) {
    op(st, small ? "awgn256_planes_kernel<small>" : "awgn256_planes_kernel", {d_planes}, {stage});
    return BBB_OK;
}
int unplane_launch(const void *stage, void *dst, uint64_t win_lo, uint64_t nbytes, unsigned, uint64_t, unsigned, hipStream_t st) {
    op(st, "unplane_kernel (mover) window [" + std::to_string(win_lo) + ", +" + std::to_string(nbytes) + ")", {stage}, {dst});
    return BBB_OK;
}
int unplane_tx_launch(const void *stage, int16_t *dst, uint64_t, uint64_t, unsigned, uint64_t, unsigned, const int16_t *, const uint32_t *d_bits,
                      uint32_t, uint32_t, uint32_t, int, int, int use_bits, hipStream_t st) {
    if (use_bits) op(st, "unplane_kernel<TX> (shaping mover)", {stage, d_bits}, {dst});
    else op(st, "unplane_kernel<TX> (shaping mover)", {stage}, {dst});
    return BBB_OK;
}
int awgn256_tx_launch(const uint32_t *d_planes, int16_t *dst, uint64_t, unsigned, uint64_t, unsigned, const int16_t *, const uint32_t *d_bits,
                      uint32_t, uint32_t, uint32_t, int, int, int use_bits, hipStream_t st) {
    if (use_bits) op(st, "awgn256_kernel<TX>", {d_planes, d_bits}, {dst});
    else op(st, "awgn256_kernel<TX>", {d_planes}, {dst});
    return BBB_OK;
}
int pulse_bits_launch(uint64_t *dst, int64_t, uint64_t nwords, hipStream_t st) {
    if (nwords) op(st, "pulse_bits_kernel", {}, {dst});
    return BBB_OK;
}
int widen_i8_i16_launch(const int8_t *src, int16_t *dst, uint64_t, hipStream_t st) {
    op(st, "widen_i8_i16_kernel", {src}, {dst});
    return BBB_OK;
}
int awgn_generic_fill_launch(int, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2, void *dst, int, uint64_t, unsigned,
                             uint64_t, unsigned, hipStream_t st) {
    op(st, "awgn_generic_kernel", {d_taps, d_row_off, d_planes2}, {d_planes2, dst});
    return BBB_OK;
}
int lutopt_words_launch(int, const uint16_t *d_taps, const uint32_t *d_row_off, uint32_t *d_planes2, uint32_t *dst, uint64_t, unsigned, uint64_t,
                        unsigned, bool, hipStream_t st) {
    op(st, "lutopt_words_kernel", {d_taps, d_row_off, d_planes2}, {d_planes2, dst});
    return BBB_OK;
}
int lutopt_words256_launch(const uint32_t *d_planes, uint32_t *dst, uint64_t, unsigned, uint64_t, unsigned, bool, hipStream_t st) {
    op(st, "lutopt_words256_kernel", {d_planes}, {dst});
    return BBB_OK;
}
int clt_tree_launch(int, const uint64_t *states, uint64_t, int16_t *out, hipStream_t st) {
    op(st, "clt_tree_kernel", {states}, {out});
    return BBB_OK;
}
int awgn_small_fill_launch(int, const uint32_t *d_planes, int8_t *dst, uint64_t, unsigned, uint64_t, unsigned, hipStream_t st) {
    op(st, "awgn_small_kernel", {d_planes}, {dst});
    return BBB_OK;
}
int ber256_launch(uint32_t *d_planes, uint32_t *d_prbs_planes, const TrialDev *t, int, unsigned, unsigned long long *d_counters, hipStream_t st) {
    if (t[0].flags & kTrialSaveState) op(st, "ber256_fused_kernel (continued: leaves its states)", {d_planes, d_prbs_planes, d_counters}, {d_planes, d_prbs_planes, d_counters});
    else op(st, "ber256_fused_kernel", {d_planes, d_prbs_planes, d_counters}, {d_counters});
    return BBB_OK;
}
int tx_waveform_launch(const int16_t *, const uint64_t *d_bits, int64_t, uint64_t, int source, const int8_t *d_noise, int, int, int noise_en,
                       uint64_t, uint64_t, int16_t *d_out, hipStream_t st) {
    (void)source;
    if (noise_en && d_noise) op(st, "tx_waveform_kernel", {d_bits, d_noise}, {d_out});
    else op(st, "tx_waveform_kernel", {d_bits}, {d_out});
    return BBB_OK;
}
int rx_slice_launch(const int16_t *d_samples, uint64_t, uint64_t, uint64_t, int, uint64_t *d_out, hipStream_t st) {
    op(st, "rx_slice_kernel", {d_samples}, {d_out});
    return BBB_OK;
}
int prbs_fill_launch(int, uint64_t, uint64_t, uint64_t nbits, uint64_t *dst, hipStream_t st, int) {
    if (nbits) op(st, "prbs_stream_kernel (fill)", {}, {dst});
    return BBB_OK;
}
int prbs_check_launch(int, uint64_t, uint64_t, uint64_t nbits, const uint64_t *src, uint64_t *nerr_dev, hipStream_t st) {
    if (nbits) op(st, "prbs_check_rev_kernel", {src, nerr_dev}, {nerr_dev});
    return BBB_OK;
}
int prbs_detector_launch(int, const uint8_t *bits, uint64_t, uint64_t, uint8_t *err, uint8_t *reload, hipStream_t st) {
    op(st, "prbs_detector_kernel", {bits}, {err, reload});
    return BBB_OK;
}
int prbs_detector_stream_launch(int, const uint64_t *src, uint64_t, uint64_t *err, uint64_t *reload, bbb_detector_stats *stats, uint64_t, uint64_t,
                                hipStream_t st) {
    op(st, "detector stream (chunk pass ... read-back)", {src}, {err, reload});
    if (stats) *stats = bbb_detector_stats{};
    hipStreamSynchronize(st);
    return BBB_OK;
}
int lutopt_search_launch(int, uint64_t, uint64_t, uint64_t, uint64_t *found, uint16_t *, uint32_t *, bbb_search_stats *stats, hipStream_t st) {
    if (found) *found = ~0ull;
    if (stats) *stats = bbb_search_stats{};
    hipStreamSynchronize(st);
    return BBB_OK;
}

}  // namespace bbb
