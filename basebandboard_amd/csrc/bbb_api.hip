// bbb_api.hip -- the extern "C" boundary of libbbb_hip.so (see include/bbb.h).
// Host logic only: argument checks, GF(2) jump-ahead plans, workspace, kernel launches.
#include "bbb_common.hpp"
#include "awgn_launch.hpp"
#include "gf2.hpp"
#include "rccl_loader.hpp"
#include "sweep_shard.hpp"
#include "sweep_threads.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace bbb {

std::string &last_error() {
    static thread_local std::string s;
    return s;
}

int use_device(int device) {
    // the architecture check is done once per device (hipGetDeviceProperties is slow)
    static std::mutex mu;
    static bool checked[64] = {false};
    bool known = false;
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> g(mu);
        known = checked[device];
    }
    if (!known) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(BBB_ENODEV, "no HIP device visible");
        if (device < 0 || device >= n) return fail(BBB_ENODEV, "device index out of range");
        hipDeviceProp_t p;
        BBB_HIP(hipGetDeviceProperties(&p, device));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            return fail(BBB_ENODEV, std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
        if (device < 64) {
            std::lock_guard<std::mutex> g(mu);
            checked[device] = true;
        }
    }
    BBB_HIP(hipSetDevice(device));
    return BBB_OK;
}

// device copy of the doubling matrices for one segment length L
struct JumpPlan {
    GF2Mat B;                     // the per-generator jump itself (host copy, for the first 16 states)
    GF2Mat Bt;                    // its transpose: y = B x as the XOR of the rows of Bt that x selects (GF2Mat::matvec_t)
    uint32_t *d_cols = nullptr;   // [levels][15][k/4 * 16 * W32] nibble tables of M^(j*16^e)
    uint32_t *d_top = nullptr;    // [kSeedTopTables][k/4 * 16 * W32] tables of M^(d * 16^4), d = 1 .. 31: digits 4 and 5 as ONE level
                                  // (awgn_seed_states_launch; built on first use: ensure_top_tables)
    uint32_t qcol64[32] = {0};    // PRBS plans (n <= 31): column c of B^64 -- the step between two generators of one consumer lane
    int levels = 0;
};

}  // namespace bbb

using namespace bbb;

struct bbb_lutopt {
    int k = 0, W64 = 0, W32 = 0, device = 0;
    bool specialised = false;
    bbb_custom_fill_fn custom_fill = nullptr;   // a kernel built for this very matrix (bbb_lutopt_set_custom_fill)
    bbb_custom_ber_fn custom_ber = nullptr;     // the BER kernels built for it (bbb_lutopt_set_custom_ber)
    int small_fast = 0;          // 16 / 32 / 64 / 128 when (k, taps) is the shipped matrix a generated small kernel exists for
    bool fast512 = false;        // the shipped n512 matrix: packed-state kernel of awgn512.hip (int16 out)
    hipStream_t stream = nullptr;
    std::vector<uint16_t> taps;
    std::vector<uint32_t> row_off;
    uint64_t init[8] = {0};
    std::unique_ptr<GF2Powers> pw;                 // powers of A
    std::map<uint64_t, JumpPlan> plans;            // keyed by L
    std::map<uint64_t, JumpPlan> prbs_plans;       // keyed by (k << 48 | L)
    // workspace
    uint32_t *d_states = nullptr; size_t states_cap = 0;      // [W32][G] word-major
    uint32_t *d_planes = nullptr; size_t planes_cap = 0;      // [2][k][nlanes] (second half: generic kernel)
    // PRBS start states of a BER trial, two pairs taken in turn (ber_run): [G] and [32][nlanes]; pp_read[b]: behind the trial
    // kernel that last read pair b
    uint32_t *d_pstates[2] = {nullptr, nullptr}; size_t pstates_cap[2] = {0, 0};
    uint32_t *d_pplanes[2] = {nullptr, nullptr}; size_t pplanes_cap[2] = {0, 0};
    hipEvent_t pp_read[2] = {nullptr, nullptr};
    bool pp_pending[2] = {false, false};
    int pp_idx = 0;
    uint16_t *d_taps = nullptr;
    uint32_t *d_row_off = nullptr;
    unsigned long long *d_counters = nullptr; size_t counters_cap = 0;
    uint32_t *d_txnoise = nullptr; size_t txnoise_cap = 0;    // TX: int8 noise samples (as words)
    uint32_t *d_txbits = nullptr; size_t txbits_cap = 0;      // TX: packed data bits (as words)
    // fused TX (bbb_tx_fill_i16 with noise): the data bits of call s+1 are generated on the side stream while the sample
    // kernel of call s still reads its own -- two buffers, each with the event of its last reader
    uint32_t *d_fbits[2] = {nullptr, nullptr}; size_t fbits_cap[2] = {0, 0};
    hipEvent_t fbits_read[2] = {nullptr, nullptr}, fbits_ready = nullptr;
    // staged TX: the data bits of a noise kernel's windows, written on the staging slot's arithmetic stream in front of the sample kernel, read by the
    // slot's movers.  TWO buffers per slot, taken in turn (round 5): the bits of the slot's next kernel then do not wait for the movers of its last
    // one -- only the sample kernel does, for the staging slot itself -- and 55 us leave the gap between two noise kernels.  The readers
    // of the buffer taken now are the movers of the slot's kernel BEFORE last, which the last kernel -- queued on this same stream -- waited for.
    uint32_t *d_mbits[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; size_t mbits_cap[2][2] = {{0, 0}, {0, 0}};
    unsigned mbits_turn[2] = {0, 0};
    hipEvent_t ber_join = nullptr;       // ber_run: behind the PRBS seeding on the side stream
    // BER trials (round 5): the generators' start states from awgn_seed_head_launch / _tail_planes_launch (two launches: the first 65536 states packed,
    // u32[8][65536], then the planes [256][nlanes] directly), two buffer pairs taken in turn like the PRBS pairs; bs_read[b]: behind the
    // trial kernel that last read pair b; bs_ready[b]: behind the seeding that filled it (on the caller's stream or an arithmetic one)
    uint32_t *d_bstates[2] = {nullptr, nullptr}; size_t bstates_cap[2] = {0, 0};
    uint32_t *d_bplanes[2] = {nullptr, nullptr}; size_t bplanes_cap[2] = {0, 0};
    hipEvent_t bs_read[2] = {nullptr, nullptr}, bs_ready[2] = {nullptr, nullptr};
    bool bs_pending[2] = {false, false}, bs_valid[2] = {false, false};
    uint64_t bs_first[2] = {0, 0}, bs_L[2] = {0, 0}, bs_G[2] = {0, 0};
    int bs_idx = 0;
    unsigned long long *h_counters = nullptr; size_t h_counters_cap = 0;     // pinned: the read-back of bbb_ber_trials / bbb_ber_sweep_multi
    // d_counters is zeroed BEHIND the read-back of the call that used it (counters_zeroed: the event behind that memset), so that the
    // next call's seeding is the first thing it queues; a call on another stream waits for the event
    hipEvent_t counters_zeroed = nullptr;
    bool counters_clean = false;
    bool fbits_pending[2] = {false, false};
    int fbits_slot = 0;
    // which stream position the planes in d_planes currently describe
    bool planes_valid = false;
    uint64_t planes_first = 0, planes_L = 0, planes_G = 0;
    unsigned max_waves = 1024;
    // prefetched start states for an announced next fill (bbb_awgn_prefetch): seeded on a side stream
    // while the previous sample kernel runs, swapped in by the matching bbb_awgn_fill_i8
    struct Prefetch {
        bool valid = false;
        uint64_t first = 0, L = 0, G = 0;
        unsigned nlanes = 0;
        uint32_t *d_states = nullptr; size_t states_cap = 0;
        uint32_t *d_planes = nullptr; size_t planes_cap = 0;
        hipEvent_t seeded = nullptr;      // recorded on the side stream after seeding
        hipEvent_t last_read = nullptr;   // recorded on the main stream after the last kernel that read these buffers
        bool read_pending = false;
    } pf;
    hipStream_t side = nullptr;
    // The stream the library's plane-touching work of the current call goes to: the caller's stream, or -- for the
    // two-kernel ("staged") form of the sample stream -- an internal one, so that the next fill's arithmetic does
    // not queue behind the caller-visible completion of this one (see staged_fill).
    hipStream_t cs = nullptr;
    bool cs_valid = false;
    hipEvent_t handover = nullptr;
    bool staged_mode = false;             // bbb_lutopt_set_staged
    bool has_stream = false;              // a bbb_awgn_stream is open on this handle
    // internal streams: arithmetic (one per staging slot: consecutive sample kernels go to alternate streams, so that the
    // barrier packets in front of kernel s+1 -- the wait for its start states, the record behind kernel s -- are
    // processed while kernel s still runs instead of between the two: 21-27 us per step in profiles/r03_ramp_clock_per_launch.log)
    // and the piece mover
    hipStream_t xs2[2] = {nullptr, nullptr}, ys = nullptr;
    uint32_t *d_stage[2] = {nullptr, nullptr}; size_t stage_cap[2] = {0, 0};
    hipEvent_t stage_free[2] = {nullptr, nullptr};     // recorded behind the mover that read the buffer (behind ALL its movers: queue_mover_with)
    bool stage_busy[2] = {false, false};
    int stage_slot = 0;
    hipEvent_t ev_user = nullptr;
    int pf_waited_slot = -1;              // staging slot whose mover the pending prefetch's seeding waited for
    uint64_t stage_gen[2] = {0, 0};       // movers queued on the slot so far
    uint64_t pf_waited_gen = 0;           // stage_gen[pf_waited_slot] when that seeding was queued: a later mover voids the skip
    hipEvent_t stage_arith[2] = {nullptr, nullptr};   // recorded behind the sample kernel that filled the slot
    // look-ahead (bbb_lutopt_set_staged(h, m), m >= 2): the sample kernel of a fill also produced the next m - 1 fills'
    // samples, which wait in its staging slot: `left` more fills of n samples, the next one at stream position `first`
    // (kind 0: bbb_awgn_fill_i8, a generator step; kind 1: bbb_tx_fill_i16 with configuration `cfg`, a TX sample index)
    // = generator step `step`, at byte `win_lo` of the kernel's output; (L, G, nlanes) = that kernel's partition
    struct Ahead {
        bool valid = false; int kind = 0; uint64_t first = 0, step = 0, n = 0, win_lo = 0, L = 0, G = 0;
        unsigned nlanes = 0, left = 0; int slot = 0; bbb_tx_cfg cfg{};
        int64_t bits_m0 = 0; uint64_t bits_words64 = 0;      // transmitter: the slot's data-bit buffer starts at bit bits_m0
        unsigned bits_buf = 0;                               // ... and is the slot's buffer of this turn
    } ahead;
    bool last_fill_tx = false;            // the last sample-kernel launch was the transmitter variant (more LDS: see bbb_awgn_prefetch)
    bool last_staged_small = false;       // the last staged sample kernel was the small-footprint placement (two guest waves fit beside it)
    int staged_level = 0;                 // 0 off, 1 staged, m >= 2 staged with m fills per sample kernel
    hipEvent_t cur_last_read = nullptr;   // same, for the buffers currently in d_states / d_planes
    bool cur_read_pending = false;
    // optional per-call device timing of the generator kernels (bbb_lutopt_profile)
    bool profiling = false;
    struct ProfEv { hipEvent_t e0, e1, e2; };
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_mover_pending;      // around every mover, on its stream
    double prof_mover_ms = 0; uint64_t prof_mover_calls = 0;
    std::vector<ProfEv> prof_pending;
    hipEvent_t prof_prev_e2 = nullptr;    // completion of the last sample kernel bbb_lutopt_profile_read has accounted for
    double prof_seed_ms = 0, prof_main_ms = 0;
    uint64_t prof_calls = 0;
};

// (defined among the extern "C" entry points below, next to bbb_awgn_prefetch)
extern "C" {
static int seed_announced(bbb_lutopt *h, uint64_t first_step, uint64_t L, uint64_t G, unsigned nlanes, hipStream_t side, int seed_variant);
}

namespace {

// device-side state width in 32-bit words: the seeding / bit-slicing kernels exist for these widths, a k in
// between is padded with zero words (rows and columns beyond k are empty, so the padding stays zero)
int pad_w32(int k) {
    const int w = (k + 31) / 32;
    for (int c : {1, 2, 4, 6, 8, 12, 16})
        if (w <= c) return c;
    return 16;
}

int grow(uint32_t **p, size_t *cap, size_t need_words) {
    if (*cap >= need_words) return BBB_OK;
    if (*p) BBB_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    BBB_HIP(hipMalloc((void **)p, need_words * sizeof(uint32_t)));
    *cap = need_words;
    return BBB_OK;
}

// Nibble-combination table of one matrix (layout: see seed kernels in awgn_kernels.hip).
void nibble_table(const GF2Mat &M, uint32_t *out) {
    const int k = M.n, W32 = pad_w32(k), nnib = (k + 3) / 4;
    const int C = (W32 % 4 == 0) ? 4 : (W32 % 2 == 0 ? 2 : 1), NC = W32 / C;
    std::vector<uint32_t> cols((size_t)k * W32, 0u);
    for (int r = 0; r < k; r++) {
        const uint64_t *row = M.row(r);
        for (int w = 0; w < M.W; w++) {
            uint64_t bits = row[w];
            while (bits) {
                const int c = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                cols[(size_t)c * W32 + (r >> 5)] |= 1u << (r & 31);
            }
        }
    }
    std::vector<uint32_t> ent(16 * (size_t)W32);
    for (int n = 0; n < nnib; n++) {
        std::fill(ent.begin(), ent.end(), 0u);
        for (int v = 1; v < 16; v++) {
            const int c = 4 * n + __builtin_ctz((unsigned)v);
            for (int z = 0; z < W32; z++)
                ent[(size_t)v * W32 + z] = ent[(size_t)(v & (v - 1)) * W32 + z] ^ (c < k ? cols[(size_t)c * W32 + z] : 0u);
        }
        for (int v = 0; v < 16; v++)
            for (int z = 0; z < W32; z++)
                out[(((size_t)n * NC + z / C) * 16 + v) * C + z % C] = ent[(size_t)v * W32 + z];
    }
}

// Radix-16 jump plan of M: for 1 <= e < levels and j = 1..15 the table of M^(j*16^e), packed
// [e][j-1][table], uploaded to the device (level 0, the first 16 states, is done on the host: first16).
int build_plan(const GF2Mat &M, int levels, JumpPlan *plan) {
    const int k = M.n, W32 = pad_w32(k), nnib = (k + 3) / 4;
    const size_t nt = (size_t)nnib * 16 * W32;
    std::vector<uint32_t> host((size_t)levels * 15 * nt, 0);
    GF2Mat m1 = M;                                   // M^(16^e)
    for (int e = 1; e < levels; e++) {
        for (int q = 0; q < 4; q++) m1 = m1.mul(m1);
        GF2Mat mj = m1;
        for (int j = 1; j <= 15; j++) {
            if (j > 1) mj = mj.mul(m1);
            nibble_table(mj, &host[((size_t)e * 15 + (j - 1)) * nt]);
        }
    }
    BBB_HIP(hipMalloc((void **)&plan->d_cols, host.size() * sizeof(uint32_t)));
    BBB_HIP(hipMemcpy(plan->d_cols, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    plan->levels = levels;
    plan->B = M;
    plan->Bt = M.transpose();
    return BBB_OK;
}

// the first 16 start states B^i s0, packed [16][16] 32-bit words
void first16(const JumpPlan &plan, const uint64_t *s0, uint32_t *out) {
    const int W = plan.B.W;
    uint64_t x[8];
    std::memcpy(x, s0, sizeof(uint64_t) * (size_t)W);
    std::memset(out, 0, sizeof(uint32_t) * 256);
    for (int i = 0; i < 16; i++) {
        if (i) plan.Bt.matvec_t(x, x);
        for (int w = 0; w < 2 * W && w < 16; w++) out[i * 16 + w] = (uint32_t)(x[w >> 1] >> (32 * (w & 1)));
    }
}

constexpr int kPlanLevels = 7;    // radix 16: up to 16^7 = 2^28 generators

// the merged top level of a plan (round 5): tables of B^(d 65536), d = 1 .. 31
int ensure_top_tables(JumpPlan *plan) {
    if (plan->d_top) return BBB_OK;
    const GF2Mat &M = plan->B;
    const int k = M.n, W32 = pad_w32(k), nnib = (k + 3) / 4;
    const size_t nt = (size_t)nnib * 16 * W32;
    std::vector<uint32_t> host((size_t)kSeedTopTables * nt, 0);
    GF2Mat m4 = M;                                   // M^(16^4)
    for (int q = 0; q < 16; q++) m4 = m4.mul(m4);
    GF2Mat md = m4;
    for (int d = 1; d <= kSeedTopTables; d++) {
        if (d > 1) md = md.mul(m4);
        nibble_table(md, &host[(size_t)(d - 1) * nt]);
    }
    BBB_HIP(hipMalloc((void **)&plan->d_top, host.size() * sizeof(uint32_t)));
    BBB_HIP(hipMemcpy(plan->d_top, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return BBB_OK;
}

int get_plan(bbb_lutopt *h, uint64_t L, JumpPlan **out) {
    auto it = h->plans.find(L);
    if (it == h->plans.end()) {
        JumpPlan p;
        int rc = build_plan(h->pw->power(L), kPlanLevels, &p);
        if (rc) return rc;
        it = h->plans.emplace(L, p).first;
    }
    *out = &it->second;
    return BBB_OK;
}

int get_prbs_plan(bbb_lutopt *h, int k, uint64_t L, JumpPlan **out) {
    const uint64_t key = ((uint64_t)k << 56) ^ L;
    auto it = h->prbs_plans.find(key);
    if (it == h->prbs_plans.end()) {
        JumpPlan p;
        GF2Powers pw(prbs_matrix(k, prbs_tap(k)));
        int rc = build_plan(pw.power(L), kPlanLevels, &p);
        if (rc) return rc;
        {
            GF2Mat q = p.B;
            for (int i = 0; i < 6; i++) q = q.mul(q);
            const GF2Mat qt = q.transpose();
            for (int c = 0; c < k && c < 32; c++) p.qcol64[c] = (uint32_t)qt.row(c)[0];
        }
        it = h->prbs_plans.emplace(key, p).first;
    }
    *out = &it->second;
    return BBB_OK;
}

// Partition n stream positions into G segments of L; granule = store alignment of a segment.
void partition(const bbb_lutopt *h, uint64_t n, unsigned granule, uint64_t *L, uint64_t *G, unsigned *nlanes) {
    const uint64_t gmax = (uint64_t)h->max_waves * 2048;
    uint64_t l = (n + gmax - 1) / gmax;
    if (l < 64) l = 64;
    l = (l + granule - 1) / granule * granule;
    const uint64_t g = (n + l - 1) / l;
    const uint64_t waves = (g + 2047) / 2048;
    *L = l;
    *G = g;
    *nlanes = (unsigned)(waves * 64);
}

// The library's internal streams: ONE set per device, shared by every handle on it and kept for the life of the process.
// The part maps HIP streams onto four hardware queues in creation order, and two streams on one queue run each other's kernels
// in order: with streams per handle, the SECOND handle of a process found its arithmetic stream on the queue of the caller's
// stream and lost 12 % (round 4: the transmitter's handle in bench.py next to the noise stream's).  With the caller's stream the
// pool makes four.  Sharing a stream between handles only adds order -- every dependency the scheduler needs is an event --
// and a pooled stream is never destroyed, so nothing can be left holding a dead handle's stream.
struct DevStreams { hipStream_t xs[2] = {nullptr, nullptr}, side = nullptr; };
static std::mutex g_streams_mu;
static std::map<int, DevStreams> g_streams;

int ensure_internal_streams(bbb_lutopt *h) {
    if (!h->xs2[0]) {
        std::lock_guard<std::mutex> g(g_streams_mu);
        DevStreams &d = g_streams[h->device];
        // (stream priorities -- arithmetic high, mover low -- made no measurable difference: profiles/README.md)
        for (hipStream_t *st : {&d.xs[0], &d.xs[1]})
            if (!*st) BBB_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
        h->xs2[0] = d.xs[0]; h->xs2[1] = d.xs[1];
    }
    return BBB_OK;
}

// the side stream (PRBS start states of a BER trial, the transmitter's data bits in the one-kernel form, start states of an
// announced fill on a handle that is not staged)
int ensure_side_stream(bbb_lutopt *h) {
    if (!h->side) {
        std::lock_guard<std::mutex> g(g_streams_mu);
        DevStreams &d = g_streams[h->device];
        if (!d.side) BBB_HIP(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
        h->side = d.side;
    }
    return BBB_OK;
}

// Choose the stream this call's plane-touching work goes to.  When it differs from the previous call's (the staged
// fills use an internal stream; the caller may also have re-bound the handle), the new one first waits for
// everything the library queued on the old one.
int begin_op(bbb_lutopt *h, bool internal, bool independent_of_previous_internal = false) {
    if (internal) {
        const int rc = ensure_internal_streams(h);
        if (rc) return rc;
    }
    // an internal operation goes to the arithmetic stream of the staging slot it is about to take (staged_fill_with flips
    // stage_slot next)
    hipStream_t want = internal ? h->xs2[h->stage_slot ^ 1] : h->stream;
    if (h->cs_valid && h->cs != want) {
        // independent_of_previous_internal: the operation reads start states that were seeded elsewhere (a prefetch: its
        // buffers are guarded by their own events) and writes a staging slot guarded by the slot's events -- nothing ties
        // it to the work on the other arithmetic stream, so it does not wait for it
        const bool from_internal = h->cs == h->xs2[0] || h->cs == h->xs2[1];
        if (!(independent_of_previous_internal && internal && from_internal)) {
            if (!h->handover) BBB_HIP(hipEventCreateWithFlags(&h->handover, hipEventDisableTiming));
            BBB_HIP(hipEventRecord(h->handover, h->cs));
            BBB_HIP(hipStreamWaitEvent(want, h->handover, 0));
        }
    }
    h->cs = want;
    h->cs_valid = true;
    return BBB_OK;
}

// Every kernel that touches d_states / d_planes on the main stream is followed by this: after a prefetch swap
// these buffers become the side stream's, which must not seed into them while such a kernel is still running.
int mark_planes_read(bbb_lutopt *h) {
    if (!h->cur_last_read) BBB_HIP(hipEventCreateWithFlags(&h->cur_last_read, hipEventDisableTiming));
    BBB_HIP(hipEventRecord(h->cur_last_read, h->cs));
    h->cur_read_pending = true;
    return BBB_OK;
}

// make d_planes hold the bit-sliced states A^(first + g*L) init, g < G
int prepare_planes(bbb_lutopt *h, uint64_t first, uint64_t L, uint64_t G, unsigned nlanes) {
    if (h->planes_valid && h->planes_first == first && h->planes_L == L && h->planes_G == G) return BBB_OK;
    JumpPlan *plan;
    int rc = get_plan(h, L, &plan);
    if (rc) return rc;
    // the shipped n256 matrix (or a k = 256 matrix with its own kernels): the two-launch seeding of the BER trials (round 5: head +
    // plane-writing tail, 85 us instead of 160).  This is the seeding that nothing hides -- a fill that was not announced derives its
    // start states in line, in front of its own sample kernel; announced fills keep the level chain, which is built to run as a
    // guest beside the previous sample kernel and its mover (8-16 KiB of LDS per block; the head kernel takes 64)
    // (BBB_EXP_SEED_CHAIN=1, experiments build: the level chain here as well, for the A/B of experiments/r05_unhinted.py)
    const bool two_launch = h->k == 256 && (h->specialised || h->custom_fill) && !h->fast512 && G <= ((uint64_t)kSeedTopTables + 1) * 65536 &&
                            !env_knob("BBB_EXP_SEED_CHAIN", 0);
    const size_t states_words = two_launch && (size_t)G * h->W32 < (size_t)65536 * 8 ? (size_t)65536 * 8 : (size_t)G * h->W32;
    if ((rc = grow(&h->d_states, &h->states_cap, states_words))) return rc;
    if ((rc = grow(&h->d_planes, &h->planes_cap, (size_t)2 * h->k * nlanes))) return rc;
    uint64_t s0[8];
    h->pw->apply(first, h->init, s0);
    uint32_t s16[256];
    first16(*plan, s0, s16);
    h->planes_valid = false;
    if (two_launch) {
        if ((rc = ensure_top_tables(plan))) return rc;
        if ((rc = awgn_seed_head_launch(h->k, plan->d_cols, s16, G, h->d_states, h->cs))) return rc;
        rc = awgn_seed_tail_planes_launch(h->k, plan->d_top, G, h->d_states, nlanes, h->d_planes, h->cs);
    } else {
        rc = awgn_seed_launch(h->k, plan->d_cols, s16, G, h->d_states, G, nlanes, h->d_planes, h->cs);
    }
    if (rc) return rc;
    if ((rc = mark_planes_read(h))) return rc;
    h->planes_valid = true;
    h->planes_first = first; h->planes_L = L; h->planes_G = G;
    return BBB_OK;
}

// the start states of (first_step, L, G): those an announced prefetch seeded on the side stream, or seeded now
int acquire_planes(bbb_lutopt *h, uint64_t first_step, uint64_t L, uint64_t G, unsigned nlanes, bool may_use_prefetch,
                   bool *from_prefetch = nullptr) {
    if (from_prefetch) *from_prefetch = false;
    if (may_use_prefetch && h->pf.valid && h->pf.first == first_step && h->pf.L == L && h->pf.G == G) {
        if (from_prefetch) *from_prefetch = true;
        // the announced fill: its start states were seeded on the side stream -- swap them in
        BBB_HIP(hipStreamWaitEvent(h->cs, h->pf.seeded, 0));
        std::swap(h->d_states, h->pf.d_states); std::swap(h->states_cap, h->pf.states_cap);
        std::swap(h->d_planes, h->pf.d_planes); std::swap(h->planes_cap, h->pf.planes_cap);
        std::swap(h->cur_last_read, h->pf.last_read); std::swap(h->cur_read_pending, h->pf.read_pending);
        h->pf.valid = false;
        h->planes_valid = true;
        h->planes_first = first_step; h->planes_L = L; h->planes_G = G;
        return BBB_OK;
    }
    return prepare_planes(h, first_step, L, G, nlanes);
}

// The two-kernel ("staged") form of a fill (bbb_lutopt_set_staged), PLANES form since round 3:
//   produce   awgn256_planes_kernel: the sample kernel stores the 8 count planes of every step as they are into a staging
//             slot (no LDS, no plane -> byte work on the one wave per SIMD that owns the issue slots);
//   deliver   unplane_kernel: a mover with LDS-DMA loads transposes a WINDOW of the staged stream into bytes at `dst` (or, as
//             the shaping mover, into the transmitter's int16 samples).
// Streams (three in all; the part maps streams onto four hardware queues, and a stream that shares one with another ties the
// kernels of both together):
//   arithmetic  xs2[slot]  start states (bbb_awgn_prefetch) and the sample kernel of the fill that takes staging slot `slot`;
//                          ordered after the previous library work unless the start states were announced, NOT after the
//                          caller's stream
//   mover       the CALLER's stream, which first waits for the slot's sample kernel: the mover is then behind everything the
//               caller had queued before this call (it may still be reading `dst`), whatever the caller queues next sees `dst`
//               complete, as with one kernel, and consecutive movers follow each other on one queue.  (Rounds 2-4 had a
//               fourth stream for it, tied to the caller's by an event each way: two cross-queue round trips, 40 us, between
//               two movers -- 1 % of a 20-step region, whose last two movers run alone: experiments/mover_stream_ab.py.)
// Since the next call's arithmetic does not wait for this call's mover, the mover (latency bound, 64 registers, its
// instructions in the issue slots the sample kernel's wave cannot use) runs beside it.  Two staging slots alternate.

// the mover of staging slot `slot`: on the caller's stream, behind the slot's sample kernel.  `launch_mover(staging buffer,
// stream)` queues the kernel
template <typename LaunchMover>
int queue_mover_with(bbb_lutopt *h, int slot, LaunchMover launch_mover) {
    if (!h->stage_free[slot]) BBB_HIP(hipEventCreateWithFlags(&h->stage_free[slot], hipEventDisableTiming));
    hipStream_t ms = h->stream;
    // (-DBBB_EXPERIMENTS, BBB_EXP_MOVER_OWN_STREAM=1: rounds 2-4's mover on a fourth internal stream tied to the caller's by an
    // event each way, for the A/B of experiments/mover_stream_ab.py)
    const bool own_stream = env_knob("BBB_EXP_MOVER_OWN_STREAM", 0) != 0;
    if (own_stream) {
        if (!h->ys) BBB_HIP(hipStreamCreateWithFlags(&h->ys, hipStreamNonBlocking));
        if (!h->ev_user) BBB_HIP(hipEventCreateWithFlags(&h->ev_user, hipEventDisableTiming));
        ms = h->ys;
        BBB_HIP(hipEventRecord(h->ev_user, h->stream));
        BBB_HIP(hipStreamWaitEvent(ms, h->ev_user, 0));
    }
    BBB_HIP(hipStreamWaitEvent(ms, h->stage_arith[slot], 0));
    // stage_free[slot] stands for ALL movers that read the slot: the caller may have re-bound the handle to another stream since
    // the slot's previous mover, so that one is chained in front of this call's record (same stream: already in its past).
    // (tests/test_sched_model.py takes this line out of a COPY of this file -- mutant "mover_chain" -- and requires the model to find
    // a sample kernel overwriting a slot that a mover on the caller's OTHER stream still reads)
    if (h->stage_busy[slot]) BBB_HIP(hipStreamWaitEvent(ms, h->stage_free[slot], 0));
    hipEvent_t m0 = nullptr, m1 = nullptr;
    if (h->profiling) {
        BBB_HIP(hipEventCreate(&m0)); BBB_HIP(hipEventCreate(&m1));
        BBB_HIP(hipEventRecord(m0, ms));
    }
    int rc = launch_mover((const void *)h->d_stage[slot], ms);
    if (rc) return rc;
    if (h->profiling) {
        BBB_HIP(hipEventRecord(m1, ms));
        h->prof_mover_pending.emplace_back(m0, m1);
    }
    BBB_HIP(hipEventRecord(h->stage_free[slot], ms));
    h->stage_busy[slot] = true;
    h->stage_gen[slot]++;
    if (own_stream) BBB_HIP(hipStreamWaitEvent(h->stream, h->stage_free[slot], 0));
    return BBB_OK;
}

// deliver bytes [win_lo, win_lo + n) of the stream staged in `slot` (partition L, G, nlanes) to dst
int deliver_i8(bbb_lutopt *h, int slot, void *dst, uint64_t win_lo, uint64_t n, uint64_t L, uint64_t G, unsigned nlanes) {
    return queue_mover_with(h, slot, [&](const void *stage, hipStream_t ys) {
        return unplane_launch(stage, dst, win_lo, n, (unsigned)L, G, nlanes, ys);
    });
}

// produce: the sample kernel for the stream positions the planes in h->d_planes describe, L steps per generator, into the
// next staging slot.  planes_seeded_after_mover: the start states came from a prefetch whose seeding had itself waited for
// the mover that last read this slot (bbb_awgn_prefetch on a staged handle), so the arithmetic need not wait for it again.
int produce_planes(bbb_lutopt *h, uint64_t L, unsigned nlanes, bbb_lutopt::ProfEv *ev, bool planes_seeded_after_mover, int *slot_out,
                   bool small_footprint = false) {
    const size_t need_words = (size_t)nlanes * (size_t)L * 8;              // nlanes / 64 waves x L steps x 2 KiB
    const int slot = h->stage_slot ^= 1;
    *slot_out = slot;
    if (h->ahead.valid && h->ahead.slot == slot) h->ahead.valid = false;            // what waited there is overwritten now
    for (hipEvent_t *e : {&h->stage_free[slot], &h->stage_arith[slot]})
        if (!*e) BBB_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (h->stage_cap[slot] < need_words) {
        if (h->stage_busy[slot]) BBB_HIP(hipEventSynchronize(h->stage_free[slot]));     // growing frees the old buffer
        h->stage_busy[slot] = false;
        int rc = grow(&h->d_stage[slot], &h->stage_cap[slot], need_words);
        if (rc) return rc;
    }
    // (the skip holds only while no LATER mover was queued on the slot: a prefetch stays valid across fills that do not
    // match it, and those may have put new movers on this very slot since its seeding waited)
    // (the round-2 advisor's case; tests/test_sched_model.py's mutant "stale_skip" drops the generation check from a copy of this file)
    const bool seeding_saw_last_mover = planes_seeded_after_mover && h->pf_waited_slot == slot && h->pf_waited_gen == h->stage_gen[slot];
    if (planes_seeded_after_mover) h->pf_waited_slot = -1;          // consumed
    if (h->stage_busy[slot] && !seeding_saw_last_mover)
        BBB_HIP(hipStreamWaitEvent(h->cs, h->stage_free[slot], 0));   // its last mover has read it
    if (ev) BBB_HIP(hipEventRecord(ev->e1, h->cs));
    int rc = awgn256_planes_launch(h->d_planes, (void *)h->d_stage[slot], (unsigned)L, nlanes, h->cs, small_footprint);
    h->last_staged_small = small_footprint;
    if (rc) return rc;
    if ((rc = mark_planes_read(h))) return rc;
    if (ev) BBB_HIP(hipEventRecord(ev->e2, h->cs));
    BBB_HIP(hipEventRecord(h->stage_arith[slot], h->cs));
    return BBB_OK;
}

// the packed n512 kernel's partition: 16 generators per lane, 1024 per wave, segments in multiples of 8 samples (16-byte
// stores); false if the segment length does not fit 32 bits
bool partition512(const bbb_lutopt *h, uint64_t nsamples, uint64_t *L, uint64_t *G, unsigned *nlanes) {
    const uint64_t gmax = (uint64_t)h->max_waves * 1024;
    uint64_t l = (nsamples + gmax - 1) / gmax;
    if (l < 64) l = 64;
    l = (l + 7) / 8 * 8;
    if (l > 0xffffff00ull) return false;
    *L = l;
    *G = (nsamples + l - 1) / l;
    *nlanes = (unsigned)((*G + 1023) / 1024 * 64);
    return true;
}

int awgn_fill(bbb_lutopt *h, void *dst, int elem_size, uint64_t nsamples, uint64_t first_step) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (nsamples == 0) return BBB_OK;
    if (!dst || ((uintptr_t)dst & 15)) return fail(BBB_EINVAL, "dst must be a 16-byte aligned device pointer");
    if (first_step + nsamples < first_step) return fail(BBB_EINVAL, "first_step + nsamples overflows");
    if (h->k & (h->k - 1)) return fail(BBB_EUNSUP, "CLTGRNG needs k to be a power of two (rng.py:72-76)");
    if (elem_size == 1 && h->k > 256) return fail(BBB_EUNSUP, "k > 256 needs the int16 output");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot generate samples");
    BBB_HIP(hipSetDevice(h->device));
    uint64_t L, G;
    unsigned nlanes;
    if (h->fast512 && elem_size == 2) {
        if (!partition512(h, nsamples, &L, &G, &nlanes))
            return fail(BBB_EINVAL, "nsamples too large for one call (segment length must fit 32 bits): split it");
        int rc5 = begin_op(h, false);
        if (rc5) return rc5;
        h->planes_valid = false;                 // (another layout than the one prepare_planes caches)
        if (h->pf.valid && h->pf.first == first_step && h->pf.L == L && h->pf.G == G) {
            // announced (bbb_awgn_prefetch): the start states were seeded on the side stream -- swap them in
            BBB_HIP(hipStreamWaitEvent(h->cs, h->pf.seeded, 0));
            std::swap(h->d_states, h->pf.d_states); std::swap(h->states_cap, h->pf.states_cap);
            std::swap(h->d_planes, h->pf.d_planes); std::swap(h->planes_cap, h->pf.planes_cap);
            std::swap(h->cur_last_read, h->pf.last_read); std::swap(h->cur_read_pending, h->pf.read_pending);
            h->pf.valid = false;
        } else {
            JumpPlan *plan;
            if ((rc5 = get_plan(h, L, &plan))) return rc5;
            if ((rc5 = grow(&h->d_states, &h->states_cap, (size_t)G * h->W32))) return rc5;
            if ((rc5 = grow(&h->d_planes, &h->planes_cap, (size_t)2 * h->k * nlanes))) return rc5;
            uint64_t s0[8];
            h->pw->apply(first_step, h->init, s0);
            uint32_t s16[256];
            first16(*plan, s0, s16);
            if ((rc5 = awgn_seed_launch(h->k, plan->d_cols, s16, G, h->d_states, G, nlanes, h->d_planes, h->cs, 1))) return rc5;
        }
        if ((rc5 = awgn512p_fill_launch(h->d_planes, (int16_t *)dst, nsamples, (unsigned)L, G, nlanes, h->cs))) return rc5;
        return mark_planes_read(h);
    }
    partition(h, nsamples, 16, &L, &G, &nlanes);
    if (L > 0xffffff00ull) return fail(BBB_EINVAL, "nsamples too large for one call (segment length must fit 32 bits): split it");
    const bool fast256 = h->specialised && elem_size == 1;
    h->last_fill_tx = false;
    const bool staged = fast256 && h->staged_mode && nsamples >= (1ull << 24);
    // look-ahead: this very range was produced by the previous fill's sample kernel and waits in its staging slot
    if (staged && h->ahead.valid && h->ahead.kind == 0 && h->ahead.first == first_step && h->ahead.n == nsamples) {
        // (no begin_op: a delivery queues nothing on an arithmetic stream -- its mover waits for the slot's sample kernel and the
        // caller's stream by their own events.  Taking the NEXT slot's arithmetic stream here, as rounds 2-3 did, tied that
        // stream to the running sample kernel with a handover event: the next sample kernel, whose announced start states make
        // it independent of the running one, then started an event round trip (26-47 us) after it instead of into its tail)
        if (env_knob("BBB_EXP_DELIVER_HANDOVER", 0)) {
            int rc0 = begin_op(h, true);
            if (rc0) return rc0;
        }
        const bbb_lutopt::Ahead a = h->ahead;
        h->ahead.first += nsamples; h->ahead.step += nsamples;
        h->ahead.win_lo += nsamples;
        h->ahead.valid = --h->ahead.left > 0;
        return deliver_i8(h, a.slot, dst, a.win_lo, nsamples, a.L, a.G, a.nlanes);
    }
    const uint64_t m = (uint64_t)h->staged_level;
    const bool ahead = staged && m >= 2 && (nsamples % 16) == 0 && m * nsamples < (1ull << 40) &&
                       first_step + m * nsamples > first_step;
    const uint64_t ntotal = ahead ? m * nsamples : nsamples;      // what the sample kernel produces
    if (ahead) {
        partition(h, ntotal, 16, &L, &G, &nlanes);
        if (L > 0xffffff00ull) return fail(BBB_EINVAL, "nsamples too large for one call (segment length must fit 32 bits): split it");
    }
    // At one read per sample kernel the noise stream is bound by the kernel's guests (mover, then seeding: together longer than
    // the kernel): it takes the small form of the kernel, beside which they run at the same time, and that form is given the
    // state OF its first sample.  With two and more reads per kernel the stream is bound by the kernel: the other form.
    const bool small_form = staged && (h->staged_level == 1 || env_knob("BBB_EXP_NOISE_SMALL", 0));
    const uint64_t seed_step = first_step + (small_form ? 1 : 0);
    // (a fill that will take the announced start states does not depend on the previous sample kernel: see begin_op)
    const bool takes_prefetch = fast256 && h->pf.valid && h->pf.first == seed_step && h->pf.L == L && h->pf.G == G;
    int rc = begin_op(h, staged, staged && takes_prefetch);
    if (rc) return rc;
    bbb_lutopt::ProfEv ev{};
    if (h->profiling) {
        BBB_HIP(hipEventCreate(&ev.e0)); BBB_HIP(hipEventCreate(&ev.e1)); BBB_HIP(hipEventCreate(&ev.e2));
        BBB_HIP(hipEventRecord(ev.e0, h->cs));
    }
    bool from_pf = false;
    rc = acquire_planes(h, seed_step, L, G, nlanes, fast256, &from_pf);
    if (rc) return rc;
    if (staged) {
        int slot = 0;
        rc = produce_planes(h, L, nlanes, h->profiling ? &ev : nullptr, from_pf, &slot, small_form);
        if (!rc) rc = deliver_i8(h, slot, dst, 0, nsamples, L, G, nlanes);
        if (h->profiling) h->prof_pending.push_back(ev);
        if (!rc && ahead) {
            h->ahead.valid = true; h->ahead.kind = 0;
            h->ahead.first = h->ahead.step = first_step + nsamples; h->ahead.n = nsamples; h->ahead.win_lo = nsamples;
            h->ahead.left = (unsigned)m - 1;
            h->ahead.L = L; h->ahead.G = G; h->ahead.nlanes = nlanes; h->ahead.slot = slot;
        }
        return rc;
    }
    if (fast256) {
        if (h->profiling) BBB_HIP(hipEventRecord(ev.e1, h->cs));
        rc = awgn256_fill_launch(h->d_planes, (int8_t *)dst, nsamples, (unsigned)L, G, nlanes, h->cs);
        if (!rc) rc = mark_planes_read(h);
        if (h->profiling) {
            BBB_HIP(hipEventRecord(ev.e2, h->cs));
            h->prof_pending.push_back(ev);
        }
        return rc;
    }
    if (h->profiling) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); (void)hipEventDestroy(ev.e2); }
    if (h->custom_fill && elem_size == 1) {
        const int e = h->custom_fill(h->d_planes, (int8_t *)dst, nsamples, (uint32_t)L, G, nlanes, (void *)h->cs);
        if (e) return fail(BBB_EHIP, std::string("custom sample kernel: ") + hipGetErrorString((hipError_t)e));
        return mark_planes_read(h);
    }
    if (h->small_fast && elem_size == 1) {
        rc = awgn_small_fill_launch(h->small_fast, h->d_planes, (int8_t *)dst, nsamples, (unsigned)L, G, nlanes, h->cs);
        return rc ? rc : mark_planes_read(h);
    }
    h->planes_valid = false;    // the table-driven kernel advances the planes in place
    rc = awgn_generic_fill_launch(h->k, h->d_taps, h->d_row_off, h->d_planes, dst, elem_size, nsamples, (unsigned)L, G,
                                  nlanes, h->cs);
    return rc ? rc : mark_planes_read(h);
}

// tabulate the channel of tx.py:75-81 / rx.py:29 into threshold lists over T = sample + 128
int wrap12(int v) {
    unsigned u = (unsigned)v & 0xfffu;
    return (u & 0x800u) ? (int)u - 4096 : (int)u;
}
int channel_thresholds(int amp, int noise_var, TrialDev *t) {
    for (int bv = 0; bv < 2; bv++) {
        int n = 0, prev = 0;
        for (int T = 0; T < 256; T++) {
            const int g = T - 128;                                  // int8 CLT sample
            const int noise = wrap12(g * noise_var);                // tx.py:75-77
            const int x = wrap12((bv ? amp : -amp) + noise);        // tx.py:80-81
            const int decided = x >= 0;                             // rx.py:29
            const int err = decided != bv;
            if (err != prev) {
                if (n == 4) return fail(BBB_EUNSUP, "channel needs more than 4 decision thresholds");
                t->thr[bv][n++] = T;
                prev = err;
            }
        }
        t->nthr[bv] = n;
    }
    return BBB_OK;
}

int ber_run(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, unsigned long long *counters_dev) {
    {
        const int rc0 = begin_op(h, false);
        if (rc0) return rc0;
    }
    if (!h->specialised && !h->custom_ber)
        return fail(BBB_EUNSUP, "BER trials need the shipped n256 matrix, or a k = 256 matrix with its own kernels attached "
                                "(bbb_lutopt_attach_custom_library / LUTOPT.specialise)");
    std::vector<TrialDev> td((size_t)ncfg);
    for (int i = 0; i < ncfg; i++) {
        const bbb_trial_cfg &c = cfgs[i];
        const int tap = prbs_tap(c.prbs_k);
        if (!tap) return fail(BBB_EINVAL, "k=" + std::to_string(c.prbs_k) + " invalid for PRBS");
        if (c.prbs_state == 0 || (c.prbs_state >> c.prbs_k)) return fail(BBB_EINVAL, "PRBS state must be in [1, 2^k)");
        if (c.amp < 0 || c.amp > 2047 || c.noise_var < 0 || c.noise_var > 15)
            return fail(BBB_EINVAL, "amp must be 0..2047 and noise_var 0..15");
        if (c.warmup + c.first_bit < c.warmup || c.warmup + c.first_bit + c.nbits < c.nbits)
            return fail(BBB_EINVAL, "warmup + first_bit + nbits overflows");
        td[(size_t)i].prbs_k = c.prbs_k;
        td[(size_t)i].prbs_tap = tap;
        int rc = channel_thresholds(c.amp, c.noise_var, &td[(size_t)i]);
        if (rc) return rc;
    }
    // Consecutive trials that read the SAME noise and PRBS streams (same seeds, offsets, length)
    // differ only in the channel thresholds: they are evaluated together, on one pass over the
    // streams.  The counters are identical to running them one by one.
    for (int i = 0; i < ncfg;) {
        const bbb_trial_cfg &c = cfgs[i];
        // only settings with one decision threshold per bit value can share a launch (the
        // straight-line kernel); 12-bit wrap-around settings run alone on the general kernel
        auto single = [&](int idx) {
            const TrialDev &t = td[(size_t)idx];
            for (int bv = 0; bv < 2; bv++) {
                int real = 0;                       // thresholds strictly inside (0, 256)
                for (int j = 0; j < t.nthr[bv]; j++) real += t.thr[bv][j] > 0 && t.thr[bv][j] < 256;
                if (real != 1) return false;
            }
            return true;
        };
        int n = 1;
        while (single(i) && i + n < ncfg && n < BBB_BER_MAX_GROUP) {
            const bbb_trial_cfg &d = cfgs[i + n];
            if (d.prbs_k != c.prbs_k || d.prbs_state != c.prbs_state || d.warmup != c.warmup ||
                d.first_bit != c.first_bit || d.nbits != c.nbits || !single(i + n))
                break;
            n++;
        }
        if (c.nbits == 0) { i += n; continue; }
        uint64_t L, G;
        unsigned nlanes;
        partition(h, c.nbits, 2, &L, &G, &nlanes);
        // per-lane counters are 32 bits wide and count up to 32 L
        if (L >= (1ull << 27)) return fail(BBB_EINVAL, "nbits too large for one trial (about 2^47): split it with first_bit");
        for (int j = 0; j < n; j++) { td[(size_t)(i + j)].L = (uint32_t)L; td[(size_t)(i + j)].G = G; td[(size_t)(i + j)].nbits = c.nbits; }
        int rc;
        // The BER kernels take the state OF their first sample: one clock past the stream position.  Neither set of start states
        // touches the handle's stream-fill buffers (d_states / d_planes, the prefetch set): trials own two pairs of generator buffers
        // (the first 65536 states packed + the bit planes) and two PRBS pairs, each taken in turn.
        //
        // Round 5 (profiles/r05_base_ber_timeline.txt: an isolated trial's kernel started 155 us after the first launch -- seven
        // launches of generator seeding, 131 us, and the PRBS's states, queued 58 us later and 70-80 us long beside the big levels,
        // finishing last).  Now: everything the host has to compute comes first; then the launches, in the order the GPU needs
        // them -- the generators' head kernel (25 us) with the PRBS seeding as extra blocks of the same launch (15 us of VALU work, one
        // state per consumer lane), the tail kernel, which writes the planes (60 us) -- two launches where there were nine, no
        // bit-slicing pass, no side stream and no event in front of the trial kernel.
        // Where the generators' states are derived depends on what the caller's stream is doing:
        //   busy   (a trial queued behind another: bbb_ber_trials_dev does not synchronise) -- on an arithmetic stream, BESIDE the
        //          kernel of the trial before (the seedings are latency, not work; the trial kernel leaves 140 registers per SIMD),
        //          and this trial's kernel follows that one directly: round 4's 1.29 -> 1.17 ms per sweep in a sequence;
        //   idle   (an isolated call) -- in line on the caller's stream: nothing to overlap with, and a cross-queue event in
        //          front of the trial kernel is a round trip through the command processor (20-26 us: DESIGN.md 3.4).
        // Both orders carry the same dependencies (a buffer's last reader -> its seeding -> the trial kernel), by stream order
        // where the stream is the same and by events where it is not.
        if ((rc = ensure_internal_streams(h))) return rc;
        const uint64_t gen_first = c.warmup + c.first_bit + 1;
        // -- host: plans, buffers, the first sixteen states of both chains
        JumpPlan *pp, *plan = nullptr;
        if ((rc = get_prbs_plan(h, c.prbs_k, L, &pp))) return rc;
        const int pb = h->pp_idx ^= 1;
        for (int b : {pb, pb ^ 1}) {               // (both pairs sized by the first trial of a size: an allocation sits in the call's path)
            if (!h->pp_read[b]) BBB_HIP(hipEventCreateWithFlags(&h->pp_read[b], hipEventDisableTiming));
            if (h->pplanes_cap[b] < (size_t)32 * nlanes) {
                if (h->pp_pending[b]) BBB_HIP(hipEventSynchronize(h->pp_read[b]));          // growing frees the old buffer
                h->pp_pending[b] = false;
                if ((rc = grow(&h->d_pplanes[b], &h->pplanes_cap[b], (size_t)32 * nlanes))) return rc;
            }
        }
        uint32_t ps16[256], s16[256];
        {
            uint64_t ps0 = 0;
            if ((rc = prbs_state_at_host(c.prbs_k, c.prbs_state, c.first_bit, &ps0))) return rc;
            uint64_t ps64[8] = {ps0};
            first16(*pp, ps64, ps16);
        }
        int sb = -1;
        for (int b : {0, 1})
            if (h->bs_valid[b] && h->bs_first[b] == gen_first && h->bs_L[b] == L && h->bs_G[b] == G) sb = b;     // (the same trial again)
        const bool seed_gen = sb < 0;
        if (seed_gen) {
            sb = h->bs_idx ^= 1;
            for (int b : {sb, sb ^ 1}) {
                if (!h->bs_read[b]) BBB_HIP(hipEventCreateWithFlags(&h->bs_read[b], hipEventDisableTiming));
                if (h->bstates_cap[b] < (size_t)65536 * h->W32 || h->bplanes_cap[b] < (size_t)h->k * nlanes) {
                    if (h->bs_pending[b]) BBB_HIP(hipEventSynchronize(h->bs_read[b]));      // growing frees the old buffers
                    h->bs_pending[b] = false;
                    h->bs_valid[b] = false;
                    if ((rc = grow(&h->d_bstates[b], &h->bstates_cap[b], (size_t)65536 * h->W32))) return rc;
                    if ((rc = grow(&h->d_bplanes[b], &h->bplanes_cap[b], (size_t)h->k * nlanes))) return rc;
                }
            }
            for (int b : {0, 1})
                if (!h->bs_ready[b]) BBB_HIP(hipEventCreateWithFlags(&h->bs_ready[b], hipEventDisableTiming));
            if ((rc = get_plan(h, L, &plan))) return rc;
            if ((rc = ensure_top_tables(plan))) return rc;
            uint64_t s0[8] = {0};
            h->pw->apply(gen_first, h->init, s0);
            first16(*plan, s0, s16);
        }
        // -- launches
        const bool busy = seed_gen && hipStreamQuery(h->cs) == hipErrorNotReady;
        hipStream_t ss = busy ? h->xs2[0] : h->cs;
        // (the PRBS pair's last reader, the trial before last: wherever its seeding goes now)
        if (h->pp_pending[pb]) BBB_HIP(hipStreamWaitEvent(seed_gen ? ss : h->cs, h->pp_read[pb], 0));
        if (seed_gen) {
            h->bs_valid[sb] = false;
            if (h->bs_pending[sb]) BBB_HIP(hipStreamWaitEvent(ss, h->bs_read[sb], 0));    // the trial before last read this pair
            // the PRBS start states ride on the head launch (extra blocks: VALU work beside the head's LDS and L2 latency)
            const PrbsSeedRide ride{c.prbs_k, pp->d_cols, ps16, pp->qcol64, nlanes, h->d_pplanes[pb]};
            if ((rc = awgn_seed_head_launch(h->k, plan->d_cols, s16, G, h->d_bstates[sb], ss, &ride))) return rc;
            if ((rc = awgn_seed_tail_planes_launch(h->k, plan->d_top, G, h->d_bstates[sb], nlanes, h->d_bplanes[sb], ss))) return rc;
            BBB_HIP(hipEventRecord(h->bs_ready[sb], ss));
            if (busy) BBB_HIP(hipStreamWaitEvent(h->cs, h->bs_ready[sb], 0));
            h->bs_valid[sb] = true;
            h->bs_first[sb] = gen_first; h->bs_L[sb] = L; h->bs_G[sb] = G;
        } else {
            // the same trial again: its generator states were derived by an earlier call -- on whatever stream that call found right;
            // the PRBS states (their pairs alternate) in line
            BBB_HIP(hipStreamWaitEvent(h->cs, h->bs_ready[sb], 0));
            if ((rc = prbs_seed_lanes_launch(c.prbs_k, pp->d_cols, ps16, pp->qcol64, G, nlanes, h->d_pplanes[pb], h->cs))) return rc;
        }
        if (h->specialised) {
            if ((rc = ber256_launch(h->d_bplanes[sb], h->d_pplanes[pb], &td[(size_t)i], n, nlanes, counters_dev + 2 * (size_t)i, h->cs))) return rc;
        } else {
            const int e = h->custom_ber(h->d_bplanes[sb], h->d_pplanes[pb], &td[(size_t)i], n, nlanes, (uint64_t *)(counters_dev + 2 * (size_t)i), (void *)h->cs);
            if (e) return fail(e < 0 ? e : BBB_EHIP, "custom BER kernel failed");
        }
        BBB_HIP(hipEventRecord(h->pp_read[pb], h->cs));
        h->pp_pending[pb] = true;
        BBB_HIP(hipEventRecord(h->bs_read[sb], h->cs));
        h->bs_pending[sb] = true;
        i += n;
    }
    return BBB_OK;
}

}  // namespace

extern "C" {

int bbb_abi_version(void) { return BBB_ABI_VERSION; }

const char *bbb_strerror(int code) {
    switch (code) {
    case BBB_OK: return "ok";
    case BBB_EINVAL: return "invalid argument";
    case BBB_ENOMEM: return "out of memory";
    case BBB_EHIP: return "HIP runtime error";
    case BBB_EIO: return "matrix file unreadable or malformed";
    case BBB_ENODEV: return "no gfx950 device";
    case BBB_EUNSUP: return "unsupported request";
    default: return "unknown error";
    }
}

const char *bbb_last_error_detail(void) { return last_error().c_str(); }

int bbb_device_count(int *count) {
    if (!count) return fail(BBB_EINVAL, "null count");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return BBB_OK;
}

void bbb_free(void *p) { std::free(p); }

int bbb_lutopt_load_matrix_file(const char *path, int *k, uint16_t **taps, uint32_t **row_off) {
    if (!path || !k || !taps || !row_off) return fail(BBB_EINVAL, "null argument");
    FILE *f = std::fopen(path, "r");
    if (!f) return fail(BBB_EIO, std::string("cannot open ") + path);
    std::vector<std::string> rows;
    char line[BBB_MAX_K + 16];
    while (std::fgets(line, sizeof line, f)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
        if (!s.empty()) rows.push_back(s);
    }
    std::fclose(f);
    const int n = (int)rows.size();
    if (n == 0 || n > BBB_MAX_K) return fail(BBB_EIO, "matrix must have 1..512 rows");
    std::vector<uint16_t> t;
    std::vector<uint32_t> off(1, 0);
    for (int r = 0; r < n; r++) {
        if ((int)rows[r].size() != n) return fail(BBB_EIO, "matrix is not square");
        for (int c = 0; c < n; c++) {
            if (rows[r][c] == '1') t.push_back((uint16_t)c);
            else if (rows[r][c] != '0') return fail(BBB_EIO, "matrix holds a character other than 0/1");
        }
        off.push_back((uint32_t)t.size());
    }
    *taps = (uint16_t *)std::malloc(sizeof(uint16_t) * (t.empty() ? 1 : t.size()));
    *row_off = (uint32_t *)std::malloc(sizeof(uint32_t) * off.size());
    if (!*taps || !*row_off) return fail(BBB_ENOMEM, "malloc");
    std::memcpy(*taps, t.data(), sizeof(uint16_t) * t.size());
    std::memcpy(*row_off, off.data(), sizeof(uint32_t) * off.size());
    *k = n;
    return BBB_OK;
}

int bbb_lutopt_create(bbb_lutopt **out, int k, const uint16_t *taps, const uint32_t *row_off,
                      const uint64_t *init_words, int device) {
    if (!out || !taps || !row_off || !init_words) return fail(BBB_EINVAL, "null argument");
    // LUTOPT itself puts no constraint on k (rng.py:21-40; rng_recurrences.py:105 ships n192, rnghunt searches
    // n = 192, rnghunt.rs:14); only CLTGRNG needs a power of two (rng.py:72-76) -- checked where samples are asked for
    if (k < 2 || k > BBB_MAX_K) return fail(BBB_EINVAL, "k must be in [2, 512]");
    GF2Mat A(k);
    for (int r = 0; r < k; r++) {
        const uint32_t n = row_off[r + 1] - row_off[r];
        if (row_off[r + 1] < row_off[r] || n < 1 || n > 8) return fail(BBB_EINVAL, "every row needs 1..8 taps");
        for (uint32_t j = row_off[r]; j < row_off[r + 1]; j++) {
            if (taps[j] >= k) return fail(BBB_EINVAL, "tap index out of range");
            if (A.get(r, taps[j])) return fail(BBB_EINVAL, "duplicate tap in a row");
            A.set(r, taps[j]);
        }
    }
    // device == -1: a host-only handle for the jump-ahead algebra (bbb_lutopt_state_at);
    // every compute entry point on it fails with BBB_ENODEV -- there is no CPU compute path.
    if (device != -1) {
        int rc = use_device(device);
        if (rc) return rc;
    }
    std::unique_ptr<bbb_lutopt> h(new bbb_lutopt);
    h->k = k;
    h->W64 = (k + 63) / 64;
    h->W32 = pad_w32(k);
    h->device = device;
    h->taps.assign(taps, taps + row_off[k]);
    h->row_off.assign(row_off, row_off + k + 1);
    bool any = false;
    for (int w = 0; w < h->W64; w++) {
        uint64_t v = init_words[w];
        if (w == h->W64 - 1 && (k & 63)) v &= (1ull << (k & 63)) - 1ull;
        h->init[w] = v;
        any |= v != 0;
    }
    (void)any;   // an all-zero state is legal in the HDL too (it just stays zero)
    h->pw.reset(new GF2Powers(A));
    h->specialised = awgn256_matches(k, taps, row_off);
    h->small_fast = awgn_small_matches(k, taps, row_off);
    h->fast512 = awgn512p_matches(k, taps, row_off);
    if (device == -1) {
        *out = h.release();
        return BBB_OK;
    }
    hipDeviceProp_t p;
    BBB_HIP(hipGetDeviceProperties(&p, device));
    h->max_waves = (unsigned)p.multiProcessorCount * 4;    // one wave per SIMD
    BBB_HIP(hipMalloc((void **)&h->d_taps, sizeof(uint16_t) * h->taps.size()));
    BBB_HIP(hipMalloc((void **)&h->d_row_off, sizeof(uint32_t) * h->row_off.size()));
    BBB_HIP(hipMemcpy(h->d_taps, h->taps.data(), sizeof(uint16_t) * h->taps.size(), hipMemcpyHostToDevice));
    BBB_HIP(hipMemcpy(h->d_row_off, h->row_off.data(), sizeof(uint32_t) * h->row_off.size(), hipMemcpyHostToDevice));
    *out = h.release();
    return BBB_OK;
}

int bbb_lutopt_destroy(bbb_lutopt *h) {
    if (!h) return BBB_OK;
    if (h->device < 0) { delete h; return BBB_OK; }
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto &p : h->plans) { (void)hipFree(p.second.d_cols); if (p.second.d_top) (void)hipFree(p.second.d_top); }
    for (auto &p : h->prbs_plans) (void)hipFree(p.second.d_cols);
    for (void *p : {(void *)h->d_states, (void *)h->d_planes, (void *)h->d_pstates[0], (void *)h->d_pplanes[0], (void *)h->d_pstates[1], (void *)h->d_pplanes[1],
                    (void *)h->d_taps, (void *)h->d_row_off, (void *)h->d_counters, (void *)h->d_txnoise,
                    (void *)h->d_txbits, (void *)h->d_fbits[0], (void *)h->d_fbits[1], (void *)h->d_mbits[0][0], (void *)h->d_mbits[0][1], (void *)h->d_mbits[1][0], (void *)h->d_mbits[1][1], (void *)h->pf.d_states, (void *)h->pf.d_planes,
                    (void *)h->d_bstates[0], (void *)h->d_bstates[1], (void *)h->d_bplanes[0], (void *)h->d_bplanes[1]})
        (void)hipFree(p);
    if (h->h_counters) (void)hipHostFree(h->h_counters);
    for (hipEvent_t e : {h->pf.seeded, h->pf.last_read, h->cur_last_read, h->handover, h->stage_free[0], h->stage_free[1],
                         h->stage_arith[0], h->stage_arith[1], h->ev_user, h->fbits_read[0], h->fbits_read[1], h->fbits_ready, h->ber_join, h->pp_read[0], h->pp_read[1],
                         h->bs_read[0], h->bs_read[1], h->bs_ready[0], h->bs_ready[1], h->counters_zeroed})
        if (e) (void)hipEventDestroy(e);
    // (profiling events of calls whose times were never read: found by the scheduler model's leak check, tests/sched_model)
    for (auto &pr : h->prof_mover_pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &ev : h->prof_pending) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); (void)hipEventDestroy(ev.e2); }
    if (h->prof_prev_e2) (void)hipEventDestroy(h->prof_prev_e2);
    if (h->ys) (void)hipStreamDestroy(h->ys);        // (the other internal streams are the device's pool: ensure_internal_streams)
    (void)hipFree(h->d_stage[0]);
    (void)hipFree(h->d_stage[1]);
    delete h;
    return BBB_OK;
}

int bbb_lutopt_set_stream(bbb_lutopt *h, void *hip_stream) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    h->stream = (hipStream_t)hip_stream;
    return BBB_OK;
}

int bbb_lutopt_set_custom_fill(bbb_lutopt *h, bbb_custom_fill_fn fn) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (fn && (h->k > 256 || (h->k & (h->k - 1)))) return fail(BBB_EUNSUP, "custom kernels exist for power-of-two k <= 256");
    h->custom_fill = fn;
    h->planes_valid = false;
    return BBB_OK;
}

int bbb_lutopt_set_custom_ber(bbb_lutopt *h, bbb_custom_ber_fn fn) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (fn && h->k != 256) return fail(BBB_EUNSUP, "the fused BER kernels exist for k = 256");
    h->custom_ber = fn;
    return BBB_OK;
}

int bbb_lutopt_attach_custom_library(bbb_lutopt *h, const char *path) {
    if (!h || !path) return fail(BBB_EINVAL, "null argument");
    void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(BBB_EIO, std::string("cannot load ") + path + ": " + dlerror());
    auto bail = [&](int code, const std::string &what) { dlclose(lib); return fail(code, what); };
    auto order = reinterpret_cast<int (*)(void)>(dlsym(lib, "bbb_custom_order"));
    auto abi = reinterpret_cast<int (*)(void)>(dlsym(lib, "bbb_custom_abi"));
    auto fill = reinterpret_cast<bbb_custom_fill_fn>(dlsym(lib, "bbb_custom_fill"));
    if (!order || !fill) return bail(BBB_EIO, std::string(path) + " does not export bbb_custom_order / bbb_custom_fill");
    // a library built by another checkout shares plane layout and trial records with THAT library, not with this one
    if (!abi || abi() != BBB_CUSTOM_ABI)
        return bail(BBB_EINVAL, std::string(path) + " was built for another version of this library (custom ABI " +
                                    (abi ? std::to_string(abi()) : std::string("none")) + ", expected " + std::to_string(BBB_CUSTOM_ABI) + "): rebuild it");
    if (order() != h->k) return bail(BBB_EINVAL, std::string(path) + " was built for another order");
    int rc = bbb_lutopt_set_custom_fill(h, fill);
    if (rc) { dlclose(lib); return rc; }
    auto ber = reinterpret_cast<bbb_custom_ber_fn>(dlsym(lib, "bbb_custom_ber"));
    if (ber && h->k == 256) rc = bbb_lutopt_set_custom_ber(h, ber);
    return rc;            // (on success the library stays loaded for the life of the process: the handle calls into it)
}

int bbb_lutopt_set_staged(bbb_lutopt *h, int enable) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (enable < 0 || enable > 8) return fail(BBB_EINVAL, "0 = off, 1 = staged, 2..8 = staged with that many fills per sample kernel");
    h->staged_mode = enable != 0;
    h->staged_level = enable;
    h->ahead.valid = false;               // (also the way to drop a waiting look-ahead half)
    return BBB_OK;
}

int bbb_lutopt_is_specialised(const bbb_lutopt *h) { return h && h->specialised ? 1 : 0; }

int bbb_lutopt_profile(bbb_lutopt *h, int enable) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    h->profiling = enable != 0;
    return BBB_OK;
}

int bbb_lutopt_profile_read(bbb_lutopt *h, double *seed_ms, double *kernel_ms, uint64_t *calls, int reset) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    for (auto &ev : h->prof_pending) {
        BBB_HIP(hipEventSynchronize(ev.e2));
        float a = 0, b = 0;
        BBB_HIP(hipEventElapsedTime(&a, ev.e0, ev.e1));
        BBB_HIP(hipEventElapsedTime(&b, ev.e1, ev.e2));
        // A sample kernel whose start states were announced is queued while the previous one still holds every SIMD (one
        // 512-register wave each): between e1 and e2 it first WAITS for those waves to retire.  Its time on the machine is
        // counted from the later of its own dispatch and the previous sample kernel's completion.
        if (h->prof_prev_e2) {
            float c = 0;
            BBB_HIP(hipEventElapsedTime(&c, h->prof_prev_e2, ev.e2));
            if (c > 0 && c < b) b = c;
            (void)hipEventDestroy(h->prof_prev_e2);
        }
        h->prof_prev_e2 = ev.e2;
        h->prof_seed_ms += a; h->prof_main_ms += b; h->prof_calls++;
        (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1);
    }
    h->prof_pending.clear();
    if (seed_ms) *seed_ms = h->prof_seed_ms;
    if (kernel_ms) *kernel_ms = h->prof_main_ms;
    if (calls) *calls = h->prof_calls;
    if (reset) { h->prof_seed_ms = h->prof_main_ms = 0; h->prof_calls = 0; }
    return BBB_OK;
}

int bbb_lutopt_profile_read_mover(bbb_lutopt *h, double *mover_ms, uint64_t *calls, int reset) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    for (auto &ev : h->prof_mover_pending) {
        BBB_HIP(hipEventSynchronize(ev.second));
        float a = 0;
        BBB_HIP(hipEventElapsedTime(&a, ev.first, ev.second));
        h->prof_mover_ms += a; h->prof_mover_calls++;
        (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
    }
    h->prof_mover_pending.clear();
    if (mover_ms) *mover_ms = h->prof_mover_ms;
    if (calls) *calls = h->prof_mover_calls;
    if (reset) { h->prof_mover_ms = 0; h->prof_mover_calls = 0; }
    return BBB_OK;
}

int bbb_lutopt_state_at(bbb_lutopt *h, uint64_t nsteps, uint64_t *state_words) {
    if (!h || !state_words) return fail(BBB_EINVAL, "null argument");
    h->pw->apply(nsteps, h->init, state_words);       // ceil(k/64) words
    return BBB_OK;
}

// Start states of (first_step, L, G, nlanes) into the handle's SECOND set of start-state buffers (h->pf), on `side`, beside
// whatever reads the first set: an announced fill's (bbb_awgn_prefetch) or, just in time, a BER trial's (ber_run).
// acquire_planes swaps the sets.
static int seed_announced(bbb_lutopt *h, uint64_t first_step, uint64_t L, uint64_t G, unsigned nlanes, hipStream_t side, int seed_variant) {
    JumpPlan *plan;
    int rc = get_plan(h, L, &plan);
    if (rc) return rc;
    bbb_lutopt::Prefetch &pf = h->pf;
    pf.valid = false;
    // An announcement that was never taken leaves its seeding behind -- queued on the arithmetic stream of the fill it expected,
    // which need not be the one this seeding goes to: without this wait the two would write the same buffers side by side
    // (the soak test's case: a hint whose fill came with another partition, then the next hint).
    // (tests/test_sched_model.py's mutant "untaken_hint" is a copy of this file as it stood BEFORE this wait existed: the model of
    // tests/sched_model/ must find that race)
    if (pf.seeded) BBB_HIP(hipStreamWaitEvent(side, pf.seeded, 0));
    else BBB_HIP(hipEventCreateWithFlags(&pf.seeded, hipEventDisableTiming));
    // the buffers may still be read by the sample kernel that used them last (main stream)
    if (pf.read_pending) BBB_HIP(hipStreamWaitEvent(side, pf.last_read, 0));
    if (pf.states_cap < (size_t)G * h->W32 || pf.planes_cap < (size_t)2 * h->k * nlanes) {
        if (pf.read_pending) BBB_HIP(hipEventSynchronize(pf.last_read));     // growing frees the old buffers
        if ((rc = grow(&pf.d_states, &pf.states_cap, (size_t)G * h->W32))) return rc;
        if ((rc = grow(&pf.d_planes, &pf.planes_cap, (size_t)2 * h->k * nlanes))) return rc;
    }
    pf.read_pending = false;
    uint64_t s0[8] = {0};
    h->pw->apply(first_step, h->init, s0);
    uint32_t s16[256];
    first16(*plan, s0, s16);
    if ((rc = awgn_seed_launch(h->k, plan->d_cols, s16, G, pf.d_states, G, nlanes, pf.d_planes, side, h->fast512 ? 1 : 0, seed_variant))) return rc;
    BBB_HIP(hipEventRecord(pf.seeded, side));
    pf.valid = true;
    pf.first = first_step; pf.L = L; pf.G = G; pf.nlanes = nlanes;
    return BBB_OK;
}

// for_tx: the announced fill will run the SMALL form of the sample kernel (the transmitter's, and the noise stream's at one read
// per kernel), which takes the state OF its first sample
static int awgn_prefetch(bbb_lutopt *h, uint64_t nsamples, uint64_t first_step, bool for_tx);

int bbb_awgn_prefetch(bbb_lutopt *h, uint64_t nsamples, uint64_t first_step) {
    // (which kind of fill comes next is not known here: the kind of the last staged one is assumed; a wrong guess only means
    // that the fill seeds for itself)
    return awgn_prefetch(h, nsamples, first_step, h && (h->staged_level == 1 || h->last_staged_small || env_knob("BBB_EXP_NOISE_SMALL", 0)));
}

static int awgn_prefetch(bbb_lutopt *h, uint64_t nsamples, uint64_t first_step, bool for_tx) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1)");
    if ((!h->specialised && !h->fast512) || nsamples == 0) return BBB_OK;          // a hint: nothing to do for the table-driven path
    BBB_HIP(hipSetDevice(h->device));
    uint64_t L, G;
    unsigned nlanes;
    const bool staged_size = nsamples >= (1ull << 24);            // (of the announced fill itself, before look-ahead widens it)
    if (h->staged_level >= 2 && nsamples >= (1ull << 24) && (nsamples % 16) == 0) {
        // look-ahead: a sample kernel covers m fills.  If the announced fill is one that already waits in a staging slot,
        // what needs start states is the fill after the LAST waiting one; and they are those of an m-fold request
        // (which kind of fill comes next is not known here: the kind of the last one is assumed; a wrong guess only
        // means that the fill seeds for itself)
        const uint64_t m = (uint64_t)h->staged_level;
        if (h->ahead.valid && h->ahead.step == first_step && h->ahead.n == nsamples) {
            const uint64_t skip = (uint64_t)h->ahead.left * nsamples;
            if (first_step + skip < first_step) return BBB_OK;
            first_step += skip;
        }
        if (m * nsamples < (1ull << 40) && first_step + m * nsamples > first_step) nsamples *= m;
    }
    // the transmitter's noise kernel on a staged handle is given the state OF its first sample (bbb_tx_fill_i16)
    if (for_tx && h->staged_mode && h->specialised && staged_size) {
        if (first_step + 1 == 0) return BBB_OK;
        first_step += 1;
    }
    if (h->fast512) {
        if (!partition512(h, nsamples, &L, &G, &nlanes)) return BBB_OK;
    } else {
        partition(h, nsamples, 16, &L, &G, &nlanes);
    }
    if (L > 0xffffff00ull) return BBB_OK;                         // the matching fill will refuse; nothing to prepare
    if (h->pf.valid && h->pf.first == first_step && h->pf.L == L && h->pf.G == G) return BBB_OK;      // already under way
    // Which stream seeds.  A staged k = 256 handle: the ARITHMETIC stream of the staging slot the announced fill will take --
    // the sample kernel then follows its start states on one stream, with no event between them.  (With the seeding on a
    // stream of its own beside two arithmetic streams the mover and the seeding ran BETWEEN the sample kernels instead of
    // beside them: a stream more than the part has hardware queues for.)
    hipStream_t side = nullptr;
    int rc;
    if (h->staged_mode && h->specialised) {
        if ((rc = ensure_internal_streams(h))) return rc;
        side = h->xs2[h->stage_slot ^ 1];
    } else {
        if ((rc = ensure_side_stream(h))) return rc;
        side = h->side;
    }
    // staged handles: the running sample kernel (fill s) shares its CUs with ONE guest kernel at a time: a SIMD's registers
    // hold the sample kernel's wave and one guest wave.  The guests of fill s are the piece mover of fill s-1 and this seeding
    // (for fill s+1): the seeding waits for that mover.  (The mover of fill s itself starts when fill s has finished.)
    h->pf_waited_slot = -1;
    h->pf.valid = false;
    // (Beside the small form of the sample kernel -- the transmitter's -- there is room for both guests at once.)
    if (h->staged_mode && h->stage_busy[h->stage_slot ^ 1] && !h->last_staged_small) {
        BBB_HIP(hipStreamWaitEvent(side, h->stage_free[h->stage_slot ^ 1], 0));
        h->pf_waited_slot = h->stage_slot ^ 1;
        h->pf_waited_gen = h->stage_gen[h->stage_slot ^ 1];
    }
    // (this seeding runs beside the kernel of the fill before: the transmitter variant leaves LDS for 8 KiB pieces only)
    return seed_announced(h, first_step, L, G, nlanes, side, h->last_fill_tx ? 4 : 2);
}

int bbb_awgn_fill_i8(bbb_lutopt *h, int8_t *dst_dev, uint64_t nsamples, uint64_t first_step) {
    return awgn_fill(h, dst_dev, 1, nsamples, first_step);
}

int bbb_awgn_fill_i16(bbb_lutopt *h, int16_t *dst_dev, uint64_t nsamples, uint64_t first_step) {
    if (h && h->specialised && h->device >= 0 && nsamples && dst_dev && !((uintptr_t)dst_dev & 15) && (nsamples % 16) == 0) {
        // the generated k = 256 kernel writes int8: fill a scratch buffer with it, then sign-extend
        BBB_HIP(hipSetDevice(h->device));
        int rc = grow(&h->d_txnoise, &h->txnoise_cap, (size_t)(nsamples + 15) / 4 + 4);
        if (rc) return rc;
        if ((rc = awgn_fill(h, h->d_txnoise, 1, nsamples, first_step))) return rc;
        return widen_i8_i16_launch((const int8_t *)h->d_txnoise, dst_dev, nsamples, h->stream);
    }
    return awgn_fill(h, dst_dev, 2, nsamples, first_step);
}

/* ---- the sample stream as an object (include/bbb.h) ------------------------------------------------------------ */

struct bbb_awgn_stream {
    bbb_lutopt *h = nullptr;
    uint64_t n = 0, pos = 0;
    int elem = 1;
    int saved_level = 0;          // the handle's bbb_lutopt_set_staged level before the stream was opened
};

int bbb_awgn_stream_open(bbb_lutopt *h, uint64_t nsamples_per_call, uint64_t first_step, int elem_bytes, bbb_awgn_stream **out) {
    if (!h || !out) return fail(BBB_EINVAL, "null argument");
    if (nsamples_per_call == 0) return fail(BBB_EINVAL, "nsamples_per_call must be positive");
    if (elem_bytes != 1 && elem_bytes != 2) return fail(BBB_EINVAL, "elem_bytes must be 1 (int8) or 2 (int16)");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot generate samples");
    if (h->k & (h->k - 1)) return fail(BBB_EUNSUP, "CLTGRNG needs k to be a power of two (rng.py:72-76)");
    if (elem_bytes == 1 && h->k > 256) return fail(BBB_EUNSUP, "k > 256 needs the int16 output");
    if (h->has_stream) return fail(BBB_EINVAL, "the handle already has an open stream");
    if (first_step + nsamples_per_call < first_step) return fail(BBB_EINVAL, "first_step + nsamples_per_call overflows");
    std::unique_ptr<bbb_awgn_stream> s(new bbb_awgn_stream);
    s->h = h; s->n = nsamples_per_call; s->pos = first_step; s->elem = elem_bytes;
    s->saved_level = h->staged_level;
    if (h->staged_level == 0) {            // (a caller who chose a level keeps it)
        // two reads per sample kernel (look-ahead 2): a sequential reader is exactly the consumer that level is for, and it
        // halves the start-state derivations -- beside the sample kernel the guests (mover, then seeding) otherwise take
        // about as long as the kernel itself (same box, 1e9 per read: 1.16 ms per read at level 1, 1.055 at 2, 1.09 at 4)
        int rc = bbb_lutopt_set_staged(h, 2);
        if (rc) return rc;
    }
    int rc = bbb_awgn_prefetch(h, s->n, s->pos);
    if (rc) return rc;
    h->has_stream = true;
    *out = s.release();
    return BBB_OK;
}

int bbb_awgn_stream_read(bbb_awgn_stream *s, void *dst_dev, uint64_t nsamples) {
    if (!s || !s->h) return fail(BBB_EINVAL, "null stream");
    if (s->pos + nsamples < s->pos) return fail(BBB_EINVAL, "the stream position overflows");
    bbb_lutopt *h = s->h;
    int rc = s->elem == 1 ? bbb_awgn_fill_i8(h, (int8_t *)dst_dev, nsamples, s->pos) : bbb_awgn_fill_i16(h, (int16_t *)dst_dev, nsamples, s->pos);
    if (rc) return rc;
    s->pos += nsamples;
    // the next read is expected to be a whole one: its start states are derived now, beside this read's kernels
    if (s->pos + s->n >= s->pos) rc = bbb_awgn_prefetch(h, s->n, s->pos);
    return rc;
}

int bbb_awgn_stream_next(bbb_awgn_stream *s, void *dst_dev) {
    if (!s) return fail(BBB_EINVAL, "null stream");
    return bbb_awgn_stream_read(s, dst_dev, s->n);
}

int bbb_awgn_stream_seek(bbb_awgn_stream *s, uint64_t first_step) {
    if (!s || !s->h) return fail(BBB_EINVAL, "null stream");
    if (first_step + s->n < first_step) return fail(BBB_EINVAL, "first_step + nsamples_per_call overflows");
    s->pos = first_step;
    s->h->ahead.valid = false;            // what a sample kernel produced ahead at the old position is dropped
    return bbb_awgn_prefetch(s->h, s->n, s->pos);
}

int bbb_awgn_stream_tell(const bbb_awgn_stream *s, uint64_t *next_step) {
    if (!s || !next_step) return fail(BBB_EINVAL, "null argument");
    *next_step = s->pos;
    return BBB_OK;
}

int bbb_awgn_stream_close(bbb_awgn_stream *s) {
    if (!s) return BBB_OK;
    int rc = BBB_OK;
    if (s->h) {
        s->h->has_stream = false;
        if (s->h->staged_level != s->saved_level) rc = bbb_lutopt_set_staged(s->h, s->saved_level);
    }
    delete s;
    return rc;
}

int bbb_lutopt_fill_words(bbb_lutopt *h, uint32_t *dst_dev, uint64_t nstates, uint64_t first_step, int msb_first) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    if (h->k % 32) return fail(BBB_EUNSUP, "the word stream is defined for k a multiple of 32 (util/verify.py:41,50)");
    if (nstates == 0) return BBB_OK;
    if (!dst_dev || ((uintptr_t)dst_dev & 3)) return fail(BBB_EINVAL, "dst must be a 4-byte aligned device pointer");
    if (first_step + nstates < first_step) return fail(BBB_EINVAL, "first_step + nstates overflows");
    if (nstates > (1ull << 48)) return fail(BBB_EINVAL, "nstates must be <= 2^48 per call");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot generate words");
    BBB_HIP(hipSetDevice(h->device));
    uint64_t L, G;
    unsigned nlanes;
    const bool fast = h->specialised && !((uintptr_t)dst_dev & 7);
    partition(h, nstates, fast ? 2 : 1, &L, &G, &nlanes);
    int rc = begin_op(h, false);
    if (rc) return rc;
    rc = prepare_planes(h, first_step, L, G, nlanes);
    if (rc) return rc;
    if (fast) {        // the shipped n256 matrix: generated network, planes stay valid (not advanced in place)
        rc = lutopt_words256_launch(h->d_planes, dst_dev, nstates, (unsigned)L, G, nlanes, msb_first != 0, h->cs);
        return rc ? rc : mark_planes_read(h);
    }
    h->planes_valid = false;    // the table-driven kernel advances the planes in place
    rc = lutopt_words_launch(h->k, h->d_taps, h->d_row_off, h->d_planes, dst_dev, nstates, (unsigned)L, G, nlanes,
                             msb_first != 0, h->cs);
    if (!rc) rc = mark_planes_read(h);
    return rc;
}

int bbb_clt_tree_i16(int k, const uint64_t *states_dev, uint64_t nstates, int16_t *out_dev, int device,
                     void *hip_stream) {
    if (k < 2 || k > BBB_MAX_K || (k & (k - 1)) || (k & 63)) return fail(BBB_EINVAL, "k must be 64, 128, 256 or 512");
    if (nstates && (!states_dev || !out_dev)) return fail(BBB_EINVAL, "null device pointer");
    int rc = use_device(device);
    if (rc) return rc;
    return clt_tree_launch(k, states_dev, nstates, out_dev, (hipStream_t)hip_stream);
}

int bbb_prbs_fill(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits, uint64_t *dst_packed_dev,
                  int device, void *hip_stream) {
    if (nbits && !dst_packed_dev) return fail(BBB_EINVAL, "null device pointer");
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    int rc = use_device(device);
    if (rc) return rc;
    return prbs_fill_launch(k, init_state, first_bit, nbits, dst_packed_dev, (hipStream_t)hip_stream);
}

int bbb_prbs_fill_hint(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits, uint64_t *dst_packed_dev,
                       unsigned flags, int device, void *hip_stream) {
    if (flags & ~(unsigned)BBB_PRBS_WILL_READ_BACK) return fail(BBB_EINVAL, "unknown flag");
    if (nbits && !dst_packed_dev) return fail(BBB_EINVAL, "null device pointer");
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    int rc = use_device(device);
    if (rc) return rc;
    return prbs_fill_launch(k, init_state, first_bit, nbits, dst_packed_dev, (hipStream_t)hip_stream, (flags & BBB_PRBS_WILL_READ_BACK) ? 1 : 0);
}

int bbb_prbs_check_dev(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits,
                       const uint64_t *src_packed_dev, uint64_t *nerr_dev, int device, void *hip_stream) {
    if ((nbits && !src_packed_dev) || !nerr_dev) return fail(BBB_EINVAL, "null device pointer");
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    int rc = use_device(device);
    if (rc) return rc;
    return prbs_check_launch(k, init_state, first_bit, nbits, src_packed_dev, nerr_dev, (hipStream_t)hip_stream);
}

int bbb_prbs_check(int k, uint64_t init_state, uint64_t first_bit, uint64_t nbits, const uint64_t *src_packed_dev,
                   uint64_t *nerr, int device, void *hip_stream) {
    if (!nerr) return fail(BBB_EINVAL, "null nerr");
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    int rc = use_device(device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    // the counter belongs to this call (stream-ordered allocation from the device's pool: microseconds, and
    // two threads or streams checking on one device never share it)
    uint64_t *d = nullptr;
    BBB_HIP(hipMallocAsync((void **)&d, sizeof(uint64_t), st));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(uint64_t), st);
    rc = e == hipSuccess ? bbb_prbs_check_dev(k, init_state, first_bit, nbits, src_packed_dev, d, device, hip_stream)
                         : fail(BBB_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    if (rc == BBB_OK) {
        e = hipMemcpyAsync(nerr, d, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) rc = fail(BBB_EHIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    (void)hipFreeAsync(d, st);
    if (rc) return rc;
    BBB_HIP(hipStreamSynchronize(st));
    return BBB_OK;
}

int bbb_prbs_state_at(int k, uint64_t init_state, uint64_t nbits, uint64_t *state) {
    if (!state) return fail(BBB_EINVAL, "null state");
    if (init_state >> (k > 0 && k < 64 ? k : 63)) return fail(BBB_EINVAL, "PRBS state must be < 2^k");
    return prbs_state_at_host(k, init_state, nbits, state);
}

int bbb_prbs_detector_run(int k, const uint8_t *bits_dev, uint64_t nstreams, uint64_t n, uint8_t *err_dev,
                          uint8_t *reload_dev, int device, void *hip_stream) {
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (nstreams && n && !bits_dev) return fail(BBB_EINVAL, "null device pointer");
    int rc = use_device(device);
    if (rc) return rc;
    return prbs_detector_launch(k, bits_dev, nstreams, n, err_dev, reload_dev, (hipStream_t)hip_stream);
}

int bbb_prbs_detector_stream(int k, const uint64_t *bits_packed_dev, uint64_t nbits, uint64_t *err_packed_dev,
                             uint64_t *reload_packed_dev, bbb_detector_stats *stats, uint64_t chunk_bits, uint64_t warm_bits,
                             int device, void *hip_stream) {
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (nbits && !bits_packed_dev) return fail(BBB_EINVAL, "null device pointer");
    int rc = use_device(device);
    if (rc) return rc;
    return prbs_detector_stream_launch(k, bits_packed_dev, nbits, err_packed_dev, reload_packed_dev, stats, chunk_bits,
                                       warm_bits, (hipStream_t)hip_stream);
}

// data bits needed by samples [first, first + n): indices M-7 .. M with M = floor((sample - 17) / 8)
static void tx_bit_range(uint64_t first, uint64_t n, int64_t *m0, uint64_t *nbits) {
    const int64_t lo = ((int64_t)first - 17 >= 0 ? ((int64_t)first - 17) / 8 : -1) - 7;
    const int64_t hi = (int64_t)(first + n - 1) - 17 >= 0 ? ((int64_t)(first + n - 1) - 17) / 8 : -1;
    *m0 = lo < 0 ? 0 : lo;
    *nbits = hi >= *m0 ? (uint64_t)(hi - *m0 + 1) : 0;
}

static bool tx_cfg_equal(const bbb_tx_cfg &a, const bbb_tx_cfg &b) {
    return std::memcmp(a.coeffs, b.coeffs, sizeof a.coeffs) == 0 && a.source == b.source && a.prbs_k == b.prbs_k &&
           a.prbs_state == b.prbs_state && a.bit_en == b.bit_en && a.noise_en == b.noise_en && a.noise_var == b.noise_var &&
           a.warmup == b.warmup;
}

static int tx_check(const bbb_tx_cfg *cfg) {
    if (!cfg) return fail(BBB_EINVAL, "null cfg");
    if (cfg->source != 0 && cfg->source != 1) return fail(BBB_EINVAL, "source must be 0 (PRBS) or 1 (pulse)");
    if (cfg->source == 0) {
        if (!prbs_tap(cfg->prbs_k)) return fail(BBB_EINVAL, "k=" + std::to_string(cfg->prbs_k) + " invalid for PRBS");
        if (cfg->prbs_state == 0 || (cfg->prbs_state >> cfg->prbs_k)) return fail(BBB_EINVAL, "PRBS state must be in [1, 2^k)");
    }
    if (cfg->noise_var < 0 || cfg->noise_var > 15) return fail(BBB_EINVAL, "noise_var must be 0..15");
    for (int i = 0; i < 64; i++)
        if (cfg->coeffs[i] <= -256 || cfg->coeffs[i] > 255) return fail(BBB_EINVAL, "coefficients must be in (-256, 255]");
    return BBB_OK;
}

int bbb_shaper_fill_i16(const bbb_tx_cfg *cfg, int16_t *out_dev, uint64_t nsamples, uint64_t first_sample, int device,
                        void *hip_stream) {
    int rc = tx_check(cfg);
    if (rc) return rc;
    if (nsamples == 0) return BBB_OK;
    if (!out_dev || ((uintptr_t)out_dev & 15)) return fail(BBB_EINVAL, "out must be a 16-byte aligned device pointer");
    if ((rc = use_device(device))) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    int64_t m0;
    uint64_t nbits;
    tx_bit_range(first_sample, nsamples, &m0, &nbits);
    uint64_t *d_bits = nullptr;
    if (cfg->source == 0 && nbits) {
        BBB_HIP(hipMallocAsync((void **)&d_bits, ((nbits + 63) / 64 + 2) * sizeof(uint64_t), st));
        rc = prbs_fill_launch(cfg->prbs_k, cfg->prbs_state, (uint64_t)m0, nbits, d_bits, st);
    }
    if (!rc) rc = tx_waveform_launch(cfg->coeffs, d_bits, m0, d_bits ? nbits : 0, cfg->source, nullptr, 0, 1, 0, first_sample, nsamples, out_dev, st);
    if (d_bits) (void)hipFreeAsync(d_bits, st);
    return rc;
}

int bbb_tx_fill_i16(bbb_lutopt *h, const bbb_tx_cfg *cfg, int16_t *out_dev, uint64_t nsamples, uint64_t first_sample) {
    if (!h) return fail(BBB_EINVAL, "null handle");
    int rc = tx_check(cfg);
    if (rc) return rc;
    if (nsamples == 0) return BBB_OK;
    if (!out_dev || ((uintptr_t)out_dev & 15)) return fail(BBB_EINVAL, "out must be a 16-byte aligned device pointer");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot generate samples");
    if (cfg->noise_en && (!h->specialised && h->k > 256)) return fail(BBB_EUNSUP, "TX noise needs an 8-bit CLT generator (k <= 256)");
    BBB_HIP(hipSetDevice(h->device));
    int64_t m0;
    uint64_t nbits;
    if (cfg->noise_en && h->specialised && first_sample + nsamples < (1ull << 62) && nsamples < (1ull << 31)) {
        if (cfg->warmup + first_sample + nsamples < nsamples) return fail(BBB_EINVAL, "warmup + first_sample + nsamples overflows");
        uint64_t L, G;
        unsigned nlanes;
        const bool staged = h->staged_mode && nsamples >= (1ull << 24);
        // the data bits of THIS call's samples, and where its sample 0 sits among them
        tx_bit_range(first_sample, nsamples, &m0, &nbits);
        const int64_t F = (int64_t)first_sample - 17, FM = F >> 3;            // arithmetic shift = floor
        const bool use_bits = cfg->bit_en && nbits;
        const uint32_t rel_base = (uint32_t)(FM - 7 - (m0 - 128));
        if (staged) {
            // Noise kernel + SHAPING mover: the plain sample kernel leaves its count planes in a staging slot, the mover turns
            // this call's window of them into the transmitter's int16 samples (unplane_kernel<true>): the call is bounded by
            // the noise kernel, and the int8 noise crosses HBM once each way instead of the int16 output doing so twice.
            if ((rc = ensure_internal_streams(h))) return rc;
            // The data bits: two zero 64-bit words lead (data bits before the first read as 0: the shaper's reset register).
            // One buffer per staging slot, holding the bits of ALL the windows the slot's sample kernel produces; they are
            // generated on the slot's ARITHMETIC stream, straight in front of the sample kernel -- in the gap between two sample
            // kernels.  (On the mover's stream the generator, a 100-register kernel, could not be placed beside the running
            // sample kernel and held the mover back until that kernel had ended: the transmitter's mover then ran AFTER the
            // noise kernel instead of beside it.)
            uint64_t words64 = 0;
            uint32_t *d_bits = nullptr;
            unsigned bits_buf = 0;
            uint32_t rel = rel_base;
            bool bits_on = use_bits;
            auto make_bits = [&](int slot, uint64_t nbits_all) -> int {
                words64 = 2 + (nbits_all + 63) / 64 + 3;
                // the slot's OTHER buffer than last time: its readers were the movers of the slot's kernel before last; the slot's last
                // kernel waited for them (stage_free stands for every mover of the slot so far) and was queued on this stream
                bits_buf = h->mbits_turn[slot] ^= 1u;
                if (h->mbits_cap[slot][bits_buf] < (size_t)words64 * 2) {
                    if (h->stage_busy[slot]) BBB_HIP(hipEventSynchronize(h->stage_free[slot]));      // growing frees the old buffer
                    int rcg = grow(&h->d_mbits[slot][bits_buf], &h->mbits_cap[slot][bits_buf], (size_t)words64 * 2);
                    if (rcg) return rcg;
                }
                d_bits = h->d_mbits[slot][bits_buf];
                if (!cfg->bit_en || !nbits_all) return BBB_OK;
                BBB_HIP(hipMemsetAsync(d_bits, 0, 16, h->cs));
                uint64_t *bits64 = (uint64_t *)d_bits + 2;
                return cfg->source == 0 ? prbs_fill_launch(cfg->prbs_k, cfg->prbs_state, (uint64_t)m0, nbits_all, bits64, h->cs)
                                        : pulse_bits_launch(bits64, m0, (nbits_all + 63) / 64, h->cs);
            };
            auto deliver_tx = [&](int slot, uint64_t win_lo, uint64_t Lk, uint64_t Gk, unsigned nl) {
                return queue_mover_with(h, slot, [&](const void *stage, hipStream_t ys) {
                    return unplane_tx_launch(stage, out_dev, win_lo, nsamples, (unsigned)Lk, Gk, nl, cfg->coeffs, d_bits, (uint32_t)(words64 * 2),
                                             rel, (uint32_t)(F & 7), cfg->noise_var, cfg->bit_en, bits_on ? 1 : 0, ys);
                });
            };
            // look-ahead (bbb_lutopt_set_staged(h, m >= 2)): the noise of these very samples was produced by an earlier call's
            // sample kernel and waits in its staging slot (same configuration: the noise does not depend on it, but a reader
            // who changes it mid-stream gets a fresh kernel, as before)
            if (h->ahead.valid && h->ahead.kind == 1 && h->ahead.first == first_sample && h->ahead.n == nsamples &&
                tx_cfg_equal(h->ahead.cfg, *cfg)) {
                // (no begin_op: see awgn_fill's look-ahead delivery)
                if (env_knob("BBB_EXP_DELIVER_HANDOVER", 0) && (rc = begin_op(h, true))) return rc;
                const bbb_lutopt::Ahead a = h->ahead;
                // its data bits sit in the slot's buffer, behind those of the windows before it
                d_bits = h->d_mbits[a.slot][a.bits_buf]; words64 = a.bits_words64;
                rel = (uint32_t)(FM - 7 - (a.bits_m0 - 128));
                bits_on = cfg->bit_en != 0;
                h->ahead.first += nsamples; h->ahead.step += nsamples;
                h->ahead.win_lo += nsamples;
                h->ahead.valid = --h->ahead.left > 0;
                return deliver_tx(a.slot, a.win_lo, a.L, a.G, a.nlanes);
            }
            const uint64_t mla = (uint64_t)h->staged_level;
            // (< 2^34 samples per kernel: the windows' bit offsets into the slot's data-bit buffer are 32-bit)
            const bool ahead = mla >= 2 && (nsamples % 16) == 0 && mla * nsamples < (1ull << 34) &&
                               first_sample + mla * nsamples < (1ull << 62) && cfg->warmup + first_sample + mla * nsamples >= mla * nsamples;
            const uint64_t ntotal = ahead ? mla * nsamples : nsamples;        // what the sample kernel produces
            partition(h, ntotal, 16, &L, &G, &nlanes);
            if (L > 0xffffff00ull) return fail(BBB_EINVAL, "nsamples too large for one call (segment length must fit 32 bits): split it");
            const uint64_t step0 = cfg->warmup + first_sample;                // tx.py:70-71
            const uint64_t seed_step = step0 + 1;      // (the small form of the noise kernel is given the state OF its first sample)
            const bool takes_prefetch = h->pf.valid && h->pf.first == seed_step && h->pf.L == L && h->pf.G == G;
            if ((rc = begin_op(h, true, takes_prefetch))) return rc;
            h->last_fill_tx = false;            // (what runs on the SIMDs is the plain kernel)
            bool from_pf = false;
            if ((rc = acquire_planes(h, seed_step, L, G, nlanes, true, &from_pf))) return rc;
            int64_t m0_all;
            uint64_t nbits_all;
            tx_bit_range(first_sample, ntotal, &m0_all, &nbits_all);           // (m0_all == m0: the same first sample)
            bits_on = cfg->bit_en && nbits_all;
            if ((rc = make_bits(h->stage_slot ^ 1, nbits_all))) return rc;     // (the slot produce_planes takes next)
            int slot = 0;
            // (the small-footprint placement of the noise kernel: the transmitter is bound by the kernel's guests -- a shaping
            // mover per call and the next kernel's seeding -- and beside this placement they run at the same time)
            if ((rc = produce_planes(h, L, nlanes, nullptr, from_pf, &slot, true))) return rc;
            if ((rc = deliver_tx(slot, 0, L, G, nlanes))) return rc;
            if (ahead) {
                bbb_lutopt::Ahead &a = h->ahead;
                a.valid = true; a.kind = 1; a.cfg = *cfg;
                a.first = first_sample + nsamples; a.step = step0 + nsamples;
                a.n = nsamples; a.win_lo = nsamples; a.left = (unsigned)mla - 1;
                a.L = L; a.G = G; a.nlanes = nlanes; a.slot = slot;
                a.bits_m0 = m0; a.bits_words64 = words64; a.bits_buf = bits_buf;
            }
            return BBB_OK;
        }
        // one kernel: the shaper fused into the sample kernel's round end, int16 straight to its place
        partition(h, nsamples, 16, &L, &G, &nlanes);
        if ((rc = begin_op(h, false))) return rc;
        h->last_fill_tx = true;
        // buffer: two zero 64-bit words (data bits before the first read as 0, the shaper's reset shift register), the
        // bits m0 .. m0+nbits-1, slack for the windows of rounds past the end of the request
        const uint64_t words64 = 2 + (nbits + 63) / 64 + (L / 8 + 63) / 64 + 2;
        const int bs = h->fbits_slot ^= 1;
        { const int rcs_ = ensure_side_stream(h); if (rcs_) return rcs_; }
        for (hipEvent_t *e : {&h->fbits_read[bs], &h->fbits_ready})
            if (!*e) BBB_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (h->fbits_cap[bs] < (size_t)words64 * 2) {
            if (h->fbits_pending[bs]) BBB_HIP(hipEventSynchronize(h->fbits_read[bs]));      // growing frees the old buffer
            h->fbits_pending[bs] = false;
            if ((rc = grow(&h->d_fbits[bs], &h->fbits_cap[bs], (size_t)words64 * 2))) return rc;
        }
        uint32_t *const d_bits = h->d_fbits[bs];
        if (use_bits) {
            // on the side stream: not behind the previous call's sample kernel, which is what h->cs would mean
            if (h->fbits_pending[bs]) BBB_HIP(hipStreamWaitEvent(h->side, h->fbits_read[bs], 0));
            BBB_HIP(hipMemsetAsync(d_bits, 0, 16, h->side));
            uint64_t *bits64 = (uint64_t *)d_bits + 2;
            if (cfg->source == 0) rc = prbs_fill_launch(cfg->prbs_k, cfg->prbs_state, (uint64_t)m0, nbits, bits64, h->side);
            else rc = pulse_bits_launch(bits64, m0, (nbits + 63) / 64, h->side);
            if (rc) return rc;
            BBB_HIP(hipEventRecord(h->fbits_ready, h->side));
            BBB_HIP(hipStreamWaitEvent(h->cs, h->fbits_ready, 0));
        }
        if ((rc = acquire_planes(h, cfg->warmup + first_sample, L, G, nlanes, true))) return rc;      // tx.py:70-71
        rc = awgn256_tx_launch(h->d_planes, out_dev, nsamples, (unsigned)L, G, nlanes, cfg->coeffs, d_bits, (uint32_t)(words64 * 2), rel_base,
                               (uint32_t)(F & 7), cfg->noise_var, cfg->bit_en, use_bits ? 1 : 0, h->cs);
        if (!rc) rc = mark_planes_read(h);
        if (rc) return rc;
        BBB_HIP(hipEventRecord(h->fbits_read[bs], h->cs));            // the fused kernel is the last reader of the data bits
        h->fbits_pending[bs] = true;
        return BBB_OK;
    }
    if ((rc = begin_op(h, false))) return rc;
    tx_bit_range(first_sample, nsamples, &m0, &nbits);
    const bool have_bits = cfg->source == 0 && nbits && cfg->bit_en;
    if (have_bits) {
        if ((rc = grow(&h->d_txbits, &h->txbits_cap, (size_t)((nbits + 63) / 64 + 2) * 2))) return rc;
        if ((rc = prbs_fill_launch(cfg->prbs_k, cfg->prbs_state, (uint64_t)m0, nbits, (uint64_t *)h->d_txbits, h->stream))) return rc;
    }
    if (cfg->noise_en) {
        if ((rc = grow(&h->d_txnoise, &h->txnoise_cap, (size_t)(nsamples + 15) / 4 + 4))) return rc;
        if ((rc = awgn_fill(h, h->d_txnoise, 1, nsamples, cfg->warmup + first_sample))) return rc;   // tx.py:70-71
    }
    return tx_waveform_launch(cfg->coeffs, have_bits ? (const uint64_t *)h->d_txbits : nullptr, m0, have_bits ? nbits : 0,
                              cfg->bit_en ? cfg->source : 1, (const int8_t *)h->d_txnoise, cfg->noise_var, cfg->bit_en,
                              cfg->noise_en, first_sample, nsamples, out_dev, h->stream);
}


// The transmitter as a sequential-stream object: TX.x is one sample per clock (tx.py:39-81), and a host that reads it in equal
// calls should get the staged form without choreographing bbb_lutopt_set_staged / bbb_awgn_prefetch itself (the
// counterpart of bbb_awgn_stream_* for the waveform).
struct bbb_tx_stream {
    bbb_lutopt *h = nullptr;
    bbb_tx_cfg cfg{};
    uint64_t n = 0, pos = 0;
    int saved_level = 0;
};

static int tx_stream_hint(bbb_tx_stream *s) {
    // the noise of sample p is LUTOPT clock warmup + p (tx.py:70-71): what bbb_tx_fill_i16 will ask the generator for
    if (!s->cfg.noise_en) return BBB_OK;
    const uint64_t step = s->cfg.warmup + s->pos;
    if (step < s->pos || step + s->n < step) return BBB_OK;
    return awgn_prefetch(s->h, s->n, step, true);
}

int bbb_tx_stream_open(bbb_lutopt *h, const bbb_tx_cfg *cfg, uint64_t nsamples_per_call, uint64_t first_sample, bbb_tx_stream **out) {
    if (!h || !out) return fail(BBB_EINVAL, "null argument");
    int rc = tx_check(cfg);
    if (rc) return rc;
    if (nsamples_per_call == 0) return fail(BBB_EINVAL, "nsamples_per_call must be positive");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot generate samples");
    if (h->has_stream) return fail(BBB_EINVAL, "the handle already has an open stream");
    if (first_sample + nsamples_per_call < first_sample) return fail(BBB_EINVAL, "first_sample + nsamples_per_call overflows");
    std::unique_ptr<bbb_tx_stream> s(new bbb_tx_stream);
    s->h = h; s->cfg = *cfg; s->n = nsamples_per_call; s->pos = first_sample;
    s->saved_level = h->staged_level;
    if (h->staged_level == 0 && cfg->noise_en) {      // (a caller who chose a level keeps it)
        // one noise kernel per two calls: per noise kernel the guests (a shaping mover per call, then the next kernel's
        // seeding) run one at a time and must fit into its time (1e9 samples per call: 1.44 ms at level 1, 1.28 at 2)
        if ((rc = bbb_lutopt_set_staged(h, 2))) return rc;
    }
    if ((rc = tx_stream_hint(s.get()))) return rc;
    h->has_stream = true;
    *out = s.release();
    return BBB_OK;
}

int bbb_tx_stream_read(bbb_tx_stream *s, int16_t *out_dev, uint64_t nsamples) {
    if (!s || !s->h) return fail(BBB_EINVAL, "null stream");
    if (s->pos + nsamples < s->pos) return fail(BBB_EINVAL, "the stream position overflows");
    int rc = bbb_tx_fill_i16(s->h, &s->cfg, out_dev, nsamples, s->pos);
    if (rc) return rc;
    s->pos += nsamples;
    return tx_stream_hint(s);
}

int bbb_tx_stream_next(bbb_tx_stream *s, int16_t *out_dev) {
    if (!s) return fail(BBB_EINVAL, "null stream");
    return bbb_tx_stream_read(s, out_dev, s->n);
}

int bbb_tx_stream_seek(bbb_tx_stream *s, uint64_t first_sample) {
    if (!s || !s->h) return fail(BBB_EINVAL, "null stream");
    if (first_sample + s->n < first_sample) return fail(BBB_EINVAL, "first_sample + nsamples_per_call overflows");
    s->pos = first_sample;
    s->h->ahead.valid = false;            // what a noise kernel produced ahead at the old position is dropped
    return tx_stream_hint(s);
}

int bbb_tx_stream_tell(const bbb_tx_stream *s, uint64_t *next_sample) {
    if (!s || !next_sample) return fail(BBB_EINVAL, "null argument");
    *next_sample = s->pos;
    return BBB_OK;
}

int bbb_tx_stream_close(bbb_tx_stream *s) {
    if (!s) return BBB_OK;
    int rc = BBB_OK;
    if (s->h) {
        s->h->has_stream = false;
        if (s->h->staged_level != s->saved_level) rc = bbb_lutopt_set_staged(s->h, s->saved_level);
    }
    delete s;
    return rc;
}

int bbb_rx_slice(const int16_t *samples_dev, uint64_t nsamples, uint64_t stride, uint64_t phase, int strict,
                 uint64_t *bits_packed_dev, uint64_t *nbits_out, int device, void *hip_stream) {
    if (stride == 0) return fail(BBB_EINVAL, "stride must be >= 1");
    const uint64_t nbits = phase < nsamples ? (nsamples - phase + stride - 1) / stride : 0;
    if (nbits_out) *nbits_out = nbits;
    if (nbits == 0) return BBB_OK;
    if (!samples_dev || !bits_packed_dev) return fail(BBB_EINVAL, "null device pointer");
    int rc = use_device(device);
    if (rc) return rc;
    return rx_slice_launch(samples_dev, nbits, stride, phase, strict, bits_packed_dev, (hipStream_t)hip_stream);
}

int bbb_rx_phase_search(const int16_t *samples_dev, uint64_t nsamples, uint64_t stride, uint64_t nphases, int strict, int k,
                        bbb_detector_stats *stats_out, int device, void *hip_stream) {
    if (stride == 0) return fail(BBB_EINVAL, "stride must be >= 1");
    if (!prbs_tap(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (nphases == 0) return BBB_OK;
    if (!stats_out || (nsamples && !samples_dev)) return fail(BBB_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint64_t maxbits = nsamples ? (nsamples + stride - 1) / stride : 0;
    uint64_t *bits = nullptr;
    if (maxbits) BBB_HIP(hipMalloc(&bits, ((maxbits + 63) / 64) * sizeof(uint64_t)));
    for (uint64_t p = 0; p < nphases && rc == BBB_OK; p++) {
        const uint64_t nbits = p < nsamples ? (nsamples - p + stride - 1) / stride : 0;
        stats_out[p] = bbb_detector_stats{};
        if (nbits == 0) continue;
        rc = rx_slice_launch(samples_dev, nbits, stride, p, strict, bits, st);
        if (rc == BBB_OK) rc = prbs_detector_stream_launch(k, bits, nbits, nullptr, nullptr, &stats_out[p], 0, 0, st);
    }
    if (bits) (void)hipFree(bits);
    return rc;
}

int bbb_lutopt_search(int k, uint64_t seed, uint64_t first_candidate, uint64_t ncandidates, uint64_t *found_index,
                      uint16_t *taps_out, uint32_t *row_off_out, bbb_search_stats *stats, int device, void *hip_stream) {
    if (!found_index) return fail(BBB_EINVAL, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    return lutopt_search_launch(k, seed, first_candidate, ncandidates, found_index, taps_out, row_off_out, stats,
                                (hipStream_t)hip_stream);
}

int bbb_ber_trials_dev(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, uint64_t *counters_dev) {
    if (!h || (ncfg && (!cfgs || !counters_dev)) || ncfg < 0) return fail(BBB_EINVAL, "null argument");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot run trials");
    BBB_HIP(hipSetDevice(h->device));
    return ber_run(h, cfgs, ncfg, (unsigned long long *)counters_dev);
}

// the handle's device counters and their pinned host mirror (a pageable read-back goes through the runtime's staging buffer)
static int ensure_counters(bbb_lutopt *h, size_t need) {
    if (h->counters_cap < need) {
        if (h->d_counters) BBB_HIP(hipFree(h->d_counters));
        h->d_counters = nullptr;
        h->counters_cap = 0;
        const size_t cap = need < 64 ? 64 : need;
        BBB_HIP(hipMalloc((void **)&h->d_counters, cap * sizeof(unsigned long long)));
        h->counters_cap = cap;
        h->counters_clean = false;
    }
    if (h->h_counters_cap < need) {
        if (h->h_counters) BBB_HIP(hipHostFree(h->h_counters));
        h->h_counters = nullptr;
        h->h_counters_cap = 0;
        const size_t cap = need < 64 ? 64 : need;
        BBB_HIP(hipHostMalloc((void **)&h->h_counters, cap * sizeof(unsigned long long), hipHostMallocDefault));
        h->h_counters_cap = cap;
    }
    return BBB_OK;
}

// d_counters[0 .. need) zero on h->stream: the zeroing queued behind the previous read-back is taken (its event waited for: the
// handle may have been re-bound to another stream since), else a memset now
static int take_clean_counters(bbb_lutopt *h, size_t need) {
    if (h->counters_clean && h->counters_zeroed) {
        BBB_HIP(hipStreamWaitEvent(h->stream, h->counters_zeroed, 0));
    } else {
        BBB_HIP(hipMemsetAsync(h->d_counters, 0, need * sizeof(unsigned long long), h->stream));
    }
    h->counters_clean = false;
    return BBB_OK;
}

// queues the pinned read-back of d_counters[0 .. need) on h->stream and, behind it, the zeroing for the next call
static int read_back_counters(bbb_lutopt *h, size_t need) {
    BBB_HIP(hipMemcpyAsync(h->h_counters, h->d_counters, need * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    BBB_HIP(hipMemsetAsync(h->d_counters, 0, h->counters_cap * sizeof(unsigned long long), h->stream));
    if (!h->counters_zeroed) BBB_HIP(hipEventCreateWithFlags(&h->counters_zeroed, hipEventDisableTiming));
    BBB_HIP(hipEventRecord(h->counters_zeroed, h->stream));
    h->counters_clean = true;
    return BBB_OK;
}

int bbb_ber_trials(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, bbb_ber *out) {
    if (!h || (ncfg && (!cfgs || !out)) || ncfg < 0) return fail(BBB_EINVAL, "null argument");
    if (ncfg == 0) return BBB_OK;
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot run trials");
    BBB_HIP(hipSetDevice(h->device));
    const size_t need = 2 * (size_t)ncfg;
    int rc = ensure_counters(h, need);
    if (rc) return rc;
    if ((rc = take_clean_counters(h, need))) return rc;
    rc = ber_run(h, cfgs, ncfg, h->d_counters);
    if (rc) return rc;
    if ((rc = read_back_counters(h, need))) return rc;
    BBB_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < ncfg; i++) {
        out[i].bits = h->h_counters[2 * (size_t)i];
        out[i].errors = h->h_counters[2 * (size_t)i + 1];
    }
    return BBB_OK;
}

/* ---- a trial group continued over several calls ---------------------------------------------------------- */

// One seeding per BLOCK of m calls: the block's m n bits are cut into G segments of Lb = m Lc steps, call c runs steps
// [c Lc, (c + 1) Lc) of every generator and leaves the states it ends in (TrialDev.flags: kTrialSaveState) in the run's own
// plane buffers, where call c + 1 finds them.  The bit / sample pairs of a block are those of ONE bbb_ber_trials call over
// its m n bits, visited in another order: after every m-th call the totals are that call's counters, bit for bit.
struct bbb_ber_run {
    bbb_lutopt *h = nullptr;
    std::vector<bbb_trial_cfg> cfgs;
    std::vector<TrialDev> td;
    uint32_t m = 1, call = 0;
    uint64_t n = 0, block = 0, Lb = 0, G = 0;
    unsigned nlanes = 0;
    uint32_t *d_states = nullptr, *d_planes = nullptr, *d_pstates = nullptr, *d_pplanes = nullptr;
    size_t states_cap = 0, planes_cap = 0, pstates_cap = 0, pplanes_cap = 0;
    unsigned long long *d_totals = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

int bbb_ber_run_open(bbb_lutopt *h, const bbb_trial_cfg *cfgs, int ncfg, uint32_t calls_per_block, bbb_ber_run **out) {
    if (!h || !cfgs || !out || ncfg < 1) return fail(BBB_EINVAL, "null argument");
    if (ncfg > BBB_BER_MAX_GROUP) return fail(BBB_EINVAL, "a continued trial group holds at most BBB_BER_MAX_GROUP settings");
    if (calls_per_block < 1 || calls_per_block > 4096) return fail(BBB_EINVAL, "calls_per_block must be in 1..4096");
    if (h->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot run trials");
    if (!h->specialised && !h->custom_ber)
        return fail(BBB_EUNSUP, "BER trials need the shipped n256 matrix, or a k = 256 matrix with its own kernels attached");
    BBB_HIP(hipSetDevice(h->device));
    const bbb_trial_cfg &c0 = cfgs[0];
    if (c0.nbits == 0) return fail(BBB_EINVAL, "nbits (bits per call) must be positive");
    std::unique_ptr<bbb_ber_run> r(new bbb_ber_run);
    r->h = h;
    r->cfgs.assign(cfgs, cfgs + ncfg);
    r->td.resize((size_t)ncfg);
    for (int i = 0; i < ncfg; i++) {
        const bbb_trial_cfg &c = cfgs[i];
        const int tap = prbs_tap(c.prbs_k);
        if (!tap) return fail(BBB_EINVAL, "k=" + std::to_string(c.prbs_k) + " invalid for PRBS");
        if (c.prbs_state == 0 || (c.prbs_state >> c.prbs_k)) return fail(BBB_EINVAL, "PRBS state must be in [1, 2^k)");
        if (c.amp < 0 || c.amp > 2047 || c.noise_var < 0 || c.noise_var > 15) return fail(BBB_EINVAL, "amp must be 0..2047 and noise_var 0..15");
        if (c.prbs_k != c0.prbs_k || c.prbs_state != c0.prbs_state || c.warmup != c0.warmup || c.first_bit != c0.first_bit || c.nbits != c0.nbits)
            return fail(BBB_EINVAL, "the settings of a continued trial group must share the PRBS, the offsets and nbits (they read one noise stream)");
        r->td[(size_t)i].prbs_k = c.prbs_k;
        r->td[(size_t)i].prbs_tap = tap;
        int rc = channel_thresholds(c.amp, c.noise_var, &r->td[(size_t)i]);
        if (rc) return rc;
        int real[2] = {0, 0};
        for (int bv = 0; bv < 2; bv++)
            for (int j = 0; j < r->td[(size_t)i].nthr[bv]; j++) real[bv] += r->td[(size_t)i].thr[bv][j] > 0 && r->td[(size_t)i].thr[bv][j] < 256;
        if (ncfg > 1 && (real[0] != 1 || real[1] != 1))
            return fail(BBB_EINVAL, "grouped trials must be single-threshold (a wrap-around setting runs alone)");
    }
    r->m = calls_per_block;
    r->n = c0.nbits;
    const uint64_t nblock = r->n * (uint64_t)r->m;
    if (nblock / r->m != r->n || c0.warmup + c0.first_bit + nblock < nblock) return fail(BBB_EINVAL, "bits per block overflow");
    // segments in multiples of 2 m steps: every call runs an even number of steps of every generator
    partition(h, nblock, 2 * r->m, &r->Lb, &r->G, &r->nlanes);
    if (r->Lb / r->m >= (1ull << 27)) return fail(BBB_EINVAL, "nbits too large for one call (about 2^47): lower it");
    BBB_HIP(hipMalloc((void **)&r->d_totals, 2 * (size_t)ncfg * sizeof(unsigned long long)));
    if (hipMemset(r->d_totals, 0, 2 * (size_t)ncfg * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipFree(r->d_totals);            // (the unique_ptr frees the struct only)
        return fail(BBB_EHIP, "hipMemset of the run's totals failed");
    }
    *out = r.release();
    return BBB_OK;
}

// queues the next call of the run on the handle's stream; its counters are ADDED to counters_dev (device, [ncfg][2])
static int ber_run_step(bbb_ber_run *r, unsigned long long *counters_dev) {
    bbb_lutopt *h = r->h;
    BBB_HIP(hipSetDevice(h->device));
    int rc = begin_op(h, false);
    if (rc) return rc;
    const bbb_trial_cfg &c = r->cfgs[0];
    const int ncfg = (int)r->cfgs.size();
    const uint64_t nblock = r->n * (uint64_t)r->m, block_first = c.first_bit + r->block * nblock;
    if (block_first + nblock < block_first || c.warmup + block_first + nblock + 1 < nblock) return fail(BBB_EINVAL, "stream position overflows");
    if (r->call == 0) {
        // a new block: start states of its G generators, one clock past the stream position (the kernels take the state OF
        // their first sample), into the run's own buffers -- other calls on the handle keep theirs
        JumpPlan *plan, *pp;
        if ((rc = get_plan(h, r->Lb, &plan))) return rc;
        if ((rc = get_prbs_plan(h, c.prbs_k, r->Lb, &pp))) return rc;
        if ((rc = ensure_top_tables(plan))) return rc;
        if ((rc = grow(&r->d_states, &r->states_cap, (size_t)65536 * h->W32))) return rc;
        if ((rc = grow(&r->d_planes, &r->planes_cap, (size_t)h->k * r->nlanes))) return rc;
        if ((rc = grow(&r->d_pplanes, &r->pplanes_cap, (size_t)32 * r->nlanes))) return rc;
        uint64_t s0[8];
        h->pw->apply(c.warmup + block_first + 1, h->init, s0);
        uint32_t s16[256];
        first16(*plan, s0, s16);
        uint64_t ps0 = 0;
        if ((rc = prbs_state_at_host(c.prbs_k, c.prbs_state, block_first, &ps0))) return rc;
        uint64_t ps64[8] = {ps0};
        uint32_t ps16[256];
        first16(*pp, ps64, ps16);
        { const int rcs_ = ensure_side_stream(h); if (rcs_) return rcs_; }
        for (hipEvent_t *e : {&r->fork, &r->join})
            if (!*e) BBB_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        BBB_HIP(hipEventRecord(r->fork, h->cs));                      // (the previous block's last kernel still reads the buffers)
        BBB_HIP(hipStreamWaitEvent(h->side, r->fork, 0));
        if ((rc = prbs_seed_lanes_launch(c.prbs_k, pp->d_cols, ps16, pp->qcol64, r->G, r->nlanes, r->d_pplanes, h->side))) return rc;
        BBB_HIP(hipEventRecord(r->join, h->side));
        if ((rc = awgn_seed_head_launch(h->k, plan->d_cols, s16, r->G, r->d_states, h->cs))) return rc;
        if ((rc = awgn_seed_tail_planes_launch(h->k, plan->d_top, r->G, r->d_states, r->nlanes, r->d_planes, h->cs))) return rc;
        BBB_HIP(hipStreamWaitEvent(h->cs, r->join, 0));
    }
    const uint64_t Lc = r->Lb / r->m;
    // the (possibly short) last generator of the block: it owns block_len - (G - 1) Lb steps, of which this call sees
    // those inside [call Lc, (call + 1) Lc)
    const uint64_t last_total = nblock - (r->G - 1) * r->Lb, lo = (uint64_t)r->call * Lc;
    const uint64_t last_len = last_total <= lo ? 0 : (last_total - lo < Lc ? last_total - lo : Lc);
    const uint64_t bits_now = (r->G - 1) * Lc + last_len;
    for (int i = 0; i < ncfg; i++) {
        TrialDev &t = r->td[(size_t)i];
        t.L = (uint32_t)Lc; t.G = r->G; t.nbits = bits_now;
        t.flags = kTrialLastLen | kTrialSaveState; t.last_len = (uint32_t)last_len;
    }
    if (h->specialised) {
        if ((rc = ber256_launch(r->d_planes, r->d_pplanes, r->td.data(), ncfg, r->nlanes, counters_dev, h->cs))) return rc;
    } else {
        const int e = h->custom_ber(r->d_planes, r->d_pplanes, r->td.data(), ncfg, r->nlanes, (uint64_t *)counters_dev, (void *)h->cs);
        if (e) return fail(e < 0 ? e : BBB_EHIP, "custom BER kernel failed");
    }
    if (++r->call == r->m) { r->call = 0; r->block++; }
    return BBB_OK;
}

int bbb_ber_run_next_dev(bbb_ber_run *r, uint64_t *counters_dev) {
    if (!r || !counters_dev) return fail(BBB_EINVAL, "null argument");
    return ber_run_step(r, (unsigned long long *)counters_dev);
}

int bbb_ber_run_next(bbb_ber_run *r, bbb_ber *totals) {
    if (!r) return fail(BBB_EINVAL, "null argument");
    int rc = ber_run_step(r, r->d_totals);
    if (rc || !totals) return rc;
    const size_t need = 2 * r->cfgs.size();
    std::vector<unsigned long long> host(need);
    BBB_HIP(hipMemcpyAsync(host.data(), r->d_totals, need * sizeof(unsigned long long), hipMemcpyDeviceToHost, r->h->stream));
    BBB_HIP(hipStreamSynchronize(r->h->stream));
    for (size_t i = 0; i < r->cfgs.size(); i++) {
        totals[i].bits = host[2 * i];
        totals[i].errors = host[2 * i + 1];
    }
    return BBB_OK;
}

int bbb_ber_run_tell(const bbb_ber_run *r, uint64_t *calls_done, uint64_t *next_block_first_bit) {
    if (!r) return fail(BBB_EINVAL, "null argument");
    if (calls_done) *calls_done = r->block * r->m + r->call;
    if (next_block_first_bit) *next_block_first_bit = r->cfgs[0].first_bit + (r->block + (r->call ? 1 : 0)) * r->n * (uint64_t)r->m;
    return BBB_OK;
}

int bbb_ber_run_close(bbb_ber_run *r) {
    if (!r) return BBB_OK;
    if (r->h && r->h->device >= 0) (void)hipSetDevice(r->h->device);
    if (r->h && r->h->stream) (void)hipStreamSynchronize(r->h->stream);
    if (r->h && r->h->cs_valid) (void)hipStreamSynchronize(r->h->cs);
    if (r->h && r->h->side) (void)hipStreamSynchronize(r->h->side);
    for (void *p : {(void *)r->d_states, (void *)r->d_planes, (void *)r->d_pstates, (void *)r->d_pplanes, (void *)r->d_totals})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : {r->fork, r->join})
        if (e) (void)hipEventDestroy(e);
    delete r;
    return BBB_OK;
}

/* ---- the path's one collective: a sweep sharded over the GPUs of this process ------------------------- */

int bbb_sweep_shard(const bbb_trial_cfg *cfgs, int ncfg, int ndev, int rank, int mode, bbb_trial_cfg *mine) {
    const int e = sweep_shard(cfgs, ncfg, ndev, rank, mode, mine);
    if (e == -2) return fail(BBB_EINVAL, "first_bit + nbits overflows");
    if (e) return fail(BBB_EINVAL, "bad shard arguments (ndev >= 1, 0 <= rank < ndev, mode BBB_SHARD_TRIALS / _SEEDS / _BITS)");
    return BBB_OK;
}

namespace {

// communicators of this process, one set per ordered device list (creation costs far more than a sweep)
struct CommSet { std::vector<int> devs; std::vector<ncclComm_t> comms; };
std::mutex g_comm_mu;
std::vector<CommSet> g_comm_sets;

int get_comms(const std::vector<int> &devs, std::vector<ncclComm_t> *out) {
    const Rccl &r = rccl();
    if (!r.ok) return fail(BBB_EUNSUP, r.error);
    std::lock_guard<std::mutex> g(g_comm_mu);
    for (const CommSet &c : g_comm_sets)
        if (c.devs == devs) { *out = c.comms; return BBB_OK; }
    CommSet c;
    c.devs = devs;
    c.comms.resize(devs.size());
    const ncclResult_t e = r.CommInitAll(c.comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) return fail(BBB_EHIP, std::string("ncclCommInitAll: ") + r.GetErrorString(e));
    g_comm_sets.push_back(c);
    *out = c.comms;
    return BBB_OK;
}

}  // namespace

namespace {
std::mutex g_multi_info_mu;
bbb_multi_info g_multi_info{};         // what the last bbb_ber_sweep_multi of this process saw
}  // namespace

int bbb_multi_last_info(bbb_multi_info *out) {
    if (!out) return fail(BBB_EINVAL, "null argument");
    std::lock_guard<std::mutex> g(g_multi_info_mu);
    *out = g_multi_info;
    return BBB_OK;
}

int bbb_multi_release(void) {
    {   // the pooled internal streams as well: a host that is about to call hipDeviceReset releases them here first (handles created
        // afterwards get fresh ones; handles that are still alive must have been destroyed before)
        std::lock_guard<std::mutex> gs(g_streams_mu);
        for (auto &kv : g_streams) {
            if (hipSetDevice(kv.first) != hipSuccess) continue;
            for (hipStream_t st : {kv.second.xs[0], kv.second.xs[1], kv.second.side})
                if (st) (void)hipStreamDestroy(st);
        }
        g_streams.clear();
    }
    std::lock_guard<std::mutex> g(g_comm_mu);
    const Rccl &r = rccl();
    for (CommSet &c : g_comm_sets)
        for (ncclComm_t comm : c.comms)
            if (r.ok && comm) (void)r.CommDestroy(comm);
    g_comm_sets.clear();
    return BBB_OK;
}

int bbb_ber_sweep_multi(bbb_lutopt *const *handles, int ndev, const bbb_trial_cfg *cfgs, int ncfg, int mode, bbb_ber *out) {
    if (!handles || ndev < 1 || ncfg < 0 || (ncfg && (!cfgs || !out))) return fail(BBB_EINVAL, "null argument");
    if (ncfg == 0) return BBB_OK;
    // the calling thread's current device is restored on every way out (the loops below select each handle's device in turn)
    struct DeviceGuard {
        int dev = -1;
        DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
        ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    } device_guard;
    // REHEARSAL (experiments build only, BBB_MULTI_REHEARSAL=1): handles may share a device and the all-reduce is replaced by
    // a host-side sum -- what a one-GPU box can exercise of this function: ndev host threads inside ber_run at once (plan
    // caches, per-device statics, the per-thread error text), the sharding, the bookkeeping around the collective
    const bool rehearsal = env_knob("BBB_MULTI_REHEARSAL", 0) != 0;
    std::vector<int> devs((size_t)ndev);
    for (int r = 0; r < ndev; r++) {
        if (!handles[r]) return fail(BBB_EINVAL, "null handle");
        if (handles[r]->device < 0) return fail(BBB_ENODEV, "host-only handle (device -1) cannot run trials");
        devs[(size_t)r] = handles[r]->device;
        for (int q = 0; q < r; q++) {
            if (handles[q] == handles[r]) return fail(BBB_EINVAL, "the same handle twice");
            if (devs[(size_t)q] == devs[(size_t)r] && !rehearsal) return fail(BBB_EINVAL, "every handle must live on its own device");
        }
    }
    {   // argument errors are found before any device is touched (and before a rank could miss the collective)
        std::vector<bbb_trial_cfg> probe((size_t)ncfg);
        const int e = bbb_sweep_shard(cfgs, ncfg, ndev, 0, mode, probe.data());
        if (e) return e;
    }
    std::vector<ncclComm_t> comms;
    int rc = rehearsal ? BBB_OK : get_comms(devs, &comms);
    if (rc) return rc;
    const size_t nwords = 2 * (size_t)ncfg;
    // one host thread per device launches that device's share of the trials (csrc/sweep_threads.hpp: persistent workers, rank 0 on
    // this thread; the same orchestration is built for the host under ThreadSanitizer with a stub launch, tests/san_sweep.cpp)
    {
        std::string err;
        int bad = 0;
        rc = run_shares_on_threads(cfgs, ncfg, ndev, mode, [&](int r, const bbb_trial_cfg *mine, std::string *etext) -> int {
            bbb_lutopt *h = handles[r];
            auto body = [&]() -> int {
                BBB_HIP(hipSetDevice(h->device));
                int e = ensure_counters(h, nwords);
                if (e) return e;
                if ((e = take_clean_counters(h, nwords))) return e;
                return ber_run(h, mine, ncfg, h->d_counters);
            };
            const int e = body();
            if (e) *etext = last_error();          // (the error text is per thread)
            return e;
        }, &err, &bad);
        if (rc) return fail(rc, "device " + std::to_string(devs[(size_t)bad]) + ": " + err);
    }
    // The ONE collective of the path: all-reduce (sum) of the uint64 {bits, errors} counters over xGMI, queued on
    // each device's stream behind its trials.  Every rank must enter it, so it is issued only after all shares
    // were launched without error; one group call from this thread (the single-process multi-device form).
    if (!rehearsal) {
        const Rccl &nccl = rccl();
        // (over ONE device the sum is the identity: the communicator is still created and asked for its size -- RCCL loads and answers
        // on this host -- but no collective is queued: a one-rank ncclAllReduce cost 10 us in most processes and 100 us in some)
        if (ndev > 1) {
            ncclResult_t e = nccl.GroupStart();
            for (int r = 0; r < ndev && e == ncclSuccess; r++)
                e = nccl.AllReduce(handles[r]->d_counters, handles[r]->d_counters, nwords, ncclUint64, ncclSum, comms[(size_t)r],
                                   handles[r]->stream);
            const ncclResult_t e2 = nccl.GroupEnd();
            if (e == ncclSuccess) e = e2;
            if (e != ncclSuccess) return fail(BBB_EHIP, std::string("ncclAllReduce: ") + nccl.GetErrorString(e));
        }
        // what the communicator itself says (the first multi-device run checks itself with this: bbb_multi_last_info)
        int seen = 0;
        if (nccl.CommCount(comms[0], &seen) != ncclSuccess) seen = -1;
        std::lock_guard<std::mutex> gi(g_multi_info_mu);
        g_multi_info = bbb_multi_info{};
        g_multi_info.n_devices = ndev;
        g_multi_info.n_ranks_seen = seen;
        g_multi_info.rccl_reused = nccl.reused ? 1 : 0;
        std::snprintf(g_multi_info.rccl_path, sizeof g_multi_info.rccl_path, "%s", nccl.path.c_str());
    } else {
        std::lock_guard<std::mutex> gi(g_multi_info_mu);
        g_multi_info = bbb_multi_info{};
        g_multi_info.n_devices = ndev;
        g_multi_info.n_ranks_seen = 0;         // rehearsal: no communicator
    }
    // ONE pinned read-back per device, the zeroing for the next call queued behind it, one synchronisation per device
    for (int r = 0; r < ndev; r++) {
        BBB_HIP(hipSetDevice(handles[r]->device));
        if ((rc = read_back_counters(handles[r], nwords))) return rc;
    }
    for (int r = 0; r < ndev; r++) {
        BBB_HIP(hipSetDevice(handles[r]->device));
        BBB_HIP(hipStreamSynchronize(handles[r]->stream));
    }
    std::vector<std::vector<unsigned long long>> host((size_t)ndev);
    for (int r = 0; r < ndev; r++) host[(size_t)r].assign(handles[r]->h_counters, handles[r]->h_counters + nwords);
    if (rehearsal) {                  // the sum the collective would have left on every device
        for (int r = 1; r < ndev; r++)
            for (size_t i = 0; i < nwords; i++) host[0][i] += host[(size_t)r][i];
    } else {
        for (int r = 1; r < ndev; r++)
            if (host[(size_t)r] != host[0]) return fail(BBB_EHIP, "ranks disagree after the all-reduce");
    }
    for (int i = 0; i < ncfg; i++) {
        out[i].bits = host[0][2 * (size_t)i];
        out[i].errors = host[0][2 * (size_t)i + 1];
    }
    return BBB_OK;
}

}  // extern "C"
