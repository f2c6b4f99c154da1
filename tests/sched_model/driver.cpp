// Random call sequences on the library's C ABI, over the stream / event model of model.cpp.
//   driver <taps file> <number of sequences> <seed> [mode [max_bad]]   mode: "all" (default) or "hints" (fills and announcements only)
// A sequence is 8-40 calls of one kind after another on one handle -- fills that match an announcement and fills that do
// not, announcements that are never taken, look-ahead levels, the stream objects, the transmitter, BER trials and continued
// trials, word fills, a re-bound caller stream, host synchronisations, profiling on and off, now and then a new handle.
// Around every call the driver plays the CALLER: it writes the destination buffer on its stream before the call (what a
// consumer of the previous contents would be doing) and reads it on its stream right after -- the contract of include/bbb.h is
// that work queued on the handle's stream after a call sees the output complete, and that the library does not touch the
// buffer before what the caller had queued.  The model checks every access of every buffer for ordering.
// Exit code 0: no unordered access in any sequence; 1: at least one (the first reports are printed with their traces).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/bbb.h"
#include "model.hpp"

#define CK(call)                                                                                                       \
    do {                                                                                                               \
        const int rc_ = (call);                                                                                        \
        if (rc_ != BBB_OK) { std::fprintf(stderr, "%s failed: %s (%s)\n", #call, bbb_strerror(rc_), bbb_last_error_detail()); std::exit(2); } \
    } while (0)

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    uint64_t below(uint64_t n) { return n ? g() % n : 0; }
    bool chance(int pct) { return (int)below(100) < pct; }
};

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: driver <taps> <nseq> <seed> [all|hints]\n"); return 2; }
    const std::string mode = argc > 4 ? argv[4] : "all";
    const long max_bad = argc > 5 ? std::atol(argv[5]) : 1000000;      // stop after this many sequences with a report
    std::vector<uint16_t> taps;
    std::vector<uint32_t> off;
    {
        std::ifstream f(argv[1]);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty()) continue;
            off.push_back((uint32_t)taps.size());
            std::istringstream is(line);
            int v;
            while (is >> v) taps.push_back((uint16_t)v);
        }
        off.push_back((uint32_t)taps.size());
    }
    const int k = (int)off.size() - 1;
    if (k != 256) { std::fprintf(stderr, "expected the n256 tap list\n"); return 2; }
    const long nseq = std::atol(argv[2]);
    Rng rng((uint64_t)std::atoll(argv[3]));

    hipStream_t user[2];
    hipStreamCreateWithFlags(&user[0], 0);
    hipStreamCreateWithFlags(&user[1], 0);
    hipEvent_t uev;
    hipEventCreate(&uev);
    // caller-side buffers (address space only)
    const uint64_t kMax = 3ull << 24;
    // two sets, one per caller stream: a caller that re-binds the handle WITHOUT ordering its two streams (allowed: nothing of
    // its own is shared between them) uses other buffers on the other stream
    void *sets[2][4] = {};
    for (int i = 0; i < 2; i++) {
        const std::string sfx = i ? " (stream 1)" : " (stream 0)";
        hipMalloc(&sets[i][0], kMax + 64); model::tag(sets[i][0], "caller: int8 samples" + sfx);
        hipMalloc(&sets[i][1], 2 * (kMax + 64)); model::tag(sets[i][1], "caller: int16 samples" + sfx);
        hipMalloc(&sets[i][2], 2 * 8 * 12); model::tag(sets[i][2], "caller: BER counters" + sfx);
        hipMalloc(&sets[i][3], 1 << 20); model::tag(sets[i][3], "caller: state words" + sfx);
    }
    void *dst8 = sets[0][0], *dst16 = sets[0][1], *counters = sets[0][2], *words = sets[0][3];
    auto use_set = [&](int i) { dst8 = sets[i][0]; dst16 = sets[i][1]; counters = sets[i][2]; words = sets[i][3]; };

    const uint64_t sizes[4] = {1ull << 24, (1ull << 24) + 4096, 1ull << 25, 3ull << 24};
    bbb_tx_cfg tx{};
    for (int i = 0; i < 64; i++) tx.coeffs[i] = (int16_t)(i - 32);
    tx.source = 0; tx.prbs_k = 31; tx.prbs_state = 1; tx.bit_en = 1; tx.noise_en = 1; tx.noise_var = 8; tx.warmup = 16;

    uint64_t init[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    bbb_lutopt *h = nullptr;
    int cur = 0;                       // which caller stream the handle is bound to
    auto fresh_handle = [&]() {
        if (h) CK(bbb_lutopt_destroy(h));
        CK(bbb_lutopt_create(&h, k, taps.data(), off.data(), init, 0));
        cur = 0;
        use_set(0);
        CK(bbb_lutopt_set_stream(h, (void *)user[0]));
    };
    fresh_handle();

    long bad_sequences = 0;
    uint64_t calls = 0;
    for (long seq = 0; seq < nseq; seq++) {
        if (seq % 400 == 399) fresh_handle();
        hipDeviceSynchronize();                     // a sequence starts on an idle device
        model::reset_trace();
        model::host_note("sequence " + std::to_string(seq));
        bbb_awgn_stream *ns = nullptr;
        bbb_tx_stream *ts = nullptr;
        bbb_ber_run *run = nullptr;
        uint64_t pos = 16 + rng.below(1000) * 16;          // where a "sequential reader" is
        uint64_t n = sizes[rng.below(4)];
        const int ncalls = 8 + (int)rng.below(33);
        for (int c = 0; c < ncalls; c++, calls++) {
            hipStream_t us = user[cur];
            int what = (int)rng.below(mode == "hints" ? 5 : 16);
            if ((ns || ts) && what >= 2 && what <= 4 && rng.chance(70)) what = 0;      // (an open stream object: mostly reads of it)
            switch (what) {
            case 0: case 1: {          // a fill: through the open stream object, or plain -- at the reader's position or elsewhere
                if (ns) {
                    model::op(us, "CALLER writes dst (previous consumer)", {}, {dst8});
                    if (rng.chance(80)) { model::host_note("bbb_awgn_stream_next"); CK(bbb_awgn_stream_next(ns, dst8)); }
                    else { const uint64_t m = 16 + rng.below(1u << 20); model::host_note("bbb_awgn_stream_read " + std::to_string(m)); CK(bbb_awgn_stream_read(ns, dst8, m)); }
                    model::op(us, "CALLER reads dst", {dst8}, {});
                } else if (ts) {
                    model::op(us, "CALLER writes dst16 (previous consumer)", {}, {dst16});
                    if (rng.chance(80)) { model::host_note("bbb_tx_stream_next"); CK(bbb_tx_stream_next(ts, (int16_t *)dst16)); }
                    else { const uint64_t m = 16 + rng.below(1u << 20); model::host_note("bbb_tx_stream_read " + std::to_string(m)); CK(bbb_tx_stream_read(ts, (int16_t *)dst16, m)); }
                    model::op(us, "CALLER reads dst16", {dst16}, {});
                } else {
                    const bool here = rng.chance(75);
                    const uint64_t first = here ? pos : 16 + rng.below(1u << 30) * 16;
                    const uint64_t nn = rng.chance(85) ? n : sizes[rng.below(4)];
                    model::op(us, "CALLER writes dst (previous consumer)", {}, {dst8});
                    model::host_note("bbb_awgn_fill_i8 n " + std::to_string(nn) + " first " + std::to_string(first));
                    CK(bbb_awgn_fill_i8(h, (int8_t *)dst8, nn, first));
                    model::op(us, "CALLER reads dst", {dst8}, {});
                    if (here) pos = first + nn;
                }
                break;
            }
            case 2: {                  // an announcement: the right one, or one that will not be taken
                if (ns || ts) break;
                const bool right = rng.chance(65);
                const uint64_t first = right ? pos : 16 + rng.below(1u << 30) * 16;
                const uint64_t nn = right || rng.chance(50) ? n : sizes[rng.below(4)];
                model::host_note(std::string("bbb_awgn_prefetch ") + (right ? "(right) " : "(will not be taken) ") + std::to_string(nn) + " first " + std::to_string(first));
                CK(bbb_awgn_prefetch(h, nn, first));
                break;
            }
            case 3: {                  // staging level
                if (ns || ts) break;
                const int lv[5] = {0, 1, 1, 2, 4};
                const int l = lv[rng.below(5)];
                model::host_note("bbb_lutopt_set_staged " + std::to_string(l));
                CK(bbb_lutopt_set_staged(h, l));
                break;
            }
            case 4: {                  // the reader changes its request size
                n = sizes[rng.below(4)];
                break;
            }
            case 5: {                  // noise stream object: open / seek / close
                if (ts) break;
                if (!ns) { model::host_note("bbb_awgn_stream_open"); CK(bbb_awgn_stream_open(h, n, pos, 1, &ns)); }
                else if (rng.chance(50)) { pos = 16 + rng.below(1u << 30) * 16; model::host_note("bbb_awgn_stream_seek"); CK(bbb_awgn_stream_seek(ns, pos)); }
                else { uint64_t t = 0; CK(bbb_awgn_stream_tell(ns, &t)); pos = t; model::host_note("bbb_awgn_stream_close"); CK(bbb_awgn_stream_close(ns)); ns = nullptr; }
                break;
            }
            case 6: {                  // transmitter stream object
                if (ns) break;
                if (!ts) { tx.noise_en = rng.chance(85); model::host_note("bbb_tx_stream_open"); CK(bbb_tx_stream_open(h, &tx, n, 8 * rng.below(1u << 20), &ts)); }
                else if (rng.chance(40)) { model::host_note("bbb_tx_stream_seek"); CK(bbb_tx_stream_seek(ts, 8 * rng.below(1u << 20))); }
                else { model::host_note("bbb_tx_stream_close"); CK(bbb_tx_stream_close(ts)); ts = nullptr; }
                break;
            }
            case 7: {                  // a plain transmitter call (with and without noise, here and elsewhere)
                if (ns || ts) break;
                tx.noise_en = rng.chance(80);
                const uint64_t first = rng.chance(60) ? pos : 8 * rng.below(1u << 24);
                model::op(us, "CALLER writes dst16 (previous consumer)", {}, {dst16});
                model::host_note("bbb_tx_fill_i16 n " + std::to_string(n) + " first " + std::to_string(first));
                CK(bbb_tx_fill_i16(h, &tx, (int16_t *)dst16, n, first));
                model::op(us, "CALLER reads dst16", {dst16}, {});
                pos = first + n;
                break;
            }
            case 8: {                  // BER trials on the handle
                bbb_trial_cfg t[3]{};
                const int nt = 1 + (int)rng.below(3);
                // (often the same few trials: the library keeps the start states of its last two and hands them to a repeated trial --
                // which may come on the caller's other stream)
                for (int i = 0; i < nt; i++) { t[i].prbs_k = 31; t[i].amp = 90 + 10 * i; t[i].noise_var = 8; t[i].prbs_state = 1; t[i].warmup = 16; t[i].first_bit = rng.below(4) * 1000; t[i].nbits = rng.chance(60) ? (1u << 20) + 7 * rng.below(2) : 1 + rng.below(1u << 22); }
                model::op(us, "CALLER zeroes the counters", {}, {counters});
                model::host_note("bbb_ber_trials_dev x" + std::to_string(nt));
                CK(bbb_ber_trials_dev(h, t, nt, (uint64_t *)counters));
                model::op(us, "CALLER reads the counters", {counters}, {});
                break;
            }
            case 9: {                  // continued trials
                if (!run) {
                    bbb_trial_cfg t[2]{};
                    for (int i = 0; i < 2; i++) { t[i].prbs_k = 31; t[i].amp = 90 + 10 * i; t[i].noise_var = 8; t[i].prbs_state = 1; t[i].warmup = 16; t[i].first_bit = 0; t[i].nbits = 1u << 22; }
                    model::host_note("bbb_ber_run_open");
                    CK(bbb_ber_run_open(h, t, 2, 1 + (uint32_t)rng.below(3), &run));
                } else if (rng.chance(75)) {
                    model::op(us, "CALLER touches the counters", {counters}, {counters});
                    model::host_note("bbb_ber_run_next_dev");
                    CK(bbb_ber_run_next_dev(run, (uint64_t *)counters));
                    model::op(us, "CALLER reads the counters", {counters}, {});
                } else {
                    model::host_note("bbb_ber_run_close");
                    CK(bbb_ber_run_close(run)); run = nullptr;
                }
                break;
            }
            case 10: {                 // uniform words
                if (ns || ts) break;
                model::op(us, "CALLER writes the word buffer", {}, {words});
                model::host_note("bbb_lutopt_fill_words");
                CK(bbb_lutopt_fill_words(h, (uint32_t *)words, 1 + rng.below(1u << 14), rng.below(1u << 20), (int)rng.below(2)));
                model::op(us, "CALLER reads the word buffer", {words}, {});
                break;
            }
            case 11: {                 // the caller re-binds the handle to its other stream
                if (ns || ts || run) break;
                // Half of the time the caller orders its streams (what it queued on the old one is in front of what it queues on
                // the new one); otherwise it does not -- it works on another set of buffers there, and whatever of the LIBRARY's
                // the two streams share (staging slots a mover still reads, start states) is the library's to order
                const bool ordered = rng.chance(50);
                if (ordered) hipEventRecord(uev, user[cur]);
                cur ^= 1;
                if (ordered) hipStreamWaitEvent(user[cur], uev, 0);
                use_set(cur);
                model::host_note(std::string("bbb_lutopt_set_stream -> user stream ") + std::to_string(cur) + (ordered ? " (ordered)" : " (NOT ordered)"));
                CK(bbb_lutopt_set_stream(h, (void *)user[cur]));
                break;
            }
            case 12: hipStreamSynchronize(us); break;
            case 13: { model::host_note("bbb_lutopt_profile"); CK(bbb_lutopt_profile(h, (int)rng.below(2))); double a, b; uint64_t cc; CK(bbb_lutopt_profile_read(h, &a, &b, &cc, 1)); CK(bbb_lutopt_profile_read_mover(h, &a, &cc, 1)); break; }
            case 14: {                 // a short fill (below the staged threshold: one kernel on the caller's stream)
                if (ns || ts) break;
                model::op(us, "CALLER writes dst (previous consumer)", {}, {dst8});
                model::host_note("bbb_awgn_fill_i8 (short)");
                CK(bbb_awgn_fill_i8(h, (int8_t *)dst8, 16 * (1 + rng.below(1000)), pos));
                model::op(us, "CALLER reads dst", {dst8}, {});
                break;
            }
            default: break;
            }
            if (!model::errors().empty()) break;
        }
        if (ns) CK(bbb_awgn_stream_close(ns));
        if (ts) CK(bbb_tx_stream_close(ts));
        if (run) CK(bbb_ber_run_close(run));
        if (!model::errors().empty()) {
            bad_sequences++;
            if (bad_sequences <= 3)
                for (const std::string &e : model::errors()) std::fprintf(stderr, "sequence %ld: UNORDERED ACCESS\n  %s\n", seq, e.c_str());
            model::clear_errors();
            if (bad_sequences >= max_bad) break;
            fresh_handle();
        }
    }
    if (h) CK(bbb_lutopt_destroy(h));
    hipEventDestroy(uev);
    for (auto &st : sets) for (void *p : st) hipFree(p);
    std::printf("{\"sequences\": %ld, \"calls\": %llu, \"operations_checked\": %llu, \"sequences_with_unordered_access\": %ld}\n", nseq,
                (unsigned long long)calls, (unsigned long long)model::ops_checked(), bad_sequences);
    return bad_sequences ? 1 : 0;
}
