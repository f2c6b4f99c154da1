// bbb_mc -- a plain C++ caller of the C ABI (include/bbb.h): Eb/N0 sweep of BPSK over PRBS-k through the
// LUTOPT-256 / CLT noise generator, the Monte-Carlo loop of BASELINE.json configs[3].  No torch, no Python:
// the library, hipMalloc'd buffers and printf.  Build: make -C examples   (hipcc, links ../basebandboard_amd/libbbb_hip.so)
//
//   bbb_mc [--matrix FILE] [--init HEX] [--seeds N] [--k 31] [--bits 1e9] [--nv 8] [--from 0] [--to 10] [--step 1]
//          [--loopback BITS]
//
// --matrix takes the reference's 0/1 text format (software/rnghunt/matrices/256) or nothing (the shipped
// matrix).  --init is the generator's reset state (hex, default 1: gateware/bbb/rng.py:21); --seeds N repeats the sweep
// with reset states init, init+1, ... and sums the counters (the points x seeds shape of BASELINE.json configs[4]).
// --loopback additionally runs generator -> exact detector on BITS bits.
#include "../include/bbb.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != BBB_OK) {                                                                            \
            std::fprintf(stderr, "%s: %s (%s)\n", #call, bbb_strerror(rc_), bbb_last_error_detail());   \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

static bool load_taps_file(const std::string &path, int *k, std::vector<uint16_t> *taps, std::vector<uint32_t> *off) {
    // packed tap lists: one row per line, space separated column indices (basebandboard_amd/data/*.taps)
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char line[8192];
    off->assign(1, 0);
    while (std::fgets(line, sizeof line, f)) {
        char *p = line, *e = nullptr;
        bool any = false;
        for (;;) {
            const long v = std::strtol(p, &e, 10);
            if (e == p) break;
            taps->push_back((uint16_t)v);
            p = e;
            any = true;
        }
        if (any) off->push_back((uint32_t)taps->size());
    }
    std::fclose(f);
    *k = (int)off->size() - 1;
    return *k > 0;
}

int main(int argc, char **argv) {
    std::string matrix;
    int k = 31, nv = 8, seeds = 1;
    unsigned long long init0 = 1;
    double bits = 1e9, from = 0, to = 10, step = 1, loopback = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--matrix") matrix = argv[i + 1];
        else if (a == "--k") k = std::atoi(argv[i + 1]);
        else if (a == "--init") init0 = std::strtoull(argv[i + 1], nullptr, 16);
        else if (a == "--seeds") seeds = std::atoi(argv[i + 1]);
        else if (a == "--nv") nv = std::atoi(argv[i + 1]);
        else if (a == "--bits") bits = std::atof(argv[i + 1]);
        else if (a == "--from") from = std::atof(argv[i + 1]);
        else if (a == "--to") to = std::atof(argv[i + 1]);
        else if (a == "--step") step = std::atof(argv[i + 1]);
        else if (a == "--loopback") loopback = std::atof(argv[i + 1]);
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (bbb_abi_version() != BBB_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }

    int n = 0;
    std::vector<uint16_t> taps;
    std::vector<uint32_t> off;
    if (matrix.empty()) {
        const char *here = std::getenv("BBB_DATA");
        const std::string path = std::string(here ? here : "basebandboard_amd/data") + "/lutopt_256.taps";
        if (!load_taps_file(path, &n, &taps, &off)) { std::fprintf(stderr, "cannot read %s (set BBB_DATA)\n", path.c_str()); return 1; }
    } else {
        uint16_t *t = nullptr;
        uint32_t *o = nullptr;
        CHECK(bbb_lutopt_load_matrix_file(matrix.c_str(), &n, &t, &o));
        taps.assign(t, t + o[n]);
        off.assign(o, o + n + 1);
        bbb_free(t);
        bbb_free(o);
    }
    if (seeds < 1 || init0 == 0) { std::fprintf(stderr, "--seeds >= 1 and a non-zero --init expected\n"); return 2; }
    // amplitude for an Eb/N0: sigma of the scaled CLT sample is 8 * nv (CLTGRNG variance 64), one sample per bit
    std::vector<bbb_trial_cfg> cfg;
    for (double db = from; db <= to + 1e-9; db += step) {
        bbb_trial_cfg c{};
        c.prbs_k = k;
        c.noise_var = nv;
        c.amp = (int)std::lround(8.0 * nv * std::sqrt(2.0 * std::pow(10.0, db / 10.0)));
        c.prbs_state = 1;
        c.warmup = 16;                                                  // 2 * log2(n): rng.py:161-162
        c.first_bit = 0;
        c.nbits = (uint64_t)bits;
        cfg.push_back(c);
    }
    std::vector<bbb_ber> out(cfg.size()), part(cfg.size());
    bbb_lutopt *h = nullptr;
    float ms = 0;
    for (int sd = 0; sd < seeds; sd++) {
        const uint64_t init[8] = {init0 + (uint64_t)sd, 0, 0, 0, 0, 0, 0, 0};     // reset value (gateware/bbb/rng.py:21)
        if (h) CHECK(bbb_lutopt_destroy(h));
        CHECK(bbb_lutopt_create(&h, n, taps.data(), off.data(), init, 0));
        CHECK(bbb_ber_trials(h, cfg.data(), (int)cfg.size(), part.data()));  // first call builds the jump plans
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, nullptr);
        CHECK(bbb_ber_trials(h, cfg.data(), (int)cfg.size(), part.data()));
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float one = 0;
        (void)hipEventElapsedTime(&one, e0, e1);
        ms += one;
        for (size_t i = 0; i < cfg.size(); i++) { out[i].bits += part[i].bits; out[i].errors += part[i].errors; }
    }
    std::printf("# PRBS-%d, noise_var %d, %.3g bits per point and seed, %d seed(s), %zu points in %.3f ms\n", k, nv, bits, seeds,
                cfg.size(), ms);
    std::printf("# EbN0_dB  amp  bits  errors  BER  Q(sqrt(2EbN0))\n");
    for (size_t i = 0; i < cfg.size(); i++) {
        const double ebn0 = (double)cfg[i].amp * cfg[i].amp / (2.0 * 64.0 * nv * nv);
        std::printf("%7.3f %4d %llu %llu %.4e %.4e\n", 10 * std::log10(ebn0), cfg[i].amp, (unsigned long long)out[i].bits,
                    (unsigned long long)out[i].errors, (double)out[i].errors / (double)out[i].bits, 0.5 * std::erfc(std::sqrt(ebn0)));
    }
    if (loopback > 0) {
        const uint64_t nb = (uint64_t)loopback;
        uint64_t *buf = nullptr;
        if (hipMalloc(&buf, ((nb + 63) / 64) * 8) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
        CHECK(bbb_prbs_fill(k, 1, 0, nb, buf, 0, nullptr));
        bbb_detector_stats st{};
        CHECK(bbb_prbs_detector_stream(k, buf, nb, nullptr, nullptr, &st, 0, 0, 0, nullptr));
        std::printf("# loopback: %llu bits, %llu errors, %llu resyncs, %llu reload clocks\n", (unsigned long long)st.bits,
                    (unsigned long long)st.errors, (unsigned long long)st.resyncs, (unsigned long long)st.reload_clocks);
        (void)hipFree(buf);
    }
    CHECK(bbb_lutopt_destroy(h));
    return 0;
}
