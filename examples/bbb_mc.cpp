// bbb_mc -- a plain C++ caller of the C ABI (include/bbb.h): the Monte-Carlo loops of BASELINE.json configs[1..4].
// No torch, no Python: the library, hipMalloc'd buffers, std::thread and printf.
// Build: make -C examples   (hipcc, links ../basebandboard_amd/libbbb_hip.so)
//
//   bbb_mc [--matrix FILE] [--init HEX] [--gpus N] [--json 1]
//          BER sweep:   [--prbs 31] [--bits 1e9] [--nv 8] [--ebn0 A:B:STEP] [--seeds N] [--shard bits|seeds|trials|groups]
//          AWGN fill:   --nsamples 1e9 [--steps 5] [--staged 0|1|m]   the sample stream drained through bbb_awgn_stream_next;
//                       --staged picks the level of bbb_lutopt_set_staged for it (default: the stream's own choice, two reads
//                       per sample kernel; 0 = plain bbb_awgn_fill_i8 calls in the one-kernel form)
//          loopback:    --loopback BITS
//          search:      --search K [--seed S] [--count N] --out FILE    the reference's rnghunt (software/rnghunt/src/bin/
//                       rnghunt.rs:13-66) on the GPU: candidates of `seed` are examined in windows of N (default 65536)
//                       until one has period 2^K - 1; it is written to FILE in the reference's `out` format (K lines of K
//                       characters 0/1), read back and re-checked (bbb_lutopt_is_full_period)
//
// --matrix   the reference's 0/1 text format (software/rnghunt/matrices/256); default: the shipped n256 matrix
//            (gateware/bbb/rng_recurrences.py:172-259, used by tx.py:70).
// --init     the generator's reset state (hex, default 1: gateware/bbb/rng.py:21).
// --ebn0     A:B:STEP in dB (also --from/--to/--step); amplitude per point for sigma = 8 noise_var, one sample per bit.
// --gpus N   BER sweep: bbb_ber_sweep_multi over devices 0..N-1 -- one host thread per device and ONE RCCL
//            all-reduce of the uint64 counters.  --shard bits (default): every device runs every point over its
//            slice of the bit range, the counters equal the 1-GPU counters exactly; seeds: device d runs every
//            point on its own seed (below), counters summed (points x seeds, BASELINE configs[4]); trials: point i
//            on device i % N; groups (with --seeds S): the sweep once per seed as ONE call of S x points trials, a seed's
//            sweep on one device (group q on device q % N: configs[4] as eight sweeps on one device, one each on eight).  AWGN fill: device d reads stream positions [16 + 2^48 d + s n, +n) in step s -- its own
//            contiguous stretch of the one sequential stream, no collective.
// --multi 1  take the bbb_ber_sweep_multi route (RCCL) even with --gpus 1.
// --seeds N  (1 GPU) repeat the sweep on N seeds and sum the counters.  Seed d = the reset state `init` advanced 2^48 d
//            clocks (GF(2) jump-ahead, bbb_lutopt_state_at): disjoint stretches of the generator's one cycle.  (Reset
//            states that differ by small integers are XOR-dependent streams, and nothing would keep them from overlapping.)
// --json 1   one JSON object per line (points, then a summary with rates and the roofline fraction) instead of
//            the table.
#include "../include/bbb.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != BBB_OK) {                                                                            \
            std::fprintf(stderr, "%s: %s (%s)\n", #call, bbb_strerror(rc_), bbb_last_error_detail());   \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

static const double kHbmPeakGBs = 8000.0;     // MI355X HBM3E, nominal

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

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Matrix {
    int n = 0;
    std::vector<uint16_t> taps;
    std::vector<uint32_t> off;
};

// AWGN fill on one device: `steps` consecutive fills of n samples; device d reads its own stretch of the stream, 2^48 d steps in
struct FillResult { int rc = 0; std::string err; double kernel_ms = 0, seed_ms = 0, wall_s = 0; uint64_t launches = 0; std::vector<int8_t> head; };
static void fill_worker(const Matrix &m, unsigned long long init0, int dev, int ndev, uint64_t n, int steps, int staged, FillResult *res) {
    auto body = [&]() -> int {
        if (hipSetDevice(dev) != hipSuccess) { res->err = "hipSetDevice failed"; return 1; }
        const uint64_t init[8] = {init0, 0, 0, 0, 0, 0, 0, 0};
        bbb_lutopt *h = nullptr;
        int rc = bbb_lutopt_create(&h, m.n, m.taps.data(), m.off.data(), init, dev);
        if (rc) { res->err = bbb_last_error_detail(); return rc; }
        int8_t *buf = nullptr;
        if (hipMalloc((void **)&buf, (n + 15) / 16 * 16) != hipSuccess) { res->err = "hipMalloc failed"; return 1; }
        (void)ndev;
        const uint64_t first0 = (uint64_t)16 + ((uint64_t)dev << 48);
        auto first = [&](int s) { return first0 + (uint64_t)s * n; };
        // the sample stream as an object: the library owns staging, the two-kernel form and the announcement of every next read
        bbb_awgn_stream *st = nullptr;
        if (staged > 0) (void)bbb_lutopt_set_staged(h, staged);             // an explicit level; otherwise the stream picks its own
        if (staged != 0) {
            if ((rc = bbb_awgn_stream_open(h, n, first0, 1, &st))) { res->err = bbb_last_error_detail(); return rc; }
            if ((rc = bbb_awgn_stream_next(st, buf))) { res->err = bbb_last_error_detail(); return rc; }       // builds the jump plan
        } else if ((rc = bbb_awgn_fill_i8(h, buf, n, first(0)))) { res->err = bbb_last_error_detail(); return rc; }
        res->head.resize(n < 64 ? n : 64);
        (void)hipMemcpy(res->head.data(), buf, res->head.size(), hipMemcpyDeviceToHost);
        if (st) (void)bbb_awgn_stream_seek(st, first(1));      // (drops what a sample kernel produced ahead: the timed steps start on a launch)
        (void)bbb_lutopt_profile(h, 1);
        (void)hipDeviceSynchronize();
        const double t0 = now_s();
        for (int s = 1; s <= steps; s++) {
            if (st) rc = bbb_awgn_stream_next(st, buf);
            else {
                rc = bbb_awgn_fill_i8(h, buf, n, first(s));
                if (!rc) (void)bbb_awgn_prefetch(h, n, first(s + 1));       // the next step's seeding runs beside this step's kernel
            }
            if (rc) { res->err = bbb_last_error_detail(); return rc; }
        }
        (void)hipDeviceSynchronize();
        res->wall_s = now_s() - t0;
        uint64_t calls = 0;
        (void)bbb_lutopt_profile_read(h, &res->seed_ms, &res->kernel_ms, &calls, 1);
        if (calls) { res->kernel_ms /= (double)calls; res->seed_ms /= (double)calls; }
        res->launches = calls;
        if (st) (void)bbb_awgn_stream_close(st);
        (void)hipFree(buf);
        (void)bbb_lutopt_destroy(h);
        return 0;
    };
    res->rc = body();
}

int main(int argc, char **argv) {
    std::string matrix, shard = "bits";
    int k = 31, nv = 8, seeds = 1, gpus = 1, json = 0, steps = 5, multi = 0, staged = -1, search_k = 0;
    unsigned long long search_seed = 1, search_count = 65536;
    std::string outfile;
    unsigned long long init0 = 1;
    double bits = 1e9, from = 0, to = 10, step = 1, loopback = 0, nsamples = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const char *v = argv[i + 1];
        if (a == "--matrix") matrix = v;
        else if (a == "--k" || a == "--prbs") k = std::atoi(v);
        else if (a == "--init") init0 = std::strtoull(v, nullptr, 16);
        else if (a == "--seeds") seeds = std::atoi(v);
        else if (a == "--nv") nv = std::atoi(v);
        else if (a == "--bits") bits = std::atof(v);
        else if (a == "--from") from = std::atof(v);
        else if (a == "--to") to = std::atof(v);
        else if (a == "--step") step = std::atof(v);
        else if (a == "--ebn0") {
            if (std::sscanf(v, "%lf:%lf:%lf", &from, &to, &step) != 3) { std::fprintf(stderr, "--ebn0 A:B:STEP\n"); return 2; }
        }
        else if (a == "--loopback") loopback = std::atof(v);
        else if (a == "--nsamples") nsamples = std::atof(v);
        else if (a == "--steps") steps = std::atoi(v);
        else if (a == "--gpus") gpus = std::atoi(v);
        else if (a == "--shard") shard = v;
        else if (a == "--json") json = std::atoi(v);
        else if (a == "--multi") multi = std::atoi(v);
        else if (a == "--staged") staged = std::atoi(v);
        else if (a == "--search") search_k = std::atoi(v);
        else if (a == "--seed") search_seed = std::strtoull(v, nullptr, 0);
        else if (a == "--count") search_count = std::strtoull(v, nullptr, 0);
        else if (a == "--out") outfile = v;
        else if (a == "--gen") { if (std::string(v) != "lutopt") { std::fprintf(stderr, "--gen lutopt is the only generator the reference has\n"); return 2; } }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (argc % 2 == 0) { std::fprintf(stderr, "every option takes a value\n"); return 2; }
    if (bbb_abi_version() != BBB_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    int mode = BBB_SHARD_BITS;
    if (shard == "seeds") mode = BBB_SHARD_SEEDS;
    else if (shard == "trials") mode = BBB_SHARD_TRIALS;
    else if (shard == "groups") mode = BBB_SHARD_GROUPS;
    else if (shard != "bits") { std::fprintf(stderr, "--shard bits|seeds|trials|groups\n"); return 2; }
    if (seeds < 1 || init0 == 0 || gpus < 1 || step <= 0 || steps < 1) { std::fprintf(stderr, "bad --seeds / --init / --gpus / --step / --steps\n"); return 2; }
    int ndev_seen = 0;
    CHECK(bbb_device_count(&ndev_seen));
    if (gpus > ndev_seen) { std::fprintf(stderr, "--gpus %d but %d device(s) visible\n", gpus, ndev_seen); return 1; }

    // ---- matrix search (software/rnghunt/src/bin/rnghunt.rs:13-66) -------------------------------------------------------
    if (search_k > 0) {
        if (outfile.empty()) { std::fprintf(stderr, "--search K needs --out FILE\n"); return 2; }
        std::vector<uint16_t> taps((size_t)search_k * 4);
        std::vector<uint32_t> off((size_t)search_k + 1);
        uint64_t found = ~0ull, first = 0, tested = 0, full = 0;
        double kernel_ms = 0;
        const double t0 = now_s();
        while (found == ~0ull) {                       // the reference's workers loop until one reports a hit (rnghunt.rs:23-57)
            bbb_search_stats st{};
            CHECK(bbb_lutopt_search(search_k, search_seed, first, search_count, &found, taps.data(), off.data(), &st, 0, nullptr));
            tested += st.tested; full += st.full_degree; kernel_ms += (double)st.kernel_ns * 1e-6;
            first += search_count;
            if (first > (1ull << 40)) { std::fprintf(stderr, "no matrix among 2^40 candidates\n"); return 1; }
        }
        CHECK(bbb_lutopt_save_matrix_file(outfile.c_str(), search_k, taps.data(), off.data()));      // rnghunt.rs:51-53
        int k2 = 0, ok = 0;
        uint16_t *t2 = nullptr;
        uint32_t *o2 = nullptr;
        CHECK(bbb_lutopt_load_matrix_file(outfile.c_str(), &k2, &t2, &o2));                              // what --matrix reads
        CHECK(bbb_lutopt_is_full_period(k2, t2, o2, &ok));
        bbb_free(t2);
        bbb_free(o2);
        std::printf("{\"mode\": \"search\", \"k\": %d, \"seed\": %llu, \"candidate\": %llu, \"tested\": %llu, \"full_degree\": %llu, "
                    "\"kernel_ms\": %.3f, \"seconds\": %.3f, \"out\": \"%s\", \"reloaded_k\": %d, \"full_period\": %s}\n",
                    search_k, search_seed, (unsigned long long)found, (unsigned long long)tested, (unsigned long long)full, kernel_ms, now_s() - t0,
                    outfile.c_str(), k2, ok ? "true" : "false");
        return ok && k2 == search_k ? 0 : 1;
    }

    Matrix m;
    if (matrix.empty()) {
        const char *here = std::getenv("BBB_DATA");
        const std::string path = std::string(here ? here : "basebandboard_amd/data") + "/lutopt_256.taps";
        if (!load_taps_file(path, &m.n, &m.taps, &m.off)) { std::fprintf(stderr, "cannot read %s (set BBB_DATA)\n", path.c_str()); return 1; }
    } else {
        uint16_t *t = nullptr;
        uint32_t *o = nullptr;
        CHECK(bbb_lutopt_load_matrix_file(matrix.c_str(), &m.n, &t, &o));
        m.taps.assign(t, t + o[m.n]);
        m.off.assign(o, o + m.n + 1);
        bbb_free(t);
        bbb_free(o);
    }

    // seed d of the run = the reset state advanced 2^48 d clocks (host-side GF(2) jump-ahead on a device -1 handle)
    auto seed_words = [&](int d, uint64_t (&w)[8]) -> int {
        const uint64_t base[8] = {init0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; i++) w[i] = base[i];
        if (d == 0) return BBB_OK;
        bbb_lutopt *hh = nullptr;
        int rc = bbb_lutopt_create(&hh, m.n, m.taps.data(), m.off.data(), base, -1);
        if (rc) return rc;
        rc = bbb_lutopt_state_at(hh, (uint64_t)d << 48, w);
        (void)bbb_lutopt_destroy(hh);
        return rc;
    };

    // ---- AWGN fill mode (BASELINE configs[1]) -----------------------------------------------------------------------
    if (nsamples > 0) {
        const uint64_t n = (uint64_t)nsamples;
        std::vector<FillResult> res((size_t)gpus);
        std::vector<std::thread> th;
        for (int d = 1; d < gpus; d++) th.emplace_back(fill_worker, std::cref(m), init0, d, gpus, n, steps, staged, &res[(size_t)d]);
        fill_worker(m, init0, 0, gpus, n, steps, staged, &res[0]);
        for (auto &t : th) t.join();
        double wall = 0, kms = 0;
        for (int d = 0; d < gpus; d++) {
            if (res[(size_t)d].rc) { std::fprintf(stderr, "device %d: %s\n", d, res[(size_t)d].err.c_str()); return 1; }
            wall = std::fmax(wall, res[(size_t)d].wall_s);
            kms = std::fmax(kms, res[(size_t)d].kernel_ms);
        }
        const double gs = (double)gpus * steps * (double)n / wall / 1e9;
        // what one launch of the sample kernel produces (bbb.h, bbb_lutopt_set_staged: look-ahead applies to fills of at
        // least 2^24 samples, a multiple of 16)
        const uint64_t per_launch = res[0].launches ? (uint64_t)steps * n / res[0].launches : n;
        const double kernel_gbs = kms > 0 ? (double)per_launch / (kms * 1e-3) / 1e9 : 0;        // 1 B per sample
        if (json) {
            std::printf("{\"mode\": \"awgn_fill\", \"n_gpus\": %d, \"samples_per_step_per_gpu\": %llu, \"steps\": %d, \"gsample_s\": %.3f, "
                        "\"kernel_ms_avg\": %.4f, \"samples_per_launch\": %llu, \"seed_ms_avg\": %.4f, \"hbm_write_gb_s_per_gpu\": %.1f, \"hbm_roofline_frac\": %.4f, "
                        "\"collective\": \"none (independent shards of one sequential stream)\", \"head\": [",
                        gpus, (unsigned long long)n, steps, gs, kms, (unsigned long long)per_launch, res[0].seed_ms, kernel_gbs, kernel_gbs / kHbmPeakGBs);
            for (size_t i = 0; i < res[0].head.size(); i++) std::printf("%s%d", i ? ", " : "", (int)res[0].head[i]);
            std::printf("]}\n");
        } else {
            std::printf("# AWGN fill: %d GPU(s) x %d steps x %llu samples: %.3f Gsample/s, sample kernel %.4f ms = %.1f GB/s = %.4f of the HBM peak\n",
                        gpus, steps, (unsigned long long)n, gs, kms, kernel_gbs, kernel_gbs / kHbmPeakGBs);
            std::printf("# head:");
            for (size_t i = 0; i < res[0].head.size(); i++) std::printf(" %d", (int)res[0].head[i]);
            std::printf("\n");
        }
        return 0;
    }

    // ---- BER sweep (BASELINE configs[3], [4]) -----------------------------------------------------------------------
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
    // --shard groups (BASELINE configs[4] as ONE call): the sweep once per seed -- the seeds as stretches of the one cycle 2^48
    // clocks apart, all on the same reset state -- as --seeds groups of consecutive trials; a group stays on one device
    // (bbb_ber_sweep_multi, BBB_SHARD_GROUPS: group q on device q mod N), the rows below are the sums over the seeds
    const size_t npoints = cfg.size();
    if (mode == BBB_SHARD_GROUPS) {
        for (int sd = 1; sd < seeds; sd++)
            for (size_t i = 0; i < npoints; i++) {
                bbb_trial_cfg c = cfg[i];
                c.warmup = 16 + ((uint64_t)sd << 48);
                cfg.push_back(c);
            }
    }
    const int group_seeds = seeds;
    std::vector<bbb_ber> out(cfg.size()), part(cfg.size());
    double ms = 0;
    const char *reduce = "single device";
    bbb_multi_info minfo{};
    int equals_single = -1;              // -1: not compared (single device, or one seed per device)
    if (gpus > 1 || multi) {
        // one handle per device; seeds mode gives device d seed d
        std::vector<bbb_lutopt *> hs((size_t)gpus, nullptr);
        for (int d = 0; d < gpus; d++) {
            uint64_t init[8];
            CHECK(seed_words(mode == BBB_SHARD_SEEDS ? d : 0, init));
            CHECK(bbb_lutopt_create(&hs[(size_t)d], m.n, m.taps.data(), m.off.data(), init, d));
        }
        CHECK(bbb_ber_sweep_multi(hs.data(), gpus, cfg.data(), (int)cfg.size(), mode, out.data()));   // plans + communicators
        const double t0 = now_s();
        CHECK(bbb_ber_sweep_multi(hs.data(), gpus, cfg.data(), (int)cfg.size(), mode, out.data()));
        ms = (now_s() - t0) * 1e3;
        // a multi-device run checks itself: what the communicator says, which RCCL it was, and -- for the sharding modes whose
        // totals do not depend on the device count -- the same trials on device 0 alone
        CHECK(bbb_multi_last_info(&minfo));
        if (mode != BBB_SHARD_SEEDS) {
            std::vector<bbb_ber> single(cfg.size());
            CHECK(bbb_ber_trials(hs[0], cfg.data(), (int)cfg.size(), single.data()));
            equals_single = 1;
            for (size_t i = 0; i < cfg.size(); i++)
                if (single[i].bits != out[i].bits || single[i].errors != out[i].errors) equals_single = 0;
        }
        for (auto *h : hs) CHECK(bbb_lutopt_destroy(h));
        CHECK(bbb_multi_release());
        reduce = "ncclAllReduce(uint64[2 x points], sum) over the devices of this process";
        seeds = mode == BBB_SHARD_SEEDS ? gpus : (mode == BBB_SHARD_GROUPS ? group_seeds : 1);
        if (mode == BBB_SHARD_GROUPS) {                                  // fold the seeds' rows into the points'
            for (int sd = 1; sd < group_seeds; sd++)
                for (size_t i = 0; i < npoints; i++) { out[i].bits += out[(size_t)sd * npoints + i].bits; out[i].errors += out[(size_t)sd * npoints + i].errors; }
            out.resize(npoints);
            cfg.resize(npoints);
        }
    } else if (mode == BBB_SHARD_GROUPS) {
        std::fprintf(stderr, "--shard groups runs through bbb_ber_sweep_multi: add --multi 1 (one device) or --gpus N\n");
        return 2;
    } else {
        bbb_lutopt *h = nullptr;
        for (int sd = 0; sd < seeds; sd++) {
            uint64_t init[8];                                                      // reset value (gateware/bbb/rng.py:21), jumped
            CHECK(seed_words(sd, init));
            if (h) CHECK(bbb_lutopt_destroy(h));
            CHECK(bbb_lutopt_create(&h, m.n, m.taps.data(), m.off.data(), init, 0));
            CHECK(bbb_ber_trials(h, cfg.data(), (int)cfg.size(), part.data()));  // first call builds the jump plans
            const double t0 = now_s();
            CHECK(bbb_ber_trials(h, cfg.data(), (int)cfg.size(), part.data()));
            ms += (now_s() - t0) * 1e3;
            for (size_t i = 0; i < cfg.size(); i++) { out[i].bits += part[i].bits; out[i].errors += part[i].errors; }
        }
        CHECK(bbb_lutopt_destroy(h));
    }
    unsigned long long total_bits = 0;
    for (auto &o : out) total_bits += o.bits;
    if (!json) {
        std::printf("# PRBS-%d, noise_var %d, %.3g bits per point and seed, %d seed(s), %d GPU(s), %zu points in %.3f ms\n", k, nv, bits, seeds,
                    gpus, cfg.size(), ms);
        std::printf("# EbN0_dB  amp  bits  errors  BER  Q(sqrt(2EbN0))\n");
    }
    for (size_t i = 0; i < cfg.size(); i++) {
        const double ebn0 = (double)cfg[i].amp * cfg[i].amp / (2.0 * 64.0 * nv * nv);
        // the slicer threshold falls on the integer lattice of the sigma = 8 sample: an error needs
        // |g| >= ceil(amp / nv), which is what a continuous Gaussian would see at this effective Eb/N0
        const double thr = std::ceil((double)cfg[i].amp / nv) - 0.5;
        const double ebn0_eff = thr * thr / (2.0 * 64.0);
        const double ber = out[i].bits ? (double)out[i].errors / (double)out[i].bits : 0.0;
        if (json)
            std::printf("{\"ebn0_db\": %.3f, \"ebn0_db_effective\": %.3f, \"amp\": %d, \"noise_var\": %d, \"bits\": %llu, \"errors\": %llu, "
                        "\"ber\": %.6e, \"q_theory\": %.6e, \"q_theory_effective\": %.6e}\n",
                        10 * std::log10(ebn0), 10 * std::log10(ebn0_eff), cfg[i].amp, nv, (unsigned long long)out[i].bits,
                        (unsigned long long)out[i].errors, ber, 0.5 * std::erfc(std::sqrt(ebn0)), 0.5 * std::erfc(std::sqrt(ebn0_eff)));
        else
            std::printf("%7.3f %4d %llu %llu %.4e %.4e\n", 10 * std::log10(ebn0), cfg[i].amp, (unsigned long long)out[i].bits,
                        (unsigned long long)out[i].errors, ber, 0.5 * std::erfc(std::sqrt(ebn0)));
    }
    if (json)
        std::printf("{\"mode\": \"ber_sweep\", \"prbs_k\": %d, \"points\": %zu, \"n_gpus\": %d, \"shard\": \"%s\", \"seeds\": %d, \"total_bits\": %llu, "
                    "\"ms\": %.3f, \"gbit_trials_s\": %.2f, \"hbm_bytes_per_bit\": 0, \"hbm_roofline_frac\": null, "
                    "\"bound\": \"integer VALU (no sample stream is written)\", \"reduce\": \"%s\", \"n_ranks_seen\": %d, "
                    "\"rccl_path\": \"%s\", \"rccl_reused\": %s, \"equals_single_device_counters\": %s}\n",
                    k, cfg.size(), gpus, shard.c_str(), seeds, total_bits, ms, ms > 0 ? (double)total_bits / (ms * 1e-3) / 1e9 : 0.0, reduce,
                    minfo.n_ranks_seen, minfo.rccl_path, minfo.rccl_reused ? "true" : "false",
                    equals_single < 0 ? "null" : (equals_single ? "true" : "false"));
    if (loopback > 0) {
        const uint64_t nb = (uint64_t)loopback;
        uint64_t *buf = nullptr;
        (void)hipSetDevice(0);
        if (hipMalloc(&buf, ((nb + 63) / 64) * 8) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
        CHECK(bbb_prbs_fill(k, 1, 0, nb, buf, 0, nullptr));
        uint64_t nerr = 0;
        CHECK(bbb_prbs_check(k, 1, 0, nb, buf, &nerr, 0, nullptr));         // warm
        hipEvent_t e0, e1, e2;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
        (void)hipEventRecord(e0, nullptr);
        CHECK(bbb_prbs_fill(k, 1, 0, nb, buf, 0, nullptr));
        (void)hipEventRecord(e1, nullptr);
        CHECK(bbb_prbs_check(k, 1, 0, nb, buf, &nerr, 0, nullptr));
        (void)hipEventRecord(e2, nullptr);
        (void)hipEventSynchronize(e2);
        float fill_ms = 0, chk_ms = 0;
        (void)hipEventElapsedTime(&fill_ms, e0, e1);
        (void)hipEventElapsedTime(&chk_ms, e1, e2);
        bbb_detector_stats st{};
        CHECK(bbb_prbs_detector_stream(k, buf, nb, nullptr, nullptr, &st, 0, 0, 0, nullptr));
        const double fgb = (double)nb / 8 / (fill_ms * 1e-3) / 1e9, cgb = (double)nb / 8 / (chk_ms * 1e-3) / 1e9;
        if (json)
            std::printf("{\"mode\": \"prbs_loopback\", \"prbs_k\": %d, \"bits\": %llu, \"check_errors\": %llu, \"detector_errors\": %llu, "
                        "\"resyncs\": %llu, \"reload_clocks\": %llu, \"fill_gb_s\": %.1f, \"check_gb_s\": %.1f, "
                        "\"fill_hbm_frac\": %.4f, \"check_hbm_frac\": %.4f}\n",
                        k, (unsigned long long)st.bits, (unsigned long long)nerr, (unsigned long long)st.errors,
                        (unsigned long long)st.resyncs, (unsigned long long)st.reload_clocks, fgb, cgb, fgb / kHbmPeakGBs, cgb / kHbmPeakGBs);
        else
            std::printf("# loopback: %llu bits, %llu errors, %llu resyncs, %llu reload clocks\n", (unsigned long long)st.bits,
                        (unsigned long long)st.errors, (unsigned long long)st.resyncs, (unsigned long long)st.reload_clocks);
        (void)hipFree(buf);
    }
    return 0;
}
