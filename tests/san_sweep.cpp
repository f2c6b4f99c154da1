// Host build of the multi-device sweep's orchestration (csrc/sweep_threads.hpp + csrc/sweep_shard.hpp) under ThreadSanitizer:
// one host thread per "device", each computing its share and running it through a STUB in place of the kernel launches (a
// deterministic function of the share's bit range, accumulated into that rank's own counters); the ranks' counters are then
// summed as the all-reduce would.  Checked: no data race, and the totals equal the undivided trials in all three modes for
// 1, 2, 3, 8 ranks; an overflowing share fails the launch with its rank named.  Test infrastructure only.
#include "../basebandboard_amd/csrc/sweep_threads.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

static unsigned long long fake_errors(const bbb_trial_cfg &c) {
    // additive over bit ranges: sum of a hash over the trial's bit positions (closed form: count of positions = 3 mod 7)
    auto upto = [](unsigned long long x) { return x / 7 + (x % 7 > 3 ? 1 : 0); };
    return upto(c.first_bit + c.nbits) - upto(c.first_bit);
}

int main() {
    std::vector<bbb_trial_cfg> cfgs;
    for (int i = 0; i < 11; i++) {
        bbb_trial_cfg c{};
        c.prbs_k = 31; c.amp = 100 + i; c.noise_var = 8; c.prbs_state = 1; c.warmup = 16;
        c.first_bit = 1000ull * i; c.nbits = 1000003ull + 77777ull * i;
        cfgs.push_back(c);
    }
    const int ncfg = (int)cfgs.size();
    // (BBB_SHARD_GROUPS: every trial of this list has its own bit range, so every trial is its own group)
    for (int mode : {BBB_SHARD_TRIALS, BBB_SHARD_SEEDS, BBB_SHARD_BITS, BBB_SHARD_GROUPS}) {
        for (int ndev : {1, 2, 3, 8}) {
            std::vector<std::vector<unsigned long long>> counters((size_t)ndev, std::vector<unsigned long long>(2 * (size_t)ncfg, 0));
            std::string err;
            int bad = -1;
            const int rc = bbb::run_shares_on_threads(cfgs.data(), ncfg, ndev, mode, [&](int r, const bbb_trial_cfg *mine, std::string *) -> int {
                for (int i = 0; i < ncfg; i++) {              // the "kernel": this rank's own buffer only
                    counters[(size_t)r][2 * (size_t)i] += mine[i].nbits;
                    counters[(size_t)r][2 * (size_t)i + 1] += fake_errors(mine[i]);
                }
                return 0;
            }, &err, &bad);
            if (rc) { std::printf("FAIL rc %d\n", rc); return 1; }
            for (int i = 0; i < ncfg; i++) {
                unsigned long long bits = 0, errs = 0;
                for (int r = 0; r < ndev; r++) { bits += counters[(size_t)r][2 * (size_t)i]; errs += counters[(size_t)r][2 * (size_t)i + 1]; }
                const unsigned long long mult = mode == BBB_SHARD_SEEDS ? (unsigned long long)ndev : 1ull;
                if (bits != mult * cfgs[(size_t)i].nbits || errs != mult * fake_errors(cfgs[(size_t)i])) {
                    std::printf("FAIL mode %d ndev %d trial %d\n", mode, ndev, i);
                    return 1;
                }
            }
        }
    }
    // BASELINE configs[4]: 8 seeds x 11 points as eight groups of eleven same-stream trials -- a group stays on one rank, the groups
    // go round robin, and the pool's persistent workers serve calls of changing width (8, 3, 8, 1 ranks) one after the other
    {
        std::vector<bbb_trial_cfg> g88;
        for (int s = 0; s < 8; s++)
            for (int i = 0; i < 11; i++) {
                bbb_trial_cfg c{};
                c.prbs_k = 31; c.amp = 90 + i; c.noise_var = 8; c.prbs_state = 1; c.warmup = 16 + ((unsigned long long)s << 48);
                c.first_bit = 0; c.nbits = 1000003ull;
                g88.push_back(c);
            }
        for (int ndev : {8, 3, 8, 1}) {
            std::vector<std::vector<int>> ran((size_t)ndev);
            std::string err;
            int bad = -1;
            const int rc = bbb::run_shares_on_threads(g88.data(), 88, ndev, BBB_SHARD_GROUPS, [&](int r, const bbb_trial_cfg *mine, std::string *) -> int {
                for (int i = 0; i < 88; i++)
                    if (mine[i].nbits) ran[(size_t)r].push_back(i);
                return 0;
            }, &err, &bad);
            if (rc) { std::printf("FAIL groups rc %d\n", rc); return 1; }
            std::vector<int> owner(88, -1);
            for (int r = 0; r < ndev; r++)
                for (int i : ran[(size_t)r]) {
                    if (owner[(size_t)i] != -1 || (i / 11) % ndev != r) { std::printf("FAIL groups: trial %d on rank %d\n", i, r); return 1; }
                    owner[(size_t)i] = r;
                }
            for (int i = 0; i < 88; i++)
                if (owner[(size_t)i] < 0) { std::printf("FAIL groups: trial %d not run\n", i); return 1; }
        }
    }
    // a failing rank: its code and text come back, the other threads are joined
    cfgs[3].first_bit = ~0ull - 5;
    std::string err;
    int bad = -1;
    const int rc = bbb::run_shares_on_threads(cfgs.data(), ncfg, 4, BBB_SHARD_BITS, [&](int, const bbb_trial_cfg *, std::string *) { return 0; }, &err, &bad);
    if (rc != BBB_EINVAL || bad < 0 || err.empty()) { std::printf("FAIL overflow not reported\n"); return 1; }
    std::printf("ok tsan sweep\n");
    return 0;
}
