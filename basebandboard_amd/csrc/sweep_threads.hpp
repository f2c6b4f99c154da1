// sweep_threads.hpp -- the host orchestration of bbb_ber_sweep_multi: one host thread per device, each computing its share
// of the sweep (sweep_shard) and running it through `run(rank, share)`; errors travel back per rank.  Plain C++ (no HIP), so
// that the same code is built for the host under ThreadSanitizer with a stub in place of the kernel launches
// (tests/san_sweep.cpp); bbb_api.hip instantiates it with the real launch.  The reference's analogue is rnghunt's worker
// pool with one channel back (software/rnghunt/src/bin/rnghunt.rs:16-18,54-65).
#pragma once
#include <string>
#include <thread>
#include <vector>

#include "sweep_shard.hpp"

namespace bbb {

// run(rank, mine, &err_text) -> 0 or a negative BBB_E* code.  Returns the first failing rank's code (its text in *err, its rank
// in *bad_rank), or 0.  Rank 0 runs on the calling thread.
template <typename Run>
int run_shares_on_threads(const bbb_trial_cfg *cfgs, int ncfg, int ndev, int mode, Run run, std::string *err, int *bad_rank) {
    std::vector<int> rcs((size_t)ndev, 0);
    std::vector<std::string> errs((size_t)ndev);
    auto work = [&](int r) {
        std::vector<bbb_trial_cfg> mine((size_t)ncfg);
        const int e = sweep_shard(cfgs, ncfg, ndev, r, mode, mine.data());
        if (e) {
            rcs[(size_t)r] = BBB_EINVAL;
            errs[(size_t)r] = e == -2 ? "first_bit + nbits overflows" : "bad shard arguments";
            return;
        }
        rcs[(size_t)r] = run(r, mine.data(), &errs[(size_t)r]);
    };
    {
        std::vector<std::thread> th;
        for (int r = 1; r < ndev; r++) th.emplace_back(work, r);
        work(0);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < ndev; r++)
        if (rcs[(size_t)r]) {
            if (err) *err = errs[(size_t)r];
            if (bad_rank) *bad_rank = r;
            return rcs[(size_t)r];
        }
    return 0;
}

}  // namespace bbb
