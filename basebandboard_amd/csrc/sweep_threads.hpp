// sweep_threads.hpp -- the host orchestration of bbb_ber_sweep_multi: one host thread per device, each computing its share
// of the sweep (sweep_shard) and running it through `run(rank, share)`; errors travel back per rank.  Plain C++ (no HIP), so
// that the same code is built for the host under ThreadSanitizer with a stub in place of the kernel launches
// (tests/san_sweep.cpp); bbb_api.hip instantiates it with the real launch.  The reference's analogue is rnghunt's worker
// pool with one channel back (software/rnghunt/src/bin/rnghunt.rs:16-18,54-65).
//
// Round 5: the threads are PERSISTENT (rnghunt's workers live for the whole search too).  Rounds 2-4 created and joined a
// std::thread per device and call: 60-100 us per call on the GPU box's host, in front of the first launch of every device but
// the caller's -- a tenth of a sweep whose kernel takes 1.1 ms.  A pool of ndev - 1 workers now waits on a condition variable;
// rank 0 still runs on the calling thread.  The pool is created on first use, grows with the largest ndev seen, and is never
// destroyed (its threads are detached and idle between calls: nothing to join at exit, no static destructor that could
// run while a worker is still inside the HIP runtime).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sweep_shard.hpp"

namespace bbb {

class SweepWorkers {
public:
    static SweepWorkers &instance() {
        static SweepWorkers *p = new SweepWorkers;       // (leaked on purpose: see above)
        return *p;
    }
    // work(r) for r = 0 .. n - 1: r = 0 on the calling thread, the others on the pool's workers; returns when all are done.
    // Calls from several threads take turns.
    void run(int n, const std::function<void(int)> &work) {
        std::lock_guard<std::mutex> turn(call_mu_);
        {
            std::unique_lock<std::mutex> g(mu_);
            while ((int)nworkers_ < n - 1) {
                const int rank = ++nworkers_;
                std::thread(&SweepWorkers::loop, this, rank, gen_).detach();
            }
            job_ = &work;
            active_ = n;
            pending_ = n - 1;
            gen_++;
        }
        cv_go_.notify_all();
        work(0);
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int rank, unsigned long long seen) {
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_go_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (rank < active_) job = job_;
            }
            if (!job) continue;                          // (a call with fewer ranks than the pool has workers)
            (*job)(rank);
            {
                std::lock_guard<std::mutex> g(mu_);
                pending_--;
            }
            cv_done_.notify_all();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    unsigned long long gen_ = 0;
    int nworkers_ = 0, active_ = 0, pending_ = 0;
};

// run(rank, mine, &err_text) -> 0 or a negative BBB_E* code.  Returns the first failing rank's code (its text in *err, its rank
// in *bad_rank), or 0.  Rank 0 runs on the calling thread.
template <typename Run>
int run_shares_on_threads(const bbb_trial_cfg *cfgs, int ncfg, int ndev, int mode, Run run, std::string *err, int *bad_rank) {
    std::vector<int> rcs((size_t)ndev, 0);
    std::vector<std::string> errs((size_t)ndev);
    const std::function<void(int)> work = [&](int r) {
        std::vector<bbb_trial_cfg> mine((size_t)ncfg);
        const int e = sweep_shard(cfgs, ncfg, ndev, r, mode, mine.data());
        if (e) {
            rcs[(size_t)r] = BBB_EINVAL;
            errs[(size_t)r] = e == -2 ? "first_bit + nbits overflows" : "bad shard arguments";
            return;
        }
        rcs[(size_t)r] = run(r, mine.data(), &errs[(size_t)r]);
    };
    if (ndev == 1) work(0);
    else SweepWorkers::instance().run(ndev, work);
    for (int r = 0; r < ndev; r++)
        if (rcs[(size_t)r]) {
            if (err) *err = errs[(size_t)r];
            if (bad_rank) *bad_rank = r;
            return rcs[(size_t)r];
        }
    return 0;
}

}  // namespace bbb
