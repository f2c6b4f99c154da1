// sweep_shard.hpp -- which part of a sweep one of `ndev` devices runs (bbb_sweep_shard, bbb_ber_sweep_multi).
// Plain C++ (no HIP): also built for the host under the sanitizers (tests/san_gf2.cpp).
#pragma once
#include <cstdint>

#include "../../include/bbb.h"

namespace bbb {

// mine[i] = trial i as rank `rank` runs it (nbits = 0: not at all).  Returns 0, or a negative code:
// -1 bad arguments / mode, -2 first_bit + nbits overflows.
inline int sweep_shard(const bbb_trial_cfg *cfgs, int ncfg, int ndev, int rank, int mode, bbb_trial_cfg *mine) {
    if (ncfg < 0 || (ncfg && (!cfgs || !mine)) || ndev < 1 || rank < 0 || rank >= ndev) return -1;
    int group = -1;                                   // BBB_SHARD_GROUPS: index of the run of same-stream trials trial i lies in
    for (int i = 0; i < ncfg; i++) {
        bbb_trial_cfg c = cfgs[i];
        if (i == 0 || cfgs[i].prbs_k != cfgs[i - 1].prbs_k || cfgs[i].prbs_state != cfgs[i - 1].prbs_state ||
            cfgs[i].warmup != cfgs[i - 1].warmup || cfgs[i].first_bit != cfgs[i - 1].first_bit || cfgs[i].nbits != cfgs[i - 1].nbits)
            group++;
        switch (mode) {
        case BBB_SHARD_TRIALS:                        // trial i belongs to rank i % ndev
            if (i % ndev != rank) c.nbits = 0;
            break;
        case BBB_SHARD_SEEDS:                         // every rank runs every trial (on its own reset state)
            break;
        case BBB_SHARD_BITS: {                        // rank r takes bits [r*nbits/ndev, (r+1)*nbits/ndev) of every trial
            const unsigned __int128 nb = cfgs[i].nbits;
            const uint64_t lo = (uint64_t)(nb * (unsigned)rank / (unsigned)ndev);
            const uint64_t hi = (uint64_t)(nb * (unsigned)(rank + 1) / (unsigned)ndev);
            if (c.first_bit + lo < c.first_bit) return -2;
            c.first_bit += lo;
            c.nbits = hi - lo;
            break;
        }
        case BBB_SHARD_GROUPS:                        // a run of trials on one noise / PRBS stream stays on one rank
            if (group % ndev != rank) c.nbits = 0;
            break;
        default:
            return -1;
        }
        mine[i] = c;
    }
    return 0;
}

}  // namespace bbb
