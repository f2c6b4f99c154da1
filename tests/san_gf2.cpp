// Sanitizer run of the product's host-side GF(2) arithmetic (tests/test_gf2.py::test_host_arithmetic_under_sanitizers):
// csrc/gf2poly.hpp, csrc/search_rng.hpp and csrc/gf2.hpp are plain C++ headers, built here with
// g++ -fsanitize=address,undefined (sanitizers on the host build only).
#define __host__
#define __device__
#include "../basebandboard_amd/csrc/gf2.hpp"
#include "../basebandboard_amd/csrc/gf2poly.hpp"
#include "../basebandboard_amd/csrc/search_rng.hpp"
#include "../basebandboard_amd/csrc/sweep_shard.hpp"

#include <cstdio>

using namespace bbb;

static int check_shards() {
    // bbb_sweep_shard's arithmetic on ragged and extreme sizes: slices contiguous, in rank order, covering exactly
    bbb_trial_cfg t[4] = {};
    const unsigned long long nb[4] = {1000000007ull, 3, 0, ~0ull - 5};
    const unsigned long long fb[4] = {0, 5, 9, 2};
    for (int i = 0; i < 4; i++) { t[i].prbs_k = 31; t[i].nbits = nb[i]; t[i].first_bit = fb[i]; t[i].prbs_state = 1; }
    for (int ndev : {1, 2, 3, 8, 64}) {
        unsigned long long pos[4], tot[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) pos[i] = fb[i];
        for (int r = 0; r < ndev; r++) {
            bbb_trial_cfg mine[4];
            if (sweep_shard(t, 4, ndev, r, BBB_SHARD_BITS, mine)) return 20;
            for (int i = 0; i < 4; i++) {
                if (mine[i].first_bit != pos[i]) return 21;
                pos[i] += mine[i].nbits;
            }
            if (sweep_shard(t, 4, ndev, r, BBB_SHARD_TRIALS, mine)) return 22;
            for (int i = 0; i < 4; i++) tot[i] += mine[i].nbits != 0;
        }
        for (int i = 0; i < 4; i++) {
            if (pos[i] != fb[i] + nb[i]) return 23;
            if (tot[i] != (nb[i] != 0)) return 24;
        }
    }
    bbb_trial_cfg o{};
    o.nbits = 100; o.first_bit = ~0ull - 10;
    bbb_trial_cfg m1;
    if (sweep_shard(&o, 1, 2, 1, BBB_SHARD_BITS, &m1) != -2) return 25;       // first_bit + lo overflows
    if (sweep_shard(&o, 1, 2, 2, BBB_SHARD_BITS, &m1) != -1) return 26;
    if (sweep_shard(&o, 1, 2, 0, 7, &m1) != -1) return 27;
    return 0;
}

int main() {
    if (int e = check_shards()) return e;
    // x^4 + x^3 + 1 primitive, x^4 + x^2 + x + 1 not (binary_polynomial.rs:352-360)
    GF2Poly p;
    p.set(4); p.set(3); p.set(0);
    if (gf2_is_primitive(p) != 1) return 1;
    GF2Poly q;
    q.set(4); q.set(2); q.set(1); q.set(0);
    if (gf2_is_primitive(q) != 0) return 2;
    // x^200 + x^5 + x^3 + x^2 + 1 primitive (:337-339, :366-369)
    GF2Poly r;
    r.set(200); r.set(5); r.set(3); r.set(2); r.set(0);
    if (gf2_is_primitive(r) != 1) return 3;
    // degree 512: the widest polynomial the layout holds
    GF2Poly big;
    big.set(512); big.set(8); big.set(5); big.set(2); big.set(0);
    (void)gf2_is_primitive(big);
    unsigned long long sum = 0;
    for (int k : {16, 64, 256, 512}) {
        for (unsigned long long cand = 0; cand < 3; cand++) {
            std::vector<uint16_t> t4;
            std::vector<uint8_t> w;
            search_candidate_host(k, 12345, cand + (k == 512 ? (1ull << 40) : 0), t4, w);
            std::vector<uint16_t> taps;
            std::vector<uint32_t> off(k + 1, 0);
            for (int row = 0; row < k; row++) {
                off[row] = (uint32_t)taps.size();
                for (int j = 0; j < w[row]; j++) taps.push_back(t4[(size_t)4 * row + j]);
            }
            off[k] = (uint32_t)taps.size();
            GF2Poly P;
            const int L = lutopt_charpoly(k, taps.data(), off.data(), P);
            sum += (unsigned long long)L + (L == k ? (unsigned long long)gf2_is_primitive(P) : 0ull);
        }
    }
    // jump-ahead matrices
    GF2Mat A(256);
    for (int i = 0; i < 256; i++) { A.set(i, (i + 1) % 256); A.set(i, (i * 7 + 3) % 256); }
    GF2Powers pw(A);
    uint64_t x[8] = {1, 2, 3, 4, 0, 0, 0, 0}, y[8];
    pw.apply(1000000007ull, x, y);
    sum += y[0] & 0xff;
    std::printf("ok %llu\n", sum);
    return 0;
}
