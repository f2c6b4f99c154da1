// search_rng.hpp -- the candidate matrices of the recurrence search, shared by host and device.
//
// The reference draws them with rand::thread_rng (software/rnghunt/src/binary_matrix.rs:81-101), an
// unseeded generator: which matrices it tries is not reproducible, only HOW they are made is
// specified -- row weights drawn uniformly from [3,4,4,4,4,4,4,4] (src/bin/rnghunt.rs:25) and
// columns picked lowest-weight-first, uniformly among the lightest columns not yet used by the row,
// so that all column weights stay within one of each other.  Picking "uniformly among the columns
// of minimum weight" until every column has been picked once is a uniformly random permutation of
// the columns; the tap stream of a matrix is therefore a concatenation of random permutations, cut
// into rows of 3 or 4.  That is what is built here, from a counter-based hash so that candidate
// number c of seed s is the same matrix on the host, on the device and on every GPU:
//
//   w_r        = 3 if hash(s, c, r) % 8 == 0 else 4
//   perm_j     = columns sorted by key_j(col) = (hash(s, c, 0x10000 + j*k + col) & ~1023) | col,  j = 0, 1, ...
//                (22 random bits, ties broken by the column number carried in the low bits)
//   stream     = perm_0 ++ perm_1 ++ ...;  row r takes positions [P_r, P_r + w_r), P = prefix sums of w
//   a row that straddles two permutations must not use a column twice: each of its entries in the
//   later permutation that repeats one from the earlier part is swapped with the first entry behind
//   the row's part (h, h+1, ...) that does not.
#pragma once
#include <cstdint>
#include <algorithm>
#include <vector>

namespace bbb {

__host__ __device__ inline uint64_t search_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint32_t search_hash(uint64_t seed, uint64_t cand, uint32_t ctr) {
    return (uint32_t)(search_mix64(search_mix64(seed + 0x9e3779b97f4a7c15ull * (cand + 1)) + ctr) >> 32);
}

__host__ __device__ inline int search_row_weight(uint64_t seed, uint64_t cand, int r) {
    return (search_hash(seed, cand, (uint32_t)r) & 7u) == 0u ? 3 : 4;
}

__host__ __device__ inline uint32_t search_perm_key(uint64_t seed, uint64_t cand, int j, int k, int col) {
    return (search_hash(seed, cand, 0x10000u + (uint32_t)(j * k + col)) & ~1023u) | (uint32_t)col;
}

// host form; taps_out holds 4 entries per row (the 4th is unused for a weight-3 row), w_out the weights
inline void search_candidate_host(int k, uint64_t seed, uint64_t cand, std::vector<uint16_t> &taps_out,
                                  std::vector<uint8_t> &w_out) {
    w_out.assign(k, 0);
    std::vector<int> P(k + 1, 0);
    for (int r = 0; r < k; r++) {
        w_out[r] = (uint8_t)search_row_weight(seed, cand, r);
        P[r + 1] = P[r] + w_out[r];
    }
    const int rounds = (P[k] + k - 1) / k;
    std::vector<std::vector<uint16_t>> perm(rounds, std::vector<uint16_t>(k));
    for (int j = 0; j < rounds; j++) {
        std::vector<std::pair<uint32_t, uint16_t>> key(k);
        for (int c = 0; c < k; c++) key[c] = {search_perm_key(seed, cand, j, k, c), (uint16_t)c};
        std::sort(key.begin(), key.end());
        for (int c = 0; c < k; c++) perm[j][c] = key[c].second;
    }
    for (int j = 1; j < rounds; j++) {
        int r = 0;
        while (r < k && !(P[r] < j * k && j * k < P[r] + w_out[r])) r++;
        if (r == k) continue;                                   // the boundary falls between two rows
        const int ntail = j * k - P[r], h = P[r] + w_out[r] - j * k;
        int t = 0;
        for (int hp = 0; hp < h; hp++) {
            for (;;) {
                bool clash = false;
                for (int q = 0; q < ntail; q++) clash |= perm[j][hp] == perm[j - 1][k - ntail + q];
                if (!clash) break;
                std::swap(perm[j][hp], perm[j][h + t]);
                t++;
            }
        }
    }
    taps_out.assign((size_t)4 * k, 0);
    for (int r = 0; r < k; r++)
        for (int q = 0; q < w_out[r]; q++) {
            const int pos = P[r] + q;
            taps_out[(size_t)4 * r + q] = perm[pos / k][pos % k];
        }
}

}  // namespace bbb
