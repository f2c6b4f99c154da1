// Sanitizer run of the product's host-side GF(2) arithmetic (tests/test_gf2.py::test_host_arithmetic_under_sanitizers):
// csrc/gf2poly.hpp, csrc/search_rng.hpp and csrc/gf2.hpp are plain C++ headers, built here with
// g++ -fsanitize=address,undefined (sanitizers on the host build only).
#define __host__
#define __device__
#include "../basebandboard_amd/csrc/gf2.hpp"
#include "../basebandboard_amd/csrc/gf2poly.hpp"
#include "../basebandboard_amd/csrc/search_rng.hpp"

#include <cstdio>

using namespace bbb;

int main() {
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
