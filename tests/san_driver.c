/* Sanitizer run of the oracle (tests/test_oracle.py::test_oracle_under_sanitizers): built with
 * gcc -fsanitize=address,undefined together with oracle/bbb_oracle.c; exercises every entry point on small
 * inputs so that out-of-bounds accesses or undefined shifts in the checker itself would be reported. */
#include "../oracle/bbb_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    static bbo_lutopt m;
    /* packed taps file: one row per line */
    FILE *f = fopen(argv[1], "r");
    if (!f) return 3;
    static uint16_t taps[4 * 512];
    static uint32_t off[513];
    char line[4096];
    int k = 0, nt = 0;
    while (fgets(line, sizeof line, f)) {
        char *p = line, *e;
        off[k] = (uint32_t)nt;
        for (;;) {
            long v = strtol(p, &e, 10);
            if (e == p) break;
            taps[nt++] = (uint16_t)v;
            p = e;
        }
        if ((uint32_t)nt > off[k]) k++;
    }
    off[k] = (uint32_t)nt;
    fclose(f);
    if (bbo_lutopt_from_packed(&m, k, taps, off)) return 4;
    uint64_t init[BBO_WORDS] = {1}, x[BBO_WORDS];
    unsigned long long sum = 0;
    int8_t a[1000], b[1000];
    bbo_awgn_stream_i8(&m, init, 3, 1000, a);
    if (k == 256) {
        bbo_awgn_stream_i8_fast256(&m, init, 3, 1000, b);
        if (memcmp(a, b, sizeof a)) return 5;
    }
    for (int i = 0; i < 1000; i++) sum += (unsigned char)a[i];
    bbo_lutopt_run(&m, init, 77, x);
    sum += (unsigned long long)bbo_clt_wrap(bbo_clt_tree(x, k), k) + (unsigned long long)bbo_clt_popcount(x, k);
    const int ks[7] = {7, 9, 11, 15, 20, 23, 31};
    for (int q = 0; q < 7; q++) {
        uint64_t st = 1, st2 = 1, st3 = 1, words[40], words2[40], nerr = 0;
        uint8_t bits[2500], err[2500], rl[2500];
        bbo_prbs_bits(ks[q], &st, 2500, bits);
        bbo_prbs_packed(ks[q], &st2, 2500, words);
        bbo_prbs_packed_fast(ks[q], &st3, 2500, words2);
        if (memcmp(words, words2, 39 * 8) || st != st2 || st2 != st3) return 6;
        uint64_t s4 = 1;
        bbo_prbs_check_packed(ks[q], &s4, 2500, words, &nerr);
        if (nerr) return 7;
        bits[1200] ^= 1;
        bbo_prbs_detector_run(ks[q], bits, 2500, err, rl);
        uint64_t ew[40], rw[40], stats[4];
        words[10] ^= 1ull << 63;
        bbo_prbs_detector_packed(ks[q], words, 2500, ew, rw, stats);
        sum += stats[0] + stats[3] + err[1200];
    }
    if (k == 256) {
        bbo_trial t = {31, 1, 100, 8, 16, 5, 3000};
        uint64_t nb, ne;
        bbo_ber_trial(&m, init, &t, &nb, &ne);
        sum += ne;
        int16_t coeffs[64], out[500];
        for (int i = 0; i < 64; i++) coeffs[i] = (int16_t)((i * 37) % 200 - 100);
        bbo_shaper_i16(coeffs, 0, 31, 1, 5, 500, out);
        bbo_shaper_i16(coeffs, 1, 31, 1, 0, 500, out);
        bbo_tx_i16(&m, init, coeffs, 0, 31, 1, 1, 1, 15, 16, 9, 500, out);
        uint8_t sb[500];
        sum += bbo_rx_slice(out, 500, 8, 3, 0, sb) + bbo_rx_slice(out, 500, 4, 499, 1, sb);
    }
    const uint64_t colw[8] = {0x7400000000000000ull, 0x5800000000000000ull, 0xC500000000000000ull, 0xD000000000000000ull,
                              0xD500000000000000ull, 0xE600000000000000ull, 0xF100000000000000ull, 0x4700000000000000ull};
    const uint8_t xb[8] = {1, 0, 1, 0, 1, 0, 1, 0};
    uint8_t ob[24];
    bbo_rnghunt_recur(8, 8, colw, xb, 24, ob);
    for (int i = 0; i < 24; i++) sum += ob[i];
    printf("ok %llu\n", sum);
    return 0;
}
