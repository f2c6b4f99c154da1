/*
 * bbb_oracle.c -- CPU restatement of the basebandboard AWGN / PRBS hot path.
 * TEST INFRASTRUCTURE ONLY (see bbb_oracle.h for the rules and the parity status).
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout root).
 */
#include "bbb_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int get_bit(const uint64_t *x, int i) { return (int)((x[i >> 6] >> (i & 63)) & 1u); }
static inline int words_for(int k) { return (k + 63) / 64; }

/* ---- LUTOPT ------------------------------------------------------------------ */

/* software/rnghunt/util/pack.py:6-18 reads the same file: one line per row, one
 * character per column; the row's non-zero columns are its taps (rng.py:38-39). */
int bbo_lutopt_load(bbo_lutopt *m, const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    static char line[BBO_MAX_K + 16];
    int r = 0, k = -1;
    memset(m, 0, sizeof *m);
    while (fgets(line, sizeof line, f)) {
        int len = (int)strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ')) len--;
        if (!len) continue;
        if (k < 0) k = len;
        if (len != k || k > BBO_MAX_K || r >= k) { fclose(f); return -1; }
        for (int c = 0; c < k; c++) {
            if (line[c] == '1') {
                if (m->ntaps[r] >= 8) { fclose(f); return -1; }
                m->taps[r][m->ntaps[r]++] = (uint16_t)c;
            } else if (line[c] != '0') { fclose(f); return -1; }
        }
        r++;
    }
    fclose(f);
    if (r != k || k <= 0) return -1;
    m->k = k;
    return 0;
}

/* rng.py:42-55 from_packed: a[row, idx] = 1 for idx in packed[row]. */
int bbo_lutopt_from_packed(bbo_lutopt *m, int k, const uint16_t *taps_flat, const uint32_t *row_off)
{
    if (k <= 0 || k > BBO_MAX_K) return -1;
    memset(m, 0, sizeof *m);
    m->k = k;
    for (int r = 0; r < k; r++) {
        int n = (int)(row_off[r + 1] - row_off[r]);
        if (n < 0 || n > 8) return -1;
        m->ntaps[r] = n;
        for (int j = 0; j < n; j++) {
            uint16_t c = taps_flat[row_off[r] + j];
            if (c >= k) return -1;
            m->taps[r][j] = c;
        }
    }
    return 0;
}

/* rng.py:38-40: every new state bit is the XOR of its row's taps taken from the OLD
 * state (synchronous assignment), equivalently x = (A x) mod 2 (rng.py:134). */
void bbo_lutopt_step(const bbo_lutopt *m, const uint64_t *x, uint64_t *xnew)
{
    uint64_t out[BBO_WORDS] = {0};
    for (int r = 0; r < m->k; r++) {
        int b = 0;
        for (int j = 0; j < m->ntaps[r]; j++) b ^= get_bit(x, m->taps[r][j]);
        out[r >> 6] |= (uint64_t)b << (r & 63);
    }
    memcpy(xnew, out, sizeof(uint64_t) * (size_t)words_for(m->k));
}

void bbo_lutopt_run(const bbo_lutopt *m, const uint64_t *init, uint64_t nsteps, uint64_t *xout)
{
    uint64_t x[BBO_WORDS] = {0};
    memcpy(x, init, sizeof(uint64_t) * (size_t)words_for(m->k));
    for (uint64_t t = 0; t < nsteps; t++) bbo_lutopt_step(m, x, x);
    memcpy(xout, x, sizeof(uint64_t) * (size_t)words_for(m->k));
}

/* Bulk forms used by the fixtures that were produced by running the reference's own Python
 * (tests/golden/ref_*): the states A^(first_step+1+i) init, i < nstates, as ceil(k/64) words each. */
void bbo_lutopt_states(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step, uint64_t nstates,
                       uint64_t *out)
{
    uint64_t x[BBO_WORDS] = {0};
    int nw = words_for(m->k);
    bbo_lutopt_run(m, init, first_step, x);
    for (uint64_t i = 0; i < nstates; i++) {
        bbo_lutopt_step(m, x, x);
        memcpy(out + i * (uint64_t)nw, x, sizeof(uint64_t) * (size_t)nw);
    }
}

/* The uniform word stream LUTOPT.x as 32-bit words, k/32 per state (k a multiple of 32).
 * msb_first = 0: word j = state bits 32j .. 32j+31, bit 32j the LSB (the HDL integer, rng.py:135).
 * msb_first = 1: the dieharder dump of software/rnghunt/util/verify.py:46-52 -- the state printed as
 * a string x[0] x[1] ... and cut into 32-character binary numbers, so bit 32j is the MSB of word j. */
int bbo_lutopt_words_u32(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step, uint64_t nstates,
                         int msb_first, uint32_t *out)
{
    if (m->k % 32) return -1;
    uint64_t x[BBO_WORDS] = {0};
    int wps = m->k / 32;
    bbo_lutopt_run(m, init, first_step, x);
    for (uint64_t i = 0; i < nstates; i++) {
        bbo_lutopt_step(m, x, x);
        for (int j = 0; j < wps; j++) {
            uint32_t w = 0;
            for (int b = 0; b < 32; b++)
                w |= (uint32_t)get_bit(x, 32 * j + b) << (msb_first ? 31 - b : b);
            out[i * (uint64_t)wps + (uint64_t)j] = w;
        }
    }
    return 0;
}

/* ---- CLT tree ---------------------------------------------------------------- */

/* rng.py:96-105 / clt-grng-evaluate.py:10-15: level 0 is x[2j]-x[2j+1] over the state
 * bits, every further level subtracts neighbouring pairs, until one value remains. */
int bbo_clt_tree(const uint64_t *x, int n)
{
    int v[BBO_MAX_K];
    for (int i = 0; i < n; i++) v[i] = get_bit(x, i);
    for (int width = n; width > 1; width /= 2)
        for (int j = 0; j < width / 2; j++) v[j] = v[2 * j] - v[2 * j + 1];
    return v[0];
}

/* The tree over many caller-supplied words (stride ceil(n/64) u64), un-truncated values. */
void bbo_clt_tree_bulk(const uint64_t *x, int n, uint64_t nstates, int16_t *out)
{
    int nw = words_for(n);
    for (uint64_t i = 0; i < nstates; i++) out[i] = (int16_t)bbo_clt_tree(x + i * (uint64_t)nw, n);
}

int bbo_clt_popcount(const uint64_t *x, int n)
{
    int s = 0;
    for (int i = 0; i < n; i++)
        if (get_bit(x, i)) s += (__builtin_popcount((unsigned)i) & 1) ? -1 : 1;
    return s;
}

/* rng.py:78,108: output Signal is log2(n) bits wide, signed; wider tree values wrap. */
int bbo_clt_wrap(int v, int n)
{
    int bits = 0;
    while ((1 << bits) < n) bits++;
    unsigned mask = (1u << bits) - 1u;
    unsigned u = (unsigned)v & mask;
    return (u >> (bits - 1)) ? (int)u - (1 << bits) : (int)u;
}

void bbo_awgn_stream_i8(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step,
                        uint64_t nsamples, int8_t *out)
{
    uint64_t x[BBO_WORDS] = {0};
    bbo_lutopt_run(m, init, first_step, x);
    for (uint64_t i = 0; i < nsamples; i++) {
        bbo_lutopt_step(m, x, x);
        out[i] = (int8_t)bbo_clt_wrap(bbo_clt_tree(x, m->k), m->k);
    }
}

/* k = 256 fast path: x' = XOR over set state bits c of column c of A, with the columns
 * pre-combined per state byte (32 tables x 256 entries x 256 bit); sample by the
 * popcount closed form with the Thue-Morse sign mask.  Same function as above. */
void bbo_awgn_stream_i8_fast256(const bbo_lutopt *m, const uint64_t *init, uint64_t first_step,
                                uint64_t nsamples, int8_t *out)
{
    if (m->k != 256) { bbo_awgn_stream_i8(m, init, first_step, nsamples, out); return; }
    typedef struct { uint64_t w[4]; } v256;
    v256 *tab = (v256 *)calloc(32 * 256, sizeof(v256));
    v256 col[256];
    memset(col, 0, sizeof col);
    for (int r = 0; r < 256; r++)
        for (int j = 0; j < m->ntaps[r]; j++) col[m->taps[r][j]].w[r >> 6] |= 1ull << (r & 63);
    for (int p = 0; p < 32; p++)
        for (int v = 1; v < 256; v++) {
            int low = __builtin_ctz((unsigned)v);
            v256 a = tab[p * 256 + (v & (v - 1))];
            for (int q = 0; q < 4; q++) a.w[q] ^= col[p * 8 + low].w[q];
            tab[p * 256 + v] = a;
        }
    uint64_t mplus[4] = {0};
    for (int i = 0; i < 256; i++)
        if (!(__builtin_popcount((unsigned)i) & 1)) mplus[i >> 6] |= 1ull << (i & 63);
    uint64_t x[4];
    uint64_t tmp[BBO_WORDS];
    bbo_lutopt_run(m, init, first_step, tmp);
    memcpy(x, tmp, sizeof x);
    for (uint64_t i = 0; i < nsamples; i++) {
        uint64_t y0 = 0, y1 = 0, y2 = 0, y3 = 0;
        for (int p = 0; p < 32; p++) {
            const v256 *e = &tab[p * 256 + ((x[p >> 3] >> ((p & 7) * 8)) & 0xff)];
            y0 ^= e->w[0]; y1 ^= e->w[1]; y2 ^= e->w[2]; y3 ^= e->w[3];
        }
        x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3;
        int pos = __builtin_popcountll(y0 & mplus[0]) + __builtin_popcountll(y1 & mplus[1]) +
                  __builtin_popcountll(y2 & mplus[2]) + __builtin_popcountll(y3 & mplus[3]);
        int all = __builtin_popcountll(y0) + __builtin_popcountll(y1) +
                  __builtin_popcountll(y2) + __builtin_popcountll(y3);
        out[i] = (int8_t)(2 * pos - all);   /* +128 wraps to -128 exactly as an 8-bit signed Signal */
    }
    free(tab);
}

/* ---- PRBS -------------------------------------------------------------------- */

int bbo_prbs_tap(int k)
{
    switch (k) {            /* TAPS, prbs.py:14 */
    case 7: return 6;  case 9: return 5;  case 11: return 9; case 15: return 14;
    case 20: return 3; case 23: return 18; case 31: return 28;
    default: return 0;
    }
}

/* prbs.py:32-35 (model prbs.py:112-113): x = s[k-1] ^ s[tap-1]; s = (s << 1 | x) mod 2^k. */
static inline int prbs_next(int k, int tap, uint64_t *s)
{
    int bit = (int)(((*s >> (k - 1)) ^ (*s >> (tap - 1))) & 1u);
    *s = ((*s << 1) | (uint64_t)bit) & ((1ull << k) - 1ull);
    return bit;
}

int bbo_prbs_bits(int k, uint64_t *state, uint64_t nbits, uint8_t *bits)
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;
    for (uint64_t t = 0; t < nbits; t++) bits[t] = (uint8_t)prbs_next(k, tap, state);
    return 0;
}

int bbo_prbs_packed(int k, uint64_t *state, uint64_t nbits, uint64_t *words)
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;
    uint64_t nw = (nbits + 63) / 64;
    memset(words, 0, nw * sizeof(uint64_t));
    for (uint64_t t = 0; t < nbits; t++)
        words[t >> 6] |= (uint64_t)prbs_next(k, tap, state) << (t & 63);
    return 0;
}

/* Word-parallel form for the timed baseline.  The emitted stream obeys
 * b[t] = b[t-k] ^ b[t-tap] (from prbs.py:32-35), hence also with both lags doubled
 * (squaring over GF(2)); lags >= 64 let a whole 64-bit word be formed from earlier
 * words with two funnel shifts.  History before t = 0 is the initial LFSR state:
 * b[-1-i] = s[i]. */
int bbo_prbs_packed_fast(int k, uint64_t *state, uint64_t nbits, uint64_t *words)
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;
    int e = 0;
    while ((tap << e) < 64) e++;
    const int lk = k << e, lt = tap << e;          /* both >= 64 */
    const int hist = (lk + 63) / 64 + 1;           /* words of history kept in front */
    uint64_t nw = (nbits + 63) / 64;
    uint64_t *buf = (uint64_t *)calloc((size_t)hist + nw, sizeof(uint64_t));
    /* history bits: index h*64 + j (h < hist) is stream position t = (h - hist)*64 + j < 0.
     * The state only defines b[-k..-1]; extend backwards with b[t] = b[t+k] ^ b[t+k-tap]. */
    {
        int nh = hist * 64;
        uint8_t *hb = (uint8_t *)calloc((size_t)nh, 1);
        for (int i = 0; i < k; i++) hb[nh - 1 - i] = (uint8_t)((*state >> i) & 1u);
        for (int p = nh - k - 1; p >= 0; p--) hb[p] = hb[p + k] ^ hb[p + k - tap];
        for (int p = 0; p < nh; p++) buf[p >> 6] |= (uint64_t)hb[p] << (p & 63);
        free(hb);
    }
    for (uint64_t n = 0; n < nw; n++) {
        uint64_t pos = (uint64_t)hist * 64 + n * 64;   /* bit index of this word in buf */
        uint64_t a = pos - (uint64_t)lk, b = pos - (uint64_t)lt;
        uint64_t wa = buf[a >> 6] >> (a & 63);
        if (a & 63) wa |= buf[(a >> 6) + 1] << (64 - (a & 63));
        uint64_t wb = buf[b >> 6] >> (b & 63);
        if (b & 63) wb |= buf[(b >> 6) + 1] << (64 - (b & 63));
        buf[hist + n] = wa ^ wb;
    }
    memcpy(words, buf + hist, nw * sizeof(uint64_t));
    if (nbits & 63) words[nw - 1] &= (1ull << (nbits & 63)) - 1ull;
    /* new state: s[i] = b[nbits-1-i] */
    uint64_t s = 0;
    for (int i = 0; i < k; i++) {
        uint64_t p = (uint64_t)hist * 64 + nbits - 1 - (uint64_t)i;
        s |= ((buf[p >> 6] >> (p & 63)) & 1ull) << i;
    }
    *state = s;
    free(buf);
    return 0;
}

int bbo_prbs_check_packed(int k, uint64_t *state, uint64_t nbits, const uint64_t *words, uint64_t *nerr)
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;
    uint64_t e = 0;
    for (uint64_t t = 0; t < nbits; t++)
        e += (uint64_t)(prbs_next(k, tap, state) ^ (int)((words[t >> 6] >> (t & 63)) & 1u));
    *nerr = e;
    return 0;
}

/* ---- PRBSErrorDetector (prbs.py:61-99) ----------------------------------------- */

int bbo_prbs_detector_run(int k, const uint8_t *bits, uint64_t n, uint8_t *err, uint8_t *reload)
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;                          /* prbs.py:55-56 ValueError */
    const uint64_t mask = (1ull << k) - 1ull;
    /* register reset values: prbs=1 (:62), bit_in=0, err_sr=all ones (:80), reload_ctr=0 */
    uint64_t prbs = 1, err_sr = mask;
    int bit_in = 0, reload_ctr = 0;
    for (uint64_t i = 0; i < n; i++) {
        /* combinational values seen by the clock edge (pre-edge registers) */
        int feedback = (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1u);   /* :67 */
        int rl = reload_ctr != 0;                                              /* :99 */
        int prbs_in = rl ? bit_in : feedback;                                  /* :75-76 */
        int e = bit_in != feedback;                                            /* :79 */
        int err_count = __builtin_popcountll(err_sr);                          /* :86-87 */
        /* clock edge: all right-hand sides use the pre-edge values */
        uint64_t prbs_n = ((prbs << 1) | (uint64_t)prbs_in) & mask;            /* :68 */
        uint64_t err_sr_n = ((err_sr << 1) | (uint64_t)e) & mask;              /* :81 */
        int reload_ctr_n = reload_ctr;
        if (err_count > k / 2) {                                               /* :92-94 */
            reload_ctr_n = k + k / 2;
            err_sr_n = 0;                 /* later statement wins over the shift */
        } else if (rl) {                                                       /* :95-97 */
            reload_ctr_n = reload_ctr - 1;
        }
        bit_in = bits[i] & 1;                                                  /* :66 */
        prbs = prbs_n; err_sr = err_sr_n; reload_ctr = reload_ctr_n;
        /* outputs as a testbench reads them after the edge (prbs.py:149-150) */
        int fb2 = (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1u);
        if (err) err[i] = (uint8_t)(bit_in != fb2);
        if (reload) reload[i] = (uint8_t)(reload_ctr != 0);
    }
    return 0;
}

/* The same machine on a packed stream, with the totals the GPU stream runner reports. */
int bbo_prbs_detector_packed(int k, const uint64_t *words, uint64_t nbits, uint64_t *err_words,
                             uint64_t *reload_words, uint64_t stats[4])
{
    int tap = bbo_prbs_tap(k);
    if (!tap) return -1;
    const uint64_t mask = (1ull << k) - 1ull;
    uint64_t prbs = 1, err_sr = mask;
    int bit_in = 0, reload_ctr = 0;
    uint64_t nw = (nbits + 63) / 64;
    if (err_words) memset(err_words, 0, nw * 8);
    if (reload_words) memset(reload_words, 0, nw * 8);
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    for (uint64_t i = 0; i < nbits; i++) {
        int feedback = (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1u);
        int rl = reload_ctr != 0;
        int prbs_in = rl ? bit_in : feedback;
        int e = bit_in != feedback;
        int err_count = __builtin_popcountll(err_sr);
        uint64_t prbs_n = ((prbs << 1) | (uint64_t)prbs_in) & mask;
        uint64_t err_sr_n = ((err_sr << 1) | (uint64_t)e) & mask;
        int reload_ctr_n = reload_ctr;
        if (err_count > k / 2) {
            reload_ctr_n = k + k / 2;
            err_sr_n = 0;
            stats[3]++;                                /* resyncs */
        } else if (rl) {
            reload_ctr_n = reload_ctr - 1;
        }
        bit_in = (int)((words[i >> 6] >> (i & 63)) & 1u);
        prbs = prbs_n; err_sr = err_sr_n; reload_ctr = reload_ctr_n;
        int eo = bit_in != (int)(((prbs >> (k - 1)) ^ (prbs >> (tap - 1))) & 1u);
        int ro = reload_ctr != 0;
        if (eo && err_words) err_words[i >> 6] |= 1ull << (i & 63);
        if (ro && reload_words) reload_words[i >> 6] |= 1ull << (i & 63);
        stats[1] += (uint64_t)eo;                      /* errors_raw */
        stats[2] += (uint64_t)ro;                      /* reload_clocks */
        stats[0] += (uint64_t)(eo && !ro);             /* errors while synced */
    }
    return 0;
}

/* ---- TX noise path + RX slicer --------------------------------------------------- */

static inline int wrap12(int v)
{
    unsigned u = (unsigned)v & 0xfffu;
    return (u & 0x800u) ? (int)u - 4096 : (int)u;
}

/* tx.py:75-77 noise = grng.x * noise_var into a 12-bit signed register;
 * tx.py:80-81 x = bits + noise into a 12-bit signed register;
 * rx.py:29 sliced = ~sample[-1], i.e. 1 when the sample is >= 0.
 * bit = 1 selects the positive pulse (bitshaper.py:52-58: ROM address LSB = data bit,
 * negative coefficient stored first). */
int bbo_txrx_decide(int g_i8, int bit, int amp, int noise_var)
{
    int noise = wrap12(g_i8 * noise_var);
    int x = wrap12((bit ? amp : -amp) + noise);
    return x >= 0;
}

/* BUILD-DEFINED trial (no counterpart in the reference). */
int bbo_ber_trial(const bbo_lutopt *m, const uint64_t *init, const bbo_trial *t,
                  uint64_t *bits_out, uint64_t *errors_out)
{
    int tap = bbo_prbs_tap(t->prbs_k);
    if (!tap) return -1;
    uint64_t x[BBO_WORDS] = {0};
    bbo_lutopt_run(m, init, t->warmup + t->first_bit, x);
    uint64_t s = t->prbs_state;
    for (uint64_t i = 0; i < t->first_bit; i++) (void)prbs_next(t->prbs_k, tap, &s);
    uint64_t errors = 0;
    for (uint64_t i = 0; i < t->nbits; i++) {
        bbo_lutopt_step(m, x, x);
        int g = bbo_clt_wrap(bbo_clt_tree(x, m->k), m->k);
        int b = prbs_next(t->prbs_k, tap, &s);
        errors += (uint64_t)(bbo_txrx_decide(g, b, t->amp, t->noise_var) != b);
    }
    *bits_out = t->nbits;
    *errors_out = errors;
    return 0;
}

/* ---- PRBSShaper + TX (bitshaper.py:25-86, tx.py:60-81) ------------------------------------ */

int bbo_shaper_i16(const int16_t coeffs[64], int source, int k, uint64_t prbs_state,
                   uint64_t first_sample, uint64_t nsamples, int16_t *out)
{
    int tap = bbo_prbs_tap(k);
    if (source == 0 && !tap) return -1;
    if (nsamples == 0) return 0;
    /* data bits 0 .. mmax */
    int64_t last = (int64_t)(first_sample + nsamples - 1) - 17;
    int64_t mmax = last >= 0 ? last / 8 : -1;
    uint8_t *bits = (uint8_t *)calloc((size_t)(mmax + 2), 1);
    if (source == 0) {
        uint64_t s = prbs_state;
        for (int64_t m = 0; m <= mmax; m++) bits[m] = (uint8_t)prbs_next(k, tap, &s);
    } else {
        for (int64_t m = 0; m <= mmax; m++) bits[m] = (uint8_t)((m & 255) == 0);   /* Pulser: counter == 0 (tx.py:28-30) */
    }
    for (uint64_t i = 0; i < nsamples; i++) {
        int64_t np = (int64_t)(first_sample + i) - 17;
        int64_t M = np >= 0 ? np / 8 : -((-np + 7) / 8);      /* floor division */
        int ph = (int)(np - 8 * M);
        int sum = 0;
        for (int idx = 0; idx < 8; idx++) {                   /* ROM idx holds c[8 idx .. 8 idx + 7] (:44-58) */
            int64_t mm = M - idx;
            int b = mm >= 0 ? bits[mm] : 0;                   /* sr resets to 0 */
            int c = coeffs[8 * idx + ph];
            sum += b ? c : -c;                                /* address LSB = data bit, -c stored first (:52-58,:74) */
        }
        out[i] = (int16_t)wrap12(sum);
    }
    free(bits);
    return 0;
}

int bbo_tx_i16(const bbo_lutopt *m, const uint64_t *init, const int16_t coeffs[64], int source, int k,
               uint64_t prbs_state, int bit_en, int noise_en, int noise_var, uint64_t warmup,
               uint64_t first_sample, uint64_t nsamples, int16_t *out)
{
    if (bbo_shaper_i16(coeffs, source, k, prbs_state, first_sample, nsamples, out)) return -1;
    int8_t *g = (int8_t *)malloc(nsamples ? nsamples : 1);
    bbo_awgn_stream_i8(m, init, warmup + first_sample, nsamples, g);
    for (uint64_t i = 0; i < nsamples; i++) {
        int bitmux = bit_en ? out[i] : 0;                              /* tx.py:65-66 */
        int noisemux = noise_en ? wrap12(g[i] * noise_var) : 0;        /* tx.py:75-77 */
        out[i] = (int16_t)wrap12(bitmux + noisemux);                   /* tx.py:80-81 */
    }
    free(g);
    return 0;
}

/* ---- RX slicer (rx.py:29; decode.py:15-16) ---------------------------------------------- */

uint64_t bbo_rx_slice(const int16_t *samples, uint64_t nsamples, uint64_t stride, uint64_t phase, int strict,
                      uint8_t *bits)
{
    uint64_t n = 0;
    for (uint64_t i = phase; i < nsamples; i += stride)
        bits[n++] = (uint8_t)(strict ? samples[i] > 0 : samples[i] >= 0);   /* sliced = ~sample[-1] */
    return n;
}

/* ---- rnghunt BinaryMatrix::dot / recur (binary_matrix.rs:53-76) ------------------- */

int bbo_rnghunt_recur(int nrows, int ncols, const uint64_t *col_words, const uint8_t *x_bits,
                      int n, uint8_t *out_bits)
{
    if (nrows != ncols || nrows > BBO_MAX_K) return -1;
    int wpc = (nrows + 63) / 64;
    uint8_t x[BBO_MAX_K], y[BBO_MAX_K];
    memcpy(x, x_bits, (size_t)ncols);
    for (int s = 0; s < n; s++) {
        memset(y, 0, (size_t)nrows);
        for (int c = 0; c < ncols; c++) {
            if (!x[c]) continue;                                  /* :56-57 */
            for (int r = 0; r < nrows; r++)                       /* column XORed into result, MSbit = row 0 */
                y[r] ^= (uint8_t)((col_words[c * wpc + r / 64] >> (63 - (r % 64))) & 1u);
        }
        memcpy(x, y, (size_t)nrows);
        out_bits[s] = x[0];                                       /* :73 */
    }
    return 0;
}
