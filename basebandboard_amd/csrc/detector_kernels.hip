// PRBSErrorDetector (gateware/bbb/prbs.py:43-99) over ONE long packed bit stream, cycle exact,
// in parallel chunks with state hand-off.
//
// The detector is a serial machine (LFSR, error history, reload counter, input register), but it
// forgets: whatever its past, some hundred clocks of the same input drive two instances into the
// same state (both reload from the received bits, both histories fill with the same error bits).
// So every chunk is run by one lane from a SPECULATIVE start state -- the reset state placed
// `warm` bits before the chunk -- and the speculation is then checked, not trusted:
//
//   run     lane c: reset state at (start_c - warm), run silently to start_c, remember that state
//           (spec[c]), run the chunk writing err / reload bits and counters, remember end[c]
//   verify  chunk c is consistent when spec[c] == end[c-1]; chunk 0 starts from the true reset
//           state, so by induction a fully consistent chain IS the serial run, bit for bit
//   repair  every inconsistent chunk is re-run from end[c-1]; verify again; repeat.  The first
//           inconsistent chunk always has a true predecessor, so every pass fixes at least one
//           chunk for good; in practice one pass (a few percent of the chunks at BER 1e-2) suffices.
//
// State per lane: two 32-bit registers (k <= 31), a counter and the input register; the work is
// integer VALU (about 16 lane-ops per clock), HBM traffic is 1/8 B read + 2/8 B written per bit.
#include "bbb_common.hpp"

#include <mutex>
#include <vector>

namespace bbb {

typedef unsigned long long u64;

struct DetState {
    uint32_t prbs, err_sr;
    int32_t reload_ctr;
    uint32_t bit_in;      // the input register (prbs.py:66)
};

__host__ __device__ constexpr int det_tap_of(int k) {
    return k == 7 ? 6 : k == 9 ? 5 : k == 11 ? 9 : k == 15 ? 14 : k == 20 ? 3 : k == 23 ? 18 : k == 31 ? 28 : 0;   // prbs.py:14
}

struct DetCount { u64 err_synced, err_raw, reload_clocks, resyncs; };

// 64 (or `nvalid` < 64) clocks on input word w.  Output bit i of errw / rlw = `err` / `reload` sampled
// after clock i, what the reference testbench reads (prbs.py:146-150).  All one-bit signals are kept as
// 0 / 1 in bit 0 of a register so that every line below is one instruction:
//   e   = bit_in ^ feedback                      (:79; it is also the `err` output of the previous clock)
//   rl  = reload_ctr != 0 = min(reload_ctr, 1)   (:99)
//   pin = rl ? bit_in : feedback = fb ^ (e & rl) (:75-76)
template <int K, bool EMIT>
__device__ __forceinline__ void det_word(DetState &s, u64 w, int nvalid, u64 &errw, u64 &rlw, unsigned &trig) {
    constexpr int TAP = det_tap_of(K);
    constexpr uint32_t MASK = (uint32_t)((1ull << K) - 1ull);
    uint32_t prbs = s.prbs, err_sr = s.err_sr, bit_in = s.bit_in;
    uint32_t ctr = (uint32_t)s.reload_ctr;
    uint32_t fb = ((prbs >> (K - 1)) ^ (prbs >> (TAP - 1))) & 1u;            // prbs.py:73-74
    uint32_t e = bit_in ^ fb, rl = ctr < 1u ? ctr : 1u;
    errw = 0; rlw = 0;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        uint32_t wh = (uint32_t)(w >> (32 * half)), eh = 0, rh = 0;
        const int n = nvalid - 32 * half < 0 ? 0 : (nvalid - 32 * half > 32 ? 32 : nvalid - 32 * half);
#pragma unroll 8
        for (int i = 0; i < n; i++) {
            const bool t = __builtin_popcount(err_sr) > K / 2;                // :86-87, :92
            const uint32_t pin = __builtin_amdgcn_bitop3_b32(fb, e, rl, 0x78);     // fb ^ (e & rl)
            prbs = (prbs << 1) | pin;                                         // :68 (bits >= K are never read)
            err_sr = t ? 0u : (((err_sr << 1) | e) & MASK);                   // :81, :94
            const uint32_t dec = ctr - rl;                                    // :95-97
            ctr = t ? (uint32_t)(K + K / 2) : dec;                            // :93
            if (EMIT) trig += t ? 1u : 0u;
            bit_in = (wh >> i) & 1u;                                          // :66
            fb = ((prbs >> (K - 1)) ^ (prbs >> (TAP - 1))) & 1u;
            e = bit_in ^ fb;
            rl = ctr < 1u ? ctr : 1u;
            if (EMIT) {
                eh |= e << i;
                rh |= rl << i;
            }
        }
        if (EMIT) {
            errw |= (u64)eh << (32 * half);
            rlw |= (u64)rh << (32 * half);
        }
    }
    s.prbs = prbs & MASK; s.err_sr = err_sr; s.reload_ctr = (int32_t)ctr; s.bit_in = bit_in;
}

// The common case, 64 clocks at once: the detector is locked (reload_ctr == 0) and stays locked through the word.
// Then the LFSR free-runs (prbs.py:75-76 with reload == 0), so its next 64 feedback bits F follow from the state
// alone -- x[t] = x[t-K] ^ x[t-TAP], TAP new bits per pair of 64-bit shifts -- the error bits are
// E = (input delayed by the bit_in register) ^ F, and no reload can trigger inside the word if the old history plus
// E hold at most K/2 ones (every K-bit window is a subset).  Otherwise: false, and det_word runs the clocks one by one.
// The free-running sequence as WORDS.  While word after word takes the fast path the feedback words F_n are consecutive
// 64-bit pieces of one LFSR output sequence x, and x[t] = x[t - K 2^m] ^ x[t - TAP 2^m] for every m (the trinomial squared
// m times).  With TAP 2^m >= 64 all 64 bits of the next word depend on EARLIER words only: F_n = two 64-bit windows of
// the last NH = ceil(K 2^m / 64) words, XORed -- four funnel shifts and two XORs instead of a bit-serial-in-chunks-of-TAP
// expansion of the 31-bit state (three dependent rounds for K = 31).  DetAux holds those words (registers only, never
// stored: any word that leaves the fast path empties it).  K = 20 (TAP = 3: NH = 10) keeps the expansion from the state.
template <int K> struct DetLag {
    static constexpr int TAP = det_tap_of(K);
    static constexpr int M = TAP >= 64 ? 0 : TAP * 2 >= 64 ? 1 : TAP * 4 >= 64 ? 2 : TAP * 8 >= 64 ? 3 : TAP * 16 >= 64 ? 4 : 5;
    static constexpr int LAGK = K << M, LAGT = TAP << M;
    static constexpr int NH = (LAGK + 63) / 64;
    static constexpr bool OK = NH <= 3;
};
struct DetAux { u64 fh[3]; int n; };          // fh[NH-1] = the newest feedback word; n = how many are valid

// 64 bits of the history starting at bit OFF (bit 0 = the oldest bit of fh[0]): two V_ALIGNBIT_B32 over three of the
// history's dwords (a 64-bit shift pair + OR costs five to six instructions, two of them quarter rate)
template <int OFF, int NH>
__device__ __forceinline__ u64 det_hist_window(const u64 (&fh)[3]) {
    constexpr int i = OFF / 32, sh = OFF % 32;
    static_assert(OFF >= 0 && OFF + 64 <= 64 * NH, "window must lie inside the history");
    auto dw = [&](int j) -> uint32_t { return (j & 1) ? (uint32_t)(fh[j >> 1] >> 32) : (uint32_t)fh[j >> 1]; };
    if constexpr (sh == 0) {
        return (u64)dw(i) | ((u64)dw(i + 1) << 32);
    } else {
        const uint32_t lo = __builtin_amdgcn_alignbit(dw(i + 1), dw(i), sh);
        const uint32_t hi = __builtin_amdgcn_alignbit(dw(i + 2), dw(i + 1), sh);
        return (u64)lo | ((u64)hi << 32);
    }
}

template <int K, bool EMIT>
__device__ __forceinline__ bool det_word_fast(DetState &s, DetAux &a, u64 w, u64 &errw) {
    constexpr int TAP = det_tap_of(K);
    constexpr uint32_t MASK = (uint32_t)((1ull << K) - 1ull);
    typedef DetLag<K> LG;
    if (s.reload_ctr != 0) { a.n = 0; return false; }
    u64 F = 0;                                              // F bit j = x[t+j]
    bool from_words = false;
    if constexpr (LG::OK) {
        if (a.n >= LG::NH) {
            F = det_hist_window<64 * LG::NH - LG::LAGK, LG::NH>(a.fh) ^ det_hist_window<64 * LG::NH - LG::LAGT, LG::NH>(a.fh);
            from_words = true;
        }
    }
    if (!from_words) {
        // H bit p = x[t-64+p]: the last K sequence bits sit at the top (prbs bit m = x[t-1-m])
        const u64 H = (u64)__builtin_bitreverse32(s.prbs) << 32;
#pragma unroll
        for (int j0 = 0; j0 < 64; j0 += TAP) {
            const int pk = 64 + j0 - K, pt = 64 + j0 - TAP;     // where the two lagged copies start in (H : F)
            const u64 gk = pk < 64 ? ((H >> (pk & 63)) | (pk ? F << ((64 - pk) & 63) : 0ull)) : (pk == 64 ? F : F >> ((pk - 64) & 63));
            const u64 gt = pt < 64 ? ((H >> (pt & 63)) | (pt ? F << ((64 - pt) & 63) : 0ull)) : (pt == 64 ? F : F >> ((pt - 64) & 63));
            F |= ((gk ^ gt) & ((1ull << TAP) - 1ull)) << j0;
        }
    }
    const u64 E = ((w << 1) | (u64)s.bit_in) ^ F;           // e_i = bit_in ^ feedback (:79), bit_in = the previous input bit
    if (__builtin_popcountll(E) + __builtin_popcount(s.err_sr) > K / 2) { a.n = 0; return false; }
    s.prbs = __builtin_bitreverse32((uint32_t)(F >> 32)) & MASK;      // the newest bit at bit 0 (:68)
    s.err_sr = __builtin_bitreverse32((uint32_t)(E >> 32)) & MASK;    // (:81)
    s.bit_in = (uint32_t)(w >> 63);
    if constexpr (LG::OK) {
#pragma unroll
        for (int i = 0; i + 1 < LG::NH; i++) a.fh[i] = a.fh[i + 1];
        a.fh[LG::NH - 1] = F;
        a.n = a.n < LG::NH ? a.n + 1 : LG::NH;
    }
    if (EMIT) {
        const u64 f64 = ((s.prbs >> (K - 1)) ^ (s.prbs >> (TAP - 1))) & 1u;
        errw = w ^ ((F >> 1) | (f64 << 63));                // err after clock i = input i ^ next feedback
    }
    return true;
}

__device__ __forceinline__ DetState det_reset(int k) {
    DetState s;
    s.prbs = 1u;                                   // prbs.py:62
    s.err_sr = (uint32_t)((1ull << k) - 1ull);     // :80 reset all ones
    s.reload_ctr = 0;
    s.bit_in = 0;
    return s;
}

__device__ __forceinline__ bool det_equal(const DetState &a, const DetState &b) {
    return a.prbs == b.prbs && a.err_sr == b.err_sr && a.reload_ctr == b.reload_ctr && a.bit_in == b.bit_in;
}

// runs words [w0, w1) of the stream from state s; emits outputs and counts when EMIT
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// one input word: the word-at-once path when it applies, else clock by clock; counters when EMIT
// FULL: the caller knows that all 64 bits of the word lie inside the stream (the tiled path: whole chunks), which saves the
// 64-bit bookkeeping of the tail on every word
template <int K, bool EMIT, bool FULL = false>
__device__ __forceinline__ void det_core(DetState &s, DetAux &a, u64 w, u64 word, u64 nbits, u64 &ew, u64 &rw, DetCount &cnt) {
    int nvalid = 64;
    if (!FULL) {
        const u64 left = nbits - w * 64;
        nvalid = left >= 64 ? 64 : (int)left;
    }
    ew = 0; rw = 0;
    unsigned trig = 0;
    bool fast = false;
    if (FULL || nvalid == 64) fast = det_word_fast<K, EMIT>(s, a, word, ew);
    if (!fast) { a.n = 0; det_word<K, EMIT>(s, word, nvalid, ew, rw, trig); }
    if (EMIT) {
        if (fast) {
            // rw = 0 and trig = 0 on the locked path: one popcount serves both error counters
            const u64 pe = (u64)__builtin_popcountll(ew);
            cnt.err_raw += pe;
            cnt.err_synced += pe;
        } else {
            cnt.err_raw += __builtin_popcountll(ew);
            cnt.reload_clocks += __builtin_popcountll(rw);
            cnt.err_synced += __builtin_popcountll(ew & ~rw);
            cnt.resyncs += trig;
        }
    }
}

template <int K, bool EMIT>
__device__ __forceinline__ void det_one(DetState &s, DetAux &a, u64 w, u64 word, u64 nbits, u64 *__restrict err, u64 *__restrict reload,
                                        DetCount &cnt) {
    u64 ew, rw;
    det_core<K, EMIT>(s, a, w, word, nbits, ew, rw, cnt);
    if (EMIT) {
        if (err) err[w] = ew;
        if (reload) reload[w] = rw;
    }
}

// runs words [w0, w1) of the stream from state s, every lane on its own (16-byte loads where aligned)
template <int K, bool EMIT>
__device__ __forceinline__ void det_span(DetState &s, const u64 *__restrict src, u64 w0, u64 w1, u64 nbits,
                                         u64 *__restrict err, u64 *__restrict reload, DetCount &cnt) {
    DetAux a;
    a.n = 0;                                     // the word history of the free-running sequence starts empty
    u64 w = w0;
    if ((w & 1) && w < w1) { det_one<K, EMIT>(s, a, w, src[w], nbits, err, reload, cnt); w++; }
    for (; w + 2 <= w1; w += 2) {
        const u64x2 v = *reinterpret_cast<const u64x2 *>(src + w);
        det_one<K, EMIT>(s, a, w, v.x, nbits, err, reload, cnt);
        det_one<K, EMIT>(s, a, w + 1, v.y, nbits, err, reload, cnt);
    }
    if (w < w1) det_one<K, EMIT>(s, a, w, src[w], nbits, err, reload, cnt);
}

// The same for a whole wave of 64 consecutive full chunks: the 64 lanes' next 16 words (128 bytes each) are fetched
// together as 64 full lines (8 load instructions, each covering 8 rows) into LDS and every lane then reads its own
// row -- one line request per 128 bytes instead of one per 16 bytes.  [first, first + nw) per lane, nw a multiple of 16.
// OUT: the err / reload words are wanted (a totals-only call holds none of the 16 words' outputs: 96 registers less, twice the
// waves per SIMD -- the chunk pass is bound by the latency of its tile loads, not by its instructions)
template <int K, bool EMIT, bool OUT>
__device__ __forceinline__ void det_span_tiled(DetState &s, const u64 *__restrict src, u64 first_row0, u64 row_stride, u64 nw,
                                               u64 my_first, u64 nbits, u64 *__restrict err, u64 *__restrict reload,
                                               DetCount &cnt, u64x2 (*tile)[9], unsigned lane) {
    DetAux a;
    a.n = 0;
    // The CLEAN word (round 4).  In the streams this detector is made for almost every word is error free while the detector
    // is locked with an empty error history.  Then nothing of the machine moves but the free-running LFSR, and its next 64
    // feedback bits are a function of the last NH feedback words -- which, on a clean stretch, are the (one bit delayed)
    // input words themselves.  So a lane in that state (`clean`: reload_ctr == 0, err_sr == 0, history valid) keeps NO
    // DetState: per word it forms V = the input delayed by the bit_in register, predicts F from the history (four
    // V_ALIGNBIT + two XOR) and compares; equal -> the outputs are zero, the counters stand, the history shifts: ~12
    // instructions where the general word-at-once path (popcounts, state and history bookkeeping, output formulae,
    // 64-bit counters, exec-mask handling) takes 80-110.  When every lane of the wave is clean the general path is skipped by
    // a scalar branch.  A lane whose word is not clean re-creates the DetState the general path would hold (prbs = the top K
    // bits of the newest feedback word, bit-reversed; err_sr = 0; reload_ctr = 0; bit_in) and goes through det_core as before;
    // it is clean again after the first word that leaves it locked with an empty history.
    typedef DetLag<K> LGc;
    constexpr uint32_t MASKc = (uint32_t)((1ull << K) - 1ull);
    bool clean = false;
    uint32_t cbit = 0;                               // the bit_in register while `clean`
    auto materialise = [&]() {
        if constexpr (LGc::OK) {
            s.prbs = __builtin_bitreverse32((uint32_t)(a.fh[LGc::NH - 1] >> 32)) & MASKc;
            s.err_sr = 0; s.reload_ctr = 0; s.bit_in = cbit;
        }
    };
    auto step = [&](u64 widx, u64 word, u64 &ew, u64 &rw) {
        if constexpr (LGc::OK) {
            const u64 V = (word << 1) | (u64)cbit;
            const u64 Fp = det_hist_window<64 * LGc::NH - LGc::LAGK, LGc::NH>(a.fh) ^ det_hist_window<64 * LGc::NH - LGc::LAGT, LGc::NH>(a.fh);
            // ... and the `err` output after the word's LAST clock is this word's last input bit against the NEXT feedback bit,
            // x[t+64] = V[64-K] ^ V[64-TAP] on a clean word: an input error in bit 63 shows there and nowhere in V
            const uint32_t vh = (uint32_t)(V >> 32);
            const uint32_t last = (uint32_t)(word >> 32) ^ (vh << (K - 1)) ^ (vh << (det_tap_of(K) - 1));      // bit 31 = err after clock 63
            const bool ok = clean && V == Fp && (int32_t)last >= 0;
            if (ok) {
#pragma unroll
                for (int i = 0; i + 1 < LGc::NH; i++) a.fh[i] = a.fh[i + 1];
                a.fh[LGc::NH - 1] = V;
                cbit = (uint32_t)(word >> 63);
                ew = 0; rw = 0;
            }
            if (__all(ok)) return;                   // (wave uniform: the whole wave is on a clean stretch)
            if (!ok) {
                if (clean) materialise();
                clean = false;
                det_core<K, EMIT, true>(s, a, widx, word, nbits, ew, rw, cnt);
                cbit = s.bit_in;
                clean = s.reload_ctr == 0 && s.err_sr == 0 && a.n >= LGc::NH;
            }
        } else {
            det_core<K, EMIT, true>(s, a, widx, word, nbits, ew, rw, cnt);
        }
    };
    auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // the tile written back the same way: 64 whole lines per 8 store instructions
    auto store_tile = [&](u64 *__restrict out, u64 off) {
        sync();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned row = (unsigned)q * 8 + (lane >> 3), piece = lane & 7;
            *reinterpret_cast<u64x2 *>(out + first_row0 + off + (u64)row * row_stride + piece * 2) = tile[row][piece];
        }
        sync();
    };
    for (u64 off = 0; off < nw; off += 16) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned row = (unsigned)q * 8 + (lane >> 3), piece = lane & 7;
            tile[row][piece] = *reinterpret_cast<const u64x2 *>(src + first_row0 + off + (u64)row * row_stride + piece * 2);
        }
        sync();
        u64x2 in[8];
#pragma unroll
        for (int p = 0; p < 8; p++) in[p] = tile[lane][p];
        sync();                                    // every lane has its row: the tile can take the outputs
        u64x2 eo[OUT ? 8 : 1], ro[OUT ? 8 : 1];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            u64 e0, r0, e1, r1;
            step(my_first + off + 2 * p, in[p].x, e0, r0);
            step(my_first + off + 2 * p + 1, in[p].y, e1, r1);
            if constexpr (OUT) { eo[p].x = e0; eo[p].y = e1; ro[p].x = r0; ro[p].y = r1; }
        }
        if constexpr (!OUT) continue;
        if (EMIT && err) {
#pragma unroll
            for (int p = 0; p < 8; p++) tile[lane][p] = eo[p];
            store_tile(err, off);
        }
        if (EMIT && reload) {
#pragma unroll
            for (int p = 0; p < 8; p++) tile[lane][p] = ro[p];
            store_tile(reload, off);
        }
    }
    if (clean) materialise();
}

// ---------------------------------------------------------------------------------------------
// The SPARSE form of the chunk pass (round 4).  The clean-word test above reads nothing but input words: with V_n = the input
// delayed by one bit (word n shifted left, the top bit of word n - 1 below it), a locked detector with an empty error history
// whose last NH feedback words were the inputs themselves finds word n clean exactly when
//     V_n == R(V_{n-1}, ..., V_{n-NH})   and   bit 63 of word n == V_n[64-K] ^ V_n[64-TAP]
// -- a STATELESS predicate of words n - NH - 1 ... n.  So the stream is classified first, by a streaming kernel that reads
// it once at memory speed and leaves one flag per word (det_classify_kernel: 1/64 of the stream's size), and the serial
// machine then runs only where it has to: a lane in the clean state with NH clean words behind it looks up the next unset
// flag of its chunk and JUMPS there, re-creating its history from the NH + 1 input words in front of the target; everything
// else -- reloads, words with errors, the NH words after one (whose true feedback words differ from the inputs) -- goes
// through the same per-word code as before.  At 1e-3 errors per word a 512-word chunk is ~10 words of work and five cache
// lines instead of 528 words: the call becomes the classification pass.  Outputs, when asked for, are zeroed first and
// written for the words the machine visited.  K = 20 (ten history words) keeps the dense pass.
// ---------------------------------------------------------------------------------------------
// Flags: two 64-bit masks per block of 128 words, flags[2 b] = the EVEN words of the block (bit i = word 128 b + 2 i),
// flags[2 b + 1] = the odd ones -- the shape in which two ballots of a wave that holds 16 bytes per lane deliver them.
// the flags of 256 words held by a wave as 16 bytes per lane: a0 = words base + 2t, base + 2t + 1, a1 = the same 128 words
// further, h = the four words in front of the block in lanes 62, 63
template <int K>
__device__ __forceinline__ void det_classify_256(const u64x2 a0, const u64x2 a1, const u64x2 h, const u64 base, const u64 nfull,
                                                 const unsigned lane, u64 &me0, u64 &mo0, u64 &me1, u64 &mo1) {
    typedef DetLag<K> LG;
    constexpr int NH = LG::NH, TAP = det_tap_of(K);
    auto pair_flags = [&](const u64x2 cur, const u64x2 lower, const u64 first, bool &fe, bool &fo) {
        // first = index of cur.x.  w[j] = word first + 1 - j, j = 0 .. NH + 2, from this lane and the two below it
        const int l1 = (int)((lane - 1u) & 63u), l2 = (int)((lane - 2u) & 63u);
        const u64 c1x = __shfl(cur.x, l1, 64), c1y = __shfl(cur.y, l1, 64), b1x = __shfl(lower.x, l1, 64), b1y = __shfl(lower.y, l1, 64);
        const u64 c2x = __shfl(cur.x, l2, 64), c2y = __shfl(cur.y, l2, 64), b2x = __shfl(lower.x, l2, 64), b2y = __shfl(lower.y, l2, 64);
        u64 w[6];
        w[0] = cur.y; w[1] = cur.x;
        w[2] = lane >= 1 ? c1y : b1y; w[3] = lane >= 1 ? c1x : b1x;
        w[4] = lane >= 2 ? c2y : b2y; w[5] = lane >= 2 ? c2x : b2x;
        auto one = [&](const int o, const u64 n) -> bool {          // the word w[o], index n
            u64 fh[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < NH; q++) fh[q] = (w[o + NH - q] << 1) | (w[o + NH - q + 1] >> 63);       // fh[NH-1] = V_{n-1}
            const u64 V = (w[o] << 1) | (w[o + 1] >> 63);
            const u64 Fp = det_hist_window<64 * NH - LG::LAGK, NH>(fh) ^ det_hist_window<64 * NH - LG::LAGT, NH>(fh);
            const uint32_t vh = (uint32_t)(V >> 32);
            const uint32_t last = (uint32_t)(w[o] >> 32) ^ (vh << (K - 1)) ^ (vh << (TAP - 1));
            return V == Fp && (int32_t)last >= 0 && n >= (u64)(NH + 1) && n < nfull;
        };
        static_assert(NH + 2 <= 5, "the two lanes below hold the words a flag needs");
        fo = one(0, first + 1);
        fe = one(1, first);
    };
    bool e0, o0, e1, o1;
    pair_flags(a0, h, base + 2 * lane, e0, o0);
    pair_flags(a1, a0, base + 128 + 2 * lane, e1, o1);
    me0 = __ballot(e0); mo0 = __ballot(o0); me1 = __ballot(e1); mo1 = __ballot(o1);
}

// The same for a half (128 words) in the INTERIOR of the stream, with what the half in front of it left behind: the cheap form
// (the form above spends 35 instructions per word and lane, most of them cross-lane traffic through LDS and the recomputation
// of every word's history; this one 25 and no LDS.  The kernel's time did not move -- 0.255-0.26 ms either way: it is bound by
// the memory system's rate for 4768 waves streaming 4768 separate 256 KiB stretches, where the grid-strided classification
// kernel of the two-kernel form reads at 5.7-6.3 TB/s -- but the instructions it no longer issues are energy.)
// Every lane forms the delayed words V of its own two words ONCE; a neighbour's V comes by DPP (wave_shr:1: lane l reads lane
// l - 1, lane 0 keeps the `old` operand, which is the previous half's lane 63 by V_READLANE): no LDS, no select.
struct DetHalfV { u64 vx, vy; };
__device__ __forceinline__ uint32_t det_shr1(const uint32_t cur, const uint32_t prev_half) {      // [l] = cur[l - 1]; [0] = prev_half[63]
    const uint32_t fill = (uint32_t)__builtin_amdgcn_readlane((int)prev_half, 63);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)cur, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t det_shr2(const uint32_t cur_shr1, const uint32_t prev_half) { // [l] = cur[l - 2]; [1] = prev_half[63]; [0] = prev_half[62]
    const uint32_t fill = (uint32_t)__builtin_amdgcn_readlane((int)prev_half, 62);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)cur_shr1, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 det_u64(uint32_t lo, uint32_t hi) { return (u64)lo | ((u64)hi << 32); }
// V of this lane's two words; prev_yhi = the top dword of the previous half's y word (this lane's: lane 63's is what counts)
__device__ __forceinline__ DetHalfV det_half_v(const u64x2 cur, const uint32_t prev_yhi) {
    const uint32_t yh = (uint32_t)(cur.y >> 32);
    const uint32_t before_x = det_shr1(yh, prev_yhi);            // top dword of the word in front of cur.x
    DetHalfV v;
    v.vx = (cur.x << 1) | (u64)(before_x >> 31);
    v.vy = (cur.y << 1) | (cur.x >> 63);
    return v;
}
template <int K>
__device__ __forceinline__ void det_half_flags(const u64x2 cur, const DetHalfV v, const DetHalfV pv, bool &fe, bool &fo) {
    typedef DetLag<K> LG;
    constexpr int NH = LG::NH, TAP = det_tap_of(K);
    // the neighbours' V: lane l - 1 (and for three history words lane l - 2)
    const uint32_t y1l = det_shr1((uint32_t)v.vy, (uint32_t)pv.vy), y1h = det_shr1((uint32_t)(v.vy >> 32), (uint32_t)(pv.vy >> 32));
    const uint32_t x1l = det_shr1((uint32_t)v.vx, (uint32_t)pv.vx), x1h = det_shr1((uint32_t)(v.vx >> 32), (uint32_t)(pv.vx >> 32));
    const u64 vy1 = det_u64(y1l, y1h), vx1 = det_u64(x1l, x1h);            // V_{n0-1}, V_{n0-2}
    u64 he[3] = {0, 0, 0}, ho[3] = {0, 0, 0};                              // histories of the even (x) and the odd (y) word, oldest first
    if constexpr (NH == 2) {
        he[0] = vx1; he[1] = vy1;
        ho[0] = vy1; ho[1] = v.vx;
    } else {
        const u64 vy2 = det_u64(det_shr2(y1l, (uint32_t)pv.vy), det_shr2(y1h, (uint32_t)(pv.vy >> 32)));      // V_{n0-3}
        he[0] = vy2; he[1] = vx1; he[2] = vy1;
        ho[0] = vx1; ho[1] = vy1; ho[2] = v.vx;
    }
    auto one = [&](const u64 (&fh)[3], const u64 V, const u64 w) -> bool {
        const u64 Fp = det_hist_window<64 * NH - LG::LAGK, NH>(fh) ^ det_hist_window<64 * NH - LG::LAGT, NH>(fh);
        const uint32_t vh = (uint32_t)(V >> 32);
        const uint32_t last = (uint32_t)(w >> 32) ^ (vh << (K - 1)) ^ (vh << (TAP - 1));
        return V == Fp && (int32_t)last >= 0;
    };
    fe = one(he, v.vx, cur.x);
    fo = one(ho, v.vy, cur.y);
}

// words i, i + 1 of a stream that is read once: non-temporal (the stream's last word may stand alone)
__device__ __forceinline__ u64x2 det_load2(const u64 *__restrict src, const u64 i, const u64 nwords) {
    if (i + 1 < nwords) return __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(src + i));
    return (u64x2){i < nwords ? src[i] : 0ull, 0ull};
}

template <int K>
__global__ void __launch_bounds__(256)
det_classify_kernel(const u64 *__restrict src, u64 nbits, u64 nwords, u64 *__restrict flags) {
    const unsigned lane = threadIdx.x & 63;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 nfull = nbits / 64;                    // words with all 64 bits inside the stream
    // lane t holds words base + 2t, base + 2t + 1 (one 16-byte load: 1 KiB per wave) and the same 128 words further; the four
    // words in front of the block sit in lanes 62, 63 of `h`.  The next block's loads are in flight while this one is judged.
    u64 base = ((((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6)) * 256;
    u64x2 n0 = {0, 0}, n1 = {0, 0}, nh = {0, 0};
    if (base < nwords) {
        n0 = det_load2(src, base + 2 * lane, nwords); n1 = det_load2(src, base + 128 + 2 * lane, nwords);
        if (lane >= 62 && base >= 128) nh = det_load2(src, base - 128 + 2 * lane, nwords);
    }
    for (; base < nwords; base += nwaves * 256) {
        const u64x2 a0 = n0, a1 = n1, h = nh;
        const u64 nb = base + nwaves * 256;
        if (nb < nwords) {
            n0 = det_load2(src, nb + 2 * lane, nwords); n1 = det_load2(src, nb + 128 + 2 * lane, nwords);
            if (lane >= 62) nh = det_load2(src, nb - 128 + 2 * lane, nwords);
        }
        u64 me0, mo0, me1, mo1;
        det_classify_256<K>(a0, a1, h, base, nfull, lane, me0, mo0, me1, mo1);
        if (lane == 0) {
            *reinterpret_cast<u64x2 *>(flags + (base >> 6)) = (u64x2){me0, mo0};
            if (base + 128 < nwords) *reinterpret_cast<u64x2 *>(flags + (base >> 6) + 2) = (u64x2){me1, mo1};
        }
    }
}

// the first word at or after w, below w1, whose flag is not set (w1 if there is none)
__device__ __forceinline__ u64 det_next_unflagged(const u64 *__restrict flags, u64 w, u64 w1) {
    for (u64 b = w >> 7; b <= (w1 - 1) >> 7; b++) {
        const u64x2 f = *reinterpret_cast<const u64x2 *>(flags + 2 * b);      // (a chunk's flags are one or two cache lines: the first read fetches them)
        u64 de = ~f.x, dn = ~f.y;                    // unset among the even / the odd words of block b
        if (b == (w >> 7)) {
            const unsigned r = (unsigned)(w & 127);
            const unsigned se = (r + 1) >> 1, so = r >> 1;       // even word 2 i >= r <=> i >= ceil(r / 2); odd 2 i + 1 >= r <=> i >= floor(r / 2)
            de = se >= 64 ? 0ull : de & (~0ull << se);
            dn &= ~0ull << so;
        }
        const u64 ce = de ? 2ull * (u64)__builtin_ctzll(de) : 1000ull, co = dn ? 2ull * (u64)__builtin_ctzll(dn) + 1 : 1000ull;
        const u64 c = ce < co ? ce : co;
        if (c < 1000ull) {
            const u64 cand = b * 128 + c;
            return cand < w1 ? cand : w1;
        }
    }
    return w1;
}

// what a lane carries from word to word in the sparse pass
struct DetCtx { DetAux a; bool clean; uint32_t cbit; int vrun; };     // vrun: clean words in a row just passed (saturates at NH)

// words [w0, w1) from state s: jumps over flagged stretches, the per-word machine elsewhere
template <int K, bool EMIT>
__device__ __forceinline__ void det_span_sparse(DetState &s, DetCtx &x, const u64 *__restrict src, const u64 *__restrict flags, u64 w0,
                                                u64 w1, u64 nbits, u64 *__restrict err, u64 *__restrict reload, DetCount &cnt) {
    typedef DetLag<K> LG;
    constexpr int NH = LG::NH, TAP = det_tap_of(K);
    constexpr uint32_t MASK = (uint32_t)((1ull << K) - 1ull);
    const u64 nfull = nbits / 64, nwords = (nbits + 63) / 64;
    u64 w = w0;
    // the words a jump lands on, fetched WITH the history in front of them (one round trip to memory per jump instead of one
    // per word: a lane's time is the latency of its dependent loads): the unflagged word j and the NH behind it
    u64 pre[NH + 1], pre0 = 1ull << 63;       // (w - pre0 is then huge for every real w)
    while (w < w1) {
        if (x.clean && x.vrun >= NH) {
            const u64 j = det_next_unflagged(flags, w, w1);
            if (j > w) {
                // every word of [w, j) is clean for a machine in this state: go to j with the history those words leave
                // (a flagged word has index >= NH + 1: the loads stay inside the stream)
                u64 ww[NH + 1];
#pragma unroll
                for (int q = 0; q <= NH; q++) ww[q] = src[j - 1 - (u64)NH + (u64)q];
#pragma unroll
                for (int q = 0; q <= NH; q++) pre[q] = src[j + (u64)q < nwords ? j + (u64)q : nwords - 1];
                pre0 = j;
#pragma unroll
                for (int q = 0; q < NH; q++) x.a.fh[q] = (ww[q + 1] << 1) | (ww[q] >> 63);
                x.a.n = NH;
                x.cbit = (uint32_t)(ww[NH] >> 63);
                w = j;
                continue;
            }
        }
        u64 word;
        if (w - pre0 <= (u64)NH) {
            word = pre[0];
#pragma unroll
            for (int q = 1; q <= NH; q++) word = (w - pre0 == (u64)q) ? pre[q] : word;
        } else {
            word = src[w];
        }
        u64 ew = 0, rw = 0;
        bool ok = false;
        if (x.clean && w < nfull) {
            const u64 V = (word << 1) | (u64)x.cbit;
            const u64 Fp = det_hist_window<64 * NH - LG::LAGK, NH>(x.a.fh) ^ det_hist_window<64 * NH - LG::LAGT, NH>(x.a.fh);
            const uint32_t vh = (uint32_t)(V >> 32);
            const uint32_t last = (uint32_t)(word >> 32) ^ (vh << (K - 1)) ^ (vh << (TAP - 1));
            ok = V == Fp && (int32_t)last >= 0;
            if (ok) {
#pragma unroll
                for (int i = 0; i + 1 < NH; i++) x.a.fh[i] = x.a.fh[i + 1];
                x.a.fh[NH - 1] = V;
                x.cbit = (uint32_t)(word >> 63);
                x.vrun = x.vrun < NH ? x.vrun + 1 : NH;
            }
        }
        if (!ok) {
            if (x.clean) {
                s.prbs = __builtin_bitreverse32((uint32_t)(x.a.fh[NH - 1] >> 32)) & MASK;
                s.err_sr = 0; s.reload_ctr = 0; s.bit_in = x.cbit;
            }
            det_core<K, EMIT>(s, x.a, w, word, nbits, ew, rw, cnt);
            x.cbit = s.bit_in;
            x.clean = s.reload_ctr == 0 && s.err_sr == 0 && x.a.n >= NH;
            x.vrun = 0;
        }
        if (EMIT) {                                  // (a clean word's outputs are zero: the arrays were zeroed)
            if (err && ew) err[w] = ew;
            if (reload && rw) reload[w] = rw;
        }
        w++;
    }
    if (x.clean) {
        s.prbs = __builtin_bitreverse32((uint32_t)(x.a.fh[NH - 1] >> 32)) & MASK;
        s.err_sr = 0; s.reload_ctr = 0; s.bit_in = x.cbit;
    }
}

// the speculative run of every chunk (det_chunk_kernel's mode 0) in the sparse form: one lane per chunk
template <int K>
__global__ void __launch_bounds__(256)
det_sparse_kernel(const u64 *__restrict src, const u64 *__restrict flags, u64 nbits, u64 nwords, u64 chunk_words, u64 warm_words,
                  u64 nchunks, DetState *__restrict spec, DetState *__restrict endst, DetCount *__restrict counts, u64 *__restrict err,
                  u64 *__restrict reload) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const u64 w0 = c * chunk_words;
    const u64 w1 = w0 + chunk_words < nwords ? w0 + chunk_words : nwords;
    DetState s = det_reset(K);
    DetCtx x;
    x.a.n = 0; x.clean = false; x.cbit = 0; x.vrun = 0;
    DetCount cnt = {0, 0, 0, 0}, dummy = {0, 0, 0, 0};
    const u64 ws = w0 > warm_words ? w0 - warm_words : 0;     // ws == 0: the whole prefix is run, the start is exact
    typedef DetLag<K> LGs;
    constexpr int NHs = LGs::NH;
    // The speculative start.  When the NH words in front of the chunk are flagged, a machine that was locked on them
    // arrives in the clean state those words define: take that state without running anything (the verification against
    // the predecessor's true end decides, as for every speculation).  Otherwise: the reset state `warm` bits earlier.
    u64 ww[NHs + 1];                                 // (fetched beside the flags, whatever they will say)
#pragma unroll
    for (int q = 0; q <= NHs; q++) ww[q] = w0 >= (u64)(NHs + 1) ? src[w0 - 1 - (u64)NHs + (u64)q] : 0ull;
    if (ws > 0 && w0 >= (u64)(2 * NHs + 2) && det_next_unflagged(flags, w0 - NHs, w0) == w0) {
#pragma unroll
        for (int q = 0; q < NHs; q++) x.a.fh[q] = (ww[q + 1] << 1) | (ww[q] >> 63);
        x.a.n = NHs; x.clean = true; x.vrun = NHs;
        x.cbit = (uint32_t)(ww[NHs] >> 63);
        s.prbs = __builtin_bitreverse32((uint32_t)(x.a.fh[NHs - 1] >> 32)) & (uint32_t)((1ull << K) - 1ull);
        s.err_sr = 0; s.reload_ctr = 0; s.bit_in = x.cbit;
    } else {
        det_span_sparse<K, false>(s, x, src, flags, ws, w0, nbits, nullptr, nullptr, dummy);
    }
    spec[c] = s;
    det_span_sparse<K, true>(s, x, src, flags, w0, w1, nbits, err, reload, cnt);
    endst[c] = s;
    counts[c] = cnt;
}

// Both passes in ONE kernel (the form a default call takes): a wave classifies the words of ITS 64 chunks (and the block in
// front of them) into LDS -- 64 chunks of 4 KiB are a contiguous 256 KiB stretch, streamed 2 KiB per step -- and its lanes
// then run their chunks from those flags.  The classification of one wave overlaps the serial part of the others on the same
// CU (latency-bound lanes beside a memory-bound stream), the flags never travel through memory, and the few words a lane
// goes back to were read by its own wave microseconds earlier.  Needs chunks of whole 128-word flag blocks and a warm-up of
// at most one; other geometries take the two kernels above.
constexpr int kDetFusedMaxChunkWords = 1024;
template <int K>
__global__ void __launch_bounds__(256)
det_fused_kernel(const u64 *__restrict src, u64 nbits, u64 nwords, u64 chunk_words, u64 warm_words, u64 nchunks,
                 DetState *__restrict spec, DetState *__restrict endst, DetCount *__restrict counts, u64 *__restrict err,
                 u64 *__restrict reload) {
    // 2 flag words per 128-word block of a wave's region (64 chunks + the block in front), + slack: DYNAMIC, sized by the chunk length
    // (round 5: chunks of up to 1024 words; a static array for the largest would cost the shorter ones two blocks per CU)
    extern __shared__ __attribute__((aligned(16))) u64 lflags_dyn[];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 c0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x - lane);                    // the wave's first chunk
    if (c0 >= nchunks) return;
    const u64 nfull = nbits / 64;
    const u64 r0 = c0 ? c0 * chunk_words - 128 : 0;                                        // region: one block in front of the first chunk ...
    const u64 r1e = (c0 + 64) * chunk_words;
    const u64 r1 = r1e < nwords ? r1e : nwords;                                            // ... to the end of the last
    u64 *const lf = lflags_dyn + (size_t)wv * (chunk_words + 8);
    {
        // two steps (4 KiB per wave) of loads in flight beyond the one being judged
        u64 base = r0;
        auto fetch = [&](const u64 b, u64x2 &x0, u64x2 &x1, u64x2 &xh) {
            x0 = (u64x2){0, 0}; x1 = x0; xh = x0;
            if (b < r1) {
                x0 = det_load2(src, b + 2 * lane, nwords); x1 = det_load2(src, b + 128 + 2 * lane, nwords);
                if (lane >= 62 && b >= 128) xh = det_load2(src, b - 128 + 2 * lane, nwords);
            }
        };
        u64x2 p0, p1, ph, q0, q1, qh;
        fetch(base, p0, p1, ph);
        fetch(base + 256, q0, q1, qh);
        DetHalfV pv = {0, 0};                        // the V words of the half in front (valid from the second step on)
        uint32_t pyh = 0;
        for (; base < r1; base += 256) {
            const u64x2 a0 = p0, a1 = p1, h = ph;
            p0 = q0; p1 = q1; ph = qh;
            fetch(base + 512, q0, q1, qh);
            u64 me0, mo0, me1, mo1;
            // interior steps (every word has its NH + 1 predecessors, all 256 words are whole words of the stream, and the
            // step in front left its V words): the cheap form; the region's first step and the stream's last: the general one
            const bool interior = base > r0 && base >= 256 && base + 256 <= nfull;
            DetHalfV v0 = det_half_v(a0, pyh);
            DetHalfV v1 = det_half_v(a1, (uint32_t)(a0.y >> 32));
            if (interior) {
                bool e0, o0, e1, o1;
                det_half_flags<K>(a0, v0, pv, e0, o0);
                det_half_flags<K>(a1, v1, v0, e1, o1);
                me0 = __ballot(e0); mo0 = __ballot(o0); me1 = __ballot(e1); mo1 = __ballot(o1);
            } else {
                det_classify_256<K>(a0, a1, h, base, nfull, lane, me0, mo0, me1, mo1);
            }
            pv = v1;
            pyh = (uint32_t)(a1.y >> 32);
            if (lane == 0) {
                const u64 i = (base - r0) >> 6;
                *reinterpret_cast<u64x2 *>(lf + i) = (u64x2){me0, mo0};
                *reinterpret_cast<u64x2 *>(lf + i + 2) = (u64x2){me1, mo1};
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u64 *const flags = lf - 2 * (r0 >> 7);     // flags + 2 b = block b of the stream
    const u64 c = c0 + lane;
    if (c >= nchunks) return;
    const u64 w0 = c * chunk_words;
    const u64 w1 = w0 + chunk_words < nwords ? w0 + chunk_words : nwords;
    DetState s = det_reset(K);
    DetCtx x;
    x.a.n = 0; x.clean = false; x.cbit = 0; x.vrun = 0;
    DetCount cnt = {0, 0, 0, 0}, dummy = {0, 0, 0, 0};
    const u64 ws = w0 > warm_words ? w0 - warm_words : 0;
    typedef DetLag<K> LGs;
    constexpr int NHs = LGs::NH;
    u64 ww[NHs + 1];
#pragma unroll
    for (int q = 0; q <= NHs; q++) ww[q] = w0 >= (u64)(NHs + 1) ? src[w0 - 1 - (u64)NHs + (u64)q] : 0ull;
    if (ws > 0 && w0 >= (u64)(2 * NHs + 2) && det_next_unflagged(flags, w0 - NHs, w0) == w0) {
#pragma unroll
        for (int q = 0; q < NHs; q++) x.a.fh[q] = (ww[q + 1] << 1) | (ww[q] >> 63);
        x.a.n = NHs; x.clean = true; x.vrun = NHs;
        x.cbit = (uint32_t)(ww[NHs] >> 63);
        s.prbs = __builtin_bitreverse32((uint32_t)(x.a.fh[NHs - 1] >> 32)) & (uint32_t)((1ull << K) - 1ull);
        s.err_sr = 0; s.reload_ctr = 0; s.bit_in = x.cbit;
    } else {
        det_span_sparse<K, false>(s, x, src, flags, ws, w0, nbits, nullptr, nullptr, dummy);
    }
    spec[c] = s;
    det_span_sparse<K, true>(s, x, src, flags, w0, w1, nbits, err, reload, cnt);
    endst[c] = s;
    counts[c] = cnt;
}

// mode 0: speculative run of every chunk.  mode 1: re-run of the chunks in `list` from end[c-1].
template <int K, bool OUT>
__global__ void __launch_bounds__(256, OUT ? 2 : 4)      // totals only: <= 128 registers, four waves per SIMD
det_chunk_kernel(int mode, const u64 *__restrict src, u64 nbits, u64 nwords, u64 chunk_words, u64 warm_words,
                 u64 nchunks, const unsigned *__restrict list, unsigned nlist, DetState *__restrict spec,
                 DetState *__restrict endst, DetCount *__restrict counts, u64 *__restrict err, u64 *__restrict reload,
                 int tiles_ok, const unsigned *__restrict nlist_dev) {
    __shared__ u64x2 tiles[4][64][9];             // [wave][row][16-byte piece], rows padded to 144 bytes
    if (nlist_dev) nlist = *nlist_dev < nlist ? *nlist_dev : nlist;       // mode 1 queued before the host knew the count
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    if (mode == 0) {
        // a wave whose 64 chunks are all full, lie behind a full warm-up and start on 128-byte boundaries
        const u64 c0 = idx - lane;
        if (tiles_ok && c0 + 64 <= nchunks && c0 * chunk_words >= warm_words && (c0 + 64) * chunk_words * 64 <= nbits) {
            const u64 c = idx, w0 = c * chunk_words;
            DetState s = det_reset(K);
            DetCount cnt = {0, 0, 0, 0}, dummy = {0, 0, 0, 0};
            u64x2 (*tile)[9] = tiles[threadIdx.x >> 6];
            det_span_tiled<K, false, false>(s, src, c0 * chunk_words - warm_words, chunk_words, warm_words, w0 - warm_words, nbits, nullptr,
                                     nullptr, dummy, tile, lane);
            spec[c] = s;
            det_span_tiled<K, true, OUT>(s, src, c0 * chunk_words, chunk_words, chunk_words, w0, nbits, err, reload, cnt, tile, lane);
            endst[c] = s;
            counts[c] = cnt;
            return;
        }
    }
    u64 c;
    if (mode == 0) {
        if (idx >= nchunks) return;
        c = idx;
    } else {
        if (idx >= nlist) return;
        c = list[idx];
    }
    const u64 w0 = c * chunk_words;
    const u64 w1 = w0 + chunk_words < nwords ? w0 + chunk_words : nwords;
    DetState s;
    DetCount cnt = {0, 0, 0, 0};
    if (mode == 0) {
        s = det_reset(K);
        const u64 ws = w0 > warm_words ? w0 - warm_words : 0;     // ws == 0: the whole prefix is run, the start is exact
        DetCount dummy = {0, 0, 0, 0};
        det_span<K, false>(s, src, ws, w0, nbits, nullptr, nullptr, dummy);
    } else {
        // Re-run from the true end of the predecessor -- beside the speculative trajectory of the first pass, word for word,
        // until the two states are equal: from there on the chunk is what the first pass made of it (the detector's future
        // depends on its state alone), so only the prefix is run again and the chunk's counters are corrected by the
        // difference of the two prefixes.  Both trajectories are locked on the same stream, so they meet within a word or
        // two of the speculative one's last reload -- microseconds instead of the 0.2 ms a whole 32768-bit chunk takes one
        // lane (which every call with a single inconsistent chunk used to wait for).
        s = endst[c - 1];
        DetState t = spec[c];                       // what counts[c] / endst[c] were computed from
        DetCount ct = {0, 0, 0, 0};
        DetAux as, at;
        as.n = 0; at.n = 0;
        const DetState s0 = s;
        bool met = false;
        for (u64 w = w0; w < w1; w++) {
            const u64 word = src[w];
            u64 ew, rw, ew2, rw2;
            det_core<K, true>(s, as, w, word, nbits, ew, rw, cnt);
            det_core<K, true>(t, at, w, word, nbits, ew2, rw2, ct);
            if (err) err[w] = ew;
            if (reload) reload[w] = rw;
            if (det_equal(s, t)) { met = w + 1 < w1; break; }
        }
        spec[c] = s0;
        if (met) {
            DetCount old = counts[c];
            old.err_synced += cnt.err_synced - ct.err_synced; old.err_raw += cnt.err_raw - ct.err_raw;
            old.reload_clocks += cnt.reload_clocks - ct.reload_clocks; old.resyncs += cnt.resyncs - ct.resyncs;
            counts[c] = old;                        // (endst[c] stands)
        } else {
            endst[c] = s;                           // ran to the chunk's end (or met on its last word: the same thing)
            counts[c] = cnt;
        }
        return;
    }
    spec[c] = s;
    det_span<K, true>(s, src, w0, w1, nbits, err, reload, cnt);
    endst[c] = s;
    counts[c] = cnt;
}

// (list <- chunks whose speculative start differs from the end of their predecessor: in det_reduce_kernel)

// exact serial continuation from chunk c0 to the end, one lane (guard for streams on which the
// speculation keeps failing; never taken in the tests' regimes)
template <int K>
__global__ void __launch_bounds__(64)
det_serial_kernel(const u64 *__restrict src, u64 nbits, u64 nwords, u64 chunk_words, u64 nchunks, u64 c0,
                  DetState *__restrict spec, DetState *__restrict endst, DetCount *__restrict counts, u64 *__restrict err,
                  u64 *__restrict reload) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DetState s = endst[c0 - 1];
    for (u64 c = c0; c < nchunks; c++) {
        if (c > c0 && det_equal(spec[c], s)) return;            // from here on the speculative results stand...
        const u64 w0 = c * chunk_words;
        const u64 w1 = w0 + chunk_words < nwords ? w0 + chunk_words : nwords;
        DetCount cnt = {0, 0, 0, 0};
        spec[c] = s;
        det_span<K, true>(s, src, w0, w1, nbits, err, reload, cnt);
        endst[c] = s;
        counts[c] = cnt;
    }
}

__global__ void __launch_bounds__(256)
det_reduce_kernel(u64 nchunks, const DetCount *__restrict counts, u64 *__restrict totals, const DetState *__restrict spec,
                  const DetState *__restrict endst, unsigned *__restrict list, unsigned *__restrict nlist,
                  const unsigned *__restrict skip_if_zero) {
    if (skip_if_zero && *skip_if_zero == 0) return;       // the pass before this one found the chain consistent
    // verification of the chain (chunk c's speculative start against chunk c-1's end) and the totals in ONE pass: one
    // launch less on the only round trip of a consistent run; a chunk's four counters as two 16-byte loads
    typedef u64 u64x2v __attribute__((ext_vector_type(2)));
    u64 v[4] = {0, 0, 0, 0};
    for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (u64)gridDim.x * blockDim.x) {
        const u64x2v lo = reinterpret_cast<const u64x2v *>(counts + c)[0], hi = reinterpret_cast<const u64x2v *>(counts + c)[1];
        v[0] += lo.x; v[1] += lo.y; v[2] += hi.x; v[3] += hi.y;
        if (c >= 1 && !det_equal(spec[c], endst[c - 1])) list[atomicAdd(nlist, 1u)] = (unsigned)c;
    }
    // wave sums, then ONE atomic per block and counter (the atomics on four addresses were the kernel's cost: 0.05-0.12 ms)
    __shared__ u64 part[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u64 x = v[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const u64 x = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (x) atomicAdd(&totals[threadIdx.x], x);
    }
}

// The call's device workspace and its pinned read-back buffer, kept between calls (grow-only, one per concurrent call and
// device).  With a stream-ordered allocation per call the pool gave the 22 MB of a 1e10-bit call back to the driver at every
// synchronisation and fetched them again at the next call, and the totals came back through a pageable bounce buffer: 0.26 ms
// of host time around 0.49 ms of kernels.
// zeroed_at / ev_zero: the 16 tail words at that offset of d are zero once ev_zero has happened -- a call that ends the plain way
// zeroes them for the next one BEHIND its read-back instead of the next call doing so in front of its first kernel
struct DetWorkspace { int dev; char *d; size_t cap; u64 *h; bool busy; hipEvent_t ev_copy, ev_zero; size_t zeroed_at; bool zeroed; };
static std::mutex g_det_ws_mu;
static std::vector<DetWorkspace> g_det_ws;

static int det_ws_acquire(size_t need, int *slot) {
    int dev = 0;
    BBB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_det_ws_mu);
    int found = -1;
    for (size_t i = 0; i < g_det_ws.size(); i++)
        if (!g_det_ws[i].busy && g_det_ws[i].dev == dev) { found = (int)i; break; }
    if (found < 0) {
        DetWorkspace w{dev, nullptr, 0, nullptr, false, nullptr, nullptr, 0, false};
        BBB_HIP(hipHostMalloc((void **)&w.h, 16 * sizeof(u64), hipHostMallocDefault));
        BBB_HIP(hipEventCreateWithFlags(&w.ev_copy, hipEventDisableTiming));
        BBB_HIP(hipEventCreateWithFlags(&w.ev_zero, hipEventDisableTiming));
        g_det_ws.push_back(w);
        found = (int)g_det_ws.size() - 1;
    }
    DetWorkspace &w = g_det_ws[found];
    if (w.cap < need) {
        if (w.d) (void)hipFree(w.d);
        w.d = nullptr; w.cap = 0; w.zeroed = false;
        BBB_HIP(hipMalloc((void **)&w.d, need));
        w.cap = need;
    }
    w.busy = true;
    *slot = found;
    return BBB_OK;
}

static void det_ws_release(int slot) {
    std::lock_guard<std::mutex> g(g_det_ws_mu);
    DetWorkspace &w = g_det_ws[slot];
    if (w.cap > (size_t)256 << 20) {        // (a very long stream's workspace is not kept: 256 MiB = 1.2e11 bits at the default chunking)
        (void)hipFree(w.d);
        w.d = nullptr; w.cap = 0; w.zeroed = false;
    }
    w.busy = false;
}

template <int K>
static int detector_stream_k(const u64 *src, u64 nbits, u64 *err, u64 *reload, bbb_detector_stats *stats, u64 chunk_bits,
                             u64 warm_bits, hipStream_t st) {
    const u64 nwords = (nbits + 63) / 64;
    const u64 chunk_words = chunk_bits / 64, warm_words = (warm_bits + 63) / 64;
    const u64 nchunks = (nwords + chunk_words - 1) / chunk_words;
    if (nchunks > 0x7fffffffull) return fail(BBB_EINVAL, "too many chunks; raise chunk_bits");
    // one stream-ordered allocation: spec | endst | counts | list | nlist + totals
    constexpr bool kSparse = DetLag<K>::OK;             // (K = 20: ten history words -- the dense pass)
    const size_t o_spec = 0, o_end = o_spec + nchunks * sizeof(DetState), o_cnt = o_end + nchunks * sizeof(DetState),
                 o_list = o_cnt + nchunks * sizeof(DetCount), o_tail = (o_list + 2 * nchunks * sizeof(unsigned) + 7) & ~(size_t)7,
                 o_flags = (o_tail + 16 * sizeof(u64) + 15) & ~(size_t)15, total = o_flags + (kSparse ? ((nwords + 255) / 256) * 4 * sizeof(u64) : 0);
    int ws_slot = -1;
    int rc_ws = det_ws_acquire(total, &ws_slot);
    if (rc_ws) return rc_ws;
    char *ws;
    u64 *hpin;
    hipEvent_t ev_copy, ev_zero;
    bool tail_zeroed, zero_pending;
    {
        std::lock_guard<std::mutex> g(g_det_ws_mu);
        DetWorkspace &w = g_det_ws[ws_slot];
        ws = w.d; hpin = w.h; ev_copy = w.ev_copy; ev_zero = w.ev_zero;
        tail_zeroed = w.zeroed && w.zeroed_at == o_tail;
        zero_pending = w.zeroed;
        w.zeroed = false;                     // (until this call has ended the way that leaves them zeroed again)
    }
    // The previous user of this workspace left a 128-byte memset queued behind its read-back (ev_zero) and handed the workspace back
    // without waiting for it.  A call with ANOTHER layout (other nbits or chunking: the old tail lies inside this call's chunk
    // records), on another stream, must not start before that memset has run -- round 4's advisor: a late memset zeroing 16 words
    // of the new call's records.  Whoever finds the flag set waits for the event first (same stream: nothing to wait for).
    if (zero_pending) (void)hipStreamWaitEvent(st, ev_zero, 0);
    DetState *spec = (DetState *)(ws + o_spec), *endst = (DetState *)(ws + o_end);
    DetCount *counts = (DetCount *)(ws + o_cnt);
    unsigned *list = (unsigned *)(ws + o_list), *list2 = list + nchunks;
    u64 *tail = (u64 *)(ws + o_tail), *tail2 = tail + 8;       // tail[0..3] totals, low half of tail[4] = number of bad chunks
    unsigned *nlist = (unsigned *)(tail + 4), *nlist2 = (unsigned *)(tail2 + 4);
    // (every path below has synchronised the stream before it calls this, or failed in a HIP call and does so here)
    auto cleanup = [&]() { (void)hipStreamSynchronize(st); det_ws_release(ws_slot); };
    const unsigned grid = (unsigned)((nchunks + 255) / 256);
    // (the reduce kernel's time is its atomics -- one per block and counter: 512 blocks 11 us, 128 blocks 6 us, 64 too few to
    // stream 20 MB of chunk records; BBB_DET_RGRID: A/B timing, -DBBB_EXPERIMENTS only)
    const unsigned rmax = (unsigned)env_knob("BBB_DET_RGRID", 128);
    const unsigned rgrid = grid < rmax ? grid : rmax;
    // cooperative 128-byte loads need chunks and warm-up in whole 16-word rows on 16-byte aligned data
    const int tiles_ok = chunk_words % 16 == 0 && warm_words % 16 == 0 && warm_words > 0 && ((uintptr_t)src & 15) == 0;
    static const bool dense = env_knob("BBB_DET_DENSE", 0) != 0;          // (A/B timing of the two forms; -DBBB_EXPERIMENTS only)
    const bool sparse = kSparse && !dense && ((uintptr_t)src & 15) == 0;
    if (sparse) {
        if constexpr (kSparse) {
            u64 *flags = (u64 *)(ws + o_flags);
            if (err) (void)hipMemsetAsync(err, 0, nwords * sizeof(u64), st);
            if (reload) (void)hipMemsetAsync(reload, 0, nwords * sizeof(u64), st);
            const u64 nblk = (nwords + 255) / 256;                        // one wave per 256 words, four waves per block
            const unsigned cgrid = (unsigned)((nblk + 3) / 4 < 8192 ? (nblk + 3) / 4 : 8192);
            static const bool two_kernels = env_knob("BBB_DET_TWO_KERNELS", 0) != 0;      // (A/B timing; -DBBB_EXPERIMENTS only)
            if (chunk_words % 128 == 0 && chunk_words <= (u64)kDetFusedMaxChunkWords && warm_words <= 128 && !two_kernels) {
                hipLaunchKernelGGL(det_fused_kernel<K>, dim3(grid), dim3(256), 4 * (chunk_words + 8) * sizeof(u64), st, src, nbits, nwords, chunk_words,
                                   warm_words, nchunks, spec, endst, counts, err, reload);
            } else {
                hipLaunchKernelGGL(det_classify_kernel<K>, dim3(cgrid), dim3(256), 0, st, src, nbits, nwords, flags);
                hipLaunchKernelGGL(det_sparse_kernel<K>, dim3(grid), dim3(256), 0, st, src, (const u64 *)flags, nbits, nwords, chunk_words,
                                   warm_words, nchunks, spec, endst, counts, err, reload);
            }
        }
    } else if (err || reload)
        hipLaunchKernelGGL((det_chunk_kernel<K, true>), dim3(grid), dim3(256), 0, st, 0, src, nbits, nwords, chunk_words, warm_words,
                           nchunks, (const unsigned *)nullptr, 0u, spec, endst, counts, err, reload, tiles_ok, (const unsigned *)nullptr);
    else
        hipLaunchKernelGGL((det_chunk_kernel<K, false>), dim3(grid), dim3(256), 0, st, 0, src, nbits, nwords, chunk_words, warm_words,
                           nchunks, (const unsigned *)nullptr, 0u, spec, endst, counts, err, reload, tiles_ok, (const unsigned *)nullptr);
    u64 rerun = 0, passes = 0;
    bool serial = false;
    u64 h[5] = {0, 0, 0, 0, 0};
    if (env_knob("BBB_DET_ONE_TRIP", 1)) {
        // ONE round trip for the two common outcomes: the chain is consistent as speculated, or it is after the few
        // inconsistent chunks have been run again from their predecessors' end states.  Verify + totals (pass 1), a re-run
        // of at most kSpec listed chunks whose count the device reads itself, verify + totals again (pass 2, which returns
        // at once when pass 1 found nothing) -- all queued before the host looks.
        constexpr unsigned kSpec = 4096;
        u64 *const hh = hpin;
        if (!tail_zeroed) (void)hipMemsetAsync(tail, 0, 16 * sizeof(u64), st);      // (else: zeroed behind the previous call's read-back; waited for above)
        hipLaunchKernelGGL(det_reduce_kernel, dim3(rgrid), dim3(256), 0, st, nchunks, counts, tail, (const DetState *)spec,
                           (const DetState *)endst, list, nlist, (const unsigned *)nullptr);
        // The repair stages are queued blind only for the dense pass, whose reset-state speculation leaves a few dozen
        // inconsistent chunks on every noisy stream.  The sparse pass starts a chunk from the clean state its flags imply and
        // is consistent as speculated unless an error or a reload sits right on a chunk boundary: there the two idle launches
        // (10 us of a 0.3 ms call) cost more than the second round trip they save, so the host looks first.
        if (!sparse) {
            hipLaunchKernelGGL((det_chunk_kernel<K, true>), dim3(kSpec / 256), dim3(256), 0, st, 1, src, nbits, nwords, chunk_words, warm_words,
                               nchunks, (const unsigned *)list, kSpec, spec, endst, counts, err, reload, 0, (const unsigned *)nlist);
            hipLaunchKernelGGL(det_reduce_kernel, dim3(rgrid), dim3(256), 0, st, nchunks, counts, tail2, (const DetState *)spec,
                               (const DetState *)endst, list2, nlist2, (const unsigned *)nlist);
        }
        hipError_t e = hipMemcpyAsync(hh, tail, 16 * sizeof(u64), hipMemcpyDeviceToHost, st);
        // The host waits for the read-back only; the zeroing of the tail words for the NEXT call is queued behind it and runs
        // while the host returns (5 us of a 0.29 ms call that were in front of the first kernel)
        bool zero_queued = false;
        if (e == hipSuccess && sparse) {
            e = hipEventRecord(ev_copy, st);
            if (e == hipSuccess) e = hipMemsetAsync(tail, 0, 16 * sizeof(u64), st);
            if (e == hipSuccess) e = hipEventRecord(ev_zero, st);
            zero_queued = e == hipSuccess;
            if (e == hipSuccess) e = hipEventSynchronize(ev_copy);
        } else if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { cleanup(); BBB_HIP(e); }
        const unsigned nbad1 = (unsigned)(hh[4] & 0xffffffffull), nbad2 = (unsigned)(hh[12] & 0xffffffffull);
        bool settled = false;
        if (!nbad1) {
            for (int i = 0; i < 5; i++) h[i] = hh[i];
            settled = true;
            if (zero_queued) {
                // (nothing of this call but that memset is in flight: the workspace goes back without a stream synchronisation)
                {
                    std::lock_guard<std::mutex> g(g_det_ws_mu);
                    g_det_ws[ws_slot].zeroed = true; g_det_ws[ws_slot].zeroed_at = o_tail;
                }
                det_ws_release(ws_slot);
                BBB_HIP(hipGetLastError());
                if (stats) {
                    stats->bits = nbits; stats->errors = h[0]; stats->errors_raw = h[1]; stats->reload_clocks = h[2];
                    stats->resyncs = h[3]; stats->chunks = nchunks; stats->chunks_rerun = 0; stats->serial_fallback = 0;
                }
                return BBB_OK;
            }
        }
        else if (sparse) {
            // second trip: the listed chunks again from their predecessors' ends, verify + totals, look again
            const unsigned nb = nbad1 < kSpec ? nbad1 : kSpec;
            hipLaunchKernelGGL((det_chunk_kernel<K, true>), dim3((nb + 255) / 256), dim3(256), 0, st, 1, src, nbits, nwords, chunk_words,
                               warm_words, nchunks, (const unsigned *)list, nb, spec, endst, counts, err, reload, 0, (const unsigned *)nullptr);
            hipLaunchKernelGGL(det_reduce_kernel, dim3(rgrid), dim3(256), 0, st, nchunks, counts, tail2, (const DetState *)spec,
                               (const DetState *)endst, list2, nlist2, (const unsigned *)nullptr);
            e = hipMemcpyAsync(hh, tail, 16 * sizeof(u64), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { cleanup(); BBB_HIP(e); }
            rerun += nb;
            passes = 1;
            if (nbad1 <= kSpec && !(unsigned)(hh[12] & 0xffffffffull)) { for (int i = 0; i < 5; i++) h[i] = hh[8 + i]; settled = true; }
        } else {
            rerun += nbad1 < kSpec ? nbad1 : kSpec;
            passes = 1;
            if (nbad1 <= kSpec && !nbad2) { for (int i = 0; i < 5; i++) h[i] = hh[8 + i]; settled = true; }
        }
        if (settled) {
            cleanup();
            BBB_HIP(hipGetLastError());
            if (stats) {
                stats->bits = nbits; stats->errors = h[0]; stats->errors_raw = h[1]; stats->reload_clocks = h[2];
                stats->resyncs = h[3]; stats->chunks = nchunks; stats->chunks_rerun = rerun; stats->serial_fallback = 0;
            }
            return BBB_OK;
        }
    }
    for (;;) {
        // verify, and reduce at once in the hope that the chain is already consistent: one round trip
        (void)hipMemsetAsync(tail, 0, 5 * sizeof(u64), st);
        hipLaunchKernelGGL(det_reduce_kernel, dim3(rgrid), dim3(256), 0, st, nchunks, counts, tail, (const DetState *)spec,
                           (const DetState *)endst, list, nlist, (const unsigned *)nullptr);
        hipError_t e = hipMemcpyAsync(h, tail, sizeof h, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { cleanup(); BBB_HIP(e); }
        const unsigned nbad = (unsigned)(h[4] & 0xffffffffull);
        if (!nbad) break;
        passes++;
        if (passes > 32) {
            // the speculation does not settle on this stream: continue serially from the first bad chunk
            std::vector<unsigned> hb(nbad);
            e = hipMemcpy(hb.data(), list, nbad * sizeof(unsigned), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { cleanup(); BBB_HIP(e); }
            unsigned c0 = hb[0];
            for (unsigned x : hb) c0 = x < c0 ? x : c0;
            hipLaunchKernelGGL(det_serial_kernel<K>, dim3(1), dim3(64), 0, st, src, nbits, nwords, chunk_words, nchunks, (u64)c0,
                               spec, endst, counts, err, reload);
            serial = true;
            passes = 0;                     // the serial kernel stops where the chain is consistent again: verify anew
            rerun += 1;
            continue;
        }
        rerun += nbad;
        hipLaunchKernelGGL((det_chunk_kernel<K, true>), dim3((nbad + 255) / 256), dim3(256), 0, st, 1, src, nbits, nwords, chunk_words,
                           warm_words, nchunks, (const unsigned *)list, nbad, spec, endst, counts, err, reload, 0, (const unsigned *)nullptr);
    }
    cleanup();
    BBB_HIP(hipGetLastError());
    if (stats) {
        stats->bits = nbits;
        stats->errors = h[0];
        stats->errors_raw = h[1];
        stats->reload_clocks = h[2];
        stats->resyncs = h[3];
        stats->chunks = nchunks;
        stats->chunks_rerun = rerun;
        stats->serial_fallback = serial ? 1 : 0;
    }
    return BBB_OK;
}

int prbs_detector_stream_launch(int k, const uint64_t *src, uint64_t nbits, uint64_t *err, uint64_t *reload,
                                bbb_detector_stats *stats, uint64_t chunk_bits, uint64_t warm_bits, hipStream_t st) {
    if (!det_tap_of(k)) return fail(BBB_EINVAL, "k=" + std::to_string(k) + " invalid for PRBS");
    if (chunk_bits == 0) {
        // about four waves of lanes per SIMD, but chunks of 4096 ... 32768 bits: the warm-up before every chunk is
        // 1024 bits of extra work, and short inputs should still fill the device.  (Sizing the chunks so that they fill
        // whole generations of resident lanes -- 2 x 38912-bit chunks per lane slot at 1e10 bits instead of 2.33 x 32768 --
        // changed nothing measurable: 0.455-0.46 ms either way.)
        const uint64_t want = (nbits / 262144 + 127) / 128 * 128;
        chunk_bits = want < 4096 ? 4096 : (want > 32768 ? 32768 : want);
        if (want > 32768) {
            // Long streams (round 5; profiles/r05_det_chunk_sweep.log): the fused kernel takes chunks of 512 ... 1024 words, a block
            // four waves x 64 chunks, and all blocks are resident at once -- the kernel then takes as long as the CU with the MOST
            // blocks, so the chunk length is chosen for the smallest (blocks on the fullest CU) x (words per chunk); ties go to the
            // longer chunk (fewer chunk records, fewer speculative starts).  1e10 bits: 640 words (954 blocks on 256 CUs: 4 x 640)
            // where round 4's fixed 512 gave 1192 blocks (5 x 512): 0.277 -> 0.258 ms per call.
            int dev = 0, ncu = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
            const uint64_t nwords = (nbits + 63) / 64;
            uint64_t best_cost = ~0ull, best_cw = 512;
            for (uint64_t cw = 512; cw <= (uint64_t)kDetFusedMaxChunkWords; cw += 128) {
                const uint64_t nblocks = ((nwords + cw - 1) / cw + 255) / 256;
                const uint64_t cost = ((nblocks + (uint64_t)ncu - 1) / (uint64_t)ncu) * cw;
                if (cost <= best_cost) { best_cost = cost; best_cw = cw; }
            }
            chunk_bits = best_cw * 64;
        }
    }
    if (warm_bits == 0) warm_bits = 1024;
    if (chunk_bits % 64) return fail(BBB_EINVAL, "chunk_bits must be a multiple of 64");
    if (nbits == 0) {
        if (stats) *stats = bbb_detector_stats{};
        return BBB_OK;
    }
    const u64 *s = reinterpret_cast<const u64 *>(src);
    u64 *e = reinterpret_cast<u64 *>(err), *r = reinterpret_cast<u64 *>(reload);
    switch (k) {
    case 7: return detector_stream_k<7>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    case 9: return detector_stream_k<9>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    case 11: return detector_stream_k<11>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    case 15: return detector_stream_k<15>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    case 20: return detector_stream_k<20>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    case 23: return detector_stream_k<23>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    default: return detector_stream_k<31>(s, nbits, e, r, stats, chunk_bits, warm_bits, st);
    }
}

}  // namespace bbb
