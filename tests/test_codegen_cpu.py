"""The generated straight-line LUTOPT/CLT network and the bit-matrix helpers, compiled for the
HOST with g++ (V_BITOP3 emulated by its truth table) and checked against the oracle.  This
exercises the exact text the HIP kernels include, without a GPU."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

HARNESS = r"""
#include <cstdint>
#include <cstring>
#define __device__
#define __forceinline__ inline
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, unsigned tt) {
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) if ((tt >> i) & 1) { uint32_t m = ~0u; m &= (i & 4) ? a : ~a; m &= (i & 2) ? b : ~b; m &= (i & 1) ? c : ~c; r |= m; }
  return r;
}
#define BBB_ACC_WRITE(dst, src) ((dst) = (src))
#define BBB_ACC_READ(dst, src) ((dst) = (src))
#include "GEN_INC"
#include "bitslice_util.hpp"
extern "C" void step(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[NN], B[NN], Cn[LOGN]; memcpy(A, a, sizeof A); STEPFN(A, B, Cn); memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
#ifdef PARKFN
// the variant with explicit AGPR placement: planes listed in LUTOPT256_PARKED travel in pa / pb
extern "C" void step_parked(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[NN], PA[NN], B[NN], PB[NN], Cn[LOGN];
  memcpy(A, a, sizeof A); memcpy(PA, a, sizeof PA);
  for (int i = 0; i < LUTOPT256_NPARKED; i++) A[LUTOPT256_PARKED[i]] = 0xdeadbeefu;      // must not be read
  PARKFN(A, PA, B, PB, Cn);
  for (int i = 0; i < LUTOPT256_NPARKED; i++) B[LUTOPT256_PARKED[i]] = PB[LUTOPT256_PARKED[i]];
  memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
// the second placement (budget 230, the PLANES kernel's): lutopt256_step_parked_hi with its own parked set
extern "C" void step_parked_hi(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[NN], PA[NN], B[NN], PB[NN], Cn[LOGN];
  memcpy(A, a, sizeof A); memcpy(PA, a, sizeof PA);
  for (int i = 0; i < LUTOPT256_NPARKED_HI; i++) A[LUTOPT256_PARKED_HI[i]] = 0xdeadbeefu;
  lutopt256_step_parked_hi(A, PA, B, PB, Cn);
  for (int i = 0; i < LUTOPT256_NPARKED_HI; i++) B[LUTOPT256_PARKED_HI[i]] = PB[LUTOPT256_PARKED_HI[i]];
  memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
#endif
extern "C" void step_new(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[NN], B[NN], Cn[LOGN]; memcpy(A, a, sizeof A); STEPNEWFN(A, B, Cn); memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
extern "C" void advance(const uint32_t* a, uint32_t* b) {
  uint32_t A[NN], B[NN]; memcpy(A, a, sizeof A); ADVFN(A, B); memcpy(b, B, sizeof B);
}
extern "C" void t32(uint32_t* q) { uint32_t Q[32]; memcpy(Q, q, sizeof Q); bbb::transpose32(Q); memcpy(q, Q, sizeof Q); }
extern "C" unsigned long long gidx(unsigned long long w, unsigned l, unsigned j) { return bbb::gen_index(w, l, j); }
"""


def build(tmp_path, n):
    inc = tmp_path / f"lutopt{n}_gen.inc"
    subprocess.check_call([sys.executable, str(ROOT / "basebandboard_amd" / "gen_lutopt_kernel.py"),
                           str(ROOT / "basebandboard_amd" / "data" / f"lutopt_{n}.taps"), str(inc)])
    src = tmp_path / f"h{n}.cpp"
    src.write_text(HARNESS.replace("GEN_INC", inc.name))
    so = tmp_path / f"h{n}.so"
    logn = n.bit_length() - 1
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-DNN={n}", f"-DLOGN={logn}",
                           f"-DSTEPFN=lutopt{n}_step", f"-DADVFN=lutopt{n}_advance", f"-DSTEPNEWFN=lutopt{n}_step_new", *([f"-DPARKFN=lutopt{n}_step_parked"] if n == 256 else []), f"-I{tmp_path}", f"-I{ROOT / 'basebandboard_amd' / 'csrc'}",
                           str(src), "-o", str(so)])
    return C.CDLL(str(so))


@pytest.mark.parametrize("n", (32, 256))
def test_generated_network_matches_oracle(oracle, tmp_path, n):
    lib = build(tmp_path, n)
    logn = n.bit_length() - 1
    m = oracle.Lutopt(path=oracle.data_path(n))
    rng = np.random.default_rng(n)
    states = [int.from_bytes(rng.bytes(n // 8), "little") for _ in range(32)]
    states[0] = 1
    states[1] = (1 << n) - 1
    a = np.zeros(n, dtype=np.uint32)
    for g, s in enumerate(states):
        for p in range(n):
            if (s >> p) & 1:
                a[p] |= np.uint32(1 << g)
    b = np.zeros(n, dtype=np.uint32)
    cnt = np.zeros(logn, dtype=np.uint32)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
    b2 = np.zeros(n, dtype=np.uint32)
    for _ in range(12):
        lib.step(P(a), P(b), P(cnt))          # b = A a, cnt = sample of the OLD state a
        lib.advance(P(a), P(b2))
        assert np.array_equal(b, b2)
        if n == 256:
            b3, cnt3 = np.zeros(n, dtype=np.uint32), np.zeros(logn, dtype=np.uint32)
            lib.step_parked(P(a), P(b3), P(cnt3))
            assert np.array_equal(b, b3) and np.array_equal(cnt, cnt3)
            b5, cnt5 = np.zeros(n, dtype=np.uint32), np.zeros(logn, dtype=np.uint32)
            lib.step_parked_hi(P(a), P(b5), P(cnt5))
            assert np.array_equal(b, b5) and np.array_equal(cnt, cnt5)
        b4, cnt4 = np.zeros(n, dtype=np.uint32), np.zeros(logn, dtype=np.uint32)
        lib.step_new(P(a), P(b4), P(cnt4))    # cnt4 = sample of the NEW state
        assert np.array_equal(b, b4)
        for g in range(32):
            v = sum(((int(cnt[q]) >> g) & 1) << q for q in range(logn))
            v = v - n if v >= n // 2 else v
            assert v == m.clt_wrap(m.clt_tree(states[g]))
            states[g] = m.step_int(states[g])
            v = sum(((int(cnt4[q]) >> g) & 1) << q for q in range(logn))
            v = v - n if v >= n // 2 else v
            assert v == m.clt_wrap(m.clt_tree(states[g]))
            got = sum(((int(b[p]) >> g) & 1) << p for p in range(n))
            assert got == states[g]
        a, b = b.copy(), a


def test_committed_generated_file_is_current(tmp_path):
    """csrc/gen/lutopt256_gen.inc (built by build()) must be what the generator emits today."""
    cur = ROOT / "basebandboard_amd" / "csrc" / "gen" / "lutopt256_gen.inc"
    if not cur.exists():
        pytest.skip("not built yet")
    out = tmp_path / "x.inc"
    subprocess.check_call([sys.executable, str(ROOT / "basebandboard_amd" / "gen_lutopt_kernel.py"),
                           str(ROOT / "basebandboard_amd" / "data" / "lutopt_256.taps"), str(out)])
    assert out.read_text() == cur.read_text()


def test_transpose32_and_gen_index(tmp_path):
    lib = build(tmp_path, 32)
    rng = np.random.default_rng(3)
    q = rng.integers(0, 2**32, size=32, dtype=np.uint64).astype(np.uint32)
    orig = q.copy()
    lib.t32(q.ctypes.data_as(C.POINTER(C.c_uint32)))
    for i in range(32):
        for j in range(32):
            assert (int(q[i]) >> j) & 1 == (int(orig[j]) >> i) & 1
    lib.gidx.restype = C.c_ulonglong
    lib.gidx.argtypes = [C.c_ulonglong, C.c_uint, C.c_uint]
    seen = {lib.gidx(w, l, j) for w in range(3) for l in range(64) for j in range(32)}
    assert seen == set(range(3 * 2048))
    assert lib.gidx(2, 5, 7) == (2 * 32 + 7) * 64 + 5


HARNESS512 = r"""
#include <cstdint>
#include <cstring>
#define __device__
#define __forceinline__ inline
#define BBB_ACC_WRITE(dst, src) ((dst) = (src))
#define BBB_ACC_READ(dst, src) ((dst) = (src))
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, unsigned tt) {
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) if ((tt >> i) & 1) { uint32_t m = ~0u; m &= (i & 4) ? a : ~a; m &= (i & 2) ? b : ~b; m &= (i & 1) ? c : ~c; r |= m; }
  return r;
}
// V_PERM_B32: the 8 source bytes are {hi, lo} (lo = bytes 0..3); selector byte 0x0c yields 0x00
static inline uint32_t __builtin_amdgcn_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  const uint64_t in = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) { const unsigned s = (sel >> (8 * i)) & 0xff; if (s < 8) r |= (uint32_t)((in >> (8 * s)) & 0xff) << (8 * i); }
  return r;
}
#include "GEN_INC"
extern "C" void step_new(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[256], B[256], Cn[9]; memcpy(A, a, sizeof A); lutopt512p_step_new(A, B, Cn); memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
// the generator-placed form: the registers in LUTOPT512_PARKED travel in pa / pb
extern "C" void step_new_parked(const uint32_t* a, uint32_t* b, uint32_t* cnt) {
  uint32_t A[256], PA[256], B[256], PB[256], Cn[9];
  memcpy(A, a, sizeof A); memcpy(PA, a, sizeof PA);
  for (int i = 0; i < LUTOPT512_NPARKED; i++) A[LUTOPT512_PARKED[i]] = 0xdeadbeefu;      // must not be read
  lutopt512p_step_new_parked(A, PA, B, PB, Cn);
  for (int i = 0; i < LUTOPT512_NPARKED; i++) B[LUTOPT512_PARKED[i]] = PB[LUTOPT512_PARKED[i]];
  memcpy(b, B, sizeof B); memcpy(cnt, Cn, sizeof Cn);
}
"""


def test_packed_n512_network_matches_oracle(oracle, tmp_path):
    """n = 512 on 256 registers: register p = plane p of 16 generators (low half) | plane 256 + (p ^ 1) (high half).
    The generated step (V_PERM pairs + XOR) and the two half counters against the oracle, 16 generators, 10 steps."""
    inc = tmp_path / "lutopt512_gen.inc"
    subprocess.check_call([sys.executable, str(ROOT / "basebandboard_amd" / "gen_lutopt_kernel.py"),
                           str(ROOT / "basebandboard_amd" / "data" / "lutopt_512.taps"), str(inc)])
    src = tmp_path / "h512.cpp"
    src.write_text(HARNESS512.replace("GEN_INC", inc.name))
    so = tmp_path / "h512.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-I{tmp_path}", str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    m = oracle.Lutopt(path=oracle.data_path(512))
    rng = np.random.default_rng(512)
    states = [int.from_bytes(rng.bytes(64), "little") for _ in range(16)]
    states[0], states[1] = 1, (1 << 512) - 1

    def pack(sts):
        a = np.zeros(256, dtype=np.uint32)
        for g, s in enumerate(sts):
            for p in range(256):
                if (s >> p) & 1:
                    a[p] |= np.uint32(1 << g)
                if (s >> (256 + (p ^ 1))) & 1:
                    a[p] |= np.uint32(1 << (16 + g))
        return a

    a = pack(states)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
    for _ in range(10):
        b, cnt = np.zeros(256, dtype=np.uint32), np.zeros(9, dtype=np.uint32)
        lib.step_new(P(a), P(b), P(cnt))
        bp, cntp = np.zeros(256, dtype=np.uint32), np.zeros(9, dtype=np.uint32)
        lib.step_new_parked(P(a), P(bp), P(cntp))             # the generator-placed form: same state, same counters
        assert np.array_equal(b, bp) and np.array_equal(cnt, cntp)
        states = [m.step_int(s) for s in states]
        assert np.array_equal(b, pack(states))
        for g in range(16):
            lo = sum(((int(cnt[q]) >> g) & 1) << q for q in range(9))
            hi = sum(((int(cnt[q]) >> (16 + g)) & 1) << q for q in range(9))
            assert lo + hi - 256 == m.clt_tree(states[g])          # T = number of +1 terms, tree = T - n/2
        a = b
