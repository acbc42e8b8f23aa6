"""Philox4x32-10 known-answer vectors (Random123 kat_vectors) for the oracle's generator; the HIP
kernels carry the same function (csrc/grx_rng.h) and are compared stream-by-stream in the GPU tests."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = r'''
#include "philox.h"
void kat(const uint32_t* c, const uint32_t* k, uint32_t* o) { gro_philox4x32_10(c, k, o); }
float u01(uint32_t x) { return gro_u01(x); }
float rnd(uint64_t seed, uint32_t e, uint32_t s, uint32_t st, uint32_t i) { return gro_rand(seed, e, s, st, i); }
'''


def _lib(tmp_path):
    src = tmp_path / "kat.c"
    src.write_text(SRC)
    so = tmp_path / "kat.so"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-I", os.path.join(HERE, "..", "oracle"), str(src), "-o", str(so)], check=True)
    return C.CDLL(str(so))


def test_known_answers(tmp_path):
    lib = _lib(tmp_path)
    A = C.c_uint32 * 4
    K = C.c_uint32 * 2
    cases = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
             ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in cases:
        out = A()
        lib.kat(A(*ctr), K(*key), out)
        assert tuple(out) == want


def test_uniform_range_and_streams(tmp_path):
    lib = _lib(tmp_path)
    lib.u01.restype = C.c_float
    lib.rnd.restype = C.c_float
    lib.rnd.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    assert lib.u01(0) == 0.0 and lib.u01(0xffffffff) < 1.0
    vals = [lib.rnd(1, e, 3, 6, i) for e in range(8) for i in range(39)]
    assert all(0.0 <= v < 1.0 for v in vals) and len(set(vals)) > 300
    assert lib.rnd(1, 5, 3, 6, 7) != lib.rnd(2, 5, 3, 6, 7) != lib.rnd(1, 6, 3, 6, 7)
    assert 0.4 < sum(vals) / len(vals) < 0.6
