"""field.h / ec.h (the device arithmetic source) compiled for the host vs the oracle.

Not a GPU parity claim: this pins the *source* of the device arithmetic, bit for bit, before any
kernel runs.  The same vectors run on the real GPU in test_gpu_parity.py.
"""
import ctypes as C
import random

import numpy as np
import pytest

import bn254_ref as o

R = 1 << 256


def _mont(vals, p):
    return np.frombuffer(b"".join(((v % p) * R % p).to_bytes(32, "little") for v in vals), dtype=np.uint32).copy()


def _unmont(arr, p):
    b = arr.tobytes()
    ri = pow(R, -1, p)
    return [int.from_bytes(b[i:i + 32], "little") * ri % p for i in range(0, len(b), 32)]


def _edge(p, rng, n):
    base = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 253) % p, (1 << 32) - 1, (1 << 64) - 1, R % p, (R * R) % p]
    return base + [rng.randrange(p) for _ in range(n)]


@pytest.mark.parametrize("field", [0, 1])
def test_fp_ops(emu, field):
    p = o.R_MOD if field == 0 else o.Q_MOD
    rng = random.Random(100 + field)
    xs = _edge(p, rng, 300)
    ys = list(reversed(_edge(p, rng, 300)))
    n = len(xs)
    a, b = _mont(xs, p), _mont(ys, p)
    out = np.empty_like(a)
    f = emu.L.emu_fp_op
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    ops = {0: lambda x, y: x * y % p, 1: lambda x, y: (x + y) % p, 2: lambda x, y: (x - y) % p,
           3: lambda x, y: (-x) % p, 5: lambda x, y: x * x % p, 8: lambda x, y: 2 * x % p,
           4: lambda x, y: pow(x, p - 2, p)}
    for op, ref in ops.items():
        f(field, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
        got = _unmont(out, p)
        assert got == [ref(x, y) for x, y in zip(xs, ys)], f"field {field} op {op}"
    # Montgomery <-> canonical (into_bigint / from_bigint)
    f(field, 6, a.ctypes.data, None, out.ctypes.data, n)
    raw = out.tobytes()
    assert [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)] == xs
    canon = np.frombuffer(b"".join(x.to_bytes(32, "little") for x in xs), dtype=np.uint32).copy()
    f(field, 7, canon.ctypes.data, None, out.ctypes.data, n)
    assert _unmont(out, p) == xs


def test_fq2_ops(emu):
    rng = random.Random(7)
    p = o.Q_MOD
    n = 100
    xs = [(rng.randrange(p), rng.randrange(p)) for _ in range(n)] + [(0, 0), (1, 0), (0, 1), (p - 1, p - 1)]
    ys = [(rng.randrange(p), rng.randrange(p)) for _ in range(n)] + [(5, 7), (0, 0), (p - 1, 1), (p - 1, p - 1)]
    n = len(xs)
    flat = lambda zs: _mont([c for z in zs for c in z], p)
    a, b = flat(xs), flat(ys)
    out = np.empty_like(a)
    f = emu.L.emu_fq2_op
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    refs = {0: o.f2_mul, 1: o.f2_add, 2: o.f2_sub, 3: lambda x, y: o.f2_neg(x), 5: lambda x, y: o.f2_sqr(x)}
    for op, ref in refs.items():
        f(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
        got = _unmont(out, p)
        got = [(got[2 * i], got[2 * i + 1]) for i in range(n)]
        assert got == [ref(x, y) for x, y in zip(xs, ys)], op
    nz = [x for x in xs if x != (0, 0)]
    a = flat(nz)
    out = np.empty_like(a)
    f(4, a.ctypes.data, None, out.ctypes.data, len(nz))
    got = _unmont(out, p)
    assert [(got[2 * i], got[2 * i + 1]) for i in range(len(nz))] == [o.f2_inv(x) for x in nz]


def _ec_case(emu, g2):
    rng = random.Random(11 + g2)
    Cv = o.G2 if g2 else o.G1
    gen = o.G2_GEN if g2 else o.G1_GEN
    to_b = o.g2_to_bytes if g2 else o.g1_to_bytes
    from_b = o.g2_from_bytes if g2 else o.g1_from_bytes
    ps = 128 if g2 else 64
    fs = 64 if g2 else 32
    f = emu.L.emu_g2_op if g2 else emu.L.emu_g1_op
    f.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_size_t]
    n = 24
    P = [Cv.mul(gen, rng.randrange(1, o.R_MOD)) for _ in range(n)]
    Q = [Cv.mul(gen, rng.randrange(1, o.R_MOD)) for _ in range(n)]
    # special cases: P == Q (doubling inside add), P == -Q (infinity), infinities
    P[0], Q[0] = P[1], P[1]
    P[2], Q[2] = Q[3], Cv.neg(Q[3])
    P[4] = None
    Q[5] = None
    P[6], Q[6] = None, None

    def lam():
        if g2:
            return o.fq_to_mont_bytes(rng.randrange(1, o.Q_MOD)) + o.fq_to_mont_bytes(rng.randrange(o.Q_MOD))
        return o.fq_to_mont_bytes(rng.randrange(1, o.Q_MOD))

    pb = np.frombuffer(b"".join(map(to_b, P)), dtype=np.uint8).copy()
    qb = np.frombuffer(b"".join(map(to_b, Q)), dtype=np.uint8).copy()
    l1 = np.frombuffer(b"".join(lam() for _ in range(n)), dtype=np.uint8).copy()
    l2 = np.frombuffer(b"".join(lam() for _ in range(n)), dtype=np.uint8).copy()
    ks = [rng.randrange(o.R_MOD) for _ in range(n)]
    ks[0], ks[1], ks[2] = 0, 1, o.R_MOD - 1
    kb = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint8).copy()
    out = np.empty(n * ps, dtype=np.uint8)

    def run(op):
        f(op, pb.ctypes.data, qb.ctypes.data, l1.ctypes.data, l2.ctypes.data, kb.ctypes.data, out.ctypes.data, n)
        raw = out.tobytes()
        return [from_b(raw[i * ps:(i + 1) * ps]) for i in range(n)]

    assert run(0) == [Cv.add(p, q) for p, q in zip(P, Q)]
    assert run(1) == [Cv.add(p, q) for p, q in zip(P, Q)]
    assert run(2) == [Cv.add(p, p) for p in P]
    assert run(4) == [Cv.add(p, p) for p in P]
    assert run(3) == [Cv.mul(p, k) if p is not None else None for p, k in zip(P, ks)]
    assert run(5) == [Cv.mul(p, k & 0xffffffff) if p is not None else None for p, k in zip(P, ks)]


def test_g1_ops(emu):
    _ec_case(emu, 0)


def test_g2_ops(emu):
    _ec_case(emu, 1)
