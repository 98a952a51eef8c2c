"""Throughput of g16_verify_batch (one lane per proof): n copies of a valid proof of the reference's
test.zkey + one wrong public input at a known position.  python scripts/verify_bench.py [n=16384]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import circom_compat_amd as cc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pk, mats = cc.read_zkey(os.path.join(ROOT, "tests", "golden", "test.zkey"))
proof = cc.Prover(pk, mats).prove(12345, 67890, [1, 33, 3, 11])
pubs = [[33]] * n
pubs[n // 3] = [34]
for rep in range(2):
    t = time.perf_counter()
    ok = cc.verify_batch(pk.vk, [proof] * n, pubs)
    dt = time.perf_counter() - t
    assert ok.count(False) == 1 and ok[n // 3] is False
    print(f"n={n} rep={rep}: {dt * 1e3:.1f} ms, {n / dt:.0f} proofs/s (host packing included)")
