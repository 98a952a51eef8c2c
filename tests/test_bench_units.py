"""CPU-side units of bench.py's round-5 measurement legs (no GPU): the PMC traffic arithmetic on a fabricated
rocprofv3 counter file, and the clock sampler's behaviour on a box without a device (it must never fail a bench)."""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _write_pass(root, name, rows):
    d = os.path.join(root, name)
    os.makedirs(d)
    with open(os.path.join(d, name + "_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(w.fieldnames, r)))


def test_pmc_traffic_sizes_requests_by_their_size_counters(tmp_path):
    """scripts/pmc_traffic.collect (what bench.py's own `rocprofv3 --pmc` passes are parsed with): read requests
    are sized by TCC_EA0_RDREQ_32B / _64B / _128B (the rest as 64 B), writes by TCC_EA0_WRREQ[_64B]; the optimistic
    G1 launches (FAST = true) and the exact G2 launch are the ones that count, the early-exit exact G1 launch behind
    an optimistic one is ignored; launches are averaged per kernel class."""
    import pmc_traffic
    g2 = "void g16::(anonymous namespace)::k_bucket_accumulate<g16::Fq2, 1, false, false>(args)"
    g1 = "void g16::(anonymous namespace)::k_bucket_accumulate<g16::Fp<g16::FqParams>, 1, false, true>(args)"
    g1_exact = "void g16::(anonymous namespace)::k_bucket_accumulate<g16::Fp<g16::FqParams>, 1, false, false>(args)"
    pair = "void g16::(anonymous namespace)::k_bucket_accumulate<g16::Fp<g16::FqParams>, 2, true, true>(args)"
    rd = []
    for k, n128 in ((g2, 100), (g2, 300), (g1, 50), (g1_exact, 7), (pair, 80)):
        rd += [(k, "TCC_EA0_RDREQ", n128 + 10), (k, "TCC_EA0_RDREQ_32B", 2), (k, "TCC_EA0_RDREQ_64B", 3),
               (k, "TCC_EA0_RDREQ_128B", n128)]
    _write_pass(str(tmp_path), "p0", rd)
    _write_pass(str(tmp_path), "p1", [(g2, "TCC_EA0_WRREQ", 12), (g2, "TCC_EA0_WRREQ_64B", 10),
                                      (g1, "TCC_EA0_WRREQ", 4), (g1, "TCC_EA0_WRREQ_64B", 4)])
    rec = pmc_traffic.collect(str(tmp_path), 22)
    # G2: mean of the two launches: 128B 200, 32B 2, 64B 3, other 5 -> 200*128 + 2*32 + (3+5)*64
    assert rec["g2_read_bytes_per_launch"] == 200 * 128 + 2 * 32 + 8 * 64
    assert rec["g2_write_bytes_per_launch"] == 10 * 64 + 2 * 32
    assert rec["read_bytes_per_launch"] == 50 * 128 + 2 * 32 + 8 * 64 and rec["launches_averaged"] == 1
    assert rec["pair_read_bytes_per_launch"] == 80 * 128 + 2 * 32 + 8 * 64
    assert rec["traffic_bytes_per_launch"] == rec["read_bytes_per_launch"] + 4 * 64


def test_clock_sampler_never_fails_without_a_device(monkeypatch):
    """bench.ClockSampler on a box with no GPU: no HIP runtime to ask for a PCI address, no hwmon node, possibly no
    rocm-smi -- the sampler reports what it has (nothing) and stops at once; ordinal None (idle ranks) starts no thread."""
    import bench
    s = bench.ClockSampler(None)
    s.mark("timed")
    s.stop()
    out = s.summary()
    assert out["timed"]["samples"] == 0 and out["timed"]["sclk_mhz_median"] is None
    monkeypatch.setenv("PATH", "/nonexistent")          # no rocm-smi either
    t0 = time.time()
    s = bench.ClockSampler(0)
    s.mark("timed")
    time.sleep(0.05)
    s.mark("after")
    s.stop()
    assert time.time() - t0 < 60.0      # "does not hang": a loaded box (or the first HIP runtime probe) may take seconds
    out = s.summary()
    assert set(out) >= {"source", "timed", "after"}
    for label in ("timed", "after"):
        assert out[label]["samples"] >= 0 and (out[label]["samples"] == 0) == (out[label]["sclk_mhz_median"] is None)


def test_pmc_self_measurement_degrades_to_a_note(monkeypatch):
    """bench.measure_pmc_traffic on a box without rocprofv3: no exception, no record, a note that says why (bench.py
    then falls back to the stamped profiles/pmc_traffic.json when its library hash matches, or reports no traffic)."""
    import argparse
    import bench
    monkeypatch.setenv("PATH", "/nonexistent")
    rec, note = bench.measure_pmc_traffic(argparse.Namespace(workload="chain", window_bits=0, planes=0), 22)
    assert rec is None and "rocprofv3" in note
