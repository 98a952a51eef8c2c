"""How many host cores does a process on this box really get?  (cpu_baseline.cores in bench.py reports the thread
count the CPU restatement used; this prints what the scheduler / cgroup grants.)  Usage: python scripts/host_cores_probe.py"""
import multiprocessing as mp
import os
import time


def burn(_):
    t0 = time.perf_counter()
    x = 0
    for i in range(6_000_000):
        x += i * i
    return time.perf_counter() - t0


if __name__ == "__main__":
    print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(p, open(p).read().strip())
        except OSError:
            pass
    base = burn(0)
    print("one worker: %.2f s" % base)
    for n in (8, 16, 32, 64, 128):
        if n > 2 * (os.cpu_count() or 1):
            break
        with mp.Pool(n) as pool:
            t0 = time.perf_counter()
            ts = pool.map(burn, range(n), chunksize=1)
            wall = time.perf_counter() - t0
        print("%3d workers: wall %.2f s, mean per-worker %.2f s -> effective cores %.1f" % (n, wall, sum(ts) / n, n * base / wall))
