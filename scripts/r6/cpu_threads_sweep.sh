#!/bin/bash
# round 6: how many threads, and which OpenMP wait policy, make the CPU restatement fastest under the box's cgroup
# CPU quota (16 of 256 visible CPUs)?  2^20 chain proof, two proofs per setting.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_r06; mkdir -p $O; export TMPDIR=/tmp
K=${1:-20}
for pol in default passive; do
for t in 16 24 32 48 64 128; do
if [ "$pol" = "passive" ]; then export OMP_WAIT_POLICY=passive; else unset OMP_WAIT_POLICY; fi
G16_CPU_THREADS=$t python - $t $pol $K <<'PY'
import random, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import bench, cpu_ref
import circom_compat_amd as cc
k = int(sys.argv[3])
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, [rng.randrange(1, bench.R_MOD) for _ in range(5)])
rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)]); w = cc.fr_from_ints(w_ints)
ts = []
for _ in range(2):
    t0 = time.perf_counter(); cpu_ref.prove(pk, mats, rs[0:1].copy(), rs[1:2].copy(), w); ts.append(time.perf_counter() - t0)
print("2^%d CPU proof, %3s threads, OMP_WAIT_POLICY %-7s: %s s" % (k, sys.argv[1], sys.argv[2], " ".join("%.2f" % t for t in ts)))
PY
done; done 2>&1 | grep "CPU proof" | tee $O/r06_cpu_threads_sweep.txt
