#!/bin/bash
# round 6, end-of-round validation F: 2^27 (the largest domain the reference accepts) on one GPU with the FINAL
# binary: planes per the memory plan, pairing-verified line, no CPU proof; plus ctx-create wall time of the
# fixed-base-table path (what the 127 per-entry inversions of k_tbl_entries cost at create)
TAG=r06
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final_$TAG; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import hashlib; print("library sha16", hashlib.sha256(open("circom_compat_amd/libg16_amd.so","rb").read()).hexdigest()[:16])
PY
python - <<'PY' 2>&1 | tee $O/${TAG}_table_create_time.txt
import random, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import bench
import circom_compat_amd as cc
for k in (10, 12, 14):
    mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
    rng = random.Random(k)
    pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, [rng.randrange(1, bench.R_MOD) for _ in range(5)])
    for tables in (-1, 0, -1, 0):
        t0 = time.perf_counter(); pr = cc.Prover(pk, mats, tables=tables); t1 = time.perf_counter()
        info = pr.info(); pr.close()
        print(f"2^{k} wires, tables={tables}: ctx create {1e3 * (t1 - t0):8.1f} ms  fixed_tables={info['fixed_tables']}")
PY
free -g | head -2
avail=$(free -g | awk '/^Mem:/ {print $7}')
if [ "$avail" -lt 400 ]; then echo "less than 400 GB of host memory available: not attempting 2^27"; exit 0; fi
G16_BENCH_NO_PIPELINE=1 timeout 1500 python bench.py --log2 27 --steps 2 --warmup 1 --cpu-log2 0 --no-pmc --no-secondary > $O/${TAG}_bench_chain27.json 2> $O/k27.err; echo "rc=$?"
tail -3 $O/k27.err | cut -c1-300
python - $O/${TAG}_bench_chain27.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 2), "M/s", d["parity"], d["config"]["msm"], d["stages_ms_per_step"])
PY
