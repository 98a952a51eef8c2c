#!/bin/bash
# hardware_day.sh -- everything that needs MORE than one MI355X, in one go, for the day an 8-GPU node
# is available (the builder never had one: gpurun boxes are 1-GPU).  Nothing here runs in the default
# suites.  Usage: scripts/hardware_day.sh [log2=24] [steps=10]   (from the repo root, on the node)
#
#   0. topology: rocm-smi --showtopo, visible devices, peer-access matrix as HIP reports it
#   1. g16_ctx_create_multi on 2 / 4 / 8 DISTINCT devices: the create-time self-test (all-to-all echo +
#      record gather over real peer copies and cross-device events: csrc/multi.hip) must pass, peer state
#      is printed, one proof is byte-compared with the single-GPU proof of the same inputs
#   2. bench.py at N = 1, 2, 4, 8 in-library (one process, one host thread per device, peer copies)
#   3. bench.py at N = 2, 4, 8 with one process per GPU (G16_BENCH_MODE=ranks: RCCL all_to_all /
#      all_gather over xGMI on the registered exchange stream), RCCL's rank count printed
#   3b. one process per GPU with the collectives issued by the LIBRARY (g16_dist_attach_rccl / g16_prove_dist,
#      scripts/rccl_inlib_ranks.py), N = 2, 4, 8: every rank's proof == the single-GPU proof
#   4. the scaling table T1 / (N x T_N) for all three, next to the one-GPU projection of
#      profiles/r06_proj_k24.json
set -u
K=${1:-24}; STEPS=${2:-10}
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/hardware_day; mkdir -p $O; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
# G16_HWDAY_FAKE=1: rehearsal of THIS SCRIPT on a one-GPU box -- every rank on device 0, gloo between the
# processes (what tests/test_bench_shapes.py does): checks the script's own plumbing, measures nothing
FAKE=${G16_HWDAY_FAKE:-0}
if [ "$FAKE" = "1" ]; then NDEV=8; export G16_BENCH_BACKEND=gloo G16_BENCH_DEVICE=0; echo "REHEARSAL on one GPU: every number below is functional only"; fi
export G16_HWDAY_FAKE=$FAKE
echo "== 0. topology: $NDEV visible device(s)"
rocm-smi --showtopo 2>/dev/null | tee $O/topo.txt | head -60
if [ "$NDEV" -lt 2 ]; then echo "this box has one GPU: nothing to do here (tests/ and bench.py cover N = 1)"; exit 0; fi
echo "== 1. in-library multi-device ctx: self-test at create, peer state, bytes vs the single-GPU proof"
python - $K <<'PY' 2>&1 | tee $O/create_multi.txt
import os, random, sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch, bench
import circom_compat_amd as cc
k = min(int(sys.argv[1]), 22)          # the functional check does not need the full size
mats, (A, B, Cm), w_ints, n_vars = bench.chain_circuit(cc, k)
rng = random.Random(k)
tox = [rng.randrange(1, bench.R_MOD) for _ in range(5)]
pk = cc.trapdoor_setup(A, B, Cm, n_vars, 1, tox)
rs = cc.fr_from_ints([rng.randrange(bench.R_MOD), rng.randrange(bench.R_MOD)])
w = cc.fr_from_ints(w_ints)
single = cc.Prover(pk, mats)
want = single.prove(rs[0], rs[1], w).raw
single.close()
fake = os.environ.get("G16_HWDAY_FAKE") == "1"
ndev = 8 if fake else torch.cuda.device_count()
for n in (2, 4, 8):
    if n > ndev:
        break
    for shard in ("points", "buckets"):
        try:
            pr = cc.Prover(pk, mats, devices=[0] * n if fake else list(range(n)), shard=shard)   # runs the self-test; raises on a broken peer path
        except Exception as e:
            print(f"N={n} {shard}: CREATE FAILED: {e}")
            continue
        info = pr.info()
        got = pr.prove(rs[0], rs[1], w).raw
        print(f"N={n} {shard}: self-test passed, peer_access={info['peer_access']} (1 = direct for every pair, 2 = some staged), "
              f"bytes == single-GPU proof: {got == want}")
        if shard == "points":
            # the create-time link probe (g16_multi_links): GB/s of one large peer copy per ordered pair, and a
            # 4 KiB there-and-back -- the measurement behind the 48 GB/s-per-link figure of scripts/dist_projection.py
            lk = pr.links()
            off = [lk["gbps"][a][b] for a in range(n) for b in range(n) if a != b]
            echo = [lk["echo_us"][a][b] for a in range(n) for b in range(n) if a != b]
            print(f"N={n} links ({lk['probe_bytes'] >> 20} MiB per copy): GB/s min {min(off):.1f} / max {max(off):.1f}; "
                  f"4 KiB there-and-back us min {min(echo):.1f} / max {max(echo):.1f}")
            for a in range(n):
                print("   from rank %d: " % a + " ".join("%7.1f" % x for x in lk["gbps"][a]))
        pr.close()
PY
echo "== 2. bench.py in-library, N = 1, 2, 4, 8"
for N in 1 2 4 8; do
  [ "$N" -gt "$NDEV" ] && break
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --log2 $K --steps $STEPS --warmup 2 --cpu-log2 0 --no-pmc --no-secondary > $O/inlib_$N.json 2> $O/inlib_$N.err
  else
    G16_BENCH_MODE=inlib python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) \
      bench.py --gpus $N --log2 $K --steps $STEPS --warmup 2 --cpu-log2 0 > $O/inlib_$N.json 2> $O/inlib_$N.err
  fi
  tail -1 $O/inlib_$N.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inlib N=%d' % d['n_gpus'], round(d['ms_per_step'],2), 'ms', d['parity'], d['config'].get('parallelism'))"
done
echo "== 3. bench.py one process per GPU over RCCL, N = 2, 4, 8"
for N in 2 4 8; do
  [ "$N" -gt "$NDEV" ] && break
  NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT G16_BENCH_MODE=ranks python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29700+N)) bench.py --gpus $N --log2 $K --steps $STEPS --warmup 2 --cpu-log2 0 > $O/ranks_$N.json 2> $O/ranks_$N.err
  echo "RCCL: $(grep -c 'Init COMPLETE' $O/ranks_$N.err) rank(s) report Init COMPLETE; $(grep -m1 -o 'nranks [0-9]*' $O/ranks_$N.err)"
  tail -1 $O/ranks_$N.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ranks N=%d' % d['n_gpus'], round(d['ms_per_step'],2), 'ms', d['parity'], d['config'].get('parallelism'))"
done
echo "== 3b. one process per GPU, RCCL issued by the LIBRARY (g16_dist_attach_rccl / g16_prove_dist), N = 2, 4, 8"
# (rehearsal on one GPU: RCCL does not put two ranks on one device -- only the one-rank communicator runs there)
if [ "$FAKE" = "1" ]; then LIST="1"; else LIST="2 4 8"; fi
for N in $LIST; do
  [ "$N" -gt "$NDEV" ] && break
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29800+N)) \
    scripts/rccl_inlib_ranks.py $K $STEPS points > $O/rccl_inlib_$N.json 2> $O/rccl_inlib_$N.err
  tail -1 $O/rccl_inlib_$N.json
done
echo "== 4. scaling table"
python - $O <<'PY'
import glob, json, os, sys
def ms(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        return None
t1 = ms(os.path.join(sys.argv[1], "inlib_1.json"))
for mode in ("inlib", "ranks", "rccl_inlib"):
    for n in (2, 4, 8):
        t = ms(os.path.join(sys.argv[1], f"{mode}_{n}.json"))
        if t1 and t:
            print(f"{mode} N={n}: {t:.2f} ms, strong-scaling efficiency T1/(N T_N) = {t1 / (n * t):.3f}")
try:
    p = json.load(open("profiles/r06_proj_k24.json"))
    for k, v in p["ranks"].items():
        print("one-GPU projection", k, round(v["efficiency_before_xgmi"], 3), "/ with all link time exposed", round(v["efficiency_if_all_link_time_exposed"], 3))
except Exception as e:
    print("no projection file:", e)
PY
