# same-box A/B at small sizes: B2 MSM first (its reduction under everything else) vs last
# NOTE: the kernels / knob this script A/B-ed were measured and dropped; the env variables are no longer read.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { env "${@:2}" python bench.py $1 --steps 50 --warmup 5 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 ${@:2}', round(d['ms_per_step'],3), d['parity']['proof_verifies'])"; }
for rep in 1 2; do
for wl in "--workload complex-circuit" "--log2 12" "--log2 14" "--log2 16" "--log2 17" "--log2 18"; do
  one "$wl" G16_B2_FIRST=0
  one "$wl" G16_B2_FIRST=1
done
done
