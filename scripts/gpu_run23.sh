# window-size candidates at small sizes, two repetitions, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --window-bits $2 --steps 60 --warmup 5 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity']['proof_verifies'])"; }
for rep in 1 2; do
for wl in "--workload complex-circuit" "--log2 12" "--log2 13" "--log2 14" "--log2 15"; do
  for c in 0 13 15 16 17; do one "$wl" $c; done
done
done
