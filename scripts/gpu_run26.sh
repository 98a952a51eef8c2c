cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --window-bits $2 --steps 40 --warmup 5 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity']['proof_verifies'])"; }
one "--log2 11" 0
for rep in 1 2; do
for wl in "--log2 17" "--log2 18" "--log2 19"; do
  for c in 0 15 16 17; do one "$wl" $c; done
done
done
