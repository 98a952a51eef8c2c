cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --window-bits $2 --steps 60 --warmup 5 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity']['proof_verifies'])"; }
for wl in "--log2 4" "--log2 6" "--log2 8" "--log2 10" "--log2 11"; do
  for c in 0 5 13 15; do one "$wl" $c; done
done
