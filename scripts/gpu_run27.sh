cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --window-bits $2 --steps 20 --warmup 3 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity']['proof_verifies'])"; }
for rep in 1 2; do
for c in 0 16 17; do one "--log2 20" $c; done
for c in 0 17 20 22; do one "--log2 21" $c; done
for c in 0 16 17; do one "--workload dense-skewed --log2 20" $c; done
done
