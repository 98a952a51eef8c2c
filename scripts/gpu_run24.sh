cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity'])"; }
for wl in "--workload complex-circuit" "--log2 11" "--log2 12" "--log2 13" "--log2 14" "--log2 15" "--log2 16"; do one "$wl"; done
python bench.py --workload complex-circuit --steps 20 > gpurun_out/r02_bench_complex.json 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q -k "not headline_sizes" 2>&1 | tail -3
