# window-size sweep at small sizes (latency regime): ms per proof, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { python bench.py $1 --window-bits $2 --steps 40 --warmup 5 --cpu-log2 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); i=d['config']['msm']; print('$1 c=%d W=%d' % (i['c_w'], i['W_w']), round(d['ms_per_step'],3), d['parity']['proof_verifies'], {k: round(v,2) for k,v in d['stages_ms_per_step'].items() if v})"; }
for wl in "--workload complex-circuit" "--log2 12" "--log2 14" "--log2 16" "--log2 17"; do
  for c in 0 8 9 10 11 12 13 14 15; do one "$wl" $c; done
done
