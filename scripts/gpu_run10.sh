# hipGraph replay of small proofs: same-box A/B (G16_GRAPH=0 / 1) + correctness
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for g in 0 1; do
  for k in 12 14 16 17 18; do
    G16_GRAPH=$g python bench.py --log2 $k --steps 50 --warmup 5 --cpu-log2 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('graph=$g k=$k ms=%.4f verifies=%s' % (d['ms_per_step'], d['parity']['proof_verifies']))"
  done
  G16_GRAPH=$g python bench.py --workload complex-circuit --steps 50 --warmup 5 --cpu-log2 14 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('graph=$g complex ms=%.4f %s' % (d['ms_per_step'], d['parity']))"
done; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
