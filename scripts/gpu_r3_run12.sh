#!/bin/bash
# round 3, GPU run 12: fix-up of the optimistic kernel moved to the reducing stream (sharded ranks / mid-size proofs)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_kernels.py -m gpu -x -q -k "msm or shard or multi or rank or config5 or determinism or sibling" > gpurun_out/r3_run12_pytest.log 2>&1
tail -2 gpurun_out/r3_run12_pytest.log
for k in 18 20 22; do G16_BENCH_NO_PIPELINE=1 timeout 300 python bench.py --log2 $k --steps 12 --warmup 3 --cpu-log2 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k$k', round(d['ms_per_step'],3))"; done
timeout 900 python scripts/dist_projection.py 22 8 5 points,buckets > gpurun_out/r3_proj12_k22.json 2>/dev/null
timeout 1500 python scripts/dist_projection.py 24 8 3 points,buckets > gpurun_out/r3_proj12_k24.json 2>/dev/null
python - <<'PY'
import json
for f in ("r3_proj12_k22","r3_proj12_k24"):
    d=json.load(open('/root/repo/gpurun_out/%s.json'%f))
    print(f, round(d["single_gpu_ms"],2))
    for k,v in d["ranks"].items():
        print(" ", k, round(v["per_rank_ms"],2), "eff", round(v["efficiency_before_xgmi"],3), v["ranks_timed"])
PY
