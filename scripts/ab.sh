#!/bin/bash
# Same-box A/B driver: runs bench.py once per knob set (in the order given, REPS rounds) and prints one
# line per run.  Usage: scripts/ab.sh OUTDIR "BENCH ARGS" "NAME1=V1 NAME2=V2" "NAME3=V3" ...
# ("-" = no knobs).  REPS=2 by default.  Every comparison quoted in DESIGN.md is such a same-box pair:
# fresh gpurun boxes differ by ~3 % on the same binary.
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$1; ARGS=$2; shift 2; mkdir -p $O; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for knobs in "$@"; do
    i=$((i+1)); [ "$knobs" = "-" ] && knobs=""
    env $knobs G16_BENCH_NO_PIPELINE=1 python bench.py $ARGS --cpu-log2 0 > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err
    python - $O/ab_${i}_$rep.json "$knobs" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s]" % sys.argv[2], round(d["ms_per_step"], 3), "ms", d["parity"], {k: round(v, 2) for k, v in d["stages_ms_per_step"].items()})
except Exception as e:
    print("[%s] no line: %s" % (sys.argv[2], e))
PY
  done
done
