# HBM read requests per accumulate launch: A|B1 interleaved pair (one launch) vs separate arrays
cd /tmp; export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT; mkdir -p $ROOT/gpurun_out/r17
for v in 0 1; do
  G16_NO_PAIR_AB=$v G16_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B \
      --kernel-include-regex k_bucket_accumulate -f csv -d $ROOT/gpurun_out/r17/nopair$v -o p -- \
      python $ROOT/bench.py --log2 22 --steps 1 --warmup 0 --cpu-log2 0 > $ROOT/gpurun_out/r17/nopair$v.log 2>&1
  echo "variant G16_NO_PAIR_AB=$v rc=$?"
  G16_NO_PAIR_AB=$v G16_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ \
      --kernel-include-regex k_bucket_accumulate -f csv -d $ROOT/gpurun_out/r17/hit$v -o p -- \
      python $ROOT/bench.py --log2 22 --steps 1 --warmup 0 --cpu-log2 0 > $ROOT/gpurun_out/r17/hit$v.log 2>&1
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r17"
for d in sorted(os.listdir(root)):
    for path in glob.glob(os.path.join(root, d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(root, d, "*counter_collection.csv")):
        rows = defaultdict(dict)
        for r in csv.DictReader(open(path)):
            rows[(int(r["Dispatch_Id"]), r["Kernel_Name"][:60])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (k, name), c in sorted(rows.items()):
            print(d, k, name[38:60], {a: round(b / 1e6, 2) for a, b in c.items()})
PY
