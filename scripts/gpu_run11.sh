cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in t g r; do
G16_GRAPH=1 G16_GRAPH_DEBUG=1 G16_GRAPH_MODE=$mode python bench.py --log2 12 --steps 3 --warmup 1 --cpu-log2 0 > gpurun_out/graph_dbg.out 2> gpurun_out/graph_dbg.err; echo "mode=$mode rc=$?"
grep "graph:" gpurun_out/graph_dbg.err | head -12
done
