# kernel traces of the view-sharded path on the one-GPU box (one gpurun call):
#   (1) one rank's share of an 8-view run: fused F chain against one launch per view (tools/time_f_chain.py)
#   (2) bench.py's multi-rank code path with one rank on RCCL (compact block, fold, in-place all-gather)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_chain -- python3 $R/tools/time_f_chain.py --views 8 --reps 200 > $O/prof_chain.log 2>&1
RESNMTF_FORCE_SHARDED=1 RESNMTF_FORCE_REPLICATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard1 -- python3 $R/bench.py --steps 300 --warmup 30 > $O/prof_shard1.log 2>&1
cd $R
for d in prof_chain prof_shard1; do
  f=$(find $O/$d -name '*kernel_trace.csv' | head -n 1)
  python3 tools/trace_summary.py $f > $O/${d}_summary.txt
done
echo done
