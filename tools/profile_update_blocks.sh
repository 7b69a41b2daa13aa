# kernel-trace of c2 under different update-kernel blockings: how the update kernels and the pass launches
# (whose workgroup 0 sums the update kernels' partial records) move with the number of records
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
for ub in 80 160 320; do
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_ub$ub -- python3 $R/tools/tune_c2.py c2 update_blocks=$ub > $O/prof_ub$ub.log 2>&1
  f=$(find $O/prof_ub$ub -name '*kernel_trace.csv' | head -n 1)
  echo "update_blocks=$ub"; python3 $R/tools/trace_summary.py $f | head -n 5 | cut -c1-120
  grep "us/sweep" $O/prof_ub$ub.log
done
