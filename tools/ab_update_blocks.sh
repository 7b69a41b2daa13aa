# kernel-trace A/B of the update kernels' grid size:  bash tools/ab_update_blocks.sh c4v1 0 512 256
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; shift
cd /tmp
for UB in "$@"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${CFG}_ub$UB -- python3 $R/tools/bench_configs.py $CFG --update-blocks $UB > $R/gpurun_out/prof_${CFG}_ub$UB.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_${CFG}_ub$UB/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_${CFG}_ub${UB}_summary.txt 2>&1
done
