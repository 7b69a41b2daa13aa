# kernel trace of ONE rank's kernels in the replicated layout (no exchange), default build and every tools/micro/libresnmtf_*.so:
#   bash tools/profile_replica_updates.sh 8 50000 8000 64
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=rep$1_k$4
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/time_replica_updates.py "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_${TAG}_summary.txt 2>&1
for L in $R/tools/micro/libresnmtf_*.so; do
[ -e "$L" ] || continue
D=$(basename $L .so | sed 's/libresnmtf_//')
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$D -- python3 $R/tools/run_with_lib.py $L $R/tools/time_replica_updates.py "$@" > $R/gpurun_out/prof_${TAG}_$D.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_${TAG}_$D/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_${TAG}_${D}_summary.txt 2>&1
done
