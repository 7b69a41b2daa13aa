# Kernel trace + PMC passes of what ONE rank of the view-sharded layout runs per sweep (one GPU, no exchange):
#   bash tools/profile_sliced_rank.sh 8 50000 8000 64 [--sliced]        (without --sliced: the replicated chains of round 2)
# PMC passes are their own runs with --kernel-trace only (SQ counters, FETCH_SIZE, WRITE_SIZE), the program directly after `--`.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=rank$1_k$4$( [[ " $* " == *" --sliced "* ]] && echo _sliced || echo _replicated )
ARGS="$1 $2 $3 $4 12 $5"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/time_replica_updates.py $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_${TAG}_summary.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -- python3 $R/tools/time_replica_updates.py $ARGS > $R/gpurun_out/pmc_sq_$TAG.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/tools/time_replica_updates.py $ARGS > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/tools/time_replica_updates.py $ARGS > $R/gpurun_out/pmc_write_$TAG.log 2>&1 || exit 1
cd $R
for d in pmc_sq_$TAG pmc_fetch_$TAG pmc_write_$TAG; do python3 tools/pmc_summary.py gpurun_out/$d > gpurun_out/${d}_summary.txt 2>&1; done
( echo "# PMC passes (separate runs) of one rank's kernels, $TAG: chain / S-chain / pack kernels and the passes"
  grep -hE "chain_kernel|slice_|slab_fold|pass_kernel" gpurun_out/pmc_sq_${TAG}_summary.txt gpurun_out/pmc_fetch_${TAG}_summary.txt gpurun_out/pmc_write_${TAG}_summary.txt ) > gpurun_out/pmc_${TAG}_summary.txt
tail -3 gpurun_out/prof_$TAG.log
