export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rep8 -- python3 $R/tools/time_replica_updates.py 8 50000 8000 64 > $R/gpurun_out/prof_rep8.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_rep8/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_rep8_summary.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rep4 -- python3 $R/tools/time_replica_updates.py 4 20000 4000 32 > $R/gpurun_out/prof_rep4.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_rep4/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_rep4_summary.txt 2>&1
