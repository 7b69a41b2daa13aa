export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/bench_configs.py c3 c4 > $R/gpurun_out/prof_c3.log 2>&1
cd $R
python3 tools/trace_summary.py gpurun_out/prof_c3/*/*_kernel_trace.csv > gpurun_out/prof_c3_summary.txt 2>&1
cat gpurun_out/prof_c3_summary.txt
