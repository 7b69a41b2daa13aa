#!/bin/bash
# c2 only: base vs max-ilp, alternating, then the kernel trace of both
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r3c; mkdir -p $out
cd $R
for i in 1 2 3; do
  echo "base $(timeout -k 10 100 python3 tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-80)"
  echo "ilp  $(timeout -k 10 100 python3 tools/run_with_lib.py tools/micro/libresnmtf_ilp.so tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-80)"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2_base -- python3 $R/tools/bench_configs.py c2 > $out/prof_c2_base.log 2>&1
python3 $R/tools/trace_summary.py $(ls -t $out/prof_c2_base/*/*_kernel_trace.csv | head -1) | head -6
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2_ilp -- python3 $R/tools/run_with_lib.py $R/tools/micro/libresnmtf_ilp.so $R/tools/bench_configs.py c2 > $out/prof_c2_ilp.log 2>&1
python3 $R/tools/trace_summary.py $(ls -t $out/prof_c2_ilp/*/*_kernel_trace.csv | head -1) | head -6
