# A/B of compile-time variants of the library under the kernel trace:  bash tools/ab_variant_libs.sh c5v1 [c4v1 ...]
#   variant libraries are built beforehand into tools/micro/libresnmtf_<tag>.so (e.g. -DRESNMTF_UPD_DEPTH=2 -> d2)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for CFG in "$@"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_base_$CFG -- python3 $R/tools/bench_configs.py $CFG > $R/gpurun_out/prof_base_$CFG.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_base_$CFG/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_base_${CFG}_summary.txt 2>&1
for L in $R/tools/micro/libresnmtf_*.so; do
D=$(basename $L .so | sed 's/libresnmtf_//')
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${D}_$CFG -- python3 $R/tools/run_with_lib.py $L $R/tools/bench_configs.py $CFG > $R/gpurun_out/prof_${D}_$CFG.log 2>&1 || exit 1
python3 $R/tools/trace_summary.py $(ls -t $R/gpurun_out/prof_${D}_$CFG/*/*_kernel_trace.csv | head -1) > $R/gpurun_out/prof_${D}_${CFG}_summary.txt 2>&1
done
done
