# kernel-trace summary of a bench_configs.py config:  bash tools/profile_cfg.sh c5v1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-c5v1}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$CFG -- python3 $R/tools/bench_configs.py $CFG > $R/gpurun_out/prof_$CFG.log 2>&1
cd $R
python3 tools/trace_summary.py $(ls -t gpurun_out/prof_$CFG/*/*_kernel_trace.csv | head -1) > gpurun_out/prof_${CFG}_summary.txt 2>&1
head -12 gpurun_out/prof_${CFG}_summary.txt
