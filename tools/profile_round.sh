set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/bench_final.log 2>&1
tail -1 $O/bench_final.log > $O/bench_final.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 $R/bench.py --no-cpu-baseline > $O/prof_final.log 2>&1
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch2 -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_fetch2.log 2>&1
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write2 -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_write2.log 2>&1
cd $R
python3 tools/bench_configs.py c2 c3 c4v1 c4 c5v1 > $O/configs_final.log 2>&1
echo done
