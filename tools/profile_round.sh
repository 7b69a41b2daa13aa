# The round's evidence session on the GPU box (one gpurun call):  bash tools/profile_round.sh
#   bench line (driver's flags and the default), rocprofv3 --kernel-trace --stats of the same command, PMC passes
#   (FETCH_SIZE / WRITE_SIZE / SQ counters: each its own run, --kernel-trace only), the other BASELINE shapes on one GPU,
#   kernel traces and PMC passes of one c4-sized and one c5-sized view.  Everything lands in gpurun_out/; the summaries
#   are copied by hand into profiles/ (named per round).  Second call of the session: tools/profile_round_sharded.sh.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.log 2>&1
tail -1 $O/bench_driver_flags.log > $O/bench_driver_flags.json
python3 $R/bench.py > $O/bench_final.log 2>&1
tail -1 $O/bench_final.log > $O/bench_final.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 $R/bench.py --no-cpu-baseline > $O/prof_final.log 2>&1
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch2 -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_fetch2.log 2>&1
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write2 -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_write2.log 2>&1
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
cd $R
python3 tools/trace_summary.py $(ls -t gpurun_out/prof_final/*/*_kernel_trace.csv | head -1) > gpurun_out/prof_final_summary.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_summary.txt 2>&1
python3 tools/bench_configs.py c2 c3 c4v1 c4 c5v1 > $O/configs_final.log 2>&1
bash tools/profile_cfg.sh c4v1 > /dev/null 2>&1
bash tools/profile_cfg.sh c5v1 > /dev/null 2>&1
bash tools/profile_cfg_pmc.sh c4v1 > $O/pmc_c4v1.log 2>&1
bash tools/profile_cfg_pmc.sh c5v1 > $O/pmc_c5v1.log 2>&1
python3 tools/make_traffic.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 80768000 gpurun_out/traffic_new.json "round 3 final build: tools/profile_round.sh (bench.py --steps 40 --warmup 5, eager launches, separate --pmc FETCH_SIZE / WRITE_SIZE runs)" > /dev/null 2>&1
echo done
