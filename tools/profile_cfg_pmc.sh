# PMC passes of one single-view shape (c4v1 / c5v1), each its own run with --kernel-trace only:
#   bash tools/profile_cfg_pmc.sh c5v1 [extra args of run_cfg_eager.py]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-c5v1}
shift
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$CFG -- python3 $R/tools/run_cfg_eager.py $CFG "$@" > $R/gpurun_out/pmc_sq_$CFG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$CFG -- python3 $R/tools/run_cfg_eager.py $CFG "$@" > $R/gpurun_out/pmc_fetch_$CFG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$CFG -- python3 $R/tools/run_cfg_eager.py $CFG "$@" > $R/gpurun_out/pmc_write_$CFG.log 2>&1
cd $R
for d in pmc_sq_$CFG pmc_fetch_$CFG pmc_write_$CFG; do python3 tools/pmc_summary.py gpurun_out/$d > gpurun_out/${d}_summary.txt 2>&1; done
grep -E "pass_kernel|factor_update" gpurun_out/pmc_sq_${CFG}_summary.txt gpurun_out/pmc_fetch_${CFG}_summary.txt gpurun_out/pmc_write_${CFG}_summary.txt
