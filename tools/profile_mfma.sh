# PMC pass for MFMA utilisation of the streaming pass (own run, --kernel-trace only: no other tracing)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
RESNMTF_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_summary.txt 2>&1
grep -E "pass_kernel|factor_update" gpurun_out/pmc_mfma_summary.txt
