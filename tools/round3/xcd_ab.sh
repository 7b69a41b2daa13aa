#!/bin/bash
out=gpurun_out/r3r; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16_split or full_size_single or consecutive" > $out/t.log 2>&1; tail -2 $out/t.log
for i in 1 2; do
echo "xcd order"; RESNMTF_XCD_ORDER=1 python3 tools/bench_configs.py c4v1 c5v1 2>/dev/null | grep view-updates
echo "plain order"; python3 tools/bench_configs.py c4v1 c5v1 2>/dev/null | grep view-updates
done
