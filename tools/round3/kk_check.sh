#!/bin/bash
out=gpurun_out/r3p; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests_all.log 2>&1; tail -3 $out/tests_all.log
timeout -k 10 200 python tools/time_replica_updates.py 8 50000 8000 64 20 --sliced > $out/time_sliced_c5.log 2>&1; tail -2 $out/time_sliced_c5.log
timeout -k 10 200 python tools/time_replica_updates.py 4 20000 4000 32 30 > $out/time_repl_c4.log 2>&1; tail -2 $out/time_repl_c4.log
python3 tools/bench_configs.py c2 c4v1 c5v1 2>/dev/null | grep view-updates
