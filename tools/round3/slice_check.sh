#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sharded.py -x -q -m gpu -k "sliced or convergence" > $out/tests_sliced.log 2>&1; tail -3 $out/tests_sliced.log
timeout -k 10 200 python tools/time_replica_updates.py 8 50000 8000 64 20 --sliced > $out/time_sliced_c5.log 2>&1; tail -2 $out/time_sliced_c5.log
RESNMTF_SLICE_WIDE=1 timeout -k 10 200 python tools/time_replica_updates.py 8 50000 8000 64 20 --sliced > $out/time_sliced_c5_wide.log 2>&1; tail -2 $out/time_sliced_c5_wide.log
timeout -k 10 200 python tools/time_replica_updates.py 4 20000 4000 32 30 --sliced > $out/time_sliced_c4.log 2>&1; tail -2 $out/time_sliced_c4.log
timeout -k 10 200 python tools/time_replica_updates.py 4 20000 4000 32 30 > $out/time_repl_c4.log 2>&1; tail -2 $out/time_repl_c4.log
