#!/bin/bash
out=gpurun_out/r3k; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_sharded.py -x -q -m gpu -k "wide_chain_one_process or replicated_g_and_s or replicated_chains or convergence" > $out/tests_gchain.log 2>&1; tail -3 $out/tests_gchain.log
timeout -k 10 200 python tools/time_replica_updates.py 8 10000 2000 16 50 > $out/time_repl_c2x8.log 2>&1; tail -2 $out/time_repl_c2x8.log
RESNMTF_NO_F_CHAIN=1 timeout -k 10 200 python tools/time_replica_updates.py 8 10000 2000 16 50 > $out/time_repl_c2x8_nochain.log 2>&1; tail -2 $out/time_repl_c2x8_nochain.log
