#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out
for i in 1 2 3; do timeout -k 10 400 python -m pytest tests/test_sharded.py -x -q -m gpu -k "peer_stores or sliced_chains" > $out/tests_p2p_$i.log 2>&1; tail -1 $out/tests_p2p_$i.log; done
