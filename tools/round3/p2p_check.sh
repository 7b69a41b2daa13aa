#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sharded.py -x -q -m gpu -k "peer_stores or sliced_chains" > $out/tests_p2p.log 2>&1; tail -3 $out/tests_p2p.log
