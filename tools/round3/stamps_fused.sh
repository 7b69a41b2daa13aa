#!/bin/bash
out=gpurun_out/r3f; mkdir -p $out
timeout -k 10 400 python3 tools/stamps_fused.py 2 > $out/stamps_fused_noprefetch.txt 2>&1; tail -14 $out/stamps_fused_noprefetch.txt
timeout -k 10 200 python3 tools/stamps_fused.py 1 > $out/stamps_fused_prefetch.txt 2>&1; tail -14 $out/stamps_fused_prefetch.txt
timeout -k 10 200 python3 tools/stamps.py c2 > $out/stamps_unfused.txt 2>&1; grep -A3 "==" $out/stamps_unfused.txt
