#!/bin/bash
# diagnostic: rebuild with a different update-kernel workgroup size and time the sweep
for T in 256 512 1024; do
  sed "s/#define UPDATE_THREADS 1024/#define UPDATE_THREADS $T/" resnmtf_amd/csrc/resnmtf_kernels.hip.inc > /tmp/k_$T.inc
  mkdir -p /tmp/b_$T && cp /tmp/k_$T.inc /tmp/b_$T/resnmtf_kernels.hip.inc && cp resnmtf_amd/csrc/resnmtf_hip.hip /tmp/b_$T/
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -I/tmp/b_$T -o resnmtf_amd/libresnmtf_hip_ut$T.so /tmp/b_$T/resnmtf_hip.hip 2>&1 | grep -E " error" 
done
