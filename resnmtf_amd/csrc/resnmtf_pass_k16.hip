// resnmtf_pass_k16.hip -- the pass_kernel<NT = 1, ...> instantiations (k <= 16: f32 MFMA, one tile per workgroup), compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (resnmtf_amd/build.py); resnmtf_hip.hip declares them `extern template`
// (-DRESNMTF_SPLIT_TU).  Everything else of resnmtf_kernels.hip.inc that is not a template has internal linkage and is
// dropped here unused.  See resnmtf_split_tu.h for the measurement behind this.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "resnmtf_hip.h"
#include "resnmtf_kernels.hip.inc"
#include "resnmtf_split_tu.h"

#define RESNMTF_INSTANTIATE(NW, UNR, XG, MA) template __global__ void pass_kernel<1, NW, UNR, XG, MA, 0>(PassArgs, KKFArgs, KKSArgs);
RESNMTF_PASS_K16_LIST(RESNMTF_INSTANTIATE)
#undef RESNMTF_INSTANTIATE
