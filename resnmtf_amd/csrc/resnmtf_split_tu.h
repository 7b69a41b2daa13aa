// The k <= 16 streaming pass is compiled in a translation unit of its own (resnmtf_pass_k16.hip) with LLVM's max-ILP
// machine scheduler: measured on MI355X (tools/round3/sched_ab*.sh, profiles/r03_sched_ab.txt) the c2 passes run 16.43 / 15.95
// instead of 16.94 / 16.30 us per launch (a sweep 42.2 instead of 43.1 us), while the same strategy costs the k > 16 passes
// 4-6 % -- and the strategy is a per-function property only through the command line.  The list below is every
// pass_kernel<NT = 1, ...> the host side launches: X(NW, UNROLL, IS_XG, MODE_A).
#ifndef RESNMTF_SPLIT_TU_H
#define RESNMTF_SPLIT_TU_H
#ifndef RESNMTF_K16_UNROLL      // loads in flight per trip of the default k <= 16 form (A/B builds: -DRESNMTF_K16_UNROLL=...)
#define RESNMTF_K16_UNROLL 8
#endif
#define RESNMTF_PASS_K16_LIST(X)                                                              \
  X(4, 8, true, true) X(4, 8, true, false) X(4, 8, false, true) X(4, 8, false, false)         \
  X(8, 4, true, true) X(8, 4, true, false) X(8, 4, false, true) X(8, 4, false, false)         \
  X(8, RESNMTF_K16_UNROLL, true, true) X(8, RESNMTF_K16_UNROLL, true, false) X(8, RESNMTF_K16_UNROLL, false, true) X(8, RESNMTF_K16_UNROLL, false, false) \
  X(16, 8, true, true) X(16, 8, true, false) X(16, 8, false, true) X(16, 8, false, false)
#endif
