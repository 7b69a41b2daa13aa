/*
 * resnmtf_hip.h -- C-ABI of the MI355X-native ResNMTF multiplicative-update inner loop.
 *
 * The reference (eso28599/resnmtf, pure R) has NO native boundary (NAMESPACE:1-4 has no
 * useDynLib, there is no src/).  The seam is cut around the body of the iteration loop of
 * res_nmtf_inner (R/main.r:48-109): {update_matrices x T, calculate_error x T} plus the
 * post-loop normalisation_check (R/utils.r:176-195) and the binary cluster matrices of
 * obtain_biclusters (R/obtain_bicl.r:162-180).  Each entry point below cites the reference
 * code it replaces.  See INTEGRATION.md for the R-side .Call() binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every call returns 0 on success, non-zero on error; resnmtf_last_error() gives the text.
 *     No exception crosses the ABI.
 *   - all host matrices are fp64 COLUMN-MAJOR (R's native layout): element (i, j) of an
 *     r x c matrix is at [i + j * r].  Vectors are plain fp64 arrays.
 *   - the caller owns every host buffer; the library copies during the call.  The handle owns
 *     all device memory.  A handle is not re-entrant; calls block unless stated otherwise.
 *   - views, rows and columns are 0-based.
 *   - there is NO CPU fallback: every compute entry point fails with RESNMTF_ERR_NO_DEVICE when
 *     no gfx950 device is usable.
 */
#ifndef RESNMTF_HIP_H
#define RESNMTF_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RESNMTF_ABI_VERSION 2   /* 2: options gained wait_mode and the sliced-chain fields, bf16_split = 1 is refused, new phases / selectors / entry points */
#define RESNMTF_MAX_K 64

enum {
  RESNMTF_OK = 0,
  RESNMTF_ERR_INVALID = 1,   /* bad argument / call order */
  RESNMTF_ERR_NO_DEVICE = 2, /* no usable HIP device */
  RESNMTF_ERR_HIP = 3,       /* a HIP runtime call failed */
  RESNMTF_ERR_ALLOC = 4,
  RESNMTF_ERR_STATE = 5      /* handle not prepared / view not owned */
};

/* factor selectors for resnmtf_factor_device_ptr */
enum { RESNMTF_FACTOR_F = 0, RESNMTF_FACTOR_G = 1, RESNMTF_FACTOR_S = 2,
       RESNMTF_FACTOR_FBLOCK = 3, /* replicate_f: the F update's inputs [U (X.G, one f32 slab) | Ma_F | Md_F | lambda], contiguous */
       RESNMTF_FACTOR_FBLOCK_ALL = 4 /* replicate_f: the blocks of ALL views, contiguous in view order (v is ignored beyond its
                                        range check); equal-shaped views have equal block sizes, so one in-place all-gather
                                        over ranks that own one view each refreshes every block */,
       RESNMTF_FACTOR_GBLOCK = 5, /* replicate_gs: the G update's inputs [T (Xt.F, one f32 slab) | Ma_G | Md_G | mu] of view v */
       RESNMTF_FACTOR_GBLOCK_ALL = 6, /* ... of all views, contiguous in view order */
       RESNMTF_FACTOR_SBLOCK = 7, /* replicate_gs: the S update's inputs of view v (fp64: old S | F^T X G | (F^T F S)(G^T G) |
                                     G^T G | F^T F | colSums(G) | colSums(F) | ||X||^2) */
       RESNMTF_FACTOR_SBLOCK_ALL = 8 /* ... of all views, contiguous in view order (equal k: equal blocks).  With equal-shaped
                                        views the S block of a view sits at the end of its F block instead (the address
                                        RESNMTF_FACTOR_SBLOCK returns lies inside RESNMTF_FACTOR_FBLOCK's range): gathering
                                        the F blocks after RESNMTF_PHASE_XG moves both, and this selector is refused */,
       /* slice_chains: the buffers of the four all-to-all exchanges of a sweep (fp32; V equal chunks each, chunk c = what
        * rank c receives / what came from rank c; `v` is ignored beyond its range check) */
       RESNMTF_FACTOR_U_SEND = 9,    /* own view's U = X.G' folded, cut into V row slices          [V][rows_per_slice][KP] */
       RESNMTF_FACTOR_U_RECV = 10,   /* rows of MY slice of every view's U                          (same shape) */
       RESNMTF_FACTOR_FNEW_SEND = 11,/* new F rows of my slice of every view (compact f32)         [V][rows_per_slice][KP] */
       RESNMTF_FACTOR_FNEW_RECV = 12,/* the own view's new F, all rows, as the V slice holders sent them */
       RESNMTF_FACTOR_T_SEND = 13,   /* own view's T = Xt.F' in V column slices, each followed by Ma_G | Md_G (fp64) */
       RESNMTF_FACTOR_T_RECV = 14,
       RESNMTF_FACTOR_GNEW_SEND = 15,
       RESNMTF_FACTOR_GNEW_RECV = 16,
       RESNMTF_FACTOR_F_SLICE = 17,  /* the rows of MY slice of view v's fp64 F (a range inside RESNMTF_FACTOR_F): during a sliced
                                        run these are the only rows of any view's F this rank keeps current -- collect the
                                        slices (all-gather) before reading a whole factor */
       RESNMTF_FACTOR_G_SLICE = 18 };

/* phases of one view's update inside a sweep (R/update_steps.r:282-314) */
enum {
  RESNMTF_PHASE_F = 0, /* update_f (R/update_steps.r:141-165) incl. star_prod_relevant (R/utils.r:63-78) */
  RESNMTF_PHASE_G = 1, /* Xt.F pass, update_g (R/update_steps.r:180-207), X.G' pass; update_s (220-240),
                          update_lm x2 (249-251, 312-313) and calculate_error (R/utils.r:157-166) ride in
                          the X.G' launch */
  RESNMTF_PHASE_S = 2, /* no work: marks the point after which the view's new S may be exchanged */
  RESNMTF_PHASE_F_ALL = 3 /* update_f of EVERY view whose F inputs this handle holds (owned views and, with replicate_f, the
                             others), in view order; `v` only has to be a valid view.  F_w' does not read G or S of this
                             sweep, so hoisting the F updates of a sweep in front of its PHASE_G calls changes nothing.
                             One launch (f_chain_kernel) when k <= 16, the views have equal row counts, share their rows in
                             the same order and at most four of them are owned; one launch per view otherwise.  resnmtf_run
                             hoists the F updates of a sweep the same way when that launch applies */,
  RESNMTF_PHASE_LOCAL_SWEEP = 4 /* RESNMTF_PHASE_F_ALL followed by RESNMTF_PHASE_G of every owned view: everything a rank
                             does between two exchanges of the F blocks when no G or S crosses ranks, in one call */,
  /* replicate_gs (view argument = any owned view for the _ALL phases):
   *   sweep = F_ALL, XTF(own), [all-gather G blocks], G_ALL, XG(own), [all-gather S blocks], S_ALL, [all-gather F blocks] */
  RESNMTF_PHASE_XTF = 5   /* Xt.F pass of owned view v + its k x k job (F'^T F', Ma_G, Md_G) + fold of T into the G block */,
  RESNMTF_PHASE_G_ALL = 6 /* update_g of EVERY view, in view order (R/update_steps.r:180-207, :295-303) from the G blocks; one launch when
                             the views have equal column counts and share their columns in the same order (f_chain_kernel's G form at
                             k <= 16, wide_chain_kernel above), one per view otherwise */,
  RESNMTF_PHASE_XG = 7    /* X.G' pass of owned view v + first half of its k x k job (inputs of the S rule -> S block) +
                             fold of U into the F block */,
  RESNMTF_PHASE_S_ALL = 8 /* update_s, update_lm, error and the F coefficients of EVERY view (s_chain_kernel) */,
  /* slice_chains (view argument = the owned view):
   *   sweep = SLICE_F, [all-to-all: new F rows -> owners], SLICE_XTF, [all-to-all: T slices + G coefficients],
   *           SLICE_G, [all-to-all: new G rows -> owners], SLICE_XG, [all-gather S blocks] S_ALL  ||  [all-to-all: U slices] */
  RESNMTF_PHASE_SLICE_F = 9   /* update_f of EVERY view on MY row slice, in view order (R/update_steps.r:141-165, :282-287) */,
  RESNMTF_PHASE_SLICE_XTF = 10 /* received F rows -> operand copies; Xt.F pass + k x k job of the owned view; T cut into slices */,
  RESNMTF_PHASE_SLICE_G = 11  /* update_g of EVERY view on MY column slice (R/update_steps.r:180-207, :295-303) */,
  RESNMTF_PHASE_SLICE_XG = 12 /* received G rows -> operand copies; X.G' pass + first half of the k x k job (S block); U cut into slices */
};

typedef struct resnmtf_handle resnmtf_handle;

typedef struct resnmtf_options {
  int struct_size;        /* = sizeof(resnmtf_options); set by resnmtf_default_options */
  int device_id;          /* HIP device ordinal; one process per GPU (default 0) */
  void* stream;           /* hipStream_t to enqueue on; NULL = library-owned stream */
  int use_graph;          /* 1 (default): the sweeps of a run are replayed from captured hipGraphs (one graph of the exact
                             length for short fixed runs, else batches of check_every and a power-of-two ladder for the rest) */
  int check_every;        /* convergence mode: sweeps enqueued per host-side check (default 32; launches after the stop
                             test fired return at once) */
  int target_workgroups;  /* workgroup slots a streaming pass is sized for (>= 64, smaller values are refused); 0 = default: CUs x resident workgroups per CU
                             (two at k <= 16, one above) */
  int time_kernels;       /* 1: bracket every streaming-pass launch with HIP events (eager mode) */
  /* tuning overrides of the streaming-pass geometry (0 = automatic), see DESIGN.md section 5 */
  int pass_waves;         /* waves per workgroup: 4, 8 or 16 (0 = auto) */
  int pass_splits_xg;     /* row splits of the X.G pass: 0 (default) = the launch model's choice, else 1 ... 16 (refused beyond) */
  int pass_splits_xtf;    /* row splits of the Xt.F pass: likewise */
  int pass_lds_pad_kb;    /* extra dynamic LDS per workgroup (caps workgroups per CU) */
  int update_blocks;      /* workgroups per factor-update launch; 0 = default: ~160 (512 above 256 MB of X) in hand-off
                             mode A, one round of resident workgroups (CUs x 1 at k > 32, x 2 at k = 32) in mode B */
  int no_pitch_pad;       /* 1: do not pad row pitches that are multiples of 4 KiB (A/B testing) */
  int kk_mode;            /* where the k x k products come from: 0 auto, 1 = A (fp64 partials of the update
                             kernels, job in workgroup 0 of the pass launch), 2 = B (MFMA aux tiles, job in
                             the last-arriving aux workgroup); DESIGN.md section 4 */
  int bf16_split;         /* MFMA form of the two big contractions for k > 16 (k <= 16 always uses the f32 MFMA):
                             0 (default) both operands as THREE bf16 pieces (each rounded to nearest even; X split in
                               registers, the factor's pieces written K-packed by the update kernels), six
                               v_mfma_f32_16x16x32_bf16 per product group in WIDE workgroups (8 or 4 tiles share one
                               LDS copy of the factor block) -- dropped terms <= 2^-26: f32-grade (F / G 2e-7 ... 7e-6
                               from the fp64 reference, the f32 MFMA 1e-7 ... 6e-6);
                             1 refused (RESNMTF_ERR_INVALID): the former two-piece form is retired -- callers that asked
                               for its speed / precision trade must choose 0 or 2 knowingly;
                             2 plain v_mfma_f32_16x16x4_f32, one tile per workgroup */
  int replicate_f;        /* view-sharded use (phase API): 1 = every rank keeps, for EVERY view, the inputs of its F
                             update (X.G folded into one f32 slab, the two k x k coefficient matrices, lambda) in one
                             contiguous exchange block (RESNMTF_FACTOR_FBLOCK) and may run RESNMTF_PHASE_F on views it does
                             not own: the host moves the blocks once per sweep (one all-gather, or one broadcast per
                             view) after the owners' PHASE_G and every rank
                             computes the phi-coupled F chain locally -- identical kernels on identical bytes,
                             so bitwise the same F everywhere -- instead of waiting for N serial F broadcasts */
  int no_f_chain;         /* 1: RESNMTF_PHASE_F_ALL always issues one launch per view (A/B testing) */
  int x_half;             /* (3: as 2, but per view only when the image's relative quantisation error || X~ - X || / || X ||,
                             measured at upload, is at most 3e-5 -- F / G move by 0.2 ... 2 x that error; else the f32 images.)
                             (2: as 1 with UNIFORM 16-bit integers -- one power-of-two step per view -- widened exactly to f32 and
                             multiplied on the f32 MFMA: F / G within 1e-6 ... 3e-5, inside the bar on every problem tried.)
                             1: k <= 16 -- the two passes stream a K-packed fp16 image of X (per-view power-of-two scale, 11-bit
                             mantissa) instead of the f32 one: half the bytes; the factor operand stays f32-grade (two fp16
                             pieces, 22 bits), f32 accumulate.  F / G then sit within ~2e-5 of the fp64 reference instead of
                             ~1e-6 (bar 1e-4); DESIGN.md section 3 */
  int half_unroll;        /* 2-byte passes: wave-steps (16 rows each) per trip: 2, 3, 4 or 6 (0 = default: 4 for fp16, 2 for integers) */
  int replicate_gs;       /* view-sharded use, with replicate_f: 1 = the G and S chains are replicated too.  Every rank keeps,
                             for EVERY view, the inputs of its G update (Xt.F folded into one f32 slab, Ma_G, Md_G, mu:
                             RESNMTF_FACTOR_GBLOCK) and of its S update (old S, F^T X G, (F^T F S)(G^T G), the two Gram
                             matrices, column sums, ||X||^2: RESNMTF_FACTOR_SBLOCK), runs RESNMTF_PHASE_G_ALL /
                             RESNMTF_PHASE_S_ALL for all views itself and only streams its own X (RESNMTF_PHASE_XTF /
                             RESNMTF_PHASE_XG): two all-gathers per sweep (T blocks; U blocks with the S blocks inside --
                             three, S blocks on their own, when the views' F blocks differ in size) instead of
                             2 V ordered broadcasts, and the two streaming passes of all ranks run at the same time
                             (psi / xi coupling; R/update_steps.r:195-204, :231-237).  Needs equal k in all views. */
  int wait_mode;          /* how a fixed-iteration resnmtf_run waits for the device: 0 (default) polls the sweep counter the last
                             k x k job mirrors into pinned host memory -- the call returns when the counter arrives, which can be
                             up to ~100 us BEFORE the stream has drained (every other entry point synchronises first), and the
                             calling thread spins (with pauses) meanwhile, at most 20 ms without progress before it falls back to
                             a stream synchronisation; 1 = hipStreamSynchronize (blocks, ~10-15 us later per call) */
  int slice_chains;       /* view-sharded use, with replicate_f and replicate_gs: 1 = ROW-SLICED chains.  star_prod_relevant is
                             row-local by name (R/utils.r:67-73), so instead of every rank walking the whole F (G) chain of all
                             V views, rank r walks it for the 1 / V of the shared rows (columns) it is assigned
                             (RESNMTF_PHASE_SLICE_F / _G): one view's worth of element-wise work per rank.  The rows travel by
                             all-to-all (the RESNMTF_FACTOR_*_SEND / _RECV buffers), the S chain stays replicated (k x k).  Needs
                             one owned view per handle (view index = slice_index), slice_count = number of views <= 8, equal
                             shapes and k, every coupled pair sharing ALL rows / columns in the same order (identity maps);
                             hand-off mode B is used at every k.  Otherwise fall back to replicate_gs. */
  int slice_index;        /* which slice this handle walks (= the rank) */
  int slice_count;        /* number of slices (= ranks = views) */
  int slice_p2p;          /* with slice_chains (opt-in): the four exchanges of a sweep as PEER STORES + stream-ordered flags instead
                             of collectives -- the chain kernels and slice_pack_kernel store their output straight into the
                             receiving rank's buffers (mapped through hipIpc: xGMI peer access across GPUs), a one-wave kernel then
                             adds one arrival to every rank's counter of that exchange, and the consuming phase begins with a
                             hipStreamWaitValue32 for the V arrivals of its sweep: no collective launch, no host in the loop, no
                             device-side spin.  Set-up: resnmtf_p2p_export on every rank, the handles exchanged by the host
                             (any channel), resnmtf_p2p_import for every rank (the own one included), a host barrier, then ONE
                             resnmtf_prepare.  Receive buffers are single: the order of the sweep itself keeps a writer one
                             exchange behind its reader.  Tested with 2-4 processes on one GPU (IPC on one
                             device); not yet run across GPUs: resnmtf_p2p_selftest tells whether a node can run it.
                             WITHOUT slice_chains (block form; needs replicate_f, one owned view = slice_index, slice_count =
                             number of views <= 8): the exchange blocks of the replicated layouts are stored into every peer's
                             arena instead of all-gathered -- replicate_gs: behind RESNMTF_PHASE_XTF (T rows + G coefficients) and
                             RESNMTF_PHASE_XG (U rows + S block), G_ALL / S_ALL wait for the V arrivals of their sweep;
                             replicate_f alone: the sweep is RESNMTF_PHASE_LOCAL_SWEEP, which waits for the V blocks of the
                             previous sweep, acknowledges them after its F chain and stores its own block after V
                             acknowledgements.  The first blocks (after resnmtf_prepare) travel by the caller's collective,
                             followed by a synchronise + barrier before the first phase.
                             2: as 1 with every wait as a one-wave KERNEL instead of hipStreamWaitValue32 (device counters number
                             the waits, the spin sleeps between polls and is bounded: resnmtf_synchronize reports a wait that gave
                             up) -- kernel nodes only, so whole sweeps can be captured in a graph and replayed by the caller.
                             Bitwise the same results (tested in all three layouts); measured no faster than 1 (the sweep is not
                             bound by the host's launches): opt-in */
  int xcd_order;          /* 1 (opt-in): the main workgroups of the k > 16 passes renumbered so that every XCD works through a
                             contiguous range of the split-major list -- a row split's B block is then fetched into one or two
                             L2s instead of all eight (c5 Xt.F: 154 MB of 1.78 GB per launch).  Measured (tools/round3/xcd_ab.sh):
                             c4 view +0.7 %, c5 X.G pass +1 %, c5 Xt.F pass -5 % (430 against 409 us) -- the passes are not
                             bound by those bytes, and concentrating an XCD on one row range costs more than the re-reads: off */
  int fuse_updates;       /* 0 (default): every factor update is a launch of its own.  1 / 2 (opt-in, k <= 16, hand-off mode A, f32
                             images): an UNCOUPLED update_f / update_g (R/update_steps.r:152-155 / :190-193) runs in the first
                             workgroups of the Xt.F / X.G' launch that consumes it, the other workgroups wait for it on an arrival
                             flag (1: with the first trip of X rows requested before the wait, 2: without).  Bitwise the same
                             results (tested), one launch per update less -- but MEASURED SLOWER on MI355X (c2: 52 - 62 us per
                             sweep against 43): publishing a row block to the other XCDs takes an agent-scope release (an L2
                             write-back), 0.3 - 0.5 us per workgroup and serial within an XCD, so the flag rises 4 - 7 us after the
                             last block is done (profiles/r03_stamps_fused_c2_*.txt) */
} resnmtf_options;

typedef struct resnmtf_pass_timing {
  double xg_ms_total;     /* summed duration of the X.G streaming-pass launches */
  double xtf_ms_total;    /* summed duration of the Xt.F streaming-pass launches */
  long long xg_launches;
  long long xtf_launches;
  double xg_bytes;        /* algorithmic bytes of ONE X.G launch   */
  double xtf_bytes;       /* algorithmic bytes of ONE Xt.F launch */
  double xg_flops;        /* algorithmic flops of ONE X.G launch  */
  double xtf_flops;
} resnmtf_pass_timing;

int resnmtf_abi_version(void);
/* number of visible HIP devices (0 when none; never fails) */
int resnmtf_device_count(void);
void resnmtf_default_options(resnmtf_options* opts);
/* text of the last error on this handle (or of the last failed resnmtf_create when h == NULL) */
const char* resnmtf_last_error(const resnmtf_handle* h);

/*
 * Create a handle for n_views views; view v is n_rows[v] x n_cols[v] with k[v] biclusters
 * (2 <= k <= RESNMTF_MAX_K).  owned[v] != 0 marks the views whose data matrix lives on THIS
 * process' GPU (owned == NULL: all).  Non-owned views only have factor mirrors (F, G, S) that
 * the host refreshes through resnmtf_factor_device_ptr + its own exchange (RCCL broadcast).
 * Replaces: the R lists built by res_nmtf_inner, R/main.r:38-48.
 */
int resnmtf_create(int n_views, const int* n_rows, const int* n_cols, const int* k,
                   const int* owned, const resnmtf_options* opts, resnmtf_handle** out);
int resnmtf_destroy(resnmtf_handle* h);

/*
 * Upload the data matrix of an owned view: x is n x m fp64 column-major, ALREADY non-negative
 * and column-L1-normalised (what check_inputs produces, R/utils.r:416,422).  Also computes
 * data_norms[v] = ||X||_F^2 (R/main.r:48) on the device.
 */
int resnmtf_set_view(resnmtf_handle* h, int v, const double* x);

/*
 * Upload a RAW data matrix and pre-process it on the device, fused into the upload pass:
 * make_non_neg_inner (per COLUMN shift by |min(0, min(column))|, R/utils.r:20-27) followed by
 * matrix_normalisation (divide by colSums, R/utils.r:86-88) -- what check_inputs does at
 * R/utils.r:416,422 -- then data_norms as above.  *was_negative (may be NULL) is set to 1 when an
 * entry was negative, the condition of the reference's warning (R/utils.r:23-25).  A zero column
 * yields NaN, as in the reference.
 */
int resnmtf_set_view_raw(resnmtf_handle* h, int v, const double* x_raw, int* was_negative);

/*
 * View data without a host round trip (the callers of the loop repeat it 36-66 times per apply_resnmtf):
 *   resnmtf_copy_view     device copy of an uploaded view of another handle on the same GPU (same n x m) --
 *                         the k sweep (R/main.r:279-290) factorises ONE data set for every k;
 *   resnmtf_shuffle_view  shuffle_view (R/obtain_bicl.r:11-22): all n m entries of the source view permuted
 *                         pseudo-randomly on the device (Feistel network with cycle walking, `seed`), then
 *                         -- normalise != 0 -- non-negativity shift + column normalisation as apply_resnmtf
 *                         applies to the shuffled data (R/obtain_bicl.r:35 -> R/utils.r:416,422).  R's own
 *                         sample() stream cannot be reproduced.  The reference redraws while a row or a column
 *                         of the shuffled matrix sums to zero (:14-18): resnmtf_view_empty_lines reports that
 *                         condition for the draw just made, the caller redraws with another seed;
 *   resnmtf_subsample_view  the sub-sample X[rows, cols] of stability_repeat (R/stability_analysis.r:230-249; dst's
 *                         shape = the index counts; 0-based indices into the source view), NOT re-normalised,
 *                         exactly as the reference factorises it; all-zero rows / columns
 *                         of the sub-sample (the reference drops them, R/stability_analysis.r:165-190, :233-240)
 *                         are reported by resnmtf_view_empty_lines, masks included;
 *   resnmtf_get_view      the device copy back as fp64 column-major (fp32 precision), e.g. for a host-side
 *                         SVD or for tests.
 */
int resnmtf_copy_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src);
int resnmtf_shuffle_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src, unsigned long long seed,
                         int normalise);
int resnmtf_subsample_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src, const int* rows, const int* cols);
/* Rows / columns of view v's data that summed to exactly zero when it was last drawn on the device
 * (resnmtf_shuffle_view, resnmtf_subsample_view; before any shift / normalisation): counts, and -- if not NULL --
 * 0 / 1 masks of length n and m.  Zero after a host upload (the host has the data). */
int resnmtf_view_empty_lines(resnmtf_handle* h, int v, int* n_empty_rows, int* n_empty_cols, unsigned char* row_mask,
                             unsigned char* col_mask);
int resnmtf_get_view(resnmtf_handle* h, int v, double* x);

/*
 * Initial factors of view v (owned or mirror): F n x k, S k x k, G m x k, column-major.
 * lambda / mu may be NULL: they are then colSums(F) / colSums(G), the reference's
 * explicit-init branch (R/update_steps.r:49-56).
 */
int resnmtf_set_factors(resnmtf_handle* h, int v, const double* F, const double* S,
                        const double* G, const double* lambda, const double* mu);

/*
 * Initial factors of an owned view from its data: init_mats_inner (R/update_steps.r:78-125) --
 * F0 = |U[, 1:k]|, G0 = |V[, 1:k]| of the SVD of X, S0 = |diag(d)[1:k, 1:k]| + |N(0, sigma I)| noise,
 * S0 columns scaled by colSums(F0) * colSums(G0), F0 and G0 column-L1-normalised, lambda / mu their
 * column sums.  The reference calls a full svd(); here the k leading triplets come from a randomized
 * subspace iteration (n_power >= 1 iterations, 0 = default 3; sketch width 16 ceil((k + 8) / 16) <= 64)
 * whose big products are the streaming-pass kernels, and the noise from a std::mt19937_64 seeded with
 * `seed` (R's RNG / MASS::mvrnorm are not reproducible outside R): statistically, not bitwise,
 * equivalent; singular vectors of (near-)equal singular values are determined up to rotation in
 * either implementation.  Views whose short side is smaller than the sketch take an exact route instead
 * (Gram matrix of the short side, Jacobi).  sigma = 0.05 is the reference's default.  singular_values (k, may be NULL)
 * receives d[1:k].  Requires resnmtf_set_view / resnmtf_set_view_raw; replaces resnmtf_set_factors.
 */
int resnmtf_init_svd(resnmtf_handle* h, int v, unsigned long long seed, double sigma, int n_power,
                     double* singular_values);

/*
 * Restriction matrices, n_views x n_views column-major, ALREADY symmetrised with zero diagonal
 * (the output of init_rest_mats, R/update_steps.r:12-24).  NULL = all zero.
 */
int resnmtf_set_restrictions(resnmtf_handle* h, const double* phi, const double* xi,
                             const double* psi);

/*
 * Shared rows (columns) between views v and w, as index pairs: row idx_v[t] of view v carries
 * the same NAME as row idx_w[t] of view w.  This is the integer form of
 * row_indices[[v]][[as.character(w)]] (R/utils.r:560-601) after match() against the row names.
 * count = -1 encodes the reference's NA ("no shared names"): star_prod_relevant then skips
 * the numerator term for w (R/utils.r:70) while update_f still adds phi[w,v]*F to the
 * denominator (R/update_steps.r:158).  Pairs that were never set default to NA.
 * Must be called for (v, w) and for (w, v) separately, as the reference keeps one map per
 * ordered pair.
 */
int resnmtf_set_shared_rows(resnmtf_handle* h, int v, int w, int count, const int* idx_v,
                            const int* idx_w);
int resnmtf_set_shared_cols(resnmtf_handle* h, int v, int w, int count, const int* idx_v,
                            const int* idx_w);

/*
 * Run the loop of res_nmtf_inner (R/main.r:50-109) on a handle that owns every view.
 *   n_iters  > 0 : fixed number of sweeps (R/main.r:83-108).
 *   n_iters == 0 : convergence mode, stop after the first sweep t with
 *                  |mean_err_t - mean_err_{t-1}| <= tol, mean_err_0 = 0 (R/main.r:53-81;
 *                  the reference uses tol = 1e-6), or after max_iters sweeps (a guard the
 *                  reference lacks; max_iters <= 0 means err_capacity).
 * all_err[t] receives mean over views of ||X - F S G^T||_F^2 / ||X||_F^2 after sweep t
 * (All_Error, R/main.r:78,104); err_capacity is the length of all_err.  iters_done receives
 * the number of sweeps executed.  May be called repeatedly; state carries over.
 * Fixed-iteration runs wait on a host-mapped sweep counter (options.wait_mode = 0): the call may return while the stream is
 * still draining the last launch's workgroups, and the calling thread polls meanwhile -- see resnmtf_options.wait_mode;
 * every other entry point, resnmtf_synchronize included, waits for the stream.
 */
int resnmtf_run(resnmtf_handle* h, int n_iters, double tol, int max_iters, double* all_err,
                int err_capacity, int* iters_done);

/* Raw (un-normalised) state, so that a caller can resume exactly.  Any pointer may be NULL. */
int resnmtf_get_factors(resnmtf_handle* h, int v, double* F, double* S, double* G,
                        double* lambda, double* mu);

/*
 * normalisation_check (R/utils.r:176-195) followed by the binary cluster matrices of
 * obtain_biclusters with remove_spurious = FALSE (R/obtain_bicl.r:162-180):
 * row_clusters = 1[F > 1/n][, relations], col_clusters = 1[G > 1/m], relations[j] =
 * which.max(S[, j]).  Does not modify the handle's state.  Any output pointer may be NULL.
 */
int resnmtf_finalise(resnmtf_handle* h, int v, double* F, double* S, double* G,
                     double* row_clusters, double* col_clusters);

/* ---- phase-level entry points (views sharded one-per-GPU; host does the exchange) ---- */

/* Which image of X the streaming passes of view v use after its upload: *uses_2byte = 0 (f32 images), 1 (fp16) or
 * 2 (uniform 16-bit integers); *rel_error = || X~ - X ||_F / || X ||_F of the 2-byte image (0 when none was built).
 * With x_half = 3 this is the guard's decision (resnmtf_options). */
int resnmtf_view_image_info(resnmtf_handle* h, int v, int* uses_2byte, double* rel_error);

/* Size the per-sweep error buffer for `sweeps` sweeps (phase mode; resnmtf_run sizes it itself).
 * Must precede resnmtf_prepare. */
int resnmtf_reserve_sweeps(resnmtf_handle* h, int sweeps);
/* Validate state, build the device coupling tables, reset the sweep counter and run the first
 * X.G pass of every owned view.  Called implicitly by resnmtf_run.  Asynchronous on the stream. */
int resnmtf_prepare(resnmtf_handle* h);
/* Enqueue one phase of owned view v (asynchronous).  `sweep` is the 0-based index of the sweep being
 * executed since resnmtf_prepare (with slice_p2p it also numbers the arrivals a phase waits for: pass the true count).  Within a sweep the caller visits the owned views in index order:
 * PHASE_F, [exchange F], PHASE_G, [exchange G], PHASE_S, [exchange S].  Exchanges enqueued on the
 * handle's stream are ordered against every kernel that reads or writes the exchanged factor. */
int resnmtf_phase(resnmtf_handle* h, int v, int phase, int sweep);
/* Device address and byte size of a factor of view v (fp64 row-major [len][k]; S is [k][k]).
 * The host may overwrite a mirror (non-owned view) with an exchange enqueued on the handle's
 * stream, or read an owned factor as the source of one (S: after PHASE_S). */
int resnmtf_factor_device_ptr(resnmtf_handle* h, int v, int which, void** ptr, size_t* bytes);
/* Per-view relative errors of sweeps [first, first + count) of an owned view (blocking). */
int resnmtf_view_errors(resnmtf_handle* h, int v, int first, int count, double* out);
int resnmtf_synchronize(resnmtf_handle* h);
/* Accumulated streaming-pass timings (options.time_kernels = 1); reset when reset != 0. */
int resnmtf_pass_timings(resnmtf_handle* h, resnmtf_pass_timing* out, int reset);
/* The same for the other kernels of a view-sharded sweep (options.time_kernels = 1, phase API): summed duration (ms) and
 * launch count per kind -- RESNMTF_TIMED_* below; both arrays have RESNMTF_TIMED_KINDS entries. */
enum { RESNMTF_TIMED_XG = 0, RESNMTF_TIMED_XTF = 1, RESNMTF_TIMED_F_CHAIN = 2, RESNMTF_TIMED_G_CHAIN = 3, RESNMTF_TIMED_S_CHAIN = 4,
       RESNMTF_TIMED_PACK = 5 /* folds, slice pack / unpack */, RESNMTF_TIMED_KINDS = 6 };
int resnmtf_kernel_timings(resnmtf_handle* h, double* ms_total, long long* launches, int reset);

/* Phase API, convergence mode (R/main.r:50-81): tol >= 0 makes every phase enqueued from now on a checked one -- the S chain
 * of a sweep (RESNMTF_PHASE_S_ALL; every rank holds the full per-view error table) evaluates
 * |mean_t - mean_{t-1}| <= tol on the device and sets the stop flag, after which every kernel of the phases returns at
 * once; tol < 0 (default) = fixed sweeps.  Identical bytes on every rank: all ranks stop on the same sweep.
 * resnmtf_loop_state synchronises and reports the sweeps closed since resnmtf_prepare, the flag and the sweep count at
 * which it fired (any pointer may be NULL). */
int resnmtf_set_stop_tolerance(resnmtf_handle* h, double tol);
int resnmtf_loop_state(resnmtf_handle* h, int* sweeps_done, int* done, int* stop_sweep);
/* slice_p2p: the IPC handles of this handle's receive buffers and arrival counters (6 x 64 bytes; *bytes receives the size) /
 * map rank `rank`'s (for the own rank `handles` is ignored).  After every rank is imported the slice phases store to the peers. */
int resnmtf_p2p_export(resnmtf_handle* h, void* handles, size_t capacity, size_t* bytes);
int resnmtf_p2p_import(resnmtf_handle* h, int rank, const void* handles, size_t bytes);
/* slice_p2p: called by every rank at about the same time, after all imports and a host barrier, before resnmtf_prepare.
 * Two rounds of: 1 KB of peer stores into every rank's receive buffers + one arrival on every rank's probe counter; the host
 * polls its counter for the V arrivals; the stream wait the phases use, then a KERNEL reads the stored words (the second
 * round re-reads lines the first left in this device's caches: a wait + launch that does not drop them shows here); an
 * acknowledgement round.  Every wait is a host-side poll bounded by timeout_ms (<= 0: 10 s): nothing can hang.  An error
 * return (text in resnmtf_last_error) means this node cannot run the peer-store exchange: create the handles without it. */
int resnmtf_p2p_selftest(resnmtf_handle* h, int timeout_ms);
/* slice_chains: rows / columns per slice (multiples of 32; slice r covers [r * per_slice, min((r + 1) * per_slice, n))). */
int resnmtf_slice_info(resnmtf_handle* h, int* rows_per_slice, int* cols_per_slice);

#ifdef __cplusplus
}
#endif
#endif /* RESNMTF_HIP_H */
