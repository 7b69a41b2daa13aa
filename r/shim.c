/*
 * r/shim.c -- .Call() glue between R and libresnmtf_hip.so (include/resnmtf_hip.h).
 *
 * UNTESTED: neither R nor its headers exist in the build container or on the GPU box; this file
 * is the binding a maintainer of eso28599/resnmtf would add under src/ (with `useDynLib(resnmtf)`
 * in NAMESPACE and PKG_LIBS = -lresnmtf_hip).  It only marshals: every number is computed by the
 * HIP library.  The tested stand-in with the identical call sequence is resnmtf_amd/engine.py.
 *
 * R objects in:  data        list of n_v double matrices (already non-negative + normalised)
 *                init_f/s/g  lists of double matrices
 *                phi/xi/psi  n_v x n_v double matrices (already symmetrised, init_rest_mats)
 *                row_maps    list (per view v) of lists (per view w) of integer 2-column
 *                            matrices cbind(idx_v, idx_w), 1-based, or NULL for NA
 *                col_maps    same for columns
 *                n_iters     integer (0 = run to convergence), max_iters integer
 * R object out:  list(output_f, output_s, output_g, All_Error, row_clusters, col_clusters,
 *                     lambda, mu)   -- the fields res_nmtf_inner builds at R/main.r:131-139
 */
#include <R.h>
#include <Rinternals.h>
#include <stdlib.h>

#include "resnmtf_hip.h"

static void fail(resnmtf_handle* h, const char* where) {
  const char* msg = resnmtf_last_error(h);
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s", where, msg ? msg : "?");
  if (h) resnmtf_destroy(h);
  error("%s", buf);                      /* R's stop(); no C++ exception crosses the ABI */
}

static void set_maps(resnmtf_handle* h, SEXP maps, int n_v, int rows) {
  for (int v = 0; v < n_v; ++v) {
    SEXP mv = VECTOR_ELT(maps, v);
    for (int w = 0; w < n_v; ++w) {
      if (w == v) continue;
      SEXP m = VECTOR_ELT(mv, w);
      int rc;
      if (m == R_NilValue) {             /* NA: no shared names (R/utils.r:587-596) */
        rc = rows ? resnmtf_set_shared_rows(h, v, w, -1, NULL, NULL) : resnmtf_set_shared_cols(h, v, w, -1, NULL, NULL);
      } else {
        const int cnt = nrows(m);
        int* iv = (int*)R_alloc(cnt, sizeof(int));
        int* iw = (int*)R_alloc(cnt, sizeof(int));
        for (int t = 0; t < cnt; ++t) { iv[t] = INTEGER(m)[t] - 1; iw[t] = INTEGER(m)[cnt + t] - 1; }
        rc = rows ? resnmtf_set_shared_rows(h, v, w, cnt, iv, iw) : resnmtf_set_shared_cols(h, v, w, cnt, iv, iw);
      }
      if (rc) fail(h, "resnmtf_set_shared");
    }
  }
}

/* init_f / init_s / init_g may be R_NilValue: the initial factors then come from the device
 * (resnmtf_init_svd = init_mats_inner, R/update_steps.r:78-125; k from k_vec_, noise seed from seed_). */
SEXP resnmtf_hip_inner(SEXP data, SEXP init_f, SEXP init_s, SEXP init_g, SEXP phi, SEXP xi, SEXP psi,
                       SEXP row_maps, SEXP col_maps, SEXP n_iters_, SEXP max_iters_, SEXP k_vec_, SEXP seed_) {
  const int n_v = length(data);
  const int device_init = isNull(init_f) || isNull(init_s) || isNull(init_g);
  int* nr = (int*)R_alloc(n_v, sizeof(int));
  int* nc = (int*)R_alloc(n_v, sizeof(int));
  int* kk = (int*)R_alloc(n_v, sizeof(int));
  for (int v = 0; v < n_v; ++v) {
    nr[v] = nrows(VECTOR_ELT(data, v)); nc[v] = ncols(VECTOR_ELT(data, v));
    kk[v] = device_init ? INTEGER(k_vec_)[v] : ncols(VECTOR_ELT(init_f, v));
  }
  resnmtf_handle* h = NULL;
  if (resnmtf_create(n_v, nr, nc, kk, NULL, NULL, &h)) fail(NULL, "resnmtf_create");
  for (int v = 0; v < n_v; ++v) {
    if (resnmtf_set_view(h, v, REAL(VECTOR_ELT(data, v)))) fail(h, "resnmtf_set_view");
    if (device_init) {
      if (resnmtf_init_svd(h, v, (unsigned long long)asInteger(seed_) + (unsigned long long)v, 0.05, 0, NULL))
        fail(h, "resnmtf_init_svd");
    } else if (resnmtf_set_factors(h, v, REAL(VECTOR_ELT(init_f, v)), REAL(VECTOR_ELT(init_s, v)),
                                   REAL(VECTOR_ELT(init_g, v)), NULL, NULL)) {
      /* lambda = mu = NULL: colSums, the explicit-init branch of R/update_steps.r:55-56 */
      fail(h, "resnmtf_set_factors");
    }
  }
  if (resnmtf_set_restrictions(h, REAL(phi), REAL(xi), REAL(psi))) fail(h, "resnmtf_set_restrictions");
  if (n_v > 1) { set_maps(h, row_maps, n_v, 1); set_maps(h, col_maps, n_v, 0); }

  const int n_iters = asInteger(n_iters_), max_iters = asInteger(max_iters_);
  const int cap = n_iters > 0 ? n_iters : max_iters;
  double* errs = (double*)R_alloc(cap, sizeof(double));
  int done = 0;
  if (resnmtf_run(h, n_iters, 1.0e-6, max_iters, errs, cap, &done)) fail(h, "resnmtf_run");

  const char* names[] = {"output_f", "output_s", "output_g", "All_Error", "row_clusters", "col_clusters", "lambda", "mu", ""};
  SEXP out = PROTECT(mkNamed(VECSXP, names));
  SEXP lf = PROTECT(allocVector(VECSXP, n_v)), ls = PROTECT(allocVector(VECSXP, n_v)), lg = PROTECT(allocVector(VECSXP, n_v));
  SEXP lrc = PROTECT(allocVector(VECSXP, n_v)), lcc = PROTECT(allocVector(VECSXP, n_v));
  SEXP llam = PROTECT(allocVector(VECSXP, n_v)), lmu = PROTECT(allocVector(VECSXP, n_v));
  for (int v = 0; v < n_v; ++v) {
    SEXP f = PROTECT(allocMatrix(REALSXP, nr[v], kk[v])), s = PROTECT(allocMatrix(REALSXP, kk[v], kk[v]));
    SEXP g = PROTECT(allocMatrix(REALSXP, nc[v], kk[v]));
    SEXP rc = PROTECT(allocMatrix(REALSXP, nr[v], kk[v])), cc = PROTECT(allocMatrix(REALSXP, nc[v], kk[v]));
    SEXP lam = PROTECT(allocVector(REALSXP, kk[v])), mu = PROTECT(allocVector(REALSXP, kk[v]));
    if (resnmtf_finalise(h, v, REAL(f), REAL(s), REAL(g), REAL(rc), REAL(cc))) fail(h, "resnmtf_finalise");
    if (resnmtf_get_factors(h, v, NULL, NULL, NULL, REAL(lam), REAL(mu))) fail(h, "resnmtf_get_factors");
    SET_VECTOR_ELT(lf, v, f); SET_VECTOR_ELT(ls, v, s); SET_VECTOR_ELT(lg, v, g);
    SET_VECTOR_ELT(lrc, v, rc); SET_VECTOR_ELT(lcc, v, cc); SET_VECTOR_ELT(llam, v, lam); SET_VECTOR_ELT(lmu, v, mu);
    UNPROTECT(7);
  }
  SEXP all_err = PROTECT(allocVector(REALSXP, done));
  for (int t = 0; t < done; ++t) REAL(all_err)[t] = errs[t];
  SET_VECTOR_ELT(out, 0, lf); SET_VECTOR_ELT(out, 1, ls); SET_VECTOR_ELT(out, 2, lg); SET_VECTOR_ELT(out, 3, all_err);
  SET_VECTOR_ELT(out, 4, lrc); SET_VECTOR_ELT(out, 5, lcc); SET_VECTOR_ELT(out, 6, llam); SET_VECTOR_ELT(out, 7, lmu);
  resnmtf_destroy(h);
  UNPROTECT(9);
  return out;
}
