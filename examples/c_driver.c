/*
 * c_driver.c -- plain C (no C++, no Python) caller of the C-ABI in include/resnmtf_hip.h: what the
 * R-side shim (r/shim.c) does, minus R.  Reads one problem from a binary file, runs the loop of
 * res_nmtf_inner (R/main.r:48-110) on the GPU, writes the returned values.
 *
 *   build: gcc -std=c99 -O2 -Iinclude examples/c_driver.c -Lresnmtf_amd -lresnmtf_hip \
 *              -Wl,-rpath,$PWD/resnmtf_amd -o examples/c_driver
 *   run:   examples/c_driver problem.bin result.bin
 *
 * problem.bin (little endian): int32 n_views, n_iters; per view int32 n, m, k;
 *   phi, xi, psi (each n_views^2 fp64, column-major, already symmetrised);
 *   per view: X (n*m), F (n*k), S (k*k), G (m*k), all fp64 column-major;
 *   per ordered pair (v, w), v != w: int32 count_rows (-1 = NA), idx_v[count], idx_w[count],
 *                                    int32 count_cols,           idx_v[count], idx_w[count].
 * result.bin: fp64 all_err[n_iters]; per view: output_f (n*k), output_s (k*k), output_g (m*k),
 *   row_clusters (n*k), col_clusters (m*k).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "resnmtf_hip.h"

static void die(const char* what, const resnmtf_handle* h) {
  fprintf(stderr, "c_driver: %s: %s\n", what, resnmtf_last_error(h));
  exit(2);
}
static void rd(void* p, size_t size, size_t count, FILE* f) {
  if (count && fread(p, size, count, f) != count) { fprintf(stderr, "c_driver: short read\n"); exit(3); }
}
static double* rd_f64(size_t count, FILE* f) {
  double* p = (double*)malloc((count ? count : 1) * sizeof(double));
  if (!p) { fprintf(stderr, "c_driver: out of memory\n"); exit(3); }
  rd(p, sizeof(double), count, f);
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
  FILE* in = fopen(argv[1], "rb");
  if (!in) { perror(argv[1]); return 1; }
  int32_t nv, n_iters;
  rd(&nv, 4, 1, in); rd(&n_iters, 4, 1, in);
  int* n = (int*)malloc(nv * sizeof(int)); int* m = (int*)malloc(nv * sizeof(int)); int* k = (int*)malloc(nv * sizeof(int));
  for (int v = 0; v < nv; ++v) { int32_t t[3]; rd(t, 4, 3, in); n[v] = t[0]; m[v] = t[1]; k[v] = t[2]; }
  if (resnmtf_abi_version() != RESNMTF_ABI_VERSION) { fprintf(stderr, "c_driver: ABI mismatch\n"); return 2; }

  resnmtf_options opt;
  resnmtf_default_options(&opt);
  resnmtf_handle* h = NULL;
  if (resnmtf_create(nv, n, m, k, NULL, &opt, &h) != RESNMTF_OK) die("resnmtf_create", NULL);

  double* phi = rd_f64((size_t)nv * nv, in); double* xi = rd_f64((size_t)nv * nv, in); double* psi = rd_f64((size_t)nv * nv, in);
  if (resnmtf_set_restrictions(h, phi, xi, psi)) die("set_restrictions", h);
  for (int v = 0; v < nv; ++v) {
    double* X = rd_f64((size_t)n[v] * m[v], in);
    double* F = rd_f64((size_t)n[v] * k[v], in); double* S = rd_f64((size_t)k[v] * k[v], in); double* G = rd_f64((size_t)m[v] * k[v], in);
    if (resnmtf_set_view(h, v, X)) die("set_view", h);
    if (resnmtf_set_factors(h, v, F, S, G, NULL, NULL)) die("set_factors", h);   /* lambda, mu = colSums (R/update_steps.r:55-56) */
    free(X); free(F); free(S); free(G);
  }
  for (int v = 0; v < nv; ++v)
    for (int w = 0; w < nv; ++w) {
      if (v == w) continue;
      for (int axis = 0; axis < 2; ++axis) {
        int32_t count; rd(&count, 4, 1, in);
        int* iv = NULL; int* iw = NULL;
        if (count > 0) {
          iv = (int*)malloc(count * sizeof(int)); iw = (int*)malloc(count * sizeof(int));
          rd(iv, 4, count, in); rd(iw, 4, count, in);
        }
        int rc = axis == 0 ? resnmtf_set_shared_rows(h, v, w, count, iv, iw) : resnmtf_set_shared_cols(h, v, w, count, iv, iw);
        if (rc) die("set_shared", h);
        free(iv); free(iw);
      }
    }
  fclose(in);

  double* err = (double*)malloc((size_t)n_iters * sizeof(double));
  int done = 0;
  if (resnmtf_run(h, n_iters, 1e-6, 0, err, n_iters, &done)) die("resnmtf_run", h);
  if (done != n_iters) { fprintf(stderr, "c_driver: %d of %d sweeps\n", done, n_iters); return 2; }

  FILE* out = fopen(argv[2], "wb");
  if (!out) { perror(argv[2]); return 1; }
  fwrite(err, sizeof(double), n_iters, out);
  for (int v = 0; v < nv; ++v) {
    size_t nf = (size_t)n[v] * k[v], ns = (size_t)k[v] * k[v], ng = (size_t)m[v] * k[v];
    double* F = (double*)malloc(nf * 8); double* S = (double*)malloc(ns * 8); double* G = (double*)malloc(ng * 8);
    double* rc = (double*)malloc(nf * 8); double* cc = (double*)malloc(ng * 8);
    if (resnmtf_finalise(h, v, F, S, G, rc, cc)) die("resnmtf_finalise", h);
    fwrite(F, 8, nf, out); fwrite(S, 8, ns, out); fwrite(G, 8, ng, out); fwrite(rc, 8, nf, out); fwrite(cc, 8, ng, out);
    free(F); free(S); free(G); free(rc); free(cc);
  }
  fclose(out);
  resnmtf_destroy(h);
  printf("c_driver: %d views, %d sweeps, final error %.10g\n", (int)nv, done, err[n_iters - 1]);
  free(err); free(n); free(m); free(k); free(phi); free(xi); free(psi);
  return 0;
}
