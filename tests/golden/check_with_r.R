# Cross-check of the golden fixtures against the REFERENCE ITSELF (eso28599/resnmtf), for whoever has R.
#
#     Rscript tests/golden/check_with_r.R [tests/golden/r_export]
#
# NEVER RUN where this repository was built: there is no R interpreter in the build container or on the GPU box
# (SURVEY.md section 8 c1); the fixtures are restatement-derived (tests/golden/make_golden.py).  This script is the
# missing pin: it feeds the same inputs to the package's exported res_nmtf_inner() with EXPLICIT initial factors
# (R/main.r:32-37, R/update_steps.r:49-61 -- the only route on which R's RNG plays no part; apply_resnmtf() cannot
# take list-valued inits, R/utils.r:321-335) and compares F, S, G, the binary cluster matrices and All_Error with
# the fixture's expected outputs.  Expected: agreement to ~1e-12 (both sides are fp64; BLAS summation order differs).
suppressMessages(library(resnmtf))
root <- commandArgs(trailingOnly = TRUE)
root <- if (length(root) >= 1) root[[1]] else file.path("tests", "golden", "r_export")
read_mat <- function(path) unname(as.matrix(read.csv(path, header = FALSE)))
worst <- 0
for (fixture in list.dirs(root, recursive = FALSE)) {
  meta <- read.csv(file.path(fixture, "meta.csv"))
  n_v <- meta$n_views; n_iters <- meta$n_iters
  data <- lapply(seq_len(n_v), function(v) as.matrix(read.csv(file.path(fixture, sprintf("x%d.csv", v)), row.names = 1, check.names = FALSE)))
  init_f <- lapply(seq_len(n_v), function(v) read_mat(file.path(fixture, sprintf("f0_%d.csv", v))))
  init_s <- lapply(seq_len(n_v), function(v) read_mat(file.path(fixture, sprintf("s0_%d.csv", v))))
  init_g <- lapply(seq_len(n_v), function(v) read_mat(file.path(fixture, sprintf("g0_%d.csv", v))))
  phi <- read_mat(file.path(fixture, "phi.csv")); xi <- read_mat(file.path(fixture, "xi.csv")); psi <- read_mat(file.path(fixture, "psi.csv"))
  k_vec <- sapply(init_f, ncol)
  # shared-name maps exactly as apply_resnmtf builds them (R/main.r:230 -> R/utils.r:619-662)
  idx <- resnmtf:::reorder_data(data, n_v, lapply(data, rownames), lapply(data, colnames))
  res <- res_nmtf_inner(data, idx$row_indices, idx$col_indices, init_f, init_s, init_g, k_vec, phi, xi, psi,
                        n_iters = n_iters, spurious = FALSE, no_clusts = FALSE)
  cat(sprintf("== %s (%d views, %d sweeps)\n", basename(fixture), n_v, n_iters))
  rel <- function(a, b) { ok <- !(is.nan(a) & is.nan(b)); sqrt(sum((a[ok] - b[ok])^2)) / max(sqrt(sum(b[ok]^2)), 1e-300) }
  for (v in seq_len(n_v)) {
    d <- c(F = rel(unname(res$output_f[[v]]), read_mat(file.path(fixture, sprintf("out_f%d.csv", v)))),
           S = rel(unname(res$output_s[[v]]), read_mat(file.path(fixture, sprintf("out_s%d.csv", v)))),
           G = rel(unname(res$output_g[[v]]), read_mat(file.path(fixture, sprintf("out_g%d.csv", v)))))
    rc_same <- all(unname(res$row_clusters[[v]]) == read_mat(file.path(fixture, sprintf("rc%d.csv", v))))
    cc_same <- all(unname(res$col_clusters[[v]]) == read_mat(file.path(fixture, sprintf("cc%d.csv", v))))
    cat(sprintf("   view %d: rel-Frobenius F %.2e S %.2e G %.2e; row clusters identical %s, column clusters identical %s\n",
                v, d[["F"]], d[["S"]], d[["G"]], rc_same, cc_same))
    worst <- max(worst, d)
  }
  err <- max(abs(res$All_Error - read_mat(file.path(fixture, "all_error.csv"))[, 1]))
  cat(sprintf("   max |All_Error - expected| = %.2e\n", err))
  worst <- max(worst, err)
}
cat(sprintf("worst deviation over all fixtures: %.3e  (%s)\n", worst, if (worst < 1e-9) "the restatement IS the reference's arithmetic" else "INVESTIGATE"))
