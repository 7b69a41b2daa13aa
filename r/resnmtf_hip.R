# r/resnmtf_hip.R -- R-side wrapper that keeps res_nmtf_inner()'s signature (R/main.r:32-37) and
# hands the loop (R/main.r:49-109), normalisation_check (R/main.r:110) and the binary cluster
# matrices (R/obtain_bicl.r:162-180) to libresnmtf_hip.so through r/shim.c.
#
# UNTESTED: no R interpreter is available where this repository is built or run.  The tested
# stand-in with the same call sequence is resnmtf_amd/api.py::res_nmtf_inner.
#
# Everything else of the package is unchanged: apply_resnmtf() (naming, reorder_data,
# init_rest_mats, check_inputs), spurious-bicluster removal, bisilhouette and stability selection
# keep calling res_nmtf_inner() and receive the same list fields.

# shared names (character vectors or NA, as produced by produce_indices, R/utils.r:560-601)
# -> 1-based index pairs cbind(idx_v, idx_w); NULL stands for NA
name_maps <- function(indices, names_list) {
  n_v <- length(names_list)
  lapply(seq_len(n_v), function(v) {
    lapply(seq_len(n_v), function(w) {
      if (w == v) return(NULL)
      shared <- indices[[v]][[as.character(w)]]
      if (any(is.na(shared))) return(NULL)
      cbind(match(shared, names_list[[v]]), match(shared, names_list[[w]]))
    })
  })
}

res_nmtf_inner_hip <- function(
    data, row_indices, column_indices,
    init_f = NULL, init_s = NULL, init_g = NULL,
    k_vec = NULL, phi = NULL, xi = NULL, psi = NULL,
    n_iters = NULL, num_repeats = 5, spurious = TRUE, distance = "euclidean",
    no_clusts = FALSE, max_iters = 100000L, device_init = TRUE, seed = 0L) {
  n_v <- length(data)
  # initial factors: explicit ones as given (R/update_steps.r:49-61); otherwise init_mats_inner
  # (R/update_steps.r:78-125) either on the device (resnmtf_init_svd: randomized top-k SVD, milliseconds)
  # or, device_init = FALSE, exactly as the reference builds them (full svd() in R)
  explicit <- !(is.null(init_f) || is.null(init_g) || is.null(init_s))
  if (explicit || !device_init) {
    initial_mats <- init_mats(data, n_v, k_vec, init_f, init_g, init_s)      # R/update_steps.r:36-66
    f0 <- initial_mats$current_f; s0 <- initial_mats$current_s; g0 <- initial_mats$current_g
  } else {
    f0 <- NULL; s0 <- NULL; g0 <- NULL
  }
  row_maps <- name_maps(row_indices, lapply(data, rownames))
  col_maps <- name_maps(column_indices, lapply(data, colnames))
  res <- .Call("resnmtf_hip_inner", data, f0, s0, g0, phi, xi, psi, row_maps, col_maps,
               as.integer(ifelse(is.null(n_iters), 0L, n_iters)), as.integer(max_iters),
               as.integer(k_vec), as.integer(seed))
  for (v in seq_len(n_v)) {                                                  # R/update_steps.r:57-60
    rownames(res$output_f[[v]]) <- rownames(data[[v]])
    rownames(res$output_g[[v]]) <- colnames(data[[v]])
  }
  if (no_clusts) {                                                           # R/main.r:115-120
    return(list(output_f = res$output_f, output_s = res$output_s, output_g = res$output_g))
  }
  # From here on the package's own code runs unchanged on the returned (normalised) factors: obtain_biclusters
  # (R/main.r:122-125 -> R/obtain_bicl.r:151-204) thresholds F > 1/n, G > 1/m, pairs through which.max of S, removes
  # spurious biclusters and scores with bisilhouette -- nothing of it is restated here.  (res$row_clusters /
  # res$col_clusters, the device's thresholded matrices, equal what it computes for spurious = FALSE and are only
  # of use to callers that skip it, e.g. the stability repeats.)
  clusters <- obtain_biclusters(data, res$output_f, res$output_g, res$output_s, num_repeats, spurious, distance)
  error <- if (is.null(n_iters)) mean(utils::tail(res$All_Error, n = 10)) else utils::tail(res$All_Error, n = 1)   # R/main.r:126-130
  list(output_f = res$output_f, output_s = res$output_s, output_g = res$output_g,
       Error = error, All_Error = res$All_Error, bisil = clusters$bisil,
       row_clusters = clusters$row_clustering, col_clusters = clusters$col_clustering,
       lambda = res$lambda, mu = res$mu)                                     # R/main.r:131-139
}
