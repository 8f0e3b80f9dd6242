# gpirtMCMC -- drop-in for the function of the same name that the reference drivers load
# from the external `gpirt` package (gpirt-sythetic.R:49,88-94).  Same argument names and
# return shape; the sweep runs in libgpirt_hip.so on an MI355X.
# NOT exercised in this repository's tests: R is not available in the build image.
#
#   dyn.load("gpirt_shim.so")   # built from r/gpirt_shim.c against ../gpirt_amd/libgpirt_hip.so
gpirtMCMC <- function(data, sample_iterations, burn_iterations, THIN = 1, CHAIN = 1,
                      beta_prior_means = NULL, beta_prior_sds = NULL,
                      theta_prior_means = NULL, theta_prior_sds = NULL,
                      theta_os = 1, theta_ls = 10, theta_init = NULL, vote_codes = NULL,
                      thresholds = NULL, SEED = 1, constant_IRF = 0, KERNEL = "RBF",
                      beta_proposal_sds = NULL, return_f = TRUE, return_fstar = TRUE) {
  if (length(dim(data)) == 2) data <- array(data, c(dim(data), 1))
  n <- dim(data)[1]; m <- dim(data)[2]; H <- dim(data)[3]
  y <- array(as.integer(data), dim(data)); y[is.na(y)] <- 0L
  C <- max(2L, max(y), if (is.null(thresholds)) 0L else length(thresholds) - 1L)
  pad3 <- function(a) if (is.null(a)) NULL else if (nrow(a) == 2) rbind(a, 0) else a
  beta_prior_means <- pad3(beta_prior_means); beta_prior_sds <- pad3(beta_prior_sds)
  if (!is.null(thresholds)) thresholds[!is.finite(thresholds)] <- 0
  flags <- (if (return_f) 1L else 0L) + (if (return_fstar) 2L else 0L)
  K <- sample_iterations %/% THIN
  P <- if (constant_IRF == 1) 1L else H
  G <- 1001L
  lapply(seq_len(CHAIN), function(ch) {
    r <- .Call("gpirt_run_call", y, as.integer(c(n, m, H, C)),
               as.integer(c(sample_iterations, burn_iterations, THIN, ch - 1L, SEED, constant_IRF,
                            if (KERNEL == "RBF") 0L else 1L, flags)),
               as.double(c(theta_os, theta_ls, 1e-6)),
               if (is.null(theta_init)) NULL else as.double(theta_init),
               if (is.null(beta_prior_means)) NULL else as.double(beta_prior_means),
               if (is.null(beta_prior_sds)) NULL else as.double(beta_prior_sds),
               if (is.null(theta_prior_means)) NULL else as.double(theta_prior_means),
               if (is.null(theta_prior_sds)) NULL else as.double(theta_prior_sds),
               if (is.null(thresholds)) NULL else as.double(thresholds))
    per_draw <- function(v, d) {      # flat [.., K] -> list(K) of arrays, periods broadcast to H
      a <- array(v, c(d, K))
      lapply(seq_len(K), function(k) {
        x <- array(a[seq_len(prod(d)) + (k - 1) * prod(d)], d)
        if (d[length(d)] != H) x <- array(rep(x, H), c(d[-length(d)], H))
        x
      })
    }
    theta <- aperm(array(r[[1]], c(n, H, K)), c(3, 1, 2))          # samples$theta[iter, i, h]
    list(theta = theta,
         f = if (return_f) per_draw(r[[2]], c(n, m, H)) else NULL,
         fstar = if (return_fstar) per_draw(r[[3]], c(G, m, P)) else NULL,
         beta = per_draw(r[[4]], c(3L, m, P)),
         threshold = lapply(per_draw(r[[5]], c(m, C + 1L, P)), function(t) {
           t[, 1, ] <- -Inf; t[, C + 1L, ] <- Inf; t }),
         loglik = r[[6]])
  })
}
