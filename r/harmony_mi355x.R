## harmony_mi355x.R -- R-side stand-in for the Rcpp module object (NOT RUN HERE: no R in the build container).
## Recommended use: as the R file of the companion package `harmonymi355x` (r/make_companion_package.sh assembles it); the reference package
## then changes ONE line, R/ui.R:269 (r/ui_R_269.sed):
##     harmonyObj <- new(harmony)      ->      harmonyObj <- if (requireNamespace("harmonymi355x", quietly = TRUE) && ...) harmonymi355x::new_harmony_mi355x() else new(harmony)
## (In-package alternative: drop this file into the reference's R/ directory, remove `loadModule("harmony_module", TRUE)`
##  (R/harmony-package.R:13), `harmonyObj <- new_harmony_mi355x()` -- INTEGRATION.md says what else that implies.)
## Everything else -- RunHarmony.default (R/ui.R), harmonize() (R/utils.R:15-46), HarmonyConvergencePlot
## (R/utils.R:50-81), the Seurat / SingleCellExperiment methods (R/RunHarmony.R) -- runs unchanged, because the
## returned environment exposes the same `$` names as class_<harmony> (src/harmony.cpp:675-707).

new_harmony_mi355x <- function(seed = NULL, r_rng = FALSE, reference_arithmetic = FALSE) {
    ptr <- .Call("C_hmx_new")
    ## reference_arithmetic: every accumulator follows the reference's fp32 operation order (ridge statistics, O / E tables, objective
    ## sums, closed-form inverse; one GPU) -- for users who need the CPU package's numbers rather than the exact ones.  2: the three groups that move the
    ## result (objective sums, ridge statistics, inverse) over exact O / E tables: 4e-5 from the CPU package at 10^6 cells, at a bit over half the time of TRUE
    if (is.numeric(reference_arithmetic) && reference_arithmetic == 2) .Call("C_hmx_set_int", ptr, "ref_arith", 2)
    else if (reference_arithmetic) .Call("C_hmx_set_int", ptr, "ref_arith", 1)
    if (r_rng) {
        ## exact reference randomness: the library consumes R's own stream (unif_rand) in RcppArmadillo's draw order --
        ## `set.seed(x); RunHarmony(...)` then walks the same seeds and shuffles as the reference package (slower: N draws per round)
        .Call("C_hmx_use_r_rng", ptr)
    } else {
        if (is.null(seed)) seed <- sample.int(.Machine$integer.max, 1)   # one draw from R's RNG => set.seed() still governs the run
        .Call("C_hmx_set_seed", ptr, as.numeric(seed))
    }
    get <- function(field) .Call("C_hmx_get", ptr, field)
    mat <- function(field, nr, nc) matrix(get(field), nrow = nr, ncol = nc)
    obj <- new.env()
    ## ---- methods (src/harmony.cpp:697-707)
    obj$setup <- function(Z, Phi, sigma, theta, lambda, alpha, max_iter_kmeans, epsilon_kmeans, epsilon_harmony,
                          K, block_size, B_vec, batch_proportion_cutoff, verbose) {
        Phi <- methods::as(Phi, "dgCMatrix")
        ## a float::float32 matrix goes down as it is (its @Data integer matrix holds the fp32 bits): no double copy, half the PCIe bytes
        single <- methods::is(Z, "float32")
        invisible(.Call(if (single) "C_hmx_setup_f32" else "C_hmx_setup", ptr, if (single) Z@Data else Z, Phi@i, Phi@p, as.numeric(Phi@x), nrow(Phi), as.numeric(sigma),
                        as.numeric(theta), as.numeric(lambda), alpha, as.integer(max_iter_kmeans), epsilon_kmeans,
                        epsilon_harmony, as.integer(K), block_size, as.integer(B_vec), batch_proportion_cutoff,
                        verbose))
    }
    obj$init_cluster_cpp      <- function() invisible(.Call("C_hmx_init_cluster", ptr))
    obj$cluster_cpp           <- function() .Call("C_hmx_cluster", ptr)
    obj$moe_correct_ridge_cpp <- function() invisible(.Call("C_hmx_moe_correct_ridge", ptr))
    obj$check_convergence     <- function(type) .Call("C_hmx_check_convergence", ptr, as.integer(type))
    obj$compute_objective     <- function() invisible(.Call("C_hmx_compute_objective", ptr))
    ## single = TRUE: a float::float32 matrix (needs the `float` package), fetched as fp32 -- half the bytes, no fp64 copy on the host
    mat32 <- function(field, nr, nc) float::float32(.Call("C_hmx_get_matrix_f32", ptr, field, as.integer(nr), as.integer(nc)))
    obj$getZcorr     <- function(single = FALSE) if (single) mat32("Z_corr", get("d"), get("N")) else mat("Z_corr", get("d"), get("N"))
    obj$getZorig     <- function() mat("Z_orig", get("d"), get("N"))
    obj$getR         <- function() mat("R", get("K"), get("N"))
    obj$getCentroids <- function() mat("Y", get("d"), get("K"))
    obj$getLambda    <- function() mat("Lambda", get("K"), get("B") + 1)
    ## ---- fields (src/harmony.cpp:675-696) as active bindings: fetched from the device on access
    scal <- c("N", "B", "K", "d", "alpha")
    for (f in scal) local({ f <- f; makeActiveBinding(f, function() get(f), obj) })
    vecs <- c("Pr_b", "theta", "sigma", "lambda", "B_vec", "kmeans_rounds", "objective_kmeans", "objective_kmeans_dist",
              "objective_kmeans_entropy", "objective_kmeans_cross", "objective_harmony")
    for (f in vecs) local({ f <- f; makeActiveBinding(f, function() get(f), obj) })
    makeActiveBinding("O", function() mat("O", get("K"), get("B")), obj)
    makeActiveBinding("E", function() mat("E", get("K"), get("B")), obj)
    makeActiveBinding("Y", function() mat("Y", get("d"), get("K")), obj)
    makeActiveBinding("R", function() mat("R", get("K"), get("N")), obj)
    makeActiveBinding("W", function() mat("W", get("W_rows"), get("d")), obj)
    makeActiveBinding("max_iter_kmeans", function(v) {                 # vignettes/detailedWalkthrough.Rmd:364 writes it
        if (missing(v)) get("max_iter_kmeans") else .Call("C_hmx_set_int", ptr, "max_iter_kmeans", as.numeric(v))
    }, obj)
    obj
}
