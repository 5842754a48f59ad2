/*
 * scanpy_amd.h -- C ABI of libscanpy_amd.so (MI355X / gfx950 kernels for the
 * sc.pp.pca -> sc.pp.neighbors -> sc.tl.leiden path).
 *
 * The reference (scverse/scanpy) is pure Python and has no FFI of its own; the entry
 * points below are the native boundary a binding for that path would call.  Each one
 * cites the reference code whose arithmetic it replaces (paths relative to the scanpy
 * source tree).  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller owns all buffers (inputs, outputs, workspace); the library allocates
 *     nothing that outlives a call and never frees caller memory;
 *   - `stream` is a hipStream_t; work is enqueued on it.  Calls that must report a
 *     data-dependent size (`*_host` outputs) synchronise that stream before returning;
 *   - return value: 0 on success, a negative SCAMD_E* code otherwise, with a
 *     human-readable message available from scamd_last_error() (thread-local);
 *   - no C++ exceptions cross the boundary; one host thread per device.
 *   - row-major everywhere; CSR = (indptr int64[n+1], indices int32[nnz], data f32[nnz]).
 */
#ifndef SCANPY_AMD_H
#define SCANPY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* scamd_stream_t; /* == hipStream_t */

#define SCAMD_OK 0
#define SCAMD_EINVAL (-1)     /* bad argument */
#define SCAMD_EWORKSPACE (-2) /* workspace too small */
#define SCAMD_EHIP (-3)       /* HIP runtime error */
#define SCAMD_EUNSUPPORTED (-4)
#define SCAMD_ECAPACITY (-5)  /* caller-provided output capacity too small */
#define SCAMD_EINTERNAL (-6)  /* internal invariant violated (reported, never hidden) */

#define SCAMD_ABI_VERSION 1

int scamd_abi_version(void);
const char* scamd_last_error(void);
/* Number of HIP devices visible (0 on a CPU-only host); never fails. */
int scamd_device_count(void);

/* ------------------------------------------------------------------------------------------
 * kNN -- exact brute-force Euclidean k-nearest-neighbours of a row range of X against all of X.
 * Replaces sklearn KNeighborsTransformer(algorithm='brute') as called at
 * src/scanpy/neighbors/__init__.py:754-768 (reference `transformer='sklearn'` path) together with
 * the self-column handling of src/scanpy/neighbors/_common.py:74-98 and the in-place diagonal
 * zeroing at neighbors/__init__.py:644.
 *
 *   x        [n, d] float32, row stride ld_x (elements)
 *   queries  rows q_begin .. q_begin+n_query-1 of x (a rank's row shard; the whole matrix on 1 GPU)
 *   k        columns to return.  Column 0 is the query itself with distance exactly 0, columns
 *            1..k-1 its k-1 nearest OTHER rows ordered by (distance, index) -- i.e. what the
 *            reference feeds to the connectivity step for n_neighbors = k.
 *   out_idx  [n_query, k] int32   (-1 where fewer than k-1 other rows exist)
 *   out_dist [n_query, k] float64 Euclidean distance, sqrt of the float64 sum of squared
 *            float64 differences of the float32 inputs (+inf for missing entries)
 *   n_fallback_host  (optional, host) number of queries whose candidate list could not be
 *            certified by the float32 MFMA pass and were recomputed by the float64 scan.
 *   cert_scale  1.0 normally; >1 widens the certification margin (tests use a huge value to force
 *            every query through the float64 fallback scan).
 * Exactness: pass 1 (FP32 MFMA, ||c||^2 - 2 q.c) keeps KP > k candidates per query; pass 2 re-ranks
 * them in float64; a query is accepted only if its k-th exact distance is provably below every
 * rejected candidate given the float32 rounding bound, otherwise pass 3 rescans it in float64.
 * ---------------------------------------------------------------------------------------- */
size_t scamd_knn_workspace_bytes(int64_t n, int d, int64_t n_query, int k);
int scamd_knn_l2_f32(const float* x, int64_t n, int d, int64_t ld_x,
                     int64_t q_begin, int64_t n_query, int k,
                     int32_t* out_idx, double* out_dist,
                     double cert_scale, int64_t* n_fallback_host,
                     void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Approximate variant -- BASELINE.json configs[4] ("IVF-tiled approximate kNN"); the reference's own default above 8192
 * cells is approximate as well (pynndescent, src/scanpy/neighbors/__init__.py:734-739, 769-781).  Same arguments and
 * output conventions as scamd_knn_l2_f32, plus
 *   nprobe   every query sees the rows of the nprobe cells of the k-means quantiser nearest (centroid distance) to its
 *            own cell -- ~2048 rows per cell, at most 1024 cells.  Inside the probed cells the search IS the exact one
 *            (same kernels, float64 re-rank, certificate): the lists are the true nearest neighbours among the probed
 *            rows, recall < 1 comes from unprobed cells only.  nprobe <= 0 or >= the cell count: the exact search.
 * Answered exactly as well: n < 4096, k > 24, d > 64 (shapes outside the register-list kernel).  Workspace:
 * scamd_knn_workspace_bytes (one figure for both entry points). */
int scamd_knn_l2_ivf_f32(const float* x, int64_t n, int d, int64_t ld_x,
                         int64_t q_begin, int64_t n_query, int k, int nprobe,
                         int32_t* out_idx, double* out_dist, int64_t* n_fallback_host,
                         void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Duration (ms, HIP events on `stream`) of the FP32-MFMA selection kernel of the calling thread's most
 * recent scamd_knn_l2_f32 call; -1 if none.  Used by bench.py for the roofline figure. */
float scamd_knn_last_select_ms(void);
/* (query, candidate) pairs that call's selection kernel evaluated: n_query * n for the brute-force sweep, fewer when
 * the cell-pruned search (n >= 65536) could skip far cells (the threshold pre-pass not included); -1 if none.
 * Useful flop = 2 * d * pairs. */
double scamd_knn_last_select_pairs(void);
/* Pairs the same kernel evaluated in its threshold pre-pass (the own cell of every block, scored once more only to seed
 * the list thresholds): executed by the kernel, NOT useful work -- kept out of the roofline's algorithmic flop. */
double scamd_knn_last_select_prepass_pairs(void);
/* Scoring engine of that call's selection kernel: 0 = float32 MFMA (v_mfma_f32_32x32x2_f32, 2 (d + 2) flop per pair),
 * 1 = 3 x bf16 (v_mfma_f32_32x32x16_bf16 on the hi / lo split of the coordinates: 3 * 2 * 64 flop per pair; 32 < d <= 50,
 * SCAMD_KNN_B3=0 disables it); -1 if none.  Either way pass 2 certifies the result in float64. */
int scamd_knn_last_select_engine(void);
/* Queries of that call that the bf16 engine's certificate rejected and the float32 engine re-did (second tier of the
 * pruned search; 0 when the count stayed below SCAMD_KNN_TIER2_MIN, default 256, or the float32 engine ran anyway).
 * `n_fallback_host` of scamd_knn_l2_f32 counts what went to the float64 scan after that. */
int scamd_knn_last_second_tier_queries(void);
/* cells probed by this thread's last search: 0 = it was answered EXACTLY -- also when scamd_knn_l2_ivf_f32 was asked for a
 * shape its register-list kernel does not take (k > 24, d > 64, n < 4096); the Python layer tells the user. */
int scamd_knn_last_nprobe(void);
/* 1 if this thread's last pruned sweep ran with the coarse first stage (hi.hi product first, the other two products only for
 * sub-tiles with a coarse survivor: chosen when the cell bounds prune little; same lists either way). */
int scamd_knn_last_coarse(void);
/* The certificate's error-bound factors of an engine (0 = float32, 1 = 3 x bf16), in units of u = 2^-24:
 *   |score_engine - score_exact| <= u * (cert_k * (||c||^2 + 2 ||q|| ||c||) + cert_k2 * 2 ||q|| ||c||) (+ key_slack * u * |tau|
 * for the slot bits of the list keys).  Read by the test that measures the bound (tests/test_gpu_knn_certificate.py). */
void scamd_knn_cert_factors(int engine, double* cert_k, double* cert_k2, double* key_slack);
/* Test entry: raw scores ||c||^2 - 2 q.c (centred frame) of the bf16 engine for queries [q0, q0 + nq) x candidates
 * [c0, c0 + nc) of x [n, d <= 50] (both counts multiples of 32), through the select kernel's own image packing, operand
 * construction and MFMA chain.  out_scores [nq, nc] float32; out_mu [128] (the column means the image was centred with) and
 * out_cmax [1] (largest ||x - mu||^2) may be NULL.  Not part of the search; measures the engine's arithmetic error. */
size_t scamd_knn_debug_b3_scores_workspace_bytes(int64_t n);
int scamd_knn_debug_b3_scores_f32(const float* x, int64_t n, int d, int64_t ld_x, int64_t q0, int nq, int64_t c0, int nc,
                                  float* out_scores, float* out_mu, float* out_cmax, void* workspace,
                                  size_t workspace_bytes, scamd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fuzzy simplicial set -- umap connectivities from a kNN result.
 * Replaces umap.umap_.fuzzy_simplicial_set(..., set_op_mix_ratio=1, local_connectivity=1) as called
 * at src/scanpy/neighbors/_connectivity.py:103-138 (smooth_knn_dist + compute_membership_strengths
 * + W + W^T - W o W^T, zeros eliminated, CSR with sorted column indices).
 *
 *   knn_idx  [n, k] int32, column 0 = the row itself; knn_dist [n, k] float32
 *   out_indptr [n+1] int64; out_indices/out_data: capacity `cap` entries (2*n*(k-1) always suffices)
 *   out_sigma, out_rho [n] float32 (optional, may be NULL)
 *   nnz_host  (host) number of stored entries written
 * ---------------------------------------------------------------------------------------- */
size_t scamd_fuzzy_workspace_bytes(int64_t n, int k);
int scamd_fuzzy_simplicial_set_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n, int k,
                                   int64_t* out_indptr, int32_t* out_indices, float* out_data,
                                   int64_t cap, float* out_sigma, float* out_rho, int64_t* nnz_host,
                                   void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Row-sharded fuzzy simplicial set (one process per GPU, rows [row_begin, row_begin + n_local) of n_total; knn_idx holds
 * GLOBAL row ids).  weights: sigma / rho / membership strengths of the local rows -- w [n_local, k] (0 = absent: the self
 * column, padding), out_count[i] = entries of row i with w > 0; sum_all_dev = device double holding the sum of ALL
 * n_total * k distances (the caller all-reduces it; used only for the rows without a positive distance, as in
 * umap.smooth_knn_dist).  merge_rows: the symmetrisation C = W + W^T - W o W^T of the local rows given their in-edges
 * (in_indptr [n_local + 1], in_src ascending within a row, in_w: the (j, i, w_ji) triples the other ranks sent for
 * these rows); bit-identical to the rows scamd_fuzzy_simplicial_set_f32 produces on one device.  cap >= nnz of the
 * local rows (<= n_local * (k - 1) + number of in-edges). */
int scamd_fuzzy_weights_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n_local, int k, int64_t row_begin,
                            int64_t n_total, const double* sum_all_dev, float* w, float* out_sigma, float* out_rho,
                            int32_t* out_count, scamd_stream_t stream);
size_t scamd_fuzzy_merge_workspace_bytes(int64_t n_local, int64_t cap);
int scamd_fuzzy_merge_rows_f32(const int32_t* knn_idx, const float* w, int64_t n_local, int k, const int64_t* in_indptr,
                               const int32_t* in_src, const float* in_w, int64_t* out_indptr, int32_t* out_indices,
                               float* out_data, int64_t cap, int64_t* nnz_host, void* workspace, size_t workspace_bytes,
                               scamd_stream_t stream);
/* method='gauss' on the kNN pattern (src/scanpy/neighbors/_connectivity.py:21-100, CSR branch): sigma_i^2 = median of the
 * squared distances to the row's neighbours, w_ij = sqrt(2 s_i s_j / (s_i^2 + s_j^2)) exp(-d_ij^2 / (s_i^2 + s_j^2)),
 * w_ji := w_ij where i is not among j's neighbours.  Same in/out conventions and workspace size as the fuzzy set. */
int scamd_gauss_connectivities_f32(const int32_t* knn_idx, const float* knn_dist, int64_t n, int k, int64_t* out_indptr,
                                   int32_t* out_indices, float* out_data, int64_t cap, int64_t* nnz_host,
                                   void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* method='jaccard' (PhenoGraph weights, _connectivity.py:141-186): |N(i) & N(j)| / (2 (k-1) - |N(i) & N(j)|) on the kNN
 * pattern without the self columns, symmetrised by averaging. */
int scamd_jaccard_connectivities_f32(const int32_t* knn_idx, int64_t n, int k, int64_t* out_indptr,
                                     int32_t* out_indices, float* out_data, int64_t cap, int64_t* nnz_host,
                                     void* workspace, size_t workspace_bytes, scamd_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * PCA building blocks on a CSR float32 matrix (n rows = cells, g columns = genes).
 * Together they replace sklearn PCA(svd_solver='arpack')._fit_truncated on sparse input
 * (sklearn/decomposition/_pca.py:704-793 as called at src/scanpy/preprocessing/_pca/__init__.py:
 * 287-308): mean_variance_axis, the implicitly centred operator X - 1 mu^T
 * (sklearn/utils/sparsefuncs.py:718-742; in-repo restatement _pca/_compat.py:43-56) applied to
 * blocks of vectors, and the Gram-matrix alternative of _pca/_kernels.py:14-58.
 * ---------------------------------------------------------------------------------------- */
/* per-row sum and sum of squares in float64 (fixed summation order).  Applied to the CSC copy
 * (rows = genes) it yields the column statistics of mean_variance_axis. */
int scamd_csr_row_stats_f32(const int64_t* indptr, const float* data, int64_t n_rows,
                            double* row_sum, double* row_sumsq, scamd_stream_t stream);
/* Deterministic (stable) CSR -> CSC: t_indptr int64[g+1], t_indices int32[nnz] (row ids, ascending
 * within a column), t_data f32[nnz]. */
size_t scamd_csr_transpose_workspace_bytes(int64_t n, int64_t g, int64_t nnz);
int scamd_csr_transpose_f32(const int64_t* indptr, const int32_t* indices, const float* data,
                            int64_t n, int64_t g, int64_t nnz,
                            int64_t* t_indptr, int32_t* t_indices, float* t_data,
                            void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Y[n, l] (float32) = A[n, g] * B[g, l] - 1 * shift[l]^T   (shift may be NULL).
 * A is CSR; B row-major float32 with leading dimension l; 1 <= l <= 256 (scamd_spmm_csr_f32; the float64-accumulating variant takes l <= 128). */
int scamd_spmm_csr_f32(const int64_t* indptr, const int32_t* indices, const float* data,
                       int64_t n, int64_t g, const float* b, int l, const float* shift,
                       float* y, scamd_stream_t stream);
/* W[n_rows, l] (float64) = A * B with float64 accumulation of float32 products in a fixed order
 * (A = the CSC copy viewed as a g x n CSR, B = Y[n, l] float32: W = X^T Y); optional rank-1
 * correction W -= scale[n_rows] * colsum[l]^T (scale = column means, colsum = 1^T Y; both float64,
 * both NULL to skip). */
size_t scamd_spmm_f64acc_workspace_bytes(int64_t n_rows, int64_t nnz, int l);
int scamd_spmm_csr_f32_f64acc(const int64_t* indptr, const int32_t* indices, const float* data,
                              int64_t n_rows, int64_t nnz, const float* b, int l,
                              const double* scale, const double* colsum, double* w,
                              void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Exact Gram matrix and column sums of a CSR float32 matrix in 64-bit fixed point (the covariance route of
 * src/scanpy/preprocessing/_pca/_kernels.py:14-58 + _pca/_dask.py:28-132, and the engine of the default PCA solve):
 *   gram[i * ld_gram + j] = round-to-nearest-per-product sum_r x_ri * x_rj * 2^scale_bits   (int64, symmetric,
 *                           [g_pad x ld_gram], g_pad = g rounded up to 128, padding rows/columns zero)
 *   colsum[j]             = sum_r round(x_rj * 2^scale_bits)                                 (int64 [g_pad])
 * Integer accumulation is order independent: the result is bitwise reproducible and additive over row shards
 * (ranks all-reduce the int64 arrays).  Two-phase use: call with gram == NULL and absmax_host != NULL to get
 * max|x| (host float) and derive scale_bits <= 62 - log2(n_total * absmax^2); then call with the outputs. */
size_t scamd_csr_gram_workspace_bytes(int64_t n, int64_t g);
int scamd_csr_gram_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                       int64_t nnz, int scale_bits, int64_t* gram, int64_t ld_gram, int64_t* colsum,
                       float* absmax_host, void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Top-k eigenpairs of a symmetric POSITIVE SEMI-DEFINITE float64 matrix a [g x g] (leading dimension lda), the dense half
 * of the Gram-route PCA: what sklearn's PCA(svd_solver='arpack') obtains from ARPACK on the implicitly centred matrix
 * (sklearn/decomposition/_pca.py:704-793, called at src/scanpy/preprocessing/_pca/__init__.py:287-308) and the reference's
 * covariance route from `eigh` (src/scanpy/preprocessing/_pca/_dask.py:28-132).  Chebyshev-filtered subspace iteration on a
 * block of <= 128 vectors; hand-written float64 MFMA GEMMs, CholeskyQR2, one-workgroup Jacobi (csrc/dense.hip).
 *   lam [k] descending; v [g x k] row-major, column j = unit eigenvector j (sign unspecified)
 *   tol: Ritz residual |A v - lam v| / lam_0 of the k pairs (2e-8 reproduces ARPACK-level loadings)
 *   info_host (optional, 6 ints): outer iterations, operator applications, block size, Cholesky retries, then the final
 *             residual as a float64 in [4..5]
 * Supported: g <= 128 (full decomposition) or g >= 256 with k <= 96; SCAMD_EUNSUPPORTED otherwise or when the block turns
 * out numerically rank deficient / the iteration does not converge (callers then use a full eigendecomposition). */
size_t scamd_eigh_topk_workspace_bytes(int64_t g, int k);
int scamd_eigh_topk_f64(const double* a, int64_t g, int64_t lda, int k, uint64_t seed, double tol, double* lam, double* v,
                        int32_t* info_host, void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Building blocks of the above on caller buffers (tests): op 1: out0[m x n] = in0^T in1 with in0 [kdim x m], in1 [kdim x n];
 * op 2: out0[m x m] = CholeskyQR factor S of the Gram matrix in0 [m x m] (Z S orthonormal when in0 = Z^T Z), *flag_host =
 * 1 on a non-positive pivot; op 3: out0[m] = eigenvalues (descending), out1[m x m] = eigenvectors (columns) of the
 * symmetric PSD in0 [m x m], *flag_host = Jacobi sweeps.  m <= 128 for ops 2 and 3. */
int scamd_dense_debug_f64(int op, const double* in0, const double* in1, int m, int n, int kdim, double* out0, double* out1,
                          int32_t* flag_host, scamd_stream_t stream);
/* sc.pp.pca(svd_solver='arpack' accuracy) on a resident CSR float32 matrix in ONE call (the `pca_csr_f32` entry of the
 * boundary: src/scanpy/preprocessing/_pca/__init__.py:53-384 without its AnnData handling): max|x| -> exact fixed-point
 * Gram matrix -> top-n_comps eigenpairs -> sklearn's sign convention (svd_flip, u_based_decision=False) -> scores.
 *   scores [n x n_comps] float32 = X V - 1 (mu^T V) (zero_center) or X V
 *   components [n_comps x g] float64 (rows = components_), variance / variance_ratio [n_comps] (explained_variance_,
 *   explained_variance_ratio_ of sklearn PCA, or of TruncatedSVD when zero_center = 0), mean [g] float64
 *   info_host (optional, 8 ints): as scamd_eigh_topk_f64, then [6] = scale_bits of the fixed-point Gram matrix
 * Single device, g <= 8192, n_comps <= 96; a row-sharded run all-reduces the int64 Gram matrix between
 * scamd_csr_gram_f32 and scamd_eigh_topk_f64 instead. */
size_t scamd_pca_csr_workspace_bytes(int64_t n, int64_t g, int n_comps);
int scamd_pca_csr_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g, int64_t nnz,
                      int n_comps, int zero_center, uint64_t seed, double tol, float* scores, double* components,
                      double* variance, double* variance_ratio, double* mean, int32_t* info_host, void* workspace,
                      size_t workspace_bytes, scamd_stream_t stream);
/* The dense half of the above on its own -- what a ROW-SHARDED run calls after the all-reduce of the int64 sums (and what
 * scamd_pca_csr_f32 itself calls), so that the model is bitwise the same for any number of ranks and no library computes
 * any part of it (replaces the covariance-eigh arithmetic of src/scanpy/preprocessing/_pca/_kernels.py:14-58 and
 * _pca/_dask.py:28-132, sklearn's svd_flip and explained-variance bookkeeping, sklearn/decomposition/_pca.py:704-793):
 *   gram [g x g] int64 fixed point (2^scale_bits X^T X summed over ALL n_total rows, row stride ld_gram; only the upper
 *   triangle is read), colsum [g] int64 (2^scale_bits column sums)  ->  components [n_comps x g] float64 (sign convention
 *   applied), loadings_f32 [g x n_comps] float32 (= the B operand of scamd_spmm_csr_f32 for the scores), shift [n_comps]
 *   float32 (mu^T V: its `shift` operand), variance / variance_ratio [n_comps], mean [g], eigenvalues [n_comps] (of
 *   X^T X - n mu mu^T, clamped at 0: singular values squared; may be NULL).  info_host (optional, 8 ints) as scamd_pca_csr_f32.
 * Same range as scamd_eigh_topk_f64 (SCAMD_EUNSUPPORTED outside). */
size_t scamd_pca_solve_gram_workspace_bytes(int64_t g, int n_comps);
int scamd_pca_solve_gram_f64(const int64_t* gram, int64_t ld_gram, const int64_t* colsum, int64_t n_total, int64_t g,
                             int scale_bits, int n_comps, int zero_center, uint64_t seed, double tol, double* components,
                             float* loadings_f32, float* shift, double* variance, double* variance_ratio, double* mean,
                             double* eigenvalues, int32_t* info_host, void* workspace, size_t workspace_bytes,
                             scamd_stream_t stream);
/* Spectral initialisation of the UMAP layout: the `dim` eigenvectors of the symmetric normalised adjacency
 * S = D^-1/2 A D^-1/2 that follow the trivial one (= the smallest non-trivial ones of the normalised Laplacian) -- what
 * `sc.tl.umap(init_pos='spectral')` gets from umap-learn's `spectral_layout` (src/scanpy/tools/_umap.py:165-215; ARPACK
 * there, Chebyshev-filtered subspace iteration on (S + I) / 2 here, float32 SpMM operand, float64 everywhere else).
 *   indptr / indices / weights: the symmetric fuzzy graph (device CSR, n x n); dim <= 10; n > dim + 6
 *   out [n x dim] float64 (device): unit-norm, mutually orthogonal, orthogonal to sqrt(deg)
 *   info_host (optional, 8 doubles): outer iterations, operator applications, residual of the wanted Ritz pairs,
 *             1 if it is below `tol` (2e-6 is what the float32 operand allows), then up to four eigenvalues of S
 *   max_outer / max_degree: bounds of the iteration (60 / 64 in the Python layer)
 * SCAMD_EUNSUPPORTED when the block cannot be orthonormalised (the caller falls back to a random layout, as umap-learn does
 * when its eigensolver fails). */
size_t scamd_spectral_embedding_workspace_bytes(int64_t n, int64_t nnz, int dim);
int scamd_spectral_embedding_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                 int64_t nnz, int dim, uint64_t seed, double tol, int max_outer, int max_degree,
                                 double* out, double* info_host, void* workspace, size_t workspace_bytes,
                                 scamd_stream_t stream);
/* colsum[l] (float64) = 1^T Y for Y [n, l] float32, fixed summation order. */
size_t scamd_colsum_workspace_bytes(int l);
int scamd_colsum_f32_f64(const float* y, int64_t n, int l, double* colsum,
                         void* workspace, size_t workspace_bytes, scamd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Leiden community detection on a symmetric weighted CSR graph.
 * Replaces leidenalg.find_partition(g, RBConfigurationVertexPartition, weights, n_iterations,
 * resolution_parameter, seed) / igraph Graph.community_leiden(objective_function='modularity')
 * as called at src/scanpy/tools/_leiden.py:167-196 (graph built by
 * src/scanpy/_utils/__init__.py:278-304).  Objective:
 *   Q = 1/(2m) sum_ij (A_ij - resolution * k_i k_j / (2m)) delta(c_i, c_j).
 *   membership [n] int32: consecutive ids, renumbered by decreasing community size
 *   modularity_host, n_communities_host: host outputs
 *   n_iterations < 0: iterate until an iteration changes nothing.
 * ---------------------------------------------------------------------------------------- */
size_t scamd_leiden_workspace_bytes(int64_t n, int64_t nnz);
int scamd_leiden_csr_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                         int64_t n, int64_t nnz, double resolution, int n_iterations,
                         double beta, uint64_t seed,
                         int32_t* membership, double* modularity_host, int32_t* n_communities_host,
                         void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* The same optimiser started from a given partition instead of singletons -- `initial_membership` of
 * leidenalg.find_partition / igraph community_leiden, which the reference passes through `**clustering_args`
 * (src/scanpy/tools/_leiden.py:66, 174-196).  initial_membership [n] int32 on the device, ids in [0, n) (SCAMD_EINVAL
 * otherwise); n_iterations = 0 returns it renumbered (by decreasing size) with its modularity. */
int scamd_leiden_csr_init_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                              int64_t n, int64_t nnz, double resolution, int n_iterations,
                              double beta, uint64_t seed, const int32_t* initial_membership,
                              int32_t* membership, double* modularity_host, int32_t* n_communities_host,
                              void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* ... with the objective named: 0 = modularity (the two entry points above), 1 = CPM -- igraph's
 * community_leiden(objective_function='CPM') as reached from sc.tl.leiden(flavor='igraph', objective_function='CPM')
 * (src/scanpy/tools/_leiden.py:188-196): every vertex weighs 1, the resolution is not divided by 2m.
 * initial_membership may be NULL.  With objective 1 *modularity_host is the resolution-1 modularity of the returned
 * partition (the reference stores part.modularity, :219), not the CPM quality the run maximised. */
int scamd_leiden_csr_ex_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                            int64_t n, int64_t nnz, double resolution, int n_iterations,
                            double beta, uint64_t seed, int objective, const int32_t* initial_membership,
                            int32_t* membership, double* modularity_host, int32_t* n_communities_host,
                            void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* ... CPM with the caller's vertex weights -- igraph `community_leiden(objective_function='CPM', node_weights=...)`, which
 * `sc.tl.leiden(flavor='igraph', objective_function='CPM', node_weights=...)` reaches through **clustering_args
 * (src/scanpy/tools/_leiden.py:66, 188-196): a community pays resolution * (sum of its members' weights)^2.
 * node_weights: device pointer, n floats in [0, 1e6] (SCAMD_EINVAL otherwise; held in fixed point, 16 fractional bits) or
 * NULL (every vertex weighs 1: scamd_leiden_csr_ex_f32).  objective 0 (modularity: the vertex weights are the strengths)
 * takes NULL only (SCAMD_EUNSUPPORTED). */
int scamd_leiden_csr_nw_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                            int64_t n, int64_t nnz, double resolution, int n_iterations,
                            double beta, uint64_t seed, int objective, const float* node_weights,
                            const int32_t* initial_membership, int32_t* membership, double* modularity_host,
                            int32_t* n_communities_host, void* workspace, size_t workspace_bytes,
                            scamd_stream_t stream);
/* Statistics of the last scamd_leiden_csr_f32 call on this thread:
 *   [0] outer iterations run, [1] kernel launches, [2] blocking host round trips,
 *   [3] full sweeps / [4] rounds / [5] vertices moved by the final polish (n_iterations < 0: strictly monotone
 *       single-vertex moves until a sweep over ALL vertices finds no improving one -- the node optimality a stable
 *       partition of leidenalg / igraph has, src/scanpy/tools/_leiden.py:166-196 with n_iterations=-1),
 *   [6] 1 if the polish was skipped because the last iteration itself had proven node optimality,
 *   [7] levels of the first iteration,
 *   [8] local-moving sweeps of the levels that run as separate kernels, [9] their algorithmic traffic in MB (active rows
 *       x (12 B per entry + 16 B per vertex): SURVEY.md 8(d)'s per-sweep figure over the rows a sweep visits),
 *   [10] communities the polish split off (a departing vertex had cut them in two), [11] 1 if the cap on the outer iterations
 *        ([13]; 32 unless SCAMD_LEIDEN_ITER_CAP says otherwise) ended an n_iterations < 0 run instead of convergence --
 *        `tl.leiden` turns it into a UserWarning, [12] 1 if a polish pass stopped at its round cap (node optimality then
 *        unproven; warned about as well), [14] device-to-device copies (0 since round 6: partitions change buffers by
 *        pointer), [15] launches of the multi-region clear kernel (all that is left of the hipMemsetAsync calls).
 * out[0 .. min(n, 16)).  Diagnostics only (bench.py, tools/, the two warnings). */
void scamd_leiden_last_stats(int32_t* out, int n);
/* Test entry: the component split the polish applies after its moves -- every connected component (over the stored
 * entries) of a community of `membership` (ids in [0, n), device, rewritten in place) becomes a community of its own, id =
 * its smallest vertex; *n_split_host = components - communities (0: membership untouched).  Workspace:
 * scamd_leiden_workspace_bytes. */
int scamd_leiden_debug_split_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                                 int64_t n, int64_t nnz, int32_t* membership, int32_t* n_split_host,
                                 void* workspace, size_t workspace_bytes, scamd_stream_t stream);
/* Modularity of a given membership (replaces igraph Graph.modularity as used by
 * src/scanpy/metrics/_metrics.py:202-214). */
int scamd_modularity_csr_f32(const int64_t* indptr, const int32_t* indices, const float* weights,
                             int64_t n, int64_t nnz, const int32_t* membership, double resolution,
                             double* modularity_host, void* workspace, size_t workspace_bytes,
                             scamd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Upstream normalisation chain: normalize_total -> log1p -> highly_variable_genes -> scale on a
 * CSR float32 matrix resident on the device.  One HBM-bound sweep each; no workspace (the caller owns every output).
 * A dense matrix is passed as a CSR with the full pattern (indptr[i] = i*g, indices = column ids).
 * nnz = number of stored entries (= indptr[n]); the row-wise kernels size their lanes-per-row from nnz / n.
 * ---------------------------------------------------------------------------------------- */
/* out[r] = float(sum_j x_rj accumulated in float64): the CSR branch of _normalize_csr
 * (src/scanpy/preprocessing/_normalization.py:40-46).  col_skip != NULL: entries of columns with col_skip[c] != 0
 * are left out (second pass of exclude_highly_expressed, :60-65). */
int scamd_pp_row_sums_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t nnz,
                          const int32_t* col_skip, float* out, scamd_stream_t stream);
/* out[r] = #{stored x_rj > 0}: `stats.sum(data > 0, axis=1)` of filter_cells(min_genes= / max_genes=)
 * (src/scanpy/preprocessing/_simple.py:170-172); the per-gene counterpart of filter_genes is npos of
 * scamd_pp_col_stats_f32 */
int scamd_pp_row_count_positive_f32(const int64_t* indptr, const float* data, int64_t n, int64_t nnz, int32_t* out,
                                    scamd_stream_t stream);
/* col_counts[c] = #{r : x_rc > max_fraction * row_total[r]}  (_normalization.py:47-58); col_counts [g] is zeroed here */
int scamd_pp_count_high_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                            int64_t nnz, const float* row_total, float max_fraction, int32_t* col_counts,
                            scamd_stream_t stream);
/* x_rj /= factor[r] in float32, factor == 0 treated as 1: axis_mul_or_truediv(x, counts_per_cell, op=truediv,
 * allow_divide_by_zero=False, axis=0) (src/scanpy/_utils/__init__.py:623-659, called at _normalization.py:121-123) */
int scamd_pp_row_divide_f32(const int64_t* indptr, float* data, int64_t n, int64_t nnz, const float* factor,
                            scamd_stream_t stream);
/* data[i] = log1p(data[i]) (/ log(base) when base > 0; base == 0 means natural log): log1p_array / log1p_sparse
 * (src/scanpy/preprocessing/_simple.py:359-379) */
int scamd_pp_log1p_f32(float* data, int64_t count, double base, scamd_stream_t stream);
/* Per-gene sum[c], sumsq[c] (float64) and npos[c] = #{stored x_rc > 0} over the rows with row_mask[r] != 0
 * (row_mask == NULL: all rows); outputs [g] are zeroed here.  transform 0: x as stored; 1: expm1(x * tscale) in
 * float32 (flavor 'seurat' works in counts space, _highly_variable_genes.py:402-410).  Feeds
 * fast_array_utils.stats.mean_var(x, axis=0, correction=1) (_highly_variable_genes.py:413, _scale.py:189) and
 * filter_genes(min_cells=1) (_highly_variable_genes.py:387-395).  Float64 atomics: sums agree with a sequential
 * float64 sum to ~1e-15 relative, not bitwise. */
int scamd_pp_col_stats_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                           int64_t nnz, const uint8_t* row_mask, int transform, double tscale, double* sum,
                           double* sumsq, uint64_t* npos /* may be NULL */, scamd_stream_t stream);
/* The same sweep with every value clipped from above at clip[gene] (float64) before it is summed: `clip_square_sum` of
 * flavor='seurat_v3' (src/scanpy/preprocessing/_highly_variable_genes.py:75-115: sum and sum of squares of
 * min(x, sigma_hat * sqrt(n) + mu) per gene). */
int scamd_pp_col_stats_clip_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                                int64_t nnz, const uint8_t* row_mask, const double* clip, double* sum, double* sumsq,
                                scamd_stream_t stream);
/* zero_center=False: x_rc = min(max_value, x_rc / std[c]) on the masked rows, sparsity kept: scale_and_clip_csr
 * (src/scanpy/preprocessing/_scale.py:280-295) */
int scamd_pp_scale_csr_f32(const int64_t* indptr, const int32_t* indices, float* data, int64_t n, int64_t nnz,
                           const double* std_, double max_value, int has_max, const uint8_t* row_mask,
                           scamd_stream_t stream);
/* zero_center=True: dense out[r*g + c] = clip((x_rc - mean[c]) / std[c], +-max_value) for masked rows, x_rc for the
 * others; out is float64 (sparse input: the reference's `x -= mean` yields a float64 matrix) or float32
 * (out_is_f64 == 0, float32 dense input): scale_array (_scale.py:189-216) + clip_array (:53-69) */
int scamd_pp_scale_dense_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                             int64_t nnz, const double* mean, const double* std_, double max_value, int has_max,
                             const uint8_t* row_mask, void* out, int out_is_f64, scamd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * UMAP layout: the SGD of umap-learn's optimize_layout_euclidean, which
 * simplicial_set_embedding runs for sc.tl.umap (src/scanpy/tools/_umap.py:196-216), in a synchronous, race-free
 * gather formulation (csrc/umap.hip).  Input: CSR of the pruned symmetric fuzzy graph (indptr [n+1], indices [nnz]),
 * epochs_per_sample [nnz] (umap_.make_epochs_per_sample: n_epochs * w / max w inverted; <= 0: never sampled),
 * y [n x dim] float32 row-major = the initial embedding scaled to [0, 10], overwritten with the result.
 * a, b: curve parameters (find_ab_params); gamma, initial_alpha, negative_sample_rate as in the reference call;
 * seed drives the counter-based negative sampling.  Deterministic: same inputs -> bitwise the same embedding.
 * ---------------------------------------------------------------------------------------- */
size_t scamd_umap_workspace_bytes(int64_t n, int64_t nnz, int dim);
int scamd_umap_optimize_f32(const int64_t* indptr, const int32_t* indices, const float* epochs_per_sample, int64_t n,
                            int64_t nnz, int dim, int n_epochs, double a, double b, double gamma,
                            double initial_alpha, double negative_sample_rate, uint64_t seed, float* y,
                            void* workspace, size_t workspace_bytes, scamd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Host-side byte codecs of the on-disk readers (csrc/hostio.cpp).  No device work: callable on a
 * host without a GPU, from many threads at once.  They replace what the reference gets from h5py's bundled filters
 * when it reads `.h5ad` / 10x `.h5` files (src/scanpy/readwrite.py:15-29, 235-243; `compression='lzf'` is one of
 * the two codecs its writer offers, :657-667).
 * scamd_lzf_decompress: liblzf stream -> dst; returns the number of bytes produced (>= 0) or a negative SCAMD_E* code
 *   (malformed stream, or dst_cap too small).
 * scamd_unshuffle: inverse of the HDF5 shuffle filter -- src = elem_size byte planes of n_elem bytes, dst = elements.
 * ---------------------------------------------------------------------------------------- */
int64_t scamd_lzf_decompress(const void* src, size_t src_len, void* dst, size_t dst_cap);
int scamd_unshuffle(const void* src, void* dst, size_t n_elem, int elem_size);

/* ------------------------------------------------------------------------------------------
 * Self tests / micro benchmarks (device).  scamd_selftest_mfma_layout checks the
 * v_mfma_f32_32x32x2_f32 operand/result lane mapping the kNN kernel relies on; returns 0 if OK.
 * ---------------------------------------------------------------------------------------- */
int scamd_selftest_mfma_layout(scamd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCANPY_AMD_H */
