// Dense float64 side of the PCA: top-k eigenpairs of the g x g covariance-like matrix A = X^T X - n mu mu^T.
//
// Replaces, for the Gram route of sc.pp.pca, what sklearn's PCA(svd_solver='arpack') gets from ARPACK
// (sklearn/decomposition/_pca.py:704-793 as called at src/scanpy/preprocessing/_pca/__init__.py:287-308) and what the
// reference's own covariance route gets from `eigh` (src/scanpy/preprocessing/_pca/_dask.py:28-132): round 1 ran this
// part through torch.linalg (rocBLAS GEMMs + ~100 small rocSOLVER kernels, ~10 ms at g = 2000, 6 of them in six
// 128 x 128 `eigh` calls).  Here every piece is a hand-written gfx950 kernel:
//   * tall-skinny GEMMs (g x g x 128, 128 x 128 x g) on the float64 matrix cores, v_mfma_f64_16x16x4_f64: one kernel,
//     C = alpha P^T Q + beta1 D1 + beta2 D2 with P, Q stored k-major, so that both operand fragments are 128-byte
//     coalesced rows (A is symmetric: A Z = A^T Z); 32 x 32 output tile per workgroup, the K range split over its four
//     waves and reduced through LDS in a fixed order (bitwise reproducible);
//   * CholeskyQR of a g x 128 block = Gram matrix (same GEMM) + one-workgroup 128 x 128 Cholesky / triangular inverse in
//     LDS + a panel-times-small product; twice (CholeskyQR2), shifted on a failed pivot;
//   * the 128 x 128 Rayleigh-Ritz eigenproblem by one-sided (Hestenes) Jacobi in ONE workgroup, matrix resident in LDS
//     (133 KB of the 160): 64 disjoint column pairs per step, 16 lanes per pair, round-robin ordering;
//   * Chebyshev-filtered subspace iteration (Zhou & Saad) around them: the filter steps are the GEMM with the
//     three-term recurrence in its epilogue.
// Algorithmic work at g = 2000, b = 128: 2 g^2 b = 1.0e9 flop per operator application (~13 us at the 78.6 TFLOP/s
// float64 matrix peak), ~11-26 applications; everything else is O(g b^2) or O(b^3).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace scamd {

using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int DB_MAX = 128;      // largest block size / Jacobi dimension
constexpr int DB_LD = DB_MAX + 2;  // LDS column stride (doubles): 1040 B, not a multiple of the 256-B bank period
constexpr int DENSE_BATCH_HOST = DB_MAX - 32;  // eigenpairs per batch of the deflated solve (dense_topk_batched)

__device__ __forceinline__ unsigned int dhash32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// Cross-lane float64 through DPP (one VALU move per dword, no LDS crossbar round trip as with ds_bpermute):
// CTRL = DPP control word: 0x120 + n = row_ror:n (rotate within a row of 16 lanes), 0xB1 = quad_perm [1,0,3,2],
// 0x4E = quad_perm [2,3,0,1], 0x141 = row_half_mirror (lane i <-> 7 - i within 8).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ double row16_sum(double x) {
  x += dpp_f64<0x128>(x);
  x += dpp_f64<0x124>(x);
  x += dpp_f64<0x122>(x);
  x += dpp_f64<0x121>(x);
  return x;
}
// sum over aligned groups of 8 lanes, result in every lane
__device__ __forceinline__ double oct_sum(double x) {
  x += dpp_f64<0xB1>(x);
  x += dpp_f64<0x4E>(x);
  x += dpp_f64<0x141>(x);
  return x;
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M x N] = alpha * sum_k P[k][m] Q[k][n] + beta1 * D1[m][n] + beta2 * D2[m][n]     (float64, MFMA 16x16x4)
// P: [K x ldp], Q: [K x ldq] row-major (k-major).  grid (ceil(M/32), ceil(N/32)), 512 threads: wave w owns the
// w-th eighth of K and the whole 32 x 32 tile (2 x 2 MFMA tiles); the eight partial tiles are summed in a fixed order.
// MFMA operand layout (v_mfma_f64_16x16x4_f64): A fragment lane l = A[i = l & 15][k = l >> 4], B fragment lane l =
// B[k = l >> 4][j = l & 15], accumulator register v of lane l = D[i = (l >> 4) + 4 v][j = l & 15] (NOT the
// 4 (l >> 4) + v of the float32 16x16x4 instruction; pinned by tests/test_gpu_dense.py::test_dgemm_tn).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void dgemm_tn_kernel(const double* __restrict__ P, int64_t ldp,
                                                       const double* __restrict__ Q, int64_t ldq, int M, int N, int K,
                                                       double alpha, const double* __restrict__ D1, int64_t ldd1,
                                                       double beta1, const double* __restrict__ D2, int64_t ldd2,
                                                       double beta2, double* __restrict__ C, int64_t ldc) {
  __shared__ double red[8][32 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int kc = ((K + 7) / 8 + 3) / 4 * 4;  // K range of a wave (eight of them), a multiple of 4
  const int kb = wave * kc, ke = min(K, kb + kc);
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[a][b][v] = 0.0;
  const bool mv0 = m0 + l15 < M, mv1 = m0 + 16 + l15 < M;
  const bool nv0 = n0 + l15 < N, nv1 = n0 + 16 + l15 < N;
  const double* pp = P + m0 + l15;
  const double* qq = Q + n0 + l15;
#pragma unroll 4
  for (int k0 = kb; k0 < ke; k0 += 4) {
    const int kk = k0 + kq;
    const bool kv = kk < ke;
    const double a0 = (kv && mv0) ? pp[(int64_t)kk * ldp] : 0.0;
    const double a1 = (kv && mv1) ? pp[(int64_t)kk * ldp + 16] : 0.0;
    const double b0 = (kv && nv0) ? qq[(int64_t)kk * ldq] : 0.0;
    const double b1 = (kv && nv1) ? qq[(int64_t)kk * ldq + 16] : 0.0;
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave][(16 * a + kq + 4 * v) * 32 + 16 * b + l15] = acc[a][b][v];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + 512 * j;
    const int r = idx >> 5, c = idx & 31;
    const int m = m0 + r, n = n0 + c;
    if (m < M && n < N) {
      double s = (((red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx])) +
                  ((red[4][idx] + red[5][idx]) + (red[6][idx] + red[7][idx])));
      s *= alpha;
      if (D1) s += beta1 * D1[(int64_t)m * ldd1 + n];
      if (D2) s += beta2 * D2[(int64_t)m * ldd2 + n];
      C[(int64_t)m * ldc + n] = s;
    }
  }
}

static int dgemm_tn(hipStream_t s, const double* P, int64_t ldp, const double* Q, int64_t ldq, int M, int N, int K,
                    double alpha, const double* D1, int64_t ldd1, double beta1, const double* D2, int64_t ldd2,
                    double beta2, double* C, int64_t ldc) {
  hipLaunchKernelGGL(dgemm_tn_kernel, dim3((M + 31) / 32, (N + 31) / 32), dim3(512), 0, s, P, ldp, Q, ldq, M, N, K,
                     alpha, D1, ldd1, beta1, D2, ldd2, beta2, C, ldc);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Cholesky of the column-normalised Gram matrix + the factor CholeskyQR applies, one workgroup, matrix in LDS.
//   in : G [b x b] (ld b) = Z^T Z, shift >= 0 (relative, added to the unit diagonal)
//   out: S [b x b] row-major with Z_new = Z S orthonormal: S = D^-1 L^-T, D = sqrt(diag G), L L^T = D^-1 G D^-1 + shift I
//        flag: 1 if a pivot was not positive (S is then garbage)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void chol_factor_kernel(const double* __restrict__ G, int b, double shift,
                                                           double* __restrict__ S, int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) double sm[];  // L [b][DB_LD] (row-major, padded) + dinv[b]
  double* L = sm;
  double* dinv = sm + DB_MAX * DB_LD;
  const int tid = threadIdx.x;
  const int row = tid >> 3, part = tid & 7;  // 128 rows x 8 partial sums
  for (int j = tid; j < b; j += 1024) {
    const double d = G[(int64_t)j * b + j];
    dinv[j] = d > 0.0 ? 1.0 / sqrt(d) : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < b * b; e += 1024) {
    const int i = e / b, j = e - i * b;
    double v = G[(int64_t)i * b + j] * dinv[i] * dinv[j];
    if (i == j) v = (dinv[i] > 0.0 ? 1.0 : 0.0) + shift;
    L[i * DB_LD + j] = v;
  }
  __syncthreads();
  // left-looking Cholesky, column j from the finished columns: L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j];
  // the dot products of all rows i >= j run at once, eight lanes each (reduced through DPP)
  bool bad = false;
  __shared__ double sh_piv[2];
  for (int j = 0; j < b; ++j) {
    double acc = 0.0;
    const bool mine = row >= j && row < b;
    if (mine)
      for (int k = part; k < j; k += 8) acc = fma(L[row * DB_LD + k], L[j * DB_LD + k], acc);
    acc = oct_sum(acc);
    const double v = mine ? L[row * DB_LD + j] - acc : 0.0;
    if (row == j && part == 0) sh_piv[j & 1] = v;
    __syncthreads();
    const double piv = sh_piv[j & 1];
    if (!(piv > 0.0)) {  // uniform: every thread reads the same pivot
      bad = true;
      break;
    }
    if (mine && part == 0) L[row * DB_LD + j] = (row == j) ? sqrt(piv) : v / sqrt(piv);
    __syncthreads();  // column j is final before the next column's dot products read it
  }
  if (tid == 0) *flag = bad ? 1 : 0;  // (written either way: no clear in front of the launch)
  if (bad) return;
  // X = L^-1 (lower triangular), row by row: X[j][c] = (delta_jc - sum_{c <= k < j} L[j][k] X[k][c]) / L[j][j] for all
  // columns c <= j at once (thread group `row` = c, eight lanes split k).  X[k][c] (k > c) lives in the unused UPPER
  // triangle at (c, k); X[c][c] = 1 / L[c][c].
  {
    const int c = row;
    const double xcc = c < b ? 1.0 / L[c * DB_LD + c] : 0.0;
    for (int j = 1; j < b; ++j) {
      double acc = 0.0;
      if (c < j) {
        for (int k = c + 1 + part; k < j; k += 8) acc = fma(L[j * DB_LD + k], L[c * DB_LD + k], acc);
        if (part == 0) acc = fma(L[j * DB_LD + c], xcc, acc);
      }
      acc = oct_sum(acc);
      if (c < j && part == 0) L[c * DB_LD + j] = -acc / L[j * DB_LD + j];
      __syncthreads();
    }
  }
  // S[k][n] = X[n][k] * dinv[k] for n >= k, 0 below the diagonal
  for (int e = tid; e < b * b; e += 1024) {
    const int k = e / b, n = e - k * b;
    double v = 0.0;
    if (n == k) v = 1.0 / L[k * DB_LD + k];
    else if (n > k) v = L[k * DB_LD + n];
    S[e] = v * dinv[k];
  }
}

// C[g x b] = Z[g x b] S[b x b] (plain FMA: O(g b^2), 1/16 of a GEMM application); 256 threads = 8 rows x 32 column quads
__global__ __launch_bounds__(256) void panel_small_kernel(const double* __restrict__ Z, const double* __restrict__ S, int g,
                                                          int b, int bo /* columns of S / C */, double* __restrict__ C) {
  __shared__ double zrow[8][DB_MAX];
  const int r = threadIdx.x >> 5, cq = threadIdx.x & 31;
  const int m = blockIdx.x * 8 + r;
  for (int k = cq; k < b; k += 32) zrow[r][k] = (m < g) ? Z[(int64_t)m * b + k] : 0.0;
  __syncthreads();
  if (m >= g) return;
  for (int n0 = cq * 4; n0 < bo; n0 += 128) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
    for (int k = 0; k < b; ++k) {
      const double z = zrow[r][k];
      const double* sp = S + (int64_t)k * bo + n0;
      c0 = fma(z, sp[0], c0);
      if (n0 + 1 < bo) c1 = fma(z, sp[1], c1);
      if (n0 + 2 < bo) c2 = fma(z, sp[2], c2);
      if (n0 + 3 < bo) c3 = fma(z, sp[3], c3);
    }
    double* cp = C + (int64_t)m * bo + n0;
    cp[0] = c0;
    if (n0 + 1 < bo) cp[1] = c1;
    if (n0 + 2 < bo) cp[2] = c2;
    if (n0 + 3 < bo) cp[3] = c3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Eigen-decomposition of a symmetric b x b matrix (b <= 128) by one-sided Jacobi, one workgroup of 512 threads.
// W = T (columns in LDS); rotations orthogonalise the columns: at convergence W = T Y with Y orthogonal, column i =
// theta_i y_i.  64 disjoint pairs per step (round-robin tournament), 8 lanes per pair.  Output: theta[b] descending by
// |theta| (the sign is recovered from y^T T y = sign * |W_i|), Y [b x b] row-major, column j = eigenvector j.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void jacobi_eigh_kernel(const double* __restrict__ T, int b, double* __restrict__ theta,
                                                          double* __restrict__ Y, int* __restrict__ sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];  // W [n2][DB_LD] column-major: W[c * DB_LD + i]
  double* W = sm;
  double* nrm = sm + DB_MAX * DB_LD;  // [DB_MAX]
  __shared__ float off_arr[64];  // per group: largest |cos| between two columns it met in the sweep
  __shared__ int rank_of[DB_MAX];
  const int tid = threadIdx.x;
  const int n2 = (b + 1) & ~1;  // even number of players (a padding column of zeros never rotates)
  for (int e = tid; e < n2 * DB_LD; e += 512) {
    const int c = e / DB_LD, i = e - c * DB_LD;
    W[e] = (c < b && i < b) ? T[(int64_t)i * b + c] : 0.0;
  }
  __syncthreads();
  // 64 groups of 8 lanes, one column pair each.  The kernel is bound by the instruction THROUGHPUT of its one CU (every
  // lane of a group repeats the rotation's scalar arithmetic): eight lanes per pair instead of sixteen halves the
  // redundant work per SIMD, the rotation is computed with two rsqrt and no division, the reductions run on DPP.
  const int grp = tid >> 3, t8 = tid & 7;
  const int nm1 = n2 - 1;
  int sweep = 0;
  for (; sweep < 30; ++sweep) {
    float my_off = 0.f;  // (no shared counter inside the step loop: 64 same-address LDS atomics per step serialise)
    // round-robin tournament: player n2-1 stays, the others rotate; group g > 0 plays (step + g, step - g) mod (n2 - 1)
    int p = grp == 0 ? nm1 : grp % nm1;
    int q = grp == 0 ? 0 : (nm1 - grp % nm1) % nm1;
    for (int step = 0; step < nm1; ++step) {
      const bool act = grp < n2 / 2 && p < b && q < b;
      if (act) {
        double* wp = W + p * DB_LD;
        double* wq = W + q * DB_LD;
        double a = 0.0, bb = 0.0, c = 0.0;
        for (int i = t8; i < b; i += 8) {
          const double x = wp[i], y = wq[i];
          a = fma(x, x, a);
          bb = fma(y, y, bb);
          c = fma(x, y, c);
        }
        a = oct_sum(a);
        bb = oct_sum(bb);
        c = oct_sum(c);
        const double ab = a * bb;
        if (c * c > 1e-30 * ab) {  // |c| / sqrt(a bb) > 1e-15
          // rotation by theta with tan(2 theta) = 2c / (bb - a): cos(2 theta) = |d| / r, r = hypot(d, 2c)
          const double d = bb - a, tau = 2.0 * c;
          const double inv_r = rsqrt(fma(d, d, tau * tau));
          const double h = fma(0.5 * fabs(d), inv_r, 0.5);  // cos^2 of the smaller angle
          const double inv_cs = rsqrt(h);
          const double cs = h * inv_cs;
          const double sn = (d >= 0.0 ? 0.5 : -0.5) * tau * inv_r * inv_cs;
          for (int i = t8; i < b; i += 8) {
            const double x = wp[i], y = wq[i];
            wp[i] = cs * x - sn * y;
            wq[i] = sn * x + cs * y;
          }
          my_off = fmaxf(my_off, (float)(fabs(c) * rsqrt(ab)));
        }
      }
      if (grp != 0) {
        p = p + 1 == nm1 ? 0 : p + 1;
        q = q + 1 == nm1 ? 0 : q + 1;
      } else {
        q = step + 1;
      }
      __syncthreads();
    }
    if (t8 == 0) off_arr[grp] = my_off;
    __syncthreads();
    float off = 0.f;
    for (int i = 0; i < 64; ++i) off = fmaxf(off, off_arr[i]);
    __syncthreads();  // everybody has the sweep's maximum before the next sweep overwrites the array
    // `off` is the largest |cos| the sweep MET, before its rotations: every pair above 1e-15 was rotated in this very sweep, and
    // Jacobi converges quadratically -- a sweep that met nothing above 1e-8 leaves ~1e-16.  (1e-13 until round 6: one more sweep,
    // 0.15 ms of the solve, whose only work was to see that.)
    if (off < 1e-8f) {
      ++sweep;
      break;
    }
  }
  // column norms, ranks (descending norm, ties by index)
  if (tid < n2) {
    double a = 0.0;
    for (int i = 0; i < b; ++i) a = fma(W[tid * DB_LD + i], W[tid * DB_LD + i], a);
    nrm[tid] = tid < b ? sqrt(a) : -1.0;
  }
  __syncthreads();
  if (tid < b) {
    int r = 0;
    const double me = nrm[tid];
    for (int j = 0; j < b; ++j) r += (nrm[j] > me || (nrm[j] == me && j < tid)) ? 1 : 0;
    rank_of[tid] = r;
  }
  __syncthreads();
  // W_c = T y_c = theta_c y_c at convergence: theta_c = |W_c| (T is positive semi-definite: the Rayleigh-Ritz matrix
  // of a covariance), y_c = W_c / |W_c|
  for (int c = grp; c < b; c += 64) {
    const double nv = nrm[c];
    const double inv = nv > 0.0 ? 1.0 / nv : 0.0;
    const int r = rank_of[c];
    if (t8 == 0) theta[r] = nv;
    for (int i = t8; i < b; i += 8) Y[(int64_t)i * b + r] = W[c * DB_LD + i] * inv;
  }
  if (tid == 0 && sweeps_out) *sweeps_out = sweep;
}

// ---------------------------------------------------------------------------------------------------------------------
// small element-wise / reduction kernels
// ---------------------------------------------------------------------------------------------------------------------
// Z[g x b] <- standard normal, counter-based (Box-Muller on two hashes of (seed, element))
__global__ void randn_kernel(double* __restrict__ z, int64_t count, unsigned int seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned int h1 = dhash32((unsigned int)i * 0x9E3779B1u + seed);
  const unsigned int h2 = dhash32(h1 ^ 0x85EBCA77u ^ (unsigned int)(i >> 32));
  const double u1 = ((double)h1 + 0.5) * (1.0 / 4294967296.0), u2 = ((double)h2 + 0.5) * (1.0 / 4294967296.0);
  z[i] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// y = a x1 + b x2 (element-wise)
__global__ void axpby_kernel(int64_t count, double a, const double* __restrict__ x1, double b,
                             const double* __restrict__ x2, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) y[i] = a * x1[i] + b * x2[i];
}

// T <- (T + T^T) / 2 in place (b x b)
__global__ void symmetrize_kernel(double* __restrict__ t, int b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= b * b) return;
  const int i = e / b, j = e - i * b;
  if (j > i) {
    const double v = 0.5 * (t[(int64_t)i * b + j] + t[(int64_t)j * b + i]);
    t[(int64_t)i * b + j] = v;
    t[(int64_t)j * b + i] = v;
  }
}

// Rayleigh quotients t_j = z_j^T (A z_j) of an orthonormal block: out[1] = min_j t_j, out[2] = max_j t_j (one workgroup)
__global__ __launch_bounds__(1024) void rq_minmax_kernel(const double* __restrict__ Z, const double* __restrict__ AZ, int g,
                                                         int b, double* __restrict__ out) {
  __shared__ double part[8][DB_MAX];
  const int c = threadIdx.x & 127, rr = threadIdx.x >> 7;
  double s = 0.0;
  if (c < b) {  // (four independent chains: the loop is a walk of dependent loads otherwise, 76 us at g = 2000)
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int m = rr;
    for (; m + 24 < g; m += 32) {
      s = fma(Z[(int64_t)m * b + c], AZ[(int64_t)m * b + c], s);
      s1 = fma(Z[(int64_t)(m + 8) * b + c], AZ[(int64_t)(m + 8) * b + c], s1);
      s2 = fma(Z[(int64_t)(m + 16) * b + c], AZ[(int64_t)(m + 16) * b + c], s2);
      s3 = fma(Z[(int64_t)(m + 24) * b + c], AZ[(int64_t)(m + 24) * b + c], s3);
    }
    for (; m < g; m += 8) s = fma(Z[(int64_t)m * b + c], AZ[(int64_t)m * b + c], s);
    s = (s + s1) + (s2 + s3);
  }
  part[rr][c] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mn = INFINITY, mx = -INFINITY;
    for (int j = 0; j < b; ++j) {
      double t = 0.0;
      for (int r = 0; r < 8; ++r) t += part[r][j];
      mn = fmin(mn, t);
      mx = fmax(mx, t);
    }
    out[1] = mn;
    out[2] = mx;
  }
}

// out[0] = max_{j < k} |AV_j - theta_j V_j| / max(theta_0, tiny); one workgroup, fixed order
__global__ __launch_bounds__(1024) void residual_kernel(const double* __restrict__ V, const double* __restrict__ AV,
                                                        const double* __restrict__ theta, int g, int b, int k,
                                                        double* __restrict__ out) {
  __shared__ double part[8][DB_MAX];
  const int c = threadIdx.x & 127, rr = threadIdx.x >> 7;
  double s = 0.0;
  if (c < k) {
    const double th = theta[c];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int m = rr;
    for (; m + 24 < g; m += 32) {  // (four independent chains, see rq_minmax_kernel)
      const double d0 = AV[(int64_t)m * b + c] - th * V[(int64_t)m * b + c];
      const double d1 = AV[(int64_t)(m + 8) * b + c] - th * V[(int64_t)(m + 8) * b + c];
      const double d2 = AV[(int64_t)(m + 16) * b + c] - th * V[(int64_t)(m + 16) * b + c];
      const double d3 = AV[(int64_t)(m + 24) * b + c] - th * V[(int64_t)(m + 24) * b + c];
      s = fma(d0, d0, s);
      s1 = fma(d1, d1, s1);
      s2 = fma(d2, d2, s2);
      s3 = fma(d3, d3, s3);
    }
    for (; m < g; m += 8) {
      const double d = AV[(int64_t)m * b + c] - th * V[(int64_t)m * b + c];
      s = fma(d, d, s);
    }
    s = (s + s1) + (s2 + s3);
  }
  part[rr][c] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = 0.0;
    for (int j = 0; j < k; ++j) {
      double t = 0.0;
      for (int r = 0; r < 8; ++r) t += part[r][j];
      mx = fmax(mx, sqrt(t));
    }
    out[0] = mx / fmax(theta[0], 1e-300);
  }
}

// A[g x g] (float64) = gram_q * inv - n mu mu^T with mu = colsum_q * inv / n (the fixed-point Gram matrix of gram.hip);
// also mean[g] and var[g] (population variance of every column, clamped at 0)
__global__ void cov_from_gram_kernel(const long long* __restrict__ gram, int64_t ld_gram, const long long* __restrict__ colsum,
                                     int g, double inv, double n, int zero_center, double* __restrict__ a,
                                     double* __restrict__ mean, double* __restrict__ var) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)g * g) return;
  const int i = (int)(e / g), j = (int)(e - (int64_t)i * g);
  const double mi = (double)colsum[i] * inv / n, mj = (double)colsum[j] * inv / n;
  // gram is exactly symmetric (integer sums); read the (min, max) entry so that A is symmetric to the bit as well
  const int lo = min(i, j), hi = max(i, j);
  const double gv = (double)gram[(int64_t)lo * ld_gram + hi] * inv;
  a[e] = zero_center ? gv - n * mi * mj : gv;
  if (i == j) {
    mean[i] = mi;
    var[i] = fmax(gv / n - mi * mi, 0.0);
  }
}

// sign convention of sklearn's svd_flip(u_based_decision=False): the largest-|.| loading of a component is positive.
// V [g x b] (first k columns used) -> comp64 [k x g] (row = component), v32 [g x k] float32 for the scores SpMM.
// (k = row stride of v32 = components in all; comp64 / v32 are already offset to this batch's first component)
__global__ __launch_bounds__(256) void finalize_components_kernel(const double* __restrict__ V, int g, int b, int k,
                                                                  double* __restrict__ comp64, float* __restrict__ v32) {
  __shared__ double bv[256];
  __shared__ int bi[256];
  const int c = blockIdx.x;
  double best = -1.0;
  int besti = 0;
  for (int m = threadIdx.x; m < g; m += 256) {
    const double a = fabs(V[(int64_t)m * b + c]);
    if (a > best) {
      best = a;
      besti = m;
    }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = besti;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const double ob = bv[threadIdx.x + o];
      const int oi = bi[threadIdx.x + o];
      if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
        bv[threadIdx.x] = ob;
        bi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  const double sg = V[(int64_t)bi[0] * b + c] < 0.0 ? -1.0 : 1.0;
  for (int m = threadIdx.x; m < g; m += 256) {
    const double v = sg * V[(int64_t)m * b + c];
    comp64[(int64_t)c * g + m] = v;
    if (v32) v32[(int64_t)m * k + c] = (float)v;
  }
}

// shift[c] (float32) = sum_m mean[m] * (double)v32[m][c]  (the projection of the column means: X V - 1 shift^T)
__global__ __launch_bounds__(256) void mean_shift_kernel(const double* __restrict__ mean, const float* __restrict__ v32, int g,
                                                         int k, float* __restrict__ shift, double* __restrict__ proj) {
  __shared__ double part[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int m = threadIdx.x; m < g; m += 256) s = fma(mean[m], (double)v32[(int64_t)m * k + c], s);
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += part[i];
    shift[c] = (float)t;
    proj[c] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------------------
struct DenseBuffers {
  double* z[7];  // g x b panels
  double* gm; double* s; double* t; double* y; double* theta; double* resid; int* flags;
};

static void dense_carve(Workspace& ws, int64_t g, int b, DenseBuffers* d) {
  for (int i = 0; i < 7; ++i) d->z[i] = ws.take<double>((size_t)g * b);
  d->gm = ws.take<double>((size_t)b * b);
  d->s = ws.take<double>((size_t)b * b);
  d->t = ws.take<double>((size_t)b * b);
  d->y = ws.take<double>((size_t)b * b);
  d->theta = ws.take<double>((size_t)b + 8);
  d->resid = ws.take<double>(8);
  d->flags = ws.take<int>(8);
}

static int dense_block_size(int64_t g, int k) {
  if (g <= DB_MAX) return (int)g;  // the whole space: one Rayleigh-Ritz is the full decomposition
  // k + 32 vectors, rounded up to 16 (round 2: max(k + 64, 2 k) = 128 at k = 50).  The one-workgroup kernels of an outer
  // iteration (Jacobi b^3, Cholesky b^3) dominate the solve; measured at 1M x 2k, k = 50: b = 128 / 112 / 96 -> fit
  // 15.96 / 14.85 / 13.76 ms on the planted matrix (same 11 GEMMs, residual 2e-11 .. 4e-11, loadings equal to 1e-10)
  // and 40.8 / 37.8 / 34.9 ms on the structure-less one (8 / 9 / 10 outer iterations, each cheaper).
  int b = (k + 32 + 15) / 16 * 16;
  if (const char* e = getenv("SCAMD_DENSE_BLOCK")) {  // experiments: block size of the subspace iteration (>= k + 32)
    const int v = atoi(e);
    if (v >= k + 32 && v % 16 == 0) b = v;
  }
  b = std::min<int>(b, DB_MAX);
  return (int)std::min<int64_t>(b, g);
}

struct DenseCtx {
  hipStream_t s;
  const double* a;
  int64_t lda;
  int g, b;
  DenseBuffers d;
  int n_gemm = 0, n_chol_retry = 0;
  bool shift_first = [] {  // (A/B knob: SCAMD_DENSE_SHIFT_FIRST=0)
    const char* e = getenv("SCAMD_DENSE_SHIFT_FIRST");
    return !(e && e[0] == '0');
  }();
};

static constexpr size_t CHOL_LDS = (size_t)(DB_MAX * DB_LD + DB_MAX) * sizeof(double);
static constexpr size_t JAC_LDS = (size_t)(DB_MAX * DB_LD + DB_MAX) * sizeof(double);

// zout = orthonormal basis of span(zin) by CholeskyQR2: two rounds of (Gram matrix, Cholesky factor, block times factor).
// A failed pivot (the block is a Chebyshev-filtered one: kappa 1e9 and beyond, up to numerical rank deficiency) inserts
// a SHIFTED round (Fukaya et al. 2020: factor G + s I, s ~ 11 (g b + b (b + 1)) u |Y|^2; every such round divides kappa by
// ~1 / sqrt(s) ~ 3e3) and starts the count of plain rounds again; directions that were lost to rounding come back as
// orthonormal noise, as they do from a Householder QR.  zin and tmp are overwritten; zin, tmp, zout are distinct panels.
// filtered = the block comes out of a Chebyshev filter: its plain first round fails (kappa 1e16 and beyond: the solve of the
// bench spent two Cholesky launches and their read-backs on finding that out), so the first round is a shifted one at once
static int cholqr2(DenseCtx& cx, double* zin, double* tmp, double* zout, bool filtered = false, int plain_rounds = 2) {
  const int g = cx.g, b = cx.b;
  double* cur = zin;
  double* other = tmp;
  int plain_ok = 0, shifted_rounds = 0;
  const double s0 = 11.0 * ((double)g * b + (double)b * (b + 1)) * 2.220446049250313e-16 * b;
  while (plain_ok < plain_rounds) {
    int rc = dgemm_tn(cx.s, cur, b, cur, b, b, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, cx.d.gm, b);
    if (rc != SCAMD_OK) return rc;
    double shift = (filtered && plain_ok == 0 && shifted_rounds == 0 && cx.shift_first) ? s0 : 0.0;
    for (int attempt = shift > 0.0 ? 1 : 0;; ++attempt) {
      int bad = 0;
      hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(1024), CHOL_LDS, cx.s, cx.d.gm, b, shift, cx.d.s, cx.d.flags);
      SCAMD_LAUNCH_CHECK();
      SCAMD_READBACK_NOW(&bad, cx.d.flags, sizeof(int), cx.s);
      if (!bad) break;
      SCAMD_REQUIRE(attempt < 4 && shifted_rounds < 8, SCAMD_EUNSUPPORTED,
                    "dense eigensolver: CholeskyQR gave up on the block (%d shifted rounds, attempt %d)", shifted_rounds,
                    attempt);
      shift = shift == 0.0 ? s0 : shift * 1e3;
      ++cx.n_chol_retry;
    }
    const bool was_shifted = shift > 0.0;
    double* dst = (!was_shifted && plain_ok == plain_rounds - 1) ? zout : other;
    hipLaunchKernelGGL(panel_small_kernel, dim3((g + 7) / 8), dim3(256), 0, cx.s, cur, cx.d.s, g, b, b, dst);
    SCAMD_LAUNCH_CHECK();
    if (dst == other) std::swap(cur, other);
    if (was_shifted) {
      plain_ok = 0;
      ++shifted_rounds;
    } else {
      ++plain_ok;
    }
  }
  return SCAMD_OK;
}

// Rayleigh-Ritz on the orthonormal block z: az = A z, T = z^T az, T = Y diag(theta) Y^T, v = z Y, av = az Y
static int rayleigh_ritz(DenseCtx& cx, const double* z, double* az, double* v, double* av, double* h_theta) {
  const int g = cx.g, b = cx.b;
  int rc = dgemm_tn(cx.s, cx.a, cx.lda, z, b, g, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, az, b);
  if (rc != SCAMD_OK) return rc;
  ++cx.n_gemm;
  rc = dgemm_tn(cx.s, z, b, az, b, b, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, cx.d.t, b);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(symmetrize_kernel, dim3((b * b + 255) / 256), dim3(256), 0, cx.s, cx.d.t, b);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(512), JAC_LDS, cx.s, cx.d.t, b, cx.d.theta, cx.d.y,
                     cx.d.flags + 1);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(panel_small_kernel, dim3((g + 7) / 8), dim3(256), 0, cx.s, z, cx.d.y, g, b, b, v);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(panel_small_kernel, dim3((g + 7) / 8), dim3(256), 0, cx.s, az, cx.d.y, g, b, b, av);
  SCAMD_LAUNCH_CHECK();
  SCAMD_READBACK(h_theta, cx.d.theta, sizeof(double) * b, cx.s);  // (handed out by the caller's next synchronisation)
  return SCAMD_OK;
}

// top-k eigenpairs of the symmetric PSD matrix a [g x g]: theta (device, descending) and v = cx.d.z[3] [g x b]
static int dense_topk(DenseCtx& cx, int k, unsigned int seed, double tol, int* n_outer_out, double* resid_out) {
  const int g = cx.g, b = cx.b;
  DenseBuffers& d = cx.d;
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_factor_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_eigh_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)JAC_LDS));
  HostReadbackScope readback_scope;  // (h_theta below is the destination of a fetch handed out one synchronisation later)
  std::vector<double> h_theta(b);
  double h_resid = INFINITY;
  double *z = d.z[0], *tmp = d.z[1], *az = d.z[2], *v = d.z[3], *av = d.z[4], *y0 = d.z[5], *y1 = d.z[6];
  if (b == g) {
    // the block is the whole space: one Rayleigh-Ritz on the identity-like basis = a full eigendecomposition
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)(((int64_t)g * b + 255) / 256)), dim3(256), 0, cx.s, z, (int64_t)g * b,
                       seed);
    SCAMD_LAUNCH_CHECK();
    int rc = cholqr2(cx, z, tmp, y0);
    if (rc != SCAMD_OK) return rc;
    rc = rayleigh_ritz(cx, y0, az, v, av, h_theta.data());
    if (rc != SCAMD_OK) return rc;
    SCAMD_READBACK_SYNC(cx.s);
    *n_outer_out = 0;
    *resid_out = 0.0;
    return SCAMD_OK;
  }
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)(((int64_t)g * b + 255) / 256)), dim3(256), 0, cx.s, z, (int64_t)g * b, seed);
  SCAMD_LAUNCH_CHECK();
  const int64_t cnt = (int64_t)g * b;
  const unsigned egrid = (unsigned)((cnt + 255) / 256);
  // One Chebyshev filter of degree m on the block (vv, avv = A vv), damping [0, c] and scaled to ~1 at `top`; the result
  // ends up in z.  vv / avv are left intact.
  auto filter = [&](const double* vv, const double* avv, double c, double top, int m) -> int {
    const double e = 0.5 * c, center = 0.5 * c;
    double sigma = e / (top - center);
    const double sigma1 = sigma;
    hipLaunchKernelGGL(axpby_kernel, dim3(egrid), dim3(256), 0, cx.s, cnt, sigma1 / e, avv, -center * sigma1 / e, vv, y0);
    SCAMD_LAUNCH_CHECK();
    const double* yprev = vv;
    double* ycur = y0;
    double* ynew = y1;
    for (int it = 2; it <= m; ++it) {
      const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
      const double al = 2.0 * sigma2 / e;
      int rcf = dgemm_tn(cx.s, cx.a, cx.lda, ycur, b, g, b, g, al, ycur, b, -center * al, yprev, b, -(sigma * sigma2), ynew, b);
      if (rcf != SCAMD_OK) return rcf;
      ++cx.n_gemm;
      double* old = (yprev == vv) ? z : const_cast<double*>(yprev);  // vv is never written: z joins the rotation
      yprev = ycur;
      ycur = ynew;
      ynew = old;
      sigma = sigma2;
    }
    if (ycur != z) {  // (a copy by a kernel of ours, not a blit of the runtime)
      hipLaunchKernelGGL(axpby_kernel, dim3(egrid), dim3(256), 0, cx.s, cnt, 1.0, ycur, 0.0, ycur, z);
      SCAMD_LAUNCH_CHECK();
    }
    return SCAMD_OK;
  };
  // two power steps (A (A z): the condition number of the block grows by (lambda_1 / lambda_b)^2, well within
  // CholeskyQR2's reach) and one orthonormalisation
  int rc = dgemm_tn(cx.s, cx.a, cx.lda, z, b, g, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, y1, b);
  if (rc != SCAMD_OK) return rc;
  rc = dgemm_tn(cx.s, cx.a, cx.lda, y1, b, g, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, z, b);
  if (rc != SCAMD_OK) return rc;
  cx.n_gemm += 2;
  // (ONE plain round here: the block has been through two power steps, kappa ~ (lambda_1 / lambda_b)^2, and serves for the
  // Rayleigh quotients that bound the first filter and as that filter's start -- orthonormal to ~kappa^2 u is enough for both;
  // the block is orthonormalised properly after the filter.  SCAMD_DENSE_FIRST_QR_ROUNDS=2: as until round 6)
  static const int first_rounds = [] {
    const char* e = getenv("SCAMD_DENSE_FIRST_QR_ROUNDS");
    return (e && e[0] == '2') ? 2 : 1;
  }();
  rc = cholqr2(cx, z, tmp, v, false, first_rounds);
  if (rc != SCAMD_OK) return rc;
  // First filter straight on this basis, bounds from its Rayleigh quotients: a Rayleigh-Ritz here would only re-mix
  // the block (the filter does not care) at the price of one more 128 x 128 eigenproblem, the most expensive kernel of
  // the solve.  c = smallest quotient (>= lambda_b is not guaranteed, the filter only needs a cut inside the unwanted
  // part), top = largest quotient (a scaling).
  rc = dgemm_tn(cx.s, cx.a, cx.lda, v, b, g, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, av, b);
  if (rc != SCAMD_OK) return rc;
  ++cx.n_gemm;
  hipLaunchKernelGGL(rq_minmax_kernel, dim3(1), dim3(1024), 0, cx.s, v, av, g, b, d.resid);
  SCAMD_LAUNCH_CHECK();
  double h_rq[3] = {0.0, 0.0, 0.0};
  SCAMD_READBACK_NOW(h_rq, d.resid, sizeof(double) * 3, cx.s);
  if (h_rq[2] > h_rq[1] && h_rq[1] > 0.0) {
    // degree of the first filter: 7 (8 until round 6).  On the bench's matrix 8 / 7 / 6 / 5 leave the residual at 4.5e-11 / 4.6e-10 /
    // 4.8e-9 / 5e-8 against the tolerance 2e-8: 8 buys nothing but a block so ill-conditioned that CholeskyQR needs a second
    // shifted round (pca_fit 9.76 / 9.45 / 9.35 ms; 5: a second outer iteration).  SCAMD_DENSE_FIRST_DEGREE: A/B.
    static const int first_degree = [] {
      const char* e = getenv("SCAMD_DENSE_FIRST_DEGREE");
      return e ? std::max(2, std::min(16, atoi(e))) : 7;
    }();
    rc = filter(v, av, h_rq[1], h_rq[2], first_degree);
    if (rc != SCAMD_OK) return rc;
    rc = cholqr2(cx, z, tmp, y0, true);
    if (rc != SCAMD_OK) return rc;
    rc = rayleigh_ritz(cx, y0, az, v, av, h_theta.data());
  } else {
    hipLaunchKernelGGL(axpby_kernel, dim3(egrid), dim3(256), 0, cx.s, cnt, 1.0, v, 0.0, v, y0);
    SCAMD_LAUNCH_CHECK();
    rc = rayleigh_ritz(cx, y0, az, v, av, h_theta.data());
  }
  if (rc != SCAMD_OK) return rc;
  int outer = 0;
  const char* dbg_env = getenv("SCAMD_DENSE_DEBUG");
  const bool dbg = dbg_env && dbg_env[0] == '1';
  constexpr int MAX_OUTER = 60;
  for (outer = 1;; ++outer) {
    hipLaunchKernelGGL(residual_kernel, dim3(1), dim3(1024), 0, cx.s, v, av, d.theta, g, b, k, d.resid);
    SCAMD_LAUNCH_CHECK();
    SCAMD_READBACK(&h_resid, d.resid, sizeof(double), cx.s);
    SCAMD_READBACK_SYNC(cx.s);  // (also completes the copy of theta)
    if (dbg)
      fprintf(stderr, "[dense] outer %d: residual %.3e, theta[0] %.6e theta[k-1] %.6e theta[b-1] %.6e, gemms %d, chol retries %d\n",
              outer, h_resid, h_theta[0], h_theta[k - 1], h_theta[b - 1], cx.n_gemm, cx.n_chol_retry);
    if (h_resid < tol) break;
    SCAMD_REQUIRE(outer < MAX_OUTER, SCAMD_EUNSUPPORTED,
                  "dense eigensolver: residual %.3e above the tolerance %.3e after %d filtered iterations", h_resid, tol,
                  MAX_OUTER);
    const double c = std::max(h_theta[b - 1], 0.0), top = h_theta[0];
    bool filtered = false;
    if (!(top > c && c > 0.0)) {
      // rank-deficient block (c == 0): a plain power step on the Ritz block
      rc = dgemm_tn(cx.s, cx.a, cx.lda, av, b, g, b, g, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, z, b);
      if (rc != SCAMD_OK) return rc;
      ++cx.n_gemm;
    } else {
      // Degree: the filter grows like cosh(m acosh x), x = (lambda - center) / e, so the LARGEST wanted eigenvalue is
      // amplified exp(m (acosh x_1 - acosh x_k)) times more than the smallest wanted one.  Beyond ~1e9 every column
      // is the leading eigenvector plus rounding noise and the k-th pair never converges (seen with k = 40 on a
      // matrix with 29 separated eigenvalues above a bulk: residual stuck at 1e-7, a Cholesky retry every iteration).
      const double e = 0.5 * c, center = 0.5 * c;
      const double x1 = (top - center) / e, xk = std::max((h_theta[k - 1] - center) / e, 1.0);
      const double spread = std::acosh(x1) - std::acosh(xk);
      int m = 16;
      if (spread > 0.0) m = std::max(4, std::min(m, (int)std::floor(20.7 / spread)));
      rc = filter(v, av, c, top, m);
      if (rc != SCAMD_OK) return rc;
      filtered = m >= 8;
    }
    rc = cholqr2(cx, z, tmp, y0, filtered);
    if (rc != SCAMD_OK) return rc;
    rc = rayleigh_ritz(cx, y0, az, v, av, h_theta.data());
    if (rc != SCAMD_OK) return rc;
  }
  *n_outer_out = outer;
  *resid_out = h_resid;
  return SCAMD_OK;
}

}  // namespace scamd

using namespace scamd;

extern "C" size_t scamd_eigh_topk_workspace_bytes(int64_t g, int k) {
  if (g < 1 || k < 1) return 0;
  Workspace ws(nullptr, 0);
  DenseBuffers d;
  dense_carve(ws, g, dense_block_size(g, k), &d);
  return ws.used();
}

extern "C" int scamd_eigh_topk_f64(const double* a, int64_t g, int64_t lda, int k, uint64_t seed, double tol,
                                   double* lam, double* v, int32_t* info_host, void* workspace, size_t workspace_bytes,
                                   scamd_stream_t stream) {
  SCAMD_REQUIRE(a && lam && v, SCAMD_EINVAL, "eigh_topk: null pointer");
  SCAMD_REQUIRE(g >= 1 && lda >= g && k >= 1 && k <= g, SCAMD_EINVAL, "eigh_topk: bad shape g=%lld k=%d", (long long)g, k);
  SCAMD_REQUIRE(g < ((int64_t)1 << 30), SCAMD_EUNSUPPORTED, "eigh_topk: g too large");
  const int b = dense_block_size(g, k);
  SCAMD_REQUIRE(b == g || k + 32 <= b, SCAMD_EUNSUPPORTED,
                "eigh_topk: k=%d needs a block beyond %d columns (use a full eigendecomposition)", k, DB_MAX);
  SCAMD_REQUIRE(b == g || g >= 2 * b, SCAMD_EUNSUPPORTED, "eigh_topk: g=%lld between %d and %d is served by a full eigh",
                (long long)g, DB_MAX, 2 * DB_MAX);
  DenseCtx cx;
  cx.s = stream;
  cx.a = a;
  cx.lda = lda;
  cx.g = (int)g;
  cx.b = b;
  Workspace ws(workspace, workspace_bytes);
  dense_carve(ws, g, b, &cx.d);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "eigh_topk: workspace %zu < required %zu", workspace_bytes, ws.used());
  int n_outer = 0;
  double resid = 0.0;
  int rc = dense_topk(cx, k, (unsigned int)(seed ^ (seed >> 32)) * 0x9E3779B1u + 12345u, tol, &n_outer, &resid);
  if (rc != SCAMD_OK) return rc;
  // lam[k], v [g x k] row-major (column j = eigenvector j), unsigned
  SCAMD_HIP_CHECK(hipMemcpyAsync(lam, cx.d.theta, sizeof(double) * k, hipMemcpyDeviceToDevice, stream));
  SCAMD_HIP_CHECK(hipMemcpy2DAsync(v, sizeof(double) * k, cx.d.z[3], sizeof(double) * b, sizeof(double) * k, (size_t)g,
                                   hipMemcpyDeviceToDevice, stream));
  SCAMD_HIP_CHECK(hipStreamSynchronize(stream));
  if (info_host) {
    info_host[0] = n_outer;
    info_host[1] = cx.n_gemm;
    info_host[2] = b;
    info_host[3] = cx.n_chol_retry;
    double r = resid;
    memcpy(info_host + 4, &r, sizeof(double));  // info_host[4..5] = residual (float64)
  }
  return SCAMD_OK;
}

// debug / test entry: the building blocks on caller buffers.  op 1: C[M x N] = P^T Q (P [K x M], Q [K x N]);
// op 2: S = CholeskyQR factor of G [b x b] (flag_host = pivot failure); op 3: (theta, Y) = eigh(T [b x b]).
extern "C" int scamd_dense_debug_f64(int op, const double* in0, const double* in1, int m, int n, int kdim, double* out0,
                                     double* out1, int32_t* flag_host, scamd_stream_t stream) {
  SCAMD_REQUIRE(in0 && out0, SCAMD_EINVAL, "dense_debug: null pointer");
  if (op == 1) {
    SCAMD_REQUIRE(in1, SCAMD_EINVAL, "dense_debug: null pointer");
    int rc = dgemm_tn(stream, in0, m, in1, n, m, n, kdim, 1.0, nullptr, 0, 0.0, nullptr, 0, 0.0, out0, n);
    if (rc != SCAMD_OK) return rc;
  } else if (op == 2 || op == 3) {
    const int b = m;
    SCAMD_REQUIRE(b >= 1 && b <= DB_MAX, SCAMD_EINVAL, "dense_debug: b=%d", b);
    int* dflag = nullptr;
    SCAMD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dflag), 2 * sizeof(int)));
    SCAMD_HIP_CHECK(hipMemsetAsync(dflag, 0, 2 * sizeof(int), stream));
    if (op == 2) {
      SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_factor_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_LDS));
      hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(1024), CHOL_LDS, stream, in0, b, 0.0, out0, dflag);
    } else {
      SCAMD_REQUIRE(out1, SCAMD_EINVAL, "dense_debug: null pointer");
      SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_eigh_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)JAC_LDS));
      hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(512), JAC_LDS, stream, in0, b, out0, out1, dflag + 1);
    }
    SCAMD_LAUNCH_CHECK();
    int h[2] = {0, 0};
    SCAMD_HIP_CHECK(hipMemcpyAsync(h, dflag, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    SCAMD_HIP_CHECK(hipStreamSynchronize(stream));
    SCAMD_HIP_CHECK(hipFree(dflag));
    if (flag_host) *flag_host = op == 2 ? h[0] : h[1];
    return SCAMD_OK;
  } else {
    SCAMD_REQUIRE(false, SCAMD_EINVAL, "dense_debug: op=%d", op);
  }
  SCAMD_HIP_CHECK(hipStreamSynchronize(stream));
  return SCAMD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// sc.pp.pca on a resident CSR matrix in ONE call (SURVEY.md 8(b).3 `pca_csr_f32`): the exact Gram route of
// scanpy_amd/preprocessing/_pca_solver.py without Python -- max|x| -> fixed-point X^T X (gram.hip) -> A = G - n mu mu^T ->
// top-k eigenpairs (above) -> sign convention of sklearn's svd_flip -> scores X V - 1 (mu^T V) (pca.hip) -> variances.
// Single device; a multi-rank run all-reduces the fixed-point Gram matrix between the two halves and therefore uses
// the building blocks (scamd_csr_gram_f32, scamd_eigh_topk_f64, scamd_spmm_csr_f32) instead.
// ---------------------------------------------------------------------------------------------------------------------
namespace scamd {
struct PcaBuffers {
  long long* gram; long long* colsum; float* v32; float* shift;
  void* gram_ws; size_t gram_ws_bytes; void* solve_ws; size_t solve_ws_bytes;
};
// out[0] = sum of var[0..g) in a fixed order (256 strided partial sums in index order, then a fixed tree: one thread walking
// the array took 105 us of dependent loads at g = 2000)
__global__ __launch_bounds__(256) void sum_kernel(const double* __restrict__ x, int g, double* __restrict__ out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < g; i += 256) s += x[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0];
}
// zero_center: variance[c] = theta[c] / (n - 1) (sklearn PCA: S^2 / (n - 1)); otherwise the variance of the scores of
// the uncentred decomposition, theta[c] / n - (mu^T v_c)^2 (TruncatedSVD); ratio[c] = variance[c] / total
__global__ void variance_kernel(const double* __restrict__ theta, int k, double denom, const double* __restrict__ proj,
                                const double* __restrict__ varsum, double total_scale, double* __restrict__ variance,
                                double* __restrict__ ratio) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  double ev = fmax(theta[c], 0.0) / denom;
  if (proj) ev = fmax(ev - proj[c] * proj[c], 0.0);
  variance[c] = ev;
  const double tot = varsum[0] * total_scale;
  ratio[c] = tot > 0.0 ? ev / tot : 0.0;
}
}  // namespace scamd

// ---- the dense half of the Gram route on its own: (fixed-point Gram matrix, column sums) -> model ------------------------
// What a row-sharded run calls after the all-reduce of the int64 sums, and what scamd_pca_csr_f32 calls on one device: the
// model is therefore bitwise the same for any number of ranks, and no library (torch / rocBLAS) computes any part of it.
namespace scamd {
struct PcaSolveBuffers {
  double* a; double* var; double* varsum; double* proj; double* theta_all; double* deflate; void* dense_ws; size_t dense_ws_bytes;
};
static void pca_solve_carve(Workspace& ws, int64_t g, int k, PcaSolveBuffers* b) {
  b->a = ws.take<double>((size_t)g * g);
  b->var = ws.take<double>((size_t)g);
  b->varsum = ws.take<double>(8);
  b->proj = ws.take<double>((size_t)k + 8);
  b->theta_all = ws.take<double>((size_t)k + 8);
  b->deflate = ws.take<double>(k > DENSE_BATCH_HOST ? (size_t)DENSE_BATCH_HOST * g : 8);
  b->dense_ws_bytes = scamd_eigh_topk_workspace_bytes(g, std::min(k, DENSE_BATCH_HOST));
  b->dense_ws = ws.take<char>(b->dense_ws_bytes);
}
__global__ void copy_theta_kernel(const double* __restrict__ theta, int k, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < k) out[c] = fmax(theta[c], 0.0);
}
// q[c][m] = theta[c] * p[c][m]  (rows = components of a finished batch: A -= P^T Q deflates them)
__global__ void scale_rows_kernel(const double* __restrict__ p, const double* __restrict__ theta, int kb, int g,
                                  double* __restrict__ q) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)kb * g) q[e] = theta[e / g] * p[e];
}
// More components than one block of the subspace iteration holds (k + 32 > DB_MAX): batches of at most DB_MAX - 32, each on
// the matrix DEFLATED by the finished ones, A <- A - V diag(theta) V^T (one rank-kb update on the f64 MFMA): the leading
// eigenpairs of the deflated matrix are the next ones of A, orthogonal to the finished ones to the accuracy they were
// converged to.  Round 6: until then n_comps > 96 ran the same iteration on torch.linalg (rocSOLVER / rocBLAS).
constexpr int DENSE_BATCH = DB_MAX - 32;
static bool dense_in_range(int64_t g, int k) {
  if (g <= DB_MAX) return true;                       // the whole space: one Rayleigh-Ritz
  if (k <= 0 || k > g) return false;
  const int kb = std::min(k, DENSE_BATCH);
  const int bsz = dense_block_size(g, kb);
  return kb + 32 <= bsz && g >= 2 * bsz && (k <= DENSE_BATCH || g >= 2 * k);  // (deflating more than half the space is a full eigh's job)
}
// top-k eigenpairs of cx.a (DESTROYED when k needs more than one batch) -> comp64 [k x g] (sign convention applied),
// v32 [g x k] (may be NULL), theta_all [k] (descending, raw).  `scratch` = kb_max x g doubles for the deflation.
static int dense_topk_batched(DenseCtx& cx, double* a_mut, int k, unsigned int dseed, double tol, double* comp64, float* v32,
                              double* theta_all, double* scratch, void* dense_ws, size_t dense_ws_bytes, int* n_outer, double* resid,
                              int* bsz_out) {
  const int g = cx.g;
  *n_outer = 0;
  *resid = 0.0;
  for (int k0 = 0; k0 < k;) {
    const int kb = g <= DB_MAX ? k : std::min(k - k0, DENSE_BATCH);
    const int bsz = dense_block_size(g, kb);
    cx.b = bsz;
    Workspace dws(dense_ws, dense_ws_bytes);
    dense_carve(dws, g, bsz, &cx.d);
    SCAMD_REQUIRE(dws.ok, SCAMD_EWORKSPACE, "dense eigensolver: workspace");
    int outer = 0;
    double rs = 0.0;
    int rc = dense_topk(cx, kb, dseed + 0x9E3779B9u * (unsigned int)k0, tol, &outer, &rs);
    if (rc != SCAMD_OK) return rc;
    *n_outer += outer;
    *resid = std::max(*resid, rs);
    *bsz_out = bsz;
    hipLaunchKernelGGL(finalize_components_kernel, dim3(kb), dim3(256), 0, cx.s, cx.d.z[3], g, bsz, k, comp64 + (int64_t)k0 * g,
                       v32 ? v32 + k0 : (float*)nullptr);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(axpby_kernel, dim3((kb + 255) / 256), dim3(256), 0, cx.s, (int64_t)kb, 1.0, cx.d.theta, 0.0, cx.d.theta,
                       theta_all + k0);
    SCAMD_LAUNCH_CHECK();
    k0 += kb;
    if (k0 < k) {  // deflate: A -= P^T (diag(theta) P), P = this batch's components [kb x g]
      const double* pb = comp64 + (int64_t)(k0 - kb) * g;
      hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)(((int64_t)kb * g + 255) / 256)), dim3(256), 0, cx.s, pb,
                         theta_all + (k0 - kb), kb, g, scratch);
      SCAMD_LAUNCH_CHECK();
      rc = dgemm_tn(cx.s, pb, g, scratch, g, g, g, kb, -1.0, a_mut, cx.lda, 1.0, nullptr, 0, 0.0, a_mut, cx.lda);
      if (rc != SCAMD_OK) return rc;
      ++cx.n_gemm;
    }
  }
  return SCAMD_OK;
}
// steps 3-5 and 7 of the route; `dseed` is handed to the eigensolver as it is.  No synchronisation: the caller drains.
static int pca_solve_gram(const long long* gram, int64_t ld_gram, const long long* colsum, int64_t n, int64_t g, int scale_bits,
                          int k, int zero_center, unsigned int dseed, double tol, double* components, float* v32, float* shift,
                          double* variance, double* variance_ratio, double* mean, double* theta_out, PcaSolveBuffers& b,
                          hipStream_t s, int* n_outer, double* resid, int* n_gemm, int* bsz_out, int* n_chol_retry) {
  // 3. A = G - n mu mu^T, means, column variances
  const double inv = std::ldexp(1.0, -scale_bits);
  hipLaunchKernelGGL(cov_from_gram_kernel, dim3((unsigned)(((int64_t)g * g + 255) / 256)), dim3(256), 0, s, gram, ld_gram,
                     colsum, (int)g, inv, (double)n, zero_center ? 1 : 0, b.a, mean, b.var);
  SCAMD_LAUNCH_CHECK();
  // 4. top-k eigenpairs (batches of DENSE_BATCH with deflation when k needs more than one block)
  SCAMD_REQUIRE(dense_in_range(g, k), SCAMD_EUNSUPPORTED, "pca: n_comps=%d / g=%lld outside the device eigensolver's range", k,
                (long long)g);
  DenseCtx cx;
  cx.s = s;
  cx.a = b.a;
  cx.lda = g;
  cx.g = (int)g;
  int bsz = 0;
  // 5. (inside the batches) sign convention, float32 loadings; then the projected means
  int rc = dense_topk_batched(cx, b.a, k, dseed, tol, components, v32, b.theta_all, b.deflate, b.dense_ws, b.dense_ws_bytes, n_outer,
                              resid, &bsz);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(mean_shift_kernel, dim3(k), dim3(256), 0, s, mean, v32, (int)g, k, shift, b.proj);
  SCAMD_LAUNCH_CHECK();
  // 7. explained variance (sklearn: S^2 / (n - 1); ratio against the total variance with the same n / (n - 1) factor;
  //    zero_center = False is TruncatedSVD: the variance of the scores of the uncentred decomposition, lam / n - (mu^T v)^2)
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, b.var, (int)g, b.varsum);
  SCAMD_LAUNCH_CHECK();
  const double denom = zero_center ? (double)(n - 1) : (double)n;
  const double total_scale = zero_center ? (double)n / (double)(n - 1) : 1.0;
  hipLaunchKernelGGL(variance_kernel, dim3((k + 255) / 256), dim3(256), 0, s, b.theta_all, k, denom,
                     zero_center ? (const double*)nullptr : (const double*)b.proj, b.varsum, total_scale, variance,
                     variance_ratio);
  SCAMD_LAUNCH_CHECK();
  if (theta_out) {
    hipLaunchKernelGGL(copy_theta_kernel, dim3((k + 255) / 256), dim3(256), 0, s, b.theta_all, k, theta_out);
    SCAMD_LAUNCH_CHECK();
  }
  *n_gemm = cx.n_gemm;
  *bsz_out = bsz;
  *n_chol_retry = cx.n_chol_retry;
  return SCAMD_OK;
}
}  // namespace scamd

extern "C" size_t scamd_pca_solve_gram_workspace_bytes(int64_t g, int n_comps) {
  if (g < 1 || n_comps < 1) return 0;
  Workspace ws(nullptr, 0);
  PcaSolveBuffers b;
  pca_solve_carve(ws, g, n_comps, &b);
  return ws.used();
}

extern "C" int scamd_pca_solve_gram_f64(const int64_t* gram, int64_t ld_gram, const int64_t* colsum, int64_t n_total, int64_t g,
                                        int scale_bits, int n_comps, int zero_center, uint64_t seed, double tol,
                                        double* components, float* loadings_f32, float* shift, double* variance,
                                        double* variance_ratio, double* mean, double* eigenvalues, int32_t* info_host,
                                        void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(gram && colsum && components && loadings_f32 && shift && variance && variance_ratio && mean, SCAMD_EINVAL,
                "pca solve: null pointer");
  const int k = n_comps;
  SCAMD_REQUIRE(n_total >= 2 && g >= 1 && ld_gram >= g && k >= 1 && k <= g && scale_bits >= 0 && scale_bits <= 60, SCAMD_EINVAL,
                "pca solve: bad shape n=%lld g=%lld ld=%lld k=%d S=%d", (long long)n_total, (long long)g, (long long)ld_gram, k, scale_bits);
  SCAMD_REQUIRE(g <= 8192, SCAMD_EUNSUPPORTED, "pca solve: the Gram route takes up to 8192 genes (g=%lld)", (long long)g);
  Workspace ws(workspace, workspace_bytes);
  PcaSolveBuffers b;
  pca_solve_carve(ws, g, k, &b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "pca solve: workspace %zu < required %zu", workspace_bytes, ws.used());
  int n_outer = 0, n_gemm = 0, bsz = 0, n_chol = 0;
  double resid = 0.0;
  int rc = pca_solve_gram(reinterpret_cast<const long long*>(gram), ld_gram, reinterpret_cast<const long long*>(colsum), n_total, g,
                          scale_bits, k, zero_center, (unsigned int)(seed ^ (seed >> 32)) * 0x9E3779B1u + 12345u, tol, components,
                          loadings_f32, shift, variance, variance_ratio, mean, eigenvalues, b, stream, &n_outer, &resid, &n_gemm,
                          &bsz, &n_chol);
  if (rc != SCAMD_OK) return rc;
  SCAMD_HIP_CHECK(hipStreamSynchronize(stream));
  if (info_host) {
    info_host[0] = n_outer;
    info_host[1] = n_gemm;
    info_host[2] = bsz;
    info_host[3] = n_chol;
    memcpy(info_host + 4, &resid, sizeof(double));
    info_host[6] = scale_bits;
    info_host[7] = 0;
  }
  return SCAMD_OK;
}

namespace scamd {
static void pca_carve(Workspace& ws, int64_t n, int64_t g, int k, PcaBuffers* b) {
  const int64_t gp = (g + 127) / 128 * 128;
  b->gram = ws.take<long long>((size_t)gp * gp);
  b->colsum = ws.take<long long>((size_t)gp);
  b->v32 = ws.take<float>((size_t)g * k);
  b->shift = ws.take<float>((size_t)k + 8);
  b->gram_ws_bytes = scamd_csr_gram_workspace_bytes(n, g);
  b->gram_ws = ws.take<char>(b->gram_ws_bytes);
  b->solve_ws_bytes = scamd_pca_solve_gram_workspace_bytes(g, k);
  b->solve_ws = ws.take<char>(b->solve_ws_bytes);
}
}  // namespace scamd

extern "C" size_t scamd_pca_csr_workspace_bytes(int64_t n, int64_t g, int n_comps) {
  if (n < 1 || g < 1 || n_comps < 1) return 0;
  Workspace ws(nullptr, 0);
  PcaBuffers b;
  pca_carve(ws, n, g, n_comps, &b);
  return ws.used();
}

extern "C" int scamd_pca_csr_f32(const int64_t* indptr, const int32_t* indices, const float* data, int64_t n, int64_t g,
                                 int64_t nnz, int n_comps, int zero_center, uint64_t seed, double tol, float* scores,
                                 double* components, double* variance, double* variance_ratio, double* mean,
                                 int32_t* info_host, void* workspace, size_t workspace_bytes, scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && scores && components && variance && variance_ratio && mean, SCAMD_EINVAL, "pca: null pointer");
  SCAMD_REQUIRE(n >= 2 && g >= 1 && nnz >= 0, SCAMD_EINVAL, "pca: bad shape n=%lld g=%lld", (long long)n, (long long)g);
  const int k = n_comps;
  SCAMD_REQUIRE(k >= 1 && k < std::min<int64_t>(n, g), SCAMD_EINVAL,
                "n_components=%d must be strictly less than min(n_samples, n_features)=%lld with svd_solver='arpack'", k,
                (long long)std::min<int64_t>(n, g));
  SCAMD_REQUIRE(g <= 8192, SCAMD_EUNSUPPORTED, "pca: the Gram route takes up to 8192 genes (g=%lld)", (long long)g);
  Workspace ws(workspace, workspace_bytes);
  PcaBuffers b;
  pca_carve(ws, n, g, k, &b);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "pca: workspace %zu < required %zu", workspace_bytes, ws.used());
  hipStream_t s = stream;
  const int64_t gp = (g + 127) / 128 * 128;
  // 1. max |x| -> scale of the fixed-point sums (both sum x_a x_b 2^S and sum x 2^S must stay below 2^62)
  float absmax = 0.f;
  int rc = scamd_csr_gram_f32(indptr, indices, data, n, g, nnz, 0, nullptr, 0, nullptr, &absmax, b.gram_ws, b.gram_ws_bytes, s);
  if (rc != SCAMD_OK) return rc;
  const double am = (double)absmax;
  const double bound = std::max(std::max((double)n * am * am, (double)n * am), 1e-300);
  int scale_bits = std::min((int)std::floor(62.0 - std::log2(bound)), 60);
  SCAMD_REQUIRE(!(am > 0.0) || scale_bits + 2.0 * std::log2(am) >= 24.0, SCAMD_EUNSUPPORTED,
                "pca: fixed-point resolution below float32 for this n and dynamic range (max|x| = %g)", am);
  SCAMD_REQUIRE(scale_bits >= 0, SCAMD_EUNSUPPORTED,
                "pca: n * max|x|^2 = %g exceeds the int64 fixed-point range (2^62): values are not normalised", bound);
  // 2. G = X^T X, column sums (int64, exact)
  rc = scamd_csr_gram_f32(indptr, indices, data, n, g, nnz, scale_bits, reinterpret_cast<int64_t*>(b.gram), gp,
                          reinterpret_cast<int64_t*>(b.colsum), nullptr, b.gram_ws, b.gram_ws_bytes, s);
  if (rc != SCAMD_OK) return rc;
  // 3-5, 7. the dense half (shared with the row-sharded route)
  Workspace sws(b.solve_ws, b.solve_ws_bytes);
  PcaSolveBuffers sb;
  pca_solve_carve(sws, g, k, &sb);
  int n_outer = 0, n_gemm = 0, bsz = 0, n_chol = 0;
  double resid = 0.0;
  rc = pca_solve_gram(b.gram, gp, b.colsum, n, g, scale_bits, k, zero_center, (unsigned int)(seed ^ (seed >> 32)) * 0x9E3779B1u + 12345u,
                      tol, components, b.v32, b.shift, variance, variance_ratio, mean, nullptr, sb, s, &n_outer, &resid, &n_gemm, &bsz,
                      &n_chol);
  if (rc != SCAMD_OK) return rc;
  // 6. scores = X V - 1 shift^T
  rc = scamd_spmm_csr_f32(indptr, indices, data, n, g, b.v32, k, zero_center ? b.shift : nullptr, scores, s);
  if (rc != SCAMD_OK) return rc;
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));
  if (info_host) {
    info_host[0] = n_outer;
    info_host[1] = n_gemm;
    info_host[2] = bsz;
    info_host[3] = n_chol;
    memcpy(info_host + 4, &resid, sizeof(double));
    info_host[6] = scale_bits;
    info_host[7] = 0;
  }
  return SCAMD_OK;
}

// =====================================================================================================================
// Spectral initialisation of the UMAP layout on the device (round 6).  `sc.tl.umap(init_pos='spectral')`
// (src/scanpy/tools/_umap.py:165-215 -> umap-learn `spectral_layout`: the eigenvectors of the normalised Laplacian that
// follow the trivial one, by ARPACK) as ONE call: Chebyshev-filtered subspace iteration on M = (S + I) / 2,
// S = D^-1/2 A D^-1/2, with the known trivial eigenvector sqrt(deg) projected out -- the algorithm of
// scanpy_amd/tools/_umap.py:_top_eigenvectors_below_trivial (which ran on torch.linalg QR / Cholesky / eigh + torch.bmm
// until this round and remains the CPU stand-in of the tests), on the kernels of this file: the block is n x b with
// b = dim + 6 <= 16 columns, S y is the float32 SpMM of pca.hip, every reduction over the n rows is a two-stage sum in a
// fixed order (bitwise reproducible).
// =====================================================================================================================
namespace scamd {
constexpr int SP_MAXB = 16;
constexpr int SP_GRID = 1024;

// deg[v] = sum of the row (float64), one wave per row
__global__ __launch_bounds__(256) void sp_degree_kernel(const int64_t* __restrict__ indptr, const float* __restrict__ w, int64_t n,
                                                        double* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n) return;
  double s = 0.0;
  for (int64_t e = indptr[v] + lane; e < indptr[v + 1]; e += 64) s += (double)w[e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) deg[v] = s;
}
// s[e] = w[e] / sqrt(deg[row] deg[col]) (float32: the SpMM's operand), t0[v] = sqrt(deg[v]) (the trivial eigenvector, unnormalised)
__global__ __launch_bounds__(256) void sp_scale_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                       const float* __restrict__ w, int64_t n, const double* __restrict__ deg,
                                                       float* __restrict__ s, double* __restrict__ t0) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n) return;
  const double dv = deg[v];
  const double dis_v = dv > 0.0 ? 1.0 / sqrt(dv) : 0.0;
  if (lane == 0) t0[v] = sqrt(fmax(dv, 0.0));
  for (int64_t e = indptr[v] + lane; e < indptr[v + 1]; e += 64) {
    const double du = deg[indices[e]];
    const double dis_u = du > 0.0 ? 1.0 / sqrt(du) : 0.0;
    s[e] = (float)((double)w[e] * dis_v * dis_u);
  }
}
// part[blk][i * bq + j] = sum over the block's rows of p[row][i] q[row][j]   (p: n x bp, q: n x bq, both <= 16 columns).
// 256 threads = the 16 x 16 output entries; 64 rows at a time staged through LDS.
__global__ __launch_bounds__(256) void sp_tall_gram_kernel(const double* __restrict__ p, int bp, const double* __restrict__ q,
                                                           int bq, int64_t n, double* __restrict__ part) {
  __shared__ double sp[64][SP_MAXB + 1], sq[64][SP_MAXB + 1];
  const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int64_t rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
  double acc = 0.0;
  for (int64_t base = r0; base < r1; base += 64) {
    const int cnt = (int)(r1 - base < 64 ? r1 - base : 64);
    for (int e = threadIdx.x; e < 64 * SP_MAXB; e += 256) {
      const int r = e >> 4, c = e & 15;
      sp[r][c] = (r < cnt && c < bp) ? p[(base + r) * bp + c] : 0.0;
      sq[r][c] = (r < cnt && c < bq) ? q[(base + r) * bq + c] : 0.0;
    }
    __syncthreads();
    for (int r = 0; r < 64; ++r) acc = fma(sp[r][i], sq[r][j], acc);
    __syncthreads();
  }
  if (i < bp && j < bq) part[(int64_t)blockIdx.x * (bp * bq) + i * bq + j] = acc;
}
// out[e] = sum over the blocks in index order
__global__ void sp_reduce_kernel(const double* __restrict__ part, int nblk, int cnt, double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cnt) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * cnt + e];
  out[e] = s;
}
// y[row][j] -= t0[row] * c[j] / |t0|^2   (c = t0^T y, nrm2[0] = t0^T t0)
__global__ void sp_deflate_kernel(double* __restrict__ y, const double* __restrict__ t0, const double* __restrict__ c,
                                  const double* __restrict__ nrm2, int64_t n, int b) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * b) return;
  const int64_t row = e / b;
  const int j = (int)(e - row * b);
  const double d = nrm2[0];
  if (d > 0.0) y[e] -= t0[row] * (c[j] / d);
}
__global__ void sp_to_f32_kernel(const double* __restrict__ y, int64_t count, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) out[e] = (float)y[e];
}
// out = a * (M y - center y) - bcoef * yprev with M y = (S y + y) / 2; a = 1, center = 0, bcoef = 0: out = M y
__global__ void sp_cheb_kernel(int64_t count, const float* __restrict__ sy, const double* __restrict__ y,
                               const double* __restrict__ yprev, double a, double center, double bcoef, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const double yy = y[e];
  const double my = 0.5 * ((double)sy[e] + yy);
  double r = a * (my - center * yy);
  if (bcoef != 0.0) r -= bcoef * yprev[e];
  out[e] = r;
}
// part[blk][j] = sum over the block's rows of (mv[row][j] - theta[j] v[row][j])^2, j < dim
__global__ __launch_bounds__(256) void sp_resid_kernel(const double* __restrict__ v, const double* __restrict__ mv,
                                                       const double* __restrict__ theta, int64_t n, int b, int dim,
                                                       double* __restrict__ part) {
  __shared__ double red[256];
  const int j = threadIdx.x & 15, rl = threadIdx.x >> 4;  // 16 row lanes x 16 columns
  const int64_t rows_per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
  double acc = 0.0;
  if (j < dim) {
    const double th = theta[j];
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const double d = mv[r * b + j] - th * v[r * b + j];
      acc = fma(d, d, acc);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0) {
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += red[k * 16 + j];
    if (j < dim) part[(int64_t)blockIdx.x * dim + j] = s;
  }
}
__global__ void sp_zero_i32_kernel(int* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}
// out[row][j] = v[row][j], j < dim (row stride b -> dim)
__global__ void sp_take_kernel(const double* __restrict__ v, int64_t n, int b, int dim, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * dim) return;
  const int64_t row = e / dim;
  out[e] = v[row * b + (e - row * dim)];
}

struct SpectralBuffers {
  double* deg; double* t0; float* s; double* pan[7]; float* y32; float* sy32; double* part;
  double* gm; double* smat; double* tmat; double* ymat; double* theta; double* cvec; double* nrm2; double* rnorm; int* flags;
};
static void spectral_carve(Workspace& ws, int64_t n, int64_t nnz, int b, SpectralBuffers* sb) {
  sb->deg = ws.take<double>((size_t)n);
  sb->t0 = ws.take<double>((size_t)n);
  sb->s = ws.take<float>((size_t)std::max<int64_t>(nnz, 1));
  for (int i = 0; i < 7; ++i) sb->pan[i] = ws.take<double>((size_t)n * b);
  sb->y32 = ws.take<float>((size_t)n * b);
  sb->sy32 = ws.take<float>((size_t)n * b);
  sb->part = ws.take<double>((size_t)SP_GRID * SP_MAXB * SP_MAXB);
  sb->gm = ws.take<double>(SP_MAXB * SP_MAXB);
  sb->smat = ws.take<double>(SP_MAXB * SP_MAXB);
  sb->tmat = ws.take<double>(SP_MAXB * SP_MAXB);
  sb->ymat = ws.take<double>(SP_MAXB * SP_MAXB);
  sb->theta = ws.take<double>(SP_MAXB);
  sb->cvec = ws.take<double>(SP_MAXB);
  sb->nrm2 = ws.take<double>(8);
  sb->rnorm = ws.take<double>(SP_MAXB);
  sb->flags = ws.take<int>(8);
}

struct SpectralCtx {
  hipStream_t s;
  const int64_t* indptr;
  const int32_t* indices;
  int64_t n;
  int b, dim;
  SpectralBuffers sb;
  int n_apply = 0;
  int grid_rows() const { return (int)std::min<int64_t>(SP_GRID, (n + 63) / 64); }
  unsigned egrid() const { return (unsigned)((n * b + 255) / 256); }
};
// out[bp x bq] = p^T q over the n rows
static int sp_gram(SpectralCtx& cx, const double* p, int bp, const double* q, int bq, double* out) {
  const int g = cx.grid_rows();
  hipLaunchKernelGGL(sp_tall_gram_kernel, dim3(g), dim3(256), 0, cx.s, p, bp, q, bq, cx.n, cx.sb.part);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sp_reduce_kernel, dim3(1), dim3(256), 0, cx.s, cx.sb.part, g, bp * bq, out);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}
// y <- y - t0 (t0^T y) / |t0|^2
static int sp_deflate(SpectralCtx& cx, double* y) {
  int rc = sp_gram(cx, cx.sb.t0, 1, y, cx.b, cx.sb.cvec);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(sp_deflate_kernel, dim3(cx.egrid()), dim3(256), 0, cx.s, y, cx.sb.t0, cx.sb.cvec, cx.sb.nrm2, cx.n, cx.b);
  SCAMD_LAUNCH_CHECK();
  return SCAMD_OK;
}
// out = a (M y - center y) - bcoef yprev
static int sp_apply(SpectralCtx& cx, const double* y, const double* yprev, double a, double center, double bcoef, double* out) {
  const int64_t cnt = cx.n * cx.b;
  hipLaunchKernelGGL(sp_to_f32_kernel, dim3(cx.egrid()), dim3(256), 0, cx.s, y, cnt, cx.sb.y32);
  SCAMD_LAUNCH_CHECK();
  int rc = scamd_spmm_csr_f32(cx.indptr, cx.indices, cx.sb.s, cx.n, cx.n, cx.sb.y32, cx.b, nullptr, cx.sb.sy32, cx.s);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(sp_cheb_kernel, dim3(cx.egrid()), dim3(256), 0, cx.s, cnt, (const float*)cx.sb.sy32, y, yprev, a, center, bcoef, out);
  SCAMD_LAUNCH_CHECK();
  ++cx.n_apply;
  return SCAMD_OK;
}
// CholeskyQR2 of the n x b block `cur` (overwritten) with scratch `other`; the orthonormal block ends in *result (one of the two)
static int sp_cholqr2(SpectralCtx& cx, double* cur, double* other, double** result) {
  const int b = cx.b;
  int plain_ok = 0, shifted_rounds = 0;
  const double s0 = 11.0 * ((double)cx.n * b + (double)b * (b + 1)) * 2.220446049250313e-16 * b;
  while (plain_ok < 2) {
    int rc = sp_gram(cx, cur, b, cur, b, cx.sb.gm);
    if (rc != SCAMD_OK) return rc;
    double shift = 0.0;
    for (int attempt = 0;; ++attempt) {
      int bad = 0;
      hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(1024), CHOL_LDS, cx.s, cx.sb.gm, b, shift, cx.sb.smat, cx.sb.flags);
      SCAMD_LAUNCH_CHECK();
      SCAMD_READBACK_NOW(&bad, cx.sb.flags, sizeof(int), cx.s);
      if (!bad) break;
      SCAMD_REQUIRE(attempt < 4 && shifted_rounds < 8, SCAMD_EUNSUPPORTED,
                    "spectral init: CholeskyQR gave up on the block (%d shifted rounds, attempt %d)", shifted_rounds, attempt);
      shift = attempt == 0 ? s0 : shift * 1e3;
      hipLaunchKernelGGL(sp_zero_i32_kernel, dim3(1), dim3(64), 0, cx.s, cx.sb.flags, 1);  // (chol_factor_kernel only RAISES the flag)
      SCAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(panel_small_kernel, dim3((unsigned)((cx.n + 7) / 8)), dim3(256), 0, cx.s, cur, cx.sb.smat, (int)cx.n, b, b, other);
    SCAMD_LAUNCH_CHECK();
    std::swap(cur, other);
    if (shift > 0.0) {
      plain_ok = 0;
      ++shifted_rounds;
    } else {
      ++plain_ok;
    }
  }
  *result = cur;
  return SCAMD_OK;
}
// Rayleigh-Ritz on the orthonormal block z: mz = M z, T = z^T mz = Y diag(theta) Y^T, v = z Y, mv = mz Y; theta -> host
static int sp_rayleigh_ritz(SpectralCtx& cx, const double* z, double* mz, double* v, double* mv, double* h_theta) {
  const int b = cx.b;
  int rc = sp_apply(cx, z, nullptr, 1.0, 0.0, 0.0, mz);
  if (rc != SCAMD_OK) return rc;
  rc = sp_gram(cx, z, b, mz, b, cx.sb.tmat);
  if (rc != SCAMD_OK) return rc;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(1), dim3(256), 0, cx.s, cx.sb.tmat, b);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(512), JAC_LDS, cx.s, cx.sb.tmat, b, cx.sb.theta, cx.sb.ymat, cx.sb.flags + 1);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(panel_small_kernel, dim3((unsigned)((cx.n + 7) / 8)), dim3(256), 0, cx.s, z, cx.sb.ymat, (int)cx.n, b, b, v);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(panel_small_kernel, dim3((unsigned)((cx.n + 7) / 8)), dim3(256), 0, cx.s, mz, cx.sb.ymat, (int)cx.n, b, b, mv);
  SCAMD_LAUNCH_CHECK();
  SCAMD_READBACK_NOW(h_theta, cx.sb.theta, sizeof(double) * b, cx.s);
  return SCAMD_OK;
}
}  // namespace scamd

extern "C" size_t scamd_spectral_embedding_workspace_bytes(int64_t n, int64_t nnz, int dim) {
  if (n < 1 || nnz < 0 || dim < 1 || dim + 6 > SP_MAXB) return 0;
  Workspace ws(nullptr, 0);
  SpectralBuffers sb;
  spectral_carve(ws, n, nnz, dim + 6, &sb);
  return ws.used();
}

extern "C" int scamd_spectral_embedding_f32(const int64_t* indptr, const int32_t* indices, const float* weights, int64_t n,
                                            int64_t nnz, int dim, uint64_t seed, double tol, int max_outer, int max_degree,
                                            double* out, double* info_host, void* workspace, size_t workspace_bytes,
                                            scamd_stream_t stream) {
  SCAMD_REQUIRE(indptr && indices && weights && out, SCAMD_EINVAL, "spectral init: null pointer");
  SCAMD_REQUIRE(dim >= 1 && dim + 6 <= SP_MAXB, SCAMD_EUNSUPPORTED, "spectral init: %d components (at most %d)", dim, SP_MAXB - 6);
  SCAMD_REQUIRE(n > dim + 6 && n < ((int64_t)1 << 31) && nnz >= 1, SCAMD_EINVAL, "spectral init: bad shape n=%lld nnz=%lld",
                (long long)n, (long long)nnz);
  SpectralCtx cx;
  cx.s = stream;
  cx.indptr = indptr;
  cx.indices = indices;
  cx.n = n;
  cx.dim = dim;
  cx.b = dim + 6;
  const int b = cx.b;
  Workspace ws(workspace, workspace_bytes);
  spectral_carve(ws, n, nnz, b, &cx.sb);
  SCAMD_REQUIRE(workspace && ws.ok, SCAMD_EWORKSPACE, "spectral init: workspace %zu < required %zu", workspace_bytes, ws.used());
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_factor_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_LDS));
  SCAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_eigh_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)JAC_LDS));
  SpectralBuffers& sb = cx.sb;
  hipStream_t s = stream;
  // the flag words, then the operator: degrees, S = D^-1/2 A D^-1/2 in float32, the trivial eigenvector sqrt(deg)
  hipLaunchKernelGGL(sp_zero_i32_kernel, dim3(1), dim3(64), 0, s, sb.flags, 8);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sp_degree_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, indptr, weights, n, sb.deg);
  SCAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sp_scale_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, indptr, indices, weights, n,
                     (const double*)sb.deg, sb.s, sb.t0);
  SCAMD_LAUNCH_CHECK();
  int rc = sp_gram(cx, sb.t0, 1, sb.t0, 1, sb.nrm2);
  if (rc != SCAMD_OK) return rc;
  double *z = sb.pan[0], *tmp = sb.pan[1], *mz = sb.pan[2], *v = sb.pan[3], *mv = sb.pan[4], *y0 = sb.pan[5], *y1 = sb.pan[6];
  const int64_t cnt = n * b;
  hipLaunchKernelGGL(randn_kernel, dim3(cx.egrid()), dim3(256), 0, s, z, cnt,
                     (unsigned int)(seed ^ (seed >> 32)) * 0x9E3779B1u + 0x5bd1e995u);
  SCAMD_LAUNCH_CHECK();
  rc = sp_deflate(cx, z);
  if (rc != SCAMD_OK) return rc;
  double* zq = nullptr;
  rc = sp_cholqr2(cx, z, tmp, &zq);
  if (rc != SCAMD_OK) return rc;
  double h_theta[SP_MAXB];
  rc = sp_rayleigh_ritz(cx, zq, mz, v, mv, h_theta);
  if (rc != SCAMD_OK) return rc;
  double resid = INFINITY;
  int outer = 0;
  for (outer = 1; outer <= max_outer; ++outer) {
    // residual of the wanted Ritz pairs (|M| = 1: absolute = relative)
    const int g = cx.grid_rows();
    hipLaunchKernelGGL(sp_resid_kernel, dim3(g), dim3(256), 0, s, (const double*)v, (const double*)mv, (const double*)sb.theta, n, b, dim, sb.part);
    SCAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sp_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)sb.part, g, dim, sb.rnorm);
    SCAMD_LAUNCH_CHECK();
    double h_r[SP_MAXB];
    SCAMD_READBACK_NOW(h_r, sb.rnorm, sizeof(double) * dim, s);
    resid = 0.0;
    for (int j = 0; j < dim; ++j) resid = std::max(resid, std::sqrt(std::max(h_r[j], 0.0)));
    if (resid < tol) break;
    const double c = h_theta[b - 1];
    double* blk = nullptr;  // the block that is deflated and orthonormalised next
    if (!(c > 0.0 && c < 1.0)) {  // a degenerate block: one plain step keeps it simple
      blk = mv;
    } else {
      const double e = 0.5 * c, center = 0.5 * c;
      // the degree: as high as the amplification SPREAD inside the wanted set allows (beyond ~1e9 every column is the
      // leading wanted vector plus rounding noise); the spectrum's upper end is 1
      const double x1 = (1.0 - center) / e, xk = std::max((h_theta[dim - 1] - center) / e, 1.0);
      const double spread = std::acosh(x1) - std::acosh(xk);
      const int m = spread <= 0.0 ? max_degree : std::max(4, std::min(max_degree, (int)std::floor(20.7 / spread)));
      double sigma = e / (1.0 - center);
      const double sigma1 = sigma;
      // y = (mv - center v) sigma1 / e = (sigma1 / e) mv - (center sigma1 / e) v
      hipLaunchKernelGGL(axpby_kernel, dim3(cx.egrid()), dim3(256), 0, s, cnt, sigma1 / e, (const double*)mv, -center * sigma1 / e,
                         (const double*)v, y0);
      SCAMD_LAUNCH_CHECK();
      const double* yprev = v;
      double* ycur = y0;
      double* ynew = y1;
      for (int it = 2; it <= m; ++it) {
        const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
        rc = sp_apply(cx, ycur, yprev, 2.0 * sigma2 / e, center, sigma * sigma2, ynew);
        if (rc != SCAMD_OK) return rc;
        double* old = (yprev == v) ? z : const_cast<double*>(yprev);  // v is never written: z joins the rotation
        yprev = ycur;
        ycur = ynew;
        ynew = old;
        sigma = sigma2;
      }
      blk = ycur;
    }
    rc = sp_deflate(cx, blk);
    if (rc != SCAMD_OK) return rc;
    // scratch for the orthonormalisation: any panel that is neither the block nor v / mv / mz (those are rewritten below)
    double* scratch = (blk == tmp) ? z : tmp;
    if (scratch == blk) scratch = y1;
    if (blk == mv) {  // (the plain step orthonormalises a COPY: mv is an output of the Rayleigh-Ritz that follows)
      hipLaunchKernelGGL(axpby_kernel, dim3(cx.egrid()), dim3(256), 0, s, cnt, 1.0, (const double*)mv, 0.0, (const double*)mv, y0);
      SCAMD_LAUNCH_CHECK();
      blk = y0;
      scratch = y1;
    }
    rc = sp_cholqr2(cx, blk, scratch, &zq);
    if (rc != SCAMD_OK) return rc;
    rc = sp_rayleigh_ritz(cx, zq, mz, v, mv, h_theta);
    if (rc != SCAMD_OK) return rc;
  }
  hipLaunchKernelGGL(sp_take_kernel, dim3((unsigned)((n * dim + 255) / 256)), dim3(256), 0, s, (const double*)v, n, b, dim, out);
  SCAMD_LAUNCH_CHECK();
  SCAMD_HIP_CHECK(hipStreamSynchronize(s));
  if (info_host) {
    info_host[0] = (double)std::min(outer, max_outer);
    info_host[1] = (double)cx.n_apply;
    info_host[2] = resid;
    info_host[3] = resid < tol ? 1.0 : 0.0;
    for (int j = 0; j < dim && j < 4; ++j) info_host[4 + j] = 2.0 * h_theta[j] - 1.0;  // eigenvalues of S
  }
  return SCAMD_OK;
}
