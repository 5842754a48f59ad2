/*
 * oracle/leiden.c -- CPU restatement of the Leiden algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference delegates to third-party code that is NOT in /root/reference:
 *   leidenalg >= 0.10.1  find_partition(g, RBConfigurationVertexPartition, weights, n_iterations,
 *                        resolution_parameter, seed)          (src/scanpy/tools/_leiden.py:167-187)
 *   igraph   >= 0.10.8   Graph.community_leiden(objective_function='modularity', weights,
 *                        resolution, n_iterations)            (src/scanpy/tools/_leiden.py:188-196)
 * Both implement Traag, Waltman & van Eck, "From Louvain to Leiden" (Sci. Rep. 2019), whose
 * published algorithm is restated here:
 *   repeat { fast local moving (queue, random order) -> refinement inside each community
 *            (singletons merge into well-connected sub-communities, chosen ~ exp(gain/beta)) ->
 *            aggregation on the refined partition, initial partition = non-refined one }
 *   until nothing aggregates; whole thing repeated n_iterations times (or until stable if < 0).
 * Quality (both flavors on a symmetric matrix):
 *   Q = 1/(2m) sum_ij (A_ij - gamma k_i k_j / (2m)) delta(c_i, c_j)
 * PARITY UNPINNED at label level: the reference ships no golden labels
 * (tests/test_clustering.py:67-163 pin determinism, seed sensitivity and NMI > 0.9 only).
 * Pinned instead to what the paper PROVES of any correct implementation (section "Guarantees"): after a stable
 * iteration no single vertex move and no merge of two communities improves the quality, every community is connected --
 * oracle/leiden_guarantees.py, tests/test_leiden_guarantees_cpu.py (0 violations on every graph tried).
 * Output ids are consecutive and ordered by decreasing community size (leidenalg convention).
 *
 * Input: symmetric CSR (both (i,j) and (j,i) stored) as src/scanpy/_utils/__init__.py:278-304
 * feeds it; float64 weights (src/scanpy/tools/_leiden.py:176-177).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t n;
  int64_t* indptr;
  int32_t* indices;
  double* w;
  double* k; /* node strength = row sum (self loops counted once, they hold 2x internal weight) */
  int owns;
} graph_t;

static uint64_t rng_state;
static uint64_t rng_next(void) { /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double rng_uniform(void) { return (rng_next() >> 11) * (1.0 / 9007199254740992.0); }
static void shuffle(int32_t* a, int64_t n) {
  for (int64_t i = n - 1; i > 0; --i) {
    int64_t j = (int64_t)(rng_next() % (uint64_t)(i + 1));
    int32_t t = a[i];
    a[i] = a[j];
    a[j] = t;
  }
}

static void graph_free(graph_t* g) {
  if (g->owns) {
    free(g->indptr);
    free(g->indices);
    free(g->w);
  }
  free(g->k);
}

static void compute_strength(graph_t* g) {
  g->k = (double*)malloc(sizeof(double) * (size_t)(g->n > 0 ? g->n : 1));
  for (int64_t v = 0; v < g->n; ++v) {
    double s = 0;
    for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) s += g->w[e];
    g->k[v] = s;
  }
}

/* ---- fast local moving ------------------------------------------------------------------------ */
static int move_nodes(const graph_t* g, int32_t* comm, double gamma, double m2) {
  const int64_t n = g->n;
  double* K = (double*)calloc((size_t)n, sizeof(double));
  int32_t* csize = (int32_t*)calloc((size_t)n, sizeof(int32_t));
  double* wc = (double*)calloc((size_t)n, sizeof(double));
  int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  char* inq = (char*)malloc((size_t)n);
  int32_t* empties = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int64_t n_empty = 0;
  for (int64_t v = 0; v < n; ++v) {
    K[comm[v]] += g->k[v];
    csize[comm[v]]++;
    queue[v] = (int32_t)v;
    inq[v] = 1;
  }
  for (int64_t c = 0; c < n; ++c)
    if (csize[c] == 0) empties[n_empty++] = (int32_t)c;
  shuffle(queue, n);
  int64_t head = 0, count = n;
  int moved_any = 0;
  while (count > 0) {
    const int32_t v = queue[head];
    head = (head + 1) % n;
    --count;
    inq[v] = 0;
    const int32_t a = comm[v];
    const double kv = g->k[v];
    int64_t nt = 0;
    for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
      const int32_t u = g->indices[e];
      if (u == v) continue;
      const int32_t c = comm[u];
      if (wc[c] == 0.0) touched[nt++] = c;
      wc[c] += g->w[e];
    }
    K[a] -= kv;
    csize[a]--;
    if (csize[a] == 0) empties[n_empty++] = a;
    int32_t best = a;
    double best_val = wc[a] - gamma * kv * K[a] / m2;
    for (int64_t t = 0; t < nt; ++t) {
      const int32_t c = touched[t];
      const double val = wc[c] - gamma * kv * K[c] / m2;
      if (val > best_val) {
        best_val = val;
        best = c;
      }
    }
    if (best_val < 0.0 && n_empty > 0) best = empties[n_empty - 1]; /* an empty community is better */
    for (int64_t t = 0; t < nt; ++t) wc[touched[t]] = 0.0;
    wc[a] = 0.0;
    if (csize[best] == 0) { /* taking an empty slot (possibly the one just vacated) */
      for (int64_t i = n_empty - 1; i >= 0; --i)
        if (empties[i] == best) {
          empties[i] = empties[--n_empty];
          break;
        }
    }
    K[best] += kv;
    csize[best]++;
    comm[v] = best;
    if (best != a) {
      moved_any = 1;
      for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
        const int32_t u = g->indices[e];
        if (u != v && comm[u] != best && !inq[u]) {
          queue[(head + count) % n] = u;
          ++count;
          inq[u] = 1;
        }
      }
    }
  }
  free(K);
  free(csize);
  free(wc);
  free(touched);
  free(queue);
  free(inq);
  free(empties);
  return moved_any;
}

/* ---- refinement ---------------------------------------------------------------------------------- */
static void refine(const graph_t* g, const int32_t* comm, int32_t* ref, double gamma, double beta, double m2) {
  const int64_t n = g->n;
  double* KC = (double*)calloc((size_t)n, sizeof(double));   /* strength of the constraining community */
  double* Kr = (double*)malloc(sizeof(double) * (size_t)n);  /* strength of refined community */
  double* Er = (double*)calloc((size_t)n, sizeof(double));   /* w(r, C - r) */
  int32_t* rsize = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  double* wr = (double*)calloc((size_t)n, sizeof(double));
  int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  double* cand_val = (double*)malloc(sizeof(double) * (size_t)n);
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int64_t v = 0; v < n; ++v) {
    ref[v] = (int32_t)v;
    rsize[v] = 1;
    Kr[v] = g->k[v];
    KC[comm[v]] += g->k[v];
    order[v] = (int32_t)v;
  }
  for (int64_t v = 0; v < n; ++v)
    for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
      const int32_t u = g->indices[e];
      if (u != v && comm[u] == comm[v]) Er[v] += g->w[e];
    }
  shuffle(order, n);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t v = order[i];
    if (rsize[ref[v]] != 1) continue; /* only singletons may merge */
    const double kv = g->k[v];
    const double KCv = KC[comm[v]];
    if (Er[ref[v]] < gamma * kv * (KCv - kv) / m2) continue; /* v not well connected inside C */
    int64_t nt = 0;
    for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
      const int32_t u = g->indices[e];
      if (u == v || comm[u] != comm[v]) continue;
      const int32_t r = ref[u];
      if (wr[r] == 0.0) touched[nt++] = r;
      wr[r] += g->w[e];
    }
    /* candidates: well-connected refined communities with non-negative gain; staying has gain 0 */
    int64_t nc = 0;
    double vmax = 0.0;
    int32_t self_r = ref[v];
    for (int64_t t = 0; t < nt; ++t) {
      const int32_t r = touched[t];
      if (r == self_r) continue;
      if (Er[r] < gamma * Kr[r] * (KCv - Kr[r]) / m2) continue;
      const double gain = wr[r] - gamma * kv * Kr[r] / m2;
      if (gain < 0.0) continue;
      touched[nc] = r; /* compact in place (nc <= t) */
      cand_val[nc] = gain;
      if (gain > vmax) vmax = gain;
      ++nc;
    }
    int32_t target = -1;
    if (nc > 0) {
      /* Pr(r) ~ exp(gain / beta); staying (gain 0) is a candidate too */
      double tot = exp((0.0 - vmax) / beta);
      for (int64_t t = 0; t < nc; ++t) tot += exp((cand_val[t] - vmax) / beta);
      double x = rng_uniform() * tot;
      x -= exp((0.0 - vmax) / beta);
      if (x >= 0.0) {
        target = touched[nc - 1];
        for (int64_t t = 0; t < nc; ++t) {
          x -= exp((cand_val[t] - vmax) / beta);
          if (x < 0.0) {
            target = touched[t];
            break;
          }
        }
      }
    }
    /* reset scratch: every refined community adjacent to v inside C */
    for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
      const int32_t u = g->indices[e];
      if (u != v && comm[u] == comm[v]) wr[ref[u]] = 0.0;
    }
    if (target >= 0) {
      /* w(v, target) recomputed (scratch already cleared) */
      double wvt = 0.0;
      for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
        const int32_t u = g->indices[e];
        if (u != v && comm[u] == comm[v] && ref[u] == target) wvt += g->w[e];
      }
      Er[target] = Er[target] + Er[self_r] - 2.0 * wvt;
      Kr[target] += kv;
      rsize[target] += 1;
      rsize[self_r] = 0;
      Kr[self_r] = 0.0;
      Er[self_r] = 0.0;
      ref[v] = target;
    }
  }
  free(KC);
  free(Kr);
  free(Er);
  free(rsize);
  free(wr);
  free(touched);
  free(cand_val);
  free(order);
}

/* ---- aggregation ------------------------------------------------------------------------------- */
/* relabel `lab` (n values in [0,range)) to consecutive ids in order of first appearance; returns count */
static int64_t relabel(int32_t* lab, int64_t n, int64_t range) {
  int32_t* map = (int32_t*)malloc(sizeof(int32_t) * (size_t)(range > 0 ? range : 1));
  for (int64_t i = 0; i < range; ++i) map[i] = -1;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (map[lab[i]] < 0) map[lab[i]] = (int32_t)c++;
    lab[i] = map[lab[i]];
  }
  free(map);
  return c;
}

static graph_t aggregate(const graph_t* g, const int32_t* ref, int64_t nc) {
  /* ref already relabelled to [0, nc) */
  const int64_t n = g->n;
  graph_t a;
  a.n = nc;
  a.owns = 1;
  int64_t* start = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
  int32_t* members = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t v = 0; v < n; ++v) start[ref[v] + 1]++;
  for (int64_t c = 0; c < nc; ++c) start[c + 1] += start[c];
  int64_t* pos = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nc > 0 ? nc : 1));
  memcpy(pos, start, sizeof(int64_t) * (size_t)nc);
  for (int64_t v = 0; v < n; ++v) members[pos[ref[v]]++] = (int32_t)v;
  double* acc = (double*)calloc((size_t)nc, sizeof(double));
  char* seen = (char*)calloc((size_t)nc, 1);
  int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
  int64_t cap = g->indptr[n] > 16 ? g->indptr[n] : 16;
  a.indptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nc + 1));
  a.indices = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  a.w = (double*)malloc(sizeof(double) * (size_t)cap);
  int64_t nnz = 0;
  a.indptr[0] = 0;
  for (int64_t c = 0; c < nc; ++c) {
    int64_t nt = 0;
    for (int64_t i = start[c]; i < start[c + 1]; ++i) {
      const int32_t v = members[i];
      for (int64_t e = g->indptr[v]; e < g->indptr[v + 1]; ++e) {
        const int32_t d = ref[g->indices[e]];
        if (!seen[d]) {
          seen[d] = 1;
          touched[nt++] = d;
        }
        acc[d] += g->w[e];
      }
    }
    for (int64_t t = 0; t < nt; ++t) {
      const int32_t d = touched[t];
      a.indices[nnz] = d;
      a.w[nnz] = acc[d];
      ++nnz;
      acc[d] = 0.0;
      seen[d] = 0;
    }
    a.indptr[c + 1] = nnz;
  }
  free(start);
  free(members);
  free(pos);
  free(acc);
  free(seen);
  free(touched);
  compute_strength(&a);
  return a;
}

/* ---- quality ----------------------------------------------------------------------------------- */
double oracle_modularity(int64_t n, const int64_t* indptr, const int32_t* indices, const double* w,
                         const int32_t* membership, double gamma) {
  double m2 = 0.0, in = 0.0;
  int32_t maxc = 0;
  for (int64_t v = 0; v < n; ++v)
    if (membership[v] > maxc) maxc = membership[v];
  double* K = (double*)calloc((size_t)maxc + 1, sizeof(double));
  for (int64_t v = 0; v < n; ++v)
    for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
      m2 += w[e];
      K[membership[v]] += w[e];
      if (membership[indices[e]] == membership[v]) in += w[e];
    }
  double q = in;
  for (int32_t c = 0; c <= maxc; ++c) q -= gamma * K[c] * K[c] / m2;
  free(K);
  return m2 > 0 ? q / m2 : 0.0;
}

/* ---- driver ------------------------------------------------------------------------------------ */
static int cmp_size_desc(const void* a, const void* b) {
  const int64_t* x = (const int64_t*)a;
  const int64_t* y = (const int64_t*)b;
  if (x[0] != y[0]) return x[0] > y[0] ? -1 : 1; /* size desc */
  return x[1] < y[1] ? -1 : (x[1] > y[1]); /* first member asc */
}

int oracle_leiden(int64_t n, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
                  double beta, int n_iterations, uint64_t seed, int32_t* membership, double* quality,
                  int32_t* n_communities) {
  if (n <= 0) return -1;
  rng_state = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  graph_t g0;
  g0.n = n;
  g0.indptr = (int64_t*)indptr;
  g0.indices = (int32_t*)indices;
  g0.w = (double*)weights;
  g0.owns = 0;
  compute_strength(&g0);
  double m2 = 0.0;
  for (int64_t v = 0; v < n; ++v) m2 += g0.k[v];
  for (int64_t v = 0; v < n; ++v) membership[v] = (int32_t)v;
  if (m2 <= 0.0) {
    *quality = 0.0;
    *n_communities = (int32_t)n;
    free(g0.k);
    return 0;
  }
  double q_prev = oracle_modularity(n, indptr, indices, weights, membership, gamma);
  int iter = 0;
  const int max_iter = n_iterations < 0 ? 1000 : n_iterations;
  while (iter < max_iter) {
    /* one Leiden iteration: levels until nothing aggregates */
    graph_t g = g0;
    int32_t* comm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);     /* partition of level nodes */
    int32_t* node_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);  /* original vertex -> level node */
    memcpy(comm, membership, sizeof(int32_t) * (size_t)n);
    relabel(comm, n, n);
    for (int64_t v = 0; v < n; ++v) node_of[v] = (int32_t)v;
    int level = 0;
    for (;;) {
      move_nodes(&g, comm, gamma, m2);
      int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.n);
      memcpy(tmp, comm, sizeof(int32_t) * (size_t)g.n);
      const int64_t ncomm = relabel(tmp, g.n, g.n);
      free(tmp);
      if (ncomm == g.n) break; /* every node its own community: done */
      int32_t* ref = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.n);
      refine(&g, comm, ref, gamma, beta, m2);
      const int64_t nref = relabel(ref, g.n, g.n);
      if (nref == g.n) { /* refinement merged nothing: aggregate on the non-refined partition instead */
        free(ref);
        break;
      }
      if (getenv("ORACLE_LEIDEN_DEBUG") && getenv("ORACLE_LEIDEN_DEBUG")[0] == '2')
        fprintf(stderr, "[oracle leiden] iteration %d level %d: n = %lld, nnz = %lld, communities %lld -> %lld refined\n", iter + 1, level,
                (long long)g.n, (long long)g.indptr[g.n], (long long)ncomm, (long long)nref);
      graph_t a = aggregate(&g, ref, nref);
      int32_t* comm2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)nref);
      for (int64_t v = 0; v < g.n; ++v) comm2[ref[v]] = comm[v];
      relabel(comm2, nref, g.n);
      for (int64_t v = 0; v < n; ++v) node_of[v] = ref[node_of[v]];
      free(ref);
      free(comm);
      comm = comm2;
      if (level > 0) graph_free(&g);
      g = a;
      ++level;
    }
    for (int64_t v = 0; v < n; ++v) membership[v] = comm[node_of[v]];
    if (level > 0) graph_free(&g);
    free(comm);
    free(node_of);
    ++iter;
    const double q = oracle_modularity(n, indptr, indices, weights, membership, gamma);
    if (getenv("ORACLE_LEIDEN_DEBUG")) fprintf(stderr, "[oracle leiden] iteration %d: Q = %.10f (best before %.10f)\n", iter, q, q_prev);
    const int improved = q > q_prev + 1e-12;
    q_prev = q > q_prev ? q : q_prev;
    if (n_iterations < 0 && !improved) break;
  }
  /* consecutive ids by decreasing size */
  const int64_t nc = relabel(membership, n, n);
  int64_t* stat = (int64_t*)calloc((size_t)nc * 3, sizeof(int64_t)); /* size, first member, old id */
  for (int64_t c = 0; c < nc; ++c) {
    stat[3 * c + 1] = n;
    stat[3 * c + 2] = c;
  }
  for (int64_t v = 0; v < n; ++v) {
    stat[3 * membership[v]]++;
    if (v < stat[3 * membership[v] + 1]) stat[3 * membership[v] + 1] = v;
  }
  qsort(stat, (size_t)nc, sizeof(int64_t) * 3, cmp_size_desc);
  int32_t* newid = (int32_t*)malloc(sizeof(int32_t) * (size_t)nc);
  for (int64_t r = 0; r < nc; ++r) newid[stat[3 * r + 2]] = (int32_t)r;
  for (int64_t v = 0; v < n; ++v) membership[v] = newid[membership[v]];
  free(newid);
  free(stat);
  *n_communities = (int32_t)nc;
  *quality = oracle_modularity(n, indptr, indices, weights, membership, gamma);
  free(g0.k);
  return 0;
}
