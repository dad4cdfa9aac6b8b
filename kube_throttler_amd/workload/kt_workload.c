/*
 * kt_workload.c — synthetic snapshot generator (see kt_workload.h and DESIGN.md "Workloads").
 * Distributions follow SURVEY.md 8d; every entity draws from its own splitmix64 stream.
 */
#include "kt_workload.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- PRNG */
typedef struct {
  uint64_t s;
} rng_t;
static inline uint64_t sm64(rng_t* r) {
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline rng_t stream(uint64_t seed, uint64_t tag, uint64_t index) {
  rng_t r = {seed ^ (tag * 0xD6E8FEB86659FD93ull)};
  r.s += index * 0xA0761D6478BD642Full;
  sm64(&r);
  return r;
}
static inline uint32_t below(rng_t* r, uint32_t n) { return (uint32_t)((sm64(r) >> 32) * (uint64_t)n >> 32); }
static inline double unit(rng_t* r) { return (double)(sm64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double between(rng_t* r, double a, double b) { return a + (b - a) * unit(r); }
static inline int chance(rng_t* r, double p) { return unit(r) < p; }

enum { TAG_POD = 1, TAG_THR = 2, TAG_NS = 3 };

/* ---------------------------------------------------------------- resource dimensions */
static const double DIM_P[8] = {.9, .9, .3, .2, .15, .15, .15, .15};
static int64_t dim_value(rng_t* r, int d) {
  switch (d) {
    case 0: return 50 * (int64_t)(1 + below(r, 40));                 /* cpu, milli: 50m..2000m     */
    case 1: return (64ll << 20) * (int64_t)(1 + below(r, 128));      /* memory, bytes: 64Mi..8Gi   */
    case 2: return (1ll << 30) * (int64_t)(1 + below(r, 32));        /* ephemeral-storage, bytes   */
    case 3: return 1ll << below(r, 4);                               /* amd.com/gpu: 1,2,4,8       */
    case 4: return (2ll << 20) * (int64_t)(1 + below(r, 64));        /* hugepages-2Mi, bytes       */
    default: return 1 + below(r, 16);                                /* example.com/{a,b,c}        */
  }
}
static double dim_mean(int d) {
  switch (d) {
    case 0: return 50 * 20.5;
    case 1: return (double)(64ll << 20) * 64.5;
    case 2: return (double)(1ll << 30) * 16.5;
    case 3: return 3.75;
    case 4: return (double)(2ll << 20) * 32.5;
    default: return 8.5;
  }
}
static double dim_p(int d) { return DIM_P[d < 8 ? d : 7]; }
static uint32_t draw_requests(rng_t* r, int D, int64_t* v) {
  uint32_t present = 0;
  for (int d = 0; d < D; ++d) {
    v[d] = 0;
    if (!chance(r, dim_p(d))) continue;
    present |= 1u << d;
    v[d] = chance(r, .05) ? 0 : dim_value(r, d); /* 5 % explicit zero */
  }
  return present;
}

/* ---------------------------------------------------------------- presets */
#define KT_SEED_BASE 0x6B7468726F74ull
#define KT_NOW_2026 1767225600ll /* 2026-01-01T00:00:00Z */

int kt_workload_preset(int index, kt_workload_cfg* c) {
  memset(c, 0, sizeof(*c));
  c->seed = KT_SEED_BASE + (uint64_t)index;
  c->now_s = KT_NOW_2026;
  c->terms_min = c->terms_max = 1;
  switch (index) {
    case 1: /* 10k pods x 100 Throttles, D=4, single selectorTerm */
      c->n_pods_total = 10000; c->n_thr = 100; c->n_cluster = 0; c->D = 4; c->n_ns = 16;
      c->K = 8; c->V = 8; c->L = 4; c->reqs_min = c->reqs_max = 1;
      break;
    case 2: /* 1M pods x 1k Throttle+ClusterThrottle, D=8 */
    case 3: /* ... with temporaryThresholdOverrides active */
      c->n_pods_total = 1000000; c->n_thr = 1000; c->n_cluster = 500; c->D = 8; c->n_ns = 64;
      c->K = 16; c->V = 16; c->L = 8; c->reqs_min = 1; c->reqs_max = 2;
      c->overrides = index == 3;
      break;
    case 4: /* 10M pods x 10k throttles, multi-term OR-of-AND selectors (whole job; shard pods per rank) */
      c->n_pods_total = 10000000; c->n_thr = 10000; c->n_cluster = 5000; c->D = 8; c->n_ns = 256;
      c->K = 16; c->V = 16; c->L = 8; c->terms_min = 2; c->terms_max = 4; c->reqs_min = 1; c->reqs_max = 3;
      c->rich_ops = 1;
      break;
    default:
      return -1;
  }
  c->pod_begin = 0;
  c->n_pods = c->n_pods_total;
  return 0;
}

/* ---------------------------------------------------------------- allocation helpers */
static void* zalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }
static void amounts_alloc(kt_amounts* a, size_t n, int D) {
  a->v = (int64_t*)zalloc(n * (size_t)D, 8);
  a->present = (uint32_t*)zalloc(n, 4);
  a->count = (int64_t*)zalloc(n, 8);
  a->has_count = (uint8_t*)zalloc(n, 1);
}
static void amounts_free(kt_amounts* a) {
  free(a->v); free(a->present); free(a->count); free(a->has_count);
}
typedef struct {
  kt_reqs r;
  uint32_t cap, vcap;
} reqpool;
static void pool_add(reqpool* p, uint8_t op, uint32_t key, const uint32_t* vals, uint32_t nvals) {
  if (p->r.n + 1 >= p->cap) {
    p->cap = p->cap ? p->cap * 2 : 1024;
    p->r.op = (uint8_t*)realloc(p->r.op, p->cap);
    p->r.key = (uint32_t*)realloc(p->r.key, (size_t)p->cap * 4);
    p->r.val_off = (uint32_t*)realloc(p->r.val_off, ((size_t)p->cap + 1) * 4);
    if (p->r.n == 0) p->r.val_off[0] = 0;
  }
  uint32_t off = p->r.val_off[p->r.n];
  if (off + nvals + 1 >= p->vcap) {
    p->vcap = p->vcap ? p->vcap * 2 + nvals : 4096;
    p->r.val = (uint32_t*)realloc(p->r.val, (size_t)p->vcap * 4);
  }
  for (uint32_t i = 0; i < nvals; ++i) p->r.val[off + i] = vals[i];
  p->r.op[p->r.n] = op;
  p->r.key[p->r.n] = key;
  p->r.val_off[p->r.n + 1] = off + nvals;
  p->r.n++;
}
static void pool_init(reqpool* p) {
  memset(p, 0, sizeof(*p));
  p->cap = 1024;
  p->vcap = 4096;
  p->r.op = (uint8_t*)malloc(p->cap);
  p->r.key = (uint32_t*)malloc((size_t)p->cap * 4);
  p->r.val_off = (uint32_t*)malloc(((size_t)p->cap + 1) * 4);
  p->r.val = (uint32_t*)malloc((size_t)p->vcap * 4);
  p->r.val_off[0] = 0;
}

/* ---------------------------------------------------------------- id layout */
static inline uint32_t pod_key_id(int k) { return 1u + (uint32_t)k; }
static inline uint32_t pod_pair_id(const kt_workload_cfg* c, int k, int v) { return 1u + (uint32_t)(k * c->V + v); }
#define NS_ZONES 4
#define NS_TEAMS 8
static inline uint32_t ns_key_name(const kt_workload_cfg* c) { return (uint32_t)c->K + 1; }
static inline uint32_t ns_key_zone(const kt_workload_cfg* c) { return (uint32_t)c->K + 2; }
static inline uint32_t ns_key_team(const kt_workload_cfg* c) { return (uint32_t)c->K + 3; }
static inline uint32_t ns_pair_name(const kt_workload_cfg* c, int ns) { return 1u + (uint32_t)(c->K * c->V) + (uint32_t)ns; }
static inline uint32_t ns_pair_zone(const kt_workload_cfg* c, int z) { return 1u + (uint32_t)(c->K * c->V + c->n_ns + z); }
static inline uint32_t ns_pair_team(const kt_workload_cfg* c, int t) { return 1u + (uint32_t)(c->K * c->V + c->n_ns + NS_ZONES + t); }
static inline int ns_zone_of(const kt_workload_cfg* c, int ns) { rng_t r = stream(c->seed, TAG_NS, (uint64_t)ns); return (int)below(&r, NS_ZONES); }
static inline int ns_team_of(const kt_workload_cfg* c, int ns) { rng_t r = stream(c->seed, TAG_NS, (uint64_t)ns); below(&r, NS_ZONES); return (int)below(&r, NS_TEAMS); }

/* ---------------------------------------------------------------- pods */
static void gen_pods(const kt_workload_cfg* c, kt_snapshot* s) {
  const int D = c->D, L = c->L, K = c->K;
  const int64_t n = c->n_pods;
  s->n_pods = n;
  s->pod_ns = (uint32_t*)zalloc((size_t)n, 4);
  s->pod_flags = (uint32_t*)zalloc((size_t)n, 4);
  s->pod_label_off = (uint32_t*)zalloc((size_t)n + 1, 4);
  s->pod_label_key = (uint32_t*)zalloc((size_t)n * (size_t)L, 4);
  s->pod_label_pair = (uint32_t*)zalloc((size_t)n * (size_t)L, 4);
  s->pod_ctr_off = (uint32_t*)zalloc((size_t)n + 1, 4);
  s->pod_ovh_present = (uint32_t*)zalloc((size_t)n, 4);
  s->pod_ovh = (int64_t*)zalloc((size_t)n * (size_t)D, 8);
  /* pass 1: container counts (own stream position 0..1 of each pod) */
  uint64_t n_ctr = 0;
  for (int64_t i = 0; i < n; ++i) {
    rng_t r = stream(c->seed, TAG_POD, (uint64_t)(c->pod_begin + i));
    uint32_t nc = 1 + below(&r, 3), ni = below(&r, 2);
    n_ctr += nc + ni;
    s->pod_ctr_off[i + 1] = (uint32_t)n_ctr;
  }
  s->ctr_init = (uint8_t*)zalloc(n_ctr, 1);
  s->ctr_present = (uint32_t*)zalloc(n_ctr, 4);
  s->ctr_req = (int64_t*)zalloc(n_ctr * (size_t)D, 8);
  int keys[KT_MAX_LABELS * 8];
  for (int64_t i = 0; i < n; ++i) {
    rng_t r = stream(c->seed, TAG_POD, (uint64_t)(c->pod_begin + i));
    uint32_t nc = 1 + below(&r, 3), ni = below(&r, 2);
    s->pod_ns[i] = below(&r, (uint32_t)c->n_ns);
    /* 60 % counted, 5 % scheduled+finished, 5 % other scheduler (scheduled), 30 % pending */
    double u = unit(&r);
    uint32_t f = KT_POD_VALID;
    if (u < .60) f |= KT_POD_SCHED_MATCH | KT_POD_SCHEDULED;
    else if (u < .65) f |= KT_POD_SCHED_MATCH | KT_POD_SCHEDULED | KT_POD_FINISHED;
    else if (u < .70) f |= KT_POD_SCHEDULED;
    else f |= KT_POD_SCHED_MATCH;
    s->pod_flags[i] = f;
    /* L distinct keys (partial Fisher-Yates over K), sorted by key id so rows look like sorted maps */
    for (int k = 0; k < K; ++k) keys[k] = k;
    for (int l = 0; l < L; ++l) {
      int j = l + (int)below(&r, (uint32_t)(K - l));
      int t = keys[l]; keys[l] = keys[j]; keys[j] = t;
    }
    for (int a = 1; a < L; ++a) { /* insertion sort */
      int x = keys[a], b = a - 1;
      while (b >= 0 && keys[b] > x) { keys[b + 1] = keys[b]; --b; }
      keys[b + 1] = x;
    }
    uint32_t lo = (uint32_t)(i * L);
    for (int l = 0; l < L; ++l) {
      s->pod_label_key[lo + l] = pod_key_id(keys[l]);
      s->pod_label_pair[lo + l] = pod_pair_id(c, keys[l], (int)below(&r, (uint32_t)c->V));
    }
    s->pod_label_off[i + 1] = lo + (uint32_t)L;
    uint32_t c0 = s->pod_ctr_off[i];
    for (uint32_t k = 0; k < nc + ni; ++k) {
      s->ctr_init[c0 + k] = k >= nc;
      s->ctr_present[c0 + k] = draw_requests(&r, D, s->ctr_req + (size_t)(c0 + k) * D);
    }
    if (chance(&r, .05)) { /* RuntimeClass overhead on 5 % */
      uint32_t pm = 0;
      for (int d = 0; d < D && d < 2; ++d) {
        pm |= 1u << d;
        s->pod_ovh[i * D + d] = d == 0 ? 10 * (int64_t)(1 + below(&r, 25)) : (1ll << 20) * (int64_t)(8 + below(&r, 120));
      }
      s->pod_ovh_present[i] = pm | 0x80000000u;
    }
  }
}

/* ---------------------------------------------------------------- namespaces */
static void gen_namespaces(const kt_workload_cfg* c, kt_snapshot* s) {
  s->n_ns = c->n_ns;
  s->ns_valid = (uint8_t*)zalloc((size_t)c->n_ns, 1);
  s->ns_label_off = (uint32_t*)zalloc((size_t)c->n_ns + 1, 4);
  s->ns_label_key = (uint32_t*)zalloc((size_t)c->n_ns * 3, 4);
  s->ns_label_pair = (uint32_t*)zalloc((size_t)c->n_ns * 3, 4);
  for (int i = 0; i < c->n_ns; ++i) {
    s->ns_valid[i] = i < c->n_ns - c->n_missing_ns;
    uint32_t o = (uint32_t)i * 3;
    s->ns_label_key[o] = ns_key_name(c);
    s->ns_label_pair[o] = ns_pair_name(c, i);
    s->ns_label_key[o + 1] = ns_key_zone(c);
    s->ns_label_pair[o + 1] = ns_pair_zone(c, ns_zone_of(c, i));
    s->ns_label_key[o + 2] = ns_key_team(c);
    s->ns_label_pair[o + 2] = ns_pair_team(c, ns_team_of(c, i));
    s->ns_label_off[i + 1] = o + 3;
  }
}

/* ---------------------------------------------------------------- throttles */
static double draw_pod_requirement(const kt_workload_cfg* c, rng_t* r, reqpool* pool, int key, int positive_only) {
  /* returns the probability that a random pod satisfies the requirement */
  const double has = (double)c->L / (double)c->K;
  uint32_t vals[3];
  int op = KT_OP_IN;
  if (c->rich_ops && !positive_only) {
    double u = unit(r);
    op = u < .60 ? KT_OP_IN : u < .75 ? KT_OP_NOT_IN : u < .90 ? KT_OP_EXISTS : KT_OP_DOES_NOT_EXIST;
  }
  if (op == KT_OP_IN || op == KT_OP_NOT_IN) {
    uint32_t nv = c->rich_ops ? 1 + below(r, 3) : 1;
    uint32_t v0 = below(r, (uint32_t)c->V);
    for (uint32_t i = 0; i < nv; ++i) vals[i] = pod_pair_id(c, key, (int)((v0 + i) % (uint32_t)c->V));
    pool_add(pool, (uint8_t)op, pod_key_id(key), vals, nv);
    double p = has * (double)nv / (double)c->V;
    return op == KT_OP_IN ? p : 1.0 - p;
  }
  pool_add(pool, (uint8_t)op, pod_key_id(key), NULL, 0);
  return op == KT_OP_EXISTS ? has : 1.0 - has;
}

static void draw_threshold(const kt_workload_cfg* c, rng_t* r, double est_matches, double factor, int64_t* v,
                           uint32_t* present, int64_t* count, uint8_t* has_count, double p_count) {
  const int D = c->D;
  double m = est_matches < 1.0 ? 1.0 : est_matches;
  uint32_t pm = 0;
  for (int d = 0; d < D; ++d) {
    v[d] = 0;
    if (!chance(r, .6)) continue;
    pm |= 1u << d;
    double x = m * 2.0 * dim_p(d) * .95 * dim_mean(d) * factor; /* ~2 containers per pod */
    v[d] = x < 1.0 ? 1 : (int64_t)llround(x);
  }
  if (!pm) {
    pm = 1u;
    double x = m * 2.0 * dim_p(0) * .95 * dim_mean(0) * factor;
    v[0] = x < 1.0 ? 1 : (int64_t)llround(x);
  }
  *present = pm;
  *has_count = (uint8_t)chance(r, p_count);
  *count = *has_count ? (int64_t)llround(m * factor) : 0;
}

static void gen_throttles(const kt_workload_cfg* c, kt_snapshot* s) {
  const int D = c->D, T = c->n_thr;
  s->n_thr = T;
  s->thr_flags = (uint32_t*)zalloc((size_t)T, 4);
  s->thr_ns = (uint32_t*)zalloc((size_t)T, 4);
  amounts_alloc(&s->thr_spec, (size_t)T, D);
  amounts_alloc(&s->thr_calc, (size_t)T, D);
  amounts_alloc(&s->thr_used, (size_t)T, D);
  amounts_alloc(&s->thr_reserved, (size_t)T, D);
  s->thr_thrl_flag = (uint32_t*)zalloc((size_t)T, 4);
  s->thr_thrl_has = (uint32_t*)zalloc((size_t)T, 4);
  s->thr_status_msgs_fp = (uint64_t*)zalloc((size_t)T, 8);
  s->thr_spec_msgs_fp = (uint64_t*)zalloc((size_t)T, 8);
  s->thr_ovr_off = (uint32_t*)zalloc((size_t)T + 1, 4);
  s->thr_term_off = (uint32_t*)zalloc((size_t)T + 1, 4);
  const size_t max_ovr = (size_t)T * 3, max_term = (size_t)T * ((size_t)c->terms_max + 2);
  s->ovr_begin_s = (int64_t*)zalloc(max_ovr, 8);
  s->ovr_begin_ns = (int32_t*)zalloc(max_ovr, 4);
  s->ovr_end_s = (int64_t*)zalloc(max_ovr, 8);
  s->ovr_end_ns = (int32_t*)zalloc(max_ovr, 4);
  s->ovr_flags = (uint8_t*)zalloc(max_ovr, 1);
  amounts_alloc(&s->ovr_thr, max_ovr, D);
  s->term_flags = (uint8_t*)zalloc(max_term, 1);
  s->term_preq_off = (uint32_t*)zalloc(max_term + 1, 4);
  s->term_nreq_off = (uint32_t*)zalloc(max_term + 1, 4);
  reqpool preq, nreq;
  pool_init(&preq);
  pool_init(&nreq);
  const int n_namespaced = T - c->n_cluster;
  const double counted = .60 * (double)c->n_pods_total;
  uint32_t n_term = 0, n_ovr = 0;
  int used_keys[KT_MAX_LABELS * 8];
  for (int t = 0; t < T; ++t) {
    rng_t r = stream(c->seed, TAG_THR, (uint64_t)t);
    const int cluster = t >= n_namespaced;
    uint32_t f = KT_THR_VALID | (cluster ? KT_THR_CLUSTER : 0);
    if (!chance(&r, .005)) f |= KT_THR_RESPONSIBLE; /* 0.5 % belong to another throttler */
    s->thr_ns[t] = cluster ? 0 : below(&r, (uint32_t)c->n_ns);
    /* ---- selector */
    int n_terms = c->terms_min + (int)below(&r, (uint32_t)(c->terms_max - c->terms_min + 1));
    double p_none = 1.0;
    const int invalid_pod_sel = c->n_invalid_pod_sel > 0 && (t % (T / c->n_invalid_pod_sel > 0 ? T / c->n_invalid_pod_sel : 1)) == 0 &&
                                t / (T / c->n_invalid_pod_sel > 0 ? T / c->n_invalid_pod_sel : 1) < c->n_invalid_pod_sel;
    const int ci = t - n_namespaced;
    const int invalid_ns_sel = cluster && c->n_invalid_ns_sel > 0 && ci < c->n_invalid_ns_sel;
    for (int j = 0; j < n_terms; ++j) {
      double p_term = 1.0;
      if (cluster) {
        double u = unit(&r);
        uint32_t val;
        if (u < .85 || (!c->rich_ops && u >= .95)) {
          val = ns_pair_zone(c, (int)below(&r, NS_ZONES));
          pool_add(&nreq, KT_OP_IN, ns_key_zone(c), &val, 1);
          p_term *= 1.0 / NS_ZONES;
        } else if (u < .90) {
          p_term *= 1.0; /* empty namespaceSelector: every namespace */
        } else if (u < .95) {
          val = ns_pair_name(c, (int)below(&r, (uint32_t)c->n_ns));
          pool_add(&nreq, KT_OP_IN, ns_key_name(c), &val, 1);
          p_term *= 1.0 / (double)c->n_ns;
        } else {
          val = ns_pair_zone(c, (int)below(&r, NS_ZONES));
          pool_add(&nreq, KT_OP_NOT_IN, ns_key_zone(c), &val, 1);
          p_term *= 1.0 - 1.0 / NS_ZONES;
        }
        if (invalid_ns_sel && j == 0) s->term_flags[n_term] |= KT_TERM_NS_SEL_INVALID;
      } else {
        p_term *= 1.0 / (double)c->n_ns;
      }
      int n_reqs = c->reqs_min + (int)below(&r, (uint32_t)(c->reqs_max - c->reqs_min + 1));
      int nk = 0;
      for (int q = 0; q < n_reqs; ++q) {
        int key;
        for (;;) { /* distinct keys inside a term */
          key = (int)below(&r, (uint32_t)c->K);
          int dup = 0;
          for (int a = 0; a < nk; ++a) dup |= used_keys[a] == key;
          if (!dup) break;
        }
        used_keys[nk++] = key;
        p_term *= draw_pod_requirement(c, &r, &preq, key, q == 0);
      }
      s->term_preq_off[n_term + 1] = preq.r.n;
      s->term_nreq_off[n_term + 1] = nreq.r.n;
      n_term++;
      p_none *= 1.0 - p_term;
    }
    if (invalid_pod_sel) { /* an extra, unconvertible LAST term (e.g. In with no values) */
      s->term_flags[n_term] |= KT_TERM_POD_SEL_INVALID;
      s->term_preq_off[n_term + 1] = preq.r.n;
      s->term_nreq_off[n_term + 1] = nreq.r.n;
      n_term++;
    }
    s->thr_term_off[t + 1] = n_term;
    /* ---- threshold: 1/3 already throttled, 1/3 nearly full, 1/3 open; 1 % below a single pod */
    const double est = counted * (1.0 - p_none);
    double u = unit(&r);
    double factor = u < 1.0 / 3 ? between(&r, .3, .9) : u < 2.0 / 3 ? between(&r, 1.0, 1.03) : between(&r, 1.5, 4.0);
    draw_threshold(c, &r, est, factor, s->thr_spec.v + (size_t)t * D, &s->thr_spec.present[t], &s->thr_spec.count[t],
                   &s->thr_spec.has_count[t], .5);
    if (chance(&r, .01)) {
      s->thr_spec.present[t] |= 1u;
      s->thr_spec.v[(size_t)t * D] = 100; /* cpu: 100m, below most single-pod requests */
    }
    /* ---- reserved amounts of the scheduler-side cache on 10 % */
    if (chance(&r, .10)) {
      int64_t np = 1 + below(&r, 8);
      s->thr_reserved.has_count[t] = 1;
      s->thr_reserved.count[t] = np;
      for (int d = 0; d < D; ++d)
        if (chance(&r, dim_p(d))) {
          s->thr_reserved.present[t] |= 1u << d;
          s->thr_reserved.v[(size_t)t * D + d] = np * dim_value(&r, d);
        }
    }
    /* ---- temporaryThresholdOverrides */
    if (c->overrides) {
      int no = 2 + (int)below(&r, 2);
      uint64_t fp = 0;
      for (int j = 0; j < no; ++j) {
        uint32_t o = n_ovr++;
        double k = unit(&r);
        int64_t span_a = 3600 + (int64_t)below(&r, 30 * 86400), span_b = 3600 + (int64_t)below(&r, 30 * 86400);
        s->ovr_begin_ns[o] = 0;
        s->ovr_end_ns[o] = 0;
        if (k < .50) { /* active */
          s->ovr_begin_s[o] = chance(&r, .1) ? KT_ZERO_TIME_S : c->now_s - span_a;
          s->ovr_end_s[o] = chance(&r, .1) ? KT_ZERO_TIME_S : c->now_s + span_b;
          if (chance(&r, .02)) { s->ovr_end_s[o] = c->now_s; }          /* inclusive end boundary */
          else if (chance(&r, .02)) { s->ovr_begin_s[o] = c->now_s; }   /* inclusive begin boundary */
        } else if (k < .75) { /* expired */
          s->ovr_begin_s[o] = c->now_s - span_a - span_b;
          s->ovr_end_s[o] = c->now_s - span_b;
          if (chance(&r, .05)) { s->ovr_end_s[o] = c->now_s - 1; s->ovr_end_ns[o] = 999999999; }
        } else if (k < .95) { /* future */
          s->ovr_begin_s[o] = c->now_s + span_a;
          s->ovr_end_s[o] = c->now_s + span_a + span_b;
          if (chance(&r, .05)) { s->ovr_begin_s[o] = c->now_s; s->ovr_begin_ns[o] = 1; }
        } else { /* unparsable begin/end */
          s->ovr_flags[o] |= KT_OVR_PARSE_ERROR;
          s->ovr_begin_s[o] = KT_ZERO_TIME_S;
          s->ovr_end_s[o] = KT_ZERO_TIME_S;
          if (chance(&r, .5)) { /* only `end` is unparsable: begin is a real instant, before or after now */
            s->ovr_flags[o] |= KT_OVR_BEGIN_PARSED;
            s->ovr_begin_s[o] = chance(&r, .5) ? c->now_s + span_a : c->now_s - span_a;
          }
          fp = fp * 1099511628211ull + (uint64_t)(j + 1) + ((uint64_t)t << 8) + 0x9E3779B97F4A7C15ull;
        }
        double of = between(&r, .3, 4.0);
        draw_threshold(c, &r, est, of, s->ovr_thr.v + (size_t)o * D, &s->ovr_thr.present[o], &s->ovr_thr.count[o],
                       &s->ovr_thr.has_count[o], .7);
      }
      s->thr_spec_msgs_fp[t] = fp;
    }
    s->thr_ovr_off[t + 1] = n_ovr;
    s->thr_flags[t] = f;
  }
  s->preq = preq.r;
  s->nreq = nreq.r;
}

kt_snapshot* kt_workload_generate(const kt_workload_cfg* c) {
  if (c->D < 1 || c->D > KT_MAX_DIMS || c->L < 1 || c->L > KT_MAX_LABELS || c->L > c->K || c->K > KT_MAX_LABELS * 8 ||
      c->n_ns < 1 || c->n_cluster > c->n_thr || c->reqs_max > c->K || c->reqs_min < 0 || c->terms_min < 0 ||
      c->terms_max < c->terms_min || c->reqs_max < c->reqs_min)
    return NULL;
  kt_snapshot* s = (kt_snapshot*)calloc(1, sizeof(kt_snapshot));
  s->D = c->D;
  s->L = c->L;
  gen_namespaces(c, s);
  gen_pods(c, s);
  gen_throttles(c, s);
  return s;
}

void kt_workload_free(kt_snapshot* s) {
  if (!s) return;
  free(s->ns_valid); free(s->ns_label_off); free(s->ns_label_key); free(s->ns_label_pair);
  free(s->pod_ns); free(s->pod_flags); free(s->pod_label_off); free(s->pod_label_key); free(s->pod_label_pair);
  free(s->pod_ctr_off); free(s->ctr_init); free(s->ctr_present); free(s->ctr_req); free(s->pod_ovh_present);
  free(s->pod_ovh);
  free(s->thr_flags); free(s->thr_ns);
  amounts_free(&s->thr_spec); amounts_free(&s->thr_calc); amounts_free(&s->thr_used); amounts_free(&s->thr_reserved);
  free(s->thr_thrl_flag); free(s->thr_thrl_has); free(s->thr_status_msgs_fp); free(s->thr_spec_msgs_fp);
  free(s->thr_ovr_off); free(s->ovr_begin_s); free(s->ovr_begin_ns); free(s->ovr_end_s); free(s->ovr_end_ns);
  free(s->ovr_flags); amounts_free(&s->ovr_thr);
  free(s->thr_term_off); free(s->term_flags); free(s->term_preq_off); free(s->term_nreq_off);
  free(s->preq.op); free(s->preq.key); free(s->preq.val_off); free(s->preq.val);
  free(s->nreq.op); free(s->nreq.key); free(s->nreq.val_off); free(s->nreq.val);
  free(s);
}
