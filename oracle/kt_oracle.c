/*
 * kt_oracle.c — CPU oracle for the kube-throttler hot path.  TEST INFRASTRUCTURE ONLY
 * (see kt_oracle.h for who may load it and for the parity status).
 *
 * Structure mirrors the reference object-by-object, NOT the GPU engine's dense algebra:
 *   - ResourceList is a sparse name->Quantity map (here: small insertion-ordered array),
 *   - Quantity is an exact integer (__int128, so no sum of int64 inputs can overflow),
 *   - every (pod, throttle) check rebuilds selectors and pod request lists the way the Go code does.
 * All file:line citations are into /root/reference.
 */
#include "kt_oracle.h"

#include <stdbool.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 q_t; /* resource.Quantity value at the dimension's fixed scale (exact) */

/* ------------------------------------------------------------------------------------------ */
/* corev1.ResourceList: map[ResourceName]resource.Quantity                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int n;
  int32_t name[KT_MAX_DIMS];
  q_t q[KT_MAX_DIMS];
} rl_t;

static int rl_find(const rl_t* m, int32_t name) {
  for (int i = 0; i < m->n; ++i)
    if (m->name[i] == name) return i;
  return -1;
}
static void rl_put(rl_t* m, int32_t name, q_t q) {
  int i = rl_find(m, name);
  if (i < 0) {
    i = m->n++;
    m->name[i] = name;
  }
  m->q[i] = q;
}
static void rl_from_dense(rl_t* m, const int64_t* v, uint32_t present, int D) {
  m->n = 0;
  for (int d = 0; d < D; ++d)
    if (present >> d & 1u) rl_put(m, d, (q_t)v[d]);
}

/* ResourceList.Add — pkg/resourcelist/resourcelist.go:48-54.
 * lhs[name] of a missing key is the zero Quantity, and the key is created even when adding 0. */
static void rl_add(rl_t* lhs, const rl_t* rhs) {
  for (int i = 0; i < rhs->n; ++i) {
    int j = rl_find(lhs, rhs->name[i]);
    q_t qlhs = j >= 0 ? lhs->q[j] : 0;
    qlhs += rhs->q[i];
    rl_put(lhs, rhs->name[i], qlhs);
  }
}

/* quantityMax — resourcelist.go:113-123 (ties keep qx). */
static q_t quantity_max(q_t qx, q_t qy) { return qx >= qy ? qx : qy; }

/* ResourceList.SetMax — resourcelist.go:76-84: missing key => copy rhs (even if zero). */
static void rl_set_max(rl_t* lhs, const rl_t* rhs) {
  for (int i = 0; i < rhs->n; ++i) {
    int j = rl_find(lhs, rhs->name[i]);
    if (j >= 0) {
      lhs->q[j] = quantity_max(lhs->q[j], rhs->q[i]);
      continue;
    }
    rl_put(lhs, rhs->name[i], rhs->q[i]);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* v1alpha1.ResourceAmount / IsResourceAmountThrottled — resource_amount.go:28-44              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  bool counts_nonnil; /* ResourceCounts != nil */
  int64_t pod;        /* ResourceCounts.Pod    */
  bool req_nonnil;    /* ResourceRequests != nil (nil and empty behave alike in all arithmetic) */
  rl_t req;
} ra_t;

typedef struct {
  bool pod; /* ResourceCounts.Pod */
  int n;    /* ResourceRequests map[ResourceName]bool */
  int32_t name[KT_MAX_DIMS];
  bool val[KT_MAX_DIMS];
} irat_t;

static void ra_zero(ra_t* a) {
  a->counts_nonnil = false;
  a->pod = 0;
  a->req_nonnil = false;
  a->req.n = 0;
}
static void ra_from_row(ra_t* a, const kt_amounts* t, int64_t row, int D) {
  a->counts_nonnil = t->has_count[row] != 0;
  a->pod = a->counts_nonnil ? t->count[row] : 0;
  rl_from_dense(&a->req, t->v + row * D, t->present[row], D);
  a->req_nonnil = a->req.n > 0;
}

/* ResourceAmount.Add — resource_amount.go:91-110 (+ ResourceCounts.Add :78-81). */
static ra_t ra_add(ra_t a, const ra_t* b) {
  if (!a.req_nonnil) {
    a.req_nonnil = true;
    a.req.n = 0;
  }
  if (!a.counts_nonnil) {
    if (b->counts_nonnil) {
      a.counts_nonnil = true;
      a.pod = b->pod;
    }
  } else {
    if (b->counts_nonnil) a.pod += b->pod;
  }
  rl_add(&a.req, &b->req);
  return a;
}

/* ResourceAmount.IsThrottled — resource_amount.go:127-159. */
static irat_t ra_is_throttled(const ra_t* threshold, const ra_t* used, bool on_equal) {
  irat_t r;
  r.pod = false;
  r.n = 0;
  if (threshold->counts_nonnil && used->counts_nonnil)
    r.pod = on_equal ? used->pod >= threshold->pod : used->pod > threshold->pod;
  for (int i = 0; i < threshold->req.n; ++i) {
    int32_t rn = threshold->req.name[i];
    q_t qt = threshold->req.q[i];
    int j = rl_find(&used->req, rn);
    bool f = false;
    if (j >= 0) f = on_equal ? used->req.q[j] >= qt : used->req.q[j] > qt;
    r.name[r.n] = rn;
    r.val[r.n] = f;
    r.n++;
  }
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* pods                                                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const kt_snapshot* s;
  int64_t row;
} pod_t;

/* resourcelist.PodRequestResourceList — resourcelist.go:27-46. */
static rl_t pod_request_resource_list(pod_t p) {
  const kt_snapshot* s = p.s;
  const int D = s->D;
  rl_t ic_res, c_res, tmp;
  ic_res.n = 0;
  c_res.n = 0;
  for (uint32_t c = s->pod_ctr_off[p.row]; c < s->pod_ctr_off[p.row + 1]; ++c) {
    if (!s->ctr_init[c]) continue;
    rl_from_dense(&tmp, s->ctr_req + (int64_t)c * D, s->ctr_present[c], D);
    rl_set_max(&ic_res, &tmp);
  }
  for (uint32_t c = s->pod_ctr_off[p.row]; c < s->pod_ctr_off[p.row + 1]; ++c) {
    if (s->ctr_init[c]) continue;
    rl_from_dense(&tmp, s->ctr_req + (int64_t)c * D, s->ctr_present[c], D);
    rl_add(&c_res, &tmp);
  }
  rl_set_max(&c_res, &ic_res);
  if (s->pod_ovh_present[p.row] >> 31) {
    rl_from_dense(&tmp, s->pod_ovh + p.row * D, s->pod_ovh_present[p.row] & 0x7fffffffu, D);
    rl_add(&c_res, &tmp);
  }
  return c_res;
}

/* ResourceAmountOfPod — resource_amount.go:71-76. */
static ra_t resource_amount_of_pod(pod_t p) {
  ra_t a;
  a.counts_nonnil = true;
  a.pod = 1;
  a.req_nonnil = true;
  a.req = pod_request_resource_list(p);
  return a;
}

/* IsResourceAmountThrottled.IsThrottledFor — resource_amount.go:46-65. */
static bool irat_is_throttled_for(const irat_t* t, pod_t p) {
  if (t->pod) return true;
  ra_t pa = resource_amount_of_pod(p);
  for (int i = 0; i < pa.req.n; ++i) {
    if (pa.req.q[i] == 0) continue; /* rq.IsZero() */
    int found = -1;
    for (int j = 0; j < t->n; ++j)
      if (t->name[j] == pa.req.name[i]) {
        found = j;
        break;
      }
    if (found < 0) continue;
    if (t->val[found]) return true;
  }
  return false;
}

/* ------------------------------------------------------------------------------------------ */
/* label selectors (k8s.io/apimachinery v0.26.4 pkg/apis/meta/v1 + pkg/labels, restated)       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint32_t* key;
  const uint32_t* pair;
  int n;
} labels_t;

typedef struct {
  const kt_reqs* pool;
  uint32_t begin, end;
} selector_t;

/* metav1.LabelSelectorAsSelector: (selector, err).  Validity is a property of the selector text and
 * is decided by the caller at ingest (KT_TERM_*_INVALID); an empty selector matches everything. */
static bool label_selector_as_selector(const kt_reqs* pool, uint32_t begin, uint32_t end, bool invalid,
                                       selector_t* out) {
  if (invalid) return false; /* err != nil */
  out->pool = pool;
  out->begin = begin;
  out->end = end;
  return true;
}

static bool labels_has(labels_t ls, uint32_t key) {
  for (int i = 0; i < ls.n; ++i)
    if (ls.key[i] == key) return true;
  return false;
}
static bool labels_has_pair_in(labels_t ls, uint32_t key, const uint32_t* vals, uint32_t nvals) {
  /* ls.Get(key) is in the requirement's value set <=> the pod's (key,value) pair id is in vals */
  for (int i = 0; i < ls.n; ++i)
    if (ls.key[i] == key) {
      for (uint32_t j = 0; j < nvals; ++j)
        if (vals[j] == ls.pair[i]) return true;
      return false;
    }
  return false;
}

/* labels.Requirement.Matches + internalSelector.Matches: conjunction of requirements. */
static bool selector_matches(const selector_t* sel, labels_t ls) {
  const kt_reqs* p = sel->pool;
  for (uint32_t r = sel->begin; r < sel->end; ++r) {
    const uint32_t* vals = p->val + p->val_off[r];
    uint32_t nvals = p->val_off[r + 1] - p->val_off[r];
    bool ok;
    switch (p->op[r]) {
      case KT_OP_IN:
        ok = labels_has(ls, p->key[r]) && labels_has_pair_in(ls, p->key[r], vals, nvals);
        break;
      case KT_OP_NOT_IN:
        ok = !labels_has(ls, p->key[r]) || !labels_has_pair_in(ls, p->key[r], vals, nvals);
        break;
      case KT_OP_EXISTS:
        ok = labels_has(ls, p->key[r]);
        break;
      case KT_OP_DOES_NOT_EXIST:
        ok = !labels_has(ls, p->key[r]);
        break;
      default:
        ok = false;
    }
    if (!ok) return false;
  }
  return true;
}

static labels_t pod_labels(pod_t p) {
  labels_t ls;
  uint32_t b = p.s->pod_label_off[p.row];
  ls.key = p.s->pod_label_key + b;
  ls.pair = p.s->pod_label_pair + b;
  ls.n = (int)(p.s->pod_label_off[p.row + 1] - b);
  return ls;
}
static labels_t ns_labels(const kt_snapshot* s, uint32_t ns) {
  labels_t ls;
  uint32_t b = s->ns_label_off[ns];
  ls.key = s->ns_label_key + b;
  ls.pair = s->ns_label_pair + b;
  ls.n = (int)(s->ns_label_off[ns + 1] - b);
  return ls;
}

/* ThrottleSelectorTerm.MatchesToPod — throttle_selector.go:48-54.  *err set on conversion error. */
static bool term_matches_to_pod(const kt_snapshot* s, uint32_t term, pod_t p, bool* err) {
  selector_t sel;
  if (!label_selector_as_selector(&s->preq, s->term_preq_off[term], s->term_preq_off[term + 1],
                                  (s->term_flags[term] & KT_TERM_POD_SEL_INVALID) != 0, &sel)) {
    *err = true;
    return false;
  }
  return selector_matches(&sel, pod_labels(p));
}

/* ThrottleSelector.MatchesToPod — throttle_selector.go:30-42 (OR over terms; no terms => false). */
static bool throttle_selector_matches_to_pod(const kt_snapshot* s, int32_t t, pod_t p, bool* err) {
  for (uint32_t term = s->thr_term_off[t]; term < s->thr_term_off[t + 1]; ++term) {
    bool match = term_matches_to_pod(s, term, p, err);
    if (*err) return false;
    if (match) return true;
  }
  return false;
}

/* ClusterThrottleSelectorTerm.MatchesToNamespace — clusterthrottle_selector.go:63-69
 * (conversion error is swallowed: returns false, nil). */
static bool cterm_matches_to_namespace(const kt_snapshot* s, uint32_t term, uint32_t ns) {
  selector_t sel;
  if (!label_selector_as_selector(&s->nreq, s->term_nreq_off[term], s->term_nreq_off[term + 1],
                                  (s->term_flags[term] & KT_TERM_NS_SEL_INVALID) != 0, &sel))
    return false;
  return selector_matches(&sel, ns_labels(s, ns));
}

/* TEST-MODE memo of cterm_matches_to_namespace (kto_enable_ns_memo): the namespace side of a term is a pure function of
 * (term, namespace object), and the literal loop re-evaluates it for every pod of the namespace — 1.25M x 15k times on a
 * configs[4] shard.  memo[term * n_ns + ns]: 0 = not evaluated yet, 1 = false, 2 = true (filled on first use; concurrent
 * threads can only store the same value).  NULL = the literal evaluation (the timed CPU baseline always passes NULL). */
static bool cterm_ns_ok(const kt_snapshot* s, uint8_t* memo, uint32_t term, uint32_t ns) {
  if (!memo) return cterm_matches_to_namespace(s, term, ns);
  uint8_t* m = memo + (size_t)term * (size_t)s->n_ns + ns;
  uint8_t v = *m;
  if (v == 0) {
    v = cterm_matches_to_namespace(s, term, ns) ? 2 : 1;
    *m = v;
  }
  return v == 2;
}

/* ClusterThrottleSelectorTerm.MatchesToPod — clusterthrottle_selector.go:71-87. */
static bool cterm_matches_to_pod(const kt_snapshot* s, uint32_t term, pod_t p, uint32_t ns, bool* err, uint8_t* memo) {
  bool match_ns = cterm_ns_ok(s, memo, term, ns);
  if (!match_ns) return false;
  bool match = term_matches_to_pod(s, term, p, err);
  if (*err) return false;
  return match;
}

/* ClusterThrottleSelector.MatchesToNamespace — clusterthrottle_selector.go:30-42. */
static bool cluster_selector_matches_to_namespace(const kt_snapshot* s, int32_t t, uint32_t ns, uint8_t* memo) {
  for (uint32_t term = s->thr_term_off[t]; term < s->thr_term_off[t + 1]; ++term)
    if (cterm_ns_ok(s, memo, term, ns)) return true;
  return false;
}

/* ClusterThrottleSelector.MatchesToPod — clusterthrottle_selector.go:44-56. */
static bool cluster_selector_matches_to_pod(const kt_snapshot* s, int32_t t, pod_t p, uint32_t ns,
                                            bool* err, uint8_t* memo) {
  for (uint32_t term = s->thr_term_off[t]; term < s->thr_term_off[t + 1]; ++term) {
    bool match = cterm_matches_to_pod(s, term, p, ns, err, memo);
    if (*err) return false;
    if (match) return true;
  }
  return false;
}

/* ------------------------------------------------------------------------------------------ */
/* thresholds                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static int instant_cmp(int64_t as, int32_t an, int64_t bs, int32_t bn) {
  if (as != bs) return as < bs ? -1 : 1;
  if (an != bn) return an < bn ? -1 : 1;
  return 0;
}

/* TemporaryThresholdOverride.IsActive — temporary_threshold_override.go:57-70.
 * Empty begin/end parse to Go's zero time; err on unparsable text. */
static bool override_is_active(const kt_snapshot* s, uint32_t o, int64_t now_s, int32_t now_ns, bool* err) {
  if (s->ovr_flags[o] & KT_OVR_PARSE_ERROR) {
    *err = true;
    return false;
  }
  bool begin = instant_cmp(s->ovr_begin_s[o], s->ovr_begin_ns[o], now_s, now_ns) <= 0;
  bool end_is_zero = s->ovr_end_s[o] == KT_ZERO_TIME_S && s->ovr_end_ns[o] == 0;
  bool end = end_is_zero || instant_cmp(now_s, now_ns, s->ovr_end_s[o], s->ovr_end_ns[o]) <= 0;
  return begin && end;
}

/* ThrottleSpecBase.CalculateThreshold — throttle_types.go:65-106.  *any_err: messages non-empty. */
static ra_t calculate_threshold(const kt_snapshot* s, int32_t t, int64_t now_s, int32_t now_ns, bool* any_err) {
  ra_t calculated;
  ra_from_row(&calculated, &s->thr_spec, t, s->D);
  bool active_found = false;
  ra_t override_result;
  ra_zero(&override_result);
  override_result.req_nonnil = true;
  *any_err = false;
  for (uint32_t o = s->thr_ovr_off[t]; o < s->thr_ovr_off[t + 1]; ++o) {
    bool err = false;
    bool is_active = override_is_active(s, o, now_s, now_ns, &err);
    if (err) {
      *any_err = true;
      continue;
    }
    if (is_active) {
      active_found = true;
      ra_t othr;
      ra_from_row(&othr, &s->ovr_thr, o, s->D);
      if (!override_result.counts_nonnil && othr.counts_nonnil) {
        override_result.counts_nonnil = true;
        override_result.pod = othr.pod;
      }
      for (int i = 0; i < othr.req.n; ++i)
        if (rl_find(&override_result.req, othr.req.name[i]) < 0)
          rl_put(&override_result.req, othr.req.name[i], othr.req.q[i]);
    }
  }
  if (active_found) calculated = override_result;
  return calculated;
}

/* apiequality.Semantic.DeepEqual on two ResourceAmount values: pointers both nil or pointees equal,
 * nil map == empty map, Quantity by Cmp. */
static bool ra_semantic_equal(const ra_t* a, const ra_t* b) {
  if (a->counts_nonnil != b->counts_nonnil) return false;
  if (a->counts_nonnil && a->pod != b->pod) return false;
  if (a->req.n != b->req.n) return false;
  for (int i = 0; i < a->req.n; ++i) {
    int j = rl_find(&b->req, a->req.name[i]);
    if (j < 0 || a->req.q[i] != b->req.q[j]) return false;
  }
  return true;
}

/* ------------------------------------------------------------------------------------------ */
/* CheckThrottledFor — throttle_types.go:128-153 / clusterthrottle_types.go:30-55              */
/* ------------------------------------------------------------------------------------------ */
static irat_t stored_throttled(const kt_snapshot* s, int32_t t) {
  irat_t f;
  f.pod = (s->thr_flags[t] & KT_THR_THROTTLED_POD) != 0;
  f.n = 0;
  for (int d = 0; d < s->D; ++d)
    if (s->thr_thrl_has[t] >> d & 1u) {
      f.name[f.n] = d;
      f.val[f.n] = (s->thr_thrl_flag[t] >> d & 1u) != 0;
      f.n++;
    }
  return f;
}

struct kto_ctx;
static void status_used_of(const struct kto_ctx* c, int32_t t, ra_t* a);

static int check_throttled_for(const struct kto_ctx* c, const kt_snapshot* s, int32_t t, pod_t pod, const ra_t* reserved, bool on_equal) {
  const bool is_cluster = (s->thr_flags[t] & KT_THR_CLUSTER) != 0;
  ra_t threshold;
  ra_from_row(&threshold, &s->thr_spec, t, s->D);
  if (s->thr_flags[t] & KT_THR_CALC_AT_NONZERO) ra_from_row(&threshold, &s->thr_calc, t, s->D);

  ra_t pa = resource_amount_of_pod(pod);
  irat_t f = ra_is_throttled(&threshold, &pa, false);
  if (irat_is_throttled_for(&f, pod)) return KTO_EXCEEDS;

  irat_t st = stored_throttled(s, t);
  if (irat_is_throttled_for(&st, pod)) return KTO_ACTIVE;

  ra_t status_used;
  status_used_of(c, t, &status_used);
  ra_t zero;
  ra_zero(&zero);
  ra_t already_used = ra_add(ra_add(zero, &status_used), reserved);
  /* Throttle hard-codes true here (throttle_types.go:143); ClusterThrottle passes the caller's
   * flag (clusterthrottle_types.go:45). */
  f = ra_is_throttled(&threshold, &already_used, is_cluster ? on_equal : true);
  if (irat_is_throttled_for(&f, pod)) return KTO_ACTIVE;

  pa = resource_amount_of_pod(pod);
  ra_t used = ra_add(ra_add(ra_add(zero, &status_used), &pa), reserved);
  f = ra_is_throttled(&threshold, &used, on_equal);
  if (irat_is_throttled_for(&f, pod)) return KTO_INSUFFICIENT;

  return KTO_NOT_THROTTLED;
}

/* ------------------------------------------------------------------------------------------ */
/* controllers                                                                                 */
/* ------------------------------------------------------------------------------------------ */
struct kto_ctx {
  const kt_snapshot* s;
  /* informer namespace indexes */
  uint32_t* ns_thr_off; /* [n_ns+1] Throttle rows by namespace */
  int32_t* ns_thr;
  int32_t n_cluster;
  int32_t* cluster_thr; /* ClusterThrottle rows */
  uint64_t* ns_pod_off; /* [n_ns+1] pod rows by namespace */
  int64_t* ns_pod;
  /* resource.Quantity never overflows (resourcelist.go:48-54): a status.used beyond int64 keeps its high 64 bits here
   * ([n_thr][D], two's complement with thr_used.v as the low words; NULL = every value is the int64 in thr_used.v), and
   * kto_reconcile writes the high words of what it computes there when asked to (both set by kto_set_wide). */
  const int64_t* status_used_hi;
  int64_t* out_used_hi; /* [n][D] by output position */
  uint8_t* ns_memo;     /* test mode (kto_enable_ns_memo): [terms][n_ns] memo of cterm_matches_to_namespace, or NULL */
};

int kto_enable_ns_memo(kto_ctx* c) {
  const kt_snapshot* s = c->s;
  const uint64_t cells = (uint64_t)s->thr_term_off[s->n_thr] * (uint64_t)(s->n_ns > 0 ? s->n_ns : 1);
  if (c->ns_memo || cells == 0 || cells > (1ull << 29)) return c->ns_memo != NULL;
  c->ns_memo = (uint8_t*)calloc((size_t)cells, 1);
  return c->ns_memo != NULL;
}

void kto_set_wide(kto_ctx* c, const int64_t* status_used_hi, int64_t* out_used_hi) {
  c->status_used_hi = status_used_hi;
  c->out_used_hi = out_used_hi;
}



static void status_used_of(const kto_ctx* c, int32_t t, ra_t* a) {
  const kt_snapshot* s = c->s;
  ra_from_row(a, &s->thr_used, t, s->D);
  if (c->status_used_hi)
    for (int k = 0; k < a->req.n; ++k) {
      const int d = a->req.name[k];
      a->req.q[k] = (q_t)(((unsigned __int128)(uint64_t)c->status_used_hi[(int64_t)t * s->D + d] << 64) |
                          (unsigned __int128)(uint64_t)s->thr_used.v[(int64_t)t * s->D + d]);
    }
}

kto_ctx* kto_create(const kt_snapshot* s) {
  kto_ctx* c = (kto_ctx*)calloc(1, sizeof(kto_ctx));
  c->s = s;
  const int32_t nns = s->n_ns;
  c->ns_thr_off = (uint32_t*)calloc((size_t)nns + 1, sizeof(uint32_t));
  c->ns_pod_off = (uint64_t*)calloc((size_t)nns + 1, sizeof(uint64_t));
  c->ns_thr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(s->n_thr > 0 ? s->n_thr : 1));
  c->cluster_thr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(s->n_thr > 0 ? s->n_thr : 1));
  c->ns_pod = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->n_pods > 0 ? s->n_pods : 1));
  for (int32_t t = 0; t < s->n_thr; ++t) {
    if (!(s->thr_flags[t] & KT_THR_VALID)) continue;
    if (s->thr_flags[t] & KT_THR_CLUSTER)
      c->cluster_thr[c->n_cluster++] = t;
    else if (s->thr_ns[t] < (uint32_t)nns)
      c->ns_thr_off[s->thr_ns[t] + 1]++;
  }
  for (int64_t p = 0; p < s->n_pods; ++p)
    if ((s->pod_flags[p] & KT_POD_VALID) && s->pod_ns[p] < (uint32_t)nns) c->ns_pod_off[s->pod_ns[p] + 1]++;
  for (int32_t i = 0; i < nns; ++i) {
    c->ns_thr_off[i + 1] += c->ns_thr_off[i];
    c->ns_pod_off[i + 1] += c->ns_pod_off[i];
  }
  uint32_t* tcur = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nns + 1));
  uint64_t* pcur = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)nns + 1));
  memcpy(tcur, c->ns_thr_off, sizeof(uint32_t) * ((size_t)nns + 1));
  memcpy(pcur, c->ns_pod_off, sizeof(uint64_t) * ((size_t)nns + 1));
  for (int32_t t = 0; t < s->n_thr; ++t)
    if ((s->thr_flags[t] & KT_THR_VALID) && !(s->thr_flags[t] & KT_THR_CLUSTER) && s->thr_ns[t] < (uint32_t)nns)
      c->ns_thr[tcur[s->thr_ns[t]]++] = t;
  for (int64_t p = 0; p < s->n_pods; ++p)
    if ((s->pod_flags[p] & KT_POD_VALID) && s->pod_ns[p] < (uint32_t)nns) c->ns_pod[pcur[s->pod_ns[p]]++] = p;
  free(tcur);
  free(pcur);
  return c;
}

void kto_destroy(kto_ctx* c) {
  if (!c) return;
  free(c->ns_thr_off);
  free(c->ns_thr);
  free(c->cluster_thr);
  free(c->ns_pod_off);
  free(c->ns_pod);
  free(c->ns_memo);
  free(c);
}

/* isResponsibleFor — throttle_controller.go:213-215 / clusterthrottle_controller.go:216-218. */
static bool is_responsible_for(const kt_snapshot* s, int32_t t) { return (s->thr_flags[t] & KT_THR_RESPONSIBLE) != 0; }
/* shouldCountIn — throttle_controller.go:217-219 (+ isScheduled pod_util.go:22-24). */
static bool should_count_in(const kt_snapshot* s, int64_t p) {
  return (s->pod_flags[p] & KT_POD_SCHED_MATCH) && (s->pod_flags[p] & KT_POD_SCHEDULED);
}
/* isNotFinished — pod_util.go:26-28. */
static bool is_not_finished(const kt_snapshot* s, int64_t p) { return !(s->pod_flags[p] & KT_POD_FINISHED); }

/* reservedResourceAmount — reserved_resource_amounts.go:113-126,148-156: the per-throttle total is
 * an input here (the pod map itself stays host-side). */
static ra_t reserved_resource_amount(const kt_snapshot* s, int32_t t) {
  ra_t r;
  ra_from_row(&r, &s->thr_reserved, t, s->D);
  return r;
}

static volatile int64_t g_log_sink; /* keeps the eager-log-argument work from being optimised away */

/* One controller's CheckThrottled — throttle_controller.go:349-397 / clusterthrottle_controller.go:378-425.
 * Writes statuses into row[] (size n_thr); returns false on error. */
static bool check_throttled(kto_ctx* c, pod_t pod, bool cluster, bool on_equal, uint8_t* row, int mimic_log_args,
                            const ra_t* reserved_now /* nullable: per-throttle totals that replace the snapshot's */) {
  const kt_snapshot* s = c->s;
  const uint32_t ns = s->pod_ns[pod.row];
  const int32_t* cand;
  int32_t ncand;
  if (!cluster) {
    /* affectedThrottles — throttle_controller.go:248-269: Throttles(pod.Namespace).List */
    if (ns < (uint32_t)s->n_ns) {
      cand = c->ns_thr + c->ns_thr_off[ns];
      ncand = (int32_t)(c->ns_thr_off[ns + 1] - c->ns_thr_off[ns]);
    } else {
      cand = NULL;
      ncand = 0;
    }
  } else {
    /* affectedClusterThrottles — clusterthrottle_controller.go:272-298: namespace must exist */
    if (ns >= (uint32_t)s->n_ns || !s->ns_valid[ns]) return false;
    cand = c->cluster_thr;
    ncand = c->n_cluster;
  }
  /* pass 1: affected throttles (any selector error aborts the whole check) */
  for (int32_t i = 0; i < ncand; ++i) {
    int32_t t = cand[i];
    if (!is_responsible_for(s, t)) continue;
    bool err = false;
    bool match = cluster ? cluster_selector_matches_to_pod(s, t, pod, ns, &err, mimic_log_args ? NULL : c->ns_memo)
                         : throttle_selector_matches_to_pod(s, t, pod, &err);
    if (err) return false;
    if (match) row[t] = KTO_NOT_THROTTLED; /* provisional: "affected" */
  }
  /* pass 2: classify each affected throttle */
  for (int32_t i = 0; i < ncand; ++i) {
    int32_t t = cand[i];
    if (row[t] != KTO_NOT_THROTTLED) continue;
    ra_t reserved = reserved_now ? reserved_now[t] : reserved_resource_amount(s, t);
    row[t] = (uint8_t)check_throttled_for(c, s, t, pod, &reserved, on_equal);
    if (mimic_log_args) {
      /* klog.V(3).InfoS arguments are evaluated even when V(3) is off (throttle_controller.go:376-386):
       * one more ResourceAmountOfPod and ResourceAmount{}.Add(used).Add(pod).Add(reserved). */
      ra_t pa = resource_amount_of_pod(pod);
      ra_t su, zero;
      status_used_of(c, t, &su);
      ra_zero(&zero);
      ra_t chk = ra_add(ra_add(ra_add(zero, &su), &pa), &reserved);
      g_log_sink += (int64_t)chk.req.n + chk.pod;
    }
  }
  return true;
}

static uint64_t summarize(const uint8_t* row, int32_t n_thr, bool error) {
  if (error) return KTO_VERDICT_ERROR;
  uint64_t n_exc = 0, n_act = 0, n_ins = 0;
  for (int32_t t = 0; t < n_thr; ++t) {
    n_exc += row[t] == KTO_EXCEEDS;
    n_act += row[t] == KTO_ACTIVE;
    n_ins += row[t] == KTO_INSUFFICIENT;
  }
  /* plugin.go:177-180: Success iff all six lists are empty */
  uint64_t verdict = (n_exc + n_act + n_ins) == 0 ? KTO_VERDICT_ALLOW : KTO_VERDICT_BLOCK;
  return verdict | n_exc << 4 | n_act << 24 | n_ins << 44;
}

/* KubeThrottler.PreFilter — pkg/scheduler_plugin/plugin.go:148-215 (isThrottledOnEqual=false there). */
int kto_check(kto_ctx* c, int64_t n, const int64_t* rows, int on_equal, uint8_t* out_status,
              uint64_t* out_summary, int nthreads, int mimic_log_args) {
  const kt_snapshot* s = c->s;
  const int32_t T = s->n_thr;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint8_t* scratch = (uint8_t*)malloc((size_t)(T > 0 ? T : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < n; ++i) {
      int64_t prow = rows ? rows[i] : i;
      uint8_t* row = out_status ? out_status + i * T : scratch;
      memset(row, KTO_NOT_AFFECTED, (size_t)T);
      bool error = false;
      if (prow >= 0 && prow < s->n_pods && (s->pod_flags[prow] & KT_POD_VALID)) {
        pod_t pod = {s, prow};
        if (!check_throttled(c, pod, false, on_equal != 0, row, mimic_log_args, NULL))
          error = true; /* plugin.go:154-156 */
        else if (!check_throttled(c, pod, true, on_equal != 0, row, mimic_log_args, NULL))
          error = true; /* plugin.go:166-168 */
      }
      if (error) memset(row, KTO_ERROR, (size_t)T);
      if (out_summary) out_summary[i] = summarize(row, T, error);
    }
    free(scratch);
  }
  return 0;
}

static void ra_to_row(const ra_t* a, const kt_amounts* t, int64_t row, int D, bool* overflow);

/* A scheduling pass over a queue of pending pods, one at a time and in order:
 *   KubeThrottler.PreFilter (plugin.go:148-215); on Success KubeThrottler.Reserve (plugin.go:217-239) ->
 *   [Cluster]ThrottleController.Reserve (throttle_controller.go:271-300, clusterthrottle_controller.go:300-329):
 *   for every affected throttle cache.addPod (reserved_resource_amounts.go:66-77) stores
 *   ResourceAmountOfPod(pod) under the pod's name; the throttle's reserved total is the fold of Add over that
 *   map (reserved_resource_amounts.go:148-156), which the NEXT pod's CheckThrottledFor reads.
 * The snapshot's thr_reserved rows are the totals before the pass.  out_status[i] / out_summary[i] are what
 * PreFilter returned for pod i at its turn; out_reserved (nullable, n_thr rows) the totals after the pass. */
int kto_admit(kto_ctx* c, int64_t n, const int64_t* rows, int on_equal, uint8_t* out_status, uint64_t* out_summary,
              const kt_amounts* out_reserved) {
  const kt_snapshot* s = c->s;
  const int32_t T = s->n_thr;
  ra_t* res = (ra_t*)malloc(sizeof(ra_t) * (size_t)(T > 0 ? T : 1));
  uint8_t* scratch = (uint8_t*)malloc((size_t)(T > 0 ? T : 1));
  for (int32_t t = 0; t < T; ++t) res[t] = reserved_resource_amount(s, t);
  for (int64_t i = 0; i < n; ++i) {
    int64_t prow = rows ? rows[i] : i;
    uint8_t* row = out_status ? out_status + i * T : scratch;
    memset(row, KTO_NOT_AFFECTED, (size_t)T);
    bool error = false, valid = false;
    pod_t pod = {s, prow};
    if (prow >= 0 && prow < s->n_pods && (s->pod_flags[prow] & KT_POD_VALID)) {
      valid = true;
      if (!check_throttled(c, pod, false, on_equal != 0, row, 0, res)) error = true;
      else if (!check_throttled(c, pod, true, on_equal != 0, row, 0, res)) error = true;
    }
    if (error) memset(row, KTO_ERROR, (size_t)T);
    uint64_t sum = summarize(row, T, error);
    if (out_summary) out_summary[i] = sum;
    if (valid && !error && (sum & 3u) == KTO_VERDICT_ALLOW) {
      ra_t pa = resource_amount_of_pod(pod);
      for (int32_t t = 0; t < T; ++t)
        if (row[t] != KTO_NOT_AFFECTED) res[t] = ra_add(res[t], &pa);
    }
  }
  if (out_reserved) {
    bool ovf = false;
    for (int32_t t = 0; t < T; ++t) ra_to_row(&res[t], out_reserved, t, s->D, &ovf);
  }
  free(scratch);
  free(res);
  return 0;
}

/* ThrottleSpecBase.NextOverrideHappensIn — throttle_types.go:37-63: the earliest begin / end instant strictly after
 * `now` over all overrides, skipping what does not parse (a bad `begin` skips the override, a bad `end` only the end).
 * Returned as the instant itself (out_has = 0: none); the caller's enqueueAfter delay is (instant - now).
 * No reference test covers it: parity unpinned, restated from the code. */
int kto_next_override(kto_ctx* c, int32_t n, const int32_t* rows, int64_t now_s, int32_t now_ns, int64_t* out_s,
                      int32_t* out_ns, uint8_t* out_has) {
  const kt_snapshot* s = c->s;
  for (int32_t i = 0; i < n; ++i) {
    int32_t t = rows ? rows[i] : i;
    bool has = false;
    int64_t bs = 0;
    int32_t bn = 0;
    for (uint32_t o = s->thr_ovr_off[t]; o < s->thr_ovr_off[t + 1]; ++o) {
      const bool err = (s->ovr_flags[o] & KT_OVR_PARSE_ERROR) != 0;
      if (err && !(s->ovr_flags[o] & KT_OVR_BEGIN_PARSED)) continue; /* BeginTime() failed */
      if (instant_cmp(s->ovr_begin_s[o], s->ovr_begin_ns[o], now_s, now_ns) > 0 &&
          (!has || instant_cmp(s->ovr_begin_s[o], s->ovr_begin_ns[o], bs, bn) < 0))
        has = true, bs = s->ovr_begin_s[o], bn = s->ovr_begin_ns[o];
      if (err) continue; /* EndTime() failed */
      if (instant_cmp(s->ovr_end_s[o], s->ovr_end_ns[o], now_s, now_ns) > 0 &&
          (!has || instant_cmp(s->ovr_end_s[o], s->ovr_end_ns[o], bs, bn) < 0))
        has = true, bs = s->ovr_end_s[o], bn = s->ovr_end_ns[o];
    }
    out_has[i] = has;
    out_s[i] = has ? bs : 0;
    out_ns[i] = has ? bn : 0;
  }
  return 0;
}

int kto_pod_requests(kto_ctx* c, int64_t n, const int64_t* rows, int64_t* out_v, uint32_t* out_present) {
  const kt_snapshot* s = c->s;
  for (int64_t i = 0; i < n; ++i) {
    pod_t pod = {s, rows ? rows[i] : i};
    rl_t r = pod_request_resource_list(pod);
    uint32_t present = 0;
    for (int d = 0; d < s->D; ++d) out_v[i * s->D + d] = 0;
    for (int k = 0; k < r.n; ++k) {
      present |= 1u << r.name[k];
      out_v[i * s->D + r.name[k]] = (int64_t)r.q[k];
    }
    out_present[i] = present;
  }
  return 0;
}

static void ra_to_row(const ra_t* a, const kt_amounts* t, int64_t row, int D, bool* overflow) {
  t->has_count[row] = a->counts_nonnil;
  t->count[row] = a->counts_nonnil ? a->pod : 0;
  uint32_t present = 0;
  for (int d = 0; d < D; ++d) t->v[row * D + d] = 0;
  for (int k = 0; k < a->req.n; ++k) {
    present |= 1u << a->req.name[k];
    q_t q = a->req.q[k];
    if (q > (q_t)INT64_MAX || q < (q_t)INT64_MIN) *overflow = true;
    t->v[row * D + a->req.name[k]] = (int64_t)q;
  }
  t->present[row] = present;
}

/* [Cluster]ThrottleController.reconcile, aggregation part — throttle_controller.go:103-133,
 * clusterthrottle_controller.go:106-136 (affectedPods :221-246 / :224-270). */
static bool reconcile_one(kto_ctx* c, int32_t t, int64_t now_s, int32_t now_ns, int32_t i, kto_reconcile_out* out) {
  const kt_snapshot* s = c->s;
  const int D = s->D;
  const bool cluster = (s->thr_flags[t] & KT_THR_CLUSTER) != 0;
  ra_t used;
  ra_zero(&used);
  bool err = false;
  if (!cluster) {
    uint32_t ns = s->thr_ns[t];
    if (ns < (uint32_t)s->n_ns)
      for (uint64_t k = c->ns_pod_off[ns]; k < c->ns_pod_off[ns + 1]; ++k) {
        int64_t p = c->ns_pod[k];
        if (!should_count_in(s, p)) continue;
        pod_t pod = {s, p};
        bool match = throttle_selector_matches_to_pod(s, t, pod, &err);
        if (err) return false;
        if (match && is_not_finished(s, p)) {
          ra_t pa = resource_amount_of_pod(pod);
          used = ra_add(used, &pa);
        }
      }
  } else {
    for (int32_t ns = 0; ns < s->n_ns; ++ns) {
      if (!s->ns_valid[ns]) continue;
      if (!cluster_selector_matches_to_namespace(s, t, (uint32_t)ns, c->ns_memo)) continue;
      for (uint64_t k = c->ns_pod_off[ns]; k < c->ns_pod_off[ns + 1]; ++k) {
        int64_t p = c->ns_pod[k];
        if (!should_count_in(s, p)) continue;
        pod_t pod = {s, p};
        bool match = cluster_selector_matches_to_pod(s, t, pod, (uint32_t)ns, &err, c->ns_memo);
        if (err) return false;
        if (!match) continue;
        if (is_not_finished(s, p)) {
          ra_t pa = resource_amount_of_pod(pod);
          used = ra_add(used, &pa);
        }
      }
    }
  }
  bool any_err = false;
  ra_t calculated = calculate_threshold(s, t, now_s, now_ns, &any_err);
  uint64_t calc_msgs_fp = any_err ? s->thr_spec_msgs_fp[t] : 0;
  ra_t stored;
  ra_from_row(&stored, &s->thr_calc, t, D);
  bool replace = !ra_semantic_equal(&stored, &calculated) || s->thr_status_msgs_fp[t] != calc_msgs_fp;
  ra_t new_calc = replace ? calculated : stored;
  irat_t thrl = ra_is_throttled(&new_calc, &used, true);
  bool overflow = false;
  ra_to_row(&used, &out->used, i, D, &overflow);
  if (c->out_used_hi) {  /* the caller takes `used` as 128-bit values: low words in out->used.v, high words here */
    overflow = false;
    for (int d = 0; d < D; ++d) c->out_used_hi[(int64_t)i * D + d] = 0;
    for (int k = 0; k < used.req.n; ++k) c->out_used_hi[(int64_t)i * D + used.req.name[k]] = (int64_t)(used.req.q[k] >> 64);
  }
  ra_to_row(&new_calc, &out->calc, i, D, &overflow);
  out->calc_updated[i] = replace;
  uint32_t has = 0, flag = 0;
  for (int k = 0; k < thrl.n; ++k) {
    has |= 1u << thrl.name[k];
    if (thrl.val[k]) flag |= 1u << thrl.name[k];
  }
  out->thrl_has[i] = has;
  out->thrl_flag[i] = flag;
  out->thrl_pod[i] = thrl.pod;
  return !overflow;
}

int kto_reconcile(kto_ctx* c, int32_t n, const int32_t* rows, int64_t now_s, int32_t now_ns,
                  kto_reconcile_out* out, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (int32_t i = 0; i < n; ++i) {
    int32_t t = rows ? rows[i] : i;
    out->error[i] = 0;
    if (t < 0 || t >= c->s->n_thr || !(c->s->thr_flags[t] & KT_THR_VALID)) {
      out->error[i] = 2;
      continue;
    }
    if (!reconcile_one(c, t, now_s, now_ns, i, out)) out->error[i] = 1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* function-level entry points, so the reference's unit-test tables can be replayed verbatim   */
/* ------------------------------------------------------------------------------------------ */
int kto_unit_is_throttled(int D, const kt_amounts* threshold, const kt_amounts* used, int on_equal,
                          uint32_t* out_flag, uint32_t* out_has, uint8_t* out_pod) {
  ra_t thr, u;
  ra_from_row(&thr, threshold, 0, D);
  ra_from_row(&u, used, 0, D);
  irat_t f = ra_is_throttled(&thr, &u, on_equal != 0);
  *out_flag = 0;
  *out_has = 0;
  for (int k = 0; k < f.n; ++k) {
    *out_has |= 1u << f.name[k];
    if (f.val[k]) *out_flag |= 1u << f.name[k];
  }
  *out_pod = f.pod;
  return 0;
}

int kto_unit_is_throttled_for(const kt_snapshot* s, int64_t pod_row, uint32_t flag, uint32_t has, int pod_flag) {
  irat_t f;
  f.pod = pod_flag != 0;
  f.n = 0;
  for (int d = 0; d < s->D; ++d)
    if (has >> d & 1u) {
      f.name[f.n] = d;
      f.val[f.n] = (flag >> d & 1u) != 0;
      f.n++;
    }
  pod_t p = {s, pod_row};
  return irat_is_throttled_for(&f, p) ? 1 : 0;
}

int kto_unit_override_is_active(const kt_snapshot* s, uint32_t o, int64_t now_s, int32_t now_ns) {
  bool err = false;
  bool a = override_is_active(s, o, now_s, now_ns, &err);
  return err ? -1 : (a ? 1 : 0);
}

int kto_unit_calculate_threshold(const kt_snapshot* s, int32_t t, int64_t now_s, int32_t now_ns,
                                 const kt_amounts* out, uint8_t* out_any_err) {
  bool any_err = false, overflow = false;
  ra_t c = calculate_threshold(s, t, now_s, now_ns, &any_err);
  ra_to_row(&c, out, 0, s->D, &overflow);
  *out_any_err = any_err;
  return overflow ? -1 : 0;
}

/* [Cluster]ThrottleSelector.MatchesToPod(pod[, namespace of the pod]) -> 1 / 0 / -1 (error). */
int kto_unit_selector_matches(const kt_snapshot* s, int32_t t, int64_t pod_row) {
  bool err = false;
  pod_t p = {s, pod_row};
  bool m = (s->thr_flags[t] & KT_THR_CLUSTER) ? cluster_selector_matches_to_pod(s, t, p, s->pod_ns[pod_row], &err, NULL)
                                              : throttle_selector_matches_to_pod(s, t, p, &err);
  return err ? -1 : (m ? 1 : 0);
}
