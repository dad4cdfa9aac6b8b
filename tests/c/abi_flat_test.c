/*
 * abi_flat_test.c — a plain C99 host (no C++, no Python, no framework in the process) drives libkt_engine.so the way the
 * Go plugin's informer handlers would: one object per call through the flat single-object entry points
 * (kt_upsert_namespace / kt_upsert_throttle / kt_upsert_pod: every pointer a direct argument to pointer-free memory),
 * then the multi-GPU reconcile sequence with the native RCCL exchange (kt_aggregate_launch -> kt_comm_allreduce_partial
 * -> kt_finalize_launch; world = 1 here), then PreFilter for every pod (kt_check).  Input: a scenario file written by
 * tests/test_abi_flat_gpu.py (arrays of a kt_snapshot, each as u32 byte length + bytes); output: the results as arrays
 * in the same framing, which the test compares with the CPU oracle.
 *   usage: abi_flat_test <scenario.bin> <results.bin> [comm]     (comm: reconcile through kt_comm_*; RCCL start-up
 *                                                                  takes tens of seconds, so it is opt-in)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kt_engine.h"

static void* rd(FILE* f, size_t* n_bytes) {
  uint32_t n = 0;
  void* p;
  if (fread(&n, 4, 1, f) != 1) { fprintf(stderr, "short scenario file\n"); exit(2); }
  p = malloc(n ? n : 1);
  if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short scenario file\n"); exit(2); }
  if (n_bytes) *n_bytes = n;
  return p;
}
static void wr(FILE* f, const void* p, size_t n_bytes) {
  uint32_t n = (uint32_t)n_bytes;
  fwrite(&n, 4, 1, f);
  if (n) fwrite(p, 1, n, f);
}
#define CK(call)                                                                          \
  do {                                                                                    \
    int32_t rc_ = (call);                                                                 \
    if (rc_ != KT_OK) {                                                                   \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, kt_last_error(e));               \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

int main(int argc, char** argv) {
  FILE* f;
  kt_engine* e = NULL;
  kt_config cfg;
  int32_t* hdr;
  int32_t D, n_ns, n_thr, i;
  int64_t n_pods, now_s, p;
  if (argc < 3) { fprintf(stderr, "usage: %s scenario.bin results.bin\n", argv[0]); return 2; }
  f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  hdr = (int32_t*)rd(f, NULL); /* D, L, n_ns, n_thr, n_pods, now_s(lo), now_s(hi) */
  D = hdr[0], n_ns = hdr[2], n_thr = hdr[3], n_pods = hdr[4];
  now_s = (int64_t)(uint32_t)hdr[5] | ((int64_t)hdr[6] << 32);
  {
    /* ---- the snapshot's arrays, in the order the writer emits them */
    uint8_t* ns_valid = (uint8_t*)rd(f, NULL);
    uint32_t* ns_label_off = (uint32_t*)rd(f, NULL);
    uint32_t* ns_label_key = (uint32_t*)rd(f, NULL);
    uint32_t* ns_label_pair = (uint32_t*)rd(f, NULL);
    uint32_t* pod_ns = (uint32_t*)rd(f, NULL);
    uint32_t* pod_flags = (uint32_t*)rd(f, NULL);
    uint32_t* pod_label_off = (uint32_t*)rd(f, NULL);
    uint32_t* pod_label_key = (uint32_t*)rd(f, NULL);
    uint32_t* pod_label_pair = (uint32_t*)rd(f, NULL);
    uint32_t* pod_ctr_off = (uint32_t*)rd(f, NULL);
    uint8_t* ctr_init = (uint8_t*)rd(f, NULL);
    uint32_t* ctr_present = (uint32_t*)rd(f, NULL);
    int64_t* ctr_req = (int64_t*)rd(f, NULL);
    uint32_t* pod_ovh_present = (uint32_t*)rd(f, NULL);
    int64_t* pod_ovh = (int64_t*)rd(f, NULL);
    uint32_t* thr_flags = (uint32_t*)rd(f, NULL);
    uint32_t* thr_ns = (uint32_t*)rd(f, NULL);
    int64_t* amt_v[4];
    uint32_t* amt_p[4];
    int64_t* amt_c[4];
    uint8_t* amt_h[4];
    uint32_t *thrl_flag, *thrl_has, *thr_ovr_off, *ovr_present, *thr_term_off, *term_preq_off, *term_nreq_off;
    uint64_t *status_fp, *spec_fp;
    int64_t *ovr_begin_s, *ovr_end_s, *ovr_v, *ovr_count;
    int32_t *ovr_begin_ns, *ovr_end_ns;
    uint8_t *ovr_flags, *ovr_hc, *term_flags;
    uint8_t* rq_op[2];
    uint32_t* rq_key[2];
    uint32_t* rq_val_off[2];
    uint32_t* rq_val[2];
    int k;
    for (k = 0; k < 4; ++k) { /* spec, calc, used, reserved */
      amt_v[k] = (int64_t*)rd(f, NULL), amt_p[k] = (uint32_t*)rd(f, NULL), amt_c[k] = (int64_t*)rd(f, NULL), amt_h[k] = (uint8_t*)rd(f, NULL);
    }
    thrl_flag = (uint32_t*)rd(f, NULL), thrl_has = (uint32_t*)rd(f, NULL);
    status_fp = (uint64_t*)rd(f, NULL), spec_fp = (uint64_t*)rd(f, NULL);
    thr_ovr_off = (uint32_t*)rd(f, NULL);
    ovr_begin_s = (int64_t*)rd(f, NULL), ovr_begin_ns = (int32_t*)rd(f, NULL), ovr_end_s = (int64_t*)rd(f, NULL), ovr_end_ns = (int32_t*)rd(f, NULL);
    ovr_flags = (uint8_t*)rd(f, NULL);
    ovr_v = (int64_t*)rd(f, NULL), ovr_present = (uint32_t*)rd(f, NULL), ovr_count = (int64_t*)rd(f, NULL), ovr_hc = (uint8_t*)rd(f, NULL);
    thr_term_off = (uint32_t*)rd(f, NULL);
    term_flags = (uint8_t*)rd(f, NULL), term_preq_off = (uint32_t*)rd(f, NULL), term_nreq_off = (uint32_t*)rd(f, NULL);
    for (k = 0; k < 2; ++k) { /* preq, nreq */
      rq_op[k] = (uint8_t*)rd(f, NULL), rq_key[k] = (uint32_t*)rd(f, NULL), rq_val_off[k] = (uint32_t*)rd(f, NULL), rq_val[k] = (uint32_t*)rd(f, NULL);
    }
    fclose(f);

    memset(&cfg, 0, sizeof cfg);
    cfg.n_dims = D, cfg.max_labels = hdr[1] > 0 ? hdr[1] : 1;
    cfg.pod_capacity = n_pods > 0 ? n_pods : 1, cfg.throttle_capacity = n_thr > 0 ? n_thr : 1, cfg.namespace_capacity = n_ns > 0 ? n_ns : 1;
    cfg.device = -1, cfg.kernel_variant = 0;
    if (kt_engine_create(&cfg, &e) != KT_OK) { fprintf(stderr, "kt_engine_create: %s\n", kt_last_error(NULL)); return 1; }

    /* ---- informer handlers: one object per call */
    for (i = 0; i < n_ns; ++i)
      CK(kt_upsert_namespace(e, i, ns_valid[i], (int32_t)(ns_label_off[i + 1] - ns_label_off[i]), ns_label_key + ns_label_off[i],
                             ns_label_pair + ns_label_off[i]));
    for (i = 0; i < n_thr; ++i) {
      int64_t av[4 * KT_MAX_DIMS];
      uint32_t ap[4];
      int64_t ac[4];
      uint8_t ah[4];
      const uint32_t o0 = thr_ovr_off[i], o1 = thr_ovr_off[i + 1], t0 = thr_term_off[i], t1 = thr_term_off[i + 1];
      const uint32_t nt = t1 - t0;
      uint32_t* poff = (uint32_t*)malloc((nt + 1) * 4);
      uint32_t* noff = (uint32_t*)malloc((nt + 1) * 4);
      uint32_t r0[2], r1[2], q;
      uint32_t* voff[2];
      r0[0] = term_preq_off[t0], r1[0] = term_preq_off[t1], r0[1] = term_nreq_off[t0], r1[1] = term_nreq_off[t1];
      for (q = 0; q <= nt; ++q) poff[q] = term_preq_off[t0 + q] - r0[0], noff[q] = term_nreq_off[t0 + q] - r0[1];
      for (k = 0; k < 2; ++k) { /* value offsets of this throttle's requirements, rebased to its first value */
        const uint32_t nr = r1[k] - r0[k];
        voff[k] = (uint32_t*)malloc((nr + 1) * 4);
        for (q = 0; q <= nr; ++q) voff[k][q] = rq_val_off[k][r0[k] + q] - rq_val_off[k][r0[k]];
      }
      for (k = 0; k < 4; ++k) {
        memcpy(av + (size_t)k * D, amt_v[k] + (size_t)i * D, (size_t)D * 8);
        ap[k] = amt_p[k][i], ac[k] = amt_c[k][i], ah[k] = amt_h[k][i];
      }
      if (i == 0) { /* shapes that do not fit each other are refused (a slice in the wrong position must not be a silent mis-feed) */
        const uint8_t one_flag[1] = {0}, bad_op[1] = {9}, in_op[1] = {KT_OP_IN};
        const uint32_t one_term[2] = {0, 1}, no_req[2] = {0, 0}, key1[1] = {1}, dec_off[2] = {2, 1}, ok_off[2] = {0, 1}, val1[1] = {7};
        const uint32_t late_start[2] = {1, 1};
        int32_t rc;
#define KT_EXPECT_INVALID(call)                                                              \
  rc = (call);                                                                               \
  if (rc != KT_ERR_INVALID_ARGUMENT) {                                                       \
    fprintf(stderr, "abi_flat_test: %s -> %d, expected KT_ERR_INVALID_ARGUMENT\n", #call, rc); \
    return 4;                                                                                \
  }
        KT_EXPECT_INVALID(kt_upsert_throttle(e, i, thr_flags[i], thr_ns[i], av, ap, ac, ah, 0, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                                             NULL, 1, one_flag, one_term, no_req, 1, in_op, key1, dec_off, val1, 0, NULL, NULL, NULL, NULL));
        KT_EXPECT_INVALID(kt_upsert_throttle(e, i, thr_flags[i], thr_ns[i], av, ap, ac, ah, 0, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                                             NULL, 1, one_flag, one_term, no_req, 1, bad_op, key1, ok_off, val1, 0, NULL, NULL, NULL, NULL));
        KT_EXPECT_INVALID(kt_upsert_throttle(e, i, thr_flags[i], thr_ns[i], av, ap, ac, ah, 0, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                                             NULL, 1, one_flag, late_start, no_req, 1, in_op, key1, ok_off, val1, 0, NULL, NULL, NULL, NULL));
        KT_EXPECT_INVALID(kt_upsert_throttle(e, i, thr_flags[i], thr_ns[i], av, ap, ac, ah, 0xFFFF0000u, 0xFFFF0000u, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                                             NULL, NULL, NULL, 1, one_flag, one_term, no_req, 1, in_op, key1, ok_off, val1, 0, NULL, NULL, NULL, NULL));
#undef KT_EXPECT_INVALID
      }
      CK(kt_upsert_throttle(e, i, thr_flags[i], thr_ns[i], av, ap, ac, ah, thrl_flag[i], thrl_has[i], status_fp[i], spec_fp[i], (int32_t)(o1 - o0),
                            ovr_begin_s + o0, ovr_begin_ns + o0, ovr_end_s + o0, ovr_end_ns + o0, ovr_flags + o0, ovr_v + (size_t)o0 * D,
                            ovr_present + o0, ovr_count + o0, ovr_hc + o0, (int32_t)nt, term_flags + t0, poff, noff, r1[0] - r0[0],
                            rq_op[0] + r0[0], rq_key[0] + r0[0], voff[0], rq_val[0] + rq_val_off[0][r0[0]], r1[1] - r0[1], rq_op[1] + r0[1],
                            rq_key[1] + r0[1], voff[1], rq_val[1] + rq_val_off[1][r0[1]]));
      free(poff), free(noff), free(voff[0]), free(voff[1]);
    }
    for (p = 0; p < n_pods; ++p) {
      const uint32_t l0 = pod_label_off[p], l1 = pod_label_off[p + 1], c0 = pod_ctr_off[p], c1 = pod_ctr_off[p + 1];
      CK(kt_upsert_pod(e, p, pod_ns[p], pod_flags[p], (int32_t)(l1 - l0), pod_label_key + l0, pod_label_pair + l0, (int32_t)(c1 - c0),
                       ctr_init + c0, ctr_present + c0, ctr_req + (size_t)c0 * D, pod_ovh_present[p], pod_ovh + (size_t)p * D));
    }
  }
  {
    /* ---- reconcile the multi-GPU way (world = 1): scan -> RCCL all-reduce of the partials -> finalize, status applied */
    unsigned char id[KT_COMM_ID_BYTES];
    kt_status st;
    int32_t T = 0;
    size_t nt;
    FILE* o;
    uint64_t* summary;
    uint8_t* status;
    const int with_comm = argc > 3 && strcmp(argv[3], "comm") == 0;
    if (with_comm) {
      CK(kt_comm_unique_id(id));
      CK(kt_comm_init(e, 0, 1, id));
      CK(kt_aggregate_launch(e, NULL));
      CK(kt_comm_allreduce_partial(e, NULL));
      CK(kt_finalize_launch(e, now_s, 0, KT_RECONCILE_APPLY, NULL));
    } else {
      CK(kt_reconcile_launch(e, now_s, 0, KT_RECONCILE_APPLY, NULL));
    }
    CK(kt_throttle_rows(e, &T));
    nt = (size_t)(T > 0 ? T : 1);
    memset(&st, 0, sizeof st);
    st.used.v = (int64_t*)calloc(nt * D, 8), st.used.present = (uint32_t*)calloc(nt, 4), st.used.count = (int64_t*)calloc(nt, 8);
    st.used.has_count = (uint8_t*)calloc(nt, 1);
    st.calc.v = (int64_t*)calloc(nt * D, 8), st.calc.present = (uint32_t*)calloc(nt, 4), st.calc.count = (int64_t*)calloc(nt, 8);
    st.calc.has_count = (uint8_t*)calloc(nt, 1);
    st.calc_at_nonzero = (uint8_t*)calloc(nt, 1), st.thrl_flag = (uint32_t*)calloc(nt, 4), st.thrl_has = (uint32_t*)calloc(nt, 4);
    st.thrl_pod = (uint8_t*)calloc(nt, 1), st.error = (uint8_t*)calloc(nt, 1);
    CK(kt_reconcile_fetch(e, T, &st));
    /* ---- PreFilter for every pod, full status rows */
    summary = (uint64_t*)calloc((size_t)(n_pods > 0 ? n_pods : 1), 8);
    status = (uint8_t*)calloc((size_t)(n_pods > 0 ? n_pods : 1) * nt, 1);
    CK(kt_check(e, n_pods, NULL, 0, summary, status));
    if (with_comm) CK(kt_comm_destroy(e));
    o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    wr(o, &T, 4);
    wr(o, st.used.v, (size_t)T * D * 8), wr(o, st.used.present, (size_t)T * 4), wr(o, st.used.count, (size_t)T * 8), wr(o, st.used.has_count, (size_t)T);
    wr(o, st.calc.v, (size_t)T * D * 8), wr(o, st.calc.present, (size_t)T * 4), wr(o, st.calc.count, (size_t)T * 8), wr(o, st.calc.has_count, (size_t)T);
    wr(o, st.calc_at_nonzero, (size_t)T), wr(o, st.thrl_flag, (size_t)T * 4), wr(o, st.thrl_has, (size_t)T * 4), wr(o, st.thrl_pod, (size_t)T);
    wr(o, st.error, (size_t)T);
    wr(o, summary, (size_t)n_pods * 8), wr(o, status, (size_t)n_pods * (size_t)T);
    fclose(o);
    printf("abi_flat_test: %d namespaces, %d throttles, %lld pods through the single-object C-ABI; %s\n", (int)n_ns, (int)n_thr,
           (long long)n_pods, kt_version());
  }
  CK(kt_engine_destroy(e));
  return 0;
}
