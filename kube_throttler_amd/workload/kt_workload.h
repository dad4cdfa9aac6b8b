/*
 * kt_workload.h — deterministic synthetic cluster snapshots for parity tests and bench.py.
 *
 * The reference ships no benchmark and its largest test is 50 ClusterThrottles x 100 pods
 * (test/integration/clusterthrottle_stress_test.go:33-35), so the BASELINE.json configs are
 * synthesised here, once, on the host, by a fixed PRNG (splitmix64, seed 0x6B7468726F74 + config
 * index) and fed identically to the engine and to the CPU oracle.
 * Every entity (pod, throttle, namespace) draws from its own stream keyed by its index, so a pod
 * shard [pod_begin, pod_begin+n) is bit-identical to the same rows of the full snapshot — ranks of a
 * multi-GPU run generate only their shard.
 */
#ifndef KT_WORKLOAD_H
#define KT_WORKLOAD_H

#include <stdint.h>
#include "../../include/kt_snapshot.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kt_workload_cfg {
  uint64_t seed;       /* 0x6B7468726F74 + config index */
  int64_t n_pods_total;/* P of the whole job (used for threshold calibration) */
  int64_t pod_begin;   /* first pod row generated */
  int64_t n_pods;      /* rows generated */
  int32_t n_thr;       /* throttles, both kinds */
  int32_t n_cluster;   /* how many of them are ClusterThrottles (placed after the Throttles) */
  int32_t D;           /* resource dimensions: cpu(milli), memory, ephemeral-storage, amd.com/gpu, ... */
  int32_t n_ns;
  int32_t K, V, L;     /* label keys, values per key, distinct keys per pod */
  int32_t terms_min, terms_max; /* selector terms per throttle */
  int32_t reqs_min, reqs_max;   /* requirements per term */
  int32_t rich_ops;    /* 0: matchLabels only; 1: In(1-3 values)/NotIn/Exists/DoesNotExist mix */
  int32_t overrides;   /* 0: none; 1: 2-3 temporaryThresholdOverrides per throttle around now_s */
  int64_t now_s;       /* the instant overrides are laid out around */
  int32_t n_invalid_pod_sel; /* throttles that get an unconvertible podSelector term (tests) */
  int32_t n_invalid_ns_sel;  /* cluster throttles that get an unconvertible namespaceSelector term (tests) */
  int32_t n_missing_ns;      /* namespaces without a Namespace object (tests) */
} kt_workload_cfg;

/* Fills cfg with BASELINE.json configs[index] (index 1..4; 4 = the 10M x 10k config, whole job). */
int kt_workload_preset(int index, kt_workload_cfg* cfg);

/* Allocates and fills a snapshot; free with kt_workload_free. */
kt_snapshot* kt_workload_generate(const kt_workload_cfg* cfg);
void kt_workload_free(kt_snapshot* s);

#ifdef __cplusplus
}
#endif
#endif
