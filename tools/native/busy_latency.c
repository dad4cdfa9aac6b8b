/* busy_latency.c — kt_check(n = 1) timed from a native thread, alone and beside a native thread that reconciles in a
 * loop: what a cgo caller sees (tools/latency_bench.py's Python loops share the interpreter lock between the timing
 * thread and the reconciling thread: whenever the reconciler is between two foreign calls, the timer waits for it).
 *   gcc -O2 -shared -fPIC busy_latency.c -L../../kube_throttler_amd/csrc -lkt_engine -lpthread -o _busy_latency.so */
#include <pthread.h>
#include <stdint.h>
#include <time.h>
#include "kt_engine.h"

typedef struct {
  kt_engine* e;
  volatile int stop;
  int64_t now_s;
  int64_t n_rec;
  int32_t rc;
} reconciler_t;

static void* reconcile_loop(void* p) {
  reconciler_t* r = (reconciler_t*)p;
  while (!r->stop) {
    int32_t rc = kt_reconcile_launch(r->e, r->now_s, 0, KT_RECONCILE_APPLY, NULL);
    if (rc == 0) rc = kt_synchronize(r->e, NULL);
    if (rc != 0) {
      r->rc = rc;
      break;
    }
    ++r->n_rec;
  }
  return NULL;
}

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

/* out_us[i] = duration of kt_check(n = 1) for pod row rows[i]; returns the reconciles that ran meanwhile (with_reconciler)
 * or 0, negative = an engine error code */
int64_t kt_native_check_latency(kt_engine* e, int64_t now_s, int64_t n, const int64_t* rows, double* out_us, int32_t with_reconciler) {
  reconciler_t r = {e, 0, now_s, 0, 0};
  pthread_t th;
  if (with_reconciler && pthread_create(&th, NULL, reconcile_loop, &r) != 0) return -1000;
  int32_t rc = 0;
  for (int64_t i = 0; i < n && rc == 0; ++i) {
    uint64_t summary = 0;
    const double t0 = now_us();
    rc = kt_check(e, 1, rows + i, 0, &summary, NULL);
    out_us[i] = now_us() - t0;
  }
  if (with_reconciler) {
    r.stop = 1;
    pthread_join(th, NULL);
  }
  if (rc != 0) return rc < 0 ? rc : -rc;
  if (r.rc != 0) return r.rc < 0 ? r.rc : -r.rc;
  return r.n_rec;
}
