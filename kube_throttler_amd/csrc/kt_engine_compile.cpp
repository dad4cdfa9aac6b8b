// kt_engine_compile.cpp — compile_program: the throttles and namespaces the engine holds -> the device selector program, the
// namespace side of every term, the chunked term-bitmap index (kt_index.cpp) and the pods' atom rows for it.
#include "kt_engine_impl.h"

// ---- host-side label selector evaluation (namespace selectors only; pods are matched on device) -----
static bool ns_selector_matches(const std::vector<Req>& reqs, const HostNamespace& n) {
  for (const Req& r : reqs) {
    bool has = false, in = false;
    for (auto& kv : n.labels) {
      if (kv.first == r.key) has = true;
      for (uint32_t v : r.vals) in |= kv.second == v;
    }
    bool ok;
    switch (r.op) {
      case KT_OP_IN: ok = in; break;
      case KT_OP_NOT_IN: ok = !in; break;
      case KT_OP_EXISTS: ok = has; break;
      case KT_OP_DOES_NOT_EXIST: ok = !has; break;
      default: ok = false;
    }
    if (!ok) return false;
  }
  return true;
}


int32_t compile_program(kt_engine* e, hipStream_t s) {
  static const bool dbg_time = getenv("KT_DEBUG_COMPILE") != nullptr;  // phase times of a recompile on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "compile_program: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const int D = e->D;
  const size_t T = (size_t)e->thr_rows_hi;
  // namespace rows the program covers: the rows in USE (namespace objects, pods, namespaced Throttles), not the
  // configured capacity — the per-namespace tables of the program and of every index chunk scale with this number.
  // A pod that later arrives with a higher namespace row marks the program dirty (upsert_pods_locked).
  size_t NS = std::max<size_t>(1, std::max<size_t>((size_t)e->ns_rows_hi, (size_t)e->pod_ns_hi));
  for (size_t t = 0; t < T; ++t)
    if ((e->thr[t].flags & KT_THR_VALID) && !(e->thr[t].flags & KT_THR_CLUSTER)) NS = std::max(NS, (size_t)e->thr[t].ns + 1);
  NS = std::min(NS, (size_t)e->cfg.namespace_capacity);
  e->ns_compiled = NS;
  std::vector<uint32_t> thr_term_off(T + 1, 0), term_thr, term_req_off{0}, req_key, req_val_off{0}, req_val;
  std::vector<uint8_t> term_flags, req_op;
  {
    size_t n_terms = 0, n_reqs = 0, n_vals = 0;
    for (size_t t = 0; t < T; ++t)
      if (e->thr[t].flags & KT_THR_VALID)
        for (const Term& tm : e->thr[t].terms) {
          ++n_terms, n_reqs += tm.preq.size();
          for (const Req& r : tm.preq) n_vals += r.vals.size();
        }
    term_thr.reserve(n_terms), term_flags.reserve(n_terms), term_req_off.reserve(n_terms + 1);
    req_op.reserve(n_reqs), req_key.reserve(n_reqs), req_val_off.reserve(n_reqs + 1), req_val.reserve(n_vals);
  }
  e->uses_keys = false;
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    if (h.flags & KT_THR_VALID)
      for (const Term& tm : h.terms) {
        term_thr.push_back((uint32_t)t);
        term_flags.push_back(tm.flags);
        for (const Req& r : tm.preq) {
          req_op.push_back(r.op);
          req_key.push_back(r.key);
          for (uint32_t v : r.vals) req_val.push_back(v);
          req_val_off.push_back((uint32_t)req_val.size());
          if (r.op == KT_OP_EXISTS || r.op == KT_OP_DOES_NOT_EXIST) e->uses_keys = true;
        }
        term_req_off.push_back((uint32_t)req_op.size());
      }
    thr_term_off[t + 1] = (uint32_t)term_thr.size();
  }
  lap("flatten throttles");
  const size_t G = term_thr.size();
  const uint32_t gw = (uint32_t)((G + 31) / 32 + 1);
  // ns x term applicability: the namespace side of every term, cached per throttle (HostThrottle::adm) and re-evaluated
  // only for throttles that changed since — or for all of them, on several host threads, after a namespace event
  const uint32_t nsw = (uint32_t)((NS + 31) / 32);
  std::vector<uint8_t> ns_valid(NS, 0);
  for (size_t n = 0; n < NS; ++n) ns_valid[n] = n < e->ns.size() && e->ns[n].valid;
  {
    std::vector<uint32_t> stale;
    for (size_t t = 0; t < T; ++t) {
      const HostThrottle& h = e->thr[t];
      if ((h.flags & KT_THR_VALID) && (h.adm_gen != e->ns_gen || h.adm_ns != (uint32_t)NS || h.adm.size() != h.terms.size() * nsw)) stale.push_back((uint32_t)t);
    }
    kt::parallel_for(stale.size(), 64, [&](size_t b0, size_t b1, size_t) {
      for (size_t q = b0; q < b1; ++q) {
        HostThrottle& h = e->thr[stale[q]];
        h.adm.assign(h.terms.size() * nsw, 0u);
        h.adm_gen = e->ns_gen, h.adm_ns = (uint32_t)NS;
        const uint32_t need = KT_THR_VALID | KT_THR_RESPONSIBLE;
        if ((h.flags & need) != need) continue;
        for (size_t k = 0; k < h.terms.size(); ++k) {
          const Term& tm = h.terms[k];
          uint32_t* row = h.adm.data() + k * nsw;
          if (!(h.flags & KT_THR_CLUSTER)) {
            // Throttles(pod.Namespace).List: implicit namespace equality, no Namespace object needed
            if (h.ns < NS) row[h.ns >> 5] |= 1u << (h.ns & 31);
          } else {
            if (tm.flags & KT_TERM_NS_SEL_INVALID) continue;  // swallowed to "no match" (clusterthrottle_selector.go:63-69)
            for (size_t n = 0; n < NS && n < e->ns.size(); ++n)
              if (e->ns[n].valid && ns_selector_matches(tm.nreq, e->ns[n])) row[n >> 5] |= 1u << (n & 31);
          }
        }
      }
    }, nullptr);
  }
  std::vector<uint32_t> adm_all(G * nsw, 0u);  // [term][namespace words]: what the index build wants
  for (size_t t = 0; t < T; ++t) {
    const HostThrottle& h = e->thr[t];
    if (!(h.flags & KT_THR_VALID) || h.terms.empty()) continue;
    memcpy(adm_all.data() + (size_t)thr_term_off[t] * nsw, h.adm.data(), h.adm.size() * 4);
  }
  std::vector<uint32_t> ns_term_ok;  // [namespace][term words]: what the kernels' rare paths and the dense variant read
  kt::transpose_term_ns_bits(adm_all, G, nsw, (uint32_t)NS, gw, ns_term_ok);
  lap("namespace side of the terms");
  int32_t rc;
#define UP(dev, host) if ((rc = upload(e, e->dev, host, s)) != KT_OK) return rc
  UP(d_thr_term_off, thr_term_off);
  UP(d_term_thr, term_thr);
  UP(d_term_flags, term_flags);
  UP(d_term_req_off, term_req_off);
  UP(d_req_op, req_op);
  UP(d_req_key, req_key);
  UP(d_req_val_off, req_val_off);
  UP(d_req_val, req_val);
  UP(d_ns_term_ok, ns_term_ok);
  UP(d_ns_valid, ns_valid);
#undef UP
  if ((rc = upload_spec_tables(e, s)) != KT_OK) return rc;
  // result / scratch buffers sized by T
  KT_HIP(e, e->d_partial.reserve(2 * T * kt::partial_stride(D) + 1));  // room for the two limb-sum blocks of a wide reconcile
  e->clean_partial = nullptr;
  KT_HIP(e, e->d_out_used.reserve(T + 1, D));
  KT_HIP(e, e->d_out_used_hi.reserve((T + 1) * (size_t)D));
  KT_HIP(e, e->d_out_calc.reserve(T + 1, D));
  KT_HIP(e, e->d_out_calc_updated.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_pod.reserve(T + 1));
  KT_HIP(e, e->d_out_error.reserve(T + 1));
  KT_HIP(e, e->d_out_next_s.reserve(T + 1));
  KT_HIP(e, e->d_out_next_ns.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_flag.reserve(T + 1));
  KT_HIP(e, e->d_out_thrl_has.reserve(T + 1));
  KT_HIP(e, e->d_recs2[0].reserve(kt::recs_bytes((int)T)));
  KT_HIP(e, e->d_recs2[1].reserve(kt::recs_bytes((int)T)));
  e->recs_prev_valid = false;
  lap("uploads + buffers");
  // index for the work ~ (pods + matches) kernels
  {
    auto thr_info = [&](uint32_t t) {
      const HostThrottle& h = e->thr[t];
      const uint32_t need = KT_THR_VALID | KT_THR_RESPONSIBLE;
      kt::ThrInfo ti;
      ti.live = (h.flags & need) == need;
      ti.cluster = (h.flags & KT_THR_CLUSTER) != 0;
      ti.ns = h.ns;
      return ti;
    };
    // KT_CHUNK_BUDGET (bytes): test hook that forces small chunks so that tiny programs exercise the multi-chunk path too
    const char* hook = getenv("KT_CHUNK_BUDGET");
    const uint32_t lds_all = 160u * 1024u;
    const uint32_t agg_budget = hook ? (uint32_t)atoi(hook) : lds_all - kt::aggregate_fixed_lds();
    const uint32_t thr_bytes = kt::agg_rec_bytes(D, e->incremental);
    // a program of several chunks is cut for the packed fold's records (at most 40 bytes, 72 beyond 8 dimensions: PackPlan) while this engine's scans
    // can pack — a chunk then holds more words, a namespace-ordered scan makes fewer chunk passes; the first scan that needs
    // the plain fold (a negative request, sums beyond int64, KT_NO_PACK) has the program cut again for plain records
    // (aggregate_locked: cut_plain)
    // (scans that gather through the row lists — KT_NO_SCAN_VIEW, KT_NO_NS_ORDER — fold plain records)
    const uint32_t thr_packed = (!e->incremental && !e->wide && !e->neg_seen && !e->cut_plain && !e->sw[kSw_NO_PACK] && !e->sw[kSw_NO_SCAN_VIEW] &&
                                 !e->sw[kSw_NO_NS_ORDER]) ? kt::packed_rec_max(D) : 0u;
    // the check kernel runs two workgroups per CU when the whole program fits half the LDS; otherwise the chunks are cut
    // for one workgroup per CU (fewer, larger chunks)
    const uint32_t chk_half = hook ? (uint32_t)atoi(hook) : lds_all / 2 - kt::check_fixed_lds();
    // KT_CHUNK_HALF=1 (A/B runs): keep the half-LDS chunks — more of them, but two workgroups per CU
    const bool full_when_chunked = !hook && !getenv("KT_CHUNK_HALF");
    kt::build_index(e->hindex, thr_term_off, term_thr, term_flags, term_req_off, req_op, req_key, req_val_off, req_val, thr_info,
                    (uint32_t)NS, ns_term_ok, gw, agg_budget, chk_half, thr_bytes, e->L, &adm_all,
                    full_when_chunked ? lds_all - kt::check_fixed_lds() : 0u, kt::check_word_lds(D), nullptr, thr_packed);
    lap("build_index");
    // a program that fits half the LDS as rows but still came out in several chunks (per-term tables): larger chunks
    if (full_when_chunked && e->hindex.bm_chunks.size() > 1 && e->hindex.cut_chk_budget != lds_all - kt::check_fixed_lds())
      kt::cut_chunks(e->hindex, agg_budget, lds_all - kt::check_fixed_lds(), thr_bytes, kt::check_word_lds(D), thr_packed);
    lap("cut_chunks (full LDS)");
  }
  kt::index_group_counts(e->hindex, (uint32_t)T);
  e->ctr_index_chunks.store((int64_t)e->hindex.bm_chunks.size(), std::memory_order_relaxed);
  e->ctr_index_words.store((int64_t)e->hindex.bm_words, std::memory_order_relaxed);
  e->ctr_index_image_words.store((int64_t)e->hindex.img_words, std::memory_order_relaxed);
  e->ctr_ns_rows.store((int64_t)e->hindex.n_ns, std::memory_order_relaxed);
  e->ctr_ns_word_visits.store(e->hindex.ns_word_visits, std::memory_order_relaxed);
  e->ctr_ns_chunk_visits.store(e->hindex.ns_chunk_visits, std::memory_order_relaxed);
  e->ctr_slow_throttles.store((int64_t)e->hindex.slow_thr.size(), std::memory_order_relaxed);
  e->agg_valid = false;  // a new selector program: the maintained partials are void
  KT_HIP(e, e->d_slab.reserve((size_t)e->hindex.bm_slab_bytes + 64));
  {
    hipError_t he = kt::upload_index(e->hindex, e->dindex, s);
    if (he != hipSuccess) return e->fail(KT_ERR_DEVICE, "upload_index: %s", hipGetErrorString(he));
  }
  lap("upload_index");
  // the pods' labels as atom ids of THIS program (labels no selector mentions drop out here)
  e->pods.LA = (int32_t)e->hindex.la;
  KT_HIP(e, e->d_latom.reserve((size_t)e->cfg.pod_capacity * (size_t)e->pods.LA + 64));
  e->pods.latom = e->d_latom.p;
  KT_HIP(e, e->d_overflow.reserve(1));
  KT_HIP(e, hipMemsetAsync(e->d_overflow.p, 0, 8, s));
  kt::launch_translate_pods(e->pods, e->pod_rows_hi, nullptr, 0, e->dindex, e->d_overflow.p, s);
  KT_HIP(e, hipGetLastError());
  e->countable_valid = false, e->order_all_valid = false;  // the scan views hold copies of the atom rows
  KT_HIP(e, hipMemcpyAsync(&e->n_overflow, e->d_overflow.p, 8, hipMemcpyDeviceToHost, s));
  KT_HIP(e, hipStreamSynchronize(s));  // host vectors go out of scope
  e->sp.thr_term_off = e->d_thr_term_off.p;
  e->sp.term_thr = e->d_term_thr.p;
  e->sp.term_flags = e->d_term_flags.p;
  e->sp.term_req_off = e->d_term_req_off.p;
  e->sp.req_op = e->d_req_op.p;
  e->sp.req_key = e->d_req_key.p;
  e->sp.req_val_off = e->d_req_val_off.p;
  e->sp.req_val = e->d_req_val.p;
  e->sp.ns_term_ok = e->d_ns_term_ok.p;
  e->sp.ns_valid = e->d_ns_valid.p;
  e->sp.gw = gw;
  e->sp.T = (int32_t)T;
  e->sp.G = (int32_t)G;
  e->sp.n_ns = (int32_t)NS;
  KT_HIP(e, e->d_sp.reserve(1));
  KT_HIP(e, hipMemcpyAsync(e->d_sp.p, &e->sp, sizeof(kt::SelProgram), hipMemcpyHostToDevice, s));
  KT_HIP(e, hipStreamSynchronize(s));
  lap("translate pods + sync");
  e->program_dirty = false;
  ++e->program_gen;
  e->n_compiles.fetch_add(1, std::memory_order_relaxed);
  return KT_OK;
}

