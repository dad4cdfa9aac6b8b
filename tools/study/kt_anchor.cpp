// kt_anchor.cpp — see kt_anchor.h
#include "kt_anchor.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include <algorithm>
#include <map>

#include "../../include/kt_snapshot.h"

namespace kt {

namespace {

struct TermAnchor {
  bool dead = false;              // an `In` without values: the term never matches
  bool has = false;               // has an `In` with values
  uint32_t key = 0;               // key of the anchor requirement
  std::vector<uint32_t> vals;     // its distinct values, sorted
};

void emit_term(AnchorSubProgram& sp, uint32_t g, const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
               const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key, const std::vector<uint32_t>& req_val_off,
               const std::vector<uint32_t>& req_val, const std::vector<std::pair<uint32_t, uint32_t>>& veto /* (key, pair), sorted */) {
  sp.term_orig.push_back(g);
  sp.term_flags.push_back(term_flags[g]);
  for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
    sp.req_op.push_back(req_op[r]);
    sp.req_key.push_back(req_key[r]);
    for (uint32_t q = req_val_off[r]; q < req_val_off[r + 1]; ++q) sp.req_val.push_back(req_val[q]);
    sp.req_val_off.push_back((uint32_t)sp.req_val.size());
  }
  // the earlier anchors, one NotIn per key (a requirement names pairs of ONE key)
  for (size_t i = 0; i < veto.size();) {
    size_t j = i;
    sp.req_op.push_back((uint8_t)KT_OP_NOT_IN);
    sp.req_key.push_back(veto[i].first);
    while (j < veto.size() && veto[j].first == veto[i].first) sp.req_val.push_back(veto[j++].second);
    sp.req_val_off.push_back((uint32_t)sp.req_val.size());
    i = j;
  }
  sp.term_req_off.push_back((uint32_t)sp.req_op.size());
}

AnchorSubProgram& sub_of(std::map<uint32_t, AnchorSubProgram>& subs, uint32_t anchor) {
  auto it = subs.find(anchor);
  if (it != subs.end()) return it->second;
  AnchorSubProgram& sp = subs[anchor];
  sp.anchor = anchor;
  sp.thr_term_off.push_back(0u);
  sp.term_req_off.push_back(0u);
  sp.req_val_off.push_back(0u);
  return sp;
}

}  // namespace

std::vector<AnchorSubProgram> anchor_split(const std::vector<uint32_t>& thr_term_off, const std::vector<uint8_t>& term_flags,
                                           const std::vector<uint32_t>& term_req_off, const std::vector<uint8_t>& req_op,
                                           const std::vector<uint32_t>& req_key, const std::vector<uint32_t>& req_val_off,
                                           const std::vector<uint32_t>& req_val, const std::vector<uint8_t>& thr_anchorable,
                                           AnchorSplitStats* stats, const std::vector<uint8_t>* term_kept) {
  const size_t T = thr_term_off.empty() ? 0 : thr_term_off.size() - 1;
  std::map<uint32_t, AnchorSubProgram> subs;
  AnchorSplitStats st;
  std::vector<TermAnchor> ta;
  const std::vector<std::pair<uint32_t, uint32_t>> no_veto;
  for (size_t t = 0; t < T; ++t) {
    const uint32_t g0 = thr_term_off[t], g1 = thr_term_off[t + 1];
    st.n_terms_in += g1 - g0;
    // ---- the anchor requirement of every term: its `In` with the fewest distinct values
    ta.assign(g1 - g0, TermAnchor());
    bool anchorable = t < thr_anchorable.size() && thr_anchorable[t] != 0 && g1 > g0;
    for (uint32_t g = g0; g < g1 && anchorable; ++g) {
      TermAnchor& a = ta[g - g0];
      if (term_kept && !(*term_kept)[g]) {
        a.dead = true;  // matches nowhere: no copy, no anchors
        continue;
      }
      for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
        if (req_op[r] != KT_OP_IN) continue;
        std::vector<uint32_t> v(req_val.begin() + req_val_off[r], req_val.begin() + req_val_off[r + 1]);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        if (v.empty()) {
          a.dead = true;  // In with no values: never satisfied
          break;
        }
        if (!a.has || v.size() < a.vals.size()) a.has = true, a.key = req_key[r], a.vals.swap(v);
      }
      if (!a.dead && !a.has) anchorable = false;  // a term every pod may match: the throttle stays whole
    }
    if (!anchorable) {
      // anchor 0: the throttle as it is
      AnchorSubProgram& sp = sub_of(subs, 0u);
      for (uint32_t g = g0; g < g1; ++g) emit_term(sp, g, term_flags, term_req_off, req_op, req_key, req_val_off, req_val, no_veto);
      sp.thr_orig.push_back((uint32_t)t);
      sp.thr_term_off.push_back((uint32_t)sp.term_orig.size());
      st.n_terms_out += g1 - g0;
      ++st.n_unanchored_throttles;
      ++st.n_virtual_throttles;
      continue;
    }
    // ---- the throttle's anchors with their keys, ascending by pair id
    std::vector<std::pair<uint32_t, uint32_t>> anchors;  // (pair, key)
    for (const TermAnchor& a : ta)
      if (!a.dead)
        for (uint32_t v : a.vals) anchors.push_back({v, a.key});
    std::sort(anchors.begin(), anchors.end());
    anchors.erase(std::unique(anchors.begin(), anchors.end(), [](const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) { return x.first == y.first; }),
                  anchors.end());
    std::vector<std::pair<uint32_t, uint32_t>> veto;  // (key, pair) of the earlier anchors, kept sorted
    for (size_t i = 0; i < anchors.size(); ++i) {
      const uint32_t a = anchors[i].first, a_key = anchors[i].second;
      AnchorSubProgram* sp = nullptr;
      size_t kept = 0;
      for (uint32_t g = g0; g < g1; ++g) {
        const TermAnchor& x = ta[g - g0];
        if (x.dead) continue;
        // the pod carries `a` on a_key: a term anchored on that key needs `a` among its values
        if (x.key == a_key && !std::binary_search(x.vals.begin(), x.vals.end(), a)) continue;
        // every anchor value of the term is vetoed in this copy (vals and the earlier anchors are both ascending by pair id)
        bool all_earlier = true;
        for (uint32_t v : x.vals) all_earlier &= v < a;  // earlier anchors = the throttle's anchors below `a`, and v IS one of them
        if (all_earlier) continue;
        if (!sp) sp = &sub_of(subs, a);
        emit_term(*sp, g, term_flags, term_req_off, req_op, req_key, req_val_off, req_val, veto);
        ++kept;
      }
      if (sp) {
        sp->thr_orig.push_back((uint32_t)t);
        sp->thr_term_off.push_back((uint32_t)sp->term_orig.size());
        st.n_terms_out += kept;
        ++st.n_virtual_throttles;
      }
      veto.insert(std::upper_bound(veto.begin(), veto.end(), std::make_pair(a_key, a)), {a_key, a});
    }
  }
  std::vector<AnchorSubProgram> out;
  out.reserve(subs.size());
  for (auto& kv : subs) out.push_back(std::move(kv.second));  // (std::map: ascending by anchor, anchor 0 first)
  st.n_anchors = out.size();
  if (stats) *stats = st;
  return out;
}


void build_anchored_index(AnchoredIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint8_t>& term_flags,
                          const std::vector<uint32_t>& term_req_off, const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                          const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                          const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns, const std::vector<uint32_t>& ns_term_ok, uint32_t gw,
                          uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int max_labels, uint32_t chk_word, const HostIndex& classic) {
  const size_t T = thr_term_off.empty() ? 0 : thr_term_off.size() - 1;
  // what may be split: live throttles without an unconvertible podSelector term and with at most 64 terms (the others keep
  // their place on the slow list of block 0)
  std::vector<uint8_t> anchorable(T, 0);
  for (size_t t = 0; t < T; ++t) {
    bool ok = thr_info((uint32_t)t).live && thr_term_off[t + 1] - thr_term_off[t] <= 64u;
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1] && ok; ++g) ok = !(term_flags[g] & KT_TERM_POD_SEL_INVALID);
    anchorable[t] = ok;
  }
  // the terms that can match at all (build_index's own rules: a live throttle, a namespace side that admits something)
  const size_t G_all = term_req_off.empty() ? 0 : term_req_off.size() - 1;
  std::vector<uint8_t> kept(G_all, 0);
  for (size_t t = 0; t < T; ++t) {
    const ThrInfo ti = thr_info((uint32_t)t);
    if (!ti.live) continue;
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g) {
      if (ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID)) continue;
      if (!ti.cluster && ti.ns >= n_ns) continue;
      bool any = false;
      for (uint32_t n = 0; n < n_ns && !any; ++n) any = (ns_term_ok[(size_t)n * gw + (g >> 5)] >> (g & 31)) & 1u;
      kept[g] = any;
    }
  }
  out = AnchoredIndex();
  const bool dbg = getenv("KT_DEBUG_COMPILE") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto t = std::chrono::steady_clock::now();
    if (dbg) fprintf(stderr, "build_anchored_index: %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  lap("anchorable / kept terms");
  const std::vector<AnchorSubProgram> subs = anchor_split(thr_term_off, term_flags, term_req_off, req_op, req_key, req_val_off, req_val, anchorable, &out.stats, &kept);
  out.n_ns = n_ns;
  lap("anchor_split");
  HostIndex& ox = out.ix;
  ox.atoms = classic.atoms, ox.atom_key = classic.atom_key, ox.atom_table = classic.atom_table;
  ox.bm_rows = classic.bm_rows, ox.la = classic.la, ox.rich = true;
  ox.n_pair_keys = classic.n_pair_keys, ox.n_key_atoms = classic.n_key_atoms, ox.n_keys = classic.n_keys, ox.n_ns = n_ns;
  const uint32_t n_blocks = (uint32_t)subs.size();
  ox.ns_words = ((size_t)n_blocks * n_ns + 31) / 32;
  if (!ox.ns_words) ox.ns_words = 1;
  uint64_t slab_run = 0;
  // the sub-indexes are independent of one another: built side by side (build_index keeps its scratch per thread and stays
  // single-threaded at these sizes), concatenated in block order afterwards
  std::vector<HostIndex> sxs(n_blocks);
  {
    uint32_t n_thr = std::min<uint32_t>(std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency())), std::max(1u, n_blocks / 4u));
    if (const char* ev = getenv("KT_ANCHOR_THREADS")) n_thr = std::max(1, atoi(ev));  // (measurements)
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
      std::vector<uint32_t> term_thr_unused;
      for (uint32_t c = next.fetch_add(1); c < n_blocks; c = next.fetch_add(1)) {
        const AnchorSubProgram& sp = subs[c];
        // the namespace side of the copied terms: the original term's
        const uint32_t G = (uint32_t)sp.term_orig.size(), gws = (G + 31) / 32 + 1;
        std::vector<uint32_t> ok((size_t)n_ns * gws, 0u);
        for (uint32_t g = 0; g < G; ++g) {
          const uint32_t og = sp.term_orig[g];
          for (uint32_t n = 0; n < n_ns; ++n)
            if ((ns_term_ok[(size_t)n * gw + (og >> 5)] >> (og & 31)) & 1u) ok[(size_t)n * gws + (g >> 5)] |= 1u << (g & 31);
        }
        build_index(sxs[c], sp.thr_term_off, term_thr_unused, sp.term_flags, sp.term_req_off, sp.req_op, sp.req_key, sp.req_val_off, sp.req_val,
                    [&](uint32_t v) { return thr_info(sp.thr_orig[v]); }, n_ns, ok, gws, agg_budget, chk_budget, thr_bytes, max_labels, nullptr, 0u,
                    chk_word, &classic);
      }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 1; k < n_thr; ++k) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  lap("sub-index builds");
  // which block a pod's atom opens: the anchors are pairs some kept term names, i.e. atoms of the classic numbering
  out.atom_block.assign(classic.bm_rows, 0u);
  for (uint32_t c = 1; c < n_blocks; ++c)
    for (const AtomId& a : classic.atoms)
      if (a.atom == subs[c].anchor) out.atom_block[a.id] = c;
  for (uint32_t c = 0; c < n_blocks; ++c) {
    const AnchorSubProgram& sp = subs[c];
    out.block_anchor.push_back(sp.anchor);
    out.blk_chunk0.push_back((uint32_t)ox.bm_chunks.size());
    const HostIndex& sx = sxs[c];
    const uint32_t rank_base = (uint32_t)ox.bm_rank_t.size();
    for (uint32_t r : sx.bm_rank_t) ox.bm_rank_t.push_back(sp.thr_orig[r]);
    for (uint32_t v : sx.slow_thr) ox.slow_thr.push_back(sp.thr_orig[v]);
    for (size_t k = 0; k < sx.bm_chunks.size(); ++k) {
      BmChunk ch = sx.bm_chunks[k];
      const size_t img0 = ox.bm_images.size();
      ox.bm_images.insert(ox.bm_images.end(), sx.bm_images.begin() + ch.img_off, sx.bm_images.begin() + ch.img_off + ch.img_bytes);
      // throttle rows and term ids of the original program
      uint32_t* tt = (uint32_t*)(ox.bm_images.data() + img0 + ch.off_term_t);
      uint32_t* tg = (uint32_t*)(ox.bm_images.data() + img0 + ch.off_term_g);
      for (uint32_t i = 0; i < ch.n_words * 64u; ++i) {
        if (!(tt[i] & kTermReal)) continue;
        tt[i] = (tt[i] & ~kTermRowMask) | sp.thr_orig[tt[i] & kTermRowMask];
        tg[i] = sp.term_orig[tg[i]];
      }
      ch.img_off = (uint32_t)img0;
      ch.rank0 += rank_base;
      ch.ns_base = c * n_ns, ch.ns_cnt = n_ns;
      ch.slab_off = (uint32_t)(slab_run / 16);
      slab_run += 256ull * (((uint64_t)ch.n_thr * thr_bytes + 15) & ~15ull) + 8192ull;
      // which virtual namespaces have words here
      const size_t b0 = ox.bm_chunk_ns.size();
      ox.bm_chunk_ns.resize(b0 + ox.ns_words, 0u);
      for (uint32_t n = 0; n < n_ns; ++n)
        if ((sx.bm_chunk_ns[k * sx.ns_words + (n >> 5)] >> (n & 31)) & 1u) {
          const uint32_t vns = c * n_ns + n;
          ox.bm_chunk_ns[b0 + (vns >> 5)] |= 1u << (vns & 31);
        }
      ox.bm_max_lds = std::max(ox.bm_max_lds, ch.lds_bytes), ox.bm_max_thr = std::max(ox.bm_max_thr, ch.n_thr);
      ox.bm_max_words = std::max(ox.bm_max_words, ch.n_words);
      ox.bm_chunks.push_back(ch);
    }
    ox.bm_words += sx.bm_words;
    ox.has_veto |= sx.has_veto, ox.has_slow |= sx.has_slow, ox.max_need = std::max(ox.max_need, sx.max_need);
  }
  out.blk_chunk0.push_back((uint32_t)ox.bm_chunks.size());
  ox.bm_slab_bytes = slab_run;
  lap("concatenation");
}

}  // namespace kt
