// kt_anchor.cpp — see kt_anchor.h
#include "kt_anchor.h"

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
                                           AnchorSplitStats* stats) {
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

}  // namespace kt
