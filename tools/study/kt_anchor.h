// kt_anchor.h — the selector program split by ANCHOR ATOM (groundwork of the inverted scan, NEXT.md #1; host only, not
// wired into the engine yet: validated against brute force by tests/cpp/index_sim_test.cpp `--anchored`).
//
// A term with a positive `In` requirement can only match pods that carry one of its values.  Per throttle t the ANCHOR
// requirement of a term is its `In` with the fewest values; the throttle's anchors a_0 < a_1 < ... are the union of its
// terms' anchor values.  The throttle is copied once per anchor: copy i keeps the terms that can still match a pod that
// carries a_i and none of a_0 .. a_{i-1} — a term whose anchor requirement is on a_i's key but does not name a_i cannot,
// nor can one whose anchor values are all among the earlier anchors — and every kept term gets the extra requirement
// `NotIn {a_0 .. a_{i-1}}`.  For any pod exactly ONE copy of a throttle can fire (that of the first anchor the pod
// carries), and it fires iff the original throttle matches: the "reported once" rule of the scans holds per copy.  A
// throttle with a term that has no `In` (or with an unconvertible / oversized selector) stays whole in the sub-program of
// anchor 0, which every pod visits.
//
// The copies of anchor a form an ordinary selector program over the REAL namespaces (virtual throttle rows, terms as
// before): kt::build_index takes it as it is — one small index per anchor (configs[4]: 6 words, one chunk), walked by the
// items (pod, a) of the pods that carry a.  A pod visits the sub-indexes of its own atoms and of anchor 0:
// tools/anchor_study.py measured 21.9 word visits per pod instead of 75 on the configs[4] shard (group-level anchoring; this
// is the throttle-level form of the same idea, which needs no knowledge of the namespace cells).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "kt_index.h"  // (kube_throttler_amd/csrc, on the include path of index_sim_test)

namespace kt {

// one anchor's sub-program, in the flat form kt::build_index takes
struct AnchorSubProgram {
  uint32_t anchor = 0;                  // pair id of the anchor atom; 0 = the unanchored rest
  std::vector<uint32_t> thr_orig;       // virtual throttle row -> original throttle row
  std::vector<uint32_t> thr_term_off;   // [n_virtual + 1]
  std::vector<uint32_t> term_orig;      // virtual term -> original term (its namespace side and flags are the original's)
  std::vector<uint8_t> term_flags;
  std::vector<uint32_t> term_req_off;   // [n_terms + 1]
  std::vector<uint8_t> req_op;
  std::vector<uint32_t> req_key, req_val_off, req_val;
};

struct AnchorSplitStats {
  size_t n_anchors = 0, n_virtual_throttles = 0, n_terms_in = 0, n_terms_out = 0, n_unanchored_throttles = 0;
};

// thr_anchorable[t]: the throttle may be split (live, no unconvertible podSelector term, <= 64 terms); everything else goes
// to anchor 0 untouched.  term_kept[g] (optional): the term can match at all (its throttle is live, some namespace admits
// it) — the others are left out: they contribute neither copies nor anchors, so that every anchor is a pair some KEPT term
// names, i.e. an atom of the classic index.
std::vector<AnchorSubProgram> anchor_split(const std::vector<uint32_t>& thr_term_off, const std::vector<uint8_t>& term_flags,
                                           const std::vector<uint32_t>& term_req_off, const std::vector<uint8_t>& req_op,
                                           const std::vector<uint32_t>& req_key, const std::vector<uint32_t>& req_val_off,
                                           const std::vector<uint32_t>& req_val, const std::vector<uint8_t>& thr_anchorable,
                                           AnchorSplitStats* stats = nullptr, const std::vector<uint8_t>* term_kept = nullptr);


// The per-anchor indexes CONCATENATED into one chunked index the scan kernels can walk as it is (NEXT.md #1): block c = the
// c-th anchor (block 0: anchor 0) serves the VIRTUAL namespaces [c * n_ns, (c + 1) * n_ns) — an item (pod, anchor it
// carries) is a virtual pod of namespace c * n_ns + pod.ns — through BmChunk::ns_base / ns_cnt; every sub-index is built
// by kt::build_index with the atom numbering of `classic` imposed (the pods' atom rows are translated once), its ranks,
// throttle rows and term ids rewritten to the original program's.
struct AnchoredIndex {
  HostIndex ix;                        // chunks of all blocks back to back; atoms / atom_table = classic's
  std::vector<uint32_t> block_anchor;  // block -> anchor pair id (block 0: 0)
  std::vector<uint32_t> atom_block;    // [ix.bm_rows] atom id (a pod's atom row holds these) -> the block it opens, 0 = none:
                                       // a pod's items are block 0 and atom_block[id] of its atoms
  std::vector<uint32_t> blk_chunk0;    // [blocks + 1] first chunk of every block
  uint32_t n_ns = 0;                   // real namespaces per block
  AnchorSplitStats stats;
};
void build_anchored_index(AnchoredIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint8_t>& term_flags,
                          const std::vector<uint32_t>& term_req_off, const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                          const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                          const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns, const std::vector<uint32_t>& ns_term_ok, uint32_t gw,
                          uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int max_labels, uint32_t chk_word, const HostIndex& classic);

}  // namespace kt
