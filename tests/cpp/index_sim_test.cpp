// index_sim_test — CPU-only check of the selector index builder (kube_throttler_amd/csrc/kt_index.cpp): random selector
// programs are compiled with kt::build_index, then every chunk image is decoded and the scan of kt_bitmap_scan.h is
// replayed on the host, pod by pod (advance over the namespace's word list -> candidate bits -> TermRec decision:
// second pair / inline extras / generic walk -> adjacent-throttle dedup; slow list walked in order).  The matched /
// errored throttle set of every pod must equal the brute-force evaluation of the program (OR of terms in order, AND of
// requirements, throttle_selector.go:30-54, clusterthrottle_selector.go:30-87), for LDS budgets from "one chunk" down to
// "a few words per chunk".  Structural invariants of the chunking are checked on the way.  No device call is made.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "kt_index.h"
#include "../../include/kt_snapshot.h"

using namespace kt;

static int g_fail = 0;
static long g_peel_steps = 0, g_peel_busy = 0;      // today's peel loop on the same full tiles
static long g_decide_steps = 0, g_busy_lanes = 0;
static long g_pair2_fail = 0;
static long g_expand_steps = 0, g_adv_today = 0;  // advance/expand rounds of the blueprint; advance rounds of today's loop  // candidates that only fail their second matchLabels pair (what a second row family would never list)  // lane-parallel blueprint: decision steps and the lanes busy in them
#define EXPECT(cond, ...)                                             \
  do {                                                                \
    if (!(cond)) {                                                    \
      if (++g_fail < 20) {                                            \
        fprintf(stderr, "FAIL %s:%d: %s  ", __FILE__, __LINE__, #cond); \
        fprintf(stderr, __VA_ARGS__);                                 \
        fprintf(stderr, "\n");                                        \
      }                                                               \
    }                                                                 \
  } while (0)

struct Program {
  std::vector<uint32_t> thr_term_off{0}, term_thr, term_req_off{0}, req_key, req_val_off{0}, req_val;
  std::vector<uint8_t> term_flags, req_op;
  std::vector<ThrInfo> thr;
  uint32_t n_ns = 0, gw = 0;
  std::vector<uint32_t> ns_term_ok;  // [n_ns][gw]
  uint32_t K = 0, V = 0;
  uint32_t pair(uint32_t key, uint32_t val) const { return 1 + (key - 1) * V + val; }  // key ids 1..K, values 0..V-1
};

struct PodLabels {
  uint32_t ns;
  std::vector<uint32_t> keys, pairs;  // one value per key
  bool has_pair(uint32_t p) const { return std::find(pairs.begin(), pairs.end(), p) != pairs.end(); }
  bool has_key(uint32_t k) const { return std::find(keys.begin(), keys.end(), k) != keys.end(); }
};

static bool req_ok(const Program& p, uint32_t r, const PodLabels& pod) {
  const uint8_t op = p.req_op[r];
  if (op == KT_OP_IN || op == KT_OP_NOT_IN) {
    bool in = false;
    for (uint32_t j = p.req_val_off[r]; j < p.req_val_off[r + 1]; ++j) in |= pod.has_pair(p.req_val[j]);
    return op == KT_OP_IN ? in : !in;
  }
  const bool has = pod.has_key(p.req_key[r]);
  return op == KT_OP_EXISTS ? has : !has;
}
static bool term_ok(const Program& p, uint32_t g, const PodLabels& pod) {
  for (uint32_t r = p.term_req_off[g]; r < p.term_req_off[g + 1]; ++r)
    if (!req_ok(p, r, pod)) return false;
  return true;
}
static bool ns_ok(const Program& p, uint32_t g, uint32_t ns) { return (p.ns_term_ok[(size_t)ns * p.gw + (g >> 5)] >> (g & 31)) & 1u; }

// 0 = no match, 1 = match, 2 = error (the unconvertible term is reached before a match)
static int brute(const Program& p, uint32_t t, const PodLabels& pod) {
  if (!p.thr[t].live) return 0;
  for (uint32_t g = p.thr_term_off[t]; g < p.thr_term_off[t + 1]; ++g) {
    const bool applies = ns_ok(p, g, pod.ns);
    if (p.term_flags[g] & KT_TERM_POD_SEL_INVALID) {
      if (applies) return 2;
      continue;
    }
    if (applies && term_ok(p, g, pod)) return 1;
  }
  return 0;
}

static Program random_program(std::mt19937& rng, uint32_t T, uint32_t n_ns, uint32_t K, uint32_t V, int max_terms, int max_reqs,
                              double p_bad) {
  auto U = [&](uint32_t n) { return (uint32_t)(rng() % n); };
  auto chance = [&](double x) { return (rng() % 100000) < x * 100000; };
  Program p;
  p.n_ns = n_ns, p.K = K, p.V = V;
  std::vector<std::vector<uint32_t>> term_ns;  // namespaces a term applies to
  for (uint32_t t = 0; t < T; ++t) {
    ThrInfo ti;
    ti.live = !chance(0.08);
    ti.cluster = chance(0.5);
    ti.ns = ti.cluster ? 0 : (chance(0.03) ? n_ns + 3 : U(n_ns));  // a Throttle in a namespace nobody registered
    p.thr.push_back(ti);
    const int nt = (int)U((uint32_t)max_terms + 1);
    std::vector<uint32_t> cluster_adm;  // most ClusterThrottles use one namespaceSelector for all their terms
    for (uint32_t n = 0; n < n_ns; ++n)
      if (chance(0.4)) cluster_adm.push_back(n);
    for (int k = 0; k < nt; ++k) {
      uint8_t fl = 0;
      if (chance(p_bad)) fl |= KT_TERM_POD_SEL_INVALID;
      if (ti.cluster && chance(0.05)) fl |= KT_TERM_NS_SEL_INVALID;
      p.term_thr.push_back(t);
      p.term_flags.push_back(fl);
      std::vector<uint32_t> adm;
      if (ti.live && !(fl & KT_TERM_NS_SEL_INVALID)) {
        if (!ti.cluster) {
          if (ti.ns < n_ns) adm.push_back(ti.ns);
        } else if (chance(0.8)) adm = cluster_adm;
        else
          for (uint32_t n = 0; n < n_ns; ++n)
            if (chance(0.5)) adm.push_back(n);
      }
      term_ns.push_back(adm);
      const int nr = (fl & KT_TERM_POD_SEL_INVALID) ? 0 : (int)U((uint32_t)max_reqs + 1);
      std::set<uint32_t> used_keys;
      for (int r = 0; r < nr; ++r) {
        const uint32_t key = 1 + U(K);
        const uint32_t opsel = U(10);
        const uint8_t op = opsel < 6 ? KT_OP_IN : opsel < 8 ? KT_OP_NOT_IN : opsel < 9 ? KT_OP_EXISTS : KT_OP_DOES_NOT_EXIST;
        p.req_op.push_back(op);
        p.req_key.push_back(key);
        if (op == KT_OP_IN || op == KT_OP_NOT_IN) {
          const uint32_t nv = 1 + (chance(0.3) ? U(std::min<uint32_t>(V, 5)) : 0);
          for (uint32_t j = 0; j < nv; ++j) p.req_val.push_back(p.pair(key, U(V)));  // duplicates allowed
        }
        p.req_val_off.push_back((uint32_t)p.req_val.size());
      }
      p.term_req_off.push_back((uint32_t)p.req_op.size());
    }
    p.thr_term_off.push_back((uint32_t)p.term_thr.size());
  }
  const uint32_t G = (uint32_t)p.term_thr.size();
  p.gw = (G + 31) / 32 + 1;
  p.ns_term_ok.assign((size_t)n_ns * p.gw, 0u);
  for (uint32_t g = 0; g < G; ++g)
    for (uint32_t n : term_ns[g]) p.ns_term_ok[(size_t)n * p.gw + (g >> 5)] |= 1u << (g & 31);
  return p;
}

static uint32_t row_of_atom(const HostIndex& ix, uint32_t atom) {
  if (!atom) return 1;
  const AtomBucket& b = ix.bm_buckets[atom_bucket(atom, ix.bm_bucket_mask, ix.bm_bucket_mult)];
  for (int k = 0; k < 4; ++k)
    if (b.atom[k] == atom) return b.row[k];
  return 1;
}

static bool extra_ok(const uint32_t e[4], const PodLabels& pod) {
  const uint32_t op = e[0] & 0xFFu;
  if (op == 0xFFu) return true;
  bool hit = false;
  for (int j = 1; j < 4; ++j) {
    if (e[j] == kNoAtom) continue;
    hit |= (op >= KT_OP_EXISTS) ? pod.has_key(e[j]) : pod.has_pair(e[j]);
  }
  return (op == KT_OP_IN || op == KT_OP_EXISTS) ? hit : !hit;
}

// replays bitmap_scan_tile for one pod -> per-throttle result (1 match / 2 error), checks "reported once"
static std::map<uint32_t, int> scan(const Program& p, const HostIndex& ix, const PodLabels& pod) {
  std::map<uint32_t, int> out;
  for (const BmChunk& ch : ix.bm_chunks) {
    const unsigned char* img = ix.bm_images.data() + ch.img_off;
    const uint64_t* rows = (const uint64_t*)img;
    const uint64_t* nsrows = (const uint64_t*)(img + ch.off_nsrows);
    const uint32_t* nsw_off = (const uint32_t*)(img + ch.off_nsw_off);
    const uint32_t* nsw = (const uint32_t*)(img + ch.off_nsw);
    const TermRec* trec = (const TermRec*)(img + ch.off_trec);
    const TermX* trecx = (const TermX*)(img + ch.off_trecx);
    std::vector<uint32_t> prow{0};
    for (uint32_t pr : pod.pairs) prow.push_back(row_of_atom(ix, pr));
    if (ix.bm_has_key_rows)
      for (uint32_t k : pod.keys) prow.push_back(row_of_atom(ix, kKeyAtom | k));
    uint32_t last_t = ~0u;
    for (uint32_t k = nsw_off[pod.ns]; k < nsw_off[pod.ns + 1]; ++k) {
      const uint32_t w = nsw[k];
      EXPECT(w < ch.n_words, "word %u of %u", w, ch.n_words);
      uint64_t x = 0;
      for (uint32_t r : prow) {
        EXPECT(r < ix.bm_rows, "row %u of %u", r, ix.bm_rows);
        x |= rows[(size_t)r * ch.stride + w];
      }
      x &= nsrows[(size_t)pod.ns * ch.stride + w];
      while (x) {
        const uint32_t c = w * 64 + (uint32_t)__builtin_ctzll(x);
        x &= x - 1;
        const TermRec& tr = trec[c];
        bool ok = !(tr.flags & kPostPair2) || pod.has_pair(tr.pair2);
        if (ok && (tr.flags & kPostInline)) {
          EXPECT(ix.bm_has_inline, "inline term without TermX table");
          ok = extra_ok(trecx[c].e[0], pod) && extra_ok(trecx[c].e[1], pod);
        }
        if (ok && (tr.flags & kPostComplex)) ok = term_ok(p, tr.g, pod);
        EXPECT(ch.rank0 + (tr.flags >> 8) < ix.bm_rank_t.size() && ix.bm_rank_t[ch.rank0 + (tr.flags >> 8)] == tr.t,
               "rank of term %u", tr.g);
        EXPECT((tr.flags >> 8) < ch.n_thr, "chunk-local rank %u of %u", tr.flags >> 8, ch.n_thr);
        if ((tr.flags & kPostAdj) && tr.t == last_t) ok = false;
        if (ok) {
          last_t = tr.t;
          EXPECT(!out.count(tr.t), "throttle %u reported twice", tr.t);
          out[tr.t] = 1;
        }
      }
    }
    // the namespace's word list is exactly the words its admission row can touch
    std::vector<uint32_t> want;
    for (uint32_t w = 0; w < ch.n_words; ++w)
      if (nsrows[(size_t)pod.ns * ch.stride + w]) want.push_back(w);
    EXPECT(std::vector<uint32_t>(nsw + nsw_off[pod.ns], nsw + nsw_off[pod.ns + 1]) == want, "word list of ns %u", pod.ns);
  }
  for (uint32_t t : ix.slow_thr) {
    const int r = brute(p, t, pod);  // walk_slow IS the in-order walk
    EXPECT(!out.count(t), "slow throttle %u also indexed", t);
    if (r) out[t] = r;
  }
  return out;
}

// ---- step count of TODAY's wave-level loop on a tile (kt_bitmap_scan.h: peel while any lane has a bit, else advance
//      all lanes): only the control flow, to put the blueprint's occupancy next to the current one on the same inputs
static void count_current_steps(const HostIndex& ix, const std::vector<PodLabels>& tile, long* peel_steps, long* busy) {
  for (const BmChunk& ch : ix.bm_chunks) {
    const unsigned char* img = ix.bm_images.data() + ch.img_off;
    const uint64_t* rows = (const uint64_t*)img;
    const uint64_t* nsrows = (const uint64_t*)(img + ch.off_nsrows);
    const uint32_t* nsw_off = (const uint32_t*)(img + ch.off_nsw_off);
    const uint32_t* nsw = (const uint32_t*)(img + ch.off_nsw);
    const size_t n = tile.size();
    std::vector<uint32_t> k(n), k1(n);
    std::vector<uint64_t> x(n, 0);
    std::vector<std::vector<uint32_t>> prow(n);
    for (size_t l = 0; l < n; ++l) {
      prow[l].push_back(0);
      for (uint32_t pr : tile[l].pairs) prow[l].push_back(row_of_atom(ix, pr));
      if (ix.bm_has_key_rows)
        for (uint32_t key : tile[l].keys) prow[l].push_back(row_of_atom(ix, kKeyAtom | key));
      k[l] = nsw_off[tile[l].ns], k1[l] = nsw_off[tile[l].ns + 1];
    }
    for (;;) {
      long has = 0;
      for (size_t l = 0; l < n; ++l) has += x[l] != 0;
      if (has) {
        ++*peel_steps, *busy += has;
        for (size_t l = 0; l < n; ++l) x[l] &= x[l] - 1;
        continue;
      }
      bool adv = false;
      for (size_t l = 0; l < n; ++l)
        if (k[l] < k1[l]) {
          const uint32_t w = nsw[k[l]++];
          uint64_t xx = 0;
          for (uint32_t r : prow[l]) xx |= rows[(size_t)r * ch.stride + w];
          x[l] = xx & nsrows[(size_t)tile[l].ns * ch.stride + w];
          adv = true;
        }
      if (!adv) break;
      ++g_adv_today;
    }
  }
}

// ---- blueprint of the lane-parallel scan planned for dense programs (DESIGN.md 7, next lever 4), written the way a
//      wave executes it: every "step" below is one pass over the 64 lanes, ballots are explicit masks.
//   advance : a lane without candidate bits takes its next word (as today);
//   expand  : every lane moves up to kQuota of its candidate bits into the wave's LDS list as (lane, term) entries;
//             slots come from an exclusive prefix sum of the per-lane counts (no decisions, no dependent reads);
//   decide  : whenever kListCap - 64 * kQuota entries are waiting (and at the end) the list is decided 64 entries at a
//             time with lane = entry; a match of a multi-term throttle counts only if its bit in
//             seen[pod lane][chunk-local throttle rank] was clear (LDS atomic OR returning the old word).
static std::vector<std::map<uint32_t, int>> scan_tile_lane_parallel(const Program& p, const HostIndex& ix,
                                                                   const std::vector<PodLabels>& tile, long* decide_steps,
                                                                   long* busy_lanes) {
  constexpr uint32_t kQuota = 4, kListCap = 512;
  const size_t n = tile.size();  // <= 64
  std::vector<std::map<uint32_t, int>> out(n);
  struct Entry { uint32_t lane, c; };
  for (const BmChunk& ch : ix.bm_chunks) {
    const unsigned char* img = ix.bm_images.data() + ch.img_off;
    const uint64_t* rows = (const uint64_t*)img;
    const uint64_t* nsrows = (const uint64_t*)(img + ch.off_nsrows);
    const uint32_t* nsw_off = (const uint32_t*)(img + ch.off_nsw_off);
    const uint32_t* nsw = (const uint32_t*)(img + ch.off_nsw);
    const TermRec* trec = (const TermRec*)(img + ch.off_trec);
    const TermX* trecx = (const TermX*)(img + ch.off_trecx);
    const uint32_t seen_words = (ch.n_thr + 63) / 64;
    std::vector<uint64_t> seen((size_t)64 * (seen_words ? seen_words : 1), 0ull);  // cleared per tile and chunk
    std::vector<std::vector<uint32_t>> prow(n);
    std::vector<uint32_t> k(n), k1(n), w(n, 0);
    std::vector<uint64_t> x(n, 0);
    for (size_t l = 0; l < n; ++l) {
      prow[l].push_back(0);
      for (uint32_t pr : tile[l].pairs) prow[l].push_back(row_of_atom(ix, pr));
      if (ix.bm_has_key_rows)
        for (uint32_t key : tile[l].keys) prow[l].push_back(row_of_atom(ix, kKeyAtom | key));
      k[l] = nsw_off[tile[l].ns], k1[l] = nsw_off[tile[l].ns + 1];
    }
    std::vector<Entry> list;
    auto decide = [&]() {
      for (size_t base = 0; base < list.size(); base += 64) {
        ++*decide_steps;
        // lane = entry: everything below is per-lane work on registers / LDS reads, one LDS atomic for kPostAdj matches
        for (size_t e = base; e < std::min(list.size(), base + 64); ++e) {
          ++*busy_lanes;
          const PodLabels& pod = tile[list[e].lane];
          const TermRec& tr = trec[list[e].c];
          bool ok = !(tr.flags & kPostPair2) || pod.has_pair(tr.pair2);
          g_pair2_fail += !ok;
          if (ok && (tr.flags & kPostInline)) ok = extra_ok(trecx[list[e].c].e[0], pod) && extra_ok(trecx[list[e].c].e[1], pod);
          if (ok && (tr.flags & kPostComplex)) ok = term_ok(p, tr.g, pod);
          if (ok && (tr.flags & kPostAdj)) {
            const uint32_t rank = tr.flags >> 8;
            uint64_t& word = seen[(size_t)list[e].lane * seen_words + (rank >> 6)];
            const uint64_t bit = 1ull << (rank & 63);
            ok = !(word & bit);  // ds_or_rtn_b64: first match of this (pod, throttle) wins, whatever the order
            word |= bit;
          }
          if (ok) {
            EXPECT(!out[list[e].lane].count(tr.t), "throttle %u reported twice (lane-parallel)", tr.t);
            out[list[e].lane][tr.t] = 1;
          }
        }
      }
      list.clear();
    };
    for (;;) {
      // advance
      bool any = false;
      for (size_t l = 0; l < n; ++l) {
        if (x[l] == 0 && k[l] < k1[l]) {
          w[l] = nsw[k[l]++];
          uint64_t xx = 0;
          for (uint32_t r : prow[l]) xx |= rows[(size_t)r * ch.stride + w[l]];
          x[l] = xx & nsrows[(size_t)tile[l].ns * ch.stride + w[l]];
        }
        any |= x[l] != 0 || k[l] < k1[l];
      }
      if (!any) break;
      ++g_expand_steps;
      // expand: per-lane counts -> exclusive prefix sum -> entries in (lane-major, ascending term) order
      uint32_t cnt[64] = {0}, pre[64] = {0}, total = 0;
      for (size_t l = 0; l < n; ++l) cnt[l] = std::min<uint32_t>((uint32_t)__builtin_popcountll(x[l]), kQuota);
      for (size_t l = 0; l < n; ++l) pre[l] = total, total += cnt[l];
      const size_t at = list.size();
      list.resize(at + total);
      for (size_t l = 0; l < n; ++l)
        for (uint32_t q = 0; q < cnt[l]; ++q) {
          list[at + pre[l] + q] = Entry{(uint32_t)l, w[l] * 64 + (uint32_t)__builtin_ctzll(x[l])};
          x[l] &= x[l] - 1;
        }
      EXPECT(list.size() <= kListCap, "list overflow %zu", list.size());
      if (list.size() > kListCap - 64 * kQuota) decide();
    }
    decide();
  }
  for (size_t l = 0; l < n; ++l)
    for (uint32_t t : ix.slow_thr) {
      const int r = brute(p, t, tile[l]);
      if (r) out[l][t] = r;
    }
  return out;
}

static void check_structure(const Program& p, const HostIndex& ix, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes) {
  uint32_t w = 0, rank = 0;
  const size_t bucket_bytes = ix.bm_buckets.size() * sizeof(AtomBucket);
  std::set<uint32_t> seen_t;
  for (size_t i = 0; i < ix.bm_chunks.size(); ++i) {
    const BmChunk& ch = ix.bm_chunks[i];
    EXPECT(ch.w0 == w && ch.n_words >= 1, "chunk %zu starts at word %u, expected %u", i, ch.w0, w);
    EXPECT(ch.stride == (ch.n_words | 1u) && ch.img_off % 16 == 0 && ch.img_bytes % 16 == 0, "chunk %zu layout", i);
    EXPECT(ch.img_off + (size_t)ch.img_bytes <= ix.bm_images.size(), "chunk %zu image range", i);
    if (ch.n_thr) {
      EXPECT(ch.rank0 == rank, "chunk %zu rank0 %u, expected %u (a throttle straddles chunks?)", i, ch.rank0, rank);
      rank = ch.rank0 + ch.n_thr;
    }
    // a multi-word chunk respects both budgets (a single unsplittable stretch may not: the launchers refuse it)
    const bool fits = bucket_bytes + ch.img_bytes <= chk_budget &&
                      bucket_bytes + ch.img_bytes + (size_t)ch.n_thr * thr_bytes + 16 <= agg_budget;
    if (!fits && ch.n_words > 1) {
      // over budget: then no earlier cut was possible — every inner word boundary splits a throttle
      const unsigned char* img = ix.bm_images.data() + ch.img_off;
      const uint64_t* rows = (const uint64_t*)img;
      const TermRec* trec = (const TermRec*)(img + ch.off_trec);
      auto real = [&](uint32_t c) {  // a term number in use has a bit in some atom row (or in row 0)
        for (uint32_t r = 0; r < ix.bm_rows; ++r)
          if ((rows[(size_t)r * ch.stride + (c >> 6)] >> (c & 63)) & 1ull) return true;
        return false;
      };
      for (uint32_t wb = 1; wb < ch.n_words; ++wb) {
        int64_t a = -1, b = -1;
        for (int k = 63; k >= 0 && a < 0; --k)
          if (real((wb - 1) * 64 + k)) a = (int64_t)(wb - 1) * 64 + k;
        for (int k = 0; k < 64 && b < 0; ++k)
          if (real(wb * 64 + k)) b = (int64_t)wb * 64 + k;
        EXPECT(a >= 0 && b >= 0 && (trec[a].flags >> 8) == (trec[b].flags >> 8),
               "chunk %zu exceeds the budget although it could have been cut at word %u", i, wb);
      }
    }
    EXPECT(ch.img_bytes <= ix.bm_max_img && ch.n_thr <= ix.bm_max_thr, "chunk %zu maxima", i);
    w += ch.n_words;
  }
  // the launchers size LDS from the LARGEST image and the LARGEST throttle count of any chunk (kt_kernels_aggregate.hip
  // make_bm_agg_args): the pair of maxima has to fit too, or the engine refuses the program
  bool all_fit = true;
  for (const BmChunk& ch : ix.bm_chunks)
    all_fit &= bucket_bytes + ch.img_bytes <= chk_budget && bucket_bytes + ch.img_bytes + (size_t)ch.n_thr * thr_bytes + 16 <= agg_budget;
  if (all_fit)
    EXPECT(bucket_bytes + ix.bm_max_img + (size_t)ix.bm_max_thr * thr_bytes + 16 <= agg_budget,
           "largest image %u B + table of the largest chunk (%u throttles) = %zu B exceed the aggregate budget %u B", ix.bm_max_img,
           ix.bm_max_thr, bucket_bytes + ix.bm_max_img + (size_t)ix.bm_max_thr * thr_bytes + 16, agg_budget);
  EXPECT(w == ix.bm_words, "chunks cover %u of %u words", w, ix.bm_words);
  EXPECT(rank == ix.bm_rank_t.size(), "ranks cover %u of %zu", rank, ix.bm_rank_t.size());
  for (uint32_t t : ix.bm_rank_t) {
    EXPECT(!seen_t.count(t), "throttle %u has two ranks", t);
    seen_t.insert(t);
    EXPECT(p.thr[t].live, "dead throttle %u indexed", t);
  }
  for (uint32_t t : ix.slow_thr) EXPECT(!seen_t.count(t) && p.thr[t].live, "slow throttle %u", t);
}

static long run_case(uint32_t seed, uint32_t T, uint32_t n_ns, uint32_t K, uint32_t V, int max_terms, int max_reqs, double p_bad,
                     uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int n_pods) {
  std::mt19937 rng(seed);
  const Program p = random_program(rng, T, n_ns, K, V, max_terms, max_reqs, p_bad);
  HostIndex ix;
  build_index(ix, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
              [&](uint32_t t) { return p.thr[t]; }, n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes);
  check_structure(p, ix, agg_budget, chk_budget, thr_bytes);
  long matches = 0;
  std::vector<PodLabels> tile;
  auto flush_tile = [&]() {
    if (tile.empty()) return;
    const auto got = scan_tile_lane_parallel(p, ix, tile, &g_decide_steps, &g_busy_lanes);
    if (tile.size() == 64) count_current_steps(ix, tile, &g_peel_steps, &g_peel_busy);
    for (size_t l = 0; l < tile.size(); ++l)
      for (uint32_t t = 0; t < T; ++t) {
        const int want = brute(p, t, tile[l]);
        const auto it = got[l].find(t);
        EXPECT((it == got[l].end() ? 0 : it->second) == want, "seed %u throttle %u: lane-parallel scan says %d, program says %d",
               seed, t, it == got[l].end() ? 0 : it->second, want);
      }
    tile.clear();
  };
  for (int i = 0; i < n_pods; ++i) {
    PodLabels pod;
    pod.ns = rng() % n_ns;
    for (uint32_t k = 1; k <= K; ++k)
      if (rng() % 100 < 55) pod.keys.push_back(k), pod.pairs.push_back(p.pair(k, rng() % V));
    const std::map<uint32_t, int> got = scan(p, ix, pod);
    for (uint32_t t = 0; t < T; ++t) {
      const int want = brute(p, t, pod);
      const auto it = got.find(t);
      const int have = it == got.end() ? 0 : it->second;
      EXPECT(have == want, "seed %u pod %d throttle %u: index says %d, program says %d", seed, i, t, have, want);
      matches += want == 1;
    }
    tile.push_back(pod);
    if (tile.size() == 64) flush_tile();
  }
  flush_tile();
  return (long)ix.bm_chunks.size() * 1000000L + matches % 1000000L;
}

// ---- file mode: the REAL selector program of a BASELINE config + a pod sample (tools/dump_program.py)
static std::vector<uint32_t> read_array(FILE* fh) {
  uint32_t n = 0;
  if (fread(&n, 4, 1, fh) != 1) return {};
  std::vector<uint32_t> v(n);
  if (n && fread(v.data(), 4, n, fh) != n) v.clear();
  return v;
}
static int run_file(const char* path) {
  FILE* fh = fopen(path, "rb");
  if (!fh) { fprintf(stderr, "cannot open %s\n", path); return 2; }
  const std::vector<uint32_t> hdr = read_array(fh);
  Program p;
  p.thr_term_off = read_array(fh);
  p.term_thr = read_array(fh);
  for (uint32_t f : read_array(fh)) p.term_flags.push_back((uint8_t)f);
  p.term_req_off = read_array(fh);
  for (uint32_t o : read_array(fh)) p.req_op.push_back((uint8_t)o);
  p.req_key = read_array(fh);
  p.req_val_off = read_array(fh);
  p.req_val = read_array(fh);
  const std::vector<uint32_t> live = read_array(fh), cluster = read_array(fh), thr_ns = read_array(fh);
  p.ns_term_ok = read_array(fh);
  const std::vector<uint32_t> pod_ns = read_array(fh), loff = read_array(fh), lkey = read_array(fh), lpair = read_array(fh);
  fclose(fh);
  const uint32_t T = hdr[0], NS = hdr[2], D = hdr[5];
  p.n_ns = NS, p.gw = hdr[3];
  for (uint32_t t = 0; t < T; ++t) p.thr.push_back(ThrInfo{live[t] != 0, cluster[t] != 0, thr_ns[t]});
  HostIndex ix;
  const uint32_t thr_bytes = 8 * D + 8;
  build_index(ix, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
              [&](uint32_t t) { return p.thr[t]; }, NS, p.ns_term_ok, p.gw, 160u * 1024u - aggregate_fixed_lds(),
              160u * 1024u - check_fixed_lds(), thr_bytes);
  check_structure(p, ix, 160u * 1024u - aggregate_fixed_lds(), 160u * 1024u - check_fixed_lds(), thr_bytes);
  std::vector<PodLabels> tile;
  long cand = 0, matches = 0, pods = 0;
  for (size_t i = 0; i < pod_ns.size(); ++i) {
    PodLabels pod;
    pod.ns = pod_ns[i];
    for (uint32_t j = loff[i]; j < loff[i + 1]; ++j) pod.keys.push_back(lkey[j]), pod.pairs.push_back(lpair[j]);
    tile.push_back(pod);
    if (tile.size() < 64) continue;
    const auto got = scan_tile_lane_parallel(p, ix, tile, &g_decide_steps, &g_busy_lanes);
    count_current_steps(ix, tile, &g_peel_steps, &g_peel_busy);
    for (size_t l = 0; l < 64; ++l) {
      if (pods % 16 == 0) {  // brute force (T x terms per pod) on a sample; the per-pod replay on the same pods
        const auto one = scan(p, ix, tile[l]);
        for (uint32_t t = 0; t < T; ++t) {
          const int want = brute(p, t, tile[l]);
          const auto a = got[l].find(t), b = one.find(t);
          EXPECT((a == got[l].end() ? 0 : a->second) == want && (b == one.end() ? 0 : b->second) == want, "pod %ld throttle %u", pods, t);
        }
      }
      matches += (long)got[l].size();
      ++pods;
    }
    tile.clear();
  }
  cand = g_busy_lanes;
  printf("%s: %u throttles, %zu terms, %u namespaces -> %zu chunks (max image %u B, max %u throttles per chunk), %zu slow\n", path, T,
         p.term_thr.size(), NS, ix.bm_chunks.size(), ix.bm_max_img, ix.bm_max_thr, ix.slow_thr.size());
  {  // what the LDS image is made of, and which decision shapes the program has
    long n_pair2 = 0, n_inline = 0, n_complex = 0, n_plain = 0, n_adj = 0;
    for (const BmChunk& ch : ix.bm_chunks) {
      const TermRec* trec = (const TermRec*)(ix.bm_images.data() + ch.img_off + ch.off_trec);
      const uint64_t* rows = (const uint64_t*)(ix.bm_images.data() + ch.img_off);
      for (uint32_t c = 0; c < ch.n_words * 64; ++c) {
        bool real = false;
        for (uint32_t r = 0; r < ix.bm_rows && !real; ++r) real = (rows[(size_t)r * ch.stride + (c >> 6)] >> (c & 63)) & 1ull;
        if (!real) continue;
        const uint32_t f = trec[c].flags;
        n_pair2 += (f & kPostPair2) != 0, n_inline += (f & kPostInline) != 0, n_complex += (f & kPostComplex) != 0;
        n_plain += !(f & (kPostPair2 | kPostInline | kPostComplex)), n_adj += (f & kPostAdj) != 0;
      }
    }
    const BmChunk& c0 = ix.bm_chunks[0];
    printf("  term shapes: %ld anchor only, %ld + second pair, %ld + inline extras (TermX), %ld generic walk; %ld in multi-term throttles\n",
           n_plain, n_pair2, n_inline, n_complex, n_adj);
    printf("  image of chunk 0 (%u B): rows %u B, nsrows %u B, word lists %u B, TermRec %u B, TermX %u B; atom buckets %zu B, %u rows\n",
           c0.img_bytes, c0.off_nsrows, c0.off_nsw_off - c0.off_nsrows, c0.off_trec - c0.off_nsw_off, c0.off_trecx - c0.off_trec,
           c0.img_bytes - c0.off_trecx, ix.bm_buckets.size() * sizeof(AtomBucket), ix.bm_rows);
  }
  printf("  %ld pods: %.1f candidate terms and %.1f matches per pod (%.1f candidates per pod fail only their second matchLabels pair)\n", pods,
         (double)cand / (double)pods, (double)matches / (double)pods, (double)g_pair2_fail / (double)pods);
  printf("  today's peel loop      : %8ld steps per 64-pod tile x chunk walk, %5.1f %% of lanes busy\n", g_peel_steps,
         100.0 * (double)g_peel_busy / (64.0 * (double)g_peel_steps));
  printf("  lane-parallel blueprint: %8ld decision steps,                    %5.1f %% of lanes busy  (%.1fx fewer steps)\n", g_decide_steps,
         100.0 * (double)g_busy_lanes / (64.0 * (double)g_decide_steps), (double)g_peel_steps / (double)g_decide_steps);
  printf("  rounds without decisions : today %ld advance rounds, blueprint %ld advance+expand rounds (up to 4 candidates per lane each)\n",
         g_adv_today, g_expand_steps);
  if (g_fail) fprintf(stderr, "%d expectation(s) failed\n", g_fail);
  return g_fail ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc > 1) return run_file(argv[1]);
  long chunks_seen = 0, matches = 0;
  auto acc = [&](long r) { chunks_seen = std::max(chunks_seen, r / 1000000L), matches += r % 1000000L; };
  for (uint32_t seed = 1; seed <= 40; ++seed) {
    // everything resident in one chunk
    acc(run_case(seed, 40 + seed % 60, 1 + seed % 9, 6, 4, 3, 3, seed % 5 == 0 ? 0.05 : 0.0, 160 << 10, 160 << 10, 160, 300));
    // tight budgets: a few words per chunk, different for the two kernels
    acc(run_case(1000 + seed, 200 + seed * 7, 2 + seed % 17, 8, 5, 4, 4, seed % 4 == 0 ? 0.02 : 0.0, 9000 + 500 * (seed % 7),
                 7000 + 300 * (seed % 5), 16 + 8 * (seed % 20), 120));
  }
  // a program the size of BASELINE configs[4]'s shard: 10k throttles, ~30k terms, 64 namespaces, real LDS budgets
  acc(run_case(77, 10000, 64, 16, 16, 5, 3, 0.0, 120 << 10, 140 << 10, 152, 64));
  // more than 4096 terms with single-namespace classes (128-bit class granularity) and a universe of 1 namespace
  acc(run_case(78, 3000, 1, 10, 8, 3, 2, 0.001, 60 << 10, 60 << 10, 152, 64));
  if (chunks_seen < 8) ++g_fail, fprintf(stderr, "FAIL: the tight budgets never produced a multi-chunk index (%ld)\n", chunks_seen);
  if (matches < 1000) ++g_fail, fprintf(stderr, "FAIL: only %ld matches — the cases are too sparse to mean anything\n", matches);
  if (g_fail) {
    fprintf(stderr, "%d expectation(s) failed\n", g_fail);
    return 1;
  }
  printf("index_sim_test: all expectations held (max %ld chunks, %ld matches; lane-parallel blueprint: %ld decision steps at %.0f %% lane occupancy, today's peel loop: %ld steps at %.0f %%)\n",
         chunks_seen, matches, g_decide_steps, g_decide_steps ? 100.0 * (double)g_busy_lanes / (64.0 * (double)g_decide_steps) : 0.0,
         g_peel_steps, g_peel_steps ? 100.0 * (double)g_peel_busy / (64.0 * (double)g_peel_steps) : 0.0);
  return 0;
}
