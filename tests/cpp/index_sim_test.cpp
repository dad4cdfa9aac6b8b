// index_sim_test — CPU-only check of the selector index builder (kube_throttler_amd/csrc/kt_index.cpp): random selector
// programs are compiled with kt::build_index, then every chunk image is decoded and the scan of kt_scan.h is replayed on
// the host, pod by pod (labels -> atom ids through the index's table, as kt_translate_pods does; advance over the
// namespace's word list with the any / two / three / veto accumulators -> exact match bits -> generic confirmation of
// `slow` shapes -> adjacent-throttle dedup; slow list walked in order).  The matched / errored throttle set of every
// pod must equal the brute-force evaluation of the program (OR of terms in order, AND of requirements,
// throttle_selector.go:30-54, clusterthrottle_selector.go:30-87), for LDS budgets from "one chunk" down to "a few words
// per chunk", for the simple {any} and the rich {any, veto} image form.  Structural invariants of the chunking are
// checked on the way.  No device call is made.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "../../tools/study/kt_anchor.h"
#include "kt_index.h"
#include "../../include/kt_snapshot.h"

using namespace kt;

static int g_fail = 0;
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

// a cell of a chunk image's planes (kt_index.h: any[n_words][col_rows], then veto[n_words][col_rows])
static inline uint64_t cell_any(const BmChunk& ch, const uint64_t* rows, size_t id, size_t w) { return rows[w * ch.col_rows + id]; }
// (the veto plane holds the columns of local words [0, n_veto) and — when there are others — one all-zero column, which is what
//  the device reads for a word without a veto column: BmChunk::zero_col)
static inline uint64_t cell_veto(const BmChunk& ch, const uint64_t* rows, size_t id, size_t w) {
  if (w < ch.n_veto) return rows[((size_t)ch.n_words + w) * ch.col_rows + id];
  return rows[(size_t)ch.zero_col / 8 + id];
}

// kt_translate_pods on the host: the pod's labels as ids of referenced atoms (open-addressing table of the index)
static uint32_t atom_id_of(const HostIndex& ix, uint32_t atom) {
  const uint32_t mask = (uint32_t)ix.atom_table.size() - 1;
  uint32_t s = atom_slot(atom, mask);
  for (;;) {
    const uint64_t e = ix.atom_table[s];
    if (e == 0ull) return 0;
    if ((uint32_t)e == atom) return (uint32_t)(e >> 32);
    s = (s + 1) & mask;
  }
}
static long g_late_atoms = 0, g_placed_atoms = 0;  // atoms that found the home slot of their key taken / all atoms placed
static std::vector<uint32_t> translate(const HostIndex& ix, const PodLabels& pod, bool* overflow) {
  // kt_translate_pods: one atom per label, each to the home slot of its key when that is free, the others to the free slots
  // in label order
  std::vector<uint32_t> ids(ix.la, 0u), late;
  size_t cnt = 0;
  for (size_t l = 0; l < pod.pairs.size(); ++l) {
    uint32_t e = atom_id_of(ix, pod.pairs[l]);
    if (!e && ix.n_key_atoms) e = atom_id_of(ix, kKeyAtom | pod.keys[l]);
    if (!e) continue;
    ++cnt;
    const uint32_t home = (e >> 16) & (ix.la - 1u);
    if (ids[home] == 0u) ids[home] = e & kAtomIdMask;
    else late.push_back(e & kAtomIdMask);
  }
  *overflow = cnt > ix.la;
  EXPECT(cnt <= ix.la, "pod carries %zu relevant atoms, index promised <= %u", cnt, ix.la);
  for (uint32_t id : late)
    for (uint32_t sl = 0; sl < ix.la; ++sl)
      if (ids[sl] == 0u) { ids[sl] = id; break; }
  g_late_atoms += (long)late.size(), g_placed_atoms += (long)cnt;
  return ids;
}

// replays scan_tile (kt_scan.h) for one pod -> per-throttle result (1 match / 2 error), checks "reported once"
static long g_matches_exact = 0, g_slow_confirms = 0, g_word_steps = 0, g_admitted = 0;
static long g_word_veto = 0, g_word_m3 = 0, g_word_plain = 0;  // visited words that hold a veto bit in some row / a need-3 term / neither
// per index: does word w of chunk ci hold a veto bit in ANY row (the rich image's second family)
static const std::vector<std::vector<uint8_t>>& word_veto_flags(const HostIndex& ix) {
  static std::map<const HostIndex*, std::vector<std::vector<uint8_t>>> cache;
  static std::map<const HostIndex*, size_t> stamp;
  auto it = cache.find(&ix);
  if (it != cache.end() && stamp[&ix] == ix.bm_images.size() + ix.bm_chunks.size() * 7919u) return it->second;
  std::vector<std::vector<uint8_t>> f(ix.bm_chunks.size());
  for (size_t ci = 0; ci < ix.bm_chunks.size(); ++ci) {
    const BmChunk& ch = ix.bm_chunks[ci];
    f[ci].assign(ch.n_words, 0);
    if (!ix.rich) continue;
    const uint64_t* rows = (const uint64_t*)(ix.bm_images.data() + ch.img_off);
    for (uint32_t r = 0; r < ix.bm_rows; ++r)
      for (uint32_t w = 0; w < ch.n_words; ++w)
        if (cell_veto(ch, rows, r, w)) f[ci][w] = 1;
  }
  stamp[&ix] = ix.bm_images.size() + ix.bm_chunks.size() * 7919u;
  return cache[&ix] = f;
}
// per index: the keys whose atoms hold a bit (either family) in word w of chunk ci, as a bit mask over key ranks
static long g_word_keys = 0;
static const std::vector<std::vector<uint32_t>>& word_key_masks(const HostIndex& ix) {
  static std::map<const HostIndex*, std::vector<std::vector<uint32_t>>> cache;
  static std::map<const HostIndex*, size_t> stamp;
  auto it = cache.find(&ix);
  if (it != cache.end() && stamp[&ix] == ix.bm_images.size() + ix.bm_chunks.size() * 7919u) return it->second;
  std::map<uint32_t, uint32_t> key_rank;
  for (uint32_t k : ix.atom_key) key_rank.emplace(k, 0u);
  uint32_t nk = 0;
  for (auto& kv : key_rank) kv.second = nk++;
  std::vector<std::vector<uint32_t>> f(ix.bm_chunks.size());
  const size_t fam = ix.rich ? 2 : 1;
  for (size_t ci = 0; ci < ix.bm_chunks.size(); ++ci) {
    const BmChunk& ch = ix.bm_chunks[ci];
    f[ci].assign(ch.n_words, 0u);
    const uint64_t* rows = (const uint64_t*)(ix.bm_images.data() + ch.img_off);
    for (size_t a = 0; a < ix.atoms.size(); ++a) {
      const uint32_t r = ix.atoms[a].id, kr = std::min(31u, key_rank[ix.atom_key[a]]);
      for (uint32_t w = 0; w < ch.n_words; ++w) {
        const uint64_t any = cell_any(ch, rows, r, w) | (fam == 2 ? cell_veto(ch, rows, r, w) : 0ull);
        if (any) f[ci][w] |= 1u << kr;
      }
    }
  }
  stamp[&ix] = ix.bm_images.size() + ix.bm_chunks.size() * 7919u;
  return cache[&ix] = f;
}
static long g_word_useful = 0, g_word_hit = 0;  // visited words in which some atom of the pod (or a term without positive requirement) has an admitted bit / that hold a match
// c0 / c1: the chunks walked (an anchored index: those of the item's block); walk_slow: the slow list too
static std::map<uint32_t, int> scan(const Program& p, const HostIndex& ix, const PodLabels& pod, size_t c0 = 0, size_t c1 = ~(size_t)0,
                                    bool walk_slow = true) {
  std::map<uint32_t, int> out;
  bool overflow = false;
  const std::vector<uint32_t> ids = translate(ix, pod, &overflow);
  for (size_t ci = c0; ci < std::min(c1, ix.bm_chunks.size()); ++ci) {
    const BmChunk& ch = ix.bm_chunks[ci];
    // the namespace rows the chunk's word lists serve (kt_index.h: BmChunk::ns_base / ns_cnt)
    if (pod.ns < ch.ns_base || pod.ns - ch.ns_base >= ch.ns_cnt) continue;
    const uint32_t ns_rel = pod.ns - ch.ns_base;
    const unsigned char* img = ix.bm_images.data() + ch.img_off;
    const uint64_t* rows = (const uint64_t*)img;
    const WordHdr* hdr = (const WordHdr*)(img + ch.off_hdr);
    const uint32_t* nsl_rng = (const uint32_t*)(img + ch.off_nsl_rng);
    const NsWord* nsl = (const NsWord*)(img + ch.off_nsl);
    const uint32_t* term_t = (const uint32_t*)(img + ch.off_term_t);
    const uint16_t* term_rank = (const uint16_t*)(img + ch.off_term_rank);
    const uint32_t* term_g = (const uint32_t*)(img + ch.off_term_g);
    EXPECT(ch.off_term_t == ch.lds_bytes, "LDS part must end where term_t starts");
    uint32_t last_t = ~0u, last_r = ~0u, prev_w = ~0u;
    EXPECT(ch.n_veto <= ch.n_words && (ch.zero_col == 0u) == (!ix.rich || ch.n_veto == ch.n_words), "veto columns of chunk %zu", ci);
    for (uint32_t k = nsl_rng[2 * ns_rel]; k < nsl_rng[2 * ns_rel + 1]; ++k) {
      const uint32_t w = nsl[k].w;
      EXPECT(w < ch.n_words, "word %u of %u", w, ch.n_words);
      EXPECT(prev_w == ~0u || w > prev_w, "word list of ns %u not ascending", pod.ns);
      prev_w = w;
      ++g_word_steps;
      if (getenv("KT_SIM_WORD_FORMS")) {
        const bool wv = (nsl[k].flags & kNsWordVeto) != 0, w3 = hdr[w].m3 != 0;
        EXPECT(!ix.rich || wv == (word_veto_flags(ix)[ci][w] != 0), "veto flag of word %u", w);
        EXPECT(!ix.rich || wv == (w < ch.n_veto), "word %u: veto flag %d, veto column %d", w, (int)wv, (int)(w < ch.n_veto));
        g_word_veto += wv, g_word_m3 += w3, g_word_plain += !wv && !w3;
        g_word_keys += __builtin_popcount(word_key_masks(ix)[ci][w]);
      }
      g_admitted += __builtin_popcountll(nsl[k].mask);
      uint64_t any = hdr[w].univ, two = 0, three = 0, four = 0, five = 0, vet = 0, par = 0;
      for (uint32_t id : ids) {
        EXPECT(id < ix.bm_rows, "row %u of %u", id, ix.bm_rows);
        const uint64_t r = cell_any(ch, rows, id, w);
        if (ix.rich) vet |= cell_veto(ch, rows, id, w);
        five |= four & r;
        four |= three & r;
        three |= two & r;
        two |= any & r;
        any |= r;
        par ^= r;
      }
      EXPECT((hdr[w].m5 & ~hdr[w].m4) == 0 && (hdr[w].m4 & ~hdr[w].m3) == 0 && (hdr[w].m3 & ~hdr[w].m2) == 0, "need masks of word %u are not nested", w);
      EXPECT(ix.max_need > 3 || hdr[w].m4 == 0, "a need-4 term in an index that says max_need %u", ix.max_need);
      // the word's form (NsWord::flags): what scan_tile's cheaper paths rely on
      const uint32_t fl = nsl[k].flags;
      EXPECT(((fl & kNsWordNeed3) != 0u) == (hdr[w].m3 != 0ull), "need-3 flag of word %u", w);
      EXPECT(ix.rich || fl == 0u, "a simple image has one form");
      if (!(fl & kNsWordVeto)) EXPECT(vet == 0ull, "word %u is flagged veto-free, yet a row of the pod holds veto bits", w);
      if (!(fl & kNsWordNeed3)) {
        const uint64_t general = (any & ~hdr[w].m2) | (two & hdr[w].m2), simple = any & ~(par & hdr[w].m2);
        EXPECT(general == simple, "word %u without need-3 terms: OR / XOR accumulation disagrees with the counting form", w);
      }
      if (any & nsl[k].mask) ++g_word_useful;
      uint64_t x = (any & ~hdr[w].m2) | (two & hdr[w].m2);
      x = (x & ~hdr[w].m3) | (three & hdr[w].m3);
      x = (x & ~hdr[w].m4) | (four & hdr[w].m4);
      x = (x & ~hdr[w].m5) | (five & hdr[w].m5);
      if (x & ~vet & nsl[k].mask) ++g_word_hit;
      if (!ix.rich) EXPECT(hdr[w].m3 == 0 && hdr[w].slow == 0, "simple image with need-3 / slow terms");
      x &= ~vet & nsl[k].mask;
      uint64_t sl = x & hdr[w].slow;
      EXPECT(!sl || ch.has_slow, "slow candidate in a chunk without has_slow");
      while (sl) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(sl);
        sl &= sl - 1;
        ++g_slow_confirms;
        if (!term_ok(p, term_g[w * 64 + bit], pod)) x &= ~(1ull << bit);
      }
      while (x) {
        const uint32_t c = w * 64 + (uint32_t)__builtin_ctzll(x);
        x &= x - 1;
        const uint32_t tt = term_t[c];
        EXPECT(tt & kTermReal, "match on a padding term number %u", c);
        const uint32_t t = tt & kTermRowMask;
        const uint32_t r = term_rank[c] & 0x7FFFu;
        EXPECT(((term_rank[c] & kRankAdj) != 0) == ((tt & kTermAdj) != 0), "adj flags of term %u", c);
        EXPECT(r < ch.n_thr && ch.rank0 + r < ix.bm_rank_t.size() && ix.bm_rank_t[ch.rank0 + r] == t, "rank of term number %u", c);
        EXPECT(p.term_thr[term_g[c]] == t, "term_g of number %u", c);
        bool ok = !((tt & kTermAdj) && t == last_t);
        const bool ok_r = !((term_rank[c] & kRankAdj) && r == last_r);  // the aggregate's form of the same rule
        EXPECT(ok == ok_r, "dedup by throttle row and by rank disagree at %u", c);
        if (ok) {
          last_t = t, last_r = r;
          EXPECT(!out.count(t), "throttle %u reported twice", t);
          out[t] = 1;
          ++g_matches_exact;
        }
      }
    }
    // the namespace's word list is exactly the words its admission can touch
    for (uint32_t k = nsl_rng[2 * ns_rel]; k < nsl_rng[2 * ns_rel + 1]; ++k) EXPECT(nsl[k].mask != 0, "empty mask in the word list of ns %u", pod.ns);
  }
  for (uint32_t t : walk_slow ? ix.slow_thr : std::vector<uint32_t>()) {
    const int r = brute(p, t, pod);  // walk_slow_mem IS the in-order walk
    EXPECT(!out.count(t), "slow throttle %u also indexed", t);
    if (r) out[t] = r;
  }
  return out;
}

static uint32_t g_chk_word = kCheckWordLds;  // the check's table bytes per word the index was cut for (file mode: KT_SIM_CHK_WORD)
static void check_structure(const Program& p, const HostIndex& ix, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes) {
  uint32_t w = 0, rank = 0;
  std::set<uint32_t> seen_t;
  EXPECT(ix.cut_thr_bytes != 0 && ix.cut_thr_bytes <= thr_bytes, "record size of the cut: %u (plain: %u)", ix.cut_thr_bytes, thr_bytes);
  thr_bytes = ix.cut_thr_bytes;  // (the packed fold's record size when the caller offered it and the program needs several chunks)
  // (agg_windowed: ONE chunk whose table the aggregate holds a window of ranks at a time — only a minimal table has to fit)
  EXPECT(!ix.agg_windowed || ix.bm_chunks.size() == 1, "a windowed aggregate is for single-chunk programs (%zu chunks)", ix.bm_chunks.size());
  auto fits_one = [&](const BmChunk& ch) {
    const size_t thr = ix.agg_windowed ? std::min<size_t>(ch.n_thr, 256) : ch.n_thr;
    return (size_t)ch.lds_bytes + (size_t)ch.n_words * g_chk_word <= chk_budget &&
           (size_t)ch.lds_bytes + (((size_t)ch.n_words * 128 + 15) & ~(size_t)15) + (size_t)ch.n_words * 16 + thr * thr_bytes + 16 <= agg_budget;
  };
  for (size_t i = 0; i < ix.bm_chunks.size(); ++i) {
    const BmChunk& ch = ix.bm_chunks[i];
    EXPECT(ch.w0 == w && ch.n_words >= 1, "chunk %zu starts at word %u, expected %u", i, ch.w0, w);
    EXPECT(ch.col_rows == image_col_rows(ix.bm_rows) && ch.img_off % 16 == 0 && ch.img_bytes % 16 == 0 && ch.lds_bytes % 16 == 0, "chunk %zu layout", i);
    EXPECT(ch.img_off + (size_t)ch.img_bytes <= ix.bm_images.size(), "chunk %zu image range", i);
    EXPECT(ch.n_thr < 0x8000u, "chunk %zu: %u throttles do not fit the 15-bit rank", i, ch.n_thr);
    if (ch.n_thr) {
      EXPECT(ch.rank0 == rank, "chunk %zu rank0 %u, expected %u (a throttle straddles chunks?)", i, ch.rank0, rank);
      rank = ch.rank0 + ch.n_thr;
    }
    // a multi-word chunk respects both budgets (a single unsplittable stretch may not: the launchers refuse it)
    if (!fits_one(ch) && ch.n_words > 1) {
      // over budget: then no earlier cut was possible — every inner word boundary splits a throttle
      const unsigned char* img = ix.bm_images.data() + ch.img_off;
      const uint32_t* term_t = (const uint32_t*)(img + ch.off_term_t);
      const uint16_t* term_rank = (const uint16_t*)(img + ch.off_term_rank);
      for (uint32_t wb = 1; wb < ch.n_words; ++wb) {
        int64_t a = -1, b = -1;
        for (int k = 63; k >= 0 && a < 0; --k)
          if (term_t[(wb - 1) * 64 + k] & kTermReal) a = (int64_t)(wb - 1) * 64 + k;
        for (int k = 0; k < 64 && b < 0; ++k)
          if (term_t[wb * 64 + k] & kTermReal) b = (int64_t)wb * 64 + k;
        EXPECT(a >= 0 && b >= 0 && (term_rank[a] & 0x7FFF) == (term_rank[b] & 0x7FFF),
               "chunk %zu exceeds the budget although it could have been cut at word %u", i, wb);
      }
    }
    EXPECT(ch.lds_bytes <= ix.bm_max_lds && ch.n_thr <= ix.bm_max_thr && ch.n_words <= ix.bm_max_words, "chunk %zu maxima", i);
    w += ch.n_words;
  }
  // the launchers size LDS from the maxima over all chunks: the combination of maxima has to fit too
  bool all_fit = true;
  for (const BmChunk& ch : ix.bm_chunks) all_fit &= fits_one(ch);
  if (all_fit) {
    EXPECT((size_t)ix.bm_max_lds + (size_t)ix.bm_max_words * g_chk_word <= chk_budget, "maxima exceed the check budget");
    EXPECT((size_t)ix.bm_max_lds + (((size_t)ix.bm_max_words * 128 + 15) & ~(size_t)15) + (size_t)ix.bm_max_words * 16 +
               (ix.agg_windowed ? std::min<size_t>(ix.bm_max_thr, 256) : (size_t)ix.bm_max_thr) * thr_bytes + 16 <= agg_budget,
           "maxima exceed the aggregate budget");
  }
  EXPECT(w == ix.img_words && w >= ix.bm_words, "chunk images hold %u words, index says %u (program: %u)", w, ix.img_words, ix.bm_words);
  EXPECT(ix.cut_grouped || w == ix.bm_words, "the global plan copies no word: %u of %u", w, ix.bm_words);
  EXPECT(rank == ix.bm_rank_t.size(), "ranks cover %u of %zu", rank, ix.bm_rank_t.size());
  for (uint32_t t : ix.bm_rank_t) {
    // a throttle has one rank per GROUP (namespace cell, kt_index.cpp): that a pod never meets two of them is what
    // scan()'s "reported twice" expectation pins
    seen_t.insert(t);
    EXPECT(p.thr[t].live, "dead throttle %u indexed", t);
  }
  for (uint32_t t : ix.slow_thr) EXPECT(!seen_t.count(t) && p.thr[t].live, "slow throttle %u", t);
  // the slow list is for unconvertible pod selectors and for throttles beyond kMaxIndexedTerms; a throttle with 65 .. 512 terms is
  // indexed as one run of numbers across words (round 6) and flags the index has_long
  bool any_long = false;
  for (uint32_t t : ix.slow_thr) {
    bool invalid = false;
    for (uint32_t g = p.thr_term_off[t]; g < p.thr_term_off[t + 1]; ++g) invalid |= (p.term_flags[g] & KT_TERM_POD_SEL_INVALID) != 0;
    EXPECT(invalid || p.thr_term_off[t + 1] - p.thr_term_off[t] > kMaxIndexedTerms, "throttle %u with %u convertible terms is on the slow list", t,
           p.thr_term_off[t + 1] - p.thr_term_off[t]);
  }
  for (uint32_t t : seen_t) any_long |= p.thr_term_off[t + 1] - p.thr_term_off[t] > 64u;
  EXPECT(!any_long || ix.has_long, "an indexed throttle has more than 64 terms, yet the index does not say has_long");
  // atom ids are dense, unique and fit 16 bits; the table finds every one of them
  EXPECT(ix.atoms.size() + 1 == ix.bm_rows && ix.bm_rows <= 65536, "atom ids");
  for (const AtomId& a : ix.atoms) EXPECT(atom_id_of(ix, a.atom) == (a.id | a.home << 16) && a.id >= 1 && a.id < ix.bm_rows && a.home < ix.la, "atom %u", a.atom);
  EXPECT(ix.rich == (ix.has_veto || ix.has_slow || ix.max_need > 2 || ix.la != 8) && (ix.la == 8 || ix.la == 16 || ix.la == 32), "rich flag / atom slots");
}

// fingerprint of everything the device gets of an index: two builders that claim the same output must agree bit for bit
// (how a faster index build is accepted: same fingerprints on the dumped programs and on every random case below)
static uint64_t index_fingerprint(const HostIndex& ix, uint64_t h) {
  auto mix = [&](const void* d, size_t n) {
    const unsigned char* b = (const unsigned char*)d;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  };
  mix(ix.bm_images.data(), ix.bm_images.size());
  mix(ix.bm_chunks.data(), ix.bm_chunks.size() * sizeof(BmChunk));
  mix(ix.bm_rank_t.data(), ix.bm_rank_t.size() * 4);
  mix(ix.bm_chunk_ns.data(), ix.bm_chunk_ns.size() * 4);
  mix(ix.atom_table.data(), ix.atom_table.size() * 8);
  mix(ix.slow_thr.data(), ix.slow_thr.size() * 4);
  return h;
}
static uint64_t g_fingerprint = 1469598103934665603ull;  // over all random cases of a run

static long run_case(uint32_t seed, uint32_t T, uint32_t n_ns, uint32_t K, uint32_t V, int max_terms, int max_reqs, double p_bad,
                     uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int n_pods, int positive_only = 0, uint32_t max_labels = 0) {
  // max_labels: the pods carry at most this many of the K keys (0: any number) — with more keys than atom slots two keys
  // share a home slot and kt_translate_pods' second pass places what found its home taken
  std::mt19937 rng(seed);
  Program p = random_program(rng, T, n_ns, K, V, max_terms, max_reqs, p_bad);
  if (positive_only)  // matchLabels-style programs (the simple instantiation): every requirement becomes In
    for (size_t r = 0; r < p.req_op.size(); ++r) {
      if (p.req_op[r] == KT_OP_NOT_IN) p.req_op[r] = KT_OP_IN;
      if (p.req_op[r] == KT_OP_EXISTS || p.req_op[r] == KT_OP_DOES_NOT_EXIST) {
        p.req_op[r] = KT_OP_IN;  // value list is empty: In{} never matches, exercised on purpose
      }
    }
  HostIndex ix;
  build_index(ix, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
              [&](uint32_t t) { return p.thr[t]; }, n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)(max_labels ? max_labels : K));
  check_structure(p, ix, agg_budget, chk_budget, thr_bytes);
  g_fingerprint = index_fingerprint(ix, g_fingerprint);
  long matches = 0;
  for (int i = 0; i < n_pods; ++i) {
    PodLabels pod;
    pod.ns = rng() % n_ns;
    for (uint32_t k = 1; k <= K; ++k)
      if (rng() % 100 < (max_labels ? 75u : 55u) && (!max_labels || pod.keys.size() < max_labels)) pod.keys.push_back(k), pod.pairs.push_back(p.pair(k, rng() % V));
    const std::map<uint32_t, int> got = scan(p, ix, pod);
    for (uint32_t t = 0; t < T; ++t) {
      const int want = brute(p, t, pod);
      const auto it = got.find(t);
      const int have = it == got.end() ? 0 : it->second;
      EXPECT(have == want, "seed %u pod %d throttle %u: index says %d, program says %d", seed, i, t, have, want);
      matches += want == 1;
    }
  }
  const size_t chunks_first = ix.bm_chunks.size();
  const int64_t visits_first = ix.ns_chunk_visits;
  {  // cut_chunks(): the same numbering cut again for other budgets (what the engine does when the half-LDS cut needs
     // several chunks) must describe the same matches
    const uint32_t agg2 = agg_budget * 2, chk2 = chk_budget * 2;
    cut_chunks(ix, agg2, chk2, thr_bytes);
    check_structure(p, ix, agg2, chk2, thr_bytes);
    std::mt19937 rng2(seed ^ 0x5bd1e995u);
    for (int i = 0; i < n_pods / 4 + 1; ++i) {
      PodLabels pod;
      pod.ns = rng2() % n_ns;
      for (uint32_t k = 1; k <= K; ++k)
        if (rng2() % 100 < (max_labels ? 75u : 55u) && (!max_labels || pod.keys.size() < max_labels)) pod.keys.push_back(k), pod.pairs.push_back(p.pair(k, rng2() % V));
      const std::map<uint32_t, int> got = scan(p, ix, pod);
      for (uint32_t t = 0; t < T; ++t) {
        const auto it = got.find(t);
        EXPECT((it == got.end() ? 0 : it->second) == brute(p, t, pod), "seed %u re-cut pod %d throttle %u", seed, i, t);
      }
    }
    // (larger budgets never cost a namespace-ordered scan chunk passes; the grouped plan may well have MORE chunks than the
    //  global one it replaces — what it lowers are the chunks per namespace, and it is only taken when it saves a fifth of them)
    EXPECT(ix.ns_chunk_visits * 4 <= visits_first * 5 + 4, "doubling the budgets gave %lld chunk visits over the namespaces instead of %lld", (long long)ix.ns_chunk_visits,
           (long long)visits_first);
    EXPECT(ix.cut_grouped || ix.bm_chunks.size() <= chunks_first, "doubling the budgets gave %zu chunks instead of %zu", ix.bm_chunks.size(), chunks_first);
  }
  return (long)chunks_first * 1000000L + matches % 1000000L + (ix.rich ? 0 : 500000000L);
}

// ---- anchored mode (groundwork of the inverted scan, tools/study/kt_anchor.h): the program split by anchor
//      atom, one index per anchor built by the SAME kt::build_index, a pod walked through the sub-indexes of anchor 0 and of
//      the pairs it carries — the union must be exactly the brute-force result of the ORIGINAL program, every throttle
//      reported once.  Returns word visits over all pods (g_word_steps delta).
struct AnchoredIndexes {
  std::vector<AnchorSubProgram> subs;
  std::vector<Program> progs;  // the sub-programs as Programs (scan() confirms slow shapes / walks slow throttles through them)
  std::vector<HostIndex> ix;
  std::map<uint32_t, size_t> by_anchor;
  AnchorSplitStats stats;
  size_t words = 0, chunks = 0;
};
static void build_anchored(const Program& p, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int max_labels, AnchoredIndexes& A) {
  const size_t T = p.thr.size();
  std::vector<uint8_t> anchorable(T, 0);
  for (size_t t = 0; t < T; ++t) {
    bool ok = p.thr[t].live && p.thr_term_off[t + 1] - p.thr_term_off[t] <= 64u;
    for (uint32_t g = p.thr_term_off[t]; g < p.thr_term_off[t + 1] && ok; ++g) ok = !(p.term_flags[g] & KT_TERM_POD_SEL_INVALID);
    anchorable[t] = ok;
  }
  A.subs = anchor_split(p.thr_term_off, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val, anchorable, &A.stats);
  A.progs.resize(A.subs.size()), A.ix.resize(A.subs.size());
  for (size_t k = 0; k < A.subs.size(); ++k) {
    const AnchorSubProgram& sp = A.subs[k];
    Program& q = A.progs[k];
    q.thr_term_off = sp.thr_term_off, q.term_flags = sp.term_flags, q.term_req_off = sp.term_req_off, q.req_op = sp.req_op;
    q.req_key = sp.req_key, q.req_val_off = sp.req_val_off, q.req_val = sp.req_val;
    q.n_ns = p.n_ns, q.K = p.K, q.V = p.V;
    for (uint32_t t : sp.thr_orig) q.thr.push_back(p.thr[t]);
    const uint32_t G = (uint32_t)sp.term_orig.size();
    for (size_t v = 0; v + 1 < sp.thr_term_off.size(); ++v)
      for (uint32_t g = sp.thr_term_off[v]; g < sp.thr_term_off[v + 1]; ++g) q.term_thr.push_back((uint32_t)v);
    q.gw = (G + 31) / 32 + 1;
    q.ns_term_ok.assign((size_t)p.n_ns * q.gw, 0u);
    for (uint32_t n = 0; n < p.n_ns; ++n)
      for (uint32_t g = 0; g < G; ++g)
        if (ns_ok(p, sp.term_orig[g], n)) q.ns_term_ok[(size_t)n * q.gw + (g >> 5)] |= 1u << (g & 31);
    build_index(A.ix[k], q.thr_term_off, q.term_thr, q.term_flags, q.term_req_off, q.req_op, q.req_key, q.req_val_off, q.req_val,
                [&](uint32_t t) { return q.thr[t]; }, p.n_ns, q.ns_term_ok, q.gw, agg_budget, chk_budget, thr_bytes, max_labels);
    check_structure(q, A.ix[k], agg_budget, chk_budget, thr_bytes);
    A.by_anchor[sp.anchor] = k;
    A.words += A.ix[k].bm_words, A.chunks += A.ix[k].bm_chunks.size();
  }
}
// the walk of ONE pod: anchor 0 + the anchors it carries; out: original throttle -> 1 (match) / 2 (error)
static std::map<uint32_t, int> scan_anchored(const AnchoredIndexes& A, const PodLabels& pod) {
  std::map<uint32_t, int> out;
  auto walk = [&](uint32_t anchor) {
    const auto it = A.by_anchor.find(anchor);
    if (it == A.by_anchor.end()) return;
    const size_t k = it->second;
    for (const auto& kv : scan(A.progs[k], A.ix[k], pod)) {
      const uint32_t t = A.subs[k].thr_orig[kv.first];
      EXPECT(!out.count(t), "throttle %u reported by two anchors (second: %u)", t, anchor);
      out[t] = kv.second;
    }
  };
  walk(0u);
  for (uint32_t pr : pod.pairs) walk(pr);
  return out;
}
// the same walk through the CONCATENATED anchored index (kt_anchor.h: build_anchored_index — the structure the device will
// get): an item (pod, block) is a virtual pod of namespace block * n_ns + pod.ns walking the block's chunks
static long g_items = 0;
static std::map<uint32_t, int> scan_concat(const Program& p, const AnchoredIndex& AX, const std::map<uint32_t, uint32_t>& block_of, const PodLabels& pod) {
  std::map<uint32_t, int> out;
  // the pod's items as the device finds them: from its translated atom row through AnchoredIndex::atom_block — the same
  // blocks as "the anchors among the pairs it carries"
  {
    bool overflow = false;
    std::set<uint32_t> by_row, by_pair;
    for (uint32_t id : translate(AX.ix, pod, &overflow))
      if (id && AX.atom_block[id]) by_row.insert(AX.atom_block[id]);
    for (uint32_t pr : pod.pairs) {
      const auto it = block_of.find(pr);
      if (it != block_of.end()) by_pair.insert(it->second);
    }
    EXPECT(by_row == by_pair, "items through the atom row (%zu) and through the pairs (%zu) differ", by_row.size(), by_pair.size());
    g_items += 1 + (long)by_row.size();
  }
  auto walk = [&](uint32_t c) {
    PodLabels item = pod;
    item.ns = c * AX.n_ns + pod.ns;
    PodLabels real = pod;  // (slow confirmations / the slow list evaluate the ORIGINAL program: the pod's real namespace)
    (void)real;
    for (const auto& kv : scan(p, AX.ix, item, AX.blk_chunk0[c], AX.blk_chunk0[c + 1], /*walk_slow=*/false)) {
      EXPECT(!out.count(kv.first), "throttle %u reported by two blocks (second: %u)", kv.first, c);
      out[kv.first] = kv.second;
    }
  };
  walk(0u);
  for (uint32_t pr : pod.pairs) {
    const auto it = block_of.find(pr);
    if (it != block_of.end()) walk(it->second);
  }
  for (uint32_t t : AX.ix.slow_thr) {  // walked with block 0's item, in the pod's real namespace
    const int r = brute(p, t, pod);
    EXPECT(!out.count(t), "slow throttle %u also indexed", t);
    if (r) out[t] = r;
  }
  return out;
}
static long anchored_case(uint32_t seed, uint32_t T, uint32_t n_ns, uint32_t K, uint32_t V, int max_terms, int max_reqs, double p_bad,
                          uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, int n_pods, long* visits_classic, long* visits_anchored) {
  std::mt19937 rng(seed);
  Program p = random_program(rng, T, n_ns, K, V, max_terms, max_reqs, p_bad);
  HostIndex classic;
  build_index(classic, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
              [&](uint32_t t) { return p.thr[t]; }, n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)K);
  AnchoredIndexes A;
  build_anchored(p, agg_budget, chk_budget, thr_bytes, (int)K, A);
  AnchoredIndex AX;
  build_anchored_index(AX, p.thr_term_off, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
                       [&](uint32_t t) { return p.thr[t]; }, n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)K, kCheckWordLds, classic);
  std::map<uint32_t, uint32_t> block_of;
  for (uint32_t c = 1; c < AX.block_anchor.size(); ++c) block_of[AX.block_anchor[c]] = c;
  long matches = 0;
  for (int i = 0; i < n_pods; ++i) {
    PodLabels pod;
    pod.ns = rng() % n_ns;
    for (uint32_t k = 1; k <= K; ++k)
      if (rng() % 100 < 55) pod.keys.push_back(k), pod.pairs.push_back(p.pair(k, rng() % V));
    long w0 = g_word_steps;
    (void)scan(p, classic, pod);
    *visits_classic += g_word_steps - w0, w0 = g_word_steps;
    const std::map<uint32_t, int> got = scan_anchored(A, pod);
    *visits_anchored += g_word_steps - w0;
    const std::map<uint32_t, int> got2 = scan_concat(p, AX, block_of, pod);
    EXPECT(got2 == got, "anchored: seed %u pod %d: the concatenated index and the per-anchor indexes disagree (%zu vs %zu)", seed, i, got2.size(), got.size());
    for (uint32_t t = 0; t < T; ++t) {
      const int want = brute(p, t, pod);
      const auto it = got.find(t);
      const int have = it == got.end() ? 0 : it->second;
      EXPECT(have == want, "anchored: seed %u pod %d throttle %u: sub-indexes say %d, program says %d", seed, i, t, have, want);
      matches += want == 1;
    }
  }
  return matches;
}


// ---- LDS bank model of the atom-row gathers (KT_SIM_BANKS): the passes one 64-bit ds_read takes for a group of 32 lanes =
// the largest number of DISTINCT addresses that share one of the 32 bank slots of 8 bytes (lanes with equal addresses
// are one broadcast; lanes that do not advance read an always-valid address — word 0 — like the kernel's).  The cell of
// (row id, word w) is cell w * col_rows + id of its plane; a word with veto bits takes a second read from the veto plane.
static long gather_passes(const std::vector<std::vector<uint32_t>>& ids, const std::vector<uint32_t>& ww, uint32_t col_rows) {
  long passes = 0;
  const size_t la = ids[0].size();
  for (size_t a = 0; a < la; ++a)
    for (uint32_t g0 = 0; g0 < 64; g0 += 32) {
      std::map<uint32_t, std::set<uint32_t>> by_slot;
      for (uint32_t l = g0; l < g0 + 32; ++l) {
        const uint32_t cell = ww[l] * col_rows + ids[l][a];
        by_slot[cell % 32u].insert(cell);
      }
      size_t mx = 1;
      for (auto& kv : by_slot) mx = std::max(mx, kv.second.size());
      passes += (long)mx;
    }
  return passes;
}

// The atoms of every pod of a 32-lane group in the order a balancing pass would leave them (KT_SIM_BALANCE=mode):
// lane by lane; an atom some earlier lane of the group carries goes to the slot it has there when that is free here (the
// read is one broadcast), the others to the free slot where their bank class (id mod 32) has the fewest distinct atoms.
//   mode 1: exact knowledge of where an atom sits   2: only the LAST atom placed per (slot, class) is remembered
//   mode 3: mode 1 + one more sweep that takes every lane out and places it again
static std::vector<std::vector<uint32_t>> balance_atom_slots(const std::vector<std::vector<uint32_t>>& ids, int mode) {
  std::vector<std::vector<uint32_t>> out = ids;
  const size_t la = ids[0].size();
  for (uint32_t g0 = 0; g0 < 64; g0 += 32) {
    std::vector<std::vector<std::map<uint32_t, int>>> cell(la, std::vector<std::map<uint32_t, int>>(32));  // (slot, class): id -> lanes
    std::vector<std::vector<uint32_t>> last(la, std::vector<uint32_t>(32, 0u));
    std::vector<std::vector<uint32_t>> cnt(la, std::vector<uint32_t>(32, 0u));
    auto place = [&](uint32_t l) {
      const std::vector<uint32_t> mine = out[l];
      std::vector<uint32_t> put(la, 0u);
      std::vector<uint8_t> used(la, 0), done(mine.size(), 0);
      for (size_t i = 0; i < mine.size(); ++i) {  // broadcasts first
        const uint32_t id = mine[i], c = id % 32u;
        if (!id) { done[i] = 1; continue; }
        for (size_t sl = 0; sl < la; ++sl) {
          const bool there = mode == 2 ? last[sl][c] == id : cell[sl][c].count(id) != 0;
          if (!used[sl] && there) { used[sl] = 1, put[sl] = id, done[i] = 1; break; }
        }
      }
      for (size_t i = 0; i < mine.size(); ++i) {
        if (done[i]) continue;
        const uint32_t id = mine[i], c = id % 32u;
        size_t best = la;
        for (size_t sl = 0; sl < la; ++sl)
          if (!used[sl] && (best == la || cnt[sl][c] < cnt[best][c])) best = sl;
        used[best] = 1, put[best] = id;
      }
      for (size_t sl = 0; sl < la; ++sl) {
        const uint32_t id = put[sl], c = id % 32u;
        if (!id) continue;
        if (mode == 2) { if (last[sl][c] != id) ++cnt[sl][c]; last[sl][c] = id; }
        else if (cell[sl][c][id]++ == 0) ++cnt[sl][c];
      }
      out[l] = put;
    };
    for (uint32_t l = g0; l < g0 + 32; ++l) place(l);
    if (mode == 3)
      for (uint32_t l = g0; l < g0 + 32; ++l) {
        for (size_t sl = 0; sl < la; ++sl) {
          const uint32_t id = out[l][sl], c = id % 32u;
          if (id && --cell[sl][c][id] == 0) cell[sl][c].erase(id), --cnt[sl][c];
        }
        place(l);
      }
  }
  return out;
}

// ---- file mode: the REAL selector program of a BASELINE config + a pod sample (tools/dump_program.py)
static std::vector<uint32_t> read_array(FILE* fh) {
  uint32_t n = 0;
  if (fread(&n, 4, 1, fh) != 1) return {};
  std::vector<uint32_t> v(n);
  if (n && fread(v.data(), 4, n, fh) != n) v.clear();
  return v;
}
static int run_file(const char* path, uint32_t chk_budget) {
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
  uint32_t L = 1;
  for (size_t i = 0; i + 1 < loff.size(); ++i) L = std::max(L, loff[i + 1] - loff[i]);
  HostIndex ix;
  const uint32_t thr_bytes = getenv("KT_SIM_THR_BYTES") ? (uint32_t)atoi(getenv("KT_SIM_THR_BYTES")) : 8 * D + 8;  // (a 16-dimension engine: 136)
  const uint32_t agg_budget = 160u * 1024u - aggregate_fixed_lds();
  // the per-term admission sets as the engine hands them over (it caches them per throttle): [term][namespace words]
  const uint32_t nsw = (NS + 31) / 32;
  const size_t G = p.term_thr.size();
  std::vector<uint32_t> adm_all(G * nsw, 0u);
  for (uint32_t n = 0; n < NS; ++n)
    for (size_t g = 0; g < G; ++g)
      if ((p.ns_term_ok[(size_t)n * p.gw + (g >> 5)] >> (g & 31)) & 1u) adm_all[g * nsw + (n >> 5)] |= 1u << (n & 31);
  const int reps = getenv("KT_SIM_BUILD_REPS") ? atoi(getenv("KT_SIM_BUILD_REPS")) : 1;
  if (getenv("KT_SIM_CHK_WORD")) g_chk_word = (uint32_t)atoi(getenv("KT_SIM_CHK_WORD"));
  for (int rep = 0; rep < reps; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    build_index(ix, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
                [&](uint32_t t) { return p.thr[t]; }, NS, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)L, &adm_all, 0u,
                getenv("KT_SIM_CHK_WORD") ? (uint32_t)atoi(getenv("KT_SIM_CHK_WORD")) : kCheckWordLds, nullptr,
                getenv("KT_SIM_PACKED") ? (uint32_t)atoi(getenv("KT_SIM_PACKED")) : 0u);  // (the engine: check_word_lds(D) = 816 at D <= 8, 40-byte packed records)
    if (reps > 1) fprintf(stderr, "build_index: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  check_structure(p, ix, agg_budget, chk_budget, thr_bytes);
  printf("  index fingerprint %016llx\n", (unsigned long long)index_fingerprint(ix, 1469598103934665603ull));
  long matches = 0, pods = 0;
  // wave-level step counts of scan_tile on full 64-pod tiles: advance rounds and peel steps (max over lanes per round)
  long adv_rounds = 0, peel_steps = 0, peel_busy = 0;
  std::vector<PodLabels> tile;
  // KT_SIM_SORT=1: the tiles are cut from the pods ordered by namespace (the engine's scan order for multi-chunk programs)
  std::vector<size_t> pod_order(pod_ns.size());
  for (size_t i = 0; i < pod_order.size(); ++i) pod_order[i] = i;
  if (getenv("KT_SIM_SORT")) std::stable_sort(pod_order.begin(), pod_order.end(), [&](size_t a, size_t b) { return pod_ns[a] < pod_ns[b]; });
  long chunk_visits = 0, atoms_sum = 0, atoms_tile_max = 0, atoms_tiles = 0;
  // KT_SIM_BANKS=1: LDS passes of the atom-row gathers (gather_passes)
  const bool sim_banks = getenv("KT_SIM_BANKS") != nullptr;
  long bank_instr = 0, bank_pass_now = 0, bank_rounds = 0;
  for (size_t oi = 0; oi < pod_order.size(); ++oi) {
    const size_t i = pod_order[oi];
    PodLabels pod;
    pod.ns = pod_ns[i];
    for (uint32_t j = loff[i]; j < loff[i + 1]; ++j) pod.keys.push_back(lkey[j]), pod.pairs.push_back(lpair[j]);
    const auto one = scan(p, ix, pod);
    if (pods % 16 == 0)  // brute force (T x terms per pod) on a sample
      for (uint32_t t = 0; t < T; ++t) {
        const int want = brute(p, t, pod);
        const auto b = one.find(t);
        EXPECT((b == one.end() ? 0 : b->second) == want, "pod %ld throttle %u", pods, t);
      }
    matches += (long)one.size();
    ++pods;
    tile.push_back(pod);
    if (tile.size() == 64) {
      for (const BmChunk& ch : ix.bm_chunks) {
        const unsigned char* img = ix.bm_images.data() + ch.img_off;
        const uint64_t* rows = (const uint64_t*)img;
        const WordHdr* hd = (const WordHdr*)(img + ch.off_hdr);
        const uint32_t* nsl_rng = (const uint32_t*)(img + ch.off_nsl_rng);
        const NsWord* nsl = (const NsWord*)(img + ch.off_nsl);
        std::vector<uint32_t> k(64), k1(64);
        std::vector<uint64_t> x(64, 0);
        std::vector<std::vector<uint32_t>> ids(64);
        for (int l = 0; l < 64; ++l) {
          bool ov;
          ids[l] = translate(ix, tile[l], &ov);
          k[l] = nsl_rng[2 * tile[l].ns], k1[l] = nsl_rng[2 * tile[l].ns + 1];
        }
        if (&ch == &ix.bm_chunks[0]) {  // atoms the pods of a tile carry (labels no selector refers to have none)
          size_t mx = 0;
          for (int l = 0; l < 64; ++l) atoms_sum += (long)ids[l].size(), mx = std::max(mx, ids[l].size());
          atoms_tile_max += (long)mx, ++atoms_tiles;
        }
        {
          bool any_words = false;
          for (int l = 0; l < 64; ++l) any_words |= k[l] < k1[l];
          chunk_visits += any_words;
        }
        if (sim_banks && getenv("KT_SIM_BALANCE")) ids = balance_atom_slots(ids, atoi(getenv("KT_SIM_BALANCE")));
        for (;;) {
          long has = 0;
          for (int l = 0; l < 64; ++l) has += x[l] != 0;
          if (has) {
            ++peel_steps, peel_busy += has;
            for (int l = 0; l < 64; ++l) x[l] &= x[l] - 1;
            continue;
          }
          bool adv = false;
          if (sim_banks) {
            std::vector<uint32_t> ww(64);
            bool any_on = false, veto_word = false;
            for (int l = 0; l < 64; ++l) {
              const bool on = k[l] < k1[l];
              ww[l] = on ? nsl[k[l]].w : 0u;
              any_on |= on, veto_word |= on && (nsl[k[l]].flags & kNsWordVeto) != 0u;
            }
            if (any_on) {
              const long planes = ix.rich && veto_word ? 2 : 1;
              bank_instr += (long)ix.la * 2 * planes;
              bank_pass_now += planes * gather_passes(ids, ww, ch.col_rows);
              if (getenv("KT_SIM_BANKS_DUMP") && bank_rounds == 0)
                for (int l = 0; l < 32; ++l) {
                  fprintf(stderr, "lane %2d w %2u:", l, ww[l]);
                  for (uint32_t id : ids[l]) fprintf(stderr, " %3u(%2u)", id, id % 32u);
                  fprintf(stderr, "\n");
                }
              ++bank_rounds;
            }
          }
          for (int l = 0; l < 64; ++l)
            if (k[l] < k1[l]) {
              const uint32_t w = nsl[k[l]].w;
              uint64_t any = hd[w].univ, two = 0, three = 0, four = 0, five = 0, vet = 0;
              for (uint32_t id : ids[l]) {
                const uint64_t r = cell_any(ch, rows, id, w);
                if (ix.rich) vet |= cell_veto(ch, rows, id, w);
                five |= four & r, four |= three & r, three |= two & r, two |= any & r, any |= r;
              }
              uint64_t xx = (any & ~hd[w].m2) | (two & hd[w].m2);
              xx = (xx & ~hd[w].m3) | (three & hd[w].m3);
              xx = (xx & ~hd[w].m4) | (four & hd[w].m4);
              xx = (xx & ~hd[w].m5) | (five & hd[w].m5);
              x[l] = xx & ~vet & nsl[k[l]].mask;
              ++k[l];
              adv = true;
            }
          if (!adv) break;
          ++adv_rounds;
        }
      }
      tile.clear();
    }
  }
  const long tiles = pods / 64;
  printf("%s: %u throttles, %zu terms, %u namespaces, L=%u -> %zu chunks (largest LDS part %u B, max %u throttles / %u words per chunk), %zu slow throttles\n",
         path, T, p.term_thr.size(), NS, L, ix.bm_chunks.size(), ix.bm_max_lds, ix.bm_max_thr, ix.bm_max_words, ix.slow_thr.size());
  printf("  %u atoms, %u words, form: %s (veto %d, max need %u, slow terms %d), %u atom slots per pod\n", (uint32_t)ix.atoms.size(), ix.bm_words,
         ix.rich ? "rich {any, veto}" : "simple {any}", (int)ix.has_veto, ix.max_need, (int)ix.has_slow, ix.la);
  printf("  LDS: check %u B (image part + term info), aggregate %u B (image part + ranks + table)\n",
         ix.bm_max_lds + ix.bm_max_words * kCheckWordLds + check_fixed_lds(), ix.bm_max_lds + ix.bm_max_words * 128 + ix.bm_max_thr * thr_bytes);
  printf("  %ld pods: %.2f matches per pod (exact, no candidates), %.2f word steps per pod (%.1f admitted term copies per visited word), %ld slow confirmations\n", pods,
         (double)matches / (double)pods, (double)g_word_steps / (double)pods, (double)g_admitted / (double)std::max(1L, g_word_steps), g_slow_confirms);
  if (getenv("KT_SIM_WORD_FORMS"))
    printf("  visited words by form: %.1f %% hold a veto bit in some row, %.1f %% a term with three positive keys, %.1f %% neither\n",
           100.0 * g_word_veto / std::max(1L, g_word_steps), 100.0 * g_word_m3 / std::max(1L, g_word_steps), 100.0 * g_word_plain / std::max(1L, g_word_steps));
  if (getenv("KT_SIM_WORD_FORMS"))
    printf("  keys with a bit in a visited word: %.2f on average (a pod reads one atom row per key it carries: %u atom slots)\n",
           (double)g_word_keys / std::max(1L, g_word_steps), ix.la);
  if (tiles)
    printf("  per 64-pod tile (all chunks): %.1f advance rounds, %.1f peel steps at %.1f %% busy lanes\n", (double)adv_rounds / tiles,
           (double)peel_steps / tiles, peel_steps ? 100.0 * (double)peel_busy / (64.0 * (double)peel_steps) : 0.0);
  if (tiles) printf("  chunks with any word for a tile: %.1f of %zu\n", (double)chunk_visits / tiles, ix.bm_chunks.size());
  {  // chunk passes of a namespace-ordered scan: the chunks that hold a word list of the namespace
    std::vector<uint32_t> per_ns(NS, 0u);
    for (const BmChunk& ch : ix.bm_chunks) {
      const uint32_t* nsl_rng = (const uint32_t*)(ix.bm_images.data() + ch.img_off + ch.off_nsl_rng);
      for (uint32_t n = 0; n < NS; ++n) per_ns[n] += nsl_rng[2 * n + 1] > nsl_rng[2 * n];
    }
    uint64_t sum = 0;
    uint32_t mx = 0, used = 0;
    for (uint32_t n = 0; n < NS; ++n) sum += per_ns[n], mx = std::max(mx, per_ns[n]), used += per_ns[n] != 0;
    printf("  %s plan: %zu chunks over %u image words (program: %u); chunks per namespace: %.2f on average, %u at most; images %.1f MB\n",
           ix.cut_grouped ? "grouped" : "global", ix.bm_chunks.size(), ix.img_words, ix.bm_words, used ? (double)sum / used : 0.0, mx, ix.bm_images.size() / 1048576.0);
  }
  if (atoms_tiles) printf("  atoms per pod: %.2f (largest of a tile: %.2f)\n", (double)atoms_sum / (64.0 * atoms_tiles), (double)atoms_tile_max / atoms_tiles);
  if (sim_banks && bank_instr)
    printf("  LDS passes of the atom-row gathers: %.2f per instruction and 32 lanes, %.1f per advance round of a tile (conflict-free: %.1f)\n",
           (double)bank_pass_now / bank_instr, (double)bank_pass_now / bank_rounds, (double)bank_instr / bank_rounds);
  if (g_fail) fprintf(stderr, "%d expectation(s) failed\n", g_fail);
  return g_fail ? 1 : 0;
}

// ---- packed request words (PackPlan, kt_index.h): the host mirror of write_view_record (kt_kernels.hip), the fold of
// the aggregate kernel (whole-word adds per matching pod) and packed_record_sums / packed_field (kt_index_device.h:
// whole words summed over up to 256 slabs class by class, the fields cut out of the sums afterwards).  For random request
// populations: fields must not overlap or cross a word, a slab of n_slab pods carrying the maxima must not carry out of
// its field, 256 such slabs must not carry out of the field's headroom, and the unpacked sums must be the plain sums.
static void pack_plan_cases() {
  std::mt19937_64 rng(4242);
  int packed = 0, refused = 0, wide = 0;
  for (int it = 0; it < 4000; ++it) {
    const int D = 1 + (int)(rng() % 16);
    const uint64_t n_slab = 1 + rng() % (it % 3 == 0 ? 5000 : 70000);
    const int n = (int)std::min<uint64_t>(n_slab, 1 + rng() % 200);
    const int n_slabs = 1 + (int)(rng() % 256);
    std::vector<std::vector<uint64_t>> v(n, std::vector<uint64_t>(D, 0));
    unsigned __int128 max_abs[16] = {0};
    uint64_t or_abs[16] = {0};
    for (int d = 0; d < D; ++d) {
      const int kind = (int)(rng() % 6);
      if (kind == 0) continue;  // nobody requests this resource
      const int unit = kind == 1 ? 0 : (int)(rng() % 30);  // common trailing zeros (memory in MiB, ...)
      const int bits = 1 + (int)(rng() % (kind == 5 ? 44 : 14));
      for (int i = 0; i < n; ++i) {
        if (rng() % 4 == 0) continue;
        const uint64_t x = (rng() & ((1ull << bits) - 1ull)) << unit;
        v[i][d] = x, or_abs[d] |= x;
        if (x > max_abs[d]) max_abs[d] = x;
      }
    }
    const bool pad = it & 1;
    const uint32_t max_words = it % 5 == 0 ? 4u : pack_max_words(D);  // (the sweep always asks for 4)
    const PackPlan pk = make_pack_plan(D, max_abs, or_abs, false, n_slab, pad, max_words);
    if (pk.nw > 4) ++wide;
    if (pk.nw == 0) {
      ++refused;
      continue;
    }
    ++packed;
    if (pk.nw > max_words || (pk.stride != 2 && pk.stride != 4 && pk.stride != 8) || pk.stride < pk.nw || pk.stride >= 2 * std::max(pk.nw, 2u) ||
        pk.rec_bytes > packed_rec_max(D) || pk.rec_bytes < (pk.nw + 1) * 8 || (pad && !((pk.rec_bytes / 8) & 1)))
      ++g_fail, fprintf(stderr, "FAIL: pack plan shape nw=%u stride=%u rec=%u\n", pk.nw, pk.stride, pk.rec_bytes);
    // no overlap, nothing crosses a word, the count owns the low bits of word 0
    uint64_t occ[8] = {(1ull << pk.cnt_width) - 1ull, 0, 0, 0, 0, 0, 0, 0};
    if (pk.cnt_width < 64 && (n_slab >> pk.cnt_width) != 0) ++g_fail, fprintf(stderr, "FAIL: the pod count field is too narrow\n");
    for (int d = 0; d < D; ++d) {
      if (!pk.width[d]) {
        if (max_abs[d] != 0) ++g_fail, fprintf(stderr, "FAIL: a used dimension has no field\n");
        if (pk.desc[d] != 0) ++g_fail, fprintf(stderr, "FAIL: a descriptor without a field\n");
        continue;
      }
      if (pk.word[d] >= pk.nw || pk.pos[d] + pk.width[d] > 64 || pk.width[d] < kPackHeadroomBits || pk.width[d] > 64 - kPackHeadroomBits) { ++g_fail, fprintf(stderr, "FAIL: field outside its word\n"); continue; }
      const uint64_t m = ((1ull << pk.width[d]) - 1ull) << pk.pos[d];
      if (occ[pk.word[d]] & m) ++g_fail, fprintf(stderr, "FAIL: fields overlap\n");
      occ[pk.word[d]] |= m;
      if ((or_abs[d] & ((1ull << pk.shift[d]) - 1ull)) != 0) ++g_fail, fprintf(stderr, "FAIL: the shift drops set bits\n");
      // a full slab of the maximum stays inside the field
      const unsigned __int128 worst = (max_abs[d] >> pk.shift[d]) * (unsigned __int128)n_slab;
      if ((worst >> pk.width[d]) != 0) ++g_fail, fprintf(stderr, "FAIL: a full slab carries out of field %d\n", d);
    }
    // every slab: pack its pods and fold by whole-word adds (the aggregate kernel); slab 0 carries the random pods, the
    // others a full slab of the per-dimension maxima (the worst case the headroom has to hold)
    std::vector<unsigned __int128> want(D, 0);
    unsigned __int128 want_pods = 0;
    uint64_t cls[8][3] = {};
    for (int sl = 0; sl < n_slabs; ++sl) {
      uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (sl == 0) {
        for (int i = 0; i < n; ++i) {
          uint64_t w[8] = {1ull, 0, 0, 0, 0, 0, 0, 0};
          for (int d = 0; d < D; ++d) {
            want[d] += v[i][d];
            if (pk.width[d]) w[pk.word[d]] += (v[i][d] >> pk.shift[d]) << pk.pos[d];
          }
          for (int k = 0; k < 8; ++k) acc[k] += w[k];
        }
        want_pods += (unsigned)n;
      } else {
        acc[0] = n_slab;
        for (int d = 0; d < D; ++d)
          if (pk.width[d]) {
            acc[pk.word[d]] += (uint64_t)((unsigned __int128)((uint64_t)max_abs[d] >> pk.shift[d]) * n_slab) << pk.pos[d];
            want[d] += max_abs[d] * (unsigned __int128)n_slab;
          }
        want_pods += n_slab;
      }
      // block_record_sums (kt_index_device.h): the top field shifted down, the fields below it in two classes
      for (uint32_t k = 0; k < pk.nw; ++k) {
        const uint64_t low = (1ull << pk.top_pos[k]) - 1ull;
        cls[k][0] += acc[k] & pk.even[k] & low, cls[k][1] += acc[k] & ~pk.even[k] & low, cls[k][2] += acc[k] >> pk.top_pos[k];
      }
    }
    auto field = [&](uint32_t desc) -> unsigned __int128 {  // packed_field
      const uint32_t sel = desc & 31u, pos = (desc >> 8) & 63u, wext = (desc >> 16) & 127u, shift = (desc >> 24) & 63u;
      if (!wext) return 0;
      return (unsigned __int128)((cls[sel >> 2][sel & 3u] >> pos) & (wext >= 64 ? ~0ull : (1ull << wext) - 1ull)) << shift;
    };
    if (field(pk.cnt_desc) != want_pods) ++g_fail, fprintf(stderr, "FAIL: packed pod count (case %d)\n", it);
    for (int d = 0; d < D; ++d)
      if (field(pk.desc[d]) != want[d]) { ++g_fail, fprintf(stderr, "FAIL: packed sum of dimension %d differs (case %d, %d slabs)\n", d, it, n_slabs); break; }
  }
  // negative requests never pack, nor does a field that leaves no headroom
  {
    unsigned __int128 mx[16] = {5};
    uint64_t oa[16] = {5};
    if (make_pack_plan(1, mx, oa, true, 100, false).nw != 0) ++g_fail, fprintf(stderr, "FAIL: negative requests packed\n");
    mx[0] = (unsigned __int128)1 << 50, oa[0] = 1ull << 50 | 1ull;
    if (make_pack_plan(1, mx, oa, false, 1000, false).nw != 0) ++g_fail, fprintf(stderr, "FAIL: a field whose sum over the slabs leaves 64 bits packed\n");
  }
  if (packed < 1000 || refused < 10 || wide < 100)
    ++g_fail, fprintf(stderr, "FAIL: pack plan cases too one-sided (%d packed, %d refused, %d of more than four words)\n", packed, refused, wide);
  printf("pack plan: %d packed (%d of more than four words), %d refused\n", packed, wide, refused);
}

// ---- plan_wg_ranges (kt_kernels.hip, host code since round 6): the record ranges of the workgroups of a namespace-ordered scan
namespace kt {
uint32_t wg_range_cap(int64_t n, int G);
void plan_wg_ranges(const unsigned long long* ns_end, uint32_t n_keys, int64_t n, int G, uint32_t* range);
}
static void plan_ranges_cases() {
  std::mt19937_64 rng(777);
  long at_boundary = 0, ends = 0;
  for (int it = 0; it < 2000; ++it) {
    const uint32_t n_keys = 1 + (uint32_t)(rng() % (it % 5 == 0 ? 3000 : 300));
    const int G = 1 + (int)(rng() % 256);
    std::vector<unsigned long long> ns_end(n_keys);
    unsigned long long pos = 0;
    for (uint32_t k = 0; k < n_keys; ++k) {
      const int kind = (int)(rng() % 8);
      pos += kind == 0 ? 0 : kind == 1 ? rng() % 200000 : rng() % 6000;  // empty, huge and ordinary namespaces
      ns_end[k] = pos;
    }
    const int64_t n = (int64_t)pos;
    std::vector<uint32_t> range((size_t)G + 2, 0xDEADBEEFu);
    plan_wg_ranges(ns_end.data(), n_keys, n, G, range.data());
    const uint32_t cap = wg_range_cap(n, G);
    uint32_t largest = 0;
    EXPECT(range[0] == 0u && range[G] == (uint32_t)n, "ranges cover [0, %lld): %u .. %u", (long long)n, range[0], range[G]);
    for (int g = 0; g < G; ++g) {
      EXPECT(range[g + 1] >= range[g] && range[g + 1] - range[g] <= cap, "range %d of %d: %u .. %u (cap %u)", g, G, range[g], range[g + 1], cap);
      largest = std::max(largest, range[g + 1] - range[g]);
      if (g + 1 < G && range[g + 1] > range[g] && range[g + 1] < (uint32_t)n) ++ends, at_boundary += std::binary_search(ns_end.begin(), ns_end.end(), (unsigned long long)range[g + 1]);
    }
    EXPECT(range[G + 1] == largest, "largest range %u, reported %u", largest, range[G + 1]);
  }
  printf("workgroup ranges: %ld of %ld inner ends at a namespace boundary\n", at_boundary, ends);
  EXPECT(at_boundary * 4 > ends, "hardly any range ends at a namespace boundary (%ld of %ld)", at_boundary, ends);
}

static int run_anchored(int argc, char** argv);
int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--anchored")) return run_anchored(argc, argv);
  if (argc > 1) return run_file(argv[1], argc > 2 ? (uint32_t)atoi(argv[2]) : 80u * 1024u - check_fixed_lds());
  pack_plan_cases();
  plan_ranges_cases();
  long chunks_seen = 0, matches = 0, simple_seen = 0;
  auto acc = [&](long r) {
    if (r >= 500000000L) r -= 500000000L, ++simple_seen;
    chunks_seen = std::max(chunks_seen, r / 1000000L), matches += r % 1000000L;
  };
  // KT_SIM_SEED_OFFSET=n: the same suite on other random programs (the pinned fingerprint belongs to offset 0)
  const uint32_t seed_off = getenv("KT_SIM_SEED_OFFSET") ? (uint32_t)atoi(getenv("KT_SIM_SEED_OFFSET")) * 100u : 0u;
  for (uint32_t seed = 1 + seed_off; seed <= 40 + seed_off; ++seed) {
    // everything resident in one chunk
    acc(run_case(seed, 40 + seed % 60, 1 + seed % 9, 6, 4, 3, 3, seed % 5 == 0 ? 0.05 : 0.0, 160 << 10, 160 << 10, 160, 300));
    // tight budgets: a few words per chunk, different for the two kernels
    acc(run_case(1000 + seed, 200 + seed * 7, 2 + seed % 17, 8, 5, 4, 4, seed % 4 == 0 ? 0.02 : 0.0, 12000 + 700 * (seed % 7),
                 9000 + 500 * (seed % 5), 16 + 8 * (seed % 20), 120));
    // matchLabels-style programs: the simple image form, one and several chunks
    acc(run_case(2000 + seed, 60 + seed * 3, 1 + seed % 11, 8, 4, 2, 2, seed % 6 == 0 ? 0.03 : 0.0, seed % 2 ? 160 << 10 : 9000,
                 seed % 2 ? 160 << 10 : 7000, 72, 200, 1));
  }
  // a program the size of BASELINE configs[4]'s shard: 10k throttles, ~30k terms, 64 namespaces, real LDS budgets
  acc(run_case(77, 10000, 64, 16, 16, 5, 3, 0.0, 160 << 10, 140 << 10, 72, 64));
  // more than 4096 terms with single-namespace classes (128-bit class granularity) and a universe of 1 namespace
  acc(run_case(78, 3000, 1, 10, 8, 3, 2, 0.001, 60 << 10, 60 << 10, 152, 64));
  // four to six requirements per term: up to five positive keys are counted by the bitmaps (max_need 4-5: the NEED = 5 kernels),
  // a term with six keeps five and takes the slow confirmation; eight requirements: most terms are slow
  acc(run_case(79, 300, 5, 12, 3, 3, 6, 0.0, 160 << 10, 160 << 10, 72, 400));
  acc(run_case(82, 200, 4, 14, 2, 2, 8, 0.0, 160 << 10, 160 << 10, 72, 400));
  acc(run_case(83, 120, 3, 10, 1, 2, 8, 0.0, 160 << 10, 160 << 10, 72, 600));  // one value per key: a pod that carries the five anchor keys is a candidate
  // throttles with up to 150 selector terms: beyond 64 a throttle's run of numbers spans words — ONE group, its copies with their
  // own admission sets; scan() dedupes match by match across the words as the full check and the plain fold do (has_long)
  acc(run_case(80, 24, 6, 8, 3, 150, 2, 0.0, 160 << 10, 160 << 10, 72, 150));
  acc(run_case(81, 60, 3, 6, 3, 70, 2, 0.01, 20000, 16000, 72, 150));
  // more keys than atom slots: 12 and 20 keys in programs whose pods carry at most 8 labels
  acc(run_case(90, 40, 4, 12, 3, 2, 3, 0.01, 160 << 10, 160 << 10, 72, 200, 0, 8));
  acc(run_case(91, 60, 5, 20, 4, 3, 2, 0.0, 24000, 20000, 72, 200, 1, 8));
  acc(run_case(92, 30, 3, 20, 2, 2, 3, 0.02, 160 << 10, 160 << 10, 72, 200, 0, 16));
  if (chunks_seen < 8) ++g_fail, fprintf(stderr, "FAIL: the tight budgets never produced a multi-chunk index (%ld)\n", chunks_seen);
  if (matches < 1000) ++g_fail, fprintf(stderr, "FAIL: only %ld matches — the cases are too sparse to mean anything\n", matches);
  if (simple_seen < 10) ++g_fail, fprintf(stderr, "FAIL: only %ld programs took the simple image form\n", simple_seen);
  if (g_slow_confirms < 10) ++g_fail, fprintf(stderr, "FAIL: only %ld slow confirmations — the slow shapes were not exercised\n", g_slow_confirms);
  // kt_translate_pods' second pass (two keys of a pod with one home slot): programs that name more keys than the pods have slots
  if (g_late_atoms < 300) ++g_fail, fprintf(stderr, "FAIL: only %ld of %ld atoms found their home slot taken — the spill path of the translation was hardly exercised\n", g_late_atoms, g_placed_atoms);
  if (g_fail) {
    fprintf(stderr, "%d expectation(s) failed\n", g_fail);
    return 1;
  }
  printf("index_sim_test: all expectations held (max %ld chunks, %ld matches, %ld simple-form programs, %ld slow confirmations, %.1f %% of the atoms outside their home slot); fingerprint of all indexes %016llx\n",
         chunks_seen, matches, simple_seen, g_slow_confirms, 100.0 * (double)g_late_atoms / (double)std::max(1L, g_placed_atoms), (unsigned long long)g_fingerprint);
  return 0;
}

// the dumped program of a BASELINE config (tools/dump_program.py) as a Program + its pod sample
static bool load_dump(const char* path, Program& p, std::vector<PodLabels>& pods, uint32_t* D, uint32_t* L) {
  FILE* fh = fopen(path, "rb");
  if (!fh) return false;
  const std::vector<uint32_t> hdr = read_array(fh);
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
  p.n_ns = hdr[2], p.gw = hdr[3], *D = hdr[5];
  for (uint32_t t = 0; t < hdr[0]; ++t) p.thr.push_back(ThrInfo{live[t] != 0, cluster[t] != 0, thr_ns[t]});
  *L = 1;
  for (size_t i = 0; i + 1 < loff.size(); ++i) {
    *L = std::max(*L, loff[i + 1] - loff[i]);
    PodLabels pod;
    pod.ns = pod_ns[i];
    for (uint32_t j = loff[i]; j < loff[i + 1]; ++j) pod.keys.push_back(lkey[j]), pod.pairs.push_back(lpair[j]);
    pods.push_back(pod);
  }
  return true;
}

static int run_anchored(int argc, char** argv) {
  if (argc > 2) {  // a dumped program: visits per pod of the classic index and of the per-anchor indexes, exactness on a sample
    Program p;
    std::vector<PodLabels> pods;
    uint32_t D = 8, L = 8;
    if (!load_dump(argv[2], p, pods, &D, &L)) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    const uint32_t thr_bytes = 8 * D + 8, agg_budget = 160u * 1024u - aggregate_fixed_lds(), chk_budget = 160u * 1024u - check_fixed_lds();
    HostIndex classic;
    build_index(classic, p.thr_term_off, p.term_thr, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
                [&](uint32_t t) { return p.thr[t]; }, p.n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)L);
    const auto t0 = std::chrono::steady_clock::now();
    AnchoredIndexes A;
    build_anchored(p, agg_budget, chk_budget, thr_bytes, (int)L, A);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const auto t1 = std::chrono::steady_clock::now();
    AnchoredIndex AX;
    build_anchored_index(AX, p.thr_term_off, p.term_flags, p.term_req_off, p.req_op, p.req_key, p.req_val_off, p.req_val,
                         [&](uint32_t t) { return p.thr[t]; }, p.n_ns, p.ns_term_ok, p.gw, agg_budget, chk_budget, thr_bytes, (int)L, check_word_lds((int)D), classic);
    const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    std::map<uint32_t, uint32_t> block_of;
    for (uint32_t c = 1; c < AX.block_anchor.size(); ++c) block_of[AX.block_anchor[c]] = c;
    long vc = 0, va = 0, vx = 0, matches = 0, v_max = 0;
    g_items = 0;
    std::vector<long> useful(pods.size()), hit(pods.size()), visited(pods.size());
    const size_t T = p.thr.size();
    for (size_t i = 0; i < pods.size(); ++i) {
      long w0 = g_word_steps;
      const long u0 = g_word_useful, h0 = g_word_hit;
      const auto one = scan(p, classic, pods[i]);
      useful[i] = g_word_useful - u0, hit[i] = g_word_hit - h0, visited[i] = g_word_steps - w0;
      vc += g_word_steps - w0, w0 = g_word_steps;
      const auto got = scan_anchored(A, pods[i]);
      va += g_word_steps - w0, v_max = std::max(v_max, g_word_steps - w0), w0 = g_word_steps;
      const auto got2 = scan_concat(p, AX, block_of, pods[i]);
      vx += g_word_steps - w0;
      EXPECT(got2 == one, "pod %zu: the concatenated anchored index and the classic index disagree (%zu vs %zu throttles)", i, got2.size(), one.size());
      EXPECT(got == one, "pod %zu: the per-anchor indexes and the classic index disagree (%zu vs %zu throttles)", i, got.size(), one.size());
      if (i % 16 == 0)
        for (uint32_t t = 0; t < T; ++t) {
          const int want = brute(p, t, pods[i]);
          const auto b = got.find(t);
          EXPECT((b == got.end() ? 0 : b->second) == want, "anchored: pod %zu throttle %u", i, t);
        }
      matches += (long)got.size();
    }
    size_t max_words = 0, max_chunks = 0;
    for (const HostIndex& ix : A.ix) max_words = std::max<size_t>(max_words, ix.bm_words), max_chunks = std::max(max_chunks, ix.bm_chunks.size());
    printf("%s: %zu throttles, %zu terms -> %zu anchors (anchor 0: %zu throttles kept whole), %zu virtual throttles, %zu term copies (%.2fx)\n", argv[2], T,
           A.stats.n_terms_in, A.stats.n_anchors, A.stats.n_unanchored_throttles, A.stats.n_virtual_throttles, A.stats.n_terms_out,
           (double)A.stats.n_terms_out / (double)std::max<size_t>(1, A.stats.n_terms_in));
    printf("  per-anchor indexes: %zu words in all (classic: %u), largest %zu words / %zu chunk(s); split + %zu index builds %.1f ms (one thread)\n", A.words,
           classic.bm_words, max_words, max_chunks, A.ix.size(), ms);
    printf("  %zu pods: %.2f matches per pod; word visits per pod: classic %.2f, per-anchor %.2f (max %ld)\n", pods.size(),
           (double)matches / (double)pods.size(), (double)vc / (double)pods.size(), (double)va / (double)pods.size(), v_max);
    {
      // what a lane-private word skip would walk (NEXT.md): per pod the words with an admitted bit of one of its atoms; a
      // tile of 64 pods in namespace order steps max-over-lanes times
      std::vector<size_t> ord(pods.size());
      for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return pods[a].ns < pods[b].ns; });
      double su = 0, sh = 0, tmax_u = 0, tmax_v = 0, tmax_h = 0;
      size_t tiles = 0;
      for (size_t i0 = 0; i0 < ord.size(); i0 += 64, ++tiles) {
        long mu = 0, mv = 0, mh = 0;
        for (size_t i = i0; i < std::min(ord.size(), i0 + 64); ++i) mu = std::max(mu, useful[ord[i]]), mv = std::max(mv, visited[ord[i]]), mh = std::max(mh, hit[ord[i]]);
        tmax_u += mu, tmax_v += mv, tmax_h += mh;
      }
      for (size_t i = 0; i < pods.size(); ++i) su += useful[i], sh += hit[i];
      printf("  classic index, words with an admitted atom bit per pod: %.2f (with a match: %.2f); per tile of 64 pods in namespace order, max over lanes: visited %.1f, useful %.1f, with a match %.1f\n",
             su / pods.size(), sh / pods.size(), tmax_v / tiles, tmax_u / tiles, tmax_h / tiles);
    }
    printf("  items per pod (block 0 + one per anchor atom it carries): %.2f\n", (double)g_items / (double)pods.size());
    {
      // rows a block's image really needs: the atoms with a bit in some word of the block (an item record could carry
      // block-local row numbers, translated when the item view is built — the image then holds these rows only)
      size_t sum_rows = 0, max_rows = 0, sum_bytes = 0, max_bytes = 0;
      for (size_t c = 0; c + 1 < AX.blk_chunk0.size(); ++c) {
        std::set<uint32_t> live;
        size_t bytes = 0;
        for (uint32_t ci = AX.blk_chunk0[c]; ci < AX.blk_chunk0[c + 1]; ++ci) {
          const BmChunk& ch = AX.ix.bm_chunks[ci];
          const uint64_t* rows = (const uint64_t*)(AX.ix.bm_images.data() + ch.img_off);
          for (uint32_t r = 1; r < AX.ix.bm_rows; ++r)
            for (uint32_t w = 0; w < ch.n_words; ++w)
              if (cell_any(ch, rows, r, w) | cell_veto(ch, rows, r, w)) { live.insert(r); break; }
          bytes += (size_t)ch.n_words * 16u;
        }
        sum_rows += live.size(), max_rows = std::max(max_rows, live.size());
        const size_t b = (live.size() + 1) * bytes;
        sum_bytes += b, max_bytes = std::max(max_bytes, b);
      }
      const size_t nb = AX.blk_chunk0.size() - 1;
      printf("  atoms with a bit in a block: %.1f on average, %zu at most (of %u rows) -> rows of a block-local image %.1f KB on average, %.1f KB at most\n",
             (double)sum_rows / nb, max_rows, AX.ix.bm_rows, sum_bytes / 1024.0 / nb, max_bytes / 1024.0);
    }
    printf("  concatenated (shared atom numbering, one chunked index over %zu virtual namespaces): %zu chunks, %u words, images %.1f MB, largest LDS part %u B / %u throttles / %u words per chunk, slab scratch %.0f MB; %.2f word visits per pod; built in %.1f ms\n",
           (size_t)AX.block_anchor.size() * AX.n_ns, AX.ix.bm_chunks.size(), AX.ix.bm_words, AX.ix.bm_images.size() / 1048576.0, AX.ix.bm_max_lds, AX.ix.bm_max_thr,
           AX.ix.bm_max_words, AX.ix.bm_slab_bytes / 1048576.0, (double)vx / (double)pods.size(), ms2);
    if (g_fail) { fprintf(stderr, "%d expectation(s) failed\n", g_fail); return 1; }
    printf("index_sim_test --anchored: all expectations held\n");
    return 0;
  }
  long vc = 0, va = 0, matches = 0;
  for (uint32_t seed = 1; seed <= 30; ++seed) {
    matches += anchored_case(seed, 40 + seed % 60, 1 + seed % 9, 6, 4, 3, 3, seed % 5 == 0 ? 0.05 : 0.0, 160 << 10, 160 << 10, 160, 200, &vc, &va);
    matches += anchored_case(1000 + seed, 200 + seed * 7, 2 + seed % 17, 8, 5, 4, 4, seed % 4 == 0 ? 0.02 : 0.0, 12000 + 700 * (seed % 7),
                             9000 + 500 * (seed % 5), 16 + 8 * (seed % 20), 100, &vc, &va);
  }
  matches += anchored_case(77, 3000, 32, 16, 16, 4, 3, 0.0, 160 << 10, 140 << 10, 72, 64, &vc, &va);  // configs[4]-like shapes
  matches += anchored_case(79, 300, 5, 12, 3, 3, 6, 0.0, 160 << 10, 160 << 10, 72, 300, &vc, &va);    // slow shapes (> 3 positive keys)
  matches += anchored_case(80, 24, 6, 8, 3, 150, 2, 0.0, 160 << 10, 160 << 10, 72, 100, &vc, &va);    // > 64 terms: kept whole
  if (matches < 1000) ++g_fail, fprintf(stderr, "FAIL: only %ld matches\n", matches);
  if (g_fail) { fprintf(stderr, "%d expectation(s) failed\n", g_fail); return 1; }
  printf("index_sim_test --anchored: all expectations held (%ld matches; word visits: classic %ld, per-anchor %ld)\n", matches, vc, va);
  return 0;
}
