// kt_index.cpp — host-side construction and upload of the exact term-bitmap index (see kt_index.h).
#include "kt_index.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "../../include/kt_snapshot.h"

namespace kt {

namespace {

struct BT {  // one indexed term (plain data: tens of thousands of them are built, moved and dropped per index build)
  uint32_t g, t;
  bool adj;                                 // owner has several terms
  bool slow;                                // needs the generic walk to confirm a candidate
  uint32_t need;                            // positive requirements that are kept and counted (<= 5; slow terms: their five anchors)
  // positive requirements that enter `any`, one per key (at most 5 are kept): an explicit
  // set of pair atoms (pos_key < 0), or — Exists, not narrowed by an In on the same key — EVERY atom a pod carrying that
  // key can show up with (pos_key = the key: the atoms are those of whole_key(key), expanded only where rows are set).
  // The pair atoms of all kept positives sit side by side in the atom pool.
  uint32_t n_pos;
  int64_t pos_key[5];
  uint32_t pos_off, pos_cnt;                // atom pool: pair atoms of the positive requirements
  uint32_t neg_off, neg_cnt;                // atom pool: pair atoms of the NotIn requirements
  uint32_t nk_off, nk_cnt;                  // key pool: keys of the DoesNotExist requirements (all atoms of the key)
  const uint32_t* adm;                      // namespace admission set as nsw words (points into the caller's / build_index's array)
};

struct TC { uint32_t bt, grp; };  // a term copy: the term and the group (namespace cell of its throttle) it belongs to
// The construction's larger containers, one set per host thread, kept from build to build: a recompile is on the
// scheduler's critical path, and a few megabytes of fresh vectors are a few thousand page faults per build.
struct BuildScratch {
  std::vector<BT> bts;
  std::vector<TC> tcs;
  std::vector<uint32_t> first_of, atom_pool, key_pool, grp_adm, grp_first, gorder, order, num, cls, pool_row, by_num;
};

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// lexicographic order of two runs of n words (what std::vector<uint32_t>::operator< gives)
inline int memcmp_words(const uint32_t* x, const uint32_t* y, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}

}  // namespace

// f(begin, end, part) over [0, n) on up to 16 host threads (a recompile after a throttle event is on the scheduler's
// critical path: 10k throttles / 30k terms are 25 ms of single-threaded index construction)
void parallel_for(size_t n, size_t min_per_part, const std::function<void(size_t, size_t, size_t)>& f, size_t* parts_out) {
  size_t parts = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n / std::max<size_t>(1, min_per_part)));
  if (getenv("KT_INDEX_THREADS")) parts = std::max(1, atoi(getenv("KT_INDEX_THREADS")));
  if (parts_out) *parts_out = parts;
  if (parts <= 1) {
    f(0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  for (size_t k = 0; k < parts; ++k) th.emplace_back([&, k] { f(n * k / parts, n * (k + 1) / parts, k); });
  for (auto& t : th) t.join();
}
size_t parallel_parts(size_t n, size_t min_per_part) {
  size_t parts = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n / std::max<size_t>(1, min_per_part)));
  if (getenv("KT_INDEX_THREADS")) parts = std::max(1, atoi(getenv("KT_INDEX_THREADS")));
  return parts;
}

// out[n][gw] (bit g of row n) = in[g][nsw] (bit n of row g): 32 x 32 bit blocks (Hacker's Delight 7-3)
void transpose_term_ns_bits(const std::vector<uint32_t>& in, size_t G, uint32_t nsw, uint32_t n_ns, uint32_t gw, std::vector<uint32_t>& out) {
  out.assign((size_t)n_ns * gw, 0u);
  const size_t gb = (G + 31) / 32;
  parallel_for(gb, 256, [&](size_t b0, size_t b1, size_t) {
    for (size_t bg = b0; bg < b1; ++bg)
      for (uint32_t bn = 0; bn < nsw; ++bn) {
        uint32_t a[32];
        bool any = false;
        for (uint32_t k = 0; k < 32; ++k) {
          const size_t g = bg * 32 + k;
          a[k] = g < G ? in[g * nsw + bn] : 0u;
          any |= a[k] != 0u;
        }
        if (!any) continue;
        // a[k] bit j  ->  t[j] bit k
        uint32_t m = 0x0000FFFFu;
        for (uint32_t j = 16; j != 0; j >>= 1, m ^= m << j)
          for (uint32_t k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = (a[k] ^ (a[k + j] << j)) & ~m;  // (lsb-first variant of the classic step)
            a[k] ^= t;
            a[k + j] ^= t >> j;
          }
        for (uint32_t j = 0; j < 32; ++j) {
          const uint32_t n = bn * 32 + j;
          if (n < n_ns && a[j]) out[(size_t)n * gw + bg] = a[j];
        }
      }
  }, nullptr);
}

// Atom ids and home slots.  A pod carries at most one atom per key, and the scans read, for every visited word, the pod's
// atom rows slot by slot — one gather per slot, lane = pod.  LDS serves the lanes of a gather that fall into distinct
// bank slots in one pass, and the cell of (row id, word w) sits in bank slot id mod 32 of the word's column (kt_index.h:
// image layout).  Hence:
//   * every KEY gets a home slot (the keys with the most atoms first, each to the slot with the fewest atoms so far) and
//     kt_translate_pods puts a pod's atom of that key there when the slot is free: lane by lane, a gather reads atoms of
//     the same few keys;
//   * the atoms of a slot are numbered side by side (key by key): any 32 consecutive ids fall into 32 distinct bank slots,
//     so the distinct atoms of one slot collide only where the slot holds more than 32 atoms.
// A program that names no more keys than the pods have slots is scanned without bank conflicts this way.  The BASELINE
// programs name 16 keys of which a pod carries 8 in 8 slots: two keys share a home, a quarter of the atoms sit in a foreign
// slot, and the LDS bank model of tests/cpp/index_sim_test.cpp (KT_SIM_BANKS=1) counts 2.4 passes per gather in namespace
// order where the numbering by pair id had 2.5 (profiles/r05_lds_bank_model.txt has the model and what the GPU said).
static void number_atoms_by_home_slot(HostIndex& out) {
  const uint32_t la = out.la;
  std::map<uint32_t, std::vector<uint32_t>> of_key;  // key -> entries of out.atoms (ascending atom)
  for (uint32_t i = 0; i < out.atoms.size(); ++i) of_key[out.atom_key[i]].push_back(i);
  std::vector<std::pair<uint32_t, uint32_t>> keys;  // (atoms, key)
  for (auto& kv : of_key) keys.emplace_back((uint32_t)kv.second.size(), kv.first);
  std::sort(keys.begin(), keys.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
    return a.first != b.first ? a.first > b.first : a.second < b.second;
  });
  std::vector<uint32_t> load(la, 0u);
  std::vector<std::vector<uint32_t>> keys_of_slot(la);
  for (auto& k : keys) {
    uint32_t best = 0;
    for (uint32_t sl = 1; sl < la; ++sl)
      if (load[sl] < load[best]) best = sl;
    load[best] += k.first;
    keys_of_slot[best].push_back(k.second);
  }
  uint32_t next = 1;
  for (uint32_t sl = 0; sl < la; ++sl)
    for (uint32_t key : keys_of_slot[sl])
      for (uint32_t i : of_key[key]) out.atoms[i].id = next++, out.atoms[i].home = sl;
}


void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns,
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes,
                 int max_labels, const std::vector<uint32_t>* adm_in, uint32_t chk_budget_full, uint32_t chk_word, const HostIndex* atoms_from,
                 uint32_t thr_bytes_packed) {
  {
    // a fresh index in the OLD index's storage: the full bitmaps and the chunk images are a few megabytes that would
    // otherwise be unmapped and faulted in again page by page on every build
    HostIndex fresh;
    auto keep = [](auto& dst, auto& src) { src.clear(), dst = std::move(src); };
    keep(fresh.full_any, out.full_any), keep(fresh.full_veto, out.full_veto), keep(fresh.full_nsrows, out.full_nsrows);
    keep(fresh.full_hdr, out.full_hdr), keep(fresh.full_term_t, out.full_term_t), keep(fresh.full_term_g, out.full_term_g);
    keep(fresh.full_term_rank, out.full_term_rank), keep(fresh.full_real, out.full_real), keep(fresh.bm_images, out.bm_images);
    out = std::move(fresh);
  }
  (void)term_thr;
  static const bool dbg_time = getenv("KT_DEBUG_COMPILE") != nullptr;  // phase times on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  build_index: %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const size_t T = thr_term_off.empty() ? 0 : thr_term_off.size() - 1;
  // a reachable unconvertible podSelector makes term ORDER matter: these throttles are walked term by term
  std::vector<uint8_t> is_slow_thr(T, 0);
  for (size_t t = 0; t < T; ++t) {
    const ThrInfo ti = thr_info((uint32_t)t);
    if (!ti.live) continue;
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g)
      if ((term_flags[g] & KT_TERM_POD_SEL_INVALID) && !(ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID))) {
        out.slow_thr.push_back((uint32_t)t);
        is_slow_thr[t] = 1;
        break;
      }
  }
  // A throttle with more than 64 selector terms: its run of term numbers spans several 64-bit words.  The wordwise settle of
  // the lean scans applies "reported once" per WORD (run masks), so a program that holds such a throttle is scanned by the
  // instantiations that dedupe match by match (a lane meets its matches in ascending number, and a repeat of its previous
  // throttle is dropped — across words too): the FULL check and the plain fold (HostIndex::has_long, IndexDev::has_long).
  // Until round 5 these throttles went to the slow list — walked term by term for every pod: 16.9 ms per step at 100k pods
  // where the same terms in the index take a fraction of a millisecond (profiles/r05_offpath_timing.jsonl).  The run never
  // straddles a chunk (cut_chunks: splittable), so a throttle whose terms do not fit ONE chunk image still takes the slow
  // list: kMaxIndexedTerms.
  for (size_t t = 0; t < T; ++t) {
    if (is_slow_thr[t] || !thr_info((uint32_t)t).live) continue;
    const uint32_t nt = thr_term_off[t + 1] - thr_term_off[t];
    if (nt > kMaxIndexedTerms) out.slow_thr.push_back((uint32_t)t), is_slow_thr[t] = 1;
    else if (nt > 64u) out.has_long = true;
  }
  // ---- referenced atoms.  A pod label (k, v) is ONE atom: the pair (k, v) when some In / NotIn requirement names it,
  //      else the key atom of k when some Exists / DoesNotExist requirement names k (else nothing).  A key-level
  //      requirement therefore covers the key atom AND every referenced pair of that key.
  const uint32_t nsw = (n_ns + 31) / 32;
  std::unordered_set<uint32_t> pair_keys, key_atoms;
  std::unordered_map<uint32_t, std::vector<uint32_t>> pairs_of_key;
  {
    // a program names a few hundred distinct pairs tens of thousands of times: every key's list is kept sorted and
    // duplicate-free as it grows (a binary search per value), the map lookups go through a small direct-mapped cache
    constexpr uint32_t kSlots = 256;
    uint32_t pk_key[kSlots], ka_key[kSlots];
    std::vector<uint32_t>* pk_list[kSlots];
    for (uint32_t i = 0; i < kSlots; ++i) pk_key[i] = ka_key[i] = 0u, pk_list[i] = nullptr;
    bool ka_zero_done = false;
    for (size_t t = 0; t < T; ++t) {
      const ThrInfo ti = thr_info((uint32_t)t);
      if (!ti.live || is_slow_thr[t]) continue;
      for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g)
        for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
          const uint32_t key = req_key[r], slot = key & (kSlots - 1);
          if (req_op[r] == KT_OP_IN || req_op[r] == KT_OP_NOT_IN) {
            if (pk_list[slot] == nullptr || pk_key[slot] != key) pk_list[slot] = &pairs_of_key[key], pk_key[slot] = key;  // (node-based map: the address stays)
            std::vector<uint32_t>& v = *pk_list[slot];
            for (uint32_t q = req_val_off[r]; q < req_val_off[r + 1]; ++q) {
              const auto it = std::lower_bound(v.begin(), v.end(), req_val[q]);
              if (it == v.end() || *it != req_val[q]) v.insert(it, req_val[q]);
            }
          } else if (key == 0u ? !ka_zero_done : ka_key[slot] != key) {
            key_atoms.insert(key);
            if (key == 0u) ka_zero_done = true; else ka_key[slot] = key;
          }
        }
    }
    for (auto& kv : pairs_of_key)
      if (!kv.second.empty()) pair_keys.insert(kv.first);
  }
  if (atoms_from) {
    // an imposed numbering: which atoms a pod carrying a key can show up with is a fact of THAT table (a pair another
    // part of the full program names is a pair atom for every pod, whether this program names it or not)
    pairs_of_key.clear(), pair_keys.clear(), key_atoms.clear();
    for (size_t i = 0; i < atoms_from->atoms.size(); ++i) {
      const uint32_t atom = atoms_from->atoms[i].atom, key = atoms_from->atom_key[i];
      if (atom & kKeyAtom) {
        key_atoms.insert(key);
      } else {
        std::vector<uint32_t>& v = pairs_of_key[key];
        v.insert(std::lower_bound(v.begin(), v.end(), atom), atom);
        pair_keys.insert(key);
      }
    }
  }
  auto whole_key = [&](uint32_t key) {  // every atom a pod carrying `key` can show up with, sorted
    std::vector<uint32_t> a;
    auto it = pairs_of_key.find(key);
    if (it != pairs_of_key.end()) a = it->second;
    a.push_back(kKeyAtom | key);
    std::sort(a.begin(), a.end());
    return a;
  };
  // admission set of every term as namespace bit words, transposed from ns_term_ok by walking its SET bits (the rows
  // are sparse: a namespace admits a few thousand of the terms)
  const size_t G_all = term_req_off.empty() ? 0 : term_req_off.size() - 1;
  std::vector<uint32_t> adm_own;
  if (!adm_in || adm_in->size() != G_all * nsw) {
    adm_own.assign(G_all * nsw, 0u);
    for (uint32_t n = 0; n < n_ns; ++n) {
      const uint32_t* row = ns_term_ok.data() + (size_t)n * gw;
      for (uint32_t wi = 0; wi < gw; ++wi)
        for (uint32_t m = row[wi]; m; m &= m - 1) {
          const size_t g = (size_t)wi * 32 + (size_t)__builtin_ctz(m);
          if (g < G_all) adm_own[g * nsw + (n >> 5)] |= 1u << (n & 31);
        }
    }
  }
  const std::vector<uint32_t>& adm_all = adm_own.empty() && adm_in ? *adm_in : adm_own;  // the caller may hold the per-term sets already
  lap("atoms + admission transpose");
  // ---- terms (throttles are independent: built by ranges on several host threads, joined in throttle order).  The terms
  //      are plain records; their atoms and keys go to pools (one pair per part, joined below), the scratch containers
  //      of the construction are reused from term to term: no allocation per term.
  static thread_local BuildScratch scratch;
  std::vector<BT>& bts = scratch.bts;
  std::vector<uint32_t>& first_of = scratch.first_of;  // index into bts of the first term of the same throttle
  std::vector<uint32_t>&atom_pool = scratch.atom_pool, &key_pool = scratch.key_pool;
  bts.clear(), first_of.clear(), atom_pool.clear(), key_pool.clear();
  // (a part is at least 8000 throttles: the terms are cheap now, a thread has to earn its start-up and the joining of
  //  the parts; a single part writes straight into the final containers)
  const size_t term_parts = parallel_parts(T, 8000);
  std::vector<std::vector<BT>> bts_part(term_parts > 1 ? term_parts : 0);
  std::vector<std::vector<uint32_t>> first_part(bts_part.size()), atoms_part(bts_part.size()), keys_part(bts_part.size());
  std::vector<BT>& bts_all = bts;
  std::vector<uint32_t>&first_all = first_of, &atoms_all = atom_pool, &keys_all = key_pool;
  parallel_for(T, 8000, [&](size_t t_begin, size_t t_end, size_t part) {
  std::vector<BT>& bts = term_parts > 1 ? bts_part[part] : bts_all;
  std::vector<uint32_t>& first_of = term_parts > 1 ? first_part[part] : first_all;
  std::vector<uint32_t>& atom_pool = term_parts > 1 ? atoms_part[part] : atoms_all;
  std::vector<uint32_t>& key_pool = term_parts > 1 ? keys_part[part] : keys_all;
  size_t n_terms = 0;
  for (size_t t = t_begin; t < t_end; ++t) n_terms += thr_term_off[t + 1] - thr_term_off[t];
  bts.reserve(n_terms), first_of.reserve(n_terms), atom_pool.reserve(n_terms * 4), key_pool.reserve(n_terms);
  // positive requirements are merged per KEY (In S1 and In S2 = In S1∩S2, In S and Exists = In S): a pod carries
  // one atom per key, so the number of rows in which the term's bit is met is the number of satisfied positive keys
  struct PosReq {
    uint32_t key;
    std::vector<uint32_t> atoms;  // sorted pair atoms
    bool whole;                   // every atom of the key (Exists)
  };
  std::vector<PosReq> pos_by_key;  // slots, reused: n_pos of them are in use
  std::vector<uint32_t> atoms, both, neg, neg_keys;
  for (size_t t = t_begin; t < t_end; ++t) {
    const ThrInfo ti = thr_info((uint32_t)t);
    if (!ti.live || is_slow_thr[t]) continue;
    const size_t first = bts.size();
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g) {
      if (ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID)) continue;
      if (!ti.cluster && ti.ns >= n_ns) continue;
      BT b;
      b.g = g, b.t = (uint32_t)t, b.slow = false, b.need = 0;
      b.adj = thr_term_off[t + 1] - thr_term_off[t] > 1;
      b.adm = adm_all.data() + (size_t)g * nsw;
      bool any_ns = false;
      for (uint32_t wi = 0; wi < nsw; ++wi) any_ns |= b.adm[wi] != 0u;
      if (!any_ns) continue;  // admitted nowhere: can never match
      bool never = false;
      size_t n_pos = 0;
      neg.clear(), neg_keys.clear();
      for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
        const bool pair_op = req_op[r] == KT_OP_IN || req_op[r] == KT_OP_NOT_IN;
        atoms.clear();
        if (pair_op) {
          atoms.assign(req_val.begin() + req_val_off[r], req_val.begin() + req_val_off[r + 1]);
          if (atoms.size() > 1) {
            std::sort(atoms.begin(), atoms.end());
            atoms.erase(std::unique(atoms.begin(), atoms.end()), atoms.end());
          }
        }
        if (req_op[r] == KT_OP_IN || req_op[r] == KT_OP_EXISTS) {
          size_t q = 0;
          while (q < n_pos && pos_by_key[q].key != req_key[r]) ++q;
          if (q == n_pos) {
            if (pos_by_key.size() == n_pos) pos_by_key.emplace_back();
            pos_by_key[q].key = req_key[r], pos_by_key[q].atoms.assign(atoms.begin(), atoms.end()), pos_by_key[q].whole = !pair_op;
            ++n_pos;
          } else if (pos_by_key[q].whole) {
            // Exists and In S = In S ; Exists and Exists = Exists
            pos_by_key[q].atoms.assign(atoms.begin(), atoms.end()), pos_by_key[q].whole = !pair_op;
          } else if (pair_op) {
            both.clear();
            std::set_intersection(pos_by_key[q].atoms.begin(), pos_by_key[q].atoms.end(), atoms.begin(), atoms.end(), std::back_inserter(both));
            pos_by_key[q].atoms.swap(both);
          }  // In S and Exists = In S: nothing to do
        } else if (pair_op) {
          for (uint32_t a : atoms) neg.push_back(a);  // NotIn with no values: always satisfied
        } else {
          neg_keys.push_back(req_key[r]);
        }
      }
      for (size_t q = 0; q < n_pos; ++q)
        if (!pos_by_key[q].whole && pos_by_key[q].atoms.empty()) never = true;  // In with no values (or contradictory In sets): never satisfied
      if (never) continue;
      // exact shape: <= 5 positive keys (their atom sets are disjoint by construction; the scans count hits per term as a
      // 2-bit number for programs whose terms stop at three keys, as a 3-bit number otherwise)
      size_t keep0 = 0, keep1 = n_pos;  // the positives that are kept: [keep0, keep1)
      constexpr size_t kMaxNeed = 5;
      if (n_pos <= kMaxNeed) {
        b.need = (uint32_t)n_pos;
      } else {
        // More positive keys than the count holds: the term keeps its FIVE most selective positives (fewest atoms; a bare
        // key atom last) with need = 5 and is flagged `slow`: a pod that meets those is a candidate, the generic walk
        // confirms the rest.  (Until round 5 a term with more than three positive keys kept ONE anchor: every pod that
        // carried it — several per cent of the pods per term — went through the lane-divergent walk, 13.6x the step of the
        // same cluster with <= 3 keys per term, profiles/r05_offpath_timing.jsonl.)
        auto cost_of = [&](size_t i) -> size_t {
          if (!pos_by_key[i].whole) return pos_by_key[i].atoms.size();
          auto it = pairs_of_key.find(pos_by_key[i].key);
          const size_t n_atoms = (it == pairs_of_key.end() ? 0 : it->second.size()) + 1;  // its pairs + the key atom
          return n_atoms == 1 ? ((size_t)1 << 20) : n_atoms;
        };
        // selection sort of the cheapest to the front (the order of the kept requirements does not matter)
        for (size_t a = 0; a < kMaxNeed; ++a) {
          size_t best = a;
          for (size_t i = a + 1; i < n_pos; ++i)
            if (cost_of(i) < cost_of(best)) best = i;
          if (best != a) std::swap(pos_by_key[a], pos_by_key[best]);
        }
        keep0 = 0, keep1 = kMaxNeed;
        b.need = (uint32_t)kMaxNeed;
        b.slow = true;
      }
      b.n_pos = (uint32_t)(keep1 - keep0);
      b.pos_off = (uint32_t)atom_pool.size();
      for (size_t q = keep0; q < keep1; ++q) {
        b.pos_key[q - keep0] = pos_by_key[q].whole ? (int64_t)pos_by_key[q].key : -1;
        if (!pos_by_key[q].whole) atom_pool.insert(atom_pool.end(), pos_by_key[q].atoms.begin(), pos_by_key[q].atoms.end());
      }
      b.pos_cnt = (uint32_t)atom_pool.size() - b.pos_off;
      b.neg_off = (uint32_t)atom_pool.size(), b.neg_cnt = (uint32_t)neg.size();
      atom_pool.insert(atom_pool.end(), neg.begin(), neg.end());
      b.nk_off = (uint32_t)key_pool.size(), b.nk_cnt = (uint32_t)neg_keys.size();
      key_pool.insert(key_pool.end(), neg_keys.begin(), neg_keys.end());
      first_of.push_back((uint32_t)first);
      bts.push_back(b);
    }
  }
  }, nullptr);
  if (term_parts > 1) {
    size_t n_bt = 0, n_at = 0, n_ky = 0;
    for (size_t k = 0; k < term_parts; ++k) n_bt += bts_part[k].size(), n_at += atoms_part[k].size(), n_ky += keys_part[k].size();
    bts.reserve(n_bt), first_of.reserve(n_bt), atom_pool.reserve(n_at), key_pool.reserve(n_ky);
  }
  for (size_t k = 0; k < bts_part.size(); ++k) {
    const uint32_t off = (uint32_t)bts.size(), a_off = (uint32_t)atom_pool.size(), k_off = (uint32_t)key_pool.size();
    for (uint32_t f : first_part[k]) first_of.push_back(f + off);
    for (BT b : bts_part[k]) {
      b.pos_off += a_off, b.neg_off += a_off, b.nk_off += k_off;
      bts.push_back(b);
    }
    atom_pool.insert(atom_pool.end(), atoms_part[k].begin(), atoms_part[k].end());
    key_pool.insert(key_pool.end(), keys_part[k].begin(), keys_part[k].end());
  }
  lap("terms");
  out.n_pair_keys = (uint32_t)pair_keys.size();
  out.n_key_atoms = (uint32_t)key_atoms.size();
  {
    std::unordered_set<uint32_t> keys = pair_keys;
    keys.insert(key_atoms.begin(), key_atoms.end());
    out.n_keys = (uint32_t)keys.size();
  }
  // ---- groups.  For one namespace the terms of a throttle that can match are those whose namespace side admits it; the
  //      namespaces are partitioned by WHICH of the throttle's terms admit them ("cells": at most a handful per
  //      throttle — one for a namespaced Throttle).  Every cell becomes a group of term COPIES (the admitted terms, in
  //      term order, each with the cell's namespaces as its admission set).  For any pod exactly one group of a
  //      throttle is live, its copies are numbered contiguously (so the "reported once" rule of the scans holds), and
  //      groups sort by their admission set: the words of the bitmaps become class-pure even when the terms of a
  //      ClusterThrottle select different namespaces, and a pod only visits the words of classes that admit its
  //      namespace (config 4: 73 instead of 320 word steps per pod).
  std::vector<TC>& tcs = scratch.tcs;
  tcs.clear();
  // groups: their admission sets in one flat array (n_grps x nsw words: a vector per group was 40 000 small allocations
  // at 10k throttles), the index of their first copy in tcs (the copies of a group are contiguous there)
  std::vector<uint32_t>&grp_adm = scratch.grp_adm, &grp_first = scratch.grp_first;
  grp_adm.clear(), grp_first.clear();
  auto n_grps = [&]() { return grp_first.size(); };
  auto new_group = [&](const uint32_t* adm) {
    grp_first.push_back((uint32_t)tcs.size());
    const size_t o = grp_adm.size();
    grp_adm.resize(o + nsw, 0u);
    if (adm) memcpy(grp_adm.data() + o, adm, (size_t)nsw * 4);
  };
  tcs.reserve(bts.size() + bts.size() / 2);
  grp_first.reserve(bts.size());
  grp_adm.reserve((bts.size() + 16) * nsw);
  std::vector<uint64_t> cell_pat, cell_pat_sorted, cell_ord;
  std::vector<uint32_t> cell_adm;
  for (size_t i = 0; i < bts.size();) {
    size_t j = i;
    while (j < bts.size() && first_of[j] == first_of[i]) ++j;
    const size_t nt = j - i;
    bool same = true;
    for (size_t q = i + 1; q < j && same; ++q) same = memcmp_words(bts[q].adm, bts[i].adm, nsw) == 0;
    if (same || nt > 64) {
      // one group; with more than 64 terms the copies keep their own admission sets (class = that of the first)
      new_group(bts[i].adm);
      for (size_t q = i; q < j; ++q) tcs.push_back(TC{(uint32_t)q, (uint32_t)n_grps() - 1});
    } else {
      // cells by refinement: start from the union of the terms' admission sets and split every cell by every term in
      // turn (inside / outside its set) — a throttle has a handful of cells at most, so this is a few dozen word
      // operations where walking the namespaces bit by bit was a few hundred steps per throttle
      cell_adm.assign(nsw, 0u);
      for (size_t q = i; q < j; ++q)
        for (uint32_t wi = 0; wi < nsw; ++wi) cell_adm[wi] |= bts[q].adm[wi];
      if (n_ns & 31u) cell_adm[nsw - 1] &= (1u << (n_ns & 31u)) - 1u;  // (bits past the last namespace never count)
      cell_pat.assign(1, 0ull);
      for (size_t q = i; q < j; ++q) {
        const size_t nc = cell_pat.size();
        for (size_t c = 0; c < nc; ++c) {
          bool in_any = false, out_any = false;
          for (uint32_t wi = 0; wi < nsw; ++wi) {
            const uint32_t cw = cell_adm[c * nsw + wi], aw = bts[q].adm[wi];
            in_any |= (cw & aw) != 0u, out_any |= (cw & ~aw) != 0u;
          }
          if (in_any && out_any) {  // split: the part outside the term's set becomes a new cell
            cell_adm.resize(cell_adm.size() + nsw);
            for (uint32_t wi = 0; wi < nsw; ++wi) {
              const uint32_t cw = cell_adm[c * nsw + wi], aw = bts[q].adm[wi];
              cell_adm[(cell_pat.size()) * nsw + wi] = cw & ~aw, cell_adm[c * nsw + wi] = cw & aw;
            }
            cell_pat.push_back(cell_pat[c]);
            cell_pat[c] |= 1ull << (q - i);
          } else if (in_any) {
            cell_pat[c] |= 1ull << (q - i);
          }
        }
      }
      // cells in the order of their lowest namespace (the order the bit-by-bit walk created them in)
      cell_ord.resize(cell_pat.size());
      for (size_t c = 0; c < cell_pat.size(); ++c) {
        uint32_t low = ~0u;
        for (uint32_t wi = 0; wi < nsw && low == ~0u; ++wi)
          if (cell_adm[c * nsw + wi]) low = wi * 32u + (uint32_t)__builtin_ctz(cell_adm[c * nsw + wi]);
        cell_ord[c] = (uint64_t)low << 32 | (uint64_t)c;
      }
      std::sort(cell_ord.begin(), cell_ord.end());
      const size_t g0 = n_grps();
      {
        std::vector<uint64_t>& sorted_pat = cell_pat_sorted;
        sorted_pat.clear();
        for (uint64_t oc : cell_ord) {
          const size_t c = (size_t)(oc & 0xFFFFFFFFull);
          if (cell_pat[c] == 0ull) continue;  // (an empty union: no namespace in range admits any term)
          new_group(cell_adm.data() + c * nsw);  // (its first copy is set below: the copies follow once all cells are known)
          sorted_pat.push_back(cell_pat[c]);
        }
        cell_pat.swap(sorted_pat);
      }
      // copies in group order, term order inside a group
      for (size_t g = g0; g < n_grps(); ++g) {
        grp_first[g] = (uint32_t)tcs.size();
        for (size_t q = i; q < j; ++q)
          if ((cell_pat[g - g0] >> (q - i)) & 1ull) tcs.push_back(TC{(uint32_t)q, (uint32_t)g});
      }
    }
    i = j;
  }
  lap("groups (cells)");
  const size_t NG = n_grps();
  grp_first.push_back((uint32_t)tcs.size());  // sentinel: group g's copies are tcs[grp_first[g] .. grp_first[g + 1])
  auto adm_of = [&](uint32_t g) { return grp_adm.data() + (size_t)g * nsw; };
  std::vector<uint8_t> grp_own_adm(NG, 0);  // group of a >64-term throttle: copies use the term's own set
  for (size_t g = 0; g < NG; ++g) grp_own_adm[g] = grp_first[g + 1] - grp_first[g] > 64u;
  // order: groups by admission set — the LARGER sets first (round 4), then by the set's words, then by their index — the
  // copies of a group contiguous and in term order; the GROUPS are sorted, the copies follow them.  Classes that admit many
  // namespaces are the ones every scan visits: numbered side by side they share words and chunks, and the many small
  // classes no longer sit between them (configs[4]: 79.7 -> 75.3 visited words and 8.8 -> 7.8 visited chunks per namespace;
  // the order does not matter for exactness, only groups of one class have to be contiguous)
  std::vector<uint32_t>& gorder = scratch.gorder;
  gorder.assign(NG, 0u);
  {
    // The distinct admission sets are few (configs[4]: a few hundred classes for 17 000 groups): they are interned through a hash
    // table, ranked ONCE by (size descending, words ascending) — and the groups then sort by an integer key (class rank, form,
    // index): 3.8 -> 0.6 ms at 10k throttles, where every comparison of the one big sort walked the sets' words.
    // inside a class: by FORM (kNsWord*, kt_index.h) — groups without a negative requirement first, inside those the groups
    // without a need-3 term first — so that the words of a rich program come out (mostly) pure and the scans can take the
    // cheaper path per word; a program of the simple form has one form and keeps its order
    std::vector<uint8_t> form(NG, 0);
    for (uint32_t g = 0; g < NG; ++g)
      for (uint32_t q = grp_first[g]; q < grp_first[g + 1]; ++q) {
        const BT& b = bts[tcs[q].bt];
        if (b.neg_cnt != 0u || b.nk_cnt != 0u) form[g] |= 2u;
        if (b.need >= 3u) form[g] |= 1u;
      }
    size_t cap = 64;
    while (cap < 2 * NG) cap <<= 1;
    std::vector<uint32_t> slot(cap, ~0u), rep, set_of(NG, 0u);  // slot -> set id; set id -> a group that has the set; group -> set id
    for (uint32_t g = 0; g < NG; ++g) {
      const uint32_t* a = adm_of(g);
      uint64_t h = 0xCBF29CE484222325ull;
      for (uint32_t wi = 0; wi < nsw; ++wi) h = (h ^ a[wi]) * 0x100000001B3ull;
      size_t i = (size_t)(h ^ (h >> 29)) & (cap - 1);
      while (slot[i] != ~0u && memcmp_words(adm_of(rep[slot[i]]), a, nsw) != 0) i = (i + 1) & (cap - 1);
      if (slot[i] == ~0u) slot[i] = (uint32_t)rep.size(), rep.push_back(g);
      set_of[g] = slot[i];
    }
    // the size of the set and its first 32 namespaces as one 64-bit key decide most comparisons; the rest of the words
    // only on a tie
    std::vector<std::pair<uint64_t, uint32_t>> keyed(rep.size());
    for (uint32_t c = 0; c < rep.size(); ++c) {
      const uint32_t* a = adm_of(rep[c]);
      uint32_t pc = 0;
      for (uint32_t wi = 0; wi < nsw; ++wi) pc += (uint32_t)__builtin_popcount(a[wi]);
      keyed[c] = {(uint64_t)(0xFFFFFFFFu - pc) << 32 | (nsw >= 1 ? a[0] : 0u), c};
    }
    std::sort(keyed.begin(), keyed.end(), [&](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) {
      if (x.first != y.first) return x.first < y.first;
      return nsw > 1 && memcmp_words(adm_of(rep[x.second]) + 1, adm_of(rep[y.second]) + 1, nsw - 1) < 0;
    });
    std::vector<uint32_t> rank(rep.size());
    for (uint32_t i = 0; i < rep.size(); ++i) rank[keyed[i].second] = i;
    std::vector<uint64_t> gkey(NG);  // (class rank: 30 bits, form: 2 bits, group index: 32 bits)
    for (uint32_t g = 0; g < NG; ++g) gkey[g] = (uint64_t)rank[set_of[g]] << 34 | (uint64_t)form[g] << 32 | g;
    std::sort(gkey.begin(), gkey.end());
    for (uint32_t g = 0; g < NG; ++g) gorder[g] = (uint32_t)gkey[g];
  }
  std::vector<uint32_t>& order = scratch.order;
  order.clear();
  order.reserve(tcs.size());
  for (uint32_t g : gorder)
    for (uint32_t q = grp_first[g]; q < grp_first[g + 1]; ++q) order.push_back(q);
  if (dbg_time) fprintf(stderr, "  build_index: (%zu groups, %zu copies, %zu terms)\n", NG, tcs.size(), bts.size());
  lap("sort by admission set");
  // term numbers: a class (run of groups with the same admission set) never straddles a 64-bit word of the bitmaps
  // unless it is larger than one (128 for big programs: fewer, fuller words per namespace)
  const uint32_t gran = tcs.size() <= 4096 ? 64u : 128u;
  std::vector<uint32_t>&num = scratch.num, &cls = scratch.cls;  // term number and class (= admission set) of every copy
  num.assign(tcs.size(), 0u), cls.assign(tcs.size(), 0u);
  uint32_t pos = 0, n_cls = 0;
  for (size_t i = 0; i < order.size();) {
    size_t j = i;
    while (j < order.size() && (tcs[order[j]].grp == tcs[order[i]].grp || memcmp_words(adm_of(tcs[order[j]].grp), adm_of(tcs[order[i]].grp), nsw) == 0)) ++j;
    for (size_t q = i; q < j; ++q) cls[order[q]] = n_cls;
    ++n_cls;
    const uint32_t sz = (uint32_t)(j - i);
    if ((pos & (gran - 1)) != 0 && ((pos & (gran - 1)) + sz > gran)) pos = (pos + gran - 1) & ~(gran - 1);
    for (size_t q = i; q < j;) {
      // the copies of one group never straddle a 64-bit word when they fit one: every word boundary then is a place
      // where a chunk may be cut
      size_t q1 = q;
      while (q1 < j && tcs[order[q1]].grp == tcs[order[q]].grp) ++q1;
      const uint32_t nt = (uint32_t)(q1 - q);
      if (nt <= 64 && (pos & 63u) + nt > 64u) pos = (pos + 63u) & ~63u;
      for (; q < q1; ++q) num[order[q]] = pos++;
    }
    i = j;
  }
  const uint32_t G2 = pos;
  const uint32_t W = G2 ? (G2 + 63) / 64 : 1u;  // 64-bit words per full row (an empty program keeps one zero word)
  out.bm_words = W;
  // ---- atoms -> ids (= bitmap rows; row 0 = no atom)
  {
    // the distinct atoms of the kept terms: the atom pool holds exactly their pair atoms (with many repeats: a program
    // names a few hundred distinct pairs tens of thousands of times — a hash set instead of sorting the whole pool)
    std::unordered_set<uint32_t> distinct, whole_keys;  // whole_keys: keys some kept term names at key level
    distinct.reserve(4096);
    for (uint32_t a : atom_pool) distinct.insert(a);
    for (uint32_t k : key_pool) whole_keys.insert(k);
    for (auto& b : bts) {
      for (uint32_t i = 0; i < b.n_pos; ++i)
        if (b.pos_key[i] >= 0) whole_keys.insert((uint32_t)b.pos_key[i]);
      out.has_veto |= b.neg_cnt != 0u || b.nk_cnt != 0u;
      out.has_slow |= b.slow;
      out.max_need = std::max(out.max_need, b.need);  // (a `slow` term counts its three kept positives like any other)
    }
    for (uint32_t k : whole_keys)
      for (uint32_t a : whole_key(k)) distinct.insert(a);
    std::vector<uint32_t> atoms(distinct.begin(), distinct.end());
    out.la = atom_slots(out.n_keys, max_labels);
    // the simple instantiation <8 atoms, no veto family, need <= 2> covers matchLabels-style programs; everything else
    // takes the rich one, whose images carry a veto plane behind the `any` plane
    out.rich = out.has_veto || out.has_slow || out.max_need > 2 || out.la != 8;
    std::sort(atoms.begin(), atoms.end());
    atoms.erase(std::unique(atoms.begin(), atoms.end()), atoms.end());
    if (atoms_from) {
      out.atoms = atoms_from->atoms, out.atom_key = atoms_from->atom_key;
      out.la = atoms_from->la, out.rich = true;
    } else {
      std::unordered_map<uint32_t, uint32_t> key_of_pair;
      for (auto& kv : pairs_of_key)
        for (uint32_t a : kv.second) key_of_pair.emplace(a, kv.first);
      for (uint32_t i = 0; i < atoms.size(); ++i) {
        out.atoms.push_back(AtomId{atoms[i], 0u, 0u});
        out.atom_key.push_back((atoms[i] & kKeyAtom) ? (atoms[i] & ~kKeyAtom) : key_of_pair[atoms[i]]);
      }
      number_atoms_by_home_slot(out);
    }
  }
  const uint32_t A = (uint32_t)out.atoms.size();
  uint32_t R = 1;  // (an imposed numbering may have holes)
  for (const AtomId& ai : out.atoms) R = std::max(R, ai.id + 1u);
  out.bm_rows = R;
  std::unordered_map<uint32_t, uint32_t> row_of;
  row_of.reserve(A * 2 + 1);
  for (const AtomId& ai : out.atoms) row_of.emplace(ai.atom, ai.id);
  {  // device translation table: open addressing, linear probing, load factor <= 1/2
    size_t n = 16;
    while (n < (size_t)A * 2 + 2) n <<= 1;
    out.atom_table.assign(n, 0ull);
    for (const AtomId& ai : out.atoms) {
      uint32_t s = atom_slot(ai.atom, (uint32_t)n - 1);
      while (out.atom_table[s] != 0ull) s = (s + 1) & ((uint32_t)n - 1);
      out.atom_table[s] = (uint64_t)ai.atom | (uint64_t)ai.id << 32 | (uint64_t)ai.home << kAtomHomeShift;
    }
  }
  lap("numbering + atom ids");
  // ---- full bitmaps (host only), dense throttle ranks in term order
  const bool veto = out.rich;  // the image form follows the kernel instantiation
  // (kept in the HostIndex: cut_chunks() lays them out as chunk images, and can do so again for other LDS budgets)
  std::vector<uint64_t>&any = out.full_any, &vet = out.full_veto, &nsrows = out.full_nsrows;
  any.assign((size_t)R * W, 0ull), vet.assign(veto ? (size_t)R * W : 0, 0ull), nsrows.assign((size_t)n_ns * W, 0ull);
  std::vector<WordHdr>& hdr = out.full_hdr;
  hdr.assign(W, WordHdr{0, 0, 0, 0, 0, 0, {0, 0}});
  std::vector<uint32_t>&term_t = out.full_term_t, &term_g = out.full_term_g, &term_rank = out.full_term_rank;
  term_t.assign((size_t)W * 64, 0u), term_g.assign((size_t)W * 64, 0u), term_rank.assign((size_t)W * 64, 0u);
  std::vector<uint8_t>& real = out.full_real;  // term number in use (not padding)
  real.assign((size_t)W * 64, 0);
  out.n_ns = n_ns;
  out.bm_rank_t.clear();
  lap("  (bitmaps: zero fill)");
  // bitmap rows: of every key named at key level (all its atoms), and of every term's pair atoms
  std::unordered_map<uint32_t, std::vector<uint32_t>> rows_of_key;
  for (auto& b : bts)
    for (uint32_t i = 0; i < b.n_pos; ++i)
      if (b.pos_key[i] >= 0) rows_of_key.emplace((uint32_t)b.pos_key[i], std::vector<uint32_t>());
  for (uint32_t k : key_pool) rows_of_key.emplace(k, std::vector<uint32_t>());
  for (auto& kv : rows_of_key)
    for (uint32_t a : whole_key(kv.first)) kv.second.push_back(row_of[a]);
  lap("  (bitmaps: rows of keys)");
  std::vector<uint32_t>& pool_row = scratch.pool_row;  // bitmap row of every pooled pair atom
  pool_row.assign(atom_pool.size(), 0u);
  parallel_for(atom_pool.size(), 32768, [&](size_t q0, size_t q1, size_t) {
    const uint32_t mask = (uint32_t)out.atom_table.size() - 1u;
    for (size_t q = q0; q < q1; ++q) {  // the device's translation table (open addressing) is the faster map here too
      uint32_t sl = atom_slot(atom_pool[q], mask);
      while (out.atom_table[sl] != 0ull && (uint32_t)out.atom_table[sl] != atom_pool[q]) sl = (sl + 1) & mask;
      // (an imposed numbering may lack a pair this program names — an anchor veto on a value no term that can match names:
      //  no pod carries it as an atom, row 0 = "no row")
      pool_row[q] = (uint32_t)(out.atom_table[sl] >> 32) & kAtomIdMask;
    }
  }, nullptr);
  {
    std::vector<uint32_t>& by_num = scratch.by_num;
    by_num.assign(G2, ~0u);
    for (size_t q = 0; q < tcs.size(); ++q) by_num[num[q]] = (uint32_t)q;
    // dense ranks: one per GROUP in number order (a throttle with several cells has several; the slab reduction adds
    // them all into the throttle's row) — the one sequential pass
    uint32_t last_g = ~0u;
    for (uint32_t c = 0; c < G2; ++c) {
      if (by_num[c] == ~0u) continue;  // padding
      const TC& tc = tcs[by_num[c]];
      if (tc.grp != last_g) out.bm_rank_t.push_back(bts[tc.bt].t), last_g = tc.grp;
      term_rank[c] = (uint32_t)out.bm_rank_t.size() - 1;
    }
    lap("  (bitmaps: pool rows + ranks)");
    // the bits: every 64-bit word column of the bitmaps belongs to one range of term numbers — ranges on several threads
    parallel_for(W, 160, [&](size_t w_begin, size_t w_end, size_t) {
      // the namespace rows are set once per RUN of copies of one class inside a word (they share the admission set)
      // instead of once per copy
      uint32_t run_cls = ~0u;
      uint64_t run_bits = 0;
      size_t run_w = 0;
      const uint32_t* run_adm = nullptr;
      auto flush = [&]() {
        if (run_bits && run_adm)
          for (uint32_t wi = 0; wi < nsw; ++wi)
            for (uint32_t m = run_adm[wi]; m; m &= m - 1) {
              const uint32_t n = wi * 32u + (uint32_t)__builtin_ctz(m);
              if (n < n_ns) nsrows[(size_t)n * W + run_w] |= run_bits;
            }
        run_bits = 0, run_cls = ~0u, run_adm = nullptr;
      };
      for (uint32_t c = (uint32_t)w_begin * 64u; c < std::min<uint32_t>((uint32_t)w_end * 64u, G2); ++c) {
        if ((c & 63u) == 0u) flush();
        if (by_num[c] == ~0u) continue;  // padding
        const uint32_t qi = by_num[c];
        const TC& tc = tcs[qi];
        const BT& b = bts[tc.bt];
        real[c] = 1;
        const uint64_t bit = 1ull << (c & 63);
        const size_t w = c >> 6;
        term_t[c] = b.t | kTermReal | (b.adj ? kTermAdj : 0u);
        term_g[c] = b.g;
        if (b.n_pos == 0u) hdr[w].univ |= bit;
        for (uint32_t q = b.pos_off; q < b.pos_off + b.pos_cnt; ++q)
          if (pool_row[q]) any[(size_t)pool_row[q] * W + w] |= bit;
        for (uint32_t q = b.neg_off; q < b.neg_off + b.neg_cnt; ++q)
          if (pool_row[q]) vet[(size_t)pool_row[q] * W + w] |= bit;
        for (uint32_t i = 0; i < b.n_pos; ++i)
          if (b.pos_key[i] >= 0)
            for (uint32_t r : rows_of_key.find((uint32_t)b.pos_key[i])->second) any[(size_t)r * W + w] |= bit;
        for (uint32_t q = b.nk_off; q < b.nk_off + b.nk_cnt; ++q)
          for (uint32_t r : rows_of_key.find(key_pool[q])->second) vet[(size_t)r * W + w] |= bit;
        if (b.need >= 2) hdr[w].m2 |= bit;
        if (b.need >= 3) hdr[w].m3 |= bit;
        if (b.need >= 4) hdr[w].m4 |= bit;
        if (b.need >= 5) hdr[w].m5 |= bit;
        if (b.slow) hdr[w].slow |= bit;
        if (grp_own_adm[tc.grp]) {  // a copy of a >64-term throttle: its own admission set
          flush();
          run_adm = b.adm, run_w = w, run_bits = bit;
          flush();
        } else {
          if (cls[qi] != run_cls) flush();
          run_cls = cls[qi], run_adm = adm_of(tc.grp), run_w = w, run_bits |= bit;
        }
      }
      flush();
    }, nullptr);
  }
  lap("full bitmaps");
  {
    // what is not kept for the next build (BuildScratch) and is large goes to a helper thread to be freed
    struct Trash {
      std::vector<std::vector<BT>> bts_part;
      std::vector<std::vector<uint32_t>> atoms_part, keys_part;
      std::unordered_map<uint32_t, std::vector<uint32_t>> pairs_of_key, rows_of_key;
    };
    Trash* trash = new Trash{std::move(bts_part), std::move(atoms_part), std::move(keys_part), std::move(pairs_of_key), std::move(rows_of_key)};
    std::thread([trash] { delete trash; }).detach();
  }
  lap("hand-off of the scratch containers");
  // chunks for chk_budget (two check workgroups per CU) — unless the caller wants larger chunks when the program needs
  // several anyway (chk_budget_full) and it plainly does: then only that cut is made
  const size_t rows_bytes = (size_t)image_col_rows(R) * W * 8u * (veto ? 2u : 1u);
  if (chk_budget_full && rows_bytes > (size_t)chk_budget) cut_chunks(out, agg_budget, chk_budget_full, thr_bytes, chk_word, thr_bytes_packed);
  else cut_chunks(out, agg_budget, chk_budget, thr_bytes, chk_word, thr_bytes_packed);
  lap("cut_chunks");
}

// Cuts the numbered bitmaps of `out` (build_index) into chunk images for the given LDS budgets; callable again with
// other budgets without renumbering (the engine first asks for half-LDS chunks — two check workgroups per CU — and
// re-cuts for the full LDS when the program needs several chunks anyway).
//
// Two plans (round 6).  A chunk is a LIST of words of the numbered program plus the namespaces whose word lists it
// carries; a word may sit in several chunks (the images are copies in HBM), a namespace meets every word it visits in
// exactly ONE of the chunks that serve it.
//   * global  : consecutive word ranges, every chunk serves every namespace that has a word in it (rounds 1-5) — the default;
//   * grouped : the namespaces are grouped by the words they visit, and every group gets its own run of chunks that
//               holds exactly the words its members visit — the words most of them share first.  A namespace-ordered
//               scan (a workgroup = a range of namespaces) then stages as many images as the words of ITS namespaces
//               need instead of every chunk of the global numbering that holds one of them.  Built to VERDICT r5 #1 and
//               MEASURED on the configs[4] shard (256 namespaces in 4 zones, a namespace visits 75 of 606 words): chunk
//               passes per namespace 7.0 -> 3.7 (126 chunks instead of 29), results identical — and the check went from
//               0.379 to 0.365 ms, the aggregate from 0.458 to 0.454, while the slab reduction (a block per chunk and
//               record tile) went from 0.024 to 0.100: the step 0.851 -> 0.908 ms (profiles/r06_cut_plans.txt).  The
//               per-(tile, chunk) work — barrier, image staged, records fetched again, carry words — overlaps with the
//               other waves' scans; what bounds both kernels is VALU issue in the scan itself (2.0e8 wave instructions per
//               check launch = 0.33 ms at one instruction per four cycles and SIMD).  KT_CUT_PLAN=grouped selects the plan
//               (A/B runs, tests/cpp/index_sim_test.cpp replays it on the host); the default stays global.
// What IS kept of the round: the veto plane holds only the columns of words that have a veto bit (kt_index.h), namespaces
// with the same word list share it, and a program of several chunks is cut for the packed fold's 40-byte records: 29
// chunks instead of 35 on the shard, 0.884 -> 0.851 ms.
namespace {

struct ChunkPlan {
  std::vector<uint32_t> words;   // words of the numbered program, in chunk-local order
  std::vector<uint32_t> served;  // namespaces whose word lists the chunk carries, ascending; `all`: every namespace
  bool all = true;
};

struct Cutter {
  HostIndex& out;
  const uint32_t W, R, Rp, n_ns;
  const bool veto;
  const size_t fam;
  const uint32_t agg_budget, chk_budget, chk_word;
  uint32_t thr_bytes;  // the record size the aggregate's tables are sized for (cut_chunks tries the plain one, then the packed fold's)
  std::vector<uint8_t> splittable;    // a chunk may START at word w
  std::vector<uint32_t> word_form;    // NsWord::flags of every word
  std::vector<uint32_t> word_groups;  // groups (= ranks = slab records) whose numbers start in word w
  std::vector<std::vector<uint32_t>> ns_words;  // words every namespace visits, ascending
  bool all_splittable = true;
  bool windowed = false;  // the aggregate may scan a chunk once per window of ranks (kt_kernels_aggregate.hip): only a minimal table counts

  Cutter(HostIndex& o, uint32_t agg, uint32_t chk, uint32_t tb, uint32_t cw)
      : out(o), W(o.bm_words), R(o.bm_rows), Rp(image_col_rows(o.bm_rows)), n_ns(o.n_ns), veto(o.rich), fam(o.rich ? 2 : 1),
        agg_budget(agg), chk_budget(chk), chk_word(cw), thr_bytes(tb) {
    const std::vector<uint32_t>& term_rank = out.full_term_rank;
    const std::vector<uint8_t>& real = out.full_real;
    splittable.assign(W + 1, 1);
    for (uint32_t w = 1; w < W; ++w) {
      // the last real term of word w-1 and the first real term of word w belong to different throttles?
      int64_t a = -1, b2 = -1;
      for (int k = 63; k >= 0 && a < 0; --k)
        if (real[(size_t)(w - 1) * 64 + k]) a = (int64_t)(w - 1) * 64 + k;
      for (int k = 0; k < 64 && b2 < 0; ++k)
        if (real[(size_t)w * 64 + k]) b2 = (int64_t)w * 64 + k;
      if (a >= 0 && b2 >= 0 && term_rank[a] == term_rank[b2]) splittable[w] = 0, all_splittable = false;
    }
    // the form of every word (NsWord::flags): does some atom row hold a veto bit in it, does some term need three hits
    word_form.assign(W, 0u);
    for (uint32_t w = 0; w < W; ++w) word_form[w] = out.full_hdr[w].m3 != 0ull ? kNsWordNeed3 : 0u;
    if (veto)
      for (uint32_t r = 0; r < R; ++r)
        for (uint32_t w = 0; w < W; ++w)
          if (out.full_veto[(size_t)r * W + w]) word_form[w] |= kNsWordVeto;
    word_groups.assign(W, 0u);
    {
      uint32_t last = ~0u;
      for (size_t c = 0; c < (size_t)W * 64; ++c)
        if (real[c] && term_rank[c] != last) last = term_rank[c], ++word_groups[c >> 6];
    }
    ns_words.assign(n_ns, {});
    for (uint32_t n = 0; n < n_ns; ++n)
      for (uint32_t w = 0; w < W; ++w)
        if (out.full_nsrows[(size_t)n * W + w]) ns_words[n].push_back(w);
  }

  // LDS bytes of the image part of a chunk of nw words, nv of them with a veto column, whose word lists hold nsl_entries entries
  size_t image_lds(size_t nw, size_t nv, size_t nsl_entries) const {
    const size_t cols = nw + (veto ? nv + (nv < nw ? 1 : 0) : 0);
    return (size_t)Rp * cols * 8 + nw * sizeof(WordHdr) + align16((size_t)n_ns * 8) + nsl_entries * sizeof(NsWord);
  }
  // does word w get a column in the veto plane (a program with a run of numbers across a word boundary keeps them all)
  bool has_veto(uint32_t w) const { return veto && (!all_splittable || (word_form[w] & kNsWordVeto) != 0u); }
  // does a chunk fit both kernels' budgets next to the maxima of the chunks planned so far?  The kernels lay LDS out ONCE
  // for all chunks — the largest image next to the tables of the largest chunk — so a chunk has to fit together with those
  // maxima, not only on its own.  (aggregate: ranks u16[64] + the run masks {seg_lo, seg_hi} per word; nw < 1024: the
  // packed fold queues chunk-local word numbers as 10-bit values)
  struct Maxima { size_t lds = 0, thr = 0, nw = 0; };
  bool fits(size_t lds, size_t nw, size_t nthr, const Maxima& mx) const {
    const size_t lds_hi = std::max(lds, mx.lds), nw_hi = std::max(nw, mx.nw);
    const size_t thr_hi = windowed ? std::min<size_t>(std::max(nthr, mx.thr), 256) : std::max(nthr, mx.thr);
    return lds_hi + nw_hi * chk_word <= chk_budget && lds_hi + align16(nw_hi * 64 * 2) + nw_hi * 16 + thr_hi * thr_bytes + 16 <= agg_budget &&
           nthr < 0x8000u && nw < 1024;
  }

  // ---- plan: consecutive word ranges, cut at the last boundary that splits no throttle
  std::vector<ChunkPlan> plan_global() const {
    std::vector<ChunkPlan> plan;
    const std::vector<uint32_t>& term_rank = out.full_term_rank;
    const std::vector<uint8_t>& real = out.full_real;
    std::vector<uint32_t> ns_per_word(W, 0u);
    for (uint32_t n = 0; n < n_ns; ++n)
      for (uint32_t w : ns_words[n]) ++ns_per_word[w];
    Maxima mx;
    uint32_t w0 = 0;
    while (w0 < W) {
      // grow the chunk word by word while it fits; cut at the last boundary that splits no throttle
      uint32_t w1 = 0;
      size_t nsl_entries = 0, lds1 = 0, thr1 = 0, nv = 0;
      uint32_t r_lo = ~0u, r_hi = 0;
      for (uint32_t cand = w0 + 1; cand <= W; ++cand) {
        nsl_entries += ns_per_word[cand - 1], nv += has_veto(cand - 1);
        for (size_t c = (size_t)(cand - 1) * 64; c < (size_t)cand * 64; ++c)
          if (real[c]) r_lo = std::min(r_lo, term_rank[c]), r_hi = std::max(r_hi, term_rank[c]);
        const uint32_t nthr = r_lo == ~0u ? 0 : r_hi - r_lo + 1;
        const size_t nw = cand - w0, lds = image_lds(nw, nv, nsl_entries);
        const bool ok = fits(lds, nw, nthr, mx);
        if (!ok && w1 != 0) break;
        if (cand == W || splittable[cand]) {
          w1 = cand, lds1 = lds, thr1 = nthr;
          if (!ok) break;  // a single stretch larger than the budget: the launchers notice
        }
      }
      ChunkPlan cp;
      for (uint32_t w = w0; w < w1; ++w) cp.words.push_back(w);
      mx.lds = std::max(mx.lds, lds1), mx.thr = std::max(mx.thr, thr1), mx.nw = std::max(mx.nw, (size_t)(w1 - w0));
      plan.push_back(std::move(cp));
      w0 = w1;
    }
    return plan;
  }

  // ---- plan: chunks per group of namespaces.  `cnt[w]` = members of the group that visit word w.  The group's words go
  // into its chunks in the order (members that visit the word, descending; word number): what most members share comes
  // first, the words of single members last.  chunks_of() cuts that sequence greedily; mx == nullptr: against the budgets
  // alone (estimates), else against the maxima of the plan so far, which it then raises.
  size_t chunks_of(const std::vector<uint32_t>& words, const std::vector<uint16_t>& cnt, size_t n_members, Maxima* mx, std::vector<uint32_t>* cuts) const {
    // (word-list entries: the lists of namespaces that visit the same words of a chunk with the same masks are stored once —
    //  a chunk of words ALL members visit is charged one list, any other chunk one entry per (member, visited word); emit()
    //  measures the real image and splits a chunk that turns out larger)
    size_t n_chunks = 0, nw = 0, nv = 0, sum = 0, nthr = 0;
    bool full = true;
    const Maxima none;
    auto entries = [&](size_t nw_, size_t sum_, bool full_) { return full_ ? nw_ : sum_; };
    for (size_t i = 0; i < words.size(); ++i) {
      const uint32_t w = words[i];
      const bool full1 = full && cnt[w] >= n_members;
      if (nw && !fits(image_lds(nw + 1, nv + has_veto(w), entries(nw + 1, sum + cnt[w], full1)), nw + 1, nthr + word_groups[w], mx ? *mx : none)) {
        if (mx) mx->lds = std::max(mx->lds, image_lds(nw, nv, entries(nw, sum, full))), mx->thr = std::max(mx->thr, nthr), mx->nw = std::max(mx->nw, nw);
        if (cuts) cuts->push_back((uint32_t)i);
        ++n_chunks, nw = 0, nv = 0, sum = 0, nthr = 0, full = true;
      }
      ++nw, nv += has_veto(w), sum += cnt[w], nthr += word_groups[w], full = full && cnt[w] >= n_members;
    }
    if (nw) {
      if (mx) mx->lds = std::max(mx->lds, image_lds(nw, nv, entries(nw, sum, full))), mx->thr = std::max(mx->thr, nthr), mx->nw = std::max(mx->nw, nw);
      if (cuts) cuts->push_back((uint32_t)words.size());
      ++n_chunks;
    }
    return n_chunks;
  }
  std::vector<ChunkPlan> plan_grouped() const {
    struct Group {
      std::vector<uint32_t> members, words;  // words: those with cnt > 0, unordered
      std::vector<uint16_t> cnt;
      size_t ideal = ~(size_t)0;  // the FEWEST chunks a member would need by itself: nobody joins at the price of a pass a member would not make alone
    };
    std::vector<Group> groups;
    std::vector<uint16_t> one(W, 0);
    std::vector<uint32_t> seq;
    auto ordered = [&](const std::vector<uint32_t>& words, const std::vector<uint16_t>& cnt) {
      seq = words;
      std::sort(seq.begin(), seq.end(), [&](uint32_t a, uint32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
    };
    constexpr size_t kOpen = 24;  // groups a namespace is tried against (the most recent ones)
    for (uint32_t n = 0; n < n_ns; ++n) {
      const std::vector<uint32_t>& nw = ns_words[n];
      if (nw.empty()) continue;
      for (uint32_t w : nw) one[w] = 1;
      ordered(nw, one);
      const size_t ideal_n = chunks_of(seq, one, 1, nullptr, nullptr);
      for (uint32_t w : nw) one[w] = 0;
      // the open group that shares the most words with n among those n can join without costing ANY member a chunk pass
      // it would not make alone (ties: the most recent group — neighbours in namespace order share workgroups)
      size_t best = ~(size_t)0, best_shared = 0;
      for (size_t k = groups.size(); k-- > 0 && groups.size() - k <= kOpen;) {
        Group& g = groups[k];
        if (g.members.size() >= 0xFFFFu) continue;
        size_t shared = 0;
        for (uint32_t w : nw) shared += g.cnt[w] != 0;
        if (best != ~(size_t)0 && shared <= best_shared) continue;
        std::vector<uint32_t> uni = g.words;
        for (uint32_t w : nw)
          if (!g.cnt[w]) uni.push_back(w);
        for (uint32_t w : nw) ++g.cnt[w];
        ordered(uni, g.cnt);
        const size_t c = chunks_of(seq, g.cnt, g.members.size() + 1, nullptr, nullptr);
        for (uint32_t w : nw) --g.cnt[w];
        if (c <= std::min(g.ideal, ideal_n)) best = k, best_shared = shared;
      }
      if (best == ~(size_t)0) {
        groups.emplace_back();
        groups.back().cnt.assign(W, 0);
        best = groups.size() - 1;
      }
      Group& g = groups[best];
      g.members.push_back(n);
      for (uint32_t w : nw)
        if (g.cnt[w]++ == 0) g.words.push_back(w);
      g.ideal = std::min(g.ideal, ideal_n);
    }
    std::vector<ChunkPlan> plan;
    Maxima mx;
    for (Group& g : groups) {
      ordered(g.words, g.cnt);
      std::vector<uint32_t> cuts;
      (void)chunks_of(seq, g.cnt, g.members.size(), &mx, &cuts);
      size_t i0 = 0;
      for (uint32_t i1 : cuts) {
        ChunkPlan cp;
        cp.all = false, cp.served = g.members;
        cp.words.assign(seq.begin() + i0, seq.begin() + i1);
        std::sort(cp.words.begin(), cp.words.end());  // (chunk-local order = word order: the lists stay ascending)
        plan.push_back(std::move(cp));
        i0 = i1;
      }
    }
    return plan;
  }
  // chunks a namespace-ordered scan stages per namespace, summed over the namespaces that visit anything
  size_t visits(const std::vector<ChunkPlan>& plan) const {
    size_t v = 0;
    std::vector<uint8_t> in_chunk(W, 0);
    for (const ChunkPlan& cp : plan) {
      for (uint32_t w : cp.words) in_chunk[w] = 1;
      auto touches = [&](uint32_t n) {
        for (uint32_t w : ns_words[n])
          if (in_chunk[w]) return true;
        return false;
      };
      if (cp.all) {
        for (uint32_t n = 0; n < n_ns; ++n) v += touches(n);
      } else {
        for (uint32_t n : cp.served) v += touches(n);
      }
      for (uint32_t w : cp.words) in_chunk[w] = 0;
    }
    return v;
  }

  // ---- images.  A chunk whose real image turns out larger than the plan's estimate allowed (word lists that could not
  // be shared) is cut in two.
  void emit(const std::vector<ChunkPlan>& plan_in) {
    const std::vector<uint64_t>&any = out.full_any, &vet = out.full_veto, &nsrows = out.full_nsrows;
    const std::vector<WordHdr>& hdr = out.full_hdr;
    const std::vector<uint32_t>&term_t = out.full_term_t, &term_g = out.full_term_g, &term_rank = out.full_term_rank;
    const std::vector<uint8_t>& real = out.full_real;
    uint64_t slab_run = 0;
    out.bm_chunks.clear();
    out.bm_images.clear();
    size_t total_words = 0;
    for (const ChunkPlan& cp : plan_in) total_words += cp.words.size();
    out.bm_images.reserve(((size_t)Rp * total_words * 8 * fam + total_words * (sizeof(WordHdr) + 64 * 10 + 64)) * 5 / 4 + plan_in.size() * ((size_t)n_ns + 8) * 4 * 4);  // (one allocation instead of a regrowth per chunk)
    out.bm_chunk_ns.clear();
    const uint32_t nsw = (n_ns + 31) / 32;
    out.ns_words = nsw ? nsw : 1u;
    out.bm_max_lds = 0, out.bm_max_thr = 0, out.bm_max_words = 0, out.bm_slab_bytes = 0;
    out.bm_rank_t.clear();
    out.ns_word_visits = 0, out.ns_chunk_visits = 0;
    uint32_t w_run = 0;
    Maxima mx;
    std::vector<ChunkPlan> work(plan_in.rbegin(), plan_in.rend());  // (a stack: the next chunk on top)
    std::vector<uint64_t> irows;
    std::vector<uint32_t> words;
    while (!work.empty()) {
      ChunkPlan cp = std::move(work.back());
      work.pop_back();
      BmChunk ch{};
      const uint32_t nw = (uint32_t)cp.words.size();
      // chunk-local word order: the words with a veto column first (their columns are the veto plane), word order inside
      // both kinds.  (A program with a run of numbers across a word boundary keeps its order and every veto column.)
      words = cp.words;
      uint32_t nv = 0;
      if (veto) {
        std::stable_partition(words.begin(), words.end(), [&](uint32_t w) { return has_veto(w); });  // (a no-op when every word has one)
        for (uint32_t w : words) nv += has_veto(w);
      }
      const uint32_t veto_cols = veto ? nv + (nv < nw ? 1u : 0u) : 0u;
      ch.n_words = nw, ch.n_veto = nv;
      ch.zero_col = veto && nv < nw ? (nw + nv) * Rp * 8u : 0u;
      ch.ns_base = 0u, ch.ns_cnt = 0xFFFFFFFFu;
      ch.col_rows = Rp;
      // the word lists; namespaces with the same list share it
      std::vector<uint32_t> nsl_rng((size_t)n_ns * 2, 0u);
      std::vector<NsWord> nsl, cur;
      {
        std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;  // list hash -> begins of the lists stored so far
        size_t si = 0;
        for (uint32_t n = 0; n < n_ns; ++n) {
          bool serve = cp.all;
          if (!cp.all) {
            while (si < cp.served.size() && cp.served[si] < n) ++si;
            serve = si < cp.served.size() && cp.served[si] == n;
          }
          if (!serve) continue;
          cur.clear();
          uint64_t h = 1469598103934665603ull;
          for (uint32_t w = 0; w < nw; ++w) {
            const uint64_t m = nsrows[(size_t)n * W + words[w]];
            if (!m) continue;
            cur.push_back(NsWord{w, word_form[words[w]], m});
            h = (h ^ w) * 1099511628211ull, h = (h ^ m) * 1099511628211ull;
          }
          if (cur.empty()) continue;
          uint32_t begin = ~0u;
          std::vector<uint32_t>& cands = by_hash[h];
          for (uint32_t b0 : cands) {
            // (a stored list that starts at b0 and equals `cur` entry by entry — its end is not stored: compare the length through
            //  the entries themselves, the next list or the end of the array follows)
            if ((size_t)b0 + cur.size() > nsl.size()) continue;
            bool same = true;
            for (size_t q = 0; q < cur.size() && same; ++q) same = nsl[b0 + q].w == cur[q].w && nsl[b0 + q].mask == cur[q].mask;
            if (same) { begin = b0; break; }
          }
          if (begin == ~0u) {
            begin = (uint32_t)nsl.size();
            nsl.insert(nsl.end(), cur.begin(), cur.end());
            cands.push_back(begin);
          }
          nsl_rng[(size_t)n * 2] = begin, nsl_rng[(size_t)n * 2 + 1] = begin + (uint32_t)cur.size();
        }
      }
      // per-number tables; dense ranks of the chunk's groups in chunk-local number order
      const uint32_t rank0 = (uint32_t)out.bm_rank_t.size();
      std::vector<uint32_t> it((size_t)nw * 64, 0u), ig((size_t)nw * 64, 0u);
      std::vector<uint16_t> ir((size_t)nw * 64, 0);
      {
        uint32_t last_rank = ~0u;
        for (uint32_t w = 0; w < nw; ++w)
          for (uint32_t k = 0; k < 64; ++k) {
            const size_t c = (size_t)words[w] * 64 + k;
            it[(size_t)w * 64 + k] = term_t[c], ig[(size_t)w * 64 + k] = term_g[c];
            if (!real[c]) continue;
            if (term_rank[c] != last_rank) last_rank = term_rank[c], out.bm_rank_t.push_back(term_t[c] & kTermRowMask);
            ir[(size_t)w * 64 + k] = (uint16_t)(((uint32_t)out.bm_rank_t.size() - 1u - rank0) | ((term_t[c] & kTermAdj) ? kRankAdj : 0u));
            ch.has_slow |= (hdr[c >> 6].slow >> (c & 63)) & 1ull ? 1u : 0u;
            ch.has_adj |= (term_t[c] & kTermAdj) ? 1u : 0u;
          }
      }
      ch.n_thr = (uint32_t)out.bm_rank_t.size() - rank0;
      ch.rank0 = ch.n_thr ? rank0 : 0u;
      const size_t lds = image_lds(nw, nv, nsl.size());
      if (!fits(lds, nw, ch.n_thr, mx) && nw > 1 && all_splittable) {
        // larger than the plan thought: two halves (of the plan's order) take its place
        out.bm_rank_t.resize(rank0);
        ChunkPlan lo = cp, hi = cp;
        lo.words.assign(cp.words.begin(), cp.words.begin() + nw / 2), hi.words.assign(cp.words.begin() + nw / 2, cp.words.end());
        work.push_back(std::move(hi)), work.push_back(std::move(lo));
        continue;
      }
      ch.w0 = w_run;
      w_run += nw;
      // planes of word columns: any[nw][Rp], then (rich) veto[veto_cols][Rp] — the columns of local words [0, nv) and the zero column
      irows.assign((size_t)Rp * (nw + veto_cols), 0ull);
      const size_t veto_plane = (size_t)Rp * nw;
      for (uint32_t r = 0; r < R; ++r)
        for (uint32_t w = 0; w < nw; ++w) {
          irows[(size_t)w * Rp + r] = any[(size_t)r * W + words[w]];
          if (veto && w < nv) irows[veto_plane + (size_t)w * Rp + r] = vet[(size_t)r * W + words[w]];
        }
      std::vector<WordHdr> ihdr(nw);
      for (uint32_t w = 0; w < nw; ++w) ihdr[w] = hdr[words[w]];
      for (uint32_t n = 0; n < n_ns; ++n) out.ns_word_visits += (int64_t)(nsl_rng[(size_t)n * 2 + 1] - nsl_rng[(size_t)n * 2]);  // (pairs, whatever lists are shared)
      {  // which namespaces have words here (the namespace-ordered scans skip the chunk for workgroups without any of them)
        const size_t b0 = out.bm_chunk_ns.size();
        out.bm_chunk_ns.resize(b0 + out.ns_words, 0u);
        for (uint32_t n = 0; n < n_ns; ++n)
          if (nsl_rng[(size_t)n * 2 + 1] > nsl_rng[(size_t)n * 2]) out.bm_chunk_ns[b0 + (n >> 5)] |= 1u << (n & 31), ++out.ns_chunk_visits;
      }
      const void* src[7] = {irows.data(), ihdr.data(), nsl_rng.data(), nsl.data(), it.data(), ir.data(), ig.data()};
      const size_t bytes[7] = {irows.size() * 8, ihdr.size() * sizeof(WordHdr), nsl_rng.size() * 4, nsl.size() * sizeof(NsWord),
                               it.size() * 4,    ir.size() * 2,                 ig.size() * 4};
      uint32_t* offs[7] = {nullptr, &ch.off_hdr, &ch.off_nsl_rng, &ch.off_nsl, &ch.off_term_t, &ch.off_term_rank, &ch.off_term_g};
      size_t o = 0;
      const size_t img0 = out.bm_images.size();
      for (int k = 0; k < 7; ++k) {
        if (offs[k]) *offs[k] = (uint32_t)o;
        if (k == 4) ch.lds_bytes = (uint32_t)o;  // everything before term_t lives in LDS
        o += align16(bytes[k]);
      }
      out.bm_images.resize(img0 + o, 0);
      size_t oo = 0;
      for (int k = 0; k < 7; ++k) {
        if (bytes[k]) memcpy(out.bm_images.data() + img0 + oo, src[k], bytes[k]);
        oo += align16(bytes[k]);
      }
      ch.img_off = (uint32_t)img0, ch.img_bytes = (uint32_t)o;
      ch.slab_off = (uint32_t)(slab_run / 16);  // one table per (chunk, workgroup), 256 workgroups at most
      // ... or 512 workgroups with packed records of half the size (PackPlan): their rounding to 16 bytes needs the slack
      slab_run += 256ull * (((uint64_t)ch.n_thr * thr_bytes + 15) & ~15ull) + 8192ull;
      mx.lds = std::max(mx.lds, (size_t)ch.lds_bytes), mx.thr = std::max(mx.thr, (size_t)ch.n_thr), mx.nw = std::max(mx.nw, (size_t)nw);
      out.bm_max_lds = std::max(out.bm_max_lds, ch.lds_bytes);
      out.bm_max_thr = std::max(out.bm_max_thr, ch.n_thr);
      out.bm_max_words = std::max(out.bm_max_words, ch.n_words);
      out.bm_chunks.push_back(ch);
      if (getenv("KT_DEBUG_CHUNKS"))
        fprintf(stderr, "chunk %zu: %u words (%u with veto columns; first %u) lds %u B thr %u, serves %zu namespaces, %zu list entries\n", out.bm_chunks.size() - 1,
                nw, nv, nw ? words[0] : 0u, ch.lds_bytes, ch.n_thr, cp.all ? (size_t)n_ns : cp.served.size(), nsl.size());
      out.bm_slab_bytes = slab_run;
    }
    out.img_words = w_run;
  }
};

}  // namespace

void cut_chunks(HostIndex& out, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, uint32_t chk_word, uint32_t thr_bytes_packed) {
  out.cut_chk_budget = chk_budget;
  out.cut_thr_bytes = thr_bytes;
  out.cut_grouped = false;
  const char* force = getenv("KT_CUT_PLAN");
  const bool want_global = force && !strcmp(force, "global"), want_grouped = force && !strcmp(force, "grouped");
  out.agg_windowed = false;
  Cutter cut(out, agg_budget, chk_budget, thr_bytes, chk_word);  // (one pass over the numbered program; the plans below share it)
  std::vector<ChunkPlan> plan = cut.plan_global();
  if (plan.size() > 1) {
    // Only the aggregate's table stands against ONE chunk (the image and the check's tables fit)?  Then the program stays
    // in one chunk and the aggregate scans it once per window of ranks: the check keeps its single-chunk form — two
    // workgroups per CU, no namespace order, no carry words — and the reconcile pays a second scan instead of a second
    // chunk (a 16-dimension engine at 1M x 1k: check 67 -> 32 us, round 6).
    cut.windowed = true;
    std::vector<ChunkPlan> p1 = cut.plan_global();
    if (p1.size() == 1 && !getenv("KT_NO_AGG_WINDOW")) {
      out.agg_windowed = true;
      cut.emit(p1);  // (still `windowed`: emit holds the real image against the same rule)
      return;
    }
    cut.windowed = false;
  }
  if (plan.size() <= 1 || !cut.all_splittable) {
    cut.emit(plan);
    return;
  }
  // several chunks: the tables (and slabs) of the aggregate are sized for the packed fold's records where the caller offers
  // that — more words per chunk, fewer chunk passes
  const uint32_t tb = thr_bytes_packed && thr_bytes_packed < thr_bytes && !getenv("KT_CUT_PLAIN") ? thr_bytes_packed : thr_bytes;
  out.cut_thr_bytes = tb;
  if (tb != thr_bytes) cut.thr_bytes = tb, plan = cut.plan_global();
  (void)want_global;
  if (want_grouped || getenv("KT_CUT_PLAN_AUTO")) {
    std::vector<ChunkPlan> grouped = cut.plan_grouped();
    const size_t v_global = cut.visits(plan), v_grouped = cut.visits(grouped);
    if (getenv("KT_DEBUG_CHUNKS"))
      fprintf(stderr, "cut_chunks: global %zu chunks / %zu namespace visits, grouped %zu chunks / %zu namespace visits\n", plan.size(), v_global,
              grouped.size(), v_grouped);
    // (the slab scratch — a table per chunk and workgroup — and the tag table grow with the chunk count: a plan of thousands
    //  of chunks is not worth its visits)
    if (want_grouped || (v_grouped * 5 <= v_global * 4 && grouped.size() <= std::max<size_t>(8 * plan.size(), 64) && grouped.size() <= 2048))
      plan.swap(grouped), out.cut_grouped = true;
  }
  cut.emit(plan);
}

template <class T>
static hipError_t up(T*& dev, size_t& cap, const std::vector<T>& h, hipStream_t s) {
  const size_t need = h.size() + 1;
  if (need > cap || !dev) {
    if (dev) (void)hipFree(dev);
    dev = nullptr;
    cap = 0;
    hipError_t e = kt_alloc_device((void**)&dev, need * sizeof(T));
    if (e != hipSuccess) return e;
    cap = need;
  }
  if (!h.empty()) return hipMemcpyAsync(dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  return hipSuccess;
}

void index_group_counts(HostIndex& h, uint32_t T) {
  h.thr_ngrp.assign(T, 0u);
  for (uint32_t t : h.bm_rank_t)
    if (t < T) ++h.thr_ngrp[t];
  h.nogroup.clear();
  for (uint32_t t = 0; t < T; ++t)
    if (!h.thr_ngrp[t]) h.nogroup.push_back(t);
}

hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s) {
  hipError_t e;
  if ((e = up(d.thr_ngrp, d.cap_thr_ngrp, h.thr_ngrp, s)) != hipSuccess) return e;
  if ((e = up(d.nogroup, d.cap_nogroup, h.nogroup, s)) != hipSuccess) return e;
  d.n_nogroup = (uint32_t)h.nogroup.size();
  {
    const std::vector<uint32_t> zeros(h.thr_ngrp.size() + 1, 0u);
    if ((e = up(d.grp_arrive, d.cap_grp_arrive, zeros, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;  // `zeros` goes out of scope
  }
  if ((e = up(d.slow_thr, d.cap_slow, h.slow_thr, s)) != hipSuccess) return e;
  d.n_slow = (uint32_t)h.slow_thr.size();
  if ((e = up(d.bm_blob, d.cap_bm_blob, h.bm_images, s)) != hipSuccess) return e;
  if ((e = up(d.bm_chunks, d.cap_bm_chunks, h.bm_chunks, s)) != hipSuccess) return e;
  if ((e = up(d.bm_rank_t, d.cap_bm_rank_t, h.bm_rank_t, s)) != hipSuccess) return e;
  if ((e = up(d.bm_chunk_ns, d.cap_bm_chunk_ns, h.bm_chunk_ns, s)) != hipSuccess) return e;
  d.ns_words = h.ns_words;
  if ((e = up(d.atom_table, d.cap_atom_table, h.atom_table, s)) != hipSuccess) return e;
  d.atom_mask = (uint32_t)h.atom_table.size() - 1;
  d.h_chunks = h.bm_chunks;
  d.n_chunks = (uint32_t)h.bm_chunks.size();
  d.bm_max_lds = h.bm_max_lds, d.bm_max_thr = h.bm_max_thr, d.bm_max_words = h.bm_max_words, d.bm_rows = h.bm_rows;
  d.bm_words = h.img_words;
  d.cut_thr_bytes = h.cut_thr_bytes;
  d.has_long = h.has_long;
  d.bm_slab_bytes = h.bm_slab_bytes;
  d.has_veto = h.has_veto ? 1u : 0u;
  d.max_need = h.max_need;
  d.n_atoms = (uint32_t)h.atoms.size();
  d.has_key_atoms = h.n_key_atoms ? 1u : 0u;
  d.la = h.la, d.rich = h.rich;
  if (getenv("KT_DEBUG_LDS"))
    fprintf(stderr, "bitmap index: %u words, %u rows (veto %d, need %u), %zu chunks, largest LDS part %u B, largest chunk %u throttles / %u words, %zu slow throttles\n",
            h.bm_words, h.bm_rows, (int)h.has_veto, h.max_need, h.bm_chunks.size(), h.bm_max_lds, h.bm_max_thr, h.bm_max_words, h.slow_thr.size());
  return hipSuccess;
}

void release_index(IndexDev& d) {
  if (d.slow_thr) (void)hipFree(d.slow_thr);
  if (d.bm_blob) (void)hipFree(d.bm_blob);
  if (d.bm_chunks) (void)hipFree(d.bm_chunks);
  if (d.bm_rank_t) (void)hipFree(d.bm_rank_t);
  if (d.thr_ngrp) (void)hipFree(d.thr_ngrp);
  if (d.nogroup) (void)hipFree(d.nogroup);
  if (d.grp_arrive) (void)hipFree(d.grp_arrive);
  if (d.bm_chunk_ns) (void)hipFree(d.bm_chunk_ns);
  if (d.atom_table) (void)hipFree(d.atom_table);
  d = IndexDev();
}

PackPlan make_pack_plan(int D, const unsigned __int128* max_abs, const uint64_t* or_abs, bool neg_seen, uint64_t n_slab_pods, bool pad_odd,
                        uint32_t max_words) {
  PackPlan pk;
  if (neg_seen || D < 1 || D > 16 || n_slab_pods == 0) return pk;
  const int MW = (int)std::min<uint32_t>(max_words ? max_words : 1u, kPackMaxWords);
  auto bitlen = [](unsigned __int128 x) { int b = 0; while (x) ++b, x >>= 1; return b; };
  constexpr int H = kPackHeadroomBits;
  // fields in placement order per word: dimension (-1: the pod count), position, width
  struct F { int d; uint32_t pos, w; };
  std::vector<F> fields[kPackMaxWords];
  uint32_t fill[kPackMaxWords] = {0, 0, 0, 0, 0, 0, 0, 0};
  // every field at least H bits (it is its lower neighbour's headroom) and at most 64 - H (its own sum over the slabs)
  pk.cnt_width = (uint8_t)std::max(bitlen(n_slab_pods), H);
  if (pk.cnt_width > 64 - H) return PackPlan();
  fields[0].push_back({-1, 0u, pk.cnt_width});
  fill[0] = pk.cnt_width;
  uint32_t nw = 1;
  for (int d = 0; d < D; ++d) {
    if (max_abs[d] == 0) continue;  // no pod carries a non-zero value here: no field
    const int sh = or_abs[d] ? __builtin_ctzll(or_abs[d]) : 0;
    const int w = std::max(bitlen((max_abs[d] >> sh) * (unsigned __int128)n_slab_pods), H);
    if (w > 64 - H) return PackPlan();
    int k = 0;
    while (k < MW && fill[k] + (uint32_t)w > 64u) ++k;
    if (k == MW) return PackPlan();
    pk.word[d] = (uint8_t)k, pk.pos[d] = (uint8_t)fill[k], pk.width[d] = (uint8_t)w, pk.shift[d] = (uint8_t)sh;
    fields[k].push_back({d, fill[k], (uint32_t)w});
    fill[k] += (uint32_t)w;
    nw = std::max(nw, (uint32_t)k + 1u);
  }
  for (uint32_t k = 0; k < nw; ++k) {
    const size_t n = fields[k].size();  // >= 1: a word only exists because a field went there
    pk.top_pos[k] = (uint8_t)fields[k][n - 1].pos;
    for (size_t i = 0; i < n; ++i) {
      const F& f = fields[k][i];
      const bool top = i + 1 == n;
      const uint32_t cls = top ? 2u : (uint32_t)(i & 1u);
      if (cls == 0u) pk.even[k] |= ((1ull << f.w) - 1ull) << f.pos;
      const uint32_t desc = pack_desc(k, cls, top ? 0u : f.pos, f.w + (uint32_t)H, f.d >= 0 ? pk.shift[f.d] : 0u);
      if (f.d >= 0) pk.desc[f.d] = desc; else pk.cnt_desc = desc;
    }
  }
  pk.nw = nw;
  pk.stride = nw <= 2 ? 2u : nw <= 4 ? 4u : 8u;
  uint32_t units = nw + 1u;
  if (pad_odd && !(units & 1u)) ++units;
  pk.rec_bytes = units * 8u;
  return pk;
}

}  // namespace kt
