// kt_index.cpp — host-side construction and upload of the label-atom -> term index (see kt_index.h).
#include "kt_index.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>

#include "../../include/kt_snapshot.h"

namespace kt {

void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns,
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw) {
  out = HostIndex();
  const size_t T = thr_term_off.empty() ? 0 : thr_term_off.size() - 1;
  std::unordered_map<uint64_t, std::vector<Posting>> lists;
  std::vector<std::vector<uint32_t>> uni_ns(n_ns);

  for (size_t t = 0; t < T; ++t) {
    const ThrInfo ti = thr_info((uint32_t)t);
    if (!ti.live) continue;
    // a reachable unconvertible podSelector makes term ORDER matter: walk this throttle densely
    bool slow = false;
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g)
      if ((term_flags[g] & KT_TERM_POD_SEL_INVALID) && !(ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID))) slow = true;
    if (slow) {
      out.slow_thr.push_back((uint32_t)t);
      continue;
    }
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g) {
      if (ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID)) continue;  // never matches any namespace
      if (!ti.cluster && ti.ns >= n_ns) continue;
      // anchor: the In requirement with the fewest values, else an Exists requirement
      int64_t best = -1;
      size_t best_cost = ~(size_t)0;
      for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
        size_t cost;
        if (req_op[r] == KT_OP_IN) cost = req_val_off[r + 1] - req_val_off[r];
        else if (req_op[r] == KT_OP_EXISTS) cost = 1u << 20;
        else continue;
        if (cost < best_cost) best_cost = cost, best = r;
      }
      const uint64_t scope = ti.cluster ? 0ull : (uint64_t)(ti.ns + 1);
      if (best < 0) {
        if (ti.cluster) out.uni_cluster.push_back(g);
        else uni_ns[ti.ns].push_back(g);
        continue;
      }
      // inline description of the term for the posting fast path
      Posting base{};
      base.g = g;
      base.t = (uint32_t)t;
      if (thr_term_off[t + 1] - thr_term_off[t] > 1) base.flags |= kPostMulti;
      if (ti.cluster) {
        if (n_ns <= 64) {
          base.flags |= kPostNsMask;
          for (uint32_t n = 0; n < n_ns; ++n)
            if ((ns_term_ok[(size_t)n * gw + (g >> 5)] >> (g & 31)) & 1u) base.nsmask |= 1ull << n;
        } else {
          base.flags |= kPostNsBitmap;
        }
      }
      const uint32_t n_req = term_req_off[g + 1] - term_req_off[g];
      bool simple = n_req <= 2 && req_op[best] == KT_OP_IN;
      uint32_t other_pair = 0;
      for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1] && simple; ++r) {
        if (req_op[r] != KT_OP_IN || req_val_off[r + 1] - req_val_off[r] != 1) simple = false;
        else if ((int64_t)r != best) other_pair = req_val[req_val_off[r]];
      }
      if (!simple) base.flags |= kPostComplex;
      else if (n_req == 2) {
        base.flags |= kPostPair2;
        base.pair2 = other_pair;
      }
      if (req_op[best] == KT_OP_IN) {
        std::vector<uint32_t> vals(req_val.begin() + req_val_off[best], req_val.begin() + req_val_off[best + 1]);
        std::sort(vals.begin(), vals.end());
        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
        for (uint32_t v : vals) {
          lists[scope << 32 | v].push_back(base);  // an empty value set files nothing: never matches
        }
      } else {
        lists[scope << 32 | (kKeyAtom | req_key[best])].push_back(base);
        out.has_key_atoms = true;
      }
    }
  }
  size_t n_slots = 16;
  while (n_slots < lists.size() * 2 + 1) n_slots <<= 1;
  out.slots.assign(n_slots, IndexSlot{0, 0, 0});
  out.mask = (uint32_t)(n_slots - 1);
  // deterministic order: sort keys
  std::vector<uint64_t> keys;
  keys.reserve(lists.size());
  for (auto& kv : lists) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  for (uint64_t k : keys) {
    const std::vector<Posting>& l = lists[k];
    uint32_t h = index_hash(k, out.mask);
    while (out.slots[h].key != 0) h = (h + 1) & out.mask;
    out.slots[h] = IndexSlot{k, (uint32_t)out.postings.size(), (uint32_t)l.size()};
    out.postings.insert(out.postings.end(), l.begin(), l.end());
  }
  // ---- bitmap form (all indexed terms; see kt_index.h).  Built when it can live in LDS next to the
  //      kernel's working buffers; otherwise the kernels use the postings above.
  {
    struct BT {
      uint32_t g, t, pair2, flags;
      std::vector<uint32_t> atoms;  // anchor atoms (empty => universal)
      std::vector<uint32_t> adm;    // namespace admission set as words
    };
    std::vector<BT> bts;
    const uint32_t nsw = (n_ns + 31) / 32;
    for (size_t t = 0; t < T; ++t) {
      const ThrInfo ti = thr_info((uint32_t)t);
      if (!ti.live) continue;
      if (std::find(out.slow_thr.begin(), out.slow_thr.end(), (uint32_t)t) != out.slow_thr.end()) continue;
      for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g) {
        if (ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID)) continue;
        if (!ti.cluster && ti.ns >= n_ns) continue;
        BT b;
        b.g = g, b.t = (uint32_t)t, b.pair2 = 0, b.flags = 0;
        b.adm.assign(nsw, 0u);
        bool any_ns = false;
        for (uint32_t n = 0; n < n_ns; ++n)
          if ((ns_term_ok[(size_t)n * gw + (g >> 5)] >> (g & 31)) & 1u) b.adm[n >> 5] |= 1u << (n & 31), any_ns = true;
        if (!any_ns) continue;  // admitted nowhere: can never match
        if (thr_term_off[t + 1] - thr_term_off[t] > 1) b.flags |= kPostMulti;
        int64_t best = -1;
        size_t best_cost = ~(size_t)0;
        for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r) {
          size_t cost;
          if (req_op[r] == KT_OP_IN) cost = req_val_off[r + 1] - req_val_off[r];
          else if (req_op[r] == KT_OP_EXISTS) cost = 1u << 20;
          else continue;
          if (cost < best_cost) best_cost = cost, best = r;
        }
        const uint32_t n_req = term_req_off[g + 1] - term_req_off[g];
        if (best < 0) {
          if (n_req != 0) b.flags |= kPostComplex;  // only negative requirements
        } else {
          bool simple = n_req <= 2 && req_op[best] == KT_OP_IN;
          uint32_t other = 0;
          for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1] && simple; ++r) {
            if (req_op[r] != KT_OP_IN || req_val_off[r + 1] - req_val_off[r] != 1) simple = false;
            else if ((int64_t)r != best) other = req_val[req_val_off[r]];
          }
          if (!simple) b.flags |= kPostComplex;
          else if (n_req == 2) b.flags |= kPostPair2, b.pair2 = other;
          if (req_op[best] == KT_OP_IN) {
            b.atoms.assign(req_val.begin() + req_val_off[best], req_val.begin() + req_val_off[best + 1]);
            std::sort(b.atoms.begin(), b.atoms.end());
            b.atoms.erase(std::unique(b.atoms.begin(), b.atoms.end()), b.atoms.end());
            if (b.atoms.empty()) continue;  // In with no values: never matches
          } else {
            b.atoms.push_back(kKeyAtom | req_key[best]);
          }
        }
        bts.push_back(std::move(b));
      }
    }
    // terms with the same admission set become contiguous
    std::stable_sort(bts.begin(), bts.end(), [](const BT& a, const BT& b) { return a.adm < b.adm; });
    // term numbers: a class never straddles a 64-bit word unless it is larger than one
    std::vector<uint32_t> num(bts.size());
    uint32_t pos = 0;
    for (size_t i = 0; i < bts.size();) {
      size_t j = i;
      while (j < bts.size() && bts[j].adm == bts[i].adm) ++j;
      const uint32_t sz = (uint32_t)(j - i);
      if ((pos & 63u) != 0 && ((pos & 63u) + sz > 64u)) pos = (pos + 63u) & ~63u;
      for (size_t q = i; q < j; ++q) num[q] = pos++;
      i = j;
    }
    const uint32_t G2 = pos;
    if (G2 > 0) {
      const uint32_t W = (G2 + 63) / 64;
      out.bm_words = W;
      out.bm_stride = W | 1u;
      std::unordered_map<uint32_t, uint32_t> row_of;
      for (auto& b : bts)
        for (uint32_t a : b.atoms) row_of.emplace(a, 0u);
      std::vector<uint32_t> atoms;
      for (auto& kv : row_of) atoms.push_back(kv.first);
      std::sort(atoms.begin(), atoms.end());
      for (uint32_t i = 0; i < atoms.size(); ++i) row_of[atoms[i]] = i + 2;
      out.bm_rows = (uint32_t)atoms.size() + 2;
      out.bm_row_bits.assign((size_t)out.bm_rows * out.bm_stride, 0ull);
      out.bm_nsrows.assign((size_t)n_ns * out.bm_stride, 0ull);
      out.bm_trec.assign(G2, TermRec{0, 0, 0, 0});
      for (size_t q = 0; q < bts.size(); ++q) {
        const BT& b = bts[q];
        const uint32_t c = num[q];
        const uint64_t bit = 1ull << (c & 63);
        out.bm_trec[c] = TermRec{b.g, b.t, b.pair2, b.flags};
        if (b.atoms.empty()) out.bm_row_bits[c >> 6] |= bit;
        for (uint32_t a : b.atoms) out.bm_row_bits[(size_t)row_of[a] * out.bm_stride + (c >> 6)] |= bit;
        for (uint32_t n = 0; n < n_ns; ++n)
          if ((b.adm[n >> 5] >> (n & 31)) & 1u) out.bm_nsrows[(size_t)n * out.bm_stride + (c >> 6)] |= bit;
      }
      out.bm_nswords_off.assign((size_t)n_ns + 1, 0u);
      for (uint32_t n = 0; n < n_ns; ++n) {
        for (uint32_t w = 0; w < W; ++w)
          if (out.bm_nsrows[(size_t)n * out.bm_stride + w]) out.bm_nswords.push_back(w);
        out.bm_nswords_off[n + 1] = (uint32_t)out.bm_nswords.size();
      }
      // atoms -> rows in 4-entry buckets; grow until no bucket overflows
      size_t nb = 4;
      while (nb * 2 < atoms.size()) nb <<= 1;
      for (;;) {
        out.bm_buckets.assign(nb, AtomBucket{{0, 0, 0, 0}, {1, 1, 1, 1}});
        bool ok = true;
        for (uint32_t a : atoms) {
          AtomBucket& bk = out.bm_buckets[atom_bucket(a, (uint32_t)nb - 1)];
          int k = 0;
          while (k < 4 && bk.atom[k] != 0) ++k;
          if (k == 4) { ok = false; break; }
          bk.atom[k] = a;
          bk.row[k] = row_of[a];
        }
        if (ok) break;
        nb <<= 1;
      }
      out.bm_bucket_mask = (uint32_t)nb - 1;
    }
  }
  out.uni_ns_off.assign((size_t)n_ns + 1, 0);
  for (uint32_t n = 0; n < n_ns; ++n) {
    out.uni_ns.insert(out.uni_ns.end(), uni_ns[n].begin(), uni_ns[n].end());
    out.uni_ns_off[n + 1] = (uint32_t)out.uni_ns.size();
  }
}

template <class T>
static hipError_t up(T*& dev, size_t& cap, const std::vector<T>& h, hipStream_t s) {
  const size_t need = h.size() + 1;
  if (need > cap || !dev) {
    if (dev) (void)hipFree(dev);
    dev = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void**)&dev, need * sizeof(T));
    if (e != hipSuccess) return e;
    cap = need;
  }
  if (!h.empty()) return hipMemcpyAsync(dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  return hipSuccess;
}

hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s) {
  hipError_t e;
  if ((e = up(d.slots, d.cap_slots, h.slots, s)) != hipSuccess) return e;
  if ((e = up(d.postings, d.cap_postings, h.postings, s)) != hipSuccess) return e;
  if ((e = up(d.uni_ns_off, d.cap_uni_ns_off, h.uni_ns_off, s)) != hipSuccess) return e;
  if ((e = up(d.uni_ns, d.cap_uni_ns, h.uni_ns, s)) != hipSuccess) return e;
  if ((e = up(d.uni_cluster, d.cap_uni_cluster, h.uni_cluster, s)) != hipSuccess) return e;
  if ((e = up(d.slow_thr, d.cap_slow, h.slow_thr, s)) != hipSuccess) return e;
  d.mask = h.mask;
  d.n_uni_cluster = (uint32_t)h.uni_cluster.size();
  d.n_slow = (uint32_t)h.slow_thr.size();
  d.has_key_atoms = h.has_key_atoms ? 1u : 0u;
  d.n_slots = (uint32_t)h.slots.size();
  d.n_postings = (uint32_t)h.postings.size();
  d.n_cluster_postings = 0;
  for (const IndexSlot& sl : h.slots)
    if (sl.key != 0 && (sl.key >> 32) == 0) d.n_cluster_postings += sl.count;
  d.bm_words = h.bm_words;
  d.bm_stride = h.bm_stride;
  d.bm_bucket_mask = h.bm_bucket_mask;
  d.bm_blob_bytes = 0;
  if (h.bm_words != 0) {
    // pack the bitmap tables into one blob (the LDS image)
    const void* src[6] = {h.bm_row_bits.data(), h.bm_nsrows.data(), h.bm_nswords_off.data(),
                          h.bm_nswords.data(), h.bm_buckets.data(), h.bm_trec.data()};
    const size_t bytes[6] = {h.bm_row_bits.size() * 8, h.bm_nsrows.size() * 8, h.bm_nswords_off.size() * 4,
                             h.bm_nswords.size() * 4, h.bm_buckets.size() * sizeof(AtomBucket),
                             h.bm_trec.size() * sizeof(TermRec)};
    size_t o = 0;
    for (int k = 0; k < 6; ++k) d.bm_off[k] = (uint32_t)o, o += (bytes[k] + 15) & ~(size_t)15;
    std::vector<unsigned char> blob(o + 16, 0);
    for (int k = 0; k < 6; ++k)
      if (bytes[k]) memcpy(blob.data() + d.bm_off[k], src[k], bytes[k]);
    if ((e = up(d.bm_blob, d.cap_bm_blob, blob, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;  // `blob` is a temporary
    d.bm_blob_bytes = (uint32_t)o;
  }
  return hipSuccess;
}

void release_index(IndexDev& d) {
  if (d.slots) (void)hipFree(d.slots);
  if (d.postings) (void)hipFree(d.postings);
  if (d.uni_ns_off) (void)hipFree(d.uni_ns_off);
  if (d.uni_ns) (void)hipFree(d.uni_ns);
  if (d.uni_cluster) (void)hipFree(d.uni_cluster);
  if (d.slow_thr) (void)hipFree(d.slow_thr);
  if (d.bm_blob) (void)hipFree(d.bm_blob);
  d = IndexDev();
}

}  // namespace kt
