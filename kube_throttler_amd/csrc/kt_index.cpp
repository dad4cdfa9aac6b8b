// kt_index.cpp — host-side construction and upload of the label-atom -> term index (see kt_index.h).
#include "kt_index.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
  // a reachable unconvertible podSelector makes term ORDER matter: these throttles are walked term by term
  for (size_t t = 0; t < T; ++t) {
    const ThrInfo ti = thr_info((uint32_t)t);
    if (!ti.live) continue;
    for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g)
      if ((term_flags[g] & KT_TERM_POD_SEL_INVALID) && !(ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID))) {
        out.slow_thr.push_back((uint32_t)t);
        break;
      }
  }
  // ---- bitmap form (all indexed terms; see kt_index.h): LDS-resident when it fits, else read through L2.
  {
    struct BT {
      uint32_t g, t, pair2, flags;
      std::vector<uint32_t> atoms;  // anchor atoms (empty => universal)
      std::vector<uint32_t> adm;    // namespace admission set as words
      TermX x;                      // inline extra requirements (kPostInline)
    };
    std::vector<BT> bts;
    std::vector<uint32_t> first_of;  // index into bts of the first term of the same throttle
    const uint32_t nsw = (n_ns + 31) / 32;
    const TermX no_extras = {{{0xFFu, kNoAtom, kNoAtom, kNoAtom}, {0xFFu, kNoAtom, kNoAtom, kNoAtom}}};
    for (size_t t = 0; t < T; ++t) {
      const ThrInfo ti = thr_info((uint32_t)t);
      if (!ti.live) continue;
      if (std::find(out.slow_thr.begin(), out.slow_thr.end(), (uint32_t)t) != out.slow_thr.end()) continue;
      const size_t first = bts.size();
      for (uint32_t g = thr_term_off[t]; g < thr_term_off[t + 1]; ++g) {
        if (ti.cluster && (term_flags[g] & KT_TERM_NS_SEL_INVALID)) continue;
        if (!ti.cluster && ti.ns >= n_ns) continue;
        BT b;
        b.g = g, b.t = (uint32_t)t, b.pair2 = 0, b.flags = 0, b.x = no_extras;
        b.adm.assign(nsw, 0u);
        bool any_ns = false;
        for (uint32_t n = 0; n < n_ns; ++n)
          if ((ns_term_ok[(size_t)n * gw + (g >> 5)] >> (g & 31)) & 1u) b.adm[n >> 5] |= 1u << (n & 31), any_ns = true;
        if (!any_ns) continue;  // admitted nowhere: can never match
        // several terms: the terms of a throttle are numbered contiguously and a lane meets its candidates in
        // ascending number, so "reported once, by its first matching term" = drop a match that repeats the
        // lane's previous throttle
        if (thr_term_off[t + 1] - thr_term_off[t] > 1) b.flags |= kPostAdj;
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
        if (best >= 0) {
          if (req_op[best] == KT_OP_IN) {
            b.atoms.assign(req_val.begin() + req_val_off[best], req_val.begin() + req_val_off[best + 1]);
            std::sort(b.atoms.begin(), b.atoms.end());
            b.atoms.erase(std::unique(b.atoms.begin(), b.atoms.end()), b.atoms.end());
            if (b.atoms.empty()) continue;  // In with no values: never matches
          } else {
            b.atoms.push_back(kKeyAtom | req_key[best]);
            out.bm_has_key_rows = true;
          }
        }
        // the other requirements: one extra single-valued In -> pair2; up to two small ones -> inline; else generic
        std::vector<uint32_t> extras;
        for (uint32_t r = term_req_off[g]; r < term_req_off[g + 1]; ++r)
          if ((int64_t)r != best) extras.push_back(r);
        if (best >= 0 && req_op[best] == KT_OP_IN && n_req == 2 && req_op[extras[0]] == KT_OP_IN &&
            req_val_off[extras[0] + 1] - req_val_off[extras[0]] == 1) {
          b.flags |= kPostPair2;
          b.pair2 = req_val[req_val_off[extras[0]]];
        } else if (!extras.empty()) {
          bool fits = extras.size() <= 2;
          for (size_t k = 0; k < extras.size() && fits; ++k) {
            const uint32_t r = extras[k];
            const uint32_t nv = req_val_off[r + 1] - req_val_off[r];
            if (req_op[r] == KT_OP_IN || req_op[r] == KT_OP_NOT_IN) {
              if (nv > 3) fits = false;
              else {
                b.x.e[k][0] = req_op[r];
                for (uint32_t j = 0; j < nv; ++j) b.x.e[k][1 + j] = req_val[req_val_off[r] + j];
              }
            } else {
              b.x.e[k][0] = req_op[r];
              b.x.e[k][1] = req_key[r];
            }
          }
          if (fits) b.flags |= kPostInline;
          else b.flags |= kPostComplex, b.x = no_extras;
        }
        first_of.push_back((uint32_t)first);
        bts.push_back(std::move(b));
      }
    }
    // order: throttles by the admission set of their FIRST term (namespaces then touch few blocks as long as the
    // terms of a throttle agree on it — always for namespaced Throttles), terms of a throttle contiguous
    std::vector<uint32_t> order(bts.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return bts[first_of[a]].adm < bts[first_of[b]].adm; });
    // term numbers: a class (run of throttles with the same first admission set) never straddles a step of the
    // scan unless it is larger than one — 64-bit words for programs small enough for the LDS form, 128-bit blocks
    // beyond (the L2 form reads 16 bytes per request)
    const uint32_t gran = bts.size() <= 4096 ? 64u : 128u;
    std::vector<uint32_t> num(bts.size());
    uint32_t pos = 0;
    for (size_t i = 0; i < order.size();) {
      size_t j = i;
      while (j < order.size() && bts[first_of[order[j]]].adm == bts[first_of[order[i]]].adm) ++j;
      const uint32_t sz = (uint32_t)(j - i);
      if ((pos & (gran - 1)) != 0 && ((pos & (gran - 1)) + sz > gran)) pos = (pos + gran - 1) & ~(gran - 1);
      for (size_t q = i; q < j; ++q) num[order[q]] = pos++;
      i = j;
    }
    const uint32_t G2 = pos;
    {
      const uint32_t W = G2 ? (G2 + 63) / 64 : 1u;  // 64-bit words per row (an empty program keeps one zero word)
      const uint32_t NB = (W + 1) / 2;               // 128-bit blocks (the L2 form masks the missing half of the last)
      out.bm_words = W;
      out.bm_stride = W | 1u;  // odd: column reads spread over the LDS banks
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
      out.bm_trec.assign(G2 ? G2 : 1u, TermRec{0, 0, 0, 0});
      bool any_inline = false;
      for (auto& b : bts) any_inline |= (b.flags & kPostInline) != 0;
      if (any_inline) out.bm_trecx.assign(G2, no_extras);
      for (size_t q = 0; q < bts.size(); ++q) {
        const BT& b = bts[q];
        const uint32_t c = num[q];
        const uint64_t bit = 1ull << (c & 63);
        out.bm_trec[c] = TermRec{b.g, b.t, b.pair2, b.flags};
        if (any_inline) out.bm_trecx[c] = b.x;
        if (b.atoms.empty()) out.bm_row_bits[c >> 6] |= bit;
        for (uint32_t a : b.atoms) out.bm_row_bits[(size_t)row_of[a] * out.bm_stride + (c >> 6)] |= bit;
        for (uint32_t n = 0; n < n_ns; ++n)
          if ((b.adm[n >> 5] >> (n & 31)) & 1u) out.bm_nsrows[(size_t)n * out.bm_stride + (c >> 6)] |= bit;
      }
      out.bm_nswords_off.assign((size_t)n_ns + 1, 0u);
      for (uint32_t n = 0; n < n_ns; ++n) {
        for (uint32_t blk = 0; blk < NB; ++blk) {
          uint64_t any = out.bm_nsrows[(size_t)n * out.bm_stride + 2 * blk];
          if (2 * blk + 1 < W) any |= out.bm_nsrows[(size_t)n * out.bm_stride + 2 * blk + 1];
          if (any) out.bm_nswords.push_back(blk);
        }
        out.bm_nswords_off[n + 1] = (uint32_t)out.bm_nswords.size();
      }
      out.bm_nswords64_off.assign((size_t)n_ns + 1, 0u);
      for (uint32_t n = 0; n < n_ns; ++n) {
        for (uint32_t w = 0; w < W; ++w)
          if (out.bm_nsrows[(size_t)n * out.bm_stride + w]) out.bm_nswords64.push_back(w);
        out.bm_nswords64_off[n + 1] = (uint32_t)out.bm_nswords64.size();
      }
      // atoms -> rows in 4-entry buckets: a few multipliers per size, then double, until no bucket overflows
      size_t nb = 4;
      while (nb * 3 < atoms.size()) nb <<= 1;
      uint32_t mult = 0x9E3779B1u;
      for (int attempt = 0;; ++attempt) {
        out.bm_buckets.assign(nb, AtomBucket{{0, 0, 0, 0}, {1, 1, 1, 1}});
        bool ok = true;
        for (uint32_t a : atoms) {
          AtomBucket& bk = out.bm_buckets[atom_bucket(a, (uint32_t)nb - 1, mult)];
          int k = 0;
          while (k < 4 && bk.atom[k] != 0) ++k;
          if (k == 4) { ok = false; break; }
          bk.atom[k] = a;
          bk.row[k] = row_of[a];
        }
        if (ok) break;
        if (attempt % 24 == 23) nb <<= 1;
        mult = mult * 0x01000193u + 0x9E3779B9u;
        mult |= 1u;
      }
      out.bm_bucket_mult = mult;
      out.bm_bucket_mask = (uint32_t)nb - 1;
    }
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
  if ((e = up(d.slow_thr, d.cap_slow, h.slow_thr, s)) != hipSuccess) return e;
  d.n_slow = (uint32_t)h.slow_thr.size();
  d.bm_words = h.bm_words;
  d.bm_stride = h.bm_stride;
  d.bm_bucket_mask = h.bm_bucket_mask;
  d.bm_bucket_mult = h.bm_bucket_mult;
  d.bm_has_key_rows = h.bm_has_key_rows ? 1u : 0u;
  d.bm_has_inline = h.bm_trecx.empty() ? 0u : 1u;
  d.bm_blob_bytes = 0;
  {
    // pack the bitmap tables into one blob (the LDS image)
    const void* src[9] = {h.bm_row_bits.data(), h.bm_nsrows.data(), h.bm_nswords_off.data(), h.bm_nswords.data(),
                          h.bm_buckets.data(), h.bm_trec.data(), h.bm_trecx.data(), h.bm_nswords64_off.data(),
                          h.bm_nswords64.data()};
    const size_t bytes[9] = {h.bm_row_bits.size() * 8, h.bm_nsrows.size() * 8, h.bm_nswords_off.size() * 4,
                             h.bm_nswords.size() * 4, h.bm_buckets.size() * sizeof(AtomBucket),
                             h.bm_trec.size() * sizeof(TermRec), h.bm_trecx.size() * sizeof(TermX),
                             h.bm_nswords64_off.size() * 4, h.bm_nswords64.size() * 4};
    // blob order: what the LDS form stages first, the block lists of the L2 form last
    static const int kOrder[9] = {0, 1, 4, 5, 6, 7, 8, 2, 3};
    size_t o = 0;
    for (int q = 0; q < 9; ++q) {
      const int k = kOrder[q];
      if (q == 7) d.bm_lds_bytes = (uint32_t)o;
      d.bm_off[k] = (uint32_t)o, o += (bytes[k] + 15) & ~(size_t)15;
    }
    if (getenv("KT_DEBUG_LDS"))
      fprintf(stderr, "bitmap blob: rows=%zu nsrows=%zu nsb_off=%zu nsb=%zu buckets=%zu trec=%zu trecx=%zu w64_off=%zu w64=%zu (stride %u words, %u rows)\n",
              bytes[0], bytes[1], bytes[2], bytes[3], bytes[4], bytes[5], bytes[6], bytes[7], bytes[8], h.bm_stride, h.bm_rows);
    std::vector<unsigned char> blob(o + 16, 0);
    for (int k = 0; k < 9; ++k)
      if (bytes[k]) memcpy(blob.data() + d.bm_off[k], src[k], bytes[k]);
    if ((e = up(d.bm_blob, d.cap_bm_blob, blob, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;  // `blob` is a temporary
    d.bm_blob_bytes = (uint32_t)o;
  }
  return hipSuccess;
}

void release_index(IndexDev& d) {
  if (d.slow_thr) (void)hipFree(d.slow_thr);
  if (d.bm_blob) (void)hipFree(d.bm_blob);
  d = IndexDev();
}

}  // namespace kt
