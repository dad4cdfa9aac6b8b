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
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes) {
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
    // term numbers: a class (run of throttles with the same first admission set) never straddles a 64-bit word of
    // the bitmaps unless it is larger than one (128 for big programs: fewer, fuller words per namespace)
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
    const uint32_t W = G2 ? (G2 + 63) / 64 : 1u;  // 64-bit words per full row (an empty program keeps one zero word)
    out.bm_words = W;
    // ---- atoms -> bitmap rows
    std::unordered_map<uint32_t, uint32_t> row_of;
    for (auto& b : bts)
      for (uint32_t a : b.atoms) row_of.emplace(a, 0u);
    std::vector<uint32_t> atoms;
    for (auto& kv : row_of) atoms.push_back(kv.first);
    std::sort(atoms.begin(), atoms.end());
    for (uint32_t i = 0; i < atoms.size(); ++i) row_of[atoms[i]] = i + 2;
    const uint32_t R = (uint32_t)atoms.size() + 2;
    out.bm_rows = R;
    // ---- full bitmaps (host only), dense throttle ranks in term order
    std::vector<uint64_t> rows((size_t)R * W, 0ull), nsrows((size_t)n_ns * W, 0ull);
    std::vector<TermRec> trec(W * 64, TermRec{0, 0, 0, 0});
    std::vector<TermX> trecx(W * 64, no_extras);
    std::vector<uint32_t> term_rank(W * 64, 0u);
    std::vector<uint8_t> real(W * 64, 0);  // term number in use (not padding)
    bool any_inline = false;
    out.bm_rank_t.clear();
    {
      std::vector<uint32_t> by_num(G2, ~0u);
      for (size_t q = 0; q < bts.size(); ++q) by_num[num[q]] = (uint32_t)q;
      uint32_t last_t = ~0u;
      for (uint32_t c = 0; c < G2; ++c) {
        if (by_num[c] == ~0u) continue;  // padding
        const BT& b = bts[by_num[c]];
        if (b.t != last_t) out.bm_rank_t.push_back(b.t), last_t = b.t;
        term_rank[c] = (uint32_t)out.bm_rank_t.size() - 1;
        real[c] = 1;
        const uint64_t bit = 1ull << (c & 63);
        trec[c] = TermRec{b.g, b.t, b.pair2, b.flags};
        trecx[c] = b.x;
        any_inline |= (b.flags & kPostInline) != 0;
        if (b.atoms.empty()) rows[c >> 6] |= bit;
        for (uint32_t a : b.atoms) rows[(size_t)row_of[a] * W + (c >> 6)] |= bit;
        for (uint32_t n = 0; n < n_ns; ++n)
          if ((b.adm[n >> 5] >> (n & 31)) & 1u) nsrows[(size_t)n * W + (c >> 6)] |= bit;
      }
    }
    out.bm_has_inline = any_inline;
    // ---- atoms -> rows in 4-entry buckets: a few multipliers per size, then double, until no bucket overflows
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
    // ---- chunks: word ranges whose image (rows | nsrows | word lists | TermRec | TermX) plus the aggregate table
    //      of their throttles fits the LDS budget; a throttle's terms never straddle a chunk
    std::vector<uint8_t> splittable(W + 1, 1);  // chunk may START at word w
    for (uint32_t w = 1; w < W; ++w) {
      // the last real term of word w-1 and the first real term of word w belong to different throttles?
      int64_t a = -1, b2 = -1;
      for (int k = 63; k >= 0 && a < 0; --k)
        if (real[(size_t)(w - 1) * 64 + k]) a = (int64_t)(w - 1) * 64 + k;
      for (int k = 0; k < 64 && b2 < 0; ++k)
        if (real[(size_t)w * 64 + k]) b2 = (int64_t)w * 64 + k;
      if (a >= 0 && b2 >= 0 && term_rank[a] == term_rank[b2]) splittable[w] = 0;
    }
    // LDS left for a chunk once the atom buckets are resident: image + table (aggregate), image alone (check)
    const size_t bucket_bytes = out.bm_buckets.size() * sizeof(AtomBucket);
    const size_t agg_room = agg_budget > bucket_bytes ? agg_budget - bucket_bytes : 0;
    const size_t chk_room = chk_budget > bucket_bytes ? chk_budget - bucket_bytes : 0;
    uint64_t slab_run = 0;
    const size_t per_word = (size_t)R * 8 + (size_t)n_ns * 8 + 64 * sizeof(TermRec) + (any_inline ? 64 * sizeof(TermX) : 0);
    out.bm_chunks.clear();
    out.bm_images.clear();
    out.bm_max_img = 0, out.bm_max_thr = 0, out.bm_slab_bytes = 0;
    uint32_t w0 = 0;
    while (w0 < W) {
      // grow the chunk word by word while it fits; cut at the last boundary that splits no throttle
      uint32_t w1 = 0;
      for (uint32_t cand = w0 + 1; cand <= W; ++cand) {
        uint32_t r_lo = ~0u, r_hi = 0;
        for (size_t c = (size_t)w0 * 64; c < (size_t)cand * 64; ++c)
          if (real[c]) r_lo = std::min(r_lo, term_rank[c]), r_hi = std::max(r_hi, term_rank[c]);
        const uint32_t nthr = r_lo == ~0u ? 0 : r_hi - r_lo + 1;
        const size_t nw = cand - w0;
        const size_t img = ((nw | 1) * ((size_t)R + n_ns)) * 8 + nw * (per_word - (size_t)R * 8 - (size_t)n_ns * 8) +
                           ((size_t)n_ns + 1) * 4 + (size_t)n_ns * nw * 4 + 256;
        // the kernels lay LDS out ONCE for all chunks — the largest image next to the table of the largest throttle
        // count — so a chunk has to fit together with the maxima of the chunks cut before it, not only on its own
        const size_t img_hi = std::max(img, (size_t)out.bm_max_img);
        const size_t thr_hi = std::max((size_t)nthr, (size_t)out.bm_max_thr);
        const bool fits = img_hi <= chk_room && img_hi + thr_hi * thr_bytes + 16 <= agg_room;
        if (!fits && w1 != 0) break;
        if (cand == W || splittable[cand]) {
          w1 = cand;
          if (!fits) break;  // a single stretch larger than the budget: the launchers notice
        }
      }
      // ---- image of words [w0, w1)
      BmChunk ch{};
      ch.w0 = w0, ch.n_words = w1 - w0;
      ch.stride = ch.n_words | 1u;
      uint32_t r_lo = ~0u, r_hi = 0;
      for (size_t c = (size_t)w0 * 64; c < (size_t)w1 * 64; ++c)
        if (real[c]) r_lo = std::min(r_lo, term_rank[c]), r_hi = std::max(r_hi, term_rank[c]);
      ch.rank0 = r_lo == ~0u ? 0 : r_lo;
      ch.n_thr = r_lo == ~0u ? 0 : r_hi - r_lo + 1;
      std::vector<uint64_t> irows((size_t)R * ch.stride, 0ull), insrows((size_t)n_ns * ch.stride, 0ull);
      for (uint32_t r = 0; r < R; ++r)
        for (uint32_t w = 0; w < ch.n_words; ++w) irows[(size_t)r * ch.stride + w] = rows[(size_t)r * W + w0 + w];
      std::vector<uint32_t> nsw_off((size_t)n_ns + 1, 0u), nsw;
      for (uint32_t n = 0; n < n_ns; ++n) {
        for (uint32_t w = 0; w < ch.n_words; ++w) {
          insrows[(size_t)n * ch.stride + w] = nsrows[(size_t)n * W + w0 + w];
          if (insrows[(size_t)n * ch.stride + w]) nsw.push_back(w);
        }
        nsw_off[n + 1] = (uint32_t)nsw.size();
      }
      std::vector<TermRec> itrec((size_t)ch.n_words * 64);
      std::vector<TermX> itrecx(any_inline ? (size_t)ch.n_words * 64 : 0);
      for (uint32_t k = 0; k < ch.n_words * 64; ++k) {
        TermRec tr = trec[(size_t)w0 * 64 + k];
        if (real[(size_t)w0 * 64 + k]) tr.flags |= (term_rank[(size_t)w0 * 64 + k] - ch.rank0) << 8;  // chunk-local throttle rank
        itrec[k] = tr;
        if (any_inline) itrecx[k] = trecx[(size_t)w0 * 64 + k];
      }
      const void* src[6] = {irows.data(), insrows.data(), nsw_off.data(), nsw.data(), itrec.data(), itrecx.data()};
      const size_t bytes[6] = {irows.size() * 8, insrows.size() * 8, nsw_off.size() * 4, nsw.size() * 4,
                               itrec.size() * sizeof(TermRec), itrecx.size() * sizeof(TermX)};
      uint32_t* offs[6] = {nullptr, &ch.off_nsrows, &ch.off_nsw_off, &ch.off_nsw, &ch.off_trec, &ch.off_trecx};
      size_t o = 0;
      const size_t img0 = out.bm_images.size();
      for (int k = 0; k < 6; ++k) {
        if (offs[k]) *offs[k] = (uint32_t)o;
        o += (bytes[k] + 15) & ~(size_t)15;
      }
      out.bm_images.resize(img0 + o, 0);
      size_t oo = 0;
      for (int k = 0; k < 6; ++k) {
        if (bytes[k]) memcpy(out.bm_images.data() + img0 + oo, src[k], bytes[k]);
        oo += (bytes[k] + 15) & ~(size_t)15;
      }
      ch.img_off = (uint32_t)img0, ch.img_bytes = (uint32_t)o;
      ch.slab_off = (uint32_t)(slab_run / 16);  // one table per (chunk, workgroup), 256 workgroups at most
      slab_run += 256ull * (((uint64_t)ch.n_thr * thr_bytes + 15) & ~15ull);
      out.bm_max_img = std::max(out.bm_max_img, ch.img_bytes);
      out.bm_max_thr = std::max(out.bm_max_thr, ch.n_thr);
      out.bm_chunks.push_back(ch);
      out.bm_slab_bytes = slab_run;
      w0 = w1;
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
  if ((e = up(d.bm_blob, d.cap_bm_blob, h.bm_images, s)) != hipSuccess) return e;
  if ((e = up(d.bm_chunks, d.cap_bm_chunks, h.bm_chunks, s)) != hipSuccess) return e;
  if ((e = up(d.bm_rank_t, d.cap_bm_rank_t, h.bm_rank_t, s)) != hipSuccess) return e;
  if ((e = up(d.bm_buckets, d.cap_bm_buckets, h.bm_buckets, s)) != hipSuccess) return e;
  d.h_chunks = h.bm_chunks;
  d.n_chunks = (uint32_t)h.bm_chunks.size();
  d.bm_max_img = h.bm_max_img, d.bm_max_thr = h.bm_max_thr;
  d.bm_slab_bytes = h.bm_slab_bytes;
  d.bm_bucket_bytes = (uint32_t)(h.bm_buckets.size() * sizeof(AtomBucket));
  d.bm_bucket_mask = h.bm_bucket_mask;
  d.bm_bucket_mult = h.bm_bucket_mult;
  d.bm_has_key_rows = h.bm_has_key_rows ? 1u : 0u;
  d.bm_has_inline = h.bm_has_inline ? 1u : 0u;
  if (getenv("KT_DEBUG_LDS"))
    fprintf(stderr, "bitmap index: %u words, %u rows, %zu chunks, largest image %u B, largest chunk %u throttles, buckets %u B\n",
            h.bm_words, h.bm_rows, h.bm_chunks.size(), h.bm_max_img, h.bm_max_thr, d.bm_bucket_bytes);
  return hipSuccess;
}

void release_index(IndexDev& d) {
  if (d.slow_thr) (void)hipFree(d.slow_thr);
  if (d.bm_blob) (void)hipFree(d.bm_blob);
  if (d.bm_chunks) (void)hipFree(d.bm_chunks);
  if (d.bm_rank_t) (void)hipFree(d.bm_rank_t);
  if (d.bm_buckets) (void)hipFree(d.bm_buckets);
  d = IndexDev();
}

}  // namespace kt
