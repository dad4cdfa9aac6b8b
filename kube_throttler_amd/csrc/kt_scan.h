// kt_scan.h — the wave-autonomous selector scan shared by kt_check_bitmap and kt_aggregate_bitmap (gfx950).
//
// The exact term bitmaps of the selector program (kt_index.h) are cut into chunks that fit LDS; the kernels make one
// chunk resident at a time (open_chunk) and scan every tile of the workgroup against it.  One wave owns a tile of 64
// pods, lane = pod:
//
//   advance : every lane that still has words takes the next (word, namespace mask) entry of its namespace's list and
//             accumulates its atom rows over that word: `any |= r`, `two |= any & r`, `three |= two & r`, `veto |= r'`
//             (one 64-bit ds_read per atom from the word's column of the `any` plane, a second one from the veto plane
//             when some row holds a veto bit in the word).
//             A pod carries at most one atom of any requirement, so the accumulators count satisfied positive
//             requirements per term and
//                 x = select(need: any / two / three) & ~veto & nsmask
//             are the terms of the word the pod MATCHES — exactly, no candidates to confirm (except the rare shapes
//             flagged in the word header's `slow` mask, confirmed on the spot by the generic walk);
//   peel    : while any lane holds match bits, each such lane takes its lowest bit and hands the term number to the
//             kernel's consumer (lane = pod throughout: the consumer keeps its per-pod state in registers).
//
// All control flow is wave-uniform (ballots); per-lane work is predicated and every LDS read is issued from an
// always-valid address.  A throttle with several selector terms is reported once: its terms are numbered contiguously
// and a lane meets its matches in ascending number, so the consumers drop a match that repeats the lane's previous
// throttle.
#pragma once
#include "kt_index_device.h"

// Namespace-ordered scans hand the tiles of a workgroup's range to its waves as they get free (a counter in LDS) — with a fixed
// stride every chunk pass ended with the waves that own one tile more (configs[4] check: 12.8 % of the wave cycles at the
// barrier, 6.6 % with the counter; 0.532 -> 0.494 ms).

namespace kt {

// What the kernels need of the index (IndexDev)
struct BmIndexArgs {
  const unsigned char* blob;  // chunk images
  const BmChunk* chunks;
  uint32_t n_chunks, n_rows;
  uint32_t lds_img;  // LDS offset of the resident image part
  // scans in namespace order (rows ordered by launch_order_rows_by_ns)
  const uint32_t* chunk_ns;  // [n_chunks][ns_words] bit n: namespace n has words in the chunk
  uint32_t ns_words;
  uint32_t by_ns;    // the workgroup takes a contiguous range of tiles and skips chunks without words of its namespaces
};

template <class Take>
static inline void plan_bitmap_index(const IndexDev& ix, BmIndexArgs& a, Take&& take) {
  a.blob = ix.bm_blob, a.chunks = ix.bm_chunks;
  a.n_chunks = ix.n_chunks, a.n_rows = ix.bm_rows;
  a.lds_img = take(ix.bm_max_lds);
  a.chunk_ns = ix.bm_chunk_ns, a.ns_words = ix.ns_words, a.by_ns = 0u;
}

// Scans in namespace order: does chunk ci hold words of some namespace in [ns_lo, ns_hi]?  Wave-uniform arguments: scalar
// loads.  The workgroup walks the chunks first_ci .. last_ci that are relevant; the once-per-pod work (slow list, overflow
// pods, the start of the carry) rides on first_ci, the summary words on last_ci — chunk 0 when no chunk holds a word of the
// range's namespaces (round 6: with the grouped plan chunk 0 belongs to ONE group of namespaces; every other workgroup used
// to stage and scan it for nothing).
__device__ __forceinline__ bool chunk_relevant(const BmIndexArgs& a, uint32_t ci, uint32_t ns_lo, uint32_t ns_hi) {
  const uint32_t* m = a.chunk_ns + (size_t)ci * a.ns_words;
  const uint32_t w_lo = ns_lo >> 5, w_hi = ns_hi >> 5;
  for (uint32_t w = w_lo; w <= w_hi; ++w) {
    uint32_t mask = 0xFFFFFFFFu;
    if (w == w_lo) mask &= 0xFFFFFFFFu << (ns_lo & 31u);
    if (w == w_hi) mask &= 0xFFFFFFFFu >> (31u - (ns_hi & 31u));
    if (m[w] & mask) return true;
  }
  return false;
}
__device__ __forceinline__ void relevant_chunk_span(const BmIndexArgs& a, uint32_t ns_lo, uint32_t ns_hi, uint32_t& first, uint32_t& last) {
  first = 0xFFFFFFFFu, last = 0u;
  for (uint32_t ci = 0; ci < a.n_chunks; ++ci)
    if (chunk_relevant(a, ci, ns_lo, ns_hi)) first = first == 0xFFFFFFFFu ? ci : first, last = ci;
  if (first == 0xFFFFFFFFu) first = 0u;
}

// The tables of the chunk that is resident in LDS
struct BmView {
  KT_LDS const unsigned char* rows;  // any[n_words][col_rows], then (VETO) veto[n_words][col_rows]: 8-byte cells
  KT_LDS const WordHdr* hdr;         // [n_words]
  KT_LDS const u32x2* nsl_rng;       // [n_ns] {begin, end} of the namespace's word list
  KT_LDS const NsWord* nsl;
  const uint32_t* term_g;            // (HBM) selector-program term of every number: `slow` candidates only
  uint32_t col_bytes;                // bytes per word column
  uint32_t veto_off;                 // from a cell of the `any` plane to the same cell of the veto plane (words with a veto column)
  uint32_t zero_col;                 // offset (from rows) of the veto plane's all-zero column: what a word WITHOUT a veto column reads
  uint32_t has_slow;
};

// all threads of the workgroup: n16 16-byte pieces from src to dst, four independent loads in flight per thread
__device__ __forceinline__ void lds_stage16(KT_LDS u32x4* dst, const u32x4* src, uint32_t n16) {
  for (uint32_t i = threadIdx.x; i < n16; i += 4 * kBlockIx) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[min(i + k * kBlockIx, n16 - 1u)];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * kBlockIx < n16) dst[i + k * kBlockIx] = v[k];
  }
}

// Several copies into LDS as ONE batch: all threads of the workgroup, up to DEPTH 16-byte pieces per thread in flight over
// ALL segments before the first is stored.  A chunk prologue used to be three or four lds_stage16 calls one after the other —
// image, ranks / TermInfo, WordVerdict — each a full round trip to L2 for a few kilobytes per thread-pass: measured on the
// configs[4] shard at ~9.6 us per chunk a workgroup opens (15 of them per launch), most of it those dependent round trips.
struct StageSeg {
  uint32_t dst;        // byte offset in the workgroup's LDS
  const u32x4* src;
  uint32_t n16;
};
template <int NS, int DEPTH = 8>
__device__ __forceinline__ void lds_stage_segments(KT_LDS unsigned char* lds, const StageSeg (&sg)[NS]) {
  uint32_t total = 0;
#pragma unroll
  for (int k = 0; k < NS; ++k) total += sg[k].n16;
  if (total == 0u) return;
  // piece i of the batch -> (segment, piece of the segment): the segments back to back
  auto locate = [&](uint32_t i, uint32_t& seg_dst, uint64_t& seg_src) -> uint32_t {
    uint32_t off = i;
    seg_dst = sg[0].dst, seg_src = (uint64_t)sg[0].src;
#pragma unroll
    for (int k = 1; k < NS; ++k) {
      uint32_t before = 0;
#pragma unroll
      for (int j = 0; j < k; ++j) before += sg[j].n16;
      const bool later = i >= before && sg[k].n16 != 0u;
      seg_dst = later ? sg[k].dst : seg_dst;
      seg_src = later ? (uint64_t)sg[k].src : seg_src;
      off = later ? i - before : off;
    }
    return off;
  };
  for (uint32_t base = threadIdx.x; base < total; base += DEPTH * kBlockIx) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const uint32_t i = base + (uint32_t)j * kBlockIx;
      if (i < total) {  // (no piece is requested twice: a 44 KB image is 2.75 pieces per thread, not eight)
        uint32_t d;
        uint64_t sp;
        const uint32_t off = locate(i, d, sp);
        v[j] = ((const u32x4*)sp)[off];
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const uint32_t i = base + (uint32_t)j * kBlockIx;
      if (i < total) {
        uint32_t d;
        uint64_t sp;
        const uint32_t off = locate(i, d, sp);
        *(KT_LDS u32x4*)(lds + d + off * 16u) = v[j];
      }
    }
  }
}

// The tables of chunk `ch` as they will sit in LDS, and the copy that makes the image resident (the caller stages it — alone
// or batched with its own tables: lds_stage_segments — after a barrier: nobody still reads the previous image — and barriers
// again before the first read)
__device__ __forceinline__ StageSeg chunk_image_segment(const BmIndexArgs& a, const BmChunk& ch) {
  return StageSeg{a.lds_img, (const u32x4*)(a.blob + ch.img_off), ch.lds_bytes / 16u};
}
template <bool VETO>
__device__ __forceinline__ BmView open_chunk(KT_LDS unsigned char* lds, const BmIndexArgs& a, const BmChunk& ch) {
  KT_LDS unsigned char* base = lds + a.lds_img;
  BmView v;
  v.rows = base;
  v.hdr = (KT_LDS const WordHdr*)(base + ch.off_hdr);
  v.nsl_rng = (KT_LDS const u32x2*)(base + ch.off_nsl_rng);
  v.nsl = (KT_LDS const NsWord*)(base + ch.off_nsl);
  v.term_g = (const uint32_t*)(a.blob + ch.img_off + ch.off_term_g);
  v.col_bytes = ch.col_rows * 8u;
  v.veto_off = VETO ? ch.n_words * ch.col_rows * 8u : 0u;
  v.zero_col = VETO ? ch.zero_col : 0u;
  v.has_slow = ch.has_slow;
  return v;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}


// Three-input boolean functions on both halves of a 64-bit word: gfx950's v_bitop3_b32 takes any truth table; only the
// symmetric ones are used here (xor3 0x96, majority 0xE8, or3 0xFE), so the operand order does not matter.
template <int TT>
__device__ __forceinline__ uint64_t bitop3_64(uint64_t a, uint64_t b, uint64_t c) {
  const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
  const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT);
  return (uint64_t)lo | (uint64_t)hi << 32;
}
// (measured, profiles/r04_valu_rates.txt: v_bitop3_b32 issues in 2.7 cycles per wave and SIMD like the two-operand
//  v_or_b32 / v_xor_b32, where the three-operand forms the compiler picks by itself — v_or3_b32, v_bfi_b32, v_and_or_b32 —
//  take 4.4: every three-input boolean step of the scans is spelled as a bitop3 with its truth table.  Table index =
//  a << 2 | b << 1 | c, operand 0 the most significant bit: profiles/r04_bitop3_probe.txt.)
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0x96>(a, b, c); }
__device__ __forceinline__ uint64_t andn_64(uint64_t a, uint64_t b) { return bitop3_64<0x30>(a, b, b); }                   // a & ~b
__device__ __forceinline__ uint64_t and_not_and_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0x20>(a, b, c); }  // a & ~b & c
__device__ __forceinline__ uint64_t and_not_not_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0x10>(a, b, c); }  // a & ~b & ~c
__device__ __forceinline__ uint64_t and_nand_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0x70>(a, b, c); }     // a & ~(b & c)
__device__ __forceinline__ uint64_t mux_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0xD8>(a, b, c); }          // c ? b : a
__device__ __forceinline__ uint64_t maj3_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0xE8>(a, b, c); }
__device__ __forceinline__ uint64_t or3_64(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0xFE>(a, b, c); }

// How many of eight bitmap rows hold each bit, as a 2-bit number (ones, twos) — EXACT as long as no bit is met more than
// three times, which the index guarantees for the rows of one pod (a pod carries one atom per key, an exactly indexed
// term has at most three positive keys, a `slow` term keeps one): two full adders and a half adder over the rows, a
// full adder over their sums; at most one of the four carries can be set.  13 three-input operations per half where the
// running 2-bit counter (c1 ^= c0 & r; c0 ^= r) took 24.
__device__ __forceinline__ void count8(const uint64_t (&r)[8], uint64_t& ones, uint64_t& twos) {
  const uint64_t s1 = xor3_64(r[0], r[1], r[2]), c1 = maj3_64(r[0], r[1], r[2]);
  const uint64_t s2 = xor3_64(r[3], r[4], r[5]), c2 = maj3_64(r[3], r[4], r[5]);
  const uint64_t s3 = r[6] ^ r[7], c3 = r[6] & r[7];
  ones = xor3_64(s1, s2, s3);
  twos = or3_64(c1, c2, c3) | maj3_64(s1, s2, s3);
}

// ... and as a 3-bit number (ones, twos, fours) for programs whose terms count up to five positive keys (NEED = 5): exact as
// long as no bit is met more than seven times — a term keeps at most five positives.  13 three-input operations per half.
__device__ __forceinline__ void count8_wide(const uint64_t (&r)[8], uint64_t& ones, uint64_t& twos, uint64_t& fours) {
  const uint64_t s1 = xor3_64(r[0], r[1], r[2]), k1 = maj3_64(r[0], r[1], r[2]);
  const uint64_t s2 = xor3_64(r[3], r[4], r[5]), k2 = maj3_64(r[3], r[4], r[5]);
  const uint64_t s3 = r[6] ^ r[7], k3 = r[6] & r[7];
  ones = xor3_64(s1, s2, s3);
  const uint64_t k4 = maj3_64(s1, s2, s3);
  const uint64_t t = xor3_64(k1, k2, k3), f1 = maj3_64(k1, k2, k3);  // the four carries of weight two
  twos = t ^ k4;
  fours = f1 | (t & k4);
}

// A pod's atom row (PodTable::latom, LA u16 ids) as byte offsets of its cells inside a word column: LA/8 128-bit loads,
// issued from an always-valid address.
template <int LA>
__device__ __forceinline__ void load_atoms(const uint16_t* latom, int64_t p, u32x4 (&raw)[LA / 8]) {
  const u32x4* a = (const u32x4*)(latom + p * LA);
#pragma unroll
  for (int q = 0; q < LA / 8; ++q) raw[q] = a[q];
}
template <int LA>
__device__ __forceinline__ void atom_row_offsets(const u32x4 (&raw)[LA / 8], uint32_t (&ro)[LA]) {
#pragma unroll
  for (int q = 0; q < LA / 8; ++q) {
    const uint32_t w[4] = {raw[q].x, raw[q].y, raw[q].z, raw[q].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ro[8 * q + 2 * k] = (w[k] & 0xFFFFu) << 3;
      ro[8 * q + 2 * k + 1] = (w[k] >> 16) << 3;
    }
  }
}

// One 64-pod tile against the resident chunk.  `ns` must be a valid namespace row for EVERY lane (callers pass 0 for
// lanes without a pod).
//   lane_on    : the lane's pod takes part in selector matching
//   ro[]       : byte offsets of the pod's atom rows (atom_row_offsets; 0 = the all-zero row)
//   match(has, c) : wave-wide call per peel step; lanes with `has` matched term number c of this chunk (c = 0 for the
//                others).  Ascending c per lane.
//   confirm(c) : lane-divergent — does the lane's pod satisfy ALL requirements of `slow` term number c
//   post(w, x) : per lane, once per advanced word: x = the matched terms of word w; returns the bits that still go
//                through the peel (a consumer that can settle matches 64 at a time with mask algebra keeps the rest)
struct ScanKeepAll {
  __device__ __forceinline__ uint64_t operator()(uint32_t, uint64_t x, int) const { return x; }
};
struct ScanNoPrefetch {
  __device__ __forceinline__ int operator()(uint32_t) const { return 0; }
};
//   pre(w)      : per lane, called as soon as the word of an advance is known — BEFORE the atom rows are consumed: what the
//                 consumer's post() will need of word w (its reads then travel with the row reads instead of starting a
//                 round trip of their own after the accumulation); its result is handed to post(w, x, pre(w))
//   PIPE        : the list entry of the NEXT advance is requested during this one (one LDS round trip less per word: the
//                 configs[4] check went from 926 to 824 us with it and pre()); on in the rich instantiations with room
//                 for its four registers (not the 64-VGPR ones, and no gain measured on single-chunk simple programs)
template <int LA, bool VETO, int NEED, bool PIPE = true, class Match, class Confirm, class Post = ScanKeepAll, class Pre = ScanNoPrefetch>
__device__ __forceinline__ void scan_tile(const BmView& b, bool lane_on, uint32_t ns, const uint32_t (&ro)[LA], Match&& match,
                                          Confirm&& confirm, Post&& post = Post(), Pre&& pre = Pre()) {
  const u32x2 rng = b.nsl_rng[ns];
  uint32_t k = rng.x;
  const uint32_t k1 = lane_on ? rng.y : k;
  uint64_t x = 0;
  uint32_t w = 0;
  u32x4 e_next = {0u, 0u, 0u, 0u};
  if (PIPE) e_next = *(lds_u4p)(b.nsl + (k < k1 ? k : 0u));
  for (;;) {
    const bool has = x != 0;
    if (__ballot(has) != 0ull) {
      // ---- peel: one matched term per lane that has any
      const uint32_t c = has ? w * 64u + (uint32_t)__ffsll((unsigned long long)x) - 1u : 0u;
      x &= x - 1ull;
      match(has, c);
    } else if (__ballot(k < k1) != 0ull) {
      // ---- advance: next word of every lane that still has one
      const bool adv = k < k1;
      u32x4 e;  // {w, -, mask lo, mask hi}
      if (PIPE) {
        e = e_next;
        k += adv ? 1u : 0u;
        e_next = *(lds_u4p)(b.nsl + (k < k1 ? k : 0u));  // the entry of the next advance, in flight behind this word's reads
      } else {
        e = *(lds_u4p)(b.nsl + (adv ? k : 0u));
        k += adv ? 1u : 0u;
      }
      w = e.x;
      const u64x2 h0 = *(KT_LDS const u64x2*)(b.hdr + w);  // {univ, m2}
      const auto pf = pre(w);
      KT_LDS const unsigned char* col = b.rows + __umul24(w, b.col_bytes);  // (word < 2^10, column bytes < 2^18)
      // (the veto plane holds the columns of the words that have a veto bit somewhere and one all-zero column: a lane whose word
      //  has none — a tile that straddles namespaces, or an instantiation that reads both planes for every word — reads that)
      KT_LDS const unsigned char* colv = (!VETO || b.zero_col == 0u || (e.y & kNsWordVeto) != 0u) ? col + b.veto_off : b.rows + b.zero_col;
      uint64_t xx, vet = 0;
      if (NEED >= 3) {
        // The FORM of the word (NsWord::flags — inside a class the groups are numbered by form, so most words are pure),
        // wave-uniform: the lanes of a tile in namespace order visit the same words; a tile whose lanes visit words of
        // different forms takes the general path, which is right for every word.
        //   no veto bit in any row  : only the `any` plane is read (8 bytes per atom instead of 16: these
        //                             gathers are what the scans of large programs wait for)
        // (only in the instantiations with room for it — the PIPE ones with eight atom slots: the 64-VGPR forms and the
        //  16 / 32-slot ones spill over the second path.  A second form — the OR / XOR accumulation of the simple instantiation
        //  for words without a need-3 term, NsWord::flags carries the bit — saved VALU only and spilled in the 128-VGPR forms.)
        constexpr bool FORMS = PIPE && LA == 8;
        const bool w_veto = VETO && (!FORMS || __ballot(adv && (e.y & kNsWordVeto) != 0u) != 0ull);
        // hits per term as a 2-bit number (c1 c0): a pod carries at most one atom of any requirement and an exactly
        // indexed term has at most three positive keys, so the count never passes 3.  Eight rows at a time through a
        // small adder tree of three-input operations (count8); the groups of eight are added as 2-bit numbers.
        static_assert(LA % 8 == 0, "atom slots come in eights");
        uint64_t c0 = 0, c1 = 0, c2 = 0;  // the 2-bit counter (NEED = 5: 3 bits)
#pragma unroll
        for (int g8 = 0; g8 < LA / 8; ++g8) {
          // (one set of registers for both forms: the veto halves of a veto-free word are zeroed instead of read)
          uint64_t r[8], v8[8];
          if (w_veto) {
            // Eight atom slots: the veto plane's reads go first and are folded into `vet` before the `any` plane's are
            // issued — two dependent LDS round trips instead of one, but 16 registers are live at a time instead of 32: the
            // 128-VGPR forms lose their scratch (configs[4] lean check 16 -> 0 B, 0.422 -> 0.410 ms; issued together in
            // either order: 0.421).  With 16 / 32 slots the groups of eight overlap anyway and the split costs registers.
            if (LA == 8) {
#pragma unroll
              for (int l = 0; l < 8; ++l) v8[l] = *(KT_LDS const unsigned long long*)(colv + ro[8 * g8 + l]);
              vet |= or3_64(or3_64(v8[0], v8[1], v8[2]), or3_64(v8[3], v8[4], v8[5]), v8[6] | v8[7]);
              asm volatile("" : "+v"(vet));  // (the fold stays ahead of the next batch of reads)
#pragma unroll
              for (int l = 0; l < 8; ++l) v8[l] = 0ull, r[l] = *(KT_LDS const unsigned long long*)(col + ro[8 * g8 + l]);
            } else {
#pragma unroll
              for (int l = 0; l < 8; ++l) {
                r[l] = *(KT_LDS const unsigned long long*)(col + ro[8 * g8 + l]);
                v8[l] = *(KT_LDS const unsigned long long*)(colv + ro[8 * g8 + l]);
              }
            }
          } else {
#pragma unroll
            for (int l = 0; l < 8; ++l) r[l] = *(KT_LDS const unsigned long long*)(col + ro[8 * g8 + l]), v8[l] = 0ull;  // (the `any` plane only)
          }
          if (VETO) vet |= or3_64(or3_64(v8[0], v8[1], v8[2]), or3_64(v8[3], v8[4], v8[5]), v8[6] | v8[7]);
          if (NEED >= 4) {
            uint64_t ones, twos, fours;
            count8_wide(r, ones, twos, fours);
            if (g8 == 0) {
              c0 = ones, c1 = twos, c2 = fours;
            } else {  // (3-bit add; the total does not pass 7)
              const uint64_t k0 = c0 & ones;
              c0 ^= ones;
              c2 |= fours | maj3_64(c1, twos, k0);
              c1 = xor3_64(c1, twos, k0);
            }
          } else {
            uint64_t ones, twos;
            count8(r, ones, twos);
            if (g8 == 0) {
              c0 = ones, c1 = twos;
            } else {  // (the total still does not pass 3: at most one carry)
              c1 |= twos | (c0 & ones);
              c0 ^= ones;
            }
          }
        }
        const u64x2 h1 = *(KT_LDS const u64x2*)((KT_LDS const unsigned char*)(b.hdr + w) + 16);  // {m3, slow}
        if (NEED >= 4) {
          const u64x2 h2 = *(KT_LDS const u64x2*)((KT_LDS const unsigned char*)(b.hdr + w) + 32);  // {m4, m5}
          const uint64_t any = or3_64(h0.x, c0, c1) | c2, two = c1 | c2, three = c2 | (c0 & c1), five = c2 & (c0 | c1);  // >= 1, 2, 3, 5; >= 4 is c2
          xx = mux_64(any, two, h0.y);
          xx = mux_64(xx, three, h1.x);
          xx = mux_64(xx, c2, h2.x);
          xx = mux_64(xx, five, h2.y);
        } else {
          const uint64_t any = or3_64(h0.x, c0, c1), two = c1, three = c0 & c1;  // >= 1 (or no positive requirement), >= 2, == 3
          xx = mux_64(any, two, h0.y);    // (any & ~m2) | (two & m2)
          xx = mux_64(xx, three, h1.x);   // (xx & ~m3) | (three & m3)
        }
        xx = and_not_and_64(xx, vet, (uint64_t)e.z | (uint64_t)e.w << 32);
        xx = adv ? xx : 0ull;
        if (b.has_slow) {
          uint64_t sl = xx & h1.y;
          while (sl) {  // rare shapes: every requirement through the generic walk (lane-divergent)
            const uint32_t bit = (uint32_t)__ffsll((unsigned long long)sl) - 1u;
            sl &= sl - 1ull;
            if (!confirm(w * 64u + bit)) xx &= ~(1ull << bit);
          }
        }
      } else {
        // any = the terms met in some row; an exactly indexed term of this form has at most two positive keys, each met by
        // at most one of the pod's atoms, so "both met" is "met, and in an even number of rows": the OR and the XOR of the
        // rows — order-free, three rows per instruction — replace the running any / two pair (two |= any & r; any |= r)
        uint64_t any = h0.x, par = 0;
#pragma unroll
        for (int g8 = 0; g8 < LA / 8; ++g8) {
          uint64_t r[8];
#pragma unroll
          for (int l = 0; l < 8; ++l) {
            r[l] = *(KT_LDS const unsigned long long*)(col + ro[8 * g8 + l]);
            if (VETO) vet |= *(KT_LDS const unsigned long long*)(colv + ro[8 * g8 + l]);
          }
          any = or3_64(or3_64(r[0], r[1], r[2]), or3_64(r[3], r[4], r[5]), or3_64(r[6], r[7], any));
          if (NEED >= 2) par = xor3_64(xor3_64(r[0], r[1], r[2]), xor3_64(r[3], r[4], r[5]), xor3_64(r[6], r[7], par));
        }
        xx = any;
        if (NEED >= 2) xx = and_nand_64(any, par, h0.y);  // under m2 (two positive keys): met, and in an even number of rows
        uint64_t slow = 0;
        if (VETO) {  // the rich instantiation also serves programs with `slow` shapes
          const u64x2 h1 = *(KT_LDS const u64x2*)((KT_LDS const unsigned char*)(b.hdr + w) + 16);  // {m3, slow}
          slow = h1.y;
        }
        if (VETO) xx = and_not_and_64(xx, vet, (uint64_t)e.z | (uint64_t)e.w << 32);
        else xx &= (uint64_t)e.z | (uint64_t)e.w << 32;
        xx = adv ? xx : 0ull;
        if (VETO && b.has_slow) {
          uint64_t sl = xx & slow;
          while (sl) {
            const uint32_t bit = (uint32_t)__ffsll((unsigned long long)sl) - 1u;
            sl &= sl - 1ull;
            if (!confirm(w * 64u + bit)) xx &= ~(1ull << bit);
          }
        }
      }
      x = post(w, xx, pf);
    } else {
      break;
    }
  }
}

}  // namespace kt
