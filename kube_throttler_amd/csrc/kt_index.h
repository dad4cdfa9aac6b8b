// kt_index.h — the selector side of all throttles compiled into EXACT term bitmaps over label atoms, so that the
// pod x throttle scans do work proportional to (pods + matches) instead of P x T, with no per-candidate decisions.
//
// Atoms.  Every (key,value) pair id that occurs in an In / NotIn requirement and every key that occurs in an
// Exists / DoesNotExist requirement (atom = kKeyAtom | key id) of an indexed term is a *referenced atom* and gets a
// dense 16-bit id (1..A; 0 = "nothing").  Pod labels are translated to these ids once per program change / ingest
// (kt_translate_pods: PodTable::latom), ONE atom per label: the pair when it is referenced, else the key atom when
// that is — labels no selector mentions cannot influence any decision and are dropped there, which is also what lifts
// the label-count cap of the scan kernels.  A key-level requirement (Exists / DoesNotExist) is entered in the key
// atom's row AND in the rows of all referenced pairs of that key.
//
// Terms and groups.  For one namespace the terms of a live throttle that can match are those whose namespace side admits
// it; the namespaces are partitioned by WHICH of the throttle's terms admit them ("cells").  Every cell becomes a GROUP
// of term copies — the admitted terms, in term order, each admitting exactly the cell's namespaces — and every copy gets
// a number c.  Groups are ordered by their admission set (a "class"), a class of <= 64 numbers never straddles a 64-bit
// word, and neither does a group: for any pod exactly one group of a throttle is live and its copies sit side by side
// in one word (the scans report a throttle once by keeping the lowest match of such a run), and the words are class-pure
// even when the terms of a ClusterThrottle select different namespaces — a pod only visits words of classes that admit
// its namespace.  A throttle with more than 64 terms is ONE group whose run spans words (its copies keep their own admission
// sets): the scans that dedupe match by match take such a program (HostIndex::has_long); beyond kMaxIndexedTerms: the slow list.
// Two bitmap families over c, one row per atom:
//     any [a] : terms with a POSITIVE requirement (In / Exists) that atom a satisfies
//     veto[a] : terms with a NEGATIVE requirement (NotIn / DoesNotExist) that atom a violates
// The positive requirements of a term are merged per KEY (In S1 and In S2 = In S1∩S2, In S and Exists = In S), and a pod
// carries one atom per key: the number of rows of `any` in which a term's bit is met while OR-ing the pod's atom rows
// IS the number of its positive keys the pod satisfies.  With need(c) = number of positive keys (0..3):
//     match(pod)[w] = hits>=need (a 2- or 3-bit count per term, masks m2 .. m5) & ~OR veto & nsmask[ns][w]
// is EXACT — no TermRec, no second-pair compare, no inline extras — for terms with up to FIVE positive keys (three until round
// 5).  Terms with more keep their five most selective positives (need = 5) and are flagged in the word header's `slow` mask: a
// pod that meets those five is a candidate, decided by the generic requirement walk.  Throttles that contain an unconvertible podSelector term (error semantics of
// throttle_selector.go:30-42 depend on term order) or more than kMaxIndexedTerms terms go to a "slow list" and are walked
// term by term in order.
// The namespace side of every term (implicit namespace equality of a Throttle, throttle_controller.go:249;
// namespaceSelector of a ClusterThrottle) is pre-evaluated into SelProgram::ns_term_ok and enters as per-namespace
// (word, mask) lists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "kt_device.h"

namespace kt {

constexpr uint32_t kKeyAtom = 0x80000000u;
constexpr uint32_t kMaxIndexedTerms = 512;  // selector terms of ONE throttle the index takes (8 words in one chunk); beyond: the slow list
constexpr uint32_t kCheckWordLds = 64u * 8u + 560u;  // = check_word_lds(16): TermInfo[64] + WordVerdict<16> per word (the worst case)

// term_t[] entry of a chunk image: throttle row | flags
constexpr uint32_t kTermAdj = 0x80000000u;   // multi-term throttle: a match repeating the lane's previous throttle is dropped
constexpr uint32_t kTermReal = 0x40000000u;  // term number in use (not class padding)
constexpr uint32_t kTermRowMask = 0x000FFFFFu;
constexpr uint16_t kRankAdj = 0x8000u;       // term_rank[] entry: chunk-local throttle rank | kRankAdj

struct ThrInfo {
  bool live;
  bool cluster;
  uint32_t ns;
};

// per-word header of a chunk image
struct alignas(16) WordHdr {  // (64 bytes: the address of word w's header is a shift, not a 64-bit multiply-add; an image places
                              //  the headers at a 16-byte boundary — 16-byte reads —, so the type claims no more than that)
  uint64_t univ;  // terms without a positive requirement (hit by every pod)
  uint64_t m2;    // terms that need >= 2 positive hits
  uint64_t m3;    // terms that need >= 3
  uint64_t slow;  // candidates that the generic requirement walk has to confirm
  uint64_t m4;    // terms that need >= 4 (round 6: programs with four or five positive keys per term take the NEED = 5
  uint64_t m5;    // instantiations, which count hits as 3-bit numbers) / that need 5
  uint64_t pad[2];
};
static_assert(sizeof(WordHdr) == 64, "the scans address a word's header by a shift");
// one entry of a namespace's word list
struct alignas(16) NsWord {
  uint32_t w;
  uint32_t flags;  // kNsWord*: the FORM of word w (the same in every namespace's entry of it)
  uint64_t mask;   // terms of word w whose namespace side admits the namespace
};
// Inside a class (groups with one admission set) the groups are numbered by form — without / with a term of three
// positive keys, without / with a negative requirement — so that most words of a rich program are pure: a word without
// veto bits is scanned by reading the `any` plane only (8 bytes instead of 16 per atom: the scans of large
// programs are bound by these LDS gathers), a word without need-3 terms takes the OR / XOR accumulation of the simple form
// instead of the counting tree.  configs[4] shard: 54 % of the visited words carry no veto bit, 54 % no need-3 term.
constexpr uint32_t kNsWordVeto = 1u;   // some atom row holds a veto bit in this word
constexpr uint32_t kNsWordNeed3 = 2u;  // some term of this word needs three positive hits (WordHdr::m3 != 0)

// One LDS-sized slice of the bitmap form: n_words 64-bit words of every row — a run of consecutive words of the numbered
// program (the global plan) or the words a group of namespaces visits (the grouped plan: kt_index.cpp, cut_chunks); w0 counts
// the words of the chunks before this one (the position of the chunk's words in the per-word verdict images).  The terms of
// a throttle never straddle two chunks; `rank0 .. rank0 + n_thr` are the dense ranks of the chunk's groups (chunk-local
// number order) in HostIndex::bm_rank_t — a group that sits in several chunks (grouped plan) has a rank in each.
// Image layout (all offsets relative to img_off, every section 16-byte aligned):
//   [0, lds_bytes)        what the scan keeps in LDS:  rows | WordHdr[n_words] | nsl_rng u32[n_ns][2] | NsWord[]
//     rows: u64 any[n_words][col_rows]            one COLUMN per word: the cell of (row id, word w) at (w * col_rows + id) * 8
//           u64 veto[veto_cols][col_rows] behind it  (programs with negative requirements: the rich form)
//       col_rows = the rows rounded up to 32 (image_col_rows): a column is a whole number of LDS bank rounds, so the bank
//       slot of a cell is id mod 32 whatever the word — what a gather's lanes collide on is decided by the atom numbering
//       alone (kt_index.cpp: number_atoms_by_home_slot), for lanes that visit different words as for lanes that visit the
//       same.  (Until round 4: [row][odd stride] cells of 8 or 16 {any, veto} bytes — 3.1-3.3 LDS passes per gather.)
//       The veto plane only holds the columns of words that HAVE a veto bit in some row (round 6; half of the words of the
//       configs[4] program have none, and a chunk holds as many words as fit LDS): those words come first in the chunk —
//       local words [0, n_veto), veto column of word w at the same distance behind its `any` column as before — and when
//       there are others one all-zero column follows (column n_veto of the plane): a lane whose word carries no kNsWordVeto
//       flag reads THAT column where a mixed tile reads veto cells (kt_scan.h).
//     nsl_rng[n] = {begin, end} of namespace n's word list in NsWord[] — namespaces with the same list share it.
//   off_term_t            u32 [n_words*64]  throttle row | kTerm*       (check: staged into LDS next to the CheckRec flags)
//   off_term_rank         u16 [n_words*64]  chunk-local rank | kRankAdj  (aggregate)
//   off_term_g            u32 [n_words*64]  selector-program term of the number (only read for `slow` candidates)
struct BmChunk {
  uint32_t w0, n_words;
  uint32_t rank0, n_thr;
  uint32_t img_off, lds_bytes;
  uint32_t off_hdr, off_nsl_rng, off_nsl;
  uint32_t off_term_t, off_term_rank, off_term_g;
  uint32_t col_rows;  // cells per word column of a plane (rows rounded up to 32)
  uint32_t n_veto;    // local words [0, n_veto) have a column in the veto plane (rich images; 0 otherwise)
  uint32_t zero_col;  // byte offset (from the rows) of the all-zero column behind the veto plane's n_veto columns; 0: every word has a veto column
  uint32_t slab_off;  // (aggregate) byte offset of this chunk's tables in the slab scratch / 16
  uint32_t has_slow;  // some term of the chunk needs the generic walk
  uint32_t img_bytes;
  uint32_t has_adj;   // some throttle of the chunk has several terms (the check's wordwise form then applies its run masks)
  // the namespace rows the chunk's word lists serve: [ns_base, ns_base + ns_cnt).  A classic index serves all of them
  // (0, 0xFFFFFFFF); the chunks of an anchored index (host/kt_anchor.h: one sub-index per anchor atom, concatenated) each
  // serve the block of VIRTUAL namespaces of their anchor — nsl_off is then indexed by (namespace - ns_base).  Not read by
  // the kernels yet (NEXT.md #1b).
  uint32_t ns_base, ns_cnt;
};

inline uint32_t image_col_rows(uint32_t rows) { return (rows + 31u) & ~31u; }

struct AtomId {
  uint32_t atom, id;
  uint32_t home;  // the atom slot of a pod's row kt_translate_pods tries first (the home slot of the atom's key)
};
// atom_table entry: atom | id << 32 | home slot << 48 (0 = empty)
constexpr int kAtomHomeShift = 48;
constexpr uint32_t kAtomIdMask = 0xFFFFu;

struct HostIndex {
  std::vector<uint32_t> slow_thr;
  uint32_t bm_words = 0;  // W: 64-bit words per full bitmap row
  uint32_t bm_rows = 0;   // A + 1 (row 0 = no atom: all zero)
  bool has_veto = false;  // some indexed term has a NotIn / DoesNotExist requirement
  uint32_t max_need = 0;  // largest number of counted positive requirements of a term (<= 5; > 3: the NEED = 5 instantiations)
  uint32_t n_pair_keys = 0, n_key_atoms = 0;  // distinct keys behind the referenced pair atoms / referenced key atoms
  uint32_t n_keys = 0;                        // distinct keys referenced either way
  bool has_slow = false;  // some indexed term needs the generic walk
  bool has_long = false;  // some indexed throttle has more than 64 terms: its run of numbers spans words (see build_index)
  uint32_t la = 8;        // atom slots per pod the scan kernels are instantiated for (8 / 16 / 32)
  bool rich = false;      // image in the {any, veto} form, kernels in the <VETO, NEED 3> instantiation
  std::vector<AtomId> atoms;               // referenced atom -> id (1..A)
  std::vector<uint32_t> atom_key;          // key id of every entry of `atoms` (a pair's key; a key atom's own key)
  std::vector<uint64_t> atom_table;        // open-addressing table for the device: atom | id << 32 | home slot << 48, 0 = empty
  std::vector<BmChunk> bm_chunks;
  std::vector<unsigned char> bm_images;    // chunk images back to back
  std::vector<uint32_t> bm_rank_t;         // dense rank (one per group, number order) -> throttle row
  std::vector<uint32_t> thr_ngrp;          // [T] groups (= ranks = slab records) of every throttle row (index_group_counts)
  std::vector<uint32_t> nogroup;           // throttle rows without any group: not live, or on the slow list
  std::vector<uint32_t> bm_chunk_ns;       // [chunks][ns_words] bit n: namespace n has words in the chunk
  uint32_t ns_words = 0;                   // 32-bit words per row of bm_chunk_ns
  uint32_t bm_max_lds = 0, bm_max_thr = 0, bm_max_words = 0;
  uint64_t bm_slab_bytes = 0;  // aggregate scratch: one table per (chunk, workgroup)
  uint32_t cut_chk_budget = 0; // the check budget the chunks were last cut for
  uint32_t cut_thr_bytes = 0;  // the record size the aggregate's tables and slabs were sized for (plain, or the packed fold's)
  bool agg_windowed = false;   // ONE chunk whose table of records does not fit the aggregate's LDS: it scans the chunk per window of ranks
  bool cut_grouped = false;    // the chunks are those of the grouped plan (kt_index.cpp: cut_chunks): per group of namespaces
  uint32_t img_words = 0;      // words over all chunk images (>= bm_words: the grouped plan copies words); BmChunk::w0 counts in these
  int64_t ns_word_visits = 0;  // entries of all word lists = (namespace, visited word) pairs
  int64_t ns_chunk_visits = 0; // (namespace, chunk that serves it with at least one word) pairs
  // the numbered program before it is cut into chunks (host only; cut_chunks reads it): full bitmap rows [rows][W],
  // namespace rows [n_ns][W], per-word headers, per-number tables [W * 64]
  uint32_t n_ns = 0;
  std::vector<uint64_t> full_any, full_veto, full_nsrows;
  std::vector<WordHdr> full_hdr;
  std::vector<uint32_t> full_term_t, full_term_g, full_term_rank;
  std::vector<uint8_t> full_real;
};

// One throttle's record in the aggregate's LDS table / slab (16-byte granules):
//   v i64[D] | presence mask u32 | pods u32                      (full scans)
//   v i64[D] | pods carrying the key u32[D] | pods u32           (counts mode: incremental engines)
__host__ __device__ inline uint32_t agg_rec_bytes(int D, bool counts) {
  return (uint32_t)(((size_t)D * 8 + (counts ? (size_t)D * 4 + 4 : 8) + 15) & ~(size_t)15);
}

// Packed request words (round 3).  The aggregate's fold costs one LDS atomic INSTRUCTION per non-zero dimension and match
// whatever the number of active lanes (4.8 ns each per CU: `profiles/r02_lds_atomics_microbench.txt`), and that is what
// bounds it on dense programs.  When no pod carries a negative request, a pod's contribution to a throttle is D small
// non-negative numbers and a count of one; per launch the host proves how large the sum of any field over the pods ONE
// workgroup scans can get (largest request >> common trailing zero bits, times the pods per workgroup) and lays the
// fields out in 1..4 64-bit words: the pod's words are built once (kt_build_scan_view), the fold adds whole words — the
// pod count and the two everyday resources (cpu, memory) normally share word 0, so a match costs ONE atomic — and the
// slab reduction takes the fields apart again.  No field can carry into its neighbour: sum <= n_slab_pods * max < 2^width.
struct PackPlan {
  uint32_t nw = 0;         // 64-bit words per pod (1..8; more than 4: the NW = 8 instantiations of the fold — round 6, engines with
                           // more than 8 dimensions); 0: the requests of this engine do not pack (negative values, > 8 words)
  uint32_t stride = 0;     // words per pod in the scan view (2, 4 or 8: whole 16-byte loads)
  uint32_t rec_bytes = 0;  // record of the LDS table / slab: nw words, then one word whose low half is the OR of the
                           // request-key masks of pods that carry a key with the value 0 (+ padding)
  uint8_t word[16] = {0}, pos[16] = {0}, width[16] = {0}, shift[16] = {0};
  uint8_t cnt_width = 0;   // the pod count sits in word 0 from bit 0
  // Summing whole words over the slabs (kPackHeadroomBits: up to 256 of them) without taking the fields apart first:
  // the TOP field of a word (from bit top_pos[k] up) is shifted out and summed by itself; the fields below it alternate
  // between two classes (even[k] = mask of the 1st, 3rd, ... field of word k), and every field is at least
  // kPackHeadroomBits wide — so inside `word & even[k]` (and `word & low[k] & ~even[k]`, low[k] = bits below the top
  // field) every field has that many zero bits above it, and the sum of 256 such words carries nowhere.  desc[d] /
  // cnt_desc: where the lane of dimension d / the pod count is found afterwards (pack_desc).
  uint64_t even[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint8_t top_pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t desc[16] = {0};
  uint32_t cnt_desc = 0;
};
constexpr int kPackHeadroomBits = 8;
constexpr uint32_t kPackMaxWords = 8;
// What an engine of D dimensions asks of the packed fold: up to 4 words at D <= 8 (every fold form takes those — 40-byte records
// at most, half of the plain 80), up to 8 beyond (kt_aggregate_bitmap's NW = 8 instantiations: 72-byte records against 144)
__host__ __device__ inline uint32_t pack_max_words(int D) { return D <= 8 ? 4u : kPackMaxWords; }
// the largest packed record of such a plan: the words + the zero-key word, padded to an odd count of units (make_pack_plan)
__host__ __device__ inline uint32_t packed_rec_max(int D) { return (pack_max_words(D) + 1u) * 8u; }
constexpr int kPackClasses = 3;  // even fields, odd fields, the top field
// desc: bits 0-4 = 4 * word + class (0: even mask, 1: odd, 2: top field — already shifted down), 8-13 = pos,
// 16-22 = bits to keep (field + headroom; 0: no field), 24-29 = shift
__host__ __device__ inline uint32_t pack_desc(uint32_t word, uint32_t cls, uint32_t pos, uint32_t wext, uint32_t shift) {
  return (word * 4u + cls) | pos << 8 | wext << 16 | shift << 24;
}
// or_abs[d]: OR of every |request| fed for dimension d (its trailing zeros are common to all of them);
// pad_odd: pad the record to an odd number of 8-byte words (LDS bank spread) instead of the smallest size
// max_words: the most words the caller's fold takes (4: the fused sweep's and the plain instantiations; 8: kt_aggregate_bitmap's NW = 8)
PackPlan make_pack_plan(int D, const unsigned __int128* max_abs, const uint64_t* or_abs, bool neg_seen, uint64_t n_slab_pods, bool pad_odd,
                        uint32_t max_words = kPackMaxWords);

// what kt_patch_scan_views (kt_kernels.hip) needs of the scan views a pod event batch is applied to in place
struct ViewPatch {
  uint64_t* vc_meta;  // the aggregate's view of the countable pods (nullable)
  uint16_t* vc_latom;
  int64_t* vc_req;
  uint64_t* vc_pk;
  int64_t* vc_rows;
  int32_t* pos_c;
  unsigned long long* n_c;  // device counter of listed rows
  int64_t cap_c;
  uint64_t* va_meta;  // the check sweep's view of all rows (namespace-ordered programs; nullable)
  uint16_t* va_latom;
  int32_t* pos_a;
  int64_t rows_a;     // rows the all-rows list covers
  uint32_t by_ns;
  uint32_t* dirty;
  PackPlan pk;
};

// slot of an atom in an open-addressing table of 2^k entries (linear probing)
__host__ __device__ inline uint32_t atom_slot(uint32_t atom, uint32_t mask) { return ((atom * 0x9E3779B1u) >> 7) & mask; }

struct IndexDev {
  uint32_t* slow_thr = nullptr;
  uint32_t n_slow = 0;
  unsigned char* bm_blob = nullptr;  // chunk images
  BmChunk* bm_chunks = nullptr;
  uint32_t* bm_rank_t = nullptr;
  uint32_t* thr_ngrp = nullptr;   // [T]
  uint32_t* nogroup = nullptr;    // [n_nogroup]
  uint32_t* grp_arrive = nullptr; // [T] arrival counters of kt_reduce_finalize_packed (zero between launches)
  uint32_t n_nogroup = 0;
  size_t cap_thr_ngrp = 0, cap_nogroup = 0, cap_grp_arrive = 0;
  uint32_t* bm_chunk_ns = nullptr;
  uint32_t ns_words = 0;
  uint64_t* atom_table = nullptr;
  uint32_t atom_mask = 0;
  std::vector<BmChunk> h_chunks;     // host copy (launch planning)
  uint32_t n_chunks = 0, bm_max_lds = 0, bm_max_thr = 0, bm_max_words = 0, bm_rows = 0;
  uint64_t bm_slab_bytes = 0;
  uint32_t has_veto = 0, max_need = 0, n_atoms = 0, has_key_atoms = 0, la = 8;
  uint32_t bm_words = 0;  // 64-bit words over all chunk images (HostIndex::img_words): what the verdict images are sized for
  uint32_t cut_thr_bytes = 0;  // HostIndex::cut_thr_bytes
  bool has_long = false;       // HostIndex::has_long: only the match-by-match instantiations may scan this index
  bool rich = false;
  size_t cap_bm_blob = 0, cap_bm_chunks = 0, cap_bm_rank_t = 0, cap_bm_chunk_ns = 0, cap_atom_table = 0, cap_slow = 0;
};

// LDS budgets: a chunk must satisfy
//   check     : lds_bytes + n_words*chk_word (term info + verdict masks: check_word_lds(D)) <= chk_budget
//   aggregate : lds_bytes + n_words*(64*2 + 16) (ranks, run masks) + n_thr * thr_bytes <= agg_budget
// (both kernels lay LDS out once, for the maxima over all chunks, so the maxima have to fit too).
void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns,
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes,
                 int max_labels, const std::vector<uint32_t>* adm_in = nullptr, uint32_t chk_budget_full = 0,
                 uint32_t chk_word = kCheckWordLds, const HostIndex* atoms_from = nullptr, uint32_t thr_bytes_packed = 0);
// atoms_from (optional): the atom numbering is IMPOSED — that of another index (its atoms / atom_key / atom_table / la) —
// instead of derived from this program: the sub-indexes of an anchored index (host/kt_anchor.h) share the numbering of the
// full program, against which the pods' atom rows are translated once; the image form is then always the rich one
// adm_in (optional): the namespace admission set of every term as bit words [terms][(n_ns + 31) / 32] (= ns_term_ok
// transposed), when the caller holds it already; chk_budget_full (optional): the check budget to cut for when the
// program needs several chunks anyway (one cut instead of two)
// out[n][gw] (bit g of row n) = in[g][nsw] (bit n of row g)
void transpose_term_ns_bits(const std::vector<uint32_t>& in, size_t G, uint32_t nsw, uint32_t n_ns, uint32_t gw, std::vector<uint32_t>& out);
void parallel_for(size_t n, size_t min_per_part, const std::function<void(size_t, size_t, size_t)>& f, size_t* parts_out);
// chunk images of an index build_index numbered, for other LDS budgets (no renumbering)
// chk_word: the check kernel's LDS bytes per word beside the image (check_word_lds(D); the default is the worst case)
// thr_bytes_packed (optional, < thr_bytes): the record size to cut for when the program needs several chunks — the packed
// fold's records (PackPlan) are smaller than the plain ones, a chunk then holds more words; the plain fold does not fit such
// chunks (HostIndex::cut_thr_bytes tells; the engine cuts again for plain records when it needs that fold)
void cut_chunks(HostIndex& out, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes, uint32_t chk_word = kCheckWordLds,
                uint32_t thr_bytes_packed = 0);
// groups per throttle row and the rows without any, from bm_rank_t (after the final cut)
void index_group_counts(HostIndex& h, uint32_t T);
hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s);
void release_index(IndexDev& d);

// translated atoms per pod the scan kernels are instantiated for, given the program and the label capacity L: a pod
// carries one atom per label whose key some selector references.  8 / 16 / 32; pods that still carry more relevant
// atoms are flagged kMetaOverflow.
inline uint32_t atom_slots(uint32_t n_keys, int L) {
  const uint32_t ub = std::min<uint32_t>((uint32_t)L, n_keys);
  return ub <= 8 ? 8u : ub <= 16 ? 16u : 32u;
}

struct PodTable;
struct SelProgram;
// LDS the two scan kernels need beside the chunk image and its per-term / per-throttle tables
uint32_t aggregate_fixed_lds();
uint32_t check_fixed_lds();
uint32_t check_word_lds(int D);  // check: bytes per 64-bit word of term numbers beside the image (TermInfo + WordVerdict<DT>)
// which pods an aggregate scan covers and how they enter the target buffer
struct AggScan {
  int64_t n = 0;                 // pods
  const int64_t* rows = nullptr; // device list of pod rows, or
  int64_t row0 = 0;              // ... the contiguous range row0 + [0, n)
  bool counts = false;           // exact per-key pod counts (incremental engines) instead of presence masks
  int sign = 1;                  // -1: remove the scanned pods' contribution (delta scans)
  bool nonneg = false;           // no pod carries a negative request (lets the scan skip most presence updates)
  bool overflow_pods = false;    // some pod is flagged kMetaOverflow: its lane walks every throttle over the raw labels
  bool by_ns = false;            // `rows` is ordered by namespace (launch_order_rows_by_ns): contiguous tile ranges per
                                 // workgroup, chunks without words of the range's namespaces skipped
  const uint64_t* v_meta = nullptr;   // by_ns: scan-ordered copies of the listed pods' meta words, atom rows and request
  const uint16_t* v_latom = nullptr;  //        rows (launch_build_scan_view) — record j belongs to pod rows[j]
  const int64_t* v_req = nullptr;
  uint32_t* slab_tag = nullptr;  // [chunks][kSlabTagStride] epoch of the last launch that spilled this (chunk, workgroup) slab
  uint32_t epoch = 0;            // this launch's epoch (> 0, different from the previous launches')
  const uint32_t* wg_range = nullptr;  // by_ns: record range of every workgroup (plan_wg_ranges for aggregate_blocks(n) workgroups)
  int wg_range_G = 0;                  //        ... and the number of workgroups they were planned for
  const PackPlan* pk = nullptr;  // packed fold (full scans over the scan view only): the plan v_pk was built with
  const uint64_t* v_pk = nullptr;  // [n][pk->stride] packed request words, scan order
  int limb = 0;                  // wide sums: the limb of every request this scan adds (limb_of, kt_device.h)
  bool small_window = false;     // test switch: fold through rank windows of 64 records whatever fits (kt_kernels_aggregate.hip)
  bool defer_reduce = false;     // packed scans: leave the slabs as they are — kt_reduce_finalize_packed takes them from there
  mutable int launched_blocks = 0;  // out: workgroups (= slabs per chunk) of the scan launch
  mutable bool launched_packed = false;
};
constexpr uint32_t kSlabTagStride = 256;  // workgroups an aggregate launch may have (one per CU)
// workgroups of an aggregate launch over n listed pods, and the most pods one of them scans (the packed fields are sized
// for it)
int aggregate_blocks(int64_t n_rows);
uint64_t aggregate_slab_pods(int64_t n_rows, int blocks);
// sp_dev: device-resident copy of sp.  Both return the symbol of the scan kernel they dispatched, nullptr when a chunk
// does not fit the kernel's LDS.  after_scan (nullable) is invoked on the host right after the scan kernel is enqueued
// and before the slab reduction kernel — the engine uses it to bracket the two kernels with separate timing events.
const char* launch_aggregate_indexed(const PodTable& pods, const AggScan& scan, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, unsigned long long* partial, void* slab, hipStream_t s,
                              const std::function<void()>& after_scan = nullptr);
// the slab reduction of a packed scan + kt_finalize as one launch (kt_kernels_finalize.hip: kt_reduce_finalize_packed); one GPU
struct ThrTables;
struct ReconcileOut;
struct ReqBound;
void launch_reduce_finalize_packed(const ThrTables& tt, const SelProgram& sp, int D, const IndexDev& ix, const PackPlan& pk, const void* slab,
                                   int n_slabs, const uint32_t* slab_tag, uint32_t epoch, unsigned long long* partial, bool consume, int64_t now_s,
                                   int32_t now_ns, bool apply, const ReconcileOut& out, void* recs, int rec_DT, bool rec_eq, const ReqBound& vmax,
                                   hipStream_t s, const uint8_t* row_mask, bool scan_adds_rows);
// small launches (n <= kCheckSmallMax): one workgroup per (chunk, tile); see kt_check_bitmap's SMALL instantiation
constexpr int64_t kCheckSmallMax = 256;
struct CheckSmall {
  uint32_t* ticket;        // device, >= kCheckSmallMax / 64 words, zero between launches
  uint64_t* host_summary;  // pinned host buffer of kCheckSmallMax words (device-accessible), or nullptr
  uint32_t n_inline;       // 0, or n (<= 8): the pod rows are passed by value
  int64_t inline_rows[8];
};
// namespace-ordered sweep: scan-ordered copies of the listed pods' meta words / atom rows, and n words of scratch
// that carry the class counters from chunk to chunk
struct CheckByNs {
  const uint64_t* v_meta;
  const uint16_t* v_latom;
  uint64_t* carry;
  // TermInfo + WordVerdict of every word (launch_build_verdict_images, for the CheckRecs this launch reads), or nullptr:
  // every workgroup then rebuilds them per chunk
  const unsigned char* wv_img = nullptr;
  uint32_t wv_total_words = 0;
  // record range of every workgroup of the sweep (plan_wg_ranges for check_sweep_blocks(n) workgroups), or nullptr:
  // fixed ranges of ceil(tiles / workgroups) tiles
  const uint32_t* wg_range = nullptr;
  int wg_range_G = 0;  // the workgroups the ranges were planned for (a launch with another grid ignores them)
};
int check_sweep_blocks(int64_t n);  // workgroups of a namespace-ordered lean sweep over n pod rows (one per CU)
// the per-word check tables of the whole index in global memory: TermInfo [total_words][64], then WordVerdict [total_words]
// (total_words = HostIndex::bm_words) — built once per generation of CheckRecs instead of once per (workgroup, chunk)
size_t verdict_images_bytes(uint32_t total_words, int D);
void launch_build_verdict_images(const IndexDev& ix, uint32_t total_words, const void* recs, int T, int D, void* out, hipStream_t s);
const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s, const CheckSmall* small = nullptr, bool overflow_pods = false,
                          const CheckByNs* by_ns = nullptr, bool one_per_cu = false);
// kt_sweep: the PreFilter sweep of pod rows [0, n) and the packed reconcile scan of the same rows as ONE launch
// (kt_check_bitmap's AGG instantiation: single-chunk programs without a slow list; pk sized for aggregate_slab_pods(n,
// aggregate_blocks(n))).  The slabs are left for launch_reduce_finalize_packed (*launched_blocks of them).  nullptr: not
// dispatchable — the caller runs launch_check_indexed and launch_aggregate_indexed one after the other.
const char* launch_sweep_indexed(const PodTable& pods, int64_t n, const SelProgram& sp, const SelProgram* sp_dev, const IndexDev& ix,
                                 const void* recs, uint64_t* summary, const PackPlan& pk, void* slab, uint32_t* slab_tag, uint32_t epoch,
                                 int* launched_blocks, hipStream_t s);
// PreFilter of n <= 8 pods without staging, copies or a stream synchronisation (kt_kernels_few.hip): one wave per index
// chunk, summaries + sequence number to pinned host memory.  false: not dispatched (slow-list throttles; the caller
// also keeps programs with `slow` term shapes and overflow pods away).
bool launch_check_few(const PodTable& pods, int n, const int64_t* rows_host, const SelProgram& sp, const IndexDev& ix, const void* recs,
                      unsigned long long* acc, uint32_t* ticket, uint64_t* host_summary, uint64_t* host_seq, uint64_t seq, hipStream_t s);
// by_ns: rows_dev lists ALL n pod rows ordered by namespace (launch_order_rows_by_ns); summary / status are then
// indexed by POD ROW (as a launch without a row list would), the scan runs in namespace order
// labels -> atom ids for pod rows [row0, row0+n) or rows[0..n) (after ingest / after a program change)
// n_overflow (device counter): valid pods with more relevant atoms than PodTable::LA
void launch_translate_pods(const PodTable& pods, int64_t n, const int64_t* rows_dev, int64_t row0, const IndexDev& ix,
                           unsigned long long* n_overflow, hipStream_t s);

}  // namespace kt
