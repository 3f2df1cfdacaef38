// plan_common.h -- host-side pieces of the x-tiled plan shared by the two builders of its layout: the host builder
// (plan_host.h::build_tiled_plan) and the device builder (plan_gpu.hip::build_tiled_plan_gpu).  Both produce the same
// arrays, byte for byte; the policies that are not per-entry work -- value coding, the phase-1 work items -- live here
// once.  (The layout itself replaces SparseMatrix::cl_encode of the reference, src/sparse_matrix.cpp:122-399.)
#pragma once
#include "../../include/sparseharness_hip.h"
#include "kernels.hip.h"
#include "bits.hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace sh {

// Host threads for the plan build: the option, else the hardware's, at most 16.
static int build_threads(const sh_plan_options &opt) {
  int n = (int)std::thread::hardware_concurrency();
  if (opt.build_threads > 0) n = opt.build_threads;
  return std::max(1, std::min(n, 16));
}
// fn(item, thread) for every item in [0, n): items are handed out `grain` at a time from an atomic
// counter (bins and heavy rows differ a lot in size), on `threads` std::threads (no OpenMP: the
// library lives in processes that already carry an OpenMP runtime of their own)
template <class F> static void parallel_items(int64_t n, int64_t grain, int threads, F fn) {
  threads = (int)std::min<int64_t>(threads, std::max<int64_t>(1, (n + grain - 1) / grain));
  if (threads <= 1) {
    for (int64_t i = 0; i < n; i++) fn(i, 0);
    return;
  }
  std::atomic<int64_t> next{0};
  auto worker = [&](int th) {
    for (;;) {
      const int64_t i0 = next.fetch_add(grain);
      if (i0 >= n) return;
      for (int64_t i = i0; i < std::min(n, i0 + grain); i++) fn(i, th);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
  worker(0);
  for (auto &t : pool) t.join();
}

// Small open-addressing set of 4-byte value words with first-come codes (value dictionary of the tiled plan).
struct ValSet {
  static constexpr uint32_t VH = 16384, VEMPTY = 0xFFFFFFFFu;   // (up to VDICT16 words at a load factor of 1/4)
  std::vector<uint32_t> key, code, list;
  bool overflow = false;
  ValSet() : key(VH, 0u), code(VH, VEMPTY) {}
  static uint32_t hash(uint32_t b) { return (b * 2654435761u) >> 18; }   // 14 bits
  uint32_t find(uint32_t b) const {   // slot holding b, or the empty slot where it belongs
    uint32_t h = hash(b);
    while (code[h] != VEMPTY && key[h] != b) h = (h + 1) & (VH - 1);
    return h;
  }
  void add(uint32_t b) {
    const uint32_t h = find(b);
    if (code[h] != VEMPTY) return;
    if (list.size() == (size_t)VDICT16) { overflow = true; return; }
    key[h] = b; code[h] = (uint32_t)list.size();
    list.push_back(b);
  }
};

struct TiledHost {
  std::vector<RowBin> bins;
  std::vector<TileChunk> chunks;
  std::vector<LongRow> heavy;        // rows pre-reduced in phase 1: {row, slot0, nslots}
  std::vector<uint32_t> tval, gdest, gblk, lrp, obase;   // gblk: per 64 product groups of a bin {piece-start mask lo, hi, pieces started before, 0}
  std::vector<int32_t> ptab;                      // per (bin, piece): P group index of the piece start - its group index inside the bin
  std::vector<uint16_t> ptile;                    // per (bin, piece): its column tile (which pieces a tile of absorbing x words makes dead)
  std::vector<uint8_t> tcode;        // value coding (see kernels.hip.h): codes instead of tval
  std::vector<uint32_t> vdict;       // empty = raw values
  int vdict_used = 0;
  int code_bits = 0;                 // 8: one code per byte of tcode; 4: two per byte (<= 16 values); 16: two bytes per code (<= 4096 values)
  std::vector<uint16_t> tcol, pslot;
  int64_t stream_len = 0, p_len = 0, light_len = 0, heavy_base = 0;   // light_len: light stream entries (padding included); p_len: products in P
  int64_t light_entries = 0;         // light entries of the matrix (no padding)
  int32_t n_partials = 0;
  double tile_fill = 1.0;            // light (bin, tile) pieces / (bins x tiles): ~1 scattered columns, ~0 local columns
};

// bytes of the code stream / words of the dictionary for a code width
static inline size_t tcode_bytes(int code_bits, int64_t stream_len) {
  return (size_t)(code_bits == 4 ? stream_len / 2 : (code_bits == 16 ? stream_len * 2 : stream_len));
}
static inline size_t dict_words(int code_bits) { return code_bits == 16 ? (size_t)VDICT16 : (size_t)VDICT; }

// How one (bin, tile) piece with np pairs and ns single entries is laid out: pair blocks of two groups (4 pairs
// each; columns without a pair take a single in A and a padding entry in B), then the remaining singles four to a
// group (padding entries = padding products at the end).  A piece with pair blocks always has a singles group --
// all padding if need be -- that the builder puts in FRONT when the piece starts at an odd group index, so that
// every A lands on an even one.
struct PiecePack { int32_t blocks, sgroups, groups, products; };
__host__ __device__ static inline PiecePack pack_piece(int64_t np, int64_t ns) {
  const int64_t blocks = (np + 3) / 4, spare = 4 * blocks - np;
  int64_t sg = ((ns > spare ? ns - spare : 0) + 3) / 4;
  if (blocks > 0 && sg == 0) sg = 1;
  return PiecePack{(int32_t)blocks, (int32_t)sg, (int32_t)(2 * blocks + sg), (int32_t)(4 * (blocks + sg))};
}

// Phase timers of the plan build and the upload (tools builds only: SH_BUILD_TIMES=1 prints them); lap(nullptr) restarts the clock.
#ifdef SH_PLAN_EMULATE
static void lap(const char *what) {
  static std::chrono::steady_clock::time_point t_last;
  if (!getenv("SH_BUILD_TIMES")) return;
  const auto now = std::chrono::steady_clock::now();
  if (what) fprintf(stderr, "[build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
  t_last = now;
}
#else
static inline void lap(const char *) {}
#endif

// Value coding of the stream (see kernels.hip.h): `words` = the distinct 4-byte value words of the matrix (any order;
// sorted here so that the dictionary does not depend on who found them), `overflow` = there are more than VDICT.
// <= VDICT distinct bit patterns => the stream carries one-byte codes, <= 16 => four-bit codes, <= VDICT16 => two-byte
// codes (value_coding 8 or 4 as an upper limit of the code width rules the wider ones out).  Code 0 is the all-zero
// word (padding) unless exactly 16 finite non-zero values fill the four-bit table, in which case padding borrows code
// 0's value: its products are identity (x) finite == identity.  value_coding < 0 keeps raw values (the caller does not
// get here), 8 never packs nibbles.  On return dict.overflow says "raw values"; else code_bits is 4 or 8 and dict.list the table.
static inline void decide_value_coding(std::vector<uint32_t> &words, bool overflow, const sh_plan_options &opt, int &code_bits, ValSet &dict) {
  const bool bytes_only = opt.value_coding == 8;
  std::sort(words.begin(), words.end());
  const bool has_zero = !words.empty() && words[0] == 0u;
  bool all_finite = true;   // as floats: padding may then carry ANY code (identity (x) finite == identity in all four semirings)
  for (uint32_t b : words) all_finite = all_finite && ((b >> 23) & 0xFFu) != 0xFFu;
  dict = ValSet();
  if (overflow || words.size() > (size_t)VDICT16) {
    dict.overflow = true;
  } else if (!bytes_only && words.size() + (has_zero ? 0 : 1) <= 16) {
    code_bits = 4;
    dict.add(0u);                      // code 0 = the all-zero word: padding
    for (uint32_t b : words) dict.add(b);
  } else if (!bytes_only && words.size() == 16 && all_finite) {
    code_bits = 4;                   // 16 finite values and no zero among them: padding borrows code 0's value
    for (uint32_t b : words) dict.add(b);
  } else if (words.size() + (has_zero ? 0 : 1) <= (size_t)VDICT) {
    code_bits = 8;
    dict.add(0u);
    for (uint32_t b : words) dict.add(b);
  } else if (!bytes_only && words.size() + (has_zero ? 0 : 1) <= (size_t)VDICT16) {
    code_bits = 16;
    dict.add(0u);
    for (uint32_t b : words) dict.add(b);
  } else {
    dict.overflow = true;
  }
}

// Phase-1 work items from the per-tile run tables (run_start / run_len: the tile's light run in the stream, in entries;
// heavy_start / hrel: its heavy run; ob0[t]: the tile's first block in obase[]).  Fills H.chunks.
static inline void cut_work_items(const int CT, const std::vector<int64_t> &run_start, const std::vector<int64_t> &run_len,
                                  const std::vector<int64_t> &heavy_start, const std::vector<int64_t> &hrel,
                                  const std::vector<int64_t> &ob0, const sh_plan_options &opt, const int n_cus, TiledHost &H) {
  // 4. phase-1 work items: <= chunk entries of one tile's light run, or of one tile's heavy run
  //    (cuts are multiples of 64 groups from the run start, so wave boundaries are the ones assumed
  //    above).  Workgroups are dealt round-robin over the 8 XCDs
  //    (blocks b and b+8 share one, MI355X_MICROARCH.md), so chunk position p holds a chunk of a tile
  //    with tile % 8 == p % 8: every XCD then stages only its own eighth of x through its L2 instead
  //    of all of it (speed only; correctness does not depend on placement).
  // Entries per work item.  Every item stages a 128 KiB x tile (~2 us of a ~9 us item), so fewer, larger items cost
  // less in total, but the launch ends with its slowest workgroup and a CU needs a handful of items to even out.
  // Measured (equal cuts, 10 M / 200 M matrix and its 1/2, 1/4, 1/8 shards; profiles/r02_chunk_size_vs_shard_size.log):
  // 64 K entries is best while the launch still has >= 6 items per CU (-3 % at full size), 48 K below that
  // (-7 % on a 1/8 shard against 32 K).  opt.chunk > 0 overrides.
  auto items_at = [&](int64_t c) {
    int64_t n = 0;
    for (int t = 0; t < CT; t++) n += (run_len[(size_t)t] + c - 1) / c + (hrel[(size_t)t] + c - 1) / c;
    return n;
  };
  // Round 3 (pair folding, same box): 48 K 0.453, 64 K 0.443, 96 K 0.448, 128 K 0.437 ms at full size -> 128 K under the same rule.
  // (Six rounds per size, every round a fresh process = fresh allocations: 96 K 0.4455, 128 K 0.4360, 160 K 0.4330, 192 K 0.4324,
  // 256 K 0.4349 ms; odd multiples of 1024 entries -- no power-of-two stride between the workgroups' streams -- change nothing:
  // profiles/r03_ab_chunk_sizes_*.log.  Within the +-1 % that the placement of the arrays alone moves the time.)
  // Round 4, after phase 2's reducers had been rebalanced (same box, twice each, profiles/r04_ab_chunk_sizes_160K.log): 128 K 233.6 / 235.1 us of
  // phase 1, 160 K 228.2 / 228.7, 192 K 229.6 / 229.4, 224 K 228.8 -> 160 K while the launch keeps 5.5 items per CU.  (R-MAT-23, skewed tiles,
  // stays with its 64 K items: 160 K measured 169.8 us against 174-176, once, inside the spread between boxes.)
  const int64_t enough = 6ll * std::max(n_cus, 1);
  int64_t chunk = opt.chunk > 0 ? opt.chunk
                  : (2 * items_at(163840) >= 11ll * std::max(n_cus, 1) ? 163840
                     : (items_at(131072) >= enough ? 131072 : (items_at(65536) >= enough ? 65536 : 49152)));
  chunk = std::max<int64_t>(1024, chunk) & ~int64_t(64 * HSTRIP - 1);   // whole waves of heavy strips
  const bool xcd_order = opt.xcd_order != 0;
  // Which XCD's list a tile's chunks go to.  Uniform columns: tile % 8 (every XCD stages its own eighth of x).
  // Skewed columns (a graph's hub columns fill a few tiles) would leave one XCD with most of the work while
  // the workgroups dealt to the other seven return at once, so tiles are handed out by weight, heaviest first,
  // to the least loaded XCD, and a tile heavier than an XCD's fair share is cut across several XCDs (home_of()
  // then moves on to the next least loaded XCD after a fair share's worth of the tile's chunks).
  // tile_segs[t]: (entries of the tile up to which the XCD applies, XCD), ascending
  std::vector<std::vector<std::pair<int64_t, int>>> tile_segs((size_t)CT);
  {
    std::vector<int64_t> weight((size_t)CT, 0);
    for (int t = 0; t < CT; t++) weight[(size_t)t] = hrel[(size_t)t] + run_len[(size_t)t];
    const int64_t fair = std::max<int64_t>(chunk, H.stream_len / 8);
    bool uniform = true;   // no tile far above the mean: keep the plain tile % 8 order
    for (int t = 0; t < CT; t++) uniform = uniform && weight[(size_t)t] * CT <= 2 * H.stream_len + 2 * chunk * CT;
    std::vector<int> order((size_t)CT);
    for (int t = 0; t < CT; t++) order[(size_t)t] = t;
    if (!uniform)
      std::stable_sort(order.begin(), order.end(), [&](int a2, int b2) { return weight[(size_t)a2] > weight[(size_t)b2]; });
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t : order) {
      int64_t done = 0;
      do {
        int c = t & 7;
        if (!uniform)
          for (int k = 0; k < 8; k++) if (load[k] < load[c]) c = k;
        const int64_t take = uniform ? weight[(size_t)t] : std::min(weight[(size_t)t] - done, fair);
        load[c] += take;
        done += take;
        tile_segs[(size_t)t].emplace_back(done, c);
      } while (done < weight[(size_t)t]);
    }
  }
  // XCD list of the chunk that starts `before` entries into tile t
  auto home_of = [&](int t, int64_t before) -> int {
    if (!xcd_order) return 0;
    for (const auto &sg : tile_segs[(size_t)t])
      if (before < sg.first) return sg.second;
    return tile_segs[(size_t)t].empty() ? (t & 7) : tile_segs[(size_t)t].back().second;
  };
  // cut one run into chunks
  std::vector<TileChunk> per_xcd[8];
  // (Guided sizes -- the first 60..75 % of every run in items of 128..256 K entries, the rest in items a quarter that
  // size handed out behind all the big ones -- were measured and lost: 0.446..0.454 vs 0.435 ms same box,
  // profiles/r03_ab_guided_chunk_sizes.log; plain 128 K items: 0.431.)
  auto cut_run = [&](int t, int64_t start, int64_t len, bool heavy) {
    // equal cuts: a run of 56 K entries becomes 2 x 28 K, not 32 K + 24 K (the launch ends with its slowest workgroup)
    const int64_t pieces = (len + chunk - 1) / chunk;
    const int64_t cut = pieces > 0 ? std::min<int64_t>(chunk, ((len + pieces - 1) / pieces + 64 * HSTRIP - 1) & ~int64_t(64 * HSTRIP - 1)) : chunk;
    for (int64_t s0 = start; s0 < start + len; s0 += cut) {
      const int64_t e0 = std::min<int64_t>(s0 + cut, start + len);
      // light chunks: ob0 = the chunk's first block of obase[] (cuts are multiples of 256 entries from the run start)
      TileChunk ch{t, (int32_t)s0, (int32_t)e0, (int32_t)(heavy ? start : e0),
                   (int32_t)(heavy ? H.heavy_base : 0), (int32_t)(heavy ? 0 : ob0[(size_t)t] + (s0 - start) / 256), 0, 0};
      per_xcd[home_of(t, (heavy ? run_len[(size_t)t] : 0) + (s0 - start))].push_back(ch);
    }
  };
  // ONE phase-1 launch, tile by tile -- a tile's light chunks, then its heavy chunks -- so that memory-bound
  // light chunks and the ALU-heavier heavy chunks are in flight together.
  for (int t = 0; t < CT; t++) {
    cut_run(t, run_start[(size_t)t], run_len[(size_t)t], false);
    cut_run(t, heavy_start[(size_t)t], hrel[(size_t)t], true);
  }
  // per-XCD lists interleaved so that position p holds a chunk of XCD p % 8 (empty fillers where a list is short)
  {
    size_t longest = 0;
    for (auto &v : per_xcd) longest = std::max(longest, v.size());
    if (!xcd_order)
      H.chunks.insert(H.chunks.end(), per_xcd[0].begin(), per_xcd[0].end());
    else
      for (size_t i = 0; i < longest; i++)
        for (int c = 0; c < 8; c++)
          H.chunks.push_back(i < per_xcd[c].size() ? per_xcd[c][i] : TileChunk{0, 0, 0, 0, 0, 0, 0, 0});   // empty filler
  }
}

// What the device builder leaves on the device: the big arrays of TiledHost (same contents, element counts as the host
// builder's vectors; every allocation carries the slack the kernels' wide loads need).  The small tables (bins, work
// items, heavy rows, dictionary) come back in TiledHost as usual.
struct TiledDevArrays {
  uint16_t *tcol = nullptr, *pslot = nullptr;
  uint8_t *tcode = nullptr;
  uint32_t *tval = nullptr, *gdest = nullptr, *gblk = nullptr, *obase = nullptr, *lrp = nullptr;
  int32_t *ptab = nullptr;
  uint16_t *ptile = nullptr;
  size_t n_tcol = 0, n_pslot = 0, n_tcode = 0, n_tval = 0, n_gdest = 0, n_gblk = 0, n_obase = 0, n_lrp = 0, n_ptab = 0;
  void release();   // frees what is still set (a builder that failed half-way; the caller after adopting nothing)
};
// slack bytes behind the arrays (the host path's DEV_ARRAY calls use the same numbers)
constexpr size_t SLACK_TCODE = 64, SLACK_WIDE = 16;

// The same layout as plan_host.h::build_tiled_plan, built on the device from the CSR arrays already there (plan_gpu.hip).
// 1: built; 0: the plan does not apply (the host builder's own refusals: it would say no as well); -1: a device step
// failed (`why` says which) and the caller falls back to the host builder.
int build_tiled_plan_gpu(hipStream_t stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *h_rp, const int32_t *d_rp,
                          const int32_t *d_ci, const uint32_t *d_val, const sh_plan_options &opt, int n_cus, TiledHost &H,
                          TiledDevArrays &D, std::string &why);

// ---- the (or,and) semiring on bits (see bits.hip.h) -----------------------
struct BitsHost {
  std::vector<BitsItem> items;
  std::vector<uint32_t> ent;     // (empty when the device builder left the entries on the device)
  std::vector<int32_t> bsub, rr_item0;
  int32_t n_rr = 0, n_ct = 0;
  int64_t entries = 0;   // entries with a non-zero value and a column in range
  int64_t ent_len = 0;   // words of the entry array (padding and the 8 trailing words included)
};
// From the live entries per cell (cell = (row range, column block, sub-range), cnt[ncell + 1]): where every cell's
// entries start (each padded to a multiple of 8), and the work items: blocks with more than 2^20 entries are cut at
// sub-range boundaries.  false: the entry array would exceed int32 indexing.
static inline bool bits_starts_and_items(BitsHost &H, const std::vector<int64_t> &cnt, std::vector<int64_t> &start) {
  const int64_t ncell = (int64_t)H.n_rr * H.n_ct * BITS_NSUB;
  int64_t pos = 0;
  H.entries = 0;
  for (int64_t k = 0; k < ncell; k++) { start[(size_t)k] = pos; pos += (cnt[(size_t)k] + 7) & ~int64_t(7); H.entries += cnt[(size_t)k]; }
  start[(size_t)ncell] = pos;
  if (pos > INT32_MAX - 8) return false;
  const int64_t max_item = 1 << 20;
  H.rr_item0.assign((size_t)H.n_rr + 1, 0);
  for (int32_t rr = 0; rr < H.n_rr; rr++) {
    H.rr_item0[(size_t)rr] = (int32_t)H.items.size();
    for (int32_t ct = 0; ct < H.n_ct; ct++) {
      const int64_t c0 = ((int64_t)rr * H.n_ct + ct) * BITS_NSUB;
      if (start[(size_t)(c0 + BITS_NSUB)] == start[(size_t)c0]) continue;   // an empty block
      for (int sub0 = 0; sub0 < BITS_NSUB;) {
        int sub1 = sub0 + 1;
        while (sub1 < BITS_NSUB && start[(size_t)(c0 + sub1 + 1)] - start[(size_t)(c0 + sub0)] <= max_item) sub1++;
        if (start[(size_t)(c0 + sub1)] > start[(size_t)(c0 + sub0)]) {
          BitsItem it{rr, ct, (int32_t)start[(size_t)(c0 + sub0)], (int32_t)start[(size_t)(c0 + sub1)], sub0, sub1, (int32_t)H.bsub.size(), 0};
          for (int k = sub0; k <= sub1; k++) H.bsub.push_back((int32_t)start[(size_t)(c0 + k)]);
          H.items.push_back(it);
        }
        sub0 = sub1;
      }
    }
  }
  H.rr_item0[(size_t)H.n_rr] = (int32_t)H.items.size();
  return true;
}
// The entry array of the bit-blocked layout built on the device (plan_gpu.hip): 1 built (*d_ent holds H.ent_len words
// + SLACK_WIDE bytes, H the items), 0 not applicable (as the host builder), -1 a device step failed.
int build_bits_plan_gpu(hipStream_t stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *d_rp, const int32_t *d_ci,
                        const uint32_t *d_val, BitsHost &H, uint32_t **d_ent, std::string &why);

} // namespace sh
