// plan_host.h -- the HOST builders of the launch layouts (included by engine.hip only): the CSR-stream schedule, the
// x-tiled two-phase layout and the bit-blocked (or,and) layout, from a CSR matrix in host memory.  The device-side
// builder of the same layouts is plan_gpu.hip; both end in plan_common.h::cut_work_items and produce the same arrays
// byte for byte (tests/test_builder_gpu.py).  Small matrices are built here (a few parallel host passes beat the
// launch latencies of the device builder below ~1 M entries), large ones on the device.
#pragma once
#include "kernels.hip.h"
#include "bits.hip.h"
#include "plan_common.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <vector>

using namespace sh;

// ---------------------------------------------------------------- matrix
// Launch schedule: greedy packing of consecutive rows into stream blocks
// (<= NNZ_BLK entries counted from the 16-byte-aligned start, <= ROWS_BLK
// rows); rows that do not fit alone become long rows cut into SEG_NNZ pieces.
static void build_schedule(int64_t rows, const int32_t *rp, std::vector<int32_t> &blk_row,
                           std::vector<LongSeg> &segs, std::vector<LongRow> &longs) {
  // blk_row holds (first row, one-past-last row) pairs of stream blocks only; long rows are
  // excluded by closing the current block before them (a stream block never spans a long row).
  int64_t r = 0;
  while (r < rows) {
    const int64_t s = rp[r];
    const int64_t base = s & ~int64_t(3);
    if ((int64_t)rp[r + 1] - base > NNZ_BLK) {
      // long row
      LongRow lr;
      lr.row = (int32_t)r;
      lr.slot0 = (int32_t)segs.size();
      lr.pad = 0;
      int32_t k = 0;
      for (int64_t p = s; p < rp[r + 1]; p += SEG_NNZ, k++) {
        LongSeg sg;
        sg.row = (int32_t)r;
        sg.s = (int32_t)p;
        sg.e = (int32_t)std::min<int64_t>(p + SEG_NNZ, rp[r + 1]);
        sg.slot = lr.slot0 + k;
        segs.push_back(sg);
      }
      lr.nslots = k;
      longs.push_back(lr);
      r++;
      continue;
    }
    int64_t r1 = r + 1;
    while (r1 < rows && r1 - r < ROWS_BLK && (int64_t)rp[r1 + 1] - base <= NNZ_BLK)
      r1++;
    blk_row.push_back((int32_t)r);
    blk_row.push_back((int32_t)r1);
    r = r1;
  }
}


// ---- x-tiled two-phase plan (see kernels.hip.h) ---------------------------
// Host-side layout construction.  The tile-major stream is a partition of the light entries by
// column tile and, inside a tile, by row bin.  A lane of phase 1 owns a GROUP of 4 consecutive stream
// entries and stores their 4 products with one 16-byte store.  Two entries of one row that fall
// into the same (bin, tile) piece form a PAIR that phase 1 folds into ONE product: pairs are laid
// out column-wise over two consecutive groups A, B (entry k of B pairs with entry k of A; A sits at
// an even group index of the tile's run, so the two lanes are the two halves of a lane pair): lane A
// adds B's products to its own (one DPP move each) and stores 4 products, lane B stores nothing.  A
// piece is a whole number of groups and yields a multiple of 4 products, so it starts 16-byte
// aligned in P.
static bool build_tiled_plan(int64_t rows, int64_t cols, int64_t nnz, const int32_t *rp,
                             const int32_t *ci, const uint32_t *val, const sh_plan_options &opt, int n_cus, TiledHost &H) {
  const int CT = (int)std::max<int64_t>(1, (cols + TCOLS - 1) / TCOLS);
  if (CT > 65535) return false;   // (tile numbers travel as 16-bit values in the builder; 2.1 G columns: shard the matrix)
  const bool fold = opt.fold != 0 && TCOL_FOLD != 0;
  lap(nullptr);
  // A row is "heavy" when it averages >= 8 entries per column tile (or cannot fit a bin): its
  // (row, tile) runs are summed inside phase 1 instead of travelling through P.
  const int64_t per_tile = std::max(1, opt.heavy_per_tile);
  // (capped at TBIN/4 so that a single light row fits a bin even when every entry is padded to 4)
  const int64_t heavy_thr = std::min<int64_t>(TBIN / 4, std::max<int64_t>(512, per_tile * CT));
  auto is_heavy = [&](int64_t r) { return (int64_t)rp[r + 1] - rp[r] >= heavy_thr; };
  auto tile_of = [&](int32_t c) -> int { return ((uint32_t)c < (uint32_t)cols) ? (c / TCOLS) : 0; };
  const int NT = build_threads(opt);

  // Per-thread scratch.  ent[]: the light entries of the bin being worked on, as (tile << 32 | local row << 16 ...)
  // would not fit: two parallel arrays sorted by tile with a counting sort (stable: row order, then CSR order).
  struct Scratch {
    std::vector<int32_t> count, touched;            // per tile
    std::vector<int64_t> pos;                       // per tile
    std::vector<int32_t> e_row, e_j, s_row, s_j;    // entries of a bin: unsorted / sorted by tile
    std::vector<int32_t> start;                     // per touched tile: first entry in s_*
    std::vector<int32_t> rcount, rtouched;          // row_tiles(): per-tile counts of one row
    std::vector<int32_t> np, ns;                    // per tile: pairs / singles of the bin being sized
    std::vector<int32_t> row_next;                  // per row of the bin: next product index inside the row
    std::vector<int32_t> lp, ls;                    // pairs / singles of a piece (index of the first entry in s_*)
  };
  std::vector<Scratch> scratch((size_t)NT);
  for (auto &sc : scratch) {
    sc.count.assign((size_t)CT, 0); sc.pos.assign((size_t)CT, 0); sc.rcount.assign((size_t)CT, 0);
    sc.np.assign((size_t)CT, 0); sc.ns.assign((size_t)CT, 0);
  }

  // fn(tile, entries of row r in that tile) for every tile the row touches (any order).  Short rows: a tiny insertion
  // sort of the tile numbers; long rows: counting into the per-tile scratch (std::sort per row made the build 4x slower).
  auto row_tiles = [&](Scratch &sc, int64_t r, auto fn) {
    const int32_t d = rp[r + 1] - rp[r];
    if (d <= 12) {
      uint16_t t[12];
      for (int32_t i = 0; i < d; i++) {
        const uint16_t v = (uint16_t)tile_of(ci[rp[r] + i]);
        int32_t k = i;
        while (k > 0 && t[k - 1] > v) { t[k] = t[k - 1]; k--; }
        t[k] = v;
      }
      for (int32_t i = 0; i < d;) {
        int32_t k = i + 1;
        while (k < d && t[k] == t[i]) k++;
        fn((int)t[i], k - i);
        i = k;
      }
    } else {
      sc.rtouched.clear();
      for (int32_t j = rp[r]; j < rp[r + 1]; j++) {
        const int t = tile_of(ci[j]);
        if (sc.rcount[(size_t)t]++ == 0) sc.rtouched.push_back(t);
      }
      for (int t : sc.rtouched) { fn(t, sc.rcount[(size_t)t]); sc.rcount[(size_t)t] = 0; }
    }
  };
  // 1. products per light row (= its length without folding; with folding sum over tiles of ceil(entries in the tile / 2)),
  //    light row offsets in products (heavy rows have light length 0 and carry bit 31) and row bins
  H.lrp.assign((size_t)rows + 1, 0u);
  {
    std::vector<int32_t> nprod((size_t)rows, 0);
    parallel_items((rows + 4095) / 4096, 4, NT, [&](int64_t blk, int th) {
      Scratch &sc = scratch[(size_t)th];
      for (int64_t r = blk * 4096; r < std::min(rows, (blk + 1) * 4096); r++) {
        if (is_heavy(r)) continue;
        const int32_t d = rp[r + 1] - rp[r];
        if (!fold || d <= 1) { nprod[(size_t)r] = d; continue; }
        int32_t n = 0;
        row_tiles(sc, r, [&](int, int32_t k) { n += (k + 1) / 2; });
        nprod[(size_t)r] = n;
      }
    });
    uint32_t acc = 0;
    int64_t le = 0;
    for (int64_t r = 0; r < rows; r++) {
      const bool hv = is_heavy(r);
      H.lrp[(size_t)r] = acc | (hv ? 0x80000000u : 0u);
      if (!hv) { acc += (uint32_t)nprod[(size_t)r]; le += rp[r + 1] - rp[r]; }
    }
    H.lrp[(size_t)rows] = acc;
    H.light_entries = le;
  }
  lap("products per row + lrp");
  auto light_off = [&](int64_t r) { return (int64_t)(H.lrp[(size_t)r] & 0x7FFFFFFFu); };
  // Bins are filled up to TBIN products.  (Sizing them so that every CU gets the same number of
  // bins was tried for shard-sized matrices: the smaller (bin, tile) pieces cost more than the
  // ragged last round saves -- 0.115 vs 0.105 ms on a 1/8 shard in round 1, 0.076 vs 0.075 ms with round 2's kernels.)
  // every (bin, tile) piece is padded to 4 products: leave room so that a padded bin never exceeds TBIN
  // (phase 2 prefetches exactly TBIN products per bin into registers)
  int64_t bin_target = std::max<int64_t>(TBIN / 4, (int64_t)TBIN - 3ll * CT);   // cnt + 3*min(CT,cnt) <= TBIN
  for (int64_t r = 0; r < rows;) {
    int64_t r1 = r + 1;
    while (r1 < rows && r1 - r < TBIN_ROWS && light_off(r1 + 1) - light_off(r) <= bin_target)
      r1++;
    RowBin b{};
    b.r0 = (int32_t)r; b.nr = (int32_t)(r1 - r); b.csr0 = (int32_t)light_off(r);
    H.bins.push_back(b);
    r = r1;
  }

  // 2.-3. The layout.  Each tile's stream is [light pieces, bins in order][heavy pieces, rows in
  //    order].  Built in passes so that the O(nnz) walks run on several host threads (work items handed
  //    out dynamically; each writes only its own bin's or row's ranges) and only O(pieces) prefix sums
  //    stay sequential:
  //      P1 (parallel, bins)        sort each bin's entries by tile, find the runs -> its piece list (groups, products), bin.n
  //      P2 (parallel, heavy rows)  entries per tile of each heavy row
  //      S1 (sequential)            bin.pstart, per-tile totals; stream / P position of every light piece
  //      S2 (sequential)            heavy pieces: hrel[t] is the running position inside tile t, which
  //                                 fixes where the 64-group wave boundaries of phase 1 fall, hence how
  //                                 a heavy (row, tile) piece splits into partials
  //      P3 (parallel)              value dictionary (per-thread sets, merged)
  //      P4 (parallel, bins)        fill the light stream (entries + fold flags), pslot, the piece tables (gblk, ptab)
  //      P5 (parallel, heavy rows)  fill the heavy stream and gdest
  //      P6 (parallel, tiles)       obase: where in P the products of every 64 stream groups start
  struct Piece { int32_t tile, cnt, groups, prods; int64_t pos, ppos; int32_t part0; };   // pos / ppos: stream / P position; part0: first partial (heavy)
  const int64_t n_bins = (int64_t)H.bins.size();
  std::vector<std::vector<Piece>> bin_pieces((size_t)n_bins);
  std::vector<int64_t> heavy_rows_idx;
  for (int64_t r = 0; r < rows; r++)
    if (is_heavy(r)) heavy_rows_idx.push_back(r);
  const int64_t n_heavy = (int64_t)heavy_rows_idx.size();
  std::vector<std::vector<Piece>> heavy_pieces((size_t)n_heavy);
  // the light entries of bin b sorted by tile (stable): sc.s_row / sc.s_j, tiles in sc.touched (ascending), first entry of
  // touched tile k in sc.start[k] (sc.start has one more element: the total)
  auto sort_bin = [&](Scratch &sc, const RowBin &b) {
    sc.touched.clear(); sc.e_row.clear(); sc.e_j.clear();
    for (int64_t r = b.r0; r < (int64_t)b.r0 + b.nr; r++) {
      if (is_heavy(r)) continue;
      for (int32_t j = rp[r]; j < rp[r + 1]; j++) {
        const int t = tile_of(ci[j]);
        if (sc.count[(size_t)t]++ == 0) sc.touched.push_back(t);
        sc.e_row.push_back((int32_t)(r - b.r0));
        sc.e_j.push_back(j);
      }
    }
    std::sort(sc.touched.begin(), sc.touched.end());
    sc.start.clear();
    int32_t acc = 0;
    for (int t : sc.touched) { sc.start.push_back(acc); sc.pos[(size_t)t] = acc; acc += sc.count[(size_t)t]; }
    sc.start.push_back(acc);
    sc.s_row.resize(sc.e_row.size()); sc.s_j.resize(sc.e_row.size());
    for (size_t i = 0; i < sc.e_row.size(); i++) {
      const int64_t q = sc.pos[(size_t)tile_of(ci[sc.e_j[i]])]++;
      sc.s_row[(size_t)q] = sc.e_row[i]; sc.s_j[(size_t)q] = sc.e_j[i];
    }
    for (int t : sc.touched) sc.count[(size_t)t] = 0;   // scratch back to all-zero
  };
  // the entries [a, e) of sc.s_* (one tile) as pairs (two consecutive entries of one row; none without folding) and singles
  auto find_runs = [&](Scratch &sc, int32_t a, int32_t e) {
    sc.lp.clear(); sc.ls.clear();
    for (int32_t i = a; i < e;) {
      if (fold && i + 1 < e && sc.s_row[(size_t)i + 1] == sc.s_row[(size_t)i]) { sc.lp.push_back(i); i += 2; }
      else { sc.ls.push_back(i); i += 1; }
    }
  };
  // P1 (sizes only: pairs and singles per (bin, tile) from the rows' per-tile counts; the entries are not moved yet)
  parallel_items(n_bins, 8, NT, [&](int64_t bi, int th) {
    RowBin &b = H.bins[(size_t)bi];
    Scratch &sc = scratch[(size_t)th];
    sc.touched.clear();
    for (int64_t r = b.r0; r < (int64_t)b.r0 + b.nr; r++) {
      if (is_heavy(r)) continue;
      row_tiles(sc, r, [&](int t, int32_t k) {
        if (sc.np[(size_t)t] == 0 && sc.ns[(size_t)t] == 0) sc.touched.push_back(t);
        if (fold) { sc.np[(size_t)t] += k / 2; sc.ns[(size_t)t] += k & 1; }
        else sc.ns[(size_t)t] += k;
      });
    }
    std::sort(sc.touched.begin(), sc.touched.end());
    auto &out = bin_pieces[(size_t)bi];
    out.clear();
    out.reserve(sc.touched.size());
    int64_t n = 0;
    for (int t : sc.touched) {
      const PiecePack pk = pack_piece(sc.np[(size_t)t], sc.ns[(size_t)t]);
      out.push_back(Piece{t, 2 * sc.np[(size_t)t] + sc.ns[(size_t)t], pk.groups, pk.products, 0, 0, 0});
      n += pk.products;
      sc.np[(size_t)t] = sc.ns[(size_t)t] = 0;
    }
    b.n = (int32_t)n;
  });
  lap("bins + P1 sizes");
  // P2
  parallel_items(n_heavy, 1, NT, [&](int64_t hi, int th) {
    const int64_t r = heavy_rows_idx[(size_t)hi];
    Scratch &sc = scratch[(size_t)th];
    sc.touched.clear();
    for (int32_t j = rp[r]; j < rp[r + 1]; j++) {
      const int t = tile_of(ci[j]);
      if (sc.count[(size_t)t]++ == 0) sc.touched.push_back(t);
    }
    std::sort(sc.touched.begin(), sc.touched.end());
    auto &out = heavy_pieces[(size_t)hi];
    out.clear();
    out.reserve(sc.touched.size());
    for (int t : sc.touched) {
      out.push_back(Piece{t, sc.count[(size_t)t], 0, 0, 0, 0, 0});
      sc.count[(size_t)t] = 0;
    }
  });
  lap("P2 heavy pieces");
  // S1: positions are relative to the tile's light run until the run starts are known
  std::vector<int64_t> run_len((size_t)CT, 0), run_plen((size_t)CT, 0), run_start((size_t)CT, 0), run_pstart((size_t)CT, 0), hrel(CT, 0);
  int64_t p_off = 0, n_pieces_total = 0, n_blocks_total = 0;
  for (int64_t bi = 0; bi < n_bins; bi++) {
    RowBin &b = H.bins[(size_t)bi];
    if (b.n > TBIN) return false;   // cannot happen with the limits above; phase 2 holds exactly TBIN products
    if (p_off + b.n > INT32_MAX) return false;
    b.pstart = (int32_t)p_off;
    p_off += b.n;
    b.pt0 = (int32_t)n_pieces_total;                         // its pieces in ptab[]
    b.gb0 = (int32_t)n_blocks_total;                         // its 64-group blocks in gblk[]
    n_pieces_total += (int64_t)bin_pieces[(size_t)bi].size();
    n_blocks_total += std::max<int64_t>(1, (b.n / 4 + 63) / 64);
    for (Piece &pc : bin_pieces[(size_t)bi]) {
      pc.pos = run_len[(size_t)pc.tile];
      pc.ppos = run_plen[(size_t)pc.tile];
      run_len[(size_t)pc.tile] += 4ll * pc.groups;
      run_plen[(size_t)pc.tile] += pc.prods;
    }
  }
  H.p_len = p_off;
  H.tile_fill = (n_bins > 0) ? (double)n_pieces_total / ((double)n_bins * CT) : 1.0;
  {
    // every tile's light run starts on a multiple of 64 groups: phase 1's waves then cover whole obase[] blocks
    int64_t pos = 0, ppos = 0;
    for (int t = 0; t < CT; t++) {
      run_start[(size_t)t] = pos; pos += (run_len[(size_t)t] + 255) & ~int64_t(255);
      run_pstart[(size_t)t] = ppos; ppos += run_plen[(size_t)t];
    }
    H.light_len = pos;
  }
  // S2: heavy pieces; positions relative to the tile's heavy run
  H.heavy.resize((size_t)n_heavy);
  {
    int64_t slots = 0;
    for (int64_t hi = 0; hi < n_heavy; hi++) {
      int32_t np = 0;
      for (Piece &pc : heavy_pieces[(size_t)hi]) {
        const int32_t padded = (pc.cnt + HSTRIP - 1) / HSTRIP * HSTRIP;   // whole strips: one lane of phase 1 sums a strip
        pc.pos = hrel[(size_t)pc.tile];
        pc.part0 = np;
        // one partial per wave-part: the piece's strips spread over this many 64-strip blocks of the heavy run
        const int64_t k0 = pc.pos / HSTRIP, k1 = (pc.pos + padded) / HSTRIP - 1;
        np += (int32_t)(k1 / 64 - k0 / 64 + 1);
        hrel[(size_t)pc.tile] += padded;
      }
      H.heavy[(size_t)hi] = LongRow{(int32_t)heavy_rows_idx[(size_t)hi], (int32_t)slots, np, 0};
      slots += np;
      if (slots > (int64_t)GD_SLOT_MASK) return false;   // the slot shares its gdest word with the scan hints
    }
    H.n_partials = (int32_t)slots;
  }
  int64_t total = (H.light_len + HSTRIP - 1) / HSTRIP * HSTRIP;   // heavy strips are read with 16- and 32-byte loads: keep them aligned
  const int64_t heavy_base = total;
  H.heavy_base = heavy_base;
  std::vector<int64_t> heavy_start(CT, 0);
  for (int t = 0; t < CT; t++) { heavy_start[t] = total; total += hrel[t]; }
  if (total > INT32_MAX - 8) return false;
  H.stream_len = total;
  if (nnz > 0 && H.stream_len > nnz + nnz / 4 + 4096 + 256ll * CT)
    return false; // padding would cost more than 25 %: keep the stream plan

  lap("S1 S2 positions");
  // P3: value dictionary: <= VDICT distinct bit patterns => the stream carries one-byte codes, <= 16 => four-bit
  // codes.  Code 0 is the all-zero word (padding) unless exactly 16 finite non-zero values fill the four-bit table,
  // in which case padding borrows code 0's value: its products are identity (x) finite == identity.
  // SH_VALCODE=off keeps raw values, SH_VALCODE=8 never packs nibbles.
  ValSet dict;
  {
    bool coded = opt.value_coding >= 0;
    if (coded) {
      std::vector<ValSet> part((size_t)NT);
      parallel_items((nnz + 65535) / 65536, 4, NT, [&](int64_t blk, int th) {
        ValSet &vs = part[(size_t)th];
        const int64_t e = std::min<int64_t>(nnz, (blk + 1) * 65536);
        for (int64_t j = blk * 65536; j < e && !vs.overflow; j++) vs.add(val[j]);
      });
      // the distinct words of the data, in ascending bit-pattern order (the same dictionary whatever the
      // thread count)
      ValSet all;
      for (const ValSet &vs : part) {
        if (vs.overflow) all.overflow = true;
        for (uint32_t b : vs.list) all.add(b);
      }
      std::vector<uint32_t> words(all.list);
      decide_value_coding(words, all.overflow, opt, H.code_bits, dict);
      coded = !dict.overflow;
    }
    if (coded) H.vdict = dict.list;
  }
  const bool coded = !H.vdict.empty();

  lap("P3 dictionary");
  // P4 / P5: fill
  if (coded) {
    H.vdict_used = (int)H.vdict.size();
    H.tcode.assign(tcode_bytes(H.code_bits, H.stream_len), 0);
    H.vdict.resize(dict_words(H.code_bits), 0u);
  }
  else H.tval.assign((size_t)H.stream_len, 0u);
  H.tcol.assign((size_t)H.stream_len, TCOL_IDENTITY);   // padding: the identity column, value word / code 0, no fold flag
  H.gdest.assign((size_t)(H.stream_len - heavy_base) / HSTRIP + 1, 0u);   // partial slot of every heavy strip
  H.pslot.assign((size_t)H.p_len, TSLOT_PAD);
  H.gblk.assign((size_t)(n_blocks_total + 1) * 4, 0u);
  H.ptab.assign((size_t)n_pieces_total + 1, 0);
  H.ptile.assign((size_t)n_pieces_total + 1, 0);
  std::atomic<bool> ok{true};
  auto put_entry = [&](int64_t pos, int32_t j) {
    const int32_t c = ci[j];
    const bool in_range = (uint32_t)c < (uint32_t)cols;
    if (coded) {
      const uint32_t code = dict.code[dict.find(val[j])];
      if (H.code_bits == 4) H.tcode[(size_t)pos >> 1] |= (uint8_t)(code << ((pos & 1) * 4));   // both nibbles of a byte belong to one group, one thread
      else if (H.code_bits == 16) reinterpret_cast<uint16_t *>(H.tcode.data())[(size_t)pos] = (uint16_t)code;
      else H.tcode[(size_t)pos] = (uint8_t)code;
    }
    else H.tval[(size_t)pos] = val[j];
    H.tcol[(size_t)pos] = in_range ? (uint16_t)(c % TCOLS) : TCOL_IDENTITY;
  };
  parallel_items(n_bins, 8, NT, [&](int64_t bi, int th) {
    const RowBin &b = H.bins[(size_t)bi];
    Scratch &sc = scratch[(size_t)th];
    sort_bin(sc, b);
    sc.row_next.assign((size_t)b.nr, 0);
    int64_t off = b.pstart;      // next product of the bin (bin-major order: pslot)
    int32_t piece_k = 0;
    for (const Piece &pc : bin_pieces[(size_t)bi]) {
      const int64_t spos = run_start[(size_t)pc.tile] + pc.pos;      // stream position of the piece
      const int64_t ppos = run_pstart[(size_t)pc.tile] + pc.ppos;    // P position of the piece (a multiple of 4)
      {
        // where the piece lies in P, as a group index relative to the piece's place in the bin: a group's
        // P address is ptab[its piece] + its group index inside the bin.  gblk marks the group that starts a piece.
        const int64_t g_in_bin = (off - b.pstart) / 4;
        H.ptab[(size_t)b.pt0 + (size_t)piece_k] = (int32_t)(ppos / 4 - g_in_bin);
        H.ptile[(size_t)b.pt0 + (size_t)piece_k] = (uint16_t)pc.tile;
        uint32_t *rec = &H.gblk[((size_t)b.gb0 + (size_t)(g_in_bin / 64)) * 4];
        rec[(g_in_bin % 64) / 32] |= 1u << (g_in_bin % 32);
      }
      find_runs(sc, sc.start[(size_t)piece_k], sc.start[(size_t)piece_k + 1]);
      // Lay the piece out (see pack_piece): q = next stream position, o = next product of the piece.  Products are
      // numbered in stream order of the groups that store (A groups and singles groups).
      int64_t q = spos, o = 0;
      const PiecePack pk = pack_piece((int64_t)sc.lp.size(), (int64_t)sc.ls.size());
      auto slot_of = [&](int32_t i) -> uint16_t {   // the next free product slot of the row of entry s_*[i]
        const int32_t rl = sc.s_row[(size_t)i];
        return (uint16_t)(light_off((int64_t)b.r0 + rl) - b.csr0 + sc.row_next[(size_t)rl]++);
      };
      size_t s1 = 0;   // singles handed out so far
      auto singles_group = [&]() {   // up to 4 singles; the rest of the group stays padding (identity column, slot TSLOT_PAD)
        for (int k = 0; k < 4; k++)
          if (s1 < sc.ls.size()) {
            const int32_t i = sc.ls[s1++];
            put_entry(q + k, sc.s_j[(size_t)i]);
            H.pslot[(size_t)(off + o + k)] = slot_of(i);
          }
        q += 4; o += 4;
      };
      int32_t sg_left = pk.sgroups;
      if (pk.blocks > 0 && ((q / 4) & 1)) { singles_group(); sg_left--; }   // pair blocks start on even group indices
      for (int32_t blk = 0; blk < pk.blocks; blk++) {
        for (int k = 0; k < 4; k++) {
          const size_t pi = (size_t)blk * 4 + (size_t)k;
          if (pi < sc.lp.size()) {                       // a pair: first entry in A, second in B
            const int32_t i = sc.lp[pi];
            put_entry(q + k, sc.s_j[(size_t)i]);
            put_entry(q + 4 + k, sc.s_j[(size_t)i + 1]);
            H.pslot[(size_t)(off + o + k)] = slot_of(i);
          } else if (s1 < sc.ls.size()) {                // a single in A; B keeps its padding entry
            const int32_t i = sc.ls[s1++];
            put_entry(q + k, sc.s_j[(size_t)i]);
            H.pslot[(size_t)(off + o + k)] = slot_of(i);
          }
        }
        H.tcol[(size_t)q + 4] |= TCOL_FOLD;              // group B folds into the lane in front of it
        q += 8; o += 4;
      }
      while (sg_left-- > 0) singles_group();
      if (s1 != sc.ls.size()) ok = false;
      // (the layout above must agree with the sizes fixed in the first pass)
      if (q - spos != 4ll * pc.groups || o != pc.prods) ok = false;
      off += pc.prods;
      piece_k++;
    }
    {
      const int64_t nblk = std::max<int64_t>(1, (b.n / 4 + 63) / 64);
      uint32_t before = 0;
      for (int64_t j = 0; j < nblk; j++) {
        uint32_t *rec = &H.gblk[((size_t)b.gb0 + (size_t)j) * 4];
        rec[2] = before;
        before += (uint32_t)(__builtin_popcount(rec[0]) + __builtin_popcount(rec[1]));
      }
    }
  });
  lap("allocs + P4 light fill");
  parallel_items(n_heavy, 1, NT, [&](int64_t hi, int th) {
    const LongRow &lr = H.heavy[(size_t)hi];
    const int64_t r = lr.row;
    Scratch &sc = scratch[(size_t)th];
    for (const Piece &pc : heavy_pieces[(size_t)hi]) {
      const int32_t padded = (pc.cnt + HSTRIP - 1) / HSTRIP * HSTRIP;
      const int64_t spos = heavy_start[(size_t)pc.tile] + pc.pos;
      int32_t part = lr.slot0 + pc.part0;
      int32_t part_q0 = 0;                                               // where (in q) the current partial starts
      for (int32_t q = 0; q < padded; q += HSTRIP) {
        const int64_t krel = (pc.pos + q) / HSTRIP;                      // strip index inside the tile's heavy run
        if (q > 0 && krel % 64 == 0) { part++; part_q0 = q; }            // next wave of phase 1
        const bool last = q + HSTRIP >= padded || (krel + 1) % 64 == 0;
        H.gdest[(size_t)(spos + q - heavy_base) / HSTRIP] =
            (uint32_t)part | ((uint32_t)((q - part_q0) / HSTRIP) << GD_DIST_SHIFT) | (last ? GD_LAST : 0u);
      }
      sc.pos[(size_t)pc.tile] = spos;
    }
    for (int32_t j = rp[r]; j < rp[r + 1]; j++)
      put_entry(sc.pos[(size_t)tile_of(ci[j])]++, j);
  });
  lap("P5 heavy fill");
  // P6: obase[block] = P position of the first product of the block's 64 stream groups (a group stores 4 products
  // unless its first entry carries the fold flag).  Read back from the flags just written, so the two cannot disagree.
  std::vector<int64_t> ob0((size_t)CT + 1, 0);
  for (int t = 0; t < CT; t++) ob0[(size_t)t + 1] = ob0[(size_t)t] + (((run_len[(size_t)t] + 255) & ~int64_t(255)) / 256);
  H.obase.assign((size_t)ob0[(size_t)CT] + 1, 0u);
  H.obase[(size_t)ob0[(size_t)CT]] = (uint32_t)H.p_len;   // one entry behind the last block: obase[b + 1] - obase[b] = products of block b
  parallel_items(CT, 1, NT, [&](int64_t t, int) {
    int64_t pp = run_pstart[(size_t)t];
    const int64_t s0 = run_start[(size_t)t], s1 = s0 + run_len[(size_t)t];
    for (int64_t q = s0; q < s1; q += 4) {
      if (((q - s0) & 255) == 0) H.obase[(size_t)(ob0[(size_t)t] + (q - s0) / 256)] = (uint32_t)pp;
      if (!(H.tcol[(size_t)q] & TCOL_FOLD)) pp += 4;
      else if (((q - s0) & 255) == 0) ok = false;   // a pair never straddles a block of 64 groups
    }
    if (pp != run_pstart[(size_t)t] + run_plen[(size_t)t]) ok = false;
  });
  lap("P6 obase");
  if (!ok) return false;
  cut_work_items(CT, run_start, run_len, heavy_start, hrel, ob0, opt, n_cus, H);
  lap("work items");
  return true;
}

// ---- the (or,and) semiring on bits (see bits.hip.h) -----------------------
// Entries of row range rr / column block ct, ordered by row (the CSR walk is row-major), every 8192-row sub-range
// padded to a multiple of 8 entries with copies of its last entry; blocks with more than `max_item` entries are cut
// at sub-range boundaries into several work items.
static bool build_bits_plan(int64_t rows, int64_t cols, int64_t nnz, const int32_t *rp, const int32_t *ci, const uint32_t *val,
                            const sh_plan_options &opt, BitsHost &H) {
  H.n_rr = (int32_t)std::max<int64_t>(1, (rows + BITS_BR - 1) / BITS_BR);
  H.n_ct = (int32_t)std::max<int64_t>(1, (cols + BITS_BC - 1) / BITS_BC);
  const int NT = build_threads(opt);
  const int64_t nblk = (int64_t)H.n_rr * H.n_ct, ncell = nblk * BITS_NSUB;
  if (ncell > (int64_t)1 << 27) return false;   // (a 500 M x 500 M matrix: shard it)
  std::vector<int64_t> cnt((size_t)ncell + 1, 0), start((size_t)ncell + 1, 0);
  auto cell_of = [&](int64_t r, int32_t c) { return (((r / BITS_BR) * H.n_ct + c / BITS_BC) * BITS_NSUB) + (r % BITS_BR) / BITS_SUB; };
  auto live = [&](int32_t j) { return val[j] != 0u && (uint32_t)ci[j] < (uint32_t)cols; };   // bool_and(x, a): a != 0; out of range: identity 0
  parallel_items(H.n_rr, 1, NT, [&](int64_t rr, int) {   // a row range owns its cells
    for (int64_t r = rr * BITS_BR; r < std::min<int64_t>(rows, (rr + 1) * BITS_BR); r++)
      for (int32_t j = rp[r]; j < rp[r + 1]; j++)
        if (live(j)) cnt[(size_t)cell_of(r, ci[j])]++;
  });
  if (!bits_starts_and_items(H, cnt, start)) return false;
  const int64_t pos = start[(size_t)ncell];
  H.ent.assign((size_t)pos + 8, 0u);
  parallel_items(H.n_rr, 1, NT, [&](int64_t rr, int) {
    std::vector<int64_t> cur((size_t)H.n_ct * BITS_NSUB);
    for (int64_t k = 0; k < (int64_t)cur.size(); k++) cur[(size_t)k] = start[(size_t)(rr * H.n_ct * BITS_NSUB + k)];
    for (int64_t r = rr * BITS_BR; r < std::min<int64_t>(rows, (rr + 1) * BITS_BR); r++)
      for (int32_t j = rp[r]; j < rp[r + 1]; j++)
        if (live(j)) {
          const int64_t cell = cell_of(r, ci[j]) - rr * H.n_ct * BITS_NSUB;
          H.ent[(size_t)cur[(size_t)cell]++] = (uint32_t)(ci[j] % BITS_BC) | ((uint32_t)(r % BITS_SUB) << 19);
        }
    for (int64_t k = 0; k < (int64_t)cur.size(); k++) {   // pad with copies of the last entry
      const int64_t cell = rr * H.n_ct * BITS_NSUB + k, end = start[(size_t)cell] + ((cnt[(size_t)cell] + 7) & ~int64_t(7));
      for (int64_t q = cur[(size_t)k]; q < end; q++) H.ent[(size_t)q] = H.ent[(size_t)cur[(size_t)k] - 1];
    }
  });
  H.ent_len = (int64_t)H.ent.size();
  (void)nnz;
  return true;
}
