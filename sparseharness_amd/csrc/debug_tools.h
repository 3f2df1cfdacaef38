// debug_tools.h -- included by engine.hip under -DSH_PLAN_EMULATE only (the tools build, sparseharness_amd/variants/
// emulate.so; never the product build): the host emulator of the tiled / bit plans, the host-vs-device builder
// comparison and the placement probes (sh_debug_*), used by tests/test_plan_cpu.py, tests/test_builder_gpu.py and tools/.
#pragma once
// Tools / CPU tests only (never in the product build): build the tiled plan on the host and execute BOTH phases
// on the host, entry by entry, through exactly the tables the kernels read (fold flags, obase, gdest, gblk / ptab,
// pslot, lrp) with the kernels' indexing.  A test that compares the result with a plain CSR product thereby
// checks the layout without a GPU.  semiring: 0 = (+,x) float, 2 = (or,and) int32.  stats[0..7]: stream entries,
// light entries, products, bins, chunks, heavy rows, P words never written but read (must be 0), tiles.
namespace {
struct HPlusTimes { using T = float; static T identity() { return 0.0f; } static T mul(T x, T a) { return x * a; } static T add(T a, T b) { return a + b; } };
struct HOrAnd { using T = int32_t; static T identity() { return 0; } static T mul(T x, T a) { return (x != 0) && (a != 0); } static T add(T a, T b) { return (a != 0) || (b != 0); } };
template <class T> T hbits(uint32_t u) { T v; memcpy(&v, &u, 4); return v; }
template <class T> uint32_t tobits(T v) { uint32_t u; memcpy(&u, &v, 4); return u; }

template <class SR>
int emulate(const TiledHost &H, int64_t rows, int64_t cols, const uint32_t *x, uint32_t *y, int64_t *stats) {
  using T = typename SR::T;
  constexpr uint32_t POISON = 0x7FC0DEADu;
  std::vector<uint32_t> P((size_t)H.p_len + 4, POISON), partial((size_t)H.n_partials + 1, POISON), xs((size_t)TCOLS + 4);
  const bool coded = !H.vdict.empty();
  auto value_at = [&](int64_t q) -> uint32_t {
    if (!coded) return H.tval[(size_t)q];
    if (H.code_bits == 4) return H.vdict[(H.tcode[(size_t)q >> 1] >> ((q & 1) * 4)) & 0xFu];
    if (H.code_bits == 16) return H.vdict[reinterpret_cast<const uint16_t *>(H.tcode.data())[(size_t)q]];
    return H.vdict[H.tcode[(size_t)q]];
  };
  // ---- phase 1
  for (const TileChunk &ch : H.chunks) {
    if (ch.s >= ch.e) continue;
    const int64_t c0 = (int64_t)ch.tile * TCOLS;
    for (int i = 0; i < TCOLS; i++) xs[(size_t)i] = (c0 + i < cols) ? x[c0 + i] : tobits<T>(SR::identity());
    xs[(size_t)TCOLS] = tobits<T>(SR::identity());
    auto prod = [&](int64_t q, uint32_t colmask) { return SR::mul(hbits<T>(xs[H.tcol[(size_t)q] & colmask]), hbits<T>(value_at(q))); };
    if (ch.hs > ch.s) {   // light chunk: blocks of 64 groups; a group stores 4 products unless it folds into the one in front
      if (ch.s % 256 || ch.e % 4) return -10;
      const int64_t gs = ch.s / 4, le = ch.e / 4;
      auto flagged = [&](int64_t g) { return g < le && (H.tcol[(size_t)g * 4] & TCOL_FOLD) != 0; };
      for (int64_t blk = 0; gs + blk * 64 < le; blk++) {
        int64_t pos = H.obase[(size_t)ch.ob0 + (size_t)blk];
        for (int64_t g = gs + blk * 64; g < std::min(le, gs + blk * 64 + 64); g++) {
          const int64_t lane = g - (gs + blk * 64);
          if (flagged(g)) {
            if (!(lane & 1)) return -12;   // a B group sits on an odd lane, right behind its A
            continue;
          }
          const bool takes = (lane & 1) == 0 && lane + 1 < 64 && flagged(g + 1);
          if (pos + 4 > H.p_len) return -11;
          for (int k = 0; k < 4; k++) {
            T acc = prod(g * 4 + k, k == 0 ? TCOL_MASK : 0xFFFFu);
            if (takes) acc = SR::add(acc, prod((g + 1) * 4 + k, k == 0 ? TCOL_MASK : 0xFFFFu));
            P[(size_t)pos + k] = tobits<T>(acc);
          }
          pos += 4;
        }
      }
    } else {              // heavy chunk: strips of 16, segmented scan inside waves of 64 strips
      const int64_t s0 = ch.s / HSTRIP, s1 = ch.e / HSTRIP, sbase = ch.pdelta / HSTRIP;
      for (int64_t w0 = s0; w0 < s1; w0 += 64) {
        T lane[64];
        uint32_t d[64];
        const int n = (int)std::min<int64_t>(64, s1 - w0);
        for (int l = 0; l < n; l++) {
          T t = SR::identity();
          for (int i = 0; i < HSTRIP; i++) t = SR::add(t, prod((w0 + l) * HSTRIP + i, 0xFFFFu));
          lane[l] = t;
          d[l] = H.gdest[(size_t)(w0 + l - sbase)];
        }
        for (int l = 0; l < n; l++) {
          if (!(d[l] & GD_LAST)) continue;
          const int dist = (int)((d[l] >> GD_DIST_SHIFT) & 63u);
          if (dist > l) return -13;
          T t = lane[l - dist];
          for (int k = l - dist + 1; k <= l; k++) t = SR::add(t, lane[k]);
          partial[d[l] & GD_SLOT_MASK] = tobits<T>(t);
        }
      }
    }
  }
  // ---- phase 2
  int64_t poison_reads = 0;
  std::vector<uint32_t> img((size_t)TBIN);
  for (const RowBin &b : H.bins) {
    std::fill(img.begin(), img.end(), POISON);
    const int64_t n4 = b.n / 4;
    for (int64_t k = 0; k < n4; k++) {
      const uint32_t *rec = &H.gblk[((size_t)b.gb0 + (size_t)(k / 64)) * 4];
      const int lane = (int)(k % 64);
      uint32_t below = 0;
      for (int l = 0; l < lane; l++) below += (rec[l / 32] >> (l % 32)) & 1u;
      const uint32_t own = (rec[lane / 32] >> (lane % 32)) & 1u;
      const int64_t pg = (int64_t)H.ptab[(size_t)b.pt0 + (size_t)std::max<int64_t>((int64_t)(rec[2] + below + own) - 1, 0)] + k;
      if (pg < 0 || pg * 4 + 3 >= (int64_t)P.size()) return -20;
      for (int i = 0; i < 4; i++) {
        const uint16_t sl = H.pslot[(size_t)b.pstart + (size_t)(k * 4 + i)];
        if (sl == TSLOT_PAD) continue;
        if (sl >= TBIN) return -21;
        if (img[sl] != POISON) return -22;   // two products in one slot
        img[sl] = P[(size_t)(pg * 4 + i)];
        if (img[sl] == POISON) poison_reads++;
      }
    }
    for (int64_t r = b.r0; r < (int64_t)b.r0 + b.nr; r++) {
      if (H.lrp[(size_t)r] & 0x80000000u) continue;
      const int64_t s = (int64_t)(H.lrp[(size_t)r] & 0x7FFFFFFFu) - b.csr0, e = (int64_t)(H.lrp[(size_t)r + 1] & 0x7FFFFFFFu) - b.csr0;
      if (s < 0 || e > TBIN) return -23;
      T acc = SR::identity();
      for (int64_t j = s; j < e; j++) {
        if (img[(size_t)j] == POISON) poison_reads++;
        acc = SR::add(acc, hbits<T>(img[(size_t)j]));
      }
      y[r] = tobits<T>(acc);
    }
  }
  for (const LongRow &lr : H.heavy) {
    T acc = SR::identity();
    for (int k = 0; k < lr.nslots; k++) {
      if (partial[(size_t)lr.slot0 + k] == POISON) poison_reads++;
      acc = SR::add(acc, hbits<T>(partial[(size_t)lr.slot0 + k]));
    }
    y[lr.row] = tobits<T>(acc);
  }
  if (stats) {
    stats[0] = H.stream_len; stats[1] = H.light_entries; stats[2] = H.p_len; stats[3] = (int64_t)H.bins.size();
    stats[4] = (int64_t)H.chunks.size(); stats[5] = (int64_t)H.heavy.size(); stats[6] = poison_reads;
    stats[7] = (cols + TCOLS - 1) / TCOLS;
  }
  (void)rows;
  return 0;
}
} // namespace

// The (or,and) bit plan walked on the host: y[r] = OR over the live entries of row r of (x[col] != 0), through the
// entry stream, the sub-range offsets and the item lists exactly as bits_blocks / bits_finish index them.
extern "C" int sh_debug_emulate_bits(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx,
                                     const void *val, const void *x, int32_t *y, int64_t *stats) {
  sh_plan_options opt;
  sh_plan_options_default(&opt);
  BitsHost H;
  if (!build_bits_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, H)) return -1;
  const uint32_t *xv = (const uint32_t *)x;
  std::vector<uint32_t> xbits((size_t)H.n_ct * (BITS_BC / 32), 0u), partial((size_t)std::max<size_t>(H.items.size(), 1) * (BITS_BR / 32), 0u);
  for (int64_t c = 0; c < cols; c++) if (xv[c] != 0u) xbits[(size_t)c >> 5] |= 1u << (c & 31);
  for (size_t k = 0; k < H.items.size(); k++) {
    const BitsItem &it = H.items[k];
    if ((it.s & 7) || (it.e & 7)) return -30;
    const uint32_t *xs = &xbits[(size_t)it.ct * (BITS_BC / 32)];
    uint32_t *os = &partial[k * (BITS_BR / 32)];
    for (int sub = it.sub0; sub < it.sub1; sub++) {
      const int32_t s = H.bsub[(size_t)it.soff + (sub - it.sub0)], e = H.bsub[(size_t)it.soff + (sub - it.sub0) + 1];
      if ((s & 7) || (e & 7) || s < it.s || e > it.e) return -31;
      for (int32_t q = s; q < e; q++) {
        const uint32_t w = H.ent[(size_t)q], c = w & BITS_COL_MASK;
        if ((xs[c >> 5] >> (c & 31)) & 1u) {
          const uint32_t r = (uint32_t)sub * BITS_SUB + (w >> 19);
          os[r >> 5] |= 1u << (r & 31);
        }
      }
    }
  }
  for (int64_t r = 0; r < rows; r++) {
    const int rr = (int)(r / BITS_BR);
    const uint32_t rl = (uint32_t)(r % BITS_BR);
    uint32_t w = 0;
    for (int it = H.rr_item0[(size_t)rr]; it < H.rr_item0[(size_t)rr + 1]; it++) w |= partial[(size_t)it * (BITS_BR / 32) + (rl >> 5)];
    y[r] = (int32_t)((w >> (rl & 31)) & 1u);
  }
  if (stats) { stats[0] = (int64_t)H.ent.size(); stats[1] = H.entries; stats[2] = (int64_t)H.items.size(); stats[3] = H.n_rr; stats[4] = H.n_ct; }
  return 0;
}

extern "C" int sh_debug_emulate_plan(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx,
                                     const void *val, const sh_plan_options *opt_p, int semiring, const void *x, void *y, int64_t *stats) {
  sh_plan_options opt;
  if (opt_p) opt = *opt_p; else sh_plan_options_default(&opt);
  TiledHost H;
  if (!build_tiled_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, 256, H)) return -1;
  if (semiring == 0) return emulate<HPlusTimes>(H, rows, cols, (const uint32_t *)x, (uint32_t *)y, stats);
  if (semiring == 2) return emulate<HOrAnd>(H, rows, cols, (const uint32_t *)x, (uint32_t *)y, stats);
  return -2;
}
// Builds the tiled layout of one matrix twice -- host builder and device builder -- and compares every array.
// Returns the number of arrays (or scalars) that differ, 0 = identical byte for byte; -1: the host builder refused the
// matrix, -2: the device builder refused or failed (report says why), -3: a HIP call of this function failed.
// report (cap bytes) gets one line per difference: the array, the first differing element, both values.
extern "C" int sh_debug_compare_builds(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr, const int32_t *col_idx,
                                       const void *val, const sh_plan_options *opt_p, char *report, int64_t cap) {
  sh_plan_options opt;
  if (opt_p) opt = *opt_p; else sh_plan_options_default(&opt);
  std::string rep;
  auto finish = [&](int rc) { if (report && cap > 0) snprintf(report, (size_t)cap, "%s", rep.c_str()); return rc; };
  if (!e || nnz <= 0) { rep = "bad argument"; return finish(-3); }
  if (hipSetDevice(e->device) != hipSuccess) { rep = "hipSetDevice"; return finish(-3); }
  TiledHost hh, hg;
  TiledDevArrays td;
  struct Guard { TiledDevArrays &t; std::vector<void *> p; ~Guard() { t.release(); for (void *q : p) (void)hipFree(q); } } guard{td, {}};
  if (!build_tiled_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, e->n_cus, hh)) { rep = "host builder refused"; return finish(-1); }
  int32_t *d_rp = nullptr, *d_ci = nullptr;
  uint32_t *d_val = nullptr;
  if (hipMalloc((void **)&d_rp, (size_t)(rows + 1) * 4) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_rp);
  if (hipMalloc((void **)&d_ci, (size_t)nnz * 4 + 32) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_ci);
  if (hipMalloc((void **)&d_val, (size_t)nnz * 4 + 32) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_val);
  if (hipMemcpy(d_rp, row_ptr, (size_t)(rows + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_ci, col_idx, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_val, val, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess) { rep = "hipMemcpy"; return finish(-3); }
  std::string why;
  const int g = build_tiled_plan_gpu(e->stream, rows, cols, nnz, row_ptr, d_rp, d_ci, d_val, opt, e->n_cus, hg, td, why);
  if (g != 1) { rep = "device builder: " + why; return finish(-2); }
  int diffs = 0;
  char line[256];
  auto scalar = [&](const char *name, long long a, long long b) {
    if (a != b) { snprintf(line, sizeof line, "%s: host %lld device %lld\n", name, a, b); rep += line; diffs++; }
  };
  scalar("stream_len", hh.stream_len, hg.stream_len); scalar("p_len", hh.p_len, hg.p_len); scalar("light_len", hh.light_len, hg.light_len);
  scalar("heavy_base", hh.heavy_base, hg.heavy_base); scalar("light_entries", hh.light_entries, hg.light_entries);
  scalar("n_partials", hh.n_partials, hg.n_partials); scalar("code_bits", hh.code_bits, hg.code_bits); scalar("vdict_used", hh.vdict_used, hg.vdict_used);
  scalar("n_bins", (long long)hh.bins.size(), (long long)hg.bins.size()); scalar("n_chunks", (long long)hh.chunks.size(), (long long)hg.chunks.size());
  scalar("n_heavy", (long long)hh.heavy.size(), (long long)hg.heavy.size());
  scalar("tile_fill*1e6", (long long)(hh.tile_fill * 1e6), (long long)(hg.tile_fill * 1e6));
  auto host_bytes = [&](const char *name, const void *a, size_t na, const void *b, size_t nb, size_t elem) {
    if (na != nb) { snprintf(line, sizeof line, "%s: %zu vs %zu bytes\n", name, na, nb); rep += line; diffs++; return; }
    if (na && memcmp(a, b, na) != 0) {
      size_t k = 0;
      while (k < na && ((const uint8_t *)a)[k] == ((const uint8_t *)b)[k]) k++;
      size_t ndiff = 0;
      for (size_t q = 0; q + elem <= na; q += elem) if (memcmp((const uint8_t *)a + q, (const uint8_t *)b + q, elem) != 0) ndiff++;
      const size_t el = k / elem;
      unsigned long long va = 0, vb = 0;
      memcpy(&va, (const uint8_t *)a + el * elem, std::min<size_t>(elem, 8)); memcpy(&vb, (const uint8_t *)b + el * elem, std::min<size_t>(elem, 8));
      snprintf(line, sizeof line, "%s: %zu of %zu elements differ, first at %zu: host 0x%llx device 0x%llx\n", name, ndiff, na / elem, el, va, vb);
      rep += line; diffs++;
    }
  };
  host_bytes("bins", hh.bins.data(), hh.bins.size() * sizeof(RowBin), hg.bins.data(), hg.bins.size() * sizeof(RowBin), sizeof(RowBin));
  host_bytes("chunks", hh.chunks.data(), hh.chunks.size() * sizeof(TileChunk), hg.chunks.data(), hg.chunks.size() * sizeof(TileChunk), sizeof(TileChunk));
  host_bytes("heavy", hh.heavy.data(), hh.heavy.size() * sizeof(LongRow), hg.heavy.data(), hg.heavy.size() * sizeof(LongRow), sizeof(LongRow));
  host_bytes("vdict", hh.vdict.data(), hh.vdict.size() * 4, hg.vdict.data(), hg.vdict.size() * 4, 4);
  bool hip_ok = true;
  auto dev_bytes = [&](const char *name, const void *host, size_t nbytes_host, const void *dev, size_t n_dev, size_t elem) {
    std::vector<uint8_t> tmp(n_dev * elem);
    if (n_dev && hipMemcpy(tmp.data(), dev, n_dev * elem, hipMemcpyDeviceToHost) != hipSuccess) { hip_ok = false; return; }
    host_bytes(name, host, nbytes_host, tmp.data(), n_dev * elem, elem);
  };
  dev_bytes("lrp", hh.lrp.data(), hh.lrp.size() * 4, td.lrp, td.n_lrp, 4);
  dev_bytes("tcol", hh.tcol.data(), hh.tcol.size() * 2, td.tcol, td.n_tcol, 2);
  dev_bytes("tcode", hh.tcode.data(), hh.tcode.size(), td.tcode, td.n_tcode, 1);
  dev_bytes("tval", hh.tval.data(), hh.tval.size() * 4, td.tval, td.n_tval, 4);
  dev_bytes("gdest", hh.gdest.data(), hh.gdest.size() * 4, td.gdest, td.n_gdest, 4);
  dev_bytes("pslot", hh.pslot.data(), hh.pslot.size() * 2, td.pslot, td.n_pslot, 2);
  dev_bytes("gblk", hh.gblk.data(), hh.gblk.size() * 4, td.gblk, td.n_gblk, 4);
  dev_bytes("ptab", hh.ptab.data(), hh.ptab.size() * 4, td.ptab, td.n_ptab, 4);
  dev_bytes("ptile", hh.ptile.data(), hh.ptile.size() * 2, td.ptile, td.n_ptab, 2);
  dev_bytes("obase", hh.obase.data(), hh.obase.size() * 4, td.obase, td.n_obase, 4);
  if (!hip_ok) { rep += "hipMemcpy (download) failed\n"; return finish(-3); }
  return finish(diffs);
}
// The bit-blocked (or,and) layout built by both builders and compared: 0 = identical, else the number of differing
// arrays; -1 host refused, -2 device refused / failed, -3 HIP error here.
extern "C" int sh_debug_compare_bits_builds(sh_engine *e, int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr,
                                            const int32_t *col_idx, const void *val, char *report, int64_t cap) {
  std::string rep;
  auto finish = [&](int rc) { if (report && cap > 0) snprintf(report, (size_t)cap, "%s", rep.c_str()); return rc; };
  if (!e || nnz <= 0) { rep = "bad argument"; return finish(-3); }
  if (hipSetDevice(e->device) != hipSuccess) { rep = "hipSetDevice"; return finish(-3); }
  sh_plan_options opt;
  sh_plan_options_default(&opt);
  BitsHost hh, hg;
  if (!build_bits_plan(rows, cols, nnz, row_ptr, col_idx, (const uint32_t *)val, opt, hh)) { rep = "host builder refused"; return finish(-1); }
  struct Guard { std::vector<void *> p; ~Guard() { for (void *q : p) (void)hipFree(q); } } guard;
  int32_t *d_rp = nullptr, *d_ci = nullptr;
  uint32_t *d_val = nullptr, *d_ent = nullptr;
  if (hipMalloc((void **)&d_rp, (size_t)(rows + 1) * 4) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_rp);
  if (hipMalloc((void **)&d_ci, (size_t)nnz * 4 + 32) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_ci);
  if (hipMalloc((void **)&d_val, (size_t)nnz * 4 + 32) != hipSuccess) { rep = "hipMalloc"; return finish(-3); }
  guard.p.push_back(d_val);
  if (hipMemcpy(d_rp, row_ptr, (size_t)(rows + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_ci, col_idx, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_val, val, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess) { rep = "hipMemcpy"; return finish(-3); }
  std::string why;
  if (build_bits_plan_gpu(e->stream, rows, cols, nnz, d_rp, d_ci, d_val, hg, &d_ent, why) != 1) { rep = "device builder: " + why; return finish(-2); }
  guard.p.push_back(d_ent);
  int diffs = 0;
  char line[256];
  auto same = [&](const char *name, const void *a, size_t na, const void *b, size_t nb) {
    if (na != nb || (na && memcmp(a, b, na) != 0)) {
      size_t k = 0;
      while (k < std::min(na, nb) && ((const uint8_t *)a)[k] == ((const uint8_t *)b)[k]) k++;
      snprintf(line, sizeof line, "%s: %zu vs %zu bytes, first difference at byte %zu\n", name, na, nb, k);
      rep += line; diffs++;
    }
  };
  std::vector<uint32_t> ent((size_t)hg.ent_len);
  if (hipMemcpy(ent.data(), d_ent, ent.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { rep = "hipMemcpy (download)"; return finish(-3); }
  same("ent", hh.ent.data(), hh.ent.size() * 4, ent.data(), ent.size() * 4);
  same("items", hh.items.data(), hh.items.size() * sizeof(BitsItem), hg.items.data(), hg.items.size() * sizeof(BitsItem));
  same("bsub", hh.bsub.data(), hh.bsub.size() * 4, hg.bsub.data(), hg.bsub.size() * 4);
  same("rr_item0", hh.rr_item0.data(), hh.rr_item0.size() * 4, hg.rr_item0.data(), hg.rr_item0.size() * 4);
  if (hh.entries != hg.entries || hh.n_rr != hg.n_rr || hh.n_ct != hg.n_ct || hh.ent_len != hg.ent_len) { rep += "scalars differ\n"; diffs++; }
  return finish(diffs);
}

// Placement experiments (tools/placement_probe.py): move one array of the tiled plan to a fresh allocation (hold != 0:
// the old one is kept allocated -- and leaked until process exit -- so that the new one cannot land in the same place),
// return its new address.  which: 0 P, 1 tcol, 2 tcode / tval, 3 pslot, 4 gblk, 5 ptab, 6 obase, 7 lrp.
// align_log2 > 21: the new place is the first address aligned to 2^align_log2 inside an allocation that much larger (the
// base is leaked: experiments only).
extern "C" int sh_debug_move_array(sh_engine *e, sh_csr *m, int which, int hold, int align_log2, uint64_t *address) {
  if (!e || !m || m->plan != PLAN_TILED) return SH_EINVAL;
  if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return SH_EHIP;
  void **slot = nullptr;
  size_t bytes = 0;
  const bool coded = m->n_vdict != 0;
  switch (which) {
    case 0: slot = (void **)&m->d_P; bytes = (size_t)std::max<int64_t>(m->p_len, 4) * 4 + 16; break;
    case 1: slot = (void **)&m->d_tcol; bytes = (size_t)m->stream_len * 2 + 16; break;
    case 2: if (coded) { slot = (void **)&m->d_tcode; bytes = tcode_bytes(m->code_bits, m->stream_len) + 64; }
            else { slot = (void **)&m->d_tval; bytes = (size_t)m->stream_len * 4 + 16; }
            break;
    case 3: slot = (void **)&m->d_pslot; bytes = (size_t)m->p_len * 2 + 16; break;
    default: return SH_EINVAL;
  }
  void *fresh = nullptr;
  const size_t align = align_log2 > 21 ? (size_t)1 << align_log2 : 0;
  if (hipMalloc(&fresh, bytes + align) != hipSuccess) return SH_ENOMEM;
  if (align) fresh = (void *)(((uintptr_t)fresh + align - 1) & ~(uintptr_t)(align - 1));
  if (hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return SH_EHIP;
  if (!hold) (void)hipFree(*slot);
  *slot = fresh;
  if (address) *address = (uint64_t)(uintptr_t)fresh;
  return SH_OK;
}
// (placement experiments) point the matrix at another product array allocated by sh_debug_move_array(which = 0, hold = 1)
extern "C" int sh_debug_set_P(sh_engine *e, sh_csr *m, uint64_t address) {
  if (!e || !m || m->plan != PLAN_TILED || !address) return SH_EINVAL;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return SH_EHIP;
  m->d_P = (uint32_t *)(uintptr_t)address;
  return SH_OK;
}
// (slab experiments, tools/slab_probe.py) words of the matrix's product array, for a host-side bounds check before matrices share one
extern "C" int64_t sh_debug_p_len(const sh_csr *m) { return m && m->plan == PLAN_TILED ? m->p_len : -1; }
