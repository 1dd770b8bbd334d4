// fqsx_pe.h -- paired-end part of the DNA path (CompressPE and helpers, dna.cpp:880-1136,1559-1638,1757-1880;
// CHT_pair_kmers, ht_kmer.h:559-663 / ht_kmer.cpp).  Included by fqsx_dev.h.
//
// Per pair the PE-specific work is small (4+8 minimizer windows, 8 pair-table look-ups, <= 14 inserts); it is
// written wave-uniform with lane-parallel selection steps.  The per-base coding of both mates reuses the
// staged suffix coder (with a start position and the reversed-context variant for the left part of an
// anchored second mate).
#pragma once

// ---- direct-strand b-mer windows (find_minimizer / find_maximizer / generate_read_bmers)
FQ_DEV bool pe_valid_minimizer(const DevCfg *cfg, u64 x) { u64 f = x >> (2 * cfg->bmer - 6); return f != 0 && f != 1; }        // dna.cpp:879-889
FQ_DEV bool pe_valid_maximizer(const DevCfg *cfg, u64 x) { u64 f = x >> (2 * cfg->bmer - 6); return f != 0x3e && f != 0x3f; }  // dna.cpp:892-902
FQ_DEV u64 pe_value_mask(const DevCfg *cfg) { return (1ull << (2 * cfg->bmer)) - 1ull; }

// find_minimizer over codes[off .. off+size), dna.cpp:999-1023
FQ_DEV u64 pe_find_minimizer(const DevCfg *cfg, const u8 *codes, int off, int size) {
  const u32 k = cfg->bmer;
  const u64 vm = pe_value_mask(cfg);
  u64 best = vm, v = 0;
  u32 cur = 0;
  for (int i = 0; i < size; ++i) {
    u32 sym = codes[off + i];
    if (sym == 4) { v = 0; cur = 0; }
    else {
      v = ((v << 2) | sym) & vm;
      if (cur < k) ++cur;
      if (cur == k && v < best && pe_valid_minimizer(cfg, v)) best = v;
    }
  }
  return best;
}
// The same for a read staged in LDS (4-byte aligned, encoder): one window position per lane
FQ_DEV u64 pe_find_minimizer_w(const DevCfg *cfg, const u8 *codes, int off, int size) {
  const u32 k = cfg->bmer;
  const u64 vm = pe_value_mask(cfg);
#if FQ_WAVE > 1
  if (k <= 28) {
    // one window position per lane: the b-mer ending there comes out of the 28 bases before it, read as eight aligned
    // words squeezed to 2 bits per base (as in speculate); the minimum over the lanes is the minimizer
    u64 best = vm;
    for (int base = (int)k - 1; base < size; base += (int)FQ_WAVE) {
      const int i = base + (int)FQ_LANE;
      if (i < size) {
        const u32 e1 = (u32)(off + i + 1);                       // position after the b-mer's last base
        const u32 start = e1 >= 28 ? e1 - 28 : 0, a = start & ~3u, sh = a + 32 - e1;
        const u32 *wp = (const u32 *)(codes + a);
        u64 P = 0;
        u32 N32 = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
          const u32 x = wp[q];
          P = (P << 8) | (((x & 0x03030303u) * 0x40100401u) >> 24);
          N32 = (N32 << 4) | (((((x >> 2) & 0x01010101u) * 0x08040201u) >> 24) & 15u);
        }
        const u64 v = (P >> (2 * sh)) & vm;
        const bool clean = ((N32 >> sh) & ((1u << k) - 1u)) == 0;   // no N among the k bases (the serial scan restarts at an N)
        if (clean && v < best && pe_valid_minimizer(cfg, v)) best = v;
      }
    }
    return wave_min64(best);
  }
#endif
  return pe_find_minimizer(cfg, codes, off, size);
}
// find_maximizer (walks the window backwards), dna.cpp:1026-1050
FQ_DEV u64 pe_find_maximizer(const DevCfg *cfg, const u8 *codes, int off, int size) {
  const u32 k = cfg->bmer;
  const u64 vm = pe_value_mask(cfg);
  u64 best = 0, v = 0;
  u32 cur = 0;
  for (int i = size - 1; i >= 0; --i) {
    u32 sym = codes[off + i];
    if (sym == 4) { v = 0; cur = 0; }
    else {
      v = ((v << 2) | sym) & vm;
      if (cur < k) ++cur;
      if (cur == k && v > best && pe_valid_maximizer(cfg, v)) best = v;
    }
  }
  return best;
}

// ---- pair tables
FQ_DEV u32 pe_owner(const DevCfg *cfg, u64 h) { return (u32)((h >> 48) % cfg->T); }  // get_part_id_from_hash, ht_kmer.h:599-602

// CHT_pair_kmers::find (ht_kmer.cpp:211-228): append every (value|count) stored under `key` to the LDS candidate
// list; when the list fills up it is cut to its best 48 entries (exact: top-48 of a union is the top-48 of the
// parts' top-48, and merge_minim_results keeps only those when more than 48 exist)
FQ_DEV void pe_reduce48(Wk &w, u32 &n);
FQ_DEV void ptab_find(Wk &w, const PTab &t, u32 sub, u64 key, u32 &n) {
  const u64 h = murmur64(key);
  const u64 *tk = t.key + (u64)sub * t.stride, *tv = t.val + (u64)sub * t.stride;
  u64 p = h & t.cap_mask;
#if FQ_WAVE > 1
  // 64 slots of the cluster per round trip: the matches before the first empty slot are appended in slot order
  for (u64 base = 0; base <= t.cap_mask; base += FQ_WAVE) {
    const u64 it = base + FQ_LANE;
    const bool in = it <= t.cap_mask;
    const u64 q = (p + it) & t.cap_mask;
    const u64 k = in ? tk[q] : 0, v = in ? tv[q] : 0;
    const u64 em = wave_ballot(in && k == 0 && v == 0);
    const u32 lim = em ? ctz64(em) : FQ_WAVE;
    const bool hit = in && FQ_LANE < lim && k == key;
    const u64 hm = wave_ballot(hit);
    const u32 cnt = popc64(hm);
    if (cnt) {
      if (n + cnt > 512) pe_reduce48(w, n);   // (exact whenever it happens: see above)
      FQ_SYNC();
      if (hit) w.sm->pe_cand[n + popc64(hm & ((1ull << FQ_LANE) - 1ull))] = v;
      FQ_SYNC();
      n += cnt;
    }
    if (em) break;
  }
  return;
#endif
  for (u64 it = 0; it <= t.cap_mask; ++it) {
    u64 k = tk[p], v = tv[p];
    if (k == 0 && v == 0) break;
    if (k == key) {
      if (n == 512) pe_reduce48(w, n);
      FQ_SYNC();
      if (FQ_LANE == 0) w.sm->pe_cand[n] = v;
      FQ_SYNC();
      ++n;
    }
    p = (p + 1) & t.cap_mask;
  }
}
// CHT_pair_kmers::insert (ht_kmer.cpp:124-187), wave-uniform (used for the worker-private table)
FQ_DEV void ptab_insert_uniform(Wk &w, const PTab &t, u32 sub, u64 key, u64 value, u64 count) {
  const DevCfg *cfg = w.cfg;
  const u64 vm = pe_value_mask(cfg), maxc = (~0ull) >> (2 * cfg->bmer);
  const u32 cs = 2 * cfg->bmer;
  if (key == vm || value == vm) return;
  u64 *tk = t.key + (u64)sub * t.stride, *tv = t.val + (u64)sub * t.stride;
  u64 p = murmur64(key) & t.cap_mask;
#if FQ_WAVE > 1
  // 64 slots per round trip: the first slot that is empty or holds (key, value) is the one the serial walk stops at
  for (u64 base = 0; base <= t.cap_mask; base += FQ_WAVE) {
    const u64 it = base + FQ_LANE;
    const bool in = it <= t.cap_mask;
    const u64 q = (p + it) & t.cap_mask;
    const u64 k = in ? tk[q] : 0, v = in ? tv[q] : 0;
    const bool empty = in && k == 0 && v == 0, match = in && k == key && (v & vm) == value;
    const u64 stop_m = wave_ballot(empty || match);
    if (!stop_m) continue;
    const u32 at = ctz64(stop_m);
    const bool is_empty = (wave_ballot(empty) >> at) & 1ull;
    if (is_empty) {
      const u32 f = t.filled[sub];
      if ((u64)f * 10 >= (t.cap_mask + 1) * 9) { w.err = FQSX_ERR_PE_FULL; return; }
      if (count > maxc) count = maxc;
      if (FQ_LANE == at) { tk[q] = key; tv[q] = value + (count << cs); }
      if (FQ_LANE == 0) t.filled[sub] = f + 1;
    } else if (FQ_LANE == at) {
      const u64 cur = v >> cs;
      tv[q] = cur + count < maxc ? v + (count << cs) : v + ((maxc - cur) << cs);
    }
    FQ_SYNC_MEM();
    return;
  }
  w.err = FQSX_ERR_PE_FULL;
  return;
#endif
  for (u64 it = 0; it <= t.cap_mask; ++it) {
    u64 k = tk[p], v = tv[p];
    if (k == 0 && v == 0) {
      u32 f = t.filled[sub];
      if ((u64)f * 10 >= (t.cap_mask + 1) * 9) { w.err = FQSX_ERR_PE_FULL; return; }
      if (count > maxc) count = maxc;
      tk[p] = key;
      tv[p] = value + (count << cs);
      t.filled[sub] = f + 1;
      return;
    }
    if (k == key && (v & vm) == value) {
      u64 cur = v >> cs;
      tv[p] = cur + count < maxc ? v + (count << cs) : v + ((maxc - cur) << cs);
      return;
    }
    p = (p + 1) & t.cap_mask;
  }
  w.err = FQSX_ERR_PE_FULL;
}

// ---- candidate ranking (merge_minim_results, dna.cpp:905-971)
// order A: counter descending, then value ascending; order B: value ascending.  rank = number of entries that
// precede (ties by index); entries with rank < keep are written to dst[rank].
FQ_DEV void pe_rank_select(Wk &w, const u64 *src, u32 n, u64 *dst, u32 keep, bool by_count) {
  const u32 cs = 2 * w.cfg->bmer;
  const u64 vm = pe_value_mask(w.cfg);
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < n; i += FQ_WAVE) {
    const u64 x = src[i], xc = x >> cs, xv = x & vm;
    u32 rank = 0;
    for (u32 j = 0; j < n; ++j) {
      const u64 y = src[j], yc = y >> cs, yv = y & vm;
      bool before;
      if (by_count) before = yc != xc ? yc > xc : yv != xv ? yv < xv : j < i;
      else before = yv != xv ? yv < xv : j < i;
      rank += before ? 1u : 0u;
    }
    if (rank < keep) dst[rank] = x;
  }
  FQ_SYNC();
}
FQ_DEV void pe_reduce48(Wk &w, u32 &n) {
  WgShared *sm = w.sm;
  pe_rank_select(w, sm->pe_cand, n, sm->pe_top, 48, true);
  for (u32 i = FQ_LANE; i < 48; i += FQ_WAVE) sm->pe_cand[i] = sm->pe_top[i];
  FQ_SYNC();
  n = 48;
}
// leaves the best (<= 16) merged candidates in sm->pe_top, in order; returns their number
FQ_DEV u32 pe_merge_candidates(Wk &w, u32 n) {
  WgShared *sm = w.sm;
  const u32 cs = 2 * w.cfg->bmer;
  const u64 vm = pe_value_mask(w.cfg), maxc = (~0ull) >> cs;
  if (n == 1) {
    FQ_SYNC();
    if (FQ_LANE == 0) sm->pe_top[0] = sm->pe_cand[0];
    FQ_SYNC();
    return 1;
  }
  if (n > 48) pe_reduce48(w, n);
  pe_rank_select(w, sm->pe_cand, n, sm->pe_top, n, false);  // sort by value
  // merge equal values, summing the counters with saturation (dna.cpp:937-950); result back into pe_cand
  u32 m = 0;
  u64 cur = sm->pe_top[0];
  for (u32 i = 1; i < n; ++i) {
    u64 x = sm->pe_top[i];
    if ((cur & vm) != (x & vm)) {
      FQ_SYNC();
      if (FQ_LANE == 0) sm->pe_cand[m] = cur;
      ++m;
      cur = x;
    } else {
      u64 cx = cur >> cs, cy = x >> cs;
      if (cx + cy > maxc) cy = maxc - cx;
      cur += cy << cs;
    }
  }
  FQ_SYNC();
  if (FQ_LANE == 0) sm->pe_cand[m] = cur;
  ++m;
  FQ_SYNC();
  pe_rank_select(w, sm->pe_cand, m, sm->pe_top, 16, true);  // only the first 15 can be selected (dna.cpp:1827-1828)
  return m < 16 ? m : 16;
}

FQ_DEV void pe_push(Wk &w, u64 key, u64 value, u64 weight) {  // my_pe_mers_to_add + ht_pe_mers_local->insert
  const DevCfg *cfg = w.cfg;
  if (w.pe_n >= cfg->pe_cap) { w.err = FQSX_ERR_PE_FULL; return; }
  u64 *d = cfg->pe_list + ((u64)w.tid * cfg->pe_cap + w.pe_n) * 3;
  d[0] = key; d[1] = value; d[2] = weight;
  ++w.pe_n;
  ptab_insert_uniform(w, cfg->l_pe, w.tid, key, value, weight);
}

FQ_DEV void pe_seed_kmers(Wk &w, const u8 *codes, u32 from, u32 to) {  // dna.cpp:1577-1592,1616-1631
  const DevCfg *cfg = w.cfg;
  km_reset(w.pm); km_reset(w.sm_); km_reset(w.bm);
  km_reset(w.pm_u); km_reset(w.sm_u); km_reset(w.bm_u);
  w.cor_pos = 0;
  w.N_run = 0;
  for (u32 i = from; i < to; ++i) {
    u32 c = codes[i];
    if (c == 4) c = 0;
    insert_all(w, c);
  }
}

// CompressPE, dna.cpp:1790-1880.  The pair is coded in three steps -- first mate; second mate directly, or right of
// its anchor; the part left of the anchor on the reverse complement -- that share ONE compress_suffix call site (the
// loop below), so the kernel holds the suffix machinery once, not once per way of calling it.
FQ_DEV void compress_pair(Wk &w, const u8 *p1, u32 size1, const u8 *p2, u32 size2, const u8 *prev, u32 prev_size) {
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  const int k = (int)cfg->bmer;
  const u64 vm = pe_value_mask(cfg);
  // Code lines of the two mates: LDS (w.rdp as staged by the read head, sm->r2c) up to FQSX_RD_LDS bases, the worker's
  // scratch lines in HBM beyond (the reference takes reads of up to 2^24 bases, meta.cpp:69; the suffix machinery itself
  // reads a long sequence from its ASCII source, rd_sym)
  const bool long1 = size1 > FQSX_RD_LDS, long2 = size2 > FQSX_RD_LDS;
  u8 *scr = cfg->pe_scr ? cfg->pe_scr + (u64)w.tid * 3 * cfg->pe_scr_cap : nullptr;
  if ((long1 || long2) && (!scr || size1 + 64 > cfg->pe_scr_cap || size2 + 64 > cfg->pe_scr_cap)) { w.err = FQSX_ERR_PE_READ_TOO_LONG; return; }
  u8 *c2 = long2 ? scr + cfg->pe_scr_cap : sm->r2c;
  u64 m1[4] = {0, 0, 0, 0}, a1[3] = {0, 0, 0}, x1 = 0, a2[3] = {0, 0, 0}, x2 = 0;
  u32 mpos = 0;
  bool anchored = false;
  for (u32 step = 0; step < 3 && !w.err; ++step) {
    // what this step's compress_suffix call codes (if any)
    bool run = false, j_orig = true, j_rev = false, add_hist = false;
    const u8 *j_p = p2;
    u32 j_size = size2, j_start = 0, j_hist0 = 0, hist[4] = {0, 0, 0, 0};
    if (step == 0) {
      // first mate (CompressDirect / CompressSorted with the duplicate flag, dna.cpp:1795-1799)
      TM_BEGIN(t_h1);
      const bool same = read_head(w, p1, size1, prev, prev_size, true, hist);
      TM_END_PE(w, TM_READ_HEAD, t_h1);
      if (!same) { run = true; j_p = p1; j_size = size1; j_orig = w.mode == 2; add_hist = true; }
    } else if (step == 1) {
      // (a duplicate first mate returns early from the coder but sm->rd was staged before that)
      // minimizers of the first mate: 4 windows for the look-up (dna.cpp:1761-1769), 3 + 1 for the inserts (:1055-1083)
      TM_BEGIN(t_mz);
      if (long1) {
        FQ_SYNC_MEM();
        for (u32 i = FQ_LANE; i < size1; i += FQ_WAVE) scr[i] = (u8)dna_code(p1[i]);
        FQ_SYNC_MEM();
      }
      const u8 *c1 = long1 ? scr : w.rdp;
      {
        int mss = (int)size1 - k + 1, s1 = mss / 4, s2 = 2 * mss / 4, s3 = 3 * mss / 4;
        m1[0] = pe_find_minimizer_w(cfg, c1, 0, s1 + k - 1);
        m1[1] = pe_find_minimizer_w(cfg, c1, s1, s2 - s1 + k - 1);
        m1[2] = pe_find_minimizer_w(cfg, c1, s2, s3 - s2 + k - 1);
        m1[3] = pe_find_minimizer_w(cfg, c1, s3, (int)size1 - s3);
        int a = mss / 3, b = 2 * mss / 3;
        a1[0] = pe_find_minimizer_w(cfg, c1, 0, a + k - 1);
        a1[1] = pe_find_minimizer_w(cfg, c1, a, b - a + k - 1);
        a1[2] = pe_find_minimizer_w(cfg, c1, b, (int)size1 - b);
        int mid1 = ((int)size1 + k) / 2;
        x1 = (~pe_find_maximizer(cfg, c1, mid1 - k + 1, (int)size1 - (mid1 - k + 1))) & vm;
      }
      // second mate's codes
      FQ_SYNC();
      FQ_SYNC_MEM();
      for (u32 i = FQ_LANE; i < size2; i += FQ_WAVE) c2[i] = (u8)dna_code(p2[i]);
      FQ_SYNC_MEM();
      FQ_SYNC();
      {
        int mss = (int)size2 - k + 1, a = mss / 3, b = 2 * mss / 3;
        a2[0] = pe_find_minimizer_w(cfg, c2, 0, a + k - 1);
        a2[1] = pe_find_minimizer_w(cfg, c2, a, b - a + k - 1);
        a2[2] = pe_find_minimizer_w(cfg, c2, b, (int)size2 - b);
        int mid2 = ((int)size2 + k) / 2;
        x2 = (~pe_find_minimizer_w(cfg, c2, mid2 - k + 1, (int)size2 - (mid2 - k + 1))) & vm;  // sic: minimizer (dna.cpp:1087)
      }
      TM_END_PE(w, TM_LQ, t_mz);
      TM_BEGIN(t_pf);
      // find_minim_cand: global then local table, 4 minimizers each (dna.cpp:1771-1779)
      u32 nc = 0;
      for (u32 i = 0; i < 4; ++i) ptab_find(w, cfg->g_pe, pe_owner(cfg, murmur64(m1[i])), m1[i], nc);
      for (u32 i = 0; i < 4; ++i) ptab_find(w, cfg->l_pe, w.tid, m1[i], nc);
      TM_END_PE(w, TM_SP_ROLL, t_pf);
      TM_BEGIN(t_mg);
      int mid = -1;
      if (nc) {
        const u32 ntop = pe_merge_candidates(w, nc);
        // first listed candidate occurring in the second mate, and its first position (dna.cpp:1806-1819,974-996)
        u32 best_c = 0xffffffffu, best_pos = 0;
        for (u32 base = 0; base < size2; base += FQ_WAVE) {
          const u32 e = base + FQ_LANE;  // b-mer ending at position e
          u64 v = 0;
          bool ok = e < size2 && e + 1 >= (u32)k;
          if (ok)
            for (int t = 0; t < k; ++t) {
              u32 c = c2[e + 1 - k + t];
              if (c == 4) ok = false;
              v = (v << 2) | (c & 3);
            }
          ok = ok && pe_valid_minimizer(cfg, v);
          for (u32 c = 0; c < ntop && c < best_c; ++c) {
            u64 hit = wave_ballot(ok && v == (sm->pe_top[c] & vm));
            if (hit) {
              best_c = c;
#if FQ_WAVE > 1
              best_pos = base + ctz64(hit) + 1 - (u32)k;
#else
              best_pos = e + 1 - (u32)k;
#endif
            }
          }
        }
        mid = best_c == 0xffffffffu || best_c > 14 ? 15 : (int)best_c;
        mpos = best_pos;
      }
      TM_END_PE(w, TM_SP_PROBE, t_mg);
      TM_BEGIN(t_rest);
      u16 *sb = small_base(w);
      if (mid >= 0) sm_encode(w, sb + SM_OFF_MID, SM_NIB_N, 1u << 15, (u32)mid);  // ctx_rc_pe_minimizer_id, dna.cpp:1835
      if (mid < 0 || mid == 15) {
        // second mate coded directly: CompressDirect(..., false) -- direct prefix, no duplicate flag (dna.cpp:1836)
        (void)read_head(w, p2, size2, nullptr, 0, false, hist);
        run = true; add_hist = true;
        TM_END_PE(w, TM_SP_MISS, t_rest);
      } else {
        anchored = true;
        // position of the anchor (dna.cpp:1840-1868); models keyed by id (+0x100..0x500 for the escape bytes)
        u8 *bi = cfg->byte_init + (u64)w.tid * SM_LAZY_ENTRIES + SM_BYTE_ENTRIES;
        u16 *mp = sb + SM_OFF_MPOS;
#define MPOS_ENC(cls, sym) sm_encode256(w, mp + (u64)((cls) * 16 + mid) * (SM_BYTE_N + 1), bi + ((cls) * 16 + mid), (sym))
        if (mpos < 254) MPOS_ENC(0, mpos);
        else if (mpos < 65536) { MPOS_ENC(0, 254); MPOS_ENC(1, mpos >> 8); MPOS_ENC(2, mpos & 0xff); }
        else { MPOS_ENC(0, 255); MPOS_ENC(3, mpos >> 16); MPOS_ENC(4, (mpos >> 8) & 0xff); MPOS_ENC(5, mpos & 0xff); }
#undef MPOS_ENC
        // CompressDirectWithMinim (dna.cpp:1559-1638): right part forwards from the anchor ...
        if (!long2) {
          FQ_SYNC();
          for (u32 i = FQ_LANE; i < size2; i += FQ_WAVE) w.rdp[i] = c2[i];
          FQ_SYNC();
        }
        pe_seed_kmers(w, long2 ? c2 : w.rdp, mpos, mpos + (u32)k);
        run = true; j_start = (u32)k + mpos; j_hist0 = mpos;
        TM_END_PE(w, TM_SP_HIT, t_rest);
      }
    } else if (anchored) {
      // ... then the left part on the reverse complement, anchored at the same b-mer
      const u32 rsz = mpos + (u32)k;
      if (rsz <= FQSX_RD_LDS) {
        FQ_SYNC();
        FQ_SYNC_MEM();
        for (u32 i = FQ_LANE; i < rsz; i += FQ_WAVE) {
          u32 c = c2[rsz - 1 - i];
          w.rdp[i] = (u8)(c == 4 ? 4 : 3 - c);
        }
        FQ_SYNC();
        pe_seed_kmers(w, w.rdp, 0, (u32)k);
      } else {
        // the reverse complement as a sequence of its own in the third scratch line (ASCII: what the suffix machinery reads
        // of a long sequence); this wave has written it, so the call goes without the scout waves
        u8 *rline = scr + 2 * cfg->pe_scr_cap;
        FQ_SYNC_MEM();
        for (u32 i = FQ_LANE; i < rsz; i += FQ_WAVE) {
          u32 c = c2[rsz - 1 - i];
          rline[i] = c == 4 ? (u8)'N' : (u8)"TGCA"[c];
        }
        FQ_SYNC_MEM();
        for (u32 i = FQ_LANE; i < (u32)k; i += FQ_WAVE) { u32 c = c2[rsz - 1 - i]; scr[i] = (u8)(c == 4 ? 4 : 3 - c); }   // (mate 1's line is free by now)
        FQ_SYNC_MEM();
        pe_seed_kmers(w, scr, 0, (u32)k);
        j_p = rline;
        w.rq_noscout = true;
      }
      run = true; j_size = rsz; j_start = (u32)k; j_rev = true;
    }
    if (run) suffix(w, j_p, j_size, j_orig, j_start, j_rev, j_hist0);
    if (w.err) return;
    if (add_hist) {   // update_s_letters of a directly coded read (dna.cpp:1552-1553,1750-1751)
      add_s_letters(w, hist);
      w.st[ST_BASES] += j_size;
    } else if (step == 2 && anchored) {
      // update_s_letters(p2), dna.cpp:1635
      u32 h0 = 0, h1 = 0, h2 = 0, h3 = 0;
      for (u32 i = FQ_LANE; i < size2; i += FQ_WAVE) {
        u32 c = c2[i];
        h0 += c == 0; h1 += c == 1; h2 += c == 2; h3 += c == 3;
      }
      h0 = wave_sum32(h0); h1 = wave_sum32(h1); h2 = wave_sum32(h2); h3 = wave_sum32(h3);
      w.s_let[0] += h0 + h3; w.s_let[3] += h0 + h3;
      w.s_let[1] += h1 + h2; w.s_let[2] += h1 + h2;
      w.st[ST_BASES] += size2;
    }
  }
  if (w.err) return;
  // append_pe_mers3, dna.cpp:1090-1135
  pe_push(w, a1[0], a2[0], 2); pe_push(w, a1[0], a2[2], 4); pe_push(w, a1[0], x1, 1);
  pe_push(w, a1[1], a2[0], 3); pe_push(w, a1[1], a2[2], 3);
  pe_push(w, a1[2], a2[0], 4); pe_push(w, a1[2], a2[2], 2);
  pe_push(w, a2[0], a1[0], 2); pe_push(w, a2[0], a1[2], 4); pe_push(w, a2[0], x2, 1);
  pe_push(w, a2[1], a1[0], 3); pe_push(w, a2[1], a1[2], 4);
  pe_push(w, a2[2], a1[0], 4); pe_push(w, a2[2], a1[2], 2);
}

// ---- insert phase: owner `tid` scans every source's triples and applies those it owns.  Inserts commute
// (count = min(sum, max)), so the scan order is irrelevant; a batch of <= 64 owned triples is applied
// lane-parallel unless two of them touch the same slot.
template <class SM>
FQ_DEV void pe_apply_batch(const DevCfg &cfg, SM *sm, u32 tid, u32 n, u32 &err) {
  const PTab &t = cfg.g_pe;
  const u32 cs = 2 * cfg.bmer;
  const u64 vm = (1ull << cs) - 1ull, maxc = (~0ull) >> cs;
  u64 *tk = t.key + (u64)tid * t.stride, *tv = t.val + (u64)tid * t.stride;
  const u32 lane = FQ_LANE;
  const bool act = lane < n;
  u64 key = 0, value = 0, cnt = 0, pos = ~0ull, oldv = 0;
  bool found = false;
  if (act) {
    key = sm->pe_bk[0][lane]; value = sm->pe_bk[1][lane]; cnt = sm->pe_bk[2][lane];
    u64 p = murmur64(key) & t.cap_mask;
    for (u64 it = 0; it <= t.cap_mask; ++it) {
      u64 k = tk[p], v = tv[p];
      if (k == 0 && v == 0) break;
      if (k == key && (v & vm) == value) { found = true; oldv = v; break; }
      p = (p + 1) & t.cap_mask;
    }
    pos = p;
  }
  FQ_SYNC();
  sm->bk_key[lane] = pos;
  FQ_SYNC();
  bool clash = false;
#if FQ_WAVE > 1
  if (act)
    for (u32 j = 0; j < n; ++j)
      if (j != lane && sm->bk_key[j] == pos) clash = true;
#endif
  const u32 filled = t.filled[tid];
  const u32 n_new = popc64(wave_ballot(act && !found));
  if ((u64)(filled + n_new) * 10 >= (t.cap_mask + 1) * 9) { err = FQSX_ERR_PE_FULL; return; }
  if (!wave_any(clash)) {
    if (act) {
      if (!found) {
        tk[pos] = key;
        tv[pos] = value + ((cnt > maxc ? maxc : cnt) << cs);
      } else {
        u64 cur = oldv >> cs;
        tv[pos] = cur + cnt < maxc ? oldv + (cnt << cs) : oldv + ((maxc - cur) << cs);
      }
    }
    if (lane == 0) t.filled[tid] = filled + n_new;
    FQ_SYNC_MEM();
    return;
  }
  FQ_SYNC_MEM();
  for (u32 j = 0; j < n; ++j) {  // serial fallback
    const u64 kj = sm->pe_bk[0][j], vj = sm->pe_bk[1][j];
    u64 cj = sm->pe_bk[2][j];
    u64 p = murmur64(kj) & t.cap_mask;
    for (u64 it = 0; it <= t.cap_mask; ++it) {
      u64 k = tk[p], v = tv[p];
      if (k == 0 && v == 0) {
        tk[p] = kj;
        tv[p] = vj + ((cj > maxc ? maxc : cj) << cs);
        t.filled[tid] = t.filled[tid] + 1;
        break;
      }
      if (k == kj && (v & vm) == vj) {
        u64 cur = v >> cs;
        tv[p] = cur + cj < maxc ? v + (cj << cs) : v + ((maxc - cur) << cs);
        break;
      }
      p = (p + 1) & t.cap_mask;
    }
  }
  FQ_SYNC_MEM();
}

// ---- the same insert phase in four steps that spread the scan over the chip (paired-end encoding on one GPU): every triple is
// looked at by ONE thread, which finds its owner (count -> offsets -> scatter into per-owner groups); owner `tid` then applies
// its own group only.  With T owners each scanning all T sources the scan was T times the work and most of the phase's time.
// Grouping is by atomic cursors, so the order inside a group varies from run to run -- the inserts commute.
FQ_DEV bool pe_triple_owner(const DevCfg &cfg, u32 g, u32 &owner, u64 &key, u64 &value, u64 &cnt) {   // g = source * pe_cap + entry
  const u32 src = g / cfg.pe_cap, e = g - src * cfg.pe_cap;
  if (src >= cfg.T || e >= cfg.pe_n[src]) return false;
  const u64 vm = (1ull << (2 * cfg.bmer)) - 1ull;
  const u64 *t = cfg.pe_list + ((u64)src * cfg.pe_cap + e) * 3;
  key = t[0]; value = t[1]; cnt = t[2];
  if (key == vm || value == vm) return false;   // ht_kmer.cpp:126-127
  owner = pe_owner(&cfg, murmur64(key));
  return true;
}
FQ_DEV void pe_bucket_count_body(const DevCfg &cfg, u32 g) {
  u32 owner = 0; u64 k, v, c;
  if (pe_triple_owner(cfg, g, owner, k, v, c)) atomic_add32(&cfg.pe_bkt_n[owner], 1u);
}
FQ_DEV void pe_bucket_scatter_body(const DevCfg &cfg, u32 g) {
  u32 owner = 0; u64 k, v, c;
  if (!pe_triple_owner(cfg, g, owner, k, v, c)) return;
  const u32 at = cfg.pe_bkt_n[cfg.T + owner] + atomic_add32(&cfg.pe_bkt_cur[owner], 1u);
  u64 *d = cfg.pe_bkt + 3 * (u64)at;
  d[0] = k; d[1] = v; d[2] = c;
}
template <class SM>
FQ_DEV void pe_insert_bucket_body(const DevCfg &cfg, SM *sm, u32 tid) {
  const u32 lo = cfg.pe_bkt_n[cfg.T + tid], hi = cfg.pe_bkt_n[cfg.T + tid + 1];
  u32 err = 0;
  for (u32 base = lo; base < hi && !err; base += FQ_WAVE) {
    const u32 e = base + FQ_LANE, n = hi - base < FQ_WAVE ? hi - base : FQ_WAVE;
    FQ_SYNC();
    if (e < hi) { const u64 *t = cfg.pe_bkt + 3 * (u64)e; sm->pe_bk[0][FQ_LANE] = t[0]; sm->pe_bk[1][FQ_LANE] = t[1]; sm->pe_bk[2][FQ_LANE] = t[2]; }
    FQ_SYNC();
    pe_apply_batch(cfg, sm, tid, n, err);
  }
  if (err) *cfg.err = err;
}

// count_only: number of triples owner `tid` will insert (upper bound of new slots) -> demand[tid]
template <class SM>
FQ_DEV void pe_insert_body(const DevCfg &cfg, SM *sm, u32 tid, bool count_only, u32 *demand) {
  const u32 T = cfg.T;
  const u64 vm = (1ull << (2 * cfg.bmer)) - 1ull;
  u32 pending = 0, total = 0, err = 0;
  for (u32 src = 0; src < T && !err; ++src) {
    const u32 n = cfg.pe_n[src];
    const u64 *list = cfg.pe_list + (u64)src * cfg.pe_cap * 3;
    for (u32 base = 0; base < n && !err; base += FQ_WAVE) {
      const u32 e = base + FQ_LANE;
      u64 key = 0, value = 0, cnt = 0;
      bool mine = false;
      if (e < n) {
        key = list[3 * (u64)e]; value = list[3 * (u64)e + 1]; cnt = list[3 * (u64)e + 2];
        mine = key != vm && value != vm && pe_owner(&cfg, murmur64(key)) == tid;  // ht_kmer.cpp:126-127
      }
      const u64 bm = wave_ballot(mine);
      const u32 cntm = popc64(bm);
      total += cntm;
      if (count_only || !cntm) continue;
      if (pending + cntm > FQ_WAVE) {
        pe_apply_batch(cfg, sm, tid, pending, err);
        pending = 0;
      }
#if FQ_WAVE > 1
      const u32 slot = pending + popc64(bm & ((1ull << FQ_LANE) - 1ull));
#else
      const u32 slot = pending;
#endif
      FQ_SYNC();
      if (mine) { sm->pe_bk[0][slot] = key; sm->pe_bk[1][slot] = value; sm->pe_bk[2][slot] = cnt; }
      FQ_SYNC();
      pending += cntm;
    }
  }
  if (count_only) {
    if (FQ_LANE == 0) demand[tid] = total;
    return;
  }
  if (pending && !err) pe_apply_batch(cfg, sm, tid, pending, err);
  if (err) *cfg.err = err;
}
