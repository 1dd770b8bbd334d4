// fqsx_dec.h -- device decoder of the DNA stream (DecompressSE / DecompressPE and helpers,
// dna.cpp:1139-1514,1641-1713,1883-2044; CRangeDecoder sub_rc.h:93-158; Decode rc.h:407-421,490-503).
// Included by fqsx_dev.h.
//
// Decoding has no look-ahead (the next k-mer depends on the symbol being decoded), so there is no stage
// P: every position runs the complete reference logic serially, with the same lane-parallel batches for
// the multi-probe steps as the encoder's general path.  Tables, models, RNG streams, mailboxes and the
// insert phases are shared with the encoder, which is what makes the decoder rebuild identical state.
#pragma once

// find_rc_code_context / find_rc_letters_context (dna.cpp:2107-2286) for the decoder, whose chain per symbol is two or
// three DEPENDENT context-table round trips in find_leveled: here every level's slot is looked up at once, one level per
// lane (read-only: the table cannot change in between, the wave is the worker's only writer), and the level search then
// walks the cached answers.  Same decisions, same stores, same creations / clones as find_leveled.
FQ_DEV u32 dec_level_get(WgShared *sm, u32 l, Slot4 &s, const u64 *lev, u64 rs) {
  const u32 ls = 1;   // (the decoder's level keys are a plain array: LEVKEY)
  const u32 idx = sm->fr_idx[l];
  if (idx == FQSX_NIL) return FQSX_NIL;
  s.q0 = LEVKEY(l); s.q1 = sm->fr_q1[l]; s.q2 = sm->fr_q2[l]; s.q3 = sm->fr_q3[l];
  return idx;
}
FQ_DEV u32 find_leveled_dec(Wk &w, u32 tag, const u64 *lev, u64 rs, int n_levels, double &avg, u64 tpl_q2, u64 tpl_q3, u32 tpl_total, Slot4 &s) {
  WgShared *sm = w.sm;
  const u32 ls = 1;
  FQ_SYNC();
  for (u32 l = FQ_LANE; l < (u32)n_levels; l += FQ_WAVE) {
    u32 vis = 0;
    const RoHit h = ctx_probe_ro(w, tag, LEVKEY(l), vis);
    sm->fr_idx[l] = h.present ? h.idx : FQSX_NIL;
    sm->fr_q1[l] = h.q1; sm->fr_q2[l] = h.q2; sm->fr_q3[l] = h.q3;
    sm->fr_vis[l] = vis;
  }
  FQ_SYNC();
  int i;
  Slot4 q;
  const bool letters = tag == 2;
  int start = (int)(avg + 0.49);
  u32 p = dec_level_get(sm, (u32)start, s, lev, rs);
  w.st[ST_CTX] += sm->fr_vis[start];
  if (p != FQSX_NIL && slot_counter(s) < (letters ? letters_thr(start) : code_thr(start))) {
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
    avg = ema_update(avg, (double)start);
    return p;
  }
  if (p == FQSX_NIL) {
    for (i = start - 1; i >= 0; --i) {
      p = dec_level_get(sm, (u32)i, s, lev, rs);
      w.st[ST_CTX] += sm->fr_vis[i];
      if (p != FQSX_NIL) break;
    }
  } else {
    for (i = start + 1; i < n_levels; ++i) {
      u32 qi = dec_level_get(sm, (u32)i, q, lev, rs);
      w.st[ST_CTX] += sm->fr_vis[i];
      if (qi == FQSX_NIL) break;
      if (slot_counter(q) < code_thr(i)) {  // both routines use the *code* thresholds here (quirk, dna.cpp:2244)
        avg = ema_update(avg, (double)i);
        ctx_store_counter(w, qi, q, slot_counter(q) + 1);
        s = q;
        return qi;
      }
      p = qi;
      s = q;
    }
    --i;
  }
  if (p == FQSX_NIL) {  // nothing known: create level 0 from the template
    p = ctx_insert(w, tag, LEVKEY(0), tpl_q2, tpl_q3, tpl_total, s);
    if (p == FQSX_NIL) return p;
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
    i = 0;
  }
  if (slot_counter(s) >= code_thr(i) && i + 1 < n_levels) {  // clone into the next level (dna.cpp:2177-2184)
    u32 total = (u32)(s.q1 >> 48);
    Slot4 c;
    u32 ci = ctx_insert(w, tag, LEVKEY(i + 1), s.q2, s.q3, total, c);
    if (ci == FQSX_NIL) return ci;
    ctx_store_counter(w, ci, c, slot_counter(c) + 1);
    s = c;
    p = ci;
  } else
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
  avg = ema_update(avg, (double)i);
  return p;
}

// the worker's reads are decoded into 0..4 codes in a per-worker HBM scratch line (codes) and written as
// ASCII to the output block; a second line holds the reverse-complement part of an anchored second mate
FQ_DEV u8 dec_alpha(u32 sym) { return sym == 0 ? 'A' : sym == 1 ? 'C' : sym == 2 ? 'G' : sym == 3 ? 'T' : 'N'; }

// ---- CRangeDecoder
FQ_DEV u64 rcd_byte(Wk &w) {
  u64 b = w.din_pos < w.din_len ? w.din[w.din_pos] : 0;
  ++w.din_pos;
  return b;
}
FQ_DEV void rcd_start(Wk &w) {  // sub_rc.h:112-125
  w.din_pos = 0;
  w.din_buffer = 0;
  if (w.din_len >= 8)
    for (u32 i = 1; i <= 8; ++i) w.din_buffer |= rcd_byte(w) << (64 - i * 8);
  w.enc.low = 0;
  w.enc.range = 0xff00000000000000ULL;
}
FQ_DEV u32 rcd_cum(Wk &w, u32 tot) {  // GetCumulativeFreq, sub_rc.h:127-131
  w.enc.range = div_u64_small(w.enc.range, tot);
  return (u32)(w.din_buffer / w.enc.range);
}
FQ_DEV void rcd_update(Wk &w, u32 freq, u32 cum) {  // UpdateFrequency, sub_rc.h:133-151
  const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
  u64 r = (u64)cum * w.enc.range, low = w.enc.low + r, range = w.enc.range * freq;
  w.din_buffer -= r;
  while (range <= Top) {
    if ((low ^ (low + range)) & M) range = (low | Top) - low;
    w.din_buffer = (w.din_buffer << 8) + rcd_byte(w);
    low <<= 8;
    range <<= 8;
  }
  w.enc.low = low;
  w.enc.range = range;
  w.st[ST_CODED] += 1;
}

// ---- model decoding
FQ_DEV u32 slot_decode(Wk &w, u32 idx, Slot4 &s) {  // CRangeCoderModelFixedSize<5>::Decode, rc.h:490-503
  u32 st[5] = {(u32)(s.q2 & 0xffff), (u32)((s.q2 >> 16) & 0xffff), (u32)((s.q2 >> 32) & 0xffff), (u32)(s.q2 >> 48), (u32)(s.q3 & 0xffff)};
  u32 tot = (u32)(s.q1 >> 48);
  const u32 lt = rcd_cum(w, tot);
  u32 x = 4, cum = 0, t = 0;
  for (u32 i = 0; i < 5; ++i) {  // GetSym, rc.h:290-302
    t += st[i];
    if (t > lt) { x = i; break; }
    cum = t;
  }
  if (x == 4) cum = st[0] + st[1] + st[2] + st[3];
  rcd_update(w, st[x], cum);
  st[x] += 4;
  tot += 4;
  while (tot >= (1u << 15)) {
    tot = 0;
    for (u32 i = 0; i < 5; ++i) { st[i] = (st[i] + 1) / 2; tot += st[i]; }
  }
  s.q2 = (u64)st[0] | ((u64)st[1] << 16) | ((u64)st[2] << 32) | ((u64)st[3] << 48);
  s.q3 = (s.q3 & ~0xffffULL) | st[4];
  s.q1 = (s.q1 & 0x0000ffffffffffffULL) | ((u64)tot << 48);
  u64 *p = ctx_base(w) + 4 * (u64)idx;
  p[1] = s.q1; p[2] = s.q2; p[3] = s.q3;
  return x;
}
FQ_DEV u32 sm_decode(Wk &w, u16 *m, u32 n, u32 max_total) {  // CRangeCoderModel::Decode, rc.h:407-421
  u32 tot = m[n];
  const u32 lt = rcd_cum(w, tot);
  u32 x = n - 1, cum = 0, t = 0;
  for (u32 i = 0; i < n; ++i) {
    t += m[i];
    if (t > lt) { x = i; break; }
    cum = t;
  }
  if (t <= lt) { cum = 0; for (u32 i = 0; i + 1 < n; ++i) cum += m[i]; }
  u32 f = m[x];
  rcd_update(w, f, cum);
  f += 4;
  tot += 4;
  m[x] = (u16)f;
  while (tot >= max_total) {
    tot = 0;
    for (u32 i = 0; i < n; ++i) { u32 v = (m[i] + 1u) / 2u; m[i] = (u16)v; tot += v; }
  }
  m[n] = (u16)tot;
  return x;
}
FQ_DEV u32 sm_decode256(Wk &w, u16 *m, u8 *init_flag) {
  if (!*init_flag) {
    FQ_SYNC_MEM();
    for (u32 i = FQ_LANE; i < 256; i += FQ_WAVE) m[i] = 1;
    m[256] = 256;
    *init_flag = 1;
    FQ_SYNC_MEM();
  }
  u32 tot = m[256];
  const u32 lt = rcd_cum(w, tot);
  u32 x = 255, cum = 0, t = 0;
  for (u32 i = 0; i < 256; ++i) {
    t += m[i];
    if (t > lt) { x = i; break; }
    cum = t;
  }
  u32 f = m[x];
  rcd_update(w, f, cum);
  tot += 4;
  FQ_SYNC_MEM();
  m[x] = (u16)(f + 4);
  FQ_SYNC_MEM();
  while (tot >= (1u << 15)) {
    u32 pp = 0;
    for (u32 i = FQ_LANE; i < 256; i += FQ_WAVE) { u32 v = (m[i] + 1u) / 2u; m[i] = (u16)v; pp += v; }
    tot = wave_sum32(pp);
    FQ_SYNC_MEM();
  }
  m[256] = (u16)tot;
  return x;
}

FQ_DEV u64 dec_letters_before(const u8 *codes, u32 i, u32 hist_start) {
  u64 ctx = ~0ull;
  u32 t0 = i > 16 ? i - 16 : 0;
  if (t0 < hist_start) t0 = hist_start;
  for (u32 t = t0; t < i; ++t) ctx = (ctx << 4) + codes[t];
  return ctx;
}
FQ_DEV u32 dec_letter(Wk &w, const u8 *codes, u32 pos, u32 read_len, u32 hist_start) {  // dna.cpp:1246-1254,1353-1360
  u64 lev[10];
  ctx_letters_keys(lev, w.cfg, pos, dec_letters_before(codes, pos, hist_start), read_len);
  FQ_SYNC();
  for (u32 l = 0; l < 10; ++l) w.sm->lev_tmp[l] = lev[l];
  FQ_SYNC();
  Slot4 s;
  u32 idx = find_leveled_dec(w, 2, w.sm->lev_tmp, 0, 9, w.avg_letters, TPL_LET_Q2, TPL_LET_Q3, TPL_LET_TOT, s);
  return idx != FQSX_NIL ? slot_decode(w, idx, s) : 0;
}
FQ_DEV u32 dec_un_rank(const Wk &w, const C4 &counts, u32 r) {  // un_rank, dna.cpp:197-207
  if (r == 4) return 4;
  for (u32 i = 0; i < 4; ++i)
    if (rank_sym(w, counts, i) == r) return i;
  return 4;
}
FQ_DEV void dec_put(u8 *codes, u8 *p_out, u32 i, u32 sym) {
  codes[i] = (u8)sym;
  if (p_out) p_out[i] = dec_alpha(sym);
}

// decompress_suffix, dna.cpp:1139-1345
FQ_DEV void suffix_dec(Wk &w, u8 *codes, u8 *p_out, u32 size, bool original_order, u32 start_pos, bool reversed, u32 hist_start) {
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  u64 ctx_r_sym = 0;
  // The cluster a b-mer look-up starts in is fixed by the k-mer's kernel (symbols 2 .. k-3), so the look-up of position
  // i + 1 -- sub-table, home slot, orientation -- is known before symbol i is decoded: its two buckets are requested
  // while the context search and the range decoder of position i run, and matched against the full k-mer afterwards.
  TabG nf;
  nf.s = nullptr; nf.a = nf.b = 0;
  for (u32 x = 0; x < FQSX_BKT; ++x) { nf.ia[x] = 0; nf.ib[x] = 0; }
  for (u32 i = start_pos ? start_pos : original_order ? cfg->prefix : cfg->pmer; i < size && !w.err; ++i) {
    km_insert_zero(w.pm, cfg->gp); km_insert_zero(w.sm_, cfg->gs); km_insert_zero(w.bm, cfg->gb);
    km_insert_zero(w.pm_u, cfg->gp); km_insert_zero(w.sm_u, cfg->gs); km_insert_zero(w.bm_u, cfg->gb);
    C4 counts;
    u32 level;
    TM_BEGIN(t_fc);
    if (km_full(w.bm, cfg->gb)) {   // find_counts' first look-up (dna.cpp:461-476), with the slots requested a position ahead
      const bool nd = km_norm_dir(w.bm, cfg->gb);
      const u64 key = nd ? w.bm.dir : w.bm.rc;
      const u32 sub = sb_owner(cfg, key);
      const u64 *sl = cfg->g_b.slots + (u64)sub * cfg->g_b.stride;
      u64 ns = 0;
      c4_zero(counts);
      if (nf.s == sl && nf.a == tab_home(cfg->g_b, key >> (64 - 2 * cfg->g_b.k)).a) tab_rest(cfg->g_b, nf, key, nd, counts, ns);
      else tab_scan(cfg->g_b, sub, key, nd, counts, ns);   // (a correction changed the k-mer's kernel: ask again)
      w.st[ST_GPROBE] += 1;
      w.st[ST_GSLOT] += ns;
      if (c4_any(counts)) {
        level = LV_BMER;
        if ((counts.c[0] == 63) + (counts.c[1] == 63) + (counts.c[2] == 63) + (counts.c[3] == 63) > 1) {
          C4 c2;
          kt_find(w, cfg->g_s, true, cfg->gs, w.sm_, RNG_S, CINC_S, c2);
          counts.c[0] += c2.c[0]; counts.c[1] += c2.c[1]; counts.c[2] += c2.c[2]; counts.c[3] += c2.c[3];
          level = LV_MIXED;
        }
      } else
        level = find_counts(w, counts, true);
    } else
      level = find_counts(w, counts, false);
    nf.s = nullptr;
    if (w.bm.cur + 1 >= cfg->gb.k && i + 1 < size) {   // the next position's b-mer will be full: request its cluster's first slots
      Kmer nb = w.bm;
      km_insert_zero(nb, cfg->gb);
      const u64 nkey = km_norm_dir(nb, cfg->gb) ? nb.dir : nb.rc;
      nf = tab_first_g(cfg->g_b, sb_owner(cfg, nkey), nkey);
    }
    TM_END(w, TM_FINDC, t_fc);
    TM_BEGIN(t_rg);
    if (level == LV_BMER_UNC) {
      w.bm = w.bm_u; w.sm_ = w.sm_u; w.pm = w.pm_u;
      w.cor_pos = 0;
      level = LV_BMER;
    }
    bool rough = false;
    if (level == LV_NONE) {
      if (km_full(w.bm, cfg->gb)) {
        if (rough_kt(w, cfg->g_b, cfg->gb, w.bm, RNG_B, CINC_B, counts)) { level = LV_PMER; rough = true; }
      } else if (km_full(w.sm_, cfg->gs)) {
        if (rough_kt(w, cfg->g_s, cfg->gs, w.sm_, RNG_S, CINC_S, counts)) { level = LV_PMER; rough = true; }
      } else if (km_full(w.pm, cfg->gp)) {
        if (rough_p(w, counts)) { level = LV_PMER; rough = true; }
      }
    }
    TM_END(w, TM_ROUGH, t_rg);
    u32 sym;
    if (level != LV_NONE && w.N_run < 2) {
      TM_BEGIN(t_k);
      int cor_dist = level == LV_PMER ? (int)cfg->pmer : level == LV_SMER ? (int)cfg->smer : (int)cfg->bmer;
      int d = (int)i - (int)w.cor_pos;
      u32 cor_zone = d < cor_dist ? (u32)(1 + 2 * (cor_dist - d) / cor_dist) : 0u;
      if (rough) cor_zone = 3;
      u64 lev[7];
      if (!reversed) ctx_codes_wave(lev, cfg, counts, w.s_let, i, level, cor_zone, ctx_r_sym, size);
      else ctx_codes_wave(lev, cfg, counts, w.s_let, size - i - 1, level, cor_zone, ctx_r_sym, ~0u);
      FQ_SYNC();
      for (u32 l = 0; l < 7; ++l) sm->lev_tmp[l] = lev[l];
      FQ_SYNC();
      Slot4 s;
      TM_END(w, TM_KEYS, t_k);
      TM_BEGIN(t_l);
      u32 idx = find_leveled_dec(w, 1, sm->lev_tmp, 0, 7, w.avg_code, TPL_CODES_Q2, TPL_CODES_Q3, TPL_CODES_TOT, s);
      TM_END(w, TM_CR_S, t_l);
      TM_BEGIN(t_d);
      u32 r_sym = idx != FQSX_NIL ? slot_decode(w, idx, s) : 0;
      sym = dec_un_rank(w, counts, r_sym);
      TM_END(w, TM_CR_RC, t_d);
      ctx_r_sym = ((ctx_r_sym << 1) + (r_sym == 0 ? 1u : 0u)) & 0xff;
    } else {
      sym = dec_letter(w, codes, i, size, hist_start);
      ctx_r_sym = (ctx_r_sym << 1) & 0xff;
    }
    TM_BEGIN(t_q);
    dec_put(codes, p_out, i, sym);
    const u64 sym_k = sym == 4 ? 0 : sym;
    if (sym == 4) ++w.N_run; else w.N_run = 0;
    replace_last_all(w, sym_k);
    if (sym < 4) {
      bool pmer_insert = true;
      if (km_full(w.bm, cfg->gb)) {
        push_b_local(w);
        if ((level == LV_SMER || level == LV_BMER || level == LV_MIXED) && c4_get(counts, sym) >= 3) pmer_insert = false;
      }
      if (km_full(w.sm_, cfg->gs)) mail_push(w, MAIL_S, km_norm(w.sm_, cfg->gs));
      if (km_full(w.pm, cfg->gp) && i - w.cor_pos >= cfg->pmer - 1) {
        if (pmer_insert) push_p_both(w); else w.hidden += 2;
      }
    }
    if (km_full(w.bm, cfg->gb)) {
      bool rep = false;
      if (level == LV_BMER || level == LV_MIXED) rep = repair_existing(w, i, counts, sym);
      else if (level == LV_NONE || level == LV_PMER) rep = repair_missing(w, i);
      if (rep) push_b_local(w);
    }
    TM_END(w, TM_POST, t_q);
  }
}

FQ_DEV void prefix_direct_dec(Wk &w, u8 *codes, u8 *p_out) {  // decompress_prefix_direct, dna.cpp:1347-1381
  for (u32 i = 0; i < w.cfg->prefix; ++i) {
    u32 sym = dec_letter(w, codes, i, 0, 0);
    dec_put(codes, p_out, i, sym);
    if (sym == 4) { sym = 0; w.cor_pos = i; }
    insert_all(w, sym);
  }
}

// index of the (dif+1)-th field equal to `flag` after position `lo` of the p-mer vector (dna.cpp:1432-1437):
// the words up to the end of lo's block are swept, whole blocks are then skipped through the count index
// (DevCfg.siv_idx), and the block the answer lies in is swept again.
// One sweep: words [wa, wb), fields below `start` masked out; `run` = matching fields seen so far.
FQ_DEV bool siv_select_sweep(Wk &w, u64 wa, u64 wb, u64 start, u64 flag, u64 dif, u64 &run, u64 &ans) {
  const u64 *sv = w.cfg->siv;
  const u64 rep = flag * 0x5555555555555555ULL;
  for (u64 base = wa; base < wb; base += FQ_WAVE) {
    const u64 x = base + FQ_LANE;
    u64 eq = 0;
    if (x < wb) {
      u64 d = sv[x] ^ rep;
      eq = ~(d | (d >> 1)) & 0x5555555555555555ULL;
      if (x == (start >> 5)) eq &= ~0ull << (2 * (start & 31));
    }
    const u32 c = popc64(eq);
    const u32 ex = wave_excl_scan32(c), tot = wave_sum32(c);
    w.st[ST_SIV_WORDS] += FQ_WAVE;
    if (run + tot > dif) {
      const u64 want = dif - run;  // rank inside this group of words
      const bool mine = want >= ex && want < (u64)ex + c;
      u64 a = 0;
      if (mine) {
        u64 e = eq;
        for (u64 r = want - ex; r; --r) e &= e - 1;
        a = x * 32 + ctz64(e) / 2;
      }
#if FQ_WAVE > 1
      const u64 bm = wave_ballot(mine);
      ans = wave_bcast64(a, ctz64(bm));
#else
      ans = a;
#endif
      return true;
    }
    run += tot;
  }
  return false;
}
FQ_DEV u64 siv_select_equal(Wk &w, u64 lo, u64 dif, u64 flag) {
  const u64 start = lo + 1, n_words = (1ull << (2 * w.cfg->pmer)) / 32, n_blocks = n_words >> (FQSX_SIV_BLK_LOG - 5);
  const u32 wpb = FQSX_SIV_BLK / 32;   // words per block
  u64 run = 0, ans = 0;
  u64 b = start >> FQSX_SIV_BLK_LOG;
  if (b >= n_blocks) { w.err = FQSX_ERR_DECODE; return 0; }
  if (siv_select_sweep(w, start >> 5, (b + 1) * wpb, start, flag, dif, run, ans)) return ans;
  // whole blocks, a wave's worth at a time
  struct alignas(16) I4 { u32 c[4]; };
  const I4 *ix = (const I4 *)w.cfg->siv_idx;
  for (u64 base = b + 1; base < n_blocks; base += FQ_WAVE) {
    const u64 x = base + FQ_LANE;
    u32 c = 0;
    if (x < n_blocks) {
      const I4 v = ix[x];
      c = flag ? (flag == 1 ? v.c[1] : flag == 2 ? v.c[2] : v.c[3]) : FQSX_SIV_BLK - v.c[1] - v.c[2] - v.c[3];
    }
    const u32 ex = wave_excl_scan32(c), tot = wave_sum32(c);
    w.st[ST_SIV_WORDS] += 2 * FQ_WAVE;
    if (run + tot > dif) {
      const u64 want = dif - run;
      const bool mine = want >= ex && want < (u64)ex + c;
#if FQ_WAVE > 1
      const u32 src = ctz64(wave_ballot(mine));
      const u64 blk = wave_bcast64(x, src);
      run += wave_bcast32(ex, src);
#else
      const u64 blk = x;
      run += ex;
#endif
      if (siv_select_sweep(w, blk * wpb, (blk + 1) * wpb, 0, flag, dif, run, ans)) return ans;
      break;   // (index and vector disagree: cannot happen)
    }
    run += tot;
  }
  w.err = FQSX_ERR_DECODE;
  return 0;
}

FQ_DEV void prefix_sorted_dec(Wk &w, u8 *codes, u8 *p_out) {  // decompress_prefix_sorted, dna.cpp:1384-1514
  const DevCfg *cfg = w.cfg;
  WState *ws = w.ws;
  u16 *sb = small_base(w);
  const bool was_N = sm_decode(w, sb + SM_OFF_NS, SM_NS_N, 1u << 12) != 0;
  u64 psf = ((ws->ctx_ps_flags << 1) + (was_N ? 1u : 0u)) & 0xffff;
  const u64 flag = sm_decode(w, sb + SM_OFF_PSF + psf * (SM_PSF_N + 1), SM_PSF_N, 1u << 12);
  psf = ((psf << 3) + flag) & 0xffff;
  ws->ctx_ps_flags = psf;
  if (flag < 4) {
    const u32 nb = sm_decode(w, sb + SM_OFF_PSNB + psf * 6u, cfg->ps_nobytes_n, 1u << 12) + 1;
    u64 dif;
    if (nb == 1) {
      u32 hi = sm_decode(w, sb + SM_OFF_NIB + (u32)flag * (SM_NIB_N + 1), SM_NIB_N, 1u << 15);
      u32 lo = sm_decode(w, sb + SM_OFF_NIB + (4u + (u32)flag * 16u + hi) * (SM_NIB_N + 1), SM_NIB_N, 1u << 15);
      dif = ((u64)hi << 4) + lo;
    } else {
      u8 *bi = cfg->byte_init + (u64)w.tid * SM_LAZY_ENTRIES;
      const u32 e = (u32)flag * 4u + (nb - 2);
      const u32 hi_byte = sm_decode256(w, sb + SM_OFF_BYTE + (u64)e * (SM_BYTE_N + 1), bi + e);
      dif = (u64)hi_byte << (nb * 8 - 8);
      for (u32 i = 0; i + 1 < nb; ++i) {
        const u32 e2 = 16u + ((e * 256u + hi_byte) * 4u + i);
        dif += (u64)sm_decode256(w, sb + SM_OFF_BYTE + (u64)e2 * (SM_BYTE_N + 1), bi + e2) << (i * 8);
      }
    }
    const u64 prev_idx = ws->pmer_prev_cur ? ws->pmer_prev_dir >> (64 - 2 * ws->pmer_prev_cur) : 0;
    const u64 k = siv_select_equal(w, prev_idx, dif, flag);
    w.pm.cur = cfg->pmer;  // the p-mer with index k, as insert_front would build it (dna.cpp:1439-1443)
    w.pm.dir = k << (64 - 2 * cfg->pmer);
    w.pm.rc = 0;
    for (u32 i = 0; i < cfg->pmer; ++i) {
      u64 sym = (k >> (2 * (cfg->pmer - 1 - i))) & 3;
      w.pm.rc |= (3 - sym) << (64 - 2 * cfg->pmer + 2 * i);
    }
  } else if (ws->pmer_prev_cur == cfg->pmer) {
    w.pm.dir = ws->pmer_prev_dir; w.pm.rc = ws->pmer_prev_rc; w.pm.cur = cfg->pmer;
  } else {
    for (u32 i = 0; i < cfg->pmer; ++i) km_insert(w.pm, cfg->gp, 0);
    ws->pmer_prev_dir = w.pm.dir; ws->pmer_prev_rc = w.pm.rc; ws->pmer_prev_cur = w.pm.cur;
  }
  for (u32 i = 0; i < cfg->pmer; ++i) {
    u32 sym = (u32)km_symbol(w.pm, i);
    if (was_N && sym == 3 && sm_decode(w, sb + SM_OFF_NS + (i + 1) * (SM_NS_N + 1), SM_NS_N, 1u << 12)) sym = 4;
    dec_put(codes, p_out, i, sym);
    if (sym == 4) { sym = 3; w.N_run++; } else w.N_run = 0;
    km_insert(w.sm_, cfg->gs, sym); km_insert(w.bm, cfg->gb, sym);
    km_insert(w.pm_u, cfg->gp, sym); km_insert(w.sm_u, cfg->gs, sym); km_insert(w.bm_u, cfg->gb, sym);
  }
  ws->pmer_prev_dir = w.pm.dir; ws->pmer_prev_rc = w.pm.rc; ws->pmer_prev_cur = w.pm.cur;
  push_p_both(w);
}

FQ_DEV void dec_update_s_letters(Wk &w, const u8 *codes, u32 size) {
  u32 h0 = 0, h1 = 0, h2 = 0, h3 = 0;
  FQ_SYNC_MEM();
  for (u32 i = FQ_LANE; i < size; i += FQ_WAVE) {
    u32 c = codes[i];
    h0 += c == 0; h1 += c == 1; h2 += c == 2; h3 += c == 3;
  }
  h0 = wave_sum32(h0); h1 = wave_sum32(h1); h2 = wave_sum32(h2); h3 = wave_sum32(h3);
  w.s_let[0] += h0 + h3; w.s_let[3] += h0 + h3;
  w.s_let[1] += h1 + h2; w.s_let[2] += h1 + h2;
  w.st[ST_BASES] += size;
}

// DecompressSE, dna.cpp:1883-1928 (first_of_pair) / the direct second mate of DecompressPE, :2029-2033.
// prev_out: previous first mate of this worker in the output block (read_prev), or null.
FQ_DEV bool read_dec(Wk &w, u8 *codes, u8 *p_out, u32 size, const u8 *prev_out, bool first_of_pair) {
  const DevCfg *cfg = w.cfg;
  const bool orig = !first_of_pair || w.mode == 0 || w.mode == 2;
  if (first_of_pair) {
    u16 *m = small_base(w) + SM_OFF_FLAGS + w.ws->ctx_flags * (SM_FLAGS_N + 1);
    const bool same = sm_decode(w, m, SM_FLAGS_N, 1u << 12) != 0;
    w.ws->ctx_flags = ((w.ws->ctx_flags << 1) + (same ? 1u : 0u)) & 0xff;
    if (same) {
      FQ_SYNC_MEM();
      for (u32 i = FQ_LANE; i < size; i += FQ_WAVE) {
        u8 ch = prev_out ? prev_out[i] : (u8)'A';
        p_out[i] = ch;
        codes[i] = (u8)dna_code(ch);
      }
      FQ_SYNC_MEM();
      return true;
    }
  }
  km_reset(w.pm); km_reset(w.sm_); km_reset(w.bm);
  km_reset(w.pm_u); km_reset(w.sm_u); km_reset(w.bm_u);
  w.cor_pos = 0;
  w.N_run = 0;
  if (orig) prefix_direct_dec(w, codes, p_out); else prefix_sorted_dec(w, codes, p_out);
  suffix_dec(w, codes, p_out, size, orig, 0, false, 0);
  dec_update_s_letters(w, codes, size);
  return false;
}

// DecompressPE, dna.cpp:1931-2044
FQ_DEV void pair_dec(Wk &w, u8 *codes, u8 *rcodes, u8 *p1, u32 size1, u8 *p2, u32 size2, const u8 *prev_out) {
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  const int k = (int)cfg->bmer;
  const u64 vm = pe_value_mask(cfg);
  read_dec(w, codes, p1, size1, prev_out, true);
  if (w.err) return;
  u64 m1[4], a1[3], x1;
  {
    int mss = (int)size1 - k + 1, s1 = mss / 4, s2 = 2 * mss / 4, s3 = 3 * mss / 4;
    m1[0] = pe_find_minimizer(cfg, codes, 0, s1 + k - 1);
    m1[1] = pe_find_minimizer(cfg, codes, s1, s2 - s1 + k - 1);
    m1[2] = pe_find_minimizer(cfg, codes, s2, s3 - s2 + k - 1);
    m1[3] = pe_find_minimizer(cfg, codes, s3, (int)size1 - s3);
    int a = mss / 3, b = 2 * mss / 3;
    a1[0] = pe_find_minimizer(cfg, codes, 0, a + k - 1);
    a1[1] = pe_find_minimizer(cfg, codes, a, b - a + k - 1);
    a1[2] = pe_find_minimizer(cfg, codes, b, (int)size1 - b);
    int mid1 = ((int)size1 + k) / 2;
    x1 = (~pe_find_maximizer(cfg, codes, mid1 - k + 1, (int)size1 - (mid1 - k + 1))) & vm;
  }
  u32 nc = 0;
  for (u32 i = 0; i < 4; ++i) ptab_find(w, cfg->g_pe, pe_owner(cfg, murmur64(m1[i])), m1[i], nc);
  for (u32 i = 0; i < 4; ++i) ptab_find(w, cfg->l_pe, w.tid, m1[i], nc);
  bool direct = nc == 0;
  u32 mid = 0, mpos = 0;
  u16 *sb = small_base(w);
  if (nc) {
    pe_merge_candidates(w, nc);
    mid = sm_decode(w, sb + SM_OFF_MID, SM_NIB_N, 1u << 15);
    if (mid == 15) direct = true;
    else {
      u8 *bi = cfg->byte_init + (u64)w.tid * SM_LAZY_ENTRIES + SM_BYTE_ENTRIES;
      u16 *mp = sb + SM_OFF_MPOS;
#define MPOS_DEC(cls) sm_decode256(w, mp + (u64)((cls) * 16 + mid) * (SM_BYTE_N + 1), bi + ((cls) * 16 + mid))
      mpos = MPOS_DEC(0);
      if (mpos == 254) { mpos = MPOS_DEC(1) << 8; mpos += MPOS_DEC(2); }
      else if (mpos == 255) { mpos = MPOS_DEC(3) << 16; mpos += MPOS_DEC(4) << 8; mpos += MPOS_DEC(5); }
#undef MPOS_DEC
    }
  }
  // second mate decoded into the same code line (the first mate's codes are no longer needed)
  if (direct) read_dec(w, codes, p2, size2, nullptr, false);
  else {
    // DecompressDirectWithMinim, dna.cpp:1641-1713: the anchor b-mer is candidate `mid`
    if (mpos + (u32)k > size2) { w.err = FQSX_ERR_DECODE; return; }
    const u64 minim = sm->pe_top[mid] & vm;
    for (int i = 0; i < k; ++i) dec_put(codes, p2, mpos + (u32)i, (u32)((minim >> (2 * (k - 1 - i))) & 3));
    pe_seed_kmers(w, codes, mpos, mpos + (u32)k);
    suffix_dec(w, codes, p2, size2, true, (u32)k + mpos, false, mpos);
    for (int i = 0; i < k; ++i) rcodes[i] = (u8)(3 - codes[mpos + (u32)(k - 1 - i)]);
    pe_seed_kmers(w, rcodes, 0, (u32)k);
    suffix_dec(w, rcodes, nullptr, mpos + (u32)k, true, (u32)k, true, 0);
    for (u32 i = 0; i < mpos; ++i) {  // p[i] = rc(rc_p[k + (mpos-1-i)]), dna.cpp:1709-1710
      u32 c = rcodes[(u32)k + (mpos - 1 - i)];
      dec_put(codes, p2, i, c == 4 ? 4 : 3 - c);
    }
    dec_update_s_letters(w, codes, size2);
  }
  if (w.err) return;
  u64 a2[3], x2;
  {
    int mss = (int)size2 - k + 1, a = mss / 3, b = 2 * mss / 3;
    a2[0] = pe_find_minimizer(cfg, codes, 0, a + k - 1);
    a2[1] = pe_find_minimizer(cfg, codes, a, b - a + k - 1);
    a2[2] = pe_find_minimizer(cfg, codes, b, (int)size2 - b);
    int mid2 = ((int)size2 + k) / 2;
    x2 = (~pe_find_minimizer(cfg, codes, mid2 - k + 1, (int)size2 - (mid2 - k + 1))) & vm;
  }
  pe_push(w, a1[0], a2[0], 2); pe_push(w, a1[0], a2[2], 4); pe_push(w, a1[0], x1, 1);
  pe_push(w, a1[1], a2[0], 3); pe_push(w, a1[1], a2[2], 3);
  pe_push(w, a1[2], a2[0], 4); pe_push(w, a1[2], a2[2], 2);
  pe_push(w, a2[0], a1[0], 2); pe_push(w, a2[0], a1[2], 4); pe_push(w, a2[0], x2, 1);
  pe_push(w, a2[1], a1[0], 3); pe_push(w, a2[1], a1[2], 4);
  pe_push(w, a2[2], a1[0], 4); pe_push(w, a2[2], a1[2], 2);
}
