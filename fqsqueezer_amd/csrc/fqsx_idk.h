// fqsx_idk.h -- read-id stream on the GPU (CIdCompressor, fqs/id.cpp:84-135 models and reset,
// :152-184 CompressPE, :257-418 compress_lossless, :421-495 compress_instrument, :734-757 tokenize; mtf.cpp:52-116).
// One wavefront per logical worker, no synchronisation points (workers are independent).  An id is string work on a few
// dozen bytes with a strictly sequential coder, so the wave runs the control flow uniformly (every lane the same values:
// the compiler keeps them in scalar registers) and uses its lanes where there is width: staging and classifying the
// bytes, the cumulative frequency / rescaling of the 128- and 256-symbol models, the move-to-front list of instrument
// names.  Models live in two per-worker open-addressed tables in HBM, keyed by (map, context):
//   small  [key | 4 x u16]            the 2- and 4-symbol maps (flags, pe_flags, literal_same, literal_same_length, numeric_small)
//   big    [key | 256 x u16 | total]  the 128- and 256-symbol maps (literal, plain, numeric_size)
// plus twelve fixed big models (mtf_flag, mtf_code[7], mtf_byte[4]).  Layout-free like everything else here: results depend
// on model contents only.  Included by fqsx_api.hip after fqsx_qual.h (QEnc: the range coder).
#pragma once
#include "fqsx_plat.h"

#define IDK_MAX_ID 1024u      // bytes of one id line (with its line feed) a worker stages in LDS
#define IDK_MAX_TOK 128u      // tokens of one id
#define IDK_NAME 64u          // bytes of a move-to-front entry: length byte + up to 63 characters
#define IDK_BIG_U64 66u       // big slot: key, 64 words of statistics (256 x u16), total
#define IDK_FIXED 12u         // mtf_flag, mtf_code[0..6], mtf_byte[0..3]
enum { IDM_FLAGS = 1, IDM_PE_FLAGS, IDM_NUM_SMALL, IDM_LIT_SAME, IDM_LIT_SAME_LEN, IDM_NUM_SIZE, IDM_LITERAL, IDM_PLAIN };
enum { IDK_ERR_BYTE = 1 /* byte >= 128 (the reference's 128-symbol models, id.cpp:99,104) */, IDK_ERR_NO_INSTRUMENT = 2, IDK_ERR_OUT = 3,
       IDK_ERR_TABLE = 4, IDK_ERR_TOO_LONG = 5 /* id longer than IDK_MAX_ID, more than IDK_MAX_TOK tokens, instrument name beyond 63 bytes */,
       IDK_ERR_MTF_FULL = 6 };

struct IdCfg {
  u32 T, mode;             // mode 0 lossless, 1 instrument (params.h:18,92)
  u64 *small;              // [T][small_cap][2], key ~0 = empty
  u64 small_mask;
  u64 *big;                // [T][big_cap][IDK_BIG_U64]
  u64 big_mask;
  u64 *fixed;              // [T][IDK_FIXED][IDK_BIG_U64] (key word unused)
  u8 *mtf;                 // [T][mtf_cap][IDK_NAME]
  u32 mtf_cap;
  u32 *state;              // [T][4]: small slots used, big slots used, mtf entries
  const u8 *ids;           // block input: concatenated id lines, each with its line feed
  const u64 *off;          // n_reads + 1
  u8 *out;                 // [T][out_cap]
  u64 out_cap;
  u64 *lens;               // [T]
  u32 *err;
};

// LDS of one worker
struct IdShared {
  u8 cur[IDK_MAX_ID], prev[IDK_MAX_ID];
  u8 cls[IDK_MAX_ID];                       // 0 digit, 1 other literal byte, 2 separator
  u16 tb[2][IDK_MAX_TOK], te[2][IDK_MAX_TOK];
  u8 tn[2][IDK_MAX_TOK], ts[2][IDK_MAX_TOK];   // numeric flag, separator byte
  long long deltas[IDK_MAX_TOK];
  u8 name[IDK_NAME];
};

struct IdK {
  const IdCfg *cfg;
  IdShared *sm;
  QEnc e;
  u64 *small, *big, *fixed;
  u8 *mtf;
  u32 n_small, n_big, n_mtf;
  u32 cur_set;             // which of tb/te/tn/ts holds the current id's tokens (the other: the previous id's)
  u32 n_tok[2];
  u32 prev_size;
  u64 ctx_flags, ctx_pe_flags;
  u32 err;
};

FQ_DEV u32 idk_uniform32(u32 v) {   // (value computed alike in every lane: tell the compiler)
#if FQ_WAVE > 1
  return (u32)__builtin_amdgcn_readfirstlane((int)v);
#else
  return v;
#endif
}

// ---- models -----------------------------------------------------------------------------------------------------------
// 2- / 4-symbol model of `map` at context `ctx` (created all ones on first use, rc.h:69-74): encode symbol x
FQ_DEV void idk_small(IdK &k, u32 map, u64 ctx, u32 N, u32 x) {
  const u64 key = ((u64)map << 56) | ctx;
  u64 h = q_hash(key) & k.cfg->small_mask;
  u64 *slot = nullptr;
  u64 st = 0;
  for (u64 it = 0; it <= k.cfg->small_mask; ++it) {
    u64 *p = k.small + 2 * h;
    const u64 kk = p[0];
    if (kk == key) { slot = p; st = p[1]; break; }
    if (kk == ~0ull) {
      if ((u64)(k.n_small + 1) * 10 >= (k.cfg->small_mask + 1) * 9) { k.err = IDK_ERR_TABLE; return; }
      slot = p;
      st = N == 2 ? 0x0000000000010001ULL : 0x0001000100010001ULL;
      if (FQ_LANE == 0) p[0] = key;
      k.n_small += 1;
      break;
    }
    h = (h + 1) & k.cfg->small_mask;
  }
  if (!slot) { k.err = IDK_ERR_TABLE; return; }
  u32 s[4] = {(u32)(st & 0xffff), (u32)((st >> 16) & 0xffff), (u32)((st >> 32) & 0xffff), (u32)(st >> 48)};
  u32 tot = 0, cum = 0;
  for (u32 i = 0; i < N; ++i) { tot += s[i]; if (i < x) cum += s[i]; }
  q_encode(k.e, s[x], cum, tot);
  s[x] += 1;
  tot += 1;
  while (tot >= (1u << 15)) {
    tot = 0;
    for (u32 i = 0; i < N; ++i) { s[i] = (s[i] + 1) / 2; tot += s[i]; }
  }
  st = (u64)s[0] | ((u64)s[1] << 16) | ((u64)s[2] << 32) | ((u64)s[3] << 48);
  if (FQ_LANE == 0) slot[1] = st;
  FQ_SYNC_MEM();
}
// N-symbol model (N <= 256) in a big slot: four statistics per lane
FQ_DEV void idk_big_encode(IdK &k, u64 *slot, u32 N, u32 x, bool fresh) {
  u64 *w = slot + 1;
  u32 mine[4] = {0, 0, 0, 0};
  u32 tot, freq, cum = 0;
  if (fresh) {
    FQ_SYNC_MEM();
    for (u32 l = FQ_LANE; l < 64; l += FQ_WAVE) w[l] = 0x0001000100010001ULL;   // (entries beyond N are never read)
    if (FQ_LANE == 0) w[64] = N;
    FQ_SYNC_MEM();
    tot = N; freq = 1; cum = x;
  } else {
    u32 part = 0;
    for (u32 l = FQ_LANE; l < 64; l += FQ_WAVE) {
      if (4 * l < N && 4 * l < x) {
        const u64 v = w[l];
        for (u32 f = 0; f < 4; ++f) if (4 * l + f < x) part += (u32)((v >> (16 * f)) & 0xffff);
      }
    }
    cum = wave_sum32(part);
    freq = (u32)((w[x >> 2] >> (16 * (x & 3))) & 0xffff);
    tot = (u32)w[64];
  }
  (void)mine;
  q_encode(k.e, freq, cum, tot);
  // update (rc.h:41-55): stats[x] += 1, total += 1, halve everything while the total reaches 2^15
  tot += 1;
  if (FQ_LANE == 0) { w[x >> 2] += 1ull << (16 * (x & 3)); }
  FQ_SYNC_MEM();
  while (tot >= (1u << 15)) {
    u32 part = 0;
    for (u32 l = FQ_LANE; l < 64; l += FQ_WAVE) {
      if (4 * l < N) {
        const u64 v = w[l];
        u64 nv = 0;
        for (u32 f = 0; f < 4; ++f) {
          u32 s = (u32)((v >> (16 * f)) & 0xffff);
          if (4 * l + f < N) { s = (s + 1) / 2; part += s; }
          nv |= (u64)s << (16 * f);
        }
        w[l] = nv;
      }
    }
    tot = wave_sum32(part);
    FQ_SYNC_MEM();
  }
  if (FQ_LANE == 0) w[64] = tot;
  FQ_SYNC_MEM();
}
FQ_DEV void idk_big(IdK &k, u32 map, u64 ctx, u32 N, u32 x) {
  const u64 key = ((u64)map << 56) | ctx;
  u64 h = q_hash(key) & k.cfg->big_mask;
  for (u64 it = 0; it <= k.cfg->big_mask; ++it) {
    u64 *p = k.big + IDK_BIG_U64 * h;
    const u64 kk = p[0];
    if (kk == key) { idk_big_encode(k, p, N, x, false); return; }
    if (kk == ~0ull) {
      if ((u64)(k.n_big + 1) * 10 >= (k.cfg->big_mask + 1) * 9) { k.err = IDK_ERR_TABLE; return; }
      if (FQ_LANE == 0) p[0] = key;
      k.n_big += 1;
      idk_big_encode(k, p, N, x, true);
      return;
    }
    h = (h + 1) & k.cfg->big_mask;
  }
  k.err = IDK_ERR_TABLE;
}
FQ_DEV void idk_fixed(IdK &k, u32 which, u32 N, u32 x) {   // the host initialises these at creation
  idk_big_encode(k, k.fixed + IDK_BIG_U64 * which, N, x, false);
}

// ---- compress_lossless, id.cpp:257-418 ------------------------------------------------------------------------------------
FQ_DEV bool idk_is_num(u8 c) { return c >= '0' && c <= '9'; }
FQ_DEV bool idk_is_lit(u8 c) { return idk_is_num(c) || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '@'; }   // id.cpp:57-70
FQ_DEV u64 idk_ilog2(u64 x) { u64 r = 0; for (; x; ++r) x >>= 1; return r; }   // utils.h:154-162
FQ_DEV long long idk_get_int(const u8 *p, u32 b, u32 e) {
  long long r = 0;
  for (u32 i = b; i < e; ++i) r = r * 10 + (long long)(p[i] - '0');
  return r;
}
// `size` bytes of the id are in sm->cur (classes in sm->cls)
FQ_DEV void idk_lossless(IdK &k, u32 size) {
  IdShared *sm = k.sm;
  const u32 cs = k.cur_set, ps = cs ^ 1u;
  // tokenize (id.cpp:734-757): a token ends at every separator; an all-digit token of 1..10 digits is numeric
  u32 n_tok = 0, start = 0;
  bool numeric = true;
  for (u32 i = 0; i < size; ++i) {
    const u32 c = sm->cls[i];
    if (c == 2) {
      if (numeric && (i - start >= 11 || i == start)) numeric = false;
      if (n_tok >= IDK_MAX_TOK) { k.err = IDK_ERR_TOO_LONG; return; }
      if (FQ_LANE == 0) { sm->tb[cs][n_tok] = (u16)start; sm->te[cs][n_tok] = (u16)i; sm->tn[cs][n_tok] = numeric ? 1 : 0; sm->ts[cs][n_tok] = sm->cur[i]; }
      ++n_tok;
      numeric = true;
      start = i + 1;
    } else if (c == 1)
      numeric = false;
  }
  FQ_SYNC();
  k.n_tok[cs] = n_tok;
  bool same_types = n_tok == k.n_tok[ps];
  for (u32 i = 0; same_types && i < n_tok; ++i) same_types = sm->tn[cs][i] == sm->tn[ps][i] && sm->ts[cs][i] == sm->ts[ps][i];
  const u8 *p = sm->cur, *q = sm->prev;
  if (same_types) {
    idk_small(k, IDM_FLAGS, k.ctx_flags, 2, 1);
    k.ctx_flags = ((k.ctx_flags << 1) + 1) & 0xff;
    for (u32 i = 0; i < n_tok && !k.err; ++i) {
      const u32 cb = sm->tb[cs][i], ce = sm->te[cs][i], pb = sm->tb[ps][i], pe = sm->te[ps][i];
      if (!sm->tn[cs][i]) {
        const u32 len = ce - cb;
        const bool same_length = len == pe - pb;
        bool same = same_length;
        for (u32 j = 0; same && j < len; ++j) same = p[cb + j] == q[pb + j];
        if (same) { idk_small(k, IDM_LIT_SAME, i, 2, 1); continue; }
        idk_small(k, IDM_LIT_SAME, i, 2, 0);
        if (same_length) {
          idk_small(k, IDM_LIT_SAME_LEN, i, 2, 1);
          for (u32 j = 0; j < len && !k.err; ++j) idk_big(k, IDM_LITERAL, k.ctx_flags + (1ull << 32) + j, 128, p[cb + j] == q[pb + j] ? 0u : (u32)p[cb + j]);
        } else {
          idk_small(k, IDM_LIT_SAME_LEN, i, 2, 0);
          for (u32 j = 0; j < len && !k.err; ++j) idk_big(k, IDM_LITERAL, k.ctx_flags + j, 128, p[cb + j]);
          idk_big(k, IDM_LITERAL, k.ctx_flags + len, 128, 0);
        }
      } else {
        long long delta = idk_get_int(p, cb, ce) - idk_get_int(q, pb, pe);
        const long long d0 = sm->deltas[i];
        u64 ctx = (u64)i << 40;
        ctx += idk_ilog2((u64)(d0 < 0 ? -d0 : d0)) << 31;
        ctx += (u64)(d0 < 0) << 30;
        const u64 ctx0 = ctx;
        FQ_SYNC();
        if (FQ_LANE == 0) sm->deltas[i] = delta;
        FQ_SYNC();
        if (delta >= -1 && delta <= 1) { idk_small(k, IDM_NUM_SMALL, ctx0, 4, (u32)(delta + 1)); continue; }
        idk_small(k, IDM_NUM_SMALL, ctx0, 4, 3);
        int n_bytes = 0;
        if (delta >= -123 && delta <= 123) idk_big(k, IDM_NUM_SIZE, ctx0, 256, (u32)(delta + 123) & 0xff);
        else if (delta > 0 && delta < 0x10000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 247); n_bytes = 2; ctx += 0x10; }
        else if (delta > 0 && delta < 0x1000000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 248); n_bytes = 3; ctx += 0x20; }
        else if (delta > 0 && delta < 0x100000000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 249); n_bytes = 4; ctx += 0x30; }
        else if (delta > 0) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 250); n_bytes = 8; ctx += 0x40; }
        else if (delta > -0x10000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 251); delta = -delta; n_bytes = 2; ctx += 0x50; }
        else if (delta > -0x1000000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 252); delta = -delta; n_bytes = 3; ctx += 0x60; }
        else if (delta > -0x100000000ll) { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 253); delta = -delta; n_bytes = 4; ctx += 0x70; }
        else { idk_big(k, IDM_NUM_SIZE, ctx0, 256, 254); delta = -delta; n_bytes = 8; ctx += 0x80; }
        for (int j = 0; j < n_bytes && !k.err; ++j) idk_big(k, IDM_NUM_SIZE, ctx + (u64)j, 256, (u32)(((u64)delta >> (8 * j)) & 0xff));
      }
    }
  } else {
    idk_small(k, IDM_FLAGS, k.ctx_flags, 2, 0);
    k.ctx_flags = (k.ctx_flags << 1) & 0xff;
    for (u32 i = 0; i < size && !k.err; ++i) idk_big(k, IDM_PLAIN, i, 128, p[i]);
    FQ_SYNC();
    for (u32 i = FQ_LANE; i < n_tok; i += FQ_WAVE) sm->deltas[i] = 0;
    FQ_SYNC();
  }
  // the current id becomes the previous one
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < size; i += FQ_WAVE) sm->prev[i] = sm->cur[i];
  FQ_SYNC();
  k.prev_size = size;
  k.cur_set = ps;
}
// bytes [0, size) of a line in HBM -> sm->cur / sm->cls; false if a byte is outside the 128-symbol alphabet
FQ_DEV bool idk_stage(IdK &k, const u8 *p, u32 size) {
  IdShared *sm = k.sm;
  bool bad = false;
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < size; i += FQ_WAVE) {
    const u8 c = p[i];
    sm->cur[i] = c;
    sm->cls[i] = idk_is_num(c) ? 0 : idk_is_lit(c) ? 1 : 2;
    bad |= c >= 128;
  }
  FQ_SYNC();
  return !wave_any(bad);
}
FQ_DEV void idk_id_lossless(IdK &k, const u8 *p, u32 size) {
  if (size > IDK_MAX_ID) { k.err = IDK_ERR_TOO_LONG; return; }
  if (!idk_stage(k, p, size)) { k.err = IDK_ERR_BYTE; return; }
  idk_lossless(k, size);
}
// compress_instrument, id.cpp:421-495: only the instrument name (up to the first '.', ' ' or ':') is kept
FQ_DEV void idk_id_instrument(IdK &k, const u8 *p, u32 size) {
  IdShared *sm = k.sm;
  if (size > IDK_MAX_ID) { k.err = IDK_ERR_TOO_LONG; return; }
  if (!idk_stage(k, p, size)) { k.err = IDK_ERR_BYTE; return; }
  u32 n = 0;
  while (n < size && sm->cur[n] != '.' && sm->cur[n] != ' ' && sm->cur[n] != ':') ++n;
  if (n == size) { k.err = IDK_ERR_NO_INSTRUMENT; return; }   // the reference would write its terminator over the first base here
  u32 nl = 0;                                                  // std::string(char*) stops at an embedded NUL (id.cpp:428)
  while (nl < n && sm->cur[nl] != 0) ++nl;
  if (nl + 1 > IDK_NAME - 1) { k.err = IDK_ERR_TOO_LONG; return; }
  // move-to-front code of the name (mtf.cpp:52-116): entries are [length, bytes]
  int code = -1;
  for (u32 base = 0; base < k.n_mtf && code < 0; base += FQ_WAVE) {
    const u32 ei = base + FQ_LANE;
    bool eq = false;
    if (ei < k.n_mtf) {
      const u8 *en = k.mtf + (u64)ei * IDK_NAME;
      eq = en[0] == nl;
      for (u32 j = 0; eq && j < nl; ++j) eq = en[1 + j] == sm->cur[j];
    }
#if FQ_WAVE > 1
    const u64 m = wave_ballot(eq);
    if (m) code = (int)(base + ctz64(m));
#else
    if (eq) code = (int)ei;
#endif
  }
  if (code < 0) {
    idk_fixed(k, 0, 11, 0);
    // the name and a terminating NUL through compress_lossless (id.cpp:441-446)
    FQ_SYNC();
    if (FQ_LANE == 0) { sm->cur[n] = 0; sm->cls[n] = 2; }
    FQ_SYNC();
    idk_lossless(k, n + 1);
  } else if (code < 2)
    idk_fixed(k, 0, 11, (u32)code + 1);
  else if (code < 256) {
    int kk = 0;
    while ((4 << kk) <= code) ++kk;          // code in [2 << kk, 4 << kk)
    idk_fixed(k, 0, 11, 3 + (u32)kk);
    idk_fixed(k, 1 + (u32)kk, 2u << kk, (u32)code - (2u << kk));
  } else {
    idk_fixed(k, 0, 11, 10);
    u32 c = (u32)code;
    for (u32 i = 0; i < 4; ++i) { idk_fixed(k, 8 + i, 256, c & 0xff); c >>= 8; }
  }
  if (k.err) return;
  // mtf.insert(name): to the front
  // (after the lossless call sm->cur may have become sm->prev: the name is kept aside first -- done below before the shift)
  const u8 *src = code < 0 ? sm->prev : sm->cur;   // idk_lossless copied cur to prev
  FQ_SYNC();
  for (u32 j = FQ_LANE; j < nl; j += FQ_WAVE) sm->name[1 + j] = src[j];
  if (FQ_LANE == 0) sm->name[0] = (u8)nl;
  FQ_SYNC();
  u32 pos = code < 0 ? k.n_mtf : (u32)code;
  if (code < 0) {
    if (k.n_mtf >= k.cfg->mtf_cap) { k.err = IDK_ERR_MTF_FULL; return; }
    k.n_mtf += 1;
  }
  if (pos > 0 || code < 0) {
    // entries [0, pos) move one place down (from the back), the name goes to the front; 8 bytes per lane and entry
    FQ_SYNC_MEM();
    for (u32 e = pos; e > 0; --e) {
      u64 *d = (u64 *)(k.mtf + (u64)e * IDK_NAME);
      const u64 *s = (const u64 *)(k.mtf + (u64)(e - 1) * IDK_NAME);
      for (u32 l = FQ_LANE; l < IDK_NAME / 8; l += FQ_WAVE) d[l] = s[l];
      FQ_SYNC_MEM();
    }
    for (u32 j = FQ_LANE; j < IDK_NAME; j += FQ_WAVE) k.mtf[j] = j <= nl ? sm->name[j] : 0;
    FQ_SYNC_MEM();
  }
}
FQ_DEV bool idk_typical_pe(const u8 *a, u32 na, const u8 *b, u32 nb) {   // id.cpp:241-254
  if (na != nb || na < 3) return false;
  bool eq = true;
  for (u32 base = 0; base < na - 2; base += FQ_WAVE) {
    const u32 i = base + FQ_LANE;
    eq = eq && !(i < na - 2 && a[i] != b[i]);
  }
  if (wave_any(!eq)) return false;
  return a[na - 2] == '1' && b[nb - 2] == '2';
}

// worker `tid` codes the ids of its reads of the block (CIdCompressor::Compress / CompressPE, application.cpp:634-640)
FQ_DEV void id_encode_body(const IdCfg &cfg, IdShared *sm, u32 tid, u32 n_reads, u32 paired) {
  const u64 T = cfg.T;
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;  // reads_block.h:197-214
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  IdK k;
  k.cfg = &cfg; k.sm = sm;
  k.e.low = 0; k.e.range = 0xff00000000000000ULL; k.e.len = 0; k.e.acc = 0; k.e.cap = cfg.out_cap; k.e.out = cfg.out + (u64)tid * cfg.out_cap; k.e.err = 0;
  k.small = cfg.small + (u64)tid * (cfg.small_mask + 1) * 2;
  k.big = cfg.big + (u64)tid * (cfg.big_mask + 1) * IDK_BIG_U64;
  k.fixed = cfg.fixed + (u64)tid * IDK_FIXED * IDK_BIG_U64;
  k.mtf = cfg.mtf + (u64)tid * cfg.mtf_cap * IDK_NAME;
  k.n_small = cfg.state[4 * tid]; k.n_big = cfg.state[4 * tid + 1]; k.n_mtf = cfg.state[4 * tid + 2];
  // ResetReadPrev, id.cpp:124-135
  k.cur_set = 0; k.n_tok[0] = k.n_tok[1] = 0; k.prev_size = 0;
  k.ctx_flags = 0; k.ctx_pe_flags = 0;
  k.err = 0;
  for (u64 i = first; i < last && !k.err && !k.e.err; i += paired ? 2 : 1) {
    const u8 *p1 = cfg.ids + cfg.off[i];
    const u32 n1 = (u32)(cfg.off[i + 1] - cfg.off[i]);
    if (!paired) {
      if (cfg.mode == 0) idk_id_lossless(k, p1, n1); else idk_id_instrument(k, p1, n1);
      continue;
    }
    const u8 *p2 = cfg.ids + cfg.off[i + 1];
    const u32 n2 = (u32)(cfg.off[i + 2] - cfg.off[i + 1]);
    if (cfg.mode == 0) {   // CompressPE, id.cpp:152-184
      const bool typical = idk_typical_pe(p1, n1, p2, n2);
      const u64 c0 = k.ctx_pe_flags;
      k.ctx_pe_flags = ((k.ctx_pe_flags << 1) + (typical ? 1 : 0)) & 0xff;
      idk_small(k, IDM_PE_FLAGS, c0, 2, typical ? 1 : 0);
      idk_id_lossless(k, p1, n1);
      if (!typical && !k.err) idk_id_lossless(k, p2, n2);
    } else {
      idk_id_instrument(k, p1, n1);
      if (!k.err) idk_id_instrument(k, p2, n2);
    }
  }
  for (int i = 0; i < 8; ++i) { q_put(k.e, (u8)(k.e.low >> 56)); k.e.low <<= 8; }  // End(), sub_rc.h:79-86
  q_flush(k.e);
  if (FQ_LANE == 0) {
    cfg.lens[tid] = k.e.len;
    cfg.state[4 * tid] = k.n_small; cfg.state[4 * tid + 1] = k.n_big; cfg.state[4 * tid + 2] = k.n_mtf;
    if (k.err) *cfg.err = k.err; else if (k.e.err) *cfg.err = IDK_ERR_OUT;
  }
}
