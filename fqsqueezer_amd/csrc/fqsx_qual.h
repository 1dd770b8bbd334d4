// fqsx_qual.h -- quality stream on the GPU (CQualityCompressor, quality.cpp:152-222).
// One wavefront per logical worker, no synchronisation points (workers are independent).  Context =
// position (<<48) + the last 2/6/9/10 quantised symbols; one adaptive order-0 model per context (96/8/4/2
// symbols, adder 1, halved at 2^15) kept in a per-worker open-addressed table in HBM whose slot is
// [key | packed u16 stats | u16 total].  Included by fqsx_api.hip.
#pragma once
#include "fqsx_plat.h"

struct QualCfg {
  u32 T, mode, n_sym, bits, nctx, slot_u64;   // slot_u64 = 1 + ceil((n_sym+1)/4)
  u64 ctx_mask;
  u64 *tab;            // [T][cap][slot_u64], key ~0 = empty
  u64 cap_mask;
  u32 *filled;         // [T]
  const u8 *quals;     // block input: concatenated quality strings
  const u64 *off;      // n_reads+1
  u8 *out;             // [T][out_cap]
  u64 out_cap;
  u64 *lens;           // [T]
  u32 *err;
  u8 fwd[96];          // quality_code_map_fwd
};

struct QEnc { u64 low, range, len, cap, acc; u8 *out; u32 err; };
FQ_DEV void q_put(QEnc &e, u8 b) {   // output bytes leave as aligned 8-byte words (the stream starts at offset 0 of an aligned buffer)
  e.acc |= (u64)b << (8 * (u32)(e.len & 7));
  ++e.len;
  if ((e.len & 7) == 0) {
    if (e.len <= e.cap) ((u64 *)e.out)[(e.len >> 3) - 1] = e.acc; else e.err = 1;
    e.acc = 0;
  }
}
FQ_DEV void q_flush(QEnc &e) {
  if (e.len > e.cap) { e.err = 1; return; }   // (a stream that ends inside the word beyond the buffer: q_put has not seen it)
  if ((e.len & 7) && e.len < e.cap) ((u64 *)e.out)[e.len >> 3] = e.acc;
}
FQ_DEV u64 q_div(u64 x, u32 d) {  // exact x / d for d < 2^16 (see div_u64_small in fqsx_dev.h)
  const double rd = 1.0 / (double)d;
  u32 hi = (u32)(x >> 32), lo = (u32)x;
  u32 qh = (u32)((double)hi * rd);
  u32 ph = qh * d;
  if (ph > hi) { --qh; ph -= d; } else if (hi - ph >= d) { ++qh; ph += d; }
  u64 rem = ((u64)(hi - ph) << 32) | lo;
  u64 q = (u64)((double)rem * rd);
  u64 prod = q * d;
  if (prod > rem) --q; else if (rem - prod >= d) ++q;
  return ((u64)qh << 32) + q;
}
// CRangeEncoder::EncodeFrequency, sub_rc.h:60-77; m = floor((2^64-1) / tot): mulhi(range, m) is range / tot or one less
FQ_DEV void q_encode_m(QEnc &e, u32 freq, u32 cum, u32 tot, u64 m) {
  const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
#ifndef FQSX_EMU
  u64 range = __umul64hi(e.range, m);
#else
  u64 range = (u64)(((unsigned __int128)e.range * m) >> 64);
#endif
  if ((u32)e.range - (u32)range * tot >= tot) ++range;   // (remainder < 2 * tot < 2^17: the low words decide)
  u64 low = e.low + range * cum;
  range *= freq;
  while (range <= Top) {
    if ((low ^ (low + range)) & M) range = (low | Top) - low;
    q_put(e, (u8)(low >> 56));
    low <<= 8;
    range <<= 8;
  }
  e.low = low;
  e.range = range;
}
FQ_DEV void q_encode(QEnc &e, u32 freq, u32 cum, u32 tot) { q_encode_m(e, freq, cum, tot, q_div(~0ull, tot)); }
FQ_DEV u64 q_hash(u64 h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
  return h;
}

// One chunk of <= 64 quality symbols of a read, one position per lane.  The context of a position is its position in
// the read plus the previous 2/6/9/10 quantised symbols (update_context, quality.cpp:209-215) -- a function of the input
// alone, and different for every position of a read, so the look-ups, the model arithmetic and the model updates of a
// chunk are independent of each other; only the range coder is sequential.  New contexts claim their table slot with
// a compare-and-swap (the layout is free: results do not depend on it).
// lds_q[i - lq0] = quantised symbol of position i (the staged window of the read starts at position lq0)
FQ_DEV void qual_chunk(const QualCfg &cfg, QEnc &e, u64 *tab, u32 &filled, const u8 *lds_q, u32 lq0, u32 pos0, u32 n, u32 *lds_f /*[3][64] emu only*/) {
  const u32 N = cfg.n_sym, W = cfg.slot_u64 - 1;
  u32 lane_f = 1, lane_c = 0, lane_t = 1;
  u64 lane_m = 0;
  bool bad = false;
  u32 n_new = 0;
  for (u32 t = FQ_LANE; t < n; t += FQ_WAVE) {
    const u32 i = pos0 + t, x = lds_q[i - lq0];
    // context of position i: reset_context then i x update_context (quality.cpp:204-215)
    u64 hist = cfg.ctx_mask;
    for (u32 j = i > cfg.nctx ? i - cfg.nctx : 0; j < i; ++j) hist = ((hist << cfg.bits) + lds_q[j - lq0]) & cfg.ctx_mask;
    const u64 ctx = ((u64)i << 48) + hist;
    // find_rc_context (quality.cpp:218-226): look the context up, create its model on first use
    u64 h = q_hash(ctx) & cfg.cap_mask;
    u64 *slot = nullptr;
    bool fresh = false;
    for (u64 it = 0; it <= cfg.cap_mask; ++it) {
      u64 *p = tab + h * cfg.slot_u64;
      u64 k = p[0];
      if (k == ~0ull) {   // claim the empty slot (another lane's new context may get there first)
        k = atomic_cas64(&p[0], ~0ull, ctx);
        if (k == ~0ull) { slot = p; fresh = true; break; }
      }
      if (k == ctx) { slot = p; break; }
      h = (h + 1) & cfg.cap_mask;
    }
    if (!slot) { bad = true; continue; }
    n_new += fresh ? 1u : 0u;
    // the model: N 16-bit statistics and their total, packed four to a word (new model: all 1, total N; rc.h:69-74)
    u32 freq = 1, cum = 0, tot = N;
    const u32 xw = x >> 2, xs = 16 * (x & 3), tw = N >> 2, ts = 16 * (N & 3);
    u64 wx = 0, wt = 0;   // the words holding stats[x] and the total
    if (fresh) {
      cum = x;
    } else {
      // the model's words eight at a time (independent loads: one round trip per eight words, the same trip count in
      // every lane), summing the words below the symbol's
      for (u32 g0 = 0; g0 < W; g0 += 8) {
        u64 v[8];
#pragma unroll
        for (u32 k = 0; k < 8; ++k) v[k] = g0 + k < W ? slot[1 + g0 + k] : 0;
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
          const u32 wv = g0 + k;
          if (wv < xw) cum += (u32)((v[k] * 0x0001000100010001ULL) >> 48);   // sum of the word's four fields (< 2^16: the total is < 2^15 + 1)
          if (wv == xw) wx = v[k];
          if (wv == tw) wt = v[k];
        }
      }
      for (u32 f = 0; f < (x & 3); ++f) cum += (u32)((wx >> (16 * f)) & 0xffff);
      freq = (u32)((wx >> xs) & 0xffff);
      tot = (u32)((wt >> ts) & 0xffff);
    }
#if FQ_WAVE > 1
    lane_f = freq; lane_c = cum; lane_t = tot; lane_m = q_div(~0ull, tot);
#else
    lds_f[t] = freq; lds_f[64 + t] = cum; lds_f[128 + t] = tot;
#endif
    // update (rc.h:41-55): stats[x] += 1, total += 1, halve everything when the total reaches 2^15
    const u32 ntot = tot + 1;
    if (ntot >= (1u << 15) || fresh) {   // rewrite the whole model
      u32 sum = 0;
      u64 nv_t = 0;   // the word that also holds the total: stored last
      for (u32 wv = 0; wv < W; ++wv) {
        u64 v = fresh ? 0 : slot[1 + wv], nv = 0;
        for (u32 f = 0; f < 4; ++f) {
          const u32 fi = wv * 4 + f;
          if (fi >= N) continue;
          u32 sv = fresh ? 1u : (u32)((v >> (16 * f)) & 0xffff);
          if (fi == x) sv += 1;
          if (ntot >= (1u << 15)) sv = (sv + 1) / 2;
          sum += sv;
          nv |= (u64)sv << (16 * f);
        }
        if (wv == tw) nv_t = nv; else slot[1 + wv] = nv;
      }
      // (one halving always suffices: the sum of N <= 96 rounded-up halves of a total of 2^15 is below 2^15)
      const u32 nt = ntot >= (1u << 15) ? sum : ntot;
      slot[1 + tw] = nv_t | ((u64)nt << ts);
    } else if (tw == xw) {
      slot[1 + xw] = ((wx + (1ull << xs)) & ~(0xffffull << ts)) | ((u64)ntot << ts);
    } else {
      slot[1 + xw] = wx + (1ull << xs);
      slot[1 + tw] = (wt & ~(0xffffull << ts)) | ((u64)ntot << ts);
    }
  }
  if (wave_any(bad)) { e.err = 2; return; }
  filled += wave_sum32(n_new);
  if ((u64)filled * 10 >= (cfg.cap_mask + 1) * 9) { e.err = 2; return; }   // (the host sizes the table for every symbol of the block)
  // the range coder, in position order
#if FQ_WAVE > 1
  for (u32 t = 0; t < n; ++t) {
    const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(lane_m >> 32), t) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)lane_m, t);
    q_encode_m(e, (u32)__builtin_amdgcn_readlane((int)lane_f, t), (u32)__builtin_amdgcn_readlane((int)lane_c, t),
               (u32)__builtin_amdgcn_readlane((int)lane_t, t), m);
  }
#else
  for (u32 t = 0; t < n; ++t) q_encode(e, lds_f[t], lds_f[64 + t], lds_f[128 + t]);
#endif
  FQ_SYNC_MEM();   // the models written here are read by the next chunks (same wave, other lanes)
}

// worker `tid` codes the qualities of its reads of the block (application.cpp:641, quality.cpp:152-175)
FQ_DEV void qual_encode_body(const QualCfg &cfg, u8 *lds_q /*[4096 + 96]*/, u32 tid, u32 n_reads) {
  const u64 T = cfg.T;
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;  // reads_block.h:197-214
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  QEnc e;
  e.low = 0; e.range = 0xff00000000000000ULL; e.len = 0; e.acc = 0; e.cap = cfg.out_cap; e.out = cfg.out + (u64)tid * cfg.out_cap; e.err = 0;
  u64 *tab = cfg.tab + (u64)tid * (cfg.cap_mask + 1) * cfg.slot_u64;
  u32 filled = cfg.filled[tid];
  u8 *lds_fwd = lds_q + 4096;   // quality_code_map_fwd next to the staged symbols
  for (u32 i = FQ_LANE; i < 96; i += FQ_WAVE) lds_fwd[i] = cfg.fwd[i];
  FQ_SYNC();
#if FQ_WAVE > 1
  u32 *lds_f = nullptr;
#else
  static thread_local u32 lds_f_[192];
  u32 *lds_f = lds_f_;
#endif
  for (u64 r = first; r < last && !e.err; ++r) {
    const u8 *q = cfg.quals + cfg.off[r];
    const u32 size = (u32)(cfg.off[r + 1] - cfg.off[r]);
    // the read's symbols are staged in LDS in windows of 4032 positions plus the 64 before them (contexts reach back)
    for (u32 w0 = 0; w0 < size && !e.err; w0 += 4032) {
      const u32 lq0 = w0 ? w0 - 64 : 0, wend = size - w0 < 4032 ? size : w0 + 4032;
      FQ_SYNC();
      for (u32 i = lq0 + FQ_LANE; i < wend; i += FQ_WAVE) lds_q[i - lq0] = lds_fwd[(u8)(q[i] - 33) < 96 ? (u8)(q[i] - 33) : 95];
      FQ_SYNC();
      for (u32 pos0 = w0; pos0 < wend && !e.err; pos0 += 64) qual_chunk(cfg, e, tab, filled, lds_q, lq0, pos0, wend - pos0 < 64 ? wend - pos0 : 64, lds_f);
    }
  }
  for (int i = 0; i < 8; ++i) { q_put(e, (u8)(e.low >> 56)); e.low <<= 8; }  // End(), sub_rc.h:79-86
  q_flush(e);
  if (FQ_LANE == 0) {
    cfg.lens[tid] = e.len;
    cfg.filled[tid] = filled;
    if (e.err) *cfg.err = e.err;
  }
}
