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

struct QEnc { u64 low, range, len, cap; u8 *out; u32 err; };
FQ_DEV void q_put(QEnc &e, u8 b) {
  if (e.len < e.cap) e.out[e.len] = b; else e.err = 1;
  ++e.len;
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
FQ_DEV void q_encode(QEnc &e, u32 freq, u32 cum, u32 tot) {  // CRangeEncoder::EncodeFrequency, sub_rc.h:60-77
  const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
  u64 range = q_div(e.range, tot), low = e.low + range * cum;
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
FQ_DEV u64 q_hash(u64 h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
  return h;
}

// worker `tid` codes the qualities of its reads of the block (application.cpp:641, quality.cpp:152-175)
FQ_DEV void qual_encode_body(const QualCfg &cfg, u8 *lds_q /*[4096]*/, u32 tid, u32 n_reads) {
  const u64 T = cfg.T;
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;  // reads_block.h:197-214
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  QEnc e;
  e.low = 0; e.range = 0xff00000000000000ULL; e.len = 0; e.cap = cfg.out_cap; e.out = cfg.out + (u64)tid * cfg.out_cap; e.err = 0;
  u64 *tab = cfg.tab + (u64)tid * (cfg.cap_mask + 1) * cfg.slot_u64;
  const u32 N = cfg.n_sym, W = cfg.slot_u64 - 1;  // W words of packed 16-bit stats (+ total right after the last stat)
  u32 filled = cfg.filled[tid];
  for (u64 r = first; r < last && !e.err; ++r) {
    const u8 *q = cfg.quals + cfg.off[r];
    const u32 size = (u32)(cfg.off[r + 1] - cfg.off[r]);
    u64 ctx = cfg.ctx_mask;  // reset_context, quality.cpp:204-207
    for (u32 base = 0; base < size; base += 4096) {
      const u32 n = size - base < 4096 ? size - base : 4096;
      FQ_SYNC();
      for (u32 i = FQ_LANE; i < n; i += FQ_WAVE) lds_q[i] = cfg.fwd[(u8)(q[base + i] - 33) < 96 ? (u8)(q[base + i] - 33) : 95];
      FQ_SYNC();
      for (u32 i = 0; i < n && !e.err; ++i) {
        const u32 x = lds_q[i];
        // find_rc_context (quality.cpp:218-226): look the context up, create its model on first use
        u64 h = q_hash(ctx) & cfg.cap_mask;
        u64 *slot = nullptr;
        for (u64 it = 0; it <= cfg.cap_mask; ++it) {
          u64 *p = tab + h * cfg.slot_u64;
          const u64 k = p[0];
          if (k == ctx) { slot = p; break; }
          if (k == ~0ull) {
            if ((u64)(filled + 1) * 10 >= (cfg.cap_mask + 1) * 9) { e.err = 2; break; }
            // new model: all stats 1, total N (rc.h:69-74)
            for (u32 wv = FQ_LANE; wv < W; wv += FQ_WAVE) {
              u64 v = 0;
              for (u32 f = 0; f < 4; ++f) {
                u32 fi = wv * 4 + f;
                v |= (u64)(fi < N ? 1u : fi == N ? N : 0u) << (16 * f);
              }
              p[1 + wv] = v;
            }
            FQ_SYNC_MEM();
            p[0] = ctx;
            ++filled;
            slot = p;
            break;
          }
          h = (h + 1) & cfg.cap_mask;
        }
        if (!slot) { if (!e.err) e.err = 2; break; }
        u32 freq, cum, tot;
        if (N <= 8) {  // wave-uniform: the whole model is <= 3 words
          u64 wd[3] = {slot[1], W > 1 ? slot[2] : 0, W > 2 ? slot[3] : 0};
          u32 st[9];
          for (u32 f = 0; f <= N; ++f) st[f] = (u32)((wd[f >> 2] >> (16 * (f & 3))) & 0xffff);
          cum = 0;
          for (u32 f = 0; f < x; ++f) cum += st[f];
          freq = st[x];
          tot = st[N];
          q_encode(e, freq, cum, tot);
          st[x] += 1;
          tot += 1;
          while (tot >= (1u << 15)) {  // rescale, rc.h:28-39
            tot = 0;
            for (u32 f = 0; f < N; ++f) { st[f] = (st[f] + 1) / 2; tot += st[f]; }
          }
          st[N] = tot;
          wd[0] = wd[1] = wd[2] = 0;
          for (u32 f = 0; f <= N; ++f) wd[f >> 2] |= (u64)st[f] << (16 * (f & 3));
          slot[1] = wd[0];
          if (W > 1) slot[2] = wd[1];
          if (W > 2) slot[3] = wd[2];
        } else {  // 96 symbols: one word (4 stats) per lane
          u64 mine = 0;
          u32 part = 0;
          FQ_SYNC_MEM();
          for (u32 wv = FQ_LANE; wv < W; wv += FQ_WAVE) {
            const u64 v = slot[1 + wv];
#if FQ_WAVE > 1
            mine = v;
#endif
            for (u32 f = 0; f < 4; ++f) {
              u32 fi = wv * 4 + f;
              if (fi < x) part += (u32)((v >> (16 * f)) & 0xffff);
            }
          }
          cum = wave_sum32(part);
          freq = (u32)((slot[1 + (x >> 2)] >> (16 * (x & 3))) & 0xffff);
          tot = (u32)((slot[1 + (N >> 2)] >> (16 * (N & 3))) & 0xffff);
          q_encode(e, freq, cum, tot);
          tot += 1;
          FQ_SYNC_MEM();
          slot[1 + (x >> 2)] += 1ull << (16 * (x & 3));
          FQ_SYNC_MEM();
          while (tot >= (1u << 15)) {
            u32 pp = 0;
            for (u32 wv = FQ_LANE; wv < W; wv += FQ_WAVE) {
              u64 v = slot[1 + wv], nv = 0;
              for (u32 f = 0; f < 4; ++f) {
                u32 fi = wv * 4 + f;
                u32 sv = (u32)((v >> (16 * f)) & 0xffff);
                if (fi < N) { sv = (sv + 1) / 2; pp += sv; }
                nv |= (u64)sv << (16 * f);
              }
              slot[1 + wv] = nv;
            }
            tot = wave_sum32(pp);
            FQ_SYNC_MEM();
          }
          {
            u64 v = slot[1 + (N >> 2)];
            v = (v & ~(0xffffull << (16 * (N & 3)))) | ((u64)tot << (16 * (N & 3)));
            slot[1 + (N >> 2)] = v;
          }
          (void)mine;
        }
        // update_context, quality.cpp:209-215
        const u64 my = ctx + (1ull << 48), t = (ctx << cfg.bits) + x;
        ctx = (my & ~cfg.ctx_mask) + (t & cfg.ctx_mask);
      }
    }
  }
  for (int i = 0; i < 8; ++i) { q_put(e, (u8)(e.low >> 56)); e.low <<= 8; }  // End(), sub_rc.h:79-86
  if (FQ_LANE == 0) {
    cfg.lens[tid] = e.len;
    cfg.filled[tid] = filled;
    if (e.err) *cfg.err = e.err;
  }
}
