// fqsx_sort.h -- read ordering of `fqs e -om s` on the GPU.
//
// The reference bins reads by their first four bases and std::sort's every bin with a string comparator
// (preprocess_se, application.cpp:349-412; CSortedFASTQFile::sort_reads, io.h:499-528): N->T sequence, then
// length, then raw bytes.  std::sort is unstable, so the order of reads that compare equal is a property of the
// libstdc++ algorithm -- and it matters, because ids and qualities travel with the reads.  Design: the GPU does the
// string work -- an LSD radix sort over the comparator's keys (stable partition passes of 8-bit digits) followed
// by a dense RANK per read (equal reads share a rank) -- and the host replays std::sort per bin on the ranks, in
// the bin's input order: the comparator outcomes are the same, hence so is every move of the algorithm.
#pragma once
#include "fqsx_plat.h"

#define FQSX_SORT_TILE 2048u
enum { SORT_NT = 0, SORT_LEN = 1, SORT_RAW = 2 };

struct SortCfg {
  const u8 *bases;
  const u64 *off;
  u32 n, n_tiles;
  u32 *perm_in, *perm_out;
  u32 *tile_hist;   // [n_tiles][256]
  u32 *dig_tot;     // [256]
  u32 *dig_off;     // [257]
  u32 *flags;       // [n]: sorted position i starts a new group of equal reads
  u32 *rank;        // [n]: dense rank of read r
  u32 *info;        // [n_tiles][10]: min len, max len, 256-bit set of bytes that are not A/C/G
  u32 kind, arg, raw_bits;
  u8 raw_code[256];
};

FQ_DEV u32 sort_nt(u8 c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }  // dna_convert_NT, io.h:552-562
FQ_DEV u32 sort_digit(const SortCfg &c, u32 r) {
  const u64 b = c.off[r];
  const u32 len = (u32)(c.off[r + 1] - b);
  if (c.kind == SORT_LEN) return (len >> (8 * c.arg)) & 255u;
  u32 d = 0;
  if (c.kind == SORT_NT) {
    // a missing position sorts as 'A' (0): with the length as the next key this is the comparator's
    // "equal on the common prefix: shorter first"
    for (u32 k = 0; k < 4; ++k) {
      const u32 pos = 4 * c.arg + k;
      d = (d << 2) | (pos < len ? sort_nt(c.bases[b + pos]) : 0u);
    }
    return d;
  }
  const u32 per = 8 / c.raw_bits;   // raw bytes decide only between reads equal so far: same length, same N->T codes
  for (u32 k = 0; k < per; ++k) {
    const u32 pos = per * c.arg + k;
    d = (d << c.raw_bits) | (pos < len ? (u32)c.raw_code[c.bases[b + pos]] : 0u);
  }
  return d;
}

// per tile of reads: length range and the set of bytes outside A/C/G
FQ_DEV void sort_info_body(const SortCfg &c, u32 blk, u32 *lds /*[64][10]*/) {
  const u32 lo = blk * FQSX_SORT_TILE, hi = c.n < lo + FQSX_SORT_TILE ? c.n : lo + FQSX_SORT_TILE;
  for (u32 l = FQ_LANE; l < 64; l += FQ_WAVE) {
    u32 mn = 0xffffffffu, mx = 0, set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 r = lo + l; r < hi; r += 64) {
      const u64 b = c.off[r];
      const u32 len = (u32)(c.off[r + 1] - b);
      mn = len < mn ? len : mn;
      mx = len > mx ? len : mx;
      for (u32 i = 0; i < len; ++i) {
        const u8 ch = c.bases[b + i];
        if (ch != 'A' && ch != 'C' && ch != 'G') set[ch >> 5] |= 1u << (ch & 31);
      }
    }
    lds[l * 10 + 0] = mn; lds[l * 10 + 1] = mx;
    for (u32 k = 0; k < 8; ++k) lds[l * 10 + 2 + k] = set[k];
  }
  FQ_SYNC();
  if (FQ_LANE == 0) {
    u32 mn = 0xffffffffu, mx = 0, set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 l = 0; l < 64; ++l) {
      mn = lds[l * 10] < mn ? lds[l * 10] : mn;
      mx = lds[l * 10 + 1] > mx ? lds[l * 10 + 1] : mx;
      for (u32 k = 0; k < 8; ++k) set[k] |= lds[l * 10 + 2 + k];
    }
    u32 *o = c.info + (u64)blk * 10;
    o[0] = mn; o[1] = mx;
    for (u32 k = 0; k < 8; ++k) o[2 + k] = set[k];
  }
}
FQ_DEV void sort_iota_body(const SortCfg &c, u32 blk) {
  const u32 lo = blk * FQSX_SORT_TILE, hi = c.n < lo + FQSX_SORT_TILE ? c.n : lo + FQSX_SORT_TILE;
  for (u32 e = lo + FQ_LANE; e < hi; e += FQ_WAVE) c.perm_in[e] = e;
}
// one radix pass = tile histograms -> per-digit scan over tiles -> digit offsets -> stable scatter
FQ_DEV void sort_count_body(const SortCfg &c, u32 blk, u32 *hist /*LDS[256]*/) {
  const u32 lo = blk * FQSX_SORT_TILE, hi = c.n < lo + FQSX_SORT_TILE ? c.n : lo + FQSX_SORT_TILE;
  for (u32 d = FQ_LANE; d < 256; d += FQ_WAVE) hist[d] = 0;
  FQ_SYNC();
  for (u32 e = lo + FQ_LANE; e < hi; e += FQ_WAVE) lds_inc32(&hist[sort_digit(c, c.perm_in[e])]);
  FQ_SYNC();
  for (u32 d = FQ_LANE; d < 256; d += FQ_WAVE) c.tile_hist[(u64)blk * 256 + d] = hist[d];
}
FQ_DEV void sort_scan_body(const SortCfg &c, u32 d) {
  u32 run = 0;
  for (u32 base = 0; base < c.n_tiles; base += FQ_WAVE) {
    const u32 i = base + FQ_LANE;
    const u32 v = i < c.n_tiles ? c.tile_hist[(u64)i * 256 + d] : 0;
    const u32 ex = wave_excl_scan32(v) + run;
    if (i < c.n_tiles) c.tile_hist[(u64)i * 256 + d] = ex;
    run += wave_sum32(v);
  }
  if (FQ_LANE == 0) c.dig_tot[d] = run;
}
FQ_DEV void sort_dstoff_body(const SortCfg &c) {
  if (FQ_LANE == 0) {
    u32 run = 0;
    for (u32 d = 0; d < 256; ++d) { c.dig_off[d] = run; run += c.dig_tot[d]; }
    c.dig_off[256] = run;
  }
}
FQ_DEV void sort_scatter_body(const SortCfg &c, u32 blk, u32 *cursor /*LDS[256]*/, u32 *ld /*LDS[64]*/) {
  const u32 lo = blk * FQSX_SORT_TILE, hi = c.n < lo + FQSX_SORT_TILE ? c.n : lo + FQSX_SORT_TILE;
  if (lo >= hi) return;
  for (u32 d = FQ_LANE; d < 256; d += FQ_WAVE) cursor[d] = c.dig_off[d] + c.tile_hist[(u64)blk * 256 + d];
  FQ_SYNC();
  for (u32 base = lo; base < hi; base += FQ_WAVE) {
    const u32 e = base + FQ_LANE, cnt = hi - base < FQ_WAVE ? hi - base : FQ_WAVE;
    u32 r = 0, d = 0xffffffffu;
    if (e < hi) {
      r = c.perm_in[e];
      d = sort_digit(c, r);
    }
    ld[FQ_LANE] = d;
    FQ_SYNC();
    u32 rank = 0, later = 0;
    for (u32 q = 0; q < cnt; ++q) {
      const u32 dq = ld[q];
      rank += (q < FQ_LANE && dq == d) ? 1u : 0u;
      later += (q > FQ_LANE && dq == d) ? 1u : 0u;
    }
    const u32 cur = e < hi ? cursor[d] : 0;
    FQ_SYNC();
    if (e < hi) {
      c.perm_out[cur + rank] = r;
      if (later == 0) cursor[d] = cur + rank + 1;   // the last entry of this digit in the round advances the cursor
    }
    FQ_SYNC();
  }
}
// sorted position i opens a new group iff its read differs from the one before it (length or any raw byte)
FQ_DEV void sort_flags_body(const SortCfg &c, u32 blk) {
  const u32 lo = blk * FQSX_SORT_TILE, hi = c.n < lo + FQSX_SORT_TILE ? c.n : lo + FQSX_SORT_TILE;
  for (u32 e = lo + FQ_LANE; e < hi; e += FQ_WAVE) {
    u32 f = 0;
    if (e > 0) {
      const u32 x = c.perm_in[e - 1], y = c.perm_in[e];
      const u64 bx = c.off[x], by = c.off[y];
      const u32 lx = (u32)(c.off[x + 1] - bx), ly = (u32)(c.off[y + 1] - by);
      f = lx != ly;
      for (u32 i = 0; i < lx && !f; ++i) f = c.bases[bx + i] != c.bases[by + i];
    }
    c.flags[e] = f;
  }
}
// rank[read] = number of group starts at or before its sorted position (one wave walks the array)
FQ_DEV void sort_rank_body(const SortCfg &c) {
  u32 run = 0;
  for (u32 base = 0; base < c.n; base += FQ_WAVE) {
    const u32 i = base + FQ_LANE;
    const u32 v = i < c.n ? c.flags[i] : 0;
    const u32 inc = wave_excl_scan32(v) + v + run;
    if (i < c.n) c.rank[c.perm_in[i]] = inc;
    run += wave_sum32(v);
  }
}
