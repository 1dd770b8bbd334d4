// oracle/fqs_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see fqs_oracle.h).
//
// CPU restatement of the FQSqueezer 1.1 DNA-stream encoder for single-end data
// (-om o and -om s).  Every function cites the reference lines it follows
// (paths relative to /root/reference/fqs).  The data structures are NOT the
// reference's: the k-mer tables are flat open-addressed tables keyed by the
// k-mer kernel and the context maps are flat exact-match maps.  This is legal
// because (a) a cluster scan returns the counts of the 4 sibling k-mers, which
// does not depend on slot layout, and (b) RNG draws depend only on count
// values and on the order of operations, both of which are preserved.
//
// Parity: PINNED against DNA streams of the compiled reference (tests/golden).
#include "fqs_oracle.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// ---------------------------------------------------------------------------------------
// std::mt19937 (ISO C++ [rand.eng.mers]); the reference seeds every instance with 5481
// (utils.h:296).
struct Mt19937 {
  u32 s[624];
  int idx;
  void seed(u32 v) {
    s[0] = v;
    for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (u32)i;
    idx = 624;
  }
  void twist() {
    for (int i = 0; i < 624; ++i) {
      u32 y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
      u32 v = s[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      s[i] = v;
    }
    idx = 0;
  }
  u32 next() {
    if (idx >= 624) twist();
    u32 y = s[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};

// ---------------------------------------------------------------------------------------
// CCounterIncrementer, utils.h:256-335
struct CounterInc {
  Mt19937 mt;
  u64 draws = 0;   // random numbers consumed (counters above the threshold)
  u32 thr, mult, maxv;
  std::vector<u32> map;
  void reset(u32 thr_, u32 mult_, u32 maxv_) {  // utils.h:294-312
    mt.seed(5481);
    maxv = maxv_;
    thr = std::min(thr_, maxv_);
    mult = mult_;
    map.assign((size_t)maxv + 2, 0);
    for (u32 i = 0; i <= thr; ++i) map[i] = i;
    u32 inc = mult;
    for (u32 i = thr + 1; i <= maxv; ++i, inc += mult) map[i] = map[i - 1] + inc;
    map[(size_t)maxv + 1] = map[maxv];
  }
  u32 inc1(u32 c) {  // utils.h:314-325
    if (c <= thr) return c + 1;
    ++draws;
    return (mt.next() % (mult * (c - thr)) == 0) ? c + 1 : c;
  }
  u32 decode(u32 v) const {  // utils.h:264-270
    if (v <= thr) return v;
    return (map[v] + map[(size_t)v + 1]) / 2;
  }
  u32 encode(u32 real) {  // utils.h:272-290
    if (real <= thr) return real;
    u32 end_dist = std::min(maxv, real) + 1;
    // last position in [thr, end_dist) whose mapped value is <= real
    u32 pos = (u32)(std::upper_bound(map.begin() + thr, map.begin() + end_dist, real) - 1 - map.begin());
    if (pos >= maxv) return maxv;
    u32 rest = real - map[pos];
    ++draws;
    if (mt.next() % (map[(size_t)pos + 1] - map[pos]) < rest) ++pos;
    return pos;
  }
  u32 merge(u32 a, u32 b) { return encode(decode(a) + decode(b)); }  // utils.h:327-333
};

// ---------------------------------------------------------------------------------------
// CKmer (canonical variant), kmer.h:18-540.  Symbols are left-aligned 2-bit codes.
static inline u64 comp2(u64 s) { return 3 - s; }  // utils.h:77-89

struct Kmer {
  u64 dir, rc;
  u32 cur, maxs;
  u64 mask, kernel_mask;
  u32 shift;
  void setup(u32 k) {  // kmer.h:279-298
    maxs = k;
    dir = rc = 0;
    cur = 0;
    shift = 64 - 2 * k;
    mask = (~0ull) << shift;
    kernel_mask = ((1ull << (2 * k - 8)) - 1ull) << (64 - 2 * k + 4);
  }
  void reset() { dir = rc = 0; cur = 0; }  // kmer.h:236-240
  void insert(u64 sym) {                   // kmer.h:80-96
    rc >>= 2;
    rc += comp2(sym) << 62;
    rc &= mask;
    if (cur == maxs) {
      dir <<= 2;
      dir += sym << shift;
    } else {
      ++cur;
      dir += sym << (64 - 2 * cur);
    }
  }
  void insert_zero() {  // kmer.h:99-115
    rc >>= 2;
    rc += 3ull << 62;
    rc &= mask;
    if (cur == maxs) dir <<= 2; else ++cur;
  }
  void insert_front(u64 sym) {  // kmer.h:139-150
    if (cur < maxs) {
      dir >>= 2;
      dir += sym << 62;
      rc += comp2(sym) << (62 - 2 * cur);
      ++cur;
    }
  }
  void replace(u64 sym, u32 pos) {  // kmer.h:153-160,172-178
    u32 sh = 62 - 2 * pos;
    dir &= ~(3ull << sh);
    dir += sym << sh;
    sh = 64 - 2 * cur + 2 * pos;
    rc &= ~(3ull << sh);
    rc += comp2(sym) << sh;
  }
  void replace_last(u64 sym) {  // kmer.h:163-169,181-185
    u32 sh = 64 - 2 * cur;
    dir &= ~(3ull << sh);
    dir += sym << sh;
    rc <<= 2;
    rc >>= 2;
    rc += comp2(sym) << 62;
  }
  bool norm_dir() const { return (dir & kernel_mask) < (rc & kernel_mask); }  // kmer.h:380-385
  u64 norm() const { return norm_dir() ? dir : rc; }                          // kmer.h:366-377
  u64 aligned_dir() const { return cur ? dir >> (64 - 2 * cur) : 0; }         // kmer.h:398-400 (quirk 18)
  u64 aligned_rc() const { return cur ? rc >> (64 - 2 * cur) : 0; }           // kmer.h:403-405
  u64 symbol(u32 pos) const { return (dir >> (62 - 2 * pos)) & 3; }           // kmer.h:470-483
  bool full() const { return cur == maxs; }
  bool almost_full(u32 margin) const { return cur + margin >= maxs; }  // kmer.h:523-526
};

// ---------------------------------------------------------------------------------------
// CHT_kmer<T> semantics (ht_kmer.h:29-554) on a flat table: exact set of normalised
// k-mers with small probabilistic counters; look-up = counts of the 4 siblings that
// differ in the last symbol (direct orientation) or first symbol (rc orientation).
static inline u64 murmur64(u64 h) {  // ht_kmer.h:123-127, context_hm.h:81-85
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}

struct Counters {
  u64 probes = 0, slots = 0, inserts = 0, siv_words = 0, ctx = 0, coded = 0, lprobes = 0, linserts = 0;
  u64 lv[6] = {0, 0, 0, 0, 0, 0};   // find_counts results per counts_level_t (none, pmer, smer, bmer, mixed, bmer_unc)
};

struct KTable {
  u32 k, cbits;
  u64 cmask;
  std::vector<u64> slot;  // (kmer right-aligned) << cbits | count ; 0 = empty
  u64 hmask, filled;
  u64 *n_probe, *n_slot;
  void init(u32 k_, u32 cbits_, u64 cap, u64 *np, u64 *ns) {
    k = k_; cbits = cbits_; cmask = (1ull << cbits) - 1;
    slot.assign(cap, 0); hmask = cap - 1; filled = 0;
    n_probe = np; n_slot = ns;
  }
  void clear() { std::fill(slot.begin(), slot.end(), 0); filled = 0; }  // ht_kmer.h:412-416
  u64 home(u64 v) const {  // hash on the kernel (symbols 2..k-3), cf. ht_kmer.h:115-130
    u64 kern = (v >> 4) & ((1ull << (2 * k - 8)) - 1);
    return murmur64(kern) & hmask;
  }
  void grow() {
    std::vector<u64> old;
    old.swap(slot);
    slot.assign(old.size() * 2, 0);
    hmask = slot.size() - 1;
    for (u64 it : old)
      if (it) {
        u64 p = home(it >> cbits);
        while (slot[p]) p = (p + 1) & hmask;
        slot[p] = it;
      }
  }
  // _update_counts_full, ht_kmer.h:205-263
  void add_counts(u64 kmer_norm, bool is_dir, u32 counts[4]) const {
    u64 v = kmer_norm >> (64 - 2 * k);
    u64 p = home(v);
    ++*n_probe;
    if (is_dir) {
      u64 grp = v >> 2;
      for (;; p = (p + 1) & hmask) {
        ++*n_slot;
        u64 it = slot[p];
        if (!it) break;
        u64 iv = it >> cbits;
        if ((iv >> 2) == grp) counts[iv & 3] += (u32)(it & cmask);
      }
    } else {
      u64 lowmask = (1ull << (2 * k - 2)) - 1;
      u64 grp = v & lowmask;
      for (;; p = (p + 1) & hmask) {
        ++*n_slot;
        u64 it = slot[p];
        if (!it) break;
        u64 iv = it >> cbits;
        if ((iv & lowmask) == grp) counts[3 - (iv >> (2 * k - 2))] += (u32)(it & cmask);
      }
    }
  }
  static bool non_empty(const u32 c[4]) { return c[0] || c[1] || c[2] || c[3]; }
  // find / find_full / find_partial, ht_kmer.h:189-203,266-327,504-510
  bool find(const Kmer &km, u32 counts[4], CounterInc &cinc) const {
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (km.full()) {
      add_counts(km.norm(), km.norm_dir(), counts);
      return non_empty(counts);
    }
    Kmer t = km;
    u32 missing = km.maxs - km.cur, trials = 1;
    for (u32 i = 0; i < missing; ++i) { t.insert_front(0); trials *= 4; }
    for (u32 i = 0; i < trials; ++i) {
      if (i) {  // odometer over the padded front symbols, ht_kmer.h:291-310
        u32 j = 0;
        for (;; ++j) {
          u64 c = t.symbol(j);
          if (c < 3) { t.replace(c + 1, j); break; }
        }
        for (u32 q = 0; q < j; ++q) t.replace(0, q);
      }
      u32 loc[4] = {0, 0, 0, 0};
      add_counts(t.norm(), t.norm_dir(), loc);
      for (int s = 0; s < 4; ++s)
        if (loc[s]) counts[s] = cinc.merge(counts[s], loc[s]);  // ht_kmer.h:321-323
    }
    return non_empty(counts);
  }
  // _find(kmer,false) + count(), ht_kmer.h:330-362,441-453
  u32 count(u64 kmer_norm) const {
    u64 v = kmer_norm >> (64 - 2 * k);
    ++*n_probe;
    for (u64 p = home(v);; p = (p + 1) & hmask) {
      ++*n_slot;
      u64 it = slot[p];
      if (!it) return 0;
      if ((it >> cbits) == v) return (u32)(it & cmask);
    }
  }
  // insert(kmer,cinc), ht_kmer.h:420-438 (growth is layout-only, :88-112)
  void insert(u64 kmer_norm, CounterInc &cinc) {
    if ((filled + 1) * 2 > slot.size()) grow();
    u64 v = kmer_norm >> (64 - 2 * k);
    u64 p = home(v);
    for (;; p = (p + 1) & hmask) {
      u64 it = slot[p];
      if (!it) break;
      if ((it >> cbits) == v) {
        u32 cnt = (u32)(it & cmask);
        if (cnt < cmask && cinc.inc1(cnt) != cnt) slot[p] = it + 1;
        return;
      }
    }
    // new item: count 0 -> Increment(0) == 1 (thr >= 0), ht_kmer.h:353-358,434-436
    slot[p] = (v << cbits) | cinc.inc1(0);
    ++filled;
  }
};

// ---------------------------------------------------------------------------------------
// TSmallIntVector<2>, bit_vec.h:17-231
struct Siv {
  u32 key_bits;
  std::vector<u64> w;
  u64 no_updates = 0, no_filled = 0;
  u64 *n_words;
  void init(u32 key_bits_, u64 *nw) { key_bits = key_bits_; w.assign((1ull << key_bits) / 32, 0); n_words = nw; }
  bool increment(u64 idx) {  // bit_vec.h:53-67
    u64 &d = w[idx >> 5];
    u32 sh = 2 * (idx & 31);
    ++*n_words;
    u64 f = (d >> sh) & 3;
    if (f == 3) return false;
    d += 1ull << sh;
    return f == 0;
  }
  u64 test(u64 idx) const { return (w[idx >> 5] >> (2 * (idx & 31))) & 3; }  // bit_vec.h:69-81
  void counts(u64 idx, u32 c[4], bool add) const {                          // bit_vec.h:83-111
    u64 d = w[idx >> 5];
    u32 sh = 2 * ((idx & 31) & ~3ull);
    ++*n_words;
    for (int i = 0; i < 4; ++i, sh += 2) {
      u32 v = (u32)((d >> sh) & 3);
      if (add) c[i] += v; else c[i] = v;
    }
  }
  u64 range_sum(u64 idx, u64 size_bits) const {  // test_shorter, bit_vec.h:113-188
    u64 sh = key_bits - size_bits;
    u64 start = idx << sh, end = (idx + 1) << sh, r = 0;
    for (u64 x = start; x < end; ++x) r += test(x);
    *n_words += (end - start + 31) / 32;
    return r;
  }
  // number of i in (lo, hi) exclusive with test(i)==flag, dna.cpp:600-605
  u64 count_equal(u64 lo, u64 hi, u64 flag) const {
    u64 r = 0;
    u64 i = lo + 1;
    if (i >= hi) return 0;
    *n_words += (hi - i + 31) / 32;
    const u64 rep = flag * 0x5555555555555555ULL;
    while (i < hi && (i & 31)) { r += test(i) == flag; ++i; }
    while (i + 32 <= hi) {
      u64 x = w[i >> 5] ^ rep;
      u64 eq = ~(x | (x >> 1)) & 0x5555555555555555ULL;
      r += (u64)__builtin_popcountll(eq);
      i += 32;
    }
    while (i < hi) { r += test(i) == flag; ++i; }
    return r;
  }
  double avg_fill() const { return no_filled ? (double)no_updates / (double)no_filled : 0.0; }  // bit_vec.h:204-210
};

// ---------------------------------------------------------------------------------------
// CHT_pair_kmers (ht_kmer.h:559-663, ht_kmer.cpp:17-230): multimap key -> {(value, count)} with saturating
// counts.  insert() is commutative (count = min(sum, max)) and find() is consumed as a multiset by
// merge_minim_results, so one flat table (no T partitions) gives identical results.
struct PairTable {
  std::vector<u64> key, val;
  u64 hmask = 0, filled = 0, value_mask = 0, max_counter = 0;
  u32 cshift = 0;
  void init(u32 k, u64 cap) {
    cshift = 2 * k;
    value_mask = (1ull << cshift) - 1ull;
    max_counter = (~0ull) >> cshift;
    key.assign(cap, 0); val.assign(cap, 0);
    hmask = cap - 1; filled = 0;
  }
  void clear() { std::fill(key.begin(), key.end(), 0); std::fill(val.begin(), val.end(), 0); filled = 0; }
  bool empty_at(u64 p) const { return key[p] == 0 && val[p] == 0; }
  void grow() {
    std::vector<u64> ok, ov;
    ok.swap(key); ov.swap(val);
    key.assign(ok.size() * 2, 0); val.assign(ok.size() * 2, 0);
    hmask = key.size() - 1;
    for (size_t i = 0; i < ok.size(); ++i)
      if (ok[i] || ov[i]) {
        u64 p = murmur64(ok[i]) & hmask;
        while (!empty_at(p)) p = (p + 1) & hmask;
        key[p] = ok[i]; val[p] = ov[i];
      }
  }
  void insert(u64 k, u64 v, u64 count) {  // ht_kmer.cpp:124-187
    if (k == value_mask || v == value_mask) return;
    if ((filled + 1) * 2 > key.size()) grow();
    for (u64 p = murmur64(k) & hmask;; p = (p + 1) & hmask) {
      if (empty_at(p)) {
        if (count > max_counter) count = max_counter;
        key[p] = k; val[p] = v + (count << cshift);
        ++filled;
        return;
      }
      if (key[p] == k && (val[p] & value_mask) == v) {
        u64 cur = val[p] >> cshift;
        if (cur + count < max_counter) val[p] += count << cshift;
        else val[p] += (max_counter - cur) << cshift;
        return;
      }
    }
  }
  void find(u64 k, std::vector<u64> &out) const {  // ht_kmer.cpp:211-228
    for (u64 p = murmur64(k) & hmask; !empty_at(p); p = (p + 1) & hmask)
      if (key[p] == k) out.push_back(val[p]);
  }
};

// ---------------------------------------------------------------------------------------
// CRangeEncoder, sub_rc.h:32-87
struct RangeEnc {  // + CRangeDecoder, sub_rc.h:93-158, when dec is set
  u64 low, range;
  std::vector<u8> out;
  bool dec = false;
  const u8 *in = nullptr;
  u64 in_len = 0, in_pos = 0, buffer = 0;
  u8 get_byte() { return in_pos < in_len ? in[in_pos++] : (u8)0; }
  void start_dec(const u8 *p, u64 n) {  // sub_rc.h:112-125
    in = p; in_len = n; in_pos = 0; buffer = 0;
    if (n >= 8) for (u32 i = 1; i <= 8; ++i) buffer |= (u64)get_byte() << (64 - i * 8);
    low = 0; range = 0xff00000000000000ULL;
  }
  u64 cum_freq(u64 tot) { range /= tot; return buffer / range; }  // GetCumulativeFreq, sub_rc.h:127-131
  void update(u64 freq, u64 cum) {  // UpdateFrequency, sub_rc.h:133-151
    const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
    u64 r = cum * range;
    buffer -= r; low += r; range *= freq;
    while (range <= Top) {
      if ((low ^ (low + range)) & M) { u64 q = low; range = (q | Top) - q; }
      buffer = (buffer << 8) + get_byte();
      low <<= 8; range <<= 8;
    }
  }
  void start() { low = 0; range = 0xff00000000000000ULL; }
  void encode(u64 freq, u64 cum, u64 tot) {
    const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
    range /= tot;
    low += range * cum;
    range *= freq;
    while (range <= Top) {
      if ((low ^ (low + range)) & M) {
        u64 r = low;
        range = (r | Top) - r;
      }
      out.push_back((u8)(low >> 56));
      low <<= 8;
      range <<= 8;
    }
  }
  void end() {
    for (int i = 0; i < 8; ++i) { out.push_back((u8)(low >> 56)); low <<= 8; }
  }
};

// ---------------------------------------------------------------------------------------
// CSimpleModel / CSimpleModelFixedSize + Encode glue, rc.h:20-338,478-488
struct Model {
  u32 n, max_total, total;
  std::vector<u32> st;
  void init(u32 n_, const int *ini, u32 max_total_) {  // rc.h:55-83,208-224
    n = n_; max_total = max_total_;
    st.resize(n);
    total = 0;
    for (u32 i = 0; i < n; ++i) { st[i] = ini ? (u32)ini[i] : 1u; total += st[i]; }
    rescale();
  }
  void rescale() {  // rc.h:28-39
    while (total >= max_total) {
      total = 0;
      for (u32 i = 0; i < n; ++i) { st[i] = (st[i] + 1) / 2; total += st[i]; }
    }
  }
  // Encode (rc.h:397-405,478-488) or, on a decoding coder, Decode (rc.h:407-421,490-503); returns the symbol
  u32 encode(RangeEnc &rc, u32 x, u64 *n_coded) {
    u32 left = 0;
    if (rc.dec) {
      u64 lt = rc.cum_freq(total);
      u32 t = 0;
      x = n - 1;
      for (u32 i = 0; i < n; ++i) { t += st[i]; if (t > lt) { x = i; break; } }  // GetSym, rc.h:134-146
      for (u32 i = 0; i < x; ++i) left += st[i];
      rc.update(st[x], left);
    } else {
      for (u32 i = 0; i < x; ++i) left += st[i];
      rc.encode(st[x], left, total);
    }
    st[x] += 4;
    total += 4;
    if (total >= max_total) rescale();
    ++*n_coded;
    return x;
  }
};

// CContextHM, context_hm.h:21-248 (exact map ctx -> (model, counter); inserting an
// existing key is shadowed by the earlier entry in the reference, so it is a no-op here)
struct CtxMap {
  struct Item { u64 ctx; Model *m; u64 counter; };
  std::vector<Item> tab;
  u64 hmask, size;
  u64 *n_lookup;
  void init(u64 *nl) { tab.assign(1u << 12, Item{0, nullptr, 0}); hmask = tab.size() - 1; size = 0; n_lookup = nl; }
  ~CtxMap() { for (auto &it : tab) delete it.m; }
  Item *find(u64 ctx) {
    ++*n_lookup;
    for (u64 h = murmur64(ctx) & hmask;; h = (h + 1) & hmask) {
      if (!tab[h].m) return nullptr;
      if (tab[h].ctx == ctx) return &tab[h];
    }
  }
  void insert(u64 ctx, const Model &proto) {
    if (find(ctx)) return;
    if ((size + 1) * 2 > tab.size()) {
      std::vector<Item> old;
      old.swap(tab);
      tab.assign(old.size() * 2, Item{0, nullptr, 0});
      hmask = tab.size() - 1;
      for (auto &it : old)
        if (it.m) {
          u64 h = murmur64(it.ctx) & hmask;
          while (tab[h].m) h = (h + 1) & hmask;
          tab[h] = it;
        }
    }
    u64 h = murmur64(ctx) & hmask;
    while (tab[h].m) h = (h + 1) & hmask;
    tab[h] = Item{ctx, new Model(proto), 0};
    ++size;
  }
  Model *get(u64 ctx, const Model &tpl) {  // find_rc_context, dna.cpp:2060-2105,2289-2308
    Item *it = find(ctx);
    if (!it) { insert(ctx, tpl); it = find(ctx); }
    return it->m;
  }
};

// ---------------------------------------------------------------------------------------
// code_ctx.cpp
enum { LV_NONE = 0, LV_PMER = 1, LV_SMER = 2, LV_BMER = 3, LV_MIXED = 4, LV_BMER_UNC = 5 };  // defs.h:45

static u64 conv_lev1(u64 c, u32 cl) {  // code_ctx.cpp:26-81
  u64 f = (u64)cl << 5;
  if (cl == 0) {
    if (c < 5) return f + c;
    if (c < 8) return f + 5;
    if (c < 16) return f + 6;
    if (c < 32) return f + 7;
    if (c < 64) return f + 8;
    return f + 9;
  }
  if (cl >= 1 && cl <= 3) {
    static const u32 lim[] = {16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 288, 384, 512, 1024, 2048};
    if (c < 8) return f + c;
    for (u32 i = 0; i < sizeof(lim) / sizeof(lim[0]); ++i)
      if (c < lim[i]) return f + 8 + i;
    return f + 29;
  }
  return 0;
}
static u64 conv_lev24(u64 c, u32 cl) {  // code_ctx.cpp:84-164
  u64 f = (u64)cl << 5;
  if (cl == 0) {
    if (c < 3) return f + c;
    if (c < 5) return f + 3;
    return f + 4;
  }
  if (cl == 1) {
    if (c < 5) return f + c;
    if (c < 8) return f + 5;
    if (c < 13) return f + 6;
    if (c < 20) return f + 7;
    if (c < 30) return f + 8;
    return f + 9;
  }
  if (cl == 2) {
    if (c < 10) return f + c;
    if (c < 15) return f + 10;
    if (c < 20) return f + 11;
    if (c < 30) return f + 12;
    if (c < 50) return f + 13;
    return f + 14;
  }
  if (cl == 3) {
    static const u32 lim[] = {16, 24, 32, 48, 64, 128, 256, 512, 1024, 2048, 2048 + 32, 2048 + 64, 2048 + 128,
                              2048 + 192, 2048 + 256, 2048 + 384, 2048 + 512, 2048 + 768, 2048 + 1024};
    if (c < 10) return f + c;
    for (u32 i = 0; i < sizeof(lim) / sizeof(lim[0]); ++i)
      if (c < lim[i]) return f + 10 + i;
    return f + 29;
  }
  return 0;
}
static u64 conv_lev3(u64 c, u32 cl) {  // code_ctx.cpp:167-239
  u64 f = (u64)cl << 5;
  if (cl == 0) {
    if (c < 3) return f + c;
    if (c < 5) return f + 3;
    return f + 4;
  }
  if (cl == 1) {
    if (c < 5) return f + c;
    if (c < 8) return f + 5;
    if (c < 13) return f + 6;
    if (c < 20) return f + 7;
    if (c < 30) return f + 8;
    return f + 9;
  }
  if (cl == 2) {
    if (c < 10) return f + c;
    if (c < 13) return f + 10;
    if (c < 20) return f + 11;
    if (c < 30) return f + 12;
    if (c < 50) return f + 13;
    return f + 14;
  }
  if (cl == 3) {
    static const u32 lim[] = {18, 20, 25, 30, 40, 50, 60, 64, 68, 72, 76, 80, 84, 88};
    if (c < 15) return f + c;
    for (u32 i = 0; i < sizeof(lim) / sizeof(lim[0]); ++i)
      if (c < lim[i]) return f + 15 + i;
    return f + 29;
  }
  return 0;
}
static u64 conv_count(u64 c, u32 level, u32 cl) {  // code_ctx.cpp:15-23
  if (level == LV_PMER) return conv_lev1(c, cl);
  if (level == LV_BMER) return conv_lev3(c, cl);
  return conv_lev24(c, cl);
}
static void sort_desc4(u32 d[4], const u32 s[4]) {  // sort_copy_stats, utils.cpp:109-126
  int o0 = 0, o1 = 1, o2 = 2, o3;
  int r0 = s[1] > s[0]; o0 += r0; o1 -= r0;
  int r1 = s[2] > s[0]; o0 += r1; o2 -= r1;
  r0 = s[3] > s[0]; o0 += r0;
  r1 = s[2] > s[1]; o1 += r1; o2 -= r1;
  r0 = s[3] > s[1]; o1 += r0;
  r1 = s[3] > s[2]; o2 += r1;
  o3 = 6 - o0 - o1 - o2;
  d[o0] = s[0]; d[o1] = s[1]; d[o2] = s[2]; d[o3] = s[3];
}

// field layout, code_ctx.h:31-72
enum { SH_POS = 0, SH_LEVEL = 14, SH_C0 = 17, SH_C1 = 24, SH_C2 = 31, SH_C3 = 38, SH_RSYM = 45, SH_LETMAX = 49, SH_CORZ = 52 };
static const u64 EN_POS = 0x3fffull << SH_POS, EN_LEVEL = 7ull << SH_LEVEL;
static const u64 EN_C[4] = {0x7full << SH_C0, 0x7full << SH_C1, 0x7full << SH_C2, 0x7full << SH_C3};
static const u64 EN_RSYM = 0xfull << SH_RSYM, EN_LETMAX = 7ull << SH_LETMAX, EN_CORZ = 7ull << SH_CORZ;
static const u32 SH_CN[4] = {SH_C0, SH_C1, SH_C2, SH_C3};

struct KLen { u32 prefix, pmer, smer, bmer; };

// CCodeContext::determine_ctx_codes, code_ctx.cpp:257-324
static void ctx_codes(u64 a[7], const KLen &kl, const u32 counts[4], const u64 s_letters[4], u32 pos, u32 level,
                      u32 cor_zone, u64 ctx_r_sym, u32 read_len) {
  u64 mask = ~0ull, ctx = 0;
  const u32 pos_limit[6] = {0, kl.pmer, kl.smer, kl.bmer, kl.bmer, kl.bmer};
  u32 srt[4];
  sort_desc4(srt, counts);
  a[0] = ctx | mask;
  mask ^= EN_LEVEL | EN_C[0] | EN_C[1] | EN_C[2] | EN_C[3] | EN_POS;
  ctx += (u64)level << SH_LEVEL;
  for (int i = 0; i < 2; ++i) ctx += conv_count(srt[i], level, 1) << SH_CN[i];
  for (int i = 2; i < 4; ++i) ctx += conv_count(srt[i], level, 0) << SH_CN[i];
  if (pos < pos_limit[level]) ctx += (u64)pos << SH_POS;
  else if (pos + 5 >= read_len) ctx += (u64)(0x3fffull - (u32)(read_len - pos)) << SH_POS;
  else ctx += (u64)(pos_limit[level] + pos / 16) << SH_POS;
  a[1] = ctx | mask;
  mask ^= EN_CORZ | EN_RSYM;
  ctx += (u64)cor_zone << SH_CORZ;
  ctx += (u64)__builtin_popcountll(ctx_r_sym) << SH_RSYM;  // transform_r_sym, code_ctx.cpp:371-373
  a[2] = ctx | mask;
  ctx &= ~(EN_C[0] | EN_C[1]);
  ctx += conv_count(srt[0], level, 2) << SH_C0;
  ctx += conv_count(srt[1], level, 2) << SH_C1;
  a[3] = ctx | mask;
  ctx &= ~(EN_C[0] | EN_C[1] | EN_C[2] | EN_C[3]);
  ctx += conv_count(srt[0], level, 3) << SH_C0;
  ctx += conv_count(srt[1], level, 3) << SH_C1;
  ctx += conv_count(srt[2], level, 1) << SH_C2;
  ctx += conv_count(srt[3], level, 1) << SH_C3;
  a[4] = ctx | mask;
  mask ^= EN_LETMAX;
  u32 r = 0;  // let_max_element, code_ctx.cpp:327-338
  for (int i = 1; i < 4; ++i)
    if (counts[i] > counts[r]) r = i;
    else if (counts[i] == counts[r] && s_letters[i] > s_letters[r]) r = i;
  ctx += (u64)r << SH_LETMAX;
  a[5] = ctx | mask;
  ctx &= ~EN_POS;
  if (pos < pos_limit[level]) ctx += ((u64)pos + (1u << 13)) << SH_POS;
  else if (pos + 5 >= read_len) ctx += (u64)(0x3fffull - (u32)(read_len - pos)) << SH_POS;
  else ctx += ((u64)pos_limit[level] + pos / 8 + (1u << 13)) << SH_POS;
  a[6] = ctx | mask;
}

// CLettersContext::determine_ctx_letters, code_ctx.cpp:465-490
static void ctx_letters_keys(u64 a[10], const KLen &kl, u32 pos, u64 letters, u32 read_len) {
  u64 mask = ~0ull, ctx = 0;
  mask ^= 0x3fffull;
  if (pos < kl.pmer) ctx += (u64)pos;
  else if (pos + 5 > read_len) ctx += (u64)(0x3fffull - (u32)(read_len - pos));
  for (u32 i = 0; i < 10; ++i) {
    ctx += ((letters >> (4 * i)) & 7ull) << (14 + 3 * i);
    mask ^= 7ull << (14 + 3 * i);
    a[i] = ctx | mask;
  }
}

// ---------------------------------------------------------------------------------------
struct Shared {
  u32 T, dna_mode;
  KLen kl;
  Siv siv;
  KTable smer, bmer;
  // mailboxes [src][dst], application.h:56-59
  std::vector<std::vector<std::vector<u64>>> p_add, s_add, b_add;
  PairTable pe;                                 // ht_pe_mers, application.cpp:89
  std::vector<std::vector<u64>> pe_add;         // pe_mers_to_add, flattened per source: (key, value, weight) triples
  Counters cnt;
};

static const u64 CODE_THR[12] = {1, 32, 64, 64, 128, 512, 1024, 32, 128, 256, 2048, 4};                 // dna.h:33
static const u64 LETTERS_THR[12] = {1, 32, 64, 128, 256, 512, 2048, 4096, 8192, 16384, 16384, 4};       // dna.h:36

struct Worker {
  Shared *sh;
  u32 tid;
  RangeEnc rc;
  CtxMap m_flags, m_letters, m_codes, m_ps_flags, m_ps_nobytes, m_nibbles, m_bytes, m_Ns;
  Model t_flags, t_letters, t_codes, t_ps_flags, t_ps_nobytes, t_nibbles, t_bytes, t_Ns;
  double avg_code = 0.0, avg_letters = 0.0;
  u64 ctx_flags = 0, ctx_letters = 0, ctx_ps_flags = 0;
  CounterInc cinc_b, cinc_s, cinc_lb, cinc_ls;
  u64 s_letters[4] = {0, 0, 0, 0};
  KTable lb, ls;
  PairTable lpe;                                // ht_pe_mers_local
  Model m_minim_id;                             // ctx_rc_pe_minimizer_id, dna.cpp:142
  CtxMap m_minim_pos;                           // m_ctx_rc_minimizer_pos
  Model t_minim_pos;
  std::vector<u64> v_minim_cand, v_minim_top;
  Kmer pmer, smer, bmer, pmer_u, smer_u, bmer_u, pmer_prev;
  u32 cor_pos = 0, N_run = 0;
  u64 hidden_updates = 0;
  std::vector<u8> read_prev;
  u32 pmer_mod_shift;

  void init(Shared *s, u32 tid_) {  // CDNACompressor::Init, dna.cpp:95-174; SetKmerDS :2356-2390
    sh = s; tid = tid_;
    Counters &c = sh->cnt;
    for (CtxMap *m : {&m_flags, &m_letters, &m_codes, &m_ps_flags, &m_ps_nobytes, &m_nibbles, &m_bytes, &m_Ns}) m->init(&c.ctx);
    static const int ini_letters[5] = {10, 10, 10, 10, 1}, ini_codes[5] = {20, 6, 3, 2, 1};
    t_flags.init(2, nullptr, 1 << 12);
    t_letters.init(5, ini_letters, 1 << 15);
    t_codes.init(5, ini_codes, 1 << 15);
    t_ps_flags.init(1 + 4, nullptr, 1 << 12);
    t_ps_nobytes.init((2 * sh->kl.pmer + 7) / 8, nullptr, 1 << 12);
    t_nibbles.init(16, nullptr, 1 << 15);
    t_bytes.init(256, nullptr, 1 << 15);
    t_Ns.init(2, nullptr, 1 << 12);
    pmer.setup(sh->kl.pmer); smer.setup(sh->kl.smer); bmer.setup(sh->kl.bmer);
    pmer_u.setup(sh->kl.pmer); smer_u.setup(sh->kl.smer); bmer_u.setup(sh->kl.bmer);
    cinc_b.reset(7, 2, 63); cinc_s.reset(4095 / 2, 1, 4095);
    cinc_lb.reset(7, 2, 63); cinc_ls.reset(4095 / 2, 1, 4095);
    pmer_prev = pmer;
    lb.init(sh->kl.bmer, 6, 1u << 12, &c.lprobes, &c.slots);
    ls.init(sh->kl.smer, 12, 1u << 12, &c.lprobes, &c.slots);
    pmer_mod_shift = 2 * sh->kl.pmer - 12;
    lpe.init(sh->kl.bmer, 1u << 10);
    m_minim_id.init(16, nullptr, 1 << 15);
    m_minim_pos.init(&c.ctx);
    t_minim_pos.init(256, nullptr, 1 << 15);
  }

  u32 p_owner(u64 x) const { return (u32)((x >> pmer_mod_shift) % sh->T); }          // dna.cpp:658
  u32 sb_owner(u64 x) const { return (u32)(((x >> 46) & 0x3fffull) % sh->T); }       // dna.cpp:825, :2382-2386
  void push_p(u64 x) { sh->p_add[tid][p_owner(x)].push_back(x); }
  void push_s(u64 x) { sh->s_add[tid][sb_owner(x)].push_back(x); }
  void push_b(u64 x) { sh->b_add[tid][sb_owner(x)].push_back(x); }

  static u32 dna_code(u8 c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

  u8 rank(const u32 counts[4], u8 sym) const {  // dna.cpp:177-193
    if (sym == 4) return 4;
    u8 r = 0;
    for (int i = 0; i < 4; ++i)
      if (counts[i] != counts[sym]) r += counts[sym] < counts[i];
      else if (s_letters[i] != s_letters[sym]) r += s_letters[sym] < s_letters[i];
      else r += sym > i;
    return r;
  }

  u8 un_rank(const u32 counts[4], u8 r) const {  // dna.cpp:197-207
    if (r == 4) return 4;
    for (u8 i = 0; i < 4; ++i) if (rank(counts, i) == r) return i;
    return 4;
  }
  static u8 alpha(u32 sym) { return (u8)"ACGTN"[sym]; }

  bool find_counts_p(const Kmer &km, u32 counts[4]) {  // dna.cpp:210-226
    if (!km.full()) {
      Kmer t = km;
      for (int j = 0; j < 4; ++j) {
        t.replace_last(j);
        counts[j] = (u32)sh->siv.range_sum(t.aligned_rc(), 2 * (u64)t.cur);
      }
    } else
      sh->siv.counts(km.aligned_dir(), counts, false);
    return KTable::non_empty(counts);
  }
  bool rough_p(u32 counts[4]) {  // dna.cpp:229-254
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    for (u32 i = 0; i + 1 < sh->kl.pmer; ++i) {
      u64 d = pmer.dir;
      for (u64 j = 0; j < 4; ++j) {
        u32 shv = 62 - 2 * i;
        d = (d & ~(3ull << shv)) + (j << shv);
        sh->siv.counts(d >> (64 - 2 * pmer.cur), counts, true);
      }
    }
    return KTable::non_empty(counts);
  }
  bool rough_kt(const KTable &ht, const Kmer &can, CounterInc &cinc, u32 klen, u32 counts[4]) {  // dna.cpp:257-330
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    u32 loc[4];
    for (u32 i = 0; i + 1 < klen; ++i) {
      Kmer t = can;
      for (u64 j = 0; j < 4; ++j) {
        t.replace(j, i);
        if (ht.find(t, loc, cinc))
          for (int q = 0; q < 4; ++q) counts[q] = cinc.merge(counts[q], loc[q]);
      }
    }
    return KTable::non_empty(counts);
  }

  u32 find_counts(u32 counts[4]) {
    const u32 l = find_counts_(counts);
    sh->cnt.lv[l] += 1;
    return l;
  }
  u32 find_counts_(u32 counts[4]) {  // dna.cpp:457-502
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    u32 bmargin = sh->kl.bmer - sh->kl.smer - 1;
    u32 smargin = sh->kl.smer - sh->kl.pmer + 1;
    if (bmer.almost_full(bmargin)) {
      if (sh->bmer.find(bmer, counts, cinc_b)) {
        int sat = (counts[0] == 63) + (counts[1] == 63) + (counts[2] == 63) + (counts[3] == 63);
        if (sat > 1) {
          u32 c2[4];
          sh->smer.find(smer, c2, cinc_s);
          for (int i = 0; i < 4; ++i) counts[i] += c2[i];
          return LV_MIXED;
        }
        return LV_BMER;
      } else {
        if (lb.find(bmer, counts, cinc_lb)) return LV_BMER;
        if (bmer.dir != bmer_u.dir && sh->bmer.find(bmer_u, counts, cinc_b)) return LV_BMER_UNC;
      }
    }
    if (smer.almost_full(smargin)) {
      if (sh->smer.find(smer, counts, cinc_s) || ls.find(smer, counts, cinc_ls)) return LV_SMER;
    } else if (find_counts_p(pmer, counts))
      return LV_PMER;
    return LV_NONE;
  }

  bool repair_existing(u32 pos, const u32 counts[4], u8 sym) {  // dna.cpp:333-370
    u32 mx = 0;
    for (int i = 1; i < 4; ++i)
      if (counts[i] > counts[mx]) mx = i;
      else if (counts[i] == counts[mx] && s_letters[i] > s_letters[mx]) mx = i;
    if (sym != 4) {
      if (mx == sym) return false;
      if (counts[sym] != 0) return false;
      if (counts[mx] <= 3) return false;
    }
    pmer.replace_last(mx); smer.replace_last(mx); bmer.replace_last(mx);
    cor_pos = pos;
    return true;
  }
  bool repair_missing(u32 pos) {  // dna.cpp:374-454 (existing_count is always 0 at the call sites)
    if (sh->siv.avg_fill() < 7.0) return false;
    int best_c = 4, best_count = 0, best_pos = 0;
    const int max_dist = 6;
    const u32 min_count = 2;
    for (int j = 1; j < max_dist; ++j) {
      Kmer t = bmer;
      for (u32 c = 0; c < 4; ++c) {
        if (bmer.symbol(bmer.cur - 1 - j) == c) continue;
        t.replace(c, t.cur - 1 - j);
        int cnt = (int)sh->bmer.count(t.norm());
        if (cnt >= best_count && cnt >= (int)min_count) { best_c = (int)c; best_count = cnt; best_pos = j; }
      }
    }
    if (best_pos) {
      bmer.replace(best_c, bmer.cur - 1 - best_pos);
      if (best_pos < (int)smer.cur) smer.replace(best_c, smer.cur - 1 - best_pos);
      if (best_pos < (int)pmer.cur) pmer.replace(best_c, pmer.cur - 1 - best_pos);
      cor_pos = std::max(cor_pos, pos - (u32)best_pos);
      return true;
    }
    return false;
  }

  // find_rc_code_context / find_rc_letters_context, dna.cpp:2107-2286.  `thr_first` is the
  // table used for the start-level test; every later test uses CODE_THR (quirk, App. B).
  Model *find_leveled(CtxMap &m, const u64 *lev, int n_levels, const Model &tpl, double &avg, const u64 *thr_first) {
    int i;
    CtxMap::Item *p, *q;
    int start = (int)(avg + 0.49);
    p = m.find(lev[start]);
    if (p && p->counter < thr_first[start]) {
      p->counter += 1;
      avg = 0.999 * avg + (1.0 - 0.999) * (double)start;
      return p->m;
    }
    if (!p) {
      for (i = start - 1; i >= 0; --i)
        if ((p = m.find(lev[i])) != nullptr) break;
    } else {
      for (i = start + 1; i < n_levels; ++i) {
        q = m.find(lev[i]);
        if (!q) break;
        if (q->counter < CODE_THR[i]) {
          avg = 0.999 * avg + (1.0 - 0.999) * (double)i;
          q->counter += 1;
          return q->m;
        }
        p = q;
      }
      --i;
    }
    if (!p) {
      m.insert(lev[0], tpl);
      p = m.find(lev[0]);
      p->counter += 1;
      i = 0;
    }
    if (p->counter >= CODE_THR[i] && i + 1 < n_levels) {
      u64 key = p->ctx;  // p may move when the map grows
      Model proto = *p->m;
      (void)key;
      m.insert(lev[i + 1], proto);
      p = m.find(lev[i + 1]);
      p->counter += 1;
    } else
      p->counter += 1;
    avg = 0.999 * avg + (1.0 - 0.999) * (double)i;
    return p->m;
  }

  u8 code_letter(u32 pos, u8 sym, u32 read_len) {  // dna.cpp:776-785, :520-528 (decode: :1246-1254,1353-1360)
    u64 lev[10];
    ctx_letters_keys(lev, sh->kl, pos, ctx_letters, read_len);
    return (u8)find_leveled(m_letters, lev, 9, t_letters, avg_letters, LETTERS_THR)->encode(rc, sym, &sh->cnt.coded);
  }

  void prefix_direct(u8 *p) {  // compress_prefix_direct, dna.cpp:506-546 (start_pos == 0); decompress_prefix_direct :1347-1381
    ctx_letters = ~0ull;
    for (u32 i = 0; i < sh->kl.prefix; ++i) {
      u8 sym = rc.dec ? 0 : (u8)dna_code(p[i]);
      sym = code_letter(i, sym, 0);
      if (rc.dec) p[i] = alpha(sym);
      ctx_letters = (ctx_letters << 4) + sym;
      if (sym == 4) { sym = 0; cor_pos = i; }
      pmer.insert(sym); smer.insert(sym); bmer.insert(sym);
      pmer_u.insert(sym); smer_u.insert(sym); bmer_u.insert(sym);
    }
  }

  void prefix_sorted(const u8 *p) {  // compress_prefix_sorted, dna.cpp:549-661
    ctx_letters = ~0ull;
    bool was_N = false;
    u64 *nc = &sh->cnt.coded;
    for (u32 i = 0; i < sh->kl.pmer; ++i) {
      u8 sym = (u8)dna_code(p[i]);
      ctx_letters = (ctx_letters << 4) + sym;
      if (sym == 4) { sym = 3; was_N = true; N_run++; } else N_run = 0;
      pmer.insert(sym); smer.insert(sym); bmer.insert(sym);
      pmer_u.insert(sym); smer_u.insert(sym); bmer_u.insert(sym);
    }
    m_Ns.get(0, t_Ns)->encode(rc, was_N, nc);
    ctx_ps_flags = ((ctx_ps_flags << 1) + (u64)was_N) & 0xffff;
    u64 flag;
    if (pmer.dir == pmer_prev.dir) flag = 4;
    else flag = sh->siv.test(pmer.aligned_dir());
    m_ps_flags.get(ctx_ps_flags, t_ps_flags)->encode(rc, (u32)flag, nc);
    ctx_ps_flags = ((ctx_ps_flags << 3) + flag) & 0xffff;
    if (flag < 4) {
      u64 dif = sh->siv.count_equal(pmer_prev.aligned_dir(), pmer.aligned_dir(), flag);
      u64 nb = 1;
      for (u64 x = dif >> 8; x; x >>= 8) ++nb;  // no_bytes, utils.h:164-174
      m_ps_nobytes.get(ctx_ps_flags, t_ps_nobytes)->encode(rc, (u32)nb - 1, nc);
      if (nb == 1) {
        u32 hi = (u32)(dif >> 4), lo = (u32)(dif & 0xf);
        m_nibbles.get((1ull << 24) + flag, t_nibbles)->encode(rc, hi, nc);
        m_nibbles.get((2ull << 24) + flag * 256 + hi, t_nibbles)->encode(rc, lo, nc);
      } else {
        u64 hi_byte = dif >> (nb * 8 - 8);
        m_bytes.get(flag * 65536 + nb * 256 + nb, t_bytes)->encode(rc, (u32)hi_byte, nc);
        for (int i = 0; i < (int)nb - 1; ++i) {
          m_bytes.get((flag << 24) + (nb << 16) + (hi_byte << 8) + (u64)i, t_bytes)->encode(rc, (u32)(dif & 0xff), nc);
          dif >>= 8;
        }
      }
    }
    if (was_N)
      for (u32 i = 0; i < sh->kl.pmer; ++i)
        if (p[i] == 'T' || p[i] == 'N') m_Ns.get((u64)i + 1, t_Ns)->encode(rc, p[i] == 'N', nc);
    pmer_prev = pmer;
    push_p(pmer.aligned_dir());
    push_p(pmer.aligned_rc());
  }

  void prefix_sorted_dec(u8 *p) {  // decompress_prefix_sorted, dna.cpp:1384-1514
    u64 *nc = &sh->cnt.coded;
    bool was_N = m_Ns.get(0, t_Ns)->encode(rc, 0, nc) != 0;
    ctx_ps_flags = ((ctx_ps_flags << 1) + (u64)was_N) & 0xffff;
    u64 flag = m_ps_flags.get(ctx_ps_flags, t_ps_flags)->encode(rc, 0, nc);
    ctx_ps_flags = ((ctx_ps_flags << 3) + flag) & 0xffff;
    if (flag < 4) {
      u64 nb = m_ps_nobytes.get(ctx_ps_flags, t_ps_nobytes)->encode(rc, 0, nc) + 1, dif;
      if (nb == 1) {
        u64 hi = m_nibbles.get((1ull << 24) + flag, t_nibbles)->encode(rc, 0, nc);
        u64 lo = m_nibbles.get((2ull << 24) + flag * 256 + hi, t_nibbles)->encode(rc, 0, nc);
        dif = (hi << 4) + lo;
      } else {
        u64 hi_byte = m_bytes.get(flag * 65536 + nb * 256 + nb, t_bytes)->encode(rc, 0, nc);
        dif = hi_byte << (nb * 8 - 8);
        for (int i = 0; i < (int)nb - 1; ++i)
          dif += (u64)m_bytes.get((flag << 24) + (nb << 16) + (hi_byte << 8) + (u64)i, t_bytes)->encode(rc, 0, nc) << (i * 8);
      }
      u64 k = pmer_prev.aligned_dir() + 1;
      for (u64 j = 0; j <= dif; ++k)
        if (sh->siv.test(k) == flag) ++j;
      --k;
      for (u32 i = 0; i < sh->kl.pmer; ++i) { pmer.insert_front(k & 3); k >>= 2; }
    } else if (pmer_prev.full())
      pmer = pmer_prev;
    else {
      for (u32 i = 0; i < sh->kl.pmer; ++i) pmer.insert(0);
      pmer_prev = pmer;
    }
    ctx_letters = ~0ull;
    for (u32 i = 0; i < sh->kl.pmer; ++i) {
      u8 sym = (u8)pmer.symbol(i);
      p[i] = alpha(sym);
      if (was_N && sym == 3 && m_Ns.get((u64)i + 1, t_Ns)->encode(rc, 0, nc)) { p[i] = 'N'; sym = 4; }
      ctx_letters = (ctx_letters << 4) + sym;
      if (sym == 4) { sym = 3; N_run++; } else N_run = 0;
      smer.insert(sym); bmer.insert(sym);
      pmer_u.insert(sym); smer_u.insert(sym); bmer_u.insert(sym);
    }
    pmer_prev = pmer;
    push_p(pmer.aligned_dir());
    push_p(pmer.aligned_rc());
  }

  void suffix(u8 *p, u32 size, bool original_order, u32 start_pos = 0, bool reversed_pe = false) {  // compress_suffix, dna.cpp:674-877; decompress_suffix :1139-1345
    u32 counts[4] = {0, 0, 0, 0};
    u64 ctx_r_sym = 0;
    const KLen &kl = sh->kl;
    for (u32 i = start_pos ? start_pos : original_order ? kl.prefix : kl.pmer; i < size; ++i) {
      u8 sym = rc.dec ? 0 : (u8)dna_code(p[i]);
      pmer.insert_zero(); smer.insert_zero(); bmer.insert_zero();
      pmer_u.insert_zero(); smer_u.insert_zero(); bmer_u.insert_zero();
      u32 level = find_counts(counts);
      if (level == LV_BMER_UNC) {
        bmer = bmer_u; smer = smer_u; pmer = pmer_u;
        cor_pos = 0;
        level = LV_BMER;
      }
      bool rough = false;
      if (level == LV_NONE) {
        if (bmer.full()) {
          if (rough_kt(sh->bmer, bmer, cinc_b, kl.bmer, counts)) { level = LV_PMER; rough = true; }
        } else if (smer.full()) {
          if (rough_kt(sh->smer, smer, cinc_s, kl.smer, counts)) { level = LV_PMER; rough = true; }
        } else if (pmer.full()) {
          if (rough_p(counts)) { level = LV_PMER; rough = true; }
        }
      }
      if (level != LV_NONE && N_run < 2) {
        int cor_dist = level == LV_PMER ? (int)kl.pmer : level == LV_SMER ? (int)kl.smer : (int)kl.bmer;
        int d = (int)i - (int)cor_pos;
        u32 cor_zone = d < cor_dist ? (u32)(1 + 2 * (cor_dist - d) / cor_dist) : 0;
        if (rough) cor_zone = 3;
        u64 lev[7];
        if (!reversed_pe) ctx_codes(lev, kl, counts, s_letters, i, level, cor_zone, ctx_r_sym, size);
        else ctx_codes(lev, kl, counts, s_letters, size - i - 1, level, cor_zone, ctx_r_sym, ~0u);  // dna.cpp:750-752
        Model *m = find_leveled(m_codes, lev, 7, t_codes, avg_code, CODE_THR);
        u8 r_sym = rc.dec ? 0 : rank(counts, sym);
        r_sym = (u8)m->encode(rc, r_sym, &sh->cnt.coded);
        if (rc.dec) sym = un_rank(counts, r_sym);
        ctx_r_sym = ((ctx_r_sym << 1) + (r_sym == 0)) & 0xff;  // update_ctx_r_sym, dna.cpp:664-671
      } else {
        sym = code_letter(i, sym, size);
        ctx_r_sym = (ctx_r_sym << 1) & 0xff;
      }
      if (rc.dec) p[i] = alpha(sym);
      u64 sym_k = sym == 4 ? 0 : sym;
      ctx_letters = (ctx_letters << 4) + sym;
      if (sym == 4) ++N_run; else N_run = 0;
      pmer.replace_last(sym_k); smer.replace_last(sym_k); bmer.replace_last(sym_k);
      pmer_u.replace_last(sym_k); smer_u.replace_last(sym_k); bmer_u.replace_last(sym_k);
      if (sym < 4) {
        bool pmer_insert = true;
        if (bmer.full()) {
          u64 x = bmer.norm();
          push_b(x);
          lb.insert(x, cinc_lb); ++sh->cnt.linserts;
          if ((level == LV_SMER || level == LV_BMER || level == LV_MIXED || level == LV_BMER_UNC) && counts[sym] >= 3)
            pmer_insert = false;
        }
        if (smer.full()) {
          u64 x = smer.norm();
          push_s(x);
          ls.insert(x, cinc_ls); ++sh->cnt.linserts;
        }
        if (pmer.full() && i - cor_pos >= kl.pmer - 1) {
          if (pmer_insert) { push_p(pmer.aligned_dir()); push_p(pmer.aligned_rc()); }
          else hidden_updates += 2;
        }
      }
      if (bmer.full()) {
        bool rep = false;
        if (level == LV_BMER || level == LV_MIXED || level == LV_BMER_UNC) rep = repair_existing(i, counts, sym);
        else if (level == LV_NONE || level == LV_PMER) rep = repair_missing(i);
        if (rep) {
          u64 x = bmer.norm();
          push_b(x);
          lb.insert(x, cinc_lb); ++sh->cnt.linserts;
        }
      }
    }
  }

  void update_s_letters(const u8 *p, u32 size) {  // dna.cpp:2047-2057
    for (u32 i = 0; i < size; ++i) {
      u32 c = dna_code(p[i]);
      if (c == 4) continue;
      s_letters[c]++;
      s_letters[3 - c]++;
    }
  }

  // CompressDirect/Sorted, dna.cpp:1517-1556,1716-1754; on a decoding coder DecompressSE, dna.cpp:1883-1928 (p is written)
  bool compress_read(u8 *p, u32 size, bool original_order, bool first_of_pair = true) {
    ctx_letters = 0;
    if (first_of_pair) {
      bool same = !rc.dec && read_prev.size() == size && (size == 0 || !memcmp(read_prev.data(), p, size));
      same = m_flags.get(ctx_flags, t_flags)->encode(rc, same, &sh->cnt.coded) != 0;
      ctx_flags = ((ctx_flags << 1) + (u64)same) & 0xff;
      if (same) {
        if (rc.dec) memcpy(p, read_prev.data(), std::min<size_t>(size, read_prev.size()));
        return true;
      }
    }
    pmer.reset(); smer.reset(); bmer.reset();
    pmer_u.reset(); smer_u.reset(); bmer_u.reset();
    cor_pos = 0; N_run = 0;
    if (original_order) prefix_direct(p);
    else if (rc.dec) prefix_sorted_dec(p);
    else prefix_sorted(p);
    suffix(p, size, original_order);
    if (first_of_pair) read_prev.assign(p, p + size);
    update_s_letters(p, size);
    return false;
  }

  // ---- paired-end (dna.cpp:880-1136,1559-1638,1757-1880)
  bool valid_minimizer(u64 x) const { u64 f = x >> (2 * sh->kl.bmer - 6); return f != 0 && f != 1; }            // dna.cpp:879-889
  bool valid_maximizer(u64 x) const { u64 f = x >> (2 * sh->kl.bmer - 6); return f != 0x3e && f != 0x3f; }      // dna.cpp:892-902
  // direct-strand b-mer roller of find_minimizer/maximizer/generate_read_bmers (CKmer direct mode)
  struct DirK {
    u64 v = 0; u32 cur = 0, k = 0;
    void reset() { v = 0; cur = 0; }
    void insert(u64 s) { v = ((v << 2) | s) & ((1ull << (2 * k)) - 1ull); if (cur < k) ++cur; }
    bool full() const { return cur == k; }
  };
  u64 find_minimizer(const u8 *p, int size) const {  // dna.cpp:999-1023
    DirK b; b.k = sh->kl.bmer;
    u64 best = sh->pe.value_mask;
    for (int i = 0; i < size; ++i) {
      u32 sym = dna_code(p[i]);
      if (sym == 4) b.reset();
      else { b.insert(sym); if (b.full() && b.v < best && valid_minimizer(b.v)) best = b.v; }
    }
    return best;
  }
  u64 find_maximizer(const u8 *p, int size) const {  // dna.cpp:1026-1050 (walks the read backwards)
    DirK b; b.k = sh->kl.bmer;
    u64 best = 0;
    for (int i = size - 1; i >= 0; --i) {
      u32 sym = dna_code(p[i]);
      if (sym == 4) b.reset();
      else { b.insert(sym); if (b.full() && b.v > best && valid_maximizer(b.v)) best = b.v; }
    }
    return best;
  }
  void merge_minim_results() {  // dna.cpp:905-971
    std::vector<u64> &c = v_minim_cand, &top = v_minim_top;
    if (c.size() == 1) { std::swap(c, top); return; }
    const u64 cs = 2 * (u64)sh->kl.bmer, vm = (1ull << cs) - 1ull, maxc = (~0ull) >> cs;
    auto by_count = [&](u64 x, u64 y) { u64 xc = x >> cs, yc = y >> cs; if (xc != yc) return xc > yc; return (x & vm) < (y & vm); };
    if (c.size() > 48) { std::partial_sort(c.begin(), c.begin() + 48, c.end(), by_count); c.resize(48); }
    std::sort(c.begin(), c.end(), [vm](u64 x, u64 y) { return (x & vm) < (y & vm); });
    top.clear();
    top.push_back(c.front());
    for (size_t i = 1; i < c.size(); ++i)
      if ((top.back() & vm) != (c[i] & vm)) top.push_back(c[i]);
      else {
        u64 cx = top.back() >> cs, cy = c[i] >> cs;
        if (cx + cy > maxc) cy = maxc - cx;
        top.back() += cy << cs;
      }
    if (top.size() <= 16) std::sort(top.begin(), top.end(), by_count);
    else std::partial_sort(top.begin(), top.begin() + 16, top.end(), by_count);
  }
  bool find_minim_cand(const u8 *p, u32 size) {  // dna.cpp:1757-1787
    v_minim_cand.clear();
    int k = (int)sh->kl.bmer, mss = (int)size - k + 1;
    int sp1 = mss / 4, sp2 = 2 * mss / 4, sp3 = 3 * mss / 4;
    u64 m[4] = {find_minimizer(p, sp1 + k - 1), find_minimizer(p + sp1, sp2 - sp1 + k - 1),
                find_minimizer(p + sp2, sp3 - sp2 + k - 1), find_minimizer(p + sp3, (int)size - sp3)};
    for (int i = 0; i < 4; ++i) sh->pe.find(m[i], v_minim_cand);
    for (int i = 0; i < 4; ++i) lpe.find(m[i], v_minim_cand);
    if (v_minim_cand.empty()) return false;
    merge_minim_results();
    return true;
  }
  void seed_kmers(const u8 *p, int from, int to) {  // dna.cpp:1577-1592,1616-1631
    pmer.reset(); smer.reset(); bmer.reset(); pmer_u.reset(); smer_u.reset(); bmer_u.reset();
    cor_pos = 0; N_run = 0;
    ctx_letters = ~0ull;
    for (int i = from; i < to; ++i) {
      u32 c = dna_code(p[i]);
      ctx_letters = (ctx_letters << 4) + c;
      if (c == 4) c = 0;
      pmer.insert(c); bmer.insert(c); smer.insert(c);
      pmer_u.insert(c); bmer_u.insert(c); smer_u.insert(c);
    }
  }
  static u8 rc_alpha(u8 c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }  // utils.h:103-114
  // CompressDirectWithMinim, dna.cpp:1559-1638; decoding: DecompressDirectWithMinim, :1641-1713 (anchor = minim b-mer)
  void compress_with_minim(u8 *p, u32 size, u32 mpos, u64 minim = 0) {
    const u32 k = sh->kl.bmer;
    if (rc.dec)
      for (u32 i = 0; i < k; ++i) p[mpos + i] = alpha((u32)((minim >> (2 * (k - 1 - i))) & 3));
    seed_kmers(p, (int)mpos, (int)(mpos + k));
    suffix(p, size, true, k + mpos);
    std::vector<u8> rcp(mpos + k, 'A');
    const int from = rc.dec ? (int)mpos : 0;   // the decoder only knows the anchor yet
    for (int i = (int)mpos + (int)k - 1, j = 0; i >= from; --i, ++j) rcp[j] = rc_alpha(p[i]);
    seed_kmers(rcp.data(), 0, (int)k);
    suffix(rcp.data(), (u32)rcp.size(), true, k, true);
    if (rc.dec)
      for (int i = (int)mpos - 1, j = (int)k; i >= 0; --i, ++j) p[i] = rc_alpha(rcp[j]);
    update_s_letters(p, size);
  }
  void pe_push(u64 key, u64 value, u64 weight) {  // my_pe_mers_to_add + ht_pe_mers_local->insert, dna.cpp:1090-1135
    std::vector<u64> &v = sh->pe_add[tid];
    v.push_back(key); v.push_back(value); v.push_back(weight);
    lpe.insert(key, value, weight);
  }
  void append_pe_mers3(const u8 *p1, u32 size1, const u8 *p2, u32 size2) {  // dna.cpp:1053-1136
    const int k = (int)sh->kl.bmer;
    int mss = (int)size1 - k + 1, a = mss / 3, b = 2 * mss / 3;
    u64 m11 = find_minimizer(p1, a + k - 1), m12 = find_minimizer(p1 + a, b - a + k - 1), m13 = find_minimizer(p1 + b, (int)size1 - b);
    mss = (int)size2 - k + 1; a = mss / 3; b = 2 * mss / 3;
    u64 m21 = find_minimizer(p2, a + k - 1), m22 = find_minimizer(p2 + a, b - a + k - 1), m23 = find_minimizer(p2 + b, (int)size2 - b);
    int mid1 = ((int)size1 + k) / 2, mid2 = ((int)size2 + k) / 2;
    u64 x1 = find_maximizer(p1 + mid1 - k + 1, (int)size1 - (mid1 - k + 1));
    u64 x2 = find_minimizer(p2 + mid2 - k + 1, (int)size2 - (mid2 - k + 1));  // sic: minimizer (quirk 6)
    x1 = (~x1) & sh->pe.value_mask;
    x2 = (~x2) & sh->pe.value_mask;
    pe_push(m11, m21, 2); pe_push(m11, m23, 4); pe_push(m11, x1, 1);
    pe_push(m12, m21, 3); pe_push(m12, m23, 3);
    pe_push(m13, m21, 4); pe_push(m13, m23, 2);
    pe_push(m21, m11, 2); pe_push(m21, m13, 4); pe_push(m21, x2, 1);
    pe_push(m22, m11, 3); pe_push(m22, m13, 4);
    pe_push(m23, m11, 4); pe_push(m23, m13, 2);
  }
  void decompress_pair(u8 *p1, u32 size1, u8 *p2, u32 size2, bool original_order) {  // DecompressPE, dna.cpp:1931-2044
    u64 *nc = &sh->cnt.coded;
    compress_read(p1, size1, original_order, true);
    bool found = find_minim_cand(p1, size1);
    bool direct = !found;
    u32 mid = 0, mpos = 0;
    if (found) {
      mid = m_minim_id.encode(rc, 0, nc);
      if (mid == 15) direct = true;
      else {
        mpos = m_minim_pos.get((u64)mid, t_minim_pos)->encode(rc, 0, nc);
        if (mpos == 254) {
          mpos = m_minim_pos.get((u64)mid + 0x100, t_minim_pos)->encode(rc, 0, nc) << 8;
          mpos += m_minim_pos.get((u64)mid + 0x200, t_minim_pos)->encode(rc, 0, nc);
        } else if (mpos == 255) {
          mpos = m_minim_pos.get((u64)mid + 0x300, t_minim_pos)->encode(rc, 0, nc) << 16;
          mpos += m_minim_pos.get((u64)mid + 0x400, t_minim_pos)->encode(rc, 0, nc) << 8;
          mpos += m_minim_pos.get((u64)mid + 0x500, t_minim_pos)->encode(rc, 0, nc);
        }
      }
    }
    if (direct) compress_read(p2, size2, true, false);
    else compress_with_minim(p2, size2, mpos, v_minim_top[mid] & sh->pe.value_mask);
    append_pe_mers3(p1, size1, p2, size2);
  }
  void compress_pair(u8 *p1, u32 size1, u8 *p2, u32 size2, bool original_order) {  // CompressPE, dna.cpp:1790-1880
    if (rc.dec) { decompress_pair(p1, size1, p2, size2, original_order); return; }
    u64 *nc = &sh->cnt.coded;
    compress_read(p1, size1, original_order, true);
    bool found = find_minim_cand(p1, size1);
    u32 mpos = 0;
    int mid = -1;
    if (found) {
      const u32 k = sh->kl.bmer;
      std::vector<std::pair<u64, u32>> rb;  // generate_read_bmers, dna.cpp:974-996
      DirK b; b.k = k;
      for (u32 i = 0; i < size2; ++i) {
        u32 sym = dna_code(p2[i]);
        if (sym == 4) b.reset();
        else { b.insert(sym); if (b.full() && valid_minimizer(b.v)) rb.emplace_back(b.v, i - (k - 1)); }
      }
      for (size_t i = 0; i < v_minim_top.size() && mid < 0; ++i) {
        u64 m = v_minim_top[i] & sh->pe.value_mask;
        for (auto &x : rb)
          if (x.first == m) { mid = (int)i; mpos = x.second; break; }
      }
      if (mid < 0 || mid > 14) mid = 15;
    }
    if (mid < 0) compress_read(p2, size2, true, false);
    else {
      m_minim_id.encode(rc, (u32)mid, nc);
      if (mid == 15) compress_read(p2, size2, true, false);
      else {
        if (mpos < 254) m_minim_pos.get((u64)mid, t_minim_pos)->encode(rc, mpos, nc);
        else if (mpos < 65536) {
          m_minim_pos.get((u64)mid, t_minim_pos)->encode(rc, 254, nc);
          m_minim_pos.get((u64)mid + 0x100, t_minim_pos)->encode(rc, mpos >> 8, nc);
          m_minim_pos.get((u64)mid + 0x200, t_minim_pos)->encode(rc, mpos & 0xff, nc);
        } else {
          m_minim_pos.get((u64)mid, t_minim_pos)->encode(rc, 255, nc);
          m_minim_pos.get((u64)mid + 0x300, t_minim_pos)->encode(rc, mpos >> 16, nc);
          m_minim_pos.get((u64)mid + 0x400, t_minim_pos)->encode(rc, (mpos >> 8) & 0xff, nc);
          m_minim_pos.get((u64)mid + 0x500, t_minim_pos)->encode(rc, mpos & 0xff, nc);
        }
        compress_with_minim(p2, size2, mpos);
      }
    }
    append_pe_mers3(p1, size1, p2, size2);
  }

  void insert_phase() {  // InsertKmersToHT, dna.cpp:2393-2472 (column tid of every mailbox, source order)
    u64 nf = 0, nu = 0;
    for (u32 i = 0; i < sh->T; ++i)
      for (u64 x : sh->p_add[i][tid]) { nf += sh->siv.increment(x); ++nu; }
    sh->siv.no_filled += nf;
    sh->siv.no_updates += nu + hidden_updates;
    hidden_updates = 0;
    for (u32 i = 0; i < sh->T; ++i)
      for (u64 x : sh->s_add[i][tid]) { sh->smer.insert(x, cinc_s); ++sh->cnt.inserts; }
    for (u32 i = 0; i < sh->T; ++i)
      for (u64 x : sh->b_add[i][tid]) { sh->bmer.insert(x, cinc_b); ++sh->cnt.inserts; }
    if (tid == 0)  // pair inserts are commutative (saturating sums): owner sharding does not matter
      for (u32 i = 0; i < sh->T; ++i) {
        std::vector<u64> &v = sh->pe_add[i];
        for (size_t j = 0; j + 3 <= v.size(); j += 3) sh->pe.insert(v[j], v[j + 1], v[j + 2]);
      }
  }
  void clear_phase() {  // ClearKmersToHT, dna.cpp:2475-2488
    for (u32 i = 0; i < sh->T; ++i) { sh->p_add[tid][i].clear(); sh->s_add[tid][i].clear(); sh->b_add[tid][i].clear(); }
    lb.clear(); ls.clear();
    sh->pe_add[tid].clear();
    lpe.clear();
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------
// Quality stream (next row N1): CQualityCompressor, quality.cpp:32-222
struct QualWorker {
  u32 qmode = 4, n_sym = 0, bits = 0, nctx = 0;
  u64 ctx_mask = 0;
  u32 fwd[96];
  std::vector<u64> keys;            // open-addressed context map: key -> model index (context_hm.h semantics)
  std::vector<u32> midx;
  std::vector<std::vector<u32>> st; // stats per model
  std::vector<u32> tot;
  u64 hmask = 0;
  RangeEnc rc;
  void init(u32 mode, u32 thr) {    // Init + adjust_quality_map_*, quality.cpp:32-149
    qmode = mode;
    for (int i = 0; i < 96; ++i) fwd[i] = 0;
    auto band = [&](int a, int b, u32 v) { for (int i = a; i < b; ++i) fwd[i] = v; };
    if (mode == 0) { n_sym = 96; bits = 6; nctx = 2; for (int i = 0; i < 96; ++i) fwd[i] = i; }
    else if (mode == 1) { n_sym = 8; bits = 4; nctx = 6; band(0, 2, 0); band(2, 10, 1); band(10, 20, 2); band(20, 25, 3); band(25, 30, 4); band(30, 35, 5); band(35, 40, 6); band(40, 96, 7); }
    else if (mode == 2) { n_sym = 4; bits = 3; nctx = 9; band(0, 2, 0); band(2, 15, 1); band(15, 31, 2); band(31, 96, 3); }
    else if (mode == 3) { n_sym = 2; bits = 2; nctx = 10; band(0, (int)thr, 0); band((int)thr, 96, 1); }
    ctx_mask = (1ull << (bits * nctx)) - 1ull;
    keys.assign(1u << 12, ~0ull); midx.assign(1u << 12, 0); hmask = keys.size() - 1;
  }
  u32 model_of(u64 ctx) {           // find_rc_context, quality.cpp:218-226
    for (u64 h = murmur64(ctx) & hmask;; h = (h + 1) & hmask) {
      if (keys[h] == ctx) return midx[h];
      if (keys[h] == ~0ull) {
        if ((st.size() + 1) * 2 > keys.size()) {
          std::vector<u64> ok; std::vector<u32> om;
          ok.swap(keys); om.swap(midx);
          keys.assign(ok.size() * 2, ~0ull); midx.assign(ok.size() * 2, 0); hmask = keys.size() - 1;
          for (size_t i = 0; i < ok.size(); ++i)
            if (ok[i] != ~0ull) { u64 q = murmur64(ok[i]) & hmask; while (keys[q] != ~0ull) q = (q + 1) & hmask; keys[q] = ok[i]; midx[q] = om[i]; }
          return model_of(ctx);
        }
        keys[h] = ctx; midx[h] = (u32)st.size();
        st.emplace_back(n_sym, 1u); tot.push_back(n_sym);
        return midx[h];
      }
    }
  }
  void compress(const u8 *q, u32 size) {  // Compress, quality.cpp:152-175; model: CSimpleModel adder 1, max_total 2^15
    if (qmode == 4) return;
    u64 ctx = ctx_mask;
    for (u32 i = 0; i < size; ++i) {
      u32 m = model_of(ctx), x = fwd[q[i] - 33], left = 0;
      std::vector<u32> &s = st[m];
      for (u32 j = 0; j < x; ++j) left += s[j];
      rc.encode(s[x], left, tot[m]);
      s[x] += 1; tot[m] += 1;
      while (tot[m] >= (1u << 15)) { tot[m] = 0; for (auto &v : s) { v = (v + 1) / 2; tot[m] += v; } }
      u64 my = ctx + (1ull << 48), t = (ctx << bits) + x;   // update_context, quality.cpp:209-215
      ctx = (my & ~ctx_mask) + (t & ctx_mask);
    }
  }
};
struct fqo_qual { u32 T; std::vector<QualWorker> w; };

struct fqo_codec {
  Shared sh;
  std::vector<Worker *> w;
  ~fqo_codec() { for (auto *x : w) delete x; }
};

extern "C" {

fqo_codec *fqo_create(const uint8_t *h) {
  if (!h || h[0] != 'K' || h[1] != 'C' || h[2] != 'S' || h[3] != 'D') return nullptr;  // params.h:102-129
  u32 T = h[4], mode = h[5];
  if (T == 0 || mode > 3) return nullptr;
  fqo_codec *c = new fqo_codec;
  Shared &s = c->sh;
  s.T = T; s.dna_mode = mode;
  s.kl.prefix = h[10]; s.kl.pmer = h[11]; s.kl.smer = h[12]; s.kl.bmer = h[13];
  s.siv.init(2 * s.kl.pmer, &s.cnt.siv_words);
  s.smer.init(s.kl.smer, 12, 1u << 16, &s.cnt.probes, &s.cnt.slots);  // application.cpp:86-89, defs.h:26-27
  s.bmer.init(s.kl.bmer, 6, 1u << 16, &s.cnt.probes, &s.cnt.slots);
  s.p_add.assign(T, std::vector<std::vector<u64>>(T));
  s.s_add = s.p_add; s.b_add = s.p_add;
  s.pe.init(s.kl.bmer, 1u << 16);
  s.pe_add.assign(T, std::vector<u64>());
  for (u32 i = 0; i < T; ++i) { c->w.push_back(new Worker); c->w.back()->init(&s, i); }
  return c;
}

void fqo_destroy(fqo_codec *c) { delete c; }

// one reads block through all T workers (application.cpp:610-669 / decoder mirror :874-917); when
// dec_streams is given the workers decode from them and write the reads into `bases`
static int run_block(fqo_codec *c, u8 *bases, const uint64_t *off, uint32_t n_reads, uint32_t generation,
                     const uint8_t *const *dec_streams, const uint64_t *dec_lens) {
  Shared &s = c->sh;
  const u64 T = s.T;
  // PartitionForWorkers, reads_block.h:197-214
  std::vector<u64> first(T), last(T), cursor(T);
  u64 lower = 0;
  for (u64 i = 0; i < T; ++i) {
    u64 upper = (i + 1) * n_reads / T;
    if (i < T - 1) upper &= ~1ull;
    first[i] = lower; last[i] = upper; lower = upper;
  }
  // calc_no_synchronizations, application.h:85-92
  u64 S = generation < 100u ? 100u - generation : 0u;
  u64 cap = (u64)n_reads / T / 2;
  if (S > cap) S = cap;
  if (S) --S;
  const bool orig = s.dna_mode == 0 || s.dna_mode == 2, paired = s.dna_mode >= 2;
  for (u64 t = 0; t < T; ++t) {  // application.cpp:624-628
    Worker &w = *c->w[t];
    w.read_prev.clear();
    w.rc.out.clear();
    w.rc.dec = dec_streams != nullptr;
    if (dec_streams) w.rc.start_dec(dec_streams[t], dec_lens[t]); else w.rc.start();
    cursor[t] = first[t];
  }
  for (u64 seg = 0; seg <= S; ++seg) {
    for (u64 t = 0; t < T; ++t) {  // application.cpp:630-656
      Worker &w = *c->w[t];
      u64 stop;  // one past the last read of this segment
      if (seg < S) {
        u64 ns = (seg + 1) * (last[t] - first[t]) / (S + 1) + first[t];
        if (!paired) stop = ns + 1;                     // SE: sync after read i == next_synchro (application.cpp:643)
        else {                                          // PE: after the first pair with i >= next_synchro (application.cpp:1170)
          u64 i = cursor[t];
          while (i < ns) i += 2;
          stop = i + 2;
        }
      } else stop = last[t];
      if (stop > last[t]) stop = last[t];
      if (!paired)
        for (u64 i = cursor[t]; i < stop; ++i) w.compress_read(bases + off[i], (u32)(off[i + 1] - off[i]), orig);
      else
        for (u64 i = cursor[t]; i + 1 < stop; i += 2)
          w.compress_pair(bases + off[i], (u32)(off[i + 1] - off[i]), bases + off[i + 1], (u32)(off[i + 2] - off[i + 1]), orig);
      if (stop > cursor[t]) cursor[t] = stop;
    }
    for (u64 t = 0; t < T; ++t) c->w[t]->insert_phase();
    for (u64 t = 0; t < T; ++t) c->w[t]->clear_phase();
  }
  if (!dec_streams)
    for (u64 t = 0; t < T; ++t) c->w[t]->rc.end();  // application.cpp:664-665
  return 0;
}

int fqo_encode_block(fqo_codec *c, const uint8_t *bases, const uint64_t *off, uint32_t n_reads, uint32_t generation) {
  return run_block(c, const_cast<u8 *>(bases), off, n_reads, generation, nullptr, nullptr);
}

int fqo_decode_block(fqo_codec *c, const uint8_t *const *streams, const uint64_t *lens, const uint64_t *off, uint32_t n_reads,
                     uint32_t generation, uint8_t *bases_out) {
  return run_block(c, bases_out, off, n_reads, generation, streams, lens);
}

const uint8_t *fqo_stream(fqo_codec *c, uint32_t worker, uint64_t *len) {
  if (worker >= c->sh.T) { *len = 0; return nullptr; }
  *len = c->w[worker]->rc.out.size();
  return c->w[worker]->rc.out.data();
}

void fqo_levels(fqo_codec *c, uint64_t o[10]) {
  for (int i = 0; i < 6; ++i) o[i] = c->sh.cnt.lv[i];
  o[6] = o[7] = o[8] = o[9] = 0;
  for (auto *w : c->w) { o[6] += w->cinc_b.draws; o[7] += w->cinc_s.draws; o[8] += w->cinc_lb.draws; o[9] += w->cinc_ls.draws; }
}
void fqo_counters(fqo_codec *c, uint64_t o[8]) {
  const Counters &k = c->sh.cnt;
  o[0] = k.probes; o[1] = k.slots; o[2] = k.inserts; o[3] = k.siv_words;
  o[4] = k.ctx; o[5] = k.coded; o[6] = k.lprobes; o[7] = k.linserts;
}

fqo_qual *fqo_qual_create(const uint8_t *h) {
  if (!h || h[0] != 'K' || h[4] == 0 || h[6] > 4) return nullptr;
  fqo_qual *q = new fqo_qual;
  q->T = h[4];
  q->w.resize(q->T);
  for (auto &w : q->w) w.init(h[6], h[8]);
  return q;
}
void fqo_qual_destroy(fqo_qual *q) { delete q; }
int fqo_qual_encode_block(fqo_qual *q, const uint8_t *quals, const uint64_t *off, uint32_t n_reads) {
  const u64 T = q->T;
  for (u64 t = 0; t < T; ++t) {
    u64 first = t * n_reads / T, last = (t + 1) * n_reads / T;  // reads_block.h:197-214
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    QualWorker &w = q->w[t];
    w.rc.out.clear();
    w.rc.start();
    for (u64 i = first; i < last; ++i) w.compress(quals + off[i], (u32)(off[i + 1] - off[i]));
    w.rc.end();
  }
  return 0;
}
const uint8_t *fqo_qual_stream(fqo_qual *q, uint32_t worker, uint64_t *len) {
  *len = q->w[worker].rc.out.size();
  return q->w[worker].rc.out.data();
}

void fqo_kat_mt19937(uint32_t seed, uint32_t n, uint32_t *out) {
  Mt19937 m;
  m.seed(seed);
  for (u32 i = 0; i < n; ++i) out[i] = m.next();
}

void fqo_kat_cinc(uint32_t thr, uint32_t mult, uint32_t maxv, uint32_t n, const uint32_t *a, const uint32_t *b, uint32_t *out) {
  CounterInc c;
  c.reset(thr, mult, maxv);
  for (u32 i = 0; i < n; ++i) out[i] = b[i] == 0xffffffffu ? c.inc1(a[i]) : c.merge(a[i], b[i]);
}

uint64_t fqo_kat_rc(uint32_t n, const uint32_t *freq, const uint32_t *cum, const uint32_t *tot, uint8_t *out, uint64_t cap) {
  RangeEnc rc;
  rc.start();
  for (u32 i = 0; i < n; ++i) rc.encode(freq[i], cum[i], tot[i]);
  rc.end();
  u64 m = std::min<u64>(cap, rc.out.size());
  memcpy(out, rc.out.data(), m);
  return rc.out.size();
}

}  // extern "C"
