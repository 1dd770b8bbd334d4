// fqsx_host.cpp -- host-side (CPU) plumbing that travels with the DNA path: the read-length
// ("meta") stream every .fqs block carries (reference fqs/meta.cpp:31-73, application.cpp:633), the read-id
// stream (fqs/id.cpp, SURVEY.md §8f row N4: string tokeniser + delta coder; the host twin of the GPU coder csrc/fqsx_idk.h,
// which the tests compare it with byte for byte) and the
// per-bin read order of sorted mode.  None of it is on the hot path; it exists so that a complete container
// can be written around the GPU DNA / quality streams and handed to the reference decoder.
#include "../../include/fqsx.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct Enc {  // CRangeEncoder, sub_rc.h:32-87
  u64 low, range;
  std::vector<u8> out;
  void start() { low = 0; range = 0xff00000000000000ULL; out.clear(); }
  void encode(u64 f, u64 c, u64 t) {
    const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
    range /= t; low += range * c; range *= f;
    while (range <= Top) {
      if ((low ^ (low + range)) & M) range = (low | Top) - low;
      out.push_back((u8)(low >> 56));
      low <<= 8; range <<= 8;
    }
  }
  void end() { for (int i = 0; i < 8; ++i) { out.push_back((u8)(low >> 56)); low <<= 8; } }
};
struct Model256 {  // CRangeCoderModel(256 symbols, adder 1, max_total 1<<15), meta.cpp:33-38, rc.h:344-405
  u32 st[256], total;
  Model256() { for (auto &x : st) x = 1; total = 256; }
  void encode(Enc &e, u32 x) {
    u32 cum = 0;
    for (u32 i = 0; i < x; ++i) cum += st[i];
    e.encode(st[x], cum, total);
    st[x] += 1; total += 1;
    while (total >= (1u << 15)) { total = 0; for (auto &v : st) { v = (v + 1) / 2; total += v; } }
  }
};
struct Worker { Model256 len[2], b0[2], b1[2], b2[2]; Enc enc; };

// ---- read ids (fqs/id.cpp) ---------------------------------------------------------------------------
// N-symbol adaptive model, all ones, adder 1, max_total 2^15 (CRangeCoderModel / CRangeCoderModelFixedSize as
// constructed in CIdCompressor::Init, id.cpp:84-105; rc.h:20-173,178-338)
struct ModelN {
  std::vector<u32> st;
  u32 total;
  explicit ModelN(u32 n = 2) : st(n, 1), total(n) {}
  void encode(Enc &e, u32 x) {
    u32 cum = 0;
    for (u32 i = 0; i < x; ++i) cum += st[i];
    e.encode(st[x], cum, total);
    st[x] += 1; total += 1;
    while (total >= (1u << 15)) { total = 0; for (auto &v : st) { v = (v + 1) / 2; total += v; } }
  }
};
struct CtxMap {  // exact map context -> model, created from the all-ones template on first use (id.cpp:763-811)
  u32 n;
  std::unordered_map<u64, ModelN> m;
  explicit CtxMap(u32 n_) : n(n_) {}
  ModelN &at(u64 ctx) {
    auto it = m.find(ctx);
    if (it == m.end()) it = m.emplace(ctx, ModelN(n)).first;
    return it->second;
  }
};
struct Token { u8 numeric, sep; u32 b, e; };
struct Mtf {  // classic move-to-front list of instrument names (mtf.cpp:52-116)
  std::vector<std::string> v;
  int code(const std::string &x) const {
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == x) return (int)i;
    return -1;
  }
  void insert(const std::string &x) {
    int p = code(x);
    if (p < 0) { v.push_back(x); p = (int)v.size() - 1; }
    if (p > 0) { std::string s = v[p]; v.erase(v.begin() + p); v.insert(v.begin(), s); }
  }
};
struct IdWorker {
  Enc enc;
  u64 ctx_flags = 0, ctx_pe_flags = 0;
  CtxMap flags{2}, pe_flags{2}, numeric_size{256}, numeric_small{4}, literal{128}, literal_same{2}, literal_same_length{2}, plain{128};
  std::vector<Token> tok_prev, tok_cur;
  std::vector<int64_t> deltas;
  std::vector<u8> id_prev, id_cur;   // private copies: instrument mode terminates the name in place (id.cpp:427-428)
  Mtf mtf;
  ModelN mtf_flag{11};
  std::vector<ModelN> mtf_code, mtf_byte;
  int err = 0;
  IdWorker() {
    for (int i = 0; i < 7; ++i) mtf_code.emplace_back(2u << i);
    for (int i = 0; i < 4; ++i) mtf_byte.emplace_back(256u);
  }
  void reset_read_prev() {  // CIdCompressor::ResetReadPrev, id.cpp:124-135
    id_prev.clear(); tok_prev.clear(); tok_cur.clear(); deltas.clear();
    ctx_flags = 0; ctx_pe_flags = 0;
  }
};
inline bool is_num(u8 c) { return c >= '0' && c <= '9'; }
inline bool is_lit(u8 c) { return is_num(c) || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '@'; }  // id.cpp:57-70
// tokenize, id.cpp:734-757: a token ends at every non-literal byte (kept as its separator); an all-digit token
// of 1..10 digits is numeric
void tokenize(const u8 *p, u32 size, std::vector<Token> &v) {
  u32 start = 0;
  bool numeric = true;
  v.clear();
  for (u32 i = 0; i < size; ++i) {
    if (!is_lit(p[i])) {
      if (numeric && (i - start >= 11 || i == start)) numeric = false;
      v.push_back(Token{(u8)numeric, p[i], start, i});
      numeric = true;
      start = i + 1;
    } else if (!is_num(p[i]))
      numeric = false;
  }
}
bool types_same(const std::vector<Token> &a, const std::vector<Token> &b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i].numeric != b[i].numeric || a[i].sep != b[i].sep) return false;
  return true;
}
int64_t get_int(const u8 *p, u32 b, u32 e) {
  int64_t r = 0;
  for (u32 i = b; i < e; ++i) r = r * 10 + (int64_t)(p[i] - '0');
  return r;
}
u64 ilog2_(u64 x) { u64 r = 0; for (; x; ++r) x >>= 1; return r; }  // utils.h:154-162
// compress_lossless, id.cpp:257-418
void id_lossless(IdWorker &w, const u8 *p, u32 size) {
  for (u32 i = 0; i < size; ++i) if (p[i] >= 128) { w.err = 1; return; }  // 128-symbol models (id.cpp:99,104)
  w.id_cur.assign(p, p + size);
  p = w.id_cur.data();
  const u8 *q = w.id_prev.data();
  tokenize(p, size, w.tok_cur);
  const u32 n_tok = (u32)w.tok_cur.size();
  ModelN &fl = w.flags.at(w.ctx_flags);
  if (types_same(w.tok_cur, w.tok_prev)) {
    fl.encode(w.enc, 1);
    w.ctx_flags = ((w.ctx_flags << 1) + 1) & 0xff;
    for (u32 i = 0; i < n_tok; ++i) {
      const Token &c = w.tok_cur[i], &pv = w.tok_prev[i];
      if (!c.numeric) {
        const u32 len = c.e - c.b;
        const bool same_length = len == pv.e - pv.b;
        const bool same = same_length && std::equal(p + c.b, p + c.e, q + pv.b);
        ModelN &ms = w.literal_same.at(i);
        if (same) { ms.encode(w.enc, 1); continue; }
        ms.encode(w.enc, 0);
        ModelN &ml = w.literal_same_length.at(i);
        if (same_length) {
          ml.encode(w.enc, 1);
          for (u32 j = 0; j < len; ++j) {
            ModelN &m = w.literal.at(w.ctx_flags + (1ull << 32) + j);
            m.encode(w.enc, p[c.b + j] == q[pv.b + j] ? 0u : (u32)p[c.b + j]);
          }
        } else {
          ml.encode(w.enc, 0);
          for (u32 j = 0; j < len; ++j) w.literal.at(w.ctx_flags + j).encode(w.enc, p[c.b + j]);
          w.literal.at(w.ctx_flags + len).encode(w.enc, 0);
        }
      } else {
        int64_t delta = get_int(p, c.b, c.e) - get_int(q, pv.b, pv.e);
        const int64_t d0 = w.deltas[i];
        u64 ctx = (u64)i << 40;
        ctx += ilog2_((u64)(d0 < 0 ? -d0 : d0)) << 31;
        ctx += (u64)(d0 < 0) << 30;
        ModelN &msz = w.numeric_size.at(ctx);
        ModelN &msm = w.numeric_small.at(ctx);
        w.deltas[i] = delta;
        if (delta >= -1 && delta <= 1) { msm.encode(w.enc, (u32)(delta + 1)); continue; }
        msm.encode(w.enc, 3);
        int n_bytes = 0;
        if (delta >= -123 && delta <= 123) msz.encode(w.enc, (u32)(delta + 123) & 0xff);
        else if (delta > 0 && delta < 0x10000ll) { msz.encode(w.enc, 247); n_bytes = 2; ctx += 0x10; }
        else if (delta > 0 && delta < 0x1000000ll) { msz.encode(w.enc, 248); n_bytes = 3; ctx += 0x20; }
        else if (delta > 0 && delta < 0x100000000ll) { msz.encode(w.enc, 249); n_bytes = 4; ctx += 0x30; }
        else if (delta > 0) { msz.encode(w.enc, 250); n_bytes = 8; ctx += 0x40; }
        else if (delta > -0x10000ll) { msz.encode(w.enc, 251); delta = -delta; n_bytes = 2; ctx += 0x50; }
        else if (delta > -0x1000000ll) { msz.encode(w.enc, 252); delta = -delta; n_bytes = 3; ctx += 0x60; }
        else if (delta > -0x100000000ll) { msz.encode(w.enc, 253); delta = -delta; n_bytes = 4; ctx += 0x70; }
        else { msz.encode(w.enc, 254); delta = -delta; n_bytes = 8; ctx += 0x80; }
        for (int j = 0; j < n_bytes; ++j) w.numeric_size.at(ctx + j).encode(w.enc, (u32)(((u64)delta >> (8 * j)) & 0xff));
      }
    }
  } else {
    fl.encode(w.enc, 0);
    w.ctx_flags = (w.ctx_flags << 1) & 0xff;
    for (u32 i = 0; i < size; ++i) w.plain.at(i).encode(w.enc, p[i]);
    w.deltas.assign(n_tok, 0);
  }
  w.tok_prev.swap(w.tok_cur);
  w.id_prev.swap(w.id_cur);
}
// compress_instrument, id.cpp:421-495: only the instrument name (up to the first '.', ' ' or ':') is kept
void id_instrument(IdWorker &w, const u8 *p, u32 size) {
  u32 n = 0;
  while (n < size && p[n] != '.' && p[n] != ' ' && p[n] != ':') ++n;
  if (n == size) { w.err = 2; return; }  // the reference would write its terminator over the first base here
  std::string name((const char *)p, n);
  name = std::string(name.c_str());      // std::string(char*) stops at an embedded NUL (id.cpp:428)
  int code = w.mtf.code(name);
  if (code < 0) {
    w.mtf_flag.encode(w.enc, 0);
    std::vector<u8> tmp(p, p + n);
    tmp.push_back(0);
    id_lossless(w, tmp.data(), n + 1);
  } else if (code < 2)
    w.mtf_flag.encode(w.enc, (u32)code + 1);
  else if (code < 256) {
    int k = 0;
    while ((4 << k) <= code) ++k;          // code in [2<<k, 4<<k)
    w.mtf_flag.encode(w.enc, 3 + k);
    w.mtf_code[k].encode(w.enc, (u32)code - (2u << k));
  } else {
    w.mtf_flag.encode(w.enc, 10);
    for (int i = 0; i < 4; ++i) { w.mtf_byte[i].encode(w.enc, (u32)code & 0xff); code >>= 8; }
  }
  w.mtf.insert(name);
}
bool typical_pe_ids(const u8 *a, u32 na, const u8 *b, u32 nb) {  // id.cpp:241-254
  if (na != nb || na < 3) return false;
  if (!std::equal(a, a + na - 2, b)) return false;
  return a[na - 2] == '1' && b[nb - 2] == '2';
}
}  // namespace

struct fqsx_meta { u32 T; std::vector<Worker> w; };
struct fqsx_id { u32 T, mode; std::vector<IdWorker> w; };

extern "C" {
// Per-bin read order of `fqs e -om s`: std::sort with the comparator of CSortedFASTQFile::sort_reads
// (fqs/io.h:499-528) on the reads in their input order.  The sort is unstable; using the same libstdc++
// algorithm on the same initial sequence reproduces the reference's order of reads that compare equal
// (identical DNA), which matters for paired-end data (the mates follow this order, io.h:541-550).
int fqsx_sort_bin(const uint8_t *bases, const uint64_t *off, const uint32_t *idx_in, uint32_t n, uint32_t *idx_out) {
  if (!bases || !off || !idx_in || !idx_out) return FQSX_E_ARG;
  struct R { const u8 *p; u32 len; u32 id; };
  std::vector<R> v(n);
  for (u32 i = 0; i < n; ++i) v[i] = R{bases + off[idx_in[i]], (u32)(off[idx_in[i] + 1] - off[idx_in[i]]), idx_in[i]};
  auto nt = [](u8 c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
  std::sort(v.begin(), v.end(), [&](const R &x, const R &y) {
    u32 m = x.len < y.len ? x.len : y.len;
    for (u32 i = 0; i < m; ++i) {
      int a = nt(x.p[i]), b = nt(y.p[i]);
      if (a != b) return a < b;
    }
    if (x.len != y.len) return x.len < y.len;
    for (u32 i = 0; i < m; ++i)
      if (x.p[i] != y.p[i]) return x.p[i] < y.p[i];
    return false;
  });
  for (u32 i = 0; i < n; ++i) idx_out[i] = v[i].id;
  return FQSX_OK;
}
int fqsx_meta_create(uint32_t T, fqsx_meta **out) {
  if (!out || T == 0 || T > 255) return FQSX_E_ARG;
  fqsx_meta *m = new fqsx_meta;
  m->T = T;
  m->w.resize(T);
  *out = m;
  return FQSX_OK;
}
void fqsx_meta_destroy(fqsx_meta *m) { delete m; }
static int encode_len(Worker &w, u32 mdl, u32 L) {  // encode_len(model, len), meta.cpp:48-73
  if (L < 254) w.len[mdl].encode(w.enc, L);
  else if (L < 65536) { w.len[mdl].encode(w.enc, 254); w.b1[mdl].encode(w.enc, L >> 8); w.b2[mdl].encode(w.enc, L & 0xff); }
  else if (L < (1u << 24)) { w.len[mdl].encode(w.enc, 255); w.b0[mdl].encode(w.enc, L >> 16); w.b1[mdl].encode(w.enc, (L >> 8) & 0xff); w.b2[mdl].encode(w.enc, L & 0xff); }
  else return FQSX_E_ARG;
  return FQSX_OK;
}
int fqsx_meta_encode_block_pe(fqsx_meta *m, const uint32_t *read_len, uint32_t n_reads, int paired, const uint8_t **streams, uint64_t *lens);
int fqsx_meta_encode_block(fqsx_meta *m, const uint32_t *read_len, uint32_t n_reads, const uint8_t **streams, uint64_t *lens) {
  return fqsx_meta_encode_block_pe(m, read_len, n_reads, 0, streams, lens);
}
// paired: reads alternate mate 1 / mate 2 and use models 0 / 1 (CompressReadLenPE, meta.cpp:100-107)
int fqsx_meta_encode_block_pe(fqsx_meta *m, const uint32_t *read_len, uint32_t n_reads, int paired, const uint8_t **streams, uint64_t *lens) {
  if (!m || !read_len || !streams || !lens) return FQSX_E_ARG;
  const u64 T = m->T;
  for (u64 t = 0; t < T; ++t) {
    u64 first = t * n_reads / T, last = (t + 1) * n_reads / T;  // reads_block.h:197-214
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    Worker &w = m->w[t];
    w.enc.start();
    for (u64 i = first; i < last; ++i)
      if (encode_len(w, paired ? (u32)((i - first) & 1) : 0u, read_len[i])) return FQSX_E_ARG;
    w.enc.end();
    streams[t] = w.enc.out.data();
    lens[t] = w.enc.out.size();
  }
  return FQSX_OK;
}

// ---- read ids: CIdCompressor::Compress / CompressPE for all T workers of one block ----------------------
int fqsx_id_create(const uint8_t *header17, fqsx_id **out) {
  if (!header17 || !out || memcmp(header17, "KCSD", 4) || header17[4] == 0 || header17[7] > 2) return FQSX_E_ARG;
  fqsx_id *h = new fqsx_id;
  h->T = header17[4];
  h->mode = header17[7];   // 0 lossless, 1 instrument, 2 none (params.h:18,92)
  h->w.resize(h->T);
  *out = h;
  return FQSX_OK;
}
void fqsx_id_destroy(fqsx_id *h) { delete h; }
// ids: concatenated id lines, each INCLUDING its '\n' (read_desc_t::id_len, defs.h:70-72); id_off: n_reads+1
// offsets; paired != 0: reads alternate mate 1 / mate 2.  Workers are independent (no shared state), so they
// run on host threads.
int fqsx_id_encode_block(fqsx_id *h, const uint8_t *ids, const uint64_t *id_off, uint32_t n_reads, int paired,
                         const uint8_t **streams, uint64_t *lens) {
  if (!h || !ids || !id_off || !streams || !lens || (paired && (n_reads & 1))) return FQSX_E_ARG;
  const u64 T = h->T;
  auto run = [&](u64 t) {
    u64 first = t * n_reads / T, last = (t + 1) * n_reads / T;  // reads_block.h:197-214
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    IdWorker &w = h->w[t];
    w.reset_read_prev();
    w.enc.start();
    if (h->mode != 2)
      for (u64 i = first; i < last && !w.err; i += paired ? 2 : 1) {
        const u8 *p1 = ids + id_off[i];
        const u32 n1 = (u32)(id_off[i + 1] - id_off[i]);
        if (!paired) {
          if (h->mode == 0) id_lossless(w, p1, n1); else id_instrument(w, p1, n1);
          continue;
        }
        const u8 *p2 = ids + id_off[i + 1];
        const u32 n2 = (u32)(id_off[i + 2] - id_off[i + 1]);
        if (h->mode == 0) {   // CompressPE, id.cpp:152-184
          ModelN &m = w.pe_flags.at(w.ctx_pe_flags);
          const bool typical = typical_pe_ids(p1, n1, p2, n2);
          w.ctx_pe_flags = ((w.ctx_pe_flags << 1) + (typical ? 1 : 0)) & 0xff;
          m.encode(w.enc, typical ? 1 : 0);
          id_lossless(w, p1, n1);
          if (!typical) id_lossless(w, p2, n2);
        } else {
          id_instrument(w, p1, n1);
          id_instrument(w, p2, n2);
        }
      }
    w.enc.end();
  };
  const u32 nthr = (u32)std::max<u64>(1, std::min<u64>(T, std::thread::hardware_concurrency()));
  if (nthr <= 1 || n_reads < 4096) for (u64 t = 0; t < T; ++t) run(t);
  else {
    std::vector<std::thread> th;
    for (u32 k = 0; k < nthr; ++k) th.emplace_back([&, k] { for (u64 t = k; t < T; t += nthr) run(t); });
    for (auto &x : th) x.join();
  }
  for (u64 t = 0; t < T; ++t) {
    if (h->w[t].err) return FQSX_E_ARG;
    streams[t] = h->w[t].enc.out.data();
    lens[t] = h->w[t].enc.out.size();
  }
  return FQSX_OK;
}
}
