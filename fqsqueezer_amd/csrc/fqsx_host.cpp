// fqsx_host.cpp -- host-side (CPU) plumbing that travels with the DNA path: the read-length
// ("meta") stream every .fqs block carries (reference fqs/meta.cpp:31-73, application.cpp:633).
// It is not part of the hot path (one symbol per read); it exists so that a complete container can be
// written around the GPU DNA streams and handed to the reference decoder.
#include "../../include/fqsx.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
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
}  // namespace

struct fqsx_meta { u32 T; std::vector<Worker> w; };

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
}
