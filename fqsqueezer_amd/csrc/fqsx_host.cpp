// fqsx_host.cpp -- host-side (CPU) plumbing that travels with the DNA path: the read-length
// ("meta") stream every .fqs block carries (reference fqs/meta.cpp:31-73, application.cpp:633).
// It is not part of the hot path (one symbol per read); it exists so that a complete container can be
// written around the GPU DNA streams and handed to the reference decoder.
#include "../../include/fqsx.h"

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
struct Worker { Model256 len, b0, b1, b2; Enc enc; };
}  // namespace

struct fqsx_meta { u32 T; std::vector<Worker> w; };

extern "C" {
int fqsx_meta_create(uint32_t T, fqsx_meta **out) {
  if (!out || T == 0 || T > 255) return FQSX_E_ARG;
  fqsx_meta *m = new fqsx_meta;
  m->T = T;
  m->w.resize(T);
  *out = m;
  return FQSX_OK;
}
void fqsx_meta_destroy(fqsx_meta *m) { delete m; }
int fqsx_meta_encode_block(fqsx_meta *m, const uint32_t *read_len, uint32_t n_reads, const uint8_t **streams, uint64_t *lens) {
  if (!m || !read_len || !streams || !lens) return FQSX_E_ARG;
  const u64 T = m->T;
  for (u64 t = 0; t < T; ++t) {
    u64 first = t * n_reads / T, last = (t + 1) * n_reads / T;  // reads_block.h:197-214
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    Worker &w = m->w[t];
    w.enc.start();
    for (u64 i = first; i < last; ++i) {  // encode_len(0, len), meta.cpp:48-73
      u32 L = read_len[i];
      if (L < 254) w.len.encode(w.enc, L);
      else if (L < 65536) { w.len.encode(w.enc, 254); w.b1.encode(w.enc, L >> 8); w.b2.encode(w.enc, L & 0xff); }
      else if (L < (1u << 24)) { w.len.encode(w.enc, 255); w.b0.encode(w.enc, L >> 16); w.b1.encode(w.enc, (L >> 8) & 0xff); w.b2.encode(w.enc, L & 0xff); }
      else return FQSX_E_ARG;
    }
    w.enc.end();
    streams[t] = w.enc.out.data();
    lens[t] = w.enc.out.size();
  }
  return FQSX_OK;
}
}
