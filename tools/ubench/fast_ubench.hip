// Micro-benchmark (diagnostic, not part of the product): the fast-path body of stage C -- find_leveled +
// slot_encode on pre-computed level keys -- run by ONE wave per workgroup in isolation, to separate the cost of the
// algorithm from the cost of living inside the large k_encode_segment kernel.  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -I fqsqueezer_amd/csrc -I include tools/ubench/fast_ubench.hip -o tools/ubench/fast_ubench
#include "fqsx_dev.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

template <int VAR> __device__ __forceinline__ void body(DevCfg &cfg, const u64 *keys, const u8 *rsym, u32 n, u64 *ticks) {
  __shared__ WgShared sm;
  Wk w;
  w.cfg = &cfg; w.sm = &sm; w.tid = blockIdx.x; w.err = 0;
  w.enc.low = 0; w.enc.range = 0xff00000000000000ULL; w.enc.len = 0; w.enc.cap = cfg.out_cap; w.enc.out = cfg.out + (u64)blockIdx.x * cfg.out_cap;
  w.avg_code = 0.0;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  u64 ctx_r_sym = 0;
  u64 t0 = fq_clock();
  for (u32 i0 = 0; i0 < n; i0 += 64) {
    FQ_SYNC();
    for (u32 l = 0; l < 7; ++l) sm.sp_key[FQ_LANE][l] = keys[(u64)(i0 + FQ_LANE) * 7 + l];
    sm.sp_rsym[FQ_LANE] = rsym[i0 + FQ_LANE];
    FQ_SYNC();
    for (u32 j = 0; j < 64; ++j) {
      const u64 rs = (u64)popc64(ctx_r_sym) << SH_RSYM;
      Slot4 s;
      u32 idx = 0;
      u32 r_sym = sm.sp_rsym[j];
      if (VAR != 2) idx = find_leveled(w, 1, sm.sp_key[j], rs, 7, w.avg_code, TPL_CODES_Q2, TPL_CODES_Q3, TPL_CODES_TOT, s);
      if (VAR == 0 && idx != FQSX_NIL) slot_encode(w, idx, s, r_sym);
      if (VAR == 2) rc_encode(w, 20 + r_sym, r_sym * 7, 977 + (j & 15));   // the range coder alone
      if (VAR == 3) { if (idx != FQSX_NIL) rc_encode(w, 20 + r_sym, r_sym * 7, 977 + (j & 15)); }
      ctx_r_sym = ((ctx_r_sym << 1) + (r_sym == 0 ? 1u : 0u)) & 0xff;
    }
  }
  u64 t1 = fq_clock();
  if (FQ_LANE == 0) {
    ticks[blockIdx.x * 4 + 0] = t1 - t0;
    ticks[blockIdx.x * 4 + 1] = w.st[ST_CTX];
    ticks[blockIdx.x * 4 + 2] = w.enc.len + (w.err << 20);
    ticks[blockIdx.x * 4 + 3] = (u64)(w.avg_code * 1000);
  }
}

extern "C" __global__ __launch_bounds__(64) void k_fast(DevCfg cfg, const u64 *keys, const u8 *rsym, u32 n, u64 *ticks) { body<0>(cfg, keys, rsym, n, ticks); }
extern "C" __global__ __launch_bounds__(64) void k_search(DevCfg cfg, const u64 *keys, const u8 *rsym, u32 n, u64 *ticks) { body<1>(cfg, keys, rsym, n, ticks); }
extern "C" __global__ __launch_bounds__(64) void k_rc(DevCfg cfg, const u64 *keys, const u8 *rsym, u32 n, u64 *ticks) { body<2>(cfg, keys, rsym, n, ticks); }
extern "C" __global__ __launch_bounds__(64) void k_search_rc(DevCfg cfg, const u64 *keys, const u8 *rsym, u32 n, u64 *ticks) { body<3>(cfg, keys, rsym, n, ticks); }

// candidate: exact floor(x / d), d < 2^16, from the hardware reciprocal (no IEEE division sequence)
__device__ __forceinline__ u64 div_rcp(u64 x, u32 d) {
  const double dd = (double)d;
  const double r0 = __builtin_amdgcn_rcp(dd);                 // ~26 bits
  const double rd = __builtin_fma(r0, __builtin_fma(-dd, r0, 1.0), r0);   // one Newton step: ~52 bits
  const u32 hi = (u32)(x >> 32), lo = (u32)x;
  u32 qh = (u32)((double)hi * rd);
  u32 ph = qh * d;
  if (ph > hi) { --qh; ph -= d; } else if (hi - ph >= d) { ++qh; ph += d; }
  const u32 r1 = hi - ph;                                   // < d
  const double remd = __builtin_fma((double)r1, 4294967296.0, (double)lo);   // exact: < 2^48
  u32 q = (u32)(remd * rd);                                 // rem / d < 2^32
  const u64 rem = ((u64)r1 << 32) | lo;
  u64 prod = (u64)q * d;
  if (prod > rem) --q; else if (rem - prod >= d) ++q;
  return ((u64)qh << 32) + q;
}
extern "C" __global__ void k_divtest(u64 *bad, u32 n) {
  u64 x = 0x9E3779B97F4A7C15ULL * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  u64 nbad = 0, maxerr = 0;
  for (u32 i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    u64 v = x | (1ull << 56);                  // the coder's range is >= 2^48; also try small and huge values
    if ((i & 7) == 1) v = x >> (x & 31);
    if ((i & 7) == 2) v = ~0ull - (x & 0xffff);
    u32 d = (u32)((x >> 20) % 32771u) + 1u;
    if ((i & 15) == 3) d = 32772u - (u32)(x & 7);
    if ((i & 15) == 4) d = 1u + (u32)(x & 3);
    u64 q = div_rcp(v, d), ref = v / d;
    if (q != ref) { ++nbad; u64 e = q > ref ? q - ref : ref - q; if (e > maxerr) maxerr = e; }
  }
  atomicAdd((unsigned long long *)&bad[0], (unsigned long long)nbad);
  atomicMax((unsigned long long *)&bad[1], (unsigned long long)maxerr);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
  const u32 n = 1u << 18, T = argc > 1 ? atoi(argv[1]) : 1;
  std::vector<u64> keys((size_t)n * 7);
  std::vector<u8> rs(n);
  u64 x = 88172645463325252ULL;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  const u32 m[7] = {1, 20, 100, 500, 1500, 3000, 6000};
  for (u32 i = 0; i < n; ++i) {
    u64 r = rnd() % 6000, r2 = rnd() % 6000;
    u32 c = (u32)(r < r2 ? r : r2);   // skewed towards small ids
    for (u32 l = 0; l < 7; ++l) keys[(size_t)i * 7 + l] = ((u64)l << 14) | ((u64)(c % m[l]) << 17) | (~0ull << 56 >> (l * 2) << 56);
    rs[i] = (rnd() % 100) < 93 ? 0 : (u8)(1 + rnd() % 4);
  }
  {
    void *db;
    CK(hipMalloc(&db, 16)); CK(hipMemset(db, 0, 16));
    hipLaunchKernelGGL(k_divtest, dim3(256), dim3(256), 0, 0, (u64 *)db, 20000u);
    CK(hipDeviceSynchronize());
    u64 hb[2];
    CK(hipMemcpy(hb, db, 16, hipMemcpyDeviceToHost));
    printf("div_rcp self-test: %llu mismatches in %.2e divisions (max error %llu)\n", (unsigned long long)hb[0], 256.0 * 256 * 20000, (unsigned long long)hb[1]);
  }
  DevCfg cfg;
  memset(&cfg, 0, sizeof cfg);
  const u64 cap = 65536;
  void *ctx, *filled, *out, *dk, *dr, *dt;
  CK(hipMalloc(&ctx, cap * 32 * T)); CK(hipMemset(ctx, 0, cap * 32 * T));
  CK(hipMalloc(&filled, 4 * T)); CK(hipMemset(filled, 0, 4 * T));
  cfg.out_cap = 1 << 20;
  CK(hipMalloc(&out, cfg.out_cap * T));
  CK(hipMalloc(&dk, keys.size() * 8)); CK(hipMemcpy(dk, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dr, n)); CK(hipMemcpy(dr, rs.data(), n, hipMemcpyHostToDevice));
  CK(hipMalloc(&dt, 32 * T));
  cfg.T = T; cfg.ctx = (CtxSlot *)ctx; cfg.ctx_cap_mask = cap - 1; cfg.ctx_filled = (u32 *)filled; cfg.out = (u8 *)out;
  const char *names[4] = {"full", "search only", "range coder only", "search + range coder (no model)"};
  for (int rep = 0; rep < 8; ++rep) {
    const int var = rep < 2 ? 0 : rep < 4 ? 1 : rep < 6 ? 2 : 3;
    if (rep == 2 || rep == 6) { CK(hipMemset(ctx, 0, cap * 32 * T)); CK(hipMemset(filled, 0, 4 * T)); }
    if (var == 0) hipLaunchKernelGGL(k_fast, dim3(T), dim3(64), 0, 0, cfg, (const u64 *)dk, (const u8 *)dr, n, (u64 *)dt);
    if (var == 1) hipLaunchKernelGGL(k_search, dim3(T), dim3(64), 0, 0, cfg, (const u64 *)dk, (const u8 *)dr, n, (u64 *)dt);
    if (var == 2) hipLaunchKernelGGL(k_rc, dim3(T), dim3(64), 0, 0, cfg, (const u64 *)dk, (const u8 *)dr, n, (u64 *)dt);
    if (var == 3) hipLaunchKernelGGL(k_search_rc, dim3(T), dim3(64), 0, 0, cfg, (const u64 *)dk, (const u8 *)dr, n, (u64 *)dt);
    CK(hipDeviceSynchronize());
    std::vector<u64> t(4 * T);
    CK(hipMemcpy(t.data(), dt, 32 * T, hipMemcpyDeviceToHost));
    std::vector<u32> f(T);
    CK(hipMemcpy(f.data(), filled, 4 * T, hipMemcpyDeviceToHost));
    printf("%-32s rep %d: %.1f ns per position (wave 0), %.2f ctx slots per position, %llu bytes out, avg level %.3f, contexts %u\n", names[var], rep,
           t[0] * 10.0 / n, (double)t[1] / n, (unsigned long long)(t[2] & 0xfffff), t[3] / 1000.0, f[0]);
  }
  return 0;
}
