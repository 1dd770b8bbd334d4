// fqsx_plat.h -- wavefront primitives used by the FQSX device code.
//
// The device code is written "wave-uniform": one 64-lane wavefront (= one workgroup) is
// one logical FQSqueezer worker.  Control flow and the coder state are identical in all
// lanes; the lanes diverge only inside batch primitives (k-mer probe batches, p-mer
// vector sweeps, 256-symbol model sums, mailbox insert batches), which are written as
// strided loops `for (i = lane; i < n; i += WAVE)` followed by a wave reduction.
//
// FQSX_EMU builds the same sources for a 1-lane "wave" on the host.  That build exists
// only so the kernels can be debugged in the GPU-less development container
// (tests/emu); it is never part of the product library, which has no CPU path.
#pragma once
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;
typedef int32_t i32;

#ifndef FQSX_EMU
#include <hip/hip_runtime.h>
#define FQ_DEV __device__ __forceinline__
#define FQ_DEVN __device__ __noinline__
// One role of a worker (resolve / coder / inserter / read head / scout): a real function, called once by its wave,
// so that every role gets its own register allocation and its own stretch of code.  It takes no pointer arguments:
// the LDS block and the kernel arguments are reached through fq_wg() / fq_args(), which keeps their address spaces
// (ds_* instructions, scalar loads) visible to the compiler inside the role.
#define FQ_ROLE static __device__ __noinline__
#define FQ_KERNEL extern "C" __global__
#define FQ_KERNEL64 extern "C" __global__ __launch_bounds__(64)
#define FQ_KERNEL320 extern "C" __global__ __launch_bounds__(320)
#define FQ_KERNEL192 extern "C" __global__ __launch_bounds__(192)
#define FQ_KERNEL512 extern "C" __global__ __launch_bounds__(512)
#define FQ_WAVE 64
#define FQ_LANE ((u32)(threadIdx.x & 63u))
#define FQ_BLOCK ((u32)blockIdx.x)
#define FQ_NBLOCKS ((u32)gridDim.x)
#define FQ_SHARED __shared__
// Cross-lane hand-off through LDS inside the single wavefront of a workgroup: the LDS queue of one
// wave is in order, so draining it (and stopping compiler reordering) is all that is needed.  This
// deliberately does NOT wait for outstanding global stores (a __syncthreads() would: vmcnt(0)).
#define FQ_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Full drain: lanes are about to read global memory other lanes of the wave have written.  A wave's own
// stores and loads go through its CU's vector cache, so waiting for the stores to complete is enough.  (Not
// __syncthreads(): the encode kernel runs two waves per workgroup that must never meet at a barrier, see below.)
#define FQ_SYNC_MEM() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
// The one barrier of the two-wave encode kernel (after the LDS queue indices are initialised)
#define FQ_WG_BARRIER() __syncthreads()
#define FQ_WAVE_ID ((u32)(threadIdx.x >> 6))
// LDS hand-off words between the two waves of a workgroup (release: everything this wave wrote before is
// visible to a wave that acquires the value)
FQ_DEV u32 lds_load_acq(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
FQ_DEV void lds_store_rel(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
FQ_DEV void fq_sleep() { __builtin_amdgcn_s_sleep(4); }
FQ_DEV u64 fq_clock_raw() { return (u64)wall_clock64(); }
// Liveness guard of a wait loop (never spin forever on the GPU): `spins` counts the iterations; every 4096th one looks
// at the 100 MHz wall clock and the wait is given up after FQSX_WAIT_LIMIT_S seconds -- time, not an iteration count,
// so a wave that is merely slow (GPU shared between processes, counter collection) is not mistaken for a stalled one.
// spins: low 12 bits = iterations since the last look, upper bits = 1/64 s units waited (0 = clock not sampled yet).
#define FQSX_WAIT_LIMIT_S 8u
FQ_DEV bool spin_expired(u32 &spins) {
  const u32 c = (spins + 1u) & 0xfffu;
  spins = (spins & ~0xfffu) | c;   // (the count wraps inside its 12 bits)
  if (c != 0) return false;
  const u32 now = (u32)(fq_clock_raw() >> 20) | 1u;        // ~10.5 ms units, never 0
  const u32 first = spins >> 12;
  if (first == 0) { spins = now << 12; return false; }       // first look: remember when the wait began
  spins = first << 12;
  return (u32)((now - first) & 0xfffffu) > FQSX_WAIT_LIMIT_S * 95u;
}
// System-scope fences of the partitioned sharded mode (DevCfg.sys_scope): an owner's table writes have to reach its HBM
// before another GPU loads them over xGMI, and a reader must not serve such a load from a line of the other GPU's memory
// it cached in an earlier launch.  Per-XCD L2s are write-back and not coherent with anything outside their XCD
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): release = buffer_wbl2 sc0 sc1 +
// s_waitcnt vmcnt(0) (writes the XCD L2's dirty lines back), acquire = buffer_inv sc0 sc1 (drops the CU's L1 and the
// non-local lines of the XCD's L2).  Once per wave / workgroup and launch, never per element.
FQ_DEV void fq_release_system() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
FQ_DEV void fq_acquire_system() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }
FQ_DEV void atomic_add64(u64 *p, u64 v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }
FQ_DEV u32 atomic_add32(u32 *p, u32 v) { return atomicAdd(p, v); }   // returns the old value

FQ_DEV u32 wave_sum32(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
FQ_DEV u64 wave_sum64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// exclusive prefix sum over the lanes of the wave
FQ_DEV u32 wave_excl_scan32(u32 v) {
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 y = __shfl_up(x, o, 64);
    if ((int)(threadIdx.x & 63u) >= o) x += y;
  }
  return x - v;
}
FQ_DEV void lds_inc32(u32 *p) { atomicAdd(p, 1u); }
FQ_DEV void lds_or64(u64 *p, u64 v) { atomicOr((unsigned long long *)p, (unsigned long long)v); }
FQ_DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }
FQ_DEV bool wave_all(bool p) { return __ballot(p) == __ballot(true); }
FQ_DEV u64 wave_ballot(bool p) { return __ballot(p); }
FQ_DEV u32 wave_bcast32(u32 v, u32 lane) { return __shfl(v, (int)lane, 64); }
FQ_DEV u64 wave_bcast64(u64 v, u32 lane) { return __shfl(v, (int)lane, 64); }
FQ_DEV u32 uniform32(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
FQ_DEV u64 uniform64(u64 v) {
  u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
// cache touch: the value of a load issued early for its side effect on L2 / the TLB is "used" here
FQ_DEV u64 touch_load(const u64 *p) { return *(const volatile u64 *)p; }
FQ_DEV void keep_live(u64 v) { asm volatile("" ::"v"((u32)v), "v"((u32)(v >> 32))); }
FQ_DEV u64 wave_min64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 y = __shfl_xor(v, o, 64);
    v = y < v ? y : v;
  }
  return v;
}
FQ_DEV u32 popc64(u64 v) { return (u32)__popcll(v); }
FQ_DEV u32 ctz64(u64 v) { return (u32)__ffsll((long long)v) - 1u; }
FQ_DEV u64 fq_clock() { return (u64)wall_clock64(); }  // 100 MHz constant clock (s_memrealtime)
FQ_DEV u64 atomic_cas64(u64 *p, u64 expect, u64 desired) {
  return (u64)atomicCAS((unsigned long long *)p, (unsigned long long)expect, (unsigned long long)desired);
}
// IEEE double multiply/add with no FMA contraction (reference dna.h:100-102 is built
// without -mfma): avg = 0.999*avg + (1-0.999)*level
FQ_DEV double ema_update(double avg, double level) { return __dadd_rn(__dmul_rn(0.999, avg), __dmul_rn(1.0 - 0.999, level)); }

#else  // ---------------------------------------------------------------- host emulation
#include <math.h>
#include <string.h>
#define FQ_DEV static inline
#define FQ_DEVN static
#define FQ_ROLE static
#define FQ_KERNEL static
#define FQ_KERNEL64 static
#define FQ_KERNEL320 static
#define FQ_KERNEL192 static
#define FQ_KERNEL512 static
#define FQ_WAVE 1
#define FQ_LANE 0u
#define FQ_BLOCK (fq_emu_block)
#define FQ_NBLOCKS (fq_emu_nblocks)
#define FQ_SHARED static thread_local
#define FQ_SYNC() ((void)0)
#define FQ_SYNC_MEM() ((void)0)
#define FQ_WG_BARRIER() ((void)0)
#define FQ_WAVE_ID 0u
FQ_DEV u32 lds_load_acq(const u32 *p) { return *p; }
FQ_DEV void lds_store_rel(u32 *p, u32 v) { *p = v; }
FQ_DEV void fq_sleep() {}
FQ_DEV void fq_release_system() {}
FQ_DEV void fq_acquire_system() {}
FQ_DEV bool spin_expired(u32 &spins) { return ++spins > (1u << 20); }
FQ_DEV void atomic_add64(u64 *p, u64 v) { *p += v; }
FQ_DEV u32 atomic_add32(u32 *p, u32 v) { const u32 o = *p; *p = o + v; return o; }
static thread_local u32 fq_emu_block = 0, fq_emu_nblocks = 1;
FQ_DEV u32 wave_sum32(u32 v) { return v; }
FQ_DEV u64 wave_sum64(u64 v) { return v; }
FQ_DEV u32 wave_excl_scan32(u32) { return 0; }
FQ_DEV void lds_inc32(u32 *p) { ++*p; }
FQ_DEV void lds_or64(u64 *p, u64 v) { *p |= v; }
FQ_DEV bool wave_any(bool p) { return p; }
FQ_DEV bool wave_all(bool p) { return p; }
FQ_DEV u64 wave_ballot(bool p) { return p ? 1ull : 0ull; }
FQ_DEV u32 wave_bcast32(u32 v, u32) { return v; }
FQ_DEV u64 wave_bcast64(u64 v, u32) { return v; }
FQ_DEV u32 uniform32(u32 v) { return v; }
FQ_DEV u64 uniform64(u64 v) { return v; }
FQ_DEV u64 touch_load(const u64 *p) { return *p; }
FQ_DEV void keep_live(u64) {}
FQ_DEV u32 popc64(u64 v) { return (u32)__builtin_popcountll(v); }
FQ_DEV u32 ctz64(u64 v) { return (u32)__builtin_ctzll(v); }
FQ_DEV u64 fq_clock() { return 0; }
FQ_DEV u64 atomic_cas64(u64 *p, u64 expect, u64 desired) {
  u64 old = *p;
  if (old == expect) *p = desired;
  return old;
}
FQ_DEV double ema_update(double avg, double level) {
  volatile double a = 0.999 * avg;
  volatile double b = (1.0 - 0.999) * level;
  return a + b;
}
#endif
