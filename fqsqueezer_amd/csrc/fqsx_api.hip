// fqsx_api.hip -- kernels, device-memory management and the C ABI (include/fqsx.h).
//
// Host side of the FQSX DNA path: it owns the HBM-resident state of all T logical workers,
// sizes the per-block buffers, and drives the kernel schedule of one reads block:
//
//   for seg in 0..S:  k_encode_segment  (T workgroups, one wavefront = one worker)
//                     k_insert_phase    (3 T workgroups: (owner, mailbox kind); an inserting and a prefetching wave each)
//                     clear local tables
//   k_finish_block
//
// which is the barrier structure of the reference worker loop (fqs/application.cpp:610-669)
// with kernel boundaries in place of CBarrier.  Built with hipcc for gfx950; the FQSX_EMU
// build (tests/emu only) runs the same kernels as plain loops for debugging without a GPU.
#include "fqsx_kernels.h"
#include "fqsx_qual.h"
#include "../../include/fqsx.h"
#include "fqsx_vm.h"
#include <sys/resource.h>

#ifndef FQSX_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>   // (types only: librccl is loaded with dlopen on first use, see fqsx_rccl_comm_create)
#endif
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
extern "C" const char *fqsx_last_error(void) { return g_err.c_str(); }
extern "C" const char *fqsx_version(void) {
#ifdef FQSX_EMU
  return "fqsx 0.1 (host emulation build - tests only)";
#else
  return "fqsx 0.1 (HIP gfx950)";
#endif
}

// ---------------------------------------------------------------------------------------
// kernels.  The encode / decode kernels live in translation units of their own (fqsx_k_se.hip, fqsx_k_pe.hip,
// fqsx_k_dec.hip; launchers in fqsx_kernels.h); here: the insert phase, the mailbox partition, growth, the block epilogue.
// Deferred growth check (single-end encoding): the host queues the kernels of all phases of a block without reading
// anything back.  The group-offset kernel of a phase sees the exact demand of the coming inserts; if a sub-table would
// pass the load the host grows it at, it posts the phase (err[1] = segment + 1) instead, and every later kernel of the
// queue -- the inserts of that phase included -- does nothing.  The host finds the word with the end-of-block transfer,
// grows the tables and takes the block up again at that phase's inserts (block_recover).  err[0]: device error word.
FQ_DEV bool phase_skip(const DevCfg &cfg) { return (cfg.err[0] | cfg.err[1]) != 0; }
#define FQSX_INS_THREADS 128   /* k_insert_phase: the inserting wave and the prefetching wave */
// grid = 3 * T: (owner, mailbox kind), plus -- single-end encoding -- workgroups that clear the workers' local tables
// (ClearKmersToHT, dna.cpp:2475-2488: the insert phase does not touch them), which saves that launch
// 128 threads: wave 0 inserts, wave 1 touches the buckets of the batches ahead (insert_prefetch_body)
#ifndef FQSX_EMU
extern "C" __global__ __launch_bounds__(128)
#else
static
#endif
void k_insert_phase(DevCfg cfg, u64 nb_slots, u64 ns_slots) {
  FQ_SHARED InsShared sm;
  if (phase_skip(cfg)) return;
#ifndef FQSX_EMU
  if (threadIdx.x == 0) { sm.pf_done = 0; sm.pf_on = blockDim.x > 64 ? 1u : 0u; }
  __syncthreads();
  if (FQ_WAVE_ID == 1) {
    if (FQ_BLOCK < 3 * cfg.T) insert_prefetch_body(cfg, &sm, FQ_BLOCK / 3, FQ_BLOCK % 3);
    return;
  }
#else
  sm.pf_done = 0; sm.pf_on = 0;
#endif
  if (FQ_BLOCK >= 3 * cfg.T) {
    const u64 stride = (u64)(FQ_NBLOCKS - 3 * cfg.T) * FQ_WAVE, first = (u64)(FQ_BLOCK - 3 * cfg.T) * FQ_WAVE + FQ_LANE;
    for (u64 i = first; i < nb_slots; i += stride) cfg.l_b.slots[i] = 0;
    for (u64 i = first; i < ns_slots; i += stride) cfg.l_s.slots[i] = 0;
    for (u64 i = first; i < 2ull * cfg.T; i += stride) cfg.l_s.filled[i] = 0;   // l_s.filled and l_b.filled are adjacent
    return;
  }
  insert_phase_body(cfg, &sm, FQ_BLOCK / 3, FQ_BLOCK % 3);
  // partitioned tables: this owner's slots are read by other GPUs in the next launch -- out of this XCD's L2 into HBM
  // before the kernel ends (the phase's collective then orders the end of this kernel before anybody's next look-up)
  if (cfg.sys_scope) fq_release_system();
}
// gathers the T streams of the block into one contiguous buffer (one D2H transfer per block)
FQ_KERNEL64 void k_compact_streams(DevCfg cfg, const u64 *lens, u8 *dst) {
  const u32 tid = FQ_BLOCK;
  u64 off = 0;
  for (u32 t = 0; t < tid; ++t) off += lens[t];
  const u64 n = lens[tid] <= cfg.out_cap ? lens[tid] : 0;
  const u8 *src = cfg.out + (u64)tid * cfg.out_cap;
  for (u64 i = FQ_LANE; i < n; i += FQ_WAVE) dst[off + i] = src[i];
}
// the same for the quality and id coders ([T][out_cap] streams, one workgroup per worker)
FQ_KERNEL64 void k_compact_generic(const u8 *out, u64 out_cap, const u64 *lens, u8 *dst) {
  const u32 tid = FQ_BLOCK;
  u64 off = 0;
  for (u32 t = 0; t < tid; ++t) off += lens[t];
  const u64 n = lens[tid] <= out_cap ? lens[tid] : 0;
  const u8 *src = out + (u64)tid * out_cap;
  for (u64 i = FQ_LANE; i < n; i += FQ_WAVE) dst[off + i] = src[i];
}
// ClearKmersToHT (dna.cpp:2475-2488): the workers' local tables and their fill counters, in one launch
FQ_KERNEL64 void k_clear_local(DevCfg cfg, u64 nb_slots, u64 ns_slots) {
  if (phase_skip(cfg)) return;
  const u64 stride = (u64)FQ_NBLOCKS * FQ_WAVE, first = (u64)FQ_BLOCK * FQ_WAVE + FQ_LANE;
  for (u64 i = first; i < nb_slots; i += stride) cfg.l_b.slots[i] = 0;
  for (u64 i = first; i < ns_slots; i += stride) cfg.l_s.slots[i] = 0;
  for (u64 i = first; i < 2ull * cfg.T; i += stride) cfg.l_s.filled[i] = 0;   // l_s.filled and l_b.filled are adjacent
}
FQ_KERNEL64 void k_finish_block(DevCfg cfg, u64 *lens /*[T+1]*/, u64 *end /*[2T+2]: lengths, context occupancies, error word, posted phase*/) {
  if (FQ_LANE == 0) {
    if (FQ_BLOCK == 0) { end[2 * cfg.T] = cfg.err[0]; end[2 * cfg.T + 1] = cfg.err[1]; }
    if (cfg.err[1]) return;   // the block is not finished yet (see phase_skip)
    finish_block_body(cfg, FQ_BLOCK);
    lens[FQ_BLOCK] = cfg.ws[FQ_BLOCK].out_len;
    end[FQ_BLOCK] = cfg.ws[FQ_BLOCK].out_len;
    end[cfg.T + FQ_BLOCK] = cfg.ctx_filled[FQ_BLOCK];
  }
}
// gathers the per-worker accounting counters: out[i] = sum over workers of stat[i]
FQ_KERNEL64 void k_gather_stats(DevCfg cfg, u64 *out) {
  for (u32 i = FQ_LANE; i < 64; i += FQ_WAVE) {
    u64 s = 0;
    for (u32 t = 0; t < cfg.T; ++t) s += cfg.ws[t].stat[i];
    out[i] = s;
  }
}
// stable partition of the mailbox lists by owner (count -> scan -> group offsets -> scatter)
// (one launch covers the three mailbox kinds: blocks [0, g0) kind 0, [g0, g0+g1) kind 1, the rest kind 2)
FQ_DEV void part_split(const DevCfg &cfg, u32 &kind, u32 &blk) {
  kind = 0;
  blk = FQ_BLOCK;
  for (; kind < 2; ++kind) {
    const u32 g = cfg.T * cfg.mail[kind].n_tiles;
    if (blk < g) break;
    blk -= g;
  }
}
FQ_KERNEL64 void k_part_count(DevCfg cfg) {
  FQ_SHARED u32 hist[256];
  u32 kind, blk;
  if (phase_skip(cfg)) return;
  part_split(cfg, kind, blk);
  part_count_body(cfg, kind, blk, hist);
}
FQ_KERNEL64 void k_part_scan(DevCfg cfg) {
  if (phase_skip(cfg)) return;
  part_scan_body(cfg, FQ_BLOCK / cfg.T, FQ_BLOCK % cfg.T);
}
// group offsets of mailbox `kind`, the demand words, and -- seg1 != 0 -- the growth check (one wave per kind)
FQ_DEV void part_dstoff_leader(const DevCfg &cfg, u32 kind, u32 *demand /*[2][T]+1*/, u32 seg1);
// seg1: segment + 1 when the growth check is left to this kernel (else 0: the host reads `demand` before the inserts)
FQ_KERNEL64 void k_part_dstoff(DevCfg cfg, u32 *demand /*[2][T]+1*/, u32 seg1) {
  if (cfg.err[0] || (cfg.err[1] && cfg.err[1] != seg1)) {   // (another workgroup of this launch may just have posted the phase)
    if (FQ_BLOCK == 0 && FQ_LANE == 0) demand[2 * cfg.T] = cfg.err[0];   // the host's per-phase check reads the error word here
    return;
  }
  part_dstoff_leader(cfg, FQ_BLOCK, demand, seg1);
}
FQ_DEV void part_dstoff_leader(const DevCfg &cfg, u32 kind, u32 *demand, u32 seg1) {
  part_dstoff_body(cfg, kind);
  // per-owner demand of the coming insert phase (s- and b-mers) + the device error word
  // per-owner demand of the coming insert phase (s- and b-mers), current occupancy, and the device error word:
  // everything the host needs for its growth decision in one transfer
  if (kind != MAIL_P)
    for (u32 d = FQ_LANE; d < cfg.T; d += FQ_WAVE) {
      const u32 o = (kind == MAIL_S ? 0 : cfg.T) + d;
      demand[o] = cfg.mail[kind].dst_tot[d];
      demand[2 * cfg.T + 1 + o] = (kind == MAIL_S ? cfg.g_s : cfg.g_b).filled[d];
    }
  if (kind == 0 && FQ_LANE == 0) demand[2 * cfg.T] = *cfg.err;
  if (seg1 && kind != MAIL_P) {   // the host's growth rule (block_segment): occupancy after the inserts <= cap / 2
    const KTab &t = kind == MAIL_S ? cfg.g_s : cfg.g_b;
    bool over = false;
    for (u32 d = FQ_LANE; d < cfg.T; d += FQ_WAVE) over |= ((u64)t.filled[d] + cfg.mail[kind].dst_tot[d]) * 100 > t.nb * FQSX_BKT * cfg.tab_load_pct;
    if (wave_any(over) && FQ_LANE == 0) cfg.err[1] = seg1;
  }
}
// demand != null (single-end encoding): no k_part_dstoff launch before this one -- every workgroup scans the T group
// totals itself (LDS), and the first workgroup of a kind does that kernel's work for the kind
FQ_KERNEL64 void k_part_scatter(DevCfg cfg, u32 seg1, u32 *demand) {
  FQ_SHARED u32 cursor[256];
  FQ_SHARED u32 ld[64];
  FQ_SHARED u64 gm[256];
  FQ_SHARED u32 doff[257];
  u32 kind, blk;
  if (cfg.err[0] || (cfg.err[1] && cfg.err[1] != seg1)) return;   // (the posted phase's own scatter runs: its inserts follow the growth)
  part_split(cfg, kind, blk);
  if (demand) {
    const Mail &m = cfg.mail[kind];
    u32 run = 0;
    for (u32 base = 0; base < cfg.T; base += FQ_WAVE) {
      const u32 d = base + FQ_LANE;
      const u32 v = d < cfg.T ? m.dst_tot[d] : 0;
      const u32 ex = wave_excl_scan32(v) + run;
      if (d < cfg.T) doff[d] = ex;
      run += wave_sum32(v);
    }
    if (FQ_LANE == 0) doff[cfg.T] = run;
    FQ_SYNC();
    if (blk == 0) part_dstoff_leader(cfg, kind, demand, seg1);
  }
  part_scatter_body(cfg, kind, blk, cursor, ld, gm, demand ? doff : nullptr);
}
// paired-end insert phase: per-owner demand, then the inserts
// seg1 != 0 (paired-end encoding on one GPU): nothing is read back inside a block -- the kernel applies the host's growth
// rule to the pair table itself and posts the phase (err[1] = seg1, see phase_skip) if owner FQ_BLOCK's sub-table would pass it
FQ_KERNEL64 void k_pe_demand(DevCfg cfg, u32 *demand, u32 seg1) {
  FQ_SHARED InsShared sm;
  if (seg1 && phase_skip(cfg)) return;
  pe_insert_body(cfg, &sm, FQ_BLOCK, true, demand);
  if (seg1 && FQ_LANE == 0 && ((u64)cfg.g_pe.filled[FQ_BLOCK] + demand[FQ_BLOCK]) * 2 > cfg.g_pe.cap_mask + 1) cfg.err[1] = seg1;
}
// the bucketed form of the phase (fqsx_pe.h): count, offsets + growth check, scatter, per-owner inserts
FQ_KERNEL void k_pe_bucket_count(DevCfg cfg) {
  if (phase_skip(cfg)) return;
#ifndef FQSX_EMU
  const u32 total = cfg.T * cfg.pe_cap, stride = gridDim.x * blockDim.x;
  for (u32 g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) pe_bucket_count_body(cfg, g);
#else
  for (u32 g = 0; g < cfg.T * cfg.pe_cap; ++g) pe_bucket_count_body(cfg, g);
#endif
}
FQ_KERNEL64 void k_pe_bucket_offsets(DevCfg cfg, u32 seg1) {   // one wave
  if (phase_skip(cfg)) return;
  u32 run = 0;
  bool over = false;
  for (u32 base = 0; base < cfg.T; base += FQ_WAVE) {
    const u32 o = base + FQ_LANE;
    const u32 v = o < cfg.T ? cfg.pe_bkt_n[o] : 0u;
    const u32 ex = wave_excl_scan32(v) + run;
    if (o < cfg.T) {
      cfg.pe_bkt_n[cfg.T + o] = ex;
      cfg.pe_bkt_cur[o] = 0;
      over |= ((u64)cfg.g_pe.filled[o] + v) * 2 > cfg.g_pe.cap_mask + 1;   // the host's growth rule: at most half full after the inserts
    }
    run += wave_sum32(v);
  }
  if (FQ_LANE == 0) cfg.pe_bkt_n[2 * cfg.T] = run;
  if (wave_any(over) && FQ_LANE == 0) cfg.err[1] = seg1;
}
FQ_KERNEL void k_pe_bucket_scatter(DevCfg cfg) {
  if (phase_skip(cfg)) return;
#ifndef FQSX_EMU
  const u32 total = cfg.T * cfg.pe_cap, stride = gridDim.x * blockDim.x;
  for (u32 g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) pe_bucket_scatter_body(cfg, g);
#else
  for (u32 g = 0; g < cfg.T * cfg.pe_cap; ++g) pe_bucket_scatter_body(cfg, g);
#endif
}
FQ_KERNEL64 void k_pe_insert_buckets(DevCfg cfg) {   // grid = T (owner); also leaves the counters zero for the next phase
  FQ_SHARED InsShared sm;
  if (phase_skip(cfg)) return;
  pe_insert_bucket_body(cfg, &sm, FQ_BLOCK);
  FQ_SYNC_MEM();
  if (FQ_LANE == 0) cfg.pe_bkt_n[FQ_BLOCK] = 0;
}
FQ_KERNEL64 void k_pe_insert(DevCfg cfg) {
  FQ_SHARED InsShared sm;
  if (phase_skip(cfg)) return;
  // (partitioned pair table: every rank holds every source's triples, but a sub-table is written by its owner's rank alone)
  if (cfg.pe_part && FQ_BLOCK % cfg.shard_world != cfg.shard_rank) return;
  pe_insert_body(cfg, &sm, FQ_BLOCK, false, nullptr);
  if (cfg.sys_scope) fq_release_system();
}
// re-insert the occupied slots of sub-tables first, first + step, ... (n_sub of them) of `o` into the (empty, larger) table `n`
FQ_KERNEL void k_rehash_ptab(PTab o, PTab n, u32 n_sub, u32 first, u32 step, u32 sys) {
  const u64 ocap = o.cap_mask + 1, total = ocap * n_sub;
#ifndef FQSX_EMU
  const u64 gstride = (u64)gridDim.x * blockDim.x;
  for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gstride) {
#else
  for (u64 g = 0; g < total; ++g) {
#endif
    const u32 sub = first + (u32)(g / ocap) * step;
    const u64 k = o.key[(u64)sub * o.stride + (g % ocap)], v = o.val[(u64)sub * o.stride + (g % ocap)];
    if (k == 0 && v == 0) continue;
    u64 *nk = n.key + (u64)sub * n.stride, *nv = n.val + (u64)sub * n.stride;
    u64 p = murmur64(k) & n.cap_mask;
    for (;;) {  // valid keys are non-zero (a minimizer never starts with AAA), so the key word claims the slot
      if (nk[p] == 0 && atomic_cas64(&nk[p], 0, k) == 0) { nv[p] = v; break; }
      p = (p + 1) & n.cap_mask;
    }
  }
  if (sys) fq_release_system();
}
// re-insert every occupied slot of `o` into the (empty, larger) table `n`; layout-free, so parallel
// n zero words at p (the chunks of a chunked table are cleared by a kernel of the codec's own stream, so that the order
// against the re-insert kernel that follows does not hang on how the runtime treats a memset into a mapped range)
// sys: the words are a sub-table other GPUs will map (system-scope release at the end, see DevCfg.sys_scope)
FQ_KERNEL void k_zero_words(u64 *p, u64 n, u32 sys) {
#ifndef FQSX_EMU
  const u64 gstride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride) p[i] = 0;
#else
  for (u64 i = 0; i < n; ++i) p[i] = 0;
#endif
  if (sys) fq_release_system();
}
// (sub-tables first, first + step, ...: all of them on one GPU; a rank's own ones when the tables are partitioned)
FQ_KERNEL void k_rehash_ktab(KTab o, KTab n, u32 n_sub, u32 first, u32 step, u32 sys) {
  const u64 ocap = o.nb * FQSX_BKT;
  const u64 total = ocap * n_sub;
#ifndef FQSX_EMU
  const u64 gstride = (u64)gridDim.x * blockDim.x;
  for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gstride) {
#else
  for (u64 g = 0; g < total; ++g) {
#endif
    u32 sub = first + (u32)(g / ocap) * step;
    u64 it = o.slots[(u64)sub * o.stride + (g % ocap)];
    if (!it) continue;
    bool claimed = false;
    (void)tab_find_or_claim(n, n.slots + (u64)sub * n.stride, it >> n.cbits, it, claimed);
  }
  if (sys) fq_release_system();   // (the new chunks are mapped by the other ranks next: k_zero_words)
}
FQ_KERNEL void k_rehash_ctx(const u64 *o, u64 ocap_mask, u64 *n, u64 ncap_mask, u32 T) {
  const u64 ocap = ocap_mask + 1, total = ocap * T;
#ifndef FQSX_EMU
  const u64 gstride = (u64)gridDim.x * blockDim.x;
  for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gstride) {
#else
  for (u64 g = 0; g < total; ++g) {
#endif
    u32 w = (u32)(g / ocap);
    const u64 *src = o + 4 * g;
    u64 q1 = src[1];
    u32 tag = slot_tag(q1);
    if (!tag) continue;
    u64 key = src[0];
    u64 *b = n + 4 * (u64)w * (ncap_mask + 1);
    u64 h = (u64)ctx_mix(tag, key) & ncap_mask;
    for (;;) {
      u64 *p = b + 4 * h;
      if (p[1] == 0 && atomic_cas64(&p[1], 0, q1) == 0) {
        p[0] = key; p[2] = src[2]; p[3] = src[3];
        break;
      }
      h = (h + 1) & ncap_mask;
    }
  }
}

// ---- sharded mode: what is exchanged between the GPUs of a node ---------------------------------
// C[kind][s][o] = entries source s pushed for owner o in this phase (after k_part_count, before k_part_scan turns the
// tile counts into offsets); rows of workers that live elsewhere stay zero, so a sum over the ranks gives the matrix
// status: what this rank's host has to report (a failure of its own since the last vote); [3 T^2 + T] = (status | device error
// word) != 0, so that after the all-reduce every rank knows whether ANY rank is in trouble and all leave the phase together
FQ_KERNEL64 void k_shard_counts(DevCfg cfg, u32 *dst, u32 status) {   // grid = 3 * T + 1: (kind, source), then the tail
  const u32 T = cfg.T;
  if (FQ_BLOCK == 3 * T) {   // tail of the count matrix: [3 T^2 + s] = paired-end triples source s pushed in this phase (own sources, else 0)
    for (u32 s = FQ_LANE; s < T; s += FQ_WAVE) dst[3ull * T * T + s] = (cfg.pe_n && shard_mine(cfg, s)) ? cfg.pe_n[s] : 0u;
    if (FQ_LANE == 0) dst[3ull * T * T + T] = (status | cfg.err[0]) != 0 ? 1u : 0u;
    return;
  }
  const u32 kind = FQ_BLOCK / T, s = FQ_BLOCK % T;
  const Mail &m = cfg.mail[kind];
  for (u32 o = FQ_LANE; o < T; o += FQ_WAVE) {
    u32 n = 0;
    if (shard_mine(cfg, s))
      for (u32 t = 0; t < m.n_tiles; ++t) n += m.tile_hist[((u64)s * m.n_tiles + t) * T + cfg.vmap[o]];
    dst[((u64)kind * T + s) * T + o] = n;
  }
}
// ---- native sharded driver (fqsx_shard_encode_block): the words that ride along with the collectives ------------
// after the all-reduce of the counts: out[0/1] = table demand of the coming insert phase (occupied + incoming slots of the
// fullest s- / b-mer sub-table: every rank holds a replica of every sub-table, so every rank computes the same numbers),
// out[2] = the device error word
// Column sums of the all-reduced count matrix, once per phase for everybody who needs them (k_shard_need, the merge waves):
// cs[(kind * (G + 1)) * T + o] = entries of `kind` for owner o, cs[(kind * (G + 1) + 1 + q) * T + o] = those pushed by the sources
// of rank q.  One thread per (kind, owner): neighbouring threads read neighbouring words of a matrix row.
FQ_KERNEL void k_shard_colsum(DevCfg cfg, const u32 *C, u32 *cs) {
  const u32 T = cfg.T, G = cfg.shard_world;
#ifndef FQSX_EMU
  const u32 stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
#else
  const u32 stride = 1, first = 0;
#endif
  for (u32 i = first; i < 3 * T; i += stride) {
    const u32 kind = i / T, o = i % T;
    const u32 *Ck = C + (u64)kind * T * T;
    u32 tot = 0;
    for (u32 q = 0; q < G; ++q) {
      u32 n = 0;
      for (u32 s = q; s < T; s += 8 * G) {   // eight loads in flight (a thread's column walk is a chain of L2 round trips otherwise)
        u32 v[8];
#pragma unroll
        for (u32 x = 0; x < 8; ++x) v[x] = s + x * G < T ? Ck[(u64)(s + x * G) * T + o] : 0u;
#pragma unroll
        for (u32 x = 0; x < 8; ++x) n += v[x];
      }
      cs[((u64)kind * (G + 1) + 1 + q) * T + o] = n;
      tot += n;
    }
    cs[((u64)kind * (G + 1)) * T + o] = tot;
  }
}
FQ_KERNEL64 void k_shard_need(DevCfg cfg, const u32 *cs, u64 *out, u64 *stats_before) {
  const u32 T = cfg.T;
  for (u32 which = 0; which < 2; ++which) {
    const KTab &t = which ? cfg.g_b : cfg.g_s;
    const u32 *tot = cs + (u64)(which ? MAIL_B : MAIL_S) * (cfg.shard_world + 1) * T;
    u64 need = 0;
    for (u32 o = FQ_LANE; o < T; o += FQ_WAVE) {
      const u64 n = (u64)t.filled[o] + tot[o];
      need = n > need ? n : need;
    }
#if FQ_WAVE > 1
    for (int off = 32; off > 0; off >>= 1) { const u64 y = __shfl_xor(need, off, 64); need = y > need ? y : need; }
#endif
    if (FQ_LANE == 0) out[which] = need;
  }
  if (FQ_LANE == 0) out[2] = cfg.err[0];
  if (stats_before)   // (the p-mer statistics before the phase's inserts: what the all-gather's delta is taken against)
    for (u32 i = FQ_LANE; i < 2; i += FQ_WAVE) stats_before[i] = cfg.siv_stats[i];
}
// this rank's paired-end triples, sources in ascending order -> out (the all-gather's payload)
FQ_KERNEL64 void k_shard_pe_pack(DevCfg cfg, u64 *out) {   // grid = T (source)
  const u32 s = FQ_BLOCK;
  if (!shard_mine(cfg, s)) return;
  u64 off = 0;
  for (u32 t = cfg.shard_rank; t < s; t += cfg.shard_world) off += cfg.pe_n[t];
  const u64 n = 3ull * cfg.pe_n[s];
  const u64 *src = cfg.pe_list + (u64)s * cfg.pe_cap * 3;
  for (u64 i = FQ_LANE; i < n; i += FQ_WAVE) out[3 * off + i] = src[i];
}
// ... and the other ranks' triples into the per-source lists, so that the pair-table insert runs over all T sources on
// every rank: its inserts commute and draw nothing, so every replica of the pair table ends up with the same content.
// gathered = [world][stride] words, a rank's triples at word pe_off; pe_cnt[s] = triples of source s (count matrix tail)
FQ_KERNEL64 void k_shard_pe_unpack(DevCfg cfg, const u64 *gathered, u64 stride, u64 pe_off, const u32 *pe_cnt) {   // grid = T (source)
  const u32 s = FQ_BLOCK, q = s % cfg.shard_world;
  if (q == cfg.shard_rank) return;
  u64 off = 0;
  for (u32 t = q; t < s; t += cfg.shard_world) off += pe_cnt[t];
  const u64 n = 3ull * pe_cnt[s];
  const u64 *src = gathered + (u64)q * stride + pe_off + 3 * off;
  u64 *dst = cfg.pe_list + (u64)s * cfg.pe_cap * 3;
  for (u64 i = FQ_LANE; i < n; i += FQ_WAVE) dst[i] = src[i];
  if (FQ_LANE == 0) cfg.pe_n[s] = pe_cnt[s];
}
// p-mer vector statistics: this rank's contribution of the phase (after - before) into the payload ...
FQ_KERNEL64 void k_shard_siv_delta(DevCfg cfg, const u64 *before, u64 *out) {
  for (u32 i = FQ_LANE; i < 2; i += FQ_WAVE) out[i] = cfg.siv_stats[i] - before[i];
}
// ... and, after the all-gather, the sum over the ranks on top of the value before the phase (every rank alike)
FQ_KERNEL64 void k_shard_siv_sum(DevCfg cfg, const u64 *before, const u64 *gathered, u64 stride, u64 off) {
  for (u32 i = FQ_LANE; i < 2; i += FQ_WAVE) {
    u64 v = before[i];
    for (u32 q = 0; q < cfg.shard_world; ++q) v += gathered[(u64)q * stride + off + i];
    cfg.siv_stats[i] = v;
  }
}
// Partitioned tables: a sub-table's occupancy counter is only kept by its owner's rank, but every rank needs all of them
// for the growth rule (k_shard_need).  They ride along with the all-gather: out[which * n_max + j] = filled of this rank's
// j-th own sub-table of the s- (which 0) / b-mer (1) table ...
// (n_arr = 3: the pair table's counters behind them -- a partitioned pair table, cfg.pe_part)
FQ_KERNEL64 void k_shard_fill_pack(DevCfg cfg, u64 *out, u32 n_max, u32 n_arr) {
  for (u32 i = FQ_LANE; i < n_arr * n_max; i += FQ_WAVE) {
    const u32 which = i / n_max, o = cfg.shard_rank + (i % n_max) * cfg.shard_world;
    out[i] = o < cfg.T ? (which == 2 ? cfg.g_pe.filled : which ? cfg.g_b.filled : cfg.g_s.filled)[o] : 0;
  }
}
// ... and the other ranks' counters into this rank's copy of the arrays
FQ_KERNEL64 void k_shard_fill_unpack(DevCfg cfg, const u64 *gathered, u64 stride, u64 off, u32 n_max, u32 n_arr) {
  for (u32 q = 0; q < cfg.shard_world; ++q) {
    if (q == cfg.shard_rank) continue;
    for (u32 i = FQ_LANE; i < n_arr * n_max; i += FQ_WAVE) {
      const u32 which = i / n_max, o = q + (i % n_max) * cfg.shard_world;
      if (o < cfg.T) (which == 2 ? cfg.g_pe.filled : which ? cfg.g_b.filled : cfg.g_s.filled)[o] = (u32)gathered[(u64)q * stride + off + i];
    }
  }
}
// (partitioned tables: the statistics' delta and the counters in one launch, and the sum and the counters behind the all-gather)
FQ_KERNEL64 void k_shard_pack2(DevCfg cfg, const u64 *before, u64 *out_siv, u64 *out_fill, u32 n_max, u32 n_arr) {
  for (u32 i = FQ_LANE; i < 2; i += FQ_WAVE) out_siv[i] = cfg.siv_stats[i] - before[i];
  for (u32 i = FQ_LANE; i < n_arr * n_max; i += FQ_WAVE) {
    const u32 which = i / n_max, o = cfg.shard_rank + (i % n_max) * cfg.shard_world;
    out_fill[i] = o < cfg.T ? (which == 2 ? cfg.g_pe.filled : which ? cfg.g_b.filled : cfg.g_s.filled)[o] : 0;
  }
}
FQ_KERNEL64 void k_shard_unpack2(DevCfg cfg, const u64 *before, const u64 *gathered, u64 stride, u64 off_siv, u64 off_fill, u32 n_max, u32 n_arr) {
  for (u32 i = FQ_LANE; i < 2; i += FQ_WAVE) {
    u64 v = before[i];
    for (u32 q = 0; q < cfg.shard_world; ++q) v += gathered[(u64)q * stride + off_siv + i];
    cfg.siv_stats[i] = v;
  }
  for (u32 q = 0; q < cfg.shard_world; ++q) {
    if (q == cfg.shard_rank) continue;
    for (u32 i = FQ_LANE; i < n_arr * n_max; i += FQ_WAVE) {
      const u32 which = i / n_max, o = q + (i % n_max) * cfg.shard_world;
      if (o < cfg.T) (which == 2 ? cfg.g_pe.filled : which ? cfg.g_b.filled : cfg.g_s.filled)[o] = (u32)gathered[(u64)q * stride + off_fill + i];
    }
  }
}
// Received entries -> this rank's owners' groups in (owner, source, push) order (the order InsertKmersToHT drains
// its column in).  recv = the chunks of ranks 0..G-1 one after the other; the chunk of rank q holds, for every owner
// of this rank in ascending order, the entries of q's sources in ascending order.  C = the summed count matrix.
// cs: the matrix's column sums (k_shard_colsum), or null (the step-wise driver: every wave sums the columns it needs itself)
FQ_DEV void shard_merge_body(const DevCfg &cfg, u32 kind, u32 o, const u64 *recv, const u32 *C, const u32 *cs, u32 *col, u32 *pre, u32 *soff);
FQ_KERNEL64 void k_shard_merge(DevCfg cfg, u32 kind, const u64 *recv, const u32 *C) {   // grid = T (owner)
  FQ_SHARED u32 col[256], pre[256], soff[256];
  shard_merge_body(cfg, kind, FQ_BLOCK, recv, C, nullptr, col, pre, soff);
}
FQ_KERNEL64 void k_shard_merge3(DevCfg cfg, const u64 *recv0, const u64 *recv1, const u64 *recv2, const u32 *C, const u32 *cs) {   // grid = 3 T: (kind, owner)
  FQ_SHARED u32 col[256], pre[256], soff[256];
  const u32 kind = FQ_BLOCK / cfg.T;
  shard_merge_body(cfg, kind, FQ_BLOCK % cfg.T, kind == 0 ? recv0 : kind == 1 ? recv1 : recv2, C, cs, col, pre, soff);
}
FQ_DEV void shard_merge_body(const DevCfg &cfg, u32 kind, u32 o, const u64 *recv, const u32 *C, const u32 *cs, u32 *col, u32 *pre, u32 *soff) {
  const u32 T = cfg.T, G = cfg.shard_world, me = cfg.shard_rank;
  const Mail &m = cfg.mail[kind];
  const u32 *Ck = C + (u64)kind * T * T;
  const u32 n_own = (T - me + G - 1) / G;   // owners of this rank: me, me + G, ...
  // group offsets: this rank's owners in ascending order, the others empty
  u32 dst = 0, tot = 0;
  for (u32 jb = 0; jb < n_own; jb += FQ_WAVE) {
    const u32 j = jb + FQ_LANE, o2 = me + j * G;
    u32 n = 0;
    if (j < n_own) {
      if (cs) n = cs[((u64)kind * (G + 1)) * T + o2];
      else for (u32 s = 0; s < T; ++s) n += Ck[(u64)s * T + o2];
    }
    dst += wave_sum32(j < n_own && o2 < o ? n : 0u);
    tot += wave_sum32(j < n_own && o2 == o ? n : 0u);
  }
  if (FQ_LANE == 0) {
    m.dst_off[o] = dst;
    m.dst_tot[o] = tot;
    if (o == T - 1) m.dst_off[T] = dst + tot;
  }
  if (o % G != me) return;
  // this owner's column and its prefix over the sources
  u32 run = 0;
  for (u32 sb = 0; sb < T; sb += FQ_WAVE) {
    const u32 s = sb + FQ_LANE;
    const u32 v = s < T ? Ck[(u64)s * T + o] : 0u;
    const u32 ex = wave_excl_scan32(v) + run;
    if (s < T) { col[s] = v; pre[s] = ex; }
    run += wave_sum32(v);
  }
  FQ_SYNC();
  u32 chunk0 = 0;   // start of rank q's chunk in recv
  for (u32 q = 0; q < G; ++q) {
    // entries of q's chunk that belong to owners before o, and the chunk's size
    u32 before = 0, size = 0;
    for (u32 jb = 0; jb < n_own; jb += FQ_WAVE) {
      const u32 j = jb + FQ_LANE, o2 = me + j * G;
      u32 n = 0;
      if (j < n_own) {
        if (cs) n = cs[((u64)kind * (G + 1) + 1 + q) * T + o2];
        else for (u32 s = q; s < T; s += G) n += Ck[(u64)s * T + o2];
      }
      before += wave_sum32(j < n_own && o2 < o ? n : 0u);
      size += wave_sum32(j < n_own ? n : 0u);
    }
    u32 src = chunk0 + before;
    for (u32 s = q; s < T; s += G) {   // where source s's entries for this owner start in recv
      if (FQ_LANE == 0) soff[s] = src;
      src += col[s];
    }
    chunk0 += size;
  }
  FQ_SYNC();
  // the copy, one entry per lane whatever source it comes from: entry x of the owner's group belongs to the source s with
  // pre[s] <= x < pre[s] + col[s] (binary search over the prefix; a loop over the sources is T dependent small copies)
  for (u32 x = FQ_LANE; x < tot; x += FQ_WAVE) {
    u32 lo = 0, hi = T;   // the last s with pre[s] <= x (sources without entries share a prefix value: the last of them counts)
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (pre[mid] <= x) lo = mid; else hi = mid;
    }
    while (col[lo] == 0 && lo > 0) --lo;   // (not reached: pre[lo] <= x < tot means the last such source has entries; kept for safety)
    m.sorted[dst + x] = recv[soff[lo] + (x - pre[lo])];
  }
}
// After the insert phase: what the replicas on the other ranks have to take over.  One item per applied entry of this
// rank's owners (duplicates carry the same final value): s-/b-mers the table slot (k-mer << cbits | count), p-mers
// (index << 2 | final 2-bit value).  grid-stride over the merged list.
FQ_KERNEL void k_shard_collect(DevCfg cfg, u32 kind, u64 *out) {
  const Mail &m = cfg.mail[kind];
  const u32 total = m.dst_off[cfg.T];
#ifndef FQSX_EMU
  const u32 stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
#else
  const u32 stride = 1, first = 0;
#endif
  for (u32 e = first; e < total; e += stride) {
    const u64 x = m.sorted[e];
    if (kind == MAIL_P) { out[e] = (x << 2) | siv_test(&cfg, x); continue; }
    const KTab &t = kind == MAIL_S ? cfg.g_s : cfg.g_b;
    const u64 *sl = t.slots + (u64)sb_owner(&cfg, x) * t.stride;
    const u64 item = tab_locate(t, sl, x >> (64 - 2 * t.k)).item;
    out[e] = item;
  }
}
// ... and their application to this rank's replica of another rank's sub-tables (layout-free: find or claim the slot)
FQ_KERNEL void k_shard_apply(DevCfg cfg, u32 kind, const u64 *items, u32 n) {
#ifndef FQSX_EMU
  const u32 stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
#else
  const u32 stride = 1, first = 0;
#endif
  for (u32 e = first; e < n; e += stride) {
    const u64 item = items[e];
    if (kind == MAIL_P && cfg.siv_part) {   // a transition an owner on another rank logged: only the count index is a replica
      if (item) siv_idx_move(cfg, item >> 4, (u32)((item >> 2) & 3), (u32)(item & 3));
      continue;
    }
    if (kind == MAIL_P) {
      const u64 idx = item >> 2, val = item & 3;
      u64 *wp = cfg.siv + (idx >> 5);
      const u32 sh = 2 * (u32)(idx & 31);
      u64 old = *wp;
      for (;;) {   // the field only grows
        if (((old >> sh) & 3) >= val) break;
        const u64 seen = atomic_cas64(wp, old, (old & ~(3ull << sh)) | (val << sh));
        if (seen == old) { siv_idx_move(cfg, idx, (u32)((old >> sh) & 3), (u32)val); break; }
        old = seen;
      }
      continue;
    }
    if (!item) continue;
    const KTab &t = kind == MAIL_S ? cfg.g_s : cfg.g_b;
    const u64 v = item >> t.cbits;
    const u32 sub = sb_owner(&cfg, v << (64 - 2 * t.k));
    bool claimed = false;
    u64 *slot = tab_find_or_claim(t, t.slots + (u64)sub * t.stride, v, item, claimed);
    if (!slot) continue;
    if (claimed) {
#ifndef FQSX_EMU
      atomicAdd(&t.filled[sub], 1u);
#else
      t.filled[sub] += 1;
#endif
      continue;
    }
    u64 old = *(volatile u64 *)slot;   // (counts of a k-mer only grow: the larger value is the later one)
    while ((old & ((1ull << t.cbits) - 1ull)) < (item & ((1ull << t.cbits) - 1ull))) {
      const u64 seen = atomic_cas64(slot, old, item);
      if (seen == old) break;
      old = seen;
    }
  }
}

// ---------------------------------------------------------------------------------------
// backend
#ifndef FQSX_EMU
#define HIPCHK(x)                                                                             \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                 \
      return FQSX_E_HIP;                                                                      \
    }                                                                                         \
  } while (0)
#endif

struct fqsx_dna {
  DevCfg cfg;
  u32 T;
  int device;
#ifndef FQSX_EMU
  hipStream_t stream;
  hipEvent_t ev0, ev1;
#endif
  bool profiling;
  double k_ms[3];
  u64 k_n[3];
  // capacities (host mirror)
  u64 gs_cap, gb_cap, ls_cap, lb_cap, ctx_cap, out_cap, gpe_cap, lpe_cap;
  u32 pe_cap;
  bool paired;
  u32 mail_cap[3];
  u64 dev_bases_cap, dev_off_cap;
  u8 *d_bases;
  u64 *d_off;
  u32 *d_demand;
  u64 *d_lens;
  u8 *d_compact;
  u64 compact_cap, din_cap, dout_cap;
  std::vector<u32> h_demand, h_filled;
  std::vector<u8> h_out;
  std::vector<u64> h_lens;
  std::vector<void *> allocs;
  std::vector<u64> alloc_bytes;   // size of allocs[i]
  u64 dev_bytes, dev_bytes_peak;  // device memory held now / at most so far (fqsx_dna_capacity)
  u32 n_growths;                  // growth events of the global k-mer / pair tables
  u32 tab_small_pct;              // ... while the table is below 256 MB: to this load (40: it doubles; FQSX_TAB_AFTER_PCT sets both)
  u32 tab_load_pct, tab_after_pct;   // a global k-mer sub-table is grown before an insert phase would fill it beyond load_pct %, to a
                                     // capacity the demand fills to after_pct % (FQSX_TAB_LOAD_PCT / FQSX_TAB_AFTER_PCT; 80 / 62)
  u8 *h_pin;          // pinned host scratch for the small device-to-host transfers of the phase loop
  u64 *d_end;         // block epilogue in one transfer: [T] stream lengths, [T] context-table occupancies, error word
  bool filled_valid;  // h_filled holds the context-table occupancies as of the end of the last encoded block
  // the block being processed (block_prepare -> block_segment ... -> block_finish)
  u32 cur_n_reads, cur_S, cur_gen;
  u64 cur_need_lb, cur_need_ls, cur_need_lpe;
  bool cur_decode;
  // sharded mode (fqsx_shard_*): exchange scratch
  u32 shard_rank, shard_world;
  u8 *d_vmap;
  u64 *d_xbuf;        // received entries / upsert items
  u64 xbuf_cap;
  u32 *d_cglob;       // [3][T][T] (+ [T] paired-end triples per source) the all-reduced count matrix of the phase
  u32 *d_colsum;      // [3][world + 1][T] its column sums (k_shard_colsum)
  u64 siv_before[2];
  // native sharded driver (fqsx_shard_attach / fqsx_shard_encode_block)
  fqsx_comm comm;     // the world's collectives (RCCL on the codec's stream, or the caller's)
  bool comm_set, shard_apply_own;
  u64 *d_xrecv[3];    // received mailbox entries per kind
  u64 xrecv_cap[3];
  u64 *d_items, *d_gathered, *d_small;   // all-gather payload of this rank / of all ranks; [0..1] siv statistics before the phase, [2..4] demand + error word
  u64 items_cap, gathered_cap;
  std::vector<u32> h_cglob;
  u64 sh_phases, sh_collectives, sh_a2a_words, sh_gather_words;
  // partitioned tables (fqsx_shard_partition_tables): this rank holds the physical memory of its owners' sub-tables of
  // g_s / g_b and maps the other ranks' next to them (fqsx_vm.h)
  bool part;
  bool part_fallback;   // fqsx_shard_partition_tables found GPUs without peer access: the world keeps replicas
  u64 vm_gran;
  fqsx_vm::FdMesh mesh;
  struct VmTab {
    u8 *va = nullptr;
    u64 va_bytes = 0, chunk_bytes = 0;
    std::vector<fqsx_vm::Handle> h;   // [T] own (created) and imported chunks
    std::vector<u8> mapped;           // [T] chunk o is mapped (and its handle held)
    u64 own_bytes = 0;                // physical memory of this rank's own chunks
    u32 own_mod = 0;                  // chunk c lives on rank c % world (0), or -- the p-mer vector's owner ranges -- (c % own_mod) % world
    bool live = false;
  } vm_s, vm_b, vm_pk, vm_pv, vm_siv; // s-mer and b-mer table; the pair table's key and value arrays; the p-mer vector (4096 chunks)
  u32 *d_plog_n;                      // counter of the p-mer transition log (DevCfg.p_log_n)
  u32 ins_threads;                    // k_insert_phase: 128 = inserting wave + prefetching wave (FQSX_INS_PREFETCH=0: 64, the inserting wave alone)
  u64 vm_own_bytes;   // physical table memory held by this rank
};

namespace {

int dalloc(fqsx_dna *c, void **p, u64 bytes, bool zero) {
  if (bytes == 0) bytes = 8;
#ifndef FQSX_EMU
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    g_err = "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e);
    return FQSX_E_NOMEM;
  }
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, c->stream));
#else
  *p = zero ? calloc(1, bytes) : malloc(bytes);
  if (!*p) { g_err = "host allocation failed"; return FQSX_E_NOMEM; }
#endif
  c->allocs.push_back(*p);
  c->alloc_bytes.push_back(bytes);
  c->dev_bytes += bytes;
  c->dev_bytes_peak = std::max(c->dev_bytes_peak, c->dev_bytes);
  return FQSX_OK;
}
void dfree(fqsx_dna *c, void *p) {
  if (!p) return;
  auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
  if (it != c->allocs.end()) {
    const size_t i = it - c->allocs.begin();
    c->dev_bytes -= c->alloc_bytes[i];
    c->alloc_bytes.erase(c->alloc_bytes.begin() + i);
    c->allocs.erase(it);
  }
#ifndef FQSX_EMU
  (void)hipFree(p);
#else
  free(p);
#endif
}
int dzero(fqsx_dna *c, void *p, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemsetAsync(p, 0, bytes, c->stream));
#else
  (void)c;
  memset(p, 0, bytes);
#endif
  return FQSX_OK;
}
int h2d(fqsx_dna *c, void *d, const void *h, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
#else
  (void)c;
  memcpy(d, h, bytes);
#endif
  return FQSX_OK;
}
int d2d(fqsx_dna *c, void *d, const void *s, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, c->stream));
#else
  (void)c;
  memcpy(d, s, bytes);
#endif
  return FQSX_OK;
}
#define FQSX_PIN_BYTES (64u * 1024u)
// small transfer through the pinned scratch: begin (asynchronous), ... more launches ..., end (waits, copies out)
int d2h_small_begin(fqsx_dna *c, const void *d, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemcpyAsync(c->h_pin, d, bytes, hipMemcpyDeviceToHost, c->stream));
#else
  memcpy(c->h_pin, d, bytes);
#endif
  return FQSX_OK;
}
int d2h_small_end(fqsx_dna *c, void *h, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  memcpy(h, c->h_pin, bytes);
  return FQSX_OK;
}
int d2h_sync(fqsx_dna *c, void *h, const void *d, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
#else
  (void)c;
  memcpy(h, d, bytes);
#endif
  return FQSX_OK;
}

#ifndef FQSX_EMU
#define LAUNCH(c, kidx, kern, grid, block, ...)                                     \
  do {                                                                              \
    if ((c)->profiling) HIPCHK(hipEventRecord((c)->ev0, (c)->stream));              \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); \
    HIPCHK(hipGetLastError());                                                      \
    if ((c)->profiling) {                                                           \
      HIPCHK(hipEventRecord((c)->ev1, (c)->stream));                                \
      HIPCHK(hipEventSynchronize((c)->ev1));                                        \
      float ms_ = 0;                                                                \
      HIPCHK(hipEventElapsedTime(&ms_, (c)->ev0, (c)->ev1));                        \
      (c)->k_ms[kidx] += ms_;                                                       \
      (c)->k_n[kidx] += 1;                                                          \
    }                                                                               \
  } while (0)
#else
#define LAUNCH(c, kidx, kern, grid, block, ...)       \
  do {                                                \
    fq_emu_nblocks = (grid);                          \
    for (u32 b_ = 0; b_ < (u32)(grid); ++b_) {        \
      fq_emu_block = b_;                              \
      kern(__VA_ARGS__);                              \
    }                                                 \
    (c)->k_n[kidx] += 1;                              \
  } while (0)
#endif

#ifndef FQSX_EMU
#define REHASH_GRID 2048
#else
#define REHASH_GRID 1  /* the emulated kernel body already walks every slot */
#endif

// The T streams a quality / id kernel left in [T][out_cap] (lengths lens_host, already read back and checked against out_cap)
// -> one contiguous host buffer: a compaction launch and ONE transfer instead of a transfer and a synchronisation per worker
int collect_streams(fqsx_dna *c, u32 T, const u8 *d_out, u64 out_cap, const u64 *d_lens, const std::vector<u64> &lens_host,
                    std::vector<u8> &h_out, const u8 **streams, u64 *lens) {
  int rc;
  void *p = nullptr;
  u64 total = 0;
  for (u32 t = 0; t < T; ++t) total += lens_host[t];
  if (total > c->compact_cap) {
    dfree(c, c->d_compact);
    c->d_compact = nullptr; c->compact_cap = 0;
    if ((rc = dalloc(c, &p, total + total / 2 + 4096, false))) return rc;
    c->d_compact = (u8 *)p;
    c->compact_cap = total + total / 2 + 4096;
  }
  h_out.resize(total ? total : 1);
  if (total) {
    LAUNCH(c, 2, k_compact_generic, T, 64, d_out, out_cap, d_lens, c->d_compact);
    if ((rc = d2h_sync(c, h_out.data(), c->d_compact, total))) return rc;
  }
  u64 pos = 0;
  for (u32 t = 0; t < T; ++t) {
    streams[t] = h_out.data() + pos;
    lens[t] = lens_host[t];
    pos += lens_host[t];
  }
  return FQSX_OK;
}

u64 pow2_at_least(u64 x) {
  u64 p = 1;
  while (p < x) p <<= 1;
  return p;
}

// growth policy of the global k-mer tables (bucketed two-hash probing, fqsx_layout.h: KTab)
bool tab_over(const fqsx_dna *c, u64 need, u64 cap) { return need * 100 > cap * c->tab_load_pct; }
u64 tab_new_cap(const fqsx_dna *c, u64 need) {
  // (while a table is small its density is nobody's concern, but every growth stops the block's queue and costs a host round trip:
  // below 256 MB it grows to 40 % load, i.e. doubles)
  const u32 after = need * c->T * sizeof(u64) * 100 / 40 < (256ull << 20) ? std::min<u32>(c->tab_small_pct, c->tab_after_pct) : c->tab_after_pct;
  u64 cap = (need * 100 + after - 1) / after + 2 * FQSX_BKT;
  return (cap + FQSX_BKT - 1) / FQSX_BKT * FQSX_BKT;
}

KGeom make_geom(u32 k) {  // kmer.h:279-298
  KGeom g;
  g.k = k;
  g.shift = 64 - 2 * k;
  g.mask = (~0ull) << g.shift;
  g.kernel_mask = ((1ull << (2 * k - 8)) - 1ull) << (64 - 2 * k + 4);
  return g;
}

void mt_seed(u32 *s, u32 seed) {
  s[0] = seed;
  for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (u32)i;
}

// (re)allocate a k-mer table of n_sub sub-tables with `cap` slots each, empty
int ktab_alloc(fqsx_dna *c, KTab &t, u32 n_sub, u64 cap, u32 k, u32 cbits, bool with_filled) {
  void *p = nullptr;
  int rc = dalloc(c, &p, cap * n_sub * sizeof(u64), true);
  if (rc) return rc;
  t.slots = (u64 *)p;
  t.nb = cap / FQSX_BKT;   // (cap: a multiple of FQSX_BKT, at least two buckets)
  t.stride = cap;
  t.k = k;
  t.cbits = cbits;
  if (with_filled) {
    rc = dalloc(c, &p, (u64)n_sub * sizeof(u32), true);
    if (rc) return rc;
    t.filled = (u32 *)p;
  }
  return FQSX_OK;
}

// ---- partitioned tables: one address range, the sub-tables' memory spread over the ranks (fqsx_vm.h) -------------------
#define VMCHK(x) do { std::string e_; if (g_vm_dbg) fprintf(stderr, "[fqsx vm] %s\n", #x); if ((x)) { g_err = "partitioned tables: " + e_; return FQSX_E_HIP; } } while (0)
static const bool g_vm_dbg = getenv("FQSX_VM_DEBUG") != nullptr;
// one chunk out of the range (its memory goes back to the device once every rank that imported it has let go of it too)
u32 vt_rank(const fqsx_dna *c, const fqsx_dna::VmTab &v, u32 o) { return (v.own_mod ? o % v.own_mod : o) % c->shard_world; }
void vtab_drop(fqsx_dna *c, fqsx_dna::VmTab &v, u32 o) {
  if (!v.live || !v.mapped[o]) return;
  std::string e;
  (void)fqsx_vm::unmap(v.va + (u64)o * v.chunk_bytes, v.chunk_bytes, e);
  (void)fqsx_vm::release(v.h[o], e);
  v.mapped[o] = 0;
  if (vt_rank(c, v, o) == c->shard_rank) { c->dev_bytes -= v.chunk_bytes; c->vm_own_bytes -= v.chunk_bytes; v.own_bytes -= v.chunk_bytes; }
}
void vtab_free(fqsx_dna *c, fqsx_dna::VmTab &v) {
  if (!v.live) return;
  std::string e;
  for (u32 o = 0; o < (u32)v.h.size(); ++o) vtab_drop(c, v, o);
#ifdef FQSX_EMU
  (void)fqsx_vm::unreserve(v.va, v.va_bytes, e);
#else
  // (Round 4 tried to use kept ranges again for later tables of the process -- mapping new chunks of another size into a range
  // that had stayed reserved: the runtime aborted inside the first encode launch behind such a reuse,
  // tests/test_gpu_parity.py::test_hip_chunked_tables_grow_sub_table_by_sub_table.  So a range is used once.)
  // The address range is NOT handed back (hipMemAddressFree): with the HIP runtime a PyTorch wheel loads (ROCm 7.0) a range
  // that is freed and later reserved again -- by the next growth, or by another allocation that lands there -- is read through
  // stale translations (found with tests/test_gpu_parity.py::test_hip_chunked_tables_*: wrong table contents from the first
  // reuse on; never with the ranges kept).  Unmapped address space costs nothing; a codec's ranges add up to less than twice
  // its final tables.
#endif
  v = fqsx_dna::VmTab();
}
// the address range of T sub-tables with `cap` 8-byte slots each (*stride: slots from one sub-table to the next); nothing mapped yet
int vtab_reserve_raw(fqsx_dna *c, fqsx_dna::VmTab &v, u64 cap, u64 *stride_out, u32 n_chunks = 0) {
  const u32 T = n_chunks ? n_chunks : c->T;
#ifndef FQSX_EMU
  // a chunk is at least 2 MiB (and 2 MiB-aligned, below): with 4 KiB-granular chunks the 1 M-read file ran 7 % slower than on
  // hipMalloc'ed tables (blocks 0-69: 10 %) -- page-table fragments of the size of the chunk
  const u64 min_chunk = std::max<u64>(c->vm_gran, 2ull << 20);
#else
  const u64 min_chunk = c->vm_gran;
#endif
  // (a sub-table is one chunk: its capacity rounded up to the chunk granule -- at most 2 MiB of slack per sub-table)
  const u64 gran_slots = min_chunk / sizeof(u64);
  const u64 stride = (cap + gran_slots - 1) / gran_slots * gran_slots;
  v = fqsx_dna::VmTab();
  v.chunk_bytes = stride * sizeof(u64);
  v.va_bytes = v.chunk_bytes * T;
  v.h.assign(T, fqsx_vm::Handle());
  v.mapped.assign(T, 0);
  // (2 MiB alignment once the chunks are that large, so that the driver can use large page-table fragments)
  VMCHK(fqsx_vm::reserve(v.va_bytes, std::max<u64>(c->vm_gran, std::min<u64>(v.chunk_bytes, 2ull << 20)), &v.va, e_));
  v.live = true;
  *stride_out = stride;
  return FQSX_OK;
}
int vtab_reserve(fqsx_dna *c, KTab &t, fqsx_dna::VmTab &v, u64 cap, u32 k, u32 cbits) {
  u64 stride = 0;
  const int rc = vtab_reserve_raw(c, v, cap, &stride);
  if (rc) return rc;
  t.slots = (u64 *)v.va;
  t.nb = cap / FQSX_BKT;
  t.stride = stride;
  t.k = k;
  t.cbits = cbits;
  return FQSX_OK;
}
// this rank's sub-table o: physical memory, mapped, empty
int vtab_create_own(fqsx_dna *c, fqsx_dna::VmTab &v, u32 o) {
  VMCHK(fqsx_vm::create(c->device, v.chunk_bytes, &v.h[o], e_));
  c->dev_bytes += v.chunk_bytes; c->vm_own_bytes += v.chunk_bytes; v.own_bytes += v.chunk_bytes;
  c->dev_bytes_peak = std::max(c->dev_bytes_peak, c->dev_bytes);
  VMCHK(fqsx_vm::map(c->device, v.va + (u64)o * v.chunk_bytes, v.chunk_bytes, v.h[o], e_));
  v.mapped[o] = 1;
  const u64 words = v.chunk_bytes / sizeof(u64);   // (a peer reads the chunk only after a collective that follows on this stream)
  LAUNCH(c, 2, k_zero_words, (u32)std::min<u64>(REHASH_GRID, (words + 255) / 256), 256, (u64 *)(v.va + (u64)o * v.chunk_bytes), words, c->cfg.sys_scope);
  return FQSX_OK;
}
// Collective: every rank hands the descriptors of its own chunks to every other rank and maps what it receives
int vtab_exchange(fqsx_dna *c, fqsx_dna::VmTab &v) {
  const u32 T = (u32)v.h.size(), G = c->shard_world, me = c->shard_rank;
  if (G == 1) return FQSX_OK;
  // (in rounds of at most 128 chunks per rank, so that a table of thousands of chunks -- the p-mer vector -- never has more
  // descriptors open at once than a process may hold; every rank walks the same rounds)
  std::vector<std::vector<u32>> of(G);
  for (u32 o = 0; o < T; ++o) of[vt_rank(c, v, o)].push_back(o);
  size_t longest = 0;
  for (u32 q = 0; q < G; ++q) longest = std::max(longest, of[q].size());
  const size_t R = 128;
  for (size_t r0 = 0; r0 < longest; r0 += R) {
    std::vector<int> mine;
    for (size_t j = r0; j < std::min(of[me].size(), r0 + R); ++j) {
      int fd = -1;
      VMCHK(fqsx_vm::export_fd(v.h[of[me][j]], &fd, e_));
      mine.push_back(fd);
    }
    if (!mine.empty())
      for (u32 i = 1; i < G; ++i) VMCHK(fqsx_vm::send_fds(c->mesh.peer[(me + i) % G], mine.data(), (u32)mine.size(), e_));
    for (int fd : mine) close(fd);
    for (u32 i = 1; i < G; ++i) {
      const u32 q = (me + G - i) % G;
      if (r0 >= of[q].size()) continue;
      const u32 n_q = (u32)(std::min(of[q].size(), r0 + R) - r0);
      std::vector<int> theirs(n_q, -1);
      VMCHK(fqsx_vm::recv_fds(c->mesh.peer[q], theirs.data(), n_q, e_));
      for (u32 j = 0; j < n_q; ++j) {
        const u32 o = of[q][r0 + j];
        VMCHK(fqsx_vm::import_fd(theirs[j], &v.h[o], e_));
        close(theirs[j]);
        VMCHK(fqsx_vm::map(c->device, v.va + (u64)o * v.chunk_bytes, v.chunk_bytes, v.h[o], e_));
        v.mapped[o] = 1;
      }
    }
  }
  return FQSX_OK;
}
// Collective: an empty table
int vtab_alloc(fqsx_dna *c, KTab &t, fqsx_dna::VmTab &v, u64 cap, u32 k, u32 cbits) {
  int rc = vtab_reserve(c, t, v, cap, k, cbits);
  for (u32 o = c->shard_rank; !rc && o < c->T; o += c->shard_world) rc = vtab_create_own(c, v, o);
  return rc ? rc : vtab_exchange(c, v);
}

int grow_global(fqsx_dna *c, KTab &t, u64 &cap_field, u64 new_cap) {
  KTab n = t;
  int rc;
  fqsx_dna::VmTab &v = &t == &c->cfg.g_s ? c->vm_s : c->vm_b;
  // A table of one GPU that reaches 2 GiB (FQSX_CHUNK_AUTO_KB: another size, 0 = never) becomes a chunked table at that growth
  // by itself: from then on its growths never hold the old and the new table side by side (fqsx_dna_use_chunked_tables does
  // it from the start).
  const char *auto_env = getenv("FQSX_CHUNK_AUTO_KB");
  const u64 auto_bytes = (auto_env ? strtoull(auto_env, nullptr, 10) : (2048ull << 10)) << 10;
  bool chunked = c->part || v.live;
  if (!chunked && c->shard_world == 1 && auto_bytes && new_cap * c->T * sizeof(u64) >= auto_bytes) {
    if (!c->vm_gran) VMCHK(fqsx_vm::granularity(c->device, &c->vm_gran, e_));
    chunked = c->vm_gran >= sizeof(u64) && !(c->vm_gran & (c->vm_gran - 1));
  }
  if (chunked) {
    // Chunked tables (partitioned over the ranks, or one GPU's capacity mode).  Every rank takes the same decision in the
    // same phase (the demand follows from the all-reduced counts and the exchanged occupancies), so the descriptor exchange
    // is collective; each rank re-inserts its own sub-tables, and the phase's all-gather orders that before anybody's next
    // look-up.  Sub-table by sub-table -- new chunk, re-insert, old chunk back to the device -- so that the old and the new
    // table are never alive side by side: the peak is the new table plus one step's old sub-tables.
    fqsx_dna::VmTab nv;
    const bool old_chunked = v.live;   // (false once: the growth at which a plain table turns into a chunked one)
    if ((rc = vtab_reserve(c, n, nv, new_cap, t.k, t.cbits))) return rc;
    // (several sub-tables per step while they are small -- at least 256 MB of new chunks per step, all of them for a small
    // table: a step costs a memory-create / map / synchronise / unmap round of about a millisecond)
    const u32 G = c->shard_world, n_own = (c->T - c->shard_rank + G - 1) / G;
    const char *step_env = getenv("FQSX_CHUNK_STEP_KB");   // (tests: 1 = one sub-table per step whatever its size)
    const u64 step_bytes = (step_env ? strtoull(step_env, nullptr, 10) : (256ull << 10)) << 10;
    const u32 per_step = (u32)std::max<u64>(1, std::min<u64>(n_own, step_bytes / nv.chunk_bytes));
    for (u32 j0 = 0; j0 < n_own; j0 += per_step) {
      const u32 nb = std::min(per_step, n_own - j0), first = c->shard_rank + j0 * G;
      for (u32 j = 0; j < nb; ++j)
        if ((rc = vtab_create_own(c, nv, first + j * G))) return rc;
      LAUNCH(c, 2, k_rehash_ktab, REHASH_GRID, 256, t, n, nb, first, G, c->cfg.sys_scope);
#ifndef FQSX_EMU
      HIPCHK(hipStreamSynchronize(c->stream));
#endif
      if (old_chunked)
        for (u32 j = 0; j < nb; ++j) vtab_drop(c, v, first + j * G);
    }
    if ((rc = vtab_exchange(c, nv))) return rc;
    if (old_chunked) vtab_free(c, v); else dfree(c, t.slots);
    v = nv;
  } else {
    if ((rc = ktab_alloc(c, n, c->T, new_cap, t.k, t.cbits, false))) return rc;
    LAUNCH(c, 2, k_rehash_ktab, REHASH_GRID, 256, t, n, c->T, 0u, 1u, 0u);
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
    dfree(c, t.slots);
  }
  t = n;
  cap_field = new_cap;
  c->n_growths += 1;
  return FQSX_OK;
}

int ptab_alloc(fqsx_dna *c, PTab &t, u32 n_sub, u64 cap, bool with_filled) {
  void *p = nullptr;
  int rc;
  if ((rc = dalloc(c, &p, cap * n_sub * sizeof(u64), true))) return rc;
  t.key = (u64 *)p;
  if ((rc = dalloc(c, &p, cap * n_sub * sizeof(u64), true))) return rc;
  t.val = (u64 *)p;
  t.cap_mask = cap - 1;
  t.stride = cap;
  if (with_filled) {
    if ((rc = dalloc(c, &p, (u64)n_sub * sizeof(u32), true))) return rc;
    t.filled = (u32 *)p;
  }
  return FQSX_OK;
}
// the pair table as chunked table: key and value array in one address range each, one chunk per sub-table (both arrays have the
// same stride, so the kernels index them alike); Collective in a world of several ranks
int ptab_reserve(fqsx_dna *c, PTab &t, fqsx_dna::VmTab &vk, fqsx_dna::VmTab &vv, u64 cap) {
  u64 sk = 0, sv = 0;
  int rc;
  if ((rc = vtab_reserve_raw(c, vk, cap, &sk)) || (rc = vtab_reserve_raw(c, vv, cap, &sv))) return rc;
  t.key = (u64 *)vk.va;
  t.val = (u64 *)vv.va;
  t.cap_mask = cap - 1;
  t.stride = sk;
  return FQSX_OK;
}
int grow_gpe(fqsx_dna *c, u64 new_cap) {
  PTab n = c->cfg.g_pe;
  int rc;
  const char *auto_env = getenv("FQSX_CHUNK_AUTO_KB");
  const u64 auto_bytes = (auto_env ? strtoull(auto_env, nullptr, 10) : (2048ull << 10)) << 10;
  bool chunked = c->part || c->vm_pk.live;
  if (!chunked && c->shard_world == 1 && auto_bytes && new_cap * c->T * sizeof(u64) >= auto_bytes) {   // (as grow_global)
    if (!c->vm_gran) VMCHK(fqsx_vm::granularity(c->device, &c->vm_gran, e_));
    chunked = c->vm_gran >= sizeof(u64) && !(c->vm_gran & (c->vm_gran - 1));
  }
  if (chunked) {   // sub-table by sub-table, every rank its own (see grow_global)
    fqsx_dna::VmTab nk, nv;
    const bool old_chunked = c->vm_pk.live;
    if ((rc = ptab_reserve(c, n, nk, nv, new_cap))) return rc;
    const u32 G = c->shard_world, n_own = (c->T - c->shard_rank + G - 1) / G;
    const char *step_env = getenv("FQSX_CHUNK_STEP_KB");
    const u64 step_bytes = (step_env ? strtoull(step_env, nullptr, 10) : (256ull << 10)) << 10;
    const u32 per_step = (u32)std::max<u64>(1, std::min<u64>(n_own, step_bytes / (2 * nk.chunk_bytes)));
    for (u32 j0 = 0; j0 < n_own; j0 += per_step) {
      const u32 nb = std::min(per_step, n_own - j0), first = c->shard_rank + j0 * G;
      for (u32 j = 0; j < nb; ++j)
        if ((rc = vtab_create_own(c, nk, first + j * G)) || (rc = vtab_create_own(c, nv, first + j * G))) return rc;
      LAUNCH(c, 2, k_rehash_ptab, REHASH_GRID, 256, c->cfg.g_pe, n, nb, first, G, c->cfg.sys_scope);
#ifndef FQSX_EMU
      HIPCHK(hipStreamSynchronize(c->stream));
#endif
      if (old_chunked)
        for (u32 j = 0; j < nb; ++j) { vtab_drop(c, c->vm_pk, first + j * G); vtab_drop(c, c->vm_pv, first + j * G); }
    }
    if ((rc = vtab_exchange(c, nk)) || (rc = vtab_exchange(c, nv))) return rc;
    if (old_chunked) { vtab_free(c, c->vm_pk); vtab_free(c, c->vm_pv); }
    else { dfree(c, c->cfg.g_pe.key); dfree(c, c->cfg.g_pe.val); }
    c->vm_pk = nk;
    c->vm_pv = nv;
  } else {
    if ((rc = ptab_alloc(c, n, c->T, new_cap, false))) return rc;
    LAUNCH(c, 2, k_rehash_ptab, REHASH_GRID, 256, c->cfg.g_pe, n, c->T, 0u, 1u, 0u);
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
    dfree(c, c->cfg.g_pe.key);
    dfree(c, c->cfg.g_pe.val);
  }
  c->cfg.g_pe = n;
  c->gpe_cap = new_cap;
  c->n_growths += 1;
  return FQSX_OK;
}

int grow_ctx(fqsx_dna *c, u64 new_cap) {
  void *p = nullptr;
  int rc = dalloc(c, &p, new_cap * c->T * sizeof(CtxSlot), true);
  if (rc) return rc;
  LAUNCH(c, 2, k_rehash_ctx, REHASH_GRID, 256, (const u64 *)c->cfg.ctx, c->cfg.ctx_cap_mask, (u64 *)p, new_cap - 1, c->T);
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  dfree(c, c->cfg.ctx);
  c->cfg.ctx = (CtxSlot *)p;
  c->cfg.ctx_cap_mask = new_cap - 1;
  c->ctx_cap = new_cap;
  return FQSX_OK;
}

int mail_alloc(fqsx_dna *c, u32 kind, u32 cap) {
  Mail &m = c->cfg.mail[kind];
  const u32 T = c->T;
  if (m.list) { dfree(c, m.list); dfree(c, m.sorted); dfree(c, m.tile_hist); }
  cap = (cap + FQSX_TILE - 1) / FQSX_TILE * FQSX_TILE;
  void *p = nullptr;
  int rc;
  if ((rc = dalloc(c, &p, (u64)T * cap * sizeof(u64), false))) return rc;
  m.list = (u64 *)p;
  if ((rc = dalloc(c, &p, (u64)T * cap * sizeof(u64), false))) return rc;
  m.sorted = (u64 *)p;
  m.cap = cap;
  m.n_tiles = cap / FQSX_TILE;
  if ((rc = dalloc(c, &p, (u64)T * m.n_tiles * T * sizeof(u32), false))) return rc;
  m.tile_hist = (u32 *)p;
  c->mail_cap[kind] = cap;
  return FQSX_OK;
}

// one encode / decode launch over segment `seg` (T workgroups); timed with HIP events when profiling is on
int launch_segment(fqsx_dna *c, bool decode, u32 n_reads, u32 S, u32 seg) {
  EncArgs a;
  a.cfg = c->cfg;
  a.n_reads = n_reads; a.S = S; a.seg = seg; a.pad = (u32)c->k_n[0];   // (launch index: time stamps of the timing build)
#ifndef FQSX_EMU
  if (c->profiling) HIPCHK(hipEventRecord(c->ev0, c->stream));
  const int e = decode ? fqsx_launch_decode(c->stream, a) : c->paired ? fqsx_launch_encode_pe(c->stream, a) : fqsx_launch_encode_se(c->stream, a);
  if (e) { g_err = std::string("encode/decode kernel launch: ") + hipGetErrorString((hipError_t)e); return FQSX_E_HIP; }
  if (c->profiling) {
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->k_ms[0] += ms;
  }
#else
  if (decode) fqsx_emu_decode(a); else if (c->paired) fqsx_emu_encode_pe(a); else fqsx_emu_encode_se(a);
#endif
  c->k_n[0] += 1;
  return FQSX_OK;
}

// Sizes the per-block buffers and fixes the block's schedule (number of synchronisation points S)
int block_prepare(fqsx_dna *c, const u8 *d_bases, const u64 *d_off, const u64 *h_off, u32 n_reads, u32 generation,
                  const u8 *const *dec_streams = nullptr, const u64 *dec_lens = nullptr) {
  const bool decode = dec_streams != nullptr;
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  // ---- schedule (PartitionForWorkers reads_block.h:197-214; calc_no_synchronizations application.h:85-92)
  u64 S = generation < 100u ? 100u - generation : 0u;
  u64 cap_s = (u64)n_reads / T / 2;
  if (S > cap_s) S = cap_s;
  if (S) --S;
  u64 max_wbases = 0, max_seg_bases = 0, max_seg_reads = 0, max_wreads = 0;
  std::vector<u64> wbases(T);
  for (u32 t = 0; t < T; ++t) {
    u64 first = (u64)t * n_reads / T, last = ((u64)t + 1) * n_reads / T;
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    wbases[t] = h_off[last] - h_off[first];
    max_wbases = std::max(max_wbases, wbases[t]);
    max_wreads = std::max(max_wreads, last - first);
    u64 cur = first;
    for (u64 seg = 0; seg <= S; ++seg) {
      u64 stop = last;
      if (seg < S) {
        const u64 ns = (seg + 1) * (last - first) / (S + 1) + first;
        if (!c->paired) stop = ns + 1;
        else { u64 i = cur; if (i < ns) i += (ns - i + 1) & ~1ull; stop = i + 2; }
      }
      if (stop > last) stop = last;
      if (stop > cur) {
        max_seg_bases = std::max(max_seg_bases, h_off[stop] - h_off[cur]);
        max_seg_reads = std::max(max_seg_reads, stop - cur);
        cur = stop;
      }
    }
  }
  int rc;
  void *p = nullptr;
  if (c->paired && (n_reads & 1)) { g_err = "paired-end blocks need an even number of reads"; return FQSX_E_ARG; }
  // ---- per-block buffers
  u64 need_out = max_wbases + 16 * max_wreads + 1024;
  if (need_out > c->out_cap) {
    if (cfg.out) dfree(c, cfg.out);
    c->out_cap = pow2_at_least(need_out);
    if ((rc = dalloc(c, &p, c->out_cap * T, false))) return rc;
    cfg.out = (u8 *)p;
    cfg.out_cap = c->out_cap;
  }
  const u64 mail_entries[3] = {2 * max_seg_bases + 2 * max_seg_reads, max_seg_bases, 2 * max_seg_bases};
  for (u32 k = 0; k < 3; ++k) {
    u64 need = mail_entries[k] + 64;
    if (need > c->mail_cap[k] && (rc = mail_alloc(c, k, (u32)(need + need / 4)))) return rc;
  }
  u64 need_lb = pow2_at_least(4 * max_seg_bases + 64), need_ls = pow2_at_least(2 * max_seg_bases + 64);
  if (need_lb > c->lb_cap) {
    dfree(c, cfg.l_b.slots);
    if ((rc = ktab_alloc(c, cfg.l_b, T, need_lb, cfg.bmer, 6, false))) return rc;
    c->lb_cap = need_lb;
  }
  if (need_ls > c->ls_cap) {
    dfree(c, cfg.l_s.slots);
    if ((rc = ktab_alloc(c, cfg.l_s, T, need_ls, cfg.smer, 12, false))) return rc;
    c->ls_cap = need_ls;
  }
  u64 need_lpe = 0;
  if (c->paired) {
    const u64 seg_pairs = max_seg_reads / 2 + 1;
    const u64 need_list = 14 * seg_pairs + 16;
    if (need_list > c->pe_cap) {
      if (cfg.pe_list) dfree(c, cfg.pe_list);
      c->pe_cap = (u32)(need_list + need_list / 4);
      if ((rc = dalloc(c, &p, (u64)T * c->pe_cap * 3 * sizeof(u64), false))) return rc;
      cfg.pe_list = (u64 *)p;
      cfg.pe_cap = c->pe_cap;
      if (cfg.pe_bkt) dfree(c, cfg.pe_bkt);
      if ((rc = dalloc(c, &p, (u64)T * c->pe_cap * 3 * sizeof(u64), false))) return rc;
      cfg.pe_bkt = (u64 *)p;
    }
    need_lpe = pow2_at_least(2 * 14 * seg_pairs + 64);
    if (need_lpe > c->lpe_cap) {
      dfree(c, cfg.l_pe.key);
      dfree(c, cfg.l_pe.val);
      if ((rc = ptab_alloc(c, cfg.l_pe, T, need_lpe, false))) return rc;
      c->lpe_cap = need_lpe;
    }
    cfg.l_pe.cap_mask = need_lpe - 1;
    cfg.l_pe.stride = need_lpe;
    // mates longer than the LDS staging size: three scratch lines per worker (codes of either mate, reverse-complement line)
    u64 max_len = 0;
    for (u32 i = 0; i < n_reads; ++i) max_len = std::max(max_len, h_off[i + 1] - h_off[i]);
    if (!decode && max_len > FQSX_RD_LDS && max_len + 128 > cfg.pe_scr_cap) {
      if (cfg.pe_scr) dfree(c, cfg.pe_scr);
      const u64 cap = (max_len + max_len / 4 + 128 + 63) & ~63ull;
      if ((rc = dalloc(c, &p, (u64)T * 3 * cap, true))) return rc;
      cfg.pe_scr = (u8 *)p;
      cfg.pe_scr_cap = cap;
    }
  }
  // active geometry of the local tables for this block (cleared after every phase)
  cfg.l_b.nb = need_lb / FQSX_BKT; cfg.l_b.stride = need_lb;
  cfg.l_s.nb = need_ls / FQSX_BKT; cfg.l_s.stride = need_ls;
  // ---- context tables: every coded symbol creates at most two contexts
  if (!c->filled_valid && (rc = d2h_sync(c, c->h_filled.data(), cfg.ctx_filled, T * sizeof(u32)))) return rc;
  c->filled_valid = false;
  {
    u64 need = 0;
    for (u32 t = 0; t < T; ++t) need = std::max<u64>(need, (u64)c->h_filled[t] + 2 * wbases[t] + 64);
#ifdef FQSX_TIMING
    if (generation % 64 == 63) fprintf(stderr, "[timing] block %u: ctx slots used by worker 0: %u, capacity %llu\n", generation, c->h_filled[0], (unsigned long long)c->ctx_cap);
#endif
    if (need * 2 > c->ctx_cap && (rc = grow_ctx(c, pow2_at_least(need * 2)))) return rc;
  }
  cfg.bases = d_bases;
  cfg.read_off = d_off;
  if (decode) {  // upload the streams, size the output block and the per-worker code lines
    std::vector<u64> doff(T + 1, 0);
    for (u32 t = 0; t < T; ++t) doff[t + 1] = doff[t] + dec_lens[t];
    u64 max_len = 0;
    for (u32 i = 0; i < n_reads; ++i) max_len = std::max(max_len, h_off[i + 1] - h_off[i]);
    const u64 need_in = doff[T] + 64, need_out = h_off[n_reads] + 64, need_sc = max_len + 64;
    if (need_in > c->din_cap) {
      dfree(c, (void *)cfg.din);
      if ((rc = dalloc(c, &p, need_in + need_in / 4, false))) return rc;
      cfg.din = (const u8 *)p; c->din_cap = need_in + need_in / 4;
    }
    if (need_out > c->dout_cap) {
      dfree(c, cfg.dout);
      if ((rc = dalloc(c, &p, need_out + need_out / 4, false))) return rc;
      cfg.dout = (u8 *)p; c->dout_cap = need_out + need_out / 4;
    }
    if (need_sc > cfg.dcap) {
      dfree(c, cfg.dscratch);
      if ((rc = dalloc(c, &p, (u64)T * 2 * (need_sc + need_sc / 4), false))) return rc;
      cfg.dscratch = (u8 *)p; cfg.dcap = need_sc + need_sc / 4;
    }
    if (!cfg.din_off && (rc = dalloc(c, &p, ((u64)T + 1) * sizeof(u64), false))) return rc;
    if (!cfg.din_off) cfg.din_off = (const u64 *)p;
    std::vector<u8> flat(doff[T] ? doff[T] : 1);
    for (u32 t = 0; t < T; ++t) memcpy(flat.data() + doff[t], dec_streams[t], dec_lens[t]);
    if ((rc = h2d(c, (void *)cfg.din, flat.data(), doff[T]))) return rc;
    if ((rc = h2d(c, (void *)cfg.din_off, doff.data(), (T + 1) * sizeof(u64)))) return rc;
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
  }

  c->cur_n_reads = n_reads; c->cur_S = (u32)S; c->cur_gen = generation;
  c->cur_need_lb = need_lb; c->cur_need_ls = need_ls; c->cur_need_lpe = need_lpe;
  c->cur_decode = decode;
  return FQSX_OK;
}

// ClearKmersToHT, dna.cpp:2475-2488 (mailbox counters are rewritten by the next encode launch)
int clear_local_tables(fqsx_dna *c) {
  const u64 words = (c->cur_need_lb + c->cur_need_ls) * c->T;
  const u32 grid = (u32)std::min<u64>(2048, (words + 4095) / 4096 + 1);
  LAUNCH(c, 2, k_clear_local, grid, 64, c->cfg, c->cur_need_lb * c->T, c->cur_need_ls * c->T);
  return FQSX_OK;
}

// insert phase and ClearKmersToHT in one launch (single-end encoding)
int insert_and_clear(fqsx_dna *c) {
  const u64 words = (c->cur_need_lb + c->cur_need_ls) * c->T;
  const u32 cgrid = (u32)std::min<u64>(2048, (words + 4095) / 4096 + 1);
  LAUNCH(c, 1, k_insert_phase, 3 * c->T + cgrid, c->ins_threads, c->cfg, c->cur_need_lb * c->T, c->cur_need_ls * c->T);
  return FQSX_OK;
}
// growth decision from the demand words (k_part_dstoff): a sub-table is at most half full after the coming inserts
int grow_for_demand(fqsx_dna *c) {
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  for (int which = 0; which < 2; ++which) {
    KTab &t = which ? cfg.g_b : cfg.g_s;
    u64 &cap = which ? c->gb_cap : c->gs_cap;
    u64 need = 0;
    for (u32 o = 0; o < T; ++o) need = std::max<u64>(need, (u64)c->h_demand[2 * T + 1 + which * T + o] + c->h_demand[which * T + o]);
    if (tab_over(c, need, cap) && (rc = grow_global(c, t, cap, tab_new_cap(c, need)))) return rc;
  }
  return FQSX_OK;
}
// The device stopped the block's queue before the inserts of segment `seg` (phase_skip): grow, then take up from there
int pe_insert_and_clear(fqsx_dna *c);
int pe_clear_local(fqsx_dna *c);
int block_recover(fqsx_dna *c, u32 seg) {
  const u32 T = c->T;
  int rc;
  if ((rc = d2h_sync(c, c->h_demand.data(), c->d_demand, (4 * T + 1) * sizeof(u32)))) return rc;
  const u64 before = c->gs_cap + c->gb_cap + c->gpe_cap;
  if ((rc = grow_for_demand(c))) return rc;
  if ((rc = dzero(c, c->cfg.err + 1, sizeof(u32)))) return rc;
  if (c->paired) {   // the pair table's demand again (the queued count stopped at the posted word), growth, then the phase's inserts
    if ((rc = dzero(c, c->cfg.pe_bkt_n, T * sizeof(u32)))) return rc;   // (the per-owner counts of the stopped pass: the next phase counts from zero)
    LAUNCH(c, 2, k_pe_demand, T, 64, c->cfg, c->d_demand + 4 * T + 1, 0u);
    std::vector<u32> dem(T), fil(T);
    if ((rc = d2h_sync(c, dem.data(), c->d_demand + 4 * T + 1, T * sizeof(u32)))) return rc;
    if ((rc = d2h_sync(c, fil.data(), c->cfg.g_pe.filled, T * sizeof(u32)))) return rc;
    u64 need = 0;
    for (u32 o = 0; o < T; ++o) need = std::max<u64>(need, (u64)fil[o] + dem[o]);
    if (need * 2 > c->gpe_cap && (rc = grow_gpe(c, pow2_at_least(need * 2 + 2)))) return rc;
  }
  if (c->gs_cap + c->gb_cap + c->gpe_cap == before) { g_err = "phase " + std::to_string(seg) + " posted for growth, but no table needs it"; return FQSX_E_DEVICE; }
  if (c->paired && (rc = pe_insert_and_clear(c))) return rc;
  return insert_and_clear(c);
}
// paired-end: the pair-table inserts of a phase and the clearing of the workers' local pair tables
int pe_clear_local(fqsx_dna *c) {
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  if ((rc = dzero(c, cfg.l_pe.key, c->cur_need_lpe * T * sizeof(u64)))) return rc;
  if ((rc = dzero(c, cfg.l_pe.val, c->cur_need_lpe * T * sizeof(u64)))) return rc;
  return dzero(c, cfg.l_pe.filled, T * sizeof(u32));
}
int pe_insert_and_clear(fqsx_dna *c) {
  LAUNCH(c, 2, k_pe_insert, c->T, 64, c->cfg);
  return pe_clear_local(c);
}

// One synchronisation segment on one GPU: encode launch, mailbox partition, growth decision, insert phase, clear
int block_segment(fqsx_dna *c, u32 seg) {
  const u32 T = c->T, n_reads = c->cur_n_reads;
  const u64 S = c->cur_S, need_lpe = c->cur_need_lpe;
  const bool decode = c->cur_decode;
  DevCfg &cfg = c->cfg;
  int rc;
  {
    if ((rc = launch_segment(c, decode, n_reads, (u32)S, seg))) return rc;
    // size the global tables for this phase's inserts (exact per-owner demand)
    const u32 part_grid = T * (cfg.mail[0].n_tiles + cfg.mail[1].n_tiles + cfg.mail[2].n_tiles);
    LAUNCH(c, 2, k_part_count, part_grid, 64, cfg);
    LAUNCH(c, 2, k_part_scan, 3 * T, 64, cfg);
    if (!decode) {
      // encoding: nothing is read back inside a block -- the growth checks are the device's (phase_skip)
      LAUNCH(c, 2, k_part_scatter, part_grid, 64, cfg, seg + 1, c->d_demand);   // (group offsets, demand words and growth check included)
      if (c->paired) {   // pair table: triples grouped by owner (count, offsets + growth check, scatter), then every owner's own group
#ifndef FQSX_EMU
        const u32 bgrid = (u32)std::min<u64>(1024, ((u64)T * cfg.pe_cap + 255) / 256);
#else
        const u32 bgrid = 1;   // (the emulated kernel body already walks every triple)
#endif
        LAUNCH(c, 2, k_pe_bucket_count, bgrid, 256, cfg);
        LAUNCH(c, 2, k_pe_bucket_offsets, 1, 64, cfg, seg + 1);
        LAUNCH(c, 2, k_pe_bucket_scatter, bgrid, 256, cfg);
        LAUNCH(c, 2, k_pe_insert_buckets, T, 64, cfg);
        if ((rc = pe_clear_local(c))) return rc;
      }
      return insert_and_clear(c);
    }
    LAUNCH(c, 2, k_part_dstoff, 3, 64, cfg, c->d_demand, 0u);
    // the demand travels to the host while the scatter (which does not depend on the growth decision) runs
    if ((rc = d2h_small_begin(c, c->d_demand, (4 * T + 1) * sizeof(u32)))) return rc;
    LAUNCH(c, 2, k_part_scatter, part_grid, 64, cfg, 0u, (u32 *)nullptr);
    if ((rc = d2h_small_end(c, c->h_demand.data(), (4 * T + 1) * sizeof(u32)))) return rc;
    if (c->h_demand[2 * T]) {
      g_err = "device error " + std::to_string(c->h_demand[2 * T]) + " in encode kernel";
      return FQSX_E_DEVICE;
    }
    if ((rc = grow_for_demand(c))) return rc;
    if (c->paired) {  // pair table: size for the exact per-owner demand, then insert
      LAUNCH(c, 2, k_pe_demand, T, 64, cfg, c->d_demand, 0u);
      if ((rc = d2h_sync(c, c->h_demand.data(), c->d_demand, T * sizeof(u32)))) return rc;
      if ((rc = d2h_sync(c, c->h_filled.data(), cfg.g_pe.filled, T * sizeof(u32)))) return rc;
      u64 need = 0;
      for (u32 o = 0; o < T; ++o) need = std::max<u64>(need, (u64)c->h_filled[o] + c->h_demand[o]);
      if (need * 2 > c->gpe_cap && (rc = grow_gpe(c, pow2_at_least(need * 2 + 2)))) return rc;
      if ((rc = pe_insert_and_clear(c))) return rc;
    }
    LAUNCH(c, 1, k_insert_phase, 3 * T, c->ins_threads, cfg, (u64)0, (u64)0);
    if ((rc = clear_local_tables(c))) return rc;
  }
  return FQSX_OK;
}

// *posted (if asked for): segment + 1 of the phase the device stopped the queue at, 0 if the block is complete
int block_finish(fqsx_dna *c, const u64 *h_off, const u8 **streams, u64 *lens, u8 *bases_out, u32 *posted = nullptr) {
  if (posted) *posted = 0;
  const u32 T = c->T, n_reads = c->cur_n_reads, generation = c->cur_gen;
  const bool decode = c->cur_decode;
  DevCfg &cfg = c->cfg;
  int rc;
  void *p = nullptr;
  if (decode) {
    u32 derr = 0;
    if ((rc = d2h_sync(c, &derr, cfg.err, sizeof(u32)))) return rc;
    if (derr) {
      g_err = "device error " + std::to_string(derr) + " while decoding block " + std::to_string(generation);
      return FQSX_E_DEVICE;
    }
    return d2h_sync(c, bases_out, cfg.dout, h_off[n_reads]);
  }
  LAUNCH(c, 2, k_finish_block, T, 64, cfg, c->d_lens, c->d_end);
  // ---- results: stream lengths, context-table occupancies (for the next block's sizing) and the error word at once
  {
    std::vector<u64> e(2 * T + 2);
    if ((rc = d2h_small_begin(c, c->d_end, e.size() * sizeof(u64)))) return rc;
    if ((rc = d2h_small_end(c, e.data(), e.size() * sizeof(u64)))) return rc;
    if (e[2 * T]) {
      g_err = "device error " + std::to_string(e[2 * T]) + " while encoding block " + std::to_string(generation);
      return FQSX_E_DEVICE;
    }
    if (e[2 * T + 1]) {
      if (posted) { *posted = (u32)e[2 * T + 1]; return FQSX_OK; }
      g_err = "growth posted by the device outside the deferred path";
      return FQSX_E_DEVICE;
    }
    for (u32 t = 0; t < T; ++t) { c->h_lens[t] = e[t]; c->h_filled[t] = (u32)e[T + t]; }
    c->filled_valid = true;
  }
  u64 total = 0;
  for (u32 t = 0; t < T; ++t) {
    if (c->h_lens[t] > c->out_cap) { g_err = "stream overflow"; return FQSX_E_DEVICE; }
    total += c->h_lens[t];
  }
  if (total > c->compact_cap) {
    dfree(c, c->d_compact);
    if ((rc = dalloc(c, &p, total + total / 2 + 4096, false))) return rc;
    c->d_compact = (u8 *)p;
    c->compact_cap = total + total / 2 + 4096;
  }
  LAUNCH(c, 2, k_compact_streams, T, 64, cfg, (const u64 *)c->d_lens, c->d_compact);
  c->h_out.resize(total ? total : 1);
  if (total && (rc = d2h_sync(c, c->h_out.data(), c->d_compact, total))) return rc;
  u64 pos = 0;
  for (u32 t = 0; t < T; ++t) {
    streams[t] = c->h_out.data() + pos;
    lens[t] = c->h_lens[t];
    pos += c->h_lens[t];
  }
  return FQSX_OK;
}

// decode: dec_streams/dec_lens (host) are the T input streams, bases_out (host) receives the block
int encode_block_impl(fqsx_dna *c, const u8 *d_bases, const u64 *d_off, const u64 *h_off, u32 n_reads, u32 generation,
                      const u8 **streams, u64 *lens, const u8 *const *dec_streams = nullptr, const u64 *dec_lens = nullptr,
                      u8 *bases_out = nullptr) {
  if (c->shard_world > 1) { g_err = "a sharded codec is driven through fqsx_shard_* (fqsqueezer_amd/sharded.py)"; return FQSX_E_ARG; }
  int rc = block_prepare(c, d_bases, d_off, h_off, n_reads, generation, dec_streams, dec_lens);
  for (u32 seg = 0; !rc;) {
    for (; !rc && seg <= c->cur_S; ++seg) rc = block_segment(c, seg);
    u32 posted = 0;
    if (!rc) rc = block_finish(c, h_off, streams, lens, bases_out, &posted);
    if (rc || !posted) break;
    rc = block_recover(c, posted - 1);   // (the queue stopped before that phase's inserts)
    seg = posted;
  }
  return rc;
}

int create_impl(fqsx_dna *c, const u8 *h) {
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.T = T;
  cfg.mode = h[5];
  cfg.prefix = h[10]; cfg.pmer = h[11]; cfg.smer = h[12]; cfg.bmer = h[13];
  cfg.gp = make_geom(cfg.pmer); cfg.gs = make_geom(cfg.smer); cfg.gb = make_geom(cfg.bmer);
  cfg.pmer_mod_shift = 2 * cfg.pmer - 12;          // dna.cpp:2381
  cfg.T_magic = ((1ull << 32) + T - 1) / T;
  cfg.T_pow2 = (T & (T - 1)) == 0 ? 1u : 0u;
  cfg.dbg = 0;
  if (const char *e = getenv("FQSX_PROTO_DEBUG")) cfg.dbg = (u32)strtoul(e, nullptr, 0);   // (tests: forced fall-back branches, FQSX_DBG_*)
  cfg.ps_nobytes_n = (2 * cfg.pmer + 7) / 8;       // dna.cpp:130
  int rc;
  void *p = nullptr;
  // p-mer vector: 4^pmer 2-bit counters (application.cpp:86, bit_vec.h:29-36)
  if ((rc = dalloc(c, &p, (1ull << (2 * cfg.pmer)) / 4, true))) return rc;
  cfg.siv = (u64 *)p;
  if ((rc = dalloc(c, &p, 2 * sizeof(u64), true))) return rc;
  cfg.siv_stats = (u64 *)p;
  if ((rc = dalloc(c, &p, ((1ull << (2 * cfg.pmer)) >> FQSX_SIV_BLK_LOG) * 4 * sizeof(u32), true))) return rc;   // count index: all fields zero
  cfg.siv_idx = (u32 *)p;
  // owner-sharded global tables (application.cpp:87-88; counters defs.h:26-27)
  c->tab_load_pct = 80; c->tab_after_pct = 62;
  if (const char *e = getenv("FQSX_TAB_LOAD_PCT")) c->tab_load_pct = (u32)std::min<u64>(85, std::max<u64>(20, strtoull(e, nullptr, 10)));
  c->tab_small_pct = 40;
  if (const char *e = getenv("FQSX_TAB_AFTER_PCT")) c->tab_small_pct = c->tab_after_pct = (u32)std::min<u64>(c->tab_load_pct - 5, std::max<u64>(10, strtoull(e, nullptr, 10)));
  cfg.tab_load_pct = c->tab_load_pct;
  c->ins_threads = (getenv("FQSX_INS_PREFETCH") && atoi(getenv("FQSX_INS_PREFETCH")) == 0) ? 64u : FQSX_INS_THREADS;
  if (const char *e = getenv("FQSX_WHATIF")) { u32 r = 0, u = 0; if (sscanf(e, "%u,%u", &r, &u) == 2) cfg.whatif = (r << 16) | (u & 0xffffu); }
  c->gs_cap = c->gb_cap = pow2_at_least(std::max<u64>(1024, (1ull << 22) / T));
  if (const char *e = getenv("FQSX_GTAB_INIT")) c->gs_cap = c->gb_cap = pow2_at_least(std::max<u64>(64, strtoull(e, nullptr, 10)));   // (tests: growth from tiny tables)
  if ((rc = ktab_alloc(c, cfg.g_s, T, c->gs_cap, cfg.smer, 12, true))) return rc;
  if ((rc = ktab_alloc(c, cfg.g_b, T, c->gb_cap, cfg.bmer, 6, true))) return rc;
  cfg.g_s.two = cfg.g_b.two = 1;   // two-choice buckets (a growth copies the descriptor, the kind with it); the local tables: one sequence
  // local tables: geometry chosen per block; counters allocated here
  c->ls_cap = c->lb_cap = 1024;
  if ((rc = ktab_alloc(c, cfg.l_s, T, c->ls_cap, cfg.smer, 12, false))) return rc;
  if ((rc = ktab_alloc(c, cfg.l_b, T, c->lb_cap, cfg.bmer, 6, false))) return rc;
  if ((rc = dalloc(c, &p, 2 * (u64)T * sizeof(u32), true))) return rc;  // occupancy counters of both, adjacent
  cfg.l_s.filled = (u32 *)p;
  cfg.l_b.filled = (u32 *)p + T;
  // contexts
  c->ctx_cap = 1u << 14;
  if ((rc = dalloc(c, &p, c->ctx_cap * T * sizeof(CtxSlot), true))) return rc;
  cfg.ctx = (CtxSlot *)p;
  cfg.ctx_cap_mask = c->ctx_cap - 1;
  if ((rc = dalloc(c, &p, T * sizeof(u32), true))) return rc;
  cfg.ctx_filled = (u32 *)p;
  // small models: all stats 1 (rc.h:69-72), 256-symbol models lazily
  {
    std::vector<u16> tpl(SM_OFF_BYTE, 0);
    auto fill = [&](u32 off, u32 entries, u32 stride, u32 n) {
      for (u32 e = 0; e < entries; ++e) {
        for (u32 i = 0; i < n; ++i) tpl[off + e * stride + i] = 1;
        tpl[off + e * stride + n] = (u16)n;
      }
    };
    fill(SM_OFF_FLAGS, 256, SM_FLAGS_N + 1, SM_FLAGS_N);
    fill(SM_OFF_NS, 32, SM_NS_N + 1, SM_NS_N);
    fill(SM_OFF_PSF, 65536, SM_PSF_N + 1, SM_PSF_N);
    fill(SM_OFF_PSNB, 65536, 6, cfg.ps_nobytes_n);
    fill(SM_OFF_NIB, 68, SM_NIB_N + 1, SM_NIB_N);
    tpl.resize(SM_TOTAL_U16, 0);   // 256-symbol models stay zero (lazily initialised); ctx_rc_pe_minimizer_id at the end
    fill(SM_OFF_MID, 1, SM_NIB_N + 1, SM_NIB_N);
    if ((rc = dalloc(c, &p, (u64)T * SM_TOTAL_U16 * sizeof(u16), false))) return rc;
    cfg.small = (u16 *)p;
    if ((rc = h2d(c, cfg.small, tpl.data(), tpl.size() * sizeof(u16)))) return rc;
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
    for (u32 t = 1; t < T; ++t)
      if ((rc = d2d(c, cfg.small + (u64)t * SM_TOTAL_U16, cfg.small, tpl.size() * sizeof(u16)))) return rc;
    if ((rc = dalloc(c, &p, (u64)T * SM_LAZY_ENTRIES, true))) return rc;
    cfg.byte_init = (u8 *)p;
  }
  // worker state: four std::mt19937 seeded 5481 (utils.h:296), everything else zero (dna.cpp:148-171)
  {
    std::vector<WState> ws(1);
    memset(ws.data(), 0, sizeof(WState));
    for (int g = 0; g < 4; ++g) {
      mt_seed(ws[0].mt[g], 5481);
      ws[0].mt_idx[g] = 624;
    }
    if ((rc = dalloc(c, &p, (u64)T * sizeof(WState), false))) return rc;
    cfg.ws = (WState *)p;
    for (u32 t = 0; t < T; ++t)
      if ((rc = h2d(c, cfg.ws + t, ws.data(), sizeof(WState)))) return rc;
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
  }
  for (u32 k = 0; k < 3; ++k) {
    if ((rc = dalloc(c, &p, (u64)T * sizeof(u32), true))) return rc;
    cfg.mail[k].n = (u32 *)p;
    if ((rc = dalloc(c, &p, (u64)T * sizeof(u32), true))) return rc;
    cfg.mail[k].dst_tot = (u32 *)p;
    if ((rc = dalloc(c, &p, ((u64)T + 1) * sizeof(u32), true))) return rc;
    cfg.mail[k].dst_off = (u32 *)p;
    if ((rc = mail_alloc(c, k, FQSX_TILE))) return rc;
  }
  c->paired = cfg.mode >= 2;
  c->gpe_cap = c->lpe_cap = 0;
  c->pe_cap = 0;
  if (c->paired) {  // ht_pe_mers (application.cpp:89) and ht_pe_mers_local (dna.cpp:105-107)
    c->gpe_cap = pow2_at_least(std::max<u64>(1024, (1ull << 20) / T));
    if (const char *e = getenv("FQSX_PTAB_INIT")) c->gpe_cap = pow2_at_least(std::max<u64>(64, strtoull(e, nullptr, 10)));   // (tests: growth from a tiny pair table)
    if ((rc = ptab_alloc(c, cfg.g_pe, T, c->gpe_cap, true))) return rc;
    c->lpe_cap = 1024;
    if ((rc = ptab_alloc(c, cfg.l_pe, T, c->lpe_cap, true))) return rc;
    if ((rc = dalloc(c, &p, (u64)T * sizeof(u32), true))) return rc;
    cfg.pe_n = (u32 *)p;
    if ((rc = dalloc(c, &p, (3 * (u64)T + 1) * sizeof(u32), true))) return rc;   // per-owner counts (kept zero between phases), offsets, cursors
    cfg.pe_bkt_n = (u32 *)p;
    cfg.pe_bkt_cur = cfg.pe_bkt_n + 2 * T + 1;
  }
  {  // group order of the partitioned mailboxes: by owner (one GPU)
    u8 ident[256];
    for (u32 i = 0; i < 256; ++i) ident[i] = (u8)i;
    if ((rc = dalloc(c, &p, 256, false))) return rc;
    c->d_vmap = (u8 *)p;
    cfg.vmap = c->d_vmap;
    if ((rc = h2d(c, c->d_vmap, ident, 256))) return rc;
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
  }
  cfg.shard_rank = 0; cfg.shard_world = 1; cfg.shard_cnt = nullptr;
  if ((rc = dalloc(c, &p, sizeof(u32) * 4, true))) return rc;
  cfg.err = (u32 *)p;
#ifdef FQSX_TIMING
  if ((rc = dalloc(c, &p, (u64)FQSX_TRACE_LAUNCHES * T * FQSX_TRACE_W * sizeof(u64), true))) return rc;
  cfg.trace = (u64 *)p;
#endif
  if ((rc = dalloc(c, &p, (5 * (u64)T + 2) * sizeof(u32), true))) return rc;   // [0 .. 4T]: k-mer tables (k_part_dstoff); [4T+1 ..): pair table
  c->d_demand = (u32 *)p;
  if ((rc = dalloc(c, &p, ((u64)T + 64) * sizeof(u64), true))) return rc;
  c->d_lens = (u64 *)p;
  if ((rc = dalloc(c, &p, (2 * (u64)T + 2) * sizeof(u64), true))) return rc;
  c->d_end = (u64 *)p;
#ifndef FQSX_EMU
  HIPCHK(hipHostMalloc((void **)&c->h_pin, FQSX_PIN_BYTES));
#else
  c->h_pin = (u8 *)malloc(FQSX_PIN_BYTES);
#endif
  c->filled_valid = false;
  c->h_demand.assign(4 * T + 1, 0);
  c->h_filled.assign(T, 0);
  c->h_lens.assign(T + 64, 0);
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------
extern "C" {

int fqsx_dna_create(const uint8_t *h, int device, fqsx_dna **out) { return fqsx_dna_create_on_partition(h, device, 0, 1, out); }

int fqsx_dna_create_on_partition(const uint8_t *h, int device, uint32_t part, uint32_t n_parts, fqsx_dna **out) {
  if (n_parts == 0 || part >= n_parts) { g_err = "bad compute-unit partition"; return FQSX_E_ARG; }
  if (!h || !out || h[0] != 'K' || h[1] != 'C' || h[2] != 'S' || h[3] != 'D') {  // params.h:102-129
    g_err = "malformed .fqs parameter header";
    return FQSX_E_ARG;
  }
  if (h[4] == 0) { g_err = "no_threads must be >= 1"; return FQSX_E_ARG; }
  if (h[5] > 3) { g_err = "unknown dna_mode"; return FQSX_E_ARG; }
  if (h[11] < 12 || h[11] > 18 || h[12] <= h[11] || h[13] <= h[12] + 1 || h[13] > 27 || h[10] >= h[11]) {
    g_err = "unsupported k-mer lengths";
    return FQSX_E_ARG;
  }
#ifndef FQSX_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_err = "no HIP device available (libfqsx has no CPU path)";
    return FQSX_E_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_err = "bad device ordinal"; return FQSX_E_ARG; }
  HIPCHK(hipSetDevice(device));
#endif
  fqsx_dna *c = new fqsx_dna();
  c->T = h[4];
  c->device = device;
  c->profiling = false;
  c->k_ms[0] = c->k_ms[1] = c->k_ms[2] = 0;
  c->k_n[0] = c->k_n[1] = c->k_n[2] = 0;
  c->out_cap = 0;
  c->mail_cap[0] = c->mail_cap[1] = c->mail_cap[2] = 0;
  c->dev_bases_cap = c->dev_off_cap = 0;
  c->d_bases = nullptr;
  c->d_off = nullptr;
  c->d_compact = nullptr;
  c->compact_cap = 0;
  c->din_cap = c->dout_cap = 0;
  c->shard_rank = 0; c->shard_world = 1;
  c->comm_set = false; c->shard_apply_own = false;
  memset(&c->comm, 0, sizeof(c->comm));
  for (int k = 0; k < 3; ++k) { c->d_xrecv[k] = nullptr; c->xrecv_cap[k] = 0; }
  c->d_items = c->d_gathered = c->d_small = nullptr; c->items_cap = c->gathered_cap = 0;
  c->sh_phases = c->sh_collectives = c->sh_a2a_words = c->sh_gather_words = 0;
  c->part = false; c->part_fallback = false; c->vm_gran = 0; c->vm_own_bytes = 0;
  c->dev_bytes = c->dev_bytes_peak = 0; c->n_growths = 0;
  c->h_pin = nullptr; c->d_end = nullptr; c->filled_valid = false;
  c->d_vmap = nullptr; c->d_xbuf = nullptr; c->xbuf_cap = 0; c->d_cglob = nullptr;
  c->cur_n_reads = c->cur_S = c->cur_gen = 0;
  c->cur_need_lb = c->cur_need_ls = c->cur_need_lpe = 0;
  c->cur_decode = false;
#ifndef FQSX_EMU
  hipError_t se;
  if (n_parts > 1) {   // a stream whose kernels only run on this partition's compute units
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_err = "hipGetDeviceProperties failed"; delete c; return FQSX_E_HIP; }
    const u32 ncu = (u32)prop.multiProcessorCount, lo = (u32)((u64)part * ncu / n_parts), hi = (u32)(((u64)part + 1) * ncu / n_parts);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (u32 i = lo; i < hi; ++i) mask[i / 32] |= 1u << (i % 32);
    se = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data());
  } else
    se = hipStreamCreate(&c->stream);
  if (se != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    g_err = "hipStreamCreate / hipEventCreate failed";
    delete c;
    return FQSX_E_HIP;
  }
#endif
  int rc = create_impl(c, h);
  if (!rc && getenv("FQSX_CHUNKED_TABLES") && atoi(getenv("FQSX_CHUNKED_TABLES"))) rc = fqsx_dna_use_chunked_tables(c);
  if (rc) {
    fqsx_dna_destroy(c);
    return rc;
  }
  *out = c;
  return FQSX_OK;
}

void fqsx_dna_destroy(fqsx_dna *c) {
  if (!c) return;
#ifndef FQSX_EMU
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
#endif
  vtab_free(c, c->vm_s);
  vtab_free(c, c->vm_b);
  vtab_free(c, c->vm_pk);
  vtab_free(c, c->vm_pv);
  vtab_free(c, c->vm_siv);
  fqsx_vm::mesh_close(c->mesh);
  std::vector<void *> a = c->allocs;
  for (void *p : a) dfree(c, p);
#ifndef FQSX_EMU
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
#else
  free(c->h_pin);
#endif
  delete c;
}

int fqsx_dna_encode_block_dev(fqsx_dna *c, const uint8_t *d_bases, const uint64_t *d_off, const uint64_t *h_off,
                              uint32_t n_reads, uint32_t generation, const uint8_t **streams, uint64_t *lens) {
  if (!c || !d_bases || !d_off || !h_off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  return encode_block_impl(c, d_bases, d_off, h_off, n_reads, generation, streams, lens);
}

int fqsx_dna_encode_block(fqsx_dna *c, const uint8_t *bases, const uint64_t *off, uint32_t n_reads, uint32_t generation,
                          const uint8_t **streams, uint64_t *lens) {
  if (!c || !bases || !off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  int rc;
  void *p = nullptr;
  u64 nb = off[n_reads] + 64, no = ((u64)n_reads + 1) * sizeof(u64);
  if (nb > c->dev_bases_cap) {
    dfree(c, c->d_bases);
    if ((rc = dalloc(c, &p, nb + nb / 4, false))) return rc;
    c->d_bases = (u8 *)p;
    c->dev_bases_cap = nb + nb / 4;
  }
  if (no > c->dev_off_cap) {
    dfree(c, c->d_off);
    if ((rc = dalloc(c, &p, no + no / 4, false))) return rc;
    c->d_off = (u64 *)p;
    c->dev_off_cap = no + no / 4;
  }
  if ((rc = h2d(c, c->d_bases, bases, off[n_reads]))) return rc;
  if ((rc = h2d(c, c->d_off, off, no))) return rc;
  return encode_block_impl(c, c->d_bases, c->d_off, off, n_reads, generation, streams, lens);
}

int fqsx_dna_decode_block(fqsx_dna *c, const uint8_t *const *streams, const uint64_t *lens, const uint64_t *off, uint32_t n_reads,
                          uint32_t generation, uint8_t *bases_out) {
  if (!c || !streams || !lens || !off || !bases_out) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  int rc;
  void *p = nullptr;
  u64 no = ((u64)n_reads + 1) * sizeof(u64);
  if (no > c->dev_off_cap) {
    dfree(c, c->d_off);
    if ((rc = dalloc(c, &p, no + no / 4, false))) return rc;
    c->d_off = (u64 *)p;
    c->dev_off_cap = no + no / 4;
  }
  if ((rc = h2d(c, c->d_off, off, no))) return rc;
  return encode_block_impl(c, nullptr, c->d_off, off, n_reads, generation, nullptr, nullptr, streams, lens, bases_out);
}

// ---- sharded mode: one synchronisation phase in steps, with the collectives in between left to the caller ----------
// (fqsqueezer_amd/sharded.py drives these with torch.distributed: RCCL on GPUs, gloo for the emulation build).
// Pointers marked [codec] are in the codec's memory space: device memory for the HIP build.
int fqsx_shard_config(fqsx_dna *c, uint32_t rank, uint32_t world) {
  if (!c || world == 0 || rank >= world || world > c->T) { g_err = "bad rank / world size"; return FQSX_E_ARG; }
  if (c->part && world > 1) { g_err = "a codec with chunked tables cannot be sharded afterwards (fqsx_shard_partition_tables after the attach)"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  c->shard_rank = rank; c->shard_world = world;
  c->cfg.shard_rank = rank; c->cfg.shard_world = world;
  // groups of the partitioned mailboxes in rank-major order: what goes to one rank is contiguous
  u8 vm[256];
  for (u32 i = 0; i < 256; ++i) vm[i] = (u8)i;
  u32 v = 0;
  for (u32 q = 0; q < world; ++q)
    for (u32 o = q; o < T; o += world) vm[o] = (u8)v++;
  int rc;
  void *p = nullptr;
  if ((rc = h2d(c, c->d_vmap, vm, 256))) return rc;
  if (!c->cfg.shard_cnt) {
    if ((rc = dalloc(c, &p, (3ull * T * T + T + 1) * sizeof(u32), true))) return rc;   // (+ T: paired-end triples per source, + 1: status word)
    c->cfg.shard_cnt = (u32 *)p;
    if ((rc = dalloc(c, &p, (3ull * T * T + T + 1) * sizeof(u32), true))) return rc;
    c->d_cglob = (u32 *)p;
    if ((rc = dalloc(c, &p, 3ull * (T + 1) * T * sizeof(u32), true))) return rc;   // (room for any world <= T)
    c->d_colsum = (u32 *)p;
    if ((rc = dalloc(c, &p, 8 * sizeof(u64), true))) return rc;
    c->d_small = (u64 *)p;
    c->h_cglob.assign(3ull * T * T + T + 1, 0);
  }
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}

int fqsx_shard_begin_block(fqsx_dna *c, const uint8_t *bases /*[codec]*/, const uint64_t *off /*[codec]*/, const uint64_t *h_off,
                           uint32_t n_reads, uint32_t generation, uint32_t *n_segments) {
  if (!c || !bases || !off || !h_off || !n_segments) { g_err = "null argument"; return FQSX_E_ARG; }
  if (c->part) { g_err = "partitioned tables are driven through fqsx_shard_encode_block"; return FQSX_E_ARG; }
  // the step-wise driver exchanges the three k-mer mailboxes only: the pair-table triples of a paired-end file travel with
  // the native loop's all-gather (shard_phase_native), nowhere else -- a world of several ranks would leave the pair table empty
  if (c->paired && c->shard_world > 1) { g_err = "paired-end files are sharded through fqsx_shard_attach + fqsx_shard_encode_block (the step-wise fqsx_shard_* driver does not exchange the pair-table triples)"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  int rc = block_prepare(c, bases, off, h_off, n_reads, generation);
  *n_segments = c->cur_S + 1;
  return rc;
}

// encode launch of segment `seg` for this rank's workers, then the per-(source, owner) counts of their mailboxes
int fqsx_shard_encode(fqsx_dna *c, uint32_t seg, uint32_t *counts /*[codec] [3][T][T]*/) {
  if (!c || !counts) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  if ((rc = launch_segment(c, false, c->cur_n_reads, c->cur_S, seg))) return rc;
  const u32 part_grid = T * (cfg.mail[0].n_tiles + cfg.mail[1].n_tiles + cfg.mail[2].n_tiles);
  LAUNCH(c, 2, k_part_count, part_grid, 64, cfg);
  LAUNCH(c, 2, k_shard_counts, 3 * T + 1, 64, cfg, cfg.shard_cnt, 0u);
  if ((rc = d2d(c, counts, cfg.shard_cnt, 3ull * T * T * sizeof(u32)))) return rc;
  u32 err = 0;
  if ((rc = d2h_sync(c, &err, cfg.err, sizeof(u32)))) return rc;
  if (err) { g_err = "device error " + std::to_string(err) + " in encode kernel"; return FQSX_E_DEVICE; }
  return FQSX_OK;
}

// the entries of this rank's workers in send order (destination rank, owner, source, push) -> send[kind]
int fqsx_shard_pack(fqsx_dna *c, const uint32_t *counts_sum /*[codec] summed over ranks*/, uint64_t *const send[3] /*[codec]*/) {
  if (!c || !counts_sum || !send) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  if ((rc = d2d(c, c->d_cglob, counts_sum, 3ull * T * T * sizeof(u32)))) return rc;
  const u32 part_grid = T * (cfg.mail[0].n_tiles + cfg.mail[1].n_tiles + cfg.mail[2].n_tiles);
  LAUNCH(c, 2, k_part_scan, 3 * T, 64, cfg);
  LAUNCH(c, 2, k_part_dstoff, 3, 64, cfg, c->d_demand, 0u);
  LAUNCH(c, 2, k_part_scatter, part_grid, 64, cfg, 0u, (u32 *)nullptr);
  std::vector<u32> tot(3 * (T + 1));
  for (u32 k = 0; k < 3; ++k)
    if ((rc = d2h_sync(c, tot.data() + k * (T + 1), cfg.mail[k].dst_off, (T + 1) * sizeof(u32)))) return rc;
  for (u32 k = 0; k < 3; ++k) {
    const u64 n = tot[k * (T + 1) + T];
    if (n && (rc = d2d(c, send[k], cfg.mail[k].sorted, n * sizeof(u64)))) return rc;
  }
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}

// received entries (chunks of ranks 0..G-1 back to back) -> this rank's owners' groups; need[0/1] = this rank's table
// demand for the s- / b-mer tables (occupied + incoming slots of its fullest sub-table)
int fqsx_shard_merge(fqsx_dna *c, const uint64_t *const recv[3] /*[codec]*/, uint64_t need[2]) {
  if (!c || !recv || !need) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  for (u32 k = 0; k < 3; ++k) LAUNCH(c, 2, k_shard_merge, T, 64, cfg, k, recv[k], (const u32 *)c->d_cglob);
  need[0] = need[1] = 0;
  std::vector<u32> tot(T), fil(T);
  for (int which = 0; which < 2; ++which) {
    const KTab &t = which ? cfg.g_b : cfg.g_s;
    if ((rc = d2h_sync(c, tot.data(), cfg.mail[which ? MAIL_B : MAIL_S].dst_tot, T * sizeof(u32)))) return rc;
    if ((rc = d2h_sync(c, fil.data(), t.filled, T * sizeof(u32)))) return rc;
    for (u32 o = 0; o < T; ++o) need[which] = std::max<u64>(need[which], (u64)tot[o] + fil[o]);
  }
  return FQSX_OK;
}

// the insert phase of this rank's owners (after every rank has agreed on the tables' demand), then one item per
// applied entry for the other ranks' replicas; siv_delta = this rank's contribution to (no_updates, no_filled)
int fqsx_shard_insert(fqsx_dna *c, uint64_t need_s, uint64_t need_b, uint64_t *const items[3] /*[codec]*/, uint64_t siv_delta[2]) {
  if (!c || !items || !siv_delta) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  DevCfg &cfg = c->cfg;
  int rc;
  if (tab_over(c, need_s, c->gs_cap) && (rc = grow_global(c, cfg.g_s, c->gs_cap, tab_new_cap(c, need_s)))) return rc;
  if (tab_over(c, need_b, c->gb_cap) && (rc = grow_global(c, cfg.g_b, c->gb_cap, tab_new_cap(c, need_b)))) return rc;
  if ((rc = d2h_sync(c, c->siv_before, cfg.siv_stats, 2 * sizeof(u64)))) return rc;
  LAUNCH(c, 1, k_insert_phase, 3 * T, c->ins_threads, cfg, (u64)0, (u64)0);
  u64 after[2];
  if ((rc = d2h_sync(c, after, cfg.siv_stats, 2 * sizeof(u64)))) return rc;
  siv_delta[0] = after[0] - c->siv_before[0];
  siv_delta[1] = after[1] - c->siv_before[1];
  for (u32 k = 0; k < 3; ++k) LAUNCH(c, 2, k_shard_collect, REHASH_GRID, 256, cfg, k, items[k]);
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}

// another rank's items of one kind into this rank's replica
int fqsx_shard_apply(fqsx_dna *c, uint32_t kind, const uint64_t *items /*[codec]*/, uint64_t n) {
  if (!c || kind > 2 || (!items && n)) { g_err = "bad argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  if (n) LAUNCH(c, 2, k_shard_apply, REHASH_GRID, 256, c->cfg, kind, items, (u32)n);
  return FQSX_OK;
}

// end of the phase: the p-mer vector's statistics become the sum over the ranks, the local tables are cleared
int fqsx_shard_end_phase(fqsx_dna *c, const uint64_t siv_delta_sum[2]) {
  if (!c || !siv_delta_sum) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  int rc;
  u64 now[2] = {c->siv_before[0] + siv_delta_sum[0], c->siv_before[1] + siv_delta_sum[1]};
  if ((rc = h2d(c, c->cfg.siv_stats, now, 2 * sizeof(u64)))) return rc;
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return clear_local_tables(c);
}

// streams of the block (only those of this rank's workers are meaningful)
int fqsx_shard_finish_block(fqsx_dna *c, const uint64_t *h_off, const uint8_t **streams, uint64_t *lens) {
  if (!c || !h_off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  return block_finish(c, h_off, streams, lens, nullptr);
}

// ---- the phase loop inside the library ------------------------------------------------------------------------------
namespace {
int xbuf_fit(fqsx_dna *c, u64 *&buf, u64 &cap, u64 words) {
  if (words <= cap) return FQSX_OK;
  void *p = nullptr;
  if (buf) dfree(c, buf);
  const u64 ncap = words + words / 4 + 1024;
  int rc = dalloc(c, &p, ncap * sizeof(u64), false);
  if (rc) { buf = nullptr; cap = 0; return rc; }
  buf = (u64 *)p;
  cap = ncap;
  return FQSX_OK;
}
// A collective that fails is not something the ranks can agree on any more: the transport is told to give up (RCCL:
// ncclCommAbort, so that this rank's part of a pending collective does not keep the others' kernels spinning) and the call fails.
#define COMMCHK(x, what) do { if ((x) != 0) { if (g_err.empty() || g_err.find(what) == std::string::npos) g_err = std::string(what) + " failed"; if (c->comm.abort) c->comm.abort(c->comm.ctx); return FQSX_E_HIP; } } while (0)

// All ranks leave a phase together or not at all: RCCL has no timeout, so a rank that returned between two collectives would
// leave the others waiting in the next one for ever.  What a rank can fail on by itself -- the device error word of its encode
// launch, a device allocation, a table growth -- is therefore put to a vote before anybody moves on:
//   * the all-reduced count matrix carries one status word (k_shard_counts): encode errors are known to all after collective 1;
//   * buffers and tables are sized from numbers every rank holds alike (the summed matrix, the replicated / exchanged
//     occupancies), so "does anybody allocate or grow in this phase" has the same answer everywhere; in such a phase -- a few
//     per file -- the ranks allocate and grow first and then all-reduce one word before the mailboxes travel.
// fail: this rank's own verdict (0 = fine).  Returns 0 if every rank is fine, this rank's error code or FQSX_E_PEER otherwise.
int shard_vote(fqsx_dna *c, int fail, const char *what) {
  const u64 NC = 3ull * c->T * c->T + c->T + 1;
  u32 *word = c->d_cglob + NC - 1;   // (the status word of the count matrix: read by the host already)
  u32 v = fail ? 1u : 0u;
  const std::string mine = g_err;
  int rc;
  if ((rc = h2d(c, word, &v, sizeof(v)))) return rc;
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));   // (v is a stack variable)
#endif
  COMMCHK(c->comm.allreduce_sum_u32(c->comm.ctx, word, 1), "all-reduce of the phase's status vote");
  if ((rc = d2h_sync(c, &v, word, sizeof(v)))) return rc;
  c->sh_collectives += 1;
  if (fail) { g_err = mine; return fail; }
  if (v) { g_err = std::string("another rank of the world failed in ") + what + " (this rank was fine): the phase is given up on every rank"; return FQSX_E_PEER; }
  return FQSX_OK;
}

int shard_phase_native(fqsx_dna *c, u32 seg) {
  const u32 T = c->T, G = c->shard_world, me = c->shard_rank;
  DevCfg &cfg = c->cfg;
  const u64 NC = 3ull * T * T + T + 1;   // counts, paired-end triples per source, status word
  int rc;
  // ---- encode own workers, count what they pushed for whom
  if ((rc = launch_segment(c, false, c->cur_n_reads, c->cur_S, seg))) return rc;
  const u32 part_grid = T * (cfg.mail[0].n_tiles + cfg.mail[1].n_tiles + cfg.mail[2].n_tiles);
  LAUNCH(c, 2, k_part_count, part_grid, 64, cfg);
  LAUNCH(c, 2, k_shard_counts, 3 * T + 1, 64, cfg, c->d_cglob, 0u);   // (straight into the buffer the all-reduce runs on)
  COMMCHK(c->comm.allreduce_sum_u32(c->comm.ctx, c->d_cglob, NC), "all-reduce of the mailbox counts");   // collective 1
  LAUNCH(c, 2, k_shard_colsum, (3 * T + 63) / 64, 64, cfg, (const u32 *)c->d_cglob, c->d_colsum);
  LAUNCH(c, 2, k_shard_need, 1, 64, cfg, (const u32 *)c->d_colsum, c->d_small + 2, c->d_small);
  // ---- the phase's one host round trip: every transfer size, the table demand, the error word
  u64 small[3];
#ifndef FQSX_EMU
  HIPCHK(hipMemcpyAsync(c->h_cglob.data(), c->d_cglob, NC * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
#else
  memcpy(c->h_cglob.data(), c->d_cglob, NC * sizeof(u32));
#endif
  if ((rc = d2h_sync(c, small, c->d_small + 2, sizeof(small)))) return rc;
  const u32 *C = c->h_cglob.data();
  if (C[NC - 1]) {   // a device error somewhere in the world: every rank sees the same word and leaves here
    if (small[2]) { g_err = "device error " + std::to_string(small[2]) + " in encode kernel"; return FQSX_E_DEVICE; }
    g_err = "device error in the encode kernel of another rank of the world (this rank was fine): the phase is given up on every rank";
    return FQSX_E_PEER;
  }
  std::vector<u64> vol(3ull * G * G, 0);   // [kind][from rank][to rank]
  for (u32 k = 0; k < 3; ++k)
    for (u32 s = 0; s < T; ++s)
      for (u32 o = 0; o < T; ++o) vol[((u64)k * G + s % G) * G + o % G] += C[((u64)k * T + s) * T + o];
  // ---- own entries in send order (destination rank, owner, source, push) = the partitioned mailbox with rank-major groups
  LAUNCH(c, 2, k_part_scan, 3 * T, 64, cfg);
  LAUNCH(c, 2, k_part_dstoff, 3, 64, cfg, c->d_demand, 0u);
  LAUNCH(c, 2, k_part_scatter, part_grid, 64, cfg, 0u, (u32 *)nullptr);
  std::vector<u64> sc(3ull * G), rcnt(3ull * G);
  const u64 *send[3];
  u64 *recv[3];
  u64 recv_need[3] = {0, 0, 0};   // the largest rank's incoming entries: every rank sizes its buffer for that (same capacity everywhere)
  for (u32 k = 0; k < 3; ++k) {
    for (u32 q = 0; q < G; ++q) {
      sc[(u64)k * G + q] = vol[((u64)k * G + me) * G + q];
      rcnt[(u64)k * G + q] = vol[((u64)k * G + q) * G + me];
      if (q != me) c->sh_a2a_words += sc[(u64)k * G + q];
      u64 n_in = 0;
      for (u32 f = 0; f < G; ++f) n_in += vol[((u64)k * G + f) * G + q];
      recv_need[k] = std::max(recv_need[k], n_in + 1);
    }
  }
  // ---- the all-gather's layout: the applied items of every kind (padded to the largest rank's), the p-mer statistics, the triples
  // (partitioned tables: no s-/b-mer items -- the look-ups read the owner's memory -- but the owners' occupancy counters)
  u64 M[3] = {0, 0, 0}, PM = 0;
  std::vector<u64> n_items(3ull * G, 0), pe_tot(G, 0);
  for (u32 k = 0; k < 3; ++k)
    for (u32 q = 0; q < G; ++q) {
      if (c->part && k != MAIL_P) continue;
      if (G == 1 && !c->shard_apply_own) continue;   // (nobody to hand the items to)
      for (u32 f = 0; f < G; ++f) n_items[(u64)k * G + q] += vol[((u64)k * G + f) * G + q];
      M[k] = std::max(M[k], n_items[(u64)k * G + q]);
    }
  const u32 n_own_max = (T + G - 1) / G;
  const u32 n_fill = c->part ? (cfg.pe_part ? 3u : 2u) : 0u;   // occupancy arrays the owners report: s-mer, b-mer (, pair) table
  const u64 FW = (u64)n_fill * n_own_max;
  if (c->paired)
    for (u32 s = 0; s < T; ++s) { pe_tot[s % G] += C[3ull * T * T + s]; PM = std::max(PM, pe_tot[s % G]); }
  const u64 off_k[3] = {0, M[0], M[0] + M[1]}, off_siv = M[0] + M[1] + M[2], off_fill = off_siv + 2, off_pe = off_fill + FW, W = off_pe + 3 * PM;
  // ---- allocations and growths of this phase, if any (every rank answers alike: see shard_vote), then the vote
  const bool grow_s = tab_over(c, small[0], c->gs_cap), grow_b = tab_over(c, small[1], c->gb_cap);
  // (tests: FQSX_TEST_FAIL="rank,phase" -- that rank reports a failed allocation in that phase, which every rank then treats as
  // an allocating one: all of them must come back with an error, none may be left waiting in a collective)
  int inj_rank = -1, inj_phase = -1;
  if (const char *e = getenv("FQSX_TEST_FAIL")) (void)sscanf(e, "%d,%d", &inj_rank, &inj_phase);
  const bool injected = inj_phase >= 0 && (u64)inj_phase == c->sh_phases;
  if (injected || grow_s || grow_b || W > c->items_cap || W * G > c->gathered_cap || recv_need[0] > c->xrecv_cap[0] || recv_need[1] > c->xrecv_cap[1] ||
      recv_need[2] > c->xrecv_cap[2]) {
    int fail = 0;
    if (injected && (u32)inj_rank == me) { fail = FQSX_E_NOMEM; g_err = "injected allocation failure (FQSX_TEST_FAIL)"; }
    for (u32 k = 0; k < 3 && !fail; ++k) fail = xbuf_fit(c, c->d_xrecv[k], c->xrecv_cap[k], recv_need[k]);
    if (!fail) fail = xbuf_fit(c, c->d_items, c->items_cap, W);
    if (!fail) fail = xbuf_fit(c, c->d_gathered, c->gathered_cap, W * G);
    // growth: every rank sees the same demand (the replicas are exact / the occupancies are exchanged); nobody looks a k-mer
    // up between collective 1 and the end of the phase, so the tables can be rebuilt here as well as before the inserts
    if (!fail && grow_s) fail = grow_global(c, cfg.g_s, c->gs_cap, tab_new_cap(c, small[0]));
    if (!fail && grow_b) fail = grow_global(c, cfg.g_b, c->gb_cap, tab_new_cap(c, small[1]));
    if (fail && c->part) fqsx_vm::mesh_close(c->mesh);   // (ranks waiting for this rank's descriptors fail instead of timing out)
    if ((rc = shard_vote(c, fail, "the allocations / table growth of a phase"))) return rc;
  }
  for (u32 k = 0; k < 3; ++k) { send[k] = cfg.mail[k].sorted; recv[k] = c->d_xrecv[k]; }
  COMMCHK(c->comm.alltoallv_u64(c->comm.ctx, 3, send, sc.data(), recv, rcnt.data()), "all-to-all of the mailboxes");   // collective 2
  LAUNCH(c, 2, k_shard_merge3, 3 * T, 64, cfg, (const u64 *)c->d_xrecv[0], (const u64 *)c->d_xrecv[1], (const u64 *)c->d_xrecv[2], (const u32 *)c->d_cglob, (const u32 *)c->d_colsum);
  // ---- insert phase of own owners
  if (cfg.siv_part) {   // the owners log every field they change straight into the all-gather's p-mer items (zero = no item)
    cfg.p_log = c->d_items + off_k[MAIL_P];
    if (M[MAIL_P]) LAUNCH(c, 2, k_zero_words, (u32)std::min<u64>(REHASH_GRID, (M[MAIL_P] + 255) / 256), 256, cfg.p_log, M[MAIL_P], 0u);
    if ((rc = dzero(c, c->d_plog_n, sizeof(u32)))) return rc;
  }
  LAUNCH(c, 1, k_insert_phase, 3 * T, c->ins_threads, cfg, (u64)0, (u64)0);
  // ---- one all-gather
  for (u32 k = 0; k < 3; ++k)
    if (n_items[(u64)k * G + me] && !(k == MAIL_P && cfg.siv_part)) LAUNCH(c, 2, k_shard_collect, REHASH_GRID, 256, cfg, k, c->d_items + off_k[k]);
  if (FW) LAUNCH(c, 2, k_shard_pack2, 1, 64, cfg, (const u64 *)c->d_small, c->d_items + off_siv, c->d_items + off_fill, n_own_max, n_fill);
  else LAUNCH(c, 2, k_shard_siv_delta, 1, 64, cfg, (const u64 *)c->d_small, c->d_items + off_siv);
  if (c->paired && PM) LAUNCH(c, 2, k_shard_pe_pack, T, 64, cfg, c->d_items + off_pe);
  COMMCHK(c->comm.allgather_u64(c->comm.ctx, c->d_items, W, c->d_gathered), "all-gather of the applied items");   // collective 3
  c->sh_gather_words += W * (G - 1);
  for (u32 q = 0; q < G; ++q)
    if (q != me || c->shard_apply_own)
      for (u32 k = 0; k < 3; ++k)
        if (n_items[(u64)k * G + q] && !(k == MAIL_P && cfg.siv_part && q == me)) LAUNCH(c, 2, k_shard_apply, REHASH_GRID, 256, cfg, k, (const u64 *)(c->d_gathered + (u64)q * W + off_k[k]), (u32)n_items[(u64)k * G + q]);
  if (FW) LAUNCH(c, 2, k_shard_unpack2, 1, 64, cfg, (const u64 *)c->d_small, (const u64 *)c->d_gathered, W, off_siv, off_fill, n_own_max, n_fill);
  else LAUNCH(c, 2, k_shard_siv_sum, 1, 64, cfg, (const u64 *)c->d_small, (const u64 *)c->d_gathered, W, off_siv);
  // ---- paired-end: every rank holds every source's triples (the all-gather above) and applies them to its replica of the pair
  // table -- or, with partitioned tables, those of its own owners to its own sub-tables (k_pe_insert; the owners' occupancy
  // counters came with the all-gather, so the growth rule below sees the same numbers on every rank)
  if (c->paired) {
    if (G > 1 && PM) LAUNCH(c, 2, k_shard_pe_unpack, T, 64, cfg, (const u64 *)c->d_gathered, W, off_pe, (const u32 *)(c->d_cglob + 3ull * T * T));
    LAUNCH(c, 2, k_pe_demand, T, 64, cfg, c->d_demand, 0u);
    if ((rc = d2h_sync(c, c->h_demand.data(), c->d_demand, T * sizeof(u32)))) return rc;
    if ((rc = d2h_sync(c, c->h_filled.data(), cfg.g_pe.filled, T * sizeof(u32)))) return rc;
    u64 need = 0;
    for (u32 o = 0; o < T; ++o) need = std::max<u64>(need, (u64)c->h_filled[o] + c->h_demand[o]);
    if (need * 2 > c->gpe_cap) {   // (replicas / exchanged occupancies are exact: every rank grows in the same phase, and votes on it)
      const int fail = grow_gpe(c, pow2_at_least(need * 2 + 2));
      if (fail && c->part) fqsx_vm::mesh_close(c->mesh);
      if ((rc = shard_vote(c, fail, "the growth of the pair table"))) return rc;
    }
    LAUNCH(c, 2, k_pe_insert, T, 64, cfg);
    if (cfg.pe_part) {
      // The triples arrive with the phase's last collective, so these inserts are the one table update no collective follows --
      // and the next segment's look-ups read the other ranks' sub-tables.  One more (one-word) all-reduce on the stream: no rank
      // starts its next encode launch before every rank's inserts are complete.  (Sending the triples to their owners' ranks
      // with the mailboxes instead -- inserts before the all-gather, as for the k-mer tables -- would save it: not built.)
      COMMCHK(c->comm.allreduce_sum_u32(c->comm.ctx, c->d_cglob, 1), "barrier behind the pair-table inserts");
      c->sh_collectives += 1;
    }
    if ((rc = dzero(c, cfg.l_pe.key, c->cur_need_lpe * T * sizeof(u64)))) return rc;
    if ((rc = dzero(c, cfg.l_pe.val, c->cur_need_lpe * T * sizeof(u64)))) return rc;
    if ((rc = dzero(c, cfg.l_pe.filled, T * sizeof(u32)))) return rc;
    c->filled_valid = false;   // (h_filled was borrowed for the pair table's occupancies)
  }
  c->sh_phases += 1;
  c->sh_collectives += 3;
  return clear_local_tables(c);
}
}  // namespace

int fqsx_shard_attach(fqsx_dna *c, uint32_t rank, uint32_t world, const fqsx_comm *comm) {
  if (!c || !comm || !comm->allreduce_sum_u32 || !comm->alltoallv_u64 || !comm->allgather_u64) { g_err = "incomplete transport"; return FQSX_E_ARG; }
  int rc = fqsx_shard_config(c, rank, world);
  if (rc) return rc;
  c->comm = *comm;
  c->comm_set = true;
  if (const char *e = getenv("FQSX_SHARD_APPLY_OWN")) c->shard_apply_own = atoi(e) != 0;   // (tests: the replica update on this rank's own items must change nothing)
  return FQSX_OK;
}

namespace {
// the (still empty) s- and b-mer tables again, as chunked tables
int tables_to_chunks(fqsx_dna *c) {
  int rc;
  VMCHK(fqsx_vm::granularity(c->device, &c->vm_gran, e_));
  if (c->vm_gran < sizeof(u64) || (c->vm_gran & (c->vm_gran - 1))) { g_err = "unexpected allocation granularity"; return FQSX_E_HIP; }
  c->part = true;
  // sub-tables on other GPUs: the writers end with a system-scope release, the readers start behind a system-scope acquire
  // (FQSX_SYS_SCOPE=1: the same fences in a world of one rank -- what they cost, measured on one GPU)
  c->cfg.sys_scope = (c->shard_world > 1 || (getenv("FQSX_SYS_SCOPE") && atoi(getenv("FQSX_SYS_SCOPE")))) ? 1u : 0u;
  dfree(c, c->cfg.g_s.slots);
  dfree(c, c->cfg.g_b.slots);
  c->cfg.g_s.slots = c->cfg.g_b.slots = nullptr;
  if ((rc = vtab_alloc(c, c->cfg.g_s, c->vm_s, c->gs_cap, c->cfg.smer, 12))) return rc;
  if ((rc = vtab_alloc(c, c->cfg.g_b, c->vm_b, c->gb_cap, c->cfg.bmer, 6))) return rc;
  if (c->paired) {   // the pair table with them: a rank holds (and writes) its own owners' sub-tables, key and value array alike
    dfree(c, c->cfg.g_pe.key);
    dfree(c, c->cfg.g_pe.val);
    c->cfg.g_pe.key = c->cfg.g_pe.val = nullptr;
    if ((rc = ptab_reserve(c, c->cfg.g_pe, c->vm_pk, c->vm_pv, c->gpe_cap))) return rc;
    for (u32 o = c->shard_rank; o < c->T; o += c->shard_world)
      if ((rc = vtab_create_own(c, c->vm_pk, o)) || (rc = vtab_create_own(c, c->vm_pv, o))) return rc;
    if ((rc = vtab_exchange(c, c->vm_pk)) || (rc = vtab_exchange(c, c->vm_pv))) return rc;
    c->cfg.pe_part = c->shard_world > 1 ? 1u : 0u;
  }
  // the p-mer vector (application.h:51 siv_pmer; owner of an index: its top 12 bits mod T, dna.cpp:658): 4096 owner ranges = 4096
  // chunks, each on its owner's rank -- where a range is as large as a chunk may be small (2 MiB on the GPU: the 16 GiB vector of the
  // default geometry has 4 MiB ranges); smaller vectors stay replicas.  The count index (1/256 of the vector) stays a replica.
  if (c->shard_world > 1 && !(getenv("FQSX_SIV_REPLICA") && atoi(getenv("FQSX_SIV_REPLICA")))) {
    const u64 siv_bytes = (1ull << (2 * c->cfg.pmer)) / 4, range_bytes = siv_bytes / 4096;
#ifndef FQSX_EMU
    const u64 min_chunk = std::max<u64>(c->vm_gran, 2ull << 20);
#else
    const u64 min_chunk = c->vm_gran;
#endif
    if (range_bytes >= min_chunk && range_bytes % c->vm_gran == 0 && c->cfg.pmer_mod_shift == 2 * c->cfg.pmer - 12) {
      rlimit rl;   // (own chunks keep a descriptor each in the emulation build; exports are transient)
      if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < 16384 && rl.rlim_cur < rl.rlim_max) {
        rl.rlim_cur = std::min<rlim_t>(rl.rlim_max, 16384);
        (void)setrlimit(RLIMIT_NOFILE, &rl);
      }
      dfree(c, c->cfg.siv);
      c->cfg.siv = nullptr;
      u64 stride = 0;
      c->vm_siv = fqsx_dna::VmTab();
      if ((rc = vtab_reserve_raw(c, c->vm_siv, range_bytes / sizeof(u64), &stride, 4096))) return rc;
      if (stride != range_bytes / sizeof(u64)) { g_err = "p-mer vector ranges do not tile the address range"; return FQSX_E_HIP; }
      c->vm_siv.own_mod = c->T;
      for (u32 r = 0; r < 4096; ++r)
        if (vt_rank(c, c->vm_siv, r) == c->shard_rank && (rc = vtab_create_own(c, c->vm_siv, r))) return rc;
      if ((rc = vtab_exchange(c, c->vm_siv))) return rc;
      c->cfg.siv = (u64 *)c->vm_siv.va;
      c->cfg.siv_part = 1;
      void *p = nullptr;
      if ((rc = dalloc(c, &p, 2 * sizeof(u32), true))) return rc;
      c->d_plog_n = (u32 *)p;
      c->cfg.p_log_n = c->d_plog_n;
    }
  }
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}
}  // namespace

// One GPU's capacity mode: the s- and b-mer tables as one chunk of physical memory per sub-table inside a reserved address
// range (the kernels see the same layout), so that a growth can re-insert sub-table by sub-table and hand each old chunk
// back before the next new one is made: peak = the new table + one old sub-table, instead of old + new side by side.
int fqsx_dna_use_chunked_tables(fqsx_dna *c) {
  if (!c) { g_err = "null argument"; return FQSX_E_ARG; }
  if (c->part) return FQSX_OK;
  if (c->k_n[0] || c->shard_world > 1) { g_err = "chunked tables are chosen before the first block (sharded codecs: fqsx_shard_partition_tables)"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  return tables_to_chunks(c);
}

// Collective, after fqsx_shard_attach and before the first block: from here on this rank holds only the sub-tables of the
// s- and b-mer tables its workers own; the others' are mapped from their ranks (one node: descriptors over Unix sockets).
int fqsx_shard_partition_tables(fqsx_dna *c) {
  if (!c || !c->comm_set) { g_err = "no transport attached (fqsx_shard_attach)"; return FQSX_E_ARG; }
  if (c->part && (c->shard_world == 1 || !c->mesh.peer.empty())) return FQSX_OK;
  if (c->part || c->k_n[0]) { g_err = "the tables can only be partitioned before the first block"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 G = c->shard_world, me = c->shard_rank;
  int rc;
  c->part_fallback = false;
#ifndef FQSX_EMU
  // ---- can every GPU of the world load from every other one's memory?  The PCI bus ids are all-gathered (device ordinals are
  // per process), every rank asks the runtime about its peers, and one all-reduced word makes the answer the same everywhere:
  // if any pair cannot, ALL ranks keep table replicas (the mode fqsx_shard_attach leaves the codec in) -- slower to update,
  // G times the table memory, same streams -- and say so.
  if (G > 1) {
    u64 mine[2] = {0, 0};
    char bus[16] = {0};
    HIPCHK(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->device));
    memcpy(mine, bus, sizeof(mine));
    std::vector<u64> all(2ull * G, 0);
    if ((rc = xbuf_fit(c, c->d_items, c->items_cap, 2))) return rc;
    if ((rc = xbuf_fit(c, c->d_gathered, c->gathered_cap, 2ull * G))) return rc;
    if ((rc = h2d(c, c->d_items, mine, sizeof(mine)))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    COMMCHK(c->comm.allgather_u64(c->comm.ctx, c->d_items, 2, c->d_gathered), "all-gather of the ranks' PCI bus ids");
    if ((rc = d2h_sync(c, all.data(), c->d_gathered, all.size() * sizeof(u64)))) return rc;
    std::string why;
    for (u32 q = 0; q < G && why.empty(); ++q) {
      if (q == me) continue;
      char pb[17] = {0};
      memcpy(pb, &all[2ull * q], 16);
      int pd = -1, can = 0;
      if (hipDeviceGetByPCIBusId(&pd, pb) != hipSuccess) { (void)hipGetLastError(); why = std::string("the GPU of rank ") + std::to_string(q) + " (" + pb + ") is not visible to rank " + std::to_string(me); }
      else if (pd != c->device && (hipDeviceCanAccessPeer(&can, c->device, pd) != hipSuccess || !can)) { (void)hipGetLastError(); why = std::string("rank ") + std::to_string(me) + " (" + bus + ") has no peer access to the GPU of rank " + std::to_string(q) + " (" + pb + ")"; }
    }
    u32 v = why.empty() ? 0u : 1u;
    u32 *word = c->d_cglob;
    if ((rc = h2d(c, word, &v, sizeof(v)))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    COMMCHK(c->comm.allreduce_sum_u32(c->comm.ctx, word, 1), "all-reduce of the peer-access check");
    if ((rc = d2h_sync(c, &v, word, sizeof(v)))) return rc;
    if (v) {
      c->part_fallback = true;
      g_err = "partitioned tables need peer access between all GPUs of the world: " + (why.empty() ? std::string("another rank cannot reach a peer") : why) +
              "; every rank keeps table replicas instead (fqsx_shard_is_partitioned() == 0)";
      return FQSX_OK;
    }
  }
#endif
  // ---- the descriptor mesh: listen, exchange the names through the transport, connect
  VMCHK(fqsx_vm::mesh_listen(c->mesh, me, G, e_));
  std::vector<u64> words(G, 0);
  if ((rc = xbuf_fit(c, c->d_items, c->items_cap, 1))) return rc;
  if ((rc = xbuf_fit(c, c->d_gathered, c->gathered_cap, G))) return rc;
  if ((rc = h2d(c, c->d_items, &c->mesh.word, sizeof(u64)))) return rc;
  COMMCHK(c->comm.allgather_u64(c->comm.ctx, c->d_items, 1, c->d_gathered), "all-gather of the descriptor-socket names");
  if ((rc = d2h_sync(c, words.data(), c->d_gathered, G * sizeof(u64)))) return rc;
  VMCHK(fqsx_vm::mesh_connect(c->mesh, words.data(), e_));
  // ---- the (still empty) tables again, partitioned; nobody looks a k-mer up before every rank's chunks are cleared
  if ((rc = tables_to_chunks(c))) return rc;
  COMMCHK(c->comm.allgather_u64(c->comm.ctx, c->d_items, 1, c->d_gathered), "barrier after the table partition");
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  return FQSX_OK;
}

int fqsx_shard_encode_block(fqsx_dna *c, const uint8_t *bases, const uint64_t *off, const uint64_t *h_off, uint32_t n_reads,
                            uint32_t generation, const uint8_t **streams, uint64_t *lens) {
  if (!c || !bases || !off || !h_off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
  if (!c->comm_set) { g_err = "no transport attached (fqsx_shard_attach)"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  int rc = block_prepare(c, bases, off, h_off, n_reads, generation);
  for (u32 seg = 0; !rc && seg <= c->cur_S; ++seg) rc = shard_phase_native(c, seg);
  if (rc) return rc;
  return block_finish(c, h_off, streams, lens, nullptr);
}

int fqsx_shard_is_partitioned(fqsx_dna *c) { return c && c->part && !c->part_fallback ? 1 : 0; }

int fqsx_shard_traffic(fqsx_dna *c, uint64_t out[4]) {
  if (!c || !out) return FQSX_E_ARG;
  out[0] = c->sh_phases; out[1] = c->sh_collectives; out[2] = c->sh_a2a_words; out[3] = c->sh_gather_words;
  return FQSX_OK;
}

// ---- RCCL transport (loaded on first use: the library itself does not link against librccl) ---------------------------
#ifndef FQSX_EMU
namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
bool rccl_load() {
  if (g_rccl.lib) return true;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { g_err = std::string("librccl not found: ") + dlerror(); return false; }
#define RSYM(field, name) do { *(void **)&g_rccl.field = dlsym(h, name); if (!g_rccl.field) { g_err = std::string("librccl lacks ") + name; return false; } } while (0)
  RSYM(GetUniqueId, "ncclGetUniqueId"); RSYM(CommInitRank, "ncclCommInitRank"); RSYM(CommDestroy, "ncclCommDestroy"); RSYM(CommAbort, "ncclCommAbort"); RSYM(CommCount, "ncclCommCount"); RSYM(CommUserRank, "ncclCommUserRank");
  RSYM(AllReduce, "ncclAllReduce"); RSYM(AllGather, "ncclAllGather"); RSYM(Send, "ncclSend"); RSYM(Recv, "ncclRecv");
  RSYM(GroupStart, "ncclGroupStart"); RSYM(GroupEnd, "ncclGroupEnd"); RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
  g_rccl.lib = h;
  return true;
}
struct RcclCtx { ncclComm_t comm; hipStream_t stream; u32 world; bool aborted; };
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { g_err = std::string(#x) + ": " + g_rccl.GetErrorString(r_); return 1; } } while (0)
int rccl_allreduce(void *ctx, uint32_t *buf, uint64_t n) {
  RcclCtx *x = (RcclCtx *)ctx;
  NCHK(g_rccl.AllReduce(buf, buf, n, ncclUint32, ncclSum, x->comm, x->stream));
  return 0;
}
int rccl_alltoallv(void *ctx, uint32_t n_buf, const uint64_t *const *send, const uint64_t *sc, uint64_t *const *recv, const uint64_t *rcnt) {
  RcclCtx *x = (RcclCtx *)ctx;
  NCHK(g_rccl.GroupStart());
  for (u32 b = 0; b < n_buf; ++b) {
    u64 so = 0, ro = 0;
    for (u32 r = 0; r < x->world; ++r) {
      const u64 ns = sc[(u64)b * x->world + r], nr = rcnt[(u64)b * x->world + r];
      if (ns) NCHK(g_rccl.Send(send[b] + so, ns, ncclUint64, (int)r, x->comm, x->stream));
      if (nr) NCHK(g_rccl.Recv(recv[b] + ro, nr, ncclUint64, (int)r, x->comm, x->stream));
      so += ns; ro += nr;
    }
  }
  NCHK(g_rccl.GroupEnd());
  return 0;
}
int rccl_allgather(void *ctx, const uint64_t *send, uint64_t n, uint64_t *recv) {
  RcclCtx *x = (RcclCtx *)ctx;
  NCHK(g_rccl.AllGather(send, recv, n, ncclUint64, x->comm, x->stream));
  return 0;
}
void rccl_abort(void *ctx) {
  RcclCtx *x = (RcclCtx *)ctx;
  if (x && !x->aborted) { x->aborted = true; (void)g_rccl.CommAbort(x->comm); }
}
}  // namespace
#endif

int fqsx_rccl_unique_id(uint8_t id[128]) {
#ifndef FQSX_EMU
  if (!id) return FQSX_E_ARG;
  if (!rccl_load()) return FQSX_E_HIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  if (g_rccl.GetUniqueId(&u) != ncclSuccess) { g_err = "ncclGetUniqueId failed"; return FQSX_E_HIP; }
  memcpy(id, &u, 128);
  return FQSX_OK;
#else
  (void)id;
  g_err = "the emulation build has no RCCL transport";
  return FQSX_E_NO_DEVICE;
#endif
}
int fqsx_rccl_comm_create(fqsx_dna *c, const uint8_t id[128], uint32_t rank, uint32_t world, fqsx_comm *out) {
#ifndef FQSX_EMU
  if (!c || !id || !out || world == 0 || rank >= world) { g_err = "bad argument"; return FQSX_E_ARG; }
  if (!rccl_load()) return FQSX_E_HIP;
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, 128);
  RcclCtx *x = new RcclCtx();
  x->stream = c->stream;
  x->world = world;
  x->aborted = false;
  ncclResult_t r = g_rccl.CommInitRank(&x->comm, (int)world, u, (int)rank);
  if (r != ncclSuccess) { g_err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete x; return FQSX_E_HIP; }
  out->ctx = x;
  out->allreduce_sum_u32 = rccl_allreduce;
  out->alltoallv_u64 = rccl_alltoallv;
  out->allgather_u64 = rccl_allgather;
  out->abort = rccl_abort;
  return FQSX_OK;
#else
  (void)c; (void)id; (void)rank; (void)world; (void)out;
  g_err = "the emulation build has no RCCL transport";
  return FQSX_E_NO_DEVICE;
#endif
}
// the communicator of one codec for the next one (a file per codec, one communicator per process: ncclCommInitRank takes
// seconds): from now on the collectives run on `c`'s stream
int fqsx_rccl_comm_rebind(fqsx_comm *m, fqsx_dna *c) {
#ifndef FQSX_EMU
  if (!m || !m->ctx || !c) { g_err = "bad argument"; return FQSX_E_ARG; }
  ((RcclCtx *)m->ctx)->stream = c->stream;
  return FQSX_OK;
#else
  (void)m; (void)c;
  g_err = "the emulation build has no RCCL transport";
  return FQSX_E_NO_DEVICE;
#endif
}
// out[0] = ranks of the communicator as RCCL counts them, out[1] = this process's rank in it
int fqsx_rccl_comm_info(fqsx_comm *m, uint32_t out[2]) {
#ifndef FQSX_EMU
  if (!m || !m->ctx || !out) { g_err = "bad argument"; return FQSX_E_ARG; }
  RcclCtx *x = (RcclCtx *)m->ctx;
  int n = 0, r = 0;
  if (g_rccl.CommCount(x->comm, &n) != ncclSuccess || g_rccl.CommUserRank(x->comm, &r) != ncclSuccess) { g_err = "ncclCommCount / ncclCommUserRank failed"; return FQSX_E_HIP; }
  out[0] = (u32)n; out[1] = (u32)r;
  return FQSX_OK;
#else
  (void)m; (void)out;
  g_err = "the emulation build has no RCCL transport";
  return FQSX_E_NO_DEVICE;
#endif
}
void fqsx_rccl_comm_destroy(fqsx_comm *m) {
#ifndef FQSX_EMU
  if (!m || !m->ctx) return;
  RcclCtx *x = (RcclCtx *)m->ctx;
  if (g_rccl.CommDestroy && !x->aborted) (void)g_rccl.CommDestroy(x->comm);
  delete x;
  m->ctx = nullptr;
#else
  (void)m;
#endif
}

int fqsx_dna_stats(fqsx_dna *c, uint64_t out[64]) {
  if (!c || !out) return FQSX_E_ARG;
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  LAUNCH(c, 2, k_gather_stats, 1, 64, c->cfg, c->d_lens + c->T);
  return d2h_sync(c, out, c->d_lens + c->T, 64 * sizeof(u64));
}

int fqsx_dna_capacity(fqsx_dna *c, uint64_t out[16]) {
  if (!c || !out) return FQSX_E_ARG;
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(c->device));
#endif
  const u32 T = c->T;
  std::vector<u32> f(T);
  int rc;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if ((rc = d2h_sync(c, f.data(), c->cfg.g_s.filled, T * sizeof(u32)))) return rc;
  for (u32 t = 0; t < T; ++t) out[0] += f[t];
  if ((rc = d2h_sync(c, f.data(), c->cfg.g_b.filled, T * sizeof(u32)))) return rc;
  for (u32 t = 0; t < T; ++t) out[1] += f[t];
  out[2] = c->gs_cap * T;
  out[3] = c->gb_cap * T;
  out[4] = (1ull << (2 * c->cfg.pmer)) / 4;
  out[5] = c->ctx_cap * T;
  if ((rc = d2h_sync(c, f.data(), c->cfg.ctx_filled, T * sizeof(u32)))) return rc;
  for (u32 t = 0; t < T; ++t) out[6] += f[t];
  out[7] = c->dev_bytes;
  out[8] = c->dev_bytes_peak;
  out[9] = c->n_growths;
  if (c->paired) {
    if ((rc = d2h_sync(c, f.data(), c->cfg.g_pe.filled, T * sizeof(u32)))) return rc;
    for (u32 t = 0; t < T; ++t) out[10] += f[t];
    out[11] = c->gpe_cap * T;
  }
  out[12] = sizeof(u64);   // bytes per global-table slot
  out[13] = (c->vm_s.live ? c->vm_s.own_bytes : c->gs_cap * T * sizeof(u64)) + (c->vm_b.live ? c->vm_b.own_bytes : c->gb_cap * T * sizeof(u64));   // s- + b-mer table memory this rank holds
  out[15] = c->vm_siv.live ? c->vm_siv.own_bytes : out[4];   // p-mer vector memory this rank holds
  if (c->paired) out[14] = c->vm_pk.live ? c->vm_pk.own_bytes + c->vm_pv.own_bytes : c->gpe_cap * T * 2 * sizeof(u64);   // pair-table memory this rank holds
  return FQSX_OK;
}

// timing builds: the role time stamps of the first `max_launches` encode launches ([launch][worker][FQSX_TRACE_WORDS]: 8 stamps
// in 10 ns ticks + 24 counters); returns the number of launches copied (0 in product builds)
static_assert(FQSX_TRACE_WORDS == FQSX_TRACE_W, "include/fqsx.h documents the trace record width");
int fqsx_dna_trace(fqsx_dna *c, uint64_t *out, uint32_t max_launches) {
  if (!c || !out || !c->cfg.trace) return 0;
  u32 n = (u32)std::min<u64>(std::min<u64>(c->k_n[0], FQSX_TRACE_LAUNCHES), max_launches);
  if (n && d2h_sync(c, out, c->cfg.trace, (u64)n * c->T * FQSX_TRACE_W * sizeof(u64))) return 0;
  return (int)n;
}

int fqsx_dna_set_profiling(fqsx_dna *c, int enable) {
  if (!c) return FQSX_E_ARG;
  c->profiling = enable != 0;
  return FQSX_OK;
}
int fqsx_dna_kernel_times(fqsx_dna *c, double out[6]) {
  if (!c || !out) return FQSX_E_ARG;
  for (int i = 0; i < 3; ++i) { out[i] = c->k_ms[i]; out[3 + i] = (double)c->k_n[i]; }
  return FQSX_OK;
}

}  // extern "C"

// =======================================================================================
// quality stream
FQ_KERNEL64 void k_qual_encode(QualCfg cfg, u32 n_reads) {
  FQ_SHARED u8 lds_q[4096 + 96];
  qual_encode_body(cfg, lds_q, FQ_BLOCK, n_reads);
}
FQ_KERNEL void k_qual_rehash(const u64 *o, u64 ocap_mask, u64 *n, u64 ncap_mask, u32 T, u32 slot_u64) {
  const u64 ocap = ocap_mask + 1, total = ocap * T;
#ifndef FQSX_EMU
  const u64 gstride = (u64)gridDim.x * blockDim.x;
  for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gstride) {
#else
  for (u64 g = 0; g < total; ++g) {
#endif
    const u64 *src = o + g * slot_u64;
    const u64 key = src[0];
    if (key == ~0ull) continue;
    const u32 w = (u32)(g / ocap);
    u64 *b = n + (u64)w * (ncap_mask + 1) * slot_u64;
    u64 h = q_hash(key) & ncap_mask;
    for (;;) {
      u64 *p = b + h * slot_u64;
      if (p[0] == ~0ull && atomic_cas64(&p[0], ~0ull, key) == ~0ull) {
        for (u32 i = 1; i < slot_u64; ++i) p[i] = src[i];
        break;
      }
      h = (h + 1) & ncap_mask;
    }
  }
}

struct fqsx_qual {
  QualCfg cfg;
  u32 T;
  int device;
  u64 cap, out_cap, q_cap, off_cap;
  u8 *d_q;
  u64 *d_off;
  fqsx_dna mem;   // allocation bookkeeping / stream (reuses the helpers above)
  std::vector<u32> h_filled;
  std::vector<u64> h_lens;
  std::vector<u8> h_out;
};

extern "C" {

// Kernel timing of the quality coder: HIP events around every k_qual_encode launch on the codec's stream (as
// fqsx_dna_set_profiling / fqsx_dna_kernel_times); out[0] = accumulated milliseconds, out[1] = launches
int fqsx_qual_set_profiling(fqsx_qual *q, int enable) {
  if (!q) return FQSX_E_ARG;
  q->mem.profiling = enable != 0;
  return FQSX_OK;
}
int fqsx_qual_kernel_times(fqsx_qual *q, double out[2]) {
  if (!q || !out) return FQSX_E_ARG;
  out[0] = q->mem.k_ms[0];
  out[1] = (double)q->mem.k_n[0];
  return FQSX_OK;
}
void fqsx_qual_destroy(fqsx_qual *q) {
  if (!q) return;
  fqsx_dna *c = &q->mem;
#ifndef FQSX_EMU
  (void)hipSetDevice(q->device);
  (void)hipStreamSynchronize(c->stream);
#endif
  std::vector<void *> a = c->allocs;
  for (void *p : a) dfree(c, p);
#ifndef FQSX_EMU
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
#endif
  delete q;
}

int fqsx_qual_create(const uint8_t *h, int device, fqsx_qual **out) {
  if (!h || !out || h[0] != 'K' || h[1] != 'C' || h[2] != 'S' || h[3] != 'D' || h[4] == 0 || h[6] > 3) {
    g_err = "malformed header or quality_mode none";
    return FQSX_E_ARG;
  }
#ifndef FQSX_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (libfqsx has no CPU path)"; return FQSX_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_err = "bad device ordinal"; return FQSX_E_ARG; }
  HIPCHK(hipSetDevice(device));
#endif
  fqsx_qual *q = new fqsx_qual();
  q->T = h[4];
  q->device = device;
  fqsx_dna *c = &q->mem;
  c->T = q->T; c->device = device; c->profiling = false;
#ifndef FQSX_EMU
  HIPCHK(hipStreamCreate(&c->stream));
  HIPCHK(hipEventCreate(&c->ev0));
  HIPCHK(hipEventCreate(&c->ev1));
#endif
  QualCfg &cfg = q->cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.T = q->T;
  cfg.mode = h[6];
  // Init + adjust_quality_map_*, quality.cpp:32-149
  auto band = [&](int a, int b, u8 v) { for (int i = a; i < b; ++i) cfg.fwd[i] = v; };
  const u32 thr = h[8];
  if (cfg.mode == 0) { cfg.n_sym = 96; cfg.bits = 6; cfg.nctx = 2; for (int i = 0; i < 96; ++i) cfg.fwd[i] = (u8)i; }
  else if (cfg.mode == 1) { cfg.n_sym = 8; cfg.bits = 4; cfg.nctx = 6; band(0, 2, 0); band(2, 10, 1); band(10, 20, 2); band(20, 25, 3); band(25, 30, 4); band(30, 35, 5); band(35, 40, 6); band(40, 96, 7); }
  else if (cfg.mode == 2) { cfg.n_sym = 4; cfg.bits = 3; cfg.nctx = 9; band(0, 2, 0); band(2, 15, 1); band(15, 31, 2); band(31, 96, 3); }
  else { cfg.n_sym = 2; cfg.bits = 2; cfg.nctx = 10; band(0, (int)std::min(thr, 96u), 0); band((int)std::min(thr, 96u), 96, 1); }
  cfg.ctx_mask = (1ull << (cfg.bits * cfg.nctx)) - 1ull;
  cfg.slot_u64 = 1 + (cfg.n_sym + 1 + 3) / 4;
  q->cap = 1u << 12;
  q->out_cap = q->q_cap = q->off_cap = 0;
  q->d_q = nullptr; q->d_off = nullptr;
  int rc;
  void *p = nullptr;
  auto fail = [&](int r) { fqsx_qual_destroy(q); return r; };
  if ((rc = dalloc(c, &p, q->cap * q->T * cfg.slot_u64 * sizeof(u64), false))) return fail(rc);
  cfg.tab = (u64 *)p;
#ifndef FQSX_EMU
  HIPCHK(hipMemsetAsync(cfg.tab, 0xff, q->cap * q->T * cfg.slot_u64 * sizeof(u64), c->stream));
#else
  memset(cfg.tab, 0xff, q->cap * q->T * cfg.slot_u64 * sizeof(u64));
#endif
  cfg.cap_mask = q->cap - 1;
  if ((rc = dalloc(c, &p, q->T * sizeof(u32), true))) return fail(rc);
  cfg.filled = (u32 *)p;
  if ((rc = dalloc(c, &p, (q->T + 2) * sizeof(u64), true))) return fail(rc);   // lengths and, behind them, the error word: one transfer
  cfg.lens = (u64 *)p;
  cfg.err = (u32 *)(cfg.lens + q->T);
  q->h_filled.assign(q->T, 0);
  q->h_lens.assign(q->T + 2, 0);
  *out = q;
  return FQSX_OK;
}

static int qual_encode_impl(fqsx_qual *q, const uint8_t *quals, const uint8_t *d_quals, const uint64_t *d_off_in, const uint64_t *off,
                            uint32_t n_reads, const uint8_t **streams, uint64_t *lens);
int fqsx_qual_encode_block(fqsx_qual *q, const uint8_t *quals, const uint64_t *off, uint32_t n_reads, const uint8_t **streams,
                           uint64_t *lens) {
  if (!q || !quals || !off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
  return qual_encode_impl(q, quals, nullptr, nullptr, off, n_reads, streams, lens);
}
int fqsx_qual_encode_block_dev(fqsx_qual *q, const uint8_t *d_quals, const uint64_t *d_off, const uint64_t *h_off, uint32_t n_reads,
                               const uint8_t **streams, uint64_t *lens) {
  if (!q || !d_quals || !d_off || !h_off || !streams || !lens) { g_err = "null argument"; return FQSX_E_ARG; }
  return qual_encode_impl(q, nullptr, d_quals, d_off, h_off, n_reads, streams, lens);
}
// quals (host) or d_quals / d_off_in (already in device memory); off = host copy of the offsets
static int qual_encode_impl(fqsx_qual *q, const uint8_t *quals, const uint8_t *d_quals, const uint64_t *d_off_in, const uint64_t *off,
                            uint32_t n_reads, const uint8_t **streams, uint64_t *lens) {
  fqsx_dna *c = &q->mem;
  QualCfg &cfg = q->cfg;
  const u32 T = q->T;
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(q->device));
#endif
  int rc;
  void *p = nullptr;
  u64 max_w = 0;
  for (u32 t = 0; t < T; ++t) {
    u64 first = (u64)t * n_reads / T, last = ((u64)t + 1) * n_reads / T;
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    max_w = std::max(max_w, off[last] - off[first]);
  }
  // context table: every symbol can create one context
  if ((rc = d2h_sync(c, q->h_filled.data(), cfg.filled, T * sizeof(u32)))) return rc;
  u64 need = 0;
  for (u32 t = 0; t < T; ++t) need = std::max<u64>(need, (u64)q->h_filled[t] + max_w + 64);
  if (need * 2 > q->cap) {
    const u64 ncap = pow2_at_least(need * 2);
    if ((rc = dalloc(c, &p, ncap * T * cfg.slot_u64 * sizeof(u64), false))) return rc;
#ifndef FQSX_EMU
    HIPCHK(hipMemsetAsync(p, 0xff, ncap * T * cfg.slot_u64 * sizeof(u64), c->stream));
#else
    memset(p, 0xff, ncap * T * cfg.slot_u64 * sizeof(u64));
#endif
    LAUNCH(c, 2, k_qual_rehash, REHASH_GRID, 256, (const u64 *)cfg.tab, cfg.cap_mask, (u64 *)p, ncap - 1, T, cfg.slot_u64);
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
    dfree(c, cfg.tab);
    cfg.tab = (u64 *)p;
    cfg.cap_mask = ncap - 1;
    q->cap = ncap;
  }
  const u64 need_out = (max_w * 2 + 4096 + 7) & ~7ull, nq = off[n_reads] + 64, no = ((u64)n_reads + 1) * sizeof(u64);
  if (need_out > q->out_cap) {
    dfree(c, cfg.out);
    if ((rc = dalloc(c, &p, need_out * T, false))) return rc;
    cfg.out = (u8 *)p; cfg.out_cap = q->out_cap = need_out;
  }
  if (d_quals) {
    cfg.quals = d_quals;
    cfg.off = d_off_in;
  } else {
    if (nq > q->q_cap) { dfree(c, q->d_q); if ((rc = dalloc(c, &p, nq + nq / 4, false))) return rc; q->d_q = (u8 *)p; q->q_cap = nq + nq / 4; }
    if (no > q->off_cap) { dfree(c, q->d_off); if ((rc = dalloc(c, &p, no + no / 4, false))) return rc; q->d_off = (u64 *)p; q->off_cap = no + no / 4; }
    if ((rc = h2d(c, q->d_q, quals, off[n_reads]))) return rc;
    if ((rc = h2d(c, q->d_off, off, no))) return rc;
    cfg.quals = q->d_q;
    cfg.off = q->d_off;
  }
  LAUNCH(c, 0, k_qual_encode, T, 64, cfg, n_reads);
  if ((rc = d2h_sync(c, q->h_lens.data(), cfg.lens, ((u64)T + 1) * sizeof(u64)))) return rc;
  const u32 err = (u32)q->h_lens[T];
  if (err) { g_err = "device error " + std::to_string(err) + " in the quality kernel"; return FQSX_E_DEVICE; }
  for (u32 t = 0; t < T; ++t)
    if (q->h_lens[t] > cfg.out_cap) { g_err = "quality stream overflow"; return FQSX_E_DEVICE; }
  return collect_streams(c, T, cfg.out, cfg.out_cap, cfg.lens, q->h_lens, q->h_out, streams, lens);
}

}  // extern "C"

// =======================================================================================================
// Read-id stream on the GPU: one wavefront per worker, models in two per-worker tables in HBM
// =======================================================================================================
#include "fqsx_idk.h"

FQ_KERNEL64 void k_id_encode(IdCfg cfg, u32 n_reads, u32 paired) {
  FQ_SHARED IdShared sm;
  id_encode_body(cfg, &sm, FQ_BLOCK, n_reads, paired);
}
// the twelve fixed models of every worker: all ones (mtf_flag 11 symbols, mtf_code[k] 2 << k, mtf_byte 256; id.cpp:84-105)
FQ_KERNEL64 void k_id_init_fixed(IdCfg cfg) {
  const u32 tid = FQ_BLOCK;
  const u32 n_of[IDK_FIXED] = {11, 2, 4, 8, 16, 32, 64, 128, 256, 256, 256, 256};
  for (u32 f = 0; f < IDK_FIXED; ++f) {
    u64 *slot = cfg.fixed + ((u64)tid * IDK_FIXED + f) * IDK_BIG_U64;
    for (u32 l = FQ_LANE; l < 64; l += FQ_WAVE) slot[1 + l] = 0x0001000100010001ULL;
    if (FQ_LANE == 0) { slot[0] = 0; slot[65] = n_of[f]; }
  }
}

struct fqsx_idg {
  IdCfg cfg;
  u32 T;
  int device;
  u64 small_cap, big_cap, out_cap, ids_cap, off_cap;
  u8 *d_ids;
  u64 *d_off;
  fqsx_dna mem;   // allocation bookkeeping / stream (reuses the helpers above)
  std::vector<u32> h_state;
  std::vector<u64> h_lens;
  std::vector<u8> h_out;
};

extern "C" {

void fqsx_idg_destroy(fqsx_idg *q) {
  if (!q) return;
  fqsx_dna *c = &q->mem;
#ifndef FQSX_EMU
  (void)hipSetDevice(q->device);
  (void)hipStreamSynchronize(c->stream);
#endif
  std::vector<void *> a = c->allocs;
  for (void *p : a) dfree(c, p);
#ifndef FQSX_EMU
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
#endif
  delete q;
}

static int idg_fill_ff(fqsx_dna *c, void *p, u64 bytes) {
#ifndef FQSX_EMU
  HIPCHK(hipMemsetAsync(p, 0xff, bytes, c->stream));
#else
  (void)c;
  memset(p, 0xff, bytes);
#endif
  return FQSX_OK;
}

int fqsx_idg_create(const uint8_t *h, int device, fqsx_idg **out) {
  if (!h || !out || memcmp(h, "KCSD", 4) || h[4] == 0 || h[7] > 1) { g_err = "malformed header or id_mode none"; return FQSX_E_ARG; }
#ifndef FQSX_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (libfqsx has no CPU path)"; return FQSX_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_err = "bad device ordinal"; return FQSX_E_ARG; }
  HIPCHK(hipSetDevice(device));
#endif
  fqsx_idg *q = new fqsx_idg();
  q->T = h[4];
  q->device = device;
  fqsx_dna *c = &q->mem;
  c->T = q->T; c->device = device; c->profiling = false;
#ifndef FQSX_EMU
  HIPCHK(hipStreamCreate(&c->stream));
  HIPCHK(hipEventCreate(&c->ev0));
  HIPCHK(hipEventCreate(&c->ev1));
#endif
  IdCfg &cfg = q->cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.T = q->T;
  cfg.mode = h[7];   // 0 lossless, 1 instrument (params.h:18,92)
  q->small_cap = 1u << 10; q->big_cap = 1u << 9;
  q->out_cap = q->ids_cap = q->off_cap = 0;
  q->d_ids = nullptr; q->d_off = nullptr;
  const u64 T = q->T;
  int rc;
  void *p = nullptr;
  auto fail = [&](int r) { fqsx_idg_destroy(q); return r; };
  if ((rc = dalloc(c, &p, q->small_cap * T * 2 * sizeof(u64), false))) return fail(rc);
  cfg.small = (u64 *)p; cfg.small_mask = q->small_cap - 1;
  if ((rc = idg_fill_ff(c, cfg.small, q->small_cap * T * 2 * sizeof(u64)))) return fail(rc);
  if ((rc = dalloc(c, &p, q->big_cap * T * IDK_BIG_U64 * sizeof(u64), false))) return fail(rc);
  cfg.big = (u64 *)p; cfg.big_mask = q->big_cap - 1;
  if ((rc = idg_fill_ff(c, cfg.big, q->big_cap * T * IDK_BIG_U64 * sizeof(u64)))) return fail(rc);
  if ((rc = dalloc(c, &p, T * IDK_FIXED * IDK_BIG_U64 * sizeof(u64), true))) return fail(rc);
  cfg.fixed = (u64 *)p;
  cfg.mtf_cap = 4096;
  if ((rc = dalloc(c, &p, T * cfg.mtf_cap * IDK_NAME, true))) return fail(rc);
  cfg.mtf = (u8 *)p;
  if ((rc = dalloc(c, &p, (T + 2 + 2 * T) * sizeof(u64), true))) return fail(rc);   // lengths, error word, per-worker state: one transfer
  cfg.lens = (u64 *)p;
  cfg.err = (u32 *)(cfg.lens + T);
  cfg.state = (u32 *)(cfg.lens + T + 2);
  LAUNCH(c, 2, k_id_init_fixed, q->T, 64, cfg);
#ifndef FQSX_EMU
  HIPCHK(hipStreamSynchronize(c->stream));
#endif
  q->h_state.assign(4 * T, 0);
  q->h_lens.assign(T + 2 + 2 * T, 0);
  *out = q;
  return FQSX_OK;
}

// ids: concatenated id lines, each INCLUDING its '\n' (read_desc_t::id_len, defs.h:70-72); id_off: n_reads + 1 offsets;
// paired != 0: reads alternate mate 1 / mate 2.  streams[t] / lens[t]: worker t's id stream of the block (callee-owned until
// the next call) -- byte-identical to fqsx_id_encode_block's (the host coder) and hence to the reference's.
int fqsx_idg_encode_block(fqsx_idg *q, const uint8_t *ids, const uint64_t *off, uint32_t n_reads, int paired,
                          const uint8_t **streams, uint64_t *lens) {
  if (!q || !ids || !off || !streams || !lens || (paired && (n_reads & 1))) { g_err = "bad argument"; return FQSX_E_ARG; }
  fqsx_dna *c = &q->mem;
  IdCfg &cfg = q->cfg;
  const u64 T = q->T;
#ifndef FQSX_EMU
  HIPCHK(hipSetDevice(q->device));
#endif
  int rc;
  void *p = nullptr;
  // ---- table sizes: a worker creates at most one big model per byte it codes plus nine per numeric token, and three small
  // ones per token plus two per read; tokens end at separators, which are counted here
  u64 max_bytes = 0, max_big = 0, max_small = 0;
  for (u64 t = 0; t < T; ++t) {
    u64 first = t * n_reads / T, last = (t + 1) * n_reads / T;  // reads_block.h:197-214
    if (t) first &= ~1ull;
    if (t + 1 < T) last &= ~1ull;
    const u64 nb = off[last] - off[first];
    u64 sep = 0;
    for (u64 i = off[first]; i < off[last]; ++i) {
      const u8 ch = ids[i];
      sep += !((ch >= '0' && ch <= '9') || (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || ch == '@');
    }
    max_bytes = std::max(max_bytes, nb);
    max_big = std::max(max_big, (u64)q->h_state[4 * t + 1] + nb + 9 * sep + 2 * (last - first) + 16);
    max_small = std::max(max_small, (u64)q->h_state[4 * t] + 3 * sep + 4 * (last - first) + 16);
  }
  auto regrow = [&](u64 *&tab, u64 &cap, u64 &mask, u64 need, u32 slot_u64) -> int {
    if (need * 10 < cap * 8) return FQSX_OK;
    const u64 ncap = pow2_at_least(need * 2);
    int r = dalloc(c, &p, ncap * T * slot_u64 * sizeof(u64), false);
    if (r) return r;
    if ((r = idg_fill_ff(c, p, ncap * T * slot_u64 * sizeof(u64)))) return r;
    LAUNCH(c, 2, k_qual_rehash, REHASH_GRID, 256, (const u64 *)tab, mask, (u64 *)p, ncap - 1, (u32)T, slot_u64);
#ifndef FQSX_EMU
    HIPCHK(hipStreamSynchronize(c->stream));
#endif
    dfree(c, tab);
    tab = (u64 *)p; cap = ncap; mask = ncap - 1;
    return FQSX_OK;
  };
  if ((rc = regrow(cfg.small, q->small_cap, cfg.small_mask, max_small, 2))) return rc;
  if ((rc = regrow(cfg.big, q->big_cap, cfg.big_mask, max_big, IDK_BIG_U64))) return rc;
  const u64 need_out = (max_bytes * 2 + 4096 + 7) & ~7ull, ni = off[n_reads] + 64, no = ((u64)n_reads + 1) * sizeof(u64);
  if (need_out > q->out_cap) {
    dfree(c, cfg.out);
    if ((rc = dalloc(c, &p, need_out * T, false))) return rc;
    cfg.out = (u8 *)p; cfg.out_cap = q->out_cap = need_out;
  }
  if (ni > q->ids_cap) { dfree(c, q->d_ids); if ((rc = dalloc(c, &p, ni + ni / 4, false))) return rc; q->d_ids = (u8 *)p; q->ids_cap = ni + ni / 4; }
  if (no > q->off_cap) { dfree(c, q->d_off); if ((rc = dalloc(c, &p, no + no / 4, false))) return rc; q->d_off = (u64 *)p; q->off_cap = no + no / 4; }
  if ((rc = h2d(c, q->d_ids, ids, off[n_reads]))) return rc;
  if ((rc = h2d(c, q->d_off, off, no))) return rc;
  cfg.ids = q->d_ids;
  cfg.off = q->d_off;
  LAUNCH(c, 0, k_id_encode, q->T, 64, cfg, n_reads, (u32)(paired != 0));
  if ((rc = d2h_sync(c, q->h_lens.data(), cfg.lens, (T + 2 + 2 * T) * sizeof(u64)))) return rc;
  memcpy(q->h_state.data(), q->h_lens.data() + T + 2, 4 * T * sizeof(u32));
  const u32 err = (u32)q->h_lens[T];
  if (err) {
    static const char *what[] = {"", "byte outside the 128-symbol alphabet", "no instrument name", "stream overflow", "model table full",
                                 "id, token count or instrument name beyond the kernel's staging sizes", "more than 4096 instrument names"};
    g_err = std::string("id kernel: ") + (err < 7 ? what[err] : "error");
    return err == IDK_ERR_BYTE || err == IDK_ERR_NO_INSTRUMENT ? FQSX_E_ARG : FQSX_E_DEVICE;
  }
  for (u64 t = 0; t < T; ++t)
    if (q->h_lens[t] > cfg.out_cap) { g_err = "id stream overflow"; return FQSX_E_DEVICE; }
  return collect_streams(c, (u32)T, cfg.out, cfg.out_cap, cfg.lens, q->h_lens, q->h_out, streams, lens);
}

}  // extern "C"

// =======================================================================================================
// Read ordering of sorted mode: GPU radix sort + dense ranks, host replay of std::sort
// =======================================================================================================
#include "fqsx_sort.h"

FQ_KERNEL64 void k_sort_info(SortCfg c) {
  FQ_SHARED u32 lds[64 * 10];
  sort_info_body(c, FQ_BLOCK, lds);
}
FQ_KERNEL64 void k_sort_iota(SortCfg c) { sort_iota_body(c, FQ_BLOCK); }
FQ_KERNEL64 void k_sort_count(SortCfg c) {
  FQ_SHARED u32 hist[256];
  sort_count_body(c, FQ_BLOCK, hist);
}
FQ_KERNEL64 void k_sort_scan(SortCfg c) { sort_scan_body(c, FQ_BLOCK); }
FQ_KERNEL64 void k_sort_dstoff(SortCfg c) { sort_dstoff_body(c); }
FQ_KERNEL64 void k_sort_scatter(SortCfg c) {
  FQ_SHARED u32 cursor[256];
  FQ_SHARED u32 ld[64];
  sort_scatter_body(c, FQ_BLOCK, cursor, ld);
}
FQ_KERNEL64 void k_sort_flags(SortCfg c) { sort_flags_body(c, FQ_BLOCK); }
FQ_KERNEL64 void k_sort_rank(SortCfg c) { sort_rank_body(c); }

static int sort_rank_impl(fqsx_dna *c, const u8 *bases, const u64 *off, u32 n, std::vector<u32> &rank, u32 *passes_out) {
  int rc;
  void *p = nullptr;
  SortCfg s;
  memset(&s, 0, sizeof(s));
  s.n = n;
  s.n_tiles = (n + FQSX_SORT_TILE - 1) / FQSX_SORT_TILE;
  const u64 nb = off[n];
  if ((rc = dalloc(c, &p, nb ? nb : 8, false))) return rc;
  u8 *d_bases = (u8 *)p;
  if ((rc = dalloc(c, &p, ((u64)n + 1) * sizeof(u64), false))) return rc;
  u64 *d_off = (u64 *)p;
  if ((rc = h2d(c, d_bases, bases, nb))) return rc;
  if ((rc = h2d(c, d_off, off, ((u64)n + 1) * sizeof(u64)))) return rc;
  s.bases = d_bases; s.off = d_off;
  if ((rc = dalloc(c, &p, (u64)n * 4, false))) return rc; s.perm_in = (u32 *)p;
  if ((rc = dalloc(c, &p, (u64)n * 4, false))) return rc; s.perm_out = (u32 *)p;
  if ((rc = dalloc(c, &p, (u64)s.n_tiles * 256 * 4, false))) return rc; s.tile_hist = (u32 *)p;
  if ((rc = dalloc(c, &p, 256 * 4, false))) return rc; s.dig_tot = (u32 *)p;
  if ((rc = dalloc(c, &p, 257 * 4, false))) return rc; s.dig_off = (u32 *)p;
  if ((rc = dalloc(c, &p, (u64)n * 4, false))) return rc; s.flags = (u32 *)p;
  if ((rc = dalloc(c, &p, (u64)n * 4, false))) return rc; s.rank = (u32 *)p;
  if ((rc = dalloc(c, &p, (u64)s.n_tiles * 10 * 4, false))) return rc; s.info = (u32 *)p;
  LAUNCH(c, 2, k_sort_info, s.n_tiles, 64, s);
  LAUNCH(c, 2, k_sort_iota, s.n_tiles, 64, s);
  std::vector<u32> info((u64)s.n_tiles * 10);
  if ((rc = d2h_sync(c, info.data(), s.info, info.size() * 4))) return rc;
  u32 mn = 0xffffffffu, mx = 0, set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (u32 t = 0; t < s.n_tiles; ++t) {
    mn = std::min(mn, info[(u64)t * 10]);
    mx = std::max(mx, info[(u64)t * 10 + 1]);
    for (u32 k = 0; k < 8; ++k) set[k] |= info[(u64)t * 10 + 2 + k];
  }
  // raw-byte tie-break (io.h:518-526): only bytes outside A/C/G can differ between reads equal under N->T
  u32 n_other = 0;
  for (u32 b = 0; b < 256; ++b)
    if ((set[b >> 5] >> (b & 31)) & 1) s.raw_code[b] = (u8)(++n_other);
  u32 passes = 0;
  auto pass = [&](u32 kind, u32 arg) -> int {
    s.kind = kind; s.arg = arg;
    LAUNCH(c, 2, k_sort_count, s.n_tiles, 64, s);
    LAUNCH(c, 2, k_sort_scan, 256, 64, s);
    LAUNCH(c, 2, k_sort_dstoff, 1, 64, s);
    LAUNCH(c, 2, k_sort_scatter, s.n_tiles, 64, s);
    std::swap(s.perm_in, s.perm_out);
    ++passes;
    return FQSX_OK;
  };
  // least significant key first
  if (n_other >= 2) {
    s.raw_bits = n_other <= 3 ? 2 : n_other <= 15 ? 4 : 8;
    const u32 per = 8 / s.raw_bits;
    for (u32 b = (mx + per - 1) / per; b-- > 0;) if ((rc = pass(SORT_RAW, b))) return rc;
  }
  if (mn != mx)
    for (u32 k = 0; k < 4 && (mx >> (8 * k)); ++k) if ((rc = pass(SORT_LEN, k))) return rc;
  for (u32 b = (mx + 3) / 4; b-- > 0;) if ((rc = pass(SORT_NT, b))) return rc;
  LAUNCH(c, 2, k_sort_flags, s.n_tiles, 64, s);
  LAUNCH(c, 2, k_sort_rank, 1, 64, s);
  rank.resize(n);
  if ((rc = d2h_sync(c, rank.data(), s.rank, (u64)n * 4))) return rc;
  if (passes_out) *passes_out = passes;
  return FQSX_OK;
}

// The reference keeps device^Whost memory bounded by binning to disk first (preprocess_se, application.cpp:349-412) and
// sorting bin after bin.  Same structure here: the reads are binned on the host by their first four bases (the bins are
// ranges of the sort order), consecutive bins are packed into batches of at most `max_batch_bases` bases, and every batch
// goes through the GPU sort on its own -- device memory is bounded by the largest batch (at least the largest bin), not
// by the file.  max_batch_bases = 0: one batch (the whole file in one allocation).
extern "C" int fqsx_sort_order_batched(const uint8_t *bases, const uint64_t *read_off, uint32_t n_reads, int device, uint64_t max_batch_bases,
                                       uint32_t *order_out, uint32_t *bin_start /*[257]*/, uint32_t *n_batches_out) {
  if (!bases || !read_off || !order_out || !bin_start) { g_err = "null argument"; return FQSX_E_ARG; }
  for (u32 b = 0; b <= 256; ++b) bin_start[b] = 0;
  if (n_batches_out) *n_batches_out = 0;
  if (n_reads == 0) return FQSX_OK;
#ifndef FQSX_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (libfqsx has no CPU path)"; return FQSX_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_err = "bad device ordinal"; return FQSX_E_ARG; }
  HIPCHK(hipSetDevice(device));
#endif
  // bins in input order (preprocess_se, application.cpp:383-391: a position past the read counts as code 3 there
  // because it lands on the line feed)
  auto nt = [](u8 ch) -> u32 { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u; };
  std::vector<u8> bin(n_reads);
  std::vector<u64> bin_bases(256, 0);
  for (u32 r = 0; r < n_reads; ++r) {
    const u64 b = read_off[r];
    const u32 len = (u32)(read_off[r + 1] - b);
    u32 id = 0;
    for (u32 i = 0; i < 4; ++i) id = (id << 2) | (i < len ? nt(bases[b + i]) : 3u);
    bin[r] = (u8)id;
    ++bin_start[id + 1];
    bin_bases[id] += len;
  }
  for (u32 b = 0; b < 256; ++b) bin_start[b + 1] += bin_start[b];
  std::vector<u32> cur(bin_start, bin_start + 256);
  std::vector<u32> member(n_reads);   // reads grouped by bin, input order inside a bin
  for (u32 r = 0; r < n_reads; ++r) member[cur[bin[r]]++] = r;
  fqsx_dna mem{};
  fqsx_dna *c = &mem;
  c->T = 1; c->device = device; c->profiling = false;
#ifndef FQSX_EMU
  HIPCHK(hipStreamCreate(&c->stream));
#endif
  int rc = FQSX_OK;
  u32 n_batches = 0;
  std::vector<u8> gb;
  std::vector<u64> go;
  std::vector<u32> rank;
  std::vector<std::pair<u32, u32>> v;   // (rank, read)
  for (u32 b0 = 0; b0 < 256 && !rc;) {
    u32 b1 = b0 + 1;
    u64 nb = bin_bases[b0];
    while (b1 < 256 && (max_batch_bases == 0 || nb + bin_bases[b1] <= max_batch_bases)) nb += bin_bases[b1++];
    const u32 lo = bin_start[b0], hi = bin_start[b1], n = hi - lo;
    if (n) {
      const u8 *pb = bases;
      const u64 *po = read_off;
      const bool whole = lo == 0 && hi == n_reads;
      if (!whole) {   // the batch's reads side by side (what a bin file of the reference holds)
        gb.resize(nb ? nb : 1);
        go.resize((u64)n + 1);
        u64 w = 0;
        for (u32 i = 0; i < n; ++i) {
          const u32 r = member[lo + i];
          const u64 len = read_off[r + 1] - read_off[r];
          go[i] = w;
          memcpy(gb.data() + w, bases + read_off[r], len);
          w += len;
        }
        go[n] = w;
        pb = gb.data(); po = go.data();
      }
      rc = sort_rank_impl(c, pb, po, n, rank, nullptr);
#ifndef FQSX_EMU
      (void)hipStreamSynchronize(c->stream);
#endif
      std::vector<void *> a = c->allocs;
      for (void *p : a) dfree(c, p);
      if (rc) break;
      // libstdc++'s std::sort per bin on the ranks (ranks of one batch are comparable; a bin never spans batches)
      v.resize(n);
      for (u32 i = 0; i < n; ++i) v[i] = std::make_pair(whole ? rank[member[lo + i]] : rank[i], member[lo + i]);
      for (u32 b = b0; b < b1; ++b)
        std::sort(v.begin() + (bin_start[b] - lo), v.begin() + (bin_start[b + 1] - lo),
                  [](const std::pair<u32, u32> &x, const std::pair<u32, u32> &y) { return x.first < y.first; });
      for (u32 i = 0; i < n; ++i) order_out[lo + i] = v[i].second;
      ++n_batches;
    }
    b0 = b1;
  }
#ifndef FQSX_EMU
  (void)hipStreamDestroy(c->stream);
#endif
  if (n_batches_out) *n_batches_out = n_batches;
  return rc;
}

extern "C" int fqsx_sort_order(const uint8_t *bases, const uint64_t *read_off, uint32_t n_reads, int device, uint32_t *order_out,
                               uint32_t *bin_start /*[257]*/) {
  return fqsx_sort_order_batched(bases, read_off, n_reads, device, 0, order_out, bin_start, nullptr);
}

#ifdef FQSX_EMU   // the emulation build is one translation unit
#include "fqsx_k_se.hip"
#include "fqsx_k_pe.hip"
#include "fqsx_k_dec.hip"
#endif
