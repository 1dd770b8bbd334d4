// fqsx_dev.h -- device code of the FQSX DNA encoder (gfx950; wave-uniform, see fqsx_plat.h).
//
// Bit-exact re-design of the FQSqueezer 1.1 DNA hot path.  Each function cites the
// reference lines whose *behaviour* it reproduces (paths relative to /root/reference/fqs);
// the data structures and the execution model are this repo's own (fqsx_layout.h).
#pragma once
#include "fqsx_layout.h"

// ---------------------------------------------------------------------------------------
// workgroup-shared (LDS) state of one worker
#define FQSX_RR 6u
#define FQSX_SW 4u    // Hamming-1 sweeps a scout wave keeps in flight (one probe of each per lane)
#define FQSX_NSC 3u   // scout waves of a worker (paired-end kernels: 2): chunk number c of an epoch is made by scout c % nsc in ring slot
#define FQSX_SCR FQSX_NSC   // c % nsc of its own -- a scout wave owns one slot, so a restart needs no hand-shake between the scouts
#define FQSX_RQ 256u  // entries of the range-coder queue (power of two)
// One stage-P chunk: everything about positions i0..i0+n-1 of a read that does not depend on the adaptive models,
// computed one position per lane under the assumption "no k-mer correction since the k-mers stage P started from".
struct SpecBuf {
  u32 h_read, h_i0, h_n;       // scout chunks: read index within the launch, first position, positions
  u32 h_epoch;                 // ... the restart epoch of the k-mer state they were rolled from (see ScoutReq) ...
  u32 h_pub;                   // ... and, stored last (release), the chunk's sequence number within the epoch + 1
  u32 h_fix_lane, h_fix_end;   // scout_fix: lanes (h_fix_lane, h_fix_end) already assume the repair that fires at h_fix_lane (0xff: none)
  u32 h_pq_lo[2];              // list entries (b, s) below these were in the local tables when the chunk's probes started
  u32 h_np, h_nlp;             // probes issued (global, local) ...
  u64 h_ns, h_nls;             // ... and slots scanned, accounted when the chunk is used
  u64 sp_sdir[6][FQSX_SPEC];   // rolled k-mers after insert_zero: pm, sm, bm, pm_u, sm_u, bm_u
  u64 sp_src[6][FQSX_SPEC];
  u8 sp_scur[6][FQSX_SPEC];
  u64 sp_key[10][FQSX_SPEC];   // context keys of the position's symbol: 7 code levels (r_sym field left 0) or 10 letter levels; level-major,
                               // so that the lanes of a wave (one position each) touch consecutive 8-byte words: [position][level] rows of
                               // 80 bytes put 64 lanes on 16 LDS banks (4-way conflicts on every key store, copy and load)
  u8 sp_kind[FQSX_SPEC];       // how the position's symbol is coded: SK_* (set by stage P for settled positions, else by stage C)
  u64 sp_cq[FQSX_SPEC];        // SK_RANK_PENDING: the four counts, 16 bits each
  u8 sp_lvz[FQSX_SPEC];        // SK_RANK_PENDING: level | cor_zone << 4
  u8 sp_flag[FQSX_SPEC];       // 0 slow path, 1 b-mer hit (fast path), 3 slow path with known global b-mer miss
  u8 sp_rsym[FQSX_SPEC];       // rank of the read's symbol under those counts
  u8 sp_rep[FQSX_SPEC];        // fast path: symbol repair_kmers_existing substitutes, or 0xff
  u8 sp_nrun[FQSX_SPEC];       // N_run_len before the position
  // mailbox entries of the chunk's positions (stage Q appends them lane-parallel)
  u64 pv_b[FQSX_SPEC], pv_s[FQSX_SPEC], pv_pd[FQSX_SPEC], pv_pr[FQSX_SPEC];
  u8 pv_flag[FQSX_SPEC];       // PV_* bits
  // stage P, positions whose global b-mer probe missed: the rest of find_counts' cascade and the
  // Hamming-1 fall-back, resolved lane-parallel (valid while no pending local insert interferes)
  u32 sx_lb[FQSX_SPEC];        // local b-mer counts, 4 x 8 bit
  u64 sx_s[FQSX_SPEC];         // global s-mer counts, 4 x 16 bit
  u64 sx_ls[FQSX_SPEC];        // local s-mer counts, 4 x 16 bit
  u8 sx_flag[FQSX_SPEC];       // SX_* bits
  // scout chunks: Hamming-1 sweeps (find_counts_rough_b) probed ahead for up to FQSX_RR positions whose cascade came up empty
  u8 rr_idx[FQSX_SPEC];        // position -> sweep slot, 0xff = not probed ahead
  u32 rr_front;                // the sweeps of all positions below this one are finished (the chunk is published before them)
  u64 rr_res[FQSX_RR][64];     // per probe: the four sibling counts, 16 bits each
  u64 rr_hit[FQSX_RR];         // probes that found something
  u64 rr_ns[FQSX_RR];          // slots scanned
  // ... and for every other such position in compact form when at most 3 of the sweep's probes found something
  // (rr_idx 0xfe): which probes, and their counts in probe order
  u64 rc_hit[FQSX_SPEC];
  u64 rc_val[FQSX_SPEC][3];
  u32 rc_ns[FQSX_SPEC];
  // scout chunks: the global-table look-ups of the read's first positions, whose b-mer is still partial (find_counts
  // probes 4^m paddings of a partial k-mer): per trial the four sibling counts, in trial order
  u8 ep_off[FQSX_SPEC][2];     // position, table (0 global b, 1 global s) -> first entry of ep_res, 0xff = not probed ahead
  u64 ep_res[64];
  u8 ep_ns[64];                // slots scanned per trial
};
// What the read-head wave hands over per read: the head's symbols as finished coder triples, the p-mer list
// entries, the rolling k-mers after the prefix and the read's letter histogram.
#define FQSX_HD_RAW 56u
struct HeadRec {
  u32 idx;                     // read (index within the launch) the record describes; ~0 while it is being written
  u32 n_raw, same, n_run, n_p;
  u32 hist[4];
  u64 raw[FQSX_HD_RAW][2];     // freq | cum << 32, total
  u64 pmail[2];
  u64 kdir[3], krc[3];
  u32 kcur[3];
};
struct WgShared {
  HeadRec hd[FQSX_HD];         // (round 4: three -- with two, the wave that resolves the reads waited a quarter of its time for the first
                               // chunk of the next read: the read head could only start a read once the read before the previous one was
                               // finished, and the scouts' stage P of the read's first chunk starts behind the head)
  u32 hd_ready, hd_taken;      // read heads finished / consumed (free-running counts within the launch)
  u32 hd_early;                // read heads whose part for the scout waves (codes, k-mers after the prefix) is in place
  alignas(8) u32 mt[4][624];   // MT19937 states of cinc_b, cinc_s, cinc_lb, cinc_ls
  u32 mt_idx[4];
  alignas(8) u8 rd[FQSX_HD][FQSX_RD_LDS]; // 2-bit codes (0..4) of the current read (one buffer per head record: the head wave stages the next reads)
  u64 bk_key[256];             // probe batch: normalised k-mers
  u32 bk_res[256][4];          // probe batch: counts
  u8 bk_dir[256];              // probe batch: orientation
  union {   // the paired-end kernels run two scouts: their scratch lies over the third ring slot
    SpecBuf sb[1 + FQSX_SCR];  // [0] filled by the resolving wave itself, [1..] ring filled by the scout waves
    struct {
      SpecBuf sb_used_[FQSX_SCR];   // (= sb[0 .. 2])
      alignas(8) u8 r2c[FQSX_RD_LDS];     // paired-end: codes of the second mate
      u64 pe_cand[512];        // paired-end: candidate partner b-mers (value | count << 2k)
      u64 pe_top[64];
      u64 pe_bk[3][64];        // paired-end insert batch (key, value, weight)
    };
  };
  u32 sc_taken;                // chunks of the current epoch the resolving wave has released; chunk c lives in sb[1 + c % FQSX_SCR]
  u32 sc_hd_taken[FQSX_NSC];   // read heads each scout wave is done with
  // Restart of the scout waves: the resolving wave posts the exact state to go on from (sc_req, then sc_req_seq = the
  // new epoch) -- after a k-mer correction the state after the chunk it has just committed, after a read it finished
  // on its own the head of the next read.  Every scout wave drops what it is doing, acknowledges (sc_ack) and, once
  // all have, rolls on from the posted state; chunk numbers start again at 0 in every epoch.
  u32 sc_req_seq;
  u32 sc_ack[FQSX_NSC];
  u32 sc_dead;                 // the scout waves have given up for this launch (one found itself behind the resolving wave)
  struct ScoutReq {
    u32 read, i0, cor_pos, n_run;
    u32 flags, size;                // SCQ_*; request mode: length of the sequence being coded
    u32 slot_base, pad_;            // chunk c of the new epoch lives in ring slot (slot_base + c) % nsc
    u64 p;                          // request mode: its bases in HBM (reads longer than the LDS staging buffer)
    u64 kdir[6], krc[6];
    u32 kcur[6];
    u64 s_let[4];
  } sc_req;
  u64 lev_tmp[10];             // level keys of a position coded outside the fast path
  // fast_run: per-position results of the lane-parallel context search / model stage
  u32 fr_idx[64], fr_c0[64], fr_thr[64], fr_vis[64];   // final slot, its counter, threshold to re-validate (or ~0), slots visited
  u64 fr_q1[64], fr_q2[64], fr_q3[64];                 // the final slot's counter|tag|total and statistics at run start
  u64 fr_same[64];                                     // lanes of the run that end in the same slot
#ifdef FQSX_EMU
  u32 fr_f[64], fr_c[64], fr_t[64];                    // range-coder triple of the position (GPU: lane registers)
#endif
  u8 fr_lvl[64], fr_bad[64];
  u64 pq_key[2][FQSX_PQ];      // LDS mirror of the most recent b / s list entries (ring indexed by list position)
  u64 ib_pos[64];              // insert_batch: target slot per lane
  u64 ib_hash[256];            // ... and the lanes per slot bucket
  alignas(8) u32 qm_bits[4][256];         // quiet_miss_mask: 8192-bit sets (two hash bits per entry) of the sibling groups of recent list entries (b dir, b rc, s dir, s rc)
  // hand-off words of the local-table inserter wave: list entries published / applied per kind (b, s); quit
  u32 lq_target[2], lq_done[2], lq_quit;
  // coding queue: every symbol of the worker's stream in stream order, as the context keys of a rank-/letter-coded
  // position (SK_RANK / SK_LETTER) or as a finished (freq, cum, total) triple of a small direct-indexed model
  // (SK_RAW: key[0] = freq | cum << 32, key[1] = total).  Filled by the wave that resolves the reads, drained by
  // the coder (the same wave in the single-wave build, the workgroup's second wave in the encode kernel).
  u64 cq_key[10][FQSX_CQ];     // (level-major like sp_key: the models wave reads one entry per lane)
  u8 cq_kind[FQSX_CQ];         // SK_* | SK_RESET
  u8 cq_rsym[FQSX_CQ];
  u32 cq_tail, cq_head, cq_done;   // entries published / consumed (free-running); producer finished
  // range-coder queue (six-wave kernel): the finished (freq, cum, total, reciprocal) of every symbol in stream order,
  // from the wave that runs the context models to the wave that runs the range coder
  u32 rq_f[FQSX_RQ], rq_c[FQSX_RQ], rq_t[FQSX_RQ];
  u64 rq_m[FQSX_RQ];
  u32 rq_tail, rq_head, rq_done;
  u32 wg_stop;                 // the block's queue has been stopped (device error / posted growth): read once per workgroup, see wg_handoff_init
};
#ifndef FQSX_EMU
static_assert(sizeof(WgShared) <= 160u * 1024u, "WgShared must fit the 160 KB of LDS of a gfx950 CU");
#endif
// LDS of the insert-phase kernels (k_insert_phase, k_pe_insert): one RNG stream at a time and the batch scratch
struct InsShared {
  u32 mt[4][624];
  u32 mt_idx[4];
  u32 pf_done, pf_on;   // batches the inserting wave has applied / a prefetching wave runs beside it (k_insert_phase)
  u64 ib_pos[64];
  u64 ib_hash[256];
  u64 bk_key[64];
  u64 pe_bk[3][64];
};
// Arguments of the encode / decode kernels: one struct, so that a role (FQ_ROLE) finds them in the kernel-argument
// segment (scalar loads) instead of having them passed through vector registers.
struct EncArgs {
  DevCfg cfg;
  u32 n_reads, S, seg, pad;
};
// A role receives the address of the kernel arguments as an ordinary (vector-register) argument -- the kernarg segment
// pointer itself is only defined in the kernel function -- and makes it wave-uniform again, so that the arguments are
// read with scalar loads from the constant address space.
#ifndef FQSX_EMU
__shared__ WgShared fq_wg_lds;   // the worker's LDS block (allocated only for kernels that reach it)
FQ_DEV WgShared *fq_wg() { return &fq_wg_lds; }
typedef const __attribute__((address_space(4))) EncArgs *FqArgsP;
FQ_DEV FqArgsP fq_kernarg() { return (FqArgsP)__builtin_amdgcn_kernarg_segment_ptr(); }
FQ_DEV const EncArgs *fq_args(FqArgsP a) {
  const unsigned long v = (unsigned long)a;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return (const EncArgs *)(FqArgsP)(((unsigned long)hi << 32) | lo);
}
#else
typedef const EncArgs *FqArgsP;
FQ_DEV WgShared *fq_wg() { static thread_local WgShared s; return &s; }
FQ_DEV const EncArgs *fq_args(FqArgsP a) { return a; }
#endif
enum { SX_VALID = 1, SX_LB = 2, SX_S = 4, SX_LS = 8, SX_UNC = 16 /* the uncorrected b-mer is in the global table (counts in sx_s) */,
       SX_HITS = SX_LB | SX_S | SX_LS | SX_UNC };
enum { SCQ_FROM_HEAD = 1 /* base state = the read's head record */, SCQ_REVERSED = 2 /* positions count from the end (dna.cpp:750-752) */ };
enum { SK_NONE = 0, SK_RANK = 1, SK_LETTER = 2, SK_RANK_PENDING = 3, SK_LETTER_PENDING = 4, SK_RAW = 5, SK_KIND_MASK = 7,
       SK_RESET = 8 /* the r_sym history restarts at this entry (first symbol of a compress_suffix call, dna.cpp:676) */ };
enum { PV_B = 1, PV_S = 2, PV_P = 4, PV_PHID = 8, PV_PCAND = 16 };

struct C4 { u32 c[4]; };
struct Kmer { u64 dir, rc; u32 cur; };
struct Cinc { u32 thr, mult, maxv; };
struct Enc { u64 low, range, len, cap, acc; u8 *out; };   // acc: bytes of the 8-byte output word being filled

struct Wk {
  const DevCfg *cfg;
  WgShared *sm;
  WState *ws;
  u32 tid;
  u32 mode;                             // dna_mode, a compile-time constant of the kernel (template argument of the bodies)
  Enc enc;
  Kmer pm, sm_, bm, pm_u, sm_u, bm_u;   // corrected and uncorrected rolling k-mers (dna.h:160-168)
  u64 ctx_letters;
  u32 cor_pos, N_run;
  u64 s_let[4];
  double avg_code, avg_letters;
  u64 hidden;
  bool repm_gate;                       // siv avg_filling_factor() >= 7 (constant within a segment)
  u32 mn[3];                            // entries appended to this worker's p/s/b mailbox lists
  u32 pe_n;                             // paired-end triples pushed in this launch
  const u8 *din;                        // decoder input stream of this worker
  u64 din_len, din_pos, din_buffer;
  u32 la[3];                            // list entries already applied to the local tables (b, s)
  u32 pq_lo[2];                         // list entries (b, s) below this index were in the local tables when stage P last probed them
  bool lqh;                             // a third wave of the workgroup applies the local inserts (else: inline, on demand)
  u32 lq_pub[2];                        // entries already published to that wave
  u8 *rdp;                              // LDS staging buffer of the current read's codes
  SpecBuf *sb;                          // stage-P chunk in use
  bool scout;                           // stage P of clean chunks comes from the scout wave
  bool sc_abandoned;                    // ... but no longer for the current read (the wave went on without them)
  u32 sc_read;                          // index of the current read within the launch
  u32 sc_epoch;                         // restart epoch this wave is in (resolving wave: the one it expects chunks of)
  bool sc_poll;                         // scout wave: stage P gives up as soon as a restart request is pending
  u32 nsc;                              // scout waves (= ring slots) of this kernel
  bool sc_reqmode;                      // no read-head wave: the scouts serve one request per compress_suffix call
  const u8 *rq_p;                       // the sequence the current compress_suffix call codes (for the requests)
  u32 rq_size;
  bool rq_rev;
  u32 sc_taken;                         // scout chunks of the current epoch released so far
  u32 sc_base;                          // ring slot of the epoch's first chunk
  bool rq_early;                        // request mode: the coming suffix() call's request is posted already
  bool rq_noscout;                      // request mode: the coming suffix() call goes without the scout waves (its sequence is a
                                        // scratch line this wave has just written: long reverse-complement part of a second mate)
  HeadRec *rec;                         // read-head wave: where the head's output goes (null: code / push directly)
  u32 rec_idx;                          // ... and the read's index within the launch
  bool piped;                           // this wave only resolves; a second wave of the workgroup drains the coding queue
  u32 cq_tail, cq_head;                 // this wave's copy of its own queue index
  bool rcq;                             // coding steps go to the range-coder queue (another wave runs the coder proper)
  u32 rq_tail;                          // ... this wave's copy of its index
  u64 c_r_sym;                          // coder: ctx_r_sym, the last 8 rank-0 flags (dna.cpp:664-671)
  u64 st[ST_N];
  u64 tm[FQSX_TM_SLOTS];
  u32 err;
};
// role time stamps of one launch (timing builds): 0 resolve start, 1 head end, 2 resolve end, 3 coder end, 4 scout end,
// 5 inserter end, 6 resolve: reads done (before the final flush of the local inserts)
#define FQSX_TRACE_LAUNCHES 4096u
#define FQSX_TRACE_W 32u   /* 8 clock stamps + 24 per-launch counters / section times of the resolving wave */
#ifdef FQSX_TIMING
#define TM_STAMP(cfg, tid, launch, slot) do { if ((cfg).trace && (launch) < FQSX_TRACE_LAUNCHES && FQ_LANE == 0) (cfg).trace[((u64)(launch) * (cfg).T + (tid)) * FQSX_TRACE_W + (slot)] = fq_clock(); } while (0)
#define TM_TRACE_VAL(cfg, tid, launch, slot, v) do { if ((cfg).trace && (launch) < FQSX_TRACE_LAUNCHES && FQ_LANE == 0) (cfg).trace[((u64)(launch) * (cfg).T + (tid)) * FQSX_TRACE_W + (slot)] = (v); } while (0)
#else
#define TM_STAMP(cfg, tid, launch, slot) ((void)0)
#define TM_TRACE_VAL(cfg, tid, launch, slot, v) ((void)0)
#endif
#ifdef FQSX_TIMING
#define TM_BEGIN(v) u64 v = fq_clock()
#define TM_END(w, slot, v) (w).tm[slot] += fq_clock() - (v)
#define TM_COUNT(w, slot) (w).tm[slot] += 1
#if defined(FQSX_TIMING_PE)   /* ... or the paired-end steps of the resolving wave (compress_pair) */
#define TM_END_SP(w, slot, v) ((void)0)
#define TM_END_MD(w, slot, v) ((void)0)
#define TM_END_PE(w, slot, v) TM_END(w, slot, v)
#elif defined(FQSX_TIMING_MODELS)   /* slots 38, 39, 46, 47 time the tail of code_run instead of the sections of stage P */
#define TM_END_SP(w, slot, v) ((void)0)
#define TM_END_MD(w, slot, v) TM_END(w, slot, v)
#else
#define TM_END_SP(w, slot, v) TM_END(w, slot, v)
#define TM_END_MD(w, slot, v) ((void)0)
#endif
#ifndef TM_END_PE
#define TM_END_PE(w, slot, v) ((void)0)
#endif
#else
#define TM_BEGIN(v) ((void)0)
#define TM_END(w, slot, v) ((void)0)
#define TM_COUNT(w, slot) ((void)0)
#define TM_END_SP(w, slot, v) ((void)0)
#define TM_END_MD(w, slot, v) ((void)0)
#define TM_END_PE(w, slot, v) ((void)0)
#endif

#define CINC_B (Cinc{7u, 2u, 63u})            /* dna.cpp:162,164 */
#define CINC_S (Cinc{2047u, 1u, 4095u})       /* dna.cpp:163,165 */

FQ_DEV u64 murmur64(u64 h) {  // ht_kmer.h:123-127, context_hm.h:81-85
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}
FQ_DEV u32 dna_code(u8 c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }  // dna.cpp:19-24

FQ_DEV void c4_zero(C4 &c) { c.c[0] = c.c[1] = c.c[2] = c.c[3] = 0; }
FQ_DEV bool c4_any(const C4 &c) { return (c.c[0] | c.c[1] | c.c[2] | c.c[3]) != 0; }
FQ_DEV void c4_add(C4 &c, u32 sym, u32 v) {
  c.c[0] += sym == 0 ? v : 0;
  c.c[1] += sym == 1 ? v : 0;
  c.c[2] += sym == 2 ? v : 0;
  c.c[3] += sym == 3 ? v : 0;
}
FQ_DEV u32 c4_get(const C4 &c, u32 i) { return i == 0 ? c.c[0] : i == 1 ? c.c[1] : i == 2 ? c.c[2] : c.c[3]; }
FQ_DEV u64 sl_get(const u64 *s, u32 i) { return i == 0 ? s[0] : i == 1 ? s[1] : i == 2 ? s[2] : s[3]; }

// ---------------------------------------------------------------------------------------
// std::mt19937 in LDS; regeneration of the 624-word state is wave-parallel
FQ_DEV void mt_twist(u32 *s) {
  for (u32 base = 0; base < 624; base += FQ_WAVE) {
    u32 i = base + FQ_LANE, v = 0;
    if (i < 624) {
      u32 nxt = s[i == 623 ? 0 : i + 1];
      u32 far = s[i < 227 ? i + 397 : i - 227];
      u32 y = (s[i] & 0x80000000u) | (nxt & 0x7fffffffu);
      v = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    FQ_SYNC();
    if (i < 624) s[i] = v;
    FQ_SYNC();
  }
}
// the four 624-word states between HBM and LDS (launch prologue / epilogue): 1248 eight-byte words, every lane's loads
// issued back to back before the first store (a lone wave: what counts is how many requests are in flight)
FQ_DEV void mt_copy(u64 *dst, const u64 *src) {
#if FQ_WAVE > 1
  u64 v[20];
#pragma unroll
  for (u32 k = 0; k < 20; ++k) { const u32 i = k * FQ_WAVE + FQ_LANE; v[k] = i < 1248u ? src[i] : 0ull; }
#pragma unroll
  for (u32 k = 0; k < 20; ++k) { const u32 i = k * FQ_WAVE + FQ_LANE; if (i < 1248u) dst[i] = v[k]; }
#else
  for (u32 i = 0; i < 1248u; ++i) dst[i] = src[i];
#endif
}
FQ_DEV u32 mt_temper(u32 y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
FQ_DEV u32 mt_next(WgShared *sm, u32 g) {
  u32 idx = sm->mt_idx[g];
  if (idx >= 624) {
    mt_twist(sm->mt[g]);
    idx = 0;
  }
  u32 y = sm->mt[g][idx];
  FQ_SYNC();
  if (FQ_LANE == 0) sm->mt_idx[g] = idx + 1;
  FQ_SYNC();
  return mt_temper(y);
}

// ---------------------------------------------------------------------------------------
// CCounterIncrementer (utils.h:256-335) with the mapping table in closed form
FQ_DEV u32 cinc_map(const Cinc &c, u32 i) {  // v_mapping[i], utils.h:303-311
  if (i <= c.thr) return i;
  if (i > c.maxv) i = c.maxv;
  u32 n = i - c.thr;
  return c.thr + c.mult * (n * (n + 1) / 2);
}
FQ_DEV u32 cinc_decode(const Cinc &c, u32 v) {  // utils.h:264-270
  if (v <= c.thr) return v;
  return (cinc_map(c, v) + cinc_map(c, v + 1)) / 2;
}
FQ_DEV u32 cinc_encode(WgShared *sm, u32 g, const Cinc &c, u32 real) {  // utils.h:272-290
  if (real <= c.thr) return real;
  // last position whose mapped value is <= real (upper_bound - 1): largest n with mult*n(n+1)/2 <= real-thr,
  // from the float square root plus an exact correction
  const u32 R = real - c.thr;
  u32 n = (u32)((sqrtf(1.0f + 8.0f * (float)R / (float)c.mult) - 1.0f) * 0.5f);
  while (c.mult * (n * (n + 1) / 2) > R) --n;
  while (c.mult * ((n + 1) * (n + 2) / 2) <= R) ++n;
  u32 pos = c.thr + n;
  if (pos > c.maxv) pos = c.maxv;
  if (pos >= c.maxv) return c.maxv;
  u32 mp = cinc_map(c, pos);
  u32 rest = real - mp;
  if (mt_next(sm, g) % (cinc_map(c, pos + 1) - mp) < rest) ++pos;
  return pos;
}
FQ_DEV u32 cinc_merge(WgShared *sm, u32 g, const Cinc &c, u32 a, u32 b) {  // Increment(a,b), utils.h:327-333
  return cinc_encode(sm, g, c, cinc_decode(c, a) + cinc_decode(c, b));
}
FQ_DEV u32 cinc_inc1(WgShared *sm, u32 g, const Cinc &c, u32 v) {  // Increment(c), utils.h:314-325
  if (v <= c.thr) return v + 1;
  return (mt_next(sm, g) % (c.mult * (v - c.thr)) == 0) ? v + 1 : v;
}

// ---------------------------------------------------------------------------------------
// rolling canonical k-mer (kmer.h), symbols left-aligned
FQ_DEV void km_reset(Kmer &k) { k.dir = k.rc = 0; k.cur = 0; }
FQ_DEV void km_insert(Kmer &k, const KGeom &g, u64 sym) {  // kmer.h:80-96
  k.rc >>= 2;
  k.rc += (3 - sym) << 62;
  k.rc &= g.mask;
  if (k.cur == g.k) {
    k.dir <<= 2;
    k.dir += sym << g.shift;
  } else {
    ++k.cur;
    k.dir += sym << (64 - 2 * k.cur);
  }
}
FQ_DEV void km_insert_zero(Kmer &k, const KGeom &g) {  // kmer.h:99-115
  k.rc >>= 2;
  k.rc += 3ull << 62;
  k.rc &= g.mask;
  if (k.cur == g.k) k.dir <<= 2; else ++k.cur;
}
FQ_DEV void km_replace(Kmer &k, u64 sym, u32 pos) {  // kmer.h:153-160,172-178
  u32 sh = 62 - 2 * pos;
  k.dir = (k.dir & ~(3ull << sh)) + (sym << sh);
  sh = 64 - 2 * k.cur + 2 * pos;
  k.rc = (k.rc & ~(3ull << sh)) + ((3 - sym) << sh);
}
FQ_DEV void km_replace_last(Kmer &k, u64 sym) {  // kmer.h:163-169,181-185
  u32 sh = 64 - 2 * k.cur;
  k.dir = (k.dir & ~(3ull << sh)) + (sym << sh);
  k.rc = ((k.rc << 2) >> 2) + ((3 - sym) << 62);
}
// k-mer `b` after inserting j further symbols (kmer.h:80-96 applied j times).  fw / rv hold the last
// L = min(j,27) of them: fw packed oldest-first, rv = their complements packed newest-first.
FQ_DEV Kmer km_roll(const Kmer &b, const KGeom &g, u32 j, u64 fw, u64 rv, u32 L) {
  if (j == 0) return b;
  Kmer r;
  const u32 k = g.k;
  u32 tot = b.cur + j;
  r.cur = tot < k ? tot : k;
  if (j >= k) {  // the k-mer consists of new symbols only (L >= k here)
    r.dir = (fw & ((1ull << (2 * k)) - 1ull)) << g.shift;
    r.rc = (rv >> (2 * (L - k))) << g.shift;   // newest k symbols sit at the top of rv
  } else {       // L == j < k: old symbols survive
    u64 old = b.cur ? b.dir >> (64 - 2 * b.cur) : 0;                    // right-aligned old symbols
    u32 keep = r.cur - j;                                               // old symbols still inside
    old &= (1ull << (2 * keep)) - 1ull;
    r.dir = ((old << (2 * j)) | fw) << (64 - 2 * r.cur);
    r.rc = ((rv << (64 - 2 * j)) | (b.rc >> (2 * j))) & g.mask;
  }
  return r;
}
#if FQ_WAVE > 1
// the 32 two-bit symbols of x in reverse order
FQ_DEV u64 pairrev64(u64 x) {
  const u64 r = ((u64)__builtin_bitreverse32((u32)x) << 32) | __builtin_bitreverse32((u32)(x >> 32));
  return ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
}
#endif
FQ_DEV bool km_norm_dir(const Kmer &k, const KGeom &g) { return (k.dir & g.kernel_mask) < (k.rc & g.kernel_mask); }
FQ_DEV u64 km_norm(const Kmer &k, const KGeom &g) { return km_norm_dir(k, g) ? k.dir : k.rc; }
FQ_DEV u64 km_aligned_dir(const Kmer &k) { return k.cur ? k.dir >> (64 - 2 * k.cur) : 0; }  // kmer.h:398 (quirk 18)
FQ_DEV u64 km_aligned_rc(const Kmer &k) { return k.cur ? k.rc >> (64 - 2 * k.cur) : 0; }
FQ_DEV u64 km_symbol(const Kmer &k, u32 pos) { return (k.dir >> (62 - 2 * pos)) & 3; }
FQ_DEV bool km_full(const Kmer &k, const KGeom &g) { return k.cur == g.k; }
FQ_DEV bool km_almost_full(const Kmer &k, const KGeom &g, u32 margin) { return k.cur + margin >= g.k; }

// ---------------------------------------------------------------------------------------
// k-mer tables
FQ_DEV u32 mod_T(const DevCfg *cfg, u32 x) {  // x % T for x < 2^14, T <= 255 (exact: x*T < 2^32)
  if (cfg->T_pow2) return x & (cfg->T - 1);    // (wave-uniform branch: the reference CLI's usual -t 8 / 16 / 32 / 64)
  u32 q = (u32)(((u64)x * cfg->T_magic) >> 32);
  return x - q * cfg->T;
}
FQ_DEV u32 sb_owner(const DevCfg *cfg, u64 kmer_norm) { return mod_T(cfg, (u32)((kmer_norm >> 46) & 0x3fffull)); }  // dna.cpp:825
FQ_DEV u32 p_owner(const DevCfg *cfg, u64 idx) { return mod_T(cfg, (u32)(idx >> cfg->pmer_mod_shift)); }           // dna.cpp:658

// Home slot of a k-mer = hash of its kernel (symbols 2..k-3, <= 46 bits).  The layout is this repo's own (results do
// not depend on it), so the hash is picked for the GPU: a 32-bit finaliser (two 32-bit multiplies) instead of the
// 64-bit murmur finaliser (two 64-bit multiplies = a dozen quarter-rate operations per probe).
FQ_DEV u32 kmer_mix(u64 kern) {
  u32 x = (u32)kern ^ ((u32)(kern >> 32) * 0x9E3779B1u);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
FQ_DEV u32 kmer_mix2(u32 h) {   // the second bucket of a k-mer: an independent hash of the first
  h *= 0x9E3779B1u;
  h ^= h >> 15; h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  return h;
}
struct TabHome { u32 a, b; };
FQ_DEV TabHome tab_home(const KTab &t, u64 v) {
  const u64 kern = (v >> 4) & ((1ull << (2 * t.k - 8)) - 1ull);
  const u32 h = kmer_mix(kern);
  TabHome r;
  r.a = (u32)(((u64)h * t.nb) >> 32);              // (fastrange: any number of buckets)
  if (t.two) {
    r.b = (u32)(((u64)kmer_mix2(h) * t.nb) >> 32);
    if (r.b == r.a) r.b = (u64)r.a + 1 == t.nb ? 0u : r.a + 1u;
  } else
    r.b = (u64)r.a + 1 == t.nb ? 0u : r.a + 1u;    // one sequence: a, a + 1, a + 2, ...
  return r;
}
// the bucket after p (p != a) in the overflow chain b, b + 1, b + 2, ... (a is never visited twice)
FQ_DEV u32 tab_next(const KTab &t, u32 a, u32 p) {
  u32 q = (u64)p + 1 == t.nb ? 0u : p + 1u;
  if (q == a) q = (u64)q + 1 == t.nb ? 0u : q + 1u;
  return q;
}
struct alignas(16) Slot2 { u64 x, y; };
FQ_DEV void tab_load_bucket(const u64 *s, u32 bk, u64 it[FQSX_BKT]) {   // one 32-byte access (two 16-byte halves)
  const Slot2 *p = (const Slot2 *)__builtin_assume_aligned(s + (u64)bk * FQSX_BKT, 32);
  const Slot2 lo = p[0], hi = p[1];
  it[0] = lo.x; it[1] = lo.y; it[2] = hi.x; it[3] = hi.y;
}
// The sibling group's counts out of one bucket (_update_counts_full, ht_kmer.h:205-263).  A wave issues one instruction every
// four to five cycles, and a look-up examines eight slots whichever of them are occupied (64 lanes, 64 different buckets), so
// the per-slot work is kept to a dozen instructions: one masked 64-bit compare decides "same sibling group" for either
// orientation, the matching slot's count is added into a packed accumulator (4 x 16 bits: a sibling is stored once and a count
// has at most 12 bits), free slots need no test of their own (their count bits are zero).
struct TabQ { u64 mask, want; u32 sh, flip, cm; };
FQ_DEV TabQ tab_query(const KTab &t, u64 kmer_norm, bool is_dir) {
  const u32 k2 = 2 * t.k;
  const u64 v = kmer_norm >> (64 - k2);
  TabQ q;
  q.cm = (1u << t.cbits) - 1u;
  if (is_dir) {   // siblings differ in the LAST symbol: everything above it has to match
    q.mask = ~0ull << (t.cbits + 2);
    q.want = (v >> 2) << (t.cbits + 2);
    q.sh = t.cbits; q.flip = 0;
  } else {        // ... in the FIRST symbol (of the reverse complement: the letter is complemented)
    const u64 lowmask = (1ull << (k2 - 2)) - 1ull;
    q.mask = lowmask << t.cbits;
    q.want = (v & lowmask) << t.cbits;
    q.sh = t.cbits + k2 - 2; q.flip = 3;
  }
  return q;
}
// returns the occupied slots of the bucket (it fills from slot 0 up: = the index of its first free slot; FQSX_BKT: full)
FQ_DEV u32 tab_bucket_acc(const TabQ &q, const u64 it[FQSX_BKT], u64 &acc) {
  u32 occ = 0;
#pragma unroll
  for (u32 j = 0; j < FQSX_BKT; ++j) {
    occ += it[j] != 0 ? 1u : 0u;
    const bool m = (it[j] & q.mask) == q.want;
    const u32 sym = ((u32)(it[j] >> q.sh) & 3u) ^ q.flip;
    const u64 add = (u64)((u32)it[j] & q.cm) << (16 * sym);
    acc += m ? add : 0ull;
  }
  return occ;
}
FQ_DEV void tab_unpack(u64 acc, C4 &c) {
  c.c[0] += (u32)(acc & 0xffff); c.c[1] += (u32)((acc >> 16) & 0xffff); c.c[2] += (u32)((acc >> 32) & 0xffff); c.c[3] += (u32)(acc >> 48);
}
// One look-up in two parts, so that a caller can have the first round trips of several independent look-ups in flight at
// once: tab_first_* issues the loads, tab_rest consumes them and walks on in the rare case that the buckets are full.
// Global tables (two-choice): both buckets are fetched together.  Local tables (one sequence, never more than a quarter full): one.
struct TabG { const u64 *s; u32 a, b; u64 ia[FQSX_BKT], ib[FQSX_BKT]; };
struct TabL { const u64 *s; u32 a; u64 it[FQSX_BKT]; };
FQ_DEV TabG tab_first_g(const KTab &t, u32 sub, u64 kmer_norm) {
  TabG r;
  r.s = t.slots + (u64)sub * t.stride;
  const TabHome h = tab_home(t, kmer_norm >> (64 - 2 * t.k));
  r.a = h.a; r.b = h.b;
  tab_load_bucket(r.s, r.a, r.ia);
  tab_load_bucket(r.s, r.b, r.ib);
  return r;
}
FQ_DEV TabL tab_first_l(const KTab &t, u32 sub, u64 kmer_norm) {
  TabL r;
  r.s = t.slots + (u64)sub * t.stride;
  r.a = tab_home(t, kmer_norm >> (64 - 2 * t.k)).a;
  tab_load_bucket(r.s, r.a, r.it);
  return r;
}
FQ_DEV void tab_rest(const KTab &t, const TabG &f, u64 kmer_norm, bool is_dir, C4 &c, u64 &nslots) {
  const TabQ q = tab_query(t, kmer_norm, is_dir);
  u64 acc = 0;
  const u32 oa = tab_bucket_acc(q, f.ia, acc), ob = tab_bucket_acc(q, f.ib, acc);
  u32 ns = (oa < FQSX_BKT ? oa + 1 : FQSX_BKT) + (ob < FQSX_BKT ? ob + 1 : FQSX_BKT);   // (slots up to and including each bucket's first free one)
  if (oa == FQSX_BKT && ob == FQSX_BKT) {   // both full: the overflow chain
    u32 p = f.b;
    for (u64 n = 0; n <= t.nb; ++n) {
      p = tab_next(t, f.a, p);
      u64 it[FQSX_BKT];
      tab_load_bucket(f.s, p, it);
      const u32 oc = tab_bucket_acc(q, it, acc);
      ns += oc < FQSX_BKT ? oc + 1 : FQSX_BKT;
      if (oc != FQSX_BKT) break;
    }
  }
  tab_unpack(acc, c);
  nslots += ns;
}
FQ_DEV void tab_rest(const KTab &t, const TabL &f, u64 kmer_norm, bool is_dir, C4 &c, u64 &nslots) {
  const TabQ q = tab_query(t, kmer_norm, is_dir);
  u64 acc = 0;
  u32 oc = tab_bucket_acc(q, f.it, acc);
  u32 ns = oc < FQSX_BKT ? oc + 1 : FQSX_BKT;
  if (oc == FQSX_BKT) {
    u32 p = f.a;
    for (u64 n = 0; n <= t.nb; ++n) {
      p = (u64)p + 1 == t.nb ? 0u : p + 1u;
      u64 it[FQSX_BKT];
      tab_load_bucket(f.s, p, it);
      oc = tab_bucket_acc(q, it, acc);
      ns += oc < FQSX_BKT ? oc + 1 : FQSX_BKT;
      if (oc != FQSX_BKT) break;
    }
  }
  tab_unpack(acc, c);
  nslots += ns;
}
FQ_DEV void tab_scan(const KTab &t, u32 sub, u64 kmer_norm, bool is_dir, C4 &c, u64 &nslots) {
  if (t.two) { const TabG f = tab_first_g(t, sub, kmer_norm); tab_rest(t, f, kmer_norm, is_dir, c, nslots); }
  else { const TabL f = tab_first_l(t, sub, kmer_norm); tab_rest(t, f, kmer_norm, is_dir, c, nslots); }
}
// key v in a bucket: its slot (found_j), and the bucket's occupied slots (= its first free slot; FQSX_BKT: full)
FQ_DEV void tab_bucket_find(const KTab &t, const u64 it[FQSX_BKT], u64 v, u32 &found_j, u32 &occ, u64 &nslots) {
  const u64 km = ~0ull << t.cbits, kv = v << t.cbits;
  found_j = FQSX_BKT; occ = 0;
#pragma unroll
  for (u32 j = 0; j < FQSX_BKT; ++j) {
    occ += it[j] != 0 ? 1u : 0u;
    if ((it[j] & km) == kv && it[j] != 0) found_j = j;   // (a key is stored once)
  }
  nslots += found_j != FQSX_BKT ? found_j + 1 : occ < FQSX_BKT ? occ + 1 : FQSX_BKT;
}
FQ_DEV u64 tab_pick(const u64 it[FQSX_BKT], u32 j) {   // it[j] without dynamic indexing (the array lives in registers)
  u64 r = it[0];
#pragma unroll
  for (u32 x = 1; x < FQSX_BKT; ++x) r = j == x ? it[x] : r;
  return r;
}
// Where key v is, or where it goes: its own slot, else -- two-choice tables -- the first free slot of the emptier of its two
// buckets (ties: the first), else the first free slot along the overflow chain.  pos = slot index relative to s; item = the
// slot's content (0: free, the key is not in the table).
// (A function of its own, everything in and out BY VALUE: inlined at its dozen call sites the decode kernels took ten minutes to
// compile, and the encode / insert kernels ran 2 % slower; a reference parameter of a real function is a scratch round trip.)
struct TabLoc { u64 pos, item; u32 ns; };
FQ_DEVN TabLoc tab_locate(const KTab t, const u64 *s, u64 v) {
  TabLoc R;
  R.pos = ~0ull; R.item = 0; R.ns = 0;
  u64 nsl = 0;
  const TabHome h = tab_home(t, v);
  u64 ia[FQSX_BKT];
  tab_load_bucket(s, h.a, ia);
  u32 fa, oa;
  bool chain = false;
  if (t.two) {
    u64 ib[FQSX_BKT];
    tab_load_bucket(s, h.b, ib);
    u32 fb, ob;
    tab_bucket_find(t, ia, v, fa, oa, nsl);
    tab_bucket_find(t, ib, v, fb, ob, nsl);
    if (fa != FQSX_BKT) { R.item = tab_pick(ia, fa); R.pos = (u64)h.a * FQSX_BKT + fa; }
    else if (fb != FQSX_BKT) { R.item = tab_pick(ib, fb); R.pos = (u64)h.b * FQSX_BKT + fb; }
    else if (oa != FQSX_BKT || ob != FQSX_BKT) R.pos = oa <= ob ? (u64)h.a * FQSX_BKT + oa : (u64)h.b * FQSX_BKT + ob;   // (oa <= ob and oa full cannot both hold here)
    else chain = true;
  } else {
    tab_bucket_find(t, ia, v, fa, oa, nsl);
    if (fa != FQSX_BKT) { R.item = tab_pick(ia, fa); R.pos = (u64)h.a * FQSX_BKT + fa; }
    else if (oa != FQSX_BKT) R.pos = (u64)h.a * FQSX_BKT + oa;
    else chain = true;
  }
  if (chain) {
    u32 p = t.two ? h.b : h.a;   // the chain behind the full bucket(s)
    for (u64 n = 0; n <= t.nb; ++n) {
      p = t.two ? tab_next(t, h.a, p) : ((u64)p + 1 == t.nb ? 0u : p + 1u);
      u64 it[FQSX_BKT];
      tab_load_bucket(s, p, it);
      u32 fj, oc;
      tab_bucket_find(t, it, v, fj, oc, nsl);
      if (fj != FQSX_BKT) { R.item = tab_pick(it, fj); R.pos = (u64)p * FQSX_BKT + fj; break; }
      if (oc != FQSX_BKT) { R.pos = (u64)p * FQSX_BKT + oc; break; }
    }
  }
  R.ns = (u32)nsl;
  return R;   // (pos == ~0: cannot happen, the tables are never full)
}
// the same for a one-sequence (local) table, inlined: the inserter wave of a worker runs it for every batch of local inserts, and
// the wave that resolves the reads waits for that wave whenever a look-up could be changed by an insert still on its way
FQ_DEV TabLoc tab_locate1(const KTab &t, const u64 *s, u64 v) {
  TabLoc R;
  R.pos = ~0ull; R.item = 0; R.ns = 0;
  u64 nsl = 0;
  u32 p = tab_home(t, v).a;
  for (u64 n = 0; n <= t.nb; ++n) {
    u64 it[FQSX_BKT];
    tab_load_bucket(s, p, it);
    u32 fj, oc;
    tab_bucket_find(t, it, v, fj, oc, nsl);
    if (fj != FQSX_BKT) { R.item = tab_pick(it, fj); R.pos = (u64)p * FQSX_BKT + fj; break; }
    if (oc != FQSX_BKT) { R.pos = (u64)p * FQSX_BKT + oc; break; }
    p = (u64)p + 1 == t.nb ? 0u : p + 1u;
  }
  R.ns = (u32)nsl;
  return R;
}
// The same among concurrent writers (growth re-inserts, replica updates): the slot that holds key v afterwards -- its own, or a
// free one this thread has claimed with `item` (claimed = true).  A slot another thread takes first counts as occupied and the
// search goes on; a key ends up in the overflow chain only after both its buckets have been seen full, which they then stay.
FQ_DEV u64 *tab_find_or_claim(const KTab &t, u64 *s, u64 v, u64 item, bool &claimed) {
  const TabHome h = tab_home(t, v);
  claimed = false;
  u64 ns_ = 0;
  for (u32 attempt = 0; attempt < 64; ++attempt) {
    u64 ia[FQSX_BKT], ib[FQSX_BKT];
    for (u32 j = 0; j < FQSX_BKT; ++j) ia[j] = ((volatile u64 *)s)[(u64)h.a * FQSX_BKT + j];
    u32 fa, oa, fb = FQSX_BKT, ob = FQSX_BKT;
    tab_bucket_find(t, ia, v, fa, oa, ns_);
    if (fa != FQSX_BKT) return s + (u64)h.a * FQSX_BKT + fa;
    if (t.two) {
      for (u32 j = 0; j < FQSX_BKT; ++j) ib[j] = ((volatile u64 *)s)[(u64)h.b * FQSX_BKT + j];
      tab_bucket_find(t, ib, v, fb, ob, ns_);
      if (fb != FQSX_BKT) return s + (u64)h.b * FQSX_BKT + fb;
    }
    if (oa == FQSX_BKT && ob == FQSX_BKT) break;   // full: the chain
    u64 *slot = oa <= ob ? s + (u64)h.a * FQSX_BKT + oa : s + (u64)h.b * FQSX_BKT + ob;
    const u64 seen = atomic_cas64(slot, 0, item);
    if (seen == 0) { claimed = true; return slot; }
    if ((seen >> t.cbits) == v) return slot;
    // (somebody else's key went there: look again)
  }
  u32 p = t.two ? h.b : h.a;
  for (u64 n = 0; n <= t.nb; ++n) {
    p = t.two ? tab_next(t, h.a, p) : ((u64)p + 1 == t.nb ? 0u : p + 1u);
    for (u32 j = 0; j < FQSX_BKT; ++j) {
      u64 *slot = s + (u64)p * FQSX_BKT + j;
      u64 it = *(volatile u64 *)slot;
      if (!it) {
        const u64 seen = atomic_cas64(slot, 0, item);
        if (seen == 0) { claimed = true; return slot; }
        it = seen;
      }
      if ((it >> t.cbits) == v) return slot;
    }
  }
  return nullptr;   // (cannot happen: the tables are never full)
}
// exact look-up (count(), ht_kmer.h:330-362,441-453)
FQ_DEV u32 tab_count(const KTab &t, u32 sub, u64 kmer_norm, u64 &nslots) {
  const TabLoc L = tab_locate(t, t.slots + (u64)sub * t.stride, kmer_norm >> (64 - 2 * t.k));
  nslots += L.ns;
  return (u32)(L.item & ((1ull << t.cbits) - 1ull));
}
// wave-uniform insert used for the worker-private local tables (insert(), ht_kmer.h:420-438)
FQ_DEV void tab_insert_uniform(Wk &w, const KTab &t, u32 sub, u64 kmer_norm, u32 rng, const Cinc &ci) {
  u64 *s = t.slots + (u64)sub * t.stride;
  const u64 v = kmer_norm >> (64 - 2 * t.k);
  const u64 cm = (1ull << t.cbits) - 1ull;
  const TabLoc L = tab_locate(t, s, v);
  const u64 p = L.pos, it = L.item;
  if (p == ~0ull) { w.err = FQSX_ERR_LTAB_FULL; return; }
  if (!it) {
    const u32 f = t.filled[sub];
    if ((u64)f * 10 >= t.nb * FQSX_BKT * 9) { w.err = FQSX_ERR_LTAB_FULL; return; }
    s[p] = (v << t.cbits) | 1ull;  // Increment(0) == 1
    t.filled[sub] = f + 1;
    return;
  }
  const u32 cnt = (u32)(it & cm);
  if (cnt < (u32)cm && cinc_inc1(w.sm, rng, ci, cnt) != cnt) s[p] = it + 1;
}

// probe the first n entries of the LDS batch (keys/orientations) lane-parallel
FQ_DEV void batch_scan(Wk &w, const KTab &t, bool global, u32 n) {
  WgShared *sm = w.sm;
  u64 ns = 0;
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < n; i += FQ_WAVE) {
    C4 c;
    c4_zero(c);
    u64 key = sm->bk_key[i];
    tab_scan(t, global ? sb_owner(w.cfg, key) : w.tid, key, sm->bk_dir[i] != 0, c, ns);
    sm->bk_res[i][0] = c.c[0];
    sm->bk_res[i][1] = c.c[1];
    sm->bk_res[i][2] = c.c[2];
    sm->bk_res[i][3] = c.c[3];
  }
  FQ_SYNC();
  w.st[global ? ST_GPROBE : ST_LPROBE] += n;
  w.st[global ? ST_GSLOT : ST_LSLOT] += wave_sum64(ns);
}
// bit q of mask[q/64] = probe q of the batch returned a non-zero count
FQ_DEV void batch_hit_mask(Wk &w, u32 n, u64 mask[4]) {
  WgShared *sm = w.sm;
  mask[0] = mask[1] = mask[2] = mask[3] = 0;
#if FQ_WAVE > 1
  for (u32 r = 0; r * 64 < n; ++r) {
    u32 q = r * 64 + FQ_LANE;
    bool hit = q < n && (sm->bk_res[q][0] | sm->bk_res[q][1] | sm->bk_res[q][2] | sm->bk_res[q][3]) != 0;
    mask[r] = wave_ballot(hit);
  }
#else
  for (u32 q = 0; q < n; ++q)
    if (sm->bk_res[q][0] | sm->bk_res[q][1] | sm->bk_res[q][2] | sm->bk_res[q][3]) mask[q >> 6] |= 1ull << (q & 63);
#endif
}
FQ_DEV void batch_count(Wk &w, const KTab &t, u32 n) {  // exact counts of the batch keys (global table)
  WgShared *sm = w.sm;
  u64 ns = 0;
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < n; i += FQ_WAVE) {
    u64 key = sm->bk_key[i];
    sm->bk_res[i][0] = tab_count(t, sb_owner(w.cfg, key), key, ns);
  }
  FQ_SYNC();
  w.st[ST_GPROBE] += n;
  w.st[ST_GSLOT] += wave_sum64(ns);
}

// lane-parallel in-order insert of <= 64 keys into sub-table `sub` (defined below).  SM: the LDS block of the calling
// kernel -- WgShared in the encode kernels, the small InsShared in the insert-phase kernels (mt, mt_idx, ib_pos).
template <class SM>
FQ_DEV void insert_batch(const DevCfg &cfg, SM *sm, const KTab &t, u32 sub, const u64 *keys, u32 n, u32 rng, const Cinc &ci,
                         u64 &nslots, u32 &err);
template <class SM>
FQ_DEV void insert_batch_k(const DevCfg &cfg, SM *sm, const KTab &t, u32 sub, u64 mykey, u32 n, u32 rng, const Cinc &ci,
                           u64 &nslots, u32 &err);
// EXT_PF (the insert-phase kernel): a second wave of the workgroup touches the home buckets of the batches ahead (insert_prefetch_body),
// so this wave sends for nothing itself -- a touch issued in front of a batch's probe walk holds that walk's loads back until the touch
// (an HBM round trip) is home, because a wave's vector-memory operations retire in issue order -- and reports its progress instead.
template <class SM, bool EXT_PF = false>
FQ_DEV void insert_keys(const DevCfg &cfg, SM *sm, const KTab &t, u32 sub, const u64 *keys, u32 n, u32 rng, const Cinc &ci,
                        u64 &nslots, u32 &err) {
#if FQ_WAVE > 1
  if constexpr (EXT_PF) {
    u64 nextk = FQ_LANE < n ? keys[FQ_LANE] : 0;
    for (u32 o = 0; o < n && !err; o += FQ_WAVE) {
      const u64 k = nextk;
      if (o + FQ_WAVE + FQ_LANE < n) nextk = keys[o + FQ_WAVE + FQ_LANE];
      insert_batch_k(cfg, sm, t, sub, k, n - o < FQ_WAVE ? n - o : FQ_WAVE, rng, ci, nslots, err);
      lds_store_rel(&sm->pf_done, o / FQ_WAVE + 1);
    }
    return;
  }
  // the keys of the next batch are fetched while the current one is applied (the list is read-only here)
  // ... and the home slots of the next batch are touched one batch ahead, so that the ordered walk of a batch finds
  // its cache lines (and address translations) in L2 instead of paying the HBM round trip inside the serial chain
  const u64 *s = t.slots + (u64)sub * t.stride;
  u64 nextk = FQ_LANE < n ? keys[FQ_LANE] : 0;
  for (u32 o = 0; o < n && !err; o += FQ_WAVE) {
    const u64 k = nextk;
    u64 touch = 0;
    if (o + FQ_WAVE + FQ_LANE < n) {
      nextk = keys[o + FQ_WAVE + FQ_LANE];
      const TabHome th = tab_home(t, nextk >> (64 - 2 * t.k));
      touch = touch_load(&s[(u64)th.a * FQSX_BKT]) ^ touch_load(&s[(u64)th.b * FQSX_BKT]);
    }
    insert_batch_k(cfg, sm, t, sub, k, n - o < FQ_WAVE ? n - o : FQ_WAVE, rng, ci, nslots, err);
    keep_live(touch);
  }
#else
  for (u32 j = 0; j < n && !err; ++j) insert_batch(cfg, sm, t, sub, keys + j, 1, rng, ci, nslots, err);
#endif
}
// Local-table inserts (ht_*_local->insert, dna.cpp:826,839,861,872): every b-/s-mer a worker pushes to its
// mailbox list is also a local insert, so the list itself is the insert queue.  The entries are applied in list
// order as lane-parallel batches -- by the inserter wave of the workgroup as soon as they are published (encode
// kernel), or inline on demand (single-wave builds).  A look-up only has to wait for entries that could change its
// answer, so the tables always give exactly what the sequential algorithm would have seen.
FQ_DEV u32 lq_done_now(Wk &w, u32 qi) { return w.lqh ? lds_load_acq(&w.sm->lq_done[qi]) : w.la[qi ? MAIL_S : MAIL_B]; }
FQ_DEV void lq_publish(Wk &w, bool force = false) {   // hand the list entries written so far to the inserter wave
  if (!w.lqh) return;
  if (!force && (w.cfg->dbg & FQSX_DBG_INSERTER_STALL)) return;   // (test switch: only a waiting look-up publishes)
  const u32 nb = w.mn[MAIL_B], ns = w.mn[MAIL_S];
  if (nb != w.lq_pub[0]) { lds_store_rel(&w.sm->lq_target[0], nb); w.lq_pub[0] = nb; }
  if (ns != w.lq_pub[1]) { lds_store_rel(&w.sm->lq_target[1], ns); w.lq_pub[1] = ns; }
}
FQ_DEV void lq_flush(Wk &w, u32 kind) {
  const Mail &m = w.cfg->mail[kind];
  const u32 qi = kind == MAIL_S ? 1 : 0;
  const u32 n = w.mn[kind];
  if (w.lqh) {
    if (lds_load_acq(&w.sm->lq_done[qi]) >= n) return;
    TM_BEGIN(t_lq);
    TM_COUNT(w, CN_LQFLUSH);
    lq_publish(w, true);
    u32 spins = 0;
    while (lds_load_acq(&w.sm->lq_done[qi]) < n) {
      fq_sleep();
      if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
    }
    TM_END(w, TM_LQ, t_lq);
    return;
  }
  const u32 a = w.la[kind];
  if (a >= n) return;
  TM_BEGIN(t_lq);
  TM_COUNT(w, CN_LQFLUSH);
  u64 ns = 0;
  u32 err = 0;
  FQ_SYNC_MEM();  // the list entries were written by other lanes
  insert_keys(*w.cfg, w.sm, kind == MAIL_S ? w.cfg->l_s : w.cfg->l_b, w.tid, m.list + (u64)w.tid * m.cap + a, n - a,
              kind == MAIL_S ? RNG_LS : RNG_LB, kind == MAIL_S ? CINC_S : CINC_B, ns, err);
  w.la[kind] = n;
  w.st[ST_LINS] += n - a;
  if (err) w.err = FQSX_ERR_LTAB_FULL;
  TM_END(w, TM_LQ, t_lq);
}
// does one of the list entries [lo, hi) of kind qi fall into the sibling group of full k-mer km?
FQ_DEV bool pq_group_hit(Wk &w, u32 qi, const KGeom &g, const Kmer &km, u32 lo, u32 hi) {
  const u32 k2 = 2 * g.k;
  const bool nd = km_norm_dir(km, g);
  const u64 v = (nd ? km.dir : km.rc) >> (64 - k2);
  const u64 lowmask = (1ull << (k2 - 2)) - 1ull;
  const u64 grp = nd ? (v >> 2) : (v & lowmask);
  bool hit = false;
  FQ_SYNC();
  for (u32 t = lo + FQ_LANE; t < hi; t += FQ_WAVE) {
    u64 pv = w.sm->pq_key[qi][t & (FQSX_PQ - 1)] >> (64 - k2);
    hit |= nd ? ((pv >> 2) == grp) : ((pv & lowmask) == grp);
  }
  return wave_any(hit);
}

// Before a local look-up: wait for the pending inserts only if one of them could change the answer
// (same sibling group as the looked-up full k-mer), the mirror no longer covers them, or the k-mer is partial.
FQ_DEV void lq_sync_for(Wk &w, u32 kind, const KGeom &g, const Kmer &km) {
  const u32 qi = kind == MAIL_S ? 1 : 0;
  const u32 n = w.mn[kind], d = lq_done_now(w, qi);
  if (d >= n) return;
  bool need = km.cur != g.k || n - d > FQSX_PQ;
  if (!need) need = pq_group_hit(w, qi, g, km, d, n);
  if (need) lq_flush(w, kind);
}

// find / find_full / find_partial (ht_kmer.h:189-203,266-327,504-510)
FQ_DEV bool kt_find(Wk &w, const KTab &t, bool global, const KGeom &g, const Kmer &km, u32 rng, const Cinc &ci, C4 &out, u32 pre = 0xff) {
  c4_zero(out);
  if (pre != 0xff) {   // the scout wave has probed this look-up's trials already (ep_res, trial order): merges only
    const u32 m = g.k - km.cur, cnt = 1u << (2 * m);
    u64 hm = 0;
    u32 nsl = 0;
    FQ_SYNC();
#if FQ_WAVE > 1
    {
      const bool in = FQ_LANE < cnt;   // cnt <= 16
      const u32 e = pre + (in ? FQ_LANE : 0u);
      hm = wave_ballot(in && w.sb->ep_res[e] != 0);
      nsl = in ? w.sb->ep_ns[e] : 0u;
    }
#else
    for (u32 i = 0; i < cnt; ++i) {
      hm |= (u64)(w.sb->ep_res[pre + i] != 0) << i;
      nsl += w.sb->ep_ns[pre + i];
    }
#endif
    w.st[ST_GPROBE] += cnt;
    w.st[ST_GSLOT] += wave_sum32(nsl);
    if (m == 0) {
      const u64 v = w.sb->ep_res[pre];
      out.c[0] = (u32)(v & 0xffff); out.c[1] = (u32)((v >> 16) & 0xffff); out.c[2] = (u32)((v >> 32) & 0xffff); out.c[3] = (u32)(v >> 48);
      return c4_any(out);
    }
    for (u64 mk = hm; mk; mk &= mk - 1) {
      const u64 v = w.sb->ep_res[pre + ctz64(mk)];
      for (u32 x = 0; x < 4; ++x) {
        const u32 loc = (u32)((v >> (16 * x)) & 0xffff);
        if (loc) out.c[x] = cinc_merge(w.sm, rng, ci, out.c[x], loc);  // ht_kmer.h:321-323
      }
    }
    return c4_any(out);
  }
  if (km.cur == g.k) {
    u64 ns = 0;
    u64 key = km_norm(km, g);
    tab_scan(t, global ? sb_owner(w.cfg, key) : w.tid, key, km_norm_dir(km, g), out, ns);
    w.st[global ? ST_GPROBE : ST_LPROBE] += 1;
    w.st[global ? ST_GSLOT : ST_LSLOT] += ns;
    return c4_any(out);
  }
  // partial k-mer: all 4^m front paddings, position 0 is the fastest odometer digit (ht_kmer.h:291-310)
  WgShared *sm = w.sm;
  const u32 m = g.k - km.cur;
  const u32 trials = 1u << (2 * m);
  const u64 dir_pad = km.dir >> (2 * m);
  u64 rc_pad = km.rc & (~0ull << (64 - 2 * km.cur));  // rc already holds cur symbols at the top
  for (u32 base = 0; base < trials; base += 256) {
    u32 n = trials - base < 256 ? trials - base : 256;
    FQ_SYNC();
    for (u32 i = FQ_LANE; i < n; i += FQ_WAVE) {
      u32 t_i = base + i;
      u64 d = dir_pad, r = rc_pad;
      for (u32 j = 0; j < m; ++j) {
        u64 sym = (t_i >> (2 * j)) & 3;
        d |= sym << (62 - 2 * j);
        r |= (3 - sym) << (64 - 2 * g.k + 2 * j);
      }
      bool nd = (d & g.kernel_mask) < (r & g.kernel_mask);
      sm->bk_key[i] = nd ? d : r;
      sm->bk_dir[i] = nd ? 1 : 0;
    }
    batch_scan(w, t, global, n);
    u64 hm[4];
    batch_hit_mask(w, n, hm);
    for (u32 r = 0; r < 4; ++r)
      for (u64 mk = hm[r]; mk; mk &= mk - 1) {  // trials in order, only those that found something
        u32 i = r * 64 + ctz64(mk);
        for (u32 s = 0; s < 4; ++s) {
          u32 loc = sm->bk_res[i][s];
          if (loc) out.c[s] = cinc_merge(sm, rng, ci, out.c[s], loc);  // ht_kmer.h:321-323
        }
      }
  }
  return c4_any(out);
}

// ---------------------------------------------------------------------------------------
// p-mer vector (bit_vec.h)
FQ_DEV u64 siv_test(const DevCfg *cfg, u64 idx) { return (cfg->siv[idx >> 5] >> (2 * (idx & 31))) & 3; }  // bit_vec.h:69-81
FQ_DEV void siv_counts(Wk &w, u64 idx, C4 &c) {  // counts(), bit_vec.h:83-96
  u64 d = w.cfg->siv[idx >> 5];
  u32 sh = 2 * (u32)((idx & 31) & ~3ull);
  c.c[0] = (u32)((d >> sh) & 3);
  c.c[1] = (u32)((d >> (sh + 2)) & 3);
  c.c[2] = (u32)((d >> (sh + 4)) & 3);
  c.c[3] = (u32)((d >> (sh + 6)) & 3);
  w.st[ST_SIV_WORDS] += 1;
}
FQ_DEV u64 word_field_sum(u64 d) {  // sum of the 32 2-bit fields
  u64 x = (d & 0x3333333333333333ULL) + ((d >> 2) & 0x3333333333333333ULL);
  x = (x & 0x0f0f0f0f0f0f0f0fULL) + ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL);
  return (x * 0x0101010101010101ULL) >> 56;
}
// sum of fields [start,end)  (test_shorter, bit_vec.h:113-188), lane-parallel over words
FQ_DEV u64 siv_range_sum(Wk &w, u64 start, u64 end) {
  if (start >= end) return 0;
  const u64 *sv = w.cfg->siv;
  u64 w0 = start >> 5, w1 = (end - 1) >> 5, r = 0;
  for (u64 x = w0 + FQ_LANE; x <= w1; x += FQ_WAVE) {
    u64 d = sv[x];
    if (x == w0) d &= ~0ull << (2 * (start & 31));
    if (x == w1 && (end & 31)) d &= ~(~0ull << (2 * (end & 31)));
    r += word_field_sum(d);
  }
  w.st[ST_SIV_WORDS] += w1 - w0 + 1;
  return wave_sum64(r);
}
// #{ i in (lo,hi) : field(i) == flag }  -- the p-mer rank sweep of compress_prefix_sorted, dna.cpp:600-605.
// Pure streaming: every word of [w0,w1] is counted unmasked with eight independent 16-byte loads per lane in
// flight (8 KiB per wave iteration; a lone wave is latency-bound, so bytes in flight are what sets its rate),
// then the fields outside (lo,hi) in the two end words are taken off again.
FQ_DEV u64 siv_eq_count(u64 d, u64 rep) {
  d ^= rep;
  return popc64(~(d | (d >> 1)) & 0x5555555555555555ULL);
}
// fields of [start, hi) equal to `flag`, by sweeping the words
FQ_DEV u64 siv_count_sweep(Wk &w, u64 start, u64 hi, u64 flag) {
  if (start >= hi) return 0;
  const u64 *sv = w.cfg->siv;
  const u64 rep = flag * 0x5555555555555555ULL;
  const u64 w0 = start >> 5, w1 = (hi - 1) >> 5;
  const u64 m0 = ~0ull << (2 * (start & 31));
  const u64 m1 = (hi & 31) ? ~(~0ull << (2 * (hi & 31))) : ~0ull;
  u64 r = 0;
  u64 x = w0;
  const u64 end = w1 + 1;
  if ((x & 1) && x < end) {  // align to 16 bytes
    if (FQ_LANE == 0) r += siv_eq_count(sv[x], rep);
    ++x;
  }
  struct alignas(16) W2 { u64 a, b; };
  const W2 *sv2 = (const W2 *)(sv + x);   // x is even: 16-byte aligned (the vector itself is page aligned)
  const u64 pairs = (end - x) >> 1;
  u64 q = FQ_LANE;
  for (; q + 7 * FQ_WAVE < pairs; q += 8 * FQ_WAVE) {
    W2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = sv2[q + k * FQ_WAVE];
#pragma unroll
    for (int k = 0; k < 8; ++k) r += siv_eq_count(v[k].a, rep) + siv_eq_count(v[k].b, rep);
  }
  for (; q < pairs; q += FQ_WAVE) {
    W2 v = sv2[q];
    r += siv_eq_count(v.a, rep) + siv_eq_count(v.b, rep);
  }
  x += 2 * pairs;
  if (x < end && FQ_LANE == 0) r += siv_eq_count(sv[x], rep);   // odd tail word
  r = wave_sum64(r);
  // fields of the end words that lie outside [start, hi)
  u64 d0 = sv[w0] ^ rep, d1 = sv[w1] ^ rep;
  u64 e0 = ~(d0 | (d0 >> 1)) & 0x5555555555555555ULL, e1 = ~(d1 | (d1 >> 1)) & 0x5555555555555555ULL;
  if (w0 == w1) r -= popc64(e0 & ~(m0 & m1));
  else r -= popc64(e0 & ~m0) + popc64(e1 & ~m1);
  w.st[ST_SIV_WORDS] += w1 - w0 + 1;
  return r;
}
// fields equal to `flag` in the whole blocks [b0, b1) of the vector, from the count index (16 bytes per 4 KiB block)
FQ_DEV u64 siv_count_blocks(Wk &w, u64 b0, u64 b1, u64 flag) {
  if (b0 >= b1) return 0;
  struct alignas(16) I4 { u32 c[4]; };
  const I4 *ix = (const I4 *)w.cfg->siv_idx;
  u64 r = 0;
  u64 q = b0 + FQ_LANE;
  for (; q + 3 * FQ_WAVE < b1; q += 4 * FQ_WAVE) {
    I4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ix[q + k * FQ_WAVE];
#pragma unroll
    for (int k = 0; k < 4; ++k) r += flag ? (flag == 1 ? v[k].c[1] : flag == 2 ? v[k].c[2] : v[k].c[3]) : FQSX_SIV_BLK - v[k].c[1] - v[k].c[2] - v[k].c[3];
  }
  for (; q < b1; q += FQ_WAVE) {
    const I4 v = ix[q];
    r += flag ? (flag == 1 ? v.c[1] : flag == 2 ? v.c[2] : v.c[3]) : FQSX_SIV_BLK - v.c[1] - v.c[2] - v.c[3];
  }
  w.st[ST_SIV_WORDS] += 2 * (b1 - b0);
  return wave_sum64(r);
}
FQ_DEV u64 siv_count_equal(Wk &w, u64 lo, u64 hi, u64 flag) {
  const u64 start = lo + 1;
  if (start >= hi) return 0;
  const u64 b0 = (start + FQSX_SIV_BLK - 1) >> FQSX_SIV_BLK_LOG, b1 = hi >> FQSX_SIV_BLK_LOG;   // whole blocks inside [start, hi): b0 .. b1-1
  if (b0 + 2 > b1) return siv_count_sweep(w, start, hi, flag);   // (a short range: the sweep alone)
  const u64 read_before = w.st[ST_SIV_WORDS];
  const u64 r = siv_count_sweep(w, start, b0 << FQSX_SIV_BLK_LOG, flag) + siv_count_blocks(w, b0, b1, flag) + siv_count_sweep(w, b1 << FQSX_SIV_BLK_LOG, hi, flag);
  w.st[ST_SIV_SAVED] += (((hi - 1) >> 5) - (start >> 5) + 1) - (w.st[ST_SIV_WORDS] - read_before);
  return r;
}

// ---------------------------------------------------------------------------------------
// range coder (CRangeEncoder, sub_rc.h:32-87) writing into the worker's HBM stream
// The coder state is wave-uniform and kept in scalar registers (readfirstlane at enc_open), so the dependent chain of
// a coding step -- multiply-high, compare, add, shift -- runs on the scalar unit.  Output bytes are gathered into the
// aligned 8-byte word they belong to and leave with one store per word.
// What-if profiling (-DFQSX_WHATIF, tools/gpu_whatif.py): `ev` events of role `role` cost an extra ev * units * ~0.2 us when the
// run asks for that role (DevCfg.whatif) -- the file's slowdown per microsecond added says how much of the role is critical path.
#ifdef FQSX_WHATIF
FQ_DEV void whatif_delay(const DevCfg &cfg, u32 role, u32 ev) {
  if ((cfg.whatif >> 16) != role) return;
  const u32 n = (cfg.whatif & 0xffffu) * ev;
  for (u32 i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}
#define WHATIF(cfg, role, ev) whatif_delay(cfg, role, ev)
#else
#define WHATIF(cfg, role, ev) do { } while (0)
#endif
FQ_DEV void enc_open(Wk &w, u64 low, u64 range, u64 len, const DevCfg &cfg) {
  w.enc.low = uniform64(low); w.enc.range = uniform64(range); w.enc.len = uniform64(len);
  w.enc.cap = cfg.out_cap; w.enc.out = cfg.out + (u64)w.tid * cfg.out_cap;
  w.enc.acc = 0;
  const u32 part = (u32)(w.enc.len & 7);
  if (part && w.enc.len < w.enc.cap) w.enc.acc = uniform64(((const u64 *)w.enc.out)[w.enc.len >> 3]) & ((1ull << (8 * part)) - 1ull);
}
FQ_DEV void enc_close(Wk &w) {   // the partly filled word (its tail is rewritten when the stream goes on)
  if (w.enc.len > w.enc.cap) { w.err = FQSX_ERR_OUT_OVERFLOW; return; }   // (ended inside the word beyond the buffer: rc_put has not seen it)
  if ((w.enc.len & 7) && w.enc.len < w.enc.cap) ((u64 *)w.enc.out)[w.enc.len >> 3] = w.enc.acc;
}
FQ_DEV void rc_put(Wk &w, u8 b) {
  w.enc.acc |= (u64)b << (8 * (u32)(w.enc.len & 7));
  ++w.enc.len;
  if ((w.enc.len & 7) == 0) {
    if (w.enc.len <= w.enc.cap) ((u64 *)w.enc.out)[(w.enc.len >> 3) - 1] = w.enc.acc; else w.err = FQSX_ERR_OUT_OVERFLOW;
    w.enc.acc = 0;
  }
}
// exact range / tot for tot < 2^16 without the 64-bit software divide or an IEEE fp64 division: hardware
// reciprocal (~26 bits) + one Newton step (~52 bits), high word and the remaining < 2^48 dividend each by one fp64
// multiply with a +-1 fix-up (sub_rc.h:63).  tools/ubench checks it against u64 division on 1.3e9 operands.
FQ_DEV double recip_u16(u32 d) {
  const double dd = (double)d;
#ifndef FQSX_EMU
  const double r0 = __builtin_amdgcn_rcp(dd);
  return __builtin_fma(r0, __builtin_fma(-dd, r0, 1.0), r0);
#else
  return 1.0 / dd;
#endif
}
FQ_DEV u64 div_u64_rd(u64 x, u32 d, double rd) {
  const u32 hi = (u32)(x >> 32), lo = (u32)x;
  u32 qh = (u32)((double)hi * rd);
  u32 ph = qh * d;
  if (ph > hi) { --qh; ph -= d; } else if (hi - ph >= d) { ++qh; ph += d; }
  const u32 r1 = hi - ph;                                                        // < d
  const double remd = __builtin_fma((double)r1, 4294967296.0, (double)lo);      // exact: < 2^48
  u32 q = (u32)(remd * rd);                                                      // rem / d < 2^32
  const u64 rem = ((u64)r1 << 32) | lo;
  const u64 prod = (u64)q * d;
  if (prod > rem) --q;
  else if (rem - prod >= d) ++q;
  return ((u64)qh << 32) + q;
}
FQ_DEV u64 div_u64_small(u64 x, u32 d) { return div_u64_rd(x, d, recip_u16(d)); }
// One coding step (Encode, sub_rc.h:60-77) with the division as an integer multiply-high by
// m = floor((2^64-1) / tot), 2 <= tot < 2^16: for any range < 2^64, mulhi(range, m) is the quotient or one less.
// Integer only and wave-uniform: the whole dependent chain of a symbol runs on the scalar unit (low / range live in
// scalar registers, see enc_open).  m is computed off the chain where the caller can (one lane per position in code_run).
FQ_DEV u64 recip64_u16(u32 d) { return div_u64_rd(~0ull, d, recip_u16(d)); }
// model wave -> range-coder wave: room for `need` more entries
FQ_DEV bool rq_wait_space(Wk &w, u32 need) {
  if (w.rq_tail - lds_load_acq(&w.sm->rq_head) + need <= FQSX_RQ) return true;
  TM_BEGIN(t_rq);
  u32 spins = 0;
  while (w.rq_tail - lds_load_acq(&w.sm->rq_head) + need > FQSX_RQ) {
    fq_sleep();
    if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; return false; }   // never spin forever on the GPU
  }
  TM_END(w, TM_CQWAIT, t_rq);
  return true;
}
FQ_DEV void rc_encode_m(Wk &w, u32 freq, u32 cum, u32 tot, u64 m) {
  if (w.rcq) {   // hand the step to the range-coder wave
    if (!rq_wait_space(w, 1)) return;
    WgShared *sm = w.sm;
    const u32 e = w.rq_tail & (FQSX_RQ - 1);
    FQ_SYNC();
    if (FQ_LANE == 0) { sm->rq_f[e] = freq; sm->rq_c[e] = cum; sm->rq_t[e] = tot; sm->rq_m[e] = m; }
    FQ_SYNC();
    w.rq_tail += 1;
    lds_store_rel(&sm->rq_tail, w.rq_tail);
    return;
  }
  const u64 Top = 0x00ffffffffffffULL, M = 0xff00000000000000ULL;
  freq = uniform32(freq); cum = uniform32(cum); tot = uniform32(tot); m = uniform64(m);
  u64 low = w.enc.low;
#ifndef FQSX_EMU
  u64 range = __umul64hi(w.enc.range, m);
#else
  u64 range = (u64)(((unsigned __int128)w.enc.range * m) >> 64);
#endif
  // mulhi gives the quotient or one less, so the remainder is below 2 * tot < 2^17: its low 32 bits decide (one
  // multiply, one subtract, one compare on the scalar unit instead of a 64-bit multiply-subtract-compare)
  if ((u32)w.enc.range - (u32)range * tot >= tot) ++range;
  low += range * cum;
  range *= freq;
  while (range <= Top) {
    if ((low ^ (low + range)) & M) range = (low | Top) - low;
    rc_put(w, (u8)(low >> 56));
    low <<= 8;
    range <<= 8;
  }
  w.enc.low = low;
  w.enc.range = range;
  w.st[ST_CODED] += 1;
}
FQ_DEV void rc_encode_rd(Wk &w, u32 freq, u32 cum, u32 tot, double rd) {
  rc_encode_m(w, freq, cum, tot, div_u64_rd(~0ull, tot, rd));
}

// producer side of the coding queue: wait until `need` more entries fit
FQ_DEV bool cq_wait_space(Wk &w, u32 need) {
  if (!w.piped) return true;
  if (w.cq_tail - lds_load_acq(&w.sm->cq_head) + need <= FQSX_CQ) return true;
  TM_BEGIN(t_cq);
  u32 spins = 0;
  while (w.cq_tail - lds_load_acq(&w.sm->cq_head) + need > FQSX_CQ) {
    fq_sleep();
    if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; return false; }   // never spin forever on the GPU
  }
  TM_END(w, TM_CQWAIT, t_cq);
  return true;
}
FQ_DEV void cq_publish(Wk &w, u32 n) {
  w.cq_tail += n;
  if (w.piped) lds_store_rel(&w.sm->cq_tail, w.cq_tail);
}
// A symbol of a small direct-indexed model: coded right here, or (two-wave kernel) handed to the coder wave as a
// finished triple so that it keeps its place in the stream.
FQ_DEV void rc_encode(Wk &w, u32 freq, u32 cum, u32 tot) {
  if (w.rec) {   // read-head wave: the triple goes into the read's record
    const u32 n = w.rec->n_raw;
    FQ_SYNC();
    if (n < FQSX_HD_RAW) {
      if (FQ_LANE == 0) { w.rec->raw[n][0] = (u64)freq | ((u64)cum << 32); w.rec->raw[n][1] = tot; w.rec->n_raw = n + 1; }
    } else w.err = FQSX_ERR_PIPE;
    FQ_SYNC();
    return;
  }
  if (!w.piped) { rc_encode_rd(w, freq, cum, tot, recip_u16(tot)); return; }
  if (!cq_wait_space(w, 1)) return;
  WgShared *sm = w.sm;
  const u32 e = w.cq_tail & (FQSX_CQ - 1);
  FQ_SYNC();
  if (FQ_LANE == 0) {
    sm->cq_key[0][e] = (u64)freq | ((u64)cum << 32);
    sm->cq_key[1][e] = tot;
    sm->cq_kind[e] = SK_RAW;
    sm->cq_rsym[e] = 5;
  }
  FQ_SYNC();
  cq_publish(w, 1);
}
// small direct-indexed adaptive model in HBM: N stats + total (CSimpleModel, rc.h:20-173; Encode rc.h:397-405)
FQ_DEV void sm_encode(Wk &w, u16 *m, u32 n, u32 max_total, u32 x) {
  u32 cum = 0;
  for (u32 i = 0; i < x; ++i) cum += m[i];
  u32 f = m[x], tot = m[n];
  rc_encode(w, f, cum, tot);
  f += 4;
  tot += 4;
  m[x] = (u16)f;
  if (tot >= max_total) {  // rescale, rc.h:28-39
    while (tot >= max_total) {
      tot = 0;
      for (u32 i = 0; i < n; ++i) {
        u32 v = (m[i] + 1u) / 2u;
        m[i] = (u16)v;
        tot += v;
      }
    }
  }
  m[n] = (u16)tot;
}
// 256-symbol model (prefix_sorted_bytes): cumulative sum and rescale are lane-parallel
FQ_DEV void sm_encode256(Wk &w, u16 *m, u8 *init_flag, u32 x) {
  if (!*init_flag) {
    FQ_SYNC_MEM();
    for (u32 i = FQ_LANE; i < 256; i += FQ_WAVE) m[i] = 1;
    m[256] = 256;
    *init_flag = 1;
    FQ_SYNC_MEM();
  }
  u32 part = 0;
  for (u32 i = FQ_LANE; i < x; i += FQ_WAVE) part += m[i];
  u32 cum = wave_sum32(part);
  u32 f = m[x], tot = m[256];
  rc_encode(w, f, cum, tot);
  tot += 4;
  FQ_SYNC_MEM();
  m[x] = (u16)(f + 4);
  FQ_SYNC_MEM();
  while (tot >= (1u << 15)) {
    u32 p = 0;
    for (u32 i = FQ_LANE; i < 256; i += FQ_WAVE) {
      u32 v = (m[i] + 1u) / 2u;
      m[i] = (u16)v;
      p += v;
    }
    tot = wave_sum32(p);
    FQ_SYNC_MEM();
  }
  m[256] = (u16)tot;
}

// ---------------------------------------------------------------------------------------
// context map of 5-symbol models (CContextHM, context_hm.h:21-248; exact map semantics:
// inserting an existing key leaves the earlier entry in place, which is what shadows the
// later duplicate in the reference's linear probe)
struct Slot4 { u64 q0, q1, q2, q3; };  // key | counter,tag,total | st0..3 | st4
FQ_DEV u64 *ctx_base(Wk &w) { return (u64 *)(w.cfg->ctx + (u64)w.tid * (w.cfg->ctx_cap_mask + 1)); }
// slot hash of the context table (our layout, so any good mixer will do: 32-bit finaliser)
FQ_DEV u32 ctx_mix(u32 tag, u64 key) {
  u32 x = (u32)key ^ ((u32)(key >> 32) * 0x9E3779B1u) ^ (tag * 0x85EBCA6Bu);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
FQ_DEV u64 ctx_hash(Wk &w, u32 tag, u64 key) { return (u64)ctx_mix(tag, key) & w.cfg->ctx_cap_mask; }
FQ_DEV u32 slot_counter(const Slot4 &s) { return (u32)s.q1; }
FQ_DEV u32 slot_tag(u64 q1) { return (u32)(q1 >> 32) & 0xffffu; }
FQ_DEV u32 ctx_find(Wk &w, u32 tag, u64 key, Slot4 &s) {
  const u64 *b = ctx_base(w);
  u64 h = ctx_hash(w, tag, key);
  for (u64 n = 0; n <= w.cfg->ctx_cap_mask; ++n) {
    const u64 *p = (const u64 *)__builtin_assume_aligned(b + 4 * h, 32);
    u64 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];  // one 32-byte slot, fetched in one round trip
#ifndef FQSX_EMU
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));   // ... which the compiler would otherwise split by sinking the key load below the tag test
    q0 = uniform64(q0); q1 = uniform64(q1); q2 = uniform64(q2); q3 = uniform64(q3);   // (wave-uniform: keeps the search and the coder state on the scalar unit)
#endif
    w.st[ST_CTX] += 1;
    u32 tg = slot_tag(q1);
    if (!tg) return FQSX_NIL;
    if (tg == tag && q0 == key) {
      s.q0 = q0; s.q1 = q1; s.q2 = q2; s.q3 = q3;
      return (u32)h;
    }
    h = (h + 1) & w.cfg->ctx_cap_mask;
  }
  return FQSX_NIL;
}
// insert (key -> model copy, counter 0) unless present; returns the slot holding the key
FQ_DEV u32 ctx_insert(Wk &w, u32 tag, u64 key, u64 q2, u64 q3, u32 total, Slot4 &s) {
  u64 *b = ctx_base(w);
  u64 h = ctx_hash(w, tag, key);
  for (u64 n = 0; n <= w.cfg->ctx_cap_mask; ++n) {
    u64 *p = b + 4 * h;
    u64 q0 = p[0], q1 = p[1];
    u32 tg = slot_tag(q1);
    if (!tg) {
      u32 f = w.cfg->ctx_filled[w.tid];
      if ((u64)f * 10 >= (w.cfg->ctx_cap_mask + 1) * 9) { w.err = FQSX_ERR_CTX_FULL; return FQSX_NIL; }
      w.cfg->ctx_filled[w.tid] = f + 1;
      s.q0 = key;
      s.q1 = ((u64)tag << 32) | ((u64)total << 48);
      s.q2 = q2;
      s.q3 = q3;
      p[0] = s.q0; p[1] = s.q1; p[2] = s.q2; p[3] = s.q3;
      return (u32)h;
    }
    if (tg == tag && q0 == key) {
      s.q0 = q0; s.q1 = q1; s.q2 = p[2]; s.q3 = p[3];
      return (u32)h;
    }
    h = (h + 1) & w.cfg->ctx_cap_mask;
  }
  w.err = FQSX_ERR_CTX_FULL;
  return FQSX_NIL;
}
FQ_DEV void ctx_store_counter(Wk &w, u32 idx, Slot4 &s, u32 counter) {
  s.q1 = (s.q1 & 0xffffffff00000000ULL) | counter;
  ctx_base(w)[4 * (u64)idx + 1] = s.q1;
}

// level thresholds (dna.h:33,36)
// {1,32,64,64,128,512,1024,32,128,256,2048,4} and {1,32,64,128,256,512,2048,4096,8192,16384,16384,4}: powers of
// two, kept as packed exponents
FQ_DEV u32 code_thr(u32 i) { return 1u << ((0x2B875A976650ULL >> (4 * i)) & 15); }
FQ_DEV u32 letters_thr(u32 i) { return 1u << ((0x2EEDCB987650ULL >> (4 * i)) & 15); }

// find_rc_code_context / find_rc_letters_context (dna.cpp:2107-2286): hierarchical level
// search with lazy clone-on-threshold; returns slot index (model + already incremented counter in s)
// `lev` points to the level keys in LDS; `rs` is added to the keys of levels >= 2 (the r_sym field the
// speculation stage left open)
// ls: distance (in keys) between two levels' keys of one symbol: 1 for a plain array, FQSX_CQ inside the coding queue
#define LEVKEY(l) (lev[(u32)(l) * ls] + ((l) >= 2 ? rs : 0))
FQ_DEV u32 find_leveled(Wk &w, u32 tag, const u64 *lev, u32 ls, u64 rs, int n_levels, double &avg, u64 tpl_q2, u64 tpl_q3, u32 tpl_total, Slot4 &s) {
  int i;
  Slot4 q;
  const bool letters = tag == 2;
  int start = (int)(avg + 0.49);
  u32 p = ctx_find(w, tag, LEVKEY(start), s);
  if (p != FQSX_NIL && slot_counter(s) < (letters ? letters_thr(start) : code_thr(start))) {
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
    avg = ema_update(avg, (double)start);
    return p;
  }
  if (p == FQSX_NIL) {
    for (i = start - 1; i >= 0; --i) {
      p = ctx_find(w, tag, LEVKEY(i), s);
      if (p != FQSX_NIL) break;
    }
  } else {
    for (i = start + 1; i < n_levels; ++i) {
      u32 qi = ctx_find(w, tag, LEVKEY(i), q);
      if (qi == FQSX_NIL) break;
      if (slot_counter(q) < code_thr(i)) {  // both routines use the *code* thresholds here (quirk, dna.cpp:2244)
        avg = ema_update(avg, (double)i);
        ctx_store_counter(w, qi, q, slot_counter(q) + 1);
        s = q;
        return qi;
      }
      p = qi;
      s = q;
    }
    --i;
  }
  if (p == FQSX_NIL) {  // nothing known: create level 0 from the template
    p = ctx_insert(w, tag, LEVKEY(0), tpl_q2, tpl_q3, tpl_total, s);
    if (p == FQSX_NIL) return p;
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
    i = 0;
  }
  if (slot_counter(s) >= code_thr(i) && i + 1 < n_levels) {  // clone into the next level (dna.cpp:2177-2184)
    u32 total = (u32)(s.q1 >> 48);
    Slot4 c;
    u32 ci = ctx_insert(w, tag, LEVKEY(i + 1), s.q2, s.q3, total, c);
    if (ci == FQSX_NIL) return ci;
    ctx_store_counter(w, ci, c, slot_counter(c) + 1);
    s = c;
    p = ci;
  } else
    ctx_store_counter(w, p, s, slot_counter(s) + 1);
  avg = ema_update(avg, (double)i);
  return p;
}

// Encode with a context slot's 5-symbol model (CSimpleModelFixedSize<5>, rc.h:178-338,478-488)
FQ_DEV u32 hsum4x16(u64 v) { return (u32)((v * 0x0001000100010001ULL) >> 48); }  // sum of four 16-bit fields (< 2^16)
FQ_DEV void slot_encode(Wk &w, u32 idx, Slot4 &s, u32 x) {
  u32 tot = (u32)(s.q1 >> 48);
  const u32 sh = 16 * (x & 3);
  u32 cum = hsum4x16(x >= 4 ? s.q2 : s.q2 & ((1ull << sh) - 1ull));
  u32 f = x >= 4 ? (u32)(s.q3 & 0xffff) : (u32)((s.q2 >> sh) & 0xffff);
  rc_encode(w, f, cum, tot);
  if (x >= 4) s.q3 += 4; else s.q2 += 4ull << sh;   // fields stay < 2^16: total < 2^15 + 4
  tot += 4;
  if (tot >= (1u << 15)) {  // rescale, rc.h:186-197
    u32 st[5] = {(u32)(s.q2 & 0xffff), (u32)((s.q2 >> 16) & 0xffff), (u32)((s.q2 >> 32) & 0xffff), (u32)(s.q2 >> 48), (u32)(s.q3 & 0xffff)};
    while (tot >= (1u << 15)) {
      tot = 0;
      for (u32 i = 0; i < 5; ++i) {
        st[i] = (st[i] + 1) / 2;
        tot += st[i];
      }
    }
    s.q2 = (u64)st[0] | ((u64)st[1] << 16) | ((u64)st[2] << 32) | ((u64)st[3] << 48);
    s.q3 = (s.q3 & ~0xffffULL) | st[4];
  }
  s.q1 = (s.q1 & 0x0000ffffffffffffULL) | ((u64)tot << 48);
  u64 *p = ctx_base(w) + 4 * (u64)idx;
  p[1] = s.q1; p[2] = s.q2; p[3] = s.q3;
}

// ---------------------------------------------------------------------------------------
// context keys (code_ctx.cpp)
// The quantisers of a count (code_ctx.cpp:15-239) are "digits up to D, then one class per threshold passed".  The thresholds
// are compile-time constants: the class is a sum of comparisons against immediates -- no table in memory, no loop whose
// trip count depends on the lane's count (64 lanes with 64 different counts run the longest of them).
#define GE(x) ((u32)(c >= (x)))
FQ_DEV u64 conv_lev1(u64 c, u32 cl) {  // code_ctx.cpp:26-81
  const u64 f = (u64)cl << 5;
  if (cl == 0) return f + (c < 5 ? c : 5 + GE(8) + GE(16) + GE(32) + GE(64));
  if (c < 8) return f + c;
  return f + 8 + GE(16) + GE(24) + GE(32) + GE(40) + GE(48) + GE(56) + GE(64) + GE(80) + GE(96) + GE(112) + GE(128) + GE(144) + GE(160) +
         GE(176) + GE(192) + GE(224) + GE(288) + GE(384) + GE(512) + GE(1024) + GE(2048);
}
FQ_DEV u64 conv_low(u64 c, u32 cl, bool lev3) {  // shared rows of code_ctx.cpp:84-110 and :167-198
  const u64 f = (u64)cl << 5;
  if (cl == 0) return f + (c < 3 ? c : 3 + GE(5));
  if (cl == 1) return f + (c < 5 ? c : 5 + GE(8) + GE(13) + GE(20) + GE(30));
  // cl == 2
  return f + (c < 10 ? c : 10 + GE(lev3 ? 13u : 15u) + GE(20) + GE(30) + GE(50));
}
FQ_DEV u64 conv_lev24(u64 c, u32 cl) {  // code_ctx.cpp:84-164
  if (cl < 3) return conv_low(c, cl, false);
  const u64 f = (u64)cl << 5;
  if (c < 10) return f + c;
  return f + 10 + GE(16) + GE(24) + GE(32) + GE(48) + GE(64) + GE(128) + GE(256) + GE(512) + GE(1024) + GE(2048) + GE(2080) + GE(2112) + GE(2176) +
         GE(2240) + GE(2304) + GE(2432) + GE(2560) + GE(2816) + GE(3072);
}
FQ_DEV u64 conv_lev3(u64 c, u32 cl) {  // code_ctx.cpp:167-239
  if (cl < 3) return conv_low(c, cl, true);
  const u64 f = (u64)cl << 5;
  if (c < 15) return f + c;
  return f + 15 + GE(18) + GE(20) + GE(25) + GE(30) + GE(40) + GE(50) + GE(60) + GE(64) + GE(68) + GE(72) + GE(76) + GE(80) + GE(84) + GE(88);
}
#undef GE
enum { LV_NONE = 0, LV_PMER = 1, LV_SMER = 2, LV_BMER = 3, LV_MIXED = 4, LV_BMER_UNC = 5 };  // defs.h:45
FQ_DEV u64 conv_count(u64 c, u32 level, u32 cl) {  // code_ctx.cpp:15-23
  if (level == LV_PMER) return conv_lev1(c, cl);
  if (level == LV_BMER) return conv_lev3(c, cl);
  return conv_lev24(c, cl);
}
// The same quantisers as threshold tables (count of thresholds <= c): used by the wave-uniform
// variant below, where the 24 compares of one quantiser are one ballot.
#ifndef FQSX_EMU
#define FQ_CONST __device__ const
#else
#define FQ_CONST static const
#endif
#define NOLIM 0xffffffffu
FQ_CONST u32 CONV_D[12] = {5, 8, 8, 8, 3, 5, 10, 10, 3, 5, 10, 15};
FQ_CONST u32 CONV_LIM[12][24] = {
    // level pmer (conv_lev1), cnt_lev 0..3
    {8, 16, 32, 64, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 288, 384, 512, 1024, 2048, NOLIM, NOLIM, NOLIM},
    {16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 288, 384, 512, 1024, 2048, NOLIM, NOLIM, NOLIM},
    {16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 288, 384, 512, 1024, 2048, NOLIM, NOLIM, NOLIM},
    // levels smer / mixed / bmer_unc (conv_lev24)
    {5, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {8, 13, 20, 30, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {15, 20, 30, 50, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {16, 24, 32, 48, 64, 128, 256, 512, 1024, 2048, 2080, 2112, 2176, 2240, 2304, 2432, 2560, 2816, 3072, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    // level bmer (conv_lev3)
    {5, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {8, 13, 20, 30, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {13, 20, 30, 50, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM},
    {18, 20, 25, 30, 40, 50, 60, 64, 68, 72, 76, 80, 84, 88, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM, NOLIM}};
// wave-uniform convert_count (code_ctx.cpp:15-239): all lanes pass the same c
FQ_DEV u64 conv_wave(u64 c, u32 level, u32 cl) {
  const u32 e = (level == LV_PMER ? 0u : level == LV_BMER ? 8u : 4u) + cl;
  const u32 D = CONV_D[e];
  const u64 f = (u64)cl << 5;
  if (c < D) return f + c;
#if FQ_WAVE > 1
  const u32 lim = FQ_LANE < 24 ? CONV_LIM[e][FQ_LANE] : NOLIM;
  const u32 cnt = popc64(wave_ballot((u64)lim <= c));
#else
  u32 cnt = 0;
  for (u32 i = 0; i < 24; ++i) cnt += (u64)CONV_LIM[e][i] <= c;
#endif
  return f + D + cnt;
}
FQ_DEV void sort_desc4(u32 d[4], const C4 &c) {  // sort_copy_stats, utils.cpp:109-126 (descending, stable)
  u32 a = c.c[0], b = c.c[1], e = c.c[2], f = c.c[3], t;
  // insertion network that only swaps on strict "greater", i.e. stable
  if (b > a) { t = a; a = b; b = t; }
  if (e > b) { t = b; b = e; e = t; if (b > a) { t = a; a = b; b = t; } }
  if (f > e) { t = e; e = f; f = t; if (e > b) { t = b; b = e; e = t; if (b > a) { t = a; a = b; b = t; } } }
  d[0] = a; d[1] = b; d[2] = e; d[3] = f;
}

#define SH_POS 0
#define SH_LEVEL 14
#define SH_C0 17
#define SH_C1 24
#define SH_C2 31
#define SH_C3 38
#define SH_RSYM 45
#define SH_LETMAX 49
#define SH_CORZ 52
#define EN_POS (0x3fffull << SH_POS)
#define EN_LEVEL (7ull << SH_LEVEL)
#define EN_CN(i) (0x7full << (SH_C0 + 7 * (i)))
#define EN_RSYM (0xfull << SH_RSYM)
#define EN_LETMAX (7ull << SH_LETMAX)
#define EN_CORZ (7ull << SH_CORZ)

FQ_DEV u32 let_max(const C4 &c, const u64 *sl) {  // let_max_element, code_ctx.cpp:327-338; also dna.cpp:335-340
  u32 r = 0;
  for (u32 i = 1; i < 4; ++i) {
    u32 ci = c4_get(c, i), cr = c4_get(c, r);
    if (ci > cr) r = i;
    else if (ci == cr && sl_get(sl, i) > sl_get(sl, r)) r = i;
  }
  return r;
}
// determine_ctx_codes, code_ctx.cpp:257-324
FQ_DEV void ctx_codes(u64 a[7], const DevCfg *cfg, const C4 &counts, const u64 *sl, u32 pos, u32 level, u32 cor_zone, u64 ctx_r_sym, u32 read_len) {
  u64 mask = ~0ull, ctx = 0;
  u32 plim = level == 0 ? 0u : level == 1 ? cfg->pmer : level == 2 ? cfg->smer : cfg->bmer;
  u32 srt[4];
  sort_desc4(srt, counts);
  a[0] = ctx | mask;
  mask ^= EN_LEVEL | EN_CN(0) | EN_CN(1) | EN_CN(2) | EN_CN(3) | EN_POS;
  ctx += (u64)level << SH_LEVEL;
  ctx += conv_count(srt[0], level, 1) << SH_C0;
  ctx += conv_count(srt[1], level, 1) << SH_C1;
  ctx += conv_count(srt[2], level, 0) << SH_C2;
  ctx += conv_count(srt[3], level, 0) << SH_C3;
  const bool in_lim = pos < plim, eor = pos + 5 >= read_len;
  const u64 eor_v = 0x3fffull - (u64)(u32)(read_len - pos);
  if (in_lim) ctx += (u64)pos << SH_POS;
  else if (eor) ctx += eor_v << SH_POS;
  else ctx += (u64)(plim + pos / 16) << SH_POS;
  a[1] = ctx | mask;
  mask ^= EN_CORZ | EN_RSYM;
  ctx += (u64)cor_zone << SH_CORZ;
  ctx += (u64)popc64(ctx_r_sym) << SH_RSYM;  // transform_r_sym, code_ctx.cpp:371-373
  a[2] = ctx | mask;
  ctx &= ~(EN_CN(0) | EN_CN(1));
  ctx += conv_count(srt[0], level, 2) << SH_C0;
  ctx += conv_count(srt[1], level, 2) << SH_C1;
  a[3] = ctx | mask;
  ctx &= ~(EN_CN(0) | EN_CN(1) | EN_CN(2) | EN_CN(3));
  ctx += conv_count(srt[0], level, 3) << SH_C0;
  ctx += conv_count(srt[1], level, 3) << SH_C1;
  ctx += conv_count(srt[2], level, 1) << SH_C2;
  ctx += conv_count(srt[3], level, 1) << SH_C3;
  a[4] = ctx | mask;
  mask ^= EN_LETMAX;
  ctx += (u64)let_max(counts, sl) << SH_LETMAX;
  a[5] = ctx | mask;
  ctx &= ~EN_POS;
  if (in_lim) ctx += ((u64)pos + (1u << 13)) << SH_POS;
  else if (eor) ctx += eor_v << SH_POS;
  else ctx += ((u64)plim + pos / 8 + (1u << 13)) << SH_POS;
  a[6] = ctx | mask;
}
// determine_ctx_codes for wave-uniform arguments (the commit loop): same keys, quantisers by ballot
FQ_DEV void ctx_codes_wave(u64 a[7], const DevCfg *cfg, const C4 &counts, const u64 *sl, u32 pos, u32 level, u32 cor_zone, u64 ctx_r_sym, u32 read_len) {
  u64 mask = ~0ull, ctx = 0;
  u32 plim = level == 0 ? 0u : level == 1 ? cfg->pmer : level == 2 ? cfg->smer : cfg->bmer;
  u32 srt[4];
  sort_desc4(srt, counts);
  a[0] = ctx | mask;
  mask ^= EN_LEVEL | EN_CN(0) | EN_CN(1) | EN_CN(2) | EN_CN(3) | EN_POS;
  ctx += (u64)level << SH_LEVEL;
  const u64 c01 = conv_wave(srt[0], level, 1), c11 = conv_wave(srt[1], level, 1);
  const u64 c21 = conv_wave(srt[2], level, 1), c31 = conv_wave(srt[3], level, 1);
  ctx += c01 << SH_C0;
  ctx += c11 << SH_C1;
  ctx += conv_wave(srt[2], level, 0) << SH_C2;
  ctx += conv_wave(srt[3], level, 0) << SH_C3;
  const bool in_lim = pos < plim, eor = pos + 5 >= read_len;
  const u64 eor_v = 0x3fffull - (u64)(u32)(read_len - pos);
  if (in_lim) ctx += (u64)pos << SH_POS;
  else if (eor) ctx += eor_v << SH_POS;
  else ctx += (u64)(plim + pos / 16) << SH_POS;
  a[1] = ctx | mask;
  mask ^= EN_CORZ | EN_RSYM;
  ctx += (u64)cor_zone << SH_CORZ;
  ctx += (u64)popc64(ctx_r_sym) << SH_RSYM;
  a[2] = ctx | mask;
  ctx &= ~(EN_CN(0) | EN_CN(1));
  ctx += conv_wave(srt[0], level, 2) << SH_C0;
  ctx += conv_wave(srt[1], level, 2) << SH_C1;
  a[3] = ctx | mask;
  ctx &= ~(EN_CN(0) | EN_CN(1) | EN_CN(2) | EN_CN(3));
  ctx += conv_wave(srt[0], level, 3) << SH_C0;
  ctx += conv_wave(srt[1], level, 3) << SH_C1;
  ctx += c21 << SH_C2;
  ctx += c31 << SH_C3;
  a[4] = ctx | mask;
  mask ^= EN_LETMAX;
  ctx += (u64)let_max(counts, sl) << SH_LETMAX;
  a[5] = ctx | mask;
  ctx &= ~EN_POS;
  if (in_lim) ctx += ((u64)pos + (1u << 13)) << SH_POS;
  else if (eor) ctx += eor_v << SH_POS;
  else ctx += ((u64)plim + pos / 8 + (1u << 13)) << SH_POS;
  a[6] = ctx | mask;
}
// determine_ctx_letters, code_ctx.cpp:465-490
FQ_DEV void ctx_letters_keys(u64 a[10], const DevCfg *cfg, u32 pos, u64 letters, u32 read_len) {
  u64 mask = ~0ull ^ 0x3fffull, ctx = 0;
  if (pos < cfg->pmer) ctx += (u64)pos;
  else if (pos + 5 > read_len) ctx += 0x3fffull - (u64)(u32)(read_len - pos);
  for (u32 i = 0; i < 10; ++i) {
    ctx += ((letters >> (4 * i)) & 7ull) << (14 + 3 * i);
    mask ^= 7ull << (14 + 3 * i);
    a[i] = ctx | mask;
  }
}

// templates (dna.cpp:106-117): letters {10,10,10,10,1}, codes {20,6,3,2,1}
#define TPL_CODES_Q2 (20ull | (6ull << 16) | (3ull << 32) | (2ull << 48))
#define TPL_CODES_Q3 1ull
#define TPL_CODES_TOT 32u
#define TPL_LET_Q2 (10ull | (10ull << 16) | (10ull << 32) | (10ull << 48))
#define TPL_LET_Q3 1ull
#define TPL_LET_TOT 41u

// ---------------------------------------------------------------------------------------
// mailboxes (my_*_to_add push_back, dna.cpp:657-660,818-852): append in push order; the owner is
// derived from the key by the partition kernels
FQ_DEV void mail_push(Wk &w, u32 kind, u64 x) {
  if (w.rec) {   // read-head wave (only ever pushes the prefix p-mer, both strands)
    FQ_SYNC();
    if (FQ_LANE == 0) { const u32 n = w.rec->n_p; w.rec->pmail[n & 1] = x; w.rec->n_p = n + 1; }
    FQ_SYNC();
    return;
  }
  const Mail &m = w.cfg->mail[kind];
  u32 c = w.mn[kind];
  if (c >= m.cap) { w.err = FQSX_ERR_MAIL_FULL; return; }
  m.list[(u64)w.tid * m.cap + c] = x;
  w.mn[kind] = c + 1;
  w.st[ST_MAIL] += 1;
  if (kind != MAIL_P) w.sm->pq_key[kind == MAIL_S ? 1 : 0][c & (FQSX_PQ - 1)] = x;
}

// ---------------------------------------------------------------------------------------
// the coder proper
FQ_DEV u32 rank_sym(const Wk &w, const C4 &counts, u32 sym) {  // rank(), dna.cpp:177-193
  if (sym == 4) return 4;
  u32 r = 0, cs = c4_get(counts, sym);
  u64 ss = sl_get(w.s_let, sym);
  for (u32 i = 0; i < 4; ++i) {
    u32 ci = c4_get(counts, i);
    u64 si = sl_get(w.s_let, i);
    if (ci != cs) r += cs < ci;
    else if (si != ss) r += ss < si;
    else r += sym > i;
  }
  return r;
}

FQ_DEV bool find_counts_p(Wk &w, C4 &counts) {  // find_counts_p, dna.cpp:210-226
  const DevCfg *cfg = w.cfg;
  if (w.pm.cur != cfg->gp.k) {
    Kmer t = w.pm;
    u64 sh = 2 * (u64)cfg->pmer - 2 * (u64)t.cur;
    for (u32 j = 0; j < 4; ++j) {
      km_replace_last(t, j);
      u64 idx = km_aligned_rc(t);
      counts.c[j] = (u32)siv_range_sum(w, idx << sh, (idx + 1) << sh);
    }
  } else
    siv_counts(w, km_aligned_dir(w.pm), counts);
  return c4_any(counts);
}
FQ_DEV bool rough_p(Wk &w, C4 &counts) {  // find_counts_rough_p, dna.cpp:229-254 (order-free sums)
  const DevCfg *cfg = w.cfg;
  u32 n = 4 * (cfg->pmer - 1);
  u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (u32 q = FQ_LANE; q < n; q += FQ_WAVE) {
    u32 i = q >> 2;
    u64 j = q & 3;
    u32 sh = 62 - 2 * i;
    u64 d = (w.pm.dir & ~(3ull << sh)) + (j << sh);
    u64 idx = d >> (64 - 2 * w.pm.cur);
    u64 word = cfg->siv[idx >> 5];
    u32 fs = 2 * (u32)((idx & 31) & ~3ull);
    a0 += (u32)((word >> fs) & 3);
    a1 += (u32)((word >> (fs + 2)) & 3);
    a2 += (u32)((word >> (fs + 4)) & 3);
    a3 += (u32)((word >> (fs + 6)) & 3);
  }
  counts.c[0] = wave_sum32(a0);
  counts.c[1] = wave_sum32(a1);
  counts.c[2] = wave_sum32(a2);
  counts.c[3] = wave_sum32(a3);
  w.st[ST_SIV_WORDS] += n;
  return c4_any(counts);
}
// find_counts_rough_s / _b, dna.cpp:257-330: Hamming-1 neighbourhood, probes batched, merges serial.
// Of the 4 substitutions per position one is the k-mer itself, and every caller has just seen that k-mer miss in
// this very table (find_counts probed it, or stage P did: the global tables do not change inside a kernel), so
// that probe cannot contribute and is not issued: 3(k-1) probes, a single wave-wide round for k <= 22.  The
// remaining probes keep the reference's (position, symbol) order.  Returns their number.
FQ_DEV u32 rough_probe(Wk &w, const KTab &t, const KGeom &g, const Kmer &can) {
  WgShared *sm = w.sm;
  const u32 n = 3 * (g.k - 1);
  FQ_SYNC();
  for (u32 q = FQ_LANE; q < n; q += FQ_WAVE) {
    const u32 i = q / 3, r3 = q - 3 * i;
    u32 sh = 62 - 2 * i;
    const u32 orig = (u32)((can.dir >> sh) & 3ull);
    const u64 j = r3 + (r3 >= orig ? 1u : 0u);
    u64 d = (can.dir & ~(3ull << sh)) + (j << sh);
    sh = 64 - 2 * g.k + 2 * i;
    u64 r = (can.rc & ~(3ull << sh)) + ((3 - j) << sh);
    bool nd = (d & g.kernel_mask) < (r & g.kernel_mask);
    sm->bk_key[q] = nd ? d : r;
    sm->bk_dir[q] = nd ? 1 : 0;
  }
  batch_scan(w, t, true, n);
  // the reference also issues the k-1 look-ups of the k-mer itself (a miss scans at least the empty slot)
  w.st[ST_GPROBE] += g.k - 1;
  w.st[ST_GSLOT] += g.k - 1;
  return n;
}
FQ_DEV bool rough_kt(Wk &w, const KTab &t, const KGeom &g, const Kmer &can, u32 rng, const Cinc &ci, C4 &counts) {
  WgShared *sm = w.sm;
  c4_zero(counts);
  const u32 n = rough_probe(w, t, g, can);
  u64 hm[4];
  batch_hit_mask(w, n, hm);
  for (u32 r = 0; r < 4; ++r)
    for (u64 mk = hm[r]; mk; mk &= mk - 1) {  // probes in order, only those that found something
      u32 q = r * 64 + ctz64(mk);
      // merges all four counters, zeros included (dna.cpp:324-326)
      counts.c[0] = cinc_merge(sm, rng, ci, counts.c[0], sm->bk_res[q][0]);
      counts.c[1] = cinc_merge(sm, rng, ci, counts.c[1], sm->bk_res[q][1]);
      counts.c[2] = cinc_merge(sm, rng, ci, counts.c[2], sm->bk_res[q][2]);
      counts.c[3] = cinc_merge(sm, rng, ci, counts.c[3], sm->bk_res[q][3]);
    }
  return c4_any(counts);
}

// The same sweep with its probes done ahead of time by the scout wave (slot r of the chunk): only the merges, which
// draw from the worker's RNG in probe order, remain.
FQ_DEV bool rough_merge_pre(Wk &w, u32 r, u32 j, const KGeom &g, u32 rng, const Cinc &ci, C4 &counts) {
  WgShared *sm = w.sm;
  c4_zero(counts);
  if (r == 0xfd) {   // the hits' counts added up by the scout wave (no merge of theirs can draw: scout_rough)
    const u64 v = w.sb->rc_val[j][0];
    counts.c[0] = (u32)(v & 0xffff); counts.c[1] = (u32)((v >> 16) & 0xffff); counts.c[2] = (u32)((v >> 32) & 0xffff); counts.c[3] = (u32)(v >> 48);
    w.st[ST_GPROBE] += 4 * (g.k - 1);
    w.st[ST_GSLOT] += w.sb->rc_ns[j] + g.k - 1;
    return c4_any(counts);
  }
  if (r == 0xfe) {   // compact form: the hits' counts are stored in probe order
    u32 x = 0;
    for (u64 mk = w.sb->rc_hit[j]; mk; mk &= mk - 1, ++x) {
      const u64 v = w.sb->rc_val[j][x];
      counts.c[0] = cinc_merge(sm, rng, ci, counts.c[0], (u32)(v & 0xffff));
      counts.c[1] = cinc_merge(sm, rng, ci, counts.c[1], (u32)((v >> 16) & 0xffff));
      counts.c[2] = cinc_merge(sm, rng, ci, counts.c[2], (u32)((v >> 32) & 0xffff));
      counts.c[3] = cinc_merge(sm, rng, ci, counts.c[3], (u32)(v >> 48));
    }
    w.st[ST_GPROBE] += 4 * (g.k - 1);
    w.st[ST_GSLOT] += w.sb->rc_ns[j] + g.k - 1;
    return c4_any(counts);
  }
  for (u64 mk = w.sb->rr_hit[r]; mk; mk &= mk - 1) {
    const u64 v = w.sb->rr_res[r][ctz64(mk)];
    counts.c[0] = cinc_merge(sm, rng, ci, counts.c[0], (u32)(v & 0xffff));
    counts.c[1] = cinc_merge(sm, rng, ci, counts.c[1], (u32)((v >> 16) & 0xffff));
    counts.c[2] = cinc_merge(sm, rng, ci, counts.c[2], (u32)((v >> 32) & 0xffff));
    counts.c[3] = cinc_merge(sm, rng, ci, counts.c[3], (u32)(v >> 48));
  }
  w.st[ST_GPROBE] += 4 * (g.k - 1);
  w.st[ST_GSLOT] += w.sb->rr_ns[r] + g.k - 1;
  return c4_any(counts);
}
#if FQ_WAVE > 1
// scout wave: the sweeps of the chunk's positions that will need one (global b-mer miss, cascade empty), FQSX_SW
// sweeps in flight at a time (one probe of each per lane).  A sweep with at most 3 hits is kept in compact form,
// fuller ones take one of the FQSX_RR full-size slots; whatever does not fit is left to the resolving wave.
FQ_DEV void scout_rough(Wk &w, u32 n) {
  const DevCfg *cfg = w.cfg;
  SpecBuf *sb = w.sb;
  const KGeom &g = cfg->gb;
  const u32 n3 = 3 * (g.k - 1);
  if (n3 > 2 * FQ_WAVE) return;
  // b-mers of up to 22 symbols: one probe of a sweep per lane, FQSX_SW sweeps in flight.  Longer b-mers (the large-genome
  // geometries: k = 24 .. 27, 69 .. 78 probes): two probes per lane -- the sweep takes two of the FQSX_SW probe groups in
  // flight -- and only the compact forms (at most 3 hits), which is what a sweep in covered sequence looks like; a fuller
  // sweep is left to the resolving wave.
  const u32 halves = n3 > FQ_WAVE ? 2u : 1u, per_round = FQSX_SW / halves;
  const u32 lane = FQ_LANE;
  const u64 lt = (1ull << lane) - 1ull;
  FQ_SYNC();
  const bool cand = lane < n && sb->sp_flag[lane] == 3 && (sb->sx_flag[lane] & (SX_VALID | SX_HITS)) == SX_VALID;
  u64 cm = wave_ballot(cand);
  u32 big = 0;
  while (cm) {
    if (lds_load_acq(&w.sm->sc_req_seq) != w.sc_epoch) break;   // nobody will look at this chunk any more
    // up to per_round candidate positions; probe group q belongs to sweep q / halves (all indices compile-time constants)
    u32 cj[FQSX_SW];
    bool co[FQSX_SW];
#pragma unroll
    for (int x = 0; x < (int)FQSX_SW; ++x) {
      co[x] = (u32)x < per_round && cm != 0;
      cj[x] = co[x] ? ctz64(cm) : 0u;
      if (co[x]) cm &= cm - 1;
    }
    u32 js[FQSX_SW];
    bool ok[FQSX_SW];
#pragma unroll
    for (int q = 0; q < (int)FQSX_SW; ++q) {
      ok[q] = halves == 2 ? co[q >> 1] : co[q];
      js[q] = halves == 2 ? cj[q >> 1] : cj[q];
    }
    TabG f[FQSX_SW];
    u64 key[FQSX_SW];
    bool nd[FQSX_SW], in[FQSX_SW];
#pragma unroll
    for (int q = 0; q < (int)FQSX_SW; ++q) {
      key[q] = 0; nd[q] = false; f[q].s = nullptr; f[q].a = f[q].b = 0;
#pragma unroll
      for (u32 x = 0; x < FQSX_BKT; ++x) { f[q].ia[x] = 0; f[q].ib[x] = 0; }
      const u32 pr = ((u32)q & (halves - 1u)) * FQ_WAVE + lane;   // this lane's probe of the group
      in[q] = ok[q] && pr < n3;
      if (in[q]) {
        const u32 pi = pr / 3, r3 = pr - 3 * pi;
        const u32 shd = 62 - 2 * pi, shr = 64 - 2 * g.k + 2 * pi;
        const u64 cdir = sb->sp_sdir[2][js[q]], crc = sb->sp_src[2][js[q]];
        const u32 orig = (u32)((cdir >> shd) & 3ull);
        const u64 sy = r3 + (r3 >= orig ? 1u : 0u);
        const u64 d = (cdir & ~(3ull << shd)) + (sy << shd);
        const u64 rr = (crc & ~(3ull << shr)) + ((3 - sy) << shr);
        nd[q] = (d & g.kernel_mask) < (rr & g.kernel_mask);
        key[q] = nd[q] ? d : rr;
        f[q] = tab_first_g(cfg->g_b, sb_owner(cfg, key[q]), key[q]);
      }
    }
    u64 res[FQSX_SW], hmv[FQSX_SW];
    u32 nsv[FQSX_SW];
#pragma unroll
    for (int q = 0; q < (int)FQSX_SW; ++q) {
      res[q] = 0; hmv[q] = 0; nsv[q] = 0;
      if (!ok[q]) continue;   // (uniform)
      u64 ns = 0;
      if (in[q]) {
        C4 c;
        c4_zero(c);
        tab_rest(cfg->g_b, f[q], key[q], nd[q], c, ns);
        res[q] = (u64)c.c[0] | ((u64)c.c[1] << 16) | ((u64)c.c[2] << 32) | ((u64)c.c[3] << 48);
      }
      hmv[q] = wave_ballot(res[q] != 0);
      nsv[q] = wave_sum32((u32)ns);
    }
#pragma unroll
    for (int s = 0; s < (int)FQSX_SW; ++s) {
      if (!co[s]) continue;   // (uniform)
      constexpr int kSW = (int)FQSX_SW;
      const int qa = 2 * s < kSW ? 2 * s : 0, qb = 2 * s + 1 < kSW ? 2 * s + 1 : 0;   // the sweep's groups when it takes two
      const u32 j = cj[s];
      const u64 hm0 = halves == 2 ? hmv[qa] : hmv[s], hm1 = halves == 2 ? hmv[qb] : 0ull;
      const u64 r0 = halves == 2 ? res[qa] : res[s], r1 = halves == 2 ? res[qb] : 0ull;
      const u32 nh0 = popc64(hm0), nh = nh0 + popc64(hm1);
      const u32 nsum = halves == 2 ? nsv[qa] + nsv[qb] : nsv[s];
      // Merging the hits' counts (Increment(a, b), utils.h:327-333) is plain addition, and draws nothing, as long as every
      // sum stays within the exact range of the counter code -- and the partial sums only grow, so what decides is the total
      // over ALL the sweep's hits, however many: then the sums stand in for the hits (round 4: rounds 2-3 only looked at
      // sweeps with at most three hits)
      bool summed = false;
      if (nh) {   // (uniform)
        const u64 sum = wave_sum64(r0 + r1);   // (a 16-bit field holds 128 x 63)
        const u32 thr = (CINC_B).thr;
        summed = (u32)(sum & 0xffff) <= thr && (u32)((sum >> 16) & 0xffff) <= thr && (u32)((sum >> 32) & 0xffff) <= thr && (u32)(sum >> 48) <= thr;
        if (summed && lane == 0) { sb->rc_val[j][0] = sum; sb->rc_hit[j] = nh >= 64 ? ~0ull : (1ull << nh) - 1ull; sb->rc_ns[j] = nsum; sb->rr_idx[j] = 0xfd; }
      }
      if (summed) continue;
      if (nh <= 3) {
        // the hits' counts in probe order (the first half's probes come first)
        if (r0 != 0) sb->rc_val[j][popc64(hm0 & lt)] = r0;
        if (r1 != 0) sb->rc_val[j][nh0 + popc64(hm1 & lt)] = r1;
        // (rc_hit is only ever counted: one bit per hit)
        if (lane == 0) { sb->rc_hit[j] = halves == 2 ? (1ull << nh) - 1ull : hm0; sb->rc_ns[j] = nsum; sb->rr_idx[j] = 0xfe; }
      } else if (halves == 1 && big < FQSX_RR) {
        if (lane < n3) sb->rr_res[big][lane] = r0;
        if (lane == 0) { sb->rr_hit[big] = hm0; sb->rr_ns[big] = nsum; sb->rr_idx[j] = (u8)big; }
        ++big;
      }
    }
    // everything below the next unswept position is final
    FQ_SYNC();
    lds_store_rel(&sb->rr_front, cm ? ctz64(cm) : FQSX_SPEC);
  }
  FQ_SYNC();
  lds_store_rel(&sb->rr_front, FQSX_SPEC);
}
// the first position of the chunk that still needs a sweep (scout wave, before the chunk is published)
FQ_DEV u32 scout_rough_first(Wk &w, u32 n) {
  SpecBuf *sb = w.sb;
  const u32 lane = FQ_LANE;
  if (3 * (w.cfg->gb.k - 1) > 2 * FQ_WAVE) return FQSX_SPEC;
  FQ_SYNC();
  const bool cand = lane < n && sb->sp_flag[lane] == 3 && (sb->sx_flag[lane] & (SX_VALID | SX_HITS)) == SX_VALID;
  const u64 cm = wave_ballot(cand);
  return cm ? ctz64(cm) : FQSX_SPEC;
}
// scout wave, first chunk of a read: the global look-ups find_counts will make at the positions whose b-mer is still
// partial -- all 4^m paddings of an almost full b-/s-mer (up to 16), or the one probe of a full s-mer -- in one round
FQ_DEV void scout_early(Wk &w, u32 n) {
  const DevCfg *cfg = w.cfg;
  SpecBuf *sb = w.sb;
  const u32 lane = FQ_LANE;
  const u32 bmargin = cfg->bmer - cfg->smer - 1, smargin = cfg->smer - cfg->pmer + 1;
  // groups of trials in (position, table) order; every lane derives the same list and picks its own trial
  u32 total = 0, my_e = ~0u, my_tb = 0, my_t = 0, my_m = 0;
  FQ_SYNC();
  for (u32 e = 0; e < n && e < 32; ++e) {
    const u32 cb = sb->sp_scur[2][e], cs = sb->sp_scur[1][e];
    if (cb == cfg->gb.k) break;   // from here on stage P probes the full b-mer itself
    for (u32 tb = 0; tb < 2; ++tb) {
      const u32 cur = tb ? cs : cb, k = tb ? cfg->gs.k : cfg->gb.k, margin = tb ? smargin : bmargin;
      if (cur + margin < k) continue;
      const u32 m = k - cur;
      if (m > 2) continue;
      const u32 cnt = 1u << (2 * m);
      if (total + cnt > 64) continue;
      if (lane >= total && lane < total + cnt) { my_e = e; my_tb = tb; my_t = lane - total; my_m = m; }
      if (lane == 0) sb->ep_off[e][tb] = (u8)total;
      total += cnt;
    }
  }
  if (my_e != ~0u) {
    const KGeom &g = my_tb ? cfg->gs : cfg->gb;
    const KTab &t = my_tb ? cfg->g_s : cfg->g_b;
    const u32 kx = my_tb ? 1u : 2u;
    const u64 kd = sb->sp_sdir[kx][my_e], kr = sb->sp_src[kx][my_e];
    const u32 cur = sb->sp_scur[kx][my_e];
    u64 d = my_m ? kd >> (2 * my_m) : kd, r = kr & (~0ull << (64 - 2 * cur));   // ht_kmer.h:291-310
    for (u32 j = 0; j < my_m; ++j) {
      const u64 sym = (my_t >> (2 * j)) & 3;
      d |= sym << (62 - 2 * j);
      r |= (3 - sym) << (64 - 2 * g.k + 2 * j);
    }
    const bool nd = (d & g.kernel_mask) < (r & g.kernel_mask);
    const u64 key = nd ? d : r;
    C4 c;
    c4_zero(c);
    u64 ns = 0;
    tab_scan(t, sb_owner(cfg, key), key, nd, c, ns);
    sb->ep_res[lane] = (u64)c.c[0] | ((u64)c.c[1] << 16) | ((u64)c.c[2] << 32) | ((u64)c.c[3] << 48);
    sb->ep_ns[lane] = (u8)(ns > 255 ? 255 : ns);
  }
  FQ_SYNC();
}
// scout wave, after scout_early: an early position whose first look-up of find_counts (dna.cpp:457-502: the global b-mer
// table if the b-mer is almost full, else the global s-mer table) finds something is settled here -- level, counts --
// when merging the trials' counts (ht_kmer.h:321-323) cannot draw from the RNG: every sum stays within the exact range
// of the counter code (Increment(a, b) of values <= thr is their sum, utils.h:272-290,327-333).  That is the rule for
// the s-mer table (exact up to 2047), i.e. for the positions before the b-mer is almost full.  The resolving wave then
// takes the position through its settled path (keys and rank in code_keys) instead of the per-position one.
FQ_DEV void scout_settle_early(Wk &w, u32 i0, u32 n) {
  const DevCfg *cfg = w.cfg;
  SpecBuf *sb = w.sb;
  const u32 e = FQ_LANE;
  const u32 bmargin = cfg->bmer - cfg->smer - 1, smargin = cfg->smer - cfg->pmer + 1;
  u32 np = 0, nsl = 0;
  FQ_SYNC();
  if (e < n && e < 32 && sb->sp_flag[e] == 0 && sb->sp_scur[2][e] != cfg->gb.k && sb->sp_nrun[e] < 2) {
    const u32 cb = sb->sp_scur[2][e], cs = sb->sp_scur[1][e];
    const u32 tb = cb + bmargin >= cfg->gb.k ? 0u : 1u;   // the table find_counts asks first
    const u32 cur = tb ? cs : cb, k = tb ? cfg->gs.k : cfg->gb.k;
    const u32 off = sb->ep_off[e][tb];
    if ((tb == 0 || cs + smargin >= k) && off != 0xff) {
      const u32 thr = tb ? (CINC_S).thr : (CINC_B).thr;
      const u32 m = k - cur, cnt = 1u << (2 * m);
      u32 c[4] = {0, 0, 0, 0}, nn = 0;
      for (u32 t = 0; t < cnt; ++t) {
        const u64 v = sb->ep_res[off + t];
        nn += sb->ep_ns[off + t];
        c[0] += (u32)(v & 0xffff); c[1] += (u32)((v >> 16) & 0xffff); c[2] += (u32)((v >> 32) & 0xffff); c[3] += (u32)(v >> 48);
      }
      bool ok = (c[0] | c[1] | c[2] | c[3]) != 0;
      if (m) ok = ok && c[0] <= thr && c[1] <= thr && c[2] <= thr && c[3] <= thr;
      if (!tb) ok = ok && (c[0] == 63) + (c[1] == 63) + (c[2] == 63) + (c[3] == 63) <= 1;   // (else the mixed level, dna.cpp:466-474)
      if (ok) {
        const u32 i = i0 + e;
        const int cor_dist = tb ? (int)cfg->smer : (int)cfg->bmer, d = (int)i - (int)w.cor_pos;
        const u32 cz = d < cor_dist ? (u32)(1 + 2 * (cor_dist - d) / cor_dist) : 0u;
        sb->sp_cq[e] = (u64)c[0] | ((u64)c[1] << 16) | ((u64)c[2] << 32) | ((u64)c[3] << 48);
        sb->sp_lvz[e] = (u8)((tb ? LV_SMER : LV_BMER) | (cz << 4));
        sb->sp_kind[e] = SK_RANK_PENDING;
        sb->sp_rep[e] = 0xff;     // (no repair while the b-mer is partial, dna.cpp:856)
        sb->sp_flag[e] = 1;
        const u32 pf = sb->pv_flag[e];
        if (pf & PV_PCAND) sb->pv_flag[e] = (u8)(pf | PV_P);   // (the p-mer entry is only hidden under a full b-mer, dna.cpp:822-830)
        np = cnt;
        nsl = nn;
      } else if (!tb && (c[0] | c[1] | c[2] | c[3]) != 0)
        sb->sp_flag[e] = 4;   // the look-up hits, but its merges draw: they are all that is left to the resolving wave
    }
  }
  np = wave_sum32(np); nsl = wave_sum32(nsl);
  FQ_SYNC();
  if (FQ_LANE == 0) { sb->h_np += np; sb->h_ns += nsl; }
  FQ_SYNC();
}
#endif

// find_counts, dna.cpp:457-502.  b_miss_known: the speculation stage already probed the global
// b-mer table for exactly this (full) b-mer and found nothing.
// ep: chunk position whose global look-ups the scout wave may have probed ahead (w's k-mers are that position's)
FQ_DEV u32 find_counts(Wk &w, C4 &counts, bool b_miss_known, u32 ep = ~0u) {
  const DevCfg *cfg = w.cfg;
  c4_zero(counts);
  u32 bmargin = cfg->bmer - cfg->smer - 1;
  u32 smargin = cfg->smer - cfg->pmer + 1;
  const u32 pre_b = ep != ~0u ? w.sb->ep_off[ep][0] : 0xffu, pre_s = ep != ~0u ? w.sb->ep_off[ep][1] : 0xffu;
  if (km_almost_full(w.bm, cfg->gb, bmargin)) {
    if (!b_miss_known && kt_find(w, cfg->g_b, true, cfg->gb, w.bm, RNG_B, CINC_B, counts, pre_b)) {
      u32 sat = (counts.c[0] == 63) + (counts.c[1] == 63) + (counts.c[2] == 63) + (counts.c[3] == 63);
      if (sat > 1) {
        C4 c2;
        kt_find(w, cfg->g_s, true, cfg->gs, w.sm_, RNG_S, CINC_S, c2);
        counts.c[0] += c2.c[0]; counts.c[1] += c2.c[1]; counts.c[2] += c2.c[2]; counts.c[3] += c2.c[3];
        return LV_MIXED;
      }
      return LV_BMER;
    } else {
      lq_sync_for(w, MAIL_B, cfg->gb, w.bm);
      if (kt_find(w, cfg->l_b, false, cfg->gb, w.bm, RNG_LB, CINC_B, counts)) return LV_BMER;
      if (w.bm.dir != w.bm_u.dir && kt_find(w, cfg->g_b, true, cfg->gb, w.bm_u, RNG_B, CINC_B, counts)) return LV_BMER_UNC;
    }
  }
  if (km_almost_full(w.sm_, cfg->gs, smargin)) {
    if (kt_find(w, cfg->g_s, true, cfg->gs, w.sm_, RNG_S, CINC_S, counts, pre_s)) return LV_SMER;
    lq_sync_for(w, MAIL_S, cfg->gs, w.sm_);
    if (kt_find(w, cfg->l_s, false, cfg->gs, w.sm_, RNG_LS, CINC_S, counts)) return LV_SMER;
  } else if (find_counts_p(w, counts))
    return LV_PMER;
  return LV_NONE;
}

FQ_DEV bool repair_existing(Wk &w, u32 pos, const C4 &counts, u32 sym) {  // repair_kmers_existing, dna.cpp:333-370
  u32 mx = let_max(counts, w.s_let);
  if (sym != 4) {
    if (mx == sym) return false;
    if (c4_get(counts, sym) != 0) return false;
    if (c4_get(counts, mx) <= 3) return false;
  }
  km_replace_last(w.pm, mx);
  km_replace_last(w.sm_, mx);
  km_replace_last(w.bm, mx);
  w.cor_pos = pos;
  return true;
}
FQ_DEV bool repair_missing(Wk &w, u32 pos) {  // repair_kmers_missing, dna.cpp:374-454
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  if (!w.repm_gate) return false;  // siv_pmer->avg_filling_factor() < 7.0, dna.cpp:376
  // 5 positions x 4 symbols; the entries equal to the current symbol are skipped
  FQ_SYNC();
  for (u32 q = FQ_LANE; q < 20; q += FQ_WAVE) {
    u32 j = 1 + (q >> 2);
    u64 c = q & 3;
    Kmer t = w.bm;
    km_replace(t, c, t.cur - 1 - j);
    sm->bk_key[q] = km_norm(t, cfg->gb);
  }
  batch_count(w, cfg->g_b, 20);
  int best_c = 4, best_count = 0, best_pos = 0;
  for (u32 q = 0; q < 20; ++q) {
    u32 j = 1 + (q >> 2), c = q & 3;
    if (km_symbol(w.bm, w.bm.cur - 1 - j) == c) continue;
    int cnt = (int)sm->bk_res[q][0];
    if (cnt >= best_count && cnt >= 2) { best_c = (int)c; best_count = cnt; best_pos = (int)j; }
  }
  w.st[ST_GPROBE] -= 5;  // the 5 skipped look-ups are not part of the algorithm
  if (best_pos) {
    km_replace(w.bm, (u64)best_c, w.bm.cur - 1 - best_pos);
    if (best_pos < (int)w.sm_.cur) km_replace(w.sm_, (u64)best_c, w.sm_.cur - 1 - best_pos);
    if (best_pos < (int)w.pm.cur) km_replace(w.pm, (u64)best_c, w.pm.cur - 1 - best_pos);
    u32 np = pos - (u32)best_pos;
    w.cor_pos = w.cor_pos > np ? w.cor_pos : np;
    return true;
  }
  return false;
}

FQ_DEV void code_letter(Wk &w, u32 pos, u32 sym, u32 read_len) {  // dna.cpp:520-528,776-785
  u64 lev[10];
  ctx_letters_keys(lev, w.cfg, pos, w.ctx_letters, read_len);
  if (w.piped) {   // the coder wave owns the context models
    if (!cq_wait_space(w, 1)) return;
    const u32 e = w.cq_tail & (FQSX_CQ - 1);
    FQ_SYNC();
    if (FQ_LANE == 0) {
      for (u32 l = 0; l < 10; ++l) w.sm->cq_key[l][e] = lev[l];
      w.sm->cq_kind[e] = SK_LETTER;
      w.sm->cq_rsym[e] = (u8)sym;
    }
    FQ_SYNC();
    cq_publish(w, 1);
    return;
  }
  FQ_SYNC();
  for (u32 l = 0; l < 10; ++l) w.sm->lev_tmp[l] = lev[l];
  FQ_SYNC();
  Slot4 s;
  u32 idx = find_leveled(w, 2, w.sm->lev_tmp, 1, 0, 9, w.avg_letters, TPL_LET_Q2, TPL_LET_Q3, TPL_LET_TOT, s);
  if (idx != FQSX_NIL) slot_encode(w, idx, s, sym);
}

FQ_DEV u16 *small_base(Wk &w) { return w.cfg->small + (u64)w.tid * SM_TOTAL_U16; }

FQ_DEV void push_p_both(Wk &w) {
  mail_push(w, MAIL_P, km_aligned_dir(w.pm));
  mail_push(w, MAIL_P, km_aligned_rc(w.pm));
}
FQ_DEV void push_b_local(Wk &w) {
  u64 x = km_norm(w.bm, w.cfg->gb);
  mail_push(w, MAIL_B, x);
}

FQ_DEV void insert_all(Wk &w, u64 sym) {
  const DevCfg *cfg = w.cfg;
  km_insert(w.pm, cfg->gp, sym); km_insert(w.sm_, cfg->gs, sym); km_insert(w.bm, cfg->gb, sym);
  km_insert(w.pm_u, cfg->gp, sym); km_insert(w.sm_u, cfg->gs, sym); km_insert(w.bm_u, cfg->gb, sym);
}

FQ_DEV u32 rd_sym(Wk &w, const u8 *p, u32 i, u32 size) { return size <= FQSX_RD_LDS ? (u32)w.rdp[i] : dna_code(p[i]); }

FQ_DEV void prefix_direct(Wk &w, const u8 *p, u32 size) {  // compress_prefix_direct, dna.cpp:506-546
  w.ctx_letters = ~0ull;
  for (u32 i = 0; i < w.cfg->prefix; ++i) {
    u32 sym = rd_sym(w, p, i, size);
    code_letter(w, i, sym, 0);
    w.ctx_letters = (w.ctx_letters << 4) + sym;
    if (sym == 4) { sym = 0; w.cor_pos = i; }
    insert_all(w, sym);
  }
}

FQ_DEV void scout_restart(Wk &w, u32 read, u32 i0, const u64 s_let[4], u32 flags, bool hold);
FQ_DEV void prefix_sorted(Wk &w, const u8 *p, u32 size) {  // compress_prefix_sorted, dna.cpp:549-661
  const DevCfg *cfg = w.cfg;
  WState *ws = w.ws;
  u16 *sb = small_base(w);
  w.ctx_letters = ~0ull;
  bool was_N = false;
  for (u32 i = 0; i < cfg->pmer; ++i) {
    u32 sym = rd_sym(w, p, i, size);
    w.ctx_letters = (w.ctx_letters << 4) + sym;
    if (sym == 4) { sym = 3; was_N = true; w.N_run++; } else w.N_run = 0;
    insert_all(w, sym);
  }
  // Read-head wave: what the scout waves need of this read -- its codes (staged by read_head), the k-mers after the
  // prefix -- is complete here; they start on the read's chunks while this wave ranks and codes the prefix.
  if (w.rec) {
    HeadRec *rec = w.rec;
    FQ_SYNC();
    if (FQ_LANE == 0) {
      rec->same = 0;
      rec->n_run = w.N_run;
      rec->kdir[0] = w.pm.dir; rec->krc[0] = w.pm.rc; rec->kcur[0] = w.pm.cur;
      rec->kdir[1] = w.sm_.dir; rec->krc[1] = w.sm_.rc; rec->kcur[1] = w.sm_.cur;
      rec->kdir[2] = w.bm.dir; rec->krc[2] = w.bm.rc; rec->kcur[2] = w.bm.cur;
      rec->idx = w.rec_idx;
    }
    FQ_SYNC();
    lds_store_rel(&w.sm->hd_early, w.rec_idx + 1);
  }
  // Request mode (no read-head wave: paired-end kernels): the k-mers after the prefix are final here, so the scout waves
  // can be sent off to the suffix's first chunks now, while this wave still ranks and codes the prefix (suffix() then
  // finds the request posted).
  if (w.scout && w.sc_reqmode && cfg->pmer < size) {
    w.rq_p = p; w.rq_size = size; w.rq_rev = false;
    w.sc_read += 1;
    w.sc_abandoned = false;
    scout_restart(w, w.sc_read, cfg->pmer, w.s_let, 0, false);
    w.rq_early = true;
  }
  sm_encode(w, sb + SM_OFF_NS, SM_NS_N, 1u << 12, was_N ? 1u : 0u);
  u64 psf = ((ws->ctx_ps_flags << 1) + (was_N ? 1u : 0u)) & 0xffff;
  u64 flag;
  const u64 cur_idx = km_aligned_dir(w.pm);
  if (w.pm.dir == ws->pmer_prev_dir) flag = 4;
  else { flag = siv_test(cfg, cur_idx); w.st[ST_SIV_WORDS] += 1; }
  sm_encode(w, sb + SM_OFF_PSF + psf * (SM_PSF_N + 1), SM_PSF_N, 1u << 12, (u32)flag);
  psf = ((psf << 3) + flag) & 0xffff;
  ws->ctx_ps_flags = psf;
  if (flag < 4) {
    u64 prev_idx = ws->pmer_prev_cur ? ws->pmer_prev_dir >> (64 - 2 * ws->pmer_prev_cur) : 0;
    u64 dif = siv_count_equal(w, prev_idx, cur_idx, flag);
    u32 nb = 1;
    for (u64 x = dif >> 8; x; x >>= 8) ++nb;  // no_bytes, utils.h:164-174
    sm_encode(w, sb + SM_OFF_PSNB + psf * 6u, cfg->ps_nobytes_n, 1u << 12, nb - 1);
    if (nb == 1) {
      u32 hi = (u32)(dif >> 4), lo = (u32)(dif & 0xf);
      sm_encode(w, sb + SM_OFF_NIB + (u32)flag * (SM_NIB_N + 1), SM_NIB_N, 1u << 15, hi);
      sm_encode(w, sb + SM_OFF_NIB + (4u + (u32)flag * 16u + hi) * (SM_NIB_N + 1), SM_NIB_N, 1u << 15, lo);
    } else {
      u32 hi_byte = (u32)(dif >> (nb * 8 - 8));
      u32 e = (u32)flag * 4u + (nb - 2);
      u8 *bi = cfg->byte_init + (u64)w.tid * SM_LAZY_ENTRIES;
      sm_encode256(w, sb + SM_OFF_BYTE + (u64)e * (SM_BYTE_N + 1), bi + e, hi_byte);
      for (u32 i = 0; i + 1 < nb; ++i) {
        u32 e2 = 16u + ((e * 256u + hi_byte) * 4u + i);
        sm_encode256(w, sb + SM_OFF_BYTE + (u64)e2 * (SM_BYTE_N + 1), bi + e2, (u32)(dif & 0xff));
        dif >>= 8;
      }
    }
  }
  if (was_N)
    for (u32 i = 0; i < cfg->pmer; ++i) {
      u8 ch = p[i];
      if (ch == 'T' || ch == 'N') sm_encode(w, sb + SM_OFF_NS + (i + 1) * (SM_NS_N + 1), SM_NS_N, 1u << 12, ch == 'N' ? 1u : 0u);
    }
  ws->pmer_prev_dir = w.pm.dir;
  ws->pmer_prev_rc = w.pm.rc;
  ws->pmer_prev_cur = w.pm.cur;
  push_p_both(w);
}

// Could a local insert that stage P's probe may not have seen change a look-up of full k-mer `km`?  These are the
// list entries from pq_lo on (LDS mirror) + the entries of chunk positions [q_done, j) that stage Q has not appended yet.
FQ_DEV bool pend_conflict(Wk &w, u32 qi, const KGeom &g, const Kmer &km, u32 q_done, u32 j) {
  WgShared *sm = w.sm;
  const u32 lo = w.pq_lo[qi], hi = w.mn[qi ? MAIL_S : MAIL_B];
  if (hi - lo > FQSX_PQ) return true;
  if (pq_group_hit(w, qi, g, km, lo, hi)) return true;
  const u32 k2 = 2 * g.k;
  const bool nd = km_norm_dir(km, g);
  const u64 v = (nd ? km.dir : km.rc) >> (64 - k2);
  const u64 lowmask = (1ull << (k2 - 2)) - 1ull;
  const u64 grp = nd ? (v >> 2) : (v & lowmask);
  bool hit = false;
  for (u32 t = q_done + FQ_LANE; t < j; t += FQ_WAVE)
    if (w.sb->pv_flag[t] & (qi ? PV_S : PV_B)) {
      u64 pv = (qi ? w.sb->pv_s[t] : w.sb->pv_b[t]) >> (64 - k2);
      hit |= nd ? ((pv >> 2) == grp) : ((pv & lowmask) == grp);
    }
  return wave_any(hit);
}
// repair_kmers_existing decision (dna.cpp:333-360): symbol to substitute or 0xff
FQ_DEV u32 repair_decide(const Wk &w, const C4 &c, u32 sym) {
  u32 mx = let_max(c, w.s_let);
  if (sym == 4 || (mx != sym && c4_get(c, sym) == 0 && c4_get(c, mx) > 3)) return mx;
  return 0xff;
}

// Stage P of one chunk of <= 64 suffix positions (one per lane).  Lane j rolls the six k-mers
// forward to position i0+j, probes the global b-mer table and -- on a plain hit -- derives everything
// that depends only on (counts, position, symbol): the 7 context keys, the symbol's rank, the
// repair decision; it also prepares the position's mailbox entries.
// The k-mers in w are the state before position i0 - joff (joff = 0: before the chunk itself; the scout wave starts
// every chunk of a read from the state after the read's prefix).
// Returns false if the wave gave the chunk up because a restart request came in (scout waves only).
// lane0 != 0: a second pass over part of a chunk (scout_fix): the results go to lanes lane0 .. lane0 + n - 1, the chunk's
// header (snapshot of the local lists, sweep bookkeeping) stands and the probe counts are added to it.
// DEFER (scout waves, n <= 64: one position per lane): nothing is written to w.sb -- the lane's results come back in *Lout,
// the chunk's header values in *Hout, and spec_commit() stores them once the ring slot is free: a scout wave runs stage P
// of its next chunk while that chunk's slot is still in use, which makes the ring one chunk per scout deeper without any LDS.
struct SpecLane {
  u64 sdir[6], src[6], key[7], pv_b, pv_s, pv_pd, pv_pr, sx_s, sx_ls;
  u32 sx_lb;
  u8 scur[6], nrun, kind, flag, rsym, rep, pv_flag, sx_flag, valid;
};
struct SpecHead { u32 lo0, lo1, np, nlp; u64 ns, nls; };
FQ_DEV void spec_store_lane(SpecBuf *sb, u32 jj, const SpecLane &L) {
  for (u32 x = 0; x < 6; ++x) { sb->sp_sdir[x][jj] = L.sdir[x]; sb->sp_src[x][jj] = L.src[x]; sb->sp_scur[x][jj] = L.scur[x]; }
  sb->sp_nrun[jj] = L.nrun;
  if (L.flag == 1) {
    for (u32 l = 0; l < 7; ++l) sb->sp_key[l][jj] = L.key[l];
    sb->sp_rsym[jj] = L.rsym;
  }
  if (L.flag == 3) { sb->sx_lb[jj] = L.sx_lb; sb->sx_s[jj] = L.sx_s; sb->sx_ls[jj] = L.sx_ls; sb->sx_flag[jj] = L.sx_flag; }
  sb->sp_flag[jj] = L.flag; sb->sp_kind[jj] = L.kind; sb->sp_rep[jj] = L.rep;
  sb->pv_b[jj] = L.pv_b; sb->pv_s[jj] = L.pv_s; sb->pv_pd[jj] = L.pv_pd; sb->pv_pr[jj] = L.pv_pr; sb->pv_flag[jj] = L.pv_flag;
}
FQ_DEV void spec_store_head(SpecBuf *sb, u32 lo0, u32 lo1) {
  if (FQ_LANE == 0) { sb->h_pq_lo[0] = lo0; sb->h_pq_lo[1] = lo1; }
  for (u32 j = FQ_LANE; j < FQSX_SPEC; j += FQ_WAVE) { sb->rr_idx[j] = 0xff; sb->ep_off[j][0] = 0xff; sb->ep_off[j][1] = 0xff; }
  if (FQ_LANE == 0) { sb->rr_front = FQSX_SPEC; sb->h_fix_lane = 0xff; sb->h_fix_end = 0; }
}
// the deferred chunk into its buffer (scout waves)
FQ_DEV void spec_commit(Wk &w, const SpecLane &L, const SpecHead &H) {
  FQ_SYNC();
  spec_store_head(w.sb, H.lo0, H.lo1);
  if (L.valid) spec_store_lane(w.sb, FQ_LANE, L);
  if (FQ_LANE == 0) { w.sb->h_np = H.np; w.sb->h_nlp = H.nlp; w.sb->h_ns = H.ns; w.sb->h_nls = H.nls; }
  FQ_SYNC();
}
template <bool DEFER>
FQ_DEV bool speculate_t(Wk &w, const u8 *p, u32 size, u32 i0, u32 n, bool reversed, u32 joff, u32 lane0, SpecLane *Lout, SpecHead *Hout) {
  const DevCfg *cfg = w.cfg;
  u64 ns = 0, nls = 0;
  u32 np = 0, nlp = 0;
  const u32 lo0 = lq_done_now(w, 0), lo1 = lq_done_now(w, 1);   // everything below is in the local tables before the probes start
  FQ_SYNC();
  if (!DEFER && lane0 == 0) spec_store_head(w.sb, lo0, lo1);
  if (DEFER) Lout->valid = 0;
  const u32 b0 = i0 - joff;   // position the k-mers in w stand before
  bool gave_up = false;
  for (u32 j = FQ_LANE; j < n; j += FQ_WAVE) {
    const u32 jj = lane0 + j;   // the lane's place in the chunk
    SpecLane O;
    O.valid = 1; O.sx_lb = 0; O.sx_s = 0; O.sx_ls = 0; O.sx_flag = 0; O.rsym = 0;
    for (u32 l = 0; l < 7; ++l) O.key[l] = 0;
    TM_BEGIN(t_roll);
    // roll the six k-mers J symbols forward in closed form: only the last min(J, k) new symbols matter
    u64 fw = 0, rv = 0;   // new symbols packed oldest-first (fw) and complemented newest-first (rv)
    const u32 J = j + joff;
    const u32 L = J < 27 ? J : 27;
    u32 nrun = 0;
    bool windowed = false;
#if FQ_WAVE > 1
    if (size <= FQSX_RD_LDS) {
      // the read's codes are staged in LDS, one byte per base: the 28 bases before this position come in as eight
      // aligned words, each squeezed to 4 x 2 bits (and 4 N flags) by one multiplication
      const u32 i = i0 + j, start = i >= 28 ? i - 28 : 0, a = start & ~3u, sh = a + 32 - i;   // base x sits at pair a + 31 - x
      const u32 *wp = (const u32 *)(w.rdp + a);
      u64 P = 0;
      u32 N32 = 0;
#pragma unroll
      for (u32 q = 0; q < 8; ++q) {
        const u32 x = wp[q];
        P = (P << 8) | (((x & 0x03030303u) * 0x40100401u) >> 24);
        N32 = (N32 << 4) | (((((x >> 2) & 0x01010101u) * 0x08040201u) >> 24) & 15u);
      }
      if (L) {
        fw = (P >> (2 * sh)) & ((1ull << (2 * L)) - 1ull);
        rv = ~pairrev64(fw) >> (64 - 2 * L);
      }
      const u32 win = 32 - sh;                                     // bases a .. i-1 are in the window
      const u32 runN = sh < 32 ? ctz64(~(u64)(N32 >> sh)) : 0;     // N run ending at base i-1, as far as the window shows
      if (runN >= J) { nrun = J + w.N_run; windowed = true; }
      else if (runN < win) { nrun = runN; windowed = true; }
      else fw = rv = 0;                                            // (a run of >= 28 N: counted the long way below)
    }
#endif
    if (!windowed) {
      for (u32 s = 0; s < L; ++s) {
        u32 c = rd_sym(w, p, i0 + j - L + s, size);
        u64 ck = c == 4 ? 0 : c;
        fw = (fw << 2) | ck;
        rv |= (3 - ck) << (2 * s);
      }
      u32 s = J;
      while (s > 0 && rd_sym(w, p, b0 + s - 1, size) == 4) { --s; ++nrun; }
      if (s == 0) nrun += w.N_run;
    }
    Kmer pm = km_roll(w.pm, cfg->gp, J, fw, rv, L), sk = km_roll(w.sm_, cfg->gs, J, fw, rv, L), bm = km_roll(w.bm, cfg->gb, J, fw, rv, L);
    Kmer pu = km_roll(w.pm_u, cfg->gp, J, fw, rv, L), su = km_roll(w.sm_u, cfg->gs, J, fw, rv, L), bu = km_roll(w.bm_u, cfg->gb, J, fw, rv, L);
    km_insert_zero(pm, cfg->gp); km_insert_zero(sk, cfg->gs); km_insert_zero(bm, cfg->gb);
    km_insert_zero(pu, cfg->gp); km_insert_zero(su, cfg->gs); km_insert_zero(bu, cfg->gb);
    O.sdir[0] = pm.dir; O.src[0] = pm.rc; O.scur[0] = (u8)pm.cur;
    O.sdir[1] = sk.dir; O.src[1] = sk.rc; O.scur[1] = (u8)sk.cur;
    O.sdir[2] = bm.dir; O.src[2] = bm.rc; O.scur[2] = (u8)bm.cur;
    O.sdir[3] = pu.dir; O.src[3] = pu.rc; O.scur[3] = (u8)pu.cur;
    O.sdir[4] = su.dir; O.src[4] = su.rc; O.scur[4] = (u8)su.cur;
    O.sdir[5] = bu.dir; O.src[5] = bu.rc; O.scur[5] = (u8)bu.cur;
    O.nrun = (u8)(nrun > 255 ? 255 : nrun);
    const u32 i = i0 + j, sym = rd_sym(w, p, i, size);
    const u64 symk = sym == 4 ? 0 : sym;
    u32 flag = 0, rep = 0xff;
    C4 c;
    c4_zero(c);
    const bool b_full = bm.cur == cfg->gb.k;
    if (w.sc_poll && lds_load_acq(&w.sm->sc_req_seq) != w.sc_epoch) { gave_up = true; break; }   // (the same answer in every lane that asks)
    TM_END_SP(w, TM_SP_ROLL, t_roll);
    TM_BEGIN(t_probe);
    if (b_full) {
      bool nd = km_norm_dir(bm, cfg->gb);
      u64 key = nd ? bm.dir : bm.rc;
      // the rest of find_counts' cascade after a global b-mer miss (dna.cpp:478-499) -- local b, global s, local s
      // -- applies in the common case of a full s-mer and no pending correction; its probes are independent of
      // each other, so their first round trips are issued together with the b-mer's instead of one after another
      // ... and, within a b-mer's length after a correction (corrected != uncorrected k-mers), with the look-up of the
      // uncorrected b-mer that find_counts makes between the local b-mer table and the s-mer tables (dna.cpp:484-488)
      const bool casc = sk.cur == cfg->gs.k;
      const bool unc = bm.dir != bu.dir;
      const bool nds = km_norm_dir(sk, cfg->gs);
      const u64 ks = nds ? sk.dir : sk.rc;
      const bool ndu = km_norm_dir(bu, cfg->gb);
      const u64 ku = ndu ? bu.dir : bu.rc;
      const TabG fb = tab_first_g(cfg->g_b, sb_owner(cfg, key), key);
      TabG fs = fb, fu = fb;
      TabL flb, fls;
      flb.s = nullptr; flb.a = 0; fls.s = nullptr; fls.a = 0;
#pragma unroll
      for (u32 x = 0; x < FQSX_BKT; ++x) { flb.it[x] = 0; fls.it[x] = 0; }
      if (casc && !unc) {   // (after a correction the corrected b-mer mostly hits: those lanes start the cascade only on a miss)
        flb = tab_first_l(cfg->l_b, w.tid, key);
        fs = tab_first_g(cfg->g_s, sb_owner(cfg, ks), ks);
        fls = tab_first_l(cfg->l_s, w.tid, ks);
      }
      tab_rest(cfg->g_b, fb, key, nd, c, ns);
      ++np;
      TM_END_SP(w, TM_SP_PROBE, t_probe);
      TM_BEGIN(t_hit);
      if (c4_any(c)) {
        u32 sat = (c.c[0] == 63) + (c.c[1] == 63) + (c.c[2] == 63) + (c.c[3] == 63);
        if (sat <= 1 && nrun < 2) {
          flag = 1;
          int cor_dist = (int)cfg->bmer, d = (int)i - (int)w.cor_pos;
          u32 cz = d < cor_dist ? (u32)(1 + 2 * (cor_dist - d) / cor_dist) : 0u;
          u64 lev[7];
          if (!reversed) ctx_codes(lev, cfg, c, w.s_let, i, LV_BMER, cz, 0, size);
          else ctx_codes(lev, cfg, c, w.s_let, size - i - 1, LV_BMER, cz, 0, ~0u);  // dna.cpp:750-752
          for (u32 l = 0; l < 7; ++l) O.key[l] = lev[l];
          O.rsym = (u8)rank_sym(w, c, sym);
          // repair_kmers_existing decision (dna.cpp:333-360)
          rep = repair_decide(w, c, sym);
        }
        TM_END_SP(w, TM_SP_HIT, t_hit);
      } else {
        // global b-mer miss; the local probes see the tables as of the last flush
        flag = 3;
        u32 xf = 0;
        if (casc) {
          xf = SX_VALID;
          if (unc) {   // the four first round trips together, now that they are needed
            flb = tab_first_l(cfg->l_b, w.tid, key);
            fu = tab_first_g(cfg->g_b, sb_owner(cfg, ku), ku);
            fs = tab_first_g(cfg->g_s, sb_owner(cfg, ks), ks);
            fls = tab_first_l(cfg->l_s, w.tid, ks);
          }
          C4 l;
          c4_zero(l);
          tab_rest(cfg->l_b, flb, key, nd, l, nls);
          ++nlp;
          if (c4_any(l)) {
            xf |= SX_LB;
            O.sx_lb = l.c[0] | (l.c[1] << 8) | (l.c[2] << 16) | (l.c[3] << 24);
          } else if (unc && (tab_rest(cfg->g_b, fu, ku, ndu, l, ns), ++np, c4_any(l))) {
            xf |= SX_UNC;
            O.sx_s = (u64)l.c[0] | ((u64)l.c[1] << 16) | ((u64)l.c[2] << 32) | ((u64)l.c[3] << 48);
          } else {
            c4_zero(l);
            tab_rest(cfg->g_s, fs, ks, nds, l, ns);
            ++np;
            if (c4_any(l)) {
              xf |= SX_S;
              O.sx_s = (u64)l.c[0] | ((u64)l.c[1] << 16) | ((u64)l.c[2] << 32) | ((u64)l.c[3] << 48);
            } else {
              tab_rest(cfg->l_s, fls, ks, nds, l, nls);
              ++nlp;
              if (c4_any(l)) {
                xf |= SX_LS;
                O.sx_ls = (u64)l.c[0] | ((u64)l.c[1] << 16) | ((u64)l.c[2] << 32) | ((u64)l.c[3] << 48);
              }
            }
          }
        }
        O.sx_flag = (u8)xf;
        TM_END_SP(w, TM_SP_MISS, t_hit);
      }
    }
    O.flag = (u8)flag;
    O.kind = flag == 1 ? SK_RANK : SK_NONE;
    O.rep = (u8)rep;
    // mailbox entries of this position (dna.cpp:818-852), k-mers after replace_last(sym)
    km_replace_last(pm, symk); km_replace_last(sk, symk); km_replace_last(bm, symk);
    u32 pf = 0;
    if (sym < 4) {
      if (b_full) pf |= PV_B;
      if (sk.cur == cfg->gs.k) pf |= PV_S;
      if (pm.cur == cfg->gp.k && i - w.cor_pos >= cfg->pmer - 1) {
        pf |= PV_PCAND;
        if (flag == 1) pf |= (b_full && c4_get(c, sym) >= 3) ? PV_PHID : PV_P;
      }
    }
    O.pv_b = km_norm(bm, cfg->gb);
    O.pv_s = km_norm(sk, cfg->gs);
    O.pv_pd = km_aligned_dir(pm);
    O.pv_pr = km_aligned_rc(pm);
    O.pv_flag = (u8)pf;
    if (DEFER) *Lout = O; else spec_store_lane(w.sb, jj, O);
  }
  if (wave_any(gave_up)) return false;   // (decided for the whole wave: the lanes beyond the chunk did not ask)
  FQ_SYNC();
  np = wave_sum32(np); nlp = wave_sum32(nlp);
  ns = wave_sum64(ns); nls = wave_sum64(nls);
  if (DEFER) { Hout->lo0 = lo0; Hout->lo1 = lo1; Hout->np = np; Hout->nlp = nlp; Hout->ns = ns; Hout->nls = nls; }
  else if (FQ_LANE == 0) {
    if (lane0 == 0) { w.sb->h_np = np; w.sb->h_nlp = nlp; w.sb->h_ns = ns; w.sb->h_nls = nls; }
    else { w.sb->h_np += np; w.sb->h_nlp += nlp; w.sb->h_ns += ns; w.sb->h_nls += nls; }
  }
  FQ_SYNC();
  return true;
}
FQ_DEV bool speculate(Wk &w, const u8 *p, u32 size, u32 i0, u32 n, bool reversed, u32 joff = 0, u32 lane0 = 0) {
  return speculate_t<false>(w, p, size, i0, n, reversed, joff, lane0, nullptr, nullptr);
}
// a stage-P chunk becomes the one the resolving wave works on
FQ_DEV void spec_adopt(Wk &w) {
  w.pq_lo[0] = w.sb->h_pq_lo[0];
  w.pq_lo[1] = w.sb->h_pq_lo[1];
  w.st[ST_GPROBE] += w.sb->h_np;
  w.st[ST_GSLOT] += w.sb->h_ns;
  w.st[ST_LPROBE] += w.sb->h_nlp;
  w.st[ST_LSLOT] += w.sb->h_nls;
}

FQ_DEV void load_state(Wk &w, u32 j);
FQ_DEV void replace_last_all(Wk &w, u64 symk);
// Scout waves, after stage P of a chunk.  If the chunk's first repair (repair_kmers_existing firing at a settled
// position, dna.cpp:362-369) comes with every lane before it settled, what the resolving wave will do there is known,
// and so is the state after it: the lanes inside the correction window (see suffix()) are speculated again from that
// state, in place, so that the resolving wave finds them ready instead of running stage P for the window itself.
// Returns false if the wave gave the chunk up (a restart request came in).
FQ_DEV bool scout_fix(Wk &w, const u8 *p, u32 size, u32 i0, u32 n, bool reversed) {
#if FQ_WAVE > 1
  SpecBuf *sb = w.sb;
  const u32 lane = FQ_LANE;
  FQ_SYNC();
  const bool f = lane < n && sb->sp_flag[lane] == 1, r = f && sb->sp_rep[lane] != 0xff;
  const u64 Fm = wave_ballot(f), Rm = wave_ballot(r);
  if (!Rm) return true;
  const u32 js = ctz64(Rm);
  // (the window may run past the chunk's last position into the buffer's spare lanes -- a 150 bp read's chunks have 45 of
  // the 64 -- which then hold positions of the next chunk as they look after the repair)
  const u32 lanes_left = FQSX_SPEC - (js + 1), pos_left = size - (i0 + js + 1);
  if ((~Fm & ((1ull << js) - 1ull)) != 0 || lanes_left == 0 || pos_left == 0) return true;
  const Kmer s0 = w.pm, s1 = w.sm_, s2 = w.bm, s3 = w.pm_u, s4 = w.sm_u, s5 = w.bm_u;
  const u32 s_cor = w.cor_pos, s_nrun = w.N_run;
  const u32 sym = rd_sym(w, p, i0 + js, size), rep = sb->sp_rep[js];
  load_state(w, js);
  replace_last_all(w, sym == 4 ? 0 : sym);
  km_replace_last(w.pm, rep); km_replace_last(w.sm_, rep); km_replace_last(w.bm, rep);
  w.cor_pos = i0 + js;
  w.N_run = sym == 4 ? w.N_run + 1 : 0;
  u32 cnt = w.cfg->bmer - 1;
  if (cnt > lanes_left) cnt = lanes_left;
  if (cnt > pos_left) cnt = pos_left;
  const bool whole = speculate(w, p, size, i0 + js + 1, cnt, reversed, 0, js + 1);
  w.pm = s0; w.sm_ = s1; w.bm = s2; w.pm_u = s3; w.sm_u = s4; w.bm_u = s5;
  w.cor_pos = s_cor; w.N_run = s_nrun;
  if (!whole) return false;
  FQ_SYNC();
  if (lane == 0) { sb->h_fix_lane = js; sb->h_fix_end = js + 1 + cnt; }
  FQ_SYNC();
#endif
  return true;
}

// Stage Q: append the mailbox entries of chunk positions [a,b) to this worker's lists, lane-parallel,
// in position order (b-mers, s-mers; two p-mer entries per position: direct then rc)
FQ_DEV void flush_pushes(Wk &w, u32 a, u32 b) {
  if (a >= b) return;
  TM_BEGIN(t_q);
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  const Mail &mb = cfg->mail[MAIL_B], &ms = cfg->mail[MAIL_S], &mp = cfg->mail[MAIL_P];
  FQ_SYNC();
  for (u32 base = a; base < b; base += FQ_WAVE) {
    const u32 t = base + FQ_LANE;
    const u32 f = t < b ? w.sb->pv_flag[t] : 0;
    const u32 nb = f & PV_B ? 1u : 0u, nsm = f & PV_S ? 1u : 0u, npm = f & PV_P ? 2u : 0u, nh = f & PV_PHID ? 2u : 0u;
#if FQ_WAVE > 1
    // every position contributes 0 or 1 (b, s) or 0 or 2 (p) entries: offsets and totals from ballots
    const u64 lt = (1ull << FQ_LANE) - 1ull;
    const u64 bb = wave_ballot(nb != 0), bs = wave_ballot(nsm != 0), bp = wave_ballot(npm != 0), bh = wave_ballot(nh != 0);
    const u32 ob = popc64(bb & lt), os = popc64(bs & lt), op = 2 * popc64(bp & lt);
    const u32 tb = popc64(bb), ts = popc64(bs), tp = 2 * popc64(bp), th = 2 * popc64(bh);
#else
    const u32 ob = 0, os = 0, op = 0;
    const u32 tb = nb, ts = nsm, tp = npm, th = nh;
#endif
    if (w.mn[MAIL_B] + tb > mb.cap || w.mn[MAIL_S] + ts > ms.cap || w.mn[MAIL_P] + tp > mp.cap) { w.err = FQSX_ERR_MAIL_FULL; return; }
    if (nb) mb.list[(u64)w.tid * mb.cap + w.mn[MAIL_B] + ob] = w.sb->pv_b[t];
    if (nsm) ms.list[(u64)w.tid * ms.cap + w.mn[MAIL_S] + os] = w.sb->pv_s[t];
    // LDS mirror of the most recent list entries
    if (nb) sm->pq_key[0][(w.mn[MAIL_B] + ob) & (FQSX_PQ - 1)] = w.sb->pv_b[t];
    if (nsm) sm->pq_key[1][(w.mn[MAIL_S] + os) & (FQSX_PQ - 1)] = w.sb->pv_s[t];
    if (npm) {
      u64 *dst = mp.list + (u64)w.tid * mp.cap + w.mn[MAIL_P] + op;
      dst[0] = w.sb->pv_pd[t];
      dst[1] = w.sb->pv_pr[t];
    }
    w.mn[MAIL_B] += tb; w.mn[MAIL_S] += ts; w.mn[MAIL_P] += tp;
    w.hidden += th;
    w.st[ST_MAIL] += tb + ts + tp;
  }
  TM_END(w, TM_POST, t_q);
}

// rolled k-mer state of chunk position j (after insert_zero)
FQ_DEV void load_state(Wk &w, u32 j) {
  WgShared *sm = w.sm;
  w.pm.dir = w.sb->sp_sdir[0][j]; w.pm.rc = w.sb->sp_src[0][j]; w.pm.cur = w.sb->sp_scur[0][j];
  w.sm_.dir = w.sb->sp_sdir[1][j]; w.sm_.rc = w.sb->sp_src[1][j]; w.sm_.cur = w.sb->sp_scur[1][j];
  w.bm.dir = w.sb->sp_sdir[2][j]; w.bm.rc = w.sb->sp_src[2][j]; w.bm.cur = w.sb->sp_scur[2][j];
  w.pm_u.dir = w.sb->sp_sdir[3][j]; w.pm_u.rc = w.sb->sp_src[3][j]; w.pm_u.cur = w.sb->sp_scur[3][j];
  w.sm_u.dir = w.sb->sp_sdir[4][j]; w.sm_u.rc = w.sb->sp_src[4][j]; w.sm_u.cur = w.sb->sp_scur[4][j];
  w.bm_u.dir = w.sb->sp_sdir[5][j]; w.bm_u.rc = w.sb->sp_src[5][j]; w.bm_u.cur = w.sb->sp_scur[5][j];
  w.N_run = w.sb->sp_nrun[j];
}
FQ_DEV void replace_last_all(Wk &w, u64 symk) {
  km_replace_last(w.pm, symk); km_replace_last(w.sm_, symk); km_replace_last(w.bm, symk);
  km_replace_last(w.pm_u, symk); km_replace_last(w.sm_u, symk); km_replace_last(w.bm_u, symk);
}
// ctx_letters before position i: reset (all ones) followed by the symbols 0..i-1 (dna.cpp:108-110,803)
FQ_DEV u64 letters_before(Wk &w, const u8 *p, u32 i, u32 size, u32 hist_start) {
  u64 ctx = ~0ull;
  u32 t0 = i > 16 ? i - 16 : 0;
  if (t0 < hist_start) t0 = hist_start;   // symbols before hist_start were never fed (minimizer-anchored coding)
  for (u32 t = t0; t < i; ++t) ctx = (ctx << 4) + rd_sym(w, p, t, size);
  return ctx;
}

// ---------------------------------------------------------------------------------------
// Stage C2: the symbols of a chunk's committed positions are coded AFTER stage C has resolved them all, because
// nothing stage C decides depends on the context models or the coder.  code_keys finishes the level keys of the
// positions stage P could not settle (lane-parallel), code_run then codes runs of consecutive positions with the
// context search and the model arithmetic LANE-PARALLEL, one position per lane, under three assumptions that are
// validated in position order: (1) the starting levels (int)(avg + 0.49) of the two hierarchies stay what they are
// at the run's start, (2) no context is created, (3) no counter comparison of the search flips and no model rescales
// because of the run's own earlier positions.  Counters and model statistics only grow by one use per position,
// so the state a position sees is (state at run start) + (uses by earlier positions of the run that end in the
// same slot), which a few ballots give.  The first position that breaks an assumption ends the run; it (and only
// it) goes through the sequential find_leveled / slot_encode.  What stays serial is what has to be: the fp64
// averages (two dependent operations per position) and the range coder (one division per position).
struct RoHit { u32 idx, counter; u64 q1, q2, q3; bool present; };
FQ_DEV RoHit ctx_probe_ro(Wk &w, u32 tag, u64 key, u32 &vis) {
  RoHit r;
  r.present = false; r.idx = FQSX_NIL; r.counter = 0; r.q1 = r.q2 = r.q3 = 0;
  const u64 *b = ctx_base(w);
  u64 h = ctx_hash(w, tag, key);
  for (u64 n = 0; n <= w.cfg->ctx_cap_mask; ++n) {
    const u64 *p = (const u64 *)__builtin_assume_aligned(b + 4 * h, 32);
    const u64 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    ++vis;
    const u32 tg = slot_tag(q1);
    if (!tg) return r;
    if (tg == tag && q0 == key) { r.present = true; r.idx = (u32)h; r.counter = (u32)q1; r.q1 = q1; r.q2 = q2; r.q3 = q3; return r; }
    h = (h + 1) & w.cfg->ctx_cap_mask;
  }
  return r;
}
// level keys (and the rank) of the positions stage C resolved itself: rank-coded from counts, or plain letters
FQ_DEV void code_keys(Wk &w, const u8 *p, u32 size, u32 i0, u32 j0, u32 m, bool reversed, u32 hist_start) {
  WgShared *sm = w.sm;
  const DevCfg *cfg = w.cfg;
  FQ_SYNC();
  for (u32 j = j0 + FQ_LANE; j < m; j += FQ_WAVE) {
    const u32 kind = w.sb->sp_kind[j], pos = i0 + j, sym = rd_sym(w, p, pos, size);
    if (kind == SK_RANK_PENDING) {
      const u64 cq = w.sb->sp_cq[j];
      C4 c;
      c.c[0] = (u32)(cq & 0xffff); c.c[1] = (u32)((cq >> 16) & 0xffff); c.c[2] = (u32)((cq >> 32) & 0xffff); c.c[3] = (u32)(cq >> 48);
      const u32 level = w.sb->sp_lvz[j] & 15u, cz = w.sb->sp_lvz[j] >> 4;
      u64 lev[7];
      if (!reversed) ctx_codes(lev, cfg, c, w.s_let, pos, level, cz, 0, size);
      else ctx_codes(lev, cfg, c, w.s_let, size - pos - 1, level, cz, 0, ~0u);   // dna.cpp:750-752
      for (u32 l = 0; l < 7; ++l) w.sb->sp_key[l][j] = lev[l];
      w.sb->sp_rsym[j] = (u8)rank_sym(w, c, sym);
      w.sb->sp_kind[j] = SK_RANK;
    } else if (kind == SK_LETTER_PENDING) {
      u64 lev[10];
      ctx_letters_keys(lev, cfg, pos, letters_before(w, p, pos, size, hist_start), size);   // dna.cpp:520-528,776-785
      for (u32 l = 0; l < 10; ++l) w.sb->sp_key[l][j] = lev[l];
      w.sb->sp_rsym[j] = (u8)sym;
      w.sb->sp_kind[j] = SK_LETTER;
    }
  }
  FQ_SYNC();
}
// one queue entry through the sequential routines
FQ_DEV void code_one(Wk &w, u32 j) {
  WgShared *sm = w.sm;
  const u32 e = j & (FQSX_CQ - 1);
  const u32 kd = sm->cq_kind[e], kind = kd & SK_KIND_MASK;
  const u32 sy = sm->cq_rsym[e];
  if (kd & SK_RESET) w.c_r_sym = 0;
  Slot4 s;
  if (kind == SK_RANK) {
    const u64 rs = (u64)popc64(w.c_r_sym) << SH_RSYM;
    u32 idx = find_leveled(w, 1, &sm->cq_key[0][e], FQSX_CQ, rs, 7, w.avg_code, TPL_CODES_Q2, TPL_CODES_Q3, TPL_CODES_TOT, s);
    if (idx != FQSX_NIL) slot_encode(w, idx, s, sy);
    w.c_r_sym = ((w.c_r_sym << 1) + (sy == 0 ? 1u : 0u)) & 0xff;   // update_ctx_r_sym, dna.cpp:664-671
  } else if (kind == SK_LETTER) {
    u32 idx = find_leveled(w, 2, &sm->cq_key[0][e], FQSX_CQ, 0, 9, w.avg_letters, TPL_LET_Q2, TPL_LET_Q3, TPL_LET_TOT, s);
    if (idx != FQSX_NIL) slot_encode(w, idx, s, sy);
    w.c_r_sym = (w.c_r_sym << 1) & 0xff;
  } else {
    const u64 k0 = sm->cq_key[0][e];
    const u32 tot = (u32)sm->cq_key[1][e];
    rc_encode_rd(w, (u32)k0, (u32)(k0 >> 32), tot, recip_u16(tot));
    w.c_r_sym = (w.c_r_sym << 1) & 0xff;   // (always followed by an SK_RESET entry before the next rank)
  }
}
// Returns the number of positions coded (0: position j0 needs the sequential routine).
FQ_DEV u32 code_run(Wk &w, u32 j0, u32 len) {
  WgShared *sm = w.sm;
  const int s0c = (int)(w.avg_code + 0.49), s0l = (int)(w.avg_letters + 0.49);
  const u64 ctx_r_sym = w.c_r_sym;
  u64 Z = 0, BAD = 0, KL = 0, KR = 0, RS = 0, SM[5] = {0, 0, 0, 0, 0};
  FQ_SYNC_MEM();
  TM_BEGIN(t_s);
  // ---- per-symbol and per-kind masks of the run (queue entries j0 .. j0+len-1)
#define CQE(t) ((j0 + (t)) & (FQSX_CQ - 1))
  for (u32 t = FQ_LANE; t < 64; t += FQ_WAVE) {
    const bool act = t < len;
    const u32 kd = act ? sm->cq_kind[CQE(t)] : 0u, kind = kd & SK_KIND_MASK;
    const bool letter = kind == SK_LETTER, raw = kind == SK_RAW;
    const u32 r = act && !raw ? sm->cq_rsym[CQE(t)] : 5u;
    const bool reset = (kd & SK_RESET) != 0;
#if FQ_WAVE > 1
    Z = wave_ballot(r == 0 && kind == SK_RANK);
    KL = wave_ballot(letter);
    KR = wave_ballot(raw);
    RS = wave_ballot(reset);
    for (u32 x = 0; x < 5; ++x) SM[x] = wave_ballot(r == x);
#else
    Z |= (u64)(r == 0 && kind == SK_RANK) << t;
    KL |= (u64)letter << t;
    KR |= (u64)raw << t;
    RS |= (u64)reset << t;
    for (u32 x = 0; x < 5; ++x) SM[x] |= (u64)(r == x) << t;
#endif
  }
  // ---- stage S: read-only search per position against the table as it is now
  for (u32 t = FQ_LANE; t < 64; t += FQ_WAVE) {
    bool bad = t >= len;
    u32 fin = FQSX_NIL, c0 = 0, thr = ~0u, vis = 0, lvl = 0;
    u64 q1 = 0, q2 = 0, q3 = 0;
    const bool raw = (KR >> t) & 1;
    if (!bad && !raw) {
      const bool letter = (KL >> t) & 1;
      const u32 tag = letter ? 2u : 1u;
      const int n_levels = letter ? 9 : 7, s0 = letter ? s0l : s0c;
      // r_sym history before this position (dna.cpp:664-671): the rank-0 flags of the up to 8 entries before it,
      // back to the last restart; the incoming history fills the rest of the window if the run has no restart so far
      const u64 rs_le = RS & (t >= 63 ? ~0ull : (2ull << t) - 1ull);
      u32 lo = t >= 8 ? t - 8 : 0u;
      bool incoming = t < 8;
      if (rs_le) {
        const u32 rp = 63u - (u32)__builtin_clzll(rs_le);
        if (rp > lo) lo = rp;
        incoming = false;
      }
      u32 hist = popc64((Z >> lo) & ((1ull << (t - lo)) - 1ull));
      if (incoming) hist += popc64(ctx_r_sym & ((1ull << (8 - t)) - 1ull));
      const u64 rs = letter ? 0ull : (u64)hist << SH_RSYM;
      const u64 *lev = &sm->cq_key[0][CQE(t)];
      const u32 ls = FQSX_CQ;
      // (tried in round 4 and dropped: fetching the first slots of levels s0 - 1, s0, s0 + 1 together, so that the walk below finds
      // its neighbours answered -- 175.3 vs 176.9 Mbases/s in steady state: the models wave is not what the worker waits for)
      RoHit h = ctx_probe_ro(w, tag, LEVKEY(s0), vis);
      int i = s0;
      if (h.present) {
        const u32 thr0 = letter ? letters_thr(s0) : code_thr(s0);   // only the first test uses the letters' own thresholds (dna.cpp:2244)
        if (h.counter < thr0) thr = thr0;
        else {
          RoHit last = h;
          for (i = s0 + 1; i < n_levels; ++i) {
            RoHit q = ctx_probe_ro(w, tag, LEVKEY(i), vis);
            if (!q.present) break;
            last = q;
            if (q.counter < code_thr(i)) { thr = code_thr(i); break; }
          }
          if (thr == ~0u) {          // every level from s0 up to i-1 is full
            --i;
            if (last.counter >= code_thr(i) && i + 1 < n_levels) bad = true;   // the deepest one would be cloned: sequential routine
            else if (last.counter < code_thr(i)) thr = code_thr(i);             // (letters: full by their own threshold only)
          }
          h = last;
        }
      } else {
        for (i = s0 - 1; i >= 0; --i) {
          h = ctx_probe_ro(w, tag, LEVKEY(i), vis);
          if (h.present) break;
        }
        if (i < 0) bad = true;       // level 0 would be created from the template
        else if (h.counter >= code_thr(i)) { if (i + 1 < n_levels) bad = true; }   // clone
        else thr = code_thr(i);
      }
      if (!bad) { fin = h.idx; c0 = h.counter; q1 = h.q1; q2 = h.q2; q3 = h.q3; lvl = (u32)i; }
    }
    sm->fr_idx[t] = fin; sm->fr_c0[t] = c0; sm->fr_thr[t] = thr; sm->fr_vis[t] = vis; sm->fr_lvl[t] = (u8)lvl;
    sm->fr_q1[t] = q1; sm->fr_q2[t] = q2; sm->fr_q3[t] = q3;
    sm->fr_bad[t] = bad ? 1 : 0;
  }
  FQ_SYNC();
  TM_END(w, TM_CR_S, t_s);
  TM_BEGIN(t_mid);
  // ---- lanes that end in the same slot
#if FQ_WAVE > 1
  {
    const u32 t = FQ_LANE;
    // intersect, over the bits of the slot index, the ballots of the lanes that agree with mine on that bit
    const u32 mine = sm->fr_idx[t];
    const bool ok = !sm->fr_bad[t] && !((KR >> t) & 1);
    u64 same = wave_ballot(ok);
    for (u64 cap = w.cfg->ctx_cap_mask, b = 0; cap; cap >>= 1, ++b) {
      const bool bit = (mine >> b) & 1u;
      const u64 m = wave_ballot(bit);
      same &= bit ? m : ~m;
    }
    sm->fr_same[t] = ok ? same : 0ull;
  }
#else
  for (u32 t = 0; t < 64; ++t) {
    u64 same = 0;
    if (!sm->fr_bad[t] && !((KR >> t) & 1))
      for (u32 u = 0; u < 64; ++u) same |= (u64)(!sm->fr_bad[u] && !((KR >> u) & 1) && sm->fr_idx[u] == sm->fr_idx[t]) << u;
    sm->fr_same[t] = same;
  }
#endif
  FQ_SYNC();
  // ---- validation of assumption (3) and the model arithmetic, per position
  u32 lane_f = 0, lane_c = 0, lane_t = 1;       // GPU: the position's coder triple, reciprocal and average term stay in
  double lane_pl = 0.0;                         // its lane's registers; the serial loops fetch them with v_readlane
  u64 lane_m = 0;
  for (u32 t = FQ_LANE; t < 64; t += FQ_WAVE) {
    bool bad = sm->fr_bad[t] != 0;
    u32 f = 0, c = 0, tot = 0;
    if (!bad && ((KR >> t) & 1)) {        // finished triple
      const u64 k0 = sm->cq_key[0][CQE(t)];
      f = (u32)k0; c = (u32)(k0 >> 32); tot = (u32)sm->cq_key[1][CQE(t)];
    } else if (!bad) {
      const u64 below = sm->fr_same[t] & ((1ull << t) - 1ull);
      const u32 occ = popc64(below);
      const u32 thr = sm->fr_thr[t];
      if (thr != ~0u && sm->fr_c0[t] + occ >= thr) bad = true;           // an earlier position of the run filled the context
      tot = (u32)(sm->fr_q1[t] >> 48) + 4 * occ;
      if (tot + 4 >= (1u << 15)) bad = true;                              // this use rescales the model (rc.h:186-197)
      const u64 q2 = sm->fr_q2[t], q3 = sm->fr_q3[t];
      u32 st[5] = {(u32)(q2 & 0xffff), (u32)((q2 >> 16) & 0xffff), (u32)((q2 >> 32) & 0xffff), (u32)(q2 >> 48), (u32)(q3 & 0xffff)};
      const u32 r = sm->cq_rsym[CQE(t)];
      for (u32 x = 0; x < 5; ++x) {
        st[x] += 4 * popc64(below & SM[x]);
        if (x < r) c += st[x];
      }
      f = st[r];
    }
#if FQ_WAVE > 1
    lane_f = f; lane_c = c; lane_t = tot ? tot : 1u;
    lane_m = recip64_u16(lane_t);
    lane_pl = __dmul_rn(1.0 - 0.999, (double)sm->fr_lvl[t]);
    BAD = wave_ballot(bad);
#else
    sm->fr_f[t] = f; sm->fr_c[t] = c; sm->fr_t[t] = tot;
    BAD |= (u64)bad << t;
#endif
  }
  FQ_SYNC();
  u32 L = BAD ? ctz64(BAD) : 64u;
  TM_END(w, TM_CR_MID, t_mid);
  TM_BEGIN(t_avg);
  // ---- assumption (1) and the running averages, in position order
  {
    double ac = w.avg_code, al = w.avg_letters;
    u32 t = 0;
#if FQ_WAVE > 1
    // The recurrence runs over all L positions without looking at its values (two dependent fp64 operations per
    // position; ema_update with the level term precomputed by the position's lane); every lane keeps the average its
    // position started from, the starting-level test is then one lane-parallel comparison, and in the rare case it
    // fails somewhere the recurrence is simply run again up to that position.
    // (the two averages are chains of their own: each walks the set bits of its positions' mask -- bit scan, two
    // v_readlane, the two dependent fp64 operations and the capture of the value the position started from)
    for (int pass = 0; pass < 2; ++pass) {
      double before = 0.0;
      ac = w.avg_code; al = w.avg_letters;
      const u64 inl = L >= 64 ? ~0ull : (1ull << L) - 1ull;
      const u32 my = FQ_LANE;
      for (u64 m = ~KR & ~KL & inl; m; m &= m - 1) {
        const u32 p = ctz64(m);
        const double pl = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lane_pl), p), __builtin_amdgcn_readlane(__double2loint(lane_pl), p));
        before = my == p ? ac : before;
        ac = __dadd_rn(__dmul_rn(0.999, ac), pl);
      }
      for (u64 m = ~KR & KL & inl; m; m &= m - 1) {
        const u32 p = ctz64(m);
        const double pl = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lane_pl), p), __builtin_amdgcn_readlane(__double2loint(lane_pl), p));
        before = my == p ? al : before;
        al = __dadd_rn(__dmul_rn(0.999, al), pl);
      }
      if (pass) break;
      const u32 ln = FQ_LANE;
      const bool chk = ln < L && !((KR >> ln) & 1);
      const u64 viol = wave_ballot(chk && (int)(before + 0.49) != (((KL >> ln) & 1) ? s0l : s0c));
      if (!viol) break;
      L = ctz64(viol);   // the first position whose starting level differs: the run ends before it
    }
    t = L;
#else
    for (; t < L; ++t) {
      if ((KR >> t) & 1) continue;
      if ((KL >> t) & 1) {
        if ((int)(al + 0.49) != s0l) break;
        al = ema_update(al, (double)sm->fr_lvl[t]);
      } else {
        if ((int)(ac + 0.49) != s0c) break;
        ac = ema_update(ac, (double)sm->fr_lvl[t]);
      }
    }
#endif
    L = t;
    TM_END(w, TM_CR_AVG, t_avg);
    if (L == 0) return 0;
    w.avg_code = ac;
    w.avg_letters = al;
  }
  TM_BEGIN(t_rc);
  // ---- commit: the last position of every slot writes the slot's counter and statistics
  const u64 within = L >= 64 ? ~0ull : (1ull << L) - 1ull;
  u32 vis_sum = 0;
  for (u32 t = FQ_LANE; t < 64; t += FQ_WAVE) {
    u32 vis = 0;
    if (t < L) {
      vis = sm->fr_vis[t];
      const u64 grp = sm->fr_same[t] & within;
      if ((grp >> t) == 1ull) {       // no later position of the run ends here
        const u64 q2 = sm->fr_q2[t], q3 = sm->fr_q3[t];
        u32 st[5] = {(u32)(q2 & 0xffff), (u32)((q2 >> 16) & 0xffff), (u32)((q2 >> 32) & 0xffff), (u32)(q2 >> 48), (u32)(q3 & 0xffff)};
        for (u32 x = 0; x < 5; ++x) st[x] += 4 * popc64(grp & SM[x]);
        const u32 uses = popc64(grp);
        const u64 tot = (sm->fr_q1[t] >> 48) + 4ull * uses;
        u64 *p = ctx_base(w) + 4 * (u64)sm->fr_idx[t];
        p[1] = (sm->fr_q1[t] & 0x0000ffff00000000ULL) | (tot << 48) | (u64)(sm->fr_c0[t] + uses);
        p[2] = (u64)st[0] | ((u64)st[1] << 16) | ((u64)st[2] << 32) | ((u64)st[3] << 48);
        p[3] = (q3 & ~0xffffULL) | st[4];
      }
    }
    vis_sum += vis;
  }
  FQ_SYNC_MEM();
  w.st[ST_CTX] += wave_sum32(vis_sum);
  TM_END_MD(w, TM_SP_ROLL, t_rc);
  TM_BEGIN(t_rq);
  // ---- the range coder, in position order
#if FQ_WAVE > 1
  if (w.rcq) {   // the run's steps go to the range-coder wave, one lane per position
    const bool space_ = rq_wait_space(w, L);
    TM_END_MD(w, TM_SP_PROBE, t_rq);
    if (space_) {
      const u32 t = FQ_LANE;
      if (t < L) {
        const u32 e = (w.rq_tail + t) & (FQSX_RQ - 1);
        sm->rq_f[e] = lane_f; sm->rq_c[e] = lane_c; sm->rq_t[e] = lane_t; sm->rq_m[e] = lane_m;
      }
      FQ_SYNC();
      w.rq_tail += L;
      lds_store_rel(&sm->rq_tail, w.rq_tail);
    }
  } else
    for (u32 t = 0; t < L; ++t) {
      const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(lane_m >> 32), t) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)lane_m, t);
      rc_encode_m(w, (u32)__builtin_amdgcn_readlane((int)lane_f, t), (u32)__builtin_amdgcn_readlane((int)lane_c, t),
                  (u32)__builtin_amdgcn_readlane((int)lane_t, t), m);
    }
#else
  for (u32 t = 0; t < L; ++t) rc_encode(w, sm->fr_f[t], sm->fr_c[t], sm->fr_t[t]);
#endif
  TM_BEGIN(t_rs);
  // r_sym history after the run (a letter position shifts in a zero)
  {
#if FQ_WAVE > 1
    // closed form of the loop below: the history is the run's Z bits since the last reset (or since the old history),
    // newest in bit 0, cut to 8 bits
    const u64 within_l = L >= 64 ? ~0ull : (1ull << L) - 1ull;
    const u64 rs = RS & within_l;
    const u32 from = rs ? 63u - (u32)__builtin_clzll(rs) : 0u;        // position of the last reset (the run starts there)
    const u32 n = L - from;                                           // positions shifted in since
    const u64 zrev = (((u64)__builtin_bitreverse32((u32)Z) << 32) | __builtin_bitreverse32((u32)(Z >> 32))) >> (64 - L);   // position L-1 in bit 0
    const u64 znew = n >= 64 ? zrev : zrev & ((1ull << n) - 1ull);
    const u64 hold = rs ? 0ull : (n >= 8 ? 0ull : ctx_r_sym << n);
    w.c_r_sym = (hold | znew) & 0xff;
#else
    u64 h = ctx_r_sym;
    for (u32 t = 0; t < L; ++t) {
      if ((RS >> t) & 1ull) h = 0;
      h = ((h << 1) + ((Z >> t) & 1ull)) & 0xff;
    }
    w.c_r_sym = h;
#endif
  }
#undef CQE
  TM_END_MD(w, TM_SP_HIT, t_rs);
#ifdef FQSX_TIMING_MODELS
  TM_COUNT(w, TM_SP_MISS);   // (runs)
#endif
  TM_END(w, TM_CR_RC, t_rc);
  return L;
}
// the coder: entries [head, tail) of the queue, in order
FQ_DEV u32 cq_process(Wk &w, u32 head, u32 avail) {
  u32 L = code_run(w, head, avail < 64 ? avail : 64);
  if (L == 0) {
    code_one(w, head);
    TM_COUNT(w, CN_P2);
    L = 1;
  }
  return L;
}
// the committed positions [j0, m) of the chunk go to the coding queue; the single-wave build codes them right away
FQ_DEV void code_chunk(Wk &w, const u8 *p, u32 size, u32 i0, u32 j0, u32 m, bool reversed, u32 hist_start, bool &first) {
  if (m <= j0) return;
  TM_BEGIN(t_c2);
  WgShared *sm = w.sm;
  code_keys(w, p, size, i0, j0, m, reversed, hist_start);
  TM_END(w, TM_KEYS, t_c2);
  TM_BEGIN(t_cq);
  if (!cq_wait_space(w, m - j0)) return;
  for (u32 j = j0 + FQ_LANE; j < m; j += FQ_WAVE) {
    const u32 e = (w.cq_tail + (j - j0)) & (FQSX_CQ - 1);
    const u32 kind = w.sb->sp_kind[j];
    const u32 nk = kind == SK_LETTER ? 10u : 7u;
    for (u32 l = 0; l < nk; ++l) sm->cq_key[l][e] = w.sb->sp_key[l][j];
    sm->cq_kind[e] = (u8)(kind | (first && j == j0 ? SK_RESET : 0u));
    sm->cq_rsym[e] = w.sb->sp_rsym[j];
  }
  FQ_SYNC();
  first = false;
  cq_publish(w, m - j0);
  TM_END(w, TX_CHUNKQ, t_cq);
  if (!w.piped) {
    while (w.cq_head != w.cq_tail && !w.err) w.cq_head += cq_process(w, w.cq_head, w.cq_tail - w.cq_head);
    TM_END(w, TM_FAST, t_c2);
  }
}

#if FQ_WAVE > 1
// Positions of a chunk where stage P found nothing in any table (global b miss, cascade empty) and where no list
// entry stage P may have missed -- the entries since its snapshot (LDS mirror) and those of the chunk's earlier
// positions -- lies in the position's b or s sibling group, i.e. exactly the positions for which pend_conflict
// would say no twice.  The mirror is tested through hashed bit sets (a set bit only sends the position down the
// exact per-position path), the chunk's own entries exactly.  One position per lane.
FQ_DEV u32 grp_hash(u64 g) { return (u32)((g * 0x9E3779B97F4A7C15ull) >> 52); }   // 12 bits
// two 13-bit positions in an 8192-bit set per group: with a few hundred entries pending, one bit per entry left a quarter
// of the positions "maybe" by chance, which cut the quiet stretches short
FQ_DEV void qm_set(u32 *bits, u64 g) {
  const u64 h = g * 0x9E3779B97F4A7C15ull;
  const u32 a = (u32)(h >> 51), b = (u32)(h >> 38) & 8191u;
  atomicOr(&bits[a >> 5], 1u << (a & 31));
  atomicOr(&bits[b >> 5], 1u << (b & 31));
}
FQ_DEV bool qm_test(const u32 *bits, u64 g) {
  const u64 h = g * 0x9E3779B97F4A7C15ull;
  const u32 a = (u32)(h >> 51), b = (u32)(h >> 38) & 8191u;
  return ((bits[a >> 5] >> (a & 31)) & (bits[b >> 5] >> (b & 31)) & 1u) != 0;
}
// Round 4: the same test also answers, for EVERY position whose cascade stage P resolved (hits included), whether a pending
// local insert could change its b-mer look-up (ncb: no) / its s-mer look-up (ncs: no) -- what the per-position path otherwise asks
// pend_conflict for, twice per position, with a dozen dependent LDS round trips each.  quiet = the positions of the old mask.
struct ConfMasks { u64 quiet, ncb, ncs; };
FQ_DEV ConfMasks conflict_masks(Wk &w, u32 n, bool want_quiet) {
  ConfMasks R;
  R.quiet = R.ncb = R.ncs = 0;
  SpecBuf *sb = w.sb;
  WgShared *sm = w.sm;
  const DevCfg *cfg = w.cfg;
  const u32 lane = FQ_LANE;
  FQ_SYNC();
  const u32 xfl = lane < n && sb->sp_flag[lane] == 3 ? sb->sx_flag[lane] : 0u;
  const bool cand = (xfl & SX_VALID) != 0, qcand = (xfl & (SX_VALID | SX_HITS)) == SX_VALID;
  const u64 A = wave_ballot(cand);
  if (popc64(A) < 4) return R;   // not worth the set-up: the per-position path handles them
  const u32 lo_b = w.pq_lo[0], hi_b = w.mn[MAIL_B], lo_s = w.pq_lo[1], hi_s = w.mn[MAIL_S];
  if (hi_b - lo_b > FQSX_PQ || hi_s - lo_s > FQSX_PQ) return R;
  const u32 k2b = 2 * cfg->gb.k, k2s = 2 * cfg->gs.k;
  const u64 lmb = (1ull << (k2b - 2)) - 1ull, lms = (1ull << (k2s - 2)) - 1ull;
  // this lane's two sibling groups
  const u64 bd = sb->sp_sdir[2][lane], br = sb->sp_src[2][lane], sd = sb->sp_sdir[1][lane], sr = sb->sp_src[1][lane];
  const bool ndb = (bd & cfg->gb.kernel_mask) < (br & cfg->gb.kernel_mask), nds = (sd & cfg->gs.kernel_mask) < (sr & cfg->gs.kernel_mask);
  const u64 vb = (ndb ? bd : br) >> (64 - k2b), vs = (nds ? sd : sr) >> (64 - k2s);
  const u64 gb = ndb ? (vb >> 2) : (vb & lmb), gs = nds ? (vs >> 2) : (vs & lms);
  // bit sets of the mirror entries' groups, both readings of every entry
  for (u32 i = lane; i < 4 * 256; i += FQ_WAVE) (&sm->qm_bits[0][0])[i] = 0;
  FQ_SYNC();
  for (u32 e = lo_b + lane; e < hi_b; e += FQ_WAVE) {
    const u64 pv = sm->pq_key[0][e & (FQSX_PQ - 1)] >> (64 - k2b);
    qm_set(sm->qm_bits[0], pv >> 2);
    qm_set(sm->qm_bits[1], pv & lmb);
  }
  for (u32 e = lo_s + lane; e < hi_s; e += FQ_WAVE) {
    const u64 pv = sm->pq_key[1][e & (FQSX_PQ - 1)] >> (64 - k2s);
    qm_set(sm->qm_bits[2], pv >> 2);
    qm_set(sm->qm_bits[3], pv & lms);
  }
  FQ_SYNC();
  const u32 hb = grp_hash(gb), hs = grp_hash(gs);
  bool maybe_b = qm_test(sm->qm_bits[ndb ? 0 : 1], gb), maybe_s = qm_test(sm->qm_bits[nds ? 2 : 3], gs);
  // the entries of the chunk's earlier positions, exactly: every position files its lane under the 6-bit hashes of its
  // entries' groups (both readings; the bit sets above are reused as 4 x 64 lane masks), and a lane compares groups
  // only with the earlier lanes filed where its own group hashes to
  FQ_SYNC();
  u64 *lm = (u64 *)&sm->qm_bits[0][0];   // [4][64]
  for (u32 i = lane; i < 4 * 64; i += FQ_WAVE) lm[i] = 0;
  FQ_SYNC();
  const u32 myf = lane < n ? sb->pv_flag[lane] : 0u;
  if (myf & PV_B) {
    const u64 pv = sb->pv_b[lane] >> (64 - k2b);
    lds_or64(&lm[0 * 64 + (grp_hash(pv >> 2) & 63)], 1ull << lane);
    lds_or64(&lm[1 * 64 + (grp_hash(pv & lmb) & 63)], 1ull << lane);
  }
  if (myf & PV_S) {
    const u64 pv = sb->pv_s[lane] >> (64 - k2s);
    lds_or64(&lm[2 * 64 + (grp_hash(pv >> 2) & 63)], 1ull << lane);
    lds_or64(&lm[3 * 64 + (grp_hash(pv & lms) & 63)], 1ull << lane);
  }
  FQ_SYNC();
  const u64 lt = (1ull << lane) - 1ull;
  if (cand && !maybe_b)
    for (u64 c = lm[(ndb ? 0 : 64) + (hb & 63)] & lt; c && !maybe_b; c &= c - 1ull) {
      const u64 pv = sb->pv_b[ctz64(c)] >> (64 - k2b);
      maybe_b = ndb ? (pv >> 2) == gb : (pv & lmb) == gb;
    }
  if (cand && !maybe_s)
    for (u64 c = lm[(nds ? 128 : 192) + (hs & 63)] & lt; c && !maybe_s; c &= c - 1ull) {
      const u64 pv = sb->pv_s[ctz64(c)] >> (64 - k2s);
      maybe_s = nds ? (pv >> 2) == gs : (pv & lms) == gs;
    }
  R.ncb = wave_ballot(cand && !maybe_b);
  R.ncs = wave_ballot(cand && !maybe_s);
  if (want_quiet) R.quiet = wave_ballot(qcand && !maybe_b && !maybe_s);
  return R;
}
#endif

// A scout made the chunk's local look-ups against the local tables as they were when its probes started; entries this
// worker has pushed since (typically: the read before, which overlaps this one -- two reads of a worker per launch in the
// warm-up blocks) may change them, and the per-position path would then wait for the inserter wave and ask again, position
// by position.  Where a chunk has several such positions it is cheaper to do that once for the whole chunk: wait until
// every entry pushed so far is in the tables, redo the two local look-ups of the miss cascade one position per lane, and
// let the validation (pend_conflict, quiet_miss_mask) start from the current end of the lists -- only the chunk's own
// earlier entries remain to be checked, exactly.  Lanes [j0, n); the sweeps a scout made for positions whose cascade was
// empty stay valid for the positions it is still empty for (the tables only grow inside a segment).
FQ_DEV void local_refresh(Wk &w, u32 j0, u32 n) {
  const DevCfg *cfg = w.cfg;
  SpecBuf *sb = w.sb;
  const u32 lane = FQ_LANE;
  if (w.mn[MAIL_B] == w.pq_lo[0] && w.mn[MAIL_S] == w.pq_lo[1]) return;   // nothing pushed since the snapshot
  FQ_SYNC();
  const bool cand = lane >= j0 && lane < n && sb->sp_flag[lane] == 3 && (sb->sx_flag[lane] & SX_VALID) != 0;
  if (popc64(wave_ballot(cand)) < 6) return;   // (the per-position path handles a few)
  lq_flush(w, MAIL_B);
  lq_flush(w, MAIL_S);
  if (w.err) return;
  u64 nls = 0;
  u32 nlp = 0;
  if (cand) {
    const u64 bd = sb->sp_sdir[2][lane], br = sb->sp_src[2][lane], sd = sb->sp_sdir[1][lane], sr = sb->sp_src[1][lane];
    const bool ndb = (bd & cfg->gb.kernel_mask) < (br & cfg->gb.kernel_mask), nds = (sd & cfg->gs.kernel_mask) < (sr & cfg->gs.kernel_mask);
    const u64 kb = ndb ? bd : br, ks = nds ? sd : sr;
    const TabL fb = tab_first_l(cfg->l_b, w.tid, kb), fs = tab_first_l(cfg->l_s, w.tid, ks);
    u32 xf = sb->sx_flag[lane] & (SX_VALID | SX_UNC | SX_S);
    C4 c;
    c4_zero(c);
    tab_rest(cfg->l_b, fb, kb, ndb, c, nls);
    ++nlp;
    if (c4_any(c)) {
      xf = SX_VALID | SX_LB;
      sb->sx_lb[lane] = c.c[0] | (c.c[1] << 8) | (c.c[2] << 16) | (c.c[3] << 24);
    } else if (!(xf & (SX_UNC | SX_S))) {
      c4_zero(c);
      tab_rest(cfg->l_s, fs, ks, nds, c, nls);
      ++nlp;
      if (c4_any(c)) {
        xf |= SX_LS;
        sb->sx_ls[lane] = (u64)c.c[0] | ((u64)c.c[1] << 16) | ((u64)c.c[2] << 32) | ((u64)c.c[3] << 48);
      }
    }
    sb->sx_flag[lane] = (u8)xf;
  }
  FQ_SYNC();
  w.st[ST_LPROBE] += wave_sum32(nlp);
  w.st[ST_LSLOT] += wave_sum64(nls);
  w.pq_lo[0] = w.mn[MAIL_B];
  w.pq_lo[1] = w.mn[MAIL_S];
}

// Length of the next chunk when `rem` suffix positions are left: as few chunks as the lanes allow, of equal length (a
// 150 bp read: 45 + 45 + 45 rather than 64 + 64 + 7), so that the scout waves share a read's probes and sweeps evenly.
// The resolving wave and the scout waves enumerate the chunks with this one rule.
FQ_DEV u32 chunk_len(u32 rem) {
  const u32 nch = (rem + FQSX_SPEC - 1) / FQSX_SPEC;
  return (rem + nch - 1) / nch;
}
// hand-over of the scout waves' stage-P chunks (see scout_segment_body)
FQ_DEV void scout_release(Wk &w) {
  w.sc_taken += 1;
  FQ_SYNC();
  lds_store_rel(&w.sm->sc_taken, w.sc_taken);
}
// The scout waves start again: from the exact state before position i0 of read `read` (a k-mer correction), or, with
// i0 == pmer, from the head of that read (the resolving wave has finished the read before on its own)
// The chunks of the new epoch take the ring slots in turn starting after the slot of the chunk the resolving wave holds
// (w.sb), so with `hold` the request can go out while that chunk is still being read: the slot counts as an unreleased
// chunk (sc_taken = -1) until scout_unhold.
FQ_DEV void scout_restart(Wk &w, u32 read, u32 i0, const u64 s_let[4], u32 flags, bool hold) {
  WgShared *sm = w.sm;
  if (lds_load_acq(&sm->sc_dead)) return;
  const bool in_ring = w.sb != &sm->sb[0];
  const u32 base = in_ring ? ((u32)(w.sb - &sm->sb[1]) + 1u) % w.nsc : 0u;
  hold = hold && in_ring;
  FQ_SYNC();
  if (FQ_LANE == 0) {
    WgShared::ScoutReq &q = sm->sc_req;
    q.slot_base = base;
    q.read = read; q.i0 = i0; q.cor_pos = w.cor_pos; q.n_run = w.N_run;
    q.flags = flags | (w.rq_rev ? (u32)SCQ_REVERSED : 0u); q.size = w.rq_size; q.p = (u64)w.rq_p;
    const Kmer *k[6] = {&w.pm, &w.sm_, &w.bm, &w.pm_u, &w.sm_u, &w.bm_u};
    for (u32 x = 0; x < 6; ++x) { q.kdir[x] = k[x]->dir; q.krc[x] = k[x]->rc; q.kcur[x] = k[x]->cur; }
    for (u32 x = 0; x < 4; ++x) q.s_let[x] = s_let[x];
  }
  w.sc_taken = 0;
  w.sc_base = base;
  FQ_SYNC();
  lds_store_rel(&sm->sc_taken, hold ? 0xffffffffu : 0u);
  w.sc_epoch += 1;
  lds_store_rel(&sm->sc_req_seq, w.sc_epoch);
}
FQ_DEV void scout_unhold(Wk &w) {   // the chunk held across an early restart is done with
  FQ_SYNC();
  lds_store_rel(&w.sm->sc_taken, w.sc_taken);
}
// The scouts' chunk that covers position `at` of the current read / request becomes w.sb; chunks that lie wholly before
// it (inside a window the resolving wave has just covered itself) are released on the way.  False: go on without them.
FQ_DEV bool scout_seek(Wk &w, u32 at, const SpecBuf *fixed, u32 fixed_pub) {
  WgShared *sm = w.sm;
  for (;;) {
    SpecBuf *b = &sm->sb[1 + (w.sc_base + w.sc_taken) % w.nsc];
    u32 spins = 0;
    // the chunk with this number of this epoch (a slot may still hold the one of an earlier epoch with the same number)
    while (lds_load_acq(&b->h_pub) != w.sc_taken + 1 || b->h_epoch != w.sc_epoch) {
      fq_sleep();
      if (lds_load_acq(&sm->sc_dead) || spin_expired(spins)) { w.sc_abandoned = true; return false; }   // never spin forever on the GPU
    }
    const i32 older = (i32)(b->h_read - w.sc_read);
    if (older < 0) { scout_release(w); continue; }   // a chunk of an earlier read that ended inside a window
    if (older > 0 || at < b->h_i0) { w.sc_abandoned = true; return false; }   // (cannot happen: the waves enumerate the chunks alike)
    const u32 lanes = fixed == b && fixed_pub == b->h_pub && b->h_fix_end > b->h_n ? b->h_fix_end : b->h_n;   // (scout_fix's spare lanes count once its repair has been taken)
    if (at < b->h_i0 + lanes) { w.sb = b; return true; }
    scout_release(w);
  }
}

// compress_suffix, dna.cpp:674-877, as chunks of stage P (parallel) -> stage C (the serial loop below:
// context look-up + range coding, plus the complete reference logic for positions P could not
// settle) -> stage Q (parallel mailbox appends)
FQ_DEV void suffix(Wk &w, const u8 *p, u32 size, bool original_order, u32 start_pos = 0, bool reversed = false, u32 hist_start = 0) {
  const DevCfg *cfg = w.cfg;
  WgShared *sm = w.sm;
  bool first = true;   // the r_sym history starts empty (dna.cpp:676)
  u32 i = start_pos ? start_pos : original_order ? cfg->prefix : cfg->pmer;
  w.rq_p = p; w.rq_size = size; w.rq_rev = reversed;
  if (w.rq_early) w.rq_early = false;   // (prefix_sorted has posted this call's request already)
  else if (w.scout && w.sc_reqmode && i < size) {   // no read-head wave: this call is one request to the scout waves
    w.sc_read += 1;
    w.sc_abandoned = w.rq_noscout;
    if (!w.rq_noscout) scout_restart(w, w.sc_read, i, w.s_let, 0, false);
  }
  w.rq_noscout = false;
  const u32 dbg = cfg->dbg;
  if ((dbg & FQSX_DBG_ABANDON) && w.scout && w.sc_read % 3 == 1) w.sc_abandoned = true;   // (test switch) this read without the scouts
  // A k-mer correction at position q (repair_kmers_*, or the adoption of the uncorrected k-mers) changes what stage P
  // would have found only where a corrected k-mer still holds a corrected symbol or the distance to the correction
  // enters a context key: positions q + 1 .. q + bmer - 1.  Beyond that window the scouts' chunks -- rolled from the
  // state the read (or the request) started from -- stand as they are.  So the scouts never hear of a correction: the
  // resolving wave runs stage P itself for the window (`own_end`: positions below it are its own) and then goes on in
  // the scouts' chunk that covers the next position, from whatever lane that is.
  u32 at = i;               // next position to resolve
  u32 own_lo = 0, own_hi = 0;   // positions in [own_lo, own_hi) are speculated by this wave itself
  const SpecBuf *fixed = nullptr;   // ring chunk whose predicted repair (scout_fix) this wave has taken: its fixed lanes stand
  u32 fixed_pub = 0;
  const u32 at_first = at;
  const SpecBuf *counted = nullptr;   // ring chunk whose probes have been accounted (a chunk can be entered more than once)
  u32 counted_pub = 0;
  while (at < size && !w.err) {
    TM_BEGIN(t_sp);
    if ((dbg & FQSX_DBG_ABANDON) && w.scout && w.sc_read % 3 == 2 && at != at_first) w.sc_abandoned = true;   // ... this one after its first chunk
    bool pre = false;
    u32 n, j0 = 0;
    WHATIF(*w.cfg, 1, 1);
    if (!(at >= own_lo && at < own_hi) && w.scout && !w.sc_abandoned) pre = scout_seek(w, at, fixed, fixed_pub);   // stage P done ahead of time by a scout wave
    if (pre) {
      // lanes a scout has speculated again for a repair it foresaw (scout_fix) are only good if this wave took that
      // very repair from the chunk; entered from elsewhere they are covered by this wave itself
      const u32 fl = w.sb->h_fix_lane, fe = w.sb->h_fix_end, jx = at - w.sb->h_i0;
      if (fl != 0xff && jx > fl && jx < fe && !(fixed == w.sb && fixed_pub == w.sb->h_pub)) {
        own_lo = at;
        if (own_hi < w.sb->h_i0 + fe) own_hi = w.sb->h_i0 + fe;
        pre = false;
      }
    }
    if (pre) {
      i = w.sb->h_i0; n = w.sb->h_n; j0 = at - i;
      if (fixed == w.sb && fixed_pub == w.sb->h_pub && w.sb->h_fix_end > n) n = w.sb->h_fix_end;   // (the spare lanes scout_fix filled)
      w.pq_lo[0] = w.sb->h_pq_lo[0];
      w.pq_lo[1] = w.sb->h_pq_lo[1];
      if (counted != w.sb || counted_pub != w.sb->h_pub) { counted = w.sb; counted_pub = w.sb->h_pub; spec_adopt(w); }
    } else {
      i = at;
      n = chunk_len(size - at);
      if (w.scout && !w.sc_abandoned && at >= own_lo && at < own_hi && own_hi - at < n) n = own_hi - at;   // (just the window: a scout's chunk takes over behind it)
      w.sb = &sm->sb[0];
      TM_BEGIN(t_own);
      speculate(w, p, size, i, n, reversed);
      TM_END(w, TX_OWNSPEC, t_own);
      spec_adopt(w);
    }
    TM_END(w, TM_SPEC, t_sp);
    TM_COUNT(w, CN_CHUNK);
    u64 Fm = 0, Rm = 0;   // positions settled by stage P; settled positions whose repair fires
    for (u32 t = FQ_LANE; t < 64; t += FQ_WAVE) {
      const bool f = t < n && w.sb->sp_flag[t] == 1, r = f && w.sb->sp_rep[t] != 0xff;
#if FQ_WAVE > 1
      Fm = wave_ballot(f); Rm = wave_ballot(r);
#else
      Fm |= (u64)f << t; Rm |= (u64)r << t;
#endif
    }
    u64 Qm = 0;       // positions nothing is found for anywhere unless their Hamming-1 sweep finds something (conflict_masks)
#if FQ_WAVE <= 1
    const u64 NCb = 0, NCs = 0;
#else
    // (both only concern positions stage P could not settle: a chunk whose lanes [j0, n) are all settled -- most chunks of covered
    // sequence -- goes straight on; with the gate open repair_kmers_missing may fire: per-position path)
    const u64 live = (n >= 64 ? ~0ull : (1ull << n) - 1ull) & (~0ull << j0);
    u64 NCb = 0, NCs = 0;   // positions whose local b- / s-mer look-up no pending insert can change (conflict_masks)
    if (pre && (live & ~Fm)) {
      TM_BEGIN(t_qm);
      local_refresh(w, j0, n);
      const ConfMasks cmk = conflict_masks(w, n, !w.repm_gate);
      Qm = cmk.quiet; NCb = cmk.ncb; NCs = cmk.ncs;
      TM_END(w, TX_QMM, t_qm);
    }
#endif
    // What the serial loop below reads about a position -- stage P's flags, the N run, the mailbox flags, the symbol and the
    // repair symbol -- is fetched ONCE per chunk, one position per lane, and then handed out by v_readlane: the loop's time is a
    // chain of dependent LDS round trips (a hundred-odd cycles each for a lone wave), and these were five of them per position.
#if FQ_WAVE > 1
    u32 pk = 0, pk2 = 4;
    {
      const u32 t = FQ_LANE;
      if (t < n) {
        pk = (u32)w.sb->sp_flag[t] | ((u32)w.sb->sx_flag[t] << 8) | ((u32)w.sb->sp_nrun[t] << 16) | ((u32)w.sb->pv_flag[t] << 24);
        pk2 = (size <= FQSX_RD_LDS && i + t < size ? (u32)w.rdp[i + t] : 4u) | ((u32)w.sb->sp_rep[t] << 8) | ((u32)w.sb->sp_scur[2][t] << 16);
      }
    }
#define PKJ(v, jj) ((u32)__builtin_amdgcn_readlane((int)(v), (int)(jj)))
#endif
    u32 q_done = j0;  // chunk positions whose mailbox entries are already in the lists
    u32 w_pos = j0;   // w's k-mers = state before position w_pos of the chunk
    u32 m = j0;       // committed positions (lanes [j0, m) of the chunk)
    bool dirty = false;  // corrected k-mers were modified: the speculation of the next bmer - 1 positions is stale
    bool foreseen = false;   // ... by the repair of a settled position, which a scout may have allowed for (scout_fix)
    for (u32 j = j0; j < n && !dirty && !w.err; ++j) {
      const u32 pos = i + j;
#if FQ_WAVE > 1
      const u32 pkj = PKJ(pk, j), pk2j = PKJ(pk2, j);
      const u32 sym = size <= FQSX_RD_LDS ? (pk2j & 0xffu) : rd_sym(w, p, pos, size);
      const u32 flag = pkj & 0xffu;
#else
      const u32 sym = rd_sym(w, p, pos, size);
      const u32 flag = w.sb->sp_flag[j];
#endif
      const u64 sym_k = sym == 4 ? 0 : sym;
#if FQ_WAVE > 1
      if ((Qm >> j) & 1) {
        TM_BEGIN(t_qr);
        // A stretch of positions where every look-up came up empty and the sweep did too: level none, the symbol is
        // letter-coded, stage P's mailbox entries stand (p-mer included), nothing draws from an RNG and no repair
        // can fire (dna.cpp:706-744,776-785,840-874) -- settled for the whole stretch in one lane-parallel step.
        u32 spins = 0, front;
        TM_BEGIN(t_rrw);
        while ((front = lds_load_acq(&w.sb->rr_front)) <= j) {   // the scout wave may still be sweeping this chunk
          fq_sleep();
          if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
        }
        TM_END(w, TM_RRWAIT, t_rrw);
        const u32 t = FQ_LANE;
        const u32 form = t < n ? w.sb->rr_idx[t] : 0xffu;
        const bool summed = form == 0xfd;   // ... or the sweep's hits merge without a draw (scout_rough): level pmer, rough
        const bool empty = t < front && t < n && ((Qm >> t) & 1) && ((form == 0xfe && w.sb->rc_hit[t] == 0) || summed);
        const u64 run = wave_ballot(empty) >> j;
        const u32 len = ~run ? ctz64(~run) : 64u;
        if (len) {
          const bool in = t >= j && t < j + len;
          u32 nsl = 0;
          if (in) {
            if (summed && w.sb->sp_nrun[t] < 2) {   // dna.cpp:737-775 with cor_zone 3 (rough)
              w.sb->sp_cq[t] = w.sb->rc_val[t][0];
              w.sb->sp_lvz[t] = (u8)(LV_PMER | (3u << 4));
              w.sb->sp_kind[t] = SK_RANK_PENDING;
            } else
              w.sb->sp_kind[t] = SK_LETTER_PENDING;
            const u32 pf = w.sb->pv_flag[t];
            if (pf & PV_PCAND) w.sb->pv_flag[t] = (u8)(pf | PV_P);
            nsl = w.sb->rc_ns[t];
          }
          FQ_SYNC();
          w.st[ST_GPROBE] += (u64)len * 4 * (cfg->gb.k - 1);
          w.st[ST_GSLOT] += wave_sum32(nsl) + (u64)len * (cfg->gb.k - 1);
#ifdef FQSX_TIMING
          w.tm[CN_SLOW] += len; w.tm[CN_EXT] += len; w.tm[CN_ROUGH] += len;
#endif
          j += len - 1;
          m = j + 1;
          TM_END(w, TX_QRUN, t_qr);
          continue;
        }
        TM_END(w, TX_QRUN, t_qr);
      }
#endif
      TM_BEGIN(t_code);
      if (flag == 1) {
        // settled by stage P (level bmer): nothing to resolve; skip the whole stretch of such positions up to the
        // first one whose repair fires (their symbols are coded in stage C2, code_chunk)
        const u64 fm = Fm >> j, rm = Rm >> j;
        const u32 nf = ~fm ? ctz64(~fm) : 64u;                                  // consecutive settled positions from j
        const u64 rr = nf >= 64 ? rm : rm & ((1ull << nf) - 1ull);
        const u32 len = rr ? ctz64(rr) + 1 : nf;
        j += len - 1;
        const u32 pos = i + j;
#if FQ_WAVE > 1
        const u32 pk2e = PKJ(pk2, j);
        const u32 sym = size <= FQSX_RD_LDS ? (pk2e & 0xffu) : rd_sym(w, p, pos, size);
        const u32 rep = (pk2e >> 8) & 0xffu;
#else
        const u32 sym = rd_sym(w, p, pos, size);
        const u32 rep = w.sb->sp_rep[j];
#endif
        const u64 sym_k = sym == 4 ? 0 : sym;
#ifdef FQSX_TIMING
        w.tm[CN_FAST] += len;
#endif
        if (rep != 0xff) {  // repair_kmers_existing fires (dna.cpp:362-369,856-863)
          load_state(w, j);
          replace_last_all(w, sym_k);
          flush_pushes(w, q_done, j + 1);
          q_done = j + 1;
          km_replace_last(w.pm, rep); km_replace_last(w.sm_, rep); km_replace_last(w.bm, rep);
          w.cor_pos = pos;
          w.N_run = sym == 4 ? w.N_run + 1 : 0;
          push_b_local(w);
          w_pos = j + 1;
          dirty = true;
          foreseen = true;
        }
      } else {
        // not settled by stage P alone
        C4 counts;
        c4_zero(counts);
#if FQ_WAVE > 1
        u32 level = LV_NONE, nrun_here = (pkj >> 16) & 0xffu;
        const u32 xf = flag == 3 ? (pkj >> 8) & 0xffu : 0u;
        const u32 scur2_j = (pk2j >> 16) & 0xffu;
#else
        u32 level = LV_NONE, nrun_here = w.sb->sp_nrun[j];
        const u32 xf = flag == 3 ? w.sb->sx_flag[j] : 0;
        const u32 scur2_j = w.sb->sp_scur[2][j];
#endif
        bool rough = false, loaded = false, resolved = false;
        TM_COUNT(w, CN_SLOW);
        const bool early_b = flag == 4;
        if (early_b) {
          // early position, b-mer almost full; the scout wave has probed its paddings in the global table and seen hits
          // (scout_settle_early): find_counts (dna.cpp:461-476) comes down to the merges, in trial order
          Kmer bmj;
          bmj.dir = bmj.rc = 0; bmj.cur = scur2_j;
          kt_find(w, cfg->g_b, true, cfg->gb, bmj, RNG_B, CINC_B, counts, w.sb->ep_off[j][0]);
          resolved = true;
          level = LV_BMER;
          if ((counts.c[0] == 63) + (counts.c[1] == 63) + (counts.c[2] == 63) + (counts.c[3] == 63) > 1) {   // dna.cpp:466-474
            flush_pushes(w, q_done, j);
            q_done = j;
            load_state(w, j);
            loaded = true;
            nrun_here = w.N_run;
            C4 c2;
            kt_find(w, cfg->g_s, true, cfg->gs, w.sm_, RNG_S, CINC_S, c2);
            counts.c[0] += c2.c[0]; counts.c[1] += c2.c[1]; counts.c[2] += c2.c[2]; counts.c[3] += c2.c[3];
            level = LV_MIXED;
          }
        }
        if (xf & SX_VALID) {
          // the cascade was resolved in stage P; it stands unless a pending local insert interferes
          Kmer bmj, smj;
          bmj.dir = bmj.rc = 0; bmj.cur = 0;
          // (a position of the conflict masks has no such entry in the group: conflict_masks; the k-mers are only fetched when asked for)
          const bool ask_b = !((NCb >> j) & 1), ask_s = !((NCs >> j) & 1) && !(xf & (SX_LB | SX_UNC | SX_S));
          if (ask_b || ask_s || !(xf & SX_HITS)) { bmj.dir = w.sb->sp_sdir[2][j]; bmj.rc = w.sb->sp_src[2][j]; bmj.cur = scur2_j; }   // (a conflict, or an empty cascade without a sweep probed ahead, needs the b-mer)
          bool conflict = ask_b ? pend_conflict(w, 0, cfg->gb, bmj, q_done, j) : false;
          if (!conflict && ask_s) {
            smj.dir = w.sb->sp_sdir[1][j]; smj.rc = w.sb->sp_src[1][j]; smj.cur = w.sb->sp_scur[1][j];
            conflict = pend_conflict(w, 1, cfg->gs, smj, q_done, j);
          }
          if (conflict) {
            // An entry still on its way into the local tables lies in the position's sibling group -- typically the
            // second of two overlapping reads of this worker.  The look-up that entry can change comes first in the
            // cascade (dna.cpp:478-483): bring the local b-mer table up to date, ask it again; a hit settles the level
            // without the rest of the per-position path (on a miss that path takes over as before).
            flush_pushes(w, q_done, j);
            q_done = j;
            lq_sync_for(w, MAIL_B, cfg->gb, bmj);
            if (kt_find(w, cfg->l_b, false, cfg->gb, bmj, RNG_LB, CINC_B, counts)) {
              level = LV_BMER;
              resolved = true;
            } else
              c4_zero(counts);
          } else {
            resolved = true;
            TM_COUNT(w, CN_EXT);
            if (xf & SX_LB) {
              u32 pc = w.sb->sx_lb[j];
              counts.c[0] = pc & 0xff; counts.c[1] = (pc >> 8) & 0xff; counts.c[2] = (pc >> 16) & 0xff; counts.c[3] = pc >> 24;
              level = LV_BMER;
            } else if (xf & SX_UNC) {
              // the uncorrected b-mer is known where the corrected one is not: the correction is dropped (dna.cpp:697-705)
              const u64 pc = w.sb->sx_s[j];
              counts.c[0] = (u32)(pc & 0xffff); counts.c[1] = (u32)((pc >> 16) & 0xffff);
              counts.c[2] = (u32)((pc >> 32) & 0xffff); counts.c[3] = (u32)(pc >> 48);
              flush_pushes(w, q_done, j);
              q_done = j;
              load_state(w, j);
              loaded = true;
              nrun_here = w.N_run;
              w.bm = w.bm_u; w.sm_ = w.sm_u; w.pm = w.pm_u;
              w.cor_pos = 0;
              level = LV_BMER;
              dirty = true;
            } else if (xf & (SX_S | SX_LS)) {
              u64 pc = (xf & SX_S) ? w.sb->sx_s[j] : w.sb->sx_ls[j];
              counts.c[0] = (u32)(pc & 0xffff); counts.c[1] = (u32)((pc >> 16) & 0xffff);
              counts.c[2] = (u32)((pc >> 32) & 0xffff); counts.c[3] = (u32)(pc >> 48);
              level = LV_SMER;
            } else {
              TM_BEGIN(t_r);
              TM_COUNT(w, CN_ROUGH);
              {   // the scout wave may still be sweeping this chunk
                u32 spins = 0;
                TM_BEGIN(t_rrw);
                while (lds_load_acq(&w.sb->rr_front) <= j) {
                  fq_sleep();
                  if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
                }
                TM_END(w, TM_RRWAIT, t_rrw);
              }
              const u32 rr = w.sb->rr_idx[j];
              if (rr != 0xff) rough = rough_merge_pre(w, rr, j, cfg->gb, RNG_B, CINC_B, counts);
              else rough = rough_kt(w, cfg->g_b, cfg->gb, bmj, RNG_B, CINC_B, counts);  // dna.cpp:711-718
              if (rough) level = LV_PMER;
              TM_END(w, TM_ROUGH, t_r);
            }
          }
        }
        if (flag == 0) TM_COUNT(w, CN_EARLY);
        if (!resolved) {
          TM_COUNT(w, CN_GENERIC);
#ifdef FQSX_TIMING
          if (w.sb->sp_scur[2][j] != cfg->gb.k) w.tm[TX_G_EARLY] += 1;
#endif
          // complete reference logic on the exact k-mers of this position
          flush_pushes(w, q_done, j);
          q_done = j;
          load_state(w, j);
          loaded = true;
          nrun_here = w.N_run;
          TM_BEGIN(t_fc);
          level = find_counts(w, counts, flag == 3, j);
          TM_END(w, TM_FINDC, t_fc);
          if (level == LV_BMER_UNC) {
            w.bm = w.bm_u; w.sm_ = w.sm_u; w.pm = w.pm_u;
            w.cor_pos = 0;
            level = LV_BMER;
            dirty = true;
          }
          if (level == LV_NONE) {
            TM_BEGIN(t_r);
            TM_COUNT(w, CN_ROUGH);
            if (km_full(w.bm, cfg->gb)) {
              if (rough_kt(w, cfg->g_b, cfg->gb, w.bm, RNG_B, CINC_B, counts)) { level = LV_PMER; rough = true; }
            } else if (km_full(w.sm_, cfg->gs)) {
              if (rough_kt(w, cfg->g_s, cfg->gs, w.sm_, RNG_S, CINC_S, counts)) { level = LV_PMER; rough = true; }
            } else if (km_full(w.pm, cfg->gp)) {
              if (rough_p(w, counts)) { level = LV_PMER; rough = true; }
            }
            TM_END(w, TM_ROUGH, t_r);
          }
        }
        TM_END(w, TM_SPRE, t_code);
        // code the symbol (dna.cpp:737-801)
        if (level != LV_NONE && nrun_here < 2) {
          int cor_dist = level == LV_PMER ? (int)cfg->pmer : level == LV_SMER ? (int)cfg->smer : (int)cfg->bmer;
          int d = (int)pos - (int)w.cor_pos;
          u32 cor_zone = d < cor_dist ? (u32)(1 + 2 * (cor_dist - d) / cor_dist) : 0u;
          if (rough) cor_zone = 3;
          FQ_SYNC();
          if (FQ_LANE == 0) {   // keys, rank and coding: stage C2
            w.sb->sp_cq[j] = (u64)(counts.c[0] & 0xffff) | ((u64)(counts.c[1] & 0xffff) << 16) | ((u64)(counts.c[2] & 0xffff) << 32) | ((u64)(counts.c[3] & 0xffff) << 48);
            w.sb->sp_lvz[j] = (u8)(level | (cor_zone << 4));
            w.sb->sp_kind[j] = SK_RANK_PENDING;
          }
          FQ_SYNC();
        } else {
          FQ_SYNC();
          if (FQ_LANE == 0) w.sb->sp_kind[j] = SK_LETTER_PENDING;
          FQ_SYNC();
        }
        // mailbox entries and context repair (dna.cpp:803-874)
        const bool lvl_sbm = level == LV_SMER || level == LV_BMER || level == LV_MIXED;
        if (loaded) {
          if (sym == 4) ++w.N_run; else w.N_run = 0;
          replace_last_all(w, sym_k);
          u32 pf = 0;
          if (sym < 4) {
            bool pmer_insert = true;
            if (km_full(w.bm, cfg->gb)) {
              pf |= PV_B;
              if (lvl_sbm && c4_get(counts, sym) >= 3) pmer_insert = false;
            }
            if (km_full(w.sm_, cfg->gs)) pf |= PV_S;
            if (km_full(w.pm, cfg->gp) && pos - w.cor_pos >= cfg->pmer - 1) pf |= pmer_insert ? PV_P : PV_PHID;
          }
          FQ_SYNC();
          if (FQ_LANE == 0) {
            w.sb->pv_b[j] = km_norm(w.bm, cfg->gb);
            w.sb->pv_s[j] = km_norm(w.sm_, cfg->gs);
            w.sb->pv_pd[j] = km_aligned_dir(w.pm);
            w.sb->pv_pr[j] = km_aligned_rc(w.pm);
            w.sb->pv_flag[j] = (u8)pf;
          }
          FQ_SYNC();
          w_pos = j + 1;
        } else {
          // stage P's entries stand (k-mers unmodified); settle the p-mer entry, which depends on the level
#if FQ_WAVE > 1
          u32 pf = pkj >> 24;
#else
          u32 pf = w.sb->pv_flag[j];
#endif
          if (pf & PV_PCAND) {
            pf |= (!early_b && lvl_sbm && c4_get(counts, sym) >= 3) ? PV_PHID : PV_P;   // (hidden only under a full b-mer)
            FQ_SYNC();
            if (FQ_LANE == 0) w.sb->pv_flag[j] = (u8)pf;
            FQ_SYNC();
          }
        }
        // repairs need the exact k-mers only when they can fire
        const bool b_full = loaded ? km_full(w.bm, cfg->gb) : !early_b;
        if (b_full) {
          const bool chk_existing = level == LV_BMER || level == LV_MIXED;
          const bool chk_missing = (level == LV_NONE || level == LV_PMER) && w.repm_gate;
          const bool fire = chk_existing ? repair_decide(w, counts, sym) != 0xff : chk_missing;
          if (fire) {
            if (!loaded) {
              load_state(w, j);
              if (sym == 4) ++w.N_run; else w.N_run = 0;
              replace_last_all(w, sym_k);
              w_pos = j + 1;
            }
            flush_pushes(w, q_done, j + 1);
            q_done = j + 1;
            bool rep;
            if (chk_existing) rep = repair_existing(w, pos, counts, sym);
            else {
              TM_BEGIN(t_rm);
              TM_COUNT(w, CN_REPM);
              rep = repair_missing(w, pos);
              TM_END(w, TM_REPM, t_rm);
            }
            if (rep) {
              push_b_local(w);
              dirty = true;
            }
          }
        }
        TM_END(w, TM_SLOW, t_code);
      }
      m = j + 1;
    }
    if (dirty) TM_COUNT(w, CN_DIRTY);
    { TM_BEGIN(t_fl); flush_pushes(w, q_done, m); TM_END(w, TX_FLUSH, t_fl); }
    code_chunk(w, p, size, i, j0, m, reversed, hist_start, first);
    lq_publish(w);   // (after the queue hand-off, whose release has already drained the list stores)
    if (w_pos != m) {  // the last committed position went through the fast path: materialise its state
      const u32 sym = rd_sym(w, p, i + m - 1, size);
      load_state(w, m - 1);
      replace_last_all(w, sym == 4 ? 0 : sym);
      w.N_run = sym == 4 ? w.N_run + 1 : 0;
    }
    at = i + m;
    if (dirty || (dbg & FQSX_DBG_RESTART)) {   // (test switch: the window after every chunk, corrected or not)
      own_lo = at;
      own_hi = at + cfg->bmer - 1;
      if (pre && dirty && foreseen && !(dbg & FQSX_DBG_RESTART) && w.sb->h_fix_lane == m - 1) {   // the scout has the window's lanes of this chunk ready
        own_lo = i + w.sb->h_fix_end;
        fixed = w.sb;
        fixed_pub = w.sb->h_pub;
        if (w.sb->h_fix_end > n) n = w.sb->h_fix_end;   // (the chunk goes on into its spare lanes: not finished yet)
      }
    }
    if (pre && m == n) scout_release(w);   // (a chunk left in its middle stays the head of the ring: the position after the window may lie in it)
  }
}

// the head of a read: duplicate flag and prefix.  Returns true for a duplicate (nothing else is coded);
// hist = the read's letter counts A, C, G, T.
FQ_DEV bool read_head(Wk &w, const u8 *p, u32 size, const u8 *prev, u32 prev_size, bool first_of_pair, u32 hist[4]) {
  const DevCfg *cfg = w.cfg;
  const bool orig = !first_of_pair || w.mode == 0 || w.mode == 2;
  // duplicate test + staging of the read's codes in LDS + letter histogram, all lane-parallel
  bool diff = prev == nullptr || prev_size != size;
  u32 h0 = 0, h1 = 0, h2 = 0, h3 = 0;
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < size; i += FQ_WAVE) {
    u8 ch = p[i];
    if (!diff && prev[i] != ch) diff = true;
    u32 c = dna_code(ch);
    if (i < FQSX_RD_LDS) w.rdp[i] = (u8)c;
    h0 += c == 0; h1 += c == 1; h2 += c == 2; h3 += c == 3;
  }
  FQ_SYNC();
  hist[0] = wave_sum32(h0); hist[1] = wave_sum32(h1); hist[2] = wave_sum32(h2); hist[3] = wave_sum32(h3);
  bool same = !wave_any(diff);
  if (prev == nullptr || prev_size != size) same = false;
  if (first_of_pair) {
    u16 *m = small_base(w) + SM_OFF_FLAGS + w.ws->ctx_flags * (SM_FLAGS_N + 1);
    sm_encode(w, m, SM_FLAGS_N, 1u << 12, same ? 1u : 0u);
    w.ws->ctx_flags = ((w.ws->ctx_flags << 1) + (same ? 1u : 0u)) & 0xff;
    if (same) return true;
  }
  km_reset(w.pm); km_reset(w.sm_); km_reset(w.bm);
  km_reset(w.pm_u); km_reset(w.sm_u); km_reset(w.bm_u);
  w.cor_pos = 0;
  w.N_run = 0;
  if (w.rec) {   // (read-head wave: part of what prefix_sorted hands to the scout waves early)
    FQ_SYNC();
    if (FQ_LANE == 0) for (u32 x = 0; x < 4; ++x) w.rec->hist[x] = hist[x];
  }
  if (orig) prefix_direct(w, p, size); else prefix_sorted(w, p, size);
  return false;
}
FQ_DEV void add_s_letters(Wk &w, const u32 hist[4]) {  // update_s_letters, dna.cpp:2047-2057 (both strands)
  w.s_let[0] += hist[0] + hist[3]; w.s_let[3] += hist[0] + hist[3];
  w.s_let[1] += hist[1] + hist[2]; w.s_let[2] += hist[1] + hist[2];
}
// CompressDirect / CompressSorted, dna.cpp:1517-1556,1716-1754.  `prev` is the previous read of
// this worker inside the block (read_prev is cleared per block, application.cpp:624), or null.
// first_of_pair = false: the second mate coded by CompressDirect(..., false): direct prefix, no duplicate flag
FQ_DEV void compress_read(Wk &w, const u8 *p, u32 size, const u8 *prev, u32 prev_size, bool first_of_pair = true) {
  const bool orig = !first_of_pair || w.mode == 0 || w.mode == 2;
  u32 hist[4];
  TM_BEGIN(t_head);
  const bool same = read_head(w, p, size, prev, prev_size, first_of_pair, hist);
  TM_END(w, TM_READ_HEAD, t_head);
  if (same) return;
  suffix(w, p, size, orig);
  add_s_letters(w, hist);
  w.st[ST_BASES] += size;
}
// The same with the head taken from the read-head wave's record `idx` (single-end sorted mode of the encode kernel)
FQ_DEV void compress_read_rec(Wk &w, const u8 *p, u32 size, u32 idx, bool has_next) {
  WgShared *sm = w.sm;
  TM_BEGIN(t_head);
  u32 spins = 0;
  while ((i32)(lds_load_acq(&sm->hd_ready) - idx) <= 0) {
    fq_sleep();
    if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; return; }   // never spin forever on the GPU
  }
  const HeadRec *rec = &sm->hd[idx % FQSX_HD];
  const u32 n_raw = rec->n_raw;
  const bool same = rec->same != 0;
  // the head's symbols take their place in the stream
  if (cq_wait_space(w, n_raw)) {
    for (u32 j = FQ_LANE; j < n_raw; j += FQ_WAVE) {
      const u32 e = (w.cq_tail + j) & (FQSX_CQ - 1);
      sm->cq_key[0][e] = rec->raw[j][0];
      sm->cq_key[1][e] = rec->raw[j][1];
      sm->cq_kind[e] = SK_RAW;
      sm->cq_rsym[e] = 5;
    }
    FQ_SYNC();
    cq_publish(w, n_raw);
  }
  if (!same) {
    mail_push(w, MAIL_P, rec->pmail[0]);
    mail_push(w, MAIL_P, rec->pmail[1]);
    w.pm.dir = rec->kdir[0]; w.pm.rc = rec->krc[0]; w.pm.cur = rec->kcur[0];
    w.sm_.dir = rec->kdir[1]; w.sm_.rc = rec->krc[1]; w.sm_.cur = rec->kcur[1];
    w.bm.dir = rec->kdir[2]; w.bm.rc = rec->krc[2]; w.bm.cur = rec->kcur[2];
    w.pm_u = w.pm; w.sm_u = w.sm_; w.bm_u = w.bm;
    w.cor_pos = 0;
    w.N_run = rec->n_run;
    w.rdp = sm->rd[idx % FQSX_HD];
    w.sc_read = idx;
    w.sc_abandoned = false;
    u32 hist[4] = {rec->hist[0], rec->hist[1], rec->hist[2], rec->hist[3]};
    TM_END(w, TM_READ_HEAD, t_head);
    suffix(w, p, size, false);
    add_s_letters(w, hist);
    w.st[ST_BASES] += size;
    // the read was (partly) resolved without the scout waves: they take up again at the head of the next one
    // (posted while this read's record still counts as in use, so that the records they need are in place)
    if (w.sc_abandoned && has_next) scout_restart(w, idx + 1, w.cfg->pmer, w.s_let, SCQ_FROM_HEAD, false);
  }
  FQ_SYNC();
  lds_store_rel(&sm->hd_taken, idx + 1);   // the record and its staging buffer are free again
}

#include "fqsx_pe.h"
#include "fqsx_dec.h"

// ---------------------------------------------------------------------------------------
// kernel bodies

// worker `tid` codes its reads of segment `seg` (application.cpp:610-656)
// The coder wave of the two-wave encode kernel: drains the coding queue while the other wave of the workgroup
// resolves the reads.  Nothing the resolving wave decides depends on the context models, the level averages or the
// range coder, so this wave owns them (and the worker's output stream) outright; the hand-off is the LDS queue.
// SPLIT (six-wave kernel): this wave runs the context models and the level averages only and hands every coding step
// to the range-coder wave (rc_segment_body), which owns the coder state and the output stream.
template <bool SPLIT>
FQ_DEV void coder_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 seg, u32 launch = 0) {
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = 0;   // (unused by the coder)
  WState *ws = cfg.ws + tid;
  w.ws = ws;
  w.err = 0;
  w.piped = false;   // this wave codes directly
  w.lqh = false;
  w.rec = nullptr;
  w.rdp = sm->rd[0];
  w.cq_head = w.cq_tail = 0;
  w.c_r_sym = 0;
  w.sb = &sm->sb[0];
  w.scout = false;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  w.rcq = SPLIT;
  w.rq_tail = 0;
  if (!SPLIT) {
    if (seg == 0) enc_open(w, 0, 0xff00000000000000ULL, 0, cfg);   // application.cpp:624-628
    else enc_open(w, ws->rc_low, ws->rc_range, ws->out_len, cfg);
  }
  w.avg_code = ws->avg_code; w.avg_letters = ws->avg_letters;
  u32 head = 0, spins = 0;
  TM_BEGIN(t_all);
  for (;;) {
    const u32 done = lds_load_acq(&sm->cq_done);   // read before the tail: once set, the tail is final
    const u32 tail = lds_load_acq(&sm->cq_tail);
    if (tail != head) {
      TM_BEGIN(t_c);
      const u32 head0 = head;
      head += cq_process(w, head, tail - head);
      (void)head0;
      WHATIF(cfg, 2, (head >> 6) - (head0 >> 6));   // (per 64 queue entries ~ per chunk)
      FQ_SYNC();
      lds_store_rel(&sm->cq_head, head);
      TM_END(w, TM_FAST, t_c);
      spins = 0;
    } else if (done) break;
    else {
      TM_BEGIN(t_idle);
      fq_sleep();
      if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
      TM_END(w, TM_CODER_IDLE, t_idle);
    }
  }
  TM_END(w, TM_P2, t_all);   // launch start to the last symbol coded (compare with the resolving wave's TM_TOTAL)
  TM_STAMP(cfg, tid, launch, 3);
  if (SPLIT) {
    FQ_SYNC();
    lds_store_rel(&sm->rq_done, 1u);   // everything is queued: let the range-coder wave finish
  } else {
    enc_close(w);
    ws->rc_low = w.enc.low; ws->rc_range = w.enc.range; ws->out_len = w.enc.len;
  }
  ws->avg_code = w.avg_code; ws->avg_letters = w.avg_letters;
  if (FQ_LANE == 0) {
    for (u32 i = 0; i < ST_N; ++i) if (w.st[i]) atomic_add64(&ws->stat[i], w.st[i]);
#ifdef FQSX_TIMING
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) if (w.tm[i]) atomic_add64(&ws->stat[16 + i], w.tm[i]);
#endif
  }
  if (w.err) *cfg.err = w.err;
}

// The range-coder wave of the six-wave kernel: the strictly sequential tail of the pipeline -- one coding step per
// symbol (Encode, sub_rc.h:60-77) on wave-uniform state, byte output -- fed with finished (freq, cum, total, reciprocal)
// entries through the LDS queue.  It owns the coder state and the worker's output stream.
FQ_DEV void rc_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 seg, u32 launch = 0) {
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = 0;
  WState *ws = cfg.ws + tid;
  w.ws = ws;
  w.err = 0;
  w.piped = false;
  w.lqh = false;
  w.rec = nullptr;
  w.rcq = false;
  w.scout = false;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  if (seg == 0) enc_open(w, 0, 0xff00000000000000ULL, 0, cfg);   // application.cpp:624-628
  else enc_open(w, ws->rc_low, ws->rc_range, ws->out_len, cfg);
  u32 head = 0, spins = 0;
  for (;;) {
    const u32 done = lds_load_acq(&sm->rq_done);   // read before the tail: once set, the tail is final
    const u32 tail = lds_load_acq(&sm->rq_tail);
    if (tail != head) {
      TM_BEGIN(t_rc);
      const u32 n = tail - head < FQ_WAVE ? tail - head : FQ_WAVE;
#if FQ_WAVE > 1
      // one entry per lane, then the steps in order from the lanes' registers
      const u32 e = (head + (FQ_LANE < n ? FQ_LANE : 0u)) & (FQSX_RQ - 1);
      const u32 lf = sm->rq_f[e], lc = sm->rq_c[e], lt = sm->rq_t[e];
      const u64 lm = sm->rq_m[e];
      for (u32 t = 0; t < n; ++t) {
        const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(lm >> 32), t) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)lm, t);
        rc_encode_m(w, (u32)__builtin_amdgcn_readlane((int)lf, t), (u32)__builtin_amdgcn_readlane((int)lc, t),
                    (u32)__builtin_amdgcn_readlane((int)lt, t), m);
      }
#else
      for (u32 t = 0; t < n; ++t) {
        const u32 e = (head + t) & (FQSX_RQ - 1);
        rc_encode_m(w, sm->rq_f[e], sm->rq_c[e], sm->rq_t[e], sm->rq_m[e]);
      }
#endif
      WHATIF(cfg, 3, ((head + n) >> 6) - (head >> 6));
      head += n;
      FQ_SYNC();
      lds_store_rel(&sm->rq_head, head);
      TM_END(w, TM_CR_RC, t_rc);
      spins = 0;
    } else if (done) break;
    else {
      fq_sleep();
      if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
    }
  }
  TM_STAMP(cfg, tid, launch, 7);
  enc_close(w);
  ws->rc_low = w.enc.low; ws->rc_range = w.enc.range; ws->out_len = w.enc.len;
  if (FQ_LANE == 0) {
    for (u32 i = 0; i < ST_N; ++i) if (w.st[i]) atomic_add64(&ws->stat[i], w.st[i]);
#ifdef FQSX_TIMING
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) if (w.tm[i]) atomic_add64(&ws->stat[16 + i], w.tm[i]);
#endif
  }
  if (w.err) *cfg.err = w.err;
}

// The inserter wave of the encode kernel: applies the worker's b-/s-mer list entries to its local tables, in list
// order, as soon as the resolving wave publishes them.  It owns the local tables' contents, their fill counters and
// the two local counter RNG streams while it has work; the resolving wave touches those only after it has seen
// lq_done catch up (lq_flush), which it also does once more at the end of the segment.
FQ_DEV void inserter_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 launch = 0) {
  u32 la[2] = {0, 0}, err = 0, spins = 0;
  u64 ns = 0, lins = 0;
  bool failed = false;
  for (;;) {
    const u32 quit = lds_load_acq(&sm->lq_quit);
    if (quit) break;
    bool worked = false;
    for (u32 qi = 0; qi < 2 && !failed; ++qi) {
      const u32 kind = qi ? MAIL_S : MAIL_B;
      const u32 tgt = lds_load_acq(&sm->lq_target[qi]);
      if (tgt > la[qi]) {
        const Mail &m = cfg.mail[kind];
        const u32 n = tgt - la[qi] < FQ_WAVE ? tgt - la[qi] : FQ_WAVE;
        WHATIF(cfg, 6, 1);
        insert_batch(cfg, sm, qi ? cfg.l_s : cfg.l_b, tid, m.list + (u64)tid * m.cap + la[qi], n, qi ? RNG_LS : RNG_LB,
                     qi ? CINC_S : CINC_B, ns, err);
        if (err) {   // table full: report, and release every waiter
          failed = true;
          *cfg.err = FQSX_ERR_LTAB_FULL;
          lds_store_rel(&sm->lq_done[0], ~0u);
          lds_store_rel(&sm->lq_done[1], ~0u);
          break;
        }
        la[qi] += n;
        lins += n;
        FQ_SYNC_MEM();
        lds_store_rel(&sm->lq_done[qi], la[qi]);
        worked = true;
      }
    }
    if (worked) spins = 0;
    else {
      fq_sleep();
      if (spin_expired(spins)) break;   // never spin forever on the GPU
    }
  }
  if (FQ_LANE == 0 && lins) atomic_add64(&cfg.ws[tid].stat[ST_LINS], lins);
  TM_STAMP(cfg, tid, launch, 5);
}

// The read-head wave of the encode kernel (single-end sorted mode): for every read of the segment, one read ahead of
// the resolving wave, the duplicate test, the staging of the read's codes in LDS, the p-mer prefix (rank sweep over
// the p-mer vector, which only changes between launches) and the arithmetic of the head's small models -- everything
// that depends on the reads alone.  It owns the head's contexts (flag histories, previous p-mer) and small models;
// the symbols reach the coder wave as finished triples via the resolving wave, which keeps the stream order.
FQ_DEV void head_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 n_reads, u32 S, u32 seg, u32 launch = 0) {
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = 1;   // single-end sorted: the only mode with a read-head wave
  WState *ws = cfg.ws + tid;
  w.ws = ws;
  w.err = 0;
  w.piped = false;
  w.lqh = false;
  w.rcq = false;
  w.scout = false;
  w.sb = &sm->sb[0];
  w.cq_head = w.cq_tail = 0;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  const u64 T = cfg.T;
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;   // PartitionForWorkers, reads_block.h:197-214
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  const u64 cur = seg == 0 ? first : ws->cursor;
  u64 stop = last;
  if (seg < S) stop = ((u64)seg + 1) * (last - first) / ((u64)S + 1) + first + 1;   // application.cpp:643
  if (stop > last) stop = last;
  WHATIF(cfg, 7, 1);   // (once per launch, ahead of everything: the calibration of the what-if profile)
  for (u64 i = cur; i < stop && !w.err; ++i) {
    const u32 idx = (u32)(i - cur);
    u32 spins = 0;
    for (;;) {   // all records in use: by the resolving wave, by a scout wave, or by a restart request not yet taken up
      bool busy = (i32)(idx - lds_load_acq(&sm->hd_taken)) >= (i32)FQSX_HD;   // (read first: a request is posted before hd_taken moves on)
      if (!lds_load_acq(&sm->sc_dead)) {
        const u32 req = lds_load_acq(&sm->sc_req_seq);
        for (u32 x = 0; x < FQSX_NSC; ++x) busy |= lds_load_acq(&sm->sc_ack[x]) != req || (i32)(idx - lds_load_acq(&sm->sc_hd_taken[x])) >= (i32)FQSX_HD;
      }
      if (!busy) break;
      fq_sleep();
      if (spin_expired(spins)) { w.err = FQSX_ERR_PIPE; break; }   // never spin forever on the GPU
    }
    if (w.err) break;
    HeadRec *rec = &sm->hd[idx % FQSX_HD];
    w.rec = rec;
    w.rec_idx = idx;
    w.rdp = sm->rd[idx % FQSX_HD];
    FQ_SYNC();
    if (FQ_LANE == 0) { rec->idx = ~0u; rec->n_raw = 0; rec->n_p = 0; }
    FQ_SYNC();
    const u64 o0 = cfg.read_off[i], o1 = cfg.read_off[i + 1];
    const u8 *prev = nullptr;
    u32 prev_size = 0;
    if (i > first) {
      const u64 q0 = cfg.read_off[i - 1];
      prev = cfg.bases + q0;
      prev_size = (u32)(o0 - q0);
    }
    u32 hist[4];
    WHATIF(cfg, 5, 1);
    const bool same = read_head(w, cfg.bases + o0, (u32)(o1 - o0), prev, prev_size, true, hist);
    FQ_SYNC();
    if (FQ_LANE == 0) {
      rec->same = same ? 1u : 0u;
      rec->n_run = w.N_run;
      for (u32 x = 0; x < 4; ++x) rec->hist[x] = hist[x];
      rec->kdir[0] = w.pm.dir; rec->krc[0] = w.pm.rc; rec->kcur[0] = w.pm.cur;
      rec->kdir[1] = w.sm_.dir; rec->krc[1] = w.sm_.rc; rec->kcur[1] = w.sm_.cur;
      rec->kdir[2] = w.bm.dir; rec->krc[2] = w.bm.rc; rec->kcur[2] = w.bm.cur;
      rec->idx = idx;
    }
    FQ_SYNC();
    lds_store_rel(&sm->hd_early, idx + 1);   // (a duplicate read has no prefix: its record is complete only here)
    lds_store_rel(&sm->hd_ready, idx + 1);
  }
  if (FQ_LANE == 0) {
    for (u32 i = 0; i < ST_N; ++i) if (w.st[i]) atomic_add64(&ws->stat[i], w.st[i]);
  }
  if (w.err) *cfg.err = w.err;
  TM_STAMP(cfg, tid, launch, 1);
}

// The scout waves of the encode kernel (single-end sorted mode): stage P of every chunk of every read, ahead of the
// resolving wave.  The rolling k-mers at any position of a read follow in closed form from the read's symbols and
// a base state -- the state after the read's prefix (head record) or, after a k-mer correction, the exact state the
// resolving wave posts (ScoutReq) -- so the chunks are independent of each other and FQSX_NSC waves make them in
// turn: all waves walk the same enumeration of (read, chunk), chunk number c of an epoch is made by wave c % FQSX_NSC
// in ring slot c % FQSX_SCR and published through the slot's own word (h_pub).  A restart starts a new epoch: every
// wave acknowledges, and none writes a chunk of the new epoch before all have (a slot is never written by two).
// A chunk records which local-list entries were already applied when its probes started, so the resolving wave's
// validation of the local probes (pend_conflict) covers exactly the entries the scout may have missed.
FQ_DEV void scout_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 n_reads, u32 S, u32 seg, u32 me, u32 launch = 0) {
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = 1;
  WState *ws = cfg.ws + tid;
  w.ws = ws;
  w.err = 0;
  w.piped = false;
  w.rcq = false;
  w.lqh = true;    // (reads the inserter wave's progress)
  w.sc_poll = true;
  w.nsc = FQSX_NSC;
  w.rec = nullptr;
  w.scout = false;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  for (u32 i = 0; i < 4; ++i) w.s_let[i] = ws->s_letters[i];
  const u64 T = cfg.T;
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;   // PartitionForWorkers, reads_block.h:197-214
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  const u64 cur = seg == 0 ? first : ws->cursor;
  u64 stop = last;
  if (seg < S) stop = ((u64)seg + 1) * (last - first) / ((u64)S + 1) + first + 1;   // application.cpp:643
  if (stop > last) stop = last;
  bool quit = false;
  w.sc_epoch = 0;
  const u32 n_seg = (u32)(stop > cur ? stop - cur : 0);
  u32 idx = 0;          // read the wave is at (index within the launch)
  u32 seq = 0;          // number, within the epoch, of the next chunk of the enumeration
  u32 sbase = 0;        // ring slot of the epoch's chunk 0 (a restart names it; chunk c: slot and wave (sbase + c) % nsc)
  bool restart = false; // a request of the resolving wave is to be taken up (sc_req)
  while (!quit) {
    u32 spins = 0;
    u32 base_pos = cfg.pmer;   // the k-mers in w stand before this position of the read
    if (lds_load_acq(&sm->sc_dead)) break;
    if (idx >= n_seg && !restart) {
      // every read has its chunks; stay until the resolving wave has finished the last read (it may still ask for a restart)
      if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) { restart = true; continue; }
      if ((i32)(lds_load_acq(&sm->hd_taken) - n_seg) >= 0 || lds_load_acq(&sm->cq_done) || lds_load_acq(&sm->sc_dead)) break;
      TM_BEGIN(t_id);
      fq_sleep();
      TM_END(w, TM_SC_IDLE, t_id);
      continue;
    }
    if (!restart) {
      TM_BEGIN(t_w1);
      while ((i32)(lds_load_acq(&sm->hd_early) - idx) <= 0) {
        if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) { restart = true; break; }
        fq_sleep();
        if (lds_load_acq(&sm->cq_done) || lds_load_acq(&sm->sc_dead) || spin_expired(spins)) { quit = true; break; }
      }
      TM_END(w, TM_SCOUT_WAIT, t_w1);
      if (quit) break;
    }
    if (!restart && lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) restart = true;
    bool from_head = true;
    if (restart) {   // take up the request
      restart = false;
      // every scout wave owns one ring slot (chunk c: wave and slot c % nsc): nothing to agree on, go right on
      w.sc_epoch = lds_load_acq(&sm->sc_req_seq);
      const WgShared::ScoutReq &q = sm->sc_req;
      idx = uniform32(q.read);
      base_pos = uniform32(q.i0);
      sbase = uniform32(q.slot_base);
      seq = 0;
      from_head = (uniform32(q.flags) & SCQ_FROM_HEAD) != 0;
      if (!from_head) {
        Kmer *k[6] = {&w.pm, &w.sm_, &w.bm, &w.pm_u, &w.sm_u, &w.bm_u};
        for (u32 x = 0; x < 6; ++x) { k[x]->dir = uniform64(q.kdir[x]); k[x]->rc = uniform64(q.krc[x]); k[x]->cur = uniform32(q.kcur[x]); }
        w.cor_pos = uniform32(q.cor_pos);
        w.N_run = uniform32(q.n_run);
      }
      for (u32 x = 0; x < 4; ++x) w.s_let[x] = uniform64(q.s_let[x]);
      if (idx >= n_seg) continue;   // (cannot happen)
      // The records of this read and the next one are in place while the resolving wave counts as inside the read the
      // request was posted from (hd_taken); from now on this wave's own progress word protects them again.
      FQ_SYNC();
      lds_store_rel(&sm->sc_hd_taken[me], idx);
      lds_store_rel(&sm->sc_ack[me], w.sc_epoch);   // (the read-head wave keeps the records until every scout is here)
      if (from_head) {   // the record may not be there yet
        spins = 0;
        while ((i32)(lds_load_acq(&sm->hd_early) - idx) <= 0) {
          if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) { restart = true; break; }
          fq_sleep();
          if (lds_load_acq(&sm->cq_done) || lds_load_acq(&sm->sc_dead) || spin_expired(spins)) { quit = true; break; }
        }
        if (quit) break;
        if (restart) continue;
      }
    }
    const HeadRec *rec = &sm->hd[idx % FQSX_HD];
    const bool rec_same = uniform32(rec->same) != 0;   // (wave-uniform values: kept in scalar registers)
    const u32 hist[4] = {uniform32(rec->hist[0]), uniform32(rec->hist[1]), uniform32(rec->hist[2]), uniform32(rec->hist[3])};
    const u64 hk_dir[3] = {uniform64(rec->kdir[0]), uniform64(rec->kdir[1]), uniform64(rec->kdir[2])};
    const u64 hk_rc[3] = {uniform64(rec->krc[0]), uniform64(rec->krc[1]), uniform64(rec->krc[2])};
    const u32 hk_cur[3] = {uniform32(rec->kcur[0]), uniform32(rec->kcur[1]), uniform32(rec->kcur[2])}, h_nrun = uniform32(rec->n_run);
    FQ_SYNC();
    if (uniform32(rec->idx) != idx) {   // the record has been reused: this wave took up a request after the resolving wave had left the read
      lds_store_rel(&sm->sc_dead, 1u);
      for (u32 x = 0; x < FQSX_NSC; ++x) lds_store_rel(&sm->sc_hd_taken[x], 0x7fffffffu);   // (the read-head wave no longer waits for the scouts)
      break;
    }
    if (!rec_same) {
      const u64 ri = cur + idx;
      const u64 o0 = cfg.read_off[ri], o1 = cfg.read_off[ri + 1];
      const u8 *p = cfg.bases + o0;
      const u32 size = (u32)(o1 - o0);
      w.rdp = sm->rd[idx % FQSX_HD];
      if (from_head) {
        w.pm.dir = hk_dir[0]; w.pm.rc = hk_rc[0]; w.pm.cur = hk_cur[0];
        w.sm_.dir = hk_dir[1]; w.sm_.rc = hk_rc[1]; w.sm_.cur = hk_cur[1];
        w.bm.dir = hk_dir[2]; w.bm.rc = hk_rc[2]; w.bm.cur = hk_cur[2];
        w.pm_u = w.pm; w.sm_u = w.sm_; w.bm_u = w.bm;
        w.cor_pos = 0;
        w.N_run = h_nrun;
      }
      w.sc_read = idx;
      for (u32 i0 = base_pos, n = 0; i0 < size && !quit && !restart && (n = chunk_len(size - i0)) != 0; i0 += n, ++seq) {
        if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) { restart = true; break; }
        if ((sbase + seq) % FQSX_NSC != me) continue;   // another wave's chunk
        // stage P into registers first (one position per lane), the wait for the ring slot after it: the wave is a chunk
        // ahead of its slot, i.e. the ring is one chunk per scout deeper than its three buffers
        SpecLane spl;
        SpecHead sph;
        TM_BEGIN(t_sp);
        WHATIF(cfg, 4, 1);
        const bool whole = speculate_t<true>(w, p, size, i0, n, false, i0 - base_pos, 0, &spl, &sph);
        TM_END(w, TM_SC_SPEC, t_sp);
        if (!whole) { TM_COUNT(w, CN_SC_ABORT); restart = true; break; }
        spins = 0;
        TM_BEGIN(t_w2);
        while ((i32)(seq - lds_load_acq(&sm->sc_taken)) >= (i32)FQSX_SCR) {   // the chunk's ring slot still holds an unreleased one
          if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) { restart = true; break; }
          fq_sleep();
          if (lds_load_acq(&sm->cq_done) || lds_load_acq(&sm->sc_dead) || spin_expired(spins)) { quit = true; break; }
        }
        TM_END(w, TM_SCOUT_WAIT, t_w2);
        if (quit || restart) break;
        w.sb = &sm->sb[1 + me];
        lds_store_rel(&w.sb->h_pub, 0u);   // (the slot may hold a chunk of the same number from an earlier epoch)
        spec_commit(w, spl, sph);
        TM_COUNT(w, CN_SC_CHUNK);
#if FQ_WAVE > 1
        TM_BEGIN(t_se);
        if (i0 == base_pos) { scout_early(w, n); scout_settle_early(w, i0, n); }   // (the look-ups of positions whose b-mer is still partial, if any)
        TM_END(w, TM_SC_EARLY, t_se);
        if (!scout_fix(w, p, size, i0, n, false)) { TM_COUNT(w, CN_SC_ABORT); restart = true; break; }
        const u32 n_sw = uniform32(w.sb->h_fix_end) > n ? uniform32(w.sb->h_fix_end) : n;   // (lanes whose sweeps are probed ahead: scout_fix's spare lanes too)
        const u32 front0 = scout_rough_first(w, n_sw);
#else
        const u32 front0 = FQSX_SPEC;
#endif
        if (FQ_LANE == 0) { w.sb->h_read = idx; w.sb->h_i0 = i0; w.sb->h_n = n; w.sb->h_epoch = w.sc_epoch; w.sb->rr_front = front0; }
        FQ_SYNC();
        lds_store_rel(&w.sb->h_pub, seq + 1);   // the resolving wave may start on the chunk ...
#if FQ_WAVE > 1
        TM_BEGIN(t_sr);
        WHATIF(cfg, 8, 1);
        scout_rough(w, n_sw);                    // ... while its sweeps are still being probed (rr_front)
        TM_END(w, TM_SC_ROUGH, t_sr);
#endif
      }
      if (restart || quit) continue;
      add_s_letters(w, hist);
    }
    FQ_SYNC();
    lds_store_rel(&sm->sc_hd_taken[me], idx + 1);
    ++idx;
  }
  if (lds_load_acq(&sm->sc_dead)) lds_store_rel(&sm->sc_hd_taken[me], 0x7fffffffu);   // (the read-head wave does not wait for a scout that has left)
#ifdef FQSX_TIMING
  if (FQ_LANE == 0)
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) if (w.tm[i]) atomic_add64(&ws->stat[16 + i], w.tm[i]);
#endif
  if (me == 0) TM_STAMP(cfg, tid, launch, 4);
}

// The scout waves of the kernels without a read-head wave (original order, paired end): every compress_suffix call of
// the resolving wave -- a read after its prefix, the second mate from its anchor, the part left of the anchor on the
// reverse complement -- is one request: base state, sequence, direction.  The waves make its chunks in turn (chunk c:
// wave and ring slot c % nsc) and then wait for the next request; a k-mer correction inside the call is a request
// of the same kind from the corrected state.
FQ_DEV void scout_request_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 me, u32 nsc, u32 launch = 0) {
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = 0;
  w.ws = cfg.ws + tid;
  w.err = 0;
  w.piped = false;
  w.rcq = false;
  w.lqh = true;    // (reads the inserter wave's progress)
  w.sc_poll = true;
  w.rec = nullptr;
  w.scout = false;
  w.nsc = nsc;
  w.rdp = sm->rd[0];
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  w.sc_epoch = 0;
  for (;;) {
    // the next request (or the end of the launch)
    u32 spins = 0;
    bool quit = false;
    while (lds_load_acq(&sm->sc_req_seq) == w.sc_epoch) {
      if (lds_load_acq(&sm->cq_done)) { quit = true; break; }
      TM_BEGIN(t_id);
      fq_sleep();
      TM_END(w, TM_SC_IDLE, t_id);
      if (spin_expired(spins)) { quit = true; break; }   // never spin forever on the GPU
    }
    if (quit) break;
    w.sc_epoch = lds_load_acq(&sm->sc_req_seq);
    const WgShared::ScoutReq &q = sm->sc_req;
    const u32 call = uniform32(q.read), base_pos = uniform32(q.i0), size = uniform32(q.size), sbase = uniform32(q.slot_base);
    const bool reversed = (uniform32(q.flags) & SCQ_REVERSED) != 0;
    const u8 *p = (const u8 *)uniform64(q.p);
    Kmer *k[6] = {&w.pm, &w.sm_, &w.bm, &w.pm_u, &w.sm_u, &w.bm_u};
    for (u32 x = 0; x < 6; ++x) { k[x]->dir = uniform64(q.kdir[x]); k[x]->rc = uniform64(q.krc[x]); k[x]->cur = uniform32(q.kcur[x]); }
    w.cor_pos = uniform32(q.cor_pos);
    w.N_run = uniform32(q.n_run);
    for (u32 x = 0; x < 4; ++x) w.s_let[x] = uniform64(q.s_let[x]);
    w.sc_read = call;
    if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch) continue;   // (overwritten while it was read: take the newer one)
    u32 seq = 0;
    for (u32 i0 = base_pos, n = 0; i0 < size && (n = chunk_len(size - i0)) != 0; i0 += n, ++seq) {
      if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch || lds_load_acq(&sm->cq_done)) break;
      if ((sbase + seq) % nsc != me) continue;   // another wave's chunk
      SpecLane spl;   // (stage P into registers, then the wait for the ring slot: see scout_segment_body)
      SpecHead sph;
      TM_BEGIN(t_sp);
      const bool whole = speculate_t<true>(w, p, size, i0, n, reversed, i0 - base_pos, 0, &spl, &sph);
      TM_END(w, TM_SC_SPEC, t_sp);
      if (!whole) { TM_COUNT(w, CN_SC_ABORT); break; }
      spins = 0;
      bool stop = false;
      TM_BEGIN(t_w2);
      while ((i32)(seq - lds_load_acq(&sm->sc_taken)) >= (i32)nsc) {   // the wave's ring slot still holds an unreleased chunk
        if (lds_load_acq(&sm->sc_req_seq) != w.sc_epoch || lds_load_acq(&sm->cq_done)) { stop = true; break; }
        fq_sleep();
        if (spin_expired(spins)) { stop = true; break; }   // never spin forever on the GPU
      }
      TM_END(w, TM_SCOUT_WAIT, t_w2);
      if (stop) break;
      w.sb = &sm->sb[1 + me];
      lds_store_rel(&w.sb->h_pub, 0u);   // (the slot may hold a chunk of the same number from an earlier epoch)
      spec_commit(w, spl, sph);
      TM_COUNT(w, CN_SC_CHUNK);
#if FQ_WAVE > 1
      if (i0 == base_pos) { scout_early(w, n); scout_settle_early(w, i0, n); }   // (the look-ups of positions whose b-mer is still partial, if any)
      if (!scout_fix(w, p, size, i0, n, reversed)) { TM_COUNT(w, CN_SC_ABORT); break; }
      const u32 n_sw = uniform32(w.sb->h_fix_end) > n ? uniform32(w.sb->h_fix_end) : n;
      const u32 front0 = scout_rough_first(w, n_sw);
#else
      const u32 front0 = FQSX_SPEC;
#endif
      if (FQ_LANE == 0) { w.sb->h_read = call; w.sb->h_i0 = i0; w.sb->h_n = n; w.sb->h_epoch = w.sc_epoch; w.sb->rr_front = front0; }
      FQ_SYNC();
      lds_store_rel(&w.sb->h_pub, seq + 1);   // the resolving wave may start on the chunk ...
#if FQ_WAVE > 1
      TM_BEGIN(t_sr);
      scout_rough(w, n_sw);                    // ... while its sweeps are still being probed (rr_front)
      TM_END(w, TM_SC_ROUGH, t_sr);
#endif
    }
  }
#ifdef FQSX_TIMING
  if (FQ_LANE == 0)
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) if (w.tm[i]) atomic_add64(&w.ws->stat[16 + i], w.tm[i]);
#endif
  if (me == 0) TM_STAMP(cfg, tid, launch, 4);
}

// piped: this wave is the resolving half of a multi-wave worker (see coder_segment_body).
// MODE (dna_mode), DECODE and PIPED are compile-time constants: every kernel holds only the code of its own mode.
template <int MODE, bool DECODE, bool PIPED>
FQ_DEV void encode_segment_body(const DevCfg &cfg, WgShared *sm, u32 tid, u32 n_reads, u32 S, u32 seg, u32 launch = 0) {
  constexpr bool decode = DECODE, piped = PIPED;
  Wk w;
  w.cfg = &cfg;
  w.sm = sm;
  w.tid = tid;
  w.mode = (u32)MODE;
  WState *ws = cfg.ws + tid;
  w.ws = ws;
  w.err = 0;
  w.piped = piped;
  w.rcq = false;
  w.lqh = piped;
  w.rec = nullptr;
  w.rdp = sm->rd[0];
  constexpr bool heads = PIPED && MODE == 1;   // single-end sorted: the read heads come from the read-head wave
  w.sb = &sm->sb[0];
  w.scout = PIPED && !(cfg.dbg & FQSX_DBG_SCOUTS_OFF);   // stage P comes from the scout waves
  w.sc_reqmode = PIPED && MODE != 1;      // ... which, without a read-head wave, serve one request per compress_suffix call
  w.nsc = MODE >= 2 ? 2u : FQSX_NSC;
  w.rq_p = nullptr; w.rq_size = 0; w.rq_rev = false;
  w.sc_poll = false;
  w.sc_abandoned = false;
  w.sc_read = 0;
  w.sc_epoch = 0;
  w.sc_taken = 0;
  w.sc_base = 0;
  w.rq_early = false;
  w.rq_noscout = false;
  w.cq_head = w.cq_tail = 0;
  w.c_r_sym = 0;
  for (u32 i = 0; i < ST_N; ++i) w.st[i] = 0;
  for (u32 i = 0; i < FQSX_TM_SLOTS; ++i) w.tm[i] = 0;
  TM_BEGIN(t_total);
  TM_STAMP(cfg, tid, launch, 0);
  const u64 T = cfg.T;
  // PartitionForWorkers, reads_block.h:197-214
  u64 first = (u64)tid * n_reads / T, last = ((u64)tid + 1) * n_reads / T;
  if (tid) first &= ~1ull;
  if (tid + 1 < T) last &= ~1ull;
  if (seg == 0) {  // application.cpp:624-628
    ws->cursor = (u32)first;
    if (!piped) {   // (the coder wave initialises its own coder)
      ws->rc_low = 0;
      ws->rc_range = 0xff00000000000000ULL;
      ws->out_len = 0;
    }
    ws->dec_pos = ~0ull;   // decoder: stream not started yet
  }
  constexpr bool paired = MODE >= 2;
  u64 stop;
  if (seg < S) {
    const u64 ns = ((u64)seg + 1) * (last - first) / ((u64)S + 1) + first;
    if (!paired) stop = ns + 1;            // SE: synchronise after read i == next_synchro (application.cpp:643)
    else {                                 // PE: after the first pair with i >= next_synchro (application.cpp:1170)
      u64 i = ws->cursor;
      if (i < ns) i += (ns - i + 1) & ~1ull;
      stop = i + 2;
    }
  } else
    stop = last;
  if (stop > last) stop = last;
  // load state
  FQ_SYNC();
  mt_copy((u64 *)&sm->mt[0][0], (const u64 *)&ws->mt[0][0]);
  for (u32 g = FQ_LANE; g < 4; g += FQ_WAVE) sm->mt_idx[g] = ws->mt_idx[g];
  w.mn[0] = w.mn[1] = w.mn[2] = 0;
  {  // avg_filling_factor (bit_vec.h:204-210) only changes in insert phases (dna.cpp:2416-2418)
    u64 nu = cfg.siv_stats[0], nf = cfg.siv_stats[1];
    w.repm_gate = !((nf ? (double)nu / (double)nf : 0.0) < 7.0);
  }
  w.la[0] = w.la[1] = w.la[2] = 0;
  w.pq_lo[0] = w.pq_lo[1] = 0;
  w.lq_pub[0] = w.lq_pub[1] = 0;
  FQ_SYNC();
  enc_open(w, ws->rc_low, ws->rc_range, ws->out_len, cfg);
  w.avg_code = ws->avg_code; w.avg_letters = ws->avg_letters;
  for (u32 i = 0; i < 4; ++i) w.s_let[i] = ws->s_letters[i];
  w.hidden = ws->hidden_updates;
  w.ctx_letters = 0; w.cor_pos = 0; w.N_run = 0;
  km_reset(w.pm); km_reset(w.sm_); km_reset(w.bm); km_reset(w.pm_u); km_reset(w.sm_u); km_reset(w.bm_u);

  u64 cur = ws->cursor;
  w.pe_n = 0;
  TM_END(w, TX_PROLOG, t_total);
  if (decode) {
    w.din = cfg.din + cfg.din_off[tid];
    w.din_len = cfg.din_off[tid + 1] - cfg.din_off[tid];
    if (ws->dec_pos == ~0ull) rcd_start(w);
    else { w.din_pos = ws->dec_pos; w.din_buffer = ws->dec_buffer; }
    u8 *codes = cfg.dscratch + (u64)tid * 2 * cfg.dcap, *rcodes = codes + cfg.dcap;
    for (u64 i = cur; i < stop && !w.err; i += paired ? 2 : 1) {
      const u64 o0 = cfg.read_off[i], o1 = cfg.read_off[i + 1];
      const u8 *prev_out = nullptr;
      if (i > first) prev_out = cfg.dout + cfg.read_off[i - (paired ? 2 : 1)];
      if (!paired) read_dec(w, codes, cfg.dout + o0, (u32)(o1 - o0), prev_out, true);
      else if (i + 1 < stop) pair_dec(w, codes, rcodes, cfg.dout + o0, (u32)(o1 - o0), cfg.dout + o1, (u32)(cfg.read_off[i + 2] - o1), prev_out);
    }
    ws->dec_pos = w.din_pos;
    ws->dec_buffer = w.din_buffer;
  } else if (heads)
    for (u64 i = cur; i < stop && !w.err; ++i) {
      u64 o0 = cfg.read_off[i], o1 = cfg.read_off[i + 1];
      compress_read_rec(w, cfg.bases + o0, (u32)(o1 - o0), (u32)(i - cur), i + 1 < stop);
    }
  else if (!paired)
    for (u64 i = cur; i < stop && !w.err; ++i) {
      u64 o0 = cfg.read_off[i], o1 = cfg.read_off[i + 1];
      const u8 *prev = nullptr;
      u32 prev_size = 0;
      if (i > first) {
        u64 q0 = cfg.read_off[i - 1];
        prev = cfg.bases + q0;
        prev_size = (u32)(o0 - q0);
      }
      compress_read(w, cfg.bases + o0, (u32)(o1 - o0), prev, prev_size);
    }
  else
    for (u64 i = cur; i + 1 < stop && !w.err; i += 2) {
      u64 o0 = cfg.read_off[i], o1 = cfg.read_off[i + 1], o2 = cfg.read_off[i + 2];
      const u8 *prev = nullptr;
      u32 prev_size = 0;
      if (i > first) {  // read_prev = previous first mate of this worker in the block (dna.cpp:1748-1749)
        u64 q0 = cfg.read_off[i - 2], q1 = cfg.read_off[i - 1];
        prev = cfg.bases + q0;
        prev_size = (u32)(q1 - q0);
      }
      compress_pair(w, cfg.bases + o0, (u32)(o1 - o0), cfg.bases + o1, (u32)(o2 - o1), prev, prev_size);
    }
  if (stop > cur) cur = stop;
  TM_STAMP(cfg, tid, launch, 6);
  // The local tables are emptied next (ClearKmersToHT), but the inserts still pending for them are applied all the
  // same: an insert that finds a counter above its threshold draws from the worker's cinc_lb / cinc_ls stream
  // (dna.cpp:826,837), whose state lives on.
  lq_flush(w, MAIL_B);
  lq_flush(w, MAIL_S);

  // store state
  if (piped) {   // everything is queued: let the coder wave finish; the inserter wave has nothing left
    FQ_SYNC();
    lds_store_rel(&sm->cq_done, 1u);
    lds_store_rel(&sm->lq_quit, 1u);
  }
  ws->cursor = (u32)cur;
  if (!piped) {
    enc_close(w);
    ws->rc_low = w.enc.low; ws->rc_range = w.enc.range; ws->out_len = w.enc.len;
    ws->avg_code = w.avg_code; ws->avg_letters = w.avg_letters;
  }
  for (u32 i = 0; i < 4; ++i) ws->s_letters[i] = w.s_let[i];
  ws->hidden_updates = w.hidden;
  TM_END(w, TM_TOTAL, t_total);
  TM_STAMP(cfg, tid, launch, 2);
  // per-launch counters of this wave (timing build): what made the worker slow or fast
  TM_TRACE_VAL(cfg, tid, launch, 8, w.tm[CN_SLOW]); TM_TRACE_VAL(cfg, tid, launch, 9, w.tm[CN_ROUGH]);
  TM_TRACE_VAL(cfg, tid, launch, 10, w.tm[CN_DIRTY]); TM_TRACE_VAL(cfg, tid, launch, 11, w.tm[CN_FAST]);
  TM_TRACE_VAL(cfg, tid, launch, 12, w.tm[TM_RRWAIT]); TM_TRACE_VAL(cfg, tid, launch, 13, w.tm[TM_SPEC]);
  TM_TRACE_VAL(cfg, tid, launch, 14, w.tm[CN_GENERIC]); TM_TRACE_VAL(cfg, tid, launch, 15, w.tm[TM_LQ]);
#ifdef FQSX_TIMING
  {
    const u32 xs[16] = {TM_SLOW, TM_SPRE, TM_ROUGH, TM_FINDC, TM_POST, TM_KEYS, TM_READ_HEAD, TM_CQWAIT, TX_QMM, TX_QRUN, TX_PROLOG, TX_CHUNKQ, TX_FLUSH, TM_TOTAL, TX_G_EARLY, TX_OWNSPEC};
    for (u32 x = 0; x < 16; ++x) TM_TRACE_VAL(cfg, tid, launch, 16 + x, w.tm[xs[x]]);
  }
#endif
  if (!piped) {
    for (u32 i = 0; i < ST_N; ++i) ws->stat[i] += w.st[i];
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) ws->stat[16 + i] += w.tm[i];   // (stat[] has 64 words: [48..] only go to the trace)
  } else if (FQ_LANE == 0) {   // the coder wave adds to the same counters
    for (u32 i = 0; i < ST_N; ++i) if (w.st[i]) atomic_add64(&ws->stat[i], w.st[i]);
#ifdef FQSX_TIMING
    for (u32 i = 0; i < FQSX_TM_SLOTS && i < 48; ++i) if (w.tm[i]) atomic_add64(&ws->stat[16 + i], w.tm[i]);
#endif
  }
  FQ_SYNC();
  mt_copy((u64 *)&ws->mt[0][0], (const u64 *)&sm->mt[0][0]);
  for (u32 g = FQ_LANE; g < 4; g += FQ_WAVE) ws->mt_idx[g] = sm->mt_idx[g];
  for (u32 k = 0; k < 3; ++k) cfg.mail[k].n[tid] = w.mn[k];
  if (paired) cfg.pe_n[tid] = w.pe_n;
  if (w.err) *cfg.err = w.err;
}

// lane-parallel insert of one mailbox chunk (<= 64 keys, all owned by `tid`) into a global
// sub-table, RNG draws assigned in key order (InsertKmersToHT, dna.cpp:2426-2446; insert(),
// ht_kmer.h:420-438).  Two keys of a batch interact when they target the same slot (the same key
// twice, or two new keys racing for one empty slot): the batch is then applied in rounds, each round
// taking the longest prefix of the remaining keys that is free of such pairs, and the rest probes
// again after the round's stores -- exactly the sequential result, in 1-2 rounds instead of n steps.
template <class SM>
FQ_DEV void insert_batch(const DevCfg &cfg, SM *sm, const KTab &t, u32 tid, const u64 *keys, u32 n, u32 rng, const Cinc &ci,
                         u64 &nslots, u32 &err) {
  insert_batch_k(cfg, sm, t, tid, FQ_LANE < n ? keys[FQ_LANE] : 0, n, rng, ci, nslots, err);
}
// mykey: this lane's key of the batch (lanes >= n: ignored)
template <class SM>
FQ_DEV void insert_batch_k(const DevCfg &cfg, SM *sm, const KTab &t, u32 tid, u64 mykey, u32 n, u32 rng, const Cinc &ci,
                           u64 &nslots, u32 &err) {
  (void)cfg;
#ifdef FQSX_TIMING
  u64 ib_t[4] = {0, 0, 0, 0}, ib_rounds = 0;   // global-table batches only: probe walk / clash test / draws + stores / store wait
#define IB_MARK(i) do { const u64 now_ = fq_clock(); ib_t[i] += now_ - ib_c; ib_c = now_; } while (0)
  u64 ib_c = fq_clock();
#else
#define IB_MARK(i) ((void)0)
#endif
  u64 *s = t.slots + (u64)tid * t.stride;
  const u64 cm = (1ull << t.cbits) - 1ull;
  const u32 lane = FQ_LANE;
  const u64 v = lane < n ? mykey >> (64 - 2 * t.k) : 0;
  u32 filled = t.filled[tid];
  for (u32 done = 0; done < n;) {
    u64 pos = 0, item = 0;
    bool act = lane >= done && lane < n, found = false;
    if (act) {
      // the key's own slot, or the free slot it goes to (two-choice: the emptier of its two buckets, both fetched together; the
      // lane with the longest walk sets the pace of the whole batch)
      const TabLoc L = t.two ? tab_locate(t, s, v) : tab_locate1(t, s, v);
      pos = L.pos; item = L.item; nslots += L.ns;
      found = item != 0;
    }
    IB_MARK(0);
    // a new key can also extend a cluster another lane scanned, but that lane then targets a
    // different slot and stays correct
    u32 lim = n;
#if FQ_WAVE > 1
    // Does an earlier key of this round target my slot?  All pairs are tested through a 256-bucket LDS filter: every
    // lane sets its bit in the bucket of its slot, reads the bucket back and compares slots only with the (0-2) earlier
    // lanes it shares the bucket with -- a handful of LDS round trips instead of a 64-step scan per lane.
    FQ_SYNC();
    sm->ib_pos[lane] = act ? pos : ~0ull;
    for (u32 i = lane; i < 256; i += FQ_WAVE) sm->ib_hash[i] = 0;
    FQ_SYNC();
    const u32 hb = (((u32)pos ^ (u32)(pos >> 32)) * 0x9E3779B1u) >> 24;
    if (act) lds_or64(&sm->ib_hash[hb], 1ull << lane);
    FQ_SYNC();
    bool clash = false;
    if (act) {
      u64 cand = sm->ib_hash[hb] & ((1ull << lane) - 1ull);
      while (cand && !clash) {
        const u32 j = ctz64(cand);
        cand &= cand - 1ull;
        clash = sm->ib_pos[j] == pos;
      }
    }
    const u64 cl = wave_ballot(clash);
    if (cl) lim = ctz64(cl);
    act = act && lane < lim;
#else
    lim = done + 1;
#endif
    IB_MARK(1);
    const u32 n_new = popc64(wave_ballot(act && !found));
    if ((u64)(filled + n_new) * 10 >= t.nb * FQSX_BKT * 9) { err = FQSX_ERR_GTAB_FULL; return; }
    const u32 cnt = (u32)(item & cm);
    const bool draw = act && found && cnt > ci.thr && cnt < ci.maxv;
    const u64 dm = wave_ballot(draw);
    u32 r = 0;
    if (dm) {
      const u32 my = popc64(dm & ((1ull << lane) - 1ull)), total = popc64(dm);
      u32 idx = sm->mt_idx[rng];
      const u32 avail = idx >= 624 ? 0 : 624 - idx;
      if (draw && my < avail) r = mt_temper(sm->mt[rng][idx + my]);
      if (total > avail) {
        mt_twist(sm->mt[rng]);
        if (draw && my >= avail) r = mt_temper(sm->mt[rng][my - avail]);
        idx = total - avail;
      } else
        idx += total;
      FQ_SYNC();
      if (lane == 0) sm->mt_idx[rng] = idx;
      FQ_SYNC();
    }
    if (act) {
      if (!found) s[pos] = (v << t.cbits) | 1ull;
      else if (cnt <= ci.thr) { if (cnt < (u32)cm) s[pos] = item + 1; }
      else if (draw && (r % (ci.mult * (cnt - ci.thr)) == 0)) s[pos] = item + 1;
    }
    filled += n_new;
    done = lim;
    if (lane == 0 && done >= n) t.filled[tid] = filled;
    IB_MARK(2);
    FQ_SYNC_MEM();  // later rounds and batches of this wave read these slots from other lanes
    IB_MARK(3);
#ifdef FQSX_TIMING
    ++ib_rounds;
#endif
  }
#ifdef FQSX_TIMING
  if (rng < 2 && lane == 0) {
    for (u32 i = 0; i < 4; ++i) atomic_add64(&cfg.ws[tid].stat[16 + 40 + i], ib_t[i]);
    atomic_add64(&cfg.ws[tid].stat[16 + 44], ib_rounds);
    atomic_add64(&cfg.ws[tid].stat[16 + 45], 1);
  }
#endif
#undef IB_MARK
}


// ---- stable partition of the mailbox lists by owner -----------------------------------------
FQ_DEV u32 mail_owner(const DevCfg &cfg, u32 kind, u64 x) { return kind == MAIL_P ? p_owner(&cfg, x) : sb_owner(&cfg, x); }
// group index of an entry in the partitioned mailbox (the owner itself unless the run is sharded over GPUs)
FQ_DEV u32 mail_group(const DevCfg &cfg, u32 kind, u64 x) { return cfg.vmap[mail_owner(cfg, kind, x)]; }
FQ_DEV bool shard_mine(const DevCfg &cfg, u32 w) { return cfg.shard_world <= 1 || w % cfg.shard_world == cfg.shard_rank; }

// tile `blk` = (source, tile): histogram of owners
FQ_DEV void part_count_body(const DevCfg &cfg, u32 kind, u32 blk, u32 *hist /*LDS[256]*/) {
  const Mail &m = cfg.mail[kind];
  const u32 T = cfg.T, s = blk / m.n_tiles, t = blk % m.n_tiles;
  const u32 n = m.n[s], lo = t * FQSX_TILE, hi = n < lo + FQSX_TILE ? n : lo + FQSX_TILE;
  for (u32 d = FQ_LANE; d < 256; d += FQ_WAVE) hist[d] = 0;
  FQ_SYNC();
  for (u32 e = lo + FQ_LANE; e < hi; e += FQ_WAVE) lds_inc32(&hist[mail_group(cfg, kind, m.list[(u64)s * m.cap + e])]);
  FQ_SYNC();
  for (u32 d = FQ_LANE; d < T; d += FQ_WAVE) m.tile_hist[(u64)blk * T + d] = hist[d];
}
// owner `d`: exclusive scan of its column over all (source, tile) in order
FQ_DEV void part_scan_body(const DevCfg &cfg, u32 kind, u32 d) {
  const Mail &m = cfg.mail[kind];
  const u32 T = cfg.T, tiles = T * m.n_tiles;
  u32 run = 0;
  for (u32 base = 0; base < tiles; base += FQ_WAVE) {
    u32 i = base + FQ_LANE;
    u32 v = i < tiles ? m.tile_hist[(u64)i * T + d] : 0;
    u32 ex = wave_excl_scan32(v) + run;
    if (i < tiles) m.tile_hist[(u64)i * T + d] = ex;
    run += wave_sum32(v);
  }
  if (FQ_LANE == 0) m.dst_tot[d] = run;
}
FQ_DEV void part_dstoff_body(const DevCfg &cfg, u32 kind) {
  const Mail &m = cfg.mail[kind];
  u32 run = 0;   // exclusive scan over the owners, a wave's worth at a time
  for (u32 base = 0; base < cfg.T; base += FQ_WAVE) {
    const u32 d = base + FQ_LANE;
    const u32 v = d < cfg.T ? m.dst_tot[d] : 0;
    const u32 ex = wave_excl_scan32(v) + run;
    if (d < cfg.T) m.dst_off[d] = ex;
    run += wave_sum32(v);
  }
  if (FQ_LANE == 0) m.dst_off[cfg.T] = run;
}
// tile `blk`: stable scatter into the owners' groups
FQ_DEV void part_scatter_body(const DevCfg &cfg, u32 kind, u32 blk, u32 *cursor /*LDS[256]*/, u32 *ld /*LDS[64]*/, u64 *gm /*LDS[256]*/,
                              const u32 *doff = nullptr /*LDS[T+1]: the group offsets, if the caller has scanned them itself*/) {
  const Mail &m = cfg.mail[kind];
  const u32 T = cfg.T, s = blk / m.n_tiles, t = blk % m.n_tiles;
  const u32 n = m.n[s], lo = t * FQSX_TILE, hi = n < lo + FQSX_TILE ? n : lo + FQSX_TILE;
  if (lo >= hi) return;
  for (u32 d = FQ_LANE; d < 256; d += FQ_WAVE) gm[d] = 0;
  for (u32 d = FQ_LANE; d < T; d += FQ_WAVE) cursor[d] = (doff ? doff[d] : m.dst_off[d]) + m.tile_hist[(u64)blk * T + d];
  FQ_SYNC();
  for (u32 base = lo; base < hi; base += FQ_WAVE) {
    const u32 e = base + FQ_LANE, cnt = hi - base < FQ_WAVE ? hi - base : FQ_WAVE;
    u64 x = 0;
    u32 d = 0xffffffffu;
    if (e < hi) {
      x = m.list[(u64)s * m.cap + e];
      d = mail_group(cfg, kind, x);
    }
#if FQ_WAVE > 1
    // rank among the round's entries of the same group: the group's lanes as a bit mask in LDS (the groups are < 256)
    (void)cnt; (void)ld;
    if (e < hi) lds_or64(&gm[d], 1ull << FQ_LANE);
    FQ_SYNC();
    const u64 mine = e < hi ? gm[d] : 0;
    const u32 rank = popc64(mine & ((1ull << FQ_LANE) - 1ull));
    const u32 later = (mine >> FQ_LANE) >> 1 ? 1u : 0u;
    u32 cur = e < hi ? cursor[d] : 0;
    FQ_SYNC();
    if (e < hi) {
      m.sorted[cur + rank] = x;
      if (later == 0) { cursor[d] = cur + rank + 1; gm[d] = 0; }  // last entry of this owner in the round: advances the cursor, clears the mask
    }
    FQ_SYNC();
#else
    ld[FQ_LANE] = d;
    FQ_SYNC();
    u32 rank = 0, later = 0;
    for (u32 q = 0; q < cnt; ++q) {
      u32 dq = ld[q];
      rank += (q < FQ_LANE && dq == d) ? 1u : 0u;
      later += (q > FQ_LANE && dq == d) ? 1u : 0u;
    }
    u32 cur = e < hi ? cursor[d] : 0;
    FQ_SYNC();
    if (e < hi) {
      m.sorted[cur + rank] = x;
      if (later == 0) cursor[d] = cur + rank + 1;  // last entry of this owner in the round advances the cursor
    }
    FQ_SYNC();
#endif
  }
}

// the count index of the p-mer vector follows a field that went from value `from` to value `to` (DevCfg.siv_idx)
FQ_DEV void siv_idx_move(const DevCfg &cfg, u64 idx, u32 from, u32 to) {
  u32 *e = cfg.siv_idx + 4 * (idx >> FQSX_SIV_BLK_LOG);
#ifndef FQSX_EMU
  if (from) atomicSub(&e[from], 1u);
  atomicAdd(&e[to], 1u);
#else
  if (from) e[from] -= 1;
  e[to] += 1;
#endif
}
// owner `tid` applies its group of mailbox `kind` (InsertKmersToHT, dna.cpp:2393-2472).  The three
// mailboxes touch disjoint state (p-mer vector / ht_smer + cinc_s / ht_bmer + cinc_b), so they run as
// three independent workgroups per owner.
#ifndef FQSX_PF_AHEAD
#define FQSX_PF_AHEAD 4
#endif
// The prefetching wave of an insert-phase workgroup: walks the owner's group a few batches ahead of the inserting wave and touches
// every key's two home buckets, so that the inserting wave's ordered probe walks find them in L2.  It only reads.
#if FQ_WAVE > 1
template <class SM>
FQ_DEV void insert_prefetch_body(const DevCfg &cfg, SM *sm, u32 tid, u32 kind) {
  if (kind == MAIL_P) return;
  const Mail &m = cfg.mail[kind];
  const u32 lo = m.dst_off[tid], hi = m.dst_off[tid + 1], n = hi - lo;
  const KTab &t = kind == MAIL_S ? cfg.g_s : cfg.g_b;
  const u64 *s = t.slots + (u64)tid * t.stride, *keys = m.sorted + lo;
  u64 acc = 0;
  for (u32 o = 0; o < n; o += FQ_WAVE) {
    u32 spins = 0;
    while ((i32)(o / FQ_WAVE - lds_load_acq(&sm->pf_done)) > (i32)FQSX_PF_AHEAD) {   // a few batches ahead (their lines stay in L2 that long)
      fq_sleep();
      if (++spins > (1u << 18)) { keep_live(acc); return; }          // (the inserting wave gave up, or is far slower than ever seen: just stop)
    }
    if (o + FQ_LANE < n) {
      const TabHome th = tab_home(t, keys[o + FQ_LANE] >> (64 - 2 * t.k));
      acc ^= touch_load(&s[(u64)th.a * FQSX_BKT]) ^ touch_load(&s[(u64)th.b * FQSX_BKT]);
    }
  }
  keep_live(acc);
}
#endif
template <class SM>
FQ_DEV void insert_phase_body(const DevCfg &cfg, SM *sm, u32 tid, u32 kind) {
  WState *ws = cfg.ws + tid;
  const Mail &m = cfg.mail[kind];
  const u32 lo = m.dst_off[tid], hi = m.dst_off[tid + 1];
  if (kind == MAIL_P) {
    // p-mers: saturating 2-bit increments; order-free, so lanes update concurrently with CAS
    u64 nf = 0, nu = 0;
    for (u32 e = lo + FQ_LANE; e < hi; e += FQ_WAVE) {
      u64 idx = m.sorted[e];
      u64 *wp = cfg.siv + (idx >> 5);
      u32 sh = 2 * (u32)(idx & 31);
      u64 old = *wp;
      for (;;) {  // increment(), bit_vec.h:53-67
        u64 f = (old >> sh) & 3;
        if (f == 3) break;
        u64 seen = atomic_cas64(wp, old, old + (1ull << sh));
        if (seen == old) {
          nf += f == 0;
          siv_idx_move(cfg, idx, (u32)f, (u32)f + 1);
          if (cfg.siv_part) {   // the other ranks' replicas of the count index follow from this log (the vector itself they read here)
#ifndef FQSX_EMU
            const u32 at = atomicAdd(cfg.p_log_n, 1u);
#else
            const u32 at = (*cfg.p_log_n)++;
#endif
            cfg.p_log[at] = (idx << 4) | (f << 2) | (f + 1);
          }
          break;
        }
        old = seen;
      }
      ++nu;
    }
    nf = wave_sum64(nf);
    nu = wave_sum64(nu);
    if (FQ_LANE == 0) {  // update_no_filled / update_no_updates, dna.cpp:2416-2418 (atomics, bit_vec.h:25-26)
#ifndef FQSX_EMU
      atomicAdd((unsigned long long *)&cfg.siv_stats[1], (unsigned long long)nf);
      atomicAdd((unsigned long long *)&cfg.siv_stats[0], (unsigned long long)(nu + ws->hidden_updates));
#else
      cfg.siv_stats[1] += nf;
      cfg.siv_stats[0] += nu + ws->hidden_updates;
#endif
      ws->hidden_updates = 0;
      ws->stat[ST_SIV_WORDS] += hi - lo;
    }
    return;
  }
  const KTab &t = kind == MAIL_S ? cfg.g_s : cfg.g_b;
  const u32 rng = kind == MAIL_S ? RNG_S : RNG_B;
  const Cinc ci = kind == MAIL_S ? CINC_S : CINC_B;
  u32 err = 0;
  u64 n_slots = 0;
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < 624; i += FQ_WAVE) sm->mt[rng][i] = ws->mt[rng][i];
  if (FQ_LANE == 0) sm->mt_idx[rng] = ws->mt_idx[rng];
  FQ_SYNC();
  if (sm->pf_on) insert_keys<SM, true>(cfg, sm, t, tid, m.sorted + lo, hi - lo, rng, ci, n_slots, err);
  else insert_keys(cfg, sm, t, tid, m.sorted + lo, hi - lo, rng, ci, n_slots, err);
  FQ_SYNC();
  for (u32 i = FQ_LANE; i < 624; i += FQ_WAVE) ws->mt[rng][i] = sm->mt[rng][i];
  n_slots = wave_sum64(n_slots);
  if (FQ_LANE == 0) {
    ws->mt_idx[rng] = sm->mt_idx[rng];
#ifndef FQSX_EMU
    atomicAdd((unsigned long long *)&ws->stat[ST_GINS], (unsigned long long)(hi - lo));
    atomicAdd((unsigned long long *)&ws->stat[ST_GINS_SLOT], (unsigned long long)n_slots);
#else
    ws->stat[ST_GINS] += hi - lo;
    ws->stat[ST_GINS_SLOT] += n_slots;
#endif
  }
  if (err) *cfg.err = err;
}

// End() of the block's range coder: 8 flush bytes (sub_rc.h:79-86, application.cpp:664-665); the stream
// is then copied behind the other workers' streams so the host fetches the block in one transfer
FQ_DEV void finish_block_body(const DevCfg &cfg, u32 tid) {
  WState *ws = cfg.ws + tid;
  u64 low = ws->rc_low, len = ws->out_len;
  u8 *out = cfg.out + (u64)tid * cfg.out_cap;
  for (int i = 0; i < 8; ++i) {
    if (len < cfg.out_cap) out[len] = (u8)(low >> 56); else *cfg.err = FQSX_ERR_OUT_OVERFLOW;
    ++len;
    low <<= 8;
  }
  ws->rc_low = low;
  ws->out_len = len;
}
