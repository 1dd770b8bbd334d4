// fqsx_layout.h -- HBM data layout of the FQSX DNA codec (shared by host and device code).
//
// All state is flat and pointer-free inside the slots so that it can be sharded by owner
// worker across GPUs.  Reference structures each array replaces (paths relative to
// /root/reference/fqs):
//   siv            TSmallIntVector<2>                      bit_vec.h:17-231
//   gs_slots/gb_*  CHT_kmer<uint32_t> ht_smer / ht_bmer     ht_kmer.h:29-554, application.cpp:86-89
//   ls_/lb_slots   CHT_kmer<uint64_t> ht_*_local            dna.cpp:97-103
//   ctx            CContextHM<...,5> m_ctx_rc_codes/letters context_hm.h:21-248
//   small models   the six low-cardinality CContextHM maps  dna.h:49-56
//   mail           *_to_add[src][dst] mailboxes             application.h:56-59
#pragma once
#include "fqsx_plat.h"

#define FQSX_MAX_T 255
#define FQSX_SIV_BLK_LOG 14u     // fields per block of the p-mer vector's count index (16384 fields = 512 words = 4 KiB)
#define FQSX_SIV_BLK (1u << FQSX_SIV_BLK_LOG)
#define FQSX_NIL 0xffffffffu
#define FQSX_RD_LDS 3072u        // reads up to this length are staged in LDS (longer ones are read from HBM, one base at a time)
#define FQSX_HD 3u               // read-head records (and staging buffers) of a worker: the read-head wave runs up to FQSX_HD - 1 reads ahead
#define FQSX_SPEC 64u            // positions speculated per chunk (one per lane)
#define FQSX_PQ 512u             // entries of the LDS mirror of each local-insert list (power of two)
#define FQSX_CQ 256u             // entries of the coding queue (power of two, >= 2 * FQSX_SPEC)

// geometry of one rolling k-mer (kmer.h:279-298)
struct KGeom {
  u32 k, shift;
  u64 mask, kernel_mask;
};

// One open-addressed table of normalised k-mers: slot = (kmer_right_aligned << cbits) | count, 0 = empty; `stride` separates the
// per-owner sub-tables.  Round 4: BUCKETS of FQSX_BKT = 4 slots (32 bytes, one aligned access) and, for the global tables, TWO
// candidate buckets a and b per k-mer -- hashes of its kernel (symbols 2..k-3), so that the 4 (direct) or 4 (rc) sibling k-mers
// of a look-up share them (cf. ht_kmer.h:115-130,143-156).  A new key goes into the emptier of the two; only when both are full
// does it go down an overflow chain b + 1, b + 2, ...  Buckets fill from slot 0 up and never lose a key, so a look-up reads a and
// b -- both loads issued together -- and is done unless both are full (below 1 % of the look-ups at 80 % load).
// Why: rounds 1-3 probed linearly from one home slot and kept the tables at most half full -- 21.5 bytes of table per stored
// k-mer where the reference spends 5-6 (ht_kmer.h:34,69,88-112).  A wave probes 64 different k-mers at once and waits for the
// slowest lane, so what a denser table may not do is add DEPENDENT round trips for some of the lanes: measured on the way, a
// single sequence a, b, b + 1, ... through 8-slot buckets at 62-80 % load (one access for 7 look-ups in 8, a second one for the
// rest) ran 12 % slower on the benchmark file and 20 % slower on the 10 M-read file whose tables live in HBM.  Two choices cost
// a second, independent access per look-up and about twice the comparisons, but no second round trip: 3 % slower on both files
// (tools/ab_bench.py, gpurun_out r04_ab7 / r04_ab8_10M) at 10-13 bytes per k-mer, growth in steps of x1.3 (the number of buckets
// is any number >= 2: fastrange instead of a mask).  The workers' local tables (emptied every phase, never more than a quarter
// full) keep ONE sequence a, a + 1, ... and fetch one bucket (two = 0).
#define FQSX_BKT 4u
struct KTab {
  u64 *slots;
  u64 nb;         // buckets of every sub-table (capacity = nb * FQSX_BKT slots; < 2^32)
  u64 stride;     // slots between consecutive sub-tables (>= capacity)
  u32 *filled;    // [n_sub] occupied slots
  u32 k, cbits;
  u32 two, pad_;  // 1: two-choice (global tables), 0: one sequence (local tables)
};

// Context slot (32 B): adaptive 5-symbol model + visit counter, keyed by (tag,key).
// tag 0 = empty, 1 = rank-code contexts (m_ctx_rc_codes), 2 = letter contexts (m_ctx_rc_letters)
struct CtxSlot {
  u64 key;
  u32 counter;
  u16 tag;
  u16 total;
  u16 st[5];
  u16 pad[3];
};

// Mailbox of one kind (p-, s- or b-mers).  Sources append to a flat per-source list in push
// order; a GPU-wide stable partition (k_part_*) then groups the entries by owner, keeping
// (source, push) order inside each group -- the order InsertKmersToHT drains its column in.
#define FQSX_TILE 2048u
struct Mail {
  u64 *list;       // [T][cap]     entries in push order per source
  u32 *n;          // [T]          entries per source (written at the end of an encode launch)
  u64 *sorted;     // [T*cap]      entries grouped by owner
  u32 *tile_hist;  // [T*n_tiles][T] per-tile owner histogram, exclusive offsets after the scan
  u32 *dst_tot;    // [T]          entries per owner
  u32 *dst_off;    // [T+1]        start of each owner's group in `sorted`
  u32 cap;         // per source
  u32 n_tiles;     // ceil(cap / FQSX_TILE)
};
enum { MAIL_P = 0, MAIL_S = 1, MAIL_B = 2 };

// Paired-end minimizer-pair table (CHT_pair_kmers, ht_kmer.h:559-663): slot = (key, value | count << 2k),
// (0,0) = empty; sub-table d holds the keys with (murmur64(key) >> 48) mod T == d (ht_kmer.h:599-611).
struct PTab {
  u64 *key, *val;
  u64 cap_mask;   // capacity-1 of every sub-table
  u64 stride;     // slots between sub-tables
  u32 *filled;    // [n_sub]
};

// offsets (in u16 units) of the small direct-indexed models inside a worker's model block
// every model = N stats followed by its total
#define SM_FLAGS_N 2u       /* m_ctx_rc_flags: key = 8-bit duplicate-flag history */
#define SM_NS_N 2u          /* m_ctx_rc_prefix_Ns: key = 0 or position+1 */
#define SM_PSF_N 5u         /* m_ctx_rc_prefix_sorted_flags: key = 16-bit history */
#define SM_NIB_N 16u
#define SM_BYTE_N 256u
#define SM_OFF_FLAGS 0u
#define SM_OFF_NS (SM_OFF_FLAGS + 256u * (SM_FLAGS_N + 1u))
#define SM_OFF_PSF (SM_OFF_NS + 32u * (SM_NS_N + 1u))
#define SM_OFF_PSNB (SM_OFF_PSF + 65536u * (SM_PSF_N + 1u))
#define SM_OFF_NIB (SM_OFF_PSNB + 65536u * (5u + 1u))
#define SM_OFF_BYTE (SM_OFF_NIB + 68u * (SM_NIB_N + 1u))
#define SM_BYTE_ENTRIES (16u + 16384u)
#define SM_OFF_MPOS (SM_OFF_BYTE + SM_BYTE_ENTRIES * (SM_BYTE_N + 1u))   /* m_ctx_rc_minimizer_pos: 15 ids x 6 key classes */
#define SM_MPOS_ENTRIES 96u
#define SM_OFF_MID (SM_OFF_MPOS + SM_MPOS_ENTRIES * (SM_BYTE_N + 1u))  /* ctx_rc_pe_minimizer_id: one 16-symbol model */
#define SM_TOTAL_U16 (SM_OFF_MID + (SM_NIB_N + 1u))
#define SM_LAZY_ENTRIES (SM_BYTE_ENTRIES + SM_MPOS_ENTRIES)           /* lazily initialised 256-symbol models */

// per-worker persistent state (dna.cpp:148-171 and the encoder objects of application.cpp:580-584)
struct WState {
  double avg_code, avg_letters;        // dna.h:34,37
  u64 ctx_flags, ctx_ps_flags;         // dna.h:84,87
  u64 s_letters[4];                    // dna.h:119
  u64 pmer_prev_dir, pmer_prev_rc;     // pmer_can_prev (dna.h:166); cur is 0 or pmer_len
  u32 pmer_prev_cur;
  u32 cursor;                          // next read of the block this worker codes
  u64 hidden_updates;                  // no_pmer_hidden_updates, dna.h:43
  u64 rc_low, rc_range;                // CRangeEncoder, sub_rc.h:44-45
  u64 out_len;
  u64 dec_buffer, dec_pos;             // CRangeDecoder (sub_rc.h:154-157) when the codec decodes
  u32 mt_idx[4];                       // cinc_b, cinc_s, cinc_lb, cinc_ls (dna.h:113-116)
  alignas(8) u32 mt[4][624];
  u64 stat[64];                        // probe/byte accounting, see ST_*; [16..63] in-kernel section times (10 ns ticks; timing builds)
};
enum { RNG_B = 0, RNG_S = 1, RNG_LB = 2, RNG_LS = 3 };
enum {
  ST_GPROBE = 0,   // global table cluster scans
  ST_GSLOT,        // slots read by them
  ST_LPROBE,       // local table scans
  ST_LSLOT,
  ST_GINS,         // global inserts
  ST_GINS_SLOT,
  ST_SIV_WORDS,    // siv 64-bit words read or updated
  ST_CTX,          // context slots read
  ST_CODED,        // symbols range-coded
  ST_LINS,         // local inserts
  ST_MAIL,         // mailbox entries pushed
  ST_BASES,        // input bases consumed
  ST_SIV_SAVED,    // siv words the reference's rank loop sweeps (dna.cpp:600-605: the whole range between the previous and the current
                   // p-mer) minus the words the kernel read for the same rank (end blocks + count index): ST_SIV_WORDS + this = the
                   // algorithmic figure of SURVEY 8d
  ST_N
};
// section timers (only maintained by -DFQSX_TIMING builds)
enum { TM_TOTAL = 0, TM_SPEC, TM_FAST, TM_SLOW, TM_POST, TM_READ_HEAD, TM_LQ, TM_ROUGH, TM_REPM, TM_FINDC,
       CN_FAST, CN_SLOW, CN_CHUNK, CN_DIRTY, CN_ROUGH, CN_REPM,
       CN_EXT, CN_GENERIC, CN_LQFLUSH, TM_RRWAIT /* resolver waiting for a chunk's sweep frontier */, TM_CQWAIT /* resolver waiting for coding-queue space */, CN_EARLY, CN_P2, TM_P2,
       TM_CODER_IDLE /* coder wave: queue empty */, TM_SCOUT_WAIT /* scout wave: ring full or no read head yet */, TM_SPRE,
       TM_CR_S, TM_CR_MID, TM_CR_AVG, TM_CR_RC, TM_KEYS,
       TM_SC_SPEC /* scout waves: stage P */, TM_SC_EARLY, TM_SC_ROUGH /* ... their sweeps */, TM_SC_IDLE /* ... nothing left to do */, CN_SC_CHUNK, CN_SC_ABORT,
       TM_SP_ROLL /* stage P: k-mer roll */, TM_SP_PROBE /* ... global b-mer probe */,
       TM_N, TM_SP_HIT = 46 /* ... keys, rank, repair decision of a hit */, TM_SP_MISS = 47 /* ... miss cascade */,
       TX_QMM = 48 /* resolver: quiet_miss_mask */, TX_QRUN /* ... quiet stretches */, TX_PROLOG /* ... launch start to first read */,
       TX_CHUNKQ /* ... chunk into the coding queue (after the keys) */, TX_FLUSH /* ... end-of-chunk flush_pushes */,
       TX_G_EARLY /* generic positions: b-mer still partial */, TX_OWNSPEC /* resolver: its own stage P (correction windows, reads without the scouts) */ };
#ifdef FQSX_TIMING
#define FQSX_TM_SLOTS 56   /* [0..47] are summed into WState.stat[16..63]; [48..55] only go to the per-launch trace */
#else
#define FQSX_TM_SLOTS 1   /* (the timers only exist in the diagnostic build) */
#endif

struct DevCfg {
  u32 T, mode;                 // mode 0 = original order, 1 = sorted (params.h:18)
  u32 prefix, pmer, smer, bmer;
  KGeom gp, gs, gb;
  u32 pmer_mod_shift;          // dna.cpp:2381
  u32 ps_nobytes_n;            // alphabet of prefix_sorted_no_bytes, dna.cpp:130
  u64 T_magic;                 // ceil(2^32 / T): x % T for x < 2^14 without a division
  u32 T_pow2;                  // T is a power of two (then x % T is a mask)
  u32 dbg;                     // FQSX_DBG_*: forces the fall-back branches of the eight-wave protocol (tests only; env FQSX_PROTO_DEBUG)
  u64 *siv;                    // 4^pmer 2-bit counters
  u64 *siv_stats;              // [0] no_updates [1] no_filled (bit_vec.h:25-26)
  u32 *siv_idx;                // [4^pmer / FQSX_SIV_BLK][4]: per block of the vector, how many fields hold 1, 2, 3 ([0] unused: zeros are the rest).
                               // The rank of compress_prefix_sorted (dna.cpp:600-605) is a count over a range of the vector; whole blocks of
                               // the range come from here instead of being swept (kept up to date by the insert phase)
  KTab g_s, g_b;               // owner-sharded global tables (T sub-tables)
  KTab l_s, l_b;               // per-worker local tables (T sub-tables)
  CtxSlot *ctx;                // [T][ctx_cap]
  u64 ctx_cap_mask;
  u32 *ctx_filled;             // [T]
  u16 *small;                  // [T][SM_TOTAL_U16]
  u8 *byte_init;               // [T][SM_LAZY_ENTRIES] lazy-init flags of the 256-symbol models
  WState *ws;                  // [T]
  Mail mail[3];
  PTab g_pe, l_pe;             // paired-end: global (owner-sharded) and per-worker local pair tables
  u64 *pe_list;                // [T][pe_cap][3] (key, value, weight) triples pushed by each source
  u32 *pe_n;                   // [T]
  u32 pe_cap;                  // triples per source
  u64 *pe_bkt;                 // [T * pe_cap][3] the phase's triples grouped by owner (k_pe_bucket_*; any order inside a group: inserts commute)
  u32 *pe_bkt_n;               // [2T + 1] triples per owner, then the groups' offsets (exclusive scan, [T] = total)
  u32 *pe_bkt_cur;             // [T] scatter cursors
  const u8 *bases;             // block input: concatenated ASCII reads
  const u64 *read_off;         // n_reads+1
  u8 *out;                     // [T][out_cap] DNA streams of the block
  u64 out_cap;
  // decoding (k_decode_segment): input streams, output block, per-worker code lines
  const u8 *din;               // the T DNA streams of the block, concatenated
  const u64 *din_off;          // [T+1]
  u8 *dout;                    // decoded block (ASCII), reads at read_off[]
  u8 *dscratch;                // [T][2][dcap] 0..4 codes of the read being decoded (+ reverse-complement line)
  u64 dcap;
  u8 *pe_scr;                  // paired-end, mates longer than FQSX_RD_LDS: [T][3][pe_scr_cap] code line of either mate and the
  u64 pe_scr_cap;              // reverse-complement line of an anchored second mate (what LDS holds for shorter mates); else null
  u32 *err;                    // device error word (0 = ok)
  u64 *trace;                  // -DFQSX_TIMING builds: [launch][worker][8] clock stamps of the roles (else null)
  // Sharded mode: worker w -- its coder state, RNG streams, local tables and the sub-tables it owns --
  // lives on rank w % shard_world; every rank holds a read-only replica of all sub-tables.  shard_world == 1: one GPU.
  u32 shard_rank, shard_world;
  u32 tab_load_pct;            // a global s- / b-mer sub-table is grown before an insert phase would fill it beyond this (host: fqsx_dna.tab_load_pct)
  u32 whatif;                  // -DFQSX_WHATIF builds (tools/gpu_whatif.py): (role << 16) | delay units -- one role is slowed down, the file's rate says
                               // how much of that role's time is on the critical path; product builds never read it
  u32 sys_scope;               // the s- / b-mer tables are partitioned over more than one rank: sub-tables of other GPUs are read
                               // through peer mappings, so the kernels that write own sub-tables end with a system-scope release and
                               // the kernels that look k-mers up start behind a system-scope acquire (fqsx_plat.h)
  u32 pe_part;                 // sharded mode, partitioned tables: the pair table is partitioned too -- a rank applies the triples of its own owners only
  u32 siv_part, pad_siv_;      // sharded mode, partitioned tables: the p-mer vector is partitioned too (its 4096 owner ranges are chunks on their
                               // owners' ranks); the count index stays a replica on every rank, kept up from the owners' transition log:
  u64 *p_log;                  // this phase's (index << 4 | old << 2 | new) of every field an owner of this rank changed (zero-padded) ...
  u32 *p_log_n;                // ... and their number
  const u8 *vmap;              // [256] owner -> position of its group in the partitioned mailbox: identity, or rank-major
                               // (all owners of rank 0, then rank 1, ...) so that what goes to one rank is contiguous
  u32 *shard_cnt;              // [3][T][T] entries source s pushed for owner o in this phase (own sources; else 0)
};

// Protocol test switches (DevCfg.dbg).  Which wave settles what depends on wave timing in the product run, so the
// branches taken when a partner is late or absent are otherwise exercised only by chance; results must not change.
enum {
  FQSX_DBG_SCOUTS_OFF = 1,      // no scout waves: the resolving wave runs stage P, the sweeps and the partial look-ups itself
  FQSX_DBG_ABANDON = 2,         // the resolving wave abandons the scouts on every third read from its start, on every third after its first chunk
  FQSX_DBG_RESTART = 4,         // a correction window (the resolving wave's own stage P for bmer - 1 positions, then back into the
                                // scouts' chunk in mid-chunk) after every chunk, corrected or not
  FQSX_DBG_INSERTER_STALL = 8,  // the inserter wave gets list entries only when a look-up has to wait for them (and at the end of a segment)
};
enum {
  FQSX_ERR_OUT_OVERFLOW = 1,
  FQSX_ERR_GTAB_FULL = 2,
  FQSX_ERR_LTAB_FULL = 3,
  FQSX_ERR_CTX_FULL = 4,
  FQSX_ERR_MAIL_FULL = 5,
  FQSX_ERR_PE_FULL = 6,
  FQSX_ERR_PE_READ_TOO_LONG = 7,  // (a mate longer than the scratch lines the host sized for the block: cannot happen)
  FQSX_ERR_DECODE = 8,
  FQSX_ERR_PIPE = 9,              // the coding queue between the two waves of a worker stalled
};
