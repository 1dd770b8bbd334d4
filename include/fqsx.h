/* fqsx.h -- C ABI of the MI355X-native FQSqueezer DNA path (libfqsx.so).
 *
 * The reference (refresh-bio/fqsqueezer v1.1) has no plugin/FFI surface; the narrowest
 * seam its DNA hot path sits behind is the public interface of CDNACompressor as driven by
 * the worker lambda of CApplication::compress_se_files.  This ABI replaces that seam at
 * *reads-block* granularity:
 *
 *   fqsx_dna_create        <- CDNACompressor::SetParams/SetCoder/Init/SetKmerDS for all T
 *                             workers + CApplication::AdjustToParams
 *                             (fqs/compressor.h:42-43, fqs/dna.h:263-269,
 *                              fqs/application.cpp:77-108, :578-608)
 *   fqsx_dna_encode_block  <- one iteration of the worker loop for all T workers:
 *                             ResetReadPrev, Start, CompressDirect|CompressSorted per read,
 *                             the barrier-synchronised InsertKmersToHT/ClearKmersToHT
 *                             phases, End  (fqs/application.cpp:610-669, fqs/dna.h:271-285)
 *   fqsx_dna_destroy       <- ~CDNACompressor / ~CApplication
 *
 * The library is HIP-only: it fails with FQSX_E_NO_DEVICE when no gfx950 GPU is present.
 * There is no CPU code path behind this ABI.
 *
 * Threading: one fqsx_dna per output file, calls serialised by the caller, blocks in file
 * order (models persist across blocks exactly like the reference's per-thread compressors).
 */
#ifndef FQSX_H
#define FQSX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fqsx_dna fqsx_dna;

/* Transport of the sharded mode: the collectives among the `world` ranks (one process per GPU) that share one file, on
 * buffers in the codec's memory space (device memory).  fqsx_rccl_comm_create gives the RCCL one (xGMI inside a node);
 * a caller with a transport of its own (MPI, a test harness) fills in the three functions.  Each returns 0 on success
 * and may return before the operation has completed as long as later work on the codec's stream waits for it. */
typedef struct fqsx_comm {
  void *ctx;
  /* in-place sum over the ranks of n 32-bit words */
  int (*allreduce_sum_u32)(void *ctx, uint32_t *buf, uint64_t n);
  /* n_buf variable all-to-alls of 64-bit words in one go: buffer b sends send_counts[b * world + r] words to rank r (the
   * parts lie back to back in send[b], in rank order) and receives recv_counts[b * world + q] words from rank q likewise */
  int (*alltoallv_u64)(void *ctx, uint32_t n_buf, const uint64_t *const *send, const uint64_t *send_counts,
                       uint64_t *const *recv, const uint64_t *recv_counts);
  /* every rank contributes n 64-bit words; recv = [world][n] */
  int (*allgather_u64)(void *ctx, const uint64_t *send, uint64_t n, uint64_t *recv);
  /* optional (may be null): give the transport up after a failed collective, so that this rank's part of whatever is still
   * pending does not keep the other ranks waiting (RCCL: ncclCommAbort) */
  void (*abort)(void *ctx);
} fqsx_comm;

enum {
  FQSX_OK = 0,
  FQSX_E_ARG = -1,        /* bad argument / malformed header / unsupported dna_mode */
  FQSX_E_NO_DEVICE = -2,  /* no HIP device (the library has no CPU fallback) */
  FQSX_E_HIP = -3,        /* HIP runtime error, see fqsx_last_error() */
  FQSX_E_NOMEM = -4,      /* device allocation failed */
  FQSX_E_DEVICE = -5,     /* device-side error word set (table/stream overflow) */
  FQSX_E_PEER = -6        /* sharded mode: another rank of the world failed; every rank has left the phase (this one was fine) */
};

/* header17: the 17 parameter bytes of the .fqs file being written
 * ('K','C','S','D', no_threads T, dna_mode, quality_mode, id_mode, quality_thr,
 *  duplicates_check, prefix_len, pmer_len, smer_len, bmer_len, imer_len, hmer_len,
 *  ht_prefix_len; fqs/params.h:80-100).  T is the number of logical workers and is part of
 * the bitstream.  dna_mode 0..3 (se_original, se_sorted, pe_original, pe_sorted) are implemented; in the
 * paired modes a block holds the mates interleaved (mate 1, mate 2, mate 1, ...) and n_reads is even.
 * device: HIP device ordinal. */
int fqsx_dna_create(const uint8_t *header17, int device, fqsx_dna **out);
/* The same with the codec's kernels confined to compute units [part * CUs / n_parts, (part + 1) * CUs / n_parts) of the
 * device (a HIP stream with a CU mask).  One file with T <= 64 workers occupies T of the 256 CUs (a worker's workgroup
 * takes a whole CU's LDS), so a GPU has room for several files; giving every concurrent file a partition of its own keeps
 * their kernels from queueing behind each other for compute units.  n_parts = 1: the whole device (= fqsx_dna_create). */
int fqsx_dna_create_on_partition(const uint8_t *header17, int device, uint32_t part, uint32_t n_parts, fqsx_dna **out);
void fqsx_dna_destroy(fqsx_dna *);

/* Encode one reads block.  bases = concatenated ASCII sequences (ACGTN) of the block's
 * reads in block order, read_off = n_reads+1 byte offsets into bases (host memory).
 * generation = index of the block within the file (drives the synchronisation schedule,
 * fqs/application.h:85-92).  On return streams[w]/lens[w] (w < T) describe worker w's
 * complete DNA range-coder stream for this block in host memory owned by the codec and
 * valid until the next call on the same codec. */
int fqsx_dna_encode_block(fqsx_dna *, const uint8_t *bases, const uint64_t *read_off, uint32_t n_reads,
                          uint32_t generation, const uint8_t **streams, uint64_t *lens);

/* Same with the block already resident in device memory (HBM): d_bases / d_read_off are
 * device pointers; h_read_off is the host copy of the offsets (needed for sizing). */
int fqsx_dna_encode_block_dev(fqsx_dna *, const uint8_t *d_bases, const uint64_t *d_read_off,
                              const uint64_t *h_read_off, uint32_t n_reads, uint32_t generation,
                              const uint8_t **streams, uint64_t *lens);

/* Decode one reads block: inverse of fqsx_dna_encode_block (replaces one generation of the reference's
 * decoder workers, fqs/application.cpp:874-917 with CDNACompressor::DecompressSE / DecompressPE,
 * fqs/dna.h:277-278).  streams[w]/lens[w] = worker w's DNA stream of the block (host memory), read_off =
 * n_reads+1 offsets of the reads inside bases_out (the read lengths come from the meta stream).  A codec
 * instance is used either for encoding or for decoding a file, never both. */
int fqsx_dna_decode_block(fqsx_dna *, const uint8_t *const *streams, const uint64_t *lens, const uint64_t *read_off,
                          uint32_t n_reads, uint32_t generation, uint8_t *bases_out);

/* Accounting counters summed over workers since creation:
 * [0] global probes [1] global slots read [2] local probes [3] local slots read
 * [4] global inserts [5] slots read by them [6] siv words touched [7] context slots read
 * [8] symbols range-coded [9] local inserts [10] mailbox entries [11] input bases
 * [12] siv words of the reference's rank sweeps (fqs/dna.cpp:600-605) that the count index spared the kernels ([6] + [12] = the
 *      words the algorithm sweeps)
 * [16..63] in-kernel section timers / event counts (10 ns ticks, only in -DFQSX_TIMING diagnostic builds) */
int fqsx_dna_stats(fqsx_dna *, uint64_t out[64]);

/* Table occupancy and device memory (no counterpart in the reference, which prints table populations at -v 2,
 * fqs/application.cpp:733-741): [0] distinct s-mers stored [1] distinct b-mers stored [2] s-mer table slots (all owners)
 * [3] b-mer table slots [4] p-mer vector bytes [5] context-table slots (all workers) [6] contexts stored
 * [7] device bytes held now [8] ... at most so far (old + new table during a growth included) [9] table growth events
 * [10] minimizer pairs stored (paired-end) [11] pair-table slots (16 bytes each) [12] bytes per k-mer table slot
 * [13] bytes of s- + b-mer table memory this rank holds (all of it, or its owners' share with partitioned tables)
 * [14] bytes of pair-table memory this rank holds (likewise) [15] bytes of the p-mer vector this rank holds (likewise). */
int fqsx_dna_capacity(fqsx_dna *, uint64_t out[16]);

/* One GPU's capacity mode (no counterpart in the reference, whose sub-tables are separate heap vectors that grow one by one,
 * fqs/ht_kmer.h:88-112): before the first block, switch the s- and b-mer tables to one chunk of physical memory per
 * sub-table inside one reserved address range (HIP virtual-memory API).  The kernels see the same layout; a growth then
 * re-inserts sub-table by sub-table and returns each old chunk before the next new one is made, so the peak is the new table
 * plus one old sub-table instead of old + new side by side.  Also chosen by the environment variable FQSX_CHUNKED_TABLES=1; and a
 * table that reaches 2 GiB (FQSX_CHUNK_AUTO_KB: another size in KiB, 0 = never) turns into a chunked table at that growth by itself. */
int fqsx_dna_use_chunked_tables(fqsx_dna *);

/* Sharded mode (reference: the T x T mailboxes of fqs/application.h:56-59 and their owner-side
 * application, fqs/dna.cpp:825-847, :2393-2472): logical worker w -- coder state, RNG streams, local tables and the
 * sub-tables it owns -- lives on rank w % world; every rank keeps a replica of all sub-tables for the look-ups (or maps the
 * other ranks' sub-tables: fqsx_shard_partition_tables below).  The product path is the phase loop inside the library
 * (fqsx_shard_attach + fqsx_shard_encode_block, further down).
 * The step-wise form below is the round-2 bring-up driver, kept for single-end worlds and for callers that want to run the
 * collectives themselves: one synchronisation phase = encode -> [all-reduce of the per-(source, owner) counts] -> pack ->
 * [all-to-all of the three mailboxes] -> merge -> [all-reduce(max) of the two demand words fqsx_shard_merge returns] -> insert
 * -> [all-gather of the applied items] -> apply -> [all-reduce of the p-mer statistics] -> end_phase; the bracketed collectives
 * are the caller's (fqsqueezer_amd/sharded.py: ShardedDnaCodec over torch.distributed).  It does not exchange the pair-table
 * triples: fqsx_shard_begin_block refuses a paired-end codec in a world of more than one rank (FQSX_E_ARG).
 * The streams are bit-identical to the one-GPU run's.  Pointers marked [codec] are in the codec's memory space (device memory). */
int fqsx_shard_config(fqsx_dna *, uint32_t rank, uint32_t world);
int fqsx_shard_begin_block(fqsx_dna *, const uint8_t *bases /*[codec]*/, const uint64_t *read_off /*[codec]*/, const uint64_t *h_read_off,
                           uint32_t n_reads, uint32_t generation, uint32_t *n_segments);
int fqsx_shard_encode(fqsx_dna *, uint32_t seg, uint32_t *counts /*[codec] [3][T][T]*/);
int fqsx_shard_pack(fqsx_dna *, const uint32_t *counts_sum /*[codec]*/, uint64_t *const send[3] /*[codec]*/);
int fqsx_shard_merge(fqsx_dna *, const uint64_t *const recv[3] /*[codec]*/, uint64_t need[2]);
int fqsx_shard_insert(fqsx_dna *, uint64_t need_s, uint64_t need_b, uint64_t *const items[3] /*[codec]*/, uint64_t siv_delta[2]);
int fqsx_shard_apply(fqsx_dna *, uint32_t kind, const uint64_t *items /*[codec]*/, uint64_t n);
int fqsx_shard_end_phase(fqsx_dna *, const uint64_t siv_delta_sum[2]);
int fqsx_shard_finish_block(fqsx_dna *, const uint64_t *h_read_off, const uint8_t **streams, uint64_t *lens);

/* The same phase loop inside the library (the reference's phase is three barrier waits, fqs/application.cpp:643-655):
 * fqsx_shard_attach = fqsx_shard_config + the transport; fqsx_shard_encode_block runs a whole reads block -- per phase three
 * collectives (all-reduce of the count matrix; the three mailboxes in one grouped all-to-all; one all-gather carrying the
 * applied items, the p-mer statistics and, paired-end, the pair-table triples) and one host round trip (transfer sizes,
 * table demand, error word).  All four dna_modes.  bases / read_off: the block in the codec's memory space, the same on
 * every rank; streams[w] / lens[w] are meaningful for this rank's workers (w % world == rank).
 * fqsx_shard_traffic: [0] phases [1] collectives issued [2] all-to-all words sent to other ranks [3] all-gather words sent. */
int fqsx_shard_attach(fqsx_dna *, uint32_t rank, uint32_t world, const fqsx_comm *comm);
int fqsx_shard_encode_block(fqsx_dna *, const uint8_t *bases /*[codec]*/, const uint64_t *read_off /*[codec]*/, const uint64_t *h_read_off,
                            uint32_t n_reads, uint32_t generation, const uint8_t **streams, uint64_t *lens);
int fqsx_shard_traffic(fqsx_dna *, uint64_t out[4]);
/* Partitioned look-ups (SURVEY.md 8e option (i); the reference shares ONE table among its threads, fqs/application.h:51-54,
 * with the owner functions of fqs/dna.cpp:825 and :2381-2386 deciding who writes a key).  Collective over the ranks of one
 * node, after fqsx_shard_attach and before the first block: from then on a rank holds the physical memory of only the s- and
 * b-mer sub-tables its workers own (1/world of the k-mer tables) and maps the other ranks' sub-tables beside them
 * (hipMemCreate -> POSIX descriptor over a Unix socket -> hipMemImportFromShareableHandle -> hipMemMap), so that every rank
 * sees one table in one address range and a look-up of a foreign sub-table is a load over xGMI.  Writes stay with the owner
 * (insert phase); the phase's collectives order them before the next look-ups.  The all-gather of a phase then carries no
 * k-mer items, only the owners' occupancy counters, the p-mer items and statistics and the paired-end triples.  The p-mer
 * vector (fqs/application.h:51 siv_pmer, owner function fqs/dna.cpp:658) is partitioned as well when its 4096 owner ranges
 * reach the chunk granule (2 MiB: the 16 GiB vector of the default geometry; smaller vectors stay replicas): a range lives on
 * its owner's rank, the count index over the vector stays a replica that follows the owners' log of changed fields.  A paired-end codec's pair table (fqs/application.h:54 ht_pe_mers, owner function
 * fqs/ht_kmer.h:599-602) is partitioned the same way: every rank receives every source's triples but applies those of its
 * own owners only, and a fourth one-word all-reduce closes such a phase (the inserts follow the phase's last collective).
 * Streams are bit-identical to the one-GPU run's.
 * Memory order: the kernels that write own sub-tables (insert phase, growth) end with a system-scope release and the encode /
 * decode kernels start behind a system-scope acquire (csrc/fqsx_plat.h), so that an owner's writes are in its HBM before the
 * phase's last collective and no GPU serves a foreign look-up from a line it cached in an earlier launch.
 * Peer access: the ranks all-gather their GPUs' PCI bus ids and ask hipDeviceCanAccessPeer; if any pair of the world cannot
 * reach each other ALL ranks stay with table replicas -- the call still returns FQSX_OK, fqsx_last_error() says why and
 * fqsx_shard_is_partitioned() returns 0.
 * fqsx_dna_capacity()[13] = bytes of k-mer table memory this rank holds. */
int fqsx_shard_partition_tables(fqsx_dna *);
int fqsx_shard_is_partitioned(fqsx_dna *);
/* RCCL transport on the codec's own stream (collectives and kernels are ordered by the stream; librccl is loaded on first
 * use).  Rank 0 calls fqsx_rccl_unique_id and hands the 128 bytes to the other ranks by any means (a file, a TCP store). */
int fqsx_rccl_unique_id(uint8_t id[128]);
int fqsx_rccl_comm_create(fqsx_dna *, const uint8_t id[128], uint32_t rank, uint32_t world, fqsx_comm *out);
/* The communicator goes on with another codec (the next file of the process): its collectives run on that codec's stream. */
int fqsx_rccl_comm_rebind(fqsx_comm *, fqsx_dna *);
/* out[0] = ranks of the communicator as RCCL counts them (ncclCommCount), out[1] = this process's rank in it. */
int fqsx_rccl_comm_info(fqsx_comm *, uint32_t out[2]);
void fqsx_rccl_comm_destroy(fqsx_comm *);

/* Kernel timing: when enabled every launch is bracketed by HIP events on the codec's stream.
 * out[0..2] = accumulated milliseconds of the encode-segment, insert-phase and all other
 * kernels; out[3..5] = their launch counts. */
int fqsx_dna_set_profiling(fqsx_dna *, int enable);
int fqsx_dna_kernel_times(fqsx_dna *, double out[6]);
/* Diagnostic builds (-DFQSX_TIMING) only: per-launch, per-worker clock stamps of the five roles of the encode kernel,
 * out[launch][worker][FQSX_TRACE_WORDS]: 8 stamps in 10 ns ticks + 24 per-launch counters / section times of the resolving wave; returns the
 * number of launches copied (always 0 in the product build).  `out` must hold max_launches * T * FQSX_TRACE_WORDS words. */
#define FQSX_TRACE_WORDS 32
int fqsx_dna_trace(fqsx_dna *, uint64_t *out, uint32_t max_launches);

/* Quality stream on the GPU: replaces CQualityCompressor::Init / Compress for all T
 * workers of a block (fqs/quality.h:43-50, fqs/quality.cpp:32-71,152-175; called from fqs/application.cpp:641).
 * quality_mode and quality_thr come from the header bytes 6 and 8; quality_mode none is rejected (nothing
 * to code).  quals = concatenated quality strings of the block's reads (mates interleaved for paired data). */
typedef struct fqsx_qual fqsx_qual;
int fqsx_qual_create(const uint8_t *header17, int device, fqsx_qual **out);
int fqsx_qual_encode_block(fqsx_qual *, const uint8_t *quals, const uint64_t *read_off, uint32_t n_reads,
                           const uint8_t **streams, uint64_t *lens);
/* Same with the block already resident in device memory (d_quals / d_read_off device pointers, h_read_off the host copy). */
int fqsx_qual_encode_block_dev(fqsx_qual *, const uint8_t *d_quals, const uint64_t *d_read_off, const uint64_t *h_read_off,
                               uint32_t n_reads, const uint8_t **streams, uint64_t *lens);
int fqsx_qual_set_profiling(fqsx_qual *, int enable);         /* HIP events around every launch of the quality kernel ... */
int fqsx_qual_kernel_times(fqsx_qual *, double out[2]);       /* ... out[0] = accumulated milliseconds, out[1] = launches */
void fqsx_qual_destroy(fqsx_qual *);

/* Host-side (CPU) read-length stream that accompanies every DNA stream in the container
 * (CMetaCompressor::CompressReadLen, fqs/meta.cpp:48-113; one symbol per read, not part of the hot path).
 * Same worker partition as the DNA path; streams valid until the next call. */
typedef struct fqsx_meta fqsx_meta;
int fqsx_meta_create(uint32_t T, fqsx_meta **out);
int fqsx_meta_encode_block(fqsx_meta *, const uint32_t *read_len, uint32_t n_reads, const uint8_t **streams,
                           uint64_t *lens);
/* paired != 0: reads alternate mate 1 / mate 2 (CompressReadLenPE, fqs/meta.cpp:100-107) */
int fqsx_meta_encode_block_pe(fqsx_meta *, const uint32_t *read_len, uint32_t n_reads, int paired,
                              const uint8_t **streams, uint64_t *lens);
void fqsx_meta_destroy(fqsx_meta *);

/* Read-id stream (host CPU by design: string tokeniser + delta coder).  Replaces
 * CIdCompressor::Init / ResetReadPrev / Compress / CompressPE for all T workers of one block (fqs/id.h:152-164,
 * id.cpp:138-184; application.cpp:600-601,625,635,1165).  header17: T = byte 4, id_mode = byte 7 (0 lossless,
 * 1 instrument, 2 none).  ids = concatenated id lines each including its '\n'; id_off = n_reads+1 offsets;
 * paired != 0: reads alternate mate 1 / mate 2.  Streams are callee-owned until the next call.
 * FQSX_E_ARG for bytes >= 128 in an id (the reference indexes 128-symbol models with them) and, in instrument
 * mode, for an id without any of '.', ' ', ':' (the reference then overwrites the first base). */
typedef struct fqsx_id fqsx_id;
int fqsx_id_create(const uint8_t *header17, fqsx_id **out);
int fqsx_id_encode_block(fqsx_id *, const uint8_t *ids, const uint64_t *id_off, uint32_t n_reads, int paired,
                         const uint8_t **streams, uint64_t *lens);
void fqsx_id_destroy(fqsx_id *);

/* The read-id stream on the GPU: same arguments and the same bytes as fqsx_id_encode_block, coded by one
 * wavefront per worker (csrc/fqsx_idk.h: tokeniser, numeric deltas, move-to-front list of instrument names, adaptive models in
 * per-worker tables in HBM; reference fqs/id.cpp:152-184, 257-495, 734-757).  Staging limits of the kernel: id lines of at most
 * 1024 bytes, 128 tokens, instrument names of 63 bytes, 4096 distinct instrument names per worker (beyond: an error, use the
 * host coder).  id_mode none has no stream. */
typedef struct fqsx_idg fqsx_idg;
int fqsx_idg_create(const uint8_t *header17, int device, fqsx_idg **out);
int fqsx_idg_encode_block(fqsx_idg *, const uint8_t *ids, const uint64_t *id_off, uint32_t n_reads, int paired,
                          const uint8_t **streams, uint64_t *lens);
void fqsx_idg_destroy(fqsx_idg *);

/* Read order of `fqs e -om s` with the string work on the GPU.  Replaces preprocess_se
 * (fqs/application.cpp:349-412: 256 bins by the first four bases, N->T) plus CSortedFASTQFile::sort_reads on every
 * bin (fqs/io.h:499-528).  The GPU radix-sorts the reads by the comparator's keys and gives every read a dense
 * rank; the host then runs the same libstdc++ std::sort per bin on the ranks, which reproduces the reference's
 * order of reads that compare equal (ids and qualities follow it).  order_out[n_reads] = read indices, bin after
 * bin; bin_start[257] = offsets of the bins inside order_out. */
int fqsx_sort_order(const uint8_t *bases, const uint64_t *read_off, uint32_t n_reads, int device,
                    uint32_t *order_out, uint32_t *bin_start);
/* The same with bounded device memory, the way the reference bounds its host memory (bins on disk, sorted one after the
 * other, fqs/application.cpp:349-412, 1595-1626): the reads are binned by their first four bases on the host, consecutive
 * bins are packed into batches of at most max_batch_bases bases (a bin larger than that is a batch of its own), and every
 * batch is uploaded and sorted separately.  Same order as fqsx_sort_order (= max_batch_bases 0: one batch).
 * n_batches_out (optional): GPU passes made. */
int fqsx_sort_order_batched(const uint8_t *bases, const uint64_t *read_off, uint32_t n_reads, int device, uint64_t max_batch_bases,
                            uint32_t *order_out, uint32_t *bin_start, uint32_t *n_batches_out);

/* Host-side read order inside one bin of `fqs e -om s`: std::sort with the comparator of
 * CSortedFASTQFile::sort_reads (fqs/io.h:499-528) applied to reads idx_in[0..n) (indices into off[]),
 * result in idx_out.  Same libstdc++ algorithm on the same initial order = same order of equal reads. */
int fqsx_sort_bin(const uint8_t *bases, const uint64_t *off, const uint32_t *idx_in, uint32_t n, uint32_t *idx_out);

const char *fqsx_last_error(void);
const char *fqsx_version(void);

#ifdef __cplusplus
}
#endif
#endif
