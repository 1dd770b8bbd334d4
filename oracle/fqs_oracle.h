/* oracle/fqs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement of the FQSqueezer 1.1 DNA-stream encoder
 * (reference: /root/reference/fqs/dna.cpp, code_ctx.cpp, ht_kmer.h, bit_vec.h,
 * context_hm.h, rc.h, sub_rc.h, kmer.h, utils.h, application.cpp:610-669).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product library (libfqsx.so) never links or calls it.
 *
 * Parity status: PINNED -- the restatement is checked byte-for-byte against
 * DNA streams produced by the compiled, unmodified reference (oracle/_ref/fqs-1.1)
 * via the golden fixtures under tests/golden/ (tests/test_oracle_golden.py).
 */
#ifndef FQS_ORACLE_H
#define FQS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fqo_codec fqo_codec;

/* header17 = the 17 parameter bytes of a .fqs file ('K','C','S','D', T, dna_mode, ...;
 * reference fqs/params.h:80-100).  Returns NULL on a malformed header or an
 * unsupported dna_mode (PE modes are not restated yet). */
fqo_codec *fqo_create(const uint8_t *header17);
void fqo_destroy(fqo_codec *);

/* Encode one reads block (reference application.cpp:610-669 for all T workers).
 * bases   : concatenated ASCII read sequences (ACGTN), no separators
 * read_off: n_reads+1 offsets into bases
 * generation: index of the block in the file (drives the sync schedule,
 *             application.h:85-92)
 * After the call fqo_stream(w) returns worker w's complete DNA range-coder
 * stream for this block (valid until the next encode call). */
int fqo_encode_block(fqo_codec *, const uint8_t *bases, const uint64_t *read_off,
                     uint32_t n_reads, uint32_t generation);
const uint8_t *fqo_stream(fqo_codec *, uint32_t worker, uint64_t *len);

/* Decode one reads block (reference decoder workers, application.cpp:874-917; DecompressSE/PE,
 * dna.cpp:1883-2044): streams[w]/lens[w] = worker w's DNA stream, read_off = n_reads+1 offsets of
 * the reads inside bases_out (the read lengths come from the meta stream).  A codec object is
 * used either for encoding or for decoding, never both. */
int fqo_decode_block(fqo_codec *, const uint8_t *const *streams, const uint64_t *lens, const uint64_t *read_off,
                     uint32_t n_reads, uint32_t generation, uint8_t *bases_out);

/* Counters accumulated over all encode calls (SURVEY.md §8d accounting):
 * [0] global k-mer table probes (cluster scans)   [1] slots scanned by them
 * [2] global inserts  [3] siv word ops (incl. prefix scan words)
 * [4] context look-ups [5] symbols range-coded [6] local-table probes
 * [7] local inserts */
void fqo_counters(fqo_codec *, uint64_t out[8]);
/* Coverage of the rarely taken branches: [0..5] find_counts results per counts_level_t (none, pmer, smer, bmer,
 * mixed, bmer_unc; defs.h:45, dna.cpp:457-502), [6..9] random draws of cinc_b / cinc_s / cinc_lb / cinc_ls (counters
 * above their thresholds, utils.h:272-325). */
void fqo_levels(fqo_codec *, uint64_t out[10]);

/* Quality stream (CQualityCompressor, quality.cpp:152-175): same header bytes
 * (quality_mode = byte 6, quality_thr = byte 8), same worker partition, no synchronisation points. */
typedef struct fqo_qual fqo_qual;
fqo_qual *fqo_qual_create(const uint8_t *header17);
void fqo_qual_destroy(fqo_qual *);
int fqo_qual_encode_block(fqo_qual *, const uint8_t *quals, const uint64_t *read_off, uint32_t n_reads);
const uint8_t *fqo_qual_stream(fqo_qual *, uint32_t worker, uint64_t *len);

/* Known-answer helpers for unit tests */
void fqo_kat_mt19937(uint32_t seed, uint32_t n, uint32_t *out);
void fqo_kat_cinc(uint32_t thr, uint32_t mult, uint32_t maxv, uint32_t n,
                  const uint32_t *a, const uint32_t *b, uint32_t *out);
uint64_t fqo_kat_rc(uint32_t n, const uint32_t *freq, const uint32_t *cum,
                    const uint32_t *tot, uint8_t *out, uint64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
