/* agc_cmp.h -- C ABI of the host-side compressor (libagc_host.so on top of libagc_hip.so): the reference's
 * CAGCCompressor surface for `create` / `append` (src/core/agc_compressor.h:754-763, called from src/app/main.cpp:76-168)
 * as plain C entry points, plus the additions for HBM-resident inputs and multi-GPU jobs.
 * Every function returns 1 for success and 0 for failure (the reference's bool; the message goes to stderr), unless noted.
 * There is no CPU fallback: agc_cmp_create / agc_cmp_append fail when no HIP device can be opened.
 */
#ifndef AGC_CMP_H
#define AGC_CMP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one compressor per archive and GPU; device = HIP device ordinal */
void *agc_cmp_new(int device);
void agc_cmp_delete(void *h);

/* CAGCCompressor::Create (agc_compressor.cpp:2273-2327).  out_path "" = produce and discard the archive bytes, "-" = stdout;
 * ref_file "" = splitters come from agc_cmp_set_splitters / agc_cmp_set_reference_dev. */
int agc_cmp_create(void *h, const char *out_path, uint32_t pack_cardinality, uint32_t k, const char *ref_file, uint32_t segment_size,
                   uint32_t min_match_len, int concatenated, int adaptive, uint32_t verbosity, uint32_t n_threads, double fallback_frac);
/* CAGCCompressor::Append (agc_compressor.cpp:2330-2374) */
int agc_cmp_append(void *h, const char *in_archive, const char *out_archive, uint32_t verbosity, int concatenated, int adaptive,
                   uint32_t n_threads);
/* determine_splitters' result supplied by the caller (sorted or not, duplicates allowed) */
int agc_cmp_set_splitters(void *h, const uint64_t *kmers, uint64_t n);
/* determine_splitters (agc_compressor.cpp:428-563) on the GPU for a reference genome resident in HBM:
 * contig c = d_codes[ctg_off[c] .. ctg_off[c+1]), one symbol code per byte */
int agc_cmp_set_reference_dev(void *h, const uint8_t *d_codes, const uint64_t *ctg_off, uint32_t n_ctg);
/* CAGCCompressor::AddSampleFiles (agc_compressor.cpp:2118-2270): n (sample name, FASTA path) pairs */
int agc_cmp_add_sample_files(void *h, uint32_t n, const char **sample_names, const char **paths, uint32_t n_threads);
/* one sample whose contigs are resident in HBM = agc_cmp_prepare_sample_dev + agc_cmp_commit_prepared */
int agc_cmp_add_sample_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const uint8_t *d_codes,
                           const uint64_t *ctg_off);
/* first half: scan, classification and LZ encode of the segments whose group is known, against the present state (d_codes must
 * stay untouched until the commit); second half: registration and store, after revalidating what changed in between */
int agc_cmp_prepare_sample_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const uint8_t *d_codes,
                               const uint64_t *ctg_off);
int agc_cmp_commit_prepared(void *h);
/* ... in two steps (multi-GPU mode): after _head the sample is registered and agc_cmp_last_record (the record's head) is ready --
 * the other ranks can apply it and go on --; _finish indexes the new references on this GPU, encodes what the speculative encode
 * did not cover, and builds agc_cmp_last_record_body */
int agc_cmp_commit_prepared_head(void *h);
int agc_cmp_commit_prepared_finish(void *h);
/* the same for a sample resident in HBM in the 2-bit layout (packed: const agc_hip_packed *, include/agc_hip.h; contig c =
 * symbols [ctg_off[c], ctg_off[c+1]) of the packed buffer) */
int agc_cmp_add_sample_packed_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const void *packed,
                                  const uint64_t *ctg_off);
int agc_cmp_prepare_sample_packed_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const void *packed,
                                      const uint64_t *ctg_off);
/* announces the packed sample that will be added NEXT: its expansion and splitter scan run on the device beside the sample
 * added in between (the reference's workers take later contigs from the queue while earlier ones register,
 * agc_compressor.cpp:1093-1272); 0 when not applicable (adaptive / append / -c mode): the sample then goes the ordinary way */
int agc_cmp_set_next_sample_packed_dev(void *h, const void *packed, const uint64_t *ctg_off, uint32_t n_ctg);
/* announces a sample that is still THE BYTES OF ITS FASTA FILE in HBM (contig c = the sequence lines raw[raw_begin[c] .. raw_end[c]),
 * as genome_io hands them to the reference's workers, agc_compressor.cpp:2160-2228): preprocess_raw_contig (agc_compressor.cpp:907-951)
 * + the 2-bit packing into the caller's buffers (sizes: agc_hip_packed_words_bytes / _index_bytes; agc_hip_pack_fasta_begin's
 * arguments).  The compressor queues the conversion itself, at the point of the sample call in progress where the GPU has room for
 * it -- the reference's reader runs ahead of its workers the same way (agc_compressor.cpp:2155-2238) --, or inside
 * agc_cmp_finish_fasta_dev when no sample call came in between.  agc_cmp_finish_fasta_dev waits for it: the contigs' symbol
 * offsets (n_ctg + 1) and the number of escaped blocks; returns agc_hip_pack_fasta_end's code (AGC_HIP_OK; AGC_HIP_ECAP: announce
 * again with room for *n_esc_blocks escaped blocks).  One conversion at a time; 1 / 0 from the first call. */
int agc_cmp_set_next_fasta_dev(void *h, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *raw_begin, const uint64_t *raw_end, uint32_t n_ctg,
                               uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks);
int agc_cmp_finish_fasta_dev(void *h, uint64_t *ctg_off, uint64_t *n_esc_blocks);
/* CAGCCompressor::Close (agc_compressor.cpp:2094-2115, 2386-2400) */
int agc_cmp_close(void *h, uint32_t n_threads);
/* The entropy stage runs beside the add calls (the reference's workers compress a pack when it fills while the others go on,
 * segment.h:172-215; here one background thread drives the GPU zstd kernel and a host pool).  agc_cmp_drain waits until every
 * part handed over so far is compressed and written -- never needed for correctness (agc_cmp_close drains), only to put a
 * clean line between two measured regions. */
int agc_cmp_drain(void *h);
/* Close in steps (entropy stage spread over several GPUs): the inputs of the pending delta packs a device may compress
 * (pack i = src[off[i] .. off[i+1]), n packs), then their level-17 zstd frames in the same order (frame i =
 * frames[off[i] .. off[i+1])), then agc_cmp_close.  The buffers returned by the first call stay valid until agc_cmp_close. */
int agc_cmp_close_collect_packs(void *h, const uint8_t **src, const uint64_t **off, uint32_t *n);
int agc_cmp_close_provide_frames(void *h, const uint8_t *frames, const uint64_t *off);
/* the same in the middle of a run (writer rank of an N-rank job): the reference's workers code a delta pack the moment it is full
 * while the others go on (segment.cpp:34-80, segment.h:258-280); here the full packs wait (their parts already hold their places
 * in the archive) until the caller DEALS them to the ranks' entropy stages.  agc_cmp_deferred_pack_bytes: how much has piled up;
 * agc_cmp_deal_collect_packs: all of it as one deal (inputs back to back, pack i = src[off[i] .. off[i+1]); valid until every pack
 * of the deal is settled); a share [first, first + count) settles either with agc_cmp_deal_provide_frames (its level-17 frames,
 * frame t = frames[off[t] .. off[t+1]), from whichever rank coded them, any number of samples later) or with agc_cmp_deal_keep_own
 * (this rank's own entropy stage takes it, beside its steps).  Shares settle in any order; agc_cmp_close needs every deal settled. */
uint64_t agc_cmp_deferred_pack_bytes(void *h);
int agc_cmp_deal_collect_packs(void *h, uint32_t *deal_id, const uint8_t **src, const uint64_t **off, uint32_t *n);
int agc_cmp_deal_keep_own(void *h, uint32_t deal_id, uint32_t first, uint32_t count);
int agc_cmp_deal_provide_frames(void *h, uint32_t deal_id, uint32_t first, uint32_t count, const uint8_t *frames, const uint64_t *off);

/* one archive from N ranks (before agc_cmp_create on every rank; protocol: agc_amd/dist.py, compressor_dist.cpp).
 * After agc_cmp_add_sample_dev / agc_cmp_commit_prepared on the owner, agc_cmp_last_record gives the bytes every other rank
 * must pass to agc_cmp_apply_record (d_record: optional copy in that rank's HBM, or NULL). */
int agc_cmp_set_distributed(void *h, uint32_t rank, uint32_t world_size, uint32_t writer_rank);
int agc_cmp_last_record(void *h, const uint8_t **ptr, uint64_t *n);
/* the same bytes (pinned host memory) with 64 bytes in front of them that belong to the caller's transport: a fixed-size message
 * header written there lets header and head travel as ONE message (agc_amd/dist.py).  *n counts the 64 bytes. */
int agc_cmp_last_record_framed(void *h, uint8_t **ptr, uint64_t *n);
/* the record has a HEAD every rank applies (group ids, keys, the newly minted reference segments) and a BODY (the LZ deltas) only
 * the writer rank needs: body / body_n = NULL / 0 on the other ranks */
int agc_cmp_apply_record(void *h, const uint8_t *record, uint64_t n, const uint8_t *d_record, const uint8_t *body, uint64_t body_n);
int agc_cmp_last_record_body(void *h, const uint8_t **ptr, uint64_t *n);
/* writer rank: a pinned host buffer of n bytes to receive the next record's body into; agc_cmp_apply_record takes it over (no
 * copy) when its `body` argument is this pointer.  The bookkeeping that reads it runs beside the next sample's commit. */
int agc_cmp_record_body_buffer(void *h, uint64_t n, uint8_t **ptr);

/* version string of the libzstd in use (archives are byte-identical to the reference's only with the same libzstd) */
const char *agc_cmp_zstd_version(void *h);
/* the agc_hip_ctx (include/agc_hip.h) behind this compressor, e.g. for agc_hip_timing_* */
void *agc_cmp_hip_ctx(void *h);
/* counters and stage times in the order of agc::CompressorStats (agc_amd/host.py: STAT_NAMES); returns how many exist */
int agc_cmp_stats(void *h, double *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
