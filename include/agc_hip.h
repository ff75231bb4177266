/*
 * agc_hip.h -- C ABI of the MI355X-native (gfx950) implementation of AGC's
 * segment-compression hot path.  Plain pointers and sizes only; every entry
 * point returns 0 on success and a negative AGC_HIP_E* code on failure
 * (agc_hip_last_error() gives the text).  There is NO CPU fallback behind this
 * interface: without a HIP device every compute call fails with AGC_HIP_ENODEV.
 *
 * The reference (refresh-bio/agc v3.2.2) has no FFI seam for this path; the
 * functions below sit exactly where the reference calls its private C++
 * members, each one citing the reference interface it replaces
 * (file:line under the reference tree).
 *
 * Symbols: the codes of src/common/agc_basic.h:40-50 (A0 C1 G2 T3 N4 IUPAC5..15, 30 other letter, 32 '@'/'`').
 * In HBM a sample lives in the 2-bit layout (agc_hip_packed below: 0.25 byte per symbol, blocks holding anything
 * outside ACGT kept verbatim); group references are kept the same way.  EVERY LZ kernel reads that layout only: the
 * *_packed entry points take sequences of a packed sample as they are, the *_dev / host variants taking one byte per
 * symbol pack what they are given first (a convenience for small inputs and tests, not the fast path).
 * Pointer naming: h_* = host memory, d_* = device (HBM) memory.
 */
#ifndef AGC_HIP_H
#define AGC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGC_HIP_OK 0
#define AGC_HIP_ENODEV (-1)   /* no HIP device / runtime error            */
#define AGC_HIP_EINVAL (-2)   /* bad argument                             */
#define AGC_HIP_ENOMEM (-3)   /* device or host allocation failed         */
#define AGC_HIP_ECAP (-4)     /* caller-provided output buffer too small  */
#define AGC_HIP_ENOREF (-5)   /* group id has no registered reference     */

typedef struct agc_hip_ctx agc_hip_ctx;

/* ---- context ---------------------------------------------------------- */
int agc_hip_create(agc_hip_ctx **out, int device);
void agc_hip_destroy(agc_hip_ctx *ctx);
const char *agc_hip_last_error(const agc_hip_ctx *ctx);
/* ABI version of this header (checked by the host bindings). */
uint32_t agc_hip_abi_version(void);
/* Block until every kernel / copy issued through ctx has finished. */
int agc_hip_sync(agc_hip_ctx *ctx);

/* Per-kernel timing with HIP events on the stream the kernels run on.
 * which: one of AGC_HIP_K_*.  ms = accumulated kernel time, launches = count. */
enum {
    AGC_HIP_K_SCAN = 0,
    AGC_HIP_K_INDEX = 1,
    AGC_HIP_K_ENCODE = 2,
    AGC_HIP_K_ESTIMATE = 3,
    AGC_HIP_K_COSTVEC = 4,
    AGC_HIP_K_REVCOMP = 5,
    AGC_HIP_K_PREPROCESS = 6,
    AGC_HIP_K_REFSTORE = 7,
    AGC_HIP_K_ZSTD = 8,
    AGC_HIP_K_FILTER = 9, /* key_filter_kernel: the "may match" bitmaps of the estimate / cost-vector parses */
    AGC_HIP_K_SEGMENTS = 10, /* agc_hip_segments_packed: hits -> segments -> group look-up -> encode descriptors (seg_kernels.hip) */
    AGC_HIP_K_PACK = 11,     /* agc_hip_pack_fasta_*: raw FASTA bodies -> the 2-bit packed sample in one pass (pack_kernels.hip) */
    AGC_HIP_K_COUNT = 12
};
int agc_hip_timing_enable(agc_hip_ctx *ctx, int on);
int agc_hip_timing_reset(agc_hip_ctx *ctx);
int agc_hip_timing_get(agc_hip_ctx *ctx, int which, double *ms, uint64_t *launches);

/* ---- sample staging (host-resident inputs) ----------------------------- */
/* A context-owned HBM buffer of at least `bytes` bytes (grown on demand, reused between
 * calls; valid until the next agc_hip_sample_buffer call) and a synchronous host->HBM copy.
 * Used by hosts that read FASTA files (the contig_t vectors the reference workers own,
 * src/core/agc_compressor.cpp:1239-1246). */
int agc_hip_sample_buffer(agc_hip_ctx *ctx, uint64_t bytes, uint8_t **d_ptr);
int agc_hip_copy_to_device(agc_hip_ctx *ctx, uint8_t *d_dst, const uint8_t *h_src, uint64_t n);

/* ---- a1: raw FASTA body -> symbol codes ------------------------------- */
/* Replaces CAGCCompressor::preprocess_raw_contig (src/core/agc_compressor.cpp:907-951):
 * drops every byte < 64, maps the rest through cnv_num (agc_basic.h:40-50).
 * d_raw/d_codes are device buffers of n_raw bytes; *h_n_codes receives the count. */
int agc_hip_preprocess_dev(agc_hip_ctx *ctx, const uint8_t *d_raw, uint64_t n_raw,
                           uint8_t *d_codes, uint64_t *h_n_codes);

/* The same for a raw body in host memory (copied to HBM first): d_codes receives the codes, *h_n_codes their number. */
int agc_hip_preprocess(agc_hip_ctx *ctx, const uint8_t *h_raw, uint64_t n_raw, uint8_t *d_codes, uint64_t *h_n_codes);

/* ---- S1: splitter scan (a2-a4) ---------------------------------------- */
/* Replaces the splitter set hs_splitters + bloom_splitters
 * (src/core/agc_compressor.h:625-626; hs.h:448-497; utils_adv.h:180-282):
 * exact membership of canonical k-mers (left-aligned u64, kmer.h:360-362). */
int agc_hip_splitters_set(agc_hip_ctx *ctx, const uint64_t *h_kmers, uint64_t n);
int agc_hip_splitters_insert(agc_hip_ctx *ctx, const uint64_t *h_kmers, uint64_t n);
uint64_t agc_hip_splitters_count(const agc_hip_ctx *ctx);

/* Replaces CAGCCompressor::determine_splitters (src/core/agc_compressor.cpp:428-563; enumerate_kmers
 * :630-660, remove_non_singletons :664-704, find_splitters_in_contig :762-825) for reference contigs
 * resident in HBM (absolute offsets < 2^32): canonical k-mers -> radix sort -> singletons -> in every
 * contig the first singleton once >= segment_size symbols have passed since the previous splitter, and
 * the right-most singleton of the tail.  h_splitters receives the sorted, unique splitters
 * (AGC_HIP_ECAP + needed count if cap is too small).  When h_sorted_kmers is given (adaptive mode keeps
 * v_candidate_kmers / v_duplicated_kmers, :493-497) it receives ALL canonical k-mers of the reference,
 * sorted, duplicates included; the caller splits them into singletons and duplicated values. */
int agc_hip_determine_splitters_dev(agc_hip_ctx *ctx, const uint8_t *d_codes, const uint64_t *h_ctg_off,
                                    uint32_t n_ctg, uint32_t k, uint32_t segment_size,
                                    uint64_t cap, uint64_t *h_splitters, uint64_t *h_n_splitters,
                                    uint64_t sorted_cap, uint64_t *h_sorted_kmers, uint64_t *h_n_sorted);

/* Replaces the loop body of CAGCCompressor::compress_contig
 * (src/core/agc_compressor.cpp:2007-2036) for a batch of contigs:
 * d_codes holds the contigs back to back, contig c = [h_ctg_off[c], h_ctg_off[c+1]).
 * Emits the ACCEPTED splitter hits in (contig, position) order -- i.e. after the
 * reference's "reset the k-mer after every hit" rule -- with the k-mer at the hit
 * (dir and rev-comp forms, both left-aligned as CKmer keeps them).
 * Segment boundaries follow as in SURVEY.md App. A.3.
 * Returns AGC_HIP_ECAP (and the needed count in *h_n_hits) if cap is too small. */
int agc_hip_scan_contigs_dev(agc_hip_ctx *ctx, const uint8_t *d_codes,
                             const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                             uint64_t cap, uint64_t *h_n_hits,
                             uint32_t *h_hit_ctg, uint64_t *h_hit_pos,
                             uint64_t *h_hit_dir, uint64_t *h_hit_rc);
/* Same with host-resident contigs (copied to HBM first). */
int agc_hip_scan_contigs(agc_hip_ctx *ctx, const uint8_t *h_codes,
                         const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                         uint64_t cap, uint64_t *h_n_hits,
                         uint32_t *h_hit_ctg, uint64_t *h_hit_pos,
                         uint64_t *h_hit_dir, uint64_t *h_hit_rc);

/* ---- 2-bit packed samples (the HBM-resident form of contigs) -------------- */
/* Symbol i of the buffer (contigs back to back, codes of agc_basic.h:40-50) at bits [2*(i & 15), +1] of 32-bit word i >> 4.
 * Blocks of 1024 symbols containing anything outside ACGT are kept verbatim (one byte per symbol) in d_esc_bytes;
 * d_esc_index[block] = their slot (x 1024 bytes) or -1.  CKmer does the same packing symbol by symbol
 * (src/core/kmer.h:284-301); this is the layout the sample itself has in HBM. */
typedef struct {
    const uint32_t *d_words;      /* agc_hip_packed_words_bytes(n_symbols) bytes */
    const int32_t *d_esc_index;   /* agc_hip_packed_index_bytes(n_symbols) bytes */
    const uint8_t *d_esc_bytes;   /* 1024 bytes per escaped block */
    uint64_t n_symbols;
} agc_hip_packed;
uint64_t agc_hip_packed_words_bytes(uint64_t n_symbols);
uint64_t agc_hip_packed_index_bytes(uint64_t n_symbols);
/* codes (1 B per symbol, device) -> packed.  AGC_HIP_ECAP (+ the needed count in *h_n_esc_blocks) when more than
 * esc_cap_blocks blocks have to be escaped. */
int agc_hip_pack_dev(agc_hip_ctx *ctx, const uint8_t *d_codes, uint64_t n_symbols, uint32_t *d_words, int32_t *d_esc_index,
                     uint8_t *d_esc_bytes, uint64_t esc_cap_blocks, uint64_t *h_n_esc_blocks);
/* Raw FASTA bodies (device) -> packed, in ONE pass over the input: preprocess_raw_contig (src/core/agc_compressor.cpp:907-951 --
 * every byte < 64, i.e. the line ends, dropped; the others through cnv_num, src/common/agc_basic.h:39-49) fused with the packing,
 * no intermediate one-byte-per-symbol buffer.  Contig c = the raw bytes [h_raw_begin[c], h_raw_end[c]) of d_raw -- its sequence
 * lines as the file holds them, what genome_io hands to the reference's workers (src/core/agc_compressor.cpp:2160-2228); the
 * ranges ascend and do not overlap, whatever lies between them (header lines) is skipped.  The contigs land back to back in the
 * packed buffer; h_ctg_off[0..n_ctg] receives their symbol offsets (h_ctg_off[n_ctg] = n_symbols).  The output buffers must hold
 * agc_hip_packed_words_bytes / _index_bytes of an upper bound of the symbol count (the kept bytes: <= the sum of the range
 * lengths); d_raw and d_words 16-byte aligned.  AGC_HIP_ECAP (+ the needed count) when more than esc_cap_blocks blocks are escaped.
 * begin queues the work on a stream of its own and returns; end waits and delivers (one pack in flight per context; the buffers
 * must not be touched in between).  Every other entry point may be used meanwhile. */
int agc_hip_pack_fasta_begin(agc_hip_ctx *ctx, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end,
                             uint32_t n_ctg, uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks);
int agc_hip_pack_fasta_end(agc_hip_ctx *ctx, uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks);
int agc_hip_pack_fasta_dev(agc_hip_ctx *ctx, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end,
                           uint32_t n_ctg, uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks,
                           uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks);
/* The same for a host that reads FASTA files: the sequence lines of the window's contigs, in HOST memory as the reader cut them
 * out of the file (contig c = h_raw[c], h_len[c] bytes: what genome_io hands to the reference's workers,
 * src/core/agc_compressor.cpp:2160-2228), are uploaded as they are and converted + packed on the device into buffers the CONTEXT
 * owns (*out names them; valid until the next agc_hip_sample_pack / agc_hip_sample_pack_fasta; an encode left in flight on the
 * previous sample is waited for first).  h_ctg_off receives the contigs' symbol offsets (n_ctg + 1).  Replaces, per window,
 * preprocess_raw_contig (agc_compressor.cpp:907-951) of every contig + agc_hip_sample_pack. */
int agc_hip_sample_pack_fasta(agc_hip_ctx *ctx, uint32_t n_ctg, const uint8_t *const *h_raw, const uint64_t *h_len, agc_hip_packed *out,
                              uint64_t *h_ctg_off);
/* packed -> codes (d_codes: n_symbols bytes, 16-byte aligned).  Nothing on the create path needs it (the LZ kernels read the
 * packed form); a utility for callers and tests.  Asynchronous on the context's stream (ordered before every later call). */
int agc_hip_expand_dev(agc_hip_ctx *ctx, const agc_hip_packed *pk, uint8_t *d_codes);
/* codes (device, one byte per symbol) -> packed, into buffers the CONTEXT owns (*out names them; valid until the next
 * agc_hip_sample_pack): what a host that reads FASTA files calls once per uploaded sample before it uses the *_packed entry
 * points.  An encode left in flight on the previous sample (agc_hip_lz_encode_begin_packed) is waited for first. */
int agc_hip_sample_pack(agc_hip_ctx *ctx, const uint8_t *d_codes, uint64_t n_symbols, agc_hip_packed *out);
/* agc_hip_scan_contigs_dev on a packed sample (contig c = symbols [h_ctg_off[c], h_ctg_off[c+1]) of the buffer), 16 <= k <= 32.
 * Same results; reads 0.25 B per symbol. */
int agc_hip_scan_packed_dev(agc_hip_ctx *ctx, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                            uint64_t cap, uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos,
                            uint64_t *h_hit_dir, uint64_t *h_hit_rc);
/* The NEXT sample ahead of its turn.  The reference's workers take contigs of later samples from the priority queue while
 * earlier ones are being registered (src/core/agc_compressor.cpp:1093-1272); here the caller that knows its next sample hands
 * it over early: agc_hip_prefetch_packed_dev QUEUES, on a stream of its own and without waiting, the splitter scan of
 * agc_hip_scan_packed_dev into a hit list of its own -- it runs in the gaps the entry points of the sample in front leave on
 * the GPU.  agc_hip_scan_prefetched then waits for that work and delivers the hits exactly as agc_hip_scan_packed_dev would
 * (same arguments, same results; AGC_HIP_EINVAL when this sample was not the one prefetched).  The splitter set must not
 * change in between (not for adaptive mode). */
int agc_hip_prefetch_packed_dev(agc_hip_ctx *ctx, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k);
int agc_hip_scan_prefetched(agc_hip_ctx *ctx, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                            uint64_t cap, uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir,
                            uint64_t *h_hit_rc);

/* ---- S1 -> S2 on the device: segments and their groups (a3's cut, a5's first decision) ------------------------ */
/* The (k-mer 1, k-mer 2) -> group table, map_segments of the reference (src/core/agc_compressor.h:628; filled by store_segments,
 * agc_compressor.cpp:1003-1028, read by add_segment :1275-1330), as an open-addressing array in HBM: n_slots (a power of two)
 * slots, home slot of a key = agc_hip_group_hash(k1, k2) & (n_slots - 1), linear probing, `used` = 0 ends a chain.  The host
 * keeps the authoritative copy (it mints the groups) and mirrors it: _set replaces the whole array, _update overwrites the
 * slots named in h_idx (the handful a registration changed). */
typedef struct {
    uint64_t k1, k2;
    int32_t gid;
    uint32_t used;
} agc_hip_group_slot;
uint64_t agc_hip_group_hash(uint64_t k1, uint64_t k2);
int agc_hip_group_map_set(agc_hip_ctx *ctx, const agc_hip_group_slot *h_slots, uint64_t n_slots);
int agc_hip_group_map_update(agc_hip_ctx *ctx, uint32_t n, const uint64_t *h_idx, const agc_hip_group_slot *h_slots);

/* One segment as compress_contig cuts it (agc_compressor.cpp:2018-2048) with add_segment's first decision (:1286-1301):
 * contig, start (relative to the contig) and length; the k-mers in front and at the back (dir / rev-comp forms, left-aligned as
 * CKmer keeps them; *_full = 0: the segment starts / ends at a contig end); for a segment with both k-mers: store_rc = the
 * orientation rule of the key (min, max) of the two canonical k-mers, map_gid = the group the table holds for that key or -1;
 * encoded = its LZ encode against that group's reference was launched by the call (see below). */
typedef struct {
    uint64_t start;
    uint64_t front_dir, front_rc, back_dir, back_rc;
    uint32_t ctg, len;
    int32_t map_gid;
    uint8_t front_full, back_full, store_rc, encoded;
} agc_hip_segment;
/* A whole sample from splitter hits to classified segments without the host in between: the packed scan (or, prefetched != 0,
 * the collection of the scan agc_hip_prefetch_packed_dev queued), the "reset the k-mer after a hit" rule, the cut, the key of
 * every segment with two splitters and its look-up in the table above (bucket ranges of the table staged in LDS), all on the
 * device; h_segs receives the segments in (contig, position) order.  encode_known != 0: the LZ encode of every segment whose
 * group the table knows is LAUNCHED from here as well, on the context's second lane, from descriptors made on the device
 * (longest first) -- exactly as agc_hip_lz_encode_begin_packed would for those segments; agc_hip_lz_encode_end collects the
 * *h_n_encoded deltas, in the order of the segments flagged `encoded`.  What is left to the host are the segments that need
 * more than a table look-up (one splitter, destroyed middle splitter, new groups).
 * AGC_HIP_ECAP (+ the needed capacity in *h_n_segs) when cap is too small: nothing was launched, call again. */
int agc_hip_segments_packed(agc_hip_ctx *ctx, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                            int prefetched, int encode_known, uint64_t cap, agc_hip_segment *h_segs, uint64_t *h_n_segs,
                            uint32_t *h_n_encoded);
/* The launch of encode_known as a call of its own, for a caller that wants the segment table first (its second lane may still
 * be handing the previous sample's deltas over): the segments of the LAST agc_hip_segments_packed call (made with
 * encode_known == 0) are still on the device; the ones whose group the table knew and whose reference is registered are
 * encoded exactly as above -- these are the segments with two splitters and map_gid >= 16 (the caller tells them from its
 * table).  Nothing is waited for. */
int agc_hip_segments_encode_known(agc_hip_ctx *ctx);

/* ---- S2: LZ-diff against group references (a10, a11, a6, a7) ---------- */
/* A "slice" names one sequence inside a device buffer: symbols [off, off+len) of a packed sample *pk (the *_packed
 * entry points) or d_base[off .. off+len) of a byte buffer (*_dev), read reverse-complemented when rc != 0
 * (CAGCBasic::reverse_complement_copy, src/common/agc_basic.cpp:282-315).  Nothing is copied or staged per slice: the
 * kernels read the packed words where they lie, in either orientation.
 * Each *_packed entry point below takes the same arguments as the *_dev one of the same name, with (pk) in place of (d_base),
 * and delivers the same results. */

/* Replaces CLZDiffBase::Prepare + prepare_index (src/common/lz_diff.cpp:48-78,
 * 81-141, 375-428) as used by CSegment::add for the first sequence of a group
 * (src/common/segment.cpp:41-48): keeps the reference in HBM and builds the
 * sparse k-mer index (same slots, same 64-probe cut-off as the reference).
 * Group ids may be registered once; re-registration returns AGC_HIP_EINVAL. */
int agc_hip_ref_register(agc_hip_ctx *ctx, uint32_t gid, const uint8_t *h_ref,
                         uint32_t n, uint32_t min_match_len);
int agc_hip_ref_register_batch_dev(agc_hip_ctx *ctx, uint32_t n_refs, const uint32_t *h_gid,
                                   const uint8_t *d_base, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc,
                                   uint32_t min_match_len);
int agc_hip_ref_register_batch_packed(agc_hip_ctx *ctx, uint32_t n_refs, const uint32_t *h_gid,
                                      const agc_hip_packed *pk, const uint64_t *h_off,
                                      const uint32_t *h_len, const uint8_t *h_rc,
                                      uint32_t min_match_len);
/* Copies a registered reference back (GetReference, lz_diff.cpp:431-437);
 * also returns the index for parity checks: table entries hold pos/4 exactly
 * as ht16/ht32 do (0xFFFF / 0xFFFFFFFF = empty), widened to u32. */
int agc_hip_ref_get(agc_hip_ctx *ctx, uint32_t gid, uint8_t *h_ref, uint32_t cap, uint32_t *h_n);
int agc_hip_ref_index_get(agc_hip_ctx *ctx, uint32_t gid, uint32_t *h_table, uint64_t cap,
                          uint64_t *h_ht_size, int *h_is16);

/* Replaces CLZDiff_V2::Encode (src/common/lz_diff.cpp:669-798), one call for a
 * batch of segments.  Encoded deltas are written back to back into h_enc;
 * delta s = h_enc[h_enc_off[s] .. h_enc_off[s+1]).  An empty delta means "equal
 * to the reference" (lz_diff.cpp:678-680). */
int agc_hip_lz_encode_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                const uint8_t *d_base, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc,
                                uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off);
int agc_hip_lz_encode_batch_packed(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                   const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc,
                                   uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off);
/* Host-resident texts: text s = h_text[h_off[s] .. h_off[s]+h_len[s]). */
int agc_hip_lz_encode_batch(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                            const uint8_t *h_text, const uint64_t *h_off,
                            const uint32_t *h_len, const uint8_t *h_rc,
                            uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off);

/* The encode of agc_hip_lz_encode_batch_dev in two halves, for a caller that has other work for the device and for itself
 * while a whole sample is encoded (the reference's worker threads encode segments while others are still being classified,
 * agc_compressor.cpp:989-1050).  begin queues the work on the context's second stream and returns at once; between begin and
 * end every other entry point may be used (own stream, own scratch); end waits and delivers exactly what
 * agc_hip_lz_encode_batch_dev would have (AGC_HIP_ECAP: h_enc_off[n] holds the size needed, call end again).  One encode in
 * flight per context (a begin while one is in flight drops the earlier one).  The packed sample (begin_packed) and every reference
 * named must stay unchanged until end returns -- with one exception the library takes care of itself: the buffers of
 * agc_hip_sample_pack may be handed the next sample at once, that call waits for the parse.  begin_dev packs its byte input into
 * a buffer of the second lane before it returns to the caller's stream, so d_base need not outlive the call's device work.
 * end may be called from ANOTHER thread than the one that goes on using the context (it touches nothing but the second lane's
 * state); that thread must not call begin again before end has returned. */
int agc_hip_lz_encode_begin_dev(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc);
int agc_hip_lz_encode_begin_packed(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc);
int agc_hip_lz_encode_end(agc_hip_ctx *ctx, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off);
/* number of deltas the encode in flight will deliver (0: none in flight); waits for the parse when the encode was launched from
 * descriptors made on the device (agc_hip_segments_encode_known), whose number only the device knows until then. */
int agc_hip_lz_encode_pending(agc_hip_ctx *ctx, uint32_t *h_n);
/* The same three on a named lane.  A context has AGC_HIP_ENCODE_LANES encode lanes (streams with their own scratch), each with
 * one encode in flight; the entry points above are lane 0, which is also the lane agc_hip_segments_encode_known launches on.
 * A host that commits a sample while its whole-sample encode is still on lane 0 puts the remaining segments -- those whose group
 * the sample itself minted (the reference's followers of a new group, agc_compressor.cpp:1275-1330) -- on lane 1 and lets the
 * thread that collects lane 0 collect lane 1 after it, instead of waiting for them.  The rules of begin / end hold per lane. */
#define AGC_HIP_ENCODE_LANES 2
int agc_hip_lz_encode_begin_packed_on(agc_hip_ctx *ctx, uint32_t lane, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk,
                                      const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc);
int agc_hip_lz_encode_end_on(agc_hip_ctx *ctx, uint32_t lane, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off);
int agc_hip_lz_encode_pending_on(agc_hip_ctx *ctx, uint32_t lane, uint32_t *h_n);
/* gives up the encode in flight on a lane (waits for its kernels, delivers nothing): a host that prepared a sample ahead of its
 * turn and has to prepare it again (adaptive mode in the N-rank mode: the splitter set grew meanwhile) */
int agc_hip_lz_encode_drop_on(agc_hip_ctx *ctx, uint32_t lane);

/* Pinned host memory for result buffers (device-to-host copies into pageable memory go through a bounce buffer at a
 * fraction of the link rate).  Freed by agc_hip_host_free or with the context. */
int agc_hip_host_alloc(agc_hip_ctx *ctx, uint64_t bytes, void **out);
int agc_hip_host_free(agc_hip_ctx *ctx, void *p);

/* Replaces CLZDiff_V2::Estimate (src/common/lz_diff.cpp:839-946) evaluated
 * WITHOUT a bound: h_cost[s] is the estimate with bound = ~0, h_peak[s] the
 * largest running cost seen at a loop-top check, so the caller can replay the
 * reference's early exit for any bound b: the bounded call would have returned
 * early (with a value > b) iff h_peak[s] > b. */
int agc_hip_lz_estimate_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                  const uint8_t *d_base, const uint64_t *h_off,
                                  const uint32_t *h_len, const uint8_t *h_rc,
                                  uint32_t *h_cost, uint32_t *h_peak);
int agc_hip_lz_estimate_batch_packed(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                     const agc_hip_packed *pk, const uint64_t *h_off,
                                     const uint32_t *h_len, const uint8_t *h_rc,
                                     uint32_t *h_cost, uint32_t *h_peak);
int agc_hip_lz_estimate_batch(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                              const uint8_t *h_text, const uint64_t *h_off,
                              const uint32_t *h_len, const uint8_t *h_rc,
                              uint32_t *h_cost, uint32_t *h_peak);

/* Replaces CLZDiffBase::GetCodingCostVector (src/common/lz_diff.cpp:159-284):
 * h_costs receives sum(h_len) u32 values, segment after segment. */
int agc_hip_lz_cost_vector_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                     const uint8_t *d_base, const uint64_t *h_off,
                                     const uint32_t *h_len, const uint8_t *h_rc,
                                     const uint8_t *h_prefix_costs, uint32_t *h_costs);
int agc_hip_lz_cost_vector_batch_packed(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                        const agc_hip_packed *pk, const uint64_t *h_off,
                                        const uint32_t *h_len, const uint8_t *h_rc,
                                        const uint8_t *h_prefix_costs, uint32_t *h_costs);
int agc_hip_lz_cost_vector_batch(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid,
                                 const uint8_t *h_text, const uint64_t *h_off,
                                 const uint32_t *h_len, const uint8_t *h_rc,
                                 const uint8_t *h_prefix_costs, uint32_t *h_costs);

/* Replaces the arithmetic of find_cand_segment_with_missing_middle_splitter
 * (src/core/agc_compressor.cpp:1540-1625) for a batch of segments: two cost vectors per
 * segment (GetCodingCostVector against gid1 with text orientation rc1 / prefix1, against gid2
 * with rc2 / prefix2), the first reversed when prefix1 == 0 and prefix-summed, the second
 * reversed when prefix2 != 0 and suffix-summed; h_best_pos[s] = first position minimising their
 * sum (before the caller's k+1 clamps, agc_compressor.cpp:1621-1624), h_best_sum[s] its value. */
int agc_hip_lz_split_point_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid1, const uint32_t *h_gid2,
                                     const uint8_t *d_base, const uint64_t *h_off, const uint32_t *h_len,
                                     const uint8_t *h_rc1, const uint8_t *h_prefix1,
                                     const uint8_t *h_rc2, const uint8_t *h_prefix2,
                                     uint32_t *h_best_pos, uint32_t *h_best_sum);

int agc_hip_lz_split_point_batch_packed(agc_hip_ctx *ctx, uint32_t n, const uint32_t *h_gid1, const uint32_t *h_gid2,
                                        const agc_hip_packed *pk, const uint64_t *h_off, const uint32_t *h_len,
                                        const uint8_t *h_rc1, const uint8_t *h_prefix1,
                                        const uint8_t *h_rc2, const uint8_t *h_prefix2,
                                        uint32_t *h_best_pos, uint32_t *h_best_sum);

/* Copies slices (optionally reverse-complemented) from HBM into one host buffer, back to back:
 * slice s = h_out[h_out_off[s] .. h_out_off[s+1]).  Used for the sequences the host must pack
 * itself (new group references, raw contigs; src/common/segment.cpp:14-48). */
int agc_hip_fetch_slices_dev(agc_hip_ctx *ctx, uint32_t n, const uint8_t *d_base, const uint64_t *h_off,
                             const uint32_t *h_len, const uint8_t *h_rc, uint8_t *h_out, uint64_t out_cap,
                             uint64_t *h_out_off);

int agc_hip_fetch_slices_packed(agc_hip_ctx *ctx, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc, uint8_t *h_out, uint64_t out_cap,
                                uint64_t *h_out_off);

/* ---- a13: reference storage helpers ----------------------------------- */
/* Repetitiveness probe of CSegment::store_in_archive (src/common/segment.h:224-247):
 * for lag = 4..31 the pair (matches, ACGT positions); 28 values each per slice.
 * The caller applies the double-precision 0.5 test. */
int agc_hip_ref_lag_counts_dev(agc_hip_ctx *ctx, uint32_t n, const uint8_t *d_base,
                               const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc,
                               uint32_t *h_cnt /* n*28 */, uint32_t *h_cur /* n*28 */);
int agc_hip_ref_lag_counts_packed(agc_hip_ctx *ctx, uint32_t n, const agc_hip_packed *pk,
                                  const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc,
                                  uint32_t *h_cnt /* n*28 */, uint32_t *h_cur /* n*28 */);

/* The two calls above for the new references of ONE registration, in one submission nobody has to wait for at once: slices
 * [0, n_refs) are the new references (their counters go to h_cnt / h_cur, n_refs * 28 each), slices [0, n) -- references, then raw
 * items -- are copied to h_out back to back (offsets in h_out_off, n + 1, filled at once).  _begin queues kernels and copies on a stream
 * of the context's own and returns; _end(slot) waits for that slot (two slots: a caller that hands a registration's bookkeeping to
 * another thread alternates them; _end may be called from that thread).  The host buffers must be pinned (agc_hip_host_alloc) and,
 * like the packed sample, stay untouched until _end.  Replaces the same reference lines as the two calls it combines. */
int agc_hip_ref_store_begin_packed(agc_hip_ctx *ctx, uint32_t slot, uint32_t n_refs, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc, uint32_t *h_cnt, uint32_t *h_cur, uint8_t *h_out, uint64_t out_cap,
                                   uint64_t *h_out_off);
int agc_hip_ref_store_end(agc_hip_ctx *ctx, uint32_t slot);

/* ---- S3: entropy coding of delta packs (a14) -------------------------- */
/* Replaces ZSTD_compressCCtx(cctx, dst, bound, src, n, 17) as CSegment::add_to_archive calls it for delta packs
 * (src/common/segment.h:199-201, store_in_archive(pack) :258-280), for a batch of independent inputs: frame i =
 * h_dst[h_dst_off[i] .. h_dst_off[i+1]), byte for byte what libzstd 1.4.9 writes (the library the reference is pinned
 * against here; zstd frames depend on the encoder version).  Inputs: h_src[h_src_off[i] .. h_src_off[i+1]), each at
 * most agc_hip_zstd17_max_input() bytes (one zstd block); longer ones are the caller's to hand to libzstd itself.
 * AGC_HIP_ECAP (+ the needed size in h_dst_off[n]) when dst_cap is too small. */
uint32_t agc_hip_zstd17_max_input(void);
int agc_hip_zstd17_batch(agc_hip_ctx *ctx, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off,
                         uint8_t *h_dst, uint64_t dst_cap, uint64_t *h_dst_off);
/* The same for inputs that are in HBM already (input i = d_src[h_src_off[i] .. h_src_off[i+1]); e.g. packs another GPU sent over
 * xGMI: agc_amd/dist.py): no host round trip of the inputs.  The caller makes sure whatever wrote d_src has finished. */
int agc_hip_zstd17_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint8_t *d_src, const uint64_t *h_src_off,
                             uint8_t *h_dst, uint64_t dst_cap, uint64_t *h_dst_off);
/* The same encoder for the other two levels the archive uses: 13 -- CSegment::add_to_archive_tuples (src/common/segment.h:172-197:
 * references packed 2-4 symbols per byte) -- and 19 -- store_in_archive(ref) of repetitive references (segment.h:218-255).
 * h_level[i] = 13, 17 or 19 per input (NULL: all 17); input sizes as for level 17 (one block).  Frames equal
 * ZSTD_compressCCtx(.., level) of libzstd 1.4.9 byte for byte.  Level 13 inputs above 16 KiB (btopt, minMatch 4) take the one-lane
 * kernel, everything else the lane-group kernel. */
int agc_hip_zstd_batch(agc_hip_ctx *ctx, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, const uint8_t *h_level,
                       uint8_t *h_dst, uint64_t dst_cap, uint64_t *h_dst_off);
/* on != 0: the launches of agc_hip_zstd17_batch keep their tables out of the LDS, so that kernels of the context's other
 * streams that need most of a CU's LDS (the packed splitter scan: 128 KiB per block) can start beside a launch that runs for
 * a second.  For a caller that compresses packs in the background while it goes on adding samples; off (the default) is
 * ~10 % faster when nothing else runs. */
int agc_hip_zstd17_background(agc_hip_ctx *ctx, int on);
/* How many inputs of at most 16 KiB one launch of agc_hip_zstd17_batch works on AT THE SAME TIME (a frame is a serial chain:
 * a launch lasts about as long as its longest frame, whatever their number -- up to this many; beyond it the launch takes
 * a second round).  The caller that splits packs between the device and host threads (the reference's workers call
 * ZSTD_compressCCtx one pack at a time, src/common/segment.h:199-201) keeps the device's share below it.  0 = no device. */
uint32_t agc_hip_zstd17_resident_frames(agc_hip_ctx *ctx);
/* The compression parameters libzstd 1.4.9 derives for level 17 and a known source size (ZSTD_getCParams(17, n, 0)):
 * windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy.  Exposed so that tests can pin them. */
int agc_hip_zstd17_cparams(uint64_t src_size, uint32_t out7[7]);
int agc_hip_zstd_cparams(int level /* 13, 17, 19 */, uint64_t src_size, uint32_t out7[7]);

#ifdef __cplusplus
}
#endif
#endif /* AGC_HIP_H */
