// compressor.h -- host side of `agc create` above the HIP C ABI (include/agc_hip.h).
//
// Mirrors the reference's operator interface for this path,
//   CAGCCompressor::Create / AddSampleFiles / Close        src/core/agc_compressor.h:754-763
// (same argument meaning, bool results, messages on stderr, no exceptions across the API),
// plus AddSampleDevice for callers whose contigs already live in HBM (bench.py).
//
// All symbol-level work (scan, index, LZ encode / estimate / cost vectors, reverse
// complement, repetitiveness counters) runs in the HIP kernels; this file keeps the
// reference's ORDERING CONTRACT -- which group a segment goes to, group ids, in-group ids,
// pack boundaries, stream/part order -- and feeds libzstd on the host cores.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

struct agc_hip_ctx;

namespace agc {

struct CompressorStats {
    uint64_t bases = 0;            // symbols after preprocess_raw_contig, all samples incl. the reference
    uint64_t segments = 0;         // placed segments
    uint64_t new_groups = 0;
    uint64_t one_splitter = 0;     // segments resolved by find_cand_segment_with_one_splitter
    uint64_t middle_tried = 0;     // missing-middle searches
    uint64_t middle_split = 0;     // ... that cut a segment in two
    uint64_t lz_encoded = 0;       // segments LZ-encoded on the GPU
    uint64_t delta_bytes = 0;
    uint64_t ref_bytes = 0;        // symbols stored as group references
    uint64_t zstd_in = 0, zstd_out = 0;
    uint64_t archive_bytes = 0;
    // symbols the LZ kernels were asked to look at (algorithmic bytes of SURVEY 8d: every text once + its reference once)
    double t_zstd_dev = 0, t_zstd_host = 0, t_zstd_stage = 0; // entropy stage: device call, host pool, staging copies (all inside t_zstd)
    double t_zstd_wait = 0;        // time the caller stood still for the entropy stage (t_zstd runs beside the steps; this part was not hidden)
    uint64_t zstd_dev_in = 0;      // bytes entropy-coded on the GPU (part of zstd_in)
    uint64_t zstd_dev_out = 0;     // bytes of the frames it wrote (counted, not estimated)
    uint64_t enc_text = 0, enc_ref = 0, est_text = 0, est_ref = 0, cv_text = 0, cv_ref = 0;
    uint64_t windows_cut = 0; // adaptive mode: windows cut at a registration that had to extend the splitter set
    uint64_t windows = 0, commit_runs = 0, revalidated = 0; // process_batch calls, commit runs inside them, segments classified again
    uint64_t reprepared = 0;       // multi-GPU + adaptive mode: samples whose prepare ahead of the turn did not stand (prepared again at the turn)
    double t_scan = 0, t_classify = 0, t_gpu_aux = 0, t_register = 0, t_encode = 0, t_store = 0, t_zstd = 0, t_io = 0;
    double t_device = 0;           // part of the stage times above spent inside the device library (kernels, copies, syncs)
    double h_scan = 0, h_classify = 0, h_gpu_aux = 0, h_register = 0, h_encode = 0, h_store = 0; // host-only part of each stage
};

class CAGCCompressor {
    struct Impl;
    std::unique_ptr<Impl> p;

public:
    CAGCCompressor();
    ~CAGCCompressor();

    // device: HIP device ordinal (one process per GPU).  Must be called before Create.
    bool SetDevice(int device);

    // src/core/agc_compressor.cpp:2273-2327.  file_name "" = discard the archive bytes (bench),
    // "-" = stdout.  reference_file_name "" = splitters are supplied with SetSplitters.
    bool Create(const std::string &file_name, uint32_t pack_cardinality, uint32_t kmer_length, const std::string &reference_file_name,
                uint32_t segment_size, uint32_t min_match_len, bool concatenated_genomes, bool adaptive_compression,
                uint32_t verbosity, uint32_t no_threads, double fallback_frac);

    // src/core/agc_compressor.cpp:2330-2374: opens in_archive_name, copies what can be copied to out_archive_name and
    // restores splitters, groups and the collection so that AddSampleFiles / Close extend the archive
    bool Append(const std::string &in_archive_name, const std::string &out_archive_name, uint32_t verbosity, bool prefetch_archive,
                bool concatenated_genomes, bool adaptive_compression, uint32_t no_threads, double fallback_frac);

    // replaces determine_splitters' result (agc_compressor.cpp:543-555) when no reference file is given
    bool SetSplitters(const uint64_t *kmers, uint64_t n);
    // the same from a reference genome resident in HBM: runs determine_splitters on the GPU
    bool SetReferenceDevice(const uint8_t *d_codes, const uint64_t *ctg_off, uint32_t n_ctg);

    // src/core/agc_compressor.cpp:2118-2270
    bool AddSampleFiles(const std::vector<std::pair<std::string, std::string>> &sample_file_names, uint32_t no_threads);

    // one sample whose contigs (symbol codes, one byte each) are resident in HBM:
    // contig c = d_codes[ctg_off[c] .. ctg_off[c+1])
    bool AddSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                         const uint64_t *ctg_off);

    // ---- multi-GPU single-archive mode (protocol in compressor_dist.cpp) ----
    // Call before Create on every rank; only the writer rank should be given a real archive name.  Afterwards every sample
    // must reach every rank, in the same order: AddSampleDevice on its owner (then LastRecord is what the other ranks
    // need), ApplyRecord everywhere else.  d_record = optional copy of the record in this rank's HBM (e.g. the buffer an
    // RCCL broadcast delivered): the newly minted references are then registered from there without a host round trip.
    bool SetDistributed(uint32_t rank, uint32_t world_size, uint32_t writer_rank);
    const uint8_t *LastRecord(size_t *n) const;         // the head: every rank (pinned host memory, valid until this rank's next commit)
    // the same with 64 bytes in front of it that belong to the caller's transport (a fixed-size message header: head and header
    // then travel as one message); *n counts them
    uint8_t *LastRecordFramed(size_t *n);
    const uint8_t *LastRecordBody(size_t *n) const;     // the LZ deltas: the writer rank only (pinned host memory)
    // writer rank: where the next record's body should be received (pinned host memory, n bytes); an ApplyRecord whose `body` is
    // this pointer takes the buffer over instead of copying it (any other pointer is copied)
    uint8_t *RecordBodyBuffer(size_t n);
    bool ApplyRecord(const uint8_t *record, size_t n, const uint8_t *d_record, const uint8_t *body = nullptr, size_t body_n = 0);

    // AddSampleDevice in two halves: everything that only reads the classification state (scan, classification, LZ encode of
    // the segments whose group is known), and the order-dependent commit.  Between the two other samples may be committed
    // (ApplyRecord); the commit revalidates exactly the decisions that read what changed.
    bool PrepareSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                             const uint64_t *ctg_off);
    bool CommitPrepared();
    // CommitPrepared in two steps (multi-GPU mode): after the first, LastRecord (the head) is ready and every other rank can go
    // on; the second indexes the new references on this GPU, encodes what is left and builds LastRecordBody.
    bool CommitPreparedHead();
    bool CommitPreparedFinish();
    // the same for a sample resident in HBM in the 2-bit layout (include/agc_hip.h: agc_hip_packed; contig c = symbols
    // [ctg_off[c], ctg_off[c+1]) of the packed buffer): scan, LZ estimates / cost vectors / encode and the store of new references
    // all read the packed words where they lie -- no byte copy of the sample is made.  The caller leaves the packed buffers
    // untouched until the NEXT sample call, Drain or Close has returned (the sample's LZ encode may still be in flight on the
    // device's second lane when the call returns; the bookkeeping thread collects it)
    bool PrepareSamplePackedDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const void *packed,
                                   const uint64_t *ctg_off);
    bool AddSamplePackedDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const void *packed,
                               const uint64_t *ctg_off);
    // The caller that knows its NEXT packed sample says so before it adds the current one: the next sample's splitter
    // scan is queued on the device (include/agc_hip.h: agc_hip_prefetch_packed_dev) as soon as the current sample's classification
    // kernels are in, and runs beside its classification / encode / registration -- the reference's workers likewise take contigs of later samples
    // from the queue while earlier ones register (agc_compressor.cpp:1093-1272).  The announced sample must then be the next one
    // added (otherwise the work is dropped); not used in adaptive mode (new splitters change later scans).
    bool SetNextSamplePackedDevice(const void *packed, const uint64_t *ctg_off, uint32_t n_ctg);
    // A sample that is still the bytes of its FASTA file in HBM (contig c = raw bytes [raw_begin[c], raw_end[c]): its sequence
    // lines), to become a packed sample in the caller's buffers (include/agc_hip.h: agc_hip_pack_fasta_begin -- the reference's
    // preprocess_raw_contig, agc_compressor.cpp:907-951, fused with the 2-bit packing).  The conversion is QUEUED BY THE COMPRESSOR at
    // the point of the sample in progress where the GPU has room for it (with the announced scan: after the classification
    // kernels, beside the registration's host work) -- the reference's reader thread likewise runs ahead of its workers
    // (agc_compressor.cpp:2155-2238).  FinishFastaDevice waits for it (and queues it first when no sample call came in between)
    // and returns the contigs' symbol offsets (n_ctg + 1) and the number of escaped blocks; its result is agc_hip_pack_fasta_end's
    // (AGC_HIP_OK, AGC_HIP_ECAP: announce again with an escape buffer of *n_esc_blocks blocks, ...).  One conversion at a time.
    bool SetNextFastaDevice(const uint8_t *d_raw, uint64_t n_raw, const uint64_t *raw_begin, const uint64_t *raw_end, uint32_t n_ctg, uint32_t *d_words,
                            int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks);
    int FinishFastaDevice(uint64_t *ctg_off, uint64_t *n_esc_blocks);

    // src/core/agc_compressor.cpp:2094-2115 (close_compression) + ~CArchive
    bool Close(uint32_t no_threads);
    // Close in steps, for an entropy stage spread over several GPUs (agc_amd/dist.py): CloseCollectPacks builds the pending pack
    // jobs and hands out the inputs of those a device may compress (back to back, pack i = src[off[i] .. off[i+1])); the caller
    // has them compressed (agc_hip_zstd17_batch on any GPU) and returns the frames in the same order; Close then finishes with
    // them.  Without these two calls Close compresses everything itself.
    bool CloseCollectPacks(const uint8_t **src, const uint64_t **off, uint32_t *n);
    // The same IN THE MIDDLE OF A RUN, for the delta packs that have filled (the reference's workers code a pack the moment it is
    // full while the others go on, segment.cpp:34-80): the writer rank of an N-rank job parks them (their parts hold their places
    // in the archive already); DeferredPackBytes says how much has piled up, DealCollectPacks hands them out as one DEAL -- inputs
    // back to back, valid until every pack of the deal is settled --, the caller sends every rank its share, and the shares settle
    // independently, in any order, samples later: DealProvideFrames(deal, first, count, frames, off) for packs [first, first + count)
    // coded elsewhere (frame t = frames[off[t] .. off[t+1])), DealKeepOwn for a share this rank's own entropy stage takes.
    uint64_t DeferredPackBytes();
    bool DealCollectPacks(uint32_t *deal_id, const uint8_t **src, const uint64_t **off, uint32_t *n);
    bool DealKeepOwn(uint32_t deal_id, uint32_t first, uint32_t count);
    bool DealProvideFrames(uint32_t deal_id, uint32_t first, uint32_t count, const uint8_t *frames, const uint64_t *off);
    bool Drain(); // waits for the asynchronous entropy stage (every part handed over so far compressed and written)
    bool CloseProvideFrames(const uint8_t *frames, const uint64_t *off);

    const CompressorStats &Stats() const;
    const char *ZstdVersion() const;
    agc_hip_ctx *HipContext();
};

// AGC_AMD_START_LAPS=1 (a measuring aid): milliseconds since the first call, on stderr, at the named points of a run's start
void StartLap(const char *what);

} // namespace agc
