// capi_host.cpp -- plain C entry points of the host-side compressor (for bench.py / tests via ctypes).
#include "compressor.h"
#include <cstring>
#include <string>
#include <vector>

using agc::CAGCCompressor;

extern "C" {

void *agc_cmp_new(int device)
{
    auto *c = new CAGCCompressor();
    c->SetDevice(device);
    return c;
}

void agc_cmp_delete(void *h) { delete (CAGCCompressor *)h; }

// CAGCCompressor::Create (src/core/agc_compressor.h:754-756): 1 = ok, 0 = failure (message on stderr)
int agc_cmp_create(void *h, const char *out_path, uint32_t pack_cardinality, uint32_t k, const char *ref_file, uint32_t segment_size,
                   uint32_t min_match_len, int concatenated, int adaptive, uint32_t verbosity, uint32_t n_threads, double fallback_frac)
{
    return ((CAGCCompressor *)h)->Create(out_path ? out_path : "", pack_cardinality, k, ref_file ? ref_file : "", segment_size, min_match_len,
                                         concatenated != 0, adaptive != 0, verbosity, n_threads, fallback_frac) ? 1 : 0;
}

int agc_cmp_set_splitters(void *h, const uint64_t *kmers, uint64_t n) { return ((CAGCCompressor *)h)->SetSplitters(kmers, n) ? 1 : 0; }

int agc_cmp_set_reference_dev(void *h, const uint8_t *d_codes, const uint64_t *ctg_off, uint32_t n_ctg)
{
    return ((CAGCCompressor *)h)->SetReferenceDevice(d_codes, ctg_off, n_ctg) ? 1 : 0;
}

int agc_cmp_add_sample_files(void *h, uint32_t n, const char **sample_names, const char **paths, uint32_t n_threads)
{
    std::vector<std::pair<std::string, std::string>> v;
    for (uint32_t i = 0; i < n; ++i)
        v.emplace_back(sample_names[i], paths[i]);
    return ((CAGCCompressor *)h)->AddSampleFiles(v, n_threads) ? 1 : 0;
}

int agc_cmp_add_sample_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const uint8_t *d_codes,
                           const uint64_t *ctg_off)
{
    std::vector<std::string> names;
    for (uint32_t i = 0; i < n_ctg; ++i)
        names.emplace_back(contig_names[i]);
    return ((CAGCCompressor *)h)->AddSampleDevice(sample_name, names, d_codes, ctg_off) ? 1 : 0;
}

// multi-GPU single-archive mode (compressor.h: SetDistributed / LastRecord / ApplyRecord)
int agc_cmp_set_distributed(void *h, uint32_t rank, uint32_t world_size, uint32_t writer_rank)
{
    return ((CAGCCompressor *)h)->SetDistributed(rank, world_size, writer_rank) ? 1 : 0;
}
int agc_cmp_last_record(void *h, const uint8_t **ptr, uint64_t *n)
{
    size_t m = 0;
    *ptr = ((CAGCCompressor *)h)->LastRecord(&m);
    *n = m;
    return 1;
}
int agc_cmp_last_record_framed(void *h, uint8_t **ptr, uint64_t *n)
{
    size_t m = 0;
    *ptr = ((CAGCCompressor *)h)->LastRecordFramed(&m);
    *n = m;
    return 1;
}
int agc_cmp_apply_record(void *h, const uint8_t *record, uint64_t n, const uint8_t *d_record, const uint8_t *body, uint64_t body_n)
{
    return ((CAGCCompressor *)h)->ApplyRecord(record, n, d_record, body, body_n) ? 1 : 0;
}
int agc_cmp_last_record_body(void *h, const uint8_t **ptr, uint64_t *n)
{
    size_t m = 0;
    *ptr = ((CAGCCompressor *)h)->LastRecordBody(&m);
    *n = m;
    return 1;
}
int agc_cmp_record_body_buffer(void *h, uint64_t n, uint8_t **ptr)
{
    *ptr = ((CAGCCompressor *)h)->RecordBodyBuffer(n);
    return *ptr ? 1 : 0;
}
int agc_cmp_append(void *h, const char *in_archive, const char *out_archive, uint32_t verbosity, int concatenated, int adaptive, uint32_t n_threads)
{
    return ((CAGCCompressor *)h)->Append(in_archive, out_archive, verbosity, true, concatenated != 0, adaptive != 0, n_threads, 0.0) ? 1 : 0;
}

int agc_cmp_prepare_sample_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const uint8_t *d_codes,
                               const uint64_t *ctg_off)
{
    std::vector<std::string> names;
    for (uint32_t i = 0; i < n_ctg; ++i)
        names.emplace_back(contig_names[i]);
    return ((CAGCCompressor *)h)->PrepareSampleDevice(sample_name, names, d_codes, ctg_off) ? 1 : 0;
}
int agc_cmp_prepare_sample_packed_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const void *packed,
                                      const uint64_t *ctg_off)
{
    std::vector<std::string> names;
    for (uint32_t i = 0; i < n_ctg; ++i)
        names.emplace_back(contig_names[i]);
    return ((CAGCCompressor *)h)->PrepareSamplePackedDevice(sample_name, names, packed, ctg_off) ? 1 : 0;
}
int agc_cmp_add_sample_packed_dev(void *h, const char *sample_name, uint32_t n_ctg, const char **contig_names, const void *packed,
                                  const uint64_t *ctg_off)
{
    std::vector<std::string> names;
    for (uint32_t i = 0; i < n_ctg; ++i)
        names.emplace_back(contig_names[i]);
    return ((CAGCCompressor *)h)->AddSamplePackedDevice(sample_name, names, packed, ctg_off) ? 1 : 0;
}
int agc_cmp_set_next_sample_packed_dev(void *h, const void *packed, const uint64_t *ctg_off, uint32_t n_ctg)
{
    return ((CAGCCompressor *)h)->SetNextSamplePackedDevice(packed, ctg_off, n_ctg) ? 1 : 0;
}
int agc_cmp_set_next_fasta_dev(void *h, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *raw_begin, const uint64_t *raw_end, uint32_t n_ctg,
                               uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks)
{
    return ((CAGCCompressor *)h)->SetNextFastaDevice(d_raw, n_raw, raw_begin, raw_end, n_ctg, d_words, d_esc_index, d_esc_bytes, esc_cap_blocks) ? 1 : 0;
}
int agc_cmp_finish_fasta_dev(void *h, uint64_t *ctg_off, uint64_t *n_esc_blocks) { return ((CAGCCompressor *)h)->FinishFastaDevice(ctg_off, n_esc_blocks); }
int agc_cmp_commit_prepared(void *h) { return ((CAGCCompressor *)h)->CommitPrepared() ? 1 : 0; }
int agc_cmp_commit_prepared_head(void *h) { return ((CAGCCompressor *)h)->CommitPreparedHead() ? 1 : 0; }
int agc_cmp_commit_prepared_finish(void *h) { return ((CAGCCompressor *)h)->CommitPreparedFinish() ? 1 : 0; }

int agc_cmp_close_collect_packs(void *h, const uint8_t **src, const uint64_t **off, uint32_t *n)
{
    return ((CAGCCompressor *)h)->CloseCollectPacks(src, off, n) ? 1 : 0;
}
uint64_t agc_cmp_deferred_pack_bytes(void *h) { return ((CAGCCompressor *)h)->DeferredPackBytes(); }
int agc_cmp_deal_collect_packs(void *h, uint32_t *deal_id, const uint8_t **src, const uint64_t **off, uint32_t *n)
{
    return ((CAGCCompressor *)h)->DealCollectPacks(deal_id, src, off, n) ? 1 : 0;
}
int agc_cmp_deal_keep_own(void *h, uint32_t deal_id, uint32_t first, uint32_t count) { return ((CAGCCompressor *)h)->DealKeepOwn(deal_id, first, count) ? 1 : 0; }
int agc_cmp_deal_provide_frames(void *h, uint32_t deal_id, uint32_t first, uint32_t count, const uint8_t *frames, const uint64_t *off)
{
    return ((CAGCCompressor *)h)->DealProvideFrames(deal_id, first, count, frames, off) ? 1 : 0;
}
int agc_cmp_close_provide_frames(void *h, const uint8_t *frames, const uint64_t *off)
{
    return ((CAGCCompressor *)h)->CloseProvideFrames(frames, off) ? 1 : 0;
}
int agc_cmp_drain(void *h) { return ((CAGCCompressor *)h)->Drain() ? 1 : 0; }
int agc_cmp_close(void *h, uint32_t n_threads) { return ((CAGCCompressor *)h)->Close(n_threads) ? 1 : 0; }

const char *agc_cmp_zstd_version(void *h) { return ((CAGCCompressor *)h)->ZstdVersion(); }

void *agc_cmp_hip_ctx(void *h) { return ((CAGCCompressor *)h)->HipContext(); }

// 11 counters + 8 stage times, in the order of agc::CompressorStats
int agc_cmp_stats(void *h, double *out, uint32_t n)
{
    const agc::CompressorStats &s = ((CAGCCompressor *)h)->Stats();
    const double v[] = {(double)s.bases, (double)s.segments, (double)s.new_groups, (double)s.one_splitter, (double)s.middle_tried,
                        (double)s.middle_split, (double)s.lz_encoded, (double)s.delta_bytes, (double)s.ref_bytes, (double)s.zstd_in,
                        (double)s.zstd_out, (double)s.archive_bytes, s.t_scan, s.t_classify, s.t_gpu_aux, s.t_register, s.t_encode,
                        s.t_store, s.t_zstd, s.t_io, s.t_device, s.h_scan, s.h_classify, s.h_gpu_aux, s.h_register, s.h_encode, s.h_store,
                        (double)s.windows, (double)s.commit_runs, (double)s.revalidated,
                        (double)s.enc_text, (double)s.enc_ref, (double)s.est_text, (double)s.est_ref, (double)s.cv_text, (double)s.cv_ref, (double)s.zstd_dev_in, s.t_zstd_dev, s.t_zstd_host, s.t_zstd_stage, s.t_zstd_wait, (double)s.reprepared, (double)s.zstd_dev_out, (double)s.windows_cut};
    const uint32_t m = sizeof(v) / sizeof(v[0]);
    for (uint32_t i = 0; i < n && i < m; ++i)
        out[i] = v[i];
    return (int)m;
}

} // extern "C"
