// compressor.cpp -- see compressor.h: the public methods (create / append / add samples / close).
// Citations: file:line under the reference tree.
#include "compressor_impl.h"

namespace agc {

CAGCCompressor::CAGCCompressor() : p(new Impl) {}
CAGCCompressor::~CAGCCompressor()
{
    p->book_shutdown(); // the bookkeeping and the entropy thread use the device context
    p->z_shutdown();
    p->enc_buf.release(); // (pinned memory of that context)
    p->dev_seg_buf.release();
    p->enc_buf2.release();
    p->enc_alt.release();
    p->enc_alt2.release();
    p->dist_body_buf.release();
    p->body_recv.reset();
    p->body_pool.clear();
    if (p->hip)
        agc_hip_destroy(p->hip);
}

bool CAGCCompressor::SetDevice(int device)
{
    if (p->hip)
        return false;
    p->device = device;
    return true;
}

const CompressorStats &CAGCCompressor::Stats() const { return p->st; }
const char *CAGCCompressor::ZstdVersion() const { return p->zstd.h ? p->zstd.versionString() : ""; }
agc_hip_ctx *CAGCCompressor::HipContext() { return p->hip; }

bool CAGCCompressor::SetDistributed(uint32_t rank, uint32_t world_size, uint32_t writer_rank)
{
    if (p->created || !world_size || rank >= world_size || writer_rank >= world_size)
        return false;
    p->dist_rank = rank;
    p->dist_world = world_size;
    p->dist_writer = writer_rank;
    return true;
}
const uint8_t *CAGCCompressor::LastRecord(size_t *n) const
{
    *n = p->dist_record_n;
    return p->dist_record_n ? p->dist_record_ptr() : nullptr;
}
uint8_t *CAGCCompressor::LastRecordFramed(size_t *n)
{
    *n = p->dist_record_n ? p->dist_record_n + Impl::DIST_FRAME : 0;
    return p->dist_record_n ? p->dist_record_buf.data() : nullptr;
}
const uint8_t *CAGCCompressor::LastRecordBody(size_t *n) const
{
    *n = p->dist_body_n;
    return p->dist_body_buf.data();
}
uint8_t *CAGCCompressor::RecordBodyBuffer(size_t n)
{
    Impl &I = *p;
    if (!I.created || I.dist_world < 2)
        return nullptr;
    if (!I.body_recv) {
        std::lock_guard<std::mutex> lk(I.body_pool_mtx);
        if (!I.body_pool.empty()) {
            I.body_recv = std::move(I.body_pool.back());
            I.body_pool.pop_back();
        }
    }
    if (!I.body_recv) {
        I.body_recv.reset(new PinnedBytes());
        I.body_recv->ctx = I.hip;
    }
    if (!I.body_recv->resize(n + n / 8 + 64, false)) // (headroom: the next samples' bodies are about as long)
        return nullptr;
    return I.body_recv->data();
}
bool CAGCCompressor::ApplyRecord(const uint8_t *record, size_t n, const uint8_t *d_record, const uint8_t *body, size_t body_n)
{
    if (!p->created || p->dist_world < 2)
        return false;
    return p->apply_record(record, n, d_record, body, body_n);
}

// determine_splitters for a reference genome that already lives in HBM (bench.py)
bool CAGCCompressor::SetReferenceDevice(const uint8_t *d_codes, const uint64_t *ctg_off, uint32_t n_ctg)
{
    Impl &I = *p;
    if (!I.created || I.adaptive)
        return false;
    const uint64_t tot = n_ctg ? ctg_off[n_ctg] - ctg_off[0] : 0;
    std::vector<uint64_t> spl(std::max<uint64_t>(1024, tot / std::max(1u, I.segment_size) * 2 + 2ull * n_ctg + 16));
    uint64_t n_spl = 0;
    for (;;) {
        int rc = agc_hip_determine_splitters_dev(I.hip, d_codes, ctg_off, n_ctg, I.k, I.segment_size, spl.size(), spl.data(), &n_spl, 0, nullptr, nullptr);
        if (rc == AGC_HIP_ECAP && n_spl > spl.size()) {
            spl.resize(n_spl);
            continue;
        }
        if (!I.hip_ok(rc, "determine_splitters"))
            return false;
        break;
    }
    return SetSplitters(spl.data(), n_spl);
}

void StartLap(const char *what) { start_lap(what); }

bool CAGCCompressor::SetSplitters(const uint64_t *kmers, uint64_t n)
{
    if (!p->created)
        return false;
    p->splitters.assign(kmers, kmers + n);
    std::sort(p->splitters.begin(), p->splitters.end());
    p->splitters.erase(std::unique(p->splitters.begin(), p->splitters.end()), p->splitters.end());
    return p->hip_ok(DEVTP(agc_hip_splitters_set(p->hip, p->splitters.data(), p->splitters.size())), "splitters_set");
}

bool CAGCCompressor::Create(const std::string &file_name, uint32_t pack_cardinality, uint32_t kmer_length, const std::string &reference_file_name,
                            uint32_t segment_size, uint32_t min_match_len, bool concatenated_genomes, bool adaptive_compression,
                            uint32_t verbosity, uint32_t no_threads, double fallback_frac)
{
    Impl &I = *p;
    if (I.created)
        return false;
    if (adaptive_compression && reference_file_name.empty()) {
        I.err("adaptive mode (-a) needs the reference file (its singleton k-mers are kept)");
        return false;
    }
    if (fallback_frac != 0.0) {
        I.err("fallback minimizers (-f) are not implemented");
        return false;
    }
    I.pack_cardinality = pack_cardinality;
    I.k = kmer_length;
    I.segment_size = segment_size;
    I.mml = min_match_len;
    I.concatenated = concatenated_genomes;
    I.adaptive = adaptive_compression;
    I.verbosity = verbosity;

    std::string e;
    start_lap("Create");
    if (!I.zstd.load(e)) {
        I.err(e);
        return false;
    }
    start_lap("libzstd opened");
    if (!I.hip) {
        int rc = agc_hip_create(&I.hip, I.device);
        if (rc != AGC_HIP_OK) {
            I.err("no HIP device: the MI355X path has no CPU fallback (agc_hip_create = " + std::to_string(rc) + ")");
            return false;
        }
    }
    start_lap("device context (runtime, streams, events)");
    unsigned nt = std::max(1u, no_threads);
    I.pool.reset(new ThreadPool(nt));
    I.zpool.reset(new ThreadPool(nt, 10));
    I.bpool.reset(new ThreadPool(std::max(1u, std::min(8u, nt / 2))));
    if (const char *e = getenv("AGC_AMD_ASYNC_BOOK"))
        I.async_book = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_ASYNC_ENCODE"))
        I.async_encode = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_EARLY_COLLECT"))
        I.early_collect = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_REF_STORE_ASYNC"))
        I.ref_store_async = atoi(e) != 0;
    I.enc_alt.ctx = I.enc_alt2.ctx = I.hip;
    for (unsigned i = 0; i < nt; ++i)
        I.zctx.emplace_back(new ZstdCtx(&I.zstd));
    I.sync_entropy = getenv("AGC_AMD_SYNC_ENTROPY") != nullptr;
    if (const char *e = getenv("AGC_AMD_ENTROPY_STREAM"))
        I.entropy_stream = atoi(e) != 0;
    I.enc_buf.ctx = I.enc_buf2.ctx = I.dist_body_buf.ctx = I.dist_record_buf.ctx = I.hip;
    if (const char *e = getenv("AGC_AMD_PAR_MIN"))
        I.par_min = (size_t)std::max(1LL, atoll(e));
    if (const char *e = getenv("AGC_AMD_ENCODE_OVERLAP"))
        I.overlap_mode = !strcmp(e, "early") || !strcmp(e, "1") ? 1 : !strcmp(e, "late") || !strcmp(e, "2") ? 2 : 0;
    I.choose_entropy_stage();
    if (const char *e = getenv("AGC_AMD_DEV_SEGMENTS"))
        I.dev_segments = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_DEV_ENCODE_MIN"))
        I.dev_encode_min = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("AGC_AMD_PRE_LAUNCH_ENCODE"))
        I.pre_launch_encode = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_PLACE_AHEAD"))
        I.place_ahead = atoi(e);
    if (const char *e = getenv("AGC_AMD_SPEC_FILL_AHEAD"))
        I.spec_fill_ahead = atoi(e);
    if (!PkMap::hash_agrees()) {
        I.err("internal: the host's and the device library's group hash differ");
        return false;
    }
    I.created = true;
    start_lap("pools, zstd contexts");

    if (!reference_file_name.empty()) {
        // the reference file: big plain files through the mapped reader; contigs of 1 MiB and more are converted to symbol
        // codes on the GPU on their way in (agc_hip_preprocess = preprocess_raw_contig, agc_compressor.cpp:907-951), like the
        // samples' (AddSampleFiles)
        std::vector<bytes_t> ref;
        {
            std::vector<std::string> ids;
            static const uint64_t map_min = getenv("AGC_AMD_MAP_MIN") ? strtoull(getenv("AGC_AMD_MAP_MIN"), nullptr, 10) : (64ull << 20);
            if (!FastaReader::read_all_mapped(reference_file_name, ids, ref, std::max(1u, std::min(8u, nt)), map_min)) {
                FastaReader fr;
                if (!fr.open(reference_file_name)) {
                    I.err("Cannot open file: " + reference_file_name);
                    I.created = false;
                    return false;
                }
                std::string id;
                bytes_t c;
                while (fr.read_contig_raw(id, c)) {
                    ref.emplace_back(std::move(c));
                    c.clear();
                }
            }
        }
        uint64_t tot_raw = 0;
        for (auto &c : ref)
            tot_raw += c.size();
        // determine_splitters on the GPU: contigs go to HBM back to back, k-mers are enumerated, radix
        // sorted and reduced to singletons there (include/agc_hip.h: agc_hip_determine_splitters_dev)
        uint8_t *d_ref = nullptr;
        if (!I.hip_ok(DEVTI(agc_hip_sample_buffer(I.hip, tot_raw, &d_ref)), "sample_buffer"))
            return false;
        std::vector<uint64_t> off(ref.size() + 1, 0);
        for (size_t i = 0; i < ref.size(); ++i) {
            uint64_t n_codes = 0;
            if (ref[i].size() >= (1ull << 20)) {
                if (!I.hip_ok(DEVTI(agc_hip_preprocess(I.hip, ref[i].data(), ref[i].size(), d_ref + off[i], &n_codes)), "preprocess"))
                    return false;
            } else {
                preprocess_raw_contig(ref[i]);
                n_codes = ref[i].size();
                if (!I.hip_ok(DEVTI(agc_hip_copy_to_device(I.hip, d_ref + off[i], ref[i].data(), ref[i].size())), "copy_to_device"))
                    return false;
            }
            off[i + 1] = off[i] + n_codes;
            bytes_t().swap(ref[i]); // (the host copy is not needed any more)
        }
        const uint64_t tot = off[ref.size()];
        std::vector<uint64_t> spl(std::max<uint64_t>(1024, tot / std::max(1u, I.segment_size) * 2 + 2 * ref.size() + 16));
        std::vector<uint64_t> sorted_kmers(I.adaptive ? tot : 0);
        uint64_t n_spl = 0, n_sorted = 0;
        for (;;) {
            int rc = agc_hip_determine_splitters_dev(I.hip, d_ref, off.data(), (uint32_t)ref.size(), I.k, I.segment_size, spl.size(), spl.data(),
                                                     &n_spl, sorted_kmers.size(), I.adaptive ? sorted_kmers.data() : nullptr,
                                                     I.adaptive ? &n_sorted : nullptr);
            if (rc == AGC_HIP_ECAP && n_spl > spl.size()) {
                spl.resize(n_spl);
                continue;
            }
            if (!I.hip_ok(rc, "determine_splitters"))
                return false;
            break;
        }
        spl.resize(n_spl);
        if (I.adaptive) {
            // v_candidate_kmers (singletons) and v_duplicated_kmers (agc_compressor.cpp:493-497)
            sorted_kmers.resize(n_sorted);
            split_singletons(sorted_kmers, &I.ref_duplicates);
            I.ref_singletons.swap(sorted_kmers);
        }
        if (!SetSplitters(spl.data(), spl.size()))
            return false;
        if (I.verbosity > 1)
            std::cerr << "No. of splitters: " << spl.size() << std::endl;
    }

    if (!I.ar.open(file_name)) {
        I.err("Cannot create archive " + file_name);
        I.created = false;
        return false;
    }
    I.coll.set_archive(&I.ar, &I.zstd, I.segment_size, I.k);
    I.map_segments[{NO_KMER, NO_KMER}] = 0;
    I.groups.resize(NO_RAW_GROUPS);
    for (I.no_segments = 0; I.no_segments < NO_RAW_GROUPS; ++I.no_segments) {
        Group &g = I.groups[I.no_segments];
        g.exists = true;
        g.stream_delta = I.ar.register_stream(ss_delta_name(I.no_segments));
        g.no_seqs = 1;                  // add_raw({0x7f}), agc_compressor.cpp:2313-2321
        const uint8_t dummy = 0x7f;
        Group::push(g.raw_data, g.raw_off, &dummy, 1);
    }
    I.coll.reset_prev_sample_name();
    return true;
}

// ---------------------------------------------------------------------------
// Append (agc_compressor.cpp:2330-2374): load_file_type_info + load_metadata (agc_basic.cpp:53-236),
// CCollection_V3::prepare_for_appending_copy / _load_last_batch (collection_v3.cpp:47-108), appending_init (:303-380)
bool CAGCCompressor::Append(const std::string &in_archive_name, const std::string &out_archive_name, uint32_t verbosity, bool prefetch_archive,
                            bool concatenated_genomes, bool adaptive_compression, uint32_t no_threads, double fallback_frac)
{
    Impl &I = *p;
    (void)prefetch_archive;
    if (I.created)
        return false;
    if (fallback_frac != 0.0) {
        I.err("fallback minimizers (-f) are not implemented");
        return false;
    }
    std::string e;
    if (!I.zstd.load(e)) {
        I.err(e);
        return false;
    }
    if (!I.zd.load()) {
        I.err("cannot dlopen libzstd (set AGC_ZSTD_LIB)");
        return false;
    }
    if (!I.in_ar.open(in_archive_name, false)) {
        I.err("Cannot open archive " + in_archive_name);
        return false;
    }
    const uint8_t *ptr;
    uint64_t size, meta;
    if (!I.in_ar.get_part("file_type_info", 0, ptr, size, meta))
        return false;
    {
        const uint8_t *q = ptr, *qe = ptr + size;
        std::string key, val;
        I.in_file_type_info.clear();
        for (uint64_t i = 0; i < meta && rd::rd_str(q, qe, key) && rd::rd_str(q, qe, val); ++i)
            I.in_file_type_info[key] = val;
        if (I.in_file_type_info["file_version_major"] != "3") {
            I.err("Unsupported archive version (only the v3 format is handled)");
            return false;
        }
    }
    if (!I.in_ar.get_part("params", 0, ptr, size, meta) || size < 16) {
        I.err("Archive does not contain parameters section");
        return false;
    }
    auto le32 = [&](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
    auto le64 = [&](const uint8_t *b) { return (uint64_t)le32(b) | ((uint64_t)le32(b + 4) << 32); };
    I.k = le32(ptr);
    I.mml = le32(ptr + 4);
    I.pack_cardinality = le32(ptr + 8);
    I.segment_size = le32(ptr + 12);
    if (!I.pack_cardinality)
        return false;
    I.concatenated = concatenated_genomes;
    I.adaptive = adaptive_compression;
    I.verbosity = verbosity;
    I.appending = true;

    if (!I.hip) {
        int rc = agc_hip_create(&I.hip, I.device);
        if (rc != AGC_HIP_OK) {
            I.err("no HIP device: the MI355X path has no CPU fallback (agc_hip_create = " + std::to_string(rc) + ")");
            return false;
        }
    }
    unsigned nt = std::max(1u, no_threads);
    I.pool.reset(new ThreadPool(nt));
    I.zpool.reset(new ThreadPool(nt, 10));
    I.bpool.reset(new ThreadPool(std::max(1u, std::min(8u, nt / 2))));
    if (const char *e = getenv("AGC_AMD_ASYNC_BOOK"))
        I.async_book = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_ASYNC_ENCODE"))
        I.async_encode = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_EARLY_COLLECT"))
        I.early_collect = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_REF_STORE_ASYNC"))
        I.ref_store_async = atoi(e) != 0;
    I.enc_alt.ctx = I.enc_alt2.ctx = I.hip;
    for (unsigned i = 0; i < nt; ++i)
        I.zctx.emplace_back(new ZstdCtx(&I.zstd));
    I.sync_entropy = getenv("AGC_AMD_SYNC_ENTROPY") != nullptr;
    if (const char *e = getenv("AGC_AMD_ENTROPY_STREAM"))
        I.entropy_stream = atoi(e) != 0;
    I.enc_buf.ctx = I.enc_buf2.ctx = I.dist_body_buf.ctx = I.dist_record_buf.ctx = I.hip;
    if (const char *e = getenv("AGC_AMD_PAR_MIN"))
        I.par_min = (size_t)std::max(1LL, atoll(e));
    if (const char *e = getenv("AGC_AMD_ENCODE_OVERLAP"))
        I.overlap_mode = !strcmp(e, "early") || !strcmp(e, "1") ? 1 : !strcmp(e, "late") || !strcmp(e, "2") ? 2 : 0;
    I.choose_entropy_stage();
    if (const char *e = getenv("AGC_AMD_DEV_SEGMENTS"))
        I.dev_segments = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_DEV_ENCODE_MIN"))
        I.dev_encode_min = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("AGC_AMD_PRE_LAUNCH_ENCODE"))
        I.pre_launch_encode = atoi(e) != 0;
    if (const char *e = getenv("AGC_AMD_PLACE_AHEAD"))
        I.place_ahead = atoi(e);
    if (const char *e = getenv("AGC_AMD_SPEC_FILL_AHEAD"))
        I.spec_fill_ahead = atoi(e);
    if (!PkMap::hash_agrees()) {
        I.err("internal: the host's and the device library's group hash differ");
        return false;
    }

    if (!I.ar.open(out_archive_name)) {
        I.err("Cannot create archive " + out_archive_name);
        return false;
    }
    I.created = true;
    // ---- collection: all batches but the last are copied now (direct parts), sample names are loaded
    I.coll.set_archive(&I.ar, &I.zstd, I.segment_size, I.k);
    std::vector<rd::SampleDesc> in_samples;
    if (!rd::parse_sample_names(I.in_ar, I.zd, in_samples))
        return false;
    {
        std::vector<std::string> names;
        for (auto &sd : in_samples)
            names.push_back(sd.name);
        I.coll.load_sample_names(names);
    }
    const size_t n_batches = I.in_ar.n_parts("collection-contigs");
    auto copy_batch = [&](size_t b) -> bool {
        if (!I.in_ar.get_part("collection-contigs", b, ptr, size, meta))
            return false;
        I.ar.add_part(I.coll.stream_contigs(), ptr, size, meta);
        if (!I.in_ar.get_part("collection-details", b, ptr, size, meta))
            return false;
        I.ar.add_part(I.coll.stream_details(), ptr, size, meta);
        return true;
    };
    for (size_t b = 0; b + 1 < n_batches; ++b)
        if (!copy_batch(b))
            return false;

    // ---- adaptive mode: singleton / duplicated k-mers of the reference sample, decoded from the input archive
    // (build_candidate_kmers_from_archive, agc_compressor.cpp:828-847) and counted on the GPU
    if (I.adaptive && !in_samples.empty()) {
        CAGCFile rdr;
        std::vector<std::string> names;
        std::vector<bytes_t> ref;
        if (!rdr.Open(in_archive_name) || !rdr.GetSampleCodes(in_samples.front().name, names, ref)) {
            I.err("Cannot decode the reference sample of " + in_archive_name);
            return false;
        }
        uint64_t tot = 0;
        for (auto &c : ref)
            tot += c.size();
        uint8_t *d_ref = nullptr;
        if (!I.hip_ok(DEVTI(agc_hip_sample_buffer(I.hip, tot, &d_ref)), "sample_buffer"))
            return false;
        std::vector<uint64_t> off(ref.size() + 1, 0);
        for (size_t i = 0; i < ref.size(); ++i) {
            if (!I.hip_ok(DEVTI(agc_hip_copy_to_device(I.hip, d_ref + off[i], ref[i].data(), ref[i].size())), "copy_to_device"))
                return false;
            off[i + 1] = off[i] + ref[i].size();
        }
        std::vector<uint64_t> spl(std::max<uint64_t>(1024, tot / std::max(1u, I.segment_size) * 2 + 2 * ref.size() + 16));
        std::vector<uint64_t> sorted_kmers(tot);
        uint64_t n_spl = 0, n_sorted = 0;
        for (;;) {
            int rc = agc_hip_determine_splitters_dev(I.hip, d_ref, off.data(), (uint32_t)ref.size(), I.k, I.segment_size, spl.size(), spl.data(),
                                                     &n_spl, sorted_kmers.size(), sorted_kmers.data(), &n_sorted);
            if (rc == AGC_HIP_ECAP && n_spl > spl.size()) {
                spl.resize(n_spl);
                continue;
            }
            if (!I.hip_ok(rc, "determine_splitters"))
                return false;
            break;
        }
        sorted_kmers.resize(n_sorted);
        split_singletons(sorted_kmers, &I.ref_duplicates);
        I.ref_singletons.swap(sorted_kmers);
    }

    // ---- appending_init: the last collection batch is loaded (and copied as well when it is full)
    if (n_batches) {
        const size_t first = (n_batches - 1) * (size_t)I.pack_cardinality;
        if (!rd::parse_contig_batch(I.in_ar, I.zd, (uint32_t)(n_batches - 1), I.pack_cardinality, I.segment_size, I.k, in_samples))
            return false;
        const size_t n_last = in_samples.size() > first ? in_samples.size() - first : 0;
        if (n_last == I.pack_cardinality) {
            if (!copy_batch(n_batches - 1))
                return false;
        } else
            for (size_t i = first; i < in_samples.size(); ++i) {
                auto &dst = I.coll.sample_at(i);
                for (auto &c : in_samples[i].ctgs) {
                    dst.contigs.emplace_back();
                    dst.contigs.back().name = c.name;
                    for (auto &sg : c.segs)
                        dst.contigs.back().segments.push_back({sg.group_id, sg.in_group_id, sg.raw_length, sg.rc});
                }
            }
    }
    // ---- groups: reference part and all delta parts but the last are copied, the last one stays packed
    I.groups.clear();
    for (I.no_segments = 0;; ++I.no_segments) {
        const std::string rn = ss_ref_name(I.no_segments), dn = ss_delta_name(I.no_segments);
        const bool has_r = I.in_ar.ids.count(rn) != 0, has_d = I.in_ar.ids.count(dn) != 0;
        if (!has_r && !has_d)
            break;
        I.groups.emplace_back();
        Group &g = I.groups.back();
        g.exists = true;
        g.packed = true;
        if (has_r)
            g.stream_ref = I.ar.register_stream(rn);
        if (has_d)
            g.stream_delta = I.ar.register_stream(dn);
        if (has_r && I.in_ar.get_part(rn, 0, g.pk_ref, g.pk_ref_size, g.pk_ref_meta)) {
            I.ar.add_part(g.stream_ref, g.pk_ref, g.pk_ref_size, g.pk_ref_meta);
            g.no_seqs = 1;
        } else
            g.pk_ref = nullptr;
        if (has_d) {
            const size_t np = I.in_ar.n_parts(dn);
            for (size_t i = 0; i + 1 < np; ++i) {
                if (!I.in_ar.get_part(dn, i, ptr, size, meta))
                    return false;
                I.ar.add_part(g.stream_delta, ptr, size, meta);
                g.no_seqs += I.pack_cardinality;
            }
            if (np && !I.in_ar.get_part(dn, np - 1, g.pk_delta, g.pk_delta_size, g.pk_delta_meta))
                return false;
        }
    }
    // ---- splitters and the (kmer1, kmer2) -> group map
    if (!I.in_ar.get_part("splitters", 0, ptr, size, meta) || size < meta * 8)
        return false;
    {
        std::vector<uint64_t> spl(meta);
        for (uint64_t i = 0; i < meta; ++i)
            spl[i] = le64(ptr + 8 * i);
        if (!SetSplitters(spl.data(), spl.size()))
            return false;
    }
    if (!I.in_ar.get_part("segment-splitters", 0, ptr, size, meta) || size < meta * 20)
        return false;
    I.map_segments.clear();
    I.terminators.clear();
    I.map_segments[{NO_KMER, NO_KMER}] = 0;
    for (uint64_t i = 0; i < meta; ++i) {
        const uint64_t x1 = le64(ptr + 20 * i), x2 = le64(ptr + 20 * i + 8);
        I.map_segments[{x1, x2}] = (int32_t)le32(ptr + 20 * i + 16);
        if (x1 != NO_KMER && x2 != NO_KMER) {
            I.terminators[x1].push_back(x2);
            if (x1 != x2)
                I.terminators[x2].push_back(x1);
        }
    }
    for (auto &t : I.terminators)
        std::sort(t.second.begin(), t.second.end());
    I.coll.reset_prev_sample_name();
    // (the sample counters of a session that adds through the device API / commit records; AddSampleFiles sets them itself,
    // agc_compressor.cpp:2150-2153)
    I.processed_samples = (uint32_t)I.coll.no_samples();
    I.stored_samples = I.processed_samples / I.pack_cardinality * I.pack_cardinality;
    return true;
}

// CSegment::unpack (segment.cpp:496-577): the reference goes to HBM (index built on first use there), the last pack of the
// input archive becomes the group's current pack again
bool CAGCCompressor::Impl::unpack_group(uint32_t gid)
{
    Group &g = groups[gid];
    if (g.pk_ref) {
        bytes_t ref;
        if (!rd::decode_ref_part(zd, g.pk_ref, g.pk_ref_size, g.pk_ref_meta, ref)) {
            err("cannot decode the reference of group " + std::to_string(gid));
            return false;
        }
        if (!hip_ok(DEVT(agc_hip_ref_register(hip, gid, ref.data(), (uint32_t)ref.size(), mml)), "ref_register"))
            return false;
        g.ref_size = ref.size() + 1;
        g.pk_ref = nullptr;
    }
    if (g.pk_delta) {
        bytes_t pack;
        if (!rd::decode_pack_part(zd, g.pk_delta, g.pk_delta_size, g.pk_delta_meta, pack)) {
            err("cannot decode the last pack of group " + std::to_string(gid));
            return false;
        }
        std::vector<uint32_t> off;
        uint32_t b = 0;
        for (uint32_t i = 0; i < pack.size(); ++i)
            if (pack[i] == 0xff) {
                off.push_back(b);
                b = i + 1;
            }
        pack.resize(b); // (bytes after the last separator cannot occur)
        g.no_seqs += (uint32_t)off.size();
        if (g.ref_size == 0) { // no reference: the "deltas" are raw sequences (segment.cpp:569-570)
            g.raw_data.swap(pack);
            g.raw_off.swap(off);
        } else {
            g.lzp_data.swap(pack);
            g.lzp_off.swap(off);
        }
        g.pk_delta = nullptr;
    }
    g.packed = false;
    return true;
}

// Delta packs go to the GPU entropy stage when its frames are the ones the host's libzstd would write (the device encoder
// restates libzstd 1.4.9; 1.4.8 writes the same level-17 frames): references and metadata stay on libzstd, and one archive
// must not mix encoder versions.  AGC_AMD_HOST_ZSTD=1 keeps everything on the host, AGC_AMD_GPU_ZSTD=force overrides the check.
void CAGCCompressor::Impl::choose_entropy_stage()
{
    const std::string v = zstd.versionString ? zstd.versionString() : "";
    gpu_zstd = (v == "1.4.9" || v == "1.4.8");
    if (const char *e = getenv("AGC_AMD_GPU_ZSTD"))
        if (std::string(e) == "force")
            gpu_zstd = true;
    if (getenv("AGC_AMD_HOST_ZSTD"))
        gpu_zstd = false;
    if (const char *e = getenv("AGC_AMD_GPU_ZSTD_SHARE"))
        gpu_zstd_share = std::min(1.0, std::max(0.0, atof(e)));
    if (const char *e = getenv("AGC_AMD_GPU_ZSTD_MIN"))
        gpu_zstd_min = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("AGC_AMD_GPU_ZSTD_REFS"))
        gpu_zstd_refs_min = (uint32_t)std::max(0, atoi(e));
    if (verbosity > 0)
        std::cerr << "entropy stage: delta packs on " << (gpu_zstd ? "the GPU (zstd 1.4.9 frames)" : "host libzstd") << ", libzstd " << v << std::endl;
}

bool CAGCCompressor::AddSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                                     const uint64_t *ctg_off)
{
    return PrepareSampleDevice(sample_name, contig_names, d_codes, ctg_off) && CommitPrepared();
}

bool CAGCCompressor::AddSamplePackedDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const void *packed,
                                           const uint64_t *ctg_off)
{
    return PrepareSamplePackedDevice(sample_name, contig_names, packed, ctg_off) && CommitPrepared();
}

bool CAGCCompressor::PrepareSamplePackedDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const void *packed,
                                               const uint64_t *ctg_off)
{
    Impl &I = *p;
    if (!I.created || !packed)
        return false;
    const agc_hip_packed &pk = *(const agc_hip_packed *)packed;
    // every kernel reads the 2-bit words where they lie (no byte copy is made).  When this is the sample that was announced its
    // scan ran ahead of its turn and is collected instead of launched (scan_batch).
    I.scan_from_prefetch = I.pf_live.valid && I.pf_live.pk.d_words == pk.d_words && I.pf_live.pk.n_symbols == pk.n_symbols &&
                           I.pf_live.ctg_off.size() == contig_names.size() + 1 && std::equal(I.pf_live.ctg_off.begin(), I.pf_live.ctg_off.end(), ctg_off);
    I.pf_live.valid = false;
    I.packed_sample = pk;
    // (the caller keeps a packed sample untouched until the NEXT sample call, Drain or Close returns: its LZ encode may be
    // collected by the bookkeeping thread after this call has returned -- compressor.h)
    I.next_base_owned = true;
    const bool ok = PrepareSampleDevice(sample_name, contig_names, nullptr, ctg_off);
    I.next_base_owned = false;
    I.packed_sample.n_symbols = 0; // (scans of the commit phase -- adaptive mode -- run inside PrepareSampleDevice as well)
    I.scan_from_prefetch = false;
    return ok;
}

bool CAGCCompressor::SetNextSamplePackedDevice(const void *packed, const uint64_t *ctg_off, uint32_t n_ctg)
{
    Impl &I = *p;
    I.pf_next.valid = false;
    if (!I.created || !packed || !ctg_off || !n_ctg || I.adaptive || I.appending || I.concatenated || I.k < 16)
        return false;
    I.pf_next.pk = *(const agc_hip_packed *)packed;
    I.pf_next.ctg_off.assign(ctg_off, ctg_off + n_ctg + 1);
    I.pf_next.valid = I.pf_next.pk.n_symbols != 0;
    return I.pf_next.valid;
}

// bytes in HBM -> the 2-bit layout in the device context's own buffers: every LZ entry point reads that form only
bool CAGCCompressor::Impl::pack_sample(const uint8_t *d_codes, uint64_t n_symbols)
{
    packed_sample = agc_hip_packed{};
    if (!n_symbols)
        return true;
    return hip_ok(DEVT(agc_hip_sample_pack(hip, d_codes, n_symbols, &packed_sample)), "sample_pack");
}

bool CAGCCompressor::SetNextFastaDevice(const uint8_t *d_raw, uint64_t n_raw, const uint64_t *raw_begin, const uint64_t *raw_end, uint32_t n_ctg,
                                        uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks)
{
    Impl &I = *p;
    if (!I.created || I.fasta_next.live || (n_ctg && (!raw_begin || !raw_end)))
        return false;
    Impl::FastaNext &f = I.fasta_next;
    f.d_raw = d_raw;
    f.n_raw = n_raw;
    f.rb.assign(raw_begin, raw_begin + n_ctg);
    f.re.assign(raw_end, raw_end + n_ctg);
    f.d_words = d_words;
    f.d_idx = d_esc_index;
    f.d_esc = d_esc_bytes;
    f.esc_cap = esc_cap_blocks;
    f.valid = true;
    return true;
}

bool CAGCCompressor::Impl::launch_fasta()
{
    FastaNext &f = fasta_next;
    if (!f.valid)
        return true;
    f.valid = false;
    const int rc = agc_hip_pack_fasta_begin(hip, f.d_raw, f.n_raw, f.rb.data(), f.re.data(), (uint32_t)f.rb.size(), f.d_words, f.d_idx, f.d_esc, f.esc_cap);
    f.live = rc == AGC_HIP_OK;
    return hip_ok(rc, "pack_fasta_begin");
}

int CAGCCompressor::FinishFastaDevice(uint64_t *ctg_off, uint64_t *n_esc_blocks)
{
    Impl &I = *p;
    if (!I.created || (!I.fasta_next.valid && !I.fasta_next.live))
        return AGC_HIP_EINVAL;
    if (I.fasta_next.valid && !I.launch_fasta())
        return AGC_HIP_EINVAL;
    I.fasta_next.live = false;
    return agc_hip_pack_fasta_end(I.hip, ctg_off, n_esc_blocks);
}

// queues the announced sample's scan on the device (called once the current sample's own classification kernels are in)
void CAGCCompressor::Impl::launch_prefetch()
{
    (void)launch_fasta(); // (a conversion that does not start is reported by FinishFastaDevice)
    if (!pf_next.valid)
        return;
    pf_next.valid = false;
    if (agc_hip_prefetch_packed_dev(hip, &pf_next.pk, pf_next.ctg_off.data(), (uint32_t)pf_next.ctg_off.size() - 1, k) != AGC_HIP_OK)
        return; // (nothing lost: the sample goes the ordinary way when its turn comes)
    pf_live.pk = pf_next.pk;
    pf_live.ctg_off.swap(pf_next.ctg_off);
    pf_live.valid = true;
}

namespace {
constexpr uint64_t CompressorStats::*PREPARE_STATS[] = {&CompressorStats::bases, &CompressorStats::one_splitter, &CompressorStats::middle_tried,
                                                         &CompressorStats::middle_split, &CompressorStats::lz_encoded, &CompressorStats::delta_bytes,
                                                         &CompressorStats::enc_text, &CompressorStats::enc_ref, &CompressorStats::est_text,
                                                         &CompressorStats::est_ref, &CompressorStats::cv_text, &CompressorStats::cv_ref,
                                                         &CompressorStats::windows};
void snapshot_prepare_stats(const CompressorStats &st, uint64_t *out)
{
    size_t t = 0;
    for (uint64_t CompressorStats::*f : PREPARE_STATS)
        out[t++] = st.*f;
}
} // namespace

// scan + classification + speculative encode of a sample, against the state this process has NOW; nothing is registered
// yet.  d_codes must stay untouched until CommitPrepared.  In the multi-GPU mode a rank calls this for its next sample
// while earlier samples are still being committed elsewhere (agc_amd/dist.py).
bool CAGCCompressor::PrepareSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                                         const uint64_t *ctg_off)
{
    Impl &I = *p;
    // (-c mode: the caller hands over the reference's registration units -- runs of pack_cardinality contigs, sample name empty:
    // every contig is a sample of its own -- and, at the end, the empty one the reference always sends, agc_compressor.cpp:2231-2238)
    if (!I.created || I.prepared || I.committing || I.prep.deferred)
        return false;
    // a sample handed over one byte per symbol is packed first (context-owned buffers: they outlive this call)
    struct Unpack {
        Impl &I;
        bool on = false, owned = false;
        ~Unpack()
        {
            if (on) {
                I.packed_sample.n_symbols = 0;
                I.next_base_owned = owned;
            }
        }
    } unpack{I};
    if (!I.packed_sample.n_symbols && d_codes && !contig_names.empty() && ctg_off[contig_names.size()]) {
        if (!I.pack_sample(d_codes, ctg_off[contig_names.size()]))
            return false;
        unpack.on = true;
        unpack.owned = I.next_base_owned;
        I.next_base_owned = true;
    }
    I.prepared_ctgs.clear();
    for (size_t c = 0; c < contig_names.size(); ++c) {
        Contig ct;
        ct.sample = sample_name;
        ct.name = contig_names[c];
        ct.off = ctg_off[c];
        ct.len = ctg_off[c + 1] - ctg_off[c];
        I.prepared_ctgs.push_back(ct);
    }
    I.changed_log.clear();
    I.minted_since_prepare = false;
    I.prepared.reset(new Impl::BatchState());
    // one archive from N ranks: the sample may be ahead of its turn.  Everything that can be encoded already is; in adaptive mode
    // the scan must not extend the splitter set (the samples in front come first): a sample that would have to is prepared again
    // at its turn (CommitPreparedHead), and so is one whose splitter set grew meanwhile
    const bool ahead = I.dist_world > 1 && !I.appending;
    I.prep.d_codes = d_codes;
    I.prep.packed = I.packed_sample;
    I.prep.base_owned = I.next_base_owned;
    I.prep.deferred = false;
    I.prep.spl_version = I.spl_version;
    I.prep.ahead = ahead;
    snapshot_prepare_stats(I.st, I.prep.st_before);
    I.prepared->no_new_splitters = ahead && I.adaptive;
    if (!I.batch_prepare(*I.prepared, I.prepared_ctgs, d_codes, nullptr, ahead)) {
        I.prepared.reset();
        return false;
    }
    if (I.prepared->needs_turn) {
        I.prepared.reset();
        I.prep.deferred = true;
    }
    snapshot_prepare_stats(I.st, I.prep.st_after);
    return true;
}

// the order-dependent half: collection registration, revalidation against what changed since PrepareSampleDevice, commit
bool CAGCCompressor::CommitPrepared() { return CommitPreparedHead() && CommitPreparedFinish(); }

// ... in two steps.  After the first one the sample is registered (group ids, keys, terminators) and, in the multi-GPU mode, the
// HEAD of its commit record is ready (LastRecord): the other ranks can apply it and go on while this rank finishes.
bool CAGCCompressor::CommitPreparedHead()
{
    Impl &I = *p;
    if (I.committing)
        return false;
    // (only a prepare made AHEAD of the turn can be stale: one made at its turn -- append mode -- mined its own new splitters, which is
    // what moved spl_version)
    if (I.dist_world > 1 && I.adaptive && I.prep.ahead && (I.prep.deferred || (I.prepared && I.prep.spl_version != I.spl_version))) {
        // adaptive mode, the sample's turn: the speculative prepare did not stand (new splitters needed, or brought by the samples
        // in front) -- the plain prepare, against the state as it is now
        // (what the dropped prepare counted -- bases, texts handed to the LZ kernels, deltas -- is taken back: the sample is
        // counted once, by the prepare that stands)
        if (I.prepared) {
            size_t t = 0;
            for (uint64_t CompressorStats::*f : PREPARE_STATS)
                I.st.*f -= I.prep.st_after[t] - I.prep.st_before[t], ++t;
        }
        ++I.st.reprepared;
        if (I.prepared && I.prepared->dev_enc_n) { // (its whole-sample encode is on the device's first lane: given up)
            if (!I.hip_ok(agc_hip_lz_encode_drop_on(I.hip, 0), "lz_encode_drop"))
                return false;
            I.lane2_release();
        }
        I.prepared.reset(new Impl::BatchState());
        I.changed_log.clear();
        I.minted_since_prepare = false;
        I.prep.deferred = false;
        I.packed_sample = I.prep.packed;
        I.next_base_owned = I.prep.base_owned;
        I.scan_from_prefetch = false;
        const bool ok = I.batch_prepare(*I.prepared, I.prepared_ctgs, I.prep.d_codes, nullptr, false);
        I.packed_sample.n_symbols = 0;
        I.next_base_owned = false;
        if (!ok) {
            I.prepared.reset();
            return false;
        }
    }
    if (!I.prepared)
        return false;
    std::unique_ptr<Impl::BatchState> b = std::move(I.prepared); // (note_new_group stops logging)
    I.dist_record_n = 0;
    I.dist_body_n = 0;
    {
        std::lock_guard<std::mutex> coll_lk(I.coll_mtx); // (the bookkeeping of the previous sample may be looking its contigs up)
        I.coll.reset_prev_sample_name();
        for (auto &ct : I.prepared_ctgs)
            if (!I.coll.register_sample_contig(ct.sample, ct.name)) {
                I.err("Error: Pair sample_name:contig_name " + ct.sample + ":" + ct.name + " is already in the archive!");
                return false; // (AddSampleFiles skips such contigs; a device-resident sample is all or nothing)
            }
    }
    if (I.prepared_ctgs.empty() && !I.concatenated) {
        // a sample without contigs registers nothing (the reference warns and skips such a file, agc_compressor.cpp:2187-2195);
        // the other ranks still expect one record per sample: an empty one
        if (I.dist_world > 1)
            return I.make_empty_record();
        return true;
    }
    // Any group minted since PrepareSampleDevice -- also one keyed (k-mer, NO_KMER), which has no terminator entry and is
    // therefore not in changed_log -- may be the group of a prepared segment whose key was unknown then: the placement is
    // repeated (revalidate ends with stage_place), the classification only for what the changed terminator lists touch.
    if (I.minted_since_prepare || !I.changed_log.empty()) {
        b->changed.swap(I.changed_log);
        I.changed_log.clear();
        I.minted_since_prepare = false;
        b->s_from = 0;
        if (!I.revalidate(*b))
            return false;
    }
    // (a prepared sample is a window of one registration: one commit run, batch_commit's loop body in two halves)
    b->s_from = 0;
    ++I.st.commit_runs;
    if (!I.stage_register(*b) || !I.stage_store_head(*b))
        return false;
    I.committing = std::move(b);
    return true;
}

// the rest: the new references' index on this GPU, the deltas the speculative encode did not cover, the record's body, bookkeeping
bool CAGCCompressor::CommitPreparedFinish()
{
    Impl &I = *p;
    if (!I.committing)
        return I.created; // (a sample without contigs: nothing was left to do)
    std::unique_ptr<Impl::BatchState> b = std::move(I.committing);
    return I.stage_store_finish(*b);
}

bool CAGCCompressor::AddSampleFiles(const std::vector<std::pair<std::string, std::string>> &files, uint32_t no_threads)
{
    Impl &I = *p;
    if (!I.created)
        return false;
    if (files.empty())
        return true;
    if (!I.book_wait())
        return false;
    I.processed_samples = I.appending ? (uint32_t)I.coll.no_samples() : 0; // agc_compressor.cpp:2150-2153
    I.stored_samples = I.processed_samples / I.pack_cardinality * I.pack_cardinality;
    if (I.concatenated)
        I.cnt_contigs_in_sample = I.processed_samples % I.pack_cardinality;

    // Registration batches read from the files but not committed yet.  In the plain mode one batch = one
    // sample file and several consecutive batches may be classified together (speculation window, see
    // process_batch); in -c mode a batch = pack_cardinality contigs (windows work the same way), in -a mode the window is one batch.
    struct Pending {
        std::vector<Contig> ctgs;
        std::vector<bytes_t> data;
        std::vector<uint8_t> raw; // data[c] is a raw FASTA body (converted at upload)
        uint64_t bytes = 0;
    };
    std::deque<Pending> pending;
    uint64_t pending_bytes = 0;
    const uint64_t WINDOW_BYTES = 64ull << 20;
    // (AGC_AMD_WINDOW_MAX: tests pin the window, e.g. to 1 = every registration on its own, bookkeeping beside the next one)
    static const uint32_t window_cap = getenv("AGC_AMD_WINDOW_MAX") ? (uint32_t)std::max(1, atoi(getenv("AGC_AMD_WINDOW_MAX"))) : 256u;
    // (adaptive mode: new splitters change later scans -- a window is cut at the registration that brings some, stage_scan_dev;
    // without the device's segments there is no such cut and no speculation)
    const uint32_t WINDOW_MAX = I.adaptive && !I.adaptive_windows() ? 1u : window_cap;
    uint32_t window = 1; // grows while whole windows commit, shrinks to what did commit otherwise
    // (adaptive mode mines new splitters from host copies of the codes, k < 16 scans one byte per symbol: the earlier path.
    // AGC_AMD_FASTA_PACK=0 switches the one-pass conversion off, AGC_AMD_FASTA_PACK_MIN=<bytes> moves its threshold)
    const bool fasta_ok = !I.adaptive && I.k >= 16 && !(getenv("AGC_AMD_FASTA_PACK") && atoi(getenv("AGC_AMD_FASTA_PACK")) == 0);
    const uint64_t FASTA_PACK_MIN = getenv("AGC_AMD_FASTA_PACK_MIN") ? strtoull(getenv("AGC_AMD_FASTA_PACK_MIN"), nullptr, 10) : (1ull << 20);

    auto run_window = [&]() -> bool {
        // batch = the first `window` pending registrations (at least one)
        const uint32_t nb = (uint32_t)std::min<size_t>(pending.size(), window);
        std::vector<Contig> batch;
        std::vector<bytes_t> batch_data;
        uint64_t tot = 0;
        for (uint32_t b = 0; b < nb; ++b)
            tot += pending[b].bytes;
        uint8_t *d_base = nullptr;
        double t0 = now();
        // S1a on the product path: a window whose contigs are all still FASTA bodies goes to HBM as it is and is converted + packed
        // there in one pass (agc_hip_sample_pack_fasta -> pack_fasta_*_kernel: preprocess_raw_contig, agc_compressor.cpp:907-951, fused
        // with the 2-bit packing) -- no one-byte-per-symbol copy of the sample exists anywhere, no per-contig round trip
        bool all_raw = fasta_ok && nb > 0 && tot >= FASTA_PACK_MIN;
        for (uint32_t b = 0; b < nb && all_raw; ++b)
            for (size_t c = 0; c < pending[b].ctgs.size(); ++c)
                all_raw = all_raw && c < pending[b].raw.size() && pending[b].raw[c];
        if (all_raw) {
            std::vector<const uint8_t *> ptr;
            std::vector<uint64_t> len;
            for (uint32_t b = 0; b < nb; ++b)
                for (size_t c = 0; c < pending[b].ctgs.size(); ++c) {
                    ptr.push_back(pending[b].data[c].data());
                    len.push_back(pending[b].data[c].size());
                }
            std::vector<uint64_t> off(ptr.size() + 1, 0);
            agc_hip_packed pk{};
            if (!I.hip_ok(DEVTI(agc_hip_sample_pack_fasta(I.hip, (uint32_t)ptr.size(), ptr.data(), len.data(), &pk, off.data())), "sample_pack_fasta"))
                return false;
            size_t x = 0;
            for (uint32_t b = 0; b < nb; ++b)
                for (size_t c = 0; c < pending[b].ctgs.size(); ++c, ++x) {
                    Contig ct = pending[b].ctgs[c];
                    ct.sample_idx = b;
                    ct.off = off[x];
                    ct.len = off[x + 1] - off[x];
                    batch.push_back(ct);
                }
            I.st.t_io += now() - t0;
            uint32_t n_done = 0;
            I.next_base_owned = true; // (the context's own packed buffers)
            I.packed_sample = pk;
            const bool batch_ok = I.process_batch(batch, nullptr, nullptr, n_done);
            I.packed_sample.n_symbols = 0;
            I.next_base_owned = false;
            if (!batch_ok)
                return false;
            if (n_done == nb)
                window = std::min(WINDOW_MAX, window * 2);
            else
                window = std::max(1u, n_done);
            for (uint32_t b = 0; b < n_done; ++b) {
                pending_bytes -= pending.front().bytes;
                pending.pop_front();
            }
            return true;
        }
        if (!I.hip_ok(DEVTI(agc_hip_sample_buffer(I.hip, tot, &d_base)), "sample_buffer"))
            return false;
        uint64_t o = 0;
        for (uint32_t b = 0; b < nb; ++b)
            for (size_t c = 0; c < pending[b].ctgs.size(); ++c) {
                Contig ct = pending[b].ctgs[c];
                ct.sample_idx = b;
                ct.off = o;
                ct.len = pending[b].data[c].size();
                if (c < pending[b].raw.size() && pending[b].raw[c]) { // a1 on the GPU; the converted length comes back
                    uint64_t n_codes = 0;
                    if (!I.hip_ok(DEVTI(agc_hip_preprocess(I.hip, pending[b].data[c].data(), ct.len, d_base + o, &n_codes)), "preprocess"))
                        return false;
                    ct.len = n_codes;
                } else if (!I.hip_ok(DEVTI(agc_hip_copy_to_device(I.hip, d_base + o, pending[b].data[c].data(), ct.len)), "copy_to_device"))
                    return false;
                o += ct.len;
                batch.push_back(ct);
                if (I.adaptive)
                    batch_data.push_back(pending[b].data[c]); // new splitters are mined from the host copy
            }
        I.st.t_io += now() - t0;
        uint32_t n_done = 0;
        I.next_base_owned = true; // (agc_hip_sample_buffer / agc_hip_sample_pack)
        // the window in the 2-bit layout: what the scan (k >= 16) and every LZ entry point read
        const bool batch_ok = I.pack_sample(d_base, o) && I.process_batch(batch, d_base, I.adaptive ? &batch_data : nullptr, n_done);
        I.packed_sample.n_symbols = 0;
        I.next_base_owned = false;
        if (!batch_ok)
            return false;
        if (nb == 0) // an empty registration (the reference's trailing token in -c mode)
            return true;
        if (n_done == nb)
            window = std::min(WINDOW_MAX, window * 2);
        else
            window = std::max(1u, n_done);
        for (uint32_t b = 0; b < n_done; ++b) {
            pending_bytes -= pending.front().bytes;
            pending.pop_front();
        }
        return true;
    };
    auto push_batch = [&](Pending &&pb) -> bool {
        pending_bytes += pb.bytes;
        pending.emplace_back(std::move(pb));
        // run as soon as a full window is available (or the read-ahead budget is used up)
        while (!pending.empty() && (pending.size() >= window || pending_bytes >= WINDOW_BYTES))
            if (!run_window())
                return false;
        return true;
    };
    auto drain = [&]() -> bool {
        while (!pending.empty())
            if (!run_window())
                return false;
        return true;
    };

    // Files are read, split into contigs and converted to symbol codes by background tasks, a few files ahead of the one
    // being registered (the reference reads on its main thread and converts on the workers, agc_compressor.cpp:2160-2228);
    // registration itself stays strictly in file order on this thread.
    struct FileData {
        bool opened = false;
        std::vector<std::string> ids;
        std::vector<bytes_t> contigs;
        std::vector<uint8_t> raw; // contig still holds the FASTA body (line ends and all): converted on the GPU at upload
    };
    // preprocess_raw_contig (agc_compressor.cpp:907-951) of big contigs runs on the GPU (agc_hip_preprocess: the body goes to
    // HBM as it is, the kernels drop the line ends and map the letters); small ones -- and adaptive mode, which mines new
    // splitters from host copies -- are converted by the reading thread as before
    const uint64_t GPU_A1_MIN = I.adaptive ? ~0ull : (1ull << 20);
    const unsigned read_threads = std::max(1u, std::min(8u, no_threads / 2));
    // (a file of FASTA_PACK_MIN bytes or more keeps EVERY contig as it was read -- the small scaffolds of an assembly beside its
    // chromosomes -- so that its window qualifies for the one-pass conversion; smaller files: per contig, as before)
    auto read_file = [GPU_A1_MIN, read_threads, fasta_ok, FASTA_PACK_MIN](std::string path) {
        FileData fd;
        auto settle = [&]() {
            uint64_t total = 0;
            for (const bytes_t &c_ : fd.contigs)
                total += c_.size();
            const bool whole = fasta_ok && total >= FASTA_PACK_MIN;
            fd.raw.resize(fd.contigs.size());
            for (size_t i = 0; i < fd.contigs.size(); ++i) {
                fd.raw[i] = whole || fd.contigs[i].size() >= GPU_A1_MIN;
                if (!fd.raw[i])
                    preprocess_raw_contig(fd.contigs[i]);
            }
        };
        // a big plain file: mapped, cut and copied out by a few threads
        static const uint64_t map_min = getenv("AGC_AMD_MAP_MIN") ? strtoull(getenv("AGC_AMD_MAP_MIN"), nullptr, 10) : (64ull << 20);
        if (FastaReader::read_all_mapped(path, fd.ids, fd.contigs, read_threads, map_min)) {
            fd.opened = true;
            settle();
            return fd;
        }
        FastaReader fr;
        if (!fr.open(path))
            return fd;
        fd.opened = true;
        std::string id;
        bytes_t contig;
        while (fr.read_contig_raw(id, contig)) {
            fd.ids.emplace_back(id);
            fd.contigs.emplace_back(std::move(contig));
            contig.clear();
        }
        settle();
        return fd;
    };
    auto file_bytes = [](const std::string &path) -> uint64_t {
        std::error_code ec;
        const auto n = std::filesystem::file_size(path, ec);
        if (ec)
            return 0;
        const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        return (uint64_t)n * (gz ? 4 : 1); // what it will occupy once read
    };
    const size_t READ_AHEAD = std::max<size_t>(1, std::min<size_t>(8, no_threads)); // files in flight
    const uint64_t READ_AHEAD_BYTES = 8ull << 30;                                    // (estimated) bytes of sequence in flight
    // (small genomes -- a file per virus -- are mostly the latency of opening them and of starting a task: up to 8 of them per
    // task, four times as many files in flight)
    const uint64_t SMALL_FILE = 1u << 20;
    std::deque<std::future<std::vector<FileData>>> inflight;
    std::deque<uint64_t> inflight_sz;
    std::deque<size_t> inflight_n;
    std::deque<FileData> ready;
    uint64_t inflight_bytes = 0;
    size_t inflight_files = 0, next_launch = 0;
    auto launch = [&]() {
        while (next_launch < files.size()) {
            uint64_t sz = file_bytes(files[next_launch].second);
            const bool small = sz < SMALL_FILE;
            if (!inflight.empty() && (inflight_files + ready.size() >= (small ? 4 * READ_AHEAD : READ_AHEAD) || inflight_bytes + sz > READ_AHEAD_BYTES))
                break;
            std::vector<std::string> paths{files[next_launch++].second};
            while (small && paths.size() < 8 && next_launch < files.size()) {
                const uint64_t s2 = file_bytes(files[next_launch].second);
                if (s2 >= SMALL_FILE)
                    break;
                sz += s2;
                paths.push_back(files[next_launch++].second);
            }
            inflight_n.push_back(paths.size());
            inflight_files += paths.size();
            inflight.emplace_back(std::async(std::launch::async, [read_file](std::vector<std::string> ps) {
                std::vector<FileData> v;
                for (auto &p_ : ps)
                    v.emplace_back(read_file(p_));
                return v;
            }, std::move(paths)));
            inflight_sz.push_back(sz);
            inflight_bytes += sz;
        }
    };

    Pending cur;
    for (auto &sf : files) {
        {
            std::lock_guard<std::mutex> coll_lk(I.coll_mtx);
            I.coll.reset_prev_sample_name();
        }
        double t0 = now();
        launch();
        if (ready.empty()) {
            for (FileData &f_ : inflight.front().get())
                ready.emplace_back(std::move(f_));
            inflight.pop_front();
            inflight_bytes -= inflight_sz.front();
            inflight_sz.pop_front();
            inflight_files -= inflight_n.front();
            inflight_n.pop_front();
        }
        FileData fd = std::move(ready.front());
        ready.pop_front();
        launch();
        if (!fd.opened) {
            I.err("Cannot open file: " + sf.second);
            continue;
        }
        bool any_read = false, any_added = false;
        for (size_t ci = 0; ci < fd.ids.size(); ++ci) {
            const std::string &id = fd.ids[ci];
            bytes_t &contig = fd.contigs[ci];
            const std::string sname = I.concatenated ? std::string() : sf.first;
            bool registered;
            {
                std::lock_guard<std::mutex> coll_lk(I.coll_mtx); // (the bookkeeping of an earlier file may be looking its contigs up)
                registered = I.coll.register_sample_contig(sname, id);
            }
            if (!registered)
                I.err("Error: Pair sample_name:contig_name " + (I.concatenated ? id : sf.first) + ":" + id + " is already in the archive!");
            else {
                Contig ct;
                ct.sample = sname;
                ct.name = id;
                cur.ctgs.push_back(ct);
                cur.bytes += contig.size();
                cur.data.emplace_back(std::move(contig));
                cur.raw.push_back(ci < fd.raw.size() ? fd.raw[ci] : 0);
                any_added = true;
                if (I.concatenated && ++I.cnt_contigs_in_sample >= I.pack_cardinality) {
                    I.st.t_io += now() - t0;
                    if (!push_batch(std::move(cur)))
                        return false;
                    cur = Pending();
                    t0 = now();
                    I.cnt_contigs_in_sample = 0;
                }
            }
            any_read = true;
        }
        I.st.t_io += now() - t0;
        if (!any_read)
            I.err("Warning: Pair sample_name:file_path " + sf.first + ":" + sf.second + " contains no contigs and will not be included in the archive!");
        if (!any_added)
            I.err("Warning: Pair sample_name:file_path " + sf.first + ":" + sf.second + " contains only contigs already present in the archive!");
        if (!I.concatenated && any_added) {
            if (!push_batch(std::move(cur)))
                return false;
            cur = Pending();
        }
    }
    if (I.concatenated) {
        // the reference always sends one more registration token at the end (:2231-2238)
        if (!push_batch(std::move(cur)))
            return false;
        if (!drain())
            return false;
        I.cnt_contigs_in_sample = 0;
        I.processed_samples = (uint32_t)I.coll.no_samples();
    } else if (!drain())
        return false;
    if (!I.book_wait())
        return false;
    if (I.processed_samples % I.pack_cardinality != 0) {
        I.coll.store_contig_batch((I.processed_samples / I.pack_cardinality) * I.pack_cardinality, I.processed_samples);
        I.stored_samples = I.processed_samples;
    }
    I.ar.flush_out_buffers();
    return true;
}

// close_compression (agc_compressor.cpp:2094-2115), store_metadata (:175-284), store_file_type_info (:287-300)
bool CAGCCompressor::CloseCollectPacks(const uint8_t **src, const uint64_t **off, uint32_t *n)
{
    Impl &I = *p;
    if (!I.created || I.close_collected || !src || !off || !n)
        return false;
    if (!I.book_wait())
        return false;
    I.settle_deals_locally();
    I.z_wait_all(); // (the staging buffers below are the entropy thread's)
    I.store_open_batch();
    I.close_jobs.clear();
    for (ZJob &j : I.deferred_packs) // the packs that filled during the run come first (their parts are already placed)
        I.close_jobs.emplace_back(std::move(j));
    I.deferred_packs.clear();
    I.deferred_bytes = 0;
    const size_t n_kept = I.close_jobs.size();
    I.build_close_jobs(I.close_jobs);
    if (getenv("AGC_AMD_LAPS"))
        std::cerr << "  CloseCollectPacks: " << n_kept << " packs kept from the run + " << I.close_jobs.size() - n_kept << " open ones\n";
    I.close_dev_jobs.clear();
    const uint32_t dev_max = agc_hip_zstd17_max_input();
    I.close_src_off.assign(1, 0);
    for (uint32_t i = 0; i < I.close_jobs.size(); ++i) {
        const ZJob &j = I.close_jobs[i];
        if (I.gpu_zstd && j.kind == 1 && !j.data.empty() && j.data.size() <= dev_max) {
            I.close_dev_jobs.push_back(i);
            I.close_src_off.push_back(I.close_src_off.back() + j.data.size());
        }
    }
    I.zsrc_buf.resize(I.close_src_off.back(), false);
    I.pool->parallel_for(I.close_dev_jobs.size(), [&](size_t t, unsigned) {
        const ZJob &j = I.close_jobs[I.close_dev_jobs[t]];
        memcpy(I.zsrc_buf.data() + I.close_src_off[t], j.data.data(), j.data.size());
    });
    *src = I.zsrc_buf.data();
    *off = I.close_src_off.data();
    *n = (uint32_t)I.close_dev_jobs.size();
    I.close_collected = true;
    I.close_frames_off.clear();
    return true;
}

// a pack's frame as add_to_archive stores it (segment.h:177-187: frame + a zero byte; the pack itself, metadata 0, when that is
// not shorter), into the part's slot
void CAGCCompressor::Impl::publish_frame(ZJob &j, const uint8_t *frame, size_t n)
{
    if (n + 1 < j.data.size()) {
        j.out.assign(frame, frame + n);
        j.out.push_back(0);
        j.meta = j.data.size();
    } else {
        j.out = j.data;
        j.meta = 0;
    }
    {
        std::lock_guard<std::mutex> lk(z_mtx); // (the entropy thread adds to the same counters)
        st.zstd_in += j.data.size();
        st.zstd_out += j.out.size();
    }
    j.slot->out = std::move(j.out);
    j.slot->meta = j.meta;
    j.slot->ready.store(true, std::memory_order_release);
}

uint64_t CAGCCompressor::DeferredPackBytes()
{
    std::lock_guard<std::mutex> lk(p->deferred_mtx);
    return p->deferred_bytes;
}

bool CAGCCompressor::DealCollectPacks(uint32_t *deal_id, const uint8_t **src, const uint64_t **off, uint32_t *n)
{
    Impl &I = *p;
    if (!I.created || !deal_id || !src || !off || !n)
        return false;
    I.deals.emplace_back();
    Impl::Deal &d = I.deals.back();
    {
        std::lock_guard<std::mutex> lk(I.deferred_mtx);
        d.jobs.swap(I.deferred_packs);
        I.deferred_bytes = 0;
    }
    d.id = I.next_deal_id++;
    d.left = (uint32_t)d.jobs.size();
    d.off.assign(d.jobs.size() + 1, 0);
    for (size_t i = 0; i < d.jobs.size(); ++i)
        d.off[i + 1] = d.off[i] + d.jobs[i].data.size();
    d.src.resize(d.off.back());
    I.pool->parallel_for(std::min<size_t>(d.jobs.size(), (size_t)I.pool->size() * 4), [&](size_t ci, unsigned) {
        const size_t nc = std::min<size_t>(d.jobs.size(), (size_t)I.pool->size() * 4);
        for (size_t t = d.jobs.size() * ci / nc; t < d.jobs.size() * (ci + 1) / nc; ++t)
            memcpy(d.src.data() + d.off[t], d.jobs[t].data.data(), d.jobs[t].data.size());
    });
    *deal_id = d.id;
    *src = d.src.data();
    *off = d.off.data();
    *n = (uint32_t)d.jobs.size();
    if (d.jobs.empty())
        I.deals.pop_back();
    return true;
}

CAGCCompressor::Impl::Deal *CAGCCompressor::Impl::find_deal(uint32_t id)
{
    for (auto &d : deals)
        if (d.id == id)
            return &d;
    return nullptr;
}

bool CAGCCompressor::DealKeepOwn(uint32_t deal_id, uint32_t first, uint32_t count)
{
    Impl &I = *p;
    Impl::Deal *d = I.find_deal(deal_id);
    if (!d || (uint64_t)first + count > d->jobs.size())
        return false;
    std::vector<ZJob> mine;
    for (uint32_t t = first; t < first + count; ++t) {
        if (!d->jobs[t].slot)
            return false; // (handed out already)
        mine.emplace_back(std::move(d->jobs[t]));
        d->jobs[t].slot.reset();
    }
    d->left -= count;
    I.z_submit(std::move(mine)); // this rank's own entropy stage, beside its steps
    if (!d->left)
        I.deals.remove_if([&](const Impl::Deal &x) { return x.id == deal_id; });
    return true;
}

// a deal the caller never settled (Close came first): this rank's own entropy stage codes what is left of it
void CAGCCompressor::Impl::settle_deals_locally()
{
    for (Deal &d : deals) {
        std::vector<ZJob> mine;
        for (ZJob &j : d.jobs)
            if (j.slot)
                mine.emplace_back(std::move(j));
        z_submit(std::move(mine));
    }
    deals.clear();
}

bool CAGCCompressor::DealProvideFrames(uint32_t deal_id, uint32_t first, uint32_t count, const uint8_t *frames, const uint64_t *off)
{
    Impl &I = *p;
    Impl::Deal *d = I.find_deal(deal_id);
    if (!d || (uint64_t)first + count > d->jobs.size() || (count && (!frames || !off)))
        return false;
    for (uint32_t t = 0; t < count; ++t) {
        ZJob &j = d->jobs[first + t];
        if (!j.slot)
            return false;
        I.publish_frame(j, frames + off[t], (size_t)(off[t + 1] - off[t]));
        j.slot.reset();
        bytes_t().swap(j.data);
    }
    d->left -= count;
    if (!d->left)
        I.deals.remove_if([&](const Impl::Deal &x) { return x.id == deal_id; });
    I.ar.try_drain(); // (the parts behind them may be written now)
    return true;
}

bool CAGCCompressor::CloseProvideFrames(const uint8_t *frames, const uint64_t *off)
{
    Impl &I = *p;
    if (!I.close_collected || (!I.close_dev_jobs.empty() && (!frames || !off)))
        return false;
    const size_t nd = I.close_dev_jobs.size();
    if (nd == 0) { // nothing was handed out: nothing to read (the pointers may be null)
        I.close_frames_off.assign(1, 0);
        I.zdst_buf.resize(0, false);
        return true;
    }
    I.close_frames_off.assign(off, off + nd + 1);
    I.zdst_buf.resize(off[nd] - off[0], false);
    if (off[nd] > off[0])
        memcpy(I.zdst_buf.data(), frames + off[0], off[nd] - off[0]);
    const uint64_t base = off[0];
    for (auto &x : I.close_frames_off)
        x -= base;
    return true;
}

bool CAGCCompressor::Drain()
{
    if (!p->created || !p->book_wait())
        return false;
    p->z_wait_all();
    return true;
}

void CAGCCompressor::Impl::store_open_batch(bool flush)
{
    // samples that came through AddSampleDevice / ApplyRecord: the open collection batch is stored here, where
    // AddSampleFiles does it at its end (agc_compressor.cpp:2254-2255)
    if (stored_samples < processed_samples && processed_samples % pack_cardinality != 0) {
        coll.store_contig_batch((processed_samples / pack_cardinality) * pack_cardinality, processed_samples);
        stored_samples = processed_samples;
        if (flush)
            ar.flush_out_buffers();
    }
}

bool CAGCCompressor::Close(uint32_t no_threads)
{
    Impl &I = *p;
    (void)no_threads;
    if (!I.created)
        return false;
    if (I.close_collected && I.close_frames_off.size() != I.close_dev_jobs.size() + 1) {
        I.err("Close: CloseCollectPacks was called but the frames were never provided");
        return false;
    }
    if (!I.book_wait()) // every registration is in the books
        return false;
    I.settle_deals_locally();
    // the open collection batch (contig details of up to pack_cardinality samples: one thread of zstd-19 work, 0.4 s at human
    // scale) is serialised while the entropy stage of the delta packs runs; both only buffer parts, which are flushed below in
    // stream-id order whatever their arrival order
    std::future<void> open_batch;
    if (!I.close_collected)
        open_batch = std::async(std::launch::async, [&I] { I.store_open_batch(false); });
    const bool laps = getenv("AGC_AMD_LAPS") != nullptr;
    double lt = now();
    auto LAP = [&](const char *what) {
        if (laps)
            std::cerr << "  close lap " << what << " " << (now() - lt) * 1e3 << " ms\n";
        lt = now();
    };
    // (the payloads of `splitters` and `segment-splitters` -- 50 k sorted map entries at human scale -- are serialised beside the
    // entropy stage too: nothing changes the two tables any more)
    bytes_t v_splitters, v_segspl;
    size_t n_segspl = 0;
    std::future<void> tables = std::async(std::launch::async, [&] {
        auto a32 = [](bytes_t &d, uint32_t x) {
            for (int i = 0; i < 4; ++i, x >>= 8)
                d.push_back((uint8_t)(x & 0xff));
        };
        auto a64 = [](bytes_t &d, uint64_t x) {
            for (int i = 0; i < 8; ++i, x >>= 8)
                d.push_back((uint8_t)(x & 0xff));
        };
        v_splitters.reserve(I.splitters.size() * 8);
        for (uint64_t x : I.splitters) // sorted
            a64(v_splitters, x);
        std::vector<std::pair<pk_t, int32_t>> ms;
        ms.reserve(I.map_segments.size());
        I.map_segments.for_each([&](const pk_t &k, int32_t v) { ms.emplace_back(k, v); });
        std::sort(ms.begin(), ms.end());
        v_segspl.reserve(ms.size() * 20);
        for (auto &x : ms) {
            a64(v_segspl, x.first.first);
            a64(v_segspl, x.first.second);
            a32(v_segspl, (uint32_t)x.second);
        }
        n_segspl = ms.size();
    });
    I.finish_groups();
    I.z_wait_all();
    if (open_batch.valid())
        open_batch.get();
    tables.get();
    LAP("finish_groups (pack jobs + entropy stage + parts) || open collection batch");
    if (I.verify_bad.load()) { // (AGC_AMD_VERIFY_DEV_FRAMES)
        I.err("Close: " + std::to_string(I.verify_bad.load()) + " of " + std::to_string(I.verify_frames.load()) + " device frames differ from libzstd");
        return false;
    }
    I.ar.flush_out_buffers();
    LAP("flush_out_buffers");

    auto app32 = [](bytes_t &d, uint32_t x) {
        for (int i = 0; i < 4; ++i, x >>= 8)
            d.push_back((uint8_t)(x & 0xff));
    };
    auto appstr = [](bytes_t &d, const std::string &s) {
        d.insert(d.end(), s.begin(), s.end());
        d.push_back(0);
    };
    bytes_t v;
    app32(v, I.k);
    app32(v, I.mml);
    app32(v, I.pack_cardinality);
    app32(v, I.segment_size);
    I.ar.add_part(I.ar.register_stream("params"), v, 0);

    I.ar.add_part(I.ar.register_stream("splitters"), v_splitters, I.splitters.size());
    I.ar.add_part(I.ar.register_stream("segment-splitters"), v_segspl, n_segspl);

    LAP("params / splitters / segment-splitters");
    I.coll.complete_serialization();
    LAP("collection complete_serialization");

    // m_file_type_info (agc_compressor.cpp:53-59, std::map order).  The reference's own values are
    // written so that archives stay byte-identical to its output; AGC_AMD_PRODUCER_TAG=1 tags the
    // producer honestly instead (archives then differ in this one stream).
    std::map<std::string, std::string> info;
    const bool tag = getenv("AGC_AMD_PRODUCER_TAG") != nullptr;
    info["producer"] = tag ? "agc_amd" : "agc";
    info["producer_version_major"] = "3";
    info["producer_version_minor"] = "2";
    info["producer_version_build"] = "20260326.1";
    info["file_version_major"] = "3";
    info["file_version_minor"] = "0";
    info["comment"] = tag ? "agc_amd (MI355X-native create path), archive format of AGC v. 3.2"
                          : "AGC (Assembled Genomes Compressor) v. 3.2.2 [build 20260326.1]";
    if (I.appending) // load_file_type_info replaced the defaults with the input archive's (agc_basic.cpp:53-90)
        info = I.in_file_type_info;
    v.clear();
    for (auto &x : info) {
        appstr(v, x.first);
        appstr(v, x.second);
    }
    I.ar.add_part(I.ar.register_stream("file_type_info"), v, info.size());
    I.ar.close();
    LAP("archive close");
    I.st.archive_bytes = I.ar.bytes_written();
    I.created = false;
    if (I.ar.failed()) {
        I.err("Error: writing the archive failed (disk full or I/O error): the output is incomplete");
        return false;
    }
    return true;
}

} // namespace agc
