// compressor.cpp -- see compressor.h.  Citations: file:line under the reference tree.
#include "compressor.h"
#include "host_support.h"
#include "archive_read.h"
#include "reader.h"
#include "../../../include/agc_hip.h"

#include <chrono>
#include <deque>
#include <cmath>
#include <iostream>
#include <numeric>
#include <set>
#include <zlib.h>

namespace agc {

namespace {

using pk_t = std::pair<uint64_t, uint64_t>;
constexpr uint64_t NO_KMER = ~0ULL;
constexpr uint32_t NO_RAW_GROUPS = 16; // agc_basic.h:81

struct PairHash {
    size_t operator()(const pk_t &x) const noexcept
    {
        uint64_t h = x.first * 0x9E3779B97F4A7C15ULL;
        h ^= (h >> 32) ^ (x.second * 0xC2B2AE3D27D4EB4FULL);
        return (size_t)(h ^ (h >> 29));
    }
};

// time spent inside the device library (kernels + copies + syncs), for the stage breakdown of -v 1
#define DEVT_(stats, call) ([&] { const double t_ = now(); const int r_ = (call); (stats).t_device += now() - t_; return r_; }())
#define DEVT(call) DEVT_(st, call)
#define DEVTI(call) DEVT_(I.st, call)
#define DEVTP(call) DEVT_(p->st, call)

// (kmer1, kmer2) -> group id: flat open-addressing table (one cache line per lookup instead of a node chase)
class PkMap {
    struct Slot {
        uint64_t a, b;
        int32_t v;
        uint32_t used;
    };
    std::vector<Slot> t;
    size_t n = 0, mask = 0;
    void grow()
    {
        std::vector<Slot> old;
        old.swap(t);
        t.assign(old.empty() ? 1024 : old.size() * 2, Slot{0, 0, 0, 0});
        mask = t.size() - 1;
        for (const Slot &s : old)
            if (s.used) {
                size_t i = PairHash()(pk_t{s.a, s.b}) & mask;
                while (t[i].used)
                    i = (i + 1) & mask;
                t[i] = s;
            }
    }

public:
    size_t size() const { return n; }
    void clear()
    {
        t.clear();
        n = mask = 0;
    }
    int32_t *find(const pk_t &k)
    {
        if (t.empty())
            return nullptr;
        for (size_t i = PairHash()(k) & mask;; i = (i + 1) & mask) {
            Slot &s = t[i];
            if (!s.used)
                return nullptr;
            if (s.a == k.first && s.b == k.second)
                return &s.v;
        }
    }
    int32_t &operator[](const pk_t &k)
    {
        if (int32_t *p = find(k))
            return *p;
        if ((n + 1) * 2 > t.size())
            grow();
        size_t i = PairHash()(k) & mask;
        while (t[i].used)
            i = (i + 1) & mask;
        t[i] = Slot{k.first, k.second, 0, 1};
        ++n;
        return t[i].v;
    }
    template <typename F> void for_each(F f) const
    {
        for (const Slot &s : t)
            if (s.used)
                f(pk_t{s.a, s.b}, s.v);
    }
};

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// CKmer in canonical mode (src/core/kmer.h): both forms left-aligned
struct Kmer {
    uint64_t dir = 0, rc = 0;
    bool full = false;
    uint64_t data() const { return dir < rc ? dir : rc; }   // kmer.h:350-357
    bool is_dir_oriented() const { return dir <= rc; }      // kmer.h:545-551
    void swap_dir_rc() { std::swap(dir, rc); }              // kmer.h:554-562
};

struct Contig {
    std::string sample, name;
    uint64_t off = 0, len = 0; // inside the batch's device buffer
    uint32_t sample_idx = 0;   // position of its sample inside the speculation window (0 for single-sample batches)
};

struct Seg { // one segment as compress_contig cuts it (agc_compressor.cpp:2007-2048)
    uint32_t ctg;
    uint64_t start; // relative to the contig
    uint32_t len;
    Kmer front, back;
    // classification (add_segment, agc_compressor.cpp:1275-1499)
    pk_t pk{NO_KMER, NO_KMER};
    bool store_rc = false;
    // one-splitter search
    uint32_t cand_begin = 0, cand_end = 0;
    bool back_only = false;
    Kmer one_kmer;
    // missing-middle search
    int32_t mid_job = -1;
    int32_t known_gid = -2; // group of pk when classification looked it up (-1: not there, -2: not looked up)
    uint32_t bp = 0;        // split position of a missing-middle job (before the k+1 clamps)
    Kmer kmer1, kmer2;
    bool use_rc = false;
    uint64_t middle = NO_KMER;
};

struct Cand { // find_cand_segment_with_one_splitter, agc_compressor.cpp:1660-1690
    pk_t pk;
    bool use_rc;
    uint32_t gid;
    uint64_t ref_size;
};

struct Placed { // one entry of CBufferedSegPart (agc_compressor.h:27-536)
    uint32_t key = 0; // 2 * segment index + part: stable across re-placements of the same window
    uint32_t ctg;
    uint64_t off; // absolute offset in the device buffer
    uint32_t len;
    uint32_t part_no;
    bool rc;
    int32_t gid; // -1: new group
    pk_t pk;
};

struct Group { // CSegment, write side (src/common/segment.{h,cpp})
    bool exists = false;
    uint64_t ref_size = 0; // s.size() + 1 once the reference is set (segment.cpp:46)
    uint32_t no_seqs = 0;
    // current pack, already in stored form: every sequence followed by the 0xFF separator
    // (store_in_archive(pack), segment.h:258-280); *_off[i] = start of sequence i
    bytes_t lzp_data, raw_data;
    std::vector<uint32_t> lzp_off, raw_off;
    int stream_ref = -1, stream_delta = -1;
    // append mode: a group taken over from the input archive stays "packed" until its first add in this session
    // (CSegment::appending_init / unpack, segment.cpp:418-471, 496-577).  While packed it behaves as the
    // reference's does: ref_size == 0, so Estimate answers 0 and the cost vector is empty (segment.cpp:85-86, 103-104).
    bool packed = false;
    const uint8_t *pk_ref = nullptr, *pk_delta = nullptr; // parts inside the mapped input archive
    uint64_t pk_ref_size = 0, pk_ref_meta = 0, pk_delta_size = 0, pk_delta_meta = 0;

    static void push(bytes_t &data, std::vector<uint32_t> &off, const uint8_t *b, size_t n)
    {
        off.push_back((uint32_t)data.size());
        data.insert(data.end(), b, b + n);
        data.push_back(0xff);
    }
    // index of an equal sequence in the current pack or -1 (std::find over v_lzp, segment.cpp:66)
    static int find(const bytes_t &data, const std::vector<uint32_t> &off, const uint8_t *b, size_t n)
    {
        for (size_t i = 0; i < off.size(); ++i) {
            const size_t e = (i + 1 < off.size() ? off[i + 1] : data.size()) - 1; // without the separator
            if (e - off[i] == n && memcmp(data.data() + off[i], b, n) == 0)
                return (int)i;
        }
        return -1;
    }
};

struct SampleLists { // one registration: the items of every group it touches (CSR)
    std::vector<uint32_t> gids;  // groups touched, in order of first appearance
    std::vector<uint32_t> begin; // list li = items[begin[li] .. begin[li + 1])
    std::vector<uint32_t> items; // indices into placed, per group in (contig name, part) order
    size_t n_lists() const { return gids.size(); }
};

// what store_segments' bookkeeping needs about the committed registrations (filled by process_batch on the rank that
// classified them, or rebuilt from a commit record on the other ranks of a multi-GPU job)
struct CommitData {
    const std::vector<Contig> *ctgs = nullptr;
    const std::vector<Placed> *placed = nullptr;
    uint32_t commit_upto = 0;
    std::vector<SampleLists> per_sample;
    std::vector<uint32_t> new_ref_items, raw_items, enc_items; // placed indices
    std::vector<uint8_t> repetitive;                           // per new_ref_items entry (segment.h:224-247)
    const bytes_t *fetched = nullptr;                          // new references, then raw items
    std::vector<uint64_t> fetched_off;
    std::vector<const uint8_t *> enc_ptr;                      // delta of every enc_items entry
    std::vector<uint32_t> enc_len;
    uint32_t sample_from = 0;                                  // the registrations [sample_from, commit_upto) of the window
};

struct ZJob { // one archive part to produce
    int stream_id;
    int kind;          // 0 = reference (tuples/zstd13 or zstd19), 1 = pack (zstd17)
    bytes_t data;      // raw bytes (reference symbols or concatenated pack)
    bool repetitive = false;
    bytes_t out;
    uint64_t meta = 0;
};

// bytes2tuples, src/common/segment.h:73-138
void bytes2tuples(const bytes_t &v, bytes_t &out)
{
    uint8_t me = 0;
    for (uint8_t c : v)
        me = std::max(me, c);
    uint32_t nb, mult;
    if (me < 4) {
        nb = 4;
        mult = 4;
    } else if (me < 6) {
        nb = 3;
        mult = 6;
    } else if (me < 16) {
        nb = 2;
        mult = 16;
    } else {
        out = v;
        out.push_back(0x10u);
        return;
    }
    out.clear();
    out.reserve(v.size() / nb + 2);
    size_t i = 0;
    for (; i + nb <= v.size(); i += nb) {
        uint8_t c = 0;
        for (uint32_t j = 0; j < nb; ++j)
            c = (uint8_t)(c * mult + v[i + j]);
        out.push_back(c);
    }
    uint8_t c = 0;
    for (; i < v.size(); ++i)
        c = (uint8_t)(c * mult + v[i]);
    out.push_back(c);
    out.push_back((uint8_t)((nb << 4) + (v.size() % nb)));
}

// cnv_num, src/common/agc_basic.h:40-50; preprocess_raw_contig, agc_compressor.cpp:907-951
struct CnvTable {
    uint8_t t[256];
    CnvTable()
    {
        for (int c = 0; c < 256; ++c)
            t[c] = 30;
        t[64] = t[96] = 32;
        const char *named = "ACGTNRYSWKMBDHVU";
        for (int i = 0; named[i]; ++i) {
            t[(int)named[i]] = (uint8_t)i;
            t[(int)named[i] + 32] = (uint8_t)i;
        }
        for (int c = 128; c < 256; ++c)
            t[c] = t[c & 127];
    }
};
const CnvTable g_cnv;

void preprocess_raw_contig(bytes_t &ctg)
{
    size_t o = 0;
    for (size_t i = 0; i < ctg.size(); ++i) {
        uint8_t c = ctg[i];
        if (c >> 6)
            ctg[o++] = g_cnv.t[c];
    }
    ctg.resize(o);
}

// FASTA(.gz) reader with the reference's framing (src/core/genome_io.cpp:208-252): id = first
// line minus its first character, body = every byte up to the next '>'.
class FastaReader {
    gzFile f = nullptr;
    std::vector<uint8_t> buf;
    size_t pos = 0, filled = 0;
    bool fill()
    {
        pos = 0;
        int r = gzread(f, buf.data(), (unsigned)buf.size());
        filled = r > 0 ? (size_t)r : 0;
        return filled != 0;
    }

public:
    bool open(const std::string &fn)
    {
        f = gzopen(fn.c_str(), "rb");
        if (!f)
            return false;
        gzbuffer(f, 1 << 20);
        buf.resize(16 << 20);
        pos = filled = 0;
        return true;
    }
    void close()
    {
        if (f)
            gzclose(f);
        f = nullptr;
    }
    ~FastaReader() { close(); }
    bool read_contig_raw(std::string &id, bytes_t &ctg)
    {
        id.clear();
        ctg.clear();
        if (!f)
            return false;
        for (;;) {
            if (pos >= filled && !fill())
                return false;
            uint8_t c = buf[pos++];
            if (c == '\n' || c == '\r')
                break;
            id.push_back((char)c);
        }
        if (!id.empty())
            id.erase(id.begin());
        for (;;) {
            if (pos >= filled && !fill())
                break;
            const uint8_t *b = buf.data() + pos, *e = buf.data() + filled;
            const uint8_t *q = (const uint8_t *)memchr(b, '>', (size_t)(e - b));
            if (q) {
                ctg.insert(ctg.end(), b, q);
                pos = (size_t)(q - buf.data());
                break;
            }
            ctg.insert(ctg.end(), b, e);
            pos = filled;
        }
        return !id.empty() && !ctg.empty();
    }
};

// rolling canonical k-mer on the host (reference preprocessing only)
struct HostKmer {
    uint64_t dir = 0, rc = 0;
    uint32_t cur = 0, k;
    explicit HostKmer(uint32_t k_) : k(k_) {}
    void reset() { dir = rc = 0, cur = 0; }
    void insert(uint64_t s)
    {
        const uint32_t shift = 64 - 2 * k;
        const uint64_t mask = (~0ULL) << shift;
        rc >>= 2;
        rc += (3 - s) << 62;
        rc &= mask;
        if (cur == k) {
            dir <<= 2;
            dir += s << shift;
        } else {
            ++cur;
            dir += s << (64 - 2 * cur);
        }
    }
    bool full() const { return cur == k; }
    uint64_t data() const { return dir < rc ? dir : rc; }
};

} // namespace

// ---------------------------------------------------------------------------
// Host pieces of the adaptive mode's find_new_splitters (agc_compressor.cpp:2054-2081, 630-704,
// 762-825); the reference genome itself is preprocessed on the GPU (agc_hip_determine_splitters_dev).
// ---------------------------------------------------------------------------
// splitters of one contig given the sorted candidate k-mers (find_splitters_in_contig, :762-825)
static void find_splitters_in_contig(const bytes_t &c, uint32_t k, uint32_t segment_size, const std::vector<uint64_t> &cand,
                                     std::vector<uint64_t> &spl)
{
    auto is_cand = [&](uint64_t d) { return std::binary_search(cand.begin(), cand.end(), d); };
    HostKmer h(k);
    uint64_t current_len = segment_size;
    size_t recent_from = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        uint8_t x = c[i];
        if (x > 3)
            h.reset();
        else {
            h.insert(x);
            if (h.full() && current_len >= segment_size && is_cand(h.data())) {
                spl.push_back(h.data());
                current_len = 0;
                h.reset();
                recent_from = i + 1;
            }
        }
        ++current_len;
    }
    HostKmer t(k);
    bool have = false;
    uint64_t best = 0;
    for (size_t i = recent_from; i < c.size(); ++i) {
        uint8_t x = c[i];
        if (x > 3) {
            t.reset();
            continue;
        }
        t.insert(x);
        if (t.full() && is_cand(t.data())) {
            best = t.data();
            have = true;
        }
    }
    if (have)
        spl.push_back(best);
}

static void enumerate_kmers(const bytes_t &c, uint32_t k, std::vector<uint64_t> &km)
{
    HostKmer h(k);
    for (uint8_t x : c) {
        if (x > 3)
            h.reset();
        else {
            h.insert(x);
            if (h.full())
                km.push_back(h.data());
        }
    }
}

// sorted input -> singletons in place, duplicated values (once each) appended to dup when given
static void split_singletons(std::vector<uint64_t> &km, std::vector<uint64_t> *dup)
{
    size_t o = 0;
    for (size_t i = 0; i < km.size();) {
        size_t j = i + 1;
        while (j < km.size() && km[j] == km[i])
            ++j;
        if (j == i + 1)
            km[o++] = km[i];
        else if (dup)
            dup->push_back(km[i]);
        i = j;
    }
    km.resize(o);
}

// ===========================================================================
struct CAGCCompressor::Impl {
    int device = 0;
    agc_hip_ctx *hip = nullptr;
    ZstdApi zstd;
    std::unique_ptr<ThreadPool> pool;
    std::vector<std::unique_ptr<ZstdCtx>> zctx;

    bool created = false;
    uint32_t pack_cardinality = 50, k = 31, segment_size = 60000, mml = 20, verbosity = 0;
    bool concatenated = false, adaptive = false;

    ArchiveWriter ar;
    CollectionV3 coll;
    std::vector<uint64_t> splitters;

    PkMap map_segments;                                                       // agc_compressor.h:628
    std::unordered_map<uint64_t, std::vector<uint64_t>> terminators;          // agc_compressor.h:629
    std::vector<Group> groups;                                                // v_segments
    uint32_t no_segments = 0;
    uint32_t processed_samples = 0, stored_samples = 0;
    size_t cnt_contigs_in_sample = 0;

    CompressorStats st;

    // append mode
    bool appending = false;
    rd::Archive in_ar;
    rd::ZstdD zd;
    std::map<std::string, std::string> in_file_type_info;
    bool unpack_group(uint32_t gid);

    void err(const std::string &m) { std::cerr << m << std::endl; }
    bool hip_ok(int rc, const char *what)
    {
        if (rc == AGC_HIP_OK)
            return true;
        err(std::string(what) + ": " + agc_hip_last_error(hip) + " (code " + std::to_string(rc) + ")");
        return false;
    }

    // -----------------------------------------------------------------------
    // classifies all contigs (one or several consecutive samples) against the current state and commits the
    // leading samples whose classification is certainly valid; n_committed = number of samples done
    bool process_batch(std::vector<Contig> &ctgs, const uint8_t *d_base, const std::vector<bytes_t> *host_data, uint32_t &n_committed);
    struct BatchState { // working set of one process_batch call
        std::vector<Contig> *ctgs = nullptr;
        const uint8_t *d_base = nullptr;
        const std::vector<bytes_t> *host_data = nullptr;
        uint32_t n_ctg = 0;
        double t0 = 0, dev0 = 0, lap_t = 0;
        std::vector<uint64_t> new_splitters_added; // adaptive mode
        std::vector<uint32_t> subset;              // segments stage_classify works on
        uint32_t n_samples = 1, s_from = 0;        // registrations of the window; first one not committed yet
        struct Spec {                              // speculative delta of a placed item (by Placed::key)
            uint64_t off = 0, enc_off = 0;
            uint32_t gid = 0, len = 0, enc_len = 0;
            bool rc = false, valid = false;
        };
        std::vector<Spec> spec;
        std::vector<uint64_t> changed;             // k-mers whose terminator list changed in the last commit run
        uint32_t commit_upto = 0;                  // registrations of the window that are committed now
        std::vector<uint32_t> order;               // committed items in registration order
        std::vector<SampleLists> per_sample;
    };
    bool stage_scan(BatchState &b);
    bool stage_classify(BatchState &b);
    bool stage_place(BatchState &b);
    bool stage_register(BatchState &b);
    bool stage_store(BatchState &b);
    bool spec_encode(BatchState &b);
    bool revalidate(BatchState &b);
    bool batch_prepare(BatchState &b, std::vector<Contig> &ctgs, const uint8_t *d_base, const std::vector<bytes_t> *host_data, bool always_speculate);
    bool batch_commit(BatchState &b, uint32_t &n_committed);
    // a sample classified ahead of its turn (multi-GPU mode, PrepareSampleDevice): its working set, and the k-mers whose
    // terminator lists changed since (through other ranks' records)
    std::unique_ptr<BatchState> prepared;
    std::vector<Contig> prepared_ctgs;
    std::vector<uint64_t> changed_log;
    void lap(BatchState &b, const char *what);
    bool book_and_store(CommitData &cd);
    // stage accounting: wall time of the stage and its host-only part (wall minus the time inside the device library)
    void stage_end(double &wall, double &host_only, double &t0, double &dev0)
    {
        const double t = now();
        wall += t - t0;
        host_only += (t - t0) - (st.t_device - dev0);
        t0 = t;
        dev0 = st.t_device;
    }
    // multi-GPU single-archive mode: one registration at a time, committed on every rank from the owner's record
    uint32_t dist_rank = 0, dist_world = 1, dist_writer = 0;
    bytes_t dist_record;
    void make_record(const CommitData &cd, const std::vector<uint64_t> &new_splitters);
    bool apply_record(const uint8_t *rec, size_t n, const uint8_t *d_rec);
    void note_new_group(const pk_t &pk, uint32_t gid);
    void finish_groups();
    void run_jobs(std::vector<ZJob> &jobs, bool add_parts = true);
    void add_job_parts(std::vector<ZJob> &jobs, size_t from, size_t to);
    void make_pack_job(std::vector<ZJob> &jobs, Group &g, bytes_t &data, std::vector<uint32_t> &off);
    bytes_t enc_buf, enc_buf2, fetch_buf; // grown, never shrunk (enc_buf: the window's speculative deltas, enc_buf2: per commit run)
    // scratch reused across registrations (no reallocation / page faults / zero fill per sample)
    std::vector<uint32_t> gid_slot, gid_epoch;
    uint32_t gid_epoch_ctr = 0;
    std::vector<uint32_t> scan_ctg;
    std::vector<uint64_t> scan_pos, scan_dir, scan_rc;
    std::vector<Seg> seg_buf;
    std::vector<Placed> placed_buf;
    // adaptive mode (-a): sorted singleton / duplicated k-mers of the reference genome
    // (v_candidate_kmers / v_duplicated_kmers, agc_compressor.cpp:493-497)
    std::vector<uint64_t> ref_singletons, ref_duplicates;
    bool find_new_splitters(const bytes_t &ctg, std::vector<uint64_t> &out);
    int scan_batch(const std::vector<uint64_t> &ctg_off, uint32_t n_ctg, const uint8_t *d_base, std::vector<uint32_t> &h_ctg,
                   std::vector<uint64_t> &h_pos, std::vector<uint64_t> &h_dir, std::vector<uint64_t> &h_rc, uint64_t &n_hits);
    void after_registration();
};

CAGCCompressor::CAGCCompressor() : p(new Impl) {}
CAGCCompressor::~CAGCCompressor()
{
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
const std::vector<uint8_t> &CAGCCompressor::LastRecord() const { return p->dist_record; }
bool CAGCCompressor::ApplyRecord(const uint8_t *record, size_t n, const uint8_t *d_record)
{
    if (!p->created || p->dist_world < 2 || p->appending || p->concatenated)
        return false;
    return p->apply_record(record, n, d_record);
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
    if (!I.zstd.load(e)) {
        I.err(e);
        return false;
    }
    if (!I.hip) {
        int rc = agc_hip_create(&I.hip, I.device);
        if (rc != AGC_HIP_OK) {
            I.err("no HIP device: the MI355X path has no CPU fallback (agc_hip_create = " + std::to_string(rc) + ")");
            return false;
        }
    }
    unsigned nt = std::max(1u, no_threads);
    I.pool.reset(new ThreadPool(nt));
    for (unsigned i = 0; i < nt; ++i)
        I.zctx.emplace_back(new ZstdCtx(&I.zstd));
    I.created = true;

    if (!reference_file_name.empty()) {
        FastaReader fr;
        if (!fr.open(reference_file_name)) {
            I.err("Cannot open file: " + reference_file_name);
            I.created = false;
            return false;
        }
        std::vector<bytes_t> ref;
        std::string id;
        bytes_t c;
        uint64_t tot = 0;
        while (fr.read_contig_raw(id, c)) {
            preprocess_raw_contig(c);
            tot += c.size();
            ref.emplace_back(std::move(c));
            c.clear();
        }
        // determine_splitters on the GPU: contigs go to HBM back to back, k-mers are enumerated, radix
        // sorted and reduced to singletons there (include/agc_hip.h: agc_hip_determine_splitters_dev)
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
    if (!I.in_ar.open(in_archive_name)) {
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
    for (unsigned i = 0; i < nt; ++i)
        I.zctx.emplace_back(new ZstdCtx(&I.zstd));

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

// ---------------------------------------------------------------------------
// store_in_archive(pack), segment.h:258-280: sequences + 0xFF separators -> zstd 17
void CAGCCompressor::Impl::make_pack_job(std::vector<ZJob> &jobs, Group &g, bytes_t &data, std::vector<uint32_t> &off)
{
    ZJob j;
    if (g.stream_delta < 0) // segment.h:262-266
        g.stream_delta = ar.register_stream(ss_delta_name((uint32_t)(&g - groups.data())));
    j.stream_id = g.stream_delta;
    j.kind = 1;
    j.data.swap(data);
    data.clear();
    off.clear();
    jobs.emplace_back(std::move(j));
}

// add_to_archive / add_to_archive_tuples, segment.h:172-215; store_in_archive(ref) :218-255
void CAGCCompressor::Impl::run_jobs(std::vector<ZJob> &jobs, bool add_parts)
{
    double t0 = now();
    pool->parallel_for(jobs.size(), [&](size_t i, unsigned tid) {
        ZJob &j = jobs[i];
        ZstdCtx &z = *zctx[tid];
        const bytes_t *src = &j.data;
        bytes_t tuples;
        int level = 17;
        uint8_t marker = 0;
        if (j.kind == 0) {
            if (!j.repetitive) {
                bytes2tuples(j.data, tuples);
                src = &tuples;
                level = 13;
                marker = 1;
            } else
                level = 19;
        }
        size_t bound = zstd.compressBound(src->size());
        bytes_t packed(bound + 1);
        uint32_t ps = (uint32_t)z.compress(packed.data(), bound, src->data(), src->size(), level);
        packed[ps] = marker;
        if (ps + 1u < (uint32_t)j.data.size()) {
            packed.resize((size_t)ps + 1);
            j.out = std::move(packed);
            j.meta = j.data.size();
        } else {
            j.out = j.data;
            j.meta = 0;
        }
    });
    st.t_zstd += now() - t0;
    if (add_parts)
        add_job_parts(jobs, 0, jobs.size());
}

// hands finished parts to the archive buffer, in job order (= insertion order inside every stream)
void CAGCCompressor::Impl::add_job_parts(std::vector<ZJob> &jobs, size_t from, size_t to)
{
    for (size_t i = from; i < to; ++i) {
        ZJob &j = jobs[i];
        st.zstd_in += j.data.size();
        st.zstd_out += j.out.size();
        ar.add_part_buffered(j.stream_id, std::move(j.out), j.meta);
    }
}

// the tail of the registration token handling, agc_compressor.cpp:1136-1180
void CAGCCompressor::Impl::after_registration()
{
    if (!concatenated)
        ++processed_samples;
    else {
        processed_samples = processed_samples / pack_cardinality * pack_cardinality + pack_cardinality;
        uint32_t max_ps = (uint32_t)coll.no_samples();
        if (max_ps < processed_samples)
            processed_samples = max_ps;
    }
    if (processed_samples % pack_cardinality == 0) {
        coll.store_contig_batch(processed_samples - pack_cardinality, processed_samples);
        stored_samples = processed_samples;
    }
    ar.flush_out_buffers();
}

// ---------------------------------------------------------------------------
int CAGCCompressor::Impl::scan_batch(const std::vector<uint64_t> &ctg_off, uint32_t n_ctg, const uint8_t *d_base,
                                     std::vector<uint32_t> &h_ctg, std::vector<uint64_t> &h_pos, std::vector<uint64_t> &h_dir,
                                     std::vector<uint64_t> &h_rc, uint64_t &n_hits)
{
    uint64_t cap = std::max<uint64_t>(4096, (ctg_off[n_ctg] - ctg_off[0]) / 1000);
    for (;;) {
        if (h_ctg.size() < cap) { // the buffers only grow (they are reused by every registration)
            h_ctg.resize(cap);
            h_pos.resize(cap);
            h_dir.resize(cap);
            h_rc.resize(cap);
        }
        cap = h_ctg.size();
        int rc = DEVT(agc_hip_scan_contigs_dev(hip, d_base, ctg_off.data(), n_ctg, k, cap, &n_hits, h_ctg.data(), h_pos.data(), h_dir.data(),
                                               h_rc.data()));
        if (rc == AGC_HIP_ECAP) {
            cap = n_hits;
            continue;
        }
        if (!hip_ok(rc, "scan_contigs"))
            return rc;
        return AGC_HIP_OK;
    }
}

// find_new_splitters, agc_compressor.cpp:2054-2081: singleton k-mers of the contig that occur nowhere
// in the reference genome are the candidates
bool CAGCCompressor::Impl::find_new_splitters(const bytes_t &ctg, std::vector<uint64_t> &out)
{
    std::vector<uint64_t> km, tmp;
    enumerate_kmers(ctg, k, km);
    std::sort(km.begin(), km.end());
    split_singletons(km, nullptr);
    tmp.resize(km.size());
    auto e = std::set_difference(km.begin(), km.end(), ref_singletons.begin(), ref_singletons.end(), tmp.begin());
    tmp.erase(e, tmp.end());
    km.resize(tmp.size());
    e = std::set_difference(tmp.begin(), tmp.end(), ref_duplicates.begin(), ref_duplicates.end(), km.begin());
    km.erase(e, km.end());
    find_splitters_in_contig(ctg, k, segment_size, km, out);
    return true;
}

// One call = one pass over a window of registrations: scan -> classification -> placement -> [speculative encode] ->
// commit runs (registration + store, revalidation in between).
// The stages share their working set through BatchState (and the reusable seg_buf / placed_buf scratch of Impl).
bool CAGCCompressor::Impl::process_batch(std::vector<Contig> &ctgs, const uint8_t *d_base, const std::vector<bytes_t> *host_data,
                                         uint32_t &n_committed)
{
    n_committed = 0;
    BatchState b;
    if (!batch_prepare(b, ctgs, d_base, host_data, false))
        return false;
    return batch_commit(b, n_committed);
}

// first half: everything that only reads the classification state
bool CAGCCompressor::Impl::batch_prepare(BatchState &b, std::vector<Contig> &ctgs, const uint8_t *d_base, const std::vector<bytes_t> *host_data,
                                         bool always_speculate)
{
    b.ctgs = &ctgs;
    b.d_base = d_base;
    b.host_data = host_data;
    b.n_ctg = (uint32_t)ctgs.size();
    b.t0 = now();
    b.dev0 = st.t_device;
    b.lap_t = b.t0;
    if (!stage_scan(b))
        return false;
    b.n_samples = ctgs.empty() ? 1u : ctgs.back().sample_idx + 1;
    b.subset.resize(seg_buf.size());
    std::iota(b.subset.begin(), b.subset.end(), 0u);
    if (!stage_classify(b) || !stage_place(b))
        return false;
    // several registrations in the window (or a sample prepared ahead of its turn): everything that can be encoded already
    // (group known and stored) is, in one batch
    if ((b.n_samples > 1 || always_speculate) && !spec_encode(b))
        return false;
    ++st.windows;
    return true;
}

// second half: the commit runs
bool CAGCCompressor::Impl::batch_commit(BatchState &b, uint32_t &n_committed)
{
    for (b.s_from = 0;;) {
        ++st.commit_runs;
        if (!stage_register(b) || !stage_store(b))
            return false;
        n_committed = b.commit_upto;
        // append / adaptive mode: the caller classifies the rest again (unpacked groups and new splitters change more than
        // the dependencies revalidate() follows; their windows hold one registration anyway)
        if (b.commit_upto >= b.n_samples || appending || adaptive)
            break;
        b.s_from = b.commit_upto;
        if (!revalidate(b))
            return false;
    }
    return true;
}

// Up-front LZ encode of the window's items whose group already has its reference: group references never change
// (segment.cpp:41-48), so these deltas stay valid whatever earlier registrations of the window mint.
bool CAGCCompressor::Impl::spec_encode(BatchState &b)
{
    const std::vector<Placed> &placed = placed_buf;
    double &t0 = b.t0, &dev0 = b.dev0;
    b.spec.assign(2 * seg_buf.size(), BatchState::Spec());
    std::vector<uint32_t> items;
    for (uint32_t i = 0; i < placed.size(); ++i)
        if (placed[i].gid >= (int32_t)NO_RAW_GROUPS && groups[placed[i].gid].exists && !groups[placed[i].gid].packed)
            items.push_back(i);
    if (items.empty())
        return true;
    const size_t ne = items.size();
    std::vector<uint32_t> gid(ne), len(ne);
    std::vector<uint64_t> off(ne), eoff(ne + 1, 0);
    std::vector<uint8_t> rc(ne);
    uint64_t tot = 0;
    for (size_t i = 0; i < ne; ++i) {
        const Placed &pl = placed[items[i]];
        gid[i] = (uint32_t)pl.gid;
        off[i] = pl.off;
        len[i] = pl.len;
        rc[i] = pl.rc;
        tot += pl.len;
    }
    bytes_t &enc = enc_buf;
    uint64_t cap = std::max<uint64_t>(enc.size(), tot / 64 + (1u << 20));
    for (;;) {
        if (enc.size() < cap)
            enc.resize(cap);
        int r = DEVT(agc_hip_lz_encode_batch_dev(hip, (uint32_t)ne, gid.data(), b.d_base, off.data(), len.data(), rc.data(), enc.data(), cap, eoff.data()));
        if (r == AGC_HIP_ECAP) {
            cap = eoff[ne] + 64;
            continue;
        }
        if (!hip_ok(r, "lz_encode_batch"))
            return false;
        break;
    }
    for (size_t i = 0; i < ne; ++i) {
        const Placed &pl = placed[items[i]];
        BatchState::Spec &sp = b.spec[pl.key];
        sp.valid = true;
        sp.gid = gid[i];
        sp.off = pl.off;
        sp.len = pl.len;
        sp.rc = pl.rc;
        sp.enc_off = eoff[i];
        sp.enc_len = (uint32_t)(eoff[i + 1] - eoff[i]);
    }
    st.lz_encoded += ne;
    st.delta_bytes += eoff[ne];
    stage_end(st.t_encode, st.h_encode, t0, dev0);
    return true;
}

// After a commit run that minted groups: the not yet committed segments whose decision read what has changed are
// classified again against the current state -- exactly what processing the registrations one after the other would
// have seen.  Dependencies of a decision (add_segment, agc_compressor.cpp:1275-1499):
//   both splitters, key known          -> none (a key never leaves the map, its group never changes)
//   both splitters, key unknown        -> the key itself (minted meanwhile?) and the terminator lists of its two k-mers
//                                         (missing-middle search, :1502-1535)
//   one splitter                       -> the terminator list of that k-mer (candidate groups, :1640-1690)
//   no splitter                        -> none
bool CAGCCompressor::Impl::revalidate(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    std::vector<Seg> &segs = seg_buf;
    std::sort(b.changed.begin(), b.changed.end());
    b.changed.erase(std::unique(b.changed.begin(), b.changed.end()), b.changed.end());
    auto changed = [&](uint64_t kmer) { return std::binary_search(b.changed.begin(), b.changed.end(), kmer); };
    b.subset.clear();
    for (uint32_t si = 0; si < segs.size(); ++si) {
        const Seg &s = segs[si];
        if (ctgs[s.ctg].sample_idx < b.s_from)
            continue;
        const bool ff = s.front.full, bf = s.back.full;
        if (ff != bf) {
            if (changed(s.one_kmer.data()))
                b.subset.push_back(si);
        } else if (ff && bf && !concatenated && s.known_gid == -1) {
            if (changed(s.front.data()) || changed(s.back.data()) || map_segments.find(std::minmax(s.front.data(), s.back.data())))
                b.subset.push_back(si);
        }
    }
    st.revalidated += b.subset.size();
    if (!b.subset.empty() && !stage_classify(b))
        return false;
    return stage_place(b); // cheap, and picks up keys that are in the map by now (known_gid < 0 is looked up again)
}

// AGC_AMD_LAPS=1: wall time of every host sub-stage of a registration on stderr (profiling aid)
void CAGCCompressor::Impl::lap(BatchState &b, const char *what)
{
    static const bool laps = getenv("AGC_AMD_LAPS") != nullptr;
    if (!laps)
        return;
    std::cerr << "  lap " << what << " " << (now() - b.lap_t) * 1e3 << " ms\n";
    b.lap_t = now();
}

// compress_contig for every contig of the window: splitter hits from the GPU, adaptive-mode re-scan, segments
bool CAGCCompressor::Impl::stage_scan(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint8_t *d_base = b.d_base;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)ctgs; (void)d_base; (void)n_ctg; (void)t0; (void)dev0; (void)LAP;
    const std::vector<bytes_t> *host_data = b.host_data;
    std::vector<uint64_t> &new_splitters_added = b.new_splitters_added;
    // ---- stage 1a: splitter scan on the GPU (compress_contig's loop) ----
    std::vector<uint64_t> ctg_off(n_ctg + 1, 0);
    for (uint32_t i = 0; i < n_ctg; ++i) {
        ctg_off[i] = ctgs[i].off;
        st.bases += ctgs[i].len;
    }
    if (n_ctg)
        ctg_off[n_ctg] = ctgs.back().off + ctgs.back().len;
    for (uint32_t i = 0; i + 1 < n_ctg; ++i)
        if (ctgs[i].off + ctgs[i].len != ctgs[i + 1].off) {
            err("internal: contigs of a batch must be contiguous in HBM");
            return false;
        }
    std::vector<uint32_t> &h_ctg = scan_ctg;
    std::vector<uint64_t> &h_pos = scan_pos, &h_dir = scan_dir, &h_rc = scan_rc;
    uint64_t n_hits = 0;
    if (n_ctg && scan_batch(ctg_off, n_ctg, d_base, h_ctg, h_pos, h_dir, h_rc, n_hits) != AGC_HIP_OK)
        return false;

    // ---- adaptive mode: contigs without any splitter look for new ones, the set is extended and
    // those contigs are scanned again (agc_compressor.cpp:2038-2044, 2054-2081, 1187-1237) ----
    if (adaptive && n_ctg) {
        std::vector<uint8_t> has_hit(n_ctg, 0);
        for (uint64_t h = 0; h < n_hits; ++h)
            has_hit[h_ctg[h]] = 1;
        std::vector<uint32_t> deferred;
        for (uint32_t c = 0; c < n_ctg; ++c)
            if (!has_hit[c])
                deferred.push_back(c);
        if (!deferred.empty()) {
            // contigs long enough to carry a splitter: their symbols are needed on the host
            std::vector<uint32_t> need;
            for (uint32_t c : deferred)
                if (ctgs[c].len >= segment_size)
                    need.push_back(c);
            std::vector<bytes_t> fetched_ctg(need.size());
            if (!need.empty() && !host_data) {
                std::vector<uint64_t> off(need.size()), ooff(need.size() + 1);
                std::vector<uint32_t> len(need.size());
                uint64_t tot = 0;
                for (size_t i = 0; i < need.size(); ++i) {
                    off[i] = ctgs[need[i]].off;
                    len[i] = (uint32_t)ctgs[need[i]].len;
                    tot += len[i];
                }
                bytes_t buf(tot);
                if (!hip_ok(DEVT(agc_hip_fetch_slices_dev(hip, (uint32_t)need.size(), d_base, off.data(), len.data(), nullptr, buf.data(), tot, ooff.data())), "fetch_slices"))
                    return false;
                for (size_t i = 0; i < need.size(); ++i)
                    fetched_ctg[i].assign(buf.begin() + ooff[i], buf.begin() + ooff[i + 1]);
            }
            std::vector<std::vector<uint64_t>> found(need.size());
            pool->parallel_for(need.size(), [&](size_t i, unsigned) {
                find_new_splitters(host_data ? (*host_data)[need[i]] : fetched_ctg[i], found[i]);
            });
            size_t n_new = 0;
            for (auto &f : found)
                n_new += f.size();
            if (n_new) {
                std::vector<uint64_t> add;
                for (auto &f : found)
                    add.insert(add.end(), f.begin(), f.end());
                splitters.insert(splitters.end(), add.begin(), add.end());
                std::sort(splitters.begin(), splitters.end());
                splitters.erase(std::unique(splitters.begin(), splitters.end()), splitters.end());
                if (!hip_ok(DEVT(agc_hip_splitters_insert(hip, add.data(), add.size())), "splitters_insert"))
                    return false;
                new_splitters_added = add;
                // second scan with the extended set; only the deferred contigs take its hits
                std::vector<uint32_t> c2;
                std::vector<uint64_t> p2, d2, r2;
                uint64_t n2 = 0;
                if (scan_batch(ctg_off, n_ctg, d_base, c2, p2, d2, r2, n2) != AGC_HIP_OK)
                    return false;
                std::vector<uint32_t> mc;
                std::vector<uint64_t> mp, md, mr;
                uint64_t a = 0, b = 0;
                for (uint32_t c = 0; c < n_ctg; ++c) {
                    while (a < n_hits && h_ctg[a] < c)
                        ++a;
                    while (b < n2 && c2[b] < c)
                        ++b;
                    if (has_hit[c])
                        for (; a < n_hits && h_ctg[a] == c; ++a) {
                            mc.push_back(c);
                            mp.push_back(h_pos[a]);
                            md.push_back(h_dir[a]);
                            mr.push_back(h_rc[a]);
                        }
                    else
                        for (; b < n2 && c2[b] == c; ++b) {
                            mc.push_back(c);
                            mp.push_back(p2[b]);
                            md.push_back(d2[b]);
                            mr.push_back(r2[b]);
                        }
                }
                h_ctg.swap(mc);
                h_pos.swap(mp);
                h_dir.swap(md);
                h_rc.swap(mr);
                n_hits = h_ctg.size();
            }
        }
    }
    stage_end(st.t_scan, st.h_scan, t0, dev0);
    t0 = now();

    LAP("scan");
    // ---- stage 1b: cut into segments (agc_compressor.cpp:2018-2048) ----
    std::vector<Seg> &segs = seg_buf;
    segs.clear();
    segs.reserve(n_hits + n_ctg);
    {
        uint64_t h = 0;
        for (uint32_t c = 0; c < n_ctg; ++c) {
            uint64_t split_pos = 0;
            Kmer split_kmer;
            while (h < n_hits && h_ctg[h] == c) {
                Seg s;
                s.ctg = c;
                s.start = split_pos;
                s.len = (uint32_t)(h_pos[h] + 1 - split_pos);
                s.front = split_kmer;
                s.back.dir = h_dir[h];
                s.back.rc = h_rc[h];
                s.back.full = true;
                segs.push_back(s);
                split_pos = h_pos[h] + 1 - k;
                split_kmer = s.back;
                ++h;
            }
            if (split_pos < ctgs[c].len) {
                Seg s;
                s.ctg = c;
                s.start = split_pos;
                s.len = (uint32_t)(ctgs[c].len - split_pos);
                s.front = split_kmer;
                segs.push_back(s);
            }
        }
    }

    return true;
}

// add_segment for all segments at once: keys, one-splitter candidates (estimates on the GPU), missing-middle split points
bool CAGCCompressor::Impl::stage_classify(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint8_t *d_base = b.d_base;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)ctgs; (void)d_base; (void)n_ctg; (void)t0; (void)dev0; (void)LAP;
    std::vector<Seg> &segs = seg_buf;
    LAP("cut");
    // the segments to classify: all of the window, or the ones whose decision read state that changed since (revalidate)
    const std::vector<uint32_t> &L = b.subset;
    // ---- stage 1c: add_segment, part 1: keys and one-splitter candidates ----
    std::vector<Cand> cands;
    for (uint32_t si : L) {
        Seg &s = segs[si];
        s.pk = {NO_KMER, NO_KMER};
        s.store_rc = false;
        s.cand_begin = s.cand_end = 0;
        s.back_only = false;
        s.mid_job = -1;
        s.known_gid = -2;
        s.use_rc = false;
        s.middle = NO_KMER;
        s.bp = 0;
        const bool ff = s.front.full, bf = s.back.full;
        if (!ff && !bf) {
            s.pk = {NO_KMER, NO_KMER}; // agc_compressor.cpp:1286-1301 (fallback filter off)
        } else if (ff && bf) {
            if (s.front.data() < s.back.data())
                s.pk = {s.front.data(), s.back.data()};
            else {
                s.pk = {s.back.data(), s.front.data()};
                s.store_rc = true;
            }
        } else {
            s.back_only = !ff;
            s.one_kmer = ff ? s.front : s.back;
            if (s.back_only)
                s.one_kmer.swap_dir_rc(); // :1339-1340
            s.cand_begin = (uint32_t)cands.size();
            auto t = terminators.find(s.one_kmer.data());
            if (t != terminators.end()) {
                for (uint64_t ck : t->second) {
                    Cand c;
                    if (ck < s.one_kmer.data()) {
                        c.pk = {ck, s.one_kmer.data()};
                        c.use_rc = true;
                    } else {
                        c.pk = {s.one_kmer.data(), ck};
                        c.use_rc = false;
                    }
                    const int32_t *m = map_segments.find(c.pk);
                    if (!m) {
                        err("internal: terminator without group");
                        return false;
                    }
                    c.gid = (uint32_t)*m;
                    c.ref_size = groups[c.gid].ref_size;
                    cands.push_back(c);
                }
                // stable_sort by |segment_size - ref_size|, then ref_size (:1681-1690)
                const int64_t ssz = (int64_t)s.len;
                std::stable_sort(cands.begin() + s.cand_begin, cands.end(), [ssz](const Cand &x, const Cand &y) {
                    int64_t xs = (int64_t)x.ref_size, ys = (int64_t)y.ref_size;
                    int64_t ax = std::llabs(ssz - xs), ay = std::llabs(ssz - ys);
                    if (ax != ay)
                        return ax < ay;
                    return xs < ys;
                });
            }
            s.cand_end = (uint32_t)cands.size();
            ++st.one_splitter;
        }
    }
    stage_end(st.t_classify, st.h_classify, t0, dev0);
    t0 = now();

    LAP("keys");
    // ---- GPU: estimates for every (one-splitter segment, candidate) pair ----
    std::vector<uint32_t> est_cost(cands.size()), est_peak(cands.size());
    if (!cands.empty()) {
        // candidates without a reference in HBM (append mode: still packed) answer 0 on the host below
        std::vector<uint32_t> gid, len, which;
        std::vector<uint64_t> off;
        std::vector<uint8_t> rc;
        for (uint32_t si : L) {
            const Seg &s = segs[si];
            for (uint32_t c = s.cand_begin; c < s.cand_end; ++c) {
                if (cands[c].ref_size == 0)
                    continue;
                which.push_back(c);
                gid.push_back(cands[c].gid);
                off.push_back(ctgs[s.ctg].off + s.start);
                len.push_back(s.len);
                // front-only: segment_dir = the segment itself; back-only: segment_dir = its reverse complement (:1317-1345)
                rc.push_back((uint8_t)(s.back_only ? !cands[c].use_rc : cands[c].use_rc));
            }
        }
        std::vector<uint32_t> cost(which.size()), peak(which.size());
        if (!which.empty() &&
            !hip_ok(DEVT(agc_hip_lz_estimate_batch_dev(hip, (uint32_t)which.size(), gid.data(), d_base, off.data(), len.data(), rc.data(), cost.data(),
                                                  peak.data())),
                    "lz_estimate_batch"))
            return false;
        for (size_t i = 0; i < which.size(); ++i) {
            est_cost[which[i]] = cost[i];
            est_peak[which[i]] = peak[i];
        }
    }
    stage_end(st.t_gpu_aux, st.h_gpu_aux, t0, dev0);
    t0 = now();

    LAP("estimates");
    // ---- add_segment, part 2: resolve one-splitter keys (:1630-1808) ----
    for (uint32_t si : L) {
        Seg &s = segs[si];
        if (s.front.full == s.back.full)
            continue;
        const Kmer &kmer = s.one_kmer;
        pk_t best_pk{NO_KMER, NO_KMER};
        bool is_best_rc = false;
        uint64_t best_estim = s.len < 16 ? s.len : s.len - 16u;
        const uint32_t nc = s.cand_end - s.cand_begin;
        std::vector<uint64_t> v_est(nc);
        for (uint32_t i = 0; i < nc; ++i) {
            const uint32_t c = s.cand_begin + i;
            // CSegment::estimate returns 0 for a group without reference (segment.cpp:85-86)
            uint64_t e;
            if (groups[cands[c].gid].ref_size == 0)
                e = 0;
            else if ((uint64_t)est_peak[c] > (uint32_t)best_estim)
                e = ~0ULL; // the bounded call returned early with a value > bound: never selected
            else
                e = est_cost[c];
            v_est[i] = e;
            if (e < best_estim)
                best_estim = e;
        }
        for (uint32_t i = 0; i < nc; ++i) {
            const Cand &c = cands[s.cand_begin + i];
            if (v_est[i] < best_estim || (v_est[i] == best_estim && c.pk < best_pk) ||
                (v_est[i] == best_estim && c.pk == best_pk && !c.use_rc)) {
                best_estim = v_est[i];
                best_pk = c.pk;
                is_best_rc = c.use_rc;
            }
        }
        if (best_pk == pk_t{NO_KMER, NO_KMER}) {
            if (kmer.is_dir_oriented())
                best_pk = {kmer.data(), NO_KMER};
            else {
                best_pk = {NO_KMER, kmer.data()};
                is_best_rc = true;
            }
        }
        s.pk = best_pk;
        s.store_rc = s.back_only ? !is_best_rc : is_best_rc;
    }

    LAP("resolve");
    // ---- add_segment, part 3: missing-middle-splitter candidates (:1366-1459, 1502-1627) ----
    struct MidJob {
        uint32_t seg, gid1, gid2;
        uint8_t rc1, pf1, rc2, pf2;
    };
    std::vector<MidJob> mids;
    for (uint32_t si : L) {
        Seg &s = segs[si];
        if (concatenated || s.pk.first == NO_KMER || s.pk.second == NO_KMER)
            continue;
        if (const int32_t *m = map_segments.find(s.pk)) { // known group: remembered for the placement below
            s.known_gid = *m;
            continue;
        }
        s.known_gid = -1;
        auto tf = terminators.find(s.pk.first), tb = terminators.find(s.pk.second);
        if (tf == terminators.end() || tb == terminators.end())
            continue;
        if (s.front.data() == s.back.data()) {
            if (!s.front.is_dir_oriented())
                s.store_rc = true;
            continue;
        }
        s.kmer1 = s.front;
        s.kmer2 = s.back;
        s.use_rc = false;
        if (s.kmer1.data() > s.kmer2.data()) {
            std::swap(s.kmer1, s.kmer2);
            s.use_rc = true;
            s.kmer1.swap_dir_rc();
            s.kmer2.swap_dir_rc();
        }
        auto p_front = terminators.find(s.kmer1.data()), p_back = terminators.find(s.kmer2.data());
        std::vector<uint64_t> shared;
        std::set_intersection(p_front->second.begin(), p_front->second.end(), p_back->second.begin(), p_back->second.end(),
                              std::back_inserter(shared));
        shared.erase(std::remove(shared.begin(), shared.end(), NO_KMER), shared.end());
        ++st.middle_tried;
        if (shared.empty())
            continue;
        s.middle = shared.front();
        const int32_t *m1 = map_segments.find(std::minmax(s.kmer1.data(), s.middle)), *m2 = map_segments.find(std::minmax(s.middle, s.kmer2.data()));
        if (!m1 || !m2) {
            err("internal: shared terminator without group");
            return false;
        }
        MidJob j;
        j.seg = si;
        j.gid1 = (uint32_t)*m1;
        j.gid2 = (uint32_t)*m2;
        {
            // a group without reference leaves its cost vector empty (segment.cpp:103-104; append mode: still packed):
            // one empty vector -> sizes differ -> no split (:1604-1607); both empty -> best_pos = 0 -> left part empty
            const bool e1 = groups[j.gid1].ref_size == 0, e2 = groups[j.gid2].ref_size == 0;
            if (e1 != e2) {
                s.middle = NO_KMER;
                continue;
            }
            if (e1) {
                s.mid_job = -2;
                continue;
            }
        }
        // segment_dir here = use_rc ? rc(segment) : segment (:1394)
        const bool f_lt_m = s.kmer1.data() < s.middle, m_lt_b = s.middle < s.kmer2.data();
        j.rc1 = (uint8_t)(f_lt_m ? s.use_rc : !s.use_rc);
        j.pf1 = f_lt_m ? 1 : 0;
        j.rc2 = (uint8_t)(m_lt_b ? s.use_rc : !s.use_rc);
        j.pf2 = m_lt_b ? 0 : 1;
        s.mid_job = (int32_t)mids.size();
        mids.push_back(j);
    }
    stage_end(st.t_classify, st.h_classify, t0, dev0);
    t0 = now();
    LAP("mids");
    std::vector<uint32_t> best_pos(mids.size(), 0);
    if (!mids.empty()) {
        size_t n = mids.size();
        std::vector<uint32_t> g1(n), g2(n), len(n);
        std::vector<uint64_t> off(n);
        std::vector<uint8_t> r1(n), p1(n), r2(n), p2(n);
        for (size_t i = 0; i < n; ++i) {
            const Seg &s = segs[mids[i].seg];
            g1[i] = mids[i].gid1;
            g2[i] = mids[i].gid2;
            off[i] = ctgs[s.ctg].off + s.start;
            len[i] = s.len;
            r1[i] = mids[i].rc1;
            p1[i] = mids[i].pf1;
            r2[i] = mids[i].rc2;
            p2[i] = mids[i].pf2;
        }
        if (!hip_ok(DEVT(agc_hip_lz_split_point_batch_dev(hip, (uint32_t)n, g1.data(), g2.data(), d_base, off.data(), len.data(), r1.data(),
                                                     p1.data(), r2.data(), p2.data(), best_pos.data(), nullptr)),
                    "lz_split_point_batch"))
            return false;
    }
    for (size_t i = 0; i < mids.size(); ++i)
        segs[mids[i].seg].bp = best_pos[i];
    stage_end(st.t_gpu_aux, st.h_gpu_aux, t0, dev0);
    t0 = now();

    return true;
}

// add_segment, last part: the placed items (one or two per segment) with their part numbers
bool CAGCCompressor::Impl::stage_place(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint8_t *d_base = b.d_base;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)ctgs; (void)d_base; (void)n_ctg; (void)t0; (void)dev0; (void)LAP;
    std::vector<Seg> &segs = seg_buf;
    LAP("splitpoints");
    // ---- add_segment, part 4: final placement + part numbers ----
    std::vector<Placed> &placed = placed_buf;
    placed.clear();
    placed.reserve(segs.size() + segs.size() / 8 + 16);
    {
        uint32_t cur_ctg = ~0u, part_no = 0;
        for (uint32_t si = 0; si < segs.size(); ++si) {
            const Seg &s = segs[si];
            pk_t pk = s.pk;            // (placement never writes to the segment: it is repeated after a revalidation)
            bool store_rc = s.store_rc;
            if (s.ctg != cur_ctg) {
                cur_ctg = s.ctg;
                part_no = 0;
            }
            const uint64_t abs_off = ctgs[s.ctg].off + s.start;
            bool two = false;
            Placed a, b;
            a.ctg = b.ctg = s.ctg;
            a.key = 2 * si;
            b.key = 2 * si + 1;
            if (s.mid_job >= 0 || s.mid_job == -2) {
                uint32_t bp = s.mid_job >= 0 ? s.bp : 0;
                if (bp < k + 1u)
                    bp = 0;
                if (s.mid_job >= 0 && (size_t)bp + k + 1u > s.len)
                    bp = s.len;
                uint32_t left = bp, right = s.len - bp;
                if (left == 0) {
                    store_rc = (s.middle < s.kmer2.data()) ? s.use_rc : !s.use_rc;
                    pk = std::minmax(s.middle, s.kmer2.data());
                } else if (right == 0) {
                    store_rc = (s.kmer1.data() < s.middle) ? s.use_rc : !s.use_rc;
                    pk = std::minmax(s.kmer1.data(), s.middle);
                } else {
                    if (s.use_rc)
                        std::swap(left, right);
                    const uint32_t seg2_start = left - k / 2;
                    two = true;
                    ++st.middle_split;
                    // first part: [0, seg2_start + k)
                    a.off = abs_off;
                    a.len = seg2_start + k;
                    if (s.front.data() < s.middle) {
                        a.rc = false;
                        a.pk = {s.front.data(), s.middle};
                    } else {
                        a.rc = true;
                        a.pk = {s.middle, s.front.data()};
                    }
                    // second part: [seg2_start, len)
                    b.off = abs_off + seg2_start;
                    b.len = s.len - seg2_start;
                    if (s.middle < s.back.data()) {
                        b.rc = false;
                        b.pk = {s.middle, s.back.data()};
                    } else {
                        b.rc = true;
                        b.pk = {s.back.data(), s.middle};
                    }
                    const int32_t *ma = map_segments.find(a.pk), *mb = map_segments.find(b.pk);
                    if (!ma || !mb) {
                        err("internal: split target group missing");
                        return false;
                    }
                    a.gid = *ma;
                    b.gid = *mb;
                }
            }
            if (two) {
                a.part_no = part_no;
                b.part_no = part_no + 1;
                placed.push_back(a);
                placed.push_back(b);
                part_no += 2;
            } else {
                a.off = abs_off;
                a.len = s.len;
                a.rc = store_rc;
                a.pk = pk;
                if (s.known_gid >= 0 && s.mid_job == -1)
                    a.gid = s.known_gid; // looked up during classification, key unchanged since
                else {
                    const int32_t *m = map_segments.find(pk);
                    a.gid = m ? *m : -1;
                }
                a.part_no = part_no++;
                placed.push_back(a);
            }
        }
    }

    return true;
}

// register_segments: what is committed, in which order, with which (new) group ids
bool CAGCCompressor::Impl::stage_register(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint8_t *d_base = b.d_base;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)ctgs; (void)d_base; (void)n_ctg; (void)t0; (void)dev0; (void)LAP;
    std::vector<Placed> &placed = placed_buf;
    LAP("placement");
    // ---- speculation window: the contigs may belong to several consecutive samples that were
    // all classified against the SAME state.  State changes only when a sample mints a new group (or, in append mode,
    // unpacks one), so the classification is valid for every sample up to and including the first one that does; this
    // COMMIT RUN takes the registrations [s_from, commit_upto).  What comes after it is revalidated (process_batch).
    const uint32_t n_samples = b.n_samples, s_from = b.s_from;
    uint32_t &commit_upto = b.commit_upto;
    commit_upto = n_samples; // exclusive
    for (const Placed &pl : placed) {
        const uint32_t sx = ctgs[pl.ctg].sample_idx;
        if (sx >= s_from && (pl.gid < 0 || groups[pl.gid].packed) && sx + 1 < commit_upto)
            commit_upto = sx + 1;
    }

    // ---- register_segments per sample (agc_compressor.cpp:954-971; agc_compressor.h:384-435) ----
    // order of CBufferedSegPart's lists and of the std::set of new parts: (sample name, contig name,
    // part no) (agc_compressor.h:112-120, 157-164).  Contigs are ranked once, items sort on integers;
    // samples keep their processing order (each one is a registration of its own).
    std::vector<uint32_t> ctg_rank(n_ctg);
    {
        std::vector<uint32_t> co(n_ctg);
        std::iota(co.begin(), co.end(), 0u);
        auto cless = [&](uint32_t x, uint32_t y) {
            if (ctgs[x].sample_idx != ctgs[y].sample_idx)
                return ctgs[x].sample_idx < ctgs[y].sample_idx;
            if (ctgs[x].sample != ctgs[y].sample)
                return ctgs[x].sample < ctgs[y].sample;
            return ctgs[x].name < ctgs[y].name;
        };
        std::stable_sort(co.begin(), co.end(), cless);
        uint32_t r = 0;
        for (uint32_t i = 0; i < n_ctg; ++i) {
            if (i && cless(co[i - 1], co[i]))
                ++r;
            ctg_rank[co[i]] = r;
        }
    }
    LAP("ctg_rank");
    std::vector<uint32_t> &order = b.order; // committed items only, in (sample, contig name, part) order
    {
        std::vector<std::pair<uint64_t, uint32_t>> keyed;
        keyed.reserve(placed.size());
        for (uint32_t i = 0; i < placed.size(); ++i)
            if (ctgs[placed[i].ctg].sample_idx >= s_from && ctgs[placed[i].ctg].sample_idx < commit_upto)
                keyed.push_back({((uint64_t)ctg_rank[placed[i].ctg] << 32) | placed[i].part_no, i});
        if (!std::is_sorted(keyed.begin(), keyed.end()))
            std::sort(keyed.begin(), keyed.end());
        order.resize(keyed.size());
        for (size_t i = 0; i < keyed.size(); ++i)
            order[i] = keyed[i].second;
        st.segments += order.size();
    }
    {
    LAP("order");
        // new group ids in that order (only the last committed sample can have new items)
        std::map<pk_t, uint32_t> m_kmers;
        uint32_t gid = no_segments;
        for (uint32_t idx : order)
            if (placed[idx].gid < 0) {
                auto it = m_kmers.find(placed[idx].pk);
                if (it == m_kmers.end())
                    it = m_kmers.emplace(placed[idx].pk, gid++).first;
                placed[idx].gid = (int32_t)it->second;
            }
        const uint32_t no_new = gid - no_segments;
        for (uint32_t i = 0; i < no_new; ++i) {
            groups.emplace_back();
            Group &g = groups.back();
            g.stream_ref = ar.register_stream(ss_ref_name(no_segments + i));
            g.stream_delta = ar.register_stream(ss_delta_name(no_segments + i));
        }
        no_segments += no_new;
        st.new_groups += no_new;
    }
    LAP("newgids");
    // per sample: lists of items per group, raw groups by distribute_segments(0, 0, 16) on the sorted
    // list of group 0 (agc_compressor.h:417-435)
    std::vector<SampleLists> &per_sample = b.per_sample;
    per_sample.assign(commit_upto - s_from, SampleLists());
    {
        size_t pos = 0;
        for (uint32_t sidx = s_from; sidx < commit_upto; ++sidx) {
            size_t end = pos;
            while (end < order.size() && ctgs[placed[order[end]].ctg].sample_idx == sidx)
                ++end;
            std::vector<uint32_t> raw0;
            for (size_t i = pos; i < end; ++i)
                if (placed[order[i]].gid == 0)
                    raw0.push_back(order[i]);
            const size_t n0 = raw0.size();
            const size_t n_moved = n0 - (n0 + 15) / 16;
            for (size_t j = 0; j < n0; ++j)
                placed[raw0[j]].gid = j < n_moved ? (int32_t)(1 + (j % 15)) : 0;
            SampleLists &sl = per_sample[sidx - s_from];
            // slot of every group touched by this registration (epoch-stamped scratch instead of a hash map)
            if (gid_slot.size() < groups.size()) {
                gid_slot.resize(groups.size() + groups.size() / 4 + 64, 0);
                gid_epoch.resize(gid_slot.size(), 0);
            }
            ++gid_epoch_ctr;
            std::vector<uint32_t> cnt;
            for (size_t i = pos; i < end; ++i) {
                const uint32_t gid = (uint32_t)placed[order[i]].gid;
                if (gid_epoch[gid] != gid_epoch_ctr) {
                    gid_epoch[gid] = gid_epoch_ctr;
                    gid_slot[gid] = (uint32_t)sl.gids.size();
                    sl.gids.push_back(gid);
                    cnt.push_back(0);
                }
                ++cnt[gid_slot[gid]];
            }
            sl.begin.assign(sl.gids.size() + 1, 0);
            for (size_t li = 0; li < sl.gids.size(); ++li)
                sl.begin[li + 1] = sl.begin[li] + cnt[li];
            sl.items.resize(end - pos);
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (size_t i = pos; i < end; ++i) {
                const uint32_t li = gid_slot[(uint32_t)placed[order[i]].gid];
                sl.items[sl.begin[li] + cnt[li]++] = order[i];
            }
            pos = end;
        }
    }
    stage_end(st.t_register, st.h_register, t0, dev0);
    t0 = now();

    return true;
}

// store_segments, first half: new references into HBM, LZ-encode of everything else, then the bookkeeping stage
bool CAGCCompressor::Impl::stage_store(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint8_t *d_base = b.d_base;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)ctgs; (void)d_base; (void)n_ctg; (void)t0; (void)dev0; (void)LAP;
    std::vector<Placed> &placed = placed_buf;
    const uint32_t commit_upto = b.commit_upto;
    std::vector<SampleLists> &per_sample = b.per_sample;
    std::vector<uint64_t> &new_splitters_added = b.new_splitters_added;
    LAP("per_sample");
    // ---- store_segments (agc_compressor.cpp:974-1050) ----
    // (a) what each item needs: new groups' first item becomes the reference (segment.cpp:39-48), raw
    // groups keep the symbols, everything else is LZ-encoded -- decided per group across the committed samples
    std::vector<uint32_t> new_ref_items; // placed indices, one per new group with items
    std::vector<uint32_t> raw_items;
    std::vector<uint32_t> enc_items;
    {
        std::vector<uint8_t> will_exist(groups.size(), 0);
        for (uint32_t sidx = 0; sidx < per_sample.size(); ++sidx)
            for (size_t li = 0; li < per_sample[sidx].n_lists(); ++li) {
                const uint32_t gid = per_sample[sidx].gids[li];
                for (uint32_t ii = per_sample[sidx].begin[li]; ii < per_sample[sidx].begin[li + 1]; ++ii) {
                    const uint32_t idx = per_sample[sidx].items[ii];
                    if (gid < NO_RAW_GROUPS)
                        raw_items.push_back(idx);
                    else if (!groups[gid].exists && !will_exist[gid]) {
                        new_ref_items.push_back(idx);
                        will_exist[gid] = 1;
                    } else
                        enc_items.push_back(idx);
                }
            }
    }
    LAP("classes");
    // append mode: the first add to a group of the input archive unpacks it (segment.cpp:19-20, 39-40)
    if (appending)
        for (uint32_t sidx = 0; sidx < per_sample.size(); ++sidx)
            for (uint32_t gid : per_sample[sidx].gids)
                if (groups[gid].packed && !unpack_group(gid))
                    return false;
    // map_segments / terminators updates happen when a group is first stored (:1003-1028)
    b.changed.clear();
    for (uint32_t idx : new_ref_items) {
        note_new_group(placed[idx].pk, (uint32_t)placed[idx].gid);
        if (placed[idx].pk.first != NO_KMER && placed[idx].pk.second != NO_KMER) { // terminator lists that gained an entry
            b.changed.push_back(placed[idx].pk.first);
            b.changed.push_back(placed[idx].pk.second);
        }
    }
    // GPU: register the new references (index build) and pull back what the host must pack
    std::vector<uint32_t> lag_cnt, lag_cur;
    std::vector<uint8_t> repetitive;
    bytes_t &fetched = fetch_buf;
    std::vector<uint64_t> fetched_off;
    {
        const size_t nr = new_ref_items.size();
        if (nr) {
            std::vector<uint32_t> gid(nr), len(nr);
            std::vector<uint64_t> off(nr);
            std::vector<uint8_t> rc(nr);
            for (size_t i = 0; i < nr; ++i) {
                const Placed &pl = placed[new_ref_items[i]];
                gid[i] = (uint32_t)pl.gid;
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
                st.ref_bytes += pl.len;
            }
            if (!hip_ok(DEVT(agc_hip_ref_register_batch_dev(hip, (uint32_t)nr, gid.data(), d_base, off.data(), len.data(), rc.data(), mml)), "ref_register_batch"))
                return false;
            lag_cnt.resize(nr * 28);
            lag_cur.resize(nr * 28);
            if (!hip_ok(DEVT(agc_hip_ref_lag_counts_dev(hip, (uint32_t)nr, d_base, off.data(), len.data(), rc.data(), lag_cnt.data(), lag_cur.data())), "ref_lag_counts"))
                return false;
            // repetitiveness probe with the reference's double arithmetic (segment.h:224-247)
            repetitive.resize(nr);
            for (size_t fi = 0; fi < nr; ++fi) {
                double best_frac = 0.0;
                for (uint32_t l = 0; l < 28; ++l) {
                    const uint32_t cnt = lag_cnt[fi * 28 + l], cur = lag_cur[fi * 28 + l];
                    double frac = 0.0;
                    if (cur)
                        frac = (double)cnt / cur;
                    if (frac > best_frac) {
                        best_frac = frac;
                        if (best_frac >= 0.5)
                            break;
                    }
                }
                repetitive[fi] = !(best_frac < 0.5);
            }
        }
        const size_t nf = nr + raw_items.size();
        if (nf) {
            std::vector<uint32_t> len(nf);
            std::vector<uint64_t> off(nf);
            std::vector<uint8_t> rc(nf);
            uint64_t tot = 0;
            for (size_t i = 0; i < nf; ++i) {
                const Placed &pl = placed[i < nr ? new_ref_items[i] : raw_items[i - nr]];
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
                tot += pl.len;
            }
            if (fetched.size() < tot)
                fetched.resize(tot);
            fetched_off.resize(nf + 1);
            if (!hip_ok(DEVT(agc_hip_fetch_slices_dev(hip, (uint32_t)nf, d_base, off.data(), len.data(), rc.data(), fetched.data(), tot, fetched_off.data())), "fetch_slices"))
                return false;
        }
    }
    stage_end(st.t_register, st.h_register, t0, dev0);
    t0 = now();
    // LZ deltas (segment.cpp:50-58): items whose group already had its reference when the window was classified were encoded
    // up front in one batch (spec_encode); only the others -- followers of a group minted in this window, segments that a
    // revalidation placed differently -- are encoded now
    std::vector<const uint8_t *> enc_ptr(enc_items.size(), nullptr);
    std::vector<uint32_t> enc_len(enc_items.size(), 0);
    {
        std::vector<uint32_t> todo; // positions in enc_items
        for (uint32_t i = 0; i < enc_items.size(); ++i) {
            const Placed &pl = placed[enc_items[i]];
            const BatchState::Spec *sp = pl.key < b.spec.size() ? &b.spec[pl.key] : nullptr;
            if (sp && sp->valid && sp->gid == (uint32_t)pl.gid && sp->off == pl.off && sp->len == pl.len && sp->rc == pl.rc) {
                enc_ptr[i] = enc_buf.data() + sp->enc_off;
                enc_len[i] = sp->enc_len;
            } else
                todo.push_back(i);
        }
        if (!todo.empty()) {
            const size_t ne = todo.size();
            std::vector<uint32_t> gid(ne), len(ne);
            std::vector<uint64_t> off(ne), eoff(ne + 1, 0);
            std::vector<uint8_t> rc(ne);
            uint64_t tot = 0;
            for (size_t i = 0; i < ne; ++i) {
                const Placed &pl = placed[enc_items[todo[i]]];
                gid[i] = (uint32_t)pl.gid;
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
                tot += pl.len;
            }
            bytes_t &enc = enc_buf2;
            uint64_t cap = std::max<uint64_t>(enc.size(), tot / 64 + (1u << 16));
            for (;;) {
                if (enc.size() < cap)
                    enc.resize(cap);
                int r = DEVT(agc_hip_lz_encode_batch_dev(hip, (uint32_t)ne, gid.data(), d_base, off.data(), len.data(), rc.data(), enc.data(), cap,
                                                         eoff.data()));
                if (r == AGC_HIP_ECAP) {
                    cap = eoff[ne] + 64;
                    continue;
                }
                if (!hip_ok(r, "lz_encode_batch"))
                    return false;
                break;
            }
            for (size_t i = 0; i < ne; ++i) {
                enc_ptr[todo[i]] = enc.data() + eoff[i];
                enc_len[todo[i]] = (uint32_t)(eoff[i + 1] - eoff[i]);
            }
            st.lz_encoded += ne;
            st.delta_bytes += eoff[ne];
        }
    }
    stage_end(st.t_encode, st.h_encode, t0, dev0);
    t0 = now();

    CommitData cdta;
    cdta.ctgs = b.ctgs;
    cdta.placed = &placed;
    cdta.commit_upto = commit_upto;
    cdta.sample_from = b.s_from;
    cdta.per_sample = std::move(per_sample);
    cdta.new_ref_items = std::move(new_ref_items);
    cdta.raw_items = std::move(raw_items);
    cdta.enc_items = std::move(enc_items);
    cdta.repetitive = std::move(repetitive);
    cdta.fetched = &fetched;
    cdta.fetched_off = std::move(fetched_off);
    cdta.enc_ptr = std::move(enc_ptr);
    cdta.enc_len = std::move(enc_len);
    if (dist_world > 1) {
        make_record(cdta, new_splitters_added);
        if (dist_rank != dist_writer)
            return true; // the writer rank does the bookkeeping from the record
    }
    return book_and_store(cdta);
}

// store_segments, second half (agc_compressor.cpp:989-1050): per-group bookkeeping, zstd parts, collection records
bool CAGCCompressor::Impl::book_and_store(CommitData &cdta)
{
    double t0 = now(), dev0 = st.t_device;
    const std::vector<Contig> &ctgs = *cdta.ctgs;
    const std::vector<Placed> &placed = *cdta.placed;
    const uint32_t n_ctg = (uint32_t)ctgs.size(), commit_upto = cdta.commit_upto;
    std::vector<SampleLists> &per_sample = cdta.per_sample;
    const std::vector<uint32_t> &new_ref_items = cdta.new_ref_items, &raw_items = cdta.raw_items, &enc_items = cdta.enc_items;
    const bytes_t &fetched = *cdta.fetched;
    const std::vector<uint64_t> &fetched_off = cdta.fetched_off;
    const uint32_t sample_from = cdta.sample_from;
    // (b) per sample, per group, in list order: CSegment::add / add_raw (segment.cpp:14-80); then the sample's
    // zstd jobs, collection records and the end-of-registration steps
    std::vector<uint32_t> pos_newref(placed.size()), pos_raw(placed.size()), pos_enc(placed.size());
    for (uint32_t i = 0; i < new_ref_items.size(); ++i)
        pos_newref[new_ref_items[i]] = i;
    for (uint32_t i = 0; i < raw_items.size(); ++i)
        pos_raw[raw_items[i]] = i;
    for (uint32_t i = 0; i < enc_items.size(); ++i)
        pos_enc[enc_items[i]] = i;
    std::vector<uint32_t> in_group_id(placed.size(), 0);
    // contig descriptors of the collection (agc_compressor.cpp:1038-1049)
    std::vector<CollectionV3::ContigDesc *> cd(n_ctg, nullptr);
    bool dup_names_in_batch = false;
    {
        std::set<CollectionV3::ContigDesc *> seen;
        for (uint32_t c = 0; c < n_ctg; ++c) {
            if (ctgs[c].sample_idx < sample_from || ctgs[c].sample_idx >= commit_upto)
                continue;
            std::string stored = ctgs[c].sample.empty() ? CollectionV3::extract_contig_name(ctgs[c].name) : ctgs[c].sample;
            CollectionV3::SampleDesc &sd = coll.sample_by_name(stored);
            for (auto &x : sd.contigs)
                if (x.name == ctgs[c].name) {
                    cd[c] = &x;
                    break;
                }
            if (cd[c] && !seen.insert(cd[c]).second)
                dup_names_in_batch = true;
        }
    }
    // zstd jobs of all committed samples are compressed together (they are independent); their parts and
    // the end-of-registration steps are then replayed sample by sample, so the archive is laid out exactly
    // as if every sample had been finished before the next one started
    std::vector<ZJob> all_jobs;
    const uint32_t n_regs = (uint32_t)per_sample.size();
    std::vector<size_t> jobs_end(n_regs, 0);
    for (uint32_t sidx = 0; sidx < n_regs; ++sidx) {
        SampleLists &sl = per_sample[sidx];
        std::vector<ZJob> jobs;
        auto book = [&](size_t li_begin, size_t li_end, std::vector<ZJob> &jobs) {
            for (size_t li = li_begin; li < li_end; ++li) {
                const uint32_t gid = sl.gids[li];
                Group &g = groups[gid];
                for (uint32_t ii = sl.begin[li]; ii < sl.begin[li + 1]; ++ii) {
                    const uint32_t idx = sl.items[ii];
                    const Placed &pl = placed[idx];
                    uint32_t igid;
                    if (gid < NO_RAW_GROUPS) {
                        if (g.raw_off.size() == pack_cardinality)
                            make_pack_job(jobs, g, g.raw_data, g.raw_off);
                        const uint32_t fi = (uint32_t)new_ref_items.size() + pos_raw[idx];
                        ++g.no_seqs;
                        Group::push(g.raw_data, g.raw_off, fetched.data() + fetched_off[fi], fetched_off[fi + 1] - fetched_off[fi]);
                        igid = g.no_seqs - 1;
                    } else if (!g.exists) {
                        g.exists = true;
                        const uint32_t fi = pos_newref[idx];
                        ZJob j;
                        j.stream_id = g.stream_ref;
                        j.kind = 0;
                        j.data.assign(fetched.begin() + fetched_off[fi], fetched.begin() + fetched_off[fi + 1]);
                        j.repetitive = cdta.repetitive[fi] != 0;
                        jobs.emplace_back(std::move(j));
                        g.ref_size = (uint64_t)pl.len + 1;
                        g.no_seqs = 1;
                        igid = 0;
                    } else {
                        if (g.lzp_off.size() == pack_cardinality)
                            make_pack_job(jobs, g, g.lzp_data, g.lzp_off);
                        const uint32_t ei = pos_enc[idx];
                        const uint8_t *dp = cdta.enc_ptr[ei];
                        const size_t dn = cdta.enc_len[ei];
                        if (dn == 0)
                            igid = 0; // same sequence as the reference (segment.cpp:60-63)
                        else {
                            const int f = Group::find(g.lzp_data, g.lzp_off, dp, dn);
                            if (f >= 0)
                                igid = g.no_seqs - (uint32_t)(g.lzp_off.size() - (size_t)f);
                            else {
                                Group::push(g.lzp_data, g.lzp_off, dp, dn);
                                ++g.no_seqs;
                                igid = g.no_seqs - 1;
                            }
                        }
                    }
                    in_group_id[idx] = igid;
                }
            }
        };
        // groups are independent of each other (the reference runs them on all worker threads,
        // agc_compressor.cpp:989-1050): big samples go to the pool in chunks, jobs merged in list order
        if (sl.n_lists() >= 4096) {
            const size_t n_chunks = std::min<size_t>(sl.n_lists(), (size_t)pool->size() * 8);
            std::vector<std::vector<ZJob>> chunk_jobs(n_chunks);
            pool->parallel_for(n_chunks, [&](size_t ci, unsigned) {
                book(sl.n_lists() * ci / n_chunks, sl.n_lists() * (ci + 1) / n_chunks, chunk_jobs[ci]);
            });
            for (auto &cj : chunk_jobs)
                for (auto &j : cj)
                    jobs.emplace_back(std::move(j));
        } else
            book(0, sl.n_lists(), jobs);
        // collection records.  Two contigs of one sample with the same name share the first one's descriptor
        // (add_segments_placed looks contigs up by name, collection_v3.cpp:806-817), so where their part numbers collide
        // the LAST write wins: the reference walks the groups from the highest id down (agc_compressor.cpp:990-996,
        // agc_compressor.h:509-520) -- done the same way here so that even such inputs come out identical
        auto place = [&](uint32_t idx) {
            const Placed &pl = placed[idx];
            auto *c = cd[pl.ctg];
            if (!c)
                return;
            if (pl.part_no >= c->segments.size())
                c->segments.resize((size_t)pl.part_no + 1);
            c->segments[pl.part_no] = {(uint32_t)pl.gid, in_group_id[idx], pl.len, pl.rc};
        };
        if (!dup_names_in_batch)
            for (uint32_t idx : sl.items)
                place(idx);
        else {
            std::vector<uint32_t> lo(sl.n_lists());
            std::iota(lo.begin(), lo.end(), 0u);
            std::sort(lo.begin(), lo.end(), [&](uint32_t a, uint32_t b) { return sl.gids[a] > sl.gids[b]; });
            for (uint32_t li : lo)
                for (uint32_t ii = sl.begin[li]; ii < sl.begin[li + 1]; ++ii)
                    place(sl.items[ii]);
        }
        for (auto &j : jobs)
            all_jobs.emplace_back(std::move(j));
        jobs_end[sidx] = all_jobs.size();
    }
    stage_end(st.t_store, st.h_store, t0, dev0);
    if (verbosity > 1)
        std::cerr << "registration: " << placed.size() << " items; host-only seconds so far: scan " << st.h_scan << " classify " << st.h_classify
                  << " register " << st.h_register << " encode " << st.h_encode << " store " << st.h_store << std::endl;
    run_jobs(all_jobs, false);
    for (uint32_t sidx = 0; sidx < n_regs; ++sidx) {
        add_job_parts(all_jobs, sidx ? jobs_end[sidx - 1] : 0, jobs_end[sidx]);
        after_registration();
    }
    return true;
}

// store_segments' update of map_segments (keep the smaller id) and of the terminator lists, agc_compressor.cpp:1003-1028
void CAGCCompressor::Impl::note_new_group(const pk_t &pk, uint32_t gid)
{
    int32_t *it = map_segments.find(pk);
    if (!it)
        map_segments[pk] = (int32_t)gid;
    else if (*it > (int32_t)gid)
        *it = (int32_t)gid;
    if (prepared && pk.first != NO_KMER && pk.second != NO_KMER) {
        changed_log.push_back(pk.first);
        changed_log.push_back(pk.second);
    }
    if (pk.first != NO_KMER && pk.second != NO_KMER) {
        auto &v1 = terminators[pk.first];
        v1.push_back(pk.second);
        std::sort(v1.begin(), v1.end());
        if (pk.first != pk.second) {
            auto &v2 = terminators[pk.second];
            v2.push_back(pk.first);
            std::sort(v2.begin(), v2.end());
        }
    }
}

// ---------------------------------------------------------------------------
// Multi-GPU single-archive mode.  Samples are dealt round-robin to the ranks; every rank keeps the
// whole classification state (splitters, (k1,k2) -> group map, terminators, references in its HBM).  The owner of a
// sample classifies and encodes it (process_batch), then publishes a COMMIT RECORD: contig names, new splitters and,
// group by group in registration order, every placed item with its payload (symbols of a new reference -- the
// "newly-minted reference segments" every GPU needs --, raw symbols, or the delta).  All other ranks apply the
// record (apply_record): same group ids, same map/terminator updates, references registered in their own HBM;
// the writer rank also runs the bookkeeping / zstd / archive stage from it.  Samples are committed strictly in
// order, so the archive equals the single-GPU one byte for byte.
// Record layout (little endian): "AGCR" | n_ctg | n_lists | n_new_splitters | first_new_gid | n_new_groups |
//   contigs: sample\0 name\0 ... | splitters u64... | lists: gid, n_items, items: ctg, part_no, len, rc, kind,
//   [pk1, pk2, repetitive for kind 0], payload_len, payload
// ---------------------------------------------------------------------------
namespace {
void put32(bytes_t &d, uint32_t x)
{
    for (int i = 0; i < 4; ++i, x >>= 8)
        d.push_back((uint8_t)(x & 0xff));
}
void put64(bytes_t &d, uint64_t x)
{
    for (int i = 0; i < 8; ++i, x >>= 8)
        d.push_back((uint8_t)(x & 0xff));
}
struct RecReader {
    const uint8_t *p, *e;
    bool ok = true;
    bool need(size_t n)
    {
        if ((size_t)(e - p) < n)
            ok = false;
        return ok;
    }
    uint32_t u32()
    {
        if (!need(4))
            return 0;
        uint32_t x = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        p += 4;
        return x;
    }
    uint64_t u64()
    {
        const uint64_t lo = u32(), hi = u32();
        return lo | (hi << 32);
    }
    uint8_t u8() { return need(1) ? *p++ : 0; }
    std::string str()
    {
        const uint8_t *q = p;
        while (q < e && *q)
            ++q;
        if (q >= e) {
            ok = false;
            return std::string();
        }
        std::string r((const char *)p, (size_t)(q - p));
        p = q + 1;
        return r;
    }
};
} // namespace

void CAGCCompressor::Impl::make_record(const CommitData &cd, const std::vector<uint64_t> &new_splitters)
{
    const std::vector<Contig> &ctgs = *cd.ctgs;
    const std::vector<Placed> &placed = *cd.placed;
    bytes_t &r = dist_record;
    r.clear();
    r.insert(r.end(), {'A', 'G', 'C', 'R'});
    put32(r, (uint32_t)ctgs.size());
    const SampleLists &sl = cd.per_sample.at(0); // one registration per record
    put32(r, (uint32_t)sl.n_lists());
    put32(r, (uint32_t)new_splitters.size());
    uint32_t first_new = ~0u, n_new = 0;
    for (uint32_t idx : cd.new_ref_items) {
        first_new = std::min(first_new, (uint32_t)placed[idx].gid);
        ++n_new;
    }
    put32(r, first_new);
    put32(r, n_new);
    for (auto &c : ctgs) {
        r.insert(r.end(), c.sample.begin(), c.sample.end());
        r.push_back(0);
        r.insert(r.end(), c.name.begin(), c.name.end());
        r.push_back(0);
    }
    for (uint64_t x : new_splitters)
        put64(r, x);
    std::vector<uint32_t> pos_newref(placed.size()), pos_raw(placed.size()), pos_enc(placed.size());
    for (uint32_t i = 0; i < cd.new_ref_items.size(); ++i)
        pos_newref[cd.new_ref_items[i]] = i;
    for (uint32_t i = 0; i < cd.raw_items.size(); ++i)
        pos_raw[cd.raw_items[i]] = i;
    for (uint32_t i = 0; i < cd.enc_items.size(); ++i)
        pos_enc[cd.enc_items[i]] = i;
    std::vector<uint8_t> kind(placed.size(), 2);
    for (uint32_t idx : cd.new_ref_items)
        kind[idx] = 0;
    for (uint32_t idx : cd.raw_items)
        kind[idx] = 1;
    for (size_t li = 0; li < sl.n_lists(); ++li) {
        put32(r, sl.gids[li]);
        put32(r, sl.begin[li + 1] - sl.begin[li]);
        for (uint32_t ii = sl.begin[li]; ii < sl.begin[li + 1]; ++ii) {
            const uint32_t idx = sl.items[ii];
            const Placed &pl = placed[idx];
            put32(r, pl.ctg);
            put32(r, pl.part_no);
            put32(r, pl.len);
            r.push_back((uint8_t)pl.rc);
            r.push_back(kind[idx]);
            const uint8_t *b;
            size_t n;
            if (kind[idx] == 0) {
                put64(r, pl.pk.first);
                put64(r, pl.pk.second);
                const uint32_t fi = pos_newref[idx];
                r.push_back(cd.repetitive[fi]);
                b = cd.fetched->data() + cd.fetched_off[fi];
                n = cd.fetched_off[fi + 1] - cd.fetched_off[fi];
            } else if (kind[idx] == 1) {
                const uint32_t fi = (uint32_t)cd.new_ref_items.size() + pos_raw[idx];
                b = cd.fetched->data() + cd.fetched_off[fi];
                n = cd.fetched_off[fi + 1] - cd.fetched_off[fi];
            } else {
                const uint32_t ei = pos_enc[idx];
                b = cd.enc_ptr[ei];
                n = cd.enc_len[ei];
            }
            put32(r, (uint32_t)n);
            r.insert(r.end(), b, b + n);
        }
    }
    // the owner keeps what later classifications read of its new groups (book_and_store does it on the writer)
    if (dist_rank != dist_writer)
        for (uint32_t idx : cd.new_ref_items) {
            Group &g = groups[(uint32_t)placed[idx].gid];
            g.exists = true;
            g.ref_size = (uint64_t)placed[idx].len + 1;
        }
}

bool CAGCCompressor::Impl::apply_record(const uint8_t *rec, size_t n, const uint8_t *d_rec)
{
    RecReader rr{rec, rec + n};
    if (n < 24 || memcmp(rec, "AGCR", 4) != 0) {
        err("bad commit record");
        return false;
    }
    rr.p += 4;
    const uint32_t n_ctg = rr.u32(), n_lists = rr.u32(), n_spl = rr.u32(), first_new = rr.u32(), n_new = rr.u32();
    std::vector<Contig> ctgs(n_ctg);
    for (auto &c : ctgs) {
        c.sample = rr.str();
        c.name = rr.str();
        c.sample_idx = 0;
    }
    std::vector<uint64_t> add(n_spl);
    for (auto &x : add)
        x = rr.u64();
    if (!rr.ok) {
        err("truncated commit record");
        return false;
    }
    const bool writer = dist_rank == dist_writer;
    if (!add.empty()) { // adaptive mode: the owner's new splitters (agc_compressor.cpp:1191-1209)
        splitters.insert(splitters.end(), add.begin(), add.end());
        std::sort(splitters.begin(), splitters.end());
        splitters.erase(std::unique(splitters.begin(), splitters.end()), splitters.end());
        if (!hip_ok(DEVT(agc_hip_splitters_insert(hip, add.data(), add.size())), "splitters_insert"))
            return false;
    }
    if (writer) {
        coll.reset_prev_sample_name();
        for (auto &c : ctgs)
            if (!coll.register_sample_contig(c.sample, c.name)) {
                err("Error: Pair sample_name:contig_name " + c.sample + ":" + c.name + " is already in the archive!");
                return false;
            }
    }
    if (n_new) {
        if (first_new != no_segments) {
            err("commit record out of order: new groups start at " + std::to_string(first_new) + ", expected " + std::to_string(no_segments));
            return false;
        }
        for (uint32_t i = 0; i < n_new; ++i) {
            groups.emplace_back();
            Group &g = groups.back();
            g.stream_ref = ar.register_stream(ss_ref_name(no_segments + i));
            g.stream_delta = ar.register_stream(ss_delta_name(no_segments + i));
        }
        no_segments += n_new;
        st.new_groups += n_new;
    }
    std::vector<Placed> placed;
    CommitData cd;
    cd.commit_upto = 1;
    cd.per_sample.resize(1);
    SampleLists &sl = cd.per_sample[0];
    bytes_t refs_raw, raws, enc;
    std::vector<uint64_t> ref_off{0}, raw_off{0}, enc_off{0};
    std::vector<uint32_t> reg_gid, reg_len;
    std::vector<uint64_t> reg_off; // payload offsets inside the record (device copy)
    for (uint32_t li = 0; li < n_lists && rr.ok; ++li) {
        const uint32_t gid = rr.u32(), cnt = rr.u32();
        sl.gids.push_back(gid);
        sl.begin.push_back((uint32_t)sl.items.size());
        for (uint32_t i = 0; i < cnt && rr.ok; ++i) {
            Placed pl;
            pl.ctg = rr.u32();
            pl.part_no = rr.u32();
            pl.len = rr.u32();
            pl.rc = rr.u8() != 0;
            const uint8_t kind = rr.u8();
            pl.gid = (int32_t)gid;
            pl.off = 0;
            uint8_t rep = 0;
            if (kind == 0) {
                pl.pk.first = rr.u64();
                pl.pk.second = rr.u64();
                rep = rr.u8();
            }
            const uint32_t pn = rr.u32();
            if (!rr.need(pn) || pl.ctg >= n_ctg || gid >= groups.size())
                break;
            const uint32_t idx = (uint32_t)placed.size();
            if (kind == 0) {
                cd.new_ref_items.push_back(idx);
                cd.repetitive.push_back(rep);
                refs_raw.insert(refs_raw.end(), rr.p, rr.p + pn);
                ref_off.push_back(refs_raw.size());
                reg_gid.push_back(gid);
                reg_len.push_back(pn);
                reg_off.push_back((uint64_t)(rr.p - rec));
                note_new_group(pl.pk, gid);
                groups[gid].exists = !writer; // the writer's bookkeeping turns it on (first item = reference)
                groups[gid].ref_size = (uint64_t)pn + 1;
            } else if (kind == 1) {
                cd.raw_items.push_back(idx);
                raws.insert(raws.end(), rr.p, rr.p + pn);
                raw_off.push_back(raws.size());
            } else {
                cd.enc_items.push_back(idx);
                enc.insert(enc.end(), rr.p, rr.p + pn);
                enc_off.push_back(enc.size());
            }
            rr.p += pn;
            sl.items.push_back(idx);
            placed.push_back(pl);
        }
    }
    sl.begin.push_back((uint32_t)sl.items.size());
    if (!rr.ok || rr.p != rr.e) {
        err("malformed commit record");
        return false;
    }
    st.segments += placed.size();
    // the newly minted references go to this rank's HBM (from the device copy of the record when there is one)
    if (!reg_gid.empty()) {
        if (d_rec) {
            if (!hip_ok(DEVT(agc_hip_ref_register_batch_dev(hip, (uint32_t)reg_gid.size(), reg_gid.data(), d_rec, reg_off.data(), reg_len.data(), nullptr, mml)),
                        "ref_register_batch"))
                return false;
        } else
            for (size_t i = 0; i < reg_gid.size(); ++i)
                if (!hip_ok(DEVT(agc_hip_ref_register(hip, reg_gid[i], rec + reg_off[i], reg_len[i], mml)), "ref_register"))
                    return false;
    }
    if (!writer)
        return true;
    // fetched = new references, then raw items (the layout book_and_store indexes)
    bytes_t fetched;
    fetched.reserve(refs_raw.size() + raws.size());
    fetched.insert(fetched.end(), refs_raw.begin(), refs_raw.end());
    fetched.insert(fetched.end(), raws.begin(), raws.end());
    cd.fetched_off = ref_off;
    for (size_t i = 1; i < raw_off.size(); ++i)
        cd.fetched_off.push_back(refs_raw.size() + raw_off[i]);
    cd.ctgs = &ctgs;
    cd.placed = &placed;
    cd.fetched = &fetched;
    for (size_t i = 0; i + 1 < enc_off.size(); ++i) {
        cd.enc_ptr.push_back(enc.data() + enc_off[i]);
        cd.enc_len.push_back((uint32_t)(enc_off[i + 1] - enc_off[i]));
    }
    return book_and_store(cd);
}

// CSegment::finish for every group (agc_compressor.cpp:880-904, segment.cpp:125-133)
void CAGCCompressor::Impl::finish_groups()
{
    std::vector<ZJob> jobs;
    for (uint32_t gid = 0; gid < groups.size(); ++gid) {
        Group &g = groups[gid];
        if (!g.lzp_off.empty())
            make_pack_job(jobs, g, g.lzp_data, g.lzp_off);
        if (!g.raw_off.empty())
            make_pack_job(jobs, g, g.raw_data, g.raw_off);
        if (g.packed && g.pk_delta) { // store_compressed_delta_in_archive, segment.h:283-292
            if (g.stream_delta < 0)
                g.stream_delta = ar.register_stream(ss_delta_name(gid));
            ar.add_part_buffered(g.stream_delta, bytes_t(g.pk_delta, g.pk_delta + g.pk_delta_size), g.pk_delta_meta);
        }
    }
    run_jobs(jobs);
}

// ---------------------------------------------------------------------------
bool CAGCCompressor::AddSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                                     const uint64_t *ctg_off)
{
    return PrepareSampleDevice(sample_name, contig_names, d_codes, ctg_off) && CommitPrepared();
}

// scan + classification + speculative encode of a sample, against the state this process has NOW; nothing is registered
// yet.  d_codes must stay untouched until CommitPrepared.  In the multi-GPU mode a rank calls this for its next sample
// while earlier samples are still being committed elsewhere (agc_amd/dist.py).
bool CAGCCompressor::PrepareSampleDevice(const std::string &sample_name, const std::vector<std::string> &contig_names, const uint8_t *d_codes,
                                         const uint64_t *ctg_off)
{
    Impl &I = *p;
    if (!I.created || I.concatenated || I.prepared)
        return false;
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
    I.prepared.reset(new Impl::BatchState());
    if (!I.batch_prepare(*I.prepared, I.prepared_ctgs, d_codes, nullptr, I.dist_world > 1 && !I.adaptive && !I.appending)) {
        I.prepared.reset();
        return false;
    }
    return true;
}

// the order-dependent half: collection registration, revalidation against what changed since PrepareSampleDevice, commit
bool CAGCCompressor::CommitPrepared()
{
    Impl &I = *p;
    if (!I.prepared)
        return false;
    std::unique_ptr<Impl::BatchState> b = std::move(I.prepared); // (note_new_group stops logging)
    I.coll.reset_prev_sample_name();
    for (auto &ct : I.prepared_ctgs)
        if (!I.coll.register_sample_contig(ct.sample, ct.name)) {
            I.err("Error: Pair sample_name:contig_name " + ct.sample + ":" + ct.name + " is already in the archive!");
            return false; // (AddSampleFiles skips such contigs; a device-resident sample is all or nothing)
        }
    if (I.prepared_ctgs.empty())
        return true;
    if (!I.changed_log.empty()) {
        b->changed.swap(I.changed_log);
        I.changed_log.clear();
        b->s_from = 0;
        if (!I.revalidate(*b))
            return false;
    }
    uint32_t n_done = 0;
    return I.batch_commit(*b, n_done);
}

bool CAGCCompressor::AddSampleFiles(const std::vector<std::pair<std::string, std::string>> &files, uint32_t no_threads)
{
    Impl &I = *p;
    (void)no_threads;
    if (!I.created)
        return false;
    if (files.empty())
        return true;
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
        uint64_t bytes = 0;
    };
    std::deque<Pending> pending;
    uint64_t pending_bytes = 0;
    const uint64_t WINDOW_BYTES = 64ull << 20;
    const uint32_t WINDOW_MAX = I.adaptive ? 1u : 256u; // new splitters change later scans: no speculation in -a mode
    uint32_t window = 1; // grows while whole windows commit, shrinks to what did commit otherwise

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
        if (!I.hip_ok(DEVTI(agc_hip_sample_buffer(I.hip, tot, &d_base)), "sample_buffer"))
            return false;
        uint64_t o = 0;
        for (uint32_t b = 0; b < nb; ++b)
            for (size_t c = 0; c < pending[b].ctgs.size(); ++c) {
                Contig ct = pending[b].ctgs[c];
                ct.sample_idx = b;
                ct.off = o;
                ct.len = pending[b].data[c].size();
                if (!I.hip_ok(DEVTI(agc_hip_copy_to_device(I.hip, d_base + o, pending[b].data[c].data(), ct.len)), "copy_to_device"))
                    return false;
                o += ct.len;
                batch.push_back(ct);
                if (I.adaptive)
                    batch_data.push_back(pending[b].data[c]); // new splitters are mined from the host copy
            }
        I.st.t_io += now() - t0;
        uint32_t n_done = 0;
        if (!I.process_batch(batch, d_base, I.adaptive ? &batch_data : nullptr, n_done))
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

    Pending cur;
    for (auto &sf : files) {
        I.coll.reset_prev_sample_name();
        FastaReader fr;
        if (!fr.open(sf.second)) {
            I.err("Cannot open file: " + sf.second);
            continue;
        }
        std::string id;
        bytes_t contig;
        bool any_read = false, any_added = false;
        double t0 = now();
        while (fr.read_contig_raw(id, contig)) {
            const std::string sname = I.concatenated ? std::string() : sf.first;
            if (!I.coll.register_sample_contig(sname, id))
                I.err("Error: Pair sample_name:contig_name " + (I.concatenated ? id : sf.first) + ":" + id + " is already in the archive!");
            else {
                preprocess_raw_contig(contig);
                Contig ct;
                ct.sample = sname;
                ct.name = id;
                cur.ctgs.push_back(ct);
                cur.bytes += contig.size();
                cur.data.emplace_back(std::move(contig));
                contig.clear();
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
    if (I.processed_samples % I.pack_cardinality != 0) {
        I.coll.store_contig_batch((I.processed_samples / I.pack_cardinality) * I.pack_cardinality, I.processed_samples);
        I.stored_samples = I.processed_samples;
    }
    I.ar.flush_out_buffers();
    return true;
}

// close_compression (agc_compressor.cpp:2094-2115), store_metadata (:175-284), store_file_type_info (:287-300)
bool CAGCCompressor::Close(uint32_t no_threads)
{
    Impl &I = *p;
    (void)no_threads;
    if (!I.created)
        return false;
    // samples that came through AddSampleDevice / ApplyRecord: the open collection batch is stored here, where
    // AddSampleFiles does it at its end (agc_compressor.cpp:2254-2255)
    if (I.stored_samples < I.processed_samples && I.processed_samples % I.pack_cardinality != 0) {
        I.coll.store_contig_batch((I.processed_samples / I.pack_cardinality) * I.pack_cardinality, I.processed_samples);
        I.stored_samples = I.processed_samples;
        I.ar.flush_out_buffers();
    }
    I.finish_groups();
    I.ar.flush_out_buffers();

    auto app32 = [](bytes_t &d, uint32_t x) {
        for (int i = 0; i < 4; ++i, x >>= 8)
            d.push_back((uint8_t)(x & 0xff));
    };
    auto app64 = [](bytes_t &d, uint64_t x) {
        for (int i = 0; i < 8; ++i, x >>= 8)
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

    v.clear();
    for (uint64_t x : I.splitters) // sorted
        app64(v, x);
    I.ar.add_part(I.ar.register_stream("splitters"), v, I.splitters.size());

    std::vector<std::pair<pk_t, int32_t>> ms;
    ms.reserve(I.map_segments.size());
    I.map_segments.for_each([&](const pk_t &k, int32_t v) { ms.emplace_back(k, v); });
    std::sort(ms.begin(), ms.end());
    v.clear();
    for (auto &x : ms) {
        app64(v, x.first.first);
        app64(v, x.first.second);
        app32(v, (uint32_t)x.second);
    }
    I.ar.add_part(I.ar.register_stream("segment-splitters"), v, ms.size());

    I.coll.complete_serialization();

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
    I.st.archive_bytes = I.ar.bytes_written();
    I.created = false;
    return true;
}

} // namespace agc
