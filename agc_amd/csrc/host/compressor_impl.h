// compressor_impl.h -- internals of agc::CAGCCompressor shared by compressor.cpp (API, create / append / close),
// compressor_batch.cpp (the per-window pipeline: scan -> classification -> placement -> commit runs) and
// compressor_dist.cpp (commit records of the multi-GPU mode).  Citations: file:line under the reference tree.
#pragma once
#include "compressor.h"
#include "host_support.h"
#include <list>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "archive_read.h"
#include "reader.h"
#include "../../../include/agc_hip.h"

#include <chrono>
#include <deque>
#include <filesystem>
#include <future>
#include <cmath>
#include <iostream>
#include <numeric>
#include <set>
#include <zlib.h>

namespace agc {

namespace detail {

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

// (kmer1, kmer2) -> group id: flat open-addressing table (one cache line per lookup instead of a node chase).  The same array,
// slot for slot, is what the device's look-up kernel probes (include/agc_hip.h: agc_hip_group_slot, agc_hip_group_hash): the
// map remembers which slots changed since the mirror in HBM was last brought up to date (sync_device).
class PkMap {
    struct Slot {
        uint64_t a, b;
        int32_t v;
        uint32_t used;
    };
    static_assert(sizeof(Slot) == sizeof(agc_hip_group_slot), "the device mirrors this array");
    std::vector<Slot> t;
    size_t n = 0, mask = 0;
    std::vector<uint64_t> dirty; // slots written since the last sync_device
    bool rebuilt = true;         // the whole array was laid out anew
    // == agc_hip_group_hash (inlined: the look-ups of a sample run in the pool's inner loops; hash_agrees() checks it once)
    static size_t home(const pk_t &k) { return PairHash()(k); }

public:
    static bool hash_agrees()
    {
        const pk_t probes[3] = {{1, 2}, {0x0123456789ABCDEFULL, ~0ULL}, {~0ULL, 0x9E3779B97F4A7C15ULL}};
        for (const pk_t &q : probes)
            if (home(q) != (size_t)agc_hip_group_hash(q.first, q.second))
                return false;
        return true;
    }

private:
    void grow()
    {
        std::vector<Slot> old;
        old.swap(t);
        t.assign(old.empty() ? 1024 : old.size() * 2, Slot{0, 0, 0, 0});
        mask = t.size() - 1;
        rebuilt = true;
        dirty.clear();
        for (const Slot &s : old)
            if (s.used) {
                size_t i = home(pk_t{s.a, s.b}) & mask;
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
        rebuilt = true;
        dirty.clear();
    }
    // the slot of p (a pointer find / operator[] returned) was written
    void touch(const int32_t *p)
    {
        if (!rebuilt)
            dirty.push_back((uint64_t)(((const uint8_t *)p - (const uint8_t *)t.data()) / sizeof(Slot)));
    }
    // brings the device's copy up to date: the whole array after a re-layout, the changed slots otherwise
    int sync_device(agc_hip_ctx *ctx)
    {
        int rc = AGC_HIP_OK;
        if (rebuilt)
            rc = agc_hip_group_map_set(ctx, (const agc_hip_group_slot *)t.data(), t.size());
        else if (!dirty.empty()) {
            std::sort(dirty.begin(), dirty.end());
            dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
            std::vector<agc_hip_group_slot> sl(dirty.size());
            for (size_t i = 0; i < dirty.size(); ++i)
                memcpy(&sl[i], &t[dirty[i]], sizeof(Slot));
            rc = agc_hip_group_map_update(ctx, (uint32_t)dirty.size(), dirty.data(), sl.data());
        }
        if (rc == AGC_HIP_OK) {
            rebuilt = false;
            dirty.clear();
        }
        return rc;
    }
    int32_t *find(const pk_t &k)
    {
        if (t.empty())
            return nullptr;
        for (size_t i = home(k) & mask;; i = (i + 1) & mask) {
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
        size_t i = home(k) & mask;
        while (t[i].used)
            i = (i + 1) & mask;
        t[i] = Slot{k.first, k.second, 0, 1};
        ++n;
        touch(&t[i].v); // (the caller assigns the value right away: the slot is read when the mirror is synchronised)
        return t[i].v;
    }
    template <typename F> void for_each(F f) const
    {
        for (const Slot &s : t)
            if (s.used)
                f(pk_t{s.a, s.b}, s.v);
    }
};

inline double now()
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
    int32_t map_gid = -1;   // both splitters: what map_segments holds for pk (looked up once, by the pool; -1: not there)
    int32_t dev_gid = -2;   // ... as the device's look-up delivered it with the segment (agc_hip_segments_packed); -2: not looked up there
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

    // room_for != 0: the first sequence of a pack reserves room for that many of its size (256 KiB at most).  Every group of a
    // collection of related genomes gets one delta per sample, so 50 k vectors that double outgrew their capacity in the SAME
    // sample (the 3rd, 5th, 9th, 17th ...): 50 k reallocations + copies + the page faults of half a GB of new blocks in one
    // bookkeeping task -- 30 ms instead of 4, which the next step then waited for.  Reserved once, the pages are touched as they fill.
    static void push(bytes_t &data, std::vector<uint32_t> &off, const uint8_t *b, size_t n, size_t room_for = 0)
    {
        if (room_for && data.capacity() < data.size() + n + 1) {
            const size_t per = std::max(n + 1 + (n >> 3), data.empty() ? (size_t)0 : data.size() / std::max<size_t>(1, off.size()));
            const size_t left = room_for > off.size() ? room_for - off.size() : 1;
            data.reserve(std::max(data.size() + std::max(n + 1, std::min<size_t>(per * left, (size_t)256 << 10)), 2 * data.capacity()));
            off.reserve(std::max(room_for, off.size() + 1));
        }
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
// Grow-only array whose elements never move: chunks of 4096 behind a fixed directory.  One thread appends (the one that mints
// groups); another may use elements that existed when it was handed its work (the bookkeeping thread) -- no reference it holds
// is ever invalidated, and reading element i never touches memory an append writes.
template <class T> class StableVec {
    static constexpr size_t CH_BITS = 12, CH = (size_t)1 << CH_BITS, N_DIR = (size_t)1 << 15; // 134 M elements
    std::unique_ptr<std::unique_ptr<T[]>[]> dir{new std::unique_ptr<T[]>[N_DIR]};
    size_t n = 0;

public:
    size_t size() const { return n; }
    T &operator[](size_t i) { return dir[i >> CH_BITS][i & (CH - 1)]; }
    const T &operator[](size_t i) const { return dir[i >> CH_BITS][i & (CH - 1)]; }
    T &back() { return (*this)[n - 1]; }
    T &emplace_back()
    {
        if ((n & (CH - 1)) == 0 && !dir[n >> CH_BITS])
            dir[n >> CH_BITS].reset(new T[CH]);
        (*this)[n] = T();
        return (*this)[n++];
    }
    void resize(size_t m) // (grows only)
    {
        while (n < m)
            emplace_back();
    }
    void clear()
    {
        for (size_t c = 0; c < N_DIR && dir[c]; ++c)
            dir[c].reset();
        n = 0;
    }
};

struct PinnedBytes;
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
    // ref_slot >= 0: lag counters and symbols of the new references are still on their way (agc_hip_ref_store_begin_packed into
    // *ref_pin: counters of ref_nr references, then the symbols); whoever does the books waits for the slot first (finish_ref_store)
    int ref_slot = -1;
    size_t ref_nr = 0;
    const PinnedBytes *ref_pin = nullptr;
};

// result buffer in pinned host memory (agc_hip_host_alloc; plain malloc when that fails): grows, keeps its content, never shrinks
struct PinnedBytes {
    agc_hip_ctx *ctx = nullptr;
    uint8_t *p = nullptr;
    size_t n = 0;
    bool pinned = false;
    size_t size() const { return n; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    bool resize(size_t m, bool keep = true)
    {
        if (m <= n)
            return true;
        // (a buffer that has to grow grows by a quarter at least: the deltas of two human samples differ by a fraction of a percent,
        // and every new maximum used to cost a hipHostFree + hipHostMalloc of tens of MB -- 6 + 6 ms -- on the thread that holds a lane)
        if (n)
            m = std::max(m, n + n / 4);
        if (!keep)
            release();
        void *q = nullptr;
        static const bool laps = getenv("AGC_AMD_LAPS") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t had = n;
        const bool pin = ctx && agc_hip_host_alloc(ctx, m, &q) == AGC_HIP_OK && q;
        if (laps && m >= (4u << 20))
            fprintf(stderr, "    pinned buffer: %.1f -> %.1f MB in %.3f ms\n", had / 1e6, m / 1e6,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (!pin)
            q = malloc(m);
        if (!q)
            return false;
        if (p) {
            memcpy(q, p, n);
            release();
        }
        p = (uint8_t *)q;
        n = m;
        pinned = pin;
        return true;
    }
    void release()
    {
        if (p) {
            if (pinned)
                agc_hip_host_free(ctx, p);
            else
                free(p);
        }
        p = nullptr;
        n = 0;
    }
    void swap(PinnedBytes &o)
    {
        std::swap(ctx, o.ctx);
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(pinned, o.pinned);
    }
    PinnedBytes() = default;
    PinnedBytes(const PinnedBytes &) = delete;
    PinnedBytes &operator=(const PinnedBytes &) = delete;
    ~PinnedBytes() { release(); }
};

struct ZJob { // one archive part to produce
    int stream_id;
    uint32_t gid = 0;  // pack jobs: the group (for a deferred stream registration)
    int kind;          // 0 = reference (tuples/zstd13 or zstd19), 1 = pack (zstd17)
    bytes_t data;      // raw bytes (reference symbols or concatenated pack)
    bool repetitive = false;
    // a reference the device entropy stage takes: what it compresses (tuples of `data`; empty = `data` itself), at which level
    bytes_t staged;
    uint8_t level = 17, marker = 0;
    bytes_t out;
    uint64_t meta = 0;
    std::shared_ptr<PartSlot> slot; // asynchronous entropy stage: where the finished part goes (already queued in the archive)
};

// bytes2tuples, src/common/segment.h:73-138
inline void bytes2tuples(const bytes_t &v, bytes_t &out)
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
inline const CnvTable g_cnv;

inline void preprocess_raw_contig(bytes_t &ctg)
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
    gzFile f = nullptr;   // gzip input (zlib inflates; also its transparent mode for anything that is not a regular plain file)
    int fd = -1;          // plain regular file: read() straight into the block buffer, no zlib in between
    uint64_t remaining = 0; // plain file: bytes not yet read (upper bound for the contig being read: the reserve hint)
    std::vector<uint8_t> buf;
    size_t pos = 0, filled = 0;
    bool fill()
    {
        pos = 0;
        if (fd >= 0) {
            ssize_t r;
            do
                r = ::read(fd, buf.data(), buf.size());
            while (r < 0 && errno == EINTR);
            filled = r > 0 ? (size_t)r : 0;
            remaining -= std::min<uint64_t>(remaining, filled);
        } else {
            int r = gzread(f, buf.data(), (unsigned)buf.size());
            filled = r > 0 ? (size_t)r : 0;
        }
        return filled != 0;
    }

public:
    bool open(const std::string &fn)
    {
        close();
        // a regular file that does not start with the gzip magic is read directly
        struct stat st;
        int h = ::open(fn.c_str(), O_RDONLY);
        uint64_t gz_size = ~0ull; // (the size of a regular file that turns out to be gzip)
        if (h >= 0 && fstat(h, &st) == 0 && S_ISREG(st.st_mode)) {
            gz_size = (uint64_t)st.st_size;
            uint8_t magic[2] = {0, 0};
            const ssize_t r = ::pread(h, magic, 2, 0);
            if (!(r == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
                fd = h;
                remaining = (uint64_t)st.st_size;
                // (the block buffer is zero-filled memory: 16 MiB of it for a 30 kb genome was most of the time to read one)
                buf.resize((size_t)std::min<uint64_t>(16u << 20, std::max<uint64_t>(remaining + 1, 4096)));
                pos = filled = 0;
                return true;
            }
        }
        if (h >= 0)
            ::close(h);
        f = gzopen(fn.c_str(), "rb");
        if (!f)
            return false;
        gzbuffer(f, (unsigned)std::min<uint64_t>(1u << 20, std::max<uint64_t>(gz_size, 8192)));
        buf.resize((size_t)std::min<uint64_t>(16u << 20, std::max<uint64_t>(gz_size < (1u << 20) ? gz_size * 8 : ~0ull, 65536)));
        pos = filled = 0;
        return true;
    }
    void close()
    {
        if (f)
            gzclose(f);
        f = nullptr;
        if (fd >= 0)
            ::close(fd);
        fd = -1;
    }
    ~FastaReader() { close(); }
    // A big plain file at once: the file is mapped, its '>' positions are found and the contig bodies copied out by a few
    // threads (one 3 Gbp assembly is one file: read as a stream it is one thread's work).  Same records as the loop over
    // read_contig_raw below -- a record starts at the beginning of the file and at every '>' that is not inside a header line,
    // its header ends at the first '\n' or '\r', reading stops at the first record without name or without body.
    // false: not a plain regular file of at least min_size bytes (the caller reads it as a stream).
    static bool read_all_mapped(const std::string &fn, std::vector<std::string> &ids, std::vector<bytes_t> &ctgs, unsigned n_threads,
                                uint64_t min_size)
    {
        const int h = ::open(fn.c_str(), O_RDONLY);
        if (h < 0)
            return false;
        struct stat st;
        if (fstat(h, &st) != 0 || !S_ISREG(st.st_mode) || (uint64_t)st.st_size < std::max<uint64_t>(min_size, 2)) {
            ::close(h);
            return false;
        }
        const size_t n = (size_t)st.st_size;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, h, 0);
        ::close(h);
        if (m == MAP_FAILED)
            return false;
        const uint8_t *d = (const uint8_t *)m;
        if (d[0] == 0x1f && d[1] == 0x8b) { // gzip
            munmap(m, n);
            return false;
        }
        (void)madvise(m, n, MADV_SEQUENTIAL);
        n_threads = std::max(1u, std::min(n_threads, 16u));
        // 1. the '>' positions, range by range
        std::vector<std::vector<size_t>> gt_part(n_threads);
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_threads; ++t)
                th.emplace_back([&, t] {
                    const size_t b = n * t / n_threads, e = n * (t + 1) / n_threads;
                    for (size_t p = b; p < e;) {
                        const uint8_t *q = (const uint8_t *)memchr(d + p, '>', e - p);
                        if (!q)
                            break;
                        gt_part[t].push_back((size_t)(q - d));
                        p = (size_t)(q - d) + 1;
                    }
                });
            for (auto &x : th)
                x.join();
        }
        std::vector<size_t> gt;
        for (auto &v : gt_part)
            gt.insert(gt.end(), v.begin(), v.end());
        // 2. the records
        struct Rec {
            size_t id_b, id_e, body_b, body_e;
        };
        std::vector<Rec> recs;
        size_t gi = 0;
        for (size_t s = 0; s < n;) {
            size_t e = s;
            while (e < n && d[e] != '\n' && d[e] != '\r')
                ++e;
            if (e >= n)
                break; // (the header runs into the end of the file)
            const size_t body_b = e + 1;
            while (gi < gt.size() && gt[gi] < body_b)
                ++gi;
            const size_t next = gi < gt.size() ? gt[gi] : n;
            if (e <= s + 1 || next <= body_b)
                break; // no name or no body: the stream reader stops here too
            recs.push_back({s + 1, e, body_b, next});
            s = next;
        }
        // 3. names and bodies
        ids.resize(recs.size());
        ctgs.resize(recs.size());
        {
            std::atomic<size_t> next_rec{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < std::min<size_t>(n_threads, std::max<size_t>(recs.size(), 1)); ++t)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t i = next_rec.fetch_add(1);
                        if (i >= recs.size())
                            break;
                        ids[i].assign((const char *)d + recs[i].id_b, recs[i].id_e - recs[i].id_b);
                        ctgs[i].assign(d + recs[i].body_b, d + recs[i].body_e);
                    }
                });
            for (auto &x : th)
                x.join();
        }
        munmap(m, n);
        return true;
    }
    bool read_contig_raw(std::string &id, bytes_t &ctg)
    {
        id.clear();
        ctg.clear();
        if (!f && fd < 0)
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
        // plain file: the contig cannot be longer than what is left of the file -- one allocation instead of a doubling series
        // of ever larger copies (address space only: pages are touched as they are filled)
        if (fd >= 0)
            ctg.reserve((size_t)std::min<uint64_t>(remaining + (filled - pos), (uint64_t)512 << 20));
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
        if (ctg.capacity() > 2 * ctg.size() + (1u << 20))
            ctg.shrink_to_fit(); // (a file of many contigs: the hint was the rest of the file for each of them)
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


// ---------------------------------------------------------------------------
// Host pieces of the adaptive mode's find_new_splitters (agc_compressor.cpp:2054-2081, 630-704,
// 762-825); the reference genome itself is preprocessed on the GPU (agc_hip_determine_splitters_dev).
// ---------------------------------------------------------------------------
// splitters of one contig given the sorted candidate k-mers (find_splitters_in_contig, :762-825)
inline void find_splitters_in_contig(const bytes_t &c, uint32_t k, uint32_t segment_size, const std::vector<uint64_t> &cand,
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

inline void enumerate_kmers(const bytes_t &c, uint32_t k, std::vector<uint64_t> &km)
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
inline void split_singletons(std::vector<uint64_t> &km, std::vector<uint64_t> *dup)
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
} // namespace detail

using namespace detail;

// ===========================================================================
struct CAGCCompressor::Impl {
    int device = 0;
    agc_hip_ctx *hip = nullptr;
    ZstdApi zstd;
    std::unique_ptr<ThreadPool> pool;   // the stages of a step
    std::unique_ptr<ThreadPool> zpool;  // the entropy stage (runs beside the steps)
    std::vector<std::unique_ptr<ZstdCtx>> zctx; // one per zpool thread

    // Asynchronous entropy stage.  The reference's workers compress a pack the moment it fills (segment.h:172-215) while the
    // other workers go on; here the zstd jobs of a registration are handed to one background thread that drives the device
    // (agc_hip_zstd17_batch, own HIP stream) and zpool; their parts take their place in the archive at once (PartSlot) and are
    // written, in order, when the payload is there.  Close() only drains.  AGC_AMD_SYNC_ENTROPY=1: every registration waits.
    std::thread z_thread;
    std::mutex z_mtx;
    std::condition_variable z_cv, z_idle_cv;
    std::deque<std::vector<ZJob>> z_queue;
    bool z_busy = false, z_stop = false;
    std::atomic<bool> z_caller_waits{false}; // the caller stands still for the stage (Close, Drain): its launches may take the LDS
    bool sync_entropy = false;
    std::atomic<bool> heavy_steps{false};    // the windows being driven are human-size: the entropy thread takes a quarter of the pool for small batches
    size_t par_min = 4096; // lists shorter than this are walked by the calling thread (AGC_AMD_PAR_MIN: tests force the pool paths)
    void z_submit(std::vector<ZJob> &&jobs);
    void z_wait_all();
    void z_main();
    void z_shutdown();
    ~Impl()
    {
        book_shutdown();
        z_shutdown();
    }

    // Asynchronous bookkeeping stage.  The second half of store_segments (book_and_store: packs, in-group ids, collection
    // records, the parts' places in the archive) reads nothing the classification of the next sample needs and writes nothing
    // it reads: it runs on one background thread, in registration order, beside the next sample's scan and classification (and,
    // on the writer rank of the multi-GPU mode, beside the next owner's commit).  What orders the two threads:
    //   * a registration's buffers (placed items, fetched symbols, the pinned delta buffers) are handed over by SWAPPING them with
    //     a second set (*_alt) that only the queued task touches; before the next hand-over swaps again the previous own task
    //     must be done (book_wait_seq(last_own_seq): it has been for a long time).  book_wait() before anything reads what the
    //     stage produces (Close, the sync path); `groups` may grow meanwhile (StableVec: its elements never move);
    //   * the LZ encode of the registration may still be in flight on the device's second lane when the call returns: the task
    //     collects it.  Only for samples in a staging buffer the device context owns (whatever overwrites that buffer waits for
    //     the encode on its stream, api.hip Lane2::done); AGC_AMD_ASYNC_ENCODE=0 turns it off;
    //   * coll_mtx around every access to the collection's sample table;
    //   * Group::exists / ref_size belong to the main thread (set when the group is minted), the packs to the book thread.
    // Not used in append / concatenated mode, for windows of several registrations, or with AGC_AMD_SYNC_ENTROPY;
    // AGC_AMD_ASYNC_BOOK=0 turns it off.
    struct BookTask {
        CommitData cd;
        std::vector<Contig> ctgs;   // own copy (the caller's vector does not outlive its call)
        std::vector<Placed> placed; // apply_record: everything is the task's own
        bytes_t fetched, enc;
        std::unique_ptr<PinnedBytes> enc_recv; // (the received body itself, when it came in through RecordBodyBuffer)
        // an LZ encode still in flight on the device's second lane (agc_hip_lz_encode_begin_dev): the task collects it
        // (agc_hip_lz_encode_end) into *enc_dst before it does the books; enc_todo = positions in cd.enc_items it covers
        bool enc_pending = false;
        std::vector<uint32_t> enc_todo;
        uint64_t enc_text = 0;
        PinnedBytes *enc_dst = nullptr;
        // ... and a second one on lane 1 (agc_hip_lz_encode_begin_packed_on): the segments the sample's own new groups took,
        // launched at commit time behind the whole-sample encode of lane 0
        // early_only: nothing but the raw collection of lane 0 into *enc_dst (enc_n deltas, offsets -> Impl::early_enc): queued right
        // behind the launch of the whole-sample encode, so that the lane is free and the deltas are on the host long before the
        // registration's own task comes.  enc_collected: this (full) task finds lane 0's deltas in *enc_dst already, at enc_eoff
        bool early_only = false;
        uint32_t enc_n = 0;
        bool enc_collected = false;
        std::vector<uint64_t> enc_eoff;
        bool enc2_pending = false;
        std::vector<uint32_t> enc2_todo;
        uint64_t enc2_text = 0;
        PinnedBytes *enc2_dst = nullptr;
        Impl *owner = nullptr;
        ~BookTask()
        {
            if (enc_recv && owner) {
                std::lock_guard<std::mutex> lk(owner->body_pool_mtx);
                owner->body_pool.emplace_back(std::move(enc_recv));
            }
        }
    };
    std::thread book_thread;
    std::mutex book_mtx;
    std::condition_variable book_cv, book_idle_cv;
    std::deque<std::unique_ptr<BookTask>> book_queue;
    bool book_busy = false, book_stop = false, book_failed = false;
    bool async_book = true, async_encode = true;
    bool early_collect = true; // the whole-sample encode is collected by an early task of the book thread (AGC_AMD_EARLY_COLLECT=0: by the registration's own)
    uint64_t book_seq_submitted = 0, book_seq_done = 0; // tasks are numbered; they complete in order
    uint64_t last_own_seq = 0;          // the last task that points into this object's buffers (the *_alt set below)
    PinnedBytes ref_pin[2];             // staging of agc_hip_ref_store_begin_packed, alternating (the previous one is the queued task's)
    uint32_t ref_pin_next = 0;
    bool ref_store_async = true;        // AGC_AMD_REF_STORE_ASYNC=0: lag counters and symbols are waited for where they are asked for
    bool finish_ref_store(CommitData &cd, bytes_t &fetched_dst); // waits for the slot, makes cd.repetitive and cd.fetched
    struct EarlyEnc {                   // result of an early_only task (written by the book thread, read after book_wait_seq)
        bool ok = false;
        std::vector<uint64_t> eoff;
    } early_enc;
    double book_seconds = 0;            // the book thread's own time (added to st.t_store / h_store by book_wait)
    uint64_t book_delta_bytes = 0;      // deltas collected by the book thread (added to st.delta_bytes by book_wait)
    std::unique_ptr<ThreadPool> bpool;  // the stage's own workers (`pool` belongs to the thread that drives the steps)
    std::mutex coll_mtx;
    bool book_can_async(uint32_t n_samples) const { return async_book && !appending && !concatenated && !sync_entropy && n_samples == 1; }
    uint64_t book_submit(std::unique_ptr<BookTask> &&t);
    void book_main();
    bool book_wait();
    bool book_wait_seq(uint64_t seq); // the tasks up to `seq` are done
    void book_shutdown();
    bool book_on_thread = false; // (book thread only) book_and_store runs as a queued task

    bool created = false;
    uint32_t pack_cardinality = 50, k = 31, segment_size = 60000, mml = 20, verbosity = 0;
    bool concatenated = false, adaptive = false;

    ArchiveWriter ar;
    CollectionV3 coll;
    std::vector<uint64_t> splitters;

    PkMap map_segments;                                                       // agc_compressor.h:628
    std::unordered_map<uint64_t, std::vector<uint64_t>> terminators;          // agc_compressor.h:629
    StableVec<Group> groups;                                                  // v_segments (elements never move: see StableVec)
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

    // (also called from the entropy thread: one message at a time)
    void err(const std::string &m)
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> g(mu);
        std::cerr << m << std::endl;
    }
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
        const uint8_t *d_base = nullptr;           // one byte per symbol (nullptr when the sample came packed and nothing needed bytes)
        agc_hip_packed pk{};                       // the sample in the 2-bit layout: what every LZ entry point reads (n_symbols != 0)
        const std::vector<bytes_t> *host_data = nullptr;
        uint32_t n_ctg = 0;
        double t0 = 0, dev0 = 0, lap_t = 0;
        std::vector<uint64_t> new_splitters_added; // adaptive mode
        std::vector<uint32_t> subset;              // segments stage_classify works on
        uint32_t n_samples = 1, s_from = 0;        // registrations of the window; first one not committed yet
        bool base_owned = false;                   // d_base is a staging buffer of the device context (see Impl::next_base_owned)
        // adaptive mode, a sample prepared ahead of its turn (multi-GPU mode): the scan must not extend the splitter set; when it
        // would have to, it only says so (needs_turn) and the sample is prepared again at its turn
        bool no_new_splitters = false, needs_turn = false;
        // adaptive mode, device segments: the contigs without a splitter and what find_new_splitters made of them, handed to the
        // host's scan when the set has to grow (stage_scan_dev -> stage_scan)
        bool mined_valid = false;
        std::vector<uint32_t> mined_need;
        std::vector<std::vector<uint64_t>> mined_found;
        struct Spec {                              // speculative delta of a placed item (by Placed::key)
            uint64_t off = 0, enc_off = 0;
            uint32_t gid = 0, len = 0, enc_len = 0;
            int32_t pending = -1; // >= 0: the delta is result no. `pending` of the encode in flight on the device's second lane
            bool rc = false, valid = false;
        };
        std::vector<Spec> spec;
        uint64_t spec_bytes = 0;                   // bytes of enc_buf the speculative deltas occupy
        // single-registration window: the segments whose group is known from their two splitters (nearly all) are encoded
        // on the device's second stream while the rest of the window is still being classified
        bool overlap_encode = false, enc_in_flight = false;
        // the segments came from the device with their groups (agc_hip_segments_packed): the first classification takes the keys as
        // delivered; dev_enc_n != 0: the encode of the segments whose group was known is in flight on the device's second lane
        bool dev_keys = false;
        uint32_t dev_enc_n = 0;
        bool known_launch_due = false; // the whole-sample encode has its descriptors on the device and is not launched yet (launch_known_encode)
        uint64_t known_text = 0;       // symbols of its texts
        uint64_t early_seq = 0; // != 0: the book thread's early task that collects that encode (Impl::early_enc takes its result)
        std::vector<uint32_t> flight_keys, flight_gid, flight_len;
        std::vector<uint64_t> flight_off;
        std::vector<uint8_t> flight_rc;
        std::vector<uint64_t> changed;             // k-mers whose terminator list changed in the last commit run
        uint32_t commit_upto = 0;                  // registrations of the window that are committed now
        std::vector<uint32_t> order;               // committed items in registration order
        // the placement of the segments that have no split-point job (all but a few per cent), made on a helper thread while the
        // driving thread waits for the split points: stage_place copies these and numbers the parts (place_ahead_valid: once)
        std::future<std::pair<uint64_t, uint64_t>> spec_fill; // the helper that fills `spec` for the whole-sample encode (finish_spec_fill)
        std::vector<Placed> place_ahead;
        std::vector<uint8_t> place_ahead_ok;
        bool place_ahead_valid = false;
        std::vector<SampleLists> per_sample;
        struct Store {                             // stage_store's working set between its two halves
            std::vector<uint32_t> new_ref_items, raw_items, enc_items; // placed indices
            std::vector<uint8_t> repetitive;
            std::vector<uint64_t> fetched_off;
            int ref_slot = -1;                     // (see CommitData)
            size_t ref_nr = 0;
            const PinnedBytes *ref_pin = nullptr;
        } sto;
    };
    bool stage_scan(BatchState &b);
    bool launch_known_encode(BatchState &b, bool launched_already = false);
    void finish_spec_fill(BatchState &b);
    int stage_scan_dev(BatchState &b);
    int spec_fill_ahead = 1;           // the table of speculative deltas of the whole-sample encode filled by a helper thread (AGC_AMD_SPEC_FILL_AHEAD=0: in line; 2: any size)
    int place_ahead = 1;               // placement of the segments without a split-point job beside the wait for the split points (AGC_AMD_PLACE_AHEAD=0: after it; 2: for windows of any size -- tests)
    bool pre_launch_encode = true;     // the whole-sample encode launched inside agc_hip_segments_packed (AGC_AMD_PRE_LAUNCH_ENCODE=0: behind the table)
    uint32_t dev_encode_min = 2048;    // segments a sample needs for the device-launched whole-sample encode (AGC_AMD_DEV_ENCODE_MIN: tests)
    bool use_dev_segments(const BatchState &b) const;
    // adaptive mode with windows of several registrations: only where the device delivers the segments (the cut of a window at
    // the registration that brings new splitters lives there)
    bool adaptive_windows() const { return adaptive && dev_segments && k >= 16 && overlap_mode == 0 && !appending; }
    bool dev_segments = true;          // AGC_AMD_DEV_SEGMENTS=0: scan hits to the host, cut and key look-up there (the round-3 path)
    PinnedBytes dev_seg_buf;           // (pinned: the segment table of a human sample is 3 MB per step)
    // the device's second LZ lane carries one encode at a time: launched by the thread that drives the steps, collected by the
    // bookkeeping thread (or by the driver itself on the synchronous path)
    bool lane_inflight[2] = {false, false}; // (guarded by book_mtx) device lanes 0 / 1 (AGC_HIP_ENCODE_LANES)
    void lane2_acquire(int lane = 0);
    void lane2_release(int lane = 0);
    bool stage_classify(BatchState &b);
    bool stage_place(BatchState &b);
    bool stage_register(BatchState &b);
    bool stage_store(BatchState &b);
    bool stage_store_head(BatchState &b);
    bool stage_store_finish(BatchState &b);
    bool spec_encode(BatchState &b);
    bool overlap_encode_begin(BatchState &b);
    bool overlap_encode_end(BatchState &b);
    // AGC_AMD_ENCODE_OVERLAP: off (default) | early (from the key lookup on) | late (from the placement on).  Measured on the
    // 3 Gbp step: the encode kernel hidden behind estimates / split points costs those kernels what it saves (27.8 ms per step
    // off, 26.4-29.6 ms early), so the default keeps the kernels one after the other and their timings clean
    int overlap_mode = 0;
    bool revalidate(BatchState &b);
    bool batch_prepare(BatchState &b, std::vector<Contig> &ctgs, const uint8_t *d_base, const std::vector<bytes_t> *host_data, bool always_speculate);
    bool batch_commit(BatchState &b, uint32_t &n_committed);
    // a sample classified ahead of its turn (multi-GPU mode, PrepareSampleDevice): its working set, and the k-mers whose
    // terminator lists changed since (through other ranks' records)
    std::unique_ptr<BatchState> prepared;
    std::unique_ptr<BatchState> committing; // between CommitPreparedHead and CommitPreparedFinish
    // multi-GPU + adaptive mode: what PrepareSampleDevice was given, to prepare the sample AGAIN at its turn when the speculative
    // prepare could not stand (the sample needs new splitters, or other samples brought some since: spl_version)
    struct PrepArgs {
        const uint8_t *d_codes = nullptr;
        agc_hip_packed packed{};
        bool base_owned = false;
        bool deferred = false;      // no prepared state: prepare at the turn
        uint64_t spl_version = 0;   // splitter set the prepared state was scanned with
        // the counters a prepare adds to, around the prepare (its share is taken back when the prepared state is dropped).  Only the
        // fields the thread that drives the steps owns: a copy of the whole struct would read what the entropy thread is writing
        bool ahead = false; // prepared before its turn (N-rank mode outside append): the only kind of prepare that can go stale
        static constexpr size_t N_PREP_STATS = 13;
        uint64_t st_before[N_PREP_STATS] = {}, st_after[N_PREP_STATS] = {};
    } prep;
    uint64_t spl_version = 0;       // bumped whenever splitters are added (own samples, applied records)
    std::vector<Contig> prepared_ctgs;
    std::vector<uint64_t> changed_log;
    // the next sample, announced by SetNextSamplePackedDevice (pf_next) / already queued on the device (pf_live: its staging copy)
    struct NextSample {
        agc_hip_packed pk{};
        std::vector<uint64_t> ctg_off;
        bool valid = false;
    } pf_next, pf_live;
    bool scan_from_prefetch = false;   // the sample being prepared is pf_live: its first scan is collected, not launched
    // a FASTA -> 2-bit conversion announced by SetNextFastaDevice: queued by launch_prefetch (or by FinishFastaDevice, whichever comes first)
    struct FastaNext {
        bool valid = false, live = false;
        const uint8_t *d_raw = nullptr;
        uint64_t n_raw = 0, esc_cap = 0;
        std::vector<uint64_t> rb, re;
        uint32_t *d_words = nullptr;
        int32_t *d_idx = nullptr;
        uint8_t *d_esc = nullptr;
    } fasta_next;
    bool launch_fasta();
    void launch_prefetch();
    agc_hip_packed packed_sample{};    // the sample being prepared is resident in the 2-bit layout (n_symbols != 0): scans read it
    bool gpu_zstd = false;             // delta packs are entropy-coded on the GPU (libzstd 1.4.x frames; AGC_AMD_HOST_ZSTD=1 turns it off)
    // share of the pack bytes the device takes when both engines run (AGC_AMD_GPU_ZSTD_SHARE fixes it); the start value is the
    // ratio measured on MI355X + 16 host threads (0.7 GB/s against 0.1 GB/s), every call with real work on both sides updates it
    double gpu_zstd_share = 0.88;
    uint32_t gpu_zstd_min = 64;        // fewer packs than this in one call stay on the host (AGC_AMD_GPU_ZSTD_MIN)
    // references (level 13 on their tuples, level 19 for repetitive ones) on the device too when a call brings at least this many:
    // the reference sample's ~50 k references take 0.41 s instead of 0.95 s on 16 host threads.  On since round 4: configs[2] at
    // full size with 5 samples is byte-identical to the reference CLI's archive with it (profiles/r4/c3_full_size_identity_5_samples_
    // refs_on_device.log: 741 MB of the 3.2 GB of entropy input on the device).  AGC_AMD_GPU_ZSTD_REFS=0 keeps them on the host pool.
    uint32_t gpu_zstd_refs_min = 512;
    PinnedBytes zsrc_buf, zdst_buf;    // staging of the device entropy stage (plain malloc: no zero fill of hundreds of MB)
    bool defer_stream_reg = false;     // parallel bookkeeping: pack jobs leave a missing delta stream to the merging thread
    bool minted_since_prepare = false; // a group of any key (also one-sided) was minted while a sample was prepared
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
    // head of the commit record (every rank; compressor_dist.cpp), in pinned host memory: its next stop is a collective, i.e. a copy
    // engine.  DIST_FRAME bytes in front of it belong to the transport (agc_amd/dist.py writes its fixed-size message header there,
    // so that header and head travel as ONE message)
    static constexpr size_t DIST_FRAME = 64;
    PinnedBytes dist_record_buf;
    size_t dist_record_n = 0;
    uint8_t *dist_record_ptr() { return dist_record_buf.data() ? dist_record_buf.data() + DIST_FRAME : nullptr; }
    PinnedBytes dist_body_buf;  // its delta body (the writer only), dist_body_n bytes
    size_t dist_body_n = 0;
    // writer rank: receive buffers for the bodies of the other ranks' records (pinned; a queued bookkeeping task keeps its
    // buffer until it is done, then the buffer comes back here)
    std::mutex body_pool_mtx;
    std::vector<std::unique_ptr<PinnedBytes>> body_pool;
    std::unique_ptr<PinnedBytes> body_recv; // handed out by RecordBodyBuffer, adopted by the next apply_record
    std::vector<uint32_t> dist_body_items; // the record's delta items (placed indices) in list order: head and body agree on it
    bool make_record_head(BatchState &b);
    bool make_record_body(const CommitData &cd);
    bool make_empty_record();
    bool apply_record(const uint8_t *rec, size_t n, const uint8_t *d_rec, const uint8_t *body, size_t body_n);
    void note_new_group(const pk_t &pk, uint32_t gid);
    void finish_groups();
    void run_jobs(std::vector<ZJob> &jobs, bool add_parts = true);
    void run_jobs_round(std::vector<ZJob> &jobs);
    bool host_only_batch(const std::vector<ZJob> &jobs) const; // run_jobs_round would leave every job of it to the host pool
    void host_compress(ZJob &j, unsigned tid);                 // one job through libzstd (zctx[tid])
    void run_host_stream(std::vector<ZJob> &&first);           // z_main: host-only batches without a barrier between them
    bool entropy_stream = true;                                // (AGC_AMD_ENTROPY_STREAM=0: a parallel_for per batch, as before)
    void build_close_jobs(std::vector<ZJob> &jobs);
    void store_open_batch(bool flush = true);
    // Close in steps (multi-GPU entropy stage, compressor.h: CloseCollectPacks / CloseProvideFrames)
    // single-archive mode, writer rank: delta packs that filled during the run wait here (their parts already hold their place in
    // the archive) so that Close in steps can spread them over every rank's GPU together with the packs still open
    std::atomic<uint64_t> verify_frames{0}, verify_bad{0}; // AGC_AMD_VERIFY_DEV_FRAMES: device frames checked against libzstd / different
    std::vector<ZJob> deferred_packs;
    uint64_t deferred_bytes = 0;       // their raw bytes (a safety net bounds them: AGC_AMD_DEFER_MAX_MB -- compressor_batch.cpp)
    std::mutex deferred_mtx;           // (the bookkeeping thread parks packs, the thread that drives the steps takes them: DealCollectPacks)
    // A DEAL: the parked packs handed out in the middle of a run (DealCollectPacks), their inputs back to back, so that the caller can
    // send every rank its share; the shares' frames come back in any order, any time later (DealProvideFrames), or a share stays with
    // this rank's own entropy stage (DealKeepOwn).  The reference's workers code a pack the moment it is full while the others go on
    // (segment.cpp:34-80, segment.h:258-280): here the "workers" are the ranks' GPUs.
    struct Deal {
        uint32_t id = 0;
        std::vector<ZJob> jobs;
        bytes_t src;
        std::vector<uint64_t> off;
        uint32_t left = 0; // packs whose frame has not come back yet
    };
    std::list<Deal> deals;
    uint32_t next_deal_id = 1;
    void publish_frame(ZJob &j, const uint8_t *frame, size_t n);
    Deal *find_deal(uint32_t id);
    void settle_deals_locally();
    std::vector<ZJob> close_jobs;
    std::vector<uint32_t> close_dev_jobs;      // indices in close_jobs of the packs handed out
    std::vector<uint64_t> close_src_off, close_frames_off;
    bool close_collected = false;
    void choose_entropy_stage();
    void add_job_parts(std::vector<ZJob> &jobs, size_t from, size_t to);
    void make_pack_job(std::vector<ZJob> &jobs, uint32_t gid, bytes_t &data, std::vector<uint32_t> &off);
    PinnedBytes enc_buf, enc_buf2; // grown, never shrunk (enc_buf: the window's speculative deltas, enc_buf2: per commit run); pinned
    bytes_t fetch_buf;
    // the second set: what the queued bookkeeping task of the previous registration reads (swapped at the hand-over)
    PinnedBytes enc_alt, enc_alt2;
    bytes_t fetch_alt;
    std::vector<Placed> placed_alt;
    bool next_base_owned = false; // the sample being prepared sits in a staging buffer of the device context (not the caller's)
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
    bool pack_sample(const uint8_t *d_codes, uint64_t n_symbols); // bytes in HBM -> packed_sample (the context's own packed buffers)
    int scan_batch(const std::vector<uint64_t> &ctg_off, uint32_t n_ctg, const uint8_t *d_base, std::vector<uint32_t> &h_ctg,
                   std::vector<uint64_t> &h_pos, std::vector<uint64_t> &h_dir, std::vector<uint64_t> &h_rc, uint64_t &n_hits);
    void after_registration();
};

} // namespace agc
