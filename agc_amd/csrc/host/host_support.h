// host_support.h -- host-side pieces around the HIP hot path: libzstd binding (S3 seam),
// worker pool, .agc v3 container writer, collection (v3) metadata writer.
// Citations are file:line under the reference tree (refresh-bio/agc v3.2.2).
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <dlfcn.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace agc {

using bytes_t = std::vector<uint8_t>;

// ---------------------------------------------------------------------------
// S3 seam: libzstd's own C ABI (ZSTD_compressCCtx one-shot, no dict; the reference's call
// sites: src/common/segment.h:174-176,199-201, src/common/collection_v3.cpp:138-139).
// The library is dlopen'ed so that the SAME libzstd as the oracle (1.4.x) is used; the
// version string is exposed and checked by the tests (bit-identity depends on it).
// ---------------------------------------------------------------------------
struct ZstdApi {
    void *h = nullptr;
    void *(*createCCtx)() = nullptr;
    size_t (*freeCCtx)(void *) = nullptr;
    size_t (*compressCCtx)(void *, void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char *(*versionString)() = nullptr;
    std::string path;

    bool load(std::string &err)
    {
        if (h)
            return true;
        std::vector<std::string> cands;
        if (const char *e = getenv("AGC_ZSTD_LIB"))
            cands.push_back(e);
        cands.push_back("/opt/conda/lib/libzstd.so.1");
        cands.push_back("libzstd.so.1");
        cands.push_back("libzstd.so");
        for (auto &p : cands) {
            // DEEPBIND: this libzstd's calls to its own (exported) internals must not be interposed by another libzstd
            // already in the process (rocprofv3's tool library brings the system's 1.4.8: mixed versions crashed in free())
            h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (h) {
                path = p;
                break;
            }
        }
        if (!h) {
            err = "cannot dlopen libzstd (set AGC_ZSTD_LIB)";
            return false;
        }
        createCCtx = (void *(*)())dlsym(h, "ZSTD_createCCtx");
        freeCCtx = (size_t(*)(void *))dlsym(h, "ZSTD_freeCCtx");
        compressCCtx = (size_t(*)(void *, void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compressCCtx");
        compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        versionString = (const char *(*)())dlsym(h, "ZSTD_versionString");
        if (!createCCtx || !freeCCtx || !compressCCtx || !compressBound || !isError || !versionString) {
            err = "libzstd lacks required symbols";
            return false;
        }
        return true;
    }
};

// one compression context per worker thread (as the reference: agc_compressor.cpp:1100)
struct ZstdCtx {
    ZstdApi *api;
    void *cctx;
    explicit ZstdCtx(ZstdApi *a) : api(a), cctx(a->createCCtx()) {}
    ~ZstdCtx()
    {
        if (cctx)
            api->freeCCtx(cctx);
    }
    // returns the frame size; dst must hold compressBound(n) + 1 bytes
    size_t compress(uint8_t *dst, size_t cap, const uint8_t *src, size_t n, int level) { return api->compressCCtx(cctx, dst, cap, src, n, level); }
};

// AGC_AMD_START_LAPS=1 (a measuring aid, scripts/start_cost.py): milliseconds since the first call at the named points of a run's start
inline void start_lap(const char *what)
{
    static const bool on = getenv("AGC_AMD_START_LAPS") != nullptr;
    if (!on)
        return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "start lap %-44s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

// ---------------------------------------------------------------------------
// worker pool: parallel_for over independent jobs
// ---------------------------------------------------------------------------
class ThreadPool {
    // A call returns when its JOBS are done, not when every worker has woken up and reported: a worker the scheduler brings in late
    // (or one beyond max_workers) finds nothing left and goes back to sleep on its own.  The job counter carries the call's number
    // (epoch) in its high half, so a worker that was preempted between waking up and taking its first job can never take an index of
    // a later call by mistake: it only ever takes an index by compare-and-swap against a value of ITS epoch.
    std::vector<std::thread> th;
    std::mutex mtx;
    std::condition_variable cv, cv_done;
    std::function<void(size_t, unsigned)> fn; // (valid while a call of parallel_for waits: only jobs of that call read it)
    std::atomic<uint64_t> next{0};            // epoch << 32 | next job index
    std::atomic<uint32_t> done{0};            // jobs of the current call that have run
    uint32_t n_jobs = 0;
    unsigned limit = ~0u;
    uint32_t epoch = 0;
    bool stop = false;

    void run(unsigned tid)
    {
        uint32_t seen = 0;
        for (;;) {
            uint32_t my_n, my_epoch;
            unsigned my_limit;
            {
                std::unique_lock<std::mutex> lk(mtx);
                cv.wait(lk, [&] { return stop || epoch != seen; });
                if (stop)
                    return;
                seen = my_epoch = epoch;
                my_n = n_jobs;
                my_limit = limit;
            }
            if (tid >= my_limit)
                continue;
            for (;;) {
                uint64_t cur = next.load(std::memory_order_acquire);
                if ((uint32_t)(cur >> 32) != my_epoch || (uint32_t)cur >= my_n)
                    break;
                if (!next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel))
                    continue;
                fn((size_t)(uint32_t)cur, tid);
                if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == my_n) {
                    std::lock_guard<std::mutex> lk(mtx);
                    cv_done.notify_all();
                }
            }
        }
    }

public:
    // nice_value > 0: the workers yield to the other threads of the process when the cores run out (the entropy stage's pool
    // beside the thread that drives the steps)
    explicit ThreadPool(unsigned n, int nice_value = 0)
    {
        if (n < 1)
            n = 1;
        for (unsigned i = 0; i < n; ++i)
            th.emplace_back([this, i, nice_value] {
                if (nice_value > 0)
                    (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nice_value); // (Linux: per thread)
                run(i);
            });
    }
    ~ThreadPool()
    {
        {
            std::unique_lock<std::mutex> lk(mtx);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : th)
            t.join();
    }
    unsigned size() const { return (unsigned)th.size(); }
    // fn(job index, worker id); returns when every job has run.  max_workers: only that many workers take jobs (a small
    // batch beside other busy threads of the process).  One call at a time (every pool has one calling thread).
    void parallel_for(size_t n, std::function<void(size_t, unsigned)> f, unsigned max_workers = ~0u)
    {
        if (!n)
            return;
        if (n > 0x7fffffffu) { // (the index travels in 32 bits; never in practice)
            for (size_t from = 0; from < n; from += 0x7fffffffu)
                parallel_for(std::min<size_t>(n - from, 0x7fffffffu), [&f, from](size_t i, unsigned t) { f(from + i, t); }, max_workers);
            return;
        }
        std::unique_lock<std::mutex> lk(mtx);
        fn = std::move(f);
        limit = max_workers ? max_workers : 1;
        n_jobs = (uint32_t)n;
        ++epoch;
        done.store(0, std::memory_order_release);
        next.store((uint64_t)epoch << 32, std::memory_order_release);
        cv.notify_all();
        cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) == n_jobs; });
        fn = nullptr;
    }
};

// ---------------------------------------------------------------------------
// .agc container (write side).  src/common/archive.{h,cpp}:
//   part   = varint(metadata) + payload                      archive.cpp:280-293, archive.h:110-125
//   buffered parts are flushed in ascending stream id, insertion order inside   archive.cpp:332-351
//   footer = #streams, per stream: name\0, #parts, raw_size, (offset,size)*, then 8-byte LE footer size
//                                                             archive.cpp:142-169, io.h:371-380
// ---------------------------------------------------------------------------
// a part whose payload is still being compressed (the asynchronous entropy stage): it takes its place in the archive's
// part order when it is handed over; the bytes are written when they are there
struct PartSlot {
    bytes_t out;
    uint64_t meta = 0;
    std::atomic<bool> ready{false};
};

class ArchiveWriter {
    struct Part {
        uint64_t offset, size;
    };
    struct BufPart {
        bytes_t d;
        uint64_t meta;
        std::shared_ptr<PartSlot> slot; // set: d / meta come from the slot once it is ready
    };
    struct Stream {
        std::string name;
        uint64_t raw_size = 0;
        std::vector<Part> parts;
    };
    std::vector<Stream> streams;
    std::unordered_map<std::string, int> ids;
    typedef std::map<int, std::vector<BufPart>> buffer_t;
    buffer_t buffer;
    std::deque<buffer_t> events; // flushes waiting for the payload of one of their parts (written strictly in this order)
    bool buffer_deferred = false;
    FILE *f = nullptr;
    bool own = false;
    bytes_t wbuf;
    bool io_error = false; // a write or the final close failed
public:
    bool failed() const { return io_error; }
private:
    uint64_t f_offset = 0;
    std::mutex mtx;

    void put(const void *p, size_t n)
    {
        if (!f)
            return; // no output path (bench mode): the parts exist, nothing is assembled
        const uint8_t *b = (const uint8_t *)p;
        wbuf.insert(wbuf.end(), b, b + n);
        if (wbuf.size() >= (32u << 20))
            flush_file();
    }
    void flush_file()
    {
        if (f && !wbuf.empty() && fwrite(wbuf.data(), 1, wbuf.size(), f) != wbuf.size())
            io_error = true; // latched: close() reports it (ENOSPC, I/O error)
        wbuf.clear();
    }
    size_t write_num(uint64_t x)
    {
        int nb = 0;
        for (uint64_t t = x; t; t >>= 8)
            ++nb;
        uint8_t tmp[9];
        tmp[0] = (uint8_t)nb;
        for (int i = nb; i; --i)
            tmp[nb - i + 1] = (uint8_t)((x >> ((i - 1) * 8)) & 0xff);
        put(tmp, (size_t)nb + 1);
        return (size_t)nb + 1;
    }
    size_t write_str(const std::string &s)
    {
        put(s.data(), s.size());
        uint8_t z = 0;
        put(&z, 1);
        return s.size() + 1;
    }
    void add_part_now(int id, const bytes_t &d, uint64_t meta)
    {
        streams[id].parts.push_back({f_offset, d.size()});
        f_offset += write_num(meta);
        put(d.data(), d.size());
        f_offset += d.size();
    }

public:
    uint64_t bytes_written() const { return f_offset; }

    // path empty => discard the bytes (bench mode); "-" => stdout
    bool open(const std::string &path)
    {
        if (path.empty()) {
            f = nullptr;
            return true;
        }
        if (path == "-") {
            f = stdout;
            return true;
        }
        f = fopen(path.c_str(), "wb");
        own = true;
        return f != nullptr;
    }
    int register_stream(const std::string &name)
    {
        std::lock_guard<std::mutex> lk(mtx);
        auto p = ids.find(name);
        if (p != ids.end())
            return p->second;
        int id = (int)streams.size();
        streams.emplace_back();
        streams.back().name = name;
        ids[name] = id;
        return id;
    }
    int stream_id(const std::string &name)
    {
        std::lock_guard<std::mutex> lk(mtx);
        auto p = ids.find(name);
        return p == ids.end() ? -1 : p->second;
    }
    void add_part(int id, const bytes_t &d, uint64_t meta = 0)
    {
        std::lock_guard<std::mutex> lk(mtx);
        if (!events.empty())
            io_error = true; // (never: the callers drain first) an immediate part must not overtake a queued flush
        add_part_now(id, d, meta);
    }
    void add_part(int id, const uint8_t *d, size_t n, uint64_t meta)
    {
        std::lock_guard<std::mutex> lk(mtx);
        if (!events.empty())
            io_error = true;
        streams[id].parts.push_back({f_offset, n});
        f_offset += write_num(meta);
        put(d, n);
        f_offset += n;
    }
    void add_part_buffered(int id, bytes_t &&d, uint64_t meta)
    {
        std::lock_guard<std::mutex> lk(mtx);
        buffer[id].push_back(BufPart{std::move(d), meta, nullptr});
    }
    void add_part_deferred(int id, const std::shared_ptr<PartSlot> &slot)
    {
        std::lock_guard<std::mutex> lk(mtx);
        buffer[id].push_back(BufPart{bytes_t(), 0, slot});
        buffer_deferred = true;
    }
    void add_parts_deferred(const std::vector<std::pair<int, std::shared_ptr<PartSlot>>> &places)
    {
        std::lock_guard<std::mutex> lk(mtx);
        auto hint = buffer.end();
        for (auto &pl : places) { // (ids mostly ascending: the hint makes the insertion O(1))
            hint = buffer.emplace_hint(hint, pl.first, std::vector<BufPart>());
            hint->second.push_back(BufPart{bytes_t(), 0, pl.second});
            ++hint;
        }
        if (!places.empty())
            buffer_deferred = true;
    }

private:
    void write_event(buffer_t &b)
    {
        for (auto &x : b)
            for (auto &y : x.second) {
                if (y.slot)
                    add_part_now(x.first, y.slot->out, y.slot->meta);
                else
                    add_part_now(x.first, y.d, y.meta);
            }
    }
    static bool event_ready(const buffer_t &b)
    {
        for (auto &x : b)
            for (auto &y : x.second)
                if (y.slot && !y.slot->ready.load(std::memory_order_acquire))
                    return false;
        return true;
    }
    void drain_locked()
    {
        while (!events.empty() && event_ready(events.front())) {
            write_event(events.front());
            if (events.front().size() >= 4096) {
                // (the flush of a Close holds one part per group: 50 k map nodes, slots and payloads to free -- 20 ms at human
                // scale, on a thread of its own while the caller writes the archive's last streams)
                auto *dead = new buffer_t(std::move(events.front()));
                reapers.emplace_back([dead] { delete dead; });
            }
            events.pop_front();
        }
    }
    std::vector<std::thread> reapers;
    void join_reapers()
    {
        for (auto &t : reapers)
            if (t.joinable())
                t.join();
        reapers.clear();
    }

public:
    // archive.cpp:332-351.  With deferred parts in flight the flush is queued; queued flushes are written in order as soon as
    // their payloads are there (try_drain), so the file is laid out exactly as if every part had been ready at once
    void flush_out_buffers()
    {
        std::lock_guard<std::mutex> lk(mtx);
        if (!buffer.empty()) {
            if (events.empty() && !buffer_deferred)
                write_event(buffer);
            else
                events.emplace_back(std::move(buffer));
            buffer.clear();
            buffer_deferred = false;
        }
        drain_locked();
    }
    void try_drain()
    {
        std::lock_guard<std::mutex> lk(mtx);
        drain_locked();
    }
    bool pending_events()
    {
        std::lock_guard<std::mutex> lk(mtx);
        return !events.empty();
    }
    // ~CArchive -> Close: flush buffered parts, footer, 8-byte footer size (archive.cpp:68-85)
    ~ArchiveWriter() { join_reapers(); }
    void close()
    {
        flush_out_buffers();
        size_t fs = 0;
        fs += write_num(streams.size());
        for (auto &s : streams) {
            fs += write_str(s.name);
            fs += write_num(s.parts.size());
            fs += write_num(s.raw_size);
            for (auto &p : s.parts) {
                fs += write_num(p.offset);
                fs += write_num(p.size);
            }
        }
        uint8_t le[8];
        for (int i = 0; i < 8; ++i)
            le[i] = (uint8_t)(((uint64_t)fs >> (8 * i)) & 0xff);
        put(le, 8);
        f_offset += fs + 8;
        flush_file();
        if (f && own) {
            if (fclose(f) != 0)
                io_error = true;
        } else if (f && fflush(f) != 0)
            io_error = true;
        f = nullptr; // (the reapers are joined by the destructor: freeing what has been written is nobody's wait)
    }
};

// stream names: src/common/utils.cpp:33-84 (little-endian base-64 digits)
inline std::string int_to_base64(uint32_t n)
{
    static const char dig[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_#";
    std::string r;
    do {
        r.push_back(dig[n & 0x3fu]);
        n /= 64;
    } while (n);
    return r;
}
inline std::string ss_ref_name(uint32_t n) { return "x" + int_to_base64(n) + "r"; }
inline std::string ss_delta_name(uint32_t n) { return "x" + int_to_base64(n) + "d"; }

// ---------------------------------------------------------------------------
// Collection metadata, archive format v3 (write side).  src/common/collection_v3.cpp,
// src/common/collection.h:100-160 (prefix varint), src/common/utils.h:113-122 (zigzag vs prediction)
// ---------------------------------------------------------------------------
class CollectionV3 {
public:
    struct SegmentDesc {
        uint32_t group_id = ~0u, in_group_id = ~0u, raw_length = 0;
        bool is_rev_comp = false;
    };
    struct ContigDesc {
        std::string name;
        std::vector<SegmentDesc> segments;
    };
    struct SampleDesc {
        std::string name;
        std::vector<ContigDesc> contigs;
    };

private:
    std::vector<SampleDesc> samples;
    std::unordered_map<std::string, uint32_t> sample_ids;
    std::string prev_sample_name;
    ArchiveWriter *ar = nullptr;
    ZstdApi *z = nullptr;
    int id_samples = -1, id_contigs = -1, id_details = -1;
    uint32_t segment_size = 0, kmer_length = 0;
    std::vector<int> in_group_ids;

    static void app_str(bytes_t &d, const std::string &s)
    {
        d.insert(d.end(), s.begin(), s.end());
        d.push_back(0);
    }
    static void app_num(bytes_t &d, uint32_t num)
    {
        const uint32_t thr_1 = 1u << 7, thr_2 = thr_1 + (1u << 14), thr_3 = thr_2 + (1u << 21), thr_4 = thr_3 + (1u << 28);
        if (num < thr_1)
            d.push_back((uint8_t)num);
        else if (num < thr_2) {
            num -= thr_1;
            d.push_back((uint8_t)(0x80u + (num >> 8)));
            d.push_back((uint8_t)(num & 0xffu));
        } else if (num < thr_3) {
            num -= thr_2;
            d.push_back((uint8_t)(0xC0u + (num >> 16)));
            d.push_back((uint8_t)((num >> 8) & 0xffu));
            d.push_back((uint8_t)(num & 0xffu));
        } else if (num < thr_4) {
            num -= thr_3;
            d.push_back((uint8_t)(0xE0u + (num >> 24)));
            d.push_back((uint8_t)((num >> 16) & 0xffu));
            d.push_back((uint8_t)((num >> 8) & 0xffu));
            d.push_back((uint8_t)(num & 0xffu));
        } else {
            num -= thr_4;
            d.push_back(0xF0u);
            d.push_back((uint8_t)((num >> 24) & 0xffu));
            d.push_back((uint8_t)((num >> 16) & 0xffu));
            d.push_back((uint8_t)((num >> 8) & 0xffu));
            d.push_back((uint8_t)(num & 0xffu));
        }
    }
    static uint64_t zigzag_pred(uint64_t cur, uint64_t prev)
    {
        if (cur < prev)
            return 2 * (prev - cur) - 1u;
        if (cur < 2 * prev)
            return 2 * (cur - prev);
        return cur;
    }
    bytes_t zstd(const bytes_t &in, int level)
    {
        ZstdCtx c(z);
        bytes_t out(z->compressBound(in.size()));
        size_t n = c.compress(out.data(), out.size(), in.data(), in.size(), level);
        out.resize(n);
        return out;
    }
    static std::vector<std::string> split_string(const std::string &s)
    {
        std::vector<std::string> c;
        size_t p = 0;
        for (;;) {
            size_t q = s.find(' ', p);
            if (q == std::string::npos) {
                c.push_back(s.substr(p));
                break;
            }
            c.push_back(s.substr(p, q - p));
            p = q + 1;
        }
        return c;
    }
    // collection_v3.cpp:369-421
    static std::string encode_split(const std::vector<std::string> &prev, const std::vector<std::string> &cur)
    {
        std::string enc;
        for (size_t i = 0; i < cur.size(); ++i) {
            if (prev[i] == cur[i])
                enc.push_back((char)-127);
            else if (prev[i].size() != cur[i].size())
                enc.append(cur[i]);
            else {
                signed char cnt = 0;
                for (size_t j = 0; j < cur[i].size(); ++j) {
                    if (prev[i][j] == cur[i][j]) {
                        if (cnt == 100) {
                            enc.push_back((char)-cnt);
                            cnt = 1;
                        } else
                            ++cnt;
                    } else {
                        if (cnt) {
                            enc.push_back((char)-cnt);
                            cnt = 0;
                        }
                        enc.push_back(cur[i][j]);
                    }
                }
                if (cnt)
                    enc.push_back((char)-cnt);
            }
            enc.push_back(' ');
        }
        enc.pop_back();
        return enc;
    }

public:
    // short contig name: up to the first white space (collection.cpp:19-28)
    static std::string extract_contig_name(const std::string &s)
    {
        size_t p = 0;
        for (; p < s.size(); ++p)
            if (s[p] == ' ' || s[p] == '\n' || s[p] == '\r' || s[p] == '\t')
                break;
        return s.substr(0, p);
    }

    // prepare_for_compression (collection_v3.cpp:37-44): these three streams get ids 0,1,2
    void set_archive(ArchiveWriter *a, ZstdApi *zz, uint32_t seg_size, uint32_t k)
    {
        ar = a;
        z = zz;
        segment_size = seg_size;
        kmer_length = k;
        id_samples = ar->register_stream("collection-samples");
        id_contigs = ar->register_stream("collection-contigs");
        id_details = ar->register_stream("collection-details");
    }
    void reset_prev_sample_name() { prev_sample_name.clear(); }
    size_t no_samples() const { return samples.size(); }
    // append mode (prepare_for_appending_copy / _load_last_batch, collection_v3.cpp:47-108): every sample name of the
    // input archive; contigs + segments only for the samples of its last, still open batch
    void load_sample_names(const std::vector<std::string> &names)
    {
        samples.clear();
        sample_ids.clear();
        for (auto &n : names) {
            sample_ids[n] = (uint32_t)samples.size();
            samples.emplace_back();
            samples.back().name = n;
        }
    }
    SampleDesc &sample_at(size_t i) { return samples[i]; }
    int stream_contigs() const { return id_contigs; }
    int stream_details() const { return id_details; }

    // collection_v3.cpp:682-708
    bool register_sample_contig(const std::string &sample_name, const std::string &contig_name)
    {
        std::string stored = sample_name.empty() ? extract_contig_name(contig_name) : sample_name;
        if (stored != prev_sample_name) {
            if (sample_ids.count(stored))
                return false;
            uint32_t id = (uint32_t)sample_ids.size();
            sample_ids[stored] = id;
            samples.emplace_back();
            samples.back().name = stored;
            prev_sample_name = stored;
        }
        samples.back().contigs.emplace_back();
        samples.back().contigs.back().name = contig_name;
        return true;
    }
    // collection_v3.cpp:773-805
    void add_segment_placed(const std::string &sample_name, const std::string &contig_name, uint32_t part_no, uint32_t gid,
                            uint32_t in_gid, bool rc, uint32_t raw_len)
    {
        std::string stored = sample_name.empty() ? extract_contig_name(contig_name) : sample_name;
        auto p = sample_ids.find(stored);
        if (p == sample_ids.end())
            return;
        for (auto &c : samples[p->second].contigs)
            if (c.name == contig_name) {
                if (part_no >= c.segments.size())
                    c.segments.resize((size_t)part_no + 1);
                c.segments[part_no] = {gid, in_gid, raw_len, rc};
                break;
            }
    }
    // direct access for the compressor (avoids the linear contig search per segment)
    SampleDesc &sample_by_name(const std::string &stored) { return samples[sample_ids.at(stored)]; }

    // store_contig_batch (collection_v3.cpp:660-680): names (zstd 18) then details (5 x zstd 19).  In two halves for a caller that
    // shares the sample table with another thread: serialize_contig_batch reads (and clears) the samples' records -- short, under
    // the caller's lock --, store_serialized_batch compresses (0.4 s at human scale) and adds the parts -- outside it.
    struct SerializedBatch {
        bytes_t names;
        std::array<bytes_t, 5> details;
    };
    void store_contig_batch(uint32_t id_from, uint32_t id_to)
    {
        SerializedBatch sb;
        serialize_contig_batch(id_from, id_to, sb);
        store_serialized_batch(sb);
    }
    void serialize_contig_batch(uint32_t id_from, uint32_t id_to, SerializedBatch &sb)
    {
        // ---- contig names, collection_v3.cpp:468-495
        bytes_t &v = sb.names;
        app_num(v, id_to - id_from);
        for (uint32_t s = id_from; s < id_to; ++s) {
            app_num(v, (uint32_t)samples[s].contigs.size());
            std::vector<std::string> prev;
            for (auto &c : samples[s].contigs) {
                auto cur = split_string(c.name);
                if (cur.size() != prev.size())
                    app_str(v, c.name);
                else
                    app_str(v, encode_split(prev, cur));
                prev = std::move(cur);
            }
        }
        // ---- details, collection_v3.cpp:539-586 + 230-267
        std::array<bytes_t, 5> &d = sb.details;
        app_num(d[0], id_to - id_from);
        in_group_ids.clear();
        auto get_igid = [&](uint32_t pos) -> int { return pos >= in_group_ids.size() ? -1 : in_group_ids[pos]; };
        auto set_igid = [&](uint32_t pos, int val) {
            if (pos >= in_group_ids.size())
                in_group_ids.resize((size_t)((int)(pos * 1.2) + 1), -1);
            in_group_ids[pos] = val;
        };
        for (uint32_t s = id_from; s < id_to; ++s) {
            app_num(d[0], (uint32_t)samples[s].contigs.size());
            const uint32_t pred_raw_length = segment_size + kmer_length;
            for (auto &c : samples[s].contigs) {
                app_num(d[0], (uint32_t)c.segments.size());
                for (auto &seg : c.segments) {
                    int prev = get_igid(seg.group_id);
                    uint32_t e_in;
                    if (prev == -1)
                        e_in = seg.in_group_id;
                    else if (seg.in_group_id == 0)
                        e_in = 0;
                    else if ((int)seg.in_group_id == prev + 1)
                        e_in = 1;
                    else
                        e_in = (uint32_t)zigzag_pred(seg.in_group_id, (uint64_t)(prev + 1)) + 1u;
                    app_num(d[1], seg.group_id);
                    app_num(d[2], e_in);
                    app_num(d[3], (uint32_t)zigzag_pred(seg.raw_length, pred_raw_length));
                    app_num(d[4], (uint32_t)seg.is_rev_comp);
                    if ((int)seg.in_group_id > prev && seg.in_group_id > 0)
                        set_igid(seg.group_id, (int)seg.in_group_id);
                }
            }
        }
        for (uint32_t s = id_from; s < id_to; ++s) {
            samples[s].contigs.clear();
            samples[s].contigs.shrink_to_fit();
        }
    }
    void store_serialized_batch(SerializedBatch &sb)
    {
        ar->add_part_buffered(id_contigs, zstd(sb.names, 18), sb.names.size());
        std::array<bytes_t, 5> &d = sb.details;
        std::array<bytes_t, 5> pk;
        for (int i = 0; i < 5; ++i)
            pk[i] = zstd(d[i], 19);
        bytes_t stream;
        for (int i = 0; i < 5; ++i) {
            app_num(stream, (uint32_t)d[i].size());
            app_num(stream, (uint32_t)pk[i].size());
        }
        for (int i = 0; i < 5; ++i)
            stream.insert(stream.end(), pk[i].begin(), pk[i].end());
        ar->add_part_buffered(id_details, std::move(stream), 0);
    }
    // complete_serialization -> store_batch_sample_names (collection_v3.cpp:122-165, 329-335)
    void complete_serialization()
    {
        bytes_t v;
        app_num(v, (uint32_t)samples.size());
        for (auto &s : samples)
            app_str(v, s.name);
        ar->add_part_buffered(id_samples, zstd(v, 19), v.size());
    }
};

} // namespace agc
