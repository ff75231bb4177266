// compressor_batch.cpp -- the per-window pipeline of CAGCCompressor (see compressor.h, compressor_impl.h):
// scan -> classification -> placement -> [speculative encode] -> commit runs (registration, store, bookkeeping).
// Citations: file:line under the reference tree.
#include "compressor_impl.h"
#include <atomic>

namespace agc {

// ---------------------------------------------------------------------------
// store_in_archive(pack), segment.h:258-280: sequences + 0xFF separators -> zstd 17
void CAGCCompressor::Impl::make_pack_job(std::vector<ZJob> &jobs, uint32_t gid, bytes_t &data, std::vector<uint32_t> &off)
{
    Group &g = groups[gid];
    ZJob j;
    j.gid = gid;
    if (g.stream_delta < 0 && !defer_stream_reg) // segment.h:262-266
        g.stream_delta = ar.register_stream(ss_delta_name(j.gid));
    j.stream_id = g.stream_delta; // (< 0: registered by the caller, in list order)
    j.kind = 1;
    j.data.swap(data);
    data.clear();
    off.clear();
    jobs.emplace_back(std::move(j));
}

// add_to_archive / add_to_archive_tuples, segment.h:172-215; store_in_archive(ref) :218-255
// Delta packs (zstd level 17, the bulk of the bytes) are compressed on the GPU, one frame per lane
// (agc_hip_zstd17_batch: byte-identical to ZSTD_compressCCtx of libzstd 1.4.x); references (tuples + level 13, or level
// 19), packs beyond one zstd block and everything when the library in use is not 1.4.x go to libzstd on the host pool,
// at the same time.
void CAGCCompressor::Impl::run_jobs(std::vector<ZJob> &jobs, bool add_parts)
{
    // (one round: the device's time for a launch is close to the serial time of ONE frame whatever the number of frames --
    // rounds only add up)
    z_wait_all(); // zpool, the zstd contexts and the staging buffers are the entropy thread's while it works
    agc_hip_zstd17_background(hip, 0);
    run_jobs_round(jobs);
    if (add_parts)
        add_job_parts(jobs, 0, jobs.size());
}

// ---- the asynchronous entropy stage ----------------------------------------
void CAGCCompressor::Impl::z_submit(std::vector<ZJob> &&jobs)
{
    if (jobs.empty())
        return;
    {
        std::lock_guard<std::mutex> lk(z_mtx);
        if (!z_thread.joinable())
            z_thread = std::thread([this] { z_main(); });
        z_queue.emplace_back(std::move(jobs));
    }
    z_cv.notify_all();
}

void CAGCCompressor::Impl::z_main()
{
    for (;;) {
        std::vector<ZJob> batch;
        {
            std::unique_lock<std::mutex> lk(z_mtx);
            z_cv.wait(lk, [&] { return z_stop || !z_queue.empty(); });
            if (z_queue.empty())
                return; // (z_stop)
            batch = std::move(z_queue.front());
            z_queue.pop_front();
            // small registrations pile up while a big one is in the works: take them together (one device launch)
            while (!z_queue.empty() && batch.size() < 4096) {
                for (auto &j : z_queue.front())
                    batch.emplace_back(std::move(j));
                z_queue.pop_front();
            }
            z_busy = true;
        }
        if (entropy_stream && !heavy_steps.load() && host_only_batch(batch)) {
            run_host_stream(std::move(batch)); // (publishes every part as it is finished)
            batch.clear();
        } else {
            agc_hip_zstd17_background(hip, z_caller_waits.load() ? 0 : 1); // beside the steps: leave the LDS to the scan
            run_jobs_round(batch);
        }
        for (ZJob &j : batch) {
            st.zstd_in += j.data.size();
            st.zstd_out += j.out.size();
            j.slot->out = std::move(j.out);
            j.slot->meta = j.meta;
            j.slot->ready.store(true, std::memory_order_release);
        }
        {
            std::lock_guard<std::mutex> lk(z_mtx);
            z_busy = false;
        }
        z_idle_cv.notify_all();
        batch.clear(); // (the inputs -- half a GB at a human Close -- are freed while whoever waited goes on)
    }
}

// A batch run_jobs_round would hand to the host pool whole: no device entropy stage, or fewer device-size jobs than a launch is
// worth (the rule of run_jobs_round: dev_jobs.size() < gpu_zstd_min).
bool CAGCCompressor::Impl::host_only_batch(const std::vector<ZJob> &jobs) const
{
    if (!gpu_zstd)
        return true;
    const uint32_t dev_max = agc_hip_zstd17_max_input();
    size_t n_refs = 0, n_packs = 0;
    for (const ZJob &j : jobs) {
        n_refs += j.kind == 0 && !j.data.empty() && j.data.size() <= 65536;
        n_packs += j.kind == 1 && !j.data.empty() && j.data.size() <= dev_max;
    }
    const bool refs_too = gpu_zstd_refs_min && n_refs >= gpu_zstd_refs_min;
    return n_packs + (refs_too ? n_refs : 0) < gpu_zstd_min;
}

// add_to_archive / add_to_archive_tuples / store_in_archive(ref) on the host (segment.h:172-255): one job, libzstd
void CAGCCompressor::Impl::host_compress(ZJob &j, unsigned tid)
{
    ZstdCtx &z = *zctx[tid];
    const bytes_t *src = &j.data;
    bytes_t tuples;
    int level = 17;
    uint8_t marker = 0;
    if (j.kind == 0) {
        if (!j.repetitive) {
            if (!j.staged.empty())
                src = &j.staged; // (packed for the device, which then left it to the pool)
            else {
                bytes2tuples(j.data, tuples);
                src = &tuples;
            }
            level = 13;
            marker = 1;
        } else
            level = 19;
    }
    const size_t bound = zstd.compressBound(src->size());
    bytes_t packed(bound + 1);
    const uint32_t ps = (uint32_t)z.compress(packed.data(), bound, src->data(), src->size(), level);
    packed[ps] = marker;
    if (ps + 1u < (uint32_t)j.data.size()) {
        packed.resize((size_t)ps + 1);
        j.out = std::move(packed);
        j.meta = j.data.size();
    } else {
        j.out = j.data;
        j.meta = 0;
    }
}

// Host-only batches (collections of small genomes, packs beyond one zstd block): a pack of a few hundred KB is a few tenths of a
// second of level 17 for ONE thread, and a registration fills only a few packs -- a parallel_for per batch leaves most of the
// pool idle behind its longest job while the next batches queue up (configs[1]: 6 batches of 1, 3, 5, 6, 21 jobs one after the
// other).  Here the pool's threads take jobs from one list that later batches join as they are submitted; the stream ends when
// every thread is idle and the queue holds nothing for it.  The parts were given their places in the archive at submission
// (PartSlot), so the order of completion is free.
void CAGCCompressor::Impl::run_host_stream(std::vector<ZJob> &&first)
{
    const double t0 = now();
    std::deque<ZJob> pending;
    const unsigned nw = zpool->size();
    unsigned waiting = 0;
    bool done = false, stop_pulling = false;
    auto push_batch = [&](std::vector<ZJob> &b) { // longest jobs first (the tail is then made of short ones)
        std::stable_sort(b.begin(), b.end(), [](const ZJob &x, const ZJob &y) { return x.data.size() > y.data.size(); });
        for (ZJob &j : b)
            pending.emplace_back(std::move(j));
        b.clear();
    };
    push_batch(first);
    zpool->parallel_for(nw, [&](size_t, unsigned tid) {
        std::unique_lock<std::mutex> lk(z_mtx);
        for (;;) {
            while (!stop_pulling && !z_queue.empty()) {
                if (!host_only_batch(z_queue.front())) { // the device's: the stream ends first
                    stop_pulling = true;
                    break;
                }
                push_batch(z_queue.front());
                z_queue.pop_front();
                z_cv.notify_all();
            }
            if (!pending.empty()) {
                ZJob j = std::move(pending.front());
                pending.pop_front();
                lk.unlock();
                host_compress(j, tid);
                const uint64_t n_in = j.data.size(), n_out = j.out.size();
                j.slot->out = std::move(j.out);
                j.slot->meta = j.meta;
                j.slot->ready.store(true, std::memory_order_release);
                bytes_t().swap(j.data);
                lk.lock();
                st.zstd_in += n_in;
                st.zstd_out += n_out;
                continue;
            }
            if (done)
                return;
            if (waiting + 1 == nw) { // everybody else is idle too
                done = true;
                z_cv.notify_all();
                return;
            }
            ++waiting;
            z_cv.wait(lk, [&] { return done || !pending.empty() || (!stop_pulling && !z_queue.empty()); });
            --waiting;
        }
    });
    const double dt = now() - t0;
    st.t_zstd_host += dt;
    st.t_zstd += dt;
}

void CAGCCompressor::Impl::z_wait_all()
{
    if (!z_thread.joinable())
        return;
    const double t0 = now();
    z_caller_waits = true;
    {
        std::unique_lock<std::mutex> lk(z_mtx);
        z_idle_cv.wait(lk, [&] { return z_queue.empty() && !z_busy; });
    }
    z_caller_waits = false;
    st.t_zstd_wait += now() - t0;
    ar.try_drain();
}

void CAGCCompressor::Impl::z_shutdown()
{
    if (!z_thread.joinable())
        return;
    {
        std::lock_guard<std::mutex> lk(z_mtx);
        z_stop = true;
    }
    z_cv.notify_all();
    z_thread.join();
}

// ---- the asynchronous bookkeeping stage (compressor_impl.h) ------------------
uint64_t CAGCCompressor::Impl::book_submit(std::unique_ptr<BookTask> &&t)
{
    uint64_t seq;
    {
        std::unique_lock<std::mutex> lk(book_mtx);
        // (a task holds up to a sample's deltas: the queue stays short)
        book_idle_cv.wait(lk, [&] { return book_queue.size() < 4; });
        if (!book_thread.joinable())
            book_thread = std::thread([this] { book_main(); });
        book_queue.emplace_back(std::move(t));
        seq = ++book_seq_submitted;
    }
    book_cv.notify_all();
    return seq;
}

// room for the deltas of `text` symbols left in flight on a lane: 1/64 + 1/512 of the text is never reached by related genomes; a
// first allocation takes a quarter more (the text the device knows grows by a percent or two from sample to sample for the first
// samples of a collection, and every new maximum is a hipHostFree + hipHostMalloc of tens of MB on the thread that holds the lane:
// 2.5 ms in each of the bench's first two timed steps); a buffer that does not hold the deltas after all answers ECAP
static uint64_t delta_cap(uint64_t have, uint64_t text)
{
    const uint64_t want = text / 64 + text / 512 + (1u << 16);
    if (have >= want)
        return have;
    return have == 0 ? want + want / 4 : want;
}

void CAGCCompressor::Impl::book_main()
{
    for (;;) {
        std::unique_ptr<BookTask> t;
        {
            std::unique_lock<std::mutex> lk(book_mtx);
            book_cv.wait(lk, [&] { return book_stop || !book_queue.empty(); });
            if (book_queue.empty())
                return; // (book_stop)
            t = std::move(book_queue.front());
            book_queue.pop_front();
            book_busy = true;
        }
        const double t0 = now();
        bool ok = true;
        uint64_t delta_bytes = 0;
        // the registration's LZ encodes were left in flight on the device's lanes: collect them (only a lane's own state is touched)
        auto collect = [&](uint32_t lane, const std::vector<uint32_t> &todo, uint64_t text, PinnedBytes &enc) {
            const size_t ne = todo.size();
            std::vector<uint64_t> eoff(ne + 1, 0);
            {
                uint32_t n_dev = 0;
                ok = hip_ok(agc_hip_lz_encode_pending_on(hip, lane, &n_dev), "lz_encode_pending");
                if (ok && n_dev != ne) {
                    err("internal: the device's encode delivers another number of deltas than the host expects");
                    ok = false;
                }
            }
            if (ok) {
                uint64_t cap = delta_cap(enc.size(), text);
                for (;;) {
                    if (!enc.resize(cap, false)) {
                        err("out of memory (delta buffer)");
                        ok = false;
                        break;
                    }
                    const int r = agc_hip_lz_encode_end_on(hip, lane, enc.data(), cap, eoff.data());
                    if (r == AGC_HIP_ECAP) {
                        cap = eoff[ne] + eoff[ne] / 8 + 4096; // (headroom: the next sample's deltas are a little longer)
                        continue;
                    }
                    ok = hip_ok(r, "lz_encode_end");
                    break;
                }
            }
            lane2_release((int)lane); // (the thread that drives the steps may launch the next sample's encode)
            if (ok) {
                for (size_t i = 0; i < ne; ++i) {
                    if (todo[i] == ~0u)
                        continue; // (a delta the device made ahead of the classification and the placement did not use)
                    t->cd.enc_ptr[todo[i]] = enc.data() + eoff[i];
                    t->cd.enc_len[todo[i]] = (uint32_t)(eoff[i + 1] - eoff[i]);
                }
                delta_bytes += eoff[ne];
            }
        };
        if (t->early_only) {
            // lane 0's deltas as they are, nothing mapped: the registration's own task does that (enc_collected)
            const size_t ne = t->enc_n;
            std::vector<uint64_t> eoff(ne + 1, 0);
            uint32_t n_dev = 0;
            ok = hip_ok(agc_hip_lz_encode_pending_on(hip, 0, &n_dev), "lz_encode_pending");
            if (ok && n_dev != ne) {
                err("internal: the device's encode delivers another number of deltas than the host expects");
                ok = false;
            }
            if (ok) {
                PinnedBytes &enc = *t->enc_dst;
                uint64_t cap = delta_cap(enc.size(), t->enc_text);
                for (;;) {
                    if (!enc.resize(cap, false)) {
                        err("out of memory (delta buffer)");
                        ok = false;
                        break;
                    }
                    const int r = agc_hip_lz_encode_end_on(hip, 0, enc.data(), cap, eoff.data());
                    if (r == AGC_HIP_ECAP) {
                        cap = eoff[ne] + eoff[ne] / 8 + 4096;
                        continue;
                    }
                    ok = hip_ok(r, "lz_encode_end");
                    break;
                }
            }
            lane2_release(0);
            static const bool early_laps = getenv("AGC_AMD_LAPS") != nullptr;
            if (early_laps)
                std::cerr << "    book task (early): lane 0 collected in " << (now() - t0) * 1e3 << " ms\n";
            t.reset();
            {
                std::lock_guard<std::mutex> lk(book_mtx);
                early_enc.ok = ok;
                early_enc.eoff.swap(eoff);
                book_busy = false;
                ++book_seq_done;
                book_seconds += now() - t0;
                if (ok)
                    book_delta_bytes += early_enc.eoff[ne];
                else
                    book_failed = true;
            }
            book_idle_cv.notify_all();
            continue;
        }
        if (t->enc_collected) {
            const std::vector<uint32_t> &todo = t->enc_todo;
            const uint8_t *base = t->enc_dst->data();
            for (size_t i = 0; i < todo.size(); ++i) {
                if (todo[i] == ~0u)
                    continue;
                t->cd.enc_ptr[todo[i]] = base + t->enc_eoff[i];
                t->cd.enc_len[todo[i]] = (uint32_t)(t->enc_eoff[i + 1] - t->enc_eoff[i]);
            }
        } else if (t->enc_pending)
            collect(0, t->enc_todo, t->enc_text, *t->enc_dst);
        const double t_c0 = now();
        if (t->enc2_pending) {
            if (ok)
                collect(1, t->enc2_todo, t->enc2_text, *t->enc2_dst);
            else
                lane2_release(1);
        }
        const double t_c1 = now();
        book_on_thread = true;
        ok = ok && finish_ref_store(t->cd, t->fetched);
        const double t_c2 = now();
        ok = ok && book_and_store(t->cd);
        const double t_c3 = now();
        book_on_thread = false;
        t.reset();
        static const bool book_laps = getenv("AGC_AMD_LAPS") != nullptr;
        if (book_laps) {
            char line[256];
            snprintf(line, sizeof line, "    book task: collect lane 0 %.3f ms, lane 1 %.3f ms, reference store awaited %.3f ms, books %.3f ms, task freed %.3f ms\n",
                     (t_c0 - t0) * 1e3, (t_c1 - t_c0) * 1e3, (t_c2 - t_c1) * 1e3, (t_c3 - t_c2) * 1e3, (now() - t_c3) * 1e3);
            std::cerr << line;
        }
        {
            std::lock_guard<std::mutex> lk(book_mtx);
            book_busy = false;
            ++book_seq_done;
            book_seconds += now() - t0;
            book_delta_bytes += delta_bytes;
            if (!ok)
                book_failed = true;
        }
        book_idle_cv.notify_all();
    }
}

// every queued registration is in the books; false: one of them failed (the message is out already)
bool CAGCCompressor::Impl::book_wait()
{
    if (!book_thread.joinable())
        return true;
    std::unique_lock<std::mutex> lk(book_mtx);
    book_idle_cv.wait(lk, [&] { return book_queue.empty() && !book_busy; });
    st.t_store += book_seconds;
    st.h_store += book_seconds;
    st.delta_bytes += book_delta_bytes;
    book_seconds = 0;
    book_delta_bytes = 0;
    return !book_failed;
}

bool CAGCCompressor::Impl::book_wait_seq(uint64_t seq)
{
    if (!seq || !book_thread.joinable())
        return true;
    std::unique_lock<std::mutex> lk(book_mtx);
    book_idle_cv.wait(lk, [&] { return book_seq_done >= seq; });
    return !book_failed;
}

void CAGCCompressor::Impl::book_shutdown()
{
    if (!book_thread.joinable())
        return;
    {
        std::lock_guard<std::mutex> lk(book_mtx);
        book_stop = true;
    }
    book_cv.notify_all();
    book_thread.join();
}

// the device's second LZ lane: one encode at a time
void CAGCCompressor::Impl::lane2_acquire(int lane)
{
    std::unique_lock<std::mutex> lk(book_mtx);
    book_idle_cv.wait(lk, [&] { return !lane_inflight[lane]; });
    lane_inflight[lane] = true;
}

void CAGCCompressor::Impl::lane2_release(int lane)
{
    {
        std::lock_guard<std::mutex> lk(book_mtx);
        lane_inflight[lane] = false;
    }
    book_idle_cv.notify_all();
}

void CAGCCompressor::Impl::run_jobs_round(std::vector<ZJob> &jobs)
{
    double t0 = now();
    static const bool laps = getenv("AGC_AMD_LAPS") != nullptr;
    double lt = t0;
    auto LAP = [&](const char *what) {
        if (laps && jobs.size() > 0)
            std::cerr << "    entropy lap " << what << " " << (now() - lt) * 1e3 << " ms\n";
        lt = now();
    };
    auto finish = [](ZJob &j, bytes_t &packed, uint32_t ps, uint8_t marker) {
        packed[ps] = marker;
        if (ps + 1u < (uint32_t)j.data.size()) {
            packed.resize((size_t)ps + 1);
            j.out = std::move(packed);
            j.meta = j.data.size();
        } else {
            j.out = j.data;
            j.meta = 0;
        }
    };
    // which jobs the device takes
    std::vector<uint32_t> dev_jobs, host_jobs;
    const uint32_t dev_max = gpu_zstd ? agc_hip_zstd17_max_input() : 0;
    const bool ext = close_collected && &jobs == &close_jobs; // frames of the collected packs came in through CloseProvideFrames
    if (ext) {
        std::vector<uint8_t> is_dev(jobs.size(), 0);
        for (uint32_t i : close_dev_jobs)
            is_dev[i] = 1;
        for (uint32_t i = 0; i < jobs.size(); ++i)
            (is_dev[i] ? dev_jobs : host_jobs).push_back(i);
    } else {
        // references: only when there are many of them (the reference sample), tuple-packed here (bytes2tuples is a byte loop)
        size_t n_refs = 0;
        for (const ZJob &j : jobs)
            n_refs += j.kind == 0 && !j.data.empty() && j.data.size() <= 65536; // (tuple-packed: <= 16 KiB, the lane-group kernel's class)
        // references: to the device when a call brings many of the size a group of lanes codes (the reference sample of a
        // collection with the default segment size); the few long ones -- contigs without a splitter, -s in the hundreds of kb:
        // level 13 above 16 KiB is the one-lane kernel, a launch as long as its longest frame -- stay with the host pool
        const bool refs_too = gpu_zstd && gpu_zstd_refs_min && n_refs >= gpu_zstd_refs_min;
        if (refs_too)
            zpool->parallel_for(jobs.size(), [&](size_t i, unsigned) {
                ZJob &j = jobs[i];
                if (j.kind != 0 || j.data.empty())
                    return;
                j.staged.clear();
                if (!j.repetitive) {
                    bytes2tuples(j.data, j.staged);
                    j.level = 13;
                    j.marker = 1;
                } else {
                    j.level = 19;
                    j.marker = 0;
                }
            });
        for (uint32_t i = 0; i < jobs.size(); ++i) {
            const ZJob &j = jobs[i];
            const size_t src_n = j.staged.empty() ? j.data.size() : j.staged.size();
            if (gpu_zstd && !j.data.empty() && src_n <= dev_max && (j.kind == 1 || (j.kind == 0 && refs_too && src_n <= 16384)))
                dev_jobs.push_back(i);
            else
                host_jobs.push_back(i);
        }
    }
    // A launch lasts about as long as the serial chain of its longest frame however few frames it has (measured: ~48 us per input
    // byte for the two-pass class <= 16 KiB, ~24 us beyond, a third of that for level 13), while the host pool finishes a small
    // batch in bytes / (threads x ~5 MB/s): the device only takes a batch the pool would need longer for than that.
    bool dev_pays = true;
    if (!ext && !dev_jobs.empty() && !getenv("AGC_AMD_GPU_ZSTD_SHARE") && gpu_zstd_min > 1) {
        uint64_t tot = 0;
        double longest = 0;
        for (uint32_t i : dev_jobs) {
            const uint64_t sz = jobs[i].staged.empty() ? jobs[i].data.size() : jobs[i].staged.size();
            tot += sz;
            double t = (sz <= 16384 ? 48e-6 : 24e-6) * (double)sz;
            if (jobs[i].kind == 0 && jobs[i].level == 13)
                t *= 0.3;
            longest = std::max(longest, t);
        }
        dev_pays = (double)tot / ((double)zpool->size() * 5e6) > longest;
    }
    if (ext) {
    } else if (dev_jobs.size() < gpu_zstd_min || !dev_pays) { // not worth a launch
        host_jobs.insert(host_jobs.end(), dev_jobs.begin(), dev_jobs.end());
        std::sort(host_jobs.begin(), host_jobs.end());
        dev_jobs.clear();
    } else if (gpu_zstd_share < 1.0 && zpool->size() > 1) {
        // both engines work at the same time: the device keeps the share of the pack bytes that makes them finish together
        // (its measured rate against the host pool's, updated after every call)
        uint64_t total = 0, dev_acc = 0;
        auto src_size = [&](uint32_t i) -> uint64_t { return jobs[i].staged.empty() ? jobs[i].data.size() : jobs[i].staged.size(); }; // what the encoder reads
        for (uint32_t i : dev_jobs)
            total += src_size(i);
        // (length of a frame's serial chain: level 17 parses inputs of up to 16 KB twice -- btultra2 -- and larger ones once,
        // with a deeper search: 150 against 200 per position, from the measured launch times of the two classes)
        const uint64_t w_big = 150;
        auto chain = [&](uint32_t i) -> uint64_t {
            const uint64_t sz = src_size(i);
            if (jobs[i].kind == 0 && jobs[i].level == 13) // one pass, a shallow search (searchLog 5 / 3)
                return 60 * sz;
            return sz <= 16384 || jobs[i].level == 19 ? 200 * sz : w_big * sz; // (level 19: two passes whatever the size)
        };
        std::vector<uint32_t> by_size(dev_jobs);
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) { return chain(a) < chain(b); });
        // The device works on agc_hip_zstd17_resident_frames() frames at once and a launch lasts about as long as its longest
        // frame whatever their number (a frame is a serial chain; a second round would last as long as the first): when there are
        // more packs than that, the host pool takes the SMALLEST ones (the fewest bytes per frame taken off the device); then the
        // largest ones move over until the device's byte share is the one that lets both sides finish together.
        // (frames beyond one resident round were measured in round 3: +2 500 of the 50 k packs of a human Close() take the device
        // from 0.64 to 0.99 s -- profiles/r3/bookkeeping_thread_and_extra_frames.txt)
        const uint32_t resident = std::max<uint32_t>(1u, agc_hip_zstd17_resident_frames(hip));
        size_t lo = 0, hi = by_size.size();
        // (a handful of packs beyond 16 KiB -- the one-lane kernel's class, a launch of its own that lasts as long as a whole
        // launch of small frames -- is the host pool's: by_size ends with them)
        {
            size_t big = 0;
            while (big < hi && src_size(by_size[hi - 1 - big]) > 16384)
                ++big;
            if (big < 512)
                hi -= big;
        }
        if (hi - lo > resident)
            lo = hi - resident;
        dev_acc = 0;
        for (size_t t = lo; t < hi; ++t)
            dev_acc += src_size(by_size[t]);
        while (hi > lo && (double)dev_acc > gpu_zstd_share * (double)total) {
            --hi;
            dev_acc -= src_size(by_size[hi]);
        }
        std::vector<uint32_t> keep(by_size.begin() + lo, by_size.begin() + hi);
        host_jobs.insert(host_jobs.end(), by_size.begin(), by_size.begin() + lo);
        host_jobs.insert(host_jobs.end(), by_size.begin() + hi, by_size.end());
        std::sort(keep.begin(), keep.end());
        dev_jobs.swap(keep);
    }
    // host pool: longest jobs first (the tail of the pool is then made of short ones)
    std::stable_sort(host_jobs.begin(), host_jobs.end(), [&](uint32_t a, uint32_t b) { return jobs[a].data.size() > jobs[b].data.size(); });
    uint64_t host_bytes = 0;
    for (uint32_t i : host_jobs)
        if (jobs[i].kind == 1)
            host_bytes += jobs[i].data.size();
    double t_dev = 0, t_host = 0;
    if (laps && jobs.size() > 0) {
        uint64_t nb[4] = {0, 0, 0, 0}, nc[4] = {0, 0, 0, 0};
        for (const ZJob &j : jobs) {
            const int c = j.kind == 0 ? 0 : j.data.size() > dev_max ? 1 : j.data.size() > 16384 ? 2 : 3;
            nb[c] += j.data.size();
            ++nc[c];
        }
        std::cerr << "    entropy jobs: refs " << nc[0] << " (" << nb[0] / 1e6 << " MB), packs > 128 KiB " << nc[1] << " (" << nb[1] / 1e6 << " MB), packs 16-128 KiB "
                  << nc[2] << " (" << nb[2] / 1e6 << " MB), packs <= 16 KiB " << nc[3] << " (" << nb[3] / 1e6 << " MB); device gets " << dev_jobs.size() << "\n";
    }
    LAP("split");
    std::future<bool> dev_done;
    bool dev_results_done = false; // (written by the device's thread, read after dev_done.get())
    std::vector<uint64_t> src_off, dst_off;
    const double ts0 = now();
    if (ext) {
        dst_off = close_frames_off; // (zdst_buf holds the frames)
    } else if (!dev_jobs.empty()) {
        const size_t nd = dev_jobs.size();
        src_off.assign(nd + 1, 0);
        std::vector<uint8_t> levels(nd);
        bool any_ref = false;
        for (size_t t = 0; t < nd; ++t) {
            const ZJob &j = jobs[dev_jobs[t]];
            src_off[t + 1] = src_off[t] + (j.staged.empty() ? j.data.size() : j.staged.size());
            levels[t] = j.kind == 0 ? j.level : 17;
            any_ref = any_ref || j.kind == 0;
        }
        const uint64_t cap = src_off[nd] + 32 * nd + 64; // a frame never exceeds its input by more than the headers
        dst_off.assign(nd + 1, 0);
        // the packs are gathered by the whole pool (hundreds of MB into fresh pages: a tenth of a second for one thread; the pool's
        // own jobs wait the 7 ms this takes -- helper threads beside a pool that already uses the whole CPU quota were measured:
        // the device's side then starts late and ends 57 ms behind the pool), then the device call runs beside the pool's jobs
        // on a thread of its own, which also puts the frames back into their jobs as soon as the device is done
        const double tg = now();
        zsrc_buf.resize(src_off[nd] + src_off[nd] / 8, false); // (headroom: the next call's packs are a little longer)
        {
            const size_t n_chunks = std::min<size_t>(nd, (size_t)zpool->size() * 4);
            zpool->parallel_for(n_chunks, [&](size_t ci, unsigned) {
                for (size_t t = nd * ci / n_chunks; t < nd * (ci + 1) / n_chunks; ++t) {
                    const ZJob &j = jobs[dev_jobs[t]];
                    const bytes_t &srcb = j.staged.empty() ? j.data : j.staged;
                    memcpy(zsrc_buf.data() + src_off[t], srcb.data(), srcb.size());
                }
            });
        }
        const double t_gather = now() - tg;
        dev_done = std::async(std::launch::async, [&, nd, cap, t_gather, levels, any_ref] {
            const double td = now() - t_gather; // (the gather counts as device-side time for the split rule)
            auto helpers = [&](size_t n_items, const std::function<void(size_t, size_t)> &body) {
                const size_t nt = n_items >= 4096 ? 2 : 1;
                std::vector<std::thread> th;
                for (size_t t = 1; t < nt; ++t)
                    th.emplace_back([&, t] { body(n_items * t / nt, n_items * (t + 1) / nt); });
                body(0, n_items / nt);
                for (auto &x : th)
                    x.join();
            };
            zdst_buf.resize(cap + cap / 8, false);
            if (const char *dump = getenv("AGC_AMD_DUMP_PACKS")) { // debugging aid: the packs of this call, for scripts/zstd_gpu_probe.py
                static int dump_no = 0;
                const std::string base = std::string(dump) + "/packs_" + std::to_string(dump_no++);
                if (FILE *f = fopen((base + ".bin").c_str(), "wb")) {
                    fwrite(zsrc_buf.data(), 1, src_off[nd], f);
                    fclose(f);
                }
                if (FILE *f = fopen((base + ".off").c_str(), "wb")) {
                    fwrite(src_off.data(), 8, nd + 1, f);
                    fclose(f);
                }
            }

            const bool ok = hip_ok(agc_hip_zstd_batch(hip, (uint32_t)nd, zsrc_buf.data(), src_off.data(), any_ref ? levels.data() : nullptr, zdst_buf.data(), cap,
                                                      dst_off.data()),
                                   "zstd_batch");
            t_dev = now() - td;
            static const bool verify_dev = getenv("AGC_AMD_VERIFY_DEV_FRAMES") != nullptr;
            if (ok && !verify_dev) { // (a checking run compares the frames with libzstd's first: below, with the pool)
                helpers(nd, [&](size_t a, size_t b) {
                    for (size_t t = a; t < b; ++t) {
                        ZJob &j = jobs[dev_jobs[t]];
                        const uint32_t ps = (uint32_t)(dst_off[t + 1] - dst_off[t]);
                        bytes_t packed(zdst_buf.data() + dst_off[t], zdst_buf.data() + dst_off[t + 1]);
                        packed.push_back(0);
                        finish(j, packed, ps, j.kind == 0 ? j.marker : 0);
                        bytes_t().swap(j.staged);
                    }
                });
                dev_results_done = true;
            }
            return ok;
        });
    }
    const double th0 = now();
    st.t_zstd_stage += th0 - ts0;
    LAP("gather + launch");
    // a small batch (the new references of one registration, while the steps go on): a quarter of the pool keeps up with it
    // and leaves the cores and the caches to the thread that drives the steps
    uint64_t all_bytes = 0;
    for (const ZJob &j : jobs)
        all_bytes += j.data.size();
    // (... when that thread has human-size samples to drive; a collection of small genomes leaves it idle most of the time)
    const unsigned host_workers = all_bytes < (64u << 20) && heavy_steps && !z_caller_waits.load() ? std::max(2u, zpool->size() / 4) : zpool->size();
    zpool->parallel_for(host_jobs.size(), [&](size_t hi, unsigned tid) {
        host_compress(jobs[host_jobs[hi]], tid);
    }, host_workers);
    t_host = now() - th0;
    st.t_zstd_host += t_host;
    LAP("host pool");
    if (!dev_jobs.empty()) {
        const double t1 = now();
        const bool ok = ext ? true : dev_done.get();
        // (this runs on the entropy thread: its wait is NOT added to st.t_device, which the thread driving the steps reads for
        // its host-only stage times; t_zstd_dev below is the device call's own time)
        (void)t1;
        st.t_zstd_dev += t_dev;
        LAP("wait for the device");
        const double ts1 = now();
        if (!ok) { // the device refused: libzstd does them after all (same bytes)
            zpool->parallel_for(dev_jobs.size(), [&](size_t t, unsigned tid) {
                ZJob &j = jobs[dev_jobs[t]];
                const bytes_t &srcb = j.staged.empty() ? j.data : j.staged;
                size_t bound = zstd.compressBound(srcb.size());
                bytes_t packed(bound + 1);
                uint32_t ps = (uint32_t)zctx[tid]->compress(packed.data(), bound, srcb.data(), srcb.size(), j.kind == 0 ? j.level : 17);
                finish(j, packed, ps, j.kind == 0 ? j.marker : 0);
            });
        } else {
            // AGC_AMD_VERIFY_DEV_FRAMES=1 (a checking aid, e.g. `bench.py --verify-entropy`): every frame the device returned is
            // compressed again by libzstd and compared byte for byte -- the device entropy stage checked at full size, on the very
            // packs of this Close()
            static const bool verify_dev = getenv("AGC_AMD_VERIFY_DEV_FRAMES") != nullptr;
            if (verify_dev) {
                std::atomic<uint64_t> bad{0};
                zpool->parallel_for(dev_jobs.size(), [&](size_t t, unsigned tid) {
                    const ZJob &j = jobs[dev_jobs[t]];
                    const bytes_t &srcb = j.staged.empty() ? j.data : j.staged;
                    const size_t bound = zstd.compressBound(srcb.size());
                    bytes_t ref(bound + 1);
                    const size_t n = zctx[tid]->compress(ref.data(), bound, srcb.data(), srcb.size(), j.kind == 0 ? j.level : 17);
                    if (n != dst_off[t + 1] - dst_off[t] || memcmp(ref.data(), zdst_buf.data() + dst_off[t], n) != 0)
                        ++bad;
                });
                std::cerr << "verify: " << dev_jobs.size() << " device frames (" << src_off[dev_jobs.size()] / 1e6 << " MB) against libzstd (level 17; references 13 / 19): "
                          << bad.load() << " differ" << std::endl;
                verify_frames += dev_jobs.size();
                verify_bad += bad.load();
                if (bad.load())
                    err("entropy stage: device frames differ from libzstd");
            }
            if (!dev_results_done)
                zpool->parallel_for(dev_jobs.size(), [&](size_t t, unsigned) {
                    ZJob &j = jobs[dev_jobs[t]];
                    const uint32_t ps = (uint32_t)(dst_off[t + 1] - dst_off[t]);
                    bytes_t packed(zdst_buf.data() + dst_off[t], zdst_buf.data() + dst_off[t + 1]);
                    packed.push_back(0);
                    finish(j, packed, ps, j.kind == 0 ? j.marker : 0);
                    bytes_t().swap(j.staged);
                });
            if (!ext) {
                st.zstd_dev_in += src_off[dev_jobs.size()];
                st.zstd_dev_out += dst_off[dev_jobs.size()];
            }
            st.t_zstd_stage += now() - ts1;
            // rates of this call -> share of the next one (only meaningful when both had a real amount of work)
            if (!ext && src_off[dev_jobs.size()] > (8u << 20) && host_bytes > (8u << 20) && t_dev > 0 && t_host > 0 && !getenv("AGC_AMD_GPU_ZSTD_SHARE")) {
                const double r_dev = src_off[dev_jobs.size()] / t_dev, r_host = host_bytes / t_host;
                gpu_zstd_share = std::min(0.98, std::max(0.05, r_dev / (r_dev + r_host)));
            }
            if (verbosity > 0 && !ext)
                std::cerr << "entropy stage: device " << src_off[dev_jobs.size()] / 1e6 << " MB in " << t_dev << " s, host " << host_bytes / 1e6 << " MB in "
                          << t_host << " s; next device share " << gpu_zstd_share << std::endl;
        }
    }
    LAP("results -> jobs");
    st.t_zstd += now() - t0;
}

// hands finished parts to the archive buffer, in job order (= insertion order inside every stream)
void CAGCCompressor::Impl::add_job_parts(std::vector<ZJob> &jobs, size_t from, size_t to)
{
    for (size_t i = from; i < to; ++i) {
        ZJob &j = jobs[i];
        st.zstd_in += j.data.size();
        st.zstd_out += j.out.size();
        if (j.slot) { // the part took its place when the pack filled (a pack kept for the distributed Close)
            j.slot->out = std::move(j.out);
            j.slot->meta = j.meta;
            j.slot->ready.store(true, std::memory_order_release);
        } else
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
        // (the other thread may be adding the next sample's contigs: the lock covers the walk over the sample table only, not the
        // zstd 18 / 19 of what it wrote down -- 0.4 s at human scale)
        CollectionV3::SerializedBatch sb;
        {
            std::lock_guard<std::mutex> coll_lk(coll_mtx);
            coll.serialize_contig_batch(processed_samples - pack_cardinality, processed_samples, sb);
            stored_samples = processed_samples;
        }
        coll.store_serialized_batch(sb);
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
        if (packed_sample.n_symbols && k < 16 && !d_base) {
            // (the packed scan's suffix filter needs k >= 16: such a sample is expanded once for the byte scan)
            uint8_t *d = nullptr;
            if (!hip_ok(DEVT(agc_hip_sample_buffer(hip, packed_sample.n_symbols + 64, &d)), "sample_buffer") ||
                !hip_ok(DEVT(agc_hip_expand_dev(hip, &packed_sample, d)), "expand"))
                return AGC_HIP_ENODEV;
            d_base = d;
        }
        int rc = (packed_sample.n_symbols && k >= 16)
                     ? (scan_from_prefetch ? DEVT(agc_hip_scan_prefetched(hip, &packed_sample, ctg_off.data(), n_ctg, k, cap, &n_hits, h_ctg.data(),
                                                                          h_pos.data(), h_dir.data(), h_rc.data()))
                                           : DEVT(agc_hip_scan_packed_dev(hip, &packed_sample, ctg_off.data(), n_ctg, k, cap, &n_hits, h_ctg.data(),
                                                                          h_pos.data(), h_dir.data(), h_rc.data())))
                     : DEVT(agc_hip_scan_contigs_dev(hip, d_base, ctg_off.data(), n_ctg, k, cap, &n_hits, h_ctg.data(), h_pos.data(), h_dir.data(),
                                                     h_rc.data()));
        if (rc == AGC_HIP_ECAP) {
            cap = n_hits + n_hits / 8 + 64; // (headroom: a retry scans again)
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
    b.base_owned = next_base_owned;
    b.pk = packed_sample; // (n_symbols != 0: the LZ entry points read the sample's 2-bit words where they lie)
    {
        uint64_t bases = 0;
        for (const Contig &ct : ctgs)
            bases += ct.len;
        heavy_steps = bases >= 100000000ull; // (what the entropy thread sizes its share of the host pool by)
    }
    b.t0 = now();
    b.dev0 = st.t_device;
    b.lap_t = b.t0;
    if (!stage_scan(b))
        return false;
    if (b.needs_turn)
        return true;
    b.n_samples = ctgs.empty() ? 1u : ctgs.back().sample_idx + 1;
    b.overlap_encode = overlap_mode != 0 && b.n_samples == 1 && !always_speculate;
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
        // append mode: the caller classifies the rest again (unpacked groups change more than the dependencies revalidate()
        // follows; its windows hold one registration anyway).  Adaptive mode: a window never holds a registration that extends
        // the splitter set behind another one (stage_scan_dev cuts it there), so what revalidate() follows is all that changes
        if (b.commit_upto >= b.n_samples || appending || (adaptive && !adaptive_windows()))
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
    finish_spec_fill(b);
    const std::vector<Placed> &placed = placed_buf;
    double &t0 = b.t0, &dev0 = b.dev0;
    if (b.spec.size() != 2 * seg_buf.size()) {
        b.spec.assign(2 * seg_buf.size(), BatchState::Spec());
        b.spec_bytes = 0;
    }
    std::vector<uint32_t> items;
    for (uint32_t i = 0; i < placed.size(); ++i)
        if (placed[i].gid >= (int32_t)NO_RAW_GROUPS && groups[placed[i].gid].exists && !groups[placed[i].gid].packed &&
            !b.spec[placed[i].key].valid)
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
        st.enc_text += pl.len;
        st.enc_ref += groups[pl.gid].ref_size ? groups[pl.gid].ref_size - 1 : 0;
    }
    PinnedBytes &enc = enc_buf;
    const uint64_t base = b.spec_bytes; // the deltas are appended to what the window already holds
    uint64_t cap = std::max<uint64_t>(enc.size() > base ? enc.size() - base : 0, tot / 64 + (1u << 20));
    for (;;) {
        if (enc.size() < base + cap)
            enc.resize(base + cap);
        int r = b.pk.n_symbols ? DEVT(agc_hip_lz_encode_batch_packed(hip, (uint32_t)ne, gid.data(), &b.pk, off.data(), len.data(), rc.data(), enc.data() + base,
                                                                      cap, eoff.data()))
                               : DEVT(agc_hip_lz_encode_batch_dev(hip, (uint32_t)ne, gid.data(), b.d_base, off.data(), len.data(), rc.data(), enc.data() + base, cap,
                                                                  eoff.data()));
        if (r == AGC_HIP_ECAP) {
            cap = eoff[ne] + eoff[ne] / 8 + 4096; // (headroom: the next sample's deltas are a little longer, and a retry runs the kernel again)
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
        sp.enc_off = base + eoff[i];
        sp.enc_len = (uint32_t)(eoff[i + 1] - eoff[i]);
    }
    b.spec_bytes = base + eoff[ne];
    st.lz_encoded += ne;
    st.delta_bytes += eoff[ne];
    stage_end(st.t_encode, st.h_encode, t0, dev0);
    return true;
}

// The same for a window of ONE registration, in two halves around the rest of its classification: segments with both
// splitters whose key is in the map (97 % of a sample that resembles the collection) are placed where the map says, whatever
// the other segments turn out to be (a key never leaves the map, its group never changes: revalidate's first rule), so their
// deltas can be produced while estimates, missing-middle searches and split points are still being worked out.  The deltas are
// matched to the placed items by (group, offset, length, orientation) at commit time like every speculative delta.
bool CAGCCompressor::Impl::overlap_encode_begin(BatchState &b)
{
    finish_spec_fill(b);
    const std::vector<Contig> &ctgs = *b.ctgs;
    const std::vector<Seg> &segs = seg_buf;
    if (b.spec.size() != 2 * segs.size()) {
        b.spec.assign(2 * segs.size(), BatchState::Spec());
        b.spec_bytes = 0;
    }
    b.flight_keys.clear();
    b.flight_gid.clear();
    b.flight_len.clear();
    b.flight_off.clear();
    b.flight_rc.clear();
    for (uint32_t si : b.subset) {
        const Seg &s = segs[si];
        if (!s.front.full || !s.back.full || s.map_gid < (int32_t)NO_RAW_GROUPS)
            continue;
        const int32_t mg = s.map_gid, *m = &mg;
        const Group &g = groups[*m];
        if (!g.exists || g.packed)
            continue;
        b.flight_keys.push_back(2 * si);
        b.flight_gid.push_back((uint32_t)*m);
        b.flight_off.push_back(ctgs[s.ctg].off + s.start);
        b.flight_len.push_back(s.len);
        b.flight_rc.push_back((uint8_t)s.store_rc);
        st.enc_text += s.len;
        st.enc_ref += g.ref_size ? g.ref_size - 1 : 0;
    }
    if (b.flight_keys.empty())
        return true;
    if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_lz_encode_begin_packed(hip, (uint32_t)b.flight_keys.size(), b.flight_gid.data(), &b.pk, b.flight_off.data(),
                                                                     b.flight_len.data(), b.flight_rc.data()))
                               : DEVT(agc_hip_lz_encode_begin_dev(hip, (uint32_t)b.flight_keys.size(), b.flight_gid.data(), b.d_base, b.flight_off.data(),
                                                                  b.flight_len.data(), b.flight_rc.data())),
                "lz_encode_begin"))
        return false;
    b.enc_in_flight = true;
    return true;
}

bool CAGCCompressor::Impl::overlap_encode_end(BatchState &b)
{
    finish_spec_fill(b);
    const size_t ne = b.flight_keys.size();
    std::vector<uint64_t> eoff(ne + 1, 0);
    uint64_t tot = 0;
    for (uint32_t l : b.flight_len)
        tot += l;
    PinnedBytes &enc = enc_buf;
    const uint64_t base = b.spec_bytes;
    uint64_t cap = std::max<uint64_t>(enc.size() > base ? enc.size() - base : 0, tot / 64 + (1u << 20));
    for (;;) {
        if (!enc.resize(base + cap)) {
            err("out of memory (delta buffer)");
            return false;
        }
        int r = DEVT(agc_hip_lz_encode_end(hip, enc.data() + base, cap, eoff.data()));
        if (r == AGC_HIP_ECAP) {
            cap = eoff[ne] + 64 + eoff[ne] / 8; // (headroom: the next samples' deltas are about as long)
            continue;
        }
        b.enc_in_flight = false;
        if (!hip_ok(r, "lz_encode_end"))
            return false;
        break;
    }
    for (size_t i = 0; i < ne; ++i) {
        BatchState::Spec &sp = b.spec[b.flight_keys[i]];
        sp.valid = true;
        sp.gid = b.flight_gid[i];
        sp.off = b.flight_off[i];
        sp.len = b.flight_len[i];
        sp.rc = b.flight_rc[i] != 0;
        sp.enc_off = base + eoff[i];
        sp.enc_len = (uint32_t)(eoff[i + 1] - eoff[i]);
    }
    b.spec_bytes = base + eoff[ne];
    st.lz_encoded += ne;
    st.delta_bytes += eoff[ne];
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
// the device delivers segments, not hits: scan (or its prefetched result), reset rule, cut, keys and their look-up in the group
// table all run there (agc_hip_segments_packed), and the encode of every segment whose group is known is launched from there
// Every mode takes it (round 5): windows of several registrations and -c (the keys the device looked up are those of the window's
// first classification; a revalidation reads the host's map), append (a group that is still packed has no reference in HBM: the
// device does not launch its encode, the host does not expect it), adaptive mode (a contig without any splitter is looked at on
// the host; only when the splitter set really has to grow does the window go the host's way, stage_scan_host) and the N-rank
// mode (the prepare ahead of the turn).
bool CAGCCompressor::Impl::use_dev_segments(const BatchState &b) const
{
    return dev_segments && b.pk.n_symbols && k >= 16 && b.n_ctg && overlap_mode == 0;
}

// -> 1 done, 0 failed, 2 the window goes the host's way (adaptive mode: new splitters were mined; they wait in b.mined_*)
int CAGCCompressor::Impl::stage_scan_dev(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const uint32_t n_ctg = b.n_ctg;
    double &t0 = b.t0, &dev0 = b.dev0;
    std::vector<uint64_t> ctg_off(n_ctg + 1, 0);
    for (uint32_t i = 0; i < n_ctg; ++i) {
        ctg_off[i] = ctgs[i].off;
        st.bases += ctgs[i].len;
        if (i + 1 < n_ctg && ctgs[i].off + ctgs[i].len != ctgs[i + 1].off) {
            err("internal: contigs of a batch must be contiguous in HBM");
            return 0;
        }
    }
    ctg_off[n_ctg] = ctgs.back().off + ctgs.back().len;
    // the groups minted since the last sample go to the device's table first
    if (!hip_ok(DEVT(map_segments.sync_device(hip)), "group_map"))
        return 0;
    lap(b, "group map -> device");
    if (!dev_seg_buf.ctx)
        dev_seg_buf.ctx = hip;
    uint64_t cap = std::max<uint64_t>(dev_seg_buf.size() / sizeof(agc_hip_segment), std::max<uint64_t>(4096, (ctg_off[n_ctg] - ctg_off[0]) / 1000 + n_ctg));
    uint64_t n_segs = 0;
    // The whole-sample encode is launched INSIDE the call, right behind the group look-up, when everything that decides it is known
    // beforehand (the rule of `enc` below with the segment count guessed from the sample's length): the device has the encode to
    // work on while the segment table comes over and is converted -- 0.9 ms of an idle GPU at the front of every human-size step
    // (profiles/r6/step_gantt.txt).  A guess that turns out wrong, or any way out of this function before the launch is adopted,
    // drops the launch (PreLaunch's destructor).  Not in adaptive mode (the window may go the host's way) nor in the N-rank prepare.
    struct PreLaunch {
        Impl *I;
        bool armed = false;
        ~PreLaunch()
        {
            if (armed) {
                (void)agc_hip_lz_encode_drop_on(I->hip, 0);
                I->lane2_release();
            }
        }
    } pre{this};
    const bool pre_enc = pre_launch_encode && async_encode && book_can_async(1) && b.base_owned && ctgs.back().sample_idx == 0 && !adaptive && dist_world == 1 &&
                         (ctg_off[n_ctg] - ctg_off[0]) / std::max<uint64_t>(segment_size, 1) >= (uint64_t)dev_encode_min;
    if (pre_enc) {
        lane2_acquire(); // (the previous sample's deltas were collected by its early task long ago)
        pre.armed = true;
        lap(b, "second lane free");
    }
    uint32_t n_pre_encoded = 0;
    for (;;) {
        if (!dev_seg_buf.resize(cap * sizeof(agc_hip_segment), false)) {
            err("out of memory (segment table)");
            return 0;
        }
        const int rc = DEVT(agc_hip_segments_packed(hip, &b.pk, ctg_off.data(), n_ctg, k, scan_from_prefetch ? 1 : 0, pre_enc ? 1 : 0,
                                                    dev_seg_buf.size() / sizeof(agc_hip_segment), (agc_hip_segment *)dev_seg_buf.data(), &n_segs, &n_pre_encoded));
        if (rc == AGC_HIP_ECAP) { // (nothing was launched)
            cap = n_segs + n_segs / 8 + 64;
            continue;
        }
        if (!hip_ok(rc, "segments_packed"))
            return 0;
        break;
    }
    b.dev_keys = true;
    stage_end(st.t_scan, st.h_scan, t0, dev0);
    t0 = now();
    lap(b, "scan + segments (device)");
    // the device's records -> the host's (the pool: 50 k records of a human sample)
    const agc_hip_segment *dsegs = (const agc_hip_segment *)dev_seg_buf.data();
    std::vector<Seg> &segs = seg_buf;
    segs.resize(n_segs);
    // the encode of the segments whose group the table knew is launched on the device's second lane when its deltas can be
    // collected beside the next sample (bookkeeping thread); otherwise the commit encodes as before
    // (... and when the sample is large enough to fill the GPU with one wavefront per segment: the few dozen segments of a bacterial
    // genome are encoded from the host's descriptors at commit time instead, where the library parses them in chunks)
    const bool enc = async_encode && book_can_async(1) && b.base_owned && ctgs.back().sample_idx == 0 && n_segs >= dev_encode_min;
    std::vector<uint8_t> is_known(enc ? n_segs : 0, 0);
    {
        const size_t n_chunks = n_segs >= par_min ? std::min<size_t>(std::max<size_t>(n_segs / 2048, 2), (size_t)pool->size() * 4) : 1;
        auto conv = [&](size_t ci, unsigned) {
            for (size_t i = n_segs * ci / n_chunks; i < n_segs * (ci + 1) / n_chunks; ++i) {
                const agc_hip_segment &d = dsegs[i];
                Seg &s = segs[i];
                s.ctg = d.ctg;
                s.start = d.start;
                s.len = d.len;
                s.front.dir = d.front_dir;
                s.front.rc = d.front_rc;
                s.front.full = d.front_full != 0;
                s.back.dir = d.back_dir;
                s.back.rc = d.back_rc;
                s.back.full = d.back_full != 0;
                s.dev_gid = d.front_full && d.back_full ? d.map_gid : -2;
                // (the rule of known_flag_kernel: two splitters, a group of its own, its reference in HBM)
                if (enc && d.front_full && d.back_full && d.map_gid >= (int32_t)NO_RAW_GROUPS && (size_t)d.map_gid < groups.size() &&
                    groups[(uint32_t)d.map_gid].exists && !groups[(uint32_t)d.map_gid].packed)
                    is_known[i] = 1;
            }
        };
        if (n_chunks > 1)
            pool->parallel_for(n_chunks, conv);
        else
            conv(0, 0);
    }
    lap(b, "segments -> host records");
    // ---- adaptive mode: contigs without any splitter look for new ones (agc_compressor.cpp:2038-2044, 2054-2081).  When they
    // find none -- nearly always: a sample that resembles the collection has splitters in every contig -- the device's segments
    // stand; when the set has to grow the window goes the host's way (only those contigs take the second scan's hits,
    // :1187-1237), with what was mined here
    if (adaptive) {
        std::vector<uint32_t> need;
        for (size_t i = 0; i < n_segs; ++i)
            if ((i == 0 || dsegs[i - 1].ctg != dsegs[i].ctg) && !dsegs[i].back_full && ctgs[dsegs[i].ctg].len >= segment_size)
                need.push_back(dsegs[i].ctg);
        if (!need.empty()) {
            const std::vector<bytes_t> *host_data = b.host_data;
            std::vector<bytes_t> fetched_ctg(need.size());
            if (!host_data) {
                std::vector<uint64_t> off(need.size()), ooff(need.size() + 1);
                std::vector<uint32_t> len(need.size());
                uint64_t tot = 0;
                for (size_t i = 0; i < need.size(); ++i) {
                    off[i] = ctgs[need[i]].off;
                    len[i] = (uint32_t)ctgs[need[i]].len;
                    tot += len[i];
                }
                bytes_t buf(tot);
                if (!hip_ok(DEVT(agc_hip_fetch_slices_packed(hip, (uint32_t)need.size(), &b.pk, off.data(), len.data(), nullptr, buf.data(), tot, ooff.data())),
                            "fetch_slices"))
                    return 0;
                for (size_t i = 0; i < need.size(); ++i)
                    fetched_ctg[i].assign(buf.begin() + ooff[i], buf.begin() + ooff[i + 1]);
            }
            std::vector<std::vector<uint64_t>> found(need.size());
            pool->parallel_for(need.size(), [&](size_t i, unsigned) { find_new_splitters(host_data ? (*host_data)[need[i]] : fetched_ctg[i], found[i]); });
            // the first registration of the window that has to extend the set.  A window of several registrations (round 5: adaptive
            // mode speculates too) is valid as far as the set it was scanned with is: the registrations in front of that one
            // stand, the one itself and everything behind it come again -- it as the first of its window, where the set is its
            // to extend (agc_compressor.cpp:1187-1237: new splitters take effect for the contigs that come after)
            uint32_t first_bad = ~0u;
            for (size_t i = 0; i < need.size(); ++i)
                if (!found[i].empty())
                    first_bad = std::min(first_bad, ctgs[need[i]].sample_idx);
            if (first_bad != ~0u) {
                const uint32_t keep = first_bad == 0 ? 1u : first_bad; // registrations that stay in this window
                if (ctgs.back().sample_idx >= keep) {
                    uint32_t nc = 0;
                    while (nc < n_ctg && ctgs[nc].sample_idx < keep)
                        ++nc;
                    for (uint32_t i = nc; i < n_ctg; ++i)
                        st.bases -= ctgs[i].len;
                    size_t ns = 0;
                    while (ns < n_segs && dsegs[ns].ctg < nc)
                        ++ns;
                    b.ctgs->resize(nc);
                    b.n_ctg = nc;
                    segs.resize(ns);
                    ++st.windows_cut;
                    if (first_bad != 0) { // nothing new in what is left: the device's segments stand
                        lap(b, "window cut at the registration that brings new splitters");
                        b.dev_enc_n = 0;
                        return 1;
                    }
                }
                // the window's first registration extends the set (the window is that registration alone by now)
                for (uint32_t i = 0; i < b.n_ctg; ++i)
                    st.bases -= (*b.ctgs)[i].len; // (counted again by whoever scans the window next)
                b.dev_keys = false;
                if (b.no_new_splitters) { // prepared ahead of its turn: the set is not this sample's to extend yet
                    b.needs_turn = true;
                    return 1;
                }
                std::vector<uint32_t> need0;
                std::vector<std::vector<uint64_t>> found0;
                for (size_t i = 0; i < need.size(); ++i)
                    if (need[i] < b.n_ctg) {
                        need0.push_back(need[i]);
                        found0.emplace_back(std::move(found[i]));
                    }
                b.mined_valid = true;
                b.mined_need.swap(need0);
                b.mined_found.swap(found0);
                return 2;
            }
        }
        lap(b, "contigs without a splitter: nothing new");
    }
    uint32_t n_enc = 0;
    if (enc) {
        // The launch first: it needs nothing from the host but the count (the device made the descriptors), and the table of
        // speculative deltas below -- 100 k entries for a human sample, 0.6 ms -- used to be filled while the GPU had nothing to do
        // at the step's front (profiles/r6/step_gantt.txt).
        uint64_t known_text = 0;
        for (size_t i = 0; i < n_segs; ++i)
            if (is_known[i]) {
                ++n_enc;
                known_text += dsegs[i].len;
            }
        const bool fresh_spec = b.spec.size() != 2 * segs.size();
        if (fresh_spec)
            b.spec_bytes = 0;
        if (pre.armed && n_enc != n_pre_encoded) {
            err("internal: the device flags another number of segments for its encode than the host's rule");
            return 0;
        }
        if (n_enc) {
            b.dev_enc_n = n_enc;
            b.known_text = known_text;
            b.known_launch_due = true;
            const bool adopted = pre.armed;
            pre.armed = false; // (launch_known_encode owns the lane from here on)
            if (!launch_known_encode(b, adopted))
                return 0;
        }
        // The table of speculative deltas (every delta being made is matched to the placed item at commit time, like every
        // speculative delta) is first read by the placement stage: for a human-size sample -- 100 k entries, half a millisecond -- a
        // helper thread fills it while this one goes on to the keys and the estimates, which the step's critical path runs through
        // (profiles/r6/step_gantt.txt); finish_spec_fill() waits for it.  Nothing it reads changes before that: the segment table,
        // the contigs, the groups that exist (new ones are appended by stage_register, behind the wait).
        const size_t n_spec = 2 * segs.size();
        auto fill = [this, &b, fresh_spec, n_spec, n_segs, dsegs, &ctgs, known = std::move(is_known)]() -> std::pair<uint64_t, uint64_t> {
            if (fresh_spec)
                b.spec.assign(n_spec, BatchState::Spec());
            uint32_t k_enc = 0;
            uint64_t text = 0, ref = 0;
            for (size_t i = 0; i < n_segs; ++i) {
                if (!known[i])
                    continue;
                const agc_hip_segment &d = dsegs[i];
                BatchState::Spec &sp = b.spec[2 * i];
                sp.valid = true;
                sp.gid = (uint32_t)d.map_gid;
                sp.off = ctgs[d.ctg].off + d.start;
                sp.len = d.len;
                sp.rc = d.store_rc != 0;
                sp.enc_off = 0;
                sp.enc_len = 0;
                sp.pending = (int32_t)k_enc++;
                text += d.len;
                ref += groups[(uint32_t)d.map_gid].ref_size ? groups[(uint32_t)d.map_gid].ref_size - 1 : 0;
            }
            return {text, ref};
        };
        if (spec_fill_ahead > 0 && (n_segs >= 4096 || spec_fill_ahead > 1))
            b.spec_fill = std::async(std::launch::async, std::move(fill));
        else {
            const auto tr = fill();
            st.enc_text += tr.first;
            st.enc_ref += tr.second;
        }
        lap(b, "encode of the known segments launched");
    }
    b.dev_enc_n = n_enc;
    return 1;
}

// The launch of the whole-sample encode from the descriptors the device made (stage_scan_dev), queued at once: beside the estimates
// and the cost vectors.  (Behind the estimates, or behind the whole classification -- beside the announced scan and the FASTA
// conversion --, was measured in round 6: median step 13.7-15.5 ms against 12.6-13.2, profiles/EXPERIMENTS.md.)
bool CAGCCompressor::Impl::launch_known_encode(BatchState &b, bool launched_already)
{
    if (!b.known_launch_due)
        return true;
    b.known_launch_due = false;
    const uint32_t n_enc = b.dev_enc_n;
    {
        {
            if (!launched_already) { // (launched_already: inside agc_hip_segments_packed, the lane taken before that call)
                lane2_acquire(); // (the previous sample's deltas have been collected: they were while the table came over)
                lap(b, "second lane free");
                if (!hip_ok(DEVT(agc_hip_segments_encode_known(hip)), "segments_encode_known")) {
                    lane2_release();
                    b.dev_enc_n = 0;
                    return false;
                }
            }
            st.lz_encoded += n_enc;
            // The deltas are collected as soon as the kernel is done -- an early task of the bookkeeping thread, queued now -- when
            // it is certain already that the registration's books are that thread's too (stage_store_finish: hand_over).  The lane
            // is then free long before the next sample's launch asks for it (it used to be released by the registration's own
            // task, a whole step later: 0.9 ms of "second lane free" per human-size sample) and that task starts with its deltas
            // on the host.  enc_buf is nobody's until the hand-over swaps the buffer sets (bulk mode: spec_bytes == 0).
            if (early_collect && dist_world == 1 && book_can_async(1) && b.spec_bytes == 0) {
                const uint64_t text = b.known_text;
                std::unique_ptr<BookTask> t(new BookTask());
                t->early_only = true;
                t->enc_n = n_enc;
                t->enc_text = text;
                if (!enc_buf.ctx)
                    enc_buf.ctx = hip;
                t->enc_dst = &enc_buf;
                b.early_seq = book_submit(std::move(t));
            }
        }
    }
    lap(b, "encode of the known segments launched");
    return true;
}

bool CAGCCompressor::Impl::stage_scan(BatchState &b)
{
    if (use_dev_segments(b)) {
        const int r = stage_scan_dev(b);
        if (r != 2)
            return r == 1;
    }
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
            if (!need.empty() && !host_data && !(b.mined_valid && b.mined_need == need)) {
                std::vector<uint64_t> off(need.size()), ooff(need.size() + 1);
                std::vector<uint32_t> len(need.size());
                uint64_t tot = 0;
                for (size_t i = 0; i < need.size(); ++i) {
                    off[i] = ctgs[need[i]].off;
                    len[i] = (uint32_t)ctgs[need[i]].len;
                    tot += len[i];
                }
                bytes_t buf(tot);
                if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_fetch_slices_packed(hip, (uint32_t)need.size(), &b.pk, off.data(), len.data(), nullptr, buf.data(), tot, ooff.data()))
                                           : DEVT(agc_hip_fetch_slices_dev(hip, (uint32_t)need.size(), d_base, off.data(), len.data(), nullptr, buf.data(), tot, ooff.data())),
                            "fetch_slices"))
                    return false;
                for (size_t i = 0; i < need.size(); ++i)
                    fetched_ctg[i].assign(buf.begin() + ooff[i], buf.begin() + ooff[i + 1]);
            }
            std::vector<std::vector<uint64_t>> found(need.size());
            if (b.mined_valid && b.mined_need == need) // (the device path looked already: stage_scan_dev)
                found.swap(b.mined_found);
            else
                pool->parallel_for(need.size(), [&](size_t i, unsigned) {
                    find_new_splitters(host_data ? (*host_data)[need[i]] : fetched_ctg[i], found[i]);
                });
            b.mined_valid = false;
            size_t n_new = 0;
            for (auto &f : found)
                n_new += f.size();
            if (n_new && b.no_new_splitters) {
                // prepared ahead of its turn: the set is not this sample's to extend yet -- the sample is prepared again at its turn
                b.needs_turn = true;
                for (uint32_t i = 0; i < n_ctg; ++i)
                    st.bases -= ctgs[i].len;
                return true;
            }
            if (n_new) {
                ++spl_version;
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
    // The segment records are 208 bytes and a human sample has 50 k of them: they are laid out by a prefix sum over the contigs
    // and filled by the pool; only the fields the cut defines are written here -- classification (stage_classify) sets every
    // field it reads before it reads it, so records left over from the previous window need no clearing.
    std::vector<Seg> &segs = seg_buf;
    {
        std::vector<uint64_t> h_begin((size_t)n_ctg + 1, 0), s_begin((size_t)n_ctg + 1, 0);
        {
            uint64_t h = 0;
            for (uint32_t c = 0; c < n_ctg; ++c) {
                h_begin[c] = h;
                while (h < n_hits && h_ctg[h] == c)
                    ++h;
                const uint64_t nh = h - h_begin[c];
                const uint64_t last_split = nh ? h_pos[h - 1] + 1 - k : 0;
                s_begin[c + 1] = s_begin[c] + nh + (last_split < ctgs[c].len ? 1 : 0);
            }
            h_begin[n_ctg] = h;
        }
        segs.resize(s_begin[n_ctg]); // (grows or shrinks by the difference to the previous window only)
        LAP("cut_reserve");
        const size_t n_chunks = std::min<size_t>(n_ctg, std::max<size_t>(1, (size_t)pool->size() * 4));
        auto fill = [&](size_t ci, unsigned) {
            for (uint32_t c = (uint32_t)((uint64_t)n_ctg * ci / n_chunks); c < (uint32_t)((uint64_t)n_ctg * (ci + 1) / n_chunks); ++c) {
                uint64_t split_pos = 0;
                Kmer split_kmer;
                Seg *out = segs.data() + s_begin[c];
                for (uint64_t h = h_begin[c]; h < h_begin[c + 1]; ++h) {
                    Seg &s = *out++;
                    s.ctg = c;
                    s.start = split_pos;
                    s.len = (uint32_t)(h_pos[h] + 1 - split_pos);
                    s.front = split_kmer;
                    s.back.dir = h_dir[h];
                    s.back.rc = h_rc[h];
                    s.back.full = true;
                    split_pos = h_pos[h] + 1 - k;
                    split_kmer = s.back;
                }
                if (split_pos < ctgs[c].len) {
                    Seg &s = *out++;
                    s.ctg = c;
                    s.start = split_pos;
                    s.len = (uint32_t)(ctgs[c].len - split_pos);
                    s.front = split_kmer;
                    s.back = Kmer();
                }
            }
        };
        if (segs.size() >= par_min)
            pool->parallel_for(n_chunks, fill);
        else
            for (size_t ci = 0; ci < n_chunks; ++ci)
                fill(ci, 0);
    }
    LAP("cut_loop");

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
    // (a) per segment, independent of the others: reset of the classification fields, the key of a segment with both splitters
    // and its look-up in the map -- 50 k records and as many probes of a table that does not fit the caches: the pool
    {
        const size_t nL = L.size(), n_chunks = nL >= par_min ? std::min<size_t>(std::max<size_t>(nL / 2048, 2), (size_t)pool->size() * 4) : 1;
        auto reset_chunk = [&](size_t ci, unsigned) {
            for (size_t t = nL * ci / n_chunks; t < nL * (ci + 1) / n_chunks; ++t) {
                Seg &s = segs[L[t]];
                s.pk = {NO_KMER, NO_KMER};
                s.store_rc = false;
                s.cand_begin = s.cand_end = 0;
                s.back_only = false;
                s.mid_job = -1;
                s.known_gid = -2;
                s.use_rc = false;
                s.middle = NO_KMER;
                s.bp = 0;
                s.map_gid = -1;
                if (s.front.full && s.back.full) {
                    if (s.front.data() < s.back.data())
                        s.pk = {s.front.data(), s.back.data()};
                    else {
                        s.pk = {s.back.data(), s.front.data()};
                        s.store_rc = true;
                    }
                    if (b.dev_keys && s.dev_gid != -2)
                        s.map_gid = s.dev_gid; // (looked up on the device against the same table, with the cut)
                    else if (const int32_t *m = map_segments.find(s.pk))
                        s.map_gid = *m;
                } // (no splitter at all: pk stays {NO_KMER, NO_KMER}, agc_compressor.cpp:1286-1301, fallback filter off)
            }
        };
        if (n_chunks > 1)
            pool->parallel_for(n_chunks, reset_chunk);
        else
            reset_chunk(0, 0);
    }
    // (b) the one-splitter segments (contig ends: a few dozen per sample), in list order: their candidate lists
    for (uint32_t si : L) {
        Seg &s = segs[si];
        const bool ff = s.front.full, bf = s.back.full;
        if (ff == bf)
            continue;
        {
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
    if (b.overlap_encode && overlap_mode == 1 && !overlap_encode_begin(b))
        return false;
    LAP("encode_begin");
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
                st.est_text += s.len;
                st.est_ref += cands[c].ref_size - 1;
            }
        }
        std::vector<uint32_t> cost(which.size()), peak(which.size());
        if (!which.empty() &&
            !hip_ok(b.pk.n_symbols ? DEVT(agc_hip_lz_estimate_batch_packed(hip, (uint32_t)which.size(), gid.data(), &b.pk, off.data(), len.data(), rc.data(),
                                                                           cost.data(), peak.data()))
                                   : DEVT(agc_hip_lz_estimate_batch_dev(hip, (uint32_t)which.size(), gid.data(), d_base, off.data(), len.data(), rc.data(),
                                                                        cost.data(), peak.data())),
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
    {
        // the segments are independent here (the map and the terminator lists are only read): chunks of the list go to the pool,
        // the jobs they produce are numbered afterwards in segment order
        const size_t nL = L.size(), n_chunks = nL >= par_min ? std::min<size_t>(std::max<size_t>(nL / 1024, 2), (size_t)pool->size() * 4) : 1;
        std::vector<std::vector<MidJob>> chunk_jobs(n_chunks);
        std::vector<uint64_t> chunk_tried(n_chunks, 0);
        std::vector<uint8_t> chunk_bad(n_chunks, 0);
        auto one = [&](uint32_t si, std::vector<MidJob> &out, uint64_t &tried, bool &bad) {
        Seg &s = segs[si];
        if (concatenated || s.pk.first == NO_KMER || s.pk.second == NO_KMER)
            return;
        {
            // known group: remembered for the placement below
            int32_t mg = s.map_gid;
            if (s.front.full != s.back.full) { // (a one-splitter segment got its key just now)
                const int32_t *m = map_segments.find(s.pk);
                mg = m ? *m : -1;
            }
            if (mg >= 0) {
                s.known_gid = mg;
                return;
            }
        }
        s.known_gid = -1;
        auto tf = terminators.find(s.pk.first), tb = terminators.find(s.pk.second);
        if (tf == terminators.end() || tb == terminators.end())
            return;
        if (s.front.data() == s.back.data()) {
            if (!s.front.is_dir_oriented())
                s.store_rc = true;
            return;
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
        ++tried;
        if (shared.empty())
            return;
        s.middle = shared.front();
        const int32_t *m1 = map_segments.find(std::minmax(s.kmer1.data(), s.middle)), *m2 = map_segments.find(std::minmax(s.middle, s.kmer2.data()));
        if (!m1 || !m2) {
            bad = true;
            return;
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
                return;
            }
            if (e1) {
                s.mid_job = -2;
                return;
            }
        }
        // segment_dir here = use_rc ? rc(segment) : segment (:1394)
        const bool f_lt_m = s.kmer1.data() < s.middle, m_lt_b = s.middle < s.kmer2.data();
        j.rc1 = (uint8_t)(f_lt_m ? s.use_rc : !s.use_rc);
        j.pf1 = f_lt_m ? 1 : 0;
        j.rc2 = (uint8_t)(m_lt_b ? s.use_rc : !s.use_rc);
        j.pf2 = m_lt_b ? 0 : 1;
        out.push_back(j); // (its index in the job list is given out below, in segment order)
            };
        auto run_chunk = [&](size_t ci, unsigned) {
            bool bad = false;
            for (size_t t = nL * ci / n_chunks; t < nL * (ci + 1) / n_chunks && !bad; ++t)
                one(L[t], chunk_jobs[ci], chunk_tried[ci], bad);
            chunk_bad[ci] = bad;
        };
        if (n_chunks > 1)
            pool->parallel_for(n_chunks, run_chunk);
        else
            run_chunk(0, 0);
        for (size_t ci = 0; ci < n_chunks; ++ci) {
            if (chunk_bad[ci]) {
                err("internal: shared terminator without group");
                return false;
            }
            st.middle_tried += chunk_tried[ci];
            for (const MidJob &j : chunk_jobs[ci]) {
                segs[j.seg].mid_job = (int32_t)mids.size();
                mids.push_back(j);
            }
        }
    }
    stage_end(st.t_classify, st.h_classify, t0, dev0);
    t0 = now();
    LAP("mids");
    std::vector<uint32_t> best_pos(mids.size(), 0);
    // While this thread waits for the split points (4-5 ms at human scale) a helper places every segment that has no split-point
    // job -- what stage_place would do for it, minus the part number: the host phase between the classification kernels and the
    // index build is on the step's critical path (profiles/EXPERIMENTS.md), the wait is not.  Nothing the helper reads changes
    // meanwhile: the group map and the segments are this thread's, and this thread is inside the device call.
    std::future<void> place_helper;
    b.place_ahead_valid = false;
    if (place_ahead > 0 && !mids.empty() && (segs.size() >= 4096 || place_ahead > 1)) {
        b.place_ahead.resize(segs.size());
        b.place_ahead_ok.assign(segs.size(), 0);
        place_helper = std::async(std::launch::async, [this, &b, &segs, &ctgs] {
            for (uint32_t si = 0; si < segs.size(); ++si) {
                const Seg &s = segs[si];
                if (s.mid_job >= 0 || s.mid_job == -2)
                    continue;
                Placed &a = b.place_ahead[si];
                a.ctg = s.ctg;
                a.key = 2 * si;
                a.off = ctgs[s.ctg].off + s.start;
                a.len = s.len;
                a.rc = s.store_rc;
                a.pk = s.pk;
                a.part_no = 0;
                if (s.known_gid >= 0)
                    a.gid = s.known_gid;
                else {
                    const int32_t *m = map_segments.find(s.pk);
                    a.gid = m ? *m : -1;
                }
                b.place_ahead_ok[si] = 1;
            }
        });
    }
    struct JoinHelper { // (every way out of this function waits for the helper first)
        std::future<void> &f;
        ~JoinHelper()
        {
            if (f.valid())
                f.wait();
        }
    } join_helper{place_helper};
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
            st.cv_text += 2ull * s.len;
            st.cv_ref += groups[g1[i]].ref_size + groups[g2[i]].ref_size - 2;
        }
        if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_lz_split_point_batch_packed(hip, (uint32_t)n, g1.data(), g2.data(), &b.pk, off.data(), len.data(), r1.data(),
                                                                              p1.data(), r2.data(), p2.data(), best_pos.data(), nullptr))
                                   : DEVT(agc_hip_lz_split_point_batch_dev(hip, (uint32_t)n, g1.data(), g2.data(), d_base, off.data(), len.data(), r1.data(),
                                                                           p1.data(), r2.data(), p2.data(), best_pos.data(), nullptr)),
                    "lz_split_point_batch"))
            return false;
    }
    if (place_helper.valid()) {
        place_helper.get();
        b.place_ahead_valid = true;
    }
    for (size_t i = 0; i < mids.size(); ++i)
        segs[mids[i].seg].bp = best_pos[i];
    stage_end(st.t_gpu_aux, st.h_gpu_aux, t0, dev0);
    t0 = now();
    b.dev_keys = false; // (a later pass -- revalidation -- reads the map as it is then)

    return true;
}

// the helper that fills the table of speculative deltas (stage_scan_dev) is done: its symbol counts go to the statistics
void CAGCCompressor::Impl::finish_spec_fill(BatchState &b)
{
    if (!b.spec_fill.valid())
        return;
    const auto tr = b.spec_fill.get();
    st.enc_text += tr.first;
    st.enc_ref += tr.second;
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
    finish_spec_fill(b);
    // the announced next sample: its expansion + scan are queued NOW -- the classification kernels of this sample are done, what
    // follows is host work (placement, ordering, new group ids: ~2.5 ms at human scale) before the encode needs the GPU again
    launch_prefetch();
    if (b.overlap_encode && overlap_mode == 2 && !b.enc_in_flight && !overlap_encode_begin(b))
        return false;
    // ---- add_segment, part 4: final placement + part numbers ----
    std::vector<Placed> &placed = placed_buf;
    placed.clear();
    placed.reserve(segs.size() + segs.size() / 8 + 16);
    const bool ahead = b.place_ahead_valid && b.place_ahead.size() == segs.size();
    b.place_ahead_valid = false; // (a placement repeated after a revalidation reads the map as it is then)
    {
        uint32_t cur_ctg = ~0u, part_no = 0;
        for (uint32_t si = 0; si < segs.size(); ++si) {
            const Seg &s = segs[si];
            if (ahead && b.place_ahead_ok[si]) { // (placed while the split points were on their way: only the part number is missing)
                if (s.ctg != cur_ctg) {
                    cur_ctg = s.ctg;
                    part_no = 0;
                }
                placed.push_back(b.place_ahead[si]);
                placed.back().part_no = part_no++;
                continue;
            }
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
        if (sx >= s_from && sx + 1 < commit_upto && (pl.gid < 0 || groups[pl.gid].packed)) // (a window of one registration: never true)
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
        if (!std::is_sorted(keyed.begin(), keyed.end())) {
            // the items come contig by contig with rising part numbers: a stable counting sort by contig rank gives the
            // order at once (two contigs sharing a name, hence a rank: the comparison sort settles it)
            std::vector<uint32_t> cnt((size_t)n_ctg + 2, 0);
            for (auto &x : keyed)
                ++cnt[(x.first >> 32) + 1];
            for (size_t t = 0; t + 1 < cnt.size(); ++t)
                cnt[t + 1] += cnt[t];
            std::vector<std::pair<uint64_t, uint32_t>> sorted(keyed.size());
            for (auto &x : keyed)
                sorted[cnt[x.first >> 32]++] = x;
            keyed.swap(sorted);
            if (!std::is_sorted(keyed.begin(), keyed.end()))
                std::sort(keyed.begin(), keyed.end());
        }
        // items for NEW groups wait in a std::set keyed by (sample, contig name, part no) in the reference
        // (agc_compressor.h:111-118, 326-331): a second new item with the same key -- two contigs of one sample carrying the
        // same name -- never gets in.  Dropped here the same way (the first one, in contig order, stays).
        order.clear();
        uint64_t cur_key = ~0ULL;
        bool have_new = false;
        for (size_t i = 0; i < keyed.size(); ++i) {
            if (keyed[i].first != cur_key) {
                cur_key = keyed[i].first;
                have_new = false;
            }
            if (placed[keyed[i].second].gid < 0) {
                if (have_new)
                    continue;
                have_new = true;
            }
            order.push_back(keyed[i].second);
        }
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
bool CAGCCompressor::Impl::stage_store(BatchState &b) { return stage_store_head(b) && stage_store_finish(b); }

// store_segments (agc_compressor.cpp:974-1050), first half: everything the OTHER ranks of a multi-GPU job need to go on -- which
// item becomes a reference, the key -> group and terminator updates, the symbols of the new references and raw items -- and, in
// that mode, the head of the commit record.  Nothing here waits for an LZ encode.
bool CAGCCompressor::Impl::stage_store_head(BatchState &b)
{
    const uint8_t *d_base = b.d_base;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    (void)t0; (void)dev0;
    std::vector<Placed> &placed = placed_buf;
    std::vector<SampleLists> &per_sample = b.per_sample;
    LAP("per_sample");
    // (a) what each item needs: new groups' first item becomes the reference (segment.cpp:39-48), raw
    // groups keep the symbols, everything else is LZ-encoded -- decided per group across the committed samples
    std::vector<uint32_t> &new_ref_items = b.sto.new_ref_items; // placed indices, one per new group with items
    std::vector<uint32_t> &raw_items = b.sto.raw_items;
    std::vector<uint32_t> &enc_items = b.sto.enc_items;
    new_ref_items.clear();
    raw_items.clear();
    enc_items.clear();
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
        // (what later classifications read of the group; its packs are the bookkeeping stage's)
        groups[(uint32_t)placed[idx].gid].exists = true;
        groups[(uint32_t)placed[idx].gid].ref_size = (uint64_t)placed[idx].len + 1;
        if (placed[idx].pk.first != NO_KMER && placed[idx].pk.second != NO_KMER) { // terminator lists that gained an entry
            b.changed.push_back(placed[idx].pk.first);
            b.changed.push_back(placed[idx].pk.second);
        }
    }
    LAP("note_new_groups");
    // GPU: what the host must pack of the new references and raw items, and the repetitiveness probe of the references
    std::vector<uint32_t> lag_cnt, lag_cur;
    std::vector<uint8_t> &repetitive = b.sto.repetitive;
    repetitive.clear();
    bytes_t &fetched = fetch_buf;
    std::vector<uint64_t> &fetched_off = b.sto.fetched_off;
    fetched_off.clear();
    b.sto.ref_slot = -1;
    // The steps' thread does not need any of this itself when the registration's books are another thread's (one registration, one
    // GPU): counters and symbols are then asked for in ONE submission on a stream of their own and whoever does the books waits for
    // them (finish_ref_store) -- a one-block kernel queued behind the announced scan and the FASTA conversion took 1.4 ms to come
    // back, three round trips a step.  The reference sample's GBs of new references take the calls below as before.
    bool ref_async = false;
    if (ref_store_async && dist_world == 1 && book_can_async(b.n_samples) && b.pk.n_symbols && new_ref_items.size() + raw_items.size() > 0) {
        const size_t nr = new_ref_items.size(), nf = nr + raw_items.size();
        uint64_t tot = 0;
        for (size_t i = 0; i < nf; ++i)
            tot += placed[i < nr ? new_ref_items[i] : raw_items[i - nr]].len;
        if (tot <= (64ull << 20)) {
            std::vector<uint32_t> len(nf);
            std::vector<uint64_t> off(nf);
            std::vector<uint8_t> rc(nf);
            for (size_t i = 0; i < nf; ++i) {
                const Placed &pl = placed[i < nr ? new_ref_items[i] : raw_items[i - nr]];
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
                if (i < nr)
                    st.ref_bytes += pl.len;
            }
            const uint32_t slot = ref_pin_next;
            PinnedBytes &pin = ref_pin[slot];
            if (!pin.ctx)
                pin.ctx = hip;
            const size_t cnt_bytes = nr * 28 * 4;
            if (!pin.resize(2 * cnt_bytes + tot + 64, false)) {
                err("out of memory (reference staging)");
                return false;
            }
            fetched_off.resize(nf + 1);
            if (!hip_ok(DEVT(agc_hip_ref_store_begin_packed(hip, slot, (uint32_t)nr, (uint32_t)nf, &b.pk, off.data(), len.data(), rc.data(), (uint32_t *)pin.data(),
                                                            (uint32_t *)(pin.data() + cnt_bytes), pin.data() + 2 * cnt_bytes, tot, fetched_off.data())),
                        "ref_store_begin"))
                return false;
            ref_pin_next ^= 1u;
            b.sto.ref_slot = (int)slot;
            b.sto.ref_nr = nr;
            b.sto.ref_pin = &pin;
            ref_async = true;
            LAP("lag_counts + fetch_slices (queued)");
        }
    }
    if (!ref_async) {
        const size_t nr = new_ref_items.size();
        if (nr) {
            std::vector<uint32_t> len(nr);
            std::vector<uint64_t> off(nr);
            std::vector<uint8_t> rc(nr);
            for (size_t i = 0; i < nr; ++i) {
                const Placed &pl = placed[new_ref_items[i]];
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
                st.ref_bytes += pl.len;
            }
            lag_cnt.resize(nr * 28);
            lag_cur.resize(nr * 28);
            if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_ref_lag_counts_packed(hip, (uint32_t)nr, &b.pk, off.data(), len.data(), rc.data(), lag_cnt.data(), lag_cur.data()))
                                       : DEVT(agc_hip_ref_lag_counts_dev(hip, (uint32_t)nr, d_base, off.data(), len.data(), rc.data(), lag_cnt.data(), lag_cur.data())),
                        "ref_lag_counts"))
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
            LAP("lag_counts");
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
            if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_fetch_slices_packed(hip, (uint32_t)nf, &b.pk, off.data(), len.data(), rc.data(), fetched.data(), tot, fetched_off.data()))
                                       : DEVT(agc_hip_fetch_slices_dev(hip, (uint32_t)nf, d_base, off.data(), len.data(), rc.data(), fetched.data(), tot, fetched_off.data())),
                        "fetch_slices"))
                return false;
        }
    }
    LAP("fetch_slices");
    if (dist_world > 1) {
        if (!make_record_head(b))
            return false;
        LAP("record head");
    }
    return true;
}

// second half: the new references' index on this GPU, the deltas, the bookkeeping (and the body of the commit record)
bool CAGCCompressor::Impl::stage_store_finish(BatchState &b)
{
    const uint8_t *d_base = b.d_base;
    double &t0 = b.t0, &dev0 = b.dev0;
    auto LAP = [&](const char *what) { lap(b, what); };
    std::vector<Placed> &placed = placed_buf;
    const uint32_t commit_upto = b.commit_upto;
    std::vector<SampleLists> &per_sample = b.per_sample;
    std::vector<uint32_t> &new_ref_items = b.sto.new_ref_items, &raw_items = b.sto.raw_items, &enc_items = b.sto.enc_items;
    bytes_t &fetched = fetch_buf;
    // the previous registration's task is done with the second buffer set and with the device's second lane (long ago: it was
    // queued a whole step earlier)
    if (!book_wait_seq(last_own_seq))
        return false;
    // ... and the early collection of this sample's whole-sample encode (done since the kernel ended, a few ms ago)
    const bool early_done = b.early_seq != 0 && b.dev_enc_n != 0;
    if (b.early_seq != 0 && (!book_wait_seq(b.early_seq) || !early_enc.ok))
        return false;
    const bool hand_over = book_can_async(b.n_samples) && (dist_world == 1 || dist_rank == dist_writer);
    {
        const size_t nr = new_ref_items.size();
        if (nr) { // register the new references (index build)
            std::vector<uint32_t> gid(nr), len(nr);
            std::vector<uint64_t> off(nr);
            std::vector<uint8_t> rc(nr);
            for (size_t i = 0; i < nr; ++i) {
                const Placed &pl = placed[new_ref_items[i]];
                gid[i] = (uint32_t)pl.gid;
                off[i] = pl.off;
                len[i] = pl.len;
                rc[i] = pl.rc;
            }
            if (!hip_ok(b.pk.n_symbols ? DEVT(agc_hip_ref_register_batch_packed(hip, (uint32_t)nr, gid.data(), &b.pk, off.data(), len.data(), rc.data(), mml))
                                       : DEVT(agc_hip_ref_register_batch_dev(hip, (uint32_t)nr, gid.data(), d_base, off.data(), len.data(), rc.data(), mml)),
                        "ref_register_batch"))
                return false;
            LAP("ref_register");
        }
    }
    stage_end(st.t_register, st.h_register, t0, dev0);
    t0 = now();
    // LZ deltas (segment.cpp:50-58): items whose group already had its reference when the window was classified were encoded
    // up front in one batch (spec_encode); only the others -- followers of a group minted in this window, segments that a
    // revalidation placed differently -- are encoded now
    std::vector<const uint8_t *> enc_ptr(enc_items.size(), nullptr);
    std::vector<uint32_t> enc_len(enc_items.size(), 0);
    if (b.enc_in_flight && !overlap_encode_end(b))
        return false;
    LAP("encode_end");
    std::vector<uint32_t> enc_later; // per result of the encode in flight on the second lane: its position in enc_items (~0u: unused)
    uint64_t enc_later_text = 0;
    std::vector<uint32_t> enc_later2; // ... and of the one launched below on lane 1 while lane 0 still holds the whole-sample encode
    uint64_t enc_later2_text = 0;
    // the device launched the encode of the segments whose group it knew (stage_scan_dev).  Its deltas are first read by the
    // bookkeeping task, which then collects them beside the next sample -- unless there is no such task (several registrations in
    // the window do not reach here with one in flight; -c, append, AGC_AMD_ASYNC_BOOK=0) or this is the N-rank mode, where the
    // record's body is made below and holds every delta: then they are collected now, behind the window's other speculative
    // deltas, and are ordinary speculative deltas from here on
    if (b.dev_enc_n != 0 && (!hand_over || dist_world > 1)) {
        const size_t ne = b.dev_enc_n;
        std::vector<uint64_t> eoff(ne + 1, 0);
        const uint64_t base = b.spec_bytes;
        uint64_t cap = std::max<uint64_t>(enc_buf.size() > base ? enc_buf.size() - base : 0, (uint64_t)1 << 16);
        if (early_done) { // (collected already, at the start of enc_buf: an early task is only queued with spec_bytes == 0)
            if (base != 0 || early_enc.eoff.size() != ne + 1) {
                err("internal: early collection of the encode does not match the registration");
                return false;
            }
            eoff = early_enc.eoff;
        }
        for (; !early_done;) {
            if (!enc_buf.resize(base + cap)) {
                err("out of memory (delta buffer)");
                return false;
            }
            const int r = DEVT(agc_hip_lz_encode_end(hip, enc_buf.data() + base, cap, eoff.data()));
            if (r == AGC_HIP_ECAP) {
                cap = eoff[ne] + eoff[ne] / 8 + 4096;
                continue;
            }
            lane2_release();
            if (!hip_ok(r, "lz_encode_end"))
                return false;
            break;
        }
        for (BatchState::Spec &sp : b.spec)
            if (sp.valid && sp.pending >= 0) {
                sp.enc_off = base + eoff[(size_t)sp.pending];
                sp.enc_len = (uint32_t)(eoff[(size_t)sp.pending + 1] - eoff[(size_t)sp.pending]);
                sp.pending = -1;
            }
        b.spec_bytes = base + eoff[ne];
        if (!early_done) // (an early task's bytes are counted by the book thread)
            st.delta_bytes += eoff[ne];
        b.dev_enc_n = 0;
        LAP("encode of the known segments collected");
    }
    const bool bulk = b.dev_enc_n != 0;
    if (bulk)
        enc_later.assign(b.dev_enc_n, ~0u);
    {
        std::vector<uint32_t> todo; // positions in enc_items
        for (uint32_t i = 0; i < enc_items.size(); ++i) {
            const Placed &pl = placed[enc_items[i]];
            const BatchState::Spec *sp = pl.key < b.spec.size() ? &b.spec[pl.key] : nullptr;
            if (sp && sp->valid && sp->gid == (uint32_t)pl.gid && sp->off == pl.off && sp->len == pl.len && sp->rc == pl.rc) {
                if (sp->pending >= 0) {
                    enc_later[(uint32_t)sp->pending] = i;
                    enc_later_text += pl.len;
                } else {
                    enc_ptr[i] = enc_buf.data() + sp->enc_off;
                    enc_len[i] = sp->enc_len;
                }
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
                st.enc_text += pl.len;
                st.enc_ref += groups[pl.gid].ref_size ? groups[pl.gid].ref_size - 1 : 0;
            }
            // The encode is only LAUNCHED here when its result is first read by the bookkeeping task: the task collects it (second
            // device lane) while this thread goes on with the next sample.  Needs a sample in a staging buffer the device context
            // owns (it outlives the call) and nothing else on that lane.
            const bool later = hand_over && async_encode && dist_world == 1 && b.base_owned && b.pk.n_symbols && overlap_mode == 0 && !b.enc_in_flight && b.spec_bytes == 0;
            if (!bulk && later) {
                lane2_acquire();
                if (!hip_ok(DEVT(agc_hip_lz_encode_begin_packed(hip, (uint32_t)ne, gid.data(), &b.pk, off.data(), len.data(), rc.data())), "lz_encode_begin")) {
                    lane2_release();
                    return false;
                }
                enc_later.swap(todo);
                enc_later_text = tot;
                st.lz_encoded += ne;
            } else if (bulk && later) {
                // (lane 0 carries the encode of the segments the device knew; these -- followers of the groups this sample minted,
                // whose references were indexed a moment ago -- go to lane 1 and are collected by the same task)
                lane2_acquire(1);
                if (!hip_ok(DEVT(agc_hip_lz_encode_begin_packed_on(hip, 1, (uint32_t)ne, gid.data(), &b.pk, off.data(), len.data(), rc.data())), "lz_encode_begin")) {
                    lane2_release(1);
                    return false;
                }
                enc_later2.swap(todo);
                enc_later2_text = tot;
                st.lz_encoded += ne;
            } else {
            PinnedBytes &enc = enc_buf2;
            uint64_t cap = std::max<uint64_t>(enc.size(), tot / 64 + (1u << 16));
            for (;;) {
                if (enc.size() < cap)
                    enc.resize(cap);
                int r = b.pk.n_symbols ? DEVT(agc_hip_lz_encode_batch_packed(hip, (uint32_t)ne, gid.data(), &b.pk, off.data(), len.data(), rc.data(), enc.data(), cap,
                                                                              eoff.data()))
                                       : DEVT(agc_hip_lz_encode_batch_dev(hip, (uint32_t)ne, gid.data(), d_base, off.data(), len.data(), rc.data(), enc.data(), cap,
                                                                          eoff.data()));
                if (r == AGC_HIP_ECAP) {
                    cap = eoff[ne] + eoff[ne] / 8 + 4096; // (headroom: the next sample's deltas are a little longer, and a retry runs the kernel again)
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
    }
    LAP(enc_later.empty() && enc_later2.empty() ? "encode" : "encode (in flight)");
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
    cdta.repetitive = std::move(b.sto.repetitive);
    cdta.fetched = &fetched;
    cdta.fetched_off = std::move(b.sto.fetched_off);
    cdta.ref_slot = b.sto.ref_slot;
    cdta.ref_nr = b.sto.ref_nr;
    cdta.ref_pin = b.sto.ref_pin;
    b.sto.ref_slot = -1;
    cdta.enc_ptr = std::move(enc_ptr);
    cdta.enc_len = std::move(enc_len);
    if (dist_world > 1) {
        if (!make_record_body(cdta))
            return false;
        LAP("record body");
        if (dist_rank != dist_writer)
            return true; // the writer rank does the bookkeeping from the record
    }
    if (hand_over) {
        // beside the next sample: the registration's buffers change places with the second set, which is the task's until it is
        // done (the raw pointers of cdta stay valid: the blocks themselves do not move)
        std::unique_ptr<BookTask> t(new BookTask());
        t->ctgs = *b.ctgs;
        placed_buf.swap(placed_alt);
        fetch_buf.swap(fetch_alt);
        enc_buf.swap(enc_alt);
        enc_buf2.swap(enc_alt2);
        t->cd = std::move(cdta);
        t->cd.ctgs = &t->ctgs;
        t->cd.placed = &placed_alt;
        t->cd.fetched = &fetch_alt;
        if (!enc_later.empty()) {
            t->enc_pending = true;
            if (bulk && early_done) { // the deltas are in enc_buf (enc_alt since the swap above) already
                if (early_enc.eoff.size() != enc_later.size() + 1) {
                    err("internal: early collection of the encode does not match the registration");
                    return false;
                }
                t->enc_pending = false;
                t->enc_collected = true;
                t->enc_eoff.swap(early_enc.eoff);
            }
            t->enc_todo = std::move(enc_later);
            t->enc_text = enc_later_text;
            // (the deltas encoded at commit time sit in enc_alt2 now; the device-launched encode of the whole sample is collected
            // into the other buffer of the set, which holds nothing in this mode: spec_bytes == 0)
            t->enc_dst = bulk ? &enc_alt : &enc_alt2;
        }
        if (!enc_later2.empty()) {
            t->enc2_pending = true;
            t->enc2_todo = std::move(enc_later2);
            t->enc2_text = enc_later2_text;
            t->enc2_dst = &enc_alt2;
        }
        last_own_seq = book_submit(std::move(t));
        LAP("book_and_store (queued)");
        return true;
    }
    const bool ok = book_wait() && finish_ref_store(cdta, fetch_buf) && book_and_store(cdta);
    LAP("book_and_store");
    return ok;
}

// the second half of agc_hip_ref_store_begin_packed: waits for the slot, applies the reference's double-precision test to the lag
// counters (segment.h:224-247) and puts the symbols where book_and_store reads them
bool CAGCCompressor::Impl::finish_ref_store(CommitData &cd, bytes_t &fetched_dst)
{
    if (cd.ref_slot < 0)
        return true;
    if (!hip_ok(agc_hip_ref_store_end(hip, (uint32_t)cd.ref_slot), "ref_store_end"))
        return false;
    cd.ref_slot = -1;
    const size_t nr = cd.ref_nr, cnt_bytes = nr * 28 * 4;
    const uint32_t *lag_cnt = (const uint32_t *)cd.ref_pin->data(), *lag_cur = (const uint32_t *)(cd.ref_pin->data() + cnt_bytes);
    cd.repetitive.assign(nr, 0);
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
        cd.repetitive[fi] = !(best_frac < 0.5);
    }
    const uint64_t tot = cd.fetched_off.empty() ? 0 : cd.fetched_off.back();
    fetched_dst.assign(cd.ref_pin->data() + 2 * cnt_bytes, cd.ref_pin->data() + 2 * cnt_bytes + tot);
    cd.fetched = &fetched_dst;
    return true;
}

// store_segments, second half (agc_compressor.cpp:989-1050): per-group bookkeeping, zstd parts, collection records
bool CAGCCompressor::Impl::book_and_store(CommitData &cdta)
{
    double t0 = now(), dev0 = book_on_thread ? 0.0 : st.t_device;
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
    std::vector<uint8_t> is_new_ref(placed.size(), 0); // the item that becomes its group's reference (stage_store / the record decided)
    for (uint32_t idx : new_ref_items)
        is_new_ref[idx] = 1;
    ThreadPool *const wp = book_on_thread ? bpool.get() : pool.get();
    // contig descriptors of the collection (agc_compressor.cpp:1038-1049).  (The sample table may grow on the other thread --
    // the next sample's contigs --; the descriptors found here stay where they are.)
    std::vector<CollectionV3::ContigDesc *> cd(n_ctg, nullptr);
    bool dup_names_in_batch = false;
    {
        std::lock_guard<std::mutex> coll_lk(coll_mtx);
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
                    uint32_t igid;
                    if (gid < NO_RAW_GROUPS) {
                        if (g.raw_off.size() == pack_cardinality)
                            make_pack_job(jobs, gid, g.raw_data, g.raw_off);
                        const uint32_t fi = (uint32_t)new_ref_items.size() + pos_raw[idx];
                        ++g.no_seqs;
                        Group::push(g.raw_data, g.raw_off, fetched.data() + fetched_off[fi], fetched_off[fi + 1] - fetched_off[fi]);
                        igid = g.no_seqs - 1;
                    } else if (is_new_ref[idx]) {
                        const uint32_t fi = pos_newref[idx];
                        ZJob j;
                        j.stream_id = g.stream_ref;
                        j.kind = 0;
                        j.data.assign(fetched.begin() + fetched_off[fi], fetched.begin() + fetched_off[fi + 1]);
                        j.repetitive = cdta.repetitive[fi] != 0;
                        jobs.emplace_back(std::move(j));
                        g.no_seqs = 1;
                        igid = 0;
                    } else {
                        if (g.lzp_off.size() == pack_cardinality)
                            make_pack_job(jobs, gid, g.lzp_data, g.lzp_off);
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
                                Group::push(g.lzp_data, g.lzp_off, dp, dn, pack_cardinality);
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
        if (sl.n_lists() >= par_min) {
            const size_t n_chunks = std::min<size_t>(sl.n_lists(), (size_t)wp->size() * 8);
            std::vector<std::vector<ZJob>> chunk_jobs(n_chunks);
            defer_stream_reg = true;
            wp->parallel_for(n_chunks, [&](size_t ci, unsigned) {
                book(sl.n_lists() * ci / n_chunks, sl.n_lists() * (ci + 1) / n_chunks, chunk_jobs[ci]);
            });
            // a group taken over from an input archive without a delta stream (append mode) registers it with its first pack
            // (segment.h:262-266): here, in list order, as the serial path does -- never in thread-timing order
            defer_stream_reg = false;
            for (auto &cj : chunk_jobs)
                for (auto &j : cj) {
                    if (j.stream_id < 0) {
                        Group &g = groups[j.gid];
                        if (g.stream_delta < 0)
                            g.stream_delta = ar.register_stream(ss_delta_name(j.gid));
                        j.stream_id = g.stream_delta;
                    }
                    jobs.emplace_back(std::move(j));
                }
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
    if (!book_on_thread) // (a queued task's time is added by book_wait)
        stage_end(st.t_store, st.h_store, t0, dev0);
    if (verbosity > 1 && !book_on_thread)
        std::cerr << "registration: " << placed.size() << " items; host-only seconds so far: scan " << st.h_scan << " classify " << st.h_classify
                  << " register " << st.h_register << " encode " << st.h_encode << " store " << st.h_store << std::endl;
    // the parts take their places in the archive now (per registration, then its end-of-registration steps); their payload
    // follows from the entropy thread
    for (ZJob &j : all_jobs)
        j.slot = std::make_shared<PartSlot>();
    for (uint32_t sidx = 0; sidx < n_regs; ++sidx) {
        for (size_t i = sidx ? jobs_end[sidx - 1] : 0; i < jobs_end[sidx]; ++i)
            ar.add_part_deferred(all_jobs[i].stream_id, all_jobs[i].slot);
        after_registration();
    }
    if (dist_world > 1 && gpu_zstd) {
        // one archive from N ranks: the writer's own entropy stage would be the only one at work during the run -- full packs a
        // device can code are kept for the distributed Close (CloseCollectPacks), the rest (references) goes on now
        const uint32_t dev_max = agc_hip_zstd17_max_input();
        std::vector<ZJob> now_jobs;
        std::lock_guard<std::mutex> dlk(deferred_mtx); // (DealCollectPacks takes them from the thread that drives the steps)
        for (ZJob &j : all_jobs)
            if (j.kind == 1 && !j.data.empty() && j.data.size() <= dev_max) {
                deferred_bytes += j.data.size();
                deferred_packs.emplace_back(std::move(j));
            } else
                now_jobs.emplace_back(std::move(j));
        // The kept packs are dealt to the ranks in the middle of the run (agc_amd/dist.py: DealCollectPacks as soon as a few dozen MB
        // have piled up) or at Close.  A safety net for a caller that never deals: behind the first kept pack every later part of
        // the archive waits in host memory (ArchiveWriter writes its events in order) -- past the ceiling (a quarter of the
        // machine's memory, 16 GiB at most; round 5: 2 GiB, which the first fill of 50 k human packs at -b 100 overran) the
        // writer's own entropy stage takes what has piled up and the archive flows again.
        static const uint64_t defer_cap = []() -> uint64_t {
            if (const char *e = getenv("AGC_AMD_DEFER_MAX_MB"))
                return (uint64_t)strtoull(e, nullptr, 10) << 20;
            const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
            const uint64_t quarter = pages > 0 && psz > 0 ? (uint64_t)pages * (uint64_t)psz / 4 : (4ull << 30);
            return std::min<uint64_t>(16ull << 30, std::max<uint64_t>(2ull << 30, quarter));
        }();
        if (deferred_bytes > defer_cap) {
            for (ZJob &j : deferred_packs)
                now_jobs.emplace_back(std::move(j));
            deferred_packs.clear();
            deferred_bytes = 0;
        }
        all_jobs.swap(now_jobs);
    }
    z_submit(std::move(all_jobs));
    if (sync_entropy)
        z_wait_all();
    return true;
}

// store_segments' update of map_segments (keep the smaller id) and of the terminator lists, agc_compressor.cpp:1003-1028
void CAGCCompressor::Impl::note_new_group(const pk_t &pk, uint32_t gid)
{
    int32_t *it = map_segments.find(pk);
    if (!it)
        map_segments[pk] = (int32_t)gid;
    else if (*it > (int32_t)gid) {
        *it = (int32_t)gid;
        map_segments.touch(it);
    }
    if (prepared)
        minted_since_prepare = true;
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

// CSegment::finish for every group (agc_compressor.cpp:880-904, segment.cpp:125-133)
void CAGCCompressor::Impl::finish_groups()
{
    if (close_collected) { // CloseCollectPacks built the jobs; their device-eligible packs were compressed elsewhere
        run_jobs(close_jobs);
        close_jobs.clear();
        close_collected = false;
        return;
    }
    static const bool laps = getenv("AGC_AMD_LAPS") != nullptr;
    const double tl0 = now();
    std::vector<ZJob> jobs;
    build_close_jobs(jobs);
    const double tl1 = now();
    {
        // (50 k parts take their places in one go: one lock, not one per part)
        std::vector<std::pair<int, std::shared_ptr<PartSlot>>> places;
        places.reserve(jobs.size());
        for (ZJob &j : jobs) {
            j.slot = std::make_shared<PartSlot>();
            places.emplace_back(j.stream_id, j.slot);
        }
        ar.add_parts_deferred(places);
    }
    deferred_bytes = 0;
    for (ZJob &j : deferred_packs) // (Close without CloseCollectPacks: the kept packs are coded here after all)
        jobs.emplace_back(std::move(j));
    deferred_packs.clear();
    z_caller_waits = true;     // (Close waits for the entropy thread, then flushes)
    if (laps)
        std::cerr << "    finish_groups: pack jobs " << (tl1 - tl0) * 1e3 << " ms, their places in the archive " << (now() - tl1) * 1e3 << " ms\n";
    z_submit(std::move(jobs));
}

// the pack jobs of every group's open pack (+ the parts an appended archive's untouched groups keep as they are)
void CAGCCompressor::Impl::build_close_jobs(std::vector<ZJob> &jobs)
{
    jobs.reserve(jobs.size() + groups.size() + 16);
    for (uint32_t gid = 0; gid < groups.size(); ++gid) {
        Group &g = groups[gid];
        if (!g.lzp_off.empty())
            make_pack_job(jobs, gid, g.lzp_data, g.lzp_off);
        if (!g.raw_off.empty())
            make_pack_job(jobs, gid, g.raw_data, g.raw_off);
        if (g.packed && g.pk_delta) { // store_compressed_delta_in_archive, segment.h:283-292
            if (g.stream_delta < 0)
                g.stream_delta = ar.register_stream(ss_delta_name(gid));
            ar.add_part_buffered(g.stream_delta, bytes_t(g.pk_delta, g.pk_delta + g.pk_delta_size), g.pk_delta_meta);
        }
    }
}

} // namespace agc
