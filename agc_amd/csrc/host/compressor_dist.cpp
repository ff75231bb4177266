// compressor_dist.cpp -- commit records of the multi-GPU single-archive mode (see compressor.h, agc_amd/dist.py).
#include "compressor_impl.h"

namespace agc {

// ---------------------------------------------------------------------------
// Multi-GPU single-archive mode.  Samples are dealt round-robin to the ranks; every rank keeps the
// whole classification state (splitters, (k1,k2) -> group map, terminators, references in its HBM).  The owner of a
// sample classifies and encodes it (process_batch), then publishes a COMMIT RECORD: contig names, new splitters and,
// group by group in registration order, every placed item with its payload (symbols of a new reference -- the
// "newly-minted reference segments" every GPU needs --, raw symbols, or the delta).  All other ranks apply the
// record (apply_record): same group ids, same map/terminator updates, references registered in their own HBM;
// the writer rank also runs the bookkeeping / zstd / archive stage from it.  Samples are committed strictly in
// order, so the archive equals the single-GPU one byte for byte.
// Record layout (little endian): HEAD = "AGCR" | n_ctg | n_lists | n_new_splitters | first_new_gid | n_new_groups |
//   contigs: sample\0 name\0 ... | splitters u64... | lists: gid, n_items, items: ctg, part_no, len, rc, kind,
//   [pk1, pk2, repetitive for kind 0], [payload_len, payload for kinds 0 (new reference) and 1 (raw)]
// BODY = for every kind-2 item (LZ delta), in item order: payload_len, payload.  Every rank needs the head (ids, keys, the
// new references: ~3 MB per human-size sample); only the WRITER needs the body (~22 MB): agc_amd/dist.py broadcasts the one and
// sends the other point to point.  The head is complete before any delta is encoded (CommitPreparedHead): the other ranks
// go on while the owner indexes its new references, encodes what is left and builds the body (CommitPreparedFinish).
// ---------------------------------------------------------------------------
namespace {
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

// The head: built in the first half of store_segments (stage_store_head) -- before the new references are indexed on this GPU and
// before any leftover delta is encoded -- so that the other ranks can go on as early as possible.
bool CAGCCompressor::Impl::make_record_head(BatchState &b)
{
    const std::vector<Contig> &ctgs = *b.ctgs;
    const std::vector<Placed> &placed = placed_buf;
    const BatchState::Store &sto = b.sto;
    dist_body_n = 0;
    dist_record_n = 0;
    dist_body_items.clear();
    const SampleLists &sl = b.per_sample.at(0); // one registration per record
    std::vector<uint32_t> pos_newref(placed.size()), pos_raw(placed.size());
    for (uint32_t i = 0; i < sto.new_ref_items.size(); ++i)
        pos_newref[sto.new_ref_items[i]] = i;
    for (uint32_t i = 0; i < sto.raw_items.size(); ++i)
        pos_raw[sto.raw_items[i]] = i;
    std::vector<uint8_t> kind(placed.size(), 2);
    for (uint32_t idx : sto.new_ref_items)
        kind[idx] = 0;
    for (uint32_t idx : sto.raw_items)
        kind[idx] = 1;
    // size first (50 k items of 14 bytes and a few MB of symbols: written through a pointer, not byte by byte)
    size_t need = 24 + 8 * b.new_splitters_added.size() + 8 * sl.n_lists() + 14 * sl.items.size();
    for (auto &c : ctgs)
        need += c.sample.size() + c.name.size() + 2;
    need += 21 * sto.new_ref_items.size() + 4 * sto.raw_items.size() + (sto.fetched_off.empty() ? 0 : sto.fetched_off.back());
    if (!dist_record_buf.resize(DIST_FRAME + need + need / 8 + 64, false)) {
        err("out of memory (commit record)");
        return false;
    }
    uint8_t *w = dist_record_ptr();
    auto w32 = [&](uint32_t x) {
        w[0] = (uint8_t)x, w[1] = (uint8_t)(x >> 8), w[2] = (uint8_t)(x >> 16), w[3] = (uint8_t)(x >> 24);
        w += 4;
    };
    auto w64 = [&](uint64_t x) {
        w32((uint32_t)x);
        w32((uint32_t)(x >> 32));
    };
    memcpy(w, "AGCR", 4);
    w += 4;
    w32((uint32_t)ctgs.size());
    w32((uint32_t)sl.n_lists());
    w32((uint32_t)b.new_splitters_added.size());
    uint32_t first_new = ~0u, n_new = 0;
    for (uint32_t idx : sto.new_ref_items) {
        first_new = std::min(first_new, (uint32_t)placed[idx].gid);
        ++n_new;
    }
    w32(first_new);
    w32(n_new);
    for (auto &c : ctgs) {
        memcpy(w, c.sample.data(), c.sample.size());
        w += c.sample.size();
        *w++ = 0;
        memcpy(w, c.name.data(), c.name.size());
        w += c.name.size();
        *w++ = 0;
    }
    for (uint64_t x : b.new_splitters_added)
        w64(x);
    const bytes_t &fetched = fetch_buf;
    for (size_t li = 0; li < sl.n_lists(); ++li) {
        w32(sl.gids[li]);
        w32(sl.begin[li + 1] - sl.begin[li]);
        for (uint32_t ii = sl.begin[li]; ii < sl.begin[li + 1]; ++ii) {
            const uint32_t idx = sl.items[ii];
            const Placed &pl = placed[idx];
            w32(pl.ctg);
            w32(pl.part_no);
            w32(pl.len);
            *w++ = (uint8_t)pl.rc;
            *w++ = kind[idx];
            if (kind[idx] == 2) {
                dist_body_items.push_back(idx); // (its length and bytes travel in the body)
                continue;
            }
            uint32_t fi;
            if (kind[idx] == 0) {
                w64(pl.pk.first);
                w64(pl.pk.second);
                fi = pos_newref[idx];
                *w++ = sto.repetitive[fi];
            } else
                fi = (uint32_t)sto.new_ref_items.size() + pos_raw[idx];
            const size_t n = sto.fetched_off[fi + 1] - sto.fetched_off[fi];
            w32((uint32_t)n);
            memcpy(w, fetched.data() + sto.fetched_off[fi], n);
            w += n;
        }
    }
    dist_record_n = (size_t)(w - dist_record_ptr());
    return true;
}

// The body: every delta item of the head, in the same order: u32 length + bytes.  ~22 MB per human-size sample, gathered by the
// worker pool into pinned memory (its next stop is the writer's GPU or socket).
bool CAGCCompressor::Impl::make_record_body(const CommitData &cd)
{
    const std::vector<Placed> &placed = *cd.placed;
    std::vector<uint32_t> pos_enc(placed.size());
    for (uint32_t i = 0; i < cd.enc_items.size(); ++i)
        pos_enc[cd.enc_items[i]] = i;
    const size_t nb = dist_body_items.size();
    std::vector<uint64_t> body_off(nb + 1, 0);
    for (size_t i = 0; i < nb; ++i)
        body_off[i + 1] = body_off[i] + 4 + cd.enc_len[pos_enc[dist_body_items[i]]];
    dist_body_n = 0;
    if (!dist_body_buf.resize(body_off[nb] + body_off[nb] / 8 + 64, false)) {
        err("out of memory (commit record)");
        return false;
    }
    uint8_t *const dst = dist_body_buf.data();
    auto copy_range = [&](size_t from, size_t to) {
        for (size_t i = from; i < to; ++i) {
            const uint32_t ei = pos_enc[dist_body_items[i]], n = cd.enc_len[ei];
            uint8_t *d = dst + body_off[i];
            d[0] = (uint8_t)n, d[1] = (uint8_t)(n >> 8), d[2] = (uint8_t)(n >> 16), d[3] = (uint8_t)(n >> 24);
            if (n)
                memcpy(d + 4, cd.enc_ptr[ei], n);
        }
    };
    if (nb >= par_min) {
        const size_t n_chunks = std::min<size_t>(nb, (size_t)pool->size() * 4);
        pool->parallel_for(n_chunks, [&](size_t ci, unsigned) { copy_range(nb * ci / n_chunks, nb * (ci + 1) / n_chunks); });
    } else
        copy_range(0, nb);
    dist_body_n = body_off[nb];
    return true;
}

// the record of a sample without contigs: nothing to register anywhere
bool CAGCCompressor::Impl::make_empty_record()
{
    dist_body_n = 0;
    dist_record_n = 0;
    if (!dist_record_buf.resize(DIST_FRAME + 64, false)) {
        err("out of memory (commit record)");
        return false;
    }
    uint8_t *w = dist_record_ptr();
    memcpy(w, "AGCR", 4);
    const uint32_t f[5] = {0, 0, 0, ~0u, 0};
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 4; ++j)
            w[4 + 4 * i + j] = (uint8_t)(f[i] >> (8 * j));
    dist_record_n = 24;
    return true;
}

bool CAGCCompressor::Impl::apply_record(const uint8_t *rec, size_t n, const uint8_t *d_rec, const uint8_t *body, size_t body_n)
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
    if (n_ctg == 0 && n_lists == 0 && n_spl == 0 && n_new == 0 && !concatenated)
        return rr.p == rr.e; // empty sample: skipped on every rank (-c: the registration without contigs the reference sends at the end
                             // still moves the sample counters on the writer: after_registration)
    const bool writer = dist_rank == dist_writer;
    size_t body_pos = 0; // (a record without deltas has no body: every kind-2 payload is checked against body_n below)
    // the body came in through RecordBodyBuffer: the bookkeeping reads it where it is
    std::unique_ptr<PinnedBytes> adopted;
    if (body_recv && body == body_recv->data() && body_n <= body_recv->size())
        adopted = std::move(body_recv);
    if (!add.empty()) { // adaptive mode: the owner's new splitters (agc_compressor.cpp:1191-1209)
        ++spl_version; // (a sample prepared here before this point was scanned with the smaller set: CommitPreparedHead prepares it again)
        splitters.insert(splitters.end(), add.begin(), add.end());
        std::sort(splitters.begin(), splitters.end());
        splitters.erase(std::unique(splitters.begin(), splitters.end()), splitters.end());
        if (!hip_ok(DEVT(agc_hip_splitters_insert(hip, add.data(), add.size())), "splitters_insert"))
            return false;
    }
    if (writer) {
        std::lock_guard<std::mutex> coll_lk(coll_mtx);
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
    std::vector<uint64_t> ref_off{0}, raw_off{0}, enc_start;
    std::vector<uint32_t> enc_len;
    std::vector<uint32_t> reg_gid, reg_len;
    std::vector<uint64_t> reg_off; // payload offsets inside the record (device copy)
    for (uint32_t li = 0; li < n_lists && rr.ok; ++li) {
        const uint32_t gid = rr.u32(), cnt = rr.u32();
        // append mode: the first add to a group of the input archive unpacks it -- on every rank, as on the owner (stage_register;
        // segment.cpp:19-20, 39-40): the reference goes to this rank's HBM, the group stops answering as a packed one
        if (appending && gid < groups.size() && groups[gid].packed && !unpack_group(gid))
            return false;
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
            uint32_t pn = 0;
            if (kind != 2) {
                pn = rr.u32();
                if (!rr.need(pn))
                    break;
            } else if (writer) { // (length + bytes of a delta are in the body)
                if (body_pos + 4 > body_n) {
                    rr.ok = false;
                    break;
                }
                const uint8_t *q = body + body_pos;
                pn = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                body_pos += 4;
            }
            if (kind > 2 || pl.ctg >= n_ctg || gid >= groups.size()) {
                rr.ok = false;
                break;
            }
            if (!writer && kind != 0) { // the other ranks only need what classification reads: groups, keys, references
                if (kind == 1)
                    rr.p += pn;
                continue;
            }
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
                groups[gid].exists = true;
                groups[gid].ref_size = (uint64_t)pn + 1;
            } else if (kind == 1) {
                cd.raw_items.push_back(idx);
                raws.insert(raws.end(), rr.p, rr.p + pn);
                raw_off.push_back(raws.size());
            } else {
                if (body_pos + pn > body_n) {
                    rr.ok = false;
                    break;
                }
                cd.enc_items.push_back(idx);
                enc_start.push_back(adopted ? body_pos : enc.size());
                enc_len.push_back(pn);
                if (pn && !adopted)
                    enc.insert(enc.end(), body + body_pos, body + body_pos + pn);
                body_pos += pn;
            }
            if (kind != 2)
                rr.p += pn;
            sl.items.push_back(idx);
            placed.push_back(pl);
        }
    }
    sl.begin.push_back((uint32_t)sl.items.size());
    if (!rr.ok || rr.p != rr.e || (writer && body_pos != body_n)) {
        err("malformed commit record");
        return false;
    }
    if (writer)
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
    // fetched = new references, then raw items (the layout book_and_store indexes).  Everything the bookkeeping reads is the
    // task's own: it runs beside the next owner's commit
    std::unique_ptr<BookTask> t(new BookTask());
    bytes_t &fetched = t->fetched;
    fetched.reserve(refs_raw.size() + raws.size());
    fetched.insert(fetched.end(), refs_raw.begin(), refs_raw.end());
    fetched.insert(fetched.end(), raws.begin(), raws.end());
    cd.fetched_off = ref_off;
    for (size_t i = 1; i < raw_off.size(); ++i)
        cd.fetched_off.push_back(refs_raw.size() + raw_off[i]);
    t->ctgs = std::move(ctgs);
    t->placed = std::move(placed);
    t->enc = std::move(enc);
    t->owner = this;
    t->enc_recv = std::move(adopted);
    const uint8_t *const enc_base = t->enc_recv ? t->enc_recv->data() : t->enc.data();
    for (size_t i = 0; i < enc_start.size(); ++i)
        cd.enc_ptr.push_back(enc_base + enc_start[i]);
    cd.enc_len = std::move(enc_len);
    t->cd = std::move(cd);
    t->cd.ctgs = &t->ctgs;
    t->cd.placed = &t->placed;
    t->cd.fetched = &t->fetched;
    if (book_can_async(1)) {
        book_submit(std::move(t));
        return true;
    }
    return book_wait() && book_and_store(t->cd);
}

} // namespace agc
