// reader.cpp -- see reader.h.  Citations: file:line under the reference tree.
#include "reader.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <iostream>
#include <map>
#include <atomic>
#include <thread>
#include <unordered_map>

#include "archive_read.h"

namespace agc {

using namespace rd;

struct CAGCFile::Impl {
    bool opened = false;
    Archive ar;
    ZstdD z;
    uint32_t k = 0, mml = 0, pack = 0, segment_size = 0;
    mutable std::vector<SampleDesc> samples;
    std::unordered_map<std::string, uint32_t> sample_ids;
    std::map<std::string, std::string> file_type_info;
    // decoded group references and delta / raw packs (prefetch-style caches)
    mutable std::unordered_map<uint32_t, bytes_t> ref_cache;
    mutable std::map<std::pair<uint32_t, uint32_t>, bytes_t> pack_cache;

    bool load_batch(uint32_t batch) const { return parse_contig_batch(ar, z, batch, pack, segment_size, k, samples); }
    bool ensure_sample(uint32_t sid) const
    {
        if (sid >= samples.size())
            return false;
        if (!samples[sid].loaded && !load_batch(sid / pack))
            return false;
        return samples[sid].loaded;
    }
    void warm_refs(const SampleDesc &s) const;
    bool get_pack(uint32_t gid, uint32_t part, const bytes_t *&out) const;
    bool get_ref(uint32_t gid, const bytes_t *&out) const;
    bool get_segment(uint32_t gid, uint32_t igid, bytes_t &out) const;
    int find_contig(const std::string &sample, const std::string &name, uint32_t &sid, uint32_t &cid) const;
    bool decode_contig(const CtgDesc &c, int64_t from, int64_t to, bytes_t &out) const;
};

// one pack of a group's delta stream, decoded (CSegment::get / get_raw, segment.cpp:136-400)
bool CAGCFile::Impl::get_pack(uint32_t gid, uint32_t part, const bytes_t *&out) const
{
    auto key = std::make_pair(gid, part);
    auto it = pack_cache.find(key);
    if (it == pack_cache.end()) {
        const uint8_t *ptr;
        uint64_t size, meta;
        if (!ar.get_part("x" + int_to_base64(gid) + "d", part, ptr, size, meta))
            return false;
        bytes_t raw;
        if (!decode_pack_part(z, ptr, size, meta, raw))
            return false;
        if (pack_cache.size() > 4096)
            pack_cache.clear();
        it = pack_cache.emplace(key, std::move(raw)).first;
    }
    out = &it->second;
    return true;
}

bool CAGCFile::Impl::get_ref(uint32_t gid, const bytes_t *&out) const
{
    auto it = ref_cache.find(gid);
    if (it == ref_cache.end()) {
        const uint8_t *ptr;
        uint64_t size, meta;
        if (!ar.get_part("x" + int_to_base64(gid) + "r", 0, ptr, size, meta))
            return false;
        bytes_t ref;
        if (!decode_ref_part(z, ptr, size, meta, ref))
            return false;
        if (ref_cache.size() > 65536)
            ref_cache.clear();
        it = ref_cache.emplace(gid, std::move(ref)).first;
    }
    out = &it->second;
    return true;
}

// whole-sample extraction: the group references the sample needs are decoded (zstd + tuples) by a few threads up front;
// the sequential assembly below then finds them in the cache
void CAGCFile::Impl::warm_refs(const SampleDesc &s) const
{
    std::vector<uint32_t> need;
    for (auto &c : s.ctgs)
        for (auto &sg : c.segs)
            if (sg.group_id >= NO_RAW_GROUPS && !ref_cache.count(sg.group_id))
                need.push_back(sg.group_id);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    if (need.size() < 64 || need.size() + ref_cache.size() > 65536)
        return;
    const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<bytes_t> dec(need.size());
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < need.size();) {
            const uint8_t *ptr;
            uint64_t size, meta;
            if (ar.get_part("x" + int_to_base64(need[i]) + "r", 0, ptr, size, meta))
                decode_ref_part(z, ptr, size, meta, dec[i]);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t)
        th.emplace_back(work);
    work();
    for (auto &t : th)
        t.join();
    for (size_t i = 0; i < need.size(); ++i)
        if (!dec[i].empty())
            ref_cache.emplace(need[i], std::move(dec[i]));
}

static bool nth_in_pack(const bytes_t &pack, uint32_t idx, const uint8_t *&b, size_t &n)
{
    size_t start = 0;
    uint32_t cnt = 0;
    for (size_t i = 0; i < pack.size(); ++i)
        if (pack[i] == 0xff) {
            if (cnt == idx) {
                b = pack.data() + start;
                n = i - start;
                return true;
            }
            ++cnt;
            start = i + 1;
        }
    return false;
}

bool CAGCFile::Impl::get_segment(uint32_t gid, uint32_t igid, bytes_t &out) const
{
    const bytes_t *pk;
    const uint8_t *b;
    size_t n;
    if (gid < NO_RAW_GROUPS) {
        if (!get_pack(gid, igid / pack, pk) || !nth_in_pack(*pk, igid % pack, b, n))
            return false;
        out.assign(b, b + n);
        return true;
    }
    const bytes_t *ref;
    if (!get_ref(gid, ref))
        return false;
    if (igid == 0) {
        out = *ref;
        return true;
    }
    if (!get_pack(gid, (igid - 1) / pack, pk) || !nth_in_pack(*pk, (igid - 1) % pack, b, n))
        return false;
    return lz_decode(*ref, mml, b, n, out);
}

// get_contig_desc semantics (collection_v3.cpp:849-900): full name or short name (up to the first white space)
int CAGCFile::Impl::find_contig(const std::string &sample, const std::string &name, uint32_t &sid, uint32_t &cid) const
{
    const std::string sn = short_name(name);
    auto match = [&](uint32_t s) -> int {
        if (!ensure_sample(s))
            return -1;
        for (uint32_t c = 0; c < samples[s].ctgs.size(); ++c)
            if (samples[s].ctgs[c].name == name || short_name(samples[s].ctgs[c].name) == sn)
                return (int)c;
        return -1;
    };
    if (!sample.empty()) {
        auto it = sample_ids.find(sample);
        if (it == sample_ids.end())
            return -1;
        int c = match(it->second);
        if (c < 0)
            return -1;
        sid = it->second;
        cid = (uint32_t)c;
        return 0;
    }
    int found = 0;
    for (uint32_t s = 0; s < samples.size(); ++s) {
        int c = match(s);
        if (c >= 0) {
            if (found++)
                return -2;
            sid = s;
            cid = (uint32_t)c;
        }
    }
    return found ? 0 : -1;
}

// decompress_contig, agc_decompressor_lib.cpp:172-286
bool CAGCFile::Impl::decode_contig(const CtgDesc &c, int64_t from, int64_t to, bytes_t &out) const
{
    if (from < 0 && to < 0) {
        from = 0;
        to = 0x7fffffffffffffffLL;
    } else {
        if (from < 0)
            from = 0;
        if (to < 0)
            to = 0x7fffffffffffffffLL;
        if (from > to) {
            from = 0;
            to = 0x7fffffffffffffffLL;
        }
    }
    int64_t curr_pos = 0;
    bool first = true;
    bytes_t seg;
    out.clear();
    for (const SegDesc &s : c.segs) {
        const int64_t seg_len = s.raw_length;
        if (curr_pos + seg_len < from) {
            from -= seg_len - k;
            to -= seg_len - k;
            continue;
        } else if (curr_pos > to)
            break;
        if (!get_segment(s.group_id, s.in_group_id, seg))
            return false;
        if (s.rc)
            reverse_complement(seg);
        if (first)
            out.swap(seg);
        else {
            if (seg.size() < k)
                return false;
            out.insert(out.end(), seg.begin() + k, seg.end());
        }
        first = false;
        curr_pos += seg_len - k;
    }
    if (first)
        return true;
    if (out.size() > (uint64_t)to + 1)
        out.resize((size_t)to + 1);
    if (from != 0)
        out.erase(out.begin(), out.begin() + std::min<int64_t>(from, (int64_t)out.size()));
    return true;
}

CAGCFile::CAGCFile() : p(new Impl) {}
CAGCFile::~CAGCFile() {}

bool CAGCFile::Open(const std::string &file_name, bool)
{
    Impl &I = *p;
    if (I.opened)
        return false;
    if (!I.z.load() || !I.ar.open(file_name))
        return false;
    const uint8_t *ptr;
    uint64_t size, meta;
    // file_type_info: key\0value\0 pairs (agc_basic.cpp:60-100); only format 3.x is handled here
    if (!I.ar.get_part("file_type_info", 0, ptr, size, meta))
        return false;
    {
        std::map<std::string, std::string> &info = I.file_type_info;
        const uint8_t *q = ptr, *e = ptr + size;
        std::string key, val;
        while (q < e && rd_str(q, e, key) && rd_str(q, e, val))
            info[key] = val;
        if (info["file_version_major"] != "3")
            return false;
    }
    // params (agc_compressor.cpp:206-217): k, min_match_len, pack_cardinality, segment_size as LE u32
    if (!I.ar.get_part("params", 0, ptr, size, meta) || size < 16)
        return false;
    auto le32 = [&](size_t o) { return (uint32_t)ptr[o] | ((uint32_t)ptr[o + 1] << 8) | ((uint32_t)ptr[o + 2] << 16) | ((uint32_t)ptr[o + 3] << 24); };
    I.k = le32(0);
    I.mml = le32(4);
    I.pack = le32(8);
    I.segment_size = le32(12);
    if (!I.pack)
        return false;
    if (!parse_sample_names(I.ar, I.z, I.samples))
        return false;
    for (uint32_t i = 0; i < I.samples.size(); ++i)
        I.sample_ids[I.samples[i].name] = i;
    I.opened = true;
    return true;
}

bool CAGCFile::Close()
{
    if (!p->opened)
        return false;
    p.reset(new Impl);
    return true;
}

int CAGCFile::NSample() const { return p->opened ? (int)p->samples.size() : -1; }

int CAGCFile::NCtg(const std::string &sample) const
{
    if (!p->opened)
        return -1;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return -1;
    return (int)p->samples[it->second].ctgs.size();
}

int CAGCFile::ListSample(std::vector<std::string> &out) const
{
    if (!p->opened)
        return -1;
    out.clear();
    for (auto &s : p->samples)
        out.push_back(s.name);
    std::sort(out.begin(), out.end()); // get_samples_list(sorted = true), collection_v3.cpp:821-835
    return 0;
}

int CAGCFile::ListSampleStored(std::vector<std::string> &out) const
{
    if (!p->opened)
        return -1;
    out.clear();
    for (auto &s : p->samples)
        out.push_back(s.name);
    return 0;
}

int CAGCFile::GetReferenceSample(std::string &sample) const
{
    if (!p->opened || p->samples.empty())
        return -1;
    sample = p->samples.front().name;
    return 0;
}

int CAGCFile::ListCtg(const std::string &sample, std::vector<std::string> &names) const
{
    if (!p->opened)
        return -1;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return -1;
    names.clear();
    for (auto &c : p->samples[it->second].ctgs)
        names.push_back(c.name);
    return 0;
}

bool CAGCFile::GetParams(uint32_t &k, uint32_t &mml, uint32_t &pack, uint32_t &segment_size) const
{
    if (!p->opened)
        return false;
    k = p->k;
    mml = p->mml;
    pack = p->pack;
    segment_size = p->segment_size;
    return true;
}

int64_t CAGCFile::GetCtgLen(const std::string &sample, const std::string &name) const
{
    if (!p->opened)
        return -1;
    uint32_t sid, cid;
    int r = p->find_contig(sample, name, sid, cid);
    if (r < 0)
        return r;
    const CtgDesc &c = p->samples[sid].ctgs[cid];
    int64_t len = 0;
    for (auto &s : c.segs)
        len += s.raw_length;
    return len - ((int64_t)c.segs.size() - 1) * p->k;
}

static const char ALPHA[] = "ACGTNRYSWKMBDHVU"; // CNumAlphaConverter, agc_decompressor_lib.h:24-34 (others -> ' ')

int CAGCFile::GetCtgSeq(const std::string &sample, const std::string &name, int64_t start, int64_t end, std::string &buffer) const
{
    if (!p->opened)
        return -1;
    uint32_t sid, cid;
    int r = p->find_contig(sample, name, sid, cid);
    if (r < 0)
        return r;
    bytes_t codes;
    if (!p->decode_contig(p->samples[sid].ctgs[cid], start, end, codes))
        return -1;
    buffer.resize(codes.size());
    for (size_t i = 0; i < codes.size(); ++i)
        buffer[i] = codes[i] < 16 ? ALPHA[codes[i]] : ' ';
    return 0;
}

static void append_fasta(std::string &out, const std::string &name, const bytes_t &codes, uint32_t line_length)
{
    static const struct Lut {
        char t[256];
        Lut()
        {
            for (int i = 0; i < 256; ++i)
                t[i] = i < 16 ? ALPHA[i] : ' ';
        }
    } lut;
    out.push_back('>');
    out.append(name);
    out.push_back('\n');
    const size_t ll = line_length ? line_length : codes.size();
    size_t o = out.size();
    out.resize(o + codes.size() + (ll ? (codes.size() + ll - 1) / ll : 0));
    char *dst = &out[0];
    const uint8_t *src = codes.data();
    for (size_t i = 0; i < codes.size(); i += ll) {
        const size_t n = std::min<size_t>(ll, codes.size() - i);
        for (size_t j = 0; j < n; ++j)
            dst[o + j] = lut.t[src[i + j]];
        o += n;
        dst[o++] = '\n';
    }
}

bool CAGCFile::GetContigFasta(const std::string &query, std::string &out, uint32_t line_length, std::string &err) const
{
    if (!p->opened)
        return false;
    // the four query forms, tried in the reference's order with greedy (.+) groups
    std::string name = query, sample;
    int64_t from = -1, to = -1;
    auto parse_range = [&](const std::string &s, std::string &head) {
        // (.+):(.+)-(.+) greedy: last ':' that still leaves a '-' after it
        const size_t c = s.rfind(':');
        if (c == std::string::npos || c == 0)
            return false;
        const size_t d = s.rfind('-');
        if (d == std::string::npos || d <= c + 1 || d + 1 >= s.size())
            return false;
        head = s.substr(0, c);
        from = atoll(s.substr(c + 1, d - c - 1).c_str());
        to = atoll(s.substr(d + 1).c_str());
        return true;
    };
    const size_t at = query.rfind('@');
    if (at != std::string::npos && at > 0 && at + 1 < query.size()) {
        name = query.substr(0, at);
        std::string rest = query.substr(at + 1), head;
        if (parse_range(rest, head))
            sample = head;
        else
            sample = rest;
    } else {
        std::string head;
        if (parse_range(query, head))
            name = head;
    }
    uint32_t sid, cid;
    const int r = p->find_contig(sample, name, sid, cid);
    if (r < 0) {
        if (!sample.empty())
            err = "There is no sample:contig pair: " + sample + " : " + name;
        else if (r == -2)
            err = "There are several samples with conting: " + name;
        else
            err = "There is no contig: " + name;
        return false;
    }
    const CtgDesc &c = p->samples[sid].ctgs[cid];
    bytes_t codes;
    if (!p->decode_contig(c, from, to, codes))
        return false;
    std::string hdr = c.name;
    if (from >= 0 && to >= 0)
        hdr += ":" + std::to_string(from) + "-" + std::to_string(to);
    append_fasta(out, hdr, codes, line_length);
    return true;
}

bool CAGCFile::GetFileTypeInfo(std::vector<std::pair<std::string, std::string>> &info) const
{
    if (!p->opened)
        return false;
    info.assign(p->file_type_info.begin(), p->file_type_info.end());
    return true;
}

bool CAGCFile::WriteSampleFasta(const std::string &sample, FILE *f, uint32_t line_length) const
{
    if (!p->opened || !f)
        return false;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return false;
    p->warm_refs(p->samples[it->second]);
    bytes_t codes;
    std::string text;
    for (auto &c : p->samples[it->second].ctgs) {
        if (!p->decode_contig(c, -1, -1, codes))
            return false;
        text.clear();
        append_fasta(text, c.name, codes, line_length);
        if (fwrite(text.data(), 1, text.size(), f) != text.size())
            return false;
    }
    return true;
}

bool CAGCFile::GetSampleCodes(const std::string &sample, std::vector<std::string> &names, std::vector<std::vector<uint8_t>> &codes) const
{
    if (!p->opened)
        return false;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return false;
    names.clear();
    codes.clear();
    for (auto &c : p->samples[it->second].ctgs) {
        names.push_back(c.name);
        codes.emplace_back();
        if (!p->decode_contig(c, -1, -1, codes.back()))
            return false;
    }
    return true;
}

bool CAGCFile::GetSampleFasta(const std::string &sample, std::string &out, uint32_t line_length) const
{
    if (!p->opened)
        return false;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return false;
    out.clear();
    p->warm_refs(p->samples[it->second]);
    bytes_t codes;
    for (auto &c : p->samples[it->second].ctgs) {
        if (!p->decode_contig(c, -1, -1, codes))
            return false;
        append_fasta(out, c.name, codes, line_length);
    }
    return true;
}

} // namespace agc
