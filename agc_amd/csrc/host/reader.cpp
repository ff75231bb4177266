// reader.cpp -- see reader.h.  Citations: file:line under the reference tree.
#include "reader.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <iostream>
#include <map>
#include <unordered_map>

namespace agc {

namespace {

using bytes_t = std::vector<uint8_t>;
constexpr uint32_t NO_RAW_GROUPS = 16;

struct ZstdD {
    void *h = nullptr;
    size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool load()
    {
        if (h)
            return true;
        const char *cands[] = {getenv("AGC_ZSTD_LIB"), "/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so"};
        for (const char *c : cands) {
            if (!c)
                continue;
            h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (h)
                break;
        }
        if (!h)
            return false;
        decompress = (size_t(*)(void *, size_t, const void *, size_t))dlsym(h, "ZSTD_decompress");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        return decompress && isError;
    }
};

// container, src/common/archive.cpp:172-237 (deserialize), archive.h:127-147
struct Archive {
    struct Part {
        uint64_t offset, size;
    };
    struct Stream {
        std::string name;
        uint64_t raw_size = 0;
        std::vector<Part> parts;
    };
    bytes_t data;
    std::vector<Stream> streams;
    std::unordered_map<std::string, int> ids;

    static bool num(const bytes_t &d, uint64_t &p, uint64_t &v)
    {
        if (p >= d.size())
            return false;
        const uint32_t n = d[p++];
        if (n > 8 || p + n > d.size())
            return false;
        v = 0;
        for (uint32_t i = 0; i < n; ++i)
            v = (v << 8) | d[p++];
        return true;
    }
    bool open(const std::string &fn)
    {
        FILE *f = fopen(fn.c_str(), "rb");
        if (!f)
            return false;
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        data.resize(sz > 0 ? (size_t)sz : 0);
        const size_t rd = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
        fclose(f);
        if (rd != data.size() || data.size() < 9)
            return false;
        uint64_t fs = 0;
        for (int i = 0; i < 8; ++i)
            fs |= (uint64_t)data[data.size() - 8 + i] << (8 * i);
        if (fs + 8 > data.size())
            return false;
        uint64_t p = data.size() - 8 - fs, n_streams = 0;
        if (!num(data, p, n_streams))
            return false;
        for (uint64_t s = 0; s < n_streams; ++s) {
            Stream st;
            while (p < data.size() && data[p])
                st.name.push_back((char)data[p++]);
            ++p;
            uint64_t n_parts = 0;
            if (!num(data, p, n_parts) || !num(data, p, st.raw_size))
                return false;
            for (uint64_t i = 0; i < n_parts; ++i) {
                Part pt;
                if (!num(data, p, pt.offset) || !num(data, p, pt.size))
                    return false;
                st.parts.push_back(pt);
            }
            ids[st.name] = (int)streams.size();
            streams.emplace_back(std::move(st));
        }
        return true;
    }
    // part = varint(metadata) + payload (archive.cpp:378-402)
    bool get_part(const std::string &stream, size_t idx, const uint8_t *&ptr, uint64_t &size, uint64_t &meta) const
    {
        auto it = ids.find(stream);
        if (it == ids.end() || idx >= streams[it->second].parts.size())
            return false;
        const Part &pt = streams[it->second].parts[idx];
        uint64_t p = pt.offset;
        if (!num(data, p, meta) || p + pt.size > data.size())
            return false;
        ptr = data.data() + p;
        size = pt.size;
        return true;
    }
    size_t n_parts(const std::string &stream) const
    {
        auto it = ids.find(stream);
        return it == ids.end() ? 0 : streams[it->second].parts.size();
    }
};

std::string int_to_base64(uint32_t n)
{
    static const char dig[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_#";
    std::string r;
    do {
        r.push_back(dig[n & 0x3fu]);
        n /= 64;
    } while (n);
    return r;
}

// prefix varint of the collection streams (src/common/collection.h:175-206)
bool rd_num(const uint8_t *&p, const uint8_t *e, uint32_t &num)
{
    if (p >= e)
        return false;
    const uint32_t thr_1 = 1u << 7, thr_2 = thr_1 + (1u << 14), thr_3 = thr_2 + (1u << 21), thr_4 = thr_3 + (1u << 28);
    if ((*p & 0x80u) == 0) {
        num = *p++;
    } else if ((*p & 0xC0u) == 0x80u) {
        if (p + 2 > e)
            return false;
        num = ((uint32_t)p[0] << 8) + p[1] + thr_1 - (0x80u << 8);
        p += 2;
    } else if ((*p & 0xE0u) == 0xC0u) {
        if (p + 3 > e)
            return false;
        num = ((uint32_t)p[0] << 16) + ((uint32_t)p[1] << 8) + p[2] + thr_2 - (0xC0u << 16);
        p += 3;
    } else if ((*p & 0xF0u) == 0xE0u) {
        if (p + 4 > e)
            return false;
        num = ((uint32_t)p[0] << 24) + ((uint32_t)p[1] << 16) + ((uint32_t)p[2] << 8) + p[3] + thr_3 - (0xE0u << 24);
        p += 4;
    } else {
        if (p + 5 > e)
            return false;
        num = ((uint32_t)p[1] << 24) + ((uint32_t)p[2] << 16) + ((uint32_t)p[3] << 8) + p[4] + thr_4;
        p += 5;
    }
    return true;
}
bool rd_str(const uint8_t *&p, const uint8_t *e, std::string &s)
{
    const uint8_t *q = p;
    while (q < e && *q)
        ++q;
    if (q >= e)
        return false;
    s.assign((const char *)p, (size_t)(q - p));
    p = q + 1;
    return true;
}
// zigzag vs prediction, src/common/utils.h:125-136
uint64_t zigzag_decode_pred(uint64_t v, uint64_t prev)
{
    if (v >= 2 * prev)
        return v;
    if (v & 1)
        return (2 * prev - v) / 2;
    return (v + 2 * prev) / 2;
}
std::vector<std::string> split_string(const std::string &s)
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
// collection_v3.cpp:424-465
std::string decode_split(const std::vector<std::string> &prev, std::vector<std::string> &cur)
{
    std::string dec;
    for (size_t i = 0; i < cur.size(); ++i) {
        if (cur[i].size() == 1 && (signed char)cur[i][0] == -127) {
            dec.append(prev[i]);
            cur[i] = prev[i];
        } else {
            std::string cmp;
            size_t pp = 0;
            for (char ch : cur[i]) {
                const signed char c = (signed char)ch;
                if (c >= 0) {
                    cmp.push_back(ch);
                    ++pp;
                } else {
                    cmp.append(prev[i], pp, (size_t)(-c));
                    pp += (size_t)(-c);
                }
            }
            dec.append(cmp);
            cur[i] = std::move(cmp);
        }
        dec.push_back(' ');
    }
    dec.pop_back();
    return dec;
}
std::string short_name(const std::string &s) // collection.cpp:19-28
{
    size_t p = 0;
    for (; p < s.size(); ++p)
        if (s[p] == ' ' || s[p] == '\n' || s[p] == '\r' || s[p] == '\t')
            break;
    return s.substr(0, p);
}

struct SegDesc {
    uint32_t group_id, in_group_id, raw_length;
    bool rc;
};
struct CtgDesc {
    std::string name;
    std::vector<SegDesc> segs;
};
struct SampleDesc {
    std::string name;
    std::vector<CtgDesc> ctgs;
    bool loaded = false;
};

// tuples2bytes, src/common/segment.h:92-138
bool tuples2bytes(const bytes_t &t, bytes_t &out)
{
    if (t.size() < 2)
        return false;
    const uint8_t marker = t.back();
    const uint32_t nb = marker >> 4, trailing = marker & 0xf;
    if (nb != 4 && nb != 3 && nb != 2) {
        out.assign(t.begin(), t.end() - 1);
        return true;
    }
    const uint32_t mult = nb == 4 ? 4 : nb == 3 ? 6 : 16;
    const size_t n = (t.size() - 2) * nb + trailing;
    out.resize(n);
    size_t i = 0, j = 0;
    for (; j + nb <= n; ++i, j += nb) {
        uint8_t c = t[i];
        for (int k = (int)nb - 1; k >= 0; --k) {
            out[j + k] = c % mult;
            c /= mult;
        }
    }
    uint8_t c = t[i];
    const uint32_t r = (uint32_t)(n % nb);
    for (int k = (int)r - 1; k >= 0; --k) {
        out[j + k] = c % mult;
        c /= mult;
    }
    return true;
}

// CLZDiff_V2::Decode, src/common/lz_diff.cpp:801-836
bool lz_decode(const bytes_t &ref, uint32_t mml, const uint8_t *enc, size_t n, bytes_t &out)
{
    out.clear();
    size_t p = 0;
    uint32_t pred_pos = 0;
    auto read_int = [&](int64_t &x) {
        bool neg = false;
        x = 0;
        if (p < n && enc[p] == '-') {
            neg = true;
            ++p;
        }
        while (p < n && enc[p] >= '0' && enc[p] <= '9')
            x = x * 10 + (enc[p++] - '0');
        if (neg)
            x = -x;
    };
    while (p < n) {
        const uint8_t c = enc[p];
        if ((c >= 'A' && c <= 'A' + 20) || c == '!') {
            if (c == '!') {
                if (pred_pos >= ref.size())
                    return false;
                out.push_back(ref[pred_pos]);
            } else
                out.push_back((uint8_t)(c - 'A'));
            ++pred_pos;
            ++p;
        } else if (c == 30) {
            ++p;
            int64_t v;
            read_int(v);
            ++p;
            out.insert(out.end(), (size_t)(v + 4), 4);
        } else {
            int64_t v;
            read_int(v);
            const uint32_t ref_pos = (uint32_t)(v + (int64_t)pred_pos);
            uint32_t len;
            if (p < n && enc[p] == ',') {
                ++p;
                int64_t l;
                read_int(l);
                len = (uint32_t)(l + mml);
            } else
                len = (uint32_t)ref.size() - ref_pos;
            ++p; // '.'
            if ((uint64_t)ref_pos + len > ref.size())
                return false;
            out.insert(out.end(), ref.begin() + ref_pos, ref.begin() + ref_pos + len);
            pred_pos = ref_pos + len;
        }
    }
    return true;
}

void reverse_complement(bytes_t &s) // agc_basic.cpp:253-280
{
    std::reverse(s.begin(), s.end());
    for (auto &c : s)
        if (c < 4)
            c = (uint8_t)(3 - c);
}

} // namespace

struct CAGCFile::Impl {
    bool opened = false;
    Archive ar;
    ZstdD z;
    uint32_t k = 0, mml = 0, pack = 0, segment_size = 0;
    mutable std::vector<SampleDesc> samples;
    std::unordered_map<std::string, uint32_t> sample_ids;
    // decoded group references and delta / raw packs (prefetch-style caches)
    mutable std::unordered_map<uint32_t, bytes_t> ref_cache;
    mutable std::map<std::pair<uint32_t, uint32_t>, bytes_t> pack_cache;

    bool unzstd(const uint8_t *src, size_t n, size_t raw, bytes_t &out) const
    {
        out.resize(raw);
        const size_t r = z.decompress(out.data(), raw, src, n);
        if (z.isError(r))
            return false;
        out.resize(r);
        return true;
    }
    bool load_batch(uint32_t batch) const;
    bool ensure_sample(uint32_t sid) const
    {
        if (sid >= samples.size())
            return false;
        if (!samples[sid].loaded && !load_batch(sid / pack))
            return false;
        return samples[sid].loaded;
    }
    bool get_pack(uint32_t gid, uint32_t part, const bytes_t *&out) const;
    bool get_ref(uint32_t gid, const bytes_t *&out) const;
    bool get_segment(uint32_t gid, uint32_t igid, bytes_t &out) const;
    int find_contig(const std::string &sample, const std::string &name, uint32_t &sid, uint32_t &cid) const;
    bool decode_contig(const CtgDesc &c, int64_t from, int64_t to, bytes_t &out) const;
};

// load_batch_contig_names / load_batch_contig_details, collection_v3.cpp:196-213, 270-326, 498-659
bool CAGCFile::Impl::load_batch(uint32_t batch) const
{
    const uint8_t *ptr;
    uint64_t size, meta;
    bytes_t raw;
    const size_t first = (size_t)batch * pack;
    if (!ar.get_part("collection-contigs", batch, ptr, size, meta) || !unzstd(ptr, size, meta, raw))
        return false;
    {
        const uint8_t *p = raw.data(), *e = raw.data() + raw.size();
        uint32_t ns = 0;
        if (!rd_num(p, e, ns))
            return false;
        for (uint32_t i = 0; i < ns && first + i < samples.size(); ++i) {
            uint32_t nc = 0;
            if (!rd_num(p, e, nc))
                return false;
            auto &s = samples[first + i];
            s.ctgs.assign(nc, CtgDesc());
            std::vector<std::string> prev, cur;
            for (uint32_t j = 0; j < nc; ++j) {
                std::string enc;
                if (!rd_str(p, e, enc))
                    return false;
                cur = split_string(enc);
                s.ctgs[j].name = cur.size() != prev.size() ? enc : decode_split(prev, cur);
                prev = std::move(cur);
            }
        }
    }
    if (!ar.get_part("collection-details", batch, ptr, size, meta))
        return false;
    {
        const uint8_t *p = ptr, *e = ptr + size;
        uint32_t rs[5], ps[5];
        for (int i = 0; i < 5; ++i)
            if (!rd_num(p, e, rs[i]) || !rd_num(p, e, ps[i]))
                return false;
        std::array<bytes_t, 5> d;
        for (int i = 0; i < 5; ++i) {
            if (p + ps[i] > e || !unzstd(p, ps[i], rs[i], d[i]))
                return false;
            p += ps[i];
        }
        const uint8_t *q = d[0].data(), *qe = d[0].data() + d[0].size();
        uint32_t ns = 0;
        if (!rd_num(q, qe, ns))
            return false;
        size_t n_items = 0;
        for (uint32_t i = 0; i < ns && first + i < samples.size(); ++i) {
            uint32_t nc = 0;
            if (!rd_num(q, qe, nc))
                return false;
            auto &s = samples[first + i];
            if (s.ctgs.size() != nc)
                s.ctgs.resize(nc);
            for (uint32_t j = 0; j < nc; ++j) {
                uint32_t nseg = 0;
                if (!rd_num(q, qe, nseg))
                    return false;
                s.ctgs[j].segs.assign(nseg, SegDesc{0, 0, 0, false});
                n_items += nseg;
            }
        }
        std::array<std::vector<uint32_t>, 5> v;
        for (int i = 1; i < 5; ++i) {
            v[i].resize(n_items);
            const uint8_t *r = d[i].data(), *re = d[i].data() + d[i].size();
            for (size_t j = 0; j < n_items; ++j)
                if (!rd_num(r, re, v[i][j]))
                    return false;
        }
        std::vector<int> igids;
        auto get_ig = [&](uint32_t g) { return g >= igids.size() ? -1 : igids[g]; };
        auto set_ig = [&](uint32_t g, int val) {
            if (g >= igids.size())
                igids.resize((size_t)((int)(g * 1.2) + 1), -1);
            igids[g] = val;
        };
        const uint32_t pred_raw_length = segment_size + k;
        size_t it = 0;
        for (uint32_t i = 0; i < ns && first + i < samples.size(); ++i) {
            auto &s = samples[first + i];
            for (auto &c : s.ctgs)
                for (auto &sg : c.segs) {
                    const uint32_t g = v[1][it];
                    const int prev = get_ig(g);
                    const uint32_t e_in = v[2][it];
                    uint32_t c_in;
                    if (prev == -1)
                        c_in = e_in;
                    else if (e_in == 0)
                        c_in = 0;
                    else if (e_in == 1)
                        c_in = (uint32_t)(prev + 1);
                    else
                        c_in = (uint32_t)zigzag_decode_pred(e_in - 1u, (uint64_t)(prev + 1));
                    sg.group_id = g;
                    sg.in_group_id = c_in;
                    sg.raw_length = (uint32_t)zigzag_decode_pred(v[3][it], pred_raw_length);
                    sg.rc = v[4][it] != 0;
                    if ((int)c_in > prev && c_in > 0)
                        set_ig(g, (int)c_in);
                    ++it;
                }
            s.loaded = true;
        }
    }
    return true;
}

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
        if (meta == 0)
            raw.assign(ptr, ptr + size);
        else if (size < 1 || !unzstd(ptr, size - 1, meta, raw)) // the stored part = zstd frame + one marker byte (segment.h:177-183)
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
        if (meta == 0)
            ref.assign(ptr, ptr + size);
        else {
            if (size < 1)
                return false;
            const uint8_t marker = ptr[size - 1];
            bytes_t tmp;
            if (!unzstd(ptr, size - 1, meta + 1, tmp))
                return false;
            if (marker == 0)
                ref.swap(tmp);
            else if (!tuples2bytes(tmp, ref))
                return false;
        }
        if (ref_cache.size() > 65536)
            ref_cache.clear();
        it = ref_cache.emplace(gid, std::move(ref)).first;
    }
    out = &it->second;
    return true;
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
        std::map<std::string, std::string> info;
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
    // sample names (collection_v3.cpp:152-165, 337-353)
    if (!I.ar.get_part("collection-samples", 0, ptr, size, meta))
        return false;
    bytes_t raw;
    if (!I.unzstd(ptr, size, meta, raw))
        return false;
    const uint8_t *q = raw.data(), *e = raw.data() + raw.size();
    uint32_t ns = 0;
    if (!rd_num(q, e, ns))
        return false;
    I.samples.assign(ns, SampleDesc());
    for (uint32_t i = 0; i < ns; ++i) {
        if (!rd_str(q, e, I.samples[i].name))
            return false;
        I.sample_ids[I.samples[i].name] = i;
    }
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
    out.push_back('>');
    out.append(name);
    out.push_back('\n');
    const size_t ll = line_length ? line_length : codes.size();
    size_t o = out.size();
    out.resize(o + codes.size() + (ll ? (codes.size() + ll - 1) / ll : 0));
    for (size_t i = 0; i < codes.size(); i += ll) {
        const size_t n = std::min<size_t>(ll, codes.size() - i);
        for (size_t j = 0; j < n; ++j)
            out[o + j] = codes[i + j] < 16 ? ALPHA[codes[i + j]] : ' ';
        o += n;
        out[o++] = '\n';
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

bool CAGCFile::GetSampleFasta(const std::string &sample, std::string &out, uint32_t line_length) const
{
    if (!p->opened)
        return false;
    auto it = p->sample_ids.find(sample);
    if (it == p->sample_ids.end() || !p->ensure_sample(it->second))
        return false;
    out.clear();
    bytes_t codes;
    for (auto &c : p->samples[it->second].ctgs) {
        if (!p->decode_contig(c, -1, -1, codes))
            return false;
        append_fasta(out, c.name, codes, line_length);
    }
    return true;
}

} // namespace agc
