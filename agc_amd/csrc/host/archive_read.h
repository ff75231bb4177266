// archive_read.h -- parsing of the .agc v3 container and of its collection / segment streams, shared by the reader
// (reader.cpp) and by the append mode of the compressor (compressor.cpp).  Header-only, host code.
// Citations: file:line under the reference tree.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace agc {
namespace rd {

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
            h = dlopen(c, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (h)
                break;
        }
        if (!h)
            return false;
        decompress = (size_t(*)(void *, size_t, const void *, size_t))dlsym(h, "ZSTD_decompress");
        isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        return decompress && isError;
    }
    bool unzstd(const uint8_t *src, size_t n, size_t raw, bytes_t &out) const
    {
        out.resize(raw);
        const size_t r = decompress(out.data(), raw, src, n);
        if (isError(r))
            return false;
        out.resize(r);
        return true;
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
    // The file is mapped, not read: a query touches the footer and the parts it asks for, as the reference's reader does
    // (archive.cpp:88-139 reads the footer, :378-402 one part at a time); pointers handed out stay valid until the archive goes.
    struct Mapped {
        const uint8_t *p = nullptr;
        size_t n = 0;
        bytes_t fallback; // (something that cannot be mapped: read whole)
        bool mapped = false;
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        const uint8_t *data() const { return p; }
        uint8_t operator[](size_t i) const { return p[i]; }
        void release()
        {
            if (mapped && p)
                munmap((void *)p, n);
            p = nullptr;
            n = 0;
            mapped = false;
            fallback.clear();
        }
        bool open(const std::string &fn, bool map)
        {
            release();
            const int fd = ::open(fn.c_str(), O_RDONLY);
            if (fd < 0)
                return false;
            struct stat st;
            // (a mapped file must not be truncated or rewritten in place while it is open -- the pages would fault; append reads its
            // input whole for that reason, and AGC_AMD_NO_MMAP=1 makes every reader do so)
            static const bool no_mmap = getenv("AGC_AMD_NO_MMAP") != nullptr;
            if (map && !no_mmap && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    p = (const uint8_t *)m;
                    n = (size_t)st.st_size;
                    mapped = true;
                    ::close(fd);
                    return true;
                }
            }
            uint8_t tmp[1 << 16];
            for (;;) {
                const ssize_t r = ::read(fd, tmp, sizeof(tmp));
                if (r <= 0)
                    break;
                fallback.insert(fallback.end(), tmp, tmp + r);
            }
            ::close(fd);
            p = fallback.data();
            n = fallback.size();
            return true;
        }
        Mapped() = default;
        Mapped(const Mapped &) = delete;
        Mapped &operator=(const Mapped &) = delete;
        ~Mapped() { release(); }
    };
    Mapped data;
    std::vector<Stream> streams;
    std::unordered_map<std::string, int> ids;

    static bool num(const Mapped &d, uint64_t &p, uint64_t &v)
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
    // map = false: the whole file is read (append: the output may replace the input file while its parts are still in use)
    bool open(const std::string &fn, bool map = true)
    {
        if (!data.open(fn, map) || data.size() < 9)
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
        if (!num(data, p, meta) || pt.size > data.size() - p) // (no wrap-around on a corrupt footer)
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

// prefix varint of the collection streams (src/common/collection.h:175-206)
inline bool rd_num(const uint8_t *&p, const uint8_t *e, uint32_t &num)
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
inline bool rd_str(const uint8_t *&p, const uint8_t *e, std::string &s)
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
inline uint64_t zigzag_decode_pred(uint64_t v, uint64_t prev)
{
    if (v >= 2 * prev)
        return v;
    if (v & 1)
        return (2 * prev - v) / 2;
    return (v + 2 * prev) / 2;
}
inline std::vector<std::string> split_string(const std::string &s)
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
inline std::string decode_split(const std::vector<std::string> &prev, std::vector<std::string> &cur)
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
inline std::string short_name(const std::string &s) // collection.cpp:19-28
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

// tuples2bytes, src/common/segment.h:92-138 (one table look-up per packed byte instead of a division per symbol)
inline bool tuples2bytes(const bytes_t &t, bytes_t &out)
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
    out.resize(n + 4); // room for whole-word stores; trimmed below
    uint32_t lut[256]; // the nb symbols of a packed byte, first symbol in the low byte
    for (uint32_t v = 0; v < 256; ++v) {
        uint32_t c = v, w = 0;
        for (int k = (int)nb - 1; k >= 0; --k) {
            w |= (c % mult) << (8 * k);
            c /= mult;
        }
        lut[v] = w;
    }
    size_t i = 0, j = 0;
    uint8_t *o = out.data();
    for (; j + nb <= n; ++i, j += nb)
        memcpy(o + j, &lut[t[i]], 4); // (little endian; the bytes past nb are overwritten by the next store)
    uint8_t c = t[i];
    const uint32_t r = (uint32_t)(n % nb);
    for (int k = (int)r - 1; k >= 0; --k) {
        o[j + k] = c % mult;
        c /= mult;
    }
    out.resize(n);
    return true;
}

// CLZDiff_V2::Decode, src/common/lz_diff.cpp:801-836
inline bool lz_decode(const bytes_t &ref, uint32_t mml, const uint8_t *enc, size_t n, bytes_t &out)
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

inline void reverse_complement(bytes_t &s) // agc_basic.cpp:253-280
{
    std::reverse(s.begin(), s.end());
    for (auto &c : s)
        if (c < 4)
            c = (uint8_t)(3 - c);
}


// sample names, collection_v3.cpp:152-165, 337-353
inline bool parse_sample_names(const Archive &ar, const ZstdD &z, std::vector<SampleDesc> &samples)
{
    const uint8_t *ptr;
    uint64_t size, meta;
    if (!ar.get_part("collection-samples", 0, ptr, size, meta))
        return false;
    bytes_t raw;
    if (!z.unzstd(ptr, size, meta, raw))
        return false;
    const uint8_t *q = raw.data(), *e = raw.data() + raw.size();
    uint32_t ns = 0;
    if (!rd_num(q, e, ns))
        return false;
    samples.assign(ns, SampleDesc());
    for (uint32_t i = 0; i < ns; ++i)
        if (!rd_str(q, e, samples[i].name))
            return false;
    return true;
}

// load_batch_contig_names / load_batch_contig_details, collection_v3.cpp:196-213, 270-326, 498-659
inline bool parse_contig_batch(const Archive &ar, const ZstdD &z, uint32_t batch, uint32_t pack, uint32_t segment_size, uint32_t k,
                               std::vector<SampleDesc> &samples)
{
    const uint8_t *ptr;
    uint64_t size, meta;
    bytes_t raw;
    const size_t first = (size_t)batch * pack;
    if (!ar.get_part("collection-contigs", batch, ptr, size, meta) || !z.unzstd(ptr, size, meta, raw))
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
            if (p + ps[i] > e || !z.unzstd(p, ps[i], rs[i], d[i]))
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

// the reference sequence of a group (part 0 of x<id>r), CSegment::get / unpack (segment.cpp:253-272, 500-523)
inline bool decode_ref_part(const ZstdD &z, const uint8_t *ptr, uint64_t size, uint64_t meta, bytes_t &ref)
{
    if (meta == 0) {
        ref.assign(ptr, ptr + size);
        return true;
    }
    if (size < 1)
        return false;
    const uint8_t marker = ptr[size - 1];
    bytes_t tmp;
    if (!z.unzstd(ptr, size - 1, meta + 1, tmp))
        return false;
    if (marker == 0) {
        ref.swap(tmp);
        return true;
    }
    return tuples2bytes(tmp, ref);
}

// one pack of a delta / raw stream: sequences, each followed by 0xFF (segment.cpp:150-165, 536-545);
// the stored part = zstd frame + one marker byte (segment.h:177-183)
inline bool decode_pack_part(const ZstdD &z, const uint8_t *ptr, uint64_t size, uint64_t meta, bytes_t &raw)
{
    if (meta == 0) {
        raw.assign(ptr, ptr + size);
        return true;
    }
    return size >= 1 && z.unzstd(ptr, size - 1, meta, raw);
}

} // namespace rd
} // namespace agc
