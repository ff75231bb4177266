// capi_read.cpp -- C entry points of include/agc_read.h over agc::CAGCFile (reader.h).
// Mirrors src/lib-cxx/lib-cxx.cpp:123-330 of the reference (same names, same return conventions).
#include "reader.h"

#include <cstdlib>
#include <cstring>

#include "../../../include/agc_read.h"

struct agc_t {
    agc::CAGCFile f;
};

static char **vec2list(const std::vector<std::string> &v)
{
    char **list = (char **)malloc(sizeof(char *) * (v.size() + 1));
    if (!list)
        return nullptr;
    for (size_t i = 0; i < v.size(); ++i) {
        list[i] = (char *)malloc(v[i].size() + 1);
        if (!list[i]) {
            for (size_t j = 0; j < i; ++j)
                free(list[j]);
            free(list);
            return nullptr;
        }
        memcpy(list[i], v[i].c_str(), v[i].size() + 1);
    }
    list[v.size()] = nullptr;
    return list;
}

extern "C" {

agc_t *agc_open(char *fn, int prefetching)
{
    if (!fn)
        return nullptr;
    agc_t *a = new agc_t;
    if (!a->f.Open(fn, prefetching != 0)) {
        delete a;
        return nullptr;
    }
    return a;
}

int agc_close(agc_t *agc)
{
    if (!agc)
        return -1;
    const bool r = agc->f.Close();
    delete agc;
    return r ? 0 : -1;
}

int agc_n_sample(const agc_t *agc) { return agc ? agc->f.NSample() : -1; }

int agc_get_ctg_seq(const agc_t *agc, const char *sample, const char *name, int start, int end, char *buf)
{
    if (!agc || !name || !buf)
        return -1;
    std::string s;
    if (agc->f.GetCtgSeq(sample ? sample : "", name, start, end, s) != 0)
        return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int agc_get_ctg_len(const agc_t *agc, const char *sample, const char *name)
{
    if (!agc || !name)
        return -1;
    return (int)agc->f.GetCtgLen(sample ? sample : "", name);
}

int agc_n_ctg(const agc_t *agc, const char *sample)
{
    if (!agc || !sample)
        return -1;
    return agc->f.NCtg(sample);
}

char *agc_reference_sample(const agc_t *agc)
{
    if (!agc)
        return nullptr;
    std::string s;
    if (agc->f.GetReferenceSample(s) < 0)
        return nullptr;
    char *c = (char *)malloc(s.size() + 1);
    if (c)
        memcpy(c, s.c_str(), s.size() + 1);
    return c;
}

char **agc_list_sample(const agc_t *agc, int *n_sample)
{
    if (!agc || !n_sample)
        return nullptr;
    std::vector<std::string> v;
    agc->f.ListSample(v);
    *n_sample = (int)v.size();
    return vec2list(v);
}

char **agc_list_ctg(const agc_t *agc, const char *sample, int *n_ctg)
{
    if (!agc || !sample || !n_ctg)
        return nullptr;
    std::vector<std::string> v;
    agc->f.ListCtg(sample, v);
    *n_ctg = (int)v.size();
    return vec2list(v);
}

int agc_list_destroy(char **list)
{
    if (!list)
        return 0;
    for (char **q = list; *q; ++q)
        free(*q);
    free(list);
    return 0;
}

int agc_string_destroy(char *sample)
{
    free(sample);
    return 0;
}

char *agc_get_sample_fasta(const agc_t *agc, const char *sample, int line_length, long long *len)
{
    if (!agc || !sample || !len)
        return nullptr;
    std::string s;
    if (!agc->f.GetSampleFasta(sample, s, line_length < 0 ? 0 : (uint32_t)line_length))
        return nullptr;
    char *c = (char *)malloc(s.size() + 1);
    if (!c)
        return nullptr;
    memcpy(c, s.c_str(), s.size() + 1);
    *len = (long long)s.size();
    return c;
}

int agc_get_params(const agc_t *agc, unsigned *k, unsigned *mml, unsigned *pack, unsigned *segment_size)
{
    if (!agc || !k || !mml || !pack || !segment_size)
        return -1;
    uint32_t a, b, c, d;
    if (!agc->f.GetParams(a, b, c, d))
        return -1;
    *k = a;
    *mml = b;
    *pack = c;
    *segment_size = d;
    return 0;
}
}
