// main.cpp -- `agc_amd create`: command-line compatible with `agc create`
// (src/app/main.cpp:76-122, src/app/application.cpp:125-187, application.h:63-71).
// Exit code 0 always, messages on stderr, as the reference.
#include "compressor.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <iterator>
#include <thread>
#include <unordered_set>

namespace {

template <typename T> T clampv(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

void usage()
{
    std::cerr << "agc_amd (MI355X-native create path of AGC v. 3.2)\n"
                 "Usage: agc_amd create [options] <ref.fa> [<in1.fa> ...] > <out.agc>\n"
                 "Options:\n"
                 "   -b <int>       - batch size (default: 50; min: 1; max: 1000000000)\n"
                 "   -c             - concatenated genomes in a single file\n"
                 "   -d             - do not store cmd-line\n"
                 "   -i <file_name> - file with FASTA file names\n"
                 "   -k <int>       - k-mer length (default: 31; min: 17; max: 32)\n"
                 "   -l <int>       - min. match length (default: 20; min: 15; max: 32)\n"
                 "   -o <file_name> - output to file (default: output is sent to stdout)\n"
                 "   -s <int>       - expected segment size (default: 60000; min: 100; max: 1000000)\n"
                 "   -t <int>       - no of threads\n"
                 "   -v <int>       - verbosity level (default: 0; min: 0; max: 2)\n"
                 "   -g <int>       - HIP device ordinal (default: 0)\n";
}

// application.cpp:604-630
void remove_common_suffixes(std::string &s)
{
    const char *suf[] = {".fna", ".gz", ".fa", ".fasta"};
    for (;;) {
        bool removed = false;
        for (const char *x : suf) {
            size_t l = strlen(x);
            if (s.size() <= l)
                continue;
            if (s.compare(s.size() - l, l, x) == 0) {
                s.resize(s.size() - l);
                removed = true;
                break;
            }
        }
        if (!removed)
            break;
    }
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 2 || std::string(argv[1]) != "create") {
        usage();
        return 0;
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    uint32_t k = 31, pack = 50, seg = 60000, mml = 20, threads = std::max(1u, hw / 2), verbosity = 0;
    int device = 0;
    bool concat = false, adaptive = false;
    double ff = 0;
    std::string out = "-";
    std::vector<std::string> inputs;
    int i = 2;
    for (; i < argc; ++i) {
        std::string a = argv[i];
        if (a.size() < 2 || a[0] != '-')
            break;
        auto val = [&]() -> const char * {
            if (a.size() > 2)
                return argv[i] + 2;
            return i + 1 < argc ? argv[++i] : "";
        };
        switch (a[1]) {
        case 't': threads = clampv<uint32_t>((uint32_t)atoi(val()), 1, std::max(16u, hw)); break;
        case 'b': pack = clampv<uint32_t>((uint32_t)atoi(val()), 1, 1000000000u); break;
        case 's': seg = clampv<uint32_t>((uint32_t)atoi(val()), 100, 1000000u); break;
        case 'k': k = clampv<uint32_t>((uint32_t)atoi(val()), 17, 32); break;
        case 'l': mml = clampv<uint32_t>((uint32_t)atoi(val()), 15, 32); break;
        case 'f': ff = clampv<double>(atof(val()), 0, 0.05); break;
        case 'v': verbosity = clampv<uint32_t>((uint32_t)atoi(val()), 0, 2); break;
        case 'g': device = atoi(val()); break;
        case 'a': adaptive = true; break;
        case 'c': concat = true; break;
        case 'd': break;
        case 'o': out = val(); break;
        case 'i': {
            std::ifstream inf(val());
            if (!inf) {
                std::cerr << "Cannot open file with FASTA names\n";
                return 0;
            }
            inputs.assign(std::istream_iterator<std::string>(inf), std::istream_iterator<std::string>());
            break;
        }
        default: break;
        }
    }
    if (i >= argc) {
        std::cerr << "No reference file name\n";
        return 0;
    }
    inputs.insert(inputs.begin(), argv[i]);
    for (++i; i < argc; ++i)
        inputs.emplace_back(argv[i]);
    { // sanitize_input_file_names, application.cpp:584-601
        std::vector<std::string> v;
        std::unordered_set<std::string> seen;
        for (auto &s : inputs)
            if (seen.insert(s).second)
                v.push_back(s);
        inputs.swap(v);
    }
    agc::CAGCCompressor c;
    c.SetDevice(device);
    if (!c.Create(out, pack, k, inputs.front(), seg, mml, concat, adaptive, verbosity, threads, ff)) {
        std::cerr << "Cannot create archive " << out << std::endl;
        return 0;
    }
    std::vector<std::pair<std::string, std::string>> v;
    for (auto &fn : inputs) {
        std::string sn = std::filesystem::path(fn).stem().string();
        remove_common_suffixes(sn);
        v.emplace_back(sn, fn);
    }
    bool r = c.AddSampleFiles(v, threads);
    r &= c.Close(threads);
    if (verbosity > 0) {
        const auto &s = c.Stats();
        std::cerr << "bases " << s.bases << " segments " << s.segments << " groups " << s.new_groups << " one-splitter " << s.one_splitter
                  << " middle " << s.middle_tried << "/" << s.middle_split << " zstd " << c.ZstdVersion() << "\n"
                  << "seconds: io " << s.t_io << " scan " << s.t_scan << " classify " << s.t_classify << " gpu-aux " << s.t_gpu_aux
                  << " register " << s.t_register << " encode " << s.t_encode << " store " << s.t_store << " zstd " << s.t_zstd << "\n";
    }
    return 0;
}
