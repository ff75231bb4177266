// main.cpp -- `agc_amd create`: command-line compatible with `agc create`
// (src/app/main.cpp:76-122, src/app/application.cpp:125-187, application.h:63-71), plus the read-side
// commands getcol / getset / getctg / listref / listset / listctg (src/app/main.cpp:171-368) on the host decoder.
// Exit code 0 always, messages on stderr, as the reference.
#include "compressor.h"
#include "../../../include/agc_hip.h"
#include "reader.h"
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
#include <unistd.h>

namespace {

template <typename T> T clampv(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

void usage()
{
    std::cerr << "agc_amd (MI355X-native create path of AGC v. 3.2)\n"
                 "Usage: agc_amd create [options] <ref.fa> [<in1.fa> ...] > <out.agc>\n"
                 "       agc_amd append [-a] [-c] [-i <file>] [-o <file>] [-t <int>] [-v <int>] <in.agc> [<in1.fa> ...] > <out.agc>\n"
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
                 "   -g <int>       - HIP device ordinal (default: 0)\n"
                 "       agc_amd getcol  [-l <line>] [-o <dir>] [-r] <in.agc>\n"
                 "       agc_amd getset  [-l <line>] [-o <file>] <in.agc> <sample> [<sample> ...]\n"
                 "       agc_amd getctg  [-l <line>] [-o <file>] <in.agc> <contig[@sample][:from-to]> ...\n"
                 "       agc_amd listref|listset [-o <file>] <in.agc>\n"
                 "       agc_amd info [-v 1] <in.agc>\n"
                 "       agc_amd listctg [-o <file>] <in.agc> <sample> [<sample> ...]\n";
}

bool write_out(const std::string &name, const std::string &data, bool append = false)
{
    if (name.empty()) {
        fwrite(data.data(), 1, data.size(), stdout);
        return true;
    }
    FILE *f = fopen(name.c_str(), append ? "ab" : "wb");
    if (!f) {
        std::cerr << "Cannot open output file " << name << std::endl;
        return false;
    }
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    if (fclose(f) != 0 || !ok) {
        std::cerr << "Cannot write output file " << name << std::endl;
        return false;
    }
    return true;
}

// read-side commands; option letters as in application.cpp:190-583 (-g gzip, -p, -s, -t, -v are accepted and ignored)
int read_command(const std::string &mode, int argc, char **argv)
{
    uint32_t line_length = 80;
    std::string out;
    bool no_ref = false, verbose = false;
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
        case 'l': line_length = clampv<uint32_t>((uint32_t)atoi(val()), 40, 2000000000u); break;
        case 'o': out = val(); break;
        case 'g':
        case 't': (void)val(); break;
        case 'v': verbose = atoi(val()) > 0; break;
        case 'r': no_ref = true; break;
        default: break;
        }
    }
    if (i >= argc) {
        std::cerr << "No archive name\n";
        return 0;
    }
    agc::CAGCFile f;
    if (!f.Open(argv[i])) {
        std::cerr << "Cannot open archive " << argv[i] << std::endl;
        return 0;
    }
    std::vector<std::string> args(argv + i + 1, argv + argc);
    std::string txt;
    if (mode == "listref") {
        f.GetReferenceSample(txt);
        write_out(out, txt);
    } else if (mode == "listset") {
        std::vector<std::string> v;
        f.ListSample(v);
        for (auto &s : v)
            txt += s + "\n";
        write_out(out, txt);
    } else if (mode == "listctg") {
        if (args.empty())
            std::cerr << "No sample name\n";
        for (auto &sn : args) {
            txt += sn + "\n";
            std::vector<std::string> v;
            f.ListCtg(sn, v);
            for (auto &c : v)
                txt += "   " + c + "\n";
        }
        write_out(out, txt);
    } else if (mode == "getset") {
        if (args.empty())
            std::cerr << "No sample name\n";
        FILE *fo = out.empty() ? stdout : fopen(out.c_str(), "wb");
        if (!fo) {
            std::cerr << "Cannot open output file " << out << std::endl;
            return 0;
        }
        for (auto &sn : args)
            if (!f.WriteSampleFasta(sn, fo, line_length)) {
                std::cerr << "There is no sample " << sn << std::endl;
                break;
            }
        if (fo != stdout)
            fclose(fo);
        else
            fflush(stdout);
    } else if (mode == "getctg") {
        if (args.empty())
            std::cerr << "No contig name\n";
        for (auto &q : args) {
            std::string err;
            if (!f.GetContigFasta(q, txt, line_length, err)) {
                std::cerr << err << std::endl;
                return 0;
            }
        }
        write_out(out, txt);
    } else if (mode == "info") { // src/app/main.cpp:373-420: everything goes to stderr; v3 archives store no command lines
        std::vector<std::string> v;
        f.ListSample(v);
        uint32_t k, mml, pack, seg;
        f.GetParams(k, mml, pack, seg);
        std::string ref;
        f.GetReferenceSample(ref);
        std::cerr << "No. samples      : " << v.size() << std::endl;
        std::cerr << "k-mer length     : " << k << std::endl;
        std::cerr << "Min. match length: " << mml << std::endl;
        if (seg)
            std::cerr << "Segment size     : " << seg << std::endl;
        std::cerr << "Batch size       : " << pack << std::endl;
        std::cerr << "Reference name   : " << ref << std::endl;
        std::cerr << "Command lines:" << std::endl;
        if (verbose) {
            std::vector<std::pair<std::string, std::string>> info;
            f.GetFileTypeInfo(info);
            std::cerr << "File type info:\n";
            for (auto &x : info)
                std::cerr << "  " << x.first << " : " << x.second << std::endl;
        }
    } else if (mode == "getcol") {
        if (!out.empty() && !std::filesystem::is_directory(out)) {
            std::cerr << "Path must point to an existing directory\n";
            return 0;
        }
        std::string ref;
        f.GetReferenceSample(ref);
        std::vector<std::string> v;
        f.ListSampleStored(v);
        for (size_t j = no_ref ? 1 : 0; j < v.size(); ++j) {
            FILE *fo = out.empty() ? stdout : fopen((std::filesystem::path(out) / (v[j] + ".fa")).string().c_str(), "wb");
            if (!fo || !f.WriteSampleFasta(v[j], fo, line_length))
                return 0;
            if (fo != stdout)
                fclose(fo);
        }
        fflush(stdout);
    }
    f.Close();
    return 0;
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
    agc::StartLap("main");
    if (argc >= 2) {
        const std::string mode = argv[1];
        for (const char *m : {"getcol", "getset", "getctg", "listref", "listset", "listctg", "info"})
            if (mode == m)
                return read_command(mode, argc, argv);
    }
    const bool append = argc >= 2 && std::string(argv[1]) == "append";
    if (argc < 2 || (std::string(argv[1]) != "create" && !append)) {
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
        std::cerr << (append ? "No archive name\n" : "No reference file name\n");
        return 0;
    }
    const std::string in_archive = append ? argv[i] : "";
    if (!append)
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
    if (append) { // src/app/main.cpp:125-168
        if (!c.Append(in_archive, out, verbosity, true, concat, adaptive, threads, ff)) {
            std::cerr << "Cannot open archive " << in_archive << " or create archive " << out << std::endl;
            return 0;
        }
    } else if (!c.Create(out, pack, k, inputs.front(), seg, mml, concat, adaptive, verbosity, threads, ff)) {
        std::cerr << "Cannot create archive " << out << std::endl;
        return 0;
    }
    // AGC_AMD_KERNEL_TIMES=1 (a measuring aid, `bench.py --config`): HIP events around every kernel of the run, summed per
    // kernel family on stderr with the symbols each was asked to look at -- the rows of a roofline table
    const bool kernel_times = getenv("AGC_AMD_KERNEL_TIMES") != nullptr;
    if (kernel_times && c.HipContext())
        agc_hip_timing_enable(c.HipContext(), 1);
    std::vector<std::pair<std::string, std::string>> v;
    for (auto &fn : inputs) {
        std::string sn = std::filesystem::path(fn).stem().string();
        remove_common_suffixes(sn);
        v.emplace_back(sn, fn);
    }
    agc::StartLap("Create done (reference read, splitters)");
    bool r = c.AddSampleFiles(v, threads);
    agc::StartLap("AddSampleFiles done");
    r &= c.Close(threads);
    agc::StartLap("Close done");
    if (verbosity > 0) {
        const auto &s = c.Stats();
        std::cerr << "bases " << s.bases << " segments " << s.segments << " groups " << s.new_groups << " one-splitter " << s.one_splitter
                  << " middle " << s.middle_tried << "/" << s.middle_split << " windows " << s.windows << " commit-runs " << s.commit_runs
                  << " revalidated " << s.revalidated << " windows-cut " << s.windows_cut << " zstd " << c.ZstdVersion() << "\n"
                  << "seconds: io " << s.t_io << " scan " << s.t_scan << " classify " << s.t_classify << " gpu-aux " << s.t_gpu_aux
                  << " register " << s.t_register << " encode " << s.t_encode << " store " << s.t_store << " zstd " << s.t_zstd << " (inside the device library: " << s.t_device << ")\n"
                  << "host-only part: scan " << s.h_scan << " classify " << s.h_classify << " gpu-aux " << s.h_gpu_aux << " register " << s.h_register
                  << " encode " << s.h_encode << " store " << s.h_store << "\n"
                  << "entropy-seconds: host-pool " << s.t_zstd_host << " device " << s.t_zstd_dev << " staging " << s.t_zstd_stage << " caller-waited "
                  << s.t_zstd_wait << "\n";
    }
    if (kernel_times && c.HipContext()) {
        static const char *names[AGC_HIP_K_COUNT] = {"scan", "index", "encode", "estimate", "costvec", "revcomp", "preprocess", "refstore", "zstd", "filter",
                                                     "segments", "pack"};
        const auto &s = c.Stats();
        std::cerr << "kernel-ms:";
        for (int w = 0; w < AGC_HIP_K_COUNT; ++w) {
            double ms = 0;
            uint64_t n = 0;
            if (agc_hip_timing_get(c.HipContext(), w, &ms, &n) == AGC_HIP_OK && n)
                std::cerr << " " << names[w] << " " << ms << " " << n;
        }
        std::cerr << "\nkernel-symbols: scan " << s.bases << " encode " << s.enc_text + s.enc_ref << " estimate " << s.est_text + s.est_ref << " costvec "
                  << s.cv_text + s.cv_ref << " filter " << s.est_text + s.cv_text << " zstd_dev_in " << s.zstd_dev_in << " zstd_dev_out " << s.zstd_dev_out
                  << " zstd_in " << s.zstd_in << "\n";
    }
    // The archive is closed and on disk (Close -> ArchiveWriter::close: fclose).  What is left is giving back memory, joining
    // forty threads and tearing the HIP runtime down -- 60-100 ms of a 0.25 s run on a one-genome collection (scripts/start_cost.py):
    // the operating system does all of that for a process that simply ends.
    agc::StartLap("exit");
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    _exit(0);
}
