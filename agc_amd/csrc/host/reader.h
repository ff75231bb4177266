// reader.h -- read side of the .agc v3 archive: the CAGCFile interface of the reference
// (src/lib-cxx/agc-api.h:24-100) on top of a plain host decoder.  Used to verify round trips of the
// archives the create path writes and to offer the lib-cxx / py_agc_api surface; decoding is host code
// (zstd + LZ-diff decode + reverse complement + k-overlap stitching,
// src/common/agc_decompressor_lib.cpp:172-286, src/common/segment.cpp:136-400, src/common/lz_diff.cpp:801-836).
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace agc {

class CAGCFile {
    struct Impl;
    std::unique_ptr<Impl> p;

public:
    CAGCFile();
    ~CAGCFile();
    bool Open(const std::string &file_name, bool prefetching = true);
    bool Close();
    // contig length, or < 0 for errors (-1 unknown, -2 name not unique and sample empty)
    int64_t GetCtgLen(const std::string &sample, const std::string &name) const;
    // [start, end] inclusive as in the reference (start = end = -1: whole contig); returns 0 or < 0
    int GetCtgSeq(const std::string &sample, const std::string &name, int64_t start, int64_t end, std::string &buffer) const;
    int NSample() const;
    int NCtg(const std::string &sample) const;
    int ListSample(std::vector<std::string> &samples) const; // sorted, as the reference
    int ListSampleStored(std::vector<std::string> &samples) const; // archive order (get_samples_list(v, false), used by getcol)
    int GetReferenceSample(std::string &sample) const;
    int ListCtg(const std::string &sample, std::vector<std::string> &names) const;
    // compression parameters stored in the archive: k, min_match_len, pack_cardinality, segment_size
    bool GetParams(uint32_t &k, uint32_t &mml, uint32_t &pack, uint32_t &segment_size) const;
    // key -> value pairs of the file_type_info stream (agc info -v 1)
    bool GetFileTypeInfo(std::vector<std::pair<std::string, std::string>> &info) const;
    // whole sample as FASTA text (agc getset): ">name\n" + 80-column lines
    bool GetSampleFasta(const std::string &sample, std::string &out, uint32_t line_length = 80) const;
    // all contigs of a sample as symbol codes (GetSampleSequences, agc_decompressor_lib.cpp; used by append -a)
    bool GetSampleCodes(const std::string &sample, std::vector<std::string> &names, std::vector<std::vector<uint8_t>> &codes) const;
    // the same, written contig by contig to an open stream (what the CLI uses: no sample-size string is built)
    bool WriteSampleFasta(const std::string &sample, FILE *f, uint32_t line_length = 80) const;
    // one `agc getctg` query -- contig[@sample][:from-to] (agc_decompressor_lib.h:127-130) -- as FASTA text;
    // the header is the full contig name (+ ":from-to" when a range was given), core/agc_decompressor.cpp:478-567
    bool GetContigFasta(const std::string &query, std::string &out, uint32_t line_length, std::string &err) const;
};

} // namespace agc
