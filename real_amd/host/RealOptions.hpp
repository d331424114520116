// RealOptions -- the `real` command line of the reference (RealOptions.hpp:26-78, parser
// RealOptions.cpp:122-466), kept flag for flag, plus the device flags of this build.
#pragma once
#include <stdint.h>
#include <string>

// a file that is removed with its owner (also when the owner's constructor throws behind it)
struct SpoolFile {
    std::string path;
    SpoolFile() {}
    ~SpoolFile();
    SpoolFile(const SpoolFile &) = delete;
    SpoolFile &operator=(const SpoolFile &) = delete;
    bool empty() const { return path.empty(); }
};

struct RealOptions {
    // defaults: RealOptions.hpp:27-36
    std::string textfilename, patternfilename, outputfilename;
    unsigned seedkmax = 2;
    unsigned totalkmax = 5;
    int seedl = 32;
    bool match_unique = true;
    double fracmem = 0.75;   // -f / -m: here the fraction of *HBM* the index blocks may use
    bool scores = true;
    unsigned qualityOffset = 0;
    bool rewritepatterns = true; // -R: accepted and ignored (the binary temp format is out of scope)
    unsigned sort_threads = 2;   // -T: threads of the host index sort (only with -index host)
    int filter_level = 2;
    double filter_mult = 0;
    double similarity = 0.995, err = 0.0, trans = 0.71, gc = 0.41, gcmut_bias = 2.0; // Scoring.cpp:204-208
    bool gaps = false;
    bool fastq = false;
    SpoolFile stdin_spool;    // -p -: the file the reads from standard input were written to (removed when the options go)
    // this build
    int device = 0;           // -device: first HIP device
    int gpus = 1;             // -gpus: read batches are dealt round-robin to this many devices
    bool gpus_share_device = false; // -gpus_share_device 1: the contexts of -gpus N all sit on -device (rehearsal of the N-context paths on a one-GPU box)
    bool host_index = false;  // -index host|device: where the six lists are sorted
    uint64_t block_entries = 0; // -block: positions per index block (0 = as many as fit)
    uint64_t batch_reads = 4u << 20; // -batch: reads per device batch
    unsigned prefix_bits = 0;
    unsigned table_kind = 0;         // -table_kind: device bucket tables (real_hip.h), 0 = auto
    unsigned gpuparse = 1;           // -gpuparse: parse the read file on the device when it is in one-line-per-field form
    uint64_t chunk_bytes = 256ull << 20; // -chunk: bytes of read-file text handed to a device at a time (< 4 GiB)

    RealOptions() {}
    RealOptions(int argc, char *argv[]); // throws std::runtime_error like the reference

    void printHelp() const;
    static bool isFastQ(const std::string &filename); // RealOptions.cpp:43-72
    double getFilterValue(unsigned patl) const { return filter_mult * patl; } // RealOptions.hpp:74-77
};
