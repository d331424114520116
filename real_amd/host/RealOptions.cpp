#include "RealOptions.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <vector>
#include <unistd.h>

bool RealOptions::isFastQ(const std::string &filename)
{
    std::ifstream istr(filename.c_str());
    if (!istr.is_open()) throw std::runtime_error("Unable to open pattern file.");
    int first = istr.get();
    if (first < 0) throw std::runtime_error("Failed to read first character from pattern file.");
    if (first == '>') return false;
    if (first == '@') return true;
    throw std::runtime_error("Unable to determine type of pattern file.");
}

SpoolFile::~SpoolFile()
{
    if (!path.empty()) unlink(path.c_str());
}

void RealOptions::printHelp() const
{
    std::cerr << "Options:\n"
              << "-t <textfilename: a .fa file, or a directory searched for .fa files>\n-p <patternfilename, - = standard input>\n-o <outputfilename, - = standard output>\n"
              << "-s <maximum number of errors in seed, default=2>\n"
              << "-e <total maximum number of errors, default=5>\n"
              << "-l <length of seed, default=32>\n"
              << "-u <search for unique match, default=1>\n"
              << "-f <fraction of device memory to use for the index, default=0.75>\n"
              << "-q <compute scores, default=1>\n-Q <quality offset, default=autodetect>\n"
              << "-filter_level <0..4, default=2>\n-similarity -err -trans -gc -gcmut_bias <scoring parameters>\n"
              << "-device <first HIP device, default=0>\n-gpus <number of devices, default=1>\n"
              << "-index <device|host, where the signature lists are sorted, default=device>\n"
              << "-block <positions per index block, default=as many as fit>\n-batch <reads per device batch>\n"
              << "-gpuparse <parse the read file on the device, default=1>\n-chunk <bytes of read-file text per device call, default=268435456>\n"
              << "Reads longer than 16384 bases are refused (the reference has no such limit); reads longer than 320 bases are slow.\n";
}

// The hand-rolled argv loop of RealOptions.cpp:140-396: "-x value" pairs, unknown arguments are
// reported and skipped, a missing value is an error.
RealOptions::RealOptions(int argc, char *argv[])
{
    std::vector<std::string> o;
    for (int i = 1; i < argc; ++i) o.push_back(argv[i]);
    size_t i = 0;
    auto need = [&](const char *flag) -> const std::string & {
        if (i + 1 >= o.size()) throw std::runtime_error(std::string("Parameter for argument ") + flag + " is missing.");
        return o[i + 1];
    };
    while (i < o.size()) {
        const std::string &a = o[i];
        if (a == "-t") { textfilename = need("-t"); i += 2; }
        else if (a == "-p") { patternfilename = need("-p"); i += 2; }
        else if (a == "-o") { outputfilename = need("-o"); i += 2; }
        else if (a == "-s") { seedkmax = atoi(need("-s").c_str()); i += 2; }
        else if (a == "-e") {
            totalkmax = atoi(need("-e").c_str());
            if (totalkmax > 15) { // UniqueMatchInfoBase::getMaxErrors(), RealOptions.cpp:176-180
                totalkmax = 15;
                std::cerr << "Warning: reducing maximum amount of errors to " << totalkmax << std::endl;
            }
            i += 2;
        }
        else if (a == "-l") { seedl = atoi(need("-l").c_str()); i += 2; }
        else if (a == "-u") { match_unique = atoi(need("-u").c_str()); i += 2; }
        else if (a == "-g") { gaps = atoi(need("-g").c_str()); i += 2; }
        else if (a == "-f" || a == "-m") { fracmem = atof(need("-f").c_str()); i += 2; }
        else if (a == "-q") { scores = atoi(need("-q").c_str()); i += 2; }
        else if (a == "-Q") { qualityOffset = atoi(need("-Q").c_str()); i += 2; }
        else if (a == "-R") { rewritepatterns = atoi(need("-R").c_str()); i += 2; }
        else if (a == "-T") { sort_threads = atoi(need("-T").c_str()); i += 2; }
        else if (a == "-similarity") { similarity = atof(need("-similarity").c_str()); i += 2; }
        else if (a == "-err") { err = atof(need("-err").c_str()); i += 2; }
        else if (a == "-trans") { trans = atof(need("-trans").c_str()); i += 2; }
        else if (a == "-gc") { gc = atof(need("-gc").c_str()); i += 2; }
        else if (a == "-gcmut_bias") { gcmut_bias = atof(need("-gcmut_bias").c_str()); i += 2; }
        else if (a == "-filter_level") { filter_level = atoi(need("-filter_level").c_str()); i += 2; }
        else if (a == "-device") { device = atoi(need("-device").c_str()); i += 2; }
        else if (a == "-gpus") { gpus = atoi(need("-gpus").c_str()); i += 2; }
        else if (a == "-gpus_share_device") { gpus_share_device = atoi(need("-gpus_share_device").c_str()) != 0; i += 2; }
        else if (a == "-index") { host_index = (need("-index") == "host"); i += 2; }
        else if (a == "-block") { block_entries = strtoull(need("-block").c_str(), 0, 10); i += 2; }
        else if (a == "-batch") { batch_reads = strtoull(need("-batch").c_str(), 0, 10); i += 2; }
        else if (a == "-prefix_bits") { prefix_bits = atoi(need("-prefix_bits").c_str()); i += 2; }
        else if (a == "-gpuparse") { gpuparse = atoi(need("-gpuparse").c_str()); i += 2; }
        else if (a == "-chunk") { chunk_bytes = strtoull(need("-chunk").c_str(), 0, 10); i += 2; }
        else if (a == "-table_kind") { table_kind = atoi(need("-table_kind").c_str()); i += 2; }
        else if (a == "-h") { printHelp(); i += 1; }
        else { std::cerr << "Ignoring unknown argument " << a << std::endl; i += 1; }
    }
    if (!(textfilename.size() && patternfilename.size() && outputfilename.size())) printHelp();
    if (!textfilename.size()) throw std::runtime_error("Mandatory argument -t (text file name) is not given.");
    if (!patternfilename.size()) throw std::runtime_error("Mandatory argument -p (pattern file name) is not given.");
    if (!outputfilename.size()) throw std::runtime_error("Mandatory argument -o (output file name) is not given.");
    if (fracmem > 1.0) fracmem = 1.0;
    if (patternfilename == "-") {
        // Reads from standard input (RealOptions.cpp:418-426, real.cpp:240-257).  The reference rewrites them into its
        // temporary pattern file because it reads the patterns once per genome block and once more for the output; so does
        // this driver, through a spool file of the text as it came (the rewritten binary format is out of scope).
        const char *tmp = getenv("TMPDIR");
        std::string name = std::string(tmp && *tmp ? tmp : "/tmp") + "/real_stdin_XXXXXX";
        std::vector<char> path(name.begin(), name.end());
        path.push_back(0);
        const int fd = mkstemp(path.data());
        if (fd < 0) throw std::runtime_error("Unable to create a temporary file for the patterns from standard input.");
        stdin_spool.path = path.data();
        std::cerr << "Reading patterns from stdin, writing them to temporary file " << stdin_spool.path << std::endl;
        std::vector<char> buf((size_t)8 << 20);
        size_t got;
        bool ok = true;
        while (ok && (got = fread(buf.data(), 1, buf.size(), stdin)) > 0)
            for (size_t off = 0; ok && off < got;) {
                const ssize_t w = write(fd, buf.data() + off, got - off);
                if (w <= 0) ok = false; else off += (size_t)w;
            }
        close(fd);
        if (!ok) throw std::runtime_error("Failed to write the patterns from standard input to the temporary file.");
        patternfilename = stdin_spool.path;
        rewritepatterns = true;
    }
    fastq = isFastQ(patternfilename);
    if (fastq && !qualityOffset && !stdin_spool.empty()) {
        // real.cpp:248-257: no detection on standard input, Illumina GA assumed
        std::cerr << "WARNING: automatic quality offset detection not supported when" << std::endl;
        std::cerr << "         reading patterns from standard input. Assuming input" << std::endl;
        std::cerr << "         was produced by an Illumina  GA (i.e. -Q 64)" << std::endl;
        qualityOffset = 64;
    }
    std::cerr << "pattern file is " << (fastq ? "FASTQ" : "FASTA") << std::endl;
    // clamps of RealOptions.cpp:434-453
    if (seedl > 64) { seedl = 64; std::cerr << "reduced seed size to " << seedl << " to not exceed 64." << std::endl; }
    if (seedl % 4) { seedl -= (seedl % 4); std::cerr << "reduced seed size to " << seedl << " to have a multiple of 4." << std::endl; }
    if (seedl < 4) throw std::runtime_error("cannot handle seed length < 4");
    if (seedkmax > 2) { seedkmax = 2; std::cerr << "reduced number of mismatches in seed to " << seedkmax << " as we cannot handle more." << std::endl; }
    if (gaps) std::cerr << "gapped matching is not implemented (nor completed in the reference); ignoring -g" << std::endl;
    // RealOptions.cpp:455-463
    switch (filter_level) {
    case 1: filter_mult = 0.5 * totalkmax; break;
    case 2: filter_mult = 1 * totalkmax; break;
    case 3: filter_mult = 2 * totalkmax; break;
    case 4: filter_mult = 3 * totalkmax; break;
    case 0: default: filter_mult = 0 * totalkmax; break;
    }
    filter_mult /= 70.0;
    std::cerr << "filter_mult=" << filter_mult << std::endl;
    if (gpus < 1) gpus = 1;
    if (chunk_bytes < 4096) chunk_bytes = 4096;
    if (chunk_bytes > (4ull << 30) - (1ull << 20)) chunk_bytes = (4ull << 30) - (1ull << 20); // real_hip_parse_reads: n_bytes < 4 GiB
}
