// mapper_cli.cpp -- command line driver: `alfalfa_amd index ...` / `alfalfa_amd align ...`
// Same commands, flags and defaults as the reference (src/options.h:186-265, src/options.cpp:43-92,187-407) and
// the same batch / ordered-output structure as its driver (src/main.cpp:64-149,151-264), with the per-read
// kt_for loop replaced by worker threads that each run a fiber scheduler (hot_sched.h) feeding libalfa_hot.
#include "mapper_cli.h"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../seqio.h"

namespace alfa {

namespace {

struct Options {
    std::string command;
    int K = 12, threads = 1;
    std::string unpairedQ, pair1, pair2, ref_fasta, index_prefix, indexLocation, outputName;
    bool saveIndex = false, hasChild = true, hasSuflink = false, hasKmer = true;
    AlignOpt aln;
    PairedOpt paired;
    int verbose = 0;
    // extensions of this implementation
    int fibers = 2048, gpus = 1, device = 0;     // 2048 fibers per worker: measured best on C3 (4096: 500 k, 2048: 667 k, 1024: 650 k, 512: 589 k reads/s)
    bool stats = false;
    bool gpuIndex = false;      // --gpu-index: build the index arrays on the GPU (index_build.hip)
    int mapBatch = 16384;       // --map-batch: reads per worker and device call when whole reads are mapped on the device.  With reads
                                // rendered straight from the call's results nothing per read stays behind on the host, and a call's fixed
                                // part (a dozen launches, three host waits) is what counts: 2 000 000 short reads, map phase with 8 / 16
                                // workers: 4096: - / 0.28 s, 8192: 0.29 / 0.27, 16384: 0.30 / 0.27, 32768: 0.31 / 0.32, 65536: 0.35 / 0.38
};

int seed_type(const std::string& t) {       // parseSeedType src/options.cpp:155-172 (CLI names are shifted)
    if (t == "MEM") return SEED_MEM;
    if (t == "SMEM" || t == "SMAM") return SEED_SMAM;
    if (t == "PSMEM") return SEED_SMEM;
    if (t == "MAM" || t == "MUM") { fprintf(stderr, "ERROR: seed type %s is not supported\n", t.c_str()); exit(1); }
    fprintf(stderr, "WARNING: unrecognized seed type. SMEMs are used instead.\n");
    return SEED_SMAM;
}

bool parse(int argc, char** argv, Options& o) {
    if (argc < 2) return false;
    o.command = argv[1];
    if (o.command != "index" && o.command != "align") return false;
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) { fprintf(stderr, "ERROR: option %s needs an argument\n", argv[i]); exit(1); }
        return argv[++i];
    };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto is = [&](const char* s, const char* l) { return (s && a == std::string("-") + s) || (l && a == std::string("--") + l); };
        if (is("r", "reference")) o.ref_fasta = need(i);
        else if (is("s", "sparseness")) o.K = atoi(need(i));
        else if (is("p", "prefix")) o.index_prefix = need(i);
        else if (is(nullptr, "no-child")) o.hasChild = false;
        else if (is(nullptr, "suflink")) o.hasSuflink = true;
        else if (is(nullptr, "no-kmer")) o.hasKmer = false;
        else if (is("i", "index")) o.indexLocation = need(i);
        else if (is(nullptr, "save")) o.saveIndex = true;
        else if (is("0", "single")) o.unpairedQ = need(i);
        else if (is("1", "mates1")) o.pair1 = need(i);
        else if (is("2", "mates2")) o.pair2 = need(i);
        else if (is("o", "output")) o.outputName = need(i);
        else if (is("a", "alignments")) o.aln.alignmentCount = atoi(need(i));
        else if (is(nullptr, "no-forward")) o.aln.noFW = true;
        else if (is(nullptr, "no-reverse")) o.aln.noRC = true;
        else if (is("e", "edit-distance")) o.aln.errorPercent = atof(need(i)) + 0.01;     // src/options.cpp:296
        else if (is(nullptr, "no-rescue")) o.aln.rescue = false;
        else if (is("t", "threads")) o.threads = atoi(need(i));
        else if (is(nullptr, "seed")) o.aln.memType = seed_type(need(i));
        else if (is("l", "min-length")) { o.aln.minMemLength = atoi(need(i)); o.aln.fixedMinLength = true; }
        else if (is("m", "max-seeds")) o.aln.maxSeedCandidates = atoi(need(i));
        else if (is(nullptr, "max-smems")) o.aln.maxSmemMemStart = atoi(need(i));
        else if (is(nullptr, "max-mems")) o.aln.maxRegionMems = atoi(need(i));
        else if (is(nullptr, "min-mem-length")) o.aln.minMemRegLength = atoi(need(i));
        else if (is(nullptr, "no-sparseness")) o.aln.noSparseQ = true;
        else if (is("f", "max-failures")) o.aln.maxTrial = atoi(need(i));
        else if (is(nullptr, "reset-failures")) o.aln.maxTrialBest = false;
        else if (is(nullptr, "max-alignments")) o.aln.minTrial = atoi(need(i));              // writes minTrial (:323)
        else if (is("c", "min-coverage")) o.aln.minCoverage = atof(need(i));
        else if (is(nullptr, "skip-unique")) { o.aln.alwaysUnique = false; need(i); }        // declared Arg::Required
        else if (is(nullptr, "full-dp")) o.aln.fullDP = true;
        else if (is(nullptr, "local")) o.aln.global = false;
        else if (is("b", "bandwidth")) o.aln.fixedBandSize = atoi(need(i));
        else if (is("M", "match")) o.aln.match = atoi(need(i));
        else if (is("U", "mismatch")) o.aln.mismatch = atoi(need(i));
        else if (is("O", "gap-open")) o.aln.openGap = atoi(need(i));
        else if (is("E", "gap-extend")) o.aln.extendGap = atoi(need(i));
        else if (is("I", "min-insert")) o.paired.minInsert = atoi(need(i));
        else if (is("X", "max-insert")) o.paired.maxInsert = atoi(need(i));
        else if (is(nullptr, "orientation")) {
            std::string t = need(i);
            o.paired.orientation = t == "rf" ? PAIR_RF : t == "ff" ? PAIR_FF : PAIR_FR;
        }
        else if (is(nullptr, "no-mixed")) o.paired.mixed = false;
        else if (is(nullptr, "no-discordant")) o.paired.discordant = false;
        else if (is(nullptr, "dovetail")) o.paired.dovetail = true;
        else if (is(nullptr, "no-contain")) o.paired.contain = false;
        else if (is(nullptr, "no-overlap")) o.paired.overlap = false;
        else if (is(nullptr, "paired-mode")) o.paired.mode = atoi(need(i));
        else if (is(nullptr, "paired-rescue")) o.paired.pairedRescue = true;
        else if (is("v", "verbose")) o.verbose = atoi(need(i));
        else if (is(nullptr, "fibers")) o.fibers = atoi(need(i));
        else if (is(nullptr, "gpus")) o.gpus = atoi(need(i));
        else if (is(nullptr, "device")) o.device = atoi(need(i));
        else if (is(nullptr, "stats")) o.stats = true;
        else if (is(nullptr, "gpu-index")) o.gpuIndex = true;
        else if (is(nullptr, "map-batch")) o.mapBatch = std::max(64, atoi(need(i)));
        else if (is(nullptr, "device-regions")) o.aln.deviceRegions = true;
        else if (is(nullptr, "core-trials")) o.aln.coreTrials = true;
        else if (is(nullptr, "host-trials")) o.aln.deviceMap = false;
        else { fprintf(stderr, "ERROR: unknown option %s\n", a.c_str()); exit(1); }
    }
    if (o.command == "index") {
        if (o.ref_fasta.empty()) { fprintf(stderr, "ERROR: no reference file given.\n"); exit(1); }
        if (o.index_prefix.empty()) o.index_prefix = o.ref_fasta;
        return true;
    }
    if (o.ref_fasta.empty() && o.indexLocation.empty()) { fprintf(stderr, "ERROR: no reference or index name given.\n"); exit(1); }
    if (o.indexLocation.empty() && o.index_prefix.empty()) o.index_prefix = o.ref_fasta;
    if (o.unpairedQ.empty() && (o.pair1.empty() || o.pair2.empty())) {
        fprintf(stderr, "ERROR: ALFALFA requires query files. Specify query files with -0 or -1,-2 options.\n");
        exit(1);
    }
    if (o.outputName.empty()) o.outputName = o.ref_fasta + ".sam";
    if (o.aln.errorPercent < 0.0 || o.aln.errorPercent > 1.0) { fprintf(stderr, "ERROR: uncorrect value for edit distance.\n"); exit(1); }
    if (o.aln.minCoverage < 0.0 || o.aln.minCoverage > 1.0) { fprintf(stderr, "ERROR: uncorrect value for minimum coverage.\n"); exit(1); }
    if (o.paired.minInsert > o.paired.maxInsert) { fprintf(stderr, "ERROR: minimum insert size is larger than maximum insert size\n"); exit(1); }
    if (o.paired.mode < 1 || o.paired.mode > 6) {                            // src/options.cpp:396-397
        fprintf(stderr, "ERROR: chosen paired-end mode not supported.\n");
        exit(1);
    }
    o.aln.maxAlnCount = std::max(o.aln.alignmentCount, o.aln.minTrial);      // src/options.cpp:400
    o.aln.fill_m();
    return true;
}

// fastqInputReader::nextRead src/fasta.cpp:135-159 + the '/1' '/2' stripping of src/main.cpp:91-92
bool next_read(SeqReader& rd, Read& r) {
    static thread_local SeqRecord rec;       // buffers reused from record to record
    if (rd.next(rec) < 0) return false;
    r.qname = rec.name;
    r.sequence = rec.seq;
    for (auto& c : r.sequence) c = (char)std::tolower((unsigned char)c);
    if (!rec.qual.empty()) r.qual = rec.qual;
    else r.qual = "*";
    if (r.qname.length() > 2 && r.qname[r.qname.length() - 2] == '/') r.qname.erase(r.qname.length() - 2);
    return true;
}

// ---- parallel FASTQ parse (fastqInputReader::nextRead src/fasta.cpp:135-159 + src/main.cpp:64-92)
// The reader thread only SPLITS records: four memchr calls per record over the memory-mapped file, or -- gzip input -- over
// the chunk it has just inflated.  Everything the reference's reader does per record -- name up to the first white space, lower-casing, '/1' '/2'
// stripping, the strings themselves -- happens on the worker that maps the read (materialize()).  The split takes the fast
// path only for records it can prove kseq would parse the same way: four lines, '@' first, a non-empty sequence line that
// does not start with '>', '+' or '@', '+' on the third, a quality line exactly as long as the sequence and no '\r' anywhere.
// The first record that fails a check sends the rest of the file through SeqReader (the kseq restatement), on the reader
// thread as before.
struct RawRec {
    const char* p = nullptr;      // '@' of the record; nullptr: the Read was filled by the general parser
    uint32_t l1 = 0, l2 = 0;      // lengths of the header line (with '@') and of the sequence (= quality) line
    uint32_t o2 = 0, o4 = 0;      // offsets of the sequence and quality lines from p
};

class ReadSource {
   public:
    typedef std::shared_ptr<std::vector<char>> Chunk;
    ~ReadSource() {
        if (map_ && map_ != MAP_FAILED) munmap((void*)map_, size_);
        if (fd_ >= 0) ::close(fd_);
        if (gz_) gzclose(gz_);
    }
    bool open(const char* path) {
        static const bool no_fast = getenv("ALFA_NO_FAST_FASTQ") != nullptr;
        fd_ = ::open(path, O_RDONLY);
        struct stat st;
        if (!no_fast && fd_ >= 0 && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 2) {
            size_ = (size_t)st.st_size;
            map_ = (const char*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (map_ != MAP_FAILED && !((unsigned char)map_[0] == 0x1f && (unsigned char)map_[1] == 0x8b)) {
                (void)madvise((void*)map_, size_, MADV_SEQUENTIAL);
                fast_ = true;
                base_ = map_; end_ = size_; final_ = true;
                return true;
            }
            const bool gz = map_ != MAP_FAILED;
            if (map_ != MAP_FAILED) munmap((void*)map_, size_);
            map_ = nullptr;
            if (gz) {
                // gzip: this thread inflates into chunks and splits them; the workers still build the reads.  A chunk stays
                // alive as long as a batch holds records in it (chunk()).
                gz_ = gzopen(path, "r");
                if (gz_) {
                    (void)gzbuffer(gz_, 1 << 20);
                    fast_ = true; chunked_ = true;
                    base_ = nullptr; end_ = 0; pos_ = 0; final_ = false;
                    return true;
                }
            }
        }
        return slow_.open(path);
    }
    // the memory the last record returned by next() lies in (gzip input); null for a memory-mapped file
    const Chunk& chunk() const { return cur_; }
    // 0: end of input; 1: x describes a record (r untouched); 2: r filled by the general parser (x.p == nullptr)
    // (get() yields the Read to fill; it is only called on the general parser's path)
    template <class GetRead>
    int next(GetRead&& get, RawRec& x) {
        while (fast_) {
            if (pos_ >= end_ && final_) return 0;
            if (pos_ < end_ && split(x)) return 1;
            if (chunked_ && !final_) { refill(false); continue; }       // the record may simply be cut off: more input first
            // not provably kseq-equivalent from here on: the general parser takes over at this byte
            if (chunked_) refill(true);
            fast_ = false;
            slow_.open_mem(base_ + pos_, end_ - pos_);
        }
        x.p = nullptr;
        return next_read(slow_, get()) ? 2 : 0;
    }
   private:
    // a new chunk: the unconsumed tail of the current one, then inflated input until the chunk is full (all: until the end)
    void refill(bool all) {
        const size_t tail = end_ - pos_;
        size_t cap = std::max<size_t>(CHUNK, 2 * tail + (1 << 16));
        Chunk nc = std::make_shared<std::vector<char>>();
        nc->resize(cap);
        if (tail) memcpy(nc->data(), base_ + pos_, tail);
        size_t have = tail;
        for (;;) {
            if (have == cap) {
                if (!all) break;
                cap *= 2;
                nc->resize(cap);
            }
            const int got = gzread(gz_, nc->data() + have, (unsigned)std::min<size_t>(cap - have, 1u << 30));
            if (got <= 0) { final_ = true; break; }
            have += (size_t)got;
        }
        cur_ = nc;
        base_ = cur_->data(); end_ = have; pos_ = 0;
    }
    bool split(RawRec& x) {
        const char* b = base_ + pos_;
        const char* e = base_ + end_;
        if (*b != '@') return false;
        const char* n1 = (const char*)memchr(b, '\n', (size_t)(e - b));
        if (!n1) return false;
        const char* s2 = n1 + 1;
        const char* n2 = s2 < e ? (const char*)memchr(s2, '\n', (size_t)(e - s2)) : nullptr;
        if (!n2 || n2 == s2 || *s2 == '>' || *s2 == '+' || *s2 == '@') return false;
        const char* s3 = n2 + 1;
        if (s3 >= e || *s3 != '+') return false;
        const char* n3 = (const char*)memchr(s3, '\n', (size_t)(e - s3));
        if (!n3) return false;
        const char* s4 = n3 + 1;
        const size_t l2 = (size_t)(n2 - s2);
        if ((size_t)(e - s4) < l2) return false;
        const char* n4 = s4 + l2;                       // the quality line must end exactly here
        if (n4 < e && *n4 != '\n') return false;
        if (n4 == e && !final_) return false;           // ... which only the end of the INPUT settles, not the end of a chunk
        if (memchr(s4, '\n', l2)) return false;         // ... and not before
        const size_t reclen = (size_t)(n4 - b);
        if (memchr(b, '\r', reclen) || reclen >= (1ull << 31)) return false;
        x.p = b; x.l1 = (uint32_t)(n1 - b); x.l2 = (uint32_t)l2; x.o2 = (uint32_t)(s2 - b); x.o4 = (uint32_t)(s4 - b);
        pos_ = (size_t)(n4 - base_) + (n4 < e ? 1 : 0);
        return true;
    }
    static constexpr size_t CHUNK = 32u << 20;
    int fd_ = -1;
    const char* map_ = nullptr;
    size_t size_ = 0;
    const char* base_ = nullptr;      // the bytes being split: the mapping, or the current chunk
    size_t end_ = 0, pos_ = 0;
    bool fast_ = false, chunked_ = false, final_ = false;
    gzFile gz_ = nullptr;
    Chunk cur_;
    SeqReader slow_;
};

// what next_read() does with a record, on the worker: same fields, same bytes
void materialize(const RawRec& x, Read& r) {
    if (!x.p) return;
    const char* h = x.p + 1;
    const size_t hl = x.l1 - 1;
    size_t k = 0;
    while (k < hl && !isspace((unsigned char)h[k])) k++;
    r.qname.assign(h, k);
    r.sequence.assign(x.p + x.o2, x.l2);
    for (auto& c : r.sequence) c = (char)(c + ((unsigned char)(c - 'A') < 26u ? 32 : 0));      // tolower in the "C" locale, branch-free
    r.qual.assign(x.p + x.o4, x.l2);
    if (r.qname.length() > 2 && r.qname[r.qname.length() - 2] == '/') r.qname.erase(r.qname.length() - 2);
}

// the same record as pointers (nothing copied): the name as materialize() would store it
RawRead raw_read(const RawRec& x) {
    const char* h = x.p + 1;
    const size_t hl = x.l1 - 1;
    size_t k = 0;
    while (k < hl && !isspace((unsigned char)h[k])) k++;
    if (k > 2 && h[k - 2] == '/') k -= 2;
    return RawRead{h, (int)k, x.p + x.o2, x.p + x.o4, (int)x.l2};
}

// persistent worker pool: every batch is split into contiguous slices, one per worker
class Pool {
   public:
    Pool(int n, std::function<void(int)> init) : n_(n) {
        for (int t = 0; t < n; ++t) th_.emplace_back([this, t, init] { init(t); loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(const std::function<void(int)>& fn) {
        std::unique_lock<std::mutex> g(mu_);
        fn_ = &fn; pending_ = n_; gen_++;
        cv_.notify_all();
        done_.wait(g, [this] { return pending_ == 0; });
    }
   private:
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            (*fn)(t);
            { std::lock_guard<std::mutex> g(mu_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// bounded hand-off queue between the pipeline stages
template <class T>
class Chan {
   public:
    explicit Chan(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        cv_.notify_all();
    }
    bool try_push(T& v) {                   // false (v untouched) when the queue is full
        std::lock_guard<std::mutex> g(mu_);
        if (q_.size() >= cap_) return false;
        q_.push_back(std::move(v));
        cv_.notify_all();
        return true;
    }
    bool try_pop(T& v) {
        std::lock_guard<std::mutex> g(mu_);
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    bool pop(T& v) {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> g(mu_);
        closed_ = true;
        cv_.notify_all();
    }
   private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

}  // namespace

int mapper_main(int argc, char** argv, const HotBinding& bind) {
    Options o;
    if (!parse(argc, argv, o)) {
        fprintf(stderr, "usage: %s index -r ref.fa [-s K] [-p prefix] [--no-child] [--suflink] [--no-kmer] [--gpu-index [--device D]]\n"
                        "       %s align (-r ref.fa | -i index) (-0 reads.fq | -1 m1.fq -2 m2.fq) [-o out.sam] [reference options]\n",
                argv[0], argv[0]);
        return 1;
    }
    alfa_host_index* hidx = nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    if (o.gpuIndex) {
        if (!bind.index_build_on_device) { fprintf(stderr, "ERROR: --gpu-index: this build has no GPU index builder\n"); return 1; }
        if (bind.index_build_on_device(o.device)) { fprintf(stderr, "ERROR: %s\n", bind.last_error()); return 1; }
    }
    if (o.command == "index") {
        if (alfa_host_index_build_fasta(o.ref_fasta.c_str(), o.K, o.hasSuflink, o.hasChild, o.hasKmer, &hidx)) {
            fprintf(stderr, "ERROR: %s\n", alfa_last_error());
            return 1;
        }
        // the reference only saves with --save (src/main.cpp:321); an index command that saves nothing is useless, so we
        // always save here -- the files are byte-identical to the reference's
        if (alfa_host_index_save(hidx, o.index_prefix.c_str())) { fprintf(stderr, "ERROR: %s\n", alfa_last_error()); return 1; }
        alfa_host_index_free(hidx);
        return 0;
    }
    const int T = std::max(1, o.threads);
    // -t counts the threads the tool keeps busy: beside the workers there are the reader and the writer thread, each busy for
    // most of a run of short reads (the reference's -t workers stand still while its main thread reads and writes).  From
    // eight threads on, two of them are those: at most T - 2 workers build reads / render SAM at a time (device calls for a
    // group of long reads, which wait for the GPU, are not counted).  On a box that gives the process exactly -t cores,
    // sixteen busy workers plus the two were slower than fourteen plus the two: 4 000 000 short reads 0.62-0.65 s against
    // 0.53-0.55 s (ALFA_WORKERS sets the number)
    const int busyWorkers = getenv("ALFA_WORKERS") ? std::max(1, atoi(getenv("ALFA_WORKERS"))) : (T >= 8 ? T - 2 : T);
    const int G = std::max(1, o.gpus);
    std::vector<HotOps> ops((size_t)G);
    // A saved index goes straight from its files to the GPUs when the binding can do that (the arrays never pass through host
    // vectors; the host keeps the text, the names and the offsets): ALFA_HOST_INDEX_LOAD=1 keeps the two-step path
    bool ops_ready = false;
    if (!o.indexLocation.empty() && bind.load_ops && !o.saveIndex && getenv("ALFA_HOST_INDEX_LOAD") == nullptr) {
        ops_ready = true;
        for (int g = 0; g < G && ops_ready; ++g)
            if (bind.load_ops(o.indexLocation.c_str(), o.device + g, &hidx, &ops[g]) != 0) {
                for (int k = 0; k < g; ++k) bind.release_ops(&ops[k]);
                if (hidx) { alfa_host_index_free(hidx); hidx = nullptr; }
                ops.assign((size_t)G, HotOps());
                ops_ready = false;
            }
    }
    if (ops_ready) {
    } else if (o.indexLocation.empty() || alfa_host_index_load(o.indexLocation.c_str(), &hidx) != 0) {
        if (o.ref_fasta.empty()) { fprintf(stderr, "ERROR: index %s could not be loaded and no reference given\n", o.indexLocation.c_str()); return 1; }
        if (alfa_host_index_build_fasta(o.ref_fasta.c_str(), o.K, o.hasSuflink, o.hasChild, o.hasKmer, &hidx)) {
            fprintf(stderr, "ERROR: %s\n", alfa_last_error());
            return 1;
        }
        if (o.saveIndex) alfa_host_index_save(hidx, o.index_prefix.c_str());
    }
    const double t_index_ready = since_start();
    RefIndex idx;
    idx.h = hidx;
    idx.v = alfa_host_index_view(hidx);

    alfa_dp_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.m = 5;
    memcpy(prm.mat, o.aln.mat, 25);
    prm.gapo = -o.aln.openGap; prm.gape = -o.aln.extendGap; prm.zdrop = 100;   // call sites src/mapper.cpp:787-804

    for (int g = 0; g < G && !ops_ready; ++g)
        if (bind.make_ops(o.device + g, hidx, &ops[g])) { fprintf(stderr, "ERROR: cannot set up the hot path on device %d: %s\n", o.device + g, bind.last_error()); return 1; }

    const double t_device_ready = since_start();
    FILE* out = fopen(o.outputName.c_str(), "w");
    if (!out) { fprintf(stderr, "ERROR: cannot write %s\n", o.outputName.c_str()); return 1; }
    std::atomic<bool> write_failed{false};
    if (fputs(sam_header(idx, argc, argv).c_str(), out) < 0) write_failed = true;

    std::vector<std::unique_ptr<HotSched>> scheds((size_t)T);
    std::vector<std::unique_ptr<Mapper>> mappers((size_t)T);
    Pool pool(T, [&](int t) {
        bind.bind_thread(o.device + t % G);
        scheds[t].reset(new HotSched(ops[t % G], prm));
        if (const char* lp = getenv("ALFA_TASK_LOG")) scheds[t]->set_task_log(std::string(lp) + "." + std::to_string(t));
        mappers[t].reset(new Mapper(idx, o.aln, o.paired, *scheds[t]));
    });
    const double t_pool_ready = since_start();
    // Batches follow the reference's driver (BUFFER_SIZE bases per thread, src/main.cpp:54,116) but are also capped at one
    // wave of fibers per worker so that parsing (reader thread), mapping + SAM rendering (worker pool) and writing
    // (writer thread) of successive batches overlap.  Batch boundaries do not influence the output.
    const int64_t bpCount = (int64_t)T * 10000000;
    // with whole reads mapped on the device a worker's slice is one device call: bigger slices (o.mapBatch reads per worker) fill the GPU
    const bool devmap = o.aln.deviceMap && ops[0].map_reads_batch && !o.aln.noFW && !o.aln.noRC && !o.aln.coreTrials && !o.aln.deviceRegions &&
                        (o.pair1.empty() || o.paired.mode == 1);
    const size_t readCap = (size_t)T * (size_t)std::max(256, devmap ? std::max(o.fibers, o.mapBatch) : o.fibers);
    const int minSlice = std::max(1, getenv("ALFA_MIN_SLICE") ? atoi(getenv("ALFA_MIN_SLICE")) : 4096);      // reads per device call at least (long reads)
    const int groupReads = std::max(64, getenv("ALFA_GROUP_READS") ? atoi(getenv("ALFA_GROUP_READS")) : 8192);   // long reads: reads per group call
    const bool direct_render = getenv("ALFA_NO_DIRECT_RENDER") == nullptr;       // (the general path for every read: A/B runs and tests)

    struct Batch {
        // read objects by position in the batch, made at first use: a read that goes from the split record to the device and from
        // the call's results to its SAM line never has one (b stays empty for single-end input; the objects stay with the batch
        // and are used again by later reads at the same position).  Workers touch disjoint positions.
        struct Slots {
            std::vector<std::unique_ptr<Read>> v;
            size_t size() const { return v.size(); }
            void emplace_back() { v.emplace_back(); }
            Read& operator[](size_t i) { if (!v[i]) v[i].reset(new Read()); return *v[i]; }
        };
        Slots a, b;
        std::vector<RawRec> ra, rb;           // records the reader thread only split: the workers materialise them
        std::vector<ReadSource::Chunk> keep;  // gzip input: the inflated chunks those records lie in
        int n = 0;                            // reads (pairs) of this batch
        std::vector<std::string> text;        // rendered SAM, one chunk per worker slice, in input order
        std::vector<long> cnt;                // per slice: mapped1, printed1, mapped2, printed2
        std::atomic<int> remaining{0};        // slices still being worked on
        long seq = 0;                         // position in the input
        // long reads: the batch is cut into a few groups of thousands of reads, ONE worker maps a group in one device call (a
        // call lasts as long as its longest read whatever else it holds, and only a call of many reads keeps the GPU full),
        // then all workers render the group's slices from these results
        struct Group {                        // reads [lo, hi) of the batch, mapped by one worker in one call (+ the rescue rounds' calls)
            int lo = 0, hi = 0;
            bool mapped = false;
            std::vector<RawRead> raw;
            RawStrands strands;
            std::vector<std::unique_ptr<MapCallResult>> calls;
            std::vector<RawResult> res;
        };
        std::vector<std::unique_ptr<Group>> groups;
        int ngroups = 0;                      // groups in use (0: the batch goes to the workers slice by slice)
    };
    typedef std::unique_ptr<Batch> BatchP;
    long total = 0, mapped1 = 0, mapped2 = 0, printed1 = 0, printed2 = 0;
    double t_parse = 0, t_map = 0, t_print = 0;
    std::vector<double> w_build((size_t)T, 0.0), w_run((size_t)T, 0.0), w_render((size_t)T, 0.0), w_dev((size_t)T, 0.0);   // per worker, --stats
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    // fill(batch) -> false at end of input; work(batch, t, k, lo, hi) maps and renders reads [lo,hi) = slice k of the batch on worker t
    // whole(batch, g, t): optional -- maps group g of the batch on worker t when it can (Group::mapped); work() of its slices then only renders
    auto stream = [&](const std::function<bool(Batch&)>& fill, const std::function<void(Batch&, int, int, int, int, int)>& work,
                      const std::function<void(Batch&, int, int)>& whole = nullptr) {
        // batches circulate reader -> dispatcher -> workers -> writer -> reader, so that the reads' strings are allocated once.
        // A batch is cut into T slices that the workers take from one queue, with no barrier between batches: the workers drift
        // apart, so that while some wait for the GPU (one device call per slice) the others build reads or render SAM, instead
        // of all T calls arriving at the GPU at once and all T renderings leaving it idle.  The writer takes finished batches
        // in input order.
        struct Slice { Batch* B; int k, lo, hi, g; };        // g: the group whose call holds the reads' results, or -1
        Chan<BatchP> parsed(2), spare(8);
        // what the workers take: slices (from the dispatcher, at most 2 T waiting; from a worker that has mapped a group,
        // without a bound -- a worker never waits here) and groups to map (at most 2 T waiting)
        struct WorkQ {
            std::mutex mu;
            std::condition_variable cv;
            std::deque<Slice> q;
            std::deque<std::pair<Batch*, int>> wholes;
            size_t cap = 0;
            int wholes_running = 0;
            int slices_running = 0, slices_max = 1;
            bool closed = false;
            void push(Slice sl) {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return q.size() < cap; });
                q.push_back(sl);
                cv.notify_all();
            }
            void push_whole(Batch* B, int grp) {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return wholes.size() < cap; });
                wholes.emplace_back(B, grp);
                cv.notify_all();
            }
            void whole_done(const std::vector<Slice>& sl) {
                std::lock_guard<std::mutex> g(mu);
                for (const Slice& x : sl) q.push_back(x);
                --wholes_running;
                cv.notify_all();
            }
            void close() {
                std::lock_guard<std::mutex> g(mu);
                closed = true;
                cv.notify_all();
            }
            void slice_done() {
                std::lock_guard<std::mutex> g(mu);
                --slices_running;
                cv.notify_all();
            }
            // 1: a slice, 2: a group to map (only for a worker that may take groups), 0: nothing will come any more
            int pop(Slice& sl, std::pair<Batch*, int>& W, bool groups) {
                std::unique_lock<std::mutex> g(mu);
                for (;;) {
                    // slices first: what a finished call returned is rendered (and written) while other workers' calls run --
                    // groups first would leave every rendering and the writer's whole work for the end of the input
                    if (!q.empty() && slices_running < slices_max) { sl = q.front(); q.pop_front(); ++slices_running; cv.notify_all(); return 1; }
                    if (groups && !wholes.empty()) { W = wholes.front(); wholes.pop_front(); ++wholes_running; cv.notify_all(); return 2; }
                    if (closed && wholes_running == 0 && q.empty() && wholes.empty()) return 0;
                    cv.wait(g);
                }
            }
        } slices;
        slices.cap = (size_t)2 * (size_t)T;
        slices.slices_max = busyWorkers;
        std::mutex dm;
        std::condition_variable dcv;
        std::map<long, Batch*> finished;
        long n_dispatched = -1;                       // known once the input has ended
        std::thread reader([&] {
            for (;;) {
                double t0 = now();
                BatchP bp;
                if (!spare.try_pop(bp)) bp.reset(new Batch);
                bp->n = 0;
                bp->keep.clear();
                bool ok = fill(*bp);
                t_parse += now() - t0;
                if (!ok) { (void)bp.release(); break; }      // end of input: see the note on teardown below
                parsed.push(std::move(bp));
            }
            parsed.close();
        });
        std::thread dispatcher([&] {
            long seq = 0;
            BatchP bp;
            while (parsed.pop(bp)) {
                Batch* B = bp.release();
                const int n = B->n;
                B->seq = seq++;
                // a slice is one device call: long reads fill a batch's base budget with few reads, and T calls of a thousand
                // reads each neither fill the GPU (a call lasts as long as its longest read, whatever else it holds) nor get
                // along on the four hardware queues they share -- fewer, larger slices then (minSlice reads at least)
                const int K = devmap ? std::max(1, std::min(T, n / minSlice)) : T;
                B->ngroups = 0;
                if (whole && K < T) {          // few reads for its bases: long reads
                    const int G = std::max(1, std::min(T, (n + groupReads / 2) / groupReads));
                    const int R = (T + G - 1) / G;                  // slices a group is rendered in
                    while ((int)B->groups.size() < G) B->groups.emplace_back(new Batch::Group());
                    B->ngroups = G;
                    B->text.assign((size_t)G * R, std::string());
                    B->cnt.assign((size_t)G * R * 4, 0);
                    B->remaining.store(G * R);
                    for (int g = 0; g < G; ++g) {
                        B->groups[(size_t)g]->lo = (int)((int64_t)n * g / G);
                        B->groups[(size_t)g]->hi = (int)((int64_t)n * (g + 1) / G);
                        slices.push_whole(B, g);
                    }
                    continue;
                }
                B->text.assign((size_t)K, std::string());
                B->cnt.assign((size_t)K * 4, 0);
                B->remaining.store(K);
                for (int k = 0; k < K; ++k) slices.push(Slice{B, k, (int)((int64_t)n * k / K), (int)((int64_t)n * (k + 1) / K), -1});
            }
            { std::lock_guard<std::mutex> g(dm); n_dispatched = seq; }
            dcv.notify_all();
            slices.close();
        });
        std::thread writer([&] {
            for (long next = 0;; ++next) {
                Batch* B = nullptr;
                {
                    std::unique_lock<std::mutex> g(dm);
                    dcv.wait(g, [&] { return finished.count(next) || next == n_dispatched; });
                    auto it = finished.find(next);
                    if (it == finished.end()) break;
                    B = it->second;
                    finished.erase(it);
                }
                double t0 = now();
                total += B->n;
                for (size_t t = 0; t < B->cnt.size() / 4; ++t) {
                    mapped1 += B->cnt[t * 4]; printed1 += B->cnt[t * 4 + 1]; mapped2 += B->cnt[t * 4 + 2]; printed2 += B->cnt[t * 4 + 3];
                }
                for (const std::string& s : B->text)
                    if (fwrite(s.data(), 1, s.size(), out) != s.size()) write_failed = true;
                // back to the reader; never wait here: once the input has ended nobody takes spares any more (a batch that does
                // not fit is left to the operating system like the others, see below)
                BatchP back(B);
                if (!spare.try_push(back)) (void)back.release();
                t_print += now() - t0;
            }
        });
        const double tm0 = now();
        pool.run([&](int t) {
            Slice sl;
            std::pair<Batch*, int> wg;
            // group calls are made by half of each GPU's workers (two at least): eight callers keep a GPU as busy as sixteen, and every
            // caller keeps gigabytes of device buffers for its calls (250 000 long reads: 1.5-1.6 s either way, 119 -> 80 GB)
            static const int group_workers = getenv("ALFA_GROUP_WORKERS") ? std::max(1, atoi(getenv("ALFA_GROUP_WORKERS"))) : 0;      // per GPU
            const bool takes_groups = t / G < (group_workers ? group_workers : std::max(2, (T / G + 1) / 2));
            for (int kind; (kind = slices.pop(sl, wg, takes_groups)) != 0;) {
                if (kind == 2) {
                    Batch* wb = wg.first;
                    const int g = wg.second, G = wb->ngroups, R = (T + G - 1) / G;
                    whole(*wb, g, t);
                    const int lo = wb->groups[(size_t)g]->lo, n = wb->groups[(size_t)g]->hi - lo;
                    std::vector<Slice> cut((size_t)R);
                    for (int k = 0; k < R; ++k) cut[(size_t)k] = Slice{wb, g * R + k, lo + (int)((int64_t)n * k / R), lo + (int)((int64_t)n * (k + 1) / R), g};
                    slices.whole_done(cut);
                    continue;
                }
                work(*sl.B, t, sl.k, sl.lo, sl.hi, sl.g);
                slices.slice_done();
                if (sl.B->remaining.fetch_sub(1) == 1) {
                    for (int g = 0; g < sl.B->ngroups; ++g) sl.B->groups[(size_t)g]->calls.clear();          // (freed on a worker: the writer only writes)
                    { std::lock_guard<std::mutex> g(dm); finished[sl.B->seq] = sl.B; }
                    dcv.notify_all();
                }
            }
        });
        t_map += now() - tm0;
        reader.join();
        dispatcher.join();
        writer.join();
        // the batches left in circulation hold hundreds of thousands of reads: tearing them down string by string costs more than
        // anything else at the end of a run (0.3-0.6 s per 10^6 reads in flight) and the process is about to end -- they are left
        // to the operating system
        BatchP left;
        while (spare.try_pop(left)) (void)left.release();
    };

    if (!o.unpairedQ.empty()) {
        ReadSource rd;
        if (!rd.open(o.unpairedQ.c_str())) { fprintf(stderr, "ERROR: file %s could not be read.", o.unpairedQ.c_str()); return 1; }
        stream(
            [&](Batch& B) {
                int64_t cur = 0;
                while (cur < bpCount && (size_t)B.n < readCap) {
                    if ((int)B.a.size() <= B.n) { B.a.emplace_back(); B.ra.emplace_back(); }
                    const int got = rd.next([&]() -> Read& { return B.a[(size_t)B.n]; }, B.ra[B.n]);
                    if (got == 1 && rd.chunk() && (B.keep.empty() || B.keep.back() != rd.chunk())) B.keep.push_back(rd.chunk());
                    if (!got) break;
                    cur += got == 1 ? (int64_t)B.ra[B.n].l2 : (int64_t)B.a[(size_t)B.n].sequence.length();
                    B.n++;
                }
                return B.n > 0;
            },
            [&](Batch& B, int t, int k, int lo, int hi, int grp) {
                Mapper& mp = *mappers[t];
                std::string& txt = B.text[(size_t)k];
                long* c = &B.cnt[(size_t)k * 4];
                const double tb0 = now();
                const Batch::Group* GR = grp >= 0 && B.groups[(size_t)grp]->mapped ? B.groups[(size_t)grp].get() : nullptr;
                bool raw_ok = GR || (direct_render && mp.device_map_usable());
                for (int i = lo; i < hi && raw_ok && !GR; i++) raw_ok = B.ra[i].p != nullptr;
                if (raw_ok) {
                    // records the reader only split go to the device as they are; a read that comes back finished, with at most
                    // one alignment and its CIGAR, is rendered straight from the call's results -- no Read, no Alignment objects
                    const int n = hi - lo;
                    static thread_local std::vector<RawRead> raw_own;
                    static thread_local RawStrands strands_own;
                    std::vector<std::unique_ptr<MapCallResult>> calls;
                    std::vector<RawResult> res_own;
                    const double td0 = now();
                    if (!GR) {
                        raw_own.resize((size_t)n);
                        for (int i = 0; i < n; i++) raw_own[(size_t)i] = raw_read(B.ra[lo + i]);
                        mp.map_raw(raw_own.data(), n, strands_own, calls, res_own);
                        w_dev[t] += now() - td0;
                    }
                    // the slice's reads in the call's arrays: the group's (read lo of the batch is entry lo - GR->lo) or this slice's own
                    const int sbase = GR ? lo - GR->lo : 0;
                    const RawRead* raw = GR ? GR->raw.data() + sbase : raw_own.data();
                    const RawResult* res = GR ? GR->res.data() + sbase : res_own.data();
                    const RawStrands& strands = GR ? GR->strands : strands_own;
                    std::vector<uint8_t> direct((size_t)n);
                    std::vector<int> rest;
                    std::vector<uint8_t> done((size_t)n);
                    for (int i = 0; i < n; i++) {
                        if ((direct[(size_t)i] = mp.direct_ok(res[(size_t)i]))) continue;
                        Read& r = B.a[lo + i];
                        materialize(B.ra[lo + i], r);
                        r.init();
                        done[(size_t)i] = mp.device_done(res[(size_t)i]);
                        if (done[(size_t)i]) mp.adopt(r, *res[(size_t)i].R, res[(size_t)i].j);
                        if (!(done[(size_t)i] && mp.finish_unpaired_direct(r))) rest.push_back(i);
                    }
                    scheds[t]->run((int)rest.size(), o.fibers, [&](int k) { mp.map_unpaired_rest(B.a[lo + rest[(size_t)k]], done[(size_t)rest[(size_t)k]] != 0); });
                    const double tb2 = now();
                    w_build[t] += td0 - tb0; w_run[t] += tb2 - td0;
                    size_t bytes = 0;
                    for (int i = 0; i < n; i++) bytes += 2 * (size_t)raw[(size_t)i].len + (size_t)raw[(size_t)i].nameLen + 160;
                    txt.reserve(txt.size() + bytes);
                    long with_cigar = 0;
                    for (int i = 0; i < n; i++) {
                        if (direct[(size_t)i]) {
                            const long before = c[0];
                            mp.render_direct(strands, raw[(size_t)i], sbase + i, res[(size_t)i], txt, c);
                            with_cigar += c[0] - before;
                            continue;
                        }
                        Read& r = B.a[lo + i];
                        if (r.alignments.empty()) txt += r.emptyAlignment(false, false, true);
                        else {
                            c[0]++;
                            int k = std::min((int)r.alignments.size(), o.aln.alignmentCount);
                            c[1] += k;
                            for (int j = 0; j < k; j++) txt += r.printUnpaired(j);
                        }
                        r.release();
                    }
                    g_map_us[3] += with_cigar;
                    w_render[t] += now() - tb2;
                    return;
                }
                for (int i = lo; i < hi; i++) materialize(B.ra[i], B.a[i]);
                const double tb1 = now();
                if (mp.device_map_usable()) {
                    // whole reads on the device first (seeds, regions, trial loop with its DP: alfa_map_reads_batch); the fibers
                    // then run what is left: CIGAR recovery and MAPQ of every read, the whole mapping of the reads handed back
                    std::vector<Read*> ptr((size_t)(hi - lo));
                    for (int i = lo; i < hi; i++) { B.a[i].init(); ptr[(size_t)(i - lo)] = &B.a[i]; }
                    std::vector<uint8_t> done;
                    const double td0 = now();
                    mp.map_batch(ptr.data(), hi - lo, done);
                    w_dev[t] += now() - td0;
                    // a read whose CIGARs came with its alignments is finished right here (sort, strings, MAPQ); only the others
                    // need a fiber: ksw_global calls, or the whole mapping of a read that was handed back
                    std::vector<int> rest;
                    for (int i = lo; i < hi; i++)
                        if (!(done[(size_t)(i - lo)] && mp.finish_unpaired_direct(B.a[i]))) rest.push_back(i);
                    scheds[t]->run((int)rest.size(), o.fibers, [&](int k) { mp.map_unpaired_rest(B.a[rest[(size_t)k]], done[(size_t)(rest[(size_t)k] - lo)] != 0); });
                } else
                    scheds[t]->run(hi - lo, o.fibers, [&](int i) { mp.map_unpaired(B.a[lo + i]); });
                const double tb2 = now();
                w_build[t] += tb1 - tb0; w_run[t] += tb2 - tb1;
                struct Fin { double& acc; double t0; decltype(now)& clk; ~Fin() { acc += clk() - t0; } } fin{w_render[t], tb2, now};
                for (int i = lo; i < hi; i++) {
                    Read& r = B.a[i];
                    if (r.alignments.empty()) txt += r.emptyAlignment(false, false, true);
                    else {
                        c[0]++;
                        int k = std::min((int)r.alignments.size(), o.aln.alignmentCount);
                        c[1] += k;
                        for (int j = 0; j < k; j++) txt += r.printUnpaired(j);
                    }
                    r.release();          // on the worker: the writer thread only writes
                }
            },
            [&](Batch& B, int grp, int t) {
                // a group of long reads through the device in one call (and the rescue rounds' calls) on this worker
                Mapper& mp = *mappers[t];
                Batch::Group& G = *B.groups[(size_t)grp];
                G.mapped = false;
                G.calls.clear();
                if (!direct_render || !mp.device_map_usable() || getenv("ALFA_NO_GROUP_CALLS")) return;
                const int n = G.hi - G.lo;
                for (int i = 0; i < n; i++)
                    if (!B.ra[G.lo + i].p) return;
                const double td0 = now();
                G.raw.resize((size_t)n);
                for (int i = 0; i < n; i++) G.raw[(size_t)i] = raw_read(B.ra[G.lo + i]);
                mp.map_raw(G.raw.data(), n, G.strands, G.calls, G.res);
                const double td1 = now();
                w_dev[t] += td1 - td0; w_run[t] += td1 - td0;
                G.mapped = true;
            });
        printf("sequences read: %ld\nsequences mapped: %ld\nalignments printed: %ld\n", total, mapped1, printed1);
    }
    if (!o.pair1.empty() && !o.pair2.empty()) {
        ReadSource r1, r2;
        if (!r1.open(o.pair1.c_str()) || !r2.open(o.pair2.c_str())) { fprintf(stderr, "ERROR: mate files could not be read."); return 1; }
        total = mapped1 = printed1 = mapped2 = printed2 = 0;
        stream(
            [&](Batch& B) {
                int64_t cur = 0;
                while (cur < bpCount && (size_t)B.n < readCap) {
                    if ((int)B.a.size() <= B.n) { B.a.emplace_back(); B.b.emplace_back(); B.ra.emplace_back(); B.rb.emplace_back(); }
                    Read& a = B.a[B.n];
                    Read& b = B.b[B.n];
                    const int ok1 = r1.next([&]() -> Read& { return a; }, B.ra[B.n]);
                    const int ok2 = r2.next([&]() -> Read& { return b; }, B.rb[B.n]);     // the reference keeps only the second status (src/main.cpp:157-158)
                    for (ReadSource* rs : {&r1, &r2})
                        if (rs->chunk() && std::find(B.keep.begin(), B.keep.end(), rs->chunk()) == B.keep.end()) B.keep.push_back(rs->chunk());
                    if (!ok1) {
                        // mate file 1 ended first: the reference's reader leaves its read object as it was.  The object may be a
                        // spare of an earlier batch here, so give it the state a fresh read_t has
                        B.ra[B.n].p = nullptr;
                        a.qname.clear(); a.sequence.clear(); a.qual = "*";
                    }
                    if (!ok2) break;
                    cur += ok1 == 1 ? (int64_t)B.ra[B.n].l2 : (int64_t)a.sequence.length();
                    B.n++;
                }
                return B.n > 0;
            },
            [&](Batch& B, int t, int k, int lo, int hi, int) {
                Mapper& mp = *mappers[t];
                std::string& txt = B.text[(size_t)k];
                long* c = &B.cnt[(size_t)k * 4];
                for (int i = lo; i < hi; i++) { materialize(B.ra[i], B.a[i]); materialize(B.rb[i], B.b[i]); }
                if (mp.splittable() && mp.device_map_usable()) {
                    // paired mode 1 = unpairedMatch on each mate, then pairing: both mates of every pair go to the device as reads
                    const int nm = 2 * (hi - lo);
                    std::vector<Read*> ptr((size_t)nm);
                    for (int i = 0; i < nm; i++) { Read& r = (i & 1) ? B.b[lo + (i >> 1)] : B.a[lo + (i >> 1)]; r.init(); ptr[(size_t)i] = &r; }
                    std::vector<uint8_t> done;
                    const double td0 = now();
                    mp.map_batch(ptr.data(), nm, done);
                    w_dev[t] += now() - td0;
                    scheds[t]->run(nm, o.fibers, [&](int i) { mp.map_mate_rest(*ptr[(size_t)i], done[(size_t)i] != 0); });
                    scheds[t]->run(hi - lo, o.fibers, [&](int i) { mp.pair_mates(B.a[lo + i], B.b[lo + i]); });
                    scheds[t]->run(nm, o.fibers, [&](int i) { mp.finish_mate(*ptr[(size_t)i]); });
                } else if (mp.splittable()) {
                    // the mates of a pair are independent until they are paired: twice the fibers, half the chain of batch rounds
                    scheds[t]->run(2 * (hi - lo), o.fibers, [&](int i) { mp.map_mate((i & 1) ? B.b[lo + (i >> 1)] : B.a[lo + (i >> 1)]); });
                    scheds[t]->run(hi - lo, o.fibers, [&](int i) { mp.pair_mates(B.a[lo + i], B.b[lo + i]); });
                    scheds[t]->run(2 * (hi - lo), o.fibers, [&](int i) { mp.finish_mate((i & 1) ? B.b[lo + (i >> 1)] : B.a[lo + (i >> 1)]); });
                } else {
                    scheds[t]->run(hi - lo, o.fibers, [&](int i) { mp.map_paired(B.a[lo + i], B.b[lo + i]); });
                }
                for (int i = lo; i < hi; i++) {
                    Read& m1 = B.a[i];
                    Read& m2 = B.b[i];
                    if (m1.alignments.empty()) txt += m1.emptyAlignment(true, m2.alignments.empty(), true);
                    else {
                        c[0]++;
                        int k = o.paired.mixed ? std::min((int)m1.alignments.size(), o.aln.alignmentCount)
                                               : std::min(m1.pairedAlignmentCount, o.aln.alignmentCount);
                        c[1] += k;
                        for (int j = 0; j < k; j++) txt += m1.printPaired(j, true);
                    }
                    if (m2.alignments.empty()) txt += m2.emptyAlignment(true, m1.alignments.empty(), false);
                    else {
                        c[2]++;
                        int k = o.paired.mixed ? std::min((int)m2.alignments.size(), o.aln.alignmentCount)
                                               : std::min(m2.pairedAlignmentCount, o.aln.alignmentCount);
                        c[3] += k;
                        for (int j = 0; j < k; j++) txt += m2.printPaired(j, false);
                    }
                    m1.release();
                    m2.release();
                }
            });
        printf("sequences read: %ld\nsequences mapped (mate1): %ld\nsequences mapped (mate2): %ld\n"
               "alignments printed (mate1): %ld\nalignments printed (mate2): %ld\n", total, mapped1, mapped2, printed1, printed2);
    }
    if (fclose(out) != 0) write_failed = true;
    if (write_failed) { fprintf(stderr, "ERROR: writing %s failed (disk full?)\n", o.outputName.c_str()); return 1; }
    if (o.stats) {
        SchedStats s;
        for (auto& sc : scheds) {
            const SchedStats& x = sc->stats();
            s.rounds += x.rounds; s.seed_calls += x.seed_calls; s.seed_queries += x.seed_queries; s.ext_calls += x.ext_calls;
            s.ext_tasks += x.ext_tasks; s.nw_calls += x.nw_calls; s.nw_tasks += x.nw_tasks; s.glob_calls += x.glob_calls;
            s.glob_tasks += x.glob_tasks; s.align_calls += x.align_calls; s.align_tasks += x.align_tasks; s.hot_seconds += x.hot_seconds;
        }
        fprintf(stderr, "#PHASES parse %.3f map %.3f print %.3f\n", t_parse, t_map, t_print);
        fprintf(stderr, "#CORE reads_done %ld handed_back %ld cigars_with_alignments %ld device_call_thread_seconds: describe %.3f call %.3f adopt %.3f lane_wait %.3f\n",
                g_core_done.load(), g_core_fallback.load(), g_map_us[3].load(), g_map_us[0].load() * 1e-6, g_map_us[1].load() * 1e-6, g_map_us[2].load() * 1e-6, g_map_us[4].load() * 1e-6);
        fprintf(stderr, "#QUEUES GPU_MAX_HW_QUEUES %s\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset");
        fprintf(stderr, "#STARTUP index_loaded %.3f device_ready %.3f workers_ready %.3f done %.3f (seconds since start)\n", t_index_ready,
                t_device_ready, t_pool_ready, since_start());
        double sb = 0, sr = 0, sn = 0, mr = 0, sd = 0;
        for (int t = 0; t < T; ++t) { sb += w_build[t]; sr += w_run[t]; sn += w_render[t]; mr = std::max(mr, w_run[t]); sd += w_dev[t]; }
        fprintf(stderr, "#WORKERS threads %d build_reads %.3f map %.3f (slowest %.3f; of which whole-read device calls %.3f) render %.3f (thread-seconds)\n", T, sb, sr, mr, sd, sn);
        fprintf(stderr, "#STATS rounds %ld seed_calls %ld seed_queries %ld extend %ld/%ld nw %ld/%ld global %ld/%ld align %ld/%ld hot_seconds %.3f\n",
                (long)s.rounds, (long)s.seed_calls, (long)s.seed_queries, (long)s.ext_calls, (long)s.ext_tasks, (long)s.nw_calls,
                (long)s.nw_tasks, (long)s.glob_calls, (long)s.glob_tasks, (long)s.align_calls, (long)s.align_tasks, s.hot_seconds);
    }
    if (o.stats) {      // against the caller's clock: what loading the program cost before main, and what ending it costs after this line
        const double epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        fprintf(stderr, "#CLOCK mapper_start %.3f last_line %.3f (seconds since the epoch)\n", epoch - since_start(), epoch);
    }
    // Orderly teardown on purpose.  Ending the process right here (_exit after flushing) was measured and is SLOWER: the kernel
    // driver takes 0.24-0.37 s to reap a process that still holds a live GPU context, against 0.04 s (1 worker) to 0.2 s (16
    // workers) when HIP has shut down first (tools/exit_cost.sh, tools/probes/exit_probe.hip).
    for (int g = 0; g < G; ++g) bind.release_ops(&ops[g]);
    alfa_host_index_free(hidx);
    return 0;
}

}  // namespace alfa
