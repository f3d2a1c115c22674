// mapper.cpp -- see mapper.h.  Host logic around the hot path with the reference's semantics.
#include "mapper.h"

#include "../map_core.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <mutex>
#include <condition_variable>
#include <map>
#include <charconv>
#include <immintrin.h>
#include <type_traits>

namespace alfa {

struct Mapper_Lis_view { const std::vector<Match>* matches; bool fw; int score; };

namespace {

// append-only text builder with the stream syntax the SAM/CIGAR code uses (decimal integers, chars, strings)
struct Txt {
    std::string s;
    Txt& operator<<(const std::string& v) { s += v; return *this; }
    Txt& operator<<(const char* v) { s += v; return *this; }
    Txt& operator<<(char v) { s += v; return *this; }
    template <class I, class = typename std::enable_if<std::is_integral<I>::value>::type>
    Txt& operator<<(I v) {
        char b[24];
        auto r = std::to_chars(b, b + sizeof(b), v);
        s.append(b, (size_t)(r.ptr - b));
        return *this;
    }
    std::string& str() { return s; }
};

const int MIN_SEED_LENGTH = 20;   // src/mapper.cpp:36
const int MAX_MAPQ = 60;          // :37
const int64_t MAX_OVERLAP = -10;  // :38

// Utils::ORDVALUE src/utils.h:55-80
inline uint8_t nt4(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

// Utils::RC_TABLE src/utils.h:28-53
struct RcTable {
    char t[256];
    RcTable() {
        for (int i = 0; i < 256; ++i) t[i] = 'n';
        for (int i = 65; i <= 96; ++i) t[i] = 'N';
        const char* from = "ABCDGHKMRSTVWY";
        const char* to = "TVGHCDMKYSABWR";
        for (int i = 0; from[i]; ++i) { t[(int)from[i]] = to[i]; t[(int)from[i] + 32] = (char)(to[i] + 32); }
    }
};
const RcTable RC;

// bandSizeEstimate src/mapper.cpp:44-52: one restatement for the host and the device (csrc/map_core.h)
using mapcore::band_estimate;

// recoverCIGAR4 src/mapper.cpp:80-117 (also the body of setFieldsKsw :128-162): a BAM CIGAR written out with its M runs split
// into maximal stretches of equal ('=') and unequal ('X') residues, in order; returns mismatches + gap bases.  A run of
// length 0 cannot occur (ksw_global merges operations); the reference would print "0X" for it, and so does this.
int cigar4(const std::vector<uint32_t>& bam, const uint8_t* query, const uint8_t* rseq, Txt& sNM, Txt* sCig) {
    int x = 0, y = 0, edits = 0;
    for (const uint32_t c : bam) {
        const int op = (int)(c & 0xf), len = (int)(c >> 4);
        if (sCig) *sCig << len << "MIDSH"[op];
        if (op != 0) {
            sNM << len << "MIDSH"[op];
            if (op == 1) { x += len; edits += len; }
            else if (op == 2) { y += len; edits += len; }
            continue;
        }
        if (len <= 0) sNM << 0 << 'X';
        for (int at = 0; at < len;) {
            const bool same = query[x + at] == rseq[y + at];
            int run = 1;
            while (at + run < len && (query[x + at + run] == rseq[y + at + run]) == same) ++run;
            sNM << run << (same ? '=' : 'X');
            if (!same) edits += run;
            at += run;
        }
        x += len; y += len;
    }
    return edits;
}

// recoverCIGAR3from4 src/mapper.cpp:54-77
std::string cigar3_from4(const std::string& c4) {
    Txt out;
    size_t p = 0;
    long matchLength = 0;
    while (p < c4.size()) {
        size_t q = p;
        long long nn = 0;
        while (q < c4.size() && c4[q] >= '0' && c4[q] <= '9') { nn = nn * 10 + (c4[q] - '0'); q++; }
        if (q == p || q >= c4.size()) break;
        int n = (int)nn;
        char sym = c4[q];
        if (sym == 'X' || sym == '=') matchLength += n;
        else {
            if (matchLength > 0) { out << matchLength << 'M'; matchLength = 0; }
            out << n << sym;
        }
        p = q + 1;
    }
    if (matchLength > 0) out << matchLength << 'M';
    return std::move(out.s);
}

bool cmp_matches(const Match& i, const Match& j) { return i.ref < j.ref || (i.ref == j.ref && i.query < j.query); }            // compMatches :570
bool cmp_matches_smem(const Match& i, const Match& j) { return i.query < j.query || (i.query == j.query && i.len > j.len); }  // compMatchesSMEM :574
bool cmp_matches_query(const Match& i, const Match& j) { return i.query < j.query || (i.query == j.query && i.ref < j.ref); } // compMatchesQuery mapper.h:37
bool cmp_aln(const Alignment* i, const Alignment* j) { return i->alignmentScore > j->alignmentScore; }                          // compAlignments :295
bool cmp_aln_paired(const Alignment* i, const Alignment* j) {                                                                   // compPairedAlignments :299
    return i->alignmentScore + i->mate.alignmentScore > j->alignmentScore + j->mate.alignmentScore;
}

struct Bounds { int64_t refB, refE; int queryB, queryE; };

}  // namespace

void AlignOpt::fill_m() {
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? match : mismatch);
        mat[k++] = -1;
    }
    for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

Read::~Read() {
    for (Alignment* a : alignments) delete a;
}
Read::Read(Read&& o) noexcept
    : qname(std::move(o.qname)), sequence(std::move(o.sequence)), qual(std::move(o.qual)), rcSequence(std::move(o.rcSequence)),
      rQual(std::move(o.rQual)), readP(std::move(o.readP)), rcReadP(std::move(o.rcReadP)), alignments(std::move(o.alignments)),
      pairedAlignmentCount(o.pairedAlignmentCount) {
    o.alignments.clear();
}

void Read::release() {
    for (Alignment* a : alignments) delete a;
    alignments.clear();
    pairedAlignmentCount = 0;
}

// read_t::init src/mapper.h:155-166: reverse complement, reversed qualities, nt4 codes of both strands -- one pass with tables
// (the byte-wise switch version was a third of a worker's time once whole reads are mapped on the device)
namespace {
struct Nt4Table {
    uint8_t t[256];
    Nt4Table() { for (int i = 0; i < 256; ++i) t[i] = nt4((unsigned char)i); }
};
const Nt4Table NT4;
}  // namespace

void Read::init() {
    const size_t n = sequence.size(), nq = qual.size();
    rcSequence.resize(n);
    rQual.resize(nq);
    readP.resize(n);
    rcReadP.resize(n);
    const unsigned char* s = (const unsigned char*)sequence.data();
    char* rc = &rcSequence[0];
    uint8_t* p = readP.data();
    uint8_t* rp = rcReadP.data();
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = s[i];
        const char r = RC.t[c];
        p[i] = NT4.t[c];
        rc[n - 1 - i] = r;
        rp[n - 1 - i] = NT4.t[(unsigned char)r];
    }
    const char* q = qual.data();
    char* rq = &rQual[0];
    for (size_t i = 0; i < nq; ++i) rq[nq - 1 - i] = q[i];
}

// read_t::emptyAlingment src/mapper.h:170-179
std::string Read::emptyAlignment(bool paired, bool mateFailed, bool upstream) const {
    Txt ss;
    int flag = 4;
    if (paired) flag += 1;
    if (mateFailed) flag += 8;
    if (paired && upstream) flag += 64;
    if (paired && !upstream) flag += 128;
    ss << qname << "\t" << flag << "\t*\t0\t0\t*\t*\t0\t0\t" << sequence << "\t" << qual << "\n";
    return std::move(ss.s);
}

// read_t::printUnpairedAlignment src/mapper.h:181-193
std::string Read::printUnpaired(int i) const {
    Txt ss;
    const Alignment* a = alignments[i];
    ss << qname << "\t" << (unsigned long)a->flag << "\t" << a->rname << "\t" << a->pos << "\t" << a->mapq << "\t" << a->cigar << "\t"
       << "*" << "\t" << 0 << "\t" << 0 << "\t" << (a->rc() ? rcSequence : sequence) << "\t" << (a->rc() ? rQual : qual) << "\tAS:i:"
       << a->alignmentScore << "\tNM:i:" << a->editDist << "\tX0:Z:" << a->NMtag << "\n";
    return std::move(ss.s);
}

// read_t::printPairedAlignments src/mapper.h:196-215
std::string Read::printPaired(int i, bool firstMate) {
    Alignment* a = alignments[i];
    if (a->paired()) {
        Txt ss;
        ss << qname << "\t" << (unsigned long)(a->flag | a->mate.flag) << "\t" << a->rname << "\t" << a->pos << "\t" << a->mapq << "\t"
           << a->cigar << "\t" << (a->rname == a->mate.rnext ? "=" : a->mate.rnext) << "\t" << a->mate.pnext << "\t" << a->mate.tLength
           << "\t" << (a->rc() ? rcSequence : sequence) << "\t" << (a->rc() ? rQual : qual) << "\tAS:i:"
           << a->alignmentScore + a->mate.alignmentScore << "\tNM:i:" << a->editDist << "\tX0:Z:" << a->NMtag << "\n";
        return std::move(ss.s);
    }
    // alignment_t::setUnPaired src/mapper.h:114-119
    a->flag |= 1u;
    if (firstMate) { a->flag |= 1u << 6; a->flag &= ~(1u << 7); }
    else { a->flag &= ~(1u << 6); a->flag |= 1u << 7; }
    a->flag |= 1u << 3;
    return printUnpaired(i);
}

// createHeader src/main.cpp:266-283
std::string sam_header(const RefIndex& idx, int argc, char** argv) {
    Txt ss;
    ss << "@HD\tVN:1.3\n";
    const int n = idx.v->n_seq;
    for (int i = 0; i + 1 < n; i++) ss << "@SQ\tSN:" << idx.descr(i) << "\tLN:" << (idx.v->startpos[i + 1] - idx.v->startpos[i] - 1) << "\n";
    ss << "@SQ\tSN:" << idx.descr(n - 1) << "\tLN:" << (idx.v->N - idx.v->startpos[n - 1]) << "\n";   // includes the '$' padding
    ss << "@PG\tID:alfalfa\tCL:";
    for (int i = 0; i < argc; i++) { if (i > 0) ss << " "; ss << argv[i]; }
    ss << "\tVN:0.8.1-mi355x\n";
    return std::move(ss.s);
}

// ------------------------------------------------------------------------------------------ seeds
// calculateSeedsBothStrands src/mapper.cpp:530-568
void Mapper::seeds_both_strands(const std::string& P, const std::string& Prc, int min_len, std::vector<Match>& fw, std::vector<Match>& rc) {
    const int K = (int)idx_.K();
    int sparseSkip = 1;
    if (K >= 4) sparseSkip = std::max((min_len - 10) / K, 1);
    else sparseSkip = std::max((min_len - 12) / K, 1);
    if (opt_.noSparseQ) sparseSkip = 1;
    auto both = [&](int ml, int maxBranch, int skip) {
        if (!opt_.noFW) hot_.submit_seed(SeedReq{&P, ml, maxBranch, skip, opt_.memType, opt_.maxSmemMemStart, &fw});
        if (!opt_.noRC) hot_.submit_seed(SeedReq{&Prc, ml, maxBranch, skip, opt_.memType, opt_.maxSmemMemStart, &rc});
        hot_.wait();
    };
    both(min_len, opt_.maxSeedCandidates, sparseSkip);
    if (opt_.rescue && fw.empty() && rc.empty()) {
        int minLength = std::max(MIN_SEED_LENGTH, K + 10);
        both((minLength + min_len) / 2, 100000, 1);
        if (fw.empty() && rc.empty()) both(minLength, 100000, 1);
    }
}

// identifyAndFilter src/mapper.cpp:578-704
void Mapper::identify_and_filter(std::vector<Match>& matches, bool fw, int qLength, int editDist, std::vector<Lis>& lis) {
    if (matches.empty()) return;
    const size_t emptySize = lis.size();
    const size_t maxSmemOcc = (size_t)opt_.maxSeedCandidates;
    const int64_t minMemLength = opt_.minMemRegLength;
    const size_t maxPossOccIncrease = (size_t)opt_.maxRegionMems;
    std::sort(matches.begin(), matches.end(), cmp_matches_smem);
    std::vector<Match> smems, reserves;
    smems.reserve(matches.size());
    reserves.reserve(matches.size());
    int64_t previousStart, previousEnd, previousREnd = -1;
    size_t rangeStart = 0, currentSmemRange = 0, index = 0;
    while (index < matches.size()) {
        rangeStart = index;
        previousStart = matches[index].query;
        previousEnd = matches[index].query + matches[index].len;
        index++;
        while (index < matches.size() && matches[index].query == previousStart && matches[index].query + matches[index].len == previousEnd) index++;
        if (previousEnd > previousREnd) currentSmemRange = index - rangeStart;
        else currentSmemRange += index - rangeStart;
        if ((previousEnd > previousREnd && currentSmemRange <= maxSmemOcc) ||
            (currentSmemRange <= maxPossOccIncrease && previousEnd - previousStart >= minMemLength)) {
            for (size_t j = rangeStart; j < index; j++) smems.push_back(matches[j]);
        } else {
            for (size_t j = rangeStart; j < index; j++) reserves.push_back(matches[j]);
        }
        previousREnd = previousEnd;
    }
    std::sort(smems.begin(), smems.end(), cmp_matches);
    int begin = 0, end = 0, longest = 0;
    int64_t refEnd, refBegin;
    while (begin < (int)smems.size()) {
        refEnd = smems[begin].ref - smems[begin].query + qLength + editDist;
        longest = begin;
        end = begin + 1;
        while (end < (int)smems.size() && smems[end].ref + smems[end].len <= refEnd) {
            if (smems[longest].len < smems[end].len) {
                longest = end;
                refEnd = smems[longest].ref - smems[longest].query + qLength + editDist;
            }
            end++;
        }
        refBegin = smems[longest].ref - smems[longest].query - editDist;
        begin--;
        const int beginFix = begin;
        while (begin >= 0 && smems[begin].ref >= refBegin) begin--;
        if (emptySize == lis.size() || lis.back().refEnd < refBegin) {
            lis.push_back(Lis());
            lis.back().fw = fw;
            for (int i = begin + 1; i <= end - 1; i++) lis.back().matches.push_back(smems[i]);
            lis.back().refBegin = refBegin;
            lis.back().refEnd = refEnd;
        } else {
            for (int i = beginFix + 1; i <= end - 1; i++) lis.back().matches.push_back(smems[i]);
            lis.back().refBegin = std::min(lis.back().refBegin, refBegin);
            lis.back().refEnd = std::min(lis.back().refEnd, refEnd);     // the reference takes min for both ends (:673-674)
        }
        begin = end;
    }
    if (emptySize < lis.size()) {
        std::sort(reserves.begin(), reserves.end(), cmp_matches);
        size_t curReserve = 0, curInterval = emptySize;
        while (curReserve < reserves.size() && curInterval < lis.size()) {
            if (reserves[curReserve].ref < lis[curInterval].refBegin) curReserve++;
            else if (lis[curInterval].refEnd < reserves[curReserve].ref) curInterval++;
            else { lis[curInterval].matches.push_back(reserves[curReserve]); curReserve++; }
        }
    }
}

// coveredBases src/mapper.cpp:706-719
static int covered_bases(std::vector<Match>& seeds) {
    std::sort(seeds.begin(), seeds.end(), cmp_matches_query);
    int trueCover = seeds[0].len;
    int prevEnd = seeds[0].query + seeds[0].len - 1;
    for (size_t i = 1; i < seeds.size(); i++) {
        int start = std::max(seeds[i].query, prevEnd + 1);
        int stop = std::max(prevEnd, seeds[i].query + seeds[i].len - 1);
        prevEnd = stop;
        trueCover += std::max(0, stop - start + 1);
    }
    return trueCover;
}

// calculateCandRegions src/mapper.cpp:1220-1267
void Mapper::cand_regions(const Read& read, std::vector<Lis>& lis, double errorPercent, bool sortByScore) {
    const std::string& P = read.sequence;
    const std::string& Prc = read.rcSequence;
    const int Plength = (int)P.length();
    const int editDist = (int)(errorPercent * Plength) + 1;
    int min_len = opt_.minMemLength;
    if (!opt_.fixedMinLength && Plength >= 1000) min_len += (20 * ((int)(Plength / 500))) / ((int)100 * errorPercent);
    if (opt_.fixedMinLength && Plength / editDist > min_len) min_len = Plength / editDist;
    std::vector<Match> matches, matchesRC;
    if (opt_.deviceRegions && hot_.has_seed_regions()) {
        // the hot path returns the seeds AND the regions built from them (alfa_seed_regions_batch); strands it hands back
        // (status 1) and the rescue rounds' control flow stay here.  Same requests, same rounds as seeds_both_strands.
        std::vector<RegionOut> regF, regR;
        int stF = 0, stR = 0;
        const int K = (int)idx_.K();
        int sparseSkip = K >= 4 ? std::max((min_len - 10) / K, 1) : std::max((min_len - 12) / K, 1);
        if (opt_.noSparseQ) sparseSkip = 1;
        const alfa_region_params rp{opt_.maxSeedCandidates, opt_.maxRegionMems, opt_.minMemRegLength, 0, opt_.errorPercent};
        auto both = [&](int ml, int maxBranch, int skip) {
            if (!opt_.noFW) hot_.submit_seed_regions(SeedRegionReq{&P, ml, maxBranch, skip, opt_.memType, opt_.maxSmemMemStart, editDist, rp, &matches, &regF, &stF});
            if (!opt_.noRC) hot_.submit_seed_regions(SeedRegionReq{&Prc, ml, maxBranch, skip, opt_.memType, opt_.maxSmemMemStart, editDist, rp, &matchesRC, &regR, &stR});
            hot_.wait();
        };
        both(min_len, opt_.maxSeedCandidates, sparseSkip);
        if (opt_.rescue && matches.empty() && matchesRC.empty()) {
            int minLength = std::max(MIN_SEED_LENGTH, K + 10);
            both((minLength + min_len) / 2, 100000, 1);
            if (matches.empty() && matchesRC.empty()) both(minLength, 100000, 1);
        }
        if (matches.size() > 0 || matchesRC.size() > 0) {
            auto take = [&](std::vector<Match>& mm, std::vector<RegionOut>& rg, int st, bool fw) {
                if (st != 0) {
                    const size_t first = lis.size();
                    identify_and_filter(mm, fw, Plength, editDist, lis);
                    for (size_t i = first; i < lis.size(); i++) lis[i].score = covered_bases(lis[i].matches);
                    return;
                }
                for (RegionOut& r : rg) {
                    lis.push_back(Lis());
                    Lis& l = lis.back();
                    l.fw = fw; l.refBegin = r.refBegin; l.refEnd = r.refEnd; l.score = r.score;
                    l.matches.swap(r.matches);
                    l.chain0.swap(r.chain); l.chain0Anchor = r.chainAnchor; l.chain0Edit = r.chainEdit; l.chain0Coverage = r.chainCoverage;
                }
            };
            take(matches, regF, stF, true);
            take(matchesRC, regR, stR, false);
            std::sort(lis.begin(), lis.end(), [](const Lis& a, const Lis& b) { return a.score > b.score; });   // compIntervals :1114
        }
        return;
    }
    seeds_both_strands(P, Prc, min_len, matches, matchesRC);
    if (matches.size() > 0 || matchesRC.size() > 0) {
        // ALFA_REGION_LOG=<file> (diagnostics; tests/test_regions_oracle.py reads it): the seeds of both strands and the regions
        // built from them, as text
        static FILE* rlog = getenv("ALFA_REGION_LOG") ? fopen(getenv("ALFA_REGION_LOG"), "w") : nullptr;
        static std::mutex rlog_mu;
        std::vector<Match> in_fw, in_rc;
        if (rlog) { in_fw = matches; in_rc = matchesRC; }
        identify_and_filter(matches, true, Plength, editDist, lis);
        identify_and_filter(matchesRC, false, Plength, editDist, lis);
        for (size_t i = 0; i < lis.size(); i++) lis[i].score = covered_bases(lis[i].matches);
        if (rlog) {
            std::lock_guard<std::mutex> g(rlog_mu);
            fprintf(rlog, "R %d %d %zu %zu %zu\n", Plength, editDist, in_fw.size(), in_rc.size(), lis.size());
            for (const Match& m : in_fw) fprintf(rlog, "%lld %d %d\n", (long long)m.ref, m.query, m.len);
            for (const Match& m : in_rc) fprintf(rlog, "%lld %d %d\n", (long long)m.ref, m.query, m.len);
            for (const Lis& l : lis) {
                fprintf(rlog, "G %d %lld %lld %d %zu\n", l.fw ? 1 : 0, (long long)l.refBegin, (long long)l.refEnd, l.score, l.matches.size());
                for (const Match& m : l.matches) fprintf(rlog, "%lld %d %d\n", (long long)m.ref, m.query, m.len);
            }
            fflush(rlog);
        }
        if (sortByScore) std::sort(lis.begin(), lis.end(), [](const Lis& a, const Lis& b) { return a.score > b.score; });   // compIntervals :1114
    }
}

// ------------------------------------------------------------------------------------------ chains
// nextChain src/mapper.cpp:721-782: the greedy collinear chain around an anchor over the region's seeds (sorted by read
// position).  One sweep per direction; `tip` is the seed the chain currently ends with on that side and `far` the coordinates
// of its other end (the anchor's, which never move): a seed joins when it lies beyond the tip in read and text (up to 10
// bases of overlap) and the skew between its two gaps is within errorPercent of the chain's span.  Same arithmetic as
// mapcore::grow_chain (csrc/map_core.h), on the mapper's containers.
static int next_chain(const std::vector<Match>& matches, std::list<size_t>& chain, size_t anchorPos, double errorPercent,
                      std::vector<bool>& used, size_t qLength, int& coverage) {
    const Match& anchor = matches[anchorPos];
    chain.assign(1, anchorPos);
    coverage = anchor.len;
    int edits = 0;
    for (const int step : {+1, -1}) {
        Match tip = anchor;
        for (int64_t k = (int64_t)anchorPos + step; k >= 0 && k < (int64_t)matches.size(); k += step) {
            const Match& m = matches[(size_t)k];
            int64_t qGap, rGap, qSpan, rSpan;
            if (step > 0) {
                qGap = (int64_t)m.query - ((int64_t)tip.query + tip.len); rGap = m.ref - (tip.ref + tip.len);
                qSpan = (int64_t)m.query + m.len - tip.query;             rSpan = m.ref + m.len - tip.ref;
            } else {
                qGap = (int64_t)tip.query - ((int64_t)m.query + m.len);   rGap = tip.ref - (m.ref + m.len);
                qSpan = (int64_t)tip.query + tip.len - m.query;           rSpan = tip.ref + tip.len - m.ref;
            }
            const int64_t skew = std::llabs(qGap - rGap);
            if (qGap < MAX_OVERLAP || rGap < MAX_OVERLAP || skew > (int64_t)(errorPercent * std::max(qSpan, rSpan))) continue;
            if (step > 0) chain.push_back((size_t)k); else chain.push_front((size_t)k);
            used[(size_t)k] = true;
            edits += (int)std::max<int64_t>(1, skew);
            coverage += (int)(m.len + std::min<int64_t>(0, std::min(qGap, rGap)));
            tip = m;
        }
        // an uncovered read end costs one more edit
        if (step > 0) edits += (int)(tip.query + tip.len) < (int)qLength ? 1 : 0;
        else edits += tip.query > 0 ? 1 : 0;
    }
    return edits;
}

// leftExtension src/mapper.cpp:784-824
void Mapper::left_ext(const std::string& P, int& tmpQB, int64_t& tmpRB, int64_t chrStart, int& curScore, int bandSize, int editDist) {
    if (tmpQB > 0 && tmpRB > chrStart) {
        const char* S = idx_.S();
        int64_t boundLeft = std::max(tmpRB - tmpQB - (editDist), chrStart);
        int tLen = (int)(tmpRB - boundLeft);
        std::vector<uint8_t> target((size_t)std::max(tLen, 0)), query((size_t)tmpQB);
        for (int j = 0; j < tLen; ++j) target[j] = nt4((unsigned char)S[tmpRB - j - 1]);
        for (int j = 0; j < tmpQB; ++j) query[j] = nt4((unsigned char)P[tmpQB - j - 1]);
        alfa_extend_result r;
        hot_.submit_extend(ExtendReq{alfa_dp_task{query.data(), target.data(), tmpQB, tLen, std::min(bandSize, tmpQB), curScore}, &r});
        hot_.wait();
        curScore = r.score;
        if (opt_.global || r.gscore >= curScore + (opt_.mismatch - opt_.match) * 3) {
            curScore = r.gscore;
            tmpQB = 0;
            tmpRB = tmpRB - r.gtle;
        } else {
            tmpQB = tmpQB - r.qle;
            tmpRB = tmpRB - r.tle;
        }
    } else if (tmpQB > 0 && tmpRB == chrStart && opt_.global) {
        curScore += opt_.openGap + opt_.extendGap * tmpQB;
    }
}

// rightExtension src/mapper.cpp:826-867
void Mapper::right_ext(const std::string& P, int& tmpQE, int64_t& tmpRE, int64_t chrEnd, int& curScore, int bandSize, int editDist) {
    const int Plen = (int)P.length();
    if (tmpQE < Plen - 1 && tmpRE < chrEnd) {
        const char* S = idx_.S();
        int64_t boundRight = std::min((int64_t)(tmpRE + (Plen - 1 - tmpQE) + editDist), chrEnd);
        int tLen = (int)(boundRight - tmpRE);
        int qLen = Plen - 1 - tmpQE;
        std::vector<uint8_t> target((size_t)std::max(tLen, 0)), query((size_t)qLen);
        for (int j = 0; j < tLen; ++j) target[j] = nt4((unsigned char)S[tmpRE + 1 + j]);
        for (int j = 0; j < qLen; ++j) query[j] = nt4((unsigned char)P[tmpQE + 1 + j]);
        alfa_extend_result r;
        hot_.submit_extend(ExtendReq{alfa_dp_task{query.data(), target.data(), qLen, tLen, std::min(bandSize, qLen), curScore}, &r});
        hot_.wait();
        curScore = r.score;
        if (opt_.global || r.gscore >= curScore + (opt_.mismatch - opt_.match) * 3) {
            curScore = r.gscore;
            tmpQE = Plen - 1;
            tmpRE = tmpRE + r.gtle;
        } else {
            tmpQE = tmpQE + r.qle;
            tmpRE = tmpRE + r.tle;
        }
    } else if (tmpQE < Plen - 1 && tmpRE == chrEnd && opt_.global) {
        curScore += opt_.openGap + opt_.extendGap * (Plen - 1 - tmpQE);
    }
}

// extendAlignmentChain src/mapper.cpp:869-1081
int Mapper::extend_chain(const std::string& P, const std::vector<Match>& matches, int editDist, int minScore, int minCoverage,
                         int64_t chrStart, int64_t chrEnd, Read& read, bool fw, const Lis* first) {
    const char* S = idx_.S();
    int maxAlnScore = std::min(-1, minScore - 10);
    std::list<size_t> chain;
    std::vector<int64_t> seedOrder;
    seedOrder.reserve(matches.size());
    std::vector<bool> used(matches.size(), false);
    for (size_t i = 0; i < matches.size(); i++) seedOrder.push_back((int64_t)matches[i].len << 32 | (int64_t)i);
    std::sort(seedOrder.begin(), seedOrder.end());
    std::vector<Bounds> alnBounds;
    std::vector<int> alnScores;
    std::vector<int64_t> alnBegins;
    for (int i = (int)seedOrder.size() - 1; i >= 0; i--) {
        size_t anchorPos = (uint32_t)seedOrder[i];
        Match anchor = matches[anchorPos];
        bool extend = !used[anchorPos];
        size_t j = 0;
        while (j < alnBounds.size() && extend) {
            bool notContained = anchor.query < alnBounds[j].queryB || anchor.query + anchor.len >= alnBounds[j].queryE ||
                                anchor.ref < alnBounds[j].refB || anchor.ref + anchor.len >= alnBounds[j].refE;
            if (!notContained) {
                int qDist = anchor.query - alnBounds[j].queryB;
                int refDist = (int)(anchor.ref - alnBounds[j].refB);
                int max_gapL = std::min(opt_.fixedBandSize, (int)((std::min(qDist, refDist) * opt_.match + opt_.openGap) / opt_.extendGap + 1.));
                extend = std::max(qDist, refDist) >= max_gapL;
                if (extend) {
                    qDist = alnBounds[j].queryE - (anchor.query + anchor.len);
                    refDist = (int)(alnBounds[j].refE - (anchor.ref + anchor.len));
                    max_gapL = std::min(opt_.fixedBandSize, (int)((std::min(qDist, refDist) * opt_.match + opt_.openGap) / opt_.extendGap + 1.));
                    extend = std::max(qDist, refDist) >= max_gapL;
                }
            }
            j++;
        }
        if (!extend) continue;
        int curScore = 0;
        int minEdit;
        if (first && !first->chain0.empty() && i == (int)seedOrder.size() - 1 && (size_t)first->chain0Anchor == anchorPos) {
            // the first chain came with the region (regions_kernel): same members, same order, nextChain's outputs
            chain.clear();
            for (int32_t k : first->chain0) { chain.push_back((size_t)k); if ((size_t)k != anchorPos) used[(size_t)k] = true; }
            curScore = first->chain0Coverage;
            minEdit = first->chain0Edit;
        } else
            minEdit = next_chain(matches, chain, anchorPos, opt_.errorPercent, used, P.length(), curScore);
        if (!(minEdit <= editDist && (chain.size() > 1 || anchor.len >= minCoverage))) continue;
        std::list<size_t>::const_iterator it = chain.begin(), itEnd = chain.end();
        int bandSize = std::min(opt_.fixedBandSize, (int)(editDist - minEdit + 1));
        anchor = matches[*it];
        ++it;
        int tmpQB = anchor.query;
        int tmpQE = anchor.query + anchor.len - 1;
        int64_t tmpRB = anchor.ref;
        int64_t tmpRE = anchor.ref + anchor.len - 1;
        // inside the chain: the closed-form gap cases, and one ksw_nw per remaining gap.  The nw calls of one chain
        // are independent of each other (their scores are only summed), so they are submitted together.
        struct Gap { std::vector<uint8_t> q, t; int32_t score; };
        std::list<Gap> gaps;
        while (it != itEnd) {
            Match seed = matches[*it];
            int64_t qDist = seed.query - tmpQE;
            int64_t refDist = seed.ref - tmpRE;
            if (qDist < 0 || refDist < 0) {
                if (qDist >= refDist) qDist -= refDist;
                else refDist -= qDist;
            }
            if (qDist <= 0) curScore += (int)(opt_.openGap + opt_.extendGap * refDist);
            else if (refDist <= 0) curScore += (int)(opt_.openGap + opt_.extendGap * qDist);
            else if (qDist == 1) curScore += (int)(opt_.openGap + opt_.extendGap * (refDist - 1));
            else if (refDist == 1) curScore += (int)(opt_.openGap + opt_.extendGap * (qDist - 1));
            else if (refDist == 2 && qDist == 2) curScore += opt_.mismatch;
            else {
                int tLen = (int)(seed.ref - tmpRE - 1);
                int qLen = seed.query - tmpQE - 1;
                gaps.emplace_back();
                Gap& g = gaps.back();
                g.t.resize((size_t)tLen);
                g.q.resize((size_t)qLen);
                for (int k = 0; k < tLen; ++k) g.t[k] = nt4((unsigned char)S[tmpRE + 1 + k]);
                for (int k = 0; k < qLen; ++k) g.q[k] = nt4((unsigned char)P[tmpQE + 1 + k]);
                int innerBand = std::min(opt_.fixedBandSize, (int)(editDist - minEdit + std::abs(qLen - tLen) + 3));
                innerBand = std::max(innerBand, std::abs(qLen - tLen) + (int)(std::abs(qLen - tLen) * opt_.errorPercent));
                hot_.submit_nw(NwReq{alfa_dp_task{g.q.data(), g.t.data(), qLen, tLen, innerBand, 0}, &g.score});
            }
            tmpRE = seed.ref + seed.len - 1;
            tmpQE = seed.query + seed.len - 1;
            ++it;
        }
        if (!gaps.empty()) {
            hot_.wait();
            for (const Gap& g : gaps) curScore += g.score;
        }
        if (tmpQB <= (int)(P.length() - 1) - tmpQE) {
            left_ext(P, tmpQB, tmpRB, chrStart, curScore, bandSize, editDist - minEdit + 1);
            right_ext(P, tmpQE, tmpRE, chrEnd, curScore, bandSize, editDist);
        } else {
            right_ext(P, tmpQE, tmpRE, chrEnd, curScore, bandSize, editDist);
            left_ext(P, tmpQB, tmpRB, chrStart, curScore, bandSize, editDist - minEdit + 1);
        }
        maxAlnScore = std::max(maxAlnScore, curScore);
        if (curScore >= minScore) {
            bool equal = false;
            for (size_t k = 0; k < alnScores.size(); k++)
                if (alnScores[k] >= curScore && std::llabs(alnBegins[k] - tmpRB) < editDist) equal = true;
            if (!equal) {
                Alignment* a = new Alignment();
                a->refB = tmpRB; a->refE = tmpRE; a->qB = tmpQB; a->qE = tmpQE;
                a->alignmentScore = curScore;
                if (!fw) a->flag |= 1u << 4;
                a->refLength = (int)(tmpRE - tmpRB + 1);
                for (size_t idx : chain) a->chain.push_back(matches[idx]);
                read.alignments.push_back(a);
                alnScores.push_back(curScore);
                alnBegins.push_back(tmpRB);
            }
        }
        alnBounds.push_back(Bounds{tmpRB, tmpRE, tmpQB, tmpQE});
    }
    return maxAlnScore >= minScore ? maxAlnScore : -1;
}

// hasUnique src/mapper.cpp:1090-1112 with an ordered map (read position -> seed length) instead of the kbtree
typedef std::map<int64_t, int64_t> QCovTree;
template <class L>
static bool has_unique(const L& l, QCovTree& tree) {
    bool unique = false;
    for (const Match& s : l.matches) {
        const int64_t q = s.query, len = s.len;
        if (!tree.empty()) {
            auto ub = tree.upper_bound(q);                 // kb_interval: lower = largest key <= q
            bool hasLower = ub != tree.begin();
            auto lower = hasLower ? std::prev(ub) : tree.end();
            if (!hasLower || lower->first + lower->second < q + len) {
                if (hasLower && lower->first == q) tree.erase(lower);
                tree[q] = len;
                unique = true;
            }
        } else {
            tree[q] = len;
            unique = true;
        }
    }
    return unique;
}

// processLisIntervals src/mapper.cpp:1118-1204
void Mapper::process_lis(Read& read, const std::vector<Lis>& lis, int editDist, int maxAlnCount, int minCoverage, int minScore) {
    size_t lisIndex = 0;
    int trialF = 0, trialR = 0;
    QCovTree treeF, treeR;
    int maxAlnScore = 0;
    while ((int)read.alignments.size() < maxAlnCount && lisIndex < lis.size()) {
        const Lis& cluster = lis[lisIndex];
        bool unique = has_unique(cluster, cluster.fw ? treeF : treeR);
        bool extend = unique;
        int trial = cluster.fw ? trialF : trialR;
        bool criterium = (trial < opt_.maxTrial || extend) && (cluster.matches.size() > 1 || cluster.score >= minCoverage);
        if (criterium) {
            int64_t chrStart, chrEnd;
            idx_.chrom_bounds(cluster.matches[0].ref, chrStart, chrEnd);
            int ret = extend_chain(cluster.fw ? read.sequence : read.rcSequence, cluster.matches, editDist, minScore, minCoverage, chrStart,
                                   chrEnd, read, cluster.fw, &cluster);
            if (cluster.fw) trialF++; else trialR++;
            if (ret >= 0 && (!opt_.maxTrialBest || maxAlnScore < ret)) {
                if (cluster.fw) trialF = 0; else trialR = 0;
            }
            maxAlnScore = std::max(maxAlnScore, ret);
        }
        lisIndex++;
    }
}

// ---- the trial loop through csrc/map_core.h (checker path, --core-trials): the same source the mapping kernel is compiled
// from runs here with one "lane", its DP calls going through the scheduler like the mapper's own
namespace {
struct CoreEnv {
    const RefIndex& idx; HotSched& hot; const Read& read; const std::vector<Mapper_Lis_view>* lis;
    int lane() const { return 0; }
    int lanes() const { return 1; }
    void sync() const {}
    bool load_region(mapcore::Work& W, int k, int& n, bool& fw, int& score) {
        const Mapper_Lis_view& l = (*lis)[(size_t)k];
        if (l.matches->size() > (size_t)mapcore::MEMBER_CAP) return false;
        n = (int)l.matches->size(); fw = l.fw; score = l.score;
        for (int i = 0; i < n; ++i) { W.mref[i] = (*l.matches)[i].ref; W.mq[i] = (*l.matches)[i].query; W.ml[i] = (*l.matches)[i].len; }
        return true;
    }
    void chrom_bounds(int64_t pos, int64_t& cs, int64_t& ce) const { idx.chrom_bounds(pos, cs, ce); }
    bool nw(bool fw, int qFrom, int qLen, int64_t tFrom, int tLen, int w, int& score) {
        const std::string& P = fw ? read.sequence : read.rcSequence;
        const char* S = idx.S();
        std::vector<uint8_t> q((size_t)qLen), t((size_t)tLen);
        for (int k = 0; k < qLen; ++k) q[k] = nt4((unsigned char)P[qFrom + k]);
        for (int k = 0; k < tLen; ++k) t[k] = nt4((unsigned char)S[tFrom + k]);
        int32_t s = 0;
        hot.submit_nw(NwReq{alfa_dp_task{q.data(), t.data(), qLen, tLen, w, 0}, &s});
        hot.wait();
        score = s;
        return true;
    }
    bool extend(bool fw, bool left, int qAnchor, int qLen, int64_t tAnchor, int tLen, int w, int h0, mapcore::ExtRes& r) {
        const std::string& P = fw ? read.sequence : read.rcSequence;
        const char* S = idx.S();
        std::vector<uint8_t> q((size_t)qLen), t((size_t)std::max(tLen, 0));
        for (int j = 0; j < qLen; ++j) q[j] = nt4((unsigned char)(left ? P[qAnchor - j - 1] : P[qAnchor + 1 + j]));
        for (int j = 0; j < tLen; ++j) t[j] = nt4((unsigned char)(left ? S[tAnchor - j - 1] : S[tAnchor + 1 + j]));
        alfa_extend_result e;
        hot.submit_extend(ExtendReq{alfa_dp_task{q.data(), t.data(), qLen, tLen, w, h0}, &e});
        hot.wait();
        r.score = e.score; r.qle = e.qle; r.tle = e.tle; r.gtle = e.gtle; r.gscore = e.gscore; r.max_off = e.max_off;
        return true;
    }
    // ksw_global through the scheduler; its BAM operations expanded to extended operations (recoverCIGAR4 src/mapper.cpp:80-117:
    // every M run split into maximal runs of equal / unequal nt4 codes), kept in this object
    std::vector<uint32_t> cg, xs;
    int global_cigar(mapcore::Work& W, bool fw, int qFrom, int qLen, int64_t tFrom, int tLen, int band, int& first) {
        const uint8_t* rp = fw ? read.readP.data() : read.rcReadP.data();
        const char* S = idx.S();
        std::vector<uint8_t> t((size_t)std::max(tLen, 0));
        for (int k = 0; k < tLen; ++k) t[k] = nt4((unsigned char)S[tFrom + k]);
        int32_t score = 0;
        cg.clear();
        hot.submit_global(GlobalReq{alfa_dp_task{rp + qFrom, t.data(), qLen, tLen, band, 0}, &score, &cg});
        hot.wait();
        xs.clear();
        auto put = [&](uint32_t code, uint32_t len) {
            if (!xs.empty() && (xs.back() & 0xf) == code) xs.back() += len << 4;
            else xs.push_back(len << 4 | code);
        };
        int x = 0, y = 0;
        for (const uint32_t c : cg) {
            const int op = (int)(c & 0xf), len = (int)(c >> 4);
            if (op == 0) {
                for (int k = 0; k < len; ++k) put(rp[qFrom + x + k] == t[(size_t)(y + k)] ? mapcore::XOP_EQ : mapcore::XOP_X, 1);
                x += len; y += len;
            } else if (op == 1) { put(mapcore::XOP_INS, (uint32_t)len); x += len; }
            else if (op == 2) { put(mapcore::XOP_DEL, (uint32_t)len); y += len; }
            else return -1;
        }
        first = 0;
        W.bamp = xs.data();
        return (int)xs.size();
    }
    uint32_t xop(const mapcore::Work& W, int i) const { return W.bamp[i]; }
    uint64_t equal_mask(bool fw, int qpos, int64_t tpos, int n) const {
        const uint8_t* rp = fw ? read.readP.data() : read.rcReadP.data();
        const char* S = idx.S();
        uint64_t m = 0;
        for (int i = 0; i < n; ++i) m |= (uint64_t)(rp[qpos + i] == nt4((unsigned char)S[tpos + i])) << i;
        return m;
    }
};
}  // namespace

std::atomic<long> g_core_done{0}, g_core_fallback{0};
std::atomic<long> g_map_us[5];      // whole-read device calls, microseconds over all workers: describe, the call, adopt; [3] counts the
                                    // alignments whose CIGAR came with them

mapcore::Options core_options(const AlignOpt& o) {
    mapcore::Options O;
    O.match = o.match; O.mismatch = o.mismatch; O.openGap = o.openGap; O.extendGap = o.extendGap;
    O.global = o.global ? 1 : 0; O.fixedBandSize = o.fixedBandSize; O.maxTrial = o.maxTrial; O.maxTrialBest = o.maxTrialBest ? 1 : 0;
    O.maxAlnCount = o.maxAlnCount; O.rescue = o.rescue ? 1 : 0; O.errorPercent = o.errorPercent;
    O.cigarCount = o.fullDP ? 0 : o.alignmentCount; O.rescueOnly = 0;       // --full-dp recomputes whole alignments: no chain CIGARs
    return O;
}

// what depends on the read length alone, in the reference's arithmetic (src/mapper.cpp:1224,1280-1287)
mapcore::ReadParams core_read_params(const AlignOpt& o, int Plength) {
    mapcore::ReadParams R;
    R.plen = Plength;
    R.editDist = (int)(o.errorPercent * Plength) + 1;
    R.minScore = (int)(Plength * o.match + (o.mismatch - o.match) * R.editDist);
    R.minCoverage = (int)(Plength * o.minCoverage);
    R.minScore2 = std::max(0, (int)(Plength * o.match + (o.mismatch - o.match) * 0.2 * Plength));
    R.minCoverage2 = (int)(Plength * 0.01);
    return R;
}

// true: the read's alignments are in place (what unpaired_match's two passes would have produced); false: not handled
bool Mapper::core_trials(Read& read, const std::vector<Lis>& lis) {
    std::vector<Mapper_Lis_view> view;
    view.reserve(lis.size());
    for (const Lis& l : lis) view.push_back(Mapper_Lis_view{&l.matches, l.fw, l.score});
    CoreEnv env{idx_, hot_, read, &view};
    std::unique_ptr<mapcore::Work> W(new mapcore::Work());
    std::vector<mapcore::CigOp> opbuf((size_t)std::max<size_t>(mapcore::OPS_CAP, read.sequence.length() + 64));
    W->ops = opbuf.data(); W->opsCap = (int32_t)opbuf.size();
    std::vector<mapcore::ChainSeed> chainbuf((size_t)mapcore::CHAINOUT_CAP);
    W->chain = chainbuf.data();
    if (lis.size() > (size_t)mapcore::REGION_CAP2) return false;
    for (size_t k = 0; k < lis.size(); ++k) W->rscore[k] = lis[k].score;
    const mapcore::Options O = core_options(opt_);
    const mapcore::ReadParams R = core_read_params(opt_, (int)read.sequence.length());
    if (mapcore::map_read(env, *W, O, R, (int)lis.size()) != mapcore::DONE) return false;
    for (int k = 0; k < W->nAln; ++k) {
        const mapcore::AlnOut& o = W->alns[k];
        Alignment* a = new Alignment();
        a->refB = o.refB; a->refE = o.refE; a->qB = o.qB; a->qE = o.qE; a->alignmentScore = o.score;
        if (!o.fw) a->flag |= 1u << 4;
        a->refLength = (int)(o.refE - o.refB + 1);
        a->chain.resize((size_t)o.chainCount);
        for (int c = 0; c < o.chainCount; ++c) a->chain[(size_t)c] = Match{W->chain[o.chainFirst + c].ref, W->chain[o.chainFirst + c].q, W->chain[o.chainFirst + c].len};
        if (o.opCount >= 0) { a->ops.assign(W->ops + o.opFirst, W->ops + o.opFirst + o.opCount); a->haveOps = true; a->opsEdit = o.editDist; }
        read.alignments.push_back(a);
    }
    return true;
}

// unpairedMatch src/mapper.cpp:1269-1302
void Mapper::unpaired_match(Read& read) {
    std::vector<Lis> lis;
    if (opt_.coreTrials && read.alignments.empty()) {
        cand_regions(read, lis, opt_.errorPercent, false);           // regions in creation order: map_core sorts them itself
        if (core_trials(read, lis)) { g_core_done++; return; }
        g_core_fallback++;
        for (Alignment* a : read.alignments) delete a;
        read.alignments.clear();
        lis.clear();
    }
    cand_regions(read, lis, opt_.errorPercent);
    if (lis.empty()) return;
    if (!read.alignments.empty()) {
        // filterLISwithAlignments :1206-1218
        for (size_t i = 0; i < read.alignments.size(); i++) {
            Alignment* a = read.alignments[i];
            for (auto it = lis.begin(); it != lis.end();) {
                if (a->rc() != it->fw && a->refE >= it->refBegin && a->refB <= it->refEnd) it = lis.erase(it);
                else ++it;
            }
        }
    }
    const int Plength = (int)read.sequence.length();
    const int editDist = (int)(opt_.errorPercent * Plength) + 1;
    int minScore = (int)(Plength * opt_.match + (opt_.mismatch - opt_.match) * editDist);
    process_lis(read, lis, editDist, opt_.maxAlnCount, (int)(Plength * opt_.minCoverage), minScore);
    if (read.alignments.empty() && opt_.rescue) {
        minScore = std::max(0, (int)(Plength * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * Plength));
        int minCoverage = (int)(Plength * 0.01);
        process_lis(read, lis, editDist, 20, minCoverage, minScore);
    }
    if (read.alignments.empty() && opt_.rescue) {
        lis.clear();
        cand_regions(read, lis, 2.0 * opt_.errorPercent);
        if (!lis.empty()) {
            minScore = std::max(0, (int)(Plength * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * Plength));
            int minCoverage = (int)(Plength * 0.01);
            process_lis(read, lis, editDist, 20, minCoverage, minScore);
        }
    }
}

// alignment_t::setLocalPos src/mapper.h:104-112
void Mapper::set_local_pos(Alignment& a) {
    if (a.rname.empty() || a.rname == "*") {
        int64_t seq, pos;
        idx_.from_set(a.refB + 1, seq, pos);
        a.pos = pos;
        a.rname = idx_.descr(seq);
    }
}

// read_t::postprocess src/mapper.cpp:322-379 with constructCIGAR :172-271 and fullDP :303-320.
// All ksw_global calls of a read (one per chain gap + head + tail of every reported alignment) are independent:
// they are submitted together and the strings are assembled afterwards in the reference's order.
bool Mapper::postprocess(Read& read, bool paired, bool mayWait) {
    if (read.alignments.empty()) return true;
    auto& al = read.alignments;
    if (!paired) std::sort(al.begin(), al.end(), cmp_aln);
    else std::sort(al.begin(), al.end(), cmp_aln_paired);
    int maxDP = std::min((int)al.size(), opt_.alignmentCount);
    if (opt_.fullDP) maxDP = std::min((int)al.size(), std::max(4, opt_.alignmentCount));
    if (!mayWait) {
        // not on a fiber: only when every alignment that gets a CIGAR brought its operations along
        for (int j = 0; j < maxDP; j++)
            if (opt_.fullDP || !al[j]->haveOps) return false;
    }
    const char* S = idx_.S();
    const int Plen = (int)read.sequence.length();
    struct Piece { int kind; std::string lit; int dp; int editAdd; };   // kind 0: literal text (+edit), 1: DP result #dp
    struct DpJob { std::vector<uint32_t> cigar; int32_t score; const uint8_t* q; const uint8_t* t; };
    struct Plan { std::vector<uint8_t> target; std::vector<Piece> pieces; bool full = false; int fullJob = -1; bool fromOps = false; };
    std::vector<Plan> plans((size_t)maxDP);
    std::deque<DpJob> jobs;               // references stay valid across push_back (the scheduler holds pointers into the jobs)
    std::vector<DpJob*> jobv;
    auto add_job = [&](const uint8_t* q, int qLen, const uint8_t* t, int tLen, int band) {
        jobs.emplace_back();
        DpJob& j = jobs.back();
        j.q = q; j.t = t;
        hot_.submit_global(GlobalReq{alfa_dp_task{q, t, qLen, tLen, band, 0}, &j.score, &j.cigar});
        jobv.push_back(&j);
        return (int)jobv.size() - 1;
    };
    for (int j = 0; j < maxDP; j++) {
        Alignment& a = *al[j];
        set_local_pos(a);
        const uint8_t* readP = a.rc() ? read.rcReadP.data() : read.readP.data();
        Plan& pl = plans[j];
        if (a.haveOps && !opt_.fullDP) { pl.fromOps = true; continue; }       // constructCIGAR's result came with the alignment
        int qLen = a.qE - a.qB + 1;
        int refLen = (int)(a.refE - a.refB + 1);
        pl.target.resize((size_t)std::max(refLen, 0));
        for (int i = 0; i < refLen; ++i) pl.target[i] = nt4((unsigned char)S[a.refB + i]);
        const uint8_t* target = pl.target.data();
        if (!a.chain.empty() && !opt_.fullDP) {
            auto lit = [&](const std::string& s, int e) { pl.pieces.push_back(Piece{0, s, -1, e}); };
            auto num = [](long v, char c) { Txt o; o << v << c; return std::move(o.s); };
            if (a.qB) lit(num(a.qB, 'S'), 0);
            const std::vector<Match>& ch = a.chain;
            if (ch[0].query > a.qB) {
                qLen = ch[0].query - a.qB;
                refLen = (int)(ch[0].ref - a.refB);
                int band = band_estimate(opt_.fixedBandSize, qLen, refLen, opt_.match, opt_.openGap, opt_.extendGap, a.alignmentScore);
                pl.pieces.push_back(Piece{1, "", add_job(readP + a.qB, qLen, target, refLen, band), 0});
            }
            lit(num(ch[0].len, '='), 0);
            for (size_t i = 1; i < ch.size(); i++) {
                int qDist = ch[i].query - (ch[i - 1].query + ch[i - 1].len - 1);
                int refDist = (int)(ch[i].ref - (ch[i - 1].ref + ch[i - 1].len - 1));
                if (qDist < 0 || refDist < 0) {
                    if (qDist >= refDist) qDist -= refDist;
                    else refDist -= qDist;
                }
                if (qDist <= 0) lit(num(refDist, 'D') + num(ch[i].len - 1 + qDist, '='), refDist);
                else if (refDist <= 0) lit(num(qDist, 'I') + num(ch[i].len - 1 + refDist, '='), qDist);
                else if (qDist == 1) lit(num(refDist - 1, 'D') + num(ch[i].len, '='), refDist - 1);
                else if (refDist == 1) lit(num(qDist - 1, 'I') + num(ch[i].len, '='), qDist - 1);
                else if (refDist == 2 && qDist == 2) lit(num(1, 'X') + num(ch[i].len, '='), 1);
                else {
                    qLen = ch[i].query - 1 - (ch[i - 1].query + ch[i - 1].len) + 1;
                    refLen = (int)(ch[i].ref - 1 - (ch[i - 1].ref + ch[i - 1].len) + 1);
                    int band = band_estimate(opt_.fixedBandSize, qLen, refLen, opt_.match, opt_.openGap, opt_.extendGap, a.alignmentScore);
                    const uint8_t* q = readP + ch[i - 1].query + ch[i - 1].len;
                    const uint8_t* t = target + (ch[i - 1].ref + ch[i - 1].len - a.refB);
                    pl.pieces.push_back(Piece{1, "", add_job(q, qLen, t, refLen, band), 0});
                    lit(num(ch[i].len, '='), 0);
                }
            }
            const Match& last = ch.back();
            if (last.query + last.len - 1 < a.qE) {
                qLen = a.qE - (last.query + last.len) + 1;
                refLen = (int)(a.refE - (last.ref + last.len) + 1);
                int band = band_estimate(opt_.fixedBandSize, qLen, refLen, opt_.match, opt_.openGap, opt_.extendGap, a.alignmentScore);
                pl.pieces.push_back(Piece{1, "", add_job(readP + last.query + last.len, qLen, target + (last.ref + last.len - a.refB), refLen, band), 0});
            }
            if (Plen - 1 - a.qE) lit(num(Plen - 1 - a.qE, 'S'), 0);
        } else {
            pl.full = true;
            int band = band_estimate(opt_.fixedBandSize, qLen, refLen, opt_.match, opt_.openGap, opt_.extendGap, a.alignmentScore);
            pl.fullJob = add_job(readP + a.qB, qLen, target, refLen, band);
        }
    }
    if (!jobv.empty()) hot_.wait();
    for (int j = 0; j < maxDP; j++) {
        Alignment& a = *al[j];
        Plan& pl = plans[j];
        if (pl.fromOps) {
            g_map_us[3]++;
            Txt sNM;
            for (const mapcore::CigOp& o : a.ops) sNM << o.len << (char)o.code;
            a.editDist = a.opsEdit;
            a.NMtag = sNM.str();
            a.cigar = cigar3_from4(a.NMtag);
        } else if (!pl.full) {
            Txt sNM;
            a.editDist = 0;
            for (const Piece& p : pl.pieces) {
                if (p.kind == 0) { sNM << p.lit; a.editDist += p.editAdd; }
                else { DpJob& d = *jobv[p.dp]; a.editDist += cigar4(d.cigar, d.q, d.t, sNM, nullptr); }
            }
            a.NMtag = sNM.str();
            a.cigar = cigar3_from4(a.NMtag);
        } else {
            // fullDP + setFieldsKsw src/mapper.cpp:120-170,303-320
            DpJob& d = *jobv[pl.fullJob];
            a.alignmentScore = d.score;
            Txt sNM, sCig;
            int beginGap = a.qB, endGap = Plen - 1 - a.qE;
            if (beginGap) { sCig << beginGap << "S"; sNM << beginGap << "S"; }
            a.editDist = cigar4(d.cigar, d.q, d.t, sNM, &sCig);
            if (endGap) { sCig << endGap << "S"; sNM << endGap << "S"; }
            a.cigar = sCig.str();
            a.NMtag = sNM.str();
        }
    }
    if (!paired) std::sort(al.begin(), al.end(), cmp_aln);
    else std::sort(al.begin(), al.end(), cmp_aln_paired);
    int maxScore = al[0]->alignmentScore + al[0]->mate.alignmentScore;
    int secBest = al.size() > 1 ? al[1]->alignmentScore + al[1]->mate.alignmentScore : opt_.mismatch * (int)read.sequence.length();
    int mapq = (secBest == (int)(opt_.mismatch * (int)read.sequence.length())) ? MAX_MAPQ
                                                                                  : (int)(250 * ((double)(maxScore - secBest) / maxScore) + 0.499);
    mapq = maxScore == secBest ? 0 : mapq;
    mapq = std::min(std::max(0, mapq), MAX_MAPQ);
    for (int j = 0; j < std::min((int)al.size(), opt_.alignmentCount); j++) {
        if (al[j]->alignmentScore + al[j]->mate.alignmentScore == maxScore) al[j]->mapq = mapq;
        else al[j]->flag |= 1u << 8;
    }
    return true;
}

void Mapper::map_unpaired(Read& read) {
    read.init();
    unpaired_match(read);
    postprocess(read, false);
}

// ---- whole reads through HotOps::map_reads_batch (alfa_map_reads_batch)
bool Mapper::device_map_usable() const {
    // both strands are seeded and tried there; the checker path and the region-only path are alternatives to it
    return opt_.deviceMap && hot_.ops().map_reads_batch && !opt_.noFW && !opt_.noRC && !opt_.coreTrials && !opt_.deviceRegions;
}

// An optional cap on the whole-read device calls of one GPU that are in flight at a time (ALFA_MAP_LANES=n; default 0 = no cap).
// A call is a chain of a dozen launches, copies and three host waits; when all workers call at once their chains interleave on
// the GPU's few hardware queues and every call waits behind everybody else's kernels.  What cured that was taking the barrier
// between batches out of the pipeline (mapper_cli.cpp: the workers drift apart); with that in place a cap only adds waiting
// (2 000 000 short reads, 16 workers: map phase 0.63 s uncapped, 0.62-0.65 s with 2-8 lanes, 0.95 s with 1), so it stays off.
namespace {
struct Lanes {
    std::mutex mu;
    std::condition_variable cv;
    int free_ = 0;
    bool unlimited = false;
};
Lanes& lanes_of(const void* device_index) {
    static std::mutex mu;
    static std::map<const void*, std::unique_ptr<Lanes>> all;
    std::lock_guard<std::mutex> g(mu);
    std::unique_ptr<Lanes>& l = all[device_index];
    if (!l) {
        l.reset(new Lanes);
        const char* e = getenv("ALFA_MAP_LANES");
        const int n = e ? atoi(e) : 0;
        l->free_ = n;
        l->unlimited = n <= 0;
    }
    return *l;
}
struct LaneHold {
    Lanes& l;
    explicit LaneHold(Lanes& x) : l(x) {
        if (l.unlimited) return;
        std::unique_lock<std::mutex> g(l.mu);
        l.cv.wait(g, [&] { return l.free_ > 0; });
        --l.free_;
    }
    ~LaneHold() {
        if (l.unlimited) return;
        { std::lock_guard<std::mutex> g(l.mu); ++l.free_; }
        l.cv.notify_one();
    }
};
}  // namespace

void Mapper::map_options(alfa_map_options& mo) const {
    memset(&mo, 0, sizeof(mo));
    mo.match = opt_.match; mo.mismatch = opt_.mismatch; mo.open_gap = opt_.openGap; mo.extend_gap = opt_.extendGap;
    mo.global = opt_.global ? 1 : 0; mo.fixed_band = opt_.fixedBandSize; mo.max_trial = opt_.maxTrial; mo.max_trial_best = opt_.maxTrialBest ? 1 : 0;
    mo.max_aln_count = opt_.maxAlnCount; mo.rescue = opt_.rescue ? 1 : 0;
    mo.mem_type = opt_.memType; mo.max_seed_candidates = opt_.maxSeedCandidates; mo.max_smem_start = opt_.maxSmemMemStart;
    mo.max_region_mems = opt_.maxRegionMems; mo.min_mem_reg_length = opt_.minMemRegLength; mo.error_percent = opt_.errorPercent;
    mo.cigar_alignments = opt_.fullDP ? 0 : opt_.alignmentCount;
}

void Mapper::describe(const char* P, const char* Prc, int Plength, alfa_map_read& x, double errorPercent, int memStage) const {
    const int K = (int)idx_.K();
    const mapcore::ReadParams R = core_read_params(opt_, Plength);          // the trial loop's thresholds: always from alnOptions.errorPercent (:1280)
    // calculateCandRegions src/mapper.cpp:1224-1231 with its errorPercent parameter; sparseSkip: calculateSeedsBothStrands :533-536
    const int editDist = (int)(errorPercent * Plength) + 1;
    int min_len = opt_.minMemLength;
    if (!opt_.fixedMinLength && Plength >= 1000) min_len += (20 * ((int)(Plength / 500))) / ((int)100 * errorPercent);
    if (opt_.fixedMinLength && Plength / editDist > min_len) min_len = Plength / editDist;
    int sparseSkip = K >= 4 ? std::max((min_len - 10) / K, 1) : std::max((min_len - 12) / K, 1);
    if (opt_.noSparseQ) sparseSkip = 1;
    if (memStage) {                                  // MEMRESCUE :542-555 (the call also raises max_seed_candidates to 100000)
        const int minLength = std::max(MIN_SEED_LENGTH, K + 10);
        min_len = memStage == 1 ? (minLength + min_len) / 2 : minLength;
        sparseSkip = 1;
    }
    x.P = P; x.Prc = Prc; x.plen = Plength;
    x.min_len = min_len; x.sparse_q = sparseSkip;
    x.edit_dist = R.editDist; x.min_score = R.minScore; x.min_coverage = R.minCoverage; x.min_score2 = R.minScore2; x.min_coverage2 = R.minCoverage2;
    x.region_edit_dist = editDist; x.reserved = 0;
}

void Mapper::map_call(const alfa_map_options& mo, int n, MapCallResult& R) {
    R.st.assign((size_t)n, 0);
    R.aoff.assign((size_t)n + 1, 0); R.soff.assign((size_t)n + 1, 0); R.ooff.assign((size_t)n + 1, 0);
    R.free_fn = hot_.ops().free_fn;
    static thread_local Lanes* lanes = nullptr;
    if (!lanes) lanes = &lanes_of(hot_.ops().index);
    const auto tw0 = std::chrono::steady_clock::now();
    int rc;
    std::chrono::steady_clock::time_point t0;
    {
        LaneHold hold(*lanes);
        t0 = std::chrono::steady_clock::now();
        rc = hot_.ops().map_reads_batch(hot_.ops().index, &mo, &hot_.dp_params(), rd_.data(), n, R.st.data(), R.aoff.data(), &R.al, R.soff.data(), &R.sd,
                                        R.ooff.data(), &R.op);
    }
    const auto t1 = std::chrono::steady_clock::now();
    g_map_us[4] += (long)std::chrono::duration<double, std::micro>(t0 - tw0).count();
    hot_.add_hot_seconds(std::chrono::duration<double>(t1 - t0).count());
    g_map_us[1] += (long)std::chrono::duration<double, std::micro>(t1 - t0).count();
    if (rc) {
        fprintf(stderr, "ERROR: alfa_map_reads_batch failed (%d): %s\n", rc, hot_.ops().last_error ? hot_.ops().last_error() : "");
        exit(1);
    }
}

// the call's alignments of read i as alignment_t objects, as unpairedMatch would have left them
void Mapper::adopt(Read& r, const MapCallResult& R, int i) {
    for (int64_t k = R.aoff[(size_t)i]; k < R.aoff[(size_t)i + 1]; ++k) {
        const alfa_map_alignment& o = R.al[k];
        Alignment* a = new Alignment();
        a->refB = o.ref_begin; a->refE = o.ref_end; a->qB = o.q_begin; a->qE = o.q_end; a->alignmentScore = o.score;
        if (!o.forward) a->flag |= 1u << 4;
        a->refLength = (int)(o.ref_end - o.ref_begin + 1);
        static_assert(sizeof(Match) == sizeof(alfa_match), "Match == alfa_match");
        const Match* s0 = (const Match*)(R.sd + R.soff[(size_t)i] + o.chain_first);
        a->chain.assign(s0, s0 + o.chain_count);
        if (o.op_count >= 0) {
            static_assert(sizeof(mapcore::CigOp) == sizeof(alfa_map_cigar_op), "CigOp == alfa_map_cigar_op");
            const mapcore::CigOp* p0 = (const mapcore::CigOp*)(R.op + R.ooff[(size_t)i] + o.op_first);
            a->ops.assign(p0, p0 + o.op_count);
            a->haveOps = true;
            a->opsEdit = o.edit_dist;
        }
        r.alignments.push_back(a);
    }
}

void Mapper::map_batch(Read* const* reads, int n, std::vector<uint8_t>& done) {
    done.assign((size_t)n, 0);
    if (n <= 0) return;
    alfa_map_options mo;
    map_options(mo);
    const auto tm0 = std::chrono::steady_clock::now();
    rd_.resize((size_t)n);
    for (int i = 0; i < n; ++i)
        describe(reads[i]->sequence.data(), reads[i]->rcSequence.data(), (int)reads[i]->sequence.length(), rd_[(size_t)i], opt_.errorPercent);
    g_map_us[0] += (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tm0).count();
    MapCallResult R;
    map_call(mo, n, R);
    const auto t1 = std::chrono::steady_clock::now();
    long nd = 0;
    for (int i = 0; i < n; ++i) {
        if (R.st[(size_t)i] != 0) continue;          // (1: a capacity; 2, 3: the rescue rounds -- unpaired_match runs them for these reads)
        adopt(*reads[i], R, i);
        done[(size_t)i] = 1;
        ++nd;
    }
    g_core_done += nd;
    g_core_fallback += n - nd;
    g_map_us[2] += (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
}

// The two strands of a read sixteen residues at a time: lower case, and the complement of a / c / g / t / n out of a table
// indexed by the low four bits of the letter (1, 3, 7, 4, 14: all different), reversed into place.  Stops in front of the first
// block that holds anything else (IUPAC codes, other bytes) or does not fill sixteen bytes; returns how far it came -- the
// caller's byte loop, which is the definition, does the rest.
__attribute__((target("ssse3"))) static int strands_ssse3(const unsigned char* s, int len, char* fw, char* rc) {
    const __m128i loA = _mm_set1_epi8('A' - 1), hiZ = _mm_set1_epi8('Z' + 1), bit5 = _mm_set1_epi8(32), nib = _mm_set1_epi8(15);
    //                                 0    1    2    3    4    5    6    7    8    9    10   11   12   13   14   15
    const __m128i comp = _mm_setr_epi8(0, 't', 0, 'g', 'a', 0, 0, 'c', 0, 0, 0, 0, 0, 0, 'n', 0);
    const __m128i self = _mm_setr_epi8(1, 'a', 0, 'c', 't', 0, 0, 'g', 0, 0, 0, 0, 0, 0, 'n', 0);      // (no byte equals the entry of its own low bits elsewhere)
    const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    int k = 0;
    for (; k + 16 <= len; k += 16) {
        const __m128i c = _mm_loadu_si128((const __m128i*)(s + k));
        const __m128i upper = _mm_and_si128(_mm_cmpgt_epi8(c, loA), _mm_cmpgt_epi8(hiZ, c));
        const __m128i lower = _mm_add_epi8(c, _mm_and_si128(upper, bit5));
        const __m128i idx = _mm_and_si128(lower, nib);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_shuffle_epi8(self, idx), lower)) != 0xffff) break;
        _mm_storeu_si128((__m128i*)(fw + k), lower);
        _mm_storeu_si128((__m128i*)(rc + len - 16 - k), _mm_shuffle_epi8(_mm_shuffle_epi8(comp, idx), rev));
    }
    return k;
}

void Mapper::map_raw(const RawRead* raw, int n, RawStrands& S, std::vector<std::unique_ptr<MapCallResult>>& calls, std::vector<RawResult>& res) {
    res.assign((size_t)n, RawResult());
    if (n <= 0) return;
    const auto tm0 = std::chrono::steady_clock::now();
    S.off.resize((size_t)n);
    S.len.resize((size_t)n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) { S.off[(size_t)i] = total; S.len[(size_t)i] = raw[i].len; total += 2 * (size_t)raw[i].len; }
    if (S.bytes.size() < total) S.bytes.resize(total);
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    for (int i = 0; i < n; ++i) {
        const int len = raw[i].len;
        char* fw = &S.bytes[S.off[(size_t)i]];
        char* rc = fw + len;
        const unsigned char* s = (const unsigned char*)raw[i].seq;
        int k = ssse3 ? strands_ssse3(s, len, fw, rc) : 0;
        for (; k < len; ++k) {      // tolower in the "C" locale (the reader, src/fasta.cpp:147), then read_t::init's reverse complement
            const unsigned char c = (unsigned char)(s[k] + ((unsigned char)(s[k] - 'A') < 26u ? 32 : 0));
            fw[k] = (char)c;
            rc[len - 1 - k] = RC.t[c];
        }
    }
    g_map_us[0] += (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tm0).count();
    // The rounds of unpairedMatch (src/mapper.cpp:1269-1302), each a call for the reads that reach it:
    //   A      calculateCandRegions(errorPercent), both passes
    //   M1, M2 the same after the seed rescue's first / second round (no seed at all in the round before, :542-555)
    //   R2     RESCUE2: calculateCandRegions(2 x errorPercent), the looser pass only (no alignment in A / M1 / M2, :1290-1300)
    //   R2M1, R2M2  the seed rescue inside RESCUE2's calculateCandRegions
    // A read that has no seed after the last seed round, or no alignment after RESCUE2, is unmapped.
    enum { S_A, S_M1, S_M2, S_R2, S_R2M1, S_R2M2, S_COUNT };
    std::vector<int> pending[S_COUNT];
    pending[S_A].resize((size_t)n);
    for (int i = 0; i < n; ++i) pending[S_A][(size_t)i] = i;
    for (int state = S_A; state < S_COUNT; ++state) {
        const std::vector<int>& who = pending[state];
        if (who.empty()) continue;
        alfa_map_options mo;
        map_options(mo);
        const bool r2 = state >= S_R2;
        const int memStage = (state == S_M1 || state == S_R2M1) ? 1 : (state == S_M2 || state == S_R2M2) ? 2 : 0;
        if (r2) mo.rescue_only = 1;
        if (memStage) mo.max_seed_candidates = 100000;
        const auto td0 = std::chrono::steady_clock::now();
        rd_.resize(who.size());
        for (size_t k = 0; k < who.size(); ++k) {
            const int i = who[k];
            describe(S.strand(i, false), S.strand(i, true), raw[i].len, rd_[k], r2 ? 2.0 * opt_.errorPercent : opt_.errorPercent, memStage);
        }
        g_map_us[0] += (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - td0).count();
        calls.emplace_back(new MapCallResult());
        MapCallResult& R = *calls.back();
        map_call(mo, (int)who.size(), R);
        for (size_t k = 0; k < who.size(); ++k) {
            const int i = who[k];
            res[(size_t)i] = RawResult{&R, (int)k, false};
            const int st = R.st[k];
            int next = -2;                                         // -2: final as it is
            if (st == ALFA_MAP_NO_SEED)
                next = state == S_A ? S_M1 : state == S_M1 ? S_M2 : state == S_R2 ? S_R2M1 : state == S_R2M1 ? S_R2M2 : -1;
            else if (st == ALFA_MAP_NO_ALIGNMENT) next = r2 ? -1 : S_R2;
            if (next == -1) res[(size_t)i].unmapped = true;
            else if (next >= 0) pending[next].push_back(i);
        }
    }
    long nd = 0;
    for (int i = 0; i < n; ++i) nd += device_done(res[(size_t)i]);
    g_core_done += nd;
    g_core_fallback += n - nd;
}

// a read the device finished, with no alignment or with exactly one that brought its CIGAR along, and one alignment to print
bool Mapper::direct_ok(const RawResult& rs) const {
    if (rs.unmapped) return true;
    const MapCallResult& R = *rs.R;
    const size_t i = (size_t)rs.j;
    if (R.st[i] != 0 || opt_.fullDP || opt_.alignmentCount < 1) return false;
    const int64_t na = R.aoff[i + 1] - R.aoff[i];
    return na == 0 || (na == 1 && R.al[R.aoff[i]].op_count >= 0);
}

namespace {
inline char* put_int(char* p, long v) {
    auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}
}  // namespace

void Mapper::render_direct(const RawStrands& S, const RawRead& rr, int i, const RawResult& rs, std::string& out, long* mappedPrinted) const {
    const int len = rr.len;
    const size_t at = out.size();
    const MapCallResult& R = *rs.R;
    const size_t j = (size_t)rs.j;
    if (rs.unmapped || R.aoff[j + 1] == R.aoff[j]) {          // read_t::emptyAlingment src/mapper.h:170-179, unpaired
        out.resize(at + (size_t)rr.nameLen + 2 * (size_t)len + 32);
        char* p = &out[at];
        memcpy(p, rr.name, (size_t)rr.nameLen); p += rr.nameLen;
        static const char mid[] = "\t4\t*\t0\t0\t*\t*\t0\t0\t";
        memcpy(p, mid, sizeof(mid) - 1); p += sizeof(mid) - 1;
        memcpy(p, S.strand(i, false), (size_t)len); p += len;
        *p++ = '\t';
        memcpy(p, rr.qual, (size_t)len); p += len;
        *p++ = '\n';
        out.resize((size_t)(p - out.data()));
        return;
    }
    const alfa_map_alignment& a = R.al[R.aoff[j]];
    const alfa_map_cigar_op* ops = R.op + R.ooff[j] + a.op_first;
    const int nops = a.op_count;
    const bool rc = !a.forward;
    // setLocalPos src/mapper.h:104-112
    int64_t seq, pos;
    idx_.from_set(a.ref_begin + 1, seq, pos);
    const char* rname = idx_.descr(seq);
    const size_t rl = strlen(rname);
    // MAPQ of a single alignment, read_t::postprocess src/mapper.cpp:343-357
    const int secBest = opt_.mismatch * len;
    int mapq = a.score == secBest ? 0 : MAX_MAPQ;
    out.resize(at + (size_t)rr.nameLen + rl + 2 * (size_t)len + 24 * (size_t)nops + 160);
    char* p = &out[at];
    memcpy(p, rr.name, (size_t)rr.nameLen); p += rr.nameLen;
    *p++ = '\t';
    if (rc) { *p++ = '1'; *p++ = '6'; } else *p++ = '0';
    *p++ = '\t';
    memcpy(p, rname, rl); p += rl;
    *p++ = '\t';
    p = put_int(p, (long)pos);
    *p++ = '\t';
    p = put_int(p, mapq);
    *p++ = '\t';
    // recoverCIGAR3from4 src/mapper.cpp:54-77: '=' and 'X' runs merge into M
    long run = 0;
    for (int k = 0; k < nops; ++k) {
        const char c = (char)ops[k].code;
        if (c == '=' || c == 'X') { run += ops[k].len; continue; }
        if (run > 0) { p = put_int(p, run); *p++ = 'M'; run = 0; }
        p = put_int(p, ops[k].len); *p++ = c;
    }
    if (run > 0) { p = put_int(p, run); *p++ = 'M'; }
    static const char mid[] = "\t*\t0\t0\t";
    memcpy(p, mid, sizeof(mid) - 1); p += sizeof(mid) - 1;
    memcpy(p, S.strand(i, rc), (size_t)len); p += len;
    *p++ = '\t';
    if (rc) { for (int k = 0; k < len; ++k) p[k] = rr.qual[len - 1 - k]; }
    else memcpy(p, rr.qual, (size_t)len);
    p += len;
    static const char t1[] = "\tAS:i:", t2[] = "\tNM:i:", t3[] = "\tX0:Z:";
    memcpy(p, t1, 6); p += 6; p = put_int(p, a.score);
    memcpy(p, t2, 6); p += 6; p = put_int(p, a.edit_dist);
    memcpy(p, t3, 6); p += 6;
    for (int k = 0; k < nops; ++k) { p = put_int(p, ops[k].len); *p++ = (char)ops[k].code; }
    *p++ = '\n';
    out.resize((size_t)(p - out.data()));
    mappedPrinted[0]++; mappedPrinted[1]++;
}

void Mapper::map_unpaired_rest(Read& read, bool done) {
    if (!done) unpaired_match(read);
    postprocess(read, false);
}

bool Mapper::finish_unpaired_direct(Read& read) { return postprocess(read, false, false); }

void Mapper::map_mate_rest(Read& read, bool done) {
    if (!done) unpaired_match(read);
}

// ------------------------------------------------------------------------------------------ paired end, mode 1
// isConcordant src/mapper.cpp:1311-1392
bool Mapper::is_concordant(const Alignment& m1, const Alignment& m2) const {
    bool firstLeft = true;
    if (m1.rname != m2.rname) return false;
    if (popt_.orientation == PAIR_FF) {
        if (m1.rc() != m2.rc()) return false;
        firstLeft = !m1.rc();
    } else if (popt_.orientation == PAIR_FR) {
        if (m1.rc() == m2.rc()) return false;
        firstLeft = !m1.rc();
    } else {
        if (m1.rc() == m2.rc()) return false;
        firstLeft = m1.rc();
    }
    if (firstLeft && m2.refE <= m1.refB) return false;
    else if (!firstLeft && m1.refE <= m2.refB) return false;
    int64_t begin = std::min(m1.refB, m2.refB);
    int64_t end = std::max(m1.refE, m2.refE);
    int64_t size = end - begin + 1;
    if (size < popt_.minInsert || size > popt_.maxInsert) return false;
    int64_t begin1 = m1.pos, end1 = begin1 + (int64_t)m1.refLength - 1;
    int64_t begin2 = m2.pos, end2 = begin2 + (int64_t)m2.refLength - 1;
    bool contain = (begin1 >= begin2 && end1 <= end2) || (begin2 >= begin1 && end2 <= end1);
    if (contain && !popt_.contain) return false;
    bool overlap = contain || (begin2 >= begin1 && begin2 <= end1) || (end2 >= begin1 && end2 <= end1);
    if (!popt_.overlap && overlap) return false;
    if ((firstLeft && (begin2 < begin1)) || (!firstLeft && (begin1 < begin2))) {
        if (!overlap) return false;
        else if (!popt_.dovetail) return false;
    }
    return true;
}

// alignment_t::setMate src/mapper.cpp:273-289
static void set_mate(Alignment& a, const Alignment& o, bool concordant, bool upstream) {
    a.mate.flag |= 1u | (1u << 1);
    if (o.rc()) a.mate.flag |= 1u << 5; else a.mate.flag &= ~(1u << 5);
    if (upstream) { a.mate.flag |= 1u << 6; a.mate.flag &= ~(1u << 7); }
    else { a.mate.flag &= ~(1u << 6); a.mate.flag |= 1u << 7; }
    a.mate.concordant = concordant;
    a.mate.pnext = o.pos;
    a.mate.pnextGlob = o.refB;
    a.mate.tLength = (int)(std::max(a.refE, o.refE) - std::min(a.refB, o.refB) + 1);
    int64_t firstPos = a.rc() ? a.refE : a.refB;
    int64_t secondPos = o.rc() ? o.refE : o.refB;
    if (firstPos > secondPos) a.mate.tLength *= -1;
    a.mate.rnext = a.rname == o.rname ? "=" : o.rname;
    if (concordant) a.mate.alignmentScore = o.alignmentScore;
}

// unpairedCopy :1394-1408
static Alignment* unpaired_copy(const Alignment* a) {
    Alignment* n = new Alignment();
    n->refB = a->refB; n->refE = a->refE; n->qB = a->qB; n->qE = a->qE; n->pos = a->pos; n->rname = a->rname;
    n->refLength = a->refLength; n->alignmentScore = a->alignmentScore;
    if (a->rc()) n->flag |= 1u << 4;
    n->chain = a->chain;
    return n;
}

// setPaired (both overloads) src/mapper.cpp:1410-1437
void Mapper::set_paired(Read& m1, Read& m2, int i, int j, bool concordant) {
    set_local_pos(*m1.alignments[i]);
    set_local_pos(*m2.alignments[j]);
    int i1 = i, i2 = j;
    if (m1.alignments[i1]->paired()) { m1.alignments.push_back(unpaired_copy(m1.alignments[i1])); i1 = (int)m1.alignments.size() - 1; }
    if (m2.alignments[i2]->paired()) { m2.alignments.push_back(unpaired_copy(m2.alignments[i2])); i2 = (int)m2.alignments.size() - 1; }
    m1.pairedAlignmentCount++;
    m2.pairedAlignmentCount++;
    Alignment& a = *m1.alignments[i1];
    Alignment& b = *m2.alignments[i2];
    a.flag |= 1u; b.flag |= 1u;
    set_mate(a, b, concordant, true);
    set_mate(b, a, concordant, false);
}

// dpWindow :1464-1510 -- where the other mate has to lie, given one mate's alignment
static void dp_window(int64_t mateRefB, int64_t mateRefE, bool mateIsFirst, bool mateIsFw, int Plength, const PairedOpt& o,
                      int64_t& refB, int64_t& refE, bool& otherFW) {
    const int64_t anchor = mateIsFw ? mateRefB : mateRefE;
    bool right;
    if (o.orientation == PAIR_FF) { right = mateIsFirst ? mateIsFw : !mateIsFw; otherFW = mateIsFw; }
    else if (o.orientation == PAIR_FR) { right = mateIsFw; otherFW = !mateIsFw; }
    else { right = !mateIsFw; otherFW = !mateIsFw; }
    if (right) { refB = anchor + o.minInsert; refE = anchor + o.maxInsert; }
    else { refB = anchor - o.maxInsert; refE = anchor - o.minInsert; }
    if (otherFW) refE += Plength; else refB -= Plength;
    if (!o.overlap) { if (right) refB = std::max(refB, mateRefE); else refE = std::min(refE, mateRefB); }
    if (!o.dovetail) { if (right) refB = std::max(refB, mateRefB); else refE = std::min(refE, mateRefE); }
}

// bestAlignmentInWindow :1526-1540
static int best_alignment_in_window(const Read& mate, const std::vector<Mapper::PairInfo>& mateAlns, bool firstMate, int64_t refB, int64_t refE,
                                    bool fw) {
    int best = -1;
    int bestScore = -1 * (int)mate.sequence.size();
    for (size_t k = 0; k < mateAlns.size(); k++)
        if (firstMate != mateAlns[k].first) {
            const Alignment* ma = mate.alignments[mateAlns[k].pos];
            if (ma->refB >= refB && ma->refE <= refE && ma->rc() != fw && ma->alignmentScore > bestScore) {
                best = mateAlns[k].pos;
                bestScore = ma->alignmentScore;
            }
        }
    return best;
}

// the ksw_align call of :1565-1590 and :2190-2210: local alignment of the mate inside the window; a new alignment (no chain:
// postprocess runs fullDP on it) when the score passes.  Returns its index or -1.
int Mapper::window_align(Read& windowMate, int64_t refB, int64_t refE, bool fw, int minScore) {
    // the reference reads sa.S[refB + j] without looking at the text's bounds (:1569); positions outside it are given the
    // code of a non-ACGT character here
    const char* S = idx_.S();
    const int64_t N = idx_.v->N;
    const int Plength = (int)windowMate.sequence.length();
    const int tLen = (int)std::max<int64_t>(refE - refB + 1, 0);
    std::vector<uint8_t> target((size_t)tLen);
    for (int j = 0; j < tLen; ++j) {
        const int64_t p = refB + j;
        target[j] = (p < 0 || p >= N) ? 4 : nt4((unsigned char)S[p]);
    }
    const int xtra = ALFA_KSW_XSTART | (Plength * opt_.match < 250 ? ALFA_KSW_XBYTE : 0);
    alfa_kswr r;
    const std::vector<uint8_t>& qv = fw ? windowMate.readP : windowMate.rcReadP;
    hot_.submit_align(AlignReq{alfa_align_task{qv.data(), target.data(), Plength, tLen, xtra}, &r});
    hot_.wait();
    if (r.score <= minScore) return -1;
    Alignment* n = new Alignment();
    n->alignmentScore = r.score;
    n->qB = r.qb; n->qE = r.qe;
    n->refB = refB + r.tb; n->refE = refB + r.te;
    n->refLength = r.te - r.tb + 1;
    if (!fw) n->flag |= 1u << 4;
    windowMate.alignments.push_back(n);
    return (int)windowMate.alignments.size() - 1;
}

// pairAlignmentToWindow :1542-1606: for every still unpaired alignment, in score order, look for (or make, with ksw_align) an
// alignment of the other mate inside the window and pair the two
void Mapper::pair_alignment_to_window(Read& m1, Read& m2, const std::vector<PairInfo>& order, std::vector<PairInfo>& windowAlns,
                                      int& alnCount, int maxAlnCount) {
    size_t i = 0;
    int trial = 0;
    while (i < order.size() && alnCount < maxAlnCount && trial < opt_.maxTrial) {
        const bool alnFirst = order[i].first;
        Alignment* aln = alnFirst ? m1.alignments[order[i].pos] : m2.alignments[order[i].pos];
        if (!aln->paired()) {
            trial++;
            int64_t refB, refE;
            bool fw;
            Read& windowMate = alnFirst ? m2 : m1;
            const int Plength = (int)windowMate.sequence.length();
            dp_window(aln->refB, aln->refE, alnFirst, !aln->rc(), Plength, popt_, refB, refE, fw);
            int windowMatePos = best_alignment_in_window(windowMate, windowAlns, alnFirst, refB, refE, fw);
            if (windowMatePos < 0) {
                const int editDist = (int)(opt_.errorPercent * Plength) + 1;
                const int minScore = (int)(Plength * opt_.match + (opt_.mismatch - opt_.match) * editDist);
                windowMatePos = window_align(windowMate, refB, refE, fw, minScore);
                if (windowMatePos >= 0)
                    windowAlns.push_back(PairInfo{!alnFirst, windowMatePos, windowMate.alignments[windowMatePos]->alignmentScore});
            }
            if (windowMatePos >= 0) {
                if (alnFirst) set_paired(m1, m2, order[i].pos, windowMatePos, true);
                else set_paired(m1, m2, windowMatePos, order[i].pos, true);
                trial = 0;
                alnCount++;
            }
        }
        i++;
    }
}

// pairAlignmentsOrdered :1608-1637
void Mapper::pair_alignments_ordered(Read& m1, Read& m2, const std::vector<PairInfo>& order, int& alnCount, int maxPaired) {
    size_t i = 0;
    while (i < order.size() && alnCount < maxPaired) {
        size_t j = i + 1;
        bool paired = order[i].first ? m1.alignments[order[i].pos]->paired() : m2.alignments[order[i].pos]->paired();
        if (!paired) {
            while (j < order.size() && alnCount < maxPaired) {
                if (order[i].first != order[j].first) {
                    int i1 = order[i].first ? order[i].pos : order[j].pos;
                    int i2 = order[i].first ? order[j].pos : order[i].pos;
                    Alignment* a1 = m1.alignments[i1];
                    set_local_pos(*a1);
                    Alignment* a2 = m2.alignments[i2];
                    set_local_pos(*a2);
                    if (is_concordant(*a1, *a2)) { set_paired(m1, m2, i1, i2, true); alnCount++; }
                }
                j++;
            }
        }
        i++;
    }
}

// calculateDiscordantAlignments :1439-1462
void Mapper::discordant_alignments(Read& m1, Read& m2) {
    if (m1.alignments.empty() || m2.alignments.empty()) return;
    std::sort(m1.alignments.begin(), m1.alignments.end(), cmp_aln);
    std::sort(m2.alignments.begin(), m2.alignments.end(), cmp_aln);
    int maxFirst = m1.alignments[0]->alignmentScore, maxSecond = m2.alignments[0]->alignmentScore;
    size_t n1 = m1.alignments.size(), n2 = m2.alignments.size();
    size_t i = 0;
    while (i < n1 && m1.alignments[i]->alignmentScore == maxFirst) {
        if (!m1.alignments[i]->concordant()) {
            size_t j = i;
            while (j < n2 && m2.alignments[j]->alignmentScore == maxSecond) {
                if (!m2.alignments[j]->concordant()) set_paired(m1, m2, (int)i, (int)j, false);
                j++;
            }
        }
        i++;
    }
}

// alignments of both mates in descending score order (compAlnScorePaired :1520)
std::vector<Mapper::PairInfo> Mapper::score_order(const Read& m1, const Read& m2) {
    std::vector<PairInfo> order;
    order.reserve(m1.alignments.size() + m2.alignments.size());
    for (size_t i = 0; i < m1.alignments.size(); i++) order.push_back(PairInfo{true, (int)i, m1.alignments[i]->alignmentScore});
    for (size_t i = 0; i < m2.alignments.size(); i++) order.push_back(PairInfo{false, (int)i, m2.alignments[i]->alignmentScore});
    std::sort(order.begin(), order.end(), [](const PairInfo& a, const PairInfo& b) { return a.score > b.score; });
    return order;
}

// pairedMatch1 src/mapper.cpp:1640-1682: map both mates, pair what is concordant, rescue through windows, then discordant pairs
void Mapper::paired_match1(Read& m1, Read& m2) {
    unpaired_match(m1);
    unpaired_match(m2);
    pair_mates1(m1, m2);
}

// the part of pairedMatch1 behind the two unpairedMatch calls (:1648-1681)
void Mapper::pair_mates1(Read& m1, Read& m2) {
    const int alnCount1 = (int)m1.alignments.size(), alnCount2 = (int)m2.alignments.size();
    int alnCount = 0;
    const int maxPaired = opt_.alignmentCount + 1;
    std::vector<PairInfo> order = score_order(m1, m2);
    if (alnCount1 > 0 && alnCount2 > 0) pair_alignments_ordered(m1, m2, order, alnCount, maxPaired);
    if (popt_.pairedRescue && alnCount < maxPaired && std::max(alnCount1, alnCount2) > 0) {      // :1668-1675
        std::vector<PairInfo> fromWindow;
        pair_alignment_to_window(m1, m2, order, fromWindow, alnCount, maxPaired);
    }
    if (popt_.discordant && alnCount < maxPaired) discordant_alignments(m1, m2);
}

// pairedMatch2 :2377-2426: map mate 1, look for mate 2 in the windows of its alignments, then the other way round
void Mapper::paired_match2(Read& m1, Read& m2) {
    unpaired_match(m1);
    std::sort(m1.alignments.begin(), m1.alignments.end(), cmp_aln);
    int alnCount = 0;
    const int maxPaired = opt_.alignmentCount + 1;
    auto by_score = [](const PairInfo& a, const PairInfo& b) { return a.score > b.score; };
    std::vector<PairInfo> mate1Alns, mate2Alns;
    for (int j = 0; j < (int)m1.alignments.size(); j++) mate1Alns.push_back(PairInfo{true, j, m1.alignments[j]->alignmentScore});
    std::sort(mate1Alns.begin(), mate1Alns.end(), by_score);
    pair_alignment_to_window(m1, m2, mate1Alns, mate2Alns, alnCount, maxPaired);
    if (alnCount < maxPaired) {
        unpaired_match(m2);
        mate2Alns.clear();
        for (int j = 0; j < (int)m2.alignments.size(); j++)
            if (!m2.alignments[j]->paired()) mate2Alns.push_back(PairInfo{false, j, m2.alignments[j]->alignmentScore});
        std::sort(mate2Alns.begin(), mate2Alns.end(), by_score);
        pair_alignment_to_window(m1, m2, mate2Alns, mate1Alns, alnCount, maxPaired);
    }
    if (popt_.pairedRescue && alnCount < maxPaired && std::max((int)m1.alignments.size(), (int)m2.alignments.size()) > 0) {
        std::vector<PairInfo> order = score_order(m1, m2);
        pair_alignments_ordered(m1, m2, order, alnCount, maxPaired);
    }
    if (popt_.discordant && alnCount < maxPaired) discordant_alignments(m1, m2);
}

// isConcordantLIS :1692-1758 (its `end2` is mate1's end, :1731 -- kept)
bool Mapper::is_concordant_lis(const Lis& a, const Lis& b, int editDist1, int editDist2) const {
    bool firstLeft;
    if ((popt_.orientation == PAIR_RF || popt_.orientation == PAIR_FR) && a.fw == b.fw) return false;
    else if (popt_.orientation == PAIR_FF && a.fw != b.fw) return false;
    if (popt_.orientation == PAIR_FF || popt_.orientation == PAIR_FR) firstLeft = a.fw;
    else firstLeft = !a.fw;
    if (firstLeft && b.refEnd <= a.refBegin) return false;
    else if (!firstLeft && a.refEnd <= b.refBegin) return false;
    const int64_t begin = std::min(a.refBegin, b.refBegin), end = std::max(a.refEnd, b.refEnd), size = end - begin + 1;
    if (size + (editDist1 + editDist2) < popt_.minInsert || size - (editDist1 + editDist2) > popt_.maxInsert) return false;
    const int64_t begin1 = a.refBegin, end1 = a.refEnd, begin2 = b.refBegin, end2 = a.refEnd;
    const bool contain = (begin1 - editDist1 >= begin2 + editDist2 && end1 + editDist1 <= end2 - editDist2) ||
                         (begin2 - editDist2 >= begin1 + editDist1 && end2 + editDist2 <= end1 - editDist1);
    if (contain && !popt_.contain) return false;
    const bool overlap = contain || !(end2 - editDist2 < begin1 + editDist1 || begin2 + editDist2 > end1 - editDist1);
    if (!popt_.overlap && overlap) return false;
    if ((firstLeft && (begin2 + editDist2 < begin1 - editDist1)) || (!firstLeft && (begin1 + editDist1 < begin2 - editDist2))) {
        if (!overlap) return false;
        else if (!popt_.dovetail) return false;
    }
    return true;
}

// the tail shared by pairedMatch3 and pairedMatch4 (:1815-1841, :2102-2134): pair by score order, optionally rescue, discordant
void Mapper::pair_and_finish(Read& m1, Read& m2, bool withWindowRescue) {
    int alnCount = 0;
    const int maxPaired = opt_.alignmentCount + 1;
    const int alnCount1 = (int)m1.alignments.size(), alnCount2 = (int)m2.alignments.size();
    std::vector<PairInfo> order = score_order(m1, m2);
    if (alnCount1 > 0 && alnCount2 > 0) pair_alignments_ordered(m1, m2, order, alnCount, maxPaired);
    if (withWindowRescue && popt_.pairedRescue && alnCount < maxPaired && std::max(alnCount1, alnCount2) > 0) {
        std::vector<PairInfo> fromWindow;
        pair_alignment_to_window(m1, m2, order, fromWindow, alnCount, maxPaired);
    }
    if (popt_.discordant && alnCount < maxPaired) discordant_alignments(m1, m2);
}

// pairedMatch3 :1761-1843: candidate regions of both mates, keep the regions that pair, align those
void Mapper::paired_match3(Read& m1, Read& m2) {
    const int len1 = (int)m1.sequence.length(), len2 = (int)m2.sequence.length();
    const int editDist1 = (int)(opt_.errorPercent * len1) + 1, editDist2 = (int)(opt_.errorPercent * len2) + 1;
    std::vector<Lis> lis1, lis2, f1, f2;
    cand_regions(m1, lis1, opt_.errorPercent);
    cand_regions(m2, lis2, opt_.errorPercent);
    for (size_t i = 0; i < lis1.size(); i++)
        for (size_t j = 0; j < lis2.size(); j++)
            if (!(lis1[i].extended && lis2[j].extended) && is_concordant_lis(lis1[i], lis2[j], editDist1, editDist2)) {
                if (!lis1[i].extended) { f1.push_back(lis1[i]); lis1[i].extended = true; }
                if (!lis2[j].extended) { f2.push_back(lis2[j]); lis2[j].extended = true; }
            }
    if (f1.empty() && f2.empty()) { paired_match1(m1, m2); return; }
    int minScore1 = (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * (editDist1 * 1.5));
    int minScore2 = (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * (editDist2 * 1.5));
    process_lis(m1, f1, editDist1, opt_.maxAlnCount, 0, minScore1);
    process_lis(m2, f2, editDist2, opt_.maxAlnCount, 0, minScore2);
    if (m1.alignments.empty() && opt_.rescue) {
        minScore1 = std::max(0, (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len1));
        process_lis(m1, f1, editDist1, opt_.maxAlnCount, 0, minScore1);
    }
    if (m2.alignments.empty() && opt_.rescue) {
        minScore2 = std::max(0, (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len2));
        process_lis(m2, f2, editDist2, opt_.maxAlnCount, 0, minScore2);
    }
    pair_and_finish(m1, m2, false);
}

// processLisIntervalsSimultaneous :1857-1939: extend the two regions of a couple together and pair what they give
void Mapper::process_lis_simultaneous(Read& m1, Read& m2, const std::vector<Lis>& lis1, const std::vector<Lis>& lis2,
                                      const std::vector<LisCouple>& order, int maxAlnCount, double minCoveragePercent, int minScore1, int minScore2) {
    const int len1 = (int)m1.sequence.length(), len2 = (int)m2.sequence.length();
    const int minCoverage1 = (int)(len1 * minCoveragePercent), minCoverage2 = (int)(len2 * minCoveragePercent);
    const int editDist1 = (int)(opt_.errorPercent * len1) + 1, editDist2 = (int)(opt_.errorPercent * len2) + 1;
    size_t lisIndex = 0;
    int trial = 0, maxAlnScore = 0, alnCount = 0;
    while (alnCount < maxAlnCount && lisIndex < order.size()) {
        const Lis& c1 = lis1[order[lisIndex].first];
        const Lis& c2 = lis2[order[lisIndex].second];
        if (trial < opt_.maxTrial) {
            int64_t cs1, ce1, cs2, ce2;
            idx_.chrom_bounds(c1.matches[0].ref, cs1, ce1);
            idx_.chrom_bounds(c2.matches[0].ref, cs2, ce2);
            const size_t base1 = m1.alignments.size(), base2 = m2.alignments.size();
            int r1 = extend_chain(c1.fw ? m1.sequence : m1.rcSequence, c1.matches, editDist1, minScore1, minCoverage1, cs1, ce1, m1, c1.fw);
            int r2 = extend_chain(c2.fw ? m2.sequence : m2.rcSequence, c2.matches, editDist2, minScore2, minCoverage2, cs2, ce2, m2, c2.fw);
            if (base1 == m1.alignments.size() && base2 < m2.alignments.size() && opt_.rescue) {          // the lowered bound stays (:1904)
                minScore1 = std::max(0, (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len1));
                r1 = extend_chain(c1.fw ? m1.sequence : m1.rcSequence, c1.matches, editDist1, minScore1, minCoverage1, cs1, ce1, m1, c1.fw);
            } else if (base1 < m1.alignments.size() && base2 == m2.alignments.size() && opt_.rescue) {
                minScore2 = std::max(0, (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len2));
                r2 = extend_chain(c2.fw ? m2.sequence : m2.rcSequence, c2.matches, editDist2, minScore2, minCoverage2, cs2, ce2, m2, c2.fw);
            }
            if (base1 < m1.alignments.size() && base2 < m2.alignments.size()) {
                const size_t max1 = m1.alignments.size(), max2 = m2.alignments.size();
                for (size_t i = base1; i < max1; i++)
                    for (size_t j = base2; j < max2; j++) {
                        Alignment* a1 = m1.alignments[i];
                        set_local_pos(*a1);
                        Alignment* a2 = m2.alignments[j];
                        set_local_pos(*a2);
                        if (is_concordant(*a1, *a2)) { set_paired(m1, m2, (int)i, (int)j, true); alnCount++; }
                    }
            }
            trial++;
            const int ret = r1 + r2;
            if (ret >= 0 && (!opt_.maxTrialBest || maxAlnScore < ret)) trial = 0;
            maxAlnScore = std::max(maxAlnScore, ret);
        }
        lisIndex++;
    }
}

// pairedMatch6 :1941-1985: couples of regions that pair, best summed score first
void Mapper::paired_match6(Read& m1, Read& m2) {
    const int len1 = (int)m1.sequence.length(), len2 = (int)m2.sequence.length();
    const int editDist1 = (int)(opt_.errorPercent * len1) + 1, editDist2 = (int)(opt_.errorPercent * len2) + 1;
    std::vector<Lis> lis1, lis2;
    cand_regions(m1, lis1, opt_.errorPercent);
    cand_regions(m2, lis2, opt_.errorPercent);
    std::vector<LisCouple> couples;
    for (size_t i = 0; i < lis1.size(); i++)
        for (size_t j = 0; j < lis2.size(); j++)
            if (!(lis1[i].extended && lis2[j].extended) && is_concordant_lis(lis1[i], lis2[j], editDist1, editDist2))
                couples.push_back(LisCouple{(int)i, (int)j, lis1[i].score + lis2[j].score});
    if (couples.empty()) { paired_match1(m1, m2); return; }
    std::sort(couples.begin(), couples.end(), [](const LisCouple& a, const LisCouple& b) { return a.score > b.score; });      // compLIScouple :1853
    const int minScore1 = (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * (editDist1 * 1.5));
    const int minScore2 = (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * (editDist2 * 1.5));
    process_lis_simultaneous(m1, m2, lis1, lis2, couples, opt_.maxAlnCount, 0, minScore1, minScore2);
    if (popt_.discordant && m1.pairedAlignmentCount < opt_.alignmentCount + 1) discordant_alignments(m1, m2);
}

// processLisIntervalsPaired :1987-2050 (andPair = false) and processLisIntervalsAndPair :2225-2312 (andPair = true): the regions
// of both mates in one score order; the second form looks for the other mate right after every extension
void Mapper::process_lis_both(Read& m1, Read& m2, std::vector<Lis>& lis1, std::vector<Lis>& lis2, const std::vector<PairInfo>& order,
                              int maxAlnCount, double minCoveragePercent, int minScore1, int minScore2, bool andPair) {
    const int len1 = (int)m1.sequence.length(), len2 = (int)m2.sequence.length();
    const int minCoverage1 = (int)(len1 * minCoveragePercent), minCoverage2 = (int)(len2 * minCoveragePercent);
    const int editDist1 = (int)(opt_.errorPercent * len1) + 1, editDist2 = (int)(opt_.errorPercent * len2) + 1;
    size_t lisIndex = 0;
    int trialF = 0, trialR = 0, maxAlnScore = 0, alnCount = 0;
    QCovTree treeF, treeR;
    std::vector<PairInfo> mate1Alns, mate2Alns;
    while (alnCount < maxAlnCount && lisIndex < order.size()) {
        const bool first = order[lisIndex].first;
        Lis& cluster = first ? lis1[order[lisIndex].pos] : lis2[order[lisIndex].pos];
        const bool unique = has_unique(cluster, cluster.fw ? treeF : treeR);
        const int trial = cluster.fw ? trialF : trialR;
        const bool highCoverage = first ? cluster.score >= minCoverage1 : cluster.score >= minCoverage2;
        const bool criterium = (trial < opt_.maxTrial || unique) && (cluster.matches.size() > 1 || highCoverage);
        if (criterium && !(andPair && cluster.extended)) {
            int64_t cs, ce;
            idx_.chrom_bounds(cluster.matches[0].ref, cs, ce);
            int ret;
            if (first) {
                const size_t extendCount = m1.alignments.size();
                ret = extend_chain(cluster.fw ? m1.sequence : m1.rcSequence, cluster.matches, editDist1, minScore1, minCoverage1, cs, ce, m1, cluster.fw);
                if (andPair) {
                    const size_t lastAln = m1.alignments.size();
                    for (size_t i = extendCount; i < lastAln; i++) {          // always the first new alignment (:2280-2282), kept
                        mate1Alns.push_back(PairInfo{true, (int)extendCount, m1.alignments[extendCount]->alignmentScore});
                        find_or_align_mate(m1, m2, true, lis2, minScore2, mate2Alns, (int)extendCount);
                    }
                }
            } else {
                const size_t extendCount = m2.alignments.size();
                ret = extend_chain(cluster.fw ? m2.sequence : m2.rcSequence, cluster.matches, editDist2, minScore2, minCoverage2, cs, ce, m2, cluster.fw);
                if (andPair) {
                    const size_t lastAln = m2.alignments.size();
                    for (size_t i = extendCount; i < lastAln; i++) {          // appended to mate 1's list (:2292-2294), kept
                        mate1Alns.push_back(PairInfo{false, (int)extendCount, m2.alignments[extendCount]->alignmentScore});
                        find_or_align_mate(m2, m1, false, lis1, minScore1, mate1Alns, (int)extendCount);
                    }
                }
            }
            if (andPair) cluster.extended = true;
            if (cluster.fw) trialF++; else trialR++;
            if (ret >= 0 && (!opt_.maxTrialBest || maxAlnScore < ret)) {
                if (cluster.fw) trialF = 0; else trialR = 0;
            }
            alnCount = (int)(m1.alignments.size() + m2.alignments.size());
            maxAlnScore = std::max(maxAlnScore, ret);
        }
        lisIndex++;
    }
}

// findOrAlignMateToAlignment :2151-2223 (bestLISInWindow :2137-2149)
void Mapper::find_or_align_mate(Read& alignmentMate, Read& windowMate, bool alnMateFirst, std::vector<Lis>& lisWindow, int minScore,
                                std::vector<PairInfo>& windowAlns, int alignmentIndex) {
    Alignment* aln = alignmentMate.alignments[alignmentIndex];
    int64_t refB, refE;
    bool fw;
    const int Plength = (int)windowMate.sequence.length();
    dp_window(aln->refB, aln->refE, alnMateFirst, !aln->rc(), Plength, popt_, refB, refE, fw);
    int windowMatePos = best_alignment_in_window(windowMate, windowAlns, alnMateFirst, refB, refE, fw);
    int windowCount = 1;
    if (windowMatePos < 0) {
        const int editDist = (int)(opt_.errorPercent * Plength) + 1;
        int lisPos = -1, bestScore = -10000;
        for (size_t i = 0; i < lisWindow.size(); i++)
            if (lisWindow[i].refBegin + editDist >= refB && lisWindow[i].refEnd - editDist <= refE && lisWindow[i].fw == fw &&
                lisWindow[i].score > bestScore) {
                lisPos = (int)i;
                bestScore = lisWindow[i].score;
            }
        if (lisPos >= 0) {
            int64_t cs, ce;
            idx_.chrom_bounds(lisWindow[lisPos].matches[0].ref, cs, ce);
            const size_t extendCount = windowMate.alignments.size();
            extend_chain(lisWindow[lisPos].fw ? windowMate.sequence : windowMate.rcSequence, lisWindow[lisPos].matches, editDist, minScore,
                         (int)(Plength * 0.01), cs, ce, windowMate, lisWindow[lisPos].fw);
            if (windowMate.alignments.size() > extendCount) {
                lisWindow[lisPos].extended = true;
                windowMatePos = (int)extendCount;
                windowCount = (int)(windowMate.alignments.size() - extendCount);
            }
        }
    }
    if (windowMatePos < 0 && popt_.pairedRescue) windowMatePos = window_align(windowMate, refB, refE, fw, minScore);
    if (windowMatePos >= 0)
        for (int i = 0; i < windowCount; i++) {
            windowAlns.push_back(PairInfo{!alnMateFirst, windowMatePos + i, windowMate.alignments[windowMatePos + i]->alignmentScore});
            if (alnMateFirst) set_paired(alignmentMate, windowMate, alignmentIndex, windowMatePos + i, true);
            else set_paired(windowMate, alignmentMate, windowMatePos + i, alignmentIndex, true);
        }
}

// pairedMatch4 :2053-2135 (andPair = false) and pairedMatch5 :2315-2373 (andPair = true)
void Mapper::paired_match45(Read& m1, Read& m2, bool andPair) {
    std::vector<Lis> lis1, lis2;
    cand_regions(m1, lis1, opt_.errorPercent);
    cand_regions(m2, lis2, opt_.errorPercent);
    std::vector<PairInfo> order;
    order.reserve(lis1.size() + lis2.size());
    for (size_t i = 0; i < lis1.size(); i++) order.push_back(PairInfo{true, (int)i, lis1[i].score});
    for (size_t i = 0; i < lis2.size(); i++) order.push_back(PairInfo{false, (int)i, lis2[i].score});
    std::sort(order.begin(), order.end(), [](const PairInfo& a, const PairInfo& b) { return a.score > b.score; });
    const int len1 = (int)m1.sequence.length(), len2 = (int)m2.sequence.length();
    const int editDist1 = (int)(opt_.errorPercent * len1) + 1, editDist2 = (int)(opt_.errorPercent * len2) + 1;
    int minScore1 = (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * editDist1);
    int minScore2 = (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * editDist2);
    process_lis_both(m1, m2, lis1, lis2, order, opt_.maxAlnCount, opt_.minCoverage, minScore1, minScore2, andPair);
    if (m1.alignments.empty() && !m2.alignments.empty() && opt_.rescue) {
        minScore1 = std::max(0, (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len1));
        process_lis(m1, lis1, editDist1, 20, (int)(len1 * 0.01), minScore1);
    } else if (m2.alignments.empty() && !m1.alignments.empty() && opt_.rescue) {
        minScore2 = std::max(0, (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len2));
        process_lis(m2, lis2, editDist2, 20, (int)(len1 * 0.01), minScore2);          // mate 1's length (:2089, :2352), kept
    }
    if (m1.alignments.empty() && m2.alignments.empty() && popt_.pairedRescue) {
        minScore1 = std::max(0, (int)(len1 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len1));
        minScore2 = std::max(0, (int)(len2 * opt_.match + (opt_.mismatch - opt_.match) * 0.2 * len2));
        process_lis_both(m1, m2, lis1, lis2, order, 20, 0.01, minScore1, minScore2, andPair);
    }
    if (!andPair) { pair_and_finish(m1, m2, true); return; }
    const int alnCount = std::max(m1.pairedAlignmentCount, m2.pairedAlignmentCount);
    if (popt_.discordant && alnCount < opt_.alignmentCount + 1) discordant_alignments(m1, m2);
}

// Mode 1 in three steps, for a driver that runs the independent parts of a pair in fibers of their own (the two unpairedMatch
// calls touch one mate each, and so do the two postprocess calls): same calls, same order per mate, same results.
void Mapper::map_mate(Read& r) { r.init(); unpaired_match(r); }
void Mapper::pair_mates(Read& m1, Read& m2) { pair_mates1(m1, m2); }
void Mapper::finish_mate(Read& r) { postprocess(r, true); }

void Mapper::map_paired(Read& m1, Read& m2) {
    m1.init();
    m2.init();
    switch (popt_.mode) {                 // pairedMatch :2429-2451
        case 2: paired_match2(m1, m2); break;
        case 3: paired_match3(m1, m2); break;
        case 4: paired_match45(m1, m2, false); break;
        case 5: paired_match45(m1, m2, true); break;
        case 6: paired_match6(m1, m2); break;
        default: paired_match1(m1, m2);
    }
    postprocess(m1, true);
    postprocess(m2, true);
}

}  // namespace alfa
