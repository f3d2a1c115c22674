// host_index.cpp -- construction / persistence of the enhanced sparse suffix array on the host.
//
// Produces bit-identical arrays to the reference's sparseSA object (S, SA, LCP(vec+M), CHILD,
// KMR, ISA) so that the HBM-resident index searched by the HIP kernels is the same index the
// reference searches.  The construction *method* is our own (rank K-blocks, prefix-doubling
// suffix sort, linear 10-mer table scan); only the *result* is dictated by the reference:
//   suffix order         = Larsson-Sadakane suffixsort with a unique smallest terminator
//                          (src/sparseSA.cpp:478-532, src/qsufsort.c)
//   LCP                  = sparse Kasai (src/sparseSA.cpp:161-173) with the 255-overflow table
//   CHILD                = up/down/nextlIndex merged in one array (src/sparseSA.cpp:176-219)
//   KMR                  = SA interval of every ACGT 10-mer (src/sparseSA.cpp:222-313)
// File formats: src/sparseSA.cpp:316-476.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/alfa_index.h"
#include "errors.h"
#include "host_index_hook.h"
#include "seqio.h"

namespace alfa {
int (*g_device_construct)(int device, DeviceBuildIO* io) = nullptr;
int g_build_device = -1;
}  // namespace alfa

struct alfa_host_index {
    std::vector<std::string> descr;
    std::vector<int64_t> startpos;
    std::string S;  // kept with 16 bytes of '\0' slack after N
    int64_t N = 0, K = 1, logN = 0, NKm1 = 0;
    bool hasChild = true, hasSufLink = false, hasKmer = true;
    std::vector<uint32_t> SA;
    std::vector<int32_t> ISA;
    std::vector<uint8_t> LCP;
    std::vector<int64_t> Midx;
    std::vector<int32_t> Mval;
    std::vector<int32_t> CHILD;
    std::vector<uint32_t> KMR;  // (left,right) pairs
    alfa_index_view view;
    void refresh_view() {
        memset(&view, 0, sizeof(view));
        view.S = (const uint8_t*)S.data();
        view.N = N; view.K = K; view.logN = logN; view.NKm1 = NKm1;
        view.SA = SA.data(); view.n = (int64_t)SA.size();
        view.LCP = LCP.data();
        view.LCP_M_idx = Midx.data(); view.LCP_M_val = Mval.data(); view.nM = (int64_t)Midx.size();
        view.CHILD = hasChild ? CHILD.data() : nullptr;
        view.ISA = hasSufLink ? ISA.data() : nullptr;
        view.KMR = hasKmer ? KMR.data() : nullptr;
        view.has_child = hasChild; view.has_suflink = hasSufLink; view.has_kmer = hasKmer;
        view.n_seq = (int32_t)startpos.size();
        view.startpos = startpos.data();
    }
};

namespace {

const int KMER = 10;
const uint32_t TABLE = 1u << (2 * KMER);

// Sort the suffixes of the integer string x[0..n) (all symbols >= 1) where the end of the string
// compares smaller than every symbol.  Prefix doubling; groups carry their start index as rank.
void suffix_sort(const std::vector<int32_t>& x, std::vector<int32_t>& sa) {
    const int64_t n = (int64_t)x.size();
    sa.resize(n);
    if (n == 0) return;
    std::iota(sa.begin(), sa.end(), 0);
    std::sort(sa.begin(), sa.end(), [&](int32_t a, int32_t b) { return x[a] < x[b]; });
    std::vector<int32_t> rank(n), nrank(n);
    // initial ranks = start index of the group of equal first symbols
    {
        int64_t gs = 0;
        for (int64_t i = 0; i < n; i++) {
            if (i > 0 && x[sa[i]] != x[sa[i - 1]]) gs = i;
            rank[sa[i]] = (int32_t)gs;
        }
    }
    // list of unsorted groups [b,e)
    std::vector<std::pair<int64_t, int64_t>> groups, next;
    for (int64_t i = 0; i < n;) {
        int64_t j = i + 1;
        while (j < n && rank[sa[j]] == rank[sa[i]]) j++;
        if (j - i > 1) groups.push_back({i, j});
        i = j;
    }
    for (int64_t h = 1; !groups.empty(); h <<= 1) {
        auto key = [&](int32_t s) -> int64_t { return (int64_t)s + h < n ? (int64_t)rank[s + h] : -1; };
        next.clear();
        // sort each group by the rank of the suffix h further on, collect new ranks without
        // disturbing `rank` until every group has been processed
        for (auto& g : groups) {
            std::sort(sa.begin() + g.first, sa.begin() + g.second, [&](int32_t a, int32_t b) { return key(a) < key(b); });
        }
        for (auto& g : groups) {
            int64_t gs = g.first;
            for (int64_t i = g.first; i < g.second; i++) {
                if (i > g.first && key(sa[i]) != key(sa[i - 1])) {
                    if (i - gs > 1) next.push_back({gs, i});
                    gs = i;
                }
                nrank[sa[i]] = (int32_t)gs;
            }
            if (g.second - gs > 1) next.push_back({gs, g.second});
        }
        for (auto& g : groups)
            for (int64_t i = g.first; i < g.second; i++) rank[sa[i]] = nrank[sa[i]];
        groups.swap(next);
    }
}

struct LcpAcc {
    alfa_host_index& ix;
    int get(int64_t i) const {
        uint8_t b = ix.LCP[i];
        if (b != 255) return b;
        auto lo = std::lower_bound(ix.Midx.begin(), ix.Midx.end(), i);
        return ix.Mval[lo - ix.Midx.begin()];
    }
};

uint32_t bitadd(unsigned char c) {
    switch (c) {
        case 'a': case 'A': return 0u;
        case 'c': case 'C': return 1u;
        case 'g': case 'G': return 2u;
        case 't': case 'T': return 3u;
        default: return 0xFFFFFFFFu;
    }
}

int construct(alfa_host_index& ix) {
    const int64_t K = ix.K;
    // ---- init(): pad with '$' to a multiple of K, then K more (src/sparseSA.cpp:108-122)
    std::string& S = ix.S;
    if (S.length() % K != 0) S.append((size_t)(K - (int64_t)(S.length() % K)), '$');
    S.append((size_t)K, '$');
    ix.N = (int64_t)S.length();
    const int64_t N = ix.N, n = N / K;
    ix.logN = (int64_t)std::ceil(std::log((double)(N / K)) / std::log(2.0));
    ix.NKm1 = n - 1;

    // ---- the same arrays from the GPU builder (index_build.hip) when alfa_index_build_on_device() has switched it on.
    // (K = 1 relies on the final '$' being the smallest suffix, which a text with bytes below '$' would break: host then.)
    if (alfa::g_device_construct && (K > 1 || std::none_of(S.begin(), S.end(), [](char c) { return (unsigned char)c < (unsigned char)'$'; }))) {
        ix.SA.assign((size_t)n, 0); ix.LCP.assign((size_t)n, 0);
        if (ix.hasChild) ix.CHILD.assign((size_t)n, -1);
        if (ix.hasKmer) ix.KMR.assign(2 * (size_t)TABLE, 0);
        if (ix.hasSufLink) ix.ISA.assign((size_t)n, 0);
        S.append(16, '\0');
        alfa::DeviceBuildIO io;
        io.S = (const uint8_t*)S.data(); io.N = N; io.K = K;
        io.SA = ix.SA.data(); io.LCP = ix.LCP.data();
        io.CHILD = ix.hasChild ? ix.CHILD.data() : nullptr;
        io.KMR = ix.hasKmer ? ix.KMR.data() : nullptr;
        io.ISA = ix.hasSufLink ? ix.ISA.data() : nullptr;
        io.Midx = &ix.Midx; io.Mval = &ix.Mval;
        const int rc = alfa::g_device_construct(alfa::g_build_device, &io);
        if (rc) return rc;
        ix.refresh_view();
        return ALFA_OK;
    }

    // ---- suffix array
    std::vector<int32_t> sa32;
    if (K > 1) {
        // rank the K-character blocks (unsigned byte order), ranks start at 1
        std::vector<int32_t> order(n);
        std::iota(order.begin(), order.end(), 0);
        const unsigned char* s = (const unsigned char*)S.data();
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            int c = memcmp(s + (int64_t)a * K, s + (int64_t)b * K, (size_t)K);
            return c < 0;
        });
        std::vector<int32_t> x(n);
        int32_t r = 0;
        for (int64_t i = 0; i < n; i++) {
            if (i == 0 || memcmp(s + (int64_t)order[i] * K, s + (int64_t)order[i - 1] * K, (size_t)K) != 0) r++;
            x[order[i]] = r;
        }
        std::vector<int32_t>().swap(order);
        suffix_sort(x, sa32);
        ix.SA.resize(n);
        for (int64_t i = 0; i < n; i++) ix.SA[i] = (uint32_t)((int64_t)sa32[i] * K);
    } else {
        // the last '$' acts as the unique smallest terminator: it is SA[0]
        std::vector<int32_t> x(N - 1);
        for (int64_t i = 0; i < N - 1; i++) x[i] = (int32_t)(unsigned char)S[i] + 1;
        suffix_sort(x, sa32);
        ix.SA.resize(N);
        ix.SA[0] = (uint32_t)(N - 1);
        for (int64_t i = 0; i < N - 1; i++) ix.SA[i + 1] = (uint32_t)sa32[i];
    }
    std::vector<int32_t>().swap(sa32);
    ix.ISA.resize(n);
    for (int64_t i = 0; i < n; i++) ix.ISA[ix.SA[i] / K] = (int32_t)i;

    // ---- LCP: sparse Kasai (src/sparseSA.cpp:161-173)
    ix.LCP.assign(n, 0);
    ix.Midx.clear(); ix.Mval.clear();
    {
        std::vector<std::pair<int64_t, int32_t>> M;
        int64_t h = 0;
        for (int64_t i = 0; i < N; i += K) {
            int64_t m = ix.ISA[i / K];
            int64_t v;
            if (m == 0) v = 0;
            else {
                int64_t j = ix.SA[m - 1];
                while (i + h < N && j + h < N && S[i + h] == S[j + h]) h++;
                v = h;
            }
            if (v >= 255) { ix.LCP[m] = 255; M.push_back({m, (int32_t)v}); }
            else ix.LCP[m] = (uint8_t)v;
            h = std::max<int64_t>(0, h - K);
        }
        std::sort(M.begin(), M.end());
        for (auto& p : M) { ix.Midx.push_back(p.first); ix.Mval.push_back(p.second); }
    }
    if (!ix.hasSufLink) std::vector<int32_t>().swap(ix.ISA);

    LcpAcc L{ix};
    // ---- CHILD (src/sparseSA.cpp:176-219)
    if (ix.hasChild) {
        ix.CHILD.assign(n, -1);
        std::vector<int64_t> st;
        int64_t last = -1;
        st.push_back(0);
        for (int64_t i = 1; i < n; i++) {
            int li = L.get(i);
            while (li < L.get(st.back())) {
                last = st.back();
                st.pop_back();
                int lt = L.get(st.back());
                if (li <= lt && lt != L.get(last)) ix.CHILD[st.back()] = (int32_t)last;
            }
            if (last != -1) { ix.CHILD[i - 1] = (int32_t)last; last = -1; }
            st.push_back(i);
        }
        while (0 < L.get(st.back())) {
            last = st.back();
            st.pop_back();
            int lt = L.get(st.back());
            if (0 <= lt && lt != L.get(last)) ix.CHILD[st.back()] = (int32_t)last;
        }
        // next-l-index values
        st.clear();
        st.push_back(0);
        for (int64_t i = 1; i < n; i++) {
            int li = L.get(i);
            while (li < L.get(st.back())) st.pop_back();
            last = st.back();
            if (li == L.get(last)) { st.pop_back(); ix.CHILD[last] = (int32_t)i; }
            st.push_back(i);
        }
    }
    // ---- KMR: SA interval of every ACGT 10-mer.  The reference walks the lcp-interval tree to depth
    // 10 (src/sparseSA.cpp:222-313); the set of suffixes sharing a 10-char prefix is contiguous in SA,
    // so a linear scan yields the same (left,right) pairs.  Untouched entries stay (0,0).
    if (ix.hasKmer) {
        ix.KMR.assign(2 * (size_t)TABLE, 0);
        for (int64_t i = 0; i < n; i++) {
            int64_t p = ix.SA[i];
            if (p + KMER > N) continue;
            uint32_t code = 0;
            for (int j = 0; j < KMER; j++) code = (code << 2) | bitadd((unsigned char)S[p + j]);
            if (code >= TABLE) continue;
            uint32_t& left = ix.KMR[2 * (size_t)code];
            uint32_t& right = ix.KMR[2 * (size_t)code + 1];
            bool first = (i == 0) || [&] {
                int64_t q = ix.SA[i - 1];
                if (q + KMER > N) return true;
                return memcmp(S.data() + q, S.data() + p, KMER) != 0;
            }();
            if (first) left = (uint32_t)i;
            right = (uint32_t)i;
        }
    }
    S.append(16, '\0');  // read slack for the un-guarded S[SA+depth] probes
    ix.refresh_view();
    return ALFA_OK;
}

int finish_build(alfa_host_index* ix, int64_t K, int suflink, int child, int kmer, alfa_host_index** out) {
    if (K < 1) { delete ix; return alfa::fail(ALFA_EINVAL, "index sparseness K must be >= 1"); }
    ix->K = K; ix->hasSufLink = suflink != 0; ix->hasChild = child != 0; ix->hasKmer = kmer != 0;
    if ((int64_t)ix->S.size() / K > 0x7fffff00LL) { delete ix; return alfa::fail(ALFA_EINVAL, "N/K exceeds the int32 range of the index arrays"); }
    const int rc = construct(*ix);
    if (rc) { delete ix; return rc; }
    *out = ix;
    return ALFA_OK;
}

template <class T>
bool write_vec(const std::string& path, const std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    uint32_t n = (uint32_t)v.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(v.data(), sizeof(T), n, f);
    fclose(f);
    return true;
}
template <class T>
bool read_vec(const std::string& path, std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) { fclose(f); return false; }
    v.resize(n);
    bool ok = n == 0 || fread(v.data(), sizeof(T), n, f) == n;
    fclose(f);
    return ok;
}

struct ItemOnDisk { uint64_t idx; int32_t val; int32_t pad; };  // vec_uchar::item_t, 16 B with padding

}  // namespace

extern "C" {

int alfa_host_index_build_fasta(const char* fasta_path, int64_t K, int suflink, int child, int kmer, alfa_host_index** out) {
    if (!fasta_path || !out) return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::SeqReader rd;
    if (!rd.open(fasta_path)) return alfa::fail(ALFA_EINVAL, std::string("cannot open ") + fasta_path);
    alfa_host_index* ix = new alfa_host_index();
    alfa::SeqRecord r;
    while (rd.next(r) >= 0) {
        for (auto& c : r.seq) c = (char)std::tolower((unsigned char)c);
        if (ix->S.length() > 0) ix->S += '`';
        ix->startpos.push_back((int64_t)ix->S.length());
        ix->S += r.seq;
        ix->descr.push_back(r.name);
    }
    return finish_build(ix, K, suflink, child, kmer, out);
}

int alfa_host_index_build_mem(int n_seq, const char* const* names, const char* const* seqs, const int64_t* seq_lens, int64_t K,
                              int suflink, int child, int kmer, alfa_host_index** out) {
    if (!out || n_seq < 0) return alfa::fail(ALFA_EINVAL, "bad argument");
    alfa_host_index* ix = new alfa_host_index();
    for (int i = 0; i < n_seq; i++) {
        std::string s(seqs[i], (size_t)seq_lens[i]);
        for (auto& c : s) c = (char)std::tolower((unsigned char)c);
        if (ix->S.length() > 0) ix->S += '`';
        ix->startpos.push_back((int64_t)ix->S.length());
        ix->S += s;
        ix->descr.push_back(names[i]);
    }
    return finish_build(ix, K, suflink, child, kmer, out);
}

int alfa_host_index_save(const alfa_host_index* ix, const char* prefix) {
    if (!ix || !prefix) return alfa::fail(ALFA_EINVAL, "null argument");
    std::string b(prefix);
    {
        std::ofstream a((b + ".aux").c_str());
        if (!a.is_open()) return alfa::fail(ALFA_EINVAL, "cannot write " + b + ".aux");
        a << ix->N << "\n" << ix->K << "\n" << ix->logN << "\n" << ix->NKm1 << "\n"
          << (ix->hasSufLink ? 1 : 0) << "\n" << (ix->hasChild ? 1 : 0) << "\n" << (ix->hasKmer ? 1 : 0) << "\n"
          << ix->startpos.size() << "\n";
        for (auto p : ix->startpos) a << p << "\n";
        for (auto& d : ix->descr) a << d << "\n";
    }
    {
        std::ofstream r((b + ".ref").c_str(), std::ios::binary);
        r.write(ix->S.data(), ix->N);
        r << "\n";
    }
    if (!write_vec(b + ".sa", ix->SA)) return alfa::fail(ALFA_EINVAL, "cannot write " + b + ".sa");
    {
        FILE* f = fopen((b + ".lcp").c_str(), "wb");
        if (!f) return alfa::fail(ALFA_EINVAL, "cannot write " + b + ".lcp");
        uint32_t nl = (uint32_t)ix->LCP.size(), nm = (uint32_t)ix->Midx.size();
        fwrite(&nl, 4, 1, f); fwrite(&nm, 4, 1, f);
        fwrite(ix->LCP.data(), 1, nl, f);
        for (uint32_t i = 0; i < nm; i++) { ItemOnDisk it{(uint64_t)ix->Midx[i], ix->Mval[i], 0}; fwrite(&it, sizeof(it), 1, f); }
        fclose(f);
    }
    if (ix->hasSufLink && !write_vec(b + ".isa", ix->ISA)) return alfa::fail(ALFA_EINVAL, "cannot write .isa");
    if (ix->hasChild && !write_vec(b + ".child", ix->CHILD)) return alfa::fail(ALFA_EINVAL, "cannot write .child");
    if (ix->hasKmer) {
        FILE* f = fopen((b + ".kmer").c_str(), "wb");
        if (!f) return alfa::fail(ALFA_EINVAL, "cannot write .kmer");
        uint32_t nk = (uint32_t)(ix->KMR.size() / 2);
        fwrite(&nk, 4, 1, f);
        fwrite(ix->KMR.data(), 8, nk, f);
        fclose(f);
    }
    return ALFA_OK;
}

int alfa_host_index_load(const char* prefix, alfa_host_index** out) {
    if (!prefix || !out) return alfa::fail(ALFA_EINVAL, "null argument");
    std::string b(prefix);
    std::ifstream a((b + ".aux").c_str());
    if (!a.is_open()) return alfa::fail(ALFA_EINVAL, "unable to open " + b + ".aux");
    alfa_host_index* ix = new alfa_host_index();
    std::string line;
    auto next_long = [&]() { std::getline(a, line); return (int64_t)atol(line.c_str()); };
    ix->N = next_long(); ix->K = next_long(); ix->logN = next_long(); ix->NKm1 = next_long();
    ix->hasSufLink = next_long() == 1; ix->hasChild = next_long() == 1; ix->hasKmer = next_long() == 1;
    int nchr = (int)next_long();
    for (int i = 0; i < nchr; i++) ix->startpos.push_back(next_long());
    for (int i = 0; i < nchr; i++) { std::getline(a, line); ix->descr.push_back(line); }
    {
        std::ifstream r((b + ".ref").c_str(), std::ios::binary);
        std::getline(r, ix->S);
    }
    bool ok = read_vec(b + ".sa", ix->SA);
    {
        FILE* f = fopen((b + ".lcp").c_str(), "rb");
        if (!f) ok = false;
        else {
            uint32_t nl = 0, nm = 0;
            ok = ok && fread(&nl, 4, 1, f) == 1 && fread(&nm, 4, 1, f) == 1;
            ix->LCP.resize(nl);
            if (nl) ok = ok && fread(ix->LCP.data(), 1, nl, f) == nl;
            for (uint32_t i = 0; i < nm && ok; i++) {
                ItemOnDisk it;
                ok = fread(&it, sizeof(it), 1, f) == 1;
                ix->Midx.push_back((int64_t)it.idx); ix->Mval.push_back(it.val);
            }
            fclose(f);
        }
    }
    if (ix->hasSufLink) ok = ok && read_vec(b + ".isa", ix->ISA);
    if (ix->hasChild) ok = ok && read_vec(b + ".child", ix->CHILD);
    if (ix->hasKmer) {
        FILE* f = fopen((b + ".kmer").c_str(), "rb");
        if (!f) ok = false;
        else {
            uint32_t nk = 0;
            ok = ok && fread(&nk, 4, 1, f) == 1;
            ix->KMR.resize(2 * (size_t)nk);
            if (nk) ok = ok && fread(ix->KMR.data(), 8, nk, f) == nk;
            fclose(f);
        }
    }
    if (!ok || (int64_t)ix->S.size() != ix->N) { delete ix; return alfa::fail(ALFA_EINVAL, "index files of " + b + " are missing or truncated"); }
    ix->S.append(16, '\0');
    ix->refresh_view();
    *out = ix;
    return ALFA_OK;
}

// the host side's share of an index whose arrays go straight to device memory (alfa_index_load_to_device): .aux and .ref only
int alfa_host_index_load_text(const char* prefix, alfa_host_index** out) {
    if (!prefix || !out) return alfa::fail(ALFA_EINVAL, "null argument");
    std::string b(prefix);
    std::ifstream a((b + ".aux").c_str());
    if (!a.is_open()) return alfa::fail(ALFA_EINVAL, "unable to open " + b + ".aux");
    alfa_host_index* ix = new alfa_host_index();
    std::string line;
    auto next_long = [&]() { std::getline(a, line); return (int64_t)atol(line.c_str()); };
    ix->N = next_long(); ix->K = next_long(); ix->logN = next_long(); ix->NKm1 = next_long();
    ix->hasSufLink = next_long() == 1; ix->hasChild = next_long() == 1; ix->hasKmer = next_long() == 1;
    int nchr = (int)next_long();
    for (int i = 0; i < nchr; i++) ix->startpos.push_back(next_long());
    for (int i = 0; i < nchr; i++) { std::getline(a, line); ix->descr.push_back(line); }
    {
        std::ifstream r((b + ".ref").c_str(), std::ios::binary);
        std::getline(r, ix->S);
    }
    if ((int64_t)ix->S.size() != ix->N || ix->N <= 0 || nchr <= 0) { delete ix; return alfa::fail(ALFA_EINVAL, "index files of " + b + " are missing or truncated"); }
    ix->S.append(16, '\0');
    ix->refresh_view();
    ix->view.SA = nullptr; ix->view.LCP = nullptr; ix->view.n = 0;       // (no arrays on the host: alfa_index_upload refuses this view)
    *out = ix;
    return ALFA_OK;
}

const alfa_index_view* alfa_host_index_view(const alfa_host_index* ix) { return ix ? &ix->view : nullptr; }
const char* alfa_host_index_descr(const alfa_host_index* ix, int seq) {
    return (ix && seq >= 0 && seq < (int)ix->descr.size()) ? ix->descr[seq].c_str() : nullptr;
}
void alfa_host_index_from_set(const alfa_host_index* ix, int64_t hit, int64_t* seq, int64_t* seqpos) {
    auto it = std::upper_bound(ix->startpos.begin(), ix->startpos.end(), hit);
    *seq = (int64_t)(it - ix->startpos.begin()) - 1;
    --it;
    *seqpos = hit - *it;
}
void alfa_host_index_chrom_bounds(const alfa_host_index* ix, int64_t hit, int64_t* chr_start, int64_t* chr_end) {
    int64_t seq, pos;
    alfa_host_index_from_set(ix, hit, &seq, &pos);
    *chr_start = ix->startpos[seq];
    if (ix->startpos.size() > 1 && seq < (int64_t)ix->startpos.size() - 1) *chr_end = ix->startpos[seq + 1] - 1;
    else *chr_end = ix->N - 1;
}
void alfa_host_index_free(alfa_host_index* ix) { delete ix; }

}  // extern "C"
