// compat_main.cpp -- exercises include/alfa_compat.hpp (the reference's call signatures over libalfa_hot) and prints what
// the calls return; tests/test_gpu_compat.py compares the text with the oracle's answers for the same inputs.
// usage: _compat <ref.fa> <K> <reads.txt>   (one read per line; every call is a batch of one)
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "../include/alfa_compat.hpp"
#include "../include/alfa_index.h"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    alfa_host_index* h = nullptr;
    if (alfa_host_index_build_fasta(argv[1], atoi(argv[2]), 0, 1, 1, &h)) { fprintf(stderr, "%s\n", alfa_last_error()); return 1; }
    alfa_compat::sparseSA sa(alfa_host_index_view(h));
    int8_t mat[25];
    for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[k++] = (i == 4 || j == 4) ? -1 : (i == j ? 1 : -4);
    std::ifstream in(argv[3]);
    std::string line;
    std::vector<std::string> reads;
    while (std::getline(in, line)) reads.push_back(line);
    for (size_t r = 0; r < reads.size(); ++r) {
        std::vector<alfa_compat::match_t> m;
        m.push_back(alfa_compat::match_t(-7, -7, -7));        // the calls append: this sentinel must survive
        sa.SMAM(reads[r], m, 20, 1000, 1);
        sa.MEM(reads[r], m, 25, 1000, 1);
        sa.SMEM(reads[r], m, 25, 1000, 1, 10);
        sa.SMEMP(reads[r], m, 25, 1000, 1, 10);
        printf("R %zu %zu", r, m.size());
        for (auto& x : m) printf(" %ld,%d,%d", x.ref, x.query, x.len);
        printf("\n");
    }
    // DP: each read against the next one
    auto nt4 = [](const std::string& s) { std::vector<uint8_t> v; for (char c : s) v.push_back(c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : c == 't' ? 3 : 4); return v; };
    for (size_t r = 0; r + 1 < reads.size(); ++r) {
        std::vector<uint8_t> q = nt4(reads[r]), t = nt4(reads[r + 1]);
        if (q.empty() || t.empty()) continue;
        int qle, tle, gtle, gscore, max_off, nc = 0;
        uint32_t* cig = nullptr;
        const int w = 30 + (int)(q.size() > t.size() ? q.size() - t.size() : t.size() - q.size());
        int se = alfa_compat::ksw_extend((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, 6, 1, w, 100, 17, &qle, &tle, &gtle, &gscore, &max_off);
        int sn = alfa_compat::ksw_nw((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, 6, 1, w);
        int sg = alfa_compat::ksw_global((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, 6, 1, w, &nc, &cig);
        printf("D %zu %d %d %d %d %d %d %d %d %d", r, se, qle, tle, gtle, gscore, max_off, sn, sg, nc);
        for (int k = 0; k < nc; ++k) printf(" %u", cig[k]);
        printf("\n");
        free(cig);
        // ksw_align as src/mapper.cpp:1570-1572 calls it: negative gap numbers, KSW_XSTART, byte scores for short queries
        const int xtra = alfa_compat::KSW_XSTART | ((int)q.size() * 1 < 250 ? alfa_compat::KSW_XBYTE : 0);
        alfa_compat::kswr_t a = alfa_compat::ksw_align((int)q.size(), q.data(), (int)t.size(), t.data(), 5, mat, -6, -1, xtra, 0);
        printf("A %zu %d %d %d %d %d %d %d\n", r, a.score, a.te, a.qe, a.score2, a.te2, a.tb, a.qb);
    }
    // dynProg with the reference's call shapes (src/mapper.cpp:424-428 is its one call site): each read against the next one,
    // a window of the target picked by the boundaries; ASCII strings
    {
        alfa_compat::dp_scores scores(1, -4, -6, -1);
        scores.updateScoreMatrixDna();
        alfa_compat::dynProg dp(2000, true, scores);
        alfa_compat::dp_scores lin(2, -3, 0, -2);
        alfa_compat::dynProg dpl(2000, false, lin);
        for (size_t r = 0; r + 1 < reads.size(); ++r) {
            const std::string &q = reads[r], &t = reads[r + 1];
            if (q.size() < 20 || t.size() < 20) continue;
            for (int v = 0; v < 4; ++v) {
                alfa_compat::boundaries b(3, (long)t.size() - 4, 2, (int)q.size() - 3);
                alfa_compat::dp_type type(v == 1, v == 1, false, v == 3, false);
                alfa_compat::dp_output out;
                int ret = -1;
                try {
                    const int d = (int)(t.size() > q.size() ? t.size() - q.size() : q.size() - t.size());
                    if (v == 0) ret = dp.dpBandStatic(t, q, b, type, alfa_compat::ALL, out, 10, 10 + d + 50);
                    else if (v == 1) ret = dp.dpBandFull(t, q, b, type, alfa_compat::ALL, out, 12 + d, 12 + d);
                    else if (v == 2) ret = dp.dpBandScore(t, q, b, type, alfa_compat::ALL, out, 15 + d, 15 + d);
                    else ret = dpl.dpBasic(t, q, b, type, alfa_compat::ALL, out);
                } catch (const std::exception& e) { ret = 9; }
                printf("P %zu %d %d %d %d %ld %ld %d %d ", r, v, ret, out.dpScore, out.editDist, b.refB, b.refE, b.queryB, b.queryE);
                for (size_t k = 0; k < out.cigarChars.size(); ++k) printf("%d%c", out.cigarLengths[k], out.cigarChars[k]);
                printf("\n");
            }
        }
    }
    alfa_host_index_free(h);
    return 0;
}
