/* oracle/align_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see alfa_oracle.h).
 *
 * Scalar restatement of ksw_align (src/ksw.c:330-352) = ksw_qinit (:58-104) + ksw_u8 (:106-221) / ksw_i16 (:223-321).
 * The reference is a striped (Farrar) SSE2 kernel whose results depend on the striping in three places, so the
 * restatement keeps the striped iteration space (P lanes x slen vectors) and replaces each SSE2 instruction by its
 * per-lane arithmetic:
 *   - E(i+1,.) is computed from H before the lazy-F corrections (:158-161, comment at :168);
 *   - the lazy-F loop stops at the first vector after which no lane can improve (:177-178, :281) and never feeds
 *     the corrected H back into F;
 *   - padded query positions (score 0) take part in the row maximum (:155,:183) and the qe scan runs in memory
 *     order, vector-major (:206-207, :307-308).
 * The gap penalties are used as given: mapper.cpp:1572,2197 pass NEGATIVE numbers, which the saturating unsigned
 * subtractions (:158-163, :265-269) turn into "gaps never pay" -- reproduced by doing the same unsigned arithmetic.
 * Pinned against the compiled reference (oracle/_ref) on tests/golden/ksw_align.json.gz (tools/make_align_goldens.py).
 */
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "alfa_oracle.h"

namespace {

thread_local int64_t g_cells_align = 0;

struct Kswr { int score, te, qe, score2, te2, tb, qb; };

/* per-lane arithmetic of the two instantiations */
struct U8 {                                   // ksw_u8: 16 unsigned bytes per vector
    typedef uint8_t T;
    static constexpr int P = 16;
    static T gap(int g) { return (uint8_t)g; }                                  // _mm_set1_epi8 (:127-128)
    static T add_score(T h, T prof, T shift) {                                  // :150-151
        int v = std::min(255, (int)h + prof);
        return (T)std::max(0, v - shift);
    }
    static T prof(int s, int shift) { return (uint8_t)(int8_t)(s + shift); }    // :91 (stored through int8_t*)
    static T mx(T a, T b) { return std::max(a, b); }                            // _mm_max_epu8
    static T subs(T a, T b) { return a > b ? (T)(a - b) : (T)0; }               // _mm_subs_epu8
    static bool f_done(T f, T h) { return subs(f, h) == 0; }                    // :177
    static int as_int(T v) { return v; }
};
struct I16 {                                  // ksw_i16: 8 signed shorts per vector
    typedef int16_t T;
    static constexpr int P = 8;
    static T gap(int g) { return (int16_t)g; }                                  // _mm_set1_epi16 (:243-244)
    static T add_score(T h, T prof, T) {                                        // _mm_adds_epi16 (:259)
        int v = (int)h + prof;
        return (T)std::max(-32768, std::min(32767, v));
    }
    static T prof(int s, int) { return (int16_t)s; }                            // :100
    static T mx(T a, T b) { return std::max(a, b); }                            // _mm_max_epi16 (signed)
    static T subs(T a, T b) {                                                   // _mm_subs_epu16 (UNSIGNED, :265-269)
        uint16_t x = (uint16_t)a, y = (uint16_t)b;
        return (T)(x > y ? (uint16_t)(x - y) : (uint16_t)0);
    }
    static bool f_done(T f, T h) { return !(f > h); }                           // _mm_cmpgt_epi16 (:281)
    static int as_int(T v) { return (uint16_t)v; }                              // :306 reads Hmax as uint16_t
};

/* ksw_qinit (:58-104): shift, max and the striped profile; prof[(a*slen + j)*P + k] scores residue a against
 * query position j + k*slen (0 past the end) */
template <class V>
struct Profile {
    int qlen, slen, shift, max;
    std::vector<typename V::T> prof;
    Profile(int qlen_, const uint8_t* query, int m, const int8_t* mat) : qlen(qlen_) {
        slen = (qlen + V::P - 1) / V::P;
        uint8_t sh = 127, md = 0;
        for (int a = 0; a < m * m; ++a) {                       // :75-78, with the reference's uint8_t/int8_t casts
            if (mat[a] < (int8_t)sh) sh = (uint8_t)mat[a];
            if (mat[a] > (int8_t)md) md = (uint8_t)mat[a];
        }
        max = md;
        shift = (uint8_t)(256 - sh);
        prof.resize((size_t)std::max(1, slen) * V::P * m);
        for (int a = 0; a < m; ++a)
            for (int j = 0; j < slen; ++j)
                for (int k = 0; k < V::P; ++k) {
                    int p = j + k * slen;
                    prof[((size_t)a * slen + j) * V::P + k] = V::prof(p >= qlen ? 0 : mat[a * m + query[p]], shift);
                }
    }
};

template <class V>
Kswr striped_pass(const Profile<V>& q, int tlen, const uint8_t* target, int gapo, int gape, int xtra) {
    typedef typename V::T T;
    constexpr int P = V::P;
    const int slen = q.slen;
    Kswr r = {0, -1, -1, -1, -1, -1, -1};                                        // g_defr (:39)
    const int minsc = (xtra & 0x40000) ? (xtra & 0xffff) : 0x10000;              // KSW_XSUBO (:123)
    const int endsc = (xtra & 0x20000) ? (xtra & 0xffff) : 0x10000;              // KSW_XSTOP (:124)
    const T gapoe = V::gap(gapo + gape), gpe = V::gap(gape), shift = (T)q.shift;
    std::vector<T> H0((size_t)slen * P + P, 0), H1((size_t)slen * P + P, 0), E((size_t)slen * P + P, 0), Hmax((size_t)slen * P + P, 0);
    std::vector<std::pair<int, int>> b;                                          // (imax, row) runs (:184-192)
    int gmax = 0, te = -1;
    T* h0 = H0.data();
    T* h1 = H1.data();
    for (int i = 0; i < tlen; ++i) {
        T f[P], h[P], mx[P];
        const T* S = q.prof.data() + (size_t)target[i] * slen * P;
        for (int k = 0; k < P; ++k) f[k] = 0, mx[k] = 0;
        h[0] = 0;                                                                // :141-142: last vector shifted by one lane
        for (int k = 1; k < P; ++k) h[k] = slen ? h0[(size_t)(slen - 1) * P + k - 1] : 0;
        for (int j = 0; j < slen; ++j)                                           // :143-167
            for (int k = 0; k < P; ++k) {
                const size_t x = (size_t)j * P + k;
                T hh = V::add_score(h[k], S[x], shift);
                T e = E[x];
                hh = V::mx(V::mx(hh, e), f[k]);
                mx[k] = V::mx(mx[k], hh);
                h1[x] = hh;
                hh = V::subs(hh, gapoe);
                e = V::mx(V::subs(e, gpe), hh);
                E[x] = e;
                f[k] = V::mx(V::subs(f[k], gpe), hh);
                h[k] = h0[x];
            }
        bool done = false;                                                       // lazy F (:169-180, :273-283)
        for (int it = 0; it < 16 && !done; ++it) {
            for (int k = P - 1; k > 0; --k) f[k] = f[k - 1];
            f[0] = 0;
            for (int j = 0; j < slen && !done; ++j) {
                bool all = true;
                for (int k = 0; k < P; ++k) {
                    const size_t x = (size_t)j * P + k;
                    T hh = V::mx(h1[x], f[k]);
                    h1[x] = hh;
                    hh = V::subs(hh, gapoe);
                    f[k] = V::subs(f[k], gpe);
                    all = all && V::f_done(f[k], hh);
                }
                done = all;
            }
        }
        g_cells_align += (int64_t)q.qlen;
        int imax = 0;
        if (P == 16) { for (int k = 0; k < P; ++k) imax = std::max(imax, (int)(uint8_t)mx[k]); }          // :113-119
        else { int16_t v = (int16_t)mx[0]; for (int k = 1; k < P; ++k) v = std::max(v, (int16_t)mx[k]); imax = (uint16_t)v; }  // :230-235 (_mm_extract_epi16 zero-extends)
        if (imax >= minsc) {
            if (b.empty() || b.back().second + 1 != i) b.push_back({imax, i});
            else if (b.back().first < imax) b.back() = {imax, i};
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            std::copy(h1, h1 + (size_t)slen * P, Hmax.begin());
            if (P == 16 ? (gmax + q.shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;              // :197, :299
        }
        std::swap(h0, h1);
    }
    r.score = (P == 16) ? (gmax + q.shift < 255 ? gmax : 255) : gmax;                                    // :201, :303
    r.te = te;
    if (P == 8 || r.score != 255) {
        int best = -1;
        if (P == 8) r.qe = -1;
        for (int x = 0; x < slen * P; ++x)                                        // memory order: vector j = x/P, lane k = x%P
            if (V::as_int(Hmax[x]) > best) best = V::as_int(Hmax[x]), r.qe = x / P + x % P * slen;
        if (!b.empty()) {                                                         // :209-217 (q->max == 0 would divide by zero there)
            int d = (r.score + q.max - 1) / q.max, low = te - d, high = te + d;
            for (auto& e : b)
                if ((e.second < low || e.second > high) && e.first > r.score2) r.score2 = e.first, r.te2 = e.second;
        }
    }
    return r;
}

template <class V>
Kswr align_sized(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int xtra) {
    Profile<V> q(qlen, query, m, mat);
    Kswr r = striped_pass<V>(q, tlen, target, gapo, gape, xtra);
    if ((xtra & 0x80000) == 0 || ((xtra & 0x40000) && r.score < (xtra & 0xffff))) return r;              // :343
    // qe can name a padded lane when a gap costs nothing (gapo + gape == 0): the reference then reverses and reads memory
    // behind its query buffer (:344-345) -- undefined there; tb/qb stay -1 here and in the library
    if (r.qe >= qlen) return r;
    // :344-348 -- the prefixes are reversed in place and the second pass still runs over all tlen rows
    std::vector<uint8_t> rq(query, query + qlen), rt(target, target + tlen);
    std::reverse(rq.begin(), rq.begin() + (r.qe + 1));
    std::reverse(rt.begin(), rt.begin() + (r.te + 1));
    Profile<V> q2(r.qe + 1, rq.data(), m, mat);
    Kswr rr = striped_pass<V>(q2, tlen, rt.data(), gapo, gape, 0x20000 | r.score);
    if (r.score == rr.score) r.tb = r.te - rr.te, r.qb = r.qe - rr.qe;                                   // :349-350
    return r;
}

}  // namespace

extern "C" {

void orc_ksw_align(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat,
                   int gapo, int gape, int xtra, alfa_kswr* out) {
    Kswr r = (xtra & 0x10000) ? align_sized<U8>(qlen, query, tlen, target, m, mat, gapo, gape, xtra)    // KSW_XBYTE (:337)
                              : align_sized<I16>(qlen, query, tlen, target, m, mat, gapo, gape, xtra);
    std::memcpy(out, &r, sizeof r);
}

int orc_ksw_align_batch(const alfa_dp_params* p, const alfa_align_task* t, int n, alfa_kswr* r) {
    for (int i = 0; i < n; ++i)
        orc_ksw_align(t[i].qlen, t[i].query, t[i].tlen, t[i].target, p->m, p->mat, p->gapo, p->gape, t[i].xtra, &r[i]);
    return 0;
}

int64_t orc_cells_align_take(void) { int64_t v = g_cells_align; g_cells_align = 0; return v; }

}
