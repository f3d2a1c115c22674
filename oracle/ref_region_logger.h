/* oracle/ref_region_logger.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Force-included (-include) in front of the reference's mapper.cpp, compiled where it lies (oracle/Makefile, target
 * _ref/alfalfa_reglog).  calculateCandRegions (src/mapper.cpp:1220-1267) ends with the unqualified call
 *     sort(lisIntervals.begin(), lisIntervals.end(), compIntervals);
 * which normally resolves to std::sort.  The non-template overload below is an exact match for those arguments and wins
 * overload resolution, so the regions of EVERY calculateCandRegions call -- as identifyAndFilter (:578-704) and coveredBases
 * (:706-719) left them, in creation order, before the sort by score -- can be written to $ALFA_REF_REGION_LOG; it then sorts
 * exactly as the reference would.  Nothing of the reference is copied or patched.
 * Record (text): "R <number of regions>", then per region "G <fw> <refBegin> <refEnd> <score> <n>" and n lines "<ref> <query> <len>".
 */
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "mapper.h"

inline void sort(std::vector<lis_t>::iterator first, std::vector<lis_t>::iterator last, bool (*before)(const lis_t&, const lis_t&)) {
    static FILE* f = getenv("ALFA_REF_REGION_LOG") ? fopen(getenv("ALFA_REF_REGION_LOG"), "w") : 0;
    static std::mutex mu;
    if (f) {
        std::lock_guard<std::mutex> g(mu);
        fprintf(f, "R %ld\n", (long)(last - first));
        for (std::vector<lis_t>::iterator it = first; it != last; ++it) {
            fprintf(f, "G %d %ld %ld %d %zu\n", it->fw ? 1 : 0, it->refBegin, it->refEnd, it->score, it->matches.size());
            for (size_t k = 0; k < it->matches.size(); ++k) fprintf(f, "%ld %d %d\n", it->matches[k].ref, it->matches[k].query, it->matches[k].len);
        }
        fflush(f);
    }
    std::sort(first, last, before);
}
