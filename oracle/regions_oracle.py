"""oracle/regions_oracle.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's candidate-region building, the checker
of alfa_seed_regions_batch.  Pure Python, for the few thousand read strands a test holds.

  identify_and_filter   identifyAndFilter  /root/reference/src/mapper.cpp:578-704
  covered_bases         coveredBases       /root/reference/src/mapper.cpp:706-719
Pinned through the host mapper: its C++ restatement of the same two functions (alfalfa_amd/csrc/mapper/mapper.cpp) writes SAM
that is byte-identical to the compiled reference's (tests/test_mapper_cpu.py), and tests/test_regions_oracle.py compares this
file with the regions that code builds for every read of the lambda examples and of the synthetic set (ALFA_REGION_LOG), on the
CPU.  tests/test_gpu_regions.py then checks the device kernel against this file and the command line with --device-regions
against the reference's SAM.

std::sort is not stable: where elements compare equal under the reference's comparator but differ, the reference's own order
is an accident of the library.  `ambiguous` is set for such a strand; the device hands those back to the host (status 1)."""


def identify_and_filter(matches, q_length, edit_dist, max_smem_occ=10000, max_region_mems=20, min_mem_reg_length=50):
    """matches: list of (ref, query, len).  Returns (regions, ambiguous); a region is [ref_begin, ref_end, members]."""
    if max_smem_occ < 0:
        max_smem_occ = 1 << 62            # the reference holds it in a size_t
    if max_region_mems < 0:
        max_region_mems = 1 << 62
    regions = []
    if not matches:
        return regions, False
    ambiguous = False
    ms = sorted(matches, key=lambda m: (m[1], -m[2]))                   # compMatchesSMEM :574 (ties re-sorted below)
    smems, reserves = [], []
    previous_r_end, current_range, index = -1, 0, 0
    while index < len(ms):                                              # :600-640
        start = index
        ps, pe = ms[index][1], ms[index][1] + ms[index][2]
        index += 1
        while index < len(ms) and ms[index][1] == ps and ms[index][1] + ms[index][2] == pe:
            index += 1
        if pe > previous_r_end:
            current_range = index - start
        else:
            current_range += index - start
        if (pe > previous_r_end and current_range <= max_smem_occ) or (current_range <= max_region_mems and pe - ps >= min_mem_reg_length):
            smems.extend(ms[start:index])
        else:
            reserves.extend(ms[start:index])
        previous_r_end = pe

    def by_ref(lst):                                                    # compMatches :570
        out = sorted(lst, key=lambda m: (m[0], m[1]))
        for a, b in zip(out, out[1:]):
            if a[0] == b[0] and a[1] == b[1] and a[2] != b[2]:
                return out, True
        return out, False
    smems, amb = by_ref(smems)
    ambiguous |= amb
    begin = 0
    while begin < len(smems):                                           # :644-676
        ref_end = smems[begin][0] - smems[begin][1] + q_length + edit_dist
        longest, end = begin, begin + 1
        while end < len(smems) and smems[end][0] + smems[end][2] <= ref_end:
            if smems[longest][2] < smems[end][2]:
                longest = end
                ref_end = smems[longest][0] - smems[longest][1] + q_length + edit_dist
            end += 1
        ref_begin = smems[longest][0] - smems[longest][1] - edit_dist
        begin -= 1
        begin_fix = begin
        while begin >= 0 and smems[begin][0] >= ref_begin:
            begin -= 1
        if not regions or regions[-1][1] < ref_begin:
            regions.append([ref_begin, ref_end, list(smems[begin + 1:end])])
        else:
            regions[-1][2].extend(smems[begin_fix + 1:end])
            regions[-1][0] = min(regions[-1][0], ref_begin)
            regions[-1][1] = min(regions[-1][1], ref_end)               # min for both ends, :673-674
        begin = end
    if regions:                                                         # :688-700
        reserves, amb = by_ref(reserves)
        ambiguous |= amb
        cr = ci = 0
        while cr < len(reserves) and ci < len(regions):
            if reserves[cr][0] < regions[ci][0]:
                cr += 1
            elif regions[ci][1] < reserves[cr][0]:
                ci += 1
            else:
                regions[ci][2].append(reserves[cr])
                cr += 1
    return regions, ambiguous


def covered_bases(members):
    """coveredBases: sorts the members by (query, ref) in place and returns (score, ambiguous)."""
    members.sort(key=lambda m: (m[1], m[0]))                            # compMatchesQuery mapper.h:37
    amb = any(a[0] == b[0] and a[1] == b[1] and a[2] != b[2] for a, b in zip(members, members[1:]))
    cover = members[0][2]
    prev_end = members[0][1] + members[0][2] - 1
    for m in members[1:]:
        start = max(m[1], prev_end + 1)
        stop = max(prev_end, m[1] + m[2] - 1)
        prev_end = stop
        cover += max(0, stop - start + 1)
    return cover, amb


def first_chain(members, q_length, error_percent):
    """nextChain  /root/reference/src/mapper.cpp:721-782 from the anchor extendChain tries first (:885-893: ascending
    (len << 32 | index), taken from the back).  members as covered_bases leaves them.  Returns (chain, anchor, edit, coverage)."""
    MAX_OVERLAP = -10
    anchor = max(range(len(members)), key=lambda k: (members[k][2], k))
    a = members[anchor]
    cur_edit, coverage = 0, a[2]
    fwd, back = [], []
    ref_lb, q_lb, ref_rb, q_rb = a[0], a[1], a[0] + a[2] - 1, a[1] + a[2] - 1
    for k in range(anchor + 1, len(members)):
        mr, mq, ml = members[k]
        q_dist, ref_dist = mq - (q_rb + 1), mr - (ref_rb + 1)
        q_long, r_long = mq + ml - q_lb, mr + ml - ref_lb
        if q_dist >= MAX_OVERLAP and ref_dist >= MAX_OVERLAP and abs(q_dist - ref_dist) <= int(error_percent * float(max(q_long, r_long))):
            fwd.append(k)
            cur_edit += max(1, abs(q_dist - ref_dist))
            ref_lb, ref_rb, q_lb, q_rb = mr, mr + ml - 1, mq, mq + ml - 1
            coverage += ml + min(q_dist, ref_dist) if min(q_dist, ref_dist) < 0 else ml
    cur_edit += 1 if q_rb + 1 < q_length else 0
    ref_lb, q_lb, ref_rb, q_rb = a[0], a[1], a[0] + a[2] - 1, a[1] + a[2] - 1
    for k in range(anchor - 1, -1, -1):
        mr, mq, ml = members[k]
        q_dist, ref_dist = q_lb - (mq + ml), ref_lb - (mr + ml)
        q_long, r_long = q_rb - mq + 1, ref_rb - mr + 1
        if q_dist >= MAX_OVERLAP and ref_dist >= MAX_OVERLAP and abs(q_dist - ref_dist) <= int(error_percent * float(max(q_long, r_long))):
            back.append(k)
            cur_edit += max(1, abs(q_dist - ref_dist))
            ref_lb, ref_rb, q_lb, q_rb = mr, mr + ml - 1, mq, mq + ml - 1
            coverage += ml + min(q_dist, ref_dist) if min(q_dist, ref_dist) < 0 else ml
    cur_edit += 1 if q_lb > 0 else 0
    return back[::-1] + [anchor] + fwd, anchor, cur_edit, coverage
