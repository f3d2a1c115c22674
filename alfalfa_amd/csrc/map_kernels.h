// map_kernels.h -- launcher of the per-read mapping kernel (map_kernels.hip) for seed_kernels.hip's entry point
// alfa_map_reads_batch: seeds and candidate regions of both strands are on the device already (seed kernels, regions_kernel);
// one wavefront per read runs the reference's trial loop over them (csrc/map_core.h) with the banded DP inside the kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/alfa_hot.h"
#include "map_core.h"

namespace alfa {

struct MapKernelArgs {
    // text
    const uint8_t* S;               // S[0]
    int64_t N;
    const int64_t* startpos;        // device copy of the sequences' start offsets
    int32_t n_seq;
    // reads: strand 2r is the forward read of r, 2r + 1 its reverse complement
    const uint8_t* reads;           // packed read bytes
    const int64_t* read_off;        // [2n] offset of every strand in `reads`
    const mapcore::ReadParams* rp;  // [n]
    int32_t n;
    // seeds -> regions (regions_kernel's outputs, slotted per strand)
    const int64_t* base;            // [2n + 1] first match of every strand
    const alfa_region* reg;         // strand s: reg[base[s] ..]
    const alfa_match* members;      // strand s: members[2 * base[s] + 8 * s ..]
    const int32_t* nreg;            // [2n]
    const int32_t* rstatus;         // [2n] regions_kernel's status
    // DP
    const void* dp_params;          // DevParams (ksw_wave.h)
    // scratch of the read in flight, one slot per WORKGROUP of the launch (`slots` of them; a workgroup takes reads from
    // *next until none is left, so the scratch is sized by what the GPU holds at a time, not by the batch)
    int32_t slots;
    int32_t* next;                  // the next read to take; 0 at launch
    const int32_t* order;           // the order in which the reads are taken (longest first: the last ones to start are short), or null
    int32_t* reasons;               // [16] diagnostics: reads handed back, by cause (map_kernel's reason codes); 0 at launch
    uint8_t* scratch;               // slots of scratch_stride bytes: nt4 query and target of the task in flight
    int32_t scratch_stride, qcap;   // a slot holds a query of up to qcap bytes at 16 and a target behind it
    uint8_t* zscratch;              // slots of zslot bytes: zstride direction bytes of the ksw_global call in flight (pieces beyond the
    int64_t zstride, zslot;         // kernel's LDS tile), then room for bam_cap BAM operations of its walk
    int32_t bam_cap;
    mapcore::Options opt;
    // results, slotted per read
    int32_t* status;                // [n] mapcore::Status
    int32_t* naln;                  // [n]
    int32_t* nseed;                 // [n]
    int32_t* nops;                  // [n]
    mapcore::AlnOut* alns;          // [n * ALN_CAP]
    alfa_match* seeds;              // [n * CHAINOUT_CAP]
    mapcore::CigOp* ops;            // [n * ops_cap]
    int32_t ops_cap;                // room for a read's CIGAR operations (>= mapcore::OPS_CAP)
};

// workgroups one launch can keep resident (what `slots` is worth at most)
int map_resident_workgroups();
void launch_map(hipStream_t st, const MapKernelArgs& a);
// exclusive prefix sums of three int32 arrays of n1 entries in one launch (for the arrays of a call: up to ~10^6 entries; beyond, a library scan)
void launch_map_offsets(hipStream_t st, int n1, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* oa, int32_t* ob, int32_t* oc);
// dense copies of the slotted results: read r's alignments to out_alns[aln_off[r] ..], its chain seeds to out_seeds[seed_off[r] ..],
// its CIGAR operations to out_ops[op_off[r] ..]
void launch_map_compact(hipStream_t st, int n, const int32_t* naln, const int32_t* nseed, const int32_t* nops, const int32_t* aln_off,
                        const int32_t* seed_off, const int32_t* op_off, const mapcore::AlnOut* alns, const alfa_match* seeds,
                        const mapcore::CigOp* ops, int ops_cap, mapcore::AlnOut* out_alns, alfa_match* out_seeds, mapcore::CigOp* out_ops);

}  // namespace alfa
