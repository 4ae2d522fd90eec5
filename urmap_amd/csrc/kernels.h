// kernels.h -- launch wrappers of the gfx950 device code (kernels.hip), called by the C ABI (urmapx.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/urmapx.h"

namespace urx {

// Device-resident view of a UFIndex (ufindex.h:97-131): two big read-only arrays + the directory.
struct DevIndex {
	const uint8_t *blob;        // 5*slotCount bytes (+8 pad): {tally u8, pos u32 LE} per slot
	const uint8_t *seq;         // seqDataSize ASCII bytes (+4096 zero pad)
	uint64_t slotCount;
	uint64_t slotMagic;         // floor(2^64 / slotCount): Barrett reciprocal for h % slotCount
	uint64_t shiftMask;
	uint32_t W;
	uint32_t maxIx;
	uint32_t seqDataSize;
	uint32_t seqCount;
	const uint32_t *seqLengths; // device
	const uint32_t *seqOffsets; // device
};

// per-k-mer output of the seed+probe stage, SoA; index = 2*offs[r] + strand*L + qpos
struct ProbeOut {
	uint64_t *slots;
	uint8_t *tallies;
	uint32_t *positions;
};

// random slot gathers: blocks x 256 threads x iters x 8 loads
hipError_t launch_gather_bench(const DevIndex &X, uint32_t blocks, uint32_t iters, uint32_t *d_sink, hipStream_t s);

hipError_t launch_seed_probe(const DevIndex &X, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t n,
                             uint32_t max_read_len, ProbeOut out, hipStream_t s);

// workspace of the persistent search kernels: per-block global scratch (+ optional diagnostics buffer)
struct SearchWork {
	uint32_t *stats;  // optional cycle-stamp accumulator (URMAPX_PHASE_STATS), else nullptr
	uint8_t *scratch;
	size_t scratch_stride;  // search_scratch_stride(max_read_len)
	int blocks;             // search_block_count(max_read_len, device)
	uint32_t *ticket;       // device word: work counter of the launch (zeroed by the launcher)
	int hsp_lds_cap = 0;    // 0 = default; test aid (URMAPX_TEST_HSP_LDS_CAP) to exercise the HSP overflow list
	uint32_t *ovf_list = nullptr;  // device: [0] = count, [1..n] = reads queued for the second pass
};
size_t search_scratch_stride(uint32_t max_read_len);
size_t search_scratch_tail();
size_t search_pe_scratch_tail();
int search_block_count(uint32_t max_read_len, int device);
size_t viterbi_batch_scratch_stride();

hipError_t launch_search_se(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t n, uint32_t max_read_len, ProbeOut probe, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, hipStream_t s);

// paired-end (kernels_pe.hip): reads 2i and 2i+1 of the batch are the two mates of pair i
size_t search_pe_scratch_stride(uint32_t max_read_len);
int search_pe_block_count(uint32_t max_read_len, int device);
hipError_t launch_search_pe(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t npairs, uint32_t max_read_len, ProbeOut probe, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, int veryfast,
                            urmapx_pair_info *pair_info, hipStream_t s);

hipError_t launch_viterbi_batch(const urmapx_params &P, const uint8_t *d_a, const uint32_t *d_aoffs,
                                const uint8_t *d_b, const uint32_t *d_boffs, const uint8_t *d_flags, uint32_t n,
                                float *d_scores, uint8_t *d_status, urmapx_path_op *d_ops, uint16_t *d_nops,
                                uint8_t *d_scratch, hipStream_t s);

}  // namespace urx
