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
	const uint4 *seqp;          // the same bytes as 4-bit codes in bit planes, one uint4 per 32 bases (dev_common.h); device-built
	uint64_t slotCount;
	uint64_t slotMagic;         // floor(2^64 / slotCount): Barrett reciprocal for h % slotCount
	uint64_t shiftMask;
	uint32_t W;
	uint32_t maxIx;
	uint32_t seqDataSize;
	uint32_t seqCount;
	const uint32_t *seqLengths; // device
	const uint32_t *seqOffsets; // device
	// GetRow_Blob's rows laid out once per index (chain_rows.hip); rowinfo == nullptr: not built, the kernels walk hop by hop
	const uint2 *rowinfo;       // per slot: .x = row length | offset inside the slot's group of 1024 << 8, .y = the row's second position
	const uint64_t *rowbase;    // per group of 1024 slots: where its rows begin
	const uint32_t *rows;       // positions, row after row
	// Round 5: the slot table as the search kernel wants it (chain_rows.hip: build_slot16), 16 bytes per slot, never straddling a sector:
	//   .x = the slot's position (for a head whose own slot holds a long link's steps: the row's resolved first position)
	//   .y = tally | row length << 8 (row length 1 for PLUS1 / BOTH1 heads, 0 for slots that head nothing)
	//   .z = rows of two: the second position; longer rows: the row's index in `rows`
	// One gather per k-mer then brings what the probe AND the row lookup need: no rowinfo read, no rowbase read.  nullptr: not built.
	const uint4 *slot16;
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

// Phase 6 of Search_Lo (AlignHSP of every HSP still unaligned, search1m6.cpp:273-274) runs as launches of its own: the
// search kernel turns each such HSP into a DpJob and parks the read's state; dp_kernel runs the two flank DPs of every
// job (one wavefront per job: the DP outcome of an HSP does not depend on the search state, only whether it is USED
// does); finalize_se_kernel replays AlignHSP's bookkeeping over the read's jobs in order and writes the result.  A read
// in a repeat family has hundreds of such HSPs (the reference's list is unbounded): inside the one-wavefront-per-read
// search kernel they were a serial tail that set the kernel's duration.
struct DpJob {             // 32 bytes
	uint32_t read;         // 0xFFFFFFFF = slot not in use
	uint32_t startdb, pk;  // the HSP (SearchWave::hsp_pk packing)
	int32_t maxpen;        // m_MaxPenalty when the job was made: the cap only falls, so what fails it then fails it later
	// written by dp_kernel
	uint32_t combined_tlo; // hit start if the alignment is accepted
	int16_t left_score, right_score;  // flank scores, floored at the all-gap score (alignhsp.cpp:124-126,157-159)
	uint8_t nops;          // runs of the whole path (left flank, the HSP's M run, right flank; merged) in the job's ops slice
	uint8_t flags;         // DPJ_*
	uint8_t vst_l, vst_r;  // URMAPX_ST_* bits raised by the left / right DP
	uint16_t k;            // index of the job among its read's jobs (they are consumed in that order)
	uint8_t pad[2];
};
static_assert(sizeof(DpJob) == 32, "DpJob layout");
// flank window unusable (alignhsp.cpp:104-117 / 148-150); right flank not run (penalty already over the job's cap);
// path longer than URMAPX_MAX_PATH_OPS runs
static constexpr uint8_t DPJ_LEFT_FAIL = 1, DPJ_RIGHT_FAIL = 2, DPJ_RIGHT_SKIPPED = 4, DPJ_PATH_LONG = 8, DPJ_GATED = 16, DPJ_RIGHT_ABORTED = 32;
// The jobs of a read are run in rounds of growing size, [0,2) [2,16) [16,inf) by index: after each round the ordered
// replay consumes that round's jobs and the penalty cap it arrives at gates the next round's DPs -- most of a repeat
// read's HSPs fail AlignHSP's first test once the first few alignments have tightened the cap.
// Round 5: the boundaries are chosen per call (SearchWork::dp_bounds): [0,2) [2,16) [16,inf) for reads of up to 192 bases, [0,2) [2,8)
// [8,32) [32,inf) beyond (250-base reads with 5 % errors bring 13 HSPs each to phase 6: the fourth round's tighter gate is worth more
// than its two launches cost -- measured in round 4).  DP_ROUNDS = the most rounds a call may have.
static constexpr int DP_ROUNDS = 4;
struct DpBounds {
	int rounds;                    // rounds in use, 1 .. DP_ROUNDS
	uint32_t lo[DP_ROUNDS + 1];    // round rd = jobs with lo[rd] <= k < lo[rd + 1]; lo[rounds] = 0xFFFFFFFF, unused rounds are empty
};
inline DpBounds dp_bounds_default(bool long_reads) {
	DpBounds b;
	if (long_reads) { b.rounds = 4; b.lo[0] = 0; b.lo[1] = 2; b.lo[2] = 8; b.lo[3] = 32; b.lo[4] = 0xFFFFFFFFu; }
	else { b.rounds = 3; b.lo[0] = 0; b.lo[1] = 2; b.lo[2] = 16; b.lo[3] = 0xFFFFFFFFu; b.lo[4] = 0xFFFFFFFFu; }
	return b;
}
static constexpr int DP_TICKET_WORDS = 16;  // per pass: [rd] work counter of round rd, [8 + rd] length of its job list
static_assert(DP_ROUNDS <= 8, "DP_TICKET_WORDS");
static constexpr int DP_JOB_OPS = URMAPX_MAX_PATH_OPS;  // ops slice per job
// SearchWork::stage_events: [0] start, [1] search end (with phase 3 parked: the end of the resume launch), {dp end, finalize end} x DP_ROUNDS,
// second search end, {dp end, finalize end} x DP_ROUNDS, [STAGE_LAST] after the general kernel (recorded by the caller); then the two stamps
// inside the search stage when phase 3 is parked: [STAGE_P3_MAIN] end of the first search launch, [STAGE_P3_DP] end of phase 3's DP launch
static constexpr int STAGE_LAST = 3 + 4 * DP_ROUNDS;
static constexpr int STAGE_P3_MAIN = STAGE_LAST + 1, STAGE_P3_DP = STAGE_LAST + 2;
static constexpr int STAGE_EVENTS = STAGE_LAST + 3;

struct DpWork {
	DpJob *jobs = nullptr;          // jobs_cap entries
	uint16_t *ops = nullptr;        // jobs_cap * DP_JOB_OPS: the accepted alignment's path per job
	uint16_t *kidx = nullptr;       // jobs_cap: DpJob::k again, contiguous (a round scans it 64 jobs per load)
	uint32_t jobs_cap = 0;
	uint32_t *tickets = nullptr;    // [rd] work counter of dp_kernel's round rd, [8 + rd] length of the round's job list; zeroed with the counters
	uint32_t *round_list = nullptr; // DP_ROUNDS x jobs_cap: the jobs (indices) of each round, dealt out by dp_round_lists_kernel after the search
	uint32_t *counters = nullptr;   // [0] jobs made, [1] reads parked, [2] jobs the ordered replay needed, [3] jobs a round's gate dropped before their DP (statistics)
	uint32_t *fin_list = nullptr;   // per parked read: read, first job, job count, read length (16 bytes, one load)
	uint32_t *state = nullptr;      // per parked read: search state (dp_state_words(ovf) words each)
	uint32_t fin_cap = 0;
};
size_t dp_state_words(bool ovf);
size_t p3_state_words(uint32_t max_read_len);

// workspace of the persistent search kernels: per-block global scratch (+ optional diagnostics buffer)
struct SearchWork {
	uint32_t *stats;  // optional cycle-stamp accumulator (URMAPX_PHASE_STATS), else nullptr
	uint8_t *scratch;
	size_t scratch_stride;  // search_scratch_stride(max_read_len)
	int blocks;             // search_block_count(max_read_len, device)
	uint32_t *ticket;       // device word: work counter of the launch (zeroed by the launcher)
	uint32_t *ticket3 = nullptr;  // one more word: work counter of the launch over the reads parked at phase 3
	int hsp_lds_cap = 0;    // 0 = default; test aid (URMAPX_TEST_HSP_LDS_CAP) to exercise the HSP overflow list
	uint32_t *ovf_list = nullptr;  // device: [0] = count, [1..n] = reads queued for the second pass
	DpWork dp[2];                  // [0] first pass, [1] second pass; jobs == nullptr: phase 6 stays inside the search kernel
	// Round 5: phase 3 (AlignHSP when the best HSP of phases 1-2 is long, search1m6.cpp:162-171) parked the same way: the first
	// launch ends a read there, dp_kernel runs the flank DPs, and a second launch of the search kernel (PART 2) takes the read up
	// again -- replay of AlignHSP's bookkeeping, then phases 4-6 -- from its parked state: hits, scalars, top path, the HSP list and
	// the read's slot entries (p3_state_words(nch) words per read).  jobs == nullptr: phase 3 stays inside the search kernel.
	DpWork dp3;
	uint8_t *dp_scratch = nullptr; // dp_kernel's wide-band scratch: dp_blocks * dp_scratch_stride bytes
	size_t dp_scratch_stride = 0;
	int dp_blocks = 0;
	int fin_blocks = 0;            // grid of finalize_se_kernel (0: the search kernel's)
	DpBounds dp_bounds = dp_bounds_default(false);  // phase 6's rounds for this call
	hipEvent_t *stage_events = nullptr;  // optional, STAGE_EVENTS of them (see STAGE_LAST above)
};
size_t dp_scratch_stride(uint32_t max_read_len);
int dp_block_count(uint32_t max_read_len, int device);
int fin_block_count(uint32_t max_read_len, int device);
size_t search_scratch_stride(uint32_t max_read_len);
size_t search_scratch_tail(int blocks);
size_t search_pe_scratch_tail(int blocks);
int search_block_count(uint32_t max_read_len, int device);
size_t viterbi_batch_scratch_stride();

// single-end batch: the search kernel hashes and probes each read's k-mers itself (the slots of the next read are
// gathered straight into LDS while the current one is searched); no probe launch, no probe arrays
hipError_t launch_search_se(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t n, uint32_t max_read_len, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, hipStream_t s);

// the general search (kernels_slow.hip) over the reads the fast passes left flagged: any length up to qcap, lists in global
// scratch (blocks * slow_scratch_stride(qcap) bytes); list: n + 1 words, ticket: one word
size_t slow_scratch_stride(uint32_t qcap);
hipError_t launch_search_se_slow(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t n,
                                 uint32_t qcap, urmapx_result *d_results, urmapx_path_op *d_path_ops, uint32_t *d_path_used,
                                 uint32_t path_cap, uint8_t *scratch, int blocks, uint32_t *list, uint32_t *ticket, hipStream_t s);

// packed copy of the sequence store (4 bit planes per 32 bases): blocks = packed_seq_blocks(seqDataSize) uint4's
size_t packed_seq_blocks(uint32_t seq_data_size);
hipError_t launch_pack_seq(const uint8_t *d_seq, uint32_t seq_data_size, uint4 *d_out, hipStream_t s);

// paired-end (kernels_pe.hip): reads 2i and 2i+1 of the batch are the two mates of pair i
size_t search_pe_scratch_stride(uint32_t max_read_len);
int search_pe_block_count(uint32_t max_read_len, int device);
hipError_t launch_search_pe(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t npairs, uint32_t max_read_len, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, int veryfast,
                            urmapx_pair_info *pair_info, hipStream_t s);

// the general pair search (kernels_pe_slow.hip) over the pairs the fast passes left flagged: lists in global scratch
// (blocks * pe_slow_scratch_stride() bytes); list: npairs + 1 words, ticket: one word
size_t pe_slow_scratch_stride();
hipError_t launch_search_pe_slow(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t npairs,
                                 urmapx_result *d_results, urmapx_path_op *d_path_ops, uint32_t *d_path_used, uint32_t path_cap,
                                 uint8_t *scratch, int blocks, uint32_t *list, uint32_t *ticket, int veryfast, urmapx_pair_info *pair_info,
                                 int all_pairs, hipStream_t s);

// chain_rows.hip: the rows of every chain head of a resident slot table (all three null if they cannot be had)
hipError_t build_slot16_direct(const uint8_t *d_blob, uint64_t slot_count, uint32_t max_ix, uint4 **d_slot16, uint32_t **d_rows, uint64_t *total_rows);
hipError_t build_slot16(const uint8_t *d_blob, uint64_t slot_count, const uint2 *d_info, const uint64_t *d_base, const uint32_t *d_rows, uint4 **d_slot16);
hipError_t build_chain_rows(const uint8_t *d_blob, uint64_t slot_count, uint32_t max_ix, uint2 **d_info, uint64_t **d_base, uint32_t **d_rows,
                            uint64_t *total_rows);

// UFIndex::Validate (ufindex.cpp:611-658) over the resident table: out[9] = heads, positions, used slots, slots reached by the
// walks, bad hashes, bad positions, bad links, bad row lengths, first bad head slot (all ones: none); ms: the pass on the device
hipError_t validate_index(const DevIndex &X, uint64_t out[9], float *ms);

// sum of murmur64(word_i + (i + 1) * golden ratio) over the array's little-endian 64-bit words (chain_rows.hip); d_ptr 8-byte aligned
hipError_t checksum_device(const void *d_ptr, uint64_t nbytes, uint64_t *out);

hipError_t launch_viterbi_batch(const urmapx_params &P, const uint8_t *d_a, const uint32_t *d_aoffs,
                                const uint8_t *d_b, const uint32_t *d_boffs, const uint8_t *d_flags, uint32_t n,
                                float *d_scores, uint8_t *d_status, urmapx_path_op *d_ops, uint16_t *d_nops,
                                uint8_t *d_scratch, hipStream_t s);

}  // namespace urx
