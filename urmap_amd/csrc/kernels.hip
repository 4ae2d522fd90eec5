// kernels.hip -- gfx950 (CDNA4, wave64) device code of the urmap mapping path.
//
// Two kernels per single-end batch:
//   seed_probe_kernel  State1::SetSlotsVec (state1.cpp:396-438) + murmur64/WordToSlot (ufindex.h:50-65) +
//                      UFIndex::GetBlob (ufindex.h:184-187) for every k-mer of both strands.  One wavefront per
//                      read: the read's 2-bit letters become three 64-bit ballot planes per 64 bases, each lane
//                      cuts its own W-mer out of the planes, hashes it and gathers the 5-byte slot.  HBM-random-
//                      access bound (one 64 B sector per k-mer out of a table far larger than L2/MALL).
//   search_se_kernel   State1::Search_Lo (search1m6.cpp:35-277) incl. ExtendPen (extendpen.cpp:9-95), GetRow_Blob
//                      (ufindex.cpp:883-943), AddHitX/AddHSPX (state1.cpp:508-591), AlignHSP (alignhsp.cpp:60-172),
//                      Viterbi (viterbi.cpp:11-261), TraceBackBitMem (tracebackbitmem.cpp:8-75), CalcMAPQ6
//                      (search1m6.cpp:9-33) and SetMappedPos (state1.cpp:129-145).  One wavefront per read, reads
//                      handed out by a ticket counter to 16 persistent single-wave blocks per CU.  The
//                      order-dependent schedule is wave-uniform (scalar); the data-parallel parts use the 64
//                      lanes: one candidate window per lane in ExtendPen (mismatch bit vector + x-drop walk), one
//                      collision chain per lane in GetRow_Blob, one diagonal per lane in the banded DP (row sweep;
//                      the in-row insert dependency is a max-plus prefix scan over DPP), one hit per lane.
//                      Launched twice per batch: <NCH, false> over all reads, then <NCH, true> (lists continued in
//                      global scratch) over the few reads whose HSP / hit lists outgrew the first pass's.
//
// DP scores are kept in fp32 exactly as the reference does (small integers and a -9e9f "minus infinity" that
// absorbs small addends), so every tie and every trace bit is reproduced without re-deriving integer sentinels.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

#include "dev_common.h"
#include "probe_dev.h"
#include "viterbi_dev.h"

namespace urx {

// ------------------------------------------------------------------------------------------------
// kernel A: seed + probe
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void seed_probe_kernel(DevIndex X, const uint8_t *__restrict__ bases,
                                                         const uint64_t *__restrict__ offs, uint32_t n, ProbeOut out) {
	const int lane = threadIdx.x & 63;
	const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (r >= n) return;
	const uint64_t off = offs[r];
	const uint32_t QL = (uint32_t)(offs[r + 1] - off);
	const uint32_t W = X.W;
	if (QL < W || QL > 64u * NCH) return;
	const uint8_t *q = bases + off;
	probe_read<NCH>(X, q, QL, off, lane, out);
}


// ------------------------------------------------------------------------------------------------
// stage kernel for parity tests: a batch of independent Viterbi problems
// ------------------------------------------------------------------------------------------------
static constexpr int VB_MAXL = 448;  // narrow-path LDS capacity of the batch kernel
static constexpr int VB_WIDE_LA = 320, VB_WIDE_LB = 1800;

__global__ __launch_bounds__(64) void viterbi_batch_kernel(urmapx_params P, const uint8_t *__restrict__ a,
                                                           const uint32_t *__restrict__ aoffs,
                                                           const uint8_t *__restrict__ b,
                                                           const uint32_t *__restrict__ boffs,
                                                           const uint8_t *__restrict__ flags, uint32_t n,
                                                           float *scores, uint8_t *status_out, urmapx_path_op *ops_out,
                                                           uint16_t *nops_out, uint8_t *scratch, size_t scratch_stride) {
	__shared__ uint8_t sA[VB_MAXL];
	__shared__ uint32_t tb[(VB_MAXL / 8 + 2) * 64];
	__shared__ uint16_t rops[OPS_CAP];
	const int lane = threadIdx.x;
	const uint32_t k = blockIdx.x;
	if (k >= n) return;
	const int LA = (int)(aoffs[k + 1] - aoffs[k]), LB = (int)(boffs[k + 1] - boffs[k]);
	uint32_t status = 0;
	float score = 0.0f;
	RevOps R;
	R.ops = rops;
	R.begin();
	if (LA > VB_MAXL)
		status = URMAPX_ST_BAND_TOO_WIDE;
	else {
		for (int i = lane; i < LA; i += 64) sA[i] = a[aoffs[k] + i];
		URX_SYNC();
		WideScratch ws;
		ws.carve(scratch + (size_t)k * scratch_stride, VB_WIDE_LA, VB_WIDE_LB);
		score = viterbi_wave(VPar(P), sA, LA, b + boffs[k], LB, flags[k] & 1, (flags[k] >> 1) & 1, tb, VB_MAXL / 8 + 2, ws, R,
		                     status, lane);
	}
	int nout = R.n;
	if (nout > URMAPX_MAX_PATH_OPS) { status |= URMAPX_ST_PATH_OVERFLOW; nout = 0; }
	if (status) nout = 0;
	for (int t = lane; t < nout; t += 64) ops_out[(size_t)k * URMAPX_MAX_PATH_OPS + t] = rops[nout - 1 - t];  // forward order
	if (lane == 0) { scores[k] = score; status_out[k] = (uint8_t)status; nops_out[k] = (uint16_t)nout; }
}

// The same problems two per wavefront through VFlank / viterbi_pair_rows (viterbi_dev.h): the packed-int16 interior
// blocks that dp_kernel uses, behind the same parity test (URMAPX_VITERBI_PAIR=1 selects this kernel).
__global__ __launch_bounds__(64) void viterbi_batch_pair_kernel(urmapx_params P, const uint8_t *__restrict__ a,
                                                                const uint32_t *__restrict__ aoffs, const uint8_t *__restrict__ b,
                                                                const uint32_t *__restrict__ boffs, const uint8_t *__restrict__ flags,
                                                                uint32_t n, float *scores, uint8_t *status_out, urmapx_path_op *ops_out,
                                                                uint16_t *nops_out, uint8_t *scratch, size_t scratch_stride) {
	constexpr int TBR = VB_MAXL / 8 + 2;
	__shared__ uint8_t sA[2][VB_MAXL];
	__shared__ uint8_t sB[2][VB_MAXL + 96];
	__shared__ uint32_t tb[2][TBR * 64];
	__shared__ uint16_t rops[2][OPS_CAP];
	const int lane = threadIdx.x;
	const VPar VP(P);
	VFlank F[2];
	bool narrow[2] = {false, false};
	int LA[2] = {0, 0}, LB[2] = {0, 0};
	uint32_t kk[2];
	for (int h = 0; h < 2; ++h) {
		kk[h] = 2 * blockIdx.x + h;
		if (kk[h] >= n) continue;
		const uint32_t k = kk[h];
		LA[h] = (int)(aoffs[k + 1] - aoffs[k]); LB[h] = (int)(boffs[k + 1] - boffs[k]);
		if (LA[h] <= VB_MAXL)
			for (int i = lane; i < LA[h]; i += 64) sA[h][i] = a[aoffs[k] + i];
		if (LB[h] <= VB_MAXL + 64)
			for (int i = lane; i < LB[h]; i += 64) sB[h][16 + i] = b[boffs[k] + i];
	}
	URX_SYNC();
	for (int h = 0; h < 2; ++h)
		if (kk[h] < n && LA[h] <= VB_MAXL && LB[h] <= VB_MAXL + 64)
			narrow[h] = F[h].setup(VP, sA[h], LA[h], sB[h] + 16, LB[h], flags[kk[h]] & 1, (flags[kk[h]] >> 1) & 1, tb[h], TBR, -3.0e38f, false);
	if (!narrow[0]) F[0].active = false;
	if (!narrow[1]) F[1].active = false;
	viterbi_pair_rows(F[0], F[1]);
	for (int h = 0; h < 2; ++h) {
		if (kk[h] >= n) continue;
		const uint32_t k = kk[h];
		uint32_t status = 0;
		float score = 0.0f;
		RevOps R;
		R.ops = rops[h];
		R.begin();
		if (narrow[h]) score = F[h].finish(R, status);
		else if (LA[h] > VB_MAXL) status = URMAPX_ST_BAND_TOO_WIDE;
		else {  // degenerate or wide: the single-problem path
			WideScratch ws;
			ws.carve(scratch + (size_t)k * scratch_stride, VB_WIDE_LA, VB_WIDE_LB);
			score = viterbi_wave(VP, sA[h], LA[h], b + boffs[k], LB[h], flags[k] & 1, (flags[k] >> 1) & 1, tb[h], TBR, ws, R, status, lane);
		}
		int nout = R.n;
		if (nout > URMAPX_MAX_PATH_OPS) { status |= URMAPX_ST_PATH_OVERFLOW; nout = 0; }
		if (status) nout = 0;
		URX_SYNC();
		for (int t = lane; t < nout; t += 64) ops_out[(size_t)k * URMAPX_MAX_PATH_OPS + t] = rops[h][nout - 1 - t];
		if (lane == 0) { scores[k] = score; status_out[k] = (uint8_t)status; nops_out[k] = (uint16_t)nout; }
		URX_SYNC();
	}
}

// ------------------------------------------------------------------------------------------------
// kernel B: per-read search.  One wavefront per read.
//
// The reference's schedule (search1m6.cpp:35-277) is a chain of ~600 dependent memory accesses per read
// (one target window per ExtendPen, one slot per chain hop).  The accesses themselves do not depend on the
// search state -- only WHAT IS DONE with each window does -- so every phase is split:
//   scan     the phase's candidates are enumerated in the reference's order; those on the diagonal block of a hit
//            already found (ExtendPen would return at once) are dropped, the rest compacted into an LDS queue
//   gather   (order independent, 64-wide): each lane takes one queued candidate and computes that candidate's whole
//            mismatch bit vector against its reference window (lane_mismatch_mask) and the outcome of ExtendPen's
//            two x-drop walks on it (xdrop_walk_lane); chain walks put one collision chain on each lane (walk_all)
//   consume  (order dependent, wave-uniform): only the candidates whose outcome can still change the search state
//            are visited, in order: AddHitX / AddHSPX, penalty cap, early exits -- exactly as the reference
// ------------------------------------------------------------------------------------------------
// HSP record word: startq | len << 10 | score << 20 | aligned << 30 | plus << 31 (each field <= 1023)
static constexpr uint32_t PK_MASK = 1023u, PK_LEN_SH = 10, PK_SCORE_SH = 20, PK_ALIGNED = 1u << 30, PK_PLUS_SH = 31;
// Round 6: the windows of the next candidate batch (and a read's long rows) are touched into L2 ahead of their gathers (dev_common.h:
// glds_touch).  The sink of those loads is 256 bytes of LDS; the HSP list in LDS gives them up (192 entries instead of 256: a read
// with more keeps the rest in its block's global scratch, as before) so that the block stays under the 10 240 bytes that let 16 of
// them share a CU.  URX_PREFETCH=0: the round-5 kernel (A/B builds).
#ifndef URX_PREFETCH
#define URX_PREFETCH 0  // bit 0: the next batch's windows, bit 1: a read's long rows
#endif
// (The scan of the next batch issued beside the current batch's window gather -- URX_SCAN_AHEAD, commit f9ec288 -- was bit-identical and 5-17 % slower:
// profiles/r6/ab_scan_ahead.txt.)
#ifndef URX_HSP_CAP
#define URX_HSP_CAP (URX_PREFETCH ? 192 : 256)  // build-time experiment (profiles/r5/ab_waves5.txt): 64 frees 1.5 KB of LDS per block
#endif
static constexpr int HSP_CAP = URX_HSP_CAP;        // HSPs of a read held in LDS
static constexpr int SEARCH_OVF_BLOCKS = 2048;  // grid of the second pass (reads whose HSP list outgrew LDS): these are the
                                                // costliest reads of a batch (hundreds of AlignHSP calls each), so they get most of the chip
static constexpr int HSP_TOTAL_CAP = 8192;  // beyond that: in the block's global scratch (the reference's list is unbounded;
                                            // 8192 > 2 strands x 127 k-mers x MaxIx 32 candidate diagonals of a 150 bp read)
#ifndef URX_TICKET_CHUNK
#define URX_TICKET_CHUNK 4
#endif
static constexpr int TICKET_CHUNK = URX_TICKET_CHUNK;
static constexpr int ROW_CAP = 32;  // UFIndex m_MaxIx of every index this build accepts

// per-block global scratch of the search kernel: chain rows, the wide-band DP's rows and trace, the banded DP's trace cells
__host__ __device__ inline size_t search_tb_offset(int nch) {
	const int qmax = 64 * nch;
	size_t b = (size_t)2 * nch * ROW_CAP * 64 * 4 + WideScratch::bytes(qmax, qmax + 64);
	return (b + 255) & ~(size_t)255;
}
// behind the trace cells: hits 65..128 of a first-pass read (SE_HIT_TAIL words)
static constexpr int SE_HIT_TAIL = 64;
__host__ __device__ inline size_t search_tail_offset(int nch) {
	return (search_tb_offset(nch) + (size_t)((64 * nch - 24) / 8 + 2) * 64 * 4 + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t search_scratch_bytes(int nch) { return search_tail_offset(nch) + (size_t)SE_HIT_TAIL * 4; }

// AddHSPX over the part of a read's HSP list that lives in global scratch (reads in high-copy repeats only).  Kept out
// of line so that its registers do not count against the search loop's.  0: same diagonal found (entry updated if the
// score is higher), 1: appended at index n, 2: list full.
__device__ __noinline__ int hsp_overflow_add(uint2 *ovf, int n, int cap, uint32_t diag, uint32_t startdb, uint32_t npk, int score) {
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		bool eq = false;
		uint2 e = make_uint2(0u, 0u);
		if (i < n) { e = ovf[i]; eq = (e.x - (e.y & PK_MASK)) == diag; }
		const uint64_t m = __ballot(eq);
		if (m) {
			const int l = __builtin_ctzll(m);
			const int old = (int)((rdlane(e.y, l) >> PK_SCORE_SH) & PK_MASK);
			if (score > old && lane == 0) ovf[base + l] = make_uint2(startdb, npk);
			URX_SYNC();
			return 0;
		}
	}
	if (n >= cap) return 2;
	if (lane == 0) ovf[n] = make_uint2(startdb, npk);
	return 1;
}

#ifndef URX_SE_HITW1
#define URX_SE_HITW1 1
#endif
// hit-list words (64 hits each) of the first pass.  5 of 1 M reads of a repeat-rich genome end with more than 64 hits and
// are mapped again by a launch of their own (0.9 ms); with two words they stay in the first pass, which then runs 1.6 ms
// longer (such a read is its tail): one word it stays.  The pair kernel keeps two (kernels_pe.hip: -1.5 ms).
static constexpr int SE_HITW1 = URX_SE_HITW1;
// LAYOUT 1 (round 5): the slot entries in LDS come out of DevIndex::slot16 -- pr_lo = the slot's position, pr_hi = tally | row length
// << 8, pr_sl = the row's second position (rows of two) or its index in DevIndex::rows -- instead of the two dwords around the
// 5-byte slot and the slot number's low half
// KCH (round 6): the chunks of 64 k-MER START positions the instance holds per strand -- NCH by default (every base but the last W - 1 starts a k-mer);
// an instance for reads whose QL - W + 1 k-mers fit fewer chunks (150 bases, W = 24: 127 starts = two chunks of a three-chunk read) keeps slot entries,
// prefix array and chain groups for those only: a third less LDS in them, a third fewer group steps in every unrolled loop over them
template <int NCH, bool OVF, int LAYOUT = 0, int KCH = NCH>
struct SearchWave {
	static_assert(KCH >= 1 && KCH <= NCH, "k-mer chunks");
	static constexpr int QMAX = 64 * NCH;
	// rows of the narrow DP's trace buffer: a flank is at most QMAX - W long (the HSP holds the seed), longer problems
	// take the wide path
	static constexpr int TB_ROWS8 = (QMAX - 24) / 8 + 2;
	static constexpr int WIDE_LB = QMAX + 64;
	static constexpr int NSEG = 2 * KCH;  // candidate row segments: [strand][chunk] or [chunk]

	const DevIndex &X;
	const urmapx_params &P;
	int lane;  // not const: the search kernel refreshes it at the top of its hot loops (fresh_lane)
	const uint8_t *__restrict__ gseq;   // = X.seq / X.blob, passed as plain kernel arguments so that the
	const uint8_t *__restrict__ gblob;  // compiler emits global_load (not flat) and can keep many loads in flight
	int QL, W, nwords;
	// LDS of this wavefront
	uint8_t *sQ[2];  // [0] = plus (read as given), [1] = minus (reverse complement); 16-byte aligned
	uint8_t *sT;     // target window of the current flank
	uint32_t *tb;
	uint16_t *ropsL, *ropsR, *cand, *top;
	uint16_t *pre;   // exclusive prefix of candidate counts, NSEG*64+1 entries
	uint32_t *hsp_db;
	uint32_t *hsp_pk;  // PK_* packing: startq | len | score | aligned | plus
	uint2 *hsp_ovf;    // global scratch of this block: HSPs hsp_lds.. as {db, pk} (reads in high-copy repeats only)
	int hsp_lds;       // HSPs kept in LDS: HSP_CAP (a test aid lowers it to exercise the overflow path)
	int hit_cap;       // first pass: 64 hits (the test aid lowers it along with hsp_lds)
	// global scratch of this block
	uint32_t *rowstore;  // [strand][chunk][k][lane]
	WideScratch ws;
	// hits: entry k lives on lane k & 63 of word k >> 6.  One word (64 hits) in the first-pass kernel; the second pass
	// (reads that outgrew a list) has HITW words = 512 hits.
	static constexpr int HITW = OVF ? 8 : SE_HITW1;
	uint32_t hit_db[HITW];
	// First pass only: hits 65..128 live in global memory (this block's scratch while the read is searched, the read's
	// parked state afterwards) and are looked at only by a read that has them -- 5 reads in a million on the hg38-scale
	// genome, which used to cost every batch a second pass of ~1 ms.  The register word stays one.
	static constexpr int TAIL = OVF ? 0 : SE_HIT_TAIL;
	uint32_t *hit_tail;
	__device__ __forceinline__ bool has_tail() const { return !OVF && hitCount > 64 * SE_HITW1; }
	// hits per word: 64 (2^6).  The test aid that lowers the first pass's caps also lowers this to 16 in the second pass,
	// so that a fixture with a few dozen hits per read runs through several words.
	int hit_wsh;
	__device__ __forceinline__ int wsh() const { return OVF ? hit_wsh : 6; }
	__device__ __forceinline__ int wl() const { return 1 << wsh(); }
	int hitCount, hspCount;
	int maxPen, best, second, bestHSP;
	bool haveTop; uint32_t top_db; bool top_plus; int top_nops;
	uint32_t status;

	__device__ SearchWave(const DevIndex &X_, const urmapx_params &P_, int lane_) : X(X_), P(P_), lane(lane_) {}

	__device__ __forceinline__ bool overlaps_hit(uint32_t db) const {
		bool eq = false;
#pragma unroll
		for (int w = 0; w < HITW; ++w) eq |= lane < wl() && (w << wsh()) + lane < hitCount && (hit_db[w] >> 6) == (db >> 6);
		if (has_tail()) eq |= lane < hitCount - 64 * SE_HITW1 && (hit_tail[lane] >> 6) == (db >> 6);
		return __ballot(eq) != 0;
	}

	// per-lane form: does this lane's diagonal start fall in the 64-base block of any hit found so far
	__device__ __forceinline__ bool overlaps_any_hit(uint32_t db) const {
		bool ov = false;
#pragma unroll
		for (int w = 0; w < HITW; ++w)
			for (int k = 0; k < wl() && (w << wsh()) + k < hitCount; ++k) ov |= (rdlane(hit_db[w], k) >> 6) == (db >> 6);
		if (has_tail()) {
			const int nt = hitCount - 64 * SE_HITW1;
			const uint32_t t = lane < nt ? hit_tail[lane] : 0u;
			for (int k = 0; k < nt; ++k) ov |= (rdlane(t, k) >> 6) == (db >> 6);
		}
		return ov;
	}

	// state1.cpp:508-551.  path (if any) is in `cand` with cand_nops runs.
	__device__ __forceinline__ void add_hit(uint32_t db, bool plus, int score, int cand_nops) {
		if (score < 10) return;
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		if (overlaps_hit(db)) return;
		int mp = (QL - score) - 2 * P.mismatch_score;
		if (mp < maxPen) maxPen = mp;
		bool newTop = false;
		if (score > best) { second = best; best = score; newTop = true; }
		else if (score == best) second = score;
		else {
			if (score < best - SECONDARY_HIT_MAX_DELTA) return;
			if (score > second) second = score;
		}
		// the first pass's capacity: its register word(s) + the tail (a test aid lowers hit_cap below a word: no tail then)
		if (hitCount >= (OVF ? HITW << wsh() : (hit_cap == 64 * SE_HITW1 ? hit_cap + TAIL : hit_cap))) { status |= URMAPX_ST_HIT_OVERFLOW; return; }
		if (!OVF && hitCount >= 64 * SE_HITW1) {
			if (lane == 0) hit_tail[hitCount - 64 * SE_HITW1] = db;
			URX_SYNC();  // the other lanes read it (overlaps_hit / overlaps_any_hit)
		} else {
#pragma unroll
			for (int w = 0; w < HITW; ++w)
				if (lane < wl() && (w << wsh()) + lane == hitCount) hit_db[w] = db;
		}
		++hitCount;
		if (newTop) {
			haveTop = true; top_db = db; top_plus = plus; top_nops = cand_nops;
			if (cand_nops) {
				for (int t = lane; t < cand_nops; t += 64) top[t] = cand[t];
				URX_SYNC();
			}
		}
	}

	// state1.cpp:553-591
	__device__ __forceinline__ void add_hsp(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		if (score < best - 4) return;
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		const uint32_t diag = startdb - startq;
		const uint32_t npk = startq | (len << PK_LEN_SH) | ((uint32_t)score << PK_SCORE_SH) | (plus ? 1u << PK_PLUS_SH : 0u);
		const int nlds = hspCount < hsp_lds ? hspCount : hsp_lds;
		for (int base = 0; base < nlds; base += 64) {  // the HSPs in LDS (all of them, except for reads in high-copy repeats)
			const int i = base + lane;
			bool eq = false;
			if (i < nlds) eq = (hsp_db[i] - (hsp_pk[i] & PK_MASK)) == diag;
			uint64_t m = __ballot(eq);
			if (m) {
				const int k = base + __builtin_ctzll(m);
				const int old = (int)((hsp_pk[k] >> PK_SCORE_SH) & PK_MASK);
				if (score > old && lane == 0) { hsp_db[k] = startdb; hsp_pk[k] = npk; }
				URX_SYNC();
				return;
			}
		}
		if (hspCount >= hsp_lds) {
			// The LDS share of the list is full (a read in a repeat family: 0.2 % of 150 bp reads and 2-4 % of 250 bp reads
			// on a genome with hg38's repeat content collect more than 256 HSPs).  The list continues in this block's
			// global scratch (the reference's list is unbounded, state1.cpp:193-228), out of line.
			const int rc = hsp_overflow_add(hsp_ovf, hspCount - hsp_lds, HSP_TOTAL_CAP - HSP_CAP, diag, startdb, npk, score);
			if (rc == 0) return;
			if (rc == 2) { status |= URMAPX_ST_HSP_OVERFLOW; return; }
		} else if (lane == 0) {
			hsp_db[hspCount] = startdb; hsp_pk[hspCount] = npk;
		}
		URX_SYNC();
		++hspCount;
		if (score > bestHSP) bestHSP = score;
	}

	// load a target window into LDS; returns true if it contains a '-' pad byte
	__device__ __forceinline__ bool load_window(uint32_t tlo, int tl) {
		bool gap = false;
		for (int i = lane; i < tl; i += 64) {
			uint8_t c = gseq[tlo + i];
			sT[i] = c;
			gap |= (c == '-');
		}
		URX_SYNC();
		return __ballot(gap) != 0;
	}

	// alignhsp.cpp:60-172
	__device__ void align_hsp(int k) {
		uint32_t startdb, pk;
		if (k < hsp_lds) { startdb = hsp_db[k]; pk = hsp_pk[k]; }
		else { const uint2 e = hsp_ovf[k - hsp_lds]; startdb = e.x; pk = e.y; }
		if (pk & PK_ALIGNED) return;  // m_Aligned
		URX_SYNC();
		if (lane == 0) {
			if (k < hsp_lds) hsp_pk[k] = pk | PK_ALIGNED;
			else hsp_ovf[k - hsp_lds].y = pk | PK_ALIGNED;
		}
		const int startq = (int)(pk & PK_MASK), len = (int)((pk >> PK_LEN_SH) & PK_MASK);
		const int hscore = (int)((pk >> PK_SCORE_SH) & PK_MASK);
		const bool plus = (pk >> PK_PLUS_SH) & 1u;
		URX_SYNC();
		int totalPen = len - hscore;
		int totalScore = hscore;
		if (totalPen > maxPen) return;
		const int BR = 2 * (int)P.band_radius;
		const uint32_t TL = X.seqDataSize;
		uint32_t combinedTLo = startdb;
		const uint8_t *Q = plus ? sQ[0] : sQ[1];
		RevOps RL, RR;
		RL.ops = ropsL; RR.ops = ropsR;
		RL.begin(); RR.begin();
		int rtrim = 0;
		const VPar VP(P);
		const WideScratch wsv = ws;
		uint32_t vst = 0;

		// the two flanks through ONE copy of the banded DP (the kernel's code has to stay near the instruction cache's size)
		const int rightQLo = startq + len;
#pragma unroll 1
		for (int side = 0; side < 2; ++side) {
			const bool left = side == 0;
			int fql;
			uint32_t tlo, tl;
			const uint8_t *fq;
			if (left) {
				if (startq <= 0) continue;
				if (startdb < (uint32_t)startq) return;
				fql = startq;
				const uint32_t leftTHi = startdb - 1;
				tl = (uint32_t)(fql + BR);
				if (tl >= leftTHi) return;
				tlo = leftTHi - tl + 1;
				fq = Q;
			} else {
				if (rightQLo >= QL) continue;
				fql = QL - rightQLo;
				tlo = startdb + (uint32_t)len;
				uint32_t thi = tlo + (uint32_t)fql + (uint32_t)BR;
				if (thi >= TL) thi = TL - 1;
				tl = thi - tlo + 1;
				fq = Q + rightQLo;
			}
			if (load_window(tlo, (int)tl)) return;
			// the DP stops as soon as the flank cannot stay within what the penalty cap leaves (viterbi_dev.h); the test
			// that would discard it follows right below, so the outcome is the same
			const int allGap = P.gap_open_score + (fql - 1) * P.gap_ext_score;
			const int need = fql - (maxPen - totalPen);
			bool aborted = false;
			RevOps R;
			R.ops = left ? ropsL : ropsR;
			int score = (int)viterbi_wave<true>(VP, fq, fql, sT, (int)tl, left, !left, tb, TB_ROWS8, wsv, R, vst, lane, (float)need,
			                                    allGap < need ? &aborted : nullptr);
			if (aborted) return;
			status |= vst;
			if (left) {
				RL.n = R.n;
				// TrimLeftIs (pathinfo.cpp:153-171): the leading I run is the last run in traceback order
				int nTrimI = 0;
				if (RL.n > 0) {
					uint32_t lastop = ropsL[RL.n - 1];
					if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --RL.n; }
				}
				combinedTLo = tlo + (uint32_t)nTrimI;
			} else {
				RR.n = R.n;
				// TrimRightIs (pathinfo.cpp:173-190): trailing I run = first run in traceback order, never the whole path
				if (RR.n > 1 && (ropsR[0] & 3u) == OP_I) rtrim = 1;
			}
			if (allGap > score) score = allGap;
			totalScore += score;
			totalPen += fql - score;
			if (totalPen > maxPen) return;
		}
		if (vst & (URMAPX_ST_BAND_TOO_WIDE | URMAPX_ST_PATH_OVERFLOW)) return;
		// path = Left || M x len || Right, run-length merged, into cand (uniform; lane 0 stores)
		int nc = 0, cop = -1, clen = 0;
		bool ovf = false;
		auto put = [&](int op, int l) {
			if (l <= 0) return;
			if (op == cop) { clen += l; return; }
			if (clen) { if (nc < URMAPX_MAX_PATH_OPS) { if (lane == 0) cand[nc] = (uint16_t)((clen << 2) | cop); ++nc; } else ovf = true; }
			cop = op; clen = l;
		};
		for (int t = RL.n - 1; t >= 0; --t) { uint32_t o = ropsL[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(OP_M, len);
		for (int t = RR.n - 1; t >= rtrim; --t) { uint32_t o = ropsR[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(-2, 1);  // flush
		if (ovf) { status |= URMAPX_ST_PATH_OVERFLOW; return; }
		URX_SYNC();
		add_hit(combinedTLo, plus, totalScore, nc);
	}

	// ---- phase 6 as separate launches (kernels.h: DpJob) ----
	static constexpr int STATE_WORDS = HITW * 64 + TAIL + 16 + URMAPX_MAX_PATH_OPS / 2;  // hit words, hit tail, scalars, the top hit's path

	__device__ __forceinline__ bool hsp_get(int k, uint32_t &startdb, uint32_t &pk) const {
		startdb = 0; pk = 0;
		if (k >= hspCount) return false;
		if (k < hsp_lds) { startdb = hsp_db[k]; pk = hsp_pk[k]; }
		else { const uint2 e = hsp_ovf[k - hsp_lds]; startdb = e.x; pk = e.y; }
		return true;
	}
	// AlignHSP's entry tests (alignhsp.cpp:62-70): not aligned yet, and HSP.Length - HSP.Score within the penalty cap
	__device__ __forceinline__ bool hsp_wants_dp(int k, uint32_t &startdb, uint32_t &pk) const {
		if (!hsp_get(k, startdb, pk)) return false;
		if (pk & PK_ALIGNED) return false;
		const int len = (int)((pk >> PK_LEN_SH) & PK_MASK), hscore = (int)((pk >> PK_SCORE_SH) & PK_MASK);
		return len - hscore <= maxPen;
	}

	// Round 5: a read parked at PHASE 3 takes more with it than one parked at phase 6 -- it has phases 4-6 still to run: behind the
	// STATE_WORDS of park_state lie its HSP list (P3_HSP entries at most: the HSPs of phases 1-2 come from unique seeds, a handful per
	// read) and its slot entries (the pr_* arrays: what the probe gathered for it), so that the second launch neither hashes nor
	// probes again.
	static constexpr int P3_HSP = 128;
	static constexpr int P3_WORDS = STATE_WORDS + 2 * P3_HSP + 3 * NSEG * 64 + 2 * NSEG;

	// One DpJob per HSP that phase 6 (P3: phase 3) would align, the read's state parked for finalize_se_kernel (P3: for the
	// search kernel's second launch).  1: parked; 0: nothing to align; -1: no room left in the job array / the parking lot (P3:
	// or an HSP list beyond what a parked read carries) -- the caller then runs the phase itself, or hands the read to the second pass.
	template <bool P3 = false>
	__device__ int park_for_dp(const DpWork &dp, uint32_t r, int phase) {
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		int njobs = 0;
		bool clipped = false;
		for (int base = 0; base < hspCount; base += 64) {
			uint32_t sdb, pk;
			const bool want = hsp_wants_dp(base + lane, sdb, pk);
			njobs += __builtin_popcountll(__ballot(want));
			// a right flank window that the end of the sequence store cuts short (alignhsp.cpp:143-145) makes a band wider than
			// a wavefront: dp_kernel runs its DPs two to a wavefront in packed int16 and has no wide path -- such a read (it lies
			// within a read length of the end of the last sequence) keeps its phase 6 in this kernel
			clipped |= want && (uint64_t)(sdb - (pk & PK_MASK)) + (uint64_t)QL + 2ull * P.band_radius >= (uint64_t)X.seqDataSize;
		}
#ifdef URX_DP_PAIR
		if (__ballot(clipped) != 0) return -1;
#else
		(void)clipped;
#endif
		if (njobs == 0) return 0;
		if (P3 && (hspCount > hsp_lds || hspCount > P3_HSP)) return -1;
		uint32_t jb = 0, slot = 0;
		if (lane == 0) jb = atomicAdd(dp.counters, (uint32_t)njobs);
		jb = uni(jb);
		bool room = jb < dp.jobs_cap && (uint32_t)njobs <= dp.jobs_cap - jb;
		if (room) {
			if (lane == 0) slot = atomicAdd(dp.counters + 1, 1u);
			slot = uni(slot);
			room = slot < dp.fin_cap;
		}
		if (!room) {  // the slots taken stay unused
			for (uint32_t i = jb + lane; i < dp.jobs_cap && i - jb < (uint32_t)njobs; i += 64) { dp.jobs[i].read = 0xFFFFFFFFu; dp.kidx[i] = 0xFFFFu; }
			return -1;
		}
		int w = 0;
		for (int base = 0; base < hspCount; base += 64) {
			uint32_t sdb, pk;
			const bool ok = hsp_wants_dp(base + lane, sdb, pk);
			const uint64_t m = __ballot(ok);
			if (ok) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				DpJob J;
				J.read = r; J.startdb = sdb; J.pk = pk; J.maxpen = maxPen;
				J.combined_tlo = 0; J.left_score = 0; J.right_score = 0; J.nops = 0; J.flags = 0; J.vst_l = 0; J.vst_r = 0;
				J.k = (uint16_t)((uint32_t)w + rank); J.pad[0] = J.pad[1] = 0;
				dp.jobs[jb + (uint32_t)w + rank] = J;
				dp.kidx[jb + (uint32_t)w + rank] = J.k;
			}
			w += __builtin_popcountll(m);
		}
		if (njobs > 0xFFFF) status |= URMAPX_ST_HSP_OVERFLOW;  // cannot happen: the HSP list holds at most 8192
		if (lane == 0) *reinterpret_cast<uint4 *>(dp.fin_list + 4 * (size_t)slot) = make_uint4(r, jb, (uint32_t)njobs, (uint32_t)QL);
		uint32_t *const st = dp.state + (size_t)slot * (P3 ? P3_WORDS : STATE_WORDS);
		park_state(st, phase);
		if constexpr (P3) {
			// AlignHSP marks an HSP aligned before it tests it (alignhsp.cpp:62-70): every HSP of the list is, after phase 3
			uint32_t *const hx = st + STATE_WORDS;
			for (int i = lane; i < hspCount; i += 64) { hx[i] = hsp_db[i]; hx[P3_HSP + i] = hsp_pk[i] | PK_ALIGNED; }
			uint32_t *const px = hx + 2 * P3_HSP;
#pragma unroll
			for (int g = 0; g < NSEG; ++g)
				if (64 * (g % KCH) < nwords) {
					px[g * 64 + lane] = pr_lo[g * 64 + lane];
					px[(NSEG + g) * 64 + lane] = pr_hi[g * 64 + lane];
					px[(2 * NSEG + g) * 64 + lane] = pr_sl[g * 64 + lane];
				}
			if (lane < NSEG) reinterpret_cast<uint64_t *>(px + 3 * NSEG * 64)[lane] = pr_hb[lane];
		}
		return 1;
	}

	// phase 3 found nothing to align: its HSPs are marked all the same (alignhsp.cpp:62-70 sets m_Aligned before the penalty test)
	__device__ __forceinline__ void mark_all_aligned() {
		const int nl = hspCount < hsp_lds ? hspCount : hsp_lds;
		for (int i = lane; i < nl; i += 64) hsp_pk[i] |= PK_ALIGNED;
		for (int i = hsp_lds + lane; i < hspCount; i += 64) hsp_ovf[i - hsp_lds].y |= PK_ALIGNED;
		URX_SYNC();
	}

	// A read parked at phase 3 comes back (search kernel, PART 2): slot entries by LDS-DMA (contiguous rows of 64 words), the
	// state of restore_state, the HSP list.  Returns the phase it was parked in (3).
	__device__ int resume3(uint32_t *st) {
		const int lane = fresh_lane(this->lane);
		const uint32_t *const hx = st + STATE_WORDS;
		const uint32_t *const px = hx + 2 * P3_HSP;
		wait_lgkm0();  // every earlier LDS read of the pr_* arrays has returned
#pragma unroll
		for (int g = 0; g < NSEG; ++g)
			if (64 * (g % KCH) < nwords) {
				glds_dword(px + g * 64 + lane, lds_addr(pr_lo + g * 64));
				glds_dword(px + (NSEG + g) * 64 + lane, lds_addr(pr_hi + g * 64));
				glds_dword(px + (2 * NSEG + g) * 64 + lane, lds_addr(pr_sl + g * 64));
			}
		uint64_t hb = 0;
		if (lane < NSEG) hb = reinterpret_cast<const uint64_t *>(px + 3 * NSEG * 64)[lane];
		const int phase = restore_state(st);
		for (int i = lane; i < hspCount; i += 64) { hsp_db[i] = hx[i]; hsp_pk[i] = hx[P3_HSP + i]; }
		if (lane < NSEG) pr_hb[lane] = hb;
		wait_vm0();
		URX_SYNC();
		return phase;
	}

	__device__ void park_state(uint32_t *st, int phase) {
#pragma unroll
		for (int wd = 0; wd < HITW; ++wd) st[wd * 64 + lane] = hit_db[wd];
		if (has_tail() && hit_tail != st + HITW * 64 && lane < hitCount - 64 * SE_HITW1) st[HITW * 64 + lane] = hit_tail[lane];
		uint32_t *sc = st + HITW * 64 + TAIL;
		if (lane == 0) {
			sc[0] = (uint32_t)hitCount; sc[1] = (uint32_t)maxPen; sc[2] = (uint32_t)best; sc[3] = (uint32_t)second;
			sc[4] = top_db; sc[5] = (haveTop ? 1u : 0u) | (top_plus ? 2u : 0u) | ((uint32_t)phase << 8);
			sc[6] = (uint32_t)top_nops; sc[7] = status; sc[8] = (uint32_t)hspCount;
		}
		uint16_t *tops = reinterpret_cast<uint16_t *>(sc + 16);
		for (int t = lane; t < top_nops; t += 64) tops[t] = top[t];
	}

	// back from the parking lot (finalize_se_kernel); returns the phase the read was in
	__device__ int restore_state(uint32_t *st) {
#pragma unroll
		for (int wd = 0; wd < HITW; ++wd) hit_db[wd] = st[wd * 64 + lane];
		hit_tail = st + HITW * 64;  // in place: the replay appends to it there
		const uint32_t *sc = st + HITW * 64 + TAIL;
		// the top hit's path comes in with the same round of loads as the scalars (its length is one of them: waiting for it
		// first made the path a round trip of its own; finalize_se_kernel is a chain of such trips and little else)
		static_assert(URMAPX_MAX_PATH_OPS <= 128, "two path ops per lane");
		const uint16_t *tops = reinterpret_cast<const uint16_t *>(sc + 16);
		const uint16_t t0 = tops[lane];
		const uint16_t t1 = lane + 64 < URMAPX_MAX_PATH_OPS ? tops[lane + 64] : (uint16_t)0;
		hitCount = (int)uni(sc[0]); maxPen = (int)uni(sc[1]); best = (int)uni(sc[2]); second = (int)uni(sc[3]);
		top_db = uni(sc[4]);
		const uint32_t f = uni(sc[5]);
		haveTop = (f & 1u) != 0; top_plus = (f & 2u) != 0;
		top_nops = (int)uni(sc[6]); status = uni(sc[7]); hspCount = (int)uni(sc[8]); bestHSP = 0;
		top[lane] = t0;
		if (lane + 64 < URMAPX_MAX_PATH_OPS) top[lane + 64] = t1;
		URX_SYNC();
		return (int)(f >> 8);
	}

	// AlignHSP's bookkeeping (alignhsp.cpp:60-172) over a job whose two flank DPs dp_kernel has run: the same tests in
	// the same order against the CURRENT penalty cap, then AddHitX.  The DP outcome itself never depended on the state.
	__device__ bool consume_job(const DpJob &J, const uint16_t *jops) {
		const uint32_t pk = J.pk;
		const int startq = (int)(pk & PK_MASK), len = (int)((pk >> PK_LEN_SH) & PK_MASK);
		const int hscore = (int)((pk >> PK_SCORE_SH) & PK_MASK);
		const bool plus = (pk >> PK_PLUS_SH) & 1u;
		int totalPen = len - hscore;
		int totalScore = hscore;
		if (totalPen > maxPen) return false;
		if (J.flags & DPJ_GATED) { status |= URMAPX_ST_BAND_TOO_WIDE; return false; }  // cannot happen: gated under a cap >= this one
		uint32_t vst = 0;
		if (startq > 0) {
			if (J.flags & DPJ_LEFT_FAIL) return true;
			vst |= J.vst_l;
			status |= J.vst_l;
			totalScore += J.left_score;
			totalPen += startq - J.left_score;
			if (totalPen > maxPen) return true;
		}
		const int rightQLo = startq + len;
		if (rightQLo < QL) {
			if (J.flags & (DPJ_RIGHT_FAIL | DPJ_RIGHT_SKIPPED)) return true;
			// the right flank was run unless the penalty after the left one already exceeded the cap the job was made
			// under -- and the cap has not risen since
			vst |= J.vst_r;
			status |= J.vst_r;
			totalScore += J.right_score;
			totalPen += (QL - rightQLo) - J.right_score;
			if (totalPen > maxPen) return true;
		}
		if (vst & (URMAPX_ST_BAND_TOO_WIDE | URMAPX_ST_PATH_OVERFLOW)) return true;
		if (J.flags & DPJ_PATH_LONG) { status |= URMAPX_ST_PATH_OVERFLOW; return true; }
		const int nc = (int)J.nops;
		URX_SYNC();
		for (int t = lane; t < nc; t += 64) cand[t] = jops[t];
		URX_SYNC();
		add_hit(J.combined_tlo, plus, totalScore, nc);
		return true;
	}

	// m_Mapq, SetMappedPos (state1.cpp:129-145) with PosToCoordL (ufindex.cpp:729-755), the top hit's path
	__device__ void fill_result(urmapx_result &res, int phase, urmapx_path_op *__restrict__ path_ops, uint32_t *path_used) {
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		if (fill_result_core(res, phase)) {
			uint32_t po = 0;
			if (lane == 0) po = atomicAdd(path_used, (uint32_t)top_nops);
			po = uni(po);
			for (int t = lane; t < top_nops; t += 64) path_ops[po + t] = top[t];
			res.path_off = po; res.path_nops = (uint16_t)top_nops;
		}
	}
	// everything of the result but the place of the top hit's path; true: there is a path (top[0 .. top_nops)) to be placed
	__device__ bool fill_result_core(urmapx_result &res, int phase) {
		res.mapq = (uint8_t)calc_mapq();
		res.score = (int16_t)best; res.second = (int16_t)second;
		res.hit_count = (uint16_t)hitCount; res.exit_phase = (uint8_t)phase; res.status = (uint8_t)status;
		if (haveTop) {
			uint32_t lo = 0, hi = X.seqCount - 1;
			uint32_t found = 0xFFFFFFFFu, coord = 0xFFFFFFFFu, tl = 0;
			if (URX_SEQ_LANES && X.seqCount <= 64u) {
				// up to 64 sequences (a human genome's chromosomes): every lane tests one -- one round of loads where the search below makes
				// five or six dependent ones; the sequences do not overlap, so at most one lane holds the position
				const int lane = fresh_lane(this->lane);
				const bool mine = (uint32_t)lane < X.seqCount;
				const uint32_t o = mine ? X.seqOffsets[lane] : 0u, sl = mine ? X.seqLengths[lane] : 0u;
				const uint64_t m = __ballot(mine && top_db >= o && top_db < o + sl);
				if (m) {
					const int k = __builtin_ctzll(m);
					found = (uint32_t)k; coord = top_db - rdlane(o, k); tl = rdlane(sl, k);
				}
				hi = 0xFFFFFFFFu;  // (the search below is skipped)
			}
			while (lo <= hi && hi != 0xFFFFFFFFu) {
				uint32_t k = (lo + hi) / 2;
				uint32_t o = X.seqOffsets[k], sl = X.seqLengths[k];
				if (top_db >= o && top_db < o + sl) { found = k; coord = top_db - o; tl = sl; break; }
				if (top_db > o) lo = k + 1;
				else hi = k - 1;
			}
			if (found != 0xFFFFFFFFu && coord + (uint32_t)QL <= tl) {
				res.dbpos = top_db; res.seq_index = found; res.coord = coord; res.plus = top_plus ? 1 : 0;
				return top_nops > 0;
			}
		}
		return false;
	}

	// search1m6.cpp:9-33
	__device__ __forceinline__ uint32_t calc_mapq() const {
		if (hitCount == 0) return 0;
		if (best <= 0) return 0;
		double bp = (double)QL;
		double sec = (double)second;
		if (sec < bp / 2.0) {
			sec = bp / 2.0;
			if ((double)best <= sec) return 0;
		}
		double fract = (double)best / bp;
		double drop = (double)best - sec;
		if (drop > 40) drop = 40;
		double x = drop * fract;
		x = x * fract;
		uint32_t mapq = (uint32_t)x;
		if (mapq > 40) mapq = 40;
		return mapq;
	}

	// ---- this read's k-mer slots, as the LDS-DMA gathers of probe_gather left them ----
	// entry e = strand * KCH*64 + i, i = the PLUS-strand position of the k-mer's bases: the minus-strand k-mer at minus
	// position q covers the same bases as the plus-strand k-mer at nwords-1-q.  pr_lo / pr_hi = the two aligned dwords
	// around the 5-byte slot, pr_sl = the low half of the slot number (its two low bits say where the slot starts in
	// them: 5*slot = slot mod 4), pr_hb = bit 32 of the slot numbers of a chunk as one ballot word per [strand][chunk].
	// A position without a k-mer gathered zeros: tally 0 = TALLY_FREE.
	uint32_t *pr_lo, *pr_hi, *pr_sl;
	uint64_t *pr_hb;
	uint32_t pf_sink = 0;  // LDS address of the L2 touches' sink (dev_common.h: glds_touch); 0: no touches
	__device__ __forceinline__ void probe_get(int s, int q, uint32_t &tally, uint32_t &pos) const {
		const int e = s * KCH * 64 + (s ? nwords - 1 - q : q);
		if constexpr (LAYOUT == 1) {
			pos = pr_lo[e];
			tally = pr_hi[e] & 0xFFu;
			return;
		}
		const uint32_t lo = pr_lo[e], hi = pr_hi[e], sl = pr_sl[e];
		const uint64_t v = (((uint64_t)hi << 32) | lo) >> (8u * (sl & 3u));
		tally = (uint32_t)(v & 0xFFu);
		pos = (uint32_t)(v >> 8);
	}

	// State1::SetSlotsVec + GetBlob for the NEXT read while this one is searched, in two parts so that the hashing's
	// registers and the current read's chain heads are never alive together:
	//   probe_hash    nq = that read's bytes in LDS, QLn its length.  Per 64-position chunk: letter ballots, the 2 x 64
	//                 slot numbers (kmer_slots); their low halves go to stage_sl[group][lane], bit 32 and "has a k-mer"
	//                 as ballots to stage_b[2 * group + 0 / 1] -- LDS that is idle between two gather steps
	//   probe_gather  (the current read's chain heads are in registers now, its pr_* entries dead) copies the staged
	//                 slot numbers to pr_sl / pr_hb and issues two LDS-DMA gathers per group: the two dwords of every
	//                 slot go from the 27 GB table straight into pr_lo / pr_hi, no register holds them, and the
	//                 wavefront goes on with the current read
	__device__ __forceinline__ void probe_hash(const uint8_t *nq, int QLn, uint32_t *stage_sl, uint64_t *stage_b) {
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		QLn = fresh_uniform(QLn);
		const uint32_t nwn = (uint32_t)(QLn - (W - 1));
		auto planes = [&](int c, uint64_t &lo, uint64_t &hi, uint64_t &inv, uint64_t &invm) {
			const int p = 64 * c + lane;
			const uint32_t ch = p < QLn ? nq[p] : 0u;
			const uint32_t L = p < QLn ? letter_of(ch) : 4u;
			lo = __ballot(L & 1u);
			hi = __ballot((L >> 1) & 1u);
			inv = __ballot(L > 3u);
			invm = __ballot(L > 3u || ch == 'u');  // lower-case 'u' complements to '?' (alpha.cpp:3005)
		};
		uint64_t lo0, hi0, inv0, invm0;
		planes(0, lo0, hi0, inv0, invm0);
#pragma unroll
		for (int c = 0; c < KCH; ++c) {
			if (64u * c < nwn) {  // wave-uniform: chunks that hold a k-mer start
				uint64_t lo1 = 0, hi1 = 0, inv1 = ~0ull, invm1 = ~0ull;
				if (c + 1 < NCH && 64 * (c + 1) < QLn) planes(c + 1, lo1, hi1, inv1, invm1);
				uint64_t sp, sm;
				bool vp, vm;
				kmer_slots(X, lo0, hi0, inv0, invm0, lo1, hi1, inv1, invm1, lane, 64u * c + lane, nwn, sp, sm, vp, vm);
				stage_sl[c * 64 + lane] = (uint32_t)sp;
				stage_sl[(KCH + c) * 64 + lane] = (uint32_t)sm;
				const uint64_t hbp = __ballot(vp && ((sp >> 32) & 1ull)), hbm = __ballot(vm && ((sm >> 32) & 1ull));
				const uint64_t okp = __ballot(vp), okm = __ballot(vm);
				if (lane == 0) {
					stage_b[2 * c] = hbp; stage_b[2 * c + 1] = okp;
					stage_b[2 * (KCH + c)] = hbm; stage_b[2 * (KCH + c) + 1] = okm;
				}
				lo0 = lo1; hi0 = hi1; inv0 = inv1; invm0 = invm1;
			}
		}
	}
	__device__ __forceinline__ void probe_gather(int QLn, const uint32_t *stage_sl, const uint64_t *stage_b) {
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
		QLn = fresh_uniform(QLn);
		const uint32_t nwn = (uint32_t)(QLn - (W - 1));
		wait_lgkm0();  // every earlier LDS read of the pr_* arrays has returned
#pragma unroll
		for (int g = 0; g < NSEG; ++g) {
			if (64u * (g % KCH) < nwn) {
				const uint64_t hb = stage_b[2 * g], ok = stage_b[2 * g + 1];
				const bool v = (ok >> lane) & 1ull;
				const uint32_t sl = stage_sl[g * 64 + lane];
				if constexpr (LAYOUT == 1) {
					// one aligned 16-byte slot of DevIndex::slot16: position, tally | row length, second position or row index
					const uint64_t slot16 = (uint64_t)sl | (((hb >> lane) & 1ull) << 32);
					const uint8_t *ap = v ? reinterpret_cast<const uint8_t *>(X.slot16) + 16ull * slot16 : reinterpret_cast<const uint8_t *>(g_zero16);
					glds_dword(ap, lds_addr(pr_lo + g * 64));
					glds_dword(ap + 4, lds_addr(pr_hi + g * 64));
					glds_dword(ap + 8, lds_addr(pr_sl + g * 64));
					continue;
				}
#ifdef URX_FAULT_SLOT32  // fault injection for the test suite's own check (profiles/r5/fault_slot32.txt): the slot number cut to 32 bits
				const uint64_t slot = (uint64_t)sl; (void)hb;
#else
				const uint64_t slot = (uint64_t)sl | (((hb >> lane) & 1ull) << 32);
#endif
				pr_sl[g * 64 + lane] = v ? sl : 0u;
				if (lane == 0) pr_hb[g] = hb;
				const uint8_t *ap = v ? gblob + ((5ull * slot) & ~3ull) : reinterpret_cast<const uint8_t *>(g_zero16);
				glds_dword(ap, lds_addr(pr_lo + g * 64));
				glds_dword(ap + 4, lds_addr(pr_hi + g * 64));
			}
		}
	}

	// UFIndex::GetRow_Blob (ufindex.cpp:883-943) for all collision chains of the read at once: lane = query position
	// mod 64, the 2*NCH [strand][chunk] segments advance in lock step so that up to 2*NCH dependent slot loads per lane
	// are in flight; positions go to rowstore[seg][k][lane], row lengths to rl[seg].  In two parts: the chain heads come
	// out of the pr_* arrays (which the next read's probe may then overwrite), the hops out of the slot table.
	__device__ __forceinline__ void walk_heads(uint64_t (&sl)[NSEG], uint32_t (&T)[NSEG], uint32_t (&ps)[NSEG], bool (&act)[NSEG]) const {
		const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
#pragma unroll
		for (int g = 0; g < NSEG; ++g) {
			const int s2 = g / KCH, c = g % KCH;
			const int p = 64 * c + lane;
			T[g] = 0; ps[g] = 0; sl[g] = 0;
			if constexpr (LAYOUT == 1) {
				// sl carries what rows_fetch needs instead of a slot number: row length | (second position or row index) << 32
				if (p < nwords) {
					const int e = s2 * KCH * 64 + (s2 ? nwords - 1 - p : p);
					const uint32_t hi = pr_hi[e];
					ps[g] = pr_lo[e];
					T[g] = hi & 0xFFu;
					sl[g] = (uint64_t)((hi >> 8) & 0xFFu) | ((uint64_t)pr_sl[e] << 32);
				}
				act[g] = (T[g] & TALLY_MY_BIT) != 0 && T[g] != TALLY_BOTH1;
				continue;
			}
			if (p < nwords) probe_get(s2, p, T[g], ps[g]);
			act[g] = (T[g] & TALLY_MY_BIT) != 0 && T[g] != TALLY_BOTH1;
			if (act[g]) {
				const int i = s2 ? nwords - 1 - p : p;
				sl[g] = (uint64_t)pr_sl[s2 * KCH * 64 + i] | (((pr_hb[s2 * KCH + (i >> 6)] >> (i & 63)) & 1ull) << 32);
			}
		}
	}
	__device__ __forceinline__ void walk_run(uint64_t (&sl)[NSEG], uint32_t (&T)[NSEG], uint32_t (&ps)[NSEG], bool (&act)[NSEG], int (&rl)[NSEG]) {
		const uint64_t N = X.slotCount;
		const int maxIx = (int)X.maxIx;
		bool any = false;
#pragma unroll
		for (int g = 0; g < NSEG; ++g) { rl[g] = 0; any |= act[g]; }
		while (any) {
			const int lane = fresh_lane(this->lane);  // dev_common.h: lane-derived values are remade here, not carried (and spilled) from the top of the kernel
#pragma unroll
			for (int g = 0; g < NSEG; ++g) {
				if (!act[g]) continue;
				uint32_t *rs = rowstore + (size_t)g * ROW_CAP * 64 + lane;
				rs[rl[g] * 64] = ps[g];
				++rl[g];
				const uint32_t t = T[g];
				if (rl[g] == maxIx || rl[g] >= ROW_CAP) act[g] = false;
				else if (t == TALLY_PLUS1 || t == TALLY_BOTH1) { rl[g] = 1; act[g] = false; }
				else if (t == TALLY_END) act[g] = false;
				else if (t == TALLY_LONG_MINE || t == TALLY_LONG_OTHER) {
					const uint64_t slotA = addmod(sl[g], ps[g] & 0xFFFFu, N);
					sl[g] = addmod(slotA, ps[g] >> 16, N);
					uint32_t tA, pA;
					load_slot(gblob, slotA, tA, pA);
					rs[(rl[g] - 1) * 64] = pA;
				} else
					sl[g] = addmod(sl[g], t & TALLY_NEXT_MASK, N);
			}
			any = false;
#pragma unroll
			for (int g = 0; g < NSEG; ++g) {
				if (act[g]) load_slot(gblob, sl[g], T[g], ps[g]);
				any |= act[g];
			}
		}
	}

	// Round 4: the rows are not walked, they are LOOKED UP in the layout chain_rows.hip wrote when the index reached the device
	// (DevIndex::rowinfo): the head's info word (one random read) gives the row's length and, with its group's base (a 40 MB
	// array: L2 / MALL), where the row lies in DevIndex::rows.  Nothing is copied: rowstore[g][lane] keeps the row's index in
	// `rows` (0xFFFFFFFF for a PLUS1 head, whose row is its own position, kept in rowstore[NSEG + g][lane]) and the candidate
	// stage reads rows[index + k].  walk_run costs ~25 instructions per hop (64-bit slot arithmetic, the 5-byte unpack) for the
	// longest of the read's chains; this costs one round trip and a dozen instructions per group.
	__device__ __forceinline__ void rows_fetch(uint64_t (&sl)[NSEG], uint32_t (&T)[NSEG], uint32_t (&ps)[NSEG], bool (&act)[NSEG], int (&rl)[NSEG]) {
		const int lane = fresh_lane(this->lane);
		// Round 5: the info entry is two words -- the second is the row's second position.  The first position of a row is the head's
		// own (the probe read it with the slot), so a row of two needs nothing but this entry, and phase 4 (rows of length <= 2) reads
		// no `rows` at all: one dependent round trip and one 64-byte sector less per such row.  rowstore: [g] = the row's index in
		// `rows` (rows of three and more), [NSEG + g] = position 0, [2 NSEG + g] = position 1.
		if constexpr (LAYOUT == 1) {
			// everything came with the probe (DevIndex::slot16): no read at all.  rowstore[g] = second position (rows of two) or the row's
			// index in `rows`, [NSEG + g] = position 0
#pragma unroll
			for (int g = 0; g < NSEG; ++g) {
				rl[g] = act[g] ? (int)(sl[g] & 0xFFu) : 0;
				if (act[g]) {
					rowstore[(size_t)g * 64 + lane] = (uint32_t)(sl[g] >> 32);
					rowstore[(size_t)(NSEG + g) * 64 + lane] = ps[g];
				}
			}
#if (URX_PREFETCH & 2)
			// round 6: the rows of three and more (phase 5 reads their entries 1.. out of DevIndex::rows, one dependent round trip in front
			// of every batch's window gather) are touched into L2 now, behind the next read's probe gathers: phase 4 runs in between
			if (pf_sink) {
				const uint32_t sink = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf_sink);
#pragma unroll
				for (int g = 0; g < NSEG; ++g)
					if (rl[g] > 2) {
						const uint8_t *rp = reinterpret_cast<const uint8_t *>(X.rows + (size_t)(uint32_t)(sl[g] >> 32) + 1);
						const int nb = 4 * (rl[g] - 1);
						glds_dword(rp, sink);
						glds_dword(rp + nb - 4, sink);
						if (nb > 64) glds_dword(rp + 64, sink);
					}
			}
#endif
			return;
		}
		uint2 info[NSEG];
		bool longhead = false;
#pragma unroll
		for (int g = 0; g < NSEG; ++g) {
			info[g] = make_uint2(0u, 0u);
			if (act[g] && T[g] != TALLY_PLUS1) info[g] = X.rowinfo[sl[g]];
			longhead |= act[g] && T[g] == TALLY_LONG_MINE;
		}
		const bool any_long = __ballot(longhead) != 0;  // a head whose own slot holds a long link's steps, not a position: thousands in a table at load 0.95, a hundred at 0.6
#pragma unroll
		for (int g = 0; g < NSEG; ++g) {
			rl[g] = 0;
			if (act[g]) {
				uint32_t p0 = ps[g];
				if (T[g] == TALLY_PLUS1) rl[g] = 1;
				else {
					rl[g] = (int)(info[g].x & 0xFFu);
					uint32_t at = 0;
					if (rl[g] > 2 || (any_long && T[g] == TALLY_LONG_MINE)) at = (uint32_t)(X.rowbase[sl[g] >> 10] + (info[g].x >> 8));
					if (any_long && T[g] == TALLY_LONG_MINE) p0 = X.rows[at];  // position 0 of such a row lies in the link's middle slot: the layout has it
					rowstore[(size_t)g * 64 + lane] = at;
					rowstore[(size_t)(2 * NSEG + g) * 64 + lane] = info[g].y;
				}
				rowstore[(size_t)(NSEG + g) * 64 + lane] = p0;
			}
		}
	}
	// row entry k of the chain of lane l in group seg (ROWS kernels)
	// short_row: the row has at most two entries (the candidate belongs to phase 4's half of the list)
	__device__ __forceinline__ uint32_t row_entry(int seg, int k, int l, bool short_row) const {
		if constexpr (LAYOUT == 1) {
			if (k == 0) return rowstore[(size_t)(NSEG + seg) * 64 + l];
			const uint32_t x = rowstore[(size_t)seg * 64 + l];
			return short_row ? x : X.rows[(size_t)x + (uint32_t)k];
		}
		if (k < 2) return rowstore[(size_t)((k + 1) * NSEG + seg) * 64 + l];
		return X.rows[(size_t)rowstore[(size_t)seg * 64 + l] + (uint32_t)k];
	}

	// exclusive prefix over the first `used` of NS segments of per-lane counts -> pre[]; returns the total.  (Round 6: two segments per prefix sum -- a
	// lane's count is at most ROW_CAP = 32, a segment's total at most 2 048, so two of them ride in the halves of one word and no carry crosses --, and the
	// segments a step does not use are not summed: phases 1-2 fill a quarter of the list's segments.  URX_SCAN_PACK=0: one sum per segment, all of them.)
#ifndef URX_SCAN_PACK
#define URX_SCAN_PACK 0  // measured: -0.8 % on the 250-base search, +0.7 % on the 150-base one (18 more spilled registers in the two-chunk instance): off (profiles/r6/ab_instruction_trims.txt)
#endif
	template <int NS>
	__device__ __forceinline__ int scan_counts(const int (&cnt)[NS], int used) {
		static_assert(NS % 2 == 0, "segments are summed two at a time");
		int carry = 0;
#if URX_SCAN_PACK
#pragma unroll
		for (int sgm = 0; sgm < NS; sgm += 2) {
			if (sgm < used) {  // wave-uniform
				const int inc = wave_prefix_sum(cnt[sgm] | (cnt[sgm + 1] << 16));
				const uint32_t tot = (uint32_t)rdlane(inc, 63);
				const int t0 = (int)(tot & 0xFFFFu), t1 = (int)(tot >> 16);
				pre[sgm * 64 + lane] = (uint16_t)(carry + (inc & 0xFFFF) - cnt[sgm]);
				pre[(sgm + 1) * 64 + lane] = (uint16_t)(carry + t0 + (int)((uint32_t)inc >> 16) - cnt[sgm + 1]);
				carry += t0 + t1;
			}
		}
		if (lane == 0) pre[used * 64] = (uint16_t)carry;
#else
		used = NS;
#pragma unroll
		for (int sgm = 0; sgm < NS; ++sgm) {
			int inc = wave_prefix_sum(cnt[sgm]);
			pre[sgm * 64 + lane] = (uint16_t)(carry + inc - cnt[sgm]);
			carry += rdlane(inc, 63);
		}
		if (lane == 0) pre[NS * 64] = (uint16_t)carry;
#endif
		if (carry > 0xFFFF) status |= URMAPX_ST_HSP_OVERFLOW;  // only a 1024-base read whose every k-mer owns a full chain gets here
		URX_SYNC();
		return carry;
	}

	// candidate g (global order) -> (row, k): the row r with pre[r] <= g < pre[r+1]
	__device__ __forceinline__ void locate(int g, int nrows, int &r, int &k) const {
		int lo = 0, hi = nrows;
		while (lo < hi) {
			int mid = (lo + hi + 1) >> 1;
			if ((int)pre[mid] <= g) lo = mid;
			else hi = mid - 1;
		}
		r = lo;
		k = g - (int)pre[lo];
	}
};

// waves per SIMD the register allocation aims at.  Reads <= 192: 4 (128 VGPRs, some registers spilled to scratch in
// cold paths; LDS per block is kept under 10 KB for the same 16 blocks per CU) -- measured 10 % faster than 3 waves
// with no spills.  Reads <= 320: 2 (the five-word bit vectors and the window loads in flight do not fit fewer
// registers without hundreds of spills).
#ifndef SEARCH_WAVES_NCH3
#define SEARCH_WAVES_NCH3 4
#endif
#ifndef SEARCH_WAVES_NCH4
#define SEARCH_WAVES_NCH4 3
#endif
#define SEARCH_WAVES_PER_EU(NCH) ((NCH) <= 3 ? SEARCH_WAVES_NCH3 : (NCH) == 4 ? SEARCH_WAVES_NCH4 : (NCH) == 5 ? 2 : 1)
// The kernel is a software pipeline over the reads a block takes from the ticket counter.  While read i is searched,
//   * the bytes of read i+1 arrive in LDS (one LDS-DMA load issued right after read i's own bytes were taken out), and
//   * between phases 3 and 4 -- when the chain heads of read i are in registers and its slot entries are dead -- the
//     k-mers of read i+1 are hashed and their 2 x (QL-W+1) slots gathered straight into LDS (probe_hash, probe_gather): the loads
//     are in flight behind the chain walk and the phase-4/5 window gathers of read i and have landed when read i+1
//     begins.  There is no probe launch and no probe array in HBM: what used to be 13 bytes written and read back per
//     k-mer stays in 12 bytes of LDS.
// DBG = true: the diagnostic instantiation (URMAPX_PHASE_STATS / URMAPX_DEBUG_STOP): per-phase cycle stamps, per-read
// cycle counts and schedule cuts (stop after step 1 / 3 / 4; 100 = setup and output only; 104 = up to the chain walks).
// The production instantiation (DBG = false) contains none of that code.
// ROWS: the chain rows come out of the layout built with the index (rows_fetch) instead of being walked hop by hop
// PART (round 5): 0 = the whole of phases 1-5 with phase 3's alignments inline (the second pass, the diagnostic build, indexes
// without the row layout); 1 = the same, but a read whose phase 3 has something to align is PARKED there (DpJobs for dp_kernel,
// state for PART 2) and a read that finds no room for its phase-6 jobs goes to the second pass -- no banded DP in this kernel;
// 2 = the second launch: the reads PART 1 parked at phase 3, from the replay of AlignHSP's bookkeeping over their jobs onwards
// (phases 4-5, then parked for phase 6 like any other read).  n = the batch's reads (PART 2: read from dp3's counter).
// ROWS: 0 = chains walked hop by hop, 1 = rows looked up in the row layout (rows_fetch), 2 = everything with the probe (DevIndex::slot16)
// KCH: see SearchWave.  KCH < NCH (the slot16 layout only), and the class of reads of up to 128 bases, whose block has the room anyway: the block's row
// store -- the chain heads' first positions and second positions / row indexes that rows_fetch keeps for the candidate scans of phases 4 and 5 -- lies in
// LDS (what the smaller arrays free), not in global scratch: no store per chain group and no L2 round trip in front of every scan step
template <int NCH, bool OVF, bool DBG, int ROWS = 0, int PART = 0, int KCH = NCH>
__global__ __launch_bounds__(64, SEARCH_WAVES_PER_EU(NCH)) void search_se_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint32_t n,
                                                       urmapx_result *__restrict__ results,
                                                       urmapx_path_op *__restrict__ path_ops, uint32_t *path_used,
                                                       uint32_t *stats, uint8_t *scratch, size_t scratch_stride,
                                                       const uint8_t *__restrict__ g_seq, const uint8_t *__restrict__ g_blob,
                                                       const uint4 *__restrict__ g_seqp,
                                                       uint32_t *ticket, int hsp_lds_cap, uint32_t *ovf_list, uint2 *hsp_ovf_base,
                                                       DpWork dp, DpWork dp3) {
	static_assert(PART == 0 || (!OVF && !DBG && ROWS == 1), "phase 3 is parked by the production first pass only");
	// stats != nullptr (URMAPX_PHASE_STATS): per-phase shader cycles are accumulated into stats (u64 each, from byte 8)
	static_assert(KCH == NCH || (ROWS == 2 && PART == 0 && !OVF && !DBG), "fewer k-mer chunks: the production first pass on the slot16 layout");
	using SW = SearchWave<NCH, OVF, ROWS == 2 ? 1 : 0, KCH>;
	constexpr int NQ_BYTES = ((SW::QMAX + 4 + 255) / 256) * 256;  // the next read's bytes from a 4-byte aligned address on, in 256-byte DMA pieces
	__shared__ __attribute__((aligned(16))) uint8_t sQ2[2 * SW::QMAX];  // plus strand, then reverse complement
	uint8_t *const sQp = sQ2, *const sQm = sQ2 + SW::QMAX;
	__shared__ __attribute__((aligned(16))) uint4 qpl[2 * 2 * NCH];     // both strands as bit planes (dev_common.h): [strand][block of 32]
	__shared__ __attribute__((aligned(16))) uint8_t nextQ[NQ_BYTES];
	__shared__ uint32_t pr_lo[SW::NSEG * 64], pr_hi[SW::NSEG * 64], pr_sl[SW::NSEG * 64];
	__shared__ uint64_t pr_hb[SW::NSEG];
	__shared__ uint16_t top[URMAPX_MAX_PATH_OPS];
	__shared__ __attribute__((aligned(8))) uint16_t pre[2 * SW::NSEG * 64 + 2];
	// the flank run buffers and the candidate path live only inside align_hsp, the candidate prefix array only inside a
	// gather step: they share memory (LDS per block decides how many reads a CU keeps in flight)
	static_assert(2 * OPS_CAP + URMAPX_MAX_PATH_OPS + (SW::QMAX + 64) / 2 <= 2 * SW::NSEG * 64 + 2, "alias");
	uint16_t *const ropsL = pre, *const ropsR = pre + OPS_CAP, *const cand = pre + 2 * OPS_CAP;
	uint8_t *const sT = reinterpret_cast<uint8_t *>(pre + 2 * OPS_CAP + URMAPX_MAX_PATH_OPS);  // AlignHSP's target window
	// viterbi_wave<B_LDS> wants band_radius + 1 bytes of this block's LDS in front of the window (viterbi_dev.h): it lies inside `pre`
	static_assert((2 * OPS_CAP + URMAPX_MAX_PATH_OPS) * 2 >= 64, "AlignHSP's window must not start near LDS offset 0");
	__shared__ uint32_t hsp_db[HSP_CAP], hsp_pk[HSP_CAP];
	// candidate queue (ring): reference position, query position | plus << 14 | second phase << 15
	__shared__ __attribute__((aligned(8))) uint32_t cq_db[128];
	__shared__ uint16_t cq_qp[128];
	constexpr bool PF = (URX_PREFETCH & 1) != 0;
	__shared__ uint32_t pf_sink[URX_PREFETCH ? 64 : 1];  // where the L2 touches land (glds_touch); never read
	constexpr bool RS_LDS = KCH < NCH || (NCH == 2 && ROWS == 2 && PART == 0 && !OVF && !DBG);
	__shared__ uint32_t rs_lds[RS_LDS ? 2 * SW::NSEG * 64 : 1];  // the row store (rows_fetch, row_entry)
	// between two gather steps both are idle: the next read's slot numbers are staged there on their way to the pr_* arrays
	static_assert(SW::NSEG * 64 * 4 <= sizeof(pre) && 2 * SW::NSEG * 8 <= sizeof(cq_db), "staging");
	uint32_t *const stage_sl = reinterpret_cast<uint32_t *>(pre);
	uint64_t *const stage_b = reinterpret_cast<uint64_t *>(cq_db);

	int lane = threadIdx.x;
	const int W = (int)X.W;
	const int dbg_stop = (DBG && stats) ? (int)stats[0] : 0;  // diagnostic only (URMAPX_DEBUG_STOP)
	const bool timing = DBG && stats && stats[1] == 0;
	SW S(X, P, lane);
	S.W = W;
	S.gseq = g_seq; S.gblob = g_blob;
	S.sQ[0] = sQp; S.sQ[1] = sQm; S.sT = sT;
	S.ropsL = ropsL; S.ropsR = ropsR; S.cand = cand; S.top = top; S.pre = pre;
	S.hsp_db = hsp_db; S.hsp_pk = hsp_pk;
	S.pr_lo = pr_lo; S.pr_hi = pr_hi; S.pr_sl = pr_sl; S.pr_hb = pr_hb;
	if constexpr ((URX_PREFETCH & 2) != 0) S.pf_sink = lds_addr(pf_sink);
	{
		uint8_t *sc = scratch + (size_t)blockIdx.x * scratch_stride;
		if constexpr (RS_LDS) S.rowstore = rs_lds;
		else S.rowstore = reinterpret_cast<uint32_t *>(sc);
		S.ws.carve(sc + (size_t)SW::NSEG * ROW_CAP * 64 * 4, SW::QMAX, SW::WIDE_LB);
		// the trace cells of phase 3's banded DP live in this block's global scratch (phase 6 has kernels of its own with
		// the trace in LDS): 6 KB of LDS per block went to the slot entries instead
		S.tb = reinterpret_cast<uint32_t *>(sc + search_tb_offset(NCH));
		S.hit_tail = reinterpret_cast<uint32_t *>(sc + search_tail_offset(NCH));
		S.hsp_ovf = hsp_ovf_base + (size_t)blockIdx.x * (HSP_TOTAL_CAP - HSP_CAP);
		S.hsp_lds = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? (hsp_lds_cap & ~63) : HSP_CAP;  // multiple of 64
		S.hit_cap = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? S.hsp_lds / 4 : 64 * SE_HITW1;
		S.hit_wsh = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? 4 : 6;
	}
	const bool geom_ok = W <= 32 && X.maxIx <= (uint32_t)ROW_CAP;
	auto len_ok = [&](int ql) { return geom_ok && ql >= W && ql <= SW::QMAX && ql - (W - 1) <= 64 * KCH; };

	// Reads are handed out by a ticket counter, not by a fixed stride: the cost of a read is heavy-tailed (a read in a
	// repeat family aligns up to 256 HSPs in phase 6), and with a fixed assignment the blocks that draw such reads
	// finish long after the rest.
	// TICKET_CHUNK reads per ticket: same-address atomics retire at ~88 M/s on this device (a bare ticket loop over 1 M
	// reads takes 11.4 ms), which would cap the kernel not far above its current rate.
	if constexpr (OVF) {
		n = ovf_list[0];  // how many reads the first pass flagged (usually none: then no block asks the ticket counter)
		if (n == 0) return;
	}
	if constexpr (PART == 2) {
		n = dp3.counters[1] < dp3.fin_cap ? dp3.counters[1] : dp3.fin_cap;  // the reads the first launch parked at phase 3
		if (n == 0) return;
	}
	const uint4 *const fin3 = reinterpret_cast<const uint4 *>(dp3.fin_list);
	uint32_t r_next = 0, r_end = 0;
	auto take = [&](uint32_t &ri) -> bool {  // the next read of this block; false: the batch is used up
		if (r_next == r_end) {
			// every lane takes part in the atomic (lanes 1..63 add 0): with `if (lane == 0) atomicAdd` here the compiler's
			// wave-level atomic rewrite turns the loop divergent and the kernel hangs or faults (seen twice)
			constexpr uint32_t CHUNK = OVF ? 1u : (uint32_t)TICKET_CHUNK;  // second pass: few, heavy reads -- one per ticket
			r_next = uni(atomicAdd(ticket, lane == 0 ? CHUNK : 0u));
			if (r_next >= n) { r_end = r_next; return false; }
			r_end = r_next + CHUNK < n ? r_next + CHUNK : n;
		}
		ri = r_next++;
		if constexpr (OVF) ri = ovf_list[1 + ri];  // second pass: the reads the first pass flagged
		return true;
	};
	// a read's bytes into nextQ by LDS-DMA: whole dwords from the 4-byte aligned address at or below its first byte
	// (loads inside the last dword of a buffer stay inside its allocation)
	auto fetch_bytes = [&](uint64_t o, int ql) {
		const uintptr_t a = reinterpret_cast<uintptr_t>(bases + o);
		const uint8_t *src = reinterpret_cast<const uint8_t *>(a & ~(uintptr_t)3);
		const int nb = (int)(a & 3) + ql;
#pragma unroll
		for (int k = 0; k < NQ_BYTES / 256; ++k)
			if (256 * k < nb) {
				const int ob = 256 * k + 4 * lane;
				glds_dword(src + (ob < nb ? ob : 0), lds_addr(nextQ + 256 * k));
			}
	};

	bool have_cur = false;
	uint32_t r = 0;
	uint64_t off = 0;
	int QL = 0;
	uint32_t p3_e = 0, p3_jb = 0, p3_nj = 0;  // PART 2: this read's place in the parking lot, its phase-3 jobs
	for (;;) {
		uint32_t rn = 0;
		const bool have_next = take(rn);
		uint64_t noff = 0;
		int nQL = 0;
		uint32_t n_e = 0, n_jb = 0, n_nj = 0;
		if constexpr (PART == 2) {
			if (have_next) {
				const uint4 ent = fin3[rn];
				n_e = rn; rn = uni(ent.x); n_jb = uni(ent.y); n_nj = uni(ent.z);
				noff = offs[rn]; nQL = (int)uni(ent.w);
			}
		} else if (have_next) { noff = offs[rn]; nQL = (int)(offs[rn + 1] - noff); }
		const bool next_ok = have_next && len_ok(nQL);
		const int nmis = (int)(reinterpret_cast<uintptr_t>(bases + noff) & 3);
		const bool cur_ok = have_cur && len_ok(QL);

		const uint64_t t_read0 = timing ? __builtin_amdgcn_s_memtime() : 0;
		urmapx_result res;
		res.dbpos = 0xFFFFFFFFu; res.seq_index = 0xFFFFFFFFu; res.coord = 0xFFFFFFFFu;
		res.score = 0; res.second = 0; res.mapq = 0; res.plus = 0; res.exit_phase = 0; res.status = 0;
		res.hit_count = 0; res.path_nops = 0; res.path_off = 0;
		bool parked = false;  // phase 6 handed to dp_kernel + finalize_se_kernel, which writes the result
		if (have_cur && !cur_ok) res.status = URMAPX_ST_BAD_LENGTH;
		bool fetched = false;
		bool q_other = false;
		int phase = 1;
		bool done = !cur_ok;
		uint64_t tstamp = timing ? __builtin_amdgcn_s_memtime() : 0;
		auto lapc = [&](int slot) {
			if (!timing) return;
			uint64_t now = __builtin_amdgcn_s_memtime();
			if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(stats) + 1 + slot, (unsigned long long)(now - tstamp));
			tstamp = now;
		};
		if (cur_ok) {
			S.QL = QL;
			S.nwords = QL - (W - 1);
#pragma unroll
			for (int w = 0; w < SW::HITW; ++w) S.hit_db[w] = 0;
			S.hitCount = 0; S.hspCount = 0;
			S.maxPen = P.max_penalty; S.best = 0; S.second = 0; S.bestHSP = 0;
			S.haveTop = false; S.top_db = 0; S.top_plus = false; S.top_nops = 0; S.status = 0;

			// this read's bytes and slot entries have landed (issued while the previous read was searched)
			wait_vm0();
			URX_SYNC();
			const int mis = (int)(reinterpret_cast<uintptr_t>(bases + off) & 3);
			// (dev_common.h: URX_ACGT_FAST) a read of upper-case ACGT only -- nearly every read -- takes the four-instruction complement and code
			uint32_t chv[NCH];
			bool plain = true;
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int p = 64 * c + lane;
				chv[c] = p < QL ? (uint32_t)nextQ[mis + p] : (uint32_t)'A';
				plain = plain && is_upper_acgt(chv[c]);
			}
			const bool acgt = URX_ACGT_FAST && __ballot(!plain) == 0;  // wave-uniform
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int p = 64 * c + lane;
				if (p < QL) {
					sQp[p] = (uint8_t)chv[c];
					sQm[QL - 1 - p] = (uint8_t)(acgt ? comp_char_acgt(chv[c]) : comp_char(chv[c]));
				}
			}
			URX_SYNC();
			// both strands as bit planes of 4-bit codes (dev_common.h); a read holding a byte outside the code list
			// compares ASCII windows instead
			uint64_t oth = 0;
#pragma unroll
			for (int s = 0; s < 2; ++s) {
#pragma unroll
				for (int c = 0; c < NCH; ++c) {
					if (64 * c < QL) {
						const int p = 64 * c + lane;
						uint64_t b0, b1, b2 = 0, b3 = 0;
						if (acgt) {  // codes 0..3: the two upper planes are empty
							const uint32_t code = p < QL ? seq_code_acgt(sQ2[s * SW::QMAX + p]) : 0u;
							b0 = __ballot(code & 1u); b1 = __ballot(code & 2u);
						} else {
							const uint32_t code = p < QL ? seq_code(sQ2[s * SW::QMAX + p], SEQ_CODE_QOTHER) : 0u;
							b0 = __ballot(code & 1u); b1 = __ballot(code & 2u); b2 = __ballot(code & 4u); b3 = __ballot(code & 8u);
							oth |= __ballot(code == SEQ_CODE_QOTHER);
						}
						if (lane < 2) {
							const int shh = 32 * lane;
							qpl[s * 2 * NCH + 2 * c + lane] = make_uint4((uint32_t)(b0 >> shh), (uint32_t)(b1 >> shh), (uint32_t)(b2 >> shh), (uint32_t)(b3 >> shh));
						}
					}
				}
			}
			q_other = oth != 0;
			URX_SYNC();
			// nextQ is free again: the bytes of the read after this one
			if (next_ok) { wait_lgkm0(); fetch_bytes(noff, nQL); fetched = true; }
			lapc(0);
		}
		int nwords = QL - (W - 1);
		if constexpr (PART == 2) {
			if (cur_ok) {
				// back from the parking lot: state, HSP list and slot entries, then AlignHSP's bookkeeping (alignhsp.cpp:60-172) over the
				// jobs dp_kernel ran, in HSP order against the cap as it falls -- what finalize_se_kernel does for phase 6
				phase = S.resume3(dp3.state + (size_t)p3_e * SW::P3_WORDS);
				for (uint32_t k0 = 0; k0 < p3_nj; k0 += 64) {
					const uint32_t k = k0 + (uint32_t)lane;
					uint32_t jpk = 0;
					uint4 jw = make_uint4(0u, 0u, 0u, 0u);
					if (k < p3_nj) {
						const uint32_t *jp = reinterpret_cast<const uint32_t *>(dp3.jobs + p3_jb + k);
						jpk = jp[2];
						jw = *reinterpret_cast<const uint4 *>(jp + 4);
					}
					const int nt = (int)(p3_nj - k0 < 64u ? p3_nj - k0 : 64u);
					for (int t = 0; t < nt; ++t) {
						DpJob J;
						J.read = r; J.startdb = 0; J.maxpen = 0; J.k = 0; J.pad[0] = J.pad[1] = 0;
						J.pk = rdlane(jpk, t);
						J.combined_tlo = rdlane(jw.x, t);
						const uint32_t sc = rdlane(jw.y, t), fl = rdlane(jw.z, t);
						J.left_score = (int16_t)(sc & 0xFFFFu); J.right_score = (int16_t)(sc >> 16);
						J.nops = (uint8_t)(fl & 0xFFu); J.flags = (uint8_t)((fl >> 8) & 0xFFu); J.vst_l = (uint8_t)((fl >> 16) & 0xFFu); J.vst_r = (uint8_t)(fl >> 24);
						S.consume_job(J, dp3.ops + (size_t)(p3_jb + k0 + (uint32_t)t) * DP_JOB_OPS);
					}
				}
				if (S.best >= QL + P.xphase1 * P.mismatch_score) done = true;  // search1m6.cpp:170-171
			}
		}
		const int minScore1 = QL + P.xphase1 * P.mismatch_score;
		const int minScore3 = QL + P.xphase3 * P.mismatch_score;
		const int minScore4 = QL + P.xphase4 * P.mismatch_score;
		const int termHSP3 = (QL * P.term_hsp_score_pct_phase3) / 100;
		const int minhsp = (int)((uint32_t)P.min_hsp_score_pct * (uint32_t)(QL > 0 ? QL : 0) / 100.0);

		// The six phases of Search_Lo as ONE loop, so that the gather/consume code and the DP code exist once
		// (the kernel has to stay inside the instruction cache):
		//   1, 2  BOTH1 seeds on / off the stride W      3  AlignHSP if the best HSP is long enough
		//   5     (not a phase) chain heads out of LDS, the NEXT read's probe, then the chain walks
		//   4, 5' chain rows of length <= 2 / > 2        6  AlignHSP
		// Phases 1+2 and 4+5 each form ONE candidate list (gathered together, so that batches stay full); the ordered
		// part applies the phase boundary (exit test after phase 4) when it crosses it.  A read that is done early, or
		// has a bad length, still passes through step 5 for the next read's probe.
		int rl[SW::NSEG];
#pragma unroll
		for (int g = 0; g < SW::NSEG; ++g) rl[g] = 0;
		for (int step = (cur_ok && PART != 2) ? 1 : 5;;) {
			QL = fresh_uniform(QL);  // see dev_common.h: keeps the length-dependent masks out of the SGPR spill lanes
			S.QL = QL; nwords = QL - (W - 1); S.nwords = nwords;
			lane = fresh_lane(lane); S.lane = lane;
			if (DBG && dbg_stop && step != 5 && (dbg_stop == 100 || (dbg_stop < 100 && step > dbg_stop))) {  // diagnostic schedule cut
				done = true;
				if (step == 4 || step == 6) break;  // the next read's probe went out in step 5 already
				step = 5;
			}
			if (step == 5) {
				const bool go = !done;
				uint64_t wsl[SW::NSEG];
				uint32_t wT[SW::NSEG], wps[SW::NSEG];
				bool wact[SW::NSEG];
				if (PART != 2 && next_ok) {  // (PART 2: the next read's slot entries come out of its parked state)
					if (!fetched) fetch_bytes(noff, nQL);
					wait_vm0();
					URX_SYNC();
					S.probe_hash(nextQ + nmis, nQL, stage_sl, stage_b);
					URX_SYNC();
				}
				if (go) S.walk_heads(wsl, wT, wps, wact);
				if (PART != 2 && next_ok) S.probe_gather(nQL, stage_sl, stage_b);
				if (PART == 2 && next_ok && !fetched) { fetch_bytes(noff, nQL); fetched = true; }
				if (!go) break;
				if constexpr (ROWS != 0) S.rows_fetch(wsl, wT, wps, wact, rl);
				else S.walk_run(wsl, wT, wps, wact, rl);
				URX_SYNC();
				lapc(3);
				if (DBG && dbg_stop == 104) { done = true; break; }
				step = 4;
				continue;
			}
			phase = step;
			if (step == 3 || step == 6) {
				if constexpr (PART != 0) {
					// no banded DP in this kernel: the alignments of phase 3 and of phase 6 are DpJobs.  A read whose lists have
					// outgrown the first pass's is mapped again by the second pass whatever happens here.
					if (S.status & (URMAPX_ST_HSP_OVERFLOW | URMAPX_ST_HIT_OVERFLOW)) {
						if (step == 6) break;
						done = true;
					} else if (step == 6) {
						const int rc = S.template park_for_dp<false>(dp, r, 6);
						if (rc > 0) parked = true;
						else if (rc < 0) S.status |= URMAPX_ST_HSP_OVERFLOW;  // no room for its jobs: the second pass (which aligns inline) maps it
						break;
					} else if (S.bestHSP > termHSP3) {
						const int rc = S.template park_for_dp<true>(dp3, r, 3);
						if (rc > 0) { parked = true; done = true; }
						else if (rc < 0) { S.status |= URMAPX_ST_HSP_OVERFLOW; done = true; }
						else S.mark_all_aligned();
					}
					step = 5;
					continue;
				} else {
				if (step == 6 && !OVF && (S.status & (URMAPX_ST_HSP_OVERFLOW | URMAPX_ST_HIT_OVERFLOW))) break;  // the second pass maps this read again
				if (step == 6 && dp.jobs != nullptr && S.park_for_dp(dp, r, 6) > 0) { parked = true; break; }
				if (step == 6 || S.bestHSP > termHSP3) {
					for (int k = 0; k < S.hspCount; ++k) S.align_hsp(k);
					if (step == 3 && S.best >= minScore1) done = true;
				}
				lapc(step == 3 ? 2 : 6);
				if (step == 6) break;
				step = 5;
				continue;
				}
			}
			int cnt[2 * SW::NSEG];
#pragma unroll
			for (int g = 0; g < 2 * SW::NSEG; ++g) cnt[g] = 0;
			if (step == 1) {  // segments [0, KCH): phase 1 (stride positions), [KCH, 2 KCH): phase 2
#pragma unroll
				for (int c = 0; c < KCH; ++c) {
					const int p = 64 * c + lane;
					int nb1 = 0;
					if (p < nwords) {
						uint32_t t0, p0, t1, p1;
						S.probe_get(0, p, t0, p0);
						S.probe_get(1, p, t1, p1);
						nb1 = (t0 == TALLY_BOTH1 ? 1 : 0) + (t1 == TALLY_BOTH1 ? 1 : 0);
					}
					const bool onStride = (p % W) == 0;
					cnt[c] = onStride ? nb1 : 0;
					cnt[KCH + c] = onStride ? 0 : nb1;
				}
			} else {  // segments [0, NSEG): rows <= 2 (phase 4), [NSEG, 2 NSEG): rows > 2 (phase 5)
#pragma unroll
				for (int g = 0; g < SW::NSEG; ++g) {
					cnt[g] = rl[g] <= 2 ? rl[g] : 0;
					cnt[SW::NSEG + g] = rl[g] > 2 ? rl[g] : 0;
				}
			}
			const int used = URX_SCAN_PACK ? (step == 1 ? 2 * KCH : 2 * SW::NSEG) : 2 * SW::NSEG;  // segments of this step's list: phases 1-2 fill the first 2 KCH
			const int total = S.template scan_counts<2 * SW::NSEG>(cnt, used);
			const int totalFirst = (int)S.pre[(step == 1 ? KCH : SW::NSEG) * 64];  // candidates of the first of the two phases
			bool crossed = false;
			// The candidate stream is first filtered -- a candidate on the 64-base diagonal block of a hit already found
			// returns at once in the reference (extendpen.cpp:15-17), and hits are never removed -- and the survivors are
			// compacted, in order, into a small LDS queue; gather/consume then always run on full batches.
			int scanned = 0, qhead = 0, qcount = 0;
			while (!done && (scanned < total || qcount > 0)) {
				QL = fresh_uniform(QL);
				S.QL = QL; nwords = QL - (W - 1); S.nwords = nwords;
				lane = fresh_lane(lane); S.lane = lane;
				uint64_t tsub = timing ? __builtin_amdgcn_s_memtime() : 0;
				auto laps = [&](int slot) {
					if (!timing) return;
					uint64_t now = __builtin_amdgcn_s_memtime();
					if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(stats) + 1 + slot, (unsigned long long)(now - tsub));
					tsub = now;
				};
				// (PF: the scan runs one step ahead -- up to 128 queued -- so that the batch after this one is known when its windows are touched)
				while (qcount < (PF ? 65 : 64) && scanned < total) {
					const int g = scanned + lane;
					uint32_t s_qpos = 0, s_db = 0;
					bool s_plus = true, ok = false;
					if (g < total) {
						int row, k;
						S.locate(g, used * 64, row, k);
						if (step == 1) {  // BOTH1 seeds: plus-strand seed first, then minus (search1m6.cpp:69-108)
							if (row >= KCH * 64) row -= KCH * 64;
							s_qpos = (uint32_t)row;
							uint32_t tp, pp, tm, pm;
							S.probe_get(0, row, tp, pp);
							S.probe_get(1, row, tm, pm);
							if (k == 0 && tp == TALLY_BOTH1) { s_plus = true; s_db = pp; }
							else { s_plus = false; s_db = pm; }
						} else {  // chain rows: [strand][chunk][k][lane]
							int seg = row >> 6;
							const int l = row & 63;
							const bool short_row = seg < SW::NSEG;  // the list's first half: rows of at most two (phase 4)
							if (seg >= SW::NSEG) seg -= SW::NSEG;
							s_plus = seg < KCH;
							s_qpos = (uint32_t)((seg - (s_plus ? 0 : KCH)) * 64 + l);
							if constexpr (ROWS != 0) s_db = S.row_entry(seg, k, l, short_row);
							else s_db = S.rowstore[((size_t)seg * ROW_CAP + k) * 64 + l];
						}
						ok = s_db >= s_qpos;  // extendpen.cpp:12-13
					}
					ok = ok && !S.overlaps_any_hit(s_db - s_qpos);
					const uint64_t m = __ballot(ok);
					if (ok) {
						const int pos = (qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & 127;
						cq_db[pos] = s_db;
						cq_qp[pos] = (uint16_t)(s_qpos | (s_plus ? 0x4000u : 0u) | (g >= totalFirst ? 0x8000u : 0u));
					}
					qcount += __builtin_popcountll(m);
					scanned += 64;
				}
				URX_SYNC();
				const int nb = qcount < 64 ? qcount : 64;
				if (nb == 0) break;
				uint32_t c_qpos = 0, c_db = 0;
				bool c_plus = true, c_second = false;
				const bool c_ok = lane < nb;
				if (c_ok) {
					const int pos = (qhead + lane) & 127;
					c_db = cq_db[pos];
					const uint32_t qp = cq_qp[pos];
					c_qpos = qp & 0x3FFFu; c_plus = (qp & 0x4000u) != 0; c_second = (qp & 0x8000u) != 0;
				}
				qhead = (qhead + nb) & 127; qcount -= nb;
				URX_SYNC();
				bool c_live = c_ok;
				if constexpr (PF) {
					// scanned a step earlier than it is gathered: a candidate on the diagonal block of a hit found meanwhile returns at once
					// in the reference (extendpen.cpp:15-17) -- it is dropped here as the scan would have dropped it
					c_live = c_ok && !S.overlaps_any_hit(c_db - c_qpos);
					if (__ballot(c_live) == 0) { laps(8); continue; }
				}
				laps(8);
				uint64_t mm[NCH];
#pragma unroll
				for (int c = 0; c < NCH; ++c) mm[c] = 0;
				if (q_other) {  // wave-uniform: a read with bytes outside the code list (IUPAC beyond N, 'u')
					if (c_live) lane_mismatch_mask<NCH>(g_seq, c_db - c_qpos, sQ2 + (c_plus ? 0 : SW::QMAX), QL, mm);
				} else {
					if (c_live) lane_mismatch_planes<NCH>(g_seqp, c_db - c_qpos, qpl + (c_plus ? 0 : 2 * NCH), QL, mm);
				}
				if constexpr (PF) {
					// the windows of the batch AFTER this one (already in the queue) on their way into L2 while this one is walked and consumed
					if (!q_other && lane < qcount) {
						const int pos = (qhead + lane) & 127;
						const uint32_t ndblo = cq_db[pos] - (uint32_t)(cq_qp[pos] & 0x3FFFu);
						glds_touch(reinterpret_cast<const uint8_t *>(g_seqp + (ndblo >> 5)), 16 * ((QL - 1) / 32 + 2), lds_addr(pf_sink));
					}
				}
				laps(9);
				// ExtendPen's two x-drop walks (extendpen.cpp:25-78), every lane on its own bit vector.  The accumulated
				// penalty only grows along the walk and the cap only falls: a lane over the cap as it stands now is over
				// it at its turn and stops; the others finish and are compared with the cap in order below.
				// A candidate changes the state only as a full-length hit under the cap or as an HSP scoring at least
				// max(MinHSPScore, best - 4) (extendpen.cpp:79-95, state1.cpp:555).  With n mismatches outside its seed
				// the first costs -mis * n and no score of the second exceeds QL - n: lanes that can be neither do not walk.
				int e_kind = 0, e_bst = 0, e_start = 0, e_end = 0, e_pen = 0;
				bool worth = false;
				if (c_live) {
					const int nmis = mismatches_outside_seed<NCH>(mm, (int)c_qpos, W);
					const int floor2 = minhsp > S.best - 4 ? minhsp : S.best - 4;
					worth = -P.mismatch_score * nmis <= S.maxPen || QL - nmis >= floor2;
				}
				if (worth) {
					xdrop_walk_lane<NCH>(mm, (int)c_qpos, W, QL, P.mismatch_score, P.xdrop, S.maxPen, e_bst, e_start, e_end, e_pen);
					if (e_start == 0 && e_end == QL - 1) e_kind = 1;
					else if (e_bst >= minhsp) e_kind = 2;
				}
				laps(10);
				// order-dependent part: only candidates that can change the state, in the reference's order.  Lanes that
				// cannot change it are dropped, up front and again after every change: the penalty cap only falls, the best
				// score only rises and hits are only added, so a candidate failing extendpen.cpp:15-17, extendpen.cpp:43-44
				// or state1.cpp:555 now fails it at its turn too.
				const uint32_t my_dblo = c_db - c_qpos;
				uint64_t todo = __ballot(e_kind != 0 && e_pen <= S.maxPen && !(e_kind == 2 && e_bst < S.best - 4) &&
				                         !S.overlaps_any_hit(my_dblo));
				while (todo) {
					lane = fresh_lane(lane); S.lane = lane;
					const int t = __builtin_ctzll(todo);
					todo &= todo - 1;
					if (rdlane((uint32_t)c_second, t) != 0 && !crossed) {  // first candidate of the second phase of this list
						crossed = true;
						if (step == 4 && S.best >= minScore3) { done = true; break; }  // exit test between phases 4 and 5
					}
					const uint32_t dblo = rdlane(my_dblo, t);
					if (S.overlaps_hit(dblo)) continue;          // extendpen.cpp:15-17
					if (rdlane(e_pen, t) > S.maxPen) continue;   // extendpen.cpp:43-44,69-70
					const bool pl = rdlane((uint32_t)c_plus, t) != 0;
					const int bst = rdlane(e_bst, t);
					const int hc0 = S.hitCount, mp0 = S.maxPen, b0 = S.best;
					if (rdlane(e_kind, t) == 1) {
						S.add_hit(dblo, pl, bst, 0);
						if (step == 1 && bst >= minScore1) { done = true; phase = crossed ? 2 : 1; break; }
					} else {
						const int sp = rdlane(e_start, t), ep = rdlane(e_end, t);
						S.add_hsp((uint32_t)sp, dblo + (uint32_t)sp, pl, (uint32_t)(ep - sp + 1), bst);
						// whatever AddHSPX did, the HSP on this diagonal now scores >= bst (or bst < best - 4): a later HSP
						// candidate on the same diagonal with a score <= bst changes nothing (state1.cpp:555,563-571)
						todo &= ~__ballot(e_kind == 2 && my_dblo == dblo && e_bst <= bst);
					}
					if (S.hitCount != hc0 || S.maxPen != mp0 || S.best != b0)
						todo &= __ballot(e_pen <= S.maxPen && !(e_kind == 2 && e_bst < S.best - 4) &&
						                 !(S.hitCount != hc0 && (my_dblo >> 6) == (dblo >> 6)));
				}
				laps(11);
			}

			if (step == 1) {
				lapc(1);
				if (!done) phase = 2;
				step = done ? 5 : 3;
				continue;
			}
			// step == 4
			lapc(4);
			if (!done && !crossed && S.best >= minScore3) done = true;  // no state-changing phase-5 candidate was met
			if (!done) { phase = 5; if (S.best >= minScore4) done = true; }
			if (done) break;
			step = 6;
		}
		lapc(6);
		if (cur_ok && !parked) S.fill_result(res, phase, path_ops, path_used);
		lapc(7);
		if (have_cur && !parked) {
			if constexpr (!OVF) {
				if (res.status & (URMAPX_ST_HSP_OVERFLOW | URMAPX_ST_HIT_OVERFLOW)) {  // queue the read for the second pass
					if (lane == 0) ovf_list[1 + atomicAdd(ovf_list, 1u)] = r;
				}
			}
			if (lane == 0) results[r] = res;
			if (timing && lane == 0) stats[64 + r] = (uint32_t)((__builtin_amdgcn_s_memtime() - t_read0) >> 4);  // per-read cost, 16-cycle units
		}
		if (!have_next) break;
		have_cur = true; r = rn; off = noff; QL = nQL;
		p3_e = n_e; p3_jb = n_jb; p3_nj = n_nj;
	}
}

// ------------------------------------------------------------------------------------------------
// the jobs of each round as a list: one pass over the k of every job the search kernel made
// ------------------------------------------------------------------------------------------------
// A block takes 4096 consecutive jobs, counts those of each round in LDS, reserves room in the round's list with ONE
// atomic per round (the lists' order does not matter: it only decides which wavefront runs a job) and writes the indices;
// a thread's 16 jobs are consecutive, so the jobs of a read stay together in the list (dp_kernel keeps a read's bases in LDS).
__global__ __launch_bounds__(256) void dp_round_lists_kernel(DpWork dp, DpBounds bounds) {
	const uint32_t *const DP_ROUND_LO = bounds.lo;
	constexpr int PER = 16;
	__shared__ uint32_t cnt[DP_ROUNDS], base[DP_ROUNDS];
	const uint32_t made = dp.counters[0];
	const uint32_t njobs = made < dp.jobs_cap ? made : dp.jobs_cap;
	for (uint32_t c0 = blockIdx.x * 256u * PER; c0 < njobs; c0 += gridDim.x * 256u * PER) {
		if (threadIdx.x < DP_ROUNDS) cnt[threadIdx.x] = 0;
		__syncthreads();
		const uint32_t j0 = c0 + threadIdx.x * PER;
		uint16_t kk[PER];
		if (j0 + PER <= njobs) {
			const uint4 a = *reinterpret_cast<const uint4 *>(dp.kidx + j0), b = *reinterpret_cast<const uint4 *>(dp.kidx + j0 + 8);
			const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
			for (int i = 0; i < PER; ++i) kk[i] = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
		} else {
#pragma unroll
			for (int i = 0; i < PER; ++i) kk[i] = j0 + i < njobs ? dp.kidx[j0 + i] : (uint16_t)0xFFFFu;
		}
		uint32_t my[DP_ROUNDS], pos[DP_ROUNDS];
#pragma unroll
		for (int rd = 0; rd < DP_ROUNDS; ++rd) {
			my[rd] = 0;
#pragma unroll
			for (int i = 0; i < PER; ++i) my[rd] += (kk[i] != 0xFFFFu && kk[i] >= DP_ROUND_LO[rd] && kk[i] < DP_ROUND_LO[rd + 1]) ? 1u : 0u;
			pos[rd] = my[rd] ? atomicAdd(&cnt[rd], my[rd]) : 0u;
		}
		__syncthreads();
		if (threadIdx.x < DP_ROUNDS) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(dp.tickets + 8 + threadIdx.x, cnt[threadIdx.x]) : 0u;
		__syncthreads();
#pragma unroll
		for (int rd = 0; rd < DP_ROUNDS; ++rd) {
			uint32_t *out = dp.round_list + (size_t)rd * dp.jobs_cap + base[rd] + pos[rd];
#pragma unroll
			for (int i = 0; i < PER; ++i)
				if (kk[i] != 0xFFFFu && kk[i] >= DP_ROUND_LO[rd] && kk[i] < DP_ROUND_LO[rd + 1]) *out++ = j0 + (uint32_t)i;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------
// kernel C: the flank DPs of AlignHSP (alignhsp.cpp:98-162), one wavefront per DpJob
// ------------------------------------------------------------------------------------------------
// Everything here depends on the HSP and the sequences only: the windows, the two banded DPs, the trimming of
// terminal I runs, the all-gap floor, the path.  What depends on the search state -- whether the penalty cap still
// admits the HSP when its turn comes -- is left to finalize_se_kernel.  One conservative shortcut: if the penalty
// after the left flank already exceeds the cap the job was made under, the right flank is not run (the cap only
// falls, so the ordered replay returns at that same test).
#ifndef URX_DP_WAVES
#define URX_DP_WAVES 0  // build-time knob: register budget of dp_kernel as waves per SIMD (0 = the compiler's choice, 5)
#endif
#if URX_DP_WAVES
#define URX_DP_ATTR __attribute__((amdgpu_waves_per_eu(URX_DP_WAVES, URX_DP_WAVES)))
#else
#define URX_DP_ATTR
#endif
// URX_DP_PAIR (build-time, off): the round-3 prototype that runs the jobs two to a wavefront with every row >= 1 of
// both in packed int16 (viterbi_dev.h: VFlank, viterbi_pair_rows).  Bit-identical to the fp32 kernel on every parity test
// and bench check -- and slower: 6.8 instead of 4.1 ms per 1 M 150-base reads, 77 instead of 37 ms per 1 M 250-base reads
// (DESIGN.md section 3.4 has the instruction counts).  Kept for the A/B; tests/test_gpu_parity.py runs the packed rows
// through viterbi_batch_pair_kernel either way.
#ifdef URX_DP_PAIR
template <int NCH>
__global__ __launch_bounds__(64) URX_DP_ATTR void dp_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                const uint64_t *__restrict__ offs, DpWork dp, uint8_t *scratch,
                                                size_t scratch_stride, const uint8_t *__restrict__ g_seq, uint32_t klo, uint32_t khi,
                                                uint32_t *ticket, const uint32_t *__restrict__ list, const uint32_t *list_count) {
	constexpr int QMAX = 64 * NCH;
	constexpr int TB_ROWS8 = (QMAX - 24) / 8 + 2;
	// two jobs at a time (viterbi_dev.h: VFlank, viterbi_pair_rows): each has its query strand, target window, trace buffer
	// and run buffers; the windows sit behind the queries so that the bytes read around a window are LDS
	__shared__ __attribute__((aligned(16))) uint8_t sQ[2][QMAX];
	__shared__ uint8_t sT[2][QMAX + 64 + 32];
	__shared__ uint32_t tb[2][TB_ROWS8 * 64];
	__shared__ uint16_t ropsL[2][OPS_CAP], ropsR[2][OPS_CAP], cand[URMAPX_MAX_PATH_OPS];
	const int lane = threadIdx.x;
	WideScratch ws;
	ws.carve(scratch + (size_t)blockIdx.x * scratch_stride, QMAX, QMAX + 64);
	const VPar VP(P);
	const int BR = 2 * (int)P.band_radius;
	const uint32_t made = dp.counters[0];
	const uint32_t njobs = made < dp.jobs_cap ? made : dp.jobs_cap;
	auto load_window = [&](uint8_t *dst, uint32_t tlo, int tl) {  // true: the window holds a '-' pad byte
		bool gap = false;
		for (int i = lane; i < tl; i += 64) {
			const uint8_t c = g_seq[tlo + i];
			dst[i] = c;
			gap |= (c == '-');
		}
		URX_SYNC();
		return __ballot(gap) != 0;
	};
	uint32_t n_gated = 0;  // statistics
	// this round's jobs: those with klo <= k < khi.  Blocks take tiles of DP_TILE consecutive jobs from the round's work
	// counter (a read in a repeat family owns hundreds of consecutive jobs of the last round and none of the first: a
	// fixed tile-to-block map left blocks idle while others still had a dozen DPs to run); the k of a tile's jobs comes in
	// with one load and the block runs those of this round two at a time.
#ifndef URX_DP_TILE
#define URX_DP_TILE 32  // 8: 8.6 ms, 16: 5.7, 32: 5.3, 48: 5.6, 64: 5.9 per 1 M 150-base reads (the counter is one address for all blocks)
#endif
	constexpr uint32_t DP_TILE = URX_DP_TILE;
	for (;;) {
	uint32_t tile = 0;
	if (lane == 0) tile = atomicAdd(ticket, DP_TILE);
	tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
	if (tile >= njobs) break;
	uint32_t kk = 0xFFFFu;  // 0xFFFF: slot not in use
	if (lane < (int)DP_TILE && tile + lane < njobs) kk = dp.kidx[tile + lane];
	uint64_t todo = __ballot(kk >= klo && kk < khi && kk != 0xFFFFu);
	while (todo) {
		// ---- up to two jobs that pass AlignHSP's first test under the cap the replay has reached so far ----
		struct JobState {
			uint32_t j, startdb, pk, flags, vst_l, vst_r, combinedTLo;
			int maxpen, QL, startq, len, leftScore, rightScore, rtrim, totalPen, nL, nR;
			bool plus;
		} js[2];
		// (every index into js / F below is a compile-time constant after unrolling: a run-time index would put both
		// problems' state into scratch memory -- the first version of this loop ran five times slower for it)
		auto take_job = [&](JobState &S, uint8_t *sq) -> bool {
			while (todo) {
				const uint32_t j = tile + (uint32_t)__builtin_ctzll(todo);
				todo &= todo - 1;
				const DpJob J = dp.jobs[j];
				if (J.read == 0xFFFFFFFFu) continue;
				const int glen = (int)((J.pk >> PK_LEN_SH) & PK_MASK), gscore = (int)((J.pk >> PK_SCORE_SH) & PK_MASK);
				if (glen - gscore > J.maxpen) {  // the cap only falls
					if (lane == 0) dp.jobs[j].flags = DPJ_GATED;
					++n_gated;
					continue;
				}
				S.j = j; S.startdb = J.startdb; S.pk = J.pk; S.maxpen = J.maxpen;
				const uint64_t off = offs[J.read];
				S.QL = (int)(offs[J.read + 1] - off);
				S.startq = (int)(J.pk & PK_MASK); S.len = glen;
				S.plus = (J.pk >> PK_PLUS_SH) & 1u;
				S.flags = 0; S.vst_l = 0; S.vst_r = 0; S.combinedTLo = J.startdb;
				S.leftScore = 0; S.rightScore = 0; S.rtrim = 0; S.totalPen = glen - gscore; S.nL = 0; S.nR = 0;
				URX_SYNC();
				const uint8_t *q = bases + off;
#pragma unroll
				for (int c = 0; c < NCH; ++c) {
					const int p = 64 * c + lane;
					if (p < S.QL) sq[p] = S.plus ? q[p] : (uint8_t)comp_char(q[S.QL - 1 - p]);
				}
				return true;
			}
			return false;
		};
		bool have[2];
		have[0] = take_job(js[0], sQ[0]);
		have[1] = have[0] && take_job(js[1], sQ[1]);
		if (!have[0]) break;
		URX_SYNC();
		// ---- the left flanks of both jobs, then the right flanks (the right one's budget depends on the left one's outcome) ----
#pragma unroll 1
		for (int side = 0; side < 2; ++side) {
			const bool left = side == 0;
			VFlank F[2];
			bool run[2] = {false, false}, narrow[2] = {false, false};
			int fql[2] = {0, 0}, allGap[2] = {0, 0}, need[2] = {0, 0};
			uint32_t tlo[2] = {0, 0}, tl[2] = {0, 0};
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				F[h].active = false;
				if (!have[h]) continue;
				JobState &S = js[h];
				if (left) {
					if (S.startq <= 0) continue;
					fql[h] = S.startq;
					const uint32_t leftTHi = S.startdb - 1;
					tl[h] = (uint32_t)(fql[h] + BR);
					if (S.startdb < (uint32_t)S.startq || tl[h] >= leftTHi) { S.flags |= DPJ_LEFT_FAIL; continue; }
					tlo[h] = leftTHi - tl[h] + 1;
				} else {
					const int rightQLo = S.startq + S.len;
					if ((S.flags & DPJ_LEFT_FAIL) || rightQLo >= S.QL) continue;
					if (S.totalPen > S.maxpen) { S.flags |= DPJ_RIGHT_SKIPPED; continue; }
					fql[h] = S.QL - rightQLo;
					tlo[h] = S.startdb + (uint32_t)S.len;
					uint32_t thi = tlo[h] + (uint32_t)fql[h] + (uint32_t)BR;
					if (thi >= X.seqDataSize) thi = X.seqDataSize - 1;
					tl[h] = thi - tlo[h] + 1;
				}
				if (load_window(sT[h] + 32, tlo[h], (int)tl[h])) { S.flags |= left ? DPJ_LEFT_FAIL : DPJ_RIGHT_FAIL; continue; }
				// a flank score below `need` puts the penalty over the cap the job was made under (unless the all-gap floor
				// rescues it): the DP may stop as soon as that is certain
				allGap[h] = P.gap_open_score + (fql[h] - 1) * P.gap_ext_score;
				need[h] = fql[h] - (S.maxpen - S.totalPen);
				run[h] = true;
				narrow[h] = F[h].setup(VP, left ? sQ[h] : sQ[h] + (S.startq + S.len), fql[h], sT[h] + 32, (int)tl[h], left, !left, tb[h], TB_ROWS8,
				                       (float)need[h], allGap[h] < need[h]);
				if (!narrow[h]) F[h].active = false;
			}
			viterbi_pair_rows(F[0], F[1]);
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				if (!run[h]) continue;
				JobState &S = js[h];
				RevOps R;
				R.ops = left ? ropsL[h] : ropsR[h];
				uint32_t vst = 0;
				int score;
				bool aborted = false;
				if (narrow[h]) {
					score = (int)F[h].finish(R, vst);
					aborted = F[h].aborted;
				} else {  // cannot happen: the search kernel keeps reads with a clipped flank window to itself (park_for_dp)
					score = 0; R.begin(); vst = URMAPX_ST_BAND_TOO_WIDE;
				}
				if (aborted) { score = need[h] - 1; R.begin(); vst = 0; if (!left) S.flags |= DPJ_RIGHT_ABORTED; }
				if (left) {
					S.vst_l = vst;
					S.nL = R.n;
					// TrimLeftIs (pathinfo.cpp:153-171): the leading I run is the last run in traceback order
					int nTrimI = 0;
					if (S.nL > 0) {
						const uint32_t lastop = ropsL[h][S.nL - 1];
						if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --S.nL; }
					}
					S.combinedTLo = tlo[h] + (uint32_t)nTrimI;
					if (allGap[h] > score) score = allGap[h];
					S.leftScore = score;
					S.totalPen += fql[h] - score;
				} else {
					S.vst_r = vst;
					S.nR = R.n;
					// TrimRightIs (pathinfo.cpp:173-190): trailing I run = first run in traceback order, never the whole path
					if (S.nR > 1 && (ropsR[h][0] & 3u) == OP_I) S.rtrim = 1;
					if (allGap[h] > score) score = allGap[h];
					S.rightScore = score;
				}
			}
		}
		// ---- per job: path = Left || M x len || Right, run-length merged (uniform; lane 0 stores into LDS, then one coalesced copy) ----
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (!have[h]) continue;
			JobState &S = js[h];
			int nc = 0;
			if (!(S.flags & (DPJ_LEFT_FAIL | DPJ_RIGHT_FAIL | DPJ_RIGHT_SKIPPED | DPJ_RIGHT_ABORTED))) {
				int cop = -1, clen = 0;
				bool ovf = false;
				auto put = [&](int op, int l) {
					if (l <= 0) return;
					if (op == cop) { clen += l; return; }
					if (clen) { if (nc < URMAPX_MAX_PATH_OPS) { if (lane == 0) cand[nc] = (uint16_t)((clen << 2) | cop); ++nc; } else ovf = true; }
					cop = op; clen = l;
				};
				for (int t = S.nL - 1; t >= 0; --t) { const uint32_t o = ropsL[h][t]; put((int)(o & 3u), (int)(o >> 2)); }
				put(OP_M, S.len);
				for (int t = S.nR - 1; t >= S.rtrim; --t) { const uint32_t o = ropsR[h][t]; put((int)(o & 3u), (int)(o >> 2)); }
				put(-2, 1);  // flush
				if (ovf) { S.flags |= DPJ_PATH_LONG; nc = 0; }
				URX_SYNC();
				uint16_t *out = dp.ops + (size_t)S.j * DP_JOB_OPS;
				for (int t = lane; t < nc; t += 64) out[t] = cand[t];
				URX_SYNC();
			}
			if (lane == 0) {
				DpJob *o = dp.jobs + S.j;
				o->combined_tlo = S.combinedTLo;
				o->left_score = (int16_t)S.leftScore; o->right_score = (int16_t)S.rightScore;
				o->nops = (uint8_t)nc; o->flags = (uint8_t)S.flags; o->vst_l = (uint8_t)S.vst_l; o->vst_r = (uint8_t)S.vst_r;
			}
		}
	}
	}
	if (lane == 0 && n_gated) atomicAdd(dp.counters + 3, n_gated);
}
#else
// URX_DP_TB_GLOBAL (default 1): the trace cells of the narrow band live in the block's global scratch, not in LDS.  They were what held
// this kernel to 5.75 waves per SIMD for 150-base reads and to 4.25 for 250-base reads (9.2 KB of LDS per block); the stores are one
// dword per lane per eight rows and the traceback reads a few dozen cells.  250 bases: 31.5 -> 27.7 ms per 1 M reads, 150: 3.06 -> 2.94.
#ifndef URX_DP_TB_GLOBAL
#define URX_DP_TB_GLOBAL 1
#endif
template <int NCH>
__global__ __launch_bounds__(64) URX_DP_ATTR void dp_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                const uint64_t *__restrict__ offs, DpWork dp, uint8_t *scratch,
                                                size_t scratch_stride, const uint8_t *__restrict__ g_seq, uint32_t klo, uint32_t khi,
                                                uint32_t *ticket, const uint32_t *__restrict__ list, const uint32_t *list_count) {
	constexpr int QMAX = 64 * NCH;
	constexpr int TB_ROWS8 = (QMAX - 24) / 8 + 2;
	// ONE array for the read and the window behind it: the DP's row blocks read the window's bytes as base + immediate offset with
	// base = sT + (column of the block's first row), which is negative by up to band_radius + 1 for the lanes left of column 0 --
	// harmless as long as sT does not start at LDS offset 0: there the negative base wraps and base + offset is out of range for
	// bytes that ARE in the matrix (they read as 0).  Two separate arrays left the order to the compiler, and it happened to put sQ
	// first; adding any other LDS array to this kernel changed that and one tiny flank's score with it.
	__shared__ __attribute__((aligned(16))) uint8_t sQT[QMAX + QMAX + 64];
	uint8_t *const sQ = sQT, *const sT = sQT + QMAX;
	static_assert(QMAX >= 64, "the window lies band_radius + 1 bytes or more inside sQT (viterbi_dev.h: B_LDS)");
#if URX_DP_TB_GLOBAL  // the trace cells in the block's global scratch: LDS per block 7 -> 1.2 KB (+ the wide path's rows)
	uint32_t *const tb = reinterpret_cast<uint32_t *>(scratch + (size_t)blockIdx.x * scratch_stride +
	                                                  ((WideScratch::bytes(QMAX, QMAX + 64) + 255) & ~(size_t)255));
#ifdef URX_DP_NO_WLDS  // debugging aid: the wide path's rows in the global trace buffer too (what search_se_kernel's inline DP does)
#define URX_DP_WLDS
#else
	__shared__ uint32_t wlds[3 * QMAX];  // the wide path's three per-row arrays (it used the idle trace buffer when that was LDS)
#define URX_DP_WLDS , wlds, 3 * QMAX
#endif
#elif defined(URX_DP_WLDS_TEST)  // debugging aid: the trace buffer in LDS as shipped, the wide path's rows in an array of their own
	__shared__ uint32_t tb[TB_ROWS8 * 64];
	__shared__ uint32_t wlds[URX_DP_WLDS_TEST * QMAX];
#define URX_DP_WLDS , wlds, URX_DP_WLDS_TEST * QMAX
#else
	__shared__ uint32_t tb[TB_ROWS8 * 64];
#define URX_DP_WLDS
#endif
	__shared__ uint16_t ropsL[OPS_CAP], ropsR[OPS_CAP], cand[URMAPX_MAX_PATH_OPS];
	const int lane = threadIdx.x;
	WideScratch ws;
	ws.carve(scratch + (size_t)blockIdx.x * scratch_stride, QMAX, QMAX + 64);
	const VPar VP(P);
	const int BR = 2 * (int)P.band_radius;
	const uint32_t made = dp.counters[0];
	const uint32_t njobs = made < dp.jobs_cap ? made : dp.jobs_cap;
	auto load_window = [&](uint32_t tlo, int tl) {  // true: the window holds a '-' pad byte
		bool gap = false;
		for (int i = lane; i < tl; i += 64) {
			const uint8_t c = g_seq[tlo + i];
			sT[i] = c;
			gap |= (c == '-');
		}
		URX_SYNC();
		return __ballot(gap) != 0;
	};
	uint32_t n_gated = 0;  // statistics
	uint32_t q_read = 0xFFFFFFFFu;  // the read (and strand) whose bases sQ holds
	bool q_plus = false;
	uint64_t q_off = 0;
	int q_len = 0;
	// this round's jobs: the list dp_round_lists_kernel made of those with klo <= k < khi.  Blocks take tiles of DP_TILE list
	// entries from the round's work counter (a read in a repeat family owns hundreds of consecutive jobs of the last round
	// and none of the first: a fixed tile-to-block map left blocks idle while others still had a dozen DPs to run).
	// Until round 4 a tile was 32 consecutive jobs of the whole array, of which those of the round were picked by their k:
	// every round paid one ticket per 32 jobs MADE, and tickets are atomics on one address, which retire at 88 M/s on this
	// device whatever else the machine does -- 0.41 M tickets = 4.7 ms per round for 1 M 250-base reads, of which the first
	// two rounds had 7 % and 30 % of the jobs to run (7.5 ms each).
#ifndef URX_DP_TILE
#define URX_DP_TILE 16
#endif
#ifndef URX_DP_TB_GLOBAL
#define URX_DP_TB_GLOBAL 1
#endif
#ifndef URX_DP_EDGE2
#define URX_DP_EDGE2 false  // viterbi_dev.h: edge rows with the tests of their edge only -- bit-identical, 10 % fewer instructions per edge row and no faster: off
#endif
	constexpr uint32_t DP_TILE = URX_DP_TILE;
	(void)klo; (void)khi; (void)njobs;
	const uint32_t nlist = *list_count < dp.jobs_cap ? *list_count : dp.jobs_cap;
	// A block's first tile is its own (tile blockIdx.x), the ticket counter deals the tiles behind the grid's first sweep: a
	// launch with nothing to do -- the second pass's three, every batch -- used to cost one ticket per block, 5 888 atomics on one
	// address = 68 us each; a block that sees the list used up behind its tile does not ask again.
	uint32_t tile = blockIdx.x * DP_TILE;
	for (bool first = true;; first = false) {
	if (!first) {
		if (tile + DP_TILE >= nlist && tile >= gridDim.x * DP_TILE) break;  // this block took the list's last tile from the counter
		uint32_t t0 = 0;
		if (lane == 0) t0 = atomicAdd(ticket, DP_TILE);
		tile = gridDim.x * DP_TILE + (uint32_t)__builtin_amdgcn_readfirstlane((int)t0);
	}
	if (tile >= nlist) break;
	uint32_t jl = 0xFFFFFFFFu;
	if (lane < (int)DP_TILE && tile + lane < nlist) jl = list ? list[tile + lane] : tile + (uint32_t)lane;  // no list: every job made (phase 3's)
	uint64_t todo = __ballot(jl != 0xFFFFFFFFu);
	while (todo) {
		const uint32_t j = rdlane(jl, __builtin_ctzll(todo));
		todo &= todo - 1;
		const DpJob J = dp.jobs[j];
		if (J.read == 0xFFFFFFFFu) continue;
		{  // AlignHSP's first test under the cap the replay has reached so far (the cap only falls)
			const int glen = (int)((J.pk >> PK_LEN_SH) & PK_MASK), gscore = (int)((J.pk >> PK_SCORE_SH) & PK_MASK);
			if (glen - gscore > J.maxpen) {
				if (lane == 0) dp.jobs[j].flags = DPJ_GATED;
				++n_gated;
				continue;
			}
		}
		const uint32_t pk = J.pk, startdb = J.startdb;
		const int startq = (int)(pk & PK_MASK), len = (int)((pk >> PK_LEN_SH) & PK_MASK), hscore = (int)((pk >> PK_SCORE_SH) & PK_MASK);
		const bool plus = (pk >> PK_PLUS_SH) & 1u;
		// the jobs of a tile are consecutive HSPs, mostly of one read: its bases (one strand) stay in LDS from job to job
		URX_SYNC();
		if (J.read != q_read || plus != q_plus) {
			if (J.read != q_read) {
				q_off = offs[J.read];
				q_len = (int)(offs[J.read + 1] - q_off);
			}
			const uint8_t *q = bases + q_off;
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int p = 64 * c + lane;
				if (p < q_len) sQ[p] = plus ? q[p] : (uint8_t)comp_char(q[q_len - 1 - p]);
			}
			q_read = J.read; q_plus = plus;
			URX_SYNC();
		}
		const int QL = q_len;
		uint32_t flags = 0, vst_l = 0, vst_r = 0, combinedTLo = startdb;
		int leftScore = 0, rightScore = 0, rtrim = 0;
		int totalPen = len - hscore;
		RevOps RL, RR;
		RL.ops = ropsL; RR.ops = ropsR;
		RL.begin(); RR.begin();
#if defined(URX_DP_WLDS_TEST) && defined(URX_DP_WLDS_FILL)
		for (int t = lane; t < URX_DP_WLDS_TEST * QMAX; t += 64) wlds[t] = URX_DP_WLDS_FILL;
		URX_SYNC();
#endif
		if (startq > 0) {
			const int leftQL = startq;
			const uint32_t leftTHi = startdb - 1;
			const uint32_t leftTL = (uint32_t)(leftQL + BR);
			if (startdb < (uint32_t)startq || leftTL >= leftTHi) flags |= DPJ_LEFT_FAIL;
			else {
				const uint32_t leftTLo = leftTHi - leftTL + 1;
				if (load_window(leftTLo, (int)leftTL)) flags |= DPJ_LEFT_FAIL;
				else {
					// a flank score below `need` puts the penalty over the cap the job was made under (unless the all-gap floor
					// rescues it): the DP may stop as soon as that is certain
					const int allGap = P.gap_open_score + (leftQL - 1) * P.gap_ext_score;
					const int need = leftQL - (J.maxpen - totalPen);
					bool aborted = false;
					leftScore = (int)viterbi_wave<true, URX_DP_EDGE2>(VP, sQ, leftQL, sT, (int)leftTL, true, false, tb, TB_ROWS8, ws, RL, vst_l, lane,
					                                    (float)need, allGap < need ? &aborted : nullptr URX_DP_WLDS);
					if (aborted) { leftScore = need - 1; RL.begin(); vst_l = 0; }
					// TrimLeftIs (pathinfo.cpp:153-171): the leading I run is the last run in traceback order
					int nTrimI = 0;
					if (RL.n > 0) {
						const uint32_t lastop = ropsL[RL.n - 1];
						if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --RL.n; }
					}
					combinedTLo = leftTLo + (uint32_t)nTrimI;
					if (allGap > leftScore) leftScore = allGap;
					totalPen += leftQL - leftScore;
				}
			}
		}
		const int rightQLo = startq + len;
		if (!(flags & DPJ_LEFT_FAIL) && rightQLo < QL) {
			if (totalPen > J.maxpen) flags |= DPJ_RIGHT_SKIPPED;
			else {
				const int rightQL = QL - rightQLo;
				const uint32_t rightTLo = startdb + (uint32_t)len;
				uint32_t rightTHi = rightTLo + (uint32_t)rightQL + (uint32_t)BR;
				if (rightTHi >= X.seqDataSize) rightTHi = X.seqDataSize - 1;
				const uint32_t rightTL = rightTHi - rightTLo + 1;
				if (load_window(rightTLo, (int)rightTL)) flags |= DPJ_RIGHT_FAIL;
				else {
					const int allGap = P.gap_open_score + (rightQL - 1) * P.gap_ext_score;
					const int need = rightQL - (J.maxpen - totalPen);
					bool aborted = false;
					rightScore = (int)viterbi_wave<true, URX_DP_EDGE2>(VP, sQ + rightQLo, rightQL, sT, (int)rightTL, false, true, tb, TB_ROWS8, ws, RR, vst_r, lane,
					                                     (float)need, allGap < need ? &aborted : nullptr URX_DP_WLDS);
					if (aborted) { rightScore = need - 1; RR.begin(); vst_r = 0; flags |= DPJ_RIGHT_ABORTED; }
					// TrimRightIs (pathinfo.cpp:173-190): trailing I run = first run in traceback order, never the whole path
					if (RR.n > 1 && (ropsR[0] & 3u) == OP_I) rtrim = 1;
					if (allGap > rightScore) rightScore = allGap;
				}
			}
		}
		// path = Left || M x len || Right, run-length merged (uniform; lane 0 stores into LDS, then one coalesced copy)
		int nc = 0;
		if (!(flags & (DPJ_LEFT_FAIL | DPJ_RIGHT_FAIL | DPJ_RIGHT_SKIPPED | DPJ_RIGHT_ABORTED))) {
			int cop = -1, clen = 0;
			bool ovf = false;
			auto put = [&](int op, int l) {
				if (l <= 0) return;
				if (op == cop) { clen += l; return; }
				if (clen) { if (nc < URMAPX_MAX_PATH_OPS) { if (lane == 0) cand[nc] = (uint16_t)((clen << 2) | cop); ++nc; } else ovf = true; }
				cop = op; clen = l;
			};
			for (int t = RL.n - 1; t >= 0; --t) { const uint32_t o = ropsL[t]; put((int)(o & 3u), (int)(o >> 2)); }
			put(OP_M, len);
			for (int t = RR.n - 1; t >= rtrim; --t) { const uint32_t o = ropsR[t]; put((int)(o & 3u), (int)(o >> 2)); }
			put(-2, 1);  // flush
			if (ovf) { flags |= DPJ_PATH_LONG; nc = 0; }
			URX_SYNC();
			uint16_t *out = dp.ops + (size_t)j * DP_JOB_OPS;
			for (int t = lane; t < nc; t += 64) out[t] = cand[t];
		}
		if (lane == 0) {
			DpJob *o = dp.jobs + j;
			o->combined_tlo = combinedTLo;
			o->left_score = (int16_t)leftScore; o->right_score = (int16_t)rightScore;
			o->nops = (uint8_t)nc; o->flags = (uint8_t)flags; o->vst_l = (uint8_t)vst_l; o->vst_r = (uint8_t)vst_r;
		}
	}
	}
	if (lane == 0 && n_gated) atomicAdd(dp.counters + 3, n_gated);
}
#endif

// ------------------------------------------------------------------------------------------------
// kernel D: phase 6's ordered part for the parked reads -- the jobs of a read in HSP order through AlignHSP's tests
// and AddHitX, then CalcMAPQ6 / SetMappedPos and the result record.  One wavefront per read.
// ------------------------------------------------------------------------------------------------
#ifndef URX_FIN_WAVES
#define URX_FIN_WAVES 8  // register budget of finalize_se_kernel as waves per SIMD: 51 VGPRs, no spill; the kernel waits on memory and wants waves (0 = the compiler's choice: 85 VGPRs, 5 waves)
#endif
#if URX_FIN_WAVES
#define URX_FIN_ATTR __attribute__((amdgpu_waves_per_eu(URX_FIN_WAVES, URX_FIN_WAVES)))
#else
#define URX_FIN_ATTR
#endif
template <int NCH, bool OVF>
__global__ __launch_bounds__(64) URX_FIN_ATTR void finalize_se_kernel(DevIndex X, urmapx_params P, const uint64_t *__restrict__ offs, DpWork dp,
                                                         urmapx_result *__restrict__ results, urmapx_path_op *__restrict__ path_ops,
                                                         uint32_t *path_used, int hsp_lds_cap, uint32_t *ovf_list, uint32_t klo,
                                                         uint32_t khi) {
	using SW = SearchWave<NCH, OVF>;
	__shared__ uint16_t top[URMAPX_MAX_PATH_OPS], cand[URMAPX_MAX_PATH_OPS];
	// The paths of the reads a block finishes wait in LDS and go to the arena many at a time: room in the arena is an atomic
	// on ONE address, those retire at 88 M/s on this device whatever else it does, and one per finished read was what the
	// finalize launches took once the statistics counter was out of the way (0.3 M reads done in the first round of 1 M
	// 250-base reads: 3.2 ms).  The arena stays dense; the result records of those reads wait with their paths and are written at the flush.
	constexpr int PBUF_OPS = 1024, PEND = 64;
	__shared__ uint16_t pbuf[PBUF_OPS];
	__shared__ uint32_t pend_r[PEND];
	__shared__ urmapx_result pend_res[PEND];  // path_off = the offset inside pbuf until the flush
	int pb_used = 0, pb_n = 0;
	const int lane = threadIdx.x;
	auto flush_paths = [&]() {
		if (pb_n == 0) return;
		URX_SYNC();
		uint32_t po = 0;
		if (lane == 0) po = atomicAdd(path_used, (uint32_t)pb_used);
		po = uni(po);
		for (int t = lane; t < pb_used; t += 64) path_ops[po + t] = pbuf[t];
		if (lane < pb_n) {
			urmapx_result rr = pend_res[lane];
			rr.path_off += po;
			results[pend_r[lane]] = rr;
		}
		URX_SYNC();
		pb_used = 0; pb_n = 0;
	};
	SW S(X, P, lane);
	S.W = (int)X.W;
	S.top = top; S.cand = cand;
	S.hsp_lds = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? (hsp_lds_cap & ~63) : HSP_CAP;
	S.hit_cap = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? S.hsp_lds / 4 : 64 * SE_HITW1;
	S.hit_wsh = (hsp_lds_cap >= 64 && hsp_lds_cap <= HSP_CAP) ? 4 : 6;
	const uint32_t parked = dp.counters[1] < dp.fin_cap ? dp.counters[1] : dp.fin_cap;
	// parked reads go to blocks round-robin: the costly ones (repeat families: hundreds of jobs) were parked last, next
	// to each other, and a block that took a run of them from a work counter made the launch 0.8 ms longer
	// A read-round is a chain of memory round trips and little else (SQ_WAIT_ANY 97 % of the wave cycles), so
	// the chain is kept short: the list entry (read, first job, job count, read length: one 16-byte load) of the block's NEXT
	// read is asked for before this one is replayed, and the parked state, the top path and the first 64 jobs of a read
	// leave in one round of loads.
	const uint4 *const fin = reinterpret_cast<const uint4 *>(dp.fin_list);
	uint32_t used_block = 0;
	uint4 ent = make_uint4(0u, 0u, 0u, 0u);
	if (blockIdx.x < parked) ent = fin[blockIdx.x];
	for (uint32_t e = blockIdx.x; e < parked; e += gridDim.x) {
		const uint32_t r = uni(ent.x), jb = uni(ent.y), nj = uni(ent.z);
		S.QL = (int)uni(ent.w);
		if (e + gridDim.x < parked) ent = fin[e + gridDim.x];
		if (nj <= klo) continue;  // finished in an earlier round
		S.nwords = S.QL - (S.W - 1);
		const uint32_t kend = nj < khi ? nj : khi;
		uint32_t jpk0 = 0;
		uint4 jw0 = make_uint4(0u, 0u, 0u, 0u);
		if (klo + (uint32_t)lane < kend) {
			const uint32_t *jp = reinterpret_cast<const uint32_t *>(dp.jobs + jb + klo + (uint32_t)lane);
			jpk0 = jp[2];
			jw0 = *reinterpret_cast<const uint4 *>(jp + 4);
		}
		URX_SYNC();
		uint32_t *const st = dp.state + (size_t)e * SW::STATE_WORDS;
		const int phase = S.restore_state(st);
		uint32_t used = 0;
		// 64 jobs come in with one round of loads (one job per lane: the words the replay reads).  Round 4: every lane then runs
		// AlignHSP's tests on its own job against the state as it stands -- the penalty after the HSP, after the left flank,
		// after the right flank against the cap (alignhsp.cpp:62-70,127-130,160-162), the score against best - 12
		// (state1.cpp:530-536).  The cap only falls and the best score only rises, so a job that fails one of them now fails
		// it at its turn too and changes nothing: only the jobs that pass go through the ordered replay (consume_job, which
		// repeats the tests at the job's turn), and the set is filtered again after every change.  A read in a repeat family
		// brings hundreds of jobs of which a handful end in a hit; replaying them one by one made this launch 9.7 of 105 ms per
		// 1 M 250-base reads.  (Jobs that raise a status bit -- a DP that outgrew its buffers -- always take the ordered road.)
		for (uint32_t k0 = klo; k0 < kend; k0 += 64) {
			const uint32_t k = k0 + (uint32_t)lane;
			const uint32_t jpk = jpk0;
			const uint4 jw = jw0;
			// the next 64 jobs are asked for before these are replayed (a read in a repeat family brings thousands: one wave, one
			// round trip per 64 of them otherwise)
			jpk0 = 0; jw0 = make_uint4(0u, 0u, 0u, 0u);
			if (k + 64u < kend) {
				const uint32_t *jp = reinterpret_cast<const uint32_t *>(dp.jobs + jb + k + 64u);
				jpk0 = jp[2];
				jw0 = *reinterpret_cast<const uint4 *>(jp + 4);
			}
			const int jstartq = (int)(jpk & PK_MASK), jlen = (int)((jpk >> PK_LEN_SH) & PK_MASK), jhs = (int)((jpk >> PK_SCORE_SH) & PK_MASK);
			const int jpen = jlen - jhs;
			const int jls = (int)(int16_t)(jw.y & 0xFFFFu), jrs = (int)(int16_t)(jw.y >> 16);
			const uint32_t jfl = (jw.z >> 8) & 0xFFu;
			const bool has_l = jstartq > 0, has_r = jstartq + jlen < S.QL;
			const bool raises = (jw.z >> 16) != 0u || (jfl & (DPJ_GATED | DPJ_PATH_LONG)) != 0u;  // vst_l / vst_r / flags that set a status bit
			const bool dead = (has_l && (jfl & DPJ_LEFT_FAIL)) || (has_r && (jfl & (DPJ_RIGHT_FAIL | DPJ_RIGHT_SKIPPED)));
			const int pen1 = jpen + (has_l ? jstartq - jls : 0);
			const int pen2 = pen1 + (has_r ? (S.QL - (jstartq + jlen)) - jrs : 0);
			const int jscore = jhs + (has_l ? jls : 0) + (has_r ? jrs : 0);
			auto viable = [&]() {
				return k < kend && jpen <= S.maxPen && (raises || (!dead && pen1 <= S.maxPen && pen2 <= S.maxPen && jscore >= S.best - SECONDARY_HIT_MAX_DELTA && jscore >= 10));
			};
			used += (uint32_t)__builtin_popcountll(__ballot(k < kend && jpen <= S.maxPen));
			uint64_t todo = __ballot(viable());
			while (todo) {
				const int t = __builtin_ctzll(todo);
				todo &= todo - 1;
				DpJob J;
				J.read = r; J.startdb = 0; J.maxpen = 0; J.k = 0; J.pad[0] = J.pad[1] = 0;
				J.pk = rdlane(jpk, t);
				J.combined_tlo = rdlane(jw.x, t);
				const uint32_t sc = rdlane(jw.y, t), fl = rdlane(jw.z, t);
				J.left_score = (int16_t)(sc & 0xFFFFu); J.right_score = (int16_t)(sc >> 16);
				J.nops = (uint8_t)(fl & 0xFFu); J.flags = (uint8_t)((fl >> 8) & 0xFFu); J.vst_l = (uint8_t)((fl >> 16) & 0xFFu); J.vst_r = (uint8_t)(fl >> 24);
				const int mp0 = S.maxPen, b0 = S.best;
				S.consume_job(J, dp.ops + (size_t)(jb + k0 + (uint32_t)t) * DP_JOB_OPS);
				if (S.maxPen != mp0 || S.best != b0) todo &= __ballot(viable());
			}
		}
		used_block += used;  // statistics: jobs whose DP the ordered replay looked at
		if (nj > khi) {  // more rounds to come: park again, and tell the remaining jobs the cap reached so far
			URX_SYNC();
			S.park_state(st, phase);
			for (uint32_t k = khi + lane; k < nj; k += 64) dp.jobs[jb + k].maxpen = S.maxPen;
			continue;
		}
		urmapx_result res;
		res.dbpos = 0xFFFFFFFFu; res.seq_index = 0xFFFFFFFFu; res.coord = 0xFFFFFFFFu;
		res.score = 0; res.second = 0; res.mapq = 0; res.plus = 0; res.exit_phase = 0; res.status = 0;
		res.hit_count = 0; res.path_nops = 0; res.path_off = 0;
		if (S.fill_result_core(res, phase)) {
			if (pb_used + S.top_nops > PBUF_OPS || pb_n == PEND) flush_paths();
			URX_SYNC();
			for (int t = lane; t < S.top_nops; t += 64) pbuf[pb_used + t] = top[t];
			res.path_nops = (uint16_t)S.top_nops;
			res.path_off = (uint32_t)pb_used;  // inside pbuf; the flush adds the place of the buffer in the arena
			if (lane == 0) { pend_r[pb_n] = r; pend_res[pb_n] = res; }
			pb_used += S.top_nops; ++pb_n;
		} else if (lane == 0) results[r] = res;
		if constexpr (!OVF) {
			if (res.status & (URMAPX_ST_HSP_OVERFLOW | URMAPX_ST_HIT_OVERFLOW)) {  // the hit list outgrew the first pass's: map again
				if (lane == 0) ovf_list[1 + atomicAdd(ovf_list, 1u)] = r;
			}
		}
	}
	flush_paths();
	// one atomic per block, not per read: atomics on one address retire at 88 M/s on this device, and one per parked read was
	// what the first finalize launch of a batch took (0.5 M reads: 5.7 ms)
	if (lane == 0 && used_block) atomicAdd(dp.counters + 2, used_block);
}

size_t dp_state_words(bool ovf) { return ovf ? (size_t)SearchWave<3, true>::STATE_WORDS : (size_t)SearchWave<3, false>::STATE_WORDS; }
// words of a read parked at phase 3 (0: this read-length class keeps phase 3 inline)
size_t p3_state_words(uint32_t max_read_len) {
	return max_read_len <= 128 ? (size_t)SearchWave<2, false>::P3_WORDS : max_read_len <= 192 ? (size_t)SearchWave<3, false>::P3_WORDS :
	       max_read_len <= 256 ? (size_t)SearchWave<4, false>::P3_WORDS : max_read_len <= 320 ? (size_t)SearchWave<5, false>::P3_WORDS : 0;
}
size_t dp_scratch_stride(uint32_t max_read_len) {
	const int qmax = 64 * (max_read_len <= 128 ? 2 : max_read_len <= 192 ? 3 : max_read_len <= 256 ? 4 : max_read_len <= 320 ? 5 : max_read_len <= 512 ? 8 : 16);
	// the wide-band rows and trace bytes, then room for the narrow band's trace cells (URX_DP_TB_GLOBAL: dp_kernel keeps them here
	// instead of in LDS)
	return ((WideScratch::bytes(qmax, qmax + 64) + 255) & ~(size_t)255) + (size_t)((qmax - 24) / 8 + 2) * 256;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int nch_for(uint32_t max_read_len) {
	if (max_read_len <= 128) return 2;  // 100 bp reads: two mask words, 8 window loads
	if (max_read_len <= 192) return 3;
	if (max_read_len <= 256) return 4;  // 250 bp reads: smaller per-read state than the 320-base class, one more wave per SIMD
	if (max_read_len <= 320) return 5;
	if (max_read_len <= 512) return 8;  // 1 wave per SIMD: eight mask words and 32 window loads per lane
	if (max_read_len <= 1024) return 16;
	return 0;
}

size_t search_scratch_stride(uint32_t max_read_len) {
	const int nch = nch_for(max_read_len);
	return (search_scratch_bytes(nch) + 255) & ~(size_t)255;
}
// behind the strided per-block areas: the HSP overflow lists of the second pass's blocks
size_t search_scratch_tail(int blocks) {
	return (size_t)(blocks > SEARCH_OVF_BLOCKS ? blocks : SEARCH_OVF_BLOCKS) * (HSP_TOTAL_CAP - HSP_CAP) * sizeof(uint2);
}

int search_block_count(uint32_t max_read_len, int device) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
	int per_cu = 0;
	const int nchq = nch_for(max_read_len);
	hipError_t e = nchq == 2   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<2, false, false, true>, 64, 0)
	               : nchq == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<3, false, false, true>, 64, 0)
	               : nchq == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<4, false, false, true>, 64, 0)
	               : nchq == 5 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<5, false, false, true>, 64, 0)
	               : nchq == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<8, false, false>, 64, 0)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_se_kernel<16, false, false>, 64, 0);
	if (e != hipSuccess || per_cu < 1) per_cu = 8;
	// measurement aid: fewer resident waves with the same code (is the kernel bound by issue or by latency? DESIGN.md 5.0)
	if (const char *t = getenv("URMAPX_TEST_BLOCKS_PER_CU")) { const int v = atoi(t); if (v >= 1 && v < per_cu) per_cu = v; }
	// tuning knob (round 6): the persistent search kernel leaves room on every CU -- for the launches of ANOTHER context of the device (a second
	// lane of urmapx_map_files), which otherwise wait until this kernel's blocks exit (profiles/r6/lanes_blocks.txt)
	if (const char *t = getenv("URMAPX_BLOCKS_PER_CU")) { const int v = atoi(t); if (v >= 1 && v < per_cu) per_cu = v; }
	return per_cu * prop.multiProcessorCount;
}

// ------------------------------------------------------------------------------------------------
// the packed copy of the sequence store (dev_common.h: seq_code, lane_mismatch_planes), built once per index on the device
// ------------------------------------------------------------------------------------------------
// One wavefront per 64 bases at a time: lane = base, four ballots = the four planes of two 32-base blocks.  Bytes
// beyond the stored sequence (the reference reads past its end; the store is zero-padded) get the
// "other byte" code, like any byte outside the code list.
__global__ __launch_bounds__(256) void pack_seq_kernel(const uint8_t *__restrict__ seq, uint64_t nbytes, uint4 *__restrict__ out,
                                                       uint64_t nblocks) {
	const int lane = threadIdx.x & 63;
	const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	uint32_t *o32 = reinterpret_cast<uint32_t *>(out);
	for (uint64_t w = wave; 2 * w < nblocks; w += nwaves) {
		const uint64_t i = 64 * w + lane;
		const uint32_t code = i < nbytes ? seq_code(seq[i], SEQ_CODE_TOTHER) : SEQ_CODE_TOTHER;
		const uint64_t b0 = __ballot(code & 1u), b1 = __ballot(code & 2u), b2 = __ballot(code & 4u), b3 = __ballot(code & 8u);
		if (lane < 8 && 2 * w + (lane >> 2) < nblocks) {  // lanes 0..3: block 2w's planes, 4..7: block 2w+1's
			const int k = lane & 3;
			const uint64_t b = k == 0 ? b0 : k == 1 ? b1 : k == 2 ? b2 : b3;
			o32[8 * w + lane] = (uint32_t)(b >> (32 * (lane >> 2)));
		}
	}
}

// blocks of 32 bases: the stored bytes, the 4096-byte zero pad behind them, and the blocks a window load runs past its last base
size_t packed_seq_blocks(uint32_t seq_data_size) { return ((size_t)seq_data_size + 4096 + 31) / 32 + 40; }

hipError_t launch_pack_seq(const uint8_t *d_seq, uint32_t seq_data_size, uint4 *d_out, hipStream_t s) {
	const uint64_t nblocks = packed_seq_blocks(seq_data_size);
	hipLaunchKernelGGL(pack_seq_kernel, dim3(8192), dim3(256), 0, s, d_seq, (uint64_t)seq_data_size, d_out, nblocks);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// measurement aid: random 5-byte slot gathers over the resident slot table, nothing else -- the ceiling any probe of
// this table can reach on this device (one 64-byte sector per slot, 6 % of the slots straddle two)
// ------------------------------------------------------------------------------------------------
// MODE 0: one 8-byte load per slot (what load_slot does); 1: two 4-byte loads per slot; 2: two 4-byte LDS-DMA loads per
// slot (what the search kernel's probe_gather does); 3: one 12-byte LDS-DMA load per slot; 4: one 16-byte load per slot.
// The slot rate of each mode says whether random access is priced per sector or per request.
template <int MODE>
__global__ __launch_bounds__(256) void gather_bench_kernel(const uint8_t *__restrict__ blob, uint64_t slot_count,
                                                           uint64_t magic, uint32_t iters, uint32_t *sink) {
	__shared__ uint32_t dst[MODE == 2 ? 4 * 8 * 2 * 64 : (MODE == 3 ? 4 * 8 * 3 * 64 : 1)];
	uint64_t x = murmur64(((uint64_t)blockIdx.x << 20) + threadIdx.x + 1);
	uint32_t acc = 0;
	const int wv = threadIdx.x >> 6;
	for (uint32_t i = 0; i < iters; ++i) {
		if constexpr (MODE == 0) {
			uint32_t t[8], p[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				x = murmur64(x + 0x9E3779B97F4A7C15ull);
				load_slot(blob, mod_slots(x, slot_count, magic), t[u], p[u]);
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += t[u] ^ p[u];
		} else if constexpr (MODE == 1) {
			uint32_t lo[8], hi[8];
			const uint32_t *a[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				x = murmur64(x + 0x9E3779B97F4A7C15ull);
				a[u] = reinterpret_cast<const uint32_t *>(blob + ((5ull * mod_slots(x, slot_count, magic)) & ~3ull));
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:4" : "=&v"(lo[u]), "=&v"(hi[u]) : "v"(a[u]) : "memory");
			}
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
			             "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7])::"memory");
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += lo[u] ^ hi[u];
		} else if constexpr (MODE == 2 || MODE == 3) {
			constexpr int PER = MODE == 2 ? 2 : 3;
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				x = murmur64(x + 0x9E3779B97F4A7C15ull);
				const uint8_t *a = blob + ((5ull * mod_slots(x, slot_count, magic)) & ~3ull);
				if constexpr (MODE == 2) {
					glds_dword(a, lds_addr(dst + ((wv * 8 + u) * PER + 0) * 64));
					glds_dword(a + 4, lds_addr(dst + ((wv * 8 + u) * PER + 1) * 64));
				} else {
					unsigned keep;
					asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
					             : "=&s"(keep) : "v"(a), "s"(lds_addr(dst + (wv * 8 + u) * PER * 64)) : "memory");
				}
			}
			wait_vm0();
			acc += dst[(wv * 8 * PER) * 64 + (threadIdx.x & 63) + (i & 7) * 64];
		} else {
			uint4 v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				x = murmur64(x + 0x9E3779B97F4A7C15ull);
				v[u] = *reinterpret_cast<const uint4 *>(blob + ((5ull * mod_slots(x, slot_count, magic)) & ~15ull));
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].w;
		}
	}
	if (acc == 0x12345678u) *sink = acc;  // never true in practice; keeps the loads alive
}

hipError_t launch_gather_bench(const DevIndex &X, uint32_t blocks, uint32_t iters, uint32_t *d_sink, hipStream_t s) {
	const char *e = getenv("URMAPX_GATHER_MODE");  // measurement aid only
	const int mode = e ? atoi(e) : 0;
	if (mode == 1) hipLaunchKernelGGL(gather_bench_kernel<1>, dim3(blocks), dim3(256), 0, s, X.blob, X.slotCount, X.slotMagic, iters, d_sink);
	else if (mode == 2) hipLaunchKernelGGL(gather_bench_kernel<2>, dim3(blocks), dim3(256), 0, s, X.blob, X.slotCount, X.slotMagic, iters, d_sink);
	else if (mode == 3) hipLaunchKernelGGL(gather_bench_kernel<3>, dim3(blocks), dim3(256), 0, s, X.blob, X.slotCount, X.slotMagic, iters, d_sink);
	else if (mode == 4) hipLaunchKernelGGL(gather_bench_kernel<4>, dim3(blocks), dim3(256), 0, s, X.blob, X.slotCount, X.slotMagic, iters, d_sink);
	else hipLaunchKernelGGL(gather_bench_kernel<0>, dim3(blocks), dim3(256), 0, s, X.blob, X.slotCount, X.slotMagic, iters, d_sink);
	return hipGetLastError();
}

size_t viterbi_batch_scratch_stride() { return (WideScratch::bytes(VB_WIDE_LA, VB_WIDE_LB) + 255) & ~(size_t)255; }

hipError_t launch_seed_probe(const DevIndex &X, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t n,
                             uint32_t max_read_len, ProbeOut out, hipStream_t s) {
	if (n == 0) return hipSuccess;
	const int nch = nch_for(max_read_len);
	dim3 block(256), grid((n + 3) / 4);
	if (nch == 2) hipLaunchKernelGGL(seed_probe_kernel<2>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else if (nch == 3) hipLaunchKernelGGL(seed_probe_kernel<3>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else if (nch == 4) hipLaunchKernelGGL(seed_probe_kernel<4>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else if (nch == 5) hipLaunchKernelGGL(seed_probe_kernel<5>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else if (nch == 8) hipLaunchKernelGGL(seed_probe_kernel<8>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else hipLaunchKernelGGL(seed_probe_kernel<16>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	return hipGetLastError();
}

hipError_t launch_search_se(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t n, uint32_t max_read_len, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, hipStream_t s) {
	if (n == 0) return hipSuccess;
	const int nch = nch_for(max_read_len);
	{
		hipError_t e = hipMemsetAsync(wk.ticket, 0, 8, s);  // two words: the first pass's ticket counter and the second pass's
		if (e == hipSuccess && wk.ticket3) e = hipMemsetAsync(wk.ticket3, 0, 4, s);  // the work counter of the launch over the reads parked at phase 3
		if (e != hipSuccess) return e;
	}
	if (wk.stats) {
		hipError_t e = hipMemsetAsync(wk.stats + 2, 0, 184, s);
		if (e != hipSuccess) return e;
	}
	{
		hipError_t e = hipMemsetAsync(wk.ovf_list, 0, 4, s);
		if (e != hipSuccess) return e;
	}
	dim3 block(64), grid((unsigned)wk.blocks);
	uint2 *const no_ovf = reinterpret_cast<uint2 *>(wk.scratch + (size_t)wk.blocks * wk.scratch_stride);  // HSP lists beyond LDS
	const DpWork no_dp;
	auto stamp = [&](int i) { if (wk.stage_events) (void)hipEventRecord(wk.stage_events[i], s); };
	if (wk.dp[0].jobs && wk.dp[1].jobs && wk.dp[1].counters == wk.dp[0].counters + 4 && wk.dp[0].tickets == wk.dp[0].counters + 16 &&
	    wk.dp[1].tickets == wk.dp[0].tickets + DP_TICKET_WORDS) {
		// both passes' counters and work counters (and phase 3's behind them) are one block of the work buffer's head (urmapx.hip): one fill
		const bool with3 = wk.dp3.jobs && wk.dp3.counters == wk.dp[0].counters + 8 && wk.dp3.tickets == wk.dp[0].tickets + 2 * DP_TICKET_WORDS;
		hipError_t e = hipMemsetAsync(wk.dp[0].counters, 0, 64 + (with3 ? 3 : 2) * 4 * DP_TICKET_WORDS, s);
		if (e == hipSuccess && wk.dp3.jobs && !with3) {
			e = hipMemsetAsync(wk.dp3.counters, 0, 16, s);
			if (e == hipSuccess) e = hipMemsetAsync(wk.dp3.tickets, 0, 4 * DP_TICKET_WORDS, s);
		}
		if (e != hipSuccess) return e;
	} else
		for (int pass = 0; pass < 2; ++pass)
			if (wk.dp[pass].jobs) {
				hipError_t e = hipMemsetAsync(wk.dp[pass].counters, 0, 16, s);
				if (e == hipSuccess) e = hipMemsetAsync(wk.dp[pass].tickets, 0, 4 * DP_TICKET_WORDS, s);
				if (e != hipSuccess) return e;
			}
#define URX_LAUNCH_SE(NCH_, OVF_, DBG_, GRID_, STATS_, OVFBASE_, DP_)                                                             \
	hipLaunchKernelGGL((search_se_kernel<NCH_, OVF_, DBG_>), GRID_, block, 0, s, X, P, d_bases, d_offs, n, d_results,                \
	                   d_path_ops, d_path_used, STATS_, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp,                    \
	                   wk.ticket + ((OVF_) ? 1 : 0), wk.hsp_lds_cap, wk.ovf_list, OVFBASE_, DP_, no_dp)
#define URX_LAUNCH_SE_ROWS(NCH_, GRID_, STATS_, OVFBASE_, DP_)                                                                      \
	hipLaunchKernelGGL((search_se_kernel<NCH_, false, false, true>), GRID_, block, 0, s, X, P, d_bases, d_offs, n, d_results,          \
	                   d_path_ops, d_path_used, STATS_, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, wk.ticket,         \
	                   wk.hsp_lds_cap, wk.ovf_list, OVFBASE_, DP_, no_dp)
#define URX_LAUNCH_SE_S16(NCH_, GRID_, OVFBASE_, DP_)                                                                               \
	hipLaunchKernelGGL((search_se_kernel<NCH_, false, false, 2>), GRID_, block, 0, s, X, P, d_bases, d_offs, n, d_results,            \
	                   d_path_ops, d_path_used, no_stats3, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, wk.ticket,          \
	                   wk.hsp_lds_cap, wk.ovf_list, OVFBASE_, DP_, no_dp)
	// the same with fewer k-mer chunks than byte chunks (SearchWave: KCH) and the row store in LDS
#define URX_LAUNCH_SE_S16K(NCH_, KCH_, GRID_, OVFBASE_, DP_)                                                                        \
	hipLaunchKernelGGL((search_se_kernel<NCH_, false, false, 2, 0, KCH_>), GRID_, block, 0, s, X, P, d_bases, d_offs, n, d_results,   \
	                   d_path_ops, d_path_used, no_stats3, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, wk.ticket,          \
	                   wk.hsp_lds_cap, wk.ovf_list, OVFBASE_, DP_, no_dp)
	// phase 3 parked (round 5): the first launch (PART 1: no banded DP inside), phase 3's flank DPs (every job made, no round lists),
	// the second launch over the reads parked there (PART 2; wk.ticket + 2: a work counter of its own)
#define URX_LAUNCH_SE_P3(NCH_)                                                                                                      \
	do {                                                                                                                            \
	hipLaunchKernelGGL((search_se_kernel<NCH_, false, false, true, 1>), grid, block, 0, s, X, P, d_bases, d_offs, n, d_results,       \
	                   d_path_ops, d_path_used, no_stats3, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, wk.ticket,          \
	                   wk.hsp_lds_cap, wk.ovf_list, no_ovf, wk.dp[0], wk.dp3);                                                       \
	stamp(STAGE_P3_MAIN);                                                                                                           \
	hipLaunchKernelGGL((dp_kernel<NCH_>), dim3((unsigned)wk.dp_blocks), block, 0, s, X, P, d_bases, d_offs, wk.dp3,                   \
	                   wk.dp_scratch, wk.dp_scratch_stride, X.seq, 0u, 0xFFFFFFFFu, wk.dp3.tickets, (const uint32_t *)nullptr,       \
	                   wk.dp3.counters);                                                                                             \
	stamp(STAGE_P3_DP);                                                                                                             \
	hipLaunchKernelGGL((search_se_kernel<NCH_, false, false, true, 2>), grid, block, 0, s, X, P, d_bases, d_offs, n, d_results,       \
	                   d_path_ops, d_path_used, no_stats3, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, wk.ticket3,         \
	                   wk.hsp_lds_cap, wk.ovf_list, no_ovf, wk.dp[0], wk.dp3);                                                       \
	} while (0)
	// phase 6 of the reads a pass parked: their flank DPs, then the ordered part
#define URX_LAUNCH_DP(NCH_, OVF_, PASS_)                                                                                          \
	do { /* the second pass is a few reads with many jobs each, usually none at all: smaller grids for the launches that go by reads */ \
	const unsigned dpg = (unsigned)wk.dp_blocks;                                                                                  \
	const unsigned fing = (unsigned)(wk.fin_blocks > 0 ? wk.fin_blocks : wk.blocks), fing2 = PASS_ && fing > 1024u ? 1024u : fing;  \
	const uint32_t *const DP_ROUND_LO = wk.dp_bounds.lo;                                                                          \
	hipLaunchKernelGGL(dp_round_lists_kernel, dim3(PASS_ ? 128 : 2048), dim3(256), 0, s, wk.dp[PASS_], wk.dp_bounds);             \
	for (int rd = 0; rd < DP_ROUNDS; ++rd) {                                                                                      \
		if (rd >= wk.dp_bounds.rounds) { stamp(2 + (2 * DP_ROUNDS + 1) * PASS_ + 2 * rd); stamp(3 + (2 * DP_ROUNDS + 1) * PASS_ + 2 * rd); continue; } \
		hipLaunchKernelGGL((dp_kernel<NCH_>), dim3(dpg), block, 0, s, X, P, d_bases, d_offs, wk.dp[PASS_],                        \
		                   wk.dp_scratch, wk.dp_scratch_stride, X.seq, DP_ROUND_LO[rd], DP_ROUND_LO[rd + 1],                      \
		                   wk.dp[PASS_].tickets + rd, wk.dp[PASS_].round_list + (size_t)rd * wk.dp[PASS_].jobs_cap,               \
		                   wk.dp[PASS_].tickets + 8 + rd);                                                                       \
		stamp(2 + (2 * DP_ROUNDS + 1) * PASS_ + 2 * rd);                                                                            \
		hipLaunchKernelGGL((finalize_se_kernel<NCH_, OVF_>), dim3(fing2), block,                                                  \
		                   0, s, X, P, d_offs, wk.dp[PASS_], d_results, d_path_ops, d_path_used, wk.hsp_lds_cap, wk.ovf_list,        \
		                   DP_ROUND_LO[rd], DP_ROUND_LO[rd + 1]);                                                                  \
		stamp(3 + (2 * DP_ROUNDS + 1) * PASS_ + 2 * rd);                                                                            \
	} } while (0)
	stamp(0);
	const bool diag = wk.stats != nullptr && (nch == 3 || nch == 4);  // diagnostic instantiations: 150 / 250 bp classes, phase 6 inline
	uint32_t *const no_stats3 = nullptr;
	// phase 3 parked: reads of up to 320 bases on an index with the row layout, phase 6 as launches of its own (the default)
	// round 6: every read of the batch has at most 128 k-mer starts (150 bases at W = 24): the instance that keeps two chunks of them (URMAPX_NO_K2=1: A/B, tests)
	const bool k2 = getenv("URMAPX_NO_K2") == nullptr && max_read_len >= X.W && max_read_len - (X.W - 1) <= 128u;
	const bool p3 = !diag && wk.dp3.jobs && wk.dp[0].jobs && wk.dp_blocks > 0 && X.rowinfo && nch <= 5 && wk.stats == nullptr;
	if (p3 && nch == 2) URX_LAUNCH_SE_P3(2);
	else if (p3 && nch == 3) URX_LAUNCH_SE_P3(3);
	else if (p3 && nch == 4) URX_LAUNCH_SE_P3(4);
	else if (p3) URX_LAUNCH_SE_P3(5);
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 2) URX_LAUNCH_SE_S16(2, grid, no_ovf, wk.dp[0]);  // slots, row lengths and second positions in one gather
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 3 && k2) URX_LAUNCH_SE_S16K(3, 2, grid, no_ovf, wk.dp[0]);  // 150-base reads: 127 k-mer starts
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 3) URX_LAUNCH_SE_S16(3, grid, no_ovf, wk.dp[0]);
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 4) URX_LAUNCH_SE_S16(4, grid, no_ovf, wk.dp[0]);
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 5) URX_LAUNCH_SE_S16(5, grid, no_ovf, wk.dp[0]);
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 8) URX_LAUNCH_SE_S16(8, grid, no_ovf, wk.dp[0]);
	else if (!diag && X.slot16 && wk.stats == nullptr && nch == 16) URX_LAUNCH_SE_S16(16, grid, no_ovf, wk.dp[0]);
	else if (diag && nch == 3) URX_LAUNCH_SE(3, false, true, grid, wk.stats, no_ovf, no_dp);
	else if (diag) URX_LAUNCH_SE(4, false, true, grid, wk.stats, no_ovf, no_dp);
	else if (nch == 2 && X.rowinfo) URX_LAUNCH_SE_ROWS(2, grid, wk.stats, no_ovf, wk.dp[0]);  // the chain rows are looked up in the layout built with the index
	else if (nch == 3 && X.rowinfo) URX_LAUNCH_SE_ROWS(3, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 4 && X.rowinfo) URX_LAUNCH_SE_ROWS(4, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 5 && X.rowinfo) URX_LAUNCH_SE_ROWS(5, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 8 && X.rowinfo) URX_LAUNCH_SE_ROWS(8, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 16 && X.rowinfo) URX_LAUNCH_SE_ROWS(16, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 2) URX_LAUNCH_SE(2, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 3) URX_LAUNCH_SE(3, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 4) URX_LAUNCH_SE(4, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 5) URX_LAUNCH_SE(5, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	else if (nch == 8) URX_LAUNCH_SE(8, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	else URX_LAUNCH_SE(16, false, false, grid, wk.stats, no_ovf, wk.dp[0]);
	if (!p3) { stamp(STAGE_P3_MAIN); stamp(STAGE_P3_DP); }
	stamp(1);
	if (wk.dp[0].jobs && !diag) {
		if (nch == 2) URX_LAUNCH_DP(2, false, 0);
		else if (nch == 3) URX_LAUNCH_DP(3, false, 0);
		else if (nch == 4) URX_LAUNCH_DP(4, false, 0);
		else if (nch == 5) URX_LAUNCH_DP(5, false, 0);
		else if (nch == 8) URX_LAUNCH_DP(8, false, 0);
		else URX_LAUNCH_DP(16, false, 0);
	} else
		for (int i = 0; i < 2 * DP_ROUNDS; ++i) stamp(2 + i);
	{
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	// second pass over the reads whose HSP or hit list outgrew the first pass's (the blocks read the count and leave
	// when it is zero): the same search with the lists continued in global scratch
	dim3 grid2((unsigned)(wk.blocks < SEARCH_OVF_BLOCKS ? wk.blocks : SEARCH_OVF_BLOCKS));
	uint2 *ovf_base = reinterpret_cast<uint2 *>(wk.scratch + (size_t)wk.blocks * wk.scratch_stride);
	uint32_t *const no_stats = nullptr;
	if (nch == 2) URX_LAUNCH_SE(2, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	else if (nch == 3) URX_LAUNCH_SE(3, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	else if (nch == 4) URX_LAUNCH_SE(4, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	else if (nch == 5) URX_LAUNCH_SE(5, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	else if (nch == 8) URX_LAUNCH_SE(8, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	else URX_LAUNCH_SE(16, true, false, grid2, no_stats, ovf_base, wk.dp[1]);
	stamp(2 + 2 * DP_ROUNDS);
	if (wk.dp[1].jobs) {
		if (nch == 2) URX_LAUNCH_DP(2, true, 1);
		else if (nch == 3) URX_LAUNCH_DP(3, true, 1);
		else if (nch == 4) URX_LAUNCH_DP(4, true, 1);
		else if (nch == 5) URX_LAUNCH_DP(5, true, 1);
		else if (nch == 8) URX_LAUNCH_DP(8, true, 1);
		else URX_LAUNCH_DP(16, true, 1);
	} else
		for (int i = 0; i < 2 * DP_ROUNDS; ++i) stamp(3 + 2 * DP_ROUNDS + i);
#undef URX_LAUNCH_SE
#undef URX_LAUNCH_SE_ROWS
#undef URX_LAUNCH_SE_P3
#undef URX_LAUNCH_SE_S16
#undef URX_LAUNCH_SE_S16K
#undef URX_LAUNCH_DP
	return hipGetLastError();
}

int dp_block_count(uint32_t max_read_len, int device) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
	int per_cu = 0;
	const int nchq = nch_for(max_read_len);
	hipError_t e = nchq == 2   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<2>, 64, 0)
	               : nchq == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<3>, 64, 0)
	               : nchq == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<4>, 64, 0)
	               : nchq == 5 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<5>, 64, 0)
	               : nchq == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<8>, 64, 0)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_kernel<16>, 64, 0);
	if (e != hipSuccess || per_cu < 1) per_cu = 8;
	return per_cu * prop.multiProcessorCount;
}

// finalize_se_kernel waits on memory (state in, jobs, state out): as many waves as fit
int fin_block_count(uint32_t max_read_len, int device) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
	int per_cu = 0;
	const int nchq = nch_for(max_read_len);
	hipError_t e = nchq == 2   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<2, false>, 64, 0)
	               : nchq == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<3, false>, 64, 0)
	               : nchq == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<4, false>, 64, 0)
	               : nchq == 5 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<5, false>, 64, 0)
	               : nchq == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<8, false>, 64, 0)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, finalize_se_kernel<16, false>, 64, 0);
	if (e != hipSuccess || per_cu < 1) per_cu = 16;
	return per_cu * prop.multiProcessorCount;
}

hipError_t launch_viterbi_batch(const urmapx_params &P, const uint8_t *d_a, const uint32_t *d_aoffs,
                                const uint8_t *d_b, const uint32_t *d_boffs, const uint8_t *d_flags, uint32_t n,
                                float *d_scores, uint8_t *d_status, urmapx_path_op *d_ops, uint16_t *d_nops,
                                uint8_t *d_scratch, hipStream_t s) {
	if (n == 0) return hipSuccess;
	// the packed-int16 rows are exact only while live cells stay clear of the int16 "dead" threshold (-16000, viterbi_dev.h):
	// with penalties beyond that bound the fp32 kernel runs whatever the test aid asks for
	const long worst = (long)VB_MAXL * std::max(std::max(std::labs((long)P.mismatch_score), std::labs((long)P.gap_ext_score)), 1L) + std::labs((long)P.gap_open_score);
	if (getenv("URMAPX_VITERBI_PAIR") && worst < 16000)  // test aid: two problems per wavefront, packed-int16 interior blocks
		hipLaunchKernelGGL(viterbi_batch_pair_kernel, dim3((n + 1) / 2), dim3(64), 0, s, P, d_a, d_aoffs, d_b, d_boffs, d_flags, n, d_scores,
		                   d_status, d_ops, d_nops, d_scratch, viterbi_batch_scratch_stride());
	else
		hipLaunchKernelGGL(viterbi_batch_kernel, dim3(n), dim3(64), 0, s, P, d_a, d_aoffs, d_b, d_boffs, d_flags, n, d_scores,
		                   d_status, d_ops, d_nops, d_scratch, viterbi_batch_scratch_stride());
	return hipGetLastError();
}

}  // namespace urx
