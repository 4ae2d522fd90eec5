// kernels_pe.hip -- paired-end search (urmap -map2): State2::Search4 (search2m4.cpp:15-208) and everything under it,
// one wavefront per read PAIR.
//
// The schedule is emulated step by step, wave-uniform, with the 64 lanes used inside each primitive (one DP
// diagonal per lane in Viterbi, 64 window positions per step in ScanSlots, one hit / HSP per lane in the list
// searches).  The ExtendPen outcome of every Both1 seed does not depend on the search state, so it is computed up
// front, one seed per lane (the same gather as search_se_kernel's), and the ordered pairing loop only consumes the
// cached outcomes; repeated calls on a seed that cannot change anything any more are dropped 64 at a time.
// The pending-list stage (SearchPE_Pending) still extends its chain rows one window at a time.
//
// Reference functions: GetFirstBoth1Seed / GetNextBoth1Seed (getseed.cpp:9-138), ExtendBoth1Pair4
// (search2m4.cpp:189-208), ExtendPen (extendpen.cpp:9-95), SearchPE_Pending (search1pepend.cpp:9-130),
// GetRow_Blob (ufindex.cpp:883-943), AlignHSP (alignhsp.cpp:60-172), FindPairs / ScanPair (state2.cpp:20-137),
// Scan / ScanSlots / ExtendScan / AddHSPScan (scan.cpp:14-39, scanslots.cpp:7-62, extendscan.cpp:8-187),
// AdjustTopHitsAndMapqs (search2.cpp:8-57), CalcMAPQ6 (search1m6.cpp:9-33), SetMappedPos (state1.cpp:129-145).
#include "dev_common.h"
#include "probe_dev.h"
#include "viterbi_dev.h"

namespace urx {

// URX_PE_DIET (round 6): the first pass's block at 10 240 bytes of LDS, so that FOUR of them fit a SIMD's share of a CU
// (with 128 VGPRs; the round-5 kernel: 12 864 B, 168 VGPRs, three waves).  Where the bytes come from:
//   1  one word of hits per mate in LDS (hits 65..128 of a mate in the block's global scratch, URX_PE_TAIL; without the tail such a pair
//      goes to the second pass), 64 HSPs of a mate in LDS instead of 128 (the list goes on in global scratch, as before), 136 seeds per mate
//      instead of QMAX (a pair with more: second pass)
//   2  (ships) both words of hits in LDS: 64 HSPs, 104 seeds per mate, a hit's run count in the upper bits of its score word, pending lists
//      of QMAX - 16 positions, the pending stage's prefix array inside seed_area
//   0  the round-5 kernel
// Measured (profiles/r6/ab_pe_waves4.txt, ab_pe_tail.txt), ms per 1 M reads on one box: 0: 21.60, 1: 20.35-20.48, 2: 19.59-19.63.
#ifndef URX_PE_DIET
#define URX_PE_DIET 2
#endif
static constexpr int PE_HIT_CAP = 64;
#ifndef URX_PE_TAIL
#define URX_PE_TAIL 1  // diet 1: hits 65..128 of a mate in global scratch (0: such a pair goes to the second pass)
#endif
// URX_PE_DIET 2: BOTH hit words stay in LDS and the bytes come from elsewhere -- 104 seeds per mate, a hit's run count in the upper bits of
// its score word (reads of up to 192 bases: score < 256), pending lists of QMAX - 16 positions, no tail pointer: 10 216 B
#define URX_PE_PACK (URX_PE_DIET == 2)
static constexpr int PE_HITW1 = (URX_PE_DIET == 1 && !URX_PE_TAIL) ? 1 : 2;     // hit-list words (64 hits each) of the first pass: 19 of 1 M reads end with 65..83 hits, none with more than 128
// ... of which in LDS.  On the diet the second word lives in the block's global scratch (hit_tail): the 19-in-a-million pairs that reach it are
// also the costliest of a batch, and as the second pass's whole work list they kept a launch of their own going for 1.8 ms (profiles/r6/ab_pe_waves4.txt)
static constexpr int PE_HITW1_LDS = URX_PE_DIET == 1 ? 1 : 2;
static constexpr int PE_HSP_CAP = URX_PE_DIET ? 64 : 128;       // HSPs of a mate held in LDS
static constexpr int PE_HSP_OVF_CAP = 8064;  // per mate, in global scratch
static constexpr int PE_OVF_BLOCKS = 1024;   // grid of the second pass (the costliest pairs of a batch)
static constexpr int PE_T2_BLOCKS = 64;      // grid of the third pass (pairs with more than 256 hits on a mate)
static constexpr int PE_TICKET_CHUNK = 2;
static constexpr int PE_ROW_CAP = 32;   // UFIndex m_MaxIx of every index this build accepts
static constexpr int PE_SCAN_SEG = 1024;  // SCAN_DB_SEG_LENGTH, state2.cpp:92
static constexpr uint32_t PRIME_STRIDE = 27, SCANK = 4;
static constexpr int MAX_TL = 1000;

// per-block global scratch: hit paths of both mates, the wide-band Viterbi scratch, then the pending rows
__host__ __device__ inline size_t pe_rowstore_offset(int qmax) {
	size_t b = (size_t)2 * PE_HIT_CAP * PE_HITW1 * URMAPX_MAX_PATH_OPS * 2 + WideScratch::bytes(qmax, PE_SCAN_SEG + 2 * qmax + 64);
	return (b + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t pe_tb_offset(int qmax);
__host__ __device__ inline size_t pe_tail_offset(int qmax);

// TIER: 0 = first pass (hit lists of PE_HITW1 x 64 per mate), 1 = second pass over the pairs that outgrew a list (4 x 64
// hits, HSP lists continued in global memory), 2 = third pass over the pairs that outgrew those (PE_HITW2 x 64 = 1024
// hits per mate: the reference's list has no bound, state1.cpp:193-228; this pass exists so that a pair in a satellite
// is mapped, not flagged -- it runs a handful of pairs per run and is not tuned)
static constexpr int PE_HITW2 = 16;
// The first pass's hits beyond its LDS word (diet): kept out of line, as search_se_kernel's hsp_overflow_add, so that the registers of a
// path 19 pairs in a million take do not count against the loops every pair runs.  tail: [k] position, [64 + k] score << 1 | plus, [128 + k] runs.
__device__ __noinline__ bool pe_tail_overlaps_any(const uint32_t *tail, int nt, uint32_t db) {  // wave-uniform db: one hit per lane
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	return __ballot(lane < nt && (tail[lane] >> 6) == (db >> 6)) != 0;
}
__device__ __noinline__ bool pe_tail_overlaps_each(const uint32_t *tail, int nt, uint32_t db) {  // a db per lane
	bool ov = false;
	for (int k = 0; k < nt; ++k) ov |= (tail[k] >> 6) == (db >> 6);
	return ov;
}
__device__ __noinline__ uint32_t pe_tail_get(const uint32_t *tail, int at) { return tail[at]; }
__device__ __noinline__ void pe_tail_put(uint32_t *tail, int k, uint32_t db, uint32_t sp, uint32_t nops) {
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	if (lane == 0) { tail[k] = db; tail[64 + k] = sp; tail[128 + k] = nops; }
	__threadfence_block();
}

// The wave-uniform state of a mate that changes as the pair is searched.  It lives in LDS, not in the Mate object: the Mate
// objects are in private memory (the pairing loop picks a mate at run time), where every access is a scratch load or
// store -- a memory round trip for a counter.  Through `hot` it is a ds_read / ds_write (all lanes, same address).
struct MateHot {
	int QL, nwords;
	int pendCount[2];
	int hitCount, hspCount, topHit;
	int maxPen, best, second, bestHSP;
	uint32_t mapq;
	uint32_t status;
	// set once per kernel or per pair, read everywhere
	int W, wide_lds_dwords, hit_cap, hit_wsh, hsp_lds, dbg_cut;
	bool q_other;  // the read holds a byte outside the code list: its windows are compared as ASCII
	uint2 *hsp_ovf;             // this mate's HSP list beyond LDS, global
	urmapx_path_op *hit_paths;  // its hits' paths, global
#if !URX_PE_PACK
	uint32_t *hit_tail;         // first pass on the diet: hits beyond the LDS word -- [k] position, [64 + k] score << 1 | plus, [128 + k] runs of the path; global
#endif
};
// Everything a mate owns, as ONE LDS object per mate.  Round 4: a Mate is now a handful of pointers held in registers -- the
// pointer to this object is the only thing that differs between the two mates, so `the mate picked at run time` is an address
// computation, not an array of objects in private memory (where every member access was a scratch load: 331 scratch
// instructions per pair in round 3).  The hit list lives here too (it was four per-lane words per mate in private memory).
static constexpr int pe_hitw(int tier) { return tier == 0 ? PE_HITW1_LDS : tier == 1 ? 4 : PE_HITW2; }
template <int NCH, int TIER>
struct MateLds {
	static constexpr int QMAX = 64 * NCH;
	static constexpr int HITS = 64 * pe_hitw(TIER);
	MateHot hot;
	__attribute__((aligned(16))) uint4 qpl[2][2 * NCH];  // the two strands as bit planes of 4-bit codes (dev_common.h: seq_code), one uint4 per 32 bases
	__attribute__((aligned(16))) uint8_t sQ[2][QMAX];     // [0] read as given, [1] reverse complement
	uint32_t hit_db[HITS];
	uint16_t hit_sp[HITS];   // score << 1 | plus
	static constexpr bool PACK = URX_PE_PACK && TIER == 0 && NCH <= 3;  // the run count rides in hit_sp (bits 9..15)
	uint8_t hit_nops[PACK ? 4 : HITS];  // runs of a hit's path (<= URMAPX_MAX_PATH_OPS = 96)
	uint32_t hsp_db[PE_HSP_CAP], hsp_ql[PE_HSP_CAP];
	uint16_t hsp_sf[PE_HSP_CAP];
	uint8_t pend[2][PACK ? QMAX - 16 : QMAX];   // pending query positions (stored in a byte, state1.h:86-87); at most QL - W + 1 <= QMAX - 20 of them (W >= 21)
	uint64_t kpl[4][NCH + 1];  // letter planes of the read (probe_dev.h: probe_pair, slot_from_planes)
};
template <int NCH, int TIER>
struct Mate {
	static constexpr bool OVF = TIER > 0;
	static constexpr int QMAX = 64 * NCH;
	static constexpr int TB_ROWS8 = QMAX / 8 + 2;
	static constexpr int HITW = pe_hitw(TIER);
	// wave constants (the same in both mates' objects: registers)
	const DevIndex *X;
	const urmapx_params *P;
	const uint8_t *__restrict__ gseq;
	const uint8_t *__restrict__ gblob;
	const uint4 *__restrict__ gseqp;  // packed copy of the sequence store
#define lane ((int)threadIdx.x)
	// this mate: everything it owns is behind this one LDS pointer
	lds_ptr<MateLds<NCH, TIER>> L;
#define hot (&L->hot)
#define W (hot->W)
#define wide_lds_dwords (hot->wide_lds_dwords)
#define hit_cap (hot->hit_cap)
#define hit_wsh (hot->hit_wsh)
#define hsp_lds (hot->hsp_lds)
#define dbg_cut (hot->dbg_cut)
#define q_other (hot->q_other)
#define hsp_ovf (hot->hsp_ovf)
#define hit_paths (hot->hit_paths)
#if URX_PE_PACK
#define hit_tail ((uint32_t *)nullptr)
#else
#define hit_tail (hot->hit_tail)
#endif
#define QL (hot->QL)
#define nwords (hot->nwords)
#define pendCount (hot->pendCount)
#define hitCount (hot->hitCount)
#define hspCount (hot->hspCount)
#define topHit (hot->topHit)
#define maxPen (hot->maxPen)
#define best (hot->best)
#define second (hot->second)
#define bestHSP (hot->bestHSP)
#define mapq (hot->mapq)
#define status (hot->status)
#define sQ (L->sQ)
#define qpl (L->qpl)
#define hit_db (L->hit_db)
#define hit_sp (L->hit_sp)
#define hit_nops (L->hit_nops)
#define hsp_db (L->hsp_db)
#define hsp_ql (L->hsp_ql)
#define hsp_sf (L->hsp_sf)
#define pend (L->pend)
	// shared LDS scratch (one set per wave)
	lds_ptr<uint8_t> sT;
	uint32_t *tb;         // trace cells of the banded DP: this block's global scratch
	lds_ptr<uint32_t> wide_lds;   // LDS rows of the wide-band DP (the rescue's whole-read Viterbi)
	lds_ptr<uint16_t> ropsL, ropsR, cand;
	WideScratch ws;
	// hit k: hit_db[k], hit_sp[k], hit_nops[k] in LDS; its path in hit_paths (global) [k][URMAPX_MAX_PATH_OPS]
	// capacity: first pass hit_cap (PE_HITW1 x 64; a test aid lowers it); later passes HITW << hit_wsh (HITW x 64; the test
	// aid that lowers the first pass's caps makes it HITW x 16 in the second pass, so that a fixture with a few dozen hits
	// per mate reaches the third)
	__device__ __forceinline__ int hits_room() const { return OVF ? HITW << hit_wsh : hit_cap; }
	static constexpr int HITS_LDS = MateLds<NCH, TIER>::HITS;
	static constexpr bool TAIL = TIER == 0 && PE_HITW1_LDS < PE_HITW1;  // hits beyond HITS_LDS exist and live in hit_tail
	__device__ __forceinline__ uint32_t hdb(int i) const { return (TAIL && i >= HITS_LDS) ? pe_tail_get(hit_tail, i - HITS_LDS) : hit_db[i]; }      // i wave-uniform
	static constexpr bool PACK = MateLds<NCH, TIER>::PACK;
	__device__ __forceinline__ uint32_t hsp_of(int i) const {
		if constexpr (PACK) return (uint32_t)hit_sp[i] & 0x1FFu;
		return (TAIL && i >= HITS_LDS) ? pe_tail_get(hit_tail, 64 + i - HITS_LDS) : (uint32_t)hit_sp[i];
	}
	__device__ __forceinline__ int hnops(int i) const {
		if constexpr (PACK) return (int)(hit_sp[i] >> 9);
		return (TAIL && i >= HITS_LDS) ? (int)pe_tail_get(hit_tail, 128 + i - HITS_LDS) : (int)hit_nops[i];
	}
	// HSPs: LDS [PE_HSP_CAP] (hsp_db, hsp_ql, hsp_sf); beyond hsp_lds in global scratch as {db, startq | len << 9 | sf << 18}
	// dbg_cut (MateHot): diagnostic only (URMAPX_DEBUG_STOP_PE 41 / 42 / 43): leave search_pending after that part
	lds_ptr<uint8_t> rowlen;      // LDS [2 * QMAX]: row length of every pending position, [strand][i]
	lds_ptr<uint16_t> pre;        // LDS [65]
	lds_ptr<uint32_t> cq_db;      // LDS [128]: candidate queue of the pending stage (ring)
	lds_ptr<uint16_t> cq_qp;      // LDS [128]
	uint32_t *rowstore;   // global, this block: [strand][chunk][k][lane]
	// Round 6: with DevIndex::slot16 the pending stage keeps two words per chain head (second position or row index, first position) and keeps them in LDS --
	// 2 x 2 NCH x 64 words behind the prefix array in seed_area, which is idle while the chain rows are gathered and consumed (AlignHSP's buffers and the
	// wide-band DP's rows live there at other times): no store per group to global scratch, no L2 round trip in front of every candidate step
	lds_ptr<uint32_t> rs_lds;

	__device__ __forceinline__ bool overlaps_hit(uint32_t db) const {
		bool eq = false;
		const int n = hitCount;
		if constexpr (TAIL) {
			eq = lane < n && (hit_db[lane] >> 6) == (db >> 6);  // (HITS_LDS == 64)
			if (__ballot(eq) != 0) return true;
			return n > HITS_LDS && pe_tail_overlaps_any(hit_tail, n - HITS_LDS, db);
		}
		for (int b = 0; b < n; b += 64) eq |= b + lane < n && (hit_db[b + lane] >> 6) == (db >> 6);
		return __ballot(eq) != 0;
	}

	// per-lane form of OverlapsHit
	__device__ __forceinline__ bool overlaps_any_hit(uint32_t db) const {
		bool ov = false;
		const int n = hitCount;
		for (int k = 0; k < (TAIL && n > HITS_LDS ? HITS_LDS : n); ++k) ov |= (hit_db[k] >> 6) == (db >> 6);  // one address for all lanes: an LDS broadcast
		if (TAIL && n > HITS_LDS) ov |= pe_tail_overlaps_each(hit_tail, n - HITS_LDS, db);
		return ov;
	}

	// state1.cpp:508-551; returns the hit index or -1.  A path, if any, is in `cand`.
	__device__ __forceinline__ int add_hit(uint32_t db, bool plus, int score, int cand_nops) {
		if (score < 10) return -1;
		if (overlaps_hit(db)) return -1;
		int mp = (QL - score) - 2 * P->mismatch_score;
		if (mp < maxPen) maxPen = mp;
		const int idx = hitCount;
		bool keep = true;
		if (score > best) { second = best; best = score; topHit = idx; }
		else if (score == best) second = score;
		else {
			if (score < best - 12) keep = false;
			else if (score > second) second = score;
		}
		if (!keep) return -1;
		if (hitCount >= hits_room()) { status |= URMAPX_ST_HIT_OVERFLOW; if (topHit == idx) topHit = -1; return -1; }
		if (TAIL && idx >= HITS_LDS) pe_tail_put(hit_tail, idx - HITS_LDS, db, ((uint32_t)score << 1) | (plus ? 1u : 0u), (uint32_t)cand_nops);
		else if (lane == 0) {
			hit_db[idx] = db;
			if constexpr (PACK) hit_sp[idx] = (uint16_t)(((uint32_t)score << 1) | (plus ? 1u : 0u) | ((uint32_t)cand_nops << 9));
			else { hit_sp[idx] = (uint16_t)(((uint32_t)score << 1) | (plus ? 1u : 0u)); hit_nops[idx] = (uint8_t)cand_nops; }
		}
		for (int t = lane; t < cand_nops; t += 64) hit_paths[(size_t)idx * URMAPX_MAX_PATH_OPS + t] = cand[t];
		URX_SYNC();
		++hitCount;
		return idx;
	}

	// HSP k: LDS below hsp_lds; beyond it the list continues in this block's global scratch (both passes: re-mapping the
	// pairs with long HSP lists in a second launch cost more than carrying the code, see search_se_kernel).
	// sf = score << 2 | aligned << 1 | plus.
	__device__ __forceinline__ void hsp_get(int k, uint32_t &db, uint32_t &ql, uint32_t &sf) const {
		if (k < hsp_lds) { db = hsp_db[k]; ql = hsp_ql[k]; sf = hsp_sf[k]; }
		else { const uint2 e = hsp_ovf[k - hsp_lds]; db = e.x; ql = (e.y & 511u) | (((e.y >> 9) & 511u) << 16); sf = e.y >> 18; }
	}
	__device__ __forceinline__ void hsp_put(int k, uint32_t db, uint32_t ql, uint32_t sf) {  // one lane
		if (k < hsp_lds) { hsp_db[k] = db; hsp_ql[k] = ql; hsp_sf[k] = (uint16_t)sf; }
		else hsp_ovf[k - hsp_lds] = make_uint2(db, (ql & 511u) | ((ql >> 16) << 9) | (sf << 18));
	}
	__device__ __forceinline__ int hsp_room() const { return hsp_lds + PE_HSP_OVF_CAP; }
	__device__ __forceinline__ int hsp_score(int k) const {
		uint32_t db, ql, sf;
		hsp_get(k, db, ql, sf);
		return (int)(sf >> 2);
	}

	__device__ __forceinline__ int find_hsp_diag(uint32_t diag) const {
		for (int base = 0; base < hspCount; base += 64) {
			const int i = base + lane;
			bool eq = false;
			if (i < hspCount) {
				uint32_t db, ql, sf;
				hsp_get(i, db, ql, sf);
				eq = (db - (ql & 0xFFFFu)) == diag;
			}
			uint64_t m = __ballot(eq);
			if (m) return base + __builtin_ctzll(m);
		}
		return -1;
	}
	__device__ __forceinline__ void put_hsp(int k, uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		if (lane == 0) hsp_put(k, startdb, startq | (len << 16), (uint32_t)((score << 2) | (plus ? 1 : 0)));
		URX_SYNC();
	}
	// state1.cpp:553-591
	__device__ __forceinline__ void add_hsp(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		if (score < best - 4) return;
		int k = find_hsp_diag(startdb - startq);
		if (k >= 0) {
			if (score > hsp_score(k)) put_hsp(k, startq, startdb, plus, len, score);
			return;
		}
		if (hspCount >= hsp_room()) { status |= URMAPX_ST_HSP_OVERFLOW; return; }
		put_hsp(hspCount, startq, startdb, plus, len, score);
		++hspCount;
		if (score > bestHSP) bestHSP = score;
	}
	// extendscan.cpp:8-49
	__device__ __forceinline__ int add_hsp_scan(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		int k = find_hsp_diag(startdb - startq);
		if (k >= 0) {
			if (score > hsp_score(k)) put_hsp(k, startq, startdb, plus, len, score);
			return k;
		}
		if (hspCount >= hsp_room()) { status |= URMAPX_ST_HSP_OVERFLOW; return -1; }
		k = hspCount;
		put_hsp(k, startq, startdb, plus, len, score);
		++hspCount;
		if (score > bestHSP) bestHSP = score;
		return k;
	}

	__device__ __forceinline__ void window_mask(uint32_t dblo, bool plus, BitVec<NCH> &mm) const {
		const uint8_t *t = gseq + dblo;
		const int s = plus ? 0 : 1;
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			const int p = 64 * c + lane;
			bool ne = false;
			if (p < QL) ne = ((uint32_t)t[p] != (uint32_t)sQ[s][p]);
			mm.w[c] = __ballot(ne);
		}
	}

	// mismatch bit vector of the whole read against the window at dblo, one candidate per lane: packed planes, or ASCII for
	// a read with bytes outside the code list (wave-uniform choice)
	__device__ __forceinline__ void lane_mask(uint32_t dblo, bool plus, uint64_t (&mm)[NCH]) const {
		if (q_other) lane_mismatch_mask<NCH>(gseq, dblo, from_lds(&sQ[plus ? 0 : 1][0]), QL, mm);
		else lane_mismatch_planes<NCH>(gseqp, dblo, from_lds((lds_ptr<const uint4>)&qpl[plus ? 0 : 1][0]), QL, mm);
	}

	// The two x-drop walks shared by ExtendPen (extendpen.cpp:25-78) and ExtendScan (extendscan.cpp:77-133).
	// scan = true: the leftward walk does not add to the penalty (the reference's omission, kept).
	// Returns false if the penalty cap aborted the extension.
	__device__ __forceinline__ bool xdrop_walk(const BitVec<NCH> &mm, int seedq, bool scan, int &bst, int &startpos, int &endpos) const {
		const int mis = P->mismatch_score, xdrop = P->xdrop;
		int pen = 0, score = W;
		bst = 0;
		endpos = seedq + W - 1;
		int cur = endpos + 1;
		while (cur < QL) {
			int m = mm.next_set(cur);
			if (m > QL) m = QL;
			const int run = m - cur;
			if (run > 0) { score += run; if (score > bst) { bst = score; endpos = m - 1; } }
			if (m >= QL) break;
			pen -= mis;
			if (pen > maxPen) return false;
			score += mis;
			if (bst - score > xdrop) break;
			cur = m + 1;
		}
		startpos = seedq;
		cur = startpos - 1;
		while (cur >= 0) {
			const int m = mm.prev_set(cur);
			const int run = cur - m;
			if (run > 0) { score += run; if (score > bst) { bst = score; startpos = m + 1; } }
			if (m < 0) break;
			if (!scan) pen -= mis;
			if (pen > maxPen) return false;
			score += mis;
			if (bst - score > xdrop) break;
			cur = m - 1;
		}
		return true;
	}

	// extendpen.cpp:9-95
	__device__ __forceinline__ int extend_pen(uint32_t seedq, uint32_t seeddb, bool plus) {
		if (seeddb < seedq) return -1;
		const uint32_t dblo = seeddb - seedq;
		if (overlaps_hit(dblo)) return -1;
		BitVec<NCH> mm;
		window_mask(dblo, plus, mm);
		int bst, sp, ep;
		if (!xdrop_walk(mm, (int)seedq, false, bst, sp, ep)) return -1;
		if (sp == 0 && ep == QL - 1) {
			add_hit(dblo, plus, bst, 0);
			return bst;
		}
		const int minhsp = (int)((uint32_t)P->min_hsp_score_pct * (uint32_t)QL / 100.0);
		if (bst >= minhsp) {
			add_hsp((uint32_t)sp, dblo + (uint32_t)sp, plus, (uint32_t)(ep - sp + 1), bst);
			return -2;
		}
		return -1;
	}

	// ExtendPen (extendpen.cpp:9-95) from the outcome cached by the seed gather: res = kind << 27 | end << 18 |
	// start << 9 | score of the uncapped x-drop walk, pen = its accumulated penalty (the capped walk aborts iff the
	// final penalty exceeds the cap, because the penalty only grows along the walk)
	__device__ __forceinline__ int extend_pen_cached(uint32_t seedq, uint32_t seeddb, bool plus, uint32_t res, int pen) {
		if (seeddb < seedq) return -1;
		const uint32_t dblo = seeddb - seedq;
		if (overlaps_hit(dblo)) return -1;
		if (pen > maxPen) return -1;
		const int kind = (int)(res >> 27), bst = (int)(res & 511u);
		if (kind == 1) {
			add_hit(dblo, plus, bst, 0);
			return bst;
		}
		if (kind == 2) {
			const uint32_t sp = (res >> 9) & 511u, ep = (res >> 18) & 511u;
			add_hsp(sp, dblo + sp, plus, ep - sp + 1, bst);
			return -2;
		}
		return -1;
	}

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

	// alignhsp.cpp:60-172; returns the new hit index or -1
	__device__ __forceinline__ int align_hsp(int k) {
		uint32_t startdb, ql, sf;
		hsp_get(k, startdb, ql, sf);
		if (sf & 2u) return -1;
		URX_SYNC();
		if (lane == 0) hsp_put(k, startdb, ql, sf | 2u);
		const int startq = (int)(ql & 0xFFFFu), len = (int)(ql >> 16);
		const int hscore = (int)(sf >> 2);
		const bool plus = sf & 1u;
		URX_SYNC();
		int totalPen = len - hscore;
		int totalScore = hscore;
		if (totalPen > maxPen) return -1;
		const int BR = 2 * (int)P->band_radius;
		const uint32_t TL = X->seqDataSize;
		uint32_t combinedTLo = startdb;
		const uint8_t *Q = from_lds(&sQ[plus ? 0 : 1][0]);
		int nL = 0, nR = 0;
		int rtrim = 0;
		uint32_t vst = 0;
		const VPar VP(*P);
		// the two flanks through ONE copy of the banded DP (as search_se_kernel: the code has to stay near the instruction cache's size)
		const int rightQLo = startq + len;
#pragma unroll 1
		for (int side = 0; side < 2; ++side) {
			const bool left = side == 0;
			int fql;
			uint32_t tlo, tl;
			const uint8_t *fq;
			if (left) {
				if (startq <= 0) continue;
				if (startdb < (uint32_t)startq) return -1;
				fql = startq;
				const uint32_t leftTHi = startdb - 1;
				tl = (uint32_t)(fql + BR);
				if (tl >= leftTHi) return -1;
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
			if (load_window(tlo, (int)tl)) return -1;
			// the DP stops as soon as the flank cannot stay within what the penalty cap leaves (viterbi_dev.h); the test
			// that would discard it follows right below, so the outcome is the same
			const int allGap = P->gap_open_score + (fql - 1) * P->gap_ext_score;
			const int need = fql - (maxPen - totalPen);
			bool aborted = false;
			RevOps R;
			R.ops = from_lds(left ? ropsL : ropsR);
			int score = (int)viterbi_wave<true>(VP, fq, fql, from_lds(sT), (int)tl, left, !left, tb, TB_ROWS8, ws, R, vst, lane, (float)need,
			                                    allGap < need ? &aborted : nullptr, from_lds(wide_lds), wide_lds_dwords);
			if (aborted) return -1;
			status |= vst;
			if (left) {
				nL = R.n;
				// TrimLeftIs (pathinfo.cpp:153-171): the leading I run is the last run in traceback order
				int nTrimI = 0;
				if (nL > 0) {
					uint32_t lastop = ropsL[nL - 1];
					if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --nL; }
				}
				combinedTLo = tlo + (uint32_t)nTrimI;
			} else {
				nR = R.n;
				// TrimRightIs (pathinfo.cpp:173-190): trailing I run = first run in traceback order, never the whole path
				if (nR > 1 && (ropsR[0] & 3u) == OP_I) rtrim = 1;
			}
			if (allGap > score) score = allGap;
			totalScore += score;
			totalPen += fql - score;
			if (totalPen > maxPen) return -1;
		}
		if (status & (URMAPX_ST_BAND_TOO_WIDE | URMAPX_ST_PATH_OVERFLOW)) return -1;
		int nc = 0, cop = -1, clen = 0;
		bool ovf = false;
		auto put = [&](int op, int l) {
			if (l <= 0) return;
			if (op == cop) { clen += l; return; }
			if (clen) { if (nc < URMAPX_MAX_PATH_OPS) { if (lane == 0) cand[nc] = (uint16_t)((clen << 2) | cop); ++nc; } else ovf = true; }
			cop = op; clen = l;
		};
		for (int t = nL - 1; t >= 0; --t) { uint32_t o = ropsL[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(OP_M, len);
		for (int t = nR - 1; t >= rtrim; --t) { uint32_t o = ropsR[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(-2, 1);
		if (ovf) { status |= URMAPX_ST_PATH_OVERFLOW; return -1; }
		URX_SYNC();
		return add_hit(combinedTLo, plus, totalScore, nc);
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
		uint32_t mq = (uint32_t)x;
		if (mq > 40) mq = 40;
		return mq;
	}

	// ufindex.cpp:883-943 (wave-uniform walk); positions land on lanes 0..K-1 of `row`
	__device__ __forceinline__ int get_row(uint64_t slot, uint32_t T, uint32_t pos, uint32_t &row) const {
		if ((T & TALLY_MY_BIT) == 0) return 0;
		uint64_t slot2 = slot;
		int K = 0;
		const uint64_t N = X->slotCount;
		for (;;) {
			if (K > 0) {
				uint32_t t2, p2;
				load_slot(gblob, slot2, t2, p2);
				T = uni(t2); pos = uni(p2);
			}
			if (lane == K) row = pos;
			++K;
			if (K == (int)X->maxIx || K >= 64) return K;
			if (T == TALLY_PLUS1 || T == TALLY_BOTH1) return 1;
			if (T == TALLY_END) return K;
			if (T == TALLY_LONG_MINE || T == TALLY_LONG_OTHER) {
				uint64_t slotA = addmod(slot2, pos & 0xFFFFu, N);
				slot2 = addmod(slotA, pos >> 16, N);
				uint32_t tA, pA;
				load_slot(gblob, slotA, tA, pA);
				pA = uni(pA);
				if (lane == K - 1) row = pA;
			} else
				slot2 = addmod(slot2, T & TALLY_NEXT_MASK, N);
		}
	}

	// search1pepend.cpp:9-130.  The reference walks the pending query positions one by one (GetRow_Blob, then ExtendPen on
	// every row entry): rows of length <= 2 first, plus strand then minus, longer rows in a second round.  Here, per
	// strand, the chains of 64 pending positions are walked at once (one chain per lane), and each (round, strand,
	// chunk) group of row entries goes through the same gather / ordered-consume split as search_se_kernel: outcome of
	// the uncapped x-drop walk per lane, then only the state-changing candidates in the reference's order.
	// (round 4: inlined at its one call site -- as a function of its own it took `this` as a pointer, i.e. the Mate object in
	// private memory and every pointer in it a scratch load; AlignHSP has one call site in here, a loop that runs once for
	// the phase-3 alignments (if the best HSP is long enough) and once for the final ones)
	__device__ __forceinline__ void search_pending() {
		maxPen = P->max_penalty;
		const int minScore1 = QL + P->xphase1 * P->mismatch_score;
		const int termHSP3 = (QL * P->term_hsp_score_pct_phase3) / 100;
		if (best >= minScore1) { mapq = calc_mapq(); return; }
#pragma unroll 1
		for (int stage = bestHSP >= termHSP3 ? 0 : 1; stage < 2; ++stage) {
			if (stage == 1) {
				if (dbg_cut == 41) break;
				if (!pending_rows()) break;  // diagnostic cut
			}
			const int bmin = stage == 0 ? -0x7FFFFFFF : max(best, bestHSP) - 8;
			for (int k = 0; k < hspCount; ++k) {
				if (stage == 1 && hsp_score(k) < bmin) continue;
				align_hsp(k);
			}
			if (stage == 0 && best >= minScore1) break;
		}
		mapq = calc_mapq();
	}

	// the chain walks and ExtendPen calls of SearchPE_Pending (search1pepend.cpp:46-116); false: a diagnostic cut says stop
	__device__ __forceinline__ bool pending_rows() {
		const int minhsp = (int)((uint32_t)P->min_hsp_score_pct * (uint32_t)QL / 100.0);
		const uint64_t N = X->slotCount;
		const int maxIx = (int)X->maxIx;
		// 1. all chains: rowstore[strand][chunk][k][lane], row length in rowlen[strand][i].  The [strand][chunk] groups advance
		// in lock step (as search_se_kernel's walk_run): one dependent slot load per hop for all of the mate's chains, where
		// round 3 walked the plus strand's chains to their ends before it started on the minus strand's
		{
			constexpr int NG = 2 * NCH;
			uint64_t wsl[NG];
			uint32_t wT[NG], wps[NG];
			int wK[NG];
			bool wact[NG];
			bool any = false;
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const int st = g / NCH, i = 64 * (g % NCH) + lane;
				wK[g] = 0; wsl[g] = 0; wT[g] = 0; wps[g] = 0;
				wact[g] = i < pendCount[st];
				// the head's slot number from the read's letter planes (round 3 read slot, tally and position from probe arrays in
				// HBM that every pair had written); its tally and position from the table again
				if (wact[g]) wsl[g] = slot_from_planes<NCH + 1>(*X, (lds_ptr<const uint64_t>)&L->kpl[0][0], (uint32_t)nwords, st, pend[st][i]);
			}
			// round 5: with DevIndex::slot16 the head's slot brings its row's length and second position (or the row's index) along:
			// one read per pending k-mer instead of slot + info entry + group base
			const bool s16 = X->slot16 != nullptr;
			uint32_t wx[NG];
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				wx[g] = 0;
				if (!wact[g]) continue;
				if (s16) {
					const uint4 v = X->slot16[wsl[g]];
					wps[g] = v.x; wT[g] = v.y & 0xFFu; wK[g] = (int)((v.y >> 8) & 0xFFu); wx[g] = v.z;
				} else
					load_slot(gblob, wsl[g], wT[g], wps[g]);
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				wact[g] = wact[g] && (wT[g] & TALLY_MY_BIT) != 0;  // GetRow_Blob returns 0 for a slot that is not "mine"
				any |= wact[g];
			}
			const bool lookup = X->rowinfo != nullptr;
			if (s16) {
				// rowstore[g] = second position (rows of two) or the row's index in X->rows, [NG + g] = position 0; wK is the row's length
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					if (!wact[g]) { wK[g] = 0; continue; }
					rs_lds[g * 64 + lane] = wx[g];
					rs_lds[(NG + g) * 64 + lane] = wps[g];
				}
				any = false;
			} else if (lookup) {
				// the rows are looked up in the layout built with the index (chain_rows.hip; search_se_kernel's rows_fetch): the head's
				// info word gives the row's length and where it lies in X->rows; rowstore[g][lane] keeps that index (0xFFFFFFFF for a
				// single-entry head, whose row is its own position in rowstore[NG + g][lane]) and the candidate stage reads it there
				// (round 5: the info entry carries the row's second position -- rows of two need no read of X->rows; rowstore[NG + g] =
				// position 0, [2 NG + g] = position 1, [g] = the row's index in X->rows for rows of three and more)
				uint2 info[NG];
				bool longhead = false;
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					info[g] = make_uint2(0u, 0u);
					if (wact[g] && wT[g] != TALLY_PLUS1 && wT[g] != TALLY_BOTH1) info[g] = X->rowinfo[wsl[g]];
					longhead |= wact[g] && wT[g] == TALLY_LONG_MINE;
				}
				const bool any_long = __ballot(longhead) != 0;  // a head whose own slot holds a long link's steps, not a position
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					if (!wact[g]) continue;
					uint32_t p0 = wps[g];
					if (wT[g] == TALLY_PLUS1 || wT[g] == TALLY_BOTH1) wK[g] = 1;
					else {
						wK[g] = (int)(info[g].x & 0xFFu);
						uint32_t at = 0;
						if (wK[g] > 2 || (any_long && wT[g] == TALLY_LONG_MINE)) at = (uint32_t)(X->rowbase[wsl[g] >> 10] + (info[g].x >> 8));
						if (any_long && wT[g] == TALLY_LONG_MINE) p0 = X->rows[at];
						rowstore[(size_t)g * 64 + lane] = at;
						rowstore[(size_t)(2 * NG + g) * 64 + lane] = info[g].y;
					}
					rowstore[(size_t)(NG + g) * 64 + lane] = p0;
				}
				any = false;
			}
			while (__ballot(any)) {
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					if (!wact[g]) continue;
					uint32_t *rs = rowstore + ((size_t)g * PE_ROW_CAP) * 64 + lane;
					rs[wK[g] * 64] = wps[g];
					++wK[g];
					const uint32_t t = wT[g];
					if (wK[g] == maxIx || wK[g] >= PE_ROW_CAP) wact[g] = false;
					else if (t == TALLY_PLUS1 || t == TALLY_BOTH1) { wK[g] = 1; wact[g] = false; }
					else if (t == TALLY_END) wact[g] = false;
					else if (t == TALLY_LONG_MINE || t == TALLY_LONG_OTHER) {
						const uint64_t slotA = addmod(wsl[g], wps[g] & 0xFFFFu, N);
						wsl[g] = addmod(slotA, wps[g] >> 16, N);
						uint32_t tA, pA;
						load_slot(gblob, slotA, tA, pA);
						rs[(wK[g] - 1) * 64] = pA;
					} else
						wsl[g] = addmod(wsl[g], t & TALLY_NEXT_MASK, N);
				}
				any = false;
#pragma unroll
				for (int g = 0; g < NG; ++g) {
					if (wact[g]) load_slot(gblob, wsl[g], wT[g], wps[g]);
					any |= wact[g];
				}
			}
#pragma unroll
			for (int g = 0; g < NG; ++g) {
				const int st = g / NCH, i = 64 * (g % NCH) + lane;
				if (i < pendCount[st]) rowlen[st * QMAX + i] = (uint8_t)wK[g];
			}
		}
		URX_SYNC();
		if (dbg_cut == 42) return false;
		// 2. the four groups in the reference's order.  Candidates that survive the hit-diagonal filter are compacted, in
		// order, into a 128-entry LDS queue (reference position, query position | plus << 15), so that the gather below
		// always runs on full batches even though most (round, strand, chunk) groups hold only a few row entries.
		int qhead = 0, qcount = 0;
		auto drain = [&](bool all) {
			while (qcount >= 64 || (all && qcount > 0)) {
				URX_SYNC();
				const int nb = qcount < 64 ? qcount : 64;
				uint32_t c_q = 0, c_db = 0;
				bool c_plus = true;
				const bool ok = lane < nb;
				if (ok) {
					const int pos = (qhead + lane) & 127;
					c_db = cq_db[pos];
					const uint32_t qp = cq_qp[pos];
					c_q = qp & 0x7FFFu; c_plus = (qp & 0x8000u) != 0;
				}
				qhead = (qhead + nb) & 127; qcount -= nb;
				URX_SYNC();
				const uint32_t dblo = c_db - c_q;
				int e_kind = 0, e_bst = 0, e_sp = 0, e_ep = 0, e_pen = 0;
				if (ok) {
					uint64_t mm[NCH];
					lane_mask(dblo, c_plus, mm);
					// lanes that can be neither a hit under the cap nor an HSP that counts do not walk (see search_se_kernel)
					const int nmis = mismatches_outside_seed<NCH>(mm, (int)c_q, W);
					const int floor2 = minhsp > best - 4 ? minhsp : best - 4;
					if (-P->mismatch_score * nmis <= maxPen || QL - nmis >= floor2) {
						xdrop_walk_lane<NCH>(mm, (int)c_q, W, QL, P->mismatch_score, P->xdrop, maxPen, e_bst, e_sp, e_ep, e_pen);
						if (e_sp == 0 && e_ep == QL - 1) e_kind = 1;
						else if (e_bst >= minhsp) e_kind = 2;
					}
				}
				// ordered part (see search_se_kernel): candidates that cannot change the state are dropped, up front and
				// after every change
				uint64_t todo = __ballot(e_kind != 0 && e_pen <= maxPen && !(e_kind == 2 && e_bst < best - 4) &&
				                         !overlaps_any_hit(dblo));
				while (todo) {
					const int t = __builtin_ctzll(todo);
					todo &= todo - 1;
					const uint32_t d = rdlane(dblo, t);
					if (overlaps_hit(d)) continue;
					if (rdlane(e_pen, t) > maxPen) continue;
					const int bst = rdlane(e_bst, t);
					const bool pl = rdlane((uint32_t)c_plus, t) != 0;
					const int hc0 = hitCount, mp0 = maxPen, b0 = best;
					if (rdlane(e_kind, t) == 1) add_hit(d, pl, bst, 0);
					else {
						const uint32_t sp = (uint32_t)rdlane(e_sp, t), ep = (uint32_t)rdlane(e_ep, t);
						add_hsp(sp, d + sp, pl, ep - sp + 1, bst);
						todo &= ~__ballot(e_kind == 2 && dblo == d && e_bst <= bst);  // see search_se_kernel
					}
					if (hitCount != hc0 || maxPen != mp0 || best != b0)
						todo &= __ballot(e_pen <= maxPen && !(e_kind == 2 && e_bst < best - 4) &&
						                 !(hitCount != hc0 && (dblo >> 6) == (d >> 6)));
				}
			}
		};
		for (int round = 0; round < 2; ++round) {
			for (int s = 0; s < 2; ++s) {
				for (int base = 0; base < pendCount[s]; base += 64) {
					const int i = base + lane;
					int K = 0;
					if (i < pendCount[s]) K = rowlen[s * QMAX + i];
					const int cnt = (round == 0 ? K <= 2 : K > 2) ? K : 0;
					const int inc = wave_prefix_sum(cnt);
					const int total = rdlane(inc, 63);
					if (total == 0) continue;
					URX_SYNC();
					pre[lane] = (uint16_t)(inc - cnt);
					if (lane == 0) pre[64] = (uint16_t)total;
					URX_SYNC();
					const uint32_t *rs0 = rowstore + ((size_t)(s * NCH + (base >> 6)) * PE_ROW_CAP) * 64;
					for (int gb = 0; gb < total; gb += 64) {
						const int g = gb + lane;
						uint32_t c_q = 0, c_db = 0;
						bool ok = false;
						if (g < total) {
							int lo = 0, hi = 63;  // row r with pre[r] <= g < pre[r + 1]; empty rows share a prefix value
							while (lo < hi) {
								const int mid = (lo + hi + 1) >> 1;
								if ((int)pre[mid] <= g) lo = mid;
								else hi = mid - 1;
							}
							const int k = g - (int)pre[lo];
							c_q = pend[s][base + lo];
							if (X->slot16) {
								const int g = s * NCH + (base >> 6);
								if (k == 0) c_db = rs_lds[(2 * NCH + g) * 64 + lo];
								else {
									const uint32_t x = rs_lds[g * 64 + lo];
									c_db = round == 0 ? x : X->rows[(size_t)x + (uint32_t)k];  // round 0: rows of at most two
								}
							} else if (X->rowinfo) {
								const int g = s * NCH + (base >> 6);
								c_db = k < 2 ? rowstore[(size_t)((k + 1) * 2 * NCH + g) * 64 + lo]
								             : X->rows[(size_t)rowstore[(size_t)g * 64 + lo] + (uint32_t)k];
							} else
								c_db = rs0[k * 64 + lo];
							ok = c_db >= c_q;
						}
						ok = ok && !overlaps_any_hit(c_db - c_q);
						const uint64_t m = __ballot(ok);
						if (ok) {
							const int pos = (qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & 127;
							cq_db[pos] = c_db;
							cq_qp[pos] = (uint16_t)(c_q | (s == 0 ? 0x8000u : 0u));
						}
						qcount += __builtin_popcountll(m);
						drain(false);
					}
				}
			}
		}
		drain(true);
		return dbg_cut != 43;
	}

	// extendscan.cpp:51-187
	__device__ __forceinline__ void extend_scan(uint32_t seedq, uint32_t seeddb, bool plus) {
		if (seeddb < seedq) return;
		const uint32_t dblo = seeddb - seedq;
		BitVec<NCH> mm;
		window_mask(dblo, plus, mm);
		int bst, sp, ep;
		if (!xdrop_walk(mm, (int)seedq, true, bst, sp, ep)) return;
		if (sp == 0 && ep == QL - 1) { add_hit(dblo, plus, bst, 0); return; }
		if (bst < 2 * W) return;
		const int k = add_hsp_scan((uint32_t)sp, dblo + (uint32_t)sp, plus, (uint32_t)(ep - sp + 1), bst);
		if (k >= 0) align_hsp(k);
	}

	// scanslots.cpp:7-62: every window k-mer against the read's slots at the first SCANK prime-stride positions
	__device__ __forceinline__ void scan_slots(uint32_t dblo, uint32_t seglen, bool plus) {
		if (QL <= 4 * W) return;
		const int s = plus ? 0 : 1;
		uint64_t qslot[SCANK];
		uint32_t qposk[SCANK];
#pragma unroll
		for (uint32_t k = 0; k < SCANK; ++k) {
			qposk[k] = (k * PRIME_STRIDE) % (uint32_t)nwords;
			qslot[k] = uni64(slot_from_planes<NCH + 1>(*X, (lds_ptr<const uint64_t>)&L->kpl[0][0], (uint32_t)nwords, s, qposk[k]));
		}
		const uint64_t wmask = X->shiftMask;
		const uint64_t wbits = (W >= 32) ? 0xFFFFFFFFull : ((1ull << W) - 1ull);
		// Letter planes of 64 window positions as three ballots (low bit, high bit, not-ACGT), as in seed_probe_kernel:
		// lane p cuts the W bits of its k-mer out of two adjacent plane words, so a chunk costs one byte load per lane.
		auto planes = [&](uint32_t start, uint64_t &lo, uint64_t &hi, uint64_t &inv) {
			const uint32_t q = start + lane;
			uint32_t L = 4;
			if (q < seglen) L = letter_of(gseq[dblo + q]);
			lo = __ballot(L & 1u); hi = __ballot((L >> 1) & 1u); inv = __ballot(L > 3u);
		};
		uint64_t lo0, hi0, inv0, lo1, hi1, inv1;
		planes(0, lo0, hi0, inv0);
		// lane handles window start positions base+lane; a k-mer is valid iff its W letters are all ACGT
		for (uint32_t base = 0; base + (uint32_t)W <= seglen; base += 64) {
			planes(base + 64, lo1, hi1, inv1);
			const uint32_t p = base + lane;
			uint64_t flo = lo0 >> lane, fhi = hi0 >> lane, finv = inv0 >> lane;
			if (lane) { flo |= lo1 << (64 - lane); fhi |= hi1 << (64 - lane); finv |= inv1 << (64 - lane); }
			flo &= wbits; fhi &= wbits; finv &= wbits;
			const bool valid = p + (uint32_t)W <= seglen && finv == 0;
			// first base = most significant letter
			const uint64_t word = spread32(__brevll(flo) >> (64 - W)) | (spread32(__brevll(fhi) >> (64 - W)) << 1);
			lo0 = lo1; hi0 = hi1; inv0 = inv1;
			uint64_t slot = ~0ull;
			if (valid) slot = mod_slots(murmur64(word & wmask), X->slotCount, X->slotMagic);
			uint32_t hitmask = 0;
#pragma unroll
			for (uint32_t k = 0; k < SCANK; ++k)
				if (valid && slot == qslot[k]) hitmask |= (1u << k);
			uint64_t any = __ballot(hitmask != 0);
			while (any) {  // window positions in ascending order, k ascending inside (scanslots.cpp:50-59)
				const int l = __builtin_ctzll(any);
				any &= any - 1;
				const uint32_t hm = rdlane(hitmask, l);
#pragma unroll
				for (uint32_t k = 0; k < SCANK; ++k)
					if (hm & (1u << k)) extend_scan(qposk[k], dblo + base + (uint32_t)l, plus);
			}
		}
	}

	// scan.cpp:14-39
	__device__ __forceinline__ void scan(uint32_t dbpos, uint32_t seglen, bool plus, bool dovit) {
		const int savedMaxPen = maxPen;
		const int savedHits = hitCount;
		maxPen = 130;
		scan_slots(dbpos, seglen, plus);
		maxPen = savedMaxPen;
		if (hitCount > savedHits) return;
		if (!dovit) return;
		RevOps R;
		R.ops = from_lds(ropsL);
		uint32_t vst = 0;
		// whole read against the window: a band far wider than a wavefront -> wide path (B read from global memory)
		const float score = viterbi_wave(VPar(*P), from_lds(&sQ[plus ? 0 : 1][0]), QL, gseq + dbpos, (int)seglen, true, true, tb, TB_ROWS8, ws, R, vst, lane,
		                                 -3.0e38f, nullptr, from_lds(wide_lds), wide_lds_dwords);
		status |= vst;
		if (vst) return;
		if ((double)score >= (double)QL / 3.0) {
			int n = R.n, nI = 0, r0 = 0;
			if (n > 0 && (ropsL[n - 1] & 3u) == OP_I) { nI = (int)(ropsL[n - 1] >> 2); --n; }  // TrimLeftIs
			if (n > 1 && (ropsL[0] & 3u) == OP_I) r0 = 1;                                     // TrimRightIs
			const int nc = n - r0;
			if (nc > URMAPX_MAX_PATH_OPS) { status |= URMAPX_ST_PATH_OVERFLOW; return; }
			URX_SYNC();
			for (int t = lane; t < nc; t += 64) cand[t] = ropsL[n - 1 - t];
			URX_SYNC();
			add_hit(dbpos + (uint32_t)nI, plus, (int)score, nc);
		}
	}
};
#undef lane
#undef QL
#undef nwords
#undef pendCount
#undef hitCount
#undef hspCount
#undef topHit
#undef maxPen
#undef best
#undef second
#undef bestHSP
#undef mapq
#undef status
#undef W
#undef wide_lds_dwords
#undef hit_cap
#undef hit_wsh
#undef hsp_lds
#undef dbg_cut
#undef q_other
#undef hsp_ovf
#undef hit_paths
#undef hit_tail
#undef hot
#undef sQ
#undef qpl
#undef hit_db
#undef hit_sp
#undef hit_nops
#undef hsp_db
#undef hsp_ql
#undef hsp_sf
#undef pend

// Waves per SIMD the register allocation aims at.  The pair kernel waits on memory 70 % of its wave cycles and issues
// instructions in 40 % of its SIMD cycles at two waves per SIMD (profiles/r3/pmc_sq_pe.json): it is bound by latency, and
// a third wave hides more of it than the extra spills cost -- 2 waves (228 VGPRs): 40.4 ms per 1 M reads, 3 waves (168
// VGPRs): 34.6 ms; asking for 4 (128 VGPRs) the compiler settles at 194 = 2 waves again.  LDS allows 12 blocks per CU
// for reads <= 192 (13.2 KB each: the DP trace lives in global scratch, the seed-stage arrays share one area by lifetime).
#ifndef URX_PE_WAVES
#define URX_PE_WAVES(NCH) ((NCH) <= 3 ? (URX_PE_DIET ? 4 : 3) : 2)
#endif
template <int NCH, int TIER>
__global__ __launch_bounds__(64, TIER == 2 ? 1 : URX_PE_WAVES(NCH)) void search_pe_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint32_t npairs,
                                                       urmapx_result *__restrict__ results,
                                                       urmapx_path_op *__restrict__ path_ops, uint32_t *path_used,
                                                       uint8_t *scratch, size_t scratch_stride,
                                                       const uint8_t *__restrict__ g_seq, const uint8_t *__restrict__ g_blob,
                                                       const uint4 *__restrict__ g_seqp,
                                                       int veryfast, uint32_t *ticket, urmapx_pair_info *pair_info,
                                                       int hsp_lds_cap, uint32_t *ovf_list, uint32_t *ovf_next, uint2 *hsp_ovf_base, uint32_t hsp_area_blocks) {
	constexpr bool OVF = TIER > 0;
	using M = Mate<NCH, TIER>;
	constexpr int QMAX = M::QMAX;
	// LDS per block decides how many pairs a CU keeps in flight, so arrays share memory by lifetime:
	//   probe results staged for the seed enumeration (s_tal, s_pos)       on  the DP trace buffer (idle until AlignHSP)
	//   flank run buffers, candidate path, target window (AlignHSP / Scan)  on  seed_res  (seeds are dead by then)
	//   pending-stage row lengths and prefix                               on  seed_q
	//   the pending stage's candidate queue                                 on  seed_db
	__shared__ MateLds<NCH, TIER> ml[2];
	static_assert(URMAPX_MAX_PATH_OPS <= 127, "hit_nops is a byte (seven bits when it rides in hit_sp)");
	static_assert(!M::TAIL || M::HITS_LDS == 64, "the tail begins at hit 64");
	// Both1 seed lists of the two mates in enumeration order: qpos | plus << 15, db position
	// later passes: 2 * (QMAX - 20) >= 2 * (QMAX - W + 1) for the word lengths in use (W >= 21).  First pass: QMAX -- a mate returns a
	// seed only where the diagonal changes (getseed.cpp:60-66), a handful per read; a pair with more goes to the second pass
	// (LDS per block decides how many pairs a CU keeps in flight)
	constexpr int SEED_CAP = TIER == 0 ? (URX_PE_DIET && NCH <= 3 ? (URX_PE_PACK ? 104 : 136) : QMAX) : 2 * (QMAX - 20);
	__shared__ __attribute__((aligned(16))) uint16_t seed_q[2][SEED_CAP];
	__shared__ __attribute__((aligned(16))) uint32_t seed_db[2][SEED_CAP];
	// cached ExtendPen outcome of every seed (see extend_pen_cached); bit 15 of seed_pen = "already extended once"
	// one area, three lives: (1) the probe results staged for the seed enumeration (s_tal, s_pos); (2) the cached
	// ExtendPen outcomes of the seeds (seed_res, seed_pen); (3) AlignHSP's run buffers, candidate path and target window,
	// and behind them the per-row arrays of the wide-band DP
	constexpr int SEED_AREA_DWORDS = 3 * SEED_CAP > 5 * QMAX ? 3 * SEED_CAP : 5 * QMAX;
	__shared__ __attribute__((aligned(16))) uint32_t seed_area[SEED_AREA_DWORDS];
	uint32_t (*const seed_res)[SEED_CAP] = reinterpret_cast<uint32_t (*)[SEED_CAP]>(seed_area);
	uint16_t (*const seed_pen)[SEED_CAP] = reinterpret_cast<uint16_t (*)[SEED_CAP]>(seed_area + 2 * SEED_CAP);
	static_assert(4 * QMAX + 16 * QMAX <= sizeof(seed_area), "alias");
	uint8_t (*const s_tal)[2][QMAX] = reinterpret_cast<uint8_t (*)[2][QMAX]>(seed_area);                 // [mate][strand][qpos]
	uint32_t (*const s_pos)[2][QMAX] = reinterpret_cast<uint32_t (*)[2][QMAX]>(seed_area + QMAX);        // after the 4*QMAX tally bytes
	constexpr int ALIGN_BYTES = ((2 * OPS_CAP + URMAPX_MAX_PATH_OPS) * 2 + QMAX + 64 + 15) & ~15;
	static_assert(ALIGN_BYTES + 3 * QMAX * 4 <= sizeof(seed_area), "alias");
	uint16_t *const ropsL = reinterpret_cast<uint16_t *>(seed_area), *const ropsR = ropsL + OPS_CAP, *const cand = ropsR + OPS_CAP;
	uint8_t *const sT = reinterpret_cast<uint8_t *>(cand + URMAPX_MAX_PATH_OPS);
	static_assert((2 * OPS_CAP + URMAPX_MAX_PATH_OPS) * 2 >= 64, "AlignHSP's window lies band_radius + 1 bytes or more inside seed_area (viterbi_dev.h: B_LDS)");
	uint32_t *const wide_lds = seed_area + ALIGN_BYTES / 4;
	constexpr int WIDE_LDS_DWORDS = (int)(sizeof(seed_area) - ALIGN_BYTES) / 4;
	// (diet 2: the prefix array lies at the start of seed_area instead -- the pending stage's chain rows run while AlignHSP's buffers there are idle)
	constexpr bool PRE_IN_AREA = URX_PE_PACK && TIER == 0 && NCH <= 3;
	static_assert(2 * QMAX + (PRE_IN_AREA ? 0 : 66 * 2) <= sizeof(seed_q), "alias");
	uint8_t *const rowlen = reinterpret_cast<uint8_t *>(&seed_q[0][0]);  // shared by the two mates: SearchPE_Pending runs on one mate at a time
	uint16_t *const pre = PRE_IN_AREA ? reinterpret_cast<uint16_t *>(seed_area) : reinterpret_cast<uint16_t *>(rowlen + 2 * QMAX);
	static_assert(128 * 6 <= sizeof(seed_db), "alias");
	uint32_t *const cq_db = &seed_db[0][0];  // pending stage only
	uint16_t *const cq_qp = reinterpret_cast<uint16_t *>(cq_db + 128);

	const int lane = threadIdx.x;
	const int W = (int)X.W;
	const int dbg_stop = veryfast >> 8;  // diagnostic only (URMAPX_DEBUG_STOP_PE): results are NOT the reference's
	veryfast &= 1;
	uint8_t *sc = scratch + (size_t)blockIdx.x * scratch_stride;
	// A Mate is the shared pointers below plus the LDS address of the mate's own object: `mate(a)` with a run-time a is an
	// address computation (round 3 kept an array of two objects, which the run-time index forced into private memory)
	M mshared;
	mshared.X = &X; mshared.P = &P; mshared.gseq = g_seq; mshared.gblob = g_blob; mshared.gseqp = g_seqp;
	mshared.sT = to_lds(sT); mshared.tb = reinterpret_cast<uint32_t *>(sc + pe_tb_offset(QMAX)); mshared.wide_lds = to_lds(wide_lds);
	mshared.ropsL = to_lds(ropsL); mshared.ropsR = to_lds(ropsR); mshared.cand = to_lds(cand);
	mshared.rowlen = to_lds(rowlen); mshared.pre = to_lds(pre); mshared.cq_db = to_lds(cq_db); mshared.cq_qp = to_lds(cq_qp);
	mshared.rowstore = reinterpret_cast<uint32_t *>(sc + pe_rowstore_offset(QMAX));
	static_assert(sizeof(seed_area) >= 256 + 2 * 2 * NCH * 64 * 4, "the pending stage's row store (slot16 layout) behind the prefix array");
	mshared.rs_lds = to_lds(seed_area + 64);
	mshared.ws.carve(sc + (size_t)2 * PE_HIT_CAP * PE_HITW1 * URMAPX_MAX_PATH_OPS * 2, QMAX, PE_SCAN_SEG + 2 * QMAX + 64);
	mshared.L = to_lds(&ml[0]);
	auto mate = [&](int a) -> M { M r = mshared; r.L = to_lds(&ml[a]); return r; };
#define hot(a) ml[a].hot
	for (int a = 0; a < 2; ++a) {
		hot(a).W = W; hot(a).q_other = false; hot(a).wide_lds_dwords = WIDE_LDS_DWORDS;
		hot(a).hit_paths = OVF ? reinterpret_cast<urmapx_path_op *>(hsp_ovf_base + (size_t)hsp_area_blocks * 2 * PE_HSP_OVF_CAP) +
		                           (TIER == 2 ? (size_t)PE_OVF_BLOCKS * 2 * PE_HIT_CAP * 4 * URMAPX_MAX_PATH_OPS : (size_t)0) +
		                           ((size_t)blockIdx.x * 2 + a) * PE_HIT_CAP * M::HITW * URMAPX_MAX_PATH_OPS
		                     : reinterpret_cast<urmapx_path_op *>(sc) + (size_t)a * PE_HIT_CAP * PE_HITW1 * URMAPX_MAX_PATH_OPS;
#if !URX_PE_PACK
		hot(a).hit_tail = reinterpret_cast<uint32_t *>(sc + pe_tail_offset(QMAX)) + a * 192;
#endif
		hot(a).hit_cap = (hsp_lds_cap >= 64 && hsp_lds_cap <= PE_HSP_CAP) ? (hsp_lds_cap & ~63) / 4 : PE_HIT_CAP * PE_HITW1;
		hot(a).hit_wsh = (hsp_lds_cap >= 64 && hsp_lds_cap <= PE_HSP_CAP) ? 4 : 6;
		hot(a).hsp_lds = (hsp_lds_cap >= 64 && hsp_lds_cap <= PE_HSP_CAP) ? (hsp_lds_cap & ~63) : PE_HSP_CAP;
		hot(a).hsp_ovf = hsp_ovf_base + ((size_t)blockIdx.x * 2 + a) * PE_HSP_OVF_CAP;
		hot(a).dbg_cut = dbg_stop;
	}

	// pairs are handed out by a ticket counter (heavy-tailed cost per pair: the rescue DP), PE_TICKET_CHUNK per ticket
	// (same-address atomics retire at ~88 M/s), see search_se_kernel
	if constexpr (OVF) npairs = ovf_list[0];  // second pass: the pairs the first pass flagged (usually none)
	uint32_t pr_next = 0, pr_end = 0;
	for (;;) {
		if (pr_next == pr_end) {
			constexpr uint32_t CHUNK = OVF ? 1u : (uint32_t)PE_TICKET_CHUNK;
			pr_next = uni(atomicAdd(ticket, lane == 0 ? CHUNK : 0u));
			if (pr_next >= npairs) break;
			pr_end = pr_next + CHUNK < npairs ? pr_next + CHUNK : npairs;
		}
		uint32_t pr = pr_next++;
		if constexpr (OVF) pr = ovf_list[1 + pr];
		urmapx_result res[2];
		bool bad = false;
		for (int a = 0; a < 2; ++a) {
			const uint32_t r = 2 * pr + a;
			const uint64_t off = offs[r];
			const int QL = (int)(offs[r + 1] - off);
			res[a].dbpos = 0xFFFFFFFFu; res[a].seq_index = 0xFFFFFFFFu; res[a].coord = 0xFFFFFFFFu;
			res[a].score = 0; res[a].second = 0; res[a].mapq = 0; res[a].plus = 0; res[a].exit_phase = 0; res[a].status = 0;
			res[a].hit_count = 0; res[a].path_nops = 0; res[a].path_off = 0;
			if (QL < W || QL > QMAX || W > 32 || X.maxIx > 32 || QL - (W - 1) > 256) bad = true;  // pending positions are bytes
			// (diet 2: this pass's pending lists hold QMAX - 16 positions -- enough for every W >= 17; an index with shorter words sends such a pair to the general kernel)
			if (MateLds<NCH, TIER>::PACK && QL - (W - 1) > QMAX - 16) bad = true;
			hot(a).QL = QL; hot(a).nwords = QL - (W - 1);
		}
		if (bad) {
			for (int a = 0; a < 2; ++a) { res[a].status = URMAPX_ST_BAD_LENGTH; if (lane == 0) results[2 * pr + a] = res[a]; }
			continue;
		}
		// seed + probe of both mates (round 3: no launch of its own -- this kernel waits on latency with half of its issue
		// slots idle, so the hashing is free and the 2 x 254 slot gathers overlap the other pairs of the CU).  Round 4: the
		// entries go to LDS only, in every pass (the later passes used to read what the first had written to HBM).
		lds_sync();
		{
			const uint64_t poff[2] = {offs[2 * pr], offs[2 * pr + 1]};
			const uint32_t pql[2] = {(uint32_t)hot(0).QL, (uint32_t)hot(1).QL};
			probe_pair<NCH, QMAX>(X, bases, poff, pql, lane, s_tal, s_pos, to_lds(&ml[0].kpl[0][0]), to_lds(&ml[1].kpl[0][0]));
		}
		// ---- InitPE x2 (state1.cpp:95-127) ----
		uint32_t qch[2][2][NCH];  // this lane's bytes of the two mates, both strands (locals: every index below is a constant after unrolling)
		bool acgt[2];
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			const uint8_t *q = bases + offs[2 * pr + a];
			const int QL = hot(a).QL;
			// (dev_common.h: URX_ACGT_FAST) a mate of upper-case ACGT only takes the four-instruction complement and code
			bool plain = true;
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int p = 64 * c + lane;
				uint32_t cp = 0, cmr = 0;
				if (p < QL) { cp = q[p]; cmr = q[QL - 1 - p]; }
				plain = plain && (p >= QL || is_upper_acgt(cp));
				qch[a][0][c] = cp; qch[a][1][c] = cmr;
			}
			acgt[a] = URX_ACGT_FAST && __ballot(!plain) == 0;  // wave-uniform
#pragma unroll
			for (int c = 0; c < NCH; ++c) {
				const int p = 64 * c + lane;
				if (p < QL) {
					const uint32_t cm = acgt[a] ? comp_char_acgt(qch[a][1][c]) : comp_char(qch[a][1][c]);
					qch[a][1][c] = cm;
					ml[a].sQ[0][p] = (uint8_t)qch[a][0][c]; ml[a].sQ[1][p] = (uint8_t)cm;
				}
			}
			hot(a).pendCount[0] = hot(a).pendCount[1] = 0;
			hot(a).hitCount = 0; hot(a).hspCount = 0; hot(a).topHit = -1;
			hot(a).maxPen = P.max_penalty; hot(a).best = 0; hot(a).second = 0; hot(a).bestHSP = 0;
			hot(a).mapq = 0xFFFFFFFFu; hot(a).status = 0;
		}
		lds_sync();
		// both strands of both mates as bit planes (ExtendPen's windows are read from the packed sequence store)
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			uint64_t oth = 0;
#pragma unroll
			for (int st = 0; st < 2; ++st) {
#pragma unroll
				for (int c = 0; c < NCH; ++c) {
					if (64 * c < hot(a).QL) {
						const int p = 64 * c + lane;
						uint64_t b0, b1, b2 = 0, b3 = 0;
						if (acgt[a]) {  // codes 0..3: the two upper planes are empty
							const uint32_t code = p < hot(a).QL ? seq_code_acgt(qch[a][st][c]) : 0u;
							b0 = __ballot(code & 1u); b1 = __ballot(code & 2u);
						} else {
							const uint32_t code = p < hot(a).QL ? seq_code(qch[a][st][c], SEQ_CODE_QOTHER) : 0u;
							b0 = __ballot(code & 1u); b1 = __ballot(code & 2u); b2 = __ballot(code & 4u); b3 = __ballot(code & 8u);
							oth |= __ballot(code == SEQ_CODE_QOTHER);
						}
						if (lane < 2) {
							const int shh = 32 * lane;
							ml[a].qpl[st][2 * c + lane] = make_uint4((uint32_t)(b0 >> shh), (uint32_t)(b1 >> shh), (uint32_t)(b2 >> shh), (uint32_t)(b3 >> shh));
						}
					}
				}
			}
			hot(a).q_other = oth != 0;
		}
		lds_sync();

		// ---- seed enumeration of both mates (GetFirstBoth1Seed / GetNextBoth1Seed) ----
		// State independent, so it is run to the end up front; the pending lists are cut back to the point the
		// pairing loop reached if that loop returns early (they are not used then anyway).
		int nseed[2];
		// 64 enumeration steps k at a time, one per lane.  The only order dependence is "skip a seed on the diagonal of
		// the last seed returned" (getseed.cpp:60-66,118-124), and because a skipped seed has that very diagonal, it is
		// the same as "skip a seed on the diagonal of the previous BOTH1 candidate": a neighbour comparison.
		for (int a = 0; a < 2; ++a) {
			const int QWC = hot(a).nwords;
			int ns = 0, np = 0, nm = 0;
			bool have = false;
			uint32_t lastDiag = 0;
			for (int kb = 0; kb < QWC; kb += 64) {
				const int k = kb + lane;
				const bool in = k < QWC;
				const uint32_t qpos = in ? ((uint32_t)k * PRIME_STRIDE) % (uint32_t)QWC : 0u;
				const uint32_t Tp = in ? s_tal[a][0][qpos] : 0u, Tm = in ? s_tal[a][1][qpos] : 0u;
				const bool candP = Tp == TALLY_BOTH1, candM = Tm == TALLY_BOTH1;
				const uint32_t dbP = s_pos[a][0][qpos], dbM = s_pos[a][1][qpos];
				const uint32_t diagP = dbP - qpos, diagM = dbM - qpos;
				// diagonal of the last candidate before this lane's
				const uint64_t anyc = __ballot(candP || candM);
				const uint32_t lastHere = candM ? diagM : diagP;
				const uint64_t below = anyc & ((1ull << lane) - 1ull);
				const int src = below ? 63 - __builtin_clzll(below) : 0;
				const uint32_t fromLane = (uint32_t)__shfl((int)lastHere, src);
				const bool havePrev = below ? true : have;
				const uint32_t prev = below ? fromLane : lastDiag;
				const bool retP = candP && !(havePrev && diagP == prev);
				const bool havePrevM = candP || havePrev;
				const uint32_t prevM = candP ? diagP : prev;
				const bool retM = candM && !(havePrevM && diagM == prevM);
				// pending lists: "mine" slots that are not BOTH1; the minus slot is not looked at again when the plus seed of
				// this step was returned (the re-check branch of GetNextBoth1Seed), except that a BOTH1 minus seed skipped for
				// its diagonal in that branch is pushed (getseed.cpp:96-138)
				const bool pendP = (Tp & TALLY_MY_BIT) != 0 && !candP;
				const bool pendM = (Tm & TALLY_MY_BIT) != 0 && ((!candM && !retP) || (candM && !retM && retP));
				const uint64_t mp = __ballot(pendP), mmn = __ballot(pendM);
				const uint64_t lt = (1ull << lane) - 1ull;
				if (pendP) ml[a].pend[0][np + __builtin_popcountll(mp & lt)] = (uint8_t)qpos;
				if (pendM) ml[a].pend[1][nm + __builtin_popcountll(mmn & lt)] = (uint8_t)qpos;
				np += __builtin_popcountll(mp); nm += __builtin_popcountll(mmn);
				const int nret = (retP ? 1 : 0) + (retM ? 1 : 0);
				const int inc = wave_prefix_sum(nret);
				int at = ns + inc - nret;
				if (retP && at < SEED_CAP) { seed_q[a][at] = (uint16_t)(qpos | 0x8000u); seed_db[a][at] = dbP; }
				if (retP) ++at;
				if (retM && at < SEED_CAP) { seed_q[a][at] = (uint16_t)qpos; seed_db[a][at] = dbM; }
				ns += rdlane(inc, 63);
				if (anyc) {
					have = true;
					lastDiag = (uint32_t)__shfl((int)lastHere, 63 - __builtin_clzll(anyc));
				}
			}
			if (ns > SEED_CAP) { ns = SEED_CAP; hot(a).status |= URMAPX_ST_HSP_OVERFLOW; }
			nseed[a] = ns;
			hot(a).pendCount[0] = np; hot(a).pendCount[1] = nm;
		}
		URX_SYNC();

		// ---- seed gather: the ExtendPen outcome of every seed, one seed per lane ----
		for (int a = 0; a < 2; ++a) {
			const int QL = hot(a).QL;
			const int minhsp = (int)((uint32_t)P.min_hsp_score_pct * (uint32_t)QL / 100.0);
			for (int base = 0; base < nseed[a]; base += 64) {
				const int i = base + lane;
				uint32_t res = 0;
				int pen = 0;
				if (i < nseed[a]) {
					const uint32_t q = seed_q[a][i] & 0x7FFFu, db = seed_db[a][i];
					const bool plus = (seed_q[a][i] & 0x8000u) != 0;
					if (db >= q) {
						uint64_t mm[NCH];
						mate(a).lane_mask(db - q, plus, mm);
						int bst, sp, ep;
						xdrop_walk_lane<NCH>(mm, (int)q, W, QL, P.mismatch_score, P.xdrop, P.max_penalty, bst, sp, ep, pen);  // cached for any later cap: bounded by the initial one
						const uint32_t kind = (sp == 0 && ep == QL - 1) ? 1u : (bst >= minhsp ? 2u : 0u);
						res = (kind << 27) | ((uint32_t)ep << 18) | ((uint32_t)sp << 9) | (uint32_t)bst;
					}
					seed_res[a][i] = res;
					seed_pen[a][i] = (uint16_t)pen;
				}
			}
		}
		URX_SYNC();
		if (dbg_stop == 1) continue;
		// ExtendPen on seed i of mate a, in order (state changing), through the cache
		auto extend_seed = [&](int a, int i) -> int {
			const uint32_t sq = seed_q[a][i];
			const uint32_t pn = seed_pen[a][i];
			const int r = mate(a).extend_pen_cached(sq & 0x7FFFu, seed_db[a][i], (sq & 0x8000u) != 0, seed_res[a][i], (int)(pn & 0x7FFFu));
			if ((pn & 0x8000u) == 0) {
				URX_SYNC();
				if (lane == 0) seed_pen[a][i] = (uint16_t)(pn | 0x8000u);
				URX_SYNC();
			}
			return r;
		};
		// ExtendBoth1Pair4 extends the partner seed on the strand OPPOSITE to the new seed's, whatever strand the partner
		// seed itself was found on (search2m4.cpp:189-208).  The cache holds a seed's outcome on its own strand; the
		// rare same-strand partner is extended the slow way (one window, 64 bases per compare).
		auto extend_seed_as = [&](int a, int i, bool plus_req) -> int {
			const uint32_t sq = seed_q[a][i];
			if (((sq & 0x8000u) != 0) == plus_req) return extend_seed(a, i);
			return mate(a).extend_pen(sq & 0x7FFFu, seed_db[a][i], plus_req);
		};
		// Lanes = seeds base.. of mate a: true where a repeated ExtendPen on that seed is certain to return <= 0 without
		// changing anything: it has been extended before, and it is not a full-length hit, or its penalty exceeds the
		// cap (which only falls here), or it lies on a found hit's diagonal block (hits are never removed).  An HSP seed
		// extended again re-offers the same HSP (no change); a hit seed extended again either overlaps its own hit or
		// was rejected by AddHitX for its score, which stays rejected because the best score only rises.
		auto settled = [&](int a, int base, int n, bool plus_req) -> uint64_t {
			const int i = base + lane;
			bool st = false;
			if (i < n) {
				const uint32_t pn = seed_pen[a][i], res = seed_res[a][i];
				const uint32_t q = seed_q[a][i] & 0x7FFFu, db = seed_db[a][i];
				if ((pn & 0x8000u) && ((seed_q[a][i] & 0x8000u) != 0) == plus_req)
					st = db < q || (res >> 27) != 1u || (int)(pn & 0x7FFFu) > hot(a).maxPen || mate(a).overlaps_any_hit(db - q);
			}
			return __ballot(st);
		};

		// ---- Search4 pairing loop (search2m4.cpp:71-143) ----
		const int QLf = hot(0).QL, QLr = hot(1).QL;
		const int64_t QL2 = (int64_t)((QLf + QLr) / 2);
		const int termPair = QLf + QLr + 5 * P.mismatch_score;
		bool done = false;
		{
			const int steps = max(nseed[0], nseed[1]);
			// The template-length test of a new seed against every seed of the other mate is done 64 seeds per step
			// (ballot); only the few that pass go through ExtendBoth1Pair4, in list order.
			auto near_mask = [&](int other, int base, int n, uint32_t db) -> uint64_t {
				const int i = base + lane;
				bool ok = false;
				if (i < n) {
					int64_t d = (int64_t)db - (int64_t)seed_db[other][i];
					if (d < 0) d = -d;
					ok = d + QL2 <= MAX_TL;
				}
				return __ballot(ok);
			};
			for (int t = 0; t < steps && !done; ++t) {
				if (t < nseed[0]) {
					// ExtendBoth1Pair4(new forward seed t, earlier reverse seed i): the forward seed is extended first,
					// every time.  Once that returns <= 0 it does so for the rest of this step (settled), so the step ends.
					const uint32_t dbf = seed_db[0][t];
					const bool plusf = (seed_q[0][t] & 0x8000u) != 0;
					const int nr = min(t, nseed[1]);
					bool fwd_dead = false;
					for (int base = 0; base < nr && !done && !fwd_dead; base += 64) {
						uint64_t mk = near_mask(1, base, nr, dbf);
						while (mk && !done) {
							const int i = base + __builtin_ctzll(mk);
							mk &= mk - 1;
							const int fs = extend_seed(0, t);
							if (fs <= 0) { fwd_dead = true; break; }
							const int rs = extend_seed_as(1, i, !plusf);
							if (rs <= 0) continue;
							if (fs + rs < termPair) continue;
							hot(0).mapq = 40; hot(1).mapq = 40; done = true;
						}
					}
				}
				if (t < nseed[1] && !done) {
					// ExtendBoth1Pair4(earlier forward seed i, new reverse seed t): forward seeds that are settled are
					// skipped (their ExtendPen returns <= 0 and the pair is dropped before the reverse seed is touched)
					const uint32_t dbr = seed_db[1][t];
					const bool plusr = (seed_q[1][t] & 0x8000u) != 0;
					const int nf = min(t + 1, nseed[0]);
					for (int base = 0; base < nf && !done; base += 64) {
						uint64_t mk = near_mask(0, base, nf, dbr);
						if (mk) mk &= ~settled(0, base, nf, !plusr);
						while (mk && !done) {
							const int i = base + __builtin_ctzll(mk);
							mk &= mk - 1;
							const int fs = extend_seed_as(0, i, !plusr);
							if (fs <= 0) continue;
							const int rs = extend_seed(1, t);
							if (rs <= 0) continue;
							if (fs + rs < termPair) continue;
							hot(0).mapq = 40; hot(1).mapq = 40; done = true;
						}
					}
				}
			}
		}
		int npairs_found = 0, bestPairScore = -1, secondPairScore = -1, bestPairIndex = -1, secondPairIndex = -1;
		// the reference keeps every pair (state2.cpp:20-85) and looks only at the best and the second best afterwards: their
		// hit indexes are kept as the indexes change hands, so the pair list needs no storage (and has no capacity)
		int bestF = -1, bestR = -1, secF = -1, secR = -1;
		int secondHit[2] = {-1, -1};  // m_SecondHit of the mates (set by AdjustTopHitsAndMapqs only)
		if (dbg_stop == 2) done = true;
		bool vf_pending = false;
		if (!done) {
			// all collected seeds, each mate (search2m4.cpp:145-158)
			// (a seed extended before changes nothing when extended again, see settled(); only the others are visited)
			for (int a = 0; a < 2; ++a)
				for (int base = 0; base < nseed[a]; base += 64) {
					const int i0 = base + lane;
					uint64_t mk = __ballot(i0 < nseed[a] && (seed_pen[a][i0 < nseed[a] ? i0 : 0] & 0x8000u) == 0);
					while (mk) {
						const int i = base + __builtin_ctzll(mk);
						mk &= mk - 1;
						extend_seed(a, i);
					}
				}
			if (veryfast) {
				// Search5 (search2m5.cpp:112-127): no 90 % shortcut and no pair stage; each mate finishes on its own
				vf_pending = true;
			} else if (hot(0).best >= (QLf * 9) / 10 && hot(1).best >= (QLr * 9) / 10 && hot(0).topHit >= 0 && hot(1).topHit >= 0) {
				int64_t d = (int64_t)mate(0).hdb(hot(0).topHit) - (int64_t)mate(1).hdb(hot(1).topHit);
				if (d < 0) d = -d;
				if (d + QL2 <= MAX_TL) { hot(0).mapq = 40; hot(1).mapq = 40; done = true; }
			}
		}
		if (dbg_stop == 3 && !vf_pending) done = true;
		if (!done) {
			// SearchPE_Pending of the forward mate, then of the reverse mate: one copy of the code, the mate is an LDS address
#pragma unroll 1
			for (int a = 0; a < 2; ++a) mate(a).search_pending();
		}
		if (vf_pending) done = true;
		if (!done) {
			if (dbg_stop == 4 || (dbg_stop >= 41 && dbg_stop <= 43)) goto pe_output;
			// FindPairs (state2.cpp:20-85), ScanPair if there is none (state2.cpp:87-137), FindPairs again
			for (int attempt = 0; attempt < 2; ++attempt) {
				npairs_found = 0; bestPairScore = -1; secondPairScore = -1; bestPairIndex = -1; secondPairIndex = -1;
				bestF = bestR = secF = secR = -1;
				for (int i = 0; i < hot(0).hitCount; ++i) {
					const uint32_t spf = mate(0).hsp_of(i);
					const int sf = (int)(spf >> 1);
					if (sf < hot(0).second - 12) continue;
					const int64_t dbf = (int64_t)mate(0).hdb(i);
					for (int j = 0; j < hot(1).hitCount; ++j) {
						const uint32_t spr = mate(1).hsp_of(j);
						const int sr = (int)(spr >> 1);
						if (sr < hot(1).second - 12) continue;
						int64_t d = dbf - (int64_t)mate(1).hdb(j);
						if (d < 0) d = -d;
						if (d + QL2 > 1000) continue;
						if ((spf & 1u) == (spr & 1u)) continue;
						const int total = sf + sr;
						if (total > bestPairScore) {
							secondPairIndex = bestPairIndex; secondPairScore = bestPairScore; secF = bestF; secR = bestR;
							bestPairScore = total; bestPairIndex = npairs_found; bestF = i; bestR = j;
						} else if (total == bestPairScore) { secondPairIndex = npairs_found; secondPairScore = bestPairScore; secF = i; secR = j; }
						else if (total > secondPairScore) { secondPairIndex = bestPairIndex; secondPairScore = total; secF = bestF; secR = bestR; }  // sic, state2.cpp:74
						++npairs_found;
					}
				}
				URX_SYNC();
				if (npairs_found > 0 || attempt == 1) break;
				// ScanPair
				const bool dovitF = (int)hot(0).mapq >= 10 && dbg_stop != 5, dovitR = (int)hot(1).mapq >= 10 && dbg_stop != 5;  // 5: diagnostic, no rescue DP
				const int hcf = hot(0).hitCount, hcr = hot(1).hitCount;
				// the forward mate's hits scanned for on the reverse mate, then the other way round (state2.cpp:100-136; both use
				// the FORWARD read's length for the minus-strand segment, as the reference does): one call site, so Scan and
				// everything under it exists once
#pragma unroll 1
				for (int dir = 0; dir < 2; ++dir) {
					const int hc = dir == 0 ? hcf : hcr;
					const bool dovit = dir == 0 ? dovitF : dovitR;
					for (int i = 0; i < hc; ++i) {
						const uint32_t sp = mate(dir).hsp_of(i);
						if ((int)(sp >> 1) < hot(dir).second) continue;
						const uint32_t db = mate(dir).hdb(i);
						uint32_t sdb = db, slen = PE_SCAN_SEG;
						bool splus = false;
						if ((sp & 1u) == 0) {
							if (db < (uint32_t)PE_SCAN_SEG) continue;
							sdb = db - PE_SCAN_SEG; slen = PE_SCAN_SEG + 2 * (uint32_t)QLf; splus = true;
						}
						mate(1 - dir).scan(sdb, slen, splus, dovit);
					}
				}
			}
			// AdjustTopHitsAndMapqs (search2.cpp:8-57)
			if (npairs_found == 0) { hot(0).mapq /= 2; hot(1).mapq /= 2; }
			else {
				const double fract = (double)bestPairScore / (double)(QLf + QLr);
				double drop = (double)(bestPairScore - secondPairScore);
				if (drop > 30) drop = 30;
				double x = drop * fract;
				x = x * fract;
				uint32_t mq = (uint32_t)x;
				if (mq > 40) mq = 40;
				if (mq > hot(0).mapq) hot(0).mapq = mq;
				if (mq > hot(1).mapq) hot(1).mapq = mq;
				if (bestPairIndex >= 0) { hot(0).topHit = bestF; hot(1).topHit = bestR; }
				if (secondPairIndex >= 0) { secondHit[0] = secF; secondHit[1] = secR; }
			}
		}

	pe_output:
		if (pair_info) {  // what State2::OutputTab2 needs beyond the two results (outputtab2.cpp:85-120)
			urmapx_pair_info pi;
			for (int a = 0; a < 2; ++a) {
				pi.top_db[a] = 0xFFFFFFFFu; pi.second_db[a] = 0xFFFFFFFFu;
				pi.top_score[a] = 0; pi.second_score[a] = 0; pi.top_plus[a] = 0; pi.second_plus[a] = 0;
				if (hot(a).topHit >= 0) {
					const uint32_t sp = mate(a).hsp_of(hot(a).topHit);
					pi.top_db[a] = mate(a).hdb(hot(a).topHit); pi.top_score[a] = (int16_t)(sp >> 1); pi.top_plus[a] = (uint8_t)(sp & 1u);
				}
				if (secondHit[a] >= 0) {
					const uint32_t sp = mate(a).hsp_of(secondHit[a]);
					pi.second_db[a] = mate(a).hdb(secondHit[a]); pi.second_score[a] = (int16_t)(sp >> 1); pi.second_plus[a] = (uint8_t)(sp & 1u);
				}
			}
			if (lane == 0) pair_info[pr] = pi;
		}
		// ---- per-mate output: SetMappedPos (state1.cpp:129-145) ----
		for (int a = 0; a < 2; ++a) {
			urmapx_result &R = res[a];
			R.mapq = (uint8_t)(hot(a).mapq > 255 ? 255 : hot(a).mapq);
			R.second = (int16_t)hot(a).second; R.hit_count = (uint16_t)hot(a).hitCount; R.status = (uint8_t)(hot(a).status | hot(1 - a).status);
			R.exit_phase = done ? 1 : 2;
			if (hot(a).topHit >= 0) {
				const uint32_t db = mate(a).hdb(hot(a).topHit);
				const uint32_t sp = mate(a).hsp_of(hot(a).topHit);
				R.score = (int16_t)(sp >> 1);
				uint32_t lo = 0, hi = X.seqCount - 1;
				uint32_t found = 0xFFFFFFFFu, coord = 0xFFFFFFFFu, tl = 0;
				if (URX_SEQ_LANES && X.seqCount <= 64u) {  // every lane tests one sequence: one round of loads (search_se_kernel: fill_result_core)
					const bool mine = (uint32_t)lane < X.seqCount;
					const uint32_t o = mine ? X.seqOffsets[lane] : 0u, sl = mine ? X.seqLengths[lane] : 0u;
					const uint64_t m = __ballot(mine && db >= o && db < o + sl);
					if (m) {
						const int k = __builtin_ctzll(m);
						found = (uint32_t)k; coord = db - rdlane(o, k); tl = rdlane(sl, k);
					}
					hi = 0xFFFFFFFFu;
				}
				while (lo <= hi && hi != 0xFFFFFFFFu) {
					uint32_t k = (lo + hi) / 2;
					uint32_t o = X.seqOffsets[k], sl = X.seqLengths[k];
					if (db >= o && db < o + sl) { found = k; coord = db - o; tl = sl; break; }
					if (db > o) lo = k + 1;
					else hi = k - 1;
				}
				if (found != 0xFFFFFFFFu && coord + (uint32_t)hot(a).QL <= tl) {
					R.dbpos = db; R.seq_index = found; R.coord = coord; R.plus = (uint8_t)(sp & 1u);
					const int nops = mate(a).hnops(hot(a).topHit);
					if (nops > 0) {
						uint32_t po = 0;
						if (lane == 0) po = atomicAdd(path_used, (uint32_t)nops);
						po = uni(po);
						for (int t = lane; t < nops; t += 64) path_ops[po + t] = hot(a).hit_paths[(size_t)hot(a).topHit * URMAPX_MAX_PATH_OPS + t];
						R.path_off = po; R.path_nops = (uint16_t)nops;
					}
				}
			}
			if (lane == 0) results[2 * pr + a] = R;
		}
		if constexpr (TIER == 0) {
			if ((hot(0).status | hot(1).status) & (URMAPX_ST_HSP_OVERFLOW | URMAPX_ST_HIT_OVERFLOW)) {  // queue the pair for the second pass
				if (lane == 0) ovf_next[1 + atomicAdd(ovf_next, 1u)] = pr;
			}
		} else if constexpr (TIER == 1) {
			if ((hot(0).status | hot(1).status) & URMAPX_ST_HIT_OVERFLOW) {  // more than 256 hits on a mate: the third pass
				if (lane == 0) ovf_next[1 + atomicAdd(ovf_next, 1u)] = pr;
			}
		}
	}
}

#undef hot

static int pe_nch_for(uint32_t max_read_len) {
	return max_read_len <= 128 ? 2 : max_read_len <= 192 ? 3 : max_read_len <= 256 ? 4 : max_read_len <= 320 ? 5 : 0;
}

// behind the rows: the trace cells of the banded DP
__host__ __device__ inline size_t pe_tb_offset(int qmax) {
	return (pe_rowstore_offset(qmax) + (size_t)2 * (qmax / 64) * PE_ROW_CAP * 64 * 4 + 255) & ~(size_t)255;
}
// behind the trace cells: the first pass's hits 65..128 of both mates (3 x 64 words each)
__host__ __device__ inline size_t pe_tail_offset(int qmax) {
	return (pe_tb_offset(qmax) + (size_t)(qmax / 8 + 2) * 64 * 4 + 255) & ~(size_t)255;
}
size_t search_pe_scratch_stride(uint32_t max_read_len) {
	const int qmax = 64 * pe_nch_for(max_read_len);
	size_t b = pe_tail_offset(qmax) + (size_t)2 * 192 * 4;
	return (b + 255) & ~(size_t)255;
}

// behind the strided per-block areas, for the second pass's blocks: the HSP overflow lists (two mates), then the
// hit paths of its longer hit lists
static uint32_t pe_hsp_area_blocks(int blocks) { return (uint32_t)(blocks > PE_OVF_BLOCKS ? blocks : PE_OVF_BLOCKS); }
size_t search_pe_scratch_tail(int blocks) {
	return (size_t)pe_hsp_area_blocks(blocks) * 2 * (size_t)PE_HSP_OVF_CAP * sizeof(uint2) +
	       (size_t)PE_OVF_BLOCKS * 2 * (size_t)PE_HIT_CAP * 4 * URMAPX_MAX_PATH_OPS * 2 +
	       (size_t)PE_T2_BLOCKS * 2 * (size_t)PE_HIT_CAP * PE_HITW2 * URMAPX_MAX_PATH_OPS * 2;
}

int search_pe_block_count(uint32_t max_read_len, int device) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
	int per_cu = 0;
	const int nchq = pe_nch_for(max_read_len);
	hipError_t e = nchq == 2   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_pe_kernel<2, 0>, 64, 0)
	               : nchq == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_pe_kernel<3, 0>, 64, 0)
	               : nchq == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_pe_kernel<4, 0>, 64, 0)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search_pe_kernel<5, 0>, 64, 0);
	if (e != hipSuccess || per_cu < 1) per_cu = 4;
	if (const char *t = getenv("URMAPX_TEST_BLOCKS_PER_CU")) { const int v = atoi(t); if (v >= 1 && v < per_cu) per_cu = v; }  // measurement aid (see search_block_count)
	return per_cu * prop.multiProcessorCount;
}

hipError_t launch_search_pe(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t npairs, uint32_t max_read_len, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, const SearchWork &wk, int veryfast,
                            urmapx_pair_info *pair_info, hipStream_t s) {
	if (npairs == 0) return hipSuccess;
	uint32_t *const ovf_list2 = wk.ovf_list + (size_t)npairs + 1;  // the third pass's work list (wk.ovf_list holds 2 * (npairs + 1) words)
	{
		hipError_t e = hipMemsetAsync(wk.ticket, 0, 4, s);
		if (e != hipSuccess) return e;
		e = hipMemsetAsync(wk.ovf_list, 0, 4, s);
		if (e != hipSuccess) return e;
		e = hipMemsetAsync(ovf_list2, 0, 4, s);
		if (e != hipSuccess) return e;
	}
	dim3 block(64);
	const int nch = pe_nch_for(max_read_len);
	uint2 *const ovf_base = reinterpret_cast<uint2 *>(wk.scratch + (size_t)wk.blocks * wk.scratch_stride);  // HSP lists beyond LDS
#define URX_LAUNCH_PE(NCH_, TIER_, GRID_, IN_, OUT_)                                                                          \
	hipLaunchKernelGGL((search_pe_kernel<NCH_, TIER_>), GRID_, block, 0, s, X, P, d_bases, d_offs, npairs, d_results,      \
	                   d_path_ops, d_path_used, wk.scratch, wk.scratch_stride, X.seq, X.blob, X.seqp, veryfast, wk.ticket,        \
	                   pair_info, wk.hsp_lds_cap, IN_, OUT_, ovf_base, pe_hsp_area_blocks(wk.blocks))
#define URX_LAUNCH_PE_TIER(TIER_, GRID_, IN_, OUT_)                                                                            \
	do {                                                                                                                        \
		if (nch == 2) URX_LAUNCH_PE(2, TIER_, GRID_, IN_, OUT_);                                                                  \
		else if (nch == 3) URX_LAUNCH_PE(3, TIER_, GRID_, IN_, OUT_);                                                             \
		else if (nch == 4) URX_LAUNCH_PE(4, TIER_, GRID_, IN_, OUT_);                                                             \
		else URX_LAUNCH_PE(5, TIER_, GRID_, IN_, OUT_);                                                                           \
		hipError_t e_ = hipGetLastError();                                                                                        \
		if (e_ != hipSuccess) return e_;                                                                                          \
		e_ = hipMemsetAsync(wk.ticket, 0, 4, s);                                                                                  \
		if (e_ != hipSuccess) return e_;                                                                                          \
	} while (0)
	uint32_t *const none = nullptr;
	URX_LAUNCH_PE_TIER(0, dim3((unsigned)wk.blocks), none, wk.ovf_list);
	// second pass over the pairs whose HSP or hit lists outgrew the first pass's (see launch_search_se); third pass over
	// the pairs with more than 256 hits on a mate.  Both find their work lists on the device and usually leave at once.
	URX_LAUNCH_PE_TIER(1, dim3((unsigned)(wk.blocks < PE_OVF_BLOCKS ? wk.blocks : PE_OVF_BLOCKS)), wk.ovf_list, ovf_list2);
	URX_LAUNCH_PE_TIER(2, dim3((unsigned)(wk.blocks < PE_T2_BLOCKS ? wk.blocks : PE_T2_BLOCKS)), ovf_list2, none);
#undef URX_LAUNCH_PE_TIER
#undef URX_LAUNCH_PE
	return hipGetLastError();
}

}  // namespace urx
