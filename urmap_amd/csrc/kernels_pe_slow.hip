// kernels_pe_slow.hip -- the general paired-end search: State2::Search4 / Search5 (search2m4.cpp:15-208, search2m5.cpp:9-156)
// for the pairs the fast passes of kernels_pe.hip leave flagged.
//
// The fast pair kernel keeps a pair's state in LDS and therefore has a domain: 1024 live hits and 8192 HSPs per mate,
// alignment paths of 96 runs.  The reference has none of these limits (its lists grow by 64 without bound,
// state1.cpp:193-228; its pair list is a vector, state2.cpp:20-85).  A pair outside the fast kernel's domain -- flagged by
// it, never mis-mapped -- is mapped again here with every list of both mates in this block's global scratch (slow_dev.h:
// 65 536 hits, 65 536 HSPs, paths as long as the read, one path per hit in an arena).  One wavefront per pair, the
// reference's schedule call by call; the lanes share the byte compares of one ExtendPen, the list scans, the chain walks
// of 64 pending k-mers, the k-mers of 64 window positions in ScanSlots, the hit x hit test of FindPairs and the DP's
// diagonals.  It is for the rare pair (a tandem satellite, a centromere): its only job is to be the reference.
//
// Reference functions: InitPE (state1.cpp:95-127), GetFirstBoth1Seed / GetNextBoth1Seed (getseed.cpp:9-138),
// ExtendBoth1Pair4 (search2m4.cpp:189-208), SearchPE_Pending (search1pepend.cpp:9-130), FindPairs / ScanPair
// (state2.cpp:20-137), Scan / ScanSlots / ExtendScan / AddHSPScan (scan.cpp:14-39, scanslots.cpp:7-62,
// extendscan.cpp:8-187), AdjustTopHitsAndMapqs (search2.cpp:8-57), SetMappedPos (state1.cpp:129-145).
#include "slow_dev.h"

namespace urx {

static constexpr uint32_t PES_QCAP = 320;        // mates the pair path takes at all (internal.h: MAX_QL_PE)
static constexpr uint32_t PES_SCAN_SEG = 1024;   // SCAN_DB_SEG_LENGTH, state2.cpp:92
static constexpr uint32_t PES_PARENA = 1u << 20; // runs of all hit paths of a mate
static constexpr uint32_t PES_PRIME_STRIDE = 27, PES_SCANK = 4;
static constexpr int PES_MAX_TL = 1000;

struct PeSlowLayout {
	SlowLayout one;     // per mate (slow_layout(PES_QCAP)), without its DP areas being shared
	size_t mate_stride; // one.total + the pair-mode arrays of a mate
	size_t hit_score, hit_plus, hit_poff, hit_pn, parena, pend, seed_q, seed_db;  // offsets inside a mate's area, behind one.total
	size_t ws;          // the rescue DP's wide scratch (shared by the mates), behind the two mates
	size_t total;
};
__host__ __device__ inline PeSlowLayout pe_slow_layout() {
	PeSlowLayout L;
	L.one = slow_layout(PES_QCAP);
	size_t o = L.one.total;
	auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
	L.hit_score = take((size_t)SLOW_HITCAP * 2);
	L.hit_plus = take((size_t)SLOW_HITCAP);
	L.hit_poff = take((size_t)SLOW_HITCAP * 4);
	L.hit_pn = take((size_t)SLOW_HITCAP * 4);
	L.parena = take((size_t)PES_PARENA * 2);
	L.pend = take(2 * (size_t)PES_QCAP * 2);
	L.seed_q = take(2 * (size_t)PES_QCAP * 2);
	L.seed_db = take(2 * (size_t)PES_QCAP * 4);
	L.mate_stride = o;
	L.ws = 2 * L.mate_stride;
	L.total = L.ws + ((WideScratch::bytes((int)PES_QCAP, (int)(PES_SCAN_SEG + 2 * PES_QCAP + 64)) + 255) & ~(size_t)255);
	return L;
}
size_t pe_slow_scratch_stride() { return pe_slow_layout().total; }

struct SlowMate : SlowWave {
	uint16_t *pend[2];
	int pendCount[2];
	uint16_t *seed_q;   // qpos | plus << 15, in the order GetFirstBoth1Seed / GetNextBoth1Seed return them
	uint32_t *seed_db;
	int nseed;
	uint32_t mapq;
	WideScratch ws_scan;  // the whole-read Viterbi of Scan: QL x (1024 + 2 QL) cells

	__device__ SlowMate(const DevIndex &X_, const urmapx_params &P_, int lane_) : SlowWave(X_, P_, lane_) {}

	// GetFirstBoth1Seed / GetNextBoth1Seed (getseed.cpp:9-138) run to the end: the seeds in the order they are returned and the
	// two pending lists.  The enumeration does not depend on the search state.  64 steps k are loaded at a time (one per
	// lane), then walked in order.
	__device__ void enumerate_seeds() {
		const int QWC = nwords;
		nseed = 0; pendCount[0] = pendCount[1] = 0;
		bool have = false;
		uint32_t lastDiag = 0;
		for (int kb = 0; kb < QWC; kb += 64) {
			const int k = kb + lane;
			const bool in = k < QWC;
			const uint32_t qpos = in ? ((uint32_t)k * PES_PRIME_STRIDE) % (uint32_t)QWC : 0u;
			const uint32_t Tp = in ? tal[0][qpos] : 0u, Tm = in ? tal[1][qpos] : 0u;
			const uint32_t dbP = in ? pos[0][qpos] : 0u, dbM = in ? pos[1][qpos] : 0u;
			const int nb = QWC - kb < 64 ? QWC - kb : 64;
			for (int l = 0; l < nb; ++l) {
				const uint32_t tp = rdlane(Tp, l), tm = rdlane(Tm, l), q = rdlane(qpos, l);
				const uint32_t dp = rdlane(dbP, l), dm = rdlane(dbM, l);
				auto emit = [&](uint32_t qq, uint32_t db) {
					if (lane == 0) { seed_q[nseed] = (uint16_t)qq; seed_db[nseed] = db; }
					++nseed;
				};
				auto push = [&](int s) {
					if (lane == 0) pend[s][pendCount[s]] = (uint16_t)q;
					++pendCount[s];
				};
				bool retP = false;
				if (tp & TALLY_MY_BIT) {
					if (tp != TALLY_BOTH1) push(0);
					else {
						const uint32_t d = dp - q;
						if (!have || d != lastDiag) { emit(q | 0x8000u, dp); lastDiag = d; have = true; retP = true; }
					}
				}
				if (retP) {  // GetNextBoth1Seed's look at the minus strand of the step whose plus seed it has just returned
					if ((tm & TALLY_MY_BIT) && tm == TALLY_BOTH1) {
						const uint32_t d = dm - q;
						if (d != lastDiag) { emit(q, dm); lastDiag = d; }
						else push(1);
					}
				} else if (tm & TALLY_MY_BIT) {
					if (tm != TALLY_BOTH1) push(1);
					else {
						const uint32_t d = dm - q;
						if (!have || d != lastDiag) { emit(q, dm); lastDiag = d; have = true; }
					}
				}
			}
		}
		__syncthreads();
	}

	// search1pepend.cpp:9-130
	__device__ void search_pending() {
		maxPen = P.max_penalty;
		const int minScore1 = QL + P.xphase1 * P.mismatch_score;
		const int termHSP3 = (QL * P.term_hsp_score_pct_phase3) / 100;
		if (best >= minScore1) { mapq = calc_mapq(); return; }
		if (bestHSP >= termHSP3) {
			for (int k = 0; k < hspCount; ++k) align_hsp(k);
			if (best >= minScore1) { mapq = calc_mapq(); return; }
		}
		int count2[2] = {0, 0};
		for (int round = 0; round < 2; ++round) {
			for (int s = 0; s < 2; ++s) {
				const int n = round == 0 ? pendCount[s] : count2[s];
				for (int base = 0; base < n; base += 64) {
					const int i = base + lane;
					const bool valid = i < n;
					const int p = valid ? (int)pend[s][i] : 0;
					const int rl = get_rows(s, p, valid);
					const int nb = n - base < 64 ? n - base : 64;
					for (int b = 0; b < nb; ++b) {
						const int len = rdlane(rl, b), qp = rdlane(p, b);
						if (round == 0 && len > 2) {  // put off to the second round: compacted to the front of the list
							__syncthreads();
							if (lane == 0) pend[s][count2[s]] = (uint16_t)qp;
							++count2[s];
							continue;
						}
						for (int k = 0; k < len; ++k) extend_pen((uint32_t)qp, rows[b * SLOW_ROW_CAP + k], s == 0);
					}
					__syncthreads();
				}
			}
		}
		const int bmin = max(best, bestHSP) - 8;
		for (int k = 0; k < hspCount; ++k) {
			if (hsp_score[k] < bmin) continue;
			align_hsp(k);
		}
		mapq = calc_mapq();
	}

	// scanslots.cpp:7-62: every window k-mer against the read's slots at the first SCANK prime-stride positions
	__device__ void scan_slots(uint32_t dblo, uint32_t seglen, bool plus) {
		if (QL <= 4 * W) return;
		const int s = plus ? 0 : 1;
		uint64_t qslot[PES_SCANK];
		uint32_t qposk[PES_SCANK];
#pragma unroll
		for (uint32_t k = 0; k < PES_SCANK; ++k) {
			qposk[k] = (k * PES_PRIME_STRIDE) % (uint32_t)nwords;
			qslot[k] = uni64(slots[s][qposk[k]]);
		}
		const uint64_t wmask = X.shiftMask;
		const uint64_t wbits = (W >= 32) ? 0xFFFFFFFFull : ((1ull << W) - 1ull);
		auto planes = [&](uint32_t start, uint64_t &lo, uint64_t &hi, uint64_t &inv) {
			const uint32_t q = start + lane;
			uint32_t L = 4;
			if (q < seglen) L = letter_of(gseq[dblo + q]);
			lo = __ballot(L & 1u); hi = __ballot((L >> 1) & 1u); inv = __ballot(L > 3u);
		};
		uint64_t lo0, hi0, inv0, lo1, hi1, inv1;
		planes(0, lo0, hi0, inv0);
		for (uint32_t base = 0; base + (uint32_t)W <= seglen; base += 64) {
			planes(base + 64, lo1, hi1, inv1);
			const uint32_t p = base + lane;
			uint64_t flo = lo0 >> lane, fhi = hi0 >> lane, finv = inv0 >> lane;
			if (lane) { flo |= lo1 << (64 - lane); fhi |= hi1 << (64 - lane); finv |= inv1 << (64 - lane); }
			flo &= wbits; fhi &= wbits; finv &= wbits;
			const bool valid = p + (uint32_t)W <= seglen && finv == 0;
			const uint64_t word = spread32(__brevll(flo) >> (64 - W)) | (spread32(__brevll(fhi) >> (64 - W)) << 1);
			lo0 = lo1; hi0 = hi1; inv0 = inv1;
			uint64_t slot = ~0ull;
			if (valid) slot = mod_slots(murmur64(word & wmask), X.slotCount, X.slotMagic);
			uint32_t hitmask = 0;
#pragma unroll
			for (uint32_t k = 0; k < PES_SCANK; ++k)
				if (valid && slot == qslot[k]) hitmask |= (1u << k);
			uint64_t any = __ballot(hitmask != 0);
			while (any) {  // window positions in ascending order, k ascending inside (scanslots.cpp:50-59)
				const int l = __builtin_ctzll(any);
				any &= any - 1;
				const uint32_t hm = rdlane(hitmask, l);
				for (uint32_t k = 0; k < PES_SCANK; ++k)
					if (hm & (1u << k)) extend_pen(qposk[k], dblo + base + (uint32_t)l, plus, true);
			}
		}
	}

	// scan.cpp:14-39
	__device__ void scan(uint32_t dbpos, uint32_t seglen, bool plus, bool dovit) {
		const int savedMaxPen = maxPen;
		const int savedHits = hitCount;
		maxPen = 130;
		scan_slots(dbpos, seglen, plus);
		maxPen = savedMaxPen;
		if (hitCount > savedHits) return;
		if (!dovit) return;
		RevOps R;
		R.ops = ropsL;
		R.cap = (int)pathcap;
		uint32_t vst = 0;
		const float score = viterbi_wave<false>(VPar(P), q[plus ? 0 : 1], QL, gseq + dbpos, (int)seglen, true, true, tb, tb_rows8, ws_scan, R, vst, lane);
		status |= vst;
		if (vst) return;
		if ((double)score >= (double)QL / 3.0) {
			int n = R.n, nI = 0, r0 = 0;
			if (n > 0 && (ropsL[n - 1] & 3u) == OP_I) { nI = (int)(ropsL[n - 1] >> 2); --n; }  // TrimLeftIs
			if (n > 1 && (ropsL[0] & 3u) == OP_I) r0 = 1;                                     // TrimRightIs
			const int nc = n - r0;
			if (nc > (int)pathcap) { status |= URMAPX_ST_PATH_OVERFLOW; return; }
			__syncthreads();
			for (int t = lane; t < nc; t += 64) cand[t] = ropsL[n - 1 - t];
			__syncthreads();
			add_hit(dbpos + (uint32_t)nI, plus, (int)score, nc);
		}
	}
};

// the pairs a batch's fast passes left flagged: list[0] = count, list[1..] = pair numbers
__global__ __launch_bounds__(256) void collect_flagged_pairs_kernel(const urmapx_result *__restrict__ results, const uint64_t *__restrict__ offs,
                                                                    uint32_t npairs, uint32_t W, uint32_t *list, int all) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npairs) return;
	if (!all && (results[2 * i].status | results[2 * i + 1].status) == 0) return;  // all: test aid (URMAPX_TEST_PE_GENERAL), every pair again
	for (int a = 0; a < 2; ++a) {
		const uint64_t ql = offs[2 * i + a + 1] - offs[2 * i + a];
		// shorter than a word: the reference underflows; more than 256 k-mer starts: it keeps pending positions in a byte
		// (state1.h:86-87) and fails itself
		if (ql < W || ql > PES_QCAP || ql - (W - 1) > 256) return;
	}
	list[1 + atomicAdd(list, 1u)] = i;
}

__device__ static void carve_mate(SlowMate &S, uint8_t *sc, const PeSlowLayout &PL, uint8_t *ws_base) {
	const SlowLayout &L = PL.one;
	const uint32_t qcap = PES_QCAP;
	S.q[0] = sc + L.q; S.q[1] = sc + L.q + qcap + 64;
	S.slots[0] = reinterpret_cast<uint64_t *>(sc + L.slots); S.slots[1] = S.slots[0] + qcap;
	S.tal[0] = sc + L.tal; S.tal[1] = S.tal[0] + qcap;
	S.pos[0] = reinterpret_cast<uint32_t *>(sc + L.pos); S.pos[1] = S.pos[0] + qcap;
	S.hit_db = reinterpret_cast<uint32_t *>(sc + L.hit_db);
	S.hsp_db = reinterpret_cast<uint32_t *>(sc + L.hsp_db); S.hsp_q = reinterpret_cast<uint32_t *>(sc + L.hsp_q);
	S.hsp_len = reinterpret_cast<uint32_t *>(sc + L.hsp_len); S.hsp_score = reinterpret_cast<int32_t *>(sc + L.hsp_score);
	S.hsp_fl = sc + L.hsp_fl;
	S.todo[0] = reinterpret_cast<uint32_t *>(sc + L.todo); S.todo[1] = S.todo[0] + qcap;
	S.rows = reinterpret_cast<uint32_t *>(sc + L.rows);
	S.ropsL = reinterpret_cast<uint16_t *>(sc + L.ropsL); S.ropsR = reinterpret_cast<uint16_t *>(sc + L.ropsR);
	S.cand = reinterpret_cast<uint16_t *>(sc + L.cand); S.top = reinterpret_cast<uint16_t *>(sc + L.top);
	S.tb = reinterpret_cast<uint32_t *>(sc + L.tb);
	S.ws.carve(sc + L.ws, (int)SLOW_WIDE_CAP, (int)SLOW_WIDE_CAP);
	S.hspcap = L.hspcap; S.pathcap = L.pathcap; S.tb_rows8 = L.tb_rows8;
	S.hit_score = reinterpret_cast<int16_t *>(sc + PL.hit_score); S.hit_plus = sc + PL.hit_plus;
	S.hit_poff = reinterpret_cast<uint32_t *>(sc + PL.hit_poff); S.hit_pn = reinterpret_cast<uint32_t *>(sc + PL.hit_pn);
	S.parena = reinterpret_cast<uint16_t *>(sc + PL.parena); S.parena_cap = PES_PARENA;
	S.pend[0] = reinterpret_cast<uint16_t *>(sc + PL.pend); S.pend[1] = S.pend[0] + qcap;
	S.seed_q = reinterpret_cast<uint16_t *>(sc + PL.seed_q);
	S.seed_db = reinterpret_cast<uint32_t *>(sc + PL.seed_db);
	S.ws_scan.carve(ws_base, (int)PES_QCAP, (int)(PES_SCAN_SEG + 2 * PES_QCAP + 64));
}

__global__ __launch_bounds__(64) void search_pe_slow_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                            const uint64_t *__restrict__ offs, const uint32_t *__restrict__ list,
                                                            urmapx_result *__restrict__ results, urmapx_path_op *__restrict__ path_ops,
                                                            uint32_t *path_used, uint32_t path_cap, uint8_t *scratch, size_t stride,
                                                            const uint8_t *__restrict__ g_seq, const uint8_t *__restrict__ g_blob,
                                                            int veryfast, uint32_t *ticket, urmapx_pair_info *pair_info) {
	const int lane = threadIdx.x;
	const uint32_t n = list[0];
	if (n == 0) return;
	const PeSlowLayout PL = pe_slow_layout();
	uint8_t *sc = scratch + (size_t)blockIdx.x * stride;
	SlowMate F(X, P, lane), R(X, P, lane);
	F.gseq = R.gseq = g_seq; F.gblob = R.gblob = g_blob;
	F.W = R.W = (int)X.W;
	carve_mate(F, sc, PL, sc + PL.ws);
	carve_mate(R, sc + PL.mate_stride, PL, sc + PL.ws);
	for (;;) {
		const uint32_t idx = uni(atomicAdd(ticket, lane == 0 ? 1u : 0u));  // every lane takes part (see search_se_kernel)
		if (idx >= n) break;
		const uint32_t pr = list[1 + idx];
		urmapx_result res[2];
		bool bad = false;
		auto init = [&](SlowMate &S, int a) {
			const uint32_t r = 2 * pr + a;
			const uint64_t off = offs[r];
			const int QL = (int)(offs[r + 1] - off);
			res[a].dbpos = 0xFFFFFFFFu; res[a].seq_index = 0xFFFFFFFFu; res[a].coord = 0xFFFFFFFFu;
			res[a].score = 0; res[a].second = 0; res[a].mapq = 0; res[a].plus = 0; res[a].exit_phase = 0; res[a].status = 0;
			res[a].hit_count = 0; res[a].path_nops = 0; res[a].path_off = 0;
			if (QL < S.W || (uint32_t)QL > PES_QCAP || S.W > 32 || X.maxIx > (uint32_t)SLOW_ROW_CAP || QL - (S.W - 1) > 256) { bad = true; return; }
			// InitPE (state1.cpp:95-127)
			S.QL = QL; S.nwords = QL - (S.W - 1);
			S.hitCount = 0; S.hspCount = 0; S.topHit = -1; S.parena_used = 0;
			S.maxPen = P.max_penalty; S.best = 0; S.second = 0; S.bestHSP = 0;
			S.haveTop = false; S.top_db = 0; S.top_plus = false; S.top_nops = 0; S.status = 0;
			S.mapq = 0xFFFFFFFFu;
			__syncthreads();
			for (int p = lane; p < QL; p += 64) {
				const uint8_t ch = bases[off + p];
				S.q[0][p] = ch;
				S.q[1][QL - 1 - p] = (uint8_t)comp_char(ch);
			}
			__syncthreads();
		};
		init(F, 0);
		init(R, 1);
		if (bad) {
			for (int a = 0; a < 2; ++a) { res[a].status = URMAPX_ST_BAD_LENGTH; if (lane == 0) results[2 * pr + a] = res[a]; }
			continue;
		}
		F.probe_all(); R.probe_all();
		F.enumerate_seeds(); R.enumerate_seeds();

		// ---- Search4's seed loop (search2m4.cpp:71-143) ----
		const int QLf = F.QL, QLr = R.QL;
		const int64_t QL2 = (int64_t)((QLf + QLr) / 2);
		const int termPair = QLf + QLr + 5 * P.mismatch_score;
		bool done = false;
		auto near = [&](uint32_t a, uint32_t b) {
			int64_t d = (int64_t)a - (int64_t)b;
			if (d < 0) d = -d;
			return d + QL2 <= PES_MAX_TL;
		};
		// ExtendBoth1Pair4 (search2m4.cpp:189-208)
		auto pair4 = [&](uint32_t qf, uint32_t dbf, bool plusf, uint32_t qr, uint32_t dbr) -> bool {
			const int fs = F.extend_pen(qf, dbf, plusf);
			if (fs <= 0) return false;
			const int rs = R.extend_pen(qr, dbr, !plusf);
			if (rs <= 0) return false;
			if (fs + rs < termPair) return false;
			F.mapq = 40; R.mapq = 40;
			return true;
		};
		{
			const int steps = F.nseed > R.nseed ? F.nseed : R.nseed;
			for (int t = 0; t < steps && !done; ++t) {
				if (t < F.nseed) {
					const uint32_t sq = F.seed_q[t], dbf = F.seed_db[t];
					const int nr = t < R.nseed ? t : R.nseed;
					for (int i = 0; i < nr && !done; ++i) {
						const uint32_t dbr = R.seed_db[i];
						if (near(dbf, dbr)) done = pair4(sq & 0x7FFFu, dbf, (sq & 0x8000u) != 0, R.seed_q[i] & 0x7FFFu, dbr);
					}
				}
				if (t < R.nseed && !done) {
					const uint32_t sq = R.seed_q[t], dbr = R.seed_db[t];
					const bool plusr = (sq & 0x8000u) != 0;
					const int nf = t + 1 < F.nseed ? t + 1 : F.nseed;
					for (int i = 0; i < nf && !done; ++i) {
						const uint32_t dbf = F.seed_db[i];
						if (near(dbf, dbr)) done = pair4(F.seed_q[i] & 0x7FFFu, dbf, !plusr, sq & 0x7FFFu, dbr);
					}
				}
			}
		}
		int npairs_found = 0, bestPairScore = -1, secondPairScore = -1, bestPairIndex = -1, secondPairIndex = -1;
		int bestF = -1, bestR = -1, secF = -1, secR = -1;
		int secondHit[2] = {-1, -1};
		if (!done) {
			// all collected seeds, each mate (search2m4.cpp:145-158)
			for (int i = 0; i < F.nseed; ++i) { const uint32_t sq = F.seed_q[i]; F.extend_pen(sq & 0x7FFFu, F.seed_db[i], (sq & 0x8000u) != 0); }
			for (int i = 0; i < R.nseed; ++i) { const uint32_t sq = R.seed_q[i]; R.extend_pen(sq & 0x7FFFu, R.seed_db[i], (sq & 0x8000u) != 0); }
			if (veryfast) {
				// Search5 (search2m5.cpp:112-127): no 90 % shortcut and no pair stage; each mate finishes on its own
				F.search_pending();
				R.search_pending();
				done = true;
			} else if (F.best >= (QLf * 9) / 10 && R.best >= (QLr * 9) / 10 && F.topHit >= 0 && R.topHit >= 0) {
				if (near(F.hit_db[F.topHit], R.hit_db[R.topHit])) { F.mapq = 40; R.mapq = 40; done = true; }
			}
		}
		if (!done) {
			F.search_pending();
			R.search_pending();
			// FindPairs (state2.cpp:20-85), ScanPair if there is none (state2.cpp:87-137), FindPairs again.  The reference keeps
			// every pair and afterwards reads the best and the second-best entry only: their hit indexes are carried along.
			for (int attempt = 0; attempt < 2; ++attempt) {
				npairs_found = 0; bestPairScore = -1; secondPairScore = -1; bestPairIndex = -1; secondPairIndex = -1;
				bestF = bestR = secF = secR = -1;
				__syncthreads();
				for (int i = 0; i < F.hitCount; ++i) {
					const int sf = F.hit_score[i];
					if (sf < F.second - 12) continue;
					const uint32_t dbf = F.hit_db[i];
					const uint32_t plf = F.hit_plus[i];
					for (int jb = 0; jb < R.hitCount; jb += 64) {  // 64 hits of the reverse mate per step, those that make a pair in order
						const int j = jb + lane;
						bool ok = false;
						int sr = 0;
						if (j < R.hitCount) {
							sr = R.hit_score[j];
							ok = sr >= R.second - 12 && near(dbf, R.hit_db[j]) && (uint32_t)R.hit_plus[j] != plf;
						}
						uint64_t m = __ballot(ok);
						while (m) {
							const int l = __builtin_ctzll(m);
							m &= m - 1;
							const int total = sf + rdlane(sr, l);
							const int jj = jb + l;
							if (total > bestPairScore) {
								secondPairIndex = bestPairIndex; secondPairScore = bestPairScore; secF = bestF; secR = bestR;
								bestPairScore = total; bestPairIndex = npairs_found; bestF = i; bestR = jj;
							} else if (total == bestPairScore) { secondPairIndex = npairs_found; secondPairScore = bestPairScore; secF = i; secR = jj; }
							else if (total > secondPairScore) { secondPairIndex = bestPairIndex; secondPairScore = total; secF = bestF; secR = bestR; }  // sic, state2.cpp:74
							++npairs_found;
						}
					}
				}
				if (npairs_found > 0 || attempt == 1) break;
				// ScanPair
				const bool dovitF = (int)F.mapq >= 10, dovitR = (int)R.mapq >= 10;
				const int hcf = F.hitCount, hcr = R.hitCount;
				for (int i = 0; i < hcf; ++i) {
					if ((int)F.hit_score[i] < F.second) continue;
					const uint32_t db = F.hit_db[i];
					if (F.hit_plus[i]) R.scan(db, PES_SCAN_SEG, false, dovitF);
					else if (db >= PES_SCAN_SEG) R.scan(db - PES_SCAN_SEG, PES_SCAN_SEG + 2 * (uint32_t)QLf, true, dovitF);
				}
				for (int j = 0; j < hcr; ++j) {
					if ((int)R.hit_score[j] < R.second) continue;
					const uint32_t db = R.hit_db[j];
					if (R.hit_plus[j]) F.scan(db, PES_SCAN_SEG, false, dovitR);
					else if (db >= PES_SCAN_SEG) F.scan(db - PES_SCAN_SEG, PES_SCAN_SEG + 2 * (uint32_t)QLf, true, dovitR);  // sic: the forward read's length
				}
			}
			// AdjustTopHitsAndMapqs (search2.cpp:8-57)
			if (npairs_found == 0) { F.mapq /= 2; R.mapq /= 2; }
			else {
				const double fract = (double)bestPairScore / (double)(QLf + QLr);
				double drop = (double)(bestPairScore - secondPairScore);
				if (drop > 30) drop = 30;
				double x = drop * fract;
				x = x * fract;
				uint32_t mq = (uint32_t)x;
				if (mq > 40) mq = 40;
				if (mq > F.mapq) F.mapq = mq;
				if (mq > R.mapq) R.mapq = mq;
				if (bestPairIndex >= 0) { F.topHit = bestF; R.topHit = bestR; }
				if (secondPairIndex >= 0) { secondHit[0] = secF; secondHit[1] = secR; }
			}
		}
		__syncthreads();
		SlowMate *const M[2] = {&F, &R};
		if (pair_info) {  // what State2::OutputTab2 needs beyond the two results (outputtab2.cpp:85-120)
			urmapx_pair_info pi;
			for (int a = 0; a < 2; ++a) {
				const SlowMate &S = *M[a];
				pi.top_db[a] = 0xFFFFFFFFu; pi.second_db[a] = 0xFFFFFFFFu;
				pi.top_score[a] = 0; pi.second_score[a] = 0; pi.top_plus[a] = 0; pi.second_plus[a] = 0;
				if (S.topHit >= 0) { pi.top_db[a] = S.hit_db[S.topHit]; pi.top_score[a] = S.hit_score[S.topHit]; pi.top_plus[a] = S.hit_plus[S.topHit]; }
				if (secondHit[a] >= 0) { pi.second_db[a] = S.hit_db[secondHit[a]]; pi.second_score[a] = S.hit_score[secondHit[a]]; pi.second_plus[a] = S.hit_plus[secondHit[a]]; }
			}
			if (lane == 0) pair_info[pr] = pi;
		}
		// per-mate output: SetMappedPos (state1.cpp:129-145)
		for (int a = 0; a < 2; ++a) {
			const SlowMate &S = *M[a];
			urmapx_result &O = res[a];
			O.mapq = (uint8_t)(S.mapq > 255 ? 255 : S.mapq);
			O.second = (int16_t)S.second; O.hit_count = (uint16_t)(S.hitCount > 0xFFFF ? 0xFFFF : S.hitCount);
			O.status = (uint8_t)(F.status | R.status);
			O.exit_phase = done ? 1 : 2;
			if (S.topHit >= 0) {
				const uint32_t db = S.hit_db[S.topHit];
				O.score = S.hit_score[S.topHit];
				uint32_t lo = 0, hi = X.seqCount - 1;
				uint32_t found = 0xFFFFFFFFu, coord = 0xFFFFFFFFu, tl = 0;
				while (lo <= hi && hi != 0xFFFFFFFFu) {
					const uint32_t k = (lo + hi) / 2;
					const uint32_t o = X.seqOffsets[k], sl = X.seqLengths[k];
					if (db >= o && db < o + sl) { found = k; coord = db - o; tl = sl; break; }
					if (db > o) lo = k + 1;
					else hi = k - 1;
				}
				if (found != 0xFFFFFFFFu && coord + (uint32_t)S.QL <= tl) {
					O.dbpos = db; O.seq_index = found; O.coord = coord; O.plus = S.hit_plus[S.topHit];
					const uint32_t nops = S.hit_pn[S.topHit];
					if (nops > 0) {
						uint32_t po = 0;
						if (lane == 0) po = reserve_path(path_used, nops, path_cap);  // room is taken only if the path fits
						po = uni(po);
						if (po != 0xFFFFFFFFu && nops <= 0xFFFFu) {
							const uint16_t *src = S.parena + S.hit_poff[S.topHit];
							for (uint32_t t = lane; t < nops; t += 64) path_ops[po + t] = src[t];
							O.path_off = po; O.path_nops = (uint16_t)nops;
						} else
							O.status |= URMAPX_ST_PATH_OVERFLOW;  // the batch's path arena is full
					}
				}
			}
		}
		if (lane == 0) { results[2 * pr] = res[0]; results[2 * pr + 1] = res[1]; }
	}
}

hipError_t launch_search_pe_slow(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t npairs,
                                 urmapx_result *d_results, urmapx_path_op *d_path_ops, uint32_t *d_path_used, uint32_t path_cap,
                                 uint8_t *scratch, int blocks, uint32_t *list, uint32_t *ticket, int veryfast, urmapx_pair_info *pair_info,
                                 int all_pairs, hipStream_t s) {
	if (npairs == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(list, 0, 4, s);
	if (e == hipSuccess) e = hipMemsetAsync(ticket, 0, 4, s);
	if (e == hipSuccess && all_pairs) e = hipMemsetAsync(d_path_used, 0, 4, s);  // every result is written again: the paths too
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(collect_flagged_pairs_kernel, dim3((npairs + 255) / 256), dim3(256), 0, s, d_results, d_offs, npairs, X.W, list, all_pairs);
	hipLaunchKernelGGL(search_pe_slow_kernel, dim3((unsigned)blocks), dim3(64), 0, s, X, P, d_bases, d_offs, list, d_results, d_path_ops,
	                   d_path_used, path_cap, scratch, pe_slow_scratch_stride(), X.seq, X.blob, veryfast & 1, ticket, pair_info);
	return hipGetLastError();
}

}  // namespace urx
