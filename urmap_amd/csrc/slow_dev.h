// slow_dev.h -- the general search state shared by the single-end general kernel (kernels_slow.hip) and the pair one
// (kernels_pe_slow.hip): one wavefront per read, every list in the block's global scratch, the reference's schedule
// candidate by candidate.  See kernels_slow.hip for what it is for.
#pragma once
#include "kernels.h"

#include "dev_common.h"
#include "viterbi_dev.h"

namespace urx {

static constexpr uint32_t SLOW_HITCAP = 65536;     // live hits of one read (a tandem satellite of 10^4 copies fits)
static constexpr uint32_t SLOW_WIDE_CAP = 1100;    // wide-band DP (flank window clipped at the end of the sequence store)
static constexpr int SLOW_ROW_CAP = 32;            // UFIndex m_MaxIx of every index this build accepts

// n entries of a batch's path arena, or 0xFFFFFFFF if they do not fit (the counter then stays where it was)
__device__ inline uint32_t reserve_path(uint32_t *used, uint32_t n, uint32_t cap) {
	uint32_t cur = *used;
	for (;;) {
		if (cur > cap || n > cap - cur) return 0xFFFFFFFFu;
		const uint32_t seen = atomicCAS(used, cur, cur + n);
		if (seen == cur) return cur;
		cur = seen;
	}
}

struct SlowLayout {
	size_t q, slots, tal, pos, hit_db, hsp_db, hsp_q, hsp_len, hsp_score, hsp_fl, todo, rows, ropsL, ropsR, cand, top, tb, ws, total;
	uint32_t qcap, hspcap, pathcap;
	int tb_rows8;
};
// HSPs live one per diagonal: at most one per candidate, 2 strands x QL k-mers x MaxIx positions (capped: a read whose
// every k-mer owns a full chain)
__host__ __device__ inline SlowLayout slow_layout(uint32_t qcap) {
	SlowLayout L;
	L.qcap = qcap;
	uint64_t h = 2ull * qcap * SLOW_ROW_CAP;
	L.hspcap = (uint32_t)(h < 65536 ? 65536 : (h > 262144 ? 262144 : h));
	L.pathcap = qcap + 64;
	L.tb_rows8 = (int)(qcap / 8 + 3);
	size_t o = 0;
	auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
	L.q = take(2 * ((size_t)qcap + 64));
	L.slots = take(2 * (size_t)qcap * 8);
	L.tal = take(2 * (size_t)qcap);
	L.pos = take(2 * (size_t)qcap * 4);
	L.hit_db = take((size_t)SLOW_HITCAP * 4);
	L.hsp_db = take((size_t)L.hspcap * 4);
	L.hsp_q = take((size_t)L.hspcap * 4);
	L.hsp_len = take((size_t)L.hspcap * 4);
	L.hsp_score = take((size_t)L.hspcap * 4);
	L.hsp_fl = take((size_t)L.hspcap);
	L.todo = take(2 * (size_t)qcap * 4);
	L.rows = take(64 * SLOW_ROW_CAP * 4);
	L.ropsL = take((size_t)L.pathcap * 2);
	L.ropsR = take((size_t)L.pathcap * 2);
	L.cand = take((size_t)L.pathcap * 2);
	L.top = take((size_t)L.pathcap * 2);
	L.tb = take((size_t)L.tb_rows8 * 64 * 4);
	L.ws = take(WideScratch::bytes((int)SLOW_WIDE_CAP, (int)SLOW_WIDE_CAP));
	L.total = o;
	return L;
}


struct SlowWave {
	const DevIndex &X;
	const urmapx_params &P;
	const int lane;
	const uint8_t *__restrict__ gseq;
	const uint8_t *__restrict__ gblob;
	uint8_t *q[2];
	uint64_t *slots[2];
	uint8_t *tal[2];
	uint32_t *pos[2];
	uint32_t *hit_db, *hsp_db, *hsp_q, *hsp_len;
	int32_t *hsp_score;
	uint8_t *hsp_fl;  // bit 0 plus, bit 1 aligned
	uint32_t *todo[2];
	uint32_t *rows;
	uint16_t *ropsL, *ropsR, *cand, *top;
	uint32_t *tb;
	WideScratch ws;
	uint32_t hspcap, pathcap;
	int tb_rows8;
	int QL, W, nwords;
	int hitCount, hspCount;
	int maxPen, best, second, bestHSP;
	bool haveTop, top_plus;
	uint32_t top_db;
	int top_nops;
	uint32_t status;
	// pair mode (kernels_pe_slow.hip; hit_score != nullptr): every hit keeps its score, its strand and its path -- FindPairs
	// may make any hit the top hit (state2.cpp:20-85, search2.cpp:49-56).  Paths go to an arena, one after the other.
	int16_t *hit_score = nullptr;
	uint8_t *hit_plus = nullptr;
	uint32_t *hit_poff = nullptr, *hit_pn = nullptr;
	uint16_t *parena = nullptr;
	uint32_t parena_used = 0, parena_cap = 0;
	int topHit = -1;
	int last_added = -1;  // index of the hit the last add_hit made, or -1

	__device__ SlowWave(const DevIndex &X_, const urmapx_params &P_, int lane_) : X(X_), P(P_), lane(lane_) {}

	// state1.cpp:230-239
	__device__ bool overlaps_hit(uint32_t db) const {
		bool ov = false;
		for (int i = lane; i < hitCount; i += 64) ov |= (hit_db[i] >> 6) == (db >> 6);
		return __ballot(ov) != 0;
	}

	// state1.cpp:508-551.  path (if any) is in `cand` with cand_nops runs.
	__device__ void add_hit(uint32_t db, bool plus, int score, int cand_nops) {
		last_added = -1;
		if (score < 10) return;
		if (overlaps_hit(db)) return;
		const int mp = (QL - score) - 2 * P.mismatch_score;
		if (mp < maxPen) maxPen = mp;
		bool newTop = false;
		if (score > best) { second = best; best = score; newTop = true; }
		else if (score == best) second = score;
		else {
			if (score < best - SECONDARY_HIT_MAX_DELTA) return;
			if (score > second) second = score;
		}
		if ((uint32_t)hitCount >= SLOW_HITCAP) { status |= URMAPX_ST_HIT_OVERFLOW; return; }
		if (lane == 0) hit_db[hitCount] = db;
		if (hit_score) {
			if (parena_used + (uint32_t)cand_nops > parena_cap) { status |= URMAPX_ST_PATH_OVERFLOW; cand_nops = 0; }
			if (lane == 0) {
				hit_score[hitCount] = (int16_t)score; hit_plus[hitCount] = plus ? 1 : 0;
				hit_poff[hitCount] = parena_used; hit_pn[hitCount] = (uint32_t)cand_nops;
			}
			for (int t = lane; t < cand_nops; t += 64) parena[parena_used + t] = cand[t];
			parena_used += (uint32_t)cand_nops;
			if (newTop) topHit = hitCount;
		}
		last_added = hitCount;
		++hitCount;
		if (newTop) {
			haveTop = true; top_db = db; top_plus = plus; top_nops = cand_nops;
			for (int t = lane; t < cand_nops; t += 64) top[t] = cand[t];
		}
		__syncthreads();
	}

	// state1.cpp:553-591 (OverlapsHSP: the first HSP on the same diagonal, state1.cpp:241-252)
	__device__ void add_hsp(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		if (score < best - 4) return;
		const uint32_t diag = startdb - startq;
		for (int base = 0; base < hspCount; base += 64) {
			const int i = base + lane;
			const bool eq = i < hspCount && hsp_db[i] - hsp_q[i] == diag;
			const uint64_t m = __ballot(eq);
			if (m) {
				const int k = base + __builtin_ctzll(m);
				if (score > hsp_score[k] && lane == 0) {
					hsp_q[k] = startq; hsp_db[k] = startdb; hsp_len[k] = len; hsp_score[k] = score; hsp_fl[k] = plus ? 1 : 0;
				}
				__syncthreads();
				return;
			}
		}
		if ((uint32_t)hspCount >= hspcap) { status |= URMAPX_ST_HSP_OVERFLOW; return; }
		if (lane == 0) {
			const int k = hspCount;
			hsp_q[k] = startq; hsp_db[k] = startdb; hsp_len[k] = len; hsp_score[k] = score; hsp_fl[k] = plus ? 1 : 0;
		}
		__syncthreads();
		++hspCount;
		if (score > bestHSP) bestHSP = score;
	}

	// AddHSPScan (extendscan.cpp:8-49): as AddHSPX without the best-score floor; returns the HSP's index
	__device__ int add_hsp_scan(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		const uint32_t diag = startdb - startq;
		for (int base = 0; base < hspCount; base += 64) {
			const int i = base + lane;
			const bool eq = i < hspCount && hsp_db[i] - hsp_q[i] == diag;
			const uint64_t m = __ballot(eq);
			if (m) {
				const int k = base + __builtin_ctzll(m);
				if (score > hsp_score[k] && lane == 0) {
					hsp_q[k] = startq; hsp_db[k] = startdb; hsp_len[k] = len; hsp_score[k] = score; hsp_fl[k] = plus ? 1 : 0;
				}
				__syncthreads();
				return k;
			}
		}
		if ((uint32_t)hspCount >= hspcap) { status |= URMAPX_ST_HSP_OVERFLOW; return -1; }
		const int k = hspCount;
		if (lane == 0) { hsp_q[k] = startq; hsp_db[k] = startdb; hsp_len[k] = len; hsp_score[k] = score; hsp_fl[k] = plus ? 1 : 0; }
		__syncthreads();
		++hspCount;
		if (score > bestHSP) bestHSP = score;
		return k;
	}

	// extendpen.cpp:9-95, one candidate: the lanes compare 64 positions at a time, the walk over the mismatches is scalar
	// scan = true: ExtendScan (extendscan.cpp:51-187) -- no overlap test, the leftward walk does not add to the penalty (the
	// reference's omission, kept), HSPs from 2 W on, each aligned at once
	__device__ int extend_pen(uint32_t seedq, uint32_t seeddb, bool plus, bool scan = false) {
		if (seeddb < seedq) return -1;
		const uint32_t dblo = seeddb - seedq;
		if (!scan && overlaps_hit(dblo)) return -1;
		const uint8_t *Q = q[plus ? 0 : 1];
		const uint8_t *T = gseq + dblo;
		const int mis = P.mismatch_score, xdrop = P.xdrop;
		const int minhsp = (int)((uint32_t)P.min_hsp_score_pct * (uint32_t)QL / 100.0);
		int pen = 0, score = W, bst = 0;
		int endpos = (int)seedq + W - 1;
		{
			int cur = endpos + 1;
			bool stop = false;
			for (int base = cur; base < QL && !stop; base += 64) {
				const int p = base + lane;
				uint64_t m = __ballot(p < QL && Q[p] != T[p]);
				const int lim = base + 64 < QL ? base + 64 : QL;
				while (m) {
					const int mp = base + __builtin_ctzll(m);
					m &= m - 1;
					if (mp > cur) {
						score += mp - cur;
						if (score > bst) { bst = score; endpos = mp - 1; }
					}
					pen -= mis;
					if (pen > maxPen) return -1;
					score += mis;
					cur = mp + 1;
					if (bst - score > xdrop) { stop = true; break; }
				}
				if (!stop && lim > cur) {
					score += lim - cur;
					if (score > bst) { bst = score; endpos = lim - 1; }
					cur = lim;
				}
			}
		}
		int startpos = (int)seedq;
		{
			int cur = startpos - 1;  // next position to look at, going down
			bool stop = false;
			for (int hi = cur; hi >= 0 && !stop; hi -= 64) {
				const int p = hi - lane;
				uint64_t m = __ballot(p >= 0 && Q[p] != T[p]);
				const int lim = hi - 64 >= -1 ? hi - 64 : -1;  // first position below this chunk
				while (m) {
					const int mp = hi - __builtin_ctzll(m);
					m &= m - 1;
					if (mp < cur) {
						score += cur - mp;
						if (score > bst) { bst = score; startpos = mp + 1; }
					}
					if (!scan) pen -= mis;
					if (pen > maxPen) return -1;
					score += mis;
					cur = mp - 1;
					if (bst - score > xdrop) { stop = true; break; }
				}
				if (!stop && cur > lim) {
					score += cur - lim;
					if (score > bst) { bst = score; startpos = lim + 1; }
					cur = lim;
				}
			}
		}
		if (startpos == 0 && endpos == QL - 1) {
			add_hit(dblo, plus, bst, 0);
			return bst;
		}
		if (scan) {
			if (bst < 2 * W) return -1;
			const int k = add_hsp_scan((uint32_t)startpos, dblo + (uint32_t)startpos, plus, (uint32_t)(endpos - startpos + 1), bst);
			if (k >= 0) align_hsp(k);
			return -2;
		}
		if (bst >= minhsp) {
			add_hsp((uint32_t)startpos, dblo + (uint32_t)startpos, plus, (uint32_t)(endpos - startpos + 1), bst);
			return -2;
		}
		return -1;
	}

	__device__ bool window_has_pad(uint32_t tlo, uint32_t tl) const {
		bool gap = false;
		for (uint32_t i = lane; i < tl; i += 64) gap |= gseq[tlo + i] == '-';
		return __ballot(gap) != 0;
	}

	// alignhsp.cpp:60-172
	__device__ void align_hsp(int k) {
		const uint32_t fl = hsp_fl[k];
		if (fl & 2u) return;  // m_Aligned
		const uint32_t startdb = hsp_db[k];
		const int startq = (int)hsp_q[k], len = (int)hsp_len[k], hscore = hsp_score[k];
		const bool plus = (fl & 1u) != 0;
		__syncthreads();
		if (lane == 0) hsp_fl[k] = (uint8_t)(fl | 2u);
		__syncthreads();
		int totalPen = len - hscore;
		int totalScore = hscore;
		if (totalPen > maxPen) return;
		const int BR = 2 * (int)P.band_radius;
		const uint32_t TL = X.seqDataSize;
		uint32_t combinedTLo = startdb;
		const uint8_t *Q = q[plus ? 0 : 1];
		int nL = 0, nR = 0, rtrim = 0;
		const VPar VP(P);
		const WideScratch wsv = ws;
		uint32_t vst = 0;
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
			if (window_has_pad(tlo, tl)) return;
			RevOps R;
			R.ops = left ? ropsL : ropsR;
			R.cap = (int)pathcap;
			int score = (int)viterbi_wave<false>(VP, fq, fql, gseq + tlo, (int)tl, left, !left, tb, tb_rows8, wsv, R, vst, lane);
			status |= vst;
			if (left) {
				nL = R.n;
				// TrimLeftIs (pathinfo.cpp:153-171): the leading I run is the last run in traceback order
				int nTrimI = 0;
				if (nL > 0) {
					const uint32_t lastop = ropsL[nL - 1];
					if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --nL; }
				}
				combinedTLo = tlo + (uint32_t)nTrimI;
			} else {
				nR = R.n;
				// TrimRightIs (pathinfo.cpp:173-190): trailing I run = first run in traceback order, never the whole path
				if (nR > 1 && (ropsR[0] & 3u) == OP_I) rtrim = 1;
			}
			const int allGap = P.gap_open_score + (fql - 1) * P.gap_ext_score;
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
			// a run is 14 bits of length + 2 of kind: a longer one (reads beyond 16 383 bases) is stored as several runs of the same
			// kind, which every reader of a path merges again
			while (clen > 0) {
				const int piece = clen > 16383 ? 16383 : clen;
				if (nc < (int)pathcap) { if (lane == 0) cand[nc] = (uint16_t)((piece << 2) | cop); ++nc; } else ovf = true;
				clen -= piece;
			}
			cop = op; clen = l;
		};
		for (int t = nL - 1; t >= 0; --t) { const uint32_t o = ropsL[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(OP_M, len);
		for (int t = nR - 1; t >= rtrim; --t) { const uint32_t o = ropsR[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(-2, 1);  // flush
		if (ovf) { status |= URMAPX_ST_PATH_OVERFLOW; return; }
		__syncthreads();
		add_hit(combinedTLo, plus, totalScore, nc);
	}

	// search1m6.cpp:9-33
	__device__ uint32_t calc_mapq() const {
		if (hitCount == 0) return 0;
		if (best <= 0) return 0;
		const double bp = (double)QL;
		double sec = (double)second;
		if (sec < bp / 2.0) {
			sec = bp / 2.0;
			if ((double)best <= sec) return 0;
		}
		const double fract = (double)best / bp;
		double drop = (double)best - sec;
		if (drop > 40) drop = 40;
		double x = drop * fract;
		x = x * fract;
		uint32_t mapq = (uint32_t)x;
		if (mapq > 40) mapq = 40;
		return mapq;
	}

	// State1::SetSlotsVec (state1.cpp:396-438) for both strands + GetBlob (ufindex.h:184-187) for every k-mer
	__device__ void probe_all() {
		auto planes = [&](int c, uint64_t &lo, uint64_t &hi, uint64_t &inv, uint64_t &invm) {
			const int p = 64 * c + lane;
			const uint32_t ch = p < QL ? q[0][p] : 0u;
			const uint32_t L = p < QL ? letter_of(ch) : 4u;
			lo = __ballot(L & 1u);
			hi = __ballot((L >> 1) & 1u);
			inv = __ballot(L > 3u);
			invm = __ballot(L > 3u || ch == 'u');
		};
		uint64_t lo0, hi0, inv0, invm0;
		planes(0, lo0, hi0, inv0, invm0);
		for (int c = 0; 64 * c < nwords; ++c) {
			uint64_t lo1 = 0, hi1 = 0, inv1 = ~0ull, invm1 = ~0ull;
			if (64 * (c + 1) < QL) planes(c + 1, lo1, hi1, inv1, invm1);
			uint64_t sp, sm;
			bool vp, vm;
			const int p = 64 * c + lane;
			kmer_slots(X, lo0, hi0, inv0, invm0, lo1, hi1, inv1, invm1, lane, (uint32_t)p, (uint32_t)nwords, sp, sm, vp, vm);
			if (p < nwords) {
				uint32_t tp = TALLY_FREE, pp = 0, tm = TALLY_FREE, pm = 0;
				if (vp) load_slot(gblob, sp, tp, pp);
				if (vm) load_slot(gblob, sm, tm, pm);
				const int pmn = nwords - 1 - p;  // the reverse-complement k-mer over the same bases sits at this minus-strand position
				slots[0][p] = vp ? sp : ~0ull; tal[0][p] = (uint8_t)tp; pos[0][p] = pp;
				slots[1][pmn] = vm ? sm : ~0ull; tal[1][pmn] = (uint8_t)tm; pos[1][pmn] = pm;
			}
			lo0 = lo1; hi0 = hi1; inv0 = inv1; invm0 = invm1;
		}
		__syncthreads();
	}

	// UFIndex::GetRow_Blob (ufindex.cpp:883-943) for 64 k-mers at once: lane l walks the chain of position p (valid lanes
	// only), row l goes to rows[l * SLOW_ROW_CAP ..]; returns the row length of this lane
	__device__ int get_rows(int s, int p, bool valid) {
		uint32_t T = 0, ps = 0;
		uint64_t sl = 0;
		if (valid) { T = tal[s][p]; ps = pos[s][p]; sl = slots[s][p]; }
		bool act = valid && (T & TALLY_MY_BIT) != 0;  // TallyOther: row length 0
		const uint64_t N = X.slotCount;
		const int maxIx = (int)X.maxIx;
		int rl = 0;
		uint32_t *rs = rows + lane * SLOW_ROW_CAP;
		while (__ballot(act)) {
			if (act) {
				rs[rl] = ps;
				++rl;
				if (rl == maxIx || rl >= SLOW_ROW_CAP) act = false;
				else if (T == TALLY_PLUS1 || T == TALLY_BOTH1) { rl = 1; act = false; }
				else if (T == TALLY_END) act = false;
				else if (T == TALLY_LONG_MINE || T == TALLY_LONG_OTHER) {
					const uint64_t slotA = addmod(sl, ps & 0xFFFFu, N);
					sl = addmod(slotA, ps >> 16, N);
					uint32_t tA, pA;
					load_slot(gblob, slotA, tA, pA);
					rs[rl - 1] = pA;
				} else
					sl = addmod(sl, T & TALLY_NEXT_MASK, N);
				if (act) load_slot(gblob, sl, T, ps);
			}
		}
		__syncthreads();
		return rl;
	}

	// search1m6.cpp:35-277; returns the phase that returned
	__device__ int search_lo() {
		const int minScore1 = QL + P.xphase1 * P.mismatch_score;
		const int minScore3 = QL + P.xphase3 * P.mismatch_score;
		const int minScore4 = QL + P.xphase4 * P.mismatch_score;
		const int termHSP3 = (QL * P.term_hsp_score_pct_phase3) / 100;
		probe_all();
		// phases 1 and 2: BOTH1 seeds on / off the stride W, plus strand first at every position
		for (int ph = 1; ph <= 2; ++ph) {
			for (int base = 0; base < nwords; base += 64) {
				const int p = base + lane;
				const bool sel = p < nwords && ((p % W == 0) == (ph == 1));
				uint64_t mp = __ballot(sel && tal[0][p] == TALLY_BOTH1), mm = __ballot(sel && tal[1][p] == TALLY_BOTH1);
				uint64_t any = mp | mm;
				while (any) {
					const int b = __builtin_ctzll(any);
					any &= any - 1;
					const int qp = base + b;
					if ((mp >> b) & 1ull) { if (extend_pen((uint32_t)qp, pos[0][qp], true) >= minScore1) return ph; }
					if ((mm >> b) & 1ull) { if (extend_pen((uint32_t)qp, pos[1][qp], false) >= minScore1) return ph; }
				}
			}
		}
		// phase 3
		if (bestHSP > termHSP3) {
			for (int k = 0; k < hspCount; ++k) align_hsp(k);
			if (best >= minScore1) return 3;
		}
		// phase 4: chain rows of length <= 2; longer ones are put off (per strand, in query order)
		int ntodo[2] = {0, 0};
		for (int s = 0; s < 2; ++s) {
			for (int base = 0; base < nwords; base += 64) {
				const int p = base + lane;
				bool want = false;
				if (p < nwords) {
					const uint32_t T = tal[s][p];
					want = T != TALLY_FREE && T != TALLY_BOTH1 && (T & TALLY_MY_BIT) != 0;
				}
				uint64_t m = __ballot(want);
				if (!m) continue;
				const int rl = get_rows(s, p, want);
				while (m) {
					const int b = __builtin_ctzll(m);
					m &= m - 1;
					const int n = rdlane(rl, b);
					if (n > 2) {
						if (lane == 0) todo[s][ntodo[s]] = (uint32_t)(base + b);
						++ntodo[s];
						continue;
					}
					for (int k = 0; k < n; ++k) extend_pen((uint32_t)(base + b), rows[b * SLOW_ROW_CAP + k], s == 0);
				}
				__syncthreads();
			}
		}
		__syncthreads();
		if (best >= minScore3) return 4;
		// phase 5: the longer rows
		for (int s = 0; s < 2; ++s) {
			for (int base = 0; base < ntodo[s]; base += 64) {
				const int i = base + lane;
				const bool valid = i < ntodo[s];
				const int p = valid ? (int)todo[s][i] : 0;
				const int rl = get_rows(s, p, valid);
				const int nb = ntodo[s] - base < 64 ? ntodo[s] - base : 64;
				for (int b = 0; b < nb; ++b) {
					const int n = rdlane(rl, b), qp = rdlane(p, b);
					for (int k = 0; k < n; ++k) extend_pen((uint32_t)qp, rows[b * SLOW_ROW_CAP + k], s == 0);
				}
				__syncthreads();
			}
		}
		if (best >= minScore4) return 5;
		// phase 6
		for (int k = 0; k < hspCount; ++k) align_hsp(k);
		return 6;
	}
};

}  // namespace urx
