// viterbi_dev.h -- banded Viterbi + traceback on one wavefront; see kernels.hip for the overview.
#pragma once
#include <type_traits>
#include "dev_common.h"

namespace urx {

// ------------------------------------------------------------------------------------------------
// banded Viterbi on one wavefront (lane = diagonal), viterbi.cpp:11-261 + tracebackbitmem.cpp:8-75
// ------------------------------------------------------------------------------------------------
// Per-wavefront global scratch of the wide-band fallback: DP rows and a byte-per-cell trace matrix.
struct WideScratch {
	float *Mr;    // lb_cap + 4 floats, Mr[j + 1] = Mrow[j] (j >= -1)
	float *Dr;    // lb_cap + 4 floats
	uint8_t *TB;  // (la_cap + 1) x (lb_cap + 1) bytes
	int la_cap, lb_cap;
	__host__ __device__ static size_t bytes(int la_cap, int lb_cap) {
		return (size_t)2 * (lb_cap + 4) * 4 + (size_t)(la_cap + 1) * (lb_cap + 1);
	}
	__device__ void carve(uint8_t *base, int la, int lb) {
		la_cap = la; lb_cap = lb;
		Mr = reinterpret_cast<float *>(base);
		Dr = Mr + (lb + 4);
		TB = reinterpret_cast<uint8_t *>(Dr + (lb + 4));
	}
};

// Scoring constants the DP needs, by value (a reference to the kernel's parameter block would pin it in scratch).
struct VPar {
	int gap_open_score, gap_ext_score, mismatch_score, band_radius;
	__device__ __forceinline__ VPar(const urmapx_params &P)
	    : gap_open_score(P.gap_open_score), gap_ext_score(P.gap_ext_score), mismatch_score(P.mismatch_score),
	      band_radius((int)P.band_radius) {}
};

struct RevOps {  // run-length path, traceback order (last column first)
	uint16_t *ops;  // cap entries (LDS in the fast kernels, global scratch in the general one)
	int cap = OPS_CAP;
	int n;
	int cur_op, cur_len;
	bool overflow;
	__device__ __forceinline__ void begin() { n = 0; cur_op = -1; cur_len = 0; overflow = false; }
	__device__ __forceinline__ void push_run(int op, int len, int lane) {
		// 14 bits of length: a longer run (the general kernel's reads beyond 16 383 bases) becomes several runs of one kind
		do {
			const int piece = len > 16383 ? 16383 : len;
			if (n < cap) {
				if (lane == 0) ops[n] = (uint16_t)((piece << 2) | op);
				++n;
			} else
				overflow = true;
			len -= piece;
		} while (len > 0);
	}
	__device__ __forceinline__ void emit(int op, int lane) {
		if (op == cur_op) ++cur_len;
		else {
			if (cur_len) push_run(cur_op, cur_len, lane);
			cur_op = op; cur_len = 1;
		}
	}
	__device__ __forceinline__ void emit_run(int op, int len, int lane) {
		if (len <= 0) return;
		if (op == cur_op) cur_len += len;
		else {
			if (cur_len) push_run(cur_op, cur_len, lane);
			cur_op = op; cur_len = len;
		}
	}
	__device__ __forceinline__ void end(int lane) {
		if (cur_len) push_run(cur_op, cur_len, lane);
		cur_len = 0; cur_op = -1;
	}
};

// A, B, tb: LDS of this wavefront.  tb holds (tb_rows8*64) dwords: 8 rows of 4-bit trace cells per dword.
// Returns the score; R receives the path in traceback order.  status gets URMAPX_ST_* bits.
__device__ float viterbi_wide(const VPar P, const uint8_t *A, int LA, const uint8_t *B, int LB, bool Left, bool Right,
                              const WideScratch ws, uint32_t *lds, int lds_dwords, RevOps &R, uint32_t &status, int lane);

// ws.la_cap == 0: no wide-band scratch (the band must fit one wavefront).
// B_LDS: B is an LDS array of the caller (reads around it cannot fault, so the row blocks index it without a clamp).  The caller
// keeps at least band_radius + 1 bytes of its LDS IN FRONT of B (B inside a larger array, never a variable of its own that the
// compiler may place at LDS offset 0): the blocks address B[j0 + k] as base (B + j0) + immediate k with j0 down to -(band_radius + 1),
// and a base below zero wraps -- base + k is then out of range, and reads 0, even where j0 + k is a column of the matrix.
// abort_below / aborted (optional): the caller has no use for a score below abort_below (AlignHSP drops the HSP when a
// flank costs more penalty than the cap leaves, alignhsp.cpp:127-130,160-162).  After every block of rows the best
// M / D value of the row plus one point per query letter still to come bounds the final score from above (gaps cost,
// the free end gaps of Left / Right problems add nothing); once that bound is below abort_below the DP stops and sets
// *aborted -- the result would have been discarded anyway.
#ifndef URX_VIT_RB
#define URX_VIT_RB 8  // rows per block of viterbi_wave (8; 4 measured slower)
#endif
#ifndef URX_ABORT_FINE
#define URX_ABORT_FINE 32
#endif
template <bool B_LDS = false, bool EDGE2 = false>
__device__ __forceinline__ float viterbi_wave(const VPar P, const uint8_t *A, int LA, const uint8_t *B, int LB, bool Left, bool Right,
                              uint32_t *tb, int tb_rows8, const WideScratch ws, RevOps &R, uint32_t &status, int lane_in,
                              float abort_below = -3.0e38f, bool *aborted = nullptr, uint32_t *wide_lds = nullptr, int wide_lds_dwords = 0) {
	// the lane index is recomputed here (two mbcnt) instead of using the caller's: that one is live through the whole
	// search kernel, gets spilled in its register-hungry parts, and was then reloaded from scratch in every DP row
	(void)lane_in;
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	R.begin();
	const float GO = (float)P.gap_open_score, GE = (float)P.gap_ext_score;
	if (LA == 0 || LB == 0) {
		if (LA == 0 && LB == 0) return 0.0f;
		if (LA == 0) { R.emit_run(OP_I, LB, lane); R.end(lane); return (float)(P.gap_open_score + (LB - 1) * P.gap_ext_score); }
		R.emit_run(OP_D, LA, lane); R.end(lane);
		return (float)(P.gap_open_score + (LA - 1) * P.gap_ext_score);
	}
	const int Rad = P.band_radius;
	if constexpr (B_LDS) {
		// The contract above, enforced where it used to be a convention: a window within band_radius + 1 bytes of LDS
		// offset 0 would read zeros for columns that ARE in the matrix and score wrongly in silence; here the read is flagged
		// (URMAPX_ST_BAND_TOO_WIDE -> not a valid result, every parity test fails).  One scalar compare per DP.
		if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)B) < (uint32_t)Rad + 1u) {
			status |= URMAPX_ST_BAND_TOO_WIDE;
			return 0.0f;
		}
	}
	int dlo = min(LA, LB), dhi = max(LA, LB);
	dlo = dlo > Rad ? dlo - Rad : 1;
	dhi += Rad;
	if (dhi > LA + LB - 1) dhi = LA + LB - 1;
	const int ND = dhi - dlo + 1;
	// lanes: 0 = column Startj-1, 1..ND = band, ND+1 = column LB; final cells sit at lanes LB-dlo .. LB-dlo+2
	if (ND + 2 > 64 || LB - dlo + 2 > 63 || ((LA + 1 + 7) >> 3) > tb_rows8) {
		// the wide path keeps three per-row arrays in LDS: the narrow path's trace buffer (idle here) unless the caller
		// holds that in global memory and names another LDS area
		if (LA <= ws.la_cap && LB <= ws.lb_cap) {
			// the count, not the pointer, says whether there is an LDS area, and the two kinds of row storage are never one
			// pointer variable: a generic pointer that is LDS on one path and global on the other (or its null test) trips the
			// gfx950 backend once the LDS address is a known constant ("Illegal instruction ... src_shared_base")
			if (wide_lds_dwords > 0) return viterbi_wide(P, A, LA, B, LB, Left, Right, ws, wide_lds, wide_lds_dwords, R, status, lane);
			return viterbi_wide(P, A, LA, B, LB, Left, Right, ws, tb, tb_rows8 * 64, R, status, lane);
		}
		status |= URMAPX_ST_BAND_TOO_WIDE;
		return 0.0f;
	}
	const float flane = (float)lane;
	float M = NEG, Dn = NEG;
	uint32_t acc = 0;
	const int jbase = dlo - 1 + lane - LA;  // column of this lane in row i is jbase + i
	const bool real = lane >= 1 && lane <= ND;
	// one DP row, every special case spelled out (used for row 0)
	auto row_general = [&](int i) {
		const int j = jbase + i;
		const bool active = real && j >= 0 && j < LB;
		const bool semi = (j == LB) && lane >= 1 && lane <= ND + 1;
		const float OpenA = (Left && i == 0) ? 0.0f : GO;
		const float ExtA = (Left && i == 0) ? 0.0f : GE;
		float Mcur = M;
		if (i == 0 && j == 0) Mcur = 0.0f;
		const float D = wave_shl1(Dn, NEG);
		const uint32_t a = A[i];
		const uint32_t b = active ? B[j] : 0u;
		// DPI[i][j] for every lane: max-plus prefix over the lanes to the left
		const float v = active ? (Mcur + OpenA) : NEG;
		const float u = v - ExtA * flane;
		const float Pm = wave_prefix_max(u);
		const float I = wave_shr1(Pm, NEG) + ExtA * (flane - 1.0f);
		uint32_t bits = 0;
		if (active) {
			float xM = Mcur;
			if (D > xM) { xM = D; bits = TB_DM; }
			if (I > xM) { xM = I; bits = TB_IM; }
			M = xM + (float)(a == b ? 1 : P.mismatch_score);
			const bool freeB = (j == 0 && Left);
			const float md = Mcur + (freeB ? 0.0f : GO);
			float Dnew = D + (freeB ? 0.0f : GE);
			if (md >= Dnew) { Dnew = md; bits |= TB_MD; }
			Dn = Dnew;
			const float mi = Mcur + OpenA;
			const float Ie = I + ExtA;
			if (mi >= Ie) bits |= TB_MI;
		} else if (semi) {
			const float md = Mcur + GO;
			float Dnew = D + GE;
			if (md >= Dnew) { Dnew = md; bits = TB_MD; }
			Dn = Dnew;
			M = NEG;
		} else if (lane == 0 && j >= 0) {
			bits = TB_IM;
		}
		acc |= bits << (4 * (i & 7));
		if ((i & 7) == 7) { tb[(i >> 3) * 64 + lane] = acc; acc = 0; }
	};
	// Up to eight rows at a time, select-only (rows >= 1, so none of row 0's special cases): the same recurrences and tie rules
	// as row_general with every lane computing every value and the lane classes (band cell / column LB / column
	// Startj-1 / outside) applied by selects at the end.  No divergent branch, byte loads with immediate offsets, one
	// trace dword per lane per block -- the general row costs ~100 issue slots, half of them scalar.
	const float GOl = Left ? 0.0f : GO, GEl = Left ? 0.0f : GE;  // column 0 of a Left problem opens / extends D for free
	const float MISf = (float)P.mismatch_score;
	const float ONEl = real ? 1.0f : NEG, MISl = real ? MISf : NEG;  // letter scores of this lane in the interior of the band
	const float el = GE * flane, el1 = GE * (flane - 1.0f);
	const uint32_t LBr = real ? (uint32_t)LB : 0u;                          // band cell iff (unsigned)j < LBr
	const int LBs = (lane >= 1 && lane <= ND + 1) ? LB : -(1 << 20);        // column LB cell iff j == LBs
	const uint32_t bits0c = lane == 0 ? TB_IM : 0u;                         // column Startj-1: IM once j >= 0
	// Rows come in blocks of RB = 8 (URX_VIT_RB), the rows of one trace dword.  Blocks of 4 were measured in round 4: the
	// blocks at the band's edges are shorter (a block is "interior" only if no lane enters or leaves the matrix in ANY of its
	// rows: up to 3 + 3 rows lost to alignment instead of 7 + 7) and dp_kernel can look at its stop test twice as often -- and
	// the shorter unrolled bodies schedule worse by more than that.
	constexpr int RB = URX_VIT_RB;
	static_assert(RB == 4 || RB == 8, "a trace dword holds eight rows");
	// rows i0 .. i0+n-1 (1 <= i0, all inside one block: (i0 & (RB-1)) + n <= RB)
	auto rows_upto = [&](int i0, int n) {
		const int j0 = jbase + i0;
		const uint8_t *Ap = A + i0;
		uint32_t av[RB], bv[RB];
		if constexpr (B_LDS) {
			const uint8_t *Bp = B + j0;  // bytes outside [0, LB) are read but never used (LDS reads cannot fault)
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = Ap[k]; bv[k] = Bp[k]; }
		} else {
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = A[min(i0 + k, LA - 1)]; bv[k] = B[min(max(j0 + k, 0), LB - 1)]; }
		}
		const int sh0 = 4 * (i0 & 7);
		uint32_t word = 0;
#pragma unroll
		for (int k = 0; k < RB; ++k) {
			if (k < n) {  // wave-uniform
				const int j = j0 + k;
				const bool act = (uint32_t)j < LBr;
				const bool semi = j == LBs;
				const float D = wave_shl1(Dn, NEG);
				const float Mcur = M;
				const float vraw = Mcur + GO;
				// (a cell outside the matrix is kept dead by its letter score -- rows_interior has the argument; here the set
				// of such cells changes with the row: lanes enter at column 0 and leave behind column LB.  A lane left of
				// column 0 has a dead M, so its v is dead without a mask; its D is never read by a cell of the matrix, nor is
				// anything of a lane right of column LB; the column LB cell itself computes D like any other and gets a dead M.)
				const float Pm = wave_prefix_max(vraw - el);
				const float I = wave_shr1(Pm, NEG) + el1;
				// M state: best of M, D ('>'), I ('>')
				uint32_t bits = D > Mcur ? TB_DM : 0u;
				float xM = fmaxf(Mcur, D);
				bits = I > xM ? TB_IM : bits;
				xM = fmaxf(xM, I);
				const float sc = av[k] == bv[k] ? 1.0f : MISf;
				M = xM + (act ? sc : NEG);
				// D state: open ('>=' wins) or extend; free in column 0 of a Left problem
				const bool col0 = j == 0;
				const float md = Mcur + (col0 ? GOl : GO);
				const float de = D + (col0 ? GEl : GE);
				const uint32_t bMD = md >= de ? TB_MD : 0u;
				Dn = fmaxf(md, de);
				// I state: open ('>=' wins) or extend
				const uint32_t bMI = vraw >= I + GE ? TB_MI : 0u;
				bits |= bMD | bMI;
				bits = act ? bits : (semi ? bMD : (j >= 0 ? bits0c : 0u));
				word |= bits << (4 * k);
			}
		}
		acc |= word << sh0;
		if (((i0 + n) & 7) == 0) { tb[(i0 >> 3) * 64 + lane] = acc; acc = 0; }
	};
	// EDGE2 (dp_kernel): the rows at the band's two edges with the tests of THEIR edge only.  Top (no cell of the block at or
	// beyond column LB): a band lane is in the matrix from the row it reaches column 0 -- one compare of the row number with a
	// lane constant; column 0 is that row itself; lane 0's trace cell turns IM at row int_lo, a wave-uniform test.  Bottom (every
	// band lane at column >= 1, lane 0 at column >= 0): a band lane leaves at the row it reaches column LB -- the same compare
	// with another constant -- and is the column-LB cell in exactly that row; no column-0 cell, so D opens from M + GO as in the
	// interior.  rows_upto spends 9-11 instructions per row on lane classes, these 4-7.
	const int enter_i = -jbase;                                                       // the row in which this lane sits at column 0
	const int enter_real = real ? enter_i : 0x7FFFFFFF;                              // band cell (top) iff i >= enter_real
	const int exit_real = real ? LB - jbase : (int)0x80000000;                       // band cell (bottom) iff i < exit_real
	const int exit_semi = (lane >= 1 && lane <= ND + 1) ? LB - jbase : (int)0x80000000;  // column-LB cell iff i == exit_semi
	auto rows_edge = [&](auto top_tag, int i0, int n, int int_lo_) {
		constexpr bool TOP = decltype(top_tag)::value;
		const int j0 = jbase + i0;
		const uint8_t *Ap = A + i0;
		uint32_t av[RB], bv[RB];
		if constexpr (B_LDS) {
			const uint8_t *Bp = B + j0;
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = Ap[k]; bv[k] = Bp[k]; }
		} else {
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = A[min(i0 + k, LA - 1)]; bv[k] = B[min(max(j0 + k, 0), LB - 1)]; }
		}
		const int sh0 = 4 * (i0 & 7);
		uint32_t word = 0;
#pragma unroll
		for (int k = 0; k < RB; ++k) {
			if (k < n) {  // wave-uniform
				const int i = i0 + k;
				const bool act = TOP ? (enter_real <= i) : (i < exit_real);
				const float D = wave_shl1(Dn, NEG);
				const float Mcur = M;
				const float vraw = Mcur + GO;
				const float Pm = wave_prefix_max(vraw - el);
				const float I = wave_shr1(Pm, NEG) + el1;
				uint32_t bits = D > Mcur ? TB_DM : 0u;
				float xM = fmaxf(Mcur, D);
				bits = I > xM ? TB_IM : bits;
				xM = fmaxf(xM, I);
				const float sc = av[k] == bv[k] ? 1.0f : MISf;
				M = xM + (act ? sc : NEG);
				float md = vraw, de = D + GE;
				if constexpr (TOP) {
					if (Left) {  // compile-time at dp_kernel's call sites: column 0 of a Left problem opens / extends D for free
						const bool col0 = enter_i == i;
						md = Mcur + (col0 ? GOl : GO);
						de = D + (col0 ? GEl : GE);
					}
				}
				const uint32_t bMD = md >= de ? TB_MD : 0u;
				Dn = fmaxf(md, de);
				const uint32_t bMI = vraw >= I + GE ? TB_MI : 0u;
				bits |= bMD | bMI;
				if constexpr (TOP) bits = act ? bits : (i >= int_lo_ ? bits0c : 0u);
				else bits = act ? bits : (exit_semi == i ? bMD : bits0c);
				word |= bits << (4 * k);
			}
		}
		acc |= word << sh0;
		if (((i0 + n) & 7) == 0) { tb[(i0 >> 3) * 64 + lane] = acc; acc = 0; }
	};
	// A block of rows in the interior of the band: no lane enters (column 0) or leaves (column LB) the matrix inside the
	// block, so every band lane is a band cell in every row and the lane classes are constants -- most rows of a long
	// flank are of this kind (lanes enter during the first and leave during the last ~13 rows only).
	auto rows_interior = [&](int i0) {
		const int j0 = jbase + i0;
		const uint8_t *Ap = A + i0;
		uint32_t av[RB], bv[RB];
		if constexpr (B_LDS) {
			const uint8_t *Bp = B + j0;
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = Ap[k]; bv[k] = Bp[k]; }
		} else {
#pragma unroll
			for (int k = 0; k < RB; ++k) { av[k] = Ap[k]; bv[k] = B[min(max(j0 + k, 0), LB - 1)]; }
		}
		uint32_t word = 0;
		// The lanes outside the band (lane 0 = column Startj-1, lanes beyond ND) are kept dead by their letter score, not by
		// selects: a cell's new M is max(M, D, I) + score, and -9e9 as the score of both outcomes leaves -9e9 (or less) there
		// whatever D and I were -- a dead M opens dead gaps (v, and with it the I of every lane to the right, needs no mask),
		// and nobody reads the D of such a lane: lane 0's would go to lane -1, and beyond the band D stays dead from lane 63
		// inwards by induction.  Three selects fewer per cell row.
#pragma unroll
		for (int k = 0; k < RB; ++k) {
			const float D = wave_shl1(Dn, NEG);
			const float Mcur = M;
			const float vraw = Mcur + GO;
			const float Pm = wave_prefix_max(vraw - el);
			const float I = wave_shr1(Pm, NEG) + el1;
			uint32_t bits = D > Mcur ? TB_DM : 0u;
			float xM = fmaxf(Mcur, D);
			bits = I > xM ? TB_IM : bits;
			xM = fmaxf(xM, I);
			M = xM + (av[k] == bv[k] ? ONEl : MISl);
			const float de = D + GE;
			const uint32_t bMD = vraw >= de ? TB_MD : 0u;  // md = Mcur + GO = vraw: no column-0 cell in the block
			Dn = fmaxf(vraw, de);
			const uint32_t bMI = vraw >= I + GE ? TB_MI : 0u;
			bits |= bMD | bMI;
			word |= (real ? bits : bits0c) << (4 * k);
		}
		if constexpr (RB == 8) tb[(i0 >> 3) * 64 + lane] = word;
		else {
			acc |= word << (4 * (i0 & 7));
			if (((i0 + RB) & 7) == 0) { tb[(i0 >> 3) * 64 + lane] = acc; acc = 0; }
		}
	};
	{
		row_general(0);  // the only row with special cases of its own (free gaps of a Left problem, the origin cell)
		int i = 1;
		// interior blocks: lane 1's column >= 1 at the block's first row, lane ND+1's column < LB at its last
		const int int_lo = LA - dlo + 1, int_hi = LB - (dlo + ND - LA) - (RB - 1);  // first rows i0 with int_lo <= i0 < int_hi qualify
		while (i < LA) {
			if ((i & (RB - 1)) == 0 && i + RB <= LA && i >= int_lo && i < int_hi) {
				rows_interior(i);
				i += RB;
			} else {
				const int n = min(RB - (i & (RB - 1)), LA - i);
				if (EDGE2 && i + n - 1 < int_hi + (RB - 1)) rows_edge(std::true_type{}, i, n, int_lo);  // nothing at or beyond column LB
				else if (EDGE2 && i >= int_lo) rows_edge(std::false_type{}, i, n, int_lo);
				else rows_upto(i, n);
				i += n;
			}
			// the stop test after every block of the first URX_ABORT_FINE rows, after every second one from there on
			if (aborted != nullptr && i < LA && (RB == 8 || i < URX_ABORT_FINE || (i & 7) == 0)) {
				const float best = rdlane(wave_prefix_max(lane >= 1 ? fmaxf(M, Dn) : NEG), 63);  // (lane 0's D is not a cell's)
				if (best + (float)(LA - i) < abort_below) { *aborted = true; URX_SYNC(); return best; }
			}
		}
	}
	// last row of the insert matrix (strict '>' there)
	float FinalI;
	{
		const int jf = dlo - 1 + lane;
		const bool validf = jf < LB;
		const float GapOp = Right ? 0.0f : GO, GapEx = Right ? 0.0f : GE;
		const float Mlast = (lane == 0) ? NEG : M;
		const float v = validf ? (Mlast + GapOp) : NEG;
		const float u = v - GapEx * flane;
		const float Pm = wave_prefix_max(u);
		// cross-lane ops stay outside any lane-dependent select: a DPP read from a lane masked off by EXEC returns the
		// fill value, not that lane's register.  Lane 0 gets NEG from the shift itself (and NEG absorbs the addend).
		const float Ibefore = wave_shr1(Pm, NEG) + GapEx * (flane - 1.0f);
		const float Ie = Ibefore + GapEx;
		uint32_t bits = (validf && v > Ie) ? TB_MI : 0u;
		const float Iafter = fmaxf(v, Ie);
		acc |= bits << (4 * (LA & 7));
		tb[(LA >> 3) * 64 + lane] = acc;
		FinalI = rdlane(Iafter, LB - dlo);
	}
	const float FinalM = rdlane(M, LB - dlo + 1);
	const float FinalD = rdlane(Dn, LB - dlo + 2);
	float Score = FinalM;
	int st = OP_M;
	if (FinalD > Score) { Score = FinalD; st = OP_D; }
	if (FinalI > Score) { Score = FinalI; st = OP_I; }
	URX_SYNC();

	// traceback (tracebackbitmem.cpp:8-75), a whole run per step: lane s looks at the s-th cell in the current direction
	// (M: up the diagonal, D: up the column, I: left along the row), a ballot finds the cell whose trace bits end the run.
	// The trace cell of (row, col) is nibble (row & 7) of tb[(row >> 3) * 64 + diagonal lane].
	int i = LA, j = LB;
	int guard = LA + LB + 2;
	while ((i | j) != 0 && guard-- > 0) {
		int n, ri, cj;
		uint32_t stop;
		if (st == OP_M) { n = min(i, j); ri = i - 1 - lane; cj = j - 1 - lane; stop = TB_DM | TB_IM; }
		else if (st == OP_D) { n = i; ri = i - 1 - lane; cj = j; stop = TB_MD; }
		else { n = j; ri = i; cj = j - 1 - lane; stop = TB_MI; }
		if (n <= 0) break;
		if (n > 64) n = 64;
		uint32_t t = 0;
		if (lane < n) {
			const int l = (LA - ri + cj - dlo + 1) & 63;
			t = (tb[(ri >> 3) * 64 + l] >> (4 * (ri & 7))) & 15u;
		}
		const uint64_t ends = __ballot(lane < n && (t & stop) != 0);
		const int len = ends ? (int)__builtin_ctzll(ends) + 1 : n;  // the cell that ends the run is still in this state
		R.emit_run(st, len, lane);
		int nst = st;
		if (ends) {
			const uint32_t te = rdlane(t, len - 1);
			if (st == OP_M) nst = (te & TB_DM) ? OP_D : OP_I;
			else nst = OP_M;
		}
		if (st == OP_M) { i -= len; j -= len; }
		else if (st == OP_D) i -= len;
		else j -= len;
		st = nst;
	}
	R.end(lane);
	if (R.overflow) status |= URMAPX_ST_PATH_OVERFLOW;
	URX_SYNC();
	return Score;
}

// ------------------------------------------------------------------------------------------------
// two banded problems in one wavefront: the row blocks in the interior of both bands run in packed int16
// ------------------------------------------------------------------------------------------------
// VFlank is viterbi_wave's narrow path as a resumable object: row 0, edge row blocks, interior row blocks, the last
// insert row and the traceback are separate steps, so that a caller can advance two problems side by side
// (viterbi_pair_rows).  Where both stand before eight rows in the interior of their bands -- every band lane is a band
// cell in every row: most rows of a long flank -- the eight rows of BOTH problems are computed by ONE instruction stream
// on packed 16-bit integers (v_pk_add_i16 / v_pk_max_i16; one problem per register half), ~31 instructions per row and
// problem instead of 61.  Exactness: DP values are small integers (|x| <= 6 * 320), -9e9 becomes -32768 and every add is
// saturating, so a dead cell stays far below any live one and every comparison between live values is the comparison
// fp32 makes; a `>` / `>=` between two dead values may come out differently (fp32 absorbs small addends into -9e9, int16
// lets them drift by +1 per row), but a dead cell is never on the path the traceback follows.  At the end of a block
// the state goes back to fp32 with everything below -16000 mapped to -9e9 again, so the rows in fp32 around it -- row 0,
// the band's edges, the last insert row -- see exactly the state the all-fp32 sweep leaves.
typedef short vshort2 __attribute__((ext_vector_type(2)));
typedef unsigned short vushort2 __attribute__((ext_vector_type(2)));

struct VFlank {
	// problem
	const uint8_t *A, *B;  // LDS: query flank, target window (bytes around B may be read, never used)
	int LA, LB;
	bool Left, Right;
	uint32_t *tb;          // LDS, tb_rows8 * 64 dwords
	// geometry
	int dlo, ND, jbase, int_lo, int_hi;
	bool real;
	uint32_t LBr;
	int LBs;
	uint32_t bits0c;
	// state
	float M, Dn;
	uint32_t acc;
	int i;                 // next row
	bool active;           // rows still to do
	bool aborted;
	float abort_below;
	bool may_abort;
	// constants
	float GO, GE, GOl, GEl, MISf, flane, el, el1;
	int lane;

	// false: not a narrow problem with rows to sweep (degenerate, or a band wider than the wavefront): the caller runs
	// viterbi_wave on it instead
	__device__ __forceinline__ bool setup(const VPar P, const uint8_t *A_, int LA_, const uint8_t *B_, int LB_, bool Left_, bool Right_,
	                                      uint32_t *tb_, int tb_rows8, float abort_below_, bool may_abort_) {
		lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		A = A_; B = B_; LA = LA_; LB = LB_; Left = Left_; Right = Right_; tb = tb_;
		active = false; aborted = false; abort_below = abort_below_; may_abort = may_abort_;
		if (LA == 0 || LB == 0) return false;
		const int Rad = P.band_radius;
		dlo = min(LA, LB);
		int dhi = max(LA, LB);
		dlo = dlo > Rad ? dlo - Rad : 1;
		dhi += Rad;
		if (dhi > LA + LB - 1) dhi = LA + LB - 1;
		ND = dhi - dlo + 1;
		if (ND + 2 > 64 || LB - dlo + 2 > 63 || ((LA + 1 + 7) >> 3) > tb_rows8) return false;
		GO = (float)P.gap_open_score; GE = (float)P.gap_ext_score;
		GOl = Left ? 0.0f : GO; GEl = Left ? 0.0f : GE;
		MISf = (float)P.mismatch_score;
		flane = (float)lane;
		el = GE * flane; el1 = GE * (flane - 1.0f);
		jbase = dlo - 1 + lane - LA;
		real = lane >= 1 && lane <= ND;
		LBr = real ? (uint32_t)LB : 0u;
		LBs = (lane >= 1 && lane <= ND + 1) ? LB : -(1 << 20);
		bits0c = lane == 0 ? TB_IM : 0u;
		int_lo = LA - dlo + 1;
		int_hi = LB - (dlo + ND - LA) - 7;
		M = NEG; Dn = NEG; acc = 0; i = 0;
		active = true;
		return true;
	}

	// row 0: the only row with special cases of its own (free gaps of a Left problem, the origin cell)
	__device__ __forceinline__ void row0() {
		const int j = jbase;
		const bool act = real && j >= 0 && j < LB;
		const bool semi = (j == LB) && lane >= 1 && lane <= ND + 1;
		const float OpenA = Left ? 0.0f : GO;
		const float ExtA = Left ? 0.0f : GE;
		float Mcur = M;
		if (j == 0) Mcur = 0.0f;
		const float D = wave_shl1(Dn, NEG);
		const uint32_t a = A[0];
		const uint32_t b = act ? B[j] : 0u;
		const float v = act ? (Mcur + OpenA) : NEG;
		const float u = v - ExtA * flane;
		const float Pm = wave_prefix_max(u);
		const float I = wave_shr1(Pm, NEG) + ExtA * (flane - 1.0f);
		uint32_t bits = 0;
		if (act) {
			float xM = Mcur;
			if (D > xM) { xM = D; bits = TB_DM; }
			if (I > xM) { xM = I; bits = TB_IM; }
			M = xM + (a == b ? 1.0f : MISf);
			const bool freeB = (j == 0 && Left);
			const float md = Mcur + (freeB ? 0.0f : GO);
			float Dnew = D + (freeB ? 0.0f : GE);
			if (md >= Dnew) { Dnew = md; bits |= TB_MD; }
			Dn = Dnew;
			const float mi = Mcur + OpenA;
			const float Ie = I + ExtA;
			if (mi >= Ie) bits |= TB_MI;
		} else if (semi) {
			const float md = Mcur + GO;
			float Dnew = D + GE;
			if (md >= Dnew) { Dnew = md; bits = TB_MD; }
			Dn = Dnew;
			M = NEG;
		} else if (lane == 0 && j >= 0) {
			bits = TB_IM;
		}
		acc |= bits;
		i = 1;
		if (i >= LA) active = false;
	}

	// last row of the insert matrix, final state, traceback (viterbi_wave's tail); returns the score
	__device__ __forceinline__ float finish(RevOps &R, uint32_t &status) {
		R.begin();
		if (aborted) return 0.0f;
		float FinalI;
		{
			const int jf = dlo - 1 + lane;
			const bool validf = jf < LB;
			const float GapOp = Right ? 0.0f : GO, GapEx = Right ? 0.0f : GE;
			const float Mlast = (lane == 0) ? NEG : M;
			const float v = validf ? (Mlast + GapOp) : NEG;
			const float u = v - GapEx * flane;
			const float Pm = wave_prefix_max(u);
			const float Ibefore = wave_shr1(Pm, NEG) + GapEx * (flane - 1.0f);
			const float Ie = Ibefore + GapEx;
			const uint32_t bits = (validf && v > Ie) ? TB_MI : 0u;
			const float Iafter = fmaxf(v, Ie);
			acc |= bits << (4 * (LA & 7));
			tb[(LA >> 3) * 64 + lane] = acc;
			FinalI = rdlane(Iafter, LB - dlo);
		}
		const float FinalM = rdlane(M, LB - dlo + 1);
		const float FinalD = rdlane(Dn, LB - dlo + 2);
		float Score = FinalM;
		int st = OP_M;
		if (FinalD > Score) { Score = FinalD; st = OP_D; }
		if (FinalI > Score) { Score = FinalI; st = OP_I; }
		URX_SYNC();
		int ii = LA, j = LB;
		int guard = LA + LB + 2;
		while ((ii | j) != 0 && guard-- > 0) {
			int n, ri, cj;
			uint32_t stop;
			if (st == OP_M) { n = min(ii, j); ri = ii - 1 - lane; cj = j - 1 - lane; stop = TB_DM | TB_IM; }
			else if (st == OP_D) { n = ii; ri = ii - 1 - lane; cj = j; stop = TB_MD; }
			else { n = j; ri = ii; cj = j - 1 - lane; stop = TB_MI; }
			if (n <= 0) break;
			if (n > 64) n = 64;
			uint32_t t = 0;
			if (lane < n) {
				const int l = (LA - ri + cj - dlo + 1) & 63;
				t = (tb[(ri >> 3) * 64 + l] >> (4 * (ri & 7))) & 15u;
			}
			const uint64_t ends = __ballot(lane < n && (t & stop) != 0);
			const int len = ends ? (int)__builtin_ctzll(ends) + 1 : n;
			R.emit_run(st, len, lane);
			int nst = st;
			if (ends) {
				const uint32_t te = rdlane(t, len - 1);
				if (st == OP_M) nst = (te & TB_DM) ? OP_D : OP_I;
				else nst = OP_M;
			}
			if (st == OP_M) { ii -= len; j -= len; }
			else if (st == OP_D) ii -= len;
			else j -= len;
			st = nst;
		}
		R.end(lane);
		if (R.overflow) status |= URMAPX_ST_PATH_OVERFLOW;
		URX_SYNC();
		return Score;
	}
};

__device__ __forceinline__ vshort2 pk_s2(uint32_t x) { return __builtin_bit_cast(vshort2, x); }
__device__ __forceinline__ uint32_t pk_u(vshort2 x) { return __builtin_bit_cast(uint32_t, x); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ vshort2 pk_dpp(vshort2 v, vshort2 fill) {
	return pk_s2((uint32_t)__builtin_amdgcn_update_dpp((int)pk_u(fill), (int)pk_u(v), CTRL, ROWMASK, 0xF, false));
}
// 0xFFFF in each half where a < b (signed): the saturated difference's sign, spread over the half
__device__ __forceinline__ uint32_t pk_ltm(vshort2 a, vshort2 b) {
	return pk_u(__builtin_elementwise_sub_sat(a, b) >> (vshort2)((short)15));
}
// 0xFFFF in each half where the half of x is zero
__device__ __forceinline__ uint32_t pk_zerom(uint32_t x) {
	const vushort2 t = __builtin_elementwise_min(__builtin_bit_cast(vushort2, x), (vushort2)((unsigned short)1));
	return __builtin_bit_cast(uint32_t, t - (vushort2)((unsigned short)1));
}
__device__ __forceinline__ uint32_t pk_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }  // v_bfi_b32
__device__ __forceinline__ int f2pk(float x) { return x < -30000.0f ? -32768 : (int)x; }
__device__ __forceinline__ float pk2f(int x) { return x < -16000 ? NEG : (float)x; }
__device__ __forceinline__ uint32_t pk_pair(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

// Rows 1 .. LA-1 of two problems (either may be inactive), all of them in packed int16: the two problems advance row by
// row together (both start at row 1, so their eight-row trace blocks stay aligned); a problem that is out of rows, or whose
// score can no longer reach what its caller needs, freezes in its register half while the other goes on.  The lane
// classes of viterbi_wave's select-only rows (band cell / column LB / column 0 / outside) become 16-bit masks per half.
__device__ __forceinline__ void viterbi_pair_rows(VFlank &a, VFlank &b) {
	if (a.active) a.row0();
	if (b.active) b.row0();
	if (!a.active && !b.active) return;
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	const bool ua = a.active, ub = b.active;  // problems that have rows beyond row 0
	const VFlank &g = ua ? a : b;             // scoring constants are the same for both
	const vshort2 NEG2 = (vshort2)((short)-32768);
	const uint32_t GO2 = pk_pair((int)g.GO, (int)g.GO), GE2 = pk_pair((int)g.GE, (int)g.GE);
	const uint32_t GOl2 = pk_pair(ua ? (int)a.GOl : 0, ub ? (int)b.GOl : 0), GEl2 = pk_pair(ua ? (int)a.GEl : 0, ub ? (int)b.GEl : 0);
	const vshort2 ONE2 = (vshort2)((short)1);
	const vshort2 MISM1 = (vshort2)((short)((int)g.MISf - 1));
	const vshort2 el2 = (vshort2)((short)((int)g.GE * lane)), el12 = (vshort2)((short)((int)g.GE * (lane - 1)));
	const uint32_t realm = ((ua && a.real) ? 0x0000FFFFu : 0u) | ((ub && b.real) ? 0xFFFF0000u : 0u);
	const uint32_t semil = ((ua && lane >= 1 && lane <= a.ND + 1) ? 0x0000FFFFu : 0u) | ((ub && lane >= 1 && lane <= b.ND + 1) ? 0xFFFF0000u : 0u);
	const uint32_t b0c = lane == 0 ? (uint32_t)(TB_IM | (TB_IM << 16)) : 0u;
	const vshort2 LB2 = pk_s2(pk_pair(ua ? a.LB : 0, ub ? b.LB : 0));
	const int LAa = ua ? a.LA : 0, LAb = ub ? b.LA : 0;
	vshort2 M = pk_s2(pk_pair(ua ? f2pk(a.M) : -32768, ub ? f2pk(b.M) : -32768));
	vshort2 Dn = pk_s2(pk_pair(ua ? f2pk(a.Dn) : -32768, ub ? f2pk(b.Dn) : -32768));
	uint32_t accA = ua ? a.acc : 0u, accB = ub ? b.acc : 0u;
	bool liveA = ua, liveB = ub;
	int i = 1;
	while (liveA || liveB) {
		const int i0 = i;
		const int lim = max(liveA ? LAa : 0, liveB ? LAb : 0);
		const int n = min(8 - (i0 & 7), lim - i0);
		const uint8_t *Aa = (ua ? a.A : b.A) + i0, *Ab = (ub ? b.A : a.A) + i0;
		const uint8_t *Ba = (ua ? a.B + a.jbase : b.B + b.jbase) + i0, *Bb = (ub ? b.B + b.jbase : a.B + a.jbase) + i0;
		uint32_t av[8], bv[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			av[k] = (uint32_t)Aa[k] | ((uint32_t)Ab[k] << 16);
			bv[k] = (uint32_t)Ba[k] | ((uint32_t)Bb[k] << 16);
		}
		const uint32_t j0 = pk_pair((ua ? a.jbase : 0) + i0, (ub ? b.jbase : 0) + i0);
		uint32_t word = 0;   // both problems' trace nibbles of this block: problem A in the low halves ... see below
		uint32_t wa = 0, wb = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			if (k < n) {  // wave-uniform
				// rows a problem does not have any more (or that lie behind its abort) change nothing in its half
				const uint32_t rowm = ((liveA && i0 + k < LAa) ? 0x0000FFFFu : 0u) | ((liveB && i0 + k < LAb) ? 0xFFFF0000u : 0u);
				const vshort2 jv = pk_s2(j0 + (uint32_t)k * 0x00010001u);
				const uint32_t ge0 = ~pk_ltm(jv, (vshort2)((short)0));
				const uint32_t actm = realm & ge0 & pk_ltm(jv, LB2) & rowm;
				const uint32_t semim = semil & pk_zerom(pk_u(jv) ^ pk_u(LB2)) & rowm;
				const uint32_t col0m = pk_zerom(pk_u(jv));
				const vshort2 D = pk_dpp<0x130, 0xF>(Dn, NEG2);
				const vshort2 Mcur = M;
				const vshort2 vraw = __builtin_elementwise_add_sat(Mcur, pk_s2(GO2));
				const vshort2 v = pk_s2(pk_sel(actm, pk_u(vraw), pk_u(NEG2)));
				vshort2 Pm = __builtin_elementwise_sub_sat(v, el2);
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x111, 0xF>(Pm, NEG2));
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x112, 0xF>(Pm, NEG2));
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x114, 0xF>(Pm, NEG2));
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x118, 0xF>(Pm, NEG2));
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x142, 0xA>(Pm, NEG2));
				Pm = __builtin_elementwise_max(Pm, pk_dpp<0x143, 0xC>(Pm, NEG2));
				const vshort2 I = __builtin_elementwise_add_sat(pk_dpp<0x138, 0xF>(Pm, NEG2), el12);
				// M state: best of M, D ('>'), I ('>')
				const uint32_t dm = pk_ltm(Mcur, D) & 0x00010001u;
				vshort2 xM = __builtin_elementwise_max(Mcur, D);
				const uint32_t im = pk_ltm(xM, I) & 0x00010001u;
				xM = __builtin_elementwise_max(xM, I);
				const vshort2 nz = __builtin_bit_cast(vshort2, __builtin_elementwise_min(__builtin_bit_cast(vushort2, av[k] ^ bv[k]), (vushort2)((unsigned short)1)));
				const vshort2 Mnew = __builtin_elementwise_add_sat(xM, nz * MISM1 + ONE2);
				// D state: open ('>=' wins) or extend; free in column 0 of a Left problem
				const vshort2 md = __builtin_elementwise_add_sat(Mcur, pk_s2(pk_sel(col0m, GOl2, GO2)));
				const vshort2 de = __builtin_elementwise_add_sat(D, pk_s2(pk_sel(col0m, GEl2, GE2)));
				const uint32_t bMD = (~pk_ltm(md, de)) & 0x00040004u;
				const vshort2 Dnew = __builtin_elementwise_max(md, de);
				// I state: open ('>=' wins) or extend
				const uint32_t bMI = (~pk_ltm(vraw, __builtin_elementwise_add_sat(I, pk_s2(GE2)))) & 0x00080008u;
				uint32_t bits = (dm & ~im) | (im << 1) | bMD | bMI;
				M = pk_s2(pk_sel(actm, pk_u(Mnew), pk_sel(semim, pk_u(NEG2), pk_u(M))));
				Dn = pk_s2(pk_sel(actm | semim, pk_u(Dnew), pk_u(Dn)));
				bits = pk_sel(actm, bits, pk_sel(semim, bMD, b0c & ge0 & rowm));
				wa |= (bits & 0xFu) << (4 * k);
				wb |= ((bits >> 16) & 0xFu) << (4 * k);
			}
		}
		(void)word;
		const int sh0 = 4 * (i0 & 7);
		accA |= wa << sh0;
		accB |= wb << sh0;
		i = i0 + n;
		// a problem's block is complete when its rows reach the next multiple of eight (else finish() stores the last, partial one)
		if (liveA) {
			const int na = min(8 - (i0 & 7), LAa - i0);
			if (((i0 + na) & 7) == 0) { a.tb[(i0 >> 3) * 64 + lane] = accA; accA = 0; }
			if (i >= LAa) liveA = false;
		}
		if (liveB) {
			const int nb = min(8 - (i0 & 7), LAb - i0);
			if (((i0 + nb) & 7) == 0) { b.tb[(i0 >> 3) * 64 + lane] = accB; accB = 0; }
			if (i >= LAb) liveB = false;
		}
		// can the final score still reach what the caller needs?  (best value of the row plus one point per query letter to come)
		if ((liveA && a.may_abort) || (liveB && b.may_abort)) {
			vshort2 t = __builtin_elementwise_max(M, Dn);
			t = __builtin_elementwise_max(t, pk_dpp<0x111, 0xF>(t, NEG2));
			t = __builtin_elementwise_max(t, pk_dpp<0x112, 0xF>(t, NEG2));
			t = __builtin_elementwise_max(t, pk_dpp<0x114, 0xF>(t, NEG2));
			t = __builtin_elementwise_max(t, pk_dpp<0x118, 0xF>(t, NEG2));
			t = __builtin_elementwise_max(t, pk_dpp<0x142, 0xA>(t, NEG2));
			t = __builtin_elementwise_max(t, pk_dpp<0x143, 0xC>(t, NEG2));
			const uint32_t top = rdlane(pk_u(t), 63);
			if (liveA && a.may_abort && pk2f((int)(short)(top & 0xFFFFu)) + (float)(LAa - i) < a.abort_below) { a.aborted = true; liveA = false; }
			if (liveB && b.may_abort && pk2f((int)(short)(top >> 16)) + (float)(LAb - i) < b.abort_below) { b.aborted = true; liveB = false; }
		}
	}
	const uint32_t m2 = pk_u(M), d2 = pk_u(Dn);
	if (ua) { a.M = pk2f((int)(short)(m2 & 0xFFFFu)); a.Dn = pk2f((int)(short)(d2 & 0xFFFFu)); a.acc = accA; a.active = false; }
	if (ub) { b.M = pk2f((int)(short)(m2 >> 16)); b.Dn = pk2f((int)(short)(d2 >> 16)); b.acc = accB; b.active = false; }
}

// Wide-band fallback (band wider than one wavefront: a flank window clipped at the end of the sequence store,
// or the paired-end rescue's whole-read DP against a 1 kb window).  Same recurrences and tie rules.  The matrix is
// swept in vertical strips of 64 columns, lane = column: a column's M and D values stay in that lane's registers
// from row to row, the in-row insert chain is a prefix scan inside the strip, and what the next strip needs from
// this one -- per row, the M value of the strip's last column before the row and the insert-chain carry after it --
// goes through a per-row array in LDS (the narrow path's trace buffer, idle here).  Only the trace cells go to
// global memory (byte per cell), write-only until the traceback, which skips along runs 64 cells at a time.
__device__ float viterbi_wide(const VPar P, const uint8_t *A, int LA, const uint8_t *B, int LB, bool Left, bool Right,
                              const WideScratch ws, uint32_t *lds, int lds_dwords, RevOps &R, uint32_t &status, int lane) {
	if (3 * LA > lds_dwords) { status |= URMAPX_ST_BAND_TOO_WIDE; return 0.0f; }
	const float GO = (float)P.gap_open_score, GE = (float)P.gap_ext_score;
	const int Rad = P.band_radius;
	int dlo = min(LA, LB), dhi = max(LA, LB);
	dlo = dlo > Rad ? dlo - Rad : 1;
	dhi += Rad;
	if (dhi > LA + LB - 1) dhi = LA + LB - 1;
	float *Bm = reinterpret_cast<float *>(lds);  // [row] M of the previous strip's last column, as it stood before that row
	float *Bi = Bm + LA;                          // [row] insert-chain carry out of the previous strip
	float *ML = Bi + LA;                          // [row] M (before the row) of the row's last band column
	uint8_t *TB = ws.TB;
	const size_t stride = (size_t)LB + 1;
	const float flane = (float)lane;
	auto range_j = [&](int i, int &Startj, int &Endj) {  // diagbox.h:150-170
		Startj = (dlo + i >= LA) ? dlo + i - LA : 0;
		if (Startj >= LB) Startj = LB - 1;
		Endj = (dhi + i + 1 >= LA) ? dhi + i + 1 - LA : 0;
		if (Endj > LB) Endj = LB;
	};
	int StartjL, EndjL;
	range_j(LA - 1, StartjL, EndjL);
	const float GapOp = Right ? 0.0f : GO, GapEx = Right ? 0.0f : GE;
	float FinalI = NEG, FinalM = NEG, carryF = NEG, MleftF = NEG;
	URX_SYNC();
	const int nstrips = (LB + 63) >> 6;
	for (int s = 0; s < nstrips; ++s) {
		const int j = 64 * s + lane;
		const bool col = j < LB;
		const uint32_t b = col ? B[j] : 0u;
		float Mreg = NEG, Dreg = NEG;  // this column's M / D as the last row that had it in its band left them
		for (int i = 0; i < LA; ++i) {
			int Startj, Endj;
			range_j(i, Startj, Endj);
			if (Endj == 0) continue;
			float bm = NEG, bi = NEG;
			if (s > 0) { bm = Bm[i]; bi = Bi[i]; }
			const bool act = col && j >= Startj && j < Endj;
			const float OpenA = (Left && i == 0) ? 0.0f : GO, ExtA = (Left && i == 0) ? 0.0f : GE;
			const uint32_t a = A[i];
			const float oldM = Mreg;
			float Mcur = wave_shr1(oldM, NEG);
			if (lane == 0) Mcur = bm;
			if (i == 0 && j == Startj) Mcur = 0.0f;
			const float Dcur = Dreg;
			const float v = act ? (Mcur + OpenA) : NEG;
			const float u = v - ExtA * flane;
			const float Pm = wave_prefix_max(u);
			const float Ichain = wave_shr1(Pm, NEG) + ExtA * (flane - 1.0f);
			const float Icarry = bi + ExtA * flane;
			const float I = fmaxf(Ichain, Icarry);
			uint32_t bits = 0;
			float xM = Mcur;
			if (Dcur > xM) { xM = Dcur; bits = TB_DM; }
			if (I > xM) { xM = I; bits = TB_IM; }
			const float Mnew = xM + (float)(a == b ? 1 : P.mismatch_score);
			const bool freeB = (j == 0 && Left);
			const float md = Mcur + (freeB ? 0.0f : GO);
			float Dnew = Dcur + (freeB ? 0.0f : GE);
			if (md >= Dnew) { Dnew = md; bits |= TB_MD; }
			const float mi = Mcur + OpenA;
			const float Ie = I + ExtA;
			if (mi >= Ie) bits |= TB_MI;
			const float Iafter = act ? fmaxf(mi, Ie) : NEG;
			const float om63 = rdlane(oldM, 63), ia63 = rdlane(Iafter, 63);
			const int lastcol = Endj - 1;
			const bool haslast = (lastcol >> 6) == s;
			float ml = 0.0f;
			if (haslast) ml = rdlane(oldM, lastcol & 63);
			if (lane == 0) {
				Bm[i] = om63; Bi[i] = ia63;
				if (haslast) ML[i] = ml;
			}
			if (act) { Mreg = Mnew; Dreg = Dnew; TB[(size_t)i * stride + j] = (uint8_t)bits; }
			if (Startj > 0 && j == Startj - 1) TB[(size_t)i * stride + j] = (uint8_t)TB_IM;
		}
		// last row of the insert matrix for this strip's columns (strict '>' there)
		{
			float mprev = wave_shr1(Mreg, NEG);
			if (lane == 0) mprev = MleftF;
			if (j == StartjL) mprev = NEG;
			const bool actF = col && j >= StartjL && j < EndjL;
			const float v = actF ? (mprev + GapOp) : NEG;
			const float u = v - GapEx * flane;
			const float Pm = wave_prefix_max(u);
			const float Ichain = wave_shr1(Pm, NEG) + GapEx * (flane - 1.0f);  // unconditional: see viterbi_wave
			const float Ibefore = fmaxf(Ichain, carryF + GapEx * flane);
			const float Ie = Ibefore + GapEx;
			const float Iafter = fmaxf(v, Ie);
			if (actF) TB[(size_t)LA * stride + j] = (v > Ie) ? (uint8_t)TB_MI : (uint8_t)0;
			carryF = rdlane(Iafter, 63);
			const int lastl = EndjL - 1 - 64 * s;
			if (lastl >= 0 && lastl < 64) FinalI = rdlane(Iafter, lastl);
			MleftF = rdlane(Mreg, 63);
			if (((LB - 1) >> 6) == s) FinalM = rdlane(Mreg, (LB - 1) & 63);
		}
	}
	URX_SYNC();
	// column LB (D only): a recurrence down the rows over the per-row M values collected above
	float FinalD = NEG;
	for (int i = 0; i < LA; ++i) {
		int Startj, Endj;
		range_j(i, Startj, Endj);
		if (Endj == 0) continue;
		float d = FinalD + GE;
		const float md = ML[i] + GO;
		uint8_t t = 0;
		if (md >= d) { d = md; t = (uint8_t)TB_MD; }
		FinalD = d;
		if (lane == 0) TB[(size_t)i * stride + LB] = t;
	}
	float Score = FinalM;
	int st = OP_M;
	if (FinalD > Score) { Score = FinalD; st = OP_D; }
	if (FinalI > Score) { Score = FinalI; st = OP_I; }
	__threadfence_block();
	URX_SYNC();
	// traceback: 64 cells of the current run direction per step
	int i = LA, j = LB;
	int guard = LA + LB + 2;
	while ((i | j) != 0 && guard-- > 0) {
		int n;           // cells available in this direction
		ptrdiff_t at;    // first cell
		ptrdiff_t step;  // from one cell to the next
		uint32_t stop;   // trace bits that end the run
		if (st == OP_M) { n = min(i, j); at = (ptrdiff_t)((size_t)(i - 1) * stride + (size_t)(j - 1)); step = -(ptrdiff_t)stride - 1; stop = TB_DM | TB_IM; }
		else if (st == OP_D) { n = i; at = (ptrdiff_t)((size_t)(i - 1) * stride + (size_t)j); step = -(ptrdiff_t)stride; stop = TB_MD; }
		else { n = j; at = (ptrdiff_t)((size_t)i * stride + (size_t)(j - 1)); step = -1; stop = TB_MI; }
		if (n <= 0) break;
		if (n > 64) n = 64;
		uint32_t t = 0;
		if (lane < n) t = TB[at + step * lane];
		const uint64_t ends = __ballot(lane < n && (t & stop) != 0);
		const int len = ends ? (int)__builtin_ctzll(ends) + 1 : n;  // the cell that ends the run is still in this state
		R.emit_run(st, len, lane);
		int nst = st;
		if (ends) {
			const uint32_t te = rdlane(t, len - 1);
			if (st == OP_M) nst = (te & TB_DM) ? OP_D : OP_I;
			else nst = OP_M;
		}
		if (st == OP_M) { i -= len; j -= len; }
		else if (st == OP_D) i -= len;
		else j -= len;
		st = nst;
	}
	R.end(lane);
	if (R.overflow) status |= URMAPX_ST_PATH_OVERFLOW;
	URX_SYNC();
	return Score;
}

}  // namespace urx
