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
//                      (search1m6.cpp:9-33) and SetMappedPos (state1.cpp:129-145).  One wavefront per read: the
//                      order-dependent schedule is wave-uniform (scalar), the data-parallel parts use the 64
//                      lanes: ExtendPen compares 64 bases per instruction and walks the mismatch ballot,
//                      the banded DP puts one diagonal on each lane (row sweep; the in-row insert dependency is
//                      a max-plus prefix scan over DPP), hit and HSP lists live one-per-lane in VGPRs.
//
// DP scores are kept in fp32 exactly as the reference does (small integers and a -9e9f "minus infinity" that
// absorbs small addends), so every tie and every trace bit is reproduced without re-deriving integer sentinels.
#include "kernels.h"

namespace urx {

static constexpr uint8_t TALLY_FREE = 0, TALLY_END = 127, TALLY_MY_BIT = 128, TALLY_PLUS1 = 254, TALLY_BOTH1 = 255,
                         TALLY_NEXT_MASK = 127, TALLY_LONG_MINE = 253, TALLY_LONG_OTHER = 125;
static constexpr uint32_t TB_DM = 1, TB_IM = 2, TB_MD = 4, TB_MI = 8;
static constexpr int OP_M = 0, OP_D = 1, OP_I = 2;
static constexpr int SECONDARY_HIT_MAX_DELTA = 12;  // state1.h:16
static constexpr float NEG = -9e9f;                 // MINUS_INFINITY of viterbi.cpp
static constexpr int OPS_CAP = 64;                  // reversed run buffers per flank

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
	uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rdlane(float v, int l) {
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// lane l <- lane l+1 (lane 63 <- fill)
__device__ __forceinline__ float wave_shl1(float v, float fill) {
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xF, 0xF, false));
}
// lane l <- lane l-1 (lane 0 <- fill)
__device__ __forceinline__ float wave_shr1(float v, float fill) {
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float v, float fill) {
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
// inclusive prefix max over the 64 lanes
__device__ __forceinline__ float wave_prefix_max(float v) {
	v = fmaxf(v, dpp_f<0x111, 0xF>(v, NEG));  // row_shr:1
	v = fmaxf(v, dpp_f<0x112, 0xF>(v, NEG));  // row_shr:2
	v = fmaxf(v, dpp_f<0x114, 0xF>(v, NEG));  // row_shr:4
	v = fmaxf(v, dpp_f<0x118, 0xF>(v, NEG));  // row_shr:8
	v = fmaxf(v, dpp_f<0x142, 0xA>(v, NEG));  // row_bcast:15 -> rows 1,3
	v = fmaxf(v, dpp_f<0x143, 0xC>(v, NEG));  // row_bcast:31 -> rows 2,3
	return v;
}

// letter code of an ASCII base, alpha.cpp:1309 (g_CharToLetterNucleo): ACGTU/acgtu -> 0..3, else 4
__device__ __forceinline__ uint32_t letter_of(uint32_t c) {
	uint32_t u = c & 0xDFu;
	return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

// complement char, alpha.cpp:3005 (g_CharToCompChar): IUPAC, case preserving, 'u' and everything else -> '?'
__device__ __forceinline__ uint32_t comp_char(uint32_t c) {
	uint32_t up = c & 0xDFu;
	bool alpha = (up >= 'A' && up <= 'Z') && (c == up || c == (up | 0x20u));
	uint32_t r = '?';
	switch (up) {
	case 'A': r = 'T'; break; case 'B': r = 'V'; break; case 'C': r = 'G'; break; case 'D': r = 'H'; break;
	case 'G': r = 'C'; break; case 'H': r = 'D'; break; case 'K': r = 'M'; break; case 'M': r = 'K'; break;
	case 'N': r = 'N'; break; case 'R': r = 'Y'; break; case 'S': r = 'S'; break; case 'T': r = 'A'; break;
	case 'U': r = 'A'; break; case 'V': r = 'B'; break; case 'W': r = 'W'; break; case 'X': r = 'X'; break;
	case 'Y': r = 'R'; break; default: break;
	}
	if (!alpha || r == '?') return '?';
	if (c != up) {  // lower case
		if (up == 'U') return '?';
		r |= 0x20u;
	}
	return r;
}

__device__ __forceinline__ uint64_t murmur64(uint64_t h) {  // ufindex.h:50-58
	h ^= (h >> 33);
	h *= 0xff51afd7ed558ccdULL;
	h ^= (h >> 33);
	h *= 0xc4ceb9fe1a85ec53ULL;
	h ^= (h >> 33);
	return h;
}

// h % slotCount through the precomputed reciprocal (64-bit urem is a long software loop on CDNA)
__device__ __forceinline__ uint64_t mod_slots(uint64_t h, uint64_t d, uint64_t magic) {
	uint64_t q = __umul64hi(h, magic);
	uint64_t r = h - q * d;
	if (r >= d) r -= d;
	return r;
}

// (a + b) mod N for a < N, small b
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t N) {
	uint64_t x = a + b;
	if (x >= N) { x -= N; if (x >= N) x %= N; }
	return x;
}

// spread the low 32 bits of x to the even bit positions of a 64-bit word
__device__ __forceinline__ uint64_t spread32(uint64_t x) {
	x &= 0xFFFFFFFFull;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}

// 5-byte slot {tally, pos} at blob + 5*slot, fetched as one 4-byte-aligned 8-byte load
__device__ __forceinline__ void load_slot(const uint8_t *blob, uint64_t slot, uint32_t &tally, uint32_t &pos) {
	uint64_t addr = 5ull * slot;
	const uint32_t *p = reinterpret_cast<const uint32_t *>(blob + (addr & ~3ull));
	uint32_t lo = p[0], hi = p[1];
	uint64_t v = (((uint64_t)hi << 32) | lo) >> (8u * (uint32_t)(addr & 3ull));
	tally = (uint32_t)(v & 0xFF);
	pos = (uint32_t)(v >> 8);
}

// wave-uniform bit vector of 64*N bits kept in registers
template <int N>
struct BitVec {
	uint64_t w[N];
	__device__ __forceinline__ void clear() {
#pragma unroll
		for (int c = 0; c < N; ++c) w[c] = 0;
	}
	// smallest set bit >= from, or 64*N
	__device__ __forceinline__ int next_set(int from) const {
		int r = 64 * N;
#pragma unroll
		for (int c = N - 1; c >= 0; --c) {
			uint64_t x = w[c];
			int lo = from - 64 * c;
			if (lo >= 64) x = 0;
			else if (lo > 0) x &= (~0ull << lo);
			if (x) r = 64 * c + __builtin_ctzll(x);
		}
		return r;
	}
	// largest set bit <= from, or -1
	__device__ __forceinline__ int prev_set(int from) const {
		int r = -1;
#pragma unroll
		for (int c = 0; c < N; ++c) {
			uint64_t x = w[c];
			int hi = from - 64 * c;
			if (hi < 0) x = 0;
			else if (hi < 63) x &= (~0ull >> (63 - hi));
			if (x) r = 64 * c + 63 - __builtin_clzll(x);
		}
		return r;
	}
	__device__ __forceinline__ bool test(int i) const {
		bool r = false;
#pragma unroll
		for (int c = 0; c < N; ++c)
			if ((i >> 6) == c) r = (w[c] >> (i & 63)) & 1;
		return r;
	}
	__device__ __forceinline__ void set(int i) {
#pragma unroll
		for (int c = 0; c < N; ++c)
			if ((i >> 6) == c) w[c] |= (1ull << (i & 63));
	}
};

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

	// ballot planes: bit p of lo/hi = letter bits of base p, inv = base p is not ACGTU (or beyond the read)
	// invm: as inv for the reverse-complement strand -- lower-case 'u' complements to '?' (alpha.cpp:3005)
	uint64_t lo[NCH + 1], hi[NCH + 1], inv[NCH + 1], invm[NCH + 1];
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		uint32_t p = 64u * c + lane;
		uint32_t L = 4, ch = 0;
		if (p < QL) { ch = q[p]; L = letter_of(ch); }
		lo[c] = __ballot(L & 1u);
		hi[c] = __ballot((L >> 1) & 1u);
		inv[c] = __ballot(L > 3u);
		invm[c] = __ballot(L > 3u || ch == 'u');
	}
	lo[NCH] = hi[NCH] = 0;
	inv[NCH] = invm[NCH] = ~0ull;

	const uint64_t wmask = (W >= 32) ? 0xFFFFFFFFull : ((1ull << W) - 1ull);
	const uint32_t nwords = QL - (W - 1);
	const uint64_t base2 = 2ull * off;
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		uint32_t p = 64u * c + lane;
		if (p >= nwords) continue;
		uint64_t flo = lo[c] >> lane, fhi = hi[c] >> lane, finv = inv[c] >> lane, finvm = invm[c] >> lane;
		if (lane) {
			flo |= lo[c + 1] << (64 - lane);
			fhi |= hi[c + 1] << (64 - lane);
			finv |= inv[c + 1] << (64 - lane);
			finvm |= invm[c + 1] << (64 - lane);
		}
		flo &= wmask; fhi &= wmask; finv &= wmask; finvm &= wmask;
		const bool valid = (finv == 0), validm = (finvm == 0);
		// plus strand word at query position p: first base is the most significant letter
		uint64_t rlo = __brevll(flo) >> (64 - W), rhi = __brevll(fhi) >> (64 - W);
		uint64_t wp = spread32(rlo) | (spread32(rhi) << 1);
		// reverse-complement word covering the same bases: letters complemented, order already reversed
		uint64_t wm = spread32(~flo & wmask) | (spread32(~fhi & wmask) << 1);
		uint64_t sp = ~0ull, sm = ~0ull;
		uint32_t tp = TALLY_FREE, tm = TALLY_FREE, pp = 0xFFFFFFFFu, pm = 0xFFFFFFFFu;
		if (valid) {
			sp = mod_slots(murmur64(wp & X.shiftMask), X.slotCount, X.slotMagic);
			load_slot(X.blob, sp, tp, pp);
		}
		if (validm) {
			sm = mod_slots(murmur64(wm & X.shiftMask), X.slotCount, X.slotMagic);
			load_slot(X.blob, sm, tm, pm);
		}
		uint64_t ip = base2 + p;
		uint64_t im = base2 + QL + (QL - W - p);
		out.slots[ip] = sp; out.tallies[ip] = (uint8_t)tp; out.positions[ip] = pp;
		out.slots[im] = sm; out.tallies[im] = (uint8_t)tm; out.positions[im] = pm;
	}
}

// ------------------------------------------------------------------------------------------------
// banded Viterbi on one wavefront (lane = diagonal), viterbi.cpp:11-261 + tracebackbitmem.cpp:8-75
// ------------------------------------------------------------------------------------------------
struct RevOps {  // run-length path, traceback order (last column first)
	uint16_t *ops;  // LDS, OPS_CAP entries
	int n;
	int cur_op, cur_len;
	bool overflow;
	__device__ __forceinline__ void begin() { n = 0; cur_op = -1; cur_len = 0; overflow = false; }
	__device__ __forceinline__ void push_run(int op, int len, int lane) {
		if (n < OPS_CAP) {
			if (lane == 0) ops[n] = (uint16_t)((len << 2) | op);
			++n;
		} else
			overflow = true;
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
__device__ float viterbi_wave(const urmapx_params &P, const uint8_t *A, int LA, const uint8_t *B, int LB, bool Left,
                              bool Right, uint32_t *tb, int tb_rows8, RevOps &R, uint32_t &status, int lane) {
	R.begin();
	const float GO = (float)P.gap_open_score, GE = (float)P.gap_ext_score;
	if (LA == 0 || LB == 0) {
		if (LA == 0 && LB == 0) return 0.0f;
		if (LA == 0) { R.emit_run(OP_I, LB, lane); R.end(lane); return (float)(P.gap_open_score + (LB - 1) * P.gap_ext_score); }
		R.emit_run(OP_D, LA, lane); R.end(lane);
		return (float)(P.gap_open_score + (LA - 1) * P.gap_ext_score);
	}
	const int Rad = (int)P.band_radius;
	int dlo = min(LA, LB), dhi = max(LA, LB);
	dlo = dlo > Rad ? dlo - Rad : 1;
	dhi += Rad;
	if (dhi > LA + LB - 1) dhi = LA + LB - 1;
	const int ND = dhi - dlo + 1;
	// lanes: 0 = column Startj-1, 1..ND = band, ND+1 = column LB; final cells sit at lanes LB-dlo .. LB-dlo+2
	if (ND + 2 > 64 || LB - dlo + 2 > 63 || ((LA + 1 + 7) >> 3) > tb_rows8) {
		status |= URMAPX_ST_BAND_TOO_WIDE;
		return 0.0f;
	}
	const float flane = (float)lane;
	float M = NEG, Dn = NEG;
	uint32_t acc = 0;
	const int jbase = dlo - 1 + lane - LA;  // column of this lane in row i is jbase + i
	const bool real = lane >= 1 && lane <= ND;
	for (int i = 0; i < LA; ++i) {
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
		const float Ibefore = (lane == 0) ? NEG : (wave_shr1(Pm, NEG) + GapEx * (flane - 1.0f));
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
	__syncthreads();

	// traceback (wave-uniform; LDS reads are broadcasts)
	int i = LA, j = LB;
	int guard = LA + LB + 2;
	while ((i | j) != 0 && guard-- > 0) {
		R.emit(st, lane);
		int ri, cj;
		if (st == OP_M) { ri = i - 1; cj = j - 1; }
		else if (st == OP_D) { ri = i - 1; cj = j; }
		else { ri = i; cj = j - 1; }
		int l = (LA - ri + cj - dlo + 1) & 63;
		uint32_t t = (tb[(ri >> 3) * 64 + l] >> (4 * (ri & 7))) & 15u;
		t = uni(t);
		if (st == OP_M) { st = (t & TB_DM) ? OP_D : (t & TB_IM) ? OP_I : OP_M; --i; --j; }
		else if (st == OP_D) { st = (t & TB_MD) ? OP_M : OP_D; --i; }
		else { st = (t & TB_MI) ? OP_M : OP_I; --j; }
	}
	R.end(lane);
	if (R.overflow) status |= URMAPX_ST_PATH_OVERFLOW;
	__syncthreads();
	return Score;
}

__global__ __launch_bounds__(64) void viterbi_batch_kernel(urmapx_params P, const uint8_t *__restrict__ a,
                                                           const uint32_t *__restrict__ aoffs,
                                                           const uint8_t *__restrict__ b,
                                                           const uint32_t *__restrict__ boffs,
                                                           const uint8_t *__restrict__ flags, uint32_t n,
                                                           float *scores, uint8_t *status_out, urmapx_path_op *ops_out,
                                                           uint16_t *nops_out) {
	constexpr int MAXL = 448;
	__shared__ uint8_t sA[MAXL], sB[MAXL];
	__shared__ uint32_t tb[(MAXL / 8 + 2) * 64];
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
	if (LA > MAXL || LB > MAXL)
		status = URMAPX_ST_BAND_TOO_WIDE;
	else {
		for (int i = lane; i < LA; i += 64) sA[i] = a[aoffs[k] + i];
		for (int i = lane; i < LB; i += 64) sB[i] = b[boffs[k] + i];
		__syncthreads();
		score = viterbi_wave(P, sA, LA, sB, LB, flags[k] & 1, (flags[k] >> 1) & 1, tb, MAXL / 8 + 2, R, status, lane);
	}
	int nout = R.n;
	if (nout > URMAPX_MAX_PATH_OPS) { status |= URMAPX_ST_PATH_OVERFLOW; nout = 0; }
	if (status) nout = 0;
	// forward order = reversed buffer
	for (int t = lane; t < nout; t += 64) ops_out[(size_t)k * URMAPX_MAX_PATH_OPS + t] = rops[nout - 1 - t];
	if (lane == 0) { scores[k] = score; status_out[k] = (uint8_t)status; nops_out[k] = (uint16_t)nout; }
}

// ------------------------------------------------------------------------------------------------
// kernel B: per-read search
// ------------------------------------------------------------------------------------------------
template <int NCH>
struct SearchWave {
	static constexpr int QMAX = 64 * NCH;
	static constexpr int TB_ROWS8 = QMAX / 8 + 2;

	const DevIndex &X;
	const urmapx_params &P;
	const int lane;
	int QL, W;
	// LDS of this wavefront
	uint8_t *sQ[2];      // [0] = plus (read as given), [1] = minus (reverse complement)
	uint8_t *sT;         // target window
	uint32_t *tb;
	uint16_t *ropsL, *ropsR, *cand, *top;
	// per-lane query bytes, chunk c holds base 64c+lane
	uint32_t qch[2][NCH];
	// hits / HSPs: entry k lives on lane k
	uint32_t hit_db; int hit_score;
	uint32_t hsp_db, hsp_ql; int hsp_score; uint32_t hsp_fl;  // hsp_ql = startq | len<<16, hsp_fl = plus | aligned<<1
	int hitCount, hspCount;
	int maxPen, best, second, bestHSP;
	bool haveTop; uint32_t top_db; bool top_plus; int top_nops;
	uint32_t status;

	__device__ SearchWave(const DevIndex &X_, const urmapx_params &P_, int lane_) : X(X_), P(P_), lane(lane_) {}

	__device__ __forceinline__ bool overlaps_hit(uint32_t db) const {
		return __ballot(lane < hitCount && (hit_db >> 6) == (db >> 6)) != 0;
	}

	// state1.cpp:508-551.  path (if any) is in `cand` with cand_nops runs.
	__device__ void add_hit(uint32_t db, bool plus, int score, int cand_nops) {
		if (score < 10) return;
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
		if (hitCount >= 64) { status |= URMAPX_ST_HIT_OVERFLOW; return; }
		if (lane == hitCount) { hit_db = db; hit_score = score; }
		++hitCount;
		if (newTop) {
			haveTop = true; top_db = db; top_plus = plus; top_nops = cand_nops;
			if (cand_nops) {
				for (int t = lane; t < cand_nops; t += 64) top[t] = cand[t];
				__syncthreads();
			}
		}
	}

	// state1.cpp:553-591
	__device__ void add_hsp(uint32_t startq, uint32_t startdb, bool plus, uint32_t len, int score) {
		if (score < best - 4) return;
		const uint32_t diag = startdb - startq;
		uint64_t m = __ballot(lane < hspCount && (hsp_db - (hsp_ql & 0xFFFFu)) == diag);
		if (m) {
			int k = __builtin_ctzll(m);
			int old = rdlane(hsp_score, k);
			if (score > old && lane == k) { hsp_db = startdb; hsp_ql = startq | (len << 16); hsp_score = score; hsp_fl = plus ? 1u : 0u; }
			return;
		}
		if (hspCount >= 64) { status |= URMAPX_ST_HSP_OVERFLOW; return; }
		if (lane == hspCount) { hsp_db = startdb; hsp_ql = startq | (len << 16); hsp_score = score; hsp_fl = plus ? 1u : 0u; }
		++hspCount;
		if (score > bestHSP) bestHSP = score;
	}

	// extendpen.cpp:9-95
	__device__ int extend_pen(uint32_t seedq, uint32_t seeddb, bool plus) {
		if (seeddb < seedq) return -1;
		const uint32_t dblo = seeddb - seedq;
		if (overlaps_hit(dblo)) return -1;
		const uint8_t *t = X.seq + dblo;
		const int s = plus ? 0 : 1;
		BitVec<NCH> mm;
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			int p = 64 * c + lane;
			bool ne = false;
			if (p < QL) ne = ((uint32_t)t[p] != qch[s][c]);
			mm.w[c] = __ballot(ne);
		}
		const int mis = P.mismatch_score, xdrop = P.xdrop;
		int pen = 0, score = W, bst = 0;
		int endpos = (int)seedq + W - 1;
		int cur = endpos + 1;
		while (cur < QL) {
			int m = mm.next_set(cur);
			if (m > QL) m = QL;
			int run = m - cur;
			if (run > 0) { score += run; if (score > bst) { bst = score; endpos = m - 1; } }
			if (m >= QL) break;
			pen -= mis;
			if (pen > maxPen) return -1;
			score += mis;
			if (bst - score > xdrop) break;
			cur = m + 1;
		}
		int startpos = (int)seedq;
		cur = startpos - 1;
		while (cur >= 0) {
			int m = mm.prev_set(cur);
			int run = cur - m;
			if (run > 0) { score += run; if (score > bst) { bst = score; startpos = m + 1; } }
			if (m < 0) break;
			pen -= mis;
			if (pen > maxPen) return -1;
			score += mis;
			if (bst - score > xdrop) break;
			cur = m - 1;
		}
		if (startpos == 0 && endpos == QL - 1) {
			add_hit(dblo, plus, bst, 0);
			return bst;
		}
		const int minhsp = (int)((uint32_t)P.min_hsp_score_pct * (uint32_t)QL / 100.0);
		if (bst >= minhsp) {
			add_hsp((uint32_t)startpos, dblo + (uint32_t)startpos, plus, (uint32_t)(endpos - startpos + 1), bst);
			return -2;
		}
		return -1;
	}

	// load a target window into LDS; returns true if it contains a '-' pad byte
	__device__ bool load_window(uint32_t tlo, int tl) {
		bool gap = false;
		for (int i = lane; i < tl; i += 64) {
			uint8_t c = X.seq[tlo + i];
			sT[i] = c;
			gap |= (c == '-');
		}
		__syncthreads();
		return __ballot(gap) != 0;
	}

	// alignhsp.cpp:60-172
	__device__ void align_hsp(int k) {
		uint32_t fl = rdlane(hsp_fl, k);
		if (fl & 2u) return;
		if (lane == k) hsp_fl |= 2u;
		const uint32_t ql = rdlane(hsp_ql, k);
		const int startq = (int)(ql & 0xFFFFu), len = (int)(ql >> 16);
		const uint32_t startdb = rdlane(hsp_db, k);
		const int hscore = rdlane(hsp_score, k);
		const bool plus = fl & 1u;
		int totalPen = len - hscore;
		int totalScore = hscore;
		if (totalPen > maxPen) return;
		const int BR = 2 * (int)P.band_radius;
		const uint32_t TL = X.seqDataSize;
		uint32_t combinedTLo = startdb;
		const uint8_t *Q = sQ[plus ? 0 : 1];
		RevOps RL, RR;
		RL.ops = ropsL; RR.ops = ropsR;
		RL.begin(); RR.begin();
		int nTrimI = 0;

		if (startq > 0) {
			if (startdb < (uint32_t)startq) return;
			const int leftQL = startq;
			const uint32_t leftTHi = startdb - 1;
			const uint32_t leftTL = (uint32_t)(leftQL + BR);
			if (leftTL >= leftTHi) return;
			const uint32_t leftTLo = leftTHi - leftTL + 1;
			if (load_window(leftTLo, (int)leftTL)) return;
			int leftScore = (int)viterbi_wave(P, Q, leftQL, sT, (int)leftTL, true, false, tb, TB_ROWS8, RL, status, lane);
			// TrimLeftIs: leading I run = last run in traceback order
			if (RL.n > 0) {
				uint32_t lastop = ropsL[RL.n - 1];
				if ((lastop & 3u) == OP_I) { nTrimI = (int)(lastop >> 2); --RL.n; }
			}
			combinedTLo = leftTLo + (uint32_t)nTrimI;
			int allGap = P.gap_open_score + (leftQL - 1) * P.gap_ext_score;
			if (allGap > leftScore) leftScore = allGap;
			totalScore += leftScore;
			totalPen += leftQL - leftScore;
			if (totalPen > maxPen) return;
		}
		const int rightQLo = startq + len;
		if (rightQLo < QL) {
			const int rightQL = QL - rightQLo;
			const uint32_t rightTLo = startdb + (uint32_t)len;
			uint32_t rightTHi = rightTLo + (uint32_t)rightQL + (uint32_t)BR;
			if (rightTHi >= TL) rightTHi = TL - 1;
			const uint32_t rightTL = rightTHi - rightTLo + 1;
			if (load_window(rightTLo, (int)rightTL)) return;
			int rightScore = (int)viterbi_wave(P, Q + rightQLo, rightQL, sT, (int)rightTL, false, true, tb, TB_ROWS8, RR, status, lane);
			// TrimRightIs: trailing I run = first run in traceback order (never the whole path)
			int r0 = 0;
			if (RR.n > 1 && (ropsR[0] & 3u) == OP_I) r0 = 1;
			int allGap = P.gap_open_score + (rightQL - 1) * P.gap_ext_score;
			if (allGap > rightScore) rightScore = allGap;
			totalScore += rightScore;
			totalPen += rightQL - rightScore;
			if (totalPen > maxPen) return;
			// stash the trim start in cur_len (RR.cur_* are free after end())
			RR.cur_len = r0;
		}
		if (status & (URMAPX_ST_BAND_TOO_WIDE | URMAPX_ST_PATH_OVERFLOW)) return;
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
		for (int t = RR.n - 1; t >= RR.cur_len; --t) { uint32_t o = ropsR[t]; put((int)(o & 3u), (int)(o >> 2)); }
		put(-2, 1);  // flush
		if (ovf) { status |= URMAPX_ST_PATH_OVERFLOW; return; }
		__syncthreads();
		add_hit(combinedTLo, plus, totalScore, nc);
	}

	// search1m6.cpp:9-33
	__device__ uint32_t calc_mapq() const {
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

	// ufindex.cpp:883-943; positions land on lanes 0..K-1 of `row`
	__device__ int get_row(uint64_t slot, uint32_t T, uint32_t pos, uint32_t &row) {
		if ((T & TALLY_MY_BIT) == 0) return 0;
		uint64_t slot2 = slot;
		int K = 0;
		const uint64_t N = X.slotCount;
		for (;;) {
			if (K > 0) {
				uint32_t t2, p2;
				load_slot(X.blob, slot2, t2, p2);
				T = uni(t2); pos = uni(p2);
			}
			if (lane == K) row = pos;
			++K;
			if (K == (int)X.maxIx) return K;
			if (T == TALLY_PLUS1 || T == TALLY_BOTH1) return 1;
			if (T == TALLY_END) return K;
			if (T == TALLY_LONG_MINE || T == TALLY_LONG_OTHER) {
				uint64_t slotA = addmod(slot2, pos & 0xFFFFu, N);
				slot2 = addmod(slotA, pos >> 16, N);
				uint32_t tA, pA;
				load_slot(X.blob, slotA, tA, pA);
				pA = uni(pA);
				if (lane == K - 1) row = pA;
			} else {
				slot2 = addmod(slot2, T & TALLY_NEXT_MASK, N);
			}
			if (K > 64) return K;  // corrupt index guard
		}
	}
};

template <int NCH>
__global__ __launch_bounds__(64) void search_se_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint32_t n, ProbeOut probe,
                                                       urmapx_result *__restrict__ results,
                                                       urmapx_path_op *__restrict__ path_ops, uint32_t *path_used) {
	using SW = SearchWave<NCH>;
	__shared__ uint8_t sQp[SW::QMAX], sQm[SW::QMAX], sT[SW::QMAX + 64];
	__shared__ uint32_t tb[SW::TB_ROWS8 * 64];
	__shared__ uint16_t ropsL[OPS_CAP], ropsR[OPS_CAP], cand[URMAPX_MAX_PATH_OPS], top[URMAPX_MAX_PATH_OPS];

	const int lane = threadIdx.x;
	const uint32_t r = blockIdx.x;
	if (r >= n) return;
	const uint64_t off = offs[r];
	const int QL = (int)(offs[r + 1] - off);
	const int W = (int)X.W;

	urmapx_result res;
	res.dbpos = 0xFFFFFFFFu; res.seq_index = 0xFFFFFFFFu; res.coord = 0xFFFFFFFFu;
	res.score = 0; res.second = 0; res.mapq = 0; res.plus = 0; res.exit_phase = 0; res.status = 0;
	res.hit_count = 0; res.path_nops = 0; res.path_off = 0;
	if (QL < W || QL > SW::QMAX || W > 32) {
		res.status = URMAPX_ST_BAD_LENGTH;
		if (lane == 0) results[r] = res;
		return;
	}

	SW S(X, P, lane);
	S.QL = QL; S.W = W;
	S.sQ[0] = sQp; S.sQ[1] = sQm; S.sT = sT; S.tb = tb;
	S.ropsL = ropsL; S.ropsR = ropsR; S.cand = cand; S.top = top;
	S.hit_db = 0; S.hit_score = 0; S.hsp_db = 0; S.hsp_ql = 0; S.hsp_score = 0; S.hsp_fl = 0;
	S.hitCount = 0; S.hspCount = 0;
	S.maxPen = P.max_penalty; S.best = 0; S.second = 0; S.bestHSP = 0;
	S.haveTop = false; S.top_db = 0; S.top_plus = false; S.top_nops = 0; S.status = 0;

	// query bytes: registers (lane = base mod 64) and LDS, both strands
	const uint8_t *q = bases + off;
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		int p = 64 * c + lane;
		uint32_t cp = 0, cm = 0;
		if (p < QL) {
			cp = q[p];
			cm = comp_char(q[QL - 1 - p]);
			sQp[p] = (uint8_t)cp;
			sQm[p] = (uint8_t)cm;
		}
		S.qch[0][c] = cp;
		S.qch[1][c] = cm;
	}
	__syncthreads();

	const int nwords = QL - (W - 1);
	const uint64_t base2 = 2ull * off;
	// probe results of this read: tally classes as ballot masks, positions per lane
	BitVec<NCH> both1[2], mine[2];
	uint32_t ppos[2][NCH];
#pragma unroll
	for (int s = 0; s < 2; ++s) {
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			int p = 64 * c + lane;
			uint32_t T = 0, ps = 0;
			if (p < nwords) {
				uint64_t idx = base2 + (uint64_t)s * QL + p;
				T = probe.tallies[idx];
				ps = probe.positions[idx];
			}
			ppos[s][c] = ps;
			both1[s].w[c] = __ballot(T == TALLY_BOTH1);
			// phase 4 takes slots that are "mine" but not BOTH1 (search1m6.cpp:186-188)
			mine[s].w[c] = __ballot((T & TALLY_MY_BIT) != 0 && T != TALLY_BOTH1);
		}
	}
	auto pos_at = [&](int s, int qpos) -> uint32_t {
		uint32_t v = 0;
#pragma unroll
		for (int c = 0; c < NCH; ++c)
			if ((qpos >> 6) == c) v = rdlane(s == 0 ? ppos[0][c] : ppos[1][c], qpos & 63);
		return v;
	};

	const int minScore1 = QL + P.xphase1 * P.mismatch_score;
	const int minScore3 = QL + P.xphase3 * P.mismatch_score;
	const int minScore4 = QL + P.xphase4 * P.mismatch_score;
	const int termHSP3 = (QL * P.term_hsp_score_pct_phase3) / 100;
	int phase = 1;
	bool done = false;

	// ---- phase 1 (stride W) and phase 2 (the rest): BOTH1 seeds, plus then minus per position ----
	for (int pass = 0; pass < 2 && !done; ++pass) {
		phase = pass + 1;
		int qp = 0;
		for (;;) {
			int a = both1[0].next_set(qp), b = both1[1].next_set(qp);
			qp = min(a, b);
			if (qp >= nwords) break;
			const bool onStride = (qp % W) == 0;
			if (onStride == (pass == 0)) {
				if (both1[0].test(qp)) {
					int sc = S.extend_pen((uint32_t)qp, pos_at(0, qp), true);
					if (sc >= minScore1) { done = true; break; }
				}
				if (both1[1].test(qp)) {
					int sc = S.extend_pen((uint32_t)qp, pos_at(1, qp), false);
					if (sc >= minScore1) { done = true; break; }
				}
			}
			++qp;
		}
	}
	// ---- phase 3 ----
	if (!done) {
		phase = 3;
		if (S.bestHSP > termHSP3) {
			for (int k = 0; k < S.hspCount; ++k) S.align_hsp(k);
			if (S.best >= minScore1) done = true;
		}
	}
	// ---- phase 4: rows of length <= 2; longer rows deferred ----
	BitVec<NCH> todo[2];
	todo[0].clear(); todo[1].clear();
	if (!done) {
		phase = 4;
		for (int s = 0; s < 2; ++s) {
			int qp = 0;
			for (;;) {
				qp = mine[s].next_set(qp);
				if (qp >= nwords) break;
				const uint64_t idx = base2 + (uint64_t)s * QL + qp;
				const uint64_t slot = uni64(probe.slots[idx]);
				const uint32_t T = uni((uint32_t)probe.tallies[idx]);
				uint32_t row = 0;
				int K = S.get_row(slot, T, pos_at(s, qp), row);
				if (K > 2) todo[s].set(qp);
				else
					for (int t = 0; t < K; ++t) S.extend_pen((uint32_t)qp, rdlane(row, t), s == 0);
				++qp;
			}
		}
		if (S.best >= minScore3) done = true;
	}
	// ---- phase 5 ----
	if (!done) {
		phase = 5;
		for (int s = 0; s < 2; ++s) {
			int qp = 0;
			for (;;) {
				qp = todo[s].next_set(qp);
				if (qp >= nwords) break;
				const uint64_t idx = base2 + (uint64_t)s * QL + qp;
				const uint64_t slot = uni64(probe.slots[idx]);
				const uint32_t T = uni((uint32_t)probe.tallies[idx]);
				uint32_t row = 0;
				int K = S.get_row(slot, T, pos_at(s, qp), row);
				for (int t = 0; t < K; ++t) S.extend_pen((uint32_t)qp, rdlane(row, t), s == 0);
				++qp;
			}
		}
		if (S.best >= minScore4) done = true;
	}
	// ---- phase 6 ----
	if (!done) {
		phase = 6;
		for (int k = 0; k < S.hspCount; ++k) S.align_hsp(k);
	}

	res.mapq = (uint8_t)S.calc_mapq();
	res.score = (int16_t)S.best; res.second = (int16_t)S.second;
	res.hit_count = (uint16_t)S.hitCount; res.exit_phase = (uint8_t)phase; res.status = (uint8_t)S.status;
	// SetMappedPos (state1.cpp:129-145) with PosToCoordL (ufindex.cpp:729-755)
	if (S.haveTop) {
		uint32_t lo = 0, hi = X.seqCount - 1;
		uint32_t found = 0xFFFFFFFFu, coord = 0xFFFFFFFFu, tl = 0;
		while (lo <= hi && hi != 0xFFFFFFFFu) {
			uint32_t k = (lo + hi) / 2;
			uint32_t o = X.seqOffsets[k], sl = X.seqLengths[k];
			if (S.top_db >= o && S.top_db < o + sl) { found = k; coord = S.top_db - o; tl = sl; break; }
			if (S.top_db > o) lo = k + 1;
			else hi = k - 1;
		}
		if (found != 0xFFFFFFFFu && coord + (uint32_t)QL <= tl) {
			res.dbpos = S.top_db; res.seq_index = found; res.coord = coord; res.plus = S.top_plus ? 1 : 0;
			if (S.top_nops > 0) {
				uint32_t po = 0;
				if (lane == 0) po = atomicAdd(path_used, (uint32_t)S.top_nops);
				po = uni(po);
				for (int t = lane; t < S.top_nops; t += 64) path_ops[po + t] = top[t];
				res.path_off = po; res.path_nops = (uint16_t)S.top_nops;
			}
		}
	}
	if (lane == 0) results[r] = res;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int nch_for(uint32_t max_read_len) {
	if (max_read_len <= 192) return 3;
	if (max_read_len <= 320) return 5;
	return 0;
}

hipError_t launch_seed_probe(const DevIndex &X, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t n,
                             uint32_t max_read_len, ProbeOut out, hipStream_t s) {
	if (n == 0) return hipSuccess;
	const int nch = nch_for(max_read_len);
	dim3 block(256), grid((n + 3) / 4);
	if (nch == 3) hipLaunchKernelGGL(seed_probe_kernel<3>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	else hipLaunchKernelGGL(seed_probe_kernel<5>, grid, block, 0, s, X, d_bases, d_offs, n, out);
	return hipGetLastError();
}

hipError_t launch_search_se(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs,
                            uint32_t n, uint32_t max_read_len, ProbeOut probe, urmapx_result *d_results,
                            urmapx_path_op *d_path_ops, uint32_t *d_path_used, hipStream_t s) {
	if (n == 0) return hipSuccess;
	const int nch = nch_for(max_read_len);
	dim3 block(64), grid(n);
	if (nch == 3)
		hipLaunchKernelGGL(search_se_kernel<3>, grid, block, 0, s, X, P, d_bases, d_offs, n, probe, d_results, d_path_ops, d_path_used);
	else
		hipLaunchKernelGGL(search_se_kernel<5>, grid, block, 0, s, X, P, d_bases, d_offs, n, probe, d_results, d_path_ops, d_path_used);
	return hipGetLastError();
}

hipError_t launch_viterbi_batch(const urmapx_params &P, const uint8_t *d_a, const uint32_t *d_aoffs,
                                const uint8_t *d_b, const uint32_t *d_boffs, const uint8_t *d_flags, uint32_t n,
                                float *d_scores, uint8_t *d_status, urmapx_path_op *d_ops, uint16_t *d_nops,
                                hipStream_t s) {
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(viterbi_batch_kernel, dim3(n), dim3(64), 0, s, P, d_a, d_aoffs, d_b, d_boffs, d_flags, n, d_scores,
	                   d_status, d_ops, d_nops);
	return hipGetLastError();
}

}  // namespace urx
