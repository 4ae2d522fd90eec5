// dev_common.h -- device helpers shared by the gfx950 kernels (wave64).
#pragma once
#include "kernels.h"

namespace urx {

static constexpr uint8_t TALLY_FREE = 0, TALLY_END = 127, TALLY_MY_BIT = 128, TALLY_PLUS1 = 254, TALLY_BOTH1 = 255,
                         TALLY_NEXT_MASK = 127, TALLY_LONG_MINE = 253, TALLY_LONG_OTHER = 125;
static constexpr uint32_t TB_DM = 1, TB_IM = 2, TB_MD = 4, TB_MI = 8;
static constexpr int OP_M = 0, OP_D = 1, OP_I = 2;
static constexpr int SECONDARY_HIT_MAX_DELTA = 12;  // state1.h:16
static constexpr float NEG = -9e9f;                 // MINUS_INFINITY of viterbi.cpp
static constexpr int OPS_CAP = 128;                 // reversed run buffers per flank

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
// inclusive prefix max over the 64 lanes: six v_max_f32 with a DPP source each.  Written as one asm block because the
// compiler's form costs five instructions per step (constant fill, nop, v_mov_dpp, a canonicalising max, the max):
// with dst = src1 = the value itself a lane whose DPP source does not exist is simply not written (bound_ctrl:0) and
// keeps its value.  s_nop 1 = the two wait states a DPP read needs after the VALU write of the same register.
#ifndef URX_DPP_NOP
#define URX_DPP_NOP "s_nop 1\n\t"  // build-time knob (debugging): the wait between the steps
#endif
__device__ __forceinline__ float wave_prefix_max(float v) {
	asm volatile(
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    URX_DPP_NOP
	    "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	    "s_nop 1"
	    : "+v"(v));
	return v;
}

// letter code of an ASCII base, alpha.cpp:1309 (g_CharToLetterNucleo): ACGTU/acgtu -> 0..3, else 4
__device__ __forceinline__ uint32_t letter_of(uint32_t c) {
	uint32_t u = c & 0xDFu;
	return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

// Reads are upper-case ACGT almost always, and for those four bytes (c >> 1) & 3 is 0 / 1 / 3 / 2 (A / C / G / T): membership, complement (alpha.cpp:3005) and
// the 4-bit code (seq_code below: A 0, C 1, G 2, T 3) are one byte picked out of a constant -- four instructions where the general functions take a dozen to forty.
// The search kernels test a read once (one ballot) and take these when every byte passes; any other read (N, lower case, IUPAC) takes the general functions.
#ifndef URX_ACGT_FAST
#define URX_ACGT_FAST 1  // 0: the general functions always (A/B builds)
#endif
__device__ __forceinline__ uint32_t acgt_pick(uint32_t table, uint32_t c) { return (table >> (((c >> 1) & 3u) << 3)) & 0xFFu; }
__device__ __forceinline__ bool is_upper_acgt(uint32_t c) { return c == acgt_pick(0x47544341u, c); }
__device__ __forceinline__ uint32_t comp_char_acgt(uint32_t c) { return acgt_pick(0x43414754u, c); }
__device__ __forceinline__ uint32_t seq_code_acgt(uint32_t c) { return acgt_pick(0x02030100u, c); }

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

// (a + b) mod N for a < N, b < 2^16 (a chain step).  No `%`: a 64-bit remainder is a 150-instruction software
// routine on this target and was inlined at every one of the chain walk's 18 call sites (15 KB of a kernel that has
// to stay near the instruction cache's 64 KB).  One conditional subtraction (two selects, no branch) is exact for a
// table of >= 2^16 slots; smaller tables (test indexes) take the loop behind a wave-uniform branch.
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t N) {
	uint64_t x = a + b;
	x = x >= N ? x - N : x;
	if (N < 65536ull)
		while (x >= N) x -= N;
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

// PosToCoordL on the device (search_se_kernel: fill_result_core; search_pe_kernel's output): up to 64 sequences every lane tests one
// (one round of loads instead of the binary search's five or six dependent ones); 0: the binary search always (A/B builds)
#ifndef URX_SEQ_LANES
#define URX_SEQ_LANES 1
#endif

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

// ExtendPen's two x-drop walks (extendpen.cpp:25-78) over a precomputed mismatch bit vector, one candidate per lane.
// `cap` is an UPPER BOUND of m_MaxPenalty at the candidate's turn (the cap only falls, so its value when the batch is
// walked is one): a lane whose penalty passes it fails extendpen.cpp:43-44 / 69-70 at its turn whatever happens before,
// so it stops walking and returns that penalty (> cap); the caller's ordered test drops it.  Lanes under the bound
// return the full walk's result and the caller compares their penalty with the cap in order.
// The walk visits mismatches only.  The loop over the N words is wave-uniform and unrolled; inside a word the set
// bits are consumed one by one (x &= x - 1), so an iteration is ~20 VALU instructions with no word selection.
#ifndef URX_WALK_IN_TEST
#define URX_WALK_IN_TEST 0  // 1: the forward walk tests every mismatch position against the read's length (rounds 2-5)
#endif
template <int N>
__device__ __forceinline__ void xdrop_walk_lane(const uint64_t (&w)[N], int qpos, int W, int QL, int mis, int xdrop, int cap,
                                                int &bst_out, int &startpos_out, int &endpos_out, int &pen_out) {
	// The loop bodies are written with selects only (one divergent loop, no divergent branches inside): a lane that
	// has stopped simply carries x == 0.
	int score = W, bst = 0, pen = 0;
	int endpos = qpos + W - 1;
	int cur = endpos + 1;
	bool alive = cur < QL;
#pragma unroll
	for (int c = 0; c < N; ++c) {
		uint64_t x = w[c];
		const int lo = cur - 64 * c;  // first wanted bit of this word
		const uint64_t keep = lo > 0 ? (~0ull << (lo & 63)) : ~0ull;
		x = (!alive || lo >= 64) ? 0ull : (x & keep);
		while (x) {
			const int m = 64 * c + __builtin_ctzll(x);
			x &= x - 1;
#if URX_WALK_IN_TEST
			const bool in = m < QL;  // padding bits past the read end the word; the tail run below ends the walk
			const int s1 = score + (m - cur);
			const bool nb = in && m > cur && s1 > bst;
			bst = nb ? s1 : bst;
			endpos = nb ? m - 1 : endpos;
			score = in ? s1 + mis : score;
			pen = in ? pen - mis : pen;
			cur = in ? m + 1 : cur;
			const bool stop = in && (bst - score > xdrop || pen > cap);
			alive = alive && !stop;
			x = (stop || !in) ? 0ull : x;
#else
			// (no bit at or beyond QL is ever set: lane_mismatch_planes and lane_mismatch_mask clear them -- the test for them cost six of the loop's 22 instructions)
			const int s1 = score + (m - cur);
			const bool nb = m > cur && s1 > bst;
			bst = nb ? s1 : bst;
			endpos = nb ? m - 1 : endpos;
			score = s1 + mis;
			pen -= mis;
			cur = m + 1;
			const bool stop = bst - score > xdrop || pen > cap;
			alive = alive && !stop;
			x = stop ? 0ull : x;
#endif
		}
	}
	{  // no mismatch left: the run to the end of the read
		const int s1 = score + (QL - cur);
		const bool nb = alive && QL > cur && s1 > bst;
		score = (alive && QL > cur) ? s1 : score;
		bst = nb ? s1 : bst;
		endpos = nb ? QL - 1 : endpos;
	}
	int startpos = qpos;
	cur = startpos - 1;
	alive = cur >= 0 && pen <= cap;
#pragma unroll
	for (int c = N - 1; c >= 0; --c) {
		uint64_t x = w[c];
		const int hi = cur - 64 * c;  // last wanted bit of this word
		const uint64_t keep = hi < 63 ? (~0ull >> ((63 - hi) & 63)) : ~0ull;
		x = (!alive || hi < 0) ? 0ull : (x & keep);
		while (x) {
			const int b = 63 - __builtin_clzll(x);
			const int m = 64 * c + b;
			x ^= 1ull << b;
			const int s1 = score + (cur - m);
			const bool nb = cur > m && s1 > bst;
			bst = nb ? s1 : bst;
			startpos = nb ? m + 1 : startpos;
			score = s1 + mis;
			pen -= mis;
			cur = m - 1;
			const bool stop = bst - score > xdrop || pen > cap;
			alive = alive && !stop;
			x = stop ? 0ull : x;
		}
	}
	{
		const int s1 = score + (cur + 1);  // down to position 0
		const bool nb = alive && cur >= 0 && s1 > bst;
		bst = nb ? s1 : bst;
		startpos = nb ? 0 : startpos;
	}
	bst_out = bst; startpos_out = startpos; endpos_out = endpos; pen_out = pen;
}

// Mismatches outside the seed window [qpos, qpos + W) of a lane's bit vector.  ExtendPen never looks inside the seed
// (extendpen.cpp:24-27), so this is the most its walks can meet: a full-length hit costs exactly -mis times this, and no
// score along the walks exceeds QL minus this.
#ifndef URX_SEED_FUNNEL
#define URX_SEED_FUNNEL 1  // 0: the round-2 form (per word: two 64-bit masks from the seed's bounds, 24 instructions a word)
#endif
template <int N>
__device__ __forceinline__ int mismatches_outside_seed(const uint64_t (&w)[N], int qpos, int W) {
#if URX_SEED_FUNNEL
	// all mismatches, minus those among the W <= 32 seed bits: the two dwords that hold bits qpos .. qpos + 31 picked out of the vector, one funnel shift
	int pc = 0;
#pragma unroll
	for (int c = 0; c < N; ++c) pc += __builtin_popcountll(w[c]);
	const int d = qpos >> 5;
	uint32_t a = 0, b = 0;  // dwords d and d + 1 (0 beyond the vector)
#pragma unroll
	for (int i = 0; i < 2 * N; ++i) {
		const uint32_t dw = (i & 1) ? (uint32_t)(w[i >> 1] >> 32) : (uint32_t)w[i >> 1];
		a = d == i ? dw : a;
		b = d + 1 == i ? dw : b;
	}
	const uint32_t x = __builtin_amdgcn_alignbit(b, a, (uint32_t)qpos & 31u);
	const uint32_t m = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
	return pc - __builtin_popcount(x & m);
#else
	int pc = 0;
#pragma unroll
	for (int c = 0; c < N; ++c) {
		const int lo = qpos - 64 * c, hi = lo + W;  // seed bits of this word: [lo, hi) cut to [0, 64)
		const uint64_t below_hi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
		const uint64_t below_lo = lo >= 64 ? ~0ull : (lo <= 0 ? 0ull : ((1ull << lo) - 1ull));
		pc += __builtin_popcountll(w[c] & ~(below_hi & ~below_lo));
	}
	return pc;
#endif
}

// inclusive prefix sum of a non-negative int over the 64 lanes (DPP, same ladder as wave_prefix_max)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false); }
__device__ __forceinline__ int wave_prefix_sum(int v) {
	v += dpp_i<0x111, 0xF>(v);
	v += dpp_i<0x112, 0xF>(v);
	v += dpp_i<0x114, 0xF>(v);
	v += dpp_i<0x118, 0xF>(v);
	v += dpp_i<0x142, 0xA>(v);
	v += dpp_i<0x143, 0xC>(v);
	return v;
}

// four dwords from a 4-byte aligned address as one global_load_dwordx4
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ u32x4_a4 load4_a4(const uint32_t *p) { return *reinterpret_cast<const u32x4_a4 *>(p); }

// bit 7 of every non-zero byte of x (exact per byte: no carry crosses a byte boundary)
__device__ __forceinline__ uint32_t nonzero_bytes_b7(uint32_t x) {
	return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

// Mismatch bit vector of a whole read against the reference window starting at seq+dblo, computed by ONE lane
// (extendpen.cpp:29-78 compares the same bytes one at a time): bit p = (query[p] != seq[dblo+p]), p < QL.
// q16 = the query strand in LDS, 16-byte aligned.  Unaligned windows are re-aligned with v_alignbyte.
template <int NCH>
__device__ __forceinline__ void lane_mismatch_mask(const uint8_t *__restrict__ seq, uint32_t dblo, const uint8_t *q16,
                                                   int QL, uint64_t (&mm)[NCH]) {
	const uint32_t sh = dblo & 3u;
	const uint32_t *p = reinterpret_cast<const uint32_t *>(seq + (dblo & ~3u));
	const uint4 *q4 = reinterpret_cast<const uint4 *>(q16);
	// all window loads are issued before any is used (16-byte granules up to QL; the sequence store is padded);
	// bits at or beyond QL are cleared at the end
	uint32_t first = p[0];
	u32x4_a4 v[4 * NCH];
#pragma unroll
	for (int g = 0; g < 4 * NCH; ++g)
		if (16 * g < QL) v[g] = load4_a4(p + 1 + 4 * g);  // wave-uniform: the whole wavefront works on one read
	uint32_t prev = first;
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		uint64_t w = 0;
#pragma unroll
		for (int gg = 0; gg < 4; ++gg) {
			const int g = 4 * c + gg;
			if (16 * g < QL) {  // wave-uniform: the whole wavefront works on one read
				const uint4 q = q4[g];
				const uint32_t t0 = __builtin_amdgcn_alignbyte(v[g].x, prev, sh);
				const uint32_t t1 = __builtin_amdgcn_alignbyte(v[g].y, v[g].x, sh);
				const uint32_t t2 = __builtin_amdgcn_alignbyte(v[g].z, v[g].y, sh);
				const uint32_t t3 = __builtin_amdgcn_alignbyte(v[g].w, v[g].z, sh);
				// v_dot4_u32_u8 gathers the per-byte flags (0 or 128) of two dwords into 128 * (8-bit mask)
				uint32_t lo = __builtin_amdgcn_udot4(nonzero_bytes_b7(t0 ^ q.x), 0x08040201u, 0u, false);
				lo = __builtin_amdgcn_udot4(nonzero_bytes_b7(t1 ^ q.y), 0x80402010u, lo, false);
				uint32_t hi = __builtin_amdgcn_udot4(nonzero_bytes_b7(t2 ^ q.z), 0x08040201u, 0u, false);
				hi = __builtin_amdgcn_udot4(nonzero_bytes_b7(t3 ^ q.w), 0x80402010u, hi, false);
				const uint32_t bits = (lo >> 7) | ((hi << 1) & 0xFF00u);
				w |= (uint64_t)bits << (16 * gg);
			}
			prev = v[g].w;
		}
		const int rem = QL - 64 * c;
		if (rem <= 0) w = 0;
		else if (rem < 64) w &= ((1ull << rem) - 1ull);
		mm[c] = w;
	}
}


// ------------------------------------------------------------------------------------------------
// packed copy of m_SeqData for ExtendPen's byte compares (extendpen.cpp:29-78): four bit planes per 32 bases
// ------------------------------------------------------------------------------------------------
// Every byte gets a 4-bit code that is injective on the bytes a genome and a read are made of (ACGTN in either case,
// the '-' pad between sequences); all other bytes share one code on the reference side (15) and another on the read
// side (14).  For a read without "other" bytes -- checked once per read -- code equality IS byte equality: a coded read
// byte never equals a reference byte outside the list, whatever that byte is.  Reads with other bytes (IUPAC codes
// beyond N, 'u') compare ASCII windows as before (lane_mismatch_mask).
// Layout: block b = bases 32b .. 32b+31 = one uint4 {plane 0, 1, 2, 3}, bit i of plane k = bit k of the code of base
// 32b+i: a read's window is 16 bytes per 32 bases, contiguous, and the mismatch bits of 32 positions are
// (t0^q0)|(t1^q1)|(t2^q2)|(t3^q3) after a funnel shift of each plane.
static constexpr uint32_t SEQ_CODE_QOTHER = 14u, SEQ_CODE_TOTHER = 15u;
__device__ __forceinline__ uint32_t seq_code(uint32_t c, uint32_t other) {
	const uint32_t u = c & 0xDFu;
	uint32_t k = u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : u == 'N' ? 4u : 16u;
	k += (c & 0x20u) ? 5u : 0u;  // lower case: 5..9
	return k < 10u ? k : (c == '-' ? 10u : other);
}

// Mismatch bit vector of a whole read against the packed reference window starting at base dblo, computed by ONE lane:
// bit p = (query[p] != seq[dblo+p]), p < QL.  qpl = the query strand's planes in LDS (block j = positions 32j..32j+31).
template <int NCH>
__device__ __forceinline__ void lane_mismatch_planes(const uint4 *__restrict__ seqp, uint32_t dblo, const uint4 *qpl, int QL,
                                                     uint64_t (&mm)[NCH]) {
	const uint32_t sh = dblo & 31u;
	const uint4 *p = seqp + (dblo >> 5);
	uint4 t[2 * NCH + 1];
#pragma unroll
	for (int j = 0; j <= 2 * NCH; ++j)
		if (j == 0 || 32 * (j - 1) < QL) t[j] = p[j];  // wave-uniform: the whole wavefront works on one read
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		uint32_t w2[2] = {0u, 0u};
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int j = 2 * c + h;
			if (32 * j < QL) {
				const uint4 q = qpl[j];
				const uint32_t a0 = __builtin_amdgcn_alignbit(t[j + 1].x, t[j].x, sh);
				const uint32_t a1 = __builtin_amdgcn_alignbit(t[j + 1].y, t[j].y, sh);
				const uint32_t a2 = __builtin_amdgcn_alignbit(t[j + 1].z, t[j].z, sh);
				const uint32_t a3 = __builtin_amdgcn_alignbit(t[j + 1].w, t[j].w, sh);
				w2[h] = (a0 ^ q.x) | (a1 ^ q.y) | (a2 ^ q.z) | (a3 ^ q.w);
			}
		}
		uint64_t w = ((uint64_t)w2[1] << 32) | w2[0];
		const int rem = QL - 64 * c;
		if (rem <= 0) w = 0;
		else if (rem < 64) w &= ((1ull << rem) - 1ull);
		mm[c] = w;
	}
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA: a gather that lands in LDS without passing through registers
// ------------------------------------------------------------------------------------------------
// global_load_lds_dword: every lane names its own source dword, lane l's data lands at lds_dst + 4*l (lds_dst = a
// wave-uniform LDS byte address in M0).  The statement is not counted by the compiler's s_waitcnt bookkeeping: the
// caller waits with wait_vm0() before it reads the destination.  Loads return in order, so any wait the compiler places
// for a younger load of its own also covers these.
__device__ __forceinline__ void glds_dword(const void *gsrc, uint32_t lds_dst) {
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(gsrc), "s"(lds_dst)
	             : "memory");
}
// L2 prefetch (round 6): the 64-byte sectors of [p, p + nbytes) are asked for by LDS-DMA dword loads whose data lands in a sink that
// nobody reads -- no register is named, so nothing of the wave waits for them or can be overwritten by them.  The search kernels
// issue these for the NEXT batch of candidate windows before they walk the current one: the real gather of that batch, one
// batch-time later, finds its sectors in L2 instead of paying an HBM round trip in front of every dependent step.  (Loads return
// in order: a touch must be issued BEHIND the loads the wave is about to wait for, never in front of them.)
__device__ __forceinline__ void glds_touch(const uint8_t *p, int nbytes, uint32_t lds_sink) {  // nbytes wave-uniform, a multiple of 4
	lds_sink = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_sink);  // M0 takes a scalar
	for (int o = 0; o < nbytes; o += 64) glds_dword(p + o, lds_sink);
	glds_dword(p + nbytes - 4, lds_sink);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ uint32_t lds_addr(const T *p) {  // the LDS aperture is 4 GB aligned: the low half of a flat address is the LDS offset
	return uni((uint32_t)(uintptr_t)p);
}
// 16 zero bytes: what lanes without a k-mer gather instead of a slot (tally 0 = TALLY_FREE)
static __device__ uint32_t g_zero16[4];

// An LDS array named by an LDS pointer: 32 bit, accessed with ds_* instructions.  (A plain pointer that the compiler cannot
// trace back to its LDS array -- one loaded from memory -- is a 64-bit generic pointer and every access a flat_* instruction.)
template <class T> using lds_ptr = __attribute__((address_space(3))) T *;
template <class T> __device__ __forceinline__ lds_ptr<T> to_lds(T *p) { return (lds_ptr<T>)p; }
template <class T> __device__ __forceinline__ T *from_lds(lds_ptr<T> p) { return (T *)p; }

// A wave-uniform int the optimiser must treat as new from here on.  The search kernels derive dozens of lane masks
// and conditions from a read's length; LLVM hoists them all to the top of the read (they are loop invariant), where they
// outnumber the 102 SGPRs of a wave several times over, get spilled into VGPR lanes and come back through v_readlane +
// s_nop at every use.  Re-deriving them where they are used is one v_cmp; so the length is "refreshed" at the top of the
// hot loops (measured: search kernel 24.4 -> 22.4 ms per 1 M reads with two such points).
// the same for the lane number: values derived from it (lane + 64 c, LDS addresses, lane masks) are one instruction to make
// and a scratch reload (a memory round trip) to keep
__device__ __forceinline__ int fresh_lane(int x) {
	asm volatile("" : "+v"(x));
	return x;
}
__device__ __forceinline__ int fresh_uniform(int x) {
	asm volatile("" : "+v"(x));
	return __builtin_amdgcn_readfirstlane(x);
}

// Barrier of a one-wavefront block that orders LDS traffic only: __syncthreads() also waits for every global load and
// store in flight (s_waitcnt vmcnt(0)), which is a memory round trip where stores were just issued.
__device__ __forceinline__ void lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// The search kernels' blocks are one wavefront.  URX_LDS_SYNC (build flag, measurement only) makes every barrier of theirs
// LDS-only; the default keeps __syncthreads() wherever lanes exchange data through global scratch.
#ifdef URX_LDS_SYNC
#define URX_SYNC() lds_sync()
#else
#define URX_SYNC() __syncthreads()
#endif

// The slots of the k-mers that start in one 64-position chunk of a read, both strands (State1::SetSlotsVec,
// state1.cpp:396-438, with murmur64 / WordToSlot, ufindex.h:50-65): lane l cuts the W letters of the k-mer starting at
// position 64c + l out of the read's ballot planes (bit p of lo / hi = letter bits of base p, inv / invm = base p cannot
// be part of a plus- / minus-strand k-mer), c0 = the chunk's words, c1 = the next chunk's.  sp = plus-strand slot at that
// position; sm = slot of the reverse-complement k-mer over the same bases (minus-strand position nwords-1-p).
#ifndef URX_KMER_STREAM
#define URX_KMER_STREAM 1  // 0: the round-1 form below (each lane spreads its own planes: four bit interleaves of 15 instructions per k-mer pair)
#endif
__device__ __forceinline__ void kmer_slots(const DevIndex &X, uint64_t lo0, uint64_t hi0, uint64_t inv0, uint64_t invm0, uint64_t lo1,
                                           uint64_t hi1, uint64_t inv1, uint64_t invm1, int lane, uint32_t p, uint32_t nwords,
                                           uint64_t &sp, uint64_t &sm, bool &vp, bool &vm) {
	const uint32_t W = X.W;
	const uint64_t wmask = (W >= 32) ? 0xFFFFFFFFull : ((1ull << W) - 1ull);
#if URX_KMER_STREAM
	// Round 6: the bit interleave of the two letter planes is the same for every lane -- it is done ONCE, on the wave-uniform ballot words (scalar
	// instructions), for the 96 bases a chunk's k-mers can cover: stream S has bit 2i = low letter bit of base i, bit 2i + 1 = high bit; stream T the two
	// swapped.  A lane's words are 2W bits cut out of the streams at bit 2 * lane: the minus-strand word is the complement of S's piece (letters
	// complemented, order as it lies), the plus-strand word the bit reversal of T's piece (first base most significant; reversing the bits of a letter
	// pair whose bits were swapped leaves each letter as it was).
	uint64_t finv = inv0 >> lane, finvm = invm0 >> lane;
	if (lane) {
		finv |= inv1 << (64 - lane);
		finvm |= invm1 << (64 - lane);
	}
	finv &= wmask; finvm &= wmask;
	vp = p < nwords && finv == 0;
	vm = p < nwords && finvm == 0;
	const uint64_t a0 = spread32(lo0), a1 = spread32(lo0 >> 32), a2 = spread32(lo1);
	const uint64_t b0 = spread32(hi0), b1 = spread32(hi0 >> 32), b2 = spread32(hi1);
	const uint64_t S0 = a0 | (b0 << 1), S1 = a1 | (b1 << 1), S2 = a2 | (b2 << 1);
	const uint64_t T0 = b0 | (a0 << 1), T1 = b1 | (a1 << 1), T2 = b2 | (a2 << 1);
	const bool up = lane >= 32;
	const uint32_t sh = (2u * (uint32_t)lane) & 63u;
	const uint64_t sA = up ? S1 : S0, sB = up ? S2 : S1, tA = up ? T1 : T0, tB = up ? T2 : T1;
	uint64_t segS = sA >> sh, segT = tA >> sh;
	if (sh) {
		segS |= sB << (64u - sh);
		segT |= tB << (64u - sh);
	}
	const uint64_t m2 = (W >= 32) ? ~0ull : ((1ull << (2u * W)) - 1ull);
	const uint64_t wp = __brevll(segT & m2) >> (64u - 2u * W);
	const uint64_t wm = ~segS & m2;
#else
	uint64_t flo = lo0 >> lane, fhi = hi0 >> lane, finv = inv0 >> lane, finvm = invm0 >> lane;
	if (lane) {
		flo |= lo1 << (64 - lane);
		fhi |= hi1 << (64 - lane);
		finv |= inv1 << (64 - lane);
		finvm |= invm1 << (64 - lane);
	}
	flo &= wmask; fhi &= wmask; finv &= wmask; finvm &= wmask;
	vp = p < nwords && finv == 0;
	vm = p < nwords && finvm == 0;
	// plus strand word at query position p: first base is the most significant letter
	const uint64_t rlo = __brevll(flo) >> (64 - W), rhi = __brevll(fhi) >> (64 - W);
	const uint64_t wp = spread32(rlo) | (spread32(rhi) << 1);
	// reverse-complement word covering the same bases: letters complemented, order already reversed
	const uint64_t wm = spread32(~flo & wmask) | (spread32(~fhi & wmask) << 1);
#endif
	sp = mod_slots(murmur64(wp & X.shiftMask), X.slotCount, X.slotMagic);
	sm = mod_slots(murmur64(wm & X.shiftMask), X.slotCount, X.slotMagic);
}

}  // namespace urx
