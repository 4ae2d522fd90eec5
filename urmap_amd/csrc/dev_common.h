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
__device__ __forceinline__ float wave_prefix_max(float v) {
	asm volatile(
	    "s_nop 1\n\t"
	    "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
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

// ExtendPen's two x-drop walks (extendpen.cpp:25-78) over a precomputed mismatch bit vector, one candidate per lane.
// `cap` is an UPPER BOUND of m_MaxPenalty at the candidate's turn (the cap only falls, so its value when the batch is
// walked is one): a lane whose penalty passes it fails extendpen.cpp:43-44 / 69-70 at its turn whatever happens before,
// so it stops walking and returns that penalty (> cap); the caller's ordered test drops it.  Lanes under the bound
// return the full walk's result and the caller compares their penalty with the cap in order.
// The walk visits mismatches only.  The loop over the N words is wave-uniform and unrolled; inside a word the set
// bits are consumed one by one (x &= x - 1), so an iteration is ~20 VALU instructions with no word selection.
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
template <int N>
__device__ __forceinline__ int mismatches_outside_seed(const uint64_t (&w)[N], int qpos, int W) {
	int pc = 0;
#pragma unroll
	for (int c = 0; c < N; ++c) {
		const int lo = qpos - 64 * c, hi = lo + W;  // seed bits of this word: [lo, hi) cut to [0, 64)
		const uint64_t below_hi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
		const uint64_t below_lo = lo >= 64 ? ~0ull : (lo <= 0 ? 0ull : ((1ull << lo) - 1ull));
		pc += __builtin_popcountll(w[c] & ~(below_hi & ~below_lo));
	}
	return pc;
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

}  // namespace urx
