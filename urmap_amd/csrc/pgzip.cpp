// pgzip.cpp -- see pgzip.h.  RFC 1951 (deflate) / RFC 1952 (gzip) decoder whose output may refer to an unknown window.
#include "pgzip.h"

#include "../../include/urmapx.h"

#include <omp.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace urx {
namespace {

constexpr int LL_BITS = 11, D_BITS = 9;         // primary table widths
constexpr uint32_t LINK = 0x80000000u;
constexpr size_t SLACK = 8u << 20;              // compressed bytes a round reads behind its last segment (a block may cross the cut)
constexpr size_t WIN = 32768;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// A growable array whose new elements are not touched (std::vector value-initialises them: for the symbol buffers and the round's
// text that was a pass over 0.2-0.4 GB per round on one thread -- as long as the round's decoding on sixteen).
template <class T>
struct RawBuf {
	std::unique_ptr<T[]> p;
	size_t cap = 0;
	T *data() { return p.get(); }
	const T *data() const { return p.get(); }
	size_t size() const { return cap; }  // the capacity
	T &operator[](size_t i) { return p[i]; }
	const T &operator[](size_t i) const { return p[i]; }
	void reserve_keep(size_t need, size_t keep) {  // room for `need` elements, the first `keep` carried over
		if (need <= cap) return;
		const size_t nc = std::max(need, cap + cap / 2);
		std::unique_ptr<T[]> q(new T[nc]);
		if (keep) memcpy(q.get(), p.get(), keep * sizeof(T));
		p = std::move(q);
		cap = nc;
	}
};

void par_copy(char *d, const char *s, size_t n, int threads) {
	constexpr size_t PIECE = 4u << 20;
	if (n < 2 * PIECE || threads < 2) { memcpy(d, s, n); return; }
	const long pieces = (long)((n + PIECE - 1) / PIECE);
#pragma omp parallel for schedule(static) num_threads(threads)
	for (long i = 0; i < pieces; ++i) {
		const size_t lo = (size_t)i * PIECE, hi = std::min(n, lo + PIECE);
		memcpy(d + lo, s + lo, hi - lo);
	}
}

// ---- the two passes over a round's text that are not decoding: symbols -> bytes, CRC-32 ----
// Both were scalar loops (a branch per symbol; zlib's table CRC) and together cost more than the decode itself once 16 threads
// shared it (18 of a round's 40 ms on the GPU box).  With AVX2 / PCLMULQDQ (checked at run time; the plain loops otherwise):
//  * 32 symbols at a time: if all are literals (< 256: everything but the first few hundred KB of a segment, whose copies still
//    reach into the unknown window) they are packed to bytes with one instruction; a group with a window reference takes the old loop;
//  * CRC-32 by carry-less multiplication: 64 bytes folded per step with the constants of the reflected polynomial 0xEDB88320
//    (x^(512+64) mod P, x^512 mod P for the four-register step, x^(128+64), x^128 for one register: Gopal et al., "Fast CRC computation
//    for generic polynomials using PCLMULQDQ", Intel 2009), the last 16 bytes and the tail by the table.  The routine proves itself against
//    zlib's crc32 on its first use (several lengths and alignments); if it ever disagreed it would not be used.
uint32_t crc_table_update(uint32_t raw, const uint8_t *p, size_t n) {  // raw: the register without the pre / post inversion
	static uint32_t T[256];
	static bool init = false;
	if (!init) {
#pragma omp critical(urx_crc_table)
		{
			if (!init) {
				for (uint32_t i = 0; i < 256; ++i) {
					uint32_t c = i;
					for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
					T[i] = c;
				}
				init = true;
			}
		}
	}
	for (size_t i = 0; i < n; ++i) raw = T[(raw ^ p[i]) & 0xFFu] ^ (raw >> 8);
	return raw;
}

#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_fold_step(__m128i x, __m128i k, __m128i next) {
	const __m128i lo = _mm_clmulepi64_si128(x, k, 0x00), hi = _mm_clmulepi64_si128(x, k, 0x11);
	return _mm_xor_si128(_mm_xor_si128(lo, hi), next);
}
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_fold_pclmul(uint32_t raw, const uint8_t *p, size_t n) {  // n >= 64, a multiple of 16
	const __m128i k12 = _mm_set_epi64x(0x00000001c6e41596LL, 0x0000000154442bd4LL);  // high, low
	const __m128i k34 = _mm_set_epi64x(0x00000000ccaa009eLL, 0x00000001751997d0LL);
	__m128i x1 = _mm_loadu_si128((const __m128i *)p), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
	__m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)raw));
	p += 64; n -= 64;
	while (n >= 64) {
		x1 = crc_fold_step(x1, k12, _mm_loadu_si128((const __m128i *)p));
		x2 = crc_fold_step(x2, k12, _mm_loadu_si128((const __m128i *)(p + 16)));
		x3 = crc_fold_step(x3, k12, _mm_loadu_si128((const __m128i *)(p + 32)));
		x4 = crc_fold_step(x4, k12, _mm_loadu_si128((const __m128i *)(p + 48)));
		p += 64; n -= 64;
	}
	x1 = crc_fold_step(x1, k34, x2);
	x1 = crc_fold_step(x1, k34, x3);
	x1 = crc_fold_step(x1, k34, x4);
	while (n >= 16) {
		x1 = crc_fold_step(x1, k34, _mm_loadu_si128((const __m128i *)p));
		p += 16; n -= 16;
	}
	uint8_t last[16];
	_mm_storeu_si128((__m128i *)last, x1);
	return crc_table_update(0u, last, 16);  // what is left of the message is these 16 bytes
}
__attribute__((target("avx2"))) size_t narrow_literals_avx2(const uint16_t *s, size_t n, char *o) {  // leading groups of 32 literals; returns how many symbols it took
	size_t i = 0;
	const __m256i hi = _mm256_set1_epi16((short)0xFF00);
	for (; i + 32 <= n; i += 32) {
		const __m256i a = _mm256_loadu_si256((const __m256i *)(s + i)), b = _mm256_loadu_si256((const __m256i *)(s + i + 16));
		if (!_mm256_testz_si256(_mm256_or_si256(a, b), hi)) break;  // a symbol >= 256 in the group
		const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8);
		_mm256_storeu_si256((__m256i *)(o + i), pk);
	}
	return i;
}
#endif

bool cpu_has(const char *what) {
#if defined(__x86_64__)
	if (!strcmp(what, "avx2")) return __builtin_cpu_supports("avx2");
	if (!strcmp(what, "pclmul")) return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#endif
	(void)what;
	return false;
}

// zlib's crc32(0, p, n)
int g_crc_pclmul = -1;  // -1 not tried, 0 no, 1 yes (verified against zlib)
uint32_t crc32_of(const uint8_t *p, size_t n) {
#if defined(__x86_64__)
	int &usable = g_crc_pclmul;
	if (usable < 0) {
#pragma omp critical(urx_crc_check)
		if (usable < 0) {
			int ok = cpu_has("pclmul") && !getenv("URMAPX_PGZIP_NO_SIMD") ? 1 : 0;
			if (ok) {
				std::vector<uint8_t> t(4096 + 64);
				uint32_t x = 12345u;
				for (uint8_t &b : t) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
				for (size_t off : {(size_t)0, (size_t)1, (size_t)7})
					for (size_t len : {(size_t)64, (size_t)80, (size_t)127, (size_t)128, (size_t)1000, (size_t)4096}) {
						const size_t body = len & ~(size_t)15;
						uint32_t raw = crc_fold_pclmul(0xFFFFFFFFu, t.data() + off, body);
						raw = crc_table_update(raw, t.data() + off + body, len - body);
						if (~raw != (uint32_t)crc32(0L, t.data() + off, (uInt)len)) ok = 0;
					}
			}
			usable = ok;
		}
	}
	if (usable == 1 && n >= 64) {
		const size_t body = n & ~(size_t)15;
		uint32_t raw = crc_fold_pclmul(0xFFFFFFFFu, p, body);
		raw = crc_table_update(raw, p + body, n - body);
		return ~raw;
	}
#endif
	uint32_t c = 0;
	while (n) {  // (zlib's length is 32 bits)
		const size_t k = std::min<size_t>(n, 1u << 30);
		c = (uint32_t)crc32(c, p, (uInt)k);
		p += k; n -= k;
	}
	return c;
}

inline uint32_t rev_bits(uint32_t c, int n) {
	uint32_t r = 0;
	for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
	return r;
}

// Canonical Huffman decoding table (deflate packs codes LSB first: the table is indexed by the reversed code).  Entry: symbol << 8 |
// bits to consume; a LINK entry leads to a subtable: LINK | subtable bits << 24 | offset << 8 | primary bits.
// Returns 0: a usable complete code; 1: no code at all (all lengths zero); 2: a single code of length 1 (incomplete, allowed);
// -1: over-subscribed or incomplete -- the rules of zlib's inflate_table (inftrees.c).
struct HuffTable {
	std::vector<uint32_t> t;
	int tbits = 0;
	int build(const uint8_t *lens, int n, int primary_bits) {
		int count[16] = {0};
		for (int i = 0; i < n; ++i) ++count[lens[i]];
		tbits = primary_bits;
		if (count[0] == n) { t.assign((size_t)1 << tbits, 0u); return 1; }
		int left = 1, maxl = 0;
		for (int l = 1; l <= 15; ++l) {
			left <<= 1;
			left -= count[l];
			if (left < 0) return -1;
			if (count[l]) maxl = l;
		}
		const bool single = left > 0 && maxl == 1 && count[1] == 1;
		if (left > 0 && !single) return -1;
		uint32_t next[16];
		{
			uint32_t c = 0;
			int prev = 0;  // (the count of length 0 does not enter the recurrence)
			for (int l = 1; l <= 15; ++l) { c = (c + (uint32_t)prev) << 1; next[l] = c; prev = count[l]; }
		}
		t.assign((size_t)1 << tbits, 0u);
		// subtables for codes longer than the primary width: one per distinct primary prefix
		if (maxl > tbits) {
			// first pass: how many bits each prefix needs
			std::vector<uint8_t> need((size_t)1 << tbits, 0);
			uint32_t nx[16];
			memcpy(nx, next, sizeof nx);
			for (int s = 0; s < n; ++s) {
				const int l = lens[s];
				if (l <= tbits) { if (l) ++nx[l]; continue; }
				const uint32_t r = rev_bits(nx[l]++, l);
				const uint32_t pre = r & (((uint32_t)1 << tbits) - 1u);
				need[pre] = (uint8_t)std::max<int>(need[pre], l - tbits);
			}
			for (size_t pre = 0; pre < need.size(); ++pre)
				if (need[pre]) {
					const size_t off = t.size();
					t.resize(off + ((size_t)1 << need[pre]), 0u);
					t[pre] = LINK | ((uint32_t)need[pre] << 24) | ((uint32_t)off << 8) | (uint32_t)tbits;
				}
		}
		for (int s = 0; s < n; ++s) {
			const int l = lens[s];
			if (!l) continue;
			const uint32_t r = rev_bits(next[l]++, l);
			const uint32_t e = ((uint32_t)s << 8) | (uint32_t)l;
			if (l <= tbits) {
				for (uint32_t i = r; i < ((uint32_t)1 << tbits); i += (uint32_t)1 << l) t[i] = e;
			} else {
				const uint32_t link = t[r & (((uint32_t)1 << tbits) - 1u)];
				const int sb = (int)((link >> 24) & 15u);
				const uint32_t off = (link >> 8) & 0xFFFFu;
				const uint32_t hi = r >> tbits;
				for (uint32_t i = hi; i < ((uint32_t)1 << sb); i += (uint32_t)1 << (l - tbits)) t[off + i] = e;
			}
		}
		return single ? 2 : 0;
	}
};

struct Bits {
	const uint8_t *p = nullptr;  // the round's compressed bytes; 8 readable bytes behind `end`
	uint64_t pos = 0, end = 0;   // in bits
	inline uint64_t peek() const {
		uint64_t v;
		memcpy(&v, p + (pos >> 3), 8);
		return v >> (pos & 7);  // at least 57 bits
	}
	inline uint32_t get(int n) { const uint32_t v = (uint32_t)(peek() & (((uint64_t)1 << n) - 1u)); pos += (uint64_t)n; return v; }
	inline bool over() const { return pos > end; }
};

// dynamic block header at B.pos (behind the 3 header bits) -> the two tables; false: not a valid header
bool read_dynamic(Bits &B, HuffTable &LL, HuffTable &DD) {
	if (B.pos + 14 > B.end) return false;
	const int hlit = (int)B.get(5) + 257, hdist = (int)B.get(5) + 1, hclen = (int)B.get(4) + 4;
	if (hlit > 286 || hdist > 30) return false;
	uint8_t cl[19] = {0};
	if (B.pos + 3 * (uint64_t)hclen > B.end) return false;
	for (int i = 0; i < hclen; ++i) cl[CL_ORDER[i]] = (uint8_t)B.get(3);
	HuffTable CL;
	if (CL.build(cl, 19, 7) != 0) return false;  // zlib: the code-length code must be complete
	uint8_t lens[320];
	int n = 0;
	while (n < hlit + hdist) {
		if (B.pos + 16 > B.end) return false;
		const uint32_t e = CL.t[B.peek() & 127u];
		const int l = (int)(e & 255u), sym = (int)(e >> 8);
		if (!l) return false;
		B.pos += (uint64_t)l;
		if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
		int rep, val = 0;
		if (sym == 16) { if (n == 0) return false; val = lens[n - 1]; rep = 3 + (int)B.get(2); }
		else if (sym == 17) rep = 3 + (int)B.get(3);
		else rep = 11 + (int)B.get(7);
		if (n + rep > hlit + hdist) return false;
		while (rep--) lens[n++] = (uint8_t)val;
	}
	if (lens[256] == 0) return false;  // "missing end-of-block"
	const int r1 = LL.build(lens, hlit, LL_BITS);
	if (r1 < 0 || r1 == 1) return false;
	const int r2 = DD.build(lens + hlit, hdist, D_BITS);
	if (r2 < 0) return false;
	return true;
}

void fixed_tables(HuffTable &LL, HuffTable &DD) {
	uint8_t l[288];
	for (int i = 0; i < 144; ++i) l[i] = 8;
	for (int i = 144; i < 256; ++i) l[i] = 9;
	for (int i = 256; i < 280; ++i) l[i] = 7;
	for (int i = 280; i < 288; ++i) l[i] = 8;
	LL.build(l, 288, LL_BITS);
	uint8_t d[32];  // 32 five-bit codes; 30 and 31 never occur in a valid stream (the decoder rejects them)
	for (int i = 0; i < 32; ++i) d[i] = 5;
	DD.build(d, 32, D_BITS);
}

inline bool text_byte(uint32_t c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// One segment's decoder.  Output symbols: < 256 a literal byte; 0x8000 | k = byte k of the 32 KB in front of this decoder's start
// (k = 32767 is the byte just before it).
struct SegDecoder {
	Bits B;
	RawBuf<uint16_t> buf;  // out[0, n): size() is the capacity, the symbols are written through a raw pointer
	size_t n = 0;
	size_t hint = 0;       // symbols a segment of this run is expected to make (the first allocation: no doubling from nothing)
	void grow(size_t need) { if (need > buf.size()) buf.reserve_keep(std::max({need + (1u << 20), buf.size() * 2, hint}), n); }
	void reset() { n = 0; members.clear(); member_start = 0; from_unknown = true; failed = false; at_eof = false; in_stream = true; block_start = 0; }
	struct Member { size_t out_end; uint32_t crc, isize; };  // a member ended behind out[0, out_end)
	std::vector<Member> members;
	size_t member_start = 0;     // out index where the current member began (0 and from_unknown: it began before this segment)
	bool from_unknown = true;    // no member start seen yet: references may reach into the unknown window
	bool failed = false, at_eof = false, in_stream = true;
	uint64_t block_start = 0;    // bit position of the block about to be decoded
	HuffTable LL, DD;

	// gzip member header at byte-aligned B.pos; false: not one (end of input for the caller)
	bool member_header() {
		const uint64_t by = B.pos >> 3, endb = B.end >> 3;
		if (by + 10 > endb) return false;
		const uint8_t *h = B.p + by;
		if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return false;
		uint64_t q = by + 10;
		if (h[3] & 4) { if (q + 2 > endb) return false; q += 2 + (uint64_t)(B.p[q] | (B.p[q + 1] << 8)); }
		if (h[3] & 8) { while (q < endb && B.p[q]) ++q; ++q; }
		if (h[3] & 16) { while (q < endb && B.p[q]) ++q; ++q; }
		if (h[3] & 2) q += 2;
		if (q >= endb) return false;
		B.pos = q << 3;
		return true;
	}

	// Decodes blocks until the start of a block is at or behind `target` (bits), the input ends, or something fails.  Returns with
	// block_start = where it stands.  check_only: nothing is stored, literals must be text (the search for a block start).
	template <bool CHECK>
	bool run(uint64_t target, int max_blocks) {
		for (int nb = 0; nb < max_blocks; ++nb) {
			block_start = B.pos;
			if (B.pos >= target) return true;
			if (B.pos + 3 > B.end) { failed = true; return false; }
			const uint32_t bfinal = B.get(1), btype = B.get(2);
			if (btype == 3) { failed = true; return false; }
			if (btype == 0) {
				B.pos = (B.pos + 7) & ~(uint64_t)7;
				if (B.pos + 32 > B.end) { failed = true; return false; }
				const uint32_t len = B.get(16), nlen = B.get(16);
				if ((len ^ nlen) != 0xFFFFu || B.pos + 8ull * len > B.end) { failed = true; return false; }
				const uint8_t *s = B.p + (B.pos >> 3);
				if (CHECK) {
					for (uint32_t i = 0; i < len; ++i)
						if (!text_byte(s[i])) { failed = true; return false; }
				} else {
					grow(n + len);
					for (uint32_t i = 0; i < len; ++i) buf[n + i] = s[i];
					n += len;
				}
				B.pos += 8ull * len;
			} else {
				if (btype == 1) fixed_tables(LL, DD);
				else if (!read_dynamic(B, LL, DD)) { failed = true; return false; }
				if (!huff_block<CHECK>()) { failed = true; return false; }
			}
			if (bfinal) {
				B.pos = (B.pos + 7) & ~(uint64_t)7;
				if (B.pos + 64 > B.end) { failed = true; return false; }
				const uint8_t *tl = B.p + (B.pos >> 3);
				Member m;
				m.out_end = n;
				m.crc = (uint32_t)tl[0] | ((uint32_t)tl[1] << 8) | ((uint32_t)tl[2] << 16) | ((uint32_t)tl[3] << 24);
				m.isize = (uint32_t)tl[4] | ((uint32_t)tl[5] << 8) | ((uint32_t)tl[6] << 16) | ((uint32_t)tl[7] << 24);
				members.push_back(m);
				B.pos += 64;
				from_unknown = false;
				member_start = n;
				if (CHECK) { block_start = B.pos; return true; }
				if (!member_header()) { at_eof = true; in_stream = false; block_start = B.pos; return true; }  // end of input (or padding / garbage, as gzread treats it)
			}
		}
		block_start = B.pos;
		return true;
	}

	// One Huffman-coded block.  The bit buffer is libdeflate's: `bl` valid bits in `bb`, refilled without a branch by OR-ing the next
	// eight bytes in above them (the bits of a byte that does not fit whole are OR-ed again, identically, by the next refill); after a
	// refill 56..63 bits are valid -- more than a literal / length code, its extra bits, a distance code and its extra bits take (48).
	template <bool CHECK>
	bool huff_block() {
		const uint32_t *lt = LL.t.data(), *dt = DD.t.data();
		const uint64_t lmask = ((uint64_t)1 << LL_BITS) - 1u, dmask = ((uint64_t)1 << D_BITS) - 1u;
		size_t n = this->n;
		size_t cap = buf.size();
		uint16_t *o = buf.data();
		const uint8_t *in = B.p + (B.pos >> 3);
		const uint8_t *const in_limit = B.p + (B.end >> 3);  // the buffer has 16 readable bytes behind it
		uint64_t bb;
		memcpy(&bb, in, 8);
		bb >>= (B.pos & 7);
		uint32_t bl = 64u - (uint32_t)(B.pos & 7);
		in += 8;
		bl -= 8; in -= 1;  // keep one byte back so that `bl` stays below 64 (56 .. 63 after a refill)
		bb &= (bl < 64 ? (((uint64_t)1 << bl) - 1u) : ~0ull);
#define URX_REFILL()                                  \
	do {                                              \
		uint64_t nx_;                                 \
		memcpy(&nx_, in, 8);                          \
		bb |= nx_ << bl;                              \
		in += (63u - bl) >> 3;                        \
		bl |= 56u;                                    \
	} while (0)
		bool ok = true;
		for (;;) {
			if (in > in_limit) { ok = false; break; }
			URX_REFILL();
			uint32_t e = lt[bb & lmask];
			if (e & LINK) {
				const int pb = (int)(e & 255u), sb = (int)((e >> 24) & 15u);
				e = lt[((e >> 8) & 0xFFFFu) + ((bb >> pb) & (((uint64_t)1 << sb) - 1u))];
			}
			uint32_t l = e & 255u;
			if (!l) { ok = false; break; }
			uint32_t sym = e >> 8;
			bb >>= l; bl -= l;
			if (sym < 256) {
				if (CHECK) { if (!text_byte(sym)) { ok = false; break; } }
				else {
					if (n + 4 > cap) { this->n = n; grow(n + 4); cap = buf.size(); o = buf.data(); }
					o[n++] = (uint16_t)sym;
				}
				// up to two more literals out of the bits already in the buffer (at least 41 are left: 15 each)
				e = lt[bb & lmask];
				if (!(e & LINK) && (e >> 8) < 256 && (e & 255u)) {
					sym = e >> 8;
					if (CHECK) { if (!text_byte(sym)) { ok = false; break; } }
					else o[n++] = (uint16_t)sym;
					l = e & 255u; bb >>= l; bl -= l;
					e = lt[bb & lmask];
					if (!(e & LINK) && (e >> 8) < 256 && (e & 255u)) {
						sym = e >> 8;
						if (CHECK) { if (!text_byte(sym)) { ok = false; break; } }
						else o[n++] = (uint16_t)sym;
						l = e & 255u; bb >>= l; bl -= l;
					}
				}
				continue;
			}
			if (sym == 256) break;
			if (sym > 285) { ok = false; break; }
			const int li = (int)sym - 257;
			const uint32_t len = LEN_BASE[li] + (uint32_t)(bb & (((uint64_t)1 << LEN_EXTRA[li]) - 1u));
			bb >>= LEN_EXTRA[li]; bl -= LEN_EXTRA[li];
			uint32_t de = dt[bb & dmask];
			if (de & LINK) {
				const int pb = (int)(de & 255u), sb = (int)((de >> 24) & 15u);
				de = dt[((de >> 8) & 0xFFFFu) + ((bb >> pb) & (((uint64_t)1 << sb) - 1u))];
			}
			const uint32_t dl = de & 255u;
			if (!dl) { ok = false; break; }
			const uint32_t ds = de >> 8;
			if (ds > 29) { ok = false; break; }
			bb >>= dl; bl -= dl;
			const uint32_t dist = DIST_BASE[ds] + (uint32_t)(bb & (((uint64_t)1 << DIST_EXTRA[ds]) - 1u));
			bb >>= DIST_EXTRA[ds]; bl -= DIST_EXTRA[ds];
			if (CHECK) continue;
			if (n + len + 4 > cap) { this->n = n; grow(n + len + 4); cap = buf.size(); o = buf.data(); }
			const size_t have = n - member_start;  // symbols of this member decoded here
			if (dist <= have) {
				const uint16_t *s = o + n - dist;
				if (dist >= len) memcpy(o + n, s, (size_t)len * 2);
				else
					for (uint32_t i = 0; i < len; ++i) o[n + i] = s[i];  // an overlapping copy replicates, as deflate means it to
				n += len;
			} else {
				if (!from_unknown) { ok = false; break; }  // a reference in front of the member's start
				// part (or all) of the match lies in the unknown window in front of this segment: symbols 0x8000 | k
				for (uint32_t i = 0; i < len; ++i) {
					if (dist <= n + i) o[n + i] = o[n + i - dist];
					else o[n + i] = (uint16_t)(0x8000u | (uint16_t)(WIN - (dist - (n + i))));
				}
				n += len;
			}
		}
#undef URX_REFILL
		if (!CHECK) this->n = n;
		B.pos = (uint64_t)(in - B.p) * 8 - bl;
		return ok && B.pos <= B.end;
	}
};

}  // namespace

struct ParallelGunzip::Impl {
	int fd = -1;
	uint64_t csize = 0;
	enum Mode { PAR, RAW, GZ } mode = PAR;  // the parallel road; zlib raw inflate to the end of the current member; zlib with its gzip wrapper
	uint64_t cbits = 0;           // PAR: bit position in the file of the next block (exact)
	bool eof = false, started = false;
	std::vector<uint8_t> window;  // the last <= 32 KB of the current member's text
	uint32_t crc = 0;             // of the current member so far
	uint64_t isize = 0;
	RawBuf<char> obuf;            // decoded text not yet handed out: [obeg, osize)
	size_t obeg = 0, osize = 0;
	// zlib roads
	z_stream zs;
	bool zs_init = false, member_done = true, seen_member = false;
	std::vector<uint8_t> zin;
	size_t zbeg = 0, zhave = 0;
	uint64_t zpos = 0;            // file offset of the next byte to fetch into zin
	size_t seg_bytes = 2u << 20;
	double out_per_in = 0.0;      // text bytes per compressed byte in the last round (0: no round yet)
	std::vector<SegDecoder> segs;  // kept from round to round: their symbol buffers are touched once (fresh pages cost more than the decoding)
	std::vector<uint8_t> cbuf;

	bool pread_all(uint8_t *dst, size_t n, uint64_t off) const {
		size_t got = 0;
		while (got < n) {
			const ssize_t k = ::pread(fd, dst + got, n - got, (off_t)(off + got));
			if (k <= 0) return false;
			got += (size_t)k;
		}
		return true;
	}
	bool pread_par(uint8_t *dst, size_t n, uint64_t off, int threads) const {  // a round's 42 MB: one thread's pread was 8 of its 45 ms
		constexpr size_t PIECE = 2u << 20;
		if (n < 4 * PIECE || threads < 2) return pread_all(dst, n, off);
		const long pieces = (long)((n + PIECE - 1) / PIECE);
		bool ok = true;
#pragma omp parallel for schedule(static) num_threads(threads) reduction(&& : ok)
		for (long i = 0; i < pieces; ++i) {
			const size_t lo = (size_t)i * PIECE, hi = std::min(n, lo + PIECE);
			ok = ok && pread_all(dst + lo, hi - lo, off + lo);
		}
		return ok;
	}
	bool zfill() {  // more compressed bytes; false at the end of the file (or a read error)
		if (zbeg < zhave) return true;
		const size_t n = (size_t)std::min<uint64_t>(zin.size(), csize - zpos);
		if (n == 0 || !pread_all(zin.data(), n, zpos)) return false;
		zpos += n; zbeg = 0; zhave = n;
		return true;
	}
	bool ztake(uint8_t *dst, size_t n) {  // n bytes of the compressed stream
		for (size_t i = 0; i < n; ++i) {
			if (!zfill()) return false;
			dst[i] = zin[zbeg++];
		}
		return true;
	}
	bool zpeek2(uint8_t two[2]) {  // the next two compressed bytes without consuming them
		if (zhave - zbeg >= 2) { two[0] = zin[zbeg]; two[1] = zin[zbeg + 1]; return true; }
		const uint64_t at = zpos - (zhave - zbeg);
		return csize - at >= 2 && pread_all(two, 2, at);
	}
	bool zstart(int wbits) {
		if (zs_init) inflateEnd(&zs);
		memset(&zs, 0, sizeof zs);
		zs_init = inflateInit2(&zs, wbits) == Z_OK;
		if (zin.empty()) zin.resize(4u << 20);
		return zs_init;
	}
};

uint32_t crc32_fast(const uint8_t *p, size_t n) { return crc32_of(p, n); }

ParallelGunzip::ParallelGunzip() : d_(new Impl) {}
ParallelGunzip::~ParallelGunzip() {
	if (d_->zs_init) inflateEnd(&d_->zs);
}

bool ParallelGunzip::open(int fd, uint64_t csize) {
	d_->fd = fd; d_->csize = csize;
	if (const char *e = getenv("URMAPX_PGZIP_SEGMENT")) { const long v = atol(e); if (v >= 4096) d_->seg_bytes = (size_t)v; }  // test aid: small segments
	uint8_t h[2];
	return csize >= 18 && d_->pread_all(h, 2, 0) && h[0] == 0x1f && h[1] == 0x8b;
}

size_t ParallelGunzip::read(char *dst, size_t cap, int threads) {
	Impl &D = *d_;
	size_t done = 0;
	if (threads < 1) threads = 1;
	while (done < cap && !bad_) {
		if (D.obeg < D.osize) {  // text already decoded
			const size_t k = std::min(cap - done, D.osize - D.obeg);
			par_copy(dst + done, D.obuf.data() + D.obeg, k, threads);
			D.obeg += k; done += k;
			if (D.obeg == D.osize) { D.osize = 0; D.obeg = 0; }
			continue;
		}
		if (D.eof) break;
		if (!D.started) {
			// the first member's header, through the decoder's own parser
			D.started = true;
			std::vector<uint8_t> head((size_t)std::min<uint64_t>(D.csize, 1u << 16) + 16, 0);
			if (!D.pread_all(head.data(), head.size() - 16, 0)) { bad_ = true; break; }
			SegDecoder S;
			S.B.p = head.data(); S.B.pos = 0; S.B.end = (uint64_t)(head.size() - 16) * 8;
			const bool ok = S.member_header();  // (a header longer than 64 KB: zlib reads the file)
			if (!ok || threads < 2 || D.csize < 2 * D.seg_bytes || getenv("URMAPX_PGZIP_OFF")) {
				D.mode = Impl::GZ; D.zpos = 0;
				if (!D.zstart(15 + 32)) { bad_ = true; break; }
			} else D.cbits = S.B.pos;
		}
		if (D.mode != Impl::PAR) {
			// zlib on the calling thread, straight into dst.  GZ: the gzip wrapper (its own header parser, CRC and length checks), member after
			// member.  RAW: the rest of a member the parallel road gave up on, its trailer checked here; what follows goes the GZ way.
			const size_t want = cap - done;
			size_t out = 0;
			while (out < want && !bad_ && !D.eof) {
				if (D.mode == Impl::GZ && D.member_done && D.seen_member) {
					// behind a complete member: another member continues the text; anything else (zero padding, trailing garbage) ends the
					// input, as zlib's gzread treats it (gz_look)
					uint8_t two[2];
					if (!D.zpeek2(two) || two[0] != 0x1f || two[1] != 0x8b) { D.eof = true; break; }
				}
				if (!D.zfill()) {
					if (D.mode == Impl::RAW || !D.member_done) bad_ = true;  // the file ends inside a member
					D.eof = true;
					break;
				}
				D.zs.next_in = D.zin.data() + D.zbeg;
				D.zs.avail_in = (uInt)(D.zhave - D.zbeg);
				D.zs.next_out = (Bytef *)dst + done + out;
				D.zs.avail_out = (uInt)std::min<size_t>(want - out, 1u << 30);
				const uInt o0 = D.zs.avail_out;
				const int rc = inflate(&D.zs, Z_NO_FLUSH);
				D.zbeg = D.zhave - D.zs.avail_in;
				const size_t got = o0 - D.zs.avail_out;
				if (D.mode == Impl::RAW) { D.crc = (uint32_t)crc32(D.crc, (const Bytef *)dst + done + out, (uInt)got); D.isize += got; }
				out += got;
				D.member_done = false;
				if (rc == Z_STREAM_END) {
					D.member_done = true; D.seen_member = true;
					if (D.mode == Impl::RAW) {
						uint8_t tl[8];
						if (!D.ztake(tl, 8)) { bad_ = true; break; }
						const uint32_t wcrc = (uint32_t)tl[0] | ((uint32_t)tl[1] << 8) | ((uint32_t)tl[2] << 16) | ((uint32_t)tl[3] << 24);
						const uint32_t wlen = (uint32_t)tl[4] | ((uint32_t)tl[5] << 8) | ((uint32_t)tl[6] << 16) | ((uint32_t)tl[7] << 24);
						if (wcrc != D.crc || wlen != (uint32_t)D.isize) { bad_ = true; break; }
						D.mode = Impl::GZ;
						if (!D.zstart(15 + 32)) { bad_ = true; break; }
					} else if (inflateReset(&D.zs) != Z_OK) { bad_ = true; break; }
				} else if (rc != Z_OK && rc != Z_BUF_ERROR) { bad_ = true; break; }
			}
			ser_bytes_ += out;
			done += out;
			continue;
		}

		// ---- one round of the parallel road ----
		const double tr0 = omp_get_wtime();
		const uint64_t B0 = D.cbits >> 3;
		// Round 6: the round is fitted to what is left of the caller's buffer, so that its text is written THERE (`direct` below) instead of into obuf and copied
		// out -- the copy was a third of the reader's time per chunk of urmapx_map_files (178 MB of text per round of 16 segments against chunks of 86-173 MB: every
		// round missed).  Same number of segments (one per thread), SHORTER segments; if even segments an eighth of the usual size do not fit and this call has
		// already produced text, a short read -- the caller comes back with room for a round (urmapx_map_files takes a chunk that is three quarters full); a caller
		// whose whole buffer is smaller than that goes through obuf as before.
		size_t seg = D.seg_bytes;
		int T = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(1, (D.csize - B0 + seg - 1) / seg));
		if (D.out_per_in > 0.0 && (double)T * (double)seg * D.out_per_in * 1.04 > (double)(cap - done)) {
			const size_t seg_fit = (size_t)((double)(cap - done) / ((double)T * D.out_per_in * 1.04)) & ~(size_t)4095;
			if (seg_fit >= D.seg_bytes / 8 && seg_fit >= 4096) seg = seg_fit;
			else if (done > 0) break;
		}
		const uint64_t B1 = std::min<uint64_t>(D.csize, B0 + (uint64_t)T * seg);
		const uint64_t Bread = std::min<uint64_t>(D.csize, B1 + SLACK);
		std::vector<uint8_t> &cbuf = D.cbuf;
		if (cbuf.size() < (size_t)(Bread - B0) + 16) cbuf.resize((size_t)(Bread - B0) + 16);
		const double tread0 = omp_get_wtime();
		if (!D.pread_par(cbuf.data(), (size_t)(Bread - B0), B0, threads)) { bad_ = true; break; }
		const double tread = omp_get_wtime() - tread0;
		memset(cbuf.data() + (Bread - B0), 0, 16);
		const uint64_t end_bits = (Bread - B0) * 8, cut_bits = (B1 - B0) * 8;
		const bool last_round = B1 == D.csize;
		std::vector<SegDecoder> &segs = D.segs;
		if ((int)segs.size() < T) segs.resize((size_t)T);
		for (SegDecoder &S : segs) { S.reset(); S.hint = (size_t)seg * 6; }  // FASTQ text deflates to a fifth or less; more grows the buffer
		std::vector<uint64_t> start((size_t)T, ~0ull);
		start[0] = D.cbits - B0 * 8;
		// 1. block starts behind the cuts
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
		for (int i = 1; i < T; ++i) {
			const uint64_t lo = (uint64_t)i * seg * 8, hi = std::min<uint64_t>(cut_bits, (uint64_t)(i + 1) * seg * 8);
			SegDecoder S;
			S.B.p = cbuf.data(); S.B.end = end_bits;
			for (uint64_t b = lo; b < hi; ++b) {
				// cheap rejection first: BFINAL = 0 (a stream's last block is not looked for), BTYPE = 10, HLIT <= 29, HDIST <= 29
				uint64_t v;
				memcpy(&v, cbuf.data() + (b >> 3), 8);
				v >>= (b & 7);
				if ((v & 7u) != 4u) continue;
				if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;
				S.B.pos = b; S.failed = false; S.members.clear();
				if (!S.run<true>(~0ull, 1) || S.failed || !S.members.empty()) continue;
				// the block behind it must be one too
				const uint64_t nb = S.B.pos;
				if (nb + 3 > end_bits) continue;
				memcpy(&v, cbuf.data() + (nb >> 3), 8);
				v >>= (nb & 7);
				const uint32_t bt = (uint32_t)((v >> 1) & 3u);
				if (bt == 3) continue;
				if (bt == 2) {
					Bits B2;
					B2.p = cbuf.data(); B2.end = end_bits; B2.pos = nb + 3;
					HuffTable a, c;
					if (!read_dynamic(B2, a, c)) continue;
				}
				start[(size_t)i] = b;
				break;
			}
		}
		// 2. every segment from its start to the next one's
		const double tr1 = omp_get_wtime();
		std::vector<int> act;
		for (int i = 0; i < T; ++i)
			if (start[(size_t)i] != ~0ull) act.push_back(i);
		const int A = (int)act.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
		for (int a = 0; a < A; ++a) {
			SegDecoder &S = segs[(size_t)act[(size_t)a]];
			S.B.p = cbuf.data(); S.B.end = end_bits; S.B.pos = start[(size_t)act[(size_t)a]];
			S.grow(seg * 6);
			const uint64_t target = a + 1 < A ? start[(size_t)act[(size_t)a + 1]] : (last_round ? ~0ull : cut_bits);
			S.run<false>(target, 1 << 30);
		}
		// A segment that walked past its successor's start without landing on it proves that start false: the successor's work is
		// dropped and the segment goes on to the next start (on this thread: the rare road).  By induction from the round's exact
		// start every boundary used is a real block boundary.
		const double tr2 = omp_get_wtime();
		bool round_ok = true;
		std::vector<int> keep;
		for (int a = 0; a < A;) {
			SegDecoder &S = segs[(size_t)act[(size_t)a]];
			keep.push_back(act[(size_t)a]);
			int nx = a + 1;
			while (!S.failed && !S.at_eof && nx < A && S.block_start != start[(size_t)act[(size_t)nx]]) {
				if (S.block_start > start[(size_t)act[(size_t)nx]]) { ++nx; continue; }
				S.run<false>(start[(size_t)act[(size_t)nx]], 1 << 30);
			}
			if (!S.failed && !S.at_eof && nx == A && (last_round || S.block_start < cut_bits)) S.run<false>(last_round ? ~0ull : cut_bits, 1 << 30);
			if (S.failed) { round_ok = false; break; }
			if (S.at_eof) break;
			a = nx;
		}
		if (!round_ok) {
			// Something in this round does not decode: a corrupt file, or input this decoder does not take.  zlib gets the rest of the member
			// from the round's exact start: raw inflate primed with the window and the bit offset.  (What it makes of a corrupt file is then
			// zlib's verdict.)
			D.mode = Impl::RAW;
			if (!D.zstart(-15)) { bad_ = true; break; }
			if (!D.window.empty() && inflateSetDictionary(&D.zs, D.window.data(), (uInt)D.window.size()) != Z_OK) { bad_ = true; break; }
			D.zpos = D.cbits >> 3; D.zbeg = D.zhave = 0;
			const int pb = (int)(D.cbits & 7);
			if (pb) {
				uint8_t first;
				if (!D.ztake(&first, 1) || inflatePrime(&D.zs, 8 - pb, first >> pb) != Z_OK) { bad_ = true; break; }
			}
			D.member_done = false; D.seen_member = true;
			continue;
		}
		// 3. windows front to back, then every kept segment's symbols to bytes
		const double tr3 = omp_get_wtime();
		const int K = (int)keep.size();
		std::vector<size_t> ooff((size_t)K + 1, 0);
		for (int k = 0; k < K; ++k) ooff[(size_t)k + 1] = ooff[(size_t)k] + segs[(size_t)keep[(size_t)k]].n;
		std::vector<std::vector<uint8_t>> win((size_t)K + 1);
		win[0] = D.window;
		bool sym_ok = true;
		for (int k = 0; k < K; ++k) {
			const SegDecoder &S = segs[(size_t)keep[(size_t)k]];
			const std::vector<uint8_t> &wp = win[(size_t)k];
			std::vector<uint8_t> &w = win[(size_t)k + 1];
			// the window behind this segment: the last 32 KB of the current member's text -- nothing in front of a member start counts
			const bool whole = !S.from_unknown;
			const size_t from_seg = std::min(S.n - (whole ? S.member_start : 0), WIN);
			const size_t from_prev = whole ? 0 : std::min(WIN - from_seg, wp.size());
			w.resize(from_prev + from_seg);
			if (from_prev) memcpy(w.data(), wp.data() + (wp.size() - from_prev), from_prev);
			const size_t wbase = WIN - wp.size();
			for (size_t i = 0; i < from_seg; ++i) {
				const uint16_t v = S.buf[S.n - from_seg + i];
				if (v < 256) w[from_prev + i] = (uint8_t)v;
				else if ((size_t)(v & 0x7FFFu) < wbase) { sym_ok = false; w[from_prev + i] = 0; }
				else w[from_prev + i] = wp[(size_t)(v & 0x7FFFu) - wbase];
			}
		}
		if (!sym_ok) { bad_ = true; break; }  // a reference in front of the member's first byte: a corrupt stream
		const size_t total = ooff[(size_t)K];
		// straight into the caller's buffer when the round's text fits what is left of it (obuf is empty here); else through obuf
		const bool direct = total <= cap - done;
		if (!direct) { D.obuf.reserve_keep(total, 0); D.osize = total; D.obeg = 0; }
		char *const otext = direct ? dst + done : D.obuf.data();
		// pieces of at most 4 MB (all threads take part whatever the number of segments), cut at segment and member ends; a piece's
		// symbols become bytes and its CRC-32 is taken while they are still in the cache
		std::vector<std::pair<size_t, std::pair<uint32_t, uint32_t>>> ends;  // members that ended in this round: (offset in its text, (crc, isize))
		for (int k = 0; k < K; ++k)
			for (const SegDecoder::Member &m : segs[(size_t)keep[(size_t)k]].members) ends.push_back({ooff[(size_t)k] + m.out_end, {m.crc, m.isize}});
		struct Piece { int k; size_t lo, hi; };  // symbols [lo, hi) of kept segment k
		std::vector<Piece> pieces;
		{
			size_t ei = 0;
			for (int k = 0; k < K; ++k) {
				const size_t n = segs[(size_t)keep[(size_t)k]].n;
				size_t lo = 0;
				while (lo < n) {
					size_t hi = std::min(n, lo + (4u << 20));
					while (ei < ends.size() && ends[ei].first <= ooff[(size_t)k] + lo) ++ei;
					if (ei < ends.size() && ends[ei].first < ooff[(size_t)k] + hi) hi = ends[ei].first - ooff[(size_t)k];
					pieces.push_back(Piece{k, lo, hi});
					lo = hi;
				}
			}
		}
		std::vector<uint8_t> pok(pieces.size(), 1);
		std::vector<uint32_t> pcrc(pieces.size(), 0);
		const bool simd_narrow = cpu_has("avx2") && !getenv("URMAPX_PGZIP_NO_SIMD");
		(void)simd_narrow;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
		for (long pi = 0; pi < (long)pieces.size(); ++pi) {
			const Piece &pc = pieces[(size_t)pi];
			const SegDecoder &S = segs[(size_t)keep[(size_t)pc.k]];
			const std::vector<uint8_t> &w = win[(size_t)pc.k];
			const uint16_t *s = S.buf.data();
			char *o = otext + ooff[(size_t)pc.k];
			const size_t wbase = WIN - w.size();
			bool ok = true;
			for (size_t i = pc.lo; i < pc.hi;) {
#if defined(__x86_64__)
				if (simd_narrow) {
					i += narrow_literals_avx2(s + i, pc.hi - i, o + i);
					if (i >= pc.hi) break;
				}
#endif
				const size_t stop = std::min(pc.hi, i + 32);  // the group the packed pass stopped at (or, without it, the next 32 symbols)
				for (; i < stop; ++i) {
					const uint16_t v = s[i];
					if (v < 256) o[i] = (char)v;
					else {
						const size_t kk = v & 0x7FFFu;
						if (kk < wbase) { ok = false; o[i] = 0; }
						else o[i] = (char)w[kk - wbase];
					}
				}
			}
			if (!ok) pok[(size_t)pi] = 0;
			pcrc[(size_t)pi] = crc32_of((const uint8_t *)o + pc.lo, pc.hi - pc.lo);
		}
		for (uint8_t x : pok)
			if (!x) bad_ = true;
		if (bad_) break;
		// CRC-32 and length of every member that ended in this round, as zlib checks them: the pieces' values combined in order
		const double tr4 = omp_get_wtime();
		{
			size_t ei = 0;
			auto member_ends_at = [&](size_t at) {
				while (ei < ends.size() && ends[ei].first == at) {
					if (D.crc != ends[ei].second.first || (uint32_t)D.isize != ends[ei].second.second) bad_ = true;
					D.crc = 0; D.isize = 0;
					++ei;
				}
			};
			member_ends_at(0);
			for (size_t pi = 0; pi < pieces.size(); ++pi) {
				const size_t len = pieces[pi].hi - pieces[pi].lo;
				D.crc = (uint32_t)crc32_combine(D.crc, pcrc[pi], (z_off_t)len);
				D.isize += len;
				member_ends_at(ooff[(size_t)pieces[pi].k] + pieces[pi].hi);
			}
			if (bad_) break;
		}
		par_bytes_ += total;
		if (direct) done += total;
		if (getenv("URMAPX_PGZIP_VERBOSE"))
			fprintf(stderr, "pgzip round: %d segments (%d kept), %.1f MB in, %.1f MB out; read %.3f, find %.3f, decode %.3f, mend %.3f, resolve %.3f, crc %.3f s\n", T, K,
			        (B1 - B0) / 1e6, total / 1e6, tread, tr1 - tr0, tr2 - tr1, tr3 - tr2, tr4 - tr3, omp_get_wtime() - tr4);
		D.window = win[(size_t)K];
		const SegDecoder &L = segs[(size_t)keep[(size_t)K - 1]];
		if (L.at_eof) D.eof = true;
		else {
			const uint64_t nbits = B0 * 8 + L.block_start;
			if (nbits > D.cbits + 8) D.out_per_in = (double)total / ((double)(nbits - D.cbits) / 8.0);  // text bytes per compressed byte, for the next round's fit
			D.cbits = nbits;
		}
	}
	return done;
}

}  // namespace urx

// include/urmapx.h: a .gz file inflated by the reader of urmapx_map_files, to a file (the same bytes `gzip -dc` writes)
extern "C" int urmapx_gunzip_file(const char *gz_path, const char *out_path, int threads, uint64_t stats[3]) {
	if (!gz_path || !out_path) return URMAPX_E_ARG;
	if (stats) stats[0] = stats[1] = stats[2] = 0;
	FILE *in = fopen(gz_path, "rb");
	if (!in) return URMAPX_E_IO;
	fseeko(in, 0, SEEK_END);
	const uint64_t csize = (uint64_t)ftello(in);
	urx::ParallelGunzip g;
	if (!g.open(fileno(in), csize)) { fclose(in); return URMAPX_E_FORMAT; }
	FILE *out = fopen(out_path, "wb");
	if (!out) { fclose(in); return URMAPX_E_IO; }
	std::vector<char> buf(64u << 20);
	uint64_t total = 0;
	int rc = URMAPX_OK;
	for (;;) {
		const size_t k = g.read(buf.data(), buf.size(), threads > 0 ? threads : omp_get_max_threads());
		if (k == 0) break;
		if (fwrite(buf.data(), 1, k, out) != k) { rc = URMAPX_E_IO; break; }
		total += k;
	}
	if (g.failed() && rc == URMAPX_OK) rc = URMAPX_E_FORMAT;
	fclose(out);
	fclose(in);
	if (stats) { stats[0] = total; stats[1] = g.parallel_bytes(); stats[2] = g.serial_bytes(); }
	return rc;
}

extern "C" int urmapx_pgzip_simd(void) {
	(void)urx::crc32_of((const uint8_t *)"", 0);  // (runs the self-check)
	return (urx::cpu_has("avx2") && !getenv("URMAPX_PGZIP_NO_SIMD") ? 1 : 0) | (urx::g_crc_pclmul == 1 ? 2 : 0);
}
