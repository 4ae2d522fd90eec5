// text_gpu.hip -- FASTQ bytes in, SAM bytes out, both ends of the mapping path on the device.
//
// The reference parses FASTQ one line at a time under a lock (FASTQSeqSource::GetNextLo, fastqseqsource.cpp:9-116, over
// LineReader, linereader.cpp:54-101) and formats each record in the mapping thread (State1::SetSAM / SetSAM_Unmapped,
// setsam.cpp:12-207; GetCIGAR / PathToCIGAR / CIGAROpsFixDanglingMs, state1.cpp:707-734, cigar.cpp:4-41,141-199).  With
// the search on the GPU those two text stages were what bounded `urmap -map` file to file, so a
// chunk of the FASTQ file now crosses PCIe as it lies in the file and comes back as the bytes of its SAM records:
//
//   raw bytes --nl_count / nl_emit--> line ends --record_kernel--> read lengths --scan--> offs
//             --copy_bases_kernel--> bases   --urmapx_map_se_device--> results, paths
//             --sam_len_kernel--> record lengths --scan--> record offsets --sam_kernel--> SAM text
//
// Paired-end (urmapx_text_map_pe): one chunk of each mate file with the same number of records; both are parsed as above,
// the mates' bases are interleaved (reads 2i, 2i+1 = pair i, map2.cpp:27-32), State2::Search runs
// (urmapx_map_pe_device) and each record gets SetSAM2's flags, RNEXT, PNEXT and TLEN (output2.cpp:18-128).
//
// The device parser accepts exactly the files the reference accepts WITHOUT special handling: '\n' line ends, four
// lines per record, '@' first, letters only, as many quality bytes as bases.  Anything else ('\r', blank lines, a
// malformed record, a last line without '\n') makes the call hand the chunk back untouched (report.reason) and the
// caller runs it through the host reader (sam.cpp), which reproduces the reference's handling and messages.
#include <cstring>
#include <chrono>
#include <string>
#include <vector>

#include "internal.h"
#include "sam.h"

using namespace urx;

namespace {

constexpr int NL_TILE = 16384, NL_THREADS = 256;  // 64 bytes per thread
constexpr int SC_THREADS = 256, SC_ITEMS = 8, SC_TILE = SC_THREADS * SC_ITEMS;
constexpr int SAM_WAVES = 4;        // wavefronts per block of the record kernels
// an LDS array named by an LDS pointer (32 bit, ds_* instructions), as in dev_common.h
template <class T> using lds_ptr = __attribute__((address_space(3))) T *;
template <class T> __device__ __forceinline__ lds_ptr<T> to_lds(T *p) { return (lds_ptr<T>)p; }
constexpr int HEAD_CAP = 1600;      // bytes of one record between QNAME and SEQ held in LDS (12 * 97 CIGAR + fields)
constexpr int TNAME_MAX = 160;      // longest target label the device formatter takes

struct TextHdr {  // device; [0] and [1]: the chunk of each file (n_lines, n_records, flags, max_len); [0] also the call's totals
	uint32_t n_lines, n_records, flags, max_len;
	uint32_t total_bases, sam_total, n_reads, pad1;  // n_reads: records of the batch (single-end: n_records; pairs: twice that)
	unsigned long long cnt[4];  // accept, reject, nohit, unsupported
};

__device__ __forceinline__ uint32_t eq_mask(uint32_t w, uint32_t byte_x4) {  // bit 7 of every byte of w equal to the byte
	const uint32_t x = w ^ byte_x4;
	return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(v, d, 64);
		if (lane >= d) v += o;
	}
	return v;
}

// exclusive scan of one value per thread over a block of 256 threads; *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *lds4, uint32_t *total) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t inc = wave_incl_scan(v);
	__syncthreads();
	if (lane == 63) lds4[w] = inc;
	__syncthreads();
	uint32_t base = 0, sum = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t t = lds4[k];
		if (k < w) base += t;
		sum += t;
	}
	*total = sum;
	return base + inc - v;
}

// ---- line ends ----
__global__ __launch_bounds__(NL_THREADS) void nl_count_kernel(const uint4 *raw, uint32_t n_tiles, uint32_t *tile_counts, TextHdr *hdr) {
	__shared__ uint32_t lds4[4];
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint4 *p = raw + (size_t)tile * (NL_TILE / 16) + threadIdx.x * 4;
		uint32_t c = 0, cr = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint4 v = p[k];
			c += __popc(eq_mask(v.x, 0x0A0A0A0Au)) + __popc(eq_mask(v.y, 0x0A0A0A0Au)) + __popc(eq_mask(v.z, 0x0A0A0A0Au)) + __popc(eq_mask(v.w, 0x0A0A0A0Au));
			cr |= eq_mask(v.x, 0x0D0D0D0Du) | eq_mask(v.y, 0x0D0D0D0Du) | eq_mask(v.z, 0x0D0D0D0Du) | eq_mask(v.w, 0x0D0D0D0Du);
		}
		uint32_t total;
		(void)block_excl_scan_256(c, lds4, &total);
		if (threadIdx.x == 0) tile_counts[tile] = total;
		if (cr) atomicOr(&hdr->flags, 1u);  // a '\r' somewhere: the host reader's business
		__syncthreads();
	}
}

// one block: exclusive scan of m values in place; total to *total_out
// (offsets are 32 bit: a total that does not fit sets bit 6 of *wrap_flags, and the caller refuses the chunk)
__global__ __launch_bounds__(1024) void scan_small_kernel(uint32_t *v, const uint32_t *m_ptr, uint32_t m_div, uint32_t m_fixed, uint32_t *total_out,
                                                          uint32_t *wrap_flags = nullptr) {
	__shared__ uint32_t part[1024];
	__shared__ unsigned long long tot64;
	const uint32_t m = m_ptr ? (*m_ptr + m_div - 1) / m_div : m_fixed;
	const uint32_t per = (m + 1023) / 1024;
	const uint32_t lo = threadIdx.x * per, hi = lo + per < m ? lo + per : m;
	if (threadIdx.x == 0) tot64 = 0;
	__syncthreads();
	uint32_t s = 0;
	unsigned long long s64 = 0;
	for (uint32_t i = lo; i < hi; ++i) { s += v[i]; s64 += v[i]; }
	part[threadIdx.x] = s;
	if (wrap_flags && s64) atomicAdd(&tot64, s64);
	__syncthreads();
	if (wrap_flags && threadIdx.x == 0 && tot64 > 0xFFFFFFFFull) atomicOr(wrap_flags, 64u);
	if (threadIdx.x < 64) {  // 16 partials per lane
		uint32_t t = 0;
		for (int k = 0; k < 16; ++k) t += part[threadIdx.x * 16 + k];
		const uint32_t inc = wave_incl_scan(t);
		uint32_t run = inc - t;
		for (int k = 0; k < 16; ++k) {
			const uint32_t x = part[threadIdx.x * 16 + k];
			part[threadIdx.x * 16 + k] = run;
			run += x;
		}
		if (threadIdx.x == 63 && total_out) *total_out = inc;
	}
	__syncthreads();
	uint32_t run = part[threadIdx.x];
	for (uint32_t i = lo; i < hi; ++i) {
		const uint32_t x = v[i];
		v[i] = run;
		run += x;
	}
}

__global__ __launch_bounds__(NL_THREADS) void nl_emit_kernel(const uint4 *raw, uint32_t n_tiles, const uint32_t *tile_offs, uint32_t *ends,
                                                             uint32_t ends_cap, TextHdr *hdr) {
	__shared__ uint32_t lds4[4];
	if (blockIdx.x == 0 && threadIdx.x == 0) hdr->n_records = (hdr->n_lines < ends_cap ? hdr->n_lines : ends_cap) / 4;
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint4 *p = raw + (size_t)tile * (NL_TILE / 16) + threadIdx.x * 4;
		uint32_t w[16];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint4 v = p[k];
			w[4 * k] = eq_mask(v.x, 0x0A0A0A0Au); w[4 * k + 1] = eq_mask(v.y, 0x0A0A0A0Au);
			w[4 * k + 2] = eq_mask(v.z, 0x0A0A0A0Au); w[4 * k + 3] = eq_mask(v.w, 0x0A0A0A0Au);
		}
		uint32_t c = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k) c += __popc(w[k]);
		uint32_t total;
		uint32_t at = tile_offs[tile] + block_excl_scan_256(c, lds4, &total);
		const uint32_t pos0 = tile * (uint32_t)NL_TILE + threadIdx.x * 64u;
		if (c) {
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				uint32_t m = w[k];
				while (m) {
					const int b = __ffs(m) - 1;  // bit 7 of byte b/8
					m &= m - 1;
					if (at < ends_cap) ends[at] = pos0 + 4u * k + (uint32_t)(b >> 3);
					++at;
				}
			}
		}
		__syncthreads();
	}
}

// ---- records ----
__device__ __forceinline__ uint32_t line_start(const uint32_t *ends, uint32_t k) { return k ? ends[k - 1] + 1u : 0u; }

// read lengths; structure of every record ('@' first, a label line that is not empty, as many quality bytes as bases)
__global__ __launch_bounds__(256) void record_kernel(const uint8_t *raw, const uint32_t *ends, TextHdr *hdr, uint32_t *blen) {
	const uint32_t n = hdr->n_records;
	uint32_t mx = 0;
	bool bad = false;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t s1 = line_start(ends, 4 * i), e1 = ends[4 * i], e2 = ends[4 * i + 1], e3 = ends[4 * i + 2], e4 = ends[4 * i + 3];
		const uint32_t l2 = e2 - (e1 + 1), l4 = e4 - (e3 + 1);
		if (e1 == s1 || raw[s1] != '@' || l2 != l4) bad = true;
		blen[i] = l2;
		mx = l2 > mx ? l2 : mx;
	}
	for (int d = 32; d; d >>= 1) {
		const uint32_t o = __shfl_xor(mx, d, 64);
		mx = o > mx ? o : mx;
	}
	// (one address for every wavefront of the launch: 8 192 atomics were most of this kernel's 0.1 ms; a wavefront that cannot raise the maximum
	// as it stands does not try)
	if ((threadIdx.x & 63) == 0 && mx > __atomic_load_n(&hdr->max_len, __ATOMIC_RELAXED)) atomicMax(&hdr->max_len, mx);
	if (bad) atomicOr(&hdr->flags, 2u);
}

// tile sums of v[0, n)
__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(const uint32_t *v, const uint32_t *n_ptr, uint32_t *sums) {
	__shared__ uint32_t lds4[4];
	const uint32_t n = *n_ptr, n_tiles = (n + SC_TILE - 1) / SC_TILE;
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t lo = tile * SC_TILE + threadIdx.x * SC_ITEMS;
		uint32_t s = 0;
#pragma unroll
		for (int k = 0; k < SC_ITEMS; ++k)
			if (lo + k < n) s += v[lo + k];
		uint32_t total;
		(void)block_excl_scan_256(s, lds4, &total);
		if (threadIdx.x == 0) sums[tile] = total;
		__syncthreads();
	}
}

// out[i] = exclusive prefix of v (tile offsets already scanned); out[n] = the total
template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const uint32_t *v, const uint32_t *n_ptr, const uint32_t *tile_offs, OutT *out) {
	__shared__ uint32_t lds4[4];
	const uint32_t n = *n_ptr, n_tiles = (n + SC_TILE - 1) / SC_TILE;
	if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t lo = tile * SC_TILE + threadIdx.x * SC_ITEMS;
		uint32_t x[SC_ITEMS], s = 0;
#pragma unroll
		for (int k = 0; k < SC_ITEMS; ++k) {
			x[k] = lo + k < n ? v[lo + k] : 0u;
			s += x[k];
		}
		uint32_t total;
		uint32_t run = tile_offs[tile] + block_excl_scan_256(s, lds4, &total);
#pragma unroll
		for (int k = 0; k < SC_ITEMS; ++k) {
			if (lo + k < n) out[lo + k] = (OutT)run;
			run += x[k];
			if (lo + k + 1 == n) out[n] = (OutT)run;
		}
		__syncthreads();
	}
}

// bases of every read, back to back (what the mapping kernels take); letters only (fastqseqsource.cpp:76-84)
__global__ __launch_bounds__(256) void copy_bases_kernel(const uint8_t *raw, const uint32_t *ends, TextHdr *hdr, TextHdr *hdr0, const uint64_t *offs,
                                                         uint8_t *bases, uint32_t stride, uint32_t phase) {
	const uint32_t n = hdr0->n_reads / stride;  // pairs: the shorter of the two chunks (unequal chunks are handed back)
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
	bool bad = false;
	for (uint32_t i = wave; i < n; i += n_waves) {
		const uint32_t s2 = ends[4 * i] + 1u, L = ends[4 * i + 1] - s2;
		uint8_t *dst = bases + offs[(size_t)i * stride + phase];
		for (uint32_t k = lane; k < L; k += 64) {
			const uint8_t c = raw[s2 + k];
			if ((uint8_t)((c | 0x20u) - 'a') >= 26u) bad = true;
			dst[k] = c;
		}
	}
	if (bad) atomicOr(&hdr->flags, 4u);
	if (blockIdx.x == 0 && threadIdx.x == 0 && phase == 0) hdr0->total_bases = (uint32_t)offs[hdr0->n_reads];
}

// pairs: lens[2i] = length of mate 1 of pair i, lens[2i+1] = of mate 2; the two chunks must hold the same number of records
__global__ __launch_bounds__(256) void interleave_lens_kernel(TextHdr *hdr, const uint32_t *blen0, const uint32_t *blen1, uint32_t *lens) {
	const uint32_t n0 = hdr[0].n_records, n1 = hdr[1].n_records, n = n0 < n1 ? n0 : n1;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		hdr[0].n_reads = 2u * n;
		if (n0 != n1 || hdr[0].n_lines != hdr[1].n_lines) atomicOr(&hdr[0].flags, 16u);
		const uint32_t m0 = hdr[0].max_len, m1 = hdr[1].max_len;
		hdr[0].max_len = m0 > m1 ? m0 : m1;
		atomicOr(&hdr[0].flags, hdr[1].flags);
	}
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		lens[2 * i] = blen0[i];
		lens[2 * i + 1] = blen1[i];
	}
}

// ---- SAM ----
struct SamArgs {
	const uint8_t *raw[2];       // the chunk of each file (single-end: [0] only)
	const uint32_t *ends[2];
	uint32_t paired;             // records 2i, 2i+1 = the mates of pair i, from raw[0] and raw[1]
	const urmapx_result *results;
	const urmapx_path_op *ops;
	const char *tnames;          // target labels back to back
	const uint32_t *tname_offs;  // seqCount + 1
	const uint8_t *comp;         // 256-byte complement table (alpha.cpp:3005)
	uint32_t seq_count;
	uint32_t minq;
	TextHdr *hdr;
	uint32_t *lens;              // PASS 0 out
	uint32_t *qn;                // PASS 0 out: QNAME bytes of every record (PASS 1 does not scan the labels again)
	const uint32_t *rec_offs;    // PASS 1 in
	char *sam;                   // PASS 1 out
};

// Where the head of a record is written: LDS (HeadLds) or, for a head longer than the LDS buffer, straight into the output
// (HeadGlobal).  Two types, two instantiations of build_head: a generic pointer that is LDS on one path and global on the
// other makes the compiler cast the LDS address, and the cast of LDS offset 0 is a null pointer.
struct HeadLds {
	lds_ptr<char> base;
	uint32_t n;
	__device__ __forceinline__ void put(char c) { base[n++] = c; }
};
struct HeadGlobal {
	char *base;
	uint32_t n;
	__device__ __forceinline__ void put(char c) { base[n++] = c; }
};
template <class W>
__device__ __forceinline__ void dev_put_uint(W &w, uint32_t v) {
	char tmp[12];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
	while (n) w.put(tmp[--n]);
}

template <class W>
__device__ __forceinline__ void dev_put_int(W &w, int v) {
	if (v < 0) { w.put('-'); dev_put_uint(w, (uint32_t)(-(long long)v)); return; }
	dev_put_uint(w, (uint32_t)v);
}

// SetSAM's arguments (setsam.cpp:73-74): single-end passes 0, "*", UINT32_MAX, 0 (output1.cpp:13); pairs what SetSAM2
// works out (output2.cpp:61-128)
struct MateFields {
	uint32_t flags;
	bool mate_mapped;
	uint32_t mate_seq_index, mate_coord;
	int tlen;
};

__device__ __forceinline__ uint32_t paired_flags(bool first, bool revcomp, bool mate_revcomp, bool mate_unmapped) {  // output2.cpp:18-36
	uint32_t f = first ? 0x41u : 0x81u;
	if (revcomp) f |= 0x10u;
	if (mate_unmapped) f |= 0x08u;
	else if (mate_revcomp) f |= 0x20u;
	return f;
}

// SetSAM2 (output2.cpp:61-128) for mate `second` of a pair with results r1, r2 and read lengths len1, len2
__device__ MateFields pair_fields(const urmapx_result &r1, const urmapx_result &r2, uint32_t len1, uint32_t len2, bool second) {
	const bool m1 = r1.dbpos != 0xFFFFFFFFu, m2 = r2.dbpos != 0xFFFFFFFFu;
	const bool plus1 = m1 && r1.plus, plus2 = m2 && r2.plus;
	const bool consistent = m1 && m2 && (plus1 != plus2);
	int tlen1 = 0, tlen2 = 0;
	bool proper = false;
	if (m1 && m2) {
		if (r1.coord <= r2.coord) {
			tlen1 = (int)(r2.coord + len2) - (int)r1.coord;
			if (tlen1 > 0 && tlen1 < 1000 && consistent) proper = true;
			if (tlen1 > 1000) tlen1 = 0;
			tlen2 = -tlen1;
		} else {
			tlen2 = (int)(r1.coord + len1) - (int)r2.coord;
			if (tlen2 > 0 && tlen2 < 1000 && consistent) proper = true;
			if (tlen2 > 1000) tlen2 = 0;
			tlen1 = -tlen2;
		}
	}
	const bool rc1 = m1 && !r1.plus, rc2 = m2 && !r2.plus;
	MateFields F;
	F.flags = second ? paired_flags(false, rc2, rc1, !m1) : paired_flags(true, rc1, rc2, !m2);
	if (proper) F.flags |= 2u;
	F.mate_mapped = second ? m1 : m2;
	F.mate_seq_index = second ? r1.seq_index : r2.seq_index;
	F.mate_coord = second ? r1.coord : r2.coord;
	F.tlen = second ? tlen2 : tlen1;
	return F;
}

// Lane 0 writes the fields between QNAME and SEQ of a mapped or unmapped record (SetSAM / SetSAM_Unmapped,
// setsam.cpp:12-207) to `head`; returns the length, or 0 if it does not fit the device formatter.
// The merged CIGAR runs of a path, seen from both ends: N of them, the first three (fo / fl) and the last three (lo / ll, [2] = the
// last), and the characters they print as.
struct CigarEnds {
	uint32_t N, chars;
	char fo[3], lo[3];
	uint32_t fl[3], ll[3];
};
__device__ __forceinline__ uint32_t dev_digits(uint32_t v) {
	return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u
	     : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ void cigar_ends(const urmapx_path_op *ops, uint32_t nops, CigarEnds &E) {
	E.N = 0; E.chars = 0;
	for (int t = 0; t < 3; ++t) { E.fo[t] = 0; E.lo[t] = 0; E.fl[t] = 0; E.ll[t] = 0; }
	char cur = 0;
	uint32_t curlen = 0;
	bool have = false;
	auto close_run = [&]() {
		if (E.N < 3) { E.fo[E.N] = cur; E.fl[E.N] = curlen; }
		E.lo[0] = E.lo[1]; E.ll[0] = E.ll[1]; E.lo[1] = E.lo[2]; E.ll[1] = E.ll[2]; E.lo[2] = cur; E.ll[2] = curlen;
		E.chars += dev_digits(curlen) + 1u;
		++E.N;
	};
	for (uint32_t i = 0; i < nops; ++i) {
		const uint32_t code = ops[i] & 3u, len = ops[i] >> 2;
		const char c = code == 0 ? 'M' : code == 1 ? 'I' : 'D';
		if (have && cur == c) curlen += len;
		else {
			if (have) close_run();
			cur = c; curlen = len; have = true;
		}
	}
	if (have) close_run();
}

template <class W>
__device__ uint32_t build_head(const SamArgs &A, const urmapx_result &r, const MateFields &F, uint32_t QL, W &w) {
	if (r.dbpos == 0xFFFFFFFFu) {
		uint32_t flags = 0x04u;  // SetSAM_Unmapped keeps these bits of the flags it is given (setsam.cpp:14-27)
		if (F.flags & 0x01u) flags |= 0x01u;
		if (F.flags & 0x40u) flags |= 0x40u;
		else if (F.flags & 0x80u) flags |= 0x80u;
		if (F.flags & 0x08u) flags |= 0x08u;
		else if (F.flags & 0x20u) flags |= 0x20u;
		w.put('\t');
		dev_put_uint(w, flags);
		const char s[] = "\t*\t0\t0\t*\t*\t0\t0\t";
		for (int i = 0; i < (int)sizeof(s) - 1; ++i) w.put(s[i]);
		return w.n;
	}
	w.put('\t');
	dev_put_uint(w, F.flags);
	w.put('\t');
	if (r.seq_index >= A.seq_count) return 0;
	const uint32_t t0 = A.tname_offs[r.seq_index], tl = A.tname_offs[r.seq_index + 1] - t0;
	if (tl > (uint32_t)TNAME_MAX) return 0;
	for (uint32_t i = 0; i < tl; ++i) w.put(A.tnames[t0 + i]);
	w.put('\t');
	dev_put_uint(w, r.coord + 1u);
	w.put('\t');
	dev_put_uint(w, r.mapq);
	w.put('\t');
	const uint32_t nops = r.path_nops;
	if (nops == 0) { dev_put_uint(w, QL); w.put('M'); }
	else {
		// Any number of runs (the general kernels' paths are as long as the read): the runs are merged as they stream by, twice --
		// once for what CIGAROpsFixDanglingMs (cigar.cpp:141-199) needs to know (how many merged runs, the first three, the last
		// three: head rule XOR tail rule, as in sam.cpp), once to write them.
		const urmapx_path_op *ops = A.ops + r.path_off;
		CigarEnds E;
		cigar_ends(ops, nops, E);
		const bool head_rule = E.N >= 3 && E.fo[0] == 'M' && E.fl[0] <= 2 && E.fl[1] > 4 && E.fo[2] == 'M';
		const bool tail_rule = !head_rule && E.N >= 3 && E.lo[2] == 'M' && E.ll[2] <= 2 && E.ll[1] > 4 && E.lo[0] == 'M';
		uint32_t k = 0;  // index of the merged run being closed
		char cur = 0;
		uint32_t curlen = 0;
		bool have = false;
		auto close_run = [&]() {
			uint32_t len = curlen;
			bool skip = false;
			if (head_rule) { if (k == 0) skip = true; else if (k == 2) len += E.fl[0]; }
			if (tail_rule) { if (k == E.N - 1) skip = true; else if (k == E.N - 3) len += E.ll[2]; }
			if (!skip) { dev_put_uint(w, len); w.put(cur); }
			++k;
		};
		for (uint32_t i = 0; i < nops; ++i) {
			const uint32_t code = ops[i] & 3u, len = ops[i] >> 2;
			const char c = code == 0 ? 'M' : code == 1 ? 'I' : 'D';  // path D (query only) is CIGAR I and vice versa (cigar.cpp:22-25)
			if (have && cur == c) curlen += len;
			else {
				if (have) close_run();
				cur = c; curlen = len; have = true;
			}
		}
		close_run();
	}
	w.put('\t');
	// RNEXT: '*' without a mapped mate, '=' if the mate's target has the same label, else that label (setsam.cpp:150-166)
	if (!F.mate_mapped) w.put('*');
	else {
		if (F.mate_seq_index >= A.seq_count) return 0;
		const uint32_t m0 = A.tname_offs[F.mate_seq_index], ml = A.tname_offs[F.mate_seq_index + 1] - m0;
		if (ml > (uint32_t)TNAME_MAX) return 0;
		bool same = ml == tl;
		for (uint32_t i = 0; same && i < tl; ++i) same = A.tnames[t0 + i] == A.tnames[m0 + i];
		if (ml == 0 || (ml == 1 && A.tnames[m0] == '*')) w.put('*');
		else if (same) w.put('=');
		else
			for (uint32_t i = 0; i < ml; ++i) w.put(A.tnames[m0 + i]);
	}
	w.put('\t');
	if (!F.mate_mapped || F.mate_coord == 0 || F.mate_coord == 0xFFFFFFFFu) w.put('0');  // position 0 prints as 0 (setsam.cpp:168-172)
	else dev_put_uint(w, F.mate_coord + 1u);
	w.put('\t');
	dev_put_int(w, F.tlen);
	w.put('\t');
	return w.n;
}


// Length of what build_head writes, without writing it (the same tests in the same order; 0 = does not fit).  The CIGAR
// runs are merged as they stream by; the dangling-M rule needs the first three and the last three merged runs only.
__device__ uint32_t head_length(const SamArgs &A, const urmapx_result &r, const MateFields &F, uint32_t QL) {
	if (r.dbpos == 0xFFFFFFFFu) {
		uint32_t flags = 0x04u;
		if (F.flags & 0x01u) flags |= 0x01u;
		if (F.flags & 0x40u) flags |= 0x40u;
		else if (F.flags & 0x80u) flags |= 0x80u;
		if (F.flags & 0x08u) flags |= 0x08u;
		else if (F.flags & 0x20u) flags |= 0x20u;
		return 1u + dev_digits(flags) + (uint32_t)(sizeof("\t*\t0\t0\t*\t*\t0\t0\t") - 1);
	}
	if (r.seq_index >= A.seq_count) return 0;
	const uint32_t t0 = A.tname_offs[r.seq_index], tl = A.tname_offs[r.seq_index + 1] - t0;
	if (tl > (uint32_t)TNAME_MAX) return 0;
	uint32_t n = 1u + dev_digits(F.flags) + 1u + tl + 1u + dev_digits(r.coord + 1u) + 1u + dev_digits(r.mapq) + 1u;
	const uint32_t nops = r.path_nops;
	if (nops == 0) n += dev_digits(QL) + 1u;
	else {
		const urmapx_path_op *ops = A.ops + r.path_off;
		CigarEnds E;
		cigar_ends(ops, nops, E);
		uint32_t chars = E.chars;
		if (E.N >= 3) {
			if (E.fo[0] == 'M' && E.fl[0] <= 2 && E.fl[1] > 4 && E.fo[2] == 'M') {
				chars -= dev_digits(E.fl[0]) + 1u;
				chars += dev_digits(E.fl[2] + E.fl[0]) - dev_digits(E.fl[2]);
			} else if (E.lo[2] == 'M' && E.ll[2] <= 2 && E.ll[1] > 4 && E.lo[0] == 'M') {
				chars -= dev_digits(E.ll[2]) + 1u;
				chars += dev_digits(E.ll[0] + E.ll[2]) - dev_digits(E.ll[0]);
			}
		}
		n += chars;
	}
	n += 1u;  // the tab after CIGAR
	if (!F.mate_mapped) n += 1u;
	else {
		if (F.mate_seq_index >= A.seq_count) return 0;
		const uint32_t m0 = A.tname_offs[F.mate_seq_index], ml = A.tname_offs[F.mate_seq_index + 1] - m0;
		if (ml > (uint32_t)TNAME_MAX) return 0;
		bool same = ml == tl;
		for (uint32_t i = 0; same && i < tl; ++i) same = A.tnames[t0 + i] == A.tnames[m0 + i];
		if (ml == 0 || (ml == 1 && A.tnames[m0] == '*')) n += 1u;
		else if (same) n += 1u;
		else n += ml;
	}
	n += 1u;
	if (!F.mate_mapped || F.mate_coord == 0 || F.mate_coord == 0xFFFFFFFFu) n += 1u;
	else n += dev_digits(F.mate_coord + 1u);
	n += 1u;
	n += F.tlen < 0 ? 1u + dev_digits((uint32_t)(-(long long)F.tlen)) : dev_digits((uint32_t)F.tlen);
	n += 1u;
	return n;
}

// what a record has besides its result: where its lines are, QNAME length ("/1" "/2" dropped, cut at the first blank:
// setsam.cpp:36-46), SetSAM's mate arguments
struct RecView {
	const uint8_t *raw;
	uint32_t s1, e1, e3, QL;
	MateFields F;
};
__device__ __forceinline__ RecView record_view(const SamArgs &A, uint32_t i, const urmapx_result &r) {
	RecView V;
	const uint32_t side = A.paired ? (i & 1u) : 0u, rec = A.paired ? (i >> 1) : i;
	V.raw = A.raw[side];
	const uint32_t *ends = A.ends[side];
	V.s1 = line_start(ends, 4 * rec); V.e1 = ends[4 * rec];
	const uint32_t e2 = ends[4 * rec + 1];
	V.e3 = ends[4 * rec + 2];
	V.QL = e2 - (V.e1 + 1u);
	V.F.flags = 0; V.F.mate_mapped = false; V.F.mate_seq_index = 0; V.F.mate_coord = 0xFFFFFFFFu; V.F.tlen = 0;
	if (A.paired) {
		const urmapx_result rm = A.results[i ^ 1u];
		const uint32_t *oe = A.ends[side ^ 1u];
		const uint32_t QLm = oe[4 * rec + 1] - (oe[4 * rec] + 1u);
		V.F = side ? pair_fields(rm, r, QLm, V.QL, true) : pair_fields(r, rm, V.QL, QLm, false);
	}
	return V;
}

// Record lengths and the HitStats counters (output1.cpp:20-30), one THREAD per record: nothing is written but a number,
// so the serial part of a record (the head between QNAME and SEQ) runs for 64 records at a time.
__global__ __launch_bounds__(256) void sam_len_kernel(SamArgs A) {
	const uint32_t n = A.hdr->n_reads;
	uint32_t c_acc = 0, c_rej = 0, c_no = 0, c_uns = 0;
	bool bad = false;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const urmapx_result r = A.results[i];
		const RecView V = record_view(A, i, r);
		const uint8_t *label = V.raw + V.s1 + 1u;
		uint32_t ln = V.e1 - V.s1 - 1u;
		if (ln > 2 && label[ln - 2] == '/' && (label[ln - 1] == '1' || label[ln - 1] == '2')) ln -= 2;
		uint32_t qn = 0;
		while (qn < ln && label[qn] != ' ' && label[qn] != '\t') ++qn;
		const uint32_t hl = head_length(A, r, V.F, V.QL);
		A.lens[i] = qn + hl + 2u * V.QL + 2u;
		A.qn[i] = qn;
		if (hl == 0) bad = true;
		if (r.status) ++c_uns;
		if (r.dbpos == 0xFFFFFFFFu) ++c_no;
		else if (r.mapq >= A.minq) ++c_acc;
		else ++c_rej;
	}
	if (bad) atomicOr(&A.hdr->flags, 8u);
	for (int d = 32; d; d >>= 1) {
		c_acc += __shfl_xor(c_acc, d, 64); c_rej += __shfl_xor(c_rej, d, 64);
		c_no += __shfl_xor(c_no, d, 64); c_uns += __shfl_xor(c_uns, d, 64);
	}
	// the four counters live at four addresses for the whole launch: one atomic per BLOCK and counter, not per wavefront (the atomics of
	// 8 192 wavefronts on two addresses were two thirds of this kernel's 0.3 ms per 524 288 records)
	__shared__ uint32_t s_cnt[4];
	if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) {
		if (c_acc) atomicAdd(&s_cnt[0], c_acc);
		if (c_rej) atomicAdd(&s_cnt[1], c_rej);
		if (c_no) atomicAdd(&s_cnt[2], c_no);
		if (c_uns) atomicAdd(&s_cnt[3], c_uns);
	}
	__syncthreads();
	if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&A.hdr->cnt[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// The records' bytes, 64 records per wavefront at a time.  First every LANE builds the head of its own record (the fields between
// QNAME and SEQ: a serial string of digits and tabs, 40-80 bytes) in its slice of LDS -- the one-wavefront-per-record kernel of rounds
// 3-4 had lane 0 do this while 63 lanes waited: 3.6 ms per 1 M records, nine tenths of it that string --, then the wavefront goes
// through its 64 records and copies QNAME, the head out of LDS, SEQ and QUAL with all lanes (reverse-complemented / reversed for a
// minus-strand hit).  A head longer than a slice (a CIGAR of many runs, a long target label) is built as before: by lane 0 in the
// wavefront's large LDS buffer, or, longer than that, straight into the output.  A record's length must be the one sam_len_kernel
// reserved; if not, the chunk is flagged and handed back (flag 32).
constexpr int HEAD_SMALL = 96;
__device__ __forceinline__ uint32_t bcast(uint32_t v, int t) { return (uint32_t)__builtin_amdgcn_readlane((int)v, t); }

__global__ __launch_bounds__(SAM_WAVES * 64) void sam_kernel(SamArgs A) {
	__shared__ char s_heads[SAM_WAVES][64 * HEAD_SMALL];
	__shared__ char s_big[SAM_WAVES][HEAD_CAP];
	__shared__ uint32_t s_hl[SAM_WAVES];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t n = A.hdr->n_reads;
	const uint32_t wave = blockIdx.x * SAM_WAVES + w, n_waves = gridDim.x * SAM_WAVES;
	for (uint32_t base = wave * 64u; base < n; base += n_waves * 64u) {
		const uint32_t i = base + (uint32_t)lane;
		const bool valid = i < n;
		uint32_t qn = 0, off = 0, hl = 0, want = 0, QL = 0, s1 = 0, e1 = 0, e3 = 0, fl = 0;  // fl: 1 small, 2 bad, 4 plus
		if (valid) {
			const urmapx_result r = A.results[i];
			const RecView V = record_view(A, i, r);
			qn = A.qn[i]; off = A.rec_offs[i]; QL = V.QL; s1 = V.s1; e1 = V.e1; e3 = V.e3;
			const uint32_t reserved = A.lens[i], fixed = qn + 2u * QL + 2u;
			want = reserved - fixed;
			if (reserved <= fixed) fl |= 2u;  // (a head is never empty)
			else if (want <= (uint32_t)HEAD_SMALL) {
				HeadLds hw{to_lds(&s_heads[w][lane * HEAD_SMALL]), 0u};
				hl = build_head(A, r, V.F, QL, hw);
				fl |= hl == want ? 1u : 2u;
			}
			if (r.dbpos == 0xFFFFFFFFu || r.plus) fl |= 4u;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const int nrec = (int)(n - base < 64u ? n - base : 64u);
		for (int t = 0; t < nrec; ++t) {
			const uint32_t t_fl = bcast(fl, t), t_qn = bcast(qn, t), t_QL = bcast(QL, t), t_want = bcast(want, t);
			const uint32_t t_s1 = bcast(s1, t), t_e1 = bcast(e1, t), t_e3 = bcast(e3, t);
			const uint32_t rec_i = base + (uint32_t)t;
			const uint8_t *raw = A.raw[A.paired ? (rec_i & 1u) : 0u];
			char *out = A.sam + bcast(off, t);
			uint32_t t_hl = bcast(hl, t);
			bool direct = false;
			if (!(t_fl & 3u)) {  // a long head: lane 0 writes it into the large buffer or, longer than that, where it belongs
				direct = t_want > (uint32_t)HEAD_CAP - 64u;
				if (lane == 0) {
					const urmapx_result r = A.results[rec_i];
					const RecView V = record_view(A, rec_i, r);
					if (direct) { HeadGlobal hw{out + t_qn, 0u}; s_hl[w] = build_head(A, r, V.F, t_QL, hw); }
					else { HeadLds hw{to_lds(&s_big[w][0]), 0u}; s_hl[w] = build_head(A, r, V.F, t_QL, hw); }
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				t_hl = s_hl[w];
			}
			if ((t_fl & 2u) || t_hl != t_want) {  // never: the two kernels count the same bytes
				if (lane == 0) atomicOr(&A.hdr->flags, 32u);
#ifdef URX_DEBUG_TEXT
				if (lane == 0) printf("sam_kernel mismatch: rec %u qn %u hl %u want %u QL %u fl %u direct %d\n", rec_i, t_qn, t_hl, t_want, t_QL, t_fl, (int)direct);
#endif
				continue;
			}
			const uint8_t *label = raw + t_s1 + 1u;
			for (uint32_t k = lane; k < t_qn; k += 64) out[k] = (char)label[k];
			out += t_qn;
			if (t_fl & 1u) {
				const char *h = &s_heads[w][t * HEAD_SMALL];
				for (uint32_t k = lane; k < t_hl; k += 64) out[k] = h[k];
			} else if (!direct)
				for (uint32_t k = lane; k < t_hl; k += 64) out[k] = s_big[w][k];
			out += t_hl;
			const uint8_t *seq = raw + t_e1 + 1u, *qual = raw + t_e3 + 1u;
			if (t_fl & 4u) {
				for (uint32_t k = lane; k < t_QL; k += 64) { out[k] = (char)seq[k]; out[t_QL + 1u + k] = (char)qual[k]; }
			} else {
				for (uint32_t k = lane; k < t_QL; k += 64) { out[k] = (char)A.comp[seq[t_QL - 1u - k]]; out[t_QL + 1u + k] = (char)qual[t_QL - 1u - k]; }
			}
			if (lane == 0) { out[t_QL] = '\t'; out[2u * t_QL + 1u] = '\n'; }
		}
		__builtin_amdgcn_wave_barrier();  // (the next 64 heads go where these were)
	}
}

}  // namespace

struct urmapx_text {
	urmapx_ctx *C = nullptr;
	DevBuf<uint8_t> raw[2], bases, sam, comp;
	DevBuf<uint32_t> tile_counts[2], ends[2], blen[2], sums, lens, qn, rec_offs, used, tname_offs;
	DevBuf<uint64_t> offs;
	DevBuf<char> tnames;
	DevBuf<urmapx_result> results;
	DevBuf<urmapx_path_op> pathops;
	DevBuf<TextHdr> hdr;  // [2]
	uint32_t seq_count = 0;
	TextHdr *h_hdr = nullptr;  // page-locked, [4]: [0..1] what map_text reads in between, [2 + set] the header behind a chunk's sam_kernel
	// where the chunk's time on the stream goes (urmapx_text_report.ms_*): start, after each side's H2D [1..2], before the
	// mapping kernels [3], after them [4], before the copy back [5], after it [6]
	hipEvent_t ev[7] = {};
	bool ev_ok = false;
	int ev_sides = 1;
	// The copy back as a stage of its own (urmapx_text_set_deferred): the text of chunk i crosses PCIe on copy_st while the main stream
	// takes chunk i + 1 in, parses and maps it.  Two chunks are in flight at most; what belongs to a chunk's tail exists twice
	// (set = chunk parity): the events around its sam_kernel and its copy, the header copied behind sam_kernel, its report.
	hipStream_t copy_st = nullptr;
	bool deferred = false;
	hipEvent_t tail_ev[2][3] = {};  // [set]: before sam_kernel's group (= ev[4] of that chunk), before the copy, after it
	bool tail_ok = false;
	uint32_t chunk_no = 0;          // chunks whose tail has been enqueued
	int waiting = 0;                // tails enqueued and not waited for (deferred mode: 0..2)
	urmapx_text_report tail_rep[2];
	// a chunk mapped and measured whose text has not been fetched (urmapx_text_fetch_sam)
	uint32_t last_pairs = 0;  // pairs of the last chunk urmapx_text_map_pe mapped (urmapx_text_fetch_pairs)
	bool pending = false;
	SamArgs pending_args;
	urmapx_text_report pending_rep;
};

namespace {

constexpr int GRID = 2048;

// H2D of one file's chunk, its line ends, its records' lengths and checks
int parse_side(urmapx_text *T, int side, const char *fastq, size_t nbytes, uint32_t *ends_cap_out) {
	hipStream_t st = ctx_stream(T->C);
	const uint32_t n_tiles = (uint32_t)((nbytes + NL_TILE - 1) / NL_TILE);
	const size_t padded = (size_t)n_tiles * NL_TILE;
	// a record's four lines take at least five bytes ("@\n\n\n\n"); more line ends than that is not FASTQ
	const uint32_t ends_cap = (uint32_t)(nbytes / 5 * 4 + 16);
	const uint32_t rec_cap = ends_cap / 4 + 1;
	int rc;
	if ((rc = T->raw[side].ensure(padded + 16))) return rc;
	if ((rc = T->tile_counts[side].ensure(n_tiles))) return rc;
	if ((rc = T->ends[side].ensure(ends_cap))) return rc;
	if ((rc = T->blen[side].ensure(rec_cap))) return rc;
	TextHdr *hdr = T->hdr.p + side;
	const uint8_t *raw = T->raw[side].p;
	if (T->ev_ok && side == 0) HIP_TRY(hipEventRecord(T->ev[0], st));
	HIP_TRY(hipMemcpyAsync(T->raw[side].p, fastq, nbytes, hipMemcpyHostToDevice, st));
	if (T->ev_ok) HIP_TRY(hipEventRecord(T->ev[1 + side], st));
	if (padded + 16 > nbytes) HIP_TRY(hipMemsetAsync(T->raw[side].p + nbytes, 0, padded + 16 - nbytes, st));
	hipLaunchKernelGGL(nl_count_kernel, dim3(GRID), dim3(NL_THREADS), 0, st, (const uint4 *)raw, n_tiles, T->tile_counts[side].p, hdr);
	hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, T->tile_counts[side].p, (const uint32_t *)nullptr, 1u, n_tiles, &hdr->n_lines);
	hipLaunchKernelGGL(nl_emit_kernel, dim3(GRID), dim3(NL_THREADS), 0, st, (const uint4 *)raw, n_tiles, T->tile_counts[side].p, T->ends[side].p, ends_cap, hdr);
	hipLaunchKernelGGL(record_kernel, dim3(GRID), dim3(256), 0, st, raw, T->ends[side].p, hdr, T->blen[side].p);
	HIP_TRY(hipGetLastError());
	*ends_cap_out = ends_cap;
	return URMAPX_OK;
}

// a chunk's tail has crossed PCIe: the times of its last two stages, the formatter's verdict
int finish_tail(urmapx_text *T, int set, urmapx_text_report *rep) {
	if (T->tail_ok) HIP_TRY(hipEventSynchronize(T->tail_ev[set][2]));
	else {
		// no events (their creation failed): the header's copy went out on the main stream, the text's on the main stream too -- deferred
		// mode is refused without events (urmapx_text_set_deferred) -- so that is the stream to wait for (copy_st was waited for, idle)
		HIP_TRY(hipStreamSynchronize(ctx_stream(T->C)));
		if (T->copy_st) HIP_TRY(hipStreamSynchronize(T->copy_st));
	}
	*rep = T->tail_rep[set];
	if (T->tail_ok) {
		float d = 0, e = 0;
		(void)hipEventElapsedTime(&d, T->tail_ev[set][0], T->tail_ev[set][1]);
		(void)hipEventElapsedTime(&e, T->tail_ev[set][1], T->tail_ev[set][2]);
		rep->ms_format = d; rep->ms_d2h = e;
	}
	if (T->h_hdr[2 + set].flags & 32u) {  // the two record kernels disagreed on a length: the text is not trusted
		rep->reason = URMAPX_TEXT_INTERNAL;
		rep->records = 0;
	}
	return URMAPX_OK;
}

int fetch_sam(urmapx_text *T, char *sam, size_t sam_cap, urmapx_text_report *rep) {
	*rep = T->pending_rep;
	if (!sam || rep->sam_bytes > sam_cap) { rep->reason = URMAPX_TEXT_SAM_CAP; rep->records = 0; return URMAPX_OK; }
	hipStream_t st = ctx_stream(T->C);
	int rc;
	const int set = (int)(T->chunk_no & 1u);
	if (T->sam.cap < (size_t)rep->sam_bytes + 64 && T->waiting) HIP_TRY(hipStreamSynchronize(T->copy_st));  // (the array is about to be replaced under a copy)
	if ((rc = T->sam.ensure((size_t)rep->sam_bytes + 64))) return rc;
	SamArgs A = T->pending_args;
	A.sam = (char *)T->sam.p;
	// the text of the chunk before may still be on its way out of T->sam
	if (T->deferred && T->waiting && T->tail_ok) HIP_TRY(hipStreamWaitEvent(st, T->tail_ev[set ^ 1][2], 0));
	hipLaunchKernelGGL(sam_kernel, dim3(GRID), dim3(SAM_WAVES * 64), 0, st, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(T->h_hdr + 2 + set, T->hdr.p, sizeof(TextHdr), hipMemcpyDeviceToHost, st));  // (on the main stream: the next chunk clears the header there)
	if (T->tail_ok) HIP_TRY(hipEventRecord(T->tail_ev[set][1], st));
	hipStream_t cs = T->deferred ? T->copy_st : st;
	if (T->deferred) {
		if (T->tail_ok) HIP_TRY(hipStreamWaitEvent(cs, T->tail_ev[set][1], 0));
		else HIP_TRY(hipStreamSynchronize(st));
	}
	HIP_TRY(hipMemcpyAsync(sam, T->sam.p, rep->sam_bytes, hipMemcpyDeviceToHost, cs));
	if (T->tail_ok) HIP_TRY(hipEventRecord(T->tail_ev[set][2], cs));
	T->tail_rep[set] = *rep;
	T->pending = false;
	++T->chunk_no;
	if (T->deferred) { ++T->waiting; rep->reason = URMAPX_TEXT_DEFERRED; return URMAPX_OK; }
	return finish_tail(T, set, rep);
}

// fastq2 == nullptr: single-end
int map_text(urmapx_text *T, const char *fastq1, size_t nbytes1, const char *fastq2, size_t nbytes2, unsigned minq, char *sam, size_t sam_cap,
             urmapx_text_report *rep) {
	const bool paired = fastq2 != nullptr;
	memset(rep, 0, sizeof *rep);
	if (T->waiting >= 2) return URMAPX_E_ARG;  // deferred mode: two chunks in flight at most (urmapx_text_wait takes the older one)
	T->pending = false;
	if (nbytes1 == 0 && (!paired || nbytes2 == 0)) return URMAPX_OK;
	if (nbytes1 > (1u << 30) || nbytes2 > (1u << 30)) { rep->reason = URMAPX_TEXT_TOO_LARGE; return URMAPX_OK; }
	if (nbytes1 == 0 || fastq1[nbytes1 - 1] != '\n' || (paired && (nbytes2 == 0 || fastq2[nbytes2 - 1] != '\n'))) { rep->reason = URMAPX_TEXT_RAGGED; return URMAPX_OK; }
	urmapx_ctx *C = T->C;
	HIP_TRY(hipSetDevice(ctx_device(C)));
	hipStream_t st = ctx_stream(C);
	TextHdr *hdr = T->hdr.p;
	HIP_TRY(hipMemsetAsync(hdr, 0, 2 * sizeof(TextHdr), st));
	uint32_t ends_cap[2] = {0, 0};
	int rc;
	if ((rc = parse_side(T, 0, fastq1, nbytes1, &ends_cap[0]))) return rc;
	if (paired && (rc = parse_side(T, 1, fastq2, nbytes2, &ends_cap[1]))) return rc;
	const size_t rec_cap = (size_t)ends_cap[0] / 4 + (size_t)ends_cap[1] / 4 + 2;
	if ((rc = T->lens.ensure(rec_cap))) return rc;
	if ((rc = T->qn.ensure(rec_cap))) return rc;
	if ((rc = T->rec_offs.ensure(rec_cap + 1))) return rc;
	if ((rc = T->offs.ensure(rec_cap + 1))) return rc;
	if ((rc = T->sums.ensure(rec_cap / SC_TILE + 2))) return rc;
	if ((rc = T->bases.ensure(nbytes1 + nbytes2 + 64))) return rc;
	const uint32_t *read_lens = T->blen[0].p;
	if (paired) {
		hipLaunchKernelGGL(interleave_lens_kernel, dim3(GRID), dim3(256), 0, st, hdr, T->blen[0].p, T->blen[1].p, T->lens.p);
		read_lens = T->lens.p;
	} else
		HIP_TRY(hipMemcpyAsync(&hdr->n_reads, &hdr->n_records, 4, hipMemcpyDeviceToDevice, st));
	hipLaunchKernelGGL(scan_sums_kernel, dim3(GRID), dim3(SC_THREADS), 0, st, read_lens, &hdr->n_reads, T->sums.p);
	hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, T->sums.p, &hdr->n_reads, (uint32_t)SC_TILE, 0u, (uint32_t *)nullptr);
	hipLaunchKernelGGL(scan_apply_kernel<uint64_t>, dim3(GRID), dim3(SC_THREADS), 0, st, read_lens, &hdr->n_reads, T->sums.p, T->offs.p);
	for (int side = 0; side < (paired ? 2 : 1); ++side)
		hipLaunchKernelGGL(copy_bases_kernel, dim3(GRID), dim3(256), 0, st, T->raw[side].p, T->ends[side].p, hdr + side, hdr, T->offs.p, T->bases.p,
		                   paired ? 2u : 1u, (uint32_t)side);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(T->h_hdr, hdr, 2 * sizeof(TextHdr), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const TextHdr h1 = T->h_hdr[0], h1b = T->h_hdr[1];
	const uint32_t fl = h1.flags | (paired ? h1b.flags : 0u);
	if (fl & 1u) { rep->reason = URMAPX_TEXT_CR; return URMAPX_OK; }
	if (h1.n_lines > ends_cap[0] || (h1.n_lines & 3u) || (paired && (h1b.n_lines > ends_cap[1] || (h1b.n_lines & 3u)))) { rep->reason = URMAPX_TEXT_RAGGED; return URMAPX_OK; }
	if (fl & 16u) { rep->reason = URMAPX_TEXT_UNEQUAL; return URMAPX_OK; }
	if (fl & 6u) { rep->reason = URMAPX_TEXT_BAD_RECORD; return URMAPX_OK; }
	const uint32_t n = h1.n_reads;
	if (n == 0) return URMAPX_OK;
	if ((rc = T->results.ensure(n))) return rc;
	if ((rc = T->pathops.ensure((size_t)n * URMAPX_MAX_PATH_OPS))) return rc;
	if (T->ev_ok) HIP_TRY(hipEventRecord(T->ev[3], st));
	const auto enq0 = std::chrono::steady_clock::now();
	if (paired) {
		const uint32_t mx = h1.max_len > MAX_QL_PE ? MAX_QL_PE : h1.max_len;
		rc = urmapx_map_pe_device(C, T->bases.p, T->offs.p, n / 2, h1.total_bases, mx, T->results.p, T->pathops.p, T->used.p);
	} else {
		const uint32_t mx = h1.max_len > URMAPX_MAX_QL_SLOW ? URMAPX_MAX_QL_SLOW : h1.max_len;
		rc = urmapx_map_se_device(C, T->bases.p, T->offs.p, n, h1.total_bases, mx, T->results.p, T->pathops.p, T->used.p);
	}
	if (rc) return rc;
	if (T->ev_ok) HIP_TRY(hipEventRecord(T->ev[4], st));
	if (T->tail_ok) HIP_TRY(hipEventRecord(T->tail_ev[T->chunk_no & 1u][0], st));
	const float enq_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - enq0).count();
	T->ev_sides = paired ? 2 : 1;
	T->last_pairs = paired ? n / 2 : 0;
	SamArgs A;
	A.raw[0] = T->raw[0].p; A.raw[1] = T->raw[1].p; A.ends[0] = T->ends[0].p; A.ends[1] = T->ends[1].p; A.paired = paired ? 1u : 0u;
	A.results = T->results.p; A.ops = T->pathops.p; A.tnames = T->tnames.p;
	A.tname_offs = T->tname_offs.p; A.comp = T->comp.p; A.seq_count = T->seq_count; A.minq = minq; A.hdr = hdr;
	A.lens = T->lens.p; A.qn = T->qn.p; A.rec_offs = T->rec_offs.p; A.sam = nullptr;
	hipLaunchKernelGGL(sam_len_kernel, dim3(GRID), dim3(256), 0, st, A);
	hipLaunchKernelGGL(scan_sums_kernel, dim3(GRID), dim3(SC_THREADS), 0, st, T->lens.p, &hdr->n_reads, T->sums.p);
	hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, T->sums.p, &hdr->n_reads, (uint32_t)SC_TILE, 0u, &hdr->sam_total, &hdr->flags);
	hipLaunchKernelGGL(scan_apply_kernel<uint32_t>, dim3(GRID), dim3(SC_THREADS), 0, st, T->lens.p, &hdr->n_reads, T->sums.p, T->rec_offs.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(T->h_hdr, hdr, sizeof(TextHdr), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const TextHdr h2 = T->h_hdr[0];
	if (h2.flags & 8u) { rep->reason = URMAPX_TEXT_LONG_NAME; return URMAPX_OK; }
	if (h2.flags & 64u) { rep->reason = URMAPX_TEXT_TOO_LARGE; return URMAPX_OK; }  // the chunk's SAM text is over 4 GiB (record offsets are 32 bit)
	T->pending = true;
	T->pending_args = A;
	memset(&T->pending_rep, 0, sizeof T->pending_rep);
	T->pending_rep.records = n;
	T->pending_rep.ms_map_enqueue = enq_ms;
	if (T->ev_ok) {  // everything up to the mapping is behind the synchronisation above (the parse share holds the second side's H2D of a
		             // pair chunk's other file only as far as it is not copy)
		float a = 0, b = 0, c = 0, f = 0;
		(void)hipEventElapsedTime(&a, T->ev[0], T->ev[1]);
		if (T->ev_sides == 2) (void)hipEventElapsedTime(&f, T->ev[1], T->ev[2]);  // side 0's parse kernels + side 1's copy
		(void)hipEventElapsedTime(&b, T->ev[T->ev_sides], T->ev[3]);
		(void)hipEventElapsedTime(&c, T->ev[3], T->ev[4]);
		T->pending_rep.ms_h2d = a; T->pending_rep.ms_parse = b + f; T->pending_rep.ms_map = c;
		float st7[7];
		if (T->ev_sides == 1 && urmapx_ctx_stage_ms(T->C, st7) == URMAPX_OK) { T->pending_rep.ms_map_search = st7[0]; T->pending_rep.ms_map_dp = st7[1] + st7[2]; }
	}
	T->pending_rep.sam_bytes = h2.sam_total;
	T->pending_rep.mapped_q = h2.cnt[0]; T->pending_rep.mapped_lowq = h2.cnt[1]; T->pending_rep.unmapped = h2.cnt[2]; T->pending_rep.unsupported = h2.cnt[3];
	return fetch_sam(T, sam, sam_cap, rep);
}

}  // namespace

extern "C" {

int urmapx_text_create(urmapx_ctx *C, urmapx_text **out) {
	if (!C || !out) return URMAPX_E_ARG;
	*out = nullptr;
	HIP_TRY(hipSetDevice(ctx_device(C)));
	urmapx_text *T = new urmapx_text;
	T->C = C;
	const urmapx_index *I = ctx_index(C);
	const uint32_t n = urmapx_index_seq_count(I);
	std::string names;
	std::vector<uint32_t> offs(n + 1, 0);
	for (uint32_t i = 0; i < n; ++i) { names += urmapx_index_label(I, i); offs[i + 1] = (uint32_t)names.size(); }
	int rc = T->tnames.ensure(names.size() + 1);
	if (!rc) rc = T->tname_offs.ensure(n + 1);
	if (!rc) rc = T->comp.ensure(256);
	if (!rc) rc = T->hdr.ensure(2);
	if (!rc) rc = T->used.ensure(1);
	hipError_t e = hipSuccess;
	if (!rc && !names.empty()) e = hipMemcpy(T->tnames.p, names.data(), names.size(), hipMemcpyHostToDevice);
	if (!rc && e == hipSuccess) e = hipMemcpy(T->tname_offs.p, offs.data(), (n + 1) * 4, hipMemcpyHostToDevice);
	if (!rc && e == hipSuccess) e = hipMemcpy(T->comp.p, complement_table(), 256, hipMemcpyHostToDevice);
	if (!rc && e == hipSuccess) e = hipHostMalloc((void **)&T->h_hdr, 4 * sizeof(TextHdr), hipHostMallocDefault);
	if (!rc && e == hipSuccess) { memset(T->h_hdr, 0, 4 * sizeof(TextHdr)); e = hipStreamCreateWithFlags(&T->copy_st, hipStreamNonBlocking); }
	if (!rc && e != hipSuccess) rc = hip_rc(e);
	if (rc) { urmapx_text_destroy(T); return rc; }
	T->ev_ok = true;
	for (hipEvent_t &x : T->ev)
		if (hipEventCreate(&x) != hipSuccess) { T->ev_ok = false; x = nullptr; }
	T->tail_ok = true;
	for (auto &set : T->tail_ev)
		for (hipEvent_t &x : set)
			if (hipEventCreate(&x) != hipSuccess) { T->tail_ok = false; x = nullptr; }
	T->seq_count = n;
	*out = T;
	return URMAPX_OK;
}

void urmapx_text_destroy(urmapx_text *T) {
	if (!T) return;
	(void)hipSetDevice(ctx_device(T->C));
	(void)hipStreamSynchronize(ctx_stream(T->C));
	if (T->copy_st) { (void)hipStreamSynchronize(T->copy_st); (void)hipStreamDestroy(T->copy_st); }
	for (auto &set : T->tail_ev)
		for (hipEvent_t x : set)
			if (x) (void)hipEventDestroy(x);
	for (int k = 0; k < 2; ++k) { T->raw[k].release(); T->tile_counts[k].release(); T->ends[k].release(); T->blen[k].release(); }
	T->bases.release(); T->sam.release(); T->comp.release();
	T->sums.release(); T->lens.release(); T->qn.release(); T->rec_offs.release();
	T->used.release(); T->tname_offs.release(); T->offs.release(); T->tnames.release(); T->results.release(); T->pathops.release();
	T->hdr.release();
	if (T->h_hdr) (void)hipHostFree(T->h_hdr);
	for (hipEvent_t x : T->ev)
		if (x) (void)hipEventDestroy(x);
	delete T;
}

int urmapx_text_map_se(urmapx_text *T, const char *fastq, size_t nbytes, unsigned minq, char *sam, size_t sam_cap, urmapx_text_report *rep) {
	if (!T || !rep || (nbytes && !fastq)) return URMAPX_E_ARG;
	return map_text(T, fastq, nbytes, nullptr, 0, minq, sam, sam_cap, rep);
}

int urmapx_text_map_pe(urmapx_text *T, const char *fastq1, size_t nbytes1, const char *fastq2, size_t nbytes2, unsigned minq, char *sam,
                       size_t sam_cap, urmapx_text_report *rep) {
	if (!T || !rep || !fastq1 || !fastq2) return URMAPX_E_ARG;
	return map_text(T, fastq1, nbytes1, fastq2, nbytes2, minq, sam, sam_cap, rep);
}

int urmapx_text_set_deferred(urmapx_text *T, int on) {
	if (!T || T->waiting || T->pending) return URMAPX_E_ARG;
	if (on && !T->tail_ok) return URMAPX_E_NODEVICE;  // the second stream is ordered behind the first by events: none, no deferred copies (the caller keeps the plain mode)
	T->deferred = on != 0;
	return URMAPX_OK;
}

int urmapx_text_wait(urmapx_text *T, urmapx_text_report *rep) {
	if (!T || !rep || T->waiting <= 0) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(ctx_device(T->C)));
	const int set = (int)((T->chunk_no - (uint32_t)T->waiting) & 1u);  // the oldest tail on its way
	--T->waiting;
	return finish_tail(T, set, rep);
}

int urmapx_text_fetch_sam(urmapx_text *T, char *sam, size_t sam_cap, urmapx_text_report *rep) {
	if (!T || !rep || !T->pending) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(ctx_device(T->C)));
	return fetch_sam(T, sam, sam_cap, rep);
}

int urmapx_text_fetch_pairs(urmapx_text *T, uint32_t npairs, urmapx_result *results, urmapx_pair_info *info, uint32_t *line_ends1, uint32_t *lens2) {
	if (!T || !results || !info || !line_ends1 || !lens2) return URMAPX_E_ARG;
	if (T->last_pairs == 0 || T->last_pairs != npairs) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(ctx_device(T->C)));
	hipStream_t st = ctx_stream(T->C);
	HIP_TRY(hipMemcpyAsync(results, T->results.p, (size_t)2 * npairs * sizeof(urmapx_result), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(line_ends1, T->ends[0].p, (size_t)4 * npairs * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(lens2, T->blen[1].p, (size_t)npairs * 4, hipMemcpyDeviceToHost, st));
	return urmapx_ctx_get_pair_info(T->C, info, npairs);  // synchronises the stream
}

}  // extern "C"
