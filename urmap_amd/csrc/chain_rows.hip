// chain_rows.hip -- UFIndex::GetRow_Blob's rows (ufindex.cpp:883-943) laid out once per index, on the device, when the slot
// table reaches HBM (round 4).
//
// A k-mer's collision chain is a linked list through the slot table: GetRow_Blob follows up to MaxIx - 1 dependent links, each
// a random 5-byte read of a 27 GB table.  One of a read's 254 k-mers nearly always heads a long chain, so the search kernel's
// chain walk is a dependent sequence of a dozen memory round trips per read and 17 % of its vector instructions (the 64-bit
// slot arithmetic of every hop).  The table is read-only: every head's row can be written down once.
//   rowinfo[slot]   2 x u32 (round 5): .x = row length (low byte) | offset of the row inside its group of 1024 slots << 8; 0 where the
//                   slot heads no chain (not "mine", or a single-entry PLUS1 / BOTH1 slot, whose row is its own position);
//                   .y = the row's SECOND position: with the head's own position (which the probe has already read with the
//                   slot) a row of two -- 70 % of the rows of an hg38-like table (0.41 G rows hold 0.99 G positions) -- needs
//                   no read of `rows` at all, and phase 4 of the search (rows of length <= 2) none whatever the read
//   rowbase[group]  u64: where the rows of the group's heads begin in `rows`
//   rows            u32: the positions, row after row, in slot order
// A kernel then needs the head's info word (one random read), the group base (a 40 MB array that lives in L2 / MALL) and the
// row itself, contiguous: two dependent round trips whatever the chain's length.  47 GB at hg38 scale (43.1 info + 4.0 rows),
// built in under a second; an index whose rows do not fit the device keeps the hop-by-hop walk (DevIndex::rowinfo == nullptr).
#include "kernels.h"

#include "dev_common.h"

#include <vector>

namespace urx {

static constexpr int CR_GROUP = 1024;
static constexpr int CR_ROW_CAP = 32;  // UFIndex m_MaxIx of every index this build accepts

// GetRow_Blob from a head slot whose tally says "mine, with a next link" -- the walk of SearchWave::walk_run, one thread per
// chain.  out != nullptr: the positions are written; returns the row length.
__device__ __forceinline__ int chain_row(const uint8_t *__restrict__ blob, uint64_t N, int maxIx, uint64_t slot, uint32_t T, uint32_t pos,
                                         uint32_t *out) {
	int K = 0;
	for (;;) {
		if (out) out[K] = pos;
		++K;
		if (K == maxIx || K >= CR_ROW_CAP) return K;
		if (T == TALLY_PLUS1 || T == TALLY_BOTH1) return 1;
		if (T == TALLY_END) return K;
		if (T == TALLY_LONG_MINE || T == TALLY_LONG_OTHER) {
			const uint64_t slotA = addmod(slot, pos & 0xFFFFu, N);
			slot = addmod(slotA, pos >> 16, N);
			uint32_t tA, pA;
			load_slot(blob, slotA, tA, pA);
			if (out) out[K - 1] = pA;
		} else
			slot = addmod(slot, T & TALLY_NEXT_MASK, N);
		load_slot(blob, slot, T, pos);
	}
}

__device__ __forceinline__ bool heads_a_row(uint32_t T) { return (T & TALLY_MY_BIT) != 0 && T != TALLY_BOTH1 && T != TALLY_PLUS1; }

// pass 1: row length of every slot, and their sum per group of 1024 slots
__global__ __launch_bounds__(CR_GROUP) void rows_len_kernel(const uint8_t *__restrict__ blob, uint64_t N, int maxIx, uint2 *__restrict__ info,
                                                            uint32_t *__restrict__ groupsum, uint32_t groups) {
	__shared__ uint32_t wsum[CR_GROUP / 64];
	// (a launch holds fewer than 2^32 work-items and the table has more slots than that: the blocks loop over the groups)
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t s = (uint64_t)g * CR_GROUP + threadIdx.x;
		uint32_t len = 0;
		if (s < N) {
			uint32_t T, pos;
			load_slot(blob, s, T, pos);
			if (heads_a_row(T)) len = (uint32_t)chain_row(blob, N, maxIx, s, T, pos, nullptr);
			info[s] = make_uint2(len, 0u);
		}
		uint32_t v = len;
		for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t t = 0;
			for (int w = 0; w < CR_GROUP / 64; ++w) t += wsum[w];
			groupsum[g] = t;
		}
	}
}

// pass 2: the info words and the rows
__global__ __launch_bounds__(CR_GROUP) void rows_fill_kernel(const uint8_t *__restrict__ blob, uint64_t N, int maxIx, uint2 *__restrict__ info,
                                                             const uint64_t *__restrict__ rowbase, uint32_t *__restrict__ rows, uint32_t groups) {
	__shared__ uint32_t wsum[CR_GROUP / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t s = (uint64_t)g * CR_GROUP + threadIdx.x;
		const uint32_t len = s < N ? info[s].x : 0u;
		uint32_t inc = len;  // inclusive prefix inside the wavefront
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = __shfl_up(inc, d, 64);
			if (lane >= d) inc += up;
		}
		__syncthreads();
		if (lane == 63) wsum[w] = inc;
		__syncthreads();
		uint32_t before = 0;
		for (int k = 0; k < w; ++k) before += wsum[k];
		const uint32_t off = before + inc - len;
		if (s < N && len) {
			uint32_t T, pos;
			load_slot(blob, s, T, pos);
			uint32_t *const row = rows + rowbase[g] + off;
			(void)chain_row(blob, N, maxIx, s, T, pos, row);
			info[s] = make_uint2((off << 8) | len, len > 1 ? row[1] : 0u);  // (this thread wrote the row: its own stores are visible to it)
		}
	}
}

// device arrays of the layout; all three null on return if maxIx is beyond the kernels' row capacity or memory is short
hipError_t build_chain_rows(const uint8_t *d_blob, uint64_t slot_count, uint32_t max_ix, uint2 **d_info, uint64_t **d_base, uint32_t **d_rows,
                            uint64_t *total_rows) {
	*d_info = nullptr; *d_base = nullptr; *d_rows = nullptr; *total_rows = 0;
	if (max_ix > (uint32_t)CR_ROW_CAP || max_ix < 1 || slot_count == 0) return hipSuccess;
	const uint64_t groups = (slot_count + CR_GROUP - 1) / CR_GROUP;
	if (groups > 0x7FFFFFFFull) return hipSuccess;
	uint2 *info = nullptr;
	uint32_t *gsum = nullptr, *rows = nullptr;
	uint64_t *base = nullptr;
	auto drop = [&]() { (void)hipFree(info); (void)hipFree(gsum); (void)hipFree(base); (void)hipFree(rows); (void)hipGetLastError(); };
	if (hipMalloc((void **)&info, slot_count * sizeof(uint2)) != hipSuccess || hipMalloc((void **)&gsum, groups * 4) != hipSuccess ||
	    hipMalloc((void **)&base, groups * 8) != hipSuccess) { drop(); return hipSuccess; }  // no room: the kernels walk hop by hop
	const unsigned grid = (unsigned)(groups < (1u << 20) ? groups : (1u << 20));
	hipLaunchKernelGGL(rows_len_kernel, dim3(grid), dim3(CR_GROUP), 0, nullptr, d_blob, slot_count, (int)max_ix, info, gsum, (uint32_t)groups);
	hipError_t e = hipGetLastError();
	std::vector<uint32_t> hs(groups);
	if (e == hipSuccess) e = hipMemcpy(hs.data(), gsum, groups * 4, hipMemcpyDeviceToHost);
	if (e != hipSuccess) { drop(); return e; }
	std::vector<uint64_t> hb(groups);
	uint64_t total = 0;
	for (uint64_t g = 0; g < groups; ++g) { hb[g] = total; total += hs[g]; }
	(void)hipFree(gsum); gsum = nullptr;
	// a row's place is kept as a 32-bit index into `rows` with 0xFFFFFFFF as the "single entry" mark (SearchWave::rows_fetch); no
	// index the reference can build gets there (positions are 32 bit, each stored once), but nothing else enforced it
	if (total >= 0xFFFFFFFFull) { drop(); return hipSuccess; }
	if (hipMalloc((void **)&rows, (total + 64) * 4) != hipSuccess) { drop(); return hipSuccess; }
	e = hipMemcpy(base, hb.data(), groups * 8, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(rows_fill_kernel, dim3(grid), dim3(CR_GROUP), 0, nullptr, d_blob, slot_count, (int)max_ix, info, base, rows, (uint32_t)groups);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (e != hipSuccess) { drop(); return e; }
	*d_info = info; *d_base = base; *d_rows = rows; *total_rows = total;
	return hipSuccess;
}


// ---- the slot table with its rows' heads inline (DevIndex::slot16) ----
// One thread per slot, from the 5-byte table and the row layout above.
__global__ __launch_bounds__(256) void slot16_kernel(const uint8_t *__restrict__ blob, uint64_t N, const uint2 *__restrict__ info,
                                                     const uint64_t *__restrict__ rowbase, const uint32_t *__restrict__ rows, uint4 *__restrict__ out,
                                                     uint32_t groups) {
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {  // (more slots than a launch has work-items)
		const uint64_t s = (uint64_t)g * 256 + threadIdx.x;
		if (s >= N) continue;
		uint32_t T, pos;
		load_slot(blob, s, T, pos);
		uint32_t len = 0, x = 0;
		if ((T & TALLY_MY_BIT) != 0) {
			if (T == TALLY_BOTH1 || T == TALLY_PLUS1) len = 1;
			else {
				const uint2 i = info[s];
				len = i.x & 0xFFu;
				const uint64_t at = rowbase[s >> 10] + (i.x >> 8);
				pos = rows[at];  // the row's first position: the slot's own, except behind a long link (its middle slot's)
				x = len <= 2 ? i.y : (uint32_t)at;
			}
		}
		out[s] = make_uint4(pos, T | (len << 8), x, 0u);
	}
}

// null on return if there is no room for it (16 bytes per slot): the kernels then use the row layout, or walk
hipError_t build_slot16(const uint8_t *d_blob, uint64_t slot_count, const uint2 *d_info, const uint64_t *d_base, const uint32_t *d_rows, uint4 **d_slot16) {
	*d_slot16 = nullptr;
	if (!d_info || !d_base || !d_rows || slot_count == 0) return hipSuccess;
	uint4 *out = nullptr;
	if (hipMalloc((void **)&out, (slot_count + 1) * sizeof(uint4)) != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
	const uint64_t groups = (slot_count + 255) / 256;
	hipLaunchKernelGGL(slot16_kernel, dim3((unsigned)(groups < (1u << 20) ? groups : (1u << 20))), dim3(256), 0, nullptr, d_blob, slot_count, d_info, d_base, d_rows,
	                   out, (uint32_t)groups);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (e != hipSuccess) { (void)hipFree(out); return e; }
	*d_slot16 = out;
	return hipSuccess;
}

// ---- slot16 and the rows WITHOUT the per-slot info entries in between (round 6) ----
// build_chain_rows + build_slot16 hold 8 bytes of info per slot (43 GB at hg38 scale) beside the table, the rows and slot16 until slot16 is
// written: 160 GB at the peak of an upload for 122 GB of resident index.  Here slot16 is its own scratch: pass 1 writes every slot's own
// position, tally and row length into it (and the groups' row totals), pass 2 takes the lengths from there, writes each head's row and sets
// .x (the row's first position) and .z (second position, or the row's place in `rows`).  Same bytes in slot16 and rows as the two-step build.
__global__ __launch_bounds__(CR_GROUP) void rows_len16_kernel(const uint8_t *__restrict__ blob, uint64_t N, int maxIx, uint4 *__restrict__ out,
                                                              uint32_t *__restrict__ groupsum, uint32_t groups) {
	__shared__ uint32_t wsum[CR_GROUP / 64];
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t s = (uint64_t)g * CR_GROUP + threadIdx.x;
		uint32_t len = 0;
		if (s < N) {
			uint32_t T, pos;
			load_slot(blob, s, T, pos);
			if (heads_a_row(T)) len = (uint32_t)chain_row(blob, N, maxIx, s, T, pos, nullptr);
			const uint32_t l16 = heads_a_row(T) ? len : ((T & TALLY_MY_BIT) != 0 ? 1u : 0u);  // BOTH1 / PLUS1: a row of one, its own position
			out[s] = make_uint4(pos, T | (l16 << 8), 0u, 0u);
		}
		uint32_t v = len;
		for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t t = 0;
			for (int w = 0; w < CR_GROUP / 64; ++w) t += wsum[w];
			groupsum[g] = t;
		}
	}
}
__global__ __launch_bounds__(CR_GROUP) void rows_fill16_kernel(const uint8_t *__restrict__ blob, uint64_t N, int maxIx, uint4 *__restrict__ out,
                                                               const uint64_t *__restrict__ rowbase, uint32_t *__restrict__ rows, uint32_t groups) {
	__shared__ uint32_t wsum[CR_GROUP / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t s = (uint64_t)g * CR_GROUP + threadIdx.x;
		uint32_t T = 0, len = 0;
		if (s < N) {
			const uint32_t y = out[s].y;
			T = y & 0xFFu;
			len = heads_a_row(T) ? (y >> 8) & 0xFFu : 0u;
		}
		uint32_t inc = len;  // inclusive prefix inside the wavefront
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = __shfl_up(inc, d, 64);
			if (lane >= d) inc += up;
		}
		__syncthreads();
		if (lane == 63) wsum[w] = inc;
		__syncthreads();
		uint32_t before = 0;
		for (int k = 0; k < w; ++k) before += wsum[k];
		if (s < N && len) {
			const uint64_t at = rowbase[g] + (before + inc - len);
			uint32_t T2, pos;
			load_slot(blob, s, T2, pos);
			uint32_t *const row = rows + at;
			(void)chain_row(blob, N, maxIx, s, T2, pos, row);
			out[s].x = row[0];  // the slot's own position, except behind a long link (its middle slot's)
			out[s].z = len <= 2 ? (len > 1 ? row[1] : 0u) : (uint32_t)at;
		}
	}
}
// null / null on return if there is no room (the caller falls back to the two-step build, then to the row layout, then to the walk)
hipError_t build_slot16_direct(const uint8_t *d_blob, uint64_t slot_count, uint32_t max_ix, uint4 **d_slot16, uint32_t **d_rows, uint64_t *total_rows) {
	*d_slot16 = nullptr; *d_rows = nullptr; *total_rows = 0;
	if (max_ix > (uint32_t)CR_ROW_CAP || max_ix < 1 || slot_count == 0) return hipSuccess;
	const uint64_t groups = (slot_count + CR_GROUP - 1) / CR_GROUP;
	if (groups > 0x7FFFFFFFull) return hipSuccess;
	uint4 *out = nullptr;
	uint32_t *gsum = nullptr, *rows = nullptr;
	uint64_t *base = nullptr;
	auto drop = [&]() { (void)hipFree(out); (void)hipFree(gsum); (void)hipFree(base); (void)hipFree(rows); (void)hipGetLastError(); };
	if (hipMalloc((void **)&out, (slot_count + 1) * sizeof(uint4)) != hipSuccess || hipMalloc((void **)&gsum, groups * 4) != hipSuccess ||
	    hipMalloc((void **)&base, groups * 8) != hipSuccess) { drop(); return hipSuccess; }
	const unsigned grid = (unsigned)(groups < (1u << 20) ? groups : (1u << 20));
	hipLaunchKernelGGL(rows_len16_kernel, dim3(grid), dim3(CR_GROUP), 0, nullptr, d_blob, slot_count, (int)max_ix, out, gsum, (uint32_t)groups);
	hipError_t e = hipGetLastError();
	std::vector<uint32_t> hs(groups);
	if (e == hipSuccess) e = hipMemcpy(hs.data(), gsum, groups * 4, hipMemcpyDeviceToHost);
	if (e != hipSuccess) { drop(); return e; }
	std::vector<uint64_t> hb(groups);
	uint64_t total = 0;
	for (uint64_t g = 0; g < groups; ++g) { hb[g] = total; total += hs[g]; }
	(void)hipFree(gsum); gsum = nullptr;
	if (total >= 0xFFFFFFFFull) { drop(); return hipSuccess; }
	if (hipMalloc((void **)&rows, (total + 64) * 4) != hipSuccess) { drop(); return hipSuccess; }
	e = hipMemcpy(base, hb.data(), groups * 8, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemset(out + slot_count, 0, sizeof(uint4));
	if (e == hipSuccess) {
		hipLaunchKernelGGL(rows_fill16_kernel, dim3(grid), dim3(CR_GROUP), 0, nullptr, d_blob, slot_count, (int)max_ix, out, base, rows, (uint32_t)groups);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipDeviceSynchronize();
	(void)hipFree(base); base = nullptr;
	if (e != hipSuccess) { drop(); return e; }
	*d_slot16 = out; *d_rows = rows; *total_rows = total;
	return hipSuccess;
}

// ---- UFIndex::Validate (ufindex.cpp:611-658) over a resident table ----
// ValidateSlot for every slot, one thread per slot: a slot whose tally says "mine" heads a row; GetRow_Validate
// (ufindex.cpp:834-881) collects the row's positions link by link (the head must be "mine", every later link "other", a long
// link's middle slot TALLY_NEXT_LONG_OTHER, at most MaxIx entries), every position must lie inside the sequence store and the
// word that starts there (GetWord, ufindex.cpp:68-81) must hash back to the head slot (WordToSlot, ufindex.h:60-65).  Beyond
// the reference's test: the slots the walks pass through are counted, and every slot that is not free must lie on exactly one
// head's chain -- `reached` == `used` (a table with a lost or doubly linked slot fails that, whatever its positions hash to).
struct ValidateCounters {  // u64 each; mirrors urmapx_validate_report from `heads` on
	unsigned long long heads, positions, used, reached, bad_hash, bad_pos, bad_link, bad_len, first_bad_slot;
};
static constexpr int VC_WORDS = 9;

__device__ __forceinline__ uint64_t word_at(const uint8_t *__restrict__ seq, uint32_t pos, uint32_t W) {  // GetWord
	uint64_t w = 0;
	for (uint32_t i = 0; i < W; ++i) {
		const uint32_t L = letter_of(seq[(size_t)pos + i]);
		if (L > 3u) return ~0ull;
		w = (w << 2) | L;
	}
	return w;
}

__global__ __launch_bounds__(256) void validate_kernel(DevIndex X, unsigned long long *__restrict__ out, uint32_t groups) {
	__shared__ unsigned long long acc[VC_WORDS - 1];
	if (threadIdx.x < VC_WORDS - 1) acc[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long heads = 0, positions = 0, used = 0, reached = 0, bad_hash = 0, bad_pos = 0, bad_link = 0, bad_len = 0;
	unsigned long long first_bad = ~0ull;
	const uint64_t N = X.slotCount;
	for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
		const uint64_t head = (uint64_t)g * 256 + threadIdx.x;
		if (head >= N) continue;
		uint32_t T, pos;
		load_slot(X.blob, head, T, pos);
		if (T != TALLY_FREE) ++used;
		if ((T & TALLY_MY_BIT) == 0) continue;  // TallyOther: GetRow_Validate returns 0
		++heads;
		uint64_t slot = head;
		uint32_t K = 0;
		bool bad = false;
		for (;;) {
			++reached;
			uint32_t p = pos;
			const bool single = T == TALLY_PLUS1 || T == TALLY_BOTH1;
			const bool is_long = T == TALLY_LONG_MINE || T == TALLY_LONG_OTHER;
			++K;
			uint64_t next = slot;
			bool stop = single || K == X.maxIx;
			if (single && slot != head) { ++bad_link; bad = true; }
			if (!stop) {
				if ((slot == head) != ((T & TALLY_MY_BIT) != 0)) { ++bad_link; bad = true; }
				if (T == TALLY_END) stop = true;
				else if (is_long) {
					const uint64_t slotA = addmod(slot, pos & 0xFFFFu, N);
					next = addmod(slotA, pos >> 16, N);
					uint32_t tA, pA;
					load_slot(X.blob, slotA, tA, pA);
					p = pA;
					++reached;
					if (tA != TALLY_LONG_OTHER) { ++bad_link; bad = true; }
				} else {
					const uint32_t step = T & TALLY_NEXT_MASK;
					if (step == 0 || step > 124u) { ++bad_link; bad = true; stop = true; }  // TALLY_FREE in a chain
					next = addmod(slot, step, N);
				}
			}  // (K == MaxIx on a long link: GetRow_Validate returns before it resolves the link, PosVec[K-1] stays the step word)
			++positions;
			if (p >= X.seqDataSize) { ++bad_pos; bad = true; }
			else if (mod_slots(murmur64(word_at(X.seq, p, X.W)), N, X.slotMagic) != head) { ++bad_hash; bad = true; }
			if (stop) break;
			if (K >= 256u) { ++bad_len; bad = true; break; }  // only with a header whose MaxIx is 0 or beyond what a tally can count
			slot = next;
			load_slot(X.blob, slot, T, pos);
		}
		if (K > X.maxIx) { ++bad_len; bad = true; }
		if (bad && head < first_bad) first_bad = head;
	}
	atomicAdd(&acc[0], heads); atomicAdd(&acc[1], positions); atomicAdd(&acc[2], used); atomicAdd(&acc[3], reached);
	atomicAdd(&acc[4], bad_hash); atomicAdd(&acc[5], bad_pos); atomicAdd(&acc[6], bad_link); atomicAdd(&acc[7], bad_len);
	if (first_bad != ~0ull) atomicMin(out + 8, first_bad);
	__syncthreads();
	if (threadIdx.x < VC_WORDS - 1 && acc[threadIdx.x]) atomicAdd(out + threadIdx.x, acc[threadIdx.x]);
}

// out[9]: heads, positions, used, reached, bad_hash, bad_pos, bad_link, bad_len, first_bad_slot (all ones: none)
hipError_t validate_index(const DevIndex &X, uint64_t out[9], float *ms) {
	unsigned long long *d = nullptr;
	hipError_t e = hipMalloc((void **)&d, VC_WORDS * 8);
	if (e != hipSuccess) return e;
	unsigned long long init[VC_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0, ~0ull};
	e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
	const uint64_t groups = (X.slotCount + 255) / 256;
	hipEvent_t a = nullptr, b = nullptr;
	if (e == hipSuccess) e = hipEventCreate(&a);
	if (e == hipSuccess) e = hipEventCreate(&b);
	if (e == hipSuccess && groups <= 0xFFFFFFFFull) {
		(void)hipEventRecord(a, nullptr);
		// (more slots than a launch has work-items: the blocks loop over the groups)
		hipLaunchKernelGGL(validate_kernel, dim3((unsigned)(groups < 65536 ? groups : 65536)), dim3(256), 0, nullptr, X, d, (uint32_t)groups);
		e = hipGetLastError();
		(void)hipEventRecord(b, nullptr);
	} else if (e == hipSuccess)
		e = hipErrorInvalidValue;
	if (e == hipSuccess) e = hipMemcpy(init, d, sizeof init, hipMemcpyDeviceToHost);
	if (e == hipSuccess && ms) (void)hipEventElapsedTime(ms, a, b);
	for (int i = 0; i < VC_WORDS; ++i) out[i] = init[i];
	if (a) (void)hipEventDestroy(a);
	if (b) (void)hipEventDestroy(b);
	(void)hipFree(d);
	return e;
}

// Checksum of a resident byte array (urmapx_checksum_device, urmapx_index_checksum): sum over the array's little-endian 64-bit
// words w_i (the last one zero-padded) of murmur64(w_i + (i + 1) * 0x9E3779B97F4A7C15), modulo 2^64.  A sum: the order in which the
// threads add their parts does not matter, so one array has one value on every device and in a numpy restatement.
__global__ void __launch_bounds__(256) checksum_kernel(const uint8_t *__restrict__ p, uint64_t nbytes, unsigned long long *out) {
	const uint64_t words = nbytes >> 3;
	const uint64_t *w = (const uint64_t *)p;
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
		acc += murmur64(w[i] + (i + 1) * 0x9E3779B97F4A7C15ull);
	if (blockIdx.x == 0 && threadIdx.x == 0 && (nbytes & 7)) {
		uint64_t t = 0;
		for (uint64_t k = 0; k < (nbytes & 7); ++k) t |= (uint64_t)p[(words << 3) + k] << (8 * k);
		acc += murmur64(t + (words + 1) * 0x9E3779B97F4A7C15ull);
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}

hipError_t checksum_device(const void *d_ptr, uint64_t nbytes, uint64_t *out) {
	if (((uintptr_t)d_ptr & 7) != 0) return hipErrorInvalidValue;
	unsigned long long *d = nullptr;
	hipError_t e = hipMalloc((void **)&d, 8);
	if (e != hipSuccess) return e;
	e = hipMemset(d, 0, 8);
	if (e == hipSuccess) {
		const uint64_t blocks = ((nbytes >> 3) + 255) / 256;
		hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks < 16384 ? blocks : 16384)), dim3(256), 0, nullptr, (const uint8_t *)d_ptr, nbytes, d);
		e = hipGetLastError();
	}
	unsigned long long v = 0;
	if (e == hipSuccess) e = hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost);
	*out = v;
	(void)hipFree(d);
	return e;
}

}  // namespace urx
