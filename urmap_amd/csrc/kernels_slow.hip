// kernels_slow.hip -- the general single-end search: State1::Search_Lo (search1m6.cpp:35-277) for the reads the fast
// kernels of kernels.hip do not take.
//
// The fast path keeps a read's state in registers and LDS and therefore has a domain: reads up to 1024 bases, 512 live
// hits, 8192 HSPs, alignment paths of 96 runs.  The reference has none of these limits (its lists grow, state1.cpp:193-228;
// its scratch holds a read of ~30 kb, state1.h:113).  Whatever falls outside the fast path's domain -- flagged per read
// by it, never mis-mapped -- is mapped again here with every list in this block's global scratch: hits and HSPs by the
// tens of thousands, reads up to URMAPX_MAX_QL_SLOW bases, paths as long as the read.  One wavefront per read, the
// reference's schedule candidate by candidate (no batching: this kernel is for the rare read, its only job is to be the
// reference); the 64 lanes share the byte compares of one ExtendPen, the hit / HSP list scans, the chain walks of 64
// k-mers and the banded DP's diagonals (viterbi_dev.h, trace cells in global scratch).
#include "slow_dev.h"

namespace urx {

// the reads a batch's fast passes left flagged and that this kernel can take: list[0] = count, list[1..] = read numbers
__global__ __launch_bounds__(256) void collect_flagged_kernel(const urmapx_result *__restrict__ results, const uint64_t *__restrict__ offs,
                                                              uint32_t n, uint32_t W, uint32_t max_len, uint32_t *list) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (results[i].status == 0) return;
	const uint64_t ql = offs[i + 1] - offs[i];
	if (ql < W || ql > max_len) return;  // shorter than a word: the reference underflows; longer than this build's cap: stays flagged
	list[1 + atomicAdd(list, 1u)] = i;
}

__global__ __launch_bounds__(64) void search_se_slow_kernel(DevIndex X, urmapx_params P, const uint8_t *__restrict__ bases,
                                                            const uint64_t *__restrict__ offs, const uint32_t *__restrict__ list,
                                                            urmapx_result *__restrict__ results, urmapx_path_op *__restrict__ path_ops,
                                                            uint32_t *path_used, uint32_t path_cap, uint8_t *scratch, size_t stride,
                                                            uint32_t qcap, const uint8_t *__restrict__ g_seq,
                                                            const uint8_t *__restrict__ g_blob, uint32_t *ticket) {
	const int lane = threadIdx.x;
	const uint32_t n = list[0];
	if (n == 0) return;
	const SlowLayout L = slow_layout(qcap);
	uint8_t *sc = scratch + (size_t)blockIdx.x * stride;
	SlowWave S(X, P, lane);
	S.gseq = g_seq; S.gblob = g_blob;
	S.W = (int)X.W;
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
	for (;;) {
		const uint32_t idx = uni(atomicAdd(ticket, lane == 0 ? 1u : 0u));  // every lane takes part (see search_se_kernel)
		if (idx >= n) break;
		const uint32_t r = list[1 + idx];
		const uint64_t off = offs[r];
		const int QL = (int)(offs[r + 1] - off);
		urmapx_result res;
		res.dbpos = 0xFFFFFFFFu; res.seq_index = 0xFFFFFFFFu; res.coord = 0xFFFFFFFFu;
		res.score = 0; res.second = 0; res.mapq = 0; res.plus = 0; res.exit_phase = 0; res.status = 0;
		res.hit_count = 0; res.path_nops = 0; res.path_off = 0;
		if (QL < S.W || (uint32_t)QL > qcap || S.W > 32 || X.maxIx > (uint32_t)SLOW_ROW_CAP) {
			res.status = URMAPX_ST_BAD_LENGTH;
			if (lane == 0) results[r] = res;
			continue;
		}
		S.QL = QL; S.nwords = QL - (S.W - 1);
		S.hitCount = 0; S.hspCount = 0;
		S.maxPen = P.max_penalty; S.best = 0; S.second = 0; S.bestHSP = 0;
		S.haveTop = false; S.top_db = 0; S.top_plus = false; S.top_nops = 0; S.status = 0;
		__syncthreads();
		for (int p = lane; p < QL; p += 64) {
			const uint8_t ch = bases[off + p];
			S.q[0][p] = ch;
			S.q[1][QL - 1 - p] = (uint8_t)comp_char(ch);
		}
		__syncthreads();
		const int phase = S.search_lo();
		res.mapq = (uint8_t)S.calc_mapq();
		res.score = (int16_t)S.best; res.second = (int16_t)S.second;
		res.hit_count = (uint16_t)(S.hitCount > 0xFFFF ? 0xFFFF : S.hitCount); res.exit_phase = (uint8_t)phase; res.status = (uint8_t)S.status;
		if (S.haveTop) {  // SetMappedPos (state1.cpp:129-145) with PosToCoordL (ufindex.cpp:729-755)
			uint32_t lo = 0, hi = X.seqCount - 1;
			uint32_t found = 0xFFFFFFFFu, coord = 0xFFFFFFFFu, tl = 0;
			while (lo <= hi && hi != 0xFFFFFFFFu) {
				const uint32_t k = (lo + hi) / 2;
				const uint32_t o = X.seqOffsets[k], sl = X.seqLengths[k];
				if (S.top_db >= o && S.top_db < o + sl) { found = k; coord = S.top_db - o; tl = sl; break; }
				if (S.top_db > o) lo = k + 1;
				else hi = k - 1;
			}
			if (found != 0xFFFFFFFFu && coord + (uint32_t)QL <= tl) {
				res.dbpos = S.top_db; res.seq_index = found; res.coord = coord; res.plus = S.top_plus ? 1 : 0;
				if (S.top_nops > 0) {
					uint32_t po = 0;
					// room is taken only if the path fits (a count past the arena's end made the host reject the whole batch)
					if (lane == 0) po = reserve_path(path_used, (uint32_t)S.top_nops, path_cap);
					po = uni(po);
					if (po != 0xFFFFFFFFu && S.top_nops <= 0xFFFF) {
						for (int t = lane; t < S.top_nops; t += 64) path_ops[po + t] = S.top[t];
						res.path_off = po; res.path_nops = (uint16_t)S.top_nops;
					} else
						res.status |= URMAPX_ST_PATH_OVERFLOW;  // the batch's path arena is full
				}
			}
		}
		if (lane == 0) results[r] = res;
	}
}

size_t slow_scratch_stride(uint32_t qcap) { return slow_layout(qcap).total; }

hipError_t launch_search_se_slow(const DevIndex &X, const urmapx_params &P, const uint8_t *d_bases, const uint64_t *d_offs, uint32_t n,
                                 uint32_t qcap, urmapx_result *d_results, urmapx_path_op *d_path_ops, uint32_t *d_path_used,
                                 uint32_t path_cap, uint8_t *scratch, int blocks, uint32_t *list, uint32_t *ticket, hipStream_t s) {
	if (n == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(list, 0, 4, s);
	if (e == hipSuccess) e = hipMemsetAsync(ticket, 0, 4, s);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(collect_flagged_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_results, d_offs, n, X.W, qcap, list);
	hipLaunchKernelGGL(search_se_slow_kernel, dim3((unsigned)blocks), dim3(64), 0, s, X, P, d_bases, d_offs, list, d_results, d_path_ops,
	                   d_path_used, path_cap, scratch, slow_scratch_stride(qcap), qcap, X.seq, X.blob, ticket);
	return hipGetLastError();
}

}  // namespace urx
