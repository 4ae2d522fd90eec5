// make_ufi_gpu.hip -- the data-parallel passes of index construction on the GPU ("counting passes are
// GPU-friendly; the chain-building pass is order dependent -- keep on CPU").
//
// UFIndex::MakeIndex (ufindex.cpp:83-151) = CountSlots + CountSlots_Minus (ufindex.cpp:338-408: per-slot counts of the
// plus-strand words and of their reverse complements, saturating at 255) and then UpdateSlot for every plus-strand word in
// genome order (ufindex.cpp:194-322).  As in the host builder (make_ufi.cpp), only the second and later occurrences of a
// slot depend on order: the first indexed occurrence always lands in the slot itself, and FindFreeSlot never lends out a
// slot with 1 <= count <= MaxIx (ufindex.cpp:991-993).  So the device does
//   pass 1  both counts for every word (one thread per window start, rolling 2-bit words out of an LDS tile)
//   pass 2  the first (lowest) indexed position of every slot: atomicMin
//   pass 3  head slots written straight into the 5-byte table; the remaining ("overflow") positions counted per tile
//   pass 4  the overflow positions emitted in genome order (tile offsets from a scan of the per-tile counts)
// and the host runs the order-dependent inserts over that list (urx_finish_slots_host, make_ufi.cpp).  Output is
// byte-identical to the reference's -make_ufi.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_common.h"
#include "kernels.h"

int urx_finish_slots_host(uint8_t *blob, uint64_t slots, uint32_t max_ix, uint8_t *nplus, const uint64_t *oslot,
                          const uint32_t *opos, size_t novf, uint32_t *truncated_out);

namespace {

using namespace urx;

constexpr int TILE = 1024;     // window starts per block
constexpr int THREADS = 256;   // 4 consecutive starts per thread
constexpr uint8_t T_PLUS1 = 254, T_BOTH1 = 255;

struct Scan {
	const uint8_t *seq;
	uint32_t size, W, max_ix;
	uint64_t mask, slots, magic;
	uint32_t *cplus, *cminus, *first;
	uint8_t *blob;
	uint32_t *tile_counts;
	const uint64_t *tile_offsets;
	uint64_t *oslot;
	uint32_t *opos;
};

__device__ __forceinline__ uint32_t sat255(uint32_t c) { return c > 255u ? 255u : c; }

// MODE 0 counts, 1 first occurrences, 2 heads + per-tile overflow counts, 3 overflow list in genome order
template <int MODE>
__global__ __launch_bounds__(THREADS) void scan_kernel(Scan S) {
	__shared__ uint8_t code[TILE + 32 + 8];
	__shared__ uint32_t wsum[THREADS / 64];
	const uint32_t W = S.W;
	const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
	const uint32_t nstarts = S.size >= W ? S.size - W + 1 : 0;  // window starts 0 .. size-W
	// letter codes of the tile and its W-1 halo: bits 0-1 the letter, bit 2 = not a base (alpha.cpp:1309), bit 3 = no
	// complement letter either (a lower-case 'u' has none, alpha.cpp:3525; the command line's store is upper case)
	for (uint32_t i = threadIdx.x; i < TILE + W - 1; i += THREADS) {
		const uint64_t p = tile0 + i;
		uint32_t c = 12;
		if (p < S.size) {
			const uint32_t ch = S.seq[p];
			const uint32_t L = letter_of(ch);
			c = L > 3u ? 12u : (L | (ch == 'u' ? 8u : 0u));
		}
		code[i] = (uint8_t)c;
	}
	__syncthreads();
	const uint32_t l0 = 4u * threadIdx.x;
	uint64_t wp = 0, wm = 0;
	uint32_t ninv = 0, ninvm = 0;  // letters of the window that are not bases / have no complement
	for (uint32_t i = 0; i < W; ++i) {
		const uint32_t c = code[l0 + i];
		ninv += (c >> 2) & 1u;
		ninvm += (c >> 3) & 1u;
		wp = (wp << 2) | (c & 3u);
		wm |= (uint64_t)(3u - (c & 3u)) << (2 * i);
	}
	uint32_t novf = 0;
	uint64_t myslot[4];
	uint32_t mypos[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint64_t p = tile0 + l0 + k;
		if (k > 0) {  // roll: letter l0+k-1 leaves, letter l0+k+W-1 enters
			const uint32_t out = code[l0 + k - 1], in = code[l0 + k + W - 1];
			ninv += ((in >> 2) & 1u) - ((out >> 2) & 1u);
			ninvm += ((in >> 3) & 1u) - ((out >> 3) & 1u);
			wp = (wp << 2) | (in & 3u);
			wm = (wm >> 2) | ((uint64_t)(3u - (in & 3u)) << (2 * (W - 1)));
		}
		myslot[k] = ~0ull; mypos[k] = 0;
		if (p >= nstarts) continue;
		if (MODE == 0) {
			if (ninvm == 0) atomicAdd(S.cminus + mod_slots(murmur64(wm & S.mask), S.slots, S.magic), 1u);
			if (ninv == 0) atomicAdd(S.cplus + mod_slots(murmur64(wp & S.mask), S.slots, S.magic), 1u);
			continue;
		}
		if (ninv != 0) continue;
		const uint64_t sp = mod_slots(murmur64(wp & S.mask), S.slots, S.magic);
		const uint32_t n = sat255(S.cplus[sp]), nm = sat255(S.cminus[sp]);
		if (n > S.max_ix || nm > S.max_ix) continue;  // UpdateSlot skips such words (ufindex.cpp:208-215)
		if (MODE == 1) { atomicMin(S.first + sp, (uint32_t)p); continue; }
		if (S.first[sp] == (uint32_t)p) {
			if (MODE == 2) {  // the slot's own word: BOTH1 iff it is the only plus word and no minus word maps here
				uint8_t *b = S.blob + 5 * sp;
				b[0] = (n == 1 && nm == 0) ? T_BOTH1 : T_PLUS1;
				b[1] = (uint8_t)p; b[2] = (uint8_t)(p >> 8); b[3] = (uint8_t)(p >> 16); b[4] = (uint8_t)(p >> 24);
			}
		} else {
			myslot[k] = sp; mypos[k] = (uint32_t)p;
			++novf;
		}
	}
	if (MODE < 2) return;
	// overflow positions of this tile, in position order: exclusive prefix over the threads
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int inc = wave_prefix_sum((int)novf);
	if (lane == 63) wsum[wv] = (uint32_t)inc;
	__syncthreads();
	uint32_t before = 0, total = 0;
	for (int w = 0; w < THREADS / 64; ++w) {
		if (w < wv) before += wsum[w];
		total += wsum[w];
	}
	if (MODE == 2) {
		if (threadIdx.x == 0) S.tile_counts[blockIdx.x] = total;
		return;
	}
	uint64_t at = S.tile_offsets[blockIdx.x] + before + (uint32_t)inc - novf;
#pragma unroll
	for (int k = 0; k < 4; ++k)
		if (myslot[k] != ~0ull) { S.oslot[at] = myslot[k]; S.opos[at] = mypos[k]; ++at; }
}

__global__ void fill_free_kernel(uint8_t *blob, uint64_t slots) {  // TALLY_FREE, pos 0xFFFFFFFF (ufindex.cpp:98-104)
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += stride) {
		uint8_t *b = blob + 5 * s;
		b[0] = 0; b[1] = 0xFF; b[2] = 0xFF; b[3] = 0xFF; b[4] = 0xFF;
	}
}

__global__ void sat_u8_kernel(const uint32_t *c, uint8_t *out, uint64_t n) {
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint8_t)sat255(c[i]);
}

struct DevMem {
	std::vector<void *> held;
	template <class T>
	hipError_t get(T **p, size_t bytes) {
		hipError_t e = hipMalloc((void **)p, bytes ? bytes : 1);
		if (e == hipSuccess) held.push_back(*p);
		return e;
	}
	~DevMem() { for (void *p : held) (void)hipFree(p); }
};

}  // namespace

#define TRY(x)                                                                                         \
	do {                                                                                               \
		hipError_t e_ = (x);                                                                           \
		if (e_ != hipSuccess) return e_ == hipErrorOutOfMemory ? URMAPX_E_NOMEM : URMAPX_E_NODEVICE;   \
	} while (0)

// UFIndex::MakeIndex with the counting passes, the head slots and the ordered overflow list made on `device`.
// d_seqdata: the sequence store already resident there, or NULL (then seqdata, a host array, is uploaded).
extern "C" int urmapx_build_slots_gpu(int device, const uint8_t *seqdata, const void *d_seqdata, uint32_t size, uint32_t W,
                                      uint32_t max_ix, uint64_t slots, uint8_t *blob, uint32_t *truncated_out) {
	if ((!seqdata && !d_seqdata) || !blob || slots == 0 || W < 1 || W > 32 || size == 0) return URMAPX_E_ARG;
	const bool verbose = getenv("URMAPX_VERBOSE") != nullptr;
	TRY(hipSetDevice(device));
	struct Events {  // destroyed on every return path
		hipEvent_t a = nullptr, b = nullptr;
		~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
	} evs;
	TRY(hipEventCreate(&evs.a));
	TRY(hipEventCreate(&evs.b));
	const hipEvent_t ev0 = evs.a, ev1 = evs.b;
	DevMem mem;
	uint8_t *d_seq = nullptr;
	if (d_seqdata) d_seq = (uint8_t *)d_seqdata;
	else {
		TRY(mem.get(&d_seq, size));
		TRY(hipMemcpy(d_seq, seqdata, size, hipMemcpyHostToDevice));
	}
	Scan S;
	S.seq = d_seq; S.size = size; S.W = W; S.max_ix = max_ix; S.slots = slots;
	S.mask = W >= 32 ? ~0ull : ((1ull << (2 * W)) - 1);
	S.magic = (uint64_t)((((unsigned __int128)1) << 64) / slots);
	const uint32_t ntiles = (uint32_t)(((uint64_t)size + TILE - 1) / TILE);
	uint64_t *d_offsets = nullptr;
	TRY(mem.get(&S.cplus, slots * 4));
	TRY(mem.get(&S.cminus, slots * 4));
	TRY(mem.get(&S.first, slots * 4));
	TRY(mem.get(&S.blob, 5 * slots));
	TRY(mem.get(&S.tile_counts, (size_t)ntiles * 4));
	TRY(mem.get(&d_offsets, (size_t)ntiles * 8));
	S.tile_offsets = d_offsets; S.oslot = nullptr; S.opos = nullptr;
	TRY(hipEventRecord(ev0, nullptr));
	TRY(hipMemsetAsync(S.cplus, 0, slots * 4, nullptr));
	TRY(hipMemsetAsync(S.cminus, 0, slots * 4, nullptr));
	TRY(hipMemsetAsync(S.first, 0xFF, slots * 4, nullptr));
	hipLaunchKernelGGL(fill_free_kernel, dim3(8192), dim3(256), 0, nullptr, S.blob, slots);
	hipLaunchKernelGGL(scan_kernel<0>, dim3(ntiles), dim3(THREADS), 0, nullptr, S);
	hipLaunchKernelGGL(scan_kernel<1>, dim3(ntiles), dim3(THREADS), 0, nullptr, S);
	hipLaunchKernelGGL(scan_kernel<2>, dim3(ntiles), dim3(THREADS), 0, nullptr, S);
	TRY(hipGetLastError());
	std::vector<uint32_t> counts(ntiles);
	TRY(hipMemcpy(counts.data(), S.tile_counts, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
	std::vector<uint64_t> offsets(ntiles);
	uint64_t novf = 0;
	for (uint32_t t = 0; t < ntiles; ++t) { offsets[t] = novf; novf += counts[t]; }
	TRY(hipMemcpy(d_offsets, offsets.data(), (size_t)ntiles * 8, hipMemcpyHostToDevice));
	TRY(mem.get(&S.oslot, novf * 8));
	TRY(mem.get(&S.opos, novf * 4));
	hipLaunchKernelGGL(scan_kernel<3>, dim3(ntiles), dim3(THREADS), 0, nullptr, S);
	uint8_t *d_nplus = reinterpret_cast<uint8_t *>(S.cminus);  // the minus counts are done with: their array takes the bytes
	hipLaunchKernelGGL(sat_u8_kernel, dim3(8192), dim3(256), 0, nullptr, S.cplus, d_nplus, slots);
	TRY(hipGetLastError());
	TRY(hipEventRecord(ev1, nullptr));
	TRY(hipEventSynchronize(ev1));
	float ms = 0;
	(void)hipEventElapsedTime(&ms, ev0, ev1);
	if (verbose) fprintf(stderr, "[make_ufi gpu] device passes %.2f s, %llu overflow positions\n", ms * 1e-3, (unsigned long long)novf);
	// back to the host: the table with its heads, the plus counts, the overflow list
	struct HostBuf {  // not zero-filled: each is overwritten at once
		void *p = nullptr;
		explicit HostBuf(size_t n) : p(malloc(n ? n : 1)) {}
		~HostBuf() { free(p); }
	} nplus(slots), oslot(novf * 8), opos(novf * 4);
	if (!nplus.p || !oslot.p || !opos.p) return URMAPX_E_NOMEM;
	const auto t0 = std::chrono::steady_clock::now();
	TRY(hipMemcpy(blob, S.blob, 5 * slots, hipMemcpyDeviceToHost));
	TRY(hipMemcpy(nplus.p, d_nplus, slots, hipMemcpyDeviceToHost));
	if (novf) {
		TRY(hipMemcpy(oslot.p, S.oslot, novf * 8, hipMemcpyDeviceToHost));
		TRY(hipMemcpy(opos.p, S.opos, novf * 4, hipMemcpyDeviceToHost));
	}
	const auto t1 = std::chrono::steady_clock::now();
	const int rc = urx_finish_slots_host(blob, slots, max_ix, (uint8_t *)nplus.p, (const uint64_t *)oslot.p, (const uint32_t *)opos.p,
	                                     (size_t)novf, truncated_out);
	if (verbose)
		fprintf(stderr, "[make_ufi gpu] copies to the host %.2f s, ordered inserts on the host %.2f s\n",
		        std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
	return rc;
}
