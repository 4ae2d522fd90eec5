// urmapx.hip -- C ABI (include/urmapx.h) over the gfx950 kernels.  Host side only: .ufi parsing
// (UFIndex::FromFile, ufindexio.cpp:51-115), device upload, workspace management, batch calls.
// There is no CPU compute path in this library.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"
#include "kernels.h"

using namespace urx;

namespace {

const uint32_t MAGIC1 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '1';
const uint32_t MAGIC2 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '2';
const uint32_t MAGIC3 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '3';
const uint32_t MAGIC5 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '5';
const size_t SEQ_TAIL_PAD = 4096;  // zero bytes after the sequence: windows may run past the end
const size_t BLOB_TAIL_PAD = 8;    // the last slot is fetched with an 8-byte load

bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

}  // namespace

struct urmapx_index {
	uint32_t W = 0, maxIx = 0, seqDataSize = 0;
	uint64_t slotCount = 0;
	std::vector<std::string> labels;
	std::vector<uint32_t> seqLengths, seqOffsets;
	// host copy (owned or borrowed)
	const uint8_t *h_blob = nullptr, *h_seq = nullptr;
	uint8_t *own_blob = nullptr, *own_seq = nullptr;
	// device copy
	int device = -1;
	const uint8_t *d_blob = nullptr, *d_seq = nullptr;
	uint4 *d_seqp = nullptr;  // packed copy of d_seq (4 bit planes per 32 bases), always owned, built on the device
	// GetRow_Blob's rows laid out once (chain_rows.hip), always owned, built on the device; null: not built
	uint2 *d_rowinfo = nullptr;
	uint32_t *d_rows = nullptr;
	uint4 *d_slot16 = nullptr;  // the slot table with the rows' heads inline (chain_rows.hip), always owned; null: not built
	uint64_t *d_rowbase = nullptr;
	uint64_t n_rows = 0;
	bool own_dev = false;
	uint32_t *d_seqLengths = nullptr, *d_seqOffsets = nullptr;

	DevIndex view() const {
		DevIndex X;
		X.blob = d_blob; X.seq = d_seq; X.seqp = d_seqp; X.slotCount = slotCount;
		X.slotMagic = (uint64_t)((((unsigned __int128)1) << 64) / slotCount);
		X.shiftMask = (W >= 32) ? ~0ull : ((1ull << (2 * W)) - 1ull);
		X.W = W; X.maxIx = maxIx; X.seqDataSize = seqDataSize; X.seqCount = (uint32_t)labels.size();
		X.seqLengths = d_seqLengths; X.seqOffsets = d_seqOffsets;
		X.rowinfo = d_rowinfo; X.rowbase = d_rowbase; X.rows = d_rows; X.slot16 = d_slot16;
		return X;
	}
};

struct urmapx_ctx {
	const urmapx_index *index = nullptr;
	int device = -1;
	hipStream_t stream = nullptr;
	urmapx_params params;
	DevIndex X;
	char arch[64];
	hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
	bool ev_valid = false;
	// workspace
	DevBuf<uint8_t> bases, tallies, vflags, vstatus, va, vb;
	DevBuf<uint64_t> offs, slots;
	DevBuf<uint32_t> positions, used, vaoffs, vboffs;
	DevBuf<urmapx_result> results;
	DevBuf<urmapx_path_op> pathops, vops;
	DevBuf<float> vscores;
	DevBuf<uint16_t> vnops;
	DevBuf<uint8_t> scratch, vscratch;
	DevBuf<uint32_t> statsbuf;
	DevBuf<uint8_t> pe_scratch;
	int pe_veryfast = 0;  // State2::m_Method 5
	int dp_rounds_used = 3;  // rounds of phase 6 in the last single-end call (kernels.h: DpBounds)
	uint32_t dp_bounds_used[DP_ROUNDS + 1] = {0, 2, 16, 0xFFFFFFFFu, 0xFFFFFFFFu};
	int pair_info_on = 0;  // -tabbedout: record urmapx_pair_info per pair
	DevBuf<urmapx_pair_info> pairinfo;
	DevBuf<uint32_t> ovflist;  // reads queued for the search kernel's second pass
	DevBuf<uint8_t> dpbuf, dpscratch;  // phase 6 as its own launches: jobs, paths, parked read states (kernels.h: DpWork)
	DevBuf<uint8_t> slowscratch;       // the general kernel's per-block lists (kernels_slow.hip)
	DevBuf<uint32_t> slowlist;
	int dp_blocks[6] = {0, 0, 0, 0, 0, 0};
	int fin_blocks[6] = {0, 0, 0, 0, 0, 0};
	hipEvent_t stage_ev[STAGE_EVENTS] = {};
	bool stage_valid = false;
	bool p3_parked = false;  // the last single-end call ran with phase 3 parked (URMAPX_PARK_PHASE3 and the row layout resident): urmapx_ctx_phase3 reports only then
	uint32_t pairinfo_n = 0;
	uint32_t stats_reads = 0;  // diagnostics: reads of the last single-end call with per-read cycle counts
	int pe_blocks[4] = {0, 0, 0, 0};
	int blocks[6] = {0, 0, 0, 0, 0, 0};  // persistent grid size of the search kernel for read length classes <=192, <=320, <=256, <=128, <=512, <=1024
};

namespace urx {
AllocClock &alloc_clock() { static AllocClock c{}; return c; }

// When the library is loaded: ask the HIP runtime for more than its default of four hardware queues per device, unless the process has its own
// setting.  A process's streams are spread over those queues and streams that share one take turns; urmapx_map_files runs two streams per lane
// (the lane's and its copy-back stream), a caller with contexts of its own a few more.  It only takes effect if the runtime has not started yet
// (a host that initialises HIP first -- torch -- sets GPU_MAX_HW_QUEUES itself: INTEGRATION.md); measured: profiles/r6/hw_queues.txt.
__attribute__((constructor)) static void urx_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
hipStream_t ctx_stream(urmapx_ctx *C) { return C->stream; }
int ctx_device(const urmapx_ctx *C) { return C->device; }
const urmapx_index *ctx_index(const urmapx_ctx *C) { return C->index; }
}  // namespace urx

extern "C" {

const char *urmapx_strerror(int code) {
	switch (code) {
	case URMAPX_OK: return "ok";
	case URMAPX_E_IO: return "I/O error";
	case URMAPX_E_FORMAT: return "bad .ufi format";
	case URMAPX_E_NOMEM: return "out of memory";
	case URMAPX_E_NODEVICE: return "no usable HIP device / HIP runtime error";
	case URMAPX_E_ARG: return "invalid argument";
	case URMAPX_E_UNSUPPORTED: return "input outside the device path's domain";
	default: return "unknown error";
	}
}

int urmapx_params_for_method(unsigned method, urmapx_params *p) {  // state1.cpp:147-183
	if (!p) return URMAPX_E_ARG;
	if (method == 6 || method == 8) { *p = urmapx_params{-3, -5, -1, 20, 60, 9, 100, 1, 1, 1, 12}; return URMAPX_OK; }
	if (method == 7) { *p = urmapx_params{-4, -6, -2, 35, 35, 12, 75, 8, 6, 5, 8}; return URMAPX_OK; }
	return URMAPX_E_ARG;
}

static int set_directory(urmapx_index *I, uint32_t seq_count, const uint32_t *seq_lengths, const uint32_t *offsets,
                         const char *labels) {
	const char *p = labels;
	for (uint32_t i = 0; i < seq_count; ++i) {
		I->seqLengths.push_back(seq_lengths[i]);
		I->seqOffsets.push_back(offsets[i]);
		I->labels.push_back(std::string(p));
		p += strlen(p) + 1;
	}
	return URMAPX_OK;
}

int urmapx_index_open(const char *path, urmapx_index **out) {
	if (!path || !out) return URMAPX_E_ARG;
	*out = nullptr;
	FILE *f = fopen(path, "rb");
	if (!f) return URMAPX_E_IO;
	urmapx_index *I = new urmapx_index;
	uint32_t u = 0, seqCount = 0;
	bool ok = rd(f, &u, 4) && u == MAGIC1 && rd(f, &I->W, 4) && rd(f, &I->maxIx, 4) && rd(f, &I->seqDataSize, 4) &&
	          rd(f, &I->slotCount, 8) && rd(f, &seqCount, 4);
	for (uint32_t i = 0; ok && i < seqCount; ++i) {
		uint32_t L, off, n;
		ok = rd(f, &L, 4) && rd(f, &off, 4) && rd(f, &n, 4) && n < (1u << 20);
		if (!ok) break;
		std::string s(n, '\0');
		ok = rd(f, &s[0], n);
		I->seqLengths.push_back(L);
		I->seqOffsets.push_back(off);
		I->labels.push_back(std::string(s.c_str()));
	}
	ok = ok && rd(f, &u, 4) && u == MAGIC2 && I->slotCount > 0 && I->W >= 1 && I->W <= 32;
	if (!ok) { fclose(f); delete I; return URMAPX_E_FORMAT; }
	I->own_blob = (uint8_t *)malloc(5 * I->slotCount + BLOB_TAIL_PAD);
	I->own_seq = (uint8_t *)calloc((size_t)I->seqDataSize + SEQ_TAIL_PAD, 1);
	if (!I->own_blob || !I->own_seq) { fclose(f); urmapx_index_close(I); return URMAPX_E_NOMEM; }
	memset(I->own_blob + 5 * I->slotCount, 0, BLOB_TAIL_PAD);
	ok = rd(f, I->own_blob, 5 * I->slotCount) && rd(f, &u, 4) && u == MAGIC3 && rd(f, I->own_seq, I->seqDataSize) &&
	     rd(f, &u, 4) && u == MAGIC5;
	fclose(f);
	if (!ok) { urmapx_index_close(I); return URMAPX_E_FORMAT; }
	I->h_blob = I->own_blob;
	I->h_seq = I->own_seq;
	*out = I;
	return URMAPX_OK;
}

static int upload_directory(urmapx_index *I);

// nbytes of the file at file_off -> dev: several threads pread pieces of 128 MB into page-locked buffers, each piece goes out with
// an asynchronous copy while the next is being read (three buffers in flight)
static int stream_to_device(int fd, uint64_t file_off, size_t nbytes, uint8_t *dev, int threads) {
	constexpr size_t PIECE = 128u << 20;
	constexpr int NS = 3;
	char *stage[NS] = {nullptr, nullptr, nullptr};
	hipEvent_t ev[NS] = {};
	hipStream_t st = nullptr;
	int rc = URMAPX_OK;
	auto cleanup = [&]() {
		if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
		for (int k = 0; k < NS; ++k) {
			if (stage[k]) (void)hipHostFree(stage[k]);
			if (ev[k]) (void)hipEventDestroy(ev[k]);
		}
	};
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; cleanup(); return URMAPX_E_NODEVICE; }
	const size_t stage_bytes = std::min(PIECE, std::max<size_t>(nbytes, 1));
	for (int k = 0; k < NS; ++k)
		if (hipHostMalloc((void **)&stage[k], stage_bytes, hipHostMallocDefault) != hipSuccess ||
		    hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); cleanup(); return URMAPX_E_NOMEM; }
	if (threads < 1) threads = 1;
	size_t piece = 0;
	for (size_t off = 0; off < nbytes && rc == URMAPX_OK; off += PIECE, ++piece) {
		const int k = (int)(piece % NS);
		const size_t n = std::min(PIECE, nbytes - off);
		if (piece >= (size_t)NS && hipEventSynchronize(ev[k]) != hipSuccess) { rc = URMAPX_E_NODEVICE; break; }
		std::atomic<bool> ok{true};
		std::vector<std::thread> th;
		const int T = (int)std::min<size_t>((size_t)threads, (n + (4u << 20) - 1) / (4u << 20));
		for (int t = 0; t < T; ++t)
			th.emplace_back([&, t] {
				const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
				size_t got = lo;
				while (got < hi) {
					const ssize_t r = pread(fd, stage[k] + got, hi - got, (off_t)(file_off + off + got));
					if (r <= 0) { ok.store(false); return; }
					got += (size_t)r;
				}
			});
		for (auto &x : th) x.join();
		if (!ok.load()) { rc = URMAPX_E_FORMAT; break; }  // the file ends inside the array
		if (hipMemcpyAsync(dev + off, stage[k], n, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) rc = URMAPX_E_NODEVICE;
	}
	if (st && hipStreamSynchronize(st) != hipSuccess && rc == URMAPX_OK) rc = URMAPX_E_NODEVICE;
	cleanup();
	return rc;
}

// UFIndex::FromFile (ufindexio.cpp:51-115) straight into HBM.  urmapx_index_open reads the file into host arrays with one thread (27 GB at hg38
// scale: 5.5 s out of /dev/shm) and urmapx_index_upload copies them out of pageable memory (3.3 s more); a run of `urmap -map` that maps 10 M
// reads in 0.7 s spent 9 s there.  Here the header is parsed as before and the two arrays stream from the file to the device through
// page-locked buffers, read by several threads; nothing of them stays on the host (urmapx_index_replicate copies device to device then).
int urmapx_index_open_device(const char *path, int device, urmapx_index **out) {
	if (!path || !out) return URMAPX_E_ARG;
	*out = nullptr;
	FILE *f = fopen(path, "rb");
	if (!f) return URMAPX_E_IO;
	urmapx_index *I = new urmapx_index;
	uint32_t u = 0, seqCount = 0;
	bool ok = rd(f, &u, 4) && u == MAGIC1 && rd(f, &I->W, 4) && rd(f, &I->maxIx, 4) && rd(f, &I->seqDataSize, 4) &&
	          rd(f, &I->slotCount, 8) && rd(f, &seqCount, 4);
	for (uint32_t i = 0; ok && i < seqCount; ++i) {
		uint32_t L, off, n;
		ok = rd(f, &L, 4) && rd(f, &off, 4) && rd(f, &n, 4) && n < (1u << 20);
		if (!ok) break;
		std::string s(n, '\0');
		ok = rd(f, &s[0], n);
		I->seqLengths.push_back(L);
		I->seqOffsets.push_back(off);
		I->labels.push_back(std::string(s.c_str()));
	}
	ok = ok && rd(f, &u, 4) && u == MAGIC2 && I->slotCount > 0 && I->W >= 1 && I->W <= 32;
	if (!ok) { fclose(f); delete I; return URMAPX_E_FORMAT; }
	const uint64_t blob_off = (uint64_t)ftello(f);
	const size_t nb = 5 * I->slotCount;
	const uint64_t seq_off = blob_off + nb + 4;
	// the two magic words behind the arrays, before anything is allocated
	uint32_t m3 = 0, m5 = 0;
	const int fd = fileno(f);
	ok = pread(fd, &m3, 4, (off_t)(blob_off + nb)) == 4 && m3 == MAGIC3 && pread(fd, &m5, 4, (off_t)(seq_off + I->seqDataSize)) == 4 && m5 == MAGIC5;
	if (!ok) { fclose(f); delete I; return URMAPX_E_FORMAT; }
	hipError_t e = hipSetDevice(device);
	uint8_t *db = nullptr, *ds = nullptr;
	if (e == hipSuccess) e = hipMalloc((void **)&db, nb + BLOB_TAIL_PAD);
	if (e == hipSuccess) e = hipMalloc((void **)&ds, (size_t)I->seqDataSize + SEQ_TAIL_PAD);
	if (e != hipSuccess) { (void)hipGetLastError(); if (db) (void)hipFree(db); fclose(f); delete I; return hip_rc(e); }
	I->d_blob = db; I->d_seq = ds; I->own_dev = true; I->device = device;
	int rc = URMAPX_OK;
	if (hipMemset(db + nb, 0, BLOB_TAIL_PAD) != hipSuccess || hipMemset(ds + I->seqDataSize, 0, SEQ_TAIL_PAD) != hipSuccess) rc = URMAPX_E_NODEVICE;
	int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	if (const char *e = getenv("URMAPX_LOAD_THREADS")) threads = std::max(1, std::min(256, atoi(e)));  // (measurement)
	const auto t0 = std::chrono::steady_clock::now();
	if (!rc) rc = stream_to_device(fd, blob_off, nb, db, threads);
	if (!rc) rc = stream_to_device(fd, seq_off, I->seqDataSize, ds, threads);
	fclose(f);
	const auto t1 = std::chrono::steady_clock::now();
	if (!rc) rc = upload_directory(I);
	if (getenv("URMAPX_VERBOSE"))
		fprintf(stderr, "urmapx: index streamed to device %d: %.2f GB in %.2f s (%d reader threads), resident layouts built in %.2f s\n", device,
		        (double)(nb + I->seqDataSize) / 1e9, std::chrono::duration<double>(t1 - t0).count(), threads,
		        std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
	if (rc) { urmapx_index_close(I); return rc; }
	*out = I;
	return URMAPX_OK;
}

int urmapx_index_wrap_host(uint32_t W, uint32_t max_ix, uint64_t slot_count, const uint8_t *blob, const uint8_t *seqdata,
                           uint32_t seqdata_size, uint32_t seq_count, const uint32_t *seq_lengths,
                           const uint32_t *offsets, const char *labels, urmapx_index **out) {
	if (!out || !blob || !seqdata || slot_count == 0 || W < 1 || W > 32) return URMAPX_E_ARG;
	urmapx_index *I = new urmapx_index;
	I->W = W; I->maxIx = max_ix; I->slotCount = slot_count; I->seqDataSize = seqdata_size;
	I->h_blob = blob; I->h_seq = seqdata;
	set_directory(I, seq_count, seq_lengths, offsets, labels);
	*out = I;
	return URMAPX_OK;
}

// ExtendPen's windows are read from a packed copy of the sequence store (0.5 byte per base, dev_common.h)
static int build_packed_seq(urmapx_index *I) {
	HIP_TRY(hipMalloc((void **)&I->d_seqp, packed_seq_blocks(I->seqDataSize) * sizeof(uint4)));
	HIP_TRY(launch_pack_seq(I->d_seq, I->seqDataSize, I->d_seqp, nullptr));
	HIP_TRY(hipDeviceSynchronize());
	return URMAPX_OK;
}

static int upload_directory(urmapx_index *I) {
	int rc = build_packed_seq(I);
	if (rc) return rc;
	// every chain head's row, contiguous (chain_rows.hip): 4 bytes per slot + 4 per indexed position.  URMAPX_NO_CHAIN_ROWS=1
	// (measurement, tests): the kernels walk the chains hop by hop, as they do when the rows do not fit the device
	// round 6: slot16 and the rows in one go, without the 8 bytes of info per slot the two-step build below holds meanwhile (43 GB at hg38 scale:
	// 160 GB at the peak of an upload instead of 122).  The two-step build remains for whoever wants the info entries (URMAPX_NO_SLOT16,
	// URMAPX_PARK_PHASE3, URMAPX_KEEP_ROWINFO, URMAPX_TWO_STEP_LAYOUT=1 for the A/B) and as the fallback when there is no room for slot16.
	if (!getenv("URMAPX_NO_CHAIN_ROWS") && !getenv("URMAPX_NO_SLOT16") && !getenv("URMAPX_PARK_PHASE3") && !getenv("URMAPX_KEEP_ROWINFO") && !getenv("URMAPX_TWO_STEP_LAYOUT")) {
		const hipError_t e = build_slot16_direct(I->d_blob, I->slotCount, I->maxIx, &I->d_slot16, &I->d_rows, &I->n_rows);
		if (getenv("URMAPX_VERBOSE"))
			fprintf(stderr, "urmapx: slot16 + rows %s in one build: %llu positions in rows, %.2f GB (hip: %s)\n", I->d_slot16 ? "built" : "NOT built",
			        (unsigned long long)I->n_rows, I->d_slot16 ? (16.0 * (double)I->slotCount + 4.0 * (double)I->n_rows) / 1e9 : 0.0, hipGetErrorString(e));
		HIP_TRY(e);
	}
	if (!I->d_slot16 && !getenv("URMAPX_NO_CHAIN_ROWS")) {
		const hipError_t e = build_chain_rows(I->d_blob, I->slotCount, I->maxIx, &I->d_rowinfo, &I->d_rowbase, &I->d_rows, &I->n_rows);
		if (getenv("URMAPX_VERBOSE"))
			fprintf(stderr, "urmapx: chain rows %s: %llu positions in rows, %.2f GB (hip: %s)\n", I->d_rows ? "built" : "NOT built",
			        (unsigned long long)I->n_rows, I->d_rowinfo ? (8.0 * (double)I->slotCount + 4.0 * (double)I->n_rows) / 1e9 : 0.0, hipGetErrorString(e));
		HIP_TRY(e);
		// round 5: the 16-byte slot table for the single-end search kernel (86 GB at hg38 scale; URMAPX_NO_SLOT16=1: not built)
		if (I->d_rowinfo && !getenv("URMAPX_NO_SLOT16")) {
			const hipError_t e2 = build_slot16(I->d_blob, I->slotCount, I->d_rowinfo, I->d_rowbase, I->d_rows, &I->d_slot16);
			if (getenv("URMAPX_VERBOSE")) fprintf(stderr, "urmapx: slot16 table %s (%.2f GB)\n", I->d_slot16 ? "built" : "NOT built", 16.0 * (double)I->slotCount / 1e9);
			HIP_TRY(e2);
			// The per-slot info entries and the group bases have done their work once slot16 holds what they said (43 GB at hg38 scale):
			// every kernel that looks rows up reads slot16 then.  They stay for the one user that still wants them, the parked
			// phase 3 (URMAPX_PARK_PHASE3, measurement), and on request (URMAPX_KEEP_ROWINFO).
			if (I->d_slot16 && !getenv("URMAPX_PARK_PHASE3") && !getenv("URMAPX_KEEP_ROWINFO")) {
				(void)hipFree(I->d_rowinfo); I->d_rowinfo = nullptr;
				(void)hipFree(I->d_rowbase); I->d_rowbase = nullptr;
			}
		}
		//  a replica without the layout maps 15 % slower with the same results: say so once, whoever asked for the upload
		static bool warned = false;
		if (!I->d_rowinfo && !I->d_slot16 && !warned) {
			warned = true;
			fprintf(stderr, "urmapx: the chain-row layout of the index was not built on device %d (no room for it, or MaxIx over 32): "
			                "collision chains are walked link by link\n", I->device);
		}
	}
	size_t n = I->labels.size();
	HIP_TRY(hipMalloc((void **)&I->d_seqLengths, (n + 1) * 4));
	HIP_TRY(hipMalloc((void **)&I->d_seqOffsets, (n + 1) * 4));
	HIP_TRY(hipMemcpy(I->d_seqLengths, I->seqLengths.data(), n * 4, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(I->d_seqOffsets, I->seqOffsets.data(), n * 4, hipMemcpyHostToDevice));
	return URMAPX_OK;
}

int urmapx_index_wrap_device(int device, uint32_t W, uint32_t max_ix, uint64_t slot_count, const void *d_blob,
                             const void *d_seqdata, uint32_t seqdata_size, uint32_t seq_count,
                             const uint32_t *seq_lengths, const uint32_t *offsets, const char *labels,
                             urmapx_index **out) {
	if (!out || !d_blob || !d_seqdata || slot_count == 0 || W < 1 || W > 32) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(device));
	urmapx_index *I = new urmapx_index;
	I->W = W; I->maxIx = max_ix; I->slotCount = slot_count; I->seqDataSize = seqdata_size;
	I->device = device; I->d_blob = (const uint8_t *)d_blob; I->d_seq = (const uint8_t *)d_seqdata; I->own_dev = false;
	set_directory(I, seq_count, seq_lengths, offsets, labels);
	int rc = upload_directory(I);
	if (rc) { urmapx_index_close(I); return rc; }
	*out = I;
	return URMAPX_OK;
}

int urmapx_index_upload(urmapx_index *I, int device) {
	if (!I) return URMAPX_E_ARG;
	if (I->d_blob && I->device == device) return URMAPX_OK;
	if (I->d_blob) return URMAPX_E_ARG;  // one device per index object (one process per GPU)
	if (!I->h_blob || !I->h_seq) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(device));
	uint8_t *db = nullptr, *ds = nullptr;
	const size_t nb = 5 * I->slotCount;
	HIP_TRY(hipMalloc((void **)&db, nb + BLOB_TAIL_PAD));
	hipError_t e = hipMalloc((void **)&ds, (size_t)I->seqDataSize + SEQ_TAIL_PAD);
	if (e != hipSuccess) { (void)hipFree(db); return hip_rc(e); }
	I->d_blob = db; I->d_seq = ds; I->own_dev = true; I->device = device;
	HIP_TRY(hipMemset(db + nb, 0, BLOB_TAIL_PAD));
	HIP_TRY(hipMemset(ds + I->seqDataSize, 0, SEQ_TAIL_PAD));
	HIP_TRY(hipMemcpy(db, I->h_blob, nb, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(ds, I->h_seq, I->seqDataSize, hipMemcpyHostToDevice));
	return upload_directory(I);
}

// One more replica of the same index on another GPU (the reference shares one UFIndex between all its threads,
// map.cpp:43-61; here every device holds its own copy in HBM).  The new object borrows src's host arrays.
int urmapx_index_replicate(const urmapx_index *src, int device, urmapx_index **out) {
	if (!src || !out || ((!src->h_blob || !src->h_seq) && (!src->d_blob || !src->d_seq))) return URMAPX_E_ARG;
	*out = nullptr;
	urmapx_index *I = new urmapx_index;
	I->W = src->W; I->maxIx = src->maxIx; I->seqDataSize = src->seqDataSize; I->slotCount = src->slotCount;
	I->labels = src->labels; I->seqLengths = src->seqLengths; I->seqOffsets = src->seqOffsets;
	int rc;
	if (src->h_blob && src->h_seq) {
		I->h_blob = src->h_blob; I->h_seq = src->h_seq;
		rc = urmapx_index_upload(I, device);
	} else {
		// no host copy (urmapx_index_open_device, urmapx_index_wrap_device): the table and the sequence travel device to device
		// (xGMI between the GPUs of a node); the packed sequence, the rows and slot16 are rebuilt from them on the new device
		const size_t nb = 5 * I->slotCount;
		uint8_t *db = nullptr, *ds = nullptr;
		hipError_t e = hipSetDevice(device);
		if (e == hipSuccess) e = hipMalloc((void **)&db, nb + BLOB_TAIL_PAD);
		if (e == hipSuccess) e = hipMalloc((void **)&ds, (size_t)I->seqDataSize + SEQ_TAIL_PAD);
		if (e != hipSuccess) { (void)hipGetLastError(); if (db) (void)hipFree(db); delete I; return hip_rc(e); }
		I->d_blob = db; I->d_seq = ds; I->own_dev = true; I->device = device;
		if (e == hipSuccess) e = hipMemcpyPeer(db, device, src->d_blob, src->device, nb + BLOB_TAIL_PAD);
		if (e == hipSuccess) e = hipMemcpyPeer(ds, device, src->d_seq, src->device, (size_t)I->seqDataSize + SEQ_TAIL_PAD);
		if (e == hipSuccess) e = hipDeviceSynchronize();
		rc = e == hipSuccess ? upload_directory(I) : hip_rc(e);
	}
	if (rc) { urmapx_index_close(I); return rc; }
	*out = I;
	return URMAPX_OK;
}

void urmapx_index_close(urmapx_index *I) {
	if (!I) return;
	lane_pool_purge(I);  // mapping contexts urmapx_map_files kept for this index
	if (I->own_dev) { (void)hipFree((void *)I->d_blob); (void)hipFree((void *)I->d_seq); }
	if (I->d_seqp) (void)hipFree(I->d_seqp);
	if (I->d_slot16) (void)hipFree(I->d_slot16);
	if (I->d_rowinfo) (void)hipFree(I->d_rowinfo);
	if (I->d_rowbase) (void)hipFree(I->d_rowbase);
	if (I->d_rows) (void)hipFree(I->d_rows);
	if (I->d_seqLengths) (void)hipFree(I->d_seqLengths);
	if (I->d_seqOffsets) (void)hipFree(I->d_seqOffsets);
	free(I->own_blob);
	free(I->own_seq);
	delete I;
}

// bytes of the chain-row layout resident with the index (0: not built -- URMAPX_NO_CHAIN_ROWS, MaxIx over 32, or no room)
uint64_t urmapx_index_chain_row_bytes(const urmapx_index *I) {
	if (!I) return 0ull;
	return (I->d_rowinfo ? 8ull * I->slotCount + 8ull * ((I->slotCount + 1023) / 1024) : 0ull) + (I->d_rows ? 4ull * (I->n_rows + 64) : 0ull) +
	       (I->d_slot16 ? 16ull * (I->slotCount + 1) : 0ull);
}
int urmapx_index_validate(const urmapx_index *I, urmapx_validate_report *out) {
	if (!I || !out) return URMAPX_E_ARG;
	memset(out, 0, sizeof *out);
	out->first_bad_slot = ~0ull;
	if (!I->d_blob || !I->d_seq || I->device < 0) return URMAPX_E_ARG;  // urmapx_index_upload first
	HIP_TRY(hipSetDevice(I->device));
	uint64_t v[9];
	float ms = 0.f;
	HIP_TRY(validate_index(I->view(), v, &ms));
	out->slots = I->slotCount;
	out->heads = v[0]; out->positions = v[1]; out->used = v[2]; out->reached = v[3];
	out->bad_hash = v[4]; out->bad_pos = v[5]; out->bad_link = v[6]; out->bad_len = v[7]; out->first_bad_slot = v[8];
	out->seconds = ms * 1e-3;
	const bool ok = !v[4] && !v[5] && !v[6] && !v[7] && v[2] == v[3];
	return ok ? URMAPX_OK : URMAPX_E_FORMAT;
}
int urmapx_checksum_device(int device, const void *d_ptr, uint64_t nbytes, uint64_t *out) {
	if (!d_ptr || !out || ((uintptr_t)d_ptr & 7)) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(checksum_device(d_ptr, nbytes, out));
	return URMAPX_OK;
}
int urmapx_index_checksum(const urmapx_index *I, uint64_t out[2]) {
	if (!I || !out) return URMAPX_E_ARG;
	if (!I->d_blob || !I->d_seq || I->device < 0) return URMAPX_E_ARG;  // urmapx_index_upload first
	HIP_TRY(hipSetDevice(I->device));
	HIP_TRY(checksum_device(I->d_blob, 5ull * I->slotCount, &out[0]));
	HIP_TRY(checksum_device(I->d_seq, I->seqDataSize, &out[1]));
	return URMAPX_OK;
}
int urmapx_index_layout_checksum(const urmapx_index *I, uint64_t out[2]) {
	if (!I || !out) return URMAPX_E_ARG;
	if (!I->d_blob || I->device < 0) return URMAPX_E_ARG;
	out[0] = out[1] = 0;
	HIP_TRY(hipSetDevice(I->device));
	if (I->d_slot16) HIP_TRY(checksum_device(I->d_slot16, 16ull * I->slotCount, &out[0]));
	if (I->d_rows) HIP_TRY(checksum_device(I->d_rows, 4ull * I->n_rows, &out[1]));
	return URMAPX_OK;
}
uint32_t urmapx_index_word_length(const urmapx_index *I) { return I->W; }
uint32_t urmapx_index_max_ix(const urmapx_index *I) { return I->maxIx; }
uint64_t urmapx_index_slot_count(const urmapx_index *I) { return I->slotCount; }
uint32_t urmapx_index_seqdata_size(const urmapx_index *I) { return I->seqDataSize; }
uint32_t urmapx_index_seq_count(const urmapx_index *I) { return (uint32_t)I->labels.size(); }
const char *urmapx_index_label(const urmapx_index *I, uint32_t i) { return i < I->labels.size() ? I->labels[i].c_str() : nullptr; }
uint32_t urmapx_index_seq_length(const urmapx_index *I, uint32_t i) { return i < I->seqLengths.size() ? I->seqLengths[i] : 0; }
uint32_t urmapx_index_seq_offset(const urmapx_index *I, uint32_t i) { return i < I->seqOffsets.size() ? I->seqOffsets[i] : 0; }

int urmapx_ctx_create(const urmapx_index *I, int device, const urmapx_params *P, urmapx_ctx **out) {
	if (!I || !P || !out) return URMAPX_E_ARG;
	*out = nullptr;
	if (!I->d_blob || I->device != device) return URMAPX_E_ARG;  // urmapx_index_upload first
	HIP_TRY(hipSetDevice(device));
	urmapx_ctx *C = new urmapx_ctx;
	C->index = I; C->device = device; C->params = *P; C->X = I->view();
	hipDeviceProp_t prop;
	hipError_t e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) { delete C; return hip_rc(e); }
	snprintf(C->arch, sizeof C->arch, "%s", prop.gcnArchName);
	if (char *colon = strchr(C->arch, ':')) *colon = 0;
	e = hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete C; return hip_rc(e); }
	for (int i = 0; i < 3; ++i) {
		e = hipEventCreate(&C->ev[i]);
		if (e != hipSuccess) { urmapx_ctx_destroy(C); return hip_rc(e); }
	}
	for (int i = 0; i < STAGE_EVENTS; ++i) {
		e = hipEventCreate(&C->stage_ev[i]);
		if (e != hipSuccess) { urmapx_ctx_destroy(C); return hip_rc(e); }
	}
	*out = C;
	return URMAPX_OK;
}

void urmapx_ctx_destroy(urmapx_ctx *C) {
	if (!C) return;
	(void)hipSetDevice(C->device);
	if (C->stream) (void)hipStreamSynchronize(C->stream);
	C->bases.release(); C->tallies.release(); C->vflags.release(); C->vstatus.release(); C->va.release(); C->vb.release();
	C->offs.release(); C->slots.release(); C->positions.release(); C->used.release(); C->vaoffs.release(); C->vboffs.release();
	C->results.release(); C->pathops.release(); C->vops.release(); C->vscores.release(); C->vnops.release();
	C->dpbuf.release(); C->dpscratch.release(); C->slowscratch.release(); C->slowlist.release();
	for (int i = 0; i < STAGE_EVENTS; ++i)
		if (C->stage_ev[i]) (void)hipEventDestroy(C->stage_ev[i]);
	C->scratch.release(); C->vscratch.release(); C->statsbuf.release(); C->pe_scratch.release(); C->pairinfo.release(); C->ovflist.release();
	for (int i = 0; i < 3; ++i)
		if (C->ev[i]) (void)hipEventDestroy(C->ev[i]);
	if (C->stream) (void)hipStreamDestroy(C->stream);
	delete C;
}

const char *urmapx_device_arch(urmapx_ctx *C) { return C ? C->arch : nullptr; }

// diagnostic: shader cycles per phase of the last search kernel (needs URMAPX_PHASE_STATS=1 in the environment)
int urmapx_ctx_phase_cycles(urmapx_ctx *C, uint64_t out[12]) {
	if (!C || !out || !C->statsbuf.p) return URMAPX_E_ARG;
	HIP_TRY(hipStreamSynchronize(C->stream));
	uint64_t buf[13];
	HIP_TRY(hipMemcpy(buf, C->statsbuf.p, sizeof buf, hipMemcpyDeviceToHost));
	for (int i = 0; i < 12; ++i) out[i] = buf[i + 1];
	return URMAPX_OK;
}

// Device time (ms) of the launches of the last single-end *_device call: [0] search (first pass), [1] its flank-DP
// launches summed, [2] its finalize launches summed, [3..5] the same for the second pass over the reads whose lists
// outgrew the first, [6] the general kernel over what both passes left flagged (collect + search_se_slow_kernel).
int urmapx_ctx_stage_ms(urmapx_ctx *C, float ms[7]) {
	if (!C || !ms) return URMAPX_E_ARG;
	if (!C->stage_valid) { for (int i = 0; i < 7; ++i) ms[i] = 0; return URMAPX_OK; }  // no stamps were taken
	HIP_TRY(hipEventSynchronize(C->stage_ev[STAGE_LAST]));
	auto span = [&](int a, int b, float &out) -> hipError_t { float t = 0; hipError_t e = hipEventElapsedTime(&t, C->stage_ev[a], C->stage_ev[b]); out += t; return e; };
	for (int i = 0; i < 7; ++i) ms[i] = 0;
	HIP_TRY(span(0, 1, ms[0]));
	for (int p = 0; p < 2; ++p) {
		const int b = 1 + (2 * DP_ROUNDS + 1) * p;  // the event before this pass's first dp launch
		for (int rd = 0; rd < DP_ROUNDS; ++rd) {
			HIP_TRY(span(b + 2 * rd, b + 2 * rd + 1, ms[1 + 3 * p]));
			HIP_TRY(span(b + 2 * rd + 1, b + 2 * rd + 2, ms[2 + 3 * p]));
		}
	}
	HIP_TRY(span(1 + 2 * DP_ROUNDS, 2 + 2 * DP_ROUNDS, ms[3]));
	HIP_TRY(span(STAGE_LAST - 1, STAGE_LAST, ms[6]));
	return URMAPX_OK;
}

// Round 5: with phase 3 parked the search stage ([0] of urmapx_ctx_stage_ms) is three launches: ms[0] the first search launch,
// ms[1] phase 3's flank-DP launch, ms[2] the search launch over the reads parked at phase 3; stats[0] = DpJobs made for phase 3,
// stats[1] = reads parked there.  All zero when phase 3 ran inside the search kernel.
int urmapx_ctx_phase3(urmapx_ctx *C, float ms[3], uint32_t stats[2]) {
	if (!C || !ms || !stats) return URMAPX_E_ARG;
	ms[0] = ms[1] = ms[2] = 0; stats[0] = stats[1] = 0;
	// (what is reported is the real state of the call, not a guess from the times -- two events with nothing between them are microseconds apart
	// once another context's launches share the device)
	if (!C->stage_valid || !C->dpbuf.p || !C->p3_parked) return URMAPX_OK;
	HIP_TRY(hipEventSynchronize(C->stage_ev[STAGE_LAST]));
	HIP_TRY(hipEventElapsedTime(&ms[0], C->stage_ev[0], C->stage_ev[STAGE_P3_MAIN]));
	HIP_TRY(hipEventElapsedTime(&ms[1], C->stage_ev[STAGE_P3_MAIN], C->stage_ev[STAGE_P3_DP]));
	HIP_TRY(hipEventElapsedTime(&ms[2], C->stage_ev[STAGE_P3_DP], C->stage_ev[1]));
	uint32_t buf[2];
	HIP_TRY(hipMemcpy(buf, C->dpbuf.p + 32, sizeof buf, hipMemcpyDeviceToHost));
	stats[0] = buf[0]; stats[1] = buf[1];
	return URMAPX_OK;
}

int urmapx_ctx_round_ms(urmapx_ctx *C, float ms[16], int *rounds) {
	if (!C || !ms || !rounds) return URMAPX_E_ARG;
	static_assert(2 * DP_ROUNDS <= 16, "ms[16]");
	for (int i = 0; i < 16; ++i) ms[i] = 0;
	*rounds = C->dp_rounds_used;
	if (!C->stage_valid) return URMAPX_OK;
	HIP_TRY(hipEventSynchronize(C->stage_ev[STAGE_LAST]));
	for (int i = 0; i < 2 * C->dp_rounds_used; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], C->stage_ev[1 + i], C->stage_ev[2 + i]));
	return URMAPX_OK;
}

// the round boundaries of the last single-end call: round rd = a read's HSPs lo[rd] <= k < lo[rd + 1] (the last round is open ended)
int urmapx_ctx_dp_rounds(urmapx_ctx *C, uint32_t lo[8], int *rounds) {
	if (!C || !lo || !rounds) return URMAPX_E_ARG;
	*rounds = C->dp_rounds_used;
	for (int i = 0; i < 8; ++i) lo[i] = i <= DP_ROUNDS ? C->dp_bounds_used[i] : 0xFFFFFFFFu;
	return URMAPX_OK;
}

// Statistics of the last single-end *_device call: per pass {jobs made, reads parked, jobs whose DP the replay needed}
int urmapx_ctx_dp_stats(urmapx_ctx *C, uint32_t out[8]) {
	if (!C || !out || !C->dpbuf.p) return URMAPX_E_ARG;
	HIP_TRY(hipStreamSynchronize(C->stream));
	uint32_t buf[8];
	HIP_TRY(hipMemcpy(buf, C->dpbuf.p, sizeof buf, hipMemcpyDeviceToHost));
	for (int p = 0; p < 2; ++p)
		for (int i = 0; i < 4; ++i) out[4 * p + i] = buf[4 * p + i];
	return URMAPX_OK;
}

// diagnostic: shader cycles / 16 each read of the last single-end call took (URMAPX_PHASE_STATS=1; 150 / 250 bp classes)
int urmapx_ctx_read_cycles(urmapx_ctx *C, uint32_t *out, uint32_t n) {
	if (!C || !out || !C->statsbuf.p || n > C->stats_reads) return URMAPX_E_ARG;
	HIP_TRY(hipStreamSynchronize(C->stream));
	HIP_TRY(hipMemcpy(out, C->statsbuf.p + 64, (size_t)n * 4, hipMemcpyDeviceToHost));
	return URMAPX_OK;
}

int urmapx_ctx_sync(urmapx_ctx *C) {
	if (!C) return URMAPX_E_ARG;
	HIP_TRY(hipStreamSynchronize(C->stream));
	return URMAPX_OK;
}

int urmapx_ctx_last_kernel_ms(urmapx_ctx *C, float ms[2]) {
	if (!C || !ms || !C->ev_valid) return URMAPX_E_ARG;
	HIP_TRY(hipEventSynchronize(C->ev[2]));
	HIP_TRY(hipEventElapsedTime(&ms[0], C->ev[0], C->ev[1]));
	HIP_TRY(hipEventElapsedTime(&ms[1], C->ev[1], C->ev[2]));
	return URMAPX_OK;
}

static constexpr int PE_SLOW_BLOCKS = 32;  // grid of the general pair kernel (9.5 MB of lists per block)

static int ensure_probe(urmapx_ctx *C, uint64_t total_bases) {
	int rc;
	if ((rc = C->slots.ensure(2 * total_bases + 64))) return rc;
	if ((rc = C->tallies.ensure(2 * total_bases + 64))) return rc;
	if ((rc = C->positions.ensure(2 * total_bases + 64))) return rc;
	return URMAPX_OK;
}

int urmapx_map_se_device(urmapx_ctx *C, const void *d_bases, const void *d_offs, uint32_t n, uint64_t total_bases,
                         uint32_t max_read_len, void *d_results, void *d_path_ops, void *d_path_used) {
	if (!C || (n && (!d_bases || !d_offs || !d_results || !d_path_ops || !d_path_used))) return URMAPX_E_ARG;
	if (max_read_len > URMAPX_MAX_QL_SLOW) return URMAPX_E_UNSUPPORTED;
	HIP_TRY(hipSetDevice(C->device));
	(void)total_bases;
	int rc;
	// the fast kernels are sized for the batch's longest read up to URMAPX_MAX_QL; longer reads (and whatever else the
	// fast passes flag) go to the general kernel afterwards
	const uint32_t slow_qcap = max_read_len < URMAPX_MAX_QL ? URMAPX_MAX_QL : max_read_len;
	if (max_read_len > URMAPX_MAX_QL) max_read_len = URMAPX_MAX_QL;
	const int cls = max_read_len <= 128 ? 3 : max_read_len <= 192 ? 0 : max_read_len <= 256 ? 2 : max_read_len <= 320 ? 1 : max_read_len <= 512 ? 4 : 5;
	if (C->blocks[cls] == 0) {
		C->blocks[cls] = search_block_count(max_read_len, C->device);
		if (getenv("URMAPX_VERBOSE")) fprintf(stderr, "urmapx: search_se_kernel grid = %d persistent blocks (read class %d)\n", C->blocks[cls], cls);
	}
	if (C->blocks[cls] <= 0) return URMAPX_E_NODEVICE;
	SearchWork wk;
	wk.blocks = C->blocks[cls];
	wk.scratch_stride = search_scratch_stride(max_read_len);
	if ((rc = C->scratch.ensure(wk.scratch_stride * (size_t)wk.blocks + search_scratch_tail(wk.blocks)))) return rc;
	if ((rc = C->ovflist.ensure((size_t)n + 1))) return rc;
	wk.ovf_list = C->ovflist.p;
	const char *ds = getenv("URMAPX_DEBUG_STOP");
	const bool diag = getenv("URMAPX_PHASE_STATS") || ds;
	if ((rc = C->statsbuf.ensure(64 + (diag ? (size_t)n : 0)))) return rc;  // diagnostics: words 64.. = cycles per read
	C->stats_reads = diag ? n : 0;
	wk.scratch = C->scratch.p;
	wk.ticket = C->statsbuf.p + 62;  // words 62/63 of the diagnostics buffer are never touched by the stamps
	wk.ticket3 = C->statsbuf.p + 58; // nor is word 58 (60 is the gather microbenchmark's sink)
	if (const char *e = getenv("URMAPX_TEST_HSP_LDS_CAP")) wk.hsp_lds_cap = atoi(e);
	// diagnostics: URMAPX_PHASE_STATS = per-phase cycle counters; URMAPX_DEBUG_STOP=N = cut the schedule after step N
	// (results are then NOT the reference's).  Words 0/1 of the buffer: stop step, "no timing" flag.
	wk.stats = diag ? C->statsbuf.p : nullptr;
	if (wk.stats) {
		const uint32_t ctl[2] = {ds ? (uint32_t)atoi(ds) : 0u, getenv("URMAPX_PHASE_STATS") ? 0u : 1u};
		HIP_TRY(hipMemcpyAsync(C->statsbuf.p, ctl, 8, hipMemcpyHostToDevice, C->stream));
	}
	// phase 6 (AlignHSP of every remaining HSP) as launches of its own: job array, path slices, parked read states
	if (!getenv("URMAPX_INLINE_PHASE6")) {
		if (C->dp_blocks[cls] == 0) C->dp_blocks[cls] = dp_block_count(max_read_len, C->device);
		if (C->dp_blocks[cls] <= 0) return URMAPX_E_NODEVICE;
		wk.dp_blocks = C->dp_blocks[cls];
		if (C->fin_blocks[cls] == 0) C->fin_blocks[cls] = fin_block_count(max_read_len, C->device);
		wk.fin_blocks = C->fin_blocks[cls] > 0 ? C->fin_blocks[cls] : 0;
		wk.dp_scratch_stride = dp_scratch_stride(max_read_len);
		// phase 6's rounds: three for reads of up to 192 bases, four beyond (kernels.h); URMAPX_DP_BOUNDS="0,2,8,32" (measurement) sets them
		wk.dp_bounds = dp_bounds_default(max_read_len > 192);
		if (const char *e = getenv("URMAPX_DP_BOUNDS")) {
			DpBounds b;
			b.rounds = 0;
			for (const char *c = e; *c && b.rounds < DP_ROUNDS;) {
				char *end;
				const unsigned long v = strtoul(c, &end, 10);
				if (end == c) break;
				b.lo[b.rounds++] = (uint32_t)v;
				c = *end == ',' ? end + 1 : end;
			}
			bool ok = b.rounds >= 1 && b.lo[0] == 0;
			for (int i = 1; i < b.rounds; ++i) ok = ok && b.lo[i] > b.lo[i - 1];
			if (ok) {
				for (int i = b.rounds; i <= DP_ROUNDS; ++i) b.lo[i] = 0xFFFFFFFFu;
				wk.dp_bounds = b;
			}
		}
		C->dp_rounds_used = wk.dp_bounds.rounds;
		for (int i = 0; i <= DP_ROUNDS; ++i) C->dp_bounds_used[i] = wk.dp_bounds.lo[i];
		if ((rc = C->dpscratch.ensure(wk.dp_scratch_stride * (size_t)wk.dp_blocks))) return rc;
		wk.dp_scratch = C->dpscratch.p;
		const uint64_t jc = (uint64_t)n * 16u;  // ~10x what a repeat-rich genome needs; beyond it the search kernel runs phase 6 itself
		const uint32_t jobs_cap[2] = {(uint32_t)(jc < (1ull << 30) ? jc : (1ull << 30)) + 4096u,
		                              (uint32_t)(jc < (1ull << 30) ? jc : (1ull << 30)) + 65536u};
		const uint32_t fin_cap[2] = {n, n / 8u + 1024u};
		size_t need = 256, at[2][7];  // head: counters (2 x 16 bytes), then the work counters and list lengths (2 x 64 bytes from byte 64)
		for (int p = 0; p < 2; ++p) {
			at[p][5] = need; need += (((size_t)jobs_cap[p] * 2) + 63) & ~(size_t)63;
			at[p][0] = need; need += (size_t)jobs_cap[p] * sizeof(DpJob);
			at[p][1] = need; need += (((size_t)jobs_cap[p] * DP_JOB_OPS * 2) + 63) & ~(size_t)63;
			at[p][2] = need; need += (((size_t)fin_cap[p] * 16) + 63) & ~(size_t)63;
			at[p][3] = need; need += (size_t)fin_cap[p] * dp_state_words(p == 1) * 4;
			at[p][4] = 16 * (size_t)p;
			at[p][6] = need; need += (((size_t)jobs_cap[p] * 4 * DP_ROUNDS) + 63) & ~(size_t)63;
		}
		// phase 3 parked (kernels.h: SearchWork::dp3): jobs, their path slices, the parking lot with its larger records.
		// Off by default: measured on the hg38-scale bench it LOSES 6 % at 150 bases and 4 % at 250 (the first launch gets 1.9 ms
		// shorter, phase 3's DP launch and the second search launch cost 3.3 ms).  URMAPX_PARK_PHASE3=1 turns it on (tests, measurement).
		const char *park3 = getenv("URMAPX_PARK_PHASE3");
		size_t p3w = (park3 && atoi(park3) > 0) ? p3_state_words(max_read_len) : 0;
		if (p3w && !C->X.rowinfo) {
			//  the parked variant runs on the row layout, which an index that was uploaded WITHOUT the knob has dropped once slot16
			// was built: say so once instead of silently mapping with the inline kernel and reporting zeros
			static std::atomic<bool> said{false};
			if (!said.exchange(true)) fprintf(stderr, "urmapx: URMAPX_PARK_PHASE3 is set but the index was uploaded without it (its row layout is gone): phase 3 stays inline\n");
			p3w = 0;
		}
		uint32_t jobs_cap3 = (uint32_t)((uint64_t)n * 2u < (1ull << 30) ? (uint64_t)n * 2u : (1ull << 30)) + 4096u;
		uint32_t fin_cap3 = n;
		// test aids: a job array / parking lot too small for the batch -- reads that find no room are mapped by the second pass
		if (const char *e = getenv("URMAPX_TEST_P3_JOBS_CAP")) jobs_cap3 = (uint32_t)atoi(e) + 1u;
		if (const char *e = getenv("URMAPX_TEST_P3_FIN_CAP")) fin_cap3 = std::min<uint32_t>(n, (uint32_t)atoi(e));
		size_t at3[5] = {0, 0, 0, 0, 0};
		if (p3w) {
			at3[4] = need; need += (((size_t)jobs_cap3 * 2) + 63) & ~(size_t)63;
			at3[0] = need; need += (size_t)jobs_cap3 * sizeof(DpJob);
			at3[1] = need; need += (((size_t)jobs_cap3 * DP_JOB_OPS * 2) + 63) & ~(size_t)63;
			at3[2] = need; need += (((size_t)fin_cap3 * 16) + 63) & ~(size_t)63;
			at3[3] = need; need += (size_t)fin_cap3 * p3w * 4;
		}
		if ((rc = C->dpbuf.ensure(need))) return rc;
		C->p3_parked = p3w != 0;
		if (p3w) {
			DpWork &d = wk.dp3;
			d.jobs = reinterpret_cast<DpJob *>(C->dpbuf.p + at3[0]);
			d.ops = reinterpret_cast<uint16_t *>(C->dpbuf.p + at3[1]);
			d.kidx = reinterpret_cast<uint16_t *>(C->dpbuf.p + at3[4]);
			d.fin_list = reinterpret_cast<uint32_t *>(C->dpbuf.p + at3[2]);
			d.state = reinterpret_cast<uint32_t *>(C->dpbuf.p + at3[3]);
			d.counters = reinterpret_cast<uint32_t *>(C->dpbuf.p + 32);
			d.tickets = reinterpret_cast<uint32_t *>(C->dpbuf.p + 64 + 4 * DP_TICKET_WORDS * (size_t)2);
			d.jobs_cap = jobs_cap3; d.fin_cap = fin_cap3;
		}
		for (int p = 0; p < 2; ++p) {
			DpWork &d = wk.dp[p];
			d.jobs = reinterpret_cast<DpJob *>(C->dpbuf.p + at[p][0]);
			d.ops = reinterpret_cast<uint16_t *>(C->dpbuf.p + at[p][1]);
			d.kidx = reinterpret_cast<uint16_t *>(C->dpbuf.p + at[p][5]);
			d.round_list = reinterpret_cast<uint32_t *>(C->dpbuf.p + at[p][6]);
			d.fin_list = reinterpret_cast<uint32_t *>(C->dpbuf.p + at[p][2]);
			d.state = reinterpret_cast<uint32_t *>(C->dpbuf.p + at[p][3]);
			d.counters = reinterpret_cast<uint32_t *>(C->dpbuf.p + at[p][4]);
			d.tickets = reinterpret_cast<uint32_t *>(C->dpbuf.p + 64 + 4 * DP_TICKET_WORDS * (size_t)p);
			d.jobs_cap = jobs_cap[p]; d.fin_cap = fin_cap[p];
		}
	}
	// stage stamps: one event per launch group (urmapx_ctx_stage_ms); URMAPX_NO_STAGE_STAMPS=1 leaves them out (measurement
	// of what the stamps themselves cost: DESIGN.md 5.0)
	const bool stamps = getenv("URMAPX_NO_STAGE_STAMPS") == nullptr;
	wk.stage_events = stamps ? C->stage_ev : nullptr;
	HIP_TRY(hipMemsetAsync(d_path_used, 0, 4, C->stream));
	// seed + probe run inside the search kernel (kernels.hip): ev[0]..ev[1] brackets nothing for a single-end batch
	HIP_TRY(hipEventRecord(C->ev[0], C->stream));
	HIP_TRY(hipEventRecord(C->ev[1], C->stream));
	C->stage_valid = stamps && n > 0;  // an empty batch records no stage events (launch_search_se returns at once)
	// the general kernel's lists are sized before anything is enqueued: growing them (hipMalloc / hipFree) between the two
	// launches would stall the stream in the middle of a batch
	const int sblocks = slow_qcap <= URMAPX_MAX_QL ? 128 : 64;
	if ((rc = C->slowscratch.ensure(slow_scratch_stride(slow_qcap) * (size_t)sblocks))) return rc;
	if ((rc = C->slowlist.ensure((size_t)n + 2))) return rc;
	HIP_TRY(launch_search_se(C->X, C->params, (const uint8_t *)d_bases, (const uint64_t *)d_offs, n, max_read_len,
	                         (urmapx_result *)d_results, (urmapx_path_op *)d_path_ops, (uint32_t *)d_path_used, wk, C->stream));
	if (!getenv("URMAPX_TEST_NO_GENERAL")) {  // reads outside the fast kernels' domain: the general kernel (it finds its work list on the device; usually empty).  The test aid leaves such reads flagged: which reads do the fast kernels hand over?
		const uint64_t pcap = (uint64_t)n * URMAPX_MAX_PATH_OPS;
		HIP_TRY(launch_search_se_slow(C->X, C->params, (const uint8_t *)d_bases, (const uint64_t *)d_offs, n, slow_qcap,
		                              (urmapx_result *)d_results, (urmapx_path_op *)d_path_ops, (uint32_t *)d_path_used,
		                              (uint32_t)(pcap > 0xFFFFFFFFull ? 0xFFFFFFFFull : pcap), C->slowscratch.p, sblocks, C->slowlist.p + 1,
		                              C->slowlist.p, C->stream));
	}
	if (stamps) HIP_TRY(hipEventRecord(C->stage_ev[STAGE_LAST], C->stream));
	HIP_TRY(hipEventRecord(C->ev[2], C->stream));
	C->ev_valid = true;
	return URMAPX_OK;
}

static uint32_t max_len(const uint64_t *offs, uint32_t n) {
	uint64_t m = 0;
	for (uint32_t i = 0; i < n; ++i) m = offs[i + 1] - offs[i] > m ? offs[i + 1] - offs[i] : m;
	return m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m;
}

int urmapx_map_se(urmapx_ctx *C, const uint8_t *bases, const uint64_t *offs, uint32_t n, urmapx_result *results,
                  urmapx_path_op *path_ops, size_t path_cap, size_t *path_used) {
	if (!C || (n && (!bases || !offs || !results))) return URMAPX_E_ARG;
	if (path_used) *path_used = 0;
	if (n == 0) return URMAPX_OK;
	HIP_TRY(hipSetDevice(C->device));
	const uint64_t total = offs[n];
	uint32_t mx = max_len(offs, n);
	// reads longer than the device domain are flagged per read by the kernel; size the kernels for the cap
	if (mx > URMAPX_MAX_QL_SLOW) mx = URMAPX_MAX_QL_SLOW;
	int rc;
	if ((rc = C->bases.ensure(total + 64))) return rc;
	if ((rc = C->offs.ensure((size_t)n + 1))) return rc;
	if ((rc = C->results.ensure(n))) return rc;
	if ((rc = C->pathops.ensure((size_t)n * URMAPX_MAX_PATH_OPS))) return rc;
	if ((rc = C->used.ensure(1))) return rc;
	HIP_TRY(hipMemcpyAsync(C->bases.p, bases, total, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->offs.p, offs, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, C->stream));
	rc = urmapx_map_se_device(C, C->bases.p, C->offs.p, n, total, mx, C->results.p, C->pathops.p, C->used.p);
	if (rc) return rc;
	uint32_t used = 0;
	HIP_TRY(hipMemcpyAsync(results, C->results.p, (size_t)n * sizeof(urmapx_result), hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(&used, C->used.p, 4, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipStreamSynchronize(C->stream));
	if (used > path_cap || (used && !path_ops)) return URMAPX_E_ARG;
	if (used) HIP_TRY(hipMemcpy(path_ops, C->pathops.p, (size_t)used * sizeof(urmapx_path_op), hipMemcpyDeviceToHost));
	if (path_used) *path_used = used;
	for (uint32_t i = 0; i < n; ++i)
		if (results[i].status) return URMAPX_E_UNSUPPORTED;
	return URMAPX_OK;
}

int urmapx_ctx_set_pair_info(urmapx_ctx *C, int on) {
	if (!C) return URMAPX_E_ARG;
	C->pair_info_on = on ? 1 : 0;
	return URMAPX_OK;
}

int urmapx_ctx_get_pair_info(urmapx_ctx *C, urmapx_pair_info *out, uint32_t npairs) {
	if (!C || !out || npairs > C->pairinfo_n || !C->pairinfo.p) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(C->device));
	HIP_TRY(hipStreamSynchronize(C->stream));
	HIP_TRY(hipMemcpy(out, C->pairinfo.p, (size_t)npairs * sizeof(urmapx_pair_info), hipMemcpyDeviceToHost));
	return URMAPX_OK;
}

// cmd_map2's `-veryfast`: State2::m_Method = 5 (Search5, search2m5.cpp) with band radius 4 (map2.cpp:47-49,17-21)
int urmapx_ctx_set_pe_veryfast(urmapx_ctx *C, int on) {
	if (!C) return URMAPX_E_ARG;
	C->pe_veryfast = on ? 1 : 0;
	return URMAPX_OK;
}

// State2::Search over pairs already resident in HBM (reads 2i, 2i+1 = mates of pair i); asynchronous on the ctx stream.
int urmapx_map_pe_device(urmapx_ctx *C, const void *d_bases, const void *d_offs, uint32_t npairs, uint64_t total_bases,
                         uint32_t max_read_len, void *d_results, void *d_path_ops, void *d_path_used) {
	if (!C || (npairs && (!d_bases || !d_offs || !d_results || !d_path_ops || !d_path_used))) return URMAPX_E_ARG;
	if (max_read_len > MAX_QL_PE || npairs > 0x7FFFFFFFu) return URMAPX_E_UNSUPPORTED;
	HIP_TRY(hipSetDevice(C->device));
	int rc;
	(void)total_bases;
	C->stage_valid = false;  // the stage stamps are a single-end batch's: none of an earlier batch may be reported for this one
	const int cls = max_read_len <= 128 ? 3 : max_read_len <= 192 ? 0 : (max_read_len <= 256 ? 2 : 1);
	if (C->pe_blocks[cls] == 0) C->pe_blocks[cls] = search_pe_block_count(max_read_len, C->device);
	if (C->pe_blocks[cls] <= 0) return URMAPX_E_NODEVICE;
	SearchWork wk;
	wk.blocks = C->pe_blocks[cls];
	wk.scratch_stride = search_pe_scratch_stride(max_read_len);
	if ((rc = C->pe_scratch.ensure(wk.scratch_stride * (size_t)wk.blocks + search_pe_scratch_tail(wk.blocks)))) return rc;
	if ((rc = C->ovflist.ensure(2 * ((size_t)npairs + 1)))) return rc;  // work lists of the second and third pass
	if ((rc = C->slowscratch.ensure(pe_slow_scratch_stride() * (size_t)PE_SLOW_BLOCKS))) return rc;  // the general pair kernel's lists
	if ((rc = C->slowlist.ensure((size_t)npairs + 2))) return rc;
	wk.ovf_list = C->ovflist.p;
	if (const char *e = getenv("URMAPX_TEST_HSP_LDS_CAP")) wk.hsp_lds_cap = atoi(e);
	wk.scratch = C->pe_scratch.p;
	wk.stats = nullptr;
	if ((rc = C->statsbuf.ensure(64))) return rc;
	wk.ticket = C->statsbuf.p + 62;
	HIP_TRY(hipMemsetAsync(d_path_used, 0, 4, C->stream));
	// seed + probe run inside search_pe_kernel (round 3; round 4: without probe arrays in HBM); the two stamps bracket
	// nothing and stay for urmapx_ctx_stage_ms
	HIP_TRY(hipEventRecord(C->ev[0], C->stream));
	HIP_TRY(hipEventRecord(C->ev[1], C->stream));
	urmapx_params Ppe = C->params;
	if (C->pe_veryfast) Ppe.band_radius = 4;  // map2.cpp:17-21
	urmapx_pair_info *d_info = nullptr;
	if (C->pair_info_on) {
		if ((rc = C->pairinfo.ensure(npairs))) return rc;
		d_info = C->pairinfo.p;
		C->pairinfo_n = npairs;
	}
	HIP_TRY(launch_search_pe(C->X, Ppe, (const uint8_t *)d_bases, (const uint64_t *)d_offs, npairs, max_read_len,
	                         (urmapx_result *)d_results, (urmapx_path_op *)d_path_ops, (uint32_t *)d_path_used, wk,
	                         C->pe_veryfast | (getenv("URMAPX_DEBUG_STOP_PE") ? atoi(getenv("URMAPX_DEBUG_STOP_PE")) << 8 : 0), d_info, C->stream));  // bits 8..: diagnostic schedule cut
	{  // pairs outside the fast passes' domain: the general pair kernel (it finds its work list on the device; usually empty)
		const uint64_t pcap = (uint64_t)npairs * 2 * URMAPX_MAX_PATH_OPS;
		HIP_TRY(launch_search_pe_slow(C->X, Ppe, (const uint8_t *)d_bases, (const uint64_t *)d_offs, npairs, (urmapx_result *)d_results,
		                              (urmapx_path_op *)d_path_ops, (uint32_t *)d_path_used, (uint32_t)(pcap > 0xFFFFFFFFull ? 0xFFFFFFFFull : pcap),
		                              C->slowscratch.p, PE_SLOW_BLOCKS, C->slowlist.p + 1, C->slowlist.p, C->pe_veryfast, d_info,
		                              getenv("URMAPX_TEST_PE_GENERAL") != nullptr, C->stream));  // test aid: every pair through the general kernel
	}
	HIP_TRY(hipEventRecord(C->ev[2], C->stream));
	C->ev_valid = true;
	return URMAPX_OK;
}

// State2::Search (search2.cpp:59-73, method 4) over a batch of read pairs held in HOST memory: reads 2i and 2i+1 are
// the mates of pair i.  results[2*npairs].
int urmapx_map_pe(urmapx_ctx *C, const uint8_t *bases, const uint64_t *offs, uint32_t npairs, urmapx_result *results,
                  urmapx_path_op *path_ops, size_t path_cap, size_t *path_used) {
	if (!C || (npairs && (!bases || !offs || !results))) return URMAPX_E_ARG;
	if (path_used) *path_used = 0;
	if (npairs == 0) return URMAPX_OK;
	if (npairs > 0x7FFFFFFFu) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(C->device));
	const uint32_t n = 2 * npairs;
	const uint64_t total = offs[n];
	uint32_t mx = max_len(offs, n);
	if (mx > MAX_QL_PE) mx = MAX_QL_PE;
	int rc;
	if ((rc = C->bases.ensure(total + 64))) return rc;
	if ((rc = C->offs.ensure((size_t)n + 1))) return rc;
	if ((rc = C->results.ensure(n))) return rc;
	if ((rc = C->pathops.ensure((size_t)n * URMAPX_MAX_PATH_OPS))) return rc;
	if ((rc = C->used.ensure(1))) return rc;
	HIP_TRY(hipMemcpyAsync(C->bases.p, bases, total, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->offs.p, offs, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, C->stream));
	if ((rc = urmapx_map_pe_device(C, C->bases.p, C->offs.p, npairs, total, mx, C->results.p, C->pathops.p, C->used.p))) return rc;
	uint32_t used = 0;
	HIP_TRY(hipMemcpyAsync(results, C->results.p, (size_t)n * sizeof(urmapx_result), hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(&used, C->used.p, 4, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipStreamSynchronize(C->stream));
	if (used > path_cap || (used && !path_ops)) return URMAPX_E_ARG;
	if (used) HIP_TRY(hipMemcpy(path_ops, C->pathops.p, (size_t)used * sizeof(urmapx_path_op), hipMemcpyDeviceToHost));
	if (path_used) *path_used = used;
	for (uint32_t i = 0; i < n; ++i)
		if (results[i].status) return URMAPX_E_UNSUPPORTED;
	return URMAPX_OK;
}

int urmapx_seed_probe(urmapx_ctx *C, const uint8_t *bases, const uint64_t *offs, uint32_t n, uint64_t *slots,
                      uint8_t *tallies, uint32_t *positions) {
	if (!C || (n && (!bases || !offs || !slots || !tallies || !positions))) return URMAPX_E_ARG;
	if (n == 0) return URMAPX_OK;
	HIP_TRY(hipSetDevice(C->device));
	const uint64_t total = offs[n];
	uint32_t mx = max_len(offs, n);
	if (mx > URMAPX_MAX_QL) return URMAPX_E_UNSUPPORTED;
	int rc;
	if ((rc = C->bases.ensure(total + 64))) return rc;
	if ((rc = C->offs.ensure((size_t)n + 1))) return rc;
	if ((rc = ensure_probe(C, total))) return rc;
	HIP_TRY(hipMemcpyAsync(C->bases.p, bases, total, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->offs.p, offs, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, C->stream));
	// entries the kernel does not write (positions > L-W) read back as "no k-mer"
	HIP_TRY(hipMemsetAsync(C->slots.p, 0xFF, 2 * total * 8, C->stream));
	HIP_TRY(hipMemsetAsync(C->tallies.p, 0, 2 * total, C->stream));
	HIP_TRY(hipMemsetAsync(C->positions.p, 0xFF, 2 * total * 4, C->stream));
	ProbeOut po{C->slots.p, C->tallies.p, C->positions.p};
	HIP_TRY(launch_seed_probe(C->X, C->bases.p, C->offs.p, n, mx, po, C->stream));
	HIP_TRY(hipMemcpyAsync(slots, C->slots.p, 2 * total * 8, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(tallies, C->tallies.p, 2 * total, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(positions, C->positions.p, 2 * total * 4, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipStreamSynchronize(C->stream));
	return URMAPX_OK;
}

// The probe stage alone over reads resident in HBM (measurement: the figure the task's north_star names, "rocprof HBM GB/s on the
// probe kernel against peak"; inside a mapping call the probe is a stage of the search kernel).  Slots, tallies and positions go
// to the context's probe arrays as in urmapx_seed_probe; *ms = the launch on the context's stream, by events.
int urmapx_seed_probe_device(urmapx_ctx *C, const void *d_bases, const void *d_offs, uint32_t n, uint64_t total_bases, uint32_t max_read_len,
                             float *ms) {
	if (!C || !ms || (n && (!d_bases || !d_offs))) return URMAPX_E_ARG;
	*ms = 0;
	if (n == 0) return URMAPX_OK;
	if (max_read_len > URMAPX_MAX_QL) return URMAPX_E_UNSUPPORTED;
	HIP_TRY(hipSetDevice(C->device));
	int rc;
	if ((rc = ensure_probe(C, total_bases))) return rc;
	ProbeOut po{C->slots.p, C->tallies.p, C->positions.p};
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, C->stream));
	hipError_t e = launch_seed_probe(C->X, (const uint8_t *)d_bases, (const uint64_t *)d_offs, n, max_read_len, po, C->stream);
	if (e == hipSuccess) e = hipEventRecord(e1, C->stream);
	if (e == hipSuccess) e = hipEventSynchronize(e1);
	if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return hip_rc(e);
}

// measurement aid for the roofline: see include/urmapx.h
int urmapx_ctx_gather_microbench(urmapx_ctx *C, uint64_t n_loads, double *loads_per_s) {
	if (!C || !loads_per_s || n_loads == 0) return URMAPX_E_ARG;
	HIP_TRY(hipSetDevice(C->device));
	int rc;
	if ((rc = C->statsbuf.ensure(64))) return rc;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, C->device));
	const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 8u;  // 2048 threads per CU
	uint64_t per_iter = (uint64_t)blocks * 256u * 8u;
	uint32_t iters = (uint32_t)((n_loads + per_iter - 1) / per_iter);
	if (iters == 0) iters = 1;
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(launch_gather_bench(C->X, blocks, 1, C->statsbuf.p + 60, C->stream));  // warm-up
	HIP_TRY(hipEventRecord(e0, C->stream));
	HIP_TRY(launch_gather_bench(C->X, blocks, iters, C->statsbuf.p + 60, C->stream));
	HIP_TRY(hipEventRecord(e1, C->stream));
	HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*loads_per_s = ms > 0 ? (double)per_iter * iters / (ms * 1e-3) : 0.0;
	return URMAPX_OK;
}

int urmapx_viterbi_batch(urmapx_ctx *C, const uint8_t *a, const uint32_t *a_offs, const uint8_t *b, const uint32_t *b_offs,
                         const uint8_t *flags, uint32_t n, float *scores, uint8_t *status, urmapx_path_op *ops,
                         uint16_t *nops) {
	if (!C || (n && (!a_offs || !b_offs || !flags || !scores || !status || !ops || !nops))) return URMAPX_E_ARG;
	if (n == 0) return URMAPX_OK;
	HIP_TRY(hipSetDevice(C->device));
	const size_t ta = a_offs[n], tb = b_offs[n];
	int rc;
	if ((rc = C->va.ensure(ta + 64))) return rc;
	if ((rc = C->vb.ensure(tb + 64))) return rc;
	if ((rc = C->vaoffs.ensure((size_t)n + 1))) return rc;
	if ((rc = C->vboffs.ensure((size_t)n + 1))) return rc;
	if ((rc = C->vflags.ensure(n))) return rc;
	if ((rc = C->vstatus.ensure(n))) return rc;
	if ((rc = C->vscores.ensure(n))) return rc;
	if ((rc = C->vnops.ensure(n))) return rc;
	if ((rc = C->vops.ensure((size_t)n * URMAPX_MAX_PATH_OPS))) return rc;
	if ((rc = C->vscratch.ensure((size_t)n * viterbi_batch_scratch_stride()))) return rc;
	if (ta) HIP_TRY(hipMemcpyAsync(C->va.p, a, ta, hipMemcpyHostToDevice, C->stream));
	if (tb) HIP_TRY(hipMemcpyAsync(C->vb.p, b, tb, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->vaoffs.p, a_offs, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->vboffs.p, b_offs, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(hipMemcpyAsync(C->vflags.p, flags, n, hipMemcpyHostToDevice, C->stream));
	HIP_TRY(launch_viterbi_batch(C->params, C->va.p, C->vaoffs.p, C->vb.p, C->vboffs.p, C->vflags.p, n, C->vscores.p,
	                             C->vstatus.p, C->vops.p, C->vnops.p, C->vscratch.p, C->stream));
	HIP_TRY(hipMemcpyAsync(scores, C->vscores.p, (size_t)n * 4, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(status, C->vstatus.p, n, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(nops, C->vnops.p, (size_t)n * 2, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipMemcpyAsync(ops, C->vops.p, (size_t)n * URMAPX_MAX_PATH_OPS * 2, hipMemcpyDeviceToHost, C->stream));
	HIP_TRY(hipStreamSynchronize(C->stream));
	return URMAPX_OK;
}

}  // extern "C"
