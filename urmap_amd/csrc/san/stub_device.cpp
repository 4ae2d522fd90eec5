// stub_device.cpp -- the DEVICE side of the C ABI as a host stand-in, for the sanitizer builds only (make san).
//
// The host side of the drop-in path -- pipeline.cpp (cmd_map / cmd_map2 as reader -> lanes -> writer threads, the chunk cutter,
// the shard cutter, the SAM writer), sam.cpp (FASTQ reader, SAM / tab formatters), pgzip.cpp (the parallel gzip reader),
// make_ufi.cpp (the host index builder) -- is 5 000 lines of threaded C++ that parse untrusted input.  `make san` compiles those
// translation units UNCHANGED with -fsanitize=address,undefined and again with -fsanitize=thread, and links them against this
// file instead of the HIP translation units: a "lane" here maps a chunk on the host with a pure function of the read (no GPU,
// no search), so that urmapx_map_files itself -- its queues, buffers, offsets and error paths -- runs under the sanitizers in
// the CPU suite (tests/test_sanitizers_cpu.py).  Nothing of this is linked into liburmapx.so.
//
// The stand-in "mapping": a read is unmapped unless URX_STUB_MAP=1, in which case a read whose first base is A, C or G gets a
// position, strand and MAPQ that are a hash of its bases (A and C on the plus strand).  Text stage and host stage use the same
// function, so the SAM of a run must not depend on which road a chunk took, on the chunk size, the lane count or the shard count.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../../include/urmapx.h"
#include "../internal.h"
#include "../sam.h"

struct urmapx_index {
	std::vector<std::string> labels;
	std::vector<uint32_t> lengths, offsets;
};
struct urmapx_ctx {
	const urmapx_index *I = nullptr;
	bool pair_info = false;
	std::vector<urmapx_pair_info> info;
};
struct urmapx_text {
	urmapx_ctx *C = nullptr;
	bool deferred = false;
	struct Chunk { std::string text; char *dst; urmapx_text_report rep; };
	std::deque<Chunk> flying;   // deferred: text made, "copy" not done until urmapx_text_wait
	std::string waiting;        // after URMAPX_TEXT_SAM_CAP
	urmapx_text_report waiting_rep;
	bool have_waiting = false;
	unsigned chunk_nr = 0;
	// -tabbedout
	std::vector<urmapx_result> pair_res;
	std::vector<uint32_t> line_ends1, lens2;
};

// ---- the HIP runtime calls pipeline.cpp makes (page-locked buffers, device selection, NUMA lookup) ----
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
	*p = malloc(n ? n : 1);
	return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorInvalidDevice; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)200 << 30; *t = (size_t)288 << 30; return hipSuccess; }
}

namespace urx {
AllocClock &alloc_clock() { static AllocClock c{}; return c; }
}

static uint64_t fnv(const uint8_t *p, size_t n) {
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
	return h;
}
static bool stub_maps() { static const bool on = getenv("URX_STUB_MAP") != nullptr; return on; }

// the stand-in for State1::Search: a pure function of the read's bases
static void stub_search(const urmapx_index *I, const uint8_t *seq, uint32_t L, urmapx_result *r) {
	memset(r, 0, sizeof *r);
	r->dbpos = 0xFFFFFFFFu;
	if (!stub_maps() || L == 0 || I->labels.empty()) return;
	const uint8_t c = seq[0];
	if (c != 'A' && c != 'C' && c != 'G') return;
	const uint64_t h = fnv(seq, L);
	r->seq_index = (uint32_t)(h % I->labels.size());
	const uint32_t len = I->lengths[r->seq_index];
	r->coord = len > L ? (uint32_t)((h >> 8) % (len - L)) : 0;
	r->dbpos = I->offsets[r->seq_index] + r->coord;
	r->plus = c != 'G';
	r->mapq = (uint8_t)((h >> 40) % 41);
	r->score = (int16_t)L;
}

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
int urmapx_params_for_method(unsigned method, urmapx_params *p) {
	if (!p || (method != 6 && method != 7 && method != 8)) return URMAPX_E_ARG;
	*p = method == 7 ? urmapx_params{-4, -6, -2, 35, 35, 12, 75, 8, 6, 5, 8} : urmapx_params{-3, -5, -1, 20, 60, 9, 100, 1, 1, 1, 12};
	return URMAPX_OK;
}
// a directory without a table: what the SAM header and the records' RNAME need
urmapx_index *urx_stub_index(uint32_t n, const uint32_t *lengths, const char *const *labels) {
	urmapx_index *I = new urmapx_index;
	uint32_t off = 0;
	for (uint32_t i = 0; i < n; ++i) {
		I->labels.push_back(labels[i]);
		I->lengths.push_back(lengths[i]);
		I->offsets.push_back(off);
		off += lengths[i] + 32;
	}
	return I;
}
int urmapx_index_upload(urmapx_index *, int) { return URMAPX_OK; }
int urmapx_index_replicate(const urmapx_index *src, int, urmapx_index **out) { *out = new urmapx_index(*src); return URMAPX_OK; }
void urmapx_index_close(urmapx_index *I) { urx::lane_pool_purge(I); delete I; }
uint32_t urmapx_index_seq_count(const urmapx_index *I) { return (uint32_t)I->labels.size(); }
const char *urmapx_index_label(const urmapx_index *I, uint32_t i) { return i < I->labels.size() ? I->labels[i].c_str() : nullptr; }
uint32_t urmapx_index_seq_length(const urmapx_index *I, uint32_t i) { return i < I->lengths.size() ? I->lengths[i] : 0; }
uint32_t urmapx_index_seq_offset(const urmapx_index *I, uint32_t i) { return i < I->offsets.size() ? I->offsets[i] : 0; }

int urmapx_ctx_create(const urmapx_index *I, int, const urmapx_params *, urmapx_ctx **out) {
	*out = new urmapx_ctx;
	(*out)->I = I;
	return URMAPX_OK;
}
void urmapx_ctx_destroy(urmapx_ctx *C) { delete C; }
int urmapx_ctx_set_pe_veryfast(urmapx_ctx *, int) { return URMAPX_OK; }
int urmapx_ctx_set_pair_info(urmapx_ctx *C, int on) { C->pair_info = on != 0; return URMAPX_OK; }
static void stub_pair_info(const urmapx_result *r, urmapx_pair_info *o) {
	memset(o, 0, sizeof *o);
	for (int k = 0; k < 2; ++k) {
		o->top_db[k] = r[k].dbpos; o->second_db[k] = 0xFFFFFFFFu;
		o->top_score[k] = r[k].score; o->top_plus[k] = r[k].plus;
	}
}
int urmapx_ctx_get_pair_info(urmapx_ctx *C, urmapx_pair_info *out, uint32_t npairs) {
	if (npairs > C->info.size()) return URMAPX_E_ARG;
	memcpy(out, C->info.data(), npairs * sizeof *out);
	return URMAPX_OK;
}
int urmapx_map_se(urmapx_ctx *C, const uint8_t *bases, const uint64_t *offs, uint32_t n, urmapx_result *results, urmapx_path_op *, size_t, size_t *used) {
	for (uint32_t i = 0; i < n; ++i) stub_search(C->I, bases + offs[i], (uint32_t)(offs[i + 1] - offs[i]), &results[i]);
	if (used) *used = 0;
	return URMAPX_OK;
}
int urmapx_map_pe(urmapx_ctx *C, const uint8_t *bases, const uint64_t *offs, uint32_t npairs, urmapx_result *results, urmapx_path_op *, size_t, size_t *used) {
	for (uint32_t i = 0; i < 2 * npairs; ++i) stub_search(C->I, bases + offs[i], (uint32_t)(offs[i + 1] - offs[i]), &results[i]);
	if (C->pair_info) {
		C->info.resize(npairs);
		for (uint32_t p = 0; p < npairs; ++p) stub_pair_info(&results[2 * p], &C->info[p]);
	}
	if (used) *used = 0;
	return URMAPX_OK;
}

// ---- text stage: a chunk of FASTQ bytes -> the bytes of its SAM records, with the device parser's hand-back rules ----
int urmapx_text_create(urmapx_ctx *C, urmapx_text **out) {
	*out = new urmapx_text;
	(*out)->C = C;
	return URMAPX_OK;
}
void urmapx_text_destroy(urmapx_text *T) { delete T; }
}

namespace {
struct Rec { const char *label; size_t label_n; const uint8_t *seq, *qual; uint32_t L; uint32_t ends[4]; };
// 0, or the URMAPX_TEXT_* reason the device parser would hand the chunk back with
unsigned parse_chunk(const char *p, size_t n, std::vector<Rec> &recs) {
	if (n > (1u << 30)) return URMAPX_TEXT_TOO_LARGE;
	if (memchr(p, '\r', n)) return URMAPX_TEXT_CR;
	if (n == 0 || p[n - 1] != '\n') return URMAPX_TEXT_RAGGED;
	std::vector<size_t> ends;
	for (const char *c = p, *e = p + n; c < e;) {
		const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
		ends.push_back((size_t)(nl - p));
		c = nl + 1;
	}
	if (ends.size() % 4) return URMAPX_TEXT_RAGGED;
	for (size_t k = 0; k < ends.size(); k += 4) {
		const size_t s0 = k ? ends[k - 1] + 1 : 0, s1 = ends[k] + 1, s2 = ends[k + 1] + 1, s3 = ends[k + 2] + 1;
		Rec r;
		if (ends[k] == s0 || p[s0] != '@') return URMAPX_TEXT_BAD_RECORD;
		if (ends[k + 2] == s2 || p[s2] != '+') return URMAPX_TEXT_BAD_RECORD;
		r.label = p + s0 + 1; r.label_n = ends[k] - s0 - 1;
		r.seq = (const uint8_t *)p + s1; r.L = (uint32_t)(ends[k + 1] - s1);
		r.qual = (const uint8_t *)p + s3;
		if (r.L == 0 || ends[k + 3] - s3 != r.L) return URMAPX_TEXT_BAD_RECORD;
		for (uint32_t i = 0; i < r.L; ++i)
			if (!isalpha(r.seq[i])) return URMAPX_TEXT_BAD_RECORD;
		for (int q = 0; q < 4; ++q) r.ends[q] = (uint32_t)ends[k + (size_t)q];
		recs.push_back(r);
	}
	return 0;
}
void count(urmapx_text_report *rep, const urmapx_result &r, unsigned minq) {
	if (r.dbpos == 0xFFFFFFFFu) ++rep->unmapped;
	else if (r.mapq >= minq) ++rep->mapped_q;
	else ++rep->mapped_lowq;
}
// hands the text over as the device stage does: at once, later (deferred), or not yet (the caller's buffer is too small)
int deliver(urmapx_text *T, std::string &&text, char *sam, size_t cap, urmapx_text_report *rep) {
	rep->sam_bytes = text.size();
	++T->chunk_nr;
	const char *force = getenv("URX_STUB_FORCE_SAM_CAP");  // every N-th chunk pretends the buffer was too small
	if (text.size() > cap || (force && atoi(force) > 0 && T->chunk_nr % (unsigned)atoi(force) == 0 && !T->have_waiting)) {
		T->waiting = std::move(text); T->waiting_rep = *rep; T->have_waiting = true;
		rep->reason = URMAPX_TEXT_SAM_CAP;
		return URMAPX_OK;
	}
	if (T->deferred) {
		if (T->flying.size() >= 2) return URMAPX_E_ARG;
		rep->reason = URMAPX_TEXT_DEFERRED;
		T->flying.push_back(urmapx_text::Chunk{std::move(text), sam, *rep});
		return URMAPX_OK;
	}
	memcpy(sam, text.data(), text.size());
	return URMAPX_OK;
}
}  // namespace

extern "C" {
int urmapx_text_map_se(urmapx_text *T, const char *fastq, size_t n, unsigned minq, char *sam, size_t cap, urmapx_text_report *rep) {
	memset(rep, 0, sizeof *rep);
	std::vector<Rec> recs;
	if ((rep->reason = parse_chunk(fastq, n, recs)) != 0) return URMAPX_OK;
	std::string text, label;
	for (const Rec &q : recs) {
		urmapx_result r;
		stub_search(T->C->I, q.seq, q.L, &r);
		count(rep, r, minq);
		label.assign(q.label, q.label_n);
		urx::append_sam_record(text, T->C->I, r, nullptr, 0, "*", 0xFFFFFFFFu, 0, label.c_str(), q.seq, q.qual, q.L);
	}
	rep->records = (uint32_t)recs.size();
	return deliver(T, std::move(text), sam, cap, rep);
}
int urmapx_text_map_pe(urmapx_text *T, const char *fq1, size_t n1, const char *fq2, size_t n2, unsigned minq, char *sam, size_t cap,
                       urmapx_text_report *rep) {
	memset(rep, 0, sizeof *rep);
	std::vector<Rec> a, b;
	if ((rep->reason = parse_chunk(fq1, n1, a)) != 0) return URMAPX_OK;
	if ((rep->reason = parse_chunk(fq2, n2, b)) != 0) return URMAPX_OK;
	if (a.size() != b.size()) { rep->reason = URMAPX_TEXT_UNEQUAL; return URMAPX_OK; }
	std::string text, l1, l2;
	std::vector<char> buf;
	T->pair_res.resize(2 * a.size());
	T->line_ends1.clear(); T->lens2.clear();
	T->C->info.resize(a.size());
	for (size_t i = 0; i < a.size(); ++i) {
		urmapx_result *r = &T->pair_res[2 * i];
		stub_search(T->C->I, a[i].seq, a[i].L, &r[0]);
		stub_search(T->C->I, b[i].seq, b[i].L, &r[1]);
		count(rep, r[0], minq); count(rep, r[1], minq);
		stub_pair_info(r, &T->C->info[i]);
		l1.assign(a[i].label, a[i].label_n); l2.assign(b[i].label, b[i].label_n);
		buf.resize(l1.size() + l2.size() + 3 * (size_t)(a[i].L + b[i].L) + 2048);
		const size_t k = urmapx_sam_pe(T->C->I, &r[0], &r[1], nullptr, l1.c_str(), a[i].seq, a[i].qual, a[i].L, l2.c_str(), b[i].seq, b[i].qual, b[i].L,
		                               buf.data(), buf.size());
		text.append(buf.data(), k);
		for (int q = 0; q < 4; ++q) T->line_ends1.push_back(a[i].ends[q]);
		T->lens2.push_back(b[i].L);
	}
	rep->records = (uint32_t)(2 * a.size());
	return deliver(T, std::move(text), sam, cap, rep);
}
int urmapx_text_set_deferred(urmapx_text *T, int on) {
	if (!T->flying.empty()) return URMAPX_E_ARG;
	T->deferred = on != 0;
	return URMAPX_OK;
}
int urmapx_text_wait(urmapx_text *T, urmapx_text_report *rep) {
	if (T->flying.empty()) return URMAPX_E_ARG;
	urmapx_text::Chunk &c = T->flying.front();
	memcpy(c.dst, c.text.data(), c.text.size());  // only now: a caller that reads `sam` before the wait reads stale bytes
	*rep = c.rep;
	rep->reason = 0;
	T->flying.pop_front();
	return URMAPX_OK;
}
int urmapx_text_fetch_sam(urmapx_text *T, char *sam, size_t cap, urmapx_text_report *rep) {
	if (!T->have_waiting || cap < T->waiting.size()) return URMAPX_E_ARG;
	*rep = T->waiting_rep;
	rep->reason = 0;
	T->have_waiting = false;
	std::string text = std::move(T->waiting);
	if (T->deferred) {
		if (T->flying.size() >= 2) return URMAPX_E_ARG;
		rep->reason = URMAPX_TEXT_DEFERRED;
		T->flying.push_back(urmapx_text::Chunk{std::move(text), sam, *rep});
		return URMAPX_OK;
	}
	memcpy(sam, text.data(), text.size());
	return URMAPX_OK;
}
int urmapx_text_fetch_pairs(urmapx_text *T, uint32_t npairs, urmapx_result *results, urmapx_pair_info *info, uint32_t *line_ends1, uint32_t *lens2) {
	if ((size_t)2 * npairs != T->pair_res.size()) return URMAPX_E_ARG;
	memcpy(results, T->pair_res.data(), T->pair_res.size() * sizeof *results);
	memcpy(info, T->C->info.data(), npairs * sizeof *info);
	memcpy(line_ends1, T->line_ends1.data(), T->line_ends1.size() * 4);
	memcpy(lens2, T->lens2.data(), T->lens2.size() * 4);
	return URMAPX_OK;
}
// make_ufi.cpp's GPU-assisted builder has no device here
int urmapx_build_slots_gpu(int, const uint8_t *, const void *, uint32_t, uint32_t, uint32_t, uint64_t, uint8_t *, uint32_t *) { return URMAPX_E_NODEVICE; }
}
