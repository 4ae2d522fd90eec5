// make_ufi.cpp -- host-side index construction for the command line's -make_ufi (product code).
// Behaviour of cmd_make_ufi / UFIndex::ReadSeqData / CountSlots / CountSlots_Minus / MakeIndex / UpdateSlot /
// FindEndOfList / FindFreeSlot / TruncateSlot / ToFile (ufindexio.cpp:14-49,117-179; ufindex.cpp:83-322,338-408,
// 462-511,945-1000): the insertion pass is order dependent (each overflow position takes the first free
// borrowable slot after its chain's current end), so it stays sequential on the host; slots are hashed a
// block ahead and prefetched so the pass runs at memory-level parallelism instead of one miss at a time.
#include <algorithm>
#include <chrono>
#include <sys/mman.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/urmapx.h"

namespace {

const uint8_t T_FREE = 0, T_END = 127, T_MINE = 128, T_PLUS1 = 254, T_BOTH1 = 255, T_MAXNEXT = 124, T_LONG_MINE = 253,
              T_LONG_OTHER = 125;
const unsigned LINK_LIMIT = 0xffff, PAD = 32;
const uint32_t M1 = 0x55464931u, M2 = 0x55464932u, M3 = 0x55464933u, M5 = 0x55464935u;

inline uint64_t mix(uint64_t h) {
	h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
	return h;
}

struct Table {
	uint8_t *blob;
	uint64_t N;
	uint32_t maxIx;
	uint8_t *nplus = nullptr, *nminus = nullptr;
	unsigned truncated = 0;

	uint8_t tally(uint64_t s) const { return blob[5 * s]; }
	uint32_t pos(uint64_t s) const { uint32_t p; memcpy(&p, blob + 5 * s + 1, 4); return p; }
	void put(uint64_t s, uint8_t t, uint32_t p) { blob[5 * s] = t; memcpy(blob + 5 * s + 1, &p, 4); }

	// (a + b) mod N for a < N; 64-bit division only when the table is smaller than a link step
	uint64_t wrap(uint64_t a, uint64_t b) const {
		uint64_t x = a + b;
		if (x >= N) { x -= N; if (x >= N) x %= N; }
		return x;
	}
	bool advance(uint64_t &s) const {  // one chain link; false at the end
		uint8_t t = tally(s);
		if (t == T_PLUS1 || t == T_BOTH1 || t == T_END) return false;
		if (t == T_LONG_MINE || t == T_LONG_OTHER) {
			uint32_t p = pos(s);
			s = wrap(wrap(s, p & 0xffff), p >> 16);
		} else
			s = wrap(s, t & 127);
		return true;
	}
	unsigned free_after(uint64_t s) const {
		uint64_t c = s;
		for (unsigned i = 1; i < LINK_LIMIT; ++i) {
			if (++c >= N) c = (N > 1) ? c % N : 0;
			uint8_t n = nplus[c];
			if (n > 0 && n <= maxIx) continue;  // a home slot of some indexed word: never lent out
			if (tally(c) == T_FREE) return i;
		}
		return ~0u;
	}
	void wipe(uint64_t head) {
		++truncated;
		uint64_t s = head;
		for (;;) {
			uint64_t nxt = s;
			bool more = advance(nxt);
			put(s, T_FREE, 0xffffffffu);
			if (!more) return;
			s = nxt;
		}
	}
	// UpdateSlot for a position that is NOT the first indexed occurrence of its slot (so its plus count is >= 2 and
	// both counts are <= MaxIx): the slot is FREE only after a TruncateSlot, and then it restarts as PLUS1.
	void insert(uint64_t slot, uint32_t p) {
		if (tally(slot) == T_FREE) { put(slot, T_PLUS1, p); return; }
		uint64_t eol = slot;
		while (advance(eol)) {}
		unsigned s1 = free_after(eol);
		if (s1 == ~0u) { wipe(slot); return; }
		uint64_t f1 = wrap(eol, s1);
		if (s1 <= T_MAXNEXT) {
			blob[5 * eol] = (uint8_t)((tally(eol) & T_MINE) | s1);
			put(f1, T_END, p);
			return;
		}
		unsigned s2 = free_after(f1);
		if (s2 == ~0u) { wipe(slot); return; }
		uint64_t f2 = wrap(f1, s2);
		uint32_t eolpos = pos(eol);
		put(eol, eol == slot ? T_LONG_MINE : T_LONG_OTHER, s1 | (s2 << 16));
		put(f1, T_LONG_OTHER, eolpos);
		put(f2, T_END, p);
	}
};

inline int code_of(uint8_t c) {
	switch (c) {
	case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3; default: return -1;
	}
}
inline int comp_code_of(uint8_t c) {  // complement letter; lower-case 'u' has none (alpha.cpp:3525)
	if (c == 'u') return -1;
	int k = code_of(c);
	return k < 0 ? -1 : 3 - k;
}

bool load_fasta(const char *path, std::vector<std::string> &labels, std::vector<std::string> &seqs, bool trunc_labels) {
	FILE *f = fopen(path, "rb");
	if (!f) return false;
	std::vector<char> buf(1 << 22);
	std::string line;
	bool in_rec = false;
	auto end_rec = [&]() {
		if (in_rec && seqs.back().empty()) { seqs.pop_back(); labels.pop_back(); }  // empty records are dropped
	};
	auto take = [&](const std::string &l) {
		if (!l.empty() && l[0] == '>') {
			end_rec();
			size_t e = 1;
			if (!trunc_labels) e = l.size();  // -notrunclabels (ufindexio.cpp:123-128, fastaseqsource.cpp:31)
			while (e < l.size() && !isspace((unsigned char)l[e])) ++e;  // -make_ufi truncates labels at white space
			labels.push_back(l.substr(1, e - 1));
			seqs.emplace_back();
			in_rec = true;
		} else if (in_rec) {
			std::string &s = seqs.back();
			for (unsigned char c : l)
				if (isalpha(c)) s.push_back((char)toupper(c));  // gaps, digits, blanks dropped; upper-cased
		}
	};
	size_t n;
	while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
		for (size_t i = 0; i < n; ++i) {
			char c = buf[i];
			if (c == '\n') { take(line); line.clear(); }
			else if (c != '\r') line.push_back(c);
		}
	}
	if (!line.empty()) take(line);
	end_rec();
	fclose(f);
	return true;
}

}  // namespace

// UpdateSlot for every position that is not the first indexed occurrence of its slot, in genome order (ufindex.cpp:
// 107-148,194-322): order dependent, so one after the other.  What each insert will touch can be guessed ahead of time:
// read-only "walkers" follow the chain of the items 16..64 places ahead, one link per visit, prefetching the next
// link and finally the lines FindFreeSlot will scan after the chain's end.  Walkers may see a slightly stale
// table; they only warm the cache, so the result is unaffected.
static void insert_overflow_in_order(Table &T, const uint64_t *oslot, const uint32_t *opos, size_t novf) {
	uint8_t *const blob = T.blob;
	struct Walker { uint64_t slot; int phase; };
	Walker ring[128];
	for (auto &w : ring) w = Walker{0, 1};
	auto wstep = [&](size_t j) {
		Walker &w = ring[j & 127];
		if (w.phase != 0) return;
		uint64_t nxt = w.slot;
		if (T.advance(nxt)) {
			w.slot = nxt;
			__builtin_prefetch(blob + 5 * nxt, 0);
		} else {
			w.phase = 1;
			const uint64_t c = T.wrap(w.slot, 1);
			__builtin_prefetch(&T.nplus[c], 0);
			__builtin_prefetch(&T.nplus[c] + 64, 0);
			__builtin_prefetch(blob + 5 * c, 1);
			__builtin_prefetch(blob + 5 * c + 64, 1);
			__builtin_prefetch(blob + 5 * c + 128, 1);
		}
	};
	for (size_t i = 0; i < novf; ++i) {
		if (i + 128 < novf) __builtin_prefetch(blob + 5 * oslot[i + 128], 0);
		if (i + 64 < novf) { ring[(i + 64) & 127] = Walker{oslot[i + 64], 0}; wstep(i + 64); }
		if (i + 48 < novf) wstep(i + 48);
		if (i + 32 < novf) wstep(i + 32);
		if (i + 16 < novf) wstep(i + 16);
		T.insert(oslot[i], opos[i]);
	}
}

// The order-dependent tail of the GPU-assisted builder (make_ufi_gpu.hip): blob holds the head slots, nplus the
// plus-strand counts (saturated at 255), (oslot, opos) the overflow positions in genome order.
int urx_finish_slots_host(uint8_t *blob, uint64_t slots, uint32_t max_ix, uint8_t *nplus, const uint64_t *oslot,
                          const uint32_t *opos, size_t novf, uint32_t *truncated_out) {
	Table T;
	T.blob = blob; T.N = slots; T.maxIx = max_ix; T.nplus = nplus;
	insert_overflow_in_order(T, oslot, opos, novf);
	if (truncated_out) *truncated_out = T.truncated;
	return URMAPX_OK;
}

namespace {
}  // namespace

// Builds the slot table for an already concatenated, upper-cased sequence store.  blob must hold 5*slots bytes.
extern "C" int urmapx_build_slots(const uint8_t *seqdata, uint32_t size, uint32_t W, uint32_t max_ix, uint64_t slots,
                                  uint8_t *blob, uint32_t *truncated_out) {
	if (!seqdata || !blob || slots == 0 || W < 1 || W > 32) return URMAPX_E_ARG;
	Table T;
	T.blob = blob; T.N = slots; T.maxIx = max_ix;
	const bool verbose = getenv("URMAPX_VERBOSE") != nullptr;
	auto tprev = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		auto now = std::chrono::steady_clock::now();
		if (verbose) fprintf(stderr, "[make_ufi] %-28s %.2f s\n", what, std::chrono::duration<double>(now - tprev).count());
		tprev = now;
	};
	auto huge = [](void *p, size_t bytes) {  // transparent huge pages for the big random-access arrays
		uintptr_t a = ((uintptr_t)p + 0x1fffff) & ~(uintptr_t)0x1fffff, e = ((uintptr_t)p + bytes) & ~(uintptr_t)0x1fffff;
		if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
	};
	huge(blob, 5 * slots);
#pragma omp parallel for schedule(static)
	for (int64_t s = 0; s < (int64_t)slots; ++s) T.put((uint64_t)s, T_FREE, 0xffffffffu);
	// big scratch arrays: 2 MB aligned, huge pages requested before the first touch, first touch in parallel
	auto big_alloc = [&](size_t bytes, int fill) -> void * {
		void *p = aligned_alloc(1u << 21, (bytes + (1u << 21) - 1) & ~(size_t)((1u << 21) - 1));
		if (!p) return nullptr;
		huge(p, bytes);
		const int64_t nblk = (int64_t)((bytes + (1u << 21) - 1) >> 21);
#pragma omp parallel for schedule(static)
		for (int64_t b = 0; b < nblk; ++b) {
			const size_t o = (size_t)b << 21;
			memset((uint8_t *)p + o, fill, std::min<size_t>(1u << 21, bytes - o));
		}
		return p;
	};
	T.nplus = (uint8_t *)big_alloc(slots, 0);
	T.nminus = (uint8_t *)big_alloc(slots, 0);
	uint32_t *firstpos = (uint32_t *)big_alloc(4 * slots, 0xff);
	if (!T.nplus || !T.nminus || !firstpos) { free(T.nplus); free(T.nminus); free(firstpos); return URMAPX_E_NOMEM; }
	lap("init");
	const uint64_t mask = W >= 32 ? ~0ull : ((1ull << (2 * W)) - 1);
	// passes 1 and 2: per-slot counts of plus-strand words (forward scan) and minus-strand words (backward scan,
	// complemented letters), saturating at 255.  Counting commutes, so the sequence is cut into chunks that are
	// scanned in parallel; each chunk re-reads the W-1 bases before it to rebuild the rolling word.
	auto bump = [](uint8_t &c) {
		uint8_t old = __atomic_load_n(&c, __ATOMIC_RELAXED);
		while (old < 255 && !__atomic_compare_exchange_n(&c, &old, (uint8_t)(old + 1), true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	};
	const uint32_t CHUNK = 1u << 22;
	const int64_t nchunks = ((int64_t)size + CHUNK - 1) / CHUNK;
	for (int pass = 0; pass < 2; ++pass) {
		uint8_t *cnt = pass == 0 ? T.nplus : T.nminus;
#pragma omp parallel for schedule(dynamic, 1)
		for (int64_t ch = 0; ch < nchunks; ++ch) {
			// scan positions t in [lo, hi) of the pass order, counting words that END inside [cs, hi)
			const uint64_t cs = (uint64_t)ch * CHUNK, hi = std::min<uint64_t>(size, cs + CHUNK);
			const uint64_t lo = cs >= W - 1 ? cs - (W - 1) : 0;
			uint64_t word = 0; unsigned k = 0;
			uint64_t pend[64]; int np = 0;
			for (uint64_t t = lo; t < hi; ++t) {
				const uint32_t p = pass == 0 ? (uint32_t)t : (uint32_t)(size - 1 - t);
				const int L = pass == 0 ? code_of(seqdata[p]) : comp_code_of(seqdata[p]);
				if (L < 0) { k = 0; word = 0; continue; }
				if (k < W) ++k;
				word = (word << 2) | (uint64_t)L;
				if (k == W && t >= cs) {
					const uint64_t s = mix(word & mask) % slots;
					__builtin_prefetch(&cnt[s], 1);
					pend[np++] = s;
					if (np == 64) { for (int i = 0; i < 64; ++i) bump(cnt[pend[i]]); np = 0; }
				}
			}
			for (int i = 0; i < np; ++i) bump(cnt[pend[i]]);
		}
	}
	// Insertion (UFIndex::MakeIndex + UpdateSlot, ufindex.cpp:107-148,194-322) is order dependent only for the
	// second and later occurrences of a slot: the FIRST indexed occurrence of a slot always lands in the slot
	// itself (BOTH1/PLUS1), and FindFreeSlot never lends out a slot with 1 <= count <= MaxIx whether or not its
	// head has been written yet (ufindex.cpp:991-993).  So heads are found with a parallel minimum and written in
	// parallel, and only the remaining ("overflow") positions run through the sequential UpdateSlot, in genome order.
	lap("count passes");

	auto scan_chunk = [&](int64_t ch, auto &&fn) {  // fn(slot, startpos) for every indexed-eligible word ending in the chunk
		const uint64_t cs = (uint64_t)ch * CHUNK, hi = std::min<uint64_t>(size, cs + CHUNK);
		const uint64_t lo = cs >= W - 1 ? cs - (W - 1) : 0;
		uint64_t word = 0; unsigned k = 0;
		for (uint64_t t = lo; t < hi; ++t) {
			const int L = code_of(seqdata[t]);
			if (L < 0) { k = 0; word = 0; continue; }
			if (k < W) ++k;
			word = (word << 2) | (uint64_t)L;
			if (k == W && t >= cs) {
				const uint64_t sl = mix(word & mask) % slots;
				fn(sl, (uint32_t)(t - (W - 1)));
			}
		}
	};
#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t ch = 0; ch < nchunks; ++ch)
		scan_chunk(ch, [&](uint64_t sl, uint32_t p) {
			if (T.nplus[sl] > max_ix || T.nminus[sl] > max_ix) return;
			uint32_t old = __atomic_load_n(&firstpos[sl], __ATOMIC_RELAXED);
			while (p < old && !__atomic_compare_exchange_n(&firstpos[sl], &old, p, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
		});
	lap("first occurrences");
	struct Ovf { uint64_t slot; uint32_t pos; };
	std::vector<std::vector<Ovf>> ovf((size_t)nchunks);
#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t ch = 0; ch < nchunks; ++ch)
		scan_chunk(ch, [&](uint64_t sl, uint32_t p) {
			const uint8_t n = T.nplus[sl], nm = T.nminus[sl];
			if (n > max_ix || nm > max_ix) return;
			if (firstpos[sl] == p) T.put(sl, (n == 1 && nm == 0) ? T_BOTH1 : T_PLUS1, p);
			else ovf[(size_t)ch].push_back(Ovf{sl, p});
		});
	free(firstpos);
	lap("heads + overflow lists");
	size_t novf = 0;
	for (auto &v : ovf) novf += v.size();
	// flatten (parallel) so that the sequential pass can look ahead across chunk borders
	std::vector<size_t> obase((size_t)nchunks + 1, 0);
	for (int64_t ch = 0; ch < nchunks; ++ch) obase[(size_t)ch + 1] = obase[(size_t)ch] + ovf[(size_t)ch].size();
	std::vector<Ovf> items(novf);
#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t ch = 0; ch < nchunks; ++ch) {
		std::copy(ovf[(size_t)ch].begin(), ovf[(size_t)ch].end(), items.begin() + (ptrdiff_t)obase[(size_t)ch]);
		std::vector<Ovf>().swap(ovf[(size_t)ch]);
	}
	lap("flatten overflow list");
	{
		std::vector<uint64_t> oslot(novf);
		std::vector<uint32_t> opos(novf);
#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)novf; ++i) { oslot[(size_t)i] = items[(size_t)i].slot; opos[(size_t)i] = items[(size_t)i].pos; }
		std::vector<Ovf>().swap(items);
		insert_overflow_in_order(T, oslot.data(), opos.data(), novf);
	}
	lap("sequential overflow inserts");
	if (verbose) fprintf(stderr, "[make_ufi] %zu overflow positions, %u truncated\n", novf, T.truncated);
	free(T.nplus);
	free(T.nminus);
	if (truncated_out) *truncated_out = T.truncated;
	return URMAPX_OK;
}

extern "C" int urmapx_build_slots_gpu(int device, const uint8_t *seqdata, const void *d_seqdata, uint32_t size, uint32_t W,
                                      uint32_t max_ix, uint64_t slots, uint8_t *blob, uint32_t *truncated_out);
static int make_ufi_impl(const char *fasta_path, const char *ufi_path, uint32_t W, uint32_t max_ix, uint64_t slots, int device, unsigned flags = 0);

// -make_ufi FASTA -output UFI [-wordlength W] [-maxix N] -slots S
extern "C" int urmapx_make_ufi(const char *fasta_path, const char *ufi_path, uint32_t W, uint32_t max_ix, uint64_t slots) {
	return make_ufi_impl(fasta_path, ufi_path, W, max_ix, slots, -1);
}
// the same with the counting passes, the head slots and the overflow list made on `device` (make_ufi_gpu.hip)
extern "C" int urmapx_make_ufi_gpu(int device, const char *fasta_path, const char *ufi_path, uint32_t W, uint32_t max_ix, uint64_t slots) {
	return device < 0 ? URMAPX_E_ARG : make_ufi_impl(fasta_path, ufi_path, W, max_ix, slots, device);
}

// device < 0: host builder; flags: URMAPX_UFI_*
extern "C" int urmapx_make_ufi_opts(int device, const char *fasta_path, const char *ufi_path, uint32_t W, uint32_t max_ix, uint64_t slots,
                                    unsigned flags) {
	return make_ufi_impl(fasta_path, ufi_path, W, max_ix, slots, device, flags);
}

static int make_ufi_impl(const char *fasta_path, const char *ufi_path, uint32_t W, uint32_t max_ix, uint64_t slots, int device, unsigned flags) {
	if (!fasta_path || !ufi_path || slots == 0) return URMAPX_E_ARG;
	std::vector<std::string> labels, seqs;
	if (!load_fasta(fasta_path, labels, seqs, !(flags & URMAPX_UFI_KEEP_LABELS))) return URMAPX_E_IO;
	if (seqs.empty()) return URMAPX_E_FORMAT;
	std::vector<uint32_t> lens, offs;
	uint64_t total = 0;
	for (size_t i = 0; i < seqs.size(); ++i) {
		lens.push_back((uint32_t)seqs[i].size());
		offs.push_back((uint32_t)total);
		total += seqs[i].size();
		if (i + 1 != seqs.size()) total += PAD;
	}
	if (total > 0xFFFFFFFFull - 100000) return URMAPX_E_UNSUPPORTED;  // "Genome too big", ufindexio.cpp:152-154
	std::vector<uint8_t> store((size_t)total);
	for (size_t i = 0; i < seqs.size(); ++i) {
		memcpy(store.data() + offs[i], seqs[i].data(), seqs[i].size());
		if (i + 1 != seqs.size()) memset(store.data() + offs[i] + seqs[i].size(), '-', PAD);
		std::string().swap(seqs[i]);
	}
	uint8_t *blob = (uint8_t *)malloc(5 * slots);
	if (!blob) return URMAPX_E_NOMEM;
	int rc = device < 0 ? urmapx_build_slots(store.data(), (uint32_t)total, W, max_ix, slots, blob, nullptr)
	                    : urmapx_build_slots_gpu(device, store.data(), nullptr, (uint32_t)total, W, max_ix, slots, blob, nullptr);
	if (rc) { free(blob); return rc; }
	FILE *f = fopen(ufi_path, "wb");
	if (!f) { free(blob); return URMAPX_E_IO; }
	auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
	w32(M1); w32(W); w32(max_ix); w32((uint32_t)total);
	fwrite(&slots, 8, 1, f);
	w32((uint32_t)labels.size());
	for (size_t i = 0; i < labels.size(); ++i) {
		w32(lens[i]); w32(offs[i]); w32((uint32_t)labels[i].size());
		fwrite(labels[i].data(), 1, labels[i].size(), f);
	}
	w32(M2);
	fwrite(blob, 1, 5 * slots, f);
	w32(M3);
	fwrite(store.data(), 1, store.size(), f);
	w32(M5);
	free(blob);
	return fclose(f) == 0 ? URMAPX_OK : URMAPX_E_IO;
}
