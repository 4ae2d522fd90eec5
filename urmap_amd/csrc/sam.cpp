#include "sam.h"

#include <omp.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>

namespace urx {

// ---------------- FASTQ ----------------
FastqReader::~FastqReader() {
	if (f_ && f_ != stdin) fclose(f_);
	if (gz_) gzclose((gzFile)gz_);
}

bool FastqReader::open(const std::string &path, std::string &err) {
	path_ = path;
	buf_.resize(64u << 20);
	const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
	if (gz) {
		gz_ = gzopen(path.c_str(), "rb");
		if (!gz_) { err = "Error opening gzip file " + path; return false; }  // gzipfileio.cpp:8-14
		gzbuffer((gzFile)gz_, 1u << 20);
	} else {
		// OpenStdioFile (myutils.cpp:426-446): "-" is standard input
		if (path.empty()) { err = "Missing input file name"; return false; }
		f_ = path == "-" ? stdin : fopen(path.c_str(), "rb");
		if (!f_) { err = "Cannot open " + path + ", errno=" + std::to_string(errno) + " " + strerror(errno); return false; }
		setvbuf(f_, nullptr, _IONBF, 0);
		seekable_ = lseek(fileno(f_), 0, SEEK_CUR) != (off_t)-1;  // a pipe / FIFO is read front to back by one thread
	}
	return true;
}

bool FastqReader::resume_at(uint64_t offset, uint64_t lines_before) {
	if (have_ || started_) return false;
	if (gz_) {  // offset counts uncompressed bytes; zlib inflates its way there from the start of the file
		if (gzseek((gzFile)gz_, (z_off_t)offset, SEEK_SET) < 0) return false;
		line_nr_ = lines_before;
		return true;
	}
	if (!f_ || !seekable_) return false;
	file_off_ = offset;
	line_nr_ = lines_before;
	return true;
}

bool FastqReader::resume_with_prefix(std::vector<char> &&prefix, uint64_t lines_before) {
	if (have_ || started_ || !f_) return false;
	prefix_ = std::move(prefix);
	prefix_pos_ = 0;
	line_nr_ = lines_before;
	return true;
}

int FastqReader::fd() const { return f_ ? fileno(f_) : -1; }

size_t FastqReader::read_some(char *dst, size_t cap) {
	if (prefix_pos_ < prefix_.size()) {
		const size_t n = std::min(cap, prefix_.size() - prefix_pos_);
		memcpy(dst, prefix_.data() + prefix_pos_, n);
		prefix_pos_ += n;
		if (prefix_pos_ == prefix_.size()) { std::vector<char>().swap(prefix_); prefix_pos_ = 0; }
		return n;
	}
	if (gz_) {
		if (cap > (1u << 30)) cap = 1u << 30;
		int n = gzread((gzFile)gz_, dst, (unsigned)cap);
		if (n < 0) io_error_ = "Error reading gzip file";  // gzipfileio.cpp:16-22
		return n > 0 ? (size_t)n : 0;
	}
	// plain file: the block is read by all threads at their own offsets (one thread copies ~2.5 GB/s from the page cache)
	const int fd = fileno(f_);
	if (!seekable_) {
		size_t done = 0;
		while (done < cap) {
			ssize_t n = read(fd, dst + done, cap - done);
			if (n <= 0) break;
			done += (size_t)n;
		}
		return done;
	}
	if (file_off_ >= limit_) return 0;
	if (cap > limit_ - file_off_) cap = (size_t)(limit_ - file_off_);
	const int T = cap >= (8u << 20) ? omp_get_max_threads() : 1;
	std::vector<size_t> got((size_t)T, 0);
#pragma omp parallel for schedule(static, 1) num_threads(T)
	for (int t = 0; t < T; ++t) {
		const size_t lo = cap * (size_t)t / (size_t)T, hi = cap * (size_t)(t + 1) / (size_t)T;
		size_t done = 0;
		while (lo + done < hi) {
			ssize_t n = pread(fd, dst + lo + done, hi - lo - done, (off_t)(file_off_ + lo + done));
			if (n <= 0) break;
			done += (size_t)n;
		}
		got[(size_t)t] = done;
	}
	size_t total = 0;
	for (int t = 0; t < T; ++t) {  // a short slice means end of file: nothing after it counts
		total += got[(size_t)t];
		const size_t lo = cap * (size_t)t / (size_t)T, hi = cap * (size_t)(t + 1) / (size_t)T;
		if (got[(size_t)t] < hi - lo) break;
	}
	file_off_ += total;
	return total;
}

namespace {
// a line = [s, e) without its '\n'; '\r' is dropped anywhere in it (linereader.cpp:54-101)
inline size_t line_len(const char *s, const char *e) {
	size_t n = (size_t)(e - s);
	for (const char *c = (const char *)memchr(s, '\r', n); c; c = (const char *)memchr(c + 1, '\r', (size_t)(e - c - 1))) --n;
	return n;
}
// letters = bytes in [A-Za-z]; other = bytes that are neither a letter nor '\r'.  Eight bytes per step.
inline void count_letters(const char *s, const char *e, size_t &letters, size_t &other) {
	const uint64_t K01 = 0x0101010101010101ull, K80 = 0x80 * K01, K7F = 0x7F * K01;
	size_t nl = 0, ncr = 0;
	const size_t n = (size_t)(e - s);
	for (; s + 8 <= e; s += 8) {
		uint64_t w;
		memcpy(&w, s, 8);
		const uint64_t x = (w | 0x20 * K01) & K7F;                                   // case folded, 7 bits
		const uint64_t ge_a = x + (0x80 - 'a') * K01, gt_z = x + (0x80 - 'z' - 1) * K01;  // bit 7: x >= 'a', x > 'z'
		nl += (size_t)__builtin_popcountll(ge_a & ~gt_z & ~w & K80);
		const uint64_t y = w ^ 0x0D * K01;                                            // zero byte where '\r'
		ncr += (size_t)__builtin_popcountll(~(((y & K7F) + K7F) | y) & K80);
	}
	for (; s < e; ++s) {
		const unsigned char u = (unsigned char)*s;
		nl += (unsigned char)((u | 0x20u) - 'a') < 26u;
		ncr += u == '\r';
	}
	letters = nl;
	other = n - nl - ncr;
}
inline void line_copy(char *dst, const char *s, const char *e) {
	if (!memchr(s, '\r', (size_t)(e - s))) { memcpy(dst, s, (size_t)(e - s)); return; }
	for (; s < e; ++s)
		if (*s != '\r') *dst++ = *s;
}
}  // namespace

bool FastqReader::next_batch(FastqBatch &B, uint32_t max_reads, std::string &err) {
	if (B.offs.empty()) { B.offs.resize(1); B.offs[0] = 0; }
	if (finished_ || max_reads == 0) return false;
	started_ = true;
	// 1. line ends of up to max_reads records (a final unterminated line counts)
	const size_t want = 4 * (size_t)max_reads;
	ends_.clear();
	size_t scan = beg_;
	// bytes the missing lines are likely to take, from the line length seen so far (a little over: a short read is a second pass)
	auto likely_bytes = [&]() { return (size_t)((double)(want - ends_.size()) * bytes_per_line_ * 1.02) + 4096; };
	for (;;) {
		while (ends_.size() < want && scan < have_) {
			// look ahead about as far as the missing lines are likely to reach; big windows are scanned by all threads
			size_t win = likely_bytes();
			if (win < (1u << 20)) win = 1u << 20;
			if (win > have_ - scan) win = have_ - scan;
			const char *w0 = buf_.data() + scan;
			if (win < (4u << 20)) {
				const char *c = w0, *e = w0 + win;
				while (ends_.size() < want && c < e) {
					const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
					if (!nl) { c = e; break; }
					ends_.push_back((size_t)(nl - buf_.data()));
					c = nl + 1;
				}
				scan = (size_t)(c - buf_.data());
			} else {
				const int T = omp_get_max_threads();
				std::vector<std::vector<size_t>> part((size_t)T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
				for (int t = 0; t < T; ++t) {
					const char *c = w0 + win * (size_t)t / (size_t)T, *e = w0 + win * (size_t)(t + 1) / (size_t)T;
					std::vector<size_t> &v = part[(size_t)t];
					while (c < e) {
						const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
						if (!nl) break;
						v.push_back((size_t)(nl - buf_.data()));
						c = nl + 1;
					}
				}
				bool full = false;
				for (int t = 0; t < T && !full; ++t)
					for (size_t x : part[(size_t)t]) {
						ends_.push_back(x);
						if (ends_.size() >= want) { full = true; break; }
					}
				scan = full ? ends_.back() + 1 : scan + win;
			}
		}
		if (ends_.size() >= want || eof_) break;
		// read what the batch still needs, not the whole buffer: whatever is left over is carried to the next call
		size_t ask = likely_bytes();
		if (ask < (1u << 20)) ask = 1u << 20;
		if (buf_.size() - have_ < ask && beg_) {  // consumed input in front: move the rest down (line ends found so far move with it)
			memmove(buf_.data(), buf_.data() + beg_, have_ - beg_);
			for (size_t &e : ends_) e -= beg_;
			scan -= beg_; have_ -= beg_; beg_ = 0;
		}
		if (buf_.size() - have_ < ask) buf_.resize(std::max(buf_.size() * 2, have_ + ask));
		const size_t n = read_some(buf_.data() + have_, ask);
		if (!io_error_.empty()) { err = io_error_; return false; }
		if (n == 0) eof_ = true;
		have_ += n;
	}
	const size_t last_end = ends_.empty() ? beg_ : ends_.back() + 1;
	bool virtual_tail = false;
	if (eof_ && ends_.size() < want && last_end < have_) { ends_.push_back(have_); virtual_tail = true; }
	const size_t nlines = ends_.size();
	if (nlines == 0) { finished_ = true; return false; }
	const char *base = buf_.data();
	auto lstart = [&](size_t k) { return base + (k == 0 ? beg_ : ends_[k - 1] + 1); };
	auto lend = [&](size_t k) { return base + ends_[k]; };
	size_t nrec = nlines / 4;
	// 2. validate all complete records (parallel); the lowest-numbered problem wins
	const size_t n0 = B.label_offs.size();
	std::vector<uint32_t> blen(nrec), llen(nrec);
	size_t bad = nrec;  // first record that is blank or malformed
#pragma omp parallel for schedule(static) reduction(min : bad)
	for (size_t i = 0; i < nrec; ++i) {
		const char *s1 = lstart(4 * i), *e1 = lend(4 * i);
		const size_t l1 = line_len(s1, e1);
		bool ok = l1 > 0;
		if (ok) {
			const char *c = s1;
			while (*c == '\r') ++c;
			ok = *c == '@';
		}
		const char *s2 = lstart(4 * i + 1), *e2 = lend(4 * i + 1);
		size_t l2, other;  // letters (isalpha in the C locale), and bytes that are neither a letter nor '\r'
		count_letters(s2, e2, l2, other);
		if (other) ok = false;
		l2 += other;  // the line's length without its '\r's
		const size_t l4 = line_len(lstart(4 * i + 3), lend(4 * i + 3));
		if (l4 != l2) ok = false;
		blen[i] = (uint32_t)l2;
		llen[i] = (uint32_t)(l1 > 0 ? l1 - 1 : 0);
		if (!ok && i < bad) bad = i;
	}
	size_t leftover_from = 4 * nrec;  // first line not consumed by a good record
	if (bad < nrec) { nrec = bad; leftover_from = 4 * bad; }
	// 3. copy (parallel) after a prefix sum of the lengths
	std::vector<uint64_t> bo(nrec + 1), lo(nrec + 1);
	bo[0] = B.bases.size(); lo[0] = B.label_data.size();
	for (size_t i = 0; i < nrec; ++i) { bo[i + 1] = bo[i] + blen[i]; lo[i + 1] = lo[i] + llen[i] + 1; }
	B.bases.resize(bo[nrec]); B.quals.resize(bo[nrec]); B.label_data.resize(lo[nrec]);
	B.label_offs.resize(n0 + nrec); B.offs.resize(n0 + nrec + 1);
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < nrec; ++i) {
		const char *s1 = lstart(4 * i), *e1 = lend(4 * i);
		while (*s1 == '\r') ++s1;
		line_copy(B.label_data.data() + lo[i], s1 + 1, e1);
		B.label_data[lo[i] + llen[i]] = 0;
		line_copy((char *)B.bases.data() + bo[i], lstart(4 * i + 1), lend(4 * i + 1));
		line_copy((char *)B.quals.data() + bo[i], lstart(4 * i + 3), lend(4 * i + 3));
		B.label_offs[n0 + i] = lo[i];
		B.offs[n0 + i + 1] = bo[i + 1];
	}
	line_nr_ += 4 * nrec;
	// 4. whatever follows the good records: the next batch's input, or a problem to report in the reference's words
	const bool partial_at_eof = eof_ && leftover_from < nlines && (nlines - leftover_from) < 4 && bad >= nlines / 4;
	if (bad < nlines / 4 || partial_at_eof) {
		const size_t k = leftover_from;  // line index of the record's first line
		auto text = [&](size_t j) { std::string t(line_len(lstart(j), lend(j)), 0); line_copy(&t[0], lstart(j), lend(j)); return t; };
		const std::string l1 = text(k);
		if (l1.empty()) {
			// blank lines are only allowed at end of file (fastqseqsource.cpp:31-43); the message carries the number of
			// the last blank line before the text
			bool only_blank = true;
			uint64_t blanks = 0;
			for (size_t j = k; j < nlines && only_blank; ++j) {
				only_blank = line_len(lstart(j), lend(j)) == 0;
				blanks += only_blank;
			}
			size_t pos = virtual_tail ? have_ : ends_.back() + 1;
			while (only_blank) {
				for (; pos < have_ && only_blank; ++pos) {
					only_blank = buf_[pos] == '\n' || buf_[pos] == '\r';
					blanks += buf_[pos] == '\n';
				}
				if (!only_blank || eof_) break;
				have_ = read_some(buf_.data(), buf_.size());
				pos = beg_ = 0;
				if (have_ == 0) eof_ = true;
			}
			if (!only_blank) { line_base(); err = "Empty line nr " + std::to_string(line_nr_ + blanks) + " in FASTQ file '" + path_ + "'"; return false; }
			finished_ = true;
			have_ = beg_ = 0;
			return nrec > 0;
		}
		line_base();  // a message follows: now the lines in front of a shard are worth counting
		const uint64_t ln = line_nr_ + 1;
		if (l1[0] != '@') { err = "Bad line " + std::to_string(ln) + " in FASTQ file '" + path_ + "': expected '@'"; return false; }
		if (k + 1 >= nlines) { err = "Unexpected end-of-file in FASTQ file " + path_; return false; }
		const std::string l2 = text(k + 1);
		for (unsigned char c : l2)
			if (!isalpha(c)) {  // fastqseqsource.cpp:76-84
				char hex[8];
				snprintf(hex, sizeof hex, "0x%02x", c);
				if (isprint(c)) err = std::string("Invalid sequence letter '") + (char)c + "' in FASTQ, line " + std::to_string(ln + 1) + " file " + path_;
				else err = std::string("Non-printing byte ") + hex + " in FASTQ sequence line " + std::to_string(ln + 1) + " file " + path_ + " label " + l1.substr(1);
				return false;
			}
		if (k + 3 >= nlines) { err = "Unexpected end-of-file in FASTQ file " + path_; return false; }
		const std::string l4 = text(k + 3);
		err = "Bad FASTQ record: " + std::to_string(l2.size()) + " bases, " + std::to_string(l4.size()) + " quals line " +
		      std::to_string(ln + 3) + " file " + path_ + " label " + l1.substr(1);
		return false;
	}
	// the unconsumed tail stays where it is for the next call
	const size_t consumed = leftover_from == 0 ? beg_ : (leftover_from == nlines && virtual_tail ? have_ : ends_[leftover_from - 1] + 1);
	if (nrec) bytes_per_line_ = (double)(consumed - beg_) / (double)(4 * nrec);
	beg_ = consumed;
	if (beg_ == have_) beg_ = have_ = 0;
	if (eof_ && have_ == 0) finished_ = true;
	return nrec > 0;
}

void interleave_batches(const FastqBatch &a, const FastqBatch &b, FastqBatch &out) {
	const size_t n = a.size();
	out.clear();
	out.label_offs.resize(2 * n);
	out.offs.resize(2 * n + 1);
	uint64_t bo = 0, lo = 0;
	for (size_t i = 0; i < n; ++i) {
		for (int m = 0; m < 2; ++m) {
			const FastqBatch &s = m ? b : a;
			out.label_offs[2 * i + m] = lo;
			lo += strlen(s.label((uint32_t)i)) + 1;
			out.offs[2 * i + m] = bo;
			bo += s.offs[i + 1] - s.offs[i];
		}
	}
	out.offs[2 * n] = bo;
	out.bases.resize(bo); out.quals.resize(bo); out.label_data.resize(lo);
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < n; ++i) {
		for (int m = 0; m < 2; ++m) {
			const FastqBatch &s = m ? b : a;
			const size_t L = (size_t)(s.offs[i + 1] - s.offs[i]);
			memcpy(out.bases.data() + out.offs[2 * i + m], s.bases.data() + s.offs[i], L);
			memcpy(out.quals.data() + out.offs[2 * i + m], s.quals.data() + s.offs[i], L);
			strcpy(out.label_data.data() + out.label_offs[2 * i + m], s.label((uint32_t)i));
		}
	}
}

// ---------------- SAM ----------------
static unsigned char g_comp[256];
static struct CompInit {
	CompInit() {  // complement table of alpha.cpp:3005: IUPAC, case preserving, 'u' and non-letters -> '?'
		memset(g_comp, '?', sizeof g_comp);
		const char *from = "ABCDGHKMNRSTUVWXY", *to = "TVGHCDMKNYSAABWXR";
		for (int i = 0; from[i]; ++i) {
			g_comp[(unsigned char)from[i]] = (unsigned char)to[i];
			if (from[i] != 'U') g_comp[(unsigned char)tolower(from[i])] = (unsigned char)tolower(to[i]);
		}
	}
} g_comp_init;

const unsigned char *complement_table() { return g_comp; }

static inline void append_uint(std::string &out, uint64_t v) {
	char tmp[24];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	const size_t at = out.size();
	out.resize(at + (size_t)n);
	for (int i = 0; i < n; ++i) out[at + (size_t)i] = tmp[n - 1 - i];
}
static inline void append_int(std::string &out, int64_t v) {
	if (v < 0) { out.push_back('-'); append_uint(out, (uint64_t)(-v)); }
	else append_uint(out, (uint64_t)v);
}

// The record writers below fill a per-thread scratch area through a plain pointer and append it to the output once.
static inline char *put_uint(char *p, uint64_t v) {
	char tmp[24];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	while (n) *p++ = tmp[--n];
	return p;
}
static inline char *put_int(char *p, int64_t v) {
	if (v < 0) { *p++ = '-'; return put_uint(p, (uint64_t)(-v)); }
	return put_uint(p, (uint64_t)v);
}
static inline char *put_bytes(char *p, const void *src, size_t n) {
	memcpy(p, src, n);
	return p + n;
}
// the longest CIGAR text of a path of nops runs: up to 10 digits and a letter per run ("QL M" when there is none)
static inline size_t cigar_max_chars(unsigned nops) { return 11 * ((size_t)nops + 1) + 16; }

// Any number of runs (the general kernels' paths are as long as the read): the runs are merged as they come (path D, query
// only, is CIGAR I and vice versa); the dangling-M rule needs the first three and the last three merged runs only.
static char *put_cigar(char *p, const urmapx_path_op *ops, unsigned nops, unsigned QL) {
	if (nops == 0) { p = put_uint(p, QL); *p++ = 'M'; return p; }
	char op_fixed[URMAPX_MAX_PATH_OPS + 1];
	unsigned len_fixed[URMAPX_MAX_PATH_OPS + 1];
	std::vector<char> op_big;
	std::vector<unsigned> len_big;
	char *op = op_fixed;
	unsigned *len = len_fixed;
	if (nops > URMAPX_MAX_PATH_OPS) {
		op_big.resize((size_t)nops + 1); len_big.resize((size_t)nops + 1);
		op = op_big.data(); len = len_big.data();
	}
	unsigned N = 0;
	for (unsigned i = 0; i < nops; ++i) {
		unsigned code = ops[i] & 3u, n = ops[i] >> 2;
		char c = code == 0 ? 'M' : code == 1 ? 'I' : 'D';  // path D (query only) is CIGAR I and vice versa
		if (N && op[N - 1] == c) len[N - 1] += n;
		else { op[N] = c; len[N] = n; ++N; }
	}
	// dangling terminal M of length <= 2 next to an indel > 4 is merged into the far side M (cigar.cpp:141-199);
	// the reference evaluates its tail rule against the pre-shrink size, which can never hold once the head
	// rule has fired, so it is head rule XOR tail rule.
	unsigned first = 0;
	if (N >= 3) {
		if (op[0] == 'M' && len[0] <= 2 && len[1] > 4 && op[2] == 'M') {
			len[2] += len[0];
			first = 1;
		} else if (op[N - 1] == 'M' && len[N - 1] <= 2 && len[N - 2] > 4 && op[N - 3] == 'M') {
			len[N - 3] += len[N - 1];
			--N;
		}
	}
	for (unsigned i = first; i < N; ++i) { p = put_uint(p, len[i]); *p++ = op[i]; }
	return p;
}

std::string path_to_cigar(const urmapx_path_op *ops, unsigned nops, unsigned QL) {
	std::vector<char> buf(cigar_max_chars(nops));
	return std::string(buf.data(), (size_t)(put_cigar(buf.data(), ops, nops, QL) - buf.data()));
}

static size_t qname_len(const char *label) {
	size_t n = strlen(label);
	if (n > 2 && label[n - 2] == '/' && (label[n - 1] == '1' || label[n - 1] == '2')) n -= 2;
	size_t k = 0;
	while (k < n && label[k] != ' ' && label[k] != '\t') ++k;
	return k;
}

static char *record_scratch(size_t need) {
	static thread_local std::vector<char> scratch;
	if (scratch.size() < need) scratch.resize(2 * need);
	return scratch.data();
}

static void append_unmapped(std::string &out, uint32_t aflags, const char *label, const uint8_t *seq, const uint8_t *qual,
                            unsigned QL) {
	uint32_t flags = 0x04;
	if (aflags & 0x01) flags |= 0x01;
	if (aflags & 0x40) flags |= 0x40;
	else if (aflags & 0x80) flags |= 0x80;
	if (aflags & 0x08) flags |= 0x08;
	else if (aflags & 0x20) flags |= 0x20;
	const size_t ql = qname_len(label);
	char *const buf = record_scratch(ql + 2 * (size_t)QL + 64), *p = buf;
	p = put_bytes(p, label, ql);
	*p++ = '\t';
	p = put_uint(p, flags);
	static const char kUnmappedFields[] = "\t*\t0\t0\t*\t*\t0\t0\t";
	p = put_bytes(p, kUnmappedFields, sizeof kUnmappedFields - 1);
	p = put_bytes(p, seq, QL);
	*p++ = '\t';
	if (!qual) *p++ = '*';
	else p = put_bytes(p, qual, QL);
	*p++ = '\n';
	out.append(buf, (size_t)(p - buf));
}

void append_sam_record(std::string &out, const urmapx_index *I, const urmapx_result &r, const urmapx_path_op *ops,
                       uint32_t flags, const char *mate_label, uint32_t mate_pos, int tlen, const char *label,
                       const uint8_t *seq, const uint8_t *qual, unsigned QL) {
	if (r.dbpos == 0xFFFFFFFFu) { append_unmapped(out, flags, label, seq, qual, QL); return; }
	const char *tlabel = urmapx_index_label(I, r.seq_index);
	const size_t ql = qname_len(label), tl = strlen(tlabel), ml = mate_label ? strlen(mate_label) : 0;
	char *const buf = record_scratch(ql + tl + ml + 2 * (size_t)QL + cigar_max_chars(r.path_nops) + 128), *p = buf;
	p = put_bytes(p, label, ql);
	*p++ = '\t';
	p = put_uint(p, flags);
	*p++ = '\t';
	p = put_bytes(p, tlabel, tl);
	*p++ = '\t';
	p = put_uint(p, (uint64_t)r.coord + 1);
	*p++ = '\t';
	p = put_uint(p, (unsigned)r.mapq);
	*p++ = '\t';
	p = put_cigar(p, r.path_nops ? ops + r.path_off : nullptr, r.path_nops, QL);
	*p++ = '\t';
	if (ml == 0 || (ml == 1 && mate_label[0] == '*')) *p++ = '*';
	else if (ml == tl && memcmp(mate_label, tlabel, tl) == 0) *p++ = '=';
	else p = put_bytes(p, mate_label, ml);
	*p++ = '\t';
	if (mate_pos == 0 || mate_pos == 0xFFFFFFFFu) *p++ = '0';
	else p = put_uint(p, (uint64_t)mate_pos + 1);
	*p++ = '\t';
	p = put_int(p, tlen);
	*p++ = '\t';
	if (r.plus) p = put_bytes(p, seq, QL);
	else {
		for (unsigned i = 0; i < QL; ++i) p[i] = (char)g_comp[seq[QL - 1 - i]];
		p += QL;
	}
	*p++ = '\t';
	if (!qual) *p++ = '*';
	else if (r.plus) p = put_bytes(p, qual, QL);
	else {
		for (unsigned i = 0; i < QL; ++i) p[i] = (char)qual[QL - 1 - i];
		p += QL;
	}
	*p++ = '\n';
	out.append(buf, (size_t)(p - buf));
}

void append_sam_header(std::string &out, const urmapx_index *I, int argc, char **argv) {
	std::string cl;
	for (int i = 0; i < argc; ++i) { cl += argv[i]; cl.push_back(' '); }  // argv joined with trailing spaces (state1.cpp:749-751)
	append_sam_header_text(out, I, cl.c_str());
}

void append_sam_header_text(std::string &out, const urmapx_index *I, const char *cmdline) {
	const uint32_t n = urmapx_index_seq_count(I);
	for (uint32_t i = 0; i < n; ++i) {
		out += "@SQ\tSN:";
		out += urmapx_index_label(I, i);
		out += "\tLN:";
		out += std::to_string(urmapx_index_seq_length(I, i));
		out.push_back('\n');
	}
	out += "@PG\tID:urmap\tPN:urmap\tVN:1.0.mi355x\tCL:";
	if (cmdline) out += cmdline;
	out.push_back('\n');
}

}  // namespace urx

extern "C" size_t urmapx_sam_se(const urmapx_index *I, const urmapx_result *r, const urmapx_path_op *path_ops,
                                const char *label, const uint8_t *seq, const uint8_t *qual, uint32_t read_len,
                                char *buf, size_t cap) {
	std::string out;
	urx::append_sam_record(out, I, *r, path_ops, 0, "*", 0xFFFFFFFFu, 0, label, seq, qual, read_len);
	if (out.size() > cap) return 0;
	memcpy(buf, out.data(), out.size());
	return out.size();
}

// ---------------- -tabbedout (State2::OutputTab2, outputtab2.cpp:85-120) ----------------
namespace {
using urx::append_int;
using urx::append_uint;
struct TabHit { bool has; uint32_t db; bool plus; int score; };
// UFIndex::PosToCoord (ufindex.cpp:701-727): UINT32_MAX and an empty label when the position is in the padding
uint32_t pos_to_coord_tab(const urmapx_index *I, uint32_t pos, const char **label) {
	const uint32_t n = urmapx_index_seq_count(I);
	*label = "";
	uint32_t lo = 0, hi = n - 1;
	while (lo <= hi && hi != 0xFFFFFFFFu) {
		const uint32_t k = (lo + hi) / 2;
		const uint32_t off = urmapx_index_seq_offset(I, k), sl = urmapx_index_seq_length(I, k);
		if (pos >= off && pos < off + sl) { *label = urmapx_index_label(I, k); return pos - off; }
		if (pos > off) lo = k + 1;
		else hi = k - 1;
	}
	return 0xFFFFFFFFu;
}
void pair_pos_str1(std::string &out, const urmapx_index *I, const TabHit &h, bool fwd) {  // outputtab2.cpp:29-42
	const char *lab;
	const uint32_t c = pos_to_coord_tab(I, h.db, &lab);
	out += lab;
	out.push_back(':');
	append_uint(out, (uint32_t)(c + 1));
	out.push_back('(');
	out.push_back(h.plus ? '+' : '-');
	out += fwd ? ")/1" : ")/2";
}
void pair_pos_str(std::string &out, const urmapx_index *I, const TabHit &h1, const TabHit &h2) {  // outputtab2.cpp:44-83
	if (!h1.has && !h2.has) { out.push_back('*'); return; }
	if (h1.has && !h2.has) { pair_pos_str1(out, I, h1, true); return; }
	if (!h1.has && h2.has) { pair_pos_str1(out, I, h2, false); return; }
	const char *l1, *l2;
	const uint32_t c1 = pos_to_coord_tab(I, h1.db, &l1), c2 = pos_to_coord_tab(I, h2.db, &l2);
	if (strcmp(l1, l2) == 0 && h1.plus != h2.plus) {
		out += l1;
		out.push_back(':');
		append_uint(out, (uint32_t)(c1 + 1));
		out.push_back('-');
		append_uint(out, (uint32_t)(c2 + 1));
		return;
	}
	pair_pos_str1(out, I, h1, true);
	out.push_back(',');
	pair_pos_str1(out, I, h2, false);
}
unsigned template_length(const TabHit &h1, const TabHit &h2, uint32_t len1, uint32_t len2) {  // output2.cpp:49-69
	int t;
	if (h1.db <= h2.db) t = (int)(h2.db + len2) - (int)h1.db;
	else t = (int)(h1.db + len1) - (int)h2.db;
	if (t < 0 || t > 1000) t = 0;
	return (unsigned)t;
}
}  // namespace

namespace urx {
void append_tab_pe(std::string &out, const urmapx_index *I, const urmapx_result *r1, const urmapx_result *r2, const urmapx_pair_info *info,
                   const char *label1, size_t n, uint32_t len1, uint32_t len2, int sam_on) {
	const urmapx_result *r[2] = {r1, r2};
	TabHit top[2], sec[2];
	for (int a = 0; a < 2; ++a) {
		// with SAM output on, SetSAM2's SetMappedPos has cleared a top hit that overhangs its sequence (output2.cpp:73-74)
		top[a].has = info->top_db[a] != 0xFFFFFFFFu && (!sam_on || r[a]->dbpos != 0xFFFFFFFFu);
		top[a].db = info->top_db[a]; top[a].plus = info->top_plus[a] != 0; top[a].score = info->top_score[a];
		sec[a].has = info->second_db[a] != 0xFFFFFFFFu;
		sec[a].db = info->second_db[a]; sec[a].plus = info->second_plus[a] != 0; sec[a].score = info->second_score[a];
	}
	// GetPairLabel, state1.cpp:762-778
	if (n > 2 && label1[n - 2] == '/' && (label1[n - 1] == '1' || label1[n - 1] == '2')) n -= 2;
	for (size_t i = 0; i < n && !isspace((unsigned char)label1[i]); ++i) out.push_back(label1[i]);
	out.push_back('\t');
	pair_pos_str(out, I, top[0], top[1]);
	out.push_back('\t');
	append_uint(out, r1->mapq);
	out.push_back(',');
	append_uint(out, r2->mapq);
	out.push_back('\t');
	if (sec[0].has) pair_pos_str(out, I, sec[0], sec[1]);
	else out.push_back('*');
	if (top[0].has && top[1].has && sec[0].has && sec[1].has) {  // GetInfoStr, outputtab2.cpp:6-27
		const unsigned tl1 = template_length(top[0], top[1], len1, len2), tl2 = template_length(sec[0], sec[1], len1, len2);
		out += tl1 == tl2 ? "\tTL=" : "\tTL/";
		append_uint(out, tl1);
		if (tl1 != tl2) { out.push_back(','); append_uint(out, tl2); }
		const int s1 = top[0].score + top[1].score, s2 = sec[0].score + sec[1].score;
		out += s1 == s2 ? ";Score=" : ";Score/";
		append_int(out, s1);
		if (s1 != s2) { out.push_back(','); append_int(out, s2); }
		out.push_back(';');
	}
	out.push_back('\n');
}
}  // namespace urx

extern "C" size_t urmapx_tab_pe(const urmapx_index *I, const urmapx_result *r1, const urmapx_result *r2,
                                const urmapx_pair_info *info, const char *label1, uint32_t len1, uint32_t len2, int sam_on,
                                char *buf, size_t cap) {
	if (!I || !r1 || !r2 || !info || !label1 || !buf) return 0;
	std::string out;
	urx::append_tab_pe(out, I, r1, r2, info, label1, strlen(label1), len1, len2, sam_on);
	if (out.size() > cap) return 0;
	memcpy(buf, out.data(), out.size());
	return out.size();
}

struct urmapx_fastq {
	urx::FastqReader rd;
	urx::FastqBatch batch;
	std::string err;
};

extern "C" int urmapx_fastq_open(const char *path, urmapx_fastq **out) {
	if (!path || !out) return URMAPX_E_ARG;
	urmapx_fastq *F = new urmapx_fastq;
	if (!F->rd.open(path, F->err)) { delete F; return URMAPX_E_IO; }
	*out = F;
	return URMAPX_OK;
}

extern "C" int64_t urmapx_fastq_next(urmapx_fastq *F, uint32_t max_reads, const uint8_t **bases, const uint8_t **quals,
                                     const uint64_t **offs, const char **label_data, const uint64_t **label_offs) {
	if (!F) return URMAPX_E_ARG;
	F->batch.clear();
	F->err.clear();
	F->rd.next_batch(F->batch, max_reads, F->err);
	if (!F->err.empty()) return URMAPX_E_FORMAT;
	if (bases) *bases = F->batch.bases.data();
	if (quals) *quals = F->batch.quals.data();
	if (offs) *offs = F->batch.offs.data();
	if (label_data) *label_data = F->batch.label_data.data();
	if (label_offs) *label_offs = F->batch.label_offs.data();
	return (int64_t)F->batch.size();
}

extern "C" const char *urmapx_fastq_error(const urmapx_fastq *F) { return F ? F->err.c_str() : ""; }
extern "C" void urmapx_fastq_close(urmapx_fastq *F) { delete F; }

extern "C" size_t urmapx_sam_header_sq(const urmapx_index *I, char *buf, size_t cap) {
	std::string out;
	const uint32_t n = urmapx_index_seq_count(I);
	for (uint32_t i = 0; i < n; ++i) {
		out += "@SQ\tSN:";
		out += urmapx_index_label(I, i);
		out += "\tLN:";
		out += std::to_string(urmapx_index_seq_length(I, i));
		out.push_back('\n');
	}
	if (out.size() > cap) return 0;
	memcpy(buf, out.data(), out.size());
	return out.size();
}

static uint32_t paired_flags(bool first, bool revcomp, bool mate_revcomp, bool mate_unmapped) {  // output2.cpp:18-36
	uint32_t f = first ? 0x41u : 0x81u;
	if (revcomp) f |= 0x10u;
	if (mate_unmapped) f |= 0x08u;
	else if (mate_revcomp) f |= 0x20u;
	return f;
}

extern "C" size_t urmapx_sam_pe(const urmapx_index *I, const urmapx_result *r1, const urmapx_result *r2,
                                const urmapx_path_op *path_ops, const char *label1, const uint8_t *seq1,
                                const uint8_t *qual1, uint32_t len1, const char *label2, const uint8_t *seq2,
                                const uint8_t *qual2, uint32_t len2, char *buf, size_t cap) {
	// SetSAM2, output2.cpp:61-128 (positions compared are coordinates inside the sequences, as the reference does)
	const bool m1 = r1->dbpos != 0xFFFFFFFFu, m2 = r2->dbpos != 0xFFFFFFFFu;
	const bool plus1 = m1 && r1->plus, plus2 = m2 && r2->plus;
	const bool consistent = m1 && m2 && (plus1 != plus2);
	int tlen1 = 0, tlen2 = 0;
	bool proper = false;
	if (m1 && m2) {
		if (r1->coord <= r2->coord) {
			tlen1 = (int)((r2->coord + len2) - r1->coord);  // (32-bit wrap-around, as the reference's int arithmetic does on a 2^31-base sequence; no signed overflow)
			if (tlen1 > 0 && tlen1 < 1000 && consistent) proper = true;
			if (tlen1 > 1000) tlen1 = 0;
			tlen2 = -tlen1;
		} else {
			tlen2 = (int)((r1->coord + len1) - r2->coord);
			if (tlen2 > 0 && tlen2 < 1000 && consistent) proper = true;
			if (tlen2 > 1000) tlen2 = 0;
			tlen1 = -tlen2;
		}
	}
	const bool rc1 = m1 && !r1->plus, rc2 = m2 && !r2->plus;
	uint32_t f1 = paired_flags(true, rc1, rc2, !m2), f2 = paired_flags(false, rc2, rc1, !m1);
	if (proper) { f1 |= 2u; f2 |= 2u; }
	const char *l1 = m1 ? urmapx_index_label(I, r1->seq_index) : "";
	const char *l2 = m2 ? urmapx_index_label(I, r2->seq_index) : "";
	std::string out;
	urx::append_sam_record(out, I, *r1, path_ops, f1, l2, m2 ? r2->coord : 0xFFFFFFFFu, tlen1, label1, seq1, qual1, len1);
	urx::append_sam_record(out, I, *r2, path_ops, f2, l1, m1 ? r1->coord : 0xFFFFFFFFu, tlen2, label2, seq2, qual2, len2);
	if (out.size() > cap) return 0;
	memcpy(buf, out.data(), out.size());
	return out.size();
}
