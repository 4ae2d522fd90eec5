// sam.h -- host-side text of the mapping path: FASTQ in, SAM out.  Product code (not the oracle).
// Follows SetSAM / SetSAM_Unmapped (setsam.cpp:12-207), GetCIGAR / PathToCIGAR / CIGAROpsFixDanglingMs
// (state1.cpp:707-734, cigar.cpp:4-41,141-199) and FASTQSeqSource::GetNextLo (fastqseqsource.cpp:9-116).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <functional>
#include <string>
#include <vector>

#include "../../include/urmapx.h"

namespace urx {

// growable array of plain data whose new elements are NOT zero-filled (the batch arrays are tens of MB and are
// overwritten at once; std::vector::resize would clear them on one thread first)
template <class T>
class PodVec {
public:
	PodVec() = default;
	PodVec(const PodVec &) = delete;
	PodVec &operator=(const PodVec &) = delete;
	~PodVec() {
		if (pre_free_ && p_) pre_free_(pre_free_ctx_, p_);
		free(p_);
	}
	size_t capacity() const { return cap_; }
	// called with the current storage before it is reallocated or freed (a page-locked array is unregistered first)
	void set_pre_free(void (*fn)(void *ctx, void *ptr), void *ctx) { pre_free_ = fn; pre_free_ctx_ = ctx; }
	T *data() { return p_; }
	const T *data() const { return p_; }
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	T &operator[](size_t i) { return p_[i]; }
	const T &operator[](size_t i) const { return p_[i]; }
	void clear() { n_ = 0; }
	void resize(size_t n) {
		if (n > cap_) {
			size_t c = cap_ ? cap_ : 1024;
			while (c < n) c *= 2;
			if (pre_free_ && p_) pre_free_(pre_free_ctx_, p_);
			T *q = (T *)realloc(p_, c * sizeof(T));
			if (!q) throw std::bad_alloc();
			p_ = q; cap_ = c;
		}
		n_ = n;
	}

private:
	T *p_ = nullptr;
	size_t n_ = 0, cap_ = 0;
	void (*pre_free_)(void *, void *) = nullptr;
	void *pre_free_ctx_ = nullptr;
};

struct FastqBatch {
	PodVec<char> label_data;      // NUL-terminated labels, back to back
	PodVec<uint64_t> label_offs;  // n
	PodVec<uint8_t> bases, quals;  // concatenated
	PodVec<uint64_t> offs;         // n+1
	uint32_t size() const { return (uint32_t)label_offs.size(); }
	const char *label(uint32_t i) const { return label_data.data() + label_offs[i]; }
	void clear() { label_data.clear(); label_offs.clear(); bases.clear(); quals.clear(); offs.resize(1); offs[0] = 0; }
};

// FASTQ reader (plain or .gz by suffix, like LineReader::Open, linereader.cpp:14-29).  The file is read in large
// blocks; line ends are located once, then the records of a batch are validated and copied by all host threads
// (the reference parses one line at a time under a lock, fastqseqsource.cpp:9-116 -- same accept/reject rules).
class FastqReader {
public:
	~FastqReader();
	bool open(const std::string &path, std::string &err);
	// plain seekable files: continue at byte `offset`, which starts line number `lines_before` + 1 (the part of the file
	// in front was consumed by the device parser)
	bool resume_at(uint64_t offset, uint64_t lines_before);
	// plain seekable files: the reader stops at byte `end` as if the file ended there (one shard of a sharded run)
	void set_limit(uint64_t end) { limit_ = end; }
	// one shard of a single-end sharded run: the lines of the file in front of the shard are not counted unless a message needs
	// a line number -- `base` is asked then, once, and its answer added to the line numbers counted from the shard's start
	void set_lazy_line_base(std::function<uint64_t()> base) { line_base_ = std::move(base); }
	// pipes: continue with `prefix` (bytes another reader of the same descriptor took from it and gives back) and then
	// whatever the descriptor still holds; the prefix starts line number `lines_before` + 1
	bool resume_with_prefix(std::vector<char> &&prefix, uint64_t lines_before);
	bool is_pipe() const { return f_ && !seekable_; }  // plain input that cannot seek (a FIFO, standard input)
	int fd() const;
	// appends up to max_reads records; returns false at EOF with nothing read.  Sets err on malformed input.
	bool next_batch(FastqBatch &B, uint32_t max_reads, std::string &err);
	const std::string &path() const { return path_; }

private:
	size_t read_some(char *dst, size_t cap);
	std::string path_, io_error_;
	FILE *f_ = nullptr;
	void *gz_ = nullptr;
	std::vector<char> buf_;
	size_t beg_ = 0, have_ = 0;  // unconsumed input = buf_[beg_, have_)
	double bytes_per_line_ = 160;  // running estimate, sizes the reads and the scan windows
	uint64_t file_off_ = 0;  // plain files: next byte to read
	uint64_t limit_ = ~0ull;  // plain files: first byte not to read
	bool eof_ = false, finished_ = false, seekable_ = true;
	bool started_ = false;  // next_batch has been called: the reader cannot be positioned any more
	uint64_t line_nr_ = 0;  // lines consumed so far
	std::function<uint64_t()> line_base_;  // see set_lazy_line_base
	uint64_t line_base() { if (line_base_) { line_nr_ += line_base_(); line_base_ = nullptr; } return 0; }
	std::vector<size_t> ends_;  // scratch: end offset of every line of the batch
	std::vector<char> prefix_;  // resume_with_prefix: read before the descriptor
	size_t prefix_pos_ = 0;
};

// mates interleaved (reads 2i, 2i+1 = pair i), all host threads
void interleave_batches(const FastqBatch &a, const FastqBatch &b, FastqBatch &out);

// CIGAR of a run-length path (NULL/0 => "<QL>M"), D<->I swapped, dangling terminal M merged.
std::string path_to_cigar(const urmapx_path_op *ops, unsigned nops, unsigned QL);

// One SAM record.  flags/mate fields as SetSAM's arguments (SE passes 0, "*", UINT32_MAX, 0: output1.cpp:13).
void append_sam_record(std::string &out, const urmapx_index *I, const urmapx_result &r, const urmapx_path_op *ops,
                       uint32_t flags, const char *mate_label, uint32_t mate_pos, int tlen, const char *label,
                       const uint8_t *seq, const uint8_t *qual, unsigned QL);

// One -tabbedout line (State2::OutputTab2, outputtab2.cpp:85-120); label1 = the first mate's label, n bytes, no '@'
void append_tab_pe(std::string &out, const urmapx_index *I, const urmapx_result *r1, const urmapx_result *r2, const urmapx_pair_info *info,
                   const char *label1, size_t n, uint32_t len1, uint32_t len2, int sam_on);

// 256-entry complement table of alpha.cpp:3005 (IUPAC, case preserving, 'u' and non-letters -> '?')
const unsigned char *complement_table();

// @SQ lines + @PG (State1::WriteSAMHeader, state1.cpp:736-752)
void append_sam_header(std::string &out, const urmapx_index *I, int argc, char **argv);
void append_sam_header_text(std::string &out, const urmapx_index *I, const char *cmdline);

}  // namespace urx
