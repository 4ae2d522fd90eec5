// pgzip.h -- a gzip file inflated by several threads (host side; the .gz road of urmapx_map_files, pipeline.cpp).
//
// The reference reads .gz input through zlib, one stream, one thread (linereader.cpp:54-113, gzipfileio.cpp).  A deflate stream
// has no index: a block can only be decoded once the 32 KB of text in front of it are known, so zlib's 0.6 GB/s on one core was
// what `urmap -map reads.fastq.gz` ran at (1.9 M reads/s against 15 M from a plain file).  Here the compressed file is cut into
// segments; the thread of each segment
//   1. FINDS a block start behind its cut: a bit position where a dynamic-Huffman block header parses (complete code-length code,
//      complete literal / distance codes, an end-of-block symbol), whose symbols decode to text bytes up to the end of the block,
//      and behind which another block header parses;
//   2. DECODES from there with the window in front of it UNKNOWN: the output is 16-bit symbols, a literal byte or "byte k of the
//      32 KB in front of my start" (copies of such symbols stay symbols), until it reaches the block start the next thread found --
//      bit for bit: a thread that walks past its successor's start without landing on it proves that start false, the successor's
//      work is dropped and the thread keeps going;
//   3. the windows are RESOLVED front to back (32 K symbols per segment), then every segment's symbols become bytes in parallel.
// The first segment of a round starts where the previous round ended, so by induction every block boundary used is a real one;
// each member's CRC-32 and length are checked against its trailer as zlib does.  Files smaller than two segments, and whatever the
// parallel road cannot take (no block start found anywhere), go through zlib inflate on the calling thread.  Output bytes are
// zlib's, always.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace urx {

// zlib's crc32(0, p, n), by carry-less multiplication where the host has it (pgzip.cpp: verified against zlib at first use)
uint32_t crc32_fast(const uint8_t *p, size_t n);

class ParallelGunzip {
public:
	ParallelGunzip();
	~ParallelGunzip();
	// fd: a regular file positioned anywhere (read with pread), csize its size.  false: not a gzip file
	bool open(int fd, uint64_t csize);
	// up to cap bytes of the uncompressed text of all members, in order; 0 = end of input, or failed()
	size_t read(char *dst, size_t cap, int threads);
	bool failed() const { return bad_; }
	// statistics: bytes produced by the parallel road / by zlib
	uint64_t parallel_bytes() const { return par_bytes_; }
	uint64_t serial_bytes() const { return ser_bytes_; }

private:
	struct Impl;
	std::unique_ptr<Impl> d_;
	bool bad_ = false;
	uint64_t par_bytes_ = 0, ser_bytes_ = 0;
};

}  // namespace urx
