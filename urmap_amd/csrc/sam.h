// sam.h -- host-side text of the mapping path: FASTQ in, SAM out.  Product code (not the oracle).
// Follows SetSAM / SetSAM_Unmapped (setsam.cpp:12-207), GetCIGAR / PathToCIGAR / CIGAROpsFixDanglingMs
// (state1.cpp:707-734, cigar.cpp:4-41,141-199) and FASTQSeqSource::GetNextLo (fastqseqsource.cpp:9-116).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/urmapx.h"

namespace urx {

struct FastqBatch {
	std::vector<std::string> labels;
	std::vector<uint8_t> bases, quals;  // concatenated
	std::vector<uint64_t> offs;         // n+1
	uint32_t size() const { return (uint32_t)labels.size(); }
	void clear() { labels.clear(); bases.clear(); quals.clear(); offs.assign(1, 0); }
};

// Line-oriented FASTQ reader (plain or .gz by suffix, like LineReader::Open, linereader.cpp:14-29).
class FastqReader {
public:
	~FastqReader();
	bool open(const std::string &path, std::string &err);
	// appends up to max_reads records; returns false at EOF with nothing read.  Sets err on malformed input.
	bool next_batch(FastqBatch &B, uint32_t max_reads, std::string &err);
	const std::string &path() const { return path_; }

private:
	bool read_line(std::string &s);
	bool fill();
	std::string path_;
	FILE *f_ = nullptr;
	void *gz_ = nullptr;
	std::vector<char> buf_;
	size_t pos_ = 0, len_ = 0;
	bool eof_ = false;
	uint64_t line_nr_ = 0;
};

// CIGAR of a run-length path (NULL/0 => "<QL>M"), D<->I swapped, dangling terminal M merged.
std::string path_to_cigar(const urmapx_path_op *ops, unsigned nops, unsigned QL);

// One SAM record.  flags/mate fields as SetSAM's arguments (SE passes 0, "*", UINT32_MAX, 0: output1.cpp:13).
void append_sam_record(std::string &out, const urmapx_index *I, const urmapx_result &r, const urmapx_path_op *ops,
                       uint32_t flags, const char *mate_label, uint32_t mate_pos, int tlen, const char *label,
                       const uint8_t *seq, const uint8_t *qual, unsigned QL);

// @SQ lines + @PG (State1::WriteSAMHeader, state1.cpp:736-752)
void append_sam_header(std::string &out, const urmapx_index *I, int argc, char **argv);

}  // namespace urx
