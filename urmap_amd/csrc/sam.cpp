#include "sam.h"

#include <zlib.h>

#include <cctype>
#include <cstring>

namespace urx {

// ---------------- FASTQ ----------------
FastqReader::~FastqReader() {
	if (f_) fclose(f_);
	if (gz_) gzclose((gzFile)gz_);
}

bool FastqReader::open(const std::string &path, std::string &err) {
	path_ = path;
	buf_.resize(8u << 20);
	const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
	if (gz) {
		gz_ = gzopen(path.c_str(), "rb");
		if (!gz_) { err = "cannot open " + path; return false; }
	} else {
		f_ = fopen(path.c_str(), "rb");
		if (!f_) { err = "cannot open " + path; return false; }
	}
	return true;
}

bool FastqReader::fill() {
	if (eof_) return false;
	pos_ = 0;
	if (gz_) {
		int n = gzread((gzFile)gz_, buf_.data(), (unsigned)buf_.size());
		len_ = n > 0 ? (size_t)n : 0;
	} else
		len_ = fread(buf_.data(), 1, buf_.size(), f_);
	if (len_ == 0) eof_ = true;
	return len_ > 0;
}

// '\r' dropped anywhere, '\n' ends the line; a final unterminated line counts (linereader.cpp:54-101)
bool FastqReader::read_line(std::string &s) {
	s.clear();
	if (eof_) return false;
	for (;;) {
		if (pos_ >= len_) {
			if (!fill()) {
				if (s.empty()) return false;
				++line_nr_;
				return true;
			}
		}
		const char *p = buf_.data() + pos_;
		const char *e = buf_.data() + len_;
		const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
		const char *stop = nl ? nl : e;
		for (const char *c = p; c < stop; ++c)
			if (*c != '\r') s.push_back(*c);
		if (nl) { pos_ = (size_t)(nl - buf_.data()) + 1; ++line_nr_; return true; }
		pos_ = len_;
	}
}

bool FastqReader::next_batch(FastqBatch &B, uint32_t max_reads, std::string &err) {
	if (B.offs.empty()) B.offs.assign(1, 0);
	std::string l1, l2, l3, l4;
	uint32_t got = 0;
	while (got < max_reads) {
		if (!read_line(l1)) break;
		if (l1.empty()) {
			// blank lines are only allowed at end of file (fastqseqsource.cpp:31-43)
			while (read_line(l1))
				if (!l1.empty()) { err = "Empty line in FASTQ file '" + path_ + "'"; return false; }
			break;
		}
		if (l1[0] != '@') { err = "Bad line " + std::to_string(line_nr_) + " in FASTQ file '" + path_ + "': expected '@'"; return false; }
		if (!read_line(l2)) { err = "Unexpected end-of-file in FASTQ file " + path_; return false; }
		for (unsigned char c : l2)
			if (!isalpha(c)) { err = "Invalid sequence letter in FASTQ, line " + std::to_string(line_nr_) + " file " + path_; return false; }
		read_line(l3);
		if (!read_line(l4)) { err = "Unexpected end-of-file in FASTQ file " + path_; return false; }
		if (l4.size() != l2.size()) {
			err = "Bad FASTQ record: " + std::to_string(l2.size()) + " bases, " + std::to_string(l4.size()) + " quals line " +
			      std::to_string(line_nr_) + " file " + path_;
			return false;
		}
		B.labels.push_back(l1.substr(1));
		B.bases.insert(B.bases.end(), l2.begin(), l2.end());
		B.quals.insert(B.quals.end(), l4.begin(), l4.end());
		B.offs.push_back(B.bases.size());
		++got;
	}
	return got > 0;
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

std::string path_to_cigar(const urmapx_path_op *ops, unsigned nops, unsigned QL) {
	if (nops == 0) return std::to_string(QL) + "M";
	std::vector<char> op;
	std::vector<unsigned> len;
	for (unsigned i = 0; i < nops; ++i) {
		unsigned code = ops[i] & 3u, n = ops[i] >> 2;
		char c = code == 0 ? 'M' : code == 1 ? 'I' : 'D';  // path D (query only) is CIGAR I and vice versa
		if (!op.empty() && op.back() == c) len.back() += n;
		else { op.push_back(c); len.push_back(n); }
	}
	// dangling terminal M of length <= 2 next to an indel > 4 is merged into the far side M (cigar.cpp:141-199);
	// the reference evaluates its tail rule against the pre-shrink size, which can never hold once the head
	// rule has fired, so it is head rule XOR tail rule.
	size_t N = op.size();
	if (N >= 3) {
		if (op[0] == 'M' && len[0] <= 2 && len[1] > 4 && op[2] == 'M') {
			len[2] += len[0];
			op.erase(op.begin());
			len.erase(len.begin());
		} else if (op[N - 1] == 'M' && len[N - 1] <= 2 && len[N - 2] > 4 && op[N - 3] == 'M') {
			len[N - 3] += len[N - 1];
			op.pop_back();
			len.pop_back();
		}
	}
	std::string s;
	for (size_t i = 0; i < op.size(); ++i) { s += std::to_string(len[i]); s.push_back(op[i]); }
	return s;
}

static size_t qname_len(const char *label) {
	size_t n = strlen(label);
	if (n > 2 && label[n - 2] == '/' && (label[n - 1] == '1' || label[n - 1] == '2')) n -= 2;
	size_t k = 0;
	while (k < n && label[k] != ' ' && label[k] != '\t') ++k;
	return k;
}

static void append_unmapped(std::string &out, uint32_t aflags, const char *label, const uint8_t *seq, const uint8_t *qual,
                            unsigned QL) {
	uint32_t flags = 0x04;
	if (aflags & 0x01) flags |= 0x01;
	if (aflags & 0x40) flags |= 0x40;
	else if (aflags & 0x80) flags |= 0x80;
	if (aflags & 0x08) flags |= 0x08;
	else if (aflags & 0x20) flags |= 0x20;
	out.append(label, qname_len(label));
	out.push_back('\t');
	out += std::to_string(flags);
	out += "\t*\t0\t0\t*\t*\t0\t0\t";
	out.append((const char *)seq, QL);
	out.push_back('\t');
	if (!qual) out.push_back('*');
	else out.append((const char *)qual, QL);
	out.push_back('\n');
}

void append_sam_record(std::string &out, const urmapx_index *I, const urmapx_result &r, const urmapx_path_op *ops,
                       uint32_t flags, const char *mate_label, uint32_t mate_pos, int tlen, const char *label,
                       const uint8_t *seq, const uint8_t *qual, unsigned QL) {
	if (r.dbpos == 0xFFFFFFFFu) { append_unmapped(out, flags, label, seq, qual, QL); return; }
	const char *tlabel = urmapx_index_label(I, r.seq_index);
	out.append(label, qname_len(label));
	out.push_back('\t');
	out += std::to_string(flags);
	out.push_back('\t');
	out += tlabel;
	out.push_back('\t');
	out += std::to_string(r.coord + 1);
	out.push_back('\t');
	out += std::to_string((unsigned)r.mapq);
	out.push_back('\t');
	out += path_to_cigar(r.path_nops ? ops + r.path_off : nullptr, r.path_nops, QL);
	out.push_back('\t');
	if (!mate_label || !*mate_label || strcmp(mate_label, "*") == 0) out.push_back('*');
	else if (strcmp(mate_label, tlabel) == 0) out.push_back('=');
	else out += mate_label;
	out.push_back('\t');
	if (mate_pos == 0 || mate_pos == 0xFFFFFFFFu) out.push_back('0');
	else out += std::to_string(mate_pos + 1);
	out.push_back('\t');
	out += std::to_string(tlen);
	out.push_back('\t');
	if (r.plus) out.append((const char *)seq, QL);
	else
		for (unsigned i = 0; i < QL; ++i) out.push_back((char)g_comp[seq[QL - 1 - i]]);
	out.push_back('\t');
	if (!qual) out.push_back('*');
	else if (r.plus) out.append((const char *)qual, QL);
	else
		for (unsigned i = 1; i <= QL; ++i) out.push_back((char)qual[QL - i]);
	out.push_back('\n');
}

void append_sam_header(std::string &out, const urmapx_index *I, int argc, char **argv) {
	const uint32_t n = urmapx_index_seq_count(I);
	for (uint32_t i = 0; i < n; ++i) {
		out += "@SQ\tSN:";
		out += urmapx_index_label(I, i);
		out += "\tLN:";
		out += std::to_string(urmapx_index_seq_length(I, i));
		out.push_back('\n');
	}
	out += "@PG\tID:urmap\tPN:urmap\tVN:1.0.mi355x\tCL:";
	for (int i = 0; i < argc; ++i) { out += argv[i]; out.push_back(' '); }
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
			tlen1 = (int)(r2->coord + len2) - (int)r1->coord;
			if (tlen1 > 0 && tlen1 < 1000 && consistent) proper = true;
			if (tlen1 > 1000) tlen1 = 0;
			tlen2 = -tlen1;
		} else {
			tlen2 = (int)(r1->coord + len1) - (int)r2->coord;
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
