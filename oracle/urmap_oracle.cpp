/*
 * urmap_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see urmap_oracle.h).
 *
 * Scalar CPU restatement of urmap's per-read mapping path.  Every function cites the
 * reference file:line (under /root/reference/src) whose behaviour it follows.  It is
 * written from the behaviour, not from the text, and is deliberately simple: one
 * Searcher per thread, std::vector state, no tuning.
 *
 * Parity: pinned against the unmodified reference binary oracle/_ref/urmap (built by
 * oracle/Makefile from the sources in place) by tests/test_oracle_golden.py and the
 * fixtures under tests/golden/.
 */
#include "urmap_oracle.h"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint8_t byte;

// ---------------------------------------------------------------------------------------
// alphabet: alpha.cpp:1309 (g_CharToLetterNucleo), :3005 (g_CharToCompChar),
// :3525 (g_CharToCompLetter)
// ---------------------------------------------------------------------------------------
static byte g_Letter[256];
static byte g_CompLetter[256];
static byte g_CompChar[256];

static struct AlphaInit {
	AlphaInit() {
		memset(g_Letter, 0xff, 256);
		memset(g_CompLetter, 0xff, 256);
		memset(g_CompChar, '?', 256);
		const char *nt = "ACGTU";
		const byte code[] = {0, 1, 2, 3, 3};
		for (int i = 0; i < 5; ++i) {
			g_Letter[(byte)nt[i]] = code[i];
			g_Letter[(byte)tolower(nt[i])] = code[i];
			g_CompLetter[(byte)nt[i]] = 3 - code[i];
			if (nt[i] != 'U')  // 'u' has no entry in the comp-letter table
				g_CompLetter[(byte)tolower(nt[i])] = 3 - code[i];
		}
		const char *from = "ABCDGHKMNRSTUVWXY";
		const char *to = "TVGHCDMKNYSAABWXR";
		for (int i = 0; from[i]; ++i) {
			g_CompChar[(byte)from[i]] = to[i];
			if (from[i] != 'U')  // 'u' is '?' in the comp-char table
				g_CompChar[(byte)tolower(from[i])] = (byte)tolower(to[i]);
		}
	}
} g_AlphaInit;

// ---------------------------------------------------------------------------------------
// tally encoding, ufindex.h:23-34
// ---------------------------------------------------------------------------------------
static const byte TALLY_FREE = 0;
static const byte TALLY_END = 127;
static const byte TALLY_MY_BIT = 128;
static const byte TALLY_PLUS1 = 254;
static const byte TALLY_BOTH1 = 255;
static const byte TALLY_MAX_NEXT = 124;
static const byte TALLY_NEXT_MASK = 127;
static const byte TALLY_NEXT_LONG_MINE = 253;
static const byte TALLY_NEXT_LONG_OTHER = 125;
static const unsigned MAX_LINK_STEP = 0xffff;
static const unsigned PADGAP = 32;

// state1.h:13-19
static const int SECONDARY_HIT_MAX_DELTA = 12;
static const unsigned PRIME_STRIDE = 27;
static const unsigned SCANK = 4;
static const int MAX_TL = 1000;

// tracebit.h
static const byte TB_DM = 1, TB_IM = 2, TB_MD = 4, TB_MI = 8;

static const float MINUS_INF = -9e9f;  // myutils.h MINUS_INFINITY as used by viterbi.cpp
static const unsigned SEQ_TAIL_PAD = 4096;  // zero bytes after seqdata: reference reads past the end

// ufindex.h:50-65
static inline uint64_t murmur64(uint64_t h) {
	h ^= (h >> 33);
	h *= 0xff51afd7ed558ccdULL;
	h ^= (h >> 33);
	h *= 0xc4ceb9fe1a85ec53ULL;
	h ^= (h >> 33);
	return h;
}

struct uo_index {
	uint32_t W = 0, MaxIx = 0;
	uint64_t SlotCount = 0;
	uint64_t ShiftMask = 0;
	const byte *Blob = nullptr;
	const byte *SeqData = nullptr;
	uint32_t SeqDataSize = 0;
	std::vector<std::string> Labels;
	std::vector<uint32_t> SeqLengths, Offsets;
	// owned storage (empty when wrapping)
	byte *OwnBlob = nullptr;
	byte *OwnSeq = nullptr;

	void SetShiftMask() {  // ufindex.cpp:20-24
		ShiftMask = 0;
		for (unsigned i = 0; i < 2u * W; ++i) ShiftMask |= (uint64_t(1) << i);
	}
	inline uint64_t WordToSlot(uint64_t Word) const { return murmur64(Word) % SlotCount; }
	inline byte Tally(uint64_t Slot) const { return Blob[5 * Slot]; }
	inline uint32_t Pos(uint64_t Slot) const {
		uint32_t p;
		memcpy(&p, Blob + 5 * Slot + 1, 4);
		return p;
	}
};

static inline bool TallyOther(byte T) { return (T & TALLY_MY_BIT) == 0; }

extern "C" int uo_params_for_method(unsigned method, uo_params *p) {
	// state1.cpp:147-183
	if (method == 6 || method == 8) {
		*p = uo_params{-3, -5, -1, 20, 60, 9, 100, 1, 1, 1, 12};
		return 0;
	}
	if (method == 7) {
		*p = uo_params{-4, -6, -2, 35, 35, 12, 75, 8, 6, 5, 8};
		return 0;
	}
	return -1;
}

// ---------------------------------------------------------------------------------------
// index IO: ufindexio.cpp:14-115
// ---------------------------------------------------------------------------------------
static const uint32_t MAGIC1 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '1';
static const uint32_t MAGIC2 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '2';
static const uint32_t MAGIC3 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '3';
static const uint32_t MAGIC5 = ('U' << 24) | ('F' << 16) | ('I' << 8) | '5';

static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

extern "C" uo_index *uo_index_load(const char *path, const char **err) {
	static const char *e_open = "cannot open .ufi", *e_fmt = "bad .ufi format", *e_mem = "out of memory";
	FILE *f = fopen(path, "rb");
	if (!f) { if (err) *err = e_open; return nullptr; }
	uo_index *X = new uo_index;
	uint32_t u = 0, SeqCount = 0;
	bool ok = rd(f, &u, 4) && u == MAGIC1 && rd(f, &X->W, 4) && rd(f, &X->MaxIx, 4) &&
	          rd(f, &X->SeqDataSize, 4) && rd(f, &X->SlotCount, 8) && rd(f, &SeqCount, 4);
	for (uint32_t i = 0; ok && i < SeqCount; ++i) {
		uint32_t L, Off, n;
		ok = rd(f, &L, 4) && rd(f, &Off, 4) && rd(f, &n, 4) && n < (1u << 20);
		if (!ok) break;
		std::string s(n, '\0');
		ok = rd(f, &s[0], n);
		// the reference builds the label with string(char*), i.e. up to the first NUL
		s = std::string(s.c_str());
		X->SeqLengths.push_back(L);
		X->Offsets.push_back(Off);
		X->Labels.push_back(s);
	}
	ok = ok && rd(f, &u, 4) && u == MAGIC2;
	if (!ok) { fclose(f); delete X; if (err) *err = e_fmt; return nullptr; }
	X->SetShiftMask();
	X->OwnBlob = (byte *)malloc(5 * X->SlotCount + 16);
	X->OwnSeq = (byte *)calloc(size_t(X->SeqDataSize) + SEQ_TAIL_PAD, 1);
	if (!X->OwnBlob || !X->OwnSeq) { fclose(f); uo_index_free(X); if (err) *err = e_mem; return nullptr; }
	memset(X->OwnBlob + 5 * X->SlotCount, 0, 16);
	ok = rd(f, X->OwnBlob, 5 * X->SlotCount) && rd(f, &u, 4) && u == MAGIC3 &&
	     rd(f, X->OwnSeq, X->SeqDataSize) && rd(f, &u, 4) && u == MAGIC5;
	fclose(f);
	if (!ok) { uo_index_free(X); if (err) *err = e_fmt; return nullptr; }
	X->Blob = X->OwnBlob;
	X->SeqData = X->OwnSeq;
	return X;
}

extern "C" uo_index *uo_index_wrap(uint32_t W, uint32_t MaxIx, uint64_t SlotCount, const uint8_t *blob,
                                   const uint8_t *seqdata, uint32_t seqdata_size, uint32_t seq_count,
                                   const uint32_t *seq_lengths, const uint32_t *offsets, const char *labels) {
	uo_index *X = new uo_index;
	X->W = W; X->MaxIx = MaxIx; X->SlotCount = SlotCount;
	X->SetShiftMask();
	X->Blob = blob;
	// copy the sequence so that reads past the end see zero bytes
	X->OwnSeq = (byte *)calloc(size_t(seqdata_size) + SEQ_TAIL_PAD, 1);
	memcpy(X->OwnSeq, seqdata, seqdata_size);
	X->SeqData = X->OwnSeq;
	X->SeqDataSize = seqdata_size;
	const char *p = labels;
	for (uint32_t i = 0; i < seq_count; ++i) {
		X->SeqLengths.push_back(seq_lengths[i]);
		X->Offsets.push_back(offsets[i]);
		X->Labels.push_back(std::string(p));
		p += strlen(p) + 1;
	}
	return X;
}

extern "C" void uo_index_free(uo_index *X) {
	if (!X) return;
	free(X->OwnBlob);
	free(X->OwnSeq);
	delete X;
}

extern "C" int uo_index_save(const uo_index *X, const char *path) {  // ufindexio.cpp:14-49
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	uint32_t u = MAGIC1;
	fwrite(&u, 4, 1, f);
	fwrite(&X->W, 4, 1, f);
	fwrite(&X->MaxIx, 4, 1, f);
	fwrite(&X->SeqDataSize, 4, 1, f);
	fwrite(&X->SlotCount, 8, 1, f);
	uint32_t SeqCount = (uint32_t)X->Labels.size();
	fwrite(&SeqCount, 4, 1, f);
	for (uint32_t i = 0; i < SeqCount; ++i) {
		fwrite(&X->SeqLengths[i], 4, 1, f);
		fwrite(&X->Offsets[i], 4, 1, f);
		u = (uint32_t)X->Labels[i].size();
		fwrite(&u, 4, 1, f);
		fwrite(X->Labels[i].data(), 1, u, f);
	}
	u = MAGIC2; fwrite(&u, 4, 1, f);
	fwrite(X->Blob, 1, 5 * X->SlotCount, f);
	u = MAGIC3; fwrite(&u, 4, 1, f);
	fwrite(X->SeqData, 1, X->SeqDataSize, f);
	u = MAGIC5; fwrite(&u, 4, 1, f);
	return fclose(f) == 0 ? 0 : -1;
}

extern "C" uint32_t uo_index_word_length(const uo_index *X) { return X->W; }
extern "C" uint32_t uo_index_max_ix(const uo_index *X) { return X->MaxIx; }
extern "C" uint64_t uo_index_slot_count(const uo_index *X) { return X->SlotCount; }
extern "C" uint32_t uo_index_seqdata_size(const uo_index *X) { return X->SeqDataSize; }
extern "C" const uint8_t *uo_index_blob(const uo_index *X) { return X->Blob; }
extern "C" const uint8_t *uo_index_seqdata(const uo_index *X) { return X->SeqData; }
extern "C" uint32_t uo_index_seq_count(const uo_index *X) { return (uint32_t)X->Labels.size(); }
extern "C" const char *uo_index_label(const uo_index *X, uint32_t i) { return X->Labels[i].c_str(); }
extern "C" uint32_t uo_index_seq_length(const uo_index *X, uint32_t i) { return X->SeqLengths[i]; }
extern "C" uint32_t uo_index_seq_offset(const uo_index *X, uint32_t i) { return X->Offsets[i]; }

// ---------------------------------------------------------------------------------------
// index build: ufindex.cpp:83-151 (MakeIndex), :153-192 (TruncateSlot), :194-322 (UpdateSlot),
// :338-408 (CountSlots, CountSlots_Minus), :462-511 (ReadSeqData), :945-1000
// (FindEndOfList, FindFreeSlot); FASTA parsing fastaseqsource.cpp:27-117
// ---------------------------------------------------------------------------------------
namespace {
struct Builder {
	uo_index *X;
	byte *Blob;
	std::vector<byte> CntPlus, CntMinus;
	unsigned Truncated = 0;

	byte Tally(uint64_t s) const { return Blob[5 * s]; }
	uint32_t Pos(uint64_t s) const { uint32_t p; memcpy(&p, Blob + 5 * s + 1, 4); return p; }
	void SetTally(uint64_t s, byte t) { Blob[5 * s] = t; }
	void SetPos(uint64_t s, uint32_t p) { memcpy(Blob + 5 * s + 1, &p, 4); }
	void SetTallyNext(uint64_t s, byte n) { SetTally(s, (Tally(s) & TALLY_MY_BIT) | n); }

	// follow one link; returns false at end of chain
	bool Step(uint64_t &Slot2, byte T, uint32_t P) const {
		if (T == TALLY_PLUS1 || T == TALLY_BOTH1 || T == TALLY_END) return false;
		const uint64_t N = X->SlotCount;
		if (T == TALLY_NEXT_LONG_MINE || T == TALLY_NEXT_LONG_OTHER) {
			uint64_t SlotA = (Slot2 + (P & 0xffff)) % N;
			Slot2 = (SlotA + (P >> 16)) % N;
		} else
			Slot2 = (Slot2 + (T & TALLY_NEXT_MASK)) % N;
		return true;
	}
	uint64_t FindEndOfList(uint64_t Slot) const {
		uint64_t s = Slot;
		while (Step(s, Tally(s), Pos(s))) {}
		return s;
	}
	unsigned FindFreeSlot(uint64_t Slot) const {
		for (unsigned i = 1; i < MAX_LINK_STEP; ++i) {
			uint64_t s = (Slot + i) % X->SlotCount;
			byte n = CntPlus[s];
			if (n > 0 && n <= X->MaxIx) continue;
			if (Tally(s) == TALLY_FREE) return i;
		}
		return UINT_MAX;
	}
	void TruncateSlot(uint64_t Slot) {
		++Truncated;
		uint64_t s = Slot;
		for (;;) {
			byte T = Tally(s);
			uint32_t P = Pos(s);
			SetTally(s, TALLY_FREE);
			SetPos(s, UINT32_MAX);
			if (!Step(s, T, P)) return;
		}
	}
	void UpdateSlot(uint64_t Slot, uint32_t P) {
		byte n = CntPlus[Slot], nm = CntMinus[Slot];
		if (n > X->MaxIx || nm > X->MaxIx) return;
		if (Tally(Slot) == TALLY_FREE) {
			SetPos(Slot, P);
			SetTally(Slot, (n == 1 && nm == 0) ? TALLY_BOTH1 : TALLY_PLUS1);
			return;
		}
		uint64_t EOL = FindEndOfList(Slot);
		unsigned Step1 = FindFreeSlot(EOL);
		if (Step1 == UINT_MAX) { TruncateSlot(Slot); return; }
		uint64_t Free1 = (EOL + Step1) % X->SlotCount;
		if (Step1 > TALLY_MAX_NEXT) {
			unsigned Step2 = FindFreeSlot(Free1);
			if (Step2 == UINT_MAX) { TruncateSlot(Slot); return; }
			uint32_t EOLPos = Pos(EOL);
			uint64_t Free2 = (Free1 + Step2) % X->SlotCount;
			SetTallyNext(EOL, EOL == Slot ? TALLY_NEXT_LONG_MINE : TALLY_NEXT_LONG_OTHER);
			SetPos(EOL, Step1 | (Step2 << 16));
			SetTally(Free1, TALLY_NEXT_LONG_OTHER);
			SetPos(Free1, EOLPos);
			SetTally(Free2, TALLY_END);
			SetPos(Free2, P);
			return;
		}
		SetTallyNext(EOL, byte(Step1));
		SetTally(Free1, TALLY_END);
		SetPos(Free1, P);
	}
};

static bool ReadFasta(const char *path, std::vector<std::string> &Labels, std::vector<std::string> &Seqs) {
	FILE *f = fopen(path, "rb");
	if (!f) return false;
	std::string line, cur;
	bool have = false;
	auto flush = [&]() {
		if (have && !cur.empty()) Seqs.push_back(cur);
		else if (have) Labels.pop_back();  // empty sequences are skipped
		cur.clear();
	};
	int c;
	line.clear();
	bool eof = false;
	while (!eof) {
		line.clear();
		while ((c = fgetc(f)) != EOF && c != '\n')
			if (c != '\r') line.push_back((char)c);
		if (c == EOF) { eof = true; if (line.empty()) break; }
		if (!line.empty() && line[0] == '>') {
			flush();
			have = true;
			size_t n = 1;
			while (n < line.size() && !isspace((byte)line[n])) ++n;  // make_ufi forces trunclabels
			Labels.push_back(line.substr(1, n - 1));
		} else if (have) {
			for (char ch : line) {
				byte b = (byte)ch;
				if (isspace(b)) continue;
				if (b == '-' || b == '.') continue;  // StripGaps
				if (!isalpha(b)) continue;           // BadByte: warned and dropped
				cur.push_back((char)toupper(b));
			}
		}
	}
	flush();
	fclose(f);
	return true;
}
}  // namespace

extern "C" uo_index *uo_index_build(const char *fasta, uint32_t W, uint32_t MaxIx, uint64_t slots, const char **err) {
	static const char *e_open = "cannot open FASTA", *e_slots = "slots must be given";
	if (slots == 0) { if (err) *err = e_slots; return nullptr; }
	std::vector<std::string> Labels, Seqs;
	if (!ReadFasta(fasta, Labels, Seqs)) { if (err) *err = e_open; return nullptr; }
	uo_index *X = new uo_index;
	X->W = W; X->MaxIx = MaxIx; X->SlotCount = slots;
	X->SetShiftMask();
	uint32_t Size = 0;
	for (size_t i = 0; i < Seqs.size(); ++i) {
		X->Labels.push_back(Labels[i]);
		X->SeqLengths.push_back((uint32_t)Seqs[i].size());
		X->Offsets.push_back(Size);
		Size += (uint32_t)Seqs[i].size();
		if (i + 1 != Seqs.size()) Size += PADGAP;
	}
	X->SeqDataSize = Size;
	X->OwnSeq = (byte *)calloc(size_t(Size) + SEQ_TAIL_PAD, 1);
	uint32_t Off = 0;
	for (size_t i = 0; i < Seqs.size(); ++i) {
		memcpy(X->OwnSeq + Off, Seqs[i].data(), Seqs[i].size());
		Off += (uint32_t)Seqs[i].size();
		if (i + 1 != Seqs.size()) { memset(X->OwnSeq + Off, '-', PADGAP); Off += PADGAP; }
	}
	X->SeqData = X->OwnSeq;
	X->OwnBlob = (byte *)malloc(5 * slots + 16);
	memset(X->OwnBlob + 5 * slots, 0, 16);
	X->Blob = X->OwnBlob;

	Builder B;
	B.X = X;
	B.Blob = X->OwnBlob;
	for (uint64_t s = 0; s < slots; ++s) { B.SetTally(s, TALLY_FREE); B.SetPos(s, UINT32_MAX); }
	B.CntPlus.assign(slots, 0);
	B.CntMinus.assign(slots, 0);
	const byte *S = X->SeqData;
	{  // CountSlots
		uint64_t Word = 0; byte K = 0;
		for (uint32_t p = 0; p < Size; ++p) {
			byte L = g_Letter[S[p]];
			if (L == 0xff) { K = 0; Word = 0; continue; }
			if (K < W) ++K;
			Word = (Word << 2) | L;
			if (K == W) { uint64_t s = X->WordToSlot(Word & X->ShiftMask); if (B.CntPlus[s] < 255) ++B.CntPlus[s]; }
		}
	}
	{  // CountSlots_Minus: backwards with complement letters
		uint64_t Word = 0; byte K = 0;
		for (uint32_t p = Size; p-- > 0;) {
			byte L = g_CompLetter[S[p]];
			if (L == 0xff) { K = 0; Word = 0; continue; }
			if (K < W) ++K;
			Word = (Word << 2) | L;
			if (K == W) { uint64_t s = X->WordToSlot(Word & X->ShiftMask); if (B.CntMinus[s] < 255) ++B.CntMinus[s]; }
		}
	}
	{  // insertion pass, genome order
		uint64_t Word = 0; byte K = 0;
		for (uint32_t p = 0; p < Size; ++p) {
			byte L = g_Letter[S[p]];
			if (L == 0xff) { K = 0; Word = 0; continue; }
			if (K < W) ++K;
			Word = (Word << 2) | L;
			if (K == W) B.UpdateSlot(X->WordToSlot(Word & X->ShiftMask), p - (W - 1));
		}
	}
	return X;
}

// ---------------------------------------------------------------------------------------
// seeding: state1.cpp:396-438 (SetSlotsVec), seqinfo.cpp:9-27 (RevCompSeq)
// ---------------------------------------------------------------------------------------
static void SetSlotsVec(const uo_index *X, const byte *Seq, unsigned L, uint64_t *Slots) {
	const unsigned W = X->W;
	uint64_t Word = 0;
	byte K = 0;
	for (unsigned p = 0; p < L; ++p) {
		byte Letter = g_Letter[Seq[p]];
		bool out = p >= W - 1;
		if (Letter == 0xff) {
			K = 0; Word = 0;
			if (out) Slots[p - (W - 1)] = UINT64_MAX;
			continue;
		}
		if (K < W) ++K;
		Word = (Word << 2) | Letter;
		if (out) Slots[p - (W - 1)] = (K == W) ? X->WordToSlot(Word & X->ShiftMask) : UINT64_MAX;
	}
}

static void RevComp(const byte *Seq, unsigned L, byte *RC) {
	for (unsigned i = 0; i < L; ++i) RC[i] = g_CompChar[Seq[L - 1 - i]];
}

extern "C" void uo_slots_vec(const uo_index *X, const uint8_t *seq, uint32_t L, uint64_t *slots) { SetSlotsVec(X, seq, L, slots); }
extern "C" void uo_revcomp(const uint8_t *seq, uint32_t L, uint8_t *out) { RevComp(seq, L, out); }

// ---------------------------------------------------------------------------------------
// Viterbi: viterbi.cpp:11-261, band limits diagbox.h:150-170, traceback tracebackbitmem.cpp:8-75
// ---------------------------------------------------------------------------------------
namespace {
struct DP {
	std::vector<float> MrowBuf, DrowBuf;
	std::vector<byte> TB;  // (LA+1) x (LB+1)
	uint64_t Cells = 0;

	// returns score; Path gets the M/D/I string (D = query-only column, I = target-only)
	float Viterbi(const uo_params &P, const byte *A, unsigned LA, const byte *B, unsigned LB, bool Left, bool Right,
	              std::string &Path) {
		Path.clear();
		const float GO = float(P.gap_open_score), GE = float(P.gap_ext_score);
		if (LA == 0 || LB == 0) {
			if (LA == 0 && LB == 0) return 0.0f;
			if (LA == 0) { Path.assign(LB, 'I'); return float(P.gap_open_score + int(LB - 1) * P.gap_ext_score); }
			Path.assign(LA, 'D');
			return float(P.gap_open_score + int(LA - 1) * P.gap_ext_score);
		}
		unsigned dlo = std::min(LA, LB), dhi = std::max(LA, LB);
		const unsigned R = P.band_radius;
		dlo = dlo > R ? dlo - R : 1;
		dhi += R;
		if (dhi > LA + LB - 1) dhi = LA + LB - 1;
		auto range_j = [&](unsigned i, unsigned &Startj, unsigned &Endj) {
			Startj = (dlo + i >= LA) ? dlo + i - LA : 0;
			if (Startj >= LB) Startj = LB - 1;
			Endj = (dhi + i + 1 >= LA) ? dhi + i + 1 - LA : 0;
			if (Endj > LB) Endj = LB;
		};
		MrowBuf.assign(LB + 3, MINUS_INF);
		DrowBuf.assign(LB + 3, MINUS_INF);
		float *Mrow = MrowBuf.data() + 1, *Drow = DrowBuf.data() + 1;
		const size_t stride = LB + 1;
		TB.assign(size_t(LA + 1) * stride, 0);

		float OpenA = Left ? 0.0f : GO, ExtA = Left ? 0.0f : GE;
		for (unsigned i = 0; i < LA; ++i) {
			unsigned Startj, Endj;
			range_j(i, Startj, Endj);
			if (Endj == 0) continue;
			float OpenB = (Startj == 0 && Left) ? 0.0f : GO;
			float ExtB = (Startj == 0 && Left) ? 0.0f : GE;
			byte a = A[i];
			float I0 = MINUS_INF;
			float M0 = (i == 0) ? 0.0f : (Startj == 0 ? MINUS_INF : Mrow[int(Startj) - 1]);
			byte *TBrow = TB.data() + i * stride;
			if (Startj > 0) TBrow[Startj - 1] = TB_IM;
			for (unsigned j = Startj; j < Endj; ++j) {
				byte bits = 0;
				const float Mij = M0;
				float xM = Mij;
				if (Drow[j] > xM) { xM = Drow[j]; bits = TB_DM; }
				if (I0 > xM) { xM = I0; bits = TB_IM; }
				M0 = Mrow[j];
				Mrow[j] = xM + float(a == B[j] ? 1 : P.mismatch_score);
				float md = Mij + OpenB;
				Drow[j] += ExtB;
				if (md >= Drow[j]) { Drow[j] = md; bits |= TB_MD; }
				float mi = Mij + OpenA;
				I0 += ExtA;
				if (mi >= I0) { I0 = mi; bits |= TB_MI; }
				OpenB = GO; ExtB = GE;
				TBrow[j] = bits;
				++Cells;
			}
			// end-of-row D update for column LB
			TBrow[LB] = 0;
			float md = M0 + GO;
			Drow[LB] += GE;
			if (md >= Drow[LB]) { Drow[LB] = md; TBrow[LB] = TB_MD; }
			OpenA = GO; ExtA = GE;
		}
		unsigned Startj, Endj;
		range_j(LA - 1, Startj, Endj);
		// last row of the I matrix (strict '>' here, viterbi.cpp:231)
		byte *TBlast = TB.data() + size_t(LA) * stride;
		float I1 = MINUS_INF;
		Mrow[int(Startj) - 1] = MINUS_INF;
		const float GapOp = Right ? 0.0f : GO, GapEx = Right ? 0.0f : GE;
		for (unsigned j = Startj; j < Endj; ++j) {
			TBlast[j] = 0;
			float mi = Mrow[int(j) - 1] + GapOp;
			I1 += GapEx;
			if (mi > I1) { I1 = mi; TBlast[j] = TB_MI; }
		}
		float Score = Mrow[LB - 1];
		char State = 'M';
		if (Drow[LB] > Score) { Score = Drow[LB]; State = 'D'; }
		if (I1 > Score) { Score = I1; State = 'I'; }

		// traceback
		size_t i = LA, j = LB;
		while (i != 0 || j != 0) {
			Path.push_back(State);
			byte t;
			switch (State) {
			case 'M':
				t = TB[(i - 1) * stride + (j - 1)];
				State = (t & TB_DM) ? 'D' : (t & TB_IM) ? 'I' : 'M';
				--i; --j;
				break;
			case 'D':
				t = TB[(i - 1) * stride + j];
				State = (t & TB_MD) ? 'M' : 'D';
				--i;
				break;
			default:
				t = TB[i * stride + (j - 1)];
				State = (t & TB_MI) ? 'M' : 'I';
				--j;
				break;
			}
			if (Path.size() > size_t(LA) + LB + 2) break;  // corrupt trace guard (never hit on valid input)
		}
		std::reverse(Path.begin(), Path.end());
		return Score;
	}
};
}  // namespace

extern "C" float uo_viterbi(const uo_params *P, const uint8_t *A, unsigned LA, const uint8_t *B, unsigned LB, int left,
                            int right, char *path_out) {
	DP dp;
	std::string Path;
	float s = dp.Viterbi(*P, A, LA, B, LB, left != 0, right != 0, Path);
	memcpy(path_out, Path.c_str(), Path.size() + 1);
	return s;
}

// ---------------------------------------------------------------------------------------
// per-read search state: state1.h / state1.cpp / search1.cpp / search1m6.cpp / extendpen.cpp /
// alignhsp.cpp
// ---------------------------------------------------------------------------------------
namespace {
struct Hit {
	uint32_t DBStartPos;
	bool Plus;
	int Score;
	std::string Path;
};
struct HSP {
	uint32_t StartPosQ, StartPosDB, Length;
	int Score;
	bool Plus, Aligned;
};

struct Searcher {
	const uo_index *X;
	uo_params P;
	uo_counters C;
	DP dp;

	const byte *Q = nullptr;
	unsigned QL = 0;
	std::vector<byte> QRC;
	std::vector<Hit> Hits;   // slots; HitCount of them are live (AddHitX writes slot HitCount before deciding)
	std::vector<HSP> HSPs;
	unsigned HitCount = 0, HSPCount = 0;
	int TopHit = -1;
	int MaxPenalty = -1, BestScore = 0, SecondBestScore = 0, BestHSPScore = 0;
	unsigned Mapq = 0;
	unsigned ExitPhase = 0;
	std::vector<uint64_t> SlotsPlus, SlotsMinus;
	std::vector<byte> BlobPlus, BlobMinus;
	std::vector<uint32_t> PosVec;

	Searcher(const uo_index *X_, const uo_params &P_) : X(X_), P(P_) { memset(&C, 0, sizeof C); }

	void GetBlob(uint64_t Slot, byte *out) { memcpy(out, X->Blob + 5 * Slot, 5); ++C.n_getblob; }

	// ufindex.cpp:883-943
	unsigned GetRow_Blob(uint64_t Slot, const byte *blob, uint32_t *PV) {
		++C.n_rowcalls;
		byte T = blob[0];
		if (TallyOther(T)) return 0;
		uint64_t Slot2 = Slot;
		uint32_t Pos;
		memcpy(&Pos, blob + 1, 4);
		unsigned K = 0;
		const uint64_t N = X->SlotCount;
		for (;;) {
			if (K > 0) { T = X->Tally(Slot2); Pos = X->Pos(Slot2); ++C.n_rowhop; }
			PV[K++] = Pos;
			if (K == X->MaxIx) return K;
			if (T == TALLY_PLUS1 || T == TALLY_BOTH1) return 1;
			if (T == TALLY_END) return K;
			if (T == TALLY_NEXT_LONG_MINE || T == TALLY_NEXT_LONG_OTHER) {
				uint64_t SlotA = (Slot2 + (Pos & 0xffff)) % N;
				Slot2 = (SlotA + (Pos >> 16)) % N;
				PV[K - 1] = X->Pos(SlotA);
				++C.n_rowhop;
			} else
				Slot2 = (Slot2 + (T & TALLY_NEXT_MASK)) % N;
		}
	}

	// state1.cpp:230-252
	bool OverlapsHit(uint32_t DBStartPos) const {
		for (unsigned i = 0; i < HitCount; ++i)
			if (DBStartPos / 64 == Hits[i].DBStartPos / 64) return true;
		return false;
	}
	unsigned OverlapsHSP(uint32_t StartPosQ, uint32_t StartPosDB) const {
		int64_t Diag = int64_t(StartPosDB) - int64_t(StartPosQ);
		for (unsigned i = 0; i < HSPCount; ++i)
			if (Diag == int64_t(HSPs[i].StartPosDB) - int64_t(HSPs[i].StartPosQ)) return i;
		return UINT_MAX;
	}

	// state1.cpp:508-551
	unsigned AddHitX(uint32_t StartPosDB, bool Plus, int Score, const std::string &Path) {
		if (Score < 10) return UINT_MAX;
		if (OverlapsHit(StartPosDB)) return UINT_MAX;
		int Pen = int(QL) - Score;
		int MaxPen = Pen - 2 * P.mismatch_score;
		if (MaxPen < MaxPenalty) MaxPenalty = MaxPen;
		unsigned HitIndex = HitCount;
		if (Hits.size() <= HitIndex) Hits.resize(HitIndex + 1);
		Hit &H = Hits[HitIndex];
		H.Score = Score; H.Plus = Plus; H.DBStartPos = StartPosDB; H.Path = Path;
		if (Score > BestScore) {
			SecondBestScore = BestScore;
			BestScore = Score;
			TopHit = int(HitIndex);
		} else if (Score == BestScore)
			SecondBestScore = Score;
		else {
			if (Score < BestScore - SECONDARY_HIT_MAX_DELTA) return UINT_MAX;
			if (Score > SecondBestScore) SecondBestScore = Score;
		}
		++HitCount;
		return HitIndex;
	}

	// state1.cpp:553-591
	void AddHSPX(unsigned StartPosQ, uint32_t StartPosDB, bool Plus, unsigned Length, int Score) {
		if (Score < BestScore - 4) return;
		unsigned k = OverlapsHSP(StartPosQ, StartPosDB);
		if (k != UINT_MAX) {
			if (Score > HSPs[k].Score) HSPs[k] = HSP{StartPosQ, StartPosDB, Length, Score, Plus, false};
			return;
		}
		if (HSPs.size() <= HSPCount) HSPs.resize(HSPCount + 1);
		HSPs[HSPCount++] = HSP{StartPosQ, StartPosDB, Length, Score, Plus, false};
		if (Score > BestHSPScore) BestHSPScore = Score;
	}

	// extendpen.cpp:9-95
	int ExtendPen(uint32_t SeedPosQ, uint32_t SeedPosDB, bool Plus) {
		if (SeedPosDB < SeedPosQ) return -1;
		uint32_t DBLo = SeedPosDB - SeedPosQ;
		if (OverlapsHit(DBLo)) return -1;
		const byte *QSeq = Plus ? Q : QRC.data();
		const byte *DBSeq = X->SeqData + DBLo;
		const int MinHSPScore = int(unsigned(P.min_hsp_score_pct) * QL / 100.0);
		const int W = int(X->W);
		++C.n_extend;
		int Pen = 0, Score = W, Best = 0;
		int EndPos = int(SeedPosQ) + W - 1;
		for (int p = EndPos + 1; p < int(QL); ++p) {
			++C.n_extbases;
			if (QSeq[p] == DBSeq[p]) {
				if (++Score > Best) { Best = Score; EndPos = p; }
			} else {
				Pen -= P.mismatch_score;
				if (Pen > MaxPenalty) return -1;
				Score += P.mismatch_score;
				if (Best - Score > P.xdrop) break;
			}
		}
		int StartPos = int(SeedPosQ);
		for (int p = StartPos - 1; p >= 0; --p) {
			++C.n_extbases;
			if (QSeq[p] == DBSeq[p]) {
				if (++Score > Best) { Best = Score; StartPos = p; }
			} else {
				Pen -= P.mismatch_score;
				if (Pen > MaxPenalty) return -1;
				Score += P.mismatch_score;
				if (Best - Score > P.xdrop) break;
			}
		}
		if (StartPos == 0 && EndPos == int(QL) - 1) {
			AddHitX(DBLo, Plus, Best, std::string());
			return Best;
		}
		if (Best >= MinHSPScore) {
			AddHSPX(unsigned(StartPos), DBLo + unsigned(StartPos), Plus, unsigned(EndPos - StartPos + 1), Best);
			return -2;
		}
		return -1;
	}

	// alignhsp.cpp:60-172
	unsigned AlignHSP(unsigned HSPIndex) {
		HSP &H = HSPs[HSPIndex];
		if (H.Aligned) return UINT_MAX;
		H.Aligned = true;
		int TotalPen = int(H.Length) - H.Score;
		int TotalScore = H.Score;
		if (TotalPen > MaxPenalty) return UINT_MAX;
		++C.n_alignhsp;
		const unsigned StartPosQ = H.StartPosQ, StartPosDB = H.StartPosDB, Len = H.Length;
		const bool Plus = H.Plus;
		const unsigned TL = X->SeqDataSize;
		const unsigned BR = 2 * P.band_radius;  // BRN*GLOBAL_BAND_RADIUS
		unsigned CombinedTLo = StartPosDB;
		const byte *Qs = Plus ? Q : QRC.data();
		const byte *T = X->SeqData;
		std::string LeftPath, RightPath;

		if (StartPosQ > 0) {
			if (StartPosDB < StartPosQ) return UINT_MAX;
			unsigned LeftQL = StartPosQ;
			unsigned LeftTHi = StartPosDB - 1;
			unsigned LeftTL = LeftQL + BR;
			if (LeftTL >= LeftTHi) return UINT_MAX;
			unsigned LeftTLo = LeftTHi - LeftTL + 1;
			const byte *LeftT = T + LeftTLo;
			for (unsigned i = 0; i < LeftTL; ++i)
				if (LeftT[i] == '-') return UINT_MAX;
			++C.n_viterbi; C.n_dptarget += LeftTL;
			int LeftScore = (int)dp.Viterbi(P, Qs, LeftQL, LeftT, LeftTL, true, false, LeftPath);
			// TrimLeftIs, pathinfo.cpp:153-171
			unsigned nI = 0;
			while (nI < LeftPath.size() && LeftPath[nI] == 'I') ++nI;
			LeftPath.erase(0, nI);
			CombinedTLo = LeftTLo + nI;
			int AllGap = P.gap_open_score + int(LeftQL - 1) * P.gap_ext_score;
			if (AllGap > LeftScore) LeftScore = AllGap;
			TotalScore += LeftScore;
			TotalPen += int(LeftQL) - LeftScore;
			if (TotalPen > MaxPenalty) return UINT_MAX;
		}
		const unsigned RightQLo = StartPosQ + Len;
		if (RightQLo < QL) {
			unsigned RightQL = QL - RightQLo;
			unsigned RightTLo = StartPosDB + Len;
			unsigned RightTHi = RightTLo + RightQL + BR;
			if (RightTHi >= TL) RightTHi = TL - 1;
			unsigned RightTL = RightTHi - RightTLo + 1;
			const byte *RightT = T + RightTLo;
			for (unsigned i = 0; i < RightTL; ++i)
				if (RightT[i] == '-') return UINT_MAX;
			++C.n_viterbi; C.n_dptarget += RightTL;
			int RightScore = (int)dp.Viterbi(P, Qs + RightQLo, RightQL, RightT, RightTL, false, true, RightPath);
			// TrimRightIs, pathinfo.cpp:173-190: never trims index 0
			while (RightPath.size() > 1 && RightPath.back() == 'I') RightPath.pop_back();
			int AllGap = P.gap_open_score + int(RightQL - 1) * P.gap_ext_score;
			if (AllGap > RightScore) RightScore = AllGap;
			TotalScore += RightScore;
			TotalPen += int(RightQL) - RightScore;
			if (TotalPen > MaxPenalty) return UINT_MAX;
		}
		std::string Path = LeftPath;
		Path.append(Len, 'M');
		Path += RightPath;
		return AddHitX(CombinedTLo, Plus, TotalScore, Path);
	}

	// search1m6.cpp:9-33
	unsigned CalcMAPQ6() const {
		if (HitCount == 0) return 0;
		if (BestScore <= 0) return 0;
		double BestPossible = double(QL);
		double Second = double(SecondBestScore);
		if (Second < BestPossible / 2.0) {
			Second = BestPossible / 2.0;
			if (BestScore <= Second) return 0;
		}
		double Fract = double(BestScore) / BestPossible;
		double Drop = BestScore - Second;
		if (Drop > 40) Drop = 40;
		unsigned mapq = (unsigned)(Drop * Fract * Fract);
		if (mapq > 40) mapq = 40;
		return mapq;
	}

	void SetQuery(const byte *Seq, unsigned L) {
		Q = Seq; QL = L;
		QRC.resize(L);
		RevComp(Seq, L, QRC.data());
	}

	void ResetHits() {
		HitCount = 0; HSPCount = 0; TopHit = -1; BestScore = 0; SecondBestScore = 0; Mapq = unsigned(-1);
	}


	// =============================== paired-end additions (State1 side) ===============================
	std::vector<uint8_t> PendPlus, PendMinus;  // m_QPosPendingVec_*: query positions stored in a BYTE (state1.h:86-87)
	unsigned PendCountPlus = 0, PendCountMinus = 0;

	// state1.cpp:95-127
	void InitPE(const byte *Seq, unsigned L) {
		SetQuery(Seq, L);
		++C.n_reads; C.n_qbases += L;
		SlotsPlus.assign(L, 0); SlotsMinus.assign(L, 0);
		BlobPlus.assign(5 * L, 0); BlobMinus.assign(5 * L, 0);
		PendPlus.assign(L, 0); PendMinus.assign(L, 0);
		PendCountPlus = PendCountMinus = 0;
		PosVec.resize(X->MaxIx + 1);
		if (L >= X->W) {
			SetSlotsVec(X, Q, L, SlotsPlus.data());
			SetSlotsVec(X, QRC.data(), L, SlotsMinus.data());
		}
		HitCount = 0; HSPCount = 0; TopHit = -1;
		BestScore = 0; BestHSPScore = 0; SecondBestScore = 0;
		Mapq = unsigned(-1);
		MaxPenalty = P.max_penalty;
		ExitPhase = 0;
	}

	// getseed.cpp:9-54
	unsigned GetFirstBoth1Seed(uint32_t &QPos, bool &Plus, uint32_t &DBPos) {
		const unsigned QWC = QL - (X->W - 1);
		for (unsigned k = 0; k < QWC; ++k) {
			QPos = (k * PRIME_STRIDE) % QWC;
			for (int s = 0; s < 2; ++s) {
				const bool plus = (s == 0);
				uint64_t Slot = (plus ? SlotsPlus : SlotsMinus)[QPos];
				if (Slot == UINT64_MAX) continue;
				byte *B = &(plus ? BlobPlus : BlobMinus)[5 * QPos];
				GetBlob(Slot, B);
				byte T = B[0];
				if (TallyOther(T)) continue;
				if (T != TALLY_BOTH1) {
					if (plus) PendPlus[PendCountPlus++] = (uint8_t)QPos;
					else PendMinus[PendCountMinus++] = (uint8_t)QPos;
					continue;
				}
				memcpy(&DBPos, B + 1, 4);
				Plus = plus;
				return k;
			}
		}
		return UINT_MAX;
	}

	// getseed.cpp:56-138
	unsigned GetNextBoth1Seed(unsigned ak, uint32_t &aQPos, bool &Plus, uint32_t &DBPos) {
		const unsigned QWC = QL - (X->W - 1);
		if (Plus) {  // the minus strand of the same k has not been looked at yet
			unsigned QPos = (ak * PRIME_STRIDE) % QWC;
			uint64_t Slot = SlotsMinus[QPos];
			if (Slot != UINT64_MAX) {
				byte *B = &BlobMinus[5 * QPos];
				GetBlob(Slot, B);
				byte T = B[0];
				if (!TallyOther(T) && T == TALLY_BOTH1) {  // a "mine" slot that is not BOTH1 is NOT queued here (quirk)
					uint32_t NewDBPos;
					memcpy(&NewDBPos, B + 1, 4);
					if (uint32_t(NewDBPos - QPos) != uint32_t(DBPos - aQPos)) {
						DBPos = NewDBPos; aQPos = QPos; Plus = false;
						return ak;
					}
					PendMinus[PendCountMinus++] = (uint8_t)QPos;
				}
			}
		}
		for (unsigned k = ak + 1; k < QWC; ++k) {
			unsigned QPos = (k * PRIME_STRIDE) % QWC;
			for (int s = 0; s < 2; ++s) {
				const bool plus = (s == 0);
				uint64_t Slot = (plus ? SlotsPlus : SlotsMinus)[QPos];
				if (Slot == UINT64_MAX) continue;
				byte *B = &(plus ? BlobPlus : BlobMinus)[5 * QPos];
				GetBlob(Slot, B);
				byte T = B[0];
				if (TallyOther(T)) continue;
				if (T != TALLY_BOTH1) {
					if (plus) PendPlus[PendCountPlus++] = (uint8_t)QPos;
					else PendMinus[PendCountMinus++] = (uint8_t)QPos;
					continue;
				}
				uint32_t NewDBPos;
				memcpy(&NewDBPos, B + 1, 4);
				if (uint32_t(NewDBPos - QPos) == uint32_t(DBPos - aQPos)) continue;  // same diagonal as the last seed
				DBPos = NewDBPos; aQPos = QPos; Plus = plus;
				return k;
			}
		}
		return UINT_MAX;
	}

	// search1pepend.cpp:9-130 (the GetNextBoth1SeedEx loop never runs: k is UINT_MAX after Search4's seed loop)
	void SearchPE_Pending() {
		MaxPenalty = P.max_penalty;
		const int MinScorePhase1 = int(QL) + P.xphase1 * P.mismatch_score;
		const int TermHSPScorePhase3 = (int(QL) * P.term_hsp_score_pct_phase3) / 100;
		if (BestScore >= MinScorePhase1) { Mapq = CalcMAPQ6(); return; }
		if (BestHSPScore >= TermHSPScorePhase3) {
			for (unsigned i = 0; i < HSPCount; ++i) AlignHSP(i);
			if (BestScore >= MinScorePhase1) { Mapq = CalcMAPQ6(); return; }
		}
		unsigned Count2[2] = {0, 0};
		for (int s = 0; s < 2; ++s) {  // round 1: rows of length <= 2; longer rows are compacted to the front
			const bool plus = (s == 0);
			std::vector<uint8_t> &Pend = plus ? PendPlus : PendMinus;
			const unsigned n = plus ? PendCountPlus : PendCountMinus;
			for (unsigned i = 0; i < n; ++i) {
				unsigned QPos = Pend[i];
				uint64_t Slot = (plus ? SlotsPlus : SlotsMinus)[QPos];
				unsigned RowLength = GetRow_Blob(Slot, &(plus ? BlobPlus : BlobMinus)[5 * QPos], PosVec.data());
				if (RowLength > 2) { Pend[Count2[s]++] = (uint8_t)QPos; continue; }
				for (unsigned r = 0; r < RowLength; ++r) ExtendPen(QPos, PosVec[r], plus);
			}
		}
		for (int s = 0; s < 2; ++s) {  // round 2
			const bool plus = (s == 0);
			std::vector<uint8_t> &Pend = plus ? PendPlus : PendMinus;
			for (unsigned i = 0; i < Count2[s]; ++i) {
				unsigned QPos = Pend[i];
				uint64_t Slot = (plus ? SlotsPlus : SlotsMinus)[QPos];
				unsigned RowLength = GetRow_Blob(Slot, &(plus ? BlobPlus : BlobMinus)[5 * QPos], PosVec.data());
				for (unsigned r = 0; r < RowLength; ++r) ExtendPen(QPos, PosVec[r], plus);
			}
		}
		const int Bmin = std::max(BestScore, BestHSPScore) - 8;
		for (unsigned i = 0; i < HSPCount; ++i) {
			if (HSPs[i].Score < Bmin) continue;
			AlignHSP(i);
		}
		Mapq = CalcMAPQ6();
	}

	// extendscan.cpp:8-49
	unsigned AddHSPScan(unsigned StartPosQ, uint32_t StartPosDB, bool Plus, unsigned Length, int Score) {
		unsigned k = OverlapsHSP(StartPosQ, StartPosDB);
		if (k != UINT_MAX) {
			if (Score > HSPs[k].Score) HSPs[k] = HSP{StartPosQ, StartPosDB, Length, Score, Plus, false};
			return k;
		}
		if (HSPs.size() <= HSPCount) HSPs.resize(HSPCount + 1);
		k = HSPCount++;
		HSPs[k] = HSP{StartPosQ, StartPosDB, Length, Score, Plus, false};
		if (Score > BestHSPScore) BestHSPScore = Score;
		return k;
	}

	// extendscan.cpp:51-187: no overlap test, and the leftward loop never adds to Pen (quirk)
	unsigned ExtendScan(uint32_t SeedPosQ, uint32_t SeedPosDB, bool Plus) {
		++C.n_extscan;
		if (SeedPosDB < SeedPosQ) return UINT_MAX;
		uint32_t DBLo = SeedPosDB - SeedPosQ;
		const byte *QSeq = Plus ? Q : QRC.data();
		const byte *DBSeq = X->SeqData + DBLo;
		const int W = int(X->W);
		const int MinHSPScore = W * 2;
		int Pen = 0, Score = W, Best = 0;
		int EndPos = int(SeedPosQ) + W - 1;
		for (int p = EndPos + 1; p < int(QL); ++p) {
			if (QSeq[p] == DBSeq[p]) { if (++Score > Best) { Best = Score; EndPos = p; } }
			else {
				Pen -= P.mismatch_score;
				if (Pen > MaxPenalty) return UINT_MAX;
				Score += P.mismatch_score;
				if (Best - Score > P.xdrop) break;
			}
		}
		int StartPos = int(SeedPosQ);
		for (int p = StartPos - 1; p >= 0; --p) {
			if (QSeq[p] == DBSeq[p]) { if (++Score > Best) { Best = Score; StartPos = p; } }
			else {
				if (Pen > MaxPenalty) return UINT_MAX;
				Score += P.mismatch_score;
				if (Best - Score > P.xdrop) break;
			}
		}
		if (StartPos == 0 && EndPos == int(QL) - 1) return AddHitX(DBLo, Plus, Best, std::string());
		if (Best < MinHSPScore) return UINT_MAX;
		unsigned k = AddHSPScan(unsigned(StartPos), DBLo + unsigned(StartPos), Plus, unsigned(EndPos - StartPos + 1), Best);
		return AlignHSP(k);
	}

	// scanslots.cpp:7-62
	void ScanSlots(uint32_t DBLo, unsigned DBSegLength, bool Plus) {
		const unsigned W = X->W;
		const unsigned QWC = QL - (W - 1);
		const std::vector<uint64_t> &Slots = Plus ? SlotsPlus : SlotsMinus;
		if (QL <= W * 4) return;
		const byte *Seg = X->SeqData + DBLo;
		uint64_t Word = 0;
		byte K = 0;
		for (uint32_t p = 0; p < DBSegLength; ++p) {
			byte L = g_Letter[Seg[p]];
			if (L == 0xff) { K = 0; Word = 0; continue; }
			if (K < W) ++K;
			Word = (Word << 2) | L;
			if (p >= W - 1 && K == W) {
				uint64_t Slot = X->WordToSlot(Word & X->ShiftMask);
				for (unsigned k = 0; k < SCANK; ++k) {
					unsigned QPos = (k * PRIME_STRIDE) % QWC;
					if (Slot == Slots[QPos]) ExtendScan(QPos, DBLo + p - W + 1, Plus);
				}
			}
		}
	}

	// scan.cpp:14-39
	void Scan(uint32_t DBPos, unsigned DBSegLength, bool Plus, bool DoVit) {
		int SavedMaxPenalty = MaxPenalty;
		unsigned SavedHitCount = HitCount;
		++C.n_scan;
		MaxPenalty = 130;
		ScanSlots(DBPos, DBSegLength, Plus);
		MaxPenalty = SavedMaxPenalty;
		if (HitCount > SavedHitCount) { C.n_scan_hits += HitCount - SavedHitCount; return; }
		if (!DoVit) return;
		++C.n_scan_vit;
		const byte *Qs = Plus ? Q : QRC.data();
		std::string Path;
		++C.n_viterbi; C.n_dptarget += DBSegLength;
		float Score = dp.Viterbi(P, Qs, QL, X->SeqData + DBPos, DBSegLength, true, true, Path);
		if (Score >= QL / 3.0) {
			unsigned nI = 0;
			while (nI < Path.size() && Path[nI] == 'I') ++nI;
			Path.erase(0, nI);
			while (Path.size() > 1 && Path.back() == 'I') Path.pop_back();
			AddHitX(DBPos + nI, Plus, int(Score), Path);
			C.n_scan_hits += HitCount - SavedHitCount;
		}
	}

	// phases 1+2 body for one (QPos, strand); returns true if Search_Lo must return
	bool SeedBoth1(uint32_t QPos, bool Plus, int MinScorePhase1) {
		std::vector<uint64_t> &Slots = Plus ? SlotsPlus : SlotsMinus;
		std::vector<byte> &Blob = Plus ? BlobPlus : BlobMinus;
		uint64_t Slot = Slots[QPos];
		if (Slot == UINT64_MAX) { Blob[5 * QPos] = TALLY_FREE; return false; }
		GetBlob(Slot, &Blob[5 * QPos]);
		if (Blob[5 * QPos] != TALLY_BOTH1) return false;
		uint32_t SeedPosDB;
		memcpy(&SeedPosDB, &Blob[5 * QPos + 1], 4);
		int Score = ExtendPen(QPos, SeedPosDB, Plus);
		return Score >= MinScorePhase1;
	}

	// search1m6.cpp:35-277
	void Search_Lo() {
		const unsigned W = X->W;
		if (QL < W) { Mapq = 0; ExitPhase = 0; return; }  // outside the reference's domain (it underflows)
		const unsigned QWordCount = QL - (W - 1);
		MaxPenalty = P.max_penalty;
		const int MinScorePhase1 = int(QL) + P.xphase1 * P.mismatch_score;
		const int MinScorePhase3 = int(QL) + P.xphase3 * P.mismatch_score;
		const int MinScorePhase4 = int(QL) + P.xphase4 * P.mismatch_score;
		const int TermHSPScorePhase3 = (int(QL) * P.term_hsp_score_pct_phase3) / 100;
		BestHSPScore = 0;
		SlotsPlus.assign(QL, 0); SlotsMinus.assign(QL, 0);
		BlobPlus.assign(5 * QL, 0); BlobMinus.assign(5 * QL, 0);
		SetSlotsVec(X, Q, QL, SlotsPlus.data());
		SetSlotsVec(X, QRC.data(), QL, SlotsMinus.data());
		PosVec.resize(X->MaxIx + 1);

		// Phase 1: BOTH1 seeds at stride W
		ExitPhase = 1;
		for (uint32_t QPos = 0; QPos < QWordCount; QPos += W) {
			if (SeedBoth1(QPos, true, MinScorePhase1)) { Mapq = CalcMAPQ6(); return; }
			if (SeedBoth1(QPos, false, MinScorePhase1)) { Mapq = CalcMAPQ6(); return; }
		}
		// Phase 2: remaining BOTH1 seeds
		ExitPhase = 2;
		for (uint32_t QPos = 0; QPos < QWordCount; ++QPos) {
			if (QPos % W == 0) continue;
			if (SeedBoth1(QPos, true, MinScorePhase1)) { Mapq = CalcMAPQ6(); return; }
			if (SeedBoth1(QPos, false, MinScorePhase1)) { Mapq = CalcMAPQ6(); return; }
		}
		// Phase 3
		ExitPhase = 3;
		if (BestHSPScore > TermHSPScorePhase3) {
			for (unsigned i = 0; i < HSPCount; ++i) AlignHSP(i);
			if (BestScore >= MinScorePhase1) { Mapq = CalcMAPQ6(); return; }
		}
		// Phase 4: rows of length <= 2
		ExitPhase = 4;
		std::vector<uint32_t> Todo[2];
		for (int s = 0; s < 2; ++s) {
			const bool Plus = (s == 0);
			std::vector<uint64_t> &Slots = Plus ? SlotsPlus : SlotsMinus;
			std::vector<byte> &Blob = Plus ? BlobPlus : BlobMinus;
			for (uint32_t QPos = 0; QPos < QWordCount; ++QPos) {
				byte T = Blob[5 * QPos];
				if (T == TALLY_FREE || T == TALLY_BOTH1 || TallyOther(T)) continue;
				unsigned RowLength = GetRow_Blob(Slots[QPos], &Blob[5 * QPos], PosVec.data());
				if (RowLength > 2) { Todo[s].push_back(QPos); continue; }
				for (unsigned k = 0; k < RowLength; ++k) ExtendPen(QPos, PosVec[k], Plus);
			}
		}
		if (BestScore >= MinScorePhase3) { Mapq = CalcMAPQ6(); return; }
		// Phase 5: longer rows
		ExitPhase = 5;
		for (int s = 0; s < 2; ++s) {
			const bool Plus = (s == 0);
			std::vector<uint64_t> &Slots = Plus ? SlotsPlus : SlotsMinus;
			std::vector<byte> &Blob = Plus ? BlobPlus : BlobMinus;
			for (uint32_t QPos : Todo[s]) {
				unsigned RowLength = GetRow_Blob(Slots[QPos], &Blob[5 * QPos], PosVec.data());
				for (unsigned k = 0; k < RowLength; ++k) ExtendPen(QPos, PosVec[k], Plus);
			}
		}
		if (BestScore >= MinScorePhase4) { Mapq = CalcMAPQ6(); return; }
		// Phase 6
		ExitPhase = 6;
		for (unsigned i = 0; i < HSPCount; ++i) AlignHSP(i);
		Mapq = CalcMAPQ6();
	}

	// ufindex.cpp:729-755; returns coord or UINT32_MAX, sets SeqIndex
	uint32_t PosToCoordL(uint32_t Pos, uint32_t &SeqIndexOut, unsigned &L) const {
		const unsigned SeqCount = (unsigned)X->Labels.size();
		unsigned Lo = 0, Hi = SeqCount - 1;
		while (Lo <= Hi && Hi != UINT_MAX) {
			unsigned k = (Lo + Hi) / 2;
			uint32_t Off = X->Offsets[k], SL = X->SeqLengths[k];
			if (Pos >= Off && Pos < Off + SL) { SeqIndexOut = k; L = SL; return Pos - Off; }
			if (Pos > Off) Lo = k + 1;
			else Hi = k - 1;
		}
		return UINT32_MAX;
	}

	// search1.cpp:7-24 + state1.cpp:129-145
	void Search(const byte *Seq, unsigned L, uo_result &R, std::string &PathOut) {
		SetQuery(Seq, L);
		ResetHits();
		++C.n_reads; C.n_qbases += L;
		Search_Lo();
		Fill(R, PathOut);
	}

	void Fill(uo_result &R, std::string &PathOut) {
		C.n_dpcells += dp.Cells; dp.Cells = 0;
		R.dbpos = UINT32_MAX; R.seq_index = UINT32_MAX; R.coord = UINT32_MAX;
		R.score = BestScore; R.second = SecondBestScore; R.mapq = Mapq;
		R.hit_count = HitCount; R.hsp_count = HSPCount; R.plus = 0; R.exit_phase = (uint8_t)ExitPhase;
		R.path_len = 0; R.path_off = 0;
		PathOut.clear();
		if (TopHit < 0) return;
		const Hit &H = Hits[TopHit];
		unsigned TargetL = 0;
		uint32_t SeqIndex = UINT32_MAX;
		uint32_t Coord = PosToCoordL(H.DBStartPos, SeqIndex, TargetL);
		if (uint32_t(Coord + QL) > TargetL) return;  // SetMappedPos un-maps (also when Coord == UINT32_MAX)
		R.dbpos = H.DBStartPos; R.seq_index = SeqIndex; R.coord = Coord; R.plus = H.Plus;
		PathOut = H.Path;
		R.path_len = (uint16_t)H.Path.size();
	}
};
}  // namespace

extern "C" unsigned uo_get_row(const uo_index *X, uint64_t slot, uint32_t *posvec) {
	uo_params P;
	uo_params_for_method(6, &P);
	Searcher S(X, P);
	return S.GetRow_Blob(slot, X->Blob + 5 * slot, posvec);
}

static void AddCounters(uo_counters *dst, const uo_counters &src) {
	if (!dst) return;
	uint64_t *d = (uint64_t *)dst;
	const uint64_t *s = (const uint64_t *)&src;
	for (size_t i = 0; i < sizeof(uo_counters) / 8; ++i) d[i] += s[i];
}

extern "C" int uo_map_se(const uo_index *X, const uo_params *P, const uint8_t *bases, const uint64_t *offs, uint32_t n,
                         int threads, uo_result *results, char **path_arena, uo_counters *counters) {
	if (threads < 1) threads = 1;
	std::vector<std::string> Paths(n);
	if (counters) memset(counters, 0, sizeof *counters);
#pragma omp parallel num_threads(threads)
	{
		Searcher S(X, *P);
#pragma omp for schedule(dynamic, 256)
		for (int64_t i = 0; i < (int64_t)n; ++i)
			S.Search(bases + offs[i], unsigned(offs[i + 1] - offs[i]), results[i], Paths[i]);
#pragma omp critical
		AddCounters(counters, S.C);
	}
	size_t total = 1;
	for (uint32_t i = 0; i < n; ++i) total += Paths[i].size() + 1;
	char *arena = (char *)malloc(total);
	if (!arena) return -1;
	size_t off = 0;
	for (uint32_t i = 0; i < n; ++i) {
		results[i].path_off = (uint32_t)off;
		memcpy(arena + off, Paths[i].c_str(), Paths[i].size() + 1);
		off += Paths[i].size() + 1;
	}
	if (path_arena) *path_arena = arena;
	else free(arena);
	return 0;
}

extern "C" void uo_free(void *p) { free(p); }

// ---------------------------------------------------------------------------------------
// SAM: setsam.cpp:12-207, cigar.cpp:4-41,141-199, state1.cpp:694-734
// ---------------------------------------------------------------------------------------
static void PathToOps(const char *Path, std::vector<char> &Ops, std::vector<unsigned> &Lens) {
	Ops.clear(); Lens.clear();
	for (const char *p = Path; *p; ++p) {
		char c = (*p == 'D') ? 'I' : (*p == 'I') ? 'D' : *p;  // D<->I swap, cigar.cpp:22-25
		if (!Ops.empty() && Ops.back() == c) ++Lens.back();
		else { Ops.push_back(c); Lens.push_back(1); }
	}
}

static void FixDanglingMs(std::vector<char> &Ops, std::vector<unsigned> &Lens) {  // cigar.cpp:141-199
	size_t N = Ops.size();
	if (N < 3) return;
	if (Ops[0] == 'M' && Lens[0] <= 2 && Lens[1] > 4 && Ops[2] == 'M') {
		Lens[2] += Lens[0];
		Ops.erase(Ops.begin());
		Lens.erase(Lens.begin());
		// The reference keeps the ORIGINAL N for its tail rule after shrinking both vectors
		// (cigar.cpp:149,175): it then looks at index N-1 of vectors of size N-1.  With
		// libstdc++ the copy-assignment reuses the old storage, so the stale last element is
		// compared: Lengths[N-1] (stale) <= 2 and Lengths[N-2] (the same value, now last) > 4
		// cannot both hold, and for N == 3 the merged length is <= 4.  The tail rule therefore
		// never fires once the head rule has.
		return;
	}
	if (Ops[N - 1] == 'M' && Lens[N - 1] <= 2 && Lens[N - 2] > 4 && Ops[N - 3] == 'M') {
		Lens[N - 3] += Lens[N - 1];
		Ops.pop_back();
		Lens.pop_back();
	}
}

static std::string PathToCIGAR(const char *Path, unsigned QL) {
	char tmp[32];
	if (*Path == 0) { snprintf(tmp, sizeof tmp, "%uM", QL); return tmp; }
	std::vector<char> Ops; std::vector<unsigned> Lens;
	PathToOps(Path, Ops, Lens);
	FixDanglingMs(Ops, Lens);
	std::string s;
	for (size_t i = 0; i < Ops.size(); ++i) { snprintf(tmp, sizeof tmp, "%u%c", Lens[i], Ops[i]); s += tmp; }
	return s;
}

static size_t QNameLen(const char *Label) {  // setsam.cpp:87-98
	size_t n = strlen(Label);
	if (n > 2 && Label[n - 2] == '/' && (Label[n - 1] == '1' || Label[n - 1] == '2')) n -= 2;
	size_t k = 0;
	while (k < n && Label[k] != ' ' && Label[k] != '\t') ++k;
	return k;
}

static size_t SamUnmapped(uint32_t aFlags, const char *Label, const byte *Seq, const byte *Qual, unsigned QL, char *buf) {
	uint32_t Flags = 0x04;
	if (aFlags & 0x01) Flags |= 0x01;
	if (aFlags & 0x40) Flags |= 0x40;
	else if (aFlags & 0x80) Flags |= 0x80;
	if (aFlags & 0x08) Flags |= 0x08;
	else if (aFlags & 0x20) Flags |= 0x20;
	char *p = buf;
	size_t n = QNameLen(Label);
	memcpy(p, Label, n); p += n;
	p += sprintf(p, "\t%u\t*\t0\t0\t*\t*\t0\t0\t", Flags);
	memcpy(p, Seq, QL); p += QL;
	*p++ = '\t';
	if (!Qual) *p++ = '*';
	else { memcpy(p, Qual, QL); p += QL; }
	*p++ = '\n';
	*p = 0;
	return size_t(p - buf);
}

// general form used by SE (Flags=0, mate "*") and PE
static size_t SamRecord(const uo_index *X, bool Mapped, uint32_t SeqIndex, uint32_t Coord, bool Plus, unsigned Mapq,
                        const char *Path, uint32_t Flags, const char *MateLabel, uint32_t MatePos, int TLEN,
                        const char *Label, const byte *Seq, const byte *Qual, unsigned QL, char *buf) {
	if (!Mapped) return SamUnmapped(Flags, Label, Seq, Qual, QL, buf);
	char *p = buf;
	size_t n = QNameLen(Label);
	memcpy(p, Label, n); p += n;
	const std::string &TLabel = X->Labels[SeqIndex];
	p += sprintf(p, "\t%u\t%s\t%u\t%u\t", Flags, TLabel.c_str(), Coord + 1, Mapq);
	std::string CIGAR = PathToCIGAR(Path ? Path : "", QL);
	memcpy(p, CIGAR.data(), CIGAR.size()); p += CIGAR.size();
	*p++ = '\t';
	if (MateLabel == nullptr || MateLabel[0] == 0 || strcmp(MateLabel, "*") == 0) *p++ = '*';
	else if (TLabel == MateLabel) *p++ = '=';
	else { n = strlen(MateLabel); memcpy(p, MateLabel, n); p += n; }
	*p++ = '\t';
	if (MatePos == 0 || MatePos == UINT32_MAX) *p++ = '0';
	else p += sprintf(p, "%u", MatePos + 1);
	p += sprintf(p, "\t%d\t", TLEN);
	if (Plus) { memcpy(p, Seq, QL); p += QL; }
	else { for (unsigned i = 0; i < QL; ++i) *p++ = (char)g_CompChar[Seq[QL - 1 - i]]; }
	*p++ = '\t';
	if (!Qual) *p++ = '*';
	else if (Plus) { memcpy(p, Qual, QL); p += QL; }
	else { for (unsigned i = 1; i <= QL; ++i) *p++ = (char)Qual[QL - i]; }
	*p++ = '\n';
	*p = 0;
	return size_t(p - buf);
}

extern "C" size_t uo_sam_se(const uo_index *X, const uo_result *r, const char *path, const char *label,
                            const uint8_t *seq, const uint8_t *qual, uint32_t L, char *buf) {
	return SamRecord(X, r->dbpos != UINT32_MAX, r->seq_index, r->coord, r->plus != 0, r->mapq, path, 0, "*", UINT32_MAX,
	                 0, label, seq, qual, L, buf);
}

// ---------------------------------------------------------------------------------------
// FASTQ: fastqseqsource.cpp:9-116 (labels are not truncated on the -map path; SAM cuts at blank)
// ---------------------------------------------------------------------------------------
namespace {
struct FastqRec { std::string Label, Seq, Qual; };
static bool ReadLine(FILE *f, std::string &s) {
	s.clear();
	int c;
	bool any = false;
	while ((c = fgetc(f)) != EOF) {
		any = true;
		if (c == '\r') continue;
		if (c == '\n') return true;
		s.push_back((char)c);
	}
	return any && !s.empty();
}
static int ReadFastq(const char *path, std::vector<FastqRec> &Recs) {
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	std::string l1, l2, l3, l4;
	while (ReadLine(f, l1)) {
		if (l1.empty()) continue;
		if (l1[0] != '@' || !ReadLine(f, l2)) { fclose(f); return -2; }
		ReadLine(f, l3);
		if (!ReadLine(f, l4) || l4.size() != l2.size()) { fclose(f); return -2; }
		Recs.push_back(FastqRec{l1.substr(1), l2, l4});
	}
	fclose(f);
	return 0;
}
}  // namespace

static void WriteSQ(FILE *f, const uo_index *X) {  // state1.cpp:736-748 (@PG is excluded from comparisons)
	for (size_t i = 0; i < X->Labels.size(); ++i) fprintf(f, "@SQ\tSN:%s\tLN:%u\n", X->Labels[i].c_str(), X->SeqLengths[i]);
}

extern "C" int uo_map_file_se(const uo_index *X, const uo_params *P, const char *fastq, const char *sam, int threads,
                              uo_counters *counters) {
	std::vector<FastqRec> Recs;
	int rc = ReadFastq(fastq, Recs);
	if (rc) return rc;
	uint32_t n = (uint32_t)Recs.size();
	std::vector<uint64_t> offs(n + 1, 0);
	std::string bases;
	for (uint32_t i = 0; i < n; ++i) { bases += Recs[i].Seq; offs[i + 1] = bases.size(); }
	std::vector<uo_result> R(n);
	char *arena = nullptr;
	rc = uo_map_se(X, P, (const uint8_t *)bases.data(), offs.data(), n, threads, R.data(), &arena, counters);
	if (rc) return rc;
	FILE *f = fopen(sam, "wb");
	if (!f) { free(arena); return -3; }
	WriteSQ(f, X);
	std::vector<char> buf;
	for (uint32_t i = 0; i < n; ++i) {
		unsigned L = (unsigned)Recs[i].Seq.size();
		buf.resize(Recs[i].Label.size() + 3 * size_t(L) + 512);
		size_t k = uo_sam_se(X, &R[i], arena + R[i].path_off, Recs[i].Label.c_str(), (const uint8_t *)Recs[i].Seq.data(),
		                     (const uint8_t *)Recs[i].Qual.data(), L, buf.data());
		fwrite(buf.data(), 1, k, f);
	}
	fclose(f);
	free(arena);
	return 0;
}

// ---------------------------------------------------------------------------------------
// paired-end: State2 (state2.h/.cpp), Search4 (search2m4.cpp), AdjustTopHitsAndMapqs (search2.cpp:8-57),
// SetSAM2 / GetPairedFlags (output2.cpp:18-128)
// ---------------------------------------------------------------------------------------
namespace {
struct PairSearcher {
	Searcher F, R;
	std::vector<unsigned> PairF, PairR;
	std::vector<int> PairScore;
	int BestPairScore = 0, SecondBestPairScore = 0;
	unsigned BestPairIndex = 0, SecondPairIndex = UINT_MAX;
	int SecondF = -1, SecondR = -1;  // m_SecondHit of each mate: set by AdjustTopHitsAndMapqs only (search2.cpp:49-56)
	int TermPairScorePhase1 = 0;

	PairSearcher(const uo_index *X, const uo_params &P) : F(X, P), R(X, P) {}

	// search2m4.cpp:189-208
	bool ExtendBoth1Pair4(uint32_t QPosf, uint32_t DBPosf, bool Plusf, uint32_t QPosr, uint32_t DBPosr) {
		int FwdScore = F.ExtendPen(QPosf, DBPosf, Plusf);
		if (FwdScore <= 0) return false;
		int RevScore = R.ExtendPen(QPosr, DBPosr, !Plusf);
		if (RevScore <= 0) return false;
		if (FwdScore + RevScore < TermPairScorePhase1) return false;
		F.Mapq = 40; R.Mapq = 40;
		return true;
	}

	// state2.cpp:20-85
	void FindPairs() {
		PairF.clear(); PairR.clear(); PairScore.clear();
		const unsigned QL2 = (F.QL + R.QL) / 2;
		BestPairIndex = UINT_MAX; SecondPairIndex = UINT_MAX;
		BestPairScore = -1; SecondBestPairScore = -1;
		for (unsigned i = 0; i < F.HitCount; ++i) {
			const Hit &Hf = F.Hits[i];
			if (Hf.Score < F.SecondBestScore - 12) continue;
			for (unsigned j = 0; j < R.HitCount; ++j) {
				const Hit &Hr = R.Hits[j];
				if (Hr.Score < R.SecondBestScore - 12) continue;
				int64_t TL = std::llabs(int64_t(Hf.DBStartPos) - int64_t(Hr.DBStartPos)) + int64_t(QL2);
				if (TL > 1000) continue;
				if (Hr.Plus == Hf.Plus) continue;
				int Total = Hf.Score + Hr.Score;
				const unsigned idx = (unsigned)PairScore.size();
				if (Total > BestPairScore) {
					SecondPairIndex = BestPairIndex; SecondBestPairScore = BestPairScore;
					BestPairScore = Total; BestPairIndex = idx;
				} else if (Total == BestPairScore) {
					SecondPairIndex = idx; SecondBestPairScore = BestPairScore;
				} else if (Total > SecondBestPairScore) {
					SecondPairIndex = BestPairIndex;  // sic (state2.cpp:74): the index of the BEST pair
					SecondBestPairScore = Total;
				}
				PairScore.push_back(Total); PairF.push_back(i); PairR.push_back(j);
			}
		}
	}

	// state2.cpp:87-137
	void ScanPair() {
		const unsigned SEG = 1024;
		const bool DoVitF = int(F.Mapq) >= 10, DoVitR = int(R.Mapq) >= 10;
		const unsigned HCf = F.HitCount, HCr = R.HitCount;
		for (unsigned i = 0; i < HCf; ++i) {
			const unsigned QLx = F.QL;  // sic: the forward read's length is used for the reverse read's window
			const Hit H = F.Hits[i];
			if (H.Score < F.SecondBestScore) continue;
			if (H.Plus) R.Scan(H.DBStartPos, SEG, false, DoVitF);
			else if (H.DBStartPos >= SEG) R.Scan(H.DBStartPos - SEG, SEG + 2 * QLx, true, DoVitF);
		}
		for (unsigned j = 0; j < HCr; ++j) {
			const unsigned QLx = F.QL;
			const Hit H = R.Hits[j];
			if (H.Score < R.SecondBestScore) continue;
			if (H.Plus) F.Scan(H.DBStartPos, SEG, false, DoVitR);
			else if (H.DBStartPos >= SEG) F.Scan(H.DBStartPos - SEG, SEG + 2 * QLx, true, DoVitR);
		}
	}

	// search2.cpp:8-57
	void AdjustTopHitsAndMapqs() {
		if (PairScore.empty()) { F.Mapq /= 2; R.Mapq /= 2; return; }
		double Fract = double(BestPairScore) / double(F.QL + R.QL);
		double Drop = BestPairScore - SecondBestPairScore;
		if (Drop > 30) Drop = 30;
		unsigned mapq = (unsigned)(Drop * Fract * Fract);
		if (mapq > 40) mapq = 40;
		if (mapq > F.Mapq) F.Mapq = mapq;
		if (mapq > R.Mapq) R.Mapq = mapq;
		if (BestPairIndex != UINT_MAX) { F.TopHit = int(PairF[BestPairIndex]); R.TopHit = int(PairR[BestPairIndex]); }
		if (SecondPairIndex != UINT_MAX) { SecondF = int(PairF[SecondPairIndex]); SecondR = int(PairR[SecondPairIndex]); }
	}

	// search2m4.cpp:15-187
	// search2m4.cpp:15-187; veryfast = Search5 (search2m5.cpp:9-127): the same seed loop, then no 90 % shortcut, no
	// FindPairs / ScanPair / AdjustTopHitsAndMapqs -- each mate keeps its own top hit and CalcMAPQ6
	void Search4(const byte *Seqf, unsigned Lf, const byte *Seqr, unsigned Lr, bool veryfast = false) {
		F.InitPE(Seqf, Lf);
		R.InitPE(Seqr, Lr);
		BestPairScore = 0; SecondBestPairScore = 0; BestPairIndex = 0; SecondPairIndex = UINT_MAX;
		SecondF = -1; SecondR = -1;  // InitPE, state1.cpp:119
		PairF.clear(); PairR.clear(); PairScore.clear();
		if (Lf < F.X->W || Lr < F.X->W) { F.Mapq = 0; R.Mapq = 0; return; }  // outside the reference's domain
		const unsigned QL2 = (Lf + Lr) / 2;
		TermPairScorePhase1 = int(Lf) + int(Lr) + 5 * F.P.mismatch_score;
		std::vector<uint32_t> Qf, Qr, Df, Dr;
		std::vector<char> Pf, Pr;
		uint32_t QPosf = 0, QPosr = 0, DBPosf = 0, DBPosr = 0;
		bool Plusf = false, Plusr = false;
		unsigned kf = F.GetFirstBoth1Seed(QPosf, Plusf, DBPosf);
		unsigned kr = R.GetFirstBoth1Seed(QPosr, Plusr, DBPosr);
		do {
			if (kf != UINT_MAX) {
				Qf.push_back(QPosf); Pf.push_back(Plusf); Df.push_back(DBPosf);
				for (size_t i = 0; i < Qr.size(); ++i) {
					int64_t TL = std::llabs(int64_t(DBPosf) - int64_t(Dr[i])) + int64_t(QL2);
					if (TL <= MAX_TL && ExtendBoth1Pair4(QPosf, DBPosf, Plusf, Qr[i], Dr[i])) return;
				}
			}
			if (kr != UINT_MAX) {
				Qr.push_back(QPosr); Pr.push_back(Plusr); Dr.push_back(DBPosr);
				for (size_t i = 0; i < Qf.size(); ++i) {
					int64_t TL = std::llabs(int64_t(Df[i]) - int64_t(DBPosr)) + int64_t(QL2);
					if (TL <= MAX_TL && ExtendBoth1Pair4(Qf[i], Df[i], !Plusr, QPosr, DBPosr)) return;
				}
			}
			if (kf != UINT_MAX) kf = F.GetNextBoth1Seed(kf, QPosf, Plusf, DBPosf);
			if (kr != UINT_MAX) kr = R.GetNextBoth1Seed(kr, QPosr, Plusr, DBPosr);
		} while (kf != UINT_MAX || kr != UINT_MAX);
		for (size_t i = 0; i < Qf.size(); ++i) F.ExtendPen(Qf[i], Df[i], Pf[i] != 0);
		for (size_t i = 0; i < Qr.size(); ++i) R.ExtendPen(Qr[i], Dr[i], Pr[i] != 0);
		if (veryfast) { F.SearchPE_Pending(); R.SearchPE_Pending(); return; }
		if (F.BestScore >= int((Lf * 9) / 10) && R.BestScore >= int((Lr * 9) / 10)) {
			int64_t TL = std::llabs(int64_t(F.Hits[F.TopHit].DBStartPos) - int64_t(R.Hits[R.TopHit].DBStartPos)) + int64_t(QL2);
			if (TL <= MAX_TL) { F.Mapq = 40; R.Mapq = 40; return; }
		}
		F.SearchPE_Pending();
		R.SearchPE_Pending();
		FindPairs();
		if (PairScore.empty()) { ScanPair(); FindPairs(); }
		AdjustTopHitsAndMapqs();
	}
};

struct MateOut {
	bool Mapped; uint32_t SeqIndex, Coord; bool Plus; unsigned Mapq; std::string Path; bool HasHit;
};

static void MateMapped(Searcher &S, MateOut &M) {  // SetMappedPos, state1.cpp:129-145
	M.Mapped = false; M.SeqIndex = UINT32_MAX; M.Coord = UINT32_MAX; M.Plus = false; M.Mapq = S.Mapq; M.HasHit = false;
	M.Path.clear();
	if (S.TopHit < 0) return;
	const Hit &H = S.Hits[S.TopHit];
	unsigned TargetL = 0;
	uint32_t si = UINT32_MAX;
	uint32_t Coord = S.PosToCoordL(H.DBStartPos, si, TargetL);
	if (uint32_t(Coord + S.QL) > TargetL) return;  // un-mapped: m_TopHit = 0
	M.Mapped = true; M.SeqIndex = si; M.Coord = Coord; M.Plus = H.Plus; M.Path = H.Path; M.HasHit = true;
}
}  // namespace

// In-memory paired-end batch: reads 2i, 2i+1 are the mates of pair i.  results[2*npairs] carry, per mate, the hit
// State2 settled on after SetMappedPos (score = that hit's score, second = m_SecondBestScore, mapq = m_Mapq).
extern "C" int uo_map_pe_opts(const uo_index *X, const uo_params *P, const uint8_t *bases, const uint64_t *offs,
                              uint32_t npairs, int threads, int veryfast, uo_result *results, char **path_arena,
                              uo_counters *counters);
extern "C" int uo_map_pe_info(const uo_index *X, const uo_params *P, const uint8_t *bases, const uint64_t *offs,
                              uint32_t npairs, int threads, int veryfast, uo_result *results, char **path_arena,
                              uo_counters *counters, uo_pair_info *info);
extern "C" int uo_map_pe(const uo_index *X, const uo_params *P, const uint8_t *bases, const uint64_t *offs, uint32_t npairs,
                         int threads, uo_result *results, char **path_arena, uo_counters *counters) {
	return uo_map_pe_opts(X, P, bases, offs, npairs, threads, 0, results, path_arena, counters);
}

// veryfast: State2 method 5 (map2.cpp:17-21 sets the band radius to 4; Search5)
extern "C" int uo_map_pe_opts(const uo_index *X, const uo_params *Pin, const uint8_t *bases, const uint64_t *offs,
                              uint32_t npairs, int threads, int veryfast, uo_result *results, char **path_arena,
                              uo_counters *counters) {
	return uo_map_pe_info(X, Pin, bases, offs, npairs, threads, veryfast, results, path_arena, counters, nullptr);
}

// as uo_map_pe_opts; info[npairs] (may be NULL) also receives what State2::OutputTab2 reads beyond the results: each
// mate's m_TopHit as the pair stage left it (before SetMappedPos) and m_SecondHit (search2.cpp:49-56)
extern "C" int uo_map_pe_info(const uo_index *X, const uo_params *Pin, const uint8_t *bases, const uint64_t *offs,
                              uint32_t npairs, int threads, int veryfast, uo_result *results, char **path_arena,
                              uo_counters *counters, uo_pair_info *info) {
	uo_params Pv = *Pin;
	if (veryfast) Pv.band_radius = 4;
	const uo_params *P = &Pv;
	if (threads < 1) threads = 1;
	std::vector<std::string> Paths((size_t)2 * npairs);
	if (counters) memset(counters, 0, sizeof *counters);
#pragma omp parallel num_threads(threads)
	{
		PairSearcher S(X, *P);
#pragma omp for schedule(dynamic, 64)
		for (int64_t i = 0; i < (int64_t)npairs; ++i) {
			const uint64_t o0 = offs[2 * i], o1 = offs[2 * i + 1], o2 = offs[2 * i + 2];
			S.Search4(bases + o0, unsigned(o1 - o0), bases + o1, unsigned(o2 - o1), veryfast != 0);
			Searcher *M[2] = {&S.F, &S.R};
			for (int a = 0; a < 2; ++a) { M[a]->C.n_dpcells += M[a]->dp.Cells; M[a]->dp.Cells = 0; }
			if (info) {
				uo_pair_info &pi = info[i];
				const int second[2] = {S.SecondF, S.SecondR};
				for (int a = 0; a < 2; ++a) {
					pi.top_db[a] = pi.second_db[a] = UINT32_MAX;
					pi.top_score[a] = pi.second_score[a] = 0;
					pi.top_plus[a] = pi.second_plus[a] = 0;
					if (M[a]->TopHit >= 0) {
						const Hit &h = M[a]->Hits[M[a]->TopHit];
						pi.top_db[a] = h.DBStartPos; pi.top_score[a] = (int16_t)h.Score; pi.top_plus[a] = h.Plus;
					}
					if (second[a] >= 0) {
						const Hit &h = M[a]->Hits[second[a]];
						pi.second_db[a] = h.DBStartPos; pi.second_score[a] = (int16_t)h.Score; pi.second_plus[a] = h.Plus;
					}
				}
			}
			for (int a = 0; a < 2; ++a) {
				MateOut mo;
				MateMapped(*M[a], mo);
				uo_result &R = results[2 * i + a];
				memset(&R, 0, sizeof R);
				R.dbpos = UINT32_MAX; R.seq_index = UINT32_MAX; R.coord = UINT32_MAX;
				R.second = M[a]->SecondBestScore; R.mapq = M[a]->Mapq; R.hit_count = M[a]->HitCount; R.hsp_count = M[a]->HSPCount;
				if (M[a]->TopHit >= 0) R.score = M[a]->Hits[M[a]->TopHit].Score;
				if (mo.Mapped) {
					R.dbpos = M[a]->Hits[M[a]->TopHit].DBStartPos; R.seq_index = mo.SeqIndex; R.coord = mo.Coord; R.plus = mo.Plus;
					Paths[2 * i + a] = mo.Path;
					R.path_len = (uint16_t)mo.Path.size();
				}
			}
		}
#pragma omp critical
		{ AddCounters(counters, S.F.C); AddCounters(counters, S.R.C); }
	}
	size_t total = 1;
	for (auto &p : Paths) total += p.size() + 1;
	char *arena = (char *)malloc(total);
	if (!arena) return -1;
	size_t off = 0;
	for (size_t i = 0; i < Paths.size(); ++i) {
		results[i].path_off = (uint32_t)off;
		memcpy(arena + off, Paths[i].c_str(), Paths[i].size() + 1);
		off += Paths[i].size() + 1;
	}
	if (path_arena) *path_arena = arena;
	else free(arena);
	return 0;
}

static uint32_t PairedFlags(bool First, bool RevComp, bool MateRevComp, bool MateUnmapped) {  // output2.cpp:18-36
	uint32_t f = First ? 0x41 : 0x81;
	if (RevComp) f |= 0x10;
	if (MateUnmapped) f |= 0x08;
	else if (MateRevComp) f |= 0x20;
	return f;
}

namespace {
// UFIndex::PosToCoord (ufindex.cpp:701-727): UINT32_MAX, label untouched, when the position is in inter-sequence padding
static uint32_t PosToCoordTab(const uo_index *X, uint32_t Pos, std::string &Label) {
	const unsigned n = (unsigned)X->Labels.size();
	unsigned Lo = 0, Hi = n - 1;
	while (Lo <= Hi && Hi != UINT_MAX) {
		unsigned k = (Lo + Hi) / 2;
		uint32_t Off = X->Offsets[k], SL = X->SeqLengths[k];
		if (Pos >= Off && Pos < Off + SL) { Label = X->Labels[k]; return Pos - Off; }
		if (Pos > Off) Lo = k + 1;
		else Hi = k - 1;
	}
	return UINT32_MAX;
}
static std::string PairPosStr1(const uo_index *X, const Hit &H, bool Fwd) {  // outputtab2.cpp:29-42
	std::string L;
	uint32_t c = PosToCoordTab(X, H.DBStartPos, L);
	char b[64];
	snprintf(b, sizeof b, ":%u(%c)/%c", c + 1, H.Plus ? '+' : '-', Fwd ? '1' : '2');
	return L + b;
}
static std::string PairPosStr(const uo_index *X, const Hit *H1, const Hit *H2) {  // outputtab2.cpp:44-83
	if (!H1 && !H2) return "*";
	if (H1 && !H2) return PairPosStr1(X, *H1, true);
	if (!H1 && H2) return PairPosStr1(X, *H2, false);
	std::string L1, L2;
	uint32_t c1 = PosToCoordTab(X, H1->DBStartPos, L1), c2 = PosToCoordTab(X, H2->DBStartPos, L2);
	if (L1 == L2 && H1->Plus != H2->Plus) {
		char b[64];
		snprintf(b, sizeof b, ":%u-%u", c1 + 1, c2 + 1);
		return L1 + b;
	}
	return PairPosStr1(X, *H1, true) + "," + PairPosStr1(X, *H2, false);
}
static unsigned TemplateLength(const PairSearcher &S, const Hit &H1, const Hit &H2) {  // output2.cpp:49-69
	int t;
	if (H1.DBStartPos <= H2.DBStartPos) t = int(H2.DBStartPos + S.R.QL) - int(H1.DBStartPos);
	else t = int(H1.DBStartPos + S.F.QL) - int(H2.DBStartPos);
	if (t < 0 || t > 1000) t = 0;
	return unsigned(t);
}
// State2::OutputTab2 (outputtab2.cpp:85-120).  top1/top2: m_TopHit of the mates as they stand when the line is
// written -- after SetSAM2's SetMappedPos when SAM output is on (output2.cpp:12-13,73-74), which clears a top hit that
// overhangs its sequence.
static std::string TabLine(const uo_index *X, const PairSearcher &S, const char *Label1, const Hit *top1, const Hit *top2) {
	std::string out;
	size_t n = strlen(Label1);  // GetPairLabel, state1.cpp:762-778
	if (n > 2 && Label1[n - 2] == '/' && (Label1[n - 1] == '1' || Label1[n - 1] == '2')) n -= 2;
	for (size_t i = 0; i < n && !isspace((unsigned char)Label1[i]); ++i) out.push_back(Label1[i]);
	const Hit *s1 = S.SecondF >= 0 ? &S.F.Hits[S.SecondF] : nullptr, *s2 = S.SecondR >= 0 ? &S.R.Hits[S.SecondR] : nullptr;
	out += "\t" + PairPosStr(X, top1, top2);
	char b[96];
	snprintf(b, sizeof b, "\t%u,%u\t", S.F.Mapq, S.R.Mapq);
	out += b;
	out += s1 ? PairPosStr(X, s1, s2) : std::string("*");
	if (top1 && top2 && s1 && s2) {  // GetInfoStr, outputtab2.cpp:6-27 (Psasc appends ';' after every item)
		unsigned tl1 = TemplateLength(S, *top1, *top2), tl2 = TemplateLength(S, *s1, *s2);
		if (tl1 == tl2) snprintf(b, sizeof b, "\tTL=%u;", tl1);
		else snprintf(b, sizeof b, "\tTL/%u,%u;", tl1, tl2);
		out += b;
		int sc1 = top1->Score + top2->Score, sc2 = s1->Score + s2->Score;
		if (sc1 == sc2) snprintf(b, sizeof b, "Score=%d;", sc1);
		else snprintf(b, sizeof b, "Score/%d,%d;", sc1, sc2);
		out += b;
	}
	out.push_back('\n');
	return out;
}
}  // namespace

extern "C" int uo_map_file_pe_tab(const uo_index *X, const uo_params *Pin, const char *fq1, const char *fq2, const char *sam,
                                  const char *tab, int threads, int veryfast, uo_counters *counters);
extern "C" int uo_map_file_pe(const uo_index *X, const uo_params *Pin, const char *fq1, const char *fq2, const char *sam,
                              int threads, int veryfast, uo_counters *counters) {
	return uo_map_file_pe_tab(X, Pin, fq1, fq2, sam, nullptr, threads, veryfast, counters);
}

// sam and/or tab may be NULL (no such output, as without -samout / -tabbedout)
extern "C" int uo_map_file_pe_tab(const uo_index *X, const uo_params *Pin, const char *fq1, const char *fq2, const char *sam,
                                  const char *tab, int threads, int veryfast, uo_counters *counters) {
	uo_params Pv = *Pin;
	if (veryfast) Pv.band_radius = 4;  // map2.cpp:17-21
	const uo_params *P = &Pv;
	std::vector<FastqRec> R1, R2;
	int rc = ReadFastq(fq1, R1);
	if (rc) return rc;
	rc = ReadFastq(fq2, R2);
	if (rc) return rc;
	if (R1.size() != R2.size()) return -4;
	const int64_t n = (int64_t)R1.size();
	std::vector<std::string> Out((size_t)n), OutTab(tab ? (size_t)n : 0);
	if (threads < 1) threads = 1;
	if (counters) memset(counters, 0, sizeof *counters);
#pragma omp parallel num_threads(threads)
	{
		PairSearcher S(X, *P);
		std::vector<char> buf;
#pragma omp for schedule(dynamic, 64)
		for (int64_t i = 0; i < n; ++i) {
			const FastqRec &A = R1[(size_t)i], &B = R2[(size_t)i];
			S.Search4((const byte *)A.Seq.data(), (unsigned)A.Seq.size(), (const byte *)B.Seq.data(), (unsigned)B.Seq.size(), veryfast != 0);
			// SetSAM2, output2.cpp:61-128
			MateOut M1, M2;
			MateMapped(S.F, M1);
			MateMapped(S.R, M2);
			if (tab) {
				// with SAM output on, SetMappedPos has already cleared overhanging top hits when the tab line is written
				const Hit *t1 = S.F.TopHit >= 0 && (!sam || M1.Mapped) ? &S.F.Hits[S.F.TopHit] : nullptr;
				const Hit *t2 = S.R.TopHit >= 0 && (!sam || M2.Mapped) ? &S.R.Hits[S.R.TopHit] : nullptr;
				OutTab[(size_t)i] = TabLine(X, S, A.Label.c_str(), t1, t2);
			}
			int TLEN1 = 0, TLEN2 = 0;
			const bool Plus1 = M1.HasHit && M1.Plus, Plus2 = M2.HasHit && M2.Plus;
			const bool StrandsConsistent = M1.HasHit && M2.HasHit && (Plus1 != Plus2);
			bool CorrectlyPaired = false;
			if (M1.Mapped && M2.Mapped) {
				// NB the reference compares coordinates inside the sequences, not global positions
				if (M1.Coord <= M2.Coord) {
					TLEN1 = int(M2.Coord + S.R.QL) - int(M1.Coord);
					if (TLEN1 > 0 && TLEN1 < 1000 && StrandsConsistent) CorrectlyPaired = true;
					if (TLEN1 > 1000) TLEN1 = 0;
					TLEN2 = -TLEN1;
				} else {
					TLEN2 = int(M1.Coord + S.F.QL) - int(M2.Coord);
					if (TLEN2 > 0 && TLEN2 < 1000 && StrandsConsistent) CorrectlyPaired = true;
					if (TLEN2 > 1000) TLEN2 = 0;
					TLEN1 = -TLEN2;
				}
			}
			const bool RevComp1 = M1.Mapped && !M1.Plus, RevComp2 = M2.Mapped && !M2.Plus;
			uint32_t Flags1 = PairedFlags(true, RevComp1, RevComp2, !M2.Mapped);
			uint32_t Flags2 = PairedFlags(false, RevComp2, RevComp1, !M1.Mapped);
			if (CorrectlyPaired) { Flags1 |= 0x02; Flags2 |= 0x02; }
			const char *L1 = M1.Mapped ? X->Labels[M1.SeqIndex].c_str() : "";
			const char *L2 = M2.Mapped ? X->Labels[M2.SeqIndex].c_str() : "";
			buf.resize(A.Label.size() + B.Label.size() + 3 * (A.Seq.size() + B.Seq.size()) + 1024);
			size_t k = SamRecord(X, M1.Mapped, M1.SeqIndex, M1.Coord, M1.Plus, M1.Mapq, M1.Path.c_str(), Flags1, L2, M2.Coord, TLEN1,
			                     A.Label.c_str(), (const byte *)A.Seq.data(), (const byte *)A.Qual.data(), (unsigned)A.Seq.size(), buf.data());
			k += SamRecord(X, M2.Mapped, M2.SeqIndex, M2.Coord, M2.Plus, M2.Mapq, M2.Path.c_str(), Flags2, L1, M1.Coord, TLEN2,
			               B.Label.c_str(), (const byte *)B.Seq.data(), (const byte *)B.Qual.data(), (unsigned)B.Seq.size(), buf.data() + k);
			Out[(size_t)i].assign(buf.data(), k);
		}
#pragma omp critical
		{ AddCounters(counters, S.F.C); AddCounters(counters, S.R.C); }
	}
	if (sam) {
		FILE *f = fopen(sam, "wb");
		if (!f) return -3;
		WriteSQ(f, X);
		for (int64_t i = 0; i < n; ++i) fwrite(Out[(size_t)i].data(), 1, Out[(size_t)i].size(), f);
		fclose(f);
	}
	if (tab) {
		FILE *f = fopen(tab, "wb");
		if (!f) return -3;
		for (int64_t i = 0; i < n; ++i) fwrite(OutTab[(size_t)i].data(), 1, OutTab[(size_t)i].size(), f);
		fclose(f);
	}
	return 0;
}

