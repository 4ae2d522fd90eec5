/*
 * urmapx.h -- C ABI of the MI355X-native urmap mapping path (liburmapx.so).
 *
 * The reference (rcedgar/urmap) has no plugin / FFI layer: the per-read path is compiled
 * into its one executable.  The seam it uses internally is
 *     State1::SetMethod / State1::SetUFI / State1::Search / State1::Output1   (map.cpp:11-25)
 *     UFIndex::FromFile                                                        (ufindexio.cpp:51-115)
 * and this header is the batch form of exactly that seam: plain pointers and sizes, no
 * C++ or torch types, error codes instead of Die()/exit (myutils.cpp:915).
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * All functions return 0 on success or a negative URMAPX_E_* code.  Nothing in this
 * library has a CPU fallback: without a HIP device every compute entry point returns
 * URMAPX_E_NODEVICE.
 */
#ifndef URMAPX_H
#define URMAPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URMAPX_OK 0
#define URMAPX_E_IO (-1)          /* cannot open / short read */
#define URMAPX_E_FORMAT (-2)      /* bad .ufi magic or header */
#define URMAPX_E_NOMEM (-3)       /* host or device allocation failed */
#define URMAPX_E_NODEVICE (-4)    /* no usable HIP device / HIP runtime error */
#define URMAPX_E_ARG (-5)         /* invalid argument */
#define URMAPX_E_UNSUPPORTED (-6) /* input outside the device path's domain (see status bits) */

/* per-read status bits (urmapx_result.status); non-zero => that read's result is not valid */
/* Single-end: a read the fast kernels flag is mapped again by the general kernel (lists in global memory), so these
 * remain only beyond ITS limits: 65536 live hits, 65536..262144 HSPs, a batch's path arena full, a flank window clipped
 * by the end of the sequence store with a band beyond 1100 columns, a read shorter than the word length (the reference
 * underflows there) or longer than URMAPX_MAX_QL_SLOW (where the reference stops with an assertion).  Paired-end: none either since
 * round 4 (the general pair kernel, kernels_pe_slow.hip), except mates beyond the reference's own byte-sized pending positions. */
#define URMAPX_ST_HIT_OVERFLOW 0x01 /* more live hits than the device lists hold */
#define URMAPX_ST_HSP_OVERFLOW 0x02 /* more HSPs than the device lists hold */
#define URMAPX_ST_PATH_OVERFLOW 0x04 /* alignment path longer than the path storage */
#define URMAPX_ST_BAND_TOO_WIDE 0x08 /* DP problem larger than the wide-band scratch */
#define URMAPX_ST_BAD_LENGTH 0x10    /* read shorter than the word length or longer than the kernels take */

#define URMAPX_MAX_QL 1024    /* single-end reads the fast kernels take.  Paired-end mates: at most 320 bases AND at most 256 k-mer starts (QL - W + 1 <= 256: the reference keeps pending seed positions in a byte, state1.h:86-87), i.e. 279 bases at the default word length 24 */
#define URMAPX_MAX_QL_SLOW 30836 /* single-end reads the general kernel takes (lists in global memory): the reference's own limit -- its per-read scratch is a 1 MB bump allocator (state1.h:113,208-213) that a read of 30 837 bases overruns ("assert failed: m_AllocBuffOffset <= m_AllocBuffSize", measured with the reference binary) */
#define URMAPX_MAX_PATH_OPS 96

typedef struct urmapx_index urmapx_index;
typedef struct urmapx_ctx urmapx_ctx;

/* State1::SetMethod constants (state1.cpp:147-183). */
typedef struct urmapx_params {
	int32_t mismatch_score;
	int32_t gap_open_score;
	int32_t gap_ext_score;
	int32_t min_hsp_score_pct;
	int32_t term_hsp_score_pct_phase3;
	int32_t xdrop;
	int32_t max_penalty;
	int32_t xphase1, xphase3, xphase4;
	uint32_t band_radius;
} urmapx_params;

/* method 6 = urmap -map default, 7 = -veryfast (state1.cpp:152-179) */
int urmapx_params_for_method(unsigned method, urmapx_params *out);

/* One read's outcome of State1::Search (search1.cpp:7-24) incl. SetMappedPos (state1.cpp:129-145). */
typedef struct urmapx_result {
	uint32_t dbpos;     /* top hit start in the concatenated sequence; UINT32_MAX = unmapped */
	uint32_t seq_index; /* sequence directory index, UINT32_MAX if unmapped */
	uint32_t coord;     /* 0-based position in that sequence */
	int16_t score;      /* m_BestScore */
	int16_t second;     /* m_SecondBestScore */
	uint8_t mapq;       /* m_Mapq */
	uint8_t plus;       /* 1 = read maps as given, 0 = reverse complement */
	uint8_t exit_phase; /* 1..6: phase of Search_Lo (search1m6.cpp:35-277) that returned */
	uint8_t status;     /* URMAPX_ST_* bits */
	uint16_t hit_count; /* m_HitCount */
	uint16_t path_nops; /* 0 => ungapped ("<QL>M"); else run count in the path arena */
	uint32_t path_off;  /* first run of this read in the path arena */
} urmapx_result;

/* Path arena entry: (len << 2) | code; code 0 = M, 1 = D (query base vs gap), 2 = I (target base vs gap),
 * the reference's own path alphabet (pathinfo.h); CIGAR swaps D and I (cigar.cpp:22-25). */
typedef uint16_t urmapx_path_op;

/* ---- index: UFIndex::FromFile (ufindexio.cpp:51-115) + the device-load path ---- */
int urmapx_index_open(const char *ufi_path, urmapx_index **out);
/* The same file straight into the HBM of `device` (round 5): the header is parsed on the host, the slot table and the sequence store
 * stream from the file through page-locked buffers read by several threads, and the resident layouts are built there; no host copy
 * of the arrays is kept (urmapx_index_replicate copies device to device from such an index).  What `urmap -map / -map2` loads with. */
int urmapx_index_open_device(const char *ufi_path, int device, urmapx_index **out);
/* Wrap caller-owned HOST arrays (no copy; must outlive the index).  labels = seq_count NUL-terminated strings. */
int urmapx_index_wrap_host(uint32_t word_length, uint32_t max_ix, uint64_t slot_count, const uint8_t *blob,
                           const uint8_t *seqdata, uint32_t seqdata_size, uint32_t seq_count,
                           const uint32_t *seq_lengths, const uint32_t *offsets, const char *labels,
                           urmapx_index **out);
/* Adopt caller-owned DEVICE arrays on `device` (index already resident in HBM, e.g. built there).
 * d_blob must have >= 5*slot_count+8 bytes, d_seqdata >= seqdata_size+4096 bytes with the tail zeroed. */
int urmapx_index_wrap_device(int device, uint32_t word_length, uint32_t max_ix, uint64_t slot_count,
                             const void *d_blob, const void *d_seqdata, uint32_t seqdata_size, uint32_t seq_count,
                             const uint32_t *seq_lengths, const uint32_t *offsets, const char *labels,
                             urmapx_index **out);
/* Copy slot table + sequence to HBM of `device` (once per GPU); no-op if already resident there. */
int urmapx_index_upload(urmapx_index *, int device);
/* A second, third... replica of an index that has its host arrays (opened or wrapped) in the HBM of another device:
 * one replica per GPU, reads sharded across them, no exchange between devices (the reference's threads share one
 * UFIndex, map.cpp:43-61).  The new object borrows src's host arrays: close it before src. */
int urmapx_index_replicate(const urmapx_index *src, int device, urmapx_index **out);
void urmapx_index_close(urmapx_index *);

/* bytes of GetRow_Blob's rows laid out beside the resident slot table (chain_rows.hip); 0 if they were not built */
uint64_t urmapx_index_chain_row_bytes(const urmapx_index *);
/* UFIndex::Validate / cmd_ufi_validate (ufindex.cpp:611-658, ufistats.cpp:141-147) as one pass over the RESIDENT table (the
 * index must have been uploaded or wrapped on a device): every slot whose tally says "mine" heads a row; the row is collected
 * link by link as GetRow_Validate does (ufindex.cpp:834-881: head "mine", later links "other", the middle slot of a long link
 * TALLY_NEXT_LONG_OTHER, at most MaxIx entries), every position must lie inside the sequence store and the word that starts
 * there must hash back to the head slot (the reference dies with "WordToSlot != Slot").  Beyond the reference's test, every
 * slot that is not free must lie on exactly one head's chain (reached == used).  Returns URMAPX_OK for a valid table,
 * URMAPX_E_FORMAT if any check failed (the report says which), URMAPX_E_ARG if the index is not resident. */
typedef struct urmapx_validate_report {
	uint64_t slots;          /* slots of the table */
	uint64_t heads;          /* slots that head a row (tally "mine") */
	uint64_t positions;      /* stored positions re-hashed */
	uint64_t used;           /* slots that are not TALLY_FREE */
	uint64_t reached;        /* slots the walks passed through (chain links and the middle slots of long links) */
	uint64_t bad_hash;       /* positions whose word does not hash to the head slot */
	uint64_t bad_pos;        /* positions outside the sequence store */
	uint64_t bad_link;       /* mine / other bits, long-link middle slots or steps that break the chain rules */
	uint64_t bad_len;        /* rows longer than MaxIx */
	uint64_t first_bad_slot; /* lowest head slot with a failure; UINT64_MAX: none */
	double seconds;          /* the pass on the device */
} urmapx_validate_report;
int urmapx_index_validate(const urmapx_index *, urmapx_validate_report *out);
/* Which bytes is this?  Checksum of a device-resident array: the sum, modulo 2^64, over its little-endian 64-bit words w_i
 * (i = 0, 1, ...; the last word zero-padded) of murmur64(w_i + (i + 1) * 0x9E3779B97F4A7C15) with the reference's murmur64
 * (ufindex.h:50-58).  d_ptr must be 8-byte aligned.  A sum, so the same array gives the same value on every device and in the
 * numpy restatement of tests/oracle_lib.py.  urmapx_index_checksum: out[0] the resident slot table (5 * slot_count bytes, the
 * .ufi file's m_Blob), out[1] the sequence store (seqdata_size bytes, m_SeqData) -- bench.py prints both so that a run says which
 * genome and which table it mapped against (the reference has no counterpart; its -ufi_validate checks consistency, not identity). */
int urmapx_checksum_device(int device, const void *d_ptr, uint64_t nbytes, uint64_t *out);
int urmapx_index_checksum(const urmapx_index *, uint64_t out[2]);
/* the same over the layouts DERIVED from the table at upload: out[0] slot16 (16 * slot_count bytes), out[1] the chain rows (4 bytes per
 * position); 0 for a layout that was not built.  Two uploads of one table must agree, whichever way the layouts were built. */
int urmapx_index_layout_checksum(const urmapx_index *, uint64_t out[2]);
uint32_t urmapx_index_word_length(const urmapx_index *);
uint32_t urmapx_index_max_ix(const urmapx_index *);
uint64_t urmapx_index_slot_count(const urmapx_index *);
uint32_t urmapx_index_seqdata_size(const urmapx_index *);
uint32_t urmapx_index_seq_count(const urmapx_index *);
const char *urmapx_index_label(const urmapx_index *, uint32_t i);
uint32_t urmapx_index_seq_length(const urmapx_index *, uint32_t i);
uint32_t urmapx_index_seq_offset(const urmapx_index *, uint32_t i);

/* ---- mapping context: one per (index, device, host thread); replaces State1 + SetUFI ---- */
int urmapx_ctx_create(const urmapx_index *, int device, const urmapx_params *, urmapx_ctx **out);
void urmapx_ctx_destroy(urmapx_ctx *);

/* State1::Search over a batch of single-end reads held in HOST memory.
 * bases: concatenated ASCII reads (exactly as in the FASTQ); offs[n+1] byte offsets.
 * results[n] and path_ops[path_cap] are caller-owned host arrays; *path_used receives the
 * number of arena entries written.  Returns URMAPX_E_UNSUPPORTED (results still filled,
 * offending reads carry status bits) if any read fell outside the device path's domain. */
int urmapx_map_se(urmapx_ctx *, const uint8_t *bases, const uint64_t *offs, uint32_t n, urmapx_result *results,
                  urmapx_path_op *path_ops, size_t path_cap, size_t *path_used);

/* State2::Search (search2.cpp:59-73; -map2, method 4) over a batch of read PAIRS in host memory: reads 2i and 2i+1 of
 * (bases, offs) are the two mates of pair i (R1, R2).  results[2*npairs]: per mate the hit AdjustTopHitsAndMapqs
 * (search2.cpp:8-57) settled on, after SetMappedPos; `score` is that hit's score.  Flags, RNEXT/PNEXT and TLEN are
 * host-side text (urmapx_sam_pe).  Device-domain limits as urmapx_map_se, plus read length <= 279 (the reference
 * keeps pending seed positions in a byte, state1.h:86-87, and crashes on longer pairs). */
int urmapx_map_pe(urmapx_ctx *, const uint8_t *bases, const uint64_t *offs, uint32_t npairs, urmapx_result *results,
                  urmapx_path_op *path_ops, size_t path_cap, size_t *path_used);

/* Same with inputs and outputs already resident in HBM of the ctx's device (no PCIe in the call):
 * d_bases, d_offs (uint64[n+1]), d_results[n], d_path_ops[n*URMAPX_MAX_PATH_OPS], d_path_used (uint32).
 * Asynchronous on the ctx stream; urmapx_ctx_sync() waits. total_bases = offs[n]. */
int urmapx_map_se_device(urmapx_ctx *, const void *d_bases, const void *d_offs, uint32_t n, uint64_t total_bases,
                         uint32_t max_read_len, void *d_results, void *d_path_ops, void *d_path_used);
/* `urmap -map2 ... -veryfast`: State2 method 5 = Search5 (search2m5.cpp:9-156) with band radius 4 (map2.cpp:17-21).
 * The context's State1 parameters stay method 6, as in the reference (map2.cpp:15-16). */
int urmapx_ctx_set_pe_veryfast(urmapx_ctx *, int on);
/* Paired form of urmapx_map_se_device: 2*npairs reads resident in HBM, mates interleaved. */
int urmapx_map_pe_device(urmapx_ctx *, const void *d_bases, const void *d_offs, uint32_t npairs, uint64_t total_bases,
                         uint32_t max_read_len, void *d_results, void *d_path_ops, void *d_path_used);
int urmapx_ctx_sync(urmapx_ctx *);
/* Device time (ms, HIP events on the ctx stream) of the two kernels in the most recent *_device call that has
 * completed: [0] seed+probe, [1] search/extend. */
int urmapx_ctx_last_kernel_ms(urmapx_ctx *, float ms[2]);
/* A single-end call is: the search kernel (seed + probe + Search_Lo's phases 1-5), the flank-DP launches of phase 6, phase
 * 6's ordered part (finalize); the same three for the few reads whose hit / HSP lists outgrew the first pass's; the
 * general kernel over whatever both passes left flagged.  Device times in ms: [0] search, [1] its DP launches summed,
 * [2] its finalize launches summed, [3..5] the second pass likewise, [6] the general kernel. */
int urmapx_ctx_stage_ms(urmapx_ctx *, float ms[7]);
/* The same call's phase-6 launches one by one (first pass): ms[2*r] = the DP launch of round r, ms[2*r+1] = the finalize
 * launch behind it; *rounds = how many rounds the call had (3 or 4: urmapx_ctx_dp_rounds). */
int urmapx_ctx_round_ms(urmapx_ctx *, float ms[16], int *rounds);
/* The rounds of that call: round r = the HSPs lo[r] <= k < lo[r + 1] of a read (the last round is open ended: lo[rounds] = UINT32_MAX).
 * Reads of up to 192 bases: [0,2) [2,16) [16,..); longer reads: [0,2) [2,8) [8,32) [32,..) (round 5). */
int urmapx_ctx_dp_rounds(urmapx_ctx *, uint32_t lo[8], int *rounds);
/* Round 5: phase 3 of Search_Lo (AlignHSP when the best HSP of phases 1-2 is long, search1m6.cpp:162-171) is parked like phase 6:
 * the search stage ([0] above) is then three launches -- ms[0] the first search launch (seed + probe + phases 1-2 for every read,
 * phases 4-5 for the reads with nothing to align in phase 3), ms[1] phase 3's flank-DP launch, ms[2] the search launch over the reads
 * parked at phase 3 (replay of AlignHSP's bookkeeping, phases 4-5).  stats[0] = DpJobs made for phase 3, stats[1] = reads parked there.
 * All zero when phase 3 ran inside the search kernel (the context records which it was; round 6), which is the DEFAULT: measured at hg38 scale
 * the three launches take 6 % longer than the one (profiles/r5/README.md).  URMAPX_PARK_PHASE3=1 in the environment -- set BEFORE the index is
 * uploaded: the parked variant needs the row layout, which an upload without the knob drops once slot16 is built -- turns the parking on (reads
 * of up to 320 bases). */
int urmapx_ctx_phase3(urmapx_ctx *, float ms[3], uint32_t stats[2]);
/* Statistics of the same call, per pass (4 numbers each): HSPs handed to the DP launches, reads they belong to, how many
 * of those DPs the ordered replay of AlignHSP (alignhsp.cpp:60-172) looked at, and how many were dropped before their DP
 * because the penalty cap had fallen far enough by their round. */
int urmapx_ctx_dp_stats(urmapx_ctx *, uint32_t out[8]);

/* Diagnostic: shader cycles spent per phase by the last search kernel, summed over wavefronts:
 * [0] setup, [1] phases 1+2, [2] phase 3, [3] chain walks, [4] phase 4, [5] phase 5, [6] phase 6, [7] output;
 * inside the candidate batches of phases 1,2,4,5: [8] locate+fetch, [9] window compare, [10] x-drop walks, [11] ordered part.
 * Only collected when URMAPX_PHASE_STATS is set in the environment (otherwise all zero / E_ARG). */
int urmapx_ctx_phase_cycles(urmapx_ctx *, uint64_t out[12]);
/* Diagnostic, same switch: shader cycles / 16 spent on each of the first n reads of the last single-end call (the cost
 * of a read is heavy-tailed; this is how the tail is looked at). */
int urmapx_ctx_read_cycles(urmapx_ctx *, uint32_t *out, uint32_t n);

/* ---- stage-level entry points (same device code the batch call runs; used by parity tests and bench) ---- */
/* State1::SetSlotsVec (state1.cpp:396-438) + UFIndex::GetBlob (ufindex.h:184-187) for both strands of every read.
 * Host arrays; per read r, entries [2*offs[r], 2*offs[r]+L) are the plus strand by query position and
 * [2*offs[r]+L, 2*offs[r]+2L) the minus strand (positions > L-W unused).  slots: UINT64_MAX = no k-mer. */
int urmapx_seed_probe(urmapx_ctx *, const uint8_t *bases, const uint64_t *offs, uint32_t n, uint64_t *slots,
                      uint8_t *tallies, uint32_t *positions);
/* State1::Viterbi (viterbi.cpp:11-261) + TraceBackBitMem for a batch of (A,B) pairs.
 * a/b: concatenated sequences with offsets; flags bit0 = Left, bit1 = Right.  Outputs per problem: score,
 * status (URMAPX_ST_*), and the run-length path in ops[i*URMAPX_MAX_PATH_OPS ..] with nops[i] entries. */
int urmapx_viterbi_batch(urmapx_ctx *, const uint8_t *a, const uint32_t *a_offs, const uint8_t *b,
                         const uint32_t *b_offs, const uint8_t *flags, uint32_t n, float *scores, uint8_t *status,
                         urmapx_path_op *ops, uint16_t *nops);

/* The same stage over reads already resident in HBM of the ctx's device, nothing copied back (measurement: the probe launch alone,
 * *ms = its time on the ctx stream; inside urmapx_map_se* the probe is a stage of the search kernel).  Results stay in the context's
 * probe arrays. */
int urmapx_seed_probe_device(urmapx_ctx *, const void *d_bases, const void *d_offs, uint32_t n, uint64_t total_bases,
                             uint32_t max_read_len, float *ms);

/* Measurement aid (no reference counterpart): n_loads independent random 5-byte slot reads over the resident slot
 * table and nothing else -- the random-access ceiling of this table on this device, which bench.py reports next to
 * the probe kernel's rate (SURVEY.md section 8d asks for the denominator to be measured, not assumed). */
int urmapx_ctx_gather_microbench(urmapx_ctx *, uint64_t n_loads, double *loads_per_s);

/* ---- index construction (host side; the command line's -make_ufi) ---- */
/* cmd_make_ufi (ufindexio.cpp:117-179): FASTA -> .ufi, byte-identical to the reference's for the same slot count.
 * slots is mandatory here; the command line applies the reference's default, GetPrime(file_size / 0.6)
 * (prime.cpp:11-21), when -slots is absent. */
int urmapx_make_ufi(const char *fasta_path, const char *ufi_path, uint32_t word_length, uint32_t max_ix, uint64_t slots);
/* UFIndex::MakeIndex (ufindex.cpp:83-151) on an already concatenated upper-case sequence store; blob: 5*slots bytes. */
int urmapx_build_slots(const uint8_t *seqdata, uint32_t seqdata_size, uint32_t word_length, uint32_t max_ix,
                       uint64_t slots, uint8_t *blob, uint32_t *truncated_out);

/* The same two with the data-parallel passes on the GPU (both strands' per-slot counts, ufindex.cpp:338-408; the first
 * indexed position of every slot; the head slots; the remaining positions as a list in genome order) and only the
 * order-dependent inserts (UpdateSlot / FindFreeSlot, ufindex.cpp:194-322,987-1000) on the host.  Byte-identical output.
 * d_seqdata: the sequence store already resident on `device`, or NULL (then the host array seqdata is uploaded). */
int urmapx_make_ufi_gpu(int device, const char *fasta_path, const char *ufi_path, uint32_t word_length, uint32_t max_ix, uint64_t slots);
/* Either builder (device < 0: host) with cmd_make_ufi's label option: by default sequence labels are cut at the first
 * white space (ufindexio.cpp:123-128); URMAPX_UFI_KEEP_LABELS = the reference's -notrunclabels. */
#define URMAPX_UFI_KEEP_LABELS 1u
int urmapx_make_ufi_opts(int device, const char *fasta_path, const char *ufi_path, uint32_t word_length, uint32_t max_ix,
                         uint64_t slots, unsigned flags);
int urmapx_build_slots_gpu(int device, const uint8_t *seqdata, const void *d_seqdata, uint32_t seqdata_size, uint32_t word_length,
                           uint32_t max_ix, uint64_t slots, uint8_t *blob, uint32_t *truncated_out);

/* ---- host-side text (no device involved) ---- */
/* One SAM record of a single-end read: State1::SetSAM / SetSAM_Unmapped (setsam.cpp:12-207) with Flags = 0 as
 * State1::Output1 passes (output1.cpp:13), CIGAR per state1.cpp:707-734 + cigar.cpp.  path_ops = the batch arena.
 * Writes at most cap bytes (no NUL); returns the record length, or 0 if cap is too small. */
size_t urmapx_sam_se(const urmapx_index *, const urmapx_result *r, const urmapx_path_op *path_ops, const char *label,
                     const uint8_t *seq, const uint8_t *qual, uint32_t read_len, char *buf, size_t cap);
/* The two SAM records of a read pair: State2::SetSAM2 / GetPairedFlags (output2.cpp:18-128) + SetSAM. */
size_t urmapx_sam_pe(const urmapx_index *, const urmapx_result *r1, const urmapx_result *r2, const urmapx_path_op *path_ops,
                     const char *label1, const uint8_t *seq1, const uint8_t *qual1, uint32_t len1, const char *label2,
                     const uint8_t *seq2, const uint8_t *qual2, uint32_t len2, char *buf, size_t cap);
/* @SQ lines of State1::WriteSAMHeader (state1.cpp:736-748); same return convention. */
size_t urmapx_sam_header_sq(const urmapx_index *, char *buf, size_t cap);

/* ---- paired-end summary (-tabbedout) ---- */
/* What State2::OutputTab2 (outputtab2.cpp:85-120) reads beyond the two mates' results: each mate's m_TopHit as the
 * pair stage left it (BEFORE SetMappedPos, which SetSAM2 applies only when SAM output is on) and m_SecondHit, which
 * AdjustTopHitsAndMapqs sets from the second-best pair (search2.cpp:49-56).  UINT32_MAX = no such hit. */
typedef struct urmapx_pair_info {
	uint32_t top_db[2], second_db[2];
	int16_t top_score[2], second_score[2];
	uint8_t top_plus[2], second_plus[2];
} urmapx_pair_info;
/* on != 0: urmapx_map_pe / urmapx_map_pe_device also record one urmapx_pair_info per pair (device side) */
int urmapx_ctx_set_pair_info(urmapx_ctx *, int on);
/* copies the records of the last paired-end call (npairs of them) to the host */
int urmapx_ctx_get_pair_info(urmapx_ctx *, urmapx_pair_info *out, uint32_t npairs);
/* One line of the reference's -tabbedout file: pair label, top pair position(s), the two MAPQs, second pair or '*',
 * and "TL=..;Score=..;" when all four hits exist.  sam_on: the reference's line differs when -samout is also given
 * (a top hit overhanging its sequence has been cleared by then).  Returns the length written, 0 if cap is too small. */
size_t urmapx_tab_pe(const urmapx_index *, const urmapx_result *r1, const urmapx_result *r2, const urmapx_pair_info *info,
                     const char *label1, uint32_t len1, uint32_t len2, int sam_on, char *buf, size_t cap);

/* ---- file to file: cmd_map / cmd_map2 (map.cpp:27-67, map2.cpp:39-90) as a library call ---- */
typedef struct urmapx_map_options {
	int first_gpu;      /* -gpu D */
	int gpus;           /* -gpus N: devices D..D+N-1, one replica of the index on each; 0 = 1 */
	int streams;        /* -streams K: mapping contexts per device (copies of one overlap kernels of another); 0 = 2 */
	int host_threads;   /* -threads: FASTQ parsing and SAM formatting; 0 = min(16, hardware threads) */
	uint32_t batch;     /* reads per batch; 0 = the library chooses: 262 144, up to 524 288 for files large enough to give every lane four chunks */
	int veryfast;       /* -veryfast: State1 method 7 for -map, State2 method 5 for -map2 */
	unsigned minq;      /* -minq (only -map2 reads it, map2.cpp:76) */
	const char *cmdline; /* text after CL: in the @PG line (NULL: empty) */
	/* round 4 (0 = the reference's behaviour: one SAM file) */
	int sam_shards;     /* -samshards N: the SAM text goes to N files samout.0 .. samout.N-1, shard s = the records of the s-th
	                     * N-th of the input (cut at a record), written by its own pipeline (reader, lanes, writer) on its own
	                     * share of the devices; the header is in shard 0, so `cat samout.0 .. samout.N-1` is byte for byte what one
	                     * file would hold.  Plain (seekable, not .gz) input only: other input is mapped by ONE
	                     * pipeline over all the devices into shard 0, the other shards stay empty.  -tabbedout is split the same way
	                     * (tabout.0 ..).  Needs samout (URMAPX_E_ARG without); N in 1..64 must divide gpus or be a multiple of it.  A failed
	                     * run removes the shard files it wrote */
	int discard_sam;    /* measurement: the SAM text is made and copied to the host, then dropped instead of written (report.medium
	                     * "discarded"): what the device lanes sustain when the output medium is not in the way */
} urmapx_map_options;
typedef struct urmapx_map_report {  /* State1::HitStats' counters (state1.cpp:593-632) and where the time went */
	uint64_t reads, mapped_q, mapped_lowq, unmapped, unsupported;
	double seconds;                                  /* first read parsed .. last SAM byte written (index load excluded) */
	double parse_s, gpu_s, format_s, write_s;        /* busy seconds per stage (gpu: summed over the lanes) */
	int host_threads, lanes;
	int write_threads;                               /* threads sharing one piece's pwrite (1 on tmpfs, up to 4 on a disk file system) */
	int text_on_device;                              /* 1: FASTQ bytes went to the device and SAM bytes came back (plain or .gz files) */
	uint64_t input_bytes;                            /* uncompressed FASTQ bytes that took that road (for .gz: parse_s is the inflater's busy time) */
	char medium[24];                                 /* what the SAM file lives on: "tmpfs", "disk file system", "pipe", "none", "discarded" */
	/* round 4: where a device lane's time goes (seconds summed over the chunks of all lanes, from events on the lanes' streams;
	 * text phase only): FASTQ bytes to the device, line ends / record checks / base copy, the mapping kernels, SAM lengths + text,
	 * SAM bytes back */
	double dev_h2d_s, dev_parse_s, dev_map_s, dev_format_s, dev_d2h_s;
	int shards;                                      /* SAM files written (1 unless sam_shards) */
	char placement[256];                             /* where the run's threads were put: "gpu0@node1 gpu1@node1; reader+writer@node1" per pipeline
	                                                  * (shards separated by " | "); @any = not pinned (no NUMA node known for the device).  A lane's
	                                                  * host thread runs on the CPUs of its device's NUMA node; reader and writer too when all devices
	                                                  * of the pipeline share one node */
	double shard_scan_s;                             /* sam_shards: seconds spent cutting the input at record starts (pairs: counting the lines that
	                                                  * place the cuts of the mates' file); inside `seconds` */
	double dev_map_search_s, dev_map_dp_s;           /* of dev_map_s (single-end): the search launches; phase 6's dp + finalize launches */
	double map_enqueue_s;                            /* host seconds the lanes spent enqueueing mapping launches (a lane thread that is not scheduled shows here) */
	double alloc_dev_s, alloc_pinned_s;              /* seconds all threads of the call spent in hipMalloc / hipFree of the lanes' device arrays, and in
	                                                  * hipHostMalloc / hipHostFree of page-locked chunk buffers (kept for the next call: 0 calls when warm) */
	uint32_t alloc_dev_calls, alloc_pinned_calls;
} urmapx_map_report;
/* fastq2 NULL: single-end (-map); else the mates' file (-map2 ... -reverse).  samout / tabout may be NULL.  The index
 * needs its host arrays, or to be resident on first_gpu already (the other devices' replicas are then copied from there).  Batch b is mapped on device
 * b mod gpus; records are written in input order.  Returns URMAPX_E_UNSUPPORTED if reads fell outside the device
 * domain (report->unsupported of them), URMAPX_E_FORMAT with the reference's message in err for malformed FASTQ. */
int urmapx_map_files(urmapx_index *, const urmapx_map_options *, const char *fastq1, const char *fastq2, const char *samout,
                     const char *tabout, urmapx_map_report *report, char *err, size_t errcap);

/* What urmapx_map_files keeps for the next call of this process, and what lets go of it:
 *   - page-locked chunk buffers, up to 16 GiB of host memory (pinning a few hundred MB costs as much as mapping the chunk in it);
 *   - its lanes: up to 8 mapping contexts with their text stages, 2.5-3 GB of DEVICE memory each (DpJobs, parked states, path arena,
 *     the chunk's text both ways), keyed on index, device, parameters and the URMAPX_* environment -- a lane is kept only while an
 *     eighth of the device's memory is still free with it there, so a process that shares the device with another allocator (torch)
 *     sees at most that much held back.
 * urmapx_host_pool_trim() frees both; urmapx_index_close() destroys the lanes of its index; URMAPX_NO_LANE_POOL=1 keeps no lanes. */
void urmapx_host_pool_trim(void);

/* ---- FASTQ bytes in, SAM bytes out (both text stages of -map on the device) ---- */
/* One chunk of a FASTQ file, cut after a record's last '\n', goes to the device as it lies in the file; line ends,
 * record checks (FASTQSeqSource::GetNextLo, fastqseqsource.cpp:9-116), State1::Search and the SAM records
 * (State1::SetSAM / SetSAM_Unmapped with output1.cpp:13's arguments, setsam.cpp:12-207) run there and the records' text
 * comes back in input order.  A chunk the device parser does not take as it is ('\r', blank or missing lines, a
 * malformed record, a target label over 160 bytes) is handed back untouched with report.reason set: the caller runs it
 * through urmapx_fastq_* / urmapx_sam_se, which reproduce the reference's handling and messages. */
typedef struct urmapx_text urmapx_text;
#define URMAPX_TEXT_OK 0
#define URMAPX_TEXT_CR 1          /* a '\r' in the chunk */
#define URMAPX_TEXT_RAGGED 2      /* line count not a multiple of four, or no '\n' at the end of the chunk */
#define URMAPX_TEXT_BAD_RECORD 3  /* '@' missing, a byte that is not a letter, #bases != #quals, blank line */
#define URMAPX_TEXT_LONG_NAME 4   /* target label longer than the device formatter takes */
#define URMAPX_TEXT_SAM_CAP 5     /* sam_cap < report.sam_bytes: the chunk is mapped, its text waits for urmapx_text_fetch_sam */
#define URMAPX_TEXT_TOO_LARGE 6   /* chunk over 1 GiB, or its SAM text over 4 GiB */
#define URMAPX_TEXT_UNEQUAL 7     /* pairs: the two chunks do not hold the same number of records */
#define URMAPX_TEXT_INTERNAL 8    /* the device formatter's two passes disagreed on a record length (never expected): text not used */
#define URMAPX_TEXT_DEFERRED 9    /* urmapx_text_set_deferred: the chunk is mapped and counted, its text is on its way (urmapx_text_wait) */
typedef struct urmapx_text_report {
	uint32_t records;   /* reads of the chunk */
	uint32_t reason;    /* URMAPX_TEXT_*; non-zero: nothing was written */
	uint64_t sam_bytes; /* bytes of SAM text (written, or needed when reason is URMAPX_TEXT_SAM_CAP) */
	uint64_t mapped_q, mapped_lowq, unmapped, unsupported; /* State1::HitStats' counters (output1.cpp:20-30) against minq */
	float ms_h2d, ms_parse, ms_map, ms_format, ms_d2h;     /* the chunk on its stream, by events: copy in, parse, map, SAM text, copy out */
	float ms_map_search, ms_map_dp;                        /* of ms_map (single-end): the search launch; phase 6's dp + finalize launches */
	float ms_map_enqueue;                                  /* host time spent enqueueing the mapping launches (no wait inside: all of it is the calling thread) */
} urmapx_text_report;
/* One per mapping context; calls on it run on the context's stream (one thread at a time per context). */
int urmapx_text_create(urmapx_ctx *, urmapx_text **out);
void urmapx_text_destroy(urmapx_text *);
/* fastq[fastq_bytes] and sam[sam_cap] are host arrays (page-locked ones cross PCIe without a staging copy). */
int urmapx_text_map_se(urmapx_text *, const char *fastq, size_t fastq_bytes, unsigned minq, char *sam, size_t sam_cap,
                       urmapx_text_report *report);
/* Pairs (-map2): a chunk of each mate file holding the same number of records; record 2i and 2i+1 of the text are the
 * mates of pair i with SetSAM2's flags, RNEXT, PNEXT and TLEN (output2.cpp:18-128); report.records counts reads (2 per
 * pair).  -tabbedout lines: urmapx_text_fetch_pairs + urmapx_tab_pe. */
int urmapx_text_map_pe(urmapx_text *, const char *fastq1, size_t fastq1_bytes, const char *fastq2, size_t fastq2_bytes,
                       unsigned minq, char *sam, size_t sam_cap, urmapx_text_report *report);
/* The copy back as a stage of its own.  With deferred on, urmapx_text_map_se / _pe / _fetch_sam return once the chunk's SAM text
 * has been made on the device and its copy to `sam` has been ENQUEUED on a second stream (report.reason URMAPX_TEXT_DEFERRED;
 * records, sam_bytes, the counters and ms_h2d / ms_parse / ms_map* are final), so that the caller can hand the context its next
 * chunk while the text crosses PCIe; urmapx_text_wait then waits for the OLDEST such copy and returns that chunk's final report
 * (reason 0 or URMAPX_TEXT_INTERNAL, ms_format, ms_d2h) -- only then may `sam` be read.  At most two chunks are in flight per
 * context: map(i), map(i+1), wait(i), map(i+2), wait(i+1) ...; a third map call before a wait returns URMAPX_E_ARG.
 * Round 5: a lane of urmapx_map_files runs this way (the reference has no counterpart: its threads write records as they finish). */
int urmapx_text_set_deferred(urmapx_text *, int on);
int urmapx_text_wait(urmapx_text *, urmapx_text_report *report);
/* After URMAPX_TEXT_SAM_CAP: the text of the chunk just mapped into a buffer of at least report.sam_bytes (the search is
 * not run again).  URMAPX_E_ARG if no such chunk is waiting. */
int urmapx_text_fetch_sam(urmapx_text *, char *sam, size_t sam_cap, urmapx_text_report *report);
/* After urmapx_text_map_pe mapped a chunk (reason 0 or URMAPX_TEXT_SAM_CAP) on a context with urmapx_ctx_set_pair_info
 * on: what State2::OutputTab2 (outputtab2.cpp:85-120) needs for the chunk's -tabbedout lines, which the host formats with
 * urmapx_tab_pe -- the 2 * npairs results, the npairs pair records, the offset of the '\n' of every line of the FIRST
 * file's chunk (4 * npairs: pair i's label is the text between the '@' that starts line 4i and line_ends1[4i], its first
 * mate's length line_ends1[4i+1] - line_ends1[4i] - 1) and the second mates' lengths.  npairs must be report.records / 2. */
int urmapx_text_fetch_pairs(urmapx_text *, uint32_t npairs, urmapx_result *results, urmapx_pair_info *info, uint32_t *line_ends1,
                            uint32_t *lens2);

/* ---- FASTQ input (host) ---- */
/* Batch form of FASTQSeqSource::GetNextLo (fastqseqsource.cpp:9-116) over LineReader (linereader.cpp:14-113):
 * plain or .gz by suffix; '\r' dropped; a final unterminated line counts; blank lines only at end of file; the same
 * accept/reject rules and messages ('@' expected, letters only, #bases == #quals).  One reader per file. */
typedef struct urmapx_fastq urmapx_fastq;
int urmapx_fastq_open(const char *path, urmapx_fastq **out);
/* Reads up to max_reads records.  Returns the number read (0 at end of file) or URMAPX_E_FORMAT with the reference's
 * message in urmapx_fastq_error().  The arrays belong to the reader and stay valid until its next call:
 * bases/quals concatenated, offs[n+1]; labels (text after '@', NUL terminated) at label_data + label_offs[i]. */
int64_t urmapx_fastq_next(urmapx_fastq *, uint32_t max_reads, const uint8_t **bases, const uint8_t **quals,
                          const uint64_t **offs, const char **label_data, const uint64_t **label_offs);
const char *urmapx_fastq_error(const urmapx_fastq *);
void urmapx_fastq_close(urmapx_fastq *);

/* ---- .gz input (host) ---- */
/* The reference reads .gz files through zlib, one stream on one thread (linereader.cpp:54-113, gzipfileio.cpp).  urmapx_map_files
 * cuts a gzip stream into segments that several threads inflate side by side (urmap_amd/csrc/pgzip.h: each finds a deflate block
 * start behind its cut, decodes with the 32 KB in front of it unknown, the references into that window are filled in once the
 * segment in front is done; every member's CRC-32 and length are checked as zlib does).  This entry point runs that reader alone:
 * gz_path inflated to out_path, the bytes `gzip -dc` writes.  threads <= 0: all.  stats (optional): bytes written, bytes that
 * came by the parallel road, bytes that came through zlib (small files, one thread, input the parallel decoder hands over).
 * URMAPX_E_FORMAT: not gzip, truncated, or corrupt (what was decoded before the damage is in out_path, as with zlib). */
int urmapx_gunzip_file(const char *gz_path, const char *out_path, int threads, uint64_t stats[3]);
/* Which of the reader's vector paths this host runs: bit 0 = symbols to bytes 32 at a time (AVX2), bit 1 = CRC-32 by carry-less
 * multiplication (PCLMULQDQ; set only after the routine has reproduced zlib's crc32 on its self-test).  URMAPX_PGZIP_NO_SIMD=1: neither. */
int urmapx_pgzip_simd(void);

const char *urmapx_strerror(int code);
/* "gfx950" etc. of the ctx's device; NULL without a device */
const char *urmapx_device_arch(urmapx_ctx *);

#ifdef __cplusplus
}
#endif
#endif
