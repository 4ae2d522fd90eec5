// san_main.cpp -- driver of the sanitizer builds (make san): the host side of the drop-in path run from the command line, with the
// device side replaced by san/stub_device.cpp.  tests/test_sanitizers_cpu.py runs it on well-formed and on damaged input; a
// sanitizer report ends the process with exit code 99 (ASAN_OPTIONS / UBSAN_OPTIONS / TSAN_OPTIONS exitcode, set by the tests).
//   urmap_san map <fastq1> [-2 fastq2] -o out.sam [-tab out.tab] [-batch N] [-streams K] [-gpus N] [-shards N] [-threads T] [-null]
//   urmap_san gunzip <in.gz> <out> [threads]
//   urmap_san fastq <file> <batch>
//   urmap_san makeufi <fasta> <out.ufi> <slots>
// Exit code: 0 = the call succeeded, 1 = it refused its input with an error code (printed), 2 = usage.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../../include/urmapx.h"

extern "C" urmapx_index *urx_stub_index(uint32_t n, const uint32_t *lengths, const char *const *labels);

static int cmd_map(int argc, char **argv) {
	const char *fq1 = argv[0], *fq2 = nullptr, *sam = nullptr, *tab = nullptr;
	urmapx_map_options o;
	memset(&o, 0, sizeof o);
	o.gpus = 1; o.streams = 2; o.minq = 10; o.cmdline = "urmap_san";
	for (int i = 1; i < argc; ++i) {
		const std::string a = argv[i];
		auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
		if (a == "-2") fq2 = val();
		else if (a == "-o") sam = val();
		else if (a == "-tab") tab = val();
		else if (a == "-batch") o.batch = (uint32_t)atoi(val());
		else if (a == "-streams") o.streams = atoi(val());
		else if (a == "-gpus") o.gpus = atoi(val());
		else if (a == "-shards") o.sam_shards = atoi(val());
		else if (a == "-threads") o.host_threads = atoi(val());
		else if (a == "-null") o.discard_sam = 1;
		else return 2;
	}
	static const uint32_t lengths[3] = {1000000u, 250000u, 4000000000u};
	static const char *const labels[3] = {"chrA", "chrB a label with spaces", "chrBig"};
	urmapx_index *I = urx_stub_index(3, lengths, labels);
	urmapx_map_report rep;
	char err[512];
	const int rc = urmapx_map_files(I, &o, fq1, fq2, sam, tab, &rep, err, sizeof err);
	printf("rc=%d reads=%llu mapped_q=%llu mapped_lowq=%llu unmapped=%llu text_on_device=%d shards=%d lanes=%d err=%s\n", rc, (unsigned long long)rep.reads,
	       (unsigned long long)rep.mapped_q, (unsigned long long)rep.mapped_lowq, (unsigned long long)rep.unmapped, rep.text_on_device, rep.shards, rep.lanes, err);
	urmapx_index_close(I);
	urmapx_host_pool_trim();
	return rc == URMAPX_OK ? 0 : 1;
}

static int cmd_gunzip(int argc, char **argv) {
	if (argc < 2) return 2;
	uint64_t st[3] = {0, 0, 0};
	const int rc = urmapx_gunzip_file(argv[0], argv[1], argc > 2 ? atoi(argv[2]) : 0, st);
	printf("rc=%d bytes=%llu parallel=%llu zlib=%llu\n", rc, (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2]);
	return rc == URMAPX_OK ? 0 : 1;
}

static int cmd_fastq(int argc, char **argv) {
	if (argc < 2) return 2;
	urmapx_fastq *f = nullptr;
	int rc = urmapx_fastq_open(argv[0], &f);
	if (rc) { printf("rc=%d open\n", rc); return 1; }
	const uint32_t batch = (uint32_t)atoi(argv[1]);
	uint64_t n = 0, bases_total = 0, h = 1469598103934665603ull;
	for (;;) {
		const uint8_t *bases, *quals;
		const uint64_t *offs, *loffs;
		const char *ldata;
		const int64_t k = urmapx_fastq_next(f, batch ? batch : 1, &bases, &quals, &offs, &ldata, &loffs);
		if (k < 0) { printf("rc=%lld records=%llu err=%s\n", (long long)k, (unsigned long long)n, urmapx_fastq_error(f)); urmapx_fastq_close(f); return 1; }
		if (k == 0) break;
		for (int64_t i = 0; i < k; ++i) {
			for (uint64_t p = offs[i]; p < offs[i + 1]; ++p) h = ((h ^ bases[p]) * 1099511628211ull ^ quals[p]) * 1099511628211ull;
			for (const char *c = ldata + loffs[i]; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
		}
		n += (uint64_t)k;
		bases_total += offs[k];
	}
	printf("rc=0 records=%llu bases=%llu digest=%016llx\n", (unsigned long long)n, (unsigned long long)bases_total, (unsigned long long)h);
	urmapx_fastq_close(f);
	return 0;
}

static int cmd_makeufi(int argc, char **argv) {
	if (argc < 3) return 2;
	const int rc = urmapx_make_ufi(argv[0], argv[1], 24, 32, strtoull(argv[2], nullptr, 10));
	printf("rc=%d\n", rc);
	return rc == URMAPX_OK ? 0 : 1;
}

int main(int argc, char **argv) {
	if (argc < 3) return 2;
	const std::string cmd = argv[1];
	if (cmd == "map") return cmd_map(argc - 2, argv + 2);
	if (cmd == "gunzip") return cmd_gunzip(argc - 2, argv + 2);
	if (cmd == "fastq") return cmd_fastq(argc - 2, argv + 2);
	if (cmd == "makeufi") return cmd_makeufi(argc - 2, argv + 2);
	return 2;
}
