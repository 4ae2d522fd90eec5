// urmap_main.cpp -- command line of the MI355X build: the reference's `urmap -map` / `-make_ufi` surface
// (urmap_main.cpp:6-41, map.cpp:27-67, ufindexio.cpp:117-179) as a batch dispatcher over liburmapx.so.
//
//   urmap -map reads.fq[.gz] -ufi index.ufi -samout out.sam [-veryfast] [-threads N] [-gpu D] [-gpus N] [-streams K] [-batch N]
//   urmap -make_ufi genome.fa -output index.ufi [-slots N] [-wordlength W] [-maxix M] [-veryfast] [-gpu D | -host]
//
//   urmap -map2 R1.fq -reverse R2.fq -ufi index.ufi -samout out.sam [-tabbedout out.tab]   (paired-end, map2.cpp:39-90)
//   urmap -ufi_validate index.ufi [-gpu D]                                                   (ufistats.cpp:141-147, on the device)
//
// Pipeline of -map: one reader thread parses FASTQ into batches; batch b goes to mapping lane b mod (N*K), a host
// thread with its own mapping context on GPU D + (b mod N) (-gpus N devices, each holding its own replica of the index,
// -streams K contexts per device so that one lane's copies overlap another's kernels); a writer thread takes the
// batches back in input order and formats and writes their SAM.  The reference fans reads over its OpenMP threads the
// same way (map.cpp:58-61, seqsource.cpp:30-66) but writes in completion order; here records are written
// in input order.  No data moves between devices.  Errors: message on stderr, exit status 1
// (myutils.cpp:915), as the reference.
#include <fcntl.h>
#include <omp.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/urmapx.h"
#include "sam.h"

using namespace urx;

[[noreturn]] static void die(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	fprintf(stderr, "\n---Fatal error---\n");
	vfprintf(stderr, fmt, ap);
	fprintf(stderr, "\n");
	va_end(ap);
	exit(1);
}

struct Opts {
	std::string map, map2, reverse, make_ufi, ufi, ufi_validate, samout, tabbedout, output, log;
	bool veryfast = false, quiet = false, minq_given = false, host_build = false, notrunclabels = false;
	double load_factor = 0.6;  // myopts.h: FLT_OPT(load_factor, 0.6, ...)
	unsigned threads = 0, wordlength = 24, maxix = 0, minq = 10;
	unsigned long long slots = 0;
	int gpu = 0, gpus = 1, streams = 2, samshards = 0;
	unsigned batch = 1u << 18;
};

static Opts parse(int argc, char **argv) {
	Opts o;
	for (int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		while (a.size() > 1 && a[0] == '-' && a[1] == '-') a.erase(0, 1);
		auto val = [&]() -> const char * {
			if (i + 1 >= argc) die("Missing value for option %s", a.c_str());
			return argv[++i];
		};
		if (a == "-map") o.map = val();
		else if (a == "-map2") o.map2 = val();
		else if (a == "-reverse") o.reverse = val();
		else if (a == "-make_ufi") o.make_ufi = val();
		else if (a == "-ufi") o.ufi = val();
		else if (a == "-ufi_validate") o.ufi_validate = val();
		else if (a == "-samout") o.samout = val();
		else if (a == "-tabbedout") o.tabbedout = val();
		else if (a == "-output") o.output = val();
		else if (a == "-threads") o.threads = (unsigned)atoi(val());
		else if (a == "-wordlength") o.wordlength = (unsigned)atoi(val());
		else if (a == "-maxix") o.maxix = (unsigned)atoi(val());
		else if (a == "-slots") o.slots = strtoull(val(), nullptr, 10);
		else if (a == "-minq") { o.minq = (unsigned)atoi(val()); o.minq_given = true; }
		else if (a == "-gpu") o.gpu = atoi(val());
		else if (a == "-gpus") o.gpus = atoi(val());
		else if (a == "-streams") o.streams = atoi(val());
		else if (a == "-samshards") o.samshards = atoi(val());
		else if (a == "-batch") o.batch = (unsigned)atoi(val());
		else if (a == "-veryfast") o.veryfast = true;
		else if (a == "-host") o.host_build = true;
		else if (a == "-quiet") o.quiet = true;
		else if (a == "-log") o.log = val();
		else if (a == "-load_factor") o.load_factor = atof(val());
		else if (a == "-trunclabels") {}  // -map: SetSAM cuts the read label at the first blank whatever this says (setsam.cpp); -make_ufi: the default
		else if (a == "-notrunclabels") o.notrunclabels = true;
		else die("Unknown option %s", a.c_str());
	}
	return o;
}

// -log FILE (myutils.cpp: the log file every command opens): program line, command line, start time; whatever the
// command Log()s; finish time.  HitStats writes "@rps=" and the report to it (state1.cpp:593-632).
static FILE *g_log = nullptr;
static std::chrono::steady_clock::time_point g_t0;
static void log_open(const Opts &o, int argc, char **argv) {
	if (o.log.empty()) return;
	g_log = fopen(o.log.c_str(), "w");
	if (!g_log) die("Cannot open log file '%s'", o.log.c_str());
	g_t0 = std::chrono::steady_clock::now();
	fprintf(g_log, "urmap (MI355X build)\n");
	for (int i = 0; i < argc; ++i) fprintf(g_log, "%s ", argv[i]);
	const time_t t = time(nullptr);
	fprintf(g_log, "\nStarted %s", asctime(localtime(&t)));
}
static void log_close() {
	if (!g_log) return;
	const time_t t = time(nullptr);
	const unsigned secs = (unsigned)std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
	fprintf(g_log, "\nFinished %s", asctime(localtime(&t)));
	fprintf(g_log, "Elapsed time %02u:%02u\n", secs / 60, secs % 60);
	fclose(g_log);
	g_log = nullptr;
}
// ProgressLog (myutils.cpp): the same text to the terminal and to the log file
static void progress_log(bool quiet, const char *fmt, ...) {
	va_list ap;
	if (!quiet) { va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
	if (g_log) { va_start(ap, fmt); vfprintf(g_log, fmt, ap); va_end(ap); }
}

static void check(int rc, const char *what) {
	if (rc != URMAPX_OK && rc != URMAPX_E_UNSUPPORTED) die("%s: %s", what, urmapx_strerror(rc));
}

static int cmd_map(const Opts &o, int argc, char **argv) {
	const bool paired = !o.map2.empty();
	const unsigned minq = paired ? o.minq : 10;  // only cmd_map2 reads -minq (map2.cpp:76); -map keeps State1::m_Minq = 10
	if (paired && o.reverse.empty()) die("-reverse required");
	if (o.ufi.empty()) die("-ufi option required");
	if (o.gpus < 1 || o.gpus > 64) die("-gpus must be 1..64");
	if (o.streams < 1 || o.streams > 8) die("-streams must be 1..8");
	if (o.samshards < 0 || o.samshards > 64) die("-samshards must be 0..64");
	const auto t0 = std::chrono::steady_clock::now();
	urmapx_index *I = nullptr;
	// the file streams to the first device (urmapx_index_open_device); URMAPX_HOST_INDEX=1: through host arrays as before (measurement)
	if (getenv("URMAPX_HOST_INDEX")) check(urmapx_index_open(o.ufi.c_str(), &I), ("Reading index " + o.ufi).c_str());
	else {
		const int rc = urmapx_index_open_device(o.ufi.c_str(), getenv("URMAPX_FORCE_DEVICE") ? atoi(getenv("URMAPX_FORCE_DEVICE")) : o.gpu, &I);
		if (rc == URMAPX_E_NODEVICE || rc == URMAPX_E_NOMEM) die("Uploading index to the GPU: %s", urmapx_strerror(rc));  // (the message of the upload step, as before)
		check(rc, ("Reading index " + o.ufi).c_str());
	}
	if (o.veryfast && urmapx_index_max_ix(I) > 3) fprintf(stderr, "\nWARNING: index not optimal for -veryfast\n");
	const double load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::string cl;
	for (int i = 0; i < argc; ++i) { cl += argv[i]; cl.push_back(' '); }  // argv joined with trailing spaces (state1.cpp:749-751)
	urmapx_map_options mo;
	memset(&mo, 0, sizeof mo);
	mo.sam_shards = o.samshards;  // -samout out.sam -samshards N: out.sam.0 .. out.sam.N-1, `cat` of them = the one file
	mo.first_gpu = o.gpu; mo.gpus = o.gpus; mo.streams = o.streams; mo.host_threads = (int)o.threads; mo.batch = o.batch;
	mo.veryfast = o.veryfast ? 1 : 0; mo.minq = o.minq; mo.cmdline = cl.c_str();
	urmapx_map_report rep;
	memset(&rep, 0, sizeof rep);
	char err[1024];
	const int rc = urmapx_map_files(I, &mo, paired ? o.map2.c_str() : o.map.c_str(), paired ? o.reverse.c_str() : nullptr,
	                                o.samout.empty() ? nullptr : o.samout.c_str(), o.tabbedout.empty() ? nullptr : o.tabbedout.c_str(),
	                                &rep, err, sizeof err);
	if (rc != URMAPX_OK && rc != URMAPX_E_UNSUPPORTED) die("%s", err[0] ? err : urmapx_strerror(rc));
	const unsigned long long n_reads = rep.reads, n_accept = rep.mapped_q, n_reject = rep.mapped_lowq, n_nohit = rep.unmapped;
	const double map_s = rep.seconds;
	if (getenv("URMAPX_VERBOSE"))
		fprintf(stderr, "stage busy seconds: parse %.2f, gpu (copies + kernels, summed over %d lanes) %.2f, format %.2f, write %.2f; %d host threads; text on device: %d\n",
		        rep.parse_s, rep.lanes, rep.gpu_s, rep.format_s, rep.write_s, rep.host_threads, (int)rep.text_on_device);
	if (getenv("URMAPX_VERBOSE") || g_log) {  // where the threads ran (NUMA node of each device's PCI function; @any: not pinned)
		if (getenv("URMAPX_VERBOSE")) fprintf(stderr, "placement: %s\n", rep.placement);
		if (g_log) fprintf(g_log, "placement: %s\n", rep.placement);
	}
	if (g_log) fprintf(g_log, "@rps=%.1f\n", map_s > 0 ? (double)n_reads / map_s : 0.0);  // Log("@rps=..."), state1.cpp:607
	if (!o.quiet || g_log) {  // State1::HitStats (state1.cpp:593-632): same lines, sub-second timers, "GPU n" where it says "n threads"; ProgressLog = terminal + log file
		const bool q = o.quiet;
		auto pct = [&](unsigned long long x) { return n_reads ? 100.0 * (double)x / (double)n_reads : 0.0; };
		auto commas = [](unsigned long long x) {  // IntToStrCommas (myutils.cpp:1400-1418)
			std::string d = std::to_string(x), r;
			for (size_t i = 0; i < d.size(); ++i) {
				if (i && (d.size() - i) % 3 == 0) r += ',';
				r += d[i];
			}
			return r;
		};
		auto short_int = [](unsigned long long x) {  // IntToStr (myutils.cpp:1420-1438): the unit steps
			char b[64];
			const double d = (double)x;
			if (x < 10000) snprintf(b, sizeof b, "%u", (unsigned)x);
			else if (d < 1e6) snprintf(b, sizeof b, "%.1fk", d / 1e3);
			else if (d < 100e6) snprintf(b, sizeof b, "%.1fM", d / 1e6);
			else if (d < 1e9) snprintf(b, sizeof b, "%.0fM", d / 1e6);
			else if (d < 10e9) snprintf(b, sizeof b, "%.1fG", d / 1e9);
			else if (d < 100e9) snprintf(b, sizeof b, "%.0fG", d / 1e9);
			else snprintf(b, sizeof b, "%.3g", d);
			return std::string(b);
		};
		progress_log(q, "\n%16.1f  Seconds to load index\n", load_s);
		if (map_s < 180) progress_log(q, "%16.1f  Seconds in mapper\n", map_s);
		else if (map_s < 2 * 60 * 60) progress_log(q, "%16.1f  Minutes in mapper\n", map_s / 60.0);
		else progress_log(q, "%16.1f  Hours in mapper\n", map_s / 3600.0);
		progress_log(q, "%16s  Reads (%s)\n", commas(n_reads).c_str(), short_int(n_reads).c_str());
		if (o.gpus == 1) progress_log(q, "%16.0f  Reads/sec. (GPU %d)\n", map_s > 0 ? (double)n_reads / map_s : 0.0, o.gpu);
		else progress_log(q, "%16.0f  Reads/sec. (%d GPUs)\n", map_s > 0 ? (double)n_reads / map_s : 0.0, o.gpus);
		progress_log(q, "%16s  Mapped Q>=%u (%.1f%%)\n", commas(n_accept).c_str(), minq, pct(n_accept));
		progress_log(q, "%16s  Mapped Q< %u (%.1f%%)\n", commas(n_reject).c_str(), minq, pct(n_reject));
		progress_log(q, "%16s  Unmapped (%.1f%%)\n\n", commas(n_nohit).c_str(), pct(n_nohit));
		if (o.minq_given && !paired) progress_log(q, "\nWARNING: Option -minq not used\n\n");
	}
	urmapx_index_close(I);
	const unsigned long long n_unsupported = rep.unsupported;
	if (n_unsupported) die("%llu reads fell outside the device path's domain (length or list overflow); their records are not valid", n_unsupported);
	return 0;
}

// first prime >= n (deterministic Miller-Rabin for 64-bit integers)
static uint64_t mulmod64(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((unsigned __int128)a * b % m); }
static uint64_t powmod64(uint64_t a, uint64_t e, uint64_t m) {
	uint64_t r = 1;
	for (a %= m; e; e >>= 1, a = mulmod64(a, a, m))
		if (e & 1) r = mulmod64(r, a, m);
	return r;
}
static bool is_prime64(uint64_t n) {
	if (n < 2) return false;
	for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
		if (n % p == 0) return n == p;
	}
	uint64_t d = n - 1;
	int r = 0;
	while ((d & 1) == 0) { d >>= 1; ++r; }
	for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
		uint64_t x = powmod64(a, d, n);
		if (x == 1 || x == n - 1) continue;
		bool comp = true;
		for (int i = 1; i < r && comp; ++i) {
			x = mulmod64(x, x, n);
			if (x == n - 1) comp = false;
		}
		if (comp) return false;
	}
	return true;
}

// GetPrime (prime.cpp:11-21) walks a ladder of 410 primes and returns the first one >= n.  The ladder is regular:
// rung k is the first prime >= t_k with t_0 = 100 and t_(k+1) = trunc(t_k / 0.95) in double arithmetic (every rung of
// the reference's table follows it; tests/test_abi_cpu.py holds the spot checks), so it is generated, not stored.
// Returns 0 past the last rung, where the reference dies.
static uint64_t get_prime(uint64_t n) {
	uint64_t t = 100;
	for (int k = 0; k < 410; ++k) {
		uint64_t p = t;
		while (!is_prime64(p)) ++p;
		if (p >= n) return p;
		t = (uint64_t)((double)t / 0.95);
	}
	return 0;
}

static int cmd_make_ufi(const Opts &o) {
	if (o.output.empty()) die("-output option required");
	uint64_t slots = o.slots;
	if (slots == 0) {
		// cmd_make_ufi (ufindexio.cpp:138-150): slots = GetPrime(file size / load factor 0.6)
		FILE *f = fopen(o.make_ufi.c_str(), "rb");
		if (!f) die("Cannot open %s", o.make_ufi.c_str());
		fseeko(f, 0, SEEK_END);
		const int64_t size = (int64_t)ftello(f);
		fclose(f);
		if (!(o.load_factor > 0)) die("-load_factor must be positive");
		slots = get_prime((uint64_t)(int64_t)((double)size / o.load_factor));  // int64(GenomeSize/LoadFactor), ufindexio.cpp:146
		if (slots == 0) die("GetPrime(%.3g) overflow", (double)size / o.load_factor);
	}
	unsigned maxix = o.maxix ? o.maxix : (o.veryfast ? 3u : 32u);
	// counting passes, head slots and the overflow list on the GPU, the order-dependent inserts on the host; -host (or no
	// usable device) builds everything on the host.  Same bytes either way.
	const unsigned flags = o.notrunclabels ? URMAPX_UFI_KEEP_LABELS : 0u;
	int rc = o.host_build ? URMAPX_E_NODEVICE : urmapx_make_ufi_opts(o.gpu, o.make_ufi.c_str(), o.output.c_str(), o.wordlength, maxix, slots, flags);
	if (rc == URMAPX_E_NODEVICE || (rc == URMAPX_E_NOMEM && !o.host_build)) {  // the GPU passes need ~17 bytes per slot of HBM
		if (!o.host_build) fprintf(stderr, "make_ufi: %s, building on the host\n", rc == URMAPX_E_NOMEM ? "not enough GPU memory for the counting passes" : "no usable GPU");
		rc = urmapx_make_ufi_opts(-1, o.make_ufi.c_str(), o.output.c_str(), o.wordlength, maxix, slots, flags);
	}
	check(rc, "make_ufi");
	return 0;
}

// cmd_ufi_validate (ufistats.cpp:141-147): FromFile + UFIndex::Validate (ufindex.cpp:611-658), the pass itself on the device
static int cmd_ufi_validate(const Opts &o) {
	urmapx_index *I = nullptr;
	setenv("URMAPX_NO_CHAIN_ROWS", "1", 0);  // the pass reads the table itself; the derived row layout is not needed for it
	if (getenv("URMAPX_HOST_INDEX")) {
		check(urmapx_index_open(o.ufi_validate.c_str(), &I), ("Reading index " + o.ufi_validate).c_str());
		check(urmapx_index_upload(I, o.gpu), "index upload");
	} else {  // the file streams to the device (urmapx_index_open_device)
		const int lrc = urmapx_index_open_device(o.ufi_validate.c_str(), o.gpu, &I);
		if (lrc == URMAPX_E_NODEVICE || lrc == URMAPX_E_NOMEM) check(lrc, "index upload");
		check(lrc, ("Reading index " + o.ufi_validate).c_str());
	}
	urmapx_validate_report r;
	const int rc = urmapx_index_validate(I, &r);
	if (rc != URMAPX_OK && rc != URMAPX_E_FORMAT) check(rc, "ufi_validate");
	progress_log(o.quiet, "%llu slots, %llu used, %llu rows, %llu positions re-hashed in %.3f s (GPU %d)\n", (unsigned long long)r.slots,
	             (unsigned long long)r.used, (unsigned long long)r.heads, (unsigned long long)r.positions, r.seconds, o.gpu);
	urmapx_index_close(I);
	if (rc == URMAPX_E_FORMAT) {
		// the reference stops at the first failing slot with "WordToSlot != Slot" (ufindex.cpp:642) or a failed assertion
		const char *what = r.bad_hash ? "WordToSlot != Slot" : r.bad_pos ? "Pos >= SeqDataSize" : r.bad_link ? "broken chain link" :
		                   r.bad_len ? "row longer than MaxIx" : "used slots not all on a chain";
		die("%s: first bad slot 0x%llx (%llu hash, %llu position, %llu link, %llu length failures; %llu slots used, %llu reached)", what,
		    (unsigned long long)r.first_bad_slot, (unsigned long long)r.bad_hash, (unsigned long long)r.bad_pos, (unsigned long long)r.bad_link,
		    (unsigned long long)r.bad_len, (unsigned long long)r.used, (unsigned long long)r.reached);
	}
	return 0;
}

int main(int argc, char **argv) {
	setenv("OMP_WAIT_POLICY", "passive", 0);  // idle pool threads sleep: three pipeline stages share the cores
	// The HIP runtime spreads a process's streams over FOUR hardware queues unless told otherwise, and streams that share one take turns: two lanes are four
	// streams (a lane's own and its copy-back stream), beside the loader's.  Before the runtime starts (round 6: the lanes of a process with a few more streams
	// ran at 29 M reads/s instead of 39 M until this was set; profiles/r6/hw_queues.txt).  The library sets the same when it is loaded (urmapx.hip).
	setenv("GPU_MAX_HW_QUEUES", "16", 0);
	Opts o = parse(argc, argv);
	log_open(o, argc, argv);
	if (!o.map.empty() || !o.map2.empty()) { const int rc = cmd_map(o, argc, argv); log_close(); return rc; }
	if (!o.make_ufi.empty()) { const int rc = cmd_make_ufi(o); log_close(); return rc; }
	if (!o.ufi_validate.empty()) { const int rc = cmd_ufi_validate(o); log_close(); return rc; }
	fprintf(stderr, "urmap (MI355X build)\n  urmap -map reads.fq -ufi index.ufi -samout out.sam [-veryfast] [-gpu D] [-gpus N] [-streams K] [-samshards N]\n"
	                "  urmap -map2 R1.fq -reverse R2.fq -ufi index.ufi -samout out.sam [-tabbedout out.tab] [-gpu D] [-gpus N]\n"
	                "  urmap -make_ufi genome.fa -output index.ufi [-slots N] [-wordlength W] [-maxix M]\n"
	                "  urmap -ufi_validate index.ufi [-gpu D]\n");
	return 0;
}
