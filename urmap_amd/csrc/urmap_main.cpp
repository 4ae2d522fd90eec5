// urmap_main.cpp -- command line of the MI355X build: the reference's `urmap -map` / `-make_ufi` surface
// (urmap_main.cpp:6-41, map.cpp:27-67, ufindexio.cpp:117-179) as a batch dispatcher over liburmapx.so.
//
//   urmap -map reads.fq[.gz] -ufi index.ufi -samout out.sam [-veryfast] [-threads N] [-gpu D] [-gpus N] [-streams K] [-batch N]
//   urmap -make_ufi genome.fa -output index.ufi [-slots N] [-wordlength W] [-maxix M] [-veryfast]
//
//   urmap -map2 R1.fq -reverse R2.fq -ufi index.ufi -samout out.sam [-tabbedout out.tab]   (paired-end, map2.cpp:39-90)
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
	std::string map, map2, reverse, make_ufi, ufi, samout, tabbedout, output;
	bool veryfast = false, quiet = false, minq_given = false;
	unsigned threads = 0, wordlength = 24, maxix = 0, minq = 10;
	unsigned long long slots = 0;
	int gpu = 0, gpus = 1, streams = 2;
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
		else if (a == "-batch") o.batch = (unsigned)atoi(val());
		else if (a == "-veryfast") o.veryfast = true;
		else if (a == "-quiet") o.quiet = true;
		else if (a == "-log") (void)val();
		else die("Unknown option %s", a.c_str());
	}
	return o;
}

static void check(int rc, const char *what) {
	if (rc != URMAPX_OK && rc != URMAPX_E_UNSUPPORTED) die("%s: %s", what, urmapx_strerror(rc));
}

struct Job {
	FastqBatch reads;
	std::vector<urmapx_result> results;
	std::vector<urmapx_path_op> ops;
	std::vector<urmapx_pair_info> info;  // -tabbedout
};

template <class T>
class Channel {  // bounded single-producer single-consumer queue
public:
	explicit Channel(size_t cap) : cap_(cap) {}
	void push(T v) {
		std::unique_lock<std::mutex> l(m_);
		cv_.wait(l, [&] { return q_.size() < cap_; });
		q_.push_back(std::move(v));
		cv_.notify_all();
	}
	bool pop(T &v) {
		std::unique_lock<std::mutex> l(m_);
		cv_.wait(l, [&] { return !q_.empty() || closed_; });
		if (q_.empty()) return false;
		v = std::move(q_.front());
		q_.pop_front();
		cv_.notify_all();
		return true;
	}
	bool try_pop(T &v) {
		std::lock_guard<std::mutex> l(m_);
		if (q_.empty()) return false;
		v = std::move(q_.front());
		q_.pop_front();
		cv_.notify_all();
		return true;
	}
	void close() {
		std::lock_guard<std::mutex> l(m_);
		closed_ = true;
		cv_.notify_all();
	}

private:
	std::mutex m_;
	std::condition_variable cv_;
	std::deque<T> q_;
	size_t cap_;
	bool closed_ = false;
};

extern "C" size_t urmapx_sam_pe(const urmapx_index *, const urmapx_result *, const urmapx_result *, const urmapx_path_op *,
                                const char *, const uint8_t *, const uint8_t *, uint32_t, const char *, const uint8_t *,
                                const uint8_t *, uint32_t, char *, size_t);

static int cmd_map(const Opts &o, int argc, char **argv) {
	const bool paired = !o.map2.empty();
	const unsigned minq = paired ? o.minq : 10;  // only cmd_map2 reads -minq (map2.cpp:76); -map keeps State1::m_Minq = 10
	if (paired && o.reverse.empty()) die("-reverse required");
	if (o.ufi.empty()) die("-ufi option required");
	const auto t0 = std::chrono::steady_clock::now();
	if (o.gpus < 1 || o.gpus > 64) die("-gpus must be 1..64");
	if (o.streams < 1 || o.streams > 8) die("-streams must be 1..8");
	// URMAPX_FORCE_DEVICE=d (test aid): every lane runs on physical device d, so that the -gpus N code path can be
	// exercised on a machine with one GPU
	const char *forced = getenv("URMAPX_FORCE_DEVICE");
	auto phys = [&](int g) { return forced ? atoi(forced) : o.gpu + g; };
	urmapx_index *I = nullptr;
	check(urmapx_index_open(o.ufi.c_str(), &I), ("Reading index " + o.ufi).c_str());
	urmapx_params P;
	check(urmapx_params_for_method((o.veryfast && o.map2.empty()) ? 7 : 6, &P), "SetMethod");  // -map2 always uses method 6 (map2.cpp:15-16)
	// one replica of the index per device (uploaded concurrently), K mapping contexts on each
	const int n_lanes = o.gpus * o.streams;
	std::vector<urmapx_index *> replicas((size_t)o.gpus, nullptr);
	{
		std::vector<int> rcs((size_t)o.gpus, 0);
		std::vector<std::thread> up;
		for (int g = 0; g < o.gpus; ++g)
			up.emplace_back([&, g] {
				if (g == 0) { rcs[0] = urmapx_index_upload(I, phys(0)); replicas[0] = I; }
				else rcs[(size_t)g] = urmapx_index_replicate(I, phys(g), &replicas[(size_t)g]);
			});
		for (auto &t : up) t.join();
		for (int g = 0; g < o.gpus; ++g) check(rcs[(size_t)g], "Uploading index to the GPU");
	}
	std::vector<urmapx_ctx *> ctxs((size_t)n_lanes, nullptr);
	for (int l = 0; l < n_lanes; ++l) {
		check(urmapx_ctx_create(replicas[(size_t)(l % o.gpus)], phys(l % o.gpus), &P, &ctxs[(size_t)l]), "Creating mapping context");
		if (!o.map2.empty() && o.veryfast) check(urmapx_ctx_set_pe_veryfast(ctxs[(size_t)l], 1), "Search5");
	}
	if (o.veryfast && urmapx_index_max_ix(I) > 3) fprintf(stderr, "\nWARNING: index not optimal for -veryfast\n");
	// host threads for FASTQ parsing and SAM formatting (-threads; the mapping itself runs on the GPU)
	int host_threads = o.threads ? (int)o.threads : std::min(16, std::max(1, (int)std::thread::hardware_concurrency()));
	omp_set_num_threads(host_threads);
	int fsam = -1;
	uint64_t sam_off = 0;
	std::atomic<bool> write_failed{false};
	if (!o.samout.empty()) {
		fsam = open(o.samout.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fsam < 0) die("Cannot create %s", o.samout.c_str());
		std::string hdr;
		append_sam_header(hdr, I, argc, argv);
		if (write(fsam, hdr.data(), hdr.size()) != (ssize_t)hdr.size()) die("Cannot write %s", o.samout.c_str());
		sam_off = hdr.size();
	}
	// -tabbedout (outfiles.cpp:7-12): State2::OutputTab2's line per pair; only -map2 writes it
	FILE *ftab = nullptr;
	if (!o.tabbedout.empty()) {
		ftab = fopen(o.tabbedout.c_str(), "wb");
		if (!ftab) die("Cannot create %s", o.tabbedout.c_str());
		if (paired)
			for (urmapx_ctx *C : ctxs) check(urmapx_ctx_set_pair_info(C, 1), "pair info");
	}
	FastqReader rd, rd2;
	std::string err;
	if (!rd.open(paired ? o.map2 : o.map, err)) die("%s", err.c_str());
	if (paired && !rd2.open(o.reverse, err)) die("%s", err.c_str());
	const auto t1 = std::chrono::steady_clock::now();

	// batch b travels through parsed[b mod lanes] -> lane thread -> mapped[b mod lanes]; the writer visits the lanes in the
	// same round-robin order, so batches come back in input order without a reorder buffer
	using JobChannel = Channel<std::unique_ptr<Job>>;
	std::vector<std::unique_ptr<JobChannel>> parsed, mapped;
	for (int l = 0; l < n_lanes; ++l) {
		parsed.emplace_back(new JobChannel(2));
		mapped.emplace_back(new JobChannel(1));
	}
	JobChannel recycled((size_t)(8 + 6 * n_lanes));  // finished jobs go back to the reader: their arrays are reused
	double t_parse = 0, t_gpu = 0, t_format = 0, t_write = 0;  // busy seconds per stage (URMAPX_VERBOSE)
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
	std::string reader_err;
	// -map2: the second file is parsed by its own thread while the reader parses the first
	struct Side { FastqBatch b; std::string e; bool more = false; } side2;
	Channel<int> go2(1), done2(1);
	std::thread reader2;
	if (paired)
		reader2 = std::thread([&] {
			omp_set_num_threads(std::max(1, host_threads / 2));  // a new thread starts from the default team size, not main's
			int x;
			while (go2.pop(x)) {
				side2.b.clear();
				side2.e.clear();
				side2.more = rd2.next_batch(side2.b, o.batch / 2, side2.e);
				done2.push(1);
			}
		});
	std::thread reader([&] {
		omp_set_num_threads(paired ? std::max(1, host_threads - host_threads / 2) : host_threads);
		FastqBatch a;
		for (size_t b = 0;; ++b) {
			std::unique_ptr<Job> j;
			if (!recycled.try_pop(j)) j = std::make_unique<Job>();
			j->reads.clear();
			std::string e;
			bool more;
			const auto tp0 = now();
			if (!paired)
				more = rd.next_batch(j->reads, o.batch, e);
			else {  // mates interleaved: reads 2i, 2i+1 (map2.cpp:27-32 reads one record from each file under one lock)
				go2.push(1);
				a.clear();
				more = rd.next_batch(a, o.batch / 2, e);
				int x;
				done2.pop(x);
				const FastqBatch &b = side2.b;
				if (e.empty()) e = side2.e;
				if (e.empty() && (more != side2.more || a.size() != b.size())) e = std::string("Premature end of file in FASTQ") + (a.size() > b.size() ? "2" : "1");
				if (e.empty()) {
					omp_set_num_threads(host_threads);
					interleave_batches(a, b, j->reads);
					omp_set_num_threads(std::max(1, host_threads - host_threads / 2));
				}
			}
			t_parse += secs(tp0, now());
			if (!e.empty()) { reader_err = e; break; }
			if (!more) break;
			parsed[b % (size_t)n_lanes]->push(std::move(j));
		}
		for (auto &c : parsed) c->close();
		go2.close();
	});
	unsigned long long n_reads = 0, n_accept = 0, n_reject = 0, n_nohit = 0, n_unsupported = 0;
	std::thread writer([&] {
		omp_set_num_threads(host_threads);
		// SAM text of a batch is formatted by all host threads, each on a contiguous range of reads (pairs), and the
		// pieces are written at their file offsets in input order.
		std::unique_ptr<Job> j;
		std::vector<std::string> outs((size_t)host_threads);
		struct Cnt { unsigned long long accept = 0, reject = 0, nohit = 0, unsupported = 0; };
		for (size_t b = 0; mapped[b % (size_t)n_lanes]->pop(j); ++b) {
			const uint32_t n = j->reads.size();
			const uint32_t units = paired ? n / 2 : n;
			std::vector<Cnt> cnt((size_t)host_threads);
			const auto tf0 = now();
#pragma omp parallel for schedule(static, 1) num_threads(host_threads)
			for (int t = 0; t < host_threads; ++t) {
				std::string &out = outs[(size_t)t];
				out.clear();
				Cnt &c = cnt[(size_t)t];
				const uint32_t u0 = (uint32_t)((uint64_t)units * (uint64_t)t / (uint64_t)host_threads);
				const uint32_t u1 = (uint32_t)((uint64_t)units * (uint64_t)(t + 1) / (uint64_t)host_threads);
				std::vector<char> pbuf;
				for (uint32_t i = paired ? 2 * u0 : u0; i < (paired ? 2 * u1 : u1); ++i) {
					const urmapx_result &r = j->results[i];
					const uint64_t off = j->reads.offs[i];
					const unsigned L = (unsigned)(j->reads.offs[i + 1] - off);
					if (fsam >= 0 && !paired)
						append_sam_record(out, I, r, j->ops.data(), 0, "*", 0xFFFFFFFFu, 0, j->reads.label(i),
						                  j->reads.bases.data() + off, j->reads.quals.data() + off, L);
					if (fsam >= 0 && paired && (i & 1) == 0) {
						const uint64_t off2 = j->reads.offs[i + 1];
						const unsigned L2 = (unsigned)(j->reads.offs[i + 2] - off2);
						pbuf.resize(strlen(j->reads.label(i)) + strlen(j->reads.label(i + 1)) + 3 * (size_t)(L + L2) + 2048);
						size_t k = urmapx_sam_pe(I, &j->results[i], &j->results[i + 1], j->ops.data(), j->reads.label(i),
						                         j->reads.bases.data() + off, j->reads.quals.data() + off, L,
						                         j->reads.label(i + 1), j->reads.bases.data() + off2,
						                         j->reads.quals.data() + off2, L2, pbuf.data(), pbuf.size());
						out.append(pbuf.data(), k);
					}
					// HitStats counters (output1.cpp:20-30)
					if (r.status) ++c.unsupported;
					if (r.dbpos == 0xFFFFFFFFu) ++c.nohit;
					else if (r.mapq >= minq) ++c.accept;
					else ++c.reject;
				}
			}
			for (const Cnt &c : cnt) { n_accept += c.accept; n_reject += c.reject; n_nohit += c.nohit; n_unsupported += c.unsupported; }
			n_reads += n;
			const auto tf1 = now();
			t_format += secs(tf0, tf1);
			if (fsam >= 0) {
				std::vector<uint64_t> at((size_t)host_threads + 1);
				at[0] = sam_off;
				for (int t = 0; t < host_threads; ++t) at[(size_t)t + 1] = at[(size_t)t] + outs[(size_t)t].size();
				sam_off = at[(size_t)host_threads];
#pragma omp parallel for schedule(static, 1) num_threads(host_threads)
				for (int t = 0; t < host_threads; ++t) {
					const std::string &out = outs[(size_t)t];
					size_t done = 0;
					while (done < out.size()) {
						ssize_t w = pwrite(fsam, out.data() + done, out.size() - done, (off_t)(at[(size_t)t] + done));
						if (w <= 0) { write_failed = true; break; }
						done += (size_t)w;
					}
				}
			}
			if (ftab && paired) {  // tab lines: formatted by all host threads (pair ranges), written in order
				std::vector<std::string> tabs((size_t)host_threads);
#pragma omp parallel for schedule(static, 1) num_threads(host_threads)
				for (int t = 0; t < host_threads; ++t) {
					const uint32_t u0 = (uint32_t)((uint64_t)units * (uint64_t)t / (uint64_t)host_threads);
					const uint32_t u1 = (uint32_t)((uint64_t)units * (uint64_t)(t + 1) / (uint64_t)host_threads);
					char line[4096];
					for (uint32_t u = u0; u < u1; ++u) {
						const uint32_t i = 2 * u;
						const unsigned L1 = (unsigned)(j->reads.offs[i + 1] - j->reads.offs[i]), L2 = (unsigned)(j->reads.offs[i + 2] - j->reads.offs[i + 1]);
						const size_t k = urmapx_tab_pe(I, &j->results[i], &j->results[i + 1], &j->info[u], j->reads.label(i), L1, L2,
						                               fsam >= 0 ? 1 : 0, line, sizeof line);
						tabs[(size_t)t].append(line, k);
					}
				}
				for (const std::string &tb : tabs)
					if (fwrite(tb.data(), 1, tb.size(), ftab) != tb.size()) write_failed = true;
			}
			t_write += secs(tf1, now());
			recycled.push(std::move(j));
		}
	});
	std::mutex gpu_time_lock;
	std::vector<std::thread> lanes;
	for (int l = 0; l < n_lanes; ++l)
		lanes.emplace_back([&, l] {
			urmapx_ctx *C = ctxs[(size_t)l];
			std::unique_ptr<Job> j;
			while (parsed[(size_t)l]->pop(j)) {
				const uint32_t n = j->reads.size();
				j->results.resize(n);
				j->ops.resize((size_t)n * URMAPX_MAX_PATH_OPS);
				size_t used = 0;
				const auto tg0 = now();
				int rc = paired ? urmapx_map_pe(C, j->reads.bases.data(), j->reads.offs.data(), n / 2, j->results.data(), j->ops.data(),
				                                j->ops.size(), &used)
				                : urmapx_map_se(C, j->reads.bases.data(), j->reads.offs.data(), n, j->results.data(), j->ops.data(),
				                                j->ops.size(), &used);
				check(rc, paired ? "urmapx_map_pe" : "urmapx_map_se");
				if (paired && ftab) {
					j->info.resize(n / 2);
					check(urmapx_ctx_get_pair_info(C, j->info.data(), n / 2), "urmapx_ctx_get_pair_info");
				}
				{
					std::lock_guard<std::mutex> g(gpu_time_lock);
					t_gpu += secs(tg0, now());
				}
				mapped[(size_t)l]->push(std::move(j));
			}
			mapped[(size_t)l]->close();
		});
	for (auto &t : lanes) t.join();
	reader.join();
	if (reader2.joinable()) reader2.join();
	writer.join();
	if (!reader_err.empty()) die("%s", reader_err.c_str());
	if (fsam >= 0) close(fsam);
	if (ftab) fclose(ftab);
	if (write_failed) die("Error writing %s", o.samout.c_str());
	const auto t2 = std::chrono::steady_clock::now();
	const double load_s = std::chrono::duration<double>(t1 - t0).count();
	const double map_s = std::chrono::duration<double>(t2 - t1).count();
	if (getenv("URMAPX_VERBOSE"))
		fprintf(stderr, "stage busy seconds: parse %.2f, gpu (copies + kernels, summed over %d lanes) %.2f, format %.2f, write %.2f; %d host threads\n",
		        t_parse, n_lanes, t_gpu, t_format, t_write, host_threads);
	if (!o.quiet) {  // State1::HitStats (state1.cpp:593-632): same lines, sub-second timers, "GPU n" where it says "n threads"
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
		fprintf(stderr, "\n%16.1f  Seconds to load index\n", load_s);
		if (map_s < 180) fprintf(stderr, "%16.1f  Seconds in mapper\n", map_s);
		else if (map_s < 2 * 60 * 60) fprintf(stderr, "%16.1f  Minutes in mapper\n", map_s / 60.0);
		else fprintf(stderr, "%16.1f  Hours in mapper\n", map_s / 3600.0);
		fprintf(stderr, "%16s  Reads (%s)\n", commas(n_reads).c_str(), short_int(n_reads).c_str());
		if (o.gpus == 1) fprintf(stderr, "%16.0f  Reads/sec. (GPU %d)\n", map_s > 0 ? (double)n_reads / map_s : 0.0, o.gpu);
		else fprintf(stderr, "%16.0f  Reads/sec. (%d GPUs)\n", map_s > 0 ? (double)n_reads / map_s : 0.0, o.gpus);
		fprintf(stderr, "%16s  Mapped Q>=%u (%.1f%%)\n", commas(n_accept).c_str(), minq, pct(n_accept));
		fprintf(stderr, "%16s  Mapped Q< %u (%.1f%%)\n", commas(n_reject).c_str(), minq, pct(n_reject));
		fprintf(stderr, "%16s  Unmapped (%.1f%%)\n\n", commas(n_nohit).c_str(), pct(n_nohit));
		if (o.minq_given && !paired) fprintf(stderr, "\nWARNING: Option -minq not used\n\n");
	}
	for (urmapx_ctx *C : ctxs) urmapx_ctx_destroy(C);
	for (int g = 1; g < o.gpus; ++g) urmapx_index_close(replicas[(size_t)g]);
	urmapx_index_close(I);
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
		slots = get_prime((uint64_t)(int64_t)((double)size / 0.6));
		if (slots == 0) die("GetPrime(%.3g) overflow", (double)size / 0.6);
	}
	unsigned maxix = o.maxix ? o.maxix : (o.veryfast ? 3u : 32u);
	check(urmapx_make_ufi(o.make_ufi.c_str(), o.output.c_str(), o.wordlength, maxix, slots), "make_ufi");
	return 0;
}

int main(int argc, char **argv) {
	setenv("OMP_WAIT_POLICY", "passive", 0);  // idle pool threads sleep: three pipeline stages share the cores
	Opts o = parse(argc, argv);
	if (!o.map.empty() || !o.map2.empty()) return cmd_map(o, argc, argv);
	if (!o.make_ufi.empty()) return cmd_make_ufi(o);
	fprintf(stderr, "urmap (MI355X build)\n  urmap -map reads.fq -ufi index.ufi -samout out.sam [-veryfast] [-gpu D] [-gpus N] [-streams K]\n"
	                "  urmap -map2 R1.fq -reverse R2.fq -ufi index.ufi -samout out.sam [-tabbedout out.tab] [-gpu D] [-gpus N]\n"
	                "  urmap -make_ufi genome.fa -output index.ufi [-slots N] [-wordlength W] [-maxix M]\n");
	return 0;
}
