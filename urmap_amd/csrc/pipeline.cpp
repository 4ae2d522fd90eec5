// pipeline.cpp -- FASTQ file(s) -> SAM file through the mapping path: cmd_map / cmd_map2 (map.cpp:27-67, map2.cpp:39-90)
// as a library call (urmapx_map_files), so that the command line and an embedding program with an index already resident
// in HBM run the same code.
//
// One reader thread parses FASTQ into batches; batch b goes to mapping lane b mod (N*K), a host thread with its own
// mapping context on GPU first_gpu + (b mod N) (N devices, each holding its own replica of the index; K contexts per
// device so that one lane's copies overlap another's kernels); a writer thread takes the batches back in input order and
// formats and writes their SAM with all host threads.  The reference fans reads over its OpenMP threads the same way
// (map.cpp:58-61, seqsource.cpp:30-66) but writes in completion order; here records are written in input
// order.  No data moves between devices.  The batch arrays that cross PCIe are page-locked once they have their size.
//
// Single-end input from a plain file takes the shorter road first (text phase): the reader cuts the file into chunks at
// record starts and reads them into page-locked buffers, a lane hands the chunk's BYTES to the device and gets the bytes
// of its SAM records back (text_gpu.hip: line ends, record checks, search, SAM text all on the GPU), and the writer
// copies them into the output file.  The host then only moves bytes.  A chunk the device parser does not take as it is
// ends the text phase; the host reader continues at that chunk's first byte with the line count kept, so '\r', blank
// lines and malformed records get the reference's handling and messages.
#include <fcntl.h>
#include <sched.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <hip/hip_runtime_api.h>
#include <omp.h>
#include <unistd.h>
#include <zlib.h>

#include "pgzip.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
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
#include "internal.h"
#include "sam.h"

using namespace urx;

namespace {

// Page-locks the storage of a PodVec for as long as it keeps its address (hipHostRegister); the vector tells the pin
// before it reallocates or frees, and the next use registers the new storage.
template <class T>
struct Pin {
	PodVec<T> *v = nullptr;
	void *p = nullptr;
	size_t n = 0;
	static void on_free(void *ctx, void *) { static_cast<Pin *>(ctx)->drop(); }
	void hold(PodVec<T> &vec) {
		if (!v) { v = &vec; vec.set_pre_free(&Pin::on_free, this); }
		void *q = vec.data();
		const size_t bytes = vec.capacity() * sizeof(T);
		if (q == p && bytes == n) return;
		drop();
		if (q && bytes && hipHostRegister(q, bytes, hipHostRegisterDefault) == hipSuccess) { p = q; n = bytes; }
		else (void)hipGetLastError();
	}
	void drop() {
		if (p) (void)hipHostUnregister(p);
		p = nullptr; n = 0;
	}
	~Pin() {
		drop();
		if (v) v->set_pre_free(nullptr, nullptr);
	}
};

struct Job {
	FastqBatch reads;
	PodVec<urmapx_result> results;
	PodVec<urmapx_path_op> ops;
	std::vector<urmapx_pair_info> info;  // -tabbedout
	Pin<uint8_t> pin_bases;
	Pin<uint64_t> pin_offs;
	Pin<urmapx_result> pin_results;
	Pin<urmapx_path_op> pin_ops;
};

template <class T>
class Channel {  // bounded queue
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

// URMAPX_PIPE_TRACE=1: every stage of every chunk with its start and end (ms since the first read), to stderr at the end
struct Trace {
	bool on = getenv("URMAPX_PIPE_TRACE") != nullptr;
	std::mutex m;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	struct Ev { const char *what; int lane; size_t job; double a, b; };
	std::vector<Ev> evs;
	double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
	void add(const char *what, int lane, size_t job, double a) {
		if (!on) return;
		const double b = ms();
		std::lock_guard<std::mutex> l(m);
		evs.push_back(Ev{what, lane, job, a, b});
	}
	void dump() {
		if (!on) return;
		std::sort(evs.begin(), evs.end(), [](const Ev &x, const Ev &y) { return x.a < y.a; });
		for (const Ev &e : evs) fprintf(stderr, "trace %-10s lane %d job %3zu  %9.2f .. %9.2f  (%7.2f ms)\n", e.what, e.lane, e.job, e.a, e.b, e.b - e.a);
	}
};

struct Failure {
	std::atomic<bool> set{false};
	std::mutex m;
	int code = 0;
	std::string msg;
	void raise(int c, const std::string &s) {
		std::lock_guard<std::mutex> l(m);
		if (!set.load()) { code = c; msg = s; set.store(true); }
	}
};


// The SAM file.  Pieces are copied to their offsets by several threads: through a shared mapping of the file where that
// works (parallel page faults; write() on one file is serialised by the inode lock), else with pwrite.
class FileSink {
public:
	~FileSink() { if (fd_ >= 0) ::close(fd_); }
	// measurement (urmapx_map_options.discard_sam): the text is taken and dropped
	void open_discard() { discard_ = true; }
	bool open(const char *path) {
		const char *mode = getenv("URMAPX_SAM_WRITE");  // "pwrite" (default) | "mmap"
		const bool want_map = mode && !strcmp(mode, "mmap");
		fd_ = ::open(path, (want_map ? O_RDWR : O_WRONLY) | O_CREAT | O_TRUNC, 0644);
		if (fd_ < 0 && want_map) fd_ = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fd_ < 0) return false;
		struct stat st;
		const bool regular = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
		sequential_ = !regular;  // a pipe, a FIFO, a terminal (-samout /dev/stdout | ...): bytes go out in order with write()
		use_map_ = regular && want_map;
		return true;
	}
	bool is_open() const { return fd_ >= 0; }
	bool seekable() const { return !sequential_; }
	// Threads that should share one piece's pwrite: on tmpfs (and ramfs) write() is a memcpy under the inode lock -- one
	// thread does 5.8 GB/s, eight together 3.7 (scripts/fs_bench.cpp) -- so one; on a disk file system concurrent pwrites
	// of disjoint ranges overlap their I/O, so a few.
	int writer_threads(int host_threads) const {
		if (discard_) return 1;
		if (const char *e = getenv("URMAPX_WRITE_THREADS")) return std::max(1, atoi(e));
		if (sequential_ || fd_ < 0) return 1;
		struct statfs fs;
		if (fstatfs(fd_, &fs) != 0) return 1;
		const long t = (long)fs.f_type;
		if (t == 0x01021994L /* tmpfs */ || t == (long)0x858458f6L /* ramfs */) return 1;
		return std::max(1, std::min(4, host_threads));
	}
	const char *medium() const {
		if (discard_) return "discarded";
		if (sequential_) return "pipe";
		struct statfs fs;
		if (fd_ < 0 || fstatfs(fd_, &fs) != 0) return "file";
		const long t = (long)fs.f_type;
		return (t == 0x01021994L || t == (long)0x858458f6L) ? "tmpfs" : "disk file system";
	}
	// makes the file at least `end` bytes long (pieces below `end` can then be written from several threads at once)
	bool reserve(uint64_t end) {
		if (!use_map_ || end <= size_) return true;
		if (ftruncate(fd_, (off_t)end) != 0) { use_map_ = false; return true; }
		size_ = end;
		return true;
	}
	bool write_at(const char *p, size_t n, uint64_t off, int threads) {
		if (n == 0 || discard_) return true;
		if (threads < 1) threads = 1;
		if (sequential_) {  // the callers hand pieces over in file order when seekable() is false
			if (off != size_) return false;
			size_t done = 0;
			while (done < n) {
				ssize_t w = ::write(fd_, p + done, n - done);
				if (w <= 0) return false;
				done += (size_t)w;
			}
			size_ = off + n;
			return true;
		}
		if (use_map_) {
			if (off + n > size_ && ftruncate(fd_, (off_t)(off + n)) != 0) use_map_ = false;
			else {
				if (off + n > size_) size_ = off + n;
				const uint64_t a = off & ~(uint64_t)4095;
				const size_t len = (size_t)(off + n - a);
				char *m = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, (off_t)a);
				if (m == MAP_FAILED) use_map_ = false;
				else {
					char *dst = m + (off - a);
#pragma omp parallel for schedule(static, 1) num_threads(threads)
					for (int t = 0; t < threads; ++t) {
						const size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
						memcpy(dst + lo, p + lo, hi - lo);
					}
					munmap(m, len);
					return true;
				}
			}
		}
		std::atomic<bool> ok{true};
#pragma omp parallel for schedule(static, 1) num_threads(threads)
		for (int t = 0; t < threads; ++t) {
			const size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
			size_t done = lo;
			while (done < hi) {
				ssize_t w = pwrite(fd_, p + done, hi - done, (off_t)(off + done));
				if (w <= 0) { ok.store(false); break; }
				done += (size_t)w;
			}
		}
		if (off + n > size_) size_ = off + n;
		return ok;
	}
	bool finish(uint64_t length) {
		if (fd_ < 0 || discard_) return true;
		bool ok = sequential_ || size_ == length || ftruncate(fd_, (off_t)length) == 0;
		ok = ::close(fd_) == 0 && ok;
		fd_ = -1;
		return ok;
	}

private:
	int fd_ = -1;
	bool sequential_ = false, discard_ = false;
	std::atomic<bool> use_map_{false};
	std::atomic<uint64_t> size_{0};
};

// Page-locked host buffers are kept for the next call of this process (pinning and unpinning a few hundred MB costs tens
// of milliseconds each way, as much as mapping the chunk that travels in them); urmapx_host_pool_trim() lets go of them.
class HostPool {
public:
	static HostPool &get() { static HostPool *p = new HostPool; return *p; }  // never destroyed: the runtime may be gone by then
	char *acquire(size_t want, size_t &cap) {
		{
			std::lock_guard<std::mutex> l(m_);
			size_t best = free_.size();
			for (size_t i = 0; i < free_.size(); ++i)
				if (free_[i].cap >= want && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
			if (best < free_.size() && free_[best].cap <= want + want / 2 + (8u << 20)) {
				char *p = free_[best].p;
				cap = free_[best].cap;
				held_ -= cap;
				free_.erase(free_.begin() + (long)best);
				return p;
			}
		}
		char *p = nullptr;
		if (const char *e = getenv("URMAPX_TEST_PINNED_ALLOCS")) {  // tests: the (N+1)-th fresh allocation of the process fails
			if (fresh_.fetch_add(1) >= (uint64_t)atoll(e)) return nullptr;
		}
		AllocTimer at(1);
		if (hipHostMalloc((void **)&p, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		cap = want;
		return p;
	}
	void release(char *p, size_t cap) {
		if (!p) return;
		{
			std::lock_guard<std::mutex> l(m_);
			if (held_ + cap <= kKeep) { free_.push_back(Buf{p, cap}); held_ += cap; return; }
		}
		AllocTimer at(1);
		(void)hipHostFree(p);
	}
	void trim() {
		std::vector<Buf> v;
		{
			std::lock_guard<std::mutex> l(m_);
			v.swap(free_);
			held_ = 0;
		}
		for (const Buf &b : v) (void)hipHostFree(b.p);
	}

private:
	struct Buf { char *p; size_t cap; };
	static constexpr size_t kKeep = (size_t)16 << 30;  // 2 pipelines x 8 chunks of 524 288 reads in flight = 6.3 GB, and the other shapes a process maps (smaller files, mate pairs) beside them
	std::mutex m_;
	std::vector<Buf> free_;
	size_t held_ = 0;
	std::atomic<uint64_t> fresh_{0};
};

// A lane's mapping context and text stage are kept for the next urmapx_map_files call on the same index and device: their
// device arrays (2.5-3 GB per lane: DpJobs, parked states, path arena, the chunk's text both ways) are otherwise allocated in
// the first chunk of every call and freed at its end, and on a device that has just freed as much, hipMalloc has been measured
// at 27 ms a call instead of 0.2 (48 of them per lane: 0.6 s in front of a 0.3 s run; profiles/r5/e2e_watch.txt).
// urmapx_index_close destroys the lanes of its index, urmapx_host_pool_trim all of them; URMAPX_NO_LANE_POOL=1 keeps none.
struct PooledLane {
	const urmapx_index *I;
	int device;
	urmapx_params P;
	bool veryfast, pair_info;
	urmapx_ctx *C;
	urmapx_text *T;
	uint64_t env = 0;  // what the URMAPX_* knobs were when the context was made (a context caches what they decide): reused only under the same
};
// every URMAPX_* variable of the environment, names and values, as one number
uint64_t env_knobs() {
	uint64_t h = 1469598103934665603ull;
	for (char **e = environ; e && *e; ++e)
		if (!strncmp(*e, "URMAPX_", 7) && strncmp(*e, "URMAPX_VERBOSE", 14) && strncmp(*e, "URMAPX_PIPE_TRACE", 17))
			for (const char *c = *e; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
	return h;
}
class LanePool {
public:
	static LanePool &get() { static LanePool *p = new LanePool; return *p; }  // never destroyed, like the page-locked pool
	static bool enabled() {
		if (getenv("URMAPX_NO_LANE_POOL")) return false;
		for (char **e = environ; e && *e; ++e)
			if (!strncmp(*e, "URMAPX_TEST_", 12) || !strncmp(*e, "URMAPX_DEBUG_", 13)) return false;  // contexts cache what such knobs decide
		return true;
	}
	bool take(const urmapx_index *I, int device, const urmapx_params &P, bool veryfast, bool pair_info, urmapx_ctx *&C, urmapx_text *&T) {
		std::lock_guard<std::mutex> l(m_);
		for (size_t i = 0; i < v_.size(); ++i) {
			const PooledLane &q = v_[i];
			if (q.I == I && q.device == device && q.veryfast == veryfast && q.pair_info == pair_info && q.env == env_knobs() && !memcmp(&q.P, &P, sizeof P)) {
				C = q.C; T = q.T;
				v_.erase(v_.begin() + (long)i);
				return true;
			}
		}
		return false;
	}
	void give(const PooledLane &q) {
		// a kept lane holds 2.5-3 GB of device arrays: it stays only while the device has room to spare -- an eighth of its memory still free
		// with the lane on it (a process that shares the device with other allocators saw the pool's memory as gone)
		size_t free_b = 0, total_b = 0;
		const bool roomy = hipSetDevice(q.device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= total_b / 8;
		if (roomy) {
			std::lock_guard<std::mutex> l(m_);
			if (v_.size() < kMax) { v_.push_back(q); return; }
		}
		destroy(q);
	}
	void purge(const urmapx_index *I) {
		std::vector<PooledLane> out;
		{
			std::lock_guard<std::mutex> l(m_);
			for (size_t i = 0; i < v_.size();)
				if (!I || v_[i].I == I) { out.push_back(v_[i]); v_.erase(v_.begin() + (long)i); }
				else ++i;
		}
		for (const PooledLane &q : out) destroy(q);
	}
	static void destroy(const PooledLane &q) {
		if (q.T) urmapx_text_destroy(q.T);
		if (q.C) urmapx_ctx_destroy(q.C);
	}

private:
	static constexpr size_t kMax = 8;
	std::mutex m_;
	std::vector<PooledLane> v_;
};

// one chunk of the text phase: FASTQ bytes in, SAM bytes out, both page-locked
struct TextJob {
	char *in = nullptr, *out = nullptr;
	size_t in_cap = 0, out_cap = 0, nbytes = 0, nbytes2 = 0;  // pairs: the mate file's chunk is in[...] too, at in2
	char *in2 = nullptr;
	uint64_t file_off = 0, file_off2 = 0;
	urmapx_text_report rep;
	int rc = 0;
	// -tabbedout: the chunk's lines, formatted by the lane's host threads from what urmapx_text_fetch_pairs brings back
	std::string tab;
	std::vector<urmapx_result> tab_res;
	std::vector<urmapx_pair_info> tab_info;
	std::vector<uint32_t> tab_ends, tab_lens2;
	~TextJob() {
		HostPool::get().release(in, in_cap);
		HostPool::get().release(out, out_cap);
	}
	static bool grow(char *&p, size_t &cap, size_t want) {
		if (want <= cap) return true;
		HostPool::get().release(p, cap);
		p = HostPool::get().acquire(want, cap);
		if (!p) cap = 0;
		return p != nullptr;
	}
};

// first record start at or after `from`: a line that begins with '@' whose line after next begins with '+'.  A wrong
// guess cannot pass: the chunk in front would then hold a line count that is not a multiple of four, which the device
// parser reports.  Returns 0 if the window [from, from + 4 MB) shows none.
uint64_t find_record_start(int fd, uint64_t from, uint64_t fsize) {
	std::vector<char> w;
	for (size_t win = 1u << 16; win <= (8u << 20); win *= 4) {
		const size_t len = (size_t)std::min<uint64_t>(win, fsize - from);
		w.resize(len);
		size_t got = 0;
		while (got < len) {
			ssize_t k = pread(fd, w.data() + got, len - got, (off_t)(from + got));
			if (k <= 0) return 0;
			got += (size_t)k;
		}
		const char *b = w.data(), *e = b + len;
		for (const char *nl = (const char *)memchr(b, '\n', len); nl && nl + 1 < e; nl = (const char *)memchr(nl + 1, '\n', (size_t)(e - nl - 1))) {
			if (nl[1] != '@') continue;
			const char *n1 = (const char *)memchr(nl + 1, '\n', (size_t)(e - nl - 1));
			const char *n2 = n1 && n1 + 1 < e ? (const char *)memchr(n1 + 1, '\n', (size_t)(e - n1 - 1)) : nullptr;
			if (!n2 || n2 + 1 >= e) break;  // the window ends inside this record: look again with a larger one
			if (n2[1] == '+') return from + (uint64_t)(nl + 1 - b);
		}
		if (len < win) return 0;  // the window reached the end of the file
	}
	return 0;
}


// A FASTQ input read front to back as uncompressed bytes: a plain file, or a gzip file (linereader.cpp:14-113 reads .gz
// through zlib) inflated on the fly.  A plain gzip stream is cut into segments that the caller's OpenMP threads inflate side by
// side (pgzip.h, round 5; one zlib stream on one thread until then); a BGZF file (gzip members of <= 64 KB that carry their
// compressed size in a 'BC' extra field, the bgzip / htslib convention) is inflated a batch of blocks at a time by the same threads.
class SeqSource {
public:
	~SeqSource() {
		for (z_stream &z : bz_) inflateEnd(&z);
		if (fd_ >= 0 && own_) ::close(fd_);
	}
	// a descriptor that cannot seek (FIFO, standard input), owned by the caller: plain FASTQ text read front to back
	bool open_pipe(int fd) {
		fd_ = fd; own_ = false; pipe_ = true; gz_ = false;
		return fd >= 0;
	}
	bool is_pipe() const { return pipe_; }
	// regular files only; gz says which kind the name promises (".gz")
	bool open(const char *path, bool gz) {
		fd_ = ::open(path, O_RDONLY);
		struct stat st;
		if (fd_ < 0 || fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) return false;
		csize_ = (uint64_t)st.st_size;
		gz_ = gz;
		if (!gz_) return true;
		unsigned char h[18];
		if (pread(fd_, h, sizeof h, 0) != (ssize_t)sizeof h || h[0] != 0x1f || h[1] != 0x8b) return false;
		bgzf_ = (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
		if (bgzf_) cbuf_.resize(32u << 20);
		else if (!pg_.open(fd_, csize_)) return false;  // plain gzip: segments of the one stream inflated side by side (pgzip.h)
		return true;
	}
	bool is_gz() const { return gz_; }
	bool returns_whole_rounds() const { return gz_ && !bgzf_ && !pipe_; }  // a plain gzip stream through ParallelGunzip
	bool is_bgzf() const { return bgzf_; }
	bool failed() const { return bad_; }
	uint64_t compressed_size() const { return csize_; }
	// up to cap bytes of the uncompressed stream into dst; 0 = end of input (or failed())
	size_t read(char *dst, size_t cap, int threads) {
		if (cap == 0 || eof_ || bad_) return 0;
		if (pipe_) {
			size_t done = 0;
			while (done < cap) {
				const ssize_t k = ::read(fd_, dst + done, cap - done);
				if (k < 0) { if (errno == EINTR) continue; bad_ = true; break; }
				if (k == 0) { eof_ = true; break; }
				done += (size_t)k;
			}
			return done;
		}
		if (!gz_) {
			const size_t n = (size_t)std::min<uint64_t>(cap, csize_ - cpos_);
			if (n == 0) { eof_ = true; return 0; }
			std::atomic<bool> ok{true};
			if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static, 1) num_threads(threads)
			for (int t = 0; t < threads; ++t) {
				const size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
				size_t done = lo;
				while (done < hi) {
					const ssize_t k = pread(fd_, dst + done, hi - done, (off_t)(cpos_ + done));
					if (k <= 0) { ok.store(false); break; }
					done += (size_t)k;
				}
			}
			if (!ok) { bad_ = true; return 0; }
			cpos_ += n;
			return n;
		}
		if (bgzf_) return read_bgzf(dst, cap, threads);
		const size_t k = pg_.read(dst, cap, gz_threads_ > 0 ? gz_threads_ : threads);
		if (pg_.failed()) bad_ = true;
		if (k == 0) eof_ = true;
		return k;
	}
	// plain gzip: the inflater is what the run waits for -- it may use more threads than the readers of a plain file get
	void set_gz_threads(int t) { gz_threads_ = t; }
	uint64_t gz_parallel_bytes() const { return pg_.parallel_bytes(); }

private:
	bool refill() {  // more compressed bytes behind what cbuf_ still holds
		if (chave_ > cbeg_ && cbeg_ > 0) memmove(cbuf_.data(), cbuf_.data() + cbeg_, chave_ - cbeg_);
		chave_ -= cbeg_; cbeg_ = 0;
		bool got = false;
		while (chave_ < cbuf_.size() && cpos_ < csize_) {
			const ssize_t k = pread(fd_, cbuf_.data() + chave_, cbuf_.size() - chave_, (off_t)cpos_);
			if (k <= 0) { bad_ = true; return false; }
			chave_ += (size_t)k; cpos_ += (uint64_t)k;
			got = true;
		}
		return got;
	}
	size_t read_bgzf(char *dst, size_t cap, int threads) {
		struct Blk { size_t in, in_len, out, out_len; uint32_t crc; };
		std::vector<Blk> blks;
		size_t out = 0;
		for (;;) {
			if (chave_ - cbeg_ < 18 + 8 && !refill() && chave_ == cbeg_) { eof_ = true; break; }
			if (chave_ - cbeg_ < 18 + 8) { bad_ = true; break; }
			const uint8_t *h = cbuf_.data() + cbeg_;
			if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') { bad_ = true; break; }
			const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;
			if (bsize < 26) { bad_ = true; break; }
			if (chave_ - cbeg_ < bsize) {
				if (!blks.empty()) break;  // inflate what is complete first
				if (!refill() || chave_ - cbeg_ < bsize) { bad_ = true; break; }
				continue;
			}
			const uint8_t *tl = cbuf_.data() + cbeg_ + bsize - 4;
			const size_t isize = (size_t)tl[0] | ((size_t)tl[1] << 8) | ((size_t)tl[2] << 16) | ((size_t)tl[3] << 24);
			if (isize > 65536) { bad_ = true; break; }
			if (out + isize > cap) break;
			const uint32_t crc = (uint32_t)tl[-4] | ((uint32_t)tl[-3] << 8) | ((uint32_t)tl[-2] << 16) | ((uint32_t)tl[-1] << 24);
			blks.push_back(Blk{cbeg_ + 18, bsize - 26, out, isize, crc});
			out += isize;
			cbeg_ += bsize;
		}
		if (bad_) return 0;
		if (threads < 1) threads = 1;
		if ((int)bz_.size() < threads) {
			const size_t old = bz_.size();
			bz_.resize((size_t)threads);
			for (size_t i = old; i < bz_.size(); ++i) {
				memset(&bz_[i], 0, sizeof(z_stream));
				if (inflateInit2(&bz_[i], -15) != Z_OK) { bad_ = true; return 0; }
			}
		}
		std::atomic<bool> ok{true};
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
		for (long i = 0; i < (long)blks.size(); ++i) {
			z_stream &z = bz_[(size_t)omp_get_thread_num()];
			const Blk &b = blks[(size_t)i];
			if (b.out_len == 0) continue;
			inflateReset(&z);
			z.next_in = cbuf_.data() + b.in; z.avail_in = (uInt)b.in_len;
			z.next_out = (Bytef *)dst + b.out; z.avail_out = (uInt)b.out_len;
			if (inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0) ok.store(false);
			// the member's CRC-32, as zlib's gzread checks it for the reference (a raw inflate does not): round 5
			else if (crc32_fast((const uint8_t *)dst + b.out, b.out_len) != b.crc) ok.store(false);
		}
		if (!ok) { bad_ = true; return 0; }
		return out;
	}
	int fd_ = -1;
	bool gz_ = false, bgzf_ = false, eof_ = false, bad_ = false, pipe_ = false, own_ = true;
	uint64_t csize_ = 0, cpos_ = 0;  // compressed (or plain) file: size, next byte to fetch
	std::vector<uint8_t> cbuf_;
	size_t cbeg_ = 0, chave_ = 0;
	ParallelGunzip pg_;
	int gz_threads_ = 0;
	std::vector<z_stream> bz_;
};

// start of the last record of b[0, n) whose first three lines lie inside the buffer: a line that begins with '@' whose line
// after next begins with '+' (only label and quality lines can begin with '@', and two lines after a quality line come
// bases).  0: none in sight (n holds less than one such record: the caller reads more).
size_t last_record_start(const char *b, size_t n) {
	for (size_t win = 1u << 16;; win *= 8) {
		const size_t lo = n > win ? n - win : 0;
		const char *p = b + lo, *e = b + n;
		if (lo) {  // to the first line start at or behind lo
			p = (const char *)memchr(p, '\n', (size_t)(e - p));
			if (!p) { if (lo == 0) return 0; continue; }
			++p;
		}
		size_t best = 0;
		while (p < e) {
			const char *n1 = (const char *)memchr(p, '\n', (size_t)(e - p));
			if (!n1) break;
			if (*p == '@') {
				const char *n2 = n1 + 1 < e ? (const char *)memchr(n1 + 1, '\n', (size_t)(e - n1 - 1)) : nullptr;
				if (n2 && n2 + 1 < e && n2[1] == '+') best = (size_t)(p - b);
			}
			p = n1 + 1;
		}
		if (best) return best;
		if (lo == 0) return 0;
	}
}

}  // namespace

namespace {
// One shard of a sharded run (urmapx_map_options.sam_shards): the records of bytes [lo, hi) of the (plain, seekable) input
// file(s), cut at record starts; the SAM header only in the first shard.
struct InputRange {
	// out: when this pipeline's clock started (contexts made, files open) and stopped, on the steady clock -- the wall time of a
	// sharded run is first start .. last stop, with the set-up outside as in a run of one pipeline
	mutable double t_begin = 0, t_end = 0;
	bool on = false;
	uint64_t lo[2] = {0, 0}, hi[2] = {0, 0};
	uint64_t lines_before = 0;  // lines of each file in front of lo (they name the line in the reader's messages)
	// single-end shards: the lines in front of lo are counted only if a message needs a line number (lines_before stays 0 until then)
	std::function<uint64_t()> lazy_lines;
	bool header = true;
};

// number of '\n' in bytes [from, to) of fd, by all threads
// per_piece (optional): the count of every 8 MB piece, for line_offsets below
uint64_t count_newlines(int fd, uint64_t from, uint64_t to, int threads, std::vector<uint64_t> *per_piece = nullptr) {
	if (to <= from) return 0;
	const uint64_t piece = 8u << 20;
	const uint64_t np = (to - from + piece - 1) / piece;
	if (per_piece) per_piece->assign((size_t)np, 0);
	uint64_t total = 0;
#pragma omp parallel num_threads(threads) reduction(+ : total)
	{
	std::vector<char> buf((size_t)std::min<uint64_t>(piece, to - from));  // one buffer per thread, not per piece
#pragma omp for schedule(dynamic, 1)
	for (int64_t k = 0; k < (int64_t)np; ++k) {
		const uint64_t a = from + (uint64_t)k * piece, b = std::min(to, a + piece);
		size_t have = 0;
		while (a + have < b) {
			const ssize_t r = pread(fd, buf.data() + have, (size_t)(b - a - have), (off_t)(a + have));
			if (r <= 0) break;
			have += (size_t)r;
		}
		const char *c = buf.data(), *e = c + have;
		uint64_t n = 0;
		while (c < e) {
			const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
			if (!nl) break;
			++n;
			c = nl + 1;
		}
		total += n;
		if (per_piece) (*per_piece)[(size_t)k] = n;
	}
	}
	return total;
}

// offset of the byte behind the `lines`-th '\n' of fd at or after `from` (`fsize` if the file has fewer)
uint64_t skip_lines(int fd, uint64_t from, uint64_t fsize, uint64_t lines) {
	std::vector<char> buf(8u << 20);
	uint64_t at = from;
	while (lines && at < fsize) {
		const ssize_t r = pread(fd, buf.data(), (size_t)std::min<uint64_t>(buf.size(), fsize - at), (off_t)at);
		if (r <= 0) return fsize;
		const char *c = buf.data(), *e = c + r;
		while (lines && c < e) {
			const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
			if (!nl) { c = e; break; }
			--lines;
			c = nl + 1;
		}
		at += (uint64_t)(c - buf.data());
	}
	return lines ? fsize : at;
}

// offsets of the bytes behind the targets[i]-th '\n' of fd (`fsize` where the file has fewer), targets ascending: the pieces' line
// counts by all threads, then one scan inside the piece each target falls in (the mates' file of a sharded paired run used to be
// walked by one thread from its start to the last cut)
std::vector<uint64_t> line_offsets(int fd, uint64_t fsize, const std::vector<uint64_t> &targets, int threads) {
	std::vector<uint64_t> per, out(targets.size(), fsize);
	count_newlines(fd, 0, fsize, threads, &per);
	const uint64_t piece = 8u << 20;
	uint64_t before = 0;
	size_t k = 0;
	for (size_t i = 0; i < targets.size(); ++i) {
		if (targets[i] == 0) { out[i] = 0; continue; }
		while (k < per.size() && before + per[k] < targets[i]) before += per[k++];
		if (k == per.size()) break;  // fewer lines than asked for: fsize
		out[i] = skip_lines(fd, (uint64_t)k * piece, fsize, targets[i] - before);
	}
	return out;
}

// Where a device hangs: its PCI function's NUMA node (/sys/bus/pci/devices/<bus id>/numa_node) and that node's CPUs that this process
// may use.  On an 8-GPU node the devices sit behind two sockets: a lane thread that drives device g -- it fills and drains the
// page-locked chunk buffers that cross PCIe -- runs on g's socket, and so do the reader and the writer of a shard whose devices are
// all there.  ok == false (no such file, node -1 as on one-socket boxes and in most VMs, or none of the node's CPUs
// allowed): nothing is pinned.
struct DevicePlace {
	int node = -1;
	cpu_set_t cpus;
	bool ok = false;
};
DevicePlace place_of_device(int device) {
	DevicePlace P;
	CPU_ZERO(&P.cpus);
	char bus[64] = {0};
	if (getenv("URMAPX_NO_NUMA_PIN") || hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return P;
	for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
	const char *forced = getenv("URMAPX_TEST_NUMA_NODE");  // test aid: pretend every device hangs off this node
	if (forced) P.node = atoi(forced);
	else {
		FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r");
		if (!f) return P;
		if (fscanf(f, "%d", &P.node) != 1) P.node = -1;
		fclose(f);
	}
	if (P.node < 0) return P;
	FILE *f = fopen(("/sys/devices/system/node/node" + std::to_string(P.node) + "/cpulist").c_str(), "r");
	if (!f) return P;
	char list[4096] = {0};
	const bool got = fgets(list, sizeof list, f) != nullptr;
	fclose(f);
	if (!got) return P;
	cpu_set_t allowed;
	CPU_ZERO(&allowed);
	if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return P;
	for (const char *c = list; *c && *c != '\n';) {  // "0-63,128-191"
		char *e;
		const long a = strtol(c, &e, 10);
		long b = a;
		if (e == c) break;
		if (*e == '-') { c = e + 1; b = strtol(c, &e, 10); }
		for (long k = a; k <= b && k < CPU_SETSIZE; ++k)
			if (CPU_ISSET((int)k, &allowed)) CPU_SET((int)k, &P.cpus);
		c = *e == ',' ? e + 1 : e;
	}
	P.ok = CPU_COUNT(&P.cpus) > 0;
	return P;
}
// Idle OpenMP workers must sleep, not spin: the reader, the writer, every lane and the caller each have a team of their own, and libomp's
// default keeps a team's workers spinning for 200 ms behind every parallel region (KMP_BLOCKTIME).  Three or four teams of 16 spinning
// threads on a host that grants the process 16 CPUs (a cgroup quota, as on the GPU boxes) get the whole process throttled: the lanes' host
// threads stall between their launches and a run takes three times as long -- seen in half of the bench runs of round 5 (the command line
// has always set OMP_WAIT_POLICY=passive for this; a library call cannot count on its caller's environment).  Per thread: it applies to the
// teams this thread starts.
// kmp_set_blocktime is libomp's (the runtime hipcc links); under another OpenMP runtime -- the sanitizer builds use g++ -- idle workers
// are the caller's business (OMP_WAIT_POLICY=passive, which the command line and bench.py set anyway).
#ifdef KMP_VERSION_MAJOR
inline void omp_workers_sleep_when_idle() { kmp_set_blocktime(0); }
inline int omp_idle_blocktime() { return kmp_get_blocktime(); }
inline void omp_set_idle_blocktime(int ms) { kmp_set_blocktime(ms); }
#else
inline void omp_workers_sleep_when_idle() {}
inline int omp_idle_blocktime() { return 0; }
inline void omp_set_idle_blocktime(int) {}
#endif

// the calling thread (and the OpenMP team it starts later) onto a device's socket; false: left where it was
bool pin_to(const DevicePlace &P) { return P.ok && sched_setaffinity(0, sizeof P.cpus, &P.cpus) == 0; }
// A pipeline thread that pinned itself hands its OpenMP workers back unpinned: the workers of its team inherited its
// narrowed mask when they were made, libomp returns them to a pool the CALLER's later parallel regions draw from, and those would
// then run on one socket.  Run by the pinned thread as its last act: every thread of its team takes the mask the caller had.
void unpin_team(bool was_pinned, const cpu_set_t &caller_mask, int team) {
	if (!was_pinned) return;
#pragma omp parallel num_threads(team > 0 ? team : 1)
	(void)sched_setaffinity(0, sizeof caller_mask, &caller_mask);
}

int map_files_impl(urmapx_index *I, const urmapx_map_options *opt, const InputRange &range, const char *fastq1, const char *fastq2,
                   const char *samout, const char *tabout, urmapx_map_report *report, char *err, size_t errcap) {
	auto say = [&](const std::string &s) {
		if (err && errcap) snprintf(err, errcap, "%s", s.c_str());
	};
	if (err && errcap) err[0] = 0;
	if (!I || !opt || !fastq1) return URMAPX_E_ARG;
	const bool paired = fastq2 != nullptr;
	const int gpus = opt->gpus > 0 ? opt->gpus : 1, streams = opt->streams > 0 ? opt->streams : 2;
	if (gpus > 64 || streams > 8) { say("gpus must be 1..64, streams 1..8"); return URMAPX_E_ARG; }
	const uint32_t batch = opt->batch ? opt->batch : (1u << 18);
	const unsigned minq = paired ? opt->minq : 10;  // only cmd_map2 reads -minq (map2.cpp:76); -map keeps State1::m_Minq = 10
	// URMAPX_FORCE_DEVICE=d (test aid): every lane runs on physical device d, so that the N-device code path can be
	// exercised on a machine with one GPU
	const char *forced = getenv("URMAPX_FORCE_DEVICE");
	auto phys = [&](int g) { return forced ? atoi(forced) : opt->first_gpu + g; };
	urmapx_params P;
	int rc = urmapx_params_for_method((opt->veryfast && !paired) ? 7 : 6, &P);  // -map2 always uses method 6 (map2.cpp:15-16)
	if (rc) return rc;
	// one replica of the index per device (uploaded concurrently), K mapping contexts on each
	const int n_lanes = gpus * streams;
	std::vector<urmapx_index *> replicas((size_t)gpus, nullptr);
	std::vector<urmapx_ctx *> ctxs((size_t)n_lanes, nullptr);
	std::vector<urmapx_text *> texts((size_t)n_lanes, nullptr);  // a lane's text stage: from the pool, or made by the lane's thread
	const bool pool_lanes = LanePool::enabled();
	const bool lane_veryfast = paired && opt->veryfast, lane_pair_info = paired && tabout != nullptr;
	bool run_ok = false;  // set at the end of a run without failure: only then are the lanes worth keeping
	auto release = [&]() {
		for (int l = 0; l < n_lanes; ++l) {
			urmapx_ctx *C = ctxs[(size_t)l];
			urmapx_text *T = texts[(size_t)l];
			// (a lane of the caller's own index only: the other devices' replicas are closed below.  set_deferred refuses while a text is on its way)
			if (C && pool_lanes && run_ok && l % gpus == 0 && replicas[0] == I && (!T || urmapx_text_set_deferred(T, 0) == URMAPX_OK))
				LanePool::get().give(PooledLane{I, phys(0), P, lane_veryfast, lane_pair_info, C, T, env_knobs()});
			else LanePool::destroy(PooledLane{I, 0, P, false, false, C, T, 0});
			ctxs[(size_t)l] = nullptr; texts[(size_t)l] = nullptr;
		}
		for (int g = 1; g < gpus; ++g) urmapx_index_close(replicas[(size_t)g]);
	};
	{
		std::vector<int> rcs((size_t)gpus, 0);
		std::vector<std::thread> up;
		for (int g = 0; g < gpus; ++g)
			up.emplace_back([&, g] {
				if (g == 0) { rcs[0] = urmapx_index_upload(I, phys(0)); replicas[0] = I; }
				else rcs[(size_t)g] = urmapx_index_replicate(I, phys(g), &replicas[(size_t)g]);
			});
		for (auto &t : up) t.join();
		for (int g = 0; g < gpus; ++g)
			if (rcs[(size_t)g]) { say(std::string("Uploading index to the GPU: ") + urmapx_strerror(rcs[(size_t)g])); release(); return rcs[(size_t)g]; }
	}
	for (int l = 0; l < n_lanes; ++l) {
		if (pool_lanes && l % gpus == 0 && LanePool::get().take(I, phys(0), P, lane_veryfast, lane_pair_info, ctxs[(size_t)l], texts[(size_t)l])) continue;
		rc = urmapx_ctx_create(replicas[(size_t)(l % gpus)], phys(l % gpus), &P, &ctxs[(size_t)l]);
		if (!rc && paired && opt->veryfast) rc = urmapx_ctx_set_pe_veryfast(ctxs[(size_t)l], 1);
		if (!rc && paired && tabout) rc = urmapx_ctx_set_pair_info(ctxs[(size_t)l], 1);
		if (rc) { say(std::string("Creating mapping context: ") + urmapx_strerror(rc)); release(); return rc; }
	}
	// where each device hangs (NUMA node of its PCI function): lane threads run there; a pipeline whose devices share one node -- every
	// shard of a sharded run on a two-socket node -- has its reader and writer there too
	std::vector<DevicePlace> places((size_t)gpus);
	for (int g = 0; g < gpus; ++g) places[(size_t)g] = place_of_device(phys(g));
	bool one_node = places[0].ok;
	for (int g = 1; g < gpus; ++g) one_node = one_node && places[(size_t)g].ok && places[(size_t)g].node == places[0].node;
	std::string placement;
	for (int g = 0; g < gpus; ++g)
		placement += (g ? " gpu" : "gpu") + std::to_string(phys(g)) + (places[(size_t)g].ok ? "@node" + std::to_string(places[(size_t)g].node) : std::string("@any"));
	placement += one_node ? "; reader+writer@node" + std::to_string(places[0].node) : std::string("; reader+writer@any");
	// host threads for FASTQ parsing and SAM formatting (the mapping itself runs on the GPU)
	const int host_threads = opt->host_threads > 0 ? opt->host_threads : std::min(16, std::max(1, (int)std::thread::hardware_concurrency()));
	const int omp_threads_before = omp_get_max_threads();  // this is a library call: the caller's OpenMP setting comes back at the end
	// (the calling thread's block time comes back too -- kmp_set_blocktime(0) below applies to the teams THIS thread starts later)
	struct OmpRestore { int n, blocktime; ~OmpRestore() { omp_set_num_threads(n); omp_set_idle_blocktime(blocktime); } } omp_restore{omp_threads_before, omp_idle_blocktime()};
	omp_workers_sleep_when_idle();
	omp_set_num_threads(host_threads);
	cpu_set_t caller_mask;
	CPU_ZERO(&caller_mask);
	(void)sched_getaffinity(0, sizeof caller_mask, &caller_mask);
	FileSink sink;
	int write_threads_used = 1;
	bool text_on_device = false;
	uint64_t input_bytes = 0;  // uncompressed FASTQ bytes the text phase handed to the device
	std::string medium_name = "file";
	const bool have_sam = samout != nullptr;
	uint64_t sam_off = 0;
	Failure fail;
	if (samout) {
		if (opt->discard_sam) sink.open_discard();
		else if (!sink.open(samout)) { say(std::string("Cannot create ") + samout); release(); return URMAPX_E_IO; }
		if (range.header) {
			std::string hdr;
			append_sam_header_text(hdr, I, opt->cmdline);
			if (!sink.write_at(hdr.data(), hdr.size(), 0, 1)) { say(std::string("Cannot write ") + samout); release(); return URMAPX_E_IO; }
			sam_off = hdr.size();
		}
		medium_name = sink.medium();
	}
	// -tabbedout (outfiles.cpp:7-12): State2::OutputTab2's line per pair; only -map2 writes it
	FILE *ftab = nullptr;
	if (tabout) {
		ftab = fopen(tabout, "wb");
		if (!ftab) { say(std::string("Cannot create ") + tabout); release(); return URMAPX_E_IO; }
	}
	FastqReader rd, rd2;
	{
		std::string e;
		if (!rd.open(fastq1, e) || (paired && !rd2.open(fastq2, e))) {
			say(e);
			if (ftab) fclose(ftab);
			release();
			return URMAPX_E_IO;
		}
	}
	if (range.on) {  // the host reader, should it be needed, starts and stops where the shard does
		if (!rd.resume_at(range.lo[0], range.lines_before) || (paired && !rd2.resume_at(range.lo[1], range.lines_before))) {
			say(std::string("Cannot read a part of ") + fastq1);
			if (ftab) fclose(ftab);
			release();
			return URMAPX_E_IO;
		}
		rd.set_limit(range.hi[0]);
		if (paired) rd2.set_limit(range.hi[1]);
		if (range.lazy_lines) rd.set_lazy_line_base(range.lazy_lines);
	}
	const auto t1 = std::chrono::steady_clock::now();
	Trace trace;
	unsigned long long n_reads = 0, n_accept = 0, n_reject = 0, n_nohit = 0, n_unsupported = 0;
	double t_parse = 0, t_gpu = 0, t_format = 0, t_write = 0;  // busy seconds per stage
	double dev_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // a lane's stream time by events: copy in, parse, map, SAM text, copy out (text phase)
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
	std::mutex gpu_time_lock;

	// ---- text phase: reads from plain files, FASTQ bytes -> device -> SAM bytes ----
	bool host_phase = true;
	{
		auto open_plain = [&](const char *path, uint64_t &size) {  // a regular file that is not .gz, else -1
			const size_t l = strlen(path);
			if ((l > 3 && !strcmp(path + l - 3, ".gz")) || !strcmp(path, "-")) return -1;
			const int fd = open(path, O_RDONLY);
			struct stat st;
			if (fd >= 0 && (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0)) { close(fd); return -1; }
			if (fd >= 0) size = (uint64_t)st.st_size;
			return fd;
		};
		uint64_t fsize = 0, fsize2 = 0;
		int fq = -1, fq2 = -1;
		// .gz input: the same phase with the chunks cut out of the inflated stream (SeqSource).  A pipe (FIFO, standard input)
		// is streamed the same way; it has no way back to a chunk the device parser hands back, so the bytes of that chunk
		// and of everything read behind it are kept and given to the host reader in front of the rest of the pipe.
		auto is_gz_name = [](const char *path) { const size_t l = strlen(path); return l > 3 && !strcmp(path + l - 3, ".gz"); };
		SeqSource src1, src2;
		bool streamed = false;
		// -tabbedout (pairs only): the SAM text is made on the device all the same; the lines of the tab file are formatted by
		// host threads from the chunk's results and pair records (urmapx_text_fetch_pairs), with the labels read in place
		if (have_sam && (!ftab || paired) && !getenv("URMAPX_HOST_TEXT")) {
			auto open_src = [&](SeqSource &src, FastqReader &r, const char *path) {
				if (r.is_pipe()) return src.open_pipe(r.fd());
				return strcmp(path, "-") != 0 && src.open(path, is_gz_name(path));
			};
			if (is_gz_name(fastq1) || (paired && is_gz_name(fastq2)) || rd.is_pipe() || (paired && rd2.is_pipe())) {
				streamed = open_src(src1, rd, fastq1) && (!paired || open_src(src2, rd2, fastq2));
				src1.set_gz_threads(host_threads);  // a plain gzip stream: the inflater is what the run waits for
				src2.set_gz_threads(host_threads);
			} else {
				fq = open_plain(fastq1, fsize);
				if (fq >= 0 && paired) {
					fq2 = open_plain(fastq2, fsize2);
					if (fq2 < 0) { close(fq); fq = -1; }
				}
				if (fq >= 0 && range.on) { fsize = range.hi[0]; fsize2 = range.hi[1]; }  // a shard: its part of the file(s) is the file
			}
		}
		if (fq >= 0 || streamed) {
			size_t chunk_bytes = (size_t)std::min(330.0 * (double)(paired ? std::max(1u, batch / 2) : batch), 536870912.0);  // streamed: a 150-base record
			if (streamed && !opt->batch && src1.is_gz()) {
				// no -batch given: larger chunks for a large file, as for plain input below (FASTQ text is 4-5 times its gzip size)
				const double est = 4.0 * (double)src1.compressed_size() / 330.0 * (paired ? 2.0 : 1.0);
				const double reads = std::min(524288.0, std::max((double)batch, est / (4.0 * (double)n_lanes)));
				chunk_bytes = (size_t)std::min(330.0 * (paired ? std::max(1.0, reads / 2) : reads), 536870912.0);
			}
			// `batch` reads (pairs: batch / 2 of each file) at the record size the head of the file shows
			if (!streamed) {
				std::vector<char> head(1u << 16);
				const ssize_t k = pread(fq, head.data(), head.size(), 0);
				size_t nl = 0, last = 0;
				for (ssize_t i = 0; i < k; ++i)
					if (head[(size_t)i] == '\n' && (++nl & 3) == 0) last = (size_t)i + 1;
				const double rec = nl >= 4 ? (double)last / (double)(nl / 4) : 512.0;
				double reads_per_chunk = (double)batch;
				if (!opt->batch) {
					// no -batch given: a chunk pays 3-4 ms of fixed cost on the device (launch tails, the second pass's empty launches) --
					// two thirds of a 262 144-read chunk's mapping time at 46 M reads/s -- so a large file is cut into larger chunks, up to
					// 524 288 reads, as long as every lane still gets four of them (the lanes overlap each other's copies).  Measured on
					// 10 M reads (profiles/r5/e2e_by_chunk.txt): 262 144 -> 29.2 M reads/s with the SAM text dropped, 524 288 -> 31.2 M;
					// 1 M-read chunks need 0.7 GB of page-locked memory each, six of them outgrow the pool and the run stalls in
					// hipHostMalloc / hipHostFree (6.5 M reads/s)
					const double est = (double)(fsize - (range.on ? range.lo[0] : 0)) / rec * (paired ? 2.0 : 1.0);
					reads_per_chunk = std::min(524288.0, std::max((double)batch, est / (4.0 * (double)n_lanes)));
					if (const char *e = getenv("URMAPX_TEST_CHUNK_READS")) reads_per_chunk = std::max(4.0, atof(e));  // test aid: the library's own choice, small (the ramp below on a small file)
				}
				chunk_bytes = (size_t)std::min(std::max(rec * (paired ? std::max(1.0, reads_per_chunk / 2) : reads_per_chunk), 4096.0), 536870912.0);
			}
			using TextChannel = Channel<std::unique_ptr<TextJob>>;
			std::vector<std::unique_ptr<TextChannel>> tparsed, tmapped;
			for (int l = 0; l < n_lanes; ++l) {
				tparsed.emplace_back(new TextChannel(1));
				tmapped.emplace_back(new TextChannel(1));
			}
			const int n_jobs = 3 * n_lanes + 2;  // per lane: one waiting for it, one being mapped, one whose text is crossing PCIe; reader, writer
			TextChannel tfree((size_t)n_jobs);
			for (int k = 0; k < n_jobs; ++k) tfree.push(std::make_unique<TextJob>());
			std::atomic<bool> stop{false};
			uint64_t reader_end = 0, reader_end2 = 0;  // first byte of each file the reader did not hand to a lane
			const int read_threads = std::max(1, host_threads / 2);
			// [off, off + n) of a file into dst, by several threads
			auto read_range = [&](int fd, char *dst, uint64_t off, size_t n) {
				std::atomic<bool> ok{true};
#pragma omp parallel for schedule(static, 1) num_threads(read_threads)
				for (int t = 0; t < read_threads; ++t) {
					const size_t lo = n * (size_t)t / (size_t)read_threads, hi = n * (size_t)(t + 1) / (size_t)read_threads;
					size_t done = lo;
					while (done < hi) {
						ssize_t k = pread(fd, dst + done, hi - done, (off_t)(off + done));
						if (k <= 0) { ok.store(false); break; }
						done += (size_t)k;
					}
				}
				return ok.load();
			};
			// '\n' count of each of read_threads slices of p[0, n)
			auto count_lines = [&](const char *p, size_t n, std::vector<size_t> &per) {
				per.assign((size_t)read_threads, 0);
#pragma omp parallel for schedule(static, 1) num_threads(read_threads)
				for (int t = 0; t < read_threads; ++t) {
					const char *c = p + n * (size_t)t / (size_t)read_threads, *e = p + n * (size_t)(t + 1) / (size_t)read_threads;
					size_t k = 0;
					while (c < e) {
						const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
						if (!nl) break;
						++k;
						c = nl + 1;
					}
					per[(size_t)t] = k;
				}
			};
			// No page-locked memory for a chunk (memlock limit, little host memory), or none on the device for the text kernels:
			// the text phase ends there and the host reader / formatter, which need neither, continue from that chunk.
			std::atomic<bool> text_nomem{false};
			bool stream_done = false;  // streamed input: the reader handed every byte of the input(s) to a lane
			// streamed input: bytes read from each source and not handed to a lane yet (behind the last cut; a whole buffer the
			// reader gave up on).  Pipes: they go to the host reader in front of the rest of the pipe, behind pipe_back.
			std::vector<char> carry, carry2, pipe_back, pipe_back2;
			std::thread treader([&] {
				const bool pinned = one_node && pin_to(places[0]);  // the page-locked chunk buffers are first touched here
				struct Unpin { bool on; const cpu_set_t &m; int n; ~Unpin() { unpin_team(on, m, n); } } unpin{pinned, caller_mask, read_threads};
				omp_workers_sleep_when_idle();
				omp_set_num_threads(read_threads);
				uint64_t off = range.on ? range.lo[0] : 0, off2 = range.on ? range.lo[1] : 0;
				double bytes2_per_byte1 = 1.0;
				std::vector<size_t> per;
				if (streamed) {
					// chunks are cut out of the inflated stream: fill the job's buffer (what the last cut left over first),
					// cut it at its last record start, keep the rest for the next chunk.  Offsets are those of the
					// uncompressed text: that is where the host reader (gzseek) takes over if a chunk is handed back.
					bool eof1 = false, eof2 = false;
					auto regrow = [&](TextJob &j, size_t want, size_t keep) {  // a larger page-locked buffer holding the first `keep` bytes
						size_t ncap = 0;
						char *nb = HostPool::get().acquire(want, ncap);
						if (!nb) return false;
						if (keep) memcpy(nb, j.in, keep);
						HostPool::get().release(j.in, j.in_cap);
						j.in = nb; j.in_cap = ncap;
						return true;
					};
					for (size_t b = 0; !stop.load() && !fail.set.load() && !text_nomem.load(); ++b) {
						if (eof1 && carry.empty()) { stream_done = !paired || (eof2 && carry2.empty()); break; }
						std::unique_ptr<TextJob> j;
						if (!tfree.pop(j)) break;
						if (stop.load()) { tfree.push(std::move(j)); break; }
						const auto tp0 = now();
						const double ta = trace.ms();
						(void)hipSetDevice(phys(0));
						size_t area1 = chunk_bytes + chunk_bytes / 8 + carry.size() + (1u << 20);  // room for this file's chunk
						size_t area2 = paired ? (size_t)((double)area1 * bytes2_per_byte1 * 1.25) + carry2.size() + (4u << 20) : 0;
						if (!TextJob::grow(j->in, j->in_cap, area1 + 4096 + area2)) { text_nomem.store(true); break; }
						size_t n = carry.size(), cut = 0;
						if (n) memcpy(j->in, carry.data(), n);
						carry.clear();
						size_t target = std::max(chunk_bytes, n + 1);
						for (;;) {
							while (n < target && !eof1) {
								const size_t k = src1.read(j->in + n, std::min(area1, target + (64u << 10)) - n, read_threads);
								if (k == 0) eof1 = true;
								n += k;
								// a gzip stream's reader hands out whole rounds written straight into this buffer and returns short when the next round
								// would not fit (pgzip.cpp): a chunk that is three quarters full is a chunk -- asking for the rest would cost a copy of a whole round
								if (src1.returns_whole_rounds() && n >= target - target / 4) break;
							}
							if (src1.failed()) break;
							cut = eof1 ? n : last_record_start(j->in, n);
							if (cut || eof1) break;
							target *= 2;  // less than one record start in the buffer: records of a size beyond reason, or not FASTQ
							if (target > (1u << 30)) break;
							if (target + (64u << 10) > area1) {
								area1 = target + target / 8 + (1u << 20);
								if (!regrow(*j, area1 + 4096 + area2, n)) { text_nomem.store(true); break; }
							}
						}
						if (fail.set.load() || text_nomem.load()) { carry.assign(j->in, j->in + n); break; }
						if (src1.failed()) { fail.raise(URMAPX_E_IO, std::string(src1.is_pipe() ? "Error reading " : "Error reading gzip file ") + fastq1); break; }
						if (cut == 0) { carry.assign(j->in, j->in + n); tfree.push(std::move(j)); break; }  // the host reader says what is wrong with it
						carry.assign(j->in + cut, j->in + n);
						j->nbytes = cut; j->file_off = off; j->nbytes2 = 0; j->file_off2 = off2; j->in2 = nullptr;
						if (paired) {
							count_lines(j->in, cut, per);
							size_t lines = 0;
							for (size_t k : per) lines += k;
							const bool last = eof1 && carry.empty();  // last chunk: whatever the mate file still holds
							size_t n_pad = (area1 + 4095) & ~(size_t)4095;
							size_t have = carry2.size(), n2 = 0, seen = 0, scanned = 0;
							bool found = false, give_up = false;
							if (have > area2) {
								area2 = have + have / 4 + (4u << 20);
								if (!regrow(*j, n_pad + area2, cut)) { text_nomem.store(true); carry.insert(carry.begin(), j->in, j->in + cut); break; }
							}
							char *dst = j->in + n_pad;
							if (have) memcpy(dst, carry2.data(), have);
							carry2.clear();
							for (;;) {
								// newlines of dst[scanned, have): the chunk ends behind line number `lines`
								const char *c = dst + scanned, *e = dst + have;
								while (c < e && (last || seen < lines)) {
									const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
									if (!nl) break;
									++seen;
									c = nl + 1;
									if (!last && seen == lines) { n2 = (size_t)(c - dst); found = true; }
								}
								scanned = found ? have : (size_t)(c - dst);
								if (found) break;
								if (eof2) { if (last) { n2 = have; found = true; } else give_up = true; break; }  // the mate file ends first: the host reader words that
								if (have == area2) {
									if (area2 > 8 * area1 + (64u << 20)) { give_up = true; break; }
									const size_t na = 2 * area2;
									size_t ncap = 0;
									char *nb = HostPool::get().acquire(n_pad + na, ncap);
									if (!nb) { text_nomem.store(true); give_up = true; break; }
									memcpy(nb, j->in, cut);
									memcpy(nb + n_pad, dst, have);
									HostPool::get().release(j->in, j->in_cap);
									j->in = nb; j->in_cap = ncap; area2 = na; dst = nb + n_pad;
								}
								const size_t want = std::min(area2 - have, std::max<size_t>(1u << 20, (size_t)((double)cut * bytes2_per_byte1) + (64u << 10) - std::min(have, (size_t)((double)cut * bytes2_per_byte1))));
								const size_t k = src2.read(dst + have, want, read_threads);
								if (k == 0) eof2 = true;
								have += k;
								if (src2.failed()) { fail.raise(URMAPX_E_IO, std::string(src2.is_pipe() ? "Error reading " : "Error reading gzip file ") + fastq2); give_up = true; break; }
							}
							if (give_up || fail.set.load() || text_nomem.load()) {  // this chunk stays with the reader: both files' bytes back into the carries
								carry.insert(carry.begin(), j->in, j->in + cut);
								carry2.assign(dst, dst + have);
								tfree.push(std::move(j));
								break;
							}
							carry2.assign(dst + n2, dst + have);
							j->in2 = dst; j->nbytes2 = n2;
							if (cut && n2) bytes2_per_byte1 = (double)n2 / (double)cut;
						}
						t_parse += secs(tp0, now());
						trace.add("inflate", -1, b, ta);
						const size_t adv = j->nbytes, adv2 = j->nbytes2;
						tparsed[b % (size_t)n_lanes]->push(std::move(j));
						off += adv;
						off2 += adv2;
					}
					reader_end = off; reader_end2 = off2;
					for (auto &c : tparsed) c->close();
					return;
				}
				// Round 6: the library's own chunk size (no -batch given) is ramped.  The first chunk of a lane is a quarter of the size, its second
				// a half: the lanes start after 40 MB have been read instead of 165 MB, and the second lane a quarter-chunk later instead of a whole
				// one.  The last round of chunks is cut in halves again and again (down to an eighth), so that the run does not end with one lane
				// mapping a whole chunk while the others have nothing left (10 M reads: 19 chunks of 13 ms, of which the first and the last ran
				// beside an idle lane).  URMAPX_NO_CHUNK_RAMP=1: every chunk the same size (measurement).
				const bool ramp = !opt->batch && !getenv("URMAPX_NO_CHUNK_RAMP");
				for (size_t b = 0; off < fsize && !stop.load() && !fail.set.load() && !text_nomem.load(); ++b) {
					uint64_t end = fsize;
					size_t this_chunk = chunk_bytes;
					if (ramp) {
						if (b < (size_t)n_lanes) this_chunk = chunk_bytes / 4;
						else if (b < 2 * (size_t)n_lanes) this_chunk = chunk_bytes / 2;
						const uint64_t left = fsize - off;
						if (left <= (uint64_t)n_lanes * chunk_bytes) this_chunk = (size_t)std::min<uint64_t>(this_chunk, std::max<uint64_t>(left / (2 * (uint64_t)n_lanes), chunk_bytes / 8));
						this_chunk = std::max<size_t>(this_chunk, 1024);
					}
					if (off + this_chunk < fsize) {
						end = find_record_start(fq, off + this_chunk, fsize);
						if (end == 0) break;  // no record start in sight: the host reader takes it from here
					}
					std::unique_ptr<TextJob> j;
					if (!tfree.pop(j)) break;
					if (stop.load()) break;
					const size_t n = (size_t)(end - off);
					const auto tp0 = now();
					const double ta = trace.ms();
					(void)hipSetDevice(phys(0));
					const size_t n_pad = (n + 4095) & ~(size_t)4095;  // the mate file's chunk starts here
					size_t want = n + n / 16 + 4096;
					if (paired) want = n_pad + (size_t)((double)n * bytes2_per_byte1 * 1.25) + (4u << 20);
					if (!TextJob::grow(j->in, j->in_cap, want)) { text_nomem.store(true); break; }
					if (!read_range(fq, j->in, off, n)) { fail.raise(URMAPX_E_IO, std::string("Error reading ") + fastq1); break; }
					j->nbytes = n; j->file_off = off; j->nbytes2 = 0; j->file_off2 = off2; j->in2 = nullptr;
					if (paired) {
						// the mate file's chunk: as many lines as this one holds
						count_lines(j->in, n, per);
						size_t lines = 0;
						for (size_t k : per) lines += k;
						char *dst = j->in + n_pad;
						size_t room = j->in_cap - n_pad;
						const uint64_t left2 = fsize2 - off2;
						size_t have = 0, n2 = 0;
						bool found = false, give_up = false;
						// the mate chunk outgrew the buffer (its records are longer than the first file's were so far): a larger
						// one, with what has been read
						auto more_room = [&](size_t need_room) {
							size_t ncap = 0;
							char *nb = HostPool::get().acquire(n_pad + need_room + (4u << 20), ncap);
							if (!nb) return false;
							memcpy(nb, j->in, n);
							if (have) memcpy(nb + n_pad, dst, have);
							HostPool::get().release(j->in, j->in_cap);
							j->in = nb; j->in_cap = ncap;
							dst = nb + n_pad; room = ncap - n_pad;
							return true;
						};
						if (end == fsize) {  // last chunk: whatever the mate file still holds (the device parser says if it is not the same count)
							if (left2 > room && (left2 > 8ull * n + (64u << 20) || !more_room((size_t)left2))) give_up = true;
							else { have = (size_t)left2; found = true; n2 = have; if (have && !read_range(fq2, dst, off2, have)) { fail.raise(URMAPX_E_IO, std::string("Error reading ") + fastq2); break; } }
						}
						while (!found && !give_up) {
							size_t upto = have ? have + have / 4 + (1u << 20) : (size_t)((double)n * bytes2_per_byte1) + (1u << 16);
							upto = (size_t)std::min<uint64_t>(std::min<uint64_t>(upto, room), left2);
							if (upto <= have) {
								if (have < left2 && room < 8 * n + (64u << 20) && more_room(2 * room)) continue;  // out of room, not out of file
								give_up = true;  // out of file with too few lines (the host reader words that), or a chunk beyond reason
								break;
							}
							if (!read_range(fq2, dst + have, off2 + have, upto - have)) { fail.raise(URMAPX_E_IO, std::string("Error reading ") + fastq2); give_up = true; break; }
							have = upto;
							count_lines(dst, have, per);
							size_t cum = 0;
							for (int t = 0; t < read_threads && !found; ++t) {
								if (cum + per[(size_t)t] >= lines && lines > 0) {  // the chunk ends in this slice: walk to its last '\n'
									const char *c = dst + have * (size_t)t / (size_t)read_threads, *e = dst + have * (size_t)(t + 1) / (size_t)read_threads;
									size_t need = lines - cum;
									while (need) {
										const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
										c = nl + 1;
										--need;
									}
									n2 = (size_t)(c - dst);
									found = true;
								}
								cum += per[(size_t)t];
							}
							if (!found && have == left2) give_up = true;  // the mate file ends first: the host reader words that
						}
						if (fail.set.load() || text_nomem.load()) break;
						if (give_up) { tfree.push(std::move(j)); break; }
						j->in2 = dst; j->nbytes2 = n2;
						if (n && n2) bytes2_per_byte1 = (double)n2 / (double)n;
					}
					t_parse += secs(tp0, now());
					trace.add("read", -1, b, ta);
					const size_t adv2 = j->nbytes2;
					tparsed[b % (size_t)n_lanes]->push(std::move(j));
					off = end;
					off2 += adv2;
				}
				reader_end = off; reader_end2 = off2;
				for (auto &c : tparsed) c->close();
			});
			bool handed_back = false;
			uint64_t resume_off = 0, resume_off2 = 0, lines_done = range.on ? range.lines_before : 0;  // lines_done: per file
			// write() calls on one file take turns (inode lock): more threads only add hand-overs
			const int write_threads = sink.writer_threads(host_threads);
			write_threads_used = write_threads;
			text_on_device = true;
			std::thread twriter([&] {
				const bool pinned = one_node && pin_to(places[0]);
				struct Unpin { bool on; const cpu_set_t &m; int n; ~Unpin() { unpin_team(on, m, n); } } unpin{pinned, caller_mask, host_threads};
				omp_workers_sleep_when_idle();
				omp_set_num_threads(host_threads);
				std::unique_ptr<TextJob> j;
				for (size_t b = 0; tmapped[b % (size_t)n_lanes]->pop(j); ++b) {
					if (!handed_back && !fail.set.load()) {
						if (j->rc == URMAPX_E_NOMEM) { j->rc = 0; j->rep.reason = 0xFFFEu; text_nomem.store(true); }
						if (j->rc) fail.raise(j->rc, std::string(paired ? "urmapx_text_map_pe: " : "urmapx_text_map_se: ") + urmapx_strerror(j->rc));
						else if (j->rep.reason) { handed_back = true; stop.store(true); resume_off = j->file_off; resume_off2 = j->file_off2; }
					}
					if (handed_back && streamed) {  // pipes: this chunk's bytes and those of every chunk behind it, in order
						if (src1.is_pipe()) pipe_back.insert(pipe_back.end(), j->in, j->in + j->nbytes);
						if (paired && src2.is_pipe() && j->in2) pipe_back2.insert(pipe_back2.end(), j->in2, j->in2 + j->nbytes2);
					}
					if (!handed_back && !fail.set.load()) {
						const auto tw0 = now();
						const double ta = trace.ms();
#ifdef URX_FAULT_OFF32  // fault injection for the suite's own check (profiles/r6/fault_off32.txt): the writer's file offset cut to 32 bits
						const uint64_t at = (uint32_t)sam_off;
#else
						const uint64_t at = sam_off;
#endif
						if (!sink.write_at(j->out, (size_t)j->rep.sam_bytes, at, write_threads))
							fail.raise(URMAPX_E_IO, std::string("Error writing ") + samout);
						sam_off += j->rep.sam_bytes;
						if (ftab && !j->tab.empty() && fwrite(j->tab.data(), 1, j->tab.size(), ftab) != j->tab.size())
							fail.raise(URMAPX_E_IO, std::string("Error writing ") + tabout);
						t_write += secs(tw0, now());
						trace.add("write", -1, b, ta);
						n_reads += j->rep.records; n_accept += j->rep.mapped_q; n_reject += j->rep.mapped_lowq;
						n_nohit += j->rep.unmapped; n_unsupported += j->rep.unsupported;
						dev_ms[0] += j->rep.ms_h2d; dev_ms[1] += j->rep.ms_parse; dev_ms[2] += j->rep.ms_map; dev_ms[3] += j->rep.ms_format; dev_ms[4] += j->rep.ms_d2h;
						dev_ms[5] += j->rep.ms_map_search; dev_ms[6] += j->rep.ms_map_dp; dev_ms[7] += j->rep.ms_map_enqueue;
						lines_done += (paired ? 2ull : 4ull) * j->rep.records;
						input_bytes += j->nbytes + j->nbytes2;
					}
					tfree.push(std::move(j));
				}
			});
			std::vector<std::thread> tlanes;
			for (int l = 0; l < n_lanes; ++l)
				tlanes.emplace_back([&, l] {
					omp_workers_sleep_when_idle();
					const bool pinned = pin_to(places[(size_t)(l % gpus)]);
					struct Unpin { bool on; const cpu_set_t &m; int n; ~Unpin() { unpin_team(on, m, n); } } unpin{pinned, caller_mask, std::max(1, host_threads / n_lanes)};
					(void)hipSetDevice(phys(l % gpus));
					urmapx_text *T = texts[(size_t)l];
					const double tc = trace.ms();
					const int trc = T ? URMAPX_OK : urmapx_text_create(ctxs[(size_t)l], &T);
					texts[(size_t)l] = T;  // released with the lane's context (kept for the next call, or destroyed)
					trace.add("create", l, 0, tc);
					size_t nj = 0;
					double sam_per_fastq = 1.12;
					if (trc == URMAPX_E_NOMEM) text_nomem.store(true);  // T stays null: this lane's chunks go back unmapped
					else if (trc) fail.raise(trc, std::string("urmapx_text_create: ") + urmapx_strerror(trc));
					// The copy back is a stage of its own (urmapx_text_set_deferred): chunk i's text crosses PCIe while the lane's stream takes
					// chunk i + 1 in and maps it; the lane waits for chunk i's copy when chunk i + 1 has been enqueued to its end.
					// URMAPX_NO_DEFERRED_COPY=1 (measurement): every chunk is waited for before the next is taken.
					const bool defer = T && !getenv("URMAPX_NO_DEFERRED_COPY") && urmapx_text_set_deferred(T, 1) == URMAPX_OK;
					auto ok_so_far = [](const TextJob &q) { return !q.rc && (q.rep.reason == 0 || q.rep.reason == URMAPX_TEXT_DEFERRED); };
					auto hand_on = [&](std::unique_ptr<TextJob> &q) {
						if (!q->rc && q->rep.reason == URMAPX_TEXT_DEFERRED) {
							const auto tw0 = now();
							const double ta = trace.ms();
							q->rc = urmapx_text_wait(T, &q->rep);
							trace.add("wait", l, 0, ta);
							std::lock_guard<std::mutex> g(gpu_time_lock);
							t_gpu += secs(tw0, now());
						}
						tmapped[(size_t)l]->push(std::move(q));
					};
					std::unique_ptr<TextJob> j, prev;
					while (tparsed[(size_t)l]->pop(j)) {
						memset(&j->rep, 0, sizeof j->rep);
						j->rc = 0;
						if (T && !fail.set.load() && !stop.load()) {
							const auto tg0 = now();
							const double ta = trace.ms();
							// the SAM buffer is sized from the previous chunk's text (150-base reads: 1.09 x their FASTQ text); a chunk
							// that needs more says so and its text is fetched into a larger one
							const size_t fq_bytes = j->nbytes + j->nbytes2;
							if (!TextJob::grow(j->out, j->out_cap, (size_t)((double)fq_bytes * sam_per_fastq * 1.04) + (1u << 20))) j->rc = URMAPX_E_NOMEM;
							if (!j->rc)
								j->rc = paired ? urmapx_text_map_pe(T, j->in, j->nbytes, j->in2, j->nbytes2, minq, j->out, j->out_cap, &j->rep)
								               : urmapx_text_map_se(T, j->in, j->nbytes, minq, j->out, j->out_cap, &j->rep);
							if (!j->rc && j->rep.reason == URMAPX_TEXT_SAM_CAP) {
								if (!TextJob::grow(j->out, j->out_cap, (size_t)j->rep.sam_bytes + j->rep.sam_bytes / 16 + (1u << 20))) j->rc = URMAPX_E_NOMEM;
								else j->rc = urmapx_text_fetch_sam(T, j->out, j->out_cap, &j->rep);
							}
							if (ok_so_far(*j) && fq_bytes && j->rep.records) sam_per_fastq = (double)j->rep.sam_bytes / (double)fq_bytes;
							j->tab.clear();
							if (ok_so_far(*j) && ftab && j->rep.records) {
								const uint32_t np = j->rep.records / 2;
								j->tab_res.resize((size_t)2 * np); j->tab_info.resize(np); j->tab_ends.resize((size_t)4 * np); j->tab_lens2.resize(np);
								j->rc = urmapx_text_fetch_pairs(T, np, j->tab_res.data(), j->tab_info.data(), j->tab_ends.data(), j->tab_lens2.data());
								if (!j->rc) {
									const auto tf0 = now();
									const int TT = std::max(1, host_threads / n_lanes);
									std::vector<std::string> part((size_t)TT);
#pragma omp parallel for schedule(static, 1) num_threads(TT)
									for (int t = 0; t < TT; ++t) {
										std::string &o = part[(size_t)t];
										const uint32_t lo = (uint32_t)((uint64_t)np * (uint64_t)t / (uint64_t)TT), hi = (uint32_t)((uint64_t)np * (uint64_t)(t + 1) / (uint64_t)TT);
										o.reserve((size_t)(hi - lo) * 72);
										for (uint32_t u = lo; u < hi; ++u) {
											const uint32_t *e = j->tab_ends.data() + (size_t)4 * u;
											const uint32_t s0 = (u ? e[-1] + 1u : 0u) + 1u;  // behind the '@'
											append_tab_pe(o, I, &j->tab_res[(size_t)2 * u], &j->tab_res[(size_t)2 * u + 1], &j->tab_info[u], j->in + s0, e[0] - s0,
											              e[1] - e[0] - 1u, j->tab_lens2[u], have_sam ? 1 : 0);
										}
									}
									for (const std::string &o : part) j->tab += o;
									std::lock_guard<std::mutex> g(gpu_time_lock);
									t_format += secs(tf0, now());
								}
							}
							trace.add("gpu", l, (size_t)l + nj * (size_t)n_lanes, ta);
							std::lock_guard<std::mutex> g(gpu_time_lock);
							t_gpu += secs(tg0, now());
						} else
							j->rep.reason = 0xFFFFu;  // not mapped: the phase is ending
						if (prev) hand_on(prev);
						if (defer) prev = std::move(j);
						else tmapped[(size_t)l]->push(std::move(j));
						++nj;
					}
					if (prev) hand_on(prev);
					tmapped[(size_t)l]->close();
				});
			for (auto &t : tlanes) t.join();
			treader.join();
			twriter.join();
			{
				const double tf = trace.ms();
				std::unique_ptr<TextJob> j;
				while (tfree.try_pop(j)) j.reset();
				trace.add("unpin", -1, 0, tf);
			}
			if (fq >= 0) close(fq);
			if (fq2 >= 0) close(fq2);
			if (!handed_back) { resume_off = reader_end; resume_off2 = reader_end2; }
			host_phase = !fail.set.load() && (streamed ? (handed_back || !stream_done) : (resume_off < fsize || (paired && resume_off2 < fsize2)));
			if (host_phase) {
				// a seekable source continues at its (uncompressed) offset; a pipe with the bytes taken from it and not mapped
				auto resume = [&](FastqReader &r, SeqSource &src, uint64_t at, std::vector<char> &back, std::vector<char> &rest) {
					if (!streamed || !src.is_pipe()) return r.resume_at(at, lines_done);  // (a shard's reader: positioned again, same rules)
					back.insert(back.end(), rest.begin(), rest.end());
					return r.resume_with_prefix(std::move(back), lines_done);
				};
				if (!resume(rd, src1, resume_off, pipe_back, carry) || (paired && !resume(rd2, src2, resume_off2, pipe_back2, carry2)))
					fail.raise(URMAPX_E_IO, std::string("Cannot continue reading ") + fastq1);
			}
			if (fail.set.load()) host_phase = false;
		}
	}
	if (host_phase) {

	// batch b travels through parsed[b mod lanes] -> lane thread -> mapped[b mod lanes]; the writer visits the lanes in the
	// same round-robin order, so batches come back in input order without a reorder buffer
	using JobChannel = Channel<std::unique_ptr<Job>>;
	std::vector<std::unique_ptr<JobChannel>> parsed, mapped;
	for (int l = 0; l < n_lanes; ++l) {
		parsed.emplace_back(new JobChannel(2));
		mapped.emplace_back(new JobChannel(1));
	}
	JobChannel recycled((size_t)(8 + 6 * n_lanes));  // finished jobs go back to the reader: their arrays are reused
	// -map2: the second file is parsed by its own thread while the reader parses the first
	struct Side { FastqBatch b; std::string e; bool more = false; } side2;
	Channel<int> go2(1), done2(1);
	std::thread reader2;
	if (paired)
		reader2 = std::thread([&] {
			omp_workers_sleep_when_idle();
			omp_set_num_threads(std::max(1, host_threads / 2));  // a new thread starts from the default team size, not main's
			int x;
			while (go2.pop(x)) {
				side2.b.clear();
				side2.e.clear();
				side2.more = rd2.next_batch(side2.b, batch / 2, side2.e);
				done2.push(1);
			}
		});
	std::thread reader([&] {
		omp_workers_sleep_when_idle();
		omp_set_num_threads(paired ? std::max(1, host_threads - host_threads / 2) : host_threads);
		FastqBatch a;
		for (size_t b = 0; !fail.set.load(); ++b) {
			std::unique_ptr<Job> j;
			if (!recycled.try_pop(j)) j = std::make_unique<Job>();
			j->reads.clear();
			std::string e;
			bool more;
			const auto tp0 = now();
			if (!paired)
				more = rd.next_batch(j->reads, batch, e);
			else {  // mates interleaved: reads 2i, 2i+1 (map2.cpp:27-32 reads one record from each file under one lock)
				go2.push(1);
				a.clear();
				more = rd.next_batch(a, batch / 2, e);
				int x;
				done2.pop(x);
				const FastqBatch &bb = side2.b;
				if (e.empty()) e = side2.e;
				if (e.empty() && (more != side2.more || a.size() != bb.size())) e = std::string("Premature end of file in FASTQ") + (a.size() > bb.size() ? "2" : "1");
				if (e.empty()) {
					omp_workers_sleep_when_idle();
					omp_set_num_threads(host_threads);
					interleave_batches(a, bb, j->reads);
					omp_workers_sleep_when_idle();
					omp_set_num_threads(std::max(1, host_threads - host_threads / 2));
				}
			}
			t_parse += secs(tp0, now());
			if (!e.empty()) { fail.raise(URMAPX_E_FORMAT, e); break; }
			if (!more) break;
			parsed[b % (size_t)n_lanes]->push(std::move(j));
		}
		for (auto &c : parsed) c->close();
		go2.close();
	});
	// SAM text of a batch is formatted by all host threads, each on a contiguous range of reads (pairs); the pieces go to
	// the flusher thread, which writes them at their file offsets (input order) while the next batch is being formatted.
	struct Text { std::vector<std::string> outs; std::vector<uint64_t> at; };
	Channel<std::unique_ptr<Text>> to_flush(2), text_pool(4);
	std::thread flusher([&] {
		std::unique_ptr<Text> x;
		while (to_flush.pop(x)) {
			const auto tw0 = now();
			if (!fail.set.load()) {
				if (!sink.reserve(x->at[(size_t)host_threads])) fail.raise(URMAPX_E_IO, std::string("Error writing ") + samout);
				if (!sink.seekable()) {  // a pipe: the pieces one after the other
					for (int t = 0; t < host_threads; ++t) {
						const std::string &out = x->outs[(size_t)t];
						if (!sink.write_at(out.data(), out.size(), x->at[(size_t)t], 1)) { fail.raise(URMAPX_E_IO, std::string("Error writing ") + samout); break; }
					}
				} else {
#pragma omp parallel for schedule(static, 1) num_threads(host_threads)
				for (int t = 0; t < host_threads; ++t) {
					const std::string &out = x->outs[(size_t)t];
					if (!sink.write_at(out.data(), out.size(), x->at[(size_t)t], 1)) fail.raise(URMAPX_E_IO, std::string("Error writing ") + samout);
				}
				}
			}
			t_write += secs(tw0, now());
			text_pool.push(std::move(x));
		}
	});
	std::thread writer([&] {
		omp_workers_sleep_when_idle();
		omp_set_num_threads(host_threads);
		std::unique_ptr<Job> j;
		std::unique_ptr<Text> text;
		struct Cnt { unsigned long long accept = 0, reject = 0, nohit = 0, unsupported = 0; };
		for (size_t b = 0; mapped[b % (size_t)n_lanes]->pop(j); ++b) {
			if (fail.set.load()) { recycled.push(std::move(j)); continue; }
			const uint32_t n = j->reads.size();
			const uint32_t units = paired ? n / 2 : n;
			std::vector<Cnt> cnt((size_t)host_threads);
			if (!text && !text_pool.try_pop(text)) text = std::make_unique<Text>();
			text->outs.resize((size_t)host_threads);
			std::vector<std::string> &outs = text->outs;
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
					if (have_sam && !paired)
						append_sam_record(out, I, r, j->ops.data(), 0, "*", 0xFFFFFFFFu, 0, j->reads.label(i),
						                  j->reads.bases.data() + off, j->reads.quals.data() + off, L);
					if (have_sam && paired && (i & 1) == 0) {
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
			if (have_sam) {
				std::vector<uint64_t> &at = text->at;
				at.assign((size_t)host_threads + 1, 0);
				at[0] = sam_off;
				for (int t = 0; t < host_threads; ++t) at[(size_t)t + 1] = at[(size_t)t] + outs[(size_t)t].size();
				sam_off = at[(size_t)host_threads];
				to_flush.push(std::move(text));
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
						                               have_sam ? 1 : 0, line, sizeof line);
						tabs[(size_t)t].append(line, k);
					}
				}
				for (const std::string &tb : tabs)
					if (fwrite(tb.data(), 1, tb.size(), ftab) != tb.size()) fail.raise(URMAPX_E_IO, std::string("Error writing ") + tabout);
			}
			recycled.push(std::move(j));
		}
		to_flush.close();
	});
	std::vector<std::thread> lanes;
	for (int l = 0; l < n_lanes; ++l)
		lanes.emplace_back([&, l] {
			urmapx_ctx *C = ctxs[(size_t)l];
			omp_workers_sleep_when_idle();
			const bool pinned = pin_to(places[(size_t)(l % gpus)]);
			struct Unpin { bool on; const cpu_set_t &m; int n; ~Unpin() { unpin_team(on, m, n); } } unpin{pinned, caller_mask, 1};
			(void)hipSetDevice(phys(l % gpus));
			std::unique_ptr<Job> j;
			while (parsed[(size_t)l]->pop(j)) {
				const uint32_t n = j->reads.size();
				if (!fail.set.load()) {
					j->results.resize(n);
					j->ops.resize((size_t)n * URMAPX_MAX_PATH_OPS);
					// page-lock what crosses PCIe (the arrays are recycled, so this happens once per job object and size)
					if (!getenv("URMAPX_NO_PIN")) {
						j->pin_bases.hold(j->reads.bases);
						j->pin_offs.hold(j->reads.offs);
						j->pin_results.hold(j->results);
						j->pin_ops.hold(j->ops);
					}
					size_t used = 0;
					const auto tg0 = now();
					int mrc = paired ? urmapx_map_pe(C, j->reads.bases.data(), j->reads.offs.data(), n / 2, j->results.data(), j->ops.data(),
					                                 j->ops.size(), &used)
					                 : urmapx_map_se(C, j->reads.bases.data(), j->reads.offs.data(), n, j->results.data(), j->ops.data(),
					                                 j->ops.size(), &used);
					if (mrc != URMAPX_OK && mrc != URMAPX_E_UNSUPPORTED)
						fail.raise(mrc, std::string(paired ? "urmapx_map_pe: " : "urmapx_map_se: ") + urmapx_strerror(mrc));
					else if (paired && ftab) {
						j->info.resize(n / 2);
						mrc = urmapx_ctx_get_pair_info(C, j->info.data(), n / 2);
						if (mrc) fail.raise(mrc, std::string("urmapx_ctx_get_pair_info: ") + urmapx_strerror(mrc));
					}
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
	flusher.join();
	{  // page-locked arrays are released before their contexts go
		std::unique_ptr<Job> j;
		while (recycled.try_pop(j)) j.reset();
	}
	}  // host phase
	if (have_sam && !sink.finish(sam_off)) fail.raise(URMAPX_E_IO, std::string("Error writing ") + samout);
	if (ftab) fclose(ftab);
	const auto t2 = std::chrono::steady_clock::now();
	range.t_begin = std::chrono::duration<double>(t1.time_since_epoch()).count();
	range.t_end = std::chrono::duration<double>(t2.time_since_epoch()).count();
	trace.dump();
	run_ok = !fail.set.load();
	release();
	if (report) {
		report->reads = n_reads; report->mapped_q = n_accept; report->mapped_lowq = n_reject; report->unmapped = n_nohit;
		report->unsupported = n_unsupported;
		report->seconds = secs(t1, t2); report->parse_s = t_parse; report->gpu_s = t_gpu; report->format_s = t_format; report->write_s = t_write;
		report->host_threads = host_threads; report->lanes = n_lanes;
		report->write_threads = write_threads_used; report->text_on_device = text_on_device ? 1 : 0; report->input_bytes = input_bytes;
		snprintf(report->medium, sizeof report->medium, "%s", have_sam ? medium_name.c_str() : "none");
		report->dev_h2d_s = dev_ms[0] * 1e-3; report->dev_parse_s = dev_ms[1] * 1e-3; report->dev_map_s = dev_ms[2] * 1e-3;
		report->dev_format_s = dev_ms[3] * 1e-3; report->dev_d2h_s = dev_ms[4] * 1e-3;
		report->dev_map_search_s = dev_ms[5] * 1e-3; report->dev_map_dp_s = dev_ms[6] * 1e-3; report->map_enqueue_s = dev_ms[7] * 1e-3;
		report->shards = 1;
		snprintf(report->placement, sizeof report->placement, "%s", placement.c_str());
	}
	if (fail.set.load()) { say(fail.msg); return fail.code; }
	return n_unsupported ? URMAPX_E_UNSUPPORTED : URMAPX_OK;
}

}  // namespace

// cmd_map / cmd_map2 as a library call; with sam_shards > 1 as that many pipelines side by side, each over its part of the
// input with its own SAM file, devices and writer (the reference appends every record to one file under one lock,
// output1.cpp:10-16: one output stream is what bounds a run on several GPUs)
static int map_files_entry(urmapx_index *I, const urmapx_map_options *opt, const char *fastq1, const char *fastq2,
                           const char *samout, const char *tabout, urmapx_map_report *report, char *err, size_t errcap);
extern "C" int urmapx_map_files(urmapx_index *I, const urmapx_map_options *opt, const char *fastq1, const char *fastq2,
                                const char *samout, const char *tabout, urmapx_map_report *report, char *err, size_t errcap) {
	// what the call spent in allocation calls (all of its threads): device arrays of the lanes' contexts, page-locked chunk buffers
	AllocClock &ac = alloc_clock();
	const uint64_t ns0[2] = {ac.ns[0].load(), ac.ns[1].load()}, c0[2] = {ac.calls[0].load(), ac.calls[1].load()};
	const int rc = map_files_entry(I, opt, fastq1, fastq2, samout, tabout, report, err, errcap);
	if (report) {
		report->alloc_dev_s = (double)(ac.ns[0].load() - ns0[0]) * 1e-9; report->alloc_pinned_s = (double)(ac.ns[1].load() - ns0[1]) * 1e-9;
		report->alloc_dev_calls = (uint32_t)(ac.calls[0].load() - c0[0]); report->alloc_pinned_calls = (uint32_t)(ac.calls[1].load() - c0[1]);
	}
	return rc;
}
static int map_files_entry(urmapx_index *I, const urmapx_map_options *opt, const char *fastq1, const char *fastq2,
                           const char *samout, const char *tabout, urmapx_map_report *report, char *err, size_t errcap) {
	if (err && errcap) err[0] = 0;
	if (!I || !opt || !fastq1) return URMAPX_E_ARG;
	const int shards = opt->sam_shards > 1 ? opt->sam_shards : 1;
	if (shards == 1) return map_files_impl(I, opt, InputRange(), fastq1, fastq2, samout, tabout, report, err, errcap);
	auto say = [&](const std::string &s) {
		if (err && errcap) snprintf(err, errcap, "%s", s.c_str());
	};
	const bool paired = fastq2 != nullptr;
	const int gpus = opt->gpus > 0 ? opt->gpus : 1;
	if (!samout) { say("-samshards needs -samout (the shards are named after it)"); return URMAPX_E_ARG; }
	if (shards > 64) { say("-samshards must be 1..64"); return URMAPX_E_ARG; }
	if (gpus % shards != 0 && shards % gpus != 0) { say("-samshards must divide -gpus or be a multiple of it"); return URMAPX_E_ARG; }
	const int host_threads = opt->host_threads > 0 ? opt->host_threads : std::min(16, std::max(1, (int)std::thread::hardware_concurrency()));
	// plain seekable files are cut; anything else (.gz, a pipe) goes to shard 0 whole
	auto plain_size = [](const char *path, int &fd) -> uint64_t {
		fd = -1;
		const size_t l = strlen(path);
		if ((l > 3 && !strcmp(path + l - 3, ".gz")) || !strcmp(path, "-")) return 0;
		fd = open(path, O_RDONLY);
		struct stat st;
		if (fd >= 0 && (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0)) { close(fd); fd = -1; }
		return fd >= 0 ? (uint64_t)st.st_size : 0;
	};
	int fd1 = -1, fd2 = -1;
	const uint64_t fsize1 = plain_size(fastq1, fd1), fsize2 = paired ? plain_size(fastq2, fd2) : 0;
	const bool cut = fd1 >= 0 && (!paired || fd2 >= 0) && !getenv("URMAPX_HOST_TEXT_NO_SHARDS");
	std::vector<InputRange> ranges((size_t)shards);
	// the cutting (and, for pairs, the line counts that place the cuts of the mates' file) belongs to the run: it is inside
	// report->seconds and reported as shard_scan_s
	const auto t_scan0 = std::chrono::steady_clock::now();
	{
		// shard s starts at the first record start at or behind byte fsize * s / shards of the first file -- and, for pairs, at
		// the same line of the second file
		std::vector<uint64_t> at((size_t)shards + 1, 0), at2((size_t)shards + 1, 0), lines((size_t)shards + 1, 0);
		at[(size_t)shards] = fsize1; at2[(size_t)shards] = fsize2;
		if (cut) {
			for (int s = 1; s < shards; ++s) {
				const uint64_t nominal = fsize1 / (uint64_t)shards * (uint64_t)s;
				uint64_t a = nominal > at[(size_t)s - 1] ? find_record_start(fd1, nominal, fsize1) : at[(size_t)s - 1];
				if (a == 0 || a < at[(size_t)s - 1]) a = fsize1;  // no record start in sight: the shard in front takes the rest
				at[(size_t)s] = a;
			}
			if (paired) {  // the mates' file is cut at the same LINES: the first file's parts are counted, the second's offsets looked up
				for (int s = 1; s < shards; ++s) lines[(size_t)s] = lines[(size_t)s - 1] + count_newlines(fd1, at[(size_t)s - 1], at[(size_t)s], host_threads);
				const std::vector<uint64_t> o2 = line_offsets(fd2, fsize2, std::vector<uint64_t>(lines.begin() + 1, lines.begin() + shards), host_threads);
				for (int s = 1; s < shards; ++s) at2[(size_t)s] = o2[(size_t)s - 1];
			}
		}
		for (int s = 0; s < shards; ++s) {
			InputRange &r = ranges[(size_t)s];
			r.header = s == 0;
			r.on = cut;
			r.lo[0] = at[(size_t)s]; r.hi[0] = at[(size_t)s + 1]; r.lo[1] = at2[(size_t)s]; r.hi[1] = at2[(size_t)s + 1];
			r.lines_before = lines[(size_t)s];
			if (cut && !paired && s > 0) {
				// single-end: nothing needs the line counts but a message about a malformed record -- the shard's reader counts then
				const std::string path = fastq1;
				const uint64_t upto = at[(size_t)s];
				r.lazy_lines = [path, upto, host_threads]() -> uint64_t {
					const int fd = open(path.c_str(), O_RDONLY);
					if (fd < 0) return 0;
					const uint64_t n = count_newlines(fd, 0, upto, host_threads);
					close(fd);
					return n;
				};
			}
		}
		if (paired && cut) ranges[(size_t)shards - 1].hi[1] = fsize2;
	}
	if (fd1 >= 0) close(fd1);
	if (fd2 >= 0) close(fd2);
	const double scan_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scan0).count();
	auto empty_shards_from = [&](int s0) {
		for (int s = s0; s < shards; ++s) {
			FILE *f = fopen((std::string(samout) + "." + std::to_string(s)).c_str(), "wb");
			if (f) fclose(f);
			if (tabout) { f = fopen((std::string(tabout) + "." + std::to_string(s)).c_str(), "wb"); if (f) fclose(f); }
		}
	};
	if (!cut) {
		// Input that cannot be cut (.gz, BGZF, a pipe): ONE pipeline with all the devices and lanes the caller gave writes shard 0, the
		// other shards are empty files -- `cat` of the shards is still the one file.  (Until round 4 shard 0 ran on gpus / shards of
		// the devices: `-gpus 8 -samshards 8 reads.fq.gz` mapped the whole file on one GPU.)
		urmapx_map_options o = *opt;
		o.sam_shards = 0;
		const std::string sam0 = std::string(samout) + ".0", tab0 = tabout ? std::string(tabout) + ".0" : std::string();
		const int rc1 = map_files_impl(I, &o, InputRange(), fastq1, fastq2, sam0.c_str(), tabout ? tab0.c_str() : nullptr, report, err, errcap);
		if (rc1 == URMAPX_OK || rc1 == URMAPX_E_UNSUPPORTED) empty_shards_from(1);
		else { remove(sam0.c_str()); if (tabout) remove(tab0.c_str()); }
		if (report) report->shards = shards;
		return rc1;
	}
	// devices: shard s takes gpus / shards of them, or shares device s mod gpus
	const char *forced = getenv("URMAPX_FORCE_DEVICE");
	auto phys = [&](int g) { return forced ? atoi(forced) : opt->first_gpu + g; };
	const int per = gpus >= shards ? gpus / shards : 1;
	// one replica of the index on the first device of every shard (a shard with several devices makes the others' itself)
	std::vector<urmapx_index *> replicas((size_t)gpus, nullptr);
	int rc = urmapx_index_upload(I, phys(0));
	replicas[0] = I;
	for (int g = per; g < gpus && !rc; g += per) rc = urmapx_index_replicate(I, phys(g), &replicas[(size_t)g]);
	auto release = [&]() { for (int g = 1; g < gpus; ++g) urmapx_index_close(replicas[(size_t)g]); };
	if (rc) { say(std::string("Uploading index to the GPU: ") + urmapx_strerror(rc)); release(); return rc; }
	std::vector<urmapx_map_report> reps((size_t)shards);
	std::vector<int> rcs((size_t)shards, 0);
	std::vector<std::string> errs((size_t)shards);
	std::vector<std::thread> th;
	const auto t0 = std::chrono::steady_clock::now();
	for (int s = 0; s < shards; ++s)
		th.emplace_back([&, s] {
			urmapx_map_options o = *opt;
			o.sam_shards = 0;
			o.gpus = per;
			// pipelines that share a device share its lanes too: K contexts per device in all, as in a run of one pipeline
			if (shards > gpus) o.streams = std::max(1, (opt->streams > 0 ? opt->streams : 2) * gpus / shards);
			const int g0 = gpus >= shards ? s * per : s % gpus;
			o.first_gpu = opt->first_gpu + g0;
			o.host_threads = std::max(2, host_threads / shards);
			const std::string sam = std::string(samout) + "." + std::to_string(s);
			const std::string tab = tabout ? std::string(tabout) + "." + std::to_string(s) : std::string();
			char e[512];
			e[0] = 0;
			memset(&reps[(size_t)s], 0, sizeof reps[(size_t)s]);
			const InputRange &r = ranges[(size_t)s];
			if (r.lo[0] >= r.hi[0]) {  // nothing left for this shard (fewer records than shards)
				FILE *f = fopen(sam.c_str(), "wb");
				if (f) fclose(f);
				if (tabout) { f = fopen(tab.c_str(), "wb"); if (f) fclose(f); }
				return;
			}
			rcs[(size_t)s] = map_files_impl(replicas[(size_t)g0], &o, r, fastq1, fastq2, sam.c_str(), tabout ? tab.c_str() : nullptr, &reps[(size_t)s], e, sizeof e);
			errs[(size_t)s] = e;
		});
	for (auto &t : th) t.join();
	double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	{
		double b = 0, e = 0;
		for (const InputRange &r : ranges)
			if (r.t_end > 0) { b = b == 0 ? r.t_begin : std::min(b, r.t_begin); e = std::max(e, r.t_end); }
		if (e > b) wall = e - b;
	}
	wall += scan_s;
	release();
	int out_rc = URMAPX_OK;
	for (int s = 0; s < shards; ++s)
		if (rcs[(size_t)s] && rcs[(size_t)s] != URMAPX_E_UNSUPPORTED && out_rc == URMAPX_OK) { out_rc = rcs[(size_t)s]; say(errs[(size_t)s]); }
	if (out_rc != URMAPX_OK)  // a failed run leaves no partial shard files behind (the first error is the one reported)
		for (int s = 0; s < shards; ++s) {
			remove((std::string(samout) + "." + std::to_string(s)).c_str());
			if (tabout) remove((std::string(tabout) + "." + std::to_string(s)).c_str());
		}
	if (report) {
		memset(report, 0, sizeof *report);
		for (const urmapx_map_report &r : reps) {
			report->reads += r.reads; report->mapped_q += r.mapped_q; report->mapped_lowq += r.mapped_lowq; report->unmapped += r.unmapped;
			report->unsupported += r.unsupported; report->parse_s += r.parse_s; report->gpu_s += r.gpu_s; report->format_s += r.format_s;
			report->write_s += r.write_s; report->lanes += r.lanes; report->host_threads += r.host_threads; report->input_bytes += r.input_bytes;
			report->dev_h2d_s += r.dev_h2d_s; report->dev_parse_s += r.dev_parse_s; report->dev_map_s += r.dev_map_s;
			report->dev_format_s += r.dev_format_s; report->dev_d2h_s += r.dev_d2h_s;
			report->dev_map_search_s += r.dev_map_search_s; report->dev_map_dp_s += r.dev_map_dp_s; report->map_enqueue_s += r.map_enqueue_s;
			report->write_threads = std::max(report->write_threads, r.write_threads);
			report->text_on_device |= r.text_on_device;
			if (r.medium[0]) memcpy(report->medium, r.medium, sizeof report->medium);
			if (r.placement[0]) {
				const size_t have = strlen(report->placement);
				snprintf(report->placement + have, sizeof report->placement - have, "%s%s", have ? " | " : "", r.placement);
			}
		}
		report->seconds = wall;
		report->shards = shards;
		report->shard_scan_s = scan_s;
	}
	if (out_rc != URMAPX_OK) return out_rc;
	return report && report->unsupported ? URMAPX_E_UNSUPPORTED : URMAPX_OK;
}

extern "C" void urmapx_host_pool_trim(void) {
	LanePool::get().purge(nullptr);
	HostPool::get().trim();
}
namespace urx {
void lane_pool_purge(const urmapx_index *I) { LanePool::get().purge(I); }
}
