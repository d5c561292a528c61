// host_pipe.hpp -- the host-memory entry points' pipeline: chunked H2D / kernels / D2H on three streams.
//
// The reference's CUDA plugin stages every call through pinned buffers and chunks it at 8192 frames
// (src/cuda.rs:150-155,185-199,343-351) with one stream and a synchronise per chunk.  Here a call is cut into chunks of
// ~16 MiB of PCM (whole clips, or frame-aligned pieces of a long clip), and chunk k's upload, chunk k-1's kernels and
// chunk k-2's download run at the same time on a copy-in stream, the context's stream and a copy-out stream (two device
// buffers each way, events between the streams).  Caller memory that is already pinned (melspec_host_alloc, hipHostMalloc,
// hipHostRegister) is DMA'd in place; pageable memory goes through pinned staging buffers that a small pool of helper
// threads fills / drains with memcpy while the GPU works on the neighbouring chunks (one thread copies ~12 GB/s, a gen5
// x16 link moves ~50).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace melspec {

struct CopyJob {
    void *dst;
    const void *src;
    size_t bytes;
};

// memcpy of a job list by `workers` helper threads plus the caller; blocks until done
class CopyPool {
public:
    explicit CopyPool(int workers) {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this, i] { loop(i + 1); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(const std::vector<CopyJob> &jobs) {
        size_t total = 0;
        for (const CopyJob &j : jobs) total += j.bytes;
        if (total == 0) return;
        const int parts = static_cast<int>(threads_.size()) + 1;
        if (total < (1u << 20) || parts == 1) {          // not worth a wake-up
            for (const CopyJob &j : jobs) std::memcpy(j.dst, j.src, j.bytes);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            jobs_ = &jobs;
            total_ = total;
            pending_ = parts - 1;
            ++gen_;
        }
        cv_.notify_all();
        part(0, parts);
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [this] { return pending_ == 0; });
        jobs_ = nullptr;
    }

private:
    // participant p of n copies bytes [total*p/n, total*(p+1)/n) of the concatenated jobs (64-byte granularity)
    void part(int p, int n) const {
        const size_t lo = (total_ * static_cast<size_t>(p) / n) & ~static_cast<size_t>(63);
        const size_t hi = p + 1 == n ? total_ : (total_ * static_cast<size_t>(p + 1) / n) & ~static_cast<size_t>(63);
        size_t pos = 0;
        for (const CopyJob &j : *jobs_) {
            const size_t a = pos > lo ? pos : lo, b = pos + j.bytes < hi ? pos + j.bytes : hi;
            if (a < b) std::memcpy(static_cast<char *>(j.dst) + (a - pos), static_cast<const char *>(j.src) + (a - pos), b - a);
            pos += j.bytes;
            if (pos >= hi) break;
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            part(id, static_cast<int>(threads_.size()) + 1);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::vector<CopyJob> *jobs_ = nullptr;
    size_t total_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// one unit of a host call: n samples at src -> frames * n_mels floats at dst (a whole clip or a frame-aligned piece of one)
struct HostSeg {
    const float *src;
    uint64_t n;
    float *dst;
    uint64_t frames;
};

inline bool host_ptr_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();            // pageable memory the runtime has never seen
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

struct HostPipe {
    static constexpr int kBuf = 2;
    static constexpr size_t kDirectPieces = 32;          // more DMA requests than this per chunk: gather through staging instead
    static constexpr uint64_t kDirectBytes = 32u << 20;  // calls up to this size (PCM + mel): no pipeline, see run()
    hipStream_t s_in = nullptr, s_out = nullptr;
    void *d_in[kBuf] = {}, *d_out[kBuf] = {};
    size_t d_in_cap[kBuf] = {}, d_out_cap[kBuf] = {};
    void *h_in[kBuf] = {}, *h_out[kBuf] = {};
    size_t h_in_cap[kBuf] = {}, h_out_cap[kBuf] = {};
    hipEvent_t ev_in[kBuf] = {}, ev_cmp[kBuf] = {}, ev_out[kBuf] = {};
    CopyPool *pool = nullptr;
    bool ready = false;

    hipError_t init() {
        if (ready) return hipSuccess;
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking)) != hipSuccess) return e;
        for (int b = 0; b < kBuf; ++b) {
            if ((e = hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_cmp[b], hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_out[b], hipEventDisableTiming)) != hipSuccess) return e;
        }
        int workers = 3;
#ifdef MELSPEC_LAB
        if (const char *e = std::getenv("MELSPEC_COPY_THREADS")) workers = std::atoi(e) > 0 && std::atoi(e) <= 63 ? std::atoi(e) : workers;
#endif
        pool = new CopyPool(workers);
        ready = true;
        return hipSuccess;
    }
    void release() {
        if (s_in) { (void)hipStreamSynchronize(s_in); (void)hipStreamDestroy(s_in); s_in = nullptr; }
        if (s_out) { (void)hipStreamSynchronize(s_out); (void)hipStreamDestroy(s_out); s_out = nullptr; }
        for (int b = 0; b < kBuf; ++b) {
            if (d_in[b]) (void)hipFree(d_in[b]);
            if (d_out[b]) (void)hipFree(d_out[b]);
            if (h_in[b]) (void)hipHostFree(h_in[b]);
            if (h_out[b]) (void)hipHostFree(h_out[b]);
            d_in[b] = d_out[b] = h_in[b] = h_out[b] = nullptr;
            d_in_cap[b] = d_out_cap[b] = h_in_cap[b] = h_out_cap[b] = 0;
            if (ev_in[b]) (void)hipEventDestroy(ev_in[b]);
            if (ev_cmp[b]) (void)hipEventDestroy(ev_cmp[b]);
            if (ev_out[b]) (void)hipEventDestroy(ev_out[b]);
            ev_in[b] = ev_cmp[b] = ev_out[b] = nullptr;
        }
        delete pool;
        pool = nullptr;
        ready = false;
    }

    static hipError_t grow_dev(void *&p, size_t &cap, size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    static hipError_t grow_host(void *&p, size_t &cap, size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) cap = bytes;
        return e;
    }

    // launch(d_in, offsets, lengths, n, d_out, out_offsets, stream) -> library status (0 = ok): the kernels of one chunk,
    // queued on `compute`; clip i of the chunk = d_in + offsets[i] (lengths[i] samples) -> d_out + out_offsets[i].
    // Returns 0, a positive hipError_t, or the launch's status; *where names the failing call.
    template <class Launch>
    int run(const std::vector<HostSeg> &segs, int n_mels, uint64_t chunk_samples, hipStream_t compute, Launch &&launch, const char **where) {
        const int rc = run_impl(segs, n_mels, chunk_samples, compute, launch, where);
        if (rc != 0) {
            // A call that fails half-way must not return while copies are still reading `samples` or writing `out`, and the next call
            // reuses the buffers and events from chunk 0: drain all three streams (their own errors do not matter any more).
            if (s_in) (void)hipStreamSynchronize(s_in);
            (void)hipStreamSynchronize(compute);
            if (s_out) (void)hipStreamSynchronize(s_out);
            (void)hipGetLastError();
            if (!ready) release();          // init() failed part-way: give back what it had created
        }
        return rc;
    }

    template <class Launch>
    int run_impl(const std::vector<HostSeg> &segs, int n_mels, uint64_t chunk_samples, hipStream_t compute, Launch &&launch, const char **where) {
#define MS_PIPE_TRY(expr)                                        \
    do {                                                         \
        const hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) { *where = #expr; (void)hipGetLastError(); return static_cast<int>(e_) > 0 ? static_cast<int>(e_) : 1; } \
    } while (0)
        *where = "";
        if (segs.empty()) return 0;
        MS_PIPE_TRY(init());
        // cheap screen on the ends of the batch; every merged piece of a chunk is then checked on its own (clips may live anywhere)
        const bool in_pinned = host_ptr_is_pinned(segs.front().src) && host_ptr_is_pinned(segs.back().src + segs.back().n - 1);
        const bool out_pinned = host_ptr_is_pinned(segs.front().dst) &&
                                host_ptr_is_pinned(segs.back().dst + segs.back().frames * static_cast<uint64_t>(n_mels) - 1);
        struct Chunk { size_t first, count; uint64_t samples, out_floats; bool staged_out; };
        std::vector<Chunk> chunks;
        uint64_t total_samples = 0, total_out = 0;
        for (const HostSeg &sg : segs) { total_samples += sg.n; total_out += sg.frames * static_cast<uint64_t>(n_mels); }
        // small calls: one chunk, copied by the runtime straight from / to the caller's memory on the compute stream (its
        // pageable path is as fast as pinned DMA up to a few tens of MB and there is nothing to overlap).  Larger calls: chunks
        // of chunk_samples (16 MiB of PCM; cutting 64 x 10 s into 8 chunks instead of 3 lost 12 %: per-chunk launches and events)
        const bool direct = (total_samples + total_out) * sizeof(float) <= kDirectBytes;
        if (direct) chunk_samples = ~0ull;
        for (size_t i = 0; i < segs.size();) {
            Chunk c{i, 0, 0, 0, false};
            while (i < segs.size() && (c.count == 0 || (c.samples + segs[i].n <= chunk_samples && c.count < 65536))) {
                c.samples += segs[i].n;
                c.out_floats += segs[i].frames * static_cast<uint64_t>(n_mels);
                ++c.count; ++i;
            }
            chunks.push_back(c);
        }
        std::vector<uint64_t> offs, lens, ooffs;
        std::vector<CopyJob> jobs;
        // copies of a chunk with neighbouring pieces merged: (host pointer, bytes, byte offset in the device buffer)
        struct Piece { const char *host; size_t bytes, dev_off; };
        auto pieces_of = [&](const Chunk &c, bool output) {
            std::vector<Piece> v;
            size_t dev = 0;
            for (size_t i = c.first; i < c.first + c.count; ++i) {
                const char *h = output ? reinterpret_cast<const char *>(segs[i].dst) : reinterpret_cast<const char *>(segs[i].src);
                const size_t bytes = static_cast<size_t>(output ? segs[i].frames * static_cast<uint64_t>(n_mels) : segs[i].n) * sizeof(float);
                if (bytes) {
                    if (!v.empty() && v.back().host + v.back().bytes == h) v.back().bytes += bytes;
                    else v.push_back(Piece{h, bytes, dev});
                }
                dev += bytes;
            }
            return v;
        };
        auto all_pinned = [&](const std::vector<Piece> &v) {
            for (const Piece &p : v)
                if (!host_ptr_is_pinned(p.host) || !host_ptr_is_pinned(p.host + p.bytes - 1)) return false;
            return true;
        };
        auto retire_out = [&](size_t k) -> int {          // staged output of chunk k: wait for its D2H, scatter to the caller
            const int b = static_cast<int>(k % kBuf);
            MS_PIPE_TRY(hipEventSynchronize(ev_out[b]));
            if (!chunks[k].staged_out) return 0;
            jobs.clear();
            for (const Piece &p : pieces_of(chunks[k], true))
                jobs.push_back(CopyJob{const_cast<char *>(p.host), static_cast<const char *>(h_out[b]) + p.dev_off, p.bytes});
            pool->run(jobs);
            return 0;
        };
        if (direct && chunks.size() == 1 && pieces_of(chunks[0], false).size() <= kDirectPieces && pieces_of(chunks[0], true).size() <= kDirectPieces) {
            const Chunk &c = chunks[0];
            MS_PIPE_TRY(grow_dev(d_in[0], d_in_cap[0], static_cast<size_t>(c.samples) * sizeof(float) + 16));
            MS_PIPE_TRY(grow_dev(d_out[0], d_out_cap[0], static_cast<size_t>(c.out_floats) * sizeof(float) + 16));
            for (const Piece &p : pieces_of(c, false))
                MS_PIPE_TRY(hipMemcpyAsync(static_cast<char *>(d_in[0]) + p.dev_off, p.host, p.bytes, hipMemcpyHostToDevice, compute));
            uint64_t so = 0, oo = 0;
            for (size_t i = c.first; i < c.first + c.count; ++i) {
                offs.push_back(so); lens.push_back(segs[i].n); ooffs.push_back(oo);
                so += segs[i].n;
                oo += segs[i].frames * static_cast<uint64_t>(n_mels);
            }
            const int rc = launch(static_cast<const float *>(d_in[0]), offs.data(), lens.data(), static_cast<uint32_t>(c.count),
                                  static_cast<float *>(d_out[0]), ooffs.data(), compute);
            if (rc) { *where = "kernel launch"; return rc; }
            for (const Piece &p : pieces_of(c, true))
                MS_PIPE_TRY(hipMemcpyAsync(const_cast<char *>(p.host), static_cast<const char *>(d_out[0]) + p.dev_off, p.bytes, hipMemcpyDeviceToHost, compute));
            MS_PIPE_TRY(hipStreamSynchronize(compute));
            return 0;
        }
        for (size_t k = 0; k < chunks.size(); ++k) {
            Chunk &c = chunks[k];
            const int b = static_cast<int>(k % kBuf);
            if (k >= kBuf) { const int rc = retire_out(k - kBuf); if (rc) return rc; }      // frees h_out[b]; d_out[b] drained
            const size_t in_bytes = static_cast<size_t>(c.samples) * sizeof(float), out_bytes = static_cast<size_t>(c.out_floats) * sizeof(float);
            if (in_bytes + 16 > d_in_cap[b]) {
                if (k >= kBuf) MS_PIPE_TRY(hipEventSynchronize(ev_cmp[b]));                    // the kernels of chunk k-2 read it
                MS_PIPE_TRY(grow_dev(d_in[b], d_in_cap[b], in_bytes + 16));
            }
            if (out_bytes + 16 > d_out_cap[b]) MS_PIPE_TRY(grow_dev(d_out[b], d_out_cap[b], out_bytes + 16));   // chunk k-2 retired above
            // ---- upload
            const std::vector<Piece> pin = pieces_of(c, false);
            if (k >= kBuf) MS_PIPE_TRY(hipStreamWaitEvent(s_in, ev_cmp[b], 0));                // d_in[b] consumed by chunk k-2's kernels
            if (in_pinned && pin.size() <= kDirectPieces && all_pinned(pin)) {
                for (const Piece &p : pin)
                    MS_PIPE_TRY(hipMemcpyAsync(static_cast<char *>(d_in[b]) + p.dev_off, p.host, p.bytes, hipMemcpyHostToDevice, s_in));
            } else {
                if (k >= kBuf) MS_PIPE_TRY(hipEventSynchronize(ev_in[b]));                     // h_in[b] has left the host
                MS_PIPE_TRY(grow_host(h_in[b], h_in_cap[b], in_bytes + 16));
                jobs.clear();
                for (const Piece &p : pin) jobs.push_back(CopyJob{static_cast<char *>(h_in[b]) + p.dev_off, p.host, p.bytes});
                pool->run(jobs);
                MS_PIPE_TRY(hipMemcpyAsync(d_in[b], h_in[b], in_bytes, hipMemcpyHostToDevice, s_in));
            }
            MS_PIPE_TRY(hipEventRecord(ev_in[b], s_in));
            // ---- kernels
            MS_PIPE_TRY(hipStreamWaitEvent(compute, ev_in[b], 0));
            if (k >= kBuf) MS_PIPE_TRY(hipStreamWaitEvent(compute, ev_out[b], 0));             // d_out[b] drained by chunk k-2's download
            offs.clear(); lens.clear(); ooffs.clear();
            uint64_t so = 0, oo = 0;
            for (size_t i = c.first; i < c.first + c.count; ++i) {
                offs.push_back(so); lens.push_back(segs[i].n); ooffs.push_back(oo);
                so += segs[i].n;
                oo += segs[i].frames * static_cast<uint64_t>(n_mels);
            }
            const int rc = launch(static_cast<const float *>(d_in[b]), offs.data(), lens.data(), static_cast<uint32_t>(c.count),
                                  static_cast<float *>(d_out[b]), ooffs.data(), compute);
            if (rc) { *where = "kernel launch"; return rc; }
            MS_PIPE_TRY(hipEventRecord(ev_cmp[b], compute));
            // ---- download
            MS_PIPE_TRY(hipStreamWaitEvent(s_out, ev_cmp[b], 0));
            const std::vector<Piece> pout = pieces_of(c, true);
            c.staged_out = !(out_pinned && pout.size() <= kDirectPieces && all_pinned(pout));
            if (!c.staged_out) {
                for (const Piece &p : pout)
                    MS_PIPE_TRY(hipMemcpyAsync(const_cast<char *>(p.host), static_cast<const char *>(d_out[b]) + p.dev_off, p.bytes, hipMemcpyDeviceToHost, s_out));
            } else if (out_bytes) {
                MS_PIPE_TRY(grow_host(h_out[b], h_out_cap[b], out_bytes + 16));
                MS_PIPE_TRY(hipMemcpyAsync(h_out[b], d_out[b], out_bytes, hipMemcpyDeviceToHost, s_out));
            }
            MS_PIPE_TRY(hipEventRecord(ev_out[b], s_out));
        }
        for (size_t k = chunks.size() > kBuf ? chunks.size() - kBuf : 0; k < chunks.size(); ++k) {
            const int rc = retire_out(k);
            if (rc) return rc;
        }
        return 0;
#undef MS_PIPE_TRY
    }
};

}  // namespace melspec
