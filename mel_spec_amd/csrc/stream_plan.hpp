// stream_plan.hpp -- host bookkeeping of the streaming bank (plain C++, no HIP): which frames a push emits
// and where they start, restating Spectrogram::add (src/stft.rs:48-86) driven hop by hop the way
// RingBuffer::maybe_mel does (src/rb.rs:86-121).  Shared by aux.hip and tests/emu.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace melspec {

// Slot layout: [carry, right-aligned so that it ends at in_off][next chunk at in_off, <= max_chunk (+ < hop zeros on flush)].
struct StreamGeom {
    uint32_t n_fft, hop, n_mels, n_streams, max_chunk;
    uint32_t in_off;      // carry capacity: >= n_fft - 1 (history n_fft - hop, pending < hop), multiple of 4
    uint64_t stride;      // floats per slot
};

inline StreamGeom stream_geometry(uint32_t n_fft, uint32_t hop, uint32_t n_mels, uint32_t n_streams, uint32_t max_chunk) {
    StreamGeom g{};
    g.n_fft = n_fft; g.hop = hop; g.n_mels = n_mels; g.n_streams = n_streams; g.max_chunk = max_chunk;
    g.in_off = (n_fft + 3u) & ~3u;
    g.stride = (static_cast<uint64_t>(g.in_off) + max_chunk + hop + 7) & ~static_cast<uint64_t>(3);
    return g;
}

struct StreamBook {
    std::vector<uint32_t> pending;       // RingBuffer::accumulated_samples.len()
    std::vector<uint64_t> idx;           // Spectrogram::idx
    std::vector<uint32_t> mark;          // duplicate detection inside one push
    uint32_t epoch = 0;
    void reset(uint32_t n) { pending.assign(n, 0); idx.assign(n, 0); mark.assign(n, 0); epoch = 0; }
};

struct StreamEntry {      // one per pushed stream; also read by the scatter / carry kernels
    uint32_t stream;      // slot index
    uint32_t len;         // samples of the new chunk
    uint32_t keep;        // carry length after this push
    uint32_t zero_fill;   // flush: zeros appended after the pending samples
    uint64_t src_off;     // offset of the chunk in the flat staging buffer (host pushes)
    uint64_t out_off;     // first float of the rows this push emits for the stream ([frame][n_mels], read by the VAD stage)
    uint32_t frames;      // rows emitted
    uint32_t act_base;    // index of the stream's first activity record of this push (records are packed in entry order)
};

struct StreamPlan {
    std::vector<StreamEntry> entries;
    std::vector<uint64_t> off, len, out_off;   // ragged batch over the state buffer: sample offset / length, output offset
    std::vector<uint32_t> frames;
    uint64_t total_frames = 0;
};

// adds this push would make (h) and how many of the first ones return None (skip)
inline void stream_count(const StreamGeom &g, uint32_t pend, uint64_t idx, uint32_t len, bool flush, uint32_t &h, uint32_t &skip) {
    uint64_t idx_after_first;
    if (flush) {                         // add() with pcm_size < hop: zero-padded, idx += pcm_size (src/stft.rs:57-66)
        h = pend ? 1 : 0;
        idx_after_first = idx + pend;
    } else {
        h = (pend + len) / g.hop;
        idx_after_first = idx + g.hop;
    }
    skip = 0;                            // add j (0-based) emits iff idx_after_first + j*hop >= n_fft (src/stft.rs:68)
    if (idx_after_first < g.n_fft) {
        const uint64_t need = (g.n_fft - idx_after_first + g.hop - 1) / g.hop;
        skip = need < h ? static_cast<uint32_t>(need) : h;
    }
}

// returns 0 ok, 1 invalid argument, 2 capacity; *err names the problem
inline int stream_plan_push(const StreamGeom &g, StreamBook &bk, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush,
                            StreamPlan &pl, const char **err) {
    pl.entries.resize(n); pl.off.resize(n); pl.len.resize(n); pl.out_off.resize(n); pl.frames.resize(n);
    if (++bk.epoch == 0) { std::fill(bk.mark.begin(), bk.mark.end(), 0u); bk.epoch = 1; }
    uint64_t cursor = 0, src_cursor = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t s = ids[i];
        if (s >= g.n_streams) { *err = "stream id out of range"; return 1; }
        if (bk.mark[s] == bk.epoch) { *err = "a stream may appear only once per push"; return 1; }
        bk.mark[s] = bk.epoch;
        const uint32_t len = flush ? 0 : lens[i];
        if (len > g.max_chunk) { *err = "chunk longer than max_chunk"; return 2; }
        const uint32_t pend = bk.pending[s];
        const uint32_t c = g.n_fft - g.hop + pend;                   // current carry
        uint32_t h, skip;
        stream_count(g, pend, bk.idx[s], len, flush, h, skip);
        const uint32_t frames = h - skip;
        pl.frames[i] = frames;
        pl.off[i] = static_cast<uint64_t>(s) * g.stride + g.in_off - c + static_cast<uint64_t>(skip) * g.hop;
        pl.len[i] = frames ? static_cast<uint64_t>(frames - 1) * g.hop + g.n_fft : 0;
        pl.out_off[i] = cursor;
        cursor += static_cast<uint64_t>(frames) * g.n_mels;
        StreamEntry &e = pl.entries[i];
        e.stream = s; e.len = len; e.src_off = src_cursor;
        e.zero_fill = flush && pend ? g.hop - pend : 0;
        e.keep = g.n_fft - g.hop + (flush ? 0 : pend + len - h * g.hop);
        e.out_off = pl.out_off[i]; e.frames = frames; e.act_base = static_cast<uint32_t>(g.n_mels ? pl.out_off[i] / g.n_mels : 0);
        src_cursor += len;
    }
    pl.total_frames = g.n_mels ? cursor / g.n_mels : 0;
    return 0;
}

inline void stream_commit_push(const StreamGeom &g, StreamBook &bk, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush) {
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t s = ids[i];
        if (flush) { bk.idx[s] += bk.pending[s]; bk.pending[s] = 0; continue; }
        const uint32_t tot = bk.pending[s] + lens[i], h = tot / g.hop;
        bk.idx[s] += static_cast<uint64_t>(h) * g.hop;
        bk.pending[s] = tot - h * g.hop;
    }
}

inline size_t stream_frames_after(const StreamGeom &g, const StreamBook &bk, uint32_t id, uint32_t n_new) {
    uint32_t h, skip;
    stream_count(g, bk.pending[id], bk.idx[id], n_new, false, h, skip);
    return h - skip;
}

}  // namespace melspec
