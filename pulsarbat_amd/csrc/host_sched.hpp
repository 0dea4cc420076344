// host_sched.hpp -- the host-side ARITHMETIC of the streaming drivers and of the payload bounds check, free of any HIP
// dependency: plain C++ that compiles with g++ and runs on the CPU box under -fsanitize=address,undefined
// (tests/host_sched_test.cpp, tests/test_sanitizer.py; SURVEY.md 5 "Race detection / sanitizers").
// pbhip.hip's pbh_dedisperse_stream / pbh_dedisperse_stream_raw / pbh_decode execute exactly these schedules: every byte
// offset they hand to hipMemcpyAsync or to a kernel is computed here.
//
// Overlap-save streaming (BASELINE configs[3]; reference recipe: concatenate([coherent_dedispersion(z[k*hop : k*hop + N]) ...]),
// pulsarbat/transforms/transforms.py:59-148 + dedispersion.py:127-133): chunk k reads rows [k*hop, k*hop + N) of the stream.
// The device holds a WINDOW of N + (B-1)*hop consecutive rows; chunk k = e*B + j of epoch e reads rows [j*hop, j*hop + N) of
// its epoch's window and only the hop rows it adds are uploaded; two windows alternate between epochs, the first chunk of
// an epoch takes the N - hop rows it shares with its predecessor from the other window's tail.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/pbhip.h"

namespace pbh_host {

// ---- rows of equal size (pbh_dedisperse_stream) ---------------------------------------------------------------------
struct RowStream {
    int64_t N = 0, hop = 0, nchunk = 0, B = 1;   // rows per chunk, rows a chunk advances, chunks, chunks per epoch
    size_t row = 0;                              // bytes per row
    size_t step = 0, keep = 0;                   // row * hop; row * (N - hop): what consecutive chunks share
    size_t win_bytes = 0;                        // row * (N + (B - 1) * hop)
    int nwin = 1;                                // windows in use (2 as soon as there is a second epoch)
};

// chunks per epoch: as many as keep a window within `cap` bytes, at most 64, or exactly `forced` (1 ... 64) when set;
// a chunk must start on a 16-byte boundary of its window (the layout kernels' vector accesses), else one chunk per epoch
inline int64_t epoch_chunks(size_t first_bytes, size_t step_bytes, int64_t nchunk, size_t cap, int64_t forced) {
    int64_t B = 1;
    if (step_bytes == 0) B = 64;
    else if (cap > first_bytes) B = 1 + (int64_t)((cap - first_bytes) / step_bytes);
    B = B > 64 ? 64 : B;
    if (forced >= 1 && forced <= 64) B = forced;
    return B > nchunk ? nchunk : B;
}

// false: no chunk fits (total < N) or the geometry is degenerate
inline bool row_stream(int64_t N, int64_t hop, int64_t total_rows, size_t row, size_t cap, int64_t forced, RowStream* s) {
    if (N <= 0 || hop <= 0 || hop > N || total_rows < N || row == 0) return false;
    s->N = N;
    s->hop = hop;
    s->row = row;
    s->nchunk = (total_rows - N) / hop + 1;
    s->step = row * (size_t)hop;
    s->keep = row * (size_t)(N - hop);
    s->B = s->step % 16 == 0 ? epoch_chunks(row * (size_t)N, s->step, s->nchunk, cap, forced) : 1;
    s->win_bytes = row * (size_t)(N + (s->B - 1) * hop);
    s->nwin = s->nchunk > s->B ? 2 : 1;
    return true;
}

struct RowChunk {
    int64_t epoch = 0, j = 0;   // chunk k = epoch * B + j
    int win = 0;                // window the chunk computes in
    size_t win_off = 0;         // byte offset of the chunk's first row in that window (j * step)
    size_t up_src = 0;          // upload: host byte offset ...
    size_t up_dst = 0;          // ... to this byte offset of the window ...
    size_t up_bytes = 0;        // ... this many bytes
    bool handover = false;      // first chunk of an epoch after the first: copy the shared rows from the other window
    size_t ho_src = 0;          // ... from this byte offset of window win ^ 1 (to offset 0 of this one)
    size_t ho_bytes = 0;
};

inline RowChunk row_chunk(const RowStream& s, int64_t k) {
    RowChunk c;
    c.epoch = k / s.B;
    c.j = k - c.epoch * s.B;
    c.win = (int)(c.epoch & 1) % s.nwin;
    c.win_off = (size_t)c.j * s.step;
    // the rows this chunk adds: all N for the first chunk, afterwards rows [(k-1)*hop + N, k*hop + N) of the stream
    const size_t have = k == 0 ? 0 : s.keep;
    c.up_bytes = s.row * (size_t)s.N - have;
    c.up_src = (size_t)k * s.step + have;
    c.up_dst = c.win_off + have;
    c.handover = c.j == 0 && k > 0;
    c.ho_src = (size_t)s.B * s.step;   // the last chunk of the previous epoch sits at (B-1)*step; its rows from `hop` on
    c.ho_bytes = c.handover ? s.keep : 0;
    return c;
}

// ---- byte spans of unequal size (pbh_dedisperse_stream_raw: a chunk's samples live in bytes [off, off + len) of the file) -------
struct Span {
    int64_t b0 = 0, b1 = 0;   // first / last payload block touched
    size_t off = 0, len = 0;  // byte range of the raw buffer that is read
};

struct SpanStream {
    struct Epoch { int64_t k0; size_t base; };   // first chunk; file byte the window's byte 0 holds (16-byte aligned)
    std::vector<Span> spans;
    std::vector<Epoch> epochs;
    std::vector<int> epoch_of;
    size_t win_bytes = 0;
    int nwin = 1;
    size_t end_of(int64_t k) const { return spans[(size_t)k].off + spans[(size_t)k].len; }
    // chunk k extends what chunk k-1 left on the device when its span starts inside (or right behind) that span and ends no earlier
    bool extends(int64_t k) const {
        return k > 0 && spans[(size_t)k].off >= spans[(size_t)k - 1].off && spans[(size_t)k].off <= end_of(k - 1) && end_of(k) >= end_of(k - 1);
    }
};

// epochs: a new one when a chunk does not extend its predecessor, after `cap_chunks` chunks, or when the window would outgrow `cap`
inline void span_stream(std::vector<Span> spans, int64_t cap_chunks, size_t cap, SpanStream* s) {
    s->spans = std::move(spans);
    s->epochs.clear();
    const int64_t n = (int64_t)s->spans.size();
    s->epoch_of.assign((size_t)n, 0);
    s->win_bytes = 0;
    for (int64_t k = 0; k < n; ++k) {
        const bool fresh = s->epochs.empty() || !s->extends(k) || k - s->epochs.back().k0 >= cap_chunks ||
                           s->end_of(k) - s->epochs.back().base > cap;
        if (fresh) s->epochs.push_back({k, s->spans[(size_t)k].off - s->spans[(size_t)k].off % 16});
        s->epoch_of[(size_t)k] = (int)s->epochs.size() - 1;
        const size_t need = s->end_of(k) - s->epochs.back().base;
        s->win_bytes = need > s->win_bytes ? need : s->win_bytes;
    }
    s->nwin = s->epochs.size() > 1 ? 2 : 1;
}

struct SpanChunk {
    int epoch = 0, win = 0;
    bool head = false;          // first chunk of its epoch
    size_t base = 0;            // file byte at the window's byte 0
    size_t up_lo = 0, up_hi = 0;  // file bytes [up_lo, up_hi) are new to the device: to window offset up_lo - base
    bool handover = false;      // head of an epoch that re-uses bytes: [base, up_lo) come out of the other window ...
    size_t ho_src = 0;          // ... from this offset of it
    size_t ho_bytes = 0;
};

inline SpanChunk span_chunk(const SpanStream& s, int64_t k) {
    SpanChunk c;
    c.epoch = s.epoch_of[(size_t)k];
    const SpanStream::Epoch& ep = s.epochs[(size_t)c.epoch];
    c.head = k == ep.k0;
    c.win = (c.epoch & 1) % s.nwin;
    c.base = ep.base;
    const bool reuse = s.extends(k);
    c.up_lo = reuse ? s.end_of(k - 1) : s.spans[(size_t)k].off;
    c.up_hi = s.end_of(k);
    if (c.up_hi < c.up_lo) c.up_hi = c.up_lo;
    if (c.head && k > 0 && reuse && c.up_lo > ep.base) {
        c.handover = true;
        c.ho_src = ep.base - s.epochs[(size_t)c.epoch - 1].base;
        c.ho_bytes = c.up_lo - ep.base;
    }
    return c;
}

// ---- payload bounds (pbh_decode, pbh_dedisperse_stream_raw): which bytes of the raw buffer do samples [first, first + nsample) touch? --
// Returns PBH_OK or an error code with *why set.  Every product of caller-supplied strides, counts and sizes is checked: a layout
// whose addressing does not fit 63 bits is rejected, not wrapped (the bounds checks are only as good as the arithmetic under them).
inline int decode_span(const pbh_raw_layout_t* L, int64_t first, int64_t nsample, int nchan, int npol, size_t raw_bytes, Span* sp,
                       const char** why) {
    auto bad = [&](int code, const char* msg) { if (why) *why = msg; return code; };
    if (!L) return bad(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || first < 0 || nchan <= 0 || npol <= 0) return bad(PBH_ERR_INVALID, "bad dimensions");
    if (L->ncomp != 1 && L->ncomp != 2) return bad(PBH_ERR_INVALID, "ncomp must be 1 or 2");
    if (!((L->nbits == 8 && (L->code == 0 || L->code == 1)) || ((L->nbits == 2 || L->nbits == 4) && L->code == 0)))
        return bad(PBH_ERR_UNSUPPORTED, "payload coding: 8 bits (code 0/1), 4 bits or 2 bits");
    if (L->blk_samples <= 0 || L->blk_stride < 0 || L->hdr_bytes < 0) return bad(PBH_ERR_INVALID, "bad block geometry");
    if ((int64_t)nchan * npol > 65535LL * 64) return bad(PBH_ERR_UNSUPPORTED, "too many series");
    if (first > INT64_MAX - nsample) return bad(PBH_ERR_INVALID, "sample range overflows");
    const int64_t b0 = first / L->blk_samples, b1 = (first + nsample - 1) / L->blk_samples;
    bool ovf = false;
    auto mul = [&](int64_t a, int64_t b) { int64_t r = 0; ovf |= __builtin_mul_overflow(a, b, &r); return r; };
    auto add = [&](int64_t a, int64_t b) { int64_t r = 0; ovf |= __builtin_add_overflow(a, b, &r); return r; };
    // [lo, hi]: element indices reached over time samples [ta, tb] of one payload
    auto reach = [&](int64_t ta, int64_t tb, int64_t* lo, int64_t* hi) {
        *lo = *hi = L->elem0;
        auto span = [&](int64_t stride, int64_t a, int64_t b) {
            *lo = add(*lo, stride >= 0 ? mul(stride, a) : mul(stride, b));
            *hi = add(*hi, stride >= 0 ? mul(stride, b) : mul(stride, a));
        };
        span(L->stride_t, ta, tb);
        span(L->stride_c, 0, nchan - 1);
        span(L->stride_p, 0, npol - 1);
    };
    const int64_t bits = (int64_t)L->nbits * L->ncomp;
    const int64_t t_first = first - b0 * L->blk_samples, t_last = first + nsample - 1 - b1 * L->blk_samples;
    int64_t lo, hi, lo2, hi2;
    reach(b1 > b0 ? 0 : t_first, t_last, &lo, &hi);   // the last block: what bounds the buffer
    if (ovf || hi < 0 || hi > (INT64_MAX - 8) / bits - 1) return bad(PBH_ERR_INVALID, "payload addressing overflows");
    const int64_t pay_hi = ((hi + 1) * bits + 7) / 8;
    if (b1 > b0) {                                      // earlier blocks are read up to their last sample
        reach(b1 > b0 + 1 ? 0 : t_first, L->blk_samples - 1, &lo2, &hi2);
        if (b1 > b0 + 1) {
            int64_t lo3, hi3;
            reach(t_first, L->blk_samples - 1, &lo3, &hi3);
            lo2 = lo3 < lo2 ? lo3 : lo2;
        }
        if (ovf || hi2 < 0 || hi2 > (INT64_MAX - 8) / bits - 1) return bad(PBH_ERR_INVALID, "payload addressing overflows");
        if (add(L->hdr_bytes, ((hi2 + 1) * bits + 7) / 8) > L->blk_stride || ovf)
            return bad(PBH_ERR_INVALID, "payload addressing overruns a block");
        lo = lo2 < lo ? lo2 : lo;
    }
    if (lo < 0) return bad(PBH_ERR_INVALID, "payload addressing reaches before the payload");
    const int64_t end_byte = add(add(mul(b1, L->blk_stride), L->hdr_bytes), pay_hi);
    if (ovf) return bad(PBH_ERR_INVALID, "payload addressing overflows");
    if ((uint64_t)end_byte > (uint64_t)raw_bytes) return bad(PBH_ERR_INVALID, "raw buffer too short for the requested samples");
    sp->b0 = b0;
    sp->b1 = b1;
    // the range starts at the lowest element read in the first block (a long time-major payload is one block: only the
    // wanted samples travel), rounded down to 16 bytes
    int64_t lo_first, hi_first;
    reach(t_first, b1 > b0 ? L->blk_samples - 1 : t_last, &lo_first, &hi_first);
    if (ovf) return bad(PBH_ERR_INVALID, "payload addressing overflows");
    const int64_t skip = (L->hdr_bytes + lo_first * bits / 8) & ~(int64_t)15;
    sp->off = (size_t)(b0 * L->blk_stride + skip);
    sp->len = (size_t)end_byte - sp->off;
    return PBH_OK;
}

// ---- pitched slices (pbh_dedisperse_slices: the last pass writes a rank's channel slice into a wider block, cut into row parts) ------
// part i holds output rows [part_row[i], part_row[i+1]); row r of the result goes to part_ptr[i] + ((r - part_row[i]) * row_elems
// + col_offset) elements.  Checks the part table; returns the index of the part holding row r (or -1).
inline bool slice_parts_ok(int nparts, const int64_t* part_row, int64_t nout, int64_t row_elems, int64_t col_offset, int64_t ncol) {
    if (nparts <= 0 || !part_row || part_row[0] != 0 || part_row[nparts] != nout) return false;
    for (int i = 0; i < nparts; ++i)
        if (part_row[i + 1] < part_row[i]) return false;
    return col_offset >= 0 && ncol >= 0 && row_elems >= 0 && col_offset + ncol <= row_elems;
}
inline int slice_part_of(int nparts, const int64_t* part_row, int64_t r) {
    for (int i = 0; i < nparts; ++i)
        if (r >= part_row[i] && r < part_row[i + 1]) return i;
    return -1;
}

}  // namespace pbh_host
