// host_sched_test.cpp -- the host-side arithmetic of the streaming drivers and of the payload bounds check, executed on the CPU
// under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizer.py builds and runs this).
// Every schedule csrc/host_sched.hpp produces is EXECUTED here with memcpy on exactly-sized heap buffers (an off-by-one is an
// ASan report) and its result compared with the reference's definition of the stream: chunk k holds rows [k hop, k hop + N).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../pulsarbat_amd/csrc/host_sched.hpp"

using namespace pbh_host;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void test_row_stream(std::mt19937_64& rng) {
    int cases = 0, multi_epoch = 0, short_last_epoch = 0;
    for (int it = 0; it < 4000; ++it) {
        const int64_t N = 1 + (int64_t)(rng() % 200);
        const int64_t hop = 1 + (int64_t)(rng() % N);
        const int64_t total = N + (int64_t)(rng() % (12 * N));
        const size_t row = (size_t[]){8, 16, 24, 32, 48, 128, 136}[rng() % 7];
        const size_t cap = (size_t)(rng() % (row * N * 6 + 1));
        const int64_t forced = (rng() % 3 == 0) ? (int64_t)(rng() % 7) : 0;
        RowStream s;
        if (!row_stream(N, hop, total, row, cap, forced, &s)) { CHECK(false, "row_stream rejected N %lld hop %lld", (long long)N, (long long)hop); continue; }
        ++cases;
        CHECK(s.nchunk == (total - N) / hop + 1, "nchunk");
        CHECK(s.B >= 1 && s.B <= 64 && s.B <= s.nchunk, "B = %lld", (long long)s.B);
        if (s.step % 16 != 0) CHECK(s.B == 1, "unaligned chunk starts need one chunk per epoch");
        std::vector<unsigned char> host((size_t)total * row);
        for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned char)(i * 2654435761u >> 13);
        std::vector<unsigned char> win[2];
        for (int w = 0; w < s.nwin; ++w) win[w].assign(s.win_bytes, 0xEE);
        size_t uploaded = 0;
        for (int64_t k = 0; k < s.nchunk; ++k) {
            const RowChunk c = row_chunk(s, k);
            CHECK(c.win >= 0 && c.win < s.nwin, "window index");
            if (c.handover) {
                CHECK(s.nwin == 2, "a hand-over needs two windows");
                if (c.ho_bytes) std::memcpy(win[c.win].data(), win[c.win ^ 1].data() + c.ho_src, c.ho_bytes);   // ASan: both in bounds
            }
            std::memcpy(win[c.win].data() + c.up_dst, host.data() + c.up_src, c.up_bytes);
            uploaded += c.up_bytes;
            // what the kernels then read: rows [k hop, k hop + N) of the stream at win_off
            CHECK(c.win_off + (size_t)N * row <= s.win_bytes, "chunk reaches beyond its window");
            CHECK(std::memcmp(win[c.win].data() + c.win_off, host.data() + (size_t)k * s.step, (size_t)N * row) == 0,
                  "chunk %lld of N %lld hop %lld B %lld does not hold its rows", (long long)k, (long long)N, (long long)hop, (long long)s.B);
        }
        // every row crosses the link once: N rows for the first chunk, hop for each later one
        CHECK(uploaded == row * (size_t)(N + (s.nchunk - 1) * hop), "upload volume");
        if (s.nchunk > s.B) ++multi_epoch;
        if (s.nchunk % s.B) ++short_last_epoch;
    }
    printf("row streams: %d cases (%d with several epochs, %d with a short last epoch)\n", cases, multi_epoch, short_last_epoch);
    RowStream s;
    CHECK(!row_stream(16, 4, 15, 8, 1 << 20, 0, &s), "a stream shorter than one chunk");
    CHECK(!row_stream(16, 0, 64, 8, 1 << 20, 0, &s), "hop 0");
    CHECK(!row_stream(16, 17, 64, 8, 1 << 20, 0, &s), "hop > N");
}

static void test_span_stream(std::mt19937_64& rng) {
    int cases = 0, handovers = 0;
    for (int it = 0; it < 3000; ++it) {
        const int n = 1 + (int)(rng() % 40);
        std::vector<Span> spans((size_t)n);
        size_t off = rng() % 100, file_end = 0;
        for (int k = 0; k < n; ++k) {
            const size_t len = 1 + rng() % 400;
            spans[(size_t)k].off = off;
            spans[(size_t)k].len = len;
            file_end = off + len > file_end ? off + len : file_end;
            const int kind = (int)(rng() % 10);
            if (kind < 7) off += rng() % (len + 1);            // overlaps or abuts its predecessor
            else if (kind < 9) off += len + 1 + rng() % 50;    // a gap: a fresh epoch
            else off = off > 60 ? off - rng() % 60 : off;      // goes back: a fresh epoch
        }
        std::vector<unsigned char> file(file_end);
        for (size_t i = 0; i < file.size(); ++i) file[i] = (unsigned char)(i * 40503u >> 7);
        SpanStream s;
        span_stream(spans, 1 + (int64_t)(rng() % 8), (size_t)(64 + rng() % 2000), &s);
        ++cases;
        std::vector<unsigned char> win[2];
        for (int w = 0; w < s.nwin; ++w) win[w].assign(s.win_bytes + 16, 0xEE);   // (+16: the driver's allocation)
        for (int64_t k = 0; k < n; ++k) {
            const SpanChunk c = span_chunk(s, k);
            CHECK(c.base % 16 == 0 && c.base <= spans[(size_t)k].off, "window base");
            if (c.handover) {
                ++handovers;
                std::memcpy(win[c.win].data(), win[c.win ^ 1].data() + c.ho_src, c.ho_bytes);
            }
            if (c.up_hi > c.up_lo) std::memcpy(win[c.win].data() + (c.up_lo - c.base), file.data() + c.up_lo, c.up_hi - c.up_lo);
            const Span& sp = spans[(size_t)k];
            CHECK(sp.off - c.base + sp.len <= s.win_bytes, "span beyond the window");
            CHECK(std::memcmp(win[c.win].data() + (sp.off - c.base), file.data() + sp.off, sp.len) == 0, "chunk %lld does not hold its bytes", (long long)k);
        }
    }
    printf("span streams: %d cases, %d hand-overs between windows\n", cases, handovers);
}

static void test_decode_span(std::mt19937_64& rng) {
    int ok = 0, rejected = 0;
    for (int it = 0; it < 20000; ++it) {
        pbh_raw_layout_t L;
        std::memset(&L, 0, sizeof L);
        L.nbits = (int[]){2, 4, 8}[rng() % 3];
        L.ncomp = 1 + (int)(rng() % 2);
        L.code = L.nbits == 8 ? (int)(rng() % 2) : 0;
        const int nchan = 1 + (int)(rng() % 5), npol = 1 + (int)(rng() % 2);
        L.blk_samples = 1 + (int64_t)(rng() % 64);
        // three addressing families: time-major, channel-major (GUPPI-like), reversed channels
        const int fam = (int)(rng() % 3);
        if (fam == 0) { L.stride_p = 1; L.stride_c = npol; L.stride_t = (int64_t)nchan * npol; L.elem0 = 0; }
        else if (fam == 1) { L.stride_p = 1; L.stride_t = npol; L.stride_c = L.blk_samples * npol; L.elem0 = 0; }
        else { L.stride_p = 1; L.stride_c = -(int64_t)npol; L.stride_t = (int64_t)nchan * npol; L.elem0 = (int64_t)(nchan - 1) * npol; }
        const int64_t bits = (int64_t)L.nbits * L.ncomp;
        const int64_t payload = (L.blk_samples * nchan * npol * bits + 7) / 8;
        L.hdr_bytes = (int64_t)(rng() % 40);
        L.blk_stride = L.hdr_bytes + payload + (int64_t)(rng() % 24);
        const int64_t nblk = 1 + (int64_t)(rng() % 6);
        size_t raw_bytes = (size_t)(nblk * L.blk_stride);
        if (rng() % 5 == 0) raw_bytes -= rng() % (raw_bytes / 2 + 1);       // a buffer that may be too short
        const int64_t first = (int64_t)(rng() % (nblk * L.blk_samples));
        const int64_t nsample = 1 + (int64_t)(rng() % (nblk * L.blk_samples));
        Span sp;
        const char* why = nullptr;
        const int rc = decode_span(&L, first, nsample, nchan, npol, raw_bytes, &sp, &why);
        // brute force: every byte the decode touches
        size_t lo = SIZE_MAX, hi = 0;
        bool in_blocks = true;
        for (int64_t t = first; t < first + nsample; ++t) {
            const int64_t b = t / L.blk_samples, tb = t % L.blk_samples;
            for (int c = 0; c < nchan; ++c)
                for (int p = 0; p < npol; ++p) {
                    const int64_t e = L.elem0 + L.stride_t * tb + L.stride_c * c + L.stride_p * p;
                    if (e < 0) { in_blocks = false; continue; }
                    const int64_t b0 = b * L.blk_stride + L.hdr_bytes + e * bits / 8;
                    const int64_t b1 = b * L.blk_stride + L.hdr_bytes + ((e + 1) * bits + 7) / 8;
                    lo = (size_t)b0 < lo ? (size_t)b0 : lo;
                    hi = (size_t)b1 > hi ? (size_t)b1 : hi;
                }
        }
        if (rc == PBH_OK) {
            ++ok;
            CHECK(in_blocks, "accepted a layout that reaches before its payload");
            CHECK(sp.off <= lo && sp.off + sp.len >= hi, "span [%zu, %zu) misses touched bytes [%zu, %zu)", sp.off, sp.off + sp.len, lo, hi);
            CHECK(sp.off + sp.len <= raw_bytes, "span beyond the buffer");
            CHECK(sp.off >= (size_t)(sp.b0 * L.blk_stride) && (sp.off - (size_t)(sp.b0 * L.blk_stride)) % 16 == 0, "span start: a multiple of 16 bytes into its block");
        } else {
            ++rejected;
            CHECK(why != nullptr, "an error without a message");
            CHECK(hi > raw_bytes || !in_blocks, "rejected a request whose bytes [%zu, %zu) fit the %zu-byte buffer: %s", lo, hi, raw_bytes, why ? why : "?");
        }
    }
    printf("decode spans: %d accepted, %d rejected\n", ok, rejected);
    // addressing that does not fit 63 bits is an error, not a wrap (UBSan would report the overflow otherwise)
    pbh_raw_layout_t L;
    std::memset(&L, 0, sizeof L);
    L.nbits = 8; L.ncomp = 2; L.blk_samples = 1LL << 40; L.blk_stride = INT64_MAX / 2; L.stride_t = INT64_MAX / 3; L.stride_c = 1LL << 61; L.stride_p = 1;
    Span sp;
    const char* why = nullptr;
    CHECK(decode_span(&L, (1LL << 39), 1LL << 39, 4, 2, SIZE_MAX, &sp, &why) != PBH_OK, "overflowing strides accepted");
    CHECK(decode_span(&L, INT64_MAX - 5, 10, 1, 1, SIZE_MAX, &sp, &why) != PBH_OK, "overflowing sample range accepted");
    CHECK(decode_span(nullptr, 0, 1, 1, 1, 16, &sp, &why) != PBH_OK, "NULL layout accepted");
}

static void test_slices(std::mt19937_64& rng) {
    for (int it = 0; it < 2000; ++it) {
        const int64_t nout = (int64_t)(rng() % 500), row_elems = 1 + (int64_t)(rng() % 64);
        const int64_t ncol = (int64_t)(rng() % (row_elems + 1)), col = (int64_t)(rng() % (row_elems - ncol + 1));
        const int nparts = 1 + (int)(rng() % 6);
        std::vector<int64_t> pr((size_t)nparts + 1, 0);
        for (int i = 1; i < nparts; ++i) pr[(size_t)i] = (int64_t)(rng() % (nout + 1));
        pr[(size_t)nparts] = nout;
        std::sort(pr.begin(), pr.end());
        CHECK(slice_parts_ok(nparts, pr.data(), nout, row_elems, col, ncol), "a valid part table rejected");
        // write every element of the slice into exactly-sized parts
        std::vector<std::vector<unsigned char>> parts((size_t)nparts);
        for (int i = 0; i < nparts; ++i) parts[(size_t)i].assign((size_t)((pr[(size_t)i + 1] - pr[(size_t)i]) * row_elems), 0);
        for (int64_t r = 0; r < nout; ++r) {
            const int i = slice_part_of(nparts, pr.data(), r);
            CHECK(i >= 0, "row %lld in no part", (long long)r);
            if (i < 0) continue;
            for (int64_t c = 0; c < ncol; ++c) parts[(size_t)i][(size_t)((r - pr[(size_t)i]) * row_elems + col + c)] += 1;   // ASan: in bounds
        }
        CHECK(!slice_parts_ok(nparts, pr.data(), nout + 1, row_elems, col, ncol), "a table that does not end at nout accepted");
        CHECK(!slice_parts_ok(nparts, pr.data(), nout, row_elems, row_elems, 1), "a column range beyond the row accepted");
    }
}

int main() {
    std::mt19937_64 rng(20260004);
    test_row_stream(rng);
    test_span_stream(rng);
    test_decode_span(rng);
    test_slices(rng);
    if (failures) { printf("%d check(s) FAILED\n", failures); return 1; }
    printf("host_sched: all checks passed\n");
    return 0;
}
