// pbhip.hip -- host side of libpbhip.so, unit 1 of 3: plans, launch sequences, chirp, and the entry points of the hot path, of the
// next rows and of the stand-alone transforms (pbhip_internal.hpp lists the units).  Compiled twice (float32, and float64 with
// -DPBH_F64): every public name is renamed to the per-precision prefix (pbh32_* / pbh64_*); pbhip_api.cpp owns the public pbh_*
// symbols and dispatches on the plan's dtype.
#include "pbhip_internal.hpp"

#include "aux_kernels.hpp"
#include "kernels.hpp"
#include "mixed_kernels.hpp"
#include "fd4_kernels.hpp"

namespace PBH_NS {
thread_local std::string g_err;

static int ilog2(int64_t x) {
    int l = 0;
    while ((1LL << l) < x) ++l;
    return l;
}
static bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }
static int64_t next_pow2_at_least(int64_t x) {
    int64_t L = PBH_R;
    while (L < x) L <<= 1;
    return L;
}
// Lengths m * 2^k with m in {3, 5, 7} run natively: the odd factor becomes the radix-P stage of the split
// column transform (k_radix_p / k_deint_radix), 2^k = Q * 2^tile with PBH_R <= Q <= one-line-wide tiles.
// PBH_ODD=0 turns this off (such lengths then go through the convolution plan like any other).
static int native_odd_factor(int64_t n) {
    static const bool on = [] { const char* e = diag_env("PBH_ODD"); return e ? atoi(e) != 0 : true; }();
    if (!on) return 0;
    const int64_t qmax = (int64_t)kTilePoints * (int64_t)sizeof(cf) / 128;
    for (int m : {3, 5, 7}) {
        if (n % m) continue;
        const int64_t two = n / m;
        if (!is_pow2(two) || two % kTilePoints) continue;
        const int64_t q = two / kTilePoints;
        if (q >= PBH_R && q <= qmax) return m;
    }
    return 0;
}
// Shortest native length >= x for the convolution plan of an arbitrary-length transform: 2^k or m * 2^k.
static int64_t convolution_length(int64_t x) {
    int64_t best = next_pow2_at_least(x);
    for (int m : {3, 5, 7}) {
        int64_t two = kTilePoints * (int64_t)PBH_R;
        while (m * two < x) two <<= 1;
        const int64_t L = m * two;
        if (L < best && native_odd_factor(L) == m) best = L;
    }
    return best;
}

// 7-smooth lengths that are neither 2^k nor native m * 2^k: N = P * Q * N2 with N2 = 2^min(k, tile) >= 32 the row length
// and P, Q <= kMixMaxLen (P = 1 when N / N2 fits one column tile).  PBH_MIXED=0 sends them through the convolution plan.
static bool is_7smooth(int64_t n) {
    for (int f : {2, 3, 5, 7}) while (n % f == 0) n /= f;
    return n == 1;
}
static bool split_levels(int64_t n1, int64_t* qout) {
    int64_t q = 0;
    static const int64_t forced = [] { const char* e = diag_env("PBH_MIX_Q"); return e ? atoll(e) : 0LL; }();   // (timing runs)
    if (n1 <= kMixMaxLen) q = n1;
    else if (forced > 1 && n1 % forced == 0 && forced <= kMixMaxLen && n1 / forced <= kMixMaxLen) q = forced;
    else {
        // two levels: both factors within 512 rows if there is such a split (tiles of whole 128-byte lines in both passes;
        // beyond 512 rows a tile holds 64-byte pieces), the larger one as the Q-point pass.  (Round 3 measured the most
        // balanced split instead on 16 lengths: 0.86-1.09x, median 0.96x -- profiles/r03_mix_split.txt; what was slow about
        // the unbalanced ones was the tile kernel on 3 and 5 rows: mix_radix_p.)
        for (int64_t d = 512; d >= 2 && !q; --d)
            if (n1 % d == 0 && n1 / d <= 512) q = d;
        for (int64_t d = kMixMaxLen; d >= 2 && !q; --d)
            if (n1 % d == 0 && n1 / d <= kMixMaxLen) q = d;
    }
    *qout = q;
    return q != 0;
}
static bool mixed_geometry(int64_t n, int* N1, int* N2, int* P) {
    // PBH_MIXED: 0 = off, 1 = only lengths whose N1 fits one column pass (P = 1: 5 passes, 1.5x the rate of the padded
    // convolution), 2 (default) = two-level lengths as well (7 passes: 1.06-1.14x the convolution plan, half its memory)
    static const int mode = [] { const char* e = getenv("PBH_MIXED"); return e ? atoi(e) : 2; }();
    if (!mode || n < 64 || is_pow2(n) || !is_7smooth(n)) return false;
    int k = 0;
    while (((n >> k) & 1) == 0) ++k;
    if (k > kTileLog2) k = kTileLog2;
    if (k < 5) return false;                 // rows shorter than 32 points: a column tile would not hold whole lines
    const int64_t n1 = n >> k;
    if (n1 < 2) return false;
    int64_t q = 0;
    if (!split_levels(n1, &q)) return false;
    if (mode < 2 && n1 / q > 1) return false;
    *N2 = 1 << k;
    *N1 = (int)n1;
    *P = (int)(n1 / q);
    return true;
}

// 7-smooth lengths with at most four factors of two: no power-of-two rows of 32 points, so the rows are mixed-radix too
// (k_rowmix): N2 = 2^k * f, f odd, at most 1024 points, as long as possible such that N1 = N / N2 still splits into column
// levels of at most 1024 rows.  The column passes take pieces of up to 512 elements of a row, the last one of a row short
// (rows start at multiples of 2^k elements: 16 to 128 bytes).  PBH_ROWMIX=0: off.
static bool rowmix_geometry(int64_t n, int* N1, int* N2, int* P) {
    static const bool on = [] { const char* e = diag_env("PBH_ROWMIX"); return e ? atoi(e) != 0 : true; }();
    static const int mode = [] { const char* e = getenv("PBH_MIXED"); return e ? atoi(e) : 2; }();
    if (!on || mode < 2 || n < 4096 || is_pow2(n) || !is_7smooth(n)) return false;
    int k = 0;
    while (((n >> k) & 1) == 0) ++k;
    // (odd lengths too: their planar rows start at odd element offsets, which costs the layout kernels unaligned 16-byte
    //  accesses and nothing else; PBH_ROWMIX_KMIN=1 leaves them to the convolution plan)
    static const int kmin = [] { const char* e = diag_env("PBH_ROWMIX_KMIN"); return e ? atoi(e) : 0; }();
    if (k < kmin) return false;
    if (k > 4) k = 4;
    // the longest row that leaves a splittable N1 (measured on 10 935 000 x 16: rows of 1000 points 5.03 ms, 600: 5.06,
    // 360: 5.28, 200: 5.29, 72: 5.26 -- the row pass gets slower with its stage count, the column levels faster)
    const int64_t two = 1LL << k, odd = n >> k;
    for (int64_t f = 1024 / two; f >= 3; --f) {
        if (!(f & 1) || odd % f) continue;
        const int64_t n2 = two * f, n1 = n / n2;
        int64_t q;
        if (n1 < 2 || !split_levels(n1, &q)) continue;
        *N2 = (int)n2;
        *N1 = (int)n1;
        *P = (int)(n1 / q);
        return true;
    }
    return false;
}

// The seven passes of a plan with mixed-radix rows cost 0.43 ms per 2^20 samples x 16 series; a convolution plan's three
// transform passes run on the padded length L and its layout passes on N: it wins when L is a power of two (the fastest
// plans there are) within 2.2 N (8 268 750: 3.30 against 3.89 ms; 7 873 200: 3.29 / 3.35), and loses with L = m * 2^k
// (12 301 875: 5.82 / 5.70; 7 144 200: 3.65 / 3.09).  profiles/r02_7smooth.txt.  PBH_ROWMIX=2: always the mixed-radix rows.
static bool rowmix_pays(int64_t n) {
    static const int mode = [] { const char* e = diag_env("PBH_ROWMIX"); return e ? atoi(e) : 1; }();
    if (mode >= 2) return true;
    const int64_t L = convolution_length(2 * n - 1);
    return !is_pow2(L) || (double)L / (double)n >= 2.2;
}

// ---- plan: struct pbh_plan, Step, DetectTail, IoLayout are in pbhip_internal.hpp ----------------------------------------
int dev_alloc(pbh_plan* p, void** ptr, size_t bytes) {
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess)
        return fail(PBH_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e));
    if (p) p->owned_bytes += (int64_t)bytes;
    static const bool trace = getenv("PBH_TRACE_ALLOC") != nullptr;  // debugging aid: where the library's buffers are
    if (trace) fprintf(stderr, "[pbhip] alloc %p .. %p (%zu bytes)\n", *ptr, (char*)*ptr + bytes, bytes);
    return PBH_OK;
}

static int resolved_variant(const pbh_plan* p) {
    if (p->N1 == 1) return PBH_VARIANT_DIRECT3;  // single tile: no passes to choose
    if (p->mixed) return PBH_VARIANT_PLANAR5;    // 7-smooth lengths: planar pipeline with mixed-radix column passes
    if (p->P > 1) return PBH_VARIANT_PLANAR5;    // long blocks: only the planar pipeline has the split column pass
    if (p->variant != PBH_VARIANT_AUTO) return p->variant;
    // direct3 touches full 128-B lines only when a 16-column tile spans whole time samples of few
    // series; with many interleaved series its planar side degenerates to 8-byte pieces
    // (beyond N1 = 2^tile/16 its 16-column tile is no longer whole lines on the interleaved side either: with two
    //  series at N = 2^25 the planar pipeline is faster, 52 vs 45 Gsamples/s)
    if (p->S == 1) return PBH_VARIANT_DIRECT3;
    return (p->S <= 2 && p->N1 * 16 <= kTilePoints) ? PBH_VARIANT_DIRECT3 : PBH_VARIANT_PLANAR5;
}

// ---- kernel dispatch by FFT length ---------------------------------------------------------------------------
static int block_lane_order() {
    static int r = [] {
        const char* e = diag_env("PBH_BLOCK_LANE_ORDER");
        return e ? atoi(e) : 0;
    }();
    return r;
}

template <typename K, typename P>
static int launch_tile_kernel(K kernel, const P& prm, int64_t tiles, int threads, hipStream_t st,
                              size_t lds = lds_tile_bytes<true>() /* 132 KiB: tile + 1 pad slot per 32 */,
                              unsigned grid_y = 1) {
    // (the attribute belongs to the (kernel, device) pair: remembered per device)
    struct Conf { const void* fn; int dev; };
    static thread_local Conf configured[256];
    static thread_local int nconf = 0;
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    bool seen = false;
    for (int i = 0; i < nconf; ++i) seen |= (configured[i].fn == (const void*)kernel && configured[i].dev == dev);
    if (!seen) {
        HIPCHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (nconf < 256) configured[nconf++] = Conf{(const void*)kernel, dev};
    }
    if (tiles <= 0 || tiles > 0x7fffffffLL) return fail(PBH_ERR_INVALID, "tile count out of range");
    hipLaunchKernelGGL(kernel, dim3((unsigned)tiles, grid_y), dim3(threads), lds, st, prm);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

#ifdef PBH_F64
#define FOR_ALL_M(X) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192)
#define FOR_ROW_M(X) X(1024) X(2048) X(4096) X(8192)
#define FOR_ROW_SHORT(X) X(32) X(64) X(128) X(256) X(512)
#else
#define FOR_ALL_M(X) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192) X(16384)
#define FOR_ROW_M(X) X(1024) X(2048) X(4096) X(8192) X(16384)
#define FOR_ROW_SHORT(X) X(32) X(64) X(128) X(256) X(512)
#endif

static int row_grid() {
    static int g = [] {
        const char* e = getenv("PBH_ROW_GRID");
        int v = e ? atoi(e) : 0;
        if (v > 0) return v;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            cus = 256;   // MI355X
        }
        return cus;  // one persistent workgroup per CU (LDS admits one)
    }();
    return g;
}

template <int OP>
static int launch_col(int M, const ColParams& prm0, hipStream_t st) {
    const int F = kTilePoints / M;
    ColParams prm = prm0;
    const int64_t ntile = (prm.ncols + F - 1) / F;
    if (ntile > 0x7fffffffLL) return fail(PBH_ERR_INVALID, "too many column tiles");
    prm.ntile = (int)ntile;
    // (a split re/im exchange variant, k_col<..., true>, gives two workgroups per CU; it measured
    //  slower -- 1.14 vs 0.98 ms -- because the 128-VGPR cap spills ~30 registers; 64 points per thread x
    //  256 threads spills too: not instantiated)
    // (cache-policy experiments on the partial-line variants: non-temporal loads/stores 2-3x slower --
    //  they defeat the L2 merging --, sc1 loads -5 % at best)
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_col<m, OP, PBH_R, false>, prm, ntile, kTilePoints / PBH_R, st);
        FOR_ALL_M(X)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "column pass length " + std::to_string(M));
}

// persistent planar in-place column pass (planar5).  PBH_COLP: 0 = one-tile workgroups (k_col),
// 1 = k_colq with a static tile stride, 2 (default) = k_colq with tiles handed out by an atomic counter
static int colp_mode() {
    static int m = [] { const char* e = diag_env("PBH_COLP"); return e ? atoi(e) : 2; }();
    return m;
}
#ifndef PBH_F64
// the inverse column pass that leaves the four Stokes sums of every 8 columns (tiles of 8 columns x both pols; N1 = 1024)
static int launch_colq_det4(int M, ColpParams prm, hipStream_t st) {
    if (M != 1024 || prm.S % 2) return fail(PBH_ERR_UNSUPPORTED, "pol-pair detecting column pass: 1024 rows, polarisation pairs");
    prm.order = 0;
    int64_t tiles = (int64_t)(prm.S / 2) * (prm.N2 / 8);
    if (tiles > row_grid()) tiles = row_grid();
    if (colp_mode() < 2) prm.counter = nullptr;
    return launch_tile_kernel(k_colq<1024, OP_TW_INV, PBH_R, 2>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16);
}
// the inverse column pass that detects instead of storing (N1 = 64 ... 1024: tiles of 256 ... 16 columns)
static int launch_colq_det(int M, ColpParams prm, hipStream_t st) {
    const int F = kTilePoints / M;
    prm.order = 0;
    int64_t tiles = (int64_t)prm.S * (prm.N2 / F);
    if (tiles > row_grid()) tiles = row_grid();
    if (colp_mode() < 2) prm.counter = nullptr;
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_colq<m, OP_TW_INV, PBH_R, 1>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16);
        X(64) X(128) X(256) X(512) X(1024)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "detecting column pass: 64 ... 1024 rows");
}
#endif
template <int OP>
static int launch_colq(int M, ColpParams prm, hipStream_t st) {
    const int F = kTilePoints / M;
    if (prm.N2 % F != 0) return fail(PBH_ERR_STATE, "k_colq needs whole column groups");
    prm.order = 0;
    int64_t tiles = (int64_t)prm.S * prm.P * (prm.N2 / F);
    if (tiles > row_grid()) tiles = row_grid();
    if (colp_mode() < 2) prm.counter = nullptr;
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_colq<m, OP, PBH_R>, prm, tiles, kTilePoints / PBH_R, st, ((kTilePoints / m) < 16 ? lds_tile_bytes<true>() : lds_tile_bytes<false>()) + 16);
        FOR_ALL_M(X)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "column pass length " + std::to_string(M));
}

template <int DIR>
static int launch_radix(int P, const cf* src, int64_t src_plane, cf* dst, int64_t dst_plane, int S, int64_t N, int N2,
                        int N1, int64_t keep0, int64_t keep1, hipStream_t st) {
    const int64_t chunk = N / P;
    int64_t blocks = ((int64_t)S * chunk + 255) / 256;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    switch (P) {
#define X(pp) case pp: hipLaunchKernelGGL((k_radix_p<pp, DIR>), dim3((unsigned)blocks), dim3(256), 0, st, src, src_plane, dst, dst_plane, S, chunk, N2, N1, keep0, keep1); break;
        X(2) X(3) X(4) X(5) X(7) X(8) X(16)
#undef X
        default: return fail(PBH_ERR_UNSUPPORTED, "radix-P stage: P must be 2, 3, 4, 5, 7, 8 or 16");
    }
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

static int launch_row(int M, const RowParams& prm, hipStream_t st) {
    const int FR = kTilePoints / M;
    int64_t tiles = (prm.nrows + FR - 1) / FR;
    if (FR > 1 && prm.N1 > 1) tiles = (prm.nrows / prm.N1) * ((prm.N1 + FR - 1) / FR);   // tiles do not straddle series (k_row)
    if (tiles > row_grid()) tiles = row_grid();
#if defined(PBH_DIAGNOSTIC) && !defined(PBH_F64)  // ablation builds: -DPBH_DIAGNOSTIC, then PBH_ROW_ABL=1|2|3
    static int abl = [] { const char* e = diag_env("PBH_ROW_ABL"); return e ? atoi(e) : 0; }();
    if (M == 16384 && abl == 1) return launch_tile_kernel(k_row<16384, 32, true, 1>, prm, tiles, 512, st);
    if (M == 16384 && abl == 2) return launch_tile_kernel(k_row<16384, 32, true, 2>, prm, tiles, 512, st);
    if (M == 16384 && abl == 3) return launch_tile_kernel(k_row<16384, 32, true, 3>, prm, tiles, 512, st);
    if (M == 16384 && (abl == 4 || abl == 5)) {  // phase timeline: every launch synchronises and rewrites $PBH_ROW_DBG
        static unsigned long long* dbg = nullptr;
        const int64_t dbg_n = (int64_t)row_grid() * 64 * 8;
        if (!dbg) HIPCHECK(hipMalloc(&dbg, dbg_n * sizeof(unsigned long long)));
        HIPCHECK(hipMemsetAsync(dbg, 0, dbg_n * sizeof(unsigned long long), st));
        RowParams q = prm;
        q.dbg = dbg;
        const int rc = abl == 4 ? launch_tile_kernel(k_row<16384, 32, true, 4>, q, tiles, 512, st)
                                : launch_tile_kernel(k_row<16384, 32, true, 5>, q, tiles, 512, st);
        if (rc != PBH_OK) return rc;
        if (const char* path = diag_env("PBH_ROW_DBG")) {
            HIPCHECK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h((size_t)dbg_n);
            HIPCHECK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            if (FILE* fh = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), fh); fclose(fh); }
        }
        return PBH_OK;
    }
#endif
#ifndef PBH_F64
    if (prm.perm_w == 8) return launch_tile_kernel(k_row2<true>, prm, tiles, 512, st);
#endif
    // Product kernel: memory instructions pinned to the butterflies' ticks, tiles handed out by an atomic
    // counter.  PBH_ROW_SPREAD=2 keeps the ticks but walks tiles with a static stride; the older forms
    // (0 = three bursts, 1 = per stage) exist in -DPBH_DIAGNOSTIC builds only, for the A/B numbers of
    // DESIGN.md 6 (config 2: 1.16 / 1.14 / 1.11 / 1.085 ms).
    static const int spread = [] { const char* e = diag_env("PBH_ROW_SPREAD"); return e ? atoi(e) : 3; }();
#ifdef PBH_DIAGNOSTIC
    if (spread == 1) {
        switch (M) {
#define X(m) case m: return launch_tile_kernel(k_row<m, PBH_R, true, 0, 1>, prm, tiles, kTilePoints / PBH_R, st);
            FOR_ROW_M(X)
#undef X
        }
    }
    if (spread == 0) {
        switch (M) {
#define X(m) case m: return launch_tile_kernel(k_row<m, PBH_R, true>, prm, tiles, kTilePoints / PBH_R, st);
            FOR_ROW_M(X)
#undef X
        }
    }
#endif
    RowParams q = prm;
    if (spread == 2) q.counter = nullptr;
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_row<m, PBH_R, true, 0, 2>, q, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        FOR_ROW_M(X)
        FOR_ROW_SHORT(X)   // rows of the 7-smooth plans whose power-of-two part is small (N = 2^7 5^7: 128-point rows)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "row pass length " + std::to_string(M));
}

static int depth_mode() {
    static const int m = [] { const char* e = diag_env("PBH_DEPTH"); return e ? atoi(e) : 0; }();
    return m;
}
#ifndef PBH_F64
// PBH_ROW_PHASE=0 keeps the complex64-chirp row kernel for generated chirps too (A/B runs)
static bool row_phase_enabled() {
    static const bool on = [] { const char* e = diag_env("PBH_ROW_PHASE"); return e ? atoi(e) != 0 : true; }();
    return on;
}
// rows of M = N2 points, 2^tile / M of them (consecutive k1 of one series) per tile
static bool rowp_ok(int N1, int N2) {
    const int FR = kTilePoints / N2;
    return N2 >= 32 && N2 <= kTilePoints && FR >= 1 && N1 >= 1;   // (a series whose N1 is not a multiple of FR ends in a short tile;
                                                                  //  rows shorter than 1024 points only occur in 7-smooth plans)
}
// 2^14-point rows go through k_rowp16, which reads its phase rows in its own order; PBH_ROW16=0 keeps the 8-byte-per-lane
// kernel (A/B runs).  Decided when the chirp is written (the plan remembers: pbh_plan::phase16).
static bool rowp16_on(int N2) {
    static const bool row16 = [] { const char* e = diag_env("PBH_ROW16"); return e ? atoi(e) != 0 : true; }();
    return row16 && N2 == kTilePoints;
}
static int launch_rowp(int M, RowpParams prm, hipStream_t st) {
    const int FR = kTilePoints / M;
    int64_t tiles = (int64_t)prm.nchan * ((prm.N1 + FR - 1) / FR);
    if (tiles > row_grid()) tiles = row_grid();
    static const bool nofft = [] { const char* e = diag_env("PBH_ROW16_NOFFT"); return e ? atoi(e) != 0 : false; }();
    static const bool defer = [] { const char* e = diag_env("PBH_ROW16_DEFER"); return e ? atoi(e) != 0 : false; }();   // deferred stores (A/B)
    if (prm.phase16 && M == kTilePoints) {
        if (nofft) return launch_tile_kernel(k_rowp16<PBH_R, 1>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        if (prm.chan_freq && prm.cP == 1)
            return launch_tile_kernel(k_rowp16<PBH_R, 0, false, true>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        return defer ? launch_tile_kernel(k_rowp16<PBH_R, 0, true>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16)
                     : launch_tile_kernel(k_rowp16<PBH_R, 0, false>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
    }
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_rowp<m, PBH_R>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        FOR_ROW_M(X)
        FOR_ROW_SHORT(X)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "phase row pass length " + std::to_string(M));
}
#endif

#ifndef PBH_F64
// ---- four-pass schedule (fd4_kernels.hpp) -------------------------------------------------------------------------------
// PBH_FD4: 0 = never, 1 = where the geometry allows it (default), see fd4_ok
static int fd4_mode() {   // (read at every call: the tests switch it inside one process)
    const char* e = getenv("PBH_FD4");
    return e ? atoi(e) : 1;
}
// PBH_FD4_SP=<c><r>: pacing variant of the column (tens) and row (units) kernels (A/B runs)
static int fd4_sp() {
    const char* e = diag_env("PBH_FD4_SP");
    return e ? atoi(e) : 0;
}
// grid of a gang-scheduled kernel: a multiple of 8 * members (blocks b, b + 8, ... share an XCD), one workgroup per CU
static int gang_grid(int members, int64_t units) {
    int g = row_grid();
    if (g >= 8 * members) g -= g % (8 * members);
    else g -= g % members;
    if (g < members) g = members;
    if ((int64_t)(g / members) > units) g = (int)units * members;
    return g;
}
static int launch_colfd(int M, const ColfdParams& prm, hipStream_t st) {
    const int NQ = prm.S / 4, C = (kTilePoints / M) / 4;
    const int grid = gang_grid(NQ, prm.N2 / C);
    switch (M) {
#ifdef PBH_DIAGNOSTIC
#define X(m) case m: { const int sp = fd4_sp() / 10;   /* A/B: 0 = default, 1 = spread deferred pairs, 2 = deferred stores, 3 = twiddle tables read for every tile, 4 = ordinary (not nt) stores */ \
            return sp == 4 ? launch_tile_kernel(k_colfd<m, PBH_R, 2, 1, 0>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16) \
                 : sp == 3 ? launch_tile_kernel(k_colfd<m, PBH_R, 2, 0>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16) \
                 : sp == 2 ? launch_tile_kernel(k_colfd<m, PBH_R, 0>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16) \
                 : sp == 1 ? launch_tile_kernel(k_colfd<m, PBH_R, 1>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16) \
                           : launch_tile_kernel(k_colfd<m, PBH_R, 2>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16); }
#else
#define X(m) case m: return launch_tile_kernel(k_colfd<m, PBH_R, 2>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<false>() + 16);
#endif
        X(64) X(128) X(256) X(512) X(1024)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "direct forward column pass: 64 ... 1024 rows");
}
static int launch_rowq(const RowqParams& prm, hipStream_t st) {
    const int grid = gang_grid(4, (int64_t)(prm.S / 4) * prm.N1);
#ifdef PBH_DIAGNOSTIC
    switch (fd4_sp() % 10) {   // A/B: loads of the next tile that ride in the forward transform (default 10 of 16)
        case 1: return launch_tile_kernel(k_rowq16<PBH_R, 0>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        case 2: return launch_tile_kernel(k_rowq16<PBH_R, 4>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
        case 3: return launch_tile_kernel(k_rowq16<PBH_R, 8>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
    }
#endif
    return launch_tile_kernel(k_rowq16<PBH_R, 10>, prm, grid, kTilePoints / PBH_R, st, lds_tile_bytes<true>() + 16);
}
// geometry the four-pass schedule covers: quads of series (up to 16 of them), one row tile per row, column tiles of up to
// 1024 rows whose span of the caller's block fits a buffer descriptor's 32-bit offsets
static bool fd4_ok(const pbh_plan* p) {
    if (fd4_mode() == 0 || !p->has_phase || !row_phase_enabled() || !p->phase16) return false;
    if (p->P != 1 || p->N2 != kTilePoints || p->N1 < 64 || p->N1 > 1024 || !is_pow2(p->N1) || p->S % 4 != 0 || p->S < 4 || p->S > 64) return false;
    // a tile's rows reach over the whole block: its span and the last row's offset have to fit the 32 unsigned bits of a buffer
    // descriptor's record count and of voffset + soffset (2^24 samples: up to 32 series; 2^22: up to 64)
    const int64_t span = (((int64_t)(p->N1 - 1) * p->N2 + (kTilePoints / p->N1) / 4 - 1) * p->S + 4) * (int64_t)sizeof(cf);
    return span < (1LL << 32) && ((int64_t)(p->N1 - 1) * p->N2 + kTilePoints / p->N1) * p->S * (int64_t)sizeof(cf) < (1LL << 32);
}
#endif

static int launch_small(int M, const SmallParams& prm, hipStream_t st, int64_t nseg = 1) {
    const int F = kTilePoints / M;
    if (nseg < 1 || nseg > 65535) return fail(PBH_ERR_UNSUPPORTED, "segment count out of range for one launch");
    if ((int64_t)M * prm.S * (int64_t)sizeof(cf) > 0x7fffffffLL)
        return fail(PBH_ERR_UNSUPPORTED, "single-tile block larger than 2 GiB (too many series)");
    const int64_t tiles = prm.segs > 1 ? 1 : ((int64_t)prm.S + F - 1) / F;
    switch (M) {
#define X(m) case m: return launch_tile_kernel(k_small<m, PBH_R>, prm, tiles, kTilePoints / PBH_R, st, lds_tile_bytes<true>(), (unsigned)nseg);
        FOR_ALL_M(X)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "single-tile length " + std::to_string(M));
}

static int tr_rows(int S) {
    int tn = kTrElems / S;
    return tn < 1 ? 1 : tn;
}

// Two-axis tiles (k_deint_blk / k_reint_blk): SB series x TB samples = 32 KiB per workgroup.  Used for many
// series (S > 128: SB = 64, several series blocks) and for series counts the power-of-two row transposes do
// not cover (SB = next power of two >= S, lanes beyond S idle).  complex64 needs 16-byte aligned (n*S + s) pairs.
constexpr int kBlkElems = 32 * 1024 / (int)sizeof(cf);
static int blk_series(int S, int64_t N) {
    constexpr int VE = 16 / (int)sizeof(cf);
    if (S % VE != 0 || N % VE != 0 || S < 2) return 0;
#ifndef PBH_F64
    if ((S & (S - 1)) == 0 && S <= 128) return 0;   // k_*_p2 row transposes
#endif
    static const bool on = [] { const char* e = diag_env("PBH_BLK_LAYOUT"); return e ? atoi(e) != 0 : true; }();
    if (!on && S <= 128) return 0;
    int sb = 4;
    while (sb < S && sb < 64) sb <<= 1;
    return sb;
}
template <int SB>
static int launch_deint_blk(const cf* in, cf* work, int64_t N, int S, int64_t plane, int64_t nvalid, hipStream_t st) {
    constexpr int TB = kBlkElems / SB;
    hipLaunchKernelGGL((k_deint_blk<SB, TB>), dim3((unsigned)((N + TB - 1) / TB), (unsigned)((S + SB - 1) / SB)), dim3(256), 0, st,
                       in, work, N, S, plane, nvalid);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}
template <int SB>
static int launch_reint_blk(const cf* work, cf* out, int64_t start, int64_t stop, int S, int64_t plane, hipStream_t st) {
    constexpr int TB = kBlkElems / SB;
    hipLaunchKernelGGL((k_reint_blk<SB, TB>), dim3((unsigned)((stop - start + TB - 1) / TB), (unsigned)((S + SB - 1) / SB)), dim3(256), 0,
                       st, work, out, start, stop, S, plane);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

// nvalid: input time samples that exist (the rest of the N is zero padding; N for an ordinary call)
// the de-interleave pass can carry freq_shift's mixer (mix_ft: device array of S per-series shifts in cycles per sample)
static bool deint_can_mix(int S, int64_t N) {
#ifdef PBH_F64
    (void)S; (void)N;
    return false;
#else
    return blk_series(S, N) == 0 && (S & (S - 1)) == 0 && S >= 2 && S <= 128 && N % tr_rows(S) == 0;
#endif
}
static int launch_deinterleave(const cf* in, cf* work, int64_t N, int S, int64_t nvalid, hipStream_t st, int64_t plane = 0,
                               const double* mix_ft = nullptr) {
    if (plane <= 0) plane = N;   // elements between consecutive series of the planar side
    if (mix_ft && !deint_can_mix(S, N)) return fail(PBH_ERR_STATE, "this de-interleave pass cannot carry the mixer");
    if (!mix_ft)
    switch (blk_series(S, N)) {
        case 4: return launch_deint_blk<4>(in, work, N, S, plane, nvalid, st);
        case 8: return launch_deint_blk<8>(in, work, N, S, plane, nvalid, st);
        case 16: return launch_deint_blk<16>(in, work, N, S, plane, nvalid, st);
        case 32: return launch_deint_blk<32>(in, work, N, S, plane, nvalid, st);
        case 64: return launch_deint_blk<64>(in, work, N, S, plane, nvalid, st);
    }
    const int TN = tr_rows(S);
    int64_t done = 0;
#ifndef PBH_F64
    // the power-of-two row transposes take whole tiles; a length that is not a multiple of the tile (7-smooth lengths with
    // few factors of two) leaves its last samples to the generic kernel (the mixer rides in whole-tile launches only)
    if ((S & (S - 1)) == 0 && S <= 128 && (N % TN == 0 || (!mix_ft && N >= TN))) {
        const unsigned grid = (unsigned)(N / TN);
        switch (S) {
#define X(s) case s: if (mix_ft) hipLaunchKernelGGL((k_deinterleave_p2<s, true>), dim3(grid), dim3(256), 0, st, in, work, N, plane, nvalid, mix_ft); \
                     else hipLaunchKernelGGL((k_deinterleave_p2<s, false>), dim3(grid), dim3(256), 0, st, in, work, N, plane, nvalid, mix_ft); break;
            X(1) X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
        }
        HIPCHECK(hipGetLastError());
        done = (int64_t)grid * TN;
        if (done == N) return PBH_OK;
    }
#endif
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((N - done + TN - 1) / TN)), dim3(256),
                       (size_t)TN * (S + 1) * sizeof(cf), st, in + done * S, work + done, N - done, S, TN, plane, nvalid - done);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

// opitch: elements between output rows (0 = compact, S); pitched rows (a channel slice of a wider array) are
// written by the power-of-two row transposes and the generic kernel only -- slice_fast_ok()
static int launch_reinterleave(const cf* work, cf* out, int64_t start, int64_t stop, int S, int64_t plane,
                               hipStream_t st, int64_t opitch = 0, const int64_t* dly = nullptr) {
    if (stop <= start) return PBH_OK;
    const bool pitched = opitch > 0 && opitch != S;
    if (!pitched) opitch = S;
    if (!pitched && !dly)
    switch (blk_series(S, plane)) {
        case 4: return launch_reint_blk<4>(work, out, start, stop, S, plane, st);
        case 8: return launch_reint_blk<8>(work, out, start, stop, S, plane, st);
        case 16: return launch_reint_blk<16>(work, out, start, stop, S, plane, st);
        case 32: return launch_reint_blk<32>(work, out, start, stop, S, plane, st);
        case 64: return launch_reint_blk<64>(work, out, start, stop, S, plane, st);
    }
    const int TN = tr_rows(S);
    int64_t done = 0;
#ifndef PBH_F64
    if ((S & (S - 1)) == 0 && S <= 128 && !((pitched || dly) && (S < 2 || (opitch & 1) || (reinterpret_cast<uintptr_t>(out) & 15)))) {
        const int64_t full = (stop - start) / TN;
        if (full > 0) {
            switch (S) {
#define X(s) case s: if (dly) hipLaunchKernelGGL((k_reinterleave_p2<s, true, true>), dim3((unsigned)full), dim3(256), 0, st, work, out, start, plane, opitch, dly); \
                     else if (pitched) hipLaunchKernelGGL((k_reinterleave_p2<s, true>), dim3((unsigned)full), dim3(256), 0, st, work, out, start, plane, opitch, dly); \
                     else hipLaunchKernelGGL((k_reinterleave_p2<s, false>), dim3((unsigned)full), dim3(256), 0, st, work, out, start, plane, opitch, dly); break;
                X(1) X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
            }
            HIPCHECK(hipGetLastError());
        }
        done = full * TN;
    }
#endif
    if (start + done < stop) {  // tail (or everything, for other S) through the generic kernel
        const int64_t s2 = start + done;
        hipLaunchKernelGGL(k_reinterleave, dim3((unsigned)((stop - s2 + TN - 1) / TN)), dim3(256),
                           (size_t)TN * (S + 1) * sizeof(cf), st, work, out + done * opitch, s2, stop, S, TN, plane, opitch, dly);
        HIPCHECK(hipGetLastError());
    }
    return PBH_OK;
}

// The final layout pass as a DETECTING pass at full time resolution (k_reinterleave_p2<.., DET>): float32 to_intensity /
// to_stokes of the cropped samples straight from the planar workspace.  Whole tiles through the tile kernel, the last rows
// through k_detect_planar (one output per wavefront: fine for fewer rows than a tile).
static bool reint_detect_ok(int S, int npol, int mode, int64_t N) {
#ifdef PBH_F64
    (void)S; (void)npol; (void)mode; (void)N;
    return false;
#else
    static const bool on = [] { const char* e = diag_env("PBH_DETECT_REINT"); return e ? atoi(e) != 0 : true; }();
    // power-of-two counts up to 128: k_reinterleave_p2; many series and other even counts: k_reint_blk
    if (!on || !(((S & (S - 1)) == 0 && S >= 2 && S <= 128) || blk_series(S, N) != 0)) return false;
    return mode == PBH_DETECT_INTENSITY || npol == 2;
#endif
}
#ifndef PBH_F64
template <int SB>
static int launch_reint_blk_detect(const cf* work, real* out, int64_t start, int64_t stop, int S, int mode, int64_t plane, hipStream_t st) {
    constexpr int TB = kBlkElems / SB;
    const dim3 grid((unsigned)((stop - start + TB - 1) / TB), (unsigned)((S + SB - 1) / SB));
    cf* o = reinterpret_cast<cf*>(out);
    if (mode == 0) hipLaunchKernelGGL((k_reint_blk<SB, TB, 0>), grid, dim3(256), 0, st, work, o, start, stop, S, plane);
    else if (mode == 1) hipLaunchKernelGGL((k_reint_blk<SB, TB, 1>), grid, dim3(256), 0, st, work, o, start, stop, S, plane);
    else if (mode == 2) hipLaunchKernelGGL((k_reint_blk<SB, TB, 2>), grid, dim3(256), 0, st, work, o, start, stop, S, plane);
    else hipLaunchKernelGGL((k_reint_blk<SB, TB, 3>), grid, dim3(256), 0, st, work, o, start, stop, S, plane);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}
#endif
static int launch_reint_detect(const cf* work, real* out, int64_t start, int64_t stop, int S, int nchan, int npol, int mode, int64_t plane,
                               hipStream_t st) {
    if (stop <= start) return PBH_OK;
#ifndef PBH_F64
    switch (blk_series(S, plane)) {   // many series, or an even count that is not a power of two: two-axis tiles, edges included
        case 4: return launch_reint_blk_detect<4>(work, out, start, stop, S, mode, plane, st);
        case 8: return launch_reint_blk_detect<8>(work, out, start, stop, S, mode, plane, st);
        case 16: return launch_reint_blk_detect<16>(work, out, start, stop, S, mode, plane, st);
        case 32: return launch_reint_blk_detect<32>(work, out, start, stop, S, mode, plane, st);
        case 64: return launch_reint_blk_detect<64>(work, out, start, stop, S, mode, plane, st);
    }
    if ((S & (S - 1)) != 0 || S > 128) return fail(PBH_ERR_STATE, "detecting layout pass: no tile kernel for this series count");
    const int TN = tr_rows(S);
    const int64_t full = (stop - start) / TN;
    if (full > 0) {
        cf* o = reinterpret_cast<cf*>(out);
        switch (S) {
#define Y(s, m) hipLaunchKernelGGL((k_reinterleave_p2<s, false, false, m>), dim3((unsigned)full), dim3(256), 0, st, work, o, start, plane, (int64_t)s, (const int64_t*)nullptr)
#define X(s) case s: if (mode == 0) Y(s, 0); else if (mode == 1) Y(s, 1); else if (mode == 2) Y(s, 2); else Y(s, 3); break;
            X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
#undef Y
        }
        HIPCHECK(hipGetLastError());
    }
    const int64_t done = full * TN, rest = stop - start - done;
    if (rest > 0) {
        const int oe = mode == 0 ? npol : (mode == 1 ? 1 : 4);
        hipLaunchKernelGGL(k_detect_planar, dim3((unsigned)((rest + 3) / 4), (unsigned)nchan), dim3(256), 0, st, work,
                           out + done * nchan * oe, plane, start + done, rest, nchan, npol, mode, 1);
        HIPCHECK(hipGetLastError());
    }
    return PBH_OK;
#else
    (void)work; (void)out; (void)S; (void)nchan; (void)npol; (void)mode; (void)plane; (void)st;
    return fail(PBH_ERR_UNSUPPORTED, "detecting layout pass: float32 build only");
#endif
}

// Layout passes with the radix-P stage folded in (float32, 2 <= S <= 128 a power of two, tiles of >= 16 samples).
static bool radix_layout_ok(int S, int P, int64_t N, int N2) {
    static const bool on = [] { const char* e = diag_env("PBH_RADIX_FUSE"); return e ? atoi(e) != 0 : true; }();
    if (!on || S < 2 || S > 128 || (S & (S - 1)) != 0) return false;
    if (P != 2 && P != 3 && P != 4 && P != 5 && P != 7 && P != 8 && P != 16) return false;
    const int E = radix_tile_e(P) * 8 / (int)sizeof(cf), TN = E / S;   // RadixTile<P>
    return TN >= 128 / (int)sizeof(cf) && (N / P) % TN == 0 && N2 % TN == 0;   // planar rows of whole 128-byte lines
}
template <int S>
static int launch_deint_radix_s(int P, const cf* in, cf* work, int64_t N, int N2, int N1, int64_t nvalid, hipStream_t st) {
    const int64_t chunk = N / P;
    switch (P) {
#define X(pp) case pp: hipLaunchKernelGGL((k_deint_radix<S, pp>), dim3((unsigned)(chunk / (RadixTile<pp>::E * 8 / (int)sizeof(cf) / S))), dim3(256), 0, st, in, work, chunk, N, N2, N1, nvalid); break;
        X(2) X(3) X(4) X(5) X(7) X(8) X(16)
#undef X
    }
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}
template <int S>
static int launch_reint_radix_s(int P, const cf* work, cf* out, int64_t N, int N2, int N1, int64_t start, int64_t stop,
                                hipStream_t st) {
    const int64_t chunk = N / P;
    switch (P) {
#define X(pp) case pp: hipLaunchKernelGGL((k_reint_radix<S, pp>), dim3((unsigned)(chunk / (RadixTile<pp>::E * 8 / (int)sizeof(cf) / S))), dim3(256), 0, st, work, out, chunk, N, N2, N1, start, stop); break;
        X(2) X(3) X(4) X(5) X(7) X(8) X(16)
#undef X
    }
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}
static int launch_deint_radix(int S, int P, const cf* in, cf* work, int64_t N, int N2, int N1, int64_t nvalid,
                              hipStream_t st) {
    switch (S) {
#define X(s) case s: return launch_deint_radix_s<s>(P, in, work, N, N2, N1, nvalid, st);
        X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "fused radix layout pass: unsupported series count");
}
static int launch_reint_radix(int S, int P, const cf* work, cf* out, int64_t N, int N2, int N1, int64_t start,
                              int64_t stop, hipStream_t st) {
    if (stop <= start) return PBH_OK;
    switch (S) {
#define X(s) case s: return launch_reint_radix_s<s>(P, work, out, N, N2, N1, start, stop, st);
        X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
    }
    return fail(PBH_ERR_UNSUPPORTED, "fused radix layout pass: unsupported series count");
}

extern "C" int pbh_relayout(int device, void* hip_stream, int, const void* in_dev, int in_layout, int64_t in_pitch, void* out_dev,
                 int out_layout, int64_t out_pitch, int64_t nsample, int nseries) {
    if (!in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || nseries <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    auto bad = [](int l) { return l != PBH_LAYOUT_SAMPLE_MAJOR && l != PBH_LAYOUT_SERIES_MAJOR; };
    if (bad(in_layout) || bad(out_layout)) return fail(PBH_ERR_INVALID, "bad layout");
    if ((in_layout == PBH_LAYOUT_SERIES_MAJOR && in_pitch < nsample) || (out_layout == PBH_LAYOUT_SERIES_MAJOR && out_pitch < nsample))
        return fail(PBH_ERR_INVALID, "pitch < nsample");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    const cf* in = (const cf*)in_dev;
    cf* out = (cf*)out_dev;
    if (in_layout == out_layout) {
        if (in_layout == PBH_LAYOUT_SAMPLE_MAJOR) {
            HIPCHECK(hipMemcpyAsync(out, in, sizeof(cf) * (size_t)nsample * nseries, hipMemcpyDeviceToDevice, st));
        } else {
            HIPCHECK(hipMemcpy2DAsync(out, sizeof(cf) * (size_t)out_pitch, in, sizeof(cf) * (size_t)in_pitch,
                                      sizeof(cf) * (size_t)nsample, (size_t)nseries, hipMemcpyDeviceToDevice, st));
        }
        return PBH_OK;
    }
    if (in_layout == PBH_LAYOUT_SAMPLE_MAJOR) return launch_deinterleave(in, out, nsample, nseries, nsample, st, out_pitch);
    return launch_reinterleave(in, out, 0, nsample, nseries, in_pitch, st);
}

// Kernel sequence of one dedispersion: in (N,S) interleaved -> out (stop-start, S) interleaved.

// Ping-pong schedule of the planar pipeline's middle passes (column, row, column): each pass reads one planar buffer and
// writes the other instead of updating `work` in place.  OPT-IN (PBH_OOP=1), because it does not pay: a plain in-place
// streaming update of the column pattern tops out at 5.26 TB/s and the same traffic out of place with a padded row pitch at
// 5.58 (tools/micro/colcopy.hip, profiles/r03_colcopy.txt), but in the real passes the schedule changes nothing at the
// product's pitch (column passes 0.846 -> 0.856 / 0.754 -> 0.759 ms, row pass 1.00 -> 0.99) and 1.3 % of the step with a
// padded pitch on top (profiles/r03_colq_pitch_oop.txt) -- for a second workspace of S * N elements.
static int oop_mode() {
    static const int m = [] { const char* e = diag_env("PBH_OOP"); return e ? atoi(e) : 0; }();
    return m;
}
#ifndef PBH_F64
// ---- allocation classes ---------------------------------------------------------------------------------------------------
// Measured on MI355X (tools/micro/bufprobe.hip, classprobe2.hip; profiles/r04_bufprobe.txt, r04_classprobe2_blocks.txt): every
// large hipMalloc allocation belongs to a class (at least three exist; runs of 8 / 32 GiB of the allocator's heap share one); a
// copy between allocations of the SAME class (or inside one allocation) takes 0.755 ms per 2 GiB, between allocations of
// DIFFERENT class 0.72 ms, whatever the offsets inside them -- and the passes of this library follow: the de-interleave pass
// 0.80 / 0.75 ms, the column passes out of place 0.86 / 0.82 and 0.756 / 0.72, the direct forward pass 1.25 / 1.18.  That is
// the 2.5 % per-process spread of round 3: which classes the process's buffers happened to get.  The class of an allocation
// cannot be asked for, but a pair can be probed: one timed copy between them against a copy inside one allocation (same class
// by definition).
// Copies of `bytes` from a to b1 and from a to b2, alternating, REPS times each: the fastest of each (the first pair warms the
// TLBs and does not count).  The two times come from the same moments of the same chip, which is what makes a 4-5 %
// difference readable (single timings of a 1-GiB copy scatter by 2 %).
static bool probe_copy_pair(const void* a, void* b1, void* b2, size_t bytes, hipStream_t st, float* t1, float* t2) {
    const unsigned grid = (unsigned)(bytes / 16 / 1024);
    *t1 = *t2 = -1.f;
    if (grid == 0) return false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        (void)hipGetLastError();
        return false;
    }
    bool ok = true;
    constexpr int REPS = 6;
    for (int rep = 0; rep <= REPS && ok; ++rep) {
        for (int which = 0; which < 2 && ok; ++which) {
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, st, (const float4*)a, (float4*)(which ? b2 : b1), (int64_t)grid * 1024);
            (void)hipEventRecord(e1, st);
            float ms = 0.f;
            ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            float* t = which ? t2 : t1;
            if (ok && rep > 0 && (*t < 0 || ms < *t)) *t = ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    return ok && *t1 > 0 && *t2 > 0;
}
static bool class_probing() {   // PBH_CLASS=0: take allocations as they come (A/B runs)
    const char* e = getenv("PBH_CLASS");
    return e ? atoi(e) != 0 : true;
}
constexpr float kClassGap = 0.02f;   // relative difference of the two copies below which a probe decides nothing
// Second work buffer, chosen so that no pass of the four-/five-pass schedules streams between two allocations of one class.
// What the classes are (tools/micro/classprobe2.hip, profiles/r04_classprobe2_blocks.txt): consecutive 2-GiB allocations of a
// process keep one class for runs of 4 or 16 blocks (8 / 32 GiB of the physical heap), at least three classes exist, and a copy is
// slow (0.76-0.78 ms per 2 GiB instead of 0.72-0.74) exactly when source and destination are of the SAME class -- as it is, by
// definition, inside one allocation.  So: a 4-GiB reference allocation gives the slow time (first half -> second half), and
// candidates are allocated one after the other (all held, so that the allocator walks on through its heap) until one copies fast
// from `work`, from the caller's input and from the caller's output (read only: nothing of the caller's is written); at most
// 24 of them and never more than a quarter of the free memory (48 GiB held for ~0.2 s in the worst case at 2 GiB per block).  With PBH_CLASS=0, or
// for blocks under 1 GiB (a streaming driver's chunk plans), the first allocation is taken as it comes.
static cf* ensure_work2(pbh_plan* p, const void* in, const void* out, size_t out_bytes) {
    if (p->work2 || !p->work) return p->work2;
    const size_t bytes = sizeof(cf) * (size_t)p->S * (size_t)p->N;
    const size_t len = bytes & ~(size_t)16383;
    constexpr int kCandCap = 24;   // runs of one class are up to 16 blocks of 2 GiB long (r04_classprobe2_blocks.txt)
    void* cand[kCandCap] = {};
    int n = 0, pick = 0, kMaxCand = kCandCap;
    {   // never hold more than a quarter of the free memory
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) kMaxCand = (int)std::max<size_t>(1, std::min<size_t>(kCandCap, mfree / 4 / std::max<size_t>(bytes, 1)));
        (void)hipGetLastError();
    }
    const bool probe = class_probing() && len >= ((size_t)1 << 30) && kMaxCand > 1;
    static const bool trace = getenv("PBH_TRACE_ALLOC") != nullptr;
    if (dev_alloc(p, &cand[0], bytes) != PBH_OK) { (void)hipGetLastError(); return nullptr; }
    n = 1;
    bool opposite = false;
    float t_same = -1.f, t_diff = -1.f;
    void* ref = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (probe && hipMalloc(&ref, 2 * len) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        if (out == in || out_bytes < ((size_t)1 << 30)) out = nullptr;   // detect tails and short crops: too small to matter / to time
        auto local = [&](const void* q) {   // only this device's own memory is worth (and safe) to time: not a peer's mapping, not host memory
            hipPointerAttribute_t at;
            const bool ok = hipPointerGetAttributes(&at, q) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == p->device;
            (void)hipGetLastError();
            return ok;
        };
        if (in && !local(in)) in = nullptr;
        if (out && !local(out)) out = nullptr;
        const size_t len_out = out ? std::min(len, out_bytes & ~(size_t)16383) : 0;
        auto copy_ms = [&](const void* a, void* b, size_t l) -> float {   // fastest of three after a warm-up copy
            const unsigned grid = (unsigned)(l / 16 / 1024);
            float best = -1.f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, p->stream);
                hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, p->stream, (const float4*)a, (float4*)b, (int64_t)grid * 1024);
                (void)hipEventRecord(e1, p->stream);
                float ms = 0.f;
                if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
                if (rep > 0 && (best < 0 || ms < best)) best = ms;
            }
            return best;
        };
        int best_score = -1;
        float best_sum = 0.f;
        // `slow` = the fastest same-class copy seen so far (the reference's, taken again next to every second candidate: the part
        // warms up while this runs).  The two groups lie 4-5 % apart and scatter by 1-2 % each: a pair counts as of two classes
        // when its copy beats 0.984 x slow twice.
        float slow = -1.f;
        auto note_slow = [&](float t) { if (t > 0 && (slow < 0 || t < slow)) slow = t; };
        auto fast = [&](const void* a, void* b, size_t l, float* t) {
            const float scale = (float)((double)l / (double)len);
            *t = copy_ms(a, b, l);
            if (*t <= 0 || *t >= 0.984f * slow * scale) return false;
            const float again = copy_ms(a, b, l);
            if (again > *t) *t = again;
            return again > 0 && again < 0.984f * slow * scale;
        };
        note_slow(copy_ms(ref, (char*)ref + len, len));
        // is `work` itself of another class than the caller's arrays?  Then it can take the Q4 intermediate (input -> work -> work2
        // -> work -> output) and any candidate that differs from work will do; else the candidate has to differ from all three
        float twx = -1.f, twy = -1.f;
        const bool work_free = slow > 0 && (!in || fast(in, p->work, len, &twx)) && (!out || fast(out, p->work, len_out, &twy));
        if (trace) fprintf(stderr, "[pbhip] work %p: same-class copy %.4f ms; from the input %.4f, from the output %.4f%s\n", (void*)p->work, slow, twx, twy,
                           work_free ? " (of another class than both)" : "");
        while (true) {
            void* c = cand[n - 1];
            if ((n - 1) % 2 == 0) note_slow(copy_ms(ref, (char*)ref + len, len));
            if (slow <= 0) break;
            float tw = -1.f, tx = -1.f, ty = -1.f;
            const bool fw = fast(p->work, c, len, &tw);
            if (tw <= 0) break;
            // the caller's arrays only for candidates that differ from work (a run of one class is walked through at ~7 ms per block)
            const bool fx = fw && (work_free || !in || fast(in, c, len, &tx)), fy = fw && (work_free || !out || fast(out, c, len_out, &ty));
            const float ts = slow;
            const int score = 4 * (int)fw + 2 * (int)fx + (int)fy;   // work <-> work2 first: two passes stream between them
            const float sum = tw + (tx > 0 ? tx : 0.f) + (ty > 0 ? ty : 0.f);
            if (trace)
                fprintf(stderr, "[pbhip] work2 candidate %d at %p: same-class copy %.4f ms; from work %.4f%s, from the input %.4f%s, from the output %.4f%s\n",
                        n, c, ts, tw, fw ? " (other class)" : "", tx, in && fx ? " (other class)" : "", ty, out && fy ? " (other class)" : "");
            if (score > best_score || (score == best_score && sum < best_sum)) {
                best_score = score; best_sum = sum; pick = n - 1;
                opposite = fw; t_same = ts; t_diff = tw;
            }
            if ((fw && fx && fy) || n == kMaxCand) break;
            if (dev_alloc(p, &cand[n], bytes) != PBH_OK) { (void)hipGetLastError(); break; }
            ++n;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ref) (void)hipFree(ref);
    (void)hipGetLastError();
    for (int i = 0; i < n; ++i)
        if (i != pick) { (void)hipFree(cand[i]); p->owned_bytes -= (int64_t)bytes; }
    p->work2 = (cf*)cand[pick];
    if (opposite) {
        p->cls_t_same = t_same;
        p->cls_t_diff = t_diff;
        p->cls_len = std::min<size_t>(len, (size_t)1 << 30);
    }
    if (trace) fprintf(stderr, "[pbhip] work2: candidate %d of %d%s\n", pick + 1, n, probe ? (opposite ? "" : " (none of another class than work)") : " (no probing)");
    return p->work2;
}
static cf* ensure_work2(pbh_plan* p) { return ensure_work2(p, nullptr, nullptr, 0); }
// class of a caller's buffer relative to the plan's work buffer: 0 = the same, 1 = the other, -1 = unknown (small buffers,
// plans without an opposite-class pair, undecided probes).  The probe copies from `ptr` into work and into work2 (scratch
// between calls), alternating; whichever is faster is the buffer of the other class.  Cached per (pointer, size).
static int buffer_class(pbh_plan* p, const void* ptr, size_t bytes) {
    if (!class_probing() || !ptr || !p->work || !p->work2 || p->cls_t_diff <= 0 || bytes < p->cls_len) return -1;
    for (auto& e : p->cls_cache)
        if (e.ptr == ptr && e.bytes == bytes) return e.cls;
    constexpr int kClassProbes = 8;
    if (p->cls_probes >= kClassProbes) return -1;
    ++p->cls_probes;
    float t1 = 0.f, t2 = 0.f;
    int cls = -1;
    if (probe_copy_pair(ptr, p->work, p->work2, p->cls_len, p->stream, &t1, &t2)) {
        if (t1 < (1.f - kClassGap) * t2) cls = 1;        // faster into work: of the class work is NOT
        else if (t2 < (1.f - kClassGap) * t1) cls = 0;
    }
    auto& e = p->cls_cache[p->cls_next++ & 7];
    e.ptr = ptr;
    e.bytes = bytes;
    e.cls = cls;
    static const bool trace = getenv("PBH_TRACE_ALLOC") != nullptr;
    if (trace) fprintf(stderr, "[pbhip] class of %p (%zu bytes): copy to work %.4f ms, to work2 %.4f ms -> %d\n", ptr, bytes, t1, t2, cls);
    return cls;
}
#else
static cf* ensure_work2(pbh_plan* p) {
    if (!p->work2 && p->work) {
        void* q = nullptr;
        if (dev_alloc(p, &q, sizeof(cf) * (size_t)p->S * (size_t)p->N) == PBH_OK) p->work2 = (cf*)q;
        else (void)hipGetLastError();
    }
    return p->work2;
}
static cf* ensure_work2(pbh_plan* p, const void*, const void*, size_t) { return ensure_work2(p); }
#endif

// true when the detect tail can be fused (planar work buffer holds the full dedispersed series)

// Detection inside the inverse column pass (k_colq<.., DET> + k_detect_reduce): for |z|^2 and Stokes I, whose sums need one
// series at a time.  PBH_DETECT_COLQ=0 restores the separate read pass over the dedispersed voltages (k_detect_planar).
static bool detect_in_colq() {
    static const bool on = [] { const char* e = diag_env("PBH_DETECT_COLQ"); return e ? atoi(e) != 0 : true; }();
    return on;
}
static real* ensure_det_part(pbh_plan* p, size_t bytes) {
    if (p->det_bytes < bytes) {
        if (p->det_part) (void)hipFree(p->det_part);
        p->det_part = nullptr;
        p->det_bytes = 0;
        void* q = nullptr;
        if (dev_alloc(p, &q, bytes) == PBH_OK) {
            p->det_part = (real*)q;
            p->det_bytes = bytes;
        } else (void)hipGetLastError();
    }
    return p->det_part;
}

// columns per k_colmix tile: a power of two, as many as keep L rows within the 64-KiB tile (short transforms get wide
// tiles: a 25-point stage works on 25 x 256 points per tile); whole 128-byte lines up to L = 512, 64-byte pieces beyond
static int mix_wlog2(int L, int N2) {
    const int budget = kMixTileBytes / (int)sizeof(cf);
    int w = 64 / (int)sizeof(cf);
    while (2 * w * L <= budget && 2 * w <= 512 && 2 * w <= N2) w *= 2;
    return ilog2(w);
}
template <int DIR>
static int launch_colmix(const MixParams& prm, hipStream_t st) {
    static const bool nostage = [] { const char* e = diag_env("PBH_MIX_NOSTAGE"); return e ? atoi(e) != 0 : false; }();   // (timing runs)
    MixParams q = prm;
    if (nostage) q.nstage = 0;
    const size_t lds = ((size_t)prm.L << prm.wlog2) * sizeof(cf) + (size_t)prm.L * sizeof(cf) + (size_t)prm.L * sizeof(unsigned short) + 16 +
                       kMixMaxStages * sizeof(int) + 16;
    int64_t tiles = (int64_t)prm.S * prm.nblock * prm.ncolgrp;
    if (tiles > 512) tiles = 512;   // persistent: two workgroups per CU
    return launch_tile_kernel(k_colmix<DIR>, q, tiles, 512, st, (int)lds);
}

static int launch_rowmix(const pbh_plan* p, cf* work, hipStream_t st, bool fwd_only = false) {
    RowMixParams r{};
    r.data = work;
    r.chirp = p->chirp;
    r.nrows = (int64_t)p->S * p->N1;
    r.N1 = p->N1; r.npol = p->npol; r.N2 = p->N2;
    r.FR = (kMixTileBytes / (int)sizeof(cf)) / p->N2;
    r.nstage = p->mixR.nstage;
    for (int j = 0; j < r.nstage; ++j) r.radix[j] = p->mixR.radix[j];
    r.wl = p->mixR.wl;
    const size_t lds = ((size_t)r.FR * p->N2 + p->N2) * sizeof(cf) + 2 * kMixMaxStages * sizeof(int) + 16;
    int64_t tiles = (int64_t)p->S * ((p->N1 + r.FR - 1) / r.FR);   // tiles do not straddle series
    if (tiles > 2048) tiles = 2048;
    return fwd_only ? launch_tile_kernel(k_rowmix<true>, r, tiles, 512, st, (int)lds)
                    : launch_tile_kernel(k_rowmix<false>, r, tiles, 512, st, (int)lds);
}

// The P-point stage of a two-level 7-smooth plan with a SHORT P runs the elementwise k_radix_p (P samples one chunk apart per
// thread, fully coalesced) instead of k_colmix: a 3-row tile of at most 512 columns leaves the tile kernel 1536 points for 512
// threads (9 953 280 = 2^13 * 3 * 405, 16 series: 1.22 ms per pass as a tile, 0.75 for 5 rows; profiles/r03_mix_split.txt).
// PBH_MIX_RADIXP=0: the tile kernel for every P.
static bool mix_radix_p(int P) {
    static const bool on = [] { const char* e = diag_env("PBH_MIX_RADIXP"); return e ? atoi(e) != 0 : true; }();
    return on && (P == 2 || P == 3 || P == 4 || P == 5 || P == 7 || P == 8);
}

// k_colmix parameters of a mixed plan's two column roles (mixed_kernels.hpp): A = the P-point stage over rows one chunk
// N / P apart, B = the Q-point pass inside each of the P row blocks.  ld / st: planar arrays (series pitch ldp / stpl);
// only element (time) indices in [k0, k1) are stored, at index - shift.
static MixParams mix_role_a(const pbh_plan* p, const cf* ld, int64_t ldp, cf* stp, int64_t stpl, int64_t k0, int64_t k1,
                            int64_t shift) {
    const int64_t N = p->N;
    const int N1 = p->N1, N2 = p->N2, P = p->P;
    MixParams m{};
    m.ld = ld; m.ld_plane = ldp; m.st = stp; m.st_plane = stpl;
    m.wlog2 = mix_wlog2(P, N2);
    m.ncols = N / P;
    m.S = p->S; m.L = P; m.rstride = N / P; m.nblock = 1; m.bstride = 0; m.ncolgrp = (m.ncols + (1 << m.wlog2) - 1) >> m.wlog2;
    m.xdiv = N2; m.ystep = 1; m.nmod = N1; m.mult = N2; m.y0mul = 0;
    m.tw = BigTwiddle{p->tw_hi, p->tw_lo, p->tw_shift, N - 1};
    m.tw.nmod = N;
    m.nstage = p->mixP.nstage;
    for (int j = 0; j < m.nstage; ++j) m.radix[j] = p->mixP.radix[j];
    m.wl = p->mixP.wl; m.perm = p->mixP.perm; m.keep0 = k0; m.keep1 = k1; m.st_shift = shift;
    m.counter = nullptr;
    return m;
}
static MixParams mix_role_b(const pbh_plan* p, const cf* ld, int64_t ldp, cf* stp, int64_t stpl, int64_t k0, int64_t k1,
                            int64_t shift) {
    const int64_t N = p->N;
    const int N1 = p->N1, N2 = p->N2, P = p->P, Q = N1 / P;
    MixParams m{};
    m.ld = ld; m.ld_plane = ldp; m.st = stp; m.st_plane = stpl;
    m.wlog2 = mix_wlog2(Q, N2);
    m.ncols = N2;
    m.S = p->S; m.L = Q; m.rstride = N2; m.nblock = P; m.bstride = (int64_t)Q * N2; m.ncolgrp = (m.ncols + (1 << m.wlog2) - 1) >> m.wlog2;
    m.xdiv = 1; m.ystep = P; m.nmod = N; m.mult = 1; m.y0mul = 1;
    m.tw = BigTwiddle{p->tw_hi, p->tw_lo, p->tw_shift, N - 1};
    m.tw.nmod = N;
    m.nstage = p->mixQ.nstage;
    for (int j = 0; j < m.nstage; ++j) m.radix[j] = p->mixQ.radix[j];
    m.wl = p->mixQ.wl; m.perm = p->mixQ.perm; m.keep0 = k0; m.keep1 = k1; m.st_shift = shift;
    m.counter = nullptr;
    return m;
}

// device layouts of the two ends (pbh_dedisperse_layout); pitches in elements, used when series-major

// the final layout pass, once per output part (IoLayout::part_ptr) or once for the whole output
static int launch_reinterleave_parts(const cf* work, cf* out, int64_t start, int64_t stop, int S, int64_t plane, hipStream_t st,
                                     int64_t opitch, const std::vector<cf*>& part_ptr, const std::vector<int64_t>& part_row) {
    if (part_ptr.empty()) return launch_reinterleave(work, out, start, stop, S, plane, st, opitch);
    for (size_t i = 0; i < part_ptr.size(); ++i) {
        const int64_t r0 = part_row[i], r1 = part_row[i + 1];
        if (r1 > r0) PBHCHECK(launch_reinterleave(work, part_ptr[i], start + r0, start + r1, S, plane, st, opitch));
    }
    return PBH_OK;
}

// One-tile plans (nsample <= 2^tile) whose blocks have many series run as layout pass + planar row pass + layout pass
// instead of the single interleaved kernel (build_steps); needs the planar work buffer (allocated at plan creation).
static bool single_planar_ok(const pbh_plan* p) {
    static const int mode = [] { const char* e = diag_env("PBH_SINGLE_PLANAR"); return e ? atoi(e) : 1; }();
    if (!mode || p->N1 != 1 || p->bsL || p->plain_fft || !is_pow2(p->N)) return false;
    const int64_t M = p->N;
    if (M < 1024 || M > kTilePoints) return false;                  // the planar row kernels' lengths (FOR_ROW_M)
    const int FR = (int)(kTilePoints / M);
    if (FR > 1 && ((p->npol & (p->npol - 1)) != 0 || p->npol > FR)) return false;   // rows of a tile share chirp rows pol by pol
    // worth it from a few MiB on (below that the one-kernel form wins on launch count), and needed beyond the
    // one-kernel form's 2-GiB addressing limit
    return mode == 2 || (int64_t)p->S * M >= (1LL << 21);
}

#ifndef PBH_F64
// Which work buffer takes the Q4 intermediate of the four-pass schedule?  The allocation classes say it for clear cases, but a
// 1-GiB probe copy separates the classes by 3-5 % only and the caller's buffers can be of either; the passes themselves are the
// better instrument: on the first call with an (input, output) pair both assignments run twice, timed with events on the
// plan's stream, and the faster one is kept for that pair (both produce the same output -- the runs are real runs).
static int fd4_roles(pbh_plan* p, const cf* in, cf* out, const DetectTail& tail, const IoLayout& io) {
    if (p->fd4_force >= 0) return p->fd4_force;
    const void* okey = tail.out ? (const void*)tail.out : (const void*)out;
    for (auto& e : p->role_cache)
        if (e.in == in && e.out == okey && e.in) return e.swap;
    // bit 0: the Q4 intermediate lives in work2 (else in work); bit 1: the inverse column pass runs out of place, back into
    // the buffer the Q4 intermediate left (else in place on the planar one)
    // All one-time work belongs to a plan's FIRST call: only the first pair of a plan is timed (8 extra runs), later pairs
    // -- a caller that hands over a fresh output array at every call shows a new one each time -- take that decision as it stands.
    constexpr int kRoleTunes = 1;
    int swap = p->role_last;
    const size_t bytes = sizeof(cf) * (size_t)p->S * (size_t)p->N;
    if (class_probing() && bytes >= ((size_t)1 << 30) && p->stop > p->start && p->role_tunes < kRoleTunes) {
        ++p->role_tunes;
        swap = 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        float best[4] = {-1.f, -1.f, -1.f, -1.f};
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            for (int rep = 0; rep < 2; ++rep)
                for (int w = 0; w < 4; ++w) {
                    p->fd4_force = w;
                    auto steps = build_steps(p, in, out, tail, io);
                    (void)hipEventRecord(e0, p->stream);
                    const int rc = run_steps(steps, p->stream);
                    (void)hipEventRecord(e1, p->stream);
                    float ms = 0.f;
                    if (rc == PBH_OK && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess &&
                        (best[w] < 0 || ms < best[w]))
                        best[w] = ms;
                }
            p->fd4_force = -1;
            for (int w = 1; w < 4; ++w)
                if (best[w] > 0 && best[swap] > 0 && best[w] < best[swap]) swap = w;
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipGetLastError();
        static const bool trace = getenv("PBH_TRACE_ALLOC") != nullptr;
        if (trace) fprintf(stderr, "[pbhip] four-pass roles for (%p, %p): %.4f / %.4f ms (Q4 in work / work2, inverse in place), %.4f / %.4f (inverse out of place) -> %d\n",
                           (const void*)in, okey, best[0], best[1], best[2], best[3], swap);
    }
    auto& e = p->role_cache[p->role_next++ & 7];
    e.in = in;
    e.out = okey;
    e.swap = swap;
    p->role_last = swap;
    return swap;
}
#endif

std::vector<Step> build_steps(pbh_plan* p, const cf* in, cf* out, DetectTail tail, IoLayout io) {
    std::vector<Step> steps;
    const int S = p->S;
    if (p->bsL && p->cfilt) {
        const int64_t N = p->N, L = p->bsL;
        pbh_plan* q = p->cfilt;
        if (q->N1 > 1 && resolved_variant(q) == PBH_VARIANT_PLANAR5) {
            // the convolution plan's de-interleave pass reads the N samples that exist and zero-fills the rest
            IoLayout pad_io;
            pad_io.in_valid = N;
            return build_steps(q, in, out, DetectTail(), pad_io);
        }
        cf* pad = p->cf_in;   // 3-pass and single-tile plans read their input directly: pad it first
        steps.push_back({"k_pad", [=](hipStream_t st) {
            HIPCHECK(hipMemcpyAsync(pad, in, sizeof(cf) * (size_t)N * S, hipMemcpyDeviceToDevice, st));
            HIPCHECK(hipMemsetAsync(pad + N * S, 0, sizeof(cf) * (size_t)(L - N) * S, st));
            return (int)PBH_OK;
        }});
        for (auto& s2 : build_steps(q, pad, out)) steps.push_back(s2);
        return steps;
    }
    if (p->bsL) {
        const int64_t N = p->N, L = p->bsL, start = p->start, stop = p->stop;
        const int npol = p->npol;
        cf *a = p->bs_a, *conv = p->bs_conv;
        const cf *b = p->bs_b, *H = p->chirp;
        auto grid = [](int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
        steps.push_back({"k_bs_pre", [=](hipStream_t st) {
            hipLaunchKernelGGL(k_bs_pre, dim3(grid(L * S)), dim3(256), 0, st, in, b, a, N, L, S);
            HIPCHECK(hipGetLastError());
            return (int)PBH_OK;
        }});
        for (auto& s : build_steps(p->sub, a, conv)) steps.push_back(s);
        steps.push_back({"k_bs_mid", [=](hipStream_t st) {
            hipLaunchKernelGGL(k_bs_mid, dim3(grid(L * S)), dim3(256), 0, st, (const cf*)conv, H, a, N, L, S, npol);
            HIPCHECK(hipGetLastError());
            return (int)PBH_OK;
        }});
        for (auto& s : build_steps(p->sub, a, conv)) steps.push_back(s);
        steps.push_back({"k_bs_post", [=](hipStream_t st) {
            if (stop <= start) return (int)PBH_OK;
            hipLaunchKernelGGL(k_bs_post, dim3(grid((stop - start) * S)), dim3(256), 0, st, (const cf*)conv, b, out,
                               start, stop, S);
            HIPCHECK(hipGetLastError());
            return (int)PBH_OK;
        }});
        return steps;
    }
    if (p->N1 == 1 && p->work && single_planar_ok(p) && io.in_layout == PBH_LAYOUT_SAMPLE_MAJOR &&
        io.out_layout == PBH_LAYOUT_SAMPLE_MAJOR) {
        // One-tile blocks with many series (what a channeliser with long segments hands over: 2^14 samples x thousands of
        // narrow channels).  k_small would read 8-byte pieces of every 128-byte input line once per series; here the two
        // layout passes move full lines and the transform runs on contiguous rows of the planar copy.
        const int64_t N = p->N, start = p->start, stop = p->stop;
        const int64_t nvalid = io.in_valid >= 0 ? io.in_valid : N;
        const int M = (int)N, FR = kTilePoints / M;
        cf* work = p->work;
        unsigned* ctr0 = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
        const double* mft = io.mix_ft;
        steps.push_back({"k_deinterleave", [=](hipStream_t st) {
            HIPCHECK(hipMemsetAsync(ctr0, 0, kCounterBytes, st));
            return launch_deinterleave(in, work, N, S, nvalid, st, 0, mft);
        }});
        RowParams rp{work, p->chirp, p->tw16k, (int64_t)S, 1, p->npol, 0, ctr0 + 2};
        rp.cdiv = FR > 1 ? p->npol : 1;
        steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_row(M, rp, st); }});
        if (tail.out) {   // detect tail (can_fuse_detect): the planar copy holds the dedispersed rows in time order
            const int nchan = p->nchan, npol = p->npol, mode = tail.mode, ns = tail.nscrunch;
            real* dout = tail.out;
            if (ns == 1) {
                steps.push_back({"k_reint_detect", [=](hipStream_t st) {
                    return launch_reint_detect(work, dout, start, stop, S, nchan, npol, mode, N, st);
                }});
            } else {
                const int64_t nout = (stop - start) / ns;
                steps.push_back({"k_detect_planar", [=](hipStream_t st) {
                    if (nout <= 0) return (int)PBH_OK;
                    hipLaunchKernelGGL(k_detect_planar, dim3((unsigned)((nout + 3) / 4), (unsigned)nchan), dim3(256), 0, st,
                                       (const cf*)work, dout, N, start, nout, nchan, npol, mode, ns);
                    HIPCHECK(hipGetLastError());
                    return (int)PBH_OK;
                }});
            }
            return steps;
        }
        const int64_t orow = io.out_row_elems;
        const auto pp = io.part_ptr;
        const auto pr = io.part_row;
        steps.push_back({"k_reinterleave", [=](hipStream_t st) {
            return launch_reinterleave_parts(work, out, start, stop, S, N, st, orow, pp, pr);
        }});
        return steps;
    }
    if (p->N1 == 1) {
        SmallParams sp{in, out, p->chirp, p->tw16k, S, p->npol, p->start, p->stop, -1, (real)1};
        const int M = (int)p->N;
        steps.push_back({"k_small", [=](hipStream_t st) { return launch_small(M, sp, st); }});
        return steps;
    }
    if (p->mixed) {
        // 7-smooth length: de-interleave, [P-point stage], Q-point column pass, fused rows, and back (mixed_kernels.hpp)
        const int64_t N = p->N, start = p->start, stop = p->stop;
        const int N1 = p->N1, N2 = p->N2, P = p->P, Q = N1 / P;
        cf* work = p->work;
        const bool in_sm = io.in_layout == PBH_LAYOUT_SERIES_MAJOR, out_sm = io.out_layout == PBH_LAYOUT_SERIES_MAJOR;
        const int64_t nvalid = io.in_valid >= 0 ? io.in_valid : N;
        if (!in_sm) {
            const double* mft = io.mix_ft;
            steps.push_back({"k_deinterleave", [=](hipStream_t st) {
                return launch_deinterleave(in, work, N, S, nvalid, st, 0, mft);
            }});
        }
        auto role_a = [=](const cf* ld, int64_t ldp, cf* stp, int64_t stpl, int64_t k0, int64_t k1, int64_t shift) {
            return mix_role_a(p, ld, ldp, stp, stpl, k0, k1, shift);
        };
        auto role_b = [=](const cf* ld, int64_t ldp, cf* stp, int64_t stpl, int64_t k0, int64_t k1, int64_t shift) {
            return mix_role_b(p, ld, ldp, stp, stpl, k0, k1, shift);
        };
        const cf* src = in_sm ? in : work;          // a series-major input is read by the first column pass, out of place
        int64_t splane = in_sm ? io.in_pitch : N;
        unsigned* ctr = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);   // tile counters: 2 = rows, 3..6 = the column passes
        const pbh_plan* cp = p;
        if (P > 1) {
            if (mix_radix_p(P)) {
                const cf* rsrc = src;
                const int64_t rplane = splane;
                steps.push_back({"k_radix_fwd", [=](hipStream_t st) {
                    return launch_radix<-1>(P, rsrc, rplane, work, N, S, N, N2, N1, 0, N, st);
                }});
            } else {
                MixParams a = role_a(src, splane, work, N, 0, N, 0);
                a.counter = ctr + 3;
                steps.push_back({"k_radix_fwd", [=](hipStream_t st) { return launch_colmix<-1>(a, st); }});
            }
            src = work;
            splane = N;
        }
        {
            MixParams b = role_b(src, splane, work, N, 0, N, 0);
            b.counter = ctr + 4;
            steps.push_back({"k_col_fwd", [=](hipStream_t st) { return launch_colmix<-1>(b, st); }});
        }
        RowParams rp{work, p->chirp, p->tw16k, (int64_t)S * N1, N1, p->npol, 0, ctr + 2};
        if (p->rowmix) {
            steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowmix(cp, work, st); }});
        } else
#ifndef PBH_F64
        if (p->has_phase && row_phase_enabled()) {
            RowpParams rpp{work, p->chirp_phase, p->tw16k, p->nchan, N1, p->npol, (real)(1.0 / (double)N), ctr + 2};
            rpp.phase16 = p->phase16;
            steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowp(N2, rpp, st); }});
        } else
#endif
        steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_row(N2, rp, st); }});
        // the last inverse pass is the one whose rows are times: it crops, and writes a series-major output itself
        const bool direct_out = out_sm && !tail.out;
        cf* dst = direct_out ? out : work;
        const int64_t dplane = direct_out ? io.out_pitch : N, dshift = direct_out ? start : 0;
        {
            MixParams b = P > 1 ? role_b(work, N, work, N, 0, N, 0) : role_b(work, N, dst, dplane, start, stop, dshift);
            b.counter = ctr + 5;
            steps.push_back({"k_col_inv", [=](hipStream_t st) { return launch_colmix<+1>(b, st); }});
        }
        if (P > 1) {
            if (mix_radix_p(P)) {
                // (k_radix_p stores time t at t - keep0: cropped when it writes the caller's series-major array, whole in place)
                const int64_t k0 = direct_out ? start : 0, k1 = direct_out ? stop : N;
                steps.push_back({"k_radix_inv", [=](hipStream_t st) {
                    return launch_radix<+1>(P, work, N, dst, dplane, S, N, N2, N1, k0, k1, st);
                }});
            } else {
                MixParams a = role_a(work, N, dst, dplane, start, stop, dshift);
                a.counter = ctr + 6;
                steps.push_back({"k_radix_inv", [=](hipStream_t st) { return launch_colmix<+1>(a, st); }});
            }
        }
        if (tail.out && tail.nscrunch == 1 && reint_detect_ok(S, p->npol, tail.mode, N)) {
            const int nchan = p->nchan, npol = p->npol, mode = tail.mode;
            real* dout = tail.out;
            steps.push_back({"k_reint_detect", [=](hipStream_t st) {
                return launch_reint_detect(work, dout, start, stop, S, nchan, npol, mode, N, st);
            }});
        } else if (tail.out) {
            const int nchan = p->nchan, npol = p->npol;
            const int64_t nout = (stop - start) / tail.nscrunch;
            steps.push_back({"k_detect_planar", [=](hipStream_t st) {
                if (nout <= 0) return (int)PBH_OK;
                hipLaunchKernelGGL(k_detect_planar, dim3((unsigned)((nout + 3) / 4), (unsigned)nchan), dim3(256), 0, st,
                                   (const cf*)work, tail.out, N, start, nout, nchan, npol, tail.mode, tail.nscrunch);
                HIPCHECK(hipGetLastError());
                return (int)PBH_OK;
            }});
        } else if (!out_sm) {
            const int64_t orow = io.out_row_elems;
            const auto pp = io.part_ptr;
            const auto pr = io.part_row;
            steps.push_back({"k_reinterleave", [=](hipStream_t st) {
                return launch_reinterleave_parts(work, out, start, stop, S, N, st, orow, pp, pr);
            }});
        }
        if (!steps.empty()) {
            unsigned* ctr0 = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
            auto first = steps[0].launch;
            steps[0].launch = [=](hipStream_t st) {
                HIPCHECK(hipMemsetAsync(ctr0, 0, kCounterBytes, st));
                return first(st);
            };
        }
        return steps;
    }
    const int variant = resolved_variant(p);
    BigTwiddle tw{p->tw_hi, p->tw_lo, p->tw_shift, p->N - 1};
    tw.nmod = is_pow2(p->N) ? 0 : p->N;
    const int64_t ncols = (int64_t)S * p->N2;
    ColSide planar{LAYOUT_PLANAR, p->N, p->N2};
    ColSide inter{LAYOUT_INTERLEAVED, 0, (int64_t)p->N2 * S};
    const int N1 = p->N1, N2 = p->N2;
    cf* work = p->work;

    // one series: the (nsample, 1) block is its own series-major form -- no layout passes are needed at all
    // (matters for long blocks, where the 3-pass variant is not available)
    if (S == 1 && p->P > 1 && !tail.out && io.in_valid < 0 && io.in_layout == PBH_LAYOUT_SAMPLE_MAJOR &&
        io.out_layout == PBH_LAYOUT_SAMPLE_MAJOR) {
        io.in_layout = io.out_layout = PBH_LAYOUT_SERIES_MAJOR;
        io.in_pitch = p->N;
        io.out_pitch = p->stop - p->start;
    }
    const bool in_sm = io.in_layout == PBH_LAYOUT_SERIES_MAJOR, out_sm = io.out_layout == PBH_LAYOUT_SERIES_MAJOR;
    if (variant == PBH_VARIANT_PLANAR5 || in_sm || out_sm) {
        const int64_t N = p->N, start = p->start, stop = p->stop;
        // A radix-2 split pays only while its stage rides in the layout passes.  Series-major arrays and the detect
        // tail have no such pass on one side or both: they run the unsplit column passes (64-byte pieces) and the
        // row pass finds its chirp row through the split order the chirp was stored in.
        const bool unsplit = p->P == 2 && (in_sm || out_sm || tail.out) && N2 == kTilePoints && N1 <= kTilePoints &&
                             N2 % (kTilePoints / N1) == 0;
        const int P = unsplit ? 1 : p->P, Q = N1 / P;
        const int chirp_split = unsplit ? p->P : 1;
        const bool fuse_radix = P > 1 && !in_sm && !out_sm && radix_layout_ok(S, P, N, N2);
        const int64_t nvalid = io.in_valid >= 0 ? io.in_valid : N;
        // Four-pass schedule (fd4_kernels.hpp): pass 1 reads the caller's block itself and leaves Q4 order in `work`, the row
        // pass writes planar rows into the second work buffer, where the inverse column pass then works in place.
        cf* fdB = nullptr;
#ifndef PBH_F64
        if (fd4_ok(p) && !in_sm && !out_sm && !unsplit && !fuse_radix && !io.mix_ft && io.in_valid < 0 && !depth_mode() && !oop_mode() &&
            colp_mode() != 0)
            fdB = ensure_work2(p, in, tail.out || !io.part_ptr.empty() ? nullptr : out,   // (a split output: `out` is its first part only)
                               p->stop > p->start ? sizeof(cf) * (size_t)S * (size_t)(p->stop - p->start) : 0);
#endif
        if (fdB) {
        } else if (fuse_radix) {
            const int Pf = P;
            steps.push_back({"k_deinterleave", [=](hipStream_t st) {
                return launch_deint_radix(S, Pf, in, work, N, N2, N1, nvalid, st);
            }});
        } else if (!in_sm) {
            const double* mft = io.mix_ft;
            steps.push_back({"k_deinterleave", [=](hipStream_t st) {
                return launch_deinterleave(in, work, N, S, nvalid, st, 0, mft);
            }});
        }
        ColParams c1{work, work, planar, planar, LAYOUT_PLANAR, 0, 0, S, N2, ncols, 0, tw, p->tw16k, 0, N, 0};
        // (very short column transforms, Q < 64, leave the persistent kernel no butterflies to hide its memory
        //  traffic behind -- and its inverse form spills there: one-tile workgroups are as fast or faster)
        const bool colp = P > 1 || in_sm || out_sm ||
                          (colp_mode() != 0 && Q >= 64 && Q <= kTilePoints && N2 % (kTilePoints / Q) == 0 && N < (1LL << 31));
        unsigned* ctr = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);  // two tile counters behind the table
        ColpParams cp1{work, N, S, N2, tw, p->tw16k, 0, N, 0, ctr};
        bool det_done = false; // the detect tail ran inside the inverse column pass
        cf* workB = nullptr;   // ping-pong schedule (oop_mode): column pass A -> B, row pass B -> A, column pass A -> B
#ifndef PBH_F64
        if (oop_mode() && !depth_mode() && P == 1 && colp && !in_sm && !out_sm && !fuse_radix && p->has_phase && row_phase_enabled() &&
            p->phase16 && N2 == kTilePoints)
            workB = ensure_work2(p);
#endif
        if (workB) {
            cp1.ld = work;
            cp1.ld_plane = N;
            cp1.data = workB;
        }
        if (in_sm && P == 1) {   // pass 1 reads the caller's series-major input directly: no de-interleave pass
            cp1.ld = in;
            cp1.ld_plane = io.in_pitch;
        }
        if (P > 1) {
            cp1.P = P;
            if (!fuse_radix) {
                // a series-major input is read by the radix stage (out of place into the workspace), not by the column pass
                const cf* rsrc = in_sm ? in : work;
                const int64_t rplane = in_sm ? io.in_pitch : N;
                steps.push_back({"k_radix_fwd", [=](hipStream_t st) {
                    return launch_radix<-1>(P, rsrc, rplane, work, N, S, N, N2, N1, 0, N, st);
                }});
            }
        }
        RowParams rp{work, p->chirp, p->tw16k, (int64_t)S * N1, N1, p->npol, p->perm_w,
                     reinterpret_cast<unsigned*>(p->tw16k + kTwTable) + 2};
        rp.cP = chirp_split;
        // rows outside [start, stop) are never read by k_reinterleave: skip their stores
        ColParams c3{work, work, planar, planar, LAYOUT_PLANAR, 0, 0, S, N2, ncols, 0, tw, p->tw16k, start, stop, 0};
        ColpParams cp3{work, N, S, N2, tw, p->tw16k, start, stop, 0, ctr + 1};
#ifndef PBH_F64
        // Depth-first schedule (PBH_DEPTH=1 per series, 2 per channel): the three middle passes run unit by unit, so a
        // unit's planar intermediate (134 MB per series at N = 2^24) is still in the 256 MB Infinity Cache when the
        // next pass reads it.
        const int depth = depth_mode();
        if (depth && P == 1 && colp && !in_sm && !out_sm && p->has_phase && row_phase_enabled() &&
            3 * S + 3 <= kCounters) {
            const int unit = depth == 1 ? 1 : p->npol;
            const int npol = p->npol;
            int ci = 3;
            for (int s0 = 0; s0 < S; s0 += unit) {
                ColpParams a = cp1;
                a.data = work + (int64_t)s0 * N;
                a.S = unit;
                a.counter = ctr + ci++;
                steps.push_back({"k_col_fwd", [=](hipStream_t st) { return launch_colq<OP_FWD_TW>(Q, a, st); }});
                RowpParams r{work + (int64_t)s0 * N, p->chirp_phase + (int64_t)(s0 / npol) * N, p->tw16k, 1, N1, unit,
                             (real)(1.0 / (double)p->N), ctr + ci++};
                r.phase16 = p->phase16;
                steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowp(N2, r, st); }});
                ColpParams b = cp3;
                b.data = work + (int64_t)s0 * N;
                b.S = unit;
                b.counter = ctr + ci++;
                steps.push_back({"k_col_inv", [=](hipStream_t st) { return launch_colq<OP_TW_INV>(Q, b, st); }});
            }
        } else {
#endif
#ifndef PBH_F64
        if (fdB) {
            // roles of the two work buffers (of opposite allocation class where that could be arranged, ensure_work2): the
            // Q4 intermediate X should be the one whose class differs from the caller's input's, so that every pass streams
            // between allocations of different class, in -> X -> Y -> X -> out; fd4_roles finds out which one that is
            cf *X = work, *Y = fdB;
            const int roles = fd4_roles(p, in, out, tail, io);
            if (roles & 1) std::swap(X, Y);
            const ColfdParams fp{in, X, S, N2, tw, p->tw16k};
            steps.push_back({"k_col_fwd", [=](hipStream_t st) { return launch_colfd(Q, fp, st); }});
            const RowqParams rq{X, Y, p->chirp_phase, p->tw16k, S, N1, p->npol, (real)(1.0 / (double)p->N)};
            steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowq(rq, st); }});
            if (roles & 2) {           // inverse column pass out of place, Y -> X
                cp3.ld = Y;
                cp3.ld_plane = N;
                cp3.data = X;
                fdB = X;               // ... where the time-ordered series then are
            } else {
                cp3.data = Y;
                fdB = Y;
            }
        } else {
#endif
        steps.push_back({"k_col_fwd", [=](hipStream_t st) {
            return colp ? launch_colq<OP_FWD_TW>(Q, cp1, st) : launch_col<OP_FWD_TW>(N1, c1, st);
        }});
#ifndef PBH_F64
        if (p->has_phase && row_phase_enabled()) {
            RowpParams rpp{work, p->chirp_phase, p->tw16k, p->nchan, N1, p->npol, (real)(1.0 / (double)p->N), ctr + 2};
            rpp.cP = chirp_split;
            rpp.phase16 = p->phase16;
            static const bool otf = [] { const char* e = diag_env("PBH_ROW_OTF"); return e ? atoi(e) != 0 : false; }();
            if (otf && p->phase16 && chirp_split == 1 && is_pow2(p->N)) {   // phase computed in the kernel (A/B switch)
                rpp.chan_freq = p->chan_freq;
                rpp.coeff = p->gen_coeff;
                rpp.inv_ndt = p->gen_inv_ndt;
                rpp.inv_ref = p->gen_inv_ref;
                rpp.N = p->N;
            }
            if (workB) {
                rpp.data = workB;
                rpp.out = work;
            }
            steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowp(N2, rpp, st); }});
        } else
#endif
        steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_row(N2, rp, st); }});
#ifndef PBH_F64
        }
#endif
        if (workB) {
            cp3.ld = work;
            cp3.ld_plane = N;
            cp3.data = workB;
        }
        if (out_sm && !tail.out && P == 1) {   // pass 3 writes the caller's series-major output directly, cropped
            cp3.ld = work;
            cp3.ld_plane = N;
            cp3.data = out;
            cp3.plane = io.out_pitch;
            cp3.st_shift = start;
        }
        if (P > 1) {   // the time order only exists after the inverse radix-P stage: no crop in the column pass
            cp3.P = P;
            cp3.crop_start = 0;
            cp3.crop_stop = N;
        }
#ifndef PBH_F64
        // Detect tail inside this pass: the dedispersed voltages are never stored (saves their write and the read of
        // k_detect_planar: 2 x 8 of the 60 bytes per sample of configs[4]).
        if (tail.out && tail.mode <= PBH_DETECT_STOKES_I && detect_in_colq() && colp && P == 1 && Q >= 64 && Q <= 1024 && PBH_R == 32 && !workB &&
            N2 % tail.nscrunch == 0 && tail.nscrunch % (kTilePoints / Q) == 0 && (stop - start) / tail.nscrunch > 0) {
            const int ns = tail.nscrunch, nchan = p->nchan, npol = p->npol, mode = tail.mode;
            const size_t npart = (size_t)S * (size_t)(N2 / 16) * (size_t)Q, nside = (size_t)S * (size_t)(N2 / ns) * (size_t)Q;
            real* part = ensure_det_part(p, (npart + nside) * sizeof(real));
            if (part) {
                cp3.det_part = part;
                cp3.det_side = part + npart;
                cp3.det_ns = ns;
                const int64_t nout = (stop - start) / ns;
                real* dout = tail.out;
                const ColpParams cpd = cp3;
                steps.push_back({"k_col_inv", [=](hipStream_t st) { return launch_colq_det(Q, cpd, st); }});
                steps.push_back({"k_detect_reduce", [=](hipStream_t st) {
                    const int nj = ns / 16 + 1, parts = nj <= 9 ? 1 : (nj <= 129 ? 4 : 16), xw = 256 / parts;
                    hipLaunchKernelGGL(k_detect_reduce, dim3((unsigned)((Q + xw - 1) / xw), (unsigned)(N2 / ns), (unsigned)nchan), dim3(256), 0, st,
                                       (const real*)cpd.det_part, (const real*)cpd.det_side, dout, N2, Q, (int)PBH_R, ns, start, nout, nchan, npol, mode,
                                       parts, 16, npol);
                    HIPCHECK(hipGetLastError());
                    return (int)PBH_OK;
                }});
                det_done = true;
            }
        }
        // Round 4: all four Stokes parameters the same way -- the tile is 8 columns x both polarisations of a channel, the pass
        // leaves |a|^2, |b|^2, Re / Im conj(a) b summed over every 8 columns (4 x 4 KiB per tile instead of 128 KiB of voltages)
        // and the reducer assembles I, Q, U, V.  (N1 = 1024; scrunch factors that are multiples of 8 and divide a row.)
        if (!det_done && tail.out && (tail.mode == PBH_DETECT_STOKES_LINEAR || tail.mode == PBH_DETECT_STOKES_CIRCULAR) && p->npol == 2 &&
            detect_in_colq() && colp && P == 1 && Q == 1024 && PBH_R == 32 && N2 % tail.nscrunch == 0 && tail.nscrunch % 8 == 0 &&
            (stop - start) / tail.nscrunch > 0 && N * (int64_t)sizeof(cf) < (1LL << 31)) {
            const int ns = tail.nscrunch, nchan = p->nchan, npol = p->npol, mode = tail.mode;
            const size_t npart = (size_t)(S / 2) * 4 * (size_t)(N2 / 8) * (size_t)Q, nside = (size_t)(S / 2) * 4 * (size_t)(N2 / ns) * (size_t)Q;
            real* part = ensure_det_part(p, (npart + nside) * sizeof(real));
            if (part) {
                cp3.det_part = part;
                cp3.det_side = part + npart;
                cp3.det_ns = ns;
                const int64_t nout = (stop - start) / ns;
                real* dout = tail.out;
                const ColpParams cpd = cp3;
                steps.push_back({"k_col_inv", [=](hipStream_t st) { return launch_colq_det4(Q, cpd, st); }});
                steps.push_back({"k_detect_reduce", [=](hipStream_t st) {
                    const int nj = ns / 8 + 1, parts = nj <= 9 ? 1 : (nj <= 129 ? 4 : 16), xw = 256 / parts;
                    hipLaunchKernelGGL(k_detect_reduce, dim3((unsigned)((Q + xw - 1) / xw), (unsigned)(N2 / ns), (unsigned)nchan), dim3(256), 0, st,
                                       (const real*)cpd.det_part, (const real*)cpd.det_side, dout, N2, Q, (int)PBH_R, ns, start, nout, nchan, npol, mode,
                                       parts, 8, 4);
                    HIPCHECK(hipGetLastError());
                    return (int)PBH_OK;
                }});
                det_done = true;
            }
        }
        if (!det_done)
#endif
        steps.push_back({"k_col_inv", [=](hipStream_t st) {
            return colp ? launch_colq<OP_TW_INV>(Q, cp3, st) : launch_col<OP_TW_INV>(N1, c3, st);
        }});
#ifndef PBH_F64
        }
#endif
        const cf* wlast = fdB ? fdB : (workB ? workB : work);   // where the last column pass left the time-ordered series
        const bool fuse_out = fuse_radix && !tail.out;   // the detect tail reads time-ordered planar data
        if (P > 1 && !fuse_out) {
            if (out_sm && !tail.out) {   // the inverse stage writes the caller's series-major output, cropped
                const int64_t opitch = io.out_pitch;
                steps.push_back({"k_radix_inv", [=](hipStream_t st) {
                    return launch_radix<+1>(P, work, N, out, opitch, S, N, N2, N1, start, stop, st);
                }});
            } else {
                steps.push_back({"k_radix_inv", [=](hipStream_t st) {
                    return launch_radix<+1>(P, work, N, work, N, S, N, N2, N1, 0, N, st);
                }});
            }
        }
        if (det_done) {
        } else if (tail.out && tail.nscrunch == 1 && reint_detect_ok(S, p->npol, tail.mode, N)) {
            const int nchan = p->nchan, npol = p->npol, mode = tail.mode;
            real* dout = tail.out;
            steps.push_back({"k_reint_detect", [=](hipStream_t st) {
                return launch_reint_detect(wlast, dout, start, stop, S, nchan, npol, mode, N, st);
            }});
        } else if (tail.out) {
            const int nchan = p->nchan, npol = p->npol;
            const int64_t nout = (stop - start) / tail.nscrunch;
            steps.push_back({"k_detect_planar", [=](hipStream_t st) {
                if (nout <= 0) return (int)PBH_OK;
                hipLaunchKernelGGL(k_detect_planar, dim3((unsigned)((nout + 3) / 4), (unsigned)nchan), dim3(256), 0, st,
                                   (const cf*)wlast, tail.out, N, start, nout, nchan, npol, tail.mode, tail.nscrunch);
                HIPCHECK(hipGetLastError());
                return (int)PBH_OK;
            }});
        } else if (fuse_out) {
            steps.push_back({"k_reinterleave", [=](hipStream_t st) {
                return launch_reint_radix(S, P, work, out, N, N2, N1, start, stop, st);
            }});
        } else if (!out_sm) {
            const int64_t orow = io.out_row_elems;
            const auto pp = io.part_ptr;
            const auto pr = io.part_row;
            steps.push_back({"k_reinterleave", [=](hipStream_t st) {
                return launch_reinterleave_parts(wlast, out, start, stop, S, N, st, orow, pp, pr);
            }});
        }
    } else {
        const int el = (variant == PBH_VARIANT_BLOCK3) ? LAYOUT_BLOCK : LAYOUT_INTERLEAVED;
        const int sb = 4, lo = block_lane_order();
        ColParams c1{in, work, inter, planar, el, sb, lo, S, N2, ncols, 0, tw, p->tw16k, 0, p->N, 0};
        steps.push_back({"k_col_fwd", [=](hipStream_t st) { return launch_col<OP_FWD_TW>(N1, c1, st); }});
        RowParams rp{work, p->chirp, p->tw16k, (int64_t)S * N1, N1, p->npol, p->perm_w,
                     reinterpret_cast<unsigned*>(p->tw16k + kTwTable) + 2};
#ifndef PBH_F64
        if (p->has_phase && row_phase_enabled()) {
            RowpParams rpp{work, p->chirp_phase, p->tw16k, p->nchan, N1, p->npol, (real)(1.0 / (double)p->N),
                           reinterpret_cast<unsigned*>(p->tw16k + kTwTable) + 2};
            rpp.phase16 = p->phase16;
            steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_rowp(N2, rpp, st); }});
        } else
#endif
        steps.push_back({"k_row_fused", [=](hipStream_t st) { return launch_row(N2, rp, st); }});
        ColParams c3{work, out, planar, inter, el, sb, lo, S, N2, ncols, 0, tw, p->tw16k,
                     p->start, p->stop, p->start * S};
        steps.push_back({"k_col_inv", [=](hipStream_t st) { return launch_col<OP_TW_INV>(N1, c3, st); }});
    }
    // the persistent kernels' tile counters (three words behind the twiddle table) are zeroed once per
    // run, ahead of the first kernel, instead of once per kernel
    if (!steps.empty()) {
        unsigned* ctr0 = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
        auto first = steps[0].launch;
        steps[0].launch = [=](hipStream_t st) {
            HIPCHECK(hipMemsetAsync(ctr0, 0, kCounterBytes, st));
            return first(st);
        };
    }
    return steps;
}

bool can_fuse_detect(const pbh_plan* p, int nscrunch, int mode) {
    // multi-pass planar plans, and one-tile plans that run layout pass + planar rows + layout pass (many series)
    const bool planar = (p->N1 > 1 && resolved_variant(p) == PBH_VARIANT_PLANAR5) || (p->N1 == 1 && p->work && !p->bsL && single_planar_ok(p));
    if (!planar || p->nchan > 65535) return false;
    if (nscrunch % 64 == 0) return true;
    // full time resolution: the last layout pass of the planar pipelines (2^k, m 2^k and 7-smooth lengths) detects (launch_reint_detect)
    return nscrunch == 1 && !p->bsL && reint_detect_ok(p->S, p->npol, mode, p->N);
}

int run_steps(std::vector<Step>& steps, hipStream_t st) {
    for (auto& s : steps) PBHCHECK(s.launch(st));
    return PBH_OK;
}


// ---- host <-> device transfers of caller memory --------------------------------------------------------------
// Large copies between pageable caller memory and the device do NOT go to hipMemcpyAsync directly: the
// runtime pins such ranges on the fly and keeps the pins in a cache, and a cached pin of host addresses
// the caller's allocator has since recycled (glibc hands freed heap back and regrows it) made a later
// device-to-host copy die with "Memory access fault ... Write access to a read-only page" -- one run
// in ~15 of the GPU test suite, at a reproducible test.  The library therefore moves such data through
// two pinned bounce buffers of its own (per thread and device), chunk by chunk, the CPU copy of one
// chunk overlapping the DMA of the other.  Both functions return when the transfer is complete.
// pbh_dedisperse_stream pins its buffers explicitly (hipHostRegister) for the duration of the call instead.
namespace {
constexpr size_t kBounceBytes = 8u << 20;
constexpr size_t kBounceMin = 256u << 10;   // smaller copies: the runtime stages them itself, nothing is pinned
struct Bounce {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
thread_local Bounce g_bounce[16];
void release_bounce_buffers() {   // pbh_trim: 16 MiB of pinned memory per (thread, device) that used a host transfer
    for (Bounce& b : g_bounce)
        for (int i = 0; i < 2; ++i) {
            if (b.ev[i]) { (void)hipEventSynchronize(b.ev[i]); (void)hipEventDestroy(b.ev[i]); b.ev[i] = nullptr; }
            if (b.buf[i]) { (void)hipHostFree(b.buf[i]); b.buf[i] = nullptr; }
        }
}
Bounce* bounce_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Bounce& b = g_bounce[dev];
    if (!b.buf[0]) {
        for (int i = 0; i < 2; ++i) {
            if (hipHostMalloc(&b.buf[i], kBounceBytes, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&b.ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                release_bounce_buffers();
                return nullptr;
            }
        }
    }
    return &b;
}
}  // namespace

hipError_t xfer_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
    Bounce* b = bytes >= kBounceMin ? bounce_for_current_device() : nullptr;
    if (!b) {
        hipError_t e = hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st);
        return e == hipSuccess ? hipStreamSynchronize(st) : e;
    }
    hipError_t e = hipSuccess;
    size_t off = 0;
    for (int i = 0; off < bytes && e == hipSuccess; ++i, off += kBounceBytes) {
        const int k = i & 1;
        const size_t c = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
        if (i >= 2) e = hipEventSynchronize(b->ev[k]);
        if (e != hipSuccess) break;
        memcpy(b->buf[k], (const char*)src_host + off, c);
        e = hipMemcpyAsync((char*)dst_dev + off, b->buf[k], c, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(b->ev[k], st);
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    return e != hipSuccess ? e : e2;
}

hipError_t xfer_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st) {
    Bounce* b = bytes >= kBounceMin ? bounce_for_current_device() : nullptr;
    if (!b) {
        hipError_t e = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st);
        return e == hipSuccess ? hipStreamSynchronize(st) : e;
    }
    hipError_t e = hipSuccess;
    const size_t nchunk = (bytes + kBounceBytes - 1) / kBounceBytes;
    auto len = [&](size_t i) { return bytes - i * kBounceBytes < kBounceBytes ? bytes - i * kBounceBytes : kBounceBytes; };
    for (size_t i = 0; i <= nchunk && e == hipSuccess; ++i) {
        if (i < nchunk) {   // request chunk i (its buffer was drained two rounds ago)
            const int k = (int)(i & 1);
            e = hipMemcpyAsync(b->buf[k], (const char*)src_dev + i * kBounceBytes, len(i), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(b->ev[k], st);
        }
        if (i >= 1 && e == hipSuccess) {   // drain chunk i-1 while chunk i is on the wire
            const int k = (int)((i - 1) & 1);
            e = hipEventSynchronize(b->ev[k]);
            if (e == hipSuccess) memcpy((char*)dst_host + (i - 1) * kBounceBytes, b->buf[k], len(i - 1));
        }
    }
    return e;
}

// Page-lock a caller's host range for asynchronous copies.  0: registered here (unregister afterwards); 1: the caller
// had pinned it already (both ends of the range are host-registered memory); -1: it cannot be pinned.
int pin_host_range(void* ptr, size_t bytes) {
    if (bytes == 0) return 1;
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();
    hipPointerAttribute_t a0, a1;
    const bool ok = hipPointerGetAttributes(&a0, ptr) == hipSuccess && a0.type == hipMemoryTypeHost &&
                    hipPointerGetAttributes(&a1, (char*)ptr + bytes - 1) == hipSuccess && a1.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return ok ? 1 : -1;
}

// Layout conversion of a device (nsample, nseries) array (DeviceArray.to_series_major / contiguous): one transposing
// pass with the pipeline's own layout kernels.
extern "C" int pbh_relayout(int device, void* hip_stream, int /*dtype: this build's*/, const void* in_dev, int in_layout, int64_t in_pitch,
                 void* out_dev, int out_layout, int64_t out_pitch, int64_t nsample, int nseries);

// Plain copy between caller (host) memory and device memory through the bounce buffers above; what
// pulsarbat_amd.DeviceArray uses for from_host() / get() so that no pageable caller memory is ever
// handed to the runtime by this package.  direction: 0 = host -> device, 1 = device -> host.  Blocking.
extern "C" int pbh_transfer(int device, void* hip_stream, void* dst, const void* src, size_t bytes, int direction) {
    if (bytes == 0) return PBH_OK;
    if (!dst || !src) return fail(PBH_ERR_INVALID, "NULL argument");
    if (direction != 0 && direction != 1) return fail(PBH_ERR_INVALID, "bad direction");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHECK(direction == 0 ? xfer_h2d(dst, src, bytes, st) : xfer_d2h(dst, src, bytes, st));
    return PBH_OK;
}

// Reader-side decode (include/pbhip.h): every byte the kernel will touch is bounds-checked here first.

// (the arithmetic lives in host_sched.hpp, where the CPU box runs it under the sanitizers)
int decode_span(const pbh_raw_layout_t* L, int64_t first, int64_t nsample, int nchan, int npol, size_t raw_bytes,
                DecodeSpan* sp) {
    const char* why = "bad argument";
    const int rc = pbh_host::decode_span(L, first, nsample, nchan, npol, raw_bytes, sp, &why);
    return rc == PBH_OK ? rc : fail(rc, why);
}

// draw: device copy of the raw stream from byte `skip` on (only offsets inside the checked span are formed)
int decode_launch(const unsigned char* draw, int64_t skip, const pbh_raw_layout_t* L, int64_t first, int64_t nsample, int nchan,
                         int npol, const unsigned char* dconj, float scale, void* out_dev, int out_layout,
                         int64_t out_pitch, hipStream_t st) {
    DecodeParams q;
    q.raw = draw;
    q.skip = skip;
    q.first = first;
    q.blk_t = L->blk_samples;
    q.blk_stride = L->blk_stride;
    q.hdr = L->hdr_bytes;
    q.e0 = L->elem0;
    q.st_t = L->stride_t;
    q.st_c = L->stride_c;
    q.st_p = L->stride_p;
    q.nbits = L->nbits;
    q.code = L->code;
    const int64_t at = L->stride_t < 0 ? -L->stride_t : L->stride_t;
    const int64_t ac = L->stride_c < 0 ? -L->stride_c : L->stride_c, ap = L->stride_p < 0 ? -L->stride_p : L->stride_p;
    const int64_t as = (nchan > 1 && npol > 1) ? (ac < ap ? ac : ap) : (nchan > 1 ? ac : (npol > 1 ? ap : INT64_MAX));
    q.lanes_t = at <= as;
    q.scale = scale;
    q.conj = dconj;
    q.n = nsample;
    q.nchan = nchan;
    q.npol = npol;
    q.out = (float*)out_dev;
    q.series_major = out_layout == PBH_LAYOUT_SERIES_MAJOR;
    q.pitch = out_pitch;
    const int64_t S = (int64_t)nchan * npol;
    q.ls = 0;
    while (q.ls < 6 && (1 << q.ls) < S) ++q.ls;
    q.npol_shift = is_pow2(npol) ? ilog2(npol) : -1;
    q.pair16 = L->nbits == 8 && L->ncomp == 2 && L->blk_stride % 2 == 0 && L->hdr_bytes % 2 == 0 &&
               ((uintptr_t)draw) % 2 == 0 && skip % 2 == 0;
    const int TS = 1 << q.ls, TT = kDecodeTile / TS;
    const dim3 grid((unsigned)((nsample + TT - 1) / TT), (unsigned)((S + TS - 1) / TS));
    const bool fast = L->blk_samples >= TT && q.npol_shift >= 0 && (L->ncomp == 1 || L->nbits != 8 || q.pair16);
#define LAUNCH(NC, NB)                                                                       \
    do {                                                                                      \
        if (fast)                                                                             \
            hipLaunchKernelGGL((k_decode<NC, NB, true>), grid, dim3(256), 0, st, q);          \
        else                                                                                  \
            hipLaunchKernelGGL((k_decode<NC, NB, false>), grid, dim3(256), 0, st, q);         \
    } while (0)
    if (L->ncomp == 2 && L->nbits == 8) LAUNCH(2, 8);
    else if (L->ncomp == 2 && L->nbits == 4) LAUNCH(2, 4);
    else if (L->ncomp == 2) LAUNCH(2, 2);
    else if (L->nbits == 8) LAUNCH(1, 8);
    else if (L->nbits == 4) LAUNCH(1, 4);
    else LAUNCH(1, 2);
#undef LAUNCH
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

extern "C" int pbh_decode(int device, void* hip_stream, const void* raw, size_t raw_bytes, int raw_loc, const pbh_raw_layout_t* L,
               int64_t first, int64_t nsample, int nchan, int npol, const unsigned char* conj_mask, float scale,
               void* out_dev, int out_layout, int64_t out_pitch) {
    if (!L || (!raw && raw_bytes) || (!out_dev && nsample)) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample < 0) return fail(PBH_ERR_INVALID, "bad dimensions");
    if (out_layout != PBH_LAYOUT_SAMPLE_MAJOR && out_layout != PBH_LAYOUT_SERIES_MAJOR)
        return fail(PBH_ERR_INVALID, "bad out_layout");
    if (out_layout == PBH_LAYOUT_SERIES_MAJOR && out_pitch < nsample) return fail(PBH_ERR_INVALID, "out_pitch < nsample");
    if (nsample == 0) return PBH_OK;
    DecodeSpan sp;
    PBHCHECK(decode_span(L, first, nsample, nchan, npol, raw_bytes, &sp));
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    void *sraw = nullptr, *sconj = nullptr;
    auto cleanup = [&] {
        if (sraw) (void)hipFree(sraw);
        if (sconj) (void)hipFree(sconj);
    };
    const unsigned char* draw = (const unsigned char*)raw;
    int64_t skip = 0;
    int rc = PBH_OK;
    if (raw_loc == PBH_HOST) {   // only the blocks that are read travel
        if ((rc = dev_alloc(nullptr, &sraw, sp.len)) != PBH_OK) return rc;
        if (xfer_h2d(sraw, (const char*)raw + sp.off, sp.len, st) != hipSuccess) {
            cleanup();
            return fail(PBH_ERR_HIP, "pbh_decode: host -> device copy failed");
        }
        draw = (const unsigned char*)sraw;
        skip = (int64_t)sp.off;
    }
    bool any_conj = false;
    if (conj_mask && L->ncomp == 2)
        for (int64_t i = 0; i < (int64_t)nchan * npol; ++i) any_conj |= conj_mask[i] != 0;
    if (any_conj) {
        if ((rc = dev_alloc(nullptr, &sconj, (size_t)nchan * npol)) != PBH_OK) {
            cleanup();
            return rc;
        }
        if (xfer_h2d(sconj, conj_mask, (size_t)nchan * npol, st) != hipSuccess) {
            cleanup();
            return fail(PBH_ERR_HIP, "pbh_decode: mask copy failed");
        }
    }
    rc = decode_launch(draw, skip, L, first, nsample, nchan, npol, (const unsigned char*)sconj, scale, out_dev, out_layout,
                       out_pitch, st);
    hipError_t e = hipSuccess;
    if (sraw || sconj) e = hipStreamSynchronize(st);   // staging is freed below: the kernel must be done with it
    cleanup();
    if (rc != PBH_OK) return rc;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_decode: ") + hipGetErrorString(e));
    return PBH_OK;
}

// ---- staging helpers ---------------------------------------------------------------------------------------
static int ensure_stage(pbh_plan* p, void** buf, size_t* have, size_t need) {
    if (*have >= need) return PBH_OK;
    if (*buf) {
        (void)hipFree(*buf);
        p->owned_bytes -= (int64_t)*have;
        *buf = nullptr;
        *have = 0;
    }
    PBHCHECK(dev_alloc(p, buf, need));
    *have = need;
    return PBH_OK;
}

// stage radices, W_L table and output permutation of a mixed-radix column transform of length L (mixed_kernels.hpp)
// rows: the table of a k_rowmix row transform: `perm` holds the INVERSE map, position -> bin -- the order the chirp rows are
// stored in (its inverse stages mirror the forward ones one by one)
static int build_mix_table(pbh_plan* p, int L, pbh_plan::MixTable* t, bool rows = false) {
    t->L = L;
    t->nstage = 0;
    int left = L;
    // radix 9 = 3 x 3 in registers: two levels per LDS round trip (PBH_MIX_SQUARE=0: radix 3 only, for A/B runs).  The same
    // for 25 = 5 x 5 needs more registers than two workgroups per CU leave (230-400 B/lane of scratch): not built.
    static const bool square = [] { const char* e = diag_env("PBH_MIX_SQUARE"); return e ? atoi(e) != 0 : true; }();
    if (square) while (left % 9 == 0) { t->radix[t->nstage++] = 9; left /= 9; }
    for (int r : {7, 5, 3}) while (left % r == 0) { if (t->nstage >= kMixMaxStages) return fail(PBH_ERR_UNSUPPORTED, "too many stages"); t->radix[t->nstage++] = r; left /= r; }
    while (left % 8 == 0) { t->radix[t->nstage++] = 8; left /= 8; }
    while (left % 4 == 0) { t->radix[t->nstage++] = 4; left /= 4; }
    while (left % 2 == 0) { t->radix[t->nstage++] = 2; left /= 2; }
    if (left != 1 || t->nstage > kMixMaxStages) return fail(PBH_ERR_UNSUPPORTED, "length is not 7-smooth");
    if (rows && t->nstage > 1) {
        // k_rowmix runs its LAST forward stage, the chirp and the first inverse stage as one round with the butterfly's
        // elements, chirp values and outputs all in registers: give that round the smallest radix of the list
        int best = 0;
        for (int j = 1; j < t->nstage; ++j)
            if (t->radix[j] < t->radix[best]) best = j;
        std::swap(t->radix[best], t->radix[t->nstage - 1]);
    }
    std::vector<cf> w(L);
    for (int i = 0; i < L; ++i) {
        const double a = -2.0 * M_PI * (double)i / (double)L;
        w[i] = make_cf((real)cos(a), (real)sin(a));
    }
    // X[k], k = u1 + r1 (u2 + r2 (u3 + ...)), ends at position u1 L/r1 + u2 L/(r1 r2) + ...
    std::vector<unsigned short> perm(L);
    for (int k = 0; k < L; ++k) {
        int rest = k, m = L, pos = 0;
        for (int j = 0; j < t->nstage; ++j) {
            const int r = t->radix[j];
            m /= r;
            pos += (rest % r) * m;
            rest /= r;
        }
        perm[k] = (unsigned short)pos;
    }
    if (rows) {
        std::vector<unsigned short> inv(L);
        for (int k = 0; k < L; ++k) inv[perm[k]] = (unsigned short)k;
        perm = inv;
    }
    int rc;
    if ((rc = dev_alloc(p, (void**)&t->wl, sizeof(cf) * L)) != PBH_OK) return rc;
    if ((rc = dev_alloc(p, (void**)&t->perm, sizeof(unsigned short) * L)) != PBH_OK) return rc;
    if (hipMemcpy(t->wl, w.data(), sizeof(cf) * L, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->perm, perm.data(), sizeof(unsigned short) * L, hipMemcpyHostToDevice) != hipSuccess)
        return fail(PBH_ERR_HIP, "hipMemcpy(mixed-radix tables) failed");
    return PBH_OK;
}

static int launch_rowfft(int M, cf* data, const cf* tw, int64_t nrows, hipStream_t st, bool inverse = false) {
    const int FR = kTilePoints / M;
    const int64_t tiles = (nrows + FR - 1) / FR;
    switch (M) {
#define X(m)                                                                                               \
    case m: {                                                                                              \
        auto kern = inverse ? k_rowfft<m, PBH_R, +1> : k_rowfft<m, PBH_R, -1>;                             \
        HIPCHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                     lds_tile_bytes<true>()));                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kTilePoints / PBH_R), lds_tile_bytes<true>(), st, \
                           data, tw, nrows);                                                               \
        break;                                                                                             \
    }
        FOR_ROW_M(X)
#undef X
        default: return fail(PBH_ERR_UNSUPPORTED, "row FFT length " + std::to_string(M));
    }
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

extern "C" int pbh_plan_create(pbh_plan** out, int device, int64_t nsample, int nchan, int npol, int dtype,
                               int64_t crop_start, int64_t crop_stop);

// Bluestein set-up: sub-plan over the ring length, b table, and the spectrum of the wrapped kernel
// conj(b) written straight into the sub-plan's chirp buffer in plan order (column pass + forward
// row pass leave bin k1 + N1 k2 at position k1*N2 + k2, which is the chirp layout).
// Forward transform, in plan (row) order and in place, of `nplanes` planar length-L arrays stored where the
// sub-plan q keeps its chirp: the spectrum a convolution plan multiplies by.  Split plans (q->P > 1, which
// includes every m * 2^k length) go through their own radix-P stage and row blocks, so the rows come out in
// the split order; unsplit ones through the one-tile column kernel.  `scratch` holds nplanes * L elements.
static int spectrum_in_plan_order(pbh_plan* q, int nplanes, cf* scratch, hipStream_t st) {
    const int64_t L = q->N;
    BigTwiddle tw{q->tw_hi, q->tw_lo, q->tw_shift, L - 1};
    tw.nmod = is_pow2(L) ? 0 : L;
    if (q->P > 1 && q->N2 % (kTilePoints / (q->N1 / q->P)) == 0) {
        PBHCHECK(launch_radix<-1>(q->P, q->chirp, L, q->chirp, L, nplanes, L, q->N2, q->N1, 0, L, st));
        ColpParams cp{q->chirp, L, nplanes, q->N2, tw, q->tw16k, 0, L, 0, nullptr};
        cp.P = q->P;
        PBHCHECK(launch_colq<OP_FWD_TW>(q->N1 / q->P, cp, st));
        return launch_rowfft(q->N2, q->chirp, q->tw16k, (int64_t)nplanes * q->N1, st);
    }
    ColSide planar{LAYOUT_PLANAR, L, q->N2};
    ColParams c1{q->chirp, q->chirp, planar, planar, LAYOUT_PLANAR, 0, 0, nplanes, q->N2, (int64_t)nplanes * q->N2, 0, tw,
                 q->tw16k, 0, L, 0};
    PBHCHECK(launch_col<OP_FWD_TW>(q->N1, c1, st));
    PBHCHECK(launch_rowfft(q->N2, q->chirp, q->tw16k, (int64_t)nplanes * q->N1, st));
    if (q->P > 1) {   // rows into the split order (row_k1)
        hipLaunchKernelGGL(k_row_permute, dim3(4096), dim3(256), 0, st, (const cf*)q->chirp, scratch, q->N1, q->N2, q->P, nplanes);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(q->chirp, scratch, sizeof(cf) * (size_t)L * nplanes, hipMemcpyDeviceToDevice, st));
    }
    return PBH_OK;
}

static int setup_bluestein(pbh_plan* p) {
    const int64_t N = p->N, L = p->bsL;
    #ifdef PBH_F64
    PBHCHECK(pbh_plan_create(&p->sub, p->device, L, 1, p->S, PBH_C128, 0, L));
#else
    PBHCHECK(pbh_plan_create(&p->sub, p->device, L, 1, p->S, PBH_C64, 0, L));
#endif
    pbh_plan* q = p->sub;
    q->perm_w = 0;  // its "chirp" is produced below by a forward FFT, i.e. in natural plan order
    p->owned_bytes += q->owned_bytes;
    PBHCHECK(dev_alloc(p, (void**)&p->bs_b, sizeof(cf) * (size_t)N));
    PBHCHECK(dev_alloc(p, (void**)&p->bs_a, sizeof(cf) * (size_t)L * p->S));
    PBHCHECK(dev_alloc(p, (void**)&p->bs_conv, sizeof(cf) * (size_t)L * p->S));
    hipStream_t st = nullptr;
    hipLaunchKernelGGL(k_bs_table, dim3(1024), dim3(256), 0, st, p->bs_b, N);
    HIPCHECK(hipGetLastError());
    hipLaunchKernelGGL(k_bs_kernel, dim3(2048), dim3(256), 0, st, q->chirp, N, L, (real)(1.0 / (double)L));
    HIPCHECK(hipGetLastError());
    if (q->N1 == 1) {
        // single tile: forward FFT of one series, out of place through the (still unused) bs_a buffer
        HIPCHECK(hipMemcpyAsync(p->bs_a, q->chirp, sizeof(cf) * (size_t)L, hipMemcpyDeviceToDevice, st));
        SmallParams sp{p->bs_a, q->chirp, nullptr, q->tw16k, 1, 1, 0, L, -1, (real)1};
        PBHCHECK(launch_small((int)L, sp, st));
    } else {
        PBHCHECK(spectrum_in_plan_order(q, 1, p->bs_a, st));
    }
    HIPCHECK(hipStreamSynchronize(st));
    q->has_chirp = true;
    return PBH_OK;
}

// ================================================ C ABI ==========================================================
extern "C" {

int pbh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* pbh_last_error(void) { return g_err.c_str(); }

const char* pbh_version(void) { return "pbhip 0.1.0 (gfx950)"; }

static int create_plan(pbh_plan** out, int device, int64_t nsample, int nchan, int npol, int dtype,
                       int64_t crop_start, int64_t crop_stop, int plain_fft /* 0 dedispersion plan, 1 Bluestein ring, 2 native transform only */);

// Non-power-of-two dedispersion plans: the (bsL, nchan, npol) convolution plan and its padded input.
static int fft_c2c_ring(int device, hipStream_t st, const cf* din, cf* dout, int64_t n, int64_t batch, int inverse);

static int setup_circular(pbh_plan* p) {
    const int64_t N = p->N, L = p->bsL;
#ifdef PBH_F64
    const int dt = PBH_C128;
#else
    const int dt = PBH_C64;
#endif
    PBHCHECK(create_plan(&p->cfilt, p->device, L, p->nchan, p->npol, dt, (N - 1) + p->start, (N - 1) + p->stop, 0));
    p->cfilt->perm_w = 0;
    p->owned_bytes += p->cfilt->owned_bytes;
    PBHCHECK(dev_alloc(p, (void**)&p->cf_in, sizeof(cf) * (size_t)L * p->S));
    return PBH_OK;
}

// (Re)build the convolution plan's filter spectrum from the plan-resident natural-order chirp H/N:
// h/N = ifft_N(H/N) (Bluestein ring), taps = (N/L) * (h/N) extended N-periodically over 2N-1 samples,
// G = FFT_L(taps) in plan order (the pipeline multiplies by G and its inverse transform is unnormalised).
static int rebuild_circular_filter(pbh_plan* p) {
    pbh_plan* q = p->cfilt;
    if (!q) return PBH_OK;
    const int64_t N = p->N, L = p->bsL;
    const int nchan = p->nchan;
    hipStream_t st = p->stream;
    cf *nat = nullptr, *hh = nullptr;
    PBHCHECK(dev_alloc(nullptr, (void**)&nat, sizeof(cf) * (size_t)N * nchan));
    int rc = dev_alloc(nullptr, (void**)&hh, sizeof(cf) * (size_t)N * nchan);
    auto grid = [](int64_t m) { int64_t g = (m + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
    if (rc == PBH_OK) {
        hipLaunchKernelGGL(k_cf_nat, dim3(grid(N * nchan)), dim3(256), 0, st, (const cf*)p->chirp, nat, N, nchan);
        if (hipGetLastError() != hipSuccess) rc = fail(PBH_ERR_HIP, "k_cf_nat launch failed");
    }
    if (rc == PBH_OK) rc = fft_c2c_ring(p->device, st, nat, hh, N, nchan, 1);
    if (rc == PBH_OK) {
        hipLaunchKernelGGL(k_cf_extend, dim3(grid(L * nchan)), dim3(256), 0, st, (const cf*)hh, q->chirp, N, L, nchan,
                           (real)((double)N / (double)L));
        if (hipGetLastError() != hipSuccess) rc = fail(PBH_ERR_HIP, "k_cf_extend launch failed");
    }
    if (rc == PBH_OK) {
        if (q->N1 == 1) {
            // single tile: forward FFT of each channel's taps, out of place through `nat` (N*nchan >= ... may be
            // smaller than L): use the padded-input buffer as scratch
            for (int c = 0; c < nchan && rc == PBH_OK; ++c) {
                if (hipMemcpyAsync(p->cf_in, q->chirp + (int64_t)c * L, sizeof(cf) * (size_t)L, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    rc = fail(PBH_ERR_HIP, "hipMemcpyAsync failed");
                SmallParams sp{p->cf_in, q->chirp + (int64_t)c * L, nullptr, q->tw16k, 1, 1, 0, L, -1, (real)1};
                if (rc == PBH_OK) rc = launch_small((int)L, sp, st);
            }
        } else {
            rc = spectrum_in_plan_order(q, nchan, p->cf_in, st);
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == PBH_OK) rc = fail(PBH_ERR_HIP, "circular filter: stream synchronisation failed");
    (void)hipFree(nat);
    if (hh) (void)hipFree(hh);
    if (rc == PBH_OK) {
        q->has_chirp = true;
        q->has_phase = false;
    }
    return rc;
}

int pbh_plan_create(pbh_plan** out, int device, int64_t nsample, int nchan, int npol, int dtype,
                    int64_t crop_start, int64_t crop_stop) {
    return create_plan(out, device, nsample, nchan, npol, dtype, crop_start, crop_stop, 0);
}

static int create_plan(pbh_plan** out, int device, int64_t nsample, int nchan, int npol, int dtype,
                       int64_t crop_start, int64_t crop_stop, int plain_fft) {
    if (!out) return fail(PBH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (nsample <= 0 || nchan <= 0 || npol <= 0) return fail(PBH_ERR_INVALID, "non-positive dimension");
#ifdef PBH_F64
    if (dtype != PBH_C128) return fail(PBH_ERR_INVALID, "dtype mismatch (float64 build)");
#else
    if (dtype != PBH_C64) return fail(PBH_ERR_INVALID, "dtype mismatch (float32 build)");
#endif
    const bool pow2 = is_pow2(nsample) && nsample >= PBH_R && plain_fft != 1;
    const int odd_m = plain_fft == 1 ? 0 : native_odd_factor(nsample);   // nsample = m * 2^k, m in {3, 5, 7}: native too
    if (nsample < 2 || nsample > (1LL << 28) || (!pow2 && nsample > (1LL << 27)))
        return fail(PBH_ERR_UNSUPPORTED, "nsample must be in [2, 2^28] (powers of two) or [2, 2^27] (other lengths); got " +
                                             std::to_string(nsample));
    if (crop_start < 0 || crop_start > nsample) return fail(PBH_ERR_INVALID, "crop_start out of range");
    if (crop_stop > nsample) return fail(PBH_ERR_INVALID, "crop_stop out of range");
    int ndev = pbh_device_count();
    if (device < 0 || device >= ndev)
        return fail(PBH_ERR_INVALID, "device " + std::to_string(device) + " not present (" +
                                         std::to_string(ndev) + " HIP devices)");
    HIPCHECK(hipSetDevice(device));

    pbh_plan* p = new pbh_plan();
    p->device = device;
    p->plain_fft = plain_fft != 0;
    p->N = nsample;
    p->nchan = nchan;
    p->npol = npol;
    p->S = nchan * npol;
    p->start = crop_start;
    p->stop = crop_stop < crop_start ? crop_start : crop_stop;  // empty result if the crop is negative
    const int n = ilog2(nsample);
    int mN1 = 0, mN2 = 0, mP = 0;
    if (odd_m) {
        // the odd factor is the radix-P stage of the split column transform: N1 = m * Q rows, Q a power of two
        p->N2 = kTilePoints;
        p->N1 = (int)(nsample / kTilePoints);
        p->P = odd_m;
    } else if (!pow2 && plain_fft != 1 && mixed_geometry(nsample, &mN1, &mN2, &mP) && !(plain_fft == 2 && mN2 < 1024)) {
        // (stand-alone transforms have no row kernel for power-of-two rows under 1024 points: those lengths take mixed-radix rows)
        // 7-smooth: both column roles are mixed-radix transforms in LDS (k_colmix), the rows stay with the 2^k engine
        p->N1 = mN1;
        p->N2 = mN2;
        p->P = mP;
        p->mixed = true;
    } else if (!pow2 && plain_fft != 1 && rowmix_geometry(nsample, &mN1, &mN2, &mP) && (plain_fft == 2 || rowmix_pays(nsample))) {
        p->N1 = mN1;
        p->N2 = mN2;
        p->P = mP;
        p->mixed = true;
        p->rowmix = true;
    } else if (!pow2) {
        p->N1 = 1;  // natural-order chirp H/N; the transforms run in a native-length convolution plan
        p->N2 = (int)nsample;
        p->bsL = plain_fft ? next_pow2_at_least(2 * nsample - 1) : convolution_length(2 * nsample - 1);
        if (plain_fft == 2) { delete p; return fail(PBH_ERR_UNSUPPORTED, "not a native length"); }
    } else if (n <= kTileLog2) {
        p->N1 = 1;
        p->N2 = (int)nsample;
    } else {
        const int l2 = (n - PBH_LOG2R < kTileLog2) ? n - PBH_LOG2R : kTileLog2;
        p->N2 = 1 << l2;
        p->N1 = (int)(nsample >> l2);
    }
    if (p->N1 > 1 && !odd_m && !p->mixed) {
        // column tiles whose rows are narrower than a 128-byte line are avoided by splitting N1 = P * Q with
        // Q rows per tile such that a tile row is one line; PBH_QMAX overrides Q (tests exercise the split at
        // small sizes)
        int qmax = kTilePoints * (int)sizeof(cf) / 128;
        if (const char* e = getenv("PBH_QMAX")) {
            const int v = atoi(e);
            if (v >= PBH_R && (v & (v - 1)) == 0 && v <= kTilePoints) qmax = v;
        }
        // with stand-alone radix passes the split is worth it from 32-byte pieces down (N1 >= 4 qmax): at N1 = 2 qmax
        // the two extra passes cost more than the 64-byte pieces do (complex64 2^25 x 16 series: 11.3 ms split vs
        // 9.5 ms unsplit; complex128 2^24: 12.2 vs 9.8).  Folded into the layout passes the radix-2 stage is free:
        // 2^25 x 16: 9.53 -> 8.57 ms (callers without layout passes run such plans unsplit, build_steps)
        const int pmin = (getenv("PBH_QMAX") || radix_layout_ok(p->S, 2, nsample, p->N2)) ? 2 : 4;
        if (p->N1 >= pmin * qmax && p->N1 / qmax <= 16 && nsample < (1LL << 31) && p->N2 % (kTilePoints / qmax) == 0)
            p->P = p->N1 / qmax;
    }
    {
        const char* e = diag_env("PBH_ROW2");
        const bool want = e ? atoi(e) != 0 : false;
#ifdef PBH_F64
        p->perm_w = 0;
        (void)want;
#else
        p->perm_w = (want && p->N1 > 1 && p->N2 == 16384) ? 8 : 0;
#endif
    }
    int rc = PBH_OK;
    auto bail = [&](int code) {
        pbh_plan_destroy(p);
        return code;
    };
    // stage twiddle table W_16384^p (float32 from float64 evaluation)
    {
        std::vector<cf> h(kTwTable);
        for (int i = 0; i < kTwTable; ++i) {
            double a = -2.0 * M_PI * (double)i / (double)kTwTable;
            h[i] = make_cf((real)cos(a), (real)sin(a));
        }
        if ((rc = dev_alloc(p, (void**)&p->tw16k, sizeof(cf) * kTwTable + kCounterBytes)) != PBH_OK) return bail(rc);  // + the persistent kernels' tile counters
        if (hipMemcpy(p->tw16k, h.data(), sizeof(cf) * kTwTable, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(PBH_ERR_HIP, "hipMemcpy(tw16k) failed"));
    }
    if (!plain_fft && (rc = dev_alloc(p, (void**)&p->chirp, sizeof(cf) * (size_t)nchan * nsample)) != PBH_OK)
        return bail(rc);
    if ((rc = dev_alloc(p, (void**)&p->chan_freq, sizeof(double) * nchan)) != PBH_OK) return bail(rc);
    if (p->N1 > 1) {
        // W_N^p = hi[p >> shift] * lo[p & mask], float64
        p->tw_shift = (n + 1) / 2;
        const int64_t nlo = 1LL << p->tw_shift, nhi = ((nsample - 1) >> p->tw_shift) + 1;
        std::vector<double2> lo(nlo), hi(nhi);
        for (int64_t i = 0; i < nlo; ++i) {
            double a = -2.0 * M_PI * (double)i / (double)nsample;
            lo[i] = make_double2(cos(a), sin(a));
        }
        for (int64_t i = 0; i < nhi; ++i) {
            double a = -2.0 * M_PI * (double)(i << p->tw_shift) / (double)nsample;
            hi[i] = make_double2(cos(a), sin(a));
        }
        if ((rc = dev_alloc(p, (void**)&p->tw_lo, sizeof(double2) * nlo)) != PBH_OK) return bail(rc);
        if ((rc = dev_alloc(p, (void**)&p->tw_hi, sizeof(double2) * nhi)) != PBH_OK) return bail(rc);
        if (hipMemcpy(p->tw_lo, lo.data(), sizeof(double2) * nlo, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->tw_hi, hi.data(), sizeof(double2) * nhi, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(PBH_ERR_HIP, "hipMemcpy(twiddle tables) failed"));
        if ((rc = dev_alloc(p, (void**)&p->work, sizeof(cf) * (size_t)p->S * nsample)) != PBH_OK) return bail(rc);
    } else if (single_planar_ok(p)) {
        if ((rc = dev_alloc(p, (void**)&p->work, sizeof(cf) * (size_t)p->S * nsample)) != PBH_OK) return bail(rc);
    }
    if (p->bsL) {
        if ((rc = plain_fft ? setup_bluestein(p) : setup_circular(p)) != PBH_OK) return bail(rc);
    }
    if (p->mixed) {
        if (p->P > 1 && (rc = build_mix_table(p, p->P, &p->mixP)) != PBH_OK) return bail(rc);
        if ((rc = build_mix_table(p, p->N1 / p->P, &p->mixQ)) != PBH_OK) return bail(rc);
        if (p->rowmix && (rc = build_mix_table(p, p->N2, &p->mixR, true)) != PBH_OK) return bail(rc);
    }
    *out = p;
    return PBH_OK;
}

int pbh_plan_destroy(pbh_plan* p) {
    if (!p) return PBH_OK;
    (void)hipSetDevice(p->device);   // (a failure shows in the frees below, which are best effort)
    if (p->sub) pbh_plan_destroy(p->sub);
    if (p->cfilt) pbh_plan_destroy(p->cfilt);
    if (p->cf_in) (void)hipFree(p->cf_in);
    void* ptrs[] = {p->chirp_phase, p->work, p->work2, p->det_part, p->chirp, p->tw16k, p->tw_hi, p->tw_lo, p->chan_freq, p->mix_ft, p->stage_in, p->stage_out, p->det_mid,
                    p->bs_b, p->bs_a, p->bs_conv, p->mixP.wl, p->mixP.perm, p->mixQ.wl, p->mixQ.perm, p->mixR.wl, p->mixR.perm};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
    return PBH_OK;
}

int pbh_plan_set_stream(pbh_plan* p, void* hip_stream) {
    if (!p) return fail(PBH_ERR_INVALID, "plan is NULL");
    p->stream = (hipStream_t)hip_stream;
    return PBH_OK;
}

int pbh_plan_set_variant(pbh_plan* p, int variant) {
    if (!p) return fail(PBH_ERR_INVALID, "plan is NULL");
    if (variant < PBH_VARIANT_AUTO || variant > PBH_VARIANT_BLOCK3) return fail(PBH_ERR_INVALID, "bad variant");
    if (variant == PBH_VARIANT_BLOCK3 && p->N1 > 1) {
        const int F = kTilePoints / p->N1;
        if (F < 16 || p->S % 4 != 0 || p->N2 % (F / 4 > 16 ? F / 4 : 16) != 0)
            return fail(PBH_ERR_UNSUPPORTED, "block3 needs nchan*npol % 4 == 0 and nsample <= 2^24");
    }
    p->variant = variant;
    return PBH_OK;
}

int pbh_plan_buffer_class(pbh_plan* p, const void* dev_ptr, int64_t bytes, int* cls) {
    if (!p || !cls) return fail(PBH_ERR_INVALID, "NULL argument");
    *cls = -1;
#ifndef PBH_F64
    HIPCHECK(hipSetDevice(p->device));
    if (fd4_ok(p) && !p->work2) (void)ensure_work2(p, bytes >= (int64_t)(sizeof(cf) * (size_t)p->S * (size_t)p->N) ? dev_ptr : nullptr, nullptr, 0);
    if (dev_ptr && bytes > 0) *cls = buffer_class(p, dev_ptr, (size_t)bytes);
#else
    (void)dev_ptr;
    (void)bytes;
#endif
    return PBH_OK;
}

int pbh_plan_info(const pbh_plan* p, pbh_plan_info_t* info) {
    if (!p || !info) return fail(PBH_ERR_INVALID, "NULL argument");
    info->nsample = p->N;
    info->crop_start = p->start;
    info->crop_stop = p->stop;
    info->nchan = p->nchan;
    info->npol = p->npol;
    info->device = p->device;
    info->n1 = p->N1;
    info->n2 = p->N2;
    info->variant = resolved_variant(p);
    // (long blocks: +2 for the stand-alone radix-P stage unless it is folded into the layout passes)
    info->nkernel = p->N1 == 1 ? ((p->work && single_planar_ok(p)) ? 3 : 1) : (info->variant == PBH_VARIANT_PLANAR5 ? 5 : 3) +
                                         ((p->P > 1 && (p->mixed || !radix_layout_ok(p->S, p->P, p->N, p->N2))) ? 2 : 0);
    if (p->bsL && p->cfilt) {
        pbh_plan_info_t sub;
        pbh_plan_info(p->cfilt, &sub);
        // + the padding copy, unless the convolution plan's own de-interleave pass does the zero padding
        info->nkernel = sub.nkernel + ((p->cfilt->N1 > 1 && sub.variant == PBH_VARIANT_PLANAR5) ? 0 : 1);
    } else if (p->bsL) {
        pbh_plan_info_t sub;
        pbh_plan_info(p->sub, &sub);
        info->nkernel = 3 + 2 * sub.nkernel;
    }
    info->workspace_bytes = p->owned_bytes;
    // 4 passes * 16 B + chirp 8/npol B for multi-pass transforms; one read + one write
    // + chirp for a transform that fits one tile.
    info->alg_bytes_per_sample = (p->N1 == 1 ? 16.0 : 64.0) + 8.0 / p->npol;
    return PBH_OK;
}

// ---- chirp --------------------------------------------------------------------------------------------------
static real inv_n(const pbh_plan* p) { return (real)(1.0 / (double)p->N); }

// the complex64 form of a generated chirp that so far exists as phase rows only (pbh_chirp_generate)
static int materialize_chirp(pbh_plan* p) {
    if (!p->chirp_lazy) return PBH_OK;
    ChirpParams cp{p->chirp, p->chan_freq, p->gen_coeff, p->gen_inv_ndt, p->gen_inv_ref, p->N, p->N1, p->N2, p->nchan, inv_n(p), p->perm_w};
    cp.P = p->P;
    cp.row_perm = p->rowmix ? p->mixR.perm : nullptr;
    hipLaunchKernelGGL(k_chirp, dim3(2048), dim3(256), 0, p->stream, cp);
    HIPCHECK(hipGetLastError());
    p->chirp_lazy = false;
    return PBH_OK;
}

int pbh_chirp_generate(pbh_plan* p, double coeff_hz, double dt_s, const double* chan_freq_hz, double ref_freq_hz) {
    if (!p || !chan_freq_hz) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!(dt_s > 0)) return fail(PBH_ERR_INVALID, "dt must be positive");
    HIPCHECK(hipSetDevice(p->device));
    HIPCHECK(hipMemcpyAsync(p->chan_freq, chan_freq_hz, sizeof(double) * p->nchan, hipMemcpyHostToDevice, p->stream));
    ChirpParams cp{p->chirp, p->chan_freq, coeff_hz, 1.0 / ((double)p->N * dt_s), 1.0 / ref_freq_hz,
                   p->N, p->N1, p->N2, p->nchan, inv_n(p), p->perm_w};
    cp.P = p->P;
    cp.row_perm = p->rowmix ? p->mixR.perm : nullptr;
    p->has_phase = false;
#ifndef PBH_F64
    // the fused row pass of multi-pass float32 plans reads the chirp as a phase (k_rowp)
    if (p->N1 > 1 && rowp_ok(p->N1, p->N2) && p->perm_w == 0 && !p->bsL && !p->rowmix) {
        if (!p->chirp_phase) PBHCHECK(dev_alloc(p, (void**)&p->chirp_phase, sizeof(float) * (size_t)p->nchan * p->N));
        cp.phase = p->chirp_phase;
        cp.phase16 = rowp16_on(p->N2) ? 1 : 0;
        p->phase16 = cp.phase16 != 0;
        p->has_phase = true;
    }
    // The fused row pass of such a plan reads the PHASE rows only.  The complex64 chirp (a float64 sincospi and 8 bytes written
    // per bin: 0.6 of the 0.8 ms this call took at config 2) is made when somebody asks for it -- pbh_chirp_download, i.e.
    // chirp_from_signal -- and not for every new DM of a search.  PBH_CHIRP_LAZY=0: always both.
    static const bool lazy_on = [] { const char* e = diag_env("PBH_CHIRP_LAZY"); return e ? atoi(e) != 0 : true; }();
    p->chirp_lazy = lazy_on && p->has_phase && row_phase_enabled();
    if (p->chirp_lazy) cp.out = nullptr;
#else
    p->chirp_lazy = false;
#endif
    p->gen_coeff = cp.coeff;
    p->gen_inv_ndt = cp.inv_ndt;
    p->gen_inv_ref = cp.inv_ref;
    hipLaunchKernelGGL(k_chirp, dim3(2048), dim3(256), 0, p->stream, cp);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(p->stream));  // chan_freq_hz is a borrowed host buffer
    p->has_chirp = true;
    PBHCHECK(rebuild_circular_filter(p));
    return PBH_OK;
}

int pbh_chirp_upload_as(pbh_plan* p, const void* chirp, int chirp_dtype, int loc);
int pbh_chirp_upload(pbh_plan* p, const void* chirp_c64, int loc) { return pbh_chirp_upload_as(p, chirp_c64, PBH_C64, loc); }

// chirp_dtype: PBH_C64 (what the reference's own chirps are, dedispersion.py:23) or, for complex128 plans, PBH_C128 -- the
// reference multiplies by whatever array it is given (dedispersion.py:124-125), so a complex128 chirp keeps its precision
int pbh_chirp_upload_as(pbh_plan* p, const void* chirp_c64, int chirp_dtype, int loc) {
    if (!p || !chirp_c64) return fail(PBH_ERR_INVALID, "NULL argument");
    if (chirp_dtype != PBH_C64 && chirp_dtype != PBH_C128) return fail(PBH_ERR_INVALID, "chirp dtype must be PBH_C64 or PBH_C128");
    if (chirp_dtype == PBH_C128 && sizeof(cf) != sizeof(double2))
        return fail(PBH_ERR_UNSUPPORTED, "a complex128 chirp needs a complex128 plan (the host promotes the data as numpy would)");
    HIPCHECK(hipSetDevice(p->device));
    const size_t esz = chirp_dtype == PBH_C128 ? sizeof(double2) : sizeof(float2);
    const size_t bytes = esz * (size_t)p->nchan * p->N;
    const void* src = chirp_c64;
    if (loc == PBH_HOST) {
        PBHCHECK(ensure_stage(p, &p->stage_in, &p->stage_in_bytes, bytes));
        HIPCHECK(xfer_h2d(p->stage_in, chirp_c64, bytes, p->stream));
        src = p->stage_in;
    }
    if (chirp_dtype == PBH_C128)
        hipLaunchKernelGGL(k_chirp_reorder<double2>, dim3(2048), dim3(256), 0, p->stream, (const double2*)src, (double2*)nullptr,
                           (const cf*)nullptr, p->chirp, p->N, p->N1, p->N2, p->nchan, inv_n(p), 1, p->perm_w, p->P,
                           p->rowmix ? p->mixR.perm : (const unsigned short*)nullptr);
    else
        hipLaunchKernelGGL(k_chirp_reorder<float2>, dim3(2048), dim3(256), 0, p->stream, (const float2*)src, (float2*)nullptr,
                           (const cf*)nullptr, p->chirp, p->N, p->N1, p->N2, p->nchan, inv_n(p), 1, p->perm_w, p->P,
                           p->rowmix ? p->mixR.perm : (const unsigned short*)nullptr);
    HIPCHECK(hipGetLastError());
    if (loc == PBH_HOST) HIPCHECK(hipStreamSynchronize(p->stream));
    p->has_chirp = true;
    p->chirp_lazy = false;
    p->has_phase = false;  // a user-supplied chirp is applied as the complex64 values it is
    PBHCHECK(rebuild_circular_filter(p));
    return PBH_OK;
}

int pbh_chirp_download(pbh_plan* p, void* chirp_c64, int loc) {
    if (!p || !chirp_c64) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "plan has no chirp yet");
    HIPCHECK(hipSetDevice(p->device));
    PBHCHECK(materialize_chirp(p));
    const size_t bytes = sizeof(float2) * (size_t)p->nchan * p->N;
    float2* dst = (float2*)chirp_c64;
    if (loc == PBH_HOST) {
        PBHCHECK(ensure_stage(p, &p->stage_out, &p->stage_out_bytes, bytes));
        dst = (float2*)p->stage_out;
    }
    hipLaunchKernelGGL(k_chirp_reorder<float2>, dim3(2048), dim3(256), 0, p->stream, (const float2*)nullptr, dst,
                       (const cf*)p->chirp, (cf*)nullptr, p->N, p->N1, p->N2, p->nchan, (real)p->N, 0, p->perm_w, p->P,
                       p->rowmix ? p->mixR.perm : (const unsigned short*)nullptr);
    HIPCHECK(hipGetLastError());
    if (loc == PBH_HOST) {
        HIPCHECK(xfer_d2h(chirp_c64, dst, bytes, p->stream));
        HIPCHECK(hipStreamSynchronize(p->stream));
    }
    return PBH_OK;
}

// H = phase ramp (mode 0, arg = per-channel shift in samples) or band mask (mode 1, arg = ft*N)
int pbh_chirp_special(pbh_plan* p, const double* arg /*[nchan]*/, int mode) {
    if (!p || !arg) return fail(PBH_ERR_INVALID, "NULL argument");
    if (mode < 0 || mode > 2) return fail(PBH_ERR_INVALID, "bad mode");
    HIPCHECK(hipSetDevice(p->device));
    HIPCHECK(hipMemcpyAsync(p->chan_freq, arg, sizeof(double) * p->nchan, hipMemcpyHostToDevice, p->stream));
    ChirpParams cp{p->chirp, p->chan_freq, 0.0, 0.0, 0.0, p->N, p->N1, p->N2, p->nchan, inv_n(p), p->perm_w};
    cp.P = p->P;
    cp.row_perm = p->rowmix ? p->mixR.perm : nullptr;
    bool phase = false;
#ifndef PBH_F64
    // the time-shift ramp has unit magnitude: the row pass can read it as a phase, like a generated chirp
    if (mode == 0 && p->N1 > 1 && rowp_ok(p->N1, p->N2) && p->perm_w == 0 && !p->bsL && !p->rowmix) {
        if (!p->chirp_phase) PBHCHECK(dev_alloc(p, (void**)&p->chirp_phase, sizeof(float) * (size_t)p->nchan * p->N));
        cp.phase = p->chirp_phase;
        cp.phase16 = rowp16_on(p->N2) ? 1 : 0;
        p->phase16 = cp.phase16 != 0;
        phase = true;
    }
#endif
    hipLaunchKernelGGL(k_chirp_special, dim3(2048), dim3(256), 0, p->stream, cp, (const double*)p->chan_freq, mode);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(p->stream));
    p->has_chirp = true;
    p->chirp_lazy = false;
    p->has_phase = phase;
    PBHCHECK(rebuild_circular_filter(p));
    return PBH_OK;
}

static int with_device_doubles(const double* host, int n, hipStream_t st, double** dev) {
    PBHCHECK(dev_alloc(nullptr, (void**)dev, sizeof(double) * (size_t)n));
    hipError_t e = hipMemcpyAsync(*dev, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
        (void)hipFree(*dev);
        return fail(PBH_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    return PBH_OK;
}

// out[n, s] = in[n, s] * exp(2 pi i ft[s] n)  (device-resident (N, S) arrays; in may equal out)
int pbh_mix(int device, void* hip_stream, int /*dtype*/, const void* in_dev, void* out_dev, int64_t nsample, int nseries,
            const double* ft /*[nseries] host*/) {
    if (!in_dev || !out_dev || !ft) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || nseries <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    double* d = nullptr;
    PBHCHECK(with_device_doubles(ft, nseries, st, &d));
    int64_t blocks = (nsample * nseries + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_mix, dim3((unsigned)blocks), dim3(256), 0, st, (const cf*)in_dev, (cf*)out_dev, (const double*)d,
                       nsample, nseries);
    hipError_t e = hipGetLastError();
    const hipError_t es = hipStreamSynchronize(st);
    if (e == hipSuccess) e = es;
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("k_mix: ") + hipGetErrorString(e));
    return PBH_OK;
}

int pbh_zero_edges(int device, void* hip_stream, int /*dtype*/, void* data_dev, int64_t nsample, int nseries,
                   const double* shift /*[nseries] host*/) {
    if (!data_dev || !shift) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || nseries <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    int64_t maxrows = 0;
    for (int i = 0; i < nseries; ++i) {
        const double a = shift[i];
        const int64_t c = a < 0 ? -(int64_t)floor(a) : (int64_t)ceil(a);
        if (c > maxrows) maxrows = c;
    }
    if (maxrows > nsample) maxrows = nsample;
    if (maxrows == 0) return PBH_OK;
    double* d = nullptr;
    PBHCHECK(with_device_doubles(shift, nseries, st, &d));
    int64_t blocks = (maxrows * nseries + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_zero_edges, dim3((unsigned)blocks), dim3(256), 0, st, (cf*)data_dev, (const double*)d, nsample,
                       nseries, maxrows);
    hipError_t e = hipGetLastError();
    const hipError_t es = hipStreamSynchronize(st);
    if (e == hipSuccess) e = es;
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("k_zero_edges: ") + hipGetErrorString(e));
    return PBH_OK;
}

// real_to_complex tail: out[m, s] = (-1)^m in[2m, s], m < nout, device arrays
int pbh_decimate2(int device, void* hip_stream, int /*dtype*/, const void* in_dev, void* out_dev, int64_t nout,
                  int nseries) {
    if (!in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nout <= 0 || nseries <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    HIPCHECK(hipSetDevice(device));
    int64_t blocks = (nout * nseries + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_decimate2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (const cf*)in_dev,
                       (cf*)out_dev, nout, nseries);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

// to_circular / to_linear (core.py:882-928) on device (n, nchan, 2) data; in may equal out
int pbh_pol_basis(int device, void* hip_stream, int /*dtype*/, const void* in_dev, void* out_dev, int64_t npairs,
                  int to_circular) {
    if (!in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (npairs <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    HIPCHECK(hipSetDevice(device));
    int64_t blocks = (npairs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_pol_basis, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (const cf*)in_dev,
                       (cf*)out_dev, npairs, to_circular);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

#ifndef PBH_F64
// incoherent dedispersion gather on device data of any dtype: `unit` 4-byte words per (sample, channel)
// Stream-ordered scratch memory (hipMallocAsync / hipFreeAsync) comes from the device's default pool; by default the pool
// hands freed memory back to the system at the next synchronisation, which turns a 2-GiB scratch buffer per call into
// 20 ms of page-table work.  The pool keeps what it has been given until pbh_trim.
static void keep_pool_memory(int device) {
    static thread_local bool done[16] = {};
    if (device < 0 || device >= 16 || done[device]) return;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
        uint64_t keep = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
    done[device] = true;
}

// Sample-major arrays of 8-byte elements with a power-of-two number of series: the gather as two full-line passes
// through a series-major scratch copy (the pipeline's de-interleave kernel, then its re-interleave kernel reading
// every series at its own time offset) -- 2x the useful bytes, where the direct gather touches a whole 128-byte line
// for every 16-byte cell (~nchan x).  Returns 1 when the geometry does not fit (the caller gathers directly).
static int incoherent_two_pass(hipStream_t st, const void* in_dev, void* out_dev, int64_t nout, int nchan, int unit_words,
                               const int64_t* delay) {
    static const bool on = [] { const char* e = diag_env("PBH_INCOHERENT_2PASS"); return e ? atoi(e) != 0 : true; }();
    if (!on || unit_words % 2 != 0) return 1;
    const int per = unit_words / 2;                  // 8-byte elements per (sample, channel) cell
    const int64_t S64 = (int64_t)nchan * per;
    if (S64 < 2 || S64 > 128 || (S64 & (S64 - 1)) != 0 || nout < (1 << 16)) return 1;
    if (((uintptr_t)in_dev | (uintptr_t)out_dev) % 16 != 0) return 1;
    const int S = (int)S64;
    int64_t dmax = 0;
    std::vector<int64_t> dser((size_t)S);
    for (int s2 = 0; s2 < S; ++s2) {
        dser[s2] = delay[s2 / per];
        if (dser[s2] < 0) return 1;
        if (dser[s2] > dmax) dmax = dser[s2];
    }
    const int64_t nin = nout + dmax;                 // input rows the gather can touch
    const int TN = tr_rows(S);
    const int64_t plane = (nin + TN - 1) / TN * TN + 16;
    cf* tmp = nullptr;
    int64_t* dd = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) keep_pool_memory(dev);
    if (hipMallocAsync((void**)&tmp, sizeof(cf) * (size_t)plane * S, st) != hipSuccess) {
        (void)hipGetLastError();
        return 1;                                    // no room for the scratch copy: gather directly
    }
    hipError_t e = hipMallocAsync((void**)&dd, sizeof(int64_t) * (size_t)S, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, dser.data(), sizeof(int64_t) * (size_t)S, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // dser is a local: staged before it goes out of scope
    int rc = e == hipSuccess ? PBH_OK : fail(PBH_ERR_HIP, std::string("pbh_incoherent: ") + hipGetErrorString(e));
    if (rc == PBH_OK) {
        // whole tiles of the row transposes, the last one zero-padded beyond nin (the scratch planes are long enough)
        const unsigned grid = (unsigned)((nin + TN - 1) / TN);
        switch (S) {
#define X(s2) case s2: hipLaunchKernelGGL(k_deinterleave_p2<s2>, dim3(grid), dim3(256), 0, st, (const cf*)in_dev, tmp, nin, plane, nin); break;
            X(2) X(4) X(8) X(16) X(32) X(64) X(128)
#undef X
        }
        if (hipGetLastError() != hipSuccess) rc = fail(PBH_ERR_HIP, "pbh_incoherent: de-interleave launch failed");
    }
    if (rc == PBH_OK) rc = launch_reinterleave(tmp, (cf*)out_dev, 0, nout, S, plane, st, 0, dd);
    (void)hipFreeAsync(tmp, st);
    if (dd) (void)hipFreeAsync(dd, st);
    return rc;
}

int pbh_incoherent(int device, void* hip_stream, const void* in_dev, void* out_dev, int64_t nout, int nchan,
                   int unit_words, const int64_t* delay /*[nchan] host, >= 0*/) {
    if (!in_dev || !out_dev || !delay) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nout < 0 || nchan <= 0 || unit_words <= 0) return fail(PBH_ERR_INVALID, "bad size");
    if (nout == 0) return PBH_OK;
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    {
        const int rc2 = incoherent_two_pass(st, in_dev, out_dev, nout, nchan, unit_words, delay);
        if (rc2 <= 0) return rc2;
    }
    // device copy of the delays: allocated and freed in stream order (no device-wide synchronisation, nothing cached
    // per thread); the host array is staged by the copy before it returns
    int64_t* d = nullptr;
    HIPCHECK(hipMallocAsync((void**)&d, sizeof(int64_t) * (size_t)nchan, st));
    hipError_t e = xfer_h2d(d, delay, sizeof(int64_t) * (size_t)nchan, st);
    if (e == hipSuccess) {
        constexpr int U = 8;
        const bool vec = unit_words % 4 == 0 && ((uintptr_t)in_dev | (uintptr_t)out_dev) % 16 == 0;
        const int unit = vec ? unit_words / 4 : unit_words;
        const int64_t total = nout * nchan * unit;
        const unsigned blocks = (unsigned)((total + 256 * U - 1) / (256 * U));
        if (vec)
            hipLaunchKernelGGL((k_incoherent<uint4, U>), dim3(blocks), dim3(256), 0, st, (const uint4*)in_dev, (uint4*)out_dev,
                               (const int64_t*)d, nout, nchan, unit);
        else
            hipLaunchKernelGGL((k_incoherent<uint32_t, U>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)in_dev,
                               (uint32_t*)out_dev, (const int64_t*)d, nout, nchan, unit);
        e = hipGetLastError();
    }
    const hipError_t ef = hipFreeAsync(d, st);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_incoherent: ") + hipGetErrorString(e));
    return PBH_OK;
}

// The same gather on series-major arrays (time fastest; pitches in 4-byte words between consecutive series): every
// series is one contiguous run, so it is a shifted copy per series.  unit_words = 4-byte words per element of a series.
int pbh_incoherent_series(int device, void* hip_stream, const void* in_dev, int64_t in_pitch_words, void* out_dev,
                          int64_t out_pitch_words, int64_t nout, int nchan, int series_per_chan, int unit_words,
                          const int64_t* delay /*[nchan] host, >= 0*/) {
    if (!in_dev || !out_dev || !delay) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nout < 0 || nchan <= 0 || series_per_chan <= 0 || (unit_words != 1 && unit_words != 2 && unit_words != 4) ||
        (int64_t)nchan * series_per_chan > INT32_MAX)
        return fail(PBH_ERR_INVALID, "bad size");
    if (in_pitch_words % unit_words || out_pitch_words % unit_words) return fail(PBH_ERR_INVALID, "pitch is not a whole number of elements");
    if (nout == 0) return PBH_OK;
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    int64_t* d = nullptr;
    HIPCHECK(hipMallocAsync((void**)&d, sizeof(int64_t) * (size_t)nchan, st));
    hipError_t e = xfer_h2d(d, delay, sizeof(int64_t) * (size_t)nchan, st);
    if (e == hipSuccess) {
        const int64_t ip = in_pitch_words / unit_words, op = out_pitch_words / unit_words;
        int64_t bx = (nout + 255) / 256;
        if (bx > 4096) bx = 4096;
        const int64_t nser = (int64_t)nchan * series_per_chan;
        for (int64_t s0 = 0; s0 < nser; s0 += 65535) {   // grid.y holds at most 65535 series
            const dim3 grid((unsigned)bx, (unsigned)(nser - s0 < 65535 ? nser - s0 : 65535));
            if (unit_words == 4)
                hipLaunchKernelGGL((k_shift_rows<uint4>), grid, dim3(256), 0, st, (const uint4*)in_dev, ip, (uint4*)out_dev, op, (const int64_t*)d, series_per_chan, nout, (int)s0);
            else if (unit_words == 2)
                hipLaunchKernelGGL((k_shift_rows<uint2>), grid, dim3(256), 0, st, (const uint2*)in_dev, ip, (uint2*)out_dev, op, (const int64_t*)d, series_per_chan, nout, (int)s0);
            else
                hipLaunchKernelGGL((k_shift_rows<uint32_t>), grid, dim3(256), 0, st, (const uint32_t*)in_dev, ip, (uint32_t*)out_dev, op, (const int64_t*)d, series_per_chan, nout, (int)s0);
        }
        e = hipGetLastError();
    }
    const hipError_t ef = hipFreeAsync(d, st);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_incoherent_series: ") + hipGetErrorString(e));
    return PBH_OK;
}
#endif

#ifndef PBH_F64
int pbh_chirp_function(int device, void* hip_stream, double coeff_hz, int64_t nsample, double dt_s,
                       double center_freq_hz, double ref_freq_hz, void* chirp_c64, int loc) {
    if (!chirp_c64) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || nsample > (1LL << 30) || !(dt_s > 0)) return fail(PBH_ERR_INVALID, "bad nsample/dt");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t bytes = sizeof(cf) * (size_t)nsample;
    double* dfreq = nullptr;
    cf* dbuf = (cf*)chirp_c64;
    PBHCHECK(dev_alloc(nullptr, (void**)&dfreq, sizeof(double)));
    int rc = PBH_OK;
    if (loc == PBH_HOST && (rc = dev_alloc(nullptr, (void**)&dbuf, bytes)) != PBH_OK) {
        (void)hipFree(dfreq);
        return rc;
    }
    auto cleanup = [&]() {
        (void)hipFree(dfreq);
        if (loc == PBH_HOST) (void)hipFree(dbuf);
    };
    hipError_t e = hipMemcpyAsync(dfreq, &center_freq_hz, sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        // natural order, any nsample (not only powers of two): N1 = 1, N2 = N
        ChirpParams cp{dbuf, dfreq, coeff_hz, 1.0 / ((double)nsample * dt_s), 1.0 / ref_freq_hz,
                       nsample, 1, (int)nsample, 1, 1.0f, 0};
        hipLaunchKernelGGL(k_chirp, dim3(1024), dim3(256), 0, st, cp);
        e = hipGetLastError();
    }
    if (e == hipSuccess && loc == PBH_HOST) e = xfer_d2h(chirp_c64, dbuf, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_chirp_function: ") + hipGetErrorString(e));
    return PBH_OK;
}

#endif  // !PBH_F64

// ---- hot path --------------------------------------------------------------------------------------------------
static int resolve_io(pbh_plan* p, const void* in, void* out, size_t out_bytes, int in_loc, int out_loc,
                      const cf** din, void** dout) {
    const size_t in_bytes = sizeof(cf) * (size_t)p->S * p->N;
    *din = (const cf*)in;
    *dout = out;
    if (in_loc == PBH_HOST) {
        PBHCHECK(ensure_stage(p, &p->stage_in, &p->stage_in_bytes, in_bytes));
        HIPCHECK(xfer_h2d(p->stage_in, in, in_bytes, p->stream));
        *din = (const cf*)p->stage_in;
    }
    if (out_loc == PBH_HOST) {
        PBHCHECK(ensure_stage(p, &p->stage_out, &p->stage_out_bytes, out_bytes ? out_bytes : 16));
        *dout = p->stage_out;
    }
    return PBH_OK;
}

int pbh_dedisperse(pbh_plan* p, const void* in_c64, void* out_c64, int in_loc, int out_loc) {
    if (!p || !in_c64 || !out_c64) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    HIPCHECK(hipSetDevice(p->device));
    const size_t out_bytes = sizeof(cf) * (size_t)p->S * (size_t)(p->stop - p->start);
    const cf* din;
    void* dout;
    PBHCHECK(resolve_io(p, in_c64, out_c64, out_bytes, in_loc, out_loc, &din, &dout));
    if (p->stop > p->start) {
        auto steps = build_steps(p, din, (cf*)dout);
        PBHCHECK(run_steps(steps, p->stream));
    }
    if (out_loc == PBH_HOST && out_bytes)
        HIPCHECK(xfer_d2h(out_c64, dout, out_bytes, p->stream));
    if (in_loc == PBH_HOST || out_loc == PBH_HOST) HIPCHECK(hipStreamSynchronize(p->stream));
    return PBH_OK;
}

// Device-resident dedispersion with explicit layouts at both ends (include/pbhip.h).
int pbh_dedisperse_layout(pbh_plan* p, const void* in_dev, int in_layout, int64_t in_pitch, void* out_dev,
                          int out_layout, int64_t out_pitch) {
    if (!p || !in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    auto bad = [](int l) { return l != PBH_LAYOUT_SAMPLE_MAJOR && l != PBH_LAYOUT_SERIES_MAJOR; };
    if (bad(in_layout) || bad(out_layout)) return fail(PBH_ERR_INVALID, "bad layout");
    const int64_t nout = p->stop - p->start;
    if (in_layout == PBH_LAYOUT_SERIES_MAJOR && in_pitch < p->N) return fail(PBH_ERR_INVALID, "in_pitch < nsample");
    if (out_layout == PBH_LAYOUT_SERIES_MAJOR && out_pitch < nout) return fail(PBH_ERR_INVALID, "out_pitch < output length");
    HIPCHECK(hipSetDevice(p->device));
    if (nout <= 0) return PBH_OK;
    const bool any_sm = in_layout == PBH_LAYOUT_SERIES_MAJOR || out_layout == PBH_LAYOUT_SERIES_MAJOR;
    if (any_sm && !p->mixed && (p->bsL || p->N1 == 1 || p->N1 / p->P > kTilePoints || p->N2 % (kTilePoints / (p->N1 / p->P)) != 0 ||
                   p->N >= (1LL << 31)))
        return fail(PBH_ERR_UNSUPPORTED, "series-major I/O needs a multi-pass power-of-two plan (nsample > one tile)");
    IoLayout io;
    io.in_layout = in_layout;
    io.out_layout = out_layout;
    io.in_pitch = in_pitch;
    io.out_pitch = out_pitch;
    auto steps = build_steps(p, (const cf*)in_dev, (cf*)out_dev, DetectTail(), io);
    PBHCHECK(run_steps(steps, p->stream));
    return PBH_OK;
}

// PBH_STFT_SIBLINGS=1 (experiment, round 3): fused tiles that take fewer than 8 of many series -- nperseg 256 / 512 with 16
// series -- run with one workgroup doing all the sibling subsets of its segments (so that the shared 128-byte lines meet in
// one L2) instead of falling back to two steps.  It does not pay: 5.10 vs 4.90 ms (256) and 6.67 vs 4.95 ms (512) forward,
// 5.42 vs 4.86 and 6.53 vs 4.68 inverse (profiles/r03_stft_dedisp_fused.txt) -- the cost of 32- and 16-byte pieces is the
// address work per byte in the CU's texture path, not a re-fetch across XCDs.  Default: two steps for those geometries.
static bool stft_sibling_loop() {
    static const bool on = [] { const char* e = diag_env("PBH_STFT_SIBLINGS"); return e ? atoi(e) != 0 : false; }();
    return on;
}

// contrib.stft followed by coherent_dedispersion in one call (pulsarbat/contrib/misc.py:41-55, then
// transforms/dedispersion.py:125): `plan` is the dedispersion plan of the CHANNELISED block, (nseg, nchan_in*nperseg,
// inner); the input is the (nseg*nperseg, nchan_in, inner) block the channeliser would read.  Where the geometry
// allows (float32, nperseg = 2^m in [32, 1024], a multi-pass plan) the channeliser writes its output series-major
// straight into the plan's work buffer and the dedispersion starts at its column pass -- the channelised block is
// never written in the reference layout nor de-interleaved again.  Every other geometry runs the two steps one after
// the other through a scratch copy of the channelised block.
int PBH_FN(stft)(int device, void* hip_stream, int dtype, const void* in, void* out, int64_t nseg, int nperseg, int nchan,
                 int inner, int inverse, int in_loc, int out_loc);
int PBH_FN(stft_dedisperse)(pbh_plan* p, const void* in_dev, int nperseg, int nchan_in, void* out_dev, int out_layout,
                            int64_t out_pitch) {
    if (!p || !in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    if (nperseg <= 0 || nchan_in <= 0 || (int64_t)nchan_in * nperseg != p->nchan)
        return fail(PBH_ERR_INVALID, "pbh_stft_dedisperse: the plan must have nchan = nchan_in * nperseg channels");
    if (out_layout != PBH_LAYOUT_SAMPLE_MAJOR && out_layout != PBH_LAYOUT_SERIES_MAJOR) return fail(PBH_ERR_INVALID, "bad layout");
    const int64_t nout = p->stop - p->start, nseg = p->N;
    if (out_layout == PBH_LAYOUT_SERIES_MAJOR && out_pitch < nout) return fail(PBH_ERR_INVALID, "out_pitch < output length");
    HIPCHECK(hipSetDevice(p->device));
    if (nout <= 0) return PBH_OK;
    const int E = p->npol, M = nperseg;
    const int64_t Sin = (int64_t)nchan_in * E;
    const bool multipass = !(p->bsL || p->mixed || p->N1 == 1 || p->P > 1 || p->N1 > kTilePoints || p->N2 % (kTilePoints / p->N1) != 0 ||
                             p->N >= (1LL << 31));
    if (out_layout == PBH_LAYOUT_SERIES_MAJOR && !multipass)
        return fail(PBH_ERR_UNSUPPORTED, "series-major output needs a multi-pass power-of-two plan (nsample > one tile)");
#ifndef PBH_F64
    static const bool fuse_on = [] { const char* e = diag_env("PBH_STFT_FUSE"); return e ? atoi(e) != 0 : true; }();
    if (fuse_on && multipass && is_pow2(M) && M >= PBH_R && M * 16 <= kTilePoints) {
        const int F = kTilePoints / M;
        int SB = F / 16;
        while (SB > 1 && (Sin % SB != 0 || SB > Sin)) SB >>= 1;
        const int G = F / SB;
        // (a tile that takes fewer than 8 of many series reads 32-byte or smaller pieces of the input's lines: slower
        //  than the two steps it would replace -- nperseg 256: 5.6 vs 5.2 ms, 512: 6.6 vs 5.0 for 2^24 x 8 x 2)
        // tiles that take fewer than 8 of many series (32-byte or smaller pieces of the input's lines): one workgroup does all
        // the sibling subsets of its segments, so that the lines are fetched into ONE L2 once
        int nsub = 1;
        if (SB < 8 && SB != Sin) {
            nsub = (int)(Sin / SB);
            while (nsub > 16 || (Sin / SB) % nsub != 0) --nsub;
        }
        if (nseg % G == 0 && Sin / SB <= 16383 && (SB >= 8 || SB == Sin || (stft_sibling_loop() && SB >= 2 && nsub * SB * (int)sizeof(cf) >= 64))) {
            StftPlanarParams sp{(const cf*)in_dev, p->work, p->tw16k, p->N, (int)Sin, E, SB, G, (real)(1.0 / (double)M)};
            sp.nsub = nsub;
            const int64_t ngrp = nseg / G;
            int rc = PBH_OK;
            for (int64_t y0 = 0; y0 < ngrp && rc == PBH_OK; y0 += 65535) {
                const int64_t cnt = ngrp - y0 < 65535 ? ngrp - y0 : 65535;
                StftPlanarParams q = sp;
                q.in = sp.in + y0 * G * (int64_t)M * Sin;
                q.out = sp.out + y0 * G;
                switch (M) {
#define X(m) case m: rc = launch_tile_kernel(k_stft_planar<m, PBH_R>, q, Sin / SB / nsub, kTilePoints / PBH_R, p->stream, lds_tile_bytes<false>(), (unsigned)cnt); break;
                    X(32) X(64) X(128) X(256) X(512) X(1024)
#undef X
                    default: rc = fail(PBH_ERR_STATE, "k_stft_planar: unexpected segment length");
                }
            }
            PBHCHECK(rc);
            IoLayout io;
            io.in_layout = PBH_LAYOUT_SERIES_MAJOR;
            io.in_pitch = p->N;
            io.out_layout = out_layout;
            io.out_pitch = out_pitch;
            auto steps = build_steps(p, (const cf*)p->work, (cf*)out_dev, DetectTail(), io);
            return run_steps(steps, p->stream);
        }
    }
#endif
    // unfused: channelise into the plan's staging buffer, then the ordinary pipeline
    const size_t bytes = sizeof(cf) * (size_t)p->S * (size_t)p->N;
    PBHCHECK(ensure_stage(p, &p->stage_in, &p->stage_in_bytes, bytes));
    PBHCHECK(PBH_FN(stft)(p->device, p->stream, 0, in_dev, p->stage_in, nseg, M, nchan_in, E, 0, PBH_DEVICE, PBH_DEVICE));
    IoLayout io;
    io.out_layout = out_layout;
    io.out_pitch = out_pitch;
    auto steps = build_steps(p, (const cf*)p->stage_in, (cf*)out_dev, DetectTail(), io);
    return run_steps(steps, p->stream);
}

// coherent_dedispersion followed by contrib.istft in one call (transforms/dedispersion.py:125, then
// pulsarbat/contrib/misc.py:58-93): `plan` is the dedispersion plan of the CHANNELISED block (nseg, nchan_out*nperseg,
// inner), the output the (nout*nperseg, nchan_out, inner) time series of the synthesis filterbank, nout = stop - start.
// Where the geometry allows (float32, nperseg = 2^m, a multi-pass plan, tiles that write at least 64-byte runs) the
// dedispersion's last column pass leaves its cropped result series-major in the plan's staging buffer and k_istft_planar
// reads that: no re-interleave pass, no reading the channelised result back.  Every other geometry runs the two steps.
int PBH_FN(dedisperse_istft)(pbh_plan* p, const void* in_dev, int in_layout, int64_t in_pitch, int nperseg, int nchan_out,
                             void* out_dev) {
    if (!p || !in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    if (nperseg <= 0 || nchan_out <= 0 || (int64_t)nchan_out * nperseg != p->nchan)
        return fail(PBH_ERR_INVALID, "pbh_dedisperse_istft: the plan must have nchan = nchan_out * nperseg channels");
    if (in_layout != PBH_LAYOUT_SAMPLE_MAJOR && in_layout != PBH_LAYOUT_SERIES_MAJOR) return fail(PBH_ERR_INVALID, "bad layout");
    if (in_layout == PBH_LAYOUT_SERIES_MAJOR && in_pitch < p->N) return fail(PBH_ERR_INVALID, "in_pitch < nsample");
    const int64_t nout = p->stop - p->start;
    HIPCHECK(hipSetDevice(p->device));
    if (nout <= 0) return PBH_OK;
    const int E = p->npol, M = nperseg;
    const int64_t Sout = (int64_t)nchan_out * E;
    const bool multipass = !(p->bsL || p->mixed || p->N1 == 1 || p->P > 1 || p->N1 > kTilePoints || p->N2 % (kTilePoints / p->N1) != 0 ||
                             p->N >= (1LL << 31));
    if (in_layout == PBH_LAYOUT_SERIES_MAJOR && !multipass)
        return fail(PBH_ERR_UNSUPPORTED, "series-major input needs a multi-pass power-of-two plan (nsample > one tile)");
    IoLayout io;
    io.in_layout = in_layout;
    io.in_pitch = in_pitch;
#ifndef PBH_F64
    static const bool fuse_on = [] { const char* e = diag_env("PBH_ISTFT_FUSE"); return e ? atoi(e) != 0 : true; }();
    if (fuse_on && multipass && is_pow2(M) && M >= PBH_R && M * 16 <= kTilePoints) {
        const int F = kTilePoints / M;
        int SB = F / 16;
        while (SB > 1 && (Sout % SB != 0 || SB > Sout)) SB >>= 1;
        const int G = F / SB;
        // the tile stores runs of SB series: below 64 bytes (or a part of the series only, narrower than that) the partial-line
        // writes cost more than the pass the fusion saves
        int nsub = 1;   // as in pbh_stft_dedisperse: narrow tiles are done sibling after sibling by one workgroup
        if (SB < 8 && SB != Sout) {
            nsub = (int)(Sout / SB);
            while (nsub > 16 || (Sout / SB) % nsub != 0) --nsub;
        }
        if (Sout / SB <= 16383 && (SB >= 8 || SB == Sout || (stft_sibling_loop() && SB >= 2 && nsub * SB * (int)sizeof(cf) >= 64)) &&
            (int64_t)G * M * Sout * (int64_t)sizeof(cf) < (1LL << 31)) {
            // series-major result of the dedispersion: row q' at q'*pitch, its first kept sample on a 128-byte line
            const int64_t lead = p->start % 16, pitch = (nout + lead + 15) / 16 * 16;
            PBHCHECK(ensure_stage(p, &p->stage_out, &p->stage_out_bytes, sizeof(cf) * (size_t)p->S * (size_t)pitch));
            cf* mid = (cf*)p->stage_out + lead;
            io.out_layout = PBH_LAYOUT_SERIES_MAJOR;
            io.out_pitch = pitch;
            auto steps = build_steps(p, (const cf*)in_dev, mid, DetectTail(), io);
            PBHCHECK(run_steps(steps, p->stream));
            IstftPlanarParams sp{(const cf*)mid, (cf*)out_dev, p->tw16k, pitch, nout, (int)Sout, E, SB, G};
            sp.nsub = nsub;
            const int64_t ngrp = (nout + G - 1) / G;
            const size_t lds = (size_t)kTilePoints / 16 * 17 * sizeof(cf);   // staging rows padded to G + 1 slots, G >= 16 (one size: launch_tile_kernel sets the limit once)
            int rc = PBH_OK;
            for (int64_t y0 = 0; y0 < ngrp && rc == PBH_OK; y0 += 65535) {
                const int64_t cnt = ngrp - y0 < 65535 ? ngrp - y0 : 65535;
                IstftPlanarParams q = sp;
                q.in = sp.in + y0 * G;
                q.out = sp.out + y0 * G * (int64_t)M * Sout;
                q.nseg = nout - y0 * G;
                switch (M) {
#define X(m) case m: rc = launch_tile_kernel(k_istft_planar<m, PBH_R>, q, Sout / SB / nsub, kTilePoints / PBH_R, p->stream, lds, (unsigned)cnt); break;
                    X(32) X(64) X(128) X(256) X(512) X(1024)
#undef X
                    default: rc = fail(PBH_ERR_STATE, "k_istft_planar: unexpected segment length");
                }
            }
            return rc;
        }
    }
#endif
    // unfused: dedisperse into the plan's staging buffer (reference layout), then the synthesis filterbank
    PBHCHECK(ensure_stage(p, &p->stage_out, &p->stage_out_bytes, sizeof(cf) * (size_t)p->S * (size_t)nout));
    auto steps = build_steps(p, (const cf*)in_dev, (cf*)p->stage_out, DetectTail(), io);
    PBHCHECK(run_steps(steps, p->stream));
    return PBH_FN(stft)(p->device, p->stream, 0, p->stage_out, out_dev, nout, M, nchan_out, E, 1, PBH_DEVICE, PBH_DEVICE);
}

static int launch_place(hipStream_t st, const cf* src, int64_t ipitch, cf* dst, int64_t opitch, int64_t nrow, int ncol) {
    // 16-byte vectors when every row of both sides starts and ends on one
    const bool v16 = sizeof(cf) == 16 || (ncol % 2 == 0 && ipitch % 2 == 0 && opitch % 2 == 0 &&
                                          (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    const int64_t units = v16 && sizeof(cf) == 8 ? nrow * (ncol / 2) : nrow * ncol;
    int64_t blocks = (units + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (v16 && sizeof(cf) == 8)
        hipLaunchKernelGGL((k_place<float4>), dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)src, ipitch / 2, (float4*)dst,
                           opitch / 2, nrow, ncol / 2);
    else
        hipLaunchKernelGGL((k_place<cf>), dim3((unsigned)blocks), dim3(256), 0, st, src, ipitch, dst, opitch, nrow, ncol);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

// true when the plan's last kernel can write pitched rows itself (launch_reinterleave's row transposes / generic kernel)
static bool slice_fast_ok(const pbh_plan* p) {
    if (p->bsL || p->N1 == 1 || resolved_variant(p) != PBH_VARIANT_PLANAR5) return false;
    if (p->P > 1 && !p->mixed && radix_layout_ok(p->S, p->P, p->N, p->N2)) return false;   // the inverse radix stage rides in the layout pass
    return blk_series(p->S, p->N) == 0;
}

// The multi-GPU gather: a rank's (nout, nchan_local, npol) result is written straight into its
// channel slice of the full-band (nout, nchan_total, npol) block -- which may live on a peer GPU (pbh_node_import) --
// by the pipeline's last kernel, instead of transpose + all-gather + concatenate + transpose afterwards.
int pbh_dedisperse_slices(pbh_plan* p, const void* in_dev, int nparts, void* const* part_dev, const int64_t* part_row,
                          int64_t out_row_elems, int64_t out_col_offset);
int pbh_dedisperse_slice(pbh_plan* p, const void* in_dev, void* out_dev, int64_t out_row_elems, int64_t out_col_offset) {
    if (!p) return fail(PBH_ERR_INVALID, "NULL argument");
    const int64_t rows[2] = {0, p->stop > p->start ? p->stop - p->start : 0};
    void* const parts[1] = {out_dev};
    return pbh_dedisperse_slices(p, in_dev, 1, parts, rows, out_row_elems, out_col_offset);
}

// The same with the destination's rows split over nparts buffers: part i receives output rows [part_row[i],
// part_row[i+1]) (part_row[0] = 0, part_row[nparts] = nout) at part_dev[i], row r of the part at part_dev[i] +
// (r - part_row[i])*out_row_elems.  A destination block larger than 2 GiB has to be several allocations when peers map
// it (pbh_node_import hangs on larger ones with this ROCm stack); the pipeline runs once, its last pass once per part.
int pbh_dedisperse_slices(pbh_plan* p, const void* in_dev, int nparts, void* const* part_dev, const int64_t* part_row,
                          int64_t out_row_elems, int64_t out_col_offset) {
    if (!p || !in_dev || !part_dev || !part_row || nparts < 1) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    if (out_col_offset < 0 || out_row_elems < out_col_offset + p->S)
        return fail(PBH_ERR_INVALID, "pbh_dedisperse_slice: the slice [offset, offset + nchan*npol) does not fit the output row");
    HIPCHECK(hipSetDevice(p->device));
    const int64_t nout = p->stop - p->start;
    if (nout <= 0) return PBH_OK;
    if (part_row[0] != 0 || part_row[nparts] != nout) return fail(PBH_ERR_INVALID, "pbh_dedisperse_slices: the parts must cover rows [0, nout)");
    IoLayout io;
    for (int i = 0; i < nparts; ++i) {
        if (part_row[i + 1] < part_row[i]) return fail(PBH_ERR_INVALID, "pbh_dedisperse_slices: part rows must not decrease");
        if (part_row[i + 1] > part_row[i] && !part_dev[i]) return fail(PBH_ERR_INVALID, "pbh_dedisperse_slices: NULL part");
        io.part_ptr.push_back((cf*)part_dev[i] + out_col_offset);
        io.part_row.push_back(part_row[i]);
    }
    io.part_row.push_back(nout);
    const bool compact = out_row_elems == p->S && nparts == 1;
    if (compact || slice_fast_ok(p) || (p->N1 == 1 && p->work && single_planar_ok(p))) {
        io.out_row_elems = out_row_elems == p->S ? 0 : out_row_elems;
        cf* first = io.part_ptr[0];
        if (compact) {
            io.part_ptr.clear();
            io.part_row.clear();
        }
        auto steps = build_steps(p, (const cf*)in_dev, first, DetectTail(), io);
        return run_steps(steps, p->stream);
    }
    // other pipelines (single tile, 3-pass, two-axis layout tiles, arbitrary lengths): compact result, then one placing pass
    const size_t bytes = sizeof(cf) * (size_t)p->S * (size_t)nout;
    PBHCHECK(ensure_stage(p, &p->stage_out, &p->stage_out_bytes, bytes));
    auto steps = build_steps(p, (const cf*)in_dev, (cf*)p->stage_out);
    PBHCHECK(run_steps(steps, p->stream));
    for (int i = 0; i < nparts; ++i) {
        const int64_t r0 = part_row[i], r1 = part_row[i + 1];
        if (r1 > r0)
            PBHCHECK(launch_place(p->stream, (const cf*)p->stage_out + r0 * p->S, p->S, io.part_ptr[i], out_row_elems, r1 - r0, p->S));
    }
    return PBH_OK;
}

// 2-D copy between sample-major device arrays of this build's complex dtype: nrow rows of ncol elements, row pitches
// in elements.  dst may be a peer GPU's buffer (pbh_node_import): this is the push of the all-gather.
int pbh_place(int device, void* hip_stream, int /*dtype: this build's*/, const void* src_dev, int64_t src_row_elems,
              void* dst_dev, int64_t dst_row_elems, int64_t nrow, int64_t ncol) {
    if (!src_dev || !dst_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nrow < 0 || ncol < 0 || ncol > 0x7fffffff || src_row_elems < ncol || dst_row_elems < ncol)
        return fail(PBH_ERR_INVALID, "pbh_place: bad geometry");
    if (nrow == 0 || ncol == 0) return PBH_OK;
    HIPCHECK(hipSetDevice(device));
    return launch_place((hipStream_t)hip_stream, (const cf*)src_dev, src_row_elems, (cf*)dst_dev, dst_row_elems, nrow, (int)ncol);
}

// freq_shift (pulsarbat/transforms/transforms.py:337-361): out = IFFT(H * FFT(x * exp(2 pi i ft n))) with the plan's H (the
// band mask of pbh_chirp_special mode 1).  The mixer rides in the de-interleave pass where the plan has one that can
// carry it; otherwise it is ONE out-of-place pass into the plan's staging buffer (no separate copy of the caller's data).
int pbh_dedisperse_mix(pbh_plan* p, const void* in_dev, void* out_dev, const double* ft /*[nchan*npol] host, cycles per sample*/) {
    if (!p || !in_dev || !out_dev || !ft) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no filter: call pbh_chirp_special / pbh_chirp_upload first");
    HIPCHECK(hipSetDevice(p->device));
    if (p->stop <= p->start) return PBH_OK;
    if (!p->mix_ft) PBHCHECK(dev_alloc(p, (void**)&p->mix_ft, sizeof(double) * (size_t)p->S));
    HIPCHECK(hipMemcpyAsync(p->mix_ft, ft, sizeof(double) * (size_t)p->S, hipMemcpyHostToDevice, p->stream));
    HIPCHECK(hipStreamSynchronize(p->stream));   // ft is a borrowed host array
    const bool fold = !p->bsL && p->N1 > 1 && p->P == 1 && resolved_variant(p) == PBH_VARIANT_PLANAR5 && deint_can_mix(p->S, p->N);
    if (fold) {
        IoLayout io;
        io.mix_ft = p->mix_ft;
        auto steps = build_steps(p, (const cf*)in_dev, (cf*)out_dev, DetectTail(), io);
        return run_steps(steps, p->stream);
    }
    const size_t in_bytes = sizeof(cf) * (size_t)p->S * (size_t)p->N;
    PBHCHECK(ensure_stage(p, &p->stage_in, &p->stage_in_bytes, in_bytes));
    int64_t blocks = (p->N * p->S + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_mix, dim3((unsigned)blocks), dim3(256), 0, p->stream, (const cf*)in_dev, (cf*)p->stage_in,
                       (const double*)p->mix_ft, p->N, p->S);
    HIPCHECK(hipGetLastError());
    auto steps = build_steps(p, (const cf*)p->stage_in, (cf*)out_dev);
    return run_steps(steps, p->stream);
}

extern "C++" int detect_out_elems(int mode, int npol) {
    switch (mode) {
        case PBH_DETECT_INTENSITY: return npol;
        case PBH_DETECT_STOKES_I: return 1;
        case PBH_DETECT_STOKES_LINEAR:
        case PBH_DETECT_STOKES_CIRCULAR: return 4;
    }
    return 0;
}

static int launch_detect(hipStream_t st, const cf* in, real* out, int64_t nout, int nchan, int npol, int mode,
                         int nscrunch) {
    if (nout <= 0) return PBH_OK;
    int64_t blocks = (nout * nchan + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_detect, dim3((unsigned)blocks), dim3(256), 0, st, in, out, nout, nchan, npol, mode, nscrunch);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

int pbh_detect(int device, void* hip_stream, int /*dtype: this build's*/, const void* in_c64, void* out_f32,
               int64_t nsample, int nchan, int npol, int mode, int nscrunch, int in_loc, int out_loc) {
    if (!in_c64 || !out_f32) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nsample <= 0 || nchan <= 0 || npol <= 0 || nscrunch <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    const int oe = detect_out_elems(mode, npol);
    if (!oe) return fail(PBH_ERR_INVALID, "bad detect mode");
    if (mode != PBH_DETECT_INTENSITY && npol != 2) return fail(PBH_ERR_INVALID, "Stokes modes need npol == 2");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t nout = nsample / nscrunch;
    const size_t in_bytes = sizeof(cf) * (size_t)nsample * nchan * npol;
    const size_t out_bytes = sizeof(real) * (size_t)nout * nchan * oe;
    const cf* din = (const cf*)in_c64;
    real* dout = (real*)out_f32;
    void *sin = nullptr, *sout = nullptr;
    int rc = PBH_OK;
    if (in_loc == PBH_HOST) {
        if ((rc = dev_alloc(nullptr, &sin, in_bytes)) != PBH_OK) return rc;
        if (xfer_h2d(sin, in_c64, in_bytes, st) != hipSuccess) {
            (void)hipFree(sin);
            return fail(PBH_ERR_HIP, "host-to-device copy of the input failed");
        }
        din = (const cf*)sin;
    }
    if (out_loc == PBH_HOST) {
        if ((rc = dev_alloc(nullptr, &sout, out_bytes ? out_bytes : 16)) != PBH_OK) {
            if (sin) (void)hipFree(sin);
            return rc;
        }
        dout = (real*)sout;
    }
    rc = launch_detect(st, din, dout, nout, nchan, npol, mode, nscrunch);
    hipError_t e = hipSuccess;
    if (rc == PBH_OK && out_loc == PBH_HOST && out_bytes)
        e = xfer_d2h(out_f32, dout, out_bytes, st);
    if (in_loc == PBH_HOST || out_loc == PBH_HOST) {
        hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e2;
    }
    if (sin) (void)hipFree(sin);
    if (sout) (void)hipFree(sout);
    if (rc != PBH_OK) return rc;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_detect: ") + hipGetErrorString(e));
    return PBH_OK;
}

int pbh_dedisperse_detect(pbh_plan* p, const void* in_c64, void* out_f32, int nscrunch, int mode, int in_loc,
                          int out_loc) {
    if (!p || !in_c64 || !out_f32) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nscrunch <= 0) return fail(PBH_ERR_INVALID, "nscrunch must be positive");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    const int oe = detect_out_elems(mode, p->npol);
    if (!oe) return fail(PBH_ERR_INVALID, "bad detect mode");
    if (mode != PBH_DETECT_INTENSITY && p->npol != 2) return fail(PBH_ERR_INVALID, "Stokes modes need npol == 2");
    HIPCHECK(hipSetDevice(p->device));
    const int64_t nvalid = p->stop - p->start;
    const int64_t nout = nvalid / nscrunch;
    const size_t out_bytes = sizeof(real) * (size_t)nout * p->nchan * oe;
    const size_t mid_bytes = sizeof(cf) * (size_t)p->S * (size_t)nvalid;
    const cf* din;
    void* dout;
    PBHCHECK(resolve_io(p, in_c64, out_f32, out_bytes, in_loc, out_loc, &din, &dout));
    if (nout > 0 && can_fuse_detect(p, nscrunch, mode)) {
        DetectTail tail;
        tail.out = (real*)dout;
        tail.mode = mode;
        tail.nscrunch = nscrunch;
        auto steps = build_steps(p, din, nullptr, tail);
        PBHCHECK(run_steps(steps, p->stream));
    } else if (nout > 0) {
        // two-step form: dedisperse into a buffer the plan keeps, then detect + scrunch
        PBHCHECK(ensure_stage(p, &p->det_mid, &p->det_mid_bytes, mid_bytes));
        auto steps = build_steps(p, din, (cf*)p->det_mid);
        PBHCHECK(run_steps(steps, p->stream));
        PBHCHECK(launch_detect(p->stream, (const cf*)p->det_mid, (real*)dout, nout, p->nchan, p->npol, mode, nscrunch));
    }
    if (out_loc == PBH_HOST && out_bytes)
        HIPCHECK(xfer_d2h(out_f32, dout, out_bytes, p->stream));
    if (in_loc == PBH_HOST || out_loc == PBH_HOST) HIPCHECK(hipStreamSynchronize(p->stream));
    return PBH_OK;
}

// pbh_dedisperse_detect for a device-resident input with a stated layout (series-major inputs skip the
// de-interleave pass; the fused detect tail already reads the planar workspace): 4 kernels.
int pbh_dedisperse_detect_layout(pbh_plan* p, const void* in_dev, int in_layout, int64_t in_pitch, void* out_dev,
                                 int nscrunch, int mode) {
    if (!p || !in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nscrunch <= 0) return fail(PBH_ERR_INVALID, "nscrunch must be positive");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    if (in_layout != PBH_LAYOUT_SAMPLE_MAJOR && in_layout != PBH_LAYOUT_SERIES_MAJOR) return fail(PBH_ERR_INVALID, "bad layout");
    const int oe = detect_out_elems(mode, p->npol);
    if (!oe) return fail(PBH_ERR_INVALID, "bad detect mode");
    if (mode != PBH_DETECT_INTENSITY && p->npol != 2) return fail(PBH_ERR_INVALID, "Stokes modes need npol == 2");
    if (in_layout == PBH_LAYOUT_SERIES_MAJOR) {
        if (in_pitch < p->N) return fail(PBH_ERR_INVALID, "in_pitch < nsample");
        if ((!p->mixed && (p->bsL || p->N1 == 1 || p->N1 / p->P > kTilePoints || p->N2 % (kTilePoints / (p->N1 / p->P)) != 0)) ||
            p->N >= (1LL << 31) || !(nscrunch % 64 == 0 || can_fuse_detect(p, nscrunch, mode)) ||
            p->nchan > 65535)
            return fail(PBH_ERR_UNSUPPORTED, "series-major input needs a multi-pass plan and a fused detect tail (nscrunch % 64 == 0, or 1)");
    } else {
        return pbh_dedisperse_detect(p, in_dev, out_dev, nscrunch, mode, PBH_DEVICE, PBH_DEVICE);
    }
    HIPCHECK(hipSetDevice(p->device));
    if ((p->stop - p->start) / nscrunch <= 0) return PBH_OK;
    DetectTail tail;
    tail.out = (real*)out_dev;
    tail.mode = mode;
    tail.nscrunch = nscrunch;
    IoLayout io;
    io.in_layout = in_layout;
    io.in_pitch = in_pitch;
    auto steps = build_steps(p, (const cf*)in_dev, nullptr, tail, io);
    PBHCHECK(run_steps(steps, p->stream));
    return PBH_OK;
}

// Lengths beyond one tile, or not a power of two: Bluestein ring as a plain transform
// (forward X = b * conv(x b); inverse x = conj(b * conv(conj(X) b)) / n).  Plans are cached per thread.
// per-thread cache of plain-transform ring plans (n points, `batch` columns)
// Per-thread caches of the plans behind the plan-less entry points (pbh_fft_c2c, pbh_stft): two Bluestein ring
// plans and two native transform plans, each owning a workspace the size of its data.  pbh_trim() frees them.
struct CachedPlan { int device; int64_t n, batch; pbh_plan* plan; };
// stage twiddle table W_16384^p of the stand-alone one-tile transforms: one per (thread, device), freed by pbh_trim
static thread_local cf* g_tw_table[16] = {};
static int standalone_twiddles(int device, cf** out) {
    if (device < 0 || device >= 16) return fail(PBH_ERR_INVALID, "device index out of range");
    if (!g_tw_table[device]) {
        std::vector<cf> h(kTwTable);
        for (int i = 0; i < kTwTable; ++i) {
            double a = -2.0 * M_PI * (double)i / (double)kTwTable;
            h[i] = make_cf((real)cos(a), (real)sin(a));
        }
        cf* t = nullptr;
        PBHCHECK(dev_alloc(nullptr, (void**)&t, sizeof(cf) * kTwTable));
        if (hipMemcpy(t, h.data(), sizeof(cf) * kTwTable, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(t);
            return fail(PBH_ERR_HIP, "hipMemcpy(twiddles) failed");
        }
        g_tw_table[device] = t;
    }
    *out = g_tw_table[device];
    return PBH_OK;
}
static thread_local CachedPlan g_ring_cache[2] = {{-1, 0, 0, nullptr}, {-1, 0, 0, nullptr}};
static thread_local CachedPlan g_native_cache[2] = {{-1, 0, 0, nullptr}, {-1, 0, 0, nullptr}};

int pbh_trim(void) {
    for (auto* cache : {g_ring_cache, g_native_cache})
        for (int i = 0; i < 2; ++i)
            if (cache[i].plan) {
                pbh_plan_destroy(cache[i].plan);
                cache[i] = CachedPlan{-1, 0, 0, nullptr};
            }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < 16; ++d)
        if (g_tw_table[d]) {   // freed on the device that owns it, after its work has drained
            if (hipSetDevice(d) == hipSuccess) {
                (void)hipDeviceSynchronize();   // best effort: the table is freed either way
                (void)hipFree(g_tw_table[d]);
            }
            g_tw_table[d] = nullptr;
        }
    for (int d = 0; d < 16; ++d) {   // stream-ordered scratch memory kept by the default pools (keep_pool_memory)
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, d) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
        (void)hipGetLastError();
    }
    if (have_cur) (void)hipSetDevice(cur);
    release_bounce_buffers();
    return PBH_OK;
}

static int ring_plan(int device, int64_t n, int64_t batch, pbh_plan** out) {
    typedef CachedPlan Entry;
    CachedPlan* cache = g_ring_cache;
    static thread_local int next = 0;
    for (int i = 0; i < 2; ++i) {
        auto& e = cache[i];
        if (e.plan && e.device == device && e.n == n && e.batch == batch) {
            *out = e.plan;
            return PBH_OK;
        }
    }
    if (batch > 0x7fffffffLL) return fail(PBH_ERR_UNSUPPORTED, "too many columns for one ring plan");
#ifdef PBH_F64
    const int dt = PBH_C128;
#else
    const int dt = PBH_C64;
#endif
    pbh_plan* p = nullptr;
    PBHCHECK(create_plan(&p, device, n, 1, (int)batch, dt, 0, n, 1));
    Entry& e = cache[next];
    next ^= 1;
    if (e.plan) pbh_plan_destroy(e.plan);
    e = Entry{device, n, batch, p};
    *out = p;
    return PBH_OK;
}

// Stand-alone transform of a native length (2^k beyond one tile, or m * 2^k) without the Bluestein detour:
// de-interleave (+ radix stage), column pass, row transforms, natural-order output (k_fft_out).  Plans
// (workspace and twiddles only) are cached per thread like the ring plans.
static int native_fft_plan(int device, int64_t n, int64_t batch, pbh_plan** out) {
    typedef CachedPlan Entry;
    CachedPlan* cache = g_native_cache;
    static thread_local int next = 0;
    for (int i = 0; i < 2; ++i) {
        auto& e = cache[i];
        if (e.plan && e.device == device && e.n == n && e.batch == batch) {
            *out = e.plan;
            return PBH_OK;
        }
    }
#ifdef PBH_F64
    const int dt = PBH_C128;
#else
    const int dt = PBH_C64;
#endif
    pbh_plan* p = nullptr;
    PBHCHECK(create_plan(&p, device, n, (int)batch, 1, dt, 0, n, 2));
    Entry& e = cache[next];
    next ^= 1;
    if (e.plan) pbh_plan_destroy(e.plan);
    e = Entry{device, n, batch, p};
    *out = p;
    return PBH_OK;
}

static bool native_fft_ok(int64_t n, int64_t batch) {
    static const bool on = [] { const char* e = diag_env("PBH_NATIVE_FFT"); return e ? atoi(e) != 0 : true; }();
    if (!on || batch > 65535 || n <= kTilePoints || n > (1LL << 28)) return false;
    if ((is_pow2(n) && n >= 2 * (int64_t)kTilePoints) || native_odd_factor(n) != 0) return true;
    int n1, n2, pp;   // 7-smooth lengths with rows the stand-alone row transform has, or with mixed-radix rows (mixed_kernels.hpp)
    if (mixed_geometry(n, &n1, &n2, &pp) && n2 >= 1024) return true;
    return rowmix_geometry(n, &n1, &n2, &pp);
}

// Forward transforms of a native-length plan `p` (batch = p->S series of p->N samples) up to plan order in p->work.
// The input is `il` interleaved streams of `total` = (p->S / il) * p->N samples each: de-interleaved, stream s
// becomes the batch rows s*(p->S/il) .. (pbh_fft_c2c: il = batch, one row per stream; stft: rows = segments).
static int native_forward(pbh_plan* p, const cf* din, int il, hipStream_t st) {
    const int S = p->S, N1 = p->N1, N2 = p->N2, P = p->P, Q = N1 / P;
    const int64_t n = p->N, total = (int64_t)(S / il) * n;
    cf* work = p->work;
    if (p->mixed) {   // 7-smooth length: k_colmix plays the column roles, the rows are the engine's
        const cf* src = din;
        if (il > 1) {
            PBHCHECK(launch_deinterleave(din, work, total, il, total, st));
            src = work;
        }
        unsigned* ctr = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
        HIPCHECK(hipMemsetAsync(ctr, 0, kCounterBytes, st));
        if (P > 1) {
            if (mix_radix_p(P)) {
                PBHCHECK(launch_radix<-1>(P, src, n, work, n, S, n, N2, N1, 0, n, st));
            } else {
                MixParams a = mix_role_a(p, src, n, work, n, 0, n, 0);
                a.counter = ctr + 3;
                PBHCHECK(launch_colmix<-1>(a, st));
            }
            src = work;
        }
        MixParams b = mix_role_b(p, src, n, work, n, 0, n, 0);
        b.counter = ctr + 4;
        PBHCHECK(launch_colmix<-1>(b, st));
        if (p->rowmix) return launch_rowmix(p, work, st, true);   // rows left in digit-reversed order (k_fft_out undoes it)
        return launch_rowfft(N2, work, p->tw16k, (int64_t)S * N1, st);
    }
    BigTwiddle tw{p->tw_hi, p->tw_lo, p->tw_shift, n - 1};
    tw.nmod = is_pow2(n) ? 0 : n;
    const bool fuse = P > 1 && il == S && S > 1 && radix_layout_ok(S, P, n, N2);
    const cf* src = work;   // what the column pass reads
    if (fuse) {
        PBHCHECK(launch_deint_radix(S, P, din, work, n, N2, N1, n, st));
    } else {
        if (il > 1) PBHCHECK(launch_deinterleave(din, work, total, il, total, st));   // (one stream is its own planar form)
        if (P > 1) PBHCHECK(launch_radix<-1>(P, il > 1 ? work : din, n, work, n, S, n, N2, N1, 0, n, st));
        else if (il == 1) src = din;
    }
    const bool colq = P > 1 || (Q >= 64 && Q <= kTilePoints && N2 % (kTilePoints / Q) == 0 && n < (1LL << 31));
    if (colq) {
        unsigned* ctr = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
        HIPCHECK(hipMemsetAsync(ctr, 0, 3 * sizeof(unsigned), st));
        ColpParams cp{work, n, S, N2, tw, p->tw16k, 0, n, 0, ctr};
        cp.P = P;
        if (src != work) {
            cp.ld = src;
            cp.ld_plane = n;
        }
        PBHCHECK(launch_colq<OP_FWD_TW>(Q, cp, st));
    } else {
        ColSide planar{LAYOUT_PLANAR, n, N2};
        ColParams c1{src, work, planar, planar, LAYOUT_PLANAR, 0, 0, S, N2, (int64_t)S * N2, 0, tw, p->tw16k, 0, n, 0};
        PBHCHECK(launch_col<OP_FWD_TW>(N1, c1, st));
    }
    return launch_rowfft(N2, work, p->tw16k, (int64_t)S * N1, st);
}

static int fft_c2c_native(int device, hipStream_t st, const cf* din, cf* dout, int64_t n, int64_t batch, int inverse) {
    pbh_plan* p = nullptr;
    PBHCHECK(native_fft_plan(device, n, batch, &p));
    PBHCHECK(native_forward(p, din, p->S, st));
    constexpr int SB = 64, TB = kBlkElems / 64;
    const int64_t ncol = (int64_t)p->N1 * p->S;
    hipLaunchKernelGGL((k_fft_out<SB, TB>), dim3((unsigned)((ncol + SB - 1) / SB), (unsigned)((p->N2 + TB - 1) / TB)), dim3(256), 0, st,
                       (const cf*)p->work, dout, p->N1, p->N2, p->S, p->P, inverse, inverse ? (real)(1.0 / (double)n) : (real)1,
                       p->rowmix ? p->mixR.perm : (const unsigned short*)nullptr);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

#ifndef PBH_F64
// utils.real_to_complex (pulsarbat/utils.py:15-65) of device-resident float32 data as a half-length complex transform
// (aux_kernels.hpp, "utils.real_to_complex as a HALF-LENGTH complex transform"): in (nreal, nseries) float32 sample-major,
// out (nreal/2, nseries) complex64 sample-major.  Geometries: nreal/2 a power of two the native column / row passes
// transform unsplit (2^15 .. 2^24); anything else returns PBH_ERR_UNSUPPORTED and the caller takes the full-length route.
int pbh_real_to_complex(int device, void* hip_stream, const void* in_dev, void* out_dev, int64_t nreal, int nseries) {
    if (!in_dev || !out_dev) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nreal <= 0 || nseries <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    const int64_t M = nreal / 2;
    if (nreal % 2 || !is_pow2(M) || !native_fft_ok(M, nseries) || nseries > 65535)
        return fail(PBH_ERR_UNSUPPORTED, "half-length real_to_complex: nreal / 2 must be a power of two beyond one tile");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    pbh_plan* p = nullptr;
    PBHCHECK(native_fft_plan(device, M, nseries, &p));
    const int S = p->S, N1 = p->N1, N2 = p->N2;
    if (p->mixed || p->P != 1 || N2 % 256 != 0 || !(N1 >= 64 && N1 <= kTilePoints && N2 % (kTilePoints / N1) == 0) || M >= (1LL << 31)) {
        // (short column transforms, N1 < 64, run the one-tile column kernel: not wired here)
        if (!(p->P == 1 && !p->mixed && N1 < 64)) return fail(PBH_ERR_UNSUPPORTED, "half-length real_to_complex: plan geometry");
    }
    cf* work2 = ensure_work2(p);
    if (!work2) return fail(PBH_ERR_NOMEM, "half-length real_to_complex: second workspace");
    // 1. the real series, time fastest = the packed complex series, planar
    const cf* src = (const cf*)in_dev;
    if (S > 1) {
        hipLaunchKernelGGL(k_real_planar, dim3((unsigned)((nreal + 63) / 64), (unsigned)((S + 63) / 64)), dim3(256), 0, st,
                           (const float*)in_dev, (float*)p->work, nreal, S, nreal);
        HIPCHECK(hipGetLastError());
        src = p->work;
    }
    // 2. P = FFT_M(p), plan order, in the workspace
    PBHCHECK(native_forward(p, src, 1, st));
    // 3. C from P and its mirror, scaled by 1/M, into the second workspace
    hipLaunchKernelGGL(k_r2c_mirror, dim3((unsigned)((N1 / 2 + 1) * (N2 / 256)), (unsigned)S), dim3(256), 0, st, (const cf*)p->work,
                       work2, N1, N2, (real)(1.0 / (double)M));
    HIPCHECK(hipGetLastError());
    // 4. z = IFFT_M(C): inverse rows, inverse columns (as the dedispersion's second half), sample-major out
    PBHCHECK(launch_rowfft(N2, work2, p->tw16k, (int64_t)S * N1, st, true));
    BigTwiddle tw{p->tw_hi, p->tw_lo, p->tw_shift, M - 1};
    unsigned* ctr = reinterpret_cast<unsigned*>(p->tw16k + kTwTable);
    if (N1 >= 64) {
        HIPCHECK(hipMemsetAsync(ctr, 0, 3 * sizeof(unsigned), st));
        ColpParams cp{work2, M, S, N2, tw, p->tw16k, 0, M, 0, ctr + 1};
        PBHCHECK(launch_colq<OP_TW_INV>(N1, cp, st));
    } else {
        ColSide planar{LAYOUT_PLANAR, M, N2};
        ColParams c3{work2, work2, planar, planar, LAYOUT_PLANAR, 0, 0, S, N2, (int64_t)S * N2, 0, tw, p->tw16k, 0, M, 0};
        PBHCHECK(launch_col<OP_TW_INV>(N1, c3, st));
    }
    if (S > 1) return launch_reinterleave(work2, (cf*)out_dev, 0, M, S, M, st);
    HIPCHECK(hipMemcpyAsync(out_dev, work2, sizeof(cf) * (size_t)M, hipMemcpyDeviceToDevice, st));
    return PBH_OK;
}
#endif

// forward stft through k_stft_fwd: power-of-two segments of at most tile/16 points whose tiles cover whole segments (S <=
// tile/n) or whole channels of one segment (tile/n divides S and holds whole channels)
static bool stft_linear_ok(int64_t n, int64_t S, int inner) {
    static const bool on = [] { const char* e = diag_env("PBH_STFT_LINEAR"); return e ? atoi(e) != 0 : true; }();
    if (!on || !is_pow2(n) || n < PBH_R || n * 32 > kTilePoints) return false;   // (at tile/16 the plain kernel is as fast or faster)
    const int64_t F = kTilePoints / n;
    if (S <= F) return true;
    return S % F == 0 && F % inner == 0 && n * S * (int64_t)sizeof(cf) < 0x7fffffffLL;
}

static bool stft_pair_enabled() {
    static const bool on = [] { const char* e = diag_env("PBH_STFT_PAIR"); return e ? atoi(e) != 0 : true; }();
    return on;
}

// contrib.stft / istft with native segment lengths beyond one tile: the segments are the batch
static bool stft_native_ok(int64_t n, int64_t nseg, int64_t S) {
    static const bool on = [] { const char* e = diag_env("PBH_NATIVE_FFT"); return e ? atoi(e) != 0 : true; }();
    if (!on || n <= kTilePoints || n > (1LL << 27) || nseg * S > 0x7fffffffLL || nseg * S * (n / kTilePoints + 1) / 64 > 0x7fffffffLL)
        return false;
    if ((is_pow2(n) && n >= 2 * (int64_t)kTilePoints) || native_odd_factor(n) != 0) return true;
    int n1, n2, pp;
    if (mixed_geometry(n, &n1, &n2, &pp) && n2 >= 1024) return true;
    // (even lengths only: the inverse pass undoes the fftshift by the sign (-1)^k, which is what it is for even n alone)
    return n % 2 == 0 && rowmix_geometry(n, &n1, &n2, &pp);
}

static int stft_native(int device, hipStream_t st, const cf* din, cf* dout, int64_t nseg, int64_t n, int nchan, int inner,
                       int inverse) {
    const int64_t B = nseg * nchan * inner;
    pbh_plan* p = nullptr;
    PBHCHECK(native_fft_plan(device, n, B, &p));
    // stft: the (nseg*n, S) array de-interleaves into S streams whose segments are the batch rows (s*nseg + g);
    // istft: the (nseg*nchan*n, E) array into E streams of nseg*nchan rows
    PBHCHECK(native_forward(p, din, inverse ? inner : nchan * inner, st));
    constexpr int TB = 64;
    const int64_t NJ = B * p->N1;
    hipLaunchKernelGGL((k_stft_out<TB>), dim3((unsigned)((NJ + 63) / 64), (unsigned)((p->N2 + TB - 1) / TB)), dim3(256), 0, st,
                       (const cf*)p->work, dout, p->N1, p->N2, p->P, nseg, nchan, inner, inverse,
                       inverse ? (real)1 : (real)(1.0 / (double)n), p->rowmix ? p->mixR.perm : (const unsigned short*)nullptr);
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

static int fft_c2c_ring(int device, hipStream_t st, const cf* din, cf* dout, int64_t n, int64_t batch, int inverse) {
    pbh_plan* p = nullptr;
    PBHCHECK(ring_plan(device, n, batch, &p));
    const int64_t L = p->bsL;
    const int S = p->S;
    auto grid = [](int64_t m) { int64_t g = (m + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
    if (inverse)
        hipLaunchKernelGGL(k_bs_pre_conj, dim3(grid(L * S)), dim3(256), 0, st, din, (const cf*)p->bs_b, p->bs_a, n, L, S);
    else
        hipLaunchKernelGGL(k_bs_pre, dim3(grid(L * S)), dim3(256), 0, st, din, (const cf*)p->bs_b, p->bs_a, n, L, S);
    HIPCHECK(hipGetLastError());
    auto steps = build_steps(p->sub, p->bs_a, p->bs_conv);
    PBHCHECK(run_steps(steps, st));
    hipLaunchKernelGGL(k_bs_post_fft, dim3(grid(n * S)), dim3(256), 0, st, (const cf*)p->bs_conv, (const cf*)p->bs_b, dout,
                       n, S, inverse, (real)(1.0 / (double)n));
    HIPCHECK(hipGetLastError());
    return PBH_OK;
}

int pbh_fft_c2c(int device, void* hip_stream, int /*dtype: this build's*/, const void* in_c64, void* out_c64,
                int64_t n, int64_t batch, int inverse, int in_loc, int out_loc) {
    if (!in_c64 || !out_c64) return fail(PBH_ERR_INVALID, "NULL argument");
    if (n <= 0 || batch <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    if (batch > 0x7fffffffLL) return fail(PBH_ERR_INVALID, "batch too large");
    if (n == 1 || n > (1LL << 27)) return fail(PBH_ERR_UNSUPPORTED, "pbh_fft_c2c: n must be in [2, 2^27]");
    const bool one_tile = is_pow2(n) && n >= PBH_R && n <= kTilePoints;
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    cf* tw = nullptr;
    if (one_tile) PBHCHECK(standalone_twiddles(device, &tw));
    const size_t bytes = sizeof(cf) * (size_t)n * batch;
    const cf* din = (const cf*)in_c64;
    cf* dout = (cf*)out_c64;
    void *sin = nullptr, *sout = nullptr;
    int rc = PBH_OK;
    if (in_loc == PBH_HOST) {
        PBHCHECK(dev_alloc(nullptr, &sin, bytes));
        if (xfer_h2d(sin, in_c64, bytes, st) != hipSuccess) {
            (void)hipFree(sin);
            return fail(PBH_ERR_HIP, "host-to-device copy of the input failed");
        }
        din = (const cf*)sin;
    }
    if (out_loc == PBH_HOST) {
        if ((rc = dev_alloc(nullptr, &sout, bytes)) != PBH_OK) {
            if (sin) (void)hipFree(sin);
            return rc;
        }
        dout = (cf*)sout;
    }
    if (one_tile) {
        SmallParams sp{din, dout, nullptr, tw, (int)batch, 1, 0, n, inverse ? +1 : -1, (real)(1.0 / (double)n)};
        rc = launch_small((int)n, sp, st);
    } else if (native_fft_ok(n, batch)) {
        rc = fft_c2c_native(device, st, din, dout, n, batch, inverse);
    } else {
        rc = fft_c2c_ring(device, st, din, dout, n, batch, inverse);
    }
    hipError_t e = hipSuccess;
    if (rc == PBH_OK && out_loc == PBH_HOST) e = xfer_d2h(out_c64, dout, bytes, st);
    if (in_loc == PBH_HOST || out_loc == PBH_HOST) {
        hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e2;
    }
    if (sin) (void)hipFree(sin);
    if (sout) (void)hipFree(sout);
    if (rc != PBH_OK) return rc;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_fft_c2c: ") + hipGetErrorString(e));
    return PBH_OK;
}

// ---- contrib.stft / istft (pulsarbat/contrib/misc.py:17-93): boxcar window, no overlap, nfft = nperseg --------
// stft : in (nseg*n, nchan, E) time-ordered  -> out (nseg, nchan*n, E), out[g, c*n + (k + n/2) % n, e] = FFT_k / n
// istft: in (nseg, nchan*n, E) channelised   -> out (nseg*n, nchan, E)   (exact inverse)
int PBH_FN(stft)(int device, void* hip_stream, int /*dtype*/, const void* in, void* out, int64_t nseg, int nperseg,
                 int nchan, int inner, int inverse, int in_loc, int out_loc) {
    if (!in || !out) return fail(PBH_ERR_INVALID, "NULL argument");
    if (nseg <= 0 || nperseg <= 0 || nchan <= 0 || inner <= 0) return fail(PBH_ERR_INVALID, "non-positive size");
    const int64_t S = (int64_t)nchan * inner, n = nperseg;
    if (S > 0x7fffffffLL) return fail(PBH_ERR_INVALID, "too many series");
    HIPCHECK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t bytes = sizeof(cf) * (size_t)nseg * n * S;
    const cf* din = (const cf*)in;
    cf* dout = (cf*)out;
    void *stg_in = nullptr, *stg_out = nullptr;
    int rc = PBH_OK;
    if (in_loc == PBH_HOST) {
        PBHCHECK(dev_alloc(nullptr, &stg_in, bytes));
        if (xfer_h2d(stg_in, in, bytes, st) != hipSuccess) {
            (void)hipFree(stg_in);
            return fail(PBH_ERR_HIP, "host-to-device copy of the input failed");
        }
        din = (const cf*)stg_in;
    }
    if (out_loc == PBH_HOST) {
        if ((rc = dev_alloc(nullptr, &stg_out, bytes)) != PBH_OK) {
            if (stg_in) (void)hipFree(stg_in);
            return rc;
        }
        dout = (cf*)stg_out;
    }
    if (n == 1) {
        // a one-point transform is the identity; both layouts coincide
        hipError_t e1 = hipMemcpyAsync(dout, din, bytes, hipMemcpyDeviceToDevice, st);
        if (e1 != hipSuccess) rc = fail(PBH_ERR_HIP, "hipMemcpyAsync failed");
    } else if (is_pow2(n) && n >= PBH_R && n <= kTilePoints) {
        cf* tw = nullptr;
        rc = standalone_twiddles(device, &tw);
        if (rc == PBH_OK && n == kTilePoints && inner % 2 == 0 && stft_pair_enabled() && n * S * (int64_t)sizeof(cf) < (1LL << 31)) {
            // one segment of one series fills a tile: a workgroup takes both series of a pair (k_seg_pair)
            auto kern = k_seg_pair<kTilePoints, PBH_R>;
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tile_bytes<true>()) != hipSuccess)
                rc = fail(PBH_ERR_HIP, "hipFuncSetAttribute(k_seg_pair) failed");
            for (int64_t g0 = 0; g0 < nseg && rc == PBH_OK; g0 += 65535) {
                const int64_t cnt = nseg - g0 < 65535 ? nseg - g0 : 65535;
                SegPairParams sp{din + g0 * n * S, dout + g0 * n * S, tw, (int)S, inner, inverse ? 1 : 0,
                                 inverse ? (real)1 : (real)(1.0 / (double)n)};
                hipLaunchKernelGGL(kern, dim3((unsigned)(S / 2), (unsigned)cnt), dim3(kTilePoints / PBH_R), lds_tile_bytes<true>(), st, sp);
                if (hipGetLastError() != hipSuccess) rc = fail(PBH_ERR_HIP, "k_seg_pair failed to launch");
            }
        } else if (rc == PBH_OK && !inverse && stft_linear_ok(n, S, inner)) {
            // forward, nperseg <= tile/32: the channelised side is written through LDS in its own order (k_stft_fwd)
            const int64_t F = kTilePoints / n;
            StftFwdParams sp{din, dout, tw, (int)S, inner, 0, 0, nseg, (real)(1.0 / (double)n)};
            if (S <= F) { sp.SB = (int)S; sp.G = (int)(F / S); } else { sp.SB = (int)F; sp.G = 1; }
            const int64_t ngrp = (nseg + sp.G - 1) / sp.G;
            for (int64_t y0 = 0; y0 < ngrp && rc == PBH_OK; y0 += 65535) {
                const int64_t cnt = ngrp - y0 < 65535 ? ngrp - y0 : 65535;
                StftFwdParams q = sp;
                q.in = din + y0 * sp.G * n * S;
                q.out = dout + y0 * sp.G * n * S;
                q.nseg = nseg - y0 * sp.G;
                const size_t ldsb = lds_tile_bytes<false>() + 8192 + 64;
                switch ((int)n) {
#define X(m) case m: rc = launch_tile_kernel(k_stft_fwd<m, PBH_R>, q, S / sp.SB, kTilePoints / PBH_R, st, ldsb, (unsigned)cnt); break;
#ifdef PBH_F64
                    X(16) X(32) X(64) X(128) X(256)
#else
                    X(32) X(64) X(128) X(256) X(512)
#endif
#undef X
                    default: rc = fail(PBH_ERR_STATE, "k_stft_fwd: unexpected segment length");
                }
            }
        } else if (rc == PBH_OK) {
            // a tile holds F = tile/n columns: with F > S it spans F/S whole segments; grid.y walks the
            // segment groups, at most 65535 per launch
            const int64_t F = kTilePoints / n;
            const int segs = F > S ? (int)(F / S) : 1;
            const int64_t per_launch = 65535LL * segs;
            for (int64_t g0 = 0; g0 < nseg && rc == PBH_OK; g0 += per_launch) {
                const int64_t cnt = nseg - g0 < per_launch ? nseg - g0 : per_launch;
                SmallParams sp{din + g0 * n * S, dout + g0 * n * S, nullptr, tw, (int)S, 1, 0, n, inverse ? +1 : -1,
                               inverse ? (real)1 : (real)(1.0 / (double)n)};
                sp.seg_mode = inverse ? 2 : 1;
                sp.E = inner;
                sp.segs = segs;
                sp.nseg = cnt;
                rc = launch_small((int)n, sp, st, (cnt + segs - 1) / segs);
            }
        }
    } else if (stft_native_ok(n, nseg, S)) {
        rc = stft_native(device, st, din, dout, nseg, n, nchan, inner, inverse);
    } else if (n > (1LL << 27)) {
        rc = fail(PBH_ERR_UNSUPPORTED, "nperseg too large");
    } else {
        pbh_plan* p = nullptr;
        rc = ring_plan(device, n, nseg * S, &p);
        if (rc == PBH_OK) {
            const int64_t L = p->bsL;
            auto grid = [](int64_t m) { int64_t g = (m + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
            hipLaunchKernelGGL(k_seg_pre, dim3(grid(L * nseg * S)), dim3(256), 0, st, din, (const cf*)p->bs_b, p->bs_a, n, L,
                               (int)S, inner, nseg, inverse);
            auto steps = build_steps(p->sub, p->bs_a, p->bs_conv);
            rc = run_steps(steps, st);
            if (rc == PBH_OK) {
                hipLaunchKernelGGL(k_seg_post, dim3(grid(n * nseg * S)), dim3(256), 0, st, (const cf*)p->bs_conv,
                                   (const cf*)p->bs_b, dout, n, (int)S, inner, nseg, inverse, (real)(1.0 / (double)n));
                if (hipGetLastError() != hipSuccess) rc = fail(PBH_ERR_HIP, "stft ring kernels failed to launch");
            }
        }
    }
    hipError_t e = hipSuccess;
    if (rc == PBH_OK && out_loc == PBH_HOST) e = xfer_d2h(out, dout, bytes, st);
    if (in_loc == PBH_HOST || out_loc == PBH_HOST) {
        hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e2;
    }
    if (stg_in) (void)hipFree(stg_in);
    if (stg_out) (void)hipFree(stg_out);
    if (rc != PBH_OK) return rc;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("pbh_stft: ") + hipGetErrorString(e));
    return PBH_OK;
}

}  // extern "C"
}  // namespace PBH_NS
