// fd4_kernels.hpp -- the FOUR-pass sample-major schedule (round 4): no de-interleave pass.
//
// planar5 moves the (nsample, S) block through five full-line passes (de-interleave, column, row, column, re-interleave)
// because a 2^14-point column tile of N1 = 1024 rows has only 16 columns: taken from ONE series they are 8-byte pieces of
// the caller's 128-byte lines, taken from all 16 series they are 8-byte pieces of the planar rows.  What a
// partial-line access costs on this chip is not the bytes but the LINES an instruction touches: the CU's address path takes
// ~4 clocks per 128-byte line whatever the instruction uses of it (tools/micro/pp4bench.hip: 2 GiB of 16-byte pieces read
// in 0.92 ms whatever the cache policy = 3.8 clk per piece per CU; 32-byte pieces 0.73 ms and no longer the limit).  So the
// column tile here is 4 series x 4 columns -- 32-byte pieces of the caller's lines -- and the intermediate is stored in
// "Q4" order, whose 128-byte lines hold those 4 series x 4 columns:
//
//     Q4[quad q][row k1][n2 / 4][series s & 3][n2 & 3]        (quad = 4 consecutive series = 2 channels x 2 pols)
//
// so that pass 1 (k_colfd) reads 32-byte pieces and writes FULL lines, and pass 2 (k_rowq16) reads a series' row as 32-byte
// pieces of those lines and writes ordinary planar rows (full lines) into the second work buffer; passes 3 and 4 are
// planar5's inverse column pass and re-interleave pass on that buffer.  4 x (8 + 8) B + phase = what SURVEY.md 8(d) prices.
//
// The 128-byte lines are shared: the S/4 quads' tiles of one column group read the same input lines, and the 4 series of a
// quad read the same Q4 lines.  Sharing only pays when the sharers run AT THE SAME TIME on one XCD (one L2 miss, the rest
// hits; otherwise every sharer fetches the line again: 3.4x the bytes, measured), so both kernels are GANG-scheduled: blocks
// b with equal b % 8 share an XCD under the observed round-robin dispatch, a gang is S/4 (pass 1) or 4 (pass 2) consecutive
// ones of them, and the tile order is static -- the members do identical work and stay within a fraction of a tile of each
// other.  Placement is a speed matter only: every tile is processed exactly once whatever the dispatcher does.
//
// Reference expression: pulsarbat/transforms/dedispersion.py:125 (fft along axis 0 of the (nsample, nchan, npol) block).
#pragma once
#include "kernels.hpp"

namespace PBH_NS {
#ifndef PBH_F64

// gang of `NM` blocks that share an XCD: gang index and member index of block b in a grid of G blocks (G % (8 NM) == 0)
struct Gang { uint32_t gang, member, ngang; };
__device__ __forceinline__ Gang gang_of(uint32_t b, uint32_t G, uint32_t NM) {
    Gang r;
    r.ngang = G / NM;
    if (G % (8u * NM) == 0) {
        const uint32_t xg = b & 7u, li = b >> 3;
        r.gang = xg * (G / 8u / NM) + li / NM;
        r.member = li % NM;
    } else {   // odd grid (host keeps G a multiple of NM): neighbours in launch order
        r.gang = b / NM;
        r.member = b % NM;
    }
    return r;
}

struct ColfdParams {
    const cf* in;      // the caller's (N, S) block, sample-major
    cf* q4;            // workspace, Q4 order
    int S, N2;         // S % 4 == 0
    BigTwiddle tw;     // W_N
    const cf* tw16k;
};

// Forward column pass: M-point FFT over n1 of x[(N2 n1 + n2) S + s], * W_N^{n2 k1}, stored in Q4 order.
// Structure = k_colq with a static, gang-scheduled tile order.  SP = where the memory instructions go (the pass is bound by the
// CU's address path: 11.7 us of loads + 7.6 us of stores per tile against 8 us of arithmetic; profiles/r04_ab_colfd_pacing.txt):
//   2 (default) = the ticks carry the next tile's loads only, ONE per tick (24 of the 32; the rest right after the transform), and
//       every output is stored as soon as its twiddle product exists: the stores drain while the next transform starts and its
//       loads trickle in behind them -- 1.18 ms at config 2.  Two loads per tick: 1.34; loads from tick 4 on or the last eight
//       between the stores: 1.22; every second tick: 1.18; no loads inside the transform: 1.27
//   0 = k_colq's pacing: stores deferred into the next transform, two (store, load) pairs per tick -- 1.34 ms
//   1 = one deferred pair per tick, the last eight between the twiddle products -- 1.38 ms
// TWR: 1 (default) = the inter-pass twiddle's base and step of the NEXT tile by recurrence -- its column is a fixed distance
//   (gangs x C) from this tile's, so base' = base x W_N^{dn2 tau}, step' = step x W_N^{dn2 MR} (float64, at most N2 / C / gangs
//   products in a row: 1e-14) -- instead of four table reads per tile, two of which are 16-line gathers on the address path this
//   kernel is bound by: 1.186 -> 1.165 ms at config 2, 25 VGPRs fewer; 0 = the tables for every tile (k_colq's way, whose tile
//   order is not static)
// STAUX: cache policy of the Q4 stores: 2 = nt (default: the lines are not read again before 2 GiB of other traffic have passed;
//   1.157 -> 1.135 ms at config 2; the row pass's stores carry it too, the inverse column pass's and the last pass's do not: r04_ab_colfd_pacing.txt)
template <int M, int R, int SP = 2, int TWR = 1, int STAUX = 2>
__global__ __launch_bounds__(kTilePoints / R) void k_colfd(ColfdParams p) {
    constexpr int F = kTilePoints / M;   // columns of a tile: 4 series x C columns n2
    constexpr int C = F / 4;
    static_assert(F >= 16 && C % 4 == 0, "a tile writes whole Q4 lines (4 series x 4 columns)");
    constexpr int MR = M / R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const int ser = f & 3, c = f >> 2;   // lanes 4c .. 4c+3: the quad's series of column c = 32 contiguous input bytes
    const uint32_t NQ = (uint32_t)p.S / 4u;
    const Gang gg = gang_of(blockIdx.x, gridDim.x, NQ);
    const uint32_t ngrp = (uint32_t)(p.N2 / C);
    const uint32_t quad = gg.member;

    const int voff_in = ((tau * p.N2 + c) * p.S + ser) * (int)sizeof(cf);
    const uint32_t step_in = (uint32_t)MR * (uint32_t)p.N2 * (uint32_t)p.S * (uint32_t)sizeof(cf);
    const uint32_t span_in = (uint32_t)((((int64_t)(M - 1) * p.N2 + (C - 1)) * p.S + 4) * (int64_t)sizeof(cf));
    const int voff_out = (tau * 4 * p.N2 + (c >> 2) * 16 + ser * 4 + (c & 3)) * (int)sizeof(cf);
    const uint32_t step_out = (uint32_t)MR * 4u * (uint32_t)p.N2 * (uint32_t)sizeof(cf);
    const uint32_t span_out = (uint32_t)(((int64_t)(M - 1) * 4 * p.N2 + (C / 4) * 16) * (int64_t)sizeof(cf));
    const int shift = p.tw.shift;
    const int64_t lomask = (1LL << shift) - 1;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    auto in_rsrc = [&](uint32_t g) {
        return g < ngrp ? make_rsrc(p.in + ((int64_t)g * C * p.S + 4 * quad), span_in) : make_rsrc(p.in, 0);
    };
    auto out_rsrc = [&](uint32_t g) {
        return make_rsrc(p.q4 + ((int64_t)quad * M * 4 * p.N2 + (int64_t)g * (C / 4) * 16), span_out);
    };
    auto load_tables = [&](int n2, double2& bh, double2& bl, double2& sh, double2& sl) {
        const int64_t pb = tw_reduce(p.tw, (int64_t)n2 * tau);
        const int64_t ps = tw_reduce(p.tw, (int64_t)n2 * MR);
        bh = p.tw.hi[pb >> shift];
        bl = p.tw.lo[pb & lomask];
        sh = p.tw.hi[ps >> shift];
        sl = p.tw.lo[ps & lomask];
    };

    uint32_t g = gg.gang;
    if (g >= ngrp) return;
    const rsrc_t rd0 = in_rsrc(g);
    rsrc_t rdo = make_rsrc(p.q4, 0);   // where the outputs waiting in `out` go (none yet)
    double2 zbh, zbl, zsh, zsl;
    load_tables((int)g * C + c, zbh, zbl, zsh, zsl);
    double2 zb_next = zmul(zbh, zbl), zs_next = zmul(zsh, zsl), dzb = make_double2(1.0, 0.0), dzs = make_double2(1.0, 0.0);
    if constexpr (TWR != 0) {
        const int64_t dn2 = (int64_t)gg.ngang * C;
        dzb = big_tw(p.tw, dn2 * tau);
        dzs = big_tw(p.tw, dn2 * MR);
    }
    cf v[R], out[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        v[i] = buf_load(rd0, voff_in, (int)(i * step_in));
        out[i] = make_cf(0, 0);
    }

    while (true) {
        launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
        const double2 zb = TWR ? zb_next : zmul(zbh, zbl), zs = TWR ? zs_next : zmul(zsh, zsl);
        if constexpr (TWR != 0) { zb_next = zmul(zb, dzb); zs_next = zmul(zs, dzs); }
        const uint32_t gn = g + gg.ngang;
        const bool more = gn < ngrp;
        const rsrc_t rd2 = in_rsrc(gn);            // tile whose samples are requested during this iteration
        cf in2[R];                                 // ... into the registers the stored outputs leave
        int cnt = 0;
        auto pump = [&](int n) {                   // store n waiting outputs, request n samples
#pragma unroll
            for (int k = 0; k < n; ++k) {
                if (cnt < R) {
                    if constexpr (SP != 2) buf_store(rdo, voff_out, (int)(cnt * step_out), out[cnt]);
                    in2[cnt] = buf_load(rd2, voff_in, (int)(cnt * step_in));
                    ++cnt;
                }
            }
        };
        auto hk = [&](auto st, auto q) {
            if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                __builtin_amdgcn_sched_barrier(0x38E);  // VALU / SALU / LDS may move across, global memory ops stay put
                pump(SP == 0 ? 2 : 1);
                __builtin_amdgcn_sched_barrier(0x38E);
            }
        };
        fft_tile<M, 1, R, -1, F, false, false, false>(v, lds, tau, f, w, hk);
        if constexpr (SP == 0 || SP == 2) pump(R);  // whatever the ticks did not reach (short transforms)
        else {
            constexpr int LEFT = 8;       // pairs kept for the twiddle products below (out[i] is stored before it is rewritten)
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (cnt < R - LEFT) pump(1);
        }
        // tables of the next tile: after every request above
        if (more && SP != 1 && TWR == 0) load_tables((int)gn * C + c, zbh, zbl, zsh, zsl);
        if constexpr (SP == 2) rdo = out_rsrc(g);
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if constexpr (SP != 0) {
                if (SP == 1 && i % 3 == 0 && cnt < R) {
                    __builtin_amdgcn_sched_barrier(0x38E);
                    pump(1);
                    __builtin_amdgcn_sched_barrier(0x38E);
                }
            }
            out[i] = cmul(v[i], make_cf((real)z.x, (real)z.y));
            if constexpr (SP == 2) buf_store<STAUX>(rdo, voff_out, (int)(i * step_out), out[i]);
            z = zmul(z, zs);
        }
        if constexpr (SP == 1) {
            pump(R);
            if (more) load_tables((int)gn * C + c, zbh, zbl, zsh, zsl);
        }
        rdo = out_rsrc(g);
        if (!more) break;
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = in2[i];
        g = gn;
    }
    if constexpr (SP != 2) {
#pragma unroll
        for (int i = 0; i < R; ++i) buf_store(rdo, voff_out, (int)(i * step_out), out[i]);
    }
}

// Fused row pass over Q4 rows: k_rowp16 (kernels.hpp) with one SERIES per tile instead of a polarisation pair, because the
// four series of a Q4 line must be read at the same time: gang of 4, member = series within the quad.  The two
// polarisations of a channel load the same phase row at the same time (one of them from L2).  Loads: a series' bins
// (2 tau, 2 tau + 1) + 1024 j are 16 bytes of a 32-byte piece; stores: planar rows of `out` (16 bytes per lane).
struct RowqParams {
    const cf* q4;        // Q4 rows: (quad, k1) at (quad N1 + k1) * 4 M
    cf* out;             // planar rows: series s, row k1 at (s N1 + k1) * M
    const float* phase;  // [chan][k1] rows in k_rowp16's order (ChirpParams::phase16), revolutions
    const cf* tw16k;
    int S, N1, npol;
    real scale;          // 1/N
};

// The pass is bound by the CU's address path: a tile's 16 loads of 32-byte pieces keep it busy ~11 us (32 lines per
// instruction), its 16 stores ~4 us, against ~11 us of butterflies.  With k_rowp16's pacing (all of the next tile's loads
// in the first ticks of the INVERSE transform) the next forward transform waits for them with the VALU idle: 18.4 us per
// tile.  FWD_LOADS of the next tile's 16 loads therefore ride in the FORWARD transform (registers: while the phase row is
// live only half a tile fits next to the tile in work), the rest in the inverse one, both spread over the ticks.
//   FWD_LOADS = 0: k_rowp16's pacing, 1.16-1.18 ms at config 2; 4: 1.11-1.15; 8: 1.08-1.10; 10 (default, 256 VGPRs): 1.04-1.08; 12: 1.11.
//   Issuing the ten EARLIER (every 2nd / 3rd tick instead of every 4th): 1.19 / 1.13; the phase row requested inside the forward
//   transform instead of before it: no change (profiles/r04_ab_row_pacing2.txt).
// STAUX: cache policy of the stores: 2 = nt (default) -- this pass's own time does not move, the inverse column pass that reads the
//   rows next does (0.7255 -> 0.7137 ms at config 2, profiles/r04_ab_colfd_pacing.txt)
template <int R, int FWD_LOADS = 10, int STAUX = 2>
__global__ __launch_bounds__(kTilePoints / R) void k_rowq16(RowqParams p) {
    constexpr int M = kTilePoints;
    static_assert(M == 16384 && R == 32, "k_rowq16: 2^14-point rows, 32 points per thread");
    constexpr int MR = M / R;
    constexpr int STEP16 = (M / 16) * (int)sizeof(cf);          // planar: 1024 elements between a thread's 16-byte pairs
    constexpr int STEPQ = 4 * STEP16;                           // Q4: 1024 bins = 256 lines
    constexpr int NSF = tw_seeds_or1(M, R, 1, false), NSI = tw_seeds_or1(M, R, 0, true);
    constexpr int NL = R / 2;                                   // 16-byte loads per tile and thread
    constexpr int TICKS = 40;                                   // ticks of one transform (16, 32, 32: 2 x 8 + 12 + 12)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tau = threadIdx.x;
    const Gang gg = gang_of(blockIdx.x, gridDim.x, 4u);
    const int ser = (int)gg.member;
    const int voffq = (tau >> 1) * 128 + ser * 32 + (tau & 1) * 16;
    const int voff16 = tau * 2 * (int)sizeof(cf);
    const uint32_t nunit = (uint32_t)(p.S / 4) * (uint32_t)p.N1;   // (quad, k1)

    cf wf[NSF], wi[NSI];
    load_tw_seeds<M, 1, R, 1, false>(wf, tau, p.tw16k);
    load_tw_seeds<M, 1, R, 0, true>(wi, tau, p.tw16k);

    auto row_rsrc = [&](uint32_t u) {
        return u < nunit ? make_rsrc(p.q4 + (int64_t)u * 4 * M, (uint32_t)(4 * M * sizeof(cf))) : make_rsrc(p.q4, 0);
    };
    auto series_of = [&](uint32_t u) -> int { return (int)(u / (uint32_t)p.N1) * 4 + ser; };
    auto out_rsrc = [&](uint32_t u) {
        const uint32_t k1 = u % (uint32_t)p.N1;
        return make_rsrc(p.out + ((int64_t)series_of(u) * p.N1 + k1) * M, (uint32_t)(M * sizeof(cf)));
    };
    auto load_pair = [&](rsrc_t r, int j, cf& a, cf& b) {   // bins (2 tau, 2 tau + 1) + 1024 j of this block's series
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voffq, j * STEPQ, 0);
        a = make_cf(__uint_as_float(x.x), __uint_as_float(x.y));
        b = make_cf(__uint_as_float(x.z), __uint_as_float(x.w));
    };
    auto store_pair = [&](rsrc_t r, int j, cf a, cf b) { buf_store_pair<STAUX>(r, voff16, j * STEP16, a, b); };

    uint32_t u = gg.gang;
    if (u >= nunit) return;
    cf v[R];
    {
        const rsrc_t rd = row_rsrc(u);
#pragma unroll
        for (int j = 0; j < NL; ++j) load_pair(rd, j, v[2 * j], v[2 * j + 1]);
    }
    // the first tile's samples are waited for HERE (k_rowp16: a wait at the loop header would merge with the back edge's)
    __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));   // vmcnt(0)
    launder_all(v, std::make_integer_sequence<int, R>{});
    launder_all(wf, std::make_integer_sequence<int, NSF>{});
    launder_all(wi, std::make_integer_sequence<int, NSI>{});
    float ph[R];

    while (true) {
        launder_all(wf, std::make_integer_sequence<int, NSF>{});
        launder_all(wi, std::make_integer_sequence<int, NSI>{});
        launder_all(v, std::make_integer_sequence<int, R>{});
        const uint32_t qd = u / (uint32_t)p.N1, k1 = u - qd * (uint32_t)p.N1;
        const uint32_t chan = (qd * 4u + (uint32_t)ser) / (uint32_t)p.npol;
        const rsrc_t rp = make_rsrc(p.phase + ((int64_t)chan * p.N1 + k1) * M, (uint32_t)(M * sizeof(float)));
#pragma unroll
        for (int j = 0; j < R / 4; ++j) {   // the row is stored in this order (ChirpParams::phase16): 8 x 16 bytes per thread
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (4 * MR * (int)sizeof(float)), 0);
            ph[4 * j] = __uint_as_float(x.x);
            ph[4 * j + 1] = __uint_as_float(x.y);
            ph[4 * j + 2] = __uint_as_float(x.z);
            ph[4 * j + 3] = __uint_as_float(x.w);
        }
        const uint32_t un = u + gg.ngang;
        const bool more = un < nunit;
        const rsrc_t rdn = row_rsrc(un);
        const rsrc_t ws = out_rsrc(u);
        cf nx[R];
        int cnt = 0;
        // forward: 16 (pair-adjacent bases), 32, 32 -> natural distribution tau + 512 i
        if constexpr (FWD_LOADS > 0) {
            int tk = 0;
            auto hkf = [&](auto st, auto q) {
                if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                    constexpr int EVERY = TICKS / FWD_LOADS;
                    if (tk % EVERY == EVERY - 1 && cnt < FWD_LOADS) {
                        __builtin_amdgcn_sched_barrier(0x38E);
                        load_pair(rdn, cnt, nx[2 * cnt], nx[2 * cnt + 1]);
                        ++cnt;
                        __builtin_amdgcn_sched_barrier(0x38E);
                    }
                    ++tk;
                }
            };
            fft_tile<M, 1, R, -1, 1, true, false, false, decltype(hkf), 1, true, false>(v, lds, tau, 0, wf, hkf);
        } else {
            fft_tile<M, 1, R, -1, 1, true, false, false, NoHook, 1, true, false>(v, lds, tau, 0, wf);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const cf c = make_cf(__builtin_amdgcn_cosf(ph[i]) * p.scale, __builtin_amdgcn_sinf(ph[i]) * p.scale);
            v[i] = cmul(v[i], c);
        }
        int tki = 0;
        auto hk = [&](auto st, auto q) {
            if constexpr (std::is_same<decltype(q), tick_tag>::value) {   // the rest of the next tile's loads
                constexpr int LEFT = NL - FWD_LOADS;
                constexpr int EVERY = FWD_LOADS > 0 ? TICKS / (LEFT > 0 ? LEFT : 1) : 1;
                if (tki % EVERY == 0 && cnt < NL) {
                    __builtin_amdgcn_sched_barrier(0x38E);
                    load_pair(rdn, cnt, nx[2 * cnt], nx[2 * cnt + 1]);
                    ++cnt;
                    __builtin_amdgcn_sched_barrier(0x38E);
                }
                ++tki;
            }
        };
        // inverse: 32, 32, 16 ending on pair-adjacent bases: v[q + 2 u] = X[2 tau + q + 1024 u]
        fft_tile<M, 1, R, +1, 1, true, false, false, decltype(hk), 0, false, true>(v, lds, tau, 0, wi, hk);
#pragma unroll
        for (int j = 0; j < NL; ++j) store_pair(ws, j, v[2 * j], v[2 * j + 1]);
#pragma unroll
        for (int k = 0; k < NL; ++k)
            if (cnt < NL) { load_pair(rdn, cnt, nx[2 * cnt], nx[2 * cnt + 1]); ++cnt; }
        if (!more) break;
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = nx[i];
        u = un;
    }
}

#endif  // !PBH_F64
}  // namespace PBH_NS
