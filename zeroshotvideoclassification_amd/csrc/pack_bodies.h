// pack_bodies.h -- the element formulas of the weight-panel layouts, shared by the per-call pack kernels and by the multi-job
// pack kernel (pack_multi.hip): one definition, so a panel packed ahead of the call is the panel the call would have packed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zsv_hip.h"
#include "conv_params.h"

namespace zsv {

// ---- kind 0: direct kernel (conv_tap.hip): Wp[((cb * taps + tap) * 16 + c % 16) * Mp + m] --------------------------------
struct PackTapArgs {
    int M, gC, taps, nHW, nW, k0T, k0H, k0W, tsT, tsH, tsW, kH, kW, t2_cin, dir, w_m_stride, w_c_stride, Cpad, Mp;
};
__host__ __device__ inline PackTapArgs pack_tap_args(const IgemmParams& prm, int w_m_stride, int w_c_stride, int Cpad, int Mp) {
    PackTapArgs a = {prm.M, prm.gC, prm.taps, prm.nHW, prm.nW, prm.k0T, prm.k0H, prm.k0W, prm.tsT, prm.tsH, prm.tsW, prm.kH, prm.kW,
                     prm.t2_cin, prm.dir, w_m_stride, w_c_stride, Cpad, Mp};
    return a;
}
__device__ __forceinline__ float pack_tap_value(const PackTapArgs& a, const float* __restrict__ W, long i) {
    // row = ((channel block * taps) + tap) * 16 + channel-in-block: all taps of a 16-channel block are consecutive chunks
    const int m = (int)(i % a.Mp);
    const long rc = i / a.Mp;
    const int blk = (int)(rc / 16);
    const int cb = blk / a.taps;
    const int tap = blk - cb * a.taps;
    const int c = cb * 16 + (int)(rc % 16);
    if (!(m < a.M && c < a.gC && cb * 16 < a.Cpad)) return 0.f;
    if (a.t2_cin) return W[a.dir > 0 ? t2_weight_offset(m, c, a.t2_cin) : t2_weight_offset(c, m, a.t2_cin)];
    const int jt = tap / a.nHW;
    const int r = tap - jt * a.nHW;
    const int jh = r / a.nW;
    const int jw = r - jh * a.nW;
    const int tap_full = ((a.k0T + a.tsT * jt) * a.kH + a.k0H + a.tsH * jh) * a.kW + a.k0W + a.tsW * jw;
    return W[(size_t)m * a.w_m_stride + (size_t)c * a.w_c_stride + tap_full];
}

// ---- kinds 1 / 2: Winograd F(2,3) / F(4,3) transformed weights (conv_wino.hip): Up[(cb*R + r)*NP + pt][Mp][c%16] -------
struct PackWinoArgs { int M, Mp, C, nblk, R, flip; long sm, sc; };
template <int NP>
__device__ __forceinline__ float pack_wino_value(const PackWinoArgs& a, const float* __restrict__ W, long i) {
    const int c16 = (int)(i % 16);
    long r = i / 16;
    const int m = (int)(r % a.Mp);
    r /= a.Mp;
    const int pt = (int)(r % NP);
    r /= NP;
    const int kh = (int)(r % a.R);
    const int cb = (int)(r / a.R);
    const int c = cb * 16 + c16;
    if (!(m < a.M && c < a.C)) return 0.f;
    const float* g = W + (size_t)m * a.sm + (size_t)c * a.sc;
    const int last = 3 * a.R - 1;
    const int k0 = a.flip ? last - (3 * kh + 0) : 3 * kh + 0, k1 = a.flip ? last - (3 * kh + 1) : 3 * kh + 1,
              k2 = a.flip ? last - (3 * kh + 2) : 3 * kh + 2;
    if constexpr (NP == 4) {
        const float g0 = g[k0], g1 = g[k1], g2 = g[k2];
        return pt == 0 ? g0 : pt == 1 ? 0.5f * ((g0 + g2) + g1) : pt == 2 ? 0.5f * ((g0 + g2) - g1) : g2;
    } else {
        const double g0 = g[k0], g1 = g[k1], g2 = g[k2];          // (formed in double, rounded once)
        const double u = pt == 0 ? g0 / 4 : pt == 1 ? -((g0 + g2) + g1) / 6 : pt == 2 ? -((g0 + g2) - g1) / 6
                       : pt == 3 ? (g0 / 24 + g2 / 6) + g1 / 12 : pt == 4 ? (g0 / 24 + g2 / 6) - g1 / 12 : g2;
        return (float)u;
    }
}

// all NP points of one (channel-in-block, row, row tap, channel block) triple: the three taps are read once (the per-element
// formula above fetches them once per point) and the NP values go to NP panels, 16 consecutive channels = 64 contiguous bytes each.
// `t` indexes the triples in the panel's own order with the point axis removed; values are pack_wino_value's, bit for bit.
template <int NP>
__device__ __forceinline__ void pack_wino_triple(const PackWinoArgs& a, const float* __restrict__ W, float* __restrict__ out, long t) {
    const int c16 = (int)(t % 16);
    long r = t / 16;
    const int m = (int)(r % a.Mp);
    r /= a.Mp;
    const int kh = (int)(r % a.R);
    const int cb = (int)(r / a.R);
    const int c = cb * 16 + c16;
    float* o = out + (((size_t)cb * a.R + kh) * NP * a.Mp + m) * 16 + c16;           // point 0; point pt is pt * Mp * 16 further
    const size_t pstride = (size_t)a.Mp * 16;
    if (!(m < a.M && c < a.C)) {
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) o[pt * pstride] = 0.f;
        return;
    }
    const float* g = W + (size_t)m * a.sm + (size_t)c * a.sc;
    const int last = 3 * a.R - 1;
    const int k0 = a.flip ? last - (3 * kh + 0) : 3 * kh + 0, k1 = a.flip ? last - (3 * kh + 1) : 3 * kh + 1,
              k2 = a.flip ? last - (3 * kh + 2) : 3 * kh + 2;
    if constexpr (NP == 4) {
        const float g0 = g[k0], g1 = g[k1], g2 = g[k2];
        o[0] = g0;
        o[pstride] = 0.5f * ((g0 + g2) + g1);
        o[2 * pstride] = 0.5f * ((g0 + g2) - g1);
        o[3 * pstride] = g2;
    } else {
        const double g0 = g[k0], g1 = g[k1], g2 = g[k2];
        o[0] = (float)(g0 / 4);
        o[pstride] = (float)(-((g0 + g2) + g1) / 6);
        o[2 * pstride] = (float)(-((g0 + g2) - g1) / 6);
        o[3 * pstride] = (float)((g0 / 24 + g2 / 6) + g1 / 12);
        o[4 * pstride] = (float)((g0 / 24 + g2 / 6) - g1 / 12);
        o[5 * pstride] = (float)g2;
    }
}

// ---- kind 3: stride-2 input gradient (conv_dgrad_s2.hip): Wp[((chunk * NTAP + tap) * 8 + co % 8) * Mp + m] = W[co][m][tap] ----
struct PackS2Args { int M, Mp, Cout, ntap; };
__device__ __forceinline__ float pack_s2_value(const PackS2Args& a, const float* __restrict__ W, long i) {
    const int m = (int)(i % a.Mp);
    long r = i / a.Mp;
    const int k = (int)(r % 8);
    r /= 8;
    const int tap = (int)(r % a.ntap);
    const int co = (int)(r / a.ntap) * 8 + k;
    return (m < a.M && co < a.Cout) ? W[((size_t)co * a.M + m) * a.ntap + tap] : 0.f;
}

// ---- a pack launch written down as a job of zsv_pack_multi -----------------------------------------------------------------
inline void pack_job_tap(zsv_pack_job& j, const PackTapArgs& a, const float* W, float* out, long total) {
    j = zsv_pack_job{};
    j.kind = 0; j.total = total; j.w = W; j.out = out;
    const int v[19] = {a.M, a.gC, a.taps, a.nHW, a.nW, a.k0T, a.k0H, a.k0W, a.tsT, a.tsH, a.tsW, a.kH, a.kW, a.t2_cin, a.dir,
                       a.w_m_stride, a.w_c_stride, a.Cpad, a.Mp};
    for (int k = 0; k < 19; ++k) j.i[k] = v[k];
}
inline void pack_job_wino(zsv_pack_job& j, int points, const PackWinoArgs& a, const float* W, float* out, long total) {
    j = zsv_pack_job{};
    j.kind = points == 4 ? 1 : 2; j.total = total; j.w = W; j.out = out;
    j.i[0] = a.M; j.i[1] = a.Mp; j.i[2] = a.C; j.i[3] = a.nblk; j.i[4] = a.R; j.i[5] = a.flip;
    j.l[0] = a.sm; j.l[1] = a.sc;
}
inline void pack_job_s2(zsv_pack_job& j, const PackS2Args& a, const float* W, float* out, long total) {
    j = zsv_pack_job{};
    j.kind = 3; j.total = total; j.w = W; j.out = out;
    j.i[0] = a.M; j.i[1] = a.Mp; j.i[2] = a.Cout; j.i[3] = a.ntap;
}

}  // namespace zsv
