// conv_igemm.hip -- forward and input-gradient 3-D convolution for gfx950 (MI355X).
//
// Supplies the arithmetic of aten::conv3d and its dgrad for every Conv3d call site of
// the reference (resnet.py:23-30,40-52,63-70,170,181,184,270; network.py:102-117).
//
// Design (MI355X-first, not a port of anything):
//   * Implicit GEMM  C[m][p] = sum_k A[m][k] * B[k][p]  with the channel being produced on
//     the MFMA row axis and output voxels p = (n, t, h, w) on the column axis, so a
//     16-lane group of the accumulator maps to 16 consecutive voxels (64-B store
//     segments along W, the contiguous NCDHW axis).
//   * The contraction runs on the fp32-input matrix core, v_mfma_f32_16x16x4_f32: it is
//     bit-exact fp32 (one rounding per product, an fmaf chain in k order) and issues at
//     the same 64 FLOP/clk/SIMD as the fp32 vector pipe, but needs one VGPR per operand
//     and leaves the VALU free for the gather address arithmetic.
//   * B is never materialised in HBM: each K-chunk's (BK x BN) im2col slab is gathered
//     straight from the NCDHW tensor into LDS.  A wave owns whole k-rows of the slab, so
//     (channel, tap) decoding is per row (done once per chunk by BK lanes, broadcast
//     through LDS) and each lane only adds a row offset to its voxel's base offset and
//     tests three bits of a per-voxel padding mask.  Lanes walk W, so the global loads
//     are coalesced along W as far as stride allows.
//   * Register-staged double buffering: global loads for chunk i+1 are issued before the
//     MFMAs of chunk i and written to the other LDS buffer afterwards; one barrier per
//     chunk.
//   * The same kernel computes dgrad: rows are input channels, columns are input voxels,
//     and the gather reads dy at (t + pT - kt)/sT when divisible (per-voxel masks hold the
//     divisibility + range test per tap index), weights are addressed transposed.
//
// LDS images: As[BK][LDA], Bs[BK][LDB] with LD == 16 (mod 32) so that the two k-rows a
// 32-lane group touches in one ds_read_b32 fall on disjoint bank halves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zsv_hip.h"
#include "zsv_common.h"

namespace zsv {

struct IgemmParams {
    int M;          // rows: channels produced (Cout fwd / Cin dgrad)
    int P;          // columns: N * oS voxels of the produced tensor
    int K;          // reduction: Cred * taps
    int taps;       // kT*kH*kW
    int kHW, kW;    // for tap -> (kt, kh, kw)
    int oS, oHW, oW;    // produced tensor: voxels per clip, H*W, W
    int gC;             // channels of the gathered tensor
    int gT, gH, gW;     // gathered tensor spatial dims
    int gS, gHW;        // gathered tensor: voxels per channel, H*W
    int sT, sH, sW;
    int pT, pH, pW;
    int kT, kH;
    int a_col_stride;   // A(m, k) = a[m * a_col_stride + a_rowoff(k)]
    int relu;
};

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

template <int X> struct LdPad { static constexpr int value = (X % 32 == 16) ? X : X + 16; };

// ---------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN, int MODE, bool AVEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(IgemmParams prm,
                                                         const float* __restrict__ A,
                                                         const float* __restrict__ G,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ C, int tiles_m) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    constexpr int BK = 16;
    constexpr int LDA = LdPad<BM>::value;
    constexpr int LDB = LdPad<BN>::value;
    constexpr int NT = 256;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    static_assert(BN == 64 || BN == 128 || BN == 256, "BN must divide the workgroup");
    constexpr int BROWS = NT / BN;              // k-rows covered per staging pass
    constexpr int BPASS = BK / BROWS;           // passes per chunk
    constexpr int APASS = (BM * BK + NT - 1) / NT;          // scalar A staging
    constexpr int AVPASS = (BM * (BK / 4) + NT - 1) / NT;   // float4 A staging

    __shared__ float As[2][BK * LDA];
    __shared__ float Bs[2][BK * LDB];
    __shared__ int rinfo[2][BK][4];   // {gather offset, packed shifts, A row offset, -}

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);

    const int tile = blockIdx.x;
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    // ---- per-thread gather column: voxel -> base offset + padding mask -------------
    const int bcol = tid % BN;
    const int brow0 = __builtin_amdgcn_readfirstlane(tid / BN);
    int base_off = 0;
    unsigned vmask = 0;
    {
        const int p = n0 + bcol;
        if (p < prm.P) {
            const int n = p / prm.oS;
            int r = p - n * prm.oS;
            const int ot = r / prm.oHW;
            r -= ot * prm.oHW;
            const int oh = r / prm.oW;
            const int ow = r - oh * prm.oW;
            if (MODE == MODE_FWD) {
                const int t0 = ot * prm.sT - prm.pT, h0 = oh * prm.sH - prm.pH, w0 = ow * prm.sW - prm.pW;
                base_off = n * prm.gC * prm.gS + t0 * prm.gHW + h0 * prm.gW + w0;
                for (int k = 0; k < prm.kW; ++k) vmask |= ((unsigned)(w0 + k) < (unsigned)prm.gW) << k;
                for (int k = 0; k < prm.kH; ++k) vmask |= ((unsigned)(h0 + k) < (unsigned)prm.gH) << (8 + k);
                for (int k = 0; k < prm.kT; ++k) vmask |= ((unsigned)(t0 + k) < (unsigned)prm.gT) << (16 + k);
            } else {
                const int tb = ot + prm.pT, hb = oh + prm.pH, wb = ow + prm.pW;
                const int tq = tb / prm.sT, hq = hb / prm.sH, wq = wb / prm.sW;
                const int tr = tb - tq * prm.sT, hr = hb - hq * prm.sH, wr = wb - wq * prm.sW;
                base_off = n * prm.gC * prm.gS + tq * prm.gHW + hq * prm.gW + wq;
                for (int k = 0; k < prm.kW; ++k)
                    vmask |= ((k % prm.sW == wr) && (unsigned)(wq - k / prm.sW) < (unsigned)prm.gW) << k;
                for (int k = 0; k < prm.kH; ++k)
                    vmask |= ((k % prm.sH == hr) && (unsigned)(hq - k / prm.sH) < (unsigned)prm.gH) << (8 + k);
                for (int k = 0; k < prm.kT; ++k)
                    vmask |= ((k % prm.sT == tr) && (unsigned)(tq - k / prm.sT) < (unsigned)prm.gT) << (16 + k);
            }
        }
    }

    // ---- row decode: k -> gather offset / mask shifts / A offset (BK lanes per chunk) --
    auto decode_rows = [&](int k0, int buf) {
        if (tid < BK) {
            const int k = k0 + tid;
            int goff = 0, shifts = 31 | (31 << 8) | (31 << 16), aoff = -1;
            if (k < prm.K) {
                const int c = k / prm.taps;
                const int tap = k - c * prm.taps;
                const int kt = tap / prm.kHW;
                const int r = tap - kt * prm.kHW;
                const int kh = r / prm.kW;
                const int kw = r - kh * prm.kW;
                shifts = kw | ((8 + kh) << 8) | ((16 + kt) << 16);
                if (MODE == MODE_FWD) {
                    goff = c * prm.gS + kt * prm.gHW + kh * prm.gW + kw;
                    aoff = k;
                } else {
                    goff = c * prm.gS - (kt / prm.sT) * prm.gHW - (kh / prm.sH) * prm.gW - (kw / prm.sW);
                    aoff = c * (prm.M * prm.taps) + tap;
                }
            }
            rinfo[buf][tid][0] = goff;
            rinfo[buf][tid][1] = shifts;
            rinfo[buf][tid][2] = aoff;
        }
    };

    float breg[BPASS];
    float areg[AVEC ? AVPASS * 4 : APASS];

    auto load_chunk = [&](int k0, int buf) {
        // B: gathered slab rows brow0 + BROWS*j, column bcol
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const int row = brow0 + BROWS * j;
            const int goff = rinfo[buf][row][0];
            const int sh = rinfo[buf][row][1];
            const unsigned ok = (vmask >> (sh & 31)) & (vmask >> ((sh >> 8) & 31)) & (vmask >> ((sh >> 16) & 31)) & 1u;
            const int off = ok ? base_off + goff : 0;
            const float v = G[off];
            breg[j] = ok ? v : 0.f;
        }
        if (AVEC) {
#pragma unroll
            for (int j = 0; j < AVPASS; ++j) {
                const int e = tid + NT * j;
                const int m = e % BM, kq = e / BM;
                const bool ok = (e < BM * (BK / 4)) && (m0 + m < prm.M) && (k0 + 4 * kq < prm.K);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4*>(A + (size_t)(m0 + m) * prm.K + k0 + 4 * kq);
                areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                const int e = tid + NT * j;
                const int row = e / BM, m = e % BM;
                float v = 0.f;
                if (e < BM * BK) {
                    const int aoff = rinfo[buf][row][2];
                    const bool ok = (m0 + m < prm.M) && (aoff >= 0);
                    const int off = ok ? (m0 + m) * prm.a_col_stride + aoff : 0;
                    v = A[off];
                    v = ok ? v : 0.f;
                }
                areg[j] = v;
            }
        }
    };

    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BPASS; ++j) Bs[buf][(brow0 + BROWS * j) * LDB + bcol] = breg[j];
        if (AVEC) {
#pragma unroll
            for (int j = 0; j < AVPASS; ++j) {
                const int e = tid + NT * j;
                const int m = e % BM, kq = e / BM;
                if (e < BM * (BK / 4)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) As[buf][(4 * kq + r) * LDA + m] = areg[4 * j + r];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                const int e = tid + NT * j;
                if (e < BM * BK) As[buf][(e / BM) * LDA + (e % BM)] = areg[j];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (prm.K + BK - 1) / BK;

    // prologue: rows of chunk 0 -> load chunk 0 -> rows of chunk 1 -> LDS
    decode_rows(0, 0);
    __syncthreads();
    load_chunk(0, 0);
    decode_rows(BK, 1);
    store_chunk(0);
    __syncthreads();

    const int frag_row = lane >> 4;      // k within a 4-deep MFMA step
    const int frag_col = lane & 15;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = (ch + 1) < nchunks;
        if (more) {
            load_chunk((ch + 1) * BK, cur ^ 1);      // uses rinfo[cur^1] (decoded last iteration)
        }
        // rinfo[cur] was consumed by the loads issued in the previous iteration
        // (and nobody reads it again before the barrier below)
        if (ch + 2 < nchunks) decode_rows((ch + 2) * BK, cur);

        const float* as = &As[cur][0];
        const float* bs = &Bs[cur][0];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(kk * 4 + frag_row) * LDA + wm0 + 16 * i + frag_col];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(kk * 4 + frag_row) * LDB + wn0 + 16 * j + frag_col];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc(i,j)[r] -> C[n][m][sp]; rows m = .. + 4*(lane>>4) + r, col = lane&15
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = n0 + wn0 + 16 * j + frag_col;
        if (p >= prm.P) continue;
        const int n = p / prm.oS;
        const int sp = p - n * prm.oS;
        float* cbase = C + (size_t)n * prm.M * prm.oS + sp;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + 16 * i + 4 * frag_row + r;
                if (m < prm.M) {
                    float v = acc[i][j][r];
                    if (bias != nullptr) v += bias[m];
                    if (prm.relu) v = fmaxf(v, 0.f);
                    cbase[(size_t)m * prm.oS] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN, int MODE, bool AVEC>
static int launch_cfg(const IgemmParams& prm, const float* A, const float* G, const float* bias,
                      float* C, hipStream_t stream) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    const int tiles_m = (prm.M + BM - 1) / BM;
    const long tiles_n = ((long)prm.P + BN - 1) / BN;
    const long blocks = tiles_m * tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffL) return ZSV_E_TOO_LARGE;
    hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WGM, WGN, MODE, AVEC>), dim3((unsigned)blocks),
                       dim3(256), 0, stream, prm, A, G, bias, C, tiles_m);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

// padded work for a BM x BN tiling
static inline double padded_work(int M, long P, int BM, int BN) {
    const double tm = (M + BM - 1) / BM, tn = (double)((P + BN - 1) / BN);
    return tm * BM * tn * BN;
}

template <int MODE, bool AVEC>
static int dispatch(const IgemmParams& prm, const float* A, const float* G, const float* bias,
                    float* C, hipStream_t stream) {
    // candidate row tilings (BM): 48, 64, 80, 128, 144; pick the least padded work,
    // preferring the larger tile on ties (more reuse of the gathered slab).
    struct Cand { int bm, bn; };
    const Cand cands[] = {{144, 128}, {128, 128}, {80, 128}, {64, 128}, {48, 256}};
    int best = 0;
    double best_w = 1e300;
    for (int i = 0; i < 5; ++i) {
        double w = padded_work(prm.M, prm.P, cands[i].bm, cands[i].bn);
        // few-tile launches underfill 256 CUs: favour smaller tiles then
        const double blocks = ((prm.M + cands[i].bm - 1) / cands[i].bm) * (double)(((long)prm.P + cands[i].bn - 1) / cands[i].bn);
        if (blocks < 512) w *= (512.0 / (blocks < 1 ? 1 : blocks)) > 4.0 ? 4.0 : (512.0 / blocks);
        if (w < best_w * 0.999) { best_w = w; best = i; }
    }
    switch (best) {
        case 0: return launch_cfg<9, 2, 1, 4, MODE, AVEC>(prm, A, G, bias, C, stream);
        case 1: return launch_cfg<4, 4, 2, 2, MODE, AVEC>(prm, A, G, bias, C, stream);
        case 2: return launch_cfg<5, 2, 1, 4, MODE, AVEC>(prm, A, G, bias, C, stream);
        case 3: return launch_cfg<4, 2, 1, 4, MODE, AVEC>(prm, A, G, bias, C, stream);
        default: return launch_cfg<3, 4, 1, 4, MODE, AVEC>(prm, A, G, bias, C, stream);
    }
}

int conv_check(const zsv_conv_desc* d) {
    if (!d) return ZSV_E_NULL;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0) return ZSV_E_BAD_SHAPE;
    if (d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->kT > 7 || d->kH > 7 || d->kW > 7) return ZSV_E_BAD_SHAPE;
    if (d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0) return ZSV_E_BAD_SHAPE;
    if (d->To != (d->Ti + 2 * d->pT - d->kT) / d->sT + 1) return ZSV_E_BAD_SHAPE;
    if (d->Ho != (d->Hi + 2 * d->pH - d->kH) / d->sH + 1) return ZSV_E_BAD_SHAPE;
    if (d->Wo != (d->Wi + 2 * d->pW - d->kW) / d->sW + 1) return ZSV_E_BAD_SHAPE;
    if (d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return ZSV_E_BAD_SHAPE;
    const double lim = 2147483647.0;
    const double xin = (double)d->N * d->Cin * d->Ti * d->Hi * d->Wi;
    const double yout = (double)d->N * d->Cout * d->To * d->Ho * d->Wo;
    const double wn = (double)d->Cout * d->Cin * d->kT * d->kH * d->kW;
    if (xin >= lim || yout >= lim || wn >= lim) return ZSV_E_TOO_LARGE;
    return ZSV_OK;
}

static void fill_common(IgemmParams& p, const zsv_conv_desc* d) {
    p.taps = d->kT * d->kH * d->kW;
    p.kHW = d->kH * d->kW;
    p.kW = d->kW;
    p.kH = d->kH;
    p.kT = d->kT;
    p.sT = d->sT; p.sH = d->sH; p.sW = d->sW;
    p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
    p.relu = 0;
}

}  // namespace zsv

using namespace zsv;

extern "C" int zsv_conv3d_fwd(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                              float* y, int fuse_relu, void* stream) {
    int st = conv_check(d);
    if (st) return st;
    if (!x || !w || !y) return ZSV_E_NULL;
    IgemmParams p;
    fill_common(p, d);
    p.M = d->Cout;
    p.P = d->N * d->To * d->Ho * d->Wo;
    p.K = d->Cin * p.taps;
    p.oS = d->To * d->Ho * d->Wo; p.oHW = d->Ho * d->Wo; p.oW = d->Wo;
    p.gC = d->Cin; p.gT = d->Ti; p.gH = d->Hi; p.gW = d->Wi;
    p.gS = d->Ti * d->Hi * d->Wi; p.gHW = d->Hi * d->Wi;
    p.a_col_stride = p.K;
    p.relu = fuse_relu ? 1 : 0;
    const bool avec = (p.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    if (avec) return dispatch<MODE_FWD, true>(p, w, x, bias, y, (hipStream_t)stream);
    return dispatch<MODE_FWD, false>(p, w, x, bias, y, (hipStream_t)stream);
}

extern "C" int zsv_conv3d_dgrad(const zsv_conv_desc* d, const float* dy, const float* w, float* dx,
                                void* stream) {
    int st = conv_check(d);
    if (st) return st;
    if (!dy || !w || !dx) return ZSV_E_NULL;
    IgemmParams p;
    fill_common(p, d);
    p.M = d->Cin;
    p.P = d->N * d->Ti * d->Hi * d->Wi;
    p.K = d->Cout * p.taps;
    p.oS = d->Ti * d->Hi * d->Wi; p.oHW = d->Hi * d->Wi; p.oW = d->Wi;
    p.gC = d->Cout; p.gT = d->To; p.gH = d->Ho; p.gW = d->Wo;
    p.gS = d->To * d->Ho * d->Wo; p.gHW = d->Ho * d->Wo;
    p.a_col_stride = p.taps;
    return dispatch<MODE_DGRAD, false>(p, w, dy, nullptr, dx, (hipStream_t)stream);
}

extern "C" int zsv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t rows,
                              int32_t in_features, int32_t out_features, int fuse_relu, void* stream) {
    zsv_conv_desc d = {rows, in_features, 1, 1, 1, out_features, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    return zsv_conv3d_fwd(&d, x, w, bias, y, fuse_relu, stream);
}

extern "C" int zsv_linear_dgrad(const float* dy, const float* w, float* dx, int32_t rows,
                                int32_t in_features, int32_t out_features, void* stream) {
    zsv_conv_desc d = {rows, in_features, 1, 1, 1, out_features, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    return zsv_conv3d_dgrad(&d, dy, w, dx, stream);
}
