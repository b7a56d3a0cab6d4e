// conv_igemm.hip -- forward and input-gradient 3-D convolution for gfx950 (MI355X).
//
// Supplies the arithmetic of aten::conv3d and its dgrad for every Conv3d call site of
// the reference (resnet.py:23-30,40-52,63-70,170,181,184,270; network.py:102-117).
//
// Design (MI355X-first, not a port of anything):
//   * Implicit GEMM  C[m][p] = sum_k A[m][k] * B[k][p]  with the channel being produced on
//     the MFMA row axis and voxels p = (n, t, h, w) on the column axis, so a 16-lane group
//     of the accumulator maps to 16 consecutive voxels (64-B store segments along W, the
//     contiguous NCDHW axis).
//   * The contraction runs on the fp32-input matrix core, v_mfma_f32_16x16x4_f32: bit-exact
//     fp32 (one rounding per product, an fmaf chain in k order) at the fp32 vector rate, one
//     VGPR per operand, VALU left free for the gather arithmetic.
//   * B is never materialised in HBM: each K-chunk's (BK x BN) im2col slab is gathered
//     straight from the NCDHW tensor into LDS.  A wave owns whole k-rows of the slab, so the
//     (channel, tap) decode is per row: BK lanes decode a chunk's rows one chunk ahead and
//     publish {byte offset, tap} through LDS; consumers pull them into SGPRs
//     (v_readfirstlane).  Per element a lane does: bit-extract its voxel's tap-validity
//     mask, add the row offset, OR in the out-of-range flag -- 4 VALU -- and issues a
//     `buffer_load_dword`: the buffer descriptor's range check returns 0 for padded taps, so
//     there is no branch and no select behind the load.  Lanes walk W: loads coalesce.
//   * Register-staged double buffering: loads for chunk i+1 are issued first, then all MFMA
//     fragments of chunk i are fetched from LDS in one burst, then the chunk's MFMAs run
//     back to back; the staged registers are written to the other LDS buffer afterwards
//     (that is where the vmcnt wait lands); one barrier per chunk.
//   * dgrad runs on the same kernel: it is a correlation of dy with the transposed kernel.
//     A strided convolution's dgrad is split into sT*sH*sW residue classes of input voxels;
//     each class is a dense stride-1 problem over its own taps (no multiplies by the zeros
//     a naive "insert zeros" transposed convolution would do: 4x fewer MFMAs for the 1x3x3
//     stride-2 layers, 8x for the 1x1x1 shortcuts).
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous range of
//     tiles, so tiles that share gathered input rows and the weight panel share an L2.
//
// LDS images: As[BK][LDA], Bs[BK][LDB] with LD == 16 (mod 32) so that the two k-rows a
// 32-lane group touches in one ds_read_b32 fall on disjoint bank halves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "conv_params.h"
#include "knobs.h"

namespace zsv {

// ---------------------------------------------------------------------------------
// AVEC: A rows are contiguous in k and 16-B aligned (forward, K % 4 == 0): float4 staging.
// WIDE: more than 31 taps (7x7 stems): per-axis validity masks instead of one tap mask.
template <int TM, int TN, int WGM, int WGN, bool AVEC, bool WIDE>
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
    constexpr unsigned OOB = 0xFFFFFFFFu;

    __shared__ float As[2][BK * LDA];
    __shared__ float Bs[2][BK * LDB];
    __shared__ int rinfo[2][BK][4];   // {gather byte offset, tap | packed shifts, A offset, -}

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);

    const int tile = xcd_tile(gridDim.x, blockIdx.x);
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, prm.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, prm.a_bytes, 0x00020000);

    // ---- per-thread gather column: voxel -> base byte offset + tap-validity mask ----------
    const int bcol = tid % BN;
    const int brow0 = sgpr(tid / BN);
    int base_bytes = 0;
    unsigned vmask = 0;          // !WIDE: bit tap = tap valid;  WIDE: w bits 0-6, h 8-14, t 16-22
    {
        const int p = n0 + bcol;
        if (p < prm.P) {
            const int n = p / prm.cS;
            int r = p - n * prm.cS;
            const int ct = r / prm.cHW;
            r -= ct * prm.cHW;
            const int ch = r / prm.cW;
            const int cw = r - ch * prm.cW;
            const int t0 = ct * prm.gsT + prm.goT, h0 = ch * prm.gsH + prm.goH, w0 = cw * prm.gsW + prm.goW;
            base_bytes = 4 * (n * prm.gC * prm.gS + t0 * prm.gHW + h0 * prm.gW + w0);
            unsigned mw = 0, mh = 0, mt = 0;
            for (int j = 0; j < prm.nW; ++j) mw |= ((unsigned)(w0 + prm.dir * j) < (unsigned)prm.gW) << j;
            for (int j = 0; j < prm.nH; ++j) mh |= ((unsigned)(h0 + prm.dir * j) < (unsigned)prm.gH) << j;
            for (int j = 0; j < prm.nT; ++j) mt |= ((unsigned)(t0 + prm.dir * j) < (unsigned)prm.gT) << j;
            if (WIDE) {
                vmask = mw | (mh << 8) | (mt << 16);
            } else {
                int tap = 0;
                for (int a = 0; a < prm.nT; ++a)
                    for (int b = 0; b < prm.nH; ++b)
                        for (int c = 0; c < prm.nW; ++c, ++tap)
                            vmask |= (((mt >> a) & (mh >> b) & (mw >> c)) & 1u) << tap;
            }
        }
    }

    // ---- row decode: k -> gather offset / tap / A offset (BK lanes, one chunk ahead) ------
    auto decode_rows = [&](int k0, int buf) {
        if (tid < BK) {
            const int k = k0 + tid;
            int goff = 0, sel = WIDE ? (31 | (31 << 8) | (31 << 16)) : 31, aoff = -1;
            if (k < prm.K) {
                const int c = k / prm.taps;
                const int tap = k - c * prm.taps;
                const int jt = tap / prm.nHW;
                const int r = tap - jt * prm.nHW;
                const int jh = r / prm.nW;
                const int jw = r - jh * prm.nW;
                sel = WIDE ? (jw | ((8 + jh) << 8) | ((16 + jt) << 16)) : tap;
                goff = 4 * (c * prm.gS + prm.dir * (jt * prm.gHW + jh * prm.gW + jw));
                aoff = c * prm.a_c_stride +
                       ((prm.k0T + prm.tsT * jt) * prm.kH + prm.k0H + prm.tsH * jh) * prm.kW + prm.k0W + prm.tsW * jw;
            }
            rinfo[buf][tid][0] = goff;
            rinfo[buf][tid][1] = sel;
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
            const int goff = sgpr(rinfo[buf][row][0]);
            const int sel = sgpr(rinfo[buf][row][1]);
            unsigned ok;
            if (WIDE)
                ok = (vmask >> (sel & 31)) & (vmask >> ((sel >> 8) & 31)) & (vmask >> ((sel >> 16) & 31)) & 1u;
            else
                ok = (vmask >> sel) & 1u;
            const unsigned off = (unsigned)(base_bytes + goff) | (ok - 1u);     // invalid -> 0xFFFFFFFF (range check -> 0)
            breg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (int)off, 0, 0));
        }
        if (AVEC) {
#pragma unroll
            for (int j = 0; j < AVPASS; ++j) {
                const int e = tid + NT * j;
                const int m = e % BM, kq = e / BM;
                // (hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a one-dword load, so the
                // 16-B weight loads stay plain global loads from a clamped address; rows m >= M are
                // never stored and k >= K is zeroed when the registers are written to LDS)
                const bool ok = (e < BM * (BK / 4)) && (m0 + m < prm.M) && (k0 + 4 * kq < prm.K);
                const size_t off = ok ? (size_t)(m0 + m) * prm.a_m_stride + k0 + 4 * kq : 0;
                const float4 v = *reinterpret_cast<const float4*>(A + off);
                areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                const int e = tid + NT * j;
                const int row = (e / BM) % BK, m = e % BM;
                const int aoff = rinfo[buf][row][2];
                const bool ok = (e < BM * BK) && (m0 + m < prm.M) && (aoff >= 0);
                const unsigned off = ok ? 4u * (unsigned)((m0 + m) * prm.a_m_stride + aoff) : OOB;
                areg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, (int)off, 0, 0));
            }
        }
    };

    auto store_chunk = [&](int buf, int knext) {
#pragma unroll
        for (int j = 0; j < BPASS; ++j) Bs[buf][(brow0 + BROWS * j) * LDB + bcol] = breg[j];
        if (AVEC) {
#pragma unroll
            for (int j = 0; j < AVPASS; ++j) {
                const int e = tid + NT * j;
                const int m = e % BM, kq = e / BM;
                if (e < BM * (BK / 4)) {
                    const bool kvalid = (knext + 4 * kq) < prm.K;
#pragma unroll
                    for (int r = 0; r < 4; ++r) As[buf][(4 * kq + r) * LDA + m] = kvalid ? areg[4 * j + r] : 0.f;
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
    store_chunk(0, 0);
    __syncthreads();

    const int frag_row = lane >> 4;      // k within a 4-deep MFMA step
    const int frag_col = lane & 15;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool more = (ch + 1) < nchunks;
        if (more) load_chunk((ch + 1) * BK, cur ^ 1);      // rinfo[cur^1] was decoded last iteration
        // rinfo[cur] was consumed by the loads issued in the previous iteration
        if (ch + 2 < nchunks) decode_rows((ch + 2) * BK, cur);

        const float* as = &As[cur][0];
        const float* bs = &Bs[cur][0];
        float a[BK / 4][TM], b[BK / 4][TN];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = as[(kk * 4 + frag_row) * LDA + wm0 + 16 * i + frag_col];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kk][j] = bs[(kk * 4 + frag_row) * LDB + wn0 + 16 * j + frag_col];
        }
        // keep the whole fragment burst ahead of the MFMA chain (hipcc otherwise sinks each
        // ds_read next to its first use and waits lgkmcnt(0) every 4 MFMAs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(cur ^ 1, (ch + 1) * BK);
        __syncthreads();
    }

    store_tiles<TM, TN>(prm, acc, m0 + wm0, n0 + wn0, lane, bias, C);
}

// ---------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN, bool AVEC, bool WIDE>
static int launch_cfg(const IgemmParams& prm, const float* A, const float* G, const float* bias,
                      float* C, hipStream_t stream) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    const int tiles_m = (prm.M + BM - 1) / BM;
    const long tiles_n = ((long)prm.P + BN - 1) / BN;
    const long blocks = tiles_m * tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffL) return ZSV_E_TOO_LARGE;
    hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, WGM, WGN, AVEC, WIDE>), dim3((unsigned)blocks),
                       dim3(256), 0, stream, prm, A, G, bias, C, tiles_m);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

// padded work for a BM x BN tiling
static inline double padded_work(int M, long P, int BM, int BN) {
    const double tm = (M + BM - 1) / BM, tn = (double)((P + BN - 1) / BN);
    return tm * BM * tn * BN;
}

template <bool AVEC, bool WIDE>
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
    if (const char* e = ZSV_KNOB(CONV_CFG)) best = atoi(e);     // tuning override (tools/conv_bench.py)
    switch (best) {
        case 0: return launch_cfg<9, 2, 1, 4, AVEC, WIDE>(prm, A, G, bias, C, stream);
        case 1: return launch_cfg<4, 4, 2, 2, AVEC, WIDE>(prm, A, G, bias, C, stream);
        case 2: return launch_cfg<5, 2, 1, 4, AVEC, WIDE>(prm, A, G, bias, C, stream);
        case 3: return launch_cfg<4, 2, 1, 4, AVEC, WIDE>(prm, A, G, bias, C, stream);
        default: return launch_cfg<3, 4, 1, 4, AVEC, WIDE>(prm, A, G, bias, C, stream);
    }
}

int igemm_generic(const IgemmParams& prm, bool avec, const float* A, const float* G, const float* bias,
                  float* C, hipStream_t stream) {
    const bool wide = prm.taps > 31;
    if (wide) {
        // the 7x7 stems: never vectorised A (K = 147 / 441)
        return dispatch<false, true>(prm, A, G, bias, C, stream);
    }
    if (avec) return dispatch<true, false>(prm, A, G, bias, C, stream);
    return dispatch<false, false>(prm, A, G, bias, C, stream);
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
    // byte offsets are 32-bit (buffer addressing): every tensor must stay below 2^30 elements
    const double lim = 1073741823.0;
    const double xin = (double)d->N * d->Cin * d->Ti * d->Hi * d->Wi;
    const double yout = (double)d->N * d->Cout * d->To * d->Ho * d->Wo;
    const double wn = (double)d->Cout * d->Cin * d->kT * d->kH * d->kW;
    if (xin >= lim || yout >= lim || wn >= lim) return ZSV_E_TOO_LARGE;
    return ZSV_OK;
}

}  // namespace zsv
