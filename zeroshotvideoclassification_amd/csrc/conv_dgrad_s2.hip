// conv_dgrad_s2.hip -- input gradient of the stride-2 convolutions of Conv2Plus1D (gfx950), every residue class in ONE launch.
//
// The first block of layer2..4 strides its 1x3x3 convolution by (1,2,2) and its 3x1x1 convolution by (2,1,1)
// (resnet.py:40-52 with the strides of :217-220).  The input gradient of a stride-2, pad-1, 3-tap axis splits into two
// residue classes of input coordinates x = 2x' + r:
//     r = 0:  dx[2x']   = W[1] dy[x']                       (one tap)
//     r = 1:  dx[2x'+1] = W[0] dy[x'+1] + W[2] dy[x']       (two taps)
// conv_api.hip used to run one direct-kernel launch per class (4 for the spatial, 2 for the temporal form): every class
// gathered dy again (9 / 3 gathers of the same voxels over the launches), each launch alone under-filled the chip on
// layer3/4 (34 k / 4 k voxels per class), and a class' outputs are every second float of a row.  Here a workgroup owns BN
// consecutive dy voxels x' and produces ALL classes of those voxels:
//   * B operand: per 8-channel chunk ONE image of dy -- spatial form: the BN voxels plus Wo + 1 more of the flattened
//     (t, h', w') order, so the shifted operands dy[h'+a][w'+b] are the same LDS rows read Wo*a + b floats further (voxels
//     on the last row / column read the next row's data: zeroed in registers by per-lane masks); temporal form: the BN
//     voxels of frame t' and of frame t'+1 side by side.  4-byte LDS-DMA, no staging registers.
//   * A operand: the chunk's weights, all taps, [tap][8 k][64 m], 16-byte LDS-DMA of a panel packed once per call; odd k rows
//     are stored with their two 16-column halves swapped so the four k rows of a fragment read split the banks.
//   * accumulators: one 32 x BN/2 wave tile PER CLASS (4 or 2 classes); tap (kh, kw) multiplies image shift
//     (kh == 0, kw == 0) into class (kh != 1, kw != 1): 9 (3) MFMA groups per k step from 4 (2) B and 9 (3) A fragment sets.
//   * epilogue: the spatial classes (rh, 0) and (rh, 1) of a voxel are neighbours in a dx row: 8-byte stores, 128 contiguous
//     bytes per 16 lanes; nothing is written twice and no class needs a memset.
//   * small problems (layer4: 4 k voxels) cut the K range into parts whose slabs splitk_reduce() adds in fixed order.
#include <stdlib.h>
#include "conv_params.h"
#include "knobs.h"
#include "pack_bodies.h"

namespace zsv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef S2ABL
#define S2ABL 0        // timing-only ablation builds (tools/variant.sh): 1 no DMAs after the first chunk, 2 no fragment reads, 4 no chunk-end wait / barrier, 8 no stores
#endif

struct DgradS2Params {
    int M, Mp, tiles_m;      // rows: dx channels (Cin), padded to whole 64-row tiles
    int Cout, nchunks;       // reduction: dy channels in chunks of 8
    int P;                   // columns: dy voxels N * To * Ho * Wo
    int S, HW, Wc, Hc, Tc;   // dy: voxels per clip, per frame, row length, rows, frames
    int oS, oHW, oW;         // dx: voxels per clip, per frame, row length
    int nseg;                // 64-float DMA segments of one image row (4-byte form)
    int nimg;                // floats of one image row that are read
    unsigned dy_bytes;
    int ksplit, chunks_per_split, slab_elems;
    // != nullptr (spatial form): the gradient of a strided 1x1x1 shortcut convolution of the same input, [N][M][subT][Hc][Wc]
    // (its voxels are the class-(0,0) voxels of every sub_st-th frame), added in the epilogue: dx[.., sub_st*a, 2b, 2c] += sub[.., a, b, c]
    const float* sub;
    int sub_st, subT;
};

enum { KIND_HW = 0, KIND_T = 1 };
template <int KIND> struct S2Kind;
template <> struct S2Kind<KIND_HW> { static constexpr int NTAP = 9, NCLS = 4, NSH = 4; };
template <> struct S2Kind<KIND_T> { static constexpr int NTAP = 3, NCLS = 2, NSH = 2; };

// Wp[((chunk * NTAP + tap) * 8 + co % 8) * Mp + m] = W[co][m][tap], co = 8 * chunk + ..., zero in the padding
__global__ __launch_bounds__(256) void dgrad_s2_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp, int M, int Mp,
                                                            int Cout, int ntap, long total) {
    const PackS2Args a = {M, Mp, Cout, ntap};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) Wp[i] = pack_s2_value(a, W, i);
}

// X4: dy clips are whole 16-byte pieces (S % 4 == 0; temporal form: frames too): one 16-byte DMA instruction per image row
template <int BN, int KIND, bool X4>
__global__ __launch_bounds__(256, 2) void conv_dgrad_s2_kernel(DgradS2Params prm, const float* __restrict__ Wp,
                                                               const float* __restrict__ DY, float* __restrict__ DX) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 64, BK = 8, TM = 2, TN = BN / 32;
    constexpr int NTAP = S2Kind<KIND>::NTAP, NCLS = S2Kind<KIND>::NCLS, NSH = S2Kind<KIND>::NSH;
    constexpr int MAXSEG = X4 ? 1 : (KIND == KIND_T ? 2 * BN / 64 : (BN == 128 ? 4 : 3));
    constexpr int LBP = X4 ? 272 : 64 * MAXSEG + 16;       // = 16 mod 32: the two k rows of a 32-lane read group split the banks
    constexpr int A_FLOATS = NTAP * BK * BM, B_FLOATS = BK * LBP, STAGE = A_FLOATS + B_FLOATS;
    constexpr int APIECES = NTAP * BK * BM / 256;          // 1-KiB pieces of the weight panel (4 k rows x 64 m)
    constexpr unsigned OOB = X4 ? 0xFFFFFFF0u : 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) float pool[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int lid = xcd_tile(gridDim.x, blockIdx.x);
    const int split = lid % prm.ksplit, tile = lid / prm.ksplit;
    const int m0 = (tile % prm.tiles_m) * BM;
    const int n0 = (tile / prm.tiles_m) * BN;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (BN / 2);
    if (prm.ksplit > 1) DX += (size_t)split * prm.slab_elems;

    // ---- image DMA: element e of a k row = dy voxel n0 + e (spatial) / voxel n0 + e % BN of frame t' + e / BN (temporal)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);
    const int ch_bytes = 4 * prm.S;
    unsigned ibase[MAXSEG];
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
        const int e = X4 ? 4 * lane : 64 * s + lane;
        const int q = KIND == KIND_T ? n0 + e % BN : n0 + e;
        ibase[s] = OOB;
        if (q < prm.P && (!X4 || e < prm.nimg)) {
            const int n = q / prm.S;
            ibase[s] = 4u * (unsigned)(n * prm.Cout * prm.S + (q - n * prm.S) + (KIND == KIND_T ? (e / BN) * prm.HW : 0));
        }
    }
    // ---- weight panel DMA: piece q = 4 k rows; lane = (row lane >> 4, 16-byte slot lane & 15); odd rows: halves swapped
    const float* a_lane = Wp + (size_t)g * prm.Mp + m0 + 4 * ((lane & 15) ^ ((g & 1) << 2));

    const int c_first = split * prm.chunks_per_split;
    const int nchunks = min(prm.chunks_per_split, prm.nchunks - c_first);
    auto issue = [&](int chunk, int buf) {
        float* as = pool + buf * STAGE;
        float* bs = as + A_FLOATS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                      // k rows wave and wave + 4
            const int k = wave + 4 * h;
            const int co = chunk * BK + k;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                const int voff = (int)(co < prm.Cout ? ibase[s] : OOB), soff = co < prm.Cout ? co * ch_bytes : 0;
                if constexpr (X4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LBP), 16, voff, soff, 0, 0);
                else if (s < prm.nseg) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LBP + 64 * s), 4, voff, soff, 0, 0);
            }
        }
        const float* src = a_lane + (size_t)chunk * (NTAP * BK) * prm.Mp;
#pragma unroll
        for (int j = 0; j < (APIECES + 3) / 4; ++j) {
            const int piece = wave + 4 * j;
            if (piece < APIECES)
                __builtin_amdgcn_global_load_lds(src + (size_t)(4 * piece) * prm.Mp, (lds_ptr_t)(as + 256 * piece), 16, 0, 0);
        }
    };

    // ---- this lane's fragment columns: image position and the masks of the shifted operands
    int shoff[NSH];
    if constexpr (KIND == KIND_HW) { shoff[0] = 0; shoff[1] = 1; shoff[2] = prm.Wc; shoff[3] = prm.Wc + 1; }
    else { shoff[0] = 0; shoff[1] = BN; }
    bool last_a[TN], last_b[TN];                            // voxel on the last row (last frame) / last column of its frame
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = n0 + wn0 + 16 * j + r16;
        int r = p % prm.S;
        const int t = r / prm.HW;
        r -= t * prm.HW;
        const int hh = r / prm.Wc, ww = r - hh * prm.Wc;
        last_a[j] = KIND == KIND_T ? t == prm.Tc - 1 : hh == prm.Hc - 1;
        last_b[j] = KIND == KIND_T ? false : ww == prm.Wc - 1;
    }
    const int a_col0 = (wm0 + r16) ^ ((g & 1) << 4);

    f32x4 acc[NCLS][TM][TN];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nchunks > 0) issue(c_first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks && !(S2ABL & 1)) issue(c_first + ch + 1, cur ^ 1);
        const float* as = pool + cur * STAGE;
        const float* bs = as + A_FLOATS;
        float a[2][NTAP][TM], b[2][NSH][TN];
        auto fetch = [&](int ks, int slot) {
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
                for (int i = 0; i < TM; ++i) a[slot][tap][i] = as[(tap * BK + 4 * ks + g) * BM + (a_col0 ^ (16 * i))];
#pragma unroll
            for (int s = 0; s < NSH; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[slot][s][j] = bs[(4 * ks + g) * LBP + wn0 + 16 * j + r16 + shoff[s]];
        };
        auto mask = [&](int slot) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (KIND == KIND_HW) {
                    if (last_b[j]) { b[slot][1][j] = 0.f; b[slot][3][j] = 0.f; }
                    if (last_a[j]) { b[slot][2][j] = 0.f; b[slot][3][j] = 0.f; }
                } else {
                    if (last_a[j]) b[slot][1][j] = 0.f;
                }
            }
        };
        if (!(S2ABL & 2) || ch == 0) fetch(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int sl = ks & 1;
            if (ks + 1 < BK / 4 && (!(S2ABL & 2) || ch == 0)) fetch(ks + 1, sl ^ 1);
            mask(sl);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                // spatial: tap = 3 kh + kw -> class 2 (kh != 1) + (kw != 1), shift 2 (kh == 0) + (kw == 0); temporal: tap = kt
                const int kh = KIND == KIND_HW ? tap / 3 : tap, kw = KIND == KIND_HW ? tap % 3 : 1;
                const int cls = KIND == KIND_HW ? 2 * (kh != 1) + (kw != 1) : (kh != 1);
                const int sh = KIND == KIND_HW ? 2 * (kh == 0) + (kw == 0) : (kh == 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[cls][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sl][tap][i], b[sl][sh][j], acc[cls][i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(S2ABL & 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if ((S2ABL & 8) && prm.P > 0) return;

    // ---- epilogue: acc[cls][i][j][r]: row m0 + wm0 + 16 i + 4 g + r, dy voxel n0 + wn0 + 16 j + r16
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = n0 + wn0 + 16 * j + r16;
        if (p >= prm.P) continue;
        const int n = p / prm.S;
        int r = p - n * prm.S;
        const int t = r / prm.HW;
        r -= t * prm.HW;
        const int hh = r / prm.Wc, ww = r - hh * prm.Wc;
        float* base = DX + (size_t)n * prm.M * prm.oS;
        if constexpr (KIND == KIND_HW) base += t * prm.oHW + 2 * hh * prm.oW + 2 * ww;
        else base += 2 * t * prm.oHW + r;
        // shortcut gradient (spatial form; with K parts: added by part 0 only)
        const float* sub = nullptr;
        int sub_pitch = 0;
        if constexpr (KIND == KIND_HW) {
            if (prm.sub != nullptr && split == 0 && t % prm.sub_st == 0) {
                sub_pitch = prm.subT * prm.HW;
                sub = prm.sub + (size_t)n * prm.M * sub_pitch + (t / prm.sub_st) * prm.HW + r;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = m0 + wm0 + 16 * i + 4 * g + r4;
                if (m >= prm.M) continue;
                float* o = base + (size_t)m * prm.oS;
                if constexpr (KIND == KIND_HW) {
                    if (sub != nullptr) acc[0][i][j][r4] += sub[(size_t)m * sub_pitch];
                    *reinterpret_cast<float2*>(o) = float2{acc[0][i][j][r4], acc[1][i][j][r4]};
                    *reinterpret_cast<float2*>(o + prm.oW) = float2{acc[2][i][j][r4], acc[3][i][j][r4]};
                } else {
                    o[0] = acc[0][i][j][r4];
                    o[prm.oHW] = acc[1][i][j][r4];
                }
            }
    }
#endif
}

// ---- host side -----------------------------------------------------------------------------------
static int s2_kind(const zsv_conv_desc* d) {
    if (d->kT == 1 && d->sT == 1 && d->pT == 0 && d->kH == 3 && d->kW == 3 && d->sH == 2 && d->sW == 2 && d->pH == 1 && d->pW == 1 &&
        d->Hi % 2 == 0 && d->Wi % 2 == 0)
        return KIND_HW;
    if (d->kT == 3 && d->sT == 2 && d->pT == 1 && d->kH == 1 && d->kW == 1 && d->sH == 1 && d->sW == 1 && d->pH == 0 && d->pW == 0 &&
        d->Ti % 2 == 0)
        return KIND_T;
    return -1;
}

struct S2Plan { int kind, bn, ks, tiles_m, tiles_n, nchunks, cps; };

static bool s2_plan(const zsv_conv_desc* d, S2Plan& pl) {
    pl.kind = s2_kind(d);
    if (pl.kind < 0 || ZSV_KNOB(NO_DGRAD_S2)) return false;
    if (pl.kind == KIND_T && ZSV_KNOB(NO_DGRAD_S2T)) return false;
    if (d->Cout < 8 || d->Cin < 16 || d->Wo + 1 > 128) return false;
    const long P = (long)d->N * d->To * d->Ho * d->Wo;
    // byte offsets live in 32-bit registers, the sentinel 0xFFFFFFFF must stay out of range: tensors below 2^29 elements
    if ((long)d->Cout * P >= (1L << 29) || (long)d->N * d->Cin * d->Ti * d->Hi * d->Wi >= (1L << 29)) return false;
    pl.tiles_m = (d->Cin + 63) / 64;
    pl.nchunks = (d->Cout + 7) / 8;
    const long t128 = pl.tiles_m * ((P + 127) / 128);
    const char* e = ZSV_KNOB(DGRAD_S2_BN);
    pl.bn = e ? atoi(e) : (t128 >= 1536 ? 128 : 64);
    if (pl.bn != 64 && pl.bn != 128) return false;
    pl.tiles_n = (int)((P + pl.bn - 1) / pl.bn);
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    // two workgroups per CU are resident: below ~1.4 rounds of 512, cut K into parts (>= 12 chunks each)
    long ks = 1;
    if (tiles < 700) {
        ks = (1024 + tiles / 2) / tiles;
        if (ks > pl.nchunks / 12) ks = pl.nchunks / 12;
        if (ks > 8) ks = 8;
        if (ks < 1) ks = 1;
    }
    if (const char* k = ZSV_KNOB(DGRAD_S2_KS)) ks = atol(k) < 1 ? 1 : atol(k);
    pl.ks = (int)ks;
    pl.cps = (pl.nchunks + pl.ks - 1) / pl.ks;
    pl.ks = (pl.nchunks + pl.cps - 1) / pl.cps;          // no empty part
    return true;
}

static size_t s2_align(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t s2_panel_bytes(const zsv_conv_desc* d, const S2Plan& pl) {
    const int ntap = pl.kind == KIND_HW ? 9 : 3;
    return s2_align((size_t)pl.nchunks * ntap * 8 * pl.tiles_m * 64 * sizeof(float));
}

bool dgrad_s2_applicable(const zsv_conv_desc* d) {
    S2Plan pl;
    return s2_plan(d, pl);
}

size_t dgrad_s2_workspace_bytes(const zsv_conv_desc* d) {
    S2Plan pl;
    if (!s2_plan(d, pl)) return 0;
    const size_t out_bytes = (size_t)d->N * d->Cin * d->Ti * d->Hi * d->Wi * sizeof(float);
    return s2_panel_bytes(d, pl) + (pl.ks > 1 ? (size_t)pl.ks * out_bytes : 0);
}

template <int BN, int KIND, bool X4>
static int s2_launch(const DgradS2Params& p, const float* wp, const float* dy, float* out, hipStream_t stream) {
    constexpr int NTAP = S2Kind<KIND>::NTAP;
    constexpr int MAXSEG = KIND == KIND_T ? 2 * BN / 64 : (BN == 128 ? 4 : 3);
    constexpr int LDS_BYTES = 2 * (NTAP * 8 * 64 + 8 * (X4 ? 272 : 64 * MAXSEG + 16)) * 4;          // as in the kernel
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_dgrad_s2_kernel<BN, KIND, X4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    const long blocks = (long)p.tiles_m * ((p.P + BN - 1) / BN) * p.ksplit;
    hipLaunchKernelGGL((conv_dgrad_s2_kernel<BN, KIND, X4>), dim3((unsigned)blocks), dim3(256), LDS_BYTES, stream, p, wp, dy, out);
    return launch_status();
}
template <int BN, int KIND>
static int s2_launch_x(bool x4, const DgradS2Params& p, const float* wp, const float* dy, float* out, hipStream_t stream) {
    return x4 ? s2_launch<BN, KIND, true>(p, wp, dy, out, stream) : s2_launch<BN, KIND, false>(p, wp, dy, out, stream);
}

// dx += zero-insertion of `sub` (the gradient of a 1x1x1 convolution of stride (sub_st, 2, 2) on the same input) -- spatial form only
bool dgrad_s2_sub_supported(const zsv_conv_desc* d, int st, int sh, int sw) {
    S2Plan pl;
    return s2_plan(d, pl) && pl.kind == KIND_HW && sh == 2 && sw == 2 && (st == 1 || st == 2) && ZSV_KNOB(NO_DOWN_FUSION) == nullptr;
}

int dgrad_s2(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int sub_st, float* dx, void* workspace,
             size_t workspace_bytes, hipStream_t stream) {
    S2Plan pl;
    if (!s2_plan(d, pl)) return ZSV_E_UNSUPPORTED;
    if (sub != nullptr && !dgrad_s2_sub_supported(d, sub_st, 2, 2)) return ZSV_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < dgrad_s2_workspace_bytes(d) || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return ZSV_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(dx) & 7) != 0) return ZSV_E_UNSUPPORTED;
    const int ntap = pl.kind == KIND_HW ? 9 : 3;
    DgradS2Params p;
    p.M = d->Cin; p.tiles_m = pl.tiles_m; p.Mp = pl.tiles_m * 64;
    p.Cout = d->Cout; p.nchunks = pl.nchunks;
    p.S = d->To * d->Ho * d->Wo; p.HW = d->Ho * d->Wo; p.Wc = d->Wo; p.Hc = d->Ho; p.Tc = d->To;
    p.P = d->N * p.S;
    p.oS = d->Ti * d->Hi * d->Wi; p.oHW = d->Hi * d->Wi; p.oW = d->Wi;
    p.nimg = pl.kind == KIND_T ? 2 * pl.bn : pl.bn + d->Wo + 1;
    p.nseg = (p.nimg + 63) / 64;
    const bool x4 = p.S % 4 == 0 && (pl.kind == KIND_HW || p.HW % 4 == 0) && p.nimg <= 256 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                    ZSV_KNOB(DGRAD_S2_NO_X4) == nullptr;
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.S);
    p.ksplit = pl.ks; p.chunks_per_split = pl.cps;
    p.slab_elems = (int)((long)d->N * d->Cin * p.oS);
    p.sub = sub; p.sub_st = sub ? sub_st : 1; p.subT = sub ? (d->Ti + sub_st - 1) / sub_st : 0;
    float* wp;
    int pst;
    if (!panel_place(s2_panel_bytes(d, pl), workspace, wp, pst)) return pst;
    float* slabs = (float*)((char*)workspace + s2_panel_bytes(d, pl));
    const long total = (long)pl.nchunks * ntap * 8 * p.Mp;
    if (g_panel.mode == PANEL_RECORD) {
        pack_job_s2(g_panel.job, PackS2Args{p.M, p.Mp, p.Cout, ntap}, w, wp, total);
        g_panel.jobs++;
        return ZSV_OK;
    }
    if (g_panel.mode != PANEL_LAUNCH_ONLY) {
        long pb = (total + 255) / 256;
        if (pb > 4096) pb = 4096;
        hipLaunchKernelGGL(dgrad_s2_pack_kernel, dim3((unsigned)pb), dim3(256), 0, stream, w, wp, p.M, p.Mp, p.Cout, ntap, total);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    }
    float* out = pl.ks > 1 ? slabs : dx;
    int st;
    if (pl.kind == KIND_HW) st = pl.bn == 128 ? s2_launch_x<128, KIND_HW>(x4, p, wp, dy, out, stream) : s2_launch_x<64, KIND_HW>(x4, p, wp, dy, out, stream);
    else st = pl.bn == 128 ? s2_launch_x<128, KIND_T>(x4, p, wp, dy, out, stream) : s2_launch_x<64, KIND_T>(x4, p, wp, dy, out, stream);
    if (st || pl.ks == 1) return st;
    return splitk_reduce(slabs, pl.ks, p.slab_elems, d->Cin, p.oS, nullptr, 0, dx, stream);
}

}  // namespace zsv
