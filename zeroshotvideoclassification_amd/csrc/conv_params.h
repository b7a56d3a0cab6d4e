// conv_params.h -- launch geometry shared by the convolution kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zsv_hip.h"
#include "zsv_common.h"

namespace zsv {

struct IgemmParams {
    int M;          // rows: channels produced (Cout fwd / Cin dgrad)
    int P;          // columns: voxels of this launch (N * class grid)
    int K;          // reduction: Cred * taps
    // taps of this launch (the whole kernel for fwd, one residue class for dgrad)
    int taps, nHW, nW, nT, nH;
    // column decode over the class grid (cT, cH, cW)
    int cS, cHW, cW;
    // produced tensor (full) and where class voxel (ct, ch, cw) lands in it
    int oS, oHW, oW;
    int stT, stH, stW, rT, rH, rW;
    // gathered tensor and the gather rule: coord = c * gs + go + dir * j, valid in [0, g)
    int gC, gT, gH, gW, gS, gHW;
    int gsT, gsH, gsW, goT, goH, goW;
    int dir;
    // weights: A(m, k = (c, tap)) = a[m * a_m_stride + c * a_c_stride + tap_full]
    //   tap_full = ((k0T + tsT*jt) * kH + k0H + tsH*jh) * kW + k0W + tsW*jw
    int a_m_stride, a_c_stride;
    int k0T, k0H, k0W, tsT, tsH, tsW, kH, kW;
    unsigned g_bytes, a_bytes;     // buffer sizes for the hardware range check
    int relu;
    int ksplit;     // > 1: the K range is cut into ksplit parts, each writing a partial C slab
    int slab_elems; // elements of one C slab (= N * M * oS)
    int lds_epilogue;   // 1: output rows are voxel-contiguous -> LDS-transposed full-line stores
    float* stat_sum;    // != nullptr (lds_epilogue only): per (channel, column tile) partial sum and sum of
    float* stat_sq;     //   squares of the produced values, [M][tiles_n] each -- BatchNorm statistics for free
    int tiles_n;
    const float* pre_coef;  // != nullptr: the gathered tensor is read as relu(g * scale[c] + shift[c]) -- a BatchNorm + ReLU that
    int pre_pitch;          //   was never materialised; [2][pre_pitch] floats (scale row, shift row), zero beyond gC (conv_tap.hip PRE)
    int t2_cin;             // != 0: the two-frame temporal convolution in dense form (t2_dense below): M, gC are the DOUBLED channel
                            //   counts, the weight element of (row m, reduction channel c) is read from the ORIGINAL tensor
                            //   W[Cout][t2_cin][3]; the pack kernels ignore a_m_stride / a_c_stride then
    const float* acc_src;   // != nullptr (no split-K): C = act(result + acc_src + bias), acc_src laid out like C -- the
                            //   gradient of an identity shortcut in a dgrad, or `out += residual` of an inference forward
                            //   (resnet.py:110), added in the epilogue
};

// ---- a 3x1x1 stride-1 pad-1 convolution over TWO frames is a dense 1x1x1 convolution --------------------------------
// (resnet.py:46-52 at layer4: T9 / T10, 921 -> 512 and 1152 -> 512 channels on 2x7x7 maps.)  out[co][to] = sum_{c,kt}
// W[co][c][kt] * x[c][to + kt - 1]: with T = 2 every (to, ti) pair has exactly one tap kt = ti - to + 1 in range, so with
// channel indices c' = 2c + ti and co' = 2co + to -- which is how (c, t) pairs already lie in an NCDHW tensor with two frames:
// x[n][c][t][hw] = x'[n][2c + t][hw] -- the layer is y' = W' x' with a dense W'[2 Cout][2 Cin], W'[co'][c'] =
// W[co'>>1][c'>>1][(c'&1) - (co'&1) + 1].  The direct form multiplies the two padding frames as zeros: 6 products per
// (co, c, position) against 4 here -- 1.5x fewer MFMAs, no tensor is copied (the views are free, W' only exists in the packed panel).
inline bool t2_dense_shape(const zsv_conv_desc* d) {
    return d->kT == 3 && d->kH == 1 && d->kW == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 0 &&
           d->pW == 0 && d->Ti == 2 && d->To == 2 && d->Cin >= 16 && d->Cout >= 16;
}
inline zsv_conv_desc t2_dense_desc(const zsv_conv_desc* d) {
    zsv_conv_desc e = *d;
    e.Cin = 2 * d->Cin; e.Cout = 2 * d->Cout;
    e.Ti = e.To = 1; e.kT = 1; e.pT = 0;
    return e;
}
// element offset in W[Cout][cin][3] of the dense weight (row, col) = (co', c')
__host__ __device__ inline long t2_weight_offset(int co2, int c2, int cin) {
    return ((long)(co2 >> 1) * cin + (c2 >> 1)) * 3 + ((c2 & 1) - (co2 & 1) + 1);
}

// ---- weight panels supplied by the caller (zsv_hip.h: zsv_conv3d_*_panel) -------------------------------------------------
// The public entry points keep ONE dispatch: the *_panel / query / job forms set this thread-local context and call the
// ordinary entry point; the function that would launch a pack kernel looks at it.
enum { PANEL_NONE = 0, PANEL_LAUNCH_ONLY = 2, PANEL_RECORD = 3, PANEL_QUERY = 4 };
struct PanelCtx {
    int mode;
    void* ptr;              // LAUNCH_ONLY / RECORD: the caller's panel
    size_t bytes;
    size_t need;            // QUERY: bytes of the panel this call would pack
    int jobs;               // QUERY / RECORD: pack launches the call would make (> 1: not cacheable)
    zsv_pack_job job;       // RECORD: the pack launch as a job of zsv_pack_multi
};
extern thread_local PanelCtx g_panel;
inline bool panel_stop() { return g_panel.mode == PANEL_RECORD || g_panel.mode == PANEL_QUERY; }
// Where the panel of this call lives: the caller's buffer or the head of the workspace.  false = stop here (status in `st`).
inline bool panel_place(size_t need, void* workspace, float*& panel, int& st) {
    PanelCtx& pc = g_panel;
    st = ZSV_OK;
    panel = (float*)workspace;
    if (pc.mode == PANEL_QUERY) { pc.need = need; pc.jobs++; return false; }
    if (pc.mode != PANEL_NONE) {
        if (!pc.ptr || pc.bytes < need || (reinterpret_cast<uintptr_t>(pc.ptr) & 15) != 0) { st = ZSV_E_WORKSPACE; return false; }
        panel = (float*)pc.ptr;
    }
    return true;
}

template <int X> struct LdPad { static constexpr int value = (X % 32 == 16) ? X : X + 16; };

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }


// ---- shared epilogue: accumulator tiles -> C[n][m][sp] (+bias, ReLU) -----------------------
// acc(i,j)[r]: row m = m_base + 16*i + 4*(lane>>4) + r, column p = p_base + 16*j + (lane&15)
template <int TM, int TN>
__device__ __forceinline__ void store_tiles(const IgemmParams& prm, const f32x4 (&acc)[TM][TN], int m_base, int p_base,
                                            int lane, const float* __restrict__ bias, float* __restrict__ C) {
    const int frag_row = lane >> 4, frag_col = lane & 15;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = p_base + 16 * j + frag_col;
        if (p >= prm.P) continue;
        const int n = p / prm.cS;
        int r = p - n * prm.cS;
        const int ct = r / prm.cHW;
        r -= ct * prm.cHW;
        const int chh = r / prm.cW;
        const int cw = r - chh * prm.cW;
        const int sp = (ct * prm.stT + prm.rT) * prm.oHW + (chh * prm.stH + prm.rH) * prm.oW + cw * prm.stW + prm.rW;
        float* cbase = C + (size_t)n * prm.M * prm.oS + sp;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = m_base + 16 * i + 4 * frag_row + r4;
                if (m < prm.M) {
                    float v = acc[i][j][r4];
                    if (prm.acc_src != nullptr) v += prm.acc_src[(cbase - C) + (size_t)m * prm.oS];
                    if (bias != nullptr) v += bias[m];
                    if (prm.relu) v = fmaxf(v, 0.f);
                    cbase[(size_t)m * prm.oS] = v;
                }
            }
        }
    }
}

// ---- LDS-transposed epilogue: full-line stores ------------------------------------------------
// The accumulator layout puts 16 consecutive voxels of one channel on 16 lanes (64-B store
// segments).  When the produced tensor is voxel-contiguous for this launch (stW == 1, oS % 4 == 0)
// the workgroup instead transposes RP-row slices of its BM x BN tile through LDS (`cs`, aliasing the
// staging buffers, row pitch BN + 4 floats) and every wave-instruction stores whole 512-B (BN = 128)
// or 1-KB (BN = 256) channel rows with 16-B lanes.  `cap` = floats available in `cs`.
// All 256 threads must call it (it contains barriers).
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void store_tiles_lds(const IgemmParams& prm, const f32x4 (&acc)[TM][TN], float* cs, int cap,
                                                int m0, int n0, int wave, int lane, const float* __restrict__ bias,
                                                float* __restrict__ C) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    constexpr int PITCH = BN + 4;
    constexpr int RPI = 256 / BN == 0 ? 1 : 256 / BN;      // rows per wave-instruction of the read-back (2 or 1)
    static_assert(BN == 128 || BN == 256, "read-back assumes 32 or 64 float4 per row");
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);
    const int frag_row = lane >> 4, frag_col = lane & 15;
    int rp = (cap / PITCH) & ~15;                         // rows per pass: multiple of 16 that fits
    if (rp > BM) rp = BM;
    const int c4 = BN == 128 ? (lane & 31) : lane;          // float4 column handled in the read-back
    const int rsub = BN == 128 ? (lane >> 5) : 0;
    for (int r0 = 0; r0 < BM; r0 += rp) {
        __syncthreads();                                    // previous pass read back / main loop done
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = wm0 + 16 * i - r0;               // first row of tile i inside this pass
            if (rb >= 0 && rb < rp) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[(rb + 4 * frag_row + r) * PITCH + wn0 + 16 * j + frag_col] = acc[i][j][r];
            }
        }
        __syncthreads();
        const int rows = min(rp, BM - r0);
        const int p = n0 + 4 * c4;                          // 4 voxels of one clip (oS % 4 == 0)
        const bool pvalid = p < prm.P;
        const int n = pvalid ? p / prm.cS : 0;
        float* cbase = C + (size_t)n * prm.M * prm.oS + (p - n * prm.cS);
        // shortcut gradient to add (dgrad): the NEXT row's 16 bytes are fetched while this row is stored
        f32x4 addn = {0.f, 0.f, 0.f, 0.f};
        auto fetch_add = [&](int rr) {
            const int m = m0 + r0 + rr;
            addn = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pvalid && m < prm.M) addn = *reinterpret_cast<const f32x4*>(prm.acc_src + (cbase - C) + (size_t)m * prm.oS);
        };
        const bool has_add = prm.acc_src != nullptr;
        if (has_add && wave * RPI + rsub < rows) fetch_add(wave * RPI + rsub);
        for (int rr = wave * RPI + rsub; rr < rows; rr += 4 * RPI) {
            const int m = m0 + r0 + rr;
            f32x4 v = *reinterpret_cast<const f32x4*>(&cs[rr * PITCH + 4 * c4]);
            if (has_add) {
                v += addn;
                if (rr + 4 * RPI < rows) fetch_add(rr + 4 * RPI);
            }
            if (prm.stat_sum != nullptr) {
                // columns beyond P hold exact zeros (their gathered operand was zero): no masking
                float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                for (int o = (BN == 128 ? 16 : 32); o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (c4 == 0 && m < prm.M) {
                    const int tn = n0 / BN;
                    prm.stat_sum[(size_t)m * prm.tiles_n + tn] = s1;
                    prm.stat_sq[(size_t)m * prm.tiles_n + tn] = s2;
                }
            }
            if (pvalid && m < prm.M) {
                if (bias != nullptr) { const float b = bias[m]; v[0] += b; v[1] += b; v[2] += b; v[3] += b; }
                if (prm.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<f32x4*>(cbase + (size_t)m * prm.oS) = v;
            }
        }
    }
}

// XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_tile(int nwg, int bid) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// host-side entry points of the two kernel families
int igemm_generic(const IgemmParams& prm, bool avec, const float* A, const float* G, const float* bias, float* C,
                  hipStream_t stream);
bool igemm_tap_applicable(const IgemmParams& prm);
size_t igemm_tap_workspace_bytes(const IgemmParams& prm);
// `slabs` != nullptr with prm.ksplit > 1: partial sums go to slabs[split][...] (reduce afterwards)
int igemm_tap(const IgemmParams& prm, const float* W, int w_m_stride, int w_c_stride, const float* G,
              const float* bias, float* C, void* workspace, size_t workspace_bytes, float* slabs, hipStream_t stream);
int igemm_tap_ksplit(const IgemmParams& prm);     // suggested split of the K range (1 = none)
int igemm_tap_stat_tiles(const IgemmParams& prm, const float* C);   // column tiles if the epilogue can emit BN partials, else 0
int splitk_reduce(const float* slabs, int ksplit, long elems, int M, int oS, const float* bias, int relu, float* C,
                  hipStream_t stream);
int conv_check(const zsv_conv_desc* d);
size_t wgrad_generic_workspace_bytes(const zsv_conv_desc* d);
int wgrad_generic(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                  size_t workspace_bytes, hipStream_t stream);
// Winograd F(2,3)-along-W input gradient of the 1x3x3 stride-1 convolutions (conv_wino.hip)
bool wino_dgrad_applicable(const zsv_conv_desc* d);
size_t wino_dgrad_workspace_bytes(const zsv_conv_desc* d);
int wino_dgrad(const zsv_conv_desc* d, const float* dy, const float* w, const float* add, float* dx, void* workspace,
               size_t workspace_bytes, hipStream_t stream);
// every residue class of a stride-2 dgrad in one launch (conv_dgrad_s2.hip): 1x3x3 stride (1,2,2) and 3x1x1 stride (2,1,1)
bool dgrad_s2_applicable(const zsv_conv_desc* d);
size_t dgrad_s2_workspace_bytes(const zsv_conv_desc* d);
bool dgrad_s2_sub_supported(const zsv_conv_desc* d, int st, int sh, int sw);
int dgrad_s2(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int sub_st, float* dx, void* workspace,
             size_t workspace_bytes, hipStream_t stream);
bool wino_fwd_applicable(const zsv_conv_desc* d);
bool wino_fwd_pre_capable(const zsv_conv_desc* d);   // the temporal F(2,3)-along-T forward: can apply a BatchNorm + ReLU prologue
int wino_fwd_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int pre_pitch, const float* w, float* stat_sum,
                 float* stat_sq, float* y, void* workspace, size_t workspace_bytes, hipStream_t stream);
bool wino_fwd_fusable(const zsv_conv_desc* d);       // false: split-K form, no statistics / add / residual in the epilogue
bool wino_dgrad_fusable(const zsv_conv_desc* d);
int wino_fwd_stat_tiles(const zsv_conv_desc* d);
size_t wino_fwd_workspace_bytes(const zsv_conv_desc* d);
int wino_fwd(const zsv_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual, int relu,
             float* stat_sum, float* stat_sq, float* y, void* workspace, size_t workspace_bytes, hipStream_t stream);
// LDS-DMA weight gradient of stride-1 "same" convolutions (conv_wgrad_dma.hip): writes the per-slice
// slabs [slice][Cout][taps*Cpad] at the start of `workspace`
bool wgrad_dma_applicable(const zsv_conv_desc* d, const float* x, const float* dy);
size_t wgrad_dma_workspace_bytes(const zsv_conv_desc* d);
int wgrad_dma(const zsv_conv_desc* d, const float* x, const float* dy, void* workspace, size_t workspace_bytes,
              int* slices_out, int* cpad_out, hipStream_t stream, const unsigned* vm_ext = nullptr);
// per-voxel tap-validity words of one clip (S words; bit tap <=> the tap's input voxel is inside): used by both
// LDS-DMA weight-gradient kernels
int wgrad_vmask(const zsv_conv_desc* d, unsigned* out, hipStream_t stream);
// weight gradient of the temporal 3x1x1 stride-1 convolutions with a ring of dY frames in LDS (conv_wgrad_tring.hip): slabs,
// ordered reduction; writes dw
bool wgrad_tring_applicable(const zsv_conv_desc* d, const float* x, const float* dy);
size_t wgrad_tring_workspace_bytes(const zsv_conv_desc* d);
int wgrad_tring_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int pre_pitch, const float* dy, float* dw,
                    void* workspace, size_t workspace_bytes, hipStream_t stream);
int wgrad_tring(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                hipStream_t stream);
// Winograd-form weight gradient of the 1x3x3 / 3x3x3 stride-1 "same" convolutions (conv_wgrad_wino.hip): slabs, mask
// table, ordered reduction and output transform; writes dw
bool wgrad_wino_applicable(const zsv_conv_desc* d, const float* x, const float* dy);
size_t wgrad_wino_workspace_bytes(const zsv_conv_desc* d);
int wgrad_wino(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
               hipStream_t stream, const unsigned* vm_ext = nullptr);

}  // namespace zsv
