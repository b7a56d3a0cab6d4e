// conv_api.hip -- C-ABI entry points of the forward / input-gradient convolution (gfx950).
//
// Builds the launch geometry for aten::conv3d (resnet.py:23-30,40-52,63-70,170,181,184,270;
// network.py:102-117) and its dgrad, then picks a kernel family:
//   * conv_tap.hip   (tap-major K order, packed weights, zero per-element address math) when the
//                    gathered tensor has >= 16 channels and <= 31 taps -- every layer but the stems;
//   * conv_igemm.hip (generic (channel, tap) order) otherwise (3-channel stems, 7x7 kernels).
#include <stdlib.h>
#include "conv_params.h"
#include "knobs.h"

using namespace zsv;

namespace {

void fwd_params(IgemmParams& p, const zsv_conv_desc* d, int fuse_relu) {
    p = IgemmParams{};
    const int taps = d->kT * d->kH * d->kW;
    p.M = d->Cout;
    p.P = d->N * d->To * d->Ho * d->Wo;
    p.K = d->Cin * taps;
    p.taps = taps; p.nHW = d->kH * d->kW; p.nW = d->kW; p.nT = d->kT; p.nH = d->kH;
    p.cS = d->To * d->Ho * d->Wo; p.cHW = d->Ho * d->Wo; p.cW = d->Wo;
    p.oS = p.cS; p.oHW = p.cHW; p.oW = p.cW;
    p.stT = p.stH = p.stW = 1; p.rT = p.rH = p.rW = 0;
    p.gC = d->Cin; p.gT = d->Ti; p.gH = d->Hi; p.gW = d->Wi;
    p.gS = d->Ti * d->Hi * d->Wi; p.gHW = d->Hi * d->Wi;
    p.gsT = d->sT; p.gsH = d->sH; p.gsW = d->sW; p.goT = -d->pT; p.goH = -d->pH; p.goW = -d->pW;
    p.dir = 1;
    p.a_m_stride = p.K; p.a_c_stride = taps;
    p.k0T = p.k0H = p.k0W = 0; p.tsT = p.tsH = p.tsW = 1; p.kH = d->kH; p.kW = d->kW;
    p.g_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.gS);
    p.a_bytes = 4u * (unsigned)((long)d->Cout * p.K);
    p.relu = fuse_relu ? 1 : 0;
}

// dgrad of a strided convolution = one dense stride-1 problem per residue class of input voxels
// (x = s*x' + r per axis).  Returns false when the class has no voxel.
bool dgrad_class_params(IgemmParams& p, const zsv_conv_desc* d, int rt, int rh, int rw) {
    const int taps_full = d->kT * d->kH * d->kW;
    const int dims[3] = {d->Ti, d->Hi, d->Wi};
    const int ks[3] = {d->kT, d->kH, d->kW};
    const int ss[3] = {d->sT, d->sH, d->sW};
    const int ps[3] = {d->pT, d->pH, d->pW};
    const int rr[3] = {rt, rh, rw};
    int cdim[3], k0[3], c0[3], nt[3];
    for (int a = 0; a < 3; ++a) {
        cdim[a] = (dims[a] - rr[a] + ss[a] - 1) / ss[a];          // voxels of this class along the axis
        if (cdim[a] <= 0) return false;
        k0[a] = (rr[a] + ps[a]) % ss[a];                          // first tap with matching residue
        c0[a] = (rr[a] + ps[a] - k0[a]) / ss[a];                  // out = x' + c0 - j
        nt[a] = k0[a] < ks[a] ? (ks[a] - k0[a] + ss[a] - 1) / ss[a] : 0;
    }
    p = IgemmParams{};
    p.M = d->Cin;
    p.P = d->N * cdim[0] * cdim[1] * cdim[2];
    p.nT = nt[0]; p.nH = nt[1]; p.nW = nt[2];
    p.taps = nt[0] * nt[1] * nt[2];
    if (p.taps == 0) { p.taps = 1; p.nT = p.nH = p.nW = 1; p.K = 0; }   // class sees no tap: dx = 0
    else p.K = d->Cout * p.taps;
    p.nHW = p.nH * p.nW;
    p.cS = cdim[0] * cdim[1] * cdim[2]; p.cHW = cdim[1] * cdim[2]; p.cW = cdim[2];
    p.oS = d->Ti * d->Hi * d->Wi; p.oHW = d->Hi * d->Wi; p.oW = d->Wi;
    p.stT = d->sT; p.stH = d->sH; p.stW = d->sW; p.rT = rt; p.rH = rh; p.rW = rw;
    p.gC = d->Cout; p.gT = d->To; p.gH = d->Ho; p.gW = d->Wo;
    p.gS = d->To * d->Ho * d->Wo; p.gHW = d->Ho * d->Wo;
    p.gsT = p.gsH = p.gsW = 1; p.goT = c0[0]; p.goH = c0[1]; p.goW = c0[2];
    p.dir = -1;
    p.a_m_stride = taps_full; p.a_c_stride = d->Cin * taps_full;
    p.k0T = k0[0]; p.k0H = k0[1]; p.k0W = k0[2]; p.tsT = d->sT; p.tsH = d->sH; p.tsW = d->sW;
    p.kH = d->kH; p.kW = d->kW;
    p.g_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.gS);
    p.a_bytes = 4u * (unsigned)((long)d->Cout * d->Cin * taps_full);
    p.relu = 0;
    return true;
}

const zsv_conv_desc linear_desc(int32_t rows, int32_t in_features, int32_t out_features) {
    zsv_conv_desc d = {rows, in_features, 1, 1, 1, out_features, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    return d;
}

}  // namespace

// workspace layout of a tap-kernel call: [packed weights (16-B aligned)] [split-K slabs]
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// two-frame temporal convolutions in dense 1x1x1 form (conv_params.h: t2_dense_shape)
static bool t2_dense(const zsv_conv_desc* d) { return t2_dense_shape(d) && !ZSV_KNOB(NO_T2_DENSE); }

static size_t tap_fwd_workspace(const zsv_conv_desc* d) {
    IgemmParams p;
    fwd_params(p, d, 0);
    if (!igemm_tap_applicable(p)) return 0;
    const int ks = igemm_tap_ksplit(p);
    const size_t out_elems = (size_t)d->N * d->Cout * d->To * d->Ho * d->Wo;
    return align256(igemm_tap_workspace_bytes(p)) + (ks > 1 ? (size_t)ks * out_elems * sizeof(float) : 0);
}

extern "C" size_t zsv_conv3d_fwd_workspace_bytes(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK) return 0;
    if (wino_fwd_applicable(d)) return wino_fwd_workspace_bytes(d);
    size_t need = tap_fwd_workspace(d);
    if (t2_dense(d)) {                   // (calls with a bias / ReLU / residual / statistics take the ordinary path: cover both)
        const zsv_conv_desc d2 = t2_dense_desc(d);
        const size_t b = tap_fwd_workspace(&d2);
        if (b > need) need = b;
    }
    return need;
}

extern "C" int32_t zsv_conv3d_fwd_stat_tiles(const zsv_conv_desc* d, const float* y) {
    if (conv_check(d) != ZSV_OK) return 0;
    if (wino_fwd_applicable(d)) return (ZSV_KNOB(NO_FUSED_STATS) || !wino_fwd_fusable(d)) ? 0 : wino_fwd_stat_tiles(d);
    IgemmParams p;
    fwd_params(p, d, 0);
    return igemm_tap_stat_tiles(p, y);
}

extern "C" int zsv_conv3d_fwd(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                              float* y, int fuse_relu, void* workspace, size_t workspace_bytes, void* stream) {
    return zsv_conv3d_fwd_stats(d, x, w, bias, y, fuse_relu, nullptr, 0, workspace, workspace_bytes, stream);
}

// y = act(conv + bias + residual) in one pass: tap kernel without split-K
extern "C" int32_t zsv_conv3d_fwd_add_supported(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK) return 0;
    if (wino_fwd_applicable(d)) return wino_fwd_fusable(d) ? 1 : 0;
    IgemmParams p;
    fwd_params(p, d, 0);
    return (igemm_tap_applicable(p) && igemm_tap_ksplit(p) == 1) ? 1 : 0;
}

extern "C" int zsv_conv3d_fwd_stats(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                                    float* y, int fuse_relu, float* bn_partials, int32_t stat_tiles, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return zsv_conv3d_fwd_full(d, x, w, bias, nullptr, y, fuse_relu, bn_partials, stat_tiles, workspace, workspace_bytes,
                               stream);
}

extern "C" int zsv_conv3d_fwd_add(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                                  const float* residual, float* y, int fuse_relu, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return zsv_conv3d_fwd_full(d, x, w, bias, residual, y, fuse_relu, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int zsv_conv3d_fwd_full(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                                   const float* residual, float* y, int fuse_relu, float* bn_partials,
                                   int32_t stat_tiles, void* workspace, size_t workspace_bytes, void* stream) {
    int st = conv_check(d);
    if (st) return st;
    if (!x || !w || !y) return ZSV_E_NULL;
    if (residual != nullptr && (bn_partials != nullptr || !zsv_conv3d_fwd_add_supported(d))) return ZSV_E_UNSUPPORTED;
    if (wino_fwd_applicable(d)) {
        if (bn_partials && (bias || fuse_relu || stat_tiles != wino_fwd_stat_tiles(d))) return ZSV_E_UNSUPPORTED;
        return wino_fwd(d, x, w, bias, residual, fuse_relu ? 1 : 0, bn_partials,
                        bn_partials ? bn_partials + (size_t)d->Cout * stat_tiles : nullptr, y, workspace, workspace_bytes,
                        (hipStream_t)stream);
    }
    if (t2_dense(d) && !bias && !fuse_relu && !residual && !bn_partials) {
        const zsv_conv_desc d2 = t2_dense_desc(d);
        IgemmParams p;
        fwd_params(p, &d2, 0);
        p.t2_cin = d->Cin;
        if (igemm_tap_applicable(p)) {
            const size_t need = tap_fwd_workspace(&d2);
            if (workspace_bytes < need || !workspace) return ZSV_E_WORKSPACE;
            const int ks = igemm_tap_ksplit(p);
            const size_t wbytes = align256(igemm_tap_workspace_bytes(p));
            const long out_elems = (long)d->N * d->Cout * d->To * d->Ho * d->Wo;
            float* slabs = ks > 1 ? (float*)((char*)workspace + wbytes) : nullptr;
            p.ksplit = ks;
            p.slab_elems = (int)out_elems;
            st = igemm_tap(p, w, 0, 0, x, nullptr, y, workspace, wbytes, slabs, (hipStream_t)stream);
            if (st || ks <= 1 || panel_stop()) return st;
            return splitk_reduce(slabs, ks, out_elems, d2.Cout, p.oS, nullptr, 0, y, (hipStream_t)stream);
        }
    }
    IgemmParams p;
    fwd_params(p, d, fuse_relu);
    p.acc_src = residual;
    if (bn_partials) {
        // the partials describe the raw convolution output: only without bias / ReLU, and only on the
        // kernel path zsv_conv3d_fwd_stat_tiles() promised
        if (bias || fuse_relu || stat_tiles <= 0 || stat_tiles != igemm_tap_stat_tiles(p, y)) return ZSV_E_UNSUPPORTED;
        p.stat_sum = bn_partials;
        p.stat_sq = bn_partials + (size_t)d->Cout * stat_tiles;
        p.tiles_n = stat_tiles;
    }
    if (igemm_tap_applicable(p)) {
        if (workspace_bytes < zsv_conv3d_fwd_workspace_bytes(d) || !workspace) return ZSV_E_WORKSPACE;
        const int ks = igemm_tap_ksplit(p);
        const size_t wbytes = align256(igemm_tap_workspace_bytes(p));
        const long out_elems = (long)d->N * d->Cout * d->To * d->Ho * d->Wo;
        float* slabs = ks > 1 ? (float*)((char*)workspace + wbytes) : nullptr;
        p.ksplit = ks;
        p.slab_elems = (int)out_elems;
        st = igemm_tap(p, w, p.a_m_stride, p.a_c_stride, x, bias, y, workspace, wbytes, slabs, (hipStream_t)stream);
        if (st || ks <= 1 || panel_stop()) return st;
        return splitk_reduce(slabs, ks, out_elems, d->Cout, p.oS, bias, fuse_relu ? 1 : 0, y, (hipStream_t)stream);
    }
    if (g_panel.mode != PANEL_NONE) return panel_stop() ? ZSV_OK : ZSV_E_UNSUPPORTED;      // (no weight panel on this path: jobs stays 0)
    const bool avec = (p.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    return igemm_generic(p, avec, w, x, bias, y, (hipStream_t)stream);
}

// ---- convolution of a BatchNorm + ReLU output that is never materialised -----------------------------------------
// (Conv2Plus1D's `BatchNorm3d(mid) -> ReLU -> temporal conv`, resnet.py:46-52): both the forward (direct kernel, PRE form)
// and the weight gradient (frame-ring kernel, PRE form) must be able to apply the affine + ReLU while they read x.
extern "C" int32_t zsv_conv3d_pre_supported(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK || ZSV_KNOB(NO_BN_FUSION) || !wgrad_tring_applicable(d, nullptr, nullptr)) return 0;
    if (wino_fwd_applicable(d)) return wino_fwd_pre_capable(d) ? 1 : 0;          // (the temporal F(2,3)-along-T kernel)
    IgemmParams p;
    fwd_params(p, d, 0);
    return igemm_tap_applicable(p) ? 1 : 0;
}

extern "C" int zsv_conv3d_fwd_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch,
                                  const float* w, float* y, float* bn_partials, int32_t stat_tiles, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    int st = conv_check(d);
    if (st) return st;
    if (!x || !w || !y || !pre_coef) return ZSV_E_NULL;
    if (!zsv_conv3d_pre_supported(d) || coef_pitch < d->Cin || coef_pitch % 16 != 0 ||
        (reinterpret_cast<uintptr_t>(pre_coef) & 15) != 0)
        return ZSV_E_UNSUPPORTED;
    if (wino_fwd_applicable(d)) {
        if (bn_partials && stat_tiles != zsv_conv3d_fwd_stat_tiles(d, y)) return ZSV_E_UNSUPPORTED;
        return wino_fwd_pre(d, x, pre_coef, coef_pitch, w, bn_partials, bn_partials ? bn_partials + (size_t)d->Cout * stat_tiles : nullptr,
                            y, workspace, workspace_bytes, (hipStream_t)stream);
    }
    IgemmParams p;
    fwd_params(p, d, 0);
    p.pre_coef = pre_coef;
    p.pre_pitch = coef_pitch;
    if (bn_partials) {
        if (stat_tiles <= 0 || stat_tiles != igemm_tap_stat_tiles(p, y)) return ZSV_E_UNSUPPORTED;
        p.stat_sum = bn_partials;
        p.stat_sq = bn_partials + (size_t)d->Cout * stat_tiles;
        p.tiles_n = stat_tiles;
    }
    if (workspace_bytes < zsv_conv3d_fwd_workspace_bytes(d) || !workspace) return ZSV_E_WORKSPACE;
    const int ks = igemm_tap_ksplit(p);
    const size_t wbytes = align256(igemm_tap_workspace_bytes(p));
    const long out_elems = (long)d->N * d->Cout * d->To * d->Ho * d->Wo;
    float* slabs = ks > 1 ? (float*)((char*)workspace + wbytes) : nullptr;
    p.ksplit = ks;
    p.slab_elems = (int)out_elems;
    st = igemm_tap(p, w, p.a_m_stride, p.a_c_stride, x, nullptr, y, workspace, wbytes, slabs, (hipStream_t)stream);
    if (st || ks <= 1 || panel_stop()) return st;
    return splitk_reduce(slabs, ks, out_elems, d->Cout, p.oS, nullptr, 0, y, (hipStream_t)stream);
}

// one split factor for every residue class of a dgrad call (they share the slabs)
static int dgrad_plan(const zsv_conv_desc* d, size_t& wbytes) {
    IgemmParams p;
    int ks = 0;
    bool all_tap = true, has_empty = false;
    wbytes = 0;
    for (int rt = 0; rt < d->sT; ++rt)
        for (int rh = 0; rh < d->sH; ++rh)
            for (int rw = 0; rw < d->sW; ++rw) {
                if (!dgrad_class_params(p, d, rt, rh, rw)) continue;
                if (p.K == 0) { has_empty = true; continue; }      // zero class: handled by a memset
                if (!igemm_tap_applicable(p)) { all_tap = false; continue; }
                const size_t b = align256(igemm_tap_workspace_bytes(p));
                if (b > wbytes) wbytes = b;
                const int k = igemm_tap_ksplit(p);
                if (ks == 0 || k < ks) ks = k;
            }
    // (split-K slabs would need the empty classes' voxels zeroed in every slab: keep it simple)
    return (all_tap && !has_empty && ks > 1) ? ks : 1;
}

extern "C" size_t zsv_conv3d_dgrad_workspace_bytes(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK) return 0;
    if (wino_dgrad_applicable(d)) return wino_dgrad_workspace_bytes(d);
    if (dgrad_s2_applicable(d)) return dgrad_s2_workspace_bytes(d);
    zsv_conv_desc d2;
    if (t2_dense(d)) { d2 = t2_dense_desc(d); d = &d2; }
    size_t wbytes;
    const int ks = dgrad_plan(d, wbytes);
    const size_t out_elems = (size_t)d->N * d->Cin * d->Ti * d->Hi * d->Wi;
    return wbytes + (ks > 1 ? (size_t)ks * out_elems * sizeof(float) : 0);
}

// dx = dgrad + add in one pass: stride 1 (one residue class), tap kernel, no split-K
extern "C" int32_t zsv_conv3d_dgrad_add_supported(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK || d->sT != 1 || d->sH != 1 || d->sW != 1) return 0;
    if (wino_dgrad_applicable(d)) return wino_dgrad_fusable(d) ? 1 : 0;
    if (t2_dense(d)) return 0;
    IgemmParams p;
    if (!dgrad_class_params(p, d, 0, 0, 0) || p.K == 0 || !igemm_tap_applicable(p)) return 0;
    size_t wbytes;
    return dgrad_plan(d, wbytes) == 1 ? 1 : 0;
}

extern "C" int zsv_conv3d_dgrad(const zsv_conv_desc* d, const float* dy, const float* w, float* dx,
                                void* workspace, size_t workspace_bytes, void* stream) {
    return zsv_conv3d_dgrad_add(d, dy, w, nullptr, dx, workspace, workspace_bytes, stream);
}

extern "C" int zsv_conv3d_dgrad_add(const zsv_conv_desc* d, const float* dy, const float* w, const float* add,
                                    float* dx, void* workspace, size_t workspace_bytes, void* stream) {
    int st = conv_check(d);
    if (st) return st;
    if (!dy || !w || !dx) return ZSV_E_NULL;
    if (add != nullptr && !zsv_conv3d_dgrad_add_supported(d)) return ZSV_E_UNSUPPORTED;
    if (wino_dgrad_applicable(d)) return wino_dgrad(d, dy, w, add, dx, workspace, workspace_bytes, (hipStream_t)stream);
    if (dgrad_s2_applicable(d)) return dgrad_s2(d, dy, w, nullptr, 1, dx, workspace, workspace_bytes, (hipStream_t)stream);     // (stride 2: no `add`)
    zsv_conv_desc d2;
    int t2_cin = 0;
    if (t2_dense(d)) { t2_cin = d->Cin; d2 = t2_dense_desc(d); d = &d2; }
    size_t wbytes;
    const int ks = dgrad_plan(d, wbytes);
    const long out_elems = (long)d->N * d->Cin * d->Ti * d->Hi * d->Wi;
    const size_t need = wbytes + (ks > 1 ? (size_t)ks * out_elems * sizeof(float) : 0);
    if (need > 0 && (!workspace || workspace_bytes < need)) return ZSV_E_WORKSPACE;
    float* slabs = ks > 1 ? (float*)((char*)workspace + wbytes) : nullptr;
    IgemmParams p;
    if (g_panel.mode != PANEL_NONE && (d->sT != 1 || d->sH != 1 || d->sW != 1)) {
        // class-by-class launches pack one panel per residue class into the same workspace: no single panel to keep
        if (g_panel.mode == PANEL_QUERY) { g_panel.jobs = 2; return ZSV_OK; }
        return ZSV_E_UNSUPPORTED;
    }
    // residue classes that see no tap (e.g. 7 of the 8 classes of a 1x1x1 stride-2 shortcut) are
    // exactly zero: clear dx once instead of launching a kernel per empty class
    bool any_empty = false;
    for (int rt = 0; rt < d->sT; ++rt)
        for (int rh = 0; rh < d->sH; ++rh)
            for (int rw = 0; rw < d->sW; ++rw)
                if (dgrad_class_params(p, d, rt, rh, rw) && p.K == 0) any_empty = true;
    if (any_empty && hipMemsetAsync(dx, 0, (size_t)out_elems * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return ZSV_E_LAUNCH;
    // one launch per residue class; the classes reuse the (stream-ordered) weight workspace
    for (int rt = 0; rt < d->sT; ++rt)
        for (int rh = 0; rh < d->sH; ++rh)
            for (int rw = 0; rw < d->sW; ++rw) {
                if (!dgrad_class_params(p, d, rt, rh, rw)) continue;
                if (p.K == 0) continue;                      // cleared above
                if (igemm_tap_applicable(p)) {
                    p.ksplit = ks;
                    p.slab_elems = (int)out_elems;
                    p.acc_src = add;
                    p.t2_cin = t2_cin;
                    st = igemm_tap(p, w, p.a_m_stride, p.a_c_stride, dy, nullptr, dx, workspace, wbytes, slabs,
                                   (hipStream_t)stream);
                } else {
                    if (g_panel.mode != PANEL_NONE) return panel_stop() ? ZSV_OK : ZSV_E_UNSUPPORTED;
                    st = igemm_generic(p, false, w, dy, nullptr, dx, (hipStream_t)stream);
                }
                if (st) return st;
            }
    if (ks > 1 && !panel_stop()) return splitk_reduce(slabs, ks, out_elems, d->Cin, p.oS, nullptr, 0, dx, (hipStream_t)stream);
    return ZSV_OK;
}

// ---- dx = dgrad + the gradient of the block's strided 1x1x1 shortcut convolution (resnet.py:240-246: `downsample`), which reads the
// same input: `sub` is that gradient in compact form [N][Cin][ceil(Ti/st)][Hi/sh][Wi/sw] (its non-zero voxels only), added where
// dx[.., st*a, sh*b, sw*c] is produced -- no zero-filled full-size tensor, no separate add pass over the block input
extern "C" int32_t zsv_conv3d_dgrad_add_strided_supported(const zsv_conv_desc* d, int32_t st, int32_t sh, int32_t sw) {
    if (conv_check(d) != ZSV_OK || wino_dgrad_applicable(d)) return 0;
    return dgrad_s2_sub_supported(d, st, sh, sw) ? 1 : 0;
}

extern "C" int zsv_conv3d_dgrad_add_strided(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int32_t st,
                                            int32_t sh, int32_t sw, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = conv_check(d);
    if (rc) return rc;
    if (!dy || !w || !dx || !sub) return ZSV_E_NULL;
    if (!zsv_conv3d_dgrad_add_strided_supported(d, st, sh, sw)) return ZSV_E_UNSUPPORTED;
    return dgrad_s2(d, dy, w, sub, st, dx, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t zsv_linear_fwd_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features) {
    const zsv_conv_desc d = linear_desc(rows, in_features, out_features);
    return zsv_conv3d_fwd_workspace_bytes(&d);
}

extern "C" int zsv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t rows,
                              int32_t in_features, int32_t out_features, int fuse_relu, void* workspace,
                              size_t workspace_bytes, void* stream) {
    const zsv_conv_desc d = linear_desc(rows, in_features, out_features);
    return zsv_conv3d_fwd(&d, x, w, bias, y, fuse_relu, workspace, workspace_bytes, stream);
}

extern "C" size_t zsv_linear_dgrad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features) {
    const zsv_conv_desc d = linear_desc(rows, in_features, out_features);
    return zsv_conv3d_dgrad_workspace_bytes(&d);
}

extern "C" int zsv_linear_dgrad(const float* dy, const float* w, float* dx, int32_t rows, int32_t in_features,
                                int32_t out_features, void* workspace, size_t workspace_bytes, void* stream) {
    const zsv_conv_desc d = linear_desc(rows, in_features, out_features);
    return zsv_conv3d_dgrad(&d, dy, w, dx, workspace, workspace_bytes, stream);
}


// ---- weight panels kept by the caller (zsv_hip.h) ------------------------------------------------------------------------------
// query / job run the ordinary entry point with the thread-local panel context set (conv_params.h): the path that would pack a
// panel reports its size / writes the pack launch down and returns before anything is launched; x / y / workspace are never touched.
namespace {
struct PanelScope {
    PanelScope(int mode, void* ptr, size_t bytes) { g_panel = PanelCtx{mode, ptr, bytes, 0, 0, {}}; }
    ~PanelScope() { g_panel.mode = PANEL_NONE; g_panel.ptr = nullptr; }
};
float* const kDummy = reinterpret_cast<float*>(0x1000);      // non-null, 16-byte aligned, never dereferenced in query / record mode
int panel_probe(const zsv_conv_desc* d, int32_t direction, int32_t extras, const float* w) {
    if (direction == 0)
        return zsv_conv3d_fwd_full(d, kDummy, w, extras ? kDummy : nullptr, nullptr, kDummy, 0, nullptr, 0, kDummy, ~(size_t)0 >> 1, nullptr);
    return zsv_conv3d_dgrad_add(d, kDummy, w, nullptr, kDummy, kDummy, ~(size_t)0 >> 1, nullptr);
}
}  // namespace

extern "C" int zsv_conv3d_panel_query(const zsv_conv_desc* d, int32_t direction, int32_t extras, size_t* panel_bytes) {
    if (!panel_bytes) return ZSV_E_NULL;
    *panel_bytes = 0;
    int st = conv_check(d);
    if (st) return st;
    if (direction != 0 && direction != 1) return ZSV_E_BAD_SHAPE;
    PanelScope scope(PANEL_QUERY, nullptr, 0);
    st = panel_probe(d, direction, extras, kDummy);
    if (st) return st;
    *panel_bytes = g_panel.jobs == 1 ? g_panel.need : 0;
    return ZSV_OK;
}

extern "C" int zsv_conv3d_panel_job(const zsv_conv_desc* d, int32_t direction, int32_t extras, const float* w, void* panel,
                                    size_t panel_bytes, zsv_pack_job* job) {
    if (!w || !panel || !job) return ZSV_E_NULL;
    int st = conv_check(d);
    if (st) return st;
    if (direction != 0 && direction != 1) return ZSV_E_BAD_SHAPE;
    PanelScope scope(PANEL_RECORD, panel, panel_bytes);
    st = panel_probe(d, direction, extras, w);
    if (st) return st;
    if (g_panel.jobs != 1) return ZSV_E_UNSUPPORTED;
    *job = g_panel.job;
    return ZSV_OK;
}

extern "C" int zsv_conv3d_fwd_full_panel(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                                         const float* residual, float* y, int fuse_relu, float* bn_partials, int32_t stat_tiles,
                                         void* workspace, size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes) {
    if (!panel) return ZSV_E_NULL;
    PanelScope scope(PANEL_LAUNCH_ONLY, const_cast<void*>(panel), panel_bytes);
    return zsv_conv3d_fwd_full(d, x, w, bias, residual, y, fuse_relu, bn_partials, stat_tiles, workspace, workspace_bytes, stream);
}

extern "C" int zsv_conv3d_fwd_pre_panel(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch,
                                        const float* w, float* y, float* bn_partials, int32_t stat_tiles, void* workspace,
                                        size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes) {
    if (!panel) return ZSV_E_NULL;
    PanelScope scope(PANEL_LAUNCH_ONLY, const_cast<void*>(panel), panel_bytes);
    return zsv_conv3d_fwd_pre(d, x, pre_coef, coef_pitch, w, y, bn_partials, stat_tiles, workspace, workspace_bytes, stream);
}

extern "C" int zsv_conv3d_dgrad_add_panel(const zsv_conv_desc* d, const float* dy, const float* w, const float* add, float* dx,
                                          void* workspace, size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes) {
    if (!panel) return ZSV_E_NULL;
    PanelScope scope(PANEL_LAUNCH_ONLY, const_cast<void*>(panel), panel_bytes);
    return zsv_conv3d_dgrad_add(d, dy, w, add, dx, workspace, workspace_bytes, stream);
}

extern "C" int zsv_conv3d_dgrad_add_strided_panel(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int32_t st,
                                                  int32_t sh, int32_t sw, float* dx, void* workspace, size_t workspace_bytes,
                                                  void* stream, const void* panel, size_t panel_bytes) {
    if (!panel) return ZSV_E_NULL;
    PanelScope scope(PANEL_LAUNCH_ONLY, const_cast<void*>(panel), panel_bytes);
    return zsv_conv3d_dgrad_add_strided(d, dy, w, sub, st, sh, sw, dx, workspace, workspace_bytes, stream);
}
