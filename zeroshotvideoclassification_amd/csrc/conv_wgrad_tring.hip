// conv_wgrad_tring.hip -- weight gradient of the temporal 3x1x1 stride-1 "same" convolutions (the second half of
// Conv2Plus1D, resnet.py:46-52) with a ring of dY frames in LDS, fp32.
//
//   dW[co][ci][kt] = sum_{n,t,pos} dY[co][n,t,pos] * X[ci][n, t+kt-1, pos]
//                  = sum_{n,t',pos} X[ci][n,t',pos] * dY[co][n, t'-(kt-1), pos]            (t' = t + kt - 1)
// A chunk is one frame t' of 16 consecutive (h,w) positions of one clip and the chunks walk the T frames of a position
// segment before moving to the next segment.  The three taps of a chunk then need the dY rows of frames t'+1, t', t'-1 at
// the same 16 positions: with a 4-slot ring of dY frames in LDS every chunk fetches ONE new dY frame (4 DMAs) and its own
// X rows (one DMA per 16 input channels) -- 13 DMAs per chunk on the 144 -> 64 layer instead of the 31 of conv_wgrad_dma.hip,
// whose 64-row layers are bound by DMA issue.  A frame outside the clip is a wave-uniform skip: no per-voxel masks.
//
// GEMM: rows = 16*TM input channels (the larger side: 144 = TM 9, 128 = TM 8), columns = (kt, 64 output channels) = 12 blocks
// of 16, three per wave; MFMA k order k = 4*(lane>>4) + step, so a lane reads its row's 4 voxels for the 4 steps with one
// ds_read_b128; LDS rows are 64 bytes with the 16-byte slot XOR-swizzled on the source side (conflict-free fragments, as in
// conv_wgrad_dma.hip).  The reduction is cut into slices (about one round of resident workgroups); slices write slabs
// [slice][ci][kt*M + co], wgrad_tring_sum_kernel adds them in a fixed order: bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"

namespace zsv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct WgradTringParams {
    int M, Cin;                       // output / input channels
    int S, HW, T, nseg;               // voxels per clip / frame, frames, 16-position segments per frame
    int chunks_total, chunks_per_slice;
    unsigned x_bytes, dy_bytes;
    int tiles_m, tiles_mn;
    const float* pre_coef;            // PRE: X is read as relu(X * scale[ci] + shift[ci]) -- [2][pre_pitch] (scale row, shift row)
    int pre_pitch;
};

__device__ __forceinline__ int tring_swz(int row) { return ((row >> 2) & 1) << 1; }

// PRE: the convolution's input is the output of BatchNorm + ReLU that was never materialised (zsv_bn_fwd_train_coeffs): a
// lane's X values all belong to its own row (input channel), so the affine + ReLU is two VALU ops per value with two
// registers per row block -- bit-identical to reading the materialised activation (same fmaf, same max).
template <int TM, bool PRE = false>
__global__ __launch_bounds__(256, 3) void conv_wgrad_tring_kernel(WgradTringParams prm, const float* __restrict__ X,
                                                                  const float* __restrict__ DY, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM;
    constexpr int NAW = (TM + 3) / 4;                       // X pieces (16 rows x 64 B) per wave and chunk
    constexpr int A_BYTES = TM * 1024;
    constexpr int RING_AT = 2 * A_BYTES;                    // 4 slots x 4 KiB: dY frames f & 3
    constexpr unsigned OOB = 0xFFFFFFF0u;                   // (+12 must not wrap)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_tile(gridDim.x, blockIdx.x);        // the tiles of one slice share an XCD (one L2)
    const int tile = lid % prm.tiles_mn, slice = lid / prm.tiles_mn;
    const int m0 = (tile % prm.tiles_m) * BM, co0 = (tile / prm.tiles_m) * 64;
    const int c0 = slice * prm.chunks_per_slice;
    const int nq = min(prm.chunks_per_slice, prm.chunks_total - c0);
    if (nq <= 0) return;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);

    // ---- DMA assignment: lane l of a piece fills row l/4, slot l%4 <- source slot (l%4) ^ swz(row) ----
    const int prow = lane >> 2, srcslot = ((lane & 3) ^ tring_swz(prow)) * 4;
    int a_off[NAW];                                         // byte offset at (clip 0, frame 0, segment 0), or -1
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int pa = wave + 4 * k;
        const int ci = m0 + pa * 16 + prow;
        a_off[k] = (pa < TM && ci < prm.Cin) ? 4 * (ci * prm.S + srcslot) : -1;
    }
    const int co_l = co0 + wave * 16 + prow;                // this wave's piece of a dY frame: output channels 16*wave ..
    const int b_off = co_l < prm.M ? 4 * (co_l * prm.S + srcslot) : -1;

    auto issue_x = [&](int buf, int n_img, int t, int seg) {
        const int base = 4 * (n_img * prm.Cin * prm.S + t * prm.HW + seg * 16);
#pragma unroll
        for (int k = 0; k < NAW; ++k)
            if (wave + 4 * k < TM)                              // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(lds + buf * A_BYTES + 1024 * (wave + 4 * k)), 16,
                                                         (int)(a_off[k] >= 0 ? (unsigned)(a_off[k] + base) : OOB), 0, 0, 0);
    };
    auto issue_dy = [&](int n_img, int f, int seg) {            // frame f of the segment -> ring slot f & 3
        const int base = 4 * (n_img * prm.M * prm.S + f * prm.HW + seg * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(lds + RING_AT + (f & 3) * 4096 + 1024 * wave), 16,
                                                 (int)(b_off >= 0 ? (unsigned)(b_off + base) : OOB), 0, 0, 0);
    };

    f32x4 acc[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    const int frag = r16 * 64 + ((g ^ tring_swz(r16)) << 4);
    float pre_sc[PRE ? TM : 1], pre_sh[PRE ? TM : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ci = m0 + 16 * i + r16;
            pre_sc[i] = ci < prm.Cin ? prm.pre_coef[ci] : 0.f;
            pre_sh[i] = ci < prm.Cin ? prm.pre_coef[prm.pre_pitch + ci] : 0.f;
        }
    }
    int kt_of[3], cob_of[3];                                // tap and 16-channel block of this wave's three column blocks
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cb = 3 * wave + j;
        kt_of[j] = cb >> 2;
        cob_of[j] = cb & 3;
    }

    // chunk c -> (clip, segment, frame): frames are the fastest index
    const int per_clip = prm.nseg * prm.T;
    int n_img = c0 / per_clip;
    int seg = (c0 - n_img * per_clip) / prm.T;
    int t = c0 - n_img * per_clip - seg * prm.T;

    issue_x(0, n_img, t, seg);
    if (t > 0) issue_dy(n_img, t - 1, seg);
    issue_dy(n_img, t, seg);
    if (t + 1 < prm.T) issue_dy(n_img, t + 1, seg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < nq; ++ch) {
        const int cur = ch & 1;
        // the next chunk: next frame of the segment, or frame 0 of the next segment / clip
        int nt = t + 1, nseg_ = seg, nn = n_img;
        if (nt == prm.T) {
            nt = 0;
            if (++nseg_ == prm.nseg) { nseg_ = 0; ++nn; }
        }
        if (ch + 1 < nq) {
            issue_x(cur ^ 1, nn, nt, nseg_);
            if (nt == 0) {                                      // new segment: frames 0 and 1 (slots 0, 1; this chunk reads 2, 3)
                issue_dy(nn, 0, nseg_);
                issue_dy(nn, 1, nseg_);
            } else if (nt + 1 < prm.T) {
                issue_dy(nn, nt + 1, nseg_);                    // slot (t+2) & 3: not one of (t-1, t, t+1) & 3
            }
        }
        const unsigned char* as = lds + cur * A_BYTES + frag;
        f32x4 bf[3];
        bool live[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = t - (kt_of[j] - 1);                    // dY frame paired with X frame t under tap kt
            live[j] = (unsigned)f < (unsigned)prm.T;              // wave-uniform
            bf[j] = *reinterpret_cast<const f32x4*>(lds + RING_AT + (f & 3) * 4096 + cob_of[j] * 1024 + frag);
        }
        f32x4 af[2];
        af[0] = *reinterpret_cast<const f32x4*>(as);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) af[(i + 1) & 1] = *reinterpret_cast<const f32x4*>(as + (i + 1) * 1024);
            if constexpr (PRE) {
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i & 1][s] = fmaxf(__fmaf_rn(af[i & 1][s], pre_sc[i], pre_sh[i]), 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (live[j]) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i & 1][s], bf[j][s], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t = nt; seg = nseg_; n_img = nn;
    }

    // partial slab of this slice: OUT[slice][ci][kt * M + co]; lane holds rows 4g..4g+3 of column r16
    float* out = OUT + (size_t)slice * prm.Cin * 3 * prm.M;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int co = co0 + cob_of[j] * 16 + r16;
        if (co >= prm.M) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + 16 * i + 4 * g + r;
                if (ci < prm.Cin) out[(size_t)ci * 3 * prm.M + kt_of[j] * prm.M + co] = acc[i][j][r];
            }
    }
#endif
}

// dW[co][ci][kt] from the slabs [slice][ci][kt * M + co]: 32 elements x 8 slice groups per block, fixed order
__global__ __launch_bounds__(256) void wgrad_tring_sum_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int M,
                                                              int Cin, int slices) {
    __shared__ float part[8][32];
    const size_t slab = (size_t)Cin * 3 * M;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t j0 = (size_t)blockIdx.x * 32; j0 < slab; j0 += (size_t)gridDim.x * 32) {
        const size_t j = j0 + e;
        const bool live = j < slab;
        float s = 0.f;
        if (live)
            for (int k = grp; k < slices; k += 8) s += slabs[(size_t)k * slab + j];
        part[grp][e] = s;
        __syncthreads();
        if (grp == 0 && live) {
            float v = part[0][e];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += part[q][e];
            const int co = (int)(j % M);
            const size_t rr = j / M;
            const int kt = (int)(rr % 3), ci = (int)(rr / 3);
            dw[((size_t)co * Cin + ci) * 3 + kt] = v;
        }
        __syncthreads();
    }
}

// ---- host side -----------------------------------------------------------------------------------
struct WgradTringPlan {
    int tm, tiles_m, tiles_n, slices, chunks_per_slice;
};

static WgradTringPlan wgrad_tring_plan(const zsv_conv_desc* d) {
    WgradTringPlan pl;
    const int C = d->Cin;
    const int p9 = (C + 143) / 144 * 144, p8 = (C + 127) / 128 * 128;
    pl.tm = p9 <= p8 ? 9 : 8;
    const int bm = 16 * pl.tm;
    pl.tiles_m = (C + bm - 1) / bm;
    pl.tiles_n = (d->Cout + 63) / 64;
    const long chunks = (long)d->N * d->Ti * (d->Hi * d->Wi / 16);
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    long resident = pl.tm == 9 ? 512 : 768;                // workgroups per round (3 fit a CU; measured: 2 per CU is the better fill for the 144-row tile)
    if (const char* e = getenv("ZSV_WGRAD_TRING_RESIDENT")) resident = atol(e) > 0 ? atol(e) : resident;
    // slices: MFMA time / fill of the rounds of resident workgroups + slab write / read, >= 32 chunks per slice
    const double t_mfma = 2.0 * (double)(pl.tiles_m * bm) * (double)(pl.tiles_n * 192) * (double)chunks * 16.0 / 1.1e14;
    const double t_slice = 2.0 * (double)C * 3.0 * d->Cout * sizeof(float) / 6.0e12;
    long max_sl = chunks / 32;
    if (max_sl < 1) max_sl = 1;
    if (max_sl > 4096) max_sl = 4096;
    long sl = 1;
    double best = 1e300;
    for (long c = 1; c <= max_sl; ++c) {
        const long wgs = tiles * c, rounds = (wgs + resident - 1) / resident;
        const double cost = t_mfma * (double)(rounds * resident) / (double)wgs + t_slice * (double)c;
        if (cost < best * 0.999) { best = cost; sl = c; }
    }
    if (const char* e = getenv("ZSV_WGRAD_TRING_SLICES")) sl = atol(e) > 0 ? atol(e) : 1;
    if (sl > chunks) sl = chunks;
    pl.chunks_per_slice = (int)((chunks + sl - 1) / sl);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

bool wgrad_tring_applicable(const zsv_conv_desc* d, const float* x, const float* dy) {
    if (getenv("ZSV_NO_WGRAD_TRING")) return false;
    if (d->kT != 3 || d->kH != 1 || d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT != 1 || d->pH != 0 || d->pW != 0)
        return false;
    if (d->Ti % 4 != 0 || (d->Hi * d->Wi) % 16 != 0) return false;     // ring slots f & 3; whole 16-position segments
    if (d->Cout < 32) return false;
    const long S = (long)d->Ti * d->Hi * d->Wi;
    if ((long)d->N * d->Cin * S >= (1L << 29) || (long)d->N * d->Cout * S >= (1L << 29)) return false;      // int byte offsets
    const int C = d->Cin, p9 = (C + 143) / 144 * 144, p8 = (C + 127) / 128 * 128, pc = p9 <= p8 ? p9 : p8;
    if (pc * 10 > C * 13) return false;                                 // row padding above 30 %
    const int pm = (d->Cout + 63) / 64 * 64;
    if (pm * 10 > d->Cout * 13) return false;
    if ((long)d->N * S < 16384) return false;
    if (x != nullptr && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return false;
    return true;
}

size_t wgrad_tring_workspace_bytes(const zsv_conv_desc* d) {
    const WgradTringPlan pl = wgrad_tring_plan(d);
    return (size_t)pl.slices * d->Cin * 3 * d->Cout * sizeof(float);
}

template <int TM, bool PRE>
static int wgrad_tring_launch(const WgradTringParams& p, int slices, hipStream_t stream, const float* x, const float* dy,
                              float* out) {
    constexpr int LDS_BYTES = 2 * TM * 1024 + 4 * 4096;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_tring_kernel<TM, PRE>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_tring_kernel<TM, PRE>), dim3((unsigned)(p.tiles_mn * slices)), dim3(256), LDS_BYTES, stream, p, x,
                       dy, out);
    return launch_status();
}

int wgrad_tring(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                hipStream_t stream) {
    return wgrad_tring_pre(d, x, nullptr, 0, dy, dw, workspace, workspace_bytes, stream);
}

// pre_coef != nullptr: x is the INPUT of a BatchNorm + ReLU whose output is the convolution's input (see the kernel's PRE)
int wgrad_tring_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int pre_pitch, const float* dy, float* dw,
                    void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const WgradTringPlan pl = wgrad_tring_plan(d);
    if (!workspace || workspace_bytes < wgrad_tring_workspace_bytes(d)) return ZSV_E_WORKSPACE;
    WgradTringParams p;
    p.M = d->Cout; p.Cin = d->Cin;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.T = d->Ti; p.nseg = p.HW / 16;
    p.chunks_total = d->N * p.T * p.nseg;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.x_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.S);
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.S);
    p.tiles_m = pl.tiles_m; p.tiles_mn = pl.tiles_m * pl.tiles_n;
    p.pre_coef = pre_coef; p.pre_pitch = pre_pitch;
    float* slabs = (float*)workspace;
    int st;
    if (pre_coef) st = pl.tm == 9 ? wgrad_tring_launch<9, true>(p, pl.slices, stream, x, dy, slabs)
                                  : wgrad_tring_launch<8, true>(p, pl.slices, stream, x, dy, slabs);
    else st = pl.tm == 9 ? wgrad_tring_launch<9, false>(p, pl.slices, stream, x, dy, slabs)
                         : wgrad_tring_launch<8, false>(p, pl.slices, stream, x, dy, slabs);
    if (st) return st;
    const long n = (long)d->Cin * 3 * d->Cout;
    long blocks = (n + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_tring_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs, dw, d->Cout,
                       d->Cin, pl.slices);
    return launch_status();
}

}  // namespace zsv
