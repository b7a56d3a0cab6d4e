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
#include "knobs.h"

namespace zsv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct WgradTringParams {
    int M, Cin;                       // output / input channels
    int S, HW, T, nseg;               // voxels per clip / frame, frames, 16-position segments per frame
    int chunks_total, chunks_per_slice;
    int slices;                       // Winograd form: slice s walks the segments s, s + slices, ... of the (clip, segment) list
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

// ================================================================================================
// The same weight gradient with the kt taps in Winograd form (the transpose of F(2,3) along T, as conv_wgrad_wino.hip does along
// W): for a frame pair (t, t+1) of one position, with y0, y1 = dY at the two frames and d0..d3 = X[t-1..t+2] (zero outside the clip)
//     A = (y0, y0+y1, y0-y1, y1)   V = (d0-d2, d1+d2, d2-d1, d1-d3)   Mi = sum Ai * Vi
//     dW[kt=0] = M0 + (M1+M2)/2    dW[kt=1] = (M1-M2)/2    dW[kt=2] = (M1+M2)/2 - M3
// 4 multiplies per pair and (co, ci) instead of 6.  A chunk is one frame PAIR of a 16-position segment; rows = input channels
// (TM blocks), columns = (point, 64 output channels) = 16 blocks, wave w = output-channel block w with its four points.  X frames
// live in a FIFO ring of 6 slots (4 in use + the next pair's 2, or the next segment's first 3 while the last pair of a segment
// runs: its frame T is padding and needs no slot), dY frame pairs in 2 x 2 slots.  Slabs [slice][point][ci][co], summed in a fixed
// order by wgrad_twino_sum_kernel, which also applies the output transform.  PRE as above.
template <int TM, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_twino_kernel(WgradTringParams prm, const float* __restrict__ X,
                                                                  const float* __restrict__ DY, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM;
    constexpr int NAW = (TM + 3) / 4;                       // X pieces (16 rows x 64 B) per wave and frame
    constexpr int XSLOT = TM * 1024;                        // one X frame of the segment
    constexpr int DY_AT = 6 * XSLOT;                        // 4 dY slots x 4 KiB: (chunk parity, frame of the pair)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_tile(gridDim.x, blockIdx.x);
    const int tile = lid % prm.tiles_mn, slice = lid / prm.tiles_mn;
    const int m0 = (tile % prm.tiles_m) * BM, co0 = (tile / prm.tiles_m) * 64;
    const int TP = prm.T >> 1;
    // A segment is 16 positions = 64 bytes per (channel, frame): half a 128-byte line.  Neighbouring segments go to neighbouring
    // slices (which run at the same time on one XCD) so that the two halves of a line are fetched together: with one slice walking
    // consecutive segments the second half came T/2 chunks later, after the line had left the L2 -- every byte was read twice from HBM
    // (FETCH_SIZE 1.85 GB for 0.92 GB of operands on the 144 -> 64 layer1 convolution).
    const int njobs = prm.chunks_total / TP;                    // (clip, segment) pairs
    int job = slice;
    const int nq = job < njobs ? ((njobs - job + prm.slices - 1) / prm.slices) * TP : 0;
    if (nq <= 0) return;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);

    const int prow = lane >> 2, srcslot = ((lane & 3) ^ tring_swz(prow)) * 4;
    int a_off[NAW];
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int pa = wave + 4 * k;
        const int ci = m0 + pa * 16 + prow;
        a_off[k] = (pa < TM && ci < prm.Cin) ? 4 * (ci * prm.S + srcslot) : -1;
    }
    const int co_l = co0 + wave * 16 + prow;
    const int b_off = co_l < prm.M ? 4 * (co_l * prm.S + srcslot) : -1;

    auto issue_x = [&](int slot, int n_img, int f, int seg) {       // frame f (0 <= f < T) of the segment -> ring slot
        const int base = 4 * (n_img * prm.Cin * prm.S + f * prm.HW + seg * 16);
#pragma unroll
        for (int k = 0; k < NAW; ++k)
            if (wave + 4 * k < TM)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(lds + slot * XSLOT + 1024 * (wave + 4 * k)), 16,
                                                         (int)(a_off[k] >= 0 ? (unsigned)(a_off[k] + base) : OOB), 0, 0, 0);
    };
    auto issue_dy = [&](int slot, int n_img, int f, int seg) {
        const int base = 4 * (n_img * prm.M * prm.S + f * prm.HW + seg * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(lds + DY_AT + slot * 4096 + 1024 * wave), 16,
                                                 (int)(b_off >= 0 ? (unsigned)(b_off + base) : OOB), 0, 0, 0);
    };
    auto ring = [](int h, int j) { const int s = h + j; return s >= 6 ? s - 6 : s; };

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

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

    // job -> (clip, segment); inside a job the frame pairs in order
    int n_img = job / prm.nseg;
    int seg = job - n_img * prm.nseg;
    int tp = 0;
    int head = 0;                                           // ring slot of frame 2*tp - 1 of the current pair

    // first pair: its (up to) four X frames and its dY pair
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 2 * tp - 1 + j;
        if (f >= 0 && f < prm.T) issue_x(j, n_img, f, seg);
    }
    issue_dy(0, n_img, 2 * tp, seg);
    issue_dy(1, n_img, 2 * tp + 1, seg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < nq; ++ch) {
        const int par = ch & 1;
        int ntp = tp + 1, nseg_ = seg, nn = n_img;
        if (ntp == TP) {
            ntp = 0;
            const int nj = job + prm.slices;
            nn = nj / prm.nseg;
            nseg_ = nj - nn * prm.nseg;
        }
        if (ch + 1 < nq) {
            if (ntp != 0) {                                     // same segment: frames 2*ntp+1, 2*ntp+2 behind the four in use
                issue_x(ring(head, 4), nn, 2 * ntp + 1, nseg_);
                if (2 * ntp + 2 < prm.T) issue_x(ring(head, 5), nn, 2 * ntp + 2, nseg_);
            } else {                                            // new segment: frames 0, 1, 2 (this pair's frame T has no slot)
                issue_x(ring(head, 3), nn, 0, nseg_);
                issue_x(ring(head, 4), nn, 1, nseg_);
                issue_x(ring(head, 5), nn, 2, nseg_);
            }
            issue_dy(2 * (par ^ 1), nn, 2 * ntp, nseg_);
            issue_dy(2 * (par ^ 1) + 1, nn, 2 * ntp + 1, nseg_);
        }
        const bool zero_d0 = tp == 0, zero_d3 = tp == TP - 1;   // frames -1 / T: padding
        // B fragments: this wave's 16 output channels at the pair's two frames -> the four points
        const unsigned char* bp = lds + DY_AT + (2 * par) * 4096 + wave * 1024 + frag;
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(bp), y1 = *reinterpret_cast<const f32x4*>(bp + 4096);
        const f32x4 bsum = y0 + y1, bdif = y0 - y1;
        const unsigned char* xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = lds + ring(head, j) * XSLOT + frag;
        f32x4 xf[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[0][j] = *reinterpret_cast<const f32x4*>(xs[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sl = i & 1;
            if (i + 1 < TM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[sl ^ 1][j] = *reinterpret_cast<const f32x4*>(xs[j] + (i + 1) * 1024);
            }
            f32x4 d0 = xf[sl][0], d1 = xf[sl][1], d2 = xf[sl][2], d3 = xf[sl][3];
            if constexpr (PRE) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    d0[s] = fmaxf(__fmaf_rn(d0[s], pre_sc[i], pre_sh[i]), 0.f);
                    d1[s] = fmaxf(__fmaf_rn(d1[s], pre_sc[i], pre_sh[i]), 0.f);
                    d2[s] = fmaxf(__fmaf_rn(d2[s], pre_sc[i], pre_sh[i]), 0.f);
                    d3[s] = fmaxf(__fmaf_rn(d3[s], pre_sc[i], pre_sh[i]), 0.f);
                }
            }
            if (zero_d0) d0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (zero_d3) d3 = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[s], y0[s], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[s], bsum[s], acc[i][1], 0, 0, 0);
                acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2[s], bdif[s], acc[i][2], 0, 0, 0);
                acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3[s], y1[s], acc[i][3], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        head = ring(head, 2);
        if (ntp == 0) job += prm.slices;
        tp = ntp; seg = nseg_; n_img = nn;
    }

    // partial slab of this slice: OUT[slice][point][ci][co]; lane holds rows 4g..4g+3 (ci) of column r16 (co)
    const int co = co0 + wave * 16 + r16;
    if (co < prm.M) {
        float* out = OUT + (size_t)slice * 4 * prm.Cin * prm.M;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = m0 + 16 * i + 4 * g + r;
                    if (ci < prm.Cin) out[((size_t)p * prm.Cin + ci) * prm.M + co] = acc[i][p][r];
                }
    }
#endif
}

// dW[co][ci][kt] from the slabs [slice][point][ci][co]: fixed summation order (8 slice groups, then the groups), then the
// output transform of the Winograd form
__global__ __launch_bounds__(256) void wgrad_twino_sum_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int M,
                                                              int Cin, int slices) {
    __shared__ float part[8][4][32];
    const size_t plane = (size_t)Cin * M;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t j0 = (size_t)blockIdx.x * 32; j0 < plane; j0 += (size_t)gridDim.x * 32) {
        const size_t j = j0 + e;
        const bool live = j < plane;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            // (the loads of four slices are issued before their adds -- same order of addition, same bits: one slice at a time the
            // loop waited out an HBM round trip per slice, 86 us for the 512 slices of a layer1 gradient)
            int k = grp;
            for (; k + 24 < slices; k += 32) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[u][p] = slabs[((size_t)(k + 8 * u) * 4 + p) * plane + j];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < 4; ++p) s[p] += v[u][p];
            }
            for (; k < slices; k += 8)
#pragma unroll
                for (int p = 0; p < 4; ++p) s[p] += slabs[((size_t)k * 4 + p) * plane + j];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) part[grp][p][e] = s[p];
        __syncthreads();
        if (grp == 0 && live) {
            float Mv[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float t = part[0][p][e];
#pragma unroll
                for (int q = 1; q < 8; ++q) t += part[q][p][e];
                Mv[p] = t;
            }
            const int co = (int)(j % M), ci = (int)(j / M);
            float* o = dw + ((size_t)co * Cin + ci) * 3;
            const float h = 0.5f * (Mv[1] + Mv[2]);
            o[0] = Mv[0] + h;
            o[1] = 0.5f * (Mv[1] - Mv[2]);
            o[2] = h - Mv[3];
        }
        __syncthreads();
    }
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
        if (live) {
            int k = grp;
            for (; k + 24 < slices; k += 32) {                    // four loads in flight, added in the same order
                const float v0 = slabs[(size_t)k * slab + j], v1 = slabs[(size_t)(k + 8) * slab + j];
                const float v2 = slabs[(size_t)(k + 16) * slab + j], v3 = slabs[(size_t)(k + 24) * slab + j];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; k < slices; k += 8) s += slabs[(size_t)k * slab + j];
        }
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
    if (const char* e = ZSV_KNOB(WGRAD_TRING_RESIDENT)) resident = atol(e) > 0 ? atol(e) : resident;
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
    if (const char* e = ZSV_KNOB(WGRAD_TRING_SLICES)) sl = atol(e) > 0 ? atol(e) : 1;
    if (sl > chunks) sl = chunks;
    pl.chunks_per_slice = (int)((chunks + sl - 1) / sl);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

bool wgrad_tring_applicable(const zsv_conv_desc* d, const float* x, const float* dy) {
    if (ZSV_KNOB(NO_WGRAD_TRING)) return false;
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

// the Winograd form (conv_wgrad_twino_kernel): chunks are frame pairs, two workgroups per CU
static bool wgrad_twino_shape(const zsv_conv_desc* d) { return d->Ti >= 8 && ZSV_KNOB(NO_WGRAD_TWINO) == nullptr; }
static WgradTringPlan wgrad_twino_plan(const zsv_conv_desc* d) {
    WgradTringPlan pl;
    const int C = d->Cin;
    const int p9 = (C + 143) / 144 * 144, p8 = (C + 127) / 128 * 128;
    pl.tm = p9 <= p8 ? 9 : 8;
    const int bm = 16 * pl.tm;
    pl.tiles_m = (C + bm - 1) / bm;
    pl.tiles_n = (d->Cout + 63) / 64;
    const long chunks = (long)d->N * (d->Ti / 2) * (d->Hi * d->Wi / 16);
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    long resident = 512;
    if (const char* e = ZSV_KNOB(WGRAD_TRING_RESIDENT)) resident = atol(e) > 0 ? atol(e) : resident;
    // slices: MFMA time / fill of the rounds of resident workgroups + slab write / read, >= 16 chunks per slice
    const double t_mfma = 2.0 * (double)(pl.tiles_m * bm) * (double)(pl.tiles_n * 256) * (double)chunks * 16.0 / 1.1e14;
    const double t_slice = 2.0 * (double)C * 4.0 * d->Cout * sizeof(float) / 6.0e12;
    long max_sl = chunks / 16;
    if (max_sl < 1) max_sl = 1;
    if (max_sl > 4096) max_sl = 4096;
    long sl = 1;
    double best = 1e300;
    for (long c = 1; c <= max_sl; ++c) {
        const long wgs = tiles * c, rounds = (wgs + resident - 1) / resident;
        const double cost = t_mfma * (double)(rounds * resident) / (double)wgs + t_slice * (double)c;
        if (cost < best * 0.999) { best = cost; sl = c; }
    }
    if (const char* e = ZSV_KNOB(WGRAD_TRING_SLICES)) sl = atol(e) > 0 ? atol(e) : 1;
    // the slices take whole (clip, segment) jobs, job j to slice j % slices (see the kernel): no slice without a job
    const long jobs = (long)d->N * (d->Hi * d->Wi / 16);
    if (sl > jobs) sl = jobs;
    pl.slices = (int)sl;
    pl.chunks_per_slice = (int)(((jobs + sl - 1) / sl) * (d->Ti / 2));
    return pl;
}

size_t wgrad_tring_workspace_bytes(const zsv_conv_desc* d) {
    const WgradTringPlan pl = wgrad_tring_plan(d);
    size_t need = (size_t)pl.slices * d->Cin * 3 * d->Cout * sizeof(float);
    if (d->Ti >= 8) {                                       // (the Winograd form may be chosen at call time)
        const WgradTringPlan pw = wgrad_twino_plan(d);
        const size_t b = (size_t)pw.slices * d->Cin * 4 * d->Cout * sizeof(float);
        if (b > need) need = b;
    }
    return need;
}

template <int TM, bool PRE>
static int wgrad_twino_launch(const WgradTringParams& p, int slices, hipStream_t stream, const float* x, const float* dy,
                              float* out) {
    constexpr int LDS_BYTES = 6 * TM * 1024 + 4 * 4096;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_twino_kernel<TM, PRE>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_wgrad_twino_kernel<TM, PRE>), dim3((unsigned)(p.tiles_mn * slices)), dim3(256), LDS_BYTES, stream, p, x,
                       dy, out);
    return launch_status();
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
    const bool twino = wgrad_twino_shape(d);
    const WgradTringPlan pl = twino ? wgrad_twino_plan(d) : wgrad_tring_plan(d);
    if (!workspace || workspace_bytes < wgrad_tring_workspace_bytes(d)) return ZSV_E_WORKSPACE;
    WgradTringParams p;
    p.M = d->Cout; p.Cin = d->Cin;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.T = d->Ti; p.nseg = p.HW / 16;
    p.chunks_total = d->N * (twino ? p.T / 2 : p.T) * p.nseg;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.slices = pl.slices;
    p.x_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.S);
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.S);
    p.tiles_m = pl.tiles_m; p.tiles_mn = pl.tiles_m * pl.tiles_n;
    p.pre_coef = pre_coef; p.pre_pitch = pre_pitch;
    float* slabs = (float*)workspace;
    int st;
    if (twino) {
        if (pre_coef) st = pl.tm == 9 ? wgrad_twino_launch<9, true>(p, pl.slices, stream, x, dy, slabs)
                                      : wgrad_twino_launch<8, true>(p, pl.slices, stream, x, dy, slabs);
        else st = pl.tm == 9 ? wgrad_twino_launch<9, false>(p, pl.slices, stream, x, dy, slabs)
                             : wgrad_twino_launch<8, false>(p, pl.slices, stream, x, dy, slabs);
        if (st) return st;
        const long n = (long)d->Cin * d->Cout;
        long blocks = (n + 31) / 32;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(wgrad_twino_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)slabs, dw, d->Cout,
                           d->Cin, pl.slices);
        return launch_status();
    }
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
