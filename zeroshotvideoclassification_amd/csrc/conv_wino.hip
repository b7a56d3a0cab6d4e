// conv_wino.hip -- 1x3x3 / 3x3x3 stride-1 "same" convolution with the kw taps in Winograd F(2,3) form (fp32).
//
//   out[m][t,h,w] = sum_{c,kh,kw} G[m][c][kh][kw] * in[c][t, h+kh-1, w+kw-1]
// (forward and input gradient of Conv2Plus1D's spatial convolution, resnet.py:40-45 -- for the gradient in = dy,
// m = input channel, G = the weights with the tap axes flipped; with kT = 3 the (kt, kh) pairs take the place of
// kh: Conv3DSimple resnet.py:23-30, C3D network.py:102-117).  Along W two adjacent outputs share their
// inputs: with d0..d3 = in[w-1..w+2] of one (c, kh) row and g0..g2 the three kw weights,
//     V0 = d0-d2   V1 = d1+d2   V2 = d2-d1   V3 = d1-d3
//     U0 = g0      U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2
//     y(w) = M0+M1+M2,   y(w+1) = M1-M2-M3,   Mi = sum_{c,kh} Ui * Vi
// i.e. 4 multiplies per output pair instead of 6: 1.5x fewer MFMAs, still fp32 in / fp32 accumulate
// (the only rounding added is the 3-term sums of U, ~1e-7 relative).
//
// Implicit GEMM per point i: rows = 64 or 48 output channels (4 / 3 blocks), columns = 128 voxel PAIRS per
// workgroup (4 waves x 2 blocks of 16 pairs = 256 consecutive voxels; W is even, so a pair never
// straddles a row), K = (16-channel block, row tap).  Per chunk the four U panels [rows][16 k] and ONE raw input
// image [16][256 + 2 halo] are staged by LDS-DMA (16-byte pieces when W % 4 == 0); the V transform happens on
// the B fragment in registers (an aligned ds_read_b64 + 2 single reads + 4 VALU give the fragments of all four
// points), so the input is gathered once per (block, row tap) instead of once per tap.  MFMA k order k = 4*(lane>>4) + step:
// a lane's U values for the 4 steps of a chunk are one ds_read_b128 per (point, row block).  Row ends (w-1 < 0,
// w+2 >= W) zero d0 / d3 per lane.  Epilogue: output transform, then BatchNorm partial statistics / + add / + bias /
// ReLU as the caller asks.  Problems with fewer tiles than one round of workgroups run in 2-4 K parts (slabs summed in
// order by splitk_reduce).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "conv_params.h"
#include "zsv_common.h"
#include "zsv_hip.h"
#include "knobs.h"
#include "pack_bodies.h"

namespace zsv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WinoParams {
    int M, Mp;              // output channels, padded to 64
    int C, nblk;            // reduction channels, 16-channel blocks
    int S, HW, W, H, T;     // voxels per clip / frame, row length, rows, frames
    int kT, R;              // temporal taps (1 or 3, pad kT/2), row taps R = 3*kT (a row tap = one (kt, kh))
    int P;                  // N * S
    Magic m_S, m_HW, m_W, m_TH, m_H, m_tiles_m, m_ksplit;    // divisions of the F(4,3) kernel's prologue (conv_wino4_kernel)
    int wv_log2;
    int Wv, Pv;             // virtual-width form (conv_wino4_kernel VW): rows padded to Wv = 8 / 16 / 32 / 64 voxels, Pv = N * T * H * Wv
    unsigned in_bytes, u_bytes;   // bytes of the input tensor / of the transformed-weight array
    int tiles_m, tiles_n;
    int ksplit, chunks_per_split;   // > 1: K is cut into parts, part s writes its raw partial result to OUT + s * slab_elems
    long slab_elems;
    const float* add;       // != nullptr: out = act(result + add + bias) (shortcut gradient / inference residual)
    const float* bias;      // != nullptr: + bias[m]
    int relu;
    float* stat_sum;        // != nullptr: per (channel, column tile) sum and sum of squares of the raw result
    float* stat_sq;         //   [M][tiles_n] each (BatchNorm statistics of a training forward)
    // temporal form (conv_winot_kernel): PW = 256 / T positions per tile, segs = tiles per clip; PRE: the input is read as
    // relu(in * scale[c] + shift[c]) with pre_coef = [2][pre_pitch] (see conv_tap.hip PRE / zsv_bn_fwd_train_coeffs)
    int PW, segs;
    int pw_log2;            // log2(PW) (PW = 256 / T is 16, 32 or 64)
    unsigned out_bytes;     // bytes of the produced tensor (buffer-addressed stores of the temporal F(4,3) kernel)
    const float* pre_coef;
    int pre_pitch;
};

// Up[(cb*R + r)*4 + pt][Mp][c%16] from G[m][c][r][kw] = W[m*sm + c*sc + (flip ? 3R-1 - (3*r+kw) : 3*r+kw)], r = kt*3 + kh
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ W, float* __restrict__ Up, int M, int Mp,
                                                        int C, int nblk, int R, long sm, long sc, int flip, long total) {
    const PackWinoArgs a = {M, Mp, C, nblk, R, flip, sm, sc};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) Up[i] = pack_wino_value<4>(a, W, i);
}

// TM = 16-row blocks per workgroup (rows = 16*TM: 64 for the gradients, 48 for the 144- and 288-channel forwards);
// NCHUNKS = K chunks when known at compile time (12: the 64-channel layer1 forward, which so has its own symbol), 0 = runtime;
// X4 = the raw image is staged by 16-byte DMAs (W % 4 == 0: the 4 voxels of a piece share their row, so one validity bit and
// one 16-byte-aligned address serve them): 4 instead of 16 activation DMAs per wave and chunk
template <int TM, int NCHUNKS, bool X4>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoParams prm, const float* __restrict__ Up,
                                                           const float* __restrict__ IN, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM, BN = 256, BK = 16;
    // MFMA k order k = 4*(lane>>4) + step: the U image is [point][row][16 k] (64-byte rows, 16-byte slot XOR-swizzled on the
    // source side), so a lane reads its row's k values for the 4 steps of a chunk with ONE ds_read_b128 per (point, row block)
    constexpr int LDB = 264;                // 258 (261) used; 4 * LDB = 32 mod 64: the k rows 4g+s of the two lane halves of a ds_read_b64 split the banks
    constexpr int C0 = X4 ? 4 : 1;          // image column of the tile's first voxel (16-byte aligned for the 16-byte DMAs); halo at C0-1, C0+256
    constexpr int A_FLOATS = 4 * BM * BK, B_FLOATS = BK * LDB;
    constexpr int HALO_AT = A_FLOATS + B_FLOATS;      // 64 floats of scratch: where the halo DMA lands
    constexpr int STAGE = HALO_AT + 64;
    constexpr unsigned OOB = 0xFFFFFFFFu, OOB16 = 0xFFFFFFF0u;      // (+12 must not wrap)
    extern __shared__ __attribute__((aligned(16))) float pool[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_tile(gridDim.x, blockIdx.x);       // the K parts and row tiles of one column tile are neighbours (one L2)
    const int split = lid % prm.ksplit, tile = lid / prm.ksplit;
    const int m0 = (tile % prm.tiles_m) * BM;
    const int n0 = (tile / prm.tiles_m) * BN;
    if (prm.ksplit > 1) OUT += (size_t)split * prm.slab_elems;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN), 0, prm.in_bytes, 0x00020000);
    const int ch_bytes = 4 * prm.S;

    // ---- this lane's voxel(s) of the raw image: column C0 + 64*wave + lane (X4: C0 + 4*lane .. + 3, every k row) ----
    auto decode = [&](int p, int& base_bytes, unsigned& hmask) {
        base_bytes = 0;
        hmask = 0;
        if (p >= 0 && p < prm.P) {
            const int n = p / prm.S;
            int r = p - n * prm.S;
            const int t = r / prm.HW;
            r -= t * prm.HW;
            const int h = r / prm.W;
            base_bytes = 4 * (n * prm.C * prm.S + (p - n * prm.S));
            for (int kt = 0; kt < prm.kT; ++kt)
                for (int kh = 0; kh < 3; ++kh)
                    hmask |= (unsigned)((unsigned)(h + kh - 1) < (unsigned)prm.H && (unsigned)(t + kt - prm.kT / 2) < (unsigned)prm.T) << (kt * 3 + kh);
        }
    };
    int base_bytes;
    unsigned hmask;
    decode(X4 ? n0 + 4 * lane : n0 + 64 * wave + lane, base_bytes, hmask);
    // halo columns C0-1 and C0+256 (voxels n0-1, n0+256): wave 0, lanes 0..31 = (k row, side)
    int halo_base = 0;
    unsigned halo_mask = 0;
    if (wave == 0) decode((lane & 1) ? n0 + BN : n0 - 1, halo_base, halo_mask);

    // U panels: 4*TM pieces of 16 rows x 64 B; lane l of a piece fills row l/4, slot l%4 <- source slot (l%4) ^ swz(row)
    constexpr int APASS = TM;                         // pieces per wave: piece q = wave + 4*j = (point q / TM, row block q % TM)
    const float* a_src[APASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
        const int q = wave + 4 * j, pt = q / TM, ib = q % TM, row = lane >> 2;
        const int sw = ((lane & 3) ^ (((row >> 2) & 1) << 1)) * 4;
        a_src[j] = Up + ((size_t)pt * prm.Mp + m0 + 16 * ib + row) * 16 + sw;
    }
    const size_t a_chunk_stride = (size_t)4 * BK * prm.Mp;

    const int nchunks_all = NCHUNKS > 0 ? NCHUNKS : prm.nblk * prm.R;
    const int c_first = NCHUNKS > 0 ? 0 : split * prm.chunks_per_split;                        // this part's chunks
    const int nchunks = NCHUNKS > 0 ? NCHUNKS : min(prm.chunks_per_split, nchunks_all - c_first);
    int ld_cb = c_first / prm.R, ld_kt = (c_first % prm.R) / 3, ld_kh = c_first % 3;           // (R = 3 * kT)
    auto issue = [&](int chunk, int buf) {
        float* as = pool + buf * STAGE;
        float* bs = as + A_FLOATS;
        const int toff = 4 * ((ld_kh - 1) * prm.W + (ld_kt - prm.kT / 2) * prm.HW);
        const int ld_r = ld_kt * 3 + ld_kh;
        const unsigned ok = (hmask >> ld_r) & 1u;
        const int ci0 = ld_cb * 16;
        if constexpr (X4) {
            const unsigned voff = ok ? (unsigned)(base_bytes + toff) : OOB16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * wave + j, ci = ci0 + k;                   // wave-uniform row
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LDB + C0), 16, (int)(ci < prm.C ? voff : OOB16),
                                                         ci < prm.C ? ci * ch_bytes : 0, 0, 0);
            }
        } else {
            const unsigned voff = (unsigned)(base_bytes + toff) | (ok - 1u);
#pragma unroll
            for (int k = 0; k < BK; ++k) {
                const int ci = ci0 + k;
                const unsigned v = ci < prm.C ? voff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LDB + C0 + 64 * wave), 4, (int)v,
                                                         ci < prm.C ? ci * ch_bytes : 0, 0, 0);
            }
        }
        if (wave == 0) {
            // one instruction: lanes 0..31 -> (k = lane/2, side = lane&1); the destination must be lane-linear,
            // so the 32 halo values land in a scratch row and are moved by 32 lanes after the wait
            const int k = lane >> 1;
            const int ci = ci0 + k;
            const unsigned hok = (halo_mask >> ld_r) & 1u;
            unsigned hv = (unsigned)(halo_base + toff) | (hok - 1u);
            if (lane >= 32 || ci >= prm.C) hv = OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(as + HALO_AT), 4, (int)(hv + (hv == OOB ? 0u : (unsigned)(ci * ch_bytes))), 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < APASS; ++j)
            __builtin_amdgcn_global_load_lds(a_src[j] + (size_t)chunk * a_chunk_stride, (lds_ptr_t)(as + 256 * (wave + 4 * j)), 16, 0, 0);
        if (++ld_kh == 3) {
            ld_kh = 0;
            if (++ld_kt == prm.kT) { ld_kt = 0; ++ld_cb; }
        }
    };
    // after the DMAs of a stage have landed: scatter the 32 halo values into columns 0 / 257 of their rows
    auto place_halo = [&](int buf) {
        if (wave == 0 && lane < 32) {
            float* bs = pool + buf * STAGE + A_FLOATS;
            const float v = pool[buf * STAGE + HALO_AT + lane];
            bs[(lane >> 1) * LDB + ((lane & 1) ? C0 + BN : C0 - 1)] = v;
        }
    };

    f32x4 acc[4][TM][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    // row-end masks of this lane's two pairs: d0 is outside if the pair starts a row, d3 if it ends one
    bool zero_d0[2], zero_d3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int v = n0 + 64 * wave + 2 * (16 * j + r16);          // first voxel of the pair
        const int w = v % prm.W;
        zero_d0[j] = w == 0;
        zero_d3[j] = w + 2 >= prm.W;
    }

    issue(c_first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave 0 moves the halo values its own DMA fetched
    place_halo(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) issue(c_first + ch + 1, cur ^ 1);
        const float* as = pool + cur * STAGE;
        const float* bs = as + A_FLOATS;
        // U fragments of the whole chunk up front (one ds_read_b128 each); image fragments of k-step s+1 are fetched before the
        // MFMA burst of step s
        f32x4 a4[4][TM];
        const int a_frag = r16 * 16 + ((g ^ (((r16 >> 2) & 1) << 1)) << 2);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[p][i] = *reinterpret_cast<const f32x4*>(as + (p * TM + i) * 256 + a_frag);
        f32x2 lo[2][2], hi[2][2];
        auto fetch = [&](int s, int slot) {
            const int krow = 4 * g + s;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* src = bs + krow * LDB + (C0 - 1) + 64 * wave + 2 * (16 * j + r16);       // image column of d0
                if constexpr (X4) {                       // d0 sits at an odd column: aligned (d1, d2) + two single reads
                    const f32x2 mid = *reinterpret_cast<const f32x2*>(src + 1);
                    lo[slot][j] = f32x2{src[0], mid[0]};
                    hi[slot][j] = f32x2{mid[1], src[3]};
                } else {
                    lo[slot][j] = *reinterpret_cast<const f32x2*>(src);
                    hi[slot][j] = *reinterpret_cast<const f32x2*>(src + 2);
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = s & 1;
            float v[4][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float d0 = zero_d0[j] ? 0.f : lo[sl][j][0], d1 = lo[sl][j][1], d2 = hi[sl][j][0], d3 = zero_d3[j] ? 0.f : hi[sl][j][1];
                v[0][j] = d0 - d2; v[1][j] = d1 + d2; v[2][j] = d2 - d1; v[3][j] = d1 - d3;
            }
            if (s < 3) fetch(s + 1, sl ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[p][i][s], v[p][j], acc[p][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ch + 1 < nchunks) place_halo(cur ^ 1);
        __syncthreads();
    }

    // ---- output transform (+ statistics) (+ add, bias, ReLU) + store: lane holds rows 4g..4g+3 of pair column r16
    const bool stats = prm.stat_sum != nullptr;
    float* red = pool;                                   // [4 waves][BM][2] partial sums (the staging LDS is free now)
    if (stats) __syncthreads();
    // element offset of this lane's two pairs inside row 0 of their clip (one division per pair, not per stored value;
    // M * P < 2^30: int offsets)
    int pair_off[2];
    bool pair_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int v0 = n0 + 64 * wave + 2 * (16 * j + r16);
        pair_ok[j] = v0 < prm.P;
        const int n = v0 / prm.S;
        pair_off[j] = n * prm.M * prm.S + (v0 - n * prm.S);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * g + r;
            const int row_off = m * prm.S;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float M0 = acc[0][i][j][r], M1 = acc[1][i][j][r], M2 = acc[2][i][j][r], M3 = acc[3][i][j][r];
                f32x2 y = {(M0 + M1) + M2, (M1 - M2) - M3};
                if (pair_ok[j] && m < prm.M) {
                    s1 += y[0] + y[1];
                    s2 += y[0] * y[0] + y[1] * y[1];
                    const int off = pair_off[j] + row_off;
                    if (prm.add != nullptr) y += *reinterpret_cast<const f32x2*>(prm.add + off);
                    if (prm.bias != nullptr) y += prm.bias[m];
                    if (prm.relu) { y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); }
                    *reinterpret_cast<f32x2*>(OUT + off) = y;
                }
            }
            if (stats) {                                  // 16 lanes (r16) share row m
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (r16 == 0) {
                    red[(wave * BM + 16 * i + 4 * g + r) * 2] = s1;
                    red[(wave * BM + 16 * i + 4 * g + r) * 2 + 1] = s2;
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (tid < BM && m0 + tid < prm.M) {               // the 4 waves' partials, in wave order
            const float t1 = (red[tid * 2] + red[(BM + tid) * 2]) + (red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2]);
            const float t2 = (red[tid * 2 + 1] + red[(BM + tid) * 2 + 1]) + (red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1]);
            const int tn = n0 / BN;
            prm.stat_sum[(size_t)(m0 + tid) * prm.tiles_n + tn] = t1;
            prm.stat_sq[(size_t)(m0 + tid) * prm.tiles_n + tn] = t2;
        }
    }
#endif
}

// ================================================================================================
// F(4,3) form of the same convolution, for W % 4 == 0: four adjacent outputs along W share six inputs d0..d5 =
// in[w-1..w+4] of one (c, row tap) row; with g0..g2 the kw weights
//     V0 = 4d0 - 5d2 + d4     V1 = (d3 + d4) - 4(d1 + d2)   V2 = (d4 - d3) + 4(d1 - d2)
//     V3 = (d4 - d2) + 2(d3 - d1)   V4 = (d4 - d2) - 2(d3 - d1)   V5 = 4d1 - 5d3 + d5
//     U0 = g0/4   U1 = -(g0+g1+g2)/6   U2 = -(g0-g1+g2)/6   U3 = g0/24 + g1/12 + g2/6   U4 = g0/24 - g1/12 + g2/6   U5 = g2
//     y(w) = M0+M1+M2+M3+M4   y(w+1) = (M1-M2) + 2(M3-M4)   y(w+2) = (M1+M2) + 4(M3+M4)   y(w+3) = (M1-M2) + 8(M3-M4) + M5
// 6 multiplies per 4 outputs instead of 12: 2x fewer MFMAs than the direct form (F(2,3): 1.5x), still fp32 in / fp32
// accumulate.  The transforms' constants cost accuracy: measured max error 1.1e-6 of the output range against 4e-7 for the
// direct and the F(2,3) forms (K = 576), tested at 1e-5.
// Shape: rows = 16*TM output channels, columns = 64 voxel QUADS per workgroup (one 16-quad block per wave = the same 256
// consecutive voxels as above), K = (16-channel block, row tap); per chunk the six U panels [rows][16 k] and one raw image
// [16][256 + 2 halo] arrive by LDS-DMA exactly as above.  A lane reads (d1..d4) as one aligned ds_read_b128 plus d0, d5.
// With 64 rows the two stages of U panels do not fit twice per CU (2 x 82 KB), so there the U panel is single-buffered:
// a second barrier per chunk separates the fragment reads of a chunk from the DMAs of the next one.
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ W, float* __restrict__ Up, int M, int Mp,
                                                         int C, int nblk, int R, long sm, long sc, int flip, long total) {
    const PackWinoArgs a = {M, Mp, C, nblk, R, flip, sm, sc};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) Up[i] = pack_wino_value<6>(a, W, i);
}

// Phase timestamps of conv_wino4_kernel / conv_winot4_kernel for tools/winot_trace.py (a variant build: tools/variant.sh trace conv_wino.hip
// -DZSV_WINOT_TRACE): wave 0 of every workgroup writes s_memtime at six points + its hardware id behind the output tensor (the
// script allocates the room).  Never defined in the shipped library.
// ZSV_WINOT_ABLATE (variant builds only, wrong results): 1 = no output stores, 2 = no image DMAs, 3 = no MFMAs, 4 = no U DMAs
#ifndef ZSV_WINOT_ABLATE
#define ZSV_WINOT_ABLATE 0
#endif
#ifdef ZSV_WINOT_TRACE
#define ZSV_TRACE_MARK(i) do { if (tid == 0) trace_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define ZSV_TRACE_LAP(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); trace_lap[i] += now_ - trace_last; trace_last = now_; } } while (0)
#else
#define ZSV_TRACE_MARK(i) do { } while (0)
#define ZSV_TRACE_LAP(i) do { } while (0)
#endif

#ifndef ZSV_DMA_EVERY
#define ZSV_DMA_EVERY 6          // MFMAs between two DMA instructions of the next chunk (conv_wino4_kernel / conv_winot4_kernel)
#endif
#ifndef W4ABL
#define W4ABL 0        // timing-only ablation builds (tools/variant.sh): 1 no DMAs after the first chunk, 2 no fragment reads, 4 no chunk-end wait / barrier, 8 no stores
#endif
// VW ("virtual width"): rows whose length W is NOT a multiple of 4 (layer3's 14, layer4's 7: resnet.py:217-220 on 112x112 clips)
// are tiled as if they were Wv = 8 / 16 / 32 / 64 voxels long (W > Wv - 4, so every quad starts inside its row): a column of the
// tile is a VIRTUAL voxel (row, wv), its 16-byte piece is fetched from the real address of (row, wv) -- only 4-byte aligned, and
// a row's last piece carries 1-3 values of the next row, which the lane zeroes in registers (`nvalid`) -- and the epilogue stores
// only the real voxels.  256 % Wv == 0: tiles and the waves' 64-voxel spans begin and end on row boundaries, so no halo voxels
// exist (d0 of a row's first quad and d5 of its last are the zero padding).  7-wide rows run F(4,3) on 8 columns: 12 multiplies
// per 7 outputs against 21 (1.75x fewer MFMAs than the direct kernel they used to take); 14-wide rows 24 per 14 against the
// F(2,3) kernel's 28, with 16-byte image DMAs instead of 4-byte ones.
// EPI: the epilogue, chosen at compile time (as conv_winot4_kernel): bit 0 = BatchNorm partial statistics, bit 1 = + add (the
// identity-shortcut gradient of zsv_conv3d_dgrad_add), 8 = relu(result + bias[m]) (C3D's forward convolutions, folded-BatchNorm
// inference without a residual), 4 = the run-time-flag form (every other bias / ReLU / residual combination).
// EPI < 4 is branch-free: 16-byte buffer stores (and loads of `add`) with scalar row offsets, lanes outside the problem dropped by
// the descriptor's range check.
template <int TM, int NCHUNKS, bool VW = false, int EPI = 4>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(WinoParams prm, const float* __restrict__ Up,
                                                            const float* __restrict__ IN, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM, BN = 256, BK = 16, NP = 6;
    constexpr bool USINGLE = TM >= 4;
    constexpr int LDB = 272, C0 = 4;           // image row: left halo at column 3, the 256 voxels at 4..259, right halo at 260; 4 * LDB = 0 mod 64:
                                               // the k rows 4g+s of the lane groups of a ds_read_b128 keep their own 16-byte slots (pitch 264: 2-way conflicts)
    constexpr int A_FLOATS = NP * BM * BK, B_FLOATS = BK * LDB;
    constexpr int IMG = B_FLOATS + 64;         // image + 64 floats of scratch where the halo DMA lands
    constexpr int STAGE = A_FLOATS + IMG;
    constexpr unsigned OOB = 0xFFFFFFFFu, OOB16 = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float pool[];
    auto u_of = [&](int buf) -> float* { return pool + (USINGLE ? 0 : buf * STAGE); };
    auto img_of = [&](int buf) -> float* { return pool + (USINGLE ? A_FLOATS + buf * IMG : buf * STAGE + A_FLOATS); };

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef ZSV_WINOT_TRACE
    unsigned long long trace_t[6], trace_lap[5] = {0, 0, 0, 0, 0}, trace_last = 0;
#endif
    ZSV_TRACE_MARK(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_tile(gridDim.x, blockIdx.x);
    const int tile = (int)mdiv((unsigned)lid, prm.m_ksplit), split = lid - tile * prm.ksplit;
    const int ctile = (int)mdiv((unsigned)tile, prm.m_tiles_m);
    const int m0 = (tile - ctile * prm.tiles_m) * BM;
    const int n0 = ctile * BN;
    if (prm.ksplit > 1) OUT += (size_t)split * prm.slab_elems;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN), 0, prm.in_bytes, 0x00020000);
    const int ch_bytes = 4 * prm.S;
    // VW: `p` is a virtual voxel (row, wv), row over the N*T*H rows; `sp` = element offset of the real voxel inside its clip
    auto locate = [&](int p, int& n, int& sp, int& t, int& h, int& wv) -> bool {
        if constexpr (VW) {
            if (p < 0 || p >= prm.Pv) return false;
            const int row = p >> prm.wv_log2;
            wv = p - (row << prm.wv_log2);
            n = (int)mdiv((unsigned)row, prm.m_TH);
            const int rr = row - n * (prm.T * prm.H);
            t = (int)mdiv((unsigned)rr, prm.m_H);
            h = rr - t * prm.H;
            sp = rr * prm.W + wv;
            return true;
        } else {
            if (p < 0 || p >= prm.P) return false;
            n = (int)mdiv((unsigned)p, prm.m_S);
            sp = p - n * prm.S;
            t = (int)mdiv((unsigned)sp, prm.m_HW);
            const int r = sp - t * prm.HW;
            h = (int)mdiv((unsigned)r, prm.m_W);
            wv = r - h * prm.W;
            return true;
        }
    };
    auto decode = [&](int p, int& base_bytes, unsigned& hmask) {
        base_bytes = 0;
        hmask = 0;
        int n, sp, t, h, wv;
        if (locate(p, n, sp, t, h, wv)) {
            base_bytes = 4 * (n * prm.C * prm.S + sp);
            for (int kt = 0; kt < prm.kT; ++kt)
                for (int kh = 0; kh < 3; ++kh)
                    hmask |= (unsigned)((unsigned)(h + kh - 1) < (unsigned)prm.H && (unsigned)(t + kt - prm.kT / 2) < (unsigned)prm.T) << (kt * 3 + kh);
        }
    };
    int base_bytes;
    unsigned hmask;
    decode(n0 + 4 * lane, base_bytes, hmask);             // this lane's 16-byte piece of every k row (one row, one bit)
    int halo_base = 0;
    unsigned halo_mask = 0;
    if (!VW && wave == 0) decode((lane & 1) ? n0 + BN : n0 - 1, halo_base, halo_mask);

    // U panels: 6*TM pieces of 16 rows x 64 B, piece q = wave + 4*j = (point q / TM, row block q % TM)
    constexpr int NPIECES = NP * TM, APASS = (NPIECES + 3) / 4;
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Up), 0, prm.u_bytes, 0x00020000);
    const int u_lane_bytes = 4 * ((m0 + (lane >> 2)) * 16 + ((lane & 3) ^ ((((lane >> 2) >> 2) & 1) << 1)) * 4);
    const int u_chunk_bytes = 4 * NP * BK * prm.Mp;

    const int nchunks_all = NCHUNKS > 0 ? NCHUNKS : prm.nblk * prm.R;
    const int c_first = NCHUNKS > 0 ? 0 : split * prm.chunks_per_split;
    const int nchunks = NCHUNKS > 0 ? NCHUNKS : min(prm.chunks_per_split, nchunks_all - c_first);
    int ld_cb = c_first / prm.R, ld_kt = (c_first % prm.R) / 3, ld_kh = c_first % 3;
    // the DMAs of chunk `chunk` into stage `buf`: 4 image pieces (k rows 4*wave .. +3), the halo (wave 0), this wave's U pieces --
    // NDMA instructions, handed out one at a time (issue_piece) between the MFMAs of the chunk before: a wave's ten 1-KiB DMAs in
    // a row waited 0.9 us of a 2.6 us chunk for the CU's address path, issuing no MFMA meanwhile (measured on the temporal kernel
    // with tools/winot_trace.py; see conv_winot4_kernel)
    constexpr int NDMA = 5 + APASS;
    int is_toff = 0, is_r = 0, is_ci0 = 0;
    unsigned is_voff = 0;
    auto issue_begin = [&]() {
        is_toff = 4 * ((ld_kh - 1) * prm.W + (ld_kt - prm.kT / 2) * prm.HW);
        is_r = ld_kt * 3 + ld_kh;
        is_ci0 = ld_cb * 16;
        is_voff = ((hmask >> is_r) & 1u) ? (unsigned)(base_bytes + is_toff) : OOB16;
        if (++ld_kh == 3) {
            ld_kh = 0;
            if (++ld_kt == prm.kT) { ld_kt = 0; ++ld_cb; }
        }
    };
    auto issue_piece = [&](int chunk, int buf, int j) {
        float* us = u_of(buf);
        float* bs = img_of(buf);
        if (j < 4) {
            const int k = 4 * wave + j, ci = is_ci0 + k;                   // wave-uniform row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LDB + C0), 16, (int)(ci < prm.C ? is_voff : OOB16),
                                                     ci < prm.C ? ci * ch_bytes : 0, 0, 0);
        } else if (j == 4) {
            if (!VW && wave == 0) {
                const int k = lane >> 1;
                const int ci = is_ci0 + k;
                const unsigned hok = (halo_mask >> is_r) & 1u;
                unsigned hv = (unsigned)(halo_base + is_toff) | (hok - 1u);
                if (lane >= 32 || ci >= prm.C) hv = OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + B_FLOATS), 4, (int)(hv + (hv == OOB ? 0u : (unsigned)(ci * ch_bytes))), 0, 0, 0);
            }
        } else {
            const int piece = (3 - wave) + 4 * (j - 5);        // (wave 0 also carries the halo DMA: it gets the short share when the pieces do not divide)
            if (piece < NPIECES) {
                const int pt = piece / TM, ib = piece % TM;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_u, (lds_ptr_t)(us + 256 * piece), 16, u_lane_bytes,
                                                         chunk * u_chunk_bytes + 4 * 16 * (pt * prm.Mp + 16 * ib), 0, 0);
            }
        }
    };
    auto issue = [&](int chunk, int buf) {
        issue_begin();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_piece(chunk, buf, j);
    };
    auto place_halo = [&](int buf) {
        if (!VW && wave == 0 && lane < 32) {
            float* bs = img_of(buf);
            const float v = bs[B_FLOATS + lane];
            bs[(lane >> 1) * LDB + ((lane & 1) ? C0 + BN : C0 - 1)] = v;
        }
    };

    f32x4 acc[NP][TM];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    const int v_first = n0 + 64 * wave + 4 * r16;                  // first (virtual) voxel of this lane's quad
    int n_clip = 0, sp_first = 0, w_first = 0;
    bool quad_ok;
    {
        int t_, h_;
        quad_ok = locate(v_first, n_clip, sp_first, t_, h_, w_first);
    }
    const bool zero_d0 = w_first == 0, zero_d5 = w_first + 4 >= prm.W;
    const int nvalid = VW ? prm.W - w_first : 4;                    // VW: real voxels of the quad (1..4 at a row's end; W > Wv - 4)
    const int a_frag = r16 * 16 + ((g ^ (((r16 >> 2) & 1) << 1)) << 2);

    ZSV_TRACE_MARK(1);
    issue(c_first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    place_halo(0);
    __syncthreads();
    ZSV_TRACE_MARK(2);
#ifdef ZSV_WINOT_TRACE
    trace_last = trace_t[2];
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        ZSV_TRACE_LAP(3);
        const bool prefetching = ch + 1 < nchunks && !(W4ABL & 1);
        const float* as = u_of(cur);
        const float* bs = img_of(cur);
        f32x4 a4[NP][TM];
        if (!(W4ABL & 2) || ch == 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[p][i] = *reinterpret_cast<const f32x4*>(as + (p * TM + i) * 256 + a_frag);
        }
        // d1..d4: one ds_read_b128; d0 / d5 are the neighbouring lanes' d4 / d1 (DPP row shifts inside the 16-lane row of a
        // k group) except at the two ends of the wave's 64 voxels, which lanes 0 / 15 of a row read from LDS (`de`): one
        // 4-byte read per k step instead of two 4-way bank-conflicted ones
        float de[2];
        f32x4 dm[2];
        auto fetch = [&](int s, int slot) {
            if ((W4ABL & 2) && ch > 0) return;
            const float* src = bs + (4 * g + s) * LDB + C0 + 64 * wave + 4 * r16;        // image column of d1
            dm[slot] = *reinterpret_cast<const f32x4*>(src);
            de[slot] = src[r16 == 0 ? -1 : 4];           // (used by lanes 0 and 15 of a row only)
        };
        fetch(0, 0);
        ZSV_TRACE_LAP(4);
        if (USINGLE) {                          // every wave holds the chunk's U fragments: the panel may be overwritten
            __syncthreads();
        }
        if (prefetching) issue_begin();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = s & 1;
            const float d1 = dm[sl][0];
            float d2 = dm[sl][1], d3 = dm[sl][2], d4 = dm[sl][3];
            if constexpr (VW) {                              // the tail of a row's last piece belongs to the next row
                d2 = nvalid < 2 ? 0.f : d2;
                d3 = nvalid < 3 ? 0.f : d3;
                d4 = nvalid < 4 ? 0.f : d4;
            }
            const float dl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(de[sl]), __float_as_int(d4), 0x111, 0xf, 0xf, false));   // row_shr:1
            const float dr = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(de[sl]), __float_as_int(d1), 0x101, 0xf, 0xf, false));   // row_shl:1
            const float d0 = zero_d0 ? 0.f : dl, d5 = zero_d5 ? 0.f : dr;
            float v[NP];
            const float t12 = d1 + d2, t34 = d3 + d4, u12 = d1 - d2, u43 = d4 - d3, u42 = d4 - d2, u31 = d3 - d1;
            v[0] = __fmaf_rn(4.f, d0, __fmaf_rn(-5.f, d2, d4));
            v[1] = __fmaf_rn(-4.f, t12, t34);
            v[2] = __fmaf_rn(4.f, u12, u43);
            v[3] = __fmaf_rn(2.f, u31, u42);
            v[4] = __fmaf_rn(-2.f, u31, u42);
            v[5] = __fmaf_rn(4.f, d1, __fmaf_rn(-5.f, d3, d5));
            if (s < 3) fetch(s + 1, sl ^ 1);
            if (s == 0) ZSV_TRACE_LAP(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[p][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[p][i][s], v[p], acc[p][i], 0, 0, 0);
                    // one DMA instruction of the next chunk after every ZSV_DMA_EVERY-th MFMA, from the first burst on (at 64 rows
                    // the barrier that frees the single U panel is behind us)
                    constexpr int EV = ZSV_DMA_EVERY, PER_STEP = (NP * TM) / EV;
                    static_assert(4 * PER_STEP >= NDMA, "a slot for every DMA instruction of a chunk");
                    const int idx = p * TM + i;
                    if (idx % EV == EV - 1 && idx / EV < PER_STEP && s * PER_STEP + idx / EV < NDMA && prefetching) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(c_first + ch + 1, cur ^ 1, s * PER_STEP + idx / EV);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ZSV_TRACE_LAP(1);
        if (!(W4ABL & 4)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ch + 1 < nchunks) place_halo(cur ^ 1);
        __syncthreads();
        }
        ZSV_TRACE_LAP(2);
    }
    if ((W4ABL & 8) && prm.P > 0) return;
    ZSV_TRACE_MARK(3);

    // ---- output transform (+ statistics) (+ add, bias, ReLU) + 16-byte stores: lane holds rows 4g..4g+3 of quad column r16
    if constexpr (EPI != 4) {
        constexpr bool STATS = EPI < 4 && (EPI & 1) != 0, ADD = EPI < 4 && (EPI & 2) != 0, BIASRELU = EPI == 8;
        constexpr unsigned OOBS = 0xFFFFFFF0u;
        float* red = pool;
        if (STATS) __syncthreads();
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(OUT, 0, prm.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ADD ? prm.add : IN), 0, ADD ? prm.out_bytes : 0u, 0x00020000);
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t brsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BIASRELU ? prm.bias : IN), 0, BIASRELU ? 4u * (unsigned)prm.M : 0u, 0x00020000);
        // byte offset of (row m0 + 4g, this quad); rows 16 i + r go into the scalar offset
        const unsigned lane_off = quad_ok ? 4u * (unsigned)(n_clip * prm.M * prm.S + sp_first + (m0 + 4 * g) * prm.S) : OOBS;
        const bool ragged = m0 + BM > prm.M;                  // (wave-uniform)
        const int row_bytes = 4 * prm.S;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 addv[4];
            if constexpr (ADD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned voff = (ragged && m0 + 16 * i + 4 * g + r >= prm.M) ? OOBS : lane_off;
                    if (!VW || nvalid >= 4) {
                        addv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)voff, (16 * i + r) * row_bytes, 0));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            addv[r][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(arsrc, (int)(e < nvalid ? voff + 4u * e : OOBS), (16 * i + r) * row_bytes, 0));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float M0 = acc[0][i][r], M1 = acc[1][i][r], M2 = acc[2][i][r], M3 = acc[3][i][r], M4 = acc[4][i][r], M5 = acc[5][i][r];
                const float s12 = M1 + M2, d12 = M1 - M2, s34 = M3 + M4, d34 = M3 - M4;
                f32x4 y = {(M0 + s12) + s34, __fmaf_rn(2.f, d34, d12), __fmaf_rn(4.f, s34, s12), __fmaf_rn(8.f, d34, d12) + M5};
                bool live = quad_ok;
                unsigned voff = lane_off;
                if (ragged) {
                    live = quad_ok && m0 + 16 * i + 4 * g + r < prm.M;
                    voff = live ? lane_off : OOBS;
                }
                if constexpr (STATS) {
                    if constexpr (VW) {                        // only the real voxels of a row's last quad count
                        y[1] = nvalid < 2 ? 0.f : y[1];
                        y[2] = nvalid < 3 ? 0.f : y[2];
                        y[3] = nvalid < 4 ? 0.f : y[3];
                    }
                    float s1 = live ? (y[0] + y[1]) + (y[2] + y[3]) : 0.f;
                    float s2 = live ? (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]) : 0.f;
#define ZSV_ROW16_SUM(v)                                                                                                         \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));                               \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));                               \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
                    ZSV_ROW16_SUM(s1)
                    ZSV_ROW16_SUM(s2)
#undef ZSV_ROW16_SUM
                    if (r16 == 0) *reinterpret_cast<f32x2*>(&red[(wave * BM + 16 * i + 4 * g + r) * 2]) = f32x2{s1, s2};
                }
                if constexpr (ADD) y += addv[r];
                if constexpr (BIASRELU) {                   // relu(conv + bias[m]) (C3D's forward, network.py:147-166; folded-BatchNorm inference)
                    const float bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brsrc, 4 * (m0 + 16 * i + 4 * g + r), 0, 0));      // (rows beyond M read 0)
                    y[0] = fmaxf(y[0] + bv, 0.f); y[1] = fmaxf(y[1] + bv, 0.f); y[2] = fmaxf(y[2] + bv, 0.f); y[3] = fmaxf(y[3] + bv, 0.f);
                }
                const int soff = (16 * i + r) * row_bytes;
                if (!VW || nvalid >= 4) {
                    // The row offset goes into the VECTOR offset here (one add), not into the scalar offset as for the 4-byte
                    // stores: a 16-byte buffer store with an SGPR soffset followed at once by a VALU write of its data registers
                    // stored the NEW value of the last dword now and then on gfx950 (elements w % 4 == 3 of single rows, a few
                    // hundred per 10^7); the compiler's hazard table (1 wait state for > 64-bit store data) only covers the
                    // immediate-soffset form, which this is.
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), orsrc, (int)(live ? voff + (unsigned)soff : OOBS), 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[e]), orsrc, (int)(e < nvalid ? voff + 4u * e : OOBS), soff, 0);
                }
            }
        }
        if constexpr (STATS) {
            __syncthreads();
            if (tid < BM && m0 + tid < prm.M) {               // the 4 waves' partials, in wave order
                const float t1 = (red[tid * 2] + red[(BM + tid) * 2]) + (red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2]);
                const float t2 = (red[tid * 2 + 1] + red[(BM + tid) * 2 + 1]) + (red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1]);
                const int tn = n0 / BN;
                prm.stat_sum[(size_t)(m0 + tid) * prm.tiles_n + tn] = t1;
                prm.stat_sq[(size_t)(m0 + tid) * prm.tiles_n + tn] = t2;
            }
        }
    } else {
    const bool stats = prm.stat_sum != nullptr;
    float* red = pool;                                   // [4 waves][BM][2] partial sums (the staging LDS is free now)
    if (stats) __syncthreads();
    const int quad_off = n_clip * prm.M * prm.S + sp_first;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * g + r;
            const float M0 = acc[0][i][r], M1 = acc[1][i][r], M2 = acc[2][i][r], M3 = acc[3][i][r], M4 = acc[4][i][r], M5 = acc[5][i][r];
            const float s12 = M1 + M2, d12 = M1 - M2, s34 = M3 + M4, d34 = M3 - M4;
            f32x4 y = {(M0 + s12) + s34, __fmaf_rn(2.f, d34, d12), __fmaf_rn(4.f, s34, s12), __fmaf_rn(8.f, d34, d12) + M5};
            float s1 = 0.f, s2 = 0.f;
            if (quad_ok && m < prm.M) {
                const int off = quad_off + m * prm.S;
                if (!VW || nvalid >= 4) {
                    s1 = (y[0] + y[1]) + (y[2] + y[3]);
                    s2 = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
                    if (prm.add != nullptr) y += *reinterpret_cast<const f32x4*>(prm.add + off);      // (VW: 4-byte aligned 16-byte accesses)
                    if (prm.bias != nullptr) y += prm.bias[m];
                    if (prm.relu) { y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f); }
                    *reinterpret_cast<f32x4*>(OUT + off) = y;
                } else {                                  // a row's last quad: its first `nvalid` values are real voxels
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        if (e < nvalid) {
                            float v = y[e];
                            s1 += v;
                            s2 += v * v;
                            if (prm.add != nullptr) v += prm.add[off + e];
                            if (prm.bias != nullptr) v += prm.bias[m];
                            if (prm.relu) v = fmaxf(v, 0.f);
                            OUT[off + e] = v;
                        }
                    }
                }
            }
            if (stats) {                                  // 16 lanes (r16) share row m
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (r16 == 0) {
                    red[(wave * BM + 16 * i + 4 * g + r) * 2] = s1;
                    red[(wave * BM + 16 * i + 4 * g + r) * 2 + 1] = s2;
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (tid < BM && m0 + tid < prm.M) {               // the 4 waves' partials, in wave order
            const float t1 = (red[tid * 2] + red[(BM + tid) * 2]) + (red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2]);
            const float t2 = (red[tid * 2 + 1] + red[(BM + tid) * 2 + 1]) + (red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1]);
            const int tn = n0 / BN;
            prm.stat_sum[(size_t)(m0 + tid) * prm.tiles_n + tn] = t1;
            prm.stat_sq[(size_t)(m0 + tid) * prm.tiles_n + tn] = t2;
        }
    }
    }
#ifdef ZSV_WINOT_TRACE
    ZSV_TRACE_MARK(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZSV_TRACE_MARK(5);
    if (tid == 0) {
        unsigned long long* rec = reinterpret_cast<unsigned long long*>(OUT + (prm.out_bytes >> 2)) + (size_t)blockIdx.x * 12;
        for (int i = 0; i < 6; ++i) rec[i] = trace_t[i];
        rec[6] = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_ID
        rec[7] = (unsigned long long)tile;
        for (int i = 0; i < 3; ++i) rec[8 + i] = trace_lap[i];
        rec[11] = (trace_lap[3] << 32) | (trace_lap[4] & 0xFFFFFFFFull);
    }
#endif
#endif
}

// ================================================================================================
// Temporal 3x1x1 stride-1 "same" convolution (the second half of Conv2Plus1D, resnet.py:46-52) with the kt taps in Winograd
// F(2,3) form ALONG T: two frames (t, t+1) of one (h, w) position share the four input frames d0..d3 = in[t-1..t+2],
//     V = (d0-d2, d1+d2, d2-d1, d1-d3),  U as above from (g0, g1, g2) = the kt weights,  y(t) = M0+M1+M2,  y(t+1) = M1-M2-M3
// -- 4 multiplies per output pair instead of 6, and ONE image per 16-channel block instead of one per tap: the direct kernel
// (conv_tap.hip) gathers three frame-shifted copies of the input, which is what bounds the 64-row forward (its image DMAs cost
// 20 % of its time, profiles/r02_tap_kernel_ablation.txt).
// A workgroup tile is ALL T frames (T = 4, 8 or 16) of PW = 256 / T consecutive (h, w) positions of one clip: 128 frame PAIRS,
// one 16-pair block = 16 positions of one frame pair, two blocks per wave; the image of a chunk is [16 k][T][PW] (1 KiB per k
// row = one 16-byte DMA instruction), there is no halo: frames -1 and T are the zero padding, a wave-uniform zeroing of d0 / d3.
// K = 16-channel blocks (9 chunks for the 144 -> 64 forward).  Epilogue as above (statistics / + add), two 4-byte stores per
// value pair (64 contiguous bytes per row, frame and 16 lanes).  PRE as in conv_tap.hip: the BatchNorm + ReLU in front of the
// convolution is applied to d0..d3 in registers (no per-voxel masks needed: only whole frames are padding).
template <int TM, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_winot_kernel(WinoParams prm, const float* __restrict__ Up,
                                                            const float* __restrict__ IN, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM, BK = 16;
    constexpr int LDB = 260;                // 256 used; 4 * LDB = 16 mod 32: the k rows 4g+s of the two lane halves of a ds_read_b32 split the banks
    constexpr int A_FLOATS = 4 * BM * BK, B_FLOATS = BK * LDB;
    constexpr int COEF_AT = A_FLOATS + B_FLOATS;        // PRE: 1 KiB DMA target, 32 floats used (16 scales, 16 shifts)
    constexpr int STAGE = COEF_AT + (PRE ? 256 : 0);
    constexpr unsigned OOB16 = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float pool[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(gridDim.x, blockIdx.x);          // the row tiles of one column tile are neighbours (one L2)
    const int m0 = (tile % prm.tiles_m) * BM;
    const int ct = tile / prm.tiles_m;                          // column tile = (clip, position segment)
    const int n_img = ct / prm.segs, pos0 = (ct - n_img * prm.segs) * prm.PW;
    const int T = prm.T, PW = prm.PW, HW = prm.HW;
    const int pq = PW >> 2;                                     // 16-byte pieces per frame row

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN), 0, prm.in_bytes, 0x00020000);
    const int ch_bytes = 4 * prm.S;
    // this lane's 16-byte piece of every k row: frame lane / pq, positions pos0 + 4 * (lane % pq) ..
    const int pf = lane / pq, pp = pos0 + 4 * (lane - pf * pq);
    const unsigned piece_off = pp < HW ? (unsigned)(4 * (n_img * prm.C * prm.S + pf * HW + pp)) : OOB16;

    constexpr int APASS = TM;
    const float* a_src[APASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
        const int q = wave + 4 * j, pt = q / TM, ib = q % TM, row = lane >> 2;
        const int sw = ((lane & 3) ^ (((row >> 2) & 1) << 1)) * 4;
        a_src[j] = Up + ((size_t)pt * prm.Mp + m0 + 16 * ib + row) * 16 + sw;
    }
    const size_t a_chunk_stride = (size_t)4 * BK * prm.Mp;
    const __amdgpu_buffer_rsrc_t pre_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? prm.pre_coef : IN), 0, PRE ? 8u * (unsigned)prm.pre_pitch : 0u, 0x00020000);

    const int nchunks = prm.nblk;
    auto issue = [&](int chunk, int buf) {
        float* as = pool + buf * STAGE;
        float* bs = as + A_FLOATS;
        const int ci0 = chunk * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * wave + j, ci = ci0 + k;                   // wave-uniform row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LDB), 16, (int)(ci < prm.C ? piece_off : OOB16),
                                                     ci < prm.C ? ci * ch_bytes : 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < APASS; ++j)
            __builtin_amdgcn_global_load_lds(a_src[j] + (size_t)chunk * a_chunk_stride, (lds_ptr_t)(as + 256 * (wave + 4 * j)), 16, 0, 0);
        if constexpr (PRE) {
            if (wave == 0) {
                const unsigned off = lane < 8 ? 4u * (unsigned)((lane >> 2) * prm.pre_pitch + ci0 + 4 * (lane & 3)) : OOB16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pre_rsrc, (lds_ptr_t)(as + COEF_AT), 16, (int)off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[4][TM][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    // this wave's two 16-pair blocks: block b = 2 * wave + j = (frame pair b / (PW/16), 16-position group b % (PW/16))
    int col_of[2], tp_of[2];
    bool zero_d0[2], zero_d3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = 2 * wave + j, groups = PW >> 4;
        tp_of[j] = b / groups;
        col_of[j] = 16 * (b - tp_of[j] * groups) + r16;            // position inside the tile
        zero_d0[j] = tp_of[j] == 0;                                  // frame -1
        zero_d3[j] = 2 * tp_of[j] + 2 >= T;                          // frame T
    }

    issue(0, 0);
    __syncthreads();                                   // (vmcnt(0) before the barrier)
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) issue(ch + 1, cur ^ 1);
        const float* as = pool + cur * STAGE;
        const float* bs = as + A_FLOATS;
        f32x4 a4[4][TM];
        const int a_frag = r16 * 16 + ((g ^ (((r16 >> 2) & 1) << 1)) << 2);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[p][i] = *reinterpret_cast<const f32x4*>(as + (p * TM + i) * 256 + a_frag);
        float psc[PRE ? 4 : 1], psh[PRE ? 4 : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                psc[s] = as[COEF_AT + 4 * g + s];
                psh[s] = as[COEF_AT + 16 + 4 * g + s];
            }
        }
        float raw[2][2][4];
        auto fetch = [&](int s, int slot) {
            const float* row = bs + (4 * g + s) * LDB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int f0 = 2 * tp_of[j] - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = min(max(f0 + i, 0), T - 1);       // (clamped: the out-of-clip frames are zeroed below)
                    raw[slot][j][i] = row[f * PW + col_of[j]];
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = s & 1;
            float v[4][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i] = raw[sl][j][i];
                    if constexpr (PRE) d[i] = fmaxf(__fmaf_rn(d[i], psc[s], psh[s]), 0.f);
                }
                const float d0 = zero_d0[j] ? 0.f : d[0], d3 = zero_d3[j] ? 0.f : d[3];
                v[0][j] = d0 - d[2]; v[1][j] = d[1] + d[2]; v[2][j] = d[2] - d[1]; v[3][j] = d[1] - d3;
            }
            if (s < 3) fetch(s + 1, sl ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[p][i][s], v[p][j], acc[p][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                               // (vmcnt(0) lgkmcnt(0) + barrier)
    }

    // ---- output transform (+ statistics) (+ add) + store: lane holds rows 4g..4g+3 of pair column r16 of its two blocks
    const bool stats = prm.stat_sum != nullptr;
    float* red = pool;
    if (stats) __syncthreads();
    int off0[2];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pos = pos0 + col_of[j];
        ok[j] = pos < HW;
        off0[j] = n_img * prm.M * prm.S + 2 * tp_of[j] * HW + pos;     // frame 2*tp; frame 2*tp+1 is HW further
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * g + r;
            const int row_off = m * prm.S;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float M0 = acc[0][i][j][r], M1 = acc[1][i][j][r], M2 = acc[2][i][j][r], M3 = acc[3][i][j][r];
                float y0 = (M0 + M1) + M2, y1 = (M1 - M2) - M3;
                if (ok[j] && m < prm.M) {
                    s1 += y0 + y1;
                    s2 += y0 * y0 + y1 * y1;
                    const int off = off0[j] + row_off;
                    if (prm.add != nullptr) { y0 += prm.add[off]; y1 += prm.add[off + HW]; }
                    if (prm.bias != nullptr) { const float bv = prm.bias[m]; y0 += bv; y1 += bv; }
                    if (prm.relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
                    OUT[off] = y0;
                    OUT[off + HW] = y1;
                }
            }
            if (stats) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (r16 == 0) {
                    red[(wave * BM + 16 * i + 4 * g + r) * 2] = s1;
                    red[(wave * BM + 16 * i + 4 * g + r) * 2 + 1] = s2;
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (tid < BM && m0 + tid < prm.M) {
            const float t1 = (red[tid * 2] + red[(BM + tid) * 2]) + (red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2]);
            const float t2 = (red[tid * 2 + 1] + red[(BM + tid) * 2 + 1]) + (red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1]);
            prm.stat_sum[(size_t)(m0 + tid) * prm.tiles_n + ct] = t1;
            prm.stat_sq[(size_t)(m0 + tid) * prm.tiles_n + ct] = t2;
        }
    }
#endif
}

// ================================================================================================
// The same temporal convolution in F(4,3) form ALONG T (T % 4 == 0: every T the F(2,3) kernel takes): four frames t..t+3 of one
// (h, w) position share the six input frames d0..d5 = in[t-1..t+4]; V, U and the output transform are conv_wino4_kernel's with T in
// the place of W -- 6 multiplies per 4 outputs instead of 12: 25 % fewer MFMAs than conv_winot_kernel.  Tile and image as there (all
// T frames of PW = 256 / T positions, [16 k][T][PW], no halo: frames -1 and T..T+3 are a wave-uniform zeroing of d0 / d5); 64 frame
// QUADS per workgroup, one 16-quad block (16 positions of one quad) per wave, six accumulator sets.  At 64 rows the six U panels are
// single-buffered (a second barrier per chunk, as conv_wino4_kernel) so that two workgroups fit a CU.
// EPI selects the epilogue at compile time: 0 = plain stores (the input gradients), 1 = + BatchNorm partial statistics (the
// training forward), 2 = run-time add / bias / ReLU (+ statistics) (inference engine, shortcut gradients).  With K = 64 .. 144
// channels these kernels run 4 - 9 chunks per tile, so what surrounds the chunk loop counts: the run-time-flag epilogue compiled to
// ~130 instructions and five branches per output row (1 480 vector instructions per wave against 288 MFMAs on the layer1 dgrad,
// profiles/r03_t1_pmc.json); EPI 0 / 1 are branch-free: buffer stores whose row / frame offsets are scalar, out-of-range lanes
// dropped by the descriptor's range check, 16-lane DPP sums for the statistics.
template <int TM, bool PRE, int EPI>
__global__ __launch_bounds__(256, 2) void conv_winot4_kernel(WinoParams prm, const float* __restrict__ Up,
                                                             const float* __restrict__ IN, float* __restrict__ OUT) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 16 * TM, BK = 16, NP = 6;
    constexpr bool USINGLE = TM >= 4;
    constexpr int LDB = 260;                // 256 used; 4 * LDB = 16 mod 32: the k rows 4g+s of the two lane halves of a ds_read_b32 split the banks
    constexpr int A_FLOATS = NP * BM * BK, B_FLOATS = BK * LDB;
    constexpr int IMG = B_FLOATS + (PRE ? 256 : 0);     // image (+ PRE: 1 KiB DMA target, 32 floats used: 16 scales, 16 shifts)
    constexpr int STAGE = A_FLOATS + IMG;
    constexpr unsigned OOB16 = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) float pool[];
    auto u_of = [&](int buf) -> float* { return pool + (USINGLE ? 0 : buf * STAGE); };
    auto img_of = [&](int buf) -> float* { return pool + (USINGLE ? A_FLOATS + buf * IMG : buf * STAGE + A_FLOATS); };

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef ZSV_WINOT_TRACE
    unsigned long long trace_t[6], trace_lap[5] = {0, 0, 0, 0, 0}, trace_last = 0;
#endif
    ZSV_TRACE_MARK(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(gridDim.x, blockIdx.x);          // the row tiles of one column tile are neighbours (one L2)
    const int m0 = (tile % prm.tiles_m) * BM;
    const int ct = tile / prm.tiles_m;                          // column tile = (clip, position segment)
    const int n_img = ct / prm.segs, pos0 = (ct - n_img * prm.segs) * prm.PW;
    const int T = prm.T, PW = prm.PW, HW = prm.HW;
    const int pq_log2 = prm.pw_log2 - 2;                        // 16-byte pieces per frame row: PW / 4 (a power of two)

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN), 0, prm.in_bytes, 0x00020000);
    const int ch_bytes = 4 * prm.S;
    // this lane's 16-byte piece of every k row: frame lane / pq, positions pos0 + 4 * (lane % pq) ..
    const int pf = lane >> pq_log2, pp = pos0 + 4 * (lane - (pf << pq_log2));
    const unsigned piece_off = pp < HW ? (unsigned)(4 * (n_img * prm.C * prm.S + pf * HW + pp)) : OOB16;

    // U panels through a buffer descriptor: one per-lane byte offset, pieces and chunks in the scalar offset
    constexpr int NPIECES = NP * TM, APASS = (NPIECES + 3) / 4;
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Up), 0, prm.u_bytes, 0x00020000);
    const int u_lane_bytes = 4 * ((m0 + (lane >> 2)) * 16 + ((lane & 3) ^ ((((lane >> 2) >> 2) & 1) << 1)) * 4);
    const int u_chunk_bytes = 4 * NP * BK * prm.Mp;
    const __amdgpu_buffer_rsrc_t pre_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? prm.pre_coef : IN), 0, PRE ? 8u * (unsigned)prm.pre_pitch : 0u, 0x00020000);

    const int nchunks = prm.nblk;
    // one DMA instruction of chunk `chunk` into stage `buf`: pieces 0..3 = this wave's four image rows, 4..4+APASS-1 = its U pieces,
    // then (PRE, wave 0) the 16 scales / shifts.  Issued ONE AT A TIME between the MFMAs of the chunk before: the nine to eleven
    // 1-KiB DMAs of a wave in a row took 0.86 us of a 2.6 us chunk to get accepted by the CU's address path (64 B / clock shared
    // by eight waves that all ask at the chunk boundary; tools/winot_trace.py) while the wave issued no MFMA.
    constexpr int NDMA = 4 + APASS + (PRE ? 1 : 0);
    auto issue_piece = [&](int chunk, int buf, int j) {
        float* as = u_of(buf);
        float* bs = img_of(buf);
        const int ci0 = chunk * 16;
        if (j < 4) {
            if (ZSV_WINOT_ABLATE == 2) return;
            const int k = 4 * wave + j, ci = ci0 + k;                   // wave-uniform row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bs + k * LDB), 16, (int)(ci < prm.C ? piece_off : OOB16),
                                                     ci < prm.C ? ci * ch_bytes : 0, 0, 0);
        } else if (j < 4 + APASS) {
            const int q = (3 - wave) + 4 * (j - 4);
            if (q < NPIECES && ZSV_WINOT_ABLATE != 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_u, (lds_ptr_t)(as + 256 * q), 16, u_lane_bytes,
                                                         chunk * u_chunk_bytes + 4 * 16 * ((q / TM) * prm.Mp + 16 * (q % TM)), 0, 0);
            }
        } else if constexpr (PRE) {
            if (wave == 0) {
                const unsigned off = lane < 8 ? 4u * (unsigned)((lane >> 2) * prm.pre_pitch + ci0 + 4 * (lane & 3)) : OOB16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pre_rsrc, (lds_ptr_t)(bs + B_FLOATS), 16, (int)off, 0, 0, 0);
            }
        }
    };
    auto issue = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_piece(chunk, buf, j);
    };

    f32x4 acc[NP][TM];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    // this wave's 16-quad block: block b = wave = (frame quad b / (PW/16), 16-position group b % (PW/16))
    // quad q = 16 * wave + r16 of the tile's 64 = (frame quad q / PW, position q % PW).  With PW >= 16 (T <= 16) a wave's 16
    // quads share their frame quad (wave-uniform: scalar registers, as before); with T = 32 (PW = 8: the 32-frame clips of
    // BASELINE config 5 in fp32) a wave holds two frame quads of 8 positions and the frame quad is a per-lane value
    int tq, col;
    if (prm.pw_log2 >= 4) {
        const int groups_log2 = prm.pw_log2 - 4;                   // 16-position groups per frame: PW / 16
        const int tqw = wave >> groups_log2;
        tq = tqw;
        col = 16 * (wave - (tqw << groups_log2)) + r16;            // position inside the tile
    } else {
        const int q = 16 * wave + r16;
        tq = q >> prm.pw_log2;
        col = q & (PW - 1);
    }
    const bool zero_d0 = tq == 0;                                   // frame -1
    const bool zero_d5 = 4 * tq + 4 >= T;                           // frame T
    const int a_frag = r16 * 16 + ((g ^ (((r16 >> 2) & 1) << 1)) << 2);

    ZSV_TRACE_MARK(1);
    issue(0, 0);
    __syncthreads();                                   // (vmcnt(0) before the barrier)
    ZSV_TRACE_MARK(2);
#ifdef ZSV_WINOT_TRACE
    trace_last = trace_t[2];
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool prefetching = ch + 1 < nchunks;
        ZSV_TRACE_LAP(3);
        const float* as = u_of(cur);
        const float* bs = img_of(cur);
        f32x4 a4[NP][TM];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[p][i] = *reinterpret_cast<const f32x4*>(as + (p * TM + i) * 256 + a_frag);
        float psc[PRE ? 4 : 1], psh[PRE ? 4 : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                psc[s] = bs[B_FLOATS + 4 * g + s];
                psh[s] = bs[B_FLOATS + 16 + 4 * g + s];
            }
        }
        float raw[2][NP];
        auto fetch = [&](int s, int slot) {
            const float* row = bs + (4 * g + s) * LDB;
            const int f0 = 4 * tq - 1;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int f = min(max(f0 + i, 0), T - 1);           // (clamped: the out-of-clip frames are zeroed below)
                raw[slot][i] = row[f * PW + col];
            }
        };
        fetch(0, 0);
        ZSV_TRACE_LAP(4);                              // U fragments + first image values (the mark waits for them)
        if (USINGLE) __syncthreads();           // every wave holds the chunk's U fragments: the panel may be overwritten
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = s & 1;
            float d[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                d[i] = raw[sl][i];
                if constexpr (PRE) d[i] = fmaxf(__fmaf_rn(d[i], psc[s], psh[s]), 0.f);
            }
            const float d0 = zero_d0 ? 0.f : d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = zero_d5 ? 0.f : d[5];
            float v[NP];
            const float t12 = d1 + d2, t34 = d3 + d4, u12 = d1 - d2, u43 = d4 - d3, u42 = d4 - d2, u31 = d3 - d1;
            v[0] = __fmaf_rn(4.f, d0, __fmaf_rn(-5.f, d2, d4));
            v[1] = __fmaf_rn(-4.f, t12, t34);
            v[2] = __fmaf_rn(4.f, u12, u43);
            v[3] = __fmaf_rn(2.f, u31, u42);
            v[4] = __fmaf_rn(-2.f, u31, u42);
            v[5] = __fmaf_rn(4.f, d1, __fmaf_rn(-5.f, d3, d5));
            if (s < 3) fetch(s + 1, sl ^ 1);
            if (s == 0) ZSV_TRACE_LAP(0);           // DMA issue, U fragments, first V: everything before the first MFMA of a chunk
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (ZSV_WINOT_ABLATE == 3) acc[p][i][0] += a4[p][i][s] * v[p];
                    else acc[p][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[p][i][s], v[p], acc[p][i], 0, 0, 0);
                    // one DMA instruction of the next chunk after every ZSV_DMA_EVERY-th MFMA
                    constexpr int EV = ZSV_DMA_EVERY, PER_STEP = (NP * TM) / EV;
                    static_assert(4 * PER_STEP >= NDMA, "a slot for every DMA instruction of a chunk");
                    const int idx = p * TM + i;
                    if (idx % EV == EV - 1 && idx / EV < PER_STEP && s * PER_STEP + idx / EV < NDMA && prefetching) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(ch + 1, cur ^ 1, s * PER_STEP + idx / EV);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ZSV_TRACE_LAP(1);                              // the four k steps (transforms + 72 / 96 MFMAs)
        __syncthreads();                               // (vmcnt(0) lgkmcnt(0) + barrier)
        ZSV_TRACE_LAP(2);                              // waiting for the next chunk's DMAs and the other waves
    }

    ZSV_TRACE_MARK(3);
    // ---- output transform (+ statistics) (+ add, bias, ReLU) + store: lane holds rows 4g..4g+3 of quad column r16 of its block
    const int pos = pos0 + col;
    const bool ok = pos < HW;
    float* red = pool;
    if constexpr (EPI != 2) {
        // rows 16 i + r and frames f are scalar offsets of a buffer store; a lane outside the map (or, on the last row tile, a row
        // beyond M) gets an offset past the descriptor's range: the store is dropped, no branch
        constexpr unsigned OOB = 0xFFFFFFFFu;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(OUT, 0, prm.out_bytes, 0x00020000);
        const unsigned lane_off = ok ? 4u * (unsigned)(n_img * prm.M * prm.S + 4 * tq * HW + pos + (m0 + 4 * g) * prm.S) : OOB;
        const bool ragged = m0 + BM > prm.M;                  // (wave-uniform)
        const int row_bytes = 4 * prm.S, frame_bytes = 4 * HW;
        if (EPI == 1) __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float M0 = acc[0][i][r], M1 = acc[1][i][r], M2 = acc[2][i][r], M3 = acc[3][i][r], M4 = acc[4][i][r], M5 = acc[5][i][r];
                const float s12 = M1 + M2, d12 = M1 - M2, s34 = M3 + M4, d34 = M3 - M4;
                const float y[4] = {(M0 + s12) + s34, __fmaf_rn(2.f, d34, d12), __fmaf_rn(4.f, s34, s12), __fmaf_rn(8.f, d34, d12) + M5};
                unsigned voff = lane_off;
                bool live = ok;
                if (ragged) {
                    live = ok && m0 + 16 * i + 4 * g + r < prm.M;
                    voff = live ? lane_off : OOB;
                }
                const int soff = (16 * i + r) * row_bytes;
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[f]), orsrc, (int)(ZSV_WINOT_ABLATE == 1 && y[f] != 123.456f ? OOB : voff),
                                                          soff + f * frame_bytes, 0);
                if constexpr (EPI == 1) {
                    float s1 = live ? (y[0] + y[1]) + (y[2] + y[3]) : 0.f;
                    float s2 = live ? (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]) : 0.f;
                    // the 16 lanes of a DPP row share output row m: quad swaps, then the two mirrors
#define ZSV_ROW16_SUM(v)                                                                                                         \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));                               \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));                               \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
                    ZSV_ROW16_SUM(s1)
                    ZSV_ROW16_SUM(s2)
#undef ZSV_ROW16_SUM
                    if (r16 == 0) *reinterpret_cast<f32x2*>(&red[(wave * BM + 16 * i + 4 * g + r) * 2]) = f32x2{s1, s2};
                }
            }
        }
    } else {
    const bool stats = prm.stat_sum != nullptr;
    if (stats) __syncthreads();
    const int off0 = n_img * prm.M * prm.S + 4 * tq * HW + pos;        // frame 4*tq; the next three are HW further each
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * g + r;
            const float M0 = acc[0][i][r], M1 = acc[1][i][r], M2 = acc[2][i][r], M3 = acc[3][i][r], M4 = acc[4][i][r], M5 = acc[5][i][r];
            const float s12 = M1 + M2, d12 = M1 - M2, s34 = M3 + M4, d34 = M3 - M4;
            float y[4] = {(M0 + s12) + s34, __fmaf_rn(2.f, d34, d12), __fmaf_rn(4.f, s34, s12), __fmaf_rn(8.f, d34, d12) + M5};
            float s1 = 0.f, s2 = 0.f;
            if (ok && m < prm.M) {
                s1 = (y[0] + y[1]) + (y[2] + y[3]);
                s2 = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
                const int off = off0 + m * prm.S;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float yv = y[f];
                    if (prm.add != nullptr) yv += prm.add[off + f * HW];
                    if (prm.bias != nullptr) yv += prm.bias[m];
                    if (prm.relu) yv = fmaxf(yv, 0.f);
                    OUT[off + f * HW] = yv;
                }
            }
            if (stats) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (r16 == 0) {
                    red[(wave * BM + 16 * i + 4 * g + r) * 2] = s1;
                    red[(wave * BM + 16 * i + 4 * g + r) * 2 + 1] = s2;
                }
            }
        }
    }
    }
    if (EPI == 1 || (EPI == 2 && prm.stat_sum != nullptr)) {
        __syncthreads();
        if (tid < BM && m0 + tid < prm.M) {
            const float t1 = (red[tid * 2] + red[(BM + tid) * 2]) + (red[(2 * BM + tid) * 2] + red[(3 * BM + tid) * 2]);
            const float t2 = (red[tid * 2 + 1] + red[(BM + tid) * 2 + 1]) + (red[(2 * BM + tid) * 2 + 1] + red[(3 * BM + tid) * 2 + 1]);
            prm.stat_sum[(size_t)(m0 + tid) * prm.tiles_n + ct] = t1;
            prm.stat_sq[(size_t)(m0 + tid) * prm.tiles_n + ct] = t2;
        }
    }
#ifdef ZSV_WINOT_TRACE
    ZSV_TRACE_MARK(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZSV_TRACE_MARK(5);
    if (tid == 0) {
        unsigned long long* rec = reinterpret_cast<unsigned long long*>(OUT + (prm.out_bytes >> 2)) + (size_t)blockIdx.x * 12;
        for (int i = 0; i < 6; ++i) rec[i] = trace_t[i];
        rec[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                 (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // XCC_ID : HW_ID
        rec[7] = (unsigned long long)tile;
        for (int i = 0; i < 3; ++i) rec[8 + i] = trace_lap[i];
        rec[11] = (trace_lap[3] << 32) | (trace_lap[4] & 0xFFFFFFFFull);
    }
#endif
#endif
}

// ---- host side -----------------------------------------------------------------------------------
static size_t wino_align(size_t b) { return (b + 255) & ~(size_t)255; }

// rows per workgroup for M produced channels: 64, or 48 when that pads less (144 = 3 x 48, 288 = 6 x 48)
static int wino_tm(int M) {
    const int p64 = (M + 63) / 64 * 64, p48 = (M + 47) / 48 * 48;
    return p48 < p64 ? 3 : 4;
}

// K parts: 1 when the tiles alone fill the two-workgroups-per-CU round, else 2..4 parts of >= 12 chunks each that do
// (their raw partial results go to slabs, summed by splitk_reduce); 0 = too few tiles either way: the direct kernel
static bool winot_geometry(const zsv_conv_desc* d, int M);

// Virtual-width form of the F(4,3) kernel (conv_wino4_kernel VW): row lengths that are not a multiple of 4 but fill a power-of-two
// row of 8 / 16 / 32 / 64 voxels to more than Wv - 4 (7, 14, 15, 29-31, 61-63: at most 12.5 % empty columns)
static int wino_vw_width(const zsv_conv_desc* d) {
    if (d->Wi % 4 == 0 || ZSV_KNOB(WINO_NO_VW) || ZSV_KNOB(WINO_NO_F43)) return 0;
    for (int wv = 8; wv <= 64; wv *= 2)
        if (d->Wi < wv) return d->Wi > wv - 4 && 8 * d->Wi >= 7 * wv ? wv : 0;
    return 0;
}
// rows per workgroup and K parts of a virtual-width launch: the pair with the best (round fill x row padding) over one or two
// rounds of the 512 resident workgroups, fewer K parts on ties (each part writes a slab).  Returns the K parts (0: none fits).
static int wino_vw_plan(const zsv_conv_desc* d, int M, int& tm) {
    const int wv = wino_vw_width(d);
    const long tiles_n = ((long)d->N * d->Ti * d->Hi * wv + 255) / 256;
    const int C = M == d->Cout ? d->Cin : d->Cout;
    const long nchunks = (long)((C + 15) / 16) * 3 * d->kT;
    const char* e = ZSV_KNOB(WINO_VW_MIN_WGS);
    const long min_wgs = e ? atol(e) : 256;               // fewer workgroups than that: the direct / F(2,3) kernels (tests lower it)
    double best = 0.0;
    int best_ks = 0;
    tm = 4;
    for (int t = 4; t >= 3; --t) {
        const int bm = 16 * t;
        const long tiles = ((M + bm - 1) / bm) * tiles_n;
        const double rows = (double)M / (double)(((M + bm - 1) / bm) * bm);
        for (int ks = 1; ks <= 8; ++ks) {
            if (ks > 1 && (nchunks / ks < 12 || tiles >= 512)) break;      // K parts only below one round of workgroups
            const long wgs = tiles * ks;
            if (wgs < min_wgs || (ks > 1 && wgs > 1024)) continue;
            // (beyond one round the workgroups backfill: the partial last round costs about a third of what whole rounds would)
            const double r = (double)wgs / 512.0, rc = (double)((wgs + 511) / 512);
            const double fill = wgs <= 512 ? r : r / (r + 0.33 * (rc - r));
            const double eff = rows * fill * (1.0 - 0.03 * (ks - 1));
            if (eff > best + 1e-9) { best = eff; best_ks = ks; tm = t; }
        }
    }
    return best_ks;
}

static int wino_ksplit(const zsv_conv_desc* d, int M) {
    if (winot_geometry(d, M)) return 1;
    if (wino_vw_width(d)) { int tm; return wino_vw_plan(d, M, tm); }
    const long P = (long)d->N * d->Ti * d->Hi * d->Wi;
    const int bm = 16 * wino_tm(M), C = M == d->Cout ? d->Cin : d->Cout;
    const long tiles = ((M + bm - 1) / bm) * ((P + 255) / 256), nchunks = (long)((C + 15) / 16) * 3 * d->kT;
    const char* e = ZSV_KNOB(WINO_MIN_TILES);
    const long min_tiles = e ? atol(e) : 512;
    if (tiles >= min_tiles) return 1;
    if (ZSV_KNOB(WINO_NO_SPLITK)) return 0;
    for (long ks = 2; ks <= 4; ++ks)
        if (tiles * ks >= min_tiles && nchunks / ks >= 12) return (int)ks;
    return 0;
}

// temporal 3x1x1 stride-1 "same" convolution on the F(2,3)-along-T kernel: T = 4, 8, 16 whole clips per tile, 16-byte pieces,
// one round of workgroups at least (no K parts: the K loop is only Cin / 16 chunks)
static bool winot_shape(const zsv_conv_desc* d) {
    return d->kT == 3 && d->kH == 1 && d->kW == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 0 && d->pW == 0;
}
static int winot_segs(const zsv_conv_desc* d) {
    const int PW = 256 / d->Ti;
    return (d->Hi * d->Wi + PW - 1) / PW;
}
static bool winot_geometry(const zsv_conv_desc* d, int M) {
    if (ZSV_KNOB(NO_WINO) || ZSV_KNOB(NO_WINOT) || !winot_shape(d)) return false;
    // T = 32 (round 4): only the F(4,3) kernel takes it (a wave holds two frame quads of 8 positions)
    const bool t32 = d->Ti == 32 && ZSV_KNOB(WINOT_NO_F43) == nullptr && ZSV_KNOB(WINOT_NO_T32) == nullptr;
    if ((d->Ti != 4 && d->Ti != 8 && d->Ti != 16 && !t32) || (d->Hi * d->Wi) % 4 != 0 || d->Cin < 16 || d->Cout < 16) return false;
    const long P = (long)d->N * d->Ti * d->Hi * d->Wi;
    if ((long)d->Cout * P >= (1L << 29) || (long)d->Cin * P >= (1L << 29)) return false;
    const int bm = 16 * wino_tm(M);
    const long tiles = (long)((M + bm - 1) / bm) * d->N * winot_segs(d);
    const long hw = (long)d->Hi * d->Wi, covered = (long)winot_segs(d) * (256 / d->Ti);
    const char* e1 = ZSV_KNOB(WINOT_MIN_TILES);
    const char* e2 = ZSV_KNOB(WINOT_MAX_WASTE);
    const long min_tiles = e1 ? atol(e1) : 256, max_waste = e2 ? atol(e2) : 35;      // (layer3's 14x14 maps: 196 of 256 positions, 352 tiles: still +15 %)
    return tiles >= min_tiles && covered * 100 <= hw * (100 + max_waste);     // few empty positions in the last segment
}

static bool wino_geometry(const zsv_conv_desc* d, int M) {
    if (winot_geometry(d, M)) return true;
    if (ZSV_KNOB(NO_WINO)) return false;
    if ((d->kT != 1 && d->kT != 3) || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT != d->kT / 2 ||
        d->pH != 1 || d->pW != 1)
        return false;
    if ((d->Wi % 2 != 0 && !wino_vw_width(d)) || d->Cin < 16 || d->Cout < 16) return false;
    const long P = (long)d->N * d->Ti * d->Hi * d->Wi;
    // byte offsets (4 * element index, plus tap shifts) are formed in signed 32-bit registers and the out-of-range
    // sentinels 0xFFFFFFF0 / 0xFFFFFFFF must stay >= the descriptor's num_records: tensors of < 2^29 elements
    // (2 GiB) only, like conv_wgrad_tring; larger ones take the direct kernel.
    if ((P % 2 != 0 && !wino_vw_width(d)) || (long)d->Cout * P >= (1L << 29) || (long)d->Cin * P >= (1L << 29)) return false;
    return wino_ksplit(d, M) > 0;
}

// dgrad / forward of a 1x3x3 or 3x3x3 stride-1 "same" convolution with enough voxel tiles to fill the chip
bool wino_dgrad_applicable(const zsv_conv_desc* d) { return wino_geometry(d, d->Cin); }
bool wino_fwd_applicable(const zsv_conv_desc* d) { return wino_geometry(d, d->Cout) && ZSV_KNOB(NO_WINO_FWD) == nullptr; }
int wino_fwd_stat_tiles(const zsv_conv_desc* d) {
    if (winot_geometry(d, d->Cout)) return d->N * winot_segs(d);
    if (const int wv = wino_vw_width(d)) return (int)(((long)d->N * d->Ti * d->Hi * wv + 255) / 256);
    return (int)(((long)d->N * d->Ti * d->Hi * d->Wi + 255) / 256);
}

// W % 4 == 0 (or a virtual width): the F(4,3) kernel (6 Winograd points), else F(2,3) (4 points)
static bool wino_f43(const zsv_conv_desc* d) { return (d->Wi % 4 == 0 && ZSV_KNOB(WINO_NO_F43) == nullptr) || wino_vw_width(d) != 0; }
// rows per workgroup of this launch (16 * tm)
static int wino_tm_for(const zsv_conv_desc* d, int M) {
    int tm = wino_tm(M);
    if (!winot_geometry(d, M) && wino_vw_width(d)) wino_vw_plan(d, M, tm);
    return tm;
}
static size_t wino_bytes(const zsv_conv_desc* d, int M, int C) {
    const int bm = 16 * wino_tm_for(d, M), Mp = (M + bm - 1) / bm * bm, nblk = (C + 15) / 16;
    if (winot_shape(d)) return wino_align((size_t)nblk * 6 * 16 * Mp * sizeof(float));            // (no row taps; 6 points: the F(4,3) form, the F(2,3) form uses 4)
    const int points = wino_f43(d) ? 6 : 4;
    return wino_align((size_t)nblk * 3 * d->kT * points * 16 * Mp * sizeof(float));
}
// transformed weights + (split-K) the parts' slabs
static size_t wino_total_bytes(const zsv_conv_desc* d, int M, int C) {
    const int ks = wino_ksplit(d, M);
    const size_t out_bytes = (size_t)d->N * M * d->Ti * d->Hi * d->Wi * sizeof(float);
    return wino_bytes(d, M, C) + (ks > 1 ? (size_t)ks * out_bytes : 0);
}
size_t wino_dgrad_workspace_bytes(const zsv_conv_desc* d) { return wino_total_bytes(d, d->Cin, d->Cout); }
size_t wino_fwd_workspace_bytes(const zsv_conv_desc* d) { return wino_total_bytes(d, d->Cout, d->Cin); }
// the epilogue extras (statistics, add, bias, ReLU) need the whole K in one workgroup
bool wino_dgrad_fusable(const zsv_conv_desc* d) { return wino_ksplit(d, d->Cin) == 1; }
bool wino_fwd_fusable(const zsv_conv_desc* d) { return wino_ksplit(d, d->Cout) == 1; }

template <int TM, int NCHUNKS, bool X4>
static int wino_launch(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    constexpr int LDS_BYTES = 2 * (4 * 16 * TM * 16 + 16 * 264 + 64) * 4;        // as in the kernel
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wino_kernel<TM, NCHUNKS, X4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    const long tiles = (long)p.tiles_m * p.tiles_n * p.ksplit;
    hipLaunchKernelGGL((conv_wino_kernel<TM, NCHUNKS, X4>), dim3((unsigned)tiles), dim3(256), LDS_BYTES, stream, p, up, in, out);
    return launch_status();
}

template <int TM, int NCHUNKS, bool VW, int EPI>
static int wino4_launch_epi(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    constexpr int A_FLOATS = 6 * 16 * TM * 16, IMG = 16 * 272 + 64;                           // as in the kernel
    constexpr int LDS_BYTES = (TM >= 4 ? A_FLOATS + 2 * IMG : 2 * (A_FLOATS + IMG)) * 4;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wino4_kernel<TM, NCHUNKS, VW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    const long tiles = (long)p.tiles_m * p.tiles_n * p.ksplit;
    hipLaunchKernelGGL((conv_wino4_kernel<TM, NCHUNKS, VW, EPI>), dim3((unsigned)tiles), dim3(256), LDS_BYTES, stream, p, up, in, out);
    return launch_status();
}
template <int TM, int NCHUNKS, bool VW = false>
static int wino4_launch(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    if (p.bias != nullptr && p.relu && p.add == nullptr && p.stat_sum == nullptr && ZSV_KNOB(WINO_GENERIC_EPILOGUE) == nullptr)
        return wino4_launch_epi<TM, NCHUNKS, VW, 8>(p, up, in, out, stream);
    if (p.bias != nullptr || p.relu || (p.add != nullptr && p.stat_sum != nullptr) || ZSV_KNOB(WINO_GENERIC_EPILOGUE))
        return wino4_launch_epi<TM, NCHUNKS, VW, 4>(p, up, in, out, stream);
    if (p.add != nullptr) return wino4_launch_epi<TM, NCHUNKS, VW, 2>(p, up, in, out, stream);
    return p.stat_sum != nullptr ? wino4_launch_epi<TM, NCHUNKS, VW, 1>(p, up, in, out, stream) : wino4_launch_epi<TM, NCHUNKS, VW, 0>(p, up, in, out, stream);
}

template <int TM, bool PRE>
static int winot_launch(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    constexpr int LDS_BYTES = 2 * (4 * 16 * TM * 16 + 16 * 260 + (PRE ? 256 : 0)) * 4;          // as in the kernel
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_winot_kernel<TM, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_winot_kernel<TM, PRE>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), LDS_BYTES, stream, p, up, in, out);
    return launch_status();
}

template <int TM, bool PRE, int EPI>
static int winot4_launch_epi(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    constexpr int A_FLOATS = 6 * 16 * TM * 16, IMG = 16 * 260 + (PRE ? 256 : 0);               // as in the kernel
    constexpr int LDS_BYTES = (TM >= 4 ? A_FLOATS + 2 * IMG : 2 * (A_FLOATS + IMG)) * 4;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_winot4_kernel<TM, PRE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return ZSV_E_LAUNCH;
    hipLaunchKernelGGL((conv_winot4_kernel<TM, PRE, EPI>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), LDS_BYTES, stream, p, up, in, out);
    return launch_status();
}
template <int TM, bool PRE>
static int winot4_launch(const WinoParams& p, const float* up, const float* in, float* out, hipStream_t stream) {
    if (p.add != nullptr || p.bias != nullptr || p.relu || ZSV_KNOB(WINOT_GENERIC_EPILOGUE)) return winot4_launch_epi<TM, PRE, 2>(p, up, in, out, stream);
    return p.stat_sum != nullptr ? winot4_launch_epi<TM, PRE, 1>(p, up, in, out, stream) : winot4_launch_epi<TM, PRE, 0>(p, up, in, out, stream);
}

// the temporal form: M rows from C reduction channels; G[m][c][kt] = w[m*sm + c*sc + (flip ? 2 - kt : kt)]
static int winot_run(const zsv_conv_desc* d, int M, int C, const float* in, const float* w, long sm, long sc, int flip,
                     const float* add, const float* bias, int relu, float* stat_sum, float* stat_sq, const float* pre_coef,
                     int pre_pitch, float* out,
                     void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || workspace_bytes < wino_bytes(d, M, C)) return ZSV_E_WORKSPACE;
    const int tm = wino_tm(M), bm = 16 * tm;
    WinoParams p;
    p.M = M;
    p.Mp = (M + bm - 1) / bm * bm;
    p.C = C;
    p.nblk = (C + 15) / 16;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.W = d->Wi; p.H = d->Hi; p.T = d->Ti;
    p.kT = 3; p.R = 1;
    p.P = d->N * p.S;
    p.in_bytes = 4u * (unsigned)((long)d->N * C * p.S);
    p.PW = 256 / d->Ti;
    p.pw_log2 = d->Ti == 4 ? 6 : d->Ti == 8 ? 5 : d->Ti == 16 ? 4 : 3;
    p.out_bytes = 4u * (unsigned)((long)d->N * M * p.S);
    p.segs = winot_segs(d);
    p.tiles_m = p.Mp / bm;
    p.tiles_n = d->N * p.segs;
    p.add = add; p.bias = bias; p.relu = relu; p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.ksplit = 1; p.chunks_per_split = p.nblk; p.slab_elems = 0;
    p.pre_coef = pre_coef; p.pre_pitch = pre_pitch;
    float* up;
    int pst;
    if (!panel_place(wino_bytes(d, M, C), workspace, up, pst)) return pst;
    const bool f43 = ZSV_KNOB(WINOT_NO_F43) == nullptr;          // (T is 4, 8 or 16 here: whole frame quads)
    const long total = (long)p.nblk * (f43 ? 6 : 4) * 16 * p.Mp;
    p.u_bytes = (unsigned)(total * sizeof(float));
    if (g_panel.mode == PANEL_RECORD) {
        pack_job_wino(g_panel.job, f43 ? 6 : 4, PackWinoArgs{p.M, p.Mp, p.C, p.nblk, 1, flip, sm, sc}, w, up, total);
        g_panel.jobs++;
        return ZSV_OK;
    }
    if (g_panel.mode != PANEL_LAUNCH_ONLY) {
        long pb = (total + 255) / 256;
        if (pb > 4096) pb = 4096;
        hipLaunchKernelGGL(f43 ? wino4_pack_kernel : wino_pack_kernel, dim3((unsigned)pb), dim3(256), 0, stream, w, up, p.M, p.Mp, p.C, p.nblk, 1, sm, sc, flip,
                           total);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    }
    if (f43) {
        if (pre_coef) return tm == 3 ? winot4_launch<3, true>(p, up, in, out, stream) : winot4_launch<4, true>(p, up, in, out, stream);
        return tm == 3 ? winot4_launch<3, false>(p, up, in, out, stream) : winot4_launch<4, false>(p, up, in, out, stream);
    }
    if (pre_coef) return tm == 3 ? winot_launch<3, true>(p, up, in, out, stream) : winot_launch<4, true>(p, up, in, out, stream);
    return tm == 3 ? winot_launch<3, false>(p, up, in, out, stream) : winot_launch<4, false>(p, up, in, out, stream);
}

// out[m] = sum_c G[m][c] (*) in[c]; G[m][c][tap] = w[m*sm + c*sc + (flip ? 9*kT-1 - tap : tap)]
static int wino_run(const zsv_conv_desc* d, int M, int C, const float* in, const float* w, long sm, long sc, int flip,
                    const float* add, const float* bias, int relu, float* stat_sum, float* stat_sq, float* out,
                    void* workspace, size_t workspace_bytes, hipStream_t stream) {
    float* const final_out = out;
    if (!workspace || workspace_bytes < wino_total_bytes(d, M, C)) return ZSV_E_WORKSPACE;
    const int tm = wino_tm_for(d, M), bm = 16 * tm;
    const int ks = wino_ksplit(d, M);
    if (ks < 1) return ZSV_E_UNSUPPORTED;
    if (ks > 1 && (add != nullptr || stat_sum != nullptr)) return ZSV_E_UNSUPPORTED;
    WinoParams p;
    p.M = M;
    p.Mp = (M + bm - 1) / bm * bm;
    p.C = C;
    p.nblk = (C + 15) / 16;
    p.S = d->Ti * d->Hi * d->Wi; p.HW = d->Hi * d->Wi; p.W = d->Wi; p.H = d->Hi; p.T = d->Ti;
    p.kT = d->kT; p.R = 3 * d->kT;
    p.P = d->N * p.S;
    p.in_bytes = 4u * (unsigned)((long)d->N * C * p.S);
    p.tiles_m = p.Mp / bm;
    p.out_bytes = 4u * (unsigned)((long)d->N * M * p.S);
    p.Wv = wino_vw_width(d);
    p.wv_log2 = 0;
    while ((1 << p.wv_log2) < p.Wv) ++p.wv_log2;
    p.m_S = make_magic((unsigned)p.S); p.m_HW = make_magic((unsigned)p.HW); p.m_W = make_magic((unsigned)p.W);
    p.m_TH = make_magic((unsigned)(p.T * p.H)); p.m_H = make_magic((unsigned)p.H);
    p.m_tiles_m = make_magic((unsigned)p.tiles_m); p.m_ksplit = make_magic((unsigned)ks);
    p.Pv = p.Wv ? d->N * d->Ti * d->Hi * p.Wv : p.P;
    p.tiles_n = (p.Pv + 255) / 256;
    p.add = add; p.bias = bias; p.relu = relu; p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.ksplit = ks;
    p.chunks_per_split = (p.nblk * p.R + ks - 1) / ks;
    p.slab_elems = (long)d->N * M * p.S;
    float* up;
    int pst;
    if (!panel_place(wino_bytes(d, M, C), workspace, up, pst)) return pst;
    float* slabs = (float*)((char*)workspace + wino_bytes(d, M, C));
    if (ks > 1) { p.bias = nullptr; p.relu = 0; out = slabs; }         // bias / ReLU move to the ordered sum of the parts
    const bool f43 = wino_f43(d);
    p.u_bytes = (unsigned)wino_bytes(d, M, C);
    const long total = (long)p.nblk * p.R * (f43 ? 6 : 4) * 16 * p.Mp;
    if (g_panel.mode == PANEL_RECORD) {
        pack_job_wino(g_panel.job, f43 ? 6 : 4, PackWinoArgs{p.M, p.Mp, p.C, p.nblk, p.R, flip, sm, sc}, w, up, total);
        g_panel.jobs++;
        return ZSV_OK;
    }
    if (g_panel.mode != PANEL_LAUNCH_ONLY) {
        long pb = (total + 255) / 256;
        if (pb > 4096) pb = 4096;
        hipLaunchKernelGGL(f43 ? wino4_pack_kernel : wino_pack_kernel, dim3((unsigned)pb), dim3(256), 0, stream, w, up, p.M, p.Mp, p.C,
                           p.nblk, p.R, sm, sc, flip, total);
        if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    }
    const bool x4 = d->Wi % 4 == 0 && ZSV_KNOB(WINO_NO_X4) == nullptr;
    int st;
    if (p.Wv) {
        st = tm == 3 ? wino4_launch<3, 0, true>(p, up, in, out, stream) : wino4_launch<4, 0, true>(p, up, in, out, stream);
    } else if (f43) {
        if (tm == 3) st = (p.nblk * p.R == 12 && ks == 1) ? wino4_launch<3, 12>(p, up, in, out, stream) : wino4_launch<3, 0>(p, up, in, out, stream);
        else st = wino4_launch<4, 0>(p, up, in, out, stream);
    } else if (tm == 3) {
        if (x4) st = (p.nblk * p.R == 12 && ks == 1) ? wino_launch<3, 12, true>(p, up, in, out, stream) : wino_launch<3, 0, true>(p, up, in, out, stream);
        else st = wino_launch<3, 0, false>(p, up, in, out, stream);
    } else {
        st = x4 ? wino_launch<4, 0, true>(p, up, in, out, stream) : wino_launch<4, 0, false>(p, up, in, out, stream);
    }
    if (st || ks == 1) return st;
    return splitk_reduce(slabs, ks, p.slab_elems, M, p.S, bias, relu, final_out, stream);
}

int wino_dgrad(const zsv_conv_desc* d, const float* dy, const float* w, const float* add, float* dx, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
    if (winot_geometry(d, d->Cin))          // G[m = ci][c = co][kt] = W[co][ci][2 - kt]
        return winot_run(d, d->Cin, d->Cout, dy, w, 3, (long)d->Cin * 3, 1, add, nullptr, 0, nullptr, nullptr, nullptr, 0, dx, workspace,
                         workspace_bytes, stream);
    // G[m = ci][c = co][kt][kh][kw] = W[co][ci][kT-1-kt][2-kh][2-kw]: stride of m is 9*kT, of c is Cin*9*kT, taps flipped
    const long taps = 9L * d->kT;
    return wino_run(d, d->Cin, d->Cout, dy, w, taps, (long)d->Cin * taps, 1, add, nullptr, 0, nullptr, nullptr, dx, workspace,
                    workspace_bytes, stream);
}

bool wino_fwd_pre_capable(const zsv_conv_desc* d) { return winot_geometry(d, d->Cout) && ZSV_KNOB(NO_WINO_FWD) == nullptr; }

// the temporal forward with the BatchNorm + ReLU in front of it applied on the fly (conv_winot_kernel PRE)
int wino_fwd_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int pre_pitch, const float* w, float* stat_sum,
                 float* stat_sq, float* y, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!wino_fwd_pre_capable(d)) return ZSV_E_UNSUPPORTED;
    return winot_run(d, d->Cout, d->Cin, x, w, (long)d->Cin * 3, 3, 0, nullptr, nullptr, 0, stat_sum, stat_sq, pre_coef, pre_pitch, y, workspace,
                     workspace_bytes, stream);
}

int wino_fwd(const zsv_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual, int relu,
             float* stat_sum, float* stat_sq, float* y, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (winot_geometry(d, d->Cout))
        return winot_run(d, d->Cout, d->Cin, x, w, (long)d->Cin * 3, 3, 0, residual, bias, relu, stat_sum, stat_sq, nullptr, 0, y,
                         workspace, workspace_bytes, stream);
    const long taps = 9L * d->kT;
    return wino_run(d, d->Cout, d->Cin, x, w, (long)d->Cin * taps, taps, 0, residual, bias, relu, stat_sum, stat_sq, y, workspace,
                    workspace_bytes, stream);
}

}  // namespace zsv
