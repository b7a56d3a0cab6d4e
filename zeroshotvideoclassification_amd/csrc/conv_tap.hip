// conv_tap.hip -- implicit-GEMM convolution (forward and dgrad) for gfx950, the hot kernel.
//
// Same GEMM view as conv_igemm.hip -- C[m][p] = sum_k A[m][k] * B[k][p], rows = channels
// produced, columns = voxels (n, t, h, w), MFMA v_mfma_f32_16x16x4_f32 (bit-exact fp32) -- but
// the reduction index is ordered (16-channel block, tap, channel-in-block) and a K-chunk is
// "one tap x 16 consecutive channels".  Consequences:
//
//   * gathered slab B: all 16 rows of a chunk share the tap, so a lane tests its voxel's
//     tap-validity bit ONCE per chunk and builds one voffset (or 0xFFFFFFFF, which the buffer
//     descriptor's range check turns into the zero padding: no branch, no select); the 16 rows
//     differ only by channel, which goes into the buffer load's scalar offset operand.  The 9
//     taps of a channel block are consecutive chunks, so its rows stay hot in L1/L2 (HBM read
//     traffic is 1.05-1.07x the compulsory bytes, profiles/r01_s1_hbm_traffic.json).
//   * weight panel A: the weights are re-packed once per call into Wp[block][tap][16][m]
//     (m contiguous, padded to whole tiles, zero-filled), so a chunk's 16 x BM panel is a plain
//     2-D block.  Packing reads + writes the weight tensor once (<= 42 MB, ~10 us).
//   * both operands go global -> LDS by LDS-DMA (`buffer_load_dword ... lds` per 64-voxel k-row
//     segment, `global_load_lds_dwordx4` for the panel): no staging VGPRs, no ds_write; the
//     kernel fits 4 waves per SIMD (<= 128 VGPRs) and its only LDS instructions are the MFMA
//     fragment reads, fetched one k-step ahead of the MFMA chain that consumes them.
//   * epilogue: when the output is voxel-contiguous for the launch, the BM x BN tile is transposed
//     through the (now free) staging LDS and stored as whole 512-B channel rows with 16-B lanes;
//     the same pass can emit BatchNorm partial statistics (sum, sum of squares per channel and
//     column tile) so the following BatchNorm needs no pass of its own over the activations.
//   * launches that would under-fill the 1024 resident workgroups (layer3/4: 2x7x7 voxels) cut the
//     K range into parts; parts write partial slabs that a second kernel adds in fixed order.
//
// Used for every layer whose gathered tensor has >= 16 channels and at most 31 taps; the
// 3-channel 7x7 stems stay on conv_igemm.hip.
#include <stdlib.h>
#include "conv_params.h"
#include "pack_bodies.h"
#include "knobs.h"

namespace zsv {

// Wp[((cb * taps + tap) * 16 + c % 16) * Mp + m] = W[m * w_m_stride + c * w_c_stride + tap_full(tap)], cb = c / 16
// (0 in the padding)
__global__ __launch_bounds__(256) void pack_weights_kernel(IgemmParams prm, const float* __restrict__ W,
                                                           float* __restrict__ Wp, int w_m_stride, int w_c_stride,
                                                           int Cpad, int Mp, long total) {
    const PackTapArgs a = pack_tap_args(prm, w_m_stride, w_c_stride, Cpad, Mp);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) Wp[i] = pack_tap_value(a, W, i);
}

// The same packing with coalesced traffic on both sides.  pack_weights_kernel walks Wp (m fastest) and so reads W at
// stride w_m_stride: 4 useful bytes per fetched sector for a forward weight tensor (18 KB between neighbouring threads on
// layer4; 1.6 GB of HBM reads per step for 0.25 GB of weights, 22-41 us per layer4 call).  Here a block owns a
// [16-channel block] x [MT output rows] tile with all its taps: it reads W along whichever of its two axes is contiguous
// (forward: (c, tap) runs of 16*taps floats per row m; gradient: (m, tap) runs per channel c), parks the tile in LDS as
// [tap][c][m] and writes Wp rows of MT consecutive m.  Same values, same layout.
template <int MT, int JN>       // JN: 64-lane passes over a run (16 * taps <= 64 JN and MT * taps <= 64 JN)
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(IgemmParams prm, const float* __restrict__ W,
                                                                 float* __restrict__ Wp, int w_m_stride, int w_c_stride,
                                                                 int Cpad, int Mp, int rows_total) {
    extern __shared__ float tile[];                       // [taps * 16][MT + 1]: row = tap * 16 + channel-in-block
    constexpr int LD = MT + 1;
    const int cb = blockIdx.x, m0 = blockIdx.y * MT;
    const int taps = prm.taps;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool t2 = prm.t2_cin != 0;
    const bool c_contig = t2 ? prm.dir > 0 : w_c_stride < w_m_stride;        // forward: (c, tap) contiguous for a fixed m
    const bool cb_live = cb * 16 < Cpad;
    auto tap_full = [&](int tap) {
        const int jt = tap / prm.nHW;
        const int r = tap - jt * prm.nHW;
        const int jh = r / prm.nW;
        const int jw = r - jh * prm.nW;
        return ((prm.k0T + prm.tsT * jt) * prm.kH + prm.k0H + prm.tsH * jh) * prm.kW + prm.k0W + prm.tsW * jw;
    };
    // The (second index, tap) decomposition of a lane's run elements does not depend on the outer index: done once.  All loads
    // of a wave's share are issued before the first LDS store (the kernel is a latency chain otherwise).
    int off[JN], dst[JN], idx[JN];                        // off: -2 beyond the run, -1 padding (zero), else element offset
    if (c_contig) {                                       // a wave per row m; lanes walk the row's (c, tap) run
        const int run = 16 * taps;
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int r = lane + 64 * j;
            off[j] = -2; dst[j] = 0; idx[j] = 0;
            if (r < run) {
                const int cl = r / taps, tap = r - cl * taps, c = cb * 16 + cl;
                idx[j] = c;
                dst[j] = (tap * 16 + cl) * LD;
                off[j] = (c < prm.gC && cb_live) ? c * w_c_stride + tap_full(tap) : -1;
            }
        }
        constexpr int RW = MT / 4;                        // rows per wave
        float v[RW][JN];
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int m = m0 + wave + 4 * u;
            const float* src = W + (size_t)m * w_m_stride;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                v[u][j] = 0.f;
                if (off[j] >= 0 && m < prm.M) v[u][j] = t2 ? W[t2_weight_offset(m, idx[j], prm.t2_cin)] : src[off[j]];
            }
        }
#pragma unroll
        for (int u = 0; u < RW; ++u)
#pragma unroll
            for (int j = 0; j < JN; ++j)
                if (off[j] != -2) tile[dst[j] + wave + 4 * u] = v[u][j];
    } else {                                              // a wave per reduction channel c; lanes walk its (m, tap) run
        const int run = MT * taps;
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int r = lane + 64 * j;
            off[j] = -2; dst[j] = 0; idx[j] = 0;
            if (r < run) {
                const int ml = r / taps, tap = r - ml * taps, m = m0 + ml;
                idx[j] = m;
                dst[j] = tap * 16 * LD + ml;
                off[j] = m < prm.M ? m * w_m_stride + tap_full(tap) : -1;
            }
        }
        float v[4][JN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cl = wave + 4 * u, c = cb * 16 + cl;
            const float* src = W + (size_t)c * w_c_stride;
            const bool live = c < prm.gC && cb_live;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                v[u][j] = 0.f;
                if (off[j] >= 0 && live) v[u][j] = t2 ? W[t2_weight_offset(c, idx[j], prm.t2_cin)] : src[off[j]];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < JN; ++j)
                if (off[j] != -2) tile[dst[j] + (wave + 4 * u) * LD] = v[u][j];
    }
    __syncthreads();
    // rows of MT consecutive m: Wp row = (cb * taps + tap) * 16 + cl = cb * taps * 16 + LDS row
    const int nrows = taps * 16;
    const long row0 = (long)cb * nrows;
    for (int e = threadIdx.x; e < nrows * MT; e += 256) {
        const int ml = e % MT, r = e / MT;                // (MT is a power of two)
        if (row0 + r < rows_total && m0 + ml < Mp) Wp[(row0 + r) * Mp + m0 + ml] = tile[r * LD + ml];
    }
}

// ---------------------------------------------------------------------------------------------
// Staging: the gathered slab and the weight panel go global -> LDS directly
// (`buffer_load_dword ... lds` per k-row, `global_load_lds_dwordx4` for the panel): no staging
// VGPRs, no ds_write, so the register budget drops under 128 (4 waves per SIMD) and the only
// LDS instructions left are the MFMA fragment reads.  The LDS destination of a DMA is
// wave-uniform base + lane * size, which is exactly a k-row segment of 64 voxels; the panel is
// written as a linear image of the padded [16][LDA] array (pad slots load a dummy element).
// One chunk is in flight: issued right after the barrier, awaited (vmcnt(0)) before the next one.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef ZSV_TAP_INTERLEAVE
#define ZSV_TAP_INTERLEAVE 1
#endif
#ifndef WAVES_PER_EU
#define WAVES_PER_EU 4
#endif

// NBLK_CT / TAPS_CT: compile-time channel-block and tap counts (0 = run time).  The one
// specialisation (4, 9) is the network's dominant layer, Conv3d(64, 144, (1,3,3)) forward: its
// chunk walk is fully constant-folded, and it gets a kernel symbol of its own in profiles.
// PRE: the gathered tensor is the INPUT of a BatchNorm + ReLU that was never materialised (zsv_bn_fwd_train_coeffs): a B
// fragment value becomes relu(g * scale[c] + shift[c]) on its way into the MFMA -- the same fmaf and max as the BatchNorm
// apply pass, so results are bit-identical to reading the materialised activation.  Per chunk wave 0 DMAs the 16 channels'
// scale / shift next to the operands; zero padding must stay zero, so every lane keeps the tap-validity word of its TN
// fragment columns (exchanged through LDS once) and zeroes the padded taps' values instead.
template <int TM, int TN, int WGM, int WGN, int FK, int NBLK_CT = 0, int TAPS_CT = 0, bool PRE = false>     // FK = MFMA k-steps per fragment burst
__global__ __launch_bounds__(256, WAVES_PER_EU) void conv_tap_dma_kernel(IgemmParams prm, const float* __restrict__ Wp,
                                                           const float* __restrict__ G, const float* __restrict__ bias,
                                                           float* __restrict__ C, int tiles_m, int Mp, int nblk) {
#if defined(__HIP_DEVICE_COMPILE__)     // (address_space(3) casts: the host pass would drop the stub)
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    constexpr int BK = 16;
    constexpr int LDA = LdPad<BM>::value;
    constexpr int LDB = LdPad<BN>::value;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    static_assert(BN == 64 || BN == 128 || BN == 256, "a k-row is split into 64-voxel wave segments");
    constexpr int WPR = BN / 64;                 // waves per k-row
    constexpr int RPP = 4 / WPR;                 // k-rows per pass
    constexpr int BPASS = BK / RPP;
    constexpr int ASLOTS = BK * LDA / 4;         // float4 slots of the padded panel image
    constexpr int APASS = (ASLOTS + 255) / 256;
    constexpr unsigned OOB = 0xFFFFFFFFu;

    // one LDS pool: As[2] | Bs[2] during the main loop, re-used by the transposing epilogue
    __shared__ __attribute__((aligned(16))) float pool[2 * BK * (LDA + LDB)];
    __shared__ int tapoff[32];
    __shared__ __attribute__((aligned(16))) float pre_lds[PRE ? 2 * 256 : 4];   // per stage: 1 KiB DMA target, 32 floats used
    __shared__ unsigned vmask_lds[PRE ? BN : 1];
    float (*As)[BK * LDA] = reinterpret_cast<float (*)[BK * LDA]>(pool);
    float (*Bs)[BK * LDB] = reinterpret_cast<float (*)[BK * LDB]>(pool + 2 * BK * LDA);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);
    const int lin = xcd_tile(gridDim.x, blockIdx.x);
    const int ntiles = gridDim.x / prm.ksplit;
    const int tile = lin % ntiles;
    const int split = lin / ntiles;          // which part of the K range (split-K for small grids)
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, prm.g_bytes, 0x00020000);

    if (tid < prm.taps) {
        const int jt = tid / prm.nHW;
        const int r = tid - jt * prm.nHW;
        const int jh = r / prm.nW;
        const int jw = r - jh * prm.nW;
        tapoff[tid] = 4 * prm.dir * (jt * prm.gHW + jh * prm.gW + jw);
    }

    // ---- this lane's voxel: column (wave % WPR) * 64 + lane of the tile --------------------
    const int bcol0 = (wave % WPR) * 64;         // wave-uniform
    const int brow0 = wave / WPR;
    int base_bytes = 0;
    unsigned vmask = 0;
    {
        const int p = n0 + bcol0 + lane;
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
            int tap = 0;
            for (int a = 0; a < prm.nT; ++a)
                for (int b = 0; b < prm.nH; ++b)
                    for (int c = 0; c < prm.nW; ++c, ++tap)
                        vmask |= (((mt >> a) & (mh >> b) & (mw >> c)) & 1u) << tap;
        }
    }

    if constexpr (PRE) {
        if (brow0 == 0) vmask_lds[bcol0 + lane] = vmask;        // (the waves of the first k-row cover the BN columns once)
    }

    // ---- weight panel: per pass, the source of this lane's float4 slot of the padded image --
    const float* a_src[APASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
        const int slot = 64 * (wave + 4 * j) + lane;
        const int r = (slot / (LDA / 4)) % BK, c4 = slot % (LDA / 4);
        a_src[j] = Wp + (size_t)r * Mp + m0 + (c4 < BM / 4 ? 4 * c4 : 0);      // pad slots re-read column 0
    }
    const size_t a_chunk_stride = (size_t)BK * Mp;
    const int ch_bytes = 4 * prm.gS;
    const __amdgpu_buffer_rsrc_t pre_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? prm.pre_coef : G), 0, PRE ? 8u * (unsigned)prm.pre_pitch : 0u, 0x00020000);

    const int taps_ = TAPS_CT ? TAPS_CT : prm.taps;
    const int nblk_ = NBLK_CT ? NBLK_CT : nblk;
    const int nchunks_all = (prm.K > 0) ? taps_ * nblk_ : 0;
    const int per_split = (nchunks_all + prm.ksplit - 1) / prm.ksplit;
    const int c_begin = split * per_split;
    const int nchunks = max(0, min(nchunks_all, c_begin + per_split) - c_begin);
    int ld_cb = c_begin / taps_, ld_tap = c_begin - ld_cb * taps_;
    // The DMA instructions of a chunk: BPASS image rows, APASS panel pieces, (PRE, wave 0) the scales / shifts -- NDMA per wave.  In
    // the chunk loop they go out one at a time between the MFMAs of the chunk before (ZSV_TAP_INTERLEAVE; as conv_wino.hip's F(4,3)
    // kernels, where the burst at the head of a chunk cost a third of the chunk: profiles/r03_temporal_phase_trace.txt).
    constexpr int NDMA = BPASS + APASS + (PRE ? 1 : 0);
    constexpr bool IL = ZSV_TAP_INTERLEAVE && TM * TN <= 16 && !(PRE && TM * TN == 16);     // (the 144-row tiles have no registers to spare: 128-VGPR budget)
    unsigned ic_voff = 0;
    int ic_ci0 = 0;
    auto issue_begin = [&]() {
        const int toff = sgpr(tapoff[ld_tap]);
        const unsigned ok = (vmask >> ld_tap) & 1u;
        ic_voff = (unsigned)(base_bytes + toff) | (ok - 1u);
        ic_ci0 = ld_cb * 16;
        if (++ld_tap == taps_) { ld_tap = 0; ++ld_cb; }
    };
    auto issue_piece = [&](int chunk, int buf, int j) {
        if (j < BPASS) {
            float* bdst = &Bs[buf][brow0 * LDB + bcol0];
            const int ci = ic_ci0 + brow0 + RPP * j;
            const unsigned v = ci < prm.gC ? ic_voff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rsrc, (lds_ptr_t)(bdst + RPP * j * LDB), 4, (int)v, ci < prm.gC ? ci * ch_bytes : 0, 0, 0);
        } else if (j < BPASS + APASS) {
            const int q = j - BPASS;
            if (64 * (wave + 4 * q) < ASLOTS)          // wave-uniform
                __builtin_amdgcn_global_load_lds(a_src[q] + (size_t)(c_begin + chunk) * a_chunk_stride,
                                                 (lds_ptr_t)(&As[buf][256 * (wave + 4 * q)]), 16, 0, 0);
        } else if constexpr (PRE) {
            if (wave == 0) {        // lanes 0..3: scale[ci0 .. ci0+15], lanes 4..7: shift[...]; the others read out of range (zeros)
                const unsigned off = lane < 8 ? 4u * (unsigned)((lane >> 2) * prm.pre_pitch + ic_ci0 + 4 * (lane & 3)) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pre_rsrc, (lds_ptr_t)(&pre_lds[buf * 256]), 16, (int)off, 0, 0, 0);
            }
        }
    };
    auto issue_chunk = [&](int chunk, int buf) {
        issue_begin();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_piece(chunk, buf, j);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                      // tapoff visible
    if (nchunks > 0) issue_chunk(0, 0);
    __syncthreads();                      // (drains the DMA: vmcnt(0) before the barrier)

    const int frag_row = lane >> 4;
    const int frag_col = lane & 15;
    constexpr int NS = (BK / 4) / FK;     // fragment bursts per chunk
    unsigned vmf[PRE ? TN : 1];           // PRE: tap-validity words of this lane's fragment columns
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < TN; ++j) vmf[j] = vmask_lds[wn0 + 16 * j + frag_col];      // (published before the barriers above)
    }
    int cur_tap = c_begin - (c_begin / taps_) * taps_;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        const bool prefetching = ch + 1 < nchunks;
        if (prefetching) {
            if (IL) issue_begin();
            else issue_chunk(ch + 1, cur ^ 1);
        }
        const float* as = &As[cur][0];
        const float* bs = &Bs[cur][0];
        float psc[PRE ? BK / 4 : 1], psh[PRE ? BK / 4 : 1];
        bool pok[PRE ? TN : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int q = 0; q < BK / 4; ++q) {                 // this lane's k rows: 4 q + frag_row
                psc[q] = pre_lds[cur * 256 + 4 * q + frag_row];
                psh[q] = pre_lds[cur * 256 + 16 + 4 * q + frag_row];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) pok[j] = (vmf[j] >> cur_tap) & 1u;
            if (++cur_tap == taps_) cur_tap = 0;
        }
        float a[2][FK][TM], b[2][FK][TN];
        auto fetch = [&](int s_, int slot) {
#pragma unroll
            for (int kk = 0; kk < FK; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[slot][kk][i] = as[((FK * s_ + kk) * 4 + frag_row) * LDA + wm0 + 16 * i + frag_col];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[slot][kk][j] = bs[((FK * s_ + kk) * 4 + frag_row) * LDB + wn0 + 16 * j + frag_col];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (s_ + 1 < NS) fetch(s_ + 1, (s_ + 1) & 1);
            if constexpr (PRE) {
#pragma unroll
                for (int kk = 0; kk < FK; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float t = fmaxf(__fmaf_rn(b[s_ & 1][kk][j], psc[FK * s_ + kk], psh[FK * s_ + kk]), 0.f);
                        b[s_ & 1][kk][j] = pok[j] ? t : 0.f;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);          // the MFMA burst outranks the other waves' staging / address work (+0.5 %)
#pragma unroll
            for (int kk = 0; kk < FK; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_ & 1][kk][i], b[s_ & 1][kk][j], acc[i][j], 0, 0, 0);
                        if (IL) {          // one DMA instruction of the next chunk after every EV-th MFMA
                            constexpr int TOTAL = NS * FK * TM * TN, EV = TOTAL / NDMA > 0 ? TOTAL / NDMA : 1;
                            const int idx = ((s_ * FK + kk) * TM + i) * TN + j;
                            if (TOTAL >= NDMA && idx % EV == EV - 1 && idx / EV < NDMA && prefetching) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(ch + 1, cur ^ 1, idx / EV);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
            if (IL && s_ == NS - 1 && NS * FK * TM * TN < NDMA && prefetching) {      // (tiny tiles: the rest in a row)
#pragma unroll
                for (int q = 0; q < NDMA; ++q) issue_piece(ch + 1, cur ^ 1, q);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    IgemmParams q = prm;
    const float* ebias = bias;
    float* Cout = C;
    if (prm.ksplit > 1) {                    // raw partial sums; bias / ReLU are applied by the slab reduce
        q.relu = 0;
        ebias = nullptr;
        Cout = C + (size_t)split * prm.slab_elems;
    }
    if constexpr (BN == 128 || BN == 256) {
        if (prm.lds_epilogue) {              // voxel-contiguous output: full-line stores through LDS
            store_tiles_lds<TM, TN, WGM, WGN>(q, acc, pool, 2 * BK * (LDA + LDB), m0, n0, wave, lane, ebias, Cout);
            return;
        }
    }
    store_tiles<TM, TN>(q, acc, m0 + wm0, n0 + wn0, lane, ebias, Cout);
#endif
}

// C = sum_s slab[s] (+ bias[m]) (ReLU): fixed order, deterministic
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int ksplit, long elems, int M,
                                                            int oS, const float* __restrict__ bias, int relu,
                                                            float* __restrict__ C, bool vec4) {
    if (vec4) {
        // four consecutive voxels per thread (oS % 4 == 0: they share their channel) and the loads of four parts issued before their
        // adds -- per element the same order of addition as the scalar loop, the same bits
        const long quads = elems >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) {
            const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + i;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 3 < ksplit; s += 4) {
                const f32x4 a = src[(size_t)s * quads], b = src[(size_t)(s + 1) * quads], c = src[(size_t)(s + 2) * quads],
                            d = src[(size_t)(s + 3) * quads];
                v += a; v += b; v += c; v += d;
            }
            for (; s < ksplit; ++s) v += src[(size_t)s * quads];
            if (bias) { const float bv = bias[((4 * i) / oS) % M]; v[0] += bv; v[1] += bv; v[2] += bv; v[3] += bv; }
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            reinterpret_cast<f32x4*>(C)[i] = v;
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < elems; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < ksplit; ++s) v += slabs[(size_t)s * elems + i];
        if (bias) v += bias[(i / oS) % M];
        if (relu) v = fmaxf(v, 0.f);
        C[i] = v;
    }
}

int splitk_reduce(const float* slabs, int ksplit, long elems, int M, int oS, const float* bias, int relu, float* C,
                  hipStream_t stream) {
    const bool vec4 = oS % 4 == 0 && elems % 4 == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(C)) & 15) == 0 &&
                      ZSV_KNOB(SPLITK_REDUCE_SCALAR) == nullptr;
    long blocks = ((vec4 ? elems / 4 : elems) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, ksplit, elems, M, oS, bias,
                       relu, C, vec4);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
struct TapCfg { int bm, bn; };
static const TapCfg kTapCfgs[] = {{144, 128}, {128, 128}, {80, 128}, {64, 128}, {48, 256}, {64, 256}};
static const int kNumTapCfgs = 6;

static long tap_ksplit_for(long tiles, long nchunks) {
    // 1024 workgroups are resident at once (256 CUs x 4); split K when one round is under-filled
    long thresh = 700, target = 1024;
    if (ZSV_KNOB(NO_SPLITK)) return 1;
    if (const char* e = ZSV_KNOB(TAP_KS)) return atol(e) < 1 ? 1 : (atol(e) > nchunks ? nchunks : atol(e));      // (sweeps)
    if (const char* e = ZSV_KNOB(SPLITK_THRESH)) thresh = atol(e);
    if (tiles >= thresh || nchunks < 32) return 1;
    long ks = (target + tiles / 2) / tiles;              // nearest whole multiple of the tile count
    if (ks > nchunks / 12) ks = nchunks / 12;            // >= 12 chunks per part
    if (ks > 16) ks = 16;
    return ks < 2 ? 1 : ks;
}

static int tap_pick(const IgemmParams& prm) {
    // Cost model, fitted on a sweep of every configuration over the R(2+1)D-18 layers at N = 22
    // (tools/conv_bench.py with ZSV_CONV_CFG=0..5; within 0.5 % of the per-layer best overall):
    // padded MACs x a per-shape efficiency factor x
    //   - above one round of the 1024 resident workgroups: a partial last round costs ~30 % of what
    //     whole-round accounting would charge (slots are refilled as they free up);
    //   - below one round: (1024 / workgroups)^0.4 -- fewer co-resident workgroups each run faster,
    //     so an under-filled chip loses less than proportionally, but more (smaller) tiles still win.
    // Workgroups = tiles x the split-K factor the launch will use.
    static const double penalty[kNumTapCfgs] = {1.00, 1.00, 1.02, 1.04, 1.15, 0.97};
    // (64x256: 64x64 wave tiles, half the fragment reads per MFMA of 64x128 -- +5 % on S1 dgrad / T1 forward
    //  even at 3 waves/SIMD; a 144x256 tile drops to 1 wave/SIMD and loses 15 %)
    int best = 0;
    double best_w = 1e300;
    const long nchunks = (long)prm.taps * ((prm.gC + 15) / 16);
    for (int i = 0; i < kNumTapCfgs; ++i) {
        // 64x256 only where one row tile covers the problem (M <= 64, the 1.1 M-voxel layer1 / stem
        // launches): on wider outputs the sweep found it better on some mid-size layers and worse on as many
        if (i == 5 && (prm.M > 64 || ZSV_KNOB(NO_CFG5))) continue;
        const long tm = (prm.M + kTapCfgs[i].bm - 1) / kTapCfgs[i].bm;
        const long tn = ((long)prm.P + kTapCfgs[i].bn - 1) / kTapCfgs[i].bn;
        const double tiles = (double)(tm * tn);
        const double wgs = tiles * (double)tap_ksplit_for(tm * tn, nchunks);
        double w = tiles * kTapCfgs[i].bm * kTapCfgs[i].bn * penalty[i];
        if (wgs > 1024.0) {
            const double r = wgs / 1024.0, rc = (double)(long)((wgs + 1023.0) / 1024.0);
            w *= 1.0 + 0.3 * (rc - r) / r;
        } else {
            w *= pow(1024.0 / wgs, 0.4);
        }
        if (w < best_w * 0.999) { best_w = w; best = i; }
    }
    if (const char* e = ZSV_KNOB(CONV_CFG)) best = atoi(e) % kNumTapCfgs;
    return best;
}

bool igemm_tap_applicable(const IgemmParams& prm) {
    if (ZSV_KNOB(NO_TAP)) return false;
    return prm.gC >= 16 && prm.taps <= 31 && prm.K > 0;
}

static inline void tap_layout(const IgemmParams& prm, int& cfg, int& tiles_m, int& Mp, int& nblk, int& Cpad) {
    cfg = tap_pick(prm);
    tiles_m = (prm.M + kTapCfgs[cfg].bm - 1) / kTapCfgs[cfg].bm;
    Mp = tiles_m * kTapCfgs[cfg].bm;
    nblk = (prm.gC + 15) / 16;
    Cpad = nblk * 16;
}

size_t igemm_tap_workspace_bytes(const IgemmParams& prm) {
    int cfg, tiles_m, Mp, nblk, Cpad;
    tap_layout(prm, cfg, tiles_m, Mp, nblk, Cpad);
    return ((size_t)prm.taps * Cpad + 16) * Mp * sizeof(float);      // +16 rows: trailing half-chunk reads
}

template <int TM, int TN, int WGM, int WGN>
static int tap_launch(const IgemmParams& prm, const float* Wp, const float* G, const float* bias, float* C,
                      int tiles_m, int Mp, int nblk, hipStream_t stream) {
    constexpr int BN = 16 * TN * WGN;
    const long blocks = (long)tiles_m * (((long)prm.P + BN - 1) / BN) * (prm.ksplit > 1 ? prm.ksplit : 1);
    if (blocks <= 0 || blocks > 0x7fffffffL) return ZSV_E_TOO_LARGE;
    if constexpr (TM == 9 && TN == 2) {
        if (prm.dir == 1 && prm.taps == 9 && nblk == 4 && prm.ksplit <= 1 && prm.pre_coef == nullptr) {
            hipLaunchKernelGGL((conv_tap_dma_kernel<TM, TN, WGM, WGN, 1, 4, 9>), dim3((unsigned)blocks), dim3(256), 0,
                               stream, prm, Wp, G, bias, C, tiles_m, Mp, nblk);
            return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
        }
    }
    if (prm.pre_coef != nullptr) {
        hipLaunchKernelGGL((conv_tap_dma_kernel<TM, TN, WGM, WGN, 1, 0, 0, true>), dim3((unsigned)blocks), dim3(256), 0, stream, prm,
                           Wp, G, bias, C, tiles_m, Mp, nblk);
        return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
    }
    hipLaunchKernelGGL((conv_tap_dma_kernel<TM, TN, WGM, WGN, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, prm, Wp, G,
                       bias, C, tiles_m, Mp, nblk);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

static bool tap_lds_epilogue_ok(const IgemmParams& prm, const float* C) {
    return prm.stW == 1 && prm.stH == 1 && prm.stT == 1 && prm.oS % 4 == 0 && prm.cS == prm.oS &&
           (reinterpret_cast<uintptr_t>(C) & 15) == 0 && !ZSV_KNOB(NO_LDS_EPILOGUE);
}

int igemm_tap_stat_tiles(const IgemmParams& prm, const float* C) {
    if (!igemm_tap_applicable(prm) || !tap_lds_epilogue_ok(prm, C) || ZSV_KNOB(NO_FUSED_STATS)) return 0;
    if (igemm_tap_ksplit(prm) > 1) return 0;
    int cfg, tiles_m, Mp, nblk, Cpad;
    tap_layout(prm, cfg, tiles_m, Mp, nblk, Cpad);
    if (kTapCfgs[cfg].bn != 128 && kTapCfgs[cfg].bn != 256) return 0;
    return (int)(((long)prm.P + kTapCfgs[cfg].bn - 1) / kTapCfgs[cfg].bn);
}

int igemm_tap_ksplit(const IgemmParams& prm) {
    int cfg, tiles_m, Mp, nblk, Cpad;
    tap_layout(prm, cfg, tiles_m, Mp, nblk, Cpad);
    const long tiles = (long)tiles_m * (((long)prm.P + kTapCfgs[cfg].bn - 1) / kTapCfgs[cfg].bn);
    return (int)tap_ksplit_for(tiles, (long)prm.taps * nblk);
}

int igemm_tap(const IgemmParams& prm_in, const float* W, int w_m_stride, int w_c_stride, const float* G,
              const float* bias, float* C, void* workspace, size_t workspace_bytes, float* slabs, hipStream_t stream) {
    int cfg, tiles_m, Mp, nblk, Cpad;
    tap_layout(prm_in, cfg, tiles_m, Mp, nblk, Cpad);
    IgemmParams prm = prm_in;
    if (prm.ksplit < 1 || !slabs) prm.ksplit = 1;
    // the transposing epilogue needs 4 consecutive voxels of a launch column group to be 4
    // consecutive, 16-B aligned floats of one clip
    prm.lds_epilogue = tap_lds_epilogue_ok(prm, C) ? 1 : 0;
    if (!prm.lds_epilogue || prm.ksplit > 1) { prm.stat_sum = nullptr; prm.stat_sq = nullptr; }
    if (prm.ksplit > 1) { C = slabs; prm.acc_src = nullptr; }   // partial slabs instead of the output tensor (callers check)
    const size_t need = ((size_t)prm.taps * Cpad + 16) * Mp * sizeof(float);
    if (!workspace || workspace_bytes < need) return ZSV_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return ZSV_E_WORKSPACE;
    float* Wp;
    int pst;
    if (!panel_place(need, workspace, Wp, pst)) return pst;
    const long total = ((long)prm.taps * Cpad + 16) * Mp;           // the 16 extra rows are zero
    const int rows_total = prm.taps * Cpad + 16;
    if (g_panel.mode == PANEL_RECORD) {
        pack_job_tap(g_panel.job, pack_tap_args(prm, w_m_stride, w_c_stride, Cpad, Mp), W, Wp, total);
        g_panel.jobs++;
        return ZSV_OK;
    }
    if (g_panel.mode == PANEL_LAUNCH_ONLY) {
        // (the caller's panel was packed by zsv_pack_multi from this call's job)
    } else if ((long)prm.M * prm.gC * prm.taps >= 65536 && prm.taps <= 27 && !ZSV_KNOB(NO_PACK_TILED)) {
        // (+1 channel block: the 16 trailing zero rows; c >= gC packs zeros)
        const int mt = prm.taps <= 9 ? 32 : 16;
        const dim3 grid((unsigned)(nblk + 1), (unsigned)((Mp + mt - 1) / mt));
        const size_t lds = (size_t)prm.taps * 16 * (mt + 1) * sizeof(float);
        if (mt == 32)
            hipLaunchKernelGGL((pack_weights_tiled_kernel<32, 5>), grid, dim3(256), lds, stream, prm, W, Wp, w_m_stride, w_c_stride, Cpad, Mp, rows_total);
        else
            hipLaunchKernelGGL((pack_weights_tiled_kernel<16, 7>), grid, dim3(256), lds, stream, prm, W, Wp, w_m_stride, w_c_stride, Cpad, Mp, rows_total);
    } else {
        long pb = (total + 255) / 256;
        if (pb > 4096) pb = 4096;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)pb), dim3(256), 0, stream, prm, W, Wp, w_m_stride, w_c_stride,
                           Cpad, Mp, total);
    }
    if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    switch (cfg) {
        case 0: return tap_launch<9, 2, 1, 4>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
        case 1: return tap_launch<4, 4, 2, 2>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
        case 2: return tap_launch<5, 2, 1, 4>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
        case 3: return tap_launch<4, 2, 1, 4>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
        case 4: return tap_launch<3, 4, 1, 4>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
        default: return tap_launch<4, 4, 1, 4>(prm, Wp, G, bias, C, tiles_m, Mp, nblk, stream);
    }
}

}  // namespace zsv
