// conv_wgrad.hip -- weight gradient of the 3-D convolution for gfx950 (MI355X).
//
// Supplies aten::convolution_backward's grad_weight for every Conv3d of the reference
// (resnet.py:23-30,40-52,63-70,170,181,184,270; network.py:102-117) and, with S = 1,
// the weight gradient of nn.Linear (network.py:611-616,120,132).
//
//   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p shifted by tap],   p = (n, to, ho, wo)
//
// A GEMM whose reduction runs over output voxels p (1.1 M for layer1) and whose output
// (Cout x Cin*taps) is small.  MI355X mapping:
//   * fp32 matrix core v_mfma_f32_16x16x4_f32 (bit-exact fp32); rows = Cout, columns = k';
//   * columns are ordered tap-major, k' = tap * Cpad + ci (Cpad = Cin rounded up to 16), so a
//     16-column block shares one tap: a lane evaluates the padding test once per block and
//     chunk, the 16 channels of the block go into the buffer load's scalar offset;  voxel ->
//     (n, t, h, w) uses multiply-high "magic" division (3 VALU per divide);
//   * the voxel range is cut into `slices` contiguous ranges (grid.y) so that a launch has
//     ~1.5k workgroups although the output has a handful of tiles; every slice writes a
//     partial slab to the caller's workspace and a second kernel adds the slabs in slice
//     order while un-permuting k' -> (ci, tap): bitwise reproducible, no float atomics;
//   * dY rows are contiguous along the voxel axis: 16-B loads when S % 4 == 0;
//   * 32 voxels per chunk; LDS images As[BM][34], Bs[BN][34] (34 == 2 mod 32): the MFMA
//     operand fetch (16 rows x 2 voxels per 32-lane group) and the staging stores are both
//     bank-conflict free;  pipeline per chunk c: [regs of c+1 -> LDS][loads of c+2 -> regs]
//     [MFMAs of c, fragments fetched one 8-voxel group ahead] barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "conv_params.h"
#include "knobs.h"

namespace zsv {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WgradParams {
    int M;                  // Cout
    int Cin, Cpad, nblk;    // channels, padded to 16, 16-channel blocks per tap
    int taps, kHW, kW;
    int Kp;                 // taps * Cpad: slab row length
    int P;                  // N * oS voxels
    int oS, oHW, oW;        // dY geometry
    Magic m_oS, m_oHW, m_oW;
    int gT, gH, gW, gS, gHW, gCS;    // x geometry (gCS = Cin * gS)
    int sT, sH, sW, pT, pH, pW;
    int chunks_per_slice;   // 32-voxel chunks handled by one slice
    unsigned x_bytes, dy_bytes;
};

// TWOTAP: every column tile of the launch spans at most two taps (true for all layers with >= 64
// input channels): the padding test + gather offset are evaluated per TAP (2 per chunk) instead of
// per 16-column block (8 per chunk).
template <int TM, int TN, int WGM, int WGN, bool AV4, int BP, bool TWOTAP>      // BP = voxels per chunk (32 or 16)
__global__ __launch_bounds__(256, (BP == 16 ? 3 : 2)) void conv_wgrad_kernel(WgradParams prm, const float* __restrict__ X,
                                                         const float* __restrict__ DY,
                                                         float* __restrict__ OUT, int tiles_m, int tiles_mn) {
    constexpr int BM = 16 * TM * WGM;
    constexpr int BN = 16 * TN * WGN;
    constexpr int LDK = BP + 2;
    constexpr int NBLK = BN / 16;          // 16-column blocks of this tile
    constexpr int RW = 64 / BP;            // rows covered by one wave-instruction (2 or 4)
    constexpr int RPP = 4 * RW;            // rows per staging pass (8 or 16)
    constexpr int PB = 16 / RPP;           // passes per 16-row block (2 or 1)
    constexpr int BPASS = NBLK * PB;       // gathered rows staged per thread
    constexpr int AQ = BP / 4;             // float4 per dY row (8 or 4)
    constexpr int ARP = 256 / AQ;          // dY rows per float4 pass (32 or 64)
    constexpr int APASS = AV4 ? (BM + ARP - 1) / ARP : BM / RPP;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(BM % RPP == 0 || AV4, "tile rows must be a multiple of the staging pass");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ float Bs[2][BN * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int wm0 = (wave / WGN) * (16 * TM);
    const int wn0 = (wave % WGN) * (16 * TN);
    // 1-D grid, XCD-aware: the tiles of one slice (same dY rows, same x voxels) run on one XCD
    const int lin = xcd_tile(gridDim.x, blockIdx.x);
    const int tile = lin % tiles_mn;
    const int slice = lin / tiles_mn;
    const int m0 = (tile % tiles_m) * BM;
    const int n0 = (tile / tiles_m) * BN;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, prm.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DY), 0, prm.dy_bytes, 0x00020000);

    // ---- the tile's 16-column blocks: tap (kt, kh, kw) and first channel, all wave-uniform ----
    int b_kt[NBLK], b_kh[NBLK], b_kw[NBLK], b_ci0[NBLK];
    bool b_second[NBLK];                                 // TWOTAP: block uses the tile's second tap
    bool has_tail = false;
    const int tap_first = min((n0 / 16) / prm.nblk, prm.taps - 1);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int blk = n0 / 16 + b;
        int tap = blk / prm.nblk;
        const int cb = blk - tap * prm.nblk;
        const bool live = tap < prm.taps;
        tap = live ? tap : 0;
        const int kt = tap / prm.kHW;
        const int r = tap - kt * prm.kHW;
        const int kh = r / prm.kW;
        b_kt[b] = kt; b_kh[b] = kh; b_kw[b] = r - kh * prm.kW;
        b_ci0[b] = live ? cb * 16 : prm.Cin;            // dead block: every row is beyond Cin
        b_second[b] = live && tap != tap_first;
        has_tail = has_tail || (b_ci0[b] + 16 > prm.Cin);
    }
    // the tile's (at most) two taps
    int t2_kt[2], t2_kh[2], t2_kw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int tap = min(tap_first + q, prm.taps - 1);
        t2_kt[q] = tap / prm.kHW;
        const int r = tap - t2_kt[q] * prm.kHW;
        t2_kh[q] = r / prm.kW;
        t2_kw[q] = r - t2_kh[q] * prm.kW;
    }

    // gather lanes: voxel column pcol, two rows per wave (row parity = half of the wave)
    const int pcol = tid & (BP - 1);
    const int half = (tid / BP) % RW;      // which of the wave's RW rows this lane stages
    // dY lanes (AV4): 4 consecutive voxels, ARP rows per pass
    const int aq = tid % AQ, arow = tid / AQ;

    const int chunk_begin = slice * prm.chunks_per_slice;
    int chunk_end = chunk_begin + prm.chunks_per_slice;
    const int total_chunks = (prm.P + BP - 1) / BP;
    if (chunk_end > total_chunks) chunk_end = total_chunks;

    float breg[BPASS];
    f32x4 areg4[AV4 ? APASS : 1];
    float areg[AV4 ? 1 : APASS];
    const int ch_bytes = 4 * prm.gS;
    const int row_bytes = 4 * prm.oS;

    // The loads of chunk c+2 are issued in NG parts, one inside each MFMA group of chunk c, so their
    // address arithmetic and issue slots hide under this wave's own MFMAs (an MFMA holds the issue
    // port for 8 of its 32 cycles).  Part 0 decodes the chunk's voxels; the state lives in registers.
    bool ld_pv = false;
    int ld_t0 = 0, ld_h0 = 0, ld_w0 = 0, ld_xb = 0;
    unsigned ld_dyb = OOB;
    unsigned ld_voff2[2] = {OOB, OOB};      // TWOTAP: gather offset (or OOB) of this voxel for the tile's two taps
    const float* ld_abase = DY;
    auto load_decode = [&](int chunk) {
        const unsigned p = (unsigned)(chunk * BP + pcol);
        ld_pv = p < (unsigned)prm.P;
        const unsigned n = mdiv(p, prm.m_oS);
        const unsigned r0 = p - n * prm.oS;
        const unsigned ot = mdiv(r0, prm.m_oHW);
        const unsigned r1 = r0 - ot * prm.oHW;
        const unsigned oh = mdiv(r1, prm.m_oW);
        const unsigned ow = r1 - oh * prm.oW;
        ld_t0 = (int)ot * prm.sT - prm.pT; ld_h0 = (int)oh * prm.sH - prm.pH; ld_w0 = (int)ow * prm.sW - prm.pW;
        ld_xb = 4 * ((int)n * prm.gCS + ld_t0 * prm.gHW + ld_h0 * prm.gW + ld_w0) + half * ch_bytes;
        if (TWOTAP) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = ld_pv && (unsigned)(ld_t0 + t2_kt[q]) < (unsigned)prm.gT &&
                                (unsigned)(ld_h0 + t2_kh[q]) < (unsigned)prm.gH && (unsigned)(ld_w0 + t2_kw[q]) < (unsigned)prm.gW;
                ld_voff2[q] = ok ? (unsigned)(ld_xb + 4 * (t2_kt[q] * prm.gHW + t2_kh[q] * prm.gW + t2_kw[q])) : OOB;
            }
        }
        if (AV4) {
            // 4 voxels of one clip (oS % 4 == 0); clamp instead of masking: rows >= M are never
            // stored, voxels >= P meet zeros from the gathered operand
            unsigned p4 = (unsigned)(chunk * BP + 4 * aq);
            if (p4 + 4 > (unsigned)prm.P) p4 = (unsigned)prm.P - 4;
            const unsigned n4 = mdiv(p4, prm.m_oS);
            ld_abase = DY + (size_t)n4 * prm.M * prm.oS + (p4 - n4 * prm.oS);
        } else {
            ld_dyb = ld_pv ? 4u * (n * (unsigned)prm.M * (unsigned)prm.oS + r0) + (unsigned)(half * row_bytes) : OOB;
        }
    };
    auto load_b = [&](int b) {          // gathered rows of 16-column block b
        unsigned voff;
        if (TWOTAP) {
            voff = b_second[b] ? ld_voff2[1] : ld_voff2[0];
        } else {
            const bool ok = ld_pv && (unsigned)(ld_t0 + b_kt[b]) < (unsigned)prm.gT && (unsigned)(ld_h0 + b_kh[b]) < (unsigned)prm.gH &&
                            (unsigned)(ld_w0 + b_kw[b]) < (unsigned)prm.gW;
            voff = ok ? (unsigned)(ld_xb + 4 * (b_kt[b] * prm.gHW + b_kh[b] * prm.gW + b_kw[b])) : OOB;
        }
        // rows of block b handled by this thread: ci0 + RW*wave + half + RPP*jj
#pragma unroll
        for (int jj = 0; jj < PB; ++jj) {
            const int ci_u = b_ci0[b] + RW * wave + RPP * jj;       // wave-uniform part
            unsigned v = voff;
            int soff = ci_u * ch_bytes;
            if (has_tail) {
                if (ci_u + half >= prm.Cin) v = OOB;
                if (ci_u >= prm.Cin) soff = 0;
            }
            breg[PB * b + jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)v, soff, 0));
        }
    };
    auto load_a = [&](int j) {          // dY rows of staging pass j
        if (AV4) {
            int row = m0 + (arow + ARP * j) % BM;
            row = row < prm.M ? row : prm.M - 1;
            areg4[j] = *reinterpret_cast<const f32x4*>(ld_abase + (size_t)row * prm.oS);
        } else {
            const int row_u = m0 + RW * wave + RPP * j;           // wave-uniform part of the row
            unsigned v = (row_u + half < prm.M) ? ld_dyb : OOB;
            areg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, (int)v, row_u < prm.M ? row_u * row_bytes : 0, 0));
        }
    };
    // part g of NGP: an even share of the blocks and of the dY passes
    auto load_part = [&](int chunk, int g, int ngp) {
        if (g == 0) load_decode(chunk);
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
            if (b * ngp / NBLK == g) load_b(b);
#pragma unroll
        for (int j = 0; j < APASS; ++j)
            if (j * ngp / APASS == g) load_a(j);
    };
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) load_part(chunk, g, 4);
    };

    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int jj = 0; jj < PB; ++jj)
                Bs[buf][(16 * b + RW * wave + half + RPP * jj) * LDK + pcol] = breg[PB * b + jj];
        if (AV4) {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                float* dst = &As[buf][((arow + ARP * j) % BM) * LDK + 4 * aq];     // 8-B aligned (LDK even)
                *reinterpret_cast<float2*>(dst) = make_float2(areg4[j][0], areg4[j][1]);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(areg4[j][2], areg4[j][3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < APASS; ++j) As[buf][(RW * wave + half + RPP * j) * LDK + pcol] = areg[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (chunk_begin < chunk_end) {
        load_chunk(chunk_begin);
        store_chunk(0);
        if (chunk_begin + 1 < chunk_end) load_chunk(chunk_begin + 1);
    }
    __syncthreads();

    const int frag_k = lane >> 4;
    const int frag_r = lane & 15;
    constexpr int NG = BP / 8;             // 8-voxel groups per chunk (2 MFMA k-steps each)
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        const int cur = (ch - chunk_begin) & 1;
        if (ch + 1 < chunk_end) store_chunk(cur ^ 1);
        // always issue the loads (clamped chunk index in the last two iterations): a branch around
        // them makes hipcc drain vmcnt(0) at every join and serialises the parts
        const int lchunk = min(ch + 2, chunk_end - 1);
        const float* as = &As[cur][0];
        const float* bs = &Bs[cur][0];
        float a[2][2][TM], b[2][2][TN];
        auto fetch = [&](int g, int slot) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[slot][kk][i] = as[(wm0 + 16 * i + frag_r) * LDK + (2 * g + kk) * 4 + frag_k];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[slot][kk][j] = bs[(wn0 + 16 * j + frag_r) * LDK + (2 * g + kk) * 4 + frag_k];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);      // next group's ds_reads stay ahead of this group's MFMAs
            load_part(lchunk, g, NG);               // same scheduling region as the MFMAs below
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][kk][i], b[g & 1][kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // partial slab of this slice: OUT[slice][m][k'] (k' tap-major)
    float* out = OUT + (size_t)slice * prm.M * prm.Kp;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = n0 + wn0 + 16 * j + frag_r;
        if (k >= prm.Kp) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm0 + 16 * i + 4 * frag_k + r;
                if (m < prm.M) out[(size_t)m * prm.Kp + k] = acc[i][j][r];
            }
        }
    }
}

// dw[co][ci][tap] = sum_s slab[s][co][tap * Cpad + ci].  A block owns 32 consecutive slab elements (ci
// fastest: coalesced reads) x 8 slice groups: group g adds slices g, g+8, ... in order, the 8 partials
// are added in group order -- a fixed order, so bitwise reproducible, with 8x the loads in flight of a
// one-thread-per-element loop (the ~250 slices of a layer1 gradient made that loop latency-bound).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                       int M, int Cin, int taps, int Cpad, int slices) {
    __shared__ float part[8][32];
    const size_t slab = (size_t)M * taps * Cpad;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (size_t j0 = (size_t)blockIdx.x * 32; j0 < slab; j0 += (size_t)gridDim.x * 32) {
        const size_t j = j0 + e;
        const int ci = (int)(j % Cpad);
        const bool live = j < slab && ci < Cin;
        float s = 0.f;
        if (live) {
            const float* src = slabs + j;
            int k = g;
            for (; k + 24 < slices; k += 32) {                    // four loads in flight, added in the same order
                const float v0 = src[(size_t)k * slab], v1 = src[(size_t)(k + 8) * slab];
                const float v2 = src[(size_t)(k + 16) * slab], v3 = src[(size_t)(k + 24) * slab];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; k < slices; k += 8) s += src[(size_t)k * slab];
        }
        part[g][e] = s;
        __syncthreads();
        if (g == 0 && live) {
            float t = part[0][e];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += part[q][e];
            const size_t r = j / Cpad;
            const int tap = (int)(r % taps);
            const int co = (int)(r / taps);
            out[((size_t)co * Cin + ci) * taps + tap] = t;
        }
        __syncthreads();
    }
}

// The same sum for FEW slices and a large slab (the small-voxel layers: 2-4 slices of a 17-21 MB gradient): one block per
// output channel reads its [taps][Cpad] row of every slice with 16-byte loads (coalesced), adds the slices in order (up to 8
// slices that is also slab_sum_kernel's order: the same bits) and writes dw[co][ci][tap] through an LDS
// transpose as whole contiguous rows (slab_sum_kernel's 4-byte writes at stride `taps` made it run at 1 TB/s).
__global__ __launch_bounds__(256) void slab_sum_rows_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                            int M, int Cin, int taps, int Cpad, int slices) {
    extern __shared__ __attribute__((aligned(16))) float row[];          // [taps][Cpad]
    const int co = blockIdx.x;
    const int len = taps * Cpad;                                        // (Cpad % 16 == 0: whole float4s)
    const size_t slab = (size_t)M * len;
    const float* src = slabs + (size_t)co * len;
    for (int i = 4 * (int)threadIdx.x; i < len; i += 1024) {
        f32x4 t = *reinterpret_cast<const f32x4*>(src + i);
        int k = 1;
        for (; k + 2 < slices; k += 3) {                          // three slices' loads in flight, added in the same order
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)k * slab + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 1) * slab + i);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 2) * slab + i);
            t += v0; t += v1; t += v2;
        }
        for (; k < slices; ++k) t += *reinterpret_cast<const f32x4*>(src + (size_t)k * slab + i);
        *reinterpret_cast<f32x4*>(row + i) = t;
    }
    __syncthreads();
    float* dst = out + (size_t)co * Cin * taps;
    const int n = Cin * taps;
    for (int o = (int)threadIdx.x; o < n; o += 256) {
        const int ci = o / taps, tap = o - ci * taps;
        dst[o] = row[tap * Cpad + ci];
    }
}

// launches the slab sum that fits the geometry (same result bits either way)
int slab_sum(const float* slabs, float* dw, int M, int Cin, int taps, int Cpad, int slices, hipStream_t stream) {
    const size_t row_bytes = (size_t)taps * Cpad * sizeof(float);
    if (slices <= 32 && (long)M * slices <= 16384 && row_bytes <= 64 * 1024 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && !ZSV_KNOB(NO_SLAB_SUM_ROWS)) {
        static const hipError_t attr = hipFuncSetAttribute((const void*)slab_sum_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (attr != hipSuccess) return ZSV_E_LAUNCH;
        hipLaunchKernelGGL(slab_sum_rows_kernel, dim3((unsigned)M), dim3(256), row_bytes, stream, slabs, dw, M, Cin, taps, Cpad, slices);
        return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
    }
    const long n = (long)M * taps * Cpad;
    long blocks = (n + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, dw, M, Cin, taps, Cpad, slices);
    return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
}

static int wgrad_bp() { return 32; }

// the dense gradient dW'[2 Cout][2 Cin] of a two-frame temporal convolution (conv_params.h: t2_dense_shape) folded back
// onto its three taps: kt = ti - to + 1, so tap 0 <- (to 1, ti 0), tap 1 <- (0, 0) + (1, 1), tap 2 <- (0, 1)
__global__ __launch_bounds__(256) void t2_fold_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int Cout, int Cin) {
    const long n = (long)Cout * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long co = i / Cin, c = i - co * Cin;
        const float* r0 = dw2 + (2 * co) * (2L * Cin) + 2 * c;          // row (co, to = 0): columns (c, ti = 0), (c, ti = 1)
        const float* r1 = r0 + 2L * Cin;                               // row (co, to = 1)
        const f32x2 a = *reinterpret_cast<const f32x2*>(r0), b = *reinterpret_cast<const f32x2*>(r1);
        dw[3 * i] = b[0];
        dw[3 * i + 1] = a[0] + b[1];
        dw[3 * i + 2] = a[1];
    }
}

struct WgradPlan {
    int cfg;        // 0: 144x128, 1: 128x128, 2: 64x128, 3: 80x128
    int bm, bn;
    int tiles_m, tiles_n, slices, chunks_per_slice;
    int Cpad, nblk, Kp;
};

// Slice count for `tiles` output tiles: one full round of resident workgroups (256 CUs x the
// LDS-limited 2-4 per CU) -- every workgroup then runs start to finish concurrently and the slab
// traffic is minimal.  Returns the round-fill efficiency of the best count (fewer slices on ties).
static double wgrad_slices(long tiles, long chunks, int bp, int bm, int bn, long& slices) {
    const long lds_bytes = (long)(bm + bn) * 34 * 4 * 2;
    long per_cu = (160L * 1024) / lds_bytes;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const long resident = 256L * per_cu;
    long max_slices = (chunks * bp + 511) / 512;         // at least 512 voxels per slice
    if (max_slices < 1) max_slices = 1;
    if (max_slices > 1024) max_slices = 1024;
    slices = 1;
    double best_eff = -1.0;
    for (long s = 1; s <= max_slices && tiles * s <= 4 * resident + tiles; ++s) {
        const long wgs = tiles * s;
        const long rounds = (wgs + resident - 1) / resident;
        const double eff = (double)wgs / (double)(rounds * resident) - 0.02 * (double)wgs / (double)resident;
        if (eff > best_eff + 1e-9) { best_eff = eff; slices = s; }
    }
    return best_eff;
}

static WgradPlan wgrad_plan(const zsv_conv_desc* d) {
    WgradPlan pl;
    const int M = d->Cout;
    const int taps = d->kT * d->kH * d->kW;
    pl.nblk = (d->Cin + 15) / 16;
    pl.Cpad = pl.nblk * 16;
    pl.Kp = taps * pl.Cpad;
    const long P = (long)d->N * d->To * d->Ho * d->Wo;
    const int bp = wgrad_bp();
    const long chunks = (P + bp - 1) / bp;
    // Tile = the (rows, columns) pair with the least estimated time: padded MACs x a per-shape factor
    // (narrow tiles issue fewer MFMAs per fragment read) / round-fill efficiency.  Factors fitted on a
    // sweep of all 8 pairs over the R(2+1)D-18 layers at N = 22 (within 0.1 % of the per-layer best).
    const int bms[4] = {144, 128, 64, 80};
    const double pen_m[4] = {1.00, 1.00, 1.05, 1.08};
    const int bns[2] = {128, 64};
    const double pen_n[2] = {1.00, 1.10};
    int best_m = 0, best_n = 0;
    double best_w = 1e300;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) {
            const long tm = (M + bms[i] - 1) / bms[i], tn = (pl.Kp + bns[j] - 1) / bns[j];
            long sl;
            const double eff = wgrad_slices(tm * tn, chunks, bp, bms[i], bns[j], sl);
            const double w = (double)(tm * bms[i]) * (double)(tn * bns[j]) * pen_m[i] * pen_n[j] / (eff > 1e-3 ? eff : 1e-3);
            if (w < best_w * 0.999) { best_w = w; best_m = i; best_n = j; }
        }
    if (const char* e = ZSV_KNOB(WGRAD_CFG)) best_m = atoi(e) & 3;
    pl.cfg = best_m;
    pl.bm = bms[best_m];
    pl.bn = bns[best_n];
    if (const char* e = ZSV_KNOB(WGRAD_BN)) pl.bn = atoi(e) == 64 ? 64 : 128;
    pl.tiles_m = (M + pl.bm - 1) / pl.bm;
    pl.tiles_n = (pl.Kp + pl.bn - 1) / pl.bn;
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    long slices;
    wgrad_slices(tiles, chunks, bp, pl.bm, pl.bn, slices);
    const long max_slices = ((chunks * bp + 511) / 512) < 1 ? 1 : ((chunks * bp + 511) / 512 > 1024 ? 1024 : (chunks * bp + 511) / 512);
    if (const char* e = ZSV_KNOB(WGRAD_WGS)) slices = atol(e) / tiles;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    pl.chunks_per_slice = (int)((chunks + slices - 1) / slices);
    pl.slices = (int)((chunks + pl.chunks_per_slice - 1) / pl.chunks_per_slice);
    return pl;
}

template <int TM, int TN, int WGM, int WGN>
static void wgrad_launch(const WgradParams& p, bool av4, bool twotap, dim3 grid, hipStream_t stream, const float* x,
                         const float* dy, float* out, int tiles_m, int tiles_mn) {
#define ZSV_WG(A, T) hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, WGM, WGN, A, 32, T>), grid, dim3(256), 0, stream, p, x, dy, out, tiles_m, tiles_mn)
    if (av4) { if (twotap) ZSV_WG(true, true); else ZSV_WG(true, false); }
    else { if (twotap) ZSV_WG(false, true); else ZSV_WG(false, false); }
#undef ZSV_WG
}

// every column tile of `bn` columns spans at most two taps?
static bool wgrad_two_taps(int taps, int nblk, int bn) {
    const int nb = bn / 16, total = taps * nblk;
    for (int b0 = 0; b0 < total; b0 += nb) {
        const int last = (b0 + nb - 1 < total ? b0 + nb - 1 : total - 1);
        if (last / nblk - b0 / nblk > 1) return false;
    }
    return true;
}

}  // namespace zsv

using namespace zsv;

static bool t2_dense(const zsv_conv_desc* d) { return t2_dense_shape(d) && !ZSV_KNOB(NO_T2_DENSE); }
static size_t t2_dw_bytes(const zsv_conv_desc* d) { return ((size_t)4 * d->Cout * d->Cin * sizeof(float) + 255) & ~(size_t)255; }

extern "C" size_t zsv_conv3d_wgrad_workspace_bytes(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK) return 0;
    if (t2_dense(d)) {                  // [dense gradient dW'] [workspace of the dense 1x1x1 problem]
        const zsv_conv_desc d2 = t2_dense_desc(d);
        return t2_dw_bytes(d) + zsv_conv3d_wgrad_workspace_bytes(&d2);
    }
    if (d->Cin < 16) return wgrad_generic_workspace_bytes(d);
    const WgradPlan pl = wgrad_plan(d);
    size_t need = (size_t)pl.slices * d->Cout * pl.Kp * sizeof(float);
    if (wgrad_dma_applicable(d, nullptr, nullptr)) {        // (the pointer alignment decides at call time)
        const size_t b = wgrad_dma_workspace_bytes(d);
        if (b > need) need = b;
    }
    if (wgrad_wino_applicable(d, nullptr, nullptr)) {
        const size_t b = wgrad_wino_workspace_bytes(d);
        if (b > need) need = b;
    }
    if (wgrad_tring_applicable(d, nullptr, nullptr)) {
        const size_t b = wgrad_tring_workspace_bytes(d);
        if (b > need) need = b;
    }
    return need;
}

extern "C" int zsv_conv3d_wgrad_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch,
                                    const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream_) {
    int st = conv_check(d);
    if (st) return st;
    if (!x || !dy || !dw || !pre_coef) return ZSV_E_NULL;
    if (!zsv_conv3d_pre_supported(d) || coef_pitch < d->Cin || !wgrad_tring_applicable(d, x, dy)) return ZSV_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < zsv_conv3d_wgrad_workspace_bytes(d)) return ZSV_E_WORKSPACE;
    return wgrad_tring_pre(d, x, pre_coef, coef_pitch, dy, dw, workspace, workspace_bytes, (hipStream_t)stream_);
}

static int wgrad_dispatch(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                          const unsigned* vm_ext, void* stream_);

extern "C" int zsv_conv3d_wgrad(const zsv_conv_desc* d, const float* x, const float* dy, float* dw,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    return wgrad_dispatch(d, x, dy, dw, workspace, workspace_bytes, nullptr, stream_);
}

// The two LDS-DMA weight-gradient kernels (Winograd-form 3x3 taps, plain "same" convolutions) read a per-voxel tap-validity
// table that depends on the geometry only; they used to rebuild it on every call (13 launches per R(2+1)D-18 step in front of
// the kernels of the weight-gradient queue).  A caller that keeps the table passes it here.
static bool wgrad_takes_mask(const zsv_conv_desc* d) {
    if (conv_check(d) != ZSV_OK || t2_dense(d) || d->Cin < 16) return false;
    if (wgrad_wino_applicable(d, nullptr, nullptr)) return true;
    if (wgrad_tring_applicable(d, nullptr, nullptr)) return false;
    return wgrad_dma_applicable(d, nullptr, nullptr);
}

extern "C" size_t zsv_conv3d_wgrad_mask_bytes(const zsv_conv_desc* d) {
    return d != nullptr && wgrad_takes_mask(d) ? sizeof(unsigned) * (size_t)d->Ti * d->Hi * d->Wi : 0;
}

extern "C" int zsv_conv3d_wgrad_mask(const zsv_conv_desc* d, void* mask, void* stream_) {
    if (d == nullptr || mask == nullptr) return ZSV_E_NULL;
    if (!wgrad_takes_mask(d)) return ZSV_E_UNSUPPORTED;
    return wgrad_vmask(d, (unsigned*)mask, (hipStream_t)stream_);
}

extern "C" int zsv_conv3d_wgrad_masked(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, const void* mask, void* stream_) {
    if (d == nullptr) return ZSV_E_NULL;
    return wgrad_dispatch(d, x, dy, dw, workspace, workspace_bytes, wgrad_takes_mask(d) ? (const unsigned*)mask : nullptr, stream_);
}

static int wgrad_dispatch(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                          const unsigned* vm_ext, void* stream_) {
    int st = conv_check(d);
    if (st) return st;
    if (!x || !dy || !dw) return ZSV_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    if (t2_dense(d)) {
        const zsv_conv_desc d2 = t2_dense_desc(d);
        const size_t head = t2_dw_bytes(d);
        if (!workspace || workspace_bytes < head + zsv_conv3d_wgrad_workspace_bytes(&d2)) return ZSV_E_WORKSPACE;
        float* dw2 = (float*)workspace;
        st = zsv_conv3d_wgrad(&d2, x, dy, dw2, (char*)workspace + head, workspace_bytes - head, stream_);
        if (st) return st;
        long blocks = ((long)d->Cout * d->Cin + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(t2_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dw2, dw, d->Cout, d->Cin);
        return hipGetLastError() == hipSuccess ? ZSV_OK : ZSV_E_LAUNCH;
    }
    if (d->Cin < 16) return wgrad_generic(d, x, dy, dw, workspace, workspace_bytes, stream);
    const WgradPlan pl = wgrad_plan(d);
    const size_t need = zsv_conv3d_wgrad_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) return ZSV_E_WORKSPACE;
    if (wgrad_wino_applicable(d, x, dy)) return wgrad_wino(d, x, dy, dw, workspace, workspace_bytes, stream, vm_ext);
    if (wgrad_tring_applicable(d, x, dy)) return wgrad_tring(d, x, dy, dw, workspace, workspace_bytes, stream);
    if (wgrad_dma_applicable(d, x, dy)) {
        int slices = 0, cpad = 0;
        st = wgrad_dma(d, x, dy, workspace, workspace_bytes, &slices, &cpad, stream, vm_ext);
        if (st) return st;
        return slab_sum((const float*)workspace, dw, d->Cout, d->Cin, d->kT * d->kH * d->kW, cpad, slices, stream);
    }

    WgradParams p;
    p.M = d->Cout;
    p.Cin = d->Cin; p.Cpad = pl.Cpad; p.nblk = pl.nblk;
    p.taps = d->kT * d->kH * d->kW;
    p.kHW = d->kH * d->kW; p.kW = d->kW;
    p.Kp = pl.Kp;
    p.P = d->N * d->To * d->Ho * d->Wo;
    p.oS = d->To * d->Ho * d->Wo; p.oHW = d->Ho * d->Wo; p.oW = d->Wo;
    p.m_oS = make_magic((unsigned)p.oS); p.m_oHW = make_magic((unsigned)p.oHW); p.m_oW = make_magic((unsigned)p.oW);
    p.gT = d->Ti; p.gH = d->Hi; p.gW = d->Wi;
    p.gS = d->Ti * d->Hi * d->Wi; p.gHW = d->Hi * d->Wi; p.gCS = d->Cin * p.gS;
    p.sT = d->sT; p.sH = d->sH; p.sW = d->sW; p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
    p.chunks_per_slice = pl.chunks_per_slice;
    p.x_bytes = 4u * (unsigned)((long)d->N * d->Cin * p.gS);
    p.dy_bytes = 4u * (unsigned)((long)d->N * d->Cout * p.oS);

    const bool av4 = (p.oS % 4 == 0) && (p.P >= 4) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
    const bool twotap = wgrad_two_taps(p.taps, pl.nblk, pl.bn) && !ZSV_KNOB(WGRAD_NO_TWOTAP);
    const int tiles_mn = pl.tiles_m * pl.tiles_n;
    const dim3 grid((unsigned)(tiles_mn * pl.slices));
    float* out = (float*)workspace;
    if (pl.bn == 64) {
        switch (pl.cfg) {
            case 0: wgrad_launch<9, 1, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
            case 1: wgrad_launch<8, 1, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
            case 2: wgrad_launch<4, 1, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
            default: wgrad_launch<5, 1, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
        }
    } else
    switch (pl.cfg) {
        case 0: wgrad_launch<9, 2, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
        case 1: wgrad_launch<4, 4, 2, 2>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
        case 2: wgrad_launch<4, 2, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
        default: wgrad_launch<5, 2, 1, 4>(p, av4, twotap, grid, stream, x, dy, out, pl.tiles_m, tiles_mn); break;
    }
    if (hipGetLastError() != hipSuccess) return ZSV_E_LAUNCH;
    return slab_sum((const float*)workspace, dw, p.M, p.Cin, p.taps, pl.Cpad, pl.slices, stream);
}

extern "C" size_t zsv_linear_wgrad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features) {
    zsv_conv_desc d = {rows, in_features, 1, 1, 1, out_features, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    return zsv_conv3d_wgrad_workspace_bytes(&d);
}

extern "C" int zsv_linear_wgrad(const float* x, const float* dy, float* dw, int32_t rows, int32_t in_features,
                                int32_t out_features, void* workspace, size_t workspace_bytes, void* stream) {
    zsv_conv_desc d = {rows, in_features, 1, 1, 1, out_features, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    return zsv_conv3d_wgrad(&d, x, dy, dw, workspace, workspace_bytes, stream);
}
